// gfx950 device code of the batched DDP solver, LANE MAPPING "QUAD": for problems with n <= 4 states and one input
// (cart-pole, bipedal), the BACKWARD pass runs on the fp64 matrix cores with sixteen lanes per instance, and its
// linearisation is parallel over the horizon; the forward pass and the solver state machine are the two-wave kernel's
// (ddp_kernels_2w.hpp: one lane per instance, master + helper wavefront).
//
//   workgroup    = 16 instances, 4 wavefronts (256 threads); 4096 instances = 256 workgroups = one per CU, one
//                  wavefront per SIMD — the 2-wave kernel keeps 128 of the chip's 1024 SIMDs busy on that batch
//   backward     every wavefront owns 4 instances.  v_mfma_f64_4x4x4_4b_f64 computes four independent 4x4x4 products
//                per instruction, one per "block" of 16 lanes; lane l holds entry (row l / 16, column l % 4) of block
//                (l / 4) % 4 for the B and C/D operands and the transposed entry for A (scripts/ubench_mfma_f64_4x4.hip),
//                so mfma(X, Y, C) = X^T Y + C on registers in that "natural" layout.  One timestep of the Riccati
//                recursion (DDPSolver.hpp:386-530) is 7 such instructions plus ~100 VALU/DPP/LDS instructions, instead of
//                ~330 VALU instructions per timestep on the master wave of the 2-wave kernel.
//   linearise    the derivatives of 16 timesteps x 4 instances are evaluated by the 64 lanes at once (they do not
//                depend on the recursion) into a wave-private LDS chunk, which the next 16 recursion steps read.
//   forward      wave 0 = master, wave 1 = helper of PairSolver, lanes 0..15 (the other lanes mirror them: same
//                instance, same addresses, same values); waves 2 and 3 only take part in the barriers.
//
// Rounding: the matrix core accumulates fma(a_k, b_k, acc) for k ascending starting from C (bit-identical to a scalar
// FMA chain, profiles/r01_ubench.txt), in the same association as the reference's expressions ((Fx^T Vxx) Fx, ...);
// sums the lane kernels start from zero and add to L afterwards start from L here, so values agree with the lane
// kernels to rounding (tests/test_gpu_parity.py), not bit for bit.
#pragma once

#include <type_traits>

#include <nmpc_amd/hip/ddp_kernels_2w.hpp>

namespace nmpc_amd
{
namespace hip
{
constexpr int kQuadInstances = 16; //!< instances per workgroup of the quad kernel
constexpr int kQuadWaves = 4;

/** \tparam kFanOut step-size-parallel line search: the four lane groups of 16 — mirrors of one another otherwise — try four
    step sizes of alpha_list per forward pass, from the first pass on (at most 3 passes instead of 11).  Group 0 stores its
    rollout into the candidate half, groups 1 - 3 into the handle's fan-out scratch, and the rollout of the accepted step
    size is copied from there by the whole workgroup (PairSolver::adoptFanOut instead of another pass).
    Box-constrained solves backtrack often (cart-pole with a +-15 N box: ~3 trials per iteration) and always use it.
    Unconstrained solves have both instantiations; Configuration::line_search_fan_out = 0 picks this one (nominal bench:
    +4 %, the few instances whose first step size fails no longer cost their workgroup a pass each; M1: 2.4 x), 2 the
    sequential search (A/B, bit-identical results: tests/test_gpu_parity.py). */
template<class Problem, bool kConstrained, bool kFanOut = kConstrained>
struct QuadSolver : PairSolver<Problem, kConstrained, true, (kConstrained || kFanOut) ? 4 : 1, true>
{
  using Pair = PairSolver<Problem, kConstrained, true, (kConstrained || kFanOut) ? 4 : 1, true>;
  using Base = typename Pair::Base;
  using Base::b;
  using Base::buf;
  using Base::cfg;
  using Base::current_t;
  using Base::dV0;
  using Base::dV1;
  using Base::k_rel_norm;
  using Base::lambda;
  using Base::lane;
  using Base::problem;
  using Base::sel;
  using Base::T;
  static constexpr int N = Base::N;
  static constexpr int M = Base::M;
  static constexpr int MM = Base::MM;
  static constexpr size_t LW = Base::LW;
  using typename Base::InputDimVector;
  using typename Base::InputInputDimMatrix;
  using typename Base::QPOut;
  using typename Base::StateDimVector;
  using typename Base::StateInputDimMatrix;
  using typename Base::StateStateDimMatrix;

  static constexpr bool kShape = (N >= 1 && N <= 4 && M == 1 && !Problem::kDynamicInput);

  // ---- derivative record of one (instance, timestep), doubles; 4 x 4 blocks row-major, zero-padded beyond n ----
  static constexpr int oFx = 0;
  static constexpr int oLxx = 16;
  static constexpr int oFu = 32;
  static constexpr int oLxu = 36; // Lxu then Lx: columns 0 and 1 of one 4 x 4 operand
  static constexpr int oLx = 40;
  static constexpr int oLuu = 44; // Luu then Lu: entries (0, 0) and (0, 1) of one 4 x 4 operand
  static constexpr int oLu = 45;
  static constexpr int oU = 46; // unconstrained: u_i.  Box-constrained solves keep what the BoxQP needs instead:
  static constexpr int oLoRel = 46; // lower limit - u_i, upper limit - u_i (DDPSolver.hpp:470-472), evaluated with the
  static constexpr int oUpRel = 48; // derivatives: the limits' loads are off the recursion's dependency chain
  static constexpr int oZero = 47; // 0.0: what the lanes outside a masked operand read
  static constexpr int kRecQ = 49; // odd: the 64 lanes of the linearisation write conflict-free
  static constexpr int kChunkSteps = 16;
  // gains of the chunk's 64 (instance, timestep) pairs, staged in LDS and written to HBM at the chunk boundary
  static constexpr int gK = 0; // k
  static constexpr int gKfb = 1; // K[0 .. 3]
  static constexpr int gLive = 5; // 1.0: this timestep's gains are to be saved (DDPSolver.hpp:529-530 was reached)
  static constexpr int gDummy = 6;
  static constexpr int kGainRec = 7;
  // 8 doubles between the record blocks of a wave's four instances.  16 records of 49 doubles are 32 banks (mod 64) apart, so the
  // operand reads of instances 0 / 2 and 1 / 3 of a wavefront hit the same banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.51 in
  // rounds 2 - 4); with the pad the four blocks start 0, 48, 32, 16 banks apart [measured, profiles/r05_c2_chain_ab.txt: c2 0.451 ->
  // 0.447 ms, c3 0.681 -> 0.661, cart-pole +- 15 N 0.930 -> 0.921, M1 3.194 -> 3.181; results bit-identical].
  static constexpr int kInstPad = 8;
  static constexpr int kChunkDoubles = 64 * (kRecQ + kGainRec) + 4 * kInstPad;
  // ---- mailboxes master <-> backward waves, per instance of the workgroup ----
  static constexpr int kMailIn = 4; // need, lambda, sel, t0
  static constexpr int kMailOut = 4; // ok, dV0, dV1, k_rel_norm
  static constexpr int kMailDoubles = (kMailIn + kMailOut) * kQuadInstances;
  static constexpr int kLdsDoubles = Pair::kLdsDoubles + kMailDoubles + kQuadWaves * kChunkDoubles;
  static constexpr size_t kLdsBytes = static_cast<size_t>(kLdsDoubles) * sizeof(double);

  //! the problem object of this lane's instance in the LINEARISATION mapping (= `problem` unless the batch has
  //! per-instance objects, nmpc_hip_ddp_set_model_params_batch)
  const Problem & lin_problem;
  const int wave; //!< 0 .. 3
  const int wl; //!< lane within the wavefront
  // natural layout of this lane: entry (row, col) of block blk
  const int row, blk, col;
  double * mail;
  double * chunk; //!< this wave's derivative chunk

  NMPC_D QuadSolver(const Problem & p,
                    const Problem & p_lin,
                    const nmpc_hip_ddp_config & c,
                    const DeviceBuffers & bf,
                    int global_lane,
                    double * lds_base)
  : Pair(p, c, bf, global_lane, lds_base), lin_problem(p_lin), wave(threadIdx.x / 64), wl(threadIdx.x % 64), row(wl / 16), blk((wl / 4) % 4),
    col(wl % 4), mail(lds_base + Pair::kLdsDoubles),
    chunk(lds_base + Pair::kLdsDoubles + kMailDoubles + (threadIdx.x / 64) * kChunkDoubles)
  {
  }

  NMPC_D double & mailIn(int inst, int f) const
  {
    return mail[f * kQuadInstances + inst];
  }
  NMPC_D double & mailOut(int inst, int f) const
  {
    return mail[(kMailIn + f) * kQuadInstances + inst];
  }

  NMPC_D static double mma(double a, double b_, double c)
  {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b_, c, 0, 0, 0);
  }
  /** Entry `kSrc` of this lane's quad (= column kSrc of the same row and block) in all four lanes of the quad. */
  template<int kSrc>
  NMPC_D static double quadBroadcast(double v)
  {
    constexpr int ctrl = kSrc | (kSrc << 2) | (kSrc << 4) | (kSrc << 6); // DPP quad_perm:[s,s,s,s]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  NMPC_D static double pick(bool p, double v)
  {
    return p ? v : 0.0;
  }

  /** Derivatives of timestep i of instance `inst_l` (this lane's instance in the LINEARISATION mapping) -> record. */
  struct PointQ
  {
    double x[N], u;
    double lo, hi; //!< input limits of the timestep (box-constrained solves)
  };
  NMPC_D static void loadPointQ(int i, const double * px, const double * pu, PointQ & p)
  {
#pragma unroll
    for(int j = 0; j < N; j++)
    {
      p.x[j] = px[(static_cast<size_t>(i) * N + j) * LW];
    }
    p.u = pu[static_cast<size_t>(i) * LW];
  }
  /** One record entry.  kFull = false: entries the compiler knows to be constants (the literal zeros and ones of the
      model's Jacobians and Hessians, the padding beyond n) are not written again — the record already holds them from
      the pass's first (full) chunk.  An LDS store costs a lone wave 14 cycles, 25 when the four waves of the workgroup
      write (scripts/ubench_issue_cost.hip): the 49 stores of a record were most of the linearisation's time. */
  template<bool kFull>
  NMPC_D static void put(double * at, double v)
  {
    if(kFull || !__builtin_constant_p(v))
    {
      *at = v;
    }
  }
  /** \return 1 / (|u_i| + 1), the weight of |k_i| in the running max of DDPSolver.hpp:217-221 */
  template<bool kFull>
  NMPC_D double lineariseStep(int i, double t0_l, const PointQ & p, double * rec) const
  {
    const double t = t0_l + i * lin_problem.dt();
    StateDimVector x;
    InputDimVector u;
    u.resize(M);
#pragma unroll
    for(int j = 0; j < N; j++)
    {
      x[j] = p.x[j];
    }
    u[0] = p.u;
    StateStateDimMatrix Fx, Lxx;
    StateInputDimMatrix Fu, Lxu;
    StateDimVector Lx;
    InputDimVector Lu;
    InputInputDimMatrix Luu;
    Fu.resize(N, M);
    Lxu.resize(N, M);
    Lu.resize(M);
    Luu.resize(M, M);
    lin_problem.calcStateEqDeriv(t, x, u, Fx, Fu);
    lin_problem.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        put<kFull>(rec + oFx + 4 * r + c, (r < N && c < N) ? Fx(r < N ? r : 0, c < N ? c : 0) : 0.0);
        put<kFull>(rec + oLxx + 4 * r + c, (r < N && c < N) ? Lxx(r < N ? r : 0, c < N ? c : 0) : 0.0);
      }
      put<kFull>(rec + oFu + r, (r < N) ? Fu(r < N ? r : 0, 0) : 0.0);
      put<kFull>(rec + oLxu + r, (r < N) ? Lxu(r < N ? r : 0, 0) : 0.0);
      put<kFull>(rec + oLx + r, (r < N) ? Lx[r < N ? r : 0] : 0.0);
    }
    put<kFull>(rec + oLuu, Luu(0, 0));
    put<kFull>(rec + oLu, Lu[0]);
    if constexpr(kConstrained)
    {
      rec[oLoRel] = p.lo - u[0];
      rec[oUpRel] = p.hi - u[0];
    }
    else
    {
      rec[oU] = u[0];
    }
    put<kFull>(rec + oZero, 0.0);
    return recipFast(fabs(u[0]) + 1.0);
  }

  /** One backward pass (DDPSolver::backwardPass, DDPSolver.hpp:342-534) of the four instances of this wave.  Entered by
      all four waves after the master's post(kCmdBackward); inputs and results go through the LDS mailboxes. */
  NMPC_D void backwardQuad() const
  {
    // ---- this lane in the RECURSION mapping: entry (row, col) of instance 4 * wave + blk ----
    const int inst = wave * 4 + blk;
    const bool need = mailIn(inst, 0) != 0.0;
    const double lam = mailIn(inst, 1);
    const int lane_q = static_cast<int>(lane - lane % kQuadInstances) + inst; // lane of that instance in its tile
    // ---- this lane in the LINEARISATION mapping: timestep offset wl % 16 of instance 4 * wave + wl / 16 ----
    const int inst_l = wave * 4 + wl / 16;
    const int ts_l = wl % 16;
    const int sel_l = static_cast<int>(mailIn(inst_l, 2));
    const double t0_l = mailIn(inst_l, 3);
    const int lane_l = static_cast<int>(lane - lane % kQuadInstances) + inst_l;
    const double * px = Base::Xt + static_cast<size_t>(sel_l) * (Base::rowsX() * LW) + lane_l;
    const double * pu = Base::Ut + static_cast<size_t>(sel_l) * (Base::rowsU() * LW) + lane_l;
    double * rec_l = chunk + static_cast<size_t>(wl) * kRecQ + (wl / 16) * kInstPad;
    double * gains = chunk + 64 * kRecQ + 4 * kInstPad;
    const double * gain_l = gains + static_cast<size_t>(wl) * kGainRec; // linearisation mapping: read at the chunk boundary
    // recursion mapping: rows 0 and 1 of the block write [K | k, live, -, -] of the timestep, one double per lane
    // (rows 2 and 3 write the record's dummy slot: no branch in the recursion loop)
    double * gain_q = gains + static_cast<size_t>(blk * 16) * kGainRec
                      + (row == 0 ? gKfb + col : ((row == 1 && col == 0) ? gK : ((row == 1 && col == 1) ? gLive : gDummy)));
    // the same without the flag (unguarded chunks set the 16 flags of an instance with one store when they are done: the
    // 16 lanes of a block are then the 16 timesteps)
    double * gain_u = gains + static_cast<size_t>(blk * 16) * kGainRec
                      + (row == 0 ? gKfb + col : ((row == 1 && col == 0) ? gK : gDummy));
    double * live_q = gains + static_cast<size_t>(blk * 16 + 4 * row + col) * kGainRec + gLive;
    const double * rec_q = chunk + static_cast<size_t>(blk * 16) * kRecQ + blk * kInstPad;

    // lane predicates of the natural layout
    const bool c0 = col == 0, c1 = col == 1, r0 = row == 0;
    // this lane's offsets into a record: masked operands read the record's zero slot instead of being selected
    const int aE = 4 * row + col; // entry (row, col) of a 4 x 4 block
    const int aT = 4 * col + row; // the transposed entry
    const int aFuM = c0 ? oFu + row : oZero; // [Fu | 0 | 0 | 0]
    const int aFuB = oFu + row; // Fu in every column
    const int aLM = c0 ? oLxu + row : (c1 ? oLx + row : oZero); // [Lxu | Lx | 0 | 0]
    const int aCM = c0 ? oLuu : (c1 ? oLu : oZero); // [Luu, Lu, 0, 0] in every row
    const int aLxuRow = oLxu + col; // Lxu^T in every row
    struct Operands
    {
      double Fx, Lxx, LxxT, FuM, FuB, LM, CM, LxuRow, lo, up;
    };
    auto loadOperands = [&](int ts, Operands & o)
    {
      const double * R = rec_q + static_cast<size_t>(ts) * kRecQ;
      o.Fx = R[oFx + aE];
      o.Lxx = R[oLxx + aE];
      o.LxxT = R[oLxx + aT];
      o.FuM = R[aFuM];
      o.FuB = R[aFuB];
      o.LM = R[aLM];
      o.CM = R[aCM];
      o.LxuRow = R[aLxuRow];
      if constexpr(kConstrained)
      {
        o.lo = R[oLoRel];
        o.up = R[oUpRel];
      }
    };

    // ---- terminal value function    DDPSolver.hpp:349-352
    if(ts_l == 0)
    {
      StateDimVector xT, vx;
      StateStateDimMatrix vxx;
#pragma unroll
      for(int j = 0; j < N; j++)
      {
        xT[j] = px[(static_cast<size_t>(T) * N + j) * LW];
      }
      lin_problem.calcTerminalCostDeriv(t0_l + T * lin_problem.dt(), xT, vx, vxx);
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
#pragma unroll
        for(int c = 0; c < 4; c++)
        {
          rec_l[4 * r + c] = (r < N && c < N) ? vxx(r < N ? r : 0, c < N ? c : 0) : 0.0;
        }
        rec_l[16 + r] = (r < N) ? vx[r < N ? r : 0] : 0.0;
      }
    }
    double Vxx = rec_q[aE];
    double VxM = pick(c1, rec_q[16 + row]); // Vx in column 1, zero elsewhere

    double dV0_l = 0, dV1_l = 0;
    bool ok = true;
    double k_next = 0;
    bool have_next = false;
    const size_t tile_T = static_cast<size_t>(T);
    const int b_q = b - static_cast<int>(lane) + lane_q; // global index of this lane's instance

    /** One timestep of the recursion on the operands `o`; requests the operands of record `ts_next` into `o_next`
        first (they do not depend on the recursion: their LDS latency hides behind this timestep). */
    auto step = [&](auto reg_tag, auto guarded_tag, int i, int ts, const Operands & o, int ts_next, Operands & o_next)
    {
      constexpr int kReg = decltype(reg_tag)::value; // Configuration::reg_type, a compile-time constant in here
      // guarded: the reference's semantics for a pass that fails at this timestep or has failed before (nothing is
      // accumulated or saved from then on); unguarded: the caller looks at `ok` when the chunk is done and repeats it
      constexpr bool kGuarded = decltype(guarded_tag)::value;
      loadOperands(ts_next, o_next);
      // ---- Q terms    DDPSolver.hpp:386-408   (mma(X, Y, C) = X^T Y + C)
      const double P = mma(Vxx, o.Fx, 0.0); // Vxx^T Fx = (Fx^T Vxx)^T
      const double Rm = mma(Vxx, o.FuM, VxM); // [(Fu^T Vxx)^T | Vx | 0 | 0]
      const double Qxx = mma(P, o.Fx, o.Lxx); // Lxx + (Fx^T Vxx) Fx
      const double QxxT = mma(o.Fx, P, o.LxxT); // the same entries, transposed
      const double S = mma(o.Fx, Rm, o.LM); // column 0 = Qux^T, column 1 = Qx
      const double Wq = mma(o.FuB, Rm, o.CM); // every row: [Quu, Qu, 0, 0]
      const double QA = mma(quadBroadcast<0>(Rm), o.Fx, o.LxuRow); // Qux[col] in every row
      const double Quu = quadBroadcast<0>(Wq);
      const double Qu = quadBroadcast<1>(Wq);
      const double Qr = quadBroadcast<0>(S); // Qux[row]
      const double Qxr = quadBroadcast<1>(S); // Qx[row]

      // ---- regularisation    :421-441
      double Quu_F = Quu, QAr = QA, Qrr = Qr;
      if constexpr(kReg == 2)
      {
        const double VxxReg = (row == col) ? Vxx + lam : Vxx;
        const double R2 = mma(VxxReg, o.FuM, 0.0);
        Quu_F = quadBroadcast<0>(mma(o.FuB, R2, o.CM));
        QAr = mma(quadBroadcast<0>(R2), o.Fx, o.LxuRow);
        Qrr = quadBroadcast<0>(mma(o.Fx, R2, o.LM));
      }
      else if constexpr(kReg == 1)
      {
        Quu_F = Quu + lam;
      }

      // ---- gains    :448-517   (m = 1: the factorisation is the pivot itself; inv = 0 leaves k = K = 0)
      double k, inv;
      bool step_ok;
      [[maybe_unused]] double qp_code = 0;
      if constexpr(kConstrained)
      {
        const double initial_k = (i != T - 1 && have_next) ? k_next : 0.0;
        QPOut qp;
#ifdef NMPC_AMD_PROFILE_QP
        asm volatile("s_nop 0" ::"v"(Quu_F), "v"(Qu));
        const unsigned long long tq = __builtin_readcyclecounter();
#endif
        Base::boxQP1Fast(Quu_F, Qu, o.lo, o.up, initial_k, qp);
#ifdef NMPC_AMD_PROFILE_QP
        asm volatile("s_nop 0" ::"v"(qp.x[0]), "v"(qp.retval));
        Pair::prof_wait += __builtin_readcyclecounter() - tq;
#endif
        // retval_ and the free set of this timestep (DDPSolver.h:152-157 keeps them for the caller) travel with the gains:
        // 4 (retval + 8) + 2 [free] (+ 1: the gains are to be saved), 0 = the pass had failed before this timestep
        qp_code = (need && ok) ? static_cast<double>(4 * (qp.retval + 8) + (qp.n_free > 0 ? 2 : 0)) : 0.0;
        step_ok = qp.retval >= 0;
        k = step_ok ? qp.x[0] : 0.0;
        inv = (step_ok && qp.n_free > 0) ? qp.inv_d[0] : 0.0;
      }
      else
      {
        step_ok = !(Quu_F <= 0);
        inv = (!kGuarded || step_ok) ? recipFast(Quu_F) : 0.0;
        k = -1 * (Qu * inv);
      }
      const double Kc = -1 * (QAr * inv); // K[col]
      const double Kr = -1 * (Qrr * inv); // K[row]
      const bool live = need && ok && step_ok;
      ok = ok && step_ok;

      // ---- cost-to-go update    :522-527
      if(!kGuarded || live)
      {
        dV0_l += k * Qu;
        dV1_l += 0.5 * (k * (Quu * k));
      }
      const double KQr = Kr * Quu, KQc = Kc * Quu; // (K^T Quu)[row], [col]
      // Qxx + K^T Quu K + K^T Qux + Qux^T K, entry (row, col) and entry (col, row)
      const double Vn = fma(Qr, Kc, fma(Kr, QA, fma(KQr, Kc, Qxx)));
      const double VnT = fma(QA, Kr, fma(Kc, Qr, fma(KQc, Kr, QxxT)));
      Vxx = 0.5 * (Vn + VnT);
      // Qx + K^T Quu k + K^T Qu + Qux^T k
      VxM = pick(c1, fma(Qr, k, fma(Kr, Qu, fma(KQr, k, Qxr))));

      // ---- save gains    :529-530 (staged, see flushGains)
      if constexpr(kGuarded)
      {
        gain_q[static_cast<size_t>(ts) * kGainRec] = r0 ? Kc : (c0 ? k : (kConstrained ? qp_code + (live ? 1.0 : 0.0) : (live ? 1.0 : 0.0)));
      }
      else
      {
        gain_u[static_cast<size_t>(ts) * kGainRec] = r0 ? Kc : k;
      }
      // (what a lane computes after it stopped being live is never read: a failed pass is retried or ends the solve)
      k_next = k;
      have_next = true;
    };

    /** Gains of the chunk starting at timestep i_first: LDS -> k_list_ / K_list_ in HBM, by the lane of each (instance,
        timestep) pair in the linearisation mapping, which also takes the running max of |k_i| / (|u_i| + 1)
        (DDPSolver.hpp:217-221; uinv = this lane's 1 / (|u_i| + 1) from the chunk's linearisation): two instructions per
        chunk here instead of an LDS read and two instructions per timestep in the recursion. */
    const int b_l = b - static_cast<int>(lane) + lane_l; // global index of this lane's instance in the linearisation mapping
    auto loadLimits = [&](int i, PointQ & p)
    {
      if constexpr(kConstrained)
      {
        p.lo = inputLimitLo(buf, b_l, i, 0);
        p.hi = inputLimitHi(buf, b_l, i, 0);
      }
    };
    double krn_l = 0;
    auto flushGains = [&](int i_first, double uinv)
    {
      const int i = i_first + ts_l;
      const double flag = (i < T) ? gain_l[gLive] : 0.0;
      bool save = flag != 0.0;
      if constexpr(kConstrained)
      {
        const int code = static_cast<int>(flag);
        if(save)
        {
          Base::tileBase(buf.qp_ret, tile_T)[static_cast<size_t>(i) * LW + lane_l] = (code >> 2) - 8;
          Base::tileBase(buf.qp_free, tile_T)[static_cast<size_t>(i) * LW + lane_l] = static_cast<unsigned>((code >> 1) & 1);
        }
        save = (code & 1) != 0;
      }
      if(save)
      {
        const double kv = gain_l[gK];
        Base::kt[static_cast<size_t>(i) * LW + lane_l] = kv;
#pragma unroll
        for(int c = 0; c < N; c++)
        {
          Base::Kt[(static_cast<size_t>(i) * N + c) * LW + lane_l] = gain_l[gKfb + c];
        }
        krn_l = fmax(krn_l, fabs(kv) * uinv);
      }
    };
    auto runChunks = [&](auto reg_tag)
    {
      const int n_chunks = (T + kChunkSteps - 1) / kChunkSteps;
      PointQ pt;
      {
        const int i = ((n_chunks - 1) * kChunkSteps + ts_l < T) ? (n_chunks - 1) * kChunkSteps + ts_l : T - 1;
        loadPointQ(i, px, pu, pt);
        loadLimits(i, pt);
      }
      double uinv_prev = 0; // 1 / (|u| + 1) of this lane's timestep in the chunk whose gains are staged
      for(int ch = n_chunks - 1; ch >= 0; ch--)
      {
        const int i0 = ch * kChunkSteps;
        double uinv_now;
        {
#ifdef NMPC_AMD_PROFILE_2W
          const unsigned long long tl = __builtin_readcyclecounter();
#endif
          const int i = (i0 + ts_l < T) ? i0 + ts_l : T - 1;
          // (the terminal blocks above were parked in the records of the lanes with ts_l = 0: the first chunk rewrites all)
          uinv_now = (ch == n_chunks - 1) ? lineariseStep<true>(i, t0_l, pt, rec_l) : lineariseStep<false>(i, t0_l, pt, rec_l);
#ifdef NMPC_AMD_PROFILE_2W
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef NMPC_AMD_PROFILE_QP
          Pair::prof_wait += __builtin_readcyclecounter() - tl; // reported as "barrier wait": the linearisation share
#endif
          Pair::prof_count++;
#endif
        }
        // HBM traffic of this wave happens here, one chunk behind / ahead of the recursion: the previous chunk's gains
        // go out, then the next chunk's (x, u) are requested — both have the 16 recursion steps below to complete
        if(ch + 1 < n_chunks)
        {
          flushGains(i0 + kChunkSteps, uinv_prev);
        }
        uinv_prev = uinv_now;
        if(!Pair::kNominalTail || ch > 0) // (kNominalTail: the last chunk keeps its own (x, u) for the nominal records below)
        {
          const int in = i0 - kChunkSteps + ts_l; // (unconditional, clamped: the last request is never used)
          loadPointQ(in > 0 ? in : 0, px, pu, pt);
          loadLimits(in > 0 ? in : 0, pt);
        }
        const int hi = (i0 + kChunkSteps - 1 < T) ? i0 + kChunkSteps - 1 : T - 1;
#ifdef NMPC_AMD_QUAD_NO_STRAIGHT
        if constexpr(false)
#else
        if constexpr(!kConstrained)
#endif
        {
          const int n_steps = hi - i0 + 1;
          if((n_steps == kChunkSteps || n_steps == 4) && __all(ok))
          {
            // a full chunk as straight-line code: one scheduling region for the 16 timesteps (the compiler fills the
            // recursion's dependency stalls across timesteps), immediate LDS offsets, and NO guards: a lone wave issues
            // one instruction every ~4.3 cycles whatever its kind (scripts/ubench_issue_cost.hip), so the selects that
            // freeze a failed pass (`inv`, the two dV sums, the live flag: ~16 instructions of ~80 per timestep) cost
            // what arithmetic costs.  A pivot that fails (rare) is noticed when the chunk is done, and the chunk is
            // then repeated from its saved state by the guarded loop below.  (The BoxQP variant would not fit the
            // instruction cache.)
            // The ragged chunk of a horizon that is not a multiple of 16 (the one with the terminal step) runs the same
            // straight-line code from its own first step if it has 4 timesteps (T = 100, the reference's default horizon:
            // +0.8 %).  Measured and not kept: 12 timesteps (bipedal, T = 300) as straight-line code are 2.4 % SLOWER than
            // the guarded loop — code that runs once per pass is fetched cold, the loop's is resident.
            const double Vxx_s = Vxx, VxM_s = VxM, dV0_s = dV0_l, dV1_s = dV1_l;
            auto straight = [&](auto r_tag)
            {
              constexpr int R = decltype(r_tag)::value;
              Operands o2[2];
              loadOperands(R - 1, o2[(R - 1) & 1]);
#pragma unroll
              for(int ts = R - 1; ts >= 0; ts--)
              {
                step(reg_tag, std::false_type(), i0 + ts, ts, o2[ts & 1], ts > 0 ? ts - 1 : 0, o2[(ts & 1) ^ 1]);
              }
            };
            if(n_steps == kChunkSteps)
            {
              straight(std::integral_constant<int, kChunkSteps>());
            }
            else
            {
              straight(std::integral_constant<int, 4>());
            }
            if(__all(ok))
            {
              *live_q = need ? 1.0 : 0.0; // the 16 lanes of a block: the flags of the chunk's 16 timesteps
              continue;
            }
            Vxx = Vxx_s;
            VxM = VxM_s;
            dV0_l = dV0_s;
            dV1_l = dV1_s;
            ok = true;
          }
        }
        // two operand sets, loop unrolled by two: no register copies between timesteps
        Operands oa, ob;
        loadOperands(hi - i0, oa);
        int i = hi;
        for(; i - 1 >= i0; i -= 2)
        {
          step(reg_tag, std::true_type(), i, i - i0, oa, i - 1 - i0, ob);
          step(reg_tag, std::true_type(), i - 1, i - 1 - i0, ob, (i - 2 >= i0) ? i - 2 - i0 : 0, oa);
        }
        if(i >= i0)
        {
          step(reg_tag, std::true_type(), i, i - i0, oa, 0, ob);
        }
      }
      flushGains(0, uinv_prev);
      // The forward pass that follows starts from timestep 0 and takes its nominal (x, u, k, K) from LDS records that a prefetching
      // wave keeps ahead of it; its first groups used to be a round trip to HBM behind the pass barrier, for values that are HERE:
      // (x, u) of timesteps 0 .. 15 in the linearisation lanes' registers, the gains in the staging area.  Lane (instance, timestep)
      // writes the record of the first three groups (PairSolver::nomRec); the master says so in mailNomResident.
      if constexpr(Pair::kNominalTail)
      {
        asm volatile("" ::: "memory"); // (the staged gains were written by other lanes of this wave: LDS is in order)
        if(ts_l < 3 * Pair::kFwdGroup)
        {
          double * rec = Pair::nomRec(ts_l / Pair::kFwdGroup, ts_l % Pair::kFwdGroup, static_cast<unsigned>(inst_l));
#pragma unroll
          for(int j = 0; j < N; j++)
          {
            rec[j] = pt.x[j];
          }
          rec[N] = pt.u;
          rec[N + 1] = gain_l[gK];
#pragma unroll
          for(int c = 0; c < N; c++)
          {
            rec[N + 2 + c] = gain_l[gKfb + c];
          }
          if constexpr(Pair::kNomRec > N + 2 + N)
          {
            rec[Pair::kNomRec - 1] = 0;
          }
        }
      }
    };
    if(cfg.reg_type == 2)
    {
      runChunks(std::integral_constant<int, 2>());
    }
    else if(cfg.reg_type == 1)
    {
      runChunks(std::integral_constant<int, 1>());
    }
    else
    {
      runChunks(std::integral_constant<int, 0>());
    }
    // max over the 16 timestep lanes of an instance (the lanes of one DPP row)
#pragma unroll
    for(int m = 8; m >= 1; m >>= 1)
    {
      krn_l = fmax(krn_l, __shfl_xor(krn_l, m, 16));
    }
    if(ts_l == 0)
    {
      mailOut(inst_l, 3) = krn_l;
    }
    if(r0 && c0)
    {
      mailOut(inst, 0) = ok ? 1.0 : 0.0;
      mailOut(inst, 1) = dV0_l;
      mailOut(inst, 2) = dV1_l;
    }
    fullBarrier(); // closes the pass: results in the mailboxes, gains in HBM
  }

  /** Master side of one backward pass: publish the inputs, run this wave's share, collect the results. */
  NMPC_D bool backwardMasterQuad(bool need)
  {
    const int inst = wl % kQuadInstances;
    if(wl < kQuadInstances)
    {
      mailIn(inst, 0) = need ? 1.0 : 0.0;
      mailIn(inst, 1) = lambda;
      mailIn(inst, 2) = static_cast<double>(sel);
      mailIn(inst, 3) = current_t;
    }
    Pair::post(Pair::kCmdBackward);
    Pair::profBegin();
    backwardQuad();
    Pair::profEnd(0);
    const bool ok = mailOut(inst, 0) != 0.0;
    if(need)
    {
      dV0 = mailOut(inst, 1);
      dV1 = mailOut(inst, 2);
      k_rel_norm = mailOut(inst, 3);
    }
    return ok;
  }

  template<bool kResumable = false>
  NMPC_D void solveMasterQuad(bool valid)
  {
    if constexpr(kConstrained || kFanOut)
    {
      Pair::template solveMasterFanOut<kResumable>(valid, [this](bool need) { return backwardMasterQuad(need); });
    }
    else
    {
      Pair::template solveMasterWith<kResumable>(valid, [this](bool need) { return backwardMasterQuad(need); });
    }
  }

  /** Waves 1 .. 3: follow the master's commands.  Wave 1 is the forward helper of PairSolver, waves 2 and 3 only
      attend the forward pass's barriers. */
  NMPC_D void followerLoop() const
  {
    for(;;)
    {
      fullBarrier(); // barrier P of PairSolver::post()
      const int word = static_cast<int>(Pair::mailFlags());
      const int cmd = __builtin_amdgcn_readfirstlane(word >> 1);
      const int sel_h = word & 1;
      if(cmd == Pair::kCmdExit)
      {
#ifdef NMPC_AMD_PROFILE_2W
        if(blockIdx.x == 0 && wl == 0)
        {
          buf.qp_free[static_cast<size_t>(4 + wave) * LW] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        }
#endif
        return;
      }
      if(cmd == Pair::kCmdBackward)
      {
        backwardQuad();
      }
      else if(cmd == Pair::kCmdAdoptFanOut)
      {
        Pair::adoptFanOut(sel_h);
      }
      else if(wave == 1)
      {
        if(cmd == Pair::kCmdForwardFanOut)
        {
          Pair::template forwardHelper<true>(sel_h);
        }
        else
        {
          Pair::forwardHelper(sel_h, cmd == Pair::kCmdRollout);
        }
      }
      else if(Pair::kExtraMasters > 0 && cmd == Pair::kCmdForwardFanOut && Pair::mailWidePass() != 0.0)
      {
        // waves 2 and 3 roll out the step sizes behind the master's four, cost only; wave 2 also keeps the nominal records ahead
        if(wave == 2)
        {
          Pair::template forwardCostOnlyLds<true>(sel_h, 0);
        }
        else
        {
          Pair::template forwardCostOnlyLds<false>(sel_h, 1);
        }
      }
      else if(Pair::kLdsNominalPath && (cmd == Pair::kCmdForward || cmd == Pair::kCmdForwardFanOut) && wave == 2)
      {
        Pair::forwardPrefetch(sel_h);
      }
      else
      {
        // one barrier per group of timesteps, then E and F (+ barrier S where wave 2 feeds the master from LDS)
        const int n_bar = T / Pair::kFwdGroup + 2
                          + ((Pair::kLdsNominalPath && (cmd == Pair::kCmdForward || cmd == Pair::kCmdForwardFanOut)) ? 1 : 0);
        for(int i = 0; i < n_bar; i++)
        {
          Pair::wgBarrier();
        }
      }
    }
  }
};

/** The quad solve kernel: grid = Bp / 16 workgroups of 256 threads.  Compile the translation unit with
    `-mllvm --amdgpu-mfma-vgpr-form` (nmpc_amd/build.py does): by default a kernel that may use 512 registers gets its
    matrix-core results in accumulation registers and spends ~30 v_accvgpr_read/write per timestep moving them to the
    VALU / DPP instructions that consume them (10.3k -> 10.6k iterations/s on the headline workload). */
template<class Problem, bool kConstrained, bool kOwnProblem = false, bool kFanOut = kConstrained, bool kResumable = false>
__global__ __launch_bounds__(kQuadWaves * 64) void ddp_solve_quad_kernel(const Problem problem,
                                                                         const nmpc_hip_ddp_config cfg,
                                                                         const DeviceBuffers buf)
{
  using Solver = QuadSolver<Problem, kConstrained, kFanOut>;
  extern __shared__ __attribute__((aligned(16))) double lds_quad[];
  const int wl = threadIdx.x % 64;
  int first = 0;
  bool solver_fresh = false;
  if constexpr(kResumable)
  {
    // streamed solves (DeviceBuffers::stream_mode): see ddp_solve_tpi2w_kernel
    const int first_fresh = (buf.stream_mode != 0 && buf.first_active) ? *buf.first_active : 0;
    solver_fresh = buf.stream_mode == 2 && buf.first_active && static_cast<int>(blockIdx.x) * kQuadInstances >= first_fresh;
    first = (buf.stream_mode == 1) ? first_fresh : 0;
    if(static_cast<int>(blockIdx.x + 1) * kQuadInstances <= first
       || (buf.stream_mode != 0 && buf.n_active && static_cast<int>(blockIdx.x) * kQuadInstances >= *buf.n_active)) // (... and the empty slots behind)
    {
      return;
    }
  }
  // in the lane-per-instance roles (master, forward helper) every group of 16 lanes mirrors the workgroup's 16
  // instances: same inputs, same instruction stream, hence the same values and decisions, written to the same places
  const int b = blockIdx.x * kQuadInstances + wl % kQuadInstances;
  // kOwnProblem: per-instance problem objects (a separate instantiation, as in the two-wave kernel).  A lane serves two
  // instances: b in the lane-per-instance roles, b_lin when it linearises (timestep wl % 16 of instance 4 wave + wl / 16).
  const int b_lin = blockIdx.x * kQuadInstances + (threadIdx.x / 64) * 4 + wl / 16;
  const Problem mine = kOwnProblem ? instanceProblem(problem, buf, b) : problem;
  const Problem mine_lin = kOwnProblem ? instanceProblem(problem, buf, b_lin) : problem;
  Solver solver(mine, kOwnProblem ? mine_lin : mine, cfg, buf, b, lds_quad);
  solver.stream_fresh_wg = solver_fresh;
  if(threadIdx.x / 64 == 0)
  {
    if constexpr(kResumable)
    {
      solver.template solveMasterQuad<true>(b >= first && b < (buf.n_active ? *buf.n_active : buf.B));
    }
    else
    {
      solver.solveMasterQuad(b < buf.B);
    }
  }
  else
  {
    solver.current_t = buf.t0 ? solver.tileBase(buf.t0, 1)[solver.lane] : 0.0;
    solver.followerLoop();
  }
}
} // namespace hip
} // namespace nmpc_amd
