// gfx950 device code of the batched DDP solver: one persistent kernel runs the whole optimisation loop of
// nmpc_ddp::DDPSolver::solve (/root/reference/nmpc_ddp/include/nmpc_ddp/DDPSolver.hpp:26-141) for every
// instance of the batch; there is no host round trip per iteration and no inter-workgroup communication
// (instances are independent, DDPSolver.h:329-374).
//
// LANE MAPPING "TPI" (this file): one LANE per problem instance, 64 instances per wavefront, one wavefront per
// workgroup.  The per-instance recursion state (Vx, Vxx, Q blocks, gains) lives in VGPRs, so nothing has to be
// exchanged between lanes and every VALU instruction does 64 instances' worth of work.  Measured on MI355X
// (profiles/ubench_r01.txt): a wave64 fp64 FMA issues in 4 cycles whether 1 or 64 lanes are active, so a
// wavefront-per-instance mapping of an n = 4 problem would spend the same issue cycles on 1/64 of the work.
// All batched arrays are tile-major [tile][half][row][64] (see InstanceSolver addressing), so a wave's access to
// one row is one contiguous 512-byte segment and every stride is an immediate.  Dynamics / cost derivatives are
// evaluated on the fly inside the backward sweep (never materialised in HBM): per instance-iteration the kernel
// moves T*(3n + 5m + 2mn + 1) words instead of the reference's materialised T*(2D + ...).
//
// Arithmetic follows the reference statement by statement (cited inline); products are accumulated in ascending
// index order; hipcc contracts a*b+c into FMA, so results differ from an x86 build in the last bits (tolerances
// in tests/).  Discrete decisions (alpha index, retries, BoxQP sets, iteration counts, status) are taken by the
// same comparisons as the reference.
#pragma once

#include <hip/hip_runtime.h>

#include <nmpc_amd/DDPProblem.hpp>
#include <nmpc_hip_ddp.h>
#include <nmpc_amd/hip/fuzz_sched.hpp>

#define NMPC_D __device__ __forceinline__

namespace nmpc_amd
{
namespace hip
{
constexpr int kMaxInputDim = 16; //!< capacity of the constant input-limit arrays passed by value
constexpr int kLanesPerBlock = 64; //!< one wavefront per workgroup

/** Device pointers of one solver handle.  Bp = batch rounded up to a multiple of 64 = 64 * number of tiles.
    Every array is tile-major: element (half s, row r) of instance b sits at
        ((b / 64) * halves + s) * rows * 64  +  r * 64  +  b % 64
    with the per-array `rows` / `halves` listed below ("[tile]" = that leading index). */
template<class S>
struct DeviceBuffersT
{
  int B; //!< number of valid instances
  int Bp; //!< padded batch
  int T; //!< horizon_steps
  int trace_rows; //!< allocated trace rows (max_iter+1 if trace_level >= 1, else 1)
  const S * t0; //!< [tile][1][64]
  const S * x0; //!< [tile][N][64]
  S * X; //!< [tile][2][(T+1)*N][64]   current / candidate state sequences (x_list), row = i*N + j
  S * U; //!< [tile][2][T*MM][64]      current / candidate input sequences (u_list); half 0 = u_init on entry
  S * cost; //!< [tile][2][T+1][64]    current / candidate cost sequences (cost_list)
  S * kff; //!< [tile][T*MM][64]         k_list_
  S * Kfb; //!< [tile][T*N*MM][64]       K_list_, row = i*N*MM + (a + c*MM) of the m x N gain
  S * trace; //!< [tile][trace_rows*NMPC_HIP_NTRACE][64]
  S * trace_last; //!< [tile][NMPC_HIP_NTRACE][64]
  S * dV; //!< [tile][2][64]
  int * status; //!< [tile][1][64]
  int * iters; //!< [tile][1][64]
  int * sel; //!< [tile][1][64]  which half of X/U/cost holds control_data_ after the solve
  int * qp_ret; //!< [tile][T][64]  (constrained solves only)
  unsigned * qp_free; //!< [tile][T][64]
  int * input_dim; //!< [tile][T][64]
  S * wpi_ws; //!< per-instance workspace [B][ModelOps::wpi_workspace_doubles(T)] elements of S: wave-per-instance kernel
             //!< (ddp_kernels_wpi.hpp: derivatives, gains, candidates), fp32 tile kernel (ddp_kernels_tile32.hpp: gains);
             //!< quad-kernel shapes (n <= 4, one input): the line search's fan-out scratch [tile][3][rows of X, U, cost][64]
             //!< (PairSolver::FanDest in ddp_kernels_2w.hpp); nullptr: none (the kernels that can do without check)
  //! [Bp][4] shader-clock ticks of the last solve as seen by the wave that drives instance b: backward passes (with the
  //! linearisation fused into them), forward passes (rollouts of the line search), whole solve, reserved; or nullptr.
  //! The split of computationDuration() (DDPSolver.h:219-247) is taken from these shares of the kernel's HIP-event time.
  unsigned long long * phase_ticks;
  const unsigned char * params_batch; //!< per-instance problem objects [Bp][sizeof(Problem)], or nullptr: one for all
  const double * lim_batch; //!< per-instance input limits [Bp][2][kMaxInputDim] (lower, upper), or nullptr: lim_lo / lim_hi
  //! time-varying input limits, input_limits_func_(current_t + i dt) of DDPSolver.hpp:470-472 sampled per timestep:
  //! [1 or Bp][T][2][lim_mm] (lower, upper), or nullptr.  Takes precedence over lim_batch / lim_lo / lim_hi.
  const double * lim_steps;
  int lim_steps_per_instance; //!< 1: one table per instance (instances start at different current_t), 0: one for all
  int lim_mm; //!< row length of lim_steps (the handle's MM)
  int lim_rows; //!< rows per table (>= T): row j = limits at current_t + j dt of the FIRST solve
  int lim_offset; //!< row of timestep 0 of this solve (the tick number in the device-resident shift loop, else 0)
  double lim_lo[kMaxInputDim]; //!< input lower limits (constant in time)
  double lim_hi[kMaxInputDim]; //!< input upper limits
  // ---- resumable solves (the ragged-convergence schedule of capi.hip): one launch executes iterations
  // iter_begin .. iter_end of the instances in positions [0, *n_active) and parks the per-instance solver state in `resume`; the host
  // queues a compaction between launches (still-running instances swapped into a dense prefix) so that a workgroup is not held
  // by one unconverged instance of sixteen.  iter_end = 0: an ordinary whole solve (every field below unused).
  const int * n_active = nullptr; //!< device word: instances in the dense prefix of this launch (nullptr: B)
  int iter_begin = 0; //!< first iteration of this launch (1: fresh start with the initial rollout; > 1: resume)
  int iter_end = 0; //!< last iteration of this launch (<= max_iter); 0: not a resumable launch
  S * resume = nullptr; //!< [tile][kResumeRows][64]: lambda, dlambda, J_cur, running (1 / 0), iterations done, between launches
  // ---- streamed solves (nmpc_hip_ddp_solve_stream, stream_schedule.hpp; round 6): the handle's B positions are SLOTS that a
  // queue of N >> B instances passes through.  Launches number their iterations per instance (row 4 of `resume`), not per launch:
  //   stream_mode 1  the slots [*first_active, *n_active) have just been filled: initial rollout (DDPSolver.hpp:83-95), state parked
  //                  (workgroups below *first_active — a multiple of 64 — exit at once: they hold instances in mid-solve);
  //   stream_mode 2  every slot of [0, *n_active) whose instance still iterates runs at most iter_end further iterations, or up
  //                  to its max_iter-th.
  int stream_mode = 0;
  const int * first_active = nullptr; //!< device word (stream_mode 1), or nullptr: 0
};
constexpr int kResumeRows = 5;
/** The reference computes in double (DDPProblem.h:20-35): every kernel but the fp32 tile kernel uses this one. */
using DeviceBuffers = DeviceBuffersT<double>;

namespace detail
{
/** Unroll factor for the small dense loops.  Full unrolling makes every index static: the per-instance blocks live in
    VGPRs (n <= 4) or in statically addressed scratch, and the structural zeros / ones of the model's Jacobians fold
    away (macc()).  Measured on MI355X (scripts/config_throughput.py, 1024 instances, 4 iterations): rolled loops leave a
    lone wavefront waiting for one dependent scratch load after the other — quadrotor (n 12, m 4) 208 ms rolled, 26 ms
    unrolled by 8, 15.8 ms fully unrolled; manipulator (n 14, m 7, dense Jacobians) 270 / 76 / 85 ms; centroidal (n 9,
    m 16) 1165 / 868 / 457 ms.  Hence: full unrolling up to n^2 (n + m) = 2400, by 8 beyond (code size, compile time).
    The wave-per-instance / MFMA mapping for these shapes is the planned replacement. */
template<int N, int MM>
struct Unroll
{
  static constexpr bool kFull = (N * N * (N + MM) <= 160);
  static constexpr int kFactor = kFull ? 64 : ((N * N * (N + MM) <= 2400) ? 16 : 8);
};
} // namespace detail

/** Workgroup barrier that also orders HBM traffic between the waves of the workgroup: everything this wave has stored is
    acknowledged (s_waitcnt vmcnt(0): gfx9 counts loads and stores in one counter) before it arrives at the barrier, and the
    waves of a workgroup share their CU's vector L1, so what has landed is what they load.  hipcc's __syncthreads() is
    `s_waitcnt lgkmcnt(0); s_barrier` (ROCm 7.2, every kernel of this library: llvm-objdump) — it orders LDS only, and a wave
    that loads what another wave of its workgroup stored just before such a barrier is not ordered after the store's
    completion by anything the ISA documents.  (Not observed failing: scripts/determinism_soak.py, 2000 repetitions per kernel
    family, is clean with either form — the CU's in-order vector-memory path hides it — but the pass boundaries were written
    as full barriers and now are.) */
NMPC_D void fullBarrier()
{
  fuzzSched(1);
#ifdef NMPC_AMD_AB_LDS_ONLY_PASS_BARRIER // (A/B builds: what __syncthreads() compiles to — scripts/determinism_soak.py against it)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  fuzzSched(2);
}

/** Input limits of instance b at timestep i (input_limits_func_(current_t + i dt), DDPSolver.hpp:470-472): the sampled
    table if time-varying limits were given, else the instance's own constant limits, else the shared ones. */
template<class Buffers>
NMPC_D double inputLimit(const Buffers & buf, int b, int i, int a, int side)
{
  if(buf.lim_steps != nullptr)
  {
    const size_t inst = buf.lim_steps_per_instance ? static_cast<size_t>(b) : 0;
    const int r = (i + buf.lim_offset < buf.lim_rows) ? i + buf.lim_offset : buf.lim_rows - 1;
    return buf.lim_steps[((inst * buf.lim_rows + r) * 2 + side) * buf.lim_mm + a];
  }
  if(buf.lim_batch != nullptr)
  {
    return buf.lim_batch[(static_cast<size_t>(b) * 2 + side) * kMaxInputDim + a];
  }
  return side == 0 ? buf.lim_lo[a] : buf.lim_hi[a];
}
template<class Buffers>
NMPC_D double inputLimitLo(const Buffers & buf, int b, int i, int a)
{
  return inputLimit(buf, b, i, a, 0);
}
template<class Buffers>
NMPC_D double inputLimitHi(const Buffers & buf, int b, int i, int a)
{
  return inputLimit(buf, b, i, a, 1);
}

/** The problem object instance b solves: the handle's shared one, or its own when per-instance objects were given
    (nmpc_hip_ddp_set_model_params_batch: a batch of solvers with different robots / weights). */
template<class Problem, class Buffers>
NMPC_D Problem instanceProblem(const Problem & shared, const Buffers & buf, int b)
{
  Problem mine = shared;
  if(buf.params_batch != nullptr)
  {
    // trivially copyable (static_assert in model_registry.hpp): word-wise copy keeps the loads coalescable per field
    constexpr size_t kWords = sizeof(Problem) / sizeof(unsigned);
    const unsigned * src = reinterpret_cast<const unsigned *>(buf.params_batch + static_cast<size_t>(b) * sizeof(Problem));
    unsigned * dst = reinterpret_cast<unsigned *>(&mine);
#pragma unroll
    for(size_t w = 0; w < kWords; w++)
    {
      dst[w] = src[w];
    }
  }
  return mine;
}

/** One DDP problem instance, executed by one lane.
    \tparam kConstrained compile-time value of Configuration::with_input_constraint (DDPSolver.h:70): the
    unconstrained kernel carries no BoxQP code or registers */
template<class Problem, bool kConstrained>
struct InstanceSolver
{
  static constexpr int N = Problem::kStateDim;
  static constexpr int M = Problem::kInputDimMax;
  static constexpr int MM = (M > 0) ? M : 1;
  static constexpr bool kDyn = Problem::kDynamicInput;
  static constexpr int kU = detail::Unroll<N, MM>::kFactor;
  static_assert(M <= kMaxInputDim, "input dimension capacity exceeded");

  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using StateStateDimMatrix = typename Problem::StateStateDimMatrix;
  using InputInputDimMatrix = typename Problem::InputInputDimMatrix;
  using StateInputDimMatrix = typename Problem::StateInputDimMatrix;

  const Problem & problem;
  const nmpc_hip_ddp_config & cfg;
  const DeviceBuffers & buf;
  const int b; //!< instance index = global lane index
  const int T;
  const unsigned tile; //!< wavefront tile = b / 64
  const unsigned lane; //!< b % 64
  // this tile's slice of every big array (wave-uniform, computed once)
  double * Xt;
  double * Ut;
  double * Ct;
  double * kt;
  double * Kt;
  static constexpr size_t LW = kLanesPerBlock; //!< row stride of every array: the 64 instances of one tile

  double current_t;
  double lambda;
  double dlambda;
  double dV0, dV1;
  double k_rel_norm;
  double J_cur; //!< control_data_.cost_list.sum()
  double J_cand; //!< candidate_control_data_.cost_list.sum()
  int sel; //!< half of X/U/cost that is control_data_; 1-sel is candidate_control_data_

  NMPC_D InstanceSolver(const Problem & p, const nmpc_hip_ddp_config & c, const DeviceBuffers & bf, int global_lane)
  : problem(p), cfg(c), buf(bf), b(global_lane), T(bf.T), tile(static_cast<unsigned>(global_lane) / kLanesPerBlock),
    lane(static_cast<unsigned>(global_lane) % kLanesPerBlock)
  {
    Xt = tileBase(bf.X, rowsX(), 2);
    Ut = tileBase(bf.U, rowsU(), 2);
    Ct = tileBase(bf.cost, static_cast<size_t>(T + 1), 2);
    kt = tileBase(bf.kff, rowsU());
    Kt = tileBase(bf.Kfb, rowsU() * N);
  }

  // ---- addressing: tile-major ("AoSoA-64") ----
  // Every batched array is laid out [tile][half][row][64]: the 64 instances of one wavefront are the fastest
  // index, so (a) a wave's access to one row is one contiguous 512-byte segment, (b) ALL data of one wave is
  // one contiguous block (few TLB entries, no 32 KB strides between the fields of one timestep), and (c) every
  // stride is a compile-time constant: an access is  UNIFORM row pointer (SGPR pair, bumped once per timestep
  // by scalar code)  +  per-lane 32-bit byte offset (one VGPR, also selects the current / candidate half)  +
  // immediate (j * 512), the "saddr + voffset + imm" form of global_load/store — no vector address arithmetic.
  //! base of this wavefront's tile in an array with `rows` rows per instance and `halves` copies
  template<class Tp>
  NMPC_D Tp * tileBase(Tp * array, size_t rows, int halves = 1) const
  {
    return array + static_cast<size_t>(__builtin_amdgcn_readfirstlane(tile)) * (rows * halves * LW);
  }
  NMPC_D static double ld(const double * row, unsigned lane_off)
  {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(row) + lane_off);
  }
  NMPC_D static void st(double * row, unsigned lane_off, double v)
  {
    *reinterpret_cast<double *>(reinterpret_cast<char *>(row) + lane_off) = v;
  }
  NMPC_D size_t rowsX() const
  {
    return static_cast<size_t>(T + 1) * N;
  }
  NMPC_D size_t rowsU() const
  {
    return static_cast<size_t>(T) * MM;
  }
  NMPC_D double * xRow(int i) const
  {
    return Xt + (static_cast<size_t>(i) * N) * LW;
  }
  NMPC_D double * uRow(int i) const
  {
    return Ut + (static_cast<size_t>(i) * MM) * LW;
  }
  NMPC_D double * costRow(int i) const
  {
    return Ct + static_cast<size_t>(i) * LW;
  }
  NMPC_D double * kRow(int i) const
  {
    return kt + (static_cast<size_t>(i) * MM) * LW;
  }
  NMPC_D double * KRow(int i) const
  {
    return Kt + (static_cast<size_t>(i) * (N * MM)) * LW;
  }
  //! per-lane byte offsets of half s (0 / 1) of X, U, cost; offB addresses single-copy arrays
  NMPC_D unsigned offX(int s) const
  {
    return (static_cast<unsigned>(s) * static_cast<unsigned>(rowsX() * LW) + lane) * 8u;
  }
  NMPC_D unsigned offU(int s) const
  {
    return (static_cast<unsigned>(s) * static_cast<unsigned>(rowsU() * LW) + lane) * 8u;
  }
  NMPC_D unsigned offC(int s) const
  {
    return (static_cast<unsigned>(s) * static_cast<unsigned>((T + 1) * LW) + lane) * 8u;
  }
  NMPC_D unsigned offB() const
  {
    return lane * 8u;
  }
  //! address of per-instance element `row` of a single-copy [tile][rows][64] array of any type
  template<class Tp>
  NMPC_D Tp & elem(Tp * array, size_t rows, size_t row) const
  {
    return tileBase(array, rows)[row * LW + lane];
  }

  NMPC_D int inputDimAt(double t) const
  {
    if constexpr(kDyn)
    {
      return problem.inputDim(t);
    }
    else
    {
      return M;
    }
  }

  NMPC_D void loadX(const double * row, unsigned off, StateDimVector & x) const
  {
#pragma unroll
    for(int j = 0; j < N; j++)
    {
      x[j] = ld(row + j * LW, off);
    }
  }
  NMPC_D void storeX(double * row, unsigned off, const StateDimVector & x) const
  {
#pragma unroll
    for(int j = 0; j < N; j++)
    {
      st(row + j * LW, off, x[j]);
    }
  }
  NMPC_D void loadU(const double * row, unsigned off, InputDimVector & u, int m) const
  {
    u.resize(m);
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      u[a] = (a < m) ? ld(row + a * LW, off) : 0.0;
    }
  }
  NMPC_D void storeU(double * row, unsigned off, const InputDimVector & u, int m) const
  {
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      st(row + a * LW, off, (a < m) ? u[a] : 0.0);
    }
  }

  // -------------------------------------------------------------------------------------------------
  // initial rollout    DDPSolver.hpp:83-95
  // -------------------------------------------------------------------------------------------------
  NMPC_D void initialRollout()
  {
    const unsigned ox = offX(sel), ou = offU(sel), oc = offC(sel);
    StateDimVector x;
    loadX(tileBase(buf.x0, N), offB(), x);
    storeX(xRow(0), ox, x);
    double J = 0;
    for(int i = 0; i < T; i++)
    {
      const double t = current_t + i * problem.dt();
      const int m = inputDimAt(t);
      elem(buf.input_dim, T, i) = m;
      InputDimVector u;
      loadU(uRow(i), ou, u, m);
      storeU(uRow(i), ou, u, m); // zero the padding beyond inputDim(t)
      const StateDimVector xn = problem.stateEq(t, x, u);
      const double c = problem.runningCost(t, x, u);
      st(costRow(i), oc, c);
      J += c;
      storeX(xRow(i + 1), ox, xn);
      x = xn;
    }
    const double terminal_t = current_t + T * problem.dt();
    const double cT = problem.terminalCost(terminal_t, x);
    st(costRow(T), oc, cT);
    J += cT;
    J_cur = J;
  }

  // -------------------------------------------------------------------------------------------------
  // dense helpers on per-lane arrays (column-major)
  // -------------------------------------------------------------------------------------------------
  /** s += a * b, skipping factors that the COMPILER knows to be structurally 0 or 1.
      After the problem functor is inlined and the loops are unrolled, the entries a model writes as literal
      0 / 1 (state_eq_deriv_x.setZero(), identity diagonals, diagonal cost Hessians) are compile-time
      constants; __builtin_constant_p folds after inlining, so those terms cost no instruction while entries
      that are run-time values take the ordinary FMA path with no branch.  Dropping x + 0*y is exact for finite
      y (it differs from the reference only where that would have produced NaN from 0*Inf). */
  NMPC_D static void macc(double & s, double a, double b)
  {
    if((__builtin_constant_p(a) && a == 0.0) || (__builtin_constant_p(b) && b == 0.0))
    {
      return;
    }
    const bool a_one = __builtin_constant_p(a) && a == 1.0;
    const bool b_one = __builtin_constant_p(b) && b == 1.0;
    const double p = a_one ? b : (b_one ? a : a * b);
    if(__builtin_constant_p(s) && s == 0.0)
    {
      s = p;
    }
    else
    {
      s += p;
    }
  }

  /** l + s, skipping a structurally-zero l. */
  NMPC_D static double addc(double l, double s)
  {
    if(__builtin_constant_p(l) && l == 0.0)
    {
      return s;
    }
    if(__builtin_constant_p(s) && s == 0.0)
    {
      return l;
    }
    return l + s;
  }

  /** In-place factorisation A = L D L^T of the leading n x n block (leading dimension LD): on exit the strict
      lower part holds the unit-lower L and inv_d[k] = 1 / d_k.
      d_k is the same quantity as the squared Cholesky pivot of Eigen::LLT (A_kk minus the already eliminated
      part), so the failure test "pivot <= 0" (NaN passes) selects the same matrices as the reference's
      LLT::info() == NumericalIssue (DDPSolver.hpp:500-508, BoxQP.h:229-238) up to rounding of the pivot.
      One reciprocal per pivot replaces LLT's sqrt + 2 divisions per right-hand-side entry: on gfx950 an fp64
      divide costs ~10 FMAs and a sqrt ~13 (profiles/ubench_r01.txt). */
  template<int LD>
  NMPC_D static bool ldltInPlace(double * A, double * inv_d, int n)
  {
    bool ok = true;
#pragma unroll kU
    for(int k = 0; k < LD; k++)
    {
      if(k < n && ok)
      {
        double d = A[k + k * LD];
#pragma unroll kU
        for(int j = 0; j < LD; j++)
        {
          if(j < k)
          {
            d -= (A[k + j * LD] * A[k + j * LD]) * A[j + j * LD];
          }
        }
        if(d <= 0)
        {
          ok = false;
        }
        else
        {
          A[k + k * LD] = d;
          const double r = recipFast(d); // d > 0 here
          inv_d[k] = r;
#pragma unroll kU
          for(int i = 0; i < LD; i++)
          {
            if(i > k && i < n)
            {
              double s = A[i + k * LD];
#pragma unroll kU
              for(int j = 0; j < LD; j++)
              {
                if(j < k)
                {
                  s -= (A[i + j * LD] * A[k + j * LD]) * A[j + j * LD];
                }
              }
              A[i + k * LD] = s * r;
            }
          }
        }
      }
    }
    return ok;
  }

  /** Solve (L D L^T) x = rhs in place for one right-hand side with element stride RS. */
  template<int LD, int RS>
  NMPC_D static void ldltSolveInPlace(const double * A, const double * inv_d, int n, double * rhs)
  {
#pragma unroll kU
    for(int i = 0; i < LD; i++)
    {
      if(i < n)
      {
        double s = rhs[i * RS];
#pragma unroll kU
        for(int j = 0; j < LD; j++)
        {
          if(j < i)
          {
            s -= A[i + j * LD] * rhs[j * RS];
          }
        }
        rhs[i * RS] = s;
      }
    }
#pragma unroll kU
    for(int ii = 0; ii < LD; ii++)
    {
      const int i = LD - 1 - ii;
      if(i < n)
      {
        double s = rhs[i * RS] * inv_d[i];
#pragma unroll kU
        for(int j = 0; j < LD; j++)
        {
          if(j > i && j < n)
          {
            s -= A[j + i * LD] * rhs[j * RS];
          }
        }
        rhs[i * RS] = s;
      }
    }
  }

  // -------------------------------------------------------------------------------------------------
  // BoxQP::solve    BoxQP.h:141-347   (H, factor: leading dimension MM)
  // -------------------------------------------------------------------------------------------------
  struct QPOut
  {
    double x[MM];
    double fac[MM * MM]; //!< L D L^T factor of H[free, free], leading dimension MM (llt_free_, BoxQP.h:386)
    double inv_d[MM];
    int free_idx[MM];
    int n_free;
    int retval;
  };

  NMPC_D static double qpObjective(int m, const double * H, const double * g, const double * x)
  {
    double xg = 0;
#pragma unroll kU
    for(int i = 0; i < MM; i++)
    {
      if(i < m)
      {
        xg += x[i] * g[i];
      }
    }
    double xHx = 0;
#pragma unroll kU
    for(int i = 0; i < MM; i++)
    {
      if(i < m)
      {
        double hx = 0;
#pragma unroll kU
        for(int j = 0; j < MM; j++)
        {
          if(j < m)
          {
            hx += H[i + j * MM] * x[j];
          }
        }
        xHx += x[i] * hx;
      }
    }
    return xg + 0.5 * xHx;
  }

  /** The same algorithm for ONE input (MM = 1, m = 1), statement for statement, on scalars: the general code below keeps
      its iterates in the arrays of QPOut and indexes them through free_idx, which for one input is private-memory traffic
      and ~700 instructions per timestep where a few dozen do.  Every comparison, return code and rounding of the general
      code is kept (the factor of a 1 x 1 block is the pivot, its solve one product with the reciprocal).  Box-constrained
      cart-pole solve on the quad kernel: backward pass 277 k -> 251 k cycles (scripts/profile_quad_box.py); what remains is
      the algorithm's own: two to three projected-Newton iterations per timestep with an IEEE division per Armijo trial,
      the trials themselves (40 k cycles per pass) and trip counts that differ between the four instances of a wave. */
  NMPC_D void boxQP1(double H, double g, double lower, double upper, double initial_x, QPOut & out) const
  {
    auto clampToBox = [&](double v) { return fmax(fmin(v, upper), lower); };
    auto objective = [&](double v)
    {
      const double xg = v * g;
      const double hx = H * v;
      const double xHx = v * hx;
      return xg + 0.5 * xHx;
    };
    double x = clampToBox(initial_x); // BoxQP.h:148
    double obj = objective(x);
    double old_obj = obj;
    int retval = 0, n_free = 0;
    double inv_d = 0;
    for(int iter = 1;; iter++)
    {
      // relative improvement    BoxQP.h:176-181
      if(iter > 1 && (old_obj - obj) < cfg.qp_rel_improve_thre * fabs(old_obj))
      {
        retval = 4;
        break;
      }
      old_obj = obj;
      const double grad = g + H * x; // BoxQP.h:184
      // clamped / free sets (exact == compare)    BoxQP.h:187-213
      const bool clamped = (x == lower && grad > 0) || (x == upper && grad < 0);
      n_free = clamped ? 0 : 1;
      if(clamped)
      {
        retval = 6;
        break;
      }
      // factorise the free block iff the clamped set changed    BoxQP.h:216-241   (a clamped input ends the iteration above,
      // so the set is {free} from the first iteration on)
      if(iter == 1)
      {
        if(H <= 0)
        {
          retval = -1;
          break;
        }
        inv_d = recipFast(H);
      }
      // free gradient norm    BoxQP.h:244-253
      const double grad_norm = grad * grad;
      if(grad_norm < cfg.qp_grad_thre * cfg.qp_grad_thre)
      {
        retval = 5;
        break;
      }
      // Newton direction    BoxQP.h:256-279
      const double rhs = g * inv_d;
      const double search_dir = -1 * rhs - x;
      // descent check    BoxQP.h:282-291
      const double sdg = search_dir * grad;
      if(sdg > 1e-10)
      {
        retval = -2;
        break;
      }
      // Armijo line search with projection    BoxQP.h:294-309
      double step = 1;
      double x_cand = clampToBox(x + step * search_dir);
      double obj_cand = objective(x_cand);
      while((obj_cand - old_obj) / (step * sdg) < cfg.qp_armijo_param)
      {
        step = step * cfg.qp_step_factor;
        x_cand = clampToBox(x + step * search_dir);
        obj_cand = objective(x_cand);
        if(step < cfg.qp_min_step)
        {
          retval = 2; // leaves only the inner loop (BoxQP.h:304-308)
          break;
        }
      }
      // accept    BoxQP.h:328-329
      x = x_cand;
      obj = obj_cand;
      if(iter == cfg.qp_max_iter)
      {
        retval = 1; // BoxQP.h:332-336
        break;
      }
    }
    out.x[0] = x;
    out.fac[0] = H;
    out.inv_d[0] = inv_d;
    out.free_idx[0] = 0;
    out.n_free = n_free;
    out.retval = retval;
  }

  /** boxQP1 with the paths nearly every timestep takes as branch-free code.  Measured on the oracle (cart-pole, +-15 N,
      171 900 BoxQP calls): 52 % of the calls end in their first iteration (clamped at the start: 6; zero gradient: 5), 47 % in
      the second after one full projected Newton step (5 or 6; 4 on 24 calls), and 1 % go on — an Armijo test fails or a third
      iteration starts.  The loops of boxQP1 cost a wave the trip count of its slowest lane and a dozen exec-mask
      branches per trip; here both iterations are evaluated unconditionally with the statements of boxQP1 (same
      expressions, so the same roundings and the same exact compares) and the exit is selected afterwards.  A lane whose
      call is not decided by then runs boxQP1 from the start: the results cannot differ. */
  NMPC_D void boxQP1Fast(double H, double g, double lower, double upper, double initial_x, QPOut & out) const
  {
    auto clampToBox = [&](double v) { return fmax(fmin(v, upper), lower); };
    auto objective = [&](double v)
    {
      const double xg = v * g;
      const double hx = H * v;
      const double xHx = v * hx;
      return xg + 0.5 * xHx;
    };
    const double thre2 = cfg.qp_grad_thre * cfg.qp_grad_thre;
    // ---- iteration 1    BoxQP.h:148-329
    const double x0 = clampToBox(initial_x);
    const double obj0 = objective(x0);
    const double grad0 = g + H * x0;
    const bool clamped0 = (x0 == lower && grad0 > 0) || (x0 == upper && grad0 < 0); // -> 6
    const bool indefinite = H <= 0; // -> -1
    const double inv_d = recipFast(H);
    const bool flat0 = grad0 * grad0 < thre2; // -> 5
    const double rhs = g * inv_d;
    const double dir0 = -1 * rhs - x0;
    const double sdg0 = dir0 * grad0;
    const bool ascent0 = sdg0 > 1e-10; // -> -2
    const double x1 = clampToBox(x0 + dir0); // step = 1
    const double obj1 = objective(x1);
    // Armijo test of the full step, (obj1 - obj0) / (1 * sdg0) < armijo_param (BoxQP.h:299): for sdg0 < 0 the quotient is
    // >= armijo_param iff obj1 - obj0 <= armijo_param sdg0; sixteen ulps away from that boundary the rounding of the division
    // (and of the product here) cannot change the outcome, so the IEEE division — a dozen dependent quarter-rate
    // instructions on the recursion's chain — is left to boxQP1 for the calls that are not decided without it.
    const double armijo_rhs = cfg.qp_armijo_param * sdg0;
    const bool accept0 = sdg0 < 0 && (obj1 - obj0) < armijo_rhs - fabs(armijo_rhs) * 0x1p-48;
    const bool shrink0 = !accept0; // the Armijo loop runs (or the test is too close to call): not decided here
    // ---- iteration 2    BoxQP.h:176-253
    const bool stalled1 = (obj0 - obj1) < cfg.qp_rel_improve_thre * fabs(obj0); // -> 4
    const double grad1 = g + H * x1;
    const bool clamped1 = (x1 == lower && grad1 > 0) || (x1 == upper && grad1 < 0); // -> 6
    const bool flat1 = grad1 * grad1 < thre2; // -> 5
    // ---- the exit taken, in the order of the statements
    const bool ends1 = clamped0 || indefinite || flat0 || ascent0;
    const bool ends2 = cfg.qp_max_iter == 1 || stalled1 || clamped1 || flat1;
    if(!ends1 && (shrink0 || !ends2))
    {
      boxQP1(H, g, lower, upper, initial_x, out);
      return;
    }
    const int ret1 = clamped0 ? 6 : (indefinite ? -1 : (flat0 ? 5 : -2));
    const int ret2 = (cfg.qp_max_iter == 1) ? 1 : (stalled1 ? 4 : (clamped1 ? 6 : 5));
    out.x[0] = ends1 ? x0 : x1;
    out.fac[0] = H;
    out.inv_d[0] = (clamped0 || indefinite) ? 0.0 : inv_d;
    out.free_idx[0] = 0;
    out.n_free = ends1 ? (clamped0 ? 0 : 1) : ((cfg.qp_max_iter != 1 && !stalled1 && clamped1) ? 0 : 1);
    out.retval = ends1 ? ret1 : ret2;
  }

  NMPC_D void boxQP(int m,
                    const double * H,
                    const double * g,
                    const double * lower,
                    const double * upper,
                    const double * initial_x,
                    QPOut & out) const
  {
    if constexpr(MM == 1)
    {
      if(m == 1)
      {
        // (lane-per-instance kernels too: a wave whose 64 instances are all decided by the two branch-free iterations skips
        // the loop; two-wave kernel, cart-pole +-15 N at 8192 instances: 2.91 -> 2.76 ms, same bits)
        boxQP1Fast(H[0], g[0], lower[0], upper[0], initial_x[0], out);
        return;
      }
    }
    double * x = out.x;
#pragma unroll kU
    for(int i = 0; i < MM; i++)
    {
      x[i] = (i < m) ? fmax(fmin(initial_x[i], upper[i]), lower[i]) : 0.0; // BoxQP.h:148
    }
    double obj = qpObjective(m, H, g, x);
    double old_obj = obj;
    out.retval = 0;
    out.n_free = 0;
    double grad[MM];
    unsigned clamped = 0, old_clamped = 0;
    double search_dir[MM], x_cand[MM], rhs[MM];
    for(int iter = 1;; iter++)
    {
      // relative improvement    BoxQP.h:176-181
      if(iter > 1 && (old_obj - obj) < cfg.qp_rel_improve_thre * fabs(old_obj))
      {
        out.retval = 4;
        break;
      }
      old_obj = obj;

      // gradient    BoxQP.h:184
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        if(i < m)
        {
          double hx = 0;
#pragma unroll kU
          for(int j = 0; j < MM; j++)
          {
            if(j < m)
            {
              hx += H[i + j * MM] * x[j];
            }
          }
          grad[i] = g[i] + hx;
        }
      }

      // clamped / free sets (exact == compare)    BoxQP.h:187-213
      old_clamped = clamped;
      clamped = 0;
      int nf = 0;
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        if(i < m)
        {
          const bool c = (x[i] == lower[i] && grad[i] > 0) || (x[i] == upper[i] && grad[i] < 0);
          if(c)
          {
            clamped |= (1u << i);
          }
          else
          {
            out.free_idx[nf] = i;
            nf++;
          }
        }
      }
      out.n_free = nf;
      if(nf == 0)
      {
        out.retval = 6;
        break;
      }

      // factorise the free block iff the clamped set changed    BoxQP.h:216-241
      if(iter == 1 || clamped != old_clamped)
      {
        for(int i = 0; i < nf; i++)
        {
          for(int j = 0; j < nf; j++)
          {
            out.fac[i + j * MM] = H[out.free_idx[i] + out.free_idx[j] * MM];
          }
        }
        if(!ldltInPlace<MM>(out.fac, out.inv_d, nf))
        {
          out.retval = -1;
          break;
        }
      }

      // free gradient norm    BoxQP.h:244-253
      double grad_norm = 0;
      for(int i = 0; i < nf; i++)
      {
        grad_norm += grad[out.free_idx[i]] * grad[out.free_idx[i]];
      }
      if(grad_norm < cfg.qp_grad_thre * cfg.qp_grad_thre)
      {
        out.retval = 5;
        break;
      }

      // Newton direction on the free dimensions    BoxQP.h:256-279
      for(int i = 0; i < nf; i++)
      {
        double s = 0;
#pragma unroll kU
        for(int j = 0; j < MM; j++)
        {
          if(j < m && ((clamped >> j) & 1u))
          {
            s += H[out.free_idx[i] + j * MM] * x[j];
          }
        }
        rhs[i] = g[out.free_idx[i]] + s;
      }
      ldltSolveInPlace<MM, 1>(out.fac, out.inv_d, nf, rhs);
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        search_dir[i] = 0;
      }
      for(int i = 0; i < nf; i++)
      {
        search_dir[out.free_idx[i]] = -1 * rhs[i] - x[out.free_idx[i]];
      }

      // descent check    BoxQP.h:282-291
      double sdg = 0;
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        if(i < m)
        {
          sdg += search_dir[i] * grad[i];
        }
      }
      if(sdg > 1e-10)
      {
        out.retval = -2;
        break;
      }

      // Armijo line search with projection    BoxQP.h:294-309
      double step = 1;
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        x_cand[i] = (i < m) ? fmax(fmin(x[i] + step * search_dir[i], upper[i]), lower[i]) : 0.0;
      }
      double obj_cand = qpObjective(m, H, g, x_cand);
      while((obj_cand - old_obj) / (step * sdg) < cfg.qp_armijo_param)
      {
        step = step * cfg.qp_step_factor;
#pragma unroll kU
        for(int i = 0; i < MM; i++)
        {
          x_cand[i] = (i < m) ? fmax(fmin(x[i] + step * search_dir[i], upper[i]), lower[i]) : 0.0;
        }
        obj_cand = qpObjective(m, H, g, x_cand);
        if(step < cfg.qp_min_step)
        {
          out.retval = 2; // leaves only the inner loop (BoxQP.h:304-308)
          break;
        }
      }

      // accept    BoxQP.h:328-329
#pragma unroll kU
      for(int i = 0; i < MM; i++)
      {
        x[i] = x_cand[i];
      }
      obj = obj_cand;
      if(iter == cfg.qp_max_iter)
      {
        out.retval = 1; // BoxQP.h:332-336
        break;
      }
    }
  }

  /** What DDPSolver.hpp:473-497 reads after BoxQP::solve(), with the free set as a mask (boxQPMasked). */
  struct QPOutMasked
  {
    double x[MM];
    double fac[MM * MM], inv_d[MM]; //!< L D L^T of H with the clamped rows / columns replaced by the identity's
    unsigned free; //!< bit a: input a is free (0 if retval == 6)
    int retval;
  };
  /** K column: - H[free, free]^-1 col[free] on the free rows, zero on the clamped ones (DDPSolver.hpp:482-496). */
  NMPC_D static void maskedGainColumn(const QPOutMasked & qp, double * col)
  {
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      col[a] = ((qp.free >> a) & 1u) ? col[a] : 0.0;
    }
    ldltSolveInPlace<MM, 1>(qp.fac, qp.inv_d, MM, col);
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      col[a] = ((qp.free >> a) & 1u) ? -1 * col[a] : 0.0;
    }
  }
  /** boxQP for a STATIC input dimension (m = MM) without index lists.  The general code addresses its iterates through
      free_idx[]: run-time indices into per-lane arrays, i.e. private (scratch) memory — 752 bytes per lane and most of a
      box-constrained timestep on the matrix-core kernels, where every lane of a wavefront runs the QP of its instance.
      Here the free set is a bit mask and every loop has compile-time bounds: the free block is H with the clamped rows and
      columns replaced by those of the identity, which factorisation and substitutions pass through with exact zeros
      (x - 0 * y = x), so the free entries see the same operations in the same order as on the compacted block, and sums
      over the free / clamped indices add + 0 for the others.  Statement for statement BoxQP.h:141-347 otherwise. */
  NMPC_D void boxQPMasked(const double * H,
                          const double * g,
                          const double * lower,
                          const double * upper,
                          const double * initial_x,
                          QPOutMasked & out) const
  {
    double * x = out.x;
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      x[i] = fmax(fmin(initial_x[i], upper[i]), lower[i]); // BoxQP.h:148
      out.inv_d[i] = 0;
    }
    double obj = qpObjective(MM, H, g, x);
    double old_obj = obj;
    out.retval = 0;
    out.free = 0;
    double grad[MM];
    unsigned clamped = 0, old_clamped = 0;
    double search_dir[MM], x_cand[MM], rhs[MM];
    for(int iter = 1;; iter++)
    {
      if(iter > 1 && (old_obj - obj) < cfg.qp_rel_improve_thre * fabs(old_obj)) // BoxQP.h:176-181
      {
        out.retval = 4;
        break;
      }
      old_obj = obj;
#pragma unroll
      for(int i = 0; i < MM; i++) // BoxQP.h:184
      {
        double hx = 0;
#pragma unroll
        for(int j = 0; j < MM; j++)
        {
          hx += H[i + j * MM] * x[j];
        }
        grad[i] = g[i] + hx;
      }
      old_clamped = clamped; // BoxQP.h:187-213 (exact == compares)
      clamped = 0;
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        const bool c = (x[i] == lower[i] && grad[i] > 0) || (x[i] == upper[i] && grad[i] < 0);
        clamped |= c ? (1u << i) : 0u;
      }
      out.free = ~clamped & ((1u << MM) - 1u);
      if(out.free == 0)
      {
        out.retval = 6;
        break;
      }
      if(iter == 1 || clamped != old_clamped) // BoxQP.h:216-241
      {
#pragma unroll
        for(int i = 0; i < MM; i++)
        {
#pragma unroll
          for(int j = 0; j < MM; j++)
          {
            const bool both = (((clamped >> i) | (clamped >> j)) & 1u) == 0;
            out.fac[i + j * MM] = both ? H[i + j * MM] : ((i == j) ? 1.0 : 0.0);
          }
        }
        if(!ldltInPlace<MM>(out.fac, out.inv_d, MM))
        {
          out.retval = -1;
          break;
        }
      }
      double grad_norm = 0; // BoxQP.h:244-253
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        grad_norm += ((clamped >> i) & 1u) ? 0.0 : grad[i] * grad[i];
      }
      if(grad_norm < cfg.qp_grad_thre * cfg.qp_grad_thre)
      {
        out.retval = 5;
        break;
      }
#pragma unroll
      for(int i = 0; i < MM; i++) // BoxQP.h:256-279
      {
        double sum = 0;
#pragma unroll
        for(int j = 0; j < MM; j++)
        {
          sum += ((clamped >> j) & 1u) ? H[i + j * MM] * x[j] : 0.0;
        }
        rhs[i] = ((clamped >> i) & 1u) ? 0.0 : g[i] + sum;
      }
      ldltSolveInPlace<MM, 1>(out.fac, out.inv_d, MM, rhs);
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        search_dir[i] = ((clamped >> i) & 1u) ? 0.0 : -1 * rhs[i] - x[i];
      }
      double sdg = 0; // BoxQP.h:282-291
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        sdg += search_dir[i] * grad[i];
      }
      if(sdg > 1e-10)
      {
        out.retval = -2;
        break;
      }
      double step = 1; // BoxQP.h:294-309
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        x_cand[i] = fmax(fmin(x[i] + step * search_dir[i], upper[i]), lower[i]);
      }
      double obj_cand = qpObjective(MM, H, g, x_cand);
      while((obj_cand - old_obj) / (step * sdg) < cfg.qp_armijo_param)
      {
        step = step * cfg.qp_step_factor;
#pragma unroll
        for(int i = 0; i < MM; i++)
        {
          x_cand[i] = fmax(fmin(x[i] + step * search_dir[i], upper[i]), lower[i]);
        }
        obj_cand = qpObjective(MM, H, g, x_cand);
        if(step < cfg.qp_min_step)
        {
          out.retval = 2; // leaves only the inner loop (BoxQP.h:304-308)
          break;
        }
      }
#pragma unroll
      for(int i = 0; i < MM; i++) // BoxQP.h:328-329
      {
        x[i] = x_cand[i];
      }
      obj = obj_cand;
      if(iter == cfg.qp_max_iter)
      {
        out.retval = 1; // BoxQP.h:332-336
        break;
      }
    }
    out.free = (out.retval == 6) ? 0u : out.free;
  }

  // -------------------------------------------------------------------------------------------------
  // backward pass    DDPSolver.hpp:342-534, with the linearisation of :160-180 evaluated on the fly
  // -------------------------------------------------------------------------------------------------
  NMPC_D bool backwardPass()
  {
    double Vx[N], Vxx[N * N];
    {
      // calcTerminalCostDeriv at x[T]    :179-180, :346-347
      StateDimVector xT, vx;
      StateStateDimMatrix vxx;
      loadX(xRow(T), offX(sel), xT);
      problem.calcTerminalCostDeriv(current_t + T * problem.dt(), xT, vx, vxx);
#pragma unroll kU
      for(int j = 0; j < N; j++)
      {
        Vx[j] = vx[j];
      }
#pragma unroll kU
      for(int e = 0; e < N * N; e++)
      {
        Vxx[e] = vxx.data()[e];
      }
    }
    dV0 = 0;
    dV1 = 0;
    k_rel_norm = 0;
    bool ok = true;
    double k_next[MM]; // k_list_[i+1] (BoxQP warm start)
    int m_next = -1;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      k_next[a] = 0;
    }

    // software prefetch: (x, u) of step i-1 are requested while step i is being processed
    StateDimVector x_pref;
    InputDimVector u_pref;
    const unsigned ox = offX(sel), ou = offU(sel), ob = offB();
    loadX(xRow(T - 1), ox, x_pref);
    loadU(uRow(T - 1), ou, u_pref, MM);

    for(int i = T - 1; i >= 0 && ok; i--)
    {
      const double t = current_t + i * problem.dt();
      const int m = inputDimAt(t);

      // ---- Step 1 of procOnce for this timestep: derivatives at (x_i, u_i)    :160-178
      StateDimVector x = x_pref;
      InputDimVector u = u_pref;
      u.resize(m);
      if(i > 0)
      {
        loadX(xRow(i - 1), ox, x_pref);
        loadU(uRow(i - 1), ou, u_pref, MM);
      }
      StateStateDimMatrix Fx, Lxx;
      StateInputDimMatrix Fu, Lxu;
      StateDimVector Lx;
      InputDimVector Lu;
      InputInputDimMatrix Luu;
      Fu.resize(N, m);
      Lxu.resize(N, m);
      Lu.resize(m);
      Luu.resize(m, m);
      problem.calcStateEqDeriv(t, x, u, Fx, Fu);
      problem.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);

      // ---- Q terms    :386-408  (products left to right through a temporary)
      double Qu[MM], Qx[N], Qux[MM * N], Quu[MM * MM], Qxx[N * N];
      double FuT_V[MM * N]; // Fu^T Vxx   (m x N, ld MM)
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        if(a < m)
        {
          double s = 0;
#pragma unroll kU
          for(int r = 0; r < N; r++)
          {
            macc(s, Fu(r, a), Vx[r]);
          }
          Qu[a] = addc(Lu[a], s);
        }
      }
#pragma unroll kU
      for(int a = 0; a < N; a++)
      {
        double s = 0;
#pragma unroll kU
        for(int r = 0; r < N; r++)
        {
          macc(s, Fx(r, a), Vx[r]);
        }
        Qx[a] = addc(Lx[a], s);
      }
#pragma unroll kU
      for(int c = 0; c < N; c++)
      {
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            double s = 0;
#pragma unroll kU
            for(int r = 0; r < N; r++)
            {
              macc(s, Fu(r, a), Vxx[r + c * N]);
            }
            FuT_V[a + c * MM] = s;
          }
        }
      }
#pragma unroll kU
      for(int c = 0; c < N; c++)
      {
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            double s = 0;
#pragma unroll kU
            for(int r = 0; r < N; r++)
            {
              macc(s, FuT_V[a + r * MM], Fx(r, c));
            }
            Qux[a + c * MM] = addc(Lxu(c, a), s);
          }
        }
      }
#pragma unroll kU
      for(int bb = 0; bb < MM; bb++)
      {
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m && bb < m)
          {
            double s = 0;
#pragma unroll kU
            for(int r = 0; r < N; r++)
            {
              macc(s, FuT_V[a + r * MM], Fu(r, bb));
            }
            Quu[a + bb * MM] = addc(Luu(a, bb), s);
          }
        }
      }
      {
        double FxT_V[N * N]; // Fx^T Vxx
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
#pragma unroll kU
          for(int a = 0; a < N; a++)
          {
            double s = 0;
#pragma unroll kU
            for(int r = 0; r < N; r++)
            {
              macc(s, Fx(r, a), Vxx[r + c * N]);
            }
            FxT_V[a + c * N] = s;
          }
        }
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
#pragma unroll kU
          for(int a = 0; a < N; a++)
          {
            double s = 0;
#pragma unroll kU
            for(int r = 0; r < N; r++)
            {
              macc(s, FxT_V[a + r * N], Fx(r, c));
            }
            Qxx[a + c * N] = addc(Lxx(a, c), s);
          }
        }
      }

      // ---- regularisation    :421-441
      double Qux_reg[MM * N], Quu_F[MM * MM];
      if(cfg.reg_type == 2)
      {
        // Vxx_reg = Vxx + lambda I: recompute the products from Vxx_reg as the reference writes them
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            if(a < m)
            {
              double s = 0;
#pragma unroll kU
              for(int r = 0; r < N; r++)
              {
                const double v = (r == c) ? (Vxx[r + c * N] + lambda) : Vxx[r + c * N];
                macc(s, Fu(r, a), v);
              }
              FuT_V[a + c * MM] = s;
            }
          }
        }
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            if(a < m)
            {
              double s = 0;
#pragma unroll kU
              for(int r = 0; r < N; r++)
              {
                macc(s, FuT_V[a + r * MM], Fx(r, c));
              }
              Qux_reg[a + c * MM] = addc(Lxu(c, a), s);
            }
          }
        }
#pragma unroll kU
        for(int bb = 0; bb < MM; bb++)
        {
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            if(a < m && bb < m)
            {
              double s = 0;
#pragma unroll kU
              for(int r = 0; r < N; r++)
              {
                macc(s, FuT_V[a + r * MM], Fu(r, bb));
              }
              Quu_F[a + bb * MM] = addc(Luu(a, bb), s);
            }
          }
        }
      }
      else
      {
        // reg_type 1: Vxx_reg == Vxx, so Qux_reg and the product part of Quu_F are the very same
        // floating-point expressions as Qux / Quu (:427,:433 vs :390,:399); then Quu_F.diagonal() += lambda (:438-441)
#pragma unroll kU
        for(int e = 0; e < MM * N; e++)
        {
          Qux_reg[e] = Qux[e];
        }
#pragma unroll kU
        for(int bb = 0; bb < MM; bb++)
        {
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            Quu_F[a + bb * MM] = (a == bb && cfg.reg_type == 1) ? (Quu[a + bb * MM] + lambda) : Quu[a + bb * MM];
          }
        }
      }

      // ---- gains    :448-517
      double k[MM], K[MM * N];
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        k[a] = 0;
      }
#pragma unroll kU
      for(int e = 0; e < MM * N; e++)
      {
        K[e] = 0;
      }
      if(m > 0)
      {
        if constexpr(kConstrained)
        {
          double initial_k[MM], lo[MM], up[MM];
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            // warm start from k_list_[i+1] when its size matches    :452-467
            initial_k[a] = (i != T - 1 && m_next == m) ? k_next[a] : 0.0;
            lo[a] = inputLimitLo(buf, b, i, a) - u[a]; // :470-472
            up[a] = inputLimitHi(buf, b, i, a) - u[a];
          }
          QPOut qp;
          boxQP(m, Quu_F, Qu, lo, up, initial_k, qp);
          unsigned free_mask = 0;
          for(int j = 0; j < qp.n_free; j++)
          {
            free_mask |= (1u << qp.free_idx[j]);
          }
          elem(buf.qp_ret, T, i) = qp.retval;
          elem(buf.qp_free, T, i) = free_mask;
          if(qp.retval < 0)
          {
            ok = false; // :473-480
          }
          else
          {
#pragma unroll kU
            for(int a = 0; a < MM; a++)
            {
              k[a] = qp.x[a];
            }
            // K rows in free_idxs_ = -llt_free.solve(Qux_reg[free, :]), clamped rows 0    :482-496
            if(qp.n_free > 0)
            {
              for(int c = 0; c < N; c++)
              {
                double col[MM];
                for(int j = 0; j < qp.n_free; j++)
                {
                  col[j] = Qux_reg[qp.free_idx[j] + c * MM];
                }
                ldltSolveInPlace<MM, 1>(qp.fac, qp.inv_d, qp.n_free, col);
                for(int j = 0; j < qp.n_free; j++)
                {
                  K[qp.free_idx[j] + c * MM] = -1 * col[j];
                }
              }
            }
          }
        }
        else
        {
          // LLT(Quu_F); k = -solve(Qu); K = -solve(Qux_reg)    :500-510  (as L D L^T, see ldltInPlace)
          double fac[MM * MM], inv_d[MM];
#pragma unroll kU
          for(int e = 0; e < MM * MM; e++)
          {
            fac[e] = Quu_F[e];
          }
          if(!ldltInPlace<MM>(fac, inv_d, m))
          {
            ok = false;
          }
          else
          {
#pragma unroll kU
            for(int a = 0; a < MM; a++)
            {
              k[a] = Qu[a];
            }
            ldltSolveInPlace<MM, 1>(fac, inv_d, m, k);
#pragma unroll kU
            for(int a = 0; a < MM; a++)
            {
              k[a] = -1 * k[a];
            }
#pragma unroll kU
            for(int c = 0; c < N; c++)
            {
#pragma unroll kU
              for(int a = 0; a < MM; a++)
              {
                K[a + c * MM] = Qux_reg[a + c * MM];
              }
              ldltSolveInPlace<MM, 1>(fac, inv_d, m, &K[c * MM]);
#pragma unroll kU
              for(int a = 0; a < MM; a++)
              {
                K[a + c * MM] = -1 * K[a + c * MM];
              }
            }
          }
        }
      }
      if(!ok)
      {
        break;
      }

      // ---- cost-to-go update with the UNregularised Quu / Qux    :522-526
      {
        double kQu = 0, kQuuk = 0;
        double Quu_k[MM];
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            macc(kQu, k[a], Qu[a]);
            double s = 0;
#pragma unroll kU
            for(int bb = 0; bb < MM; bb++)
            {
              if(bb < m)
              {
                macc(s, Quu[a + bb * MM], k[bb]);
              }
            }
            Quu_k[a] = s;
          }
        }
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            macc(kQuuk, k[a], Quu_k[a]);
          }
        }
        dV0 += kQu;
        dV1 += 0.5 * kQuuk;
      }
      double KtQuu[N * MM]; // K^T Quu   (N x m, ld N)
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
#pragma unroll kU
        for(int r = 0; r < N; r++)
        {
          double s = 0;
#pragma unroll kU
          for(int p = 0; p < MM; p++)
          {
            if(p < m && a < m)
            {
              macc(s, K[p + r * MM], Quu[p + a * MM]);
            }
          }
          KtQuu[r + a * N] = s;
        }
      }
      double Vxx_new[N * N];
#pragma unroll kU
      for(int r = 0; r < N; r++)
      {
        double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            macc(s1, KtQuu[r + a * N], k[a]);
            macc(s2, K[a + r * MM], Qu[a]);
            macc(s3, Qux[a + r * MM], k[a]);
          }
        }
        Vx[r] = ((Qx[r] + s1) + s2) + s3;
      }
#pragma unroll kU
      for(int c = 0; c < N; c++)
      {
#pragma unroll kU
        for(int r = 0; r < N; r++)
        {
          double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            if(a < m)
            {
              macc(s1, KtQuu[r + a * N], K[a + c * MM]);
              macc(s2, K[a + r * MM], Qux[a + c * MM]);
              macc(s3, Qux[a + r * MM], K[a + c * MM]);
            }
          }
          Vxx_new[r + c * N] = ((Qxx[r + c * N] + s1) + s2) + s3;
        }
      }
      // Vxx = 0.5 (Vxx + Vxx^T)    DDPSolver.hpp:527.  IEEE addition commutes, so entries (r, c) and (c, r) of the
      // reference's result are the same bits: each off-diagonal pair is computed once.
#pragma unroll kU
      for(int c = 0; c < N; c++)
      {
#pragma unroll kU
        for(int r = 0; r <= c; r++)
        {
          const double v = 0.5 * (Vxx_new[r + c * N] + Vxx_new[c + r * N]);
          Vxx[r + c * N] = v;
          Vxx[c + r * N] = v;
        }
      }

      // ---- save gains    :529-530, and the running max of |k_i| / (|u_i| + 1)    :217-221
      {
        double * kp = kRow(i);
        double * Kp = KRow(i);
        double kn = 0, un = 0;
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          st(kp + a * LW, ob, (a < m) ? k[a] : 0.0); // entries beyond inputDim(t) are kept zero
          k_next[a] = (a < m) ? k[a] : 0.0;
          if(a < m)
          {
            kn += k[a] * k[a];
            un += u[a] * u[a];
          }
        }
#pragma unroll kU
        for(int e = 0; e < MM * N; e++)
        {
          st(Kp + e * LW, ob, ((e % MM) < m) ? K[e] : 0.0);
        }
        m_next = m;
        // for m == 1 the two norms are |k| and |u| exactly (sqrt(x*x) == |x| up to under/overflow)
        const double knorm = (M == 1) ? fabs(k[0]) : sqrt(kn);
        const double unorm = (M == 1) ? fabs(u[0]) : sqrt(un);
        k_rel_norm = fmax(k_rel_norm, knorm * recipFast(unorm + 1.0));
      }
    }
    return ok;
  }

  // -------------------------------------------------------------------------------------------------
  // forward pass    DDPSolver.hpp:536-560
  // -------------------------------------------------------------------------------------------------
  NMPC_D void forwardPass(double alpha)
  {
    const unsigned ox = offX(sel), ou = offU(sel), ob = offB();
    const unsigned cx = offX(1 - sel), cu = offU(1 - sel), cc = offC(1 - sel);
    StateDimVector xc;
    loadX(xRow(0), ox, xc);
    storeX(xRow(0), cx, xc);
    double J = 0;

    // software prefetch: the nominal trajectory and gains of step i+1 are requested while step i is evaluated
    StateDimVector x_pref;
    InputDimVector u_pref;
    double k_pref[MM], K_pref[MM * N];
    x_pref = xc;
    loadU(uRow(0), ou, u_pref, MM);
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      k_pref[a] = ld(kRow(0) + a * LW, ob);
    }
#pragma unroll kU
    for(int e = 0; e < MM * N; e++)
    {
      K_pref[e] = ld(KRow(0) + e * LW, ob);
    }

    for(int i = 0; i < T; i++)
    {
      const double t = current_t + i * problem.dt();
      const int m = inputDimAt(t);
      const StateDimVector x = x_pref;
      InputDimVector u = u_pref, uc;
      u.resize(m);
      uc.resize(m);
      double kk[MM], KK[MM * N];
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        kk[a] = k_pref[a];
      }
#pragma unroll kU
      for(int e = 0; e < MM * N; e++)
      {
        KK[e] = K_pref[e];
      }
      if(i + 1 < T)
      {
        loadX(xRow(i + 1), ox, x_pref);
        loadU(uRow(i + 1), ou, u_pref, MM);
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          k_pref[a] = ld(kRow(i + 1) + a * LW, ob);
        }
#pragma unroll kU
        for(int e = 0; e < MM * N; e++)
        {
          K_pref[e] = ld(KRow(i + 1) + e * LW, ob);
        }
      }
      // u' = u + alpha k + K (x' - x)    :545-546
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        if(a < m)
        {
          double s = 0;
#pragma unroll kU
          for(int c = 0; c < N; c++)
          {
            s += KK[a + c * MM] * (xc[c] - x[c]);
          }
          uc[a] = (u[a] + alpha * kk[a]) + s;
        }
        else
        {
          uc[a] = 0;
        }
      }
      storeU(uRow(i), cu, uc, m);
      const StateDimVector xn = problem.stateEq(t, xc, uc);
      const double c = problem.runningCost(t, xc, uc);
      st(costRow(i), cc, c);
      J += c;
      storeX(xRow(i + 1), cx, xn);
      xc = xn;
    }
    const double cT = problem.terminalCost(current_t + T * problem.dt(), xc);
    st(costRow(T), cc, cT);
    J += cT;
    J_cand = J;
  }

  // -------------------------------------------------------------------------------------------------
  // solve = setup + optimisation loop    DDPSolver.hpp:26-141, procOnce :143-340
  // -------------------------------------------------------------------------------------------------
  NMPC_D void writeTraceRow(int row, const double * tr) const
  {
    if(cfg.trace_level >= 1 && row < buf.trace_rows)
    {
      double * p = tileBase(buf.trace, static_cast<size_t>(buf.trace_rows) * NMPC_HIP_NTRACE)
                   + (static_cast<size_t>(row) * NMPC_HIP_NTRACE) * LW;
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        st(p + f * LW, offB(), tr[f]);
      }
    }
  }

  NMPC_D void solve()
  {
    unsigned long long ph_start = __builtin_readcyclecounter(), ph_bw = 0, ph_fw = 0, ph_t0 = 0;
    current_t = buf.t0 ? buf.t0[b] : 0.0;
    lambda = cfg.initial_lambda; // :37
    dlambda = cfg.initial_dlambda; // :38
    sel = 0;
    dV0 = dV1 = 0;
    k_rel_norm = 0;
    J_cand = 0;
    initialRollout();

    double tr[NMPC_HIP_NTRACE];
#pragma unroll
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      tr[f] = 0;
    }
    // trace[0]    :98-104
    tr[NMPC_HIP_TRACE_COST] = J_cur;
    tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
    tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
    tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
    writeTraceRow(0, tr);

    int retval = 0;
    int iter = 0;
    for(iter = 1; iter <= cfg.max_iter; iter++)
    {
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        tr[f] = 0;
      }
      tr[NMPC_HIP_TRACE_ITER] = iter;
      tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
      retval = 0;

      // Step 2 (with Step 1 fused in): backward pass with regularisation retries    :188-214
      int n_backward = 1;
      bool bw_failed = false;
      ph_t0 = __builtin_readcyclecounter();
      while(!backwardPass())
      {
        dlambda = fmax(dlambda * cfg.lambda_factor, cfg.lambda_factor);
        lambda = fmax(lambda * dlambda, cfg.lambda_min);
        if(lambda > cfg.lambda_max)
        {
          bw_failed = true;
          break;
        }
        n_backward++;
      }
      ph_bw += __builtin_readcyclecounter() - ph_t0;
      tr[NMPC_HIP_TRACE_N_BACKWARD] = n_backward;
      if(bw_failed)
      {
        retval = -1; // :196-204
      }
      else
      {
        tr[NMPC_HIP_TRACE_K_REL_NORM] = k_rel_norm;
        if(k_rel_norm < cfg.k_rel_norm_thre && lambda < cfg.lambda_thre)
        {
          retval = 1; // :223-231
        }
        else
        {
          // Step 3: backtracking line search    :234-274
          bool forward_pass_success = false;
          double alpha = 0, cost_update_actual = 0, cost_update_expected = 0, cost_update_ratio = 0;
          int ai = 0;
          for(ai = 0; ai < cfg.n_alpha; ai++)
          {
            alpha = cfg.alpha_list[ai];
            ph_t0 = __builtin_readcyclecounter();
            forwardPass(alpha);
            ph_fw += __builtin_readcyclecounter() - ph_t0;
            cost_update_actual = J_cur - J_cand;
            cost_update_expected = -1 * alpha * (dV0 + alpha * dV1);
            cost_update_ratio = cost_update_actual / cost_update_expected;
            if(cost_update_expected < 0)
            {
              cost_update_ratio = (cost_update_actual >= 0 ? 1 : -1); // :251-259
            }
            if(cost_update_ratio > cfg.cost_update_ratio_thre)
            {
              forward_pass_success = true;
              break;
            }
          }
          tr[NMPC_HIP_TRACE_ALPHA] = alpha;
          tr[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL] = cost_update_actual;
          tr[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED] = cost_update_expected;
          tr[NMPC_HIP_TRACE_COST_UPDATE_RATIO] = cost_update_ratio;
          tr[NMPC_HIP_TRACE_ALPHA_IDX] = forward_pass_success ? ai : cfg.n_alpha - 1;
          tr[NMPC_HIP_TRACE_N_FORWARD] = forward_pass_success ? ai + 1 : cfg.n_alpha;

          // Step 4: accept (swap halves instead of copying, :285-287) or reject    :280-333
          if(forward_pass_success)
          {
            sel = 1 - sel;
            J_cur = J_cand;
            if(cost_update_actual < cfg.cost_update_thre)
            {
              retval = 1;
            }
            dlambda = fmin(dlambda / cfg.lambda_factor, 1 / cfg.lambda_factor);
            if(lambda >= cfg.lambda_min)
            {
              lambda *= dlambda;
            }
            else
            {
              lambda = 0;
            }
          }
          else
          {
            dlambda = fmax(dlambda * cfg.lambda_factor, cfg.lambda_factor);
            lambda = fmax(lambda * dlambda, cfg.lambda_min);
            if(lambda > cfg.lambda_max)
            {
              retval = -1;
            }
          }
          tr[NMPC_HIP_TRACE_COST] = J_cur; // :335-337
          tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
          tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
        }
      }
      writeTraceRow(iter, tr);
      if(retval != 0)
      {
        break;
      }
    }

    // results that do not live in the big arrays
#pragma unroll
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      elem(buf.trace_last, NMPC_HIP_NTRACE, f) = tr[f];
    }
    if(buf.phase_ticks != nullptr)
    {
      unsigned long long * p = buf.phase_ticks + static_cast<size_t>(b) * 4;
      p[0] = ph_bw;
      p[1] = ph_fw;
      p[2] = __builtin_readcyclecounter() - ph_start;
    }
    buf.status[b] = retval;
    buf.iters[b] = static_cast<int>(tr[NMPC_HIP_TRACE_ITER]);
    buf.sel[b] = sel;
    elem(buf.dV, 2, 0) = dV0;
    elem(buf.dV, 2, 1) = dV1;
  }
};

/** The solve kernel: grid = Bp / 64 workgroups of one wavefront, lane = instance. */
template<class Problem, bool kConstrained>
__global__ __launch_bounds__(kLanesPerBlock) void ddp_solve_tpi_kernel(const Problem problem,
                                                                        const nmpc_hip_ddp_config cfg,
                                                                        const DeviceBuffers buf)
{
  const int b = blockIdx.x * kLanesPerBlock + threadIdx.x;
  if(b >= buf.B)
  {
    return;
  }
  InstanceSolver<Problem, kConstrained> solver(problem, cfg, buf, b);
  solver.solve();
}

// ---------------------------------------------------------------------------------------------------------
// layout kernels: reference layouts ([B][R], row-major) <-> tile-major device layout ([tile][halves][R][64])
// One workgroup moves a 64-instance x 64-row block through LDS so that both the global read and the global
// write are contiguous 512-byte segments per wavefront.
// ---------------------------------------------------------------------------------------------------------
/** in [B][R] -> out [tile][halves][R][64], written into half `half`.  TIn != TOut converts (the C-ABI exchanges doubles,
    the fp32 tile kernel stores floats). */
template<class TIn, class TOut = TIn>
__global__ __launch_bounds__(256) void batch_major_to_tile_kernel(const TIn * __restrict__ in,
                                                                   TOut * __restrict__ out,
                                                                   int B,
                                                                   int R,
                                                                   int halves,
                                                                   int half)
{
  __shared__ TOut blk[64][65];
  const int r0 = blockIdx.x * 64, tile = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
  for(int k = ty; k < 64; k += 4)
  {
    const int bb = tile * 64 + k, r = r0 + tx;
    blk[k][tx] = (bb < B && r < R) ? static_cast<TOut>(in[static_cast<size_t>(bb) * R + r]) : TOut(0);
  }
  syncThreadsFuzzed(__LINE__);
  TOut * o = out + (static_cast<size_t>(tile) * halves + half) * R * 64;
  for(int k = ty; k < 64; k += 4)
  {
    const int r = r0 + k;
    if(r < R)
    {
      o[static_cast<size_t>(r) * 64 + tx] = blk[tx][k];
    }
  }
}

/** The three input conversions of a solve in ONE launch (a kernel boundary on a stream costs a few microseconds, and a
    solve of the headline batch takes ~470): current_t [B] -> [Bp] (nullptr: zeros), current_x [B][N] -> tile-major,
    initial_u_list [B][RU] -> half 0 of the two tile-major input halves.  grid = (ceil(RU / 64) + ceil(N / 64) + 1, Bp / 64). */
template<class TIn, class TOut = TIn>
__global__ __launch_bounds__(256) void ingest_kernel(const TIn * __restrict__ t0_in,
                                                      TOut * __restrict__ t0_out,
                                                      const TIn * __restrict__ x_in,
                                                      TOut * __restrict__ x_out,
                                                      int N,
                                                      const TIn * __restrict__ u_in,
                                                      TOut * __restrict__ u_out,
                                                      int RU,
                                                      int B)
{
  __shared__ TOut blk[64][65];
  const int nu = (RU + 63) / 64, nx = (N + 63) / 64;
  const int tile = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
  if(static_cast<int>(blockIdx.x) == nu + nx)
  {
    if(ty == 0)
    {
      const int bb = tile * 64 + tx;
      t0_out[bb] = (t0_in && bb < B) ? static_cast<TOut>(t0_in[bb]) : TOut(0);
    }
    return;
  }
  const bool is_u = static_cast<int>(blockIdx.x) < nu;
  const TIn * in = is_u ? u_in : x_in;
  const int R = is_u ? RU : N, halves = is_u ? 2 : 1;
  const int r0 = (is_u ? blockIdx.x : blockIdx.x - nu) * 64;
  for(int k = ty; k < 64; k += 4)
  {
    const int bb = tile * 64 + k, r = r0 + tx;
    blk[k][tx] = (bb < B && r < R) ? static_cast<TOut>(in[static_cast<size_t>(bb) * R + r]) : TOut(0);
  }
  syncThreadsFuzzed(__LINE__);
  TOut * o = (is_u ? u_out : x_out) + static_cast<size_t>(tile) * halves * R * 64;
  for(int k = ty; k < 64; k += 4)
  {
    const int r = r0 + k;
    if(r < R)
    {
      o[static_cast<size_t>(r) * 64 + tx] = blk[tx][k];
    }
  }
}

/** in [tile][halves][R][64], half selected per instance by sel (sel == nullptr: half 0) -> out [B][R].
    row_limit != nullptr: rows >= (row_limit[b] + 1) * row_unit of instance b are written as 0 — the trace of a reused
    handle keeps rows of earlier solves beyond iters[b]; traceDataList() ends at the last iteration (DDPSolver.h:294). */
template<class TIn, class TOut = TIn>
__global__ __launch_bounds__(256) void tile_to_batch_major_kernel(const TIn * __restrict__ in,
                                                                   TOut * __restrict__ out,
                                                                   const int * __restrict__ sel,
                                                                   int B,
                                                                   int R,
                                                                   int halves,
                                                                   const int * __restrict__ row_limit,
                                                                   int row_unit)
{
  __shared__ TOut blk[64][65];
  const int r0 = blockIdx.x * 64, tile = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  {
    const int bb = tile * 64 + tx;
    const int s = (sel != nullptr && bb < B) ? sel[bb] : 0;
    const int valid = (row_limit != nullptr && bb < B) ? (row_limit[bb] + 1) * row_unit : R;
    const TIn * src = in + (static_cast<size_t>(tile) * halves + s) * R * 64;
    for(int k = ty; k < 64; k += 4)
    {
      const int r = r0 + k;
      blk[k][tx] = (r < R && r < valid) ? static_cast<TOut>(src[static_cast<size_t>(r) * 64 + tx]) : TOut(0);
    }
  }
  syncThreadsFuzzed(__LINE__);
  for(int k = ty; k < 64; k += 4)
  {
    const int bb = tile * 64 + k, r = r0 + tx;
    if(bb < B && r < R)
    {
      out[static_cast<size_t>(bb) * R + r] = blk[tx][k];
    }
  }
}

/** Instance-major gain records [B][T][rec] (k_i then K_i, as the fp32 tile kernel stores them) -> one of the reference
    layouts KFF [B][T][MM] / KFB [B][T][N][MM]: `per_step` consecutive words starting at word `offset` of every record. */
template<class TIn>
__global__ __launch_bounds__(256) void gain_records_to_batch_major_kernel(const TIn * __restrict__ ws,
                                                                           double * __restrict__ out,
                                                                           size_t total,
                                                                           int rec,
                                                                           int per_step,
                                                                           int offset)
{
  const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if(idx < total)
  {
    const size_t bt = idx / per_step, w = idx % per_step;
    out[idx] = static_cast<double>(ws[bt * rec + offset + w]);
  }
}
} // namespace hip
} // namespace nmpc_amd
