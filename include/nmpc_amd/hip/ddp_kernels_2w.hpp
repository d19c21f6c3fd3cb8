// gfx950 device code of the batched DDP solver, LANE MAPPING "TPI-2W": one lane per problem instance as in
// ddp_kernels.hpp, but TWO wavefronts per 64 instances in one workgroup, specialised by role and coupled through
// LDS records (one s_barrier per backward timestep, one per four forward timesteps):
//
//   master wave   owns the per-instance solver state and executes only what is on the sequential dependency
//                 chain: the Riccati recursion (backward) and u' -> x' (forward);
//   helper wave   backward: evaluates the problem's derivatives at (x_i, u_i) ONE TIMESTEP AHEAD of the master
//                           (they do not depend on the recursion) and stages them in an LDS record;
//                 forward : takes the master's (x', u') to evaluate runningCost / terminalCost, accumulate the
//                           candidate cost and do all HBM stores (the master prefetches the nominal x, u, k, K itself,
//                           four timesteps ahead); the initial rollout uses the same split.
//
// Every wave has only ONE kind of global memory operation in flight in any pass (backward: helper loads, master
// stores; forward: master loads, helper stores).  gfx9 counts loads and stores in one vmcnt and they may complete out
// of order with respect to each other, so a wave with both kinds pending can only wait for "everything"
// (s_waitcnt vmcnt(0)) — which exposes the full latency of prefetches issued after a store.
//
// Why (profiles/r01_*): the single-wave kernel is bound by the issue rate of its one wavefront (~4 cycles per
// instruction, one wave per SIMD, 64 of 1024 SIMDs busy at B = 4096); HBM is idle.  About 45 % of the backward
// instruction stream (sin/cos, the divide, the Jacobian arithmetic, loads, address math) is off the dependency chain
// and moves to the helper, as do the cost evaluation and all memory traffic of the forward pass.
//
// Structural sparsity survives the hand-off: the master evaluates the problem functor on DUMMY run-time inputs and
// replaces every entry that is not a compile-time constant by the value from the LDS record; the dummy arithmetic
// (sin/cos, divides) is then dead code and is eliminated, while entries the model writes as literal 0 / 1 remain
// constants for macc().  The helper writes every entry, so the two sides cannot disagree about the record layout.
#pragma once

#include <nmpc_amd/hip/ddp_kernels.hpp>

namespace nmpc_amd
{
namespace hip
{
/** \tparam kForwardRecordsOnly the LDS records carry only the forward hand-off (x', u'): for solvers that replace the
    backward pass of this class (ddp_kernels_quad.hpp)
    \tparam kAlphaGroups lane groups of the master / helper wave that hold the SAME instances (ddp_kernels_quad.hpp: 4
    groups of 16): the line search then tries kAlphaGroups step sizes per forward pass, one per group
    \tparam kLdsNominal the forward master takes the nominal (x, u, k, K) of a timestep from LDS records that a THIRD wave
    of the workgroup (forwardPrefetch) keeps two groups of timesteps ahead of it, instead of loading it itself: a lone
    wave pays ~20 cycles per global_load whatever its width (scripts/ubench_issue_cost.hip) and the master issues
    N + 2 m + m N of them per timestep (a fifth of its time on the cart-pole); from LDS the same data is
    (N + 2 m + m N) / 2 ds_read_b128.  For workgroups of 16 instances (ddp_kernels_quad.hpp). */
template<class Problem, bool kConstrained, bool kForwardRecordsOnly = false, int kAlphaGroups = 1, bool kLdsNominal = false>
struct PairSolver : InstanceSolver<Problem, kConstrained>
{
  using Base = InstanceSolver<Problem, kConstrained>;
  using Base::b;
  using Base::buf;
  using Base::cfg;
  using Base::current_t;
  using Base::dlambda;
  using Base::dV0;
  using Base::dV1;
  using Base::J_cand;
  using Base::J_cur;
  using Base::k_rel_norm;
  using Base::lambda;
  using Base::lane;
  using Base::problem;
  using Base::sel;
  using Base::T;
  static constexpr int N = Base::N;
  static constexpr int M = Base::M;
  static constexpr int MM = Base::MM;
  static constexpr int kU = Base::kU;
  static constexpr size_t LW = Base::LW;
  using typename Base::InputDimVector;
  using typename Base::InputInputDimMatrix;
  using typename Base::QPOut;
  using typename Base::StateDimVector;
  using typename Base::StateInputDimMatrix;
  using typename Base::StateStateDimMatrix;

  // ---- LDS record layouts (doubles per lane) ----
  // backward record (helper -> master): derivatives of timestep i and u_i
  static constexpr int oFx = 0;
  static constexpr int oFu = oFx + N * N;
  static constexpr int oLx = oFu + N * MM;
  static constexpr int oLu = oLx + N;
  static constexpr int oLxx = oLu + MM;
  static constexpr int oLuu = oLxx + N * N;
  static constexpr int oLxu = oLuu + MM * MM;
  static constexpr int oU = oLxu + N * MM;
  //! box-constrained solves: lower limit - u_i, upper limit - u_i (DDPSolver.hpp:470-472), requested by the helper with (x_i, u_i)
  //! kBwdAhead timesteps early — on the master they were two loads with a wait behind them on the recursion's chain
  static constexpr int oLoRel = oU + MM;
  static constexpr int oUpRel = oLoRel + MM;
  static constexpr int kBwdRec = oU + MM + (kConstrained ? 2 * MM : 0);
  // forward record (master -> helper): candidate x'_i, u'_i
  static constexpr int oXc = 0;
  static constexpr int oUc = oXc + N;
  static constexpr int kFwdRec = oUc + MM;
  static constexpr int kRec = (kBwdRec > kFwdRec && !kForwardRecordsOnly) ? kBwdRec : kFwdRec;
  //! LDS doubles per workgroup: two record slots + per-lane mailboxes (flags, J_cand)
  //! forward hand-off: one barrier per kFwdGroup timesteps; two groups of record slots + one slot for x'_T
  static constexpr int kFwdGroup = 4;
  static constexpr int kFwdSlots = 2 * kFwdGroup + 1;
  static constexpr int kRecArea = (2 * kRec > kFwdSlots * kFwdRec) ? 2 * kRec : kFwdSlots * kFwdRec;
  //! nominal records (kLdsNominal): [slot = group of kFwdGroup timesteps % 3][timestep of the group][instance][x, u, k, K]
  static constexpr bool kLdsNominalPath = kLdsNominal;
  static constexpr int kNomRec = (N + 2 * MM + MM * N + 1) / 2 * 2; // even: records are read as 16-byte pairs
  static constexpr int kNomSlots = 3;
  static constexpr int kNomInst = 16;
  static constexpr int kNomDoubles = kLdsNominal ? kNomSlots * kFwdGroup * kNomInst * kNomRec : 0;
  //! Waves beside master / helper / prefetcher that roll out further step sizes of a fan-out pass, cost only
  //! (forwardCostOnlyLds; ddp_kernels_quad.hpp: waves 2 and 3): with kAlphaGroups lane groups each, a pass of the line search
  //! covers kStepSizesPerPass step sizes — all eleven of the reference's alpha_list in ONE pass.
  static constexpr int kExtraMasters = (kLdsNominal && kAlphaGroups > 1) ? 2 : 0;
  static constexpr int kStepSizesPerPass = kAlphaGroups * (1 + kExtraMasters);
  //! rows behind the trace row: the extra masters' cost mailboxes (one row each) and a row of pass parameters
  static constexpr int kExtraRows = kLdsNominal ? 3 : 0;
  static constexpr int kNomBase = (kRecArea + 2 + NMPC_HIP_NTRACE + kExtraRows) * static_cast<int>(LW);
  //! + per-lane mailboxes (flags, J_cand) + the last trace row of every lane (kept in LDS until the solve ends)
  static constexpr int kLdsDoubles = kNomBase + kNomDoubles;
  static constexpr size_t kLdsBytes = static_cast<size_t>(kLdsDoubles) * sizeof(double);
  static constexpr bool kFits = kLdsBytes <= 64 * 1024;

  double * lds; //!< workgroup LDS base

  NMPC_D PairSolver(const Problem & p,
                    const nmpc_hip_ddp_config & c,
                    const DeviceBuffers & bf,
                    int global_lane,
                    double * lds_base)
  : Base(p, c, bf, global_lane), lds(lds_base)
  {
  }

  NMPC_D double & rec(int slot, int idx) const
  {
    return lds[(static_cast<size_t>(slot) * kRec + idx) * LW + lane];
  }
  // forward records and mailboxes are per lane OF THE WAVE (= `lane` in the two-wave kernel; with lane groups, every
  // group has its own cells: the groups roll out different step sizes)
  NMPC_D static unsigned waveLane()
  {
    return threadIdx.x % kLanesPerBlock;
  }
  NMPC_D double & frec(int slot, int idx) const
  {
    return lds[(static_cast<size_t>(slot) * kFwdRec + idx) * LW + waveLane()];
  }
  NMPC_D double * nomRec(int slot, int r, unsigned inst) const
  {
    return lds + kNomBase + ((static_cast<size_t>(slot) * kFwdGroup + r) * kNomInst + inst) * kNomRec;
  }
  NMPC_D double & mailFlags() const
  {
    return lds[static_cast<size_t>(kRecArea) * LW + waveLane()];
  }
  NMPC_D double & mailCostAt(unsigned wave_lane) const
  {
    return lds[(static_cast<size_t>(kRecArea) + 1) * LW + wave_lane];
  }
  NMPC_D double & mailCost() const
  {
    return mailCostAt(waveLane());
  }
  NMPC_D double & lastRow(int field) const
  {
    return lds[(static_cast<size_t>(kRecArea) + 2 + field) * LW + waveLane()];
  }
  //! cost of the rollout of lane `wave_lane` of extra master `which` (0, 1)
  NMPC_D double & mailCostExtra(int which, unsigned wave_lane) const
  {
    return lds[(static_cast<size_t>(kRecArea) + 2 + NMPC_HIP_NTRACE + which) * LW + wave_lane];
  }
  //! index into alpha_list of the first step size of the running fan-out pass (written by the master before post())
  NMPC_D double & mailFirstAlpha() const
  {
    return lds[(static_cast<size_t>(kRecArea) + 2 + NMPC_HIP_NTRACE + 2) * LW];
  }
  //! 1.0: the running fan-out pass is a WIDE one (the extra masters roll out too); 0.0: only the master's lane groups do
  NMPC_D double & mailWidePass() const
  {
    return lds[(static_cast<size_t>(kRecArea) + 2 + NMPC_HIP_NTRACE + 2) * LW + 1];
  }
  //! The backward pass of the quad kernel leaves the nominal records of the first three groups of timesteps in LDS (from its last
  //! chunk: ddp_kernels_quad.hpp), so that the prefetching wave of the forward pass behind it starts at group 3 and barrier S does
  //! not wait for a round trip to HBM.  Box-constrained solves only [measured, profiles/r05_c2_chain_ab.txt: cart-pole +- 15 N
  //! 0.930 -> 0.905 ms; the unconstrained headline LOSES 2 % (0.451 -> 0.461 ms: ten more LDS stores per pass on a wave that is
  //! bound by its instruction count, for a round trip the first barrier already hid)].
  static constexpr bool kNominalTail = kLdsNominal && kConstrained;
  //! 1.0: those records are resident; 0.0: a later pass of the same search (the ring has turned)
  NMPC_D double & mailNomResident() const
  {
    return lds[(static_cast<size_t>(kRecArea) + 2 + NMPC_HIP_NTRACE + 2) * LW + 3];
  }
  static constexpr unsigned kGroupLanes = kLanesPerBlock / kAlphaGroups;
#ifndef NMPC_FANOUT_FIRST_PASS
#  define NMPC_FANOUT_FIRST_PASS 1
#endif
  //! step sizes tried by the FIRST forward pass of a line search (solveMasterFanOut): all lane groups at once, or 1
  //! (A/B builds: the groups mirror the first trial and fan out only after it failed)
  static constexpr int kFanOutFirstPass = NMPC_FANOUT_FIRST_PASS ? kAlphaGroups : 1;
  //! lane group of this lane (0 when there are none); only group 0 writes trajectories to HBM
  NMPC_D static unsigned laneGroup()
  {
    return (kAlphaGroups > 1) ? waveLane() / kGroupLanes : 0u;
  }
  /** Workgroup barrier for the LDS hand-off.  Only LDS traffic has to be complete (lgkmcnt); fullBarrier() would
      also wait for vmcnt(0), i.e. drain the helper's HBM prefetches and the stores of every timestep. */
#ifdef NMPC_AMD_PROFILE_2W
  // Profiling build (NMPC_AMD_EXTRA_HIPCC_FLAGS=-DNMPC_AMD_PROFILE_2W python -m nmpc_amd.build --force, then
  // scripts/profile_2w.py): shader cycles per pass kind and the share spent waiting in the per-timestep barriers,
  // per role; written over qp_free of instances 0 (master) / 1 (helper) of tile 0.  Never enabled in product builds.
  mutable unsigned long long prof_wait = 0, prof_t0 = 0;
  mutable unsigned prof_count = 0, prof_passes[2] = {0, 0}; // stamped sections (quad kernel: NMPC_QPROF_END), passes per kind
  mutable unsigned long long prof_acc[4] = {0, 0, 0, 0}; // [2 * pass_kind + 0] = total, [+ 1] = barrier wait
  NMPC_D void wgBarrier() const
  {
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    prof_wait += __builtin_readcyclecounter() - t0;
  }
  NMPC_D void profBegin() const
  {
    prof_wait = 0;
    prof_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int kind) const
  {
    prof_acc[2 * kind] += __builtin_readcyclecounter() - prof_t0;
    prof_acc[2 * kind + 1] += prof_wait;
    prof_passes[kind]++;
  }
  NMPC_D void profFlush(int instance) const
  {
    if(blockIdx.x == 0 && lane == 0)
    {
      for(int r = 0; r < 4; r++)
      {
        buf.qp_free[static_cast<size_t>(r) * LW + instance] = static_cast<unsigned>(prof_acc[r] >> 4);
      }
      // HW_REG_HW_ID (id 4): [3:0] wave slot, [5:4] SIMD, [11:8] CU, [15:13] SE -> which SIMD each role runs on
      buf.qp_free[static_cast<size_t>(4) * LW + instance] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
      buf.qp_free[static_cast<size_t>(9) * LW + instance] = prof_count;
      buf.qp_free[static_cast<size_t>(10) * LW + instance] = prof_passes[0];
      buf.qp_free[static_cast<size_t>(11) * LW + instance] = prof_passes[1];
    }
  }
#else
  NMPC_D static void wgBarrier()
  {
    fuzzSched(3);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    fuzzSched(4);
  }
  // product builds: two accumulators per wave (backward passes, forward passes), read by the master at the end of the solve
  // (phaseFlush); the timestamps sit at pass boundaries, next to the workgroup barriers
  mutable unsigned long long phase_t0 = 0, phase_start = 0;
  mutable unsigned long long phase_acc[2] = {0, 0};
  NMPC_D void profBegin() const
  {
    phase_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int kind) const
  {
    phase_acc[kind] += __builtin_readcyclecounter() - phase_t0;
  }
  NMPC_D void profFlush(int) const {}
#endif
#ifdef NMPC_AMD_PROFILE_2W
  NMPC_D void phaseStart() const {}
  NMPC_D void phaseFlush(bool, bool = true, bool = false) const {}
#else
  NMPC_D void phaseStart() const
  {
    phase_start = __builtin_readcyclecounter();
  }
  /** took_part / resumed: under resumable launches (the ragged schedule) the ticks of ONE solve are the sum over its launches — the
      first launch writes, a later one adds, and only for the instances it actually iterated (the rows travel with their instance
      through the compaction swaps), so that nmpc_hip_ddp_last_solve_phases splits the whole solve and not its last launch. */
  NMPC_D void phaseFlush(bool valid, bool took_part = true, bool resumed = false) const
  {
    if(valid && buf.phase_ticks != nullptr && (took_part || !resumed))
    {
      unsigned long long * p = buf.phase_ticks + static_cast<size_t>(b) * 4;
      const unsigned long long whole = __builtin_readcyclecounter() - phase_start;
      p[0] = (resumed ? p[0] : 0ull) + phase_acc[0];
      p[1] = (resumed ? p[1] : 0ull) + phase_acc[1];
      p[2] = (resumed ? p[2] : 0ull) + whole;
    }
  }
#endif

  // ===================================================================================================
  // backward pass
  // ===================================================================================================
  /** Helper: derivatives of timestep i at (x, u) = (x_i, u_i) of the current trajectory -> record slot. */
  NMPC_D void produceDerivatives(int i, int slot, const StateDimVector & x, const InputDimVector & u_all, const double * lim_lo = nullptr,
                                 const double * lim_hi = nullptr) const
  {
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    InputDimVector u = u_all;
    u.resize(m);
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
#pragma unroll kU
    for(int e = 0; e < N * N; e++)
    {
      rec(slot, oFx + e) = Fx.data()[e];
      rec(slot, oLxx + e) = Lxx.data()[e];
    }
#pragma unroll kU
    for(int e = 0; e < N * MM; e++)
    {
      rec(slot, oFu + e) = Fu.data()[e];
      rec(slot, oLxu + e) = Lxu.data()[e];
    }
#pragma unroll kU
    for(int e = 0; e < N; e++)
    {
      rec(slot, oLx + e) = Lx[e];
    }
#pragma unroll kU
    for(int e = 0; e < MM; e++)
    {
      rec(slot, oLu + e) = Lu[e];
      rec(slot, oU + e) = u[e];
      if constexpr(kConstrained)
      {
        if(lim_lo != nullptr)
        {
          rec(slot, oLoRel + e) = lim_lo[e] - u[e];
          rec(slot, oUpRel + e) = lim_hi[e] - u[e];
        }
      }
    }
#pragma unroll kU
    for(int e = 0; e < MM * MM; e++)
    {
      rec(slot, oLuu + e) = Luu.data()[e];
    }
  }

  /** (x_i, u_i) in registers from its HBM request (kBwdAhead timesteps early) to its use. */
  struct Point
  {
    StateDimVector x;
    InputDimVector u;
    double lo[kConstrained ? MM : 1], hi[kConstrained ? MM : 1]; //!< input limits of the timestep (box-constrained solves)
  };
  static constexpr int kBwdAhead = 4;
  NMPC_D void loadPoint(int i, unsigned ox, unsigned ou, Point & p) const
  {
    Base::loadX(Base::xRow(i), ox, p.x);
    Base::loadU(Base::uRow(i), ou, p.u, MM);
    if constexpr(kConstrained)
    {
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        p.lo[a] = inputLimitLo(buf, b, i, a);
        p.hi[a] = inputLimitHi(buf, b, i, a);
      }
    }
  }
  /** Scheduling fence for one value: everything computed from v is issued after this point (volatile asm statements
      keep their order, so also after the preceding wgBarrier()).  Without it the compiler hoists the arithmetic of
      several unrolled timesteps in front of the first barrier: one long timestep and three short ones per four, and
      since the waves meet at a barrier every timestep, the long one sets the pace. */
  NMPC_D static void pin(double & v)
  {
    asm volatile("" : "+v"(v));
  }
  NMPC_D void backwardHelperStep(int i, unsigned ox, unsigned ou, Point & p) const
  {
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      pin(p.x[j]);
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      pin(p.u[a]);
    }
    if constexpr(kConstrained)
    {
      produceDerivatives(i, i & 1, p.x, p.u, p.lo, p.hi);
    }
    else
    {
      produceDerivatives(i, i & 1, p.x, p.u);
    }
    // unconditional (the last requests re-read timestep 0 and are never used), so that the compiler can count the
    // requests in flight instead of waiting for all of them
    loadPoint(i >= kBwdAhead ? i - kBwdAhead : 0, ox, ou, p);
    wgBarrier();
  }

  NMPC_D void backwardHelper(int sel_h) const
  {
    // barrier k separates "record of step T-1-k written" from "record of step T-1-k read".
    // (x, u) of step i - kBwdAhead are requested while the derivatives of step i are evaluated: a request takes
    // 500 .. 2000 cycles (L2 / MALL / HBM), a timestep ~900.  Ring of kBwdAhead register sets, loop unrolled by the
    // same number so that no set is ever copied.
    static_assert(kBwdAhead == 4, "backwardHelper is written for a ring of four register sets");
    const unsigned ox = Base::offX(sel_h), ou = Base::offU(sel_h);
    Point p0, p1, p2, p3;
    loadPoint(T - 1, ox, ou, p0);
    loadPoint(T > 1 ? T - 2 : 0, ox, ou, p1);
    loadPoint(T > 2 ? T - 3 : 0, ox, ou, p2);
    loadPoint(T > 3 ? T - 4 : 0, ox, ou, p3);
    int i = T - 1;
    for(; i >= 3; i -= 4)
    {
      backwardHelperStep(i, ox, ou, p0);
      backwardHelperStep(i - 1, ox, ou, p1);
      backwardHelperStep(i - 2, ox, ou, p2);
      backwardHelperStep(i - 3, ox, ou, p3);
    }
    if(i >= 0)
    {
      backwardHelperStep(i, ox, ou, p0);
    }
    if(i >= 1)
    {
      backwardHelperStep(i - 1, ox, ou, p1);
    }
    if(i >= 2)
    {
      backwardHelperStep(i - 2, ox, ou, p2);
    }
    wgBarrier(); // closes the pass: the master's last record reads are done
  }

  /** Fill v from the record unless the compiler knows it as a structural constant. */
  NMPC_D void fromRecord(double & v, int slot, int idx) const
  {
    if(!__builtin_constant_p(v))
    {
      v = rec(slot, idx);
    }
  }

  /** Master: the Riccati recursion of DDPSolver::backwardPass (DDPSolver.hpp:342-534) over the staged records.
      `need` masks the lanes this pass is run for; returns the per-lane success flag. */
  NMPC_D bool backwardMaster(bool need)
  {
    double Vx[N], Vxx[N * N];
    {
      StateDimVector xT, vx;
      StateStateDimMatrix vxx;
      Base::loadX(Base::xRow(T), Base::offX(sel), xT);
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
    double dV0_l = 0, dV1_l = 0, krn = 0;
    bool ok = true;
    double k_next[MM];
    int m_next = -1;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      k_next[a] = 0;
    }
    const unsigned ob = Base::offB();

    for(int i = T - 1; i >= 0; i--)
    {
      wgBarrier(); // record of step i is complete
      const int slot = i & 1;
      const double t = current_t + i * problem.dt();
      const int m = Base::inputDimAt(t);

      // structure of the derivatives from a dummy evaluation; values from the record (file header)
      StateDimVector xd;
      InputDimVector u;
      u.resize(m);
#pragma unroll kU
      for(int j = 0; j < N; j++)
      {
        xd[j] = rec(slot, oLx + j);
      }
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        u[a] = rec(slot, oU + a);
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
      problem.calcStateEqDeriv(t, xd, u, Fx, Fu);
      problem.calcRunningCostDeriv(t, xd, u, Lx, Lu, Lxx, Luu, Lxu);
#pragma unroll kU
      for(int e = 0; e < N * N; e++)
      {
        fromRecord(Fx.data()[e], slot, oFx + e);
        fromRecord(Lxx.data()[e], slot, oLxx + e);
      }
#pragma unroll kU
      for(int e = 0; e < N * MM; e++)
      {
        fromRecord(Fu.data()[e], slot, oFu + e);
        fromRecord(Lxu.data()[e], slot, oLxu + e);
      }
#pragma unroll kU
      for(int e = 0; e < N; e++)
      {
        fromRecord(Lx[e], slot, oLx + e);
      }
#pragma unroll kU
      for(int e = 0; e < MM; e++)
      {
        fromRecord(Lu[e], slot, oLu + e);
      }
#pragma unroll kU
      for(int e = 0; e < MM * MM; e++)
      {
        fromRecord(Luu.data()[e], slot, oLuu + e);
      }

      // ---- Q terms    DDPSolver.hpp:386-408
      double Qu[MM], Qx[N], Qux[MM * N], Quu[MM * MM], Qxx[N * N];
      double FuT_V[MM * N];
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        if(a < m)
        {
          double s = 0;
#pragma unroll kU
          for(int r = 0; r < N; r++)
          {
            Base::macc(s, Fu(r, a), Vx[r]);
          }
          Qu[a] = Base::addc(Lu[a], s);
        }
      }
#pragma unroll kU
      for(int a = 0; a < N; a++)
      {
        double s = 0;
#pragma unroll kU
        for(int r = 0; r < N; r++)
        {
          Base::macc(s, Fx(r, a), Vx[r]);
        }
        Qx[a] = Base::addc(Lx[a], s);
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
              Base::macc(s, Fu(r, a), Vxx[r + c * N]);
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
              Base::macc(s, FuT_V[a + r * MM], Fx(r, c));
            }
            Qux[a + c * MM] = Base::addc(Lxu(c, a), s);
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
              Base::macc(s, FuT_V[a + r * MM], Fu(r, bb));
            }
            Quu[a + bb * MM] = Base::addc(Luu(a, bb), s);
          }
        }
      }
      {
        double FxT_V[N * N];
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
              Base::macc(s, Fx(r, a), Vxx[r + c * N]);
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
              Base::macc(s, FxT_V[a + r * N], Fx(r, c));
            }
            Qxx[a + c * N] = Base::addc(Lxx(a, c), s);
          }
        }
      }

      // ---- regularisation    :421-441
      double Qux_reg[MM * N], Quu_F[MM * MM];
      if(cfg.reg_type == 2)
      {
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
                Base::macc(s, Fu(r, a), v);
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
                Base::macc(s, FuT_V[a + r * MM], Fx(r, c));
              }
              Qux_reg[a + c * MM] = Base::addc(Lxu(c, a), s);
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
                Base::macc(s, FuT_V[a + r * MM], Fu(r, bb));
              }
              Quu_F[a + bb * MM] = Base::addc(Luu(a, bb), s);
            }
          }
        }
      }
      else
      {
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
      bool step_ok = true;
      if(m > 0)
      {
        if constexpr(kConstrained)
        {
          double initial_k[MM], lo[MM], up[MM];
#pragma unroll kU
          for(int a = 0; a < MM; a++)
          {
            initial_k[a] = (i != T - 1 && m_next == m) ? k_next[a] : 0.0;
            lo[a] = rec(slot, oLoRel + a); // (= limit - u_i, evaluated by the helper)
            up[a] = rec(slot, oUpRel + a);
          }
          QPOut qp;
          Base::boxQP(m, Quu_F, Qu, lo, up, initial_k, qp);
          unsigned free_mask = 0;
          for(int j = 0; j < qp.n_free; j++)
          {
            free_mask |= (1u << qp.free_idx[j]);
          }
          if(need && ok)
          {
            Base::elem(buf.qp_ret, T, i) = qp.retval;
            Base::elem(buf.qp_free, T, i) = free_mask;
          }
          if(qp.retval < 0)
          {
            step_ok = false;
          }
          else
          {
#pragma unroll kU
            for(int a = 0; a < MM; a++)
            {
              k[a] = qp.x[a];
            }
            if(qp.n_free > 0)
            {
              for(int c = 0; c < N; c++)
              {
                double col[MM];
                for(int j = 0; j < qp.n_free; j++)
                {
                  col[j] = Qux_reg[qp.free_idx[j] + c * MM];
                }
                Base::template ldltSolveInPlace<MM, 1>(qp.fac, qp.inv_d, qp.n_free, col);
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
          double fac[MM * MM], inv_d[MM];
#pragma unroll kU
          for(int e = 0; e < MM * MM; e++)
          {
            fac[e] = Quu_F[e];
          }
          if(!Base::template ldltInPlace<MM>(fac, inv_d, m))
          {
            step_ok = false;
          }
          else
          {
#pragma unroll kU
            for(int a = 0; a < MM; a++)
            {
              k[a] = Qu[a];
            }
            Base::template ldltSolveInPlace<MM, 1>(fac, inv_d, m, k);
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
              Base::template ldltSolveInPlace<MM, 1>(fac, inv_d, m, &K[c * MM]);
#pragma unroll kU
              for(int a = 0; a < MM; a++)
              {
                K[a + c * MM] = -1 * K[a + c * MM];
              }
            }
          }
        }
      }
      // a lane whose factorisation failed stops updating its state (the reference returns false here, :473-480,
      // :501-508); the wave keeps running the remaining timesteps for the other lanes
      const bool live = need && ok && step_ok;
      ok = ok && step_ok;

      // ---- cost-to-go update    :522-526
      {
        double kQu = 0, kQuuk = 0;
        double Quu_k[MM];
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            Base::macc(kQu, k[a], Qu[a]);
            double s = 0;
#pragma unroll kU
            for(int bb = 0; bb < MM; bb++)
            {
              if(bb < m)
              {
                Base::macc(s, Quu[a + bb * MM], k[bb]);
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
            Base::macc(kQuuk, k[a], Quu_k[a]);
          }
        }
        if(live)
        {
          dV0_l += kQu;
          dV1_l += 0.5 * kQuuk;
        }
      }
      double KtQuu[N * MM];
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
              Base::macc(s, K[p + r * MM], Quu[p + a * MM]);
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
            Base::macc(s1, KtQuu[r + a * N], k[a]);
            Base::macc(s2, K[a + r * MM], Qu[a]);
            Base::macc(s3, Qux[a + r * MM], k[a]);
          }
        }
        const double v = ((Qx[r] + s1) + s2) + s3;
        Vx[r] = v; // pass-local: lanes that are not live never commit anything derived from it
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
              Base::macc(s1, KtQuu[r + a * N], K[a + c * MM]);
              Base::macc(s2, K[a + r * MM], Qux[a + c * MM]);
              Base::macc(s3, Qux[a + r * MM], K[a + c * MM]);
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

      // ---- save gains    :529-530, running max of |k_i| / (|u_i| + 1)    :217-221
      if(live)
      {
        double * kp = Base::kRow(i);
        double * Kp = Base::KRow(i);
        double kn = 0, un = 0;
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          Base::st(kp + a * LW, ob, (a < m) ? k[a] : 0.0);
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
          Base::st(Kp + e * LW, ob, ((e % MM) < m) ? K[e] : 0.0);
        }
        m_next = m;
        const double knorm = (M == 1) ? fabs(k[0]) : sqrt(kn);
        const double unorm = (M == 1) ? fabs(u[0]) : sqrt(un);
        krn = fmax(krn, knorm * recipFast(unorm + 1.0));
      }
    }
    wgBarrier(); // closes the pass
    if(need)
    {
      dV0 = dV0_l;
      dV1 = dV1_l;
      k_rel_norm = krn;
    }
    return ok;
  }

  // ===================================================================================================
  // forward pass    DDPSolver.hpp:536-560
  // ===================================================================================================
  /** Nominal (x_i, u_i, k_i, K_i) of one timestep, in registers from its HBM load (two timesteps early) to its use. */
  struct Nominal
  {
    double x[N], u[MM], k[MM], K[MM * N];
  };
  NMPC_D void loadNominal(int i, int sel_h, Nominal & n) const
  {
    const unsigned ox = Base::offX(sel_h), ou = Base::offU(sel_h), ob = Base::offB();
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      n.x[j] = Base::ld(Base::xRow(i) + j * LW, ox);
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      n.u[a] = Base::ld(Base::uRow(i) + a * LW, ou);
      n.k[a] = Base::ld(Base::kRow(i) + a * LW, ob);
    }
#pragma unroll kU
    for(int e = 0; e < MM * N; e++)
    {
      n.K[e] = Base::ld(Base::KRow(i) + e * LW, ob);
    }
  }
  /** Helper side of one forward pass: evaluates the cost of the step the master has just finished, stores the
      candidate trajectory (the helper issues ONLY stores during this pass, the master ONLY loads: a wave with both
      kinds in flight has to drain them all at every wait), returns the candidate total cost through
      LDS. */
  /** Where the helper's lanes store the rollout of a fan-out pass: group 0 into the candidate half of X / U / cost, group
      g > 0 into slot g - 1 of the handle's fan-out scratch [tile][kAlphaGroups - 1][rows of X, U, cost][64] (buf.wpi_ws:
      shapes of the lane-group kernels have no other use for it), from where adoptFanOut() copies the trajectory of an
      accepted step size into the candidate half.  Without the scratch (allocation failed) only group 0 stores and an
      accepted step size of another group is rolled out once more. */
  struct FanDest
  {
    char * x;
    char * u;
    char * c;
    bool on;
  };
  NMPC_D bool fanScratch() const
  {
    return kAlphaGroups > 1 && buf.wpi_ws != nullptr;
  }
  NMPC_D size_t fanRows() const
  {
    return Base::rowsX() + Base::rowsU() + static_cast<size_t>(T + 1);
  }
  //! rows of X, U, cost of scratch slot g - 1 (g >= 1) for this lane's instance
  NMPC_D FanDest fanSlot(unsigned g) const
  {
    FanDest d;
    char * f = reinterpret_cast<char *>(Base::tileBase(buf.wpi_ws, fanRows(), kAlphaGroups - 1))
               + (static_cast<size_t>(g - 1) * fanRows() * LW + Base::lane) * sizeof(double);
    d.x = f;
    d.u = f + Base::rowsX() * LW * sizeof(double);
    d.c = d.u + Base::rowsU() * LW * sizeof(double);
    d.on = true;
    return d;
  }
  NMPC_D FanDest fanDest(unsigned cx, unsigned cu, unsigned cc) const
  {
    const unsigned g = laneGroup();
    // (Measured and not kept, round 5: the groups behind the first rolling out for the cost only in the first pass of a search the
    // workgroup expects to end at the first step size — 0.47 GB less HBM write traffic per launch, but the 0.2 % of searches that
    // then take a later step size cost their workgroup a whole extra pass instead of a 2 k-cycle copy: c2 0.451 -> 0.480 ms.)
    if(g > 0 && fanScratch())
    {
      return fanSlot(g);
    }
    FanDest d;
    d.x = reinterpret_cast<char *>(Base::Xt) + cx;
    d.u = reinterpret_cast<char *>(Base::Ut) + cu;
    d.c = reinterpret_cast<char *>(Base::Ct) + cc;
    d.on = (g == 0);
    return d;
  }
  NMPC_D static void stAt(char * base, size_t row, double v)
  {
    *reinterpret_cast<double *>(base + row * (LW * sizeof(double))) = v;
  }
  NMPC_D static double ldAt(const char * base, size_t row)
  {
    return *reinterpret_cast<const double *>(base + row * (LW * sizeof(double)));
  }
  /** Every wave of the workgroup, after a fan-out pass in which instances accepted the step size of a group g > 0 (the
      master left g in the instance's cost mailbox, 0 otherwise): the trajectory of slot g - 1 of the scratch becomes the
      candidate (rows are shared out over the workgroup's threads: thread = (instance, part)).  The pass boundaries on
      either side are full barriers (post()): the helper's stores have landed, these land before the next pass reads; barrier A
      inside keeps the mailbox words of this pass from being rewritten before every wave has read them. */
  NMPC_D void adoptFanOut(int sel_lane) const
  {
    const unsigned inst = waveLane() % kGroupLanes;
    const unsigned part = threadIdx.x / kGroupLanes, parts = blockDim.x / kGroupLanes;
    const int g = static_cast<int>(mailCostAt(inst));
#ifndef NMPC_AMD_AB_REOPEN_FAN_ADOPT_RACE // (A/B builds: the race below open again — scripts/fuzz_diff.py shows it within one run)
    // Barrier A: every wave has read this pass's command word (followerLoop) and its g.  The pass has no other barrier, and
    // without this one the master — done with its share of the copy — posts the NEXT pass's command into the same LDS words while
    // a wave that was slow out of barrier P has yet to read this one's: that wave skips the copy and runs the next pass twice, one
    // barrier out of step with the workgroup from there on (costs read from the mailbox before they are written).  The copy is
    // ~2 k cycles long, so it takes a wave that late: never seen in 2000-repetition soaks of the product build, found by the
    // wave-timing fuzz build (fuzz_sched.hpp) within one solve — profiles/r05_fuzz_fan_adopt_race.txt.
    wgBarrier();
#endif
    if(g > 0)
    {
      const FanDest src = fanSlot(static_cast<unsigned>(g));
      const int cs = 1 - sel_lane;
      char * dx = reinterpret_cast<char *>(Base::Xt) + Base::offX(cs);
      char * du = reinterpret_cast<char *>(Base::Ut) + Base::offU(cs);
      char * dc = reinterpret_cast<char *>(Base::Ct) + Base::offC(cs);
      const size_t nx = Base::rowsX(), nu = Base::rowsU(), nc = static_cast<size_t>(T + 1);
      copyRows(dx, src.x, nx, part, parts);
      copyRows(du, src.u, nu, part, parts);
      copyRows(dc, src.c, nc, part, parts);
    }
  }
  /** Rows part, part + parts, ... of src to dst, eight loads requested before the first of them is stored: a store may alias the
      next load as far as the compiler knows, and written as `st(r, ld(r))` (also under `#pragma unroll`) the copy was a round trip
      to L2 per row — 39 in a row for the headline shape [disassembly: load, s_waitcnt vmcnt(0), store, branch; measured: forward
      passes of the nominal workload 0.235 -> 0.218 ms per solve].  Not inlined: as part of the kernel's body the same loop cost
      the passes around it registers (headline + 2.5 % instead of + 4.5 %, bipedal - 1 % instead of + 1.5 %). */
  __device__ __attribute__((noinline)) static void copyRows(char * dst, const char * src, size_t n, size_t part, size_t parts)
  {
    constexpr int kBatch = 8; // (4 and 16 measured the same on the headline workload)
    for(size_t r0 = part; r0 < n; r0 += kBatch * parts)
    {
      double v[kBatch];
      NMPC_UNROLL
      for(int k = 0; k < kBatch; k++)
      {
        const size_t r = r0 + k * parts;
        v[k] = ldAt(src, r < n ? r : n - 1);
      }
      NMPC_UNROLL
      for(int k = 0; k < kBatch; k++)
      {
        const size_t r = r0 + k * parts;
        if(r < n)
        {
          stAt(dst, r, v[k]);
        }
      }
    }
  }
  /** \tparam kFanOut the lane groups roll out DIFFERENT step sizes (line search after a failed first trial): group 0
      writes its trajectory into the candidate half, the others into the fan-out scratch (fanDest()); otherwise the groups
      are mirrors and all of them write (the same values) */
  template<bool kFanOut = false>
  NMPC_D void forwardHelper(int sel_h, bool initial = false) const
  {
    // forward pass: the candidate half; initial rollout (rolloutMaster): the trajectory half itself
    const int cs = initial ? sel_h : 1 - sel_h;
    const unsigned cx = Base::offX(cs), cu = Base::offU(cs), cc = Base::offC(cs);
    FanDest fd = {nullptr, nullptr, nullptr, false};
    if constexpr(kFanOut)
    {
      fd = fanDest(cx, cu, cc);
    }
    double J = 0;
    const int n_full = T / kFwdGroup;
    if constexpr(kLdsNominal)
    {
      if(!initial)
      {
        wgBarrier(); // barrier S of forwardMasterLds / forwardPrefetch
      }
    }
    for(int g = 0; g < n_full; g++)
    {
      wgBarrier(); // barrier g: the master wrote the records of timesteps 4 g .. 4 g + 3
#pragma unroll
      for(int r = 0; r < kFwdGroup; r++)
      {
        const int i = g * kFwdGroup + r;
        if(initial)
        {
          Base::elem(buf.input_dim, T, i) = Base::inputDimAt(current_t + i * problem.dt());
        }
        J += consumeStep<kFanOut>(i, cx, cu, cc, fd);
      }
    }
    wgBarrier(); // barrier E: the remaining timesteps and x'_T (slot 2 * kFwdGroup) are available
    for(int i = n_full * kFwdGroup; i < T; i++)
    {
      if(initial)
      {
        Base::elem(buf.input_dim, T, i) = Base::inputDimAt(current_t + i * problem.dt());
      }
      J += consumeStep<kFanOut>(i, cx, cu, cc, fd);
    }
    {
      StateDimVector xT;
#pragma unroll kU
      for(int j = 0; j < N; j++)
      {
        xT[j] = frec(2 * kFwdGroup, oXc + j);
      }
      const double cT = problem.terminalCost(current_t + T * problem.dt(), xT);
      if constexpr(kFanOut)
      {
        if(fd.on)
        {
#pragma unroll kU
          for(int j = 0; j < N; j++)
          {
            stAt(fd.x, static_cast<size_t>(T) * N + j, xT[j]);
          }
          stAt(fd.c, static_cast<size_t>(T), cT);
        }
      }
      else
      {
        Base::storeX(Base::xRow(T), cx, xT);
        Base::st(Base::costRow(T), cc, cT);
      }
      J += cT;
    }
    mailCost() = J;
    wgBarrier(); // barrier F: candidate cost published
  }

  /** Helper: cost + stores of step i from the "out" record the master wrote. */
  template<bool kFanOut>
  NMPC_D double consumeStep(int i, unsigned cx, unsigned cu, unsigned cc, const FanDest & fd) const
  {
    const int slot = i % (2 * kFwdGroup);
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    StateDimVector x;
    InputDimVector u;
    u.resize(m);
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      x[j] = frec(slot, oXc + j);
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      u[a] = frec(slot, oUc + a);
    }
    const double c = problem.runningCost(t, x, u);
    if constexpr(kFanOut)
    {
      if(fd.on)
      {
#pragma unroll kU
        for(int j = 0; j < N; j++)
        {
          stAt(fd.x, static_cast<size_t>(i) * N + j, x[j]);
        }
#pragma unroll kU
        for(int a = 0; a < MM; a++)
        {
          stAt(fd.u, static_cast<size_t>(i) * MM + a, (a < m) ? u[a] : 0.0);
        }
        stAt(fd.c, static_cast<size_t>(i), c);
      }
    }
    else
    {
      Base::storeX(Base::xRow(i), cx, x);
      Base::storeU(Base::uRow(i), cu, u, m);
      Base::st(Base::costRow(i), cc, c);
    }
    return c;
  }

  /** Master side: u' = u + alpha k + K (x' - x), x'' = stateEq(x', u').  The nominal (x, u, k, K) of timestep
      i + kFwdAhead is requested while timestep i is computed: a request takes 500 .. 1500 cycles (L2 / MALL / HBM), a
      timestep ~500.  The candidate (x', u') goes to the helper through LDS. */
  static constexpr int kFwdAhead = 4;
  template<bool kPinInputs = true>
  NMPC_D void forwardStep(int i, double alpha, Nominal & nom, StateDimVector & xc) const
  {
    const int slot = i % (2 * kFwdGroup);
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    if constexpr(kPinInputs)
    {
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      pin(nom.x[j]); // keep the arithmetic of this timestep behind the previous barrier, see pin()
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      pin(nom.u[a]);
      pin(nom.k[a]);
    }
#pragma unroll kU
    for(int e = 0; e < MM * N; e++)
    {
      pin(nom.K[e]);
    }
    }
    InputDimVector uc;
    uc.resize(m);
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      if(a < m)
      {
        double s = 0;
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
          s += nom.K[a + c * MM] * (xc[c] - nom.x[c]);
        }
        uc[a] = (nom.u[a] + alpha * nom.k[a]) + s;
      }
      else
      {
        uc[a] = 0;
      }
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      pin(uc[a]); // u' is complete: the register set of `nom` is dead and the requests below can land in it (no copies)
    }
    // unconditional (the last two requests re-read timestep T-1 and are never used): with a branch around the loads
    // the compiler cannot count the requests in flight and falls back to waiting for all of them at every timestep
    loadNominal(i + kFwdAhead < T ? i + kFwdAhead : T - 1, sel, nom);
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(slot, oXc + j) = xc[j];
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      frec(slot, oUc + a) = uc[a];
    }
    xc = problem.stateEq(t, xc, uc);
  }

  /** Initial rollout (DDPSolver.hpp:83-95) in the master / helper split of the forward pass: the master keeps the
      chain x_{i+1} = stateEq(x_i, u_i) and prefetches u (ring of kFwdAhead), the helper evaluates the costs and does
      all stores (a lone wave doing both waits for every load of u behind its own stores: ~1000 cycles per timestep). */
  NMPC_D void rolloutStep(int i, unsigned ou, InputDimVector & u_ring, StateDimVector & x) const
  {
    const int slot = i % (2 * kFwdGroup);
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    InputDimVector u;
    u.resize(m);
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      pin(u_ring[a]);
      u[a] = (a < m) ? u_ring[a] : 0.0;
    }
    Base::loadU(Base::uRow(i + kFwdAhead < T ? i + kFwdAhead : T - 1), ou, u_ring, MM);
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(slot, oXc + j) = x[j];
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      frec(slot, oUc + a) = u[a];
    }
    x = problem.stateEq(t, x, u);
  }

  NMPC_D void rolloutMaster()
  {
    static_assert(kFwdAhead == 4, "rolloutMaster is written for a ring of four register sets");
    const unsigned ou = Base::offU(sel);
    StateDimVector x;
    Base::loadX(Base::tileBase(buf.x0, N), Base::offB(), x);
    InputDimVector u0, u1, u2, u3;
    Base::loadU(Base::uRow(0), ou, u0, MM);
    Base::loadU(Base::uRow(T > 1 ? 1 : T - 1), ou, u1, MM);
    Base::loadU(Base::uRow(T > 2 ? 2 : T - 1), ou, u2, MM);
    Base::loadU(Base::uRow(T > 3 ? 3 : T - 1), ou, u3, MM);
    int i = 0;
    for(; i + 3 < T; i += 4)
    {
      rolloutStep(i, ou, u0, x);
      rolloutStep(i + 1, ou, u1, x);
      rolloutStep(i + 2, ou, u2, x);
      rolloutStep(i + 3, ou, u3, x);
      wgBarrier(); // barrier of this group of kFwdGroup timesteps
    }
    if(i < T)
    {
      rolloutStep(i, ou, u0, x);
    }
    if(i + 1 < T)
    {
      rolloutStep(i + 1, ou, u1, x);
    }
    if(i + 2 < T)
    {
      rolloutStep(i + 2, ou, u2, x);
    }
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(2 * kFwdGroup, oXc + j) = x[j];
    }
    wgBarrier(); // barrier E
    wgBarrier(); // barrier F
    J_cur = mailCost();
  }

  // ---- kLdsNominal: the nominal comes through LDS --------------------------------------------------------------------
  NMPC_D void readNominalLds(int slot, int r, Nominal & n) const
  {
    typedef double Pair2 __attribute__((ext_vector_type(2)));
    const Pair2 * rec = reinterpret_cast<const Pair2 *>(nomRec(slot, r, waveLane() % kNomInst));
    double v[kNomRec];
#pragma unroll
    for(int p = 0; p < kNomRec / 2; p++)
    {
      const Pair2 w = rec[p];
      v[2 * p] = w[0];
      v[2 * p + 1] = w[1];
    }
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      n.x[j] = v[j];
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      n.u[a] = v[N + a];
      n.k[a] = v[N + MM + a];
    }
#pragma unroll kU
    for(int e = 0; e < MM * N; e++)
    {
      n.K[e] = v[N + 2 * MM + e];
    }
  }
  /** forwardStep on a nominal that is already in registers (no request of its own). */
  NMPC_D void forwardStepLds(int i, double alpha, const Nominal & nom, StateDimVector & xc) const
  {
    const int slot = i % (2 * kFwdGroup);
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    InputDimVector uc;
    uc.resize(m);
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      if(a < m)
      {
        double s = 0;
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
          s += nom.K[a + c * MM] * (xc[c] - nom.x[c]);
        }
        uc[a] = (nom.u[a] + alpha * nom.k[a]) + s;
      }
      else
      {
        uc[a] = 0;
      }
    }
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(slot, oXc + j) = xc[j];
    }
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      frec(slot, oUc + a) = uc[a];
    }
    xc = problem.stateEq(t, xc, uc);
  }
  /** The wave that keeps the nominal records ahead of the master (wave 2 of the quad kernel): lane = (timestep of the
      group, instance), so one request fetches an entry of the record for the four timesteps of a group at once —
      kNomRec requests per group instead of the master's 4 kNomRec.  Group g + 2 is written while the master works on
      group g (its slot was read during group g - 1) and published by the barrier that ends group g; the requests for
      group g + 3 then have a whole group of timesteps to arrive. */
  NMPC_D void forwardPrefetch(int sel_h) const
  {
    typedef double Pair2 __attribute__((ext_vector_type(2)));
    const unsigned inst = waveLane() % kNomInst;
    const int q = static_cast<int>(waveLane() / kNomInst); // timestep within the group
    const unsigned ox = Base::offX(sel_h), ou = Base::offU(sel_h), ob = Base::offB();
    auto loadGroup = [&](int g, double (&v)[kNomRec])
    {
      const int ii = g * kFwdGroup + q;
      const int i = ii < T ? ii : T - 1; // (beyond the horizon: re-reads the last timestep, never used)
#pragma unroll kU
      for(int j = 0; j < N; j++)
      {
        v[j] = Base::ld(Base::xRow(i) + j * LW, ox);
      }
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        v[N + a] = Base::ld(Base::uRow(i) + a * LW, ou);
        v[N + MM + a] = Base::ld(Base::kRow(i) + a * LW, ob);
      }
#pragma unroll kU
      for(int e = 0; e < MM * N; e++)
      {
        v[N + 2 * MM + e] = Base::ld(Base::KRow(i) + e * LW, ob);
      }
      if constexpr(kNomRec > N + 2 * MM + MM * N)
      {
        v[kNomRec - 1] = 0;
      }
    };
    auto writeGroup = [&](int slot, const double (&v)[kNomRec])
    {
      Pair2 * rec = reinterpret_cast<Pair2 *>(nomRec(slot, q, inst));
#pragma unroll
      for(int p = 0; p < kNomRec / 2; p++)
      {
        Pair2 w;
        w[0] = v[2 * p];
        w[1] = v[2 * p + 1];
        rec[p] = w;
      }
    };
    double v0[kNomRec], v1[kNomRec], v2[kNomRec];
    if(kNominalTail && mailNomResident() != 0.0)
    {
      loadGroup(2, v2); // (written again in the loop's first trip: the same values)
    }
    else
    {
      loadGroup(0, v0);
      loadGroup(1, v1);
      loadGroup(2, v2);
      writeGroup(0, v0);
      writeGroup(1, v1);
    }
    wgBarrier(); // barrier S: groups 0 and 1 are in LDS
    const int n_full = T / kFwdGroup;
    int slot2 = 2; // slot of group g + 2
    for(int g = 0; g < n_full; g++)
    {
      writeGroup(slot2, v2);
      loadGroup(g + 3, v2);
      wgBarrier(); // barrier of this group of kFwdGroup timesteps
      slot2 = (slot2 == kNomSlots - 1) ? 0 : slot2 + 1;
    }
    wgBarrier(); // barrier E
    wgBarrier(); // barrier F
  }
  NMPC_D void forwardMasterLds(double alpha)
  {
    static_assert(kFwdGroup == 4, "forwardMasterLds is written for groups of four timesteps");
    wgBarrier(); // barrier S: the records of groups 0 and 1 are in LDS
    Nominal na, nb;
    int slot = 0; // slot of the group the master works on
    readNominalLds(slot, 0, na);
    StateDimVector xc;
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      xc[j] = na.x[j]; // x'_0 = x_0
    }
    const int n_full = T / kFwdGroup;
    for(int g = 0; g < n_full; g++)
    {
      const int slot1 = (slot == kNomSlots - 1) ? 0 : slot + 1;
      // a record is requested one timestep before its use; the first record of the next group was published by the
      // barrier before last
      readNominalLds(slot, 1, nb);
      forwardStepLds(g * kFwdGroup, alpha, na, xc);
      readNominalLds(slot, 2, na);
      forwardStepLds(g * kFwdGroup + 1, alpha, nb, xc);
      readNominalLds(slot, 3, nb);
      forwardStepLds(g * kFwdGroup + 2, alpha, na, xc);
      readNominalLds(slot1, 0, na);
      forwardStepLds(g * kFwdGroup + 3, alpha, nb, xc);
      wgBarrier(); // barrier of this group of kFwdGroup timesteps
      slot = slot1;
    }
    const int i = n_full * kFwdGroup;
    if(i < T)
    {
      readNominalLds(slot, 1, nb);
      forwardStepLds(i, alpha, na, xc);
    }
    if(i + 1 < T)
    {
      readNominalLds(slot, 2, na);
      forwardStepLds(i + 1, alpha, nb, xc);
    }
    if(i + 2 < T)
    {
      forwardStepLds(i + 2, alpha, na, xc);
    }
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(2 * kFwdGroup, oXc + j) = xc[j];
    }
    wgBarrier(); // barrier E
    wgBarrier(); // barrier F
    J_cand = mailCost();
  }

  /** One timestep of a cost-only rollout: forwardStepLds + consumeStep without the hand-off and without stores — the same
      expressions on the same values, so the cost is the one the master / helper pair would have summed for this step size. */
  NMPC_D void costOnlyStepLds(int i, double alpha, const Nominal & nom, StateDimVector & xc, double & J) const
  {
    const double t = current_t + i * problem.dt();
    const int m = Base::inputDimAt(t);
    InputDimVector uc;
    uc.resize(m);
#pragma unroll kU
    for(int a = 0; a < MM; a++)
    {
      if(a < m)
      {
        double s = 0;
#pragma unroll kU
        for(int c = 0; c < N; c++)
        {
          s += nom.K[a + c * MM] * (xc[c] - nom.x[c]);
        }
        uc[a] = (nom.u[a] + alpha * nom.k[a]) + s;
      }
      else
      {
        uc[a] = 0;
      }
    }
    J += problem.runningCost(t, xc, uc);
    xc = problem.stateEq(t, xc, uc);
  }
  /** An EXTRA MASTER of a fan-out pass (kCmdForwardFanOut): a wave that is neither master, helper nor — unless kPrefetch —
      prefetcher rolls out kAlphaGroups more step sizes of alpha_list, one per lane group, from the same LDS nominal records as
      the master, and sums their cost (DDPSolver.hpp:536-560 without the stores: a trial that is not taken is only its cost,
      :247-250).  Extra master `which` (0, 1) takes the step sizes first + kAlphaGroups (which + 1) + lane group.  A step size
      accepted from here has no stored rollout: the master rolls it out once more (solveMasterFanOut).  kPrefetch: this wave
      is also the one that keeps the nominal records ahead of everybody (forwardPrefetch's loads and LDS writes, in between its
      own timesteps).  Same barriers as forwardMasterLds / forwardHelper / forwardPrefetch. */
  template<bool kPrefetch>
  NMPC_D void forwardCostOnlyLds(int sel_h, int which) const
  {
    static_assert(kFwdGroup == 4, "written for groups of four timesteps, as forwardMasterLds");
    typedef double Pair2 __attribute__((ext_vector_type(2)));
    // ---- prefetch duty (forwardPrefetch): lane = (timestep of the group, instance)
    const unsigned p_inst = waveLane() % kNomInst;
    const int p_q = static_cast<int>(waveLane() / kNomInst);
    const unsigned ox = Base::offX(sel_h), ou = Base::offU(sel_h), ob = Base::offB();
    auto loadGroup = [&](int g, double (&v)[kNomRec])
    {
      const int ii = g * kFwdGroup + p_q;
      const int i = ii < T ? ii : T - 1;
#pragma unroll kU
      for(int j = 0; j < N; j++)
      {
        v[j] = Base::ld(Base::xRow(i) + j * LW, ox);
      }
#pragma unroll kU
      for(int a = 0; a < MM; a++)
      {
        v[N + a] = Base::ld(Base::uRow(i) + a * LW, ou);
        v[N + MM + a] = Base::ld(Base::kRow(i) + a * LW, ob);
      }
#pragma unroll kU
      for(int e = 0; e < MM * N; e++)
      {
        v[N + 2 * MM + e] = Base::ld(Base::KRow(i) + e * LW, ob);
      }
      if constexpr(kNomRec > N + 2 * MM + MM * N)
      {
        v[kNomRec - 1] = 0;
      }
    };
    auto writeGroup = [&](int slot, const double (&v)[kNomRec])
    {
      Pair2 * rec = reinterpret_cast<Pair2 *>(nomRec(slot, p_q, p_inst));
#pragma unroll
      for(int p = 0; p < kNomRec / 2; p++)
      {
        Pair2 w;
        w[0] = v[2 * p];
        w[1] = v[2 * p + 1];
        rec[p] = w;
      }
    };
    double v2[kNomRec];
    if constexpr(kPrefetch)
    {
      double v0[kNomRec], v1[kNomRec];
      if(kNominalTail && mailNomResident() != 0.0)
      {
        loadGroup(2, v2);
      }
      else
      {
        loadGroup(0, v0);
        loadGroup(1, v1);
        loadGroup(2, v2);
        writeGroup(0, v0);
        writeGroup(1, v1);
      }
    }
    // ---- this lane group's step size (uniformly indexed reads of the list, selected per lane group)
    const int first = static_cast<int>(mailFirstAlpha()) + kAlphaGroups * (which + 1);
    const int last_ai = cfg.n_alpha - 1;
    double alpha = cfg.alpha_list[first < last_ai ? first : last_ai];
#pragma unroll
    for(int gg = 1; gg < kAlphaGroups; gg++)
    {
      const double a = cfg.alpha_list[first + gg < last_ai ? first + gg : last_ai];
      alpha = (laneGroup() == static_cast<unsigned>(gg)) ? a : alpha;
    }
    wgBarrier(); // barrier S: the records of groups 0 and 1 are in LDS
    Nominal na, nb;
    int slot = 0, slot2 = 2;
    readNominalLds(slot, 0, na);
    StateDimVector xc;
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      xc[j] = na.x[j]; // x'_0 = x_0
    }
    double J = 0;
    const int n_full = T / kFwdGroup;
    for(int g = 0; g < n_full; g++)
    {
      const int slot1 = (slot == kNomSlots - 1) ? 0 : slot + 1;
      if constexpr(kPrefetch)
      {
        writeGroup(slot2, v2);
        loadGroup(g + 3, v2);
        slot2 = (slot2 == kNomSlots - 1) ? 0 : slot2 + 1;
      }
      readNominalLds(slot, 1, nb);
      costOnlyStepLds(g * kFwdGroup, alpha, na, xc, J);
      readNominalLds(slot, 2, na);
      costOnlyStepLds(g * kFwdGroup + 1, alpha, nb, xc, J);
      readNominalLds(slot, 3, nb);
      costOnlyStepLds(g * kFwdGroup + 2, alpha, na, xc, J);
      readNominalLds(slot1, 0, na);
      costOnlyStepLds(g * kFwdGroup + 3, alpha, nb, xc, J);
      wgBarrier(); // barrier of this group of kFwdGroup timesteps
      slot = slot1;
    }
    const int i = n_full * kFwdGroup;
    if(i < T)
    {
      readNominalLds(slot, 1, nb);
      costOnlyStepLds(i, alpha, na, xc, J);
    }
    if(i + 1 < T)
    {
      readNominalLds(slot, 2, na);
      costOnlyStepLds(i + 1, alpha, nb, xc, J);
    }
    if(i + 2 < T)
    {
      costOnlyStepLds(i + 2, alpha, na, xc, J);
    }
    J += problem.terminalCost(current_t + T * problem.dt(), xc);
    wgBarrier(); // barrier E
    mailCostExtra(which, waveLane()) = J;
    wgBarrier(); // barrier F: candidate costs published
  }

  NMPC_D void forwardMaster(double alpha)
  {
    if constexpr(kLdsNominal)
    {
      forwardMasterLds(alpha);
      return;
    }
    // kFwdAhead register sets form a ring: set r holds timestep i with i % kFwdAhead == r, the loop is unrolled by
    // kFwdAhead so that every set keeps its registers (no copies) and the compiler can count the requests in flight
    Nominal n0, n1, n2, n3;
    static_assert(kFwdAhead == 4 && kFwdGroup == 4, "forwardMaster is written for a ring of four register sets");
    loadNominal(0, sel, n0);
    loadNominal(T > 1 ? 1 : T - 1, sel, n1);
    loadNominal(T > 2 ? 2 : T - 1, sel, n2);
    loadNominal(T > 3 ? 3 : T - 1, sel, n3);
    StateDimVector xc;
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      xc[j] = n0.x[j]; // x'_0 = x_0
    }
    int i = 0;
    for(; i + 3 < T; i += 4)
    {
      forwardStep(i, alpha, n0, xc);
      forwardStep<false>(i + 1, alpha, n1, xc);
      forwardStep<false>(i + 2, alpha, n2, xc);
      forwardStep<false>(i + 3, alpha, n3, xc);
      wgBarrier(); // barrier of this group of kFwdGroup timesteps
    }
    if(i < T)
    {
      forwardStep(i, alpha, n0, xc);
    }
    if(i + 1 < T)
    {
      forwardStep(i + 1, alpha, n1, xc);
    }
    if(i + 2 < T)
    {
      forwardStep(i + 2, alpha, n2, xc);
    }
#pragma unroll kU
    for(int j = 0; j < N; j++)
    {
      frec(2 * kFwdGroup, oXc + j) = xc[j];
    }
    wgBarrier(); // barrier E
    wgBarrier(); // barrier F
    J_cand = mailCost();
  }

  // ===================================================================================================
  // pass protocol: the master posts {command, per-lane sel} and both waves enter the pass
  // ===================================================================================================
  enum Command
  {
    kCmdExit = 0,
    kCmdBackward = 1,
    kCmdForward = 2,
    kCmdRollout = 3,
    kCmdForwardFanOut = 4, //!< forward pass in which every lane group rolls out its own step size
    kCmdAdoptFanOut = 5 //!< adoptFanOut(): lane-group kernels only (ddp_kernels_quad.hpp)
  };

  NMPC_D void post(int cmd) const
  {
    // flags word per lane: command in the integer part (uniform over the wave), sel in the fraction
    mailFlags() = static_cast<double>(cmd * 2 + sel);
    // barrier P: command visible.  Pass boundaries use the full barrier (vmcnt(0) too): what one wave stored to HBM
    // in the previous pass (gains k, K; the candidate trajectory) is loaded by the other wave in the next one.
    fullBarrier();
  }

  NMPC_D void helperLoop() const
  {
    for(;;)
    {
      fullBarrier(); // barrier P (full: see post())
      const int word = static_cast<int>(mailFlags());
      const int cmd = __builtin_amdgcn_readfirstlane(word >> 1);
      const int sel_h = word & 1;
      if(cmd == kCmdExit)
      {
        profFlush(1);
        return;
      }
      profBegin();
      if(cmd == kCmdBackward)
      {
        backwardHelper(sel_h);
        profEnd(0);
      }
      else if(cmd == kCmdForwardFanOut)
      {
        forwardHelper<true>(sel_h);
        profEnd(1);
      }
      else
      {
        forwardHelper(sel_h, cmd == kCmdRollout);
        profEnd(1);
      }
    }
  }

  // ===================================================================================================
  // master: solve = setup + optimisation loop (DDPSolver.hpp:26-141, procOnce :143-340), written with wave-uniform
  // pass invocations (every lane of the wave enters every pass; `need_*` masks select whose state it updates)
  // ===================================================================================================
  template<bool kResumable = false>
  NMPC_D void solveMaster(bool valid)
  {
    solveMasterWith<kResumable>(valid,
                    [this](bool need)
                    {
                      post(kCmdBackward);
                      profBegin();
                      const bool ok = backwardMaster(need);
                      profEnd(0);
                      return ok;
                    });
  }

  /** Resumable launches (kResumable instantiations, DeviceBuffers::iter_end > 0): the solver state an iteration hands to the
      next — lambda, dlambda, the current cost, whether the instance still iterates; `sel` has its own array — is parked in
      buf.resume when the launch's last iteration is done and taken from there by the next launch instead of the initial rollout.
      Everything else an iteration needs it recomputes (Step 1 re-linearises, DDPSolver.hpp:157-185), so the iterations of an
      instance are the same instruction stream on the same values whether one launch runs them or several. */
  NMPC_D double & resumeWord(int row) const
  {
    return Base::elem(buf.resume, kResumeRows, row);
  }
  /** \return whether this lane's instance iterates in this launch */
  NMPC_D bool resumeState(bool valid)
  {
    // EVERY lane takes its position's `sel`, also the ones beyond the dense prefix: they hold finished instances, the wave's passes
    // run over all 64 lanes, and what a pass writes for a lane that does not iterate goes to the half 1 - sel — which has to be the
    // half the finished instance's result is NOT in (as it is in a whole-solve launch, where a finished lane keeps its sel).
    sel = buf.sel[b];
    bool running = false;
    if(valid)
    {
      lambda = resumeWord(0);
      dlambda = resumeWord(1);
      J_cur = resumeWord(2);
      running = resumeWord(3) != 0.0;
    }
    return running;
  }
  NMPC_D void parkState(bool still_running, int iterations_done = 0) const
  {
    resumeWord(0) = lambda;
    resumeWord(1) = dlambda;
    resumeWord(2) = J_cur;
    resumeWord(3) = still_running ? 1.0 : 0.0;
    resumeWord(4) = static_cast<double>(iterations_done); // (streamed solves: the instance's own iteration count so far)
  }
  //! streamed solves, stream_mode 2: this workgroup's slots were filled just now (they lie at or behind *first_active) — it starts
  //! with the initial rollout instead of resuming (set by the kernel: the fresh region starts on a workgroup boundary)
  bool stream_fresh_wg = false;
  /** Streamed solves (DeviceBuffers::stream_mode): 0 none, 1 initial rollout of freshly filled slots only, 2 a round — the
      rollout for the workgroups filled just now, then at most iter_end iterations for everyone. */
  template<bool kResumable>
  NMPC_D int streamMode() const
  {
    return kResumable ? buf.stream_mode : 0;
  }

  /** \param runBackward (need) -> ok: one backward pass for the lanes in `need`, entered by the whole wave */
  template<bool kResumable = false, class BackwardFn>
  NMPC_D void solveMasterWith(bool valid, BackwardFn && runBackward)
  {
    current_t = buf.t0 ? Base::tileBase(buf.t0, 1)[lane] : 0.0;
    lambda = cfg.initial_lambda;
    dlambda = cfg.initial_dlambda;
    sel = 0;
    dV0 = dV1 = 0;
    k_rel_norm = 0;
    J_cand = 0;
    J_cur = 0;
    phaseStart();
    const int smode = streamMode<kResumable>();
    const bool fresh = smode == 1 || (smode == 2 && stream_fresh_wg);
    const bool resumed = kResumable && ((smode == 2 && !fresh) || (smode == 0 && buf.iter_begin > 1));
    if(fresh)
    {
      valid = valid && resumeWord(3) != 0.0; // (the refill marks the slots it has put an instance into)
    }
    const int it_base = (smode == 2 && !fresh && valid) ? static_cast<int>(resumeWord(4)) : 0; // this instance's iterations before this launch
    int it_done = it_base;
    bool active = valid; // this lane still iterates
    double tr[NMPC_HIP_NTRACE];
#pragma unroll
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      tr[f] = 0;
    }
    if(resumed)
    {
      active = resumeState(valid);
    }
    else
    {
      post(kCmdRollout);
      profBegin();
      rolloutMaster();
      profEnd(1);

      tr[NMPC_HIP_TRACE_COST] = J_cur;
      tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
      tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
      tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
      if(valid)
      {
        Base::writeTraceRow(0, tr);
      }
    }
    const bool took_part = active; // (a resumed launch leaves the results of instances that had finished before it alone)
    const int iter_first = (resumed && smode == 0) ? buf.iter_begin : 1;
    const int iter_last = (smode == 1) ? 0 : ((smode == 2) ? buf.iter_end
                                              : ((kResumable && buf.iter_end > 0 && buf.iter_end < cfg.max_iter) ? buf.iter_end : cfg.max_iter));

    int retval = 0;
    for(int iter = iter_first; iter <= iter_last; iter++)
    {
      if(!__any(active))
      {
        break;
      }
      if(active)
      {
#pragma unroll
        for(int f = 0; f < NMPC_HIP_NTRACE; f++)
        {
          tr[f] = 0;
        }
        tr[NMPC_HIP_TRACE_ITER] = iter + it_base;
        tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
        retval = 0;
      }

      // ---- Step 2 (+ Step 1 in the helper): backward pass with regularisation retries    :188-214
      bool need_bw = active;
      bool bw_ok = false;
      int n_backward = 0;
      while(__any(need_bw))
      {
        const bool ok = runBackward(need_bw);
        if(need_bw)
        {
          n_backward++;
          if(ok)
          {
            bw_ok = true;
            need_bw = false;
          }
          else
          {
            dlambda = fmax(dlambda * cfg.lambda_factor, cfg.lambda_factor);
            lambda = fmax(lambda * dlambda, cfg.lambda_min);
            if(lambda > cfg.lambda_max)
            {
              need_bw = false; // failure    :196-204
            }
          }
        }
      }

      bool need_fw = false;
      if(active)
      {
        tr[NMPC_HIP_TRACE_N_BACKWARD] = n_backward;
        if(!bw_ok)
        {
          retval = -1;
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
            need_fw = true;
          }
        }
      }

      // ---- Step 3: backtracking line search    :234-274   (all searching lanes are at the same alpha index)
      const bool searched = need_fw;
      bool forward_pass_success = false;
      double alpha = 0, cost_update_actual = 0, cost_update_expected = 0, cost_update_ratio = 0;
      int ai_used = 0;
      for(int ai = 0; ai < cfg.n_alpha; ai++)
      {
        if(!__any(need_fw))
        {
          break;
        }
        const double a_try = cfg.alpha_list[ai];
        if constexpr(kNominalTail)
        {
          mailNomResident() = (ai == 0) ? 1.0 : 0.0;
        }
        post(kCmdForward);
        profBegin();
        forwardMaster(a_try);
        profEnd(1);
        if(need_fw)
        {
          alpha = a_try;
          ai_used = ai;
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
            need_fw = false;
            // accept immediately: later trials of other lanes write their candidate into THEIR candidate half
            sel = 1 - sel;
            J_cur = J_cand;
          }
        }
      }

      if(searched)
      {
        tr[NMPC_HIP_TRACE_ALPHA] = alpha;
        tr[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL] = cost_update_actual;
        tr[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED] = cost_update_expected;
        tr[NMPC_HIP_TRACE_COST_UPDATE_RATIO] = cost_update_ratio;
        tr[NMPC_HIP_TRACE_ALPHA_IDX] = ai_used;
        tr[NMPC_HIP_TRACE_N_FORWARD] = ai_used + 1;
        // ---- Step 4: accept / reject and the lambda schedule    :280-333
        if(forward_pass_success)
        {
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
        tr[NMPC_HIP_TRACE_COST] = J_cur;
        tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
        tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
      }
      if(active)
      {
        Base::writeTraceRow(iter + it_base, tr);
        it_done = iter + it_base;
        if(retval != 0 || (smode == 2 && it_done >= cfg.max_iter)) // (streamed: the instance's own max_iter-th iteration, :115-123)
        {
          active = false;
        }
      }
    }
    post(kCmdExit);
    profFlush(0);
    phaseFlush(valid, kResumable ? took_part : true, resumed);

    if(kResumable ? took_part : valid)
    {
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        Base::elem(buf.trace_last, NMPC_HIP_NTRACE, f) = tr[f];
      }
      buf.status[b] = retval;
      buf.iters[b] = static_cast<int>(tr[NMPC_HIP_TRACE_ITER]);
      buf.sel[b] = sel;
      Base::elem(buf.dV, 2, 0) = dV0;
      Base::elem(buf.dV, 2, 1) = dV1;
      if constexpr(kResumable)
      {
        if(smode != 0)
        {
          parkState(active, it_done);
        }
        else if(buf.iter_end > 0)
        {
          parkState(active && iter_last < cfg.max_iter);
        }
      }
    }
  }
  /** solveMasterWith() for solvers with lane groups (kAlphaGroups > 1): the same state machine with the step-size fan-out
      in the line search and the last trace row kept in LDS.  A separate function so that the code of the kernels without
      lane groups — the headline workload's — stays exactly as it is (it is sensitive to register allocation: -1.5 % with
      the two merged). */
  template<bool kResumable = false, class BackwardFn>
  NMPC_D void solveMasterFanOut(bool valid, BackwardFn && runBackward)
  {
    current_t = buf.t0 ? Base::tileBase(buf.t0, 1)[lane] : 0.0;
    lambda = cfg.initial_lambda;
    dlambda = cfg.initial_dlambda;
    sel = 0;
    dV0 = dV1 = 0;
    k_rel_norm = 0;
    J_cand = 0;
    phaseStart();
    const int smode = streamMode<kResumable>(); // (see solveMasterWith)
    const bool fresh = smode == 1 || (smode == 2 && stream_fresh_wg);
    const bool resumed = kResumable && ((smode == 2 && !fresh) || (smode == 0 && buf.iter_begin > 1));
    if(fresh)
    {
      valid = valid && resumeWord(3) != 0.0;
    }
    const int it_base = (smode == 2 && !fresh && valid) ? static_cast<int>(resumeWord(4)) : 0;
    int it_done = it_base;
    bool resumed_running = false;
    if(resumed)
    {
      J_cur = 0;
      resumed_running = resumeState(valid);
    }
    else
    {
      post(kCmdRollout);
      profBegin();
      rolloutMaster();
      profEnd(1);
    }

    // The trace row of an iteration is assembled and written at its end; the last row of every lane waits in LDS for the
    // end of the solve (in registers it is 24 VGPRs that are live across every pass; written to HBM every iteration it is
    // a store in front of the next pass's loads).
    auto writeLastRow = [&](const double * tr, int)
    {
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        lastRow(f) = tr[f];
      }
    };
    {
      double tr[NMPC_HIP_NTRACE];
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        tr[f] = 0;
      }
      tr[NMPC_HIP_TRACE_COST] = J_cur;
      tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
      tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
      tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
      if(valid && !resumed)
      {
        Base::writeTraceRow(0, tr);
        writeLastRow(tr, 0);
      }
    }

    int retval = 0;
    bool active = resumed ? resumed_running : valid; // this lane still iterates
    const bool took_part = active; // (a resumed launch leaves the results of instances that had finished before it alone)
    const int iter_first = (resumed && smode == 0) ? buf.iter_begin : 1;
    const int iter_last = (smode == 1) ? 0 : ((smode == 2) ? buf.iter_end
                                              : ((kResumable && buf.iter_end > 0 && buf.iter_end < cfg.max_iter) ? buf.iter_end : cfg.max_iter));
    // Extra masters: whether the FIRST pass of a line search is a wide one (twelve step sizes: waves 2 and 3 roll out too) is
    // predicted from the workgroup's previous search — wide if an instance went beyond the master's lane groups then.  The
    // nominal regime (first step size accepted) keeps the narrow pass, in which wave 2 only prefetches and the master never
    // waits for it (a wide pass is ~7 % longer: measured, profiles/r04_fanout_ab.txt); a search that exhausts the narrow pass
    // continues with wide ones.  The schedule changes which wave computes a cost, never the cost: results are independent of it.
    bool wide_first = false;
    for(int iter = iter_first; iter <= iter_last; iter++)
    {
      if(!__any(active))
      {
        break;
      }
      if(active)
      {
        retval = 0;
      }

      // ---- Step 2 (+ Step 1 in the helper): backward pass with regularisation retries    :188-214
      bool need_bw = active;
      bool bw_ok = false;
      int n_backward = 0;
      while(__any(need_bw))
      {
        const bool ok = runBackward(need_bw);
        if(need_bw)
        {
          n_backward++;
          if(ok)
          {
            bw_ok = true;
            need_bw = false;
          }
          else
          {
            dlambda = fmax(dlambda * cfg.lambda_factor, cfg.lambda_factor);
            lambda = fmax(lambda * dlambda, cfg.lambda_min);
            if(lambda > cfg.lambda_max)
            {
              need_bw = false; // failure    :196-204
            }
          }
        }
      }

      bool need_fw = false;
      if(active)
      {
        if(!bw_ok)
        {
          retval = -1;
        }
        else
        {
          if(k_rel_norm < cfg.k_rel_norm_thre && lambda < cfg.lambda_thre)
          {
            retval = 1; // :223-231
          }
          else
          {
            need_fw = true;
          }
        }
      }

      // ---- Step 3: backtracking line search    :234-274.  Every forward pass tries kAlphaGroups consecutive step sizes,
      // one per lane group (the trials of the reference's loop are independent: same nominal, same gains); the
      // instance takes the FIRST one that passes, as the sequential loop does.  Group 0's rollout goes to the candidate
      // half, the others' to the fan-out scratch, from where an accepted one is adopted (without the scratch: rolled out
      // once more, then by every group).
      const bool searched = need_fw;
      bool forward_pass_success = false;
      double alpha = 0, cost_update_actual = 0, cost_update_expected = 0, cost_update_ratio = 0;
      int ai_used = 0;
      const int last_ai = cfg.n_alpha - 1;
      // The groups fan out from the first pass on: group 0 rolls out (and stores) the first step size — the one the
      // nominal regime accepts, at the cost of a mirrored pass — while the other groups already try the next ones.
      // With extra masters (kExtraMasters: the quad kernel's waves 2 and 3, cost only) a pass covers kStepSizesPerPass step
      // sizes: the whole default alpha_list.  A step size accepted from an extra master is rolled out once more, with stores.
      for(int ai0 = 0, n_par = 0; ai0 < cfg.n_alpha; ai0 += n_par)
      {
        if constexpr(kNominalTail)
        {
          mailNomResident() = (ai0 == 0) ? 1.0 : 0.0; // the first forward pass behind a backward pass (see mailNomResident)
        }
        const bool wide_pass = kExtraMasters > 0 && (wide_first || ai0 > 0);
        n_par = wide_pass ? kStepSizesPerPass : ((ai0 == 0 && kExtraMasters == 0) ? kFanOutFirstPass : kAlphaGroups);
        if(!__any(need_fw))
        {
          break;
        }
        int g_acc = -1;
        double J_acc = 0;
        auto judge = [&](int gg, double Jc)
        {
          alpha = cfg.alpha_list[ai0 + gg];
          ai_used = ai0 + gg;
          cost_update_actual = J_cur - Jc;
          cost_update_expected = -1 * alpha * (dV0 + alpha * dV1);
          cost_update_ratio = cost_update_actual / cost_update_expected;
          if(cost_update_expected < 0)
          {
            cost_update_ratio = (cost_update_actual >= 0 ? 1 : -1); // :251-259
          }
          if(cost_update_ratio > cfg.cost_update_ratio_thre)
          {
            g_acc = gg;
            J_acc = Jc;
          }
        };
        if(kAlphaGroups == 1 || n_par == 1)
        {
          post(kCmdForward);
          profBegin();
          forwardMaster(cfg.alpha_list[ai0]); // (a wave-uniform step size: the two-wave kernel's code path)
          profEnd(1);
          if(need_fw)
          {
            judge(0, J_cand);
          }
        }
        else
        {
          // this lane group's step size, selected from uniformly indexed (scalar) reads of the list
          double my_alpha = cfg.alpha_list[ai0];
#pragma unroll
          for(int gg = 1; gg < kAlphaGroups; gg++)
          {
            const double a = cfg.alpha_list[ai0 + gg < last_ai ? ai0 + gg : last_ai];
            my_alpha = (laneGroup() == static_cast<unsigned>(gg)) ? a : my_alpha;
          }
          if constexpr(kExtraMasters > 0)
          {
            mailFirstAlpha() = static_cast<double>(ai0);
            mailWidePass() = wide_pass ? 1.0 : 0.0;
          }
          post(kCmdForwardFanOut);
          profBegin();
          forwardMaster(my_alpha);
          profEnd(1);
          if(need_fw)
          {
#pragma unroll
            for(int gg = 0; gg < kAlphaGroups; gg++)
            {
              if(ai0 + gg <= last_ai && g_acc < 0)
              {
                judge(gg, mailCostAt(waveLane() % kGroupLanes + gg * kGroupLanes));
              }
            }
            if constexpr(kExtraMasters > 0)
            {
#pragma unroll
              for(int gg = kAlphaGroups; gg < kStepSizesPerPass; gg++)
              {
                if(wide_pass && ai0 + gg <= last_ai && g_acc < 0)
                {
                  judge(gg, mailCostExtra(gg / kAlphaGroups - 1, waveLane() % kGroupLanes + (gg % kAlphaGroups) * kGroupLanes));
                }
              }
            }
          }
        }
        if constexpr(kAlphaGroups > 1)
        {
          if(__any(need_fw && g_acc > 0))
          {
            // (a step size taken from an extra master has no stored rollout: the re-roll below serves every lane of the wave)
            if(fanScratch() && !__any(need_fw && g_acc >= kAlphaGroups))
            {
              // the accepted rollout waits in the fan-out scratch: the whole workgroup copies it into the candidate half
              mailCost() = (need_fw && g_acc > 0) ? static_cast<double>(g_acc) : 0.0;
              post(kCmdAdoptFanOut);
              profBegin();
              adoptFanOut(sel);
              profEnd(1);
            }
            else
            {
              // (instances that accepted group 0's step size re-create the same candidate, the others do not care)
              if constexpr(kNominalTail)
              {
                mailNomResident() = 0.0;
              }
              post(kCmdForward);
              profBegin();
              forwardMaster((need_fw && g_acc > 0) ? alpha : cfg.alpha_list[ai0]);
              profEnd(1);
            }
          }
        }
        if(need_fw && g_acc >= 0)
        {
          forward_pass_success = true;
          need_fw = false;
          // accept immediately: later trials of other lanes write their candidate into THEIR candidate half
          sel = 1 - sel;
          J_cur = J_acc;
        }
      }

      if constexpr(kExtraMasters > 0)
      {
        if(__any(searched))
        {
          wide_first = __any(searched && (!forward_pass_success || ai_used >= kAlphaGroups));
        }
      }
      if(searched)
      {
        // ---- Step 4: accept / reject and the lambda schedule    :280-333
        if(forward_pass_success)
        {
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
      }
      if(active)
      {
        double tr[NMPC_HIP_NTRACE];
#pragma unroll
        for(int f = 0; f < NMPC_HIP_NTRACE; f++)
        {
          tr[f] = 0;
        }
        tr[NMPC_HIP_TRACE_ITER] = iter + it_base;
        tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
        tr[NMPC_HIP_TRACE_N_BACKWARD] = n_backward;
        if(bw_ok)
        {
          tr[NMPC_HIP_TRACE_K_REL_NORM] = k_rel_norm;
        }
        if(searched)
        {
          tr[NMPC_HIP_TRACE_ALPHA] = alpha;
          tr[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL] = cost_update_actual;
          tr[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED] = cost_update_expected;
          tr[NMPC_HIP_TRACE_COST_UPDATE_RATIO] = cost_update_ratio;
          tr[NMPC_HIP_TRACE_ALPHA_IDX] = ai_used;
          tr[NMPC_HIP_TRACE_N_FORWARD] = ai_used + 1;
          tr[NMPC_HIP_TRACE_COST] = J_cur;
          tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
          tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
        }
        Base::writeTraceRow(iter + it_base, tr);
        writeLastRow(tr, iter + it_base);
        it_done = iter + it_base;
        if(retval != 0 || (smode == 2 && it_done >= cfg.max_iter))
        {
          active = false;
        }
      }
    }
    post(kCmdExit);
    profFlush(0);
    phaseFlush(valid, kResumable ? took_part : true, resumed);

    if(kResumable ? took_part : valid)
    {
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        Base::elem(buf.trace_last, NMPC_HIP_NTRACE, f) = lastRow(f);
      }
      buf.iters[b] = static_cast<int>(lastRow(NMPC_HIP_TRACE_ITER));
      buf.status[b] = retval;
      buf.sel[b] = sel;
      Base::elem(buf.dV, 2, 0) = dV0;
      Base::elem(buf.dV, 2, 1) = dV1;
      if constexpr(kResumable)
      {
        if(smode != 0)
        {
          parkState(active, it_done);
        }
        else if(buf.iter_end > 0)
        {
          parkState(active && iter_last < cfg.max_iter);
        }
      }
    }
  }
};

/** The 2-wave solve kernel: grid = Bp / 64 workgroups of 128 threads; wave 0 = master, wave 1 = helper. */
template<class Problem, bool kConstrained, bool kOwnProblem = false, bool kResumable = false>
__global__ __launch_bounds__(2 * kLanesPerBlock) void ddp_solve_tpi2w_kernel(const Problem problem,
                                                                              const nmpc_hip_ddp_config cfg,
                                                                              const DeviceBuffers buf)
{
  using Solver = PairSolver<Problem, kConstrained>;
  extern __shared__ __attribute__((aligned(16))) double lds_2w[];
  const int wave = threadIdx.x / kLanesPerBlock;
  const int b = blockIdx.x * kLanesPerBlock + (threadIdx.x % kLanesPerBlock);
  int first = 0;
  bool solver_fresh = false;
  if constexpr(kResumable)
  {
    // streamed solves, the rollout of freshly filled slots: the workgroups below hold instances in mid-solve and stay out of it
    // (*first_active is a multiple of 64: no workgroup holds both kinds)
    const int first_fresh = (buf.stream_mode != 0 && buf.first_active) ? *buf.first_active : 0;
    solver_fresh = buf.stream_mode == 2 && buf.first_active && static_cast<int>(blockIdx.x) * kLanesPerBlock >= first_fresh;
    first = (buf.stream_mode == 1) ? first_fresh : 0;
    if(static_cast<int>(blockIdx.x + 1) * kLanesPerBlock <= first
       || (buf.stream_mode != 0 && buf.n_active && static_cast<int>(blockIdx.x) * kLanesPerBlock >= *buf.n_active)) // (... and the empty slots behind)
    {
      return;
    }
  }
  // kOwnProblem: every instance has its own problem object (nmpc_hip_ddp_set_model_params_batch).  A separate
  // instantiation: with it the problem's fields live in VGPRs, which costs the shared-object kernel ~1 % if merged in.
  const Problem mine = kOwnProblem ? instanceProblem(problem, buf, b) : problem;
  Solver solver(mine, cfg, buf, b, lds_2w);
  solver.stream_fresh_wg = solver_fresh;
  if(wave == 0)
  {
    if constexpr(kResumable)
    {
      // (positions [0, *n_active) hold the instances that still iterate: the host's compaction between launches, capi.hip)
      solver.template solveMaster<true>(b >= first && b < (buf.n_active ? *buf.n_active : buf.B));
    }
    else
    {
      solver.solveMaster(b < buf.B);
    }
  }
  else
  {
    solver.current_t = buf.t0 ? solver.tileBase(buf.t0, 1)[solver.lane] : 0.0;
    solver.helperLoop();
  }
}
} // namespace hip
} // namespace nmpc_amd
