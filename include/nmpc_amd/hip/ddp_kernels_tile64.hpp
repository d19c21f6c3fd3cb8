// gfx950 device code of the batched DDP solver, fp64, LANE MAPPING "TILE64": one workgroup of eight wavefronts solves a GROUP
// of up to 32 problem instances whose blocks fill 16 x 16 matrix-core tiles (5 <= n <= 15, static 1 <= m <= 8: BASELINE.json's
// manipulator n 14 m 7 — config 5 — and the quadrotor n 12 m 4 in the reference's arithmetic), derivatives never in HBM.
//
// What a lane means changes with the phase (reference: nmpc_ddp/include/nmpc_ddp/DDPSolver.hpp):
//
//   model code  — initial rollout (:83-95), the linearisation sweep (:157-185), the first step size of every line search
//                 (:234-274): lane = INSTANCE on wave 0 ("model wave"); the later step sizes of alpha_list: lane =
//                 (instance, step size) on the other waves, all at once (the trials are independent: same nominal, same gains).
//   backward    — (:342-534) lane = MATRIX ENTRY on waves 1..7 ("matrix waves"), one instance at a time per wave, on
//                 v_mfma_f64_16x16x4_f64; the instances of a group are dealt round-robin to the seven matrix waves.
//
// Derivative records.  The model wave linearises timestep i - 1 of every instance of the group into an LDS record while the
// matrix waves consume the records of timestep i (two record slots per instance, one barrier per timestep).  A record holds
// only the entries of Fx, Fu, Lxx, Lxu, Luu, Lx, Lu that are NOT structural zeros of the problem's functors: an entry the
// compiler proves to be the constant 0 after inlining (__builtin_constant_p, the same mechanism as macc() in ddp_kernels.hpp)
// is never stored, its position in the offset table points at the record's zero word.  The table is produced at kernel start by
// the very code that writes the records (emitRecord with a different sink), so writer and readers cannot disagree.  For the
// manipulator that is 204 instead of 752 doubles per record, which is what lets 32 instances share the 160 KB of a CU: a dense
// fp64 record set would cap the group at 8 instances and leave 56 of the model wave's 64 lanes idle.
// The group size follows from the table at run time: G = min(32, what the LDS holds, ceil(B / workgroups)); workgroups are
// persistent (grid = number of CUs) and loop over groups, so there is no batch-size cliff.
//
// The backward step in "natural layout".  A 16 x 16 fp64 matrix X lives in four registers: register r of lane (q = lane / 16,
// j = lane % 16) holds X[4 r + q][j] — the matrix core's C / D layout (measured: scripts/ubench_mfma_f64_16.hip).  The same four
// registers passed as the A operands of four MFMAs, with another matrix's registers as B operands, contract over the row index
// in ascending order: mma(X, Y) = X^T Y, an fma chain per entry, with no LDS round trip or cross-lane move between chained
// products.  With VV = [Vxx | Vx] (n x (n + 1), one tile since n <= 15) and A = [K | k] (m x (n + 1)):
//     Pa  = mma(VV, Fx)            rows < n: Vxx Fx = (Fx^T Vxx)^T,  row n: Vx^T Fx            :386-408 in the reference's
//     Pb  = mma(VV, Fu)            rows < n: Vxx Fu,                 row n: Vx^T Fu            left-to-right association
//     Qxx = Lxx + mma(Pa, Fx)      Qux = Lxu^T + mma(Pb, Fx)        Quu = Luu + mma(Pb, Fu)    Qx, Qu: L + row n of Pa, Pb
//     Quu_F (= Quu + lambda I, or rebuilt from Vxx + lambda I: :421-441) and [Qux_reg | Qu] pass through the wave's LDS scratch:
//     every lane factorises Quu_F (L D L^T, one reciprocal per pivot), lane (., j) solves column j                  :500-517
//     Z   = mma(Quu, A)            C1 = mma(Z, A)                   T2 = mma(A, [Qux | Qu])    T3 = mma(Qux, A)
//     VV' = (([Qxx | Qx] + C1) + T2) + T3;   entry (n, n) of C1 is k^T Quu k, of T2 is k^T Qu: dV for free        :522-526
//     Vxx <- (Vxx + Vxx^T) / 2 with the transpose taken through the wave's LDS scratch                              :527
// 5 ceil(n/4) + 4 ceil(m/4) MFMAs per instance and timestep (manipulator 28, quadrotor 19) of 64 cycles each, and ~350 other
// instructions, most of them the m x m factorisation.  fp64 MFMAs of two waves on one SIMD serialise, an fp64 VALU wave beside
// a matrix wave still gets about a third of its issue slots (profiles/r03_ubench_mfma_f64_16.txt): two waves per SIMD.
//
// Gains are kept as instance-major records in the handle's workspace (ModelOps::gain_layout 1: one contiguous run per instance
// and timestep), X / U / cost live in the two halves of the handle's tile-major arrays with a per-instance `sel` flip on acceptance,
// exactly like the lane kernels.
#pragma once

#include <atomic>

#include <cstring>
#include <new>
#include <type_traits>

#include <nmpc_amd/hip/ddp_kernels.hpp>
#include <nmpc_amd/hip/model_ops.hpp>

namespace nmpc_amd
{
namespace hip
{
typedef double v4d64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
/** What the tile kernel needs to know about its arithmetic type S (Problem::Scalar). */
template<class S>
struct T64Scalar;
template<>
struct T64Scalar<double>
{
  using Vec4 = v4d64;
  //! logical column <-> lane of a 16-lane row: v_mfma_f64_16x16x4 keeps column j in lane j
  NMPC_D static constexpr int colOf(int p)
  {
    return p;
  }
};
template<>
struct T64Scalar<float>
{
  using Vec4 = v4f32;
  //! v_mfma_f32_16x16x4 puts row i of its result into register i % 4 of lane group i / 4 (the f64 instruction: register i / 4,
  //! lane group i % 4).  With the logical COLUMN 4 (p % 4) + p / 4 in lane p of a 16-lane row — the transposition of the 4 x 4
  //! index grid, its own inverse — the result rows, which are the first operand's columns, land where the f64 layout has them:
  //! register r of lane group q = logical row 4 r + q, for operands and results alike, and the k-slices of a contraction can be
  //! skipped exactly as in double.
  NMPC_D static constexpr int colOf(int p)
  {
    return 4 * (p & 3) + (p >> 2);
  }
};

//! Wavefronts per workgroup: 8 (two per SIMD, 256 registers each) — or 12 (three per SIMD, 168 registers) for the small float
//! instantiations, whose step is a latency chain (VALU issue share 0.40, s_waitcnt / barrier 0.56 on c4: profiles/r05_pmc_summary_c4.txt)
//! and whose code fits 168 registers: eleven matrix waves of at most three slots hide more of one another's latencies.  Measured
//! (profiles/r05_tile64_waves_ab.txt, 8 / 12 / 16 waves): c4 quadrotor fp32 1.94 / 2.18 / 2.14 k it/s; quadrotor fp64 1.74 / 1.75 /
//! 1.72 k (stays at 8); manipulator 1.43 / 0.94 / 0.38 k and centroidal 396 / 243 / 159 (their steps spill at 168 registers).
//! -DNMPC_T64_WAVES=n forces one count for every instantiation (A/B builds).
template<class Problem>
struct T64Waves
{
#ifdef NMPC_T64_WAVES
  static constexpr int value = NMPC_T64_WAVES;
#else
  static constexpr int value =
      (sizeof(typename Problem::Scalar) == 4 && Problem::kStateDim * (Problem::kStateDim + Problem::kInputDimMax) <= 200) ? 12 : 8;
#endif
  static_assert(value == 8 || value == 12 || value == 16, "8, 12 or 16 wavefronts per workgroup");
};
//! Instances per group at most (lanes 0 .. 34 of the model wave; five per matrix wave of an eight-wave workgroup).  A full chip's
//! round is 256 x 32 = 8192 instances; the three extra slots take batches up to 8960 in ONE round — 8200 instances were two rounds
//! of 17-slot groups, 1.30 x the time of 8192 (profiles/r04a_tile64_chunk_ab.txt) — where the LDS holds the records (manipulator:
//! 2 x 35 x 205 doubles).
constexpr int kT64MaxGroup = 35;
constexpr size_t kT64LdsBytes = 160 * 1024; //!< the whole LDS of a CU: one workgroup per CU

template<class Problem, bool kConstrained = false, bool kOwnProblem = false>
struct TileSolver64
{
  using S = typename Problem::Scalar; //!< double: the reference's arithmetic; float: BASELINE.json's config 4 (round 4)
  static constexpr int kT64Waves = T64Waves<Problem>::value; //!< wave 0: model code; wave w runs on SIMD w % 4
  static constexpr int kT64Threads = kT64Waves * 64;
  static constexpr int kT64MatrixWaves = kT64Waves - 1;
  //! matrix waves that share SIMD 0 with the model wave (4, 8, 12), and the others ("spec" waves: numbered 0 .. kSpecWaves - 1 in
  //! wave order) — the line search's roles are dealt by these numbers (specIndex, ringOrder)
  static constexpr int kSharedWaves = (kT64Waves - 1) / 4;
  static constexpr int kSpecWaves = kT64MatrixWaves - kSharedWaves;
  NMPC_D static int specIndex(int w)
  {
    return (w >= 1 && (w & 3) != 0) ? (w - 1) - (w >> 2) : -1;
  }
  //! order in which the matrix waves take rows of the ring prefetch: the spec waves, then the waves on the model wave's SIMD
  NMPC_D static int ringOrder(int w)
  {
    return ((w & 3) == 0) ? kSpecWaves + (w >> 2) - 1 : specIndex(w);
  }
  using Vec4 = typename T64Scalar<S>::Vec4;
  static constexpr bool kF32 = std::is_same<S, float>::value;
  static_assert(std::is_same<S, double>::value || kF32, "the tile kernel computes in S or in float");
  static constexpr int kW = 8 / static_cast<int>(sizeof(S)); //!< elements of S per 8-byte word (the fixed LDS area is laid out in words)
  using Buffers = DeviceBuffersT<S>;
  /** Logical column of the 16 x 16 tiles that lane p of a 16-lane row holds, and back (the map is its own inverse). */
  NMPC_D static constexpr int colOf(int p)
  {
    return T64Scalar<S>::colOf(p);
  }
  NMPC_D static constexpr int laneOfCol(int c)
  {
    return T64Scalar<S>::colOf(c);
  }
  /** A configuration value in the arithmetic type (the float oracle casts once, then computes in float). */
  NMPC_D static S sc(double v)
  {
    return static_cast<S>(v);
  }
  static constexpr int N = Problem::kStateDim;
  static constexpr int M = Problem::kInputDimMax;
  static constexpr int MM = M;
  //! input dimension known per timestep only (inputDim(t): the reference's centroidal-motion problem, 16 / 0)
  static constexpr bool kDyn = Problem::kDynamicInput;
  //! m > 8, or a run-time m: the m x m factorisation does not fit a lane's registers (16 x 16 doubles) — the gains are computed
  //! in NATURAL LAYOUT instead, the factor spread over the wave (stepGainsNatural); m <= 8 static: every lane factorises
  static constexpr bool kBig = kDyn || M > 8;
  static_assert(N >= 1 && N <= 15, "[Vxx | Vx] is one 16-column tile");
  static_assert(M >= 1 && M <= 16, "[K | k] is one tile of m <= 16 rows");
  static_assert(!(kConstrained && kBig), "BoxQP on the tile kernel: static m <= 8 (boxQPMasked runs in a lane's registers)");
  static_assert(!(kF32 && (kBig || kConstrained)), "float: unconstrained solves with a static m <= 8 (the fp32 tile kernel of "
                                                   "ddp_kernels_tile32.hpp takes the box-constrained ones)");
  static constexpr bool kShape = true;
  static constexpr int KN = (N + 3) / 4; //!< k-slices of a contraction over state rows
  static constexpr int KM = (MM + 3) / 4; //!< ... over input rows; also the registers of an m-row tile that hold anything
  static constexpr int rN = N / 4, qN = N % 4; //!< row n of a natural-layout tile: register rN of lane group qN
  static constexpr int kStarLane = 16 * qN + T64Scalar<typename Problem::Scalar>::colOf(N); //!< the lane whose register rN is entry (n, n)
  //! n + m <= 16 with n a multiple of 4 (the quadrotor): [Fx Fu] and [[Lxx Lxu],[Lxu^T Luu]] are ONE tile each, and all five
  //! Q blocks come out of two products (G = VV^T F, Q = G^T F + L: 2 ceil(n/4) MFMAs instead of 5 ceil(n/4)); rows n .. n+m-1 of
  //! Q are then whole registers of the lane groups (n % 4 == 0), i.e. Qux / Quu without any cross-lane-group move
  static constexpr bool kAug = (N % 4 == 0) && (N + MM <= 16);
  //! the slot table's sKrel holds max_i (|k_i| / (|u_i| + 1))^2 during a sweep (see stepValueUpdate)
  static constexpr bool kKrelSquared = (M > 1);
  static constexpr int NA = N + MM;
  using Lane = InstanceSolver<Problem, kConstrained>; //!< the lane kernels' scalar helpers (ldltInPlace, boxQP): same bits

  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using StateStateDimMatrix = typename Problem::StateStateDimMatrix;
  using InputInputDimMatrix = typename Problem::InputInputDimMatrix;
  using StateInputDimMatrix = typename Problem::StateInputDimMatrix;

  // ---- record entries in canonical order (the order emitRecord() visits them)
  static constexpr int idFx = 0; //!< Fx(r, c) at idFx + c N + r
  static constexpr int idFu = idFx + N * N; //!< Fu(r, a) at idFu + a N + r
  static constexpr int idLxx = idFu + N * MM;
  static constexpr int idLxuT = idLxx + N * N; //!< Lxu(c, a) = Lxu^T[a][c] at idLxuT + c M + a
  static constexpr int idLuu = idLxuT + N * MM; //!< Luu(a, c) at idLuu + c M + a
  static constexpr int idLx = idLuu + MM * MM;
  static constexpr int idLu = idLx + N;
  static constexpr int idInvU = idLu + MM; //!< 1 / (|u_i| + 1)    :217-221
  static constexpr int idU = idInvU + 1; //!< u_i (box-constrained solves: the QP's bounds are limits - u_i, :470-472)
  static constexpr int idM = idU + (kConstrained ? MM : 0); //!< inputDim(t_i) as a S (run-time input dimension only)
  static constexpr int kNumIds = idM + (kDyn ? 1 : 0);

  // ---- LDS layout, in elements of S (tables of other types: 8-byte words, kW elements each).  Fixed part first, the record area takes the rest.
  static constexpr int kTblAt = 0; //!< unsigned short tbl[kNumIds]: entry -> offset in a record (0 = the zero word)
  static constexpr int kMetaAt = kTblAt + kW * ((kNumIds + 3) / 4); //!< ints: 0 record stride, 1 group size, 2.. flags
  enum MetaField
  {
    mStride = 0,
    mGroup,
    mAnyIter,
    mAnyRetry,
    mAnyLs,
    mAnyMore, //!< a slot's first step size was rejected: the later ones are tried
    mAnyReroll, //!< a later step size was taken: its trajectory has to be stored
    mNAct, //!< slots that take part in the coming backward sweep (sBw set), listed in act()
    mChunk, //!< timesteps the model wave linearises per pass in that sweep (see backwardSweepModel)
    mWide, //!< this line search rolls out the later step sizes WITH the first one (see solveGroup)
    mRejected, //!< the group's previous line search: 4 x slots that rejected the first step size >= slots that searched
    kNumMeta = 12
  };
  //! ints: [0..31] active index -> slot, [32..63] slot -> active index (-1: the slot does not take part in the sweep)
  static constexpr int kActAt = kMetaAt + kW * (kNumMeta / 2);
  static constexpr int kSlotAt = kActAt + kW * kT64MaxGroup; //!< the slot table: an 8-byte word per (field, slot)
  enum SlotField
  {
    // what the roles tell each other, and the per-instance solver state of the model wave between its phases (nothing of it
    // is live in registers across a sweep or a rollout)
    sB = 0, //!< int: instance index, -1 = empty slot
    sBw, //!< int: this sweep computes gains for the slot
    sLs, //!< int: the slot takes part in the running line-search pass
    sSel, //!< int: half of X / U / cost that holds control_data_
    sOk, //!< int: backwardPass() returned true
    sIter, //!< int: iterations started
    sRet, //!< int: procOnce's return value
    sFlags, //!< int: bit 0 running, 1 inside a procOnce, 2 backward pass pending, 3 in the line search, 4 step size found
    sNBw, //!< int: backward passes of this iteration
    sAi, //!< int: index of the last step size judged
    sLambda,
    sDlambda,
    sT0, //!< current_t
    sJcur, //!< control_data_.cost_list.sum()
    sJcand,
    sAlpha, //!< last step size judged (the one to re-roll)
    sActual,
    sExpected,
    sRatio,
    sDV0,
    sDV1,
    sKrel,
    sTicksBw, //!< unsigned long long
    sTicksFw,
    kNumSlotFields
  };
  enum SlotFlag
  {
    fRunning = 1,
    fInIter = 2,
    fNeedBw = 4,
    fInLs = 8,
    fSuccess = 16
  };
  static constexpr int kKnextAt = kSlotAt + kW * kNumSlotFields * kT64MaxGroup; //!< [kT64MaxGroup][8]: k_{i+1}, the BoxQP warm start (:452-467)
  static constexpr int kWaveAt = kKnextAt + (kConstrained ? kT64MaxGroup * 8 : 0);
  // per matrix wave: the column exchange of the gain computation, then (aliased: one wave's LDS traffic is ordered) the transposition
  //! leading dimension of the exchanged columns: ODD, so that the sixteen lanes of a row (one column each) hit different banks
  //! (with 8 they shared two bank groups: an 8-way conflict on every exchange access, half of the kernel's LDS cycles)
  //! ... and at least 4 KM: a lane writes the rows 4 r + q (r < KM) of ITS column whether they exist or not — rows >= m and
  //! columns >= n land in padding nobody reads — so the write addresses are the lane's column base plus constants (they were a
  //! compare, a select and a shift each)
  static constexpr int kColLd = 4 * KM + 1;
  static constexpr int wQQ = 0; //!< [Qux_reg | Qu]: column j at kColLd j, rows a < m
  static constexpr int wF = wQQ + 9 * 16; //!< Quu_F: column c at wF + kColLd c (sixteen columns are written, m are read)
  static constexpr int wX = wF + 9 * 16; //!< Qx, row 4 r + q at wX + 4 q + r
  static constexpr int wExchange = wX + 16;
  static constexpr int kTrLd = 17; //!< leading dimension of the transposition scratch: conflict-free both ways
  static constexpr int wT = 0;
  // Natural-layout gains (kBig), round 6: the factorisations of a wave's slots of a timestep run TOGETHER, sixteen lanes per slot and
  // a column per lane (gainsPrepare / gainsBatch / gainsComplete).  A slot's Quu_F and [Qux_reg | Qu] pass through this staging on
  // their way from the natural layout (register r of lane (q, c) = entry (4 r + q, c)) into the registers of ONE lane group, and the
  // solved columns the same way back: column c at kStLd c, row 4 r + q at position 4 q + r of its column (what a natural-layout lane
  // holds of a column is contiguous).  kStLd = 9 x 16 bytes: the sixteen lanes of a row hit different banks with their 16-byte accesses.
  // One slot at a time, aliased with the transposition scratch (one wave's LDS traffic is ordered).
#ifdef NMPC_AMD_AB_NATURAL_GAINS
  static constexpr bool kBatchGains = false; // A/B builds: every slot factorises over the whole wave (stepGainsNatural), as until round 5
#else
  static constexpr bool kBatchGains = kBig && !kConstrained && std::is_same<S, double>::value;
#endif
  static constexpr int kStLd = 18;
  static constexpr int wStA = 0; //!< Quu_F, both triangles
  static constexpr int wStR = 16 * kStLd; //!< [Qux_reg | Qu], then the solved columns x
#ifdef NMPC_AMD_AB_FORCE_STAGING_SPACE
  static constexpr int kStDoubles = 2 * 16 * kStLd; // (A/B: the LDS layout of the batched gains without their code)
#else
  static constexpr int kStDoubles = kBatchGains ? 2 * 16 * kStLd : 0;
#endif
  static constexpr int wZero0 = (wExchange > 16 * kTrLd ? wExchange : 16 * kTrLd);
  static constexpr int wZero = (wZero0 > kStDoubles ? wZero0 : kStDoubles); //!< sixteen zeros, read by lanes outside a block
                                                                                   //!< (with the immediate offsets of the lanes inside)
  static constexpr int wDump = wZero + 16; //!< written by lanes outside a block
  static constexpr int kWaveDoubles = wDump + 2;
  static constexpr int kScratchPerWave = 1;
  static constexpr int kLsAt = kWaveAt + kT64MatrixWaves * kScratchPerWave * kWaveDoubles; //!< lsJ[NMPC_HIP_MAX_ALPHA][kT64MaxGroup]: cost of every trial
  // Box-constrained solves with a per-lane factorisation (static m <= 8), round 5: the QPs of a matrix wave's (up to five) slots of a
  // timestep are solved TOGETHER, lane e = slot e of the wave (backwardSweepMatrix) — one pass of the QP code per wave and timestep
  // instead of five with all 64 lanes solving the same problem.  What a slot's QP reads (Quu_F, Qu) waits here between the wave's
  // "prepare" trip over its slots and the batch; the last slot's is still in the wave's exchange scratch.  Per wave: four inputs of
  // MM * MM + MM elements, five results (free set, return code).  The area starts where lsJ lives: the line search is idle during a sweep.
  static constexpr bool kBatchQP = kConstrained && !kBig && std::is_same<S, double>::value;
  // Matrix waves that OWN slots in the backward sweep, and slots per owner: seven owners of up to five slots.  (Measured and not
  // kept for the batched QP, -DNMPC_AMD_AB_QP_OWNERS=4: four owners of up to nine slots — one QP pass per SIMD and timestep instead of
  // two — are SLOWER, manipulator box 12.2 -> 15.4 ms, quadrotor box 4.8 -> 5.8: a pass lasts as long as the slowest of its lanes'
  // QPs, whose iteration counts are heavy-tailed, and nine slots' steps run one after the other.  profiles/r05_constrained_tile64_ab.txt)
#ifdef NMPC_AMD_AB_QP_OWNERS
  static constexpr int kOwnerWaves = kBatchQP ? NMPC_AMD_AB_QP_OWNERS : kT64MatrixWaves;
#else
  static constexpr int kOwnerWaves = kT64MatrixWaves;
#endif
  //! Instances per group at most for THIS instantiation.  The batched natural-layout gains keep a slot's operands of the value update in
  //! registers between its two trips (GainStash): with more than three slots per wave the trips spill, and a spilled step costs more
  //! than the batch saves [measured, profiles/r06_centroidal_gains_ab.txt].  -DNMPC_AMD_AB_BIG_GROUP=n: A/B builds.
#ifdef NMPC_AMD_AB_BIG_GROUP
  static constexpr int kGroupMax = kBatchGains ? NMPC_AMD_AB_BIG_GROUP : kT64MaxGroup;
#else
  static constexpr int kGroupMax = kBatchGains ? 3 * kT64MatrixWaves : kT64MaxGroup;
#endif
  static constexpr int kPerWave = (kGroupMax + kOwnerWaves - 1) / kOwnerWaves;
  //! The slot loop of the sweep as straight-line code — one copy of the step per slot, no rotation of the value functions' registers
  //! (eight moves per slot and step) — where the step is small: the quadrotor's (n (n + m) = 192) gains 6.5 % in float and 7 % in
  //! double, the manipulator's (294) loses 23 % (five copies of a 673-instruction step: spills) [measured, round 6, A/B on one box:
  //! c4 2.10 -> 1.97 ms, c4f64 3.90 -> 3.63, c5 2.67 -> 3.29 — and 3.29 as well with scheduling barriers between the copies: 3 026 instead of
  //! 3 365 instructions per timestep, but 60 scratch accesses behind 32 s_waitcnt vmcnt(0) in the loop; round 3 had measured no difference
  //! for the quadrotor].
#ifdef NMPC_AMD_AB_ROLLED_SLOTS
  static constexpr bool kUnrollSlots = false;
#else
  static constexpr bool kUnrollSlots = !kBig && !kConstrained && N * (N + MM) <= 200;
#endif
  static constexpr int kQpIn = MM * MM + MM;
  static constexpr int kQpPerWave = (kPerWave - 1) * kQpIn + 2 * kPerWave;
  static constexpr int kQpAt = kLsAt;
  static constexpr int kLsDoubles = NMPC_HIP_MAX_ALPHA * kT64MaxGroup;
  static constexpr int kLsOrQp = (kBatchQP && kOwnerWaves * kQpPerWave > kLsDoubles) ? kOwnerWaves * kQpPerWave : kLsDoubles;
  static constexpr int kTraceAt = kLsAt + kLsOrQp; //!< trace row of the running iteration, [field][kT64MaxGroup]
  static constexpr int kProfAt = (kTraceAt + NMPC_HIP_NTRACE * kT64MaxGroup + 1) & ~1; //!< profiling builds: 40 tick counters of workgroup 0
#ifdef NMPC_AMD_PROFILE_TILE64
  static constexpr int kFixedRaw = kProfAt + kW * 40;
#else
  static constexpr int kFixedRaw = kProfAt + kW * 16;
#endif
  static constexpr int kRecAt = (kFixedRaw + 3) & ~3; //!< records: rec[2][G][stride]; before a sweep: [Vxx | Vx] per slot (16-byte aligned)
  static constexpr int kTerm = (N + 1) * N; //!< terminal record: n + 1 columns of n rows
  // line search: ring of nominal records [depth][row][G] in the record area; rows of a timestep: k_i (m), K_i (m n, column-major),
  // x_i (n), u_i (m)
  static constexpr int kGainRows = MM + MM * N;
  static constexpr int kRingRows = kGainRows + N + MM;
  static constexpr int kRingDepth = 3;
  //! doubles per slot of a ring entry: the rows of a slot are CONTIGUOUS (round 4) — the rolling lane reads them with immediate
  //! offsets, two per instruction (with the rows G doubles apart every read had its own address: a multiply, an add and a
  //! wait each — three quarters of a rollout timestep's instructions); twice an odd number: sixteen slots' 16-byte reads
  //! cover the 64 banks once
  static constexpr int kRingAl = 16 / static_cast<int>(sizeof(S)); //!< elements per 16-byte read
  static constexpr int kRingStride =
      kRingAl * ((((kRingRows + kRingAl - 1) / kRingAl) % 2 == 1) ? (kRingRows + kRingAl - 1) / kRingAl : (kRingRows + kRingAl - 1) / kRingAl + 1);
  /** Per-instance workspace: the gains as records [T][k_i (m) | K_i (m n, column-major)] — a matrix wave writes the 105 doubles of
      an (instance, timestep) as one contiguous run; into the handle's tile-major kff / Kfb arrays the same stores would be 8 bytes
      each, 512 bytes apart (measured: 7 x the written bytes reach HBM). */
  NMPC_HD static size_t gainDoubles(int T)
  {
    return static_cast<size_t>(T) * kGainRows;
  }
  //! Behind the gain records of the whole batch: the candidate trajectories of the LATER step sizes of a line search, one per
  //! (instance, step size) as rows [x_0 .. x_T | u_0 .. u_T-1 | cost_0 .. cost_T] — what the lanes that roll them out for their
  //! cost leave behind, so that a later step size that is taken is COPIED to the handle's arrays instead of rolled out again
  //! (adoptCandidates; the default alpha_list: ten later step sizes — longer lists re-roll)
  static constexpr int kScratchAlphas = 10;
  NMPC_HD static size_t candDoubles(int T)
  {
    return static_cast<size_t>(T + 1) * N + static_cast<size_t>(T) * MM + static_cast<size_t>(T + 1);
  }
  NMPC_HD static size_t workspaceDoubles(int T)
  {
    return gainDoubles(T) + kScratchAlphas * candDoubles(T);
  }
  NMPC_D S * candidate(int b, int later_index) const
  {
    return buf.wpi_ws + static_cast<size_t>(buf.B) * gainDoubles(T) + (static_cast<size_t>(b) * kScratchAlphas + later_index) * candDoubles(T);
  }
  static constexpr int kLdsDoubles = static_cast<int>(kT64LdsBytes / sizeof(S));
  static_assert(kRecAt + 2 * (kNumIds + 2) <= kLdsDoubles && kRecAt + kTerm <= kLdsDoubles, "one instance must fit");

  const Problem & problem;
  const nmpc_hip_ddp_config & cfg;
  const Buffers & buf;
  const int T;
  const int wave;
  const int lane;
  S * lds;
  int stride = 0; //!< doubles per record (odd: the model wave's lanes spread over the banks)
  int G = 1; //!< instances per group
  int group_cap;
  int chunk_cap; //!< at most this many timesteps per pass of the model code (0: what fits; A/B measurements, tests)
  int wide_cap; //!< 0: the later step sizes of a line search never ride along with the first one (A/B measurements, tests)
  int adopt_cap; //!< 0: a later step size that is taken is rolled out again instead of copied from the workspace (A/B, tests)
  int pair_cap; //!< 0: the second step size never rides in the model wave's upper lanes (A/B measurements, tests)

  NMPC_D TileSolver64(const Problem & p, const nmpc_hip_ddp_config & c, const Buffers & bf, S * lds_base, int cap)
  : problem(p), cfg(c), buf(bf), T(bf.T), wave(static_cast<int>(threadIdx.x) >> 6), lane(static_cast<int>(threadIdx.x) & 63),
    lds(lds_base), group_cap(cap & 0xffff), chunk_cap((cap >> 16) & 0x1fff), wide_cap(((cap >> 31) & 1) == 0 ? 1 : 0),
    adopt_cap(((cap >> 30) & 1) == 0 ? 1 : 0), pair_cap(((cap >> 29) & 1) == 0 ? 1 : 0)
  {
  }

  // ---- LDS views
  NMPC_D unsigned short * tbl() const
  {
    return reinterpret_cast<unsigned short *>(lds + kTblAt);
  }
  NMPC_D int & meta(int k) const
  {
    return reinterpret_cast<int *>(lds + kMetaAt)[k];
  }
  NMPC_D S & slotF(int field, int slot) const
  {
    return lds[kSlotAt + kW * (field * kT64MaxGroup + slot)];
  }
  NMPC_D int & slotI(int field, int slot) const
  {
    return reinterpret_cast<int *>(lds + kSlotAt + kW * (field * kT64MaxGroup + slot))[0];
  }
  NMPC_D unsigned long long & slotT(int field, int slot) const
  {
    return reinterpret_cast<unsigned long long *>(lds + kSlotAt + kW * (field * kT64MaxGroup + slot))[0];
  }
  /** Record of active index a, timestep offset dt inside a chunk of `chunk` timesteps, buffer `parity` of the two. */
  NMPC_D S * recAt(int parity, int dt, int a, int chunk, int n_act) const
  {
    return lds + kRecAt + ((parity * chunk + dt) * n_act + a) * stride;
  }
  NMPC_D int & actSlot(int a) const
  {
    return reinterpret_cast<int *>(lds + kActAt)[a];
  }
  NMPC_D int & actIndex(int slot) const
  {
    return reinterpret_cast<int *>(lds + kActAt)[kT64MaxGroup + slot];
  }
  /** Records the record area holds (two buffers of chunk x n_act are needed). */
  NMPC_D int recordCapacity() const
  {
    return (kLdsDoubles - kRecAt) / stride;
  }
  NMPC_D S * term(int slot) const
  {
    return lds + kRecAt + slot * kTerm;
  }
  NMPC_D S * waveScratch(int which = 0) const
  {
    return lds + kWaveAt + ((wave - 1) * kScratchPerWave + which) * kWaveDoubles;
  }
  NMPC_D static void barrier()
  {
    syncThreadsFuzzed(6);
  }
  /** A barrier that PUBLISHES global memory written by this wave (gains, candidate trajectories) to the other waves of the
      workgroup.  __syncthreads() waits for LDS traffic only (workgroup scope: the compiler relies on the CU's shared L1 keeping
      the waves' global accesses in order); the stores are drained explicitly so that nothing depends on that. */
  NMPC_D static void publishBarrier()
  {
    fuzzSched(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fuzzSched(8);
  }
  /** Lanes of ONE wave exchange data through LDS without a barrier (the LDS executes a wave's instructions in order); the
      compiler, which reasons per thread, must be kept from moving a lane's reads above the other lanes' writes. */
  NMPC_D static void fence()
  {
    asm volatile("" ::: "memory");
  }
  NMPC_D static int uniform(int v)
  {
    return __builtin_amdgcn_readfirstlane(v);
  }
  NMPC_D static S uniformD(S v)
  {
    if constexpr(kF32)
    {
      return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    }
    else
    {
      const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
      const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
      return __hiloint2double(hi, lo);
    }
  }
  /** v of the lane whose id times four is `byte_index` (ds_bpermute: the LDS crossbar, no memory). */
  NMPC_D static S permuted(S v, int byte_index)
  {
    if constexpr(kF32)
    {
      return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_index, __float_as_int(v)));
    }
    else
    {
      const int lo = __builtin_amdgcn_ds_bpermute(byte_index, __double2loint(v));
      const int hi = __builtin_amdgcn_ds_bpermute(byte_index, __double2hiint(v));
      return __hiloint2double(hi, lo);
    }
  }

#ifdef NMPC_AMD_PROFILE_TILE64
  // profiling build (scripts/profile_tile64.py): shader-clock ticks of workgroup 0, by role, returned through qp_free.
  //  0 model wave: linearisation (its own work inside the sweeps)   1 model wave: waiting at the sweeps' barriers
  //  2 matrix wave 1: backward steps   3 matrix wave 1: waiting at the sweeps' barriers   4 matrix wave 1: steps run
  //  5 rolling lanes (wave 0, passes 0 / 1 / 3): compute   6 the same: waiting   7 matrix wave 1 as prefetcher: prefetch
  //  8 the same: waiting   9 sweeps   10 passes   11 matrix wave 5 (shares SIMD 0 with the model wave): backward steps
  NMPC_D void profAdd(int k, unsigned long long ticks, int who_wave) const
  {
    if(blockIdx.x == 0 && wave == who_wave && lane == 0)
    {
      reinterpret_cast<unsigned long long *>(lds + kProfAt)[k] += ticks;
    }
  }
  NMPC_D static unsigned long long profNow()
  {
    return __builtin_readcyclecounter();
  }
#else
  NMPC_D void profAdd(int, unsigned long long, int) const {}
  NMPC_D static unsigned long long profNow()
  {
    return 0;
  }
#endif

  /** X^T Y over the first 4 S rows of X and Y, formed from zero (k ascending: an fma chain per entry). */
  template<int S>
  NMPC_D static Vec4 mma(Vec4 X, Vec4 Y)
  {
    Vec4 acc = {0, 0, 0, 0};
#pragma unroll
    for(int s = 0; s < S; s++)
    {
      if constexpr(kF32)
      {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X[s], Y[s], acc, 0, 0, 0);
      }
      else
      {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[s], Y[s], acc, 0, 0, 0);
      }
    }
    return acc;
  }

  // tile-major addressing of the handle's arrays (ddp_kernels.hpp): element (half s, row r) of instance b
  NMPC_D static size_t tileOf(int b)
  {
    return static_cast<size_t>(b) / 64;
  }
  NMPC_D static size_t lnOf(int b)
  {
    return static_cast<size_t>(b) % 64;
  }
  NMPC_D Problem problemOf(int b) const
  {
    if constexpr(kOwnProblem)
    {
      return instanceProblem(problem, buf, b);
    }
    else
    {
      return problem;
    }
  }

  // ===================================================================================================
  // records: one visitor for the table and for the stores
  // ===================================================================================================
  /** Visits the record entries of the linearisation at (t, x, u) in canonical order. */
  template<class Sink>
  NMPC_D static void emitRecord(const Problem & p, S t, const StateDimVector & x, const InputDimVector & u, int m, Sink & sink)
  {
    StateStateDimMatrix Fx, Lxx;
    StateInputDimMatrix Fu, Lxu;
    StateDimVector Lx;
    InputDimVector Lu;
    InputInputDimMatrix Luu;
    if constexpr(kDyn)
    {
      // run-time input dimension m = inputDim(t) (u has m entries): the blocks' entries beyond m are stored as zeros — the
      // matrix waves then compute on m x m / m x n blocks padded with zeros, and skip the pivots >= m
      Fu.resize(N, m);
      Lxu.resize(N, m);
      Lu.resize(m);
      Luu.resize(m, m);
    }
    p.calcStateEqDeriv(t, x, u, Fx, Fu);
    p.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
    // Which entries are structural zeros must not depend on m (writer and readers share ONE offset table, and what the compiler
    // can prove about an entry changes with what it knows about m: with no input the total force of the centroidal problem is the
    // constant 0, with sixteen it is not).  So with a run-time input dimension the entries are CLASSIFIED on an evaluation of the
    // functors at the full dimension — of which nothing but what the compiler knows about its results survives (dead code, as the
    // table probe's arithmetic is) — and their VALUES come from the evaluation at m above.
    StateStateDimMatrix Fx_c, Lxx_c;
    StateInputDimMatrix Fu_c, Lxu_c;
    StateDimVector Lx_c;
    InputDimVector Lu_c;
    InputInputDimMatrix Luu_c;
    if constexpr(kDyn)
    {
      InputDimVector u_c;
      u_c.resize(MM);
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        u_c[a] = opaque(u[a]);
      }
      Fu_c.resize(N, MM);
      Lxu_c.resize(N, MM);
      Lu_c.resize(MM);
      Luu_c.resize(MM, MM);
      p.calcStateEqDeriv(t, x, u_c, Fx_c, Fu_c);
      p.calcRunningCostDeriv(t, x, u_c, Lx_c, Lu_c, Lxx_c, Luu_c, Lxu_c);
    }
#pragma unroll
    for(int c = 0; c < N; c++)
    {
#pragma unroll
      for(int r = 0; r < N; r++)
      {
        emitEntry(sink, kDyn ? Fx_c(r, c) : Fx(r, c), true, Fx(r, c));
      }
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
#pragma unroll
      for(int r = 0; r < N; r++)
      {
        emitEntry(sink, kDyn ? Fu_c(r, a) : Fu(r, a), a < m, Fu(r, a));
      }
    }
#pragma unroll
    for(int c = 0; c < N; c++)
    {
#pragma unroll
      for(int r = 0; r < N; r++)
      {
        emitEntry(sink, kDyn ? Lxx_c(r, c) : Lxx(r, c), true, Lxx(r, c));
      }
    }
#pragma unroll
    for(int c = 0; c < N; c++)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        emitEntry(sink, kDyn ? Lxu_c(c, a) : Lxu(c, a), a < m, Lxu(c, a));
      }
    }
#pragma unroll
    for(int c = 0; c < MM; c++)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        emitEntry(sink, kDyn ? Luu_c(a, c) : Luu(a, c), a < m && c < m, Luu(a, c));
      }
    }
#pragma unroll
    for(int r = 0; r < N; r++)
    {
      emitEntry(sink, kDyn ? Lx_c[r] : Lx[r], true, Lx[r]);
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      emitEntry(sink, kDyn ? Lu_c[a] : Lu[a], a < m, Lu[a]);
    }
    S un = 0;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      un += dynEntry(a < m, u[a] * u[a]);
    }
    const S unorm = (M == 1) ? fabs(u[0]) : sqrt(un);
    sink.putVar(recipFast(unorm + S(1)));
    if constexpr(kConstrained)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        sink.putVar(u[a]);
      }
    }
    if constexpr(kDyn)
    {
      sink.putVar(static_cast<S>(m));
    }
  }
  /** One entry to the sink: static input dimension — the entry as it is (a structural zero is not stored, a constant once per
      sweep); run-time input dimension — classified by `canonical` (see emitRecord), stored every time, zero outside the dimension. */
  template<class Sink>
  NMPC_D static void emitEntry(Sink & sink, S canonical, bool inside, S v)
  {
    if constexpr(kDyn)
    {
      sink.putClassified(canonical, inside ? v : 0.0);
    }
    else
    {
      (void)canonical;
      (void)inside;
      sink.put(v);
    }
  }
  /** A value the compiler knows nothing about. */
  NMPC_D static S opaque(S v)
  {
    asm("" : "+v"(v));
    return v;
  }
  /** An entry of a block whose size follows the run-time input dimension: zero outside (static dimension: the entry as it is —
      a structural zero stays one). */
  NMPC_D static S dynEntry(bool inside, S v)
  {
    if constexpr(kDyn)
    {
      return inside ? v : 0.0;
    }
    else
    {
      return v;
    }
  }
  /** A structural zero: the compiler knows the value, and it is zero. */
  NMPC_D static bool structuralZero(S v)
  {
    return __builtin_constant_p(v) && v == 0.0;
  }
  struct TableSink
  {
    unsigned short * tbl;
    int id = 0, cnt = 0;
    NMPC_D void put(S v)
    {
      if(structuralZero(v))
      {
        tbl[id++] = 0;
      }
      else
      {
        tbl[id++] = static_cast<unsigned short>(++cnt);
      }
    }
    NMPC_D void putVar(S)
    {
      tbl[id++] = static_cast<unsigned short>(++cnt);
    }
    NMPC_D void putClassified(S canonical, S)
    {
      put(canonical);
    }
  };
  /** full = false: entries the compiler knows to be constants (the literal ones of the Jacobians, weights of a shared
      problem object) are not written again — the record slot holds them from the sweep's first two (full) timesteps.  A RUN-TIME
      flag of ONE instantiation: as a template parameter the functors' code existed twice, and the compiler contracted their
      products into fused multiply-adds differently in the two copies — in float the records of the full and of the partial
      timesteps then differed in the last bit, and with them the results with the chunk size, i.e. with the group size
      (measured: 7e-7 between groups of 7 and of 32; in double the two copies happened to agree). */
  template<bool kFullT>
  struct StoreSink
  {
    S * rec;
    bool full;
    int cnt = 0;
    NMPC_D void put(S v)
    {
      if(!structuralZero(v))
      {
        ++cnt;
        if(!__builtin_constant_p(v) || kFullT || full)
        {
          rec[cnt] = v;
        }
      }
    }
    NMPC_D void putVar(S v)
    {
      rec[++cnt] = v;
    }
    NMPC_D void putClassified(S canonical, S v)
    {
      if(!structuralZero(canonical))
      {
        rec[++cnt] = v;
      }
    }
  };

  struct Point
  {
    S x[N], u[MM];
  };
  /** (x_i, u_i) of the slot's current trajectory: requested one timestep before lineariseStep consumes it. */
  NMPC_D void loadPoint(Point & p, int b, int sel, int i) const
  {
    const size_t tile = tileOf(b), ln = lnOf(b);
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
    const S * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
    const S * Un = buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln;
#pragma unroll
    for(int c = 0; c < N; c++)
    {
      p.x[c] = Xn[(static_cast<size_t>(i) * N + c) * 64];
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      p.u[a] = Un[(static_cast<size_t>(i) * MM + a) * 64];
    }
  }
  template<bool kFullT>
  NMPC_D void lineariseStep(const Problem & mine, S * dst, S t0, int i, const Point & p, bool full) const
  {
    StateDimVector x;
    InputDimVector u;
#pragma unroll
    for(int c = 0; c < N; c++)
    {
      x[c] = p.x[c];
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      u[a] = p.u[a];
    }
    StoreSink<kFullT> sink{dst, full};
    if(kFullT || full)
    {
      sink.rec[0] = 0.0; // the zero word
    }
    const S t = t0 + i * mine.dt();
    const int m = inputDimOf(mine, t);
    if constexpr(kDyn)
    {
      u.resize(m);
    }
    emitRecord(mine, t, x, u, m, sink);
  }
  /** inputDim(t) of a run-time input dimension (DDPSolver.hpp:381), the static one otherwise. */
  NMPC_D static int inputDimOf(const Problem & mine, S t)
  {
    if constexpr(kDyn)
    {
      return mine.inputDim(t);
    }
    else
    {
      return MM;
    }
  }
  /** [Vxx | Vx] of the terminal cost (:177-185, :346-365) -> term(slot), column-major n x (n + 1). */
  NMPC_D void lineariseTerminal(const Problem & mine, int slot, int b, int sel, S t0) const
  {
    const size_t tile = tileOf(b), ln = lnOf(b);
    const size_t rows_x = static_cast<size_t>(T + 1) * N;
    const S * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
    StateDimVector xT, vx;
    StateStateDimMatrix vxx;
#pragma unroll
    for(int c = 0; c < N; c++)
    {
      xT[c] = Xn[(static_cast<size_t>(T) * N + c) * 64];
    }
    mine.calcTerminalCostDeriv(t0 + T * mine.dt(), xT, vx, vxx);
    S * r = term(slot);
#pragma unroll
    for(int c = 0; c <= N; c++)
    {
#pragma unroll
      for(int k = 0; k < N; k++)
      {
        r[c * N + k] = (c < N) ? vxx(k, c < N ? c : 0) : vx[k];
      }
    }
  }

  // ===================================================================================================
  // line search: rollouts fed from an LDS ring of nominal records    DDPSolver.hpp:234-274, forwardPass :536-560
  // ===================================================================================================
  // A forward pass needs k_i, K_i, x_i, u_i of its instance per timestep: n + 2 m + m n doubles (126 for the manipulator).
  // Loaded by the rolling lane itself they do not fit its registers next to the model code (measured: every load spilled behind
  // a vmcnt(0), 80 k cycles per timestep against 8 k for the initial rollout).  So the waves that do not roll out in a pass
  // fetch the nominal of timestep i + 2 into a ring in LDS (the record area, idle during the line search) — coalesced rows of
  // the tile-major arrays, element (row, slot) — while the rolling waves read timestep i from it, one column of K at a time.
  // One barrier per timestep; both roles run stagedPass() with the same barrier count.
  NMPC_D S * ring(int i) const
  {
    return lds + kRecAt + (i % kRingDepth) * (kRingStride * G);
  }
  /** What a prefetching lane keeps for a whole pass: it serves ONE slot (p_lane % G) and every (p_count / G)-th row of it, so
      everything that depends on the slot is computed once. */
  //! A pointer into the handle's HBM arrays, said so: the prefetch loads must be global_load (vmcnt), not flat_load.  The compiler
  //! does not see through the selects and arrays these pointers pass and fell back to flat_load — whose completion a wait for the
  //! LDS (lgkmcnt) also waits for, and in front of which it may put a wait of its own: round 6 found one build of this file with
  //! s_waitcnt vmcnt(0) lgkmcnt(0) in front of EVERY load of prefetchIssue, i.e. nineteen sequential round trips to L2 per trip
  //! (centroidal forward passes 6.5 -> 10.4 ms); round 4's unexplained "register allocation side effect" (HISTORY) has the same signature.
  using GlobalPtr = const S __attribute__((address_space(1))) *;
  NMPC_D static GlobalPtr asGlobal(const S * p)
  {
    return (GlobalPtr)p;
  }
  struct PrefetchLane
  {
    bool want; //!< the slot takes part in the pass and this lane has rows to fetch
    int row0, row_step, slot;
    GlobalPtr pk, pK, pX, pU; //!< row 0 of timestep 0 of k_list_, K_list_, x_list, u_list of the slot's instance
  };
  NMPC_D PrefetchLane makePrefetchLane(int group, int p_lane, int p_count) const
  {
    PrefetchLane pl;
    const int lanes_per_row_set = p_count / G; // (>= 1: p_count >= 64 >= G)
    pl.slot = p_lane % G;
    pl.row0 = p_lane / G;
    pl.row_step = lanes_per_row_set;
    pl.want = pl.row0 < lanes_per_row_set && slotI(sLs, pl.slot) != 0;
    const int b = group * G + pl.slot;
    const int sel = slotI(sSel, pl.slot);
    const size_t tile = pl.want ? tileOf(b) : 0, ln = pl.want ? lnOf(b) : 0;
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
    pl.pk = asGlobal(buf.wpi_ws + static_cast<size_t>(pl.want ? b : 0) * gainDoubles(T)); // gain record of timestep 0
    pl.pK = pl.pk + MM;
    pl.pX = asGlobal(buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln);
    pl.pU = asGlobal(buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln);
    return pl;
  }
  /** Prefetch role: this lane's rows of timestep i -> ring(i): element (row, slot) at slot * kRingStride + row. */
  NMPC_D void prefetchNominal(const PrefetchLane & pl, int i) const
  {
    S * dst = ring(i) + pl.slot * kRingStride;
    constexpr int kBatch = (kRingRows + 13) / 14 < 10 ? (kRingRows + 13) / 14 : 10; // loads in flight per lane: a whole timestep's
                                                                                    // share when seven waves prefetch 32 slots
    for(int r0 = pl.row0; r0 < kRingRows; r0 += pl.row_step * kBatch)
    {
      S v[kBatch];
      int at[kBatch];
#pragma unroll
      for(int k = 0; k < kBatch; k++)
      {
        const int row = r0 + k * pl.row_step;
        const bool ok = pl.want && row < kRingRows;
        // row < m: k_i | < m + m n: K_i | < m + m n + n: x_i | else u_i
        const bool is_k = row < MM, is_K = !is_k && row < kGainRows, is_x = !is_k && !is_K && row < kGainRows + N;
        GlobalPtr base = is_k ? pl.pk : (is_K ? pl.pK : (is_x ? pl.pX : pl.pU));
        const int per_step = (is_k || is_K) ? kGainRows : (is_x ? N * 64 : MM * 64); // gains: records; x, u: tile-major rows
        const int r = is_k ? row : (is_K ? row - MM : (is_x ? (row - kGainRows) * 64 : (row - kGainRows - N) * 64));
        at[k] = ok ? row : -1;
        v[k] = 0;
        if(ok)
        {
          v[k] = base[static_cast<size_t>(i) * per_step + r];
        }
      }
#pragma unroll
      for(int k = 0; k < kBatch; k++)
      {
        if(at[k] >= 0)
        {
          dst[at[k]] = v[k];
        }
      }
    }
  }
  /** x_i, u_i, cost_i of a rollout to rows kStride doubles apart. */
  template<int kStride>
  NMPC_D static void storeTimestep(S * Xo, S * Uo, S * Co, int i, const StateDimVector & x, const S * u, S c, bool with_u)
  {
    S * xr = Xo + static_cast<size_t>(i) * (N * kStride);
#pragma unroll
    for(int cc = 0; cc < N; cc++)
    {
      xr[cc * kStride] = x[cc];
    }
    if(with_u)
    {
      S * ur = Uo + static_cast<size_t>(i) * (MM * kStride);
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        ur[a * kStride] = u[a];
      }
    }
    Co[static_cast<size_t>(i) * kStride] = c;
  }
  /** runningCost and stateEq at (x, u[0 .. m)). */
  NMPC_D static void evalModel(const Problem & mine, S t, const StateDimVector & x, const S * uv, int m, S & c,
                               StateDimVector & x_next)
  {
    InputDimVector u;
    if constexpr(kDyn)
    {
      u.resize(m);
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      u[a] = uv[a];
    }
    c = mine.runningCost(t, x, u);
    x_next = mine.stateEq(t, x, u);
  }
  // The same in two halves, a trip of stagedPass apart: requested in trip j - 1, a timestep's rows are written to the ring in trip
  // j — a prefetching lane never waits for its loads INSIDE a trip.  With request and write in one trip no trip was shorter than
  // a round trip to L2 / HBM (2 - 2.5 us), whatever the rolling lanes had to do: the quadrotor's rollout timestep is ~500
  // instructions, its trips took 6 k cycles.  For the lanes' shares of a timestep that fit kPipeRows registers (seven waves
  // serving a group: always); a lone prefetching wave keeps prefetchNominal.
  //! rows of a timestep per prefetching lane when six waves serve a full group (row_step >= 384 / 35 = 10)
  static constexpr int kPipeRows = (kRingRows + 9) / 10;
  struct PrefetchRegs
  {
    S v[kPipeRows]; //!< the rows in flight
    GlobalPtr src[kPipeRows]; //!< row k of this lane at timestep 0 (a valid address also where the lane has no row k) ...
    int step[kPipeRows]; //!< ... and the distance to the same row of the next timestep, in doubles
    int at[kPipeRows]; //!< where it goes in the slot's ring entry (-1: nowhere)
  };
  NMPC_D bool prefetchPipelined(const PrefetchLane & pl) const
  {
    return pl.row_step * kPipeRows >= kRingRows; // (wave-uniform: row_step = lanes that serve a slot)
  }
  /** Once per pass: which rows this lane fetches, from where.  The trips then cost a multiply-add, a load and a store per row. */
  NMPC_D void prefetchPlan(const PrefetchLane & pl, PrefetchRegs & pr) const
  {
#pragma unroll
    for(int k = 0; k < kPipeRows; k++)
    {
      const int row = pl.row0 + k * pl.row_step;
      const bool ok = pl.want && row < kRingRows;
      // row < m: k_i | < m + m n: K_i | < m + m n + n: x_i | else u_i
      const bool is_k = row < MM, is_K = !is_k && row < kGainRows, is_x = !is_k && !is_K && row < kGainRows + N;
      GlobalPtr base = is_k ? pl.pk : (is_K ? pl.pK : (is_x ? pl.pX : pl.pU));
      const int per_step = (is_k || is_K) ? kGainRows : (is_x ? N * 64 : MM * 64); // gains: records; x, u: tile-major rows
      const int r = is_k ? row : (is_K ? row - MM : (is_x ? (row - kGainRows) * 64 : (row - kGainRows - N) * 64));
      pr.src[k] = ok ? base + r : pl.pk; // (pk: the slot's — or instance 0's — first gain record: always readable)
      pr.step[k] = ok ? per_step : 0;
      pr.at[k] = ok ? row : -1;
      pr.v[k] = 0;
    }
  }
  /** n_rows: rows per lane at most in this pass (wave-uniform: ceil(kRingRows / row_step)) — one when a whole workgroup's
      prefetching waves serve a single slot. */
  NMPC_D void prefetchIssue(int i, PrefetchRegs & pr, int n_rows) const
  {
#pragma unroll
    for(int k = 0; k < kPipeRows; k++)
    {
      if(k < n_rows)
      {
        pr.v[k] = pr.src[k][static_cast<size_t>(i) * pr.step[k]]; // (no per-lane branch around a load: lanes without row k re-read a valid word)
      }
    }
  }
  NMPC_D void prefetchCommit(const PrefetchLane & pl, int i, const PrefetchRegs & pr, int n_rows) const
  {
    S * dst = ring(i) + pl.slot * kRingStride;
#pragma unroll
    for(int k = 0; k < kPipeRows; k++)
    {
      if(k < n_rows && pr.at[k] >= 0)
      {
        dst[pr.at[k]] = pr.v[k];
      }
    }
  }
  /** One pass over the horizon, every wave of the workgroup together (T + 2 barriers).
      compute = false: this lane prefetches for the ring (p_lane of p_count; nothing to do in an initial pass).
      compute = true:  every active lane rolls out one trajectory of slot `inst` (instance b) and sums the cost in list order:
        initial  — u_i = initial_u_list[i] (half 0 of U), x_0 = current_x    :83-95
        else     — u'_i = (u_i + alpha k_i) + K_i (x'_i - x_i) around the nominal in the ring    :536-560
      store: the trajectory goes to half out_half of X / U / cost (an initial pass leaves U as it is).
      Trip j of the loop: the prefetchers fetch timestep j, the rolling lanes compute timestep j - 2 (ring depth 3). */
  NMPC_D S stagedPass(bool compute, bool initial, const Problem & mine, bool active, int group, int b, int inst, int out_half,
                           S t0, S alpha, bool store, int p_lane, int p_count, int to_candidate = -1) const
  {
    S J = 0;
    StateDimVector x;
    S *Xo = nullptr, *Uo = nullptr, *Co = nullptr;
    const S * Uin = nullptr;
    const bool rolling = compute && active;
    const PrefetchLane pl = makePrefetchLane(group, p_lane, p_count);
    size_t ost = 64; // distance of consecutive rows at the destination (the handle's arrays are tile-major)
    if(rolling)
    {
      const size_t tile = tileOf(b), ln = lnOf(b);
      const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM, rows_c = static_cast<size_t>(T + 1);
      Xo = buf.X + ((tile * 2 + out_half) * rows_x) * 64 + ln;
      Uo = buf.U + ((tile * 2 + out_half) * rows_u) * 64 + ln;
      Co = buf.cost + ((tile * 2 + out_half) * rows_c) * 64 + ln;
      if(to_candidate >= 0)
      {
        Xo = candidate(b, to_candidate); // (rows contiguous: one run per lane and timestep)
        Uo = Xo + rows_x;
        Co = Uo + rows_u;
        ost = 1;
      }
      Uin = buf.U + ((tile * 2 + 0) * rows_u) * 64 + ln;
      if(initial)
      {
#pragma unroll
        for(int c = 0; c < N; c++)
        {
          x[c] = buf.x0[(tile * N + c) * 64 + ln];
        }
      }
    }
    profAdd(10, 1, 0);
    if(!compute)
    {
      // The prefetching waves' trips: a loop of their own, so that what a prefetching lane keeps across its trips (its plan,
      // its rows in flight) is not live in the rolling lanes' code (one loop for both roles: the rollout spilled).
      PrefetchRegs pr;
      const bool piped = !initial && prefetchPipelined(pl);
      const int n_rows = uniform((kRingRows + pl.row_step - 1) / pl.row_step);
      if(piped)
      {
        prefetchPlan(pl, pr);
        prefetchIssue(0, pr, n_rows); // (T >= 1)
      }
#pragma nounroll
      for(int j = 0; j < T + 2; j++)
      {
        const unsigned long long pa = profNow();
        if(!initial && j < T)
        {
          if(piped)
          {
            prefetchCommit(pl, j, pr, n_rows);
            if(j + 1 < T)
            {
              prefetchIssue(j + 1, pr, n_rows);
            }
          }
          else
          {
            prefetchNominal(pl, j);
          }
        }
        const unsigned long long pb = profNow();
        profAdd(7, pb - pa, 1);
        barrier();
        profAdd(8, profNow() - pb, 1);
      }
      return 0.0;
    }
#pragma nounroll
    for(int j = 0; j < T + 2; j++)
    {
      const unsigned long long pa = profNow();
      if(active && j >= 2)
      {
        const int i = j - 2;
        const S t = t0 + i * mine.dt();
        const int m = inputDimOf(mine, t); // (run-time input dimension: the rows beyond it are zeros, in U and in the gains)
        S u[MM];
        if(initial)
        {
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            u[a] = dynEntry(a < m, Uin[(static_cast<size_t>(i) * MM + a) * 64]);
          }
        }
        else
        {
          const S * R = ring(i) + inst * kRingStride;
          if(i == 0)
          {
#pragma unroll
            for(int c = 0; c < N; c++)
            {
              x[c] = R[kGainRows + c]; // x'_0 = x_0    :541
            }
          }
          S s[MM];
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            u[a] = R[kGainRows + N + a] + alpha * R[a]; // u_i + alpha k_i    :545
            s[a] = 0;
          }
#pragma unroll
          for(int c = 0; c < N; c++)
          {
            const S dxc = x[c] - R[kGainRows + c];
#pragma unroll
            for(int a = 0; a < MM; a++)
            {
              s[a] += R[MM + a + c * MM] * dxc;
            }
            if(MM * N > 64 && (c & 1) == 1)
            {
              __builtin_amdgcn_sched_barrier(0); // large gain blocks: two columns of K in registers at a time, not all of it
            }
          }
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            u[a] = dynEntry(a < m, u[a] + s[a]); // ... + K_i (x'_i - x_i)    :546
          }
        }
        // the problem's functors see an input of inputDim(t) entries.  A run-time dimension that is the full one, or none, is
        // given to them as a CONSTANT (three copies of their code): loops over u.size() then unroll, and what they index
        // stays in registers — with a run-time trip count the centroidal problem's stance tables went to private memory,
        // 15 k cycles per rollout timestep
        S c;
        StateDimVector x_next;
        if constexpr(kDyn)
        {
          if(m == MM)
          {
            evalModel(mine, t, x, u, MM, c, x_next);
          }
          else if(m == 0)
          {
            evalModel(mine, t, x, u, 0, c, x_next);
          }
          else
          {
            evalModel(mine, t, x, u, m, c, x_next);
          }
        }
        else
        {
          evalModel(mine, t, x, u, MM, c, x_next);
        }
        if(store)
        {
          // (the distance of consecutive rows as a constant in both cases: immediate offsets)
          if(ost == 1)
          {
            storeTimestep<1>(Xo, Uo, Co, i, x, u, c, !initial || kDyn);
          }
          else
          {
            storeTimestep<64>(Xo, Uo, Co, i, x, u, c, !initial || kDyn); // (run-time input dimension: the initial pass zeroes
                                                                          // the rows of U beyond inputDim(t))
          }
        }
        J += c;
        x = x_next;
      }
      const unsigned long long pb = profNow();
      profAdd(5, pb - pa, 0);
      barrier();
      profAdd(6, profNow() - pb, 0);
    }
    if(rolling)
    {
      const S cT = mine.terminalCost(t0 + T * mine.dt(), x);
      if(store)
      {
#pragma unroll
        for(int cc = 0; cc < N; cc++)
        {
          Xo[(static_cast<size_t>(T) * N + cc) * ost] = x[cc];
        }
        Co[static_cast<size_t>(T) * ost] = cT;
      }
      J += cT;
    }
    return J;
  }
  /** The slots whose line search took a LATER step size (sLs set, sAi its index): its trajectory — left in the workspace by the
      lane that rolled it out for its cost — goes to the other half of X / U / cost, where a re-roll (pass 3) would have put
      the same bits.  Every wave of the workgroup.  Few slots: a slot at a time, thread = row (coalesced reads); many: lane =
      slot, a wave per eighth of the rows, eight loads in flight (the writes are then whole runs of the tile-major arrays). */
  NMPC_D void adoptCandidates(int group) const
  {
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM, rows_c = static_cast<size_t>(T + 1);
    const size_t rows = rows_x + rows_u + rows_c;
    int n_take = 0;
    for(int k = 0; k < G; k++)
    {
      n_take += (slotI(sB, k) >= 0 && slotI(sLs, k) != 0) ? 1 : 0;
    }
    n_take = uniform(n_take);
    auto destination = [&](int b, int half, size_t r) -> S *
    {
      const size_t tile = tileOf(b), ln = lnOf(b);
      if(r < rows_x)
      {
        return buf.X + ((tile * 2 + half) * rows_x + r) * 64 + ln;
      }
      if(r < rows_x + rows_u)
      {
        return buf.U + ((tile * 2 + half) * rows_u + (r - rows_x)) * 64 + ln;
      }
      return buf.cost + ((tile * 2 + half) * rows_c + (r - rows_x - rows_u)) * 64 + ln;
    };
    if(n_take <= 4)
    {
      for(int k = 0; k < G; k++)
      {
        const int b = uniform(slotI(sB, k));
        if(b < 0 || uniform(slotI(sLs, k)) == 0)
        {
          continue;
        }
        const int half = uniform(slotI(sSel, k)) ^ 1;
        const S * src = candidate(b, uniform(slotI(sAi, k)) - 1);
        for(size_t r = threadIdx.x; r < rows; r += kT64Threads)
        {
          *destination(b, half, r) = src[r];
        }
      }
    }
    else
    {
      const int k = lane < kT64MaxGroup ? lane : 0;
      const bool mine = lane < kT64MaxGroup && k < G && slotI(sB, k) >= 0 && slotI(sLs, k) != 0;
      const int b = mine ? slotI(sB, k) : 0;
      const int half = mine ? (slotI(sSel, k) ^ 1) : 0;
      const S * src = candidate(b, mine ? slotI(sAi, k) - 1 : 0);
      constexpr int kInFlight = 8;
      for(size_t r0 = static_cast<size_t>(wave); r0 < rows; r0 += static_cast<size_t>(kT64Waves) * kInFlight)
      {
        S v[kInFlight];
#pragma unroll
        for(int q = 0; q < kInFlight; q++)
        {
          const size_t r = r0 + static_cast<size_t>(q) * kT64Waves;
          v[q] = (mine && r < rows) ? src[r] : 0.0;
        }
#pragma unroll
        for(int q = 0; q < kInFlight; q++)
        {
          const size_t r = r0 + static_cast<size_t>(q) * kT64Waves;
          if(mine && r < rows)
          {
            *destination(b, half, r) = v[q];
          }
        }
      }
    }
  }
  /** Trips of the second pass (later step sizes, lane = (slot, step size) on the matrix waves): slots covered per trip. */
  NMPC_D int laterPerWave() const
  {
    const int n_later = cfg.n_alpha - 1;
    return n_later > 0 ? 64 / n_later : 64; // >= 2 (NMPC_HIP_MAX_ALPHA = 32)
  }

  // ===================================================================================================
  // matrix waves: one backward timestep of one instance    DDPSolver.hpp:381-530
  // ===================================================================================================
  /** What a matrix lane knows for the whole kernel: where its entries of a record are (offsets in doubles). */
  //! kPackMap (the batched natural-layout gains): the offsets live PACKED two per register (they are below 2^16: a record has a few
  //! hundred entries) and the symmetrisation weights are derived from the lane id where they are used — twenty registers less across the
  //! sweep.  The batched gains hold the columns of a whole factorisation (64 registers) and two slots' stashes beside the step's
  //! own working set; with the plain map the trips spilled a handful of values, and every reload waits (vmcnt(0)) for the gain
  //! stores of the trip in front of it — a round trip to HBM per slot and timestep.
  static constexpr bool kPackMap = kBatchGains;
  struct LaneMap
  {
    int oFx[kPackMap ? 1 : 4], oFu[kPackMap ? 1 : 4], oLxx[kPackMap ? 1 : 4], oLxuT[kPackMap ? 1 : KM], oLuu[kPackMap ? 1 : KM], oLx, oLu, oInvU,
        oU[kConstrained ? MM : 1], oM;
    S wn, wt; //!< weights of (Vn, Vn^T) in the new [Vxx | Vx]: (1/2, 1/2) inside the n x n block, (1, 0) in column n
    unsigned pFxFu[kPackMap ? 4 : 1], pLxxLxuT[kPackMap ? 4 : 1], pLuu[kPackMap ? 2 : 1], pLxLu; // (kPackMap: low | high << 16)
    NMPC_D int fx(int r) const
    {
      return kPackMap ? static_cast<int>(pFxFu[kPackMap ? r : 0] & 0xffffu) : oFx[kPackMap ? 0 : r];
    }
    NMPC_D int fu(int r) const
    {
      return kPackMap ? static_cast<int>(pFxFu[kPackMap ? r : 0] >> 16) : oFu[kPackMap ? 0 : r];
    }
    NMPC_D int lxx(int r) const
    {
      return kPackMap ? static_cast<int>(pLxxLxuT[kPackMap ? r : 0] & 0xffffu) : oLxx[kPackMap ? 0 : r];
    }
    NMPC_D int lxuT(int r) const
    {
      return kPackMap ? static_cast<int>(pLxxLxuT[kPackMap ? r : 0] >> 16) : oLxuT[kPackMap ? 0 : r];
    }
    NMPC_D int luu(int r) const
    {
      return kPackMap ? static_cast<int>((r & 1) ? pLuu[kPackMap ? r >> 1 : 0] >> 16 : pLuu[kPackMap ? r >> 1 : 0] & 0xffffu) : oLuu[kPackMap ? 0 : r];
    }
    NMPC_D int lx() const
    {
      return kPackMap ? static_cast<int>(pLxLu & 0xffffu) : oLx;
    }
    NMPC_D int lu() const
    {
      return kPackMap ? static_cast<int>(pLxLu >> 16) : oLu;
    }
  };
  NMPC_D LaneMap makeLaneMap() const
  {
    const int q = lane >> 4, j = colOf(lane & 15);
    const unsigned short * t = tbl();
    LaneMap mp;
    int oFx[4], oFu[4], oLxx[4], oLxuT[4] = {0, 0, 0, 0}, oLuu[4] = {0, 0, 0, 0};
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int row = 4 * r + q;
      if constexpr(kAug)
      {
        // oFx: the augmented F = [Fx Fu]; oLxx: the augmented L = [[Lxx Lxu],[Lxu^T Luu]]
        const int ju = (j >= N && j < NA) ? j - N : 0, ru = (row >= N && row < NA) ? row - N : 0, jx = j < N ? j : 0, rx = row < N ? row : 0;
        oFx[r] = (row < N) ? ((j < N) ? t[idFx + jx * N + rx] : ((j < NA) ? t[idFu + ju * N + rx] : 0)) : 0;
        int l = 0;
        if(row < N && j < N)
        {
          l = t[idLxx + jx * N + rx];
        }
        else if(row < N && j < NA)
        {
          l = t[idLxuT + rx * MM + ju]; // Lxu(row, j - n)
        }
        else if(row < NA && j < N)
        {
          l = t[idLxuT + jx * MM + ru]; // Lxu(j, row - n)
        }
        else if(row < NA && j < NA)
        {
          l = t[idLuu + ju * MM + ru];
        }
        oLxx[r] = l;
        oFu[r] = 0;
      }
      else
      {
        oFx[r] = (row < N && j < N) ? t[idFx + j * N + row] : 0;
        oFu[r] = (row < N && j < MM) ? t[idFu + j * N + row] : 0;
        oLxx[r] = (row < N && j < N) ? t[idLxx + j * N + row] : 0;
      }
    }
#pragma unroll
    for(int r = 0; r < KM; r++)
    {
      const int a = 4 * r + q;
      oLxuT[r] = (a < MM && j < N) ? t[idLxuT + j * MM + a] : 0;
      oLuu[r] = (a < MM && j < MM) ? t[idLuu + j * MM + a] : 0;
    }
    const int oLx_ = (j < N) ? t[idLx + j] : ((kAug && j < NA) ? t[idLu + ((j >= N && j < NA) ? j - N : 0)] : 0); // (augmented: [Lx; Lu])
    const int oLu_ = (j < MM) ? t[idLu + j] : 0;
    mp.oLx = oLx_;
    mp.oLu = oLu_;
    if constexpr(kPackMap)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        mp.pFxFu[r] = static_cast<unsigned>(oFx[r]) | (static_cast<unsigned>(oFu[r]) << 16);
        mp.pLxxLxuT[r] = static_cast<unsigned>(oLxx[r]) | (static_cast<unsigned>(oLxuT[r]) << 16);
      }
      mp.pLuu[0] = static_cast<unsigned>(oLuu[0]) | (static_cast<unsigned>(oLuu[1]) << 16);
      mp.pLuu[1] = static_cast<unsigned>(oLuu[2]) | (static_cast<unsigned>(oLuu[3]) << 16);
      mp.pLxLu = static_cast<unsigned>(oLx_) | (static_cast<unsigned>(oLu_) << 16);
      mp.oFx[0] = mp.oFu[0] = mp.oLxx[0] = mp.oLxuT[0] = mp.oLuu[0] = 0;
    }
    else
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        mp.oFx[kPackMap ? 0 : r] = oFx[r];
        mp.oFu[kPackMap ? 0 : r] = oFu[r];
        mp.oLxx[kPackMap ? 0 : r] = oLxx[r];
      }
#pragma unroll
      for(int r = 0; r < KM; r++)
      {
        mp.oLxuT[kPackMap ? 0 : r] = oLxuT[r];
        mp.oLuu[kPackMap ? 0 : r] = oLuu[r];
      }
      mp.pFxFu[0] = mp.pLxxLxuT[0] = mp.pLuu[0] = mp.pLxLu = 0;
    }
    mp.oInvU = uniform(t[idInvU]);
    if constexpr(kConstrained)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        mp.oU[a] = t[idU + a];
      }
    }
    else
    {
      mp.oU[0] = 0;
    }
    mp.oM = kDyn ? uniform(t[kDyn ? idM : 0]) : 0;
    mp.wn = (j < N) ? 0.5 : ((j == N) ? 1.0 : 0.0);
    mp.wt = (j < N) ? 0.5 : 0.0;
    return mp;
  }

  /** In-place L D L^T of the m x m matrix A (column-major, leading dimension MM, lower triangle) with the pivot rule of Eigen's
      LLT: fails iff a pivot is <= 0, NaN passes.  The products L_kj d_j of a pivot row are formed once
      (m^3 / 6 multiply-adds instead of the m^3 / 3 of the lane kernels' ldltInPlace, whose (L L) d association this differs from
      by rounding); branch-free: after a failed pivot the factor is garbage and the caller stores nothing. */
  NMPC_D static bool ldlt(S * A, S * inv_d)
  {
    bool ok = true;
#pragma unroll
    for(int k = 0; k < MM; k++)
    {
      S v[MM];
      S d = A[k + k * MM];
#pragma unroll
      for(int j = 0; j < k; j++)
      {
        v[j] = A[k + j * MM] * A[j + j * MM];
      }
#pragma unroll
      for(int j = 0; j < k; j++)
      {
        d -= A[k + j * MM] * v[j];
      }
      ok = ok && !(d <= 0.0);
      A[k + k * MM] = d;
      const S r = recipFast(d);
      inv_d[k] = r;
#pragma unroll
      for(int i = k + 1; i < MM; i++)
      {
        S s = A[i + k * MM];
#pragma unroll
        for(int j = 0; j < k; j++)
        {
          s -= A[i + j * MM] * v[j];
        }
        A[i + k * MM] = s * r;
      }
    }
    return ok;
  }
  /** (L D L^T) x = rhs in place. */
  NMPC_D static void ldltSolve(const S * A, const S * inv_d, S * x)
  {
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      S s = x[i];
#pragma unroll
      for(int j = 0; j < i; j++)
      {
        s -= A[i + j * MM] * x[j];
      }
      x[i] = s;
    }
#pragma unroll
    for(int ii = 0; ii < MM; ii++)
    {
      const int i = MM - 1 - ii;
      S s = x[i] * inv_d[i];
#pragma unroll
      for(int j = i + 1; j < MM; j++)
      {
        s -= A[j + i * MM] * x[j];
      }
      x[i] = s;
    }
  }

  /** The value of lane j + n of this lane's 16-lane row (whatever another lane holds when j + n >= 16: callers mask). */
  NMPC_D S fromColumnPlusN(S v) const
  {
    const int src = ((lane & 48) | laneOfCol((colOf(lane & 15) + N) & 15)) << 2;
    return permuted(v, src);
  }

  /** Row `sel` (0..3) of four values. */
  NMPC_D static S pick4(int sel, S a0, S a1, S a2, S a3)
  {
    const S lo = (sel & 1) ? a1 : a0;
    const S hi = (sel & 1) ? a3 : a2;
    return (sel & 2) ? hi : lo;
  }

  /** Registers of one instance while its timestep is processed.  The step is written as PHASES (below) so that a wave can
      run the phases of two instances back to back: between two fences the compiler sees two independent chains. */
  struct StepCtx
  {
    Vec4 VV; //!< [Vxx | Vx] in natural layout (in / out)
    Vec4 Qxx, Qux, Quu, QuxR, QuuF, qxcol, A, Vn;
    S qxrow, qurow, inv_u;
    S fac[kBig ? 1 : MM * MM], inv_d[kBig ? 1 : MM], col[kBig ? 1 : MM], colQ[kBig ? 1 : MM]; // (per-lane factorisation, m <= 8)
    S c1nn, t2nn, krel_i; //!< k^T Quu k, k^T Qu, |k| / (|u| + 1) of this timestep (lane kStarLane)
    bool ok; //!< no factorisation of this sweep has failed yet (in / out)
  };

  /** Phase 1a: the record's operands, the Q terms (:386-408), Quu_F / Qux_reg as if unregularised.  Branch-free. */
  NMPC_D void stepQTerms(StepCtx & c, const LaneMap & mp, const S * r, Vec4 & F0, Vec4 & F1, Vec4 & L0, Vec4 & L1, Vec4 & L2) const
  {
    const int j = colOf(lane & 15);
    (void)j;
    c.Qux = Vec4{0, 0, 0, 0};
    c.Quu = Vec4{0, 0, 0, 0};
    if constexpr(kAug)
    {
      Vec4 F, L;
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        F[rr] = r[mp.fx(rr)];
        L[rr] = r[mp.lxx(rr)];
      }
      const S lv = r[mp.lx()];
      c.inv_u = r[mp.oInvU];
      // G = VV^T F: rows < n: Vxx [Fx Fu], row n: Vx^T [Fx Fu];  Q = G^T F + L = [[Qxx Qxu],[Qux Quu]] (row n of G meets the
      // zero row n of F).  Same products in the same order as the block form below: (Fu^T Vxx) Fx etc., left to right.
      const Vec4 Gm = mma<KN>(c.VV, F);
      Vec4 Q = mma<KN>(Gm, F);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        Q[rr] = L[rr] + Q[rr];
      }
      const S qrow = lv + Gm[rN]; // lane group 0: Qx[j] (j < n), Qu[j - n] (n <= j < n + m)
      c.qxrow = qrow;
      c.qurow = fromColumnPlusN(qrow);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        c.Qxx[rr] = (rr < rN) ? Q[rr] : 0.0; // (columns >= n hold Qxu: every use below selects columns < n)
      }
      splitAug(Q, c.Qux, c.Quu);
      F0 = F;
      L0 = L;
    }
    else
    {
      Vec4 Fx, Fu, Lxx, LxuT = {0, 0, 0, 0}, Luu = {0, 0, 0, 0};
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        Fx[rr] = r[mp.fx(rr)];
        Fu[rr] = r[mp.fu(rr)];
        Lxx[rr] = r[mp.lxx(rr)];
      }
#pragma unroll
      for(int rr = 0; rr < KM; rr++)
      {
        LxuT[rr] = r[mp.lxuT(rr)];
        Luu[rr] = r[mp.luu(rr)];
      }
      const S lx = r[mp.lx()], lu = r[mp.lu()];
      c.inv_u = r[mp.oInvU];
      const Vec4 Pa = mma<KN>(c.VV, Fx);
      const Vec4 Pb = mma<KN>(c.VV, Fu);
      c.Qxx = mma<KN>(Pa, Fx);
      c.Qux = mma<KN>(Pb, Fx);
      c.Quu = mma<KN>(Pb, Fu);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        c.Qxx[rr] = Lxx[rr] + c.Qxx[rr];
        c.Qux[rr] = LxuT[rr] + c.Qux[rr];
        c.Quu[rr] = Luu[rr] + c.Quu[rr];
      }
      c.qxrow = lx + Pa[rN]; // lane group qN: Qx[j]
      c.qurow = lu + Pb[rN]; // lane group qN: Qu[j]
      F0 = Fx;
      F1 = Fu;
      L0 = Lxx;
      L1 = LxuT;
      L2 = Luu;
    }
    c.QuxR = c.Qux;
    c.QuuF = c.Quu;
  }
  /** Rows n .. n+m-1 of the augmented Q: Qux in columns < n, Quu in columns n .. n+m-1 (moved to columns 0 .. m-1). */
  NMPC_D void splitAug(const Vec4 & Qa, Vec4 & qux, Vec4 & quu) const
  {
    const int j = colOf(lane & 15);
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      const S e = Qa[(rN + rr) < 4 ? rN + rr : 0];
      const S shifted = fromColumnPlusN(e);
      qux[rr] = (j < N) ? e : 0.0;
      quu[rr] = (j < MM) ? shifted : 0.0;
    }
  }
  /** Phase 1b: regularisation, reg_type 2: Quu_F and Qux_reg rebuilt from Vxx + lambda I    :421-441 */
  NMPC_D void stepRegType2(StepCtx & c, S lambda, const Vec4 & F0, const Vec4 & F1, const Vec4 & L0, const Vec4 & L1,
                           const Vec4 & L2) const
  {
    const int q = lane >> 4, j = colOf(lane & 15);
    Vec4 VVr = c.VV;
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      VVr[rr] = (4 * rr + q == j && j < N) ? c.VV[rr] + lambda : c.VV[rr];
    }
    if constexpr(kAug)
    {
      const Vec4 G2 = mma<KN>(VVr, F0);
      Vec4 Q2 = mma<KN>(G2, F0);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        Q2[rr] = L0[rr] + Q2[rr];
      }
      c.QuxR = Vec4{0, 0, 0, 0};
      c.QuuF = Vec4{0, 0, 0, 0};
      splitAug(Q2, c.QuxR, c.QuuF);
    }
    else
    {
      const Vec4 Pbr = mma<KN>(VVr, F1);
      c.QuxR = mma<KN>(Pbr, F0);
      c.QuuF = mma<KN>(Pbr, F1);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        c.QuxR[rr] = L1[rr] + c.QuxR[rr];
        c.QuuF[rr] = L2[rr] + c.QuuF[rr];
      }
    }
  }
  /** ... reg_type 1: Quu_F = Quu + lambda I */
  NMPC_D void stepRegType1(StepCtx & c, S lambda) const
  {
    const int q = lane >> 4, j = colOf(lane & 15);
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      c.QuuF[rr] = (4 * rr + q == j) ? c.Quu[rr] + lambda : c.Quu[rr];
    }
  }
  /** The per-lane LDS addresses of a step's exchange / transposition accesses, computed ONCE per sweep: byte offsets from the
      workgroup's LDS array, two per register (the wave scratch lies in the first 64 KB).  Recomputed per step from the lane id
      they were 80 of a quadrotor step's 330 VALU instructions (shifts, multiplies, compares and selects for the masked
      ones); kept as eight unpacked registers across the step the register allocator spilled them (see freshLane).  A step
      unpacks what it needs (one instruction per address) behind an opaque copy, so the packed form is what stays live. */
  struct LaneAddr
  {
    unsigned ew_equ; //!< exchange write: this lane's column, row q | Qu -> column n (or the dump word)
    unsigned eqx_rq; //!< Qx -> its column (or the dump word) | exchange read: this lane's column of [Qux_reg | Qu]
    unsigned rx_tw; //!< Qx column read (lanes of column n; the zero words elsewhere) | transposition write: row q, column j
    unsigned tr_trn; //!< transposition read-back: rows 4 r + q < 4 rN of column j (zero words outside) | row 4 rN + q (q < qN)
  };
  NMPC_D static unsigned pack2(int lo_elements, int hi_elements)
  {
    return static_cast<unsigned>(lo_elements * static_cast<int>(sizeof(S))) | (static_cast<unsigned>(hi_elements * static_cast<int>(sizeof(S))) << 16);
  }
  NMPC_D LaneAddr makeLaneAddr() const
  {
    const int q = lane >> 4, j = colOf(lane & 15);
    const int w0 = kWaveAt + ((wave - 1) * kScratchPerWave) * kWaveDoubles; // index of the wave's scratch in the LDS array
    static_assert((kWaveAt + kT64MatrixWaves * kScratchPerWave * kWaveDoubles) * static_cast<int>(sizeof(S)) < 65536,
                  "the wave scratch lies in the first 64 KB");
    LaneAddr la;
    la.ew_equ = pack2(w0 + kColLd * j + q, (q == qN && j < MM) ? w0 + wQQ + kColLd * N + j : w0 + wDump);
    la.eqx_rq = pack2((q == qN && j < N) ? w0 + wX + 4 * (j & 3) + (j >> 2) : w0 + wDump, w0 + wQQ + kColLd * ((j < N) ? j : N));
    la.rx_tw = pack2((j == N) ? w0 + wX + 4 * q : w0 + wZero, w0 + wT + q * kTrLd + j);
    la.tr_trn = pack2((j < N) ? w0 + wT + j * kTrLd + q : w0 + wZero, (q < qN && j < N) ? w0 + wT + j * kTrLd + 4 * rN + q : w0 + wZero);
    return la;
  }
  /** The S at byte offset `at` of the LDS array. */
  NMPC_D S & ldsAt(unsigned at) const
  {
    return *reinterpret_cast<S *>(reinterpret_cast<char *>(lds) + at);
  }
  NMPC_D static unsigned lo16(unsigned p)
  {
    return p & 0xffffu;
  }
  NMPC_D static unsigned hi16(unsigned p)
  {
    return p >> 16;
  }
  /** The lane id as a value the compiler cannot hoist computations on out of the timestep loop.  The per-lane LDS addresses of
      the exchange / transposition are a handful of integer operations on it; computed once per sweep they are ~10 more
      registers live across the whole step, which the register allocator spilled — and reloaded one by one behind a vmcnt(0)
      (measured: six scratch round trips in front of the exchange writes, ~2.5 k of a step's 7.9 k cycles). */
  NMPC_D int freshLane() const
  {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  }
  /** Phase 1c: column exchange through the wave's scratch W, write side: lane (., c) will get column c of [Qux_reg | Qu],
      every lane Quu_F; Qx goes from a row of lanes to a column. */
  NMPC_D void stepExchangeWrite(const StepCtx & c, const LaneAddr & la) const
  {
    static_assert(kColLd <= 9 && kColLd >= 4 * KM, "columns of the exchange area");
    S * Wc = &ldsAt(lo16(la.ew_equ)); // this lane's column, row q
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      Wc[wQQ + 4 * rr] = c.QuxR[rr]; // (unmasked: see kColLd)
      Wc[wF + 4 * rr] = c.QuuF[rr];
    }
    fence(); // column n of the area is Qu's: written after the (padding) rows the lanes of column n have just put there
    ldsAt(hi16(la.ew_equ)) = c.qurow; // (lane group qN, j < m: column n of [Qux_reg | Qu]; the other lanes: the dump word)
    ldsAt(lo16(la.eqx_rq)) = c.qxrow; // (lane group qN, j < n)
  }
  /** Phase 2: ... read side. */
  NMPC_D void stepExchangeRead(StepCtx & c, const S * W, const LaneAddr & la) const
  {
#pragma unroll
    for(int cc = 0; cc < MM; cc++)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        c.fac[a + cc * MM] = (kConstrained || a >= cc) ? W[wF + kColLd * cc + a] : 0.0; // (the factorisation reads the lower triangle)
      }
    }
    const S * col_j = &ldsAt(hi16(la.eqx_rq)); // column min(j, n) of [Qux_reg | Qu]
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      c.colQ[a] = col_j[a];
      c.inv_d[a] = 0;
    }
    // Qx as column n of the accumulator of the value update: lane (q, n) register r <- Qx[4 r + q] (the other lanes: zero words)
    const S * qx = &ldsAt(lo16(la.rx_tw));
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      c.qxcol[rr] = qx[rr];
    }
  }
  /** Phase 3a, unconstrained: every lane factorises Quu_F, lane (., c) solves column c    :500-517.  Branch-free. */
  NMPC_D void stepGains(StepCtx & c) const
  {
    const bool ok_now = ldlt(c.fac, c.inv_d);
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      c.col[a] = c.colQ[a];
    }
    ldltSolve(c.fac, c.inv_d, c.col);
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      c.col[a] = -1 * c.col[a];
    }
    c.ok = c.ok && ok_now; // (after a failure the slot computes on garbage and stores nothing)
  }
  /** Phase 3a, box-constrained (:450-497). */
  NMPC_D void stepGainsBoxQP(StepCtx & c, const LaneMap & mp, const S * r, S * W, int slot, int b, int i) const
  {
    const int j = colOf(lane & 15);
    S (&fac)[MM * MM] = c.fac;
    S (&col)[MM] = c.col;
    S (&colQ)[MM] = c.colQ;
    bool ok_now = true;
    {
      // every lane solves the same small QP (BoxQP.h:141-347, the lane kernel's implementation)    :450-497
      S initial_k[MM], lo[MM], up[MM], Qu[MM];
      S * knext = lds + kKnextAt + slot * 8;
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        const S ua = r[mp.oU[a]];
        initial_k[a] = (i != T - 1) ? knext[a] : 0.0; // warm start from k_{i+1}    :452-467
        lo[a] = inputLimitLo(buf, b, i, a) - ua; // :470-472
        up[a] = inputLimitHi(buf, b, i, a) - ua;
        Qu[a] = W[wQQ + kColLd * N + a];
      }
      const Lane lane_code(problem, cfg, buf, b);
      typename Lane::QPOutMasked qp;
      lane_code.boxQPMasked(fac, Qu, lo, up, initial_k, qp); // (static m = MM: no index lists, no private memory)
      if(lane == 0 && c.ok) // (backwardPass returns at the first failing timestep, :473-480: nothing below it is written)
      {
        const size_t tl = tileOf(b), ln = lnOf(b);
        buf.qp_ret[(tl * T + i) * 64 + ln] = qp.retval;
        buf.qp_free[(tl * T + i) * 64 + ln] = qp.free;
      }
      ok_now = !(qp.retval < 0); // :473-480
      if(j == N)
      {
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          col[a] = qp.x[a];
        }
      }
      else
      {
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          col[a] = colQ[a];
        }
        Lane::maskedGainColumn(qp, col); // clamped rows of K stay zero    :482-496
      }
      fence();
      if(lane == 0)
      {
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          knext[a] = qp.x[a];
        }
      }
      fence();
    }
    c.ok = c.ok && ok_now;
  }
  // ---- batched BoxQP (kBatchQP): prepare / batch / gains from the parked result -------------------------------------------------
  NMPC_D S * qpIn(int e) const
  {
    return lds + kQpAt + (wave - 1) * kQpPerWave + e * kQpIn;
  }
  NMPC_D S * qpOut(int e) const
  {
    return lds + kQpAt + (wave - 1) * kQpPerWave + (kPerWave - 1) * kQpIn + 2 * e;
  }
  /** Slot e of the wave, timestep i: the Q terms and the regularisation as backwardStep computes them, Quu_F and Qu through the
      exchange scratch into the slot's QP input (the last slot's stay in the scratch: nobody writes it before the batch). */
  NMPC_D void qpPrepare(const Vec4 & VV, const LaneMap & mp, const LaneAddr & la_sweep, const S * r, S lambda, int e) const
  {
    S * W = waveScratch(0);
    LaneAddr la = la_sweep;
    asm volatile("" : "+v"(la.ew_equ), "+v"(la.eqx_rq), "+v"(la.rx_tw), "+v"(la.tr_trn));
    StepCtx c;
    c.VV = VV;
    c.ok = true;
    Vec4 F0, F1, L0, L1, L2;
    stepQTerms(c, mp, r, F0, F1, L0, L1, L2);
    if(cfg.reg_type == 2)
    {
      stepRegType2(c, lambda, F0, F1, L0, L1, L2);
    }
    else if(cfg.reg_type == 1)
    {
      stepRegType1(c, lambda);
    }
    stepExchangeWrite(c, la);
    fence();
    if(e < kPerWave - 1)
    {
      S * in = qpIn(e);
      if(lane < MM * MM)
      {
        in[lane] = W[wF + kColLd * (lane / MM) + lane % MM]; // H(a, c) at a + c MM, as stepExchangeRead fills c.fac
      }
      else if(lane < kQpIn)
      {
        in[lane] = W[wQQ + kColLd * N + (lane - MM * MM)]; // Qu
      }
    }
    fence();
  }
  /** The QPs of this wave's slots of timestep i, lane e = slot e (BoxQP.h:141-347 through InstanceSolver::boxQPMasked: the code every
      lane of the wave used to run for one slot at a time).  x goes to the slot's warm-start row (k_{i+1} of the next timestep's QP
      and k_i of this timestep's gains), free set and return code to qpOut, and — while the slot's pass has not failed — to HBM. */
  NMPC_D void qpBatch(const LaneMap & mp, int i, int parity, int dt, int chunk, int n_act, unsigned ok_mask) const
  {
    const int mw = wave - 1;
    const int e = lane;
    const int a_idx = mw + kOwnerWaves * e;
    if(mw < kOwnerWaves && e < kPerWave && a_idx < n_act)
    {
      const int slot = actSlot(a_idx);
      const int b = slotI(sB, slot);
      const S * r = recAt(parity, dt, a_idx, chunk, n_act);
      const S * W = waveScratch(0);
      const S * in = qpIn((mw < kOwnerWaves && e < kPerWave - 1) ? e : 0);
      S H[MM * MM], g[MM], lo[MM], up[MM], k0[MM];
      S * knext = lds + kKnextAt + slot * 8;
#pragma unroll
      for(int cc = 0; cc < MM; cc++)
      {
#pragma unroll
        for(int aa = 0; aa < MM; aa++)
        {
          const S parked = in[aa + cc * MM], last = W[wF + kColLd * cc + aa];
          H[aa + cc * MM] = (e < kPerWave - 1) ? parked : last;
        }
      }
#pragma unroll
      for(int aa = 0; aa < MM; aa++)
      {
        const S parked = in[MM * MM + aa], last = W[wQQ + kColLd * N + aa];
        g[aa] = (e < kPerWave - 1) ? parked : last;
        const S ua = r[mp.oU[aa]];
        k0[aa] = (i != T - 1) ? knext[aa] : 0.0; // warm start from k_{i+1}    :452-467
        lo[aa] = inputLimitLo(buf, b, i, aa) - ua; // :470-472
        up[aa] = inputLimitHi(buf, b, i, aa) - ua;
      }
      const Lane lane_code(problem, cfg, buf, b);
      typename Lane::QPOutMasked qp;
      lane_code.boxQPMasked(H, g, lo, up, k0, qp);
      S * out = qpOut(e);
      out[0] = static_cast<S>(qp.free);
      out[1] = static_cast<S>(qp.retval);
#pragma unroll
      for(int aa = 0; aa < MM; aa++)
      {
        knext[aa] = qp.x[aa];
      }
      if(((ok_mask >> e) & 1u) != 0) // (backwardPass returns at the first failing timestep, :473-480: nothing below it is written)
      {
        const size_t tl = tileOf(b), ln = lnOf(b);
        buf.qp_ret[(tl * T + i) * 64 + ln] = qp.retval;
        buf.qp_free[(tl * T + i) * 64 + ln] = qp.free;
      }
    }
    fence();
  }
  /** Phase 3a of a slot whose QP the batch has solved: the factorisation of the free block is rebuilt from Quu_F and the free set —
      the statements of boxQPMasked's last refactorisation on the same values, hence the same factor — and lane (., j) solves its
      column of K on the free rows (:482-496); column n takes k = x. */
  NMPC_D void stepGainsFromQP(StepCtx & c, int slot, int e) const
  {
    const int j = colOf(lane & 15);
    const S * out = qpOut(e);
    typename Lane::QPOutMasked qp;
    qp.free = static_cast<unsigned>(out[0]);
    qp.retval = static_cast<int>(out[1]);
    const S * knext = lds + kKnextAt + slot * 8;
    const unsigned clamped = ~qp.free;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      qp.x[a] = knext[a];
      qp.inv_d[a] = 0;
#pragma unroll
      for(int cc = 0; cc < MM; cc++)
      {
        const bool both = (((clamped >> a) | (clamped >> cc)) & 1u) == 0;
        qp.fac[a + cc * MM] = both ? c.fac[a + cc * MM] : ((a == cc) ? 1.0 : 0.0);
      }
    }
    if(qp.free != 0 && qp.retval >= 0)
    {
      (void)Lane::template ldltInPlace<MM>(qp.fac, qp.inv_d, MM); // (succeeded in the batch: the same values)
    }
    if(j == N)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        c.col[a] = qp.x[a];
      }
    }
    else
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        c.col[a] = c.colQ[a];
      }
      Lane::maskedGainColumn(qp, c.col); // clamped rows of K stay zero    :482-496
    }
    c.ok = c.ok && !(qp.retval < 0); // :473-480
  }
  /** Phase 3b: A = [K | k], QQ = [Qux | Qu] in natural layout; the cost-to-go (:522-527) up to the symmetrisation; rows of the new
      value function to the scratch.  Branch-free. */
  NMPC_D void stepValueUpdate(StepCtx & c, S * W, int m, const LaneAddr & la) const
  {
    const int fl = freshLane(), q = fl >> 4, j = colOf(fl & 15);
    Vec4 QQ = {0, 0, 0, 0};
    S kn = 0; // |k|^2 (lanes of column n hold k)    :217-221
    if constexpr(kBig)
    {
      if(m > 0)
      {
        stepGainsNatural(c, W, m, QQ);
      }
      else
      {
        // no input at this timestep (:513-517): k, K empty, the value function is [Qxx | Qx]
        c.A = Vec4{0, 0, 0, 0};
#pragma unroll
        for(int rr = 0; rr < 4; rr++)
        {
          const int row = 4 * rr + q;
          const S qx_c = fromLane(c.qxrow, 16 * qN + laneOfCol(row & 15));
          c.qxcol[rr] = (j == N && row < N) ? qx_c : 0.0;
        }
      }
      fence(); // (the scratch is written again below)
      // |k|^2: the entries of k are spread over the four lane groups of column n
#pragma unroll
      for(int rr = 0; rr < KM; rr++)
      {
        kn += c.A[rr] * c.A[rr];
      }
      kn += fromLane(kn, fl ^ 16);
      kn += fromLane(kn, fl ^ 32);
    }
    else
    {
    c.A = Vec4{0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      S g[4], c4[4];
#pragma unroll
      for(int e = 0; e < 4; e++)
      {
        g[e] = (4 * rr + e < MM) ? c.col[(4 * rr + e < MM) ? 4 * rr + e : 0] : 0.0;
        c4[e] = (4 * rr + e < MM) ? c.colQ[(4 * rr + e < MM) ? 4 * rr + e : 0] : 0.0;
      }
      const S gq = pick4(q, g[0], g[1], g[2], g[3]);
      const S cq = pick4(q, c4[0], c4[1], c4[2], c4[3]);
      c.A[rr] = (j <= N) ? gq : 0.0;
      QQ[rr] = (j < N) ? c.Qux[rr] : ((j == N) ? cq : 0.0); // (column n of colQ is Qu; unregularised Qux elsewhere)
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      kn += c.col[a] * c.col[a];
    }
    }
    stepValueTail(c, QQ, kn, m, la, q, j);
  }
  /** ... the four products of the value update from A = [K | k], QQ = [Qux | Qu], |k|^2 (kn): what follows the gains, whoever computed them. */
  NMPC_D void stepValueTail(StepCtx & c, const Vec4 & QQ, S kn, int m, const LaneAddr & la, int q, int j) const
  {
    const Vec4 Z = mma<KM>(c.Quu, c.A);
    const Vec4 C1 = mma<KM>(Z, c.A);
    const Vec4 T2 = mma<KM>(c.A, QQ);
    const Vec4 T3 = mma<KM>(c.Qux, c.A);
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      const S c0 = (j < N) ? c.Qxx[rr] : c.qxcol[rr];
      c.Vn[rr] = ((c0 + C1[rr]) + T2[rr]) + T3[rr];
    }
    c.Vn[rN] = (q == qN) ? 0.0 : c.Vn[rN]; // row n: k^T (...), not part of the value function
    c.c1nn = C1[rN]; // lane kStarLane: k^T Quu k, k^T Qu    :522-523
    c.t2nn = T2[rN];
    // |k| / (|u| + 1)    :217-221
    // m > 1: the SQUARE of the ratio — the maximum over the horizon commutes with the square root, which the model wave takes
    // once per sweep instead of every matrix lane once per timestep (twenty instructions of a step's eight hundred)
    c.krel_i = kKrelSquared ? kn * (c.inv_u * c.inv_u) : fabs(c.col[0]) * c.inv_u;
    if constexpr(kDyn)
    {
      c.krel_i = (m > 0) ? c.krel_i : 0.0; // (the reference skips the timesteps without input, :220)
    }
    // Vxx <- (Vxx + Vxx^T) / 2: rows to the scratch, columns back (phase 4)
    S * row_q = &ldsAt(hi16(la.rx_tw)); // entry (q, j) of the transposition scratch
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      row_q[4 * rr * kTrLd] = c.Vn[rr];
    }
  }
  /** Phase 4: the symmetrised value function; dV and the running max of |k| / (|u| + 1) in the slot table (lane kStarLane; the
      other lanes update a dump word: no branch). */
  NMPC_D void stepFinish(StepCtx & c, const LaneMap & mp, S * W, int slot, const LaneAddr & la) const
  {
    const int fl = freshLane();
    // entry (j, 4 r + q) of the scratch where it belongs to the n x n block (the zero words elsewhere): rows < 4 rN from one
    // address with immediate offsets, row 4 rN + q from its own, the rows behind are outside the block
    const S * col_j = &ldsAt(lo16(la.tr_trn));
    S wn = mp.wn, wt = mp.wt;
    if constexpr(kPackMap)
    {
      const int jj = colOf(fl & 15); // (the weights from the lane id: two registers less across the sweep)
      wn = (jj < N) ? 0.5 : ((jj == N) ? 1.0 : 0.0);
      wt = (jj < N) ? 0.5 : 0.0;
    }
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      S vt = 0.0;
      if(rr < rN)
      {
        vt = col_j[4 * rr];
      }
      else if(rr == rN && qN > 0)
      {
        vt = ldsAt(hi16(la.tr_trn));
      }
      c.VV[rr] = wn * c.Vn[rr] + wt * vt;
    }
    const bool star = fl == kStarLane;
    S * dv0 = star ? &slotF(sDV0, slot) : W + wDump;
    S * dv1 = star ? &slotF(sDV1, slot) : W + wDump + 1;
    S * kr = star ? &slotF(sKrel, slot) : W + wDump;
    *dv0 += c.t2nn;
    *dv1 += S(0.5) * c.c1nn;
    *kr = fmax(*kr, c.krel_i);
  }
  // ---------------------------------------------------------------------------------------------------
  // gains in natural layout (kBig: m > 8 or inputDim(t))    :500-517
  // ---------------------------------------------------------------------------------------------------
  // Sixteen by sixteen doubles do not fit a lane's registers, so nobody holds the whole factor: Quu_F and the right-hand sides
  // [Qux_reg | Qu] stay where the matrix cores left them — register r of lane (q, c) = entry (4 r + q, c) — and the L D L^T runs
  // right-looking over the wave: for pivot j the normalised column j reaches the lanes of its rows by a DPP row broadcast
  // (row_newbcast: lane j of every 16-lane row), the entry (j, c) of the pivot row reaches the lanes of column c through the
  // LDS crossbar (ds_bpermute, no memory), and the trailing block gets its rank-one update A_ik -= (L_ij L_kj) d_j — the
  // association of the lane kernels' ldltInPlace — in at most four instructions per lane.  Both triangles of the trailing block
  // are kept (the updates are symmetric bit for bit), which is what makes the mirrored entry (j, c) available as L_cj d_j.
  // The forward substitution rides on the same broadcasts; the backward substitution runs column by column (axpy form).
  // Pivots >= m (run-time input dimension) are skipped: the blocks are padded with zeros.  ~50 instructions per pivot.
  /** The value of lane `src` (0 .. 63, any expression) — through the LDS crossbar. */
  NMPC_D static S fromLane(S v, int src)
  {
    return permuted(v, src << 2);
  }
  /** The value of lane J of this lane's 16-lane row (natural-layout gains: double only). */
  template<int J>
  NMPC_D static double fromRowLane(double v)
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, true); // row_newbcast:J (bound_ctrl: every lane has a source, and no "old" value has to be materialised)
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  /** The value of lane `src` (a constant) in every lane, as a wave-uniform value. */
  NMPC_D static double fromLaneUniform(double v, int src)
  {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  }
  /** Pivot J of the factorisation and of the forward substitution (see above). */
  template<int J>
  NMPC_D static void naturalPivot(Vec4 & Aq, Vec4 & R, Vec4 & invd, bool & ok, int q, int col)
  {
    constexpr int rj = J / 4, qj = J % 4;
    const S d = fromLaneUniform(Aq[rj], 16 * qj + J);
    ok = ok && !(d <= 0.0); // the pivot rule of Eigen's LLT (fails iff a pivot is <= 0, NaN passes)
    const S r = recipFast(d);
    invd[rj] = (q == qj) ? r : invd[rj];
    const S ajc = fromLane(Aq[rj], 16 * qj + col); // entry (j, c) = L_cj d_j of this lane's column
    const S yj = fromLane(R[rj], 16 * qj + col); // y_j of this lane's right-hand side
    const S lc = (col > J) ? ajc * r : 0.0; // L_cj; columns <= j are finished: they receive - 0
#pragma unroll
    for(int rr = rj; rr < KM; rr++)
    {
      S lrow = fromRowLane<J>(Aq[rr] * r); // L_ij of this lane's row i = 4 rr + q
      if(rr == rj)
      {
        lrow = (q > qj) ? lrow : 0.0; // rows <= j are finished
      }
      Aq[rr] -= (lrow * lc) * d;
      R[rr] -= lrow * yj;
    }
  }
  /** Column K of the backward substitution: x_K is final, the rows above it receive - L_Ki x_K. */
  template<int K>
  NMPC_D static void naturalBackColumn(const Vec4 & Ln, Vec4 & R, int q, int col)
  {
    constexpr int rk = K / 4, qk = K % 4;
    const S xk = fromLane(R[rk], 16 * qk + col);
#pragma unroll
    for(int rr = 0; rr <= rk; rr++)
    {
      S l = fromRowLane<K>(Ln[rr]); // entry (i, K) d_i^-1 = L_Ki of this lane's row i
      if(rr == rk)
      {
        l = (q < qk) ? l : 0.0;
      }
      R[rr] -= l * xk;
    }
  }
  template<int J>
  NMPC_D static void naturalForward(Vec4 & Aq, Vec4 & R, Vec4 & invd, bool & ok, int q, int col, int m)
  {
    if constexpr(J < MM)
    {
      if(J < m)
      {
        naturalPivot<J>(Aq, R, invd, ok, q, col);
      }
      naturalForward<J + 1>(Aq, R, invd, ok, q, col, m);
    }
  }
  template<int K>
  NMPC_D static void naturalBackward(const Vec4 & Ln, Vec4 & R, int q, int col, int m)
  {
    if constexpr(K >= 1)
    {
      if(K < m)
      {
        naturalBackColumn<K>(Ln, R, q, col);
      }
      naturalBackward<K - 1>(Ln, R, q, col, m);
    }
  }
  /** Phases 1c - 3b for kBig: A = [K | k] = - Quu_F^-1 [Qux_reg | Qu] (zero for m = 0: :513-517), then the value update's operands.
      W: the wave's transposition scratch (the lower triangle of Quu_F is mirrored through it: the reference's LLT reads the
      lower triangle only, :500). */
  NMPC_D void stepGainsNatural(StepCtx & c, S * W, int m, Vec4 & QQ) const
  {
    const int fl = freshLane(), q = fl >> 4, col = colOf(fl & 15);
    // Quu_F <- its lower triangle, mirrored
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      W[wT + (4 * rr + q) * kTrLd + col] = c.QuuF[rr];
    }
    fence();
    Vec4 Aq = {0, 0, 0, 0}, R = {0, 0, 0, 0}, invd = {1, 1, 1, 1};
    QQ = Vec4{0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      const int row = 4 * rr + q;
      const S mirrored = W[wT + col * kTrLd + row]; // entry (col, row)
      Aq[rr] = (row >= col) ? c.QuuF[rr] : mirrored;
      // Qu, Qx from a row of lanes (lane group qN: entry j in lane j) to column n of the tiles
      const S qu_c = fromLane(c.qurow, 16 * qN + laneOfCol(row & 15));
      R[rr] = (col < N) ? c.QuxR[rr] : ((col == N) ? qu_c : 0.0);
      QQ[rr] = (col < N) ? c.Qux[rr] : ((col == N) ? qu_c : 0.0);
    }
    fence();
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      const int row = 4 * rr + q;
      const S qx_c = fromLane(c.qxrow, 16 * qN + laneOfCol(row & 15));
      c.qxcol[rr] = (col == N && row < N) ? qx_c : 0.0;
    }
    bool ok_now = true;
    naturalForward<0>(Aq, R, invd, ok_now, q, col, m);
    // D^-1, then L^T x = z column by column
    Vec4 Ln = {0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      R[rr] = R[rr] * invd[rr];
      Ln[rr] = Aq[rr] * invd[rr];
    }
    naturalBackward<MM - 1>(Ln, R, q, col, m);
    c.A = Vec4{0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      c.A[rr] = (4 * rr + q < m && col <= N) ? -1 * R[rr] : 0.0;
    }
    c.ok = c.ok && ok_now;
  }

  // ---------------------------------------------------------------------------------------------------
  // natural-layout gains, the slots of a wave together (kBatchGains): sixteen lanes per slot, a column per lane
  // ---------------------------------------------------------------------------------------------------
  // stepGainsNatural spends ~37 of its ~50 instructions per pivot on moving ONE slot's pivot row and pivot across the four lane groups
  // (v_readlane, two ds_bpermute pairs and their waits) — a latency chain of ~9 k cycles per slot and timestep, run for the wave's
  // slots one after the other (centroidal, 4096 instances: three slots per wave; profiles/r05_pmc_summary_centroidal.txt: VALU : MFMA
  // 42 : 1).  Here lane c of lane group g holds COLUMN c of slot g's Quu_F — all sixteen rows, both triangles — and column c of its
  // right-hand sides [Qux_reg | Qu] in registers: the pivot row entry (j, c) = L_cj d_j is the lane's OWN register j, the pivot d_j
  // and the column L_ij come from lane j of the lane's 16-lane row by a DPP row broadcast (v_mov_b64_dpp row_newbcast: the one
  // DPP control the fp64 pipeline takes), and the back substitution is one v_fmac_f64_dpp per (row, column).  No cross-lane-group
  // move, no LDS, and up to four slots share every instruction.  The arithmetic is stepGainsNatural's, operation for operation:
  // the rank-one update (L_ij L_cj) d_j on both triangles, y_i -= L_ij y_j, z_j = y_j / d_j, x_i -= L_Ki x_K.
  /** What a slot keeps in registers between its prepare and its complete trip (next to [Qxx | Qx], which waits in the slot's value
      function registers: the old [Vxx | Vx] is dead once the Q terms are formed). */
  struct GainStash
  {
    Vec4 Quu, QQ;
  };
  /** The value of lane J of this lane's 16-lane row (one instruction; s_nop: a DPP operand written by the VALU instruction in front
      of it needs two wait states, and the hazard recogniser does not look into inline assembly). */
  template<int J>
  NMPC_D static double bcastRow(double v)
  {
    double o;
    // (without the s_nop the results are WRONG — measured, HISTORY round 6: 980 of 4 096 instances failed)
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(J));
    return o;
  }
  /** acc -= (lane K's a) * x   (the back substitution's step; a was written long before: no wait states needed) */
  template<int K>
  NMPC_D static void fmsubRow(double & acc, double a, double x)
  {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(x), "n"(K));
  }
  /** Pivot J of the factorisation and the forward substitution for every lane group at once.  m_lane: the input dimension of the
      lane's slot (pivots beyond it leave the padded blocks alone: r = 0). */
  template<int J>
  NMPC_D static void batchPivot(S (&a)[MM], S (&y)[MM], bool & ok, int c, int m_lane)
  {
    const S d = bcastRow<J>(a[J]);
    const bool act = !kDyn || J < m_lane;
    ok = ok && !(act && d <= 0.0); // the pivot rule of Eigen's LLT (fails iff a pivot is <= 0, NaN passes)
    const S r = act ? recipFast(d) : 0.0;
    const S lc = (c > J) ? a[J] * r : 0.0; // L_cj; columns <= j are finished: they receive - 0
    const S yj = y[J];
#pragma unroll
    for(int i = J + 1; i < MM; i++)
    {
      const S li = bcastRow<J>(a[i] * r); // L_ij
      a[i] -= (li * lc) * d;
      y[i] -= li * yj;
    }
    a[J] = lc; // lane K > j: L_Kj, what the back substitution takes from lane K for row j
    y[J] = act ? yj * r : yj; // z_j = y_j / d_j
  }
  // (No branch per pivot: a run-time input dimension is handled by the lanes' masks alone — a pivot beyond a slot's dimension moves
  // zeros — and the whole batch is skipped when no slot has an input.  Branches on m here cost more than they save: sixteen
  // control-flow joins with thirty-two live values each.)
  template<int J>
  NMPC_D static void batchForward(S (&a)[MM], S (&y)[MM], bool & ok, int c, int m_lane)
  {
    if constexpr(J < MM)
    {
      batchPivot<J>(a, y, ok, c, m_lane);
      batchForward<J + 1>(a, y, ok, c, m_lane);
    }
  }
  template<int K>
  NMPC_D static void batchBackward(const S (&a)[MM], S (&y)[MM])
  {
    if constexpr(K >= 1)
    {
      const S xk = y[K];
#pragma unroll
      for(int i = K - 1; i >= 0; i--) // (row K - 1 first: it is the next column's x)
      {
        fmsubRow<K>(y[i], a[i], xk);
      }
      batchBackward<K - 1>(a, y);
    }
  }
  /** a: columns of Quu_F, y: columns of [Qux_reg | Qu] -> y = Quu_F^-1 [Qux_reg | Qu] per lane group; false in the lanes of a group
      whose factorisation failed. */
  NMPC_D bool gainsBatch(S (&a)[MM], S (&y)[MM], int m_lane) const
  {
    const int c = freshLane() & 15;
    bool ok = true;
    asm volatile("s_nop 4"); // (an exec mask written by a VALU compare in front of a DPP instruction: five wait states)
    batchForward<0>(a, y, ok, c, m_lane);
    batchBackward<MM - 1>(a, y);
    return ok;
  }
  /** Prepare trip of the slot whose columns lane group G4 will hold: Q terms and regularisation as backwardStep computes them, the
      operands of the value update into VV ([Qxx | Qx]) and the stash, Quu_F (lower triangle mirrored: the reference's LLT reads the
      lower triangle only, :500) and [Qux_reg | Qu] through the staging into a[], y[] of lane group G4. */
  template<int G4>
  NMPC_D void gainsPrepare(Vec4 & VV, GainStash & st, S (&a)[MM], S (&y)[MM], int & m_lane, const LaneMap & mp, const S * r, S lambda, int m) const
  {
    S * W = waveScratch(0);
    StepCtx c;
    c.VV = VV;
    c.ok = true;
    Vec4 F0, F1, L0, L1, L2;
    stepQTerms(c, mp, r, F0, F1, L0, L1, L2);
    if(cfg.reg_type == 2)
    {
      stepRegType2(c, lambda, F0, F1, L0, L1, L2);
    }
    else if(cfg.reg_type == 1)
    {
      stepRegType1(c, lambda);
    }
    const int fl = freshLane(), q = fl >> 4, col = colOf(fl & 15);
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      W[wT + (4 * rr + q) * kTrLd + col] = c.QuuF[rr];
    }
    fence();
    Vec4 Aq = {0, 0, 0, 0}, R = {0, 0, 0, 0}, QQ = {0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      const int row = 4 * rr + q;
      const S mirrored = W[wT + col * kTrLd + row]; // entry (col, row)
      Aq[rr] = (row >= col) ? c.QuuF[rr] : mirrored;
      const S qu_c = fromLane(c.qurow, 16 * qN + laneOfCol(row & 15));
      R[rr] = (col < N) ? c.QuxR[rr] : ((col == N) ? qu_c : 0.0);
      QQ[rr] = (col < N) ? c.Qux[rr] : ((col == N) ? qu_c : 0.0);
    }
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      const int row = 4 * rr + q;
      const S qx_c = fromLane(c.qxrow, 16 * qN + laneOfCol(row & 15));
      VV[rr] = (col < N) ? c.Qxx[rr] : ((col == N && row < N) ? qx_c : 0.0); // [Qxx | Qx]: what the value update starts from
    }
    st.Quu = c.Quu;
    st.QQ = QQ;
    fence(); // (the mirror has been read: the staging aliases it)
    S * Wc = W + col * kStLd + 4 * q;
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      Wc[wStA + rr] = Aq[rr];
      Wc[wStR + rr] = R[rr];
    }
    fence();
    if(q == G4)
    {
      const S * colp = W + col * kStLd;
#pragma unroll
      for(int row = 0; row < MM; row++)
      {
        a[row] = colp[wStA + 4 * (row & 3) + (row >> 2)];
        y[row] = colp[wStR + 4 * (row & 3) + (row >> 2)];
      }
      m_lane = m;
    }
    fence();
  }
  /** Complete trip: the solved columns of lane group G4 back to the natural layout (A = - x where the slot has inputs and the
      column is one of [K | k]), then the value update, the symmetrisation and the gain record as backwardStep does them. */
  template<int G4>
  NMPC_D void gainsComplete(Vec4 & VV, const GainStash & st, const S (&y)[MM], bool ok_now, bool & ok, const LaneMap & mp,
                            const LaneAddr & la_sweep, const S * r, int slot, int b, int i, int m) const
  {
    S * W = waveScratch(0);
    LaneAddr la = la_sweep;
    asm volatile("" : "+v"(la.ew_equ), "+v"(la.eqx_rq), "+v"(la.rx_tw), "+v"(la.tr_trn));
    const int fl = freshLane(), q = fl >> 4, col = colOf(fl & 15);
    fence();
    if(q == G4)
    {
      S * colp = W + col * kStLd;
#pragma unroll
      for(int row = 0; row < MM; row++)
      {
        colp[wStR + 4 * (row & 3) + (row >> 2)] = y[row];
      }
    }
    fence();
    StepCtx c;
    c.ok = ok && ok_now;
    c.Quu = st.Quu;
    c.Qux = st.QQ; // (column n holds Qu: it only reaches row n of Qux^T A, which is not part of the value function)
    c.Qxx = VV;
    c.qxcol = VV;
    c.inv_u = r[mp.oInvU];
    const S * Wc = W + col * kStLd + 4 * q + wStR;
    c.A = Vec4{0, 0, 0, 0};
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      c.A[rr] = (4 * rr + q < m && col <= N) ? -1 * Wc[rr] : 0.0;
    }
    fence(); // (the scratch is written again below)
    S kn = 0;
#pragma unroll
    for(int rr = 0; rr < KM; rr++)
    {
      kn += c.A[rr] * c.A[rr];
    }
    kn += fromLane(kn, fl ^ 16);
    kn += fromLane(kn, fl ^ 32);
    stepValueTail(c, st.QQ, kn, m, la, q, col);
    fence();
    stepFinish(c, mp, W, slot, la);
    fence();
    stepStoreGains(c, b, i);
    VV = c.VV;
    ok = c.ok;
  }

  /** k_i, K_i -> the instance's gain record (:529-530); not after a failed factorisation: backwardPass() returned before storing (:505-508). */
  NMPC_D void stepStoreGains(const StepCtx & c, int b, int i) const
  {
    const int q = lane >> 4, j = colOf(lane & 15);
    if(c.ok && j <= N)
    {
      S * rec_g = buf.wpi_ws + (static_cast<size_t>(b) * T + i) * kGainRows;
#pragma unroll
      for(int rr = 0; rr < KM; rr++)
      {
        const int a = 4 * rr + q;
        if(a < MM)
        {
          rec_g[(j == N) ? a : MM + a + j * MM] = c.A[rr];
        }
      }
    }
  }

  /** One timestep of one instance.  VV = [Vxx | Vx] in natural layout (in / out), r = the instance's record of this timestep,
      ok = no factorisation of this sweep has failed yet (in / out).  Everything but the lane id is wave-uniform. */
  NMPC_D void backwardStep(Vec4 & VV, bool & ok, const LaneMap & mp, const LaneAddr & la_sweep, const S * r, int slot, int b, int i,
                           S lambda, int qp_e = -1) const
  {
    S * W = waveScratch(0);
    LaneAddr la = la_sweep;
    asm volatile("" : "+v"(la.ew_equ), "+v"(la.eqx_rq), "+v"(la.rx_tw), "+v"(la.tr_trn)); // (unpacked here, not once per sweep)
    StepCtx c;
    c.VV = VV;
    c.ok = ok;
    Vec4 F0, F1, L0, L1, L2;
    stepQTerms(c, mp, r, F0, F1, L0, L1, L2);
    if(cfg.reg_type == 2)
    {
      stepRegType2(c, lambda, F0, F1, L0, L1, L2);
    }
    else if(cfg.reg_type == 1)
    {
      stepRegType1(c, lambda);
    }
    int m = MM;
    if constexpr(kDyn)
    {
      m = uniform(static_cast<int>(r[mp.oM])); // inputDim(t_i), from the record (the model wave evaluated it)
    }
    if constexpr(!kBig)
    {
      stepExchangeWrite(c, la);
      fence();
      stepExchangeRead(c, W, la);
      fence();
      if constexpr(kBatchQP)
      {
        stepGainsFromQP(c, slot, qp_e); // (the QP itself: qpBatch, for all of the wave's slots of this timestep at once)
      }
      else if constexpr(kConstrained)
      {
        stepGainsBoxQP(c, mp, r, W, slot, b, i);
      }
      else
      {
        stepGains(c);
      }
    }
    stepValueUpdate(c, W, m, la);
    fence();
    stepFinish(c, mp, W, slot, la);
    fence();
    stepStoreGains(c, b, i);
    VV = c.VV;
    ok = c.ok;
  }
  /** The sweep of the matrix waves (barriers are shared with the model wave's loop, backwardSweepModel).
      The slots that take part in the sweep are dealt to the matrix waves by their ACTIVE index (a = mw + 7 e: a sweep that
      only a few slots of the group still need — the last iterations of a batch — spreads over all waves), and the value
      functions of a wave's (up to five) slots stay in registers for the whole sweep.  The slot loop is a real loop — one copy
      of the step in the instruction cache — that always works on the first register(s) and then rotates them (register moves;
      an array indexed by the trip count would live in scratch memory): after the last trip every value function is back in
      its place.  Within a chunk of timesteps (backwardSweepModel) the waves do not wait for each other: the matrix waves of a
      group share nothing but the record buffers. */
  NMPC_D void backwardSweepMatrix() const
  {
    static_assert(kPerWave <= 16, "ok_mask and the QP batch's lane = slot mapping");
    const LaneMap mp = makeLaneMap();
    const LaneAddr la = makeLaneAddr();
    const int q = lane >> 4, j = colOf(lane & 15);
    const int mw = wave - 1;
    const int n_act = uniform(meta(mNAct)), chunk = uniform(meta(mChunk));
    Vec4 V[kPerWave]; // (indexed by compile-time constants only: registers)
    Vec4 & V0 = V[0];
    auto rotate = [&]()
    {
      const Vec4 t = V[0];
#pragma unroll
      for(int k = 0; k + 1 < kPerWave; k++)
      {
        V[k] = V[k + 1];
      }
      V[kPerWave - 1] = t;
    };
    unsigned ok_mask = ~0u;
    barrier(); // the terminal records are complete
    auto loadTerminal = [&](int e) -> Vec4
    {
      const int a = mw + kOwnerWaves * e;
      Vec4 v = {0, 0, 0, 0};
      if(mw < kOwnerWaves && a < n_act)
      {
        const int slot = uniform(actSlot(a));
        const S * tr = term(slot);
#pragma unroll
        for(int rr = 0; rr < 4; rr++)
        {
          const int row = 4 * rr + q;
          const S t = tr[(row < N && j <= N) ? j * N + row : 0];
          v[rr] = (row < N && j <= N) ? t : 0.0;
        }
        if(lane == kStarLane)
        {
          slotF(sDV0, slot) = 0;
          slotF(sDV1, slot) = 0;
          slotF(sKrel, slot) = 0;
        }
      }
      return v;
    };
#pragma unroll
    for(int e = 0; e < kPerWave; e++)
    {
      V[e] = loadTerminal(e);
    }
    barrier(); // the terminal records have been read
    barrier(); // the records of the first chunk are complete
    int parity = 0;
    for(int hi = T - 1; hi >= 0; hi -= chunk, parity ^= 1)
    {
      const unsigned long long pa = profNow();
      const int lo = (hi - chunk + 1 > 0) ? hi - chunk + 1 : 0;
      for(int i = hi; i >= lo; i--)
      {
        if constexpr(kBatchGains)
        {
          // The wave's first four slots: prepare trips, ONE factorisation for all of them (lane group = slot), complete trips.
          // Straight-line code with compile-time slot indices (the stash lives in registers; no rotation).  A fifth slot (groups
          // beyond 28 instances) takes the whole-wave factorisation below.
          constexpr int kB4 = kPerWave < 4 ? kPerWave : 4;
          // (Measured and not kept: a wave that owns ONE slot of the sweep taking the whole-wave factorisation instead of a batch of
          // one — 3 % at 256 instances, but with both paths in the loop the batched one lost what it had gained at 4096:
          // backward 11.6 -> 13.5 ms, profiles/r06_centroidal_gains_ab.txt.)
          GainStash st[kB4];
          S ga[MM], gy[MM];
#pragma unroll
          for(int k = 0; k < MM; k++)
          {
            ga[k] = 0; // (lane groups without a slot: m_lane = 0, nothing but zeros moves through their lanes)
            gy[k] = 0;
          }
          int m_lane = 0, m_hi = 0;
#pragma unroll
          for(int e = 0; e < kB4; e++)
          {
            const int a = mw + kOwnerWaves * e;
            if(a < n_act)
            {
              const int slot = uniform(actSlot(a));
              const S * rec = recAt(parity, hi - i, a, chunk, n_act);
              const int m = kDyn ? uniform(static_cast<int>(rec[mp.oM])) : MM;
              if(e == 0)
              {
                gainsPrepare<0>(V[e], st[e], ga, gy, m_lane, mp, rec, uniformD(slotF(sLambda, slot)), m);
              }
              else if(e == 1)
              {
                gainsPrepare<1>(V[e], st[e], ga, gy, m_lane, mp, rec, uniformD(slotF(sLambda, slot)), m);
              }
              else if(e == 2)
              {
                gainsPrepare<2>(V[e], st[e], ga, gy, m_lane, mp, rec, uniformD(slotF(sLambda, slot)), m);
              }
              else
              {
                gainsPrepare<3>(V[e], st[e], ga, gy, m_lane, mp, rec, uniformD(slotF(sLambda, slot)), m);
              }
              m_hi = m > m_hi ? m : m_hi;
            }
          }
          bool ok_lane = true;
          if(uniform(m_hi) > 0)
          {
            ok_lane = gainsBatch(ga, gy, m_lane);
          }
          const unsigned long long ok_groups = __ballot(ok_lane);
#pragma unroll
          for(int e = 0; e < kB4; e++)
          {
            const int a = mw + kOwnerWaves * e;
            if(a < n_act)
            {
              const int slot = uniform(actSlot(a));
              const S * rec = recAt(parity, hi - i, a, chunk, n_act);
              const int m = kDyn ? uniform(static_cast<int>(rec[mp.oM])) : MM;
              bool ok = ((ok_mask >> e) & 1u) != 0;
              const bool ok_now = ((ok_groups >> (16 * e)) & 1ull) != 0;
              const int b = uniform(slotI(sB, slot));
              if(e == 0)
              {
                gainsComplete<0>(V[e], st[e], gy, ok_now, ok, mp, la, rec, slot, b, i, m);
              }
              else if(e == 1)
              {
                gainsComplete<1>(V[e], st[e], gy, ok_now, ok, mp, la, rec, slot, b, i, m);
              }
              else if(e == 2)
              {
                gainsComplete<2>(V[e], st[e], gy, ok_now, ok, mp, la, rec, slot, b, i, m);
              }
              else
              {
                gainsComplete<3>(V[e], st[e], gy, ok_now, ok, mp, la, rec, slot, b, i, m);
              }
              ok_mask = ok ? ok_mask : (ok_mask & ~(1u << e));
              profAdd(4, 1, 1);
            }
          }
#pragma unroll
          for(int e = kB4; e < kPerWave; e++)
          {
            const int a = mw + kOwnerWaves * e;
            if(a < n_act)
            {
              const int slot = uniform(actSlot(a));
              bool ok = ((ok_mask >> e) & 1u) != 0;
              backwardStep(V[e], ok, mp, la, recAt(parity, hi - i, a, chunk, n_act), slot, uniform(slotI(sB, slot)), i,
                           uniformD(slotF(sLambda, slot)), e);
              ok_mask = ok ? ok_mask : (ok_mask & ~(1u << e));
              profAdd(4, 1, 1);
            }
          }
          continue;
        }
        if constexpr(kBatchQP)
        {
#pragma nounroll
          for(int e = 0; e < kPerWave; e++)
          {
            const int a = mw + kOwnerWaves * e;
            if(mw < kOwnerWaves && a < n_act)
            {
              const int slot = uniform(actSlot(a));
              qpPrepare(V0, mp, la, recAt(parity, hi - i, a, chunk, n_act), uniformD(slotF(sLambda, slot)), e);
            }
            rotate();
          }
          qpBatch(mp, i, parity, hi - i, chunk, n_act, ok_mask);
        }
        if constexpr(kUnrollSlots)
        {
          // small steps: a copy of the step per slot, the value functions addressed by constants, no rotation moves
#pragma unroll
          for(int e = 0; e < kPerWave; e++)
          {
            const int a = mw + kOwnerWaves * e;
            if(a < n_act)
            {
              const int slot = uniform(actSlot(a));
              bool ok = ((ok_mask >> e) & 1u) != 0;
              backwardStep(V[e], ok, mp, la, recAt(parity, hi - i, a, chunk, n_act), slot, uniform(slotI(sB, slot)), i,
                           uniformD(slotF(sLambda, slot)), e);
              ok_mask = ok ? ok_mask : (ok_mask & ~(1u << e));
            }
          }
          continue;
        }
#pragma nounroll
        for(int e = 0; e < kPerWave; e++)
        {
          const int a = mw + kOwnerWaves * e;
          if(mw < kOwnerWaves && a < n_act)
          {
            const int slot = uniform(actSlot(a));
            bool ok = ((ok_mask >> e) & 1u) != 0;
            backwardStep(V0, ok, mp, la, recAt(parity, hi - i, a, chunk, n_act), slot, uniform(slotI(sB, slot)), i,
                         uniformD(slotF(sLambda, slot)), e);
            ok_mask = ok ? ok_mask : (ok_mask & ~(1u << e));
            profAdd(4, 1, 1);
          }
          rotate();
        }
      }
      const unsigned long long pb = profNow();
      profAdd(2, pb - pa, 1);
      profAdd(11, pb - pa, 5);
      barrier(); // this chunk's records have been read by all, the next chunk's are complete
      profAdd(3, profNow() - pb, 1);
      profAdd(16 + wave, pb - pa, wave); // every matrix wave: its steps of this chunk ...
      profAdd(24 + wave, profNow() - pb, wave); // ... and its wait at the barrier
    }
    if(lane == kStarLane)
    {
#pragma unroll
      for(int e = 0; e < kPerWave; e++)
      {
        const int a = mw + kOwnerWaves * e;
        if(mw < kOwnerWaves && a < n_act)
        {
          slotI(sOk, actSlot(a)) = static_cast<int>((ok_mask >> e) & 1u);
        }
      }
    }
  }

  /** The model wave's half of the sweep.  A lane linearises ONE (slot, timestep): with n_act slots in the sweep a pass of the
      model code covers a CHUNK of timesteps, lane = dt * n_act + a — as many as the lanes (64) and the record area (two
      buffers of chunk x n_act records) hold.  A full group of 32 is a chunk of one timestep (lane = slot, records of timestep
      i - 1 written while the matrix waves consume timestep i).  A sweep that few slots need — small batches, the regularisation
      retries, the last iterations of a batch, when most slots have converged — covers up to the whole horizon in one pass: its
      timestep costs the matrix waves' step, not a pass of the model code per timestep.  (x, u) of the next chunk are requested
      behind the linearisation of this one, and complete while the wave waits at the chunk's barrier.
      ceil(T / chunk) + 3 barriers, as backwardSweepMatrix. */
  NMPC_D void backwardSweepModel(int group) const
  {
    const int n_act = uniform(meta(mNAct)), chunk = uniform(meta(mChunk));
    const int a = lane % n_act, dt = lane / n_act;
    const bool lane_used = dt < chunk; // (n_act * chunk <= 64)
    const int slot = actSlot(lane_used ? a : 0);
    const int b = group * G + slot;
    const int sel = slotI(sSel, slot);
    const S t0 = slotF(sT0, slot);
    const Problem mine_p = problemOf(lane_used ? b : group * G + actSlot(0));
    Point next;
    if(lane_used && T - 1 - dt >= 0)
    {
      loadPoint(next, b, sel, T - 1 - dt);
    }
    if(lane_used && dt == 0)
    {
      lineariseTerminal(mine_p, slot, b, sel, t0);
    }
    barrier(); // the terminal records are complete
    barrier(); // (the matrix waves have taken them: the record area is free)
    profAdd(9, 1, 0);
    int parity = 0, n_chunk = 0;
    for(int hi = T - 1; hi >= 0; hi -= chunk, parity ^= 1, n_chunk++)
    {
      const unsigned long long pa = profNow();
      const int step = hi - dt;
      const bool mine = lane_used && step >= 0;
      if(mine)
      {
        S * dst = recAt(parity, dt, a, chunk, n_act);
        // The first use of a record buffer in the sweep: every entry.  float: ONE copy of the functors' code with a run-time
        // flag (the chunk count through an opaque copy, or the compiler peels the first two trips into copies of their own):
        // two copies differ in which products they fuse, and the results then depend on the chunk, i.e. the group size, in
        // the last bit.  double: the two copies agree bit for bit (tests, soak) — and with one copy the centroidal
        // problem's rollouts take 10.1 instead of 6.6 ms at 4096 instances (measured A/B on one box, round 4; a side effect
        // of the register allocation that was not tracked down), so double keeps two.
#ifdef NMPC_AMD_AB_ONE_LIN_COPY
        if constexpr(true)
#else
        if constexpr(kF32)
#endif
        {
          int nc = n_chunk;
          asm volatile("" : "+s"(nc));
          lineariseStep<false>(mine_p, dst, t0, step, next, nc < 2);
        }
        else if(n_chunk < 2)
        {
          lineariseStep<true>(mine_p, dst, t0, step, next, true);
        }
        else
        {
          lineariseStep<false>(mine_p, dst, t0, step, next, false);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if(lane_used && step - chunk >= 0)
      {
        loadPoint(next, b, sel, step - chunk);
      }
      const unsigned long long pb = profNow();
      profAdd(0, pb - pa, 0);
      barrier(); // this chunk's records are complete (and the previous chunk's have been read by all)
      profAdd(1, profNow() - pb, 0);
    }
    barrier(); // the last chunk has been read
  }

  // ===================================================================================================
  // solve    DDPSolver.hpp:26-141, procOnce :143-340
  // ===================================================================================================
  /** Kernel start: the offset table (lane 0 of the model wave), the record stride and the group size that follow from it. */
  NMPC_D void setup()
  {
    if(wave == 0 && lane == 0)
    {
      // the point is irrelevant (only what the compiler knows about each entry matters) but must not be a constant
      Point p;
      loadPoint(p, 0, 0, 0);
      StateDimVector x;
      InputDimVector u;
#pragma unroll
      for(int c = 0; c < N; c++)
      {
        x[c] = p.x[c];
      }
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        u[a] = p.u[a];
      }
      const Problem mine = problemOf(0);
      TableSink sink{tbl()};
      const S t_probe = buf.t0 ? buf.t0[0] : 0.0;
      const int m_probe = inputDimOf(mine, t_probe);
      if constexpr(kDyn)
      {
        u.resize(m_probe);
      }
      emitRecord(mine, t_probe, x, u, m_probe, sink);
      int s = sink.cnt + 1; // + the zero word
      s |= 1; // odd: the model wave's lanes (one record each) spread over the LDS banks
      const int room = kLdsDoubles - kRecAt;
      int per_instance = (2 * s > kTerm) ? 2 * s : kTerm;
      per_instance = (per_instance > kRingDepth * kRingStride) ? per_instance : kRingDepth * kRingStride; // (line search: nominal ring)
      int g = room / per_instance;
      g = g > kGroupMax ? kGroupMax : g;
      // every workgroup gets work, and the same amount: rounds = passes a workgroup makes over its groups with the largest group
      // the LDS holds; the groups are then as small as that number of rounds allows (8200 instances on 256 CUs: two rounds
      // of 17 instead of a round of 32 and one straggler group)
      const int wgs = static_cast<int>(gridDim.x);
      const int rounds = (buf.B + wgs * g - 1) / (wgs * g);
      const int spread = (buf.B + rounds * wgs - 1) / (rounds * wgs);
      g = g > spread ? spread : g;
      if(group_cap > 0 && g > group_cap)
      {
        g = group_cap;
      }
      meta(mStride) = s;
      meta(mGroup) = g < 1 ? 1 : g;
    }
    barrier();
    stride = uniform(meta(mStride));
    G = uniform(meta(mGroup));
  }

  /** The trace row of the iteration in progress lives in LDS (field-major: the owner lanes spread over the banks), not in
      twelve registers that would be live across every rollout and linearisation. */
  NMPC_D S & trF(int f, int slot) const
  {
    return lds[kTraceAt + f * kT64MaxGroup + slot];
  }
  NMPC_D void clearTrace(int slot) const
  {
#pragma unroll
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      trF(f, slot) = 0;
    }
  }
  NMPC_D void writeTraceRow(int b, int row, int slot) const
  {
    if(cfg.trace_level >= 1 && row < buf.trace_rows)
    {
      const size_t tile = tileOf(b), ln = lnOf(b);
      S * p = buf.trace + (tile * (static_cast<size_t>(buf.trace_rows) * NMPC_HIP_NTRACE)) * 64 + ln;
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        p[(static_cast<size_t>(row) * NMPC_HIP_NTRACE + f) * 64] = trF(f, slot);
      }
    }
  }

  /** The slots of the coming backward sweep as a list (model wave, lanes = slots; `takes_part` balloted into `mask`), and the
      chunk of timesteps the model wave linearises per pass: as many as its 64 lanes and two record buffers allow. */
  NMPC_D void listActive(bool takes_part, unsigned long long mask, int slot) const
  {
    const int a = __popcll(mask & ((1ull << lane) - 1ull));
    const int n_act = __popcll(mask);
    actIndex(slot) = takes_part ? a : -1;
    if(takes_part)
    {
      actSlot(a) = slot;
    }
    if(lane == 0)
    {
      int chunk = 1;
      if(n_act > 0)
      {
        const int by_lanes = 64 / n_act, by_records = recordCapacity() / (2 * n_act);
        chunk = by_lanes < by_records ? by_lanes : by_records;
        chunk = chunk > T ? T : chunk;
        chunk = chunk < 1 ? 1 : chunk;
      }
      if(chunk_cap > 0 && chunk > chunk_cap)
      {
        chunk = chunk_cap;
      }
      meta(mNAct) = n_act;
      meta(mChunk) = chunk;
    }
  }

  /** One group of instances, all eight waves.  The phases are separated by barriers; wave 0 ("model wave", lane = slot) runs the
      solver state machine of DDPSolver::solve / procOnce for its slot in short blocks between them — its state lives in the
      slot table — and the model code; waves 1 .. 7 run the backward sweeps and help in the line search. */
  NMPC_D void solveGroup(int group)
  {
    const unsigned long long solve_start = __builtin_readcyclecounter();
    const bool model_wave = (wave == 0);
    const int slot = lane < kT64MaxGroup ? lane : 0;
    const int b = group * G + slot;
    const bool owner = model_wave && lane < G && b < buf.B; // this lane drives an instance
    const bool slot_lane = model_wave && lane < kT64MaxGroup;
    const int p_lane_matrix = (wave - 1) * 64 + lane, p_count_matrix = kT64MatrixWaves * 64;
    S * lsJ = lds + kLsAt;

    // ---- solve(): reset, initial rollout    :36-38, :83-104
    if(slot_lane)
    {
      slotI(sB, slot) = owner ? b : -1;
      slotI(sBw, slot) = 0;
      slotI(sLs, slot) = 0;
      slotI(sSel, slot) = 0;
      slotI(sOk, slot) = 1;
      slotI(sIter, slot) = 0;
      slotI(sRet, slot) = 0;
      slotI(sFlags, slot) = owner ? fRunning : 0;
      slotF(sLambda, slot) = sc(cfg.initial_lambda);
      slotF(sDlambda, slot) = sc(cfg.initial_dlambda);
      slotF(sT0, slot) = (owner && buf.t0) ? buf.t0[b] : 0.0;
      slotF(sDV0, slot) = 0;
      slotF(sDV1, slot) = 0;
      slotT(sTicksBw, slot) = 0;
      slotT(sTicksFw, slot) = 0;
      clearTrace(slot);
      if(lane == 0)
      {
        meta(mWide) = 0;
        meta(mRejected) = 0;
      }
    }
    barrier();

    // Every trip of this loop ends with the rollouts of one procOnce (the line search); the first trip has nothing before them
    // and rolls out initial_u_list instead (pass 0) — so that the kernel holds ONE copy of the rollout code.
    bool first_trip = true;
    for(;;)
    {
      bool rollouts = true; // (first trip: the initial rollout)
      if(!first_trip)
      {
      // ---- which slots start procOnce number iter + 1    :115-123
      if(slot_lane)
      {
        int flags = slotI(sFlags, slot);
        const int iter = slotI(sIter, slot);
        const bool in_iter = (flags & fRunning) != 0 && iter < cfg.max_iter;
        flags = in_iter ? (fRunning | fInIter | fNeedBw) : 0;
        if(in_iter)
        {
          slotI(sIter, slot) = iter + 1;
          slotI(sRet, slot) = 0;
          slotI(sNBw, slot) = 0;
          clearTrace(slot);
          trF(NMPC_HIP_TRACE_ITER, slot) = static_cast<S>(iter + 1);
          trF(NMPC_HIP_TRACE_ALPHA_IDX, slot) = -1;
        }
        slotI(sFlags, slot) = flags;
        slotI(sBw, slot) = in_iter ? 1 : 0;
        const unsigned long long any = __ballot(in_iter);
        listActive(in_iter, any, slot);
        if(lane == 0)
        {
          meta(mAnyIter) = (any != 0) ? 1 : 0;
        }
      }
      publishBarrier(); // B1 (the initial rollout's / a re-rolled step size's trajectory is in X / U)
      if(uniform(meta(mAnyIter)) == 0)
      {
        break;
      }
      // ---- Steps 1 + 2: linearisation fused into the backward sweep, with the regularisation retries    :157-214
      for(;;)
      {
        const unsigned long long p0 = __builtin_readcyclecounter();
        if(model_wave)
        {
          backwardSweepModel(group);
        }
        else
        {
          backwardSweepMatrix();
        }
        publishBarrier(); // B2: results of the sweep are in the slot table, the gains in kff / Kfb
        if(slot_lane)
        {
          int flags = slotI(sFlags, slot);
          bool retry = false;
          if((flags & fNeedBw) != 0)
          {
            slotI(sNBw, slot) += 1;
            slotT(sTicksBw, slot) += __builtin_readcyclecounter() - p0;
            if(slotI(sOk, slot) == 0)
            {
              const S dlambda = fmax(slotF(sDlambda, slot) * sc(cfg.lambda_factor), sc(cfg.lambda_factor)); // :191-209
              const S lambda = fmax(slotF(sLambda, slot) * dlambda, sc(cfg.lambda_min));
              slotF(sDlambda, slot) = dlambda;
              slotF(sLambda, slot) = lambda;
              if(lambda > sc(cfg.lambda_max))
              {
                slotI(sRet, slot) = -1;
                flags &= ~fNeedBw;
              }
              else
              {
                retry = true;
              }
            }
            else
            {
              flags &= ~fNeedBw;
            }
            slotI(sFlags, slot) = flags;
          }
          slotI(sBw, slot) = retry ? 1 : 0;
          const unsigned long long any = __ballot(retry);
          listActive(retry, any, slot);
          if(lane == 0)
          {
            meta(mAnyRetry) = (any != 0) ? 1 : 0;
          }
        }
        barrier(); // B3
        if(uniform(meta(mAnyRetry)) == 0)
        {
          break;
        }
      }
      // ---- small-gradient termination (:217-231); who searches    :234-274
      if(slot_lane)
      {
        int flags = slotI(sFlags, slot);
        bool in_ls = false;
        if((flags & fInIter) != 0)
        {
          trF(NMPC_HIP_TRACE_N_BACKWARD, slot) = static_cast<S>(slotI(sNBw, slot));
          if(slotI(sRet, slot) == 0)
          {
            const S krel = kKrelSquared ? sqrt(slotF(sKrel, slot)) : slotF(sKrel, slot);
            trF(NMPC_HIP_TRACE_K_REL_NORM, slot) = krel;
            if(krel < sc(cfg.k_rel_norm_thre) && slotF(sLambda, slot) < sc(cfg.lambda_thre))
            {
              slotI(sRet, slot) = 1;
            }
            else
            {
              in_ls = true;
            }
          }
        }
        flags = in_ls ? (flags | fInLs) : (flags & ~fInLs);
        slotI(sFlags, slot) = flags & ~fSuccess;
        slotI(sLs, slot) = in_ls ? 1 : 0;
        slotI(sAi, slot) = cfg.n_alpha - 1;
        const unsigned long long any = __ballot(in_ls);
        if(lane == 0)
        {
          meta(mAnyLs) = (any != 0) ? 1 : 0;
          // the later step sizes ride along with the first one when the waves that roll them out have a SIMD to themselves
          // (three waves cover the group), or share it among themselves (six waves) and the group's previous search made
          // a quarter of its slots go beyond the first step size: the pass then takes as long as two, and saves a third one
          const int waves_needed = (G + laterPerWave() - 1) / laterPerWave();
          meta(mWide) = (cfg.n_alpha > 1 && wide_cap != 0 && (waves_needed <= 3 || (waves_needed <= kSpecWaves && meta(mRejected) != 0))) ? 1 : 0;
        }
      }
      barrier(); // B4
      rollouts = uniform(meta(mAnyLs)) != 0;
      }
      if(rollouts)
      {
        // Step 3, the line search.  Pass 1: the first step size — the one normally taken — lane = slot on the model wave, stored
        // (and, groups of up to 32, the second step size for its cost in the wave's upper lanes: `pair` below);
        // the matrix waves feed the ring.  Pass 2 (only if a slot rejected it): the later step sizes of those slots all at once
        // on the matrix waves, lane = (slot, step size), cost only; the model wave feeds the ring.  Pass 3 (only if a later
        // step size was taken): its trajectory, stored.  The first accepted step size in list order is the sequential loop's
        // choice (:242-265).  One call site of stagedPass: one copy of the model's rollout code in the kernel.
        const unsigned long long p0 = __builtin_readcyclecounter();
        /** judge step size ai of this lane's slot with cost Jc    :247-264 */
        auto judge = [&](int ai, S Jc) -> bool
        {
          const S alpha = sc(cfg.alpha_list[ai]);
          const S actual = slotF(sJcur, slot) - Jc;
          const S expected = -1 * alpha * (slotF(sDV0, slot) + alpha * slotF(sDV1, slot));
          S ratio = actual / expected;
          if(expected < 0)
          {
            ratio = (actual >= 0 ? 1 : -1); // :251-259
          }
          slotF(sAlpha, slot) = alpha;
          slotF(sActual, slot) = actual;
          slotF(sExpected, slot) = expected;
          slotF(sRatio, slot) = ratio;
          slotF(sJcand, slot) = Jc;
          slotI(sAi, slot) = ai;
          return ratio > sc(cfg.cost_update_ratio_thre);
        };
        const int later_per_wave = laterPerWave();
        const int covered = later_per_wave * kT64MatrixWaves;
        const bool wide = !first_trip && uniform(meta(mWide)) != 0;
        const int spec_index = specIndex(wave); // (eight waves: 1 2 3 5 6 7 -> 0 .. 5; wave 4 shares the model wave's SIMD: -1)
        const int waves_rolling = (G + later_per_wave - 1) / later_per_wave; // (a wide pass 1: <= kSpecWaves)
        // the later step sizes' trajectories go to the workspace, and the one that is taken is copied from there (no pass 3)
        const bool adopt = adopt_cap != 0 && cfg.n_alpha - 1 <= kScratchAlphas;
        // A narrow pass 1 uses the model wave's lanes 0 .. G - 1; with G <= 32 its lanes 32 .. 32 + G - 1 roll out the SECOND step size
        // of the same slots beside them, for its cost (same ring entries, same instruction stream, no stores: no extra time): a
        // slot that rejects the first step size and takes the second — most of the back-tracking of a solve far from the noise
        // floor — needs no pass 2, only pass 3 (its trajectory, stored; a third of a pass 2's time).  Measured with the second step
        // size's trajectory going to the workspace for adoption instead: c4 the same, c5 / c4f64 2 - 3 % slower (scattered stores in
        // every search for a copy few searches need).  (A launch lasts as long as its slowest group, and with 256 groups some slot of
        // some group rejects the first step size in nearly every iteration: DESIGN.md §5.)
        const bool pair = !first_trip && !wide && pair_cap != 0 && cfg.n_alpha > 1 && G <= 32;
        int pass = first_trip ? 0 : 1, trip_base = 0;
#pragma nounroll
        for(;;)
        {
          // ---- what this lane does in this pass
          bool compute, active = false, store = true;
          int pb = 0, pinst = 0, phalf = 0, pai = 0;
          S pt0 = 0, palpha = 0;
          if(pass == 0)
          {
            compute = model_wave;
            active = owner;
            pb = owner ? b : 0;
            pinst = slot;
            pt0 = slotF(sT0, slot);
          }
          else if(pass != 2 && (model_wave || !(pass == 1 && wide)))
          {
            compute = model_wave;
            if(slot_lane && slotI(sLs, slot) != 0)
            {
              active = true;
              pb = b;
              pinst = slot;
              phalf = slotI(sSel, slot) ^ 1;
              pt0 = slotF(sT0, slot);
              palpha = (pass == 1) ? sc(cfg.alpha_list[0]) : slotF(sAlpha, slot);
            }
            else if(pair && pass == 1 && model_wave && lane >= 32 && lane - 32 < G)
            {
              const int s2 = lane - 32;
              if(slotI(sB, s2) >= 0 && slotI(sLs, s2) != 0)
              {
                active = true;
                store = false;
                pb = slotI(sB, s2);
                pinst = s2;
                pai = 1;
                phalf = slotI(sSel, s2) ^ 1;
                pt0 = slotF(sT0, s2);
                palpha = sc(cfg.alpha_list[1]);
              }
            }
          }
          else
          {
            // pass 2 — or a wide pass 1: waves 1 - 3, then 5 - 7 roll out the later step sizes (wave 4, which shares its SIMD with
            // the model wave, feeds the ring)
            const int roll_index = (pass == 2) ? wave - 1 : spec_index;
            compute = (pass == 2) ? !model_wave : (spec_index >= 0 && spec_index < waves_rolling);
            store = false;
            if(compute)
            {
              const int n_later = cfg.n_alpha - 1;
              const int inst = lane / n_later;
              pai = 1 + lane - inst * n_later;
              const int fslot = trip_base + roll_index * later_per_wave + inst;
              if(inst < later_per_wave && fslot < G && slotI(sB, fslot) >= 0 && slotI(sLs, fslot) != 0)
              {
                active = true;
                pb = slotI(sB, fslot);
                pinst = fslot;
                phalf = slotI(sSel, fslot) ^ 1;
                pt0 = slotF(sT0, fslot);
                palpha = sc(cfg.alpha_list[pai]);
              }
            }
          }
          const Problem theirs = active ? problemOf(pb) : problem;
          // who feeds the ring: all matrix waves (passes 1, 3), the model wave (pass 2); a wide pass 1: the matrix waves that
          // do not roll out — wave 4 and, while three waves cover the group, waves 5 - 7
          int p_lane = model_wave ? lane : p_lane_matrix, p_count = model_wave ? 64 : p_count_matrix;
          if((pass == 1 || pass == 3) && !wide && !model_wave)
          {
            // the rolling wave's SIMD is the rolling wave's: wave 4, which shares it, stays out of the prefetch (its rows were a
            // tenth of the model wave's issue slots)
            const int order = ringOrder(wave); // 1 2 3 5 6 7 | 4
            p_lane = order * 64 + lane; // (>= p_count for wave 4: no rows)
            p_count = kSpecWaves * 64;
          }
          if(pass == 1 && wide && !model_wave)
          {
            const int order = ringOrder(wave); // 1 2 3 5 6 7 4
            p_lane = (order - waves_rolling) * 64 + lane; // (the waves that roll out do not prefetch: compute is set)
            p_count = (kT64MatrixWaves - waves_rolling) * 64;
          }
          // (a later step size, rolled out for its cost; the model wave's upper lanes of a pair pass: cost only)
          const bool to_workspace = adopt && !store && compute && active && !model_wave;
          const S Jc = stagedPass(compute, pass == 0, theirs, active, group, pb, pinst, phalf, pt0, palpha, store || to_workspace,
                                       p_lane, p_count, to_workspace ? pai - 1 : -1);
          // ---- what follows from it
          if(pass == 0)
          {
            if(owner) // :98-104
            {
              slotF(sJcur, slot) = Jc;
              slotT(sTicksFw, slot) += __builtin_readcyclecounter() - p0;
              trF(NMPC_HIP_TRACE_COST, slot) = Jc;
              trF(NMPC_HIP_TRACE_LAMBDA, slot) = sc(cfg.initial_lambda);
              trF(NMPC_HIP_TRACE_DLAMBDA, slot) = sc(cfg.initial_dlambda);
              trF(NMPC_HIP_TRACE_ALPHA_IDX, slot) = -1;
              writeTraceRow(b, 0, slot);
            }
            break;
          }
          else if(pass == 1)
          {
            if(wide)
            {
              if(!model_wave && active)
              {
                lsJ[pai * kT64MaxGroup + pinst] = Jc;
              }
              barrier(); // B5a: the later step sizes' costs are in lsJ
            }
            else if(pair)
            {
              if(model_wave && lane >= 32 && active)
              {
                lsJ[1 * kT64MaxGroup + pinst] = Jc;
              }
              barrier(); // B5a': the second step size's costs are in lsJ
            }
            if(slot_lane)
            {
              int flags = slotI(sFlags, slot);
              const bool searching = (flags & fInLs) != 0;
              bool more = false, reroll = false, rejected = false;
              if(searching)
              {
                bool success = judge(0, Jc);
                flags = success ? (flags | fSuccess) : flags;
                rejected = !success && cfg.n_alpha > 1;
                if(rejected && wide)
                {
                  for(int ai = 1; ai < cfg.n_alpha && !success; ai++)
                  {
                    success = judge(ai, lsJ[ai * kT64MaxGroup + slot]);
                  }
                  if(success)
                  {
                    flags |= fSuccess;
                    reroll = true; // its trajectory has not been stored yet
                  }
                }
                else if(rejected && pair)
                {
                  success = judge(1, lsJ[1 * kT64MaxGroup + slot]);
                  if(success)
                  {
                    flags |= fSuccess;
                    reroll = true; // its trajectory has not been stored yet
                  }
                  else
                  {
                    more = cfg.n_alpha > 2; // (pass 2 judges from the second step size on again: the same costs)
                  }
                }
                else
                {
                  more = rejected;
                }
                slotI(sFlags, slot) = flags;
              }
              slotI(sLs, slot) = (more || reroll) ? 1 : 0;
              const unsigned long long any_more = __ballot(more), any_reroll = __ballot(reroll);
              const int n_rejected = __popcll(__ballot(rejected)), n_searching = __popcll(__ballot(searching));
              if(lane == 0)
              {
                meta(mAnyMore) = (any_more != 0) ? 1 : 0;
                meta(mAnyReroll) = (any_reroll != 0) ? 1 : 0;
                meta(mRejected) = (n_rejected > 0 && 4 * n_rejected >= n_searching) ? 1 : 0;
              }
            }
            publishBarrier(); // B5: the candidate trajectory is in the other half of X / U / cost
            if(uniform(meta(mAnyMore)) != 0)
            {
              pass = 2;
              trip_base = 0;
            }
            else if((wide || pair) && uniform(meta(mAnyReroll)) != 0)
            {
              if(adopt && wide)
              {
                adoptCandidates(group);
#ifndef NMPC_AMD_AB_REOPEN_ADOPT_RACE // (test builds only: the race of commit 2b8d598 re-opened, for the fuzz experiment)
                barrier(); // (every wave has read the slot table: Step 4 below flips sSel)
#endif
                break;
              }
              pass = 3;
            }
            else
            {
              break;
            }
          }
          else if(pass == 2)
          {
            if(active)
            {
              lsJ[pai * kT64MaxGroup + pinst] = Jc;
            }
            trip_base += covered;
            if(trip_base < G)
            {
              continue;
            }
            barrier(); // B5b: the later step sizes' costs are in lsJ
            if(slot_lane)
            {
              bool reroll = false;
              if(slotI(sLs, slot) != 0)
              {
                bool success = false;
                for(int ai = 1; ai < cfg.n_alpha && !success; ai++)
                {
                  success = judge(ai, lsJ[ai * kT64MaxGroup + slot]);
                }
                if(success)
                {
                  slotI(sFlags, slot) |= fSuccess;
                  reroll = true; // its trajectory has not been stored yet
                }
              }
              slotI(sLs, slot) = reroll ? 1 : 0;
              const unsigned long long any = __ballot(reroll);
              if(lane == 0)
              {
                meta(mAnyReroll) = (any != 0) ? 1 : 0;
              }
            }
            publishBarrier(); // B5c (the later step sizes' trajectories are in the workspace)
            if(uniform(meta(mAnyReroll)) == 0)
            {
              break;
            }
            if(adopt)
            {
              adoptCandidates(group);
#ifndef NMPC_AMD_AB_REOPEN_ADOPT_RACE
              barrier(); // (every wave has read the slot table: Step 4 below flips sSel)
#endif
              break;
            }
            pass = 3;
          }
          else
          {
            if(active)
            {
              slotF(sJcand, slot) = Jc; // (the same arithmetic on the same inputs as the lane that summed its cost)
            }
            break;
          }
        }
        // ---- Step 4    :280-333
        if(!first_trip && slot_lane && (slotI(sFlags, slot) & fInLs) != 0)
        {
          const bool success = (slotI(sFlags, slot) & fSuccess) != 0;
          const int ai_taken = slotI(sAi, slot);
          S lambda = slotF(sLambda, slot), dlambda = slotF(sDlambda, slot);
          slotT(sTicksFw, slot) += __builtin_readcyclecounter() - p0;
          trF(NMPC_HIP_TRACE_ALPHA, slot) = slotF(sAlpha, slot);
          trF(NMPC_HIP_TRACE_COST_UPDATE_ACTUAL, slot) = slotF(sActual, slot);
          trF(NMPC_HIP_TRACE_COST_UPDATE_EXPECTED, slot) = slotF(sExpected, slot);
          trF(NMPC_HIP_TRACE_COST_UPDATE_RATIO, slot) = slotF(sRatio, slot);
          trF(NMPC_HIP_TRACE_ALPHA_IDX, slot) = static_cast<S>(ai_taken);
          trF(NMPC_HIP_TRACE_N_FORWARD, slot) = static_cast<S>(success ? ai_taken + 1 : cfg.n_alpha);
          if(success)
          {
            slotI(sSel, slot) ^= 1;
            slotF(sJcur, slot) = slotF(sJcand, slot);
            if(slotF(sActual, slot) < sc(cfg.cost_update_thre))
            {
              slotI(sRet, slot) = 1;
            }
            dlambda = fmin(dlambda / sc(cfg.lambda_factor), 1 / sc(cfg.lambda_factor));
            if(lambda >= sc(cfg.lambda_min))
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
            dlambda = fmax(dlambda * sc(cfg.lambda_factor), sc(cfg.lambda_factor));
            lambda = fmax(lambda * dlambda, sc(cfg.lambda_min));
            if(lambda > sc(cfg.lambda_max))
            {
              slotI(sRet, slot) = -1;
            }
          }
          slotF(sLambda, slot) = lambda;
          slotF(sDlambda, slot) = dlambda;
          trF(NMPC_HIP_TRACE_COST, slot) = slotF(sJcur, slot);
          trF(NMPC_HIP_TRACE_LAMBDA, slot) = lambda;
          trF(NMPC_HIP_TRACE_DLAMBDA, slot) = dlambda;
        }
      }
      if(!first_trip && slot_lane && (slotI(sFlags, slot) & fInIter) != 0)
      {
        if(owner)
        {
          writeTraceRow(b, slotI(sIter, slot), slot);
        }
        if(slotI(sRet, slot) != 0)
        {
          slotI(sFlags, slot) &= ~fRunning; // :118-122
        }
      }
      first_trip = false;
    }

    // ---- results the host reads per instance
    if(owner)
    {
      const size_t tile = tileOf(b), ln = lnOf(b);
      buf.status[b] = slotI(sRet, slot);
      buf.iters[b] = slotI(sIter, slot);
      buf.sel[b] = slotI(sSel, slot);
      if(buf.phase_ticks != nullptr)
      {
        unsigned long long * p = buf.phase_ticks + static_cast<size_t>(b) * 4;
        p[0] = slotT(sTicksBw, slot);
        p[1] = slotT(sTicksFw, slot);
        p[2] = __builtin_readcyclecounter() - solve_start;
      }
      buf.dV[(tile * 2 + 0) * 64 + ln] = slotF(sDV0, slot);
      buf.dV[(tile * 2 + 1) * 64 + ln] = slotF(sDV1, slot);
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        buf.trace_last[(tile * NMPC_HIP_NTRACE + f) * 64 + ln] = trF(f, slot);
      }
      const Problem mine_o = problemOf(b);
      for(int i = 0; i < T; i++)
      {
        buf.input_dim[(tile * T + i) * 64 + ln] = inputDimOf(mine_o, slotF(sT0, slot) + i * mine_o.dt());
      }
    }
    barrier(); // the next group reuses the slot table and the record area
  }

  NMPC_D void run()
  {
    if(wave != 0 && lane < 18)
    {
      waveScratch(0)[wZero + lane] = 0.0; // zero words (read by lanes outside a block) and dump words (written by them)
    }
#ifdef NMPC_AMD_PROFILE_TILE64
    if(threadIdx.x < 40)
    {
      reinterpret_cast<unsigned long long *>(lds + kProfAt)[threadIdx.x] = 0;
    }
#endif
    setup();
#ifdef NMPC_AMD_PROFILE_TILE64
    if(threadIdx.x == 0)
    {
      // (counters are flushed >> 4): record stride, group size, doubles of LDS in front of the records
      reinterpret_cast<unsigned long long *>(lds + kProfAt)[37] = static_cast<unsigned long long>(stride) << 4;
      reinterpret_cast<unsigned long long *>(lds + kProfAt)[38] = static_cast<unsigned long long>(G) << 4;
      reinterpret_cast<unsigned long long *>(lds + kProfAt)[39] = static_cast<unsigned long long>(kRecAt) << 4;
    }
#endif
    const int n_groups = (buf.B + G - 1) / G;
    for(int group = static_cast<int>(blockIdx.x); group < n_groups; group += static_cast<int>(gridDim.x))
    {
      solveGroup(group);
    }
#ifdef NMPC_AMD_PROFILE_TILE64
    if(blockIdx.x == 0 && threadIdx.x < 40)
    {
      // counter k -> row k % T of instance k / T (instances 0, 1 of tile 0)
      buf.qp_free[static_cast<size_t>(threadIdx.x % T) * 64 + threadIdx.x / T] =
          static_cast<unsigned>(reinterpret_cast<unsigned long long *>(lds + kProfAt)[threadIdx.x] >> 4);
    }
#endif
  }
};

/** The fp64 tile kernel: persistent workgroups of eight wavefronts, grid = number of CUs (or fewer for small batches). */
template<class Problem, bool kConstrained, bool kOwnProblem>
__global__ __launch_bounds__(T64Waves<Problem>::value * 64) void ddp_solve_tile64_kernel(const Problem problem,
                                                                        const nmpc_hip_ddp_config cfg,
                                                                        const DeviceBuffersT<typename Problem::Scalar> buf,
                                                                        const int group_cap)
{
  // (a STATIC array: the address of a dynamic one — extern __shared__ — is a symbol until after instruction selection, and
  // every LDS access of the kernel carried a leftover `v_add_u32 v, 0, v`: twenty of them per backward step)
  __shared__ __attribute__((aligned(16))) typename Problem::Scalar lds_tile64[kT64LdsBytes / sizeof(typename Problem::Scalar)];
  TileSolver64<Problem, kConstrained, kOwnProblem> solver(problem, cfg, buf, lds_tile64, group_cap);
  solver.run();
}

/** Launch helper for model_registry.hpp. */
template<class Problem, bool kConstrained, bool kOwnProblem>
inline hipError_t launchTile64(const Problem & problem, const nmpc_hip_ddp_config & cfg, const DeviceBuffersT<typename Problem::Scalar> & buf,
                               hipStream_t stream)
{
  // per device: 0 until the attribute is set and the CU count known (published last, with release order: handles of
  // several host threads may launch at once — DDPSolverPool, DDPSolverSharded; a second thread repeats the harmless setup)
  static std::atomic<int> n_cu[64] = {};
  int dev = 0;
  if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
  {
    return hipErrorInvalidDevice;
  }
  int cus = n_cu[dev].load(std::memory_order_acquire);
  if(cus == 0)
  {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if(e != hipSuccess)
    {
      return e;
    }
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    n_cu[dev].store(cus, std::memory_order_release);
  }
  // the handle's knobs (LaunchKnobs; the NMPC_HIP_DDP_TILE64_* variables are developer overrides read when the handle is created):
  // group: at most g instances per group (tests: full groups on small batches); chunk: at most c timesteps per pass of the model code
  // (1: round 3's schedule); pair = 0: the second step size does not ride in the model wave's upper lanes; adopt = 0: a later step
  // size that is taken is re-rolled (round 3's pass 3); wide = 0: line-search passes as in round 3 (first step size, then the others)
  const LaunchKnobs knobs = launchKnobs();
  const int cap = knobs.tile64_group, chunk_cap = knobs.tile64_chunk;
  const unsigned no_pair = knobs.tile64_pair ? 0u : 1u, no_adopt = knobs.tile64_adopt ? 0u : 1u, no_wide = knobs.tile64_wide ? 0u : 1u;
  int grid = cus;
  if(cap > 0)
  {
    const int groups = (buf.B + cap - 1) / cap; // (the kernel may still choose smaller groups: idle workgroups exit)
    grid = groups < grid ? (groups < 1 ? 1 : groups) : grid;
  }
  grid = buf.B < grid ? buf.B : grid;
  hipLaunchKernelGGL((ddp_solve_tile64_kernel<Problem, kConstrained, kOwnProblem>), dim3(grid), dim3(T64Waves<Problem>::value * 64), 0, stream, problem,
                     cfg, buf, static_cast<int>(static_cast<unsigned>(cap) | (static_cast<unsigned>(chunk_cap) << 16) | (no_wide << 31) | (no_adopt << 30) | (no_pair << 29)));
  // (the kernel's 160 KB of LDS are a static array)
  return hipGetLastError();
}
} // namespace hip
} // namespace nmpc_amd
