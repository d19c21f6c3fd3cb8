// gfx950 device code of the batched DDP solver, LANE MAPPING "WPI": ONE WAVEFRONT PER PROBLEM INSTANCE, for the shapes
// whose blocks fill a matrix-core tile (9 <= n <= 16, m <= 16; BASELINE.json's quadrotor n 12 m 4 and manipulator
// n 14 m 7, the reference's centroidal-motion test n 9 m 16).  The lane-per-instance kernels (ddp_kernels.hpp) keep a whole instance in one lane's registers; from n ~ 9
// on that state (Vxx, Fx, Qxx ... ~ 6 n^2 doubles) spills, and a batch of 8192 instances occupies 128 of 1024 SIMDs.
// Here the 64 lanes of a wavefront work on ONE instance and the batch fills the chip:
//
//   linearisation   lane = TIMESTEP: the problem functor's calcStateEqDeriv / calcRunningCostDeriv are scalar code per
//                   (x_i, u_i); 64 timesteps of the same instance are evaluated at once, the derivative blocks go to the
//                   instance's HBM workspace (DDPSolver.hpp:157-185 materialises them the same way)
//   backward pass   lane = matrix entry: the n x n / m x n / m x m triple products of DDPSolver.hpp:386-441,522-527 run
//                   on the matrix cores (v_mfma_f64_16x16x4_f64 on 16 x 16 tiles staged in LDS with leading dimension
//                   17); the m x m factorisation and the vector recursions are lane-per-column
//   line search     lane = STEP SIZE: all alpha_list entries (DDPSolver.h:50-60, 11 by default) roll out at once, each
//                   lane its own candidate trajectory; the first accepted one in list order is taken, which is exactly
//                   what the sequential loop of DDPSolver.hpp:242-265 selects
//
// The control flow of solve() / procOnce (DDPSolver.hpp:26-340) is wave-uniform: there is no divergence between
// instances at all.  Arithmetic order: an MFMA accumulates k ascending with fused multiply-adds starting from the C
// operand (measured: bit-identical to the scalar fma chain, scripts/ubench_mfma_f64.hip), i.e. the same order as the
// lane-per-instance kernels; products are formed from zero and then added to the L-blocks, as the reference's
// temporaries are.
//
// Scope: unconstrained solves, 9 <= n <= 16, m <= 16, static or time-varying input dimension (inputDim(t) and m > 8 take
// the gains through LDS: cooperative L D L^T, per-lane column solves; centroidal motion n 9, m 16 / 0).  Box-constrained
// solves (BoxQP): static m <= 8 here (kConstrained), everything else on the lane-per-instance kernel.
#pragma once

#include <nmpc_amd/hip/ddp_kernels.hpp>

namespace nmpc_amd
{
namespace hip
{
typedef double v4d __attribute__((ext_vector_type(4)));

template<class Problem, bool kConstrained = false>
struct WaveSolver
{
  static constexpr int N = Problem::kStateDim;
  static constexpr int M = Problem::kInputDimMax;
  static constexpr int MM = (M > 0) ? M : 1;
  static_assert(N <= 16 && MM <= 16, "one 16 x 16 tile per block");
  /** Gains through LDS with run-time input dimension (cooperative L D L^T, per-lane column solves in place) instead of
      per-lane register copies of the M x M blocks: for inputDim(t) problems and for m > 8 (16 x 16 doubles do not fit a
      lane's registers).  Centroidal motion (n 9, m 16 / 0) takes this path. */
  static constexpr bool kLdsGains = Problem::kDynamicInput || MM > 8;
  static_assert(!(kConstrained && kLdsGains), "BoxQP is implemented on the register path only (static m <= 8)");
  using Lane = InstanceSolver<Problem, false>; // for the shared scalar helpers (ldltInPlace, ...)
  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using StateStateDimMatrix = typename Problem::StateStateDimMatrix;
  using InputInputDimMatrix = typename Problem::InputInputDimMatrix;
  using StateInputDimMatrix = typename Problem::StateInputDimMatrix;

  // ---- LDS: column-major tiles with leading dimension 17 (conflict-free row AND column access), 16 rows, as many
  // columns as the block has (n or m).  Nothing relies on padding: mma() masks the contraction index, storeAcc() masks
  // rows and columns, so reads beyond a tile's columns only ever feed results that are not stored.
  static constexpr int LD = 17;
  enum TileId
  {
    // m-column tiles first, then n-column tiles (reads beyond a tile stay inside the allocation)
    tFu = 0, // N x M
    tLxu, // N x M
    tLuu, // M x M
    tQuu, // M x M   unregularised
    tQuuF, // M x M   regularised
    tKtQuu, // N x M
    tVxx, // N x N
    tFx, // N x N
    tQxx, // N x N   Lxx on arrival, Qxx = Lxx + Fx^T Vxx Fx in place
    tP, // N x N   Fx^T Vxx; later Vxx_reg (reg_type 2) and the unsymmetrised new Vxx
    tP2, // M x N   Fu^T Vxx
    tQux, // M x N   unregularised
    tQuxR, // M x N   regularised (reg_type 2)
    tK, // M x N
    kNumTiles
  };
  static constexpr int tLxx = tQxx, tVnew = tP;
  static constexpr int kColsSmall = MM, kColsBig = N, kNumSmall = 6;
  NMPC_HD static constexpr int tileAt(int t)
  {
    return t < kNumSmall ? t * (LD * kColsSmall) : kNumSmall * (LD * kColsSmall) + (t - kNumSmall) * (LD * kColsBig);
  }
  // vectors (16 doubles each) behind the tiles
  enum VecId
  {
    vVx = 0,
    vLx,
    vLu,
    vU,
    vQx,
    vQu,
    vKff,
    vInvD, // reciprocals of D (LDS-gains path)
    kNumVecs
  };
  static constexpr int kVecAt = tileAt(kNumTiles) + 16; // + 16: slack for the reads beyond the last tile's columns
  static constexpr int kStageAt = kVecAt + kNumVecs * 16; //!< line search: two slots for the nominal record
  static constexpr int kLdsDoubles = kStageAt + 2 * 64 * ((2 * MM + MM * N + N + 63) / 64);
  static constexpr size_t kLdsBytes = static_cast<size_t>(kLdsDoubles) * sizeof(double);
  /** Resident wavefronts per SIMD the kernel is compiled for: two (256 VGPRs, <= 20 KB of LDS each) when the records
      fit, which hides part of the LDS / matrix-core latencies (quadrotor: +17 % measured); the manipulator's functor
      code spills at 256 VGPRs (-30 %), it keeps one wavefront per SIMD and 512 VGPRs. */
  static constexpr int kWavesPerSimd = (kLdsBytes <= 20 * 1024 && N * (N + MM) <= 200) ? 2 : 1;

  // ---- per-instance HBM workspace (doubles) ----
  // Derivative block of one timestep: every matrix is stored column-major with its rows padded to 16 and its size
  // padded to a multiple of 64 doubles (a "chunk"): lane l of a wavefront reads double 64 q + l of the block — one
  // 512-byte segment per chunk — and knows at compile time which LDS tile chunk q belongs to.
  NMPC_HD static constexpr int chunks(int cols)
  {
    return (16 * cols + 63) / 64;
  }
  static constexpr int cFx = 0; // chunk index where each matrix starts
  static constexpr int cFu = cFx + chunks(N);
  static constexpr int cLxx = cFu + chunks(MM);
  static constexpr int cLxu = cLxx + chunks(N);
  static constexpr int cLuu = cLxu + chunks(MM);
  static constexpr int cVec = cLuu + chunks(MM); // one chunk: Lx at [0, 16), Lu at [16, 32), u_i at [32, 48)
  static constexpr int kDerivChunks = cVec + 1;
  static constexpr int kDeriv = 64 * kDerivChunks;
  static constexpr int kGain = MM + MM * N; // k_i, K_i
  NMPC_HD static size_t trajDoubles(int T)
  {
    return static_cast<size_t>(T + 1) * N + static_cast<size_t>(T) * MM + static_cast<size_t>(T + 1);
  }
  NMPC_HD static size_t workspaceDoubles(int T)
  {
    return trajDoubles(T) * 2 + static_cast<size_t>(T) * (kDeriv + kGain); // control_data_ + ONE candidate
  }

  const Problem & problem;
  const nmpc_hip_ddp_config & cfg;
  const DeviceBuffers & buf;
  const int b;
  const int T;
  const int lane;
  double * lds;
  double * ws; //!< this instance's workspace
  int cur = 0; //!< which of the two trajectory buffers holds control_data_
  double current_t;

  NMPC_D WaveSolver(const Problem & p, const nmpc_hip_ddp_config & c, const DeviceBuffers & bf, int instance, double * lds_base)
  : problem(p), cfg(c), buf(bf), b(instance), T(bf.T), lane(static_cast<int>(threadIdx.x)), lds(lds_base)
  {
    ws = bf.wpi_ws + static_cast<size_t>(instance) * workspaceDoubles(bf.T);
  }

  // workspace views
  /** which = 0: control_data_, 1: candidate_control_data_.  Only ONE step size's rollout is ever stored: the one the line
      search is expected to take (the first), or the taken one re-rolled; accepting flips `cur` (DDPSolver.hpp:285-287 copies). */
  NMPC_D double * trajX(int which) const
  {
    return ws + static_cast<size_t>(which ^ cur) * trajDoubles(T);
  }
  NMPC_D double * trajU(int which) const
  {
    return trajX(which) + static_cast<size_t>(T + 1) * N;
  }
  NMPC_D double * trajC(int which) const
  {
    return trajU(which) + static_cast<size_t>(T) * MM;
  }
  NMPC_D double * derivBlock(int i) const
  {
    return ws + trajDoubles(T) * 2 + static_cast<size_t>(i) * kDeriv;
  }
  NMPC_D double * gainBlock(int i) const
  {
    return ws + trajDoubles(T) * 2 + static_cast<size_t>(T) * kDeriv + static_cast<size_t>(i) * kGain;
  }
  NMPC_D double * tile(int t) const
  {
    return lds + tileAt(t);
  }
  NMPC_D double * vec(int v) const
  {
    return lds + kVecAt + v * 16;
  }
  /** inputDim(t_i): wave-uniform */
  NMPC_D int inputDimAt(int i) const
  {
    if constexpr(Problem::kDynamicInput)
    {
      return problem.inputDim(current_t + i * problem.dt());
    }
    else
    {
      return M;
    }
  }
  /** Phase boundary: what this wave wrote to HBM (derivative blocks, gains, candidates) is read back by other lanes of
      the same wave in the next phase — wait for everything in flight (one wavefront per workgroup: no one to wait for). */
  NMPC_D static void sync()
  {
    syncThreadsFuzzed(__LINE__);
  }
  /** Inside a phase only LDS is exchanged between lanes.  The LDS executes one wave's instructions in order, so a
      compiler fence is all that is needed — and prefetches / stores to HBM stay in flight. */
  NMPC_D static void fence()
  {
    asm volatile("" ::: "memory");
  }

  // ===================================================================================================
  // tiles
  // ===================================================================================================
  /** One chunk of a derivative block (register value v = double 64 q + lane of the matrix) -> its LDS tile. */
  template<int ROWS, int COLS>
  NMPC_D void chunkToTile(int t, int q_in_matrix, double v) const
  {
    const int r = lane & 15, c = 4 * q_in_matrix + (lane >> 4); // entry 64 q + lane of the 16-row padded matrix
    const bool col_ok = (4 * q_in_matrix + 3 < COLS) ? true : (c < COLS);
    if(r < ROWS && col_ok) // the padding of the HBM block is never written
    {
      tile(t)[r + LD * c] = v;
    }
  }
  /** acc = op(A) * Bm over k < kdim, MFMA D layout: acc[r] = D(i = lane / 16 + 4 r, j = lane % 16).  Rows i beyond
      op(A)'s rows and columns j beyond Bm's columns come out as garbage: storeAcc() never stores them. */
  template<bool kTransA, int KDIM>
  NMPC_D v4d mma(int tA, int tB) const
  {
    const double * A = tile(tA);
    const double * Bm = tile(tB);
    const int lj = lane & 15, lk = lane >> 4;
    v4d acc = {0, 0, 0, 0};
#pragma unroll
    for(int k0 = 0; k0 < KDIM; k0 += 4)
    {
      const int k = k0 + lk;
      if constexpr(KDIM % 4 == 0)
      {
        const double a = kTransA ? A[k + LD * lj] : A[lj + LD * k];
        const double bb = Bm[k + LD * lj];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
      }
      else
      {
        // only the last group of four can run past KDIM: mask it (the tiles hold nothing meaningful there)
        const bool kv = (k0 + 4 <= KDIM) || k < KDIM;
        const int kc = kv ? k : 0;
        const double a = kTransA ? A[kc + LD * lj] : A[lj + LD * kc];
        const double bb = Bm[kc + LD * lj];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv ? a : 0.0, kv ? bb : 0.0, acc, 0, 0, 0);
      }
    }
    return acc;
  }
  /** Run-time contraction length (LDS-gains path). */
  template<bool kTransA>
  NMPC_D v4d mmaDyn(int tA, int tB, int kdim) const
  {
    const double * A = tile(tA);
    const double * Bm = tile(tB);
    const int lj = lane & 15, lk = lane >> 4;
    v4d acc = {0, 0, 0, 0};
    for(int k0 = 0; k0 < kdim; k0 += 4)
    {
      const int k = k0 + lk;
      const bool kv = k < kdim;
      const int kc = kv ? k : 0;
      const double a = kTransA ? A[kc + LD * lj] : A[lj + LD * kc];
      const double bb = Bm[kc + LD * lj];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv ? a : 0.0, kv ? bb : 0.0, acc, 0, 0, 0);
    }
    return acc;
  }
  NMPC_D void storeAccDyn(int t, v4d acc, int rows, int cols, int tAdd = -1) const
  {
    double * d = tile(t);
    const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int i = lk + 4 * r;
      if(i < rows && lj < cols)
      {
        const int at = i + LD * lj;
        d[at] = (tAdd >= 0) ? tile(tAdd)[at] + acc[r] : acc[r];
      }
    }
  }
  NMPC_D void chunkToTileDyn(int t, int q_in_matrix, int rows, int cols, double v) const
  {
    const int r = lane & 15, c = 4 * q_in_matrix + (lane >> 4);
    if(r < rows && c < cols)
    {
      tile(t)[r + LD * c] = v;
    }
  }

  /** ROWS x COLS of tile t <- acc (+ the same entry of tile tAdd when tAdd >= 0: "L + product"; tAdd may be t).
      The row test is decided at compile time wherever ROWS allows, the column test is one lane mask. */
  template<int ROWS, int COLS>
  NMPC_D void storeAcc(int t, v4d acc, int tAdd = -1) const
  {
    double * d = tile(t);
    const int lj = lane & 15, lk = lane >> 4;
    if(lj < COLS)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        const int i = lk + 4 * r;
        const bool row_ok = (4 * r + 3 < ROWS) ? true : ((4 * r < ROWS) ? (i < ROWS) : false);
        if(row_ok)
        {
          const int at = i + LD * lj;
          d[at] = (tAdd >= 0) ? tile(tAdd)[at] + acc[r] : acc[r];
        }
      }
    }
  }

  // ---- "natural" register layout of a 16 x 16 block: register r of lane (q = lane / 16, j = lane % 16) holds
  // X[4 r + q][j].  That is the matrix core's C / D layout; its four registers are at the same time the four k-slices of a
  // B operand, and passed as A operand they are the transposed matrix (scripts/ubench_mfma_f64.hip): mmaNat(X, Y) = X^T Y
  // without any LDS round trip between products.  Entries outside ROWS x COLS are zero.
  template<int ROWS, int COLS>
  NMPC_D v4d loadNat(int t) const
  {
    const double * d = tile(t);
    const int lj = lane & 15, lk = lane >> 4;
    v4d x;
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
      const int i = lk + 4 * r;
      const bool in = (i < ROWS) && (lj < COLS);
      const double v = d[in ? i + LD * lj : 0];
      x[r] = in ? v : 0.0;
    }
    return x;
  }
  /** X^T Y, formed from zero, over the first KDIM rows of X and Y (k ascending: the order of mma()). */
  template<int KDIM>
  NMPC_D static v4d mmaNat(v4d X, v4d Y)
  {
    v4d acc = {0, 0, 0, 0};
#pragma unroll
    for(int s = 0; s < (KDIM + 3) / 4; s++)
    {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[s], Y[s], acc, 0, 0, 0);
    }
    return acc;
  }

  // ===================================================================================================
  // model evaluation
  // ===================================================================================================
  NMPC_D void loadState(const double * g, StateDimVector & x) const
  {
#pragma unroll
    for(int j = 0; j < N; j++)
    {
      x[j] = g[j];
    }
  }
  NMPC_D void loadInput(const double * g, InputDimVector & u) const
  {
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      u[a] = g[a];
    }
  }

  /** Initial rollout (DDPSolver.hpp:83-95): every lane computes the same trajectory, lane 0 stores it. */
  NMPC_D double initialRollout() const
  {
    const size_t tl = static_cast<size_t>(b) / kLanesPerBlock, ln = static_cast<size_t>(b) % kLanesPerBlock;
    StateDimVector x;
    for(int j = 0; j < N; j++)
    {
      x[j] = buf.x0[(tl * N + j) * kLanesPerBlock + ln];
    }
    double * X = trajX(0);
    double * U = trajU(0);
    double * Cc = trajC(0);
    const double * Uin = buf.U + (tl * 2 + 0) * (static_cast<size_t>(T) * MM) * kLanesPerBlock + ln;
    double J = 0;
    for(int i = 0; i < T; i++)
    {
      const double t = current_t + i * problem.dt();
      const int m = inputDimAt(i);
      InputDimVector u;
      u.resize(m);
      for(int a = 0; a < MM; a++)
      {
        u[a] = (a < m) ? Uin[(static_cast<size_t>(i) * MM + a) * kLanesPerBlock] : 0.0; // zero beyond inputDim(t)
      }
      const double c = problem.runningCost(t, x, u);
      if(lane == 0)
      {
        for(int j = 0; j < N; j++)
        {
          X[i * N + j] = x[j];
        }
        for(int a = 0; a < MM; a++)
        {
          U[i * MM + a] = u[a];
        }
        Cc[i] = c;
      }
      J += c;
      x = problem.stateEq(t, x, u);
    }
    const double cT = problem.terminalCost(current_t + T * problem.dt(), x);
    if(lane == 0)
    {
      for(int j = 0; j < N; j++)
      {
        X[T * N + j] = x[j];
      }
      Cc[T] = cT;
    }
    return J + cT;
  }

  /** Step 1 of procOnce (DDPSolver.hpp:157-185): lane = timestep. */
  NMPC_D void linearise() const
  {
    const double * X = trajX(0);
    const double * U = trajU(0);
    for(int i0 = 0; i0 < T; i0 += 64)
    {
      const int i = i0 + lane;
      if(i < T)
      {
        const double t = current_t + i * problem.dt();
        StateDimVector x;
        InputDimVector u;
        loadState(X + static_cast<size_t>(i) * N, x);
        loadInput(U + static_cast<size_t>(i) * MM, u);
        StateStateDimMatrix Fx, Lxx;
        StateInputDimMatrix Fu, Lxu;
        StateDimVector Lx;
        InputDimVector Lu;
        InputInputDimMatrix Luu;
        if constexpr(Problem::kDynamicInput)
        {
          const int m = inputDimAt(i); // per lane: columns / entries beyond m are never read back
          u.resize(m);
          Fu.resize(N, m);
          Lxu.resize(N, m);
          Lu.resize(m);
          Luu.resize(m, m);
        }
        problem.calcStateEqDeriv(t, x, u, Fx, Fu);
        problem.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
        double * d = derivBlock(i);
#pragma unroll
        for(int c = 0; c < N; c++)
        {
#pragma unroll
          for(int r = 0; r < N; r++)
          {
            d[64 * cFx + r + 16 * c] = Fx(r, c);
            d[64 * cLxx + r + 16 * c] = Lxx(r, c);
          }
        }
#pragma unroll
        for(int c = 0; c < MM; c++)
        {
#pragma unroll
          for(int r = 0; r < N; r++)
          {
            d[64 * cFu + r + 16 * c] = Fu(r, c);
            d[64 * cLxu + r + 16 * c] = Lxu(r, c);
          }
#pragma unroll
          for(int r = 0; r < MM; r++)
          {
            d[64 * cLuu + r + 16 * c] = Luu(r, c);
          }
        }
#pragma unroll
        for(int e = 0; e < N; e++)
        {
          d[64 * cVec + e] = Lx[e];
        }
#pragma unroll
        for(int e = 0; e < MM; e++)
        {
          d[64 * cVec + 16 + e] = Lu[e];
          d[64 * cVec + 32 + e] = u[e];
        }
      }
    }
  }

  // ===================================================================================================
  // backward pass    DDPSolver.hpp:342-534
  // ===================================================================================================
  struct BackwardResult
  {
    bool ok;
    double dV0, dV1, k_rel_norm;
  };

  NMPC_D BackwardResult backwardPass(double lambda) const
  {
    BackwardResult res;
    res.ok = true;
    res.dV0 = 0;
    res.dV1 = 0;
    res.k_rel_norm = 0;
    // terminal value function    :346-365
    {
      StateDimVector xT, vx;
      StateStateDimMatrix vxx;
      loadState(trajX(0) + static_cast<size_t>(T) * N, xT);
      problem.calcTerminalCostDeriv(current_t + T * problem.dt(), xT, vx, vxx);
      sync();
      if(lane == 0)
      {
        for(int j = 0; j < N; j++)
        {
          vec(vVx)[j] = vx[j];
        }
        for(int c = 0; c < N; c++)
        {
          for(int r = 0; r < N; r++)
          {
            tile(tVxx)[r + LD * c] = vxx(r, c);
          }
        }
      }
      sync();
    }
    // the derivative block of timestep i - 1 is requested while timestep i is processed (registers: one double per
    // lane and chunk)
    double pf[kDerivChunks];
    {
      const double * d = derivBlock(T - 1);
#pragma unroll
      for(int q = 0; q < kDerivChunks; q++)
      {
        pf[q] = d[64 * q + lane];
      }
    }
    double k_next[MM]; // constrained solves: k of the previous (later) timestep, the BoxQP warm start
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      k_next[a] = 0;
    }
    for(int i = T - 1; i >= 0; i--)
    {
      // ---- stage this timestep's derivatives
#pragma unroll
      for(int q = 0; q < kDerivChunks; q++)
      {
        const double v = pf[q];
        if(q < cFu)
        {
          chunkToTile<N, N>(tFx, q - cFx, v);
        }
        else if(q < cLxx)
        {
          chunkToTile<N, MM>(tFu, q - cFu, v);
        }
        else if(q < cLxu)
        {
          chunkToTile<N, N>(tLxx, q - cLxx, v);
        }
        else if(q < cLuu)
        {
          chunkToTile<N, MM>(tLxu, q - cLxu, v);
        }
        else if(q < cVec)
        {
          chunkToTile<MM, MM>(tLuu, q - cLuu, v);
        }
        else
        {
          const int seg = lane >> 4, at = lane & 15; // vLx, vLu, vU are consecutive
          if(seg < 3 && at < (seg == 0 ? N : MM))
          {
            vec(vLx + seg)[at] = v;
          }
        }
      }
      {
        const double * d = derivBlock(i > 0 ? i - 1 : 0);
#pragma unroll
        for(int q = 0; q < kDerivChunks; q++)
        {
          pf[q] = d[64 * q + lane];
        }
      }
      fence();

      // ---- Q terms    :386-408    (products formed from zero in the reference's association, then added to the L block)
      // All five products stay in registers in the natural layout (see loadNat): P = Vxx^T Fx is (Fx^T Vxx)^T entry by
      // entry; R = Vxx^T [Fu | 0] with Vx placed in column M afterwards, so that S = Fx^T R carries (Fu^T Vxx Fx)^T in
      // columns < M and Fx^T Vx in column M, and W = R^T Fu carries (Fu^T Vxx) Fu in rows < M and Vx^T Fu in row M — the
      // same products in the same order as the lane kernels' chains, without staging Fx^T Vxx / Fu^T Vxx in LDS.
      {
        const int lj = lane & 15, lk = lane >> 4;
        const v4d Vxx_n = loadNat<N, N>(tVxx);
        const v4d Fx_n = loadNat<N, N>(tFx);
        const v4d Fu_n = loadNat<N, MM>(tFu);
        const v4d P = mmaNat<N>(Vxx_n, Fx_n);
        v4d R = mmaNat<N>(Vxx_n, Fu_n);
        if(lj == MM) // (M < 16 for every shape of this kernel's register path: column M is free)
        {
#pragma unroll
          for(int r = 0; r < 4; r++)
          {
            const int k = lk + 4 * r;
            R[r] = (k < N) ? vec(vVx)[k < N ? k : 0] : 0.0;
          }
        }
        const v4d qxx = mmaNat<N>(P, Fx_n);
        const v4d S = mmaNat<N>(Fx_n, R);
        const v4d W = mmaNat<N>(R, Fu_n);
        storeAcc<N, N>(tQxx, qxx, tLxx); // Lxx + (Fx^T Vxx) Fx, in place
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int c = lk + 4 * r; // S: row = state index c, column = input index (or M: the Qx column)
          if(c < N)
          {
            if(lj < MM)
            {
              tile(tQux)[lj + LD * c] = tile(tLxu)[c + LD * lj] + S[r];
            }
            else if(lj == MM)
            {
              vec(vQx)[c] = vec(vLx)[c] + S[r];
            }
          }
          const int a = lk + 4 * r; // W: row = input index a (or M: the Qu row), column = input index
          if(lj < MM)
          {
            if(a < MM)
            {
              tile(tQuu)[a + LD * lj] = tile(tLuu)[a + LD * lj] + W[r];
            }
            else if(a == MM)
            {
              vec(vQu)[lj] = vec(vLu)[lj] + W[r];
            }
          }
        }
      }
      fence();
      // ---- regularisation    :421-441
      if(cfg.reg_type == 2)
      {
        for(int e = lane; e < 16 * N; e += 64) // Vxx_reg = Vxx + lambda I in the (free) product tile
        {
          const int r = e & 15, c = e >> 4;
          if(r < N)
          {
            tile(tP)[r + LD * c] = (r == c) ? tile(tVxx)[r + LD * c] + lambda : tile(tVxx)[r + LD * c];
          }
        }
        fence();
        storeAcc<MM, N>(tP2, mma<true, N>(tFu, tP));
        fence();
        {
          const v4d acc = mma<false, N>(tP2, tFx);
          const v4d quf = mma<false, N>(tP2, tFu);
          const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
          for(int r = 0; r < 4; r++)
          {
            const int a = lk + 4 * r;
            if(a < MM && lj < N)
            {
              tile(tQuxR)[a + LD * lj] = tile(tLxu)[lj + LD * a] + acc[r];
            }
          }
          storeAcc<MM, MM>(tQuuF, quf, tLuu); // Luu + Fu^T Vxx_reg Fu
        }
      }
      else
      {
        for(int e = lane; e < 16 * N; e += 64)
        {
          const int r = e & 15, c = e >> 4;
          if(r < MM)
          {
            tile(tQuxR)[r + LD * c] = tile(tQux)[r + LD * c];
          }
        }
        for(int e = lane; e < 16 * MM; e += 64)
        {
          const int r = e & 15, c = e >> 4;
          if(r < MM)
          {
            const double q = tile(tQuu)[r + LD * c];
            tile(tQuuF)[r + LD * c] = (r == c && cfg.reg_type == 1) ? q + lambda : q;
          }
        }
      }
      fence(); // (Qx, Qu came out of the Q-term products above)

      // ---- gains    :500-517   (every lane factorises the same M x M matrix; lane c solves column c of Qux)
      double fac[MM * MM], inv_d[MM], Quu[MM * MM], Qu[MM], kff[MM];
#pragma unroll
      for(int e = 0; e < MM * MM; e++)
      {
        fac[e] = tile(tQuuF)[(e % MM) + LD * (e / MM)];
        Quu[e] = tile(tQuu)[(e % MM) + LD * (e / MM)];
      }
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        Qu[a] = vec(vQu)[a];
        kff[a] = 0;
      }
      double Kcol[MM], Quxcol[MM];
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        Kcol[a] = 0;
        Quxcol[a] = 0;
      }
      if constexpr(kConstrained)
      {
        // ---- box-constrained gains    :450-497: every lane solves the same small QP (BoxQP.h:141-347, the lane kernel's
        // implementation), lane c then solves column c of K on the free rows
        double initial_k[MM], lo[MM], up[MM];
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          initial_k[a] = (i != T - 1) ? k_next[a] : 0.0; // warm start from k_{i+1} (same dimension), :452-467
          lo[a] = inputLimitLo(buf, b, i, a) - vec(vU)[a]; // :470-472
          up[a] = inputLimitHi(buf, b, i, a) - vec(vU)[a];
        }
        const Lane lane_code(problem, cfg, buf, b);
        typename Lane::QPOutMasked qp;
        lane_code.boxQPMasked(fac, Qu, lo, up, initial_k, qp); // (static m = MM: no index lists, no private memory)
        if(lane == 0)
        {
          const size_t tl = static_cast<size_t>(b) / kLanesPerBlock, ln = static_cast<size_t>(b) % kLanesPerBlock;
          buf.qp_ret[(tl * T + i) * kLanesPerBlock + ln] = qp.retval;
          buf.qp_free[(tl * T + i) * kLanesPerBlock + ln] = qp.free;
        }
        if(qp.retval < 0)
        {
          res.ok = false; // :473-480
          return res;
        }
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          kff[a] = qp.x[a];
          k_next[a] = qp.x[a];
        }
        if(lane < N)
        {
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            Quxcol[a] = tile(tQux)[a + LD * lane];
            Kcol[a] = tile(tQuxR)[a + LD * lane];
          }
          Lane::maskedGainColumn(qp, Kcol);
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            tile(tK)[a + LD * lane] = Kcol[a];
          }
        }
      }
      else
      {
      if(!Lane::template ldltInPlace<MM>(fac, inv_d, M))
      {
        res.ok = false; // wave-uniform: every lane factorised the same matrix
        return res;
      }
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        kff[a] = Qu[a];
      }
      Lane::template ldltSolveInPlace<MM, 1>(fac, inv_d, M, kff);
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        kff[a] = -1 * kff[a];
      }
      if(lane < N)
      {
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          Kcol[a] = tile(tQuxR)[a + LD * lane];
          Quxcol[a] = tile(tQux)[a + LD * lane];
        }
        Lane::template ldltSolveInPlace<MM, 1>(fac, inv_d, M, Kcol);
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          Kcol[a] = -1 * Kcol[a];
          tile(tK)[a + LD * lane] = Kcol[a];
        }
      }
      }

      // ---- cost-to-go    :522-526
      {
        double kQu = 0, kQuuk = 0, Quu_k[MM];
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          kQu += kff[a] * Qu[a];
          double s = 0;
#pragma unroll
          for(int p = 0; p < MM; p++)
          {
            s += Quu[a + p * MM] * kff[p];
          }
          Quu_k[a] = s;
        }
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          kQuuk += kff[a] * Quu_k[a];
        }
        res.dV0 += kQu;
        res.dV1 += 0.5 * kQuuk;
      }
      if(lane < N)
      {
        // row `lane` of K^T Quu, then Vx[lane] = ((Qx + K^T Quu k) + K^T Qu) + Qux^T k
        double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          double s = 0;
#pragma unroll
          for(int p = 0; p < MM; p++)
          {
            s += Kcol[p] * Quu[p + a * MM];
          }
          tile(tKtQuu)[lane + LD * a] = s;
          s1 += s * kff[a];
          s2 += Kcol[a] * Qu[a];
          s3 += Quxcol[a] * kff[a];
        }
        vec(vVx)[lane] = ((vec(vQx)[lane] + s1) + s2) + s3;
      }
      fence();
      // Vxx = Qxx + K^T Quu K + K^T Qux + Qux^T K, each product a temporary formed from zero    :526
      {
        const v4d t1 = mma<false, M>(tKtQuu, tK);
        const v4d t2 = mma<true, M>(tK, tQux);
        const v4d t3 = mma<true, M>(tQux, tK);
        const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int at = (lk + 4 * r) + LD * lj;
          if(lk + 4 * r < N && lj < N)
          {
            tile(tVnew)[at] = ((tile(tQxx)[at] + t1[r]) + t2[r]) + t3[r];
          }
        }
      }
      fence();
      for(int e = lane; e < 16 * N; e += 64) // Vxx = 0.5 (Vxx + Vxx^T)    :527
      {
        const int r = e & 15, c = e >> 4;
        if(r < N)
        {
          tile(tVxx)[r + LD * c] = 0.5 * (tile(tVnew)[r + LD * c] + tile(tVnew)[c + LD * r]);
        }
      }
      // ---- save gains    :529-530, running max of |k_i| / (|u_i| + 1)    :217-221
      {
        double * g = gainBlock(i);
        if(lane < MM)
        {
          double mine = 0; // kff[lane] without indexing a register array by the lane id
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            mine = (lane == a) ? kff[a] : mine;
          }
          g[lane] = mine;
        }
        if(lane < N)
        {
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            g[MM + a + MM * lane] = Kcol[a];
          }
        }
        double kn = 0, un = 0;
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          kn += kff[a] * kff[a];
          const double ua = vec(vU)[a];
          un += ua * ua;
        }
        const double knorm = (M == 1) ? fabs(kff[0]) : sqrt(kn);
        const double unorm = (M == 1) ? fabs(vec(vU)[0]) : sqrt(un);
        res.k_rel_norm = fmax(res.k_rel_norm, knorm * recipFast(unorm + 1.0));
      }
      fence();
    }
    return res;
  }

  // ===================================================================================================
  // backward pass, gains through LDS (run-time input dimension m, any m <= 16)
  // ===================================================================================================
  /** In-place solve (L D L^T) x = rhs on a column that lives in LDS (x = col[0 .. m)); L, D^-1 from tQuuF / vInvD.
      Same operation order as InstanceSolver::ldltSolveInPlace. */
  NMPC_D void solveColumnLds(double * col, int m) const
  {
    // The column goes to registers and the loops are fully unrolled with masks: rolled loops would make every
    // multiply-subtract wait for two dependent LDS reads (~130 cycles each time, m^2 times).  The reads of L are
    // wave-uniform broadcasts and independent of the arithmetic, so they stream.
    const double * L = tile(tQuuF);
    const double * inv_d = vec(vInvD);
    double x[MM];
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      x[i] = (i < m) ? col[i] : 0.0;
    }
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      double s = x[i];
#pragma unroll
      for(int j = 0; j < MM; j++)
      {
        if(j < i)
        {
          const double l = (i < m) ? L[i + LD * j] : 0.0;
          s -= l * x[j];
        }
      }
      x[i] = s;
    }
#pragma unroll
    for(int ii = 0; ii < MM; ii++)
    {
      const int i = MM - 1 - ii;
      double s = x[i] * ((i < m) ? inv_d[i] : 0.0);
#pragma unroll
      for(int j = 0; j < MM; j++)
      {
        if(j > i)
        {
          const double l = (j < m) ? L[j + LD * i] : 0.0;
          s -= l * x[j];
        }
      }
      x[i] = s;
    }
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      if(i < m)
      {
        col[i] = x[i];
      }
    }
  }

  NMPC_D BackwardResult backwardPassLds(double lambda) const
  {
    BackwardResult res;
    res.ok = true;
    res.dV0 = 0;
    res.dV1 = 0;
    res.k_rel_norm = 0;
    {
      StateDimVector xT, vx;
      StateStateDimMatrix vxx;
      loadState(trajX(0) + static_cast<size_t>(T) * N, xT);
      problem.calcTerminalCostDeriv(current_t + T * problem.dt(), xT, vx, vxx);
      sync();
      if(lane == 0)
      {
        for(int j = 0; j < N; j++)
        {
          vec(vVx)[j] = vx[j];
        }
        for(int c = 0; c < N; c++)
        {
          for(int r = 0; r < N; r++)
          {
            tile(tVxx)[r + LD * c] = vxx(r, c);
          }
        }
      }
      sync();
    }
    double pf[kDerivChunks];
    {
      const double * d = derivBlock(T - 1);
#pragma unroll
      for(int q = 0; q < kDerivChunks; q++)
      {
        pf[q] = d[64 * q + lane];
      }
    }
    for(int i = T - 1; i >= 0; i--)
    {
      const int m = inputDimAt(i);
      // ---- stage this timestep's derivatives
#pragma unroll
      for(int q = 0; q < kDerivChunks; q++)
      {
        const double v = pf[q];
        if(q < cFu)
        {
          chunkToTileDyn(tFx, q - cFx, N, N, v);
        }
        else if(q < cLxx)
        {
          chunkToTileDyn(tFu, q - cFu, N, m, v);
        }
        else if(q < cLxu)
        {
          chunkToTileDyn(tLxx, q - cLxx, N, N, v);
        }
        else if(q < cLuu)
        {
          chunkToTileDyn(tLxu, q - cLxu, N, m, v);
        }
        else if(q < cVec)
        {
          chunkToTileDyn(tLuu, q - cLuu, m, m, v);
        }
        else
        {
          const int seg = lane >> 4, at = lane & 15;
          if(seg < 3 && at < (seg == 0 ? N : m))
          {
            vec(vLx + seg)[at] = v;
          }
        }
      }
      {
        const double * d = derivBlock(i > 0 ? i - 1 : 0);
#pragma unroll
        for(int q = 0; q < kDerivChunks; q++)
        {
          pf[q] = d[64 * q + lane];
        }
      }
      fence();

      // ---- Q terms    :386-408
      {
        const v4d p1 = mma<true, N>(tFx, tVxx);
        const v4d p2 = mma<true, N>(tFu, tVxx);
        storeAcc<N, N>(tP, p1);
        storeAccDyn(tP2, p2, m, N);
      }
      fence();
      {
        const v4d qxx = mma<false, N>(tP, tFx);
        const v4d qux = mma<false, N>(tP2, tFx);
        const v4d quu = mma<false, N>(tP2, tFu);
        storeAcc<N, N>(tQxx, qxx, tLxx);
        const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int a = lk + 4 * r;
          if(a < m && lj < N)
          {
            tile(tQux)[a + LD * lj] = tile(tLxu)[lj + LD * a] + qux[r];
          }
        }
        storeAccDyn(tQuu, quu, m, m, tLuu);
      }
      fence();
      // ---- regularisation    :421-441
      if(cfg.reg_type == 2)
      {
        for(int e = lane; e < 16 * N; e += 64)
        {
          const int r = e & 15, c = e >> 4;
          if(r < N)
          {
            tile(tP)[r + LD * c] = (r == c) ? tile(tVxx)[r + LD * c] + lambda : tile(tVxx)[r + LD * c];
          }
        }
        fence();
        storeAccDyn(tP2, mma<true, N>(tFu, tP), m, N);
        fence();
        {
          const v4d acc = mma<false, N>(tP2, tFx);
          const v4d quf = mma<false, N>(tP2, tFu);
          const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
          for(int r = 0; r < 4; r++)
          {
            const int a = lk + 4 * r;
            if(a < m && lj < N)
            {
              tile(tQuxR)[a + LD * lj] = tile(tLxu)[lj + LD * a] + acc[r];
            }
          }
          storeAccDyn(tQuuF, quf, m, m, tLuu);
        }
      }
      else
      {
        for(int e = lane; e < 16 * N; e += 64)
        {
          const int r = e & 15, c = e >> 4;
          if(r < m)
          {
            tile(tQuxR)[r + LD * c] = tile(tQux)[r + LD * c];
          }
        }
        for(int e = lane; e < 16 * MM; e += 64)
        {
          const int r = e & 15, c = e >> 4;
          if(r < m && c < m)
          {
            const double q = tile(tQuu)[r + LD * c];
            tile(tQuuF)[r + LD * c] = (r == c && cfg.reg_type == 1) ? q + lambda : q;
          }
        }
      }
      // ---- Qx, Qu    :386-388
      if(lane < N)
      {
        double sacc = 0;
        for(int k = 0; k < N; k++)
        {
          sacc += tile(tFx)[k + LD * lane] * vec(vVx)[k];
        }
        vec(vQx)[lane] = vec(vLx)[lane] + sacc;
      }
      if(lane < m)
      {
        double sacc = 0;
        for(int k = 0; k < N; k++)
        {
          sacc += tile(tFu)[k + LD * lane] * vec(vVx)[k];
        }
        vec(vQu)[lane] = vec(vLu)[lane] + sacc;
      }
      fence();

      // ---- L D L^T of Quu_F in place in LDS    :500-508.  Right-looking (after pivot j the trailing block receives its
      // rank-one update, 4 entries per lane) — but every entry still receives exactly the subtractions
      // (L_ij L_kj) d_j in ascending j that ldltInPlace's left-looking loops apply, so the factor is the same bits.
      {
        double * A = tile(tQuuF);
        for(int j = 0; j < m; j++)
        {
          const double d = A[j + LD * j];
          if(d <= 0) // every lane reads the same pivot
          {
            res.ok = false;
            return res;
          }
          const double r = recipFast(d);
          fence();
          if(lane > j && lane < m)
          {
            A[lane + LD * j] = A[lane + LD * j] * r;
          }
          if(lane == 0)
          {
            vec(vInvD)[j] = r;
          }
          fence();
#pragma unroll
          for(int q = 0; q < 4; q++)
          {
            const int e = lane + 64 * q, i = e & 15, k = e >> 4;
            if(k > j && i >= k && i < m)
            {
              A[i + LD * k] -= (A[i + LD * j] * A[k + LD * j]) * d;
            }
          }
          fence();
        }
      }
      // ---- gains    :509-517: lane c < N solves column c of Qux_reg in place (tile tK), lane N solves Qu -> k
      if(lane < N)
      {
        for(int a = 0; a < m; a++)
        {
          tile(tK)[a + LD * lane] = tile(tQuxR)[a + LD * lane];
        }
        solveColumnLds(tile(tK) + LD * lane, m);
        for(int a = 0; a < m; a++)
        {
          tile(tK)[a + LD * lane] = -1 * tile(tK)[a + LD * lane];
        }
      }
      else if(lane == N)
      {
        for(int a = 0; a < m; a++)
        {
          vec(vKff)[a] = vec(vQu)[a];
        }
        solveColumnLds(vec(vKff), m);
        for(int a = 0; a < m; a++)
        {
          vec(vKff)[a] = -1 * vec(vKff)[a];
        }
      }
      fence();
      // ---- cost-to-go    :522-526
      {
        double kQu = 0, kQuuk = 0, kf[MM];
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          kf[a] = (a < m) ? vec(vKff)[a] : 0.0;
        }
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            kQu += kf[a] * vec(vQu)[a];
          }
        }
        // Quu k first (as the lane kernels do), then k^T (Quu k)
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            double sacc = 0;
#pragma unroll
            for(int q = 0; q < MM; q++)
            {
              if(q < m)
              {
                sacc += tile(tQuu)[a + LD * q] * kf[q];
              }
            }
            kQuuk += kf[a] * sacc;
          }
        }
        res.dV0 += kQu;
        res.dV1 += 0.5 * kQuuk;
      }
      // K^T Quu on the matrix cores (N x m, contraction over m: the same ascending chain the per-lane loop would run),
      // then row r of it, of K^T and of Qux^T against k / Qu in lane r
      storeAccDyn(tKtQuu, mmaDyn<true>(tK, tQuu, m), N, m);
      fence();
      if(lane < N)
      {
        double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          if(a < m)
          {
            s1 += tile(tKtQuu)[lane + LD * a] * vec(vKff)[a];
            s2 += tile(tK)[a + LD * lane] * vec(vQu)[a];
            s3 += tile(tQux)[a + LD * lane] * vec(vKff)[a];
          }
        }
        vec(vVx)[lane] = ((vec(vQx)[lane] + s1) + s2) + s3;
      }
      fence();
      {
        const v4d t1 = mmaDyn<false>(tKtQuu, tK, m);
        const v4d t2 = mmaDyn<true>(tK, tQux, m);
        const v4d t3 = mmaDyn<true>(tQux, tK, m);
        const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
        for(int r = 0; r < 4; r++)
        {
          const int at = (lk + 4 * r) + LD * lj;
          if(lk + 4 * r < N && lj < N)
          {
            tile(tVnew)[at] = ((tile(tQxx)[at] + t1[r]) + t2[r]) + t3[r];
          }
        }
      }
      fence();
      for(int e = lane; e < 16 * N; e += 64)
      {
        const int r = e & 15, c = e >> 4;
        if(r < N)
        {
          tile(tVxx)[r + LD * c] = 0.5 * (tile(tVnew)[r + LD * c] + tile(tVnew)[c + LD * r]);
        }
      }
      // ---- save gains (zero beyond the input dimension), |k_i| / (|u_i| + 1)
      {
        double * g = gainBlock(i);
        if(lane < MM)
        {
          g[lane] = (lane < m) ? vec(vKff)[lane] : 0.0;
        }
        if(lane < N)
        {
          for(int a = 0; a < MM; a++)
          {
            g[MM + a + MM * lane] = (a < m) ? tile(tK)[a + LD * lane] : 0.0;
          }
        }
        double kn = 0, un = 0;
        for(int a = 0; a < m; a++)
        {
          const double ka = vec(vKff)[a], ua = vec(vU)[a];
          kn += ka * ka;
          un += ua * ua;
        }
        const double knorm = (M == 1) ? fabs(m > 0 ? vec(vKff)[0] : 0.0) : sqrt(kn);
        const double unorm = (M == 1) ? fabs(m > 0 ? vec(vU)[0] : 0.0) : sqrt(un);
        res.k_rel_norm = fmax(res.k_rel_norm, knorm * recipFast(unorm + 1.0));
      }
      fence();
    }
    return res;
  }

  // ===================================================================================================
  // line search    DDPSolver.hpp:234-274, forwardPass :536-560 : lane j rolls out alpha_list[j]
  // ===================================================================================================
  static constexpr int kNom = kGain + N + MM; // per timestep: k_i, K_i, x_i, u_i (the same for every lane)
  static constexpr int kNomChunks = (kNom + 63) / 64;
  /** double e of the nominal record of timestep i */
  NMPC_D double nominalWord(int i, int e) const
  {
    if(e < kGain)
    {
      return gainBlock(i)[e];
    }
    if(e < kGain + N)
    {
      return trajX(0)[static_cast<size_t>(i) * N + (e - kGain)];
    }
    return trajU(0)[static_cast<size_t>(i) * MM + (e - kGain - N)];
  }

  /** \param store_lane the lane (= step-size index) whose rollout goes to the candidate buffer; the others only sum their
      cost.  (Every candidate used to be written and the accepted one copied: 11 x 6.9 KB of stores per line search.) */
  NMPC_D double forwardCandidate(double alpha, bool active_lane, int store_lane) const
  {
    const bool active = active_lane && lane == store_lane;
    double * Xc = trajX(1);
    double * Uc = trajU(1);
    double * Cc = trajC(1);
    // The nominal record is the same for all lanes: the wave fetches it once (one or two doubles per lane, requested one
    // timestep ahead), passes it through LDS, and every lane reads it from there.
    double nf[kNomChunks];
#pragma unroll
    for(int q = 0; q < kNomChunks; q++)
    {
      const int e = 64 * q + lane;
      nf[q] = (e < kNom) ? nominalWord(0, e) : 0.0;
    }
    StateDimVector xc;
    loadState(trajX(0), xc); // x'_0 = x_0
    double J = 0;
    for(int i = 0; i < T; i++)
    {
      const double t = current_t + i * problem.dt();
      double * st = lds + kStageAt + (i & 1) * (64 * kNomChunks);
#pragma unroll
      for(int q = 0; q < kNomChunks; q++)
      {
        st[64 * q + lane] = nf[q];
      }
      if(i + 1 < T)
      {
#pragma unroll
        for(int q = 0; q < kNomChunks; q++)
        {
          const int e = 64 * q + lane;
          nf[q] = (e < kNom) ? nominalWord(i + 1, e) : 0.0;
        }
      }
      fence();
      const double * g = st;
      const double * xn = st + kGain;
      const double * un = st + kGain + N;
      const int m = inputDimAt(i);
      InputDimVector uc;
      uc.resize(m);
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        double s = 0;
#pragma unroll
        for(int c = 0; c < N; c++)
        {
          s += g[MM + a + MM * c] * (xc[c] - xn[c]);
        }
        uc[a] = (a < m) ? (un[a] + alpha * g[a]) + s : 0.0;
      }
      const double c = problem.runningCost(t, xc, uc);
      if(active)
      {
#pragma unroll
        for(int j = 0; j < N; j++)
        {
          Xc[static_cast<size_t>(i) * N + j] = xc[j];
        }
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          Uc[static_cast<size_t>(i) * MM + a] = uc[a];
        }
        Cc[i] = c;
      }
      J += c;
      xc = problem.stateEq(t, xc, uc);
      fence();
    }
    const double cT = problem.terminalCost(current_t + T * problem.dt(), xc);
    if(active)
    {
#pragma unroll
      for(int j = 0; j < N; j++)
      {
        Xc[static_cast<size_t>(T) * N + j] = xc[j];
      }
      Cc[T] = cT;
    }
    return J + cT;
  }

  // ===================================================================================================
  // solve    DDPSolver.hpp:26-141, procOnce :143-340
  // ===================================================================================================
  NMPC_D void writeTraceRow(int row, const double * tr) const
  {
    if(cfg.trace_level >= 1 && row < buf.trace_rows && lane < NMPC_HIP_NTRACE)
    {
      const size_t tl = static_cast<size_t>(b) / kLanesPerBlock, ln = static_cast<size_t>(b) % kLanesPerBlock;
      double * p = buf.trace + (tl * (static_cast<size_t>(buf.trace_rows) * NMPC_HIP_NTRACE)) * kLanesPerBlock + ln;
      p[(static_cast<size_t>(row) * NMPC_HIP_NTRACE + lane) * kLanesPerBlock] = tr[lane];
    }
  }

#ifdef NMPC_AMD_PROFILE_WPI
  // profiling build (scripts/profile_wpi.py): shader cycles per phase of instance 0, returned through qp_free
  mutable unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
  mutable unsigned long long prof_t0 = 0;
  NMPC_D void profBegin() const
  {
    prof_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int k) const
  {
    prof_acc[k] += __builtin_readcyclecounter() - prof_t0;
  }
  NMPC_D void profFlush() const
  {
    if(b == 0 && lane == 0)
    {
      for(int k = 0; k < 6; k++)
      {
        buf.qp_free[static_cast<size_t>(k) * kLanesPerBlock] = static_cast<unsigned>(prof_acc[k] >> 4);
      }
    }
  }
#else
  // product builds: ticks per phase (0 initial rollout, 1 linearise, 2 backward, 3 line search, 4 adopt, 5 write-out), folded
  // into DeviceBuffers::phase_ticks at the end of the solve
  mutable unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
  mutable unsigned long long prof_t0 = 0;
  NMPC_D void profBegin() const
  {
    prof_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int k) const
  {
    prof_acc[k] += __builtin_readcyclecounter() - prof_t0;
  }
  NMPC_D void profFlush() const
  {
    if(lane == 0 && buf.phase_ticks != nullptr)
    {
      unsigned long long * p = buf.phase_ticks + static_cast<size_t>(b) * 4;
      p[0] = prof_acc[1] + prof_acc[2];
      p[1] = prof_acc[0] + prof_acc[3] + prof_acc[4];
      p[2] = prof_acc[0] + prof_acc[1] + prof_acc[2] + prof_acc[3] + prof_acc[4] + prof_acc[5];
    }
  }
#endif

  NMPC_D BackwardResult runBackward(double lambda) const
  {
    if constexpr(kLdsGains)
    {
      return backwardPassLds(lambda);
    }
    else
    {
      return backwardPass(lambda);
    }
  }

  NMPC_D void solve()
  {
    current_t = buf.t0 ? buf.t0[b] : 0.0;
    double lambda = cfg.initial_lambda, dlambda = cfg.initial_dlambda;
    profBegin();
    double J_cur = initialRollout();
    sync();
    profEnd(0);

    __shared__ double tr_sh[NMPC_HIP_NTRACE]; // trace row, indexed by lane when written out
    double tr[NMPC_HIP_NTRACE];
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      tr[f] = 0;
    }
    tr[NMPC_HIP_TRACE_COST] = J_cur;
    tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
    tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
    tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
    auto flushTrace = [&](int row)
    {
      sync();
      if(lane == 0)
      {
        for(int f = 0; f < NMPC_HIP_NTRACE; f++)
        {
          tr_sh[f] = tr[f];
        }
      }
      sync();
      writeTraceRow(row, tr_sh);
    };
    flushTrace(0);

    int retval = 0;
    double dV0 = 0, dV1 = 0;
    for(int iter = 1; iter <= cfg.max_iter; iter++)
    {
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        tr[f] = 0;
      }
      tr[NMPC_HIP_TRACE_ITER] = iter;
      tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
      retval = 0;

      profBegin();
      linearise(); // Step 1
      sync();
      profEnd(1);
      // Step 2: backward pass with regularisation retries    :188-214
      int n_backward = 1;
      bool bw_failed = false;
      profBegin();
      BackwardResult bw = runBackward(lambda);
      sync(); // the gains go to HBM and come back to other lanes in the line search
      profEnd(2);
      while(!bw.ok)
      {
        dlambda = fmax(dlambda * cfg.lambda_factor, cfg.lambda_factor);
        lambda = fmax(lambda * dlambda, cfg.lambda_min);
        if(lambda > cfg.lambda_max)
        {
          bw_failed = true;
          break;
        }
        n_backward++;
        bw = runBackward(lambda);
        sync();
      }
      tr[NMPC_HIP_TRACE_N_BACKWARD] = n_backward;
      if(bw_failed)
      {
        retval = -1;
      }
      else
      {
        dV0 = bw.dV0;
        dV1 = bw.dV1;
        tr[NMPC_HIP_TRACE_K_REL_NORM] = bw.k_rel_norm;
        if(bw.k_rel_norm < cfg.k_rel_norm_thre && lambda < cfg.lambda_thre)
        {
          retval = 1;
        }
        else
        {
          // Step 3: every step size at once; the sequential loop's choice is the first accepted one in list order
          const bool active = lane < cfg.n_alpha;
          const double alpha_l = cfg.alpha_list[active ? lane : 0];
          profBegin();
          const double J_cand = forwardCandidate(alpha_l, active, 0);
          profEnd(3);
          const double actual_l = J_cur - J_cand;
          const double expected_l = -1 * alpha_l * (dV0 + alpha_l * dV1);
          double ratio_l = actual_l / expected_l;
          if(expected_l < 0)
          {
            ratio_l = (actual_l >= 0 ? 1 : -1); // :251-259
          }
          const unsigned long long accept = __ballot(active && ratio_l > cfg.cost_update_ratio_thre);
          const bool success = accept != 0;
          const int ai = success ? __builtin_ctzll(accept) : cfg.n_alpha - 1;
          const double alpha = __shfl(alpha_l, ai);
          const double cost_update_actual = __shfl(actual_l, ai);
          const double cost_update_expected = __shfl(expected_l, ai);
          const double cost_update_ratio = __shfl(ratio_l, ai);
          tr[NMPC_HIP_TRACE_ALPHA] = alpha;
          tr[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL] = cost_update_actual;
          tr[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED] = cost_update_expected;
          tr[NMPC_HIP_TRACE_COST_UPDATE_RATIO] = cost_update_ratio;
          tr[NMPC_HIP_TRACE_ALPHA_IDX] = ai;
          tr[NMPC_HIP_TRACE_N_FORWARD] = success ? ai + 1 : cfg.n_alpha;
          // Step 4    :280-333
          if(success)
          {
            profBegin();
            sync();
            if(ai != 0)
            {
              // a later step size was taken: roll it out once more, this time into the candidate buffer (same instruction
              // stream on the same inputs: the same trajectory its lane summed the cost of)
              (void)forwardCandidate(alpha_l, active, ai);
              sync();
            }
            cur ^= 1; // the candidate buffer becomes control_data_
            profEnd(4);
            J_cur = __shfl(J_cand, ai);
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
      }
      flushTrace(iter);
      if(retval != 0)
      {
        break;
      }
    }

    // ---- results -> the handle's tile-major arrays (half 0), as the lane-per-instance kernels leave them
    profBegin();
    sync();
    const size_t tl = static_cast<size_t>(b) / kLanesPerBlock, ln = static_cast<size_t>(b) % kLanesPerBlock;
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
    {
      double * Xo = buf.X + ((tl * 2 + 0) * rows_x) * kLanesPerBlock + ln;
      const double * X = trajX(0);
      for(size_t e = lane; e < rows_x; e += 64)
      {
        Xo[e * kLanesPerBlock] = X[e];
      }
      double * Uo = buf.U + ((tl * 2 + 0) * rows_u) * kLanesPerBlock + ln;
      const double * U = trajU(0);
      for(size_t e = lane; e < rows_u; e += 64)
      {
        Uo[e * kLanesPerBlock] = U[e];
      }
      double * Co = buf.cost + ((tl * 2 + 0) * static_cast<size_t>(T + 1)) * kLanesPerBlock + ln;
      const double * Cc = trajC(0);
      for(size_t e = lane; e < static_cast<size_t>(T + 1); e += 64)
      {
        Co[e * kLanesPerBlock] = Cc[e];
      }
      double * ko = buf.kff + (tl * rows_u) * kLanesPerBlock + ln;
      double * Ko = buf.Kfb + (tl * rows_u * N) * kLanesPerBlock + ln;
      for(size_t e = lane; e < static_cast<size_t>(T) * kGain; e += 64)
      {
        const size_t i = e / kGain, w = e % kGain;
        const double v = gainBlock(static_cast<int>(i))[w];
        if(w < static_cast<size_t>(MM))
        {
          ko[(i * MM + w) * kLanesPerBlock] = v;
        }
        else
        {
          Ko[(i * (N * MM) + (w - MM)) * kLanesPerBlock] = v;
        }
      }
      for(size_t e = lane; e < static_cast<size_t>(T); e += 64)
      {
        buf.input_dim[(tl * T + e) * kLanesPerBlock + ln] = inputDimAt(static_cast<int>(e));
      }
    }
    if(lane < NMPC_HIP_NTRACE)
    {
      buf.trace_last[(tl * NMPC_HIP_NTRACE + lane) * kLanesPerBlock + ln] = tr_sh[lane];
    }
    if(lane == 0)
    {
      buf.status[b] = retval;
      buf.iters[b] = static_cast<int>(tr[NMPC_HIP_TRACE_ITER]);
      buf.sel[b] = 0;
      buf.dV[(tl * 2 + 0) * kLanesPerBlock + ln] = dV0;
      buf.dV[(tl * 2 + 1) * kLanesPerBlock + ln] = dV1;
    }
    profEnd(5);
    profFlush();
  }
};

/** The wave-per-instance solve kernel: grid = B workgroups of one wavefront. */
template<class Problem, bool kConstrained, bool kOwnProblem = false>
__global__ __launch_bounds__(kLanesPerBlock)
    __attribute__((amdgpu_waves_per_eu(WaveSolver<Problem, kConstrained>::kWavesPerSimd,
                                       WaveSolver<Problem, kConstrained>::kWavesPerSimd))) void ddp_solve_wpi_kernel(const Problem problem,
                                                                        const nmpc_hip_ddp_config cfg,
                                                                        const DeviceBuffers buf)
{
  extern __shared__ __attribute__((aligned(16))) double lds_wpi[];
  const Problem mine = kOwnProblem ? instanceProblem(problem, buf, static_cast<int>(blockIdx.x)) : problem;
  WaveSolver<Problem, kConstrained> solver(mine, cfg, buf, static_cast<int>(blockIdx.x), lds_wpi);
  solver.solve();
}
} // namespace hip
} // namespace nmpc_amd
