// gfx950 device code of the batched DDP solver, fp32, LANE MAPPING "TILE32": one workgroup of eight wavefronts solves 32
// problem instances whose (n + m) x (n + m) augmented blocks are ONE 16 x 16 matrix-core tile — BASELINE.json config 4
// (quadrotor n 12, m 4, T 50, batch 8192, fp32: 8192 / 32 = 256 workgroups = one per CU).
//
// What a lane means changes with the phase (reference: nmpc_ddp/include/nmpc_ddp/DDPSolver.hpp):
//
//   model code  — rollouts (:83-95, :536-560) and the linearisation sweep (:157-185): lane = INSTANCE (wave 0, "model
//                 wave"), the line search (:234-274) additionally lane = (instance, step size) on waves 1..7, all step sizes
//                 of alpha_list at once (the trials are independent: same nominal, same gains).
//   backward    — (:342-534) lane = MATRIX ENTRY on waves 1..7, five instances per wave (two on the model wave's SIMD
//                 partner), on v_mfma_f32_16x16x4_f32.
//
// Derivatives never reach HBM and are never materialised for the whole horizon either: the model wave linearises timestep
// i - 1 of all 32 instances into an LDS record while the seven matrix waves consume the record of timestep i (two record
// slots, one barrier per timestep).  HBM sees (x, u) once per sweep, the gains once, and the rollouts' trajectories: the
// fused lower bound of SURVEY.md §8(d).
//
// The backward step in "natural layout".  A 16 x 16 fp32 matrix X lives in four registers: register r of lane
// (q = lane / 16, j = lane % 16) holds X[4 q + r][j] — the matrix core's C / D layout (measured: scripts/ubench_mfma_f32.hip,
// profiles/r02_ubench_mfma_f32.txt).  The same four registers passed as the A operands of four MFMAs, with another
// matrix's registers as B operands, contract over the row index (in the order 0,4,8,12,1,5,...): mma(X, Y, C) = X^T Y + C,
// with no LDS round trip or cross-lane move between chained products.  With w = [dx; du] (n + m = 16):
//     VV = [Vxx | Vx]  (n x (n+1)),   F = [Fx Fu]  (n x 16),   L = [[Lxx Lxu],[Lxu^T Luu]],   l = [Lx; Lu]
//     G  = VV^T F            rows < n: Vxx F, row n: Vx^T F                                  (:386-408, all five Q blocks
//     Q  = F^T G + L         = [[Qxx Qxu],[Qux Quu]];  q = l + G[n,:]^T = [Qx; Qu]            in two products)
//     Quu_F = Quu + lambda I, LDL^T in every lane; lane (n/4, j) solves column j of [Qux | Qu]:  A = [[I 0],[K k]]  (:500-517)
//     H  = Q A + [0 | q]     column n: Q [0; k] + q
//     VV'= A^T H             = [Qxx + Qxu K + K^T Qux + K^T Quu K | Qx + Qxu k + K^T Qu + K^T Quu k]          (:522-526)
//     and H^T A for the transpose, so that Vxx <- (Vxx + Vxx^T) / 2 (:527) needs no shuffle.
// 20 MFMAs and ~100 other instructions per instance and timestep.  The association of the triple products differs from the
// reference's left-to-right order ((Fx^T Vxx) Fx there, Fx^T (Vxx Fx) here) and sums run in the matrix core's order:
// rounding-level differences, inside the fp32 tolerance of SURVEY.md §8(c).
//
// Scope of this kernel family: Scalar = float, static input dimension, n in {4, 8, 12}, 1 <= m <= 4, n + m <= 16,
// unconstrained solves (with_input_constraint is rejected at launch), one shared problem object per batch.
#pragma once

#include <cstring>
#include <new>
#include <type_traits>

#include <nmpc_amd/hip/model_ops.hpp>

namespace nmpc_amd
{
namespace hip
{
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTileInstances = 32; //!< instances per workgroup
constexpr int kTileWaves = 8; //!< two per SIMD: 256 registers each
constexpr int kTileThreads = kTileWaves * 64;
constexpr int kTileModelWave = 0; //!< model code, lane = instance
constexpr int kTileMatrixWaves = kTileWaves - 1; //!< waves 1..7: backward pass, line-search fan-out
constexpr int kTileMaxPerWave = 5; //!< instances of one matrix wave
/** Instances of matrix wave w (1..7).  Waves w and w + 4 share a SIMD: wave 4, the model wave's partner, takes two
    instances (the linearisation of 32 instances costs about as many instructions per timestep as three to four backward
    steps), the other six take five. */
__host__ __device__ constexpr int tileWaveCount(int w)
{
  return w == 4 ? 2 : 5;
}
__host__ __device__ constexpr int tileWaveFirst(int w)
{
  return w <= 4 ? 5 * (w - 1) : 17 + 5 * (w - 5);
}
static_assert(tileWaveFirst(7) + tileWaveCount(7) == kTileInstances && tileWaveFirst(4) == 15 && tileWaveFirst(5) == 17,
              "the matrix waves cover the 32 slots");

template<class Problem>
struct TileSolver32
{
  using S = float;
  static_assert(std::is_same<typename Problem::Scalar, float>::value, "the tile kernel computes in fp32");
  static constexpr int N = Problem::kStateDim;
  static constexpr int M = Problem::kInputDimMax;
  static constexpr int MM = M;
  static constexpr int NA = N + M; //!< augmented dimension [dx; du]
  static_assert(!Problem::kDynamicInput, "static input dimension only");
  static_assert(N % 4 == 0 && N >= 4 && N <= 12, "state rows fill whole register groups: n in {4, 8, 12}");
  static_assert(M >= 1 && M <= 4 && NA <= 16, "the gain rows n .. n+m-1 live in one lane group");
  static constexpr bool kShape = true;
  static constexpr int qK = N / 4; //!< lane group (lane / 16) that holds rows n .. n+3 of a natural-layout matrix
  static constexpr int kGain = MM + MM * N; //!< k_i, K_i (column-major m x n) per timestep: one record in the workspace

  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using StateStateDimMatrix = typename Problem::StateStateDimMatrix;
  using InputInputDimMatrix = typename Problem::InputInputDimMatrix;
  using StateInputDimMatrix = typename Problem::StateInputDimMatrix;

  // ---- LDS record of one (instance, timestep): what the model wave hands to the matrix waves (floats)
  static constexpr int kOffF = 0; //!< [Fx Fu | 0]: 16 columns of n rows, column-major
  static constexpr int kOffL = 16 * N; //!< [[Lxx Lxu],[Lxu^T Luu]] padded to 16 x 16, column-major
  static constexpr int kOffLv = kOffL + 256; //!< [Lx; Lu; 0]
  static constexpr int kOffInvU = kOffLv + 16; //!< 1 / (|u_i| + 1)    :217-221
  static constexpr int kOffZero = kOffInvU + 4; //!< four zeros: what lanes outside an operand read
  static constexpr int kRecRaw = kOffZero + 4;
  //! record stride: an odd number of 16-byte granules, so that the 32 lanes of the model wave (one record each) spread
  //! over the LDS banks when they write the same field
  static constexpr int kRec = ((kRecRaw / 4) % 2 == 1) ? kRecRaw : kRecRaw + 4;
  static constexpr int kRecAt = 0; //!< rec[2][32][kRec]
  // ---- per-slot scalars and flags
  static constexpr int kSlotAt = kRecAt + 2 * kTileInstances * kRec;
  enum SlotField
  {
    sB = 0, //!< int: instance index, -1 = empty slot
    sBw, //!< int: this sweep computes gains for the slot
    sLs, //!< int: the slot takes part in the line search
    sSel, //!< int: half of X / U / cost that holds control_data_
    sLambda,
    sT0, //!< current_t
    sOk, //!< int: backwardPass() returned true
    sDV0,
    sDV1,
    sKrel,
    kNumSlotFields
  };
  static constexpr int kFlagAt = kSlotAt + kNumSlotFields * kTileInstances; //!< ints: any_bw, any_retry, any_ls
  static constexpr int kScratchAt = kFlagAt + 8; //!< [32 slots][16]: q as a row -> q as a column
  static constexpr int kLsAt = kScratchAt + kTileInstances * 16; //!< lsJ[NMPC_HIP_MAX_ALPHA][32]: cost of every trial
  static constexpr int kLdsFloats = kLsAt + NMPC_HIP_MAX_ALPHA * kTileInstances;
  static constexpr size_t kLdsBytes = static_cast<size_t>(kLdsFloats) * sizeof(float);
  static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");

  NMPC_HD static size_t workspaceElems(int T)
  {
    return static_cast<size_t>(T) * kGain;
  }

  const Problem & problem;
  const nmpc_hip_ddp_config & cfg;
  const DeviceBuffersT<float> & buf;
  const int T;
  const int wave;
  const int lane;
  float * lds;

  NMPC_D TileSolver32(const Problem & p, const nmpc_hip_ddp_config & c, const DeviceBuffersT<float> & bf, float * lds_base)
  : problem(p), cfg(c), buf(bf), T(bf.T), wave(static_cast<int>(threadIdx.x) >> 6), lane(static_cast<int>(threadIdx.x) & 63),
    lds(lds_base)
  {
  }

  // ---- LDS views
  NMPC_D float * rec(int parity, int slot) const
  {
    return lds + kRecAt + (parity * kTileInstances + slot) * kRec;
  }
  NMPC_D float & slotF(int field, int slot) const
  {
    return lds[kSlotAt + field * kTileInstances + slot];
  }
  NMPC_D int & slotI(int field, int slot) const
  {
    return reinterpret_cast<int *>(lds)[kSlotAt + field * kTileInstances + slot];
  }
  NMPC_D int & flag(int k) const
  {
    return reinterpret_cast<int *>(lds)[kFlagAt + k];
  }
  NMPC_D float * gainRecord(int b, int i) const
  {
    return buf.wpi_ws + (static_cast<size_t>(b) * T + i) * kGain;
  }
  NMPC_D static void barrier()
  {
    __syncthreads();
  }
  NMPC_D static float readLane(float v, int l)
  {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
  }
  NMPC_D static int uniform(int v)
  {
    return __builtin_amdgcn_readfirstlane(v);
  }
  NMPC_D static float uniformF(float v)
  {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
  }

  /** X^T Y + C on natural-layout registers. */
  NMPC_D static v4f mma(v4f X, v4f Y, v4f C)
  {
#pragma unroll
    for(int s = 0; s < 4; s++)
    {
      C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[s], Y[s], C, 0, 0, 0);
    }
    return C;
  }

  // ===================================================================================================
  // model wave: rollouts    DDPSolver.hpp:83-95 (initial), :536-560 (forwardPass)
  // ===================================================================================================
  /** One rollout per active lane.  initial: u_i = initial_u_list[i] (half `sel` of U), x_0 = current_x; otherwise
      u'_i = (u_i + alpha k_i) + K_i (x'_i - x_i) around the nominal in half `sel`.  store: the trajectory goes to half
      `out_half` of X / U / cost.  Returns sum(cost_list) accumulated in list order. */
  NMPC_D float rollout(bool active, int b, int sel, int out_half, float t0, float alpha, bool initial, bool store) const
  {
    float J = 0;
    if(active)
    {
      const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
      const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM, rows_c = static_cast<size_t>(T + 1);
      const float * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
      const float * Un = buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln;
      float * Xo = buf.X + ((tile * 2 + out_half) * rows_x) * 64 + ln;
      float * Uo = buf.U + ((tile * 2 + out_half) * rows_u) * 64 + ln;
      float * Co = buf.cost + ((tile * 2 + out_half) * rows_c) * 64 + ln;
      StateDimVector x;
      if(initial)
      {
#pragma unroll
        for(int c = 0; c < N; c++)
        {
          x[c] = buf.x0[(tile * N + c) * 64 + ln];
        }
      }
      else
      {
#pragma unroll
        for(int c = 0; c < N; c++)
        {
          x[c] = Xn[static_cast<size_t>(c) * 64]; // x'_0 = x_0    :541
        }
      }
      for(int i = 0; i < T; i++)
      {
        const float t = t0 + i * problem.dt();
        InputDimVector u;
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          u[a] = Un[(static_cast<size_t>(i) * MM + a) * 64];
        }
        if(!initial)
        {
          const float * g = gainRecord(b, i);
          float dx[N];
#pragma unroll
          for(int c = 0; c < N; c++)
          {
            dx[c] = x[c] - Xn[(static_cast<size_t>(i) * N + c) * 64];
          }
#pragma unroll
          for(int a = 0; a < MM; a++)
          {
            float s = 0;
#pragma unroll
            for(int c = 0; c < N; c++)
            {
              s += g[MM + a + MM * c] * dx[c];
            }
            u[a] = (u[a] + alpha * g[a]) + s; // :545-546
          }
        }
        const float c = problem.runningCost(t, x, u);
        if(store)
        {
#pragma unroll
          for(int cc = 0; cc < N; cc++)
          {
            Xo[(static_cast<size_t>(i) * N + cc) * 64] = x[cc];
          }
          if(!initial)
          {
#pragma unroll
            for(int a = 0; a < MM; a++)
            {
              Uo[(static_cast<size_t>(i) * MM + a) * 64] = u[a];
            }
          }
          Co[static_cast<size_t>(i) * 64] = c;
        }
        J += c;
        x = problem.stateEq(t, x, u);
      }
      const float cT = problem.terminalCost(t0 + T * problem.dt(), x);
      if(store)
      {
#pragma unroll
        for(int cc = 0; cc < N; cc++)
        {
          Xo[(static_cast<size_t>(T) * N + cc) * 64] = x[cc];
        }
        Co[static_cast<size_t>(T) * 64] = cT;
      }
      J += cT;
    }
    return J;
  }

  // ===================================================================================================
  // model wave: linearisation of one timestep into the LDS record    DDPSolver.hpp:157-185
  // ===================================================================================================
  NMPC_D static void put4(float * at, float a, float b, float c, float d)
  {
    v4f v = {a, b, c, d};
    *reinterpret_cast<v4f *>(at) = v;
  }

  /** Derivatives at (x_i, u_i) of the slot's current trajectory -> rec(i & 1, slot). */
  NMPC_D void lineariseStep(int slot, int b, int sel, float t0, int i) const
  {
    const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
    const float * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
    const float * Un = buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln;
    StateDimVector x;
    InputDimVector u;
#pragma unroll
    for(int c = 0; c < N; c++)
    {
      x[c] = Xn[(static_cast<size_t>(i) * N + c) * 64];
    }
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      u[a] = Un[(static_cast<size_t>(i) * MM + a) * 64];
    }
    const float t = t0 + i * problem.dt();
    StateStateDimMatrix Fx, Lxx;
    StateInputDimMatrix Fu, Lxu;
    StateDimVector Lx;
    InputDimVector Lu;
    InputInputDimMatrix Luu;
    problem.calcStateEqDeriv(t, x, u, Fx, Fu);
    problem.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
    float * r = rec(i & 1, slot);
    // F = [Fx Fu | 0]
#pragma unroll
    for(int c = 0; c < 16; c++)
    {
#pragma unroll
      for(int r4 = 0; r4 < N; r4 += 4)
      {
        float v[4];
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          v[k] = (c < N) ? Fx(r4 + k, c < N ? c : 0) : ((c < NA) ? Fu(r4 + k, (c >= N && c < NA) ? c - N : 0) : 0.0f);
        }
        put4(r + kOffF + c * N + r4, v[0], v[1], v[2], v[3]);
      }
    }
    // L = [[Lxx Lxu],[Lxu^T Luu]] padded with zeros
#pragma unroll
    for(int c = 0; c < 16; c++)
    {
#pragma unroll
      for(int r4 = 0; r4 < 16; r4 += 4)
      {
        float v[4];
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          const int rr = r4 + k;
          float e = 0.0f;
          if(rr < N && c < N)
          {
            e = Lxx(rr < N ? rr : 0, c < N ? c : 0);
          }
          else if(rr < N && c < NA)
          {
            e = Lxu(rr < N ? rr : 0, (c >= N && c < NA) ? c - N : 0);
          }
          else if(rr < NA && c < N)
          {
            e = Lxu(c < N ? c : 0, (rr >= N && rr < NA) ? rr - N : 0);
          }
          else if(rr < NA && c < NA)
          {
            e = Luu((rr >= N && rr < NA) ? rr - N : 0, (c >= N && c < NA) ? c - N : 0);
          }
          v[k] = e;
        }
        put4(r + kOffL + c * 16 + r4, v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for(int r4 = 0; r4 < 16; r4 += 4)
    {
      float v[4];
#pragma unroll
      for(int k = 0; k < 4; k++)
      {
        const int rr = r4 + k;
        v[k] = (rr < N) ? Lx[rr < N ? rr : 0] : ((rr < NA) ? Lu[(rr >= N && rr < NA) ? rr - N : 0] : 0.0f);
      }
      put4(r + kOffLv + r4, v[0], v[1], v[2], v[3]);
    }
    float un = 0;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      un += u[a] * u[a];
    }
    const float unorm = (M == 1) ? fabsf(u[0]) : sqrtf(un);
    put4(r + kOffInvU, recipFast(unorm + 1.0f), 0.0f, 0.0f, 0.0f);
  }

  /** [Vxx | Vx] of the terminal cost (:177-185, :346-365) in the F region of rec(T & 1, slot). */
  NMPC_D void lineariseTerminal(int slot, int b, int sel, float t0) const
  {
    const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
    const size_t rows_x = static_cast<size_t>(T + 1) * N;
    const float * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
    StateDimVector xT, vx;
    StateStateDimMatrix vxx;
#pragma unroll
    for(int c = 0; c < N; c++)
    {
      xT[c] = Xn[(static_cast<size_t>(T) * N + c) * 64];
    }
    problem.calcTerminalCostDeriv(t0 + T * problem.dt(), xT, vx, vxx);
    float * r = rec(T & 1, slot);
#pragma unroll
    for(int c = 0; c <= N; c++)
    {
#pragma unroll
      for(int r4 = 0; r4 < N; r4 += 4)
      {
        float v[4];
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          v[k] = (c < N) ? vxx(r4 + k, c < N ? c : 0) : vx[r4 + k];
        }
        put4(r + kOffF + c * N + r4, v[0], v[1], v[2], v[3]);
      }
    }
  }

  // ===================================================================================================
  // matrix waves: one backward timestep of one instance    DDPSolver.hpp:381-530
  // ===================================================================================================
  struct BwState
  {
    v4f VV; //!< [Vxx | Vx]
    float dV0, dV1, krel;
    bool ok;
  };

  /** In-place L D L^T of the m x m matrix A (column-major, leading dimension MM) with the pivot rule of Eigen's LLT:
      fails iff a pivot is <= 0, NaN passes.  Same operation order as the lane kernels' ldltInPlace. */
  NMPC_D static bool ldlt(float * A, float * inv_d)
  {
    bool ok = true;
#pragma unroll
    for(int k = 0; k < MM; k++)
    {
      float d = A[k + k * MM];
#pragma unroll
      for(int j = 0; j < k; j++)
      {
        d -= (A[k + j * MM] * A[k + j * MM]) * A[j + j * MM];
      }
      ok = ok && !(d <= 0.0f);
      A[k + k * MM] = d;
      const float r = recipFast(d);
      inv_d[k] = r;
#pragma unroll
      for(int i = k + 1; i < MM; i++)
      {
        float s = A[i + k * MM];
#pragma unroll
        for(int j = 0; j < k; j++)
        {
          s -= (A[i + j * MM] * A[k + j * MM]) * A[j + j * MM];
        }
        A[i + k * MM] = s * r;
      }
    }
    return ok;
  }
  NMPC_D static void ldltSolve(const float * A, const float * inv_d, float * x)
  {
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      float s = x[i];
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
      float s = x[i] * inv_d[i];
#pragma unroll
      for(int j = i + 1; j < MM; j++)
      {
        s -= A[j + i * MM] * x[j];
      }
      x[i] = s;
    }
  }

  /** \param store_gains the slot takes part in this sweep */
  template<int kRegType>
  NMPC_D void backwardStep(BwState & st, int slot, int b, int i, float lambda, bool store_gains) const
  {
    const int q = lane >> 4, j = lane & 15;
    const float * r = rec(i & 1, slot);
    const v4f zero4 = {0, 0, 0, 0};
    // operands of this timestep: rows >= n of F read the record's zero slot
    const v4f F = *reinterpret_cast<const v4f *>(r + ((4 * q < N) ? kOffF + j * N + 4 * q : kOffZero));
    const v4f L = *reinterpret_cast<const v4f *>(r + kOffL + j * 16 + 4 * q);
    const float lv = r[kOffLv + j];
    const float inv_u = r[kOffInvU];
    // ---- Q terms    :386-408
    const v4f G = mma(st.VV, F, zero4);
    const v4f Q = mma(F, G, L);
    const float qrow = lv + G[0]; // lane group qK: q_j = l_j + (Vx^T F)_j  — row n of G is register 0 there (n % 4 == 0)
    // ---- regularisation    :421-441
    v4f Qr = Q;
    if constexpr(kRegType == 2)
    {
      v4f G2;
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        G2[rr] = (4 * q + rr < N) ? G[rr] + lambda * F[rr] : 0.0f; // (Vxx + lambda I) F, without the Vx row
      }
      Qr = mma(F, G2, L);
    }
    // Quu (unregularised), Quu_F and Qu to every lane: columns n .. n+m-1 of rows n .. n+m-1 sit in lanes 16 qK + n + c
    float Quu[MM * MM], QuuF[MM * MM], Qu[MM], inv_d[MM];
#pragma unroll
    for(int c = 0; c < MM; c++)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        Quu[a + c * MM] = readLane(Q[a], 16 * qK + N + c);
        QuuF[a + c * MM] = (kRegType == 2) ? readLane(Qr[a], 16 * qK + N + c) : Quu[a + c * MM];
      }
      if(kRegType == 1)
      {
        QuuF[c + c * MM] += lambda;
      }
      Qu[c] = readLane(qrow, 16 * qK + N + c);
    }
    // ---- gains    :500-517: every lane factorises Quu_F; lane (qK, j) solves column j of [Qux_reg | Qu]
    const bool ok_now = ldlt(QuuF, inv_d);
    st.ok = st.ok && ok_now; // wave-uniform; after a failure the slot keeps computing on garbage and stores nothing
    float col[MM];
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      col[a] = (j == N) ? Qu[a] : Qr[a];
    }
    ldltSolve(QuuF, inv_d, col);
    v4f A; // [[I 0],[K k]]
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      const float gain = (rr < MM && j <= N) ? -1.0f * col[rr < MM ? rr : 0] : 0.0f;
      const float ident = (4 * q + rr == j && j < N) ? 1.0f : 0.0f;
      A[rr] = (q == qK) ? gain : ident;
    }
    float kff[MM];
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      kff[a] = readLane(A[a], 16 * qK + N);
    }
    // ---- dV += [k.Qu, 0.5 k.(Quu k)], |k| / (|u| + 1)    :217-221, :522-523
    {
      float kQu = 0, kQuuk = 0, kn = 0;
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        kQu += kff[a] * Qu[a];
        float s = 0;
#pragma unroll
        for(int p = 0; p < MM; p++)
        {
          s += Quu[a + p * MM] * kff[p];
        }
        kQuuk += kff[a] * s;
        kn += kff[a] * kff[a];
      }
      st.dV0 += kQu;
      st.dV1 += 0.5f * kQuuk;
      const float knorm = (M == 1) ? fabsf(kff[0]) : sqrtf(kn);
      st.krel = fmaxf(st.krel, knorm * inv_u);
    }
    // ---- cost-to-go    :524-527
    // q as a column (lane (q', n) register r <- q[4 q' + r]): through this wave's LDS scratch (one wave's LDS traffic is ordered)
    float * scratch = lds + kScratchAt + slot * 16;
    if(q == qK)
    {
      scratch[j] = qrow;
    }
    asm volatile("" ::: "memory");
    v4f qcol = *reinterpret_cast<const v4f *>(scratch + 4 * q);
    asm volatile("" ::: "memory");
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      qcol[rr] = (j == N) ? qcol[rr] : 0.0f;
    }
    const v4f H = mma(Q, A, qcol);
    const v4f Vn = mma(A, H, zero4);
    const v4f Vt = mma(H, A, zero4);
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      const float sym = (j < N) ? 0.5f * (Vn[rr] + Vt[rr]) : Vn[rr]; // column n is Vx
      st.VV[rr] = (4 * q + rr < N && j <= N) ? sym : 0.0f;
    }
    // ---- save gains    :529-530 (not after a failed factorisation: backwardPass() returned before, :505-508)
    if(store_gains && st.ok && q == qK && j <= N)
    {
      float * g = gainRecord(b, i) + ((j < N) ? MM + MM * j : 0);
      if constexpr(MM == 4)
      {
        *reinterpret_cast<v4f *>(g) = A;
      }
      else
      {
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          g[a] = A[a];
        }
      }
    }
  }

  /** The sweep of the seven matrix waves (barriers are shared with the model wave's loop in solve()). */
  template<int kRegType>
  NMPC_D void backwardSweepMatrix() const
  {
    const int q = lane >> 4, j = lane & 15;
    const int first = tileWaveFirst(wave), count = tileWaveCount(wave);
    BwState st[kTileMaxPerWave];
    int bs[kTileMaxPerWave];
    bool act[kTileMaxPerWave];
    float lam[kTileMaxPerWave];
    barrier(); // the terminal record and the record of timestep T - 1 are complete
#pragma unroll
    for(int e = 0; e < kTileMaxPerWave; e++)
    {
      const int slot = first + (e < count ? e : 0);
      bs[e] = uniform(slotI(sB, slot));
      act[e] = e < count && uniform(slotI(sBw, slot)) != 0;
      lam[e] = uniformF(slotF(sLambda, slot));
      const float * r = rec(T & 1, slot);
      st[e].VV = *reinterpret_cast<const v4f *>(r + ((4 * q < N && j <= N) ? kOffF + j * N + 4 * q : kOffZero));
      st[e].dV0 = 0;
      st[e].dV1 = 0;
      st[e].krel = 0;
      st[e].ok = true;
    }
    barrier(); // the terminal record has been read: the model wave may overwrite its slot with timestep T - 2
    for(int i = T - 1; i >= 0; i--)
    {
#pragma unroll
      for(int e = 0; e < kTileMaxPerWave; e++)
      {
        if(e < count) // wave-uniform
        {
          backwardStep<kRegType>(st[e], first + e, bs[e], i, lam[e], act[e]);
        }
      }
      barrier();
    }
    if(lane == 0)
    {
#pragma unroll
      for(int e = 0; e < kTileMaxPerWave; e++)
      {
        if(act[e])
        {
          const int slot = first + e;
          slotI(sOk, slot) = st[e].ok ? 1 : 0;
          slotF(sDV0, slot) = st[e].dV0;
          slotF(sDV1, slot) = st[e].dV1;
          slotF(sKrel, slot) = st[e].krel;
        }
      }
    }
  }

  /** The model wave's half of the sweep: timestep i - 1 is linearised while the matrix waves consume timestep i. */
  NMPC_D void backwardSweepModel(bool mine, int slot, int b, int sel, float t0) const
  {
    if(mine)
    {
      lineariseTerminal(slot, b, sel, t0);
      lineariseStep(slot, b, sel, t0, T - 1);
    }
    barrier(); // records T (terminal) and T - 1 are complete
    barrier(); // the matrix waves have taken the terminal record
    for(int i = T - 1; i >= 0; i--)
    {
      if(mine && i > 0)
      {
        lineariseStep(slot, b, sel, t0, i - 1);
      }
      barrier();
    }
  }

  // ===================================================================================================
  // solve    DDPSolver.hpp:26-141, procOnce :143-340
  // ===================================================================================================
  NMPC_D void writeTraceRow(int b, int row, const float * tr) const
  {
    if(cfg.trace_level >= 1 && row < buf.trace_rows)
    {
      const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
      float * p = buf.trace + (tile * (static_cast<size_t>(buf.trace_rows) * NMPC_HIP_NTRACE)) * 64 + ln;
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        p[(static_cast<size_t>(row) * NMPC_HIP_NTRACE + f) * 64] = tr[f];
      }
    }
  }

  NMPC_D void solve()
  {
    const bool model_wave = (wave == kTileModelWave);
    const int slot = lane & (kTileInstances - 1);
    const int b = static_cast<int>(blockIdx.x) * kTileInstances + slot;
    // ---- per-instance solver state, held by lane `slot` of the model wave
    const bool owner = model_wave && lane < kTileInstances && b < buf.B;
    const float lambda_factor = static_cast<float>(cfg.lambda_factor), lambda_min = static_cast<float>(cfg.lambda_min),
                lambda_max = static_cast<float>(cfg.lambda_max);
    float lambda = static_cast<float>(cfg.initial_lambda), dlambda = static_cast<float>(cfg.initial_dlambda); // :36-38
    float t0 = 0, J_cur = 0, dV0 = 0, dV1 = 0;
    int sel = 0, iter = 0, retval = 0;
    bool running = owner;
    float tr[NMPC_HIP_NTRACE];
#pragma unroll
    for(int f = 0; f < NMPC_HIP_NTRACE; f++)
    {
      tr[f] = 0;
    }
    // ---- zero slots of the records, slot table
    for(int e = static_cast<int>(threadIdx.x); e < 2 * kTileInstances; e += kTileThreads)
    {
      float * r = lds + kRecAt + e * kRec;
      put4(r + kOffZero, 0.0f, 0.0f, 0.0f, 0.0f);
    }
    if(model_wave && lane < kTileInstances)
    {
      slotI(sB, slot) = (b < buf.B) ? b : -1;
      slotI(sBw, slot) = 0;
      slotI(sLs, slot) = 0;
      slotI(sSel, slot) = 0;
      slotI(sOk, slot) = 1;
    }
    // ---- initial rollout    :83-104
    if(owner)
    {
      t0 = buf.t0 ? buf.t0[b] : 0.0f;
    }
    if(model_wave)
    {
      J_cur = rollout(owner, b, 0, 0, t0, 0.0f, true, true);
      if(owner)
      {
        tr[NMPC_HIP_TRACE_COST] = J_cur;
        tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
        tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
        tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
        writeTraceRow(b, 0, tr);
      }
    }

    for(;;)
    {
      // ---- which slots start procOnce number iter + 1    :115-123
      bool in_iter = false; // this lane's slot is inside a procOnce
      int n_backward = 0;
      if(model_wave)
      {
        in_iter = running && iter < cfg.max_iter;
        if(in_iter)
        {
          iter++;
#pragma unroll
          for(int f = 0; f < NMPC_HIP_NTRACE; f++)
          {
            tr[f] = 0;
          }
          tr[NMPC_HIP_TRACE_ITER] = static_cast<float>(iter);
          tr[NMPC_HIP_TRACE_ALPHA_IDX] = -1;
          retval = 0;
        }
        else
        {
          running = false;
        }
        if(lane < kTileInstances)
        {
          slotI(sBw, slot) = in_iter ? 1 : 0;
          slotI(sSel, slot) = sel;
          slotF(sLambda, slot) = lambda;
          slotF(sT0, slot) = t0;
        }
        const unsigned long long any = __ballot(in_iter);
        if(lane == 0)
        {
          flag(0) = (any != 0) ? 1 : 0;
        }
      }
      barrier();
      if(uniform(flag(0)) == 0)
      {
        break;
      }
      // ---- Steps 1 + 2: linearisation fused into the backward sweep, with the regularisation retries    :157-214
      bool need_bw = in_iter;
      for(;;)
      {
        if(model_wave)
        {
          backwardSweepModel(need_bw && lane < kTileInstances, slot, b, sel, t0);
        }
        else if(cfg.reg_type == 2)
        {
          backwardSweepMatrix<2>();
        }
        else if(cfg.reg_type == 1)
        {
          backwardSweepMatrix<1>();
        }
        else
        {
          backwardSweepMatrix<0>();
        }
        barrier(); // results of the sweep are in the slot table
        if(model_wave)
        {
          bool retry = false;
          if(need_bw)
          {
            n_backward++;
            if(slotI(sOk, slot) == 0)
            {
              dlambda = fmaxf(dlambda * lambda_factor, lambda_factor); // :191-209
              lambda = fmaxf(lambda * dlambda, lambda_min);
              if(lambda > lambda_max)
              {
                retval = -1;
                need_bw = false;
              }
              else
              {
                retry = true;
              }
            }
            else
            {
              need_bw = false;
              dV0 = slotF(sDV0, slot);
              dV1 = slotF(sDV1, slot);
            }
          }
          if(lane < kTileInstances)
          {
            slotI(sBw, slot) = retry ? 1 : 0;
            slotF(sLambda, slot) = lambda;
          }
          const unsigned long long any = __ballot(retry);
          if(lane == 0)
          {
            flag(1) = (any != 0) ? 1 : 0;
          }
        }
        barrier();
        if(uniform(flag(1)) == 0)
        {
          break;
        }
      }
      // ---- small-gradient termination (:217-231), then Step 3: the line search, every step size at once    :234-274
      bool in_ls = false;
      if(model_wave)
      {
        if(in_iter)
        {
          tr[NMPC_HIP_TRACE_N_BACKWARD] = static_cast<float>(n_backward);
          if(retval == 0)
          {
            const float krel = slotF(sKrel, slot);
            tr[NMPC_HIP_TRACE_K_REL_NORM] = krel;
            if(krel < static_cast<float>(cfg.k_rel_norm_thre) && lambda < static_cast<float>(cfg.lambda_thre))
            {
              retval = 1;
            }
            else
            {
              in_ls = true;
            }
          }
        }
        if(lane < kTileInstances)
        {
          slotI(sLs, slot) = in_ls ? 1 : 0;
        }
        const unsigned long long any = __ballot(in_ls);
        if(lane == 0)
        {
          flag(2) = (any != 0) ? 1 : 0;
        }
      }
      barrier();
      if(uniform(flag(2)) != 0)
      {
        float J_first = 0;
        if(model_wave)
        {
          J_first = rollout(in_ls, b, sel, sel ^ 1, t0, static_cast<float>(cfg.alpha_list[0]), false, true);
        }
        else
        {
          // fan-out: matrix wave w rolls out step sizes base + 2 (w - 1) (lanes 0..31) and base + 2 (w - 1) + 1 (lanes 32..63)
          for(int base = 1; base < cfg.n_alpha; base += 2 * kTileMatrixWaves)
          {
            const int ai = base + 2 * (wave - 1) + (lane >> 5);
            const bool act = ai < cfg.n_alpha && slotI(sB, slot) >= 0 && slotI(sLs, slot) != 0;
            const int fb = slotI(sB, slot), fsel = slotI(sSel, slot);
            const float ft0 = slotF(sT0, slot);
            const float Jc = rollout(act, fb, fsel, fsel ^ 1, ft0, static_cast<float>(cfg.alpha_list[act ? ai : 0]), false, false);
            if(act)
            {
              lds[kLsAt + ai * kTileInstances + slot] = Jc;
            }
          }
        }
        barrier();
        if(model_wave)
        {
          int ai_taken = cfg.n_alpha - 1;
          bool success = false;
          float alpha = 0, actual = 0, expected = 0, ratio = 0, J_cand = 0;
          if(in_ls)
          {
            for(int ai = 0; ai < cfg.n_alpha; ai++)
            {
              const float Jc = (ai == 0) ? J_first : lds[kLsAt + ai * kTileInstances + slot];
              alpha = static_cast<float>(cfg.alpha_list[ai]);
              actual = J_cur - Jc;
              expected = -1.0f * alpha * (dV0 + alpha * dV1);
              ratio = actual / expected;
              if(expected < 0)
              {
                ratio = (actual >= 0 ? 1.0f : -1.0f); // :251-259
              }
              J_cand = Jc;
              if(ratio > static_cast<float>(cfg.cost_update_ratio_thre))
              {
                success = true;
                ai_taken = ai;
                break;
              }
            }
          }
          // a step size other than the first one was taken: its trajectory has not been stored yet
          const bool reroll = in_ls && success && ai_taken > 0;
          if(__ballot(reroll) != 0)
          {
            const float Jr = rollout(reroll, b, sel, sel ^ 1, t0, alpha, false, true);
            if(reroll)
            {
              J_cand = Jr; // (the same instruction stream on the same inputs: the same value)
            }
          }
          if(in_ls)
          {
            tr[NMPC_HIP_TRACE_ALPHA] = alpha;
            tr[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL] = actual;
            tr[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED] = expected;
            tr[NMPC_HIP_TRACE_COST_UPDATE_RATIO] = ratio;
            tr[NMPC_HIP_TRACE_ALPHA_IDX] = static_cast<float>(ai_taken);
            tr[NMPC_HIP_TRACE_N_FORWARD] = static_cast<float>(success ? ai_taken + 1 : cfg.n_alpha);
            // ---- Step 4    :280-333
            if(success)
            {
              sel ^= 1;
              J_cur = J_cand;
              if(actual < static_cast<float>(cfg.cost_update_thre))
              {
                retval = 1;
              }
              dlambda = fminf(dlambda / lambda_factor, 1.0f / lambda_factor);
              if(lambda >= lambda_min)
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
              dlambda = fmaxf(dlambda * lambda_factor, lambda_factor);
              lambda = fmaxf(lambda * dlambda, lambda_min);
              if(lambda > lambda_max)
              {
                retval = -1;
              }
            }
            tr[NMPC_HIP_TRACE_COST] = J_cur;
            tr[NMPC_HIP_TRACE_LAMBDA] = lambda;
            tr[NMPC_HIP_TRACE_DLAMBDA] = dlambda;
          }
        }
      }
      if(model_wave && in_iter)
      {
        writeTraceRow(b, iter, tr);
        if(retval != 0)
        {
          running = false; // :118-122
        }
      }
    }

    // ---- results the host reads per instance
    if(owner)
    {
      const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
      buf.status[b] = retval;
      buf.iters[b] = iter;
      buf.sel[b] = sel;
      buf.dV[(tile * 2 + 0) * 64 + ln] = dV0;
      buf.dV[(tile * 2 + 1) * 64 + ln] = dV1;
#pragma unroll
      for(int f = 0; f < NMPC_HIP_NTRACE; f++)
      {
        buf.trace_last[(tile * NMPC_HIP_NTRACE + f) * 64 + ln] = tr[f];
      }
      for(int i = 0; i < T; i++)
      {
        buf.input_dim[(tile * T + i) * 64 + ln] = M;
      }
    }
  }
};

/** The fp32 tile kernel: grid = ceil(B / 32) workgroups of eight wavefronts. */
template<class Problem>
__global__ __launch_bounds__(kTileThreads) void ddp_solve_tile32_kernel(const Problem problem,
                                                                        const nmpc_hip_ddp_config cfg,
                                                                        const DeviceBuffersT<float> buf)
{
  extern __shared__ __attribute__((aligned(16))) float lds_tile32[];
  TileSolver32<Problem> solver(problem, cfg, buf, lds_tile32);
  solver.solve();
}

/** Type-erased operations (model_ops.hpp) of an fp32 problem type served by the tile kernel. */
template<class Problem>
struct ModelOpsTile32
{
  using Solver = TileSolver32<Problem>;
  static void defaultParams(void * out)
  {
    new(out) Problem();
  }
  static const char * kernelName(int)
  {
    return "ddp_solve_tile32_kernel";
  }
  static hipError_t launchSolve(const void * params, const nmpc_hip_ddp_config & cfg, const DeviceBuffers & buf64,
                                hipStream_t stream)
  {
    if(cfg.with_input_constraint != 0 || buf64.params_batch != nullptr || buf64.wpi_ws == nullptr)
    {
      return hipErrorNotSupported; // BoxQP / per-instance problem objects: fp64 kernel families only
    }
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    // the handle allocates every Scalar array with sizeof(Problem::Scalar) = 4 (ModelOps::scalar_bytes): same pointers, float view
    DeviceBuffersT<float> buf;
    buf.B = buf64.B;
    buf.Bp = buf64.Bp;
    buf.T = buf64.T;
    buf.trace_rows = buf64.trace_rows;
    buf.t0 = reinterpret_cast<const float *>(buf64.t0);
    buf.x0 = reinterpret_cast<const float *>(buf64.x0);
    buf.X = reinterpret_cast<float *>(buf64.X);
    buf.U = reinterpret_cast<float *>(buf64.U);
    buf.cost = reinterpret_cast<float *>(buf64.cost);
    buf.kff = reinterpret_cast<float *>(buf64.kff);
    buf.Kfb = reinterpret_cast<float *>(buf64.Kfb);
    buf.trace = reinterpret_cast<float *>(buf64.trace);
    buf.trace_last = reinterpret_cast<float *>(buf64.trace_last);
    buf.dV = reinterpret_cast<float *>(buf64.dV);
    buf.status = buf64.status;
    buf.iters = buf64.iters;
    buf.sel = buf64.sel;
    buf.qp_ret = buf64.qp_ret;
    buf.qp_free = buf64.qp_free;
    buf.input_dim = buf64.input_dim;
    buf.wpi_ws = reinterpret_cast<float *>(buf64.wpi_ws);
    buf.params_batch = nullptr;
    buf.lim_batch = nullptr;
    for(int i = 0; i < kMaxInputDim; i++)
    {
      buf.lim_lo[i] = buf64.lim_lo[i];
      buf.lim_hi[i] = buf64.lim_hi[i];
    }
    constexpr size_t lds_bytes = Solver::kLdsBytes;
    static bool requested[64] = {};
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
    {
      return hipErrorInvalidDevice;
    }
    if(!requested[dev])
    {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ddp_solve_tile32_kernel<Problem>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
      if(e != hipSuccess)
      {
        return e;
      }
      requested[dev] = true;
    }
    const dim3 g((buf.B + kTileInstances - 1) / kTileInstances), blk(kTileThreads);
    hipLaunchKernelGGL((ddp_solve_tile32_kernel<Problem>), g, blk, lds_bytes, stream, problem, cfg, buf);
    return hipGetLastError();
  }
  static hipError_t launchMpcAdvance(const void *, const DeviceBuffers &, const MpcAdvanceArgs &, hipStream_t)
  {
    return hipErrorNotSupported; // the receding-horizon driver runs on the fp64 kernel families
  }
  static void inputDims(const void *, double, int T, int * out)
  {
    for(int i = 0; i < T; i++)
    {
      out[i] = Problem::kInputDimMax;
    }
  }
  static double dt(const void * params)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    return static_cast<double>(problem.dt());
  }
  static size_t workspaceElems(int T)
  {
    return Solver::workspaceElems(T);
  }
  static ModelOps make()
  {
    static_assert(std::is_trivially_copyable<Problem>::value, "a DDP problem must be trivially copyable: it is passed to the GPU by value");
    static_assert(std::is_default_constructible<Problem>::value, "a DDP problem must be default constructible");
    ModelOps ops;
    ops.name = Problem::kName;
    ops.state_dim = Problem::kStateDim;
    ops.input_dim_max = Problem::kInputDimMax;
    ops.dynamic_input = 0;
    ops.param_bytes = sizeof(Problem);
    ops.default_params = &defaultParams;
    ops.launch_solve = &launchSolve;
    ops.input_dims = &inputDims;
    ops.dt = &dt;
    ops.kernel_name = &kernelName;
    ops.launch_mpc_advance = &launchMpcAdvance;
    ops.has_plant_step = 0;
    ops.wpi_workspace_doubles = &workspaceElems;
    ops.scalar_bytes = 4;
    ops.gain_layout = 1;
    return ops;
  }
};
} // namespace hip
} // namespace nmpc_amd

#define NMPC_AMD_REGISTER_PROBLEM_TILE32(ProblemType) \
  NMPC_AMD_REGISTER_PROBLEM_WITH(ProblemType, nmpc_amd::hip::ModelOpsTile32<ProblemType>)
