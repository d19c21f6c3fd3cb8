// gfx950 device code of the batched DDP solver, fp32, LANE MAPPING "TILE32": one workgroup of sixteen wavefronts solves 32
// problem instances whose (n + m) x (n + m) augmented blocks are ONE 16 x 16 matrix-core tile — BASELINE.json config 4
// (quadrotor n 12, m 4, T 50, batch 8192, fp32: 8192 / 32 = 256 workgroups = one per CU).
//
// What a lane means changes with the phase (reference: nmpc_ddp/include/nmpc_ddp/DDPSolver.hpp):
//
//   model code  — initial rollout (:83-95) and the linearisation sweep (:157-185): lane = INSTANCE (wave 0, "model wave");
//                 the line search (:234-274): lane = (instance, step size), all step sizes of alpha_list at once (the trials
//                 are independent: same nominal, same gains), all the step sizes of an instance in one wave.
//   backward    — (:342-534) lane = MATRIX ENTRY on waves 1..15, two or three instances per wave, on v_mfma_f32_16x16x4_f32.
//
// Derivatives never reach HBM and are never materialised for the whole horizon either: the model wave linearises timestep
// i - 1 of all 32 instances into an LDS record while the fifteen matrix waves consume the record of timestep i (two record
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
// with or without box constraints on the inputs (kConstrained: the BoxQP of BoxQP.h:141-347 in float, by every lane of the
// instance's gain rows); a shared problem object or one per instance (kOwnProblem); the receding-horizon driver
// (mpc_kernels.hpp) advances the handle's float arrays in place.
#pragma once

#include <atomic>

#include <cstring>
#include <new>
#include <type_traits>

#include <nmpc_amd/hip/model_ops.hpp>
#include <nmpc_amd/hip/mpc_kernels.hpp>
#include <nmpc_amd/hip/ddp_kernels_tile64.hpp>

namespace nmpc_amd
{
namespace hip
{
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef NMPC_TILE32_WAVES
#  define NMPC_TILE32_WAVES 12
#endif
constexpr int kTileInstances = 32; //!< instances per workgroup
//! wavefronts per workgroup: 8 (two per SIMD, 256 registers each), 12 (three, 168) or 16 (four, 128).  More waves hide
//! more of the backward chains' latencies, fewer leave the model code (rollouts, linearisation) its registers.
constexpr int kTileWaves = NMPC_TILE32_WAVES;
static_assert(kTileWaves == 8 || kTileWaves == 12 || kTileWaves == 16, "8, 12 or 16 wavefronts per workgroup");
constexpr int kTileThreads = kTileWaves * 64;
constexpr int kTileModelWave = 0; //!< model code, lane = instance
constexpr int kTileMatrixWaves = kTileWaves - 1; //!< waves 1 ..: backward pass, line-search fan-out
/** Instances of matrix wave w >= 1.  Wave w runs on SIMD w % 4; the matrix waves that share SIMD 0 with the model wave
    take fewer instances (the linearisation of 32 instances costs about as many instructions per timestep as three to four
    backward steps):   8 waves: 5 5 5 | 2 | 5 5 5      12 waves: 3 3 3 | 3 | 3 3 3 | 2 | 3 3 3      16 waves: 2 each, 3 on waves 13, 14. */
__host__ __device__ constexpr int tileWaveCount(int w)
{
  return kTileWaves == 8 ? (w == 4 ? 2 : 5) : (kTileWaves == 12 ? (w == 8 ? 2 : 3) : ((w == 13 || w == 14) ? 3 : 2));
}
constexpr int kTileMaxPerWave = kTileWaves == 8 ? 5 : 3;
constexpr int kTileMinPerWave = 2;
__host__ __device__ constexpr int tileWaveFirst(int w)
{
  int first = 0;
  for(int v = 1; v < w; v++)
  {
    first += tileWaveCount(v);
  }
  return first;
}
static_assert(tileWaveFirst(kTileWaves) == kTileInstances, "the matrix waves cover the 32 slots");

template<class Problem, bool kOwnProblem = false, bool kConstrained = false>
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
  static constexpr int kOffL = 16 * N; //!< [[Lxx Lxu],[Lxu^T Luu]] padded to 16 x 16, column-major, 4-row groups rotated by column / 4
  static constexpr int kOffLv = kOffL + 256; //!< [Lx; Lu; 0]
  static constexpr int kOffInvU = kOffLv + 16; //!< 1 / (|u_i| + 1)    :217-221
  static constexpr int kRecRaw = kOffInvU + 4;
  //! record stride: an odd number of 16-byte granules, so that the 32 lanes of the model wave (one record each) spread
  //! over the LDS banks when they write the same field
  static constexpr int kRec = ((kRecRaw / 4) % 2 == 1) ? kRecRaw : kRecRaw + 4;
  static constexpr int kRecAt = 0; //!< rec[2][32][kRec]
  //! [Vxx | Vx] of the terminal cost per slot: n + 1 columns of n rows (its own area: the step records are then only ever
  //! written by lineariseStep, which lets it skip what it knows to be constant)
  static constexpr int kTermRaw = (N + 1) * N;
  static constexpr int kTerm = ((kTermRaw / 4) % 2 == 1) ? kTermRaw : kTermRaw + 4;
  static constexpr int kTermAt = kRecAt + 2 * kTileInstances * kRec;
  // ---- per-slot scalars and flags
  static constexpr int kSlotAt = kTermAt + kTileInstances * kTerm;
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
  static constexpr int kScratchAt = kFlagAt + 8; //!< [32 slots][16]: q as a row -> q as a column; then a 16-word dump
  static constexpr int kLsAt = kScratchAt + kTileInstances * 16 + 16; //!< lsJ[NMPC_HIP_MAX_ALPHA][32]: cost of every trial
  //! per wave: the new value function, written row by row (leading dimension 20: conflict-free both ways) and read back
  //! transposed for Vxx <- (Vxx + Vxx^T) / 2; then 16 zeros for the lanes outside the block
  static constexpr int kTrLd = 20;
  static constexpr int kTrFloats = 16 * kTrLd + 16;
  static constexpr int kTrAt = kLsAt + NMPC_HIP_MAX_ALPHA * kTileInstances;
  static constexpr int kLdsFloats = kTrAt + kTileWaves * kTrFloats;
  static constexpr size_t kLdsBytes = static_cast<size_t>(kLdsFloats) * sizeof(float);
  static_assert(kLdsBytes <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");

  NMPC_HD static size_t workspaceElems(int T)
  {
    return static_cast<size_t>(T) * kGain;
  }

  //! the problem object of this lane's instance: the handle's shared one (uniform: scalar registers), or — batches with
  //! per-instance objects, nmpc_hip_ddp_set_model_params_batch — the lane's own copy.  Only the model wave evaluates the
  //! problem, and its lane `slot` stands for instance 32 blockIdx.x + slot in the rollouts and in the sweeps alike.
  std::conditional_t<kOwnProblem, const Problem, const Problem &> problem;
  const nmpc_hip_ddp_config & cfg;
  const DeviceBuffersT<float> & buf;
  const int T;
  const int wave;
  const int lane;
  float * lds;

  NMPC_D static decltype(auto) problemOfLane(const Problem & p, const DeviceBuffersT<float> & bf)
  {
    if constexpr(kOwnProblem)
    {
      const int b = static_cast<int>(blockIdx.x) * kTileInstances + (static_cast<int>(threadIdx.x) & (kTileInstances - 1));
      const bool evaluates = (static_cast<int>(threadIdx.x) >> 6) == kTileModelWave && b < bf.B;
      return evaluates ? instanceProblem(p, bf, b) : p;
    }
    else
    {
      return (p);
    }
  }
  NMPC_D TileSolver32(const Problem & p, const nmpc_hip_ddp_config & c, const DeviceBuffersT<float> & bf, float * lds_base)
  : problem(problemOfLane(p, bf)), cfg(c), buf(bf), T(bf.T), wave(static_cast<int>(threadIdx.x) >> 6), lane(static_cast<int>(threadIdx.x) & 63),
    lds(lds_base)
  {
  }

  // ---- LDS views
  NMPC_D float * rec(int parity, int slot) const
  {
    return lds + kRecAt + (parity * kTileInstances + slot) * kRec;
  }
  NMPC_D float * term(int slot) const
  {
    return lds + kTermAt + slot * kTerm;
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
    syncThreadsFuzzed(5); // (LDS only: s_waitcnt lgkmcnt(0); s_barrier)
  }
  /** Between the phases of an iteration: the gains the matrix waves stored and the trajectory the model wave stored are
      loaded by other waves in the next phase (fullBarrier(), ddp_kernels.hpp). */
  NMPC_D static void phaseBarrier()
  {
    fullBarrier();
  }
  /** Lane kSrc of this lane's 16-lane row in every lane of the row (DPP row_newbcast: one VALU move, no LDS, no SGPR). */
  template<int kSrc>
  NMPC_D static float rowBcast(float v)
  {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + kSrc, 0xf, 0xf, true));
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
  /** One record entry.  kFull = false: entries the compiler knows to be constants (the literal zeros and ones of the model's
      Jacobians, the padding) are not written again — the record slot already holds them from the sweep's first two (full)
      timesteps.  Adjacent stores are merged by the compiler. */
  template<bool kFull>
  NMPC_D static void put(float * at, float v)
  {
    if(kFull || !__builtin_constant_p(v))
    {
      *at = v;
    }
  }

  struct Point
  {
    float x[N], u[MM];
  };
  /** (x_i, u_i) of the slot's current trajectory: requested one timestep before lineariseStep consumes it. */
  NMPC_D void loadPoint(Point & p, int b, int sel, int i) const
  {
    const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
    const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
    const float * Xn = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
    const float * Un = buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln;
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

  /** Derivatives at (x_i, u_i) -> rec(i & 1, slot). */
  template<bool kFull>
  NMPC_D void lineariseStep(int slot, float t0, int i, const Point & p) const
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
      for(int rr = 0; rr < N; rr++)
      {
        const float v = (c < N) ? Fx(rr, c < N ? c : 0) : ((c < NA) ? Fu(rr, (c >= N && c < NA) ? c - N : 0) : 0.0f);
        put<kFull>(r + kOffF + c * N + rr, v);
      }
    }
    // L = [[Lxx Lxu],[Lxu^T Luu]] padded with zeros
#pragma unroll
    for(int c = 0; c < 16; c++)
    {
#pragma unroll
      for(int rr = 0; rr < 16; rr++)
      {
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
        put<kFull>(r + kOffL + c * 16 + 4 * (((rr >> 2) + (c >> 2)) & 3) + (rr & 3), e);
      }
    }
#pragma unroll
    for(int rr = 0; rr < 16; rr++)
    {
      const float v = (rr < N) ? Lx[rr < N ? rr : 0] : ((rr < NA) ? Lu[(rr >= N && rr < NA) ? rr - N : 0] : 0.0f);
      put<kFull>(r + kOffLv + rr, v);
    }
    float un = 0;
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      un += u[a] * u[a];
    }
    const float unorm = (M == 1) ? fabsf(u[0]) : sqrtf(un);
    r[kOffInvU] = recipFast(unorm + 1.0f);
  }

  /** [Vxx | Vx] of the terminal cost (:177-185, :346-365) -> term(slot). */
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
    float * r = term(slot);
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
        put4(r + c * N + r4, v[0], v[1], v[2], v[3]);
      }
    }
  }

  // ===================================================================================================
  // matrix waves: one backward timestep of one instance    DDPSolver.hpp:381-530
  // ===================================================================================================
  /** Box-constrained solves: what the BoxQP of the instance's next timestep needs (DDPSolver.hpp:452-472). */
  struct BwBox
  {
    float k_next[MM]; //!< k of the timestep processed before (i + 1): the warm start
    float lo[MM], up[MM]; //!< input limits - u_i of the timestep to process next, requested one timestep ahead
    int i; //!< that timestep
    int b, sel; //!< instance, half of U that holds the nominal inputs
    bool act; //!< the sweep computes gains for this slot
  };
  struct BwNoBox
  {
  };
  struct BwState
  {
    v4f VV; //!< [Vxx | Vx]
    float dV0, dV1, krel;
    bool ok;
    std::conditional_t<kConstrained, BwBox, BwNoBox> box;
  };

  // ---- BoxQP (BoxQP.h:141-347) for the m <= 4 inputs of an instance, evaluated by every lane (the lanes of lane group qK use it)
  /** What DDPSolver.hpp:473-497 reads after solve(): x, retval, the free set and the factor of H[free, free]. */
  struct QPOut
  {
    float x[MM];
    float fac[MM * MM], inv_d[MM]; //!< L D L^T of H with the clamped rows / columns replaced by the identity's
    unsigned free; //!< bit a: input a is free
    int retval;
  };
  NMPC_D static float qpObjective(const float * H, const float * g, const float * x)
  {
#pragma clang fp contract(on)
    float xg = 0, xHx = 0;
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      xg += x[i] * g[i];
    }
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      float hx = 0;
#pragma unroll
      for(int j = 0; j < MM; j++)
      {
        hx += H[i + j * MM] * x[j];
      }
      xHx += x[i] * hx;
    }
    return xg + 0.5f * xHx;
  }
  /** The projected-Newton iteration of BoxQP.h:168-337, statement by statement as the lane kernels' boxQP (ddp_kernels.hpp),
      without index lists: the free block is H with the clamped rows and columns replaced by those of the identity, which
      the factorisation and the substitutions pass through with exact zeros — the free entries see the same operations in
      the same order as on the compacted block. */
  NMPC_D void boxQP(const float * H, const float * g, const float * lower, const float * upper, const float * initial_x, QPOut & out) const
  {
#pragma clang fp contract(on)
    const float grad_thre = static_cast<float>(cfg.qp_grad_thre), rel_thre = static_cast<float>(cfg.qp_rel_improve_thre),
                armijo = static_cast<float>(cfg.qp_armijo_param), step_factor = static_cast<float>(cfg.qp_step_factor),
                min_step = static_cast<float>(cfg.qp_min_step);
    float * x = out.x;
#pragma unroll
    for(int i = 0; i < MM; i++)
    {
      x[i] = fmaxf(fminf(initial_x[i], upper[i]), lower[i]); // BoxQP.h:148
      out.inv_d[i] = 0;
    }
    float obj = qpObjective(H, g, x);
    float old_obj = obj;
    out.retval = 0;
    out.free = 0;
    unsigned clamped = 0, old_clamped = 0;
    float grad[MM], search_dir[MM], x_cand[MM], rhs[MM];
    for(int iter = 1;; iter++)
    {
      if(iter > 1 && (old_obj - obj) < rel_thre * fabsf(old_obj)) // BoxQP.h:176-181
      {
        out.retval = 4;
        break;
      }
      old_obj = obj;
#pragma unroll
      for(int i = 0; i < MM; i++) // BoxQP.h:184
      {
        float hx = 0;
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
            out.fac[i + j * MM] = both ? H[i + j * MM] : ((i == j) ? 1.0f : 0.0f);
          }
        }
        if(!ldlt(out.fac, out.inv_d))
        {
          out.retval = -1;
          break;
        }
      }
      float grad_norm = 0; // BoxQP.h:244-253
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        grad_norm += ((clamped >> i) & 1u) ? 0.0f : grad[i] * grad[i];
      }
      if(grad_norm < grad_thre * grad_thre)
      {
        out.retval = 5;
        break;
      }
#pragma unroll
      for(int i = 0; i < MM; i++) // BoxQP.h:256-279
      {
        float sum = 0;
#pragma unroll
        for(int j = 0; j < MM; j++)
        {
          sum += ((clamped >> j) & 1u) ? H[i + j * MM] * x[j] : 0.0f;
        }
        rhs[i] = ((clamped >> i) & 1u) ? 0.0f : g[i] + sum;
      }
      ldltSolve(out.fac, out.inv_d, rhs);
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        search_dir[i] = ((clamped >> i) & 1u) ? 0.0f : -1.0f * rhs[i] - x[i];
      }
      float sdg = 0; // BoxQP.h:282-291
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        sdg += search_dir[i] * grad[i];
      }
      if(sdg > 1e-10f)
      {
        out.retval = -2;
        break;
      }
      float step = 1; // BoxQP.h:294-309
#pragma unroll
      for(int i = 0; i < MM; i++)
      {
        x_cand[i] = fmaxf(fminf(x[i] + step * search_dir[i], upper[i]), lower[i]);
      }
      float obj_cand = qpObjective(H, g, x_cand);
      while((obj_cand - old_obj) / (step * sdg) < armijo)
      {
        step = step * step_factor;
#pragma unroll
        for(int i = 0; i < MM; i++)
        {
          x_cand[i] = fmaxf(fminf(x[i] + step * search_dir[i], upper[i]), lower[i]);
        }
        obj_cand = qpObjective(H, g, x_cand);
        if(step < min_step)
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
  }
  /** Input limits - u_i of instance b (DDPSolver.hpp:470-472) into the sweep state, for the timestep st.box.i. */
  NMPC_D void requestLimits(BwState & st) const
  {
    if constexpr(kConstrained)
    {
      const int i = st.box.i > 0 ? st.box.i : 0;
      const int b = st.box.b >= 0 ? st.box.b : 0;
      const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
      const float * Un = buf.U + ((tile * 2 + st.box.sel) * (static_cast<size_t>(T) * MM)) * 64 + ln;
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        const float u = Un[(static_cast<size_t>(i) * MM + a) * 64];
        st.box.lo[a] = static_cast<float>(inputLimitLo(buf, b, i, a)) - u;
        st.box.up[a] = static_cast<float>(inputLimitHi(buf, b, i, a)) - u;
      }
    }
  }

  /** In-place L D L^T of the m x m matrix A (column-major, leading dimension MM) with the pivot rule of Eigen's LLT:
      fails iff a pivot is <= 0, NaN passes.  Same operation order as the lane kernels' ldltInPlace. */
  NMPC_D static bool ldlt(float * A, float * inv_d)
  {
#pragma clang fp contract(on) // fuse a * b + c only within a source expression: every instance gets the same instruction
                              // sequence whichever slot of the stage pipeline it runs in (hipcc's default fuses across
                              // statements, decided per inlined context) — sharded == unsharded, bit for bit

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
#pragma clang fp contract(on) // fuse a * b + c only within a source expression: every instance gets the same instruction
                              // sequence whichever slot of the stage pipeline it runs in (hipcc's default fuses across
                              // statements, decided per inlined context) — sharded == unsharded, bit for bit

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

  /** Row broadcasts of the M x M block at rows / columns n .. n+m-1 of Q (and of Q_reg, reg_type 2) and of Qu. */
  template<int kRegType>
  NMPC_D static void bcastBlock(v4f Q, v4f Qr, float qrow, float lambda, std::integral_constant<int, kRegType>, float * Quu,
                                float * QuuF, float * Qu)
  {
    bcastColumn<0, kRegType>(Q, Qr, qrow, lambda, Quu, QuuF, Qu);
  }
  template<int C, int kRegType>
  NMPC_D static void bcastColumn(v4f Q, v4f Qr, float qrow, float lambda, float * Quu, float * QuuF, float * Qu)
  {
    if constexpr(C < MM)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        Quu[a + C * MM] = rowBcast<N + C>(Q[a]);
        QuuF[a + C * MM] = (kRegType == 2) ? rowBcast<N + C>(Qr[a]) : Quu[a + C * MM];
      }
      if(kRegType == 1)
      {
        QuuF[C + C * MM] += lambda;
      }
      Qu[C] = rowBcast<N + C>(qrow);
      bcastColumn<C + 1, kRegType>(Q, Qr, qrow, lambda, Quu, QuuF, Qu);
    }
  }
  NMPC_D static void bcastK(v4f A, float * kff)
  {
#pragma unroll
    for(int a = 0; a < MM; a++)
    {
      kff[a] = rowBcast<N>(A[a]);
    }
  }

  // The backward timestep of one instance in three branch-free stages.  The instances of a matrix wave run them staggered
  // (software pipeline, see backwardSweepMatrix): while one instance is in the matrix-core heavy stage 1 or 3, another is in
  // the VALU-only stage 2 (factorisation, substitutions) — its instructions fill the gaps between the other's MFMAs.
  struct Stage1Out
  {
    v4f Q, Qr; //!< [[Qxx Qxu],[Qux Quu]] unregularised / with Vxx + lambda I (reg_type 2; else the same registers)
    float qrow; //!< lane group qK: [Qx; Qu]_j
    float inv_u; //!< 1 / (|u_i| + 1)
  };
  struct Stage2Out
  {
    v4f A; //!< [[I 0],[K k]]
    v4f qcol; //!< [Qx; Qu] as column n of the accumulator of H
  };

  /** Stage 1: the record's operands, G = VV^T F, Q = F^T G + L (8 MFMAs; 12 with reg_type 2)    :386-441 */
  template<int kRegType>
  NMPC_D Stage1Out stage1(const BwState & st, int slot, int i, float lambda) const
  {
#pragma clang fp contract(on) // fuse a * b + c only within a source expression: every instance gets the same instruction
                              // sequence whichever slot of the stage pipeline it runs in (hipcc's default fuses across
                              // statements, decided per inlined context) — sharded == unsharded, bit for bit

    const int q = lane >> 4, j = lane & 15;
    const float * r = rec(i & 1, slot);
    const v4f zero4 = {0, 0, 0, 0};
    // rows >= n of F read the zeros behind the wave's transposition scratch
    const v4f F = *reinterpret_cast<const v4f *>((4 * q < N) ? r + kOffF + j * N + 4 * q : lds + kTrAt + wave * kTrFloats + 16 * kTrLd);
    // (the four-row groups of column j are rotated by j / 4: the 16 lanes of a group then read 16 different bank quadruples)
    const v4f L = *reinterpret_cast<const v4f *>(r + kOffL + j * 16 + 4 * ((q + (j >> 2)) & 3));
    const float lv = r[kOffLv + j];
    Stage1Out o;
    o.inv_u = r[kOffInvU];
    const v4f G = mma(st.VV, F, zero4);
    o.Q = mma(F, G, L);
    o.qrow = lv + G[0]; // lane group qK: q_j = l_j + (Vx^T F)_j  — row n of G is register 0 there (n % 4 == 0)
    o.Qr = o.Q;
    if constexpr(kRegType == 2)
    {
      v4f G2;
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        G2[rr] = (4 * q + rr < N) ? G[rr] + lambda * F[rr] : 0.0f; // (Vxx + lambda I) F, without the Vx row
      }
      o.Qr = mma(F, G2, L);
    }
    return o;
  }

  /** Stage 2: gains (:500-517), dV (:522-523), |k| / (|u| + 1) (:217-221) — VALU only. */
  template<int kRegType>
  NMPC_D Stage2Out stage2(BwState & st, const Stage1Out & in, int slot, float lambda) const
  {
#pragma clang fp contract(on) // fuse a * b + c only within a source expression: every instance gets the same instruction
                              // sequence whichever slot of the stage pipeline it runs in (hipcc's default fuses across
                              // statements, decided per inlined context) — sharded == unsharded, bit for bit

    const int q = lane >> 4, j = lane & 15;
    // q as a column (lane (q', n) register r <- q[4 q' + r]) through the slot's LDS scratch, written now and read at the end
    // of the stage (one wave's LDS traffic is ordered); lanes outside group qK write to a dump behind the scratch: no branch
    float * scratch = lds + kScratchAt + slot * 16;
    lds[kScratchAt + ((q == qK) ? slot * 16 + j : kTileInstances * 16 + j)] = in.qrow;
    // Quu (unregularised), Quu_F and Qu in every lane of lane group qK: entry (a, c) is register a of lane n + c of that row
    float Quu[MM * MM], QuuF[MM * MM], Qu[MM], inv_d[MM];
    bcastBlock(in.Q, in.Qr, in.qrow, lambda, std::integral_constant<int, kRegType>(), Quu, QuuF, Qu);
    Stage2Out o;
    if constexpr(kConstrained)
    {
      // ---- box-constrained gains    :450-497: k = the BoxQP's solution, K = - H[free, free]^-1 Qux_reg[free, :], zero rows
      // for the clamped inputs.  Warm start: k of the timestep before (i + 1), zero at the end of the horizon.
      float initial_k[MM];
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        initial_k[a] = (st.box.i != T - 1) ? st.box.k_next[a] : 0.0f;
      }
      QPOut qp;
      boxQP(QuuF, Qu, st.box.lo, st.box.up, initial_k, qp);
      if(st.box.act && st.ok && lane == 16 * qK + N)
      {
        const size_t tile = static_cast<size_t>(st.box.b) / 64, ln = static_cast<size_t>(st.box.b) % 64;
        buf.qp_ret[(tile * T + st.box.i) * 64 + ln] = qp.retval;
        buf.qp_free[(tile * T + st.box.i) * 64 + ln] = (qp.retval == 6) ? 0u : qp.free;
      }
      st.ok = st.ok && qp.retval >= 0; // :473-480
      const unsigned free = (qp.retval == 6) ? 0u : qp.free;
      float col[MM];
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        col[a] = ((free >> a) & 1u) ? in.Qr[a] : 0.0f;
      }
      ldltSolve(qp.fac, qp.inv_d, col);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const int a = rr < MM ? rr : 0;
        const float Kv = (rr < MM && ((free >> a) & 1u)) ? -1.0f * col[a] : 0.0f;
        const float gain = (j < N) ? Kv : ((j == N && rr < MM) ? qp.x[a] : 0.0f);
        const float ident = (4 * q + rr == j && j < N) ? 1.0f : 0.0f;
        o.A[rr] = (q == qK) ? gain : ident;
      }
      st.box.i -= 1;
      requestLimits(st); // (the next timestep's: in flight during stage 3 and the next stage 1)
    }
    else
    {
      // every lane of the group factorises Quu_F; lane (qK, j) solves column j of [Qux_reg | Qu]
      const bool ok_now = ldlt(QuuF, inv_d);
      st.ok = st.ok && ok_now; // (per lane, valid in lane group qK; after a failure the slot computes on garbage and stores nothing)
      float col[MM];
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        col[a] = (j == N) ? Qu[a] : in.Qr[a];
      }
      ldltSolve(QuuF, inv_d, col);
#pragma unroll
      for(int rr = 0; rr < 4; rr++)
      {
        const float gain = (rr < MM && j <= N) ? -1.0f * col[rr < MM ? rr : 0] : 0.0f;
        const float ident = (4 * q + rr == j && j < N) ? 1.0f : 0.0f;
        o.A[rr] = (q == qK) ? gain : ident;
      }
    }
    float kff[MM];
    bcastK(o.A, kff);
    if constexpr(kConstrained)
    {
#pragma unroll
      for(int a = 0; a < MM; a++)
      {
        st.box.k_next[a] = kff[a];
      }
    }
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
      st.krel = fmaxf(st.krel, knorm * in.inv_u);
    }
    // (lanes outside column n read the zeros behind the wave's transposition scratch)
    o.qcol = *reinterpret_cast<const v4f *>((j == N) ? scratch + 4 * q : lds + kTrAt + wave * kTrFloats + 16 * kTrLd);
    return o;
  }

  /** Stage 3: H = Q A + [0 | q], VV' = A^T H (8 MFMAs); Vxx <- (Vxx + Vxx^T) / 2 with the transpose taken through the wave's
      LDS scratch (four row writes, one 16-byte read per lane)    :524-527 */
  NMPC_D void stage3(BwState & st, const Stage1Out & s1, const Stage2Out & s2) const
  {
#pragma clang fp contract(on) // fuse a * b + c only within a source expression: every instance gets the same instruction
                              // sequence whichever slot of the stage pipeline it runs in (hipcc's default fuses across
                              // statements, decided per inlined context) — sharded == unsharded, bit for bit

    const int q = lane >> 4, j = lane & 15;
    const v4f zero4 = {0, 0, 0, 0};
    const v4f H = mma(s1.Q, s2.A, s2.qcol);
    const v4f Vn = mma(s2.A, H, zero4); // rows < n: [Vxx' | Vx']; row n: garbage (column n of A is [0; k]); the rest: zero
    float * tr = lds + kTrAt + wave * kTrFloats;
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      tr[(4 * q + rr) * kTrLd + j] = Vn[rr];
    }
    // lane (q, j) takes Vn[j][4 q .. 4 q + 3]; lanes outside the n x n block read zeros
    const v4f Vt = *reinterpret_cast<const v4f *>(tr + ((4 * q < N && j < N) ? j * kTrLd + 4 * q : 16 * kTrLd));
    // weights per lane: (1/2, 1/2) inside the n x n block, (1, 0) in column n (Vx), (0, 0) elsewhere — exact
    const float wn = (4 * q < N) ? ((j < N) ? 0.5f : ((j == N) ? 1.0f : 0.0f)) : 0.0f;
    const float wt = (4 * q < N && j < N) ? 0.5f : 0.0f;
#pragma unroll
    for(int rr = 0; rr < 4; rr++)
    {
      st.VV[rr] = wn * Vn[rr] + wt * Vt[rr];
    }
  }

  /** k_i, K_i -> the instance's gain record (:529-530); not after a failed factorisation: backwardPass() returned before
      storing (:505-508). */
  NMPC_D void storeGains(v4f A, int b, int i, bool ok) const
  {
    const int q = lane >> 4, j = lane & 15;
    if(q == qK && j <= N && ok)
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

  /** The sweep of the matrix waves (barriers are shared with the model wave's loop in solve()).  COUNT = instances of this
      wave, a template parameter: the stages of a timestep are branch-free and the instances are staggered by one stage —
      in every slot of the loop body one instance is in stage 1, one in stage 2, one in stage 3 (COUNT = 3), three independent
      pieces of straight-line code in one scheduling region:
          slot 0:  S1_0(i)   S3_1(i+1)  S2_2(i+1)
          slot 1:  S2_0(i)   S1_1(i)    S3_2(i+1)
          slot 2:  S3_0(i)   S2_1(i)    S1_2(i)         barrier (record i has been read by all, record i - 1 is complete)
      Only stage 1 reads the record.  Waves with two instances run the stages back to back (the compiler interleaves the two). */
  template<int kRegType, int COUNT>
  NMPC_D void backwardSweepMatrix() const
  {
    const int q = lane >> 4, j = lane & 15;
    const int first = tileWaveFirst(wave);
    BwState st[COUNT];
    int bs[COUNT];
    bool act[COUNT];
    float lam[COUNT];
    barrier(); // the terminal record and the record of timestep T - 1 are complete
#pragma unroll
    for(int e = 0; e < COUNT; e++)
    {
      const int slot = first + e;
      bs[e] = uniform(slotI(sB, slot));
      act[e] = uniform(slotI(sBw, slot)) != 0;
      lam[e] = uniformF(slotF(sLambda, slot));
      st[e].VV = (4 * q < N && j <= N) ? *reinterpret_cast<const v4f *>(term(slot) + j * N + 4 * q) : v4f{0, 0, 0, 0};
      st[e].dV0 = 0;
      st[e].dV1 = 0;
      st[e].krel = 0;
      st[e].ok = true;
      if constexpr(kConstrained)
      {
        st[e].box.i = T - 1;
        st[e].box.b = bs[e];
        st[e].box.sel = uniform(slotI(sSel, slot));
        st[e].box.act = act[e];
#pragma unroll
        for(int a = 0; a < MM; a++)
        {
          st[e].box.k_next[a] = 0;
        }
        requestLimits(st[e]);
      }
    }
    barrier(); // the terminal record has been read
    if constexpr(COUNT == 3)
    {
      Stage1Out s1[3];
      Stage2Out s2[3];
      // instance e finishes timestep i + 1 (stages 2 / 3) while the earlier instances start timestep i
      auto body = [&](int i, auto first_step)
      {
        constexpr bool kFirst = decltype(first_step)::value;
        // (the gains are stored after the three slots: the conditional stores would otherwise cut the scheduling region)
        v4f A1 = {0, 0, 0, 0}, A2 = {0, 0, 0, 0};
        bool ok1 = false, ok2 = false;
        // slot 0
        if constexpr(!kFirst)
        {
          stage3(st[1], s1[1], s2[1]);
          A1 = s2[1].A;
          ok1 = st[1].ok;
          s2[2] = stage2<kRegType>(st[2], s1[2], first + 2, lam[2]);
        }
        s1[0] = stage1<kRegType>(st[0], first + 0, i, lam[0]);
        // slot 1
        if constexpr(!kFirst)
        {
          stage3(st[2], s1[2], s2[2]);
          A2 = s2[2].A;
          ok2 = st[2].ok;
        }
        s2[0] = stage2<kRegType>(st[0], s1[0], first + 0, lam[0]);
        s1[1] = stage1<kRegType>(st[1], first + 1, i, lam[1]);
        // slot 2
        stage3(st[0], s1[0], s2[0]);
        s2[1] = stage2<kRegType>(st[1], s1[1], first + 1, lam[1]);
        s1[2] = stage1<kRegType>(st[2], first + 2, i, lam[2]);
        if constexpr(!kFirst)
        {
          if(act[1])
          {
            storeGains(A1, bs[1], i + 1, ok1);
          }
          if(act[2])
          {
            storeGains(A2, bs[2], i + 1, ok2);
          }
        }
        if(act[0])
        {
          storeGains(s2[0].A, bs[0], i, st[0].ok);
        }
      };
      body(T - 1, std::true_type());
      barrier();
      for(int i = T - 2; i >= 0; i--)
      {
#ifdef NMPC_AMD_PROFILE_TILE32
        const unsigned long long p0 = __builtin_readcyclecounter();
#endif
        body(i, std::false_type());
#ifdef NMPC_AMD_PROFILE_TILE32
        if(blockIdx.x == 0 && lane == 0)
        {
          buf.qp_free[static_cast<size_t>(8 + wave) * 64] += static_cast<unsigned>((__builtin_readcyclecounter() - p0) >> 4);
        }
#endif
        barrier();
      }
      // drain: instances 1 and 2 finish timestep 0
      stage3(st[1], s1[1], s2[1]);
      if(act[1])
      {
        storeGains(s2[1].A, bs[1], 0, st[1].ok);
      }
      s2[2] = stage2<kRegType>(st[2], s1[2], first + 2, lam[2]);
      stage3(st[2], s1[2], s2[2]);
      if(act[2])
      {
        storeGains(s2[2].A, bs[2], 0, st[2].ok);
      }
    }
    else
    {
      for(int i = T - 1; i >= 0; i--)
      {
#ifdef NMPC_AMD_PROFILE_TILE32
        const unsigned long long p0 = __builtin_readcyclecounter();
#endif
        Stage1Out s1[COUNT];
        Stage2Out s2[COUNT];
#pragma unroll
        for(int e = 0; e < COUNT; e++)
        {
          s1[e] = stage1<kRegType>(st[e], first + e, i, lam[e]);
        }
#pragma unroll
        for(int e = 0; e < COUNT; e++)
        {
          s2[e] = stage2<kRegType>(st[e], s1[e], first + e, lam[e]);
        }
#pragma unroll
        for(int e = 0; e < COUNT; e++)
        {
          stage3(st[e], s1[e], s2[e]);
        }
#pragma unroll
        for(int e = 0; e < COUNT; e++)
        {
          if(act[e]) // wave-uniform
          {
            storeGains(s2[e].A, bs[e], i, st[e].ok);
          }
        }
#ifdef NMPC_AMD_PROFILE_TILE32
        if(blockIdx.x == 0 && lane == 0)
        {
          buf.qp_free[static_cast<size_t>(8 + wave) * 64] += static_cast<unsigned>((__builtin_readcyclecounter() - p0) >> 4);
        }
#endif
        barrier();
      }
    }
    if(lane == 16 * qK) // (dV, |k| / (|u| + 1) and the pivot tests are per-lane values of lane group qK)
    {
#pragma unroll
      for(int e = 0; e < COUNT; e++)
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
  template<int kRegType>
  NMPC_D void backwardSweepMatrixWave() const
  {
    if(tileWaveCount(wave) == kTileMinPerWave)
    {
      backwardSweepMatrix<kRegType, kTileMinPerWave>();
    }
    else
    {
      backwardSweepMatrix<kRegType, kTileMaxPerWave>();
    }
  }

  /** The model wave's half of the sweep: timestep i - 1 is linearised while the matrix waves consume timestep i; its
      (x, u) were requested one timestep earlier. */
  NMPC_D void backwardSweepModel(bool mine, int slot, int b, int sel, float t0) const
  {
    Point cur, next;
    if(mine)
    {
      loadPoint(cur, b, sel, T - 1);
      loadPoint(next, b, sel, T > 1 ? T - 2 : 0);
      lineariseTerminal(slot, b, sel, t0);
      lineariseStep<true>(slot, t0, T - 1, cur);
    }
    barrier(); // the terminal record and the record of timestep T - 1 are complete
    barrier(); // (the matrix waves have taken the terminal record)
    for(int i = T - 1; i >= 0; i--)
    {
#ifdef NMPC_AMD_PROFILE_TILE32
      const unsigned long long p0 = __builtin_readcyclecounter();
#endif
      if(mine && i > 0)
      {
        cur = next;
        loadPoint(next, b, sel, i > 1 ? i - 2 : 0);
        if(i == T - 1)
        {
          lineariseStep<true>(slot, t0, i - 1, cur); // the first use of this record slot in the sweep: every entry
        }
        else
        {
          lineariseStep<false>(slot, t0, i - 1, cur);
        }
      }
#ifdef NMPC_AMD_PROFILE_TILE32
      prof_acc[7] += __builtin_readcyclecounter() - p0; // the model wave's own work inside the sweep
#endif
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

#ifdef NMPC_AMD_PROFILE_TILE32
  // profiling build (scripts/profile_tile32.py): shader-clock ticks per phase seen by the model wave of workgroup 0,
  // returned through qp_free (unused by this kernel family): 0 initial rollout, 1 backward sweeps, 2 line search, 3 re-rolls,
  // 4 sweeps run, 5 re-rolls run, 6 iteration rounds
  mutable unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  mutable unsigned long long prof_t0 = 0;
  NMPC_D void profBegin() const
  {
    prof_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int k) const
  {
    prof_acc[k] += __builtin_readcyclecounter() - prof_t0;
  }
  NMPC_D void profCount(int k) const
  {
    prof_acc[k] += 1;
  }
  NMPC_D void profFlush() const
  {
    if(blockIdx.x == 0 && wave == kTileModelWave && lane == 0)
    {
      for(int k = 0; k < 8; k++)
      {
        buf.qp_free[static_cast<size_t>(k) * 64] = static_cast<unsigned>((k < 4 || k == 7) ? prof_acc[k] >> 4 : prof_acc[k]);
      }
    }
  }
#else
  // product builds: ticks of the sweeps / rollouts seen by the model wave, per instance into DeviceBuffers::phase_ticks
  mutable unsigned long long prof_acc[4] = {0, 0, 0, 0};
  mutable unsigned long long prof_t0 = 0;
  NMPC_D void profBegin() const
  {
    prof_t0 = __builtin_readcyclecounter();
  }
  NMPC_D void profEnd(int k) const
  {
    prof_acc[k] += __builtin_readcyclecounter() - prof_t0;
  }
  NMPC_D void profCount(int) const {}
  NMPC_D void profFlush() const {}
#endif

  NMPC_D void solve()
  {
    const unsigned long long solve_start = __builtin_readcyclecounter();
    (void)solve_start;
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
    // ---- zero areas, slot table
    for(int e = static_cast<int>(threadIdx.x); e < kTileWaves * 16; e += kTileThreads)
    {
      lds[kTrAt + (e >> 4) * kTrFloats + 16 * kTrLd + (e & 15)] = 0.0f;
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
      profBegin();
      J_cur = rollout(owner, b, 0, 0, t0, 0.0f, true, true);
      profEnd(0);
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
      phaseBarrier();
      if(uniform(flag(0)) == 0)
      {
        break;
      }
      // ---- Steps 1 + 2: linearisation fused into the backward sweep, with the regularisation retries    :157-214
      bool need_bw = in_iter;
      profCount(6);
      for(;;)
      {
        profBegin();
        profCount(4);
        if(model_wave)
        {
          backwardSweepModel(need_bw && lane < kTileInstances, slot, b, sel, t0);
        }
        else if(cfg.reg_type == 2)
        {
          backwardSweepMatrixWave<2>();
        }
        else if(cfg.reg_type == 1)
        {
          backwardSweepMatrixWave<1>();
        }
        else
        {
          backwardSweepMatrixWave<0>();
        }
        phaseBarrier(); // results of the sweep are in the slot table
        profEnd(1);
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
        phaseBarrier();
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
      phaseBarrier();
      if(uniform(flag(2)) != 0)
      {
        // Every step size of alpha_list at once.  The model wave rolls out the first one, lane = instance, and stores it (it is
        // the one normally taken).  On the other waves a lane is an (instance, step size) pair, cost only, ALL the later step
        // sizes of an instance in the same wave: its nominal record (x, u, k, K) is then fetched once per wave — the lanes of
        // an instance read the same addresses — instead of once per step size (measured: 2.4 -> 1.7 GB of reads per launch).
        // Loads and stores stay in different waves: a wave that does both waits for its stores whenever it waits for a load
        // (one counter for both), which doubled the time of this phase when the first step size's lanes sat among the others.
        profBegin();
        if(model_wave)
        {
          const float Jc = rollout(in_ls, b, sel, sel ^ 1, t0, static_cast<float>(cfg.alpha_list[0]), false, true);
          if(in_ls)
          {
            lds[kLsAt + slot] = Jc;
          }
        }
        else if(cfg.n_alpha > 1)
        {
          const int n_later = cfg.n_alpha - 1;
          const int per_wave = 64 / n_later; // >= 2 (NMPC_HIP_MAX_ALPHA = 32)
          const int inst = lane / n_later, ai = 1 + lane - inst * n_later;
          for(int base = 0; base < kTileInstances; base += per_wave * kTileMatrixWaves)
          {
            const int fslot = base + (wave - 1) * per_wave + inst;
            const bool in_range = inst < per_wave && fslot < kTileInstances;
            const int fs = in_range ? fslot : 0;
            const bool act = in_range && slotI(sB, fs) >= 0 && slotI(sLs, fs) != 0;
            const int fb = slotI(sB, fs), fsel = slotI(sSel, fs);
            const float ft0 = slotF(sT0, fs);
            const float Jc = rollout(act, fb, fsel, fsel ^ 1, ft0, static_cast<float>(cfg.alpha_list[act ? ai : 0]), false, false);
            if(act)
            {
              lds[kLsAt + ai * kTileInstances + fs] = Jc;
            }
          }
        }
        phaseBarrier();
        profEnd(2);
        if(model_wave)
        {
          int ai_taken = cfg.n_alpha - 1;
          bool success = false;
          float alpha = 0, actual = 0, expected = 0, ratio = 0, J_cand = 0;
          if(in_ls)
          {
            for(int ai = 0; ai < cfg.n_alpha; ai++)
            {
              const float Jc = lds[kLsAt + ai * kTileInstances + slot];
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
            profBegin();
            profCount(5);
            const float Jr = rollout(reroll, b, sel, sel ^ 1, t0, alpha, false, true);
            profEnd(3);
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

    profFlush();
    // ---- results the host reads per instance
    if(owner)
    {
      const size_t tile = static_cast<size_t>(b) / 64, ln = static_cast<size_t>(b) % 64;
      buf.status[b] = retval;
      buf.iters[b] = iter;
      buf.sel[b] = sel;
#ifndef NMPC_AMD_PROFILE_TILE32
      if(buf.phase_ticks != nullptr)
      {
        unsigned long long * p = buf.phase_ticks + static_cast<size_t>(b) * 4;
        p[0] = prof_acc[1];
        p[1] = prof_acc[0] + prof_acc[2] + prof_acc[3];
        p[2] = __builtin_readcyclecounter() - solve_start;
      }
#endif
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

/** The fp32 tile kernel: grid = ceil(B / 32) workgroups of sixteen wavefronts. */
template<class Problem, bool kOwnProblem, bool kConstrained>
__global__ __launch_bounds__(kTileThreads) void ddp_solve_tile32_kernel(const Problem problem,
                                                                        const nmpc_hip_ddp_config cfg,
                                                                        const DeviceBuffersT<float> buf)
{
  extern __shared__ __attribute__((aligned(16))) float lds_tile32[];
  TileSolver32<Problem, kOwnProblem, kConstrained> solver(problem, cfg, buf, lds_tile32);
  solver.solve();
}

/** Type-erased operations (model_ops.hpp) of an fp32 problem type served by the tile kernel. */
template<class Problem>
struct ModelOpsTile32
{
  using Solver = TileSolver32<Problem, false, false>;
  static void defaultParams(void * out)
  {
    new(out) Problem();
  }
  /** Round 4: the fp64 tile kernel has a float instantiation (ddp_kernels_tile64.hpp: the same kernel with v_mfma_f32_16x16x4 and
      the lanes of a row holding the tile's columns in the order that makes the f32 instruction's result layout the f64 one).  It
      is the fp32 kernel of the shapes THIS file's kernel does not take (m > 4, n not in {4, 8, 12}: ModelOpsTile64Float below).
      On the shapes both take (unconstrained solves of 5 <= n <= 12; BoxQP in float is this file's only) the choice is per launch,
      measured on the quadrotor (profiles/r04_c4_dispatch_sweep.txt, scripts/c4_dispatch_sweep*.py):
        * this file's kernel is the leaner one per full sweep (16 MFMAs + 125 other instructions a step against 10 + 250), but a
          workgroup is 32 instances whatever the batch, its model wave linearises all 32 lanes of every timestep and its matrix
          waves step all their slots — a sweep costs the same however few instances still iterate, and batches below 8192 leave
          CUs idle (64 .. 4096 instances: 0.77 - 0.86 ms per 2 iterations);
        * the other kernel sizes its groups to the batch and deals a sweep's work by ACTIVE index: 1.3 - 2.6 x faster up to 4096
          instances at any iteration count and threshold; on full chips (8192, 16384 instances) level for one to four iterations
          and ahead from there (c4, max_iter 8: 1.95 k against 1.46 k it/s — the iteration counts of a batch are ragged, the late
          sweeps nearly empty) AS LONG AS most line searches end at the first or second step size (its search is passes over the
          horizon: the first two step sizes in one, the later ones in a second, the taken one in a third; this file's rolls every
          step size out at once).  They do not once an fp32 solve iterates below the resolution of a float cost: 0.7 - 0.9 x at
          cost_update_thre = 1e-4 and below (the reference's default 1e-7: 0.6 x), level at 3e-4, 1.0 - 1.3 x at 1e-3.
      Hence: the float instantiation below 8192 instances; on full chips with cost_update_thre >= 5e-4.
      NMPC_HIP_DDP_KERNEL=tile32 / tile64 forces one of them (A/B measurements; tests/test_gpu_fp32.py runs on both). */
  static constexpr bool kTile64Float = Problem::kStateDim >= 5 && Problem::kStateDim <= 15 && Problem::kInputDimMax >= 1
                                       && Problem::kInputDimMax <= 8 && !Problem::kDynamicInput;
  //! batches that fill the chip with this kernel's fixed 32-instance workgroups: 32 x the number of CUs (8192 on MI355X)
  static int fullChipBatch()
  {
    static int n_cu = 0; // of the current device at first use (the handles of one process sit on like devices)
    if(n_cu == 0)
    {
      int device = 0;
      if(hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess
         || n_cu <= 0)
      {
        n_cu = 256;
      }
    }
    return 32 * n_cu;
  }
  static constexpr double kTile64FloatFromThreshold = 5e-4;
  static bool useTile64Float(int batch, const nmpc_hip_ddp_config & cfg)
  {
    if(!kTile64Float || cfg.with_input_constraint != 0)
    {
      return false;
    }
    const LaunchKnobs knobs = launchKnobs();
    if(knobs.kernelIs("tile64"))
    {
      return true;
    }
    if(knobs.kernelIs("tile32"))
    {
      return false;
    }
    // (a shard of a larger solve takes the family the WHOLE batch would get: the two fp32 kernels differ in the last bits)
    return knobs.batchFor(batch) < fullChipBatch() || cfg.cost_update_thre >= kTile64FloatFromThreshold;
  }
  static const char * kernelName(int batch, const nmpc_hip_ddp_config & cfg)
  {
    return useTile64Float(batch, cfg) ? "ddp_solve_tile64_kernel" : "ddp_solve_tile32_kernel";
  }
  /** The handle allocates every Scalar array with sizeof(Problem::Scalar) = 4 (ModelOps::scalar_bytes): same pointers, float view. */
  static DeviceBuffersT<float> floatView(const DeviceBuffers & buf64)
  {
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
    buf.phase_ticks = buf64.phase_ticks;
    buf.params_batch = buf64.params_batch;
    buf.lim_batch = buf64.lim_batch; // (the limits are read by the receding-horizon driver's clamp: doubles in every handle)
    buf.lim_steps = buf64.lim_steps;
    buf.lim_steps_per_instance = buf64.lim_steps_per_instance;
    buf.lim_mm = buf64.lim_mm;
    buf.lim_rows = buf64.lim_rows;
    buf.lim_offset = buf64.lim_offset;
    for(int i = 0; i < kMaxInputDim; i++)
    {
      buf.lim_lo[i] = buf64.lim_lo[i];
      buf.lim_hi[i] = buf64.lim_hi[i];
    }
    return buf;
  }
  static hipError_t launchSolve(const void * params, const nmpc_hip_ddp_config & cfg, const DeviceBuffers & buf64,
                                hipStream_t stream)
  {
    if(buf64.wpi_ws == nullptr)
    {
      return hipErrorNotSupported;
    }
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    const DeviceBuffersT<float> buf = floatView(buf64);
    if constexpr(kTile64Float)
    {
      if(useTile64Float(buf64.B, cfg))
      {
        if(buf.params_batch != nullptr)
        {
          return launchTile64<Problem, false, true>(problem, cfg, buf, stream);
        }
        return launchTile64<Problem, false, false>(problem, cfg, buf, stream);
      }
    }
    constexpr size_t lds_bytes = Solver::kLdsBytes;
    static std::atomic<bool> requested[64] = {}; // (several host threads may launch at once; the setup is idempotent)
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
    {
      return hipErrorInvalidDevice;
    }
    if(!requested[dev].load(std::memory_order_acquire))
    {
      for(const void * kernel : {reinterpret_cast<const void *>(&ddp_solve_tile32_kernel<Problem, false, false>),
                                 reinterpret_cast<const void *>(&ddp_solve_tile32_kernel<Problem, true, false>),
                                 reinterpret_cast<const void *>(&ddp_solve_tile32_kernel<Problem, false, true>),
                                 reinterpret_cast<const void *>(&ddp_solve_tile32_kernel<Problem, true, true>)})
      {
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
        if(e != hipSuccess)
        {
          return e;
        }
      }
      requested[dev].store(true, std::memory_order_release);
    }
    const dim3 g((buf.B + kTileInstances - 1) / kTileInstances), blk(kTileThreads);
    const bool own = buf.params_batch != nullptr, box = cfg.with_input_constraint != 0;
    if(own && box)
    {
      hipLaunchKernelGGL((ddp_solve_tile32_kernel<Problem, true, true>), g, blk, lds_bytes, stream, problem, cfg, buf);
    }
    else if(own)
    {
      hipLaunchKernelGGL((ddp_solve_tile32_kernel<Problem, true, false>), g, blk, lds_bytes, stream, problem, cfg, buf);
    }
    else if(box)
    {
      hipLaunchKernelGGL((ddp_solve_tile32_kernel<Problem, false, true>), g, blk, lds_bytes, stream, problem, cfg, buf);
    }
    else
    {
      hipLaunchKernelGGL((ddp_solve_tile32_kernel<Problem, false, false>), g, blk, lds_bytes, stream, problem, cfg, buf);
    }
    return hipGetLastError();
  }
  /** The receding-horizon driver's advance step (mpc_kernels.hpp) on the handle's float arrays. */
  static hipError_t launchMpcAdvance(const void * params, const DeviceBuffers & buf64, const MpcAdvanceArgs & args, hipStream_t stream)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    const DeviceBuffersT<float> buf = floatView(buf64);
    hipLaunchKernelGGL((mpc_advance_kernel<Problem, float>), dim3(buf.Bp / kLanesPerBlock), dim3(kLanesPerBlock), 0, stream, problem,
                       buf, args);
    return hipGetLastError();
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
    size_t n = Solver::workspaceElems(T);
    if constexpr(kTile64Float)
    {
      const size_t n64 = TileSolver64<Problem>::workspaceDoubles(T); // (elements of float: gain records + candidate trajectories)
      n = n64 > n ? n64 : n;
    }
    return n;
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
    ops.has_plant_step = HasPlantStep<Problem>::value ? 1 : 0;
    ops.wpi_workspace_doubles = &workspaceElems;
    ops.scalar_bytes = 4;
    ops.gain_layout = 1;
    ops.own_problems_supported = [](int, int) { return 1; };
    return ops;
  }
};
/** Type-erased operations of an fp32 problem type served by the FP64 TILE KERNEL'S FLOAT INSTANTIATION only: the shapes the
    fp32 tile kernel above does not take (5 <= n <= 15, m <= 8; e.g. the manipulator, n 14, m 7).  Unconstrained solves; a
    box-constrained solve is refused at launch (the handle reports the HIP error). */
template<class Problem>
struct ModelOpsTile64Float
{
  static_assert(std::is_same<typename Problem::Scalar, float>::value, "float problem types");
  static void defaultParams(void * out)
  {
    new(out) Problem();
  }
  static const char * kernelName(int, const nmpc_hip_ddp_config &)
  {
    return "ddp_solve_tile64_kernel";
  }
  static hipError_t launchSolve(const void * params, const nmpc_hip_ddp_config & cfg, const DeviceBuffers & buf64, hipStream_t stream)
  {
    if(buf64.wpi_ws == nullptr || cfg.with_input_constraint != 0)
    {
      return hipErrorNotSupported;
    }
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    const DeviceBuffersT<float> buf = ModelOpsTile32<Problem>::floatView(buf64);
    if(buf.params_batch != nullptr)
    {
      return launchTile64<Problem, false, true>(problem, cfg, buf, stream);
    }
    return launchTile64<Problem, false, false>(problem, cfg, buf, stream);
  }
  static hipError_t launchMpcAdvance(const void * params, const DeviceBuffers & buf64, const MpcAdvanceArgs & args, hipStream_t stream)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    const DeviceBuffersT<float> buf = ModelOpsTile32<Problem>::floatView(buf64);
    hipLaunchKernelGGL((mpc_advance_kernel<Problem, float>), dim3(buf.Bp / kLanesPerBlock), dim3(kLanesPerBlock), 0, stream, problem,
                       buf, args);
    return hipGetLastError();
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
    return TileSolver64<Problem>::workspaceDoubles(T);
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
    ops.has_plant_step = HasPlantStep<Problem>::value ? 1 : 0;
    ops.wpi_workspace_doubles = &workspaceElems;
    ops.scalar_bytes = 4;
    ops.gain_layout = 1;
    ops.own_problems_supported = [](int, int) { return 1; };
    return ops;
  }
};
} // namespace hip
} // namespace nmpc_amd

#define NMPC_AMD_REGISTER_PROBLEM_TILE64_FLOAT(ProblemType) \
  NMPC_AMD_REGISTER_PROBLEM_WITH(ProblemType, nmpc_amd::hip::ModelOpsTile64Float<ProblemType>)

#define NMPC_AMD_REGISTER_PROBLEM_TILE32(ProblemType) \
  NMPC_AMD_REGISTER_PROBLEM_WITH(ProblemType, nmpc_amd::hip::ModelOpsTile32<ProblemType>)
