// gfx950 kernels of the batched FMPC solver: the reference's nmpc_fmpc::FmpcSolver
// (nmpc_fmpc/include/nmpc_fmpc/FmpcSolver.hpp) for B independent instances at once.
//
// FMPC is a multiple-shooting method: every quantity of one iteration except the Riccati recursion is independent across the
// T timesteps of the horizon.  The iteration is therefore split where the reference's procOnce (FmpcSolver.hpp:356-491) has
// its steps, and each step is launched with the parallelism it has:
//
//   fmpc_barrier_kernel      (B x slices)   barrier parameter from mean(s . nu)                       :370-392
//   fmpc_coeff_kernel        (B x (T+1))    linearisation, offsets x_bar / g_bar / Lx_bar / Lu_bar,
//                                           the barrier-condensed Q~, L~ of the backward pass, per-step
//                                           KKT-error terms                                            :394-441, :562-574, :493-520
//   fmpc_riccati_kernel      (B)            KKT-error test, backward Riccati recursion, forward sweep  :443-449, :522-665, :667-687
//   fmpc_delta_kernel        (B x T)        ds, dnu, per-step fraction-to-boundary step lengths        :689-697, :713-731
//   fmpc_step_length_kernel  (B x slices)   min over the horizon, validity test                       :732-742
//   fmpc_line_search_kernel  (B, optional)  l1 merit function backtracking                             :748-792, :840-981
//   fmpc_update_kernel       (B x (T+1))    variables += step                                          :801-835
//
// Every array is laid out [timestep][element][instance] ("instance-minor"): consecutive lanes of a wavefront hold consecutive
// instances, so every global access of every kernel above is a fully coalesced 512-byte row, whether the kernel's threads
// enumerate (instance, timestep) pairs or instances only.  Per-instance state that decides control flow (status, iteration
// count, barrier parameter, step lengths) lives in [instance] arrays; an instance that has terminated (Succeeded or an error
// status) is skipped by every later kernel, the rest of the batch continues (masking, as in the DDP kernels).
// Reductions over the horizon are done in a fixed order (per-slice ascending, then slices ascending): results do not depend
// on scheduling, two runs of the same solve are bit-identical.
#pragma once

#include <cfloat>
#include <cmath>
#include <type_traits>

#include <hip/hip_runtime.h>

#include <nmpc_amd/FmpcProblem.hpp>
#include <nmpc_hip_fmpc.h>
#include <nmpc_amd/hip/fuzz_sched.hpp>

namespace nmpc_amd
{
namespace hip
{
/** Device buffers and the scalar configuration of one handle; passed to every kernel by value. */
struct FmpcBuffers
{
  int B = 0; // instances
  int T = 0; // horizon_steps
  int N = 0, M = 0, G = 0;
  int riccati_force = 0; // host side: which Riccati kernel the handle launches (0 automatic, 1 matrix-core, 2 lane; fmpc_ops.hpp)
  int fuse_tail = 1; // host side: fmpc_tail_kernel closes an iteration of the fused-Riccati sequence (0: the separate kernels; NMPC_HIP_FMPC_TAIL)
  int coef_stride = 0; // doubles per timestep in `coef`
  int gain_stride = 0; // doubles per timestep in `gain`
  // Variable (FmpcSolver.h:117-158): x [T+1][N][B], u [T][M][B], lambda [T+1][N][B], s [T][G][B], nu [T][G][B]
  double * x = nullptr;
  double * u = nullptr;
  double * lam = nullptr;
  double * s = nullptr;
  double * nu = nullptr;
  // delta_variable_ (FmpcSolver.h:402), same shapes
  double * dx = nullptr;
  double * du = nullptr;
  double * dlam = nullptr;
  double * ds = nullptr;
  double * dnu = nullptr;
  // what the Riccati recursion reads per timestep: A, B, x_bar, Qxx~, Quu~, Qxu~, Lx~, Lu~  [T][coef_stride][B]
  double * coef = nullptr;
  // what it writes: k, K, s, P  [T+1][gain_stride][B] (Coefficient::k / K / s / P, FmpcSolver.h:214-224; the terminal entry
  // holds s and P only)
  double * gain = nullptr;
  // per-timestep partial results [T+1][kPartSlots][B]: KKT-error terms, alpha_s candidate, alpha_nu candidate, s . nu
  double * part = nullptr;
  const double * t0 = nullptr; // [B] current_t
  const double * x0 = nullptr; // [N][B] current_x
  double * barrier_eps = nullptr; // [B] barrier_eps_ (FmpcSolver.h:414), kept across solves like the member it mirrors
  double * alpha = nullptr; // [3][B]: alpha_s_max, alpha_nu_max, alpha_s of the current iteration
  double * merit = nullptr; // [3][B]: merit_func_, merit_deriv_, merit_const_scale_ of the last line search
  int * status = nullptr; // [B] FmpcSolver::Status, or NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE
  int * iters = nullptr; // [B] traceDataList().back().iter
  int * flags = nullptr; // [B] bit 0: a coefficient is NaN / Inf, bit 1: delta_variable_ contains NaN / Inf
  double * trace = nullptr; // [B][max_iter][NMPC_HIP_FMPC_NTRACE]
  const void * problems = nullptr; // one problem object, or B of them
  int own_problems = 0;
  // Configuration (FmpcSolver.h:57-89)
  int max_iter = 0;
  double kkt_error_thre = 0;
  int check_nan = 1;
  int update_barrier_eps = 1;
  int break_if_llt_fails = 0;
  int enable_line_search = 0;
  int merit_const_scale_from_lagrange_multipliers = 0;
};

namespace fmpc
{
constexpr int kStatusContinued = 6; // Status::IterationContinued (FmpcSolver.h:113)
constexpr int kSlices = 8; // horizon slices of the per-instance reductions
constexpr int kPartSlots = 4; // per-timestep partials: 0 KKT-error terms, 1 / 2 step-length candidates, 3 s . nu of the timestep
#ifndef NMPC_AMD_FMPC_STAGE_STEPS
#  define NMPC_AMD_FMPC_STAGE_STEPS 4
#endif
constexpr int kStageSteps = NMPC_AMD_FMPC_STAGE_STEPS; // timesteps per cooperative fetch of fmpc_riccati_quad_kernel
// Register sets of the two recursions of fmpc_riccati_kernel: a step's record is requested (depth - 1) steps before its use.
// What was measured on MI355X around this choice (4096 x 200 cart-pole, backward recursion alone 297 us at depth 2): deeper
// rings 322 (3) / 583 us (4); all records of 2-4 steps requested at the top of a loop trip instead of a ring: 10 % slower;
// stores parked in LDS and written in bursts (steady state load-only): 428 us; records as adjacent element pairs (16-byte
// accesses, half the memory instructions): +5 %.  The recursion is not waiting for memory it could have asked for earlier: per
// step it issues ~500 instructions on one wavefront per SIMD (the VALU is busy 40 % of the kernel's time, profiles/
// r02_pmc_summary_fmpc.txt), and part of the requested data reaches its loop-carried registers through compiler-inserted copies
// that wait for the loads.  Fewer instructions per lane (several lanes per instance) is the lever, not prefetch depth.
#ifndef NMPC_AMD_FMPC_BACKWARD_DEPTH
#  define NMPC_AMD_FMPC_BACKWARD_DEPTH 2
#endif
#ifndef NMPC_AMD_FMPC_FORWARD_DEPTH
#  define NMPC_AMD_FMPC_FORWARD_DEPTH 2
#endif
constexpr int kBackwardDepth = NMPC_AMD_FMPC_BACKWARD_DEPTH;
constexpr int kForwardDepth = NMPC_AMD_FMPC_FORWARD_DEPTH;

__device__ __forceinline__ size_t at(const FmpcBuffers & buf, int i, int e, int stride, int b)
{
  return (static_cast<size_t>(i) * stride + e) * buf.B + b;
}

__device__ __forceinline__ bool bad(double v)
{
  return !(fabs(v) <= DBL_MAX); // NaN or Inf (CHECK_NAN, FmpcSolver.hpp:10-18)
}

/** bad(v_1) || bad(v_2) || ... in one instruction per value: v x 0 is NaN exactly when v is NaN or Inf and (+-)0 otherwise, a sum of
    such terms is NaN exactly when one of them is.  (Coefficient::containsNaN, FmpcSolver.hpp:136-154, checks ~75 values per
    timestep; as compare + scalar AND pairs they were a tenth of a producer wave's record in fmpc_riccati_fused_kernel [measured:
    224 -> 233 us per launch when the producers took the check over].)  Four sums, so that consecutive terms do not wait for each other. */
struct NanAccumulator
{
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int n = 0; // (a compile-time constant wherever the calls sit in straight-line code)
  __device__ __forceinline__ void add(double v)
  {
    acc[n & 3] = __builtin_fma(v, 0.0, acc[n & 3]);
    n++;
  }
  __device__ __forceinline__ bool any() const
  {
    return bad((acc[0] + acc[1]) + (acc[2] + acc[3]));
  }
};

/** Offsets of the per-timestep coefficient record. */
template<int N, int M>
struct CoefLayout
{
  static constexpr int A = 0;
  static constexpr int B = A + N * N;
  static constexpr int XBAR = B + N * M;
  static constexpr int QXX = XBAR + N;
  static constexpr int QUU = QXX + N * N;
  static constexpr int QXU = QUU + M * M;
  static constexpr int LXT = QXU + N * M;
  static constexpr int LUT = LXT + N;
  static constexpr int kStride = LUT + M;
};

/** Offsets of the per-timestep gain record. */
template<int N, int M>
struct GainLayout
{
  static constexpr int k = 0;
  static constexpr int K = k + M;
  static constexpr int S = K + M * N;
  static constexpr int P = S + N;
  static constexpr int kStride = P + N * N;
};

/** Eigen::LDLT<Matrix, Lower> of an M x M matrix with symmetric diagonal pivoting and the pseudo-inverse of D in the solve
    (what FmpcSolver.hpp:581-586 runs on G; Eigen 3.4 Cholesky/LDLT.h).  compute() returns false where info() would be
    NumericalIssue. */
template<int M>
struct Ldlt
{
  double a[M * M > 0 ? M * M : 1];
  int tr[M > 0 ? M : 1];

  __device__ bool compute(const double * Gm)
  {
    NMPC_UNROLL
    for(int i = 0; i < M * M; i++)
    {
      a[i] = Gm[i];
    }
    if constexpr(M <= 1)
    {
      if constexpr(M == 1)
      {
        tr[0] = 0;
      }
      return true;
    }
    else
    {
      bool found_zero_pivot = false;
      bool ret = true;
      for(int k = 0; k < M; k++)
      {
        int p = k;
        double big = fabs(a[k + k * M]);
        for(int i = k + 1; i < M; i++)
        {
          if(fabs(a[i + i * M]) > big)
          {
            big = fabs(a[i + i * M]);
            p = i;
          }
        }
        tr[k] = p;
        if(p != k)
        {
          for(int j = 0; j < k; j++)
          {
            swap(a[k + j * M], a[p + j * M]);
          }
          for(int i = p + 1; i < M; i++)
          {
            swap(a[i + k * M], a[i + p * M]);
          }
          swap(a[k + k * M], a[p + p * M]);
          for(int i = k + 1; i < p; i++)
          {
            swap(a[i + k * M], a[p + i * M]);
          }
        }
        if(k > 0)
        {
          double temp[M];
          double acc = 0;
          for(int j = 0; j < k; j++)
          {
            temp[j] = a[j + j * M] * a[k + j * M];
            acc += a[k + j * M] * temp[j];
          }
          a[k + k * M] -= acc;
          for(int i = k + 1; i < M; i++)
          {
            double sum = 0;
            for(int j = 0; j < k; j++)
            {
              sum += a[i + j * M] * temp[j];
            }
            a[i + k * M] -= sum;
          }
        }
        const double akk = a[k + k * M];
        const bool pivot_is_valid = fabs(akk) > 0.0;
        if(k == 0 && !pivot_is_valid)
        {
          for(int j = 0; j < M; j++)
          {
            tr[j] = j;
            for(int i = j + 1; i < M; i++)
            {
              ret = ret && (a[i + j * M] == 0.0);
            }
          }
          return ret;
        }
        if(pivot_is_valid)
        {
          for(int i = k + 1; i < M; i++)
          {
            a[i + k * M] /= akk;
          }
        }
        else
        {
          for(int i = k + 1; i < M; i++)
          {
            ret = ret && (a[i + k * M] == 0.0);
          }
        }
        if(found_zero_pivot && pivot_is_valid)
        {
          ret = false;
        }
        else if(!pivot_is_valid)
        {
          found_zero_pivot = true;
        }
      }
      return ret;
    }
  }

  __device__ static void swap(double & p, double & q)
  {
    const double t = p;
    p = q;
    q = t;
  }

  /** x = G^-1 b, in place, for one right-hand side. */
  __device__ void solveInPlace(double * x) const
  {
    if constexpr(M == 1)
    {
      x[0] = fabs(a[0]) > DBL_MIN ? x[0] / a[0] : 0.0;
    }
    else if constexpr(M > 1)
    {
      for(int k = 0; k < M; k++)
      {
        swap(x[k], x[tr[k]]);
      }
      for(int i = 0; i < M; i++)
      {
        for(int j = 0; j < i; j++)
        {
          x[i] -= a[i + j * M] * x[j];
        }
      }
      for(int i = 0; i < M; i++)
      {
        x[i] = fabs(a[i + i * M]) > DBL_MIN ? x[i] / a[i + i * M] : 0.0;
      }
      for(int i = M - 1; i >= 0; i--)
      {
        for(int j = i + 1; j < M; j++)
        {
          x[i] -= a[j + i * M] * x[j];
        }
      }
      for(int k = M - 1; k >= 0; k--)
      {
        swap(x[k], x[tr[k]]);
      }
    }
  }
};

/** Full-pivot Gaussian elimination for one right-hand side: the role of Eigen::FullPivLU at FmpcSolver.hpp:599-601, reached
    only when the LDLT reports NumericalIssue. */
template<int M>
__device__ void fullPivLuSolveInPlace(const double * Gm, double * b)
{
  if constexpr(M > 0)
  {
    double a[M * M];
    int colperm[M];
    for(int i = 0; i < M * M; i++)
    {
      a[i] = Gm[i];
    }
    for(int i = 0; i < M; i++)
    {
      colperm[i] = i;
    }
    int rank = 0;
    for(int k = 0; k < M; k++)
    {
      int pr = k, pc = k;
      double big = 0;
      for(int j = k; j < M; j++)
      {
        for(int i = k; i < M; i++)
        {
          if(fabs(a[i + j * M]) > big)
          {
            big = fabs(a[i + j * M]);
            pr = i;
            pc = j;
          }
        }
      }
      if(big == 0.0)
      {
        break;
      }
      rank++;
      for(int j = 0; j < M; j++)
      {
        Ldlt<M>::swap(a[k + j * M], a[pr + j * M]);
      }
      Ldlt<M>::swap(b[k], b[pr]);
      for(int i = 0; i < M; i++)
      {
        Ldlt<M>::swap(a[i + k * M], a[i + pc * M]);
      }
      const int tmp = colperm[k];
      colperm[k] = colperm[pc];
      colperm[pc] = tmp;
      for(int i = k + 1; i < M; i++)
      {
        const double f = a[i + k * M] / a[k + k * M];
        for(int j = k + 1; j < M; j++)
        {
          a[i + j * M] -= f * a[k + j * M];
        }
        b[i] -= f * b[k];
      }
    }
    double y[M];
    for(int i = M - 1; i >= 0; i--)
    {
      if(i >= rank)
      {
        y[i] = 0;
        continue;
      }
      double sum = b[i];
      for(int j = i + 1; j < rank; j++)
      {
        sum -= a[i + j * M] * y[j];
      }
      y[i] = sum / a[i + i * M];
    }
    for(int i = 0; i < M; i++)
    {
      b[colperm[i]] = y[i];
    }
  }
}

template<class Problem>
__device__ __forceinline__ Problem loadProblem(const FmpcBuffers & buf, int b)
{
  return static_cast<const Problem *>(buf.problems)[buf.own_problems ? b : 0];
}

template<int R, int C>
__device__ __forceinline__ void loadVec(const double * base, const FmpcBuffers & buf, int i, int b, Matrix<double, R, C> & out)
{
  NMPC_UNROLL
  for(int e = 0; e < R * C; e++)
  {
    out.data()[e] = base[at(buf, i, e, R * C, b)];
  }
}
} // namespace fmpc

// ---------------------------------------------------------------------------------------------------------------------
// problem-independent kernels: compiled into ONE translation unit (nmpc_amd/csrc/fmpc_capi.hip defines
// NMPC_AMD_FMPC_COMMON_KERNELS before including this header), the problem translation units see declarations only
// ---------------------------------------------------------------------------------------------------------------------
#ifdef NMPC_AMD_FMPC_COMMON_KERNELS

/** Start of a solve (FmpcSolver.hpp:156-231): every instance is marked as running, trace cleared. */
__global__ void fmpc_begin_kernel(FmpcBuffers buf)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= buf.B)
  {
    return;
  }
  buf.status[b] = fmpc::kStatusContinued;
  buf.iters[b] = 0;
  buf.flags[b] = 0;
  for(int k = 0; k < buf.max_iter * NMPC_HIP_FMPC_NTRACE; k++)
  {
    buf.trace[static_cast<size_t>(b) * buf.max_iter * NMPC_HIP_FMPC_NTRACE + k] = 0.0;
  }
}

/** checkVariable's non-negativity test (FmpcSolver.hpp:338-353): the reference throws std::runtime_error; here the instance
    gets NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE and is skipped, the C-ABI call reports NMPC_HIP_ERR_RUNTIME afterwards. */
__global__ void fmpc_check_variable_kernel(FmpcBuffers buf)
{
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid >= static_cast<size_t>(buf.B) * buf.T)
  {
    return;
  }
  const int b = static_cast<int>(tid % buf.B);
  const int i = static_cast<int>(tid / buf.B);
  bool negative = false;
  double dot = 0; // s_list[i] . nu_list[i]: what the first iteration's barrier update sums (later ones: fmpc_update_kernel)
  for(int j = 0; j < buf.G; j++)
  {
    const double sv = buf.s[fmpc::at(buf, i, j, buf.G, b)], nv = buf.nu[fmpc::at(buf, i, j, buf.G, b)];
    negative = negative || sv < 0 || nv < 0;
    dot += sv * nv;
  }
  buf.part[fmpc::at(buf, i, 3, fmpc::kPartSlots, b)] = dot;
  if(negative)
  {
    buf.status[b] = NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE; // every writer stores the same value
  }
}

/** Barrier parameter of the iteration, (19.19) in Nocedal & Wright (FmpcSolver.hpp:370-392); also opens the iteration's trace
    row (:363-366).  Block = 64 instances x kSlices horizon slices. */
__global__ void __launch_bounds__(64 * fmpc::kSlices) fmpc_barrier_kernel(FmpcBuffers buf, int iter)
{
  __shared__ double sh[fmpc::kSlices][64];
  const int lane = threadIdx.x & 63;
  const int q = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const bool act = b < buf.B && buf.status[b] == fmpc::kStatusContinued;
  double acc = 0;
  if(act && buf.update_barrier_eps)
  {
    const int chunk = (buf.T + fmpc::kSlices - 1) / fmpc::kSlices;
    const int i1 = min(buf.T, (q + 1) * chunk);
    for(int i = q * chunk; i < i1; i++)
    {
      acc += buf.part[fmpc::at(buf, i, 3, fmpc::kPartSlots, b)]; // s_list[i] . nu_list[i], left by the kernel that wrote s, nu
    }
  }
  sh[q][lane] = acc;
  syncThreadsFuzzed(__LINE__);
  if(q == 0 && act)
  {
    double eps = buf.barrier_eps[b];
    if(buf.update_barrier_eps)
    {
      double s_nu_ave = 0;
      for(int k = 0; k < fmpc::kSlices; k++)
      {
        s_nu_ave += sh[k][lane];
      }
      s_nu_ave /= static_cast<double>(buf.T * buf.G);
      constexpr double sigma = 0.5;
      constexpr double barrier_eps_min = 1e-8;
      constexpr double barrier_eps_max = 1e6;
      // std::clamp(v, lo, hi) = (v < lo) ? lo : (hi < v) ? hi : v — a NaN passes through
      const double v = sigma * s_nu_ave;
      eps = (v < barrier_eps_min) ? barrier_eps_min : ((barrier_eps_max < v) ? barrier_eps_max : v);
      buf.barrier_eps[b] = eps;
    }
    buf.iters[b] = iter;
    buf.flags[b] = 0;
    double * row = buf.trace + (static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE;
    row[NMPC_HIP_FMPC_TRACE_ITER] = iter;
    row[NMPC_HIP_FMPC_TRACE_BARRIER_EPS] = eps;
  }
}

/** Fraction-to-boundary rule (FmpcSolver.hpp:713-742): minimum of the per-timestep candidates, validity test; and the verdict
    on the forward pass's NaN check (:699-706), which the reference takes first. */
__global__ void __launch_bounds__(64 * fmpc::kSlices) fmpc_step_length_kernel(FmpcBuffers buf, int iter)
{
  __shared__ double sh[2][fmpc::kSlices][64];
  const int lane = threadIdx.x & 63;
  const int q = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const bool act = b < buf.B && buf.status[b] == fmpc::kStatusContinued;
  double a_s = 1.0, a_nu = 1.0;
  if(act)
  {
    const int chunk = (buf.T + fmpc::kSlices - 1) / fmpc::kSlices;
    const int i1 = min(buf.T, (q + 1) * chunk);
    for(int i = q * chunk; i < i1; i++)
    {
      // std::min(a, b) = (b < a) ? b : a — a NaN candidate is ignored exactly as the reference's loop ignores it
      const double cs = buf.part[fmpc::at(buf, i, 1, fmpc::kPartSlots, b)];
      const double cn = buf.part[fmpc::at(buf, i, 2, fmpc::kPartSlots, b)];
      a_s = (cs < a_s) ? cs : a_s;
      a_nu = (cn < a_nu) ? cn : a_nu;
    }
  }
  sh[0][q][lane] = a_s;
  sh[1][q][lane] = a_nu;
  syncThreadsFuzzed(__LINE__);
  if(q == 0 && act)
  {
    for(int k = 1; k < fmpc::kSlices; k++)
    {
      a_s = (sh[0][k][lane] < a_s) ? sh[0][k][lane] : a_s;
      a_nu = (sh[1][k][lane] < a_nu) ? sh[1][k][lane] : a_nu;
    }
    if(buf.check_nan && (buf.flags[b] & 2))
    {
      buf.status[b] = 2; // Status::ErrorInForward
      return;
    }
    double * row = buf.trace + (static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE;
    row[NMPC_HIP_FMPC_TRACE_ALPHA_S_MAX] = a_s;
    row[NMPC_HIP_FMPC_TRACE_ALPHA_NU_MAX] = a_nu;
    row[NMPC_HIP_FMPC_TRACE_ALPHA_S] = a_s;
    buf.alpha[0 * buf.B + b] = a_s;
    buf.alpha[1 * buf.B + b] = a_nu;
    buf.alpha[2 * buf.B + b] = a_s;
    if(!(a_s > 0.0 && a_s <= 1.0 && a_nu > 0.0 && a_nu <= 1.0))
    {
      buf.status[b] = 4; // Status::ErrorInUpdate
    }
  }
}

/** Variable update (FmpcSolver.hpp:801-835), one thread per (instance, timestep). */
__global__ void fmpc_update_kernel(FmpcBuffers buf)
{
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid >= static_cast<size_t>(buf.B) * (buf.T + 1))
  {
    return;
  }
  const int b = static_cast<int>(tid % buf.B);
  const int i = static_cast<int>(tid / buf.B);
  if(buf.status[b] != fmpc::kStatusContinued)
  {
    return;
  }
  const double alpha_s = buf.alpha[2 * buf.B + b];
  const double alpha_nu = buf.alpha[1 * buf.B + b];
  for(int e = 0; e < buf.N; e++)
  {
    const size_t k = fmpc::at(buf, i, e, buf.N, b);
    buf.x[k] += alpha_s * buf.dx[k];
    buf.lam[k] += alpha_nu * buf.dlam[k];
  }
  if(i < buf.T)
  {
    for(int e = 0; e < buf.M; e++)
    {
      const size_t k = fmpc::at(buf, i, e, buf.M, b);
      buf.u[k] += alpha_s * buf.du[k];
    }
    // the reference clamps at numeric_limits<double>::lowest() (= -DBL_MAX, FmpcSolver.hpp:812) when an entry went negative:
    // restated as written (array().max(c) = (a < c) ? c : a)
    constexpr double min_positive_value = -DBL_MAX;
    bool s_neg = false, nu_neg = false;
    double dot = 0; // s . nu of the updated timestep, for the next iteration's barrier update (FmpcSolver.hpp:376-380)
    for(int e = 0; e < buf.G; e++)
    {
      const size_t k = fmpc::at(buf, i, e, buf.G, b);
      const double sv = buf.s[k] + alpha_s * buf.ds[k];
      const double nv = buf.nu[k] + alpha_nu * buf.dnu[k];
      buf.s[k] = sv;
      buf.nu[k] = nv;
      dot += sv * nv;
      s_neg = s_neg || sv < 0;
      nu_neg = nu_neg || nv < 0;
    }
    if(s_neg || nu_neg)
    {
      dot = 0;
      for(int e = 0; e < buf.G; e++)
      {
        const size_t k = fmpc::at(buf, i, e, buf.G, b);
        double sv = buf.s[k], nv = buf.nu[k];
        if(s_neg && sv < min_positive_value)
        {
          sv = min_positive_value;
          buf.s[k] = sv;
        }
        if(nu_neg && nv < min_positive_value)
        {
          nv = min_positive_value;
          buf.nu[k] = nv;
        }
        dot += sv * nv;
      }
    }
    buf.part[fmpc::at(buf, i, 3, fmpc::kPartSlots, b)] = dot;
  }
}

/** End of solve (FmpcSolver.hpp:233-246): IterationContinued becomes MaxIterationReached; with max_iter = 0 the loop never
    ran and solve() returns the initial Status::Uninitialized. */
__global__ void fmpc_finish_kernel(FmpcBuffers buf)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b < buf.B && buf.status[b] == fmpc::kStatusContinued)
  {
    buf.status[b] = buf.max_iter > 0 ? 5 : 0; // Status::MaxIterationReached : Status::Uninitialized
  }
}

/** Layout change between the C-ABI's arrays ([B][steps][E], the reference's per-instance std::vector of vectors) and the
    device arrays ([steps][E][B]): the transposition of a B x (steps E) matrix, through 32 x 32 tiles in LDS so that reads and writes
    are both whole lines (element by element one side of it touched a cache line per element: 20 us per field of the 4096 x T 200
    bench step, six fields per solve).  to_device = 1: dst is the device layout.  Grid: (columns / 32, rows / 32) of the SOURCE
    matrix (fmpcTransposeGrid), 256 threads. */
__global__ void __launch_bounds__(256) fmpc_transpose_kernel(const double * src, double * dst, int B, int steps, int E, int to_device)
{
  __shared__ double tile[32][33];
  const int R = steps * E;
  const int rows = to_device ? B : R, cols = to_device ? R : B; // src: [rows][cols], dst: [cols][rows]
  const int c0 = static_cast<int>(blockIdx.x) * 32, r0 = static_cast<int>(blockIdx.y) * 32;
  const int tx = static_cast<int>(threadIdx.x) & 31, ty = static_cast<int>(threadIdx.x) >> 5;
  for(int k = ty; k < 32; k += 8)
  {
    const int r = r0 + k, c = c0 + tx;
    if(r < rows && c < cols)
    {
      tile[k][tx] = src[static_cast<size_t>(r) * cols + c];
    }
  }
  syncThreadsFuzzed(__LINE__);
  for(int k = ty; k < 32; k += 8)
  {
    const int c = c0 + k, r = r0 + tx;
    if(r < rows && c < cols)
    {
      dst[static_cast<size_t>(c) * rows + r] = tile[tx][k];
    }
  }
}
#endif // NMPC_AMD_FMPC_COMMON_KERNELS

// ---------------------------------------------------------------------------------------------------------------------
// kernels instantiated per problem type
// ---------------------------------------------------------------------------------------------------------------------

/** init_complementary_variable (FmpcSolver.hpp:170-187). */
template<class Problem>
__global__ void fmpc_init_complementary_kernel(FmpcBuffers buf)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid >= static_cast<size_t>(buf.B) * buf.T)
  {
    return;
  }
  const int b = static_cast<int>(tid % buf.B);
  const int i = static_cast<int>(tid / buf.B);
  constexpr double initial_barrier_eps = 1e-4;
  constexpr double complementary_variable_margin_rate = 1e-2;
  constexpr double complementary_variable_min = 1e-2;
  const Problem prob = fmpc::loadProblem<Problem>(buf, b);
  typename Problem::StateDimVector x;
  typename Problem::InputDimVector u;
  fmpc::loadVec(buf.x, buf, i, b, x);
  fmpc::loadVec(buf.u, buf, i, b, u);
  const double t = buf.t0[b] + i * prob.dt();
  const typename Problem::IneqDimVector g = prob.ineqConst(t, x, u);
  NMPC_UNROLL
  for(int j = 0; j < G; j++)
  {
    const double neg_g = -1 * g[j];
    const double sj = (1.0 + complementary_variable_margin_rate) * (neg_g < complementary_variable_min ? complementary_variable_min : neg_g);
    const double r = initial_barrier_eps * (1.0 / sj);
    buf.s[fmpc::at(buf, i, j, G, b)] = sj;
    buf.nu[fmpc::at(buf, i, j, G, b)] =
        (1.0 + complementary_variable_margin_rate) * (r < complementary_variable_min ? complementary_variable_min : r);
  }
  if(i == 0)
  {
    buf.barrier_eps[b] = initial_barrier_eps;
  }
  (void)N;
  (void)M;
}

/** Step 1 of procOnce (FmpcSolver.hpp:394-441) for one (instance, timestep), fused with the pre-process of the backward pass
    (:562-574: everything of it that does not depend on P) and with this timestep's terms of calcKktError (:493-520). */
namespace fmpc
{
/** The coefficient of timestep i of instance b (FmpcSolver.hpp:395-436 with the pre-process of the backward pass :562-574, and the
    terms of calcKktError :493-521) — what fmpc_coeff_kernel computes per thread, with its RESULTS handed to a sink:
      sink.coef(e, v)      element e of the coefficient record (CoefLayout)          sink.kkt(v)      the timestep's KKT-error terms
      sink.terminal(e, v)  element e of the terminal gain record (i = T: s, P)       sink.nanFlag()   Coefficient::containsNaN
    A sink that ignores a result leaves its arithmetic dead: fmpc_kkt_kernel (KKT terms, NaN flag, terminal record) and the
    producer wave of fmpc_riccati_fused_kernel (the record, into LDS) are this function with other sinks. */
/** What coefficients() reads of the variable: requested by loadCoefInputs() — the producer wave of fmpc_riccati_fused_kernel does
    that a chunk before it computes from them, so that the round trip to HBM is off its path. */
/** a b + c in ONE rounding, written out.  The sums of the KKT-error terms and of (2.25b), (2.25c) start from zero: as `acc += a * b`
    their first two products are a sum of two products, of which the compiler fuses EITHER into the addition — which one depended on
    the code around it [measured: the terminal term of coefficients() inlined into fmpc_tail_kernel came out as fma(L0, L0, L1 L1),
    in fmpc_coeff_kernel as fma(L1, L1, L0 L0): one ulp apart in a quarter of the instances].  With the fusion spelled out every
    kernel that inlines coefficients() computes the same bits by construction. */
__device__ __forceinline__ double fused(double a, double b, double c)
{
  return __builtin_fma(a, b, c);
}
template<class Problem>
struct CoefInputs
{
  typename Problem::StateDimVector x, lambda, next_x, next_lambda;
  typename Problem::InputDimVector u;
  typename Problem::IneqDimVector s, nu;
};
template<class Problem>
__device__ __forceinline__ void loadCoefInputs(const FmpcBuffers & buf, int b, int i, CoefInputs<Problem> & in)
{
  loadVec(buf.x, buf, i, b, in.x);
  loadVec(buf.lam, buf, i, b, in.lambda);
  if(i < buf.T)
  {
    loadVec(buf.u, buf, i, b, in.u);
    loadVec(buf.x, buf, i + 1, b, in.next_x);
    loadVec(buf.lam, buf, i + 1, b, in.next_lambda);
    loadVec(buf.s, buf, i, b, in.s);
    loadVec(buf.nu, buf, i, b, in.nu);
  }
}
/** \tparam kPart 0: whatever timestep i is; 1: i is the terminal timestep (i == buf.T); 2: i is not — a caller that knows which
    (fmpc_tail_kernel) leaves the other branch out of its code */
template<class Problem, class Sink, int kPart = 0>
__device__ __forceinline__ void coefficientsOf(const FmpcBuffers & buf, int b, int i, const CoefInputs<Problem> & in, Sink & sink,
                                               const Problem & prob, double t0);
template<class Problem, class Sink, int kPart = 0>
__device__ __forceinline__ void coefficients(const FmpcBuffers & buf, int b, int i, const CoefInputs<Problem> & in, Sink & sink)
{
  const Problem prob = loadProblem<Problem>(buf, b);
  coefficientsOf<Problem, Sink, kPart>(buf, b, i, in, sink, prob, buf.t0[b]);
}
/** coefficients() with the instance's problem object and current_t handed in (fmpc_tail_kernel keeps them at hand: a load from
    global memory in the middle of its walk waits for everything requested before it). */
template<class Problem, class Sink, int kPart>
__device__ __forceinline__ void coefficientsOf(const FmpcBuffers & buf, int b, int i, const CoefInputs<Problem> & in, Sink & sink,
                                               const Problem & prob, double t0)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  using CL = CoefLayout<N, M>;
  using GL = GainLayout<N, M>;
  const double dt = prob.dt();
  const double t = t0 + i * dt;
  const typename Problem::StateDimVector & x = in.x, & lambda = in.lambda;
  double kkt = 0;
  bool nan = false;
  NanAccumulator nan_acc;

  if(kPart != 2 && (kPart == 1 || i == buf.T))
  {
    // terminal coefficient (:429-436), start of the backward recursion (2.34) (:541-548)
    typename Problem::StateDimVector Vx;
    typename Problem::StateStateDimMatrix Vxx;
    prob.calcTerminalCostDeriv(t, x, Vx, Vxx);
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      const double Lx_bar = Vx[a] - lambda[a]; // (2.25a)
      kkt = fused(Lx_bar, Lx_bar, kkt);
      const double sT = -1 * Lx_bar;
      nan = nan || bad(Vx[a]) || bad(Lx_bar);
      sink.terminal(GL::S + a, sT);
    }
    NMPC_UNROLL
    for(int e = 0; e < N * N; e++)
    {
      nan = nan || bad(Vxx.data()[e]);
      sink.terminal(GL::P + e, Vxx.data()[e]);
    }
    sink.kkt(kkt);
    if(nan)
    {
      sink.nanFlag();
    }
    return;
  }

  const typename Problem::InputDimVector & u = in.u;
  const typename Problem::StateDimVector & next_x = in.next_x, & next_lambda = in.next_lambda;
  const typename Problem::IneqDimVector & s = in.s, & nu = in.nu;

  typename Problem::StateStateDimMatrix A, Lxx;
  typename Problem::StateInputDimMatrix Bm, Lxu;
  typename Problem::IneqStateDimMatrix C;
  typename Problem::IneqInputDimMatrix D;
  typename Problem::StateDimVector Lx;
  typename Problem::InputDimVector Lu;
  typename Problem::InputInputDimMatrix Luu;
  prob.calcStateEqDeriv(t, x, u, A, Bm);
  prob.calcIneqConstDeriv(t, x, u, C, D);
  prob.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
  const typename Problem::StateDimVector f = prob.stateEq(t, x, u);
  const typename Problem::IneqDimVector g = prob.ineqConst(t, x, u);

  if(i == 0)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      const double e = buf.x0[static_cast<size_t>(a) * buf.B + b] - x[a];
      kkt = fused(e, e, kkt);
    }
  }
  double x_bar[N], g_bar[G > 0 ? G : 1], Lx_bar[N], Lu_bar[M > 0 ? M : 1];
  double part = 0;
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    x_bar[a] = f[a] - next_x[a]; // (2.23c)
    part = fused(x_bar[a], x_bar[a], part);
    nan_acc.add(x_bar[a]);
  }
  kkt += part;
  part = 0;
  NMPC_UNROLL
  for(int a = 0; a < G; a++)
  {
    g_bar[a] = g[a] + s[a]; // (2.23d)
    part = fused(g_bar[a], g_bar[a], part);
    nan_acc.add(g_bar[a]);
  }
  kkt += part;
  part = 0;
  NMPC_UNROLL
  for(int a = 0; a < N; a++) // (2.25b)
  {
    double at = 0, ct = 0;
    NMPC_UNROLL
    for(int r = 0; r < N; r++)
    {
      at = fused(A(r, a), next_lambda[r], at);
    }
    NMPC_UNROLL
    for(int r = 0; r < G; r++)
    {
      ct = fused(C(r, a), nu[r], ct);
    }
    Lx_bar[a] = ((-1 * lambda[a] + dt * Lx[a]) + at) + ct;
    part = fused(Lx_bar[a], Lx_bar[a], part);
    nan_acc.add(Lx_bar[a]);
    nan_acc.add(Lx[a]);
  }
  kkt += part;
  part = 0;
  NMPC_UNROLL
  for(int a = 0; a < M; a++) // (2.25c)
  {
    double bt = 0, dn = 0;
    NMPC_UNROLL
    for(int r = 0; r < N; r++)
    {
      bt = fused(Bm(r, a), next_lambda[r], bt);
    }
    NMPC_UNROLL
    for(int r = 0; r < G; r++)
    {
      dn = fused(D(r, a), nu[r], dn);
    }
    Lu_bar[a] = (dt * Lu[a] + bt) + dn;
    part = fused(Lu_bar[a], Lu_bar[a], part);
    nan_acc.add(Lu_bar[a]);
    nan_acc.add(Lu[a]);
  }
  kkt += part;
  part = 0;
  NMPC_UNROLL
  for(int j = 0; j < G; j++) // complementarity term of calcKktError(0.0) (:509-510)
  {
    const double v = s[j] * nu[j];
    const double e = v < 0.0 ? 0.0 : v; // array().max(0): (a < 0) ? 0 : a
    part = fused(e, e, part);
  }
  kkt += part;
  sink.kkt(kkt);

  // Coefficient::containsNaN (:136-154) on what is not stored below
  NMPC_UNROLL
  for(int e = 0; e < N * N; e++)
  {
    nan_acc.add(A.data()[e]);
    nan_acc.add(Lxx.data()[e]);
  }
  NMPC_UNROLL
  for(int e = 0; e < N * M; e++)
  {
    nan_acc.add(Bm.data()[e]);
    nan_acc.add(Lxu.data()[e]);
  }
  NMPC_UNROLL
  for(int e = 0; e < G * N; e++)
  {
    nan_acc.add(C.data()[e]);
  }
  NMPC_UNROLL
  for(int e = 0; e < G * M; e++)
  {
    nan_acc.add(D.data()[e]);
  }
  NMPC_UNROLL
  for(int e = 0; e < M * M; e++)
  {
    nan_acc.add(Luu.data()[e]);
  }
  if(nan || nan_acc.any())
  {
    sink.nanFlag();
  }

  sink.midpoint(); // (everything above keeps its results in registers; every sink.coef() call is below)
  // pre-process of the backward pass (:562-574)
  const double barrier_eps = buf.barrier_eps[b];
  double nu_s[G > 0 ? G : 1], tilde_sub[G > 0 ? G : 1];
  NMPC_UNROLL
  for(int j = 0; j < G; j++)
  {
    nu_s[j] = nu[j] / s[j];
    tilde_sub[j] = (nu_s[j] * g_bar[j] - nu[j]) + barrier_eps * (1.0 / s[j]);
  }
  NMPC_UNROLL
  for(int e = 0; e < N * N; e++)
  {
    sink.coef(CL::A + e, A.data()[e]);
  }
  NMPC_UNROLL
  for(int e = 0; e < N * M; e++)
  {
    sink.coef(CL::B + e, Bm.data()[e]);
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    sink.coef(CL::XBAR + a, x_bar[a]);
  }
  NMPC_UNROLL
  for(int c = 0; c < N; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int j = 0; j < G; j++)
      {
        acc += (C(j, a) * nu_s[j]) * C(j, c);
      }
      sink.coef(CL::QXX + a + c * N, dt * Lxx(a, c) + acc); // (2.28c)
    }
  }
  NMPC_UNROLL
  for(int c = 0; c < M; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int j = 0; j < G; j++)
      {
        acc += (D(j, a) * nu_s[j]) * D(j, c);
      }
      sink.coef(CL::QUU + a + c * M, dt * Luu(a, c) + acc); // (2.28e)
    }
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int j = 0; j < G; j++)
      {
        acc += (C(j, a) * nu_s[j]) * D(j, c);
      }
      sink.coef(CL::QXU + a + c * N, dt * Lxu(a, c) + acc); // (2.28d)
    }
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    double acc = 0;
    NMPC_UNROLL
    for(int j = 0; j < G; j++)
    {
      acc += C(j, a) * tilde_sub[j];
    }
    sink.coef(CL::LXT + a, Lx_bar[a] + acc); // (2.28f)
  }
  NMPC_UNROLL
  for(int a = 0; a < M; a++)
  {
    double acc = 0;
    NMPC_UNROLL
    for(int j = 0; j < G; j++)
    {
      acc += D(j, a) * tilde_sub[j];
    }
    sink.coef(CL::LUT + a, Lu_bar[a] + acc); // (2.28g)
  }
}

/** Sink of fmpc_coeff_kernel: everything to HBM. */
template<int N, int M, bool kRecord>
struct GlobalCoefSink
{
  const FmpcBuffers & buf;
  int b, i;
  __device__ __forceinline__ void coef(int e, double v) const
  {
    if constexpr(kRecord)
    {
      buf.coef[at(buf, i, e, CoefLayout<N, M>::kStride, b)] = v;
    }
  }
  __device__ __forceinline__ void terminal(int e, double v) const
  {
    buf.gain[at(buf, i, e, GainLayout<N, M>::kStride, b)] = v;
  }
  __device__ __forceinline__ void kkt(double v) const
  {
    buf.part[at(buf, i, 0, kPartSlots, b)] = v;
  }
  __device__ __forceinline__ void nanFlag() const
  {
    atomicOr(&buf.flags[b], 1);
  }
  __device__ __forceinline__ void midpoint() const {}
};
} // namespace fmpc

/** One thread per (instance, timestep): the coefficient record, the KKT-error terms, the NaN flag.
    \tparam kRecord false: everything but the record — what is launched in front of fmpc_riccati_fused_kernel, whose producer wave
    computes the records into its staging LDS (A, B, x_bar also to HBM, for the forward sweep) */
template<class Problem, bool kRecord = true>
__global__ void __launch_bounds__(256) fmpc_coeff_kernel(FmpcBuffers buf)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax;
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid >= static_cast<size_t>(buf.B) * (buf.T + 1))
  {
    return;
  }
  const int b = static_cast<int>(tid % buf.B);
  const int i = static_cast<int>(tid / buf.B);
  if(buf.status[b] != fmpc::kStatusContinued)
  {
    return;
  }
  fmpc::GlobalCoefSink<N, M, kRecord> sink{buf, b, i};
  fmpc::CoefInputs<Problem> in;
  fmpc::loadCoefInputs<Problem>(buf, b, i, in);
  fmpc::coefficients<Problem>(buf, b, i, in, sink);
}

namespace fmpc
{
/** Sink of fmpc_tail_kernel: the timestep's KKT-error terms; of the terminal timestep also the terminal record and its NaN verdict.
    The records of the other timesteps — and THEIR NaN verdict — are the business of fmpc_riccati_fused_kernel's producer waves, so
    everything of coefficients() behind the KKT terms is dead code here. */
template<int N, int M, bool kTerminal>
struct TailSink
{
  const FmpcBuffers & buf;
  int b, i;
  int * nan_flag;
  __device__ __forceinline__ void coef(int, double) const {}
  __device__ __forceinline__ void terminal(int e, double v) const
  {
    if constexpr(kTerminal)
    {
      buf.gain[at(buf, i, e, GainLayout<N, M>::kStride, b)] = v;
    }
  }
  __device__ __forceinline__ void kkt(double v) const
  {
    buf.part[at(buf, i, 0, kPartSlots, b)] = v;
  }
  __device__ __forceinline__ void nanFlag() const
  {
    if constexpr(kTerminal)
    {
      *nan_flag = 1; // (benign race: every writer stores 1)
    }
  }
  __device__ __forceinline__ void midpoint() const {}
};

/** v + a d, the variable update (FmpcSolver.hpp:803-808): ONE function for the timestep's owner and for the neighbour that needs the
    updated value before the owner has stored it, so both evaluate the same instruction. */
__device__ __forceinline__ double stepped(double v, double a, double d)
{
  return v + a * d;
}
// Horizon slices of a workgroup of fmpc_tail_kernel at most (x 16 instances = its threads).  16: 256 threads, one wave per SIMD with
// 512 registers — the KKT terms' model code next to a timestep in flight needs ~300 (cart-pole); with 32 / 48 / 64 slices (256 /
// 168 / 128 registers) it spills, and a reload waits for every request in flight [measured, us per launch: 81 / 87 / 165 / 185].
#ifndef NMPC_AMD_FMPC_TAIL_SLICES
#  define NMPC_AMD_FMPC_TAIL_SLICES 16
#endif
constexpr int kTailMaxSlices = NMPC_AMD_FMPC_TAIL_SLICES;
/** Byte offset of element (timestep i, entry e, instance b) in 32 bits, and accesses by it: fmpc_tail_kernel's arrays are below 4 GB
    (tailFits), and a row offset is then ONE register shared by every array of the same entry count — as 64-bit addresses, one pair
    per access, the walk's body needed more registers than a 512-thread workgroup has [measured: 256 + 44 spilled]. */
__device__ __forceinline__ unsigned off32(const FmpcBuffers & buf, int i, int e, int stride, int b)
{
  return ((static_cast<unsigned>(i) * stride + e) * static_cast<unsigned>(buf.B) + b) * 8u;
}
__device__ __forceinline__ double ld32(const double * base, unsigned off)
{
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ void st32(double * base, unsigned off, double v)
{
  *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + off) = v;
}
/** Dynamic LDS of fmpc_tail_kernel for the s . nu terms of a horizon of T steps, 0 when that is more than a workgroup gets. */
inline unsigned tailDotBytes(int T)
{
  const size_t bytes = static_cast<size_t>(T) * 16 * sizeof(double);
  return bytes <= 48 * 1024 ? static_cast<unsigned>(bytes) : 0u;
}
/** Whether every array fmpc_tail_kernel touches is below 4 GB. */
inline bool tailFits(const FmpcBuffers & buf)
{
  int widest = buf.gain_stride;
  widest = buf.N > widest ? buf.N : widest;
  widest = buf.G > widest ? buf.G : widest;
  widest = kPartSlots > widest ? kPartSlots : widest;
  return static_cast<double>(buf.T + 1) * widest * buf.B * 8.0 < 4.0e9;
}
/** Horizon slices of fmpc_tail_kernel for a horizon of T steps (a workgroup has 16 x this many threads). */
inline int tailSlices(int T)
{
  const int per = (T + 1 + kTailMaxSlices - 1) / kTailMaxSlices;
  int S = (T + 1 + per - 1) / per;
  S = (S + 3) & ~3; // whole wavefronts
  return S < kSlices ? kSlices : S;
}
} // namespace fmpc

/** Everything of an iteration behind the forward pass, and the head of the next one, in ONE kernel (sequence with
    fmpc_riccati_fused_kernel, line search off): the fraction-to-boundary reduction (fmpc_step_length_kernel, FmpcSolver.hpp:713-742),
    the variable update (fmpc_update_kernel, :801-835), and for iteration iter + 1 the barrier parameter (fmpc_barrier_kernel,
    :370-392), the KKT-error terms and the terminal record (fmpc_coeff_kernel<Problem, false>, :429-436, :493-521) — computed from the
    updated variables while they are in registers instead of being read back by two more kernels; same arithmetic, same order of the
    horizon sums, same bits.  A workgroup owns the WHOLE horizon of sixteen instances (one 128-byte line per row): thread = (horizon
    slice, instance), a slice is a run of consecutive timesteps walked from its last to its first, so that the updated x, lambda of
    timestep i + 1 — which the KKT terms of timestep i need (2.23c), (2.25b) — are the ones the thread computed a moment ago; only for
    the last timestep of its run it computes the neighbour's (before anybody stores: the barrier in front of the walk).
    \param last  iter is the solve's last iteration: nothing of an iteration iter + 1
    \param dots_in_lds  the launch gave T x 16 doubles of dynamic LDS for the s . nu terms (fmpc::tailDotBytes; else they are read back
                        from FmpcBuffers::part) */
template<class Problem>
__global__ void __launch_bounds__(16 * fmpc::kTailMaxSlices) fmpc_tail_kernel(FmpcBuffers buf, int iter, int last, int dots_in_lds)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  extern __shared__ double sh_dot[]; // [T][16] s . nu of the updated timesteps when dots_in_lds (the launch sized it), else unused
  __shared__ double sh_a[2][fmpc::kTailMaxSlices][16];
  __shared__ double sh_sum[fmpc::kSlices][16];
  __shared__ double sh_alpha[2][16];
  __shared__ int sh_go[16];
  __shared__ int sh_nan[16];
  // the sixteen problem objects: read from LDS inside the walk (a global load there would wait for the requests of the next timestep)
  static_assert(sizeof(Problem) % sizeof(unsigned) == 0, "[FMPC] a problem object is a whole number of 32-bit words");
  constexpr int kProbWords = sizeof(Problem) / sizeof(unsigned);
  __shared__ __attribute__((aligned(16))) unsigned sh_prob[16][(kProbWords + 3) / 4 * 4];
  const int inst = threadIdx.x & 15;
  const int q = threadIdx.x >> 4;
  const int S = blockDim.x >> 4;
  for(int k = threadIdx.x; k < 16 * kProbWords; k += blockDim.x)
  {
    const int pi = k / kProbWords, w = k % kProbWords;
    const int pb = min(static_cast<int>(blockIdx.x) * 16 + pi, buf.B - 1);
    sh_prob[pi][w] = reinterpret_cast<const unsigned *>(static_cast<const Problem *>(buf.problems) + (buf.own_problems ? pb : 0))[w];
  }
  const int b_raw = blockIdx.x * 16 + inst;
  const int b = b_raw < buf.B ? b_raw : buf.B - 1; // lanes beyond the batch mirror the last instance and store nothing
  const int T = buf.T;
  const bool act = b_raw < buf.B && buf.status[b] == fmpc::kStatusContinued;
  const int per = (T + 1 + S - 1) / S;
  const int lo = q * per;
  const int hi = min(T + 1, lo + per); // this thread's timesteps: [lo, hi), possibly none
#ifdef NMPC_AMD_FMPC_TAIL_PROFILE
  const unsigned long long tick_a = wall_clock64();
#endif
  /** What the update of one timestep reads. */
  struct Loaded
  {
    double x[N], dx[N], lam[N], dlam[N];
    double u[M > 0 ? M : 1], du[M > 0 ? M : 1];
    double s[G > 0 ? G : 1], ds[G > 0 ? G : 1], nu[G > 0 ? G : 1], dnu[G > 0 ? G : 1];
  };
  auto request = [&](int i, Loaded & v) {
    NMPC_UNROLL
    for(int e = 0; e < N; e++)
    {
      const unsigned k = fmpc::off32(buf, i, e, N, b);
      v.x[e] = fmpc::ld32(buf.x, k);
      v.dx[e] = fmpc::ld32(buf.dx, k);
      v.lam[e] = fmpc::ld32(buf.lam, k);
      v.dlam[e] = fmpc::ld32(buf.dlam, k);
    }
    if(i < T)
    {
      NMPC_UNROLL
      for(int e = 0; e < M; e++)
      {
        const unsigned k = fmpc::off32(buf, i, e, M, b);
        v.u[e] = fmpc::ld32(buf.u, k);
        v.du[e] = fmpc::ld32(buf.du, k);
      }
      NMPC_UNROLL
      for(int e = 0; e < G; e++)
      {
        const unsigned k = fmpc::off32(buf, i, e, G, b);
        v.s[e] = fmpc::ld32(buf.s, k);
        v.ds[e] = fmpc::ld32(buf.ds, k);
        v.nu[e] = fmpc::ld32(buf.nu, k);
        v.dnu[e] = fmpc::ld32(buf.dnu, k);
      }
    }
  };
  // ---- requested first, used behind the step-length reduction (one round trip for all three): x, dx, lambda, dlambda of the
  // timestep behind this thread's run — read before anybody stores, the workgroup barriers of the reduction see to that — and the
  // run's last timestep
  const int top = hi - 1;
  const bool has_run = lo < hi;
  const bool has_behind = has_run && top < T;
  double bx[N] = {}, bdx[N] = {}, bl[N] = {}, bdl[N] = {};
  Loaded next;
  if(act && has_behind)
  {
    NMPC_UNROLL
    for(int e = 0; e < N; e++)
    {
      const unsigned k = fmpc::off32(buf, top + 1, e, N, b);
      bx[e] = fmpc::ld32(buf.x, k);
      bdx[e] = fmpc::ld32(buf.dx, k);
      bl[e] = fmpc::ld32(buf.lam, k);
      bdl[e] = fmpc::ld32(buf.dlam, k);
    }
  }
  if(act && has_run)
  {
    request(top, next);
  }
  const double t0 = buf.t0[b];

  // ---- step length: minimum of the per-timestep candidates, in timestep order (the first of equal minima wins, as in the
  // sequential loop of the reference and in fmpc_step_length_kernel)
  // (loads in batches of eight: the compiler does not move a load of the next trip over the compare of this one, and a trip per
  // round trip to L2 was a third of the kernel)
  constexpr int kBatch = 16;
  double a_s = 1.0, a_nu = 1.0;
  if(act)
  {
    const int i1 = min(hi, T);
    for(int i0 = lo; i0 < i1; i0 += kBatch)
    {
      double cs[kBatch], cn[kBatch];
      NMPC_UNROLL
      for(int j = 0; j < kBatch; j++)
      {
        const int i = min(i0 + j, i1 - 1);
        cs[j] = fmpc::ld32(buf.part, fmpc::off32(buf, i, 1, fmpc::kPartSlots, b));
        cn[j] = fmpc::ld32(buf.part, fmpc::off32(buf, i, 2, fmpc::kPartSlots, b));
      }
      NMPC_UNROLL
      for(int j = 0; j < kBatch; j++)
      {
        if(i0 + j < i1)
        {
          a_s = (cs[j] < a_s) ? cs[j] : a_s;
          a_nu = (cn[j] < a_nu) ? cn[j] : a_nu;
        }
      }
    }
  }
  sh_a[0][q][inst] = a_s;
  sh_a[1][q][inst] = a_nu;
  if(q == 0)
  {
    sh_nan[inst] = 0;
  }
  // (the values of the timestep behind the run have ARRIVED before the barrier — a use the compiler cannot move: their owner
  // stores the updated ones two barriers from here)
  NMPC_UNROLL
  for(int e = 0; e < N; e++)
  {
    asm volatile("" ::"v"(bx[e]), "v"(bdx[e]), "v"(bl[e]), "v"(bdl[e]));
  }
  syncThreadsFuzzed(__LINE__);
  if(q == 0)
  {
    bool go = false;
    if(act)
    {
      for(int k = 1; k < S; k++)
      {
        a_s = (sh_a[0][k][inst] < a_s) ? sh_a[0][k][inst] : a_s;
        a_nu = (sh_a[1][k][inst] < a_nu) ? sh_a[1][k][inst] : a_nu;
      }
      if(buf.check_nan && (buf.flags[b] & 2))
      {
        buf.status[b] = 2; // Status::ErrorInForward
      }
      else
      {
        double * row = buf.trace + (static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE;
        row[NMPC_HIP_FMPC_TRACE_ALPHA_S_MAX] = a_s;
        row[NMPC_HIP_FMPC_TRACE_ALPHA_NU_MAX] = a_nu;
        row[NMPC_HIP_FMPC_TRACE_ALPHA_S] = a_s;
        buf.alpha[0 * buf.B + b] = a_s;
        buf.alpha[1 * buf.B + b] = a_nu;
        buf.alpha[2 * buf.B + b] = a_s;
        if(!(a_s > 0.0 && a_s <= 1.0 && a_nu > 0.0 && a_nu <= 1.0))
        {
          buf.status[b] = 4; // Status::ErrorInUpdate
        }
        else
        {
          go = true;
        }
      }
    }
    sh_alpha[0][inst] = a_s;
    sh_alpha[1][inst] = a_nu;
    sh_go[inst] = go ? 1 : 0;
  }
  syncThreadsFuzzed(__LINE__);
  const bool go = sh_go[inst] != 0; // the instance takes the step
  const double alpha_s = sh_alpha[0][inst], alpha_nu = sh_alpha[1][inst];
  const Problem & prob = *reinterpret_cast<const Problem *>(sh_prob[inst]);

  // ---- the updated x, lambda of the timestep behind this thread's run
  typename Problem::StateDimVector nx, nl;
  if(go && has_behind)
  {
    NMPC_UNROLL
    for(int e = 0; e < N; e++)
    {
      nx[e] = fmpc::stepped(bx[e], alpha_s, bdx[e]);
      nl[e] = fmpc::stepped(bl[e], alpha_nu, bdl[e]);
    }
  }
#ifdef NMPC_AMD_FMPC_TAIL_PROFILE
  const unsigned long long tick_b = wall_clock64();
#endif

  // ---- update, s . nu and the KKT-error terms of the updated timestep, last timestep of the run first.  A timestep's values are
  // all requested before the previous timestep's are stored (the stores of timestep i and the loads of timestep i - 1 may alias as
  // far as the compiler knows: written in source order they were ~17 dependent round trips per timestep [measured: 118 us])
  if(go)
  {
    for(int walk = top; walk >= lo; walk--)
    {
      // (the timestep behind an empty asm: the ~120 row addresses of the body are then computed from it where they are used — as
      // induction variables of the walk they were 2 x 120 registers that lived across the whole body [measured: 251 spills])
      int i = walk;
      asm volatile("" : "+v"(i));
      const Loaded v = next;
      if(walk > lo)
      {
        request(i - 1, next);
      }
      fmpc::CoefInputs<Problem> in;
      NMPC_UNROLL
      for(int e = 0; e < N; e++)
      {
        in.x[e] = fmpc::stepped(v.x[e], alpha_s, v.dx[e]);
        in.lambda[e] = fmpc::stepped(v.lam[e], alpha_nu, v.dlam[e]);
      }
      double dot = 0; // s . nu of the updated timestep, for the barrier update below (FmpcSolver.hpp:376-380)
      if(i < T)
      {
        NMPC_UNROLL
        for(int e = 0; e < M; e++)
        {
          in.u[e] = fmpc::stepped(v.u[e], alpha_s, v.du[e]);
        }
        // the reference clamps at numeric_limits<double>::lowest() (= -DBL_MAX, FmpcSolver.hpp:812) when an entry went negative:
        // restated as written (array().max(c) = (a < c) ? c : a)
        constexpr double min_positive_value = -DBL_MAX;
        bool s_neg = false, nu_neg = false;
        NMPC_UNROLL
        for(int e = 0; e < G; e++)
        {
          in.s[e] = fmpc::stepped(v.s[e], alpha_s, v.ds[e]);
          in.nu[e] = fmpc::stepped(v.nu[e], alpha_nu, v.dnu[e]);
          s_neg = s_neg || in.s[e] < 0;
          nu_neg = nu_neg || in.nu[e] < 0;
        }
        NMPC_UNROLL
        for(int e = 0; e < G; e++)
        {
          if(s_neg && in.s[e] < min_positive_value)
          {
            in.s[e] = min_positive_value;
          }
          if(nu_neg && in.nu[e] < min_positive_value)
          {
            in.nu[e] = min_positive_value;
          }
          dot += in.s[e] * in.nu[e];
        }
      }
      // stores
      NMPC_UNROLL
      for(int e = 0; e < N; e++)
      {
        const unsigned k = fmpc::off32(buf, i, e, N, b);
        fmpc::st32(buf.x, k, in.x[e]);
        fmpc::st32(buf.lam, k, in.lambda[e]);
      }
      if(i < T)
      {
        NMPC_UNROLL
        for(int e = 0; e < M; e++)
        {
          fmpc::st32(buf.u, fmpc::off32(buf, i, e, M, b), in.u[e]);
        }
        NMPC_UNROLL
        for(int e = 0; e < G; e++)
        {
          const unsigned k = fmpc::off32(buf, i, e, G, b);
          fmpc::st32(buf.s, k, in.s[e]);
          fmpc::st32(buf.nu, k, in.nu[e]);
        }
        fmpc::st32(buf.part, fmpc::off32(buf, i, 3, fmpc::kPartSlots, b), dot);
        if(dots_in_lds)
        {
          sh_dot[i * 16 + inst] = dot;
        }
      }
      if(!last)
      {
        if(i == T)
        {
          fmpc::TailSink<N, M, true> sink{buf, b, i, &sh_nan[inst]};
          fmpc::coefficientsOf<Problem, fmpc::TailSink<N, M, true>, 1>(buf, b, i, in, sink, prob, t0);
        }
        else
        {
          in.next_x = nx;
          in.next_lambda = nl;
          fmpc::TailSink<N, M, false> sink{buf, b, i, nullptr};
          fmpc::coefficientsOf<Problem, fmpc::TailSink<N, M, false>, 2>(buf, b, i, in, sink, prob, t0);
        }
      }
      nx = in.x;
      nl = in.lambda;
    }
  }
#ifdef NMPC_AMD_FMPC_TAIL_PROFILE
  const unsigned long long tick_c = wall_clock64();
  syncThreadsFuzzed(__LINE__);
  if(q == 0 && b_raw < buf.B && !last) // developer build only: 100 MHz ticks in the merit slots — reductions, this slice's walk, wait for the slowest
  {
    buf.merit[0 * buf.B + b] = static_cast<double>(tick_b - tick_a);
    buf.merit[1 * buf.B + b] = static_cast<double>(tick_c - tick_b);
    buf.merit[2 * buf.B + b] = static_cast<double>(wall_clock64() - tick_c);
  }
#endif
  if(last)
  {
    return; // workgroup-uniform
  }

  // ---- head of iteration iter + 1: barrier parameter (19.19), the sums in fmpc_barrier_kernel's order; trace row
  syncThreadsFuzzed(__LINE__); // the s . nu terms of the sixteen instances are in memory (workgroup scope)
  if(q < fmpc::kSlices)
  {
    double acc = 0;
    if(go && buf.update_barrier_eps)
    {
      const int chunk = (T + fmpc::kSlices - 1) / fmpc::kSlices;
      const int i1 = min(T, (q + 1) * chunk);
      for(int i = q * chunk; i < (dots_in_lds ? i1 : 0); i++)
      {
        acc += sh_dot[i * 16 + inst];
      }
      for(int i0 = q * chunk; i0 < (dots_in_lds ? 0 : i1); i0 += kBatch)
      {
        double v[kBatch];
        NMPC_UNROLL
        for(int j = 0; j < kBatch; j++)
        {
          v[j] = fmpc::ld32(buf.part, fmpc::off32(buf, min(i0 + j, i1 - 1), 3, fmpc::kPartSlots, b));
        }
        NMPC_UNROLL
        for(int j = 0; j < kBatch; j++)
        {
          if(i0 + j < i1)
          {
            acc += v[j];
          }
        }
      }
    }
    sh_sum[q][inst] = acc;
  }
  syncThreadsFuzzed(__LINE__);
  if(q == 0 && go)
  {
    double eps = buf.barrier_eps[b];
    if(buf.update_barrier_eps)
    {
      double s_nu_ave = 0;
      for(int k = 0; k < fmpc::kSlices; k++)
      {
        s_nu_ave += sh_sum[k][inst];
      }
      s_nu_ave /= static_cast<double>(T * buf.G);
      constexpr double sigma = 0.5;
      constexpr double barrier_eps_min = 1e-8;
      constexpr double barrier_eps_max = 1e6;
      const double v = sigma * s_nu_ave;
      eps = (v < barrier_eps_min) ? barrier_eps_min : ((barrier_eps_max < v) ? barrier_eps_max : v);
      buf.barrier_eps[b] = eps;
    }
    buf.iters[b] = iter + 1;
    buf.flags[b] = sh_nan[inst]; // bit 0 of the terminal record; the producer waves of the Riccati kernel speak for the other records
    double * row = buf.trace + (static_cast<size_t>(b) * buf.max_iter + iter) * NMPC_HIP_FMPC_NTRACE;
    row[NMPC_HIP_FMPC_TRACE_ITER] = iter + 1;
    row[NMPC_HIP_FMPC_TRACE_BARRIER_EPS] = eps;
  }
}

namespace fmpc
{
/** What one step of the backward recursion reads: loaded one step ahead of its use (the recursion is a chain of dependent
    small-matrix products; with the loads issued a step early their latency overlaps the arithmetic of the current step). */
template<int N, int M>
struct BackwardRecord
{
  double A[N * N], Bm[N * M > 0 ? N * M : 1], x_bar[N];
  double F[N * N], H[N * M > 0 ? N * M : 1], Gm[M * M > 0 ? M * M : 1], Lx_t[N], Lu_t[M > 0 ? M : 1];

  __device__ __forceinline__ void load(const FmpcBuffers & buf, int i, int b)
  {
    using CL = CoefLayout<N, M>;
    NMPC_UNROLL
    for(int e = 0; e < N * N; e++)
    {
      A[e] = buf.coef[at(buf, i, CL::A + e, CL::kStride, b)];
      F[e] = buf.coef[at(buf, i, CL::QXX + e, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int e = 0; e < N * M; e++)
    {
      Bm[e] = buf.coef[at(buf, i, CL::B + e, CL::kStride, b)];
      H[e] = buf.coef[at(buf, i, CL::QXU + e, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int e = 0; e < M * M; e++)
    {
      Gm[e] = buf.coef[at(buf, i, CL::QUU + e, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      x_bar[a] = buf.coef[at(buf, i, CL::XBAR + a, CL::kStride, b)];
      Lx_t[a] = buf.coef[at(buf, i, CL::LXT + a, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      Lu_t[a] = buf.coef[at(buf, i, CL::LUT + a, CL::kStride, b)];
    }
  }
};

/** What one step of the forward recursion reads. */
template<int N, int M>
struct ForwardRecord
{
  double A[N * N], Bm[N * M > 0 ? N * M : 1], x_bar[N], K[M * N > 0 ? M * N : 1], k[M > 0 ? M : 1];

  __device__ __forceinline__ void load(const FmpcBuffers & buf, int i, int b)
  {
    using CL = CoefLayout<N, M>;
    using GL = GainLayout<N, M>;
    NMPC_UNROLL
    for(int e = 0; e < N * N; e++)
    {
      A[e] = buf.coef[at(buf, i, CL::A + e, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int e = 0; e < N * M; e++)
    {
      Bm[e] = buf.coef[at(buf, i, CL::B + e, CL::kStride, b)];
      K[e] = buf.gain[at(buf, i, GL::K + e, GL::kStride, b)];
    }
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      x_bar[a] = buf.coef[at(buf, i, CL::XBAR + a, CL::kStride, b)];
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      k[a] = buf.gain[at(buf, i, GL::k + a, GL::kStride, b)];
    }
  }
};

/** One step of backwardPass (FmpcSolver.hpp:550-637): (s, P) of step i + 1 in, of step i out; gains stored.  Returns false
    where the reference returns false (LDLT failure with break_if_llt_fails). */
template<int N, int M>
__device__ __forceinline__ bool backwardStep(const FmpcBuffers & buf, int i, int b, BackwardRecord<N, M> & r, double * s, double * P, bool & nan)
{
  using GL = GainLayout<N, M>;
  // F, H, G (2.35b-d) (:576-578): A^T P first, then times A / B; B^T P times B
  double AtP[N * N], BtP[N * M > 0 ? N * M : 1];
  NMPC_UNROLL
  for(int c = 0; c < N; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += r.A[q + a * N] * P[q + c * N];
      }
      AtP[a + c * N] = acc;
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += r.Bm[q + a * N] * P[q + c * N];
      }
      BtP[a + c * M] = acc;
    }
  }
  NMPC_UNROLL
  for(int c = 0; c < N; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += AtP[a + q * N] * r.A[q + c * N];
      }
      r.F[a + c * N] += acc;
    }
  }
  NMPC_UNROLL
  for(int c = 0; c < M; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += AtP[a + q * N] * r.Bm[q + c * N];
      }
      r.H[a + c * N] += acc;
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += BtP[a + q * M] * r.Bm[q + c * N];
      }
      r.Gm[a + c * M] += acc;
    }
  }

  // gains (2.35e) (:582-617)
  double Px_s[N];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    double acc = 0;
    NMPC_UNROLL
    for(int q = 0; q < N; q++)
    {
      acc += P[a + q * N] * r.x_bar[q];
    }
    Px_s[a] = acc - s[a];
  }
  double k[M > 0 ? M : 1], K[M * N > 0 ? M * N : 1];
  if constexpr(M > 0)
  {
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < N; q++)
      {
        acc += r.Bm[q + a * N] * Px_s[q];
      }
      k[a] = acc + r.Lu_t[a];
    }
    NMPC_UNROLL
    for(int c = 0; c < N; c++)
    {
      NMPC_UNROLL
      for(int a = 0; a < M; a++)
      {
        K[a + c * M] = r.H[c + a * N];
      }
    }
    Ldlt<M> ldlt;
    if(ldlt.compute(r.Gm))
    {
      ldlt.solveInPlace(k);
      NMPC_UNROLL
      for(int c = 0; c < N; c++)
      {
        ldlt.solveInPlace(K + c * M);
      }
    }
    else
    {
      if(buf.break_if_llt_fails)
      {
        return false;
      }
      fullPivLuSolveInPlace<M>(r.Gm, k);
      for(int c = 0; c < N; c++)
      {
        fullPivLuSolveInPlace<M>(r.Gm, K + c * M);
      }
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      k[a] = -1 * k[a];
    }
    NMPC_UNROLL
    for(int e = 0; e < M * N; e++)
    {
      K[e] = -1 * K[e];
    }
  }

  // post-process (2.35a) (:620-631)
  double s_new[N], P_new[N * N];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    double at_ = 0, hk = 0;
    NMPC_UNROLL
    for(int q = 0; q < N; q++)
    {
      at_ += r.A[q + a * N] * (-1 * Px_s[q]);
    }
    NMPC_UNROLL
    for(int q = 0; q < M; q++)
    {
      hk += r.H[a + q * N] * k[q];
    }
    s_new[a] = (at_ - r.Lx_t[a]) - hk;
  }
  double KtG[N * M > 0 ? N * M : 1];
  NMPC_UNROLL
  for(int c = 0; c < M; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < M; q++)
      {
        acc += K[q + a * M] * r.Gm[q + c * M];
      }
      KtG[a + c * N] = acc;
    }
  }
  NMPC_UNROLL
  for(int c = 0; c < N; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      double acc = 0;
      NMPC_UNROLL
      for(int q = 0; q < M; q++)
      {
        acc += KtG[a + q * N] * K[q + c * M];
      }
      P_new[a + c * N] = r.F[a + c * N] - acc;
    }
  }
  NMPC_UNROLL
  for(int c = 0; c < N; c++)
  {
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      P[a + c * N] = 0.5 * (P_new[a + c * N] + P_new[c + a * N]); // enforce symmetric (:627-629)
    }
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    s[a] = s_new[a];
  }

  // save gains (:634-637)
  NMPC_UNROLL
  for(int a = 0; a < M; a++)
  {
    nan = nan || bad(k[a]);
    buf.gain[at(buf, i, GL::k + a, GL::kStride, b)] = k[a];
  }
  NMPC_UNROLL
  for(int e = 0; e < M * N; e++)
  {
    nan = nan || bad(K[e]);
    buf.gain[at(buf, i, GL::K + e, GL::kStride, b)] = K[e];
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    nan = nan || bad(s[a]);
    buf.gain[at(buf, i, GL::S + a, GL::kStride, b)] = s[a];
  }
  NMPC_UNROLL
  for(int e = 0; e < N * N; e++)
  {
    nan = nan || bad(P[e]);
    buf.gain[at(buf, i, GL::P + e, GL::kStride, b)] = P[e];
  }
  return true;
}

/** One step of the forward recursion (FmpcSolver.hpp:676-685): du_i (2.36) and dx_{i+1} (2.26b) from dx_i. */
template<int N, int M>
__device__ __forceinline__ void forwardStep(const FmpcBuffers & buf, int i, int b, const ForwardRecord<N, M> & r, double * dx)
{
  double du[M > 0 ? M : 1];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    buf.dx[at(buf, i, a, N, b)] = dx[a];
  }
  NMPC_UNROLL
  for(int a = 0; a < M; a++)
  {
    double acc = 0;
    NMPC_UNROLL
    for(int q = 0; q < N; q++)
    {
      acc += r.K[a + q * M] * dx[q];
    }
    du[a] = acc + r.k[a];
    buf.du[at(buf, i, a, M, b)] = du[a];
  }
  double nx[N];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    double ax = 0, bu = 0;
    NMPC_UNROLL
    for(int q = 0; q < N; q++)
    {
      ax += r.A[a + q * N] * dx[q];
    }
    NMPC_UNROLL
    for(int q = 0; q < M; q++)
    {
      bu += r.Bm[a + q * N] * du[q];
    }
    nx[a] = (ax + bu) + r.x_bar[a];
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    dx[a] = nx[a];
  }
}
} // namespace fmpc

/** KKT-error test (FmpcSolver.hpp:443-449), backward pass (:522-665) and the recursion of the forward pass (:667-687: dx, du;
    dlambda of (2.33) depends on dx_i only and is left to fmpc_delta_kernel) of one instance per lane.  Both recursions run on a
    ring of register sets, each loaded a step before it is used. */
template<int N, int M>
__global__ void __launch_bounds__(64) fmpc_riccati_kernel(FmpcBuffers buf, int iter)
{
  using GL = fmpc::GainLayout<N, M>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= buf.B || buf.status[b] != fmpc::kStatusContinued)
  {
    return;
  }
  const int T = buf.T;
  {
    double kkt_error = 0;
    for(int i = 0; i <= T; i++)
    {
      kkt_error += buf.part[fmpc::at(buf, i, 0, fmpc::kPartSlots, b)];
    }
    kkt_error = sqrt(kkt_error);
    buf.trace[(static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE + NMPC_HIP_FMPC_TRACE_KKT_ERROR] =
        kkt_error;
    if(kkt_error <= buf.kkt_error_thre)
    {
      buf.status[b] = 1; // Status::Succeeded
      return;
    }
  }

  // ---- backward pass
  double s[N], P[N * N];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    s[a] = buf.gain[fmpc::at(buf, T, GL::S + a, GL::kStride, b)];
  }
  NMPC_UNROLL
  for(int e = 0; e < N * N; e++)
  {
    P[e] = buf.gain[fmpc::at(buf, T, GL::P + e, GL::kStride, b)];
  }
  bool nan = false;
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    nan = nan || fmpc::bad(s[a]);
  }
  bool llt_failed = false;
  {
    // ring of kBackwardDepth register sets: the record of step i - (kBackwardDepth - 1) is requested while step i is computed
    constexpr int D = fmpc::kBackwardDepth;
    fmpc::BackwardRecord<N, M> r[D];
    int i = T - 1;
    NMPC_UNROLL
    for(int d = 0; d < D - 1; d++)
    {
      if(i - d >= 0)
      {
        r[d].load(buf, i - d, b);
      }
    }
    while(i >= 0 && !llt_failed)
    {
      NMPC_UNROLL
      for(int d = 0; d < D; d++)
      {
        if(i >= 0 && !llt_failed)
        {
          if(i - (D - 1) >= 0)
          {
            r[(d + D - 1) % D].load(buf, i - (D - 1), b);
          }
          llt_failed = !fmpc::backwardStep<N, M>(buf, i, b, r[d], s, P, nan);
          i--;
        }
      }
    }
  }
  if(llt_failed || (buf.check_nan && (nan || (buf.flags[b] & 1)))) // :640-653
  {
    buf.status[b] = 3; // Status::ErrorInBackward
    return;
  }

  // ---- forward pass, the recursion over the timesteps (:669-687)
  double dx[N];
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    dx[a] = buf.x0[static_cast<size_t>(a) * buf.B + b] - buf.x[fmpc::at(buf, 0, a, N, b)];
  }
  {
    constexpr int D = fmpc::kForwardDepth;
    fmpc::ForwardRecord<N, M> r[D];
    int i = 0;
    NMPC_UNROLL
    for(int d = 0; d < D - 1; d++)
    {
      if(d < T)
      {
        r[d].load(buf, d, b);
      }
    }
    while(i < T)
    {
      NMPC_UNROLL
      for(int d = 0; d < D; d++)
      {
        if(i < T)
        {
          if(i + (D - 1) < T)
          {
            r[(d + D - 1) % D].load(buf, i + (D - 1), b);
          }
          fmpc::forwardStep<N, M>(buf, i, b, r[d], dx);
          i++;
        }
      }
    }
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    buf.dx[fmpc::at(buf, T, a, N, b)] = dx[a];
  }
}

/** The Riccati kernel for N <= 4 states and one input on the fp64 matrix cores, SIXTEEN lanes per instance.
    A workgroup is 16 instances x 4 wavefronts, a wavefront owns four instances; lane l holds entry (row l / 16, column l % 4)
    of the 4 x 4 matrices of instance (l / 4) % 4 — the C / D operand layout of v_mfma_f64_4x4x4_4b_f64, in which the
    instruction computes mma(X, Y, C) = X^T Y + C on matrices held one entry per lane (ddp_kernels_quad.hpp; measured in
    scripts/ubench_mfma_f64_4x4.hip), summing k ascending as an FMA chain from C.  One backward step (FmpcSolver.hpp:576-637) is
    seven such instructions:

        PA = mma(P, A, 0)                 (A^T P)^T                                   R  = mma(P, [B | x_bar], [0 | -s])  [P B | P x_bar - s]
        F  = mma(PA, A, Qxx~)             (2.35b)                                     S  = mma(A, R, [Qxu~ | Lx~])        [H | A^T (P x_bar - s) + Lx~]
        FT = mma(A, PA, Qxx~^T)           the same entries transposed                 Wq = mma([B B B B], R, [Quu~ Lu~])  every row [G, B^T (P x_bar - s) + Lu~]
        HA = mma(bcast(P B), A, Qxu~^T)   H[col] in every row

    then, per lane, k = -k_rhs / G, K[row] = -H[row] / G, K[col] = -H[col] / G (the pseudo-inverse rule of Eigen's LDLT solve
    for the 1 x 1 block), s' = -(A^T (P x_bar - s) + Lx~) - H k and P' = 1/2 ((F - (K[row] G) K[col]) + (FT - (K[col] G)
    K[row])) — entry (row, col) and entry (col, row) of F - K^T G K exactly as the reference's two triangles (:620-629).  The
    forward recursion (:676-685) is two more: A dx and K dx.  Against the one-lane-per-instance kernel a step issues ~12 loads
    and ~60 instructions per lane instead of ~75 and ~500, and 4096 instances are 1024 wavefronts (every SIMD of the chip)
    instead of 64.  Sums that the lane kernel starts from zero and adds to a coefficient afterwards start from the coefficient
    here (MFMA accumulates from C); H is formed as A^T (P B) + Qxu~ rather than (A^T P) B + Qxu~: results agree with the lane
    kernel and the oracle to rounding, not bit for bit (tests/test_gpu_fmpc.py compares both against the oracle). */
template<int N>
__global__ void __launch_bounds__(256) fmpc_riccati_quad_kernel(FmpcBuffers buf, int iter)
{
  static_assert(N >= 1 && N <= 4, "[FMPC] the quad Riccati kernel handles up to four states");
  constexpr int M = 1;
  using CL = fmpc::CoefLayout<N, M>;
  using GL = fmpc::GainLayout<N, M>;
  constexpr int kS = fmpc::kStageSteps;
  constexpr int kRecB = CL::kStride | 1; // odd record widths: the 16 instances of a row slot hit 16 different banks
  constexpr int kFwdCoef = CL::XBAR + N; // A, B, x_bar: elements [0, kFwdCoef) of the coefficient record
  constexpr int kFwdGain = GL::K + M * N; // k, K: elements [0, kFwdGain) of the gain record
  constexpr int kRecF = (kFwdCoef + kFwdGain) | 1;
  constexpr int kSlotDoubles = kS * 16 * (kRecB > kRecF ? kRecB : kRecF);
  constexpr int kRecG = (GL::kStride + 2) | 1; // gain record + two spare slots
  constexpr int kRecX = (N + M + 2) | 1; // dx, du + two spare slots
  __shared__ double stage_lds[2 * kSlotDoubles];
  __shared__ double gain_lds[kS * 16 * kRecG]; // what the chunk's steps produce (backward: gains, forward: dx, du), before it goes to HBM
  __shared__ double sh_kkt[16][17];
  __shared__ int sh_live[16];

  const int wl = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = wl >> 4, blk = (wl >> 2) & 3, col = wl & 3;
  const int inst = wave * 4 + blk;
  const int b_raw = blockIdx.x * 16 + inst;
  const int b = b_raw < buf.B ? b_raw : buf.B - 1; // lanes beyond the batch mirror the last instance and store nothing
  const bool head = row == 0 && col == 0; // the lane that speaks for the instance
  const bool c0 = col == 0, c1 = col == 1;
  const bool rv = row < N, cv = col < N, valid = rv && cv;
  const int rc = rv ? row : N - 1, cc = cv ? col : N - 1;
  const int T = buf.T;
  const size_t Bz = static_cast<size_t>(buf.B);
  bool live = b_raw < buf.B && buf.status[b] == fmpc::kStatusContinued;
  // staging role of this thread: element slot t_slot (+ 16 q) of instance t_inst of the workgroup
  const int t_inst = threadIdx.x & 15, t_slot = threadIdx.x >> 4;
  const int b_stage_raw = blockIdx.x * 16 + t_inst;
  const int b_stage = b_stage_raw < buf.B ? b_stage_raw : buf.B - 1;
  const size_t lane_stage = static_cast<size_t>(t_slot) * Bz + b_stage; // + (row index) * B = element index of this thread's load

  auto mma = [](double x, double y, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c, 0, 0, 0); };
  auto bcast0 = [](double v) { // entry of column 0 of this lane's row (same quad) in all four lanes of the quad
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x00, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x00, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  auto bcast1 = [](double v) { // quad_perm:[1,1,1,1]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };

  // ---- KKT-error test (FmpcSolver.hpp:443-449): sixteen partial sums per instance (timesteps t_slot, t_slot + 16, ...; every
  // 16-lane group reads one full line per row), added up by the head lane in slot order
  {
    // (eight terms requested before the first is added, the additions in the loop's order: one term per trip was a round trip to
    // L2 per term, thirteen in a row in front of the recursion for T = 200)
    double acc = 0;
    for(int i0 = t_slot; i0 <= T; i0 += 16 * 8)
    {
      double term[8];
      NMPC_UNROLL
      for(int k = 0; k < 8; k++)
      {
        const int i = i0 + 16 * k;
        term[k] = buf.part[fmpc::at(buf, i <= T ? i : T, 0, fmpc::kPartSlots, b_stage)];
      }
      NMPC_UNROLL
      for(int k = 0; k < 8; k++)
      {
        if(i0 + 16 * k <= T)
        {
          acc += term[k];
        }
      }
    }
    sh_kkt[t_inst][t_slot] = acc;
  }
  syncThreadsFuzzed(__LINE__);
  if(head)
  {
    double kkt_error = 0;
    NMPC_UNROLL
    for(int q = 0; q < 16; q++)
    {
      kkt_error += sh_kkt[inst][q];
    }
    kkt_error = sqrt(kkt_error);
    if(live)
    {
      buf.trace[(static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE + NMPC_HIP_FMPC_TRACE_KKT_ERROR] =
          kkt_error;
      if(kkt_error <= buf.kkt_error_thre)
      {
        buf.status[b] = 1; // Status::Succeeded
      }
    }
    sh_kkt[inst][16] = kkt_error;
  }
  syncThreadsFuzzed(__LINE__);
  live = live && !(sh_kkt[inst][16] <= buf.kkt_error_thre);
  if(head)
  {
    sh_live[inst] = live ? 1 : 0;
  }
  fuzzSched(__LINE__);
  const int any_live = __syncthreads_or(live ? 1 : 0);
  fuzzSched(__LINE__);
  if(any_live == 0)
  {
    return; // workgroup-uniform: none of the sixteen instances is live
  }

  // ---- operand staging.  In the [timestep][element][instance] arrays the four instances of a wavefront are 32 bytes of every
  // row: operands fetched per lane would touch sixteen cache lines per load and use a quarter of each (measured: the kernel is
  // then bound by the line rate of the vector L1, 0.95 us per step).  Instead the sixteen instances of the WORKGROUP — one full
  // 128-byte line per row — are fetched cooperatively for kStageSteps timesteps at a time (thread = (element slot, instance):
  // every 16-lane group reads one whole line), parked in registers while the previous chunk is being consumed, written to LDS
  // as per-(timestep, instance) records and read from there by the lanes that need them.
  /** Elements [0, NR) of timesteps i0, i0 + dir, ..., i0 + (kS - 1) dir of this thread's instance: request into v. */
  auto request = [&](auto nr_tag, const double * src, int stride, int i0, int dir, double * v) {
    constexpr int NR = decltype(nr_tag)::value;
    constexpr int kQ = (NR + 15) / 16;
    NMPC_UNROLL
    for(int st = 0; st < kS; st++)
    {
      const int step = i0 + dir * st; // wavefront-uniform
      if(step >= 0 && step < T)
      {
        const double * rowp = src + (static_cast<size_t>(step) * stride) * Bz + lane_stage;
        NMPC_UNROLL
        for(int q = 0; q < kQ; q++)
        {
          if(16 * q + 15 < NR || t_slot < NR - 16 * q)
          {
            v[st * kQ + q] = rowp[static_cast<size_t>(16 * q) * Bz];
          }
        }
      }
    }
  };
  /** ... and park them in LDS slot `slot` as records of width W, at offset `off` of each record. */
  auto commit = [&](auto nr_tag, int slot, int W, int off, const double * v) {
    constexpr int NR = decltype(nr_tag)::value;
    constexpr int kQ = (NR + 15) / 16;
    double * base = stage_lds + static_cast<size_t>(slot) * kSlotDoubles + t_inst * W + off + t_slot;
    NMPC_UNROLL
    for(int st = 0; st < kS; st++)
    {
      NMPC_UNROLL
      for(int q = 0; q < kQ; q++)
      {
        if(16 * q + 15 < NR || t_slot < NR - 16 * q)
        {
          base[st * 16 * W + 16 * q] = v[st * kQ + q];
        }
      }
    }
  };
  using TagB = std::integral_constant<int, CL::kStride>;
  using TagFc = std::integral_constant<int, kFwdCoef>;
  using TagFg = std::integral_constant<int, kFwdGain>;

  // ---- backward pass
#ifdef NMPC_AMD_FMPC_PROFILE
  const unsigned long long tick0 = wall_clock64();
#endif
  double P = valid ? buf.gain[fmpc::at(buf, T, GL::P + rc + cc * N, GL::kStride, b)] : 0.0;
  double s_row = rv ? buf.gain[fmpc::at(buf, T, GL::S + rc, GL::kStride, b)] : 0.0; // s[row], kept by every lane of the row
  bool nan = fmpc::bad(s_row);

  // What a lane writes per step: its entry of P, and one more value by role — s[row] on the diagonal lanes, K[col] on the
  // lanes one row below the diagonal (cyclically), k on lane (2, 0) — every value is present on every lane of its row / column,
  // so any lane of the right row / column can write it.  Two stores per step and no branch inside the recursion loop.
  const bool role_s = (row == col) && rv, role_K = (row == ((col + 1) & 3)) && cv, role_k = (row == 2 && col == 0);
  // (lanes without a value of their own write the record's spare slots)
  const int e1 = valid ? GL::P + rc + cc * N : GL::kStride;
  const int e2 = role_s ? GL::S + rc : (role_K ? GL::K + cc : (role_k ? GL::k : GL::kStride + 1));
  double * const gP = gain_lds + inst * kRecG + e1;
  double * const g2 = gain_lds + inst * kRecG + e2;

  // per-lane offsets into a staged record
  const int oA = CL::A + rc + cc * N, oQ = CL::QXX + rc + cc * N, oQT = CL::QXX + cc + rc * N, oB = CL::B + rc, oX = CL::XBAR + rc;
  const int oCM = c0 ? CL::QUU : CL::LUT, oLM = c0 ? CL::QXU + rc : CL::LXT + rc, oQR = CL::QXU + cc;
  struct Operands
  {
    double A, Qxx, QxxT, Bv, Y, CM, LM, QxuRow;
  };
  auto loadOperands = [&](const double * rec, Operands & o) {
    const double a = rec[oA], q = rec[oQ], qt = rec[oQT], bv = rec[oB], xb = rec[oX], cm = rec[oCM], lm = rec[oLM], qr = rec[oQR];
    o.A = valid ? a : 0.0;
    o.Qxx = valid ? q : 0.0;
    o.QxxT = valid ? qt : 0.0;
    o.Bv = rv ? bv : 0.0; // B[row] in every column
    o.Y = rv ? (c0 ? bv : (c1 ? xb : 0.0)) : 0.0; // [B | x_bar | 0 | 0]
    o.CM = (c0 || c1) ? cm : 0.0; // [Quu~, Lu~, 0, 0] in every row
    o.LM = (rv && (c0 || c1)) ? lm : 0.0; // [Qxu~ | Lx~ | 0 | 0]
    o.QxuRow = cv ? qr : 0.0; // Qxu~^T in every row
  };
  auto backwardStep = [&](int st, const Operands & o) {
    const double PA = mma(P, o.A, 0.0);
    const double R = mma(P, o.Y, c1 ? -1 * s_row : 0.0);
    const double F = mma(PA, o.A, o.Qxx);
    const double FT = mma(o.A, PA, o.QxxT);
    const double S = mma(o.A, R, o.LM);
    const double Wq = mma(o.Bv, R, o.CM);
    const double HA = mma(bcast0(R), o.A, o.QxuRow);
    const double G = bcast0(Wq), k_rhs = bcast1(Wq);
    const double Hr = bcast0(S), Qxl = bcast1(S);
    // Eigen's LDLT solve of the 1 x 1 system (pseudo-inverse of D), then (2.35e)
    const bool pivot = fabs(G) > DBL_MIN;
    const double k = -1 * (pivot ? k_rhs / G : 0.0);
    const double Kc = -1 * (pivot ? HA / G : 0.0);
    const double Kr = -1 * (pivot ? Hr / G : 0.0);
    const double s_new = (-1 * Qxl) - Hr * k; // (2.35a)
    const double Pn = F - (Kr * G) * Kc;
    const double PnT = FT - (Kc * G) * Kr;
    P = 0.5 * (Pn + PnT); // enforce symmetric (:627-629)
    s_row = s_new;
    nan = nan || fmpc::bad(k) || fmpc::bad(Kc) || fmpc::bad(s_new) || fmpc::bad(P);
    const double v2 = role_s ? s_new : (role_K ? Kc : (role_k ? k : P));
    gP[st * 16 * kRecG] = P; // parked in LDS; flushGains writes the chunk out in whole cache lines
    g2[st * 16 * kRecG] = v2;
  };
  /** The gains of the chunk that started at timestep i0: LDS -> HBM, thread = (element slot, instance), a full line per row. */
  auto flushGains = [&](int i0) {
    if(sh_live[t_inst] != 0)
    {
      NMPC_UNROLL
      for(int st = 0; st < kS; st++)
      {
        const int step = i0 - st;
        if(step >= 0)
        {
          double * rowp = buf.gain + (static_cast<size_t>(step) * GL::kStride) * Bz + lane_stage;
          NMPC_UNROLL
          for(int q = 0; q < (GL::kStride + 15) / 16; q++)
          {
            if(16 * q + 15 < GL::kStride || t_slot < GL::kStride - 16 * q)
            {
              rowp[static_cast<size_t>(16 * q) * Bz] = gain_lds[(st * 16 + t_inst) * kRecG + 16 * q + t_slot];
            }
          }
        }
      }
    }
  };
  {
    constexpr int kQB = (CL::kStride + 15) / 16;
    double vb[kS * kQB];
    request(TagB(), buf.coef, CL::kStride, T - 1, -1, vb);
    commit(TagB(), 0, kRecB, 0, vb);
    syncThreadsFuzzed(__LINE__);
    int slot = 0;
#ifdef NMPC_AMD_FMPC_PROFILE2
    unsigned long long pa = 0, pb = 0, pc = 0, pd = 0, t0_, t1_, t2_, t3_, t4_;
#endif
    for(int i0 = T - 1; i0 >= 0; i0 -= kS)
    {
#ifdef NMPC_AMD_FMPC_PROFILE2
      t0_ = wall_clock64();
#endif
      if(i0 - kS >= 0)
      {
        request(TagB(), buf.coef, CL::kStride, i0 - kS, -1, vb); // the next chunk travels while this one is consumed
      }
#ifdef NMPC_AMD_FMPC_PROFILE2
      t1_ = wall_clock64();
#endif
      const double * recs = stage_lds + static_cast<size_t>(slot) * kSlotDoubles + inst * kRecB;
      Operands o[2];
      loadOperands(recs, o[0]);
      NMPC_UNROLL
      for(int st = 0; st < kS; st++)
      {
        if(i0 - st >= 0)
        {
          if(st + 1 < kS)
          {
            loadOperands(recs + (st + 1) * 16 * kRecB, o[(st + 1) & 1]);
          }
          backwardStep(st, o[st & 1]);
        }
      }
#ifdef NMPC_AMD_FMPC_PROFILE2
      t2_ = wall_clock64();
#endif
      syncThreadsFuzzed(__LINE__); // every wavefront's gains of this chunk are in LDS
      if(i0 - kS >= 0)
      {
        commit(TagB(), slot ^ 1, kRecB, 0, vb);
      }
#ifdef NMPC_AMD_FMPC_PROFILE2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t3_ = wall_clock64();
#endif
      flushGains(i0);
      syncThreadsFuzzed(__LINE__);
#ifdef NMPC_AMD_FMPC_PROFILE2
      t4_ = wall_clock64();
      pa += t1_ - t0_;
      pb += t2_ - t1_;
      pc += t3_ - t2_;
      pd += t4_ - t3_;
#endif
      slot ^= 1;
    }
#ifdef NMPC_AMD_FMPC_PROFILE2
    if(head && b_raw < buf.B)
    {
      buf.trace[(static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE + 2] = static_cast<double>(pa);
      buf.merit[0 * buf.B + b] = static_cast<double>(pb);
      buf.merit[1 * buf.B + b] = static_cast<double>(pc);
      buf.merit[2 * buf.B + b] = static_cast<double>(pd);
    }
#endif
  }
  // verdict of the backward pass per instance (:640-653): OR over the sixteen lanes of the instance
  {
    int bad_any = nan ? 1 : 0;
    bad_any |= __shfl_xor(bad_any, 1);
    bad_any |= __shfl_xor(bad_any, 2);
    bad_any |= __shfl_xor(bad_any, 16);
    bad_any |= __shfl_xor(bad_any, 32);
    const bool failed = buf.check_nan && (bad_any != 0 || (buf.flags[b] & 1));
    if(live && failed && head)
    {
      buf.status[b] = 3; // Status::ErrorInBackward
    }
    live = live && !failed;
    if(head)
    {
      sh_live[inst] = live ? 1 : 0;
    }
  }
  // (no early exit from here on: every wavefront of the workgroup takes part in the staging barriers of the forward pass,
  // and the gains the other wavefronts wrote above are read below: make them visible first)
  __threadfence();
  syncThreadsFuzzed(__LINE__);

  // ---- forward pass, the recursion over the timesteps (:669-687); dx[row] is kept by every lane of the row
#ifdef NMPC_AMD_FMPC_PROFILE
  const unsigned long long tick1 = wall_clock64();
#endif
  double dx_row = rv ? buf.x0[static_cast<size_t>(rc) * Bz + b] - buf.x[fmpc::at(buf, 0, rc, N, b)] : 0.0;
  // one value per lane and step by role, parked in LDS: dx[row] on the diagonal lanes, du on lane (1, 0)
  const bool role_dx = (row == col) && rv, role_du = (row == 1 && col == 0);
  double * const fP = gain_lds + inst * kRecX + (role_dx ? rc : (role_du ? N : N + M));
  const int oAT = CL::A + cc + rc * N; // entry (col, row): the A operand of mma is used transposed
  const int oKx = kFwdCoef + GL::K + rc, ok = kFwdCoef + GL::k;
  struct ForwardOperands
  {
    double AT, Kx, k, Bv, xb;
  };
  auto loadForward = [&](const double * rec, ForwardOperands & o) {
    const double at_ = rec[oAT], kx = rec[oKx], bv = rec[oB], xb = rec[oX];
    o.k = rec[ok];
    o.AT = valid ? at_ : 0.0;
    o.Kx = rv ? kx : 0.0; // K[row] in every column
    o.Bv = rv ? bv : 0.0;
    o.xb = rv ? xb : 0.0;
  };
  auto forwardStep = [&](int st, const ForwardOperands & o) {
    const double Y = c0 ? dx_row : 0.0;
    const double ax = bcast0(mma(o.AT, Y, 0.0)); // (A dx)[row]
    const double du = bcast0(mma(o.Kx, Y, 0.0)) + o.k; // (2.36), the same in every row
    fP[st * 16 * kRecX] = role_du ? du : dx_row;
    const double nx = (ax + o.Bv * du) + o.xb; // (2.26b)
    dx_row = rv ? nx : 0.0;
  };
  {
    constexpr int kQC = (kFwdCoef + 15) / 16, kQG = (kFwdGain + 15) / 16;
    double vc[kS * kQC], vg[kS * kQG];
    request(TagFc(), buf.coef, CL::kStride, 0, 1, vc);
    request(TagFg(), buf.gain, GL::kStride, 0, 1, vg);
    commit(TagFc(), 0, kRecF, 0, vc);
    commit(TagFg(), 0, kRecF, kFwdCoef, vg);
    syncThreadsFuzzed(__LINE__);
    int slot = 0;
    for(int i0 = 0; i0 < T; i0 += kS)
    {
      if(i0 + kS < T)
      {
        request(TagFc(), buf.coef, CL::kStride, i0 + kS, 1, vc);
        request(TagFg(), buf.gain, GL::kStride, i0 + kS, 1, vg);
      }
      const double * recs = stage_lds + static_cast<size_t>(slot) * kSlotDoubles + inst * kRecF;
      ForwardOperands o[2];
      loadForward(recs, o[0]);
      NMPC_UNROLL
      for(int st = 0; st < kS; st++)
      {
        if(i0 + st < T)
        {
          if(st + 1 < kS)
          {
            loadForward(recs + (st + 1) * 16 * kRecF, o[(st + 1) & 1]);
          }
          forwardStep(st, o[st & 1]);
        }
      }
      syncThreadsFuzzed(__LINE__);
      if(i0 + kS < T)
      {
        commit(TagFc(), slot ^ 1, kRecF, 0, vc);
        commit(TagFg(), slot ^ 1, kRecF, kFwdCoef, vg);
      }
      if(sh_live[t_inst] != 0 && t_slot < N + M) // dx, du of the chunk: LDS -> HBM in whole lines
      {
        NMPC_UNROLL
        for(int st = 0; st < kS; st++)
        {
          const int step = i0 + st;
          if(step < T)
          {
            const double v = gain_lds[(st * 16 + t_inst) * kRecX + t_slot];
            if(t_slot < N)
            {
              buf.dx[(static_cast<size_t>(step) * N + t_slot) * Bz + b_stage] = v;
            }
            else
            {
              buf.du[(static_cast<size_t>(step) * M + (t_slot - N)) * Bz + b_stage] = v;
            }
          }
        }
      }
      syncThreadsFuzzed(__LINE__);
      slot ^= 1;
    }
  }
  if(live && c0 && rv)
  {
    buf.dx[fmpc::at(buf, T, rc, N, b)] = dx_row;
  }
#ifdef NMPC_AMD_FMPC_PROFILE
  if(head && b_raw < buf.B) // developer build only: 100 MHz wall-clock ticks of the two recursions in the merit slots
  {
    buf.merit[0 * buf.B + b] = static_cast<double>(tick1 - tick0);
    buf.merit[1 * buf.B + b] = static_cast<double>(wall_clock64() - tick1);
  }
#endif
}

namespace fmpc
{
/** Sink of a producer wave of fmpc_riccati_fused_kernel: the record into its LDS staging slot; A, B, x_bar also to HBM.  A producer
    works on a chunk for TWO trips of the recursion waves' chunk loop: midpoint() stands for the two workgroup barriers that end the
    first of them (all of the record's stores come after it: the slot is still being read during the first trip). */
template<int N, int M>
struct StagedCoefSink
{
  double * rec;
  const FmpcBuffers & buf;
  int b, i;
  bool on, to_hbm;
  int * nan_flag; //!< the instance's word of the workgroup: Coefficient::containsNaN of one of its records
  __device__ __forceinline__ void coef(int e, double v) const
  {
    if(on)
    {
      rec[e] = v;
      if(e < CoefLayout<N, M>::XBAR + N && to_hbm)
      {
        buf.coef[at(buf, i, e, CoefLayout<N, M>::kStride, b)] = v;
      }
    }
  }
  __device__ __forceinline__ void midpoint() const
  {
    syncThreadsFuzzed(__LINE__);
    syncThreadsFuzzed(__LINE__);
  }
  __device__ __forceinline__ void terminal(int, double) const {}
  __device__ __forceinline__ void kkt(double) const {}
  __device__ __forceinline__ void nanFlag() const
  {
    if(on)
    {
      *nan_flag = 1; // (every writer stores 1; read behind the backward pass, a workgroup barrier later)
    }
  }
};
} // namespace fmpc

/** fmpc_riccati_quad_kernel with the coefficient records never in HBM (FmpcSolver.hpp:395-436, :523-665): a FIFTH
    wavefront of the workgroup (lane = (timestep of the chunk, instance)) computes the records of the next chunk from the variables
    (fmpc::coefficients, the body of fmpc_coeff_kernel) into the staging slot the four recursion wavefronts read, while they consume
    the current one.  In front of it runs fmpc_coeff_kernel<Problem, false> (KKT-error terms, NaN flag, terminal record: the KKT test
    comes BEFORE the backward pass, :443-449, and needs every timestep's terms).  A, B, x_bar are still written to HBM once (the forward
    sweep and the line search read them).  Everything else is the quad kernel's text. */
template<class Problem>
__global__ void __launch_bounds__(384) fmpc_riccati_fused_kernel(FmpcBuffers buf, int iter)
{
  constexpr int N = Problem::kStateDim;
  static_assert(N >= 1 && N <= 4 && Problem::kInputDimMax == 1, "[FMPC] the fused Riccati kernel handles up to four states and one input");
  constexpr int M = 1;
  using CL = fmpc::CoefLayout<N, M>;
  using GL = fmpc::GainLayout<N, M>;
  constexpr int kS = fmpc::kStageSteps;
  constexpr int kRecB = CL::kStride | 1; // odd record widths: the 16 instances of a row slot hit 16 different banks
  constexpr int kFwdCoef = CL::XBAR + N; // A, B, x_bar: elements [0, kFwdCoef) of the coefficient record
  constexpr int kFwdGain = GL::K + M * N; // k, K: elements [0, kFwdGain) of the gain record
  constexpr int kRecF = (kFwdCoef + kFwdGain) | 1;
  constexpr int kSlotDoubles = kS * 16 * (kRecB > kRecF ? kRecB : kRecF);
  // the forward pass walks the horizon in chunks of kF steps through TWO slots laid over the backward pass's three: a chunk costs one
  // (partly hidden) round trip to HBM whatever its length
#ifndef NMPC_AMD_FMPC_FWD_STEPS
#  define NMPC_AMD_FMPC_FWD_STEPS (2 * NMPC_AMD_FMPC_STAGE_STEPS)
#endif
  constexpr int kF = NMPC_AMD_FMPC_FWD_STEPS;
  constexpr int kFwdSlotDoubles = 3 * kSlotDoubles / 2;
  static_assert(kF * 16 * kRecF <= kFwdSlotDoubles, "[FMPC] a forward chunk fits half of the staging area");
  constexpr int kRecG = (GL::kStride + 2) | 1; // gain record + two spare slots
  constexpr int kRecX = (N + M + 2) | 1; // dx, du + two spare slots
  __shared__ double stage_lds[3 * kSlotDoubles]; // chunk k of the backward pass in slot k % 3 (the forward pass uses two)
  constexpr int kGainLds = kS * 16 * kRecG > 2 * kF * 16 * kRecX ? kS * 16 * kRecG : 2 * kF * 16 * kRecX;
  __shared__ double gain_lds[kGainLds]; // what a chunk's steps produce (backward: gains; forward: dx, du, two areas used alternately), before it goes to HBM
  __shared__ double sh_kkt[16][17];
  __shared__ int sh_live[16];
  __shared__ int sh_coef_nan[16]; // a record of the instance holds a NaN / Inf (the producer waves' finding)

  const int wl = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = wl >> 4, blk = (wl >> 2) & 3, col = wl & 3;
  const bool producer = wave >= 4; // wavefronts 4 and 5: lane = (timestep of the chunk, instance), compute the coefficient records
  const int inst = producer ? (wl & 15) : wave * 4 + blk;
  const int b_raw = blockIdx.x * 16 + inst;
  const int b = b_raw < buf.B ? b_raw : buf.B - 1; // lanes beyond the batch mirror the last instance and store nothing
  const bool head = !producer && row == 0 && col == 0; // the lane that speaks for the instance
  const bool c0 = col == 0, c1 = col == 1;
  const bool rv = row < N, cv = col < N, valid = rv && cv;
  const int rc = rv ? row : N - 1, cc = cv ? col : N - 1;
  const int T = buf.T;
  const size_t Bz = static_cast<size_t>(buf.B);
  bool live = b_raw < buf.B && buf.status[b] == fmpc::kStatusContinued;
  // staging role of this thread: element slot t_slot (+ 16 q) of instance t_inst of the workgroup
  const int t_inst = threadIdx.x & 15, t_slot = threadIdx.x >> 4;
  const int b_stage_raw = blockIdx.x * 16 + t_inst;
  const int b_stage = b_stage_raw < buf.B ? b_stage_raw : buf.B - 1;
  const size_t lane_stage = static_cast<size_t>(t_slot) * Bz + b_stage; // + (row index) * B = element index of this thread's load

  auto mma = [](double x, double y, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c, 0, 0, 0); };
  auto bcast0 = [](double v) { // entry of column 0 of this lane's row (same quad) in all four lanes of the quad
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x00, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x00, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  auto bcast1 = [](double v) { // quad_perm:[1,1,1,1]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };

  // ---- KKT-error test (FmpcSolver.hpp:443-449): sixteen partial sums per instance (timesteps t_slot, t_slot + 16, ...; every
  // 16-lane group reads one full line per row), added up by the head lane in slot order
  if(!producer)
  {
    // (eight terms requested before the first is added, the additions in the loop's order: one term per trip was a round trip to
    // L2 per term, thirteen in a row in front of the recursion for T = 200)
    double acc = 0;
    for(int i0 = t_slot; i0 <= T; i0 += 16 * 8)
    {
      double term[8];
      NMPC_UNROLL
      for(int k = 0; k < 8; k++)
      {
        const int i = i0 + 16 * k;
        term[k] = buf.part[fmpc::at(buf, i <= T ? i : T, 0, fmpc::kPartSlots, b_stage)];
      }
      NMPC_UNROLL
      for(int k = 0; k < 8; k++)
      {
        if(i0 + 16 * k <= T)
        {
          acc += term[k];
        }
      }
    }
    sh_kkt[t_inst][t_slot] = acc;
  }
  else if(threadIdx.x < 256 + 16)
  {
    sh_coef_nan[threadIdx.x - 256] = 0;
  }
  syncThreadsFuzzed(__LINE__);
  if(head)
  {
    double kkt_error = 0;
    NMPC_UNROLL
    for(int q = 0; q < 16; q++)
    {
      kkt_error += sh_kkt[inst][q];
    }
    kkt_error = sqrt(kkt_error);
    if(live)
    {
      buf.trace[(static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE + NMPC_HIP_FMPC_TRACE_KKT_ERROR] =
          kkt_error;
      if(kkt_error <= buf.kkt_error_thre)
      {
        buf.status[b] = 1; // Status::Succeeded
      }
    }
    sh_kkt[inst][16] = kkt_error;
  }
  syncThreadsFuzzed(__LINE__);
  live = live && !(sh_kkt[inst][16] <= buf.kkt_error_thre);
  if(head)
  {
    sh_live[inst] = live ? 1 : 0;
  }
  fuzzSched(__LINE__);
  const int any_live = __syncthreads_or(live ? 1 : 0);
  fuzzSched(__LINE__);
  if(any_live == 0)
  {
    return; // workgroup-uniform: none of the sixteen instances is live
  }

  // ---- operand staging.  In the [timestep][element][instance] arrays the four instances of a wavefront are 32 bytes of every
  // row: operands fetched per lane would touch sixteen cache lines per load and use a quarter of each (measured: the kernel is
  // then bound by the line rate of the vector L1, 0.95 us per step).  Instead the sixteen instances of the WORKGROUP — one full
  // 128-byte line per row — are fetched cooperatively for kStageSteps timesteps at a time (thread = (element slot, instance):
  // every 16-lane group reads one whole line), parked in registers while the previous chunk is being consumed, written to LDS
  // as per-(timestep, instance) records and read from there by the lanes that need them.
  /** Elements [0, NR) of timesteps i0, i0 + dir, ..., i0 + (kS - 1) dir of this thread's instance: request into v. */
  using TagB = std::integral_constant<int, CL::kStride>;
  using TagFc = std::integral_constant<int, kFwdCoef>;
  using TagFg = std::integral_constant<int, kFwdGain>;

  // ---- backward pass
#ifdef NMPC_AMD_FMPC_PROFILE
  const unsigned long long tick0 = wall_clock64();
#endif
  double P = valid ? buf.gain[fmpc::at(buf, T, GL::P + rc + cc * N, GL::kStride, b)] : 0.0;
  double s_row = rv ? buf.gain[fmpc::at(buf, T, GL::S + rc, GL::kStride, b)] : 0.0; // s[row], kept by every lane of the row
  bool nan = fmpc::bad(s_row);

  // What a lane writes per step: its entry of P, and one more value by role — s[row] on the diagonal lanes, K[col] on the
  // lanes one row below the diagonal (cyclically), k on lane (2, 0) — every value is present on every lane of its row / column,
  // so any lane of the right row / column can write it.  Two stores per step and no branch inside the recursion loop.
  const bool role_s = (row == col) && rv, role_K = (row == ((col + 1) & 3)) && cv, role_k = (row == 2 && col == 0);
  // (lanes without a value of their own write the record's spare slots)
  const int e1 = valid ? GL::P + rc + cc * N : GL::kStride;
  const int e2 = role_s ? GL::S + rc : (role_K ? GL::K + cc : (role_k ? GL::k : GL::kStride + 1));
  double * const gP = gain_lds + inst * kRecG + e1;
  double * const g2 = gain_lds + inst * kRecG + e2;

  // per-lane offsets into a staged record
  const int oA = CL::A + rc + cc * N, oQ = CL::QXX + rc + cc * N, oQT = CL::QXX + cc + rc * N, oB = CL::B + rc, oX = CL::XBAR + rc;
  const int oCM = c0 ? CL::QUU : CL::LUT, oLM = c0 ? CL::QXU + rc : CL::LXT + rc, oQR = CL::QXU + cc;
  struct Operands
  {
    double A, Qxx, QxxT, Bv, Y, CM, LM, QxuRow;
  };
  auto loadOperands = [&](const double * rec, Operands & o) {
    const double a = rec[oA], q = rec[oQ], qt = rec[oQT], bv = rec[oB], xb = rec[oX], cm = rec[oCM], lm = rec[oLM], qr = rec[oQR];
    o.A = valid ? a : 0.0;
    o.Qxx = valid ? q : 0.0;
    o.QxxT = valid ? qt : 0.0;
    o.Bv = rv ? bv : 0.0; // B[row] in every column
    o.Y = rv ? (c0 ? bv : (c1 ? xb : 0.0)) : 0.0; // [B | x_bar | 0 | 0]
    o.CM = (c0 || c1) ? cm : 0.0; // [Quu~, Lu~, 0, 0] in every row
    o.LM = (rv && (c0 || c1)) ? lm : 0.0; // [Qxu~ | Lx~ | 0 | 0]
    o.QxuRow = cv ? qr : 0.0; // Qxu~^T in every row
  };
  auto backwardStep = [&](int st, const Operands & o) {
    const double PA = mma(P, o.A, 0.0);
    const double R = mma(P, o.Y, c1 ? -1 * s_row : 0.0);
    const double F = mma(PA, o.A, o.Qxx);
    const double FT = mma(o.A, PA, o.QxxT);
    const double S = mma(o.A, R, o.LM);
    const double Wq = mma(o.Bv, R, o.CM);
    const double HA = mma(bcast0(R), o.A, o.QxuRow);
    const double G = bcast0(Wq), k_rhs = bcast1(Wq);
    const double Hr = bcast0(S), Qxl = bcast1(S);
    // Eigen's LDLT solve of the 1 x 1 system (pseudo-inverse of D), then (2.35e)
    const bool pivot = fabs(G) > DBL_MIN;
    // (Measured and not kept: one recipFast and three products instead of the three IEEE divides — 0 - 1 % on the launch, and the
    // ill-conditioned golden case fmpc_cartpole_loop_tick2 then deviates 4e-8 instead of 1e-8; with the divides this kernel returns
    // the quad kernel's bits.)
    const double k = -1 * (pivot ? k_rhs / G : 0.0);
    const double Kc = -1 * (pivot ? HA / G : 0.0);
    const double Kr = -1 * (pivot ? Hr / G : 0.0);
    const double s_new = (-1 * Qxl) - Hr * k; // (2.35a)
    const double Pn = F - (Kr * G) * Kc;
    const double PnT = FT - (Kc * G) * Kr;
    P = 0.5 * (Pn + PnT); // enforce symmetric (:627-629)
    s_row = s_new;
    nan = nan || fmpc::bad(k) || fmpc::bad(Kc) || fmpc::bad(s_new) || fmpc::bad(P);
    const double v2 = role_s ? s_new : (role_K ? Kc : (role_k ? k : P));
    gP[st * 16 * kRecG] = P; // parked in LDS; flushGains writes the chunk out in whole cache lines
    g2[st * 16 * kRecG] = v2;
  };
  /** The gains of the chunk that started at timestep i0: LDS -> HBM, thread = (element slot, instance), a full line per row. */
  auto flushGains = [&](int i0) {
    if(sh_live[t_inst] != 0)
    {
      NMPC_UNROLL
      for(int st = 0; st < kS; st++)
      {
        const int step = i0 - st;
        if(step >= 0)
        {
          double * rowp = buf.gain + (static_cast<size_t>(step) * GL::kStride) * Bz + lane_stage;
          NMPC_UNROLL
          for(int q = 0; q < (GL::kStride + 15) / 16; q++)
          {
            if(16 * q + 15 < GL::kStride || t_slot < GL::kStride - 16 * q)
            {
              rowp[static_cast<size_t>(16 * q) * Bz] = gain_lds[(st * 16 + t_inst) * kRecG + 16 * q + t_slot];
            }
          }
        }
      }
    }
  };
  // ---- the producer wave: the coefficient record of (instance p_inst, timestep i0 - p_ts) straight into the staging slot the
  // recursion waves read (what fmpc_coeff_kernel would have written to HBM and this kernel read back: 51 of the ~80 doubles a step
  // moved), one chunk ahead of them.  A, B, x_bar also go to HBM: the forward sweep below and the line search read them there.
  const int p_inst = wl & 15, p_ts = wl >> 4;
  // Chunk k (timesteps T - 1 - 4 k ... down) lives in slot k % 3.  The two producers work in lockstep on chunks (2 j + 2, 2 j + 3)
  // while the recursion waves consume chunks 2 j and 2 j + 1: a producer has two trips of the chunk loop per record — a record is
  // ~1100 dependent instructions of model code, a trip is 4 x ~160 on the recursion waves, and one producer alone set the pace
  // [measured: 253 against 213 us per launch].  Its stores come in the second trip (StagedCoefSink::midpoint), when nobody reads
  // the slot any more.
  const int n_chunks = (T + kS - 1) / kS;
  auto produceChunk = [&](int k) {
    const int i = T - 1 - k * kS - p_ts;
    fmpc::CoefInputs<Problem> in;
    fmpc::loadCoefInputs<Problem>(buf, b, i > 0 ? i : 0, in);
    fmpc::StagedCoefSink<N, M> sink{stage_lds + static_cast<size_t>(k % 3) * kSlotDoubles + (p_ts * 16 + p_inst) * kRecB, buf, b,
                                    i > 0 ? i : 0, i >= 0, live, &sh_coef_nan[p_inst]};
    fmpc::coefficients<Problem>(buf, b, i > 0 ? i : 0, in, sink); // (two workgroup barriers inside: sink.midpoint())
  };
  {
    static_assert(kS == 4, "a producer wave's lane mapping is 4 timesteps x 16 instances");
    // prologue: chunks 0 and 1 (every wavefront passes the producers' two mid-record barriers)
    if(producer)
    {
      if(wave - 4 < n_chunks)
      {
        produceChunk(wave - 4);
      }
      else
      {
        syncThreadsFuzzed(__LINE__);
        syncThreadsFuzzed(__LINE__);
      }
    }
    else
    {
      syncThreadsFuzzed(__LINE__);
      syncThreadsFuzzed(__LINE__);
    }
    syncThreadsFuzzed(__LINE__);
    if(producer)
    {
      for(int c = 0; c < n_chunks; c += 2)
      {
        const int k = c + 2 + (wave - 4);
        if(k < n_chunks)
        {
          produceChunk(k); // first half, the two barriers of trip c, second half
          syncThreadsFuzzed(__LINE__); // the two barriers of trip c + 1 (it exists: k < n_chunks)
          syncThreadsFuzzed(__LINE__);
        }
        else
        {
          syncThreadsFuzzed(__LINE__);
          syncThreadsFuzzed(__LINE__);
          if(c + 1 < n_chunks)
          {
            syncThreadsFuzzed(__LINE__);
            syncThreadsFuzzed(__LINE__);
          }
        }
      }
    }
    for(int c = 0; c < (producer ? 0 : n_chunks); c++)
    {
      const int i0 = T - 1 - c * kS;
      const int slot = c % 3;
      {
        const double * recs = stage_lds + static_cast<size_t>(slot) * kSlotDoubles + inst * kRecB;
        Operands o[2];
        loadOperands(recs, o[0]);
        NMPC_UNROLL
        for(int st = 0; st < kS; st++)
        {
          if(i0 - st >= 0)
          {
            if(st + 1 < kS)
            {
              loadOperands(recs + (st + 1) * 16 * kRecB, o[(st + 1) & 1]);
            }
            backwardStep(st, o[st & 1]);
          }
        }
      }
      syncThreadsFuzzed(__LINE__); // every wavefront's gains of this chunk are in LDS
      flushGains(i0);
      syncThreadsFuzzed(__LINE__);
    }
  }
  // verdict of the backward pass per instance (:640-653): OR over the sixteen lanes of the instance
  {
    int bad_any = nan ? 1 : 0;
    bad_any |= __shfl_xor(bad_any, 1);
    bad_any |= __shfl_xor(bad_any, 2);
    bad_any |= __shfl_xor(bad_any, 16);
    bad_any |= __shfl_xor(bad_any, 32);
    const bool failed = buf.check_nan && (bad_any != 0 || (buf.flags[b] & 1) || sh_coef_nan[inst] != 0);
    if(live && failed && head)
    {
      buf.status[b] = 3; // Status::ErrorInBackward
    }
    live = live && !failed;
    if(head)
    {
      sh_live[inst] = live ? 1 : 0;
    }
  }
  // (no early exit from here on: every wavefront of the workgroup takes part in the staging barriers of the forward pass,
  // and the gains the other wavefronts wrote above are read below: make them visible first)
  __threadfence();
  syncThreadsFuzzed(__LINE__);

  // ---- forward pass, the recursion over the timesteps (:669-687); dx[row] is kept by every lane of the row
#ifdef NMPC_AMD_FMPC_PROFILE
  const unsigned long long tick1 = wall_clock64();
#endif
  double dx_row = rv ? buf.x0[static_cast<size_t>(rc) * Bz + b] - buf.x[fmpc::at(buf, 0, rc, N, b)] : 0.0;
  // one value per lane and step by role, parked in LDS: dx[row] on the diagonal lanes, du on lane (1, 0)
  const bool role_dx = (row == col) && rv, role_du = (row == 1 && col == 0);
  double * const fP = gain_lds + inst * kRecX + (role_dx ? rc : (role_du ? N : N + M));
  const int oAT = CL::A + cc + rc * N; // entry (col, row): the A operand of mma is used transposed
  const int oKx = kFwdCoef + GL::K + rc, ok = kFwdCoef + GL::k;
  struct ForwardOperands
  {
    double AT, Kx, k, Bv, xb;
  };
  auto loadForward = [&](const double * rec, ForwardOperands & o) {
    const double at_ = rec[oAT], kx = rec[oKx], bv = rec[oB], xb = rec[oX];
    o.k = rec[ok];
    o.AT = valid ? at_ : 0.0;
    o.Kx = rv ? kx : 0.0; // K[row] in every column
    o.Bv = rv ? bv : 0.0;
    o.xb = rv ? xb : 0.0;
  };
  auto forwardStep = [&](int st, const ForwardOperands & o, int park) {
    const double Y = c0 ? dx_row : 0.0;
    const double ax = bcast0(mma(o.AT, Y, 0.0)); // (A dx)[row]
    const double du = bcast0(mma(o.Kx, Y, 0.0)) + o.k; // (2.36), the same in every row
    fP[park + st * 16 * kRecX] = role_du ? du : dx_row;
    const double nx = (ax + o.Bv * du) + o.xb; // (2.26b)
    dx_row = rv ? nx : 0.0;
  };
  {
    // The forward pass, one workgroup barrier per chunk of kF steps.  The PRODUCER waves (idle in this pass) stage the operands: a
    // chunk's A, B, x_bar, k, K go from HBM through registers into one of two LDS slots, requested two chunks before the recursion
    // waves read them and parked while those compute the chunk in front; the recursion waves do the recursion and send dx, du of the
    // chunk before — parked in LDS, two areas used alternately — to HBM in whole lines.  How it got here: (1) requests one chunk
    // ahead, issued and parked by the recursion waves, were a round trip to HBM per chunk of four steps [profile build: 79 us of the
    // launch's 233]; (2) requesting further ahead only pays if the wait in front of the parking is COUNTED (s_waitcnt vmcnt(N)),
    // which needs a wave that has loads only in flight — loads and stores share the counter and complete out of order with respect
    // to each other, one pending store makes every wait vmcnt(0), also when the stores sit on another path of the same loop — no
    // per-lane branches around the requests (every lane loads for every step; one without an element of its own loads the record's
    // last), and a set requested right behind the parking of its predecessor (across the loop's back edge the compiler still counts
    // short and waits for the first request of the newest set: by then it is a chunk old) [62 us]; (3) then the recursion waves were
    // bound by their ~90 instructions per step, 20 of them the recursion: staging moved to the producers.
    constexpr int kQC = (kFwdCoef + 7) / 8, kQG = (kFwdGain + 7) / 8; // elements per staging thread (eight element slots x sixteen instances)
    constexpr int kPark = kF * 16 * kRecX;
    static_assert(2 * kPark <= kGainLds, "[FMPC] two parking areas for dx, du");
    if(producer)
    {
      // (Measured and not kept: each producer wave staging every OTHER chunk alone, one chunk in flight per wave — what a wave waits
      // for is then two chunks old, but parking + requesting a whole chunk is ~650 instructions of one wave inside one chunk's time:
      // 65 us against 53.)
      const int s_slot = t_slot - 16; // waves 4, 5: element slot 0 ... 7 of instance t_inst
      /** Elements s_slot, s_slot + 8, ... of steps i0 ... i0 + kF - 1 of this thread's instance: request (straight-line: a step
          beyond the horizon is the last step again, a lane without an element of its own re-reads the record's last one). */
      auto request = [&](auto nr_tag, const double * src, int stride, int i0, double * v) {
        constexpr int NR = decltype(nr_tag)::value;
        constexpr int kQ = (NR + 7) / 8;
        NMPC_UNROLL
        for(int st = 0; st < kF; st++)
        {
          const int step = i0 + st < T ? i0 + st : T - 1; // wavefront-uniform
          NMPC_UNROLL
          for(int q = 0; q < kQ; q++)
          {
            const int e = (8 * q + 7 < NR || s_slot < NR - 8 * q) ? 8 * q + s_slot : NR - 1;
            v[st * kQ + q] = src[(static_cast<size_t>(step) * stride + e) * Bz + b_stage];
          }
        }
      };
      /** ... and park them in forward-pass slot `slot` as records of width kRecF, at offset `off` of each record. */
      auto commit = [&](auto nr_tag, int slot, int off, const double * v) {
        constexpr int NR = decltype(nr_tag)::value;
        constexpr int kQ = (NR + 7) / 8;
        double * base = stage_lds + static_cast<size_t>(slot) * kFwdSlotDoubles + t_inst * kRecF + off + s_slot;
        NMPC_UNROLL
        for(int st = 0; st < kF; st++)
        {
          NMPC_UNROLL
          for(int q = 0; q < kQ; q++)
          {
            if(8 * q + 7 < NR || s_slot < NR - 8 * q)
            {
              base[st * 16 * kRecF + 8 * q] = v[st * kQ + q];
            }
          }
        }
      };
      double vcA[kF * kQC], vgA[kF * kQG], vcB[kF * kQC], vgB[kF * kQG];
      request(TagFc(), buf.coef, CL::kStride, 0, vcA);
      request(TagFg(), buf.gain, GL::kStride, 0, vgA);
      commit(TagFc(), 0, 0, vcA);
      commit(TagFg(), 0, kFwdCoef, vgA);
      request(TagFc(), buf.coef, CL::kStride, kF, vcB);
      request(TagFg(), buf.gain, GL::kStride, kF, vgB);
      request(TagFc(), buf.coef, CL::kStride, 2 * kF, vcA);
      request(TagFg(), buf.gain, GL::kStride, 2 * kF, vgA);
      syncThreadsFuzzed(__LINE__); // chunk 0 is staged
      for(int i0 = 0; i0 < T; i0 += 2 * kF)
      {
        // while the recursion waves compute chunk i0 from slot 0 (they are done with slot 1 since the barrier before)
        if(i0 + kF < T)
        {
          commit(TagFc(), 1, 0, vcB);
          commit(TagFg(), 1, kFwdCoef, vgB);
          request(TagFc(), buf.coef, CL::kStride, i0 + 3 * kF, vcB);
          request(TagFg(), buf.gain, GL::kStride, i0 + 3 * kF, vgB);
        }
        syncThreadsFuzzed(__LINE__);
        if(i0 + kF < T) // (workgroup-uniform) ... chunk i0 + kF from slot 1
        {
          if(i0 + 2 * kF < T)
          {
            commit(TagFc(), 0, 0, vcA);
            commit(TagFg(), 0, kFwdCoef, vgA);
            request(TagFc(), buf.coef, CL::kStride, i0 + 4 * kF, vcA);
            request(TagFg(), buf.gain, GL::kStride, i0 + 4 * kF, vgA);
          }
          syncThreadsFuzzed(__LINE__);
        }
      }
    }
    else
    {
      syncThreadsFuzzed(__LINE__); // chunk 0 is staged
      int chunk = 0;
      for(int i0 = 0; i0 < T; i0 += kF, chunk++)
      {
        const int slot = chunk & 1, park = (chunk & 1) * kPark;
        const double * recs = stage_lds + static_cast<size_t>(slot) * kFwdSlotDoubles + inst * kRecF;
        ForwardOperands o[2];
        loadForward(recs, o[0]);
        NMPC_UNROLL
        for(int st = 0; st < kF; st++)
        {
          if(i0 + st < T)
          {
            if(st + 1 < kF)
            {
              loadForward(recs + (st + 1) * 16 * kRecF, o[(st + 1) & 1]);
            }
            forwardStep(st, o[st & 1], park);
          }
        }
        syncThreadsFuzzed(__LINE__); // dx, du of the chunk are parked; the next chunk's operands are staged
        if(sh_live[t_inst] != 0 && t_slot < N + M) // dx, du of the chunk: LDS -> HBM in whole lines (the next chunk parks in the other area)
        {
          NMPC_UNROLL
          for(int st = 0; st < kF; st++)
          {
            const int step = i0 + st;
            if(step < T)
            {
              const double v = gain_lds[park + (st * 16 + t_inst) * kRecX + t_slot];
              if(t_slot < N)
              {
                buf.dx[(static_cast<size_t>(step) * N + t_slot) * Bz + b_stage] = v;
              }
              else
              {
                buf.du[(static_cast<size_t>(step) * M + (t_slot - N)) * Bz + b_stage] = v;
              }
            }
          }
        }
      }
    }
  }
  if(!producer && live && c0 && rv)
  {
    buf.dx[fmpc::at(buf, T, rc, N, b)] = dx_row;
  }
#ifdef NMPC_AMD_FMPC_PROFILE
  if(head && b_raw < buf.B) // developer build only: 100 MHz wall-clock ticks of the two recursions in the merit slots
  {
    buf.merit[0 * buf.B + b] = static_cast<double>(tick1 - tick0);
    buf.merit[1 * buf.B + b] = static_cast<double>(wall_clock64() - tick1);
  }
#endif
}

/** The timestep-parallel part of the forward pass (FmpcSolver.hpp:673: dlambda (2.33); :689-697: ds, dnu), the NaN check of
    delta_variable_ (:699) and the
    per-timestep candidates of the fraction-to-boundary rule (:713-731).  C, D and g are re-evaluated instead of being kept
    from the coefficient kernel (for the box-type rows of the reference's problems that is a handful of instructions against
    (G N + G M + G) x 16 bytes of HBM traffic per timestep). */
template<class Problem>
__global__ void __launch_bounds__(256) fmpc_delta_kernel(FmpcBuffers buf)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  using GL = fmpc::GainLayout<N, M>;
  if(tid >= static_cast<size_t>(buf.B) * (buf.T + 1))
  {
    return;
  }
  const int b = static_cast<int>(tid % buf.B);
  const int i = static_cast<int>(tid / buf.B);
  if(buf.status[b] != fmpc::kStatusContinued)
  {
    return;
  }
  typename Problem::StateDimVector x, dx;
  fmpc::loadVec(buf.dx, buf, i, b, dx);
  bool nan = false;
  NMPC_UNROLL
  for(int a = 0; a < N; a++) // (2.33)
  {
    double acc = 0;
    NMPC_UNROLL
    for(int r = 0; r < N; r++)
    {
      acc += buf.gain[fmpc::at(buf, i, GL::P + a + r * N, GL::kStride, b)] * dx[r];
    }
    const double dl = acc - buf.gain[fmpc::at(buf, i, GL::S + a, GL::kStride, b)];
    buf.dlam[fmpc::at(buf, i, a, N, b)] = dl;
    nan = nan || fmpc::bad(dl) || fmpc::bad(dx[a]);
  }
  if(i == buf.T)
  {
    if(nan)
    {
      atomicOr(&buf.flags[b], 2);
    }
    return;
  }
  const Problem prob = fmpc::loadProblem<Problem>(buf, b);
  const double t = buf.t0[b] + i * prob.dt();
  typename Problem::InputDimVector u, du;
  typename Problem::IneqDimVector s, nu;
  fmpc::loadVec(buf.x, buf, i, b, x);
  fmpc::loadVec(buf.u, buf, i, b, u);
  fmpc::loadVec(buf.du, buf, i, b, du);
  NMPC_UNROLL
  for(int a = 0; a < M; a++)
  {
    nan = nan || fmpc::bad(du[a]);
  }
  fmpc::loadVec(buf.s, buf, i, b, s);
  fmpc::loadVec(buf.nu, buf, i, b, nu);
  typename Problem::IneqStateDimMatrix C;
  typename Problem::IneqInputDimMatrix D;
  prob.calcIneqConstDeriv(t, x, u, C, D);
  const typename Problem::IneqDimVector g = prob.ineqConst(t, x, u);
  const double barrier_eps = buf.barrier_eps[b];
  constexpr double margin_ratio = 0.995;
  double alpha_s = 1.0, alpha_nu = 1.0;
  NMPC_UNROLL
  for(int j = 0; j < G; j++)
  {
    double cx = 0, dd = 0;
    NMPC_UNROLL
    for(int r = 0; r < N; r++)
    {
      cx += C(j, r) * dx[r];
    }
    NMPC_UNROLL
    for(int r = 0; r < M; r++)
    {
      dd += D(j, r) * du[r];
    }
    const double g_bar = g[j] + s[j];
    const double dsj = -1 * ((cx + dd) + g_bar); // (2.27a)
    const double dnj = -1 * (nu[j] * (dsj + s[j]) - barrier_eps) / s[j]; // (2.27b)
    buf.ds[fmpc::at(buf, i, j, G, b)] = dsj;
    buf.dnu[fmpc::at(buf, i, j, G, b)] = dnj;
    nan = nan || fmpc::bad(dsj) || fmpc::bad(dnj);
    if(dsj < 0) // (19.9) in Nocedal & Wright
    {
      const double c = -1 * margin_ratio * s[j] / dsj;
      alpha_s = (c < alpha_s) ? c : alpha_s;
    }
    if(dnj < 0)
    {
      const double c = -1 * margin_ratio * nu[j] / dnj;
      alpha_nu = (c < alpha_nu) ? c : alpha_nu;
    }
  }
  buf.part[fmpc::at(buf, i, 1, fmpc::kPartSlots, b)] = alpha_s;
  buf.part[fmpc::at(buf, i, 2, fmpc::kPartSlots, b)] = alpha_nu;
  if(nan)
  {
    atomicOr(&buf.flags[b], 2);
  }
}

/** l1NormDirectionalDeriv (MathUtils.h:17-38) for one row: func_i, (jac row i) . dir. */
__device__ __forceinline__ double fmpcL1RowDeriv(double func_i, double row_dot_dir)
{
  return func_i > 0 ? row_dot_dir : (func_i < 0 ? -1 * row_dot_dir : fabs(row_dot_dir));
}

/** Line search on the l1 merit function (FmpcSolver.hpp:748-792 with setupMeritFunc :840-936 and calcMeritFunc :938-981), one
    instance per lane.  Off by default in the reference (FmpcSolver.h:85) and in both of its tests; kept sequential over the
    horizon because the number of backtracking trials differs per instance. */
template<class Problem>
__global__ void __launch_bounds__(64) fmpc_line_search_kernel(FmpcBuffers buf, int iter)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  using CL = fmpc::CoefLayout<N, M>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= buf.B || buf.status[b] != fmpc::kStatusContinued)
  {
    return;
  }
  const int T = buf.T;
  const Problem prob = fmpc::loadProblem<Problem>(buf, b);
  const double dt = prob.dt();
  const double t0 = buf.t0[b];
  const double barrier_eps = buf.barrier_eps[b];

  // merit function at a trial step length: FmpcSolver::calcMeritFunc on variable_ + alpha * delta_variable_ (alpha = 0: the
  // function part of setupMeritFunc)
  auto merit = [&](double alpha, double & obj, double & con) {
    obj = 0;
    con = 0;
    typename Problem::StateDimVector x, next_x;
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      x[a] = buf.x[fmpc::at(buf, 0, a, N, b)] + alpha * buf.dx[fmpc::at(buf, 0, a, N, b)];
      con += fabs(buf.x0[static_cast<size_t>(a) * buf.B + b] - x[a]);
    }
    for(int i = 0; i < T; i++)
    {
      const double t = t0 + i * dt;
      typename Problem::InputDimVector u;
      NMPC_UNROLL
      for(int a = 0; a < M; a++)
      {
        u[a] = buf.u[fmpc::at(buf, i, a, M, b)] + alpha * buf.du[fmpc::at(buf, i, a, M, b)];
      }
      NMPC_UNROLL
      for(int a = 0; a < N; a++)
      {
        next_x[a] = buf.x[fmpc::at(buf, i + 1, a, N, b)] + alpha * buf.dx[fmpc::at(buf, i + 1, a, N, b)];
      }
      obj += prob.runningCost(t, x, u) * dt;
      const typename Problem::IneqDimVector g = prob.ineqConst(t, x, u);
      double logsum = 0, c2 = 0;
      NMPC_UNROLL
      for(int j = 0; j < G; j++)
      {
        const double sj = buf.s[fmpc::at(buf, i, j, G, b)] + alpha * buf.ds[fmpc::at(buf, i, j, G, b)];
        logsum += log(sj);
        c2 += fabs(g[j] + sj);
      }
      obj += -1 * barrier_eps * logsum;
      const typename Problem::StateDimVector f = prob.stateEq(t, x, u);
      double c1 = 0;
      NMPC_UNROLL
      for(int a = 0; a < N; a++)
      {
        c1 += fabs(f[a] - next_x[a]);
      }
      con += c1;
      con += c2;
      x = next_x;
    }
    obj += prob.terminalCost(t0 + T * dt, x);
  };

  // setupMeritFunc: directional derivatives (:852-905)
  double merit_func_obj, merit_func_const;
  merit(0.0, merit_func_obj, merit_func_const);
  double merit_deriv_obj = 0, merit_deriv_const = 0;
  {
    double acc = 0;
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      const double cf = buf.x0[static_cast<size_t>(a) * buf.B + b] - buf.x[fmpc::at(buf, 0, a, N, b)];
      acc += fmpcL1RowDeriv(cf, -1 * buf.dx[fmpc::at(buf, 0, a, N, b)]);
    }
    merit_deriv_const += acc;
  }
  for(int i = 0; i < T; i++)
  {
    const double t = t0 + i * dt;
    typename Problem::StateDimVector x, next_x, dx, dnx;
    typename Problem::InputDimVector u, du;
    typename Problem::IneqDimVector s, ds;
    fmpc::loadVec(buf.x, buf, i, b, x);
    fmpc::loadVec(buf.x, buf, i + 1, b, next_x);
    fmpc::loadVec(buf.dx, buf, i, b, dx);
    fmpc::loadVec(buf.dx, buf, i + 1, b, dnx);
    fmpc::loadVec(buf.u, buf, i, b, u);
    fmpc::loadVec(buf.du, buf, i, b, du);
    fmpc::loadVec(buf.s, buf, i, b, s);
    fmpc::loadVec(buf.ds, buf, i, b, ds);
    typename Problem::StateDimVector Lx;
    typename Problem::InputDimVector Lu;
    typename Problem::StateStateDimMatrix Lxx;
    typename Problem::InputInputDimMatrix Luu;
    typename Problem::StateInputDimMatrix Lxu;
    prob.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
    double lx = 0, lu = 0, invdot = 0;
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      lx += Lx[a] * dx[a];
    }
    NMPC_UNROLL
    for(int a = 0; a < M; a++)
    {
      lu += Lu[a] * du[a];
    }
    merit_deriv_obj += (lx + lu) * dt;
    NMPC_UNROLL
    for(int j = 0; j < G; j++)
    {
      invdot += (1.0 / s[j]) * ds[j];
    }
    merit_deriv_obj += -1 * barrier_eps * invdot;
    {
      const typename Problem::StateDimVector f = prob.stateEq(t, x, u);
      double dA = 0, dB = 0, dI = 0;
      NMPC_UNROLL
      for(int a = 0; a < N; a++)
      {
        const double cf = f[a] - next_x[a];
        double ra = 0, rb = 0;
        NMPC_UNROLL
        for(int r = 0; r < N; r++)
        {
          ra += buf.coef[fmpc::at(buf, i, CL::A + a + r * N, CL::kStride, b)] * dx[r];
        }
        NMPC_UNROLL
        for(int r = 0; r < M; r++)
        {
          rb += buf.coef[fmpc::at(buf, i, CL::B + a + r * N, CL::kStride, b)] * du[r];
        }
        dA += fmpcL1RowDeriv(cf, ra);
        dB += fmpcL1RowDeriv(cf, rb);
        dI += fmpcL1RowDeriv(cf, -1 * dnx[a]);
      }
      merit_deriv_const += dA;
      merit_deriv_const += dB;
      merit_deriv_const += dI;
    }
    {
      typename Problem::IneqStateDimMatrix C;
      typename Problem::IneqInputDimMatrix D;
      prob.calcIneqConstDeriv(t, x, u, C, D);
      const typename Problem::IneqDimVector g = prob.ineqConst(t, x, u);
      double dC = 0, dD = 0, dI = 0;
      NMPC_UNROLL
      for(int j = 0; j < G; j++)
      {
        const double cf = g[j] + s[j];
        double rc = 0, rd = 0;
        NMPC_UNROLL
        for(int r = 0; r < N; r++)
        {
          rc += C(j, r) * dx[r];
        }
        NMPC_UNROLL
        for(int r = 0; r < M; r++)
        {
          rd += D(j, r) * du[r];
        }
        dC += fmpcL1RowDeriv(cf, rc);
        dD += fmpcL1RowDeriv(cf, rd);
        dI += fmpcL1RowDeriv(cf, ds[j]);
      }
      merit_deriv_const += dC;
      merit_deriv_const += dD;
      merit_deriv_const += dI;
    }
  }
  {
    typename Problem::StateDimVector xT, Vx;
    typename Problem::StateStateDimMatrix Vxx;
    fmpc::loadVec(buf.x, buf, T, b, xT);
    prob.calcTerminalCostDeriv(t0 + T * dt, xT, Vx, Vxx);
    double acc = 0;
    NMPC_UNROLL
    for(int a = 0; a < N; a++)
    {
      acc += Vx[a] * buf.dx[fmpc::at(buf, T, a, N, b)];
    }
    merit_deriv_obj += acc;
  }

  constexpr double merit_const_scale_min = 1e-3;
  double merit_const_scale;
  if(buf.merit_const_scale_from_lagrange_multipliers) // (18.32) in Nocedal & Wright
  {
    merit_const_scale = merit_const_scale_min;
    for(int i = 0; i <= T; i++)
    {
      for(int a = 0; a < N; a++)
      {
        const double v = fabs(buf.lam[fmpc::at(buf, i, a, N, b)]);
        merit_const_scale = (merit_const_scale < v) ? v : merit_const_scale;
      }
      if(i < T)
      {
        for(int j = 0; j < G; j++)
        {
          const double v = fabs(buf.nu[fmpc::at(buf, i, j, G, b)]);
          merit_const_scale = (merit_const_scale < v) ? v : merit_const_scale;
        }
      }
    }
  }
  else // (18.33)
  {
    constexpr double rho = 0.5;
    const double v = merit_deriv_obj / ((1.0 - rho) * merit_func_const);
    merit_const_scale = (v < merit_const_scale_min) ? merit_const_scale_min : v; // std::max(v, min)
  }
  const double merit_func = merit_func_obj + merit_const_scale * merit_func_const;
  const double merit_deriv = merit_deriv_obj + merit_const_scale * merit_deriv_const;
  buf.merit[0 * buf.B + b] = merit_func;
  buf.merit[1 * buf.B + b] = merit_deriv;
  buf.merit[2 * buf.B + b] = merit_const_scale;

  constexpr double armijo_scale = 1e-3;
  constexpr double alpha_s_update_ratio = 0.5;
  constexpr double alpha_s_min = 1e-10;
  double alpha_s = buf.alpha[0 * buf.B + b];
  while(true)
  {
    if(alpha_s < alpha_s_min)
    {
      break;
    }
    double obj, con;
    merit(alpha_s, obj, con);
    const double merit_func_new = obj + merit_const_scale * con;
    if(merit_func_new < merit_func + armijo_scale * alpha_s * merit_deriv)
    {
      break;
    }
    alpha_s *= alpha_s_update_ratio;
  }
  buf.alpha[2 * buf.B + b] = alpha_s;
  buf.trace[(static_cast<size_t>(b) * buf.max_iter + (iter - 1)) * NMPC_HIP_FMPC_NTRACE + NMPC_HIP_FMPC_TRACE_ALPHA_S] = alpha_s;
}

/** The plant step of the reference's closed-loop tests (TestFmpcOscillator.cpp:191, TestFmpcCartPole.cpp:352):
    x <- stateEq(t, x, u + K_0 (x_list[0] - x) * use_feedback, sim_dt), t += sim_dt, `substeps` times, on the handle's resident
    arrays; u = u_list[0] of the last solve.  x_plant / t_plant: [N][B] / [B]. */
template<class Problem>
__global__ void fmpc_plant_kernel(FmpcBuffers buf, double * x_plant, double * t_plant, double sim_dt, int substeps, int use_feedback)
{
  constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax;
  using GL = fmpc::GainLayout<N, M>;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= buf.B)
  {
    return;
  }
  const Problem prob = fmpc::loadProblem<Problem>(buf, b);
  typename Problem::StateDimVector x;
  typename Problem::InputDimVector u0;
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    x[a] = x_plant[static_cast<size_t>(a) * buf.B + b];
  }
  NMPC_UNROLL
  for(int a = 0; a < M; a++)
  {
    u0[a] = buf.u[fmpc::at(buf, 0, a, M, b)];
  }
  double t = t_plant[b];
  for(int k = 0; k < substeps; k++)
  {
    typename Problem::InputDimVector u = u0;
    if(use_feedback)
    {
      NMPC_UNROLL
      for(int a = 0; a < M; a++)
      {
        double acc = 0;
        NMPC_UNROLL
        for(int r = 0; r < N; r++)
        {
          acc += buf.gain[fmpc::at(buf, 0, GL::K + a + r * M, GL::kStride, b)] * (buf.x[fmpc::at(buf, 0, r, N, b)] - x[r]);
        }
        u[a] += acc;
      }
    }
    x = prob.stateEq(t, x, u, sim_dt);
    t += sim_dt;
  }
  NMPC_UNROLL
  for(int a = 0; a < N; a++)
  {
    x_plant[static_cast<size_t>(a) * buf.B + b] = x[a];
  }
  t_plant[b] = t;
}
} // namespace hip
} // namespace nmpc_amd
