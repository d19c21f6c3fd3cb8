// C-ABI of libnmpc_hip_ddp.so (declared in include/nmpc_hip_ddp.h): solver handles, device-buffer ownership,
// layout conversion at the boundary and the launch of the persistent solve kernel.  No CPU fallback exists: if
// the HIP runtime or a gfx950 device is missing every entry point that needs the GPU fails loudly.
#include <nmpc_hip_ddp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include <nmpc_amd/hip/model_ops.hpp>
#include <nmpc_amd/hip/ragged_schedule.hpp>
#include <nmpc_amd/hip/stream_schedule.hpp>

using nmpc_amd::hip::DeviceBuffers;
using nmpc_amd::hip::ModelOps;

namespace
{
// Hardware queues.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels
// of two streams that share a queue run one after the other.  A pool of handles (DDPSolverPool: consecutive batches on their own
// streams) overlaps only as many batches as there are queues [measured, profiles/r05_m2_overlap.txt: 8 handles, ragged schedule,
// 2048 batch-iterations/s with 4 queues, 4685 with 16].  The variable is read when the runtime initialises; the library does not
// touch its host's environment on its own — a caller that wants more queues says so through nmpc_hip_ddp_request_hw_queues()
// (the pool mirrors do), which reports whether the request can still take effect.

/** Whether the ROCm runtime of this process is already up: it holds /dev/kfd open from its initialisation on. */
bool runtimeInitialised()
{
  char link[64], target[256];
  for(int fd = 0; fd < 4096; fd++)
  {
    std::snprintf(link, sizeof(link), "/proc/self/fd/%d", fd);
    const ssize_t len = readlink(link, target, sizeof(target) - 1);
    if(len > 0)
    {
      target[len] = '\0';
      if(std::strcmp(target, "/dev/kfd") == 0)
      {
        return true;
      }
    }
  }
  return false;
}

thread_local std::string g_last_error;

int fail(int code, const std::string & msg)
{
  g_last_error = msg;
  return code;
}

#define NMPC_HIP_TRY(expr)                                                                                     \
  do                                                                                                           \
  {                                                                                                            \
    hipError_t e_ = (expr);                                                                                    \
    if(e_ != hipSuccess)                                                                                       \
    {                                                                                                          \
      return fail(NMPC_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    }                                                                                                          \
  } while(0)

std::vector<const ModelOps *> & registry()
{
  static std::vector<const ModelOps *> r;
  return r;
}

const ModelOps * findModel(const char * name)
{
  if(!name)
  {
    return nullptr;
  }
  for(const ModelOps * m : registry())
  {
    if(std::strcmp(m->name, name) == 0)
    {
      return m;
    }
  }
  return nullptr;
}
} // namespace

struct nmpc_hip_ddp_solver
{
  const ModelOps * ops = nullptr;
  int device = 0;
  int B = 0, Bp = 0, T = 0, N = 0, M = 0, MM = 1;
  nmpc_hip_ddp_config cfg;
  std::vector<unsigned char> params;
  double lim_lo[nmpc_amd::hip::kMaxInputDim];
  double lim_hi[nmpc_amd::hip::kMaxInputDim];
  bool has_limits = false;
  bool has_shared_limits = false; // nmpc_hip_ddp_set_input_limits was called
  bool solved = false;
  int last_gain_layout = 0; // ModelOps::gain_layout of the kernel the last solve ran on
  hipStream_t stream = nullptr;
  // ring of HIP-event triples {begin, kernel start, end}: one per solve, harvested lazily so that timing a
  // sequence of asynchronous solves never inserts a host synchronisation between them
  static constexpr int kEvPool = 128;
  hipEvent_t ev_begin[kEvPool] = {}, ev_kernel[kEvPool] = {}, ev_end[kEvPool] = {};
  bool ev_pending[kEvPool] = {};
  long long n_solves = 0; // solves recorded since create / last reset
  long long n_harvested = 0;
  double sum_total_ms = 0, sum_kernel_ms = 0;
  float last_total_ms = 0, last_kernel_ms = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_staged = nullptr; // behind the H2D staging copies of nmpc_hip_ddp_solve_async
  bool queued_solve = false; // the solve being launched came through nmpc_hip_ddp_solve_async (ragged_schedule 0: see raggedRounds)

  int elem = 8; //!< sizeof(Problem::Scalar): element size of every Scalar array below (ModelOps::scalar_bytes)
  // device memory (Scalar arrays are typed double here; an fp32 problem type stores floats in them, see elem)
  double * d_t0 = nullptr;
  double * d_x0 = nullptr;
  double * d_X = nullptr;
  double * d_U = nullptr;
  double * d_cost = nullptr;
  double * d_kff = nullptr;
  double * d_Kfb = nullptr;
  double * d_trace = nullptr;
  double * d_trace_last = nullptr;
  double * d_dV = nullptr;
  int * d_status = nullptr;
  int * d_iters = nullptr;
  int * d_sel = nullptr;
  int * d_qp_ret = nullptr;
  unsigned * d_qp_free = nullptr;
  int * d_input_dim = nullptr;
  double * d_wpi_ws = nullptr; // per-instance workspace: wave-per-instance / fp32 tile kernels, fan-out scratch of the quad kernel
  unsigned char * d_params_batch = nullptr; // [Bp][param_bytes] per-instance problem objects, or nullptr
  double * d_lim_batch = nullptr; // [Bp][2][kMaxInputDim] per-instance input limits, or nullptr
  unsigned long long * d_phase_ticks = nullptr; // [Bp][4] per-instance phase ticks of the last solve
  double * d_lim_steps = nullptr; // [1 or Bp][T][2][MM] time-varying input limits, or nullptr
  int lim_steps_per_instance = 0;
  int lim_rows = 0; // rows per table of d_lim_steps
  int lim_offset = 0; // row of timestep 0 of the next launch (mpc_run's tick)
  int trace_rows = 0;
  // staging in the reference layouts
  void * d_stage_in = nullptr; // x0 / u_init / t0 as handed over by the host entry point
  size_t stage_in_bytes = 0;
  void * d_stage_out = nullptr; // one result field in the reference layout
  size_t stage_out_bytes = 0;
  // ragged-convergence schedule (ragged_schedule.hpp): resumable launches with a device-side compaction between them
  static constexpr int kRaggedMaxRounds = 24;
  double * d_resume = nullptr; // [tile][kResumeRows][64] parked solver state (Scalar array)
  int * d_ragged_words = nullptr; // n_active[kRaggedMaxRounds + 1], then n_swaps[kRaggedMaxRounds]
  int * d_ragged_pairs = nullptr; // [kRaggedMaxRounds][Bp]: the (p, q) position pairs of every round's swaps
  int * d_ragged_rank = nullptr; // [Bp] scratch of the compaction kernel
  int * d_ragged_used = nullptr; // [Bp / 2] trace rows in use per pair of the round being swapped
  nmpc_amd::hip::LaunchKnobs knobs; // kernel family / schedule choices of this handle (environment overrides read once at create)
  // streamed solves (nmpc_hip_ddp_solve_stream): slot -> instance, the schedule's device words, the caller-facing arrays of the last stream
  int * d_stream_id = nullptr; // [Bp]
  int * d_stream_words = nullptr; // [kSwCount]
  void * d_stream_io = nullptr; // inputs and outputs of the last streamed solve, reference layouts (stream_io_bytes allocated)
  size_t stream_io_bytes = 0;
  int stream_n = 0; // instances of the last streamed solve (0: none)
  nmpc_amd::hip::StreamArrays stream_arrays = {};
  int stream_rounds = 0;
  long long stream_instance_iterations = 0;
  float stream_ms = 0;
  bool ragged_ready = false; // all five ragged buffers are allocated and cleared (ensureRagged)
  bool ragged_unavailable = false; // their allocation failed once: the handle keeps to whole-solve launches
  int ragged_env = 0; // NMPC_HIP_DDP_RAGGED, read once at create: 1 forces the schedule on, -1 off (A/B measurements)
  int last_ragged_rounds = 0; // launches of the last solve (1: an ordinary whole-solve launch)
};

namespace
{
template<class T>
int devAlloc(T ** p, size_t count)
{
  NMPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
  NMPC_HIP_TRY(hipMemset(*p, 0, count * sizeof(T)));
  return NMPC_HIP_OK;
}

/** A Scalar array of `count` elements of `elem` bytes behind a double pointer. */
int devAllocScalar(double ** p, size_t count, int elem)
{
  NMPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * static_cast<size_t>(elem)));
  NMPC_HIP_TRY(hipMemset(*p, 0, count * static_cast<size_t>(elem)));
  return NMPC_HIP_OK;
}

int ensureStage(void ** p, size_t * have, size_t need)
{
  if(*have >= need)
  {
    return NMPC_HIP_OK;
  }
  if(*p)
  {
    NMPC_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  NMPC_HIP_TRY(hipMalloc(p, need));
  *have = need;
  return NMPC_HIP_OK;
}

int allocTrace(nmpc_hip_ddp_solver * s)
{
  const int rows = (s->cfg.trace_level >= 1) ? s->cfg.max_iter + 1 : 1;
  if(rows == s->trace_rows && s->d_trace)
  {
    return NMPC_HIP_OK;
  }
  if(s->d_trace)
  {
    NMPC_HIP_TRY(hipDeviceSynchronize()); // a solve in flight may still write the old trace buffer
    NMPC_HIP_TRY(hipFree(s->d_trace));
    s->d_trace = nullptr;
  }
  s->trace_rows = rows;
  return devAllocScalar(&s->d_trace, static_cast<size_t>(rows) * NMPC_HIP_NTRACE * s->Bp, s->elem);
}

DeviceBuffers makeBuffers(const nmpc_hip_ddp_solver * s)
{
  DeviceBuffers b;
  b.B = s->B;
  b.Bp = s->Bp;
  b.T = s->T;
  b.trace_rows = s->trace_rows;
  b.t0 = s->d_t0;
  b.x0 = s->d_x0;
  b.X = s->d_X;
  b.U = s->d_U;
  b.cost = s->d_cost;
  b.kff = s->d_kff;
  b.Kfb = s->d_Kfb;
  b.trace = s->d_trace;
  b.trace_last = s->d_trace_last;
  b.dV = s->d_dV;
  b.status = s->d_status;
  b.iters = s->d_iters;
  b.sel = s->d_sel;
  b.qp_ret = s->d_qp_ret;
  b.qp_free = s->d_qp_free;
  b.input_dim = s->d_input_dim;
  b.wpi_ws = s->d_wpi_ws;
  b.params_batch = s->d_params_batch;
  b.lim_batch = s->d_lim_batch;
  b.phase_ticks = s->d_phase_ticks;
  b.lim_steps = s->d_lim_steps;
  b.lim_steps_per_instance = s->lim_steps_per_instance;
  b.lim_mm = s->MM;
  b.lim_rows = s->lim_rows;
  b.lim_offset = s->lim_offset;
  for(int i = 0; i < nmpc_amd::hip::kMaxInputDim; i++)
  {
    b.lim_lo[i] = s->lim_lo[i];
    b.lim_hi[i] = s->lim_hi[i];
  }
  return b;
}

/** Reference layout [B][R] (T = double / int / unsigned as the C-ABI exchanges it) -> the handle's tile-major array, whose
    element type is TDev. */
template<class T, class TDev = T>
hipError_t toTile(const T * in, TDev * out, int B, int R, int Bp, int halves, int half, hipStream_t st)
{
  dim3 grid((R + 63) / 64, Bp / 64);
  hipLaunchKernelGGL((nmpc_amd::hip::batch_major_to_tile_kernel<T, TDev>), grid, dim3(256), 0, st, in, out, B, R, halves, half);
  return hipGetLastError();
}

template<class TDev, class T = TDev>
hipError_t toMajor(const TDev * in, T * out, const int * sel, int B, int R, int Bp, int halves, hipStream_t st,
                   const int * row_limit = nullptr, int row_unit = 0)
{
  dim3 grid((R + 63) / 64, Bp / 64);
  hipLaunchKernelGGL((nmpc_amd::hip::tile_to_batch_major_kernel<TDev, T>), grid, dim3(256), 0, st, in, out, sel, B, R, halves,
                     row_limit, row_unit);
  return hipGetLastError();
}

/** Scalar arrays: the device element type follows the problem's arithmetic, the C-ABI side is always double. */
hipError_t scalarToTile(const nmpc_hip_ddp_solver * s, const double * in, double * out, int R, int halves, int half, hipStream_t st)
{
  if(s->elem == 4)
  {
    return toTile<double, float>(in, reinterpret_cast<float *>(out), s->B, R, s->Bp, halves, half, st);
  }
  return toTile<double, double>(in, out, s->B, R, s->Bp, halves, half, st);
}
hipError_t scalarToMajor(const nmpc_hip_ddp_solver * s, const double * in, double * out, const int * sel, int R, int halves,
                         hipStream_t st, const int * row_limit = nullptr, int row_unit = 0)
{
  if(s->elem == 4)
  {
    return toMajor<float, double>(reinterpret_cast<const float *>(in), out, sel, s->B, R, s->Bp, halves, st, row_limit, row_unit);
  }
  return toMajor<double, double>(in, out, sel, s->B, R, s->Bp, halves, st, row_limit, row_unit);
}

struct FieldInfo
{
  size_t rows; // elements per instance
  size_t elem; // bytes per element
};

int fieldInfo(const nmpc_hip_ddp_solver * s, int field, FieldInfo * fi)
{
  const size_t T = s->T, N = s->N, MM = s->MM;
  switch(field)
  {
    case NMPC_HIP_FIELD_X:
      *fi = {(T + 1) * N, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_U:
      *fi = {T * MM, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_COST:
      *fi = {T + 1, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_KFF:
      *fi = {T * MM, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_KFB:
      *fi = {T * N * MM, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_TRACE:
      *fi = {static_cast<size_t>(s->cfg.max_iter + 1) * NMPC_HIP_NTRACE, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_STATUS:
    case NMPC_HIP_FIELD_ITERS:
      *fi = {1, sizeof(int)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_TRACE_LAST:
      *fi = {NMPC_HIP_NTRACE, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_DV:
      *fi = {2, sizeof(double)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_QP_RETVAL:
    case NMPC_HIP_FIELD_INPUT_DIM:
      *fi = {T, sizeof(int)};
      return NMPC_HIP_OK;
    case NMPC_HIP_FIELD_QP_FREE_MASK:
      *fi = {T, sizeof(unsigned)};
      return NMPC_HIP_OK;
    default:
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "unknown field " + std::to_string(field));
  }
}

/** Convert one field from the device layout into the reference layout at d_out (device memory). */
int packField(nmpc_hip_ddp_solver * s, int field, void * d_out, hipStream_t st)
{
  const int B = s->B, Bp = s->Bp;
  (void)Bp;
  FieldInfo fi;
  int rc = fieldInfo(s, field, &fi);
  if(rc != NMPC_HIP_OK)
  {
    return rc;
  }
  const int R = static_cast<int>(fi.rows);
  double * dout = static_cast<double *>(d_out);
  int * iout = static_cast<int *>(d_out);
  switch(field)
  {
    case NMPC_HIP_FIELD_X:
      NMPC_HIP_TRY(scalarToMajor(s, s->d_X, dout, s->d_sel, R, 2, st));
      break;
    case NMPC_HIP_FIELD_U:
      NMPC_HIP_TRY(scalarToMajor(s, s->d_U, dout, s->d_sel, R, 2, st));
      break;
    case NMPC_HIP_FIELD_COST:
      NMPC_HIP_TRY(scalarToMajor(s, s->d_cost, dout, s->d_sel, R, 2, st));
      break;
    case NMPC_HIP_FIELD_KFF:
    case NMPC_HIP_FIELD_KFB:
      if(s->last_gain_layout == 1)
      {
        // instance-major gain records [B][T][MM + MM N] in the workspace (the tile kernels)
        const int per_step = (field == NMPC_HIP_FIELD_KFF) ? s->MM : s->MM * s->N;
        const int offset = (field == NMPC_HIP_FIELD_KFF) ? 0 : s->MM;
        const size_t total = static_cast<size_t>(B) * s->T * per_step;
        const dim3 grid(static_cast<unsigned>((total + 255) / 256)), blk(256);
        if(s->ops->scalar_bytes == 4)
        {
          hipLaunchKernelGGL((nmpc_amd::hip::gain_records_to_batch_major_kernel<float>), grid, blk, 0, st,
                             reinterpret_cast<const float *>(s->d_wpi_ws), dout, total, s->MM + s->MM * s->N, per_step, offset);
        }
        else
        {
          hipLaunchKernelGGL((nmpc_amd::hip::gain_records_to_batch_major_kernel<double>), grid, blk, 0, st,
                             reinterpret_cast<const double *>(s->d_wpi_ws), dout, total, s->MM + s->MM * s->N, per_step, offset);
        }
        NMPC_HIP_TRY(hipGetLastError());
      }
      else
      {
        NMPC_HIP_TRY(scalarToMajor(s, field == NMPC_HIP_FIELD_KFF ? s->d_kff : s->d_Kfb, dout, nullptr, R, 1, st));
      }
      break;
    case NMPC_HIP_FIELD_TRACE:
      if(s->trace_rows != s->cfg.max_iter + 1)
      {
        return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "full trace was not recorded: set trace_level = 1 before solve");
      }
      // rows beyond the last iteration of this solve are reported as zero (a reused handle keeps older rows on the device)
      NMPC_HIP_TRY(scalarToMajor(s, s->d_trace, dout, nullptr, R, 1, st, s->d_iters, NMPC_HIP_NTRACE));
      break;
    case NMPC_HIP_FIELD_TRACE_LAST:
      NMPC_HIP_TRY(scalarToMajor(s, s->d_trace_last, dout, nullptr, R, 1, st));
      break;
    case NMPC_HIP_FIELD_DV:
      NMPC_HIP_TRY(scalarToMajor(s, s->d_dV, dout, nullptr, R, 1, st));
      break;
    case NMPC_HIP_FIELD_STATUS:
      NMPC_HIP_TRY(hipMemcpyAsync(iout, s->d_status, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
      break;
    case NMPC_HIP_FIELD_ITERS:
      NMPC_HIP_TRY(hipMemcpyAsync(iout, s->d_iters, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
      break;
    case NMPC_HIP_FIELD_QP_RETVAL:
      NMPC_HIP_TRY(toMajor<int>(s->d_qp_ret, iout, nullptr, B, R, Bp, 1, st));
      break;
    case NMPC_HIP_FIELD_INPUT_DIM:
      NMPC_HIP_TRY(toMajor<int>(s->d_input_dim, iout, nullptr, B, R, Bp, 1, st));
      break;
    case NMPC_HIP_FIELD_QP_FREE_MASK:
      NMPC_HIP_TRY(toMajor<unsigned>(s->d_qp_free, static_cast<unsigned *>(d_out), nullptr, B, R, Bp, 1, st));
      break;
    default:
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "unknown field");
  }
  return NMPC_HIP_OK;
}

/** Wait for the event triple in `slot` and fold its elapsed times into the running sums. */
int harvestSlot(nmpc_hip_ddp_solver * s, int slot)
{
  if(!s->ev_pending[slot])
  {
    return NMPC_HIP_OK;
  }
  NMPC_HIP_TRY(hipEventSynchronize(s->ev_end[slot]));
  float tot = 0, ker = 0;
  NMPC_HIP_TRY(hipEventElapsedTime(&tot, s->ev_begin[slot], s->ev_end[slot]));
  NMPC_HIP_TRY(hipEventElapsedTime(&ker, s->ev_kernel[slot], s->ev_end[slot]));
  s->sum_total_ms += tot;
  s->sum_kernel_ms += ker;
  s->last_total_ms = tot;
  s->last_kernel_ms = ker;
  s->n_harvested++;
  s->ev_pending[slot] = false;
  return NMPC_HIP_OK;
}

int harvestAll(nmpc_hip_ddp_solver * s)
{
  // oldest first so that last_* end up describing the most recent solve
  for(int k = 0; k < nmpc_hip_ddp_solver::kEvPool; k++)
  {
    const int slot = static_cast<int>((s->n_solves + k) % nmpc_hip_ddp_solver::kEvPool);
    int rc = harvestSlot(s, slot);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
  }
  return NMPC_HIP_OK;
}

int checkConfig(const nmpc_hip_ddp_solver * s, const nmpc_hip_ddp_config * c)
{
  if(c->horizon_steps != s->T)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "horizon_steps is fixed at create(): " + std::to_string(s->T) + " but "
                                                   + std::to_string(c->horizon_steps) + ".");
  }
  if(c->use_state_eq_second_derivative)
  {
    // the reference throws here as well (DDPSolver.hpp:391-414)
    return fail(NMPC_HIP_ERR_RUNTIME, "Vector-tensor product is not implemented yet.");
  }
  if(c->n_alpha < 1 || c->n_alpha > NMPC_HIP_MAX_ALPHA)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "alpha_list size must be in [1, 32]");
  }
  if(c->max_iter < 0)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "max_iter must be non-negative");
  }
  if(c->reg_type != 1 && c->reg_type != 2 && c->reg_type != 0)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "reg_type must be 1 or 2");
  }
  return NMPC_HIP_OK;
}
} // namespace

namespace
{
/** Whether the next solve of the handle runs under the ragged-convergence schedule, and the last iteration of each of its launches:
    16, 32, 48, 64, then max_iter — batches beyond 4096 instances: 128 and 256 as well.  Half of the instances of a cart-pole batch are
    done after 14 iterations, a percent runs for hundreds: compaction pays while the prefix shrinks fast; once the running instances
    fit a few workgroups (64 iterations: 137 of 4096) another boundary only costs its ~90 us of launch gaps [measured: ten launches
    per 500-iteration solve 829, whole-solve launch 858 batch-iterations/s on a lone stream]. */
bool raggedRounds(const nmpc_hip_ddp_solver * s, std::vector<int> * caps)
{
  const int mode = s->ragged_env != 0 ? s->ragged_env : s->cfg.ragged_schedule;
  if(mode < 0 || s->ragged_unavailable || s->elem != 8 || s->ops->resumable_supported == nullptr)
  {
    return false;
  }
  // Automatic mode.  max_iter is only a cap (500 by default) and says nothing about how long THIS solve runs: a warm-started MPC
  // tick converges within the first launch and would still pay four more launches, eight compaction and eight replay kernels
  // (~90 us of launch gaps per boundary), and a lone stream gains nothing from the schedule even on a long solve
  // [measured 858 -> 842 batch-iterations/s].  What gains is a caller with OTHER batches queued behind this one, so automatic
  // means: solves queued through nmpc_hip_ddp_solve_async (DDPSolverBatch::solveAsync, the C++ DDPSolverPool); synchronous solve(),
  // solve_device() and the ticks of mpc_run() take one launch.  Pools over solve_device ask with 1 (nmpc_amd.DDPSolverPool does).
  if(mode == 0 && (!s->queued_solve || s->cfg.max_iter < 64))
  {
    return false;
  }
  if(!s->ops->resumable_supported(s->B, s->cfg, s->d_params_batch ? 1 : 0))
  {
    return false;
  }
  caps->clear();
  for(int cap : {16, 32, 48, 64, 128, 256})
  {
    if(cap >= s->cfg.max_iter || (cap > 64 && s->B <= 4096))
    {
      break;
    }
    caps->push_back(cap);
  }
  caps->push_back(s->cfg.max_iter);
  return caps->size() > 1;
}

void freeRagged(nmpc_hip_ddp_solver * s)
{
  for(void ** p : {reinterpret_cast<void **>(&s->d_resume), reinterpret_cast<void **>(&s->d_ragged_words),
                   reinterpret_cast<void **>(&s->d_ragged_pairs), reinterpret_cast<void **>(&s->d_ragged_rank),
                   reinterpret_cast<void **>(&s->d_ragged_used)})
  {
    if(*p)
    {
      (void)hipFree(*p);
      *p = nullptr;
    }
  }
  s->ragged_ready = false;
}

int ensureRagged(nmpc_hip_ddp_solver * s)
{
  if(s->ragged_ready)
  {
    return NMPC_HIP_OK;
  }
  constexpr int R = nmpc_hip_ddp_solver::kRaggedMaxRounds;
  int rc = devAllocScalar(&s->d_resume, static_cast<size_t>(nmpc_amd::hip::kResumeRows) * s->Bp, s->elem);
  if(rc == NMPC_HIP_OK)
  {
    rc = devAlloc(&s->d_ragged_words, static_cast<size_t>(2 * R + 1));
  }
  if(rc == NMPC_HIP_OK)
  {
    rc = devAlloc(&s->d_ragged_pairs, static_cast<size_t>(R) * s->Bp);
  }
  if(rc == NMPC_HIP_OK)
  {
    rc = devAlloc(&s->d_ragged_rank, static_cast<size_t>(s->Bp));
  }
  if(rc == NMPC_HIP_OK)
  {
    rc = devAlloc(&s->d_ragged_used, static_cast<size_t>(s->Bp / 2 + 1));
  }
  if(rc != NMPC_HIP_OK)
  {
    // all five buffers or none: a later solve must not find d_resume set and the others null
    freeRagged(s);
    (void)hipGetLastError(); // the failed hipMalloc is handled here: the caller falls back to one whole-solve launch
    return rc;
  }
  // devAlloc clears with hipMemset, which is ordered on the NULL stream; the handle's stream is non-blocking and the first
  // launches of the schedule follow at once: without this wait the clear could land in the middle of them (once per handle)
  NMPC_HIP_TRY(hipDeviceSynchronize());
  s->ragged_ready = true;
  return NMPC_HIP_OK;
}

/** Every per-instance array of the handle (ragged_schedule.hpp: a compaction swap exchanges all of an instance's rows). */
nmpc_amd::hip::SwapTable swapTable(const nmpc_hip_ddp_solver * s)
{
  using nmpc_amd::hip::PerInstanceArray;
  nmpc_amd::hip::SwapTable t;
  const unsigned T = static_cast<unsigned>(s->T), N = static_cast<unsigned>(s->N), MM = static_cast<unsigned>(s->MM);
  const unsigned e = static_cast<unsigned>(s->elem);
  auto tile = [&](void * base, unsigned rows, unsigned elem, unsigned trace_unit = 0)
  {
    if(base != nullptr && t.n < nmpc_amd::hip::kMaxPerInstanceArrays)
    {
      t.a[t.n++] = PerInstanceArray{static_cast<char *>(base), rows, elem, 1u, trace_unit};
    }
  };
  auto major = [&](void * base, unsigned words4)
  {
    if(base != nullptr && t.n < nmpc_amd::hip::kMaxPerInstanceArrays)
    {
      t.a[t.n++] = PerInstanceArray{static_cast<char *>(base), words4, 4u, 0u, 0u};
    }
  };
  tile(s->d_t0, 1, e);
  tile(s->d_x0, N, e);
  tile(s->d_X, 2 * (T + 1) * N, e);
  tile(s->d_U, 2 * T * MM, e);
  tile(s->d_cost, 2 * (T + 1), e);
  tile(s->d_kff, T * MM, e);
  tile(s->d_Kfb, T * N * MM, e);
  tile(s->d_trace, static_cast<unsigned>(s->trace_rows) * NMPC_HIP_NTRACE, e, s->trace_rows > 1 ? NMPC_HIP_NTRACE : 0);
  tile(s->d_trace_last, NMPC_HIP_NTRACE, e);
  tile(s->d_dV, 2, e);
  tile(s->d_resume, nmpc_amd::hip::kResumeRows, e);
  tile(s->d_status, 1, 4);
  tile(s->d_sel, 1, 4); // (d_iters: exchanged by the compaction / replay-prepare kernels, which take the pairs' trace rows from it)
  tile(s->d_stream_id, 1, 4); // (streamed solves: which instance of the queue sits in the slot)
  tile(s->d_qp_ret, T, 4);
  tile(s->d_qp_free, T, 4);
  tile(s->d_input_dim, T, 4);
  major(s->d_phase_ticks, 4 * 2);
  major(s->d_params_batch, static_cast<unsigned>(s->params.size() / 4));
  major(s->d_lim_batch, 2 * nmpc_amd::hip::kMaxInputDim * 2);
  if(s->lim_steps_per_instance)
  {
    major(s->d_lim_steps, static_cast<unsigned>(s->lim_rows) * 2 * MM * 2);
  }
  return t;
}

/** The ragged-convergence schedule: resumable launches of iterations (caps[r-1], caps[r]] with a compaction between them, then
    the recorded swaps replayed in reverse.  All on stream st, no host round trip. */
int launchRagged(nmpc_hip_ddp_solver * s, hipStream_t st, DeviceBuffers buf, const std::vector<int> & caps, bool * fell_back)
{
  *fell_back = false;
  if(ensureRagged(s) != NMPC_HIP_OK)
  {
    // no memory for the schedule's buffers: this handle runs whole-solve launches from now on (the results are the same bits)
    s->ragged_unavailable = true;
    *fell_back = true;
    return NMPC_HIP_OK;
  }
  constexpr int R = nmpc_hip_ddp_solver::kRaggedMaxRounds;
  const int rounds = static_cast<int>(caps.size());
  int * n_active = s->d_ragged_words;
  int * n_swaps = s->d_ragged_words + R + 1;
  const nmpc_amd::hip::SwapTable tab = swapTable(s);
  const dim3 swap_grid(static_cast<unsigned>((s->Bp / 2 + 63) / 64), 64);
  const int wg_size = std::strcmp(s->ops->kernel_name(s->B, s->cfg), "ddp_solve_quad_kernel") == 0 ? 16 : 64;
  buf.resume = s->d_resume;
  hipLaunchKernelGGL(nmpc_amd::hip::ragged_init_kernel, dim3(1), dim3(64), 0, st, n_active, s->B);
  int swapped_rounds = 0; // rounds whose compaction swaps are queued: they are replayed in reverse whatever happens after them
  hipError_t le = hipSuccess;
  for(int r = 0; r < rounds; r++)
  {
    buf.n_active = n_active + r;
    buf.iter_begin = (r == 0) ? 1 : caps[r - 1] + 1;
    buf.iter_end = caps[r];
    le = s->ops->launch_solve(s->params.data(), s->cfg, buf, st);
    if(le != hipSuccess)
    {
      break;
    }
    if(r + 1 < rounds)
    {
      int * pairs = s->d_ragged_pairs + static_cast<size_t>(r) * s->Bp;
      hipLaunchKernelGGL((nmpc_amd::hip::ragged_compact_kernel<double>), dim3(1), dim3(1024), 0, st, s->d_resume, s->d_ragged_rank,
                         pairs, s->d_ragged_used, n_swaps + r, n_active + r, s->d_iters, wg_size);
      hipLaunchKernelGGL(nmpc_amd::hip::ragged_swap_kernel, swap_grid, dim3(256), 0, st, tab, pairs, s->d_ragged_used, n_swaps + r);
      swapped_rounds = r + 1;
    }
  }
  // A failed launch in the middle of the schedule still gets the reverse replay of the rounds already swapped: the swaps
  // moved every per-instance array of the handle — the persistent inputs too (per-instance limits and problem objects, x0, the
  // warm-start trajectories) — and later solves must find instance b at position b again.
  for(int r = swapped_rounds - 1; r >= 0; r--)
  {
    const int * pairs = s->d_ragged_pairs + static_cast<size_t>(r) * s->Bp;
    hipLaunchKernelGGL(nmpc_amd::hip::ragged_replay_prepare_kernel, dim3(static_cast<unsigned>((s->Bp / 2 + 255) / 256)), dim3(256), 0, st,
                       pairs, s->d_ragged_used, n_swaps + r, s->d_iters);
    hipLaunchKernelGGL(nmpc_amd::hip::ragged_swap_kernel, swap_grid, dim3(256), 0, st, tab, pairs, s->d_ragged_used, n_swaps + r);
  }
  if(le != hipSuccess)
  {
    (void)hipGetLastError();
    return fail(NMPC_HIP_ERR_HIP, std::string("resumable launch: ") + hipGetErrorString(le));
  }
  NMPC_HIP_TRY(hipGetLastError());
  s->last_ragged_rounds = rounds;
  return NMPC_HIP_OK;
}

/** A queue of n_total instances through the handle's B slots (stream_schedule.hpp).  in / out: device arrays in the reference
    layouts.  span: iterations per round. */
int launchStream(nmpc_hip_ddp_solver * s, hipStream_t st, const nmpc_amd::hip::StreamArrays & io, int n_total, int span)
{
  using namespace nmpc_amd::hip;
  int rc = ensureRagged(s);
  if(rc != NMPC_HIP_OK)
  {
    return rc;
  }
  if(!s->d_stream_id)
  {
    rc = devAlloc(&s->d_stream_id, static_cast<size_t>(s->Bp));
    if(rc == NMPC_HIP_OK)
    {
      rc = devAlloc(&s->d_stream_words, static_cast<size_t>(kSwCount));
    }
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    NMPC_HIP_TRY(hipDeviceSynchronize()); // (devAlloc's clears are ordered on the NULL stream)
  }
  nmpc_amd::hip::ScopedKnobs knobs_guard(&s->knobs);
  int * w = s->d_stream_words;
  DeviceBuffers buf = makeBuffers(s);
  buf.resume = s->d_resume;
  buf.n_active = w + kSwPrefix;
  buf.first_active = w + kSwFirst;
  buf.iter_begin = 1;
  buf.iter_end = span;
  const SwapTable tab = swapTable(s);
  const dim3 swap_grid(static_cast<unsigned>((s->Bp / 2 + 63) / 64), 64);
  const int wg_size = std::strcmp(s->ops->kernel_name(s->B, s->cfg), "ddp_solve_quad_kernel") == 0 ? 16 : 64;
  const dim3 fill_grid(static_cast<unsigned>(s->B), static_cast<unsigned>((s->T * s->MM + 1023) / 1024));
  struct TimingEvents
  {
    hipEvent_t a = nullptr, b = nullptr;
    ~TimingEvents()
    {
      if(a)
      {
        (void)hipEventDestroy(a);
      }
      if(b)
      {
        (void)hipEventDestroy(b);
      }
    }
  } tev;
  NMPC_HIP_TRY(hipEventCreate(&tev.a));
  NMPC_HIP_TRY(hipEventCreate(&tev.b));
  hipEvent_t ev0 = tev.a, ev1 = tev.b;
  NMPC_HIP_TRY(hipEventRecord(ev0, st));
  hipLaunchKernelGGL(stream_begin_kernel, dim3(1), dim3(64), 0, st, w, n_total);
  NMPC_HIP_TRY(hipMemsetAsync(s->d_stream_id, 0xff, static_cast<size_t>(s->Bp) * sizeof(int), st)); // every slot empty (-1)
  auto refill = [&]()
  {
    hipLaunchKernelGGL(stream_plan_kernel, dim3(1), dim3(64), 0, st, w, s->B);
    hipLaunchKernelGGL(stream_fill_kernel, fill_grid, dim3(256), 0, st, buf, io, s->d_stream_id, w, s->N, s->MM, s->d_t0, s->d_x0);
  };
  refill();
  buf.stream_mode = 2; // a round: the workgroups of the region filled last start with the initial rollout, the others resume
  hipError_t le = hipSuccess;
  int rounds = 0, done = 0;
  // an instance leaves after at most ceil(max_iter / span) rounds; a slot serves ceil(n_total / B) instances (+ slack for the rounds
  // in which too little finished for a compaction to pay)
  const long long max_rounds = 4ll * (static_cast<long long>((n_total + s->B - 1) / s->B) + 1) * ((std::max(s->cfg.max_iter, 1) + span - 1) / span + 1);
  // The host queues blocks of four rounds and looks at the count of finished instances behind each block — two blocks ahead of the
  // device, so that the queue never runs dry while the host waits (a round behind an empty queue costs a few empty launches).
  constexpr int kBlock = 4, kAhead = 2;
  struct HostSide // (released on every way out of this function)
  {
    int * done = nullptr;
    hipEvent_t ev[kAhead] = {nullptr, nullptr};
    ~HostSide()
    {
      for(hipEvent_t e : ev)
      {
        if(e)
        {
          (void)hipEventDestroy(e);
        }
      }
      if(done)
      {
        (void)hipHostFree(done);
      }
    }
  } host;
  NMPC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&host.done), kAhead * sizeof(int), hipHostMallocDefault));
  int * h_done = host.done;
  hipEvent_t * ev_blk = host.ev;
  for(int k = 0; k < kAhead; k++)
  {
    h_done[k] = 0;
    NMPC_HIP_TRY(hipEventCreateWithFlags(&ev_blk[k], hipEventDisableTiming));
  }
  long long queued_blocks = 0;
  while(le == hipSuccess && done < n_total && rounds < max_rounds)
  {
    const int slot = static_cast<int>(queued_blocks % kAhead);
    if(queued_blocks >= kAhead)
    {
      NMPC_HIP_TRY(hipEventSynchronize(ev_blk[slot])); // the block queued kAhead blocks ago
      done = h_done[slot];
      if(done >= n_total)
      {
        break;
      }
    }
    for(int k = 0; k < kBlock && le == hipSuccess; k++, rounds++)
    {
      le = s->ops->launch_solve(s->params.data(), s->cfg, buf, st);
      if(le != hipSuccess)
      {
        break;
      }
      hipLaunchKernelGGL(stream_extract_kernel, dim3(static_cast<unsigned>(s->B)), dim3(256), 0, st, buf, io, s->d_stream_id, w, s->N, s->MM);
      hipLaunchKernelGGL((ragged_compact_kernel<double>), dim3(1), dim3(1024), 0, st, s->d_resume, s->d_ragged_rank, s->d_ragged_pairs,
                         s->d_ragged_used, s->d_ragged_words, w + kSwPrefix, s->d_iters, wg_size);
      hipLaunchKernelGGL(ragged_swap_kernel, swap_grid, dim3(256), 0, st, tab, s->d_ragged_pairs, s->d_ragged_used, s->d_ragged_words);
      refill();
    }
    if(le != hipSuccess)
    {
      break;
    }
    NMPC_HIP_TRY(hipMemcpyAsync(&h_done[slot], w + kSwDone, sizeof(int), hipMemcpyDeviceToHost, st));
    NMPC_HIP_TRY(hipEventRecord(ev_blk[slot], st));
    queued_blocks++;
  }
  if(le == hipSuccess)
  {
    NMPC_HIP_TRY(hipStreamSynchronize(st));
    for(int k = 0; k < kAhead; k++)
    {
      done = h_done[k] > done ? h_done[k] : done;
    }
  }
  if(le != hipSuccess)
  {
    (void)hipGetLastError();
    return fail(NMPC_HIP_ERR_HIP, std::string("streamed solve: ") + hipGetErrorString(le));
  }
  NMPC_HIP_TRY(hipGetLastError());
  NMPC_HIP_TRY(hipEventRecord(ev1, st));
  NMPC_HIP_TRY(hipEventSynchronize(ev1));
  NMPC_HIP_TRY(hipEventElapsedTime(&s->stream_ms, ev0, ev1));
  s->stream_rounds = rounds;
  if(done < n_total)
  {
    return fail(NMPC_HIP_ERR_RUNTIME, "streamed solve: the queue did not drain (" + std::to_string(done) + " of " + std::to_string(n_total) + ")");
  }
  return NMPC_HIP_OK;
}

/** One solve on stream st, bracketed by the HIP-event triple of the timing ring.  ingest: convert the reference-layout
    device arrays into the solver's tile-major input buffers first (false: they are already in place, as after
    mpc_advance_kernel). */
int launchRecorded(nmpc_hip_ddp_solver * s,
                   hipStream_t st,
                   const double * d_t0,
                   const double * d_x0,
                   const double * d_u_init,
                   bool ingest,
                   bool timed = true)
{
  // timed = false: no event records around this solve (the inner ticks of the device-resident receding-horizon loop: three
  // event packets per 0.6 ms tick were 1 - 2 % of the loop; computationDuration() reports the loop's last solve)
  nmpc_amd::hip::ScopedKnobs knobs_guard(&s->knobs);
  s->last_stream = st;
  const int slot = static_cast<int>(s->n_solves % nmpc_hip_ddp_solver::kEvPool);
  if(timed)
  {
    int hrc = harvestSlot(s, slot); // only blocks when 128 solves are in flight
    if(hrc != NMPC_HIP_OK)
    {
      return hrc;
    }
    NMPC_HIP_TRY(hipEventRecord(s->ev_begin[slot], st));
  }
  if(ingest)
  {
    // reference layouts -> instance-minor device layout, one launch
    const int RU = s->T * s->MM;
    const dim3 grid((RU + 63) / 64 + (s->N + 63) / 64 + 1, s->Bp / 64);
    if(s->elem == 4)
    {
      hipLaunchKernelGGL((nmpc_amd::hip::ingest_kernel<double, float>), grid, dim3(256), 0, st, d_t0, reinterpret_cast<float *>(s->d_t0),
                         d_x0, reinterpret_cast<float *>(s->d_x0), s->N, d_u_init, reinterpret_cast<float *>(s->d_U), RU, s->B);
    }
    else
    {
      hipLaunchKernelGGL((nmpc_amd::hip::ingest_kernel<double, double>), grid, dim3(256), 0, st, d_t0, s->d_t0, d_x0, s->d_x0, s->N,
                         d_u_init, s->d_U, RU, s->B);
    }
    NMPC_HIP_TRY(hipGetLastError());
  }
  if(timed)
  {
    NMPC_HIP_TRY(hipEventRecord(s->ev_kernel[slot], st));
  }
  DeviceBuffers buf = makeBuffers(s);
  std::vector<int> caps;
  s->last_gain_layout = s->ops->gain_layout_of ? s->ops->gain_layout_of(s->B, s->cfg.with_input_constraint != 0 ? 1 : 0) : s->ops->gain_layout;
  s->last_ragged_rounds = 1;
  bool whole_solve = !raggedRounds(s, &caps);
  if(!whole_solve)
  {
    int rrc = launchRagged(s, st, buf, caps, &whole_solve);
    if(rrc != NMPC_HIP_OK)
    {
      return rrc;
    }
  }
  if(whole_solve)
  {
    const hipError_t le = s->ops->launch_solve(s->params.data(), s->cfg, buf, st);
    if(le == hipErrorNotSupported)
    {
      return fail(NMPC_HIP_ERR_RUNTIME,
                  s->elem == 4 ? "the fp32 tile kernel has no gain workspace on this handle (allocation failed at create)"
                               : "per-instance problem objects (set_model_params_batch) are served by the model's default kernel "
                                 "only; this solve needs the single-wavefront kernel");
    }
    NMPC_HIP_TRY(le);
  }
  if(timed)
  {
    NMPC_HIP_TRY(hipEventRecord(s->ev_end[slot], st));
    s->ev_pending[slot] = true;
    s->n_solves++;
  }
  s->solved = true;
  return NMPC_HIP_OK;
}
} // namespace

extern "C"
{
  int nmpc_hip_ddp_register_model(const ModelOps * ops)
  {
    if(!ops || !ops->name)
    {
      return NMPC_HIP_ERR_INVALID_ARGUMENT;
    }
    if(findModel(ops->name))
    {
      return NMPC_HIP_OK; // already known (same translation unit loaded twice)
    }
    registry().push_back(ops);
    return NMPC_HIP_OK;
  }

  const char * nmpc_hip_ddp_last_error(void)
  {
    return g_last_error.c_str();
  }

  int nmpc_hip_ddp_default_config(nmpc_hip_ddp_config * c)
  {
    if(!c)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "cfg is NULL");
    }
    std::memset(c, 0, sizeof(*c));
    c->with_input_constraint = 0;
    c->max_iter = 500;
    c->horizon_steps = 100;
    c->reg_type = 1;
    c->initial_lambda = 1e-4;
    c->initial_dlambda = 1.0;
    c->lambda_factor = 1.6;
    c->lambda_min = 1e-6;
    c->lambda_max = 1e10;
    c->k_rel_norm_thre = 1e-4;
    c->lambda_thre = 1e-5;
    c->cost_update_ratio_thre = 0;
    c->cost_update_thre = 1e-7;
    // alpha_list = 10^linspace(0, -3, 11)    (DDPSolver.h:50-60)
    c->n_alpha = 11;
    const double low = 0, high = -3;
    const double step = (high - low) / (c->n_alpha - 1);
    for(int i = 0; i < c->n_alpha; i++)
    {
      const double e = (i == c->n_alpha - 1) ? high : (low + i * step);
      c->alpha_list[i] = std::pow(10, e);
    }
    c->use_state_eq_second_derivative = 0;
    c->qp_max_iter = 500;
    c->qp_grad_thre = 1e-8;
    c->qp_rel_improve_thre = 1e-8;
    c->qp_step_factor = 0.6;
    c->qp_min_step = 1e-22;
    c->qp_armijo_param = 0.1;
    c->trace_level = 1;
    c->line_search_fan_out = 0;
    c->ragged_schedule = 0;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_model_count(void)
  {
    return static_cast<int>(registry().size());
  }

  int nmpc_hip_ddp_model_name(int index, const char ** name)
  {
    if(index < 0 || index >= static_cast<int>(registry().size()) || !name)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "model index out of range");
    }
    *name = registry()[index]->name;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_model_info(const char * model,
                              int * state_dim,
                              int * input_dim_max,
                              int * dynamic_input,
                              size_t * param_bytes)
  {
    const ModelOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown model: ") + (model ? model : "(null)"));
    }
    if(state_dim)
    {
      *state_dim = m->state_dim;
    }
    if(input_dim_max)
    {
      *input_dim_max = m->input_dim_max;
    }
    if(dynamic_input)
    {
      *dynamic_input = m->dynamic_input;
    }
    if(param_bytes)
    {
      *param_bytes = m->param_bytes;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_model_scalar_bytes(const char * model, int * bytes)
  {
    const ModelOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown model: ") + (model ? model : "(null)"));
    }
    if(!bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "bytes is NULL");
    }
    *bytes = m->scalar_bytes;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_model_default_params(const char * model, void * out, size_t bytes)
  {
    const ModelOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown model: ") + (model ? model : "(null)"));
    }
    if(!out || bytes != m->param_bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "params size should be " + std::to_string(m->param_bytes) + " but "
                                                     + std::to_string(bytes) + ".");
    }
    m->default_params(out);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_create(const char * model, int horizon_steps, int batch, int device, nmpc_hip_ddp_handle * out)
  {
    if(!out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "out handle is NULL");
    }
    *out = nullptr;
    const ModelOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown model: ") + (model ? model : "(null)"));
    }
    if(horizon_steps < 1 || batch < 1)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "horizon_steps and batch must be positive");
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if(e != hipSuccess || n_dev < 1)
    {
      return fail(NMPC_HIP_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e)
                                              + "); this library has no CPU fallback");
    }
    if(device < 0 || device >= n_dev)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "device index out of range");
    }
    NMPC_HIP_TRY(hipSetDevice(device));

    nmpc_hip_ddp_solver * s = new nmpc_hip_ddp_solver();
    s->ops = m;
    s->device = device;
    s->B = batch;
    s->Bp = ((batch + 63) / 64) * 64;
    s->T = horizon_steps;
    s->N = m->state_dim;
    s->M = m->input_dim_max;
    s->MM = m->input_dim_max > 0 ? m->input_dim_max : 1;
    s->elem = m->scalar_bytes;
    nmpc_hip_ddp_default_config(&s->cfg);
    s->cfg.horizon_steps = horizon_steps;
    if(const char * e = std::getenv("NMPC_HIP_DDP_RAGGED")) // (read once per handle: A/B measurements of the ragged schedule)
    {
      s->ragged_env = (std::strcmp(e, "0") == 0) ? -1 : 1;
    }
    s->knobs = nmpc_amd::hip::LaunchKnobs::fromEnvironment(); // developer overrides: read ONCE, here (not on the launch path)
    nmpc_amd::hip::ScopedKnobs knobs_guard(&s->knobs);
    s->params.resize(m->param_bytes);
    m->default_params(s->params.data());
    for(int i = 0; i < nmpc_amd::hip::kMaxInputDim; i++)
    {
      s->lim_lo[i] = -INFINITY;
      s->lim_hi[i] = INFINITY;
    }

    const size_t Bp = s->Bp, T = s->T, N = s->N, MM = s->MM;
    int rc = NMPC_HIP_OK;
    auto chk = [&](int r)
    {
      if(rc == NMPC_HIP_OK)
      {
        rc = r;
      }
    };
    chk(devAllocScalar(&s->d_t0, Bp, s->elem));
    chk(devAllocScalar(&s->d_x0, N * Bp, s->elem));
    chk(devAllocScalar(&s->d_X, 2 * (T + 1) * N * Bp, s->elem));
    chk(devAllocScalar(&s->d_U, 2 * T * MM * Bp, s->elem));
    chk(devAllocScalar(&s->d_cost, 2 * (T + 1) * Bp, s->elem));
    chk(devAllocScalar(&s->d_kff, T * MM * Bp, s->elem));
    chk(devAllocScalar(&s->d_Kfb, T * N * MM * Bp, s->elem));
    chk(devAllocScalar(&s->d_trace_last, static_cast<size_t>(NMPC_HIP_NTRACE) * Bp, s->elem));
    chk(devAllocScalar(&s->d_dV, 2 * Bp, s->elem));
    chk(devAlloc(&s->d_status, Bp));
    chk(devAlloc(&s->d_iters, Bp));
    chk(devAlloc(&s->d_sel, Bp));
    chk(devAlloc(&s->d_qp_ret, T * Bp));
    chk(devAlloc(&s->d_qp_free, T * Bp));
    chk(devAlloc(&s->d_input_dim, T * Bp));
    chk(devAlloc(&s->d_phase_ticks, 4 * Bp));
    // (NMPC_HIP_DDP_NO_WORKSPACE=1: developer switch for the tests of the path a failed workspace allocation takes)
    const char * no_ws = std::getenv("NMPC_HIP_DDP_NO_WORKSPACE");
    if(no_ws && std::strcmp(no_ws, "0") != 0)
    {
      s->knobs.have_workspace = 0;
    }
    else if(m->wpi_workspace_doubles(s->T) > 0)
    {
      // wave-per-instance kernel (9 <= n <= 16): materialised derivatives, gains and one candidate trajectory per
      // step size, per instance; quad-kernel shapes: three candidate trajectories per instance, tile-major (the line
      // search's fan-out scratch).  No memset: the kernels write everything they read.
      if(hipMalloc(reinterpret_cast<void **>(&s->d_wpi_ws),
                   m->wpi_workspace_doubles(s->T) * static_cast<size_t>(s->Bp) * static_cast<size_t>(s->elem))
         != hipSuccess)
      {
        (void)hipGetLastError();
        s->d_wpi_ws = nullptr; // not enough memory for the workspace: the lane-per-instance kernel needs none
        s->knobs.have_workspace = 0; // (kernel_name / gain_layout_of / launch_solve then choose among the kernels that need none)
      }
    }
    if(rc == NMPC_HIP_OK)
    {
      rc = allocTrace(s);
    }
    if(rc == NMPC_HIP_OK && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)
    {
      rc = fail(NMPC_HIP_ERR_HIP, "hipStreamCreate failed");
    }
    for(int i = 0; i < nmpc_hip_ddp_solver::kEvPool && rc == NMPC_HIP_OK; i++)
    {
      if(hipEventCreate(&s->ev_begin[i]) != hipSuccess || hipEventCreate(&s->ev_kernel[i]) != hipSuccess
         || hipEventCreate(&s->ev_end[i]) != hipSuccess)
      {
        rc = fail(NMPC_HIP_ERR_HIP, "hipEventCreate failed");
      }
    }
    if(rc != NMPC_HIP_OK)
    {
      std::string keep = g_last_error;
      nmpc_hip_ddp_destroy(s);
      g_last_error = keep;
      return rc;
    }
    // devAlloc clears every buffer with hipMemset, which is ordered on the NULL stream; the handle's stream is non-blocking, and a
    // caller in compiled code launches its first solve microseconds from here: the clears must have landed (once per handle)
    NMPC_HIP_TRY(hipDeviceSynchronize());
    *out = s;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_request_hw_queues(int n, int * took_effect)
  {
    if(n < 1 || n > 64)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "hardware queues: 1 .. 64");
    }
    const char * cur = std::getenv("GPU_MAX_HW_QUEUES");
    const bool up = runtimeInitialised();
    int effective = 0;
    if(up)
    {
      // too late to change: what the runtime read is what the variable held then (unset: its default of 4)
      effective = (cur && std::atoi(cur) >= n) ? 1 : 0;
    }
    else
    {
      if(!cur || std::atoi(cur) < n)
      {
        setenv("GPU_MAX_HW_QUEUES", std::to_string(n).c_str(), 1);
      }
      effective = 1;
    }
    if(took_effect)
    {
      *took_effect = effective;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_destroy(nmpc_hip_ddp_handle s)
  {
    if(!s)
    {
      return NMPC_HIP_OK;
    }
    (void)hipSetDevice(s->device);
    if(s->stream)
    {
      (void)hipStreamSynchronize(s->stream);
    }
    void * ptrs[] = {s->d_t0,     s->d_x0,     s->d_X,   s->d_U,      s->d_cost,    s->d_kff,       s->d_Kfb,
                     s->d_trace,  s->d_trace_last, s->d_dV, s->d_status, s->d_iters, s->d_sel,     s->d_qp_ret,
                     s->d_qp_free, s->d_input_dim, s->d_wpi_ws, s->d_params_batch, s->d_lim_batch, s->d_lim_steps, s->d_phase_ticks, s->d_stage_in,
                     s->d_stage_out, s->d_resume, s->d_ragged_words, s->d_ragged_pairs, s->d_ragged_rank, s->d_ragged_used,
                     s->d_stream_id, s->d_stream_words, s->d_stream_io};
    for(void * p : ptrs)
    {
      if(p)
      {
        (void)hipFree(p);
      }
    }
    for(int i = 0; i < nmpc_hip_ddp_solver::kEvPool; i++)
    {
      if(s->ev_begin[i])
      {
        (void)hipEventDestroy(s->ev_begin[i]);
      }
      if(s->ev_kernel[i])
      {
        (void)hipEventDestroy(s->ev_kernel[i]);
      }
      if(s->ev_end[i])
      {
        (void)hipEventDestroy(s->ev_end[i]);
      }
    }
    if(s->ev_staged)
    {
      (void)hipEventDestroy(s->ev_staged);
    }
    if(s->stream)
    {
      (void)hipStreamDestroy(s->stream);
    }
    delete s;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_config(nmpc_hip_ddp_handle s, const nmpc_hip_ddp_config * cfg)
  {
    if(!s || !cfg)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or cfg");
    }
    int rc = checkConfig(s, cfg);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    s->cfg = *cfg; // kernels capture the configuration by value at launch, so no synchronisation is needed
    return allocTrace(s); // (re)allocates, after a device synchronise, only when the trace size changes
  }

  int nmpc_hip_ddp_get_config(nmpc_hip_ddp_handle s, nmpc_hip_ddp_config * cfg)
  {
    if(!s || !cfg)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or cfg");
    }
    *cfg = s->cfg;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_model_params(nmpc_hip_ddp_handle s, const void * params, size_t bytes)
  {
    if(!s || !params)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or params");
    }
    if(bytes != s->ops->param_bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "params size should be " + std::to_string(s->ops->param_bytes)
                                                     + " but " + std::to_string(bytes) + ".");
    }
    std::memcpy(s->params.data(), params, bytes);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_input_limits_batch(nmpc_hip_ddp_handle s, const double * lower, const double * upper)
  {
    if(!s || (lower == nullptr) != (upper == nullptr))
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, or only one of lower / upper given");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
    if(!lower)
    {
      if(s->d_lim_batch)
      {
        NMPC_HIP_TRY(hipFree(s->d_lim_batch));
        s->d_lim_batch = nullptr;
      }
      s->has_limits = s->has_shared_limits || s->d_lim_steps != nullptr; // back to what nmpc_hip_ddp_set_input_limits gave, if anything
      return NMPC_HIP_OK;
    }
    constexpr int kMax = nmpc_amd::hip::kMaxInputDim;
    std::vector<double> host(static_cast<size_t>(s->Bp) * 2 * kMax);
    for(int b = 0; b < s->Bp; b++)
    {
      for(int a = 0; a < kMax; a++)
      {
        const bool valid = b < s->B && a < s->MM;
        host[(static_cast<size_t>(b) * 2 + 0) * kMax + a] = valid ? lower[static_cast<size_t>(b) * s->MM + a] : -INFINITY;
        host[(static_cast<size_t>(b) * 2 + 1) * kMax + a] = valid ? upper[static_cast<size_t>(b) * s->MM + a] : INFINITY;
      }
    }
    if(!s->d_lim_batch)
    {
      NMPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_lim_batch), host.size() * sizeof(double)));
    }
    NMPC_HIP_TRY(hipMemcpy(s->d_lim_batch, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    s->has_limits = true;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_model_params_batch(nmpc_hip_ddp_handle s, const void * params, size_t bytes_per_instance)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    if(!params)
    {
      if(s->d_params_batch)
      {
        NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
        NMPC_HIP_TRY(hipFree(s->d_params_batch));
        s->d_params_batch = nullptr;
      }
      return NMPC_HIP_OK;
    }
    nmpc_amd::hip::ScopedKnobs knobs_guard(&s->knobs);
    if(s->ops->own_problems_supported && !s->ops->own_problems_supported(s->B, s->cfg.with_input_constraint != 0 ? 1 : 0))
    {
      return fail(NMPC_HIP_ERR_RUNTIME, std::string("the kernel this handle solves on (") + s->ops->kernel_name(s->B, s->cfg)
                                            + ") has no instantiation with one problem object per instance");
    }
    const size_t pb = s->ops->param_bytes;
    if(bytes_per_instance != pb)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "params size per instance should be " + std::to_string(pb) + " but "
                                                     + std::to_string(bytes_per_instance) + ".");
    }
    // dt() shapes the batch: it has to be the shared object's for every instance
    const double dt_shared = s->ops->dt(s->params.data());
    const unsigned char * src = static_cast<const unsigned char *>(params);
    for(int b = 0; b < s->B; b++)
    {
      if(s->ops->dt(src + static_cast<size_t>(b) * pb) != dt_shared)
      {
        return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "instance " + std::to_string(b) + " has a different dt() than the "
                                                       "handle's shared problem object");
      }
    }
    if(!s->d_params_batch)
    {
      NMPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_params_batch), static_cast<size_t>(s->Bp) * pb));
    }
    // padding lanes (b >= B) run on the shared object
    std::vector<unsigned char> host(static_cast<size_t>(s->Bp) * pb);
    std::memcpy(host.data(), src, static_cast<size_t>(s->B) * pb);
    for(int b = s->B; b < s->Bp; b++)
    {
      std::memcpy(host.data() + static_cast<size_t>(b) * pb, s->params.data(), pb);
    }
    NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
    NMPC_HIP_TRY(hipMemcpy(s->d_params_batch, host.data(), host.size(), hipMemcpyHostToDevice));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_input_dims(nmpc_hip_ddp_handle s, double t0, int * out)
  {
    if(!s || !out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output");
    }
    s->ops->input_dims(s->params.data(), t0, s->T, out);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_input_limits(nmpc_hip_ddp_handle s, const double * lower, const double * upper)
  {
    if(!s || !lower || !upper)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or limits");
    }
    for(int i = 0; i < s->MM; i++)
    {
      s->lim_lo[i] = lower[i];
      s->lim_hi[i] = upper[i];
    }
    s->has_limits = true;
    s->has_shared_limits = true;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_input_limits_schedule(nmpc_hip_ddp_handle s, const double * lower, const double * upper, int rows, int per_instance)
  {
    if(!s || (lower == nullptr) != (upper == nullptr))
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, or only one of lower / upper given");
    }
    if(lower && rows < s->T)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "a limits table needs at least horizon_steps rows");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
    if(!lower)
    {
      if(s->d_lim_steps)
      {
        NMPC_HIP_TRY(hipFree(s->d_lim_steps));
        s->d_lim_steps = nullptr;
      }
      s->lim_steps_per_instance = 0;
      s->lim_rows = 0;
      s->has_limits = s->has_shared_limits || s->d_lim_batch != nullptr;
      return NMPC_HIP_OK;
    }
    const size_t per_table = static_cast<size_t>(rows) * s->MM;
    const size_t n_tables = per_instance ? static_cast<size_t>(s->Bp) : 1;
    // device layout [table][rows][2][MM]; padding instances (b >= B) are unbounded
    std::vector<double> host(n_tables * per_table * 2);
    for(size_t tb = 0; tb < n_tables; tb++)
    {
      const bool valid = !per_instance || tb < static_cast<size_t>(s->B);
      for(int i = 0; i < rows; i++)
      {
        for(int a = 0; a < s->MM; a++)
        {
          const size_t src = (per_instance ? tb * per_table : 0) + static_cast<size_t>(i) * s->MM + a;
          host[((tb * rows + i) * 2 + 0) * s->MM + a] = valid ? lower[src] : -INFINITY;
          host[((tb * rows + i) * 2 + 1) * s->MM + a] = valid ? upper[src] : INFINITY;
        }
      }
    }
    if(s->d_lim_steps)
    {
      NMPC_HIP_TRY(hipFree(s->d_lim_steps));
      s->d_lim_steps = nullptr;
    }
    NMPC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_lim_steps), host.size() * sizeof(double)));
    NMPC_HIP_TRY(hipMemcpy(s->d_lim_steps, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    s->lim_steps_per_instance = per_instance ? 1 : 0;
    s->lim_rows = rows;
    s->lim_offset = 0;
    s->has_limits = true;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_input_limits_horizon(nmpc_hip_ddp_handle s, const double * lower, const double * upper, int per_instance)
  {
    return nmpc_hip_ddp_set_input_limits_schedule(s, lower, upper, s ? s->T : 0, per_instance);
  }

  int nmpc_hip_ddp_solve_device(nmpc_hip_ddp_handle s,
                                const double * d_t0,
                                const double * d_x0,
                                const double * d_u_init,
                                void * stream)
  {
    if(!s || !d_x0 || !d_u_init)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, x0 or u_init");
    }
    if(s->cfg.with_input_constraint && !s->has_limits)
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "with_input_constraint is set but no input limits were given "
                                        "(setInputLimitsFunc, DDPSolver.h:282-285)");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    return launchRecorded(s, st, d_t0, d_x0, d_u_init, true);
  }

  int nmpc_hip_ddp_mpc_default_options(nmpc_hip_ddp_mpc_options * opt)
  {
    if(!opt)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL options");
    }
    opt->n_ticks = 1;
    opt->shift_warm_start = 1;
    opt->max_iter_after_first = 0;
    opt->sim_substeps = 0;
    opt->sim_dt = 0.0;
    opt->clamp_u0 = 1;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_mpc_run(nmpc_hip_ddp_handle s,
                           const double * t0,
                           const double * x0,
                           const double * u_init,
                           const nmpc_hip_ddp_mpc_options * opt,
                           double * t_log,
                           double * x_log,
                           double * u0_log,
                           int * iter_log,
                           int * status_log,
                           int * m0_log,
                           double * x_final,
                           double * t_final)
  {
    if(!s || !x0 || !u_init || !opt)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, x0, u_init or options");
    }
    if(opt->n_ticks < 1)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "n_ticks should be >= 1");
    }
    if(s->d_lim_steps && opt->n_ticks > 1 && !(opt->shift_warm_start && s->lim_rows >= s->T + opt->n_ticks - 1))
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "time-varying input limits in the device-resident loop: the shift pattern with a table of "
                                        "horizon_steps + n_ticks - 1 rows (nmpc_hip_ddp_set_input_limits_schedule); the plant "
                                        "pattern advances current_t by sim_dt and takes limits that are constant in time");
    }
    if(!opt->shift_warm_start)
    {
      if(!s->ops->has_plant_step)
      {
        return fail(NMPC_HIP_ERR_INVALID_ARGUMENT,
                    "the plant pattern needs a problem type with stateEq(t, x, u, dt) (TestDDPCartPole.cpp:63-98)");
      }
      if(opt->sim_substeps < 1 || !(opt->sim_dt > 0))
      {
        return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "the plant pattern needs sim_substeps >= 1 and sim_dt > 0");
      }
      if(opt->clamp_u0 && !s->has_limits)
      {
        return fail(NMPC_HIP_ERR_RUNTIME, "clamp_u0 is set but no input limits were given");
      }
    }
    if(s->cfg.with_input_constraint && !s->has_limits)
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "with_input_constraint is set but no input limits were given "
                                        "(setInputLimitsFunc, DDPSolver.h:282-285)");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    const size_t B = static_cast<size_t>(s->B), nt = static_cast<size_t>(opt->n_ticks);
    const size_t nx = B * s->N, nu = B * s->T * s->MM;
    // staging: inputs in the reference layout, then the logs
    const size_t in_bytes = (nx + nu + B) * sizeof(double);
    const size_t log_d = B * nt * (1 + s->N + s->MM); // doubles: t, x, u0
    const size_t log_i = B * nt * 3; // ints: iterations, status, m0
    int rc = ensureStage(&s->d_stage_in, &s->stage_in_bytes, in_bytes);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    rc = ensureStage(&s->d_stage_out, &s->stage_out_bytes,
                     std::max(log_d * sizeof(double) + log_i * sizeof(int), nx * sizeof(double)) + 8 + static_cast<size_t>(s->Bp) * sizeof(double));
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    double * dx = static_cast<double *>(s->d_stage_in);
    double * du = dx + nx;
    double * dt = du + nu;
    NMPC_HIP_TRY(hipMemcpyAsync(dx, x0, nx * sizeof(double), hipMemcpyHostToDevice, st));
    NMPC_HIP_TRY(hipMemcpyAsync(du, u_init, nu * sizeof(double), hipMemcpyHostToDevice, st));
    if(t0)
    {
      NMPC_HIP_TRY(hipMemcpyAsync(dt, t0, B * sizeof(double), hipMemcpyHostToDevice, st));
    }
    nmpc_amd::hip::MpcAdvanceArgs args;
    args.n_ticks = opt->n_ticks;
    args.shift_warm_start = opt->shift_warm_start;
    args.sim_substeps = opt->sim_substeps;
    args.sim_dt = opt->sim_dt;
    args.clamp_u0 = opt->clamp_u0;
    args.t0 = s->d_t0;
    args.x0 = s->d_x0;
    args.t_log = static_cast<double *>(s->d_stage_out);
    args.x_log = args.t_log + B * nt;
    args.u0_log = args.x_log + B * nt * s->N;
    args.iter_log = reinterpret_cast<int *>(args.u0_log + B * nt * s->MM);
    args.status_log = args.iter_log + B * nt;
    args.m0_log = args.status_log + B * nt;
    // current_t per instance in double, carried from tick to tick (behind the logs, 8-byte aligned: the logs are 3 B nt ints)
    args.t_exact = reinterpret_cast<double *>(static_cast<char *>(s->d_stage_out)
                                              + ((std::max(log_d * sizeof(double) + log_i * sizeof(int), nx * sizeof(double)) + 7) / 8) * 8);
    const int max_iter_saved = s->cfg.max_iter;
    for(int tick = 0; tick < opt->n_ticks && rc == NMPC_HIP_OK; tick++)
    {
      s->lim_offset = s->d_lim_steps ? tick : 0; // row of this tick's timestep 0 in the limits schedule
      rc = launchRecorded(s, st, t0 ? dt : nullptr, dx, du, tick == 0, tick == 0 || tick == opt->n_ticks - 1);
      if(rc != NMPC_HIP_OK)
      {
        break;
      }
      if(tick == 0 && opt->max_iter_after_first > 0)
      {
        s->cfg.max_iter = std::min(opt->max_iter_after_first, max_iter_saved); // the trace buffer was sized for it
      }
      args.tick = tick;
      const DeviceBuffers buf = makeBuffers(s);
      hipError_t e = s->ops->launch_mpc_advance(s->params.data(), buf, args, st);
      if(e != hipSuccess)
      {
        rc = fail(NMPC_HIP_ERR_HIP, std::string("mpc_advance_kernel: ") + hipGetErrorString(e));
      }
    }
    s->cfg.max_iter = max_iter_saved;
    s->lim_offset = 0;
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    auto out = [&](void * host, const void * dev, size_t bytes) -> hipError_t {
      return host ? hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    NMPC_HIP_TRY(out(t_log, args.t_log, B * nt * sizeof(double)));
    NMPC_HIP_TRY(out(x_log, args.x_log, B * nt * s->N * sizeof(double)));
    NMPC_HIP_TRY(out(u0_log, args.u0_log, B * nt * s->MM * sizeof(double)));
    NMPC_HIP_TRY(out(iter_log, args.iter_log, B * nt * sizeof(int)));
    NMPC_HIP_TRY(out(status_log, args.status_log, B * nt * sizeof(int)));
    NMPC_HIP_TRY(out(m0_log, args.m0_log, B * nt * sizeof(int)));
    // the logs have been queued for copy-out on the same stream: the input staging buffer can be reused behind them
    // (the handle's t0 / x0 are arrays of the problem's Scalar: converted on the way out)
    if(t_final)
    {
      NMPC_HIP_TRY(scalarToMajor(s, s->d_t0, dt, nullptr, 1, 1, st));
      NMPC_HIP_TRY(hipMemcpyAsync(t_final, dt, B * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if(x_final)
    {
      NMPC_HIP_TRY(scalarToMajor(s, s->d_x0, dx, nullptr, s->N, 1, st));
      NMPC_HIP_TRY(hipMemcpyAsync(x_final, dx, nx * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    NMPC_HIP_TRY(hipStreamSynchronize(st));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_synchronize(nmpc_hip_ddp_handle s)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    NMPC_HIP_TRY(hipStreamSynchronize(s->last_stream ? s->last_stream : s->stream));
    return NMPC_HIP_OK;
  }

  static int stageAndLaunch(nmpc_hip_ddp_handle s, const double * t0, const double * x0, const double * u_init, bool queued);

  int nmpc_hip_ddp_solve(nmpc_hip_ddp_handle s, const double * t0, const double * x0, const double * u_init)
  {
    int rc = stageAndLaunch(s, t0, x0, u_init, false);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_solve_async(nmpc_hip_ddp_handle s, const double * t0, const double * x0, const double * u_init)
  {
    return stageAndLaunch(s, t0, x0, u_init, true);
  }

  static int stageAndLaunch(nmpc_hip_ddp_handle s, const double * t0, const double * x0, const double * u_init, bool queued)
  {
    if(!s || !x0 || !u_init)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, x0 or u_init");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    const size_t nx = static_cast<size_t>(s->B) * s->N;
    const size_t nu = static_cast<size_t>(s->B) * s->T * s->MM;
    const size_t nt = static_cast<size_t>(s->B);
    int rc = ensureStage(&s->d_stage_in, &s->stage_in_bytes, (nx + nu + nt) * sizeof(double));
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    double * dx = static_cast<double *>(s->d_stage_in);
    double * du = dx + nx;
    double * dt = du + nu;
    NMPC_HIP_TRY(hipMemcpyAsync(dx, x0, nx * sizeof(double), hipMemcpyHostToDevice, s->stream));
    NMPC_HIP_TRY(hipMemcpyAsync(du, u_init, nu * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if(t0)
    {
      NMPC_HIP_TRY(hipMemcpyAsync(dt, t0, nt * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    // The host arrays may be reused when this call returns: wait for the staging copies (pageable memory: the runtime may pin the
    // caller's pages and copy later) — for them only, the solve behind them stays queued.
    if(!s->ev_staged)
    {
      NMPC_HIP_TRY(hipEventCreateWithFlags(&s->ev_staged, hipEventDisableTiming));
    }
    NMPC_HIP_TRY(hipEventRecord(s->ev_staged, s->stream));
    s->queued_solve = queued;
    rc = nmpc_hip_ddp_solve_device(s, t0 ? dt : nullptr, dx, du, nullptr);
    s->queued_solve = false;
    NMPC_HIP_TRY(hipEventSynchronize(s->ev_staged));
    return rc;
  }

  int nmpc_hip_ddp_solve_stream(nmpc_hip_ddp_handle s, int n_instances, const double * t0, const double * x0, const double * u_init, int span)
  {
    if(!s || !x0 || !u_init || n_instances < 1)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle, x0 or u_init, or no instances");
    }
    if(s->cfg.with_input_constraint && !s->has_limits)
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "with_input_constraint is set but no input limits were given "
                                        "(setInputLimitsFunc, DDPSolver.h:282-285)");
    }
    if(s->elem != 8 || s->ops->resumable_supported == nullptr || !s->ops->resumable_supported(s->B, s->cfg, s->d_params_batch ? 1 : 0))
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "a streamed solve needs a kernel family with resumable launches: the quad / two-wave kernels "
                                        "(n <= 4, one input, fp64) with a shared problem object");
    }
    if(s->d_lim_batch != nullptr || s->lim_steps_per_instance != 0 || s->d_lim_steps != nullptr)
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "a streamed solve takes input limits shared by all instances and constant along the horizon");
    }
    if(span <= 0)
    {
      span = 8; // [measured, profiles/r06_stream_throughput.txt: 4 / 6 / 8 / 12 / 16 iterations per round: 10.3 / 10.8 / 10.9 / 10.5 / 10.1 k]
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    NMPC_HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t N = static_cast<size_t>(n_instances), T = static_cast<size_t>(s->T), n = static_cast<size_t>(s->N), mm = static_cast<size_t>(s->MM);
    const size_t n_t0 = N, n_x0 = N * n, n_u = N * T * mm, n_X = N * (T + 1) * n, n_c = N * (T + 1), n_tr = N * NMPC_HIP_NTRACE, n_dv = N * 2;
    const size_t doubles = n_t0 + n_x0 + n_u + n_X + n_u + n_c + n_tr + n_dv;
    const size_t bytes = doubles * sizeof(double) + 2 * N * sizeof(int);
    if(bytes > s->stream_io_bytes)
    {
      if(s->d_stream_io)
      {
        NMPC_HIP_TRY(hipFree(s->d_stream_io));
        s->d_stream_io = nullptr;
        s->stream_io_bytes = 0;
      }
      NMPC_HIP_TRY(hipMalloc(&s->d_stream_io, bytes));
      s->stream_io_bytes = bytes;
    }
    s->stream_n = 0;
    double * d = static_cast<double *>(s->d_stream_io);
    nmpc_amd::hip::StreamArrays io;
    double * d_t0 = d;
    double * d_x0 = d_t0 + n_t0;
    double * d_u = d_x0 + n_x0;
    io.t0 = t0 ? d_t0 : nullptr;
    io.x0 = d_x0;
    io.u_init = d_u;
    io.X = d_u + n_u;
    io.U = io.X + n_X;
    io.cost = io.U + n_u;
    io.trace_last = io.cost + n_c;
    io.dV = io.trace_last + n_tr;
    io.status = reinterpret_cast<int *>(io.dV + n_dv);
    io.iters = io.status + N;
    if(t0)
    {
      NMPC_HIP_TRY(hipMemcpyAsync(d_t0, t0, n_t0 * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    NMPC_HIP_TRY(hipMemcpyAsync(d_x0, x0, n_x0 * sizeof(double), hipMemcpyHostToDevice, s->stream));
    NMPC_HIP_TRY(hipMemcpyAsync(d_u, u_init, n_u * sizeof(double), hipMemcpyHostToDevice, s->stream));
    int rc = launchStream(s, s->stream, io, n_instances, span);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    s->stream_arrays = io;
    s->stream_n = n_instances;
    s->solved = true;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_stream_get(nmpc_hip_ddp_handle s, int field, void * out, size_t bytes)
  {
    if(!s || !out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output");
    }
    if(s->stream_n <= 0)
    {
      return fail(NMPC_HIP_ERR_RUNTIME, "no streamed solve on this handle");
    }
    const size_t N = static_cast<size_t>(s->stream_n), T = static_cast<size_t>(s->T), n = static_cast<size_t>(s->N), mm = static_cast<size_t>(s->MM);
    const void * src = nullptr;
    size_t want = 0;
    switch(field)
    {
      case NMPC_HIP_FIELD_X: src = s->stream_arrays.X; want = N * (T + 1) * n * sizeof(double); break;
      case NMPC_HIP_FIELD_U: src = s->stream_arrays.U; want = N * T * mm * sizeof(double); break;
      case NMPC_HIP_FIELD_COST: src = s->stream_arrays.cost; want = N * (T + 1) * sizeof(double); break;
      case NMPC_HIP_FIELD_TRACE_LAST: src = s->stream_arrays.trace_last; want = N * NMPC_HIP_NTRACE * sizeof(double); break;
      case NMPC_HIP_FIELD_DV: src = s->stream_arrays.dV; want = N * 2 * sizeof(double); break;
      case NMPC_HIP_FIELD_STATUS: src = s->stream_arrays.status; want = N * sizeof(int); break;
      case NMPC_HIP_FIELD_ITERS: src = s->stream_arrays.iters; want = N * sizeof(int); break;
      default: return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "a streamed solve returns X, U, cost, status, iterations, the last trace row and dV");
    }
    if(bytes != want)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "output size: " + std::to_string(want) + " bytes expected");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    NMPC_HIP_TRY(hipMemcpy(out, src, want, hipMemcpyDeviceToHost));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_last_stream_stats(nmpc_hip_ddp_handle s, int * rounds, float * ms)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    if(rounds)
    {
      *rounds = s->stream_rounds;
    }
    if(ms)
    {
      *ms = s->stream_ms;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_field_bytes(nmpc_hip_ddp_handle s, int field, size_t * bytes)
  {
    if(!s || !bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or bytes");
    }
    FieldInfo fi;
    int rc = fieldInfo(s, field, &fi);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    *bytes = fi.rows * fi.elem * static_cast<size_t>(s->B);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_get_device(nmpc_hip_ddp_handle s, int field, void * d_out, size_t bytes, void * stream)
  {
    if(!s || !d_out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output");
    }
    if(!s->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "solve() has not been called");
    }
    size_t need = 0;
    int rc = nmpc_hip_ddp_field_bytes(s, field, &need);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    if(bytes != need)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT,
                  "field size should be " + std::to_string(need) + " but " + std::to_string(bytes) + ".");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : (s->last_stream ? s->last_stream : s->stream);
    return packField(s, field, d_out, st);
  }

  int nmpc_hip_ddp_get(nmpc_hip_ddp_handle s, int field, void * out, size_t bytes)
  {
    if(!s || !out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output");
    }
    if(!s->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "solve() has not been called");
    }
    size_t need = 0;
    int rc = nmpc_hip_ddp_field_bytes(s, field, &need);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    if(bytes != need)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT,
                  "field size should be " + std::to_string(need) + " but " + std::to_string(bytes) + ".");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->last_stream ? s->last_stream : s->stream;
    rc = ensureStage(&s->d_stage_out, &s->stage_out_bytes, need);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    rc = packField(s, field, s->d_stage_out, st);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    NMPC_HIP_TRY(hipMemcpyAsync(out, s->d_stage_out, need, hipMemcpyDeviceToHost, st));
    NMPC_HIP_TRY(hipStreamSynchronize(st));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_last_solve_ms(nmpc_hip_ddp_handle s, float * total_ms, float * kernel_ms)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    if(!s->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "solve() has not been called");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    int rc = harvestAll(s);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    if(total_ms)
    {
      *total_ms = s->last_total_ms;
    }
    if(kernel_ms)
    {
      *kernel_ms = s->last_kernel_ms;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_last_solve_phases(nmpc_hip_ddp_handle s, double * backward_ms, double * forward_ms, double * other_ms)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    if(!s->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "solve() has not been called");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    int rc = harvestAll(s); // (waits for the last solve)
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    std::vector<unsigned long long> host(static_cast<size_t>(s->B) * 4);
    NMPC_HIP_TRY(hipMemcpy(host.data(), s->d_phase_ticks, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // The waves that drive the instances run side by side for the whole launch: the split of the kernel's time is the split of
    // the longest-running wave's ticks (the one that sets the kernel's duration).
    size_t worst = 0;
    for(size_t b = 1; b < static_cast<size_t>(s->B); b++)
    {
      worst = host[b * 4 + 2] > host[worst * 4 + 2] ? b : worst;
    }
    const double total = static_cast<double>(host[worst * 4 + 2]);
    const double bw = total > 0 ? host[worst * 4 + 0] / total : 0.0, fw = total > 0 ? host[worst * 4 + 1] / total : 0.0;
    if(backward_ms)
    {
      *backward_ms = bw * s->last_kernel_ms;
    }
    if(forward_ms)
    {
      *forward_ms = fw * s->last_kernel_ms;
    }
    if(other_ms)
    {
      *other_ms = (1.0 - bw - fw) * s->last_kernel_ms;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_kernel_name(nmpc_hip_ddp_handle s, const char ** name)
  {
    if(!s || !name)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output pointer");
    }
    nmpc_amd::hip::ScopedKnobs knobs_guard(&s->knobs);
    *name = s->ops->kernel_name(s->B, s->cfg);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_kernel_name_for_batch(nmpc_hip_ddp_handle s, int batch, const char ** name)
  {
    if(!s || !name || batch < 1)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output pointer, or a non-positive batch");
    }
    nmpc_amd::hip::LaunchKnobs k = s->knobs;
    k.dispatch_batch = 0;
    nmpc_amd::hip::ScopedKnobs knobs_guard(&k);
    *name = s->ops->kernel_name(batch, s->cfg);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_kernel(nmpc_hip_ddp_handle s, const char * name)
  {
    if(!s || !name)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or kernel name");
    }
    static const char * const known[] = {"auto", "1w", "2w", "quad", "wpi", "tile64", "tile32"};
    // (also accepted: the names nmpc_hip_ddp_kernel_name reports)
    static const char * const reported[][2] = {{"ddp_solve_tpi_kernel", "1w"},        {"ddp_solve_tpi2w_kernel", "2w"},
                                               {"ddp_solve_quad_kernel", "quad"},     {"ddp_solve_wpi_kernel", "wpi"},
                                               {"ddp_solve_tile64_kernel", "tile64"}, {"ddp_solve_tile32_kernel", "tile32"}};
    const char * pick = nullptr;
    for(const char * k : known)
    {
      if(std::strcmp(k, name) == 0)
      {
        pick = k;
      }
    }
    for(const auto & r : reported)
    {
      if(std::strcmp(r[0], name) == 0)
      {
        pick = r[1];
      }
    }
    if(!pick)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, std::string("unknown kernel family: ") + name
                                                     + " (auto, 1w, 2w, quad, wpi, tile64, tile32)");
    }
    nmpc_amd::hip::LaunchKnobs k = s->knobs;
    std::memset(k.kernel, 0, sizeof(k.kernel));
    if(std::strcmp(pick, "auto") != 0)
    {
      std::strncpy(k.kernel, pick, sizeof(k.kernel) - 1);
    }
    s->knobs = k;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_set_dispatch_batch(nmpc_hip_ddp_handle s, int batch)
  {
    if(!s || batch < 0)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or a negative batch");
    }
    s->knobs.dispatch_batch = batch;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_last_solve_launches(nmpc_hip_ddp_handle s, int * launches)
  {
    if(!s || !launches)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle or output pointer");
    }
    if(!s->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "solve() has not been called");
    }
    *launches = s->last_ragged_rounds;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_ddp_timing_stats(nmpc_hip_ddp_handle s,
                                int reset,
                                long long * n_solves,
                                double * total_ms_sum,
                                double * kernel_ms_sum)
  {
    if(!s)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    NMPC_HIP_TRY(hipSetDevice(s->device));
    int rc = harvestAll(s);
    if(rc != NMPC_HIP_OK)
    {
      return rc;
    }
    if(n_solves)
    {
      *n_solves = s->n_harvested;
    }
    if(total_ms_sum)
    {
      *total_ms_sum = s->sum_total_ms;
    }
    if(kernel_ms_sum)
    {
      *kernel_ms_sum = s->sum_kernel_ms;
    }
    if(reset)
    {
      s->n_harvested = 0;
      s->sum_total_ms = 0;
      s->sum_kernel_ms = 0;
    }
    return NMPC_HIP_OK;
  }
} // extern "C"
