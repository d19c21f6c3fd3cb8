// C-ABI of the batched FMPC solver (declared in include/nmpc_hip_fmpc.h): handles, device-buffer ownership, layout
// conversion at the boundary, the per-iteration kernel sequence and its hipGraph.  No CPU fallback exists: without the HIP
// runtime or a device every entry point that needs the GPU fails loudly.
#include <nmpc_hip_fmpc.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define NMPC_AMD_FMPC_COMMON_KERNELS // the problem-independent kernels of fmpc_kernels.hpp live in this translation unit
#include <nmpc_amd/hip/fmpc_ops.hpp>

using nmpc_amd::hip::FmpcBuffers;
using nmpc_amd::hip::FmpcOps;

namespace
{
thread_local std::string g_fmpc_last_error;

int fail(int code, const std::string & msg)
{
  g_fmpc_last_error = msg;
  return code;
}

#define FMPC_TRY(expr)                                                                    \
  do                                                                                      \
  {                                                                                       \
    hipError_t e_ = (expr);                                                               \
    if(e_ != hipSuccess)                                                                  \
    {                                                                                     \
      return fail(NMPC_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    }                                                                                     \
  } while(0)

#define FMPC_CHECK(expr)          \
  do                              \
  {                               \
    int rc_ = (expr);             \
    if(rc_ != NMPC_HIP_OK)        \
    {                             \
      return rc_;                 \
    }                             \
  } while(0)

std::vector<const FmpcOps *> & registry()
{
  static std::vector<const FmpcOps *> r;
  return r;
}

const FmpcOps * findModel(const char * name)
{
  if(!name)
  {
    return nullptr;
  }
  for(const FmpcOps * m : registry())
  {
    if(std::strcmp(m->name, name) == 0)
    {
      return m;
    }
  }
  return nullptr;
}

unsigned blocks(size_t threads, unsigned block)
{
  return static_cast<unsigned>((threads + block - 1) / block);
}

/** fmpc_transpose_kernel's grid: 32 x 32 tiles of the source matrix (B x steps E towards the device, steps E x B back). */
dim3 fmpcTransposeGrid(int B, int steps, int E, int to_device)
{
  const int R = steps * E;
  const int rows = to_device ? B : R, cols = to_device ? R : B;
  return dim3(static_cast<unsigned>((cols + 31) / 32), static_cast<unsigned>((rows + 31) / 32));
}

/** One row of the closed-loop log (nmpc_hip_fmpc_mpc_run): state handed to the solve, first input, status, iterations and
    the KKT error of the last iteration.  Logs are [tick][element][instance] on the device. */
__global__ void fmpc_log_kernel(FmpcBuffers buf, int tick, double * x_log, double * u0_log, int * status_log, int * iter_log, double * kkt_log)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if(b >= buf.B)
  {
    return;
  }
  for(int a = 0; a < buf.N; a++)
  {
    x_log[(static_cast<size_t>(tick) * buf.N + a) * buf.B + b] = buf.x0[static_cast<size_t>(a) * buf.B + b];
  }
  for(int a = 0; a < buf.M; a++)
  {
    u0_log[(static_cast<size_t>(tick) * buf.M + a) * buf.B + b] = buf.u[static_cast<size_t>(a) * buf.B + b];
  }
  const int it = buf.iters[b];
  status_log[static_cast<size_t>(tick) * buf.B + b] = buf.status[b];
  iter_log[static_cast<size_t>(tick) * buf.B + b] = it;
  kkt_log[static_cast<size_t>(tick) * buf.B + b] =
      it > 0 ? buf.trace[(static_cast<size_t>(b) * buf.max_iter + (it - 1)) * NMPC_HIP_FMPC_NTRACE + NMPC_HIP_FMPC_TRACE_KKT_ERROR] : 0.0;
}

__global__ void fmpc_fill_kernel(double * p, size_t n, double v)
{
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid < n)
  {
    p[tid] = v;
  }
}

/** Gather of one gain field out of the gain records into [steps][E][B]. */
__global__ void fmpc_gather_gain_kernel(const double * gain, double * dst, int B, int steps, int E, int stride, int offset)
{
  const size_t tid = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if(tid >= static_cast<size_t>(B) * steps * E)
  {
    return;
  }
  const int b = static_cast<int>(tid % B);
  const size_t ie = tid / B;
  const int i = static_cast<int>(ie / E), e = static_cast<int>(ie % E);
  dst[tid] = gain[(static_cast<size_t>(i) * stride + offset + e) * B + b];
}
} // namespace

struct nmpc_hip_fmpc_solver
{
  const FmpcOps * ops = nullptr;
  int device = 0;
  nmpc_hip_fmpc_config cfg;
  FmpcBuffers buf;
  std::vector<void *> allocs;
  double * d_t0 = nullptr; // [B]
  double * d_x0 = nullptr; // [N][B]
  void * d_problems = nullptr;
  size_t problems_bytes = 0;
  std::vector<unsigned char> host_problem; // first problem object (dt)
  double * d_stage = nullptr; // one field in the boundary layout
  size_t stage_bytes = 0;
  int trace_rows = 0; // max_iter the trace buffer was sized for
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  float last_ms = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool solved = false;
  int solved_max_iter = 0; // row stride of the trace the last solve wrote (set_config may change cfg.max_iter afterwards)
  double * solved_trace = nullptr; // ... and the buffer it wrote it to (a larger max_iter allocates a new one)
  std::string kernel_names;
  // config.time_kernels: one event pair per launch of the last solve, with its kernel class
  std::vector<hipEvent_t> kev;
  std::vector<int> kev_class;
  size_t kev_used = 0;
  bool kev_valid = false;
};

namespace
{
int devAlloc(nmpc_hip_fmpc_solver * h, double ** p, size_t count)
{
  FMPC_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(double)));
  h->allocs.push_back(*p);
  FMPC_TRY(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(double)));
  return NMPC_HIP_OK;
}

int devAllocInt(nmpc_hip_fmpc_solver * h, int ** p, size_t count)
{
  FMPC_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(int)));
  h->allocs.push_back(*p);
  FMPC_TRY(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(int)));
  return NMPC_HIP_OK;
}

int ensureStage(nmpc_hip_fmpc_solver * h, size_t bytes)
{
  if(h->stage_bytes >= bytes)
  {
    return NMPC_HIP_OK;
  }
  if(h->d_stage)
  {
    FMPC_TRY(hipFree(h->d_stage));
    h->d_stage = nullptr;
    h->stage_bytes = 0;
  }
  FMPC_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_stage), bytes));
  h->stage_bytes = bytes;
  return NMPC_HIP_OK;
}

void dropGraph(nmpc_hip_fmpc_solver * h)
{
  if(h->graph_exec)
  {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  if(h->graph)
  {
    (void)hipGraphDestroy(h->graph);
    h->graph = nullptr;
  }
}

void applyConfig(nmpc_hip_fmpc_solver * h)
{
  FmpcBuffers & b = h->buf;
  b.max_iter = h->cfg.max_iter;
  b.kkt_error_thre = h->cfg.kkt_error_thre;
  b.check_nan = h->cfg.check_nan;
  b.update_barrier_eps = h->cfg.update_barrier_eps;
  b.break_if_llt_fails = h->cfg.break_if_llt_fails;
  b.enable_line_search = h->cfg.enable_line_search;
  b.merit_const_scale_from_lagrange_multipliers = h->cfg.merit_const_scale_from_lagrange_multipliers;
}

/** config.time_kernels: an event on `stream` that opens (begin) or closes the timing bracket of one launch. */
int timeMark(nmpc_hip_fmpc_solver * h, hipStream_t stream, int kernel_class, bool begin)
{
  if(!h->cfg.time_kernels)
  {
    return NMPC_HIP_OK;
  }
  if(h->kev_used == h->kev.size())
  {
    hipEvent_t e = nullptr;
    FMPC_TRY(hipEventCreate(&e));
    h->kev.push_back(e);
    h->kev_class.push_back(0);
  }
  h->kev_class[h->kev_used] = begin ? kernel_class : -1;
  FMPC_TRY(hipEventRecord(h->kev[h->kev_used], stream));
  h->kev_used++;
  return NMPC_HIP_OK;
}

#define FMPC_TIMED(h, stream, kernel_class, launch)           \
  do                                                          \
  {                                                           \
    FMPC_CHECK(timeMark(h, stream, kernel_class, true));      \
    launch;                                                   \
    FMPC_TRY(hipGetLastError());                              \
    FMPC_CHECK(timeMark(h, stream, kernel_class, false));     \
  } while(0)

/** The kernel sequence of FmpcSolver::solve (FmpcSolver.hpp:156-255) on `stream`. */
int enqueueSolve(nmpc_hip_fmpc_solver * h, hipStream_t stream)
{
  const FmpcBuffers & buf = h->buf;
  const FmpcOps * ops = h->ops;
  const unsigned nb = blocks(buf.B, 64);
  h->kev_used = 0;
  h->kev_valid = h->cfg.time_kernels != 0;
  FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_OTHER,
             hipLaunchKernelGGL(nmpc_amd::hip::fmpc_begin_kernel, dim3(nb), dim3(64), 0, stream, buf));
  if(h->cfg.init_complementary_variable)
  {
    FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_OTHER, FMPC_TRY(ops->launch_init_complementary(buf, stream)));
  }
  FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_OTHER,
             hipLaunchKernelGGL(nmpc_amd::hip::fmpc_check_variable_kernel, dim3(blocks(static_cast<size_t>(buf.B) * buf.T, 256)),
                                dim3(256), 0, stream, buf));
  // With the fused Riccati kernel and no line search an iteration is three launches: fmpc_tail_kernel closes it (step length, update)
  // and opens the next one (barrier parameter, KKT-error terms, terminal record); the first iteration is opened by the two kernels below.
  const bool tail = ops->tail_applies(buf);
  for(int iter = 1; iter <= h->cfg.max_iter; iter++)
  {
    if(iter == 1 || !tail)
    {
      FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_BARRIER,
                 hipLaunchKernelGGL(nmpc_amd::hip::fmpc_barrier_kernel, dim3(nb), dim3(64 * nmpc_amd::hip::fmpc::kSlices), 0, stream, buf,
                                    iter));
      FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_COEFF, FMPC_TRY(ops->launch_coeff(buf, stream)));
    }
    FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_RICCATI, FMPC_TRY(ops->launch_riccati(buf, iter, stream)));
    FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_DELTA, FMPC_TRY(ops->launch_delta(buf, stream)));
    if(tail)
    {
      FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_UPDATE, FMPC_TRY(ops->launch_tail(buf, iter, iter == h->cfg.max_iter ? 1 : 0, stream)));
      continue;
    }
    FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_STEP_LENGTH,
               hipLaunchKernelGGL(nmpc_amd::hip::fmpc_step_length_kernel, dim3(nb), dim3(64 * nmpc_amd::hip::fmpc::kSlices), 0, stream,
                                  buf, iter));
    if(h->cfg.enable_line_search)
    {
      FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_LINE_SEARCH, FMPC_TRY(ops->launch_line_search(buf, iter, stream)));
    }
    FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_UPDATE,
               hipLaunchKernelGGL(nmpc_amd::hip::fmpc_update_kernel, dim3(blocks(static_cast<size_t>(buf.B) * (buf.T + 1), 256)),
                                  dim3(256), 0, stream, buf));
  }
  FMPC_TIMED(h, stream, NMPC_HIP_FMPC_KERNEL_OTHER,
             hipLaunchKernelGGL(nmpc_amd::hip::fmpc_finish_kernel, dim3(nb), dim3(64), 0, stream, buf));
  return NMPC_HIP_OK;
}

/** Launches the solve on `stream`: through the captured graph when use_graph is set. */
int launchSolve(nmpc_hip_fmpc_solver * h, hipStream_t stream)
{
  if(h->cfg.max_iter > h->trace_rows)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] max_iter exceeds the trace buffer; call set_config");
  }
  if(!h->cfg.use_graph || h->cfg.time_kernels)
  {
    return enqueueSolve(h, stream);
  }
  if(!h->graph_exec)
  {
    FMPC_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueueSolve(h, h->stream);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(h->stream, &g);
    if(rc != NMPC_HIP_OK)
    {
      if(g)
      {
        (void)hipGraphDestroy(g);
      }
      return rc;
    }
    FMPC_TRY(e);
    h->graph = g;
    FMPC_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
  }
  FMPC_TRY(hipGraphLaunch(h->graph_exec, stream));
  return NMPC_HIP_OK;
}

struct FieldInfo
{
  double * dev = nullptr; // device array in [steps][E][B] layout (nullptr: gain field / special)
  int steps = 0, E = 0;
  int gain_offset = -1;
  bool is_int = false;
};

int fieldInfo(nmpc_hip_fmpc_solver * h, int field, FieldInfo * fi)
{
  const FmpcBuffers & b = h->buf;
  const int T = b.T, N = b.N, M = b.M, G = b.G;
  switch(field)
  {
    case NMPC_HIP_FMPC_FIELD_X: *fi = {b.x, T + 1, N}; break;
    case NMPC_HIP_FMPC_FIELD_U: *fi = {b.u, T, M}; break;
    case NMPC_HIP_FMPC_FIELD_LAMBDA: *fi = {b.lam, T + 1, N}; break;
    case NMPC_HIP_FMPC_FIELD_S: *fi = {b.s, T, G}; break;
    case NMPC_HIP_FMPC_FIELD_NU: *fi = {b.nu, T, G}; break;
    case NMPC_HIP_FMPC_FIELD_DELTA_X: *fi = {b.dx, T + 1, N}; break;
    case NMPC_HIP_FMPC_FIELD_DELTA_U: *fi = {b.du, T, M}; break;
    case NMPC_HIP_FMPC_FIELD_DELTA_LAMBDA: *fi = {b.dlam, T + 1, N}; break;
    case NMPC_HIP_FMPC_FIELD_DELTA_S: *fi = {b.ds, T, G}; break;
    case NMPC_HIP_FMPC_FIELD_DELTA_NU: *fi = {b.dnu, T, G}; break;
    case NMPC_HIP_FMPC_FIELD_GAIN_K: *fi = {nullptr, T, N * M, h->ops->gain_offset_K}; break;
    case NMPC_HIP_FMPC_FIELD_GAIN_k: *fi = {nullptr, T, M, h->ops->gain_offset_k}; break;
    case NMPC_HIP_FMPC_FIELD_GAIN_S: *fi = {nullptr, T + 1, N, h->ops->gain_offset_s}; break;
    case NMPC_HIP_FMPC_FIELD_GAIN_P: *fi = {nullptr, T + 1, N * N, h->ops->gain_offset_P}; break;
    case NMPC_HIP_FMPC_FIELD_MERIT: *fi = {b.merit, 1, 3}; break;
    case NMPC_HIP_FMPC_FIELD_PARTIALS: *fi = {b.part, T + 1, nmpc_amd::hip::fmpc::kPartSlots}; break;
    case NMPC_HIP_FMPC_FIELD_BARRIER_EPS: *fi = {b.barrier_eps, 1, 1}; break;
    case NMPC_HIP_FMPC_FIELD_TRACE: // already [B][..]; rows and stride of the solve that wrote it
      *fi = {h->solved ? h->solved_trace : b.trace, h->solved ? h->solved_max_iter : h->cfg.max_iter, NMPC_HIP_FMPC_NTRACE};
      break;
    case NMPC_HIP_FMPC_FIELD_STATUS: *fi = {nullptr, 1, 1, -1, true}; break;
    case NMPC_HIP_FMPC_FIELD_ITERS: *fi = {nullptr, 1, 1, -1, true}; break;
    default: return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] unknown field");
  }
  return NMPC_HIP_OK;
}

/** Upload of one variable field given in the boundary layout. */
int uploadField(nmpc_hip_fmpc_solver * h, const double * src, double * dst, int steps, int E, int on_device)
{
  if(!src)
  {
    return NMPC_HIP_OK;
  }
  const size_t count = static_cast<size_t>(h->buf.B) * steps * E;
  if(count == 0)
  {
    return NMPC_HIP_OK;
  }
  const double * d_src = src;
  if(!on_device)
  {
    FMPC_CHECK(ensureStage(h, count * sizeof(double)));
    FMPC_TRY(hipMemcpyAsync(h->d_stage, src, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    d_src = h->d_stage;
  }
  hipLaunchKernelGGL(nmpc_amd::hip::fmpc_transpose_kernel, fmpcTransposeGrid(h->buf.B, steps, E, 1), dim3(256), 0, h->stream, d_src, dst,
                     h->buf.B, steps, E, 1);
  FMPC_TRY(hipGetLastError());
  if(!on_device)
  {
    FMPC_TRY(hipStreamSynchronize(h->stream)); // the staging buffer is reused by the next field
  }
  return NMPC_HIP_OK;
}

int ingest(nmpc_hip_fmpc_solver * h, const double * t, const double * x0, bool on_device, hipStream_t stream)
{
  const int B = h->buf.B, N = h->buf.N;
  if(!x0)
  {
    return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] x0 is NULL");
  }
  if(t)
  {
    FMPC_TRY(hipMemcpyAsync(h->d_t0, t, static_cast<size_t>(B) * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            stream));
  }
  else
  {
    FMPC_TRY(hipMemsetAsync(h->d_t0, 0, static_cast<size_t>(B) * sizeof(double), stream));
  }
  const double * d_src = x0;
  if(!on_device)
  {
    FMPC_CHECK(ensureStage(h, static_cast<size_t>(B) * N * sizeof(double)));
    FMPC_TRY(hipMemcpyAsync(h->d_stage, x0, static_cast<size_t>(B) * N * sizeof(double), hipMemcpyHostToDevice, stream));
    d_src = h->d_stage;
  }
  hipLaunchKernelGGL(nmpc_amd::hip::fmpc_transpose_kernel, fmpcTransposeGrid(B, 1, N, 1), dim3(256), 0, stream, d_src, h->d_x0, B, 1, N, 1);
  FMPC_TRY(hipGetLastError());
  return NMPC_HIP_OK;
}
} // namespace

extern "C"
{
  int nmpc_hip_fmpc_register_model(const FmpcOps * ops)
  {
    if(!ops || findModel(ops->name))
    {
      return NMPC_HIP_ERR_INVALID_ARGUMENT;
    }
    registry().push_back(ops);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_default_config(nmpc_hip_fmpc_config * cfg)
  {
    if(!cfg)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "cfg is NULL");
    }
    cfg->horizon_steps = 100;
    cfg->max_iter = 10;
    cfg->kkt_error_thre = 1e-4;
    cfg->check_nan = 1;
    cfg->init_complementary_variable = 0;
    cfg->update_barrier_eps = 1;
    cfg->break_if_llt_fails = 0;
    cfg->enable_line_search = 0;
    cfg->merit_const_scale_from_lagrange_multipliers = 0;
    cfg->use_graph = 1;
    cfg->time_kernels = 0;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_model_count(void)
  {
    return static_cast<int>(registry().size());
  }

  int nmpc_hip_fmpc_model_name(int index, const char ** name)
  {
    if(index < 0 || index >= static_cast<int>(registry().size()) || !name)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "model index out of range");
    }
    *name = registry()[index]->name;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_model_info(const char * model, int * state_dim, int * input_dim, int * ineq_dim, size_t * param_bytes)
  {
    const FmpcOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown FMPC problem type: ") + (model ? model : "(null)"));
    }
    if(state_dim)
    {
      *state_dim = m->state_dim;
    }
    if(input_dim)
    {
      *input_dim = m->input_dim;
    }
    if(ineq_dim)
    {
      *ineq_dim = m->ineq_dim;
    }
    if(param_bytes)
    {
      *param_bytes = m->param_bytes;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_model_default_params(const char * model, void * out, size_t bytes)
  {
    const FmpcOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown FMPC problem type: ") + (model ? model : "(null)"));
    }
    if(!out || bytes != m->param_bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "param blob size mismatch");
    }
    m->default_params(out);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_destroy(nmpc_hip_fmpc_handle h)
  {
    if(!h)
    {
      return NMPC_HIP_OK;
    }
    (void)hipSetDevice(h->device);
    if(h->stream)
    {
      (void)hipStreamSynchronize(h->stream);
    }
    dropGraph(h);
    for(void * p : h->allocs)
    {
      (void)hipFree(p);
    }
    for(hipEvent_t e : h->kev)
    {
      (void)hipEventDestroy(e);
    }
    if(h->d_problems)
    {
      (void)hipFree(h->d_problems);
    }
    if(h->d_stage)
    {
      (void)hipFree(h->d_stage);
    }
    if(h->ev0)
    {
      (void)hipEventDestroy(h->ev0);
    }
    if(h->ev1)
    {
      (void)hipEventDestroy(h->ev1);
    }
    if(h->stream)
    {
      (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_create(const char * model, int horizon_steps, int batch, int device, nmpc_hip_fmpc_handle * out)
  {
    if(!out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
    }
    *out = nullptr;
    const FmpcOps * m = findModel(model);
    if(!m)
    {
      return fail(NMPC_HIP_ERR_UNKNOWN_MODEL, std::string("unknown FMPC problem type: ") + (model ? model : "(null)"));
    }
    if(horizon_steps < 1 || batch < 1)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] horizon_steps and batch must be positive");
    }
    int n_dev = 0;
    const hipError_t e = hipGetDeviceCount(&n_dev);
    if(e != hipSuccess || n_dev <= 0)
    {
      return fail(NMPC_HIP_ERR_NO_DEVICE,
                  std::string("no HIP device available (") + hipGetErrorString(e) + "): the FMPC solver has no CPU fallback");
    }
    if(device < 0 || device >= n_dev)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "device index out of range");
    }
    FMPC_TRY(hipSetDevice(device));
    auto * h = new nmpc_hip_fmpc_solver();
    h->ops = m;
    h->device = device;
    nmpc_hip_fmpc_default_config(&h->cfg);
    h->cfg.horizon_steps = horizon_steps;
    FmpcBuffers & b = h->buf;
    b.B = batch;
    b.T = horizon_steps;
    b.N = m->state_dim;
    b.M = m->input_dim;
    b.G = m->ineq_dim;
    b.coef_stride = m->coef_stride;
    b.gain_stride = m->gain_stride;
    b.riccati_force = nmpc_amd::hip::fmpcRiccatiForceFromEnvironment(); // (developer override, read once per handle)
    {
      const char * tail_env = getenv("NMPC_HIP_FMPC_TAIL"); // (developer override, read once per handle: 0 = the separate kernels)
      b.fuse_tail = (tail_env && tail_env[0] == '0') ? 0 : 1;
    }
    const size_t B = batch, T = horizon_steps, N = b.N, M = b.M, G = b.G;
    int rc = NMPC_HIP_OK;
    auto A = [&](double ** p, size_t count) {
      if(rc == NMPC_HIP_OK)
      {
        rc = devAlloc(h, p, count);
      }
    };
    A(&b.x, (T + 1) * N * B);
    A(&b.u, T * M * B);
    A(&b.lam, (T + 1) * N * B);
    A(&b.s, T * G * B);
    A(&b.nu, T * G * B);
    A(&b.dx, (T + 1) * N * B);
    A(&b.du, T * M * B);
    A(&b.dlam, (T + 1) * N * B);
    A(&b.ds, T * G * B);
    A(&b.dnu, T * G * B);
    A(&b.coef, T * b.coef_stride * B);
    A(&b.gain, (T + 1) * b.gain_stride * B);
    A(&b.part, (T + 1) * nmpc_amd::hip::fmpc::kPartSlots * B);
    A(&h->d_t0, B);
    A(&h->d_x0, N * B);
    A(&b.barrier_eps, B);
    A(&b.alpha, 3 * B);
    A(&b.merit, 3 * B);
    h->trace_rows = h->cfg.max_iter;
    A(&b.trace, B * h->trace_rows * NMPC_HIP_FMPC_NTRACE);
    if(rc == NMPC_HIP_OK)
    {
      rc = devAllocInt(h, &b.status, B);
    }
    if(rc == NMPC_HIP_OK)
    {
      rc = devAllocInt(h, &b.iters, B);
    }
    if(rc == NMPC_HIP_OK)
    {
      rc = devAllocInt(h, &b.flags, B);
    }
    if(rc != NMPC_HIP_OK)
    {
      nmpc_hip_fmpc_destroy(h);
      return rc;
    }
    b.t0 = h->d_t0;
    b.x0 = h->d_x0;
    applyConfig(h);
    auto cleanup = [&](int code) {
      nmpc_hip_fmpc_destroy(h);
      return code;
    };
    if(hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
    {
      return cleanup(fail(NMPC_HIP_ERR_HIP, "stream / event creation failed"));
    }
    // default problem object, barrier_eps_ = 1e-4 (FmpcSolver.h:414)
    h->host_problem.resize(m->param_bytes);
    m->default_params(h->host_problem.data());
    if(hipMalloc(&h->d_problems, m->param_bytes) != hipSuccess
       || hipMemcpy(h->d_problems, h->host_problem.data(), m->param_bytes, hipMemcpyHostToDevice) != hipSuccess)
    {
      return cleanup(fail(NMPC_HIP_ERR_HIP, "problem upload failed"));
    }
    h->problems_bytes = m->param_bytes;
    b.problems = h->d_problems;
    b.own_problems = 0;
    // (the allocations above were cleared with hipMemset, which is ordered on the NULL stream, and the handle's stream is non-blocking:
    // the clears must have landed before anything on it runs)
    if(hipDeviceSynchronize() != hipSuccess)
    {
      return cleanup(fail(NMPC_HIP_ERR_HIP, "hipDeviceSynchronize failed"));
    }
    hipLaunchKernelGGL(fmpc_fill_kernel, dim3(blocks(B, 256)), dim3(256), 0, h->stream, b.barrier_eps, B, 1e-4);
    if(hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
    {
      return cleanup(fail(NMPC_HIP_ERR_HIP, "initialisation kernel failed (is this a gfx950 device?)"));
    }
    h->kernel_names = "fmpc_barrier_kernel,fmpc_coeff_kernel,fmpc_riccati_kernel,fmpc_delta_kernel,fmpc_step_length_kernel,"
                      "fmpc_update_kernel";
    *out = h;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_set_config(nmpc_hip_fmpc_handle h, const nmpc_hip_fmpc_config * cfg)
  {
    if(!h || !cfg)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    if(cfg->horizon_steps != h->buf.T)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] horizon_steps is fixed at create()");
    }
    if(cfg->max_iter < 0)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] max_iter must be non-negative");
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(hipStreamSynchronize(h->stream));
    if(cfg->max_iter > h->trace_rows)
    {
      double * p = nullptr;
      FMPC_CHECK(devAlloc(h, &p, static_cast<size_t>(h->buf.B) * cfg->max_iter * NMPC_HIP_FMPC_NTRACE));
      h->buf.trace = p; // the old buffer stays in allocs until destroy
      h->trace_rows = cfg->max_iter;
    }
    h->cfg = *cfg;
    applyConfig(h);
    dropGraph(h);
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_get_config(nmpc_hip_fmpc_handle h, nmpc_hip_fmpc_config * cfg)
  {
    if(!h || !cfg)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    *cfg = h->cfg;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_set_problem(nmpc_hip_fmpc_handle h, const void * params, size_t bytes, int per_instance)
  {
    if(!h || !params)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    const size_t pb = h->ops->param_bytes;
    const size_t need = per_instance ? pb * h->buf.B : pb;
    if(bytes != need)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] problem blob size mismatch");
    }
    if(per_instance)
    {
      const double dt0 = h->ops->dt(params);
      for(int b = 1; b < h->buf.B; b++)
      {
        if(h->ops->dt(static_cast<const unsigned char *>(params) + pb * b) != dt0)
        {
          return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] dt() must be the same for every instance");
        }
      }
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(hipStreamSynchronize(h->stream));
    if(need != h->problems_bytes)
    {
      FMPC_TRY(hipFree(h->d_problems));
      h->d_problems = nullptr;
      FMPC_TRY(hipMalloc(&h->d_problems, need));
      h->problems_bytes = need;
      dropGraph(h);
    }
    FMPC_TRY(hipMemcpy(h->d_problems, params, need, hipMemcpyHostToDevice));
    std::memcpy(h->host_problem.data(), params, pb);
    if(h->buf.own_problems != (per_instance ? 1 : 0))
    {
      dropGraph(h);
    }
    h->buf.problems = h->d_problems;
    h->buf.own_problems = per_instance ? 1 : 0;
    return NMPC_HIP_OK;
  }

  /** The last solve may have been queued on a caller's stream (solve_device): whatever reads or rewrites the handle's
      buffers on the handle's own stream waits for it first (ev1 is recorded behind every solve). */
  static hipError_t waitLastSolve(nmpc_hip_fmpc_handle h)
  {
    return (h->solved && h->ev1) ? hipStreamWaitEvent(h->stream, h->ev1, 0) : hipSuccess;
  }

  int nmpc_hip_fmpc_set_variable(nmpc_hip_fmpc_handle h,
                                 const double * x,
                                 const double * u,
                                 const double * lambda,
                                 const double * s,
                                 const double * nu,
                                 const double * barrier_eps,
                                 int on_device)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(waitLastSolve(h));
    const FmpcBuffers & b = h->buf;
    FMPC_CHECK(uploadField(h, x, b.x, b.T + 1, b.N, on_device));
    FMPC_CHECK(uploadField(h, u, b.u, b.T, b.M, on_device));
    FMPC_CHECK(uploadField(h, lambda, b.lam, b.T + 1, b.N, on_device));
    FMPC_CHECK(uploadField(h, s, b.s, b.T, b.G, on_device));
    FMPC_CHECK(uploadField(h, nu, b.nu, b.T, b.G, on_device));
    if(barrier_eps)
    {
      FMPC_TRY(hipMemcpyAsync(b.barrier_eps, barrier_eps, static_cast<size_t>(b.B) * sizeof(double),
                              on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    }
    if(!on_device) // device sources: asynchronous on the solver's stream, ordered before the next solve on it
    {
      FMPC_TRY(hipStreamSynchronize(h->stream));
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_reset_variable(nmpc_hip_fmpc_handle h, double x, double u, double lambda, double s, double nu)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    FMPC_TRY(hipSetDevice(h->device));
    const FmpcBuffers & b = h->buf;
    const size_t B = b.B, T = b.T;
    struct
    {
      double * p;
      size_t n;
      double v;
    } fills[] = {{b.x, (T + 1) * b.N * B, x}, {b.u, T * b.M * B, u}, {b.lam, (T + 1) * b.N * B, lambda}, {b.s, T * b.G * B, s},
                 {b.nu, T * b.G * B, nu}};
    for(const auto & f : fills)
    {
      if(f.n > 0)
      {
        hipLaunchKernelGGL(fmpc_fill_kernel, dim3(blocks(f.n, 256)), dim3(256), 0, h->stream, f.p, f.n, f.v);
        FMPC_TRY(hipGetLastError());
      }
    }
    FMPC_TRY(hipStreamSynchronize(h->stream));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_solve_device(nmpc_hip_fmpc_handle h, const double * d_t, const double * d_x0, void * stream)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    FMPC_TRY(hipSetDevice(h->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
    FMPC_TRY(hipEventRecord(h->ev0, st));
    FMPC_CHECK(ingest(h, d_t, d_x0, true, st));
    FMPC_CHECK(launchSolve(h, st));
    FMPC_TRY(hipEventRecord(h->ev1, st));
    h->timed = true;
    h->solved = true;
    h->solved_max_iter = h->buf.max_iter;
    h->solved_trace = h->buf.trace;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_synchronize(nmpc_hip_fmpc_handle h)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(hipDeviceSynchronize());
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_solve(nmpc_hip_fmpc_handle h, const double * t, const double * x0)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(hipEventRecord(h->ev0, h->stream));
    FMPC_CHECK(ingest(h, t, x0, false, h->stream));
    FMPC_CHECK(launchSolve(h, h->stream));
    FMPC_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    h->solved = true;
    h->solved_max_iter = h->buf.max_iter;
    h->solved_trace = h->buf.trace;
    std::vector<int> status(h->buf.B);
    FMPC_TRY(hipMemcpyAsync(status.data(), h->buf.status, status.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FMPC_TRY(hipStreamSynchronize(h->stream));
    for(int b = 0; b < h->buf.B; b++)
    {
      if(status[b] == NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE)
      {
        return fail(NMPC_HIP_ERR_RUNTIME, "[FMPC] s_list[i] / nu_list[i] must be non-negative. instance: " + std::to_string(b));
      }
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_field_bytes(nmpc_hip_fmpc_handle h, int field, size_t * bytes)
  {
    if(!h || !bytes)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    FieldInfo fi;
    FMPC_CHECK(fieldInfo(h, field, &fi));
    *bytes = static_cast<size_t>(h->buf.B) * fi.steps * fi.E * (fi.is_int ? sizeof(int) : sizeof(double));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_get(nmpc_hip_fmpc_handle h, int field, void * out, size_t bytes, int on_device)
  {
    if(!h || !out)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    FieldInfo fi;
    FMPC_CHECK(fieldInfo(h, field, &fi));
    const size_t count = static_cast<size_t>(h->buf.B) * fi.steps * fi.E;
    if(bytes != count * (fi.is_int ? sizeof(int) : sizeof(double)))
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] field size mismatch");
    }
    FMPC_TRY(hipSetDevice(h->device));
    FMPC_TRY(waitLastSolve(h));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if(count == 0)
    {
      return NMPC_HIP_OK;
    }
    if(fi.is_int)
    {
      const int * src = field == NMPC_HIP_FMPC_FIELD_STATUS ? h->buf.status : h->buf.iters;
      FMPC_TRY(hipMemcpyAsync(out, src, bytes, kind, h->stream));
      FMPC_TRY(hipStreamSynchronize(h->stream));
      return NMPC_HIP_OK;
    }
    if(field == NMPC_HIP_FMPC_FIELD_TRACE || (fi.steps == 1 && fi.E == 1 && fi.gain_offset < 0))
    {
      FMPC_TRY(hipMemcpyAsync(out, fi.dev, bytes, kind, h->stream)); // already instance-major
      FMPC_TRY(hipStreamSynchronize(h->stream));
      return NMPC_HIP_OK;
    }
    // [steps][E][B] -> [B][steps][E], through the staging buffer (twice its size for gain fields: gather, then transpose)
    const bool gain = fi.gain_offset >= 0;
    FMPC_CHECK(ensureStage(h, bytes * (gain ? 2 : 1) + (on_device ? 0 : 0)));
    double * d_t = h->d_stage;
    const double * src = fi.dev;
    if(gain)
    {
      double * d_g = h->d_stage + count;
      hipLaunchKernelGGL(fmpc_gather_gain_kernel, dim3(blocks(count, 256)), dim3(256), 0, h->stream, h->buf.gain, d_g, h->buf.B, fi.steps,
                         fi.E, h->buf.gain_stride, fi.gain_offset);
      FMPC_TRY(hipGetLastError());
      src = d_g;
    }
    double * dst = on_device ? static_cast<double *>(out) : d_t;
    hipLaunchKernelGGL(nmpc_amd::hip::fmpc_transpose_kernel, fmpcTransposeGrid(h->buf.B, fi.steps, fi.E, 0), dim3(256), 0, h->stream, src,
                       dst, h->buf.B, fi.steps, fi.E, 0);
    FMPC_TRY(hipGetLastError());
    if(!on_device)
    {
      FMPC_TRY(hipMemcpyAsync(out, d_t, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    FMPC_TRY(hipStreamSynchronize(h->stream));
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_last_solve_ms(nmpc_hip_fmpc_handle h, float * ms)
  {
    if(!h || !ms)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    if(!h->solved)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "[FMPC] no solve has been run on this handle");
    }
    if(h->timed)
    {
      FMPC_TRY(hipSetDevice(h->device));
      FMPC_TRY(hipEventSynchronize(h->ev1));
      FMPC_TRY(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
      h->timed = false;
    }
    *ms = h->last_ms;
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_last_solve_kernel_ms(nmpc_hip_fmpc_handle h, double * ms, int * launches)
  {
    if(!h || !ms || !launches)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    if(!h->solved || !h->kev_valid)
    {
      return fail(NMPC_HIP_ERR_NOT_SOLVED, "[FMPC] the last solve was not run with config.time_kernels = 1");
    }
    FMPC_TRY(hipSetDevice(h->device));
    for(int k = 0; k < NMPC_HIP_FMPC_NKERNELS; k++)
    {
      ms[k] = 0;
      launches[k] = 0;
    }
    if(h->kev_used > 0)
    {
      FMPC_TRY(hipEventSynchronize(h->kev[h->kev_used - 1]));
    }
    for(size_t k = 0; k + 1 < h->kev_used; k += 2)
    {
      float t = 0;
      FMPC_TRY(hipEventElapsedTime(&t, h->kev[k], h->kev[k + 1]));
      ms[h->kev_class[k]] += t;
      launches[h->kev_class[k]]++;
    }
    return NMPC_HIP_OK;
  }

  int nmpc_hip_fmpc_mpc_run(nmpc_hip_fmpc_handle h,
                            const double * t,
                            const double * x0,
                            int n_ticks,
                            double sim_dt,
                            int sim_substeps,
                            int use_feedback,
                            double * x_log,
                            double * u0_log,
                            int * status_log,
                            int * iter_log,
                            double * kkt_log,
                            double * x_final,
                            double * t_final)
  {
    if(!h)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL handle");
    }
    if(n_ticks < 1 || sim_substeps < 1 || !(sim_dt > 0))
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "[FMPC] n_ticks, sim_substeps and sim_dt must be positive");
    }
    FMPC_TRY(hipSetDevice(h->device));
    const int B = h->buf.B, N = h->buf.N, M = h->buf.M;
    const size_t nt = n_ticks;
    double *d_xlog = nullptr, *d_ulog = nullptr, *d_klog = nullptr;
    int *d_slog = nullptr, *d_ilog = nullptr;
    auto release = [&]() {
      (void)hipFree(d_xlog);
      (void)hipFree(d_ulog);
      (void)hipFree(d_klog);
      (void)hipFree(d_slog);
      (void)hipFree(d_ilog);
    };
    if(hipMalloc(reinterpret_cast<void **>(&d_xlog), nt * N * B * sizeof(double)) != hipSuccess
       || hipMalloc(reinterpret_cast<void **>(&d_ulog), std::max<size_t>(nt * M * B, 1) * sizeof(double)) != hipSuccess
       || hipMalloc(reinterpret_cast<void **>(&d_klog), nt * B * sizeof(double)) != hipSuccess
       || hipMalloc(reinterpret_cast<void **>(&d_slog), nt * B * sizeof(int)) != hipSuccess
       || hipMalloc(reinterpret_cast<void **>(&d_ilog), nt * B * sizeof(int)) != hipSuccess)
    {
      release();
      return fail(NMPC_HIP_ERR_HIP, "[FMPC] log buffers: hipMalloc failed");
    }
    int rc = ingest(h, t, x0, false, h->stream);
    (void)hipEventRecord(h->ev0, h->stream);
    for(int k = 0; k < n_ticks && rc == NMPC_HIP_OK; k++)
    {
      rc = launchSolve(h, h->stream);
      if(rc != NMPC_HIP_OK)
      {
        break;
      }
      hipLaunchKernelGGL(fmpc_log_kernel, dim3(blocks(B, 64)), dim3(64), 0, h->stream, h->buf, k, d_xlog, d_ulog, d_slog, d_ilog, d_klog);
      if(hipGetLastError() != hipSuccess || h->ops->launch_plant(h->buf, h->d_x0, h->d_t0, sim_dt, sim_substeps, use_feedback, h->stream) != hipSuccess)
      {
        rc = fail(NMPC_HIP_ERR_HIP, "[FMPC] closed-loop kernel launch failed");
      }
    }
    if(rc != NMPC_HIP_OK)
    {
      (void)hipStreamSynchronize(h->stream);
      release();
      return rc;
    }
    (void)hipEventRecord(h->ev1, h->stream);
    h->timed = true;
    h->solved = true;
    h->solved_max_iter = h->buf.max_iter;
    h->solved_trace = h->buf.trace;
    // logs: [tick][E][B] -> [B][tick][E]
    auto fetch = [&](const double * d_src, double * host, int steps, int E) -> int {
      if(!host || static_cast<size_t>(steps) * E == 0)
      {
        return NMPC_HIP_OK;
      }
      const size_t count = static_cast<size_t>(B) * steps * E;
      FMPC_CHECK(ensureStage(h, count * sizeof(double)));
      hipLaunchKernelGGL(nmpc_amd::hip::fmpc_transpose_kernel, fmpcTransposeGrid(B, steps, E, 0), dim3(256), 0, h->stream, d_src, h->d_stage,
                         B, steps, E, 0);
      FMPC_TRY(hipGetLastError());
      FMPC_TRY(hipMemcpyAsync(host, h->d_stage, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      FMPC_TRY(hipStreamSynchronize(h->stream));
      return NMPC_HIP_OK;
    };
    rc = fetch(d_xlog, x_log, n_ticks, N);
    if(rc == NMPC_HIP_OK)
    {
      rc = fetch(d_ulog, u0_log, n_ticks, M);
    }
    if(rc == NMPC_HIP_OK)
    {
      rc = fetch(d_klog, kkt_log, n_ticks, 1);
    }
    if(rc == NMPC_HIP_OK)
    {
      rc = fetch(h->d_x0, x_final, 1, N);
    }
    if(rc == NMPC_HIP_OK && t_final)
    {
      if(hipMemcpy(t_final, h->d_t0, static_cast<size_t>(B) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      {
        rc = fail(NMPC_HIP_ERR_HIP, "t_final copy failed");
      }
    }
    if(rc == NMPC_HIP_OK && (status_log || iter_log))
    {
      std::vector<int> tmp(nt * B);
      for(int which = 0; which < 2 && rc == NMPC_HIP_OK; which++)
      {
        int * host = which == 0 ? status_log : iter_log;
        if(!host)
        {
          continue;
        }
        if(hipMemcpy(tmp.data(), which == 0 ? d_slog : d_ilog, tmp.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        {
          rc = fail(NMPC_HIP_ERR_HIP, "log copy failed");
          break;
        }
        for(int b = 0; b < B; b++)
        {
          for(int k = 0; k < n_ticks; k++)
          {
            host[static_cast<size_t>(b) * n_ticks + k] = tmp[static_cast<size_t>(k) * B + b];
          }
        }
      }
    }
    (void)hipStreamSynchronize(h->stream);
    release();
    return rc;
  }

  int nmpc_hip_fmpc_kernel_names(nmpc_hip_fmpc_handle h, const char ** names)
  {
    if(!h || !names)
    {
      return fail(NMPC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    }
    (void)hipSetDevice(h->device);
    h->kernel_names = std::string("fmpc_barrier_kernel,fmpc_coeff_kernel,")
                      + (nmpc_amd::hip::fmpcUseQuadRiccati(h->buf.N, h->buf.M, h->buf.B, h->buf.riccati_force)
                             ? (nmpc_amd::hip::fmpcUseFusedRiccati(h->buf.B, h->buf.riccati_force) ? "fmpc_riccati_fused_kernel" : "fmpc_riccati_quad_kernel")
                             : "fmpc_riccati_kernel")
                      + ",fmpc_delta_kernel,"
                      + (h->ops->tail_applies(h->buf)
                             ? std::string("fmpc_tail_kernel")
                             : std::string("fmpc_step_length_kernel,") + (h->cfg.enable_line_search ? "fmpc_line_search_kernel," : "")
                                   + "fmpc_update_kernel");
    *names = h->kernel_names.c_str();
    return NMPC_HIP_OK;
  }

  const char * nmpc_hip_fmpc_last_error(void)
  {
    return g_fmpc_last_error.c_str();
  }
}
