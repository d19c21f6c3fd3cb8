/* C-ABI of the MI355X-native batched FMPC solver (part of libnmpc_hip_ddp.so).
 *
 * Drop-in boundary for the reference's nmpc_fmpc::FmpcSolver (/root/reference/nmpc_fmpc/include/nmpc_fmpc/FmpcSolver.h:17-427):
 * a whole batch of independent `solve()` calls crosses it at once and the entire optimisation loop (FmpcSolver.hpp:233-246 ->
 * procOnce :356-491 -> backwardPass :522-665 / forwardPass :667-708 / updateVariables :710-838) runs on the GPU.  The reference
 * has no FFI layer of its own (header-only C++ templates); each entry point cites the member it replaces.  Plain pointers and
 * sizes only.  Every function returns 0 (NMPC_HIP_OK) or a negative nmpc_hip_status (nmpc_hip_ddp.h); nothing throws across
 * this boundary.  There is no CPU fallback: without a gfx950 device create() reports NMPC_HIP_ERR_NO_DEVICE.
 *
 * Layouts at this boundary (row-major in the order written, doubles unless noted; N / M / G = state / input / inequality
 * dimension of the problem type, T = horizon_steps, B = batch) — the memory image of the reference's per-solver
 * std::vector<Eigen vector> members, one block per instance:
 *   t          [B]              current_t                                              (FmpcSolver.h:283)
 *   x0         [B][N]           current_x                                              (FmpcSolver.h:283)
 *   X          [B][T+1][N]      Variable::x_list                                       (FmpcSolver.h:141)
 *   U          [B][T][M]        Variable::u_list                                       (FmpcSolver.h:144)
 *   LAMBDA     [B][T+1][N]      Variable::lambda_list                                  (FmpcSolver.h:147)
 *   S          [B][T][G]        Variable::s_list                                       (FmpcSolver.h:150)
 *   NU         [B][T][G]        Variable::nu_list                                      (FmpcSolver.h:153)
 *   STATUS     [B] int          FmpcSolver::Status returned by solve()                 (FmpcSolver.h:92-114)
 *   ITERS      [B] int          traceDataList().back().iter                            (FmpcSolver.h:233-236)
 *   TRACE      [B][max_iter][NMPC_HIP_FMPC_NTRACE]  traceDataList()                    (FmpcSolver.h:230-249)
 *   GAIN_K     [B][T][N][M]     coeffList()[i].K, per step the M x N gain column-major (FmpcSolver.h:217)
 *   GAIN_k     [B][T][M]        coeffList()[i].k                                       (FmpcSolver.h:214)
 *   GAIN_s     [B][T+1][N]      coeffList()[i].s                                       (FmpcSolver.h:220)
 *   GAIN_P     [B][T+1][N][N]   coeffList()[i].P (symmetric)                           (FmpcSolver.h:223)
 * On the device the same data is kept [step][element][instance]; the conversion happens inside set / get.
 */
#ifndef NMPC_HIP_FMPC_H
#define NMPC_HIP_FMPC_H

#include <stddef.h>

#include "nmpc_hip_ddp.h" /* nmpc_hip_status */

#ifdef __cplusplus
extern "C"
{
#endif

#define NMPC_HIP_FMPC_NTRACE 6
  /** Per-instance status beyond FmpcSolver::Status: checkVariable() found a negative s or nu (the reference throws
      std::runtime_error, FmpcSolver.hpp:338-353). */
#define NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE (-2)

  /** FmpcSolver::Status (FmpcSolver.h:92-114). */
  typedef enum
  {
    NMPC_HIP_FMPC_UNINITIALIZED = 0,
    NMPC_HIP_FMPC_SUCCEEDED = 1,
    NMPC_HIP_FMPC_ERROR_IN_FORWARD = 2,
    NMPC_HIP_FMPC_ERROR_IN_BACKWARD = 3,
    NMPC_HIP_FMPC_ERROR_IN_UPDATE = 4,
    NMPC_HIP_FMPC_MAX_ITERATION_REACHED = 5,
    NMPC_HIP_FMPC_ITERATION_CONTINUED = 6
  } nmpc_hip_fmpc_solve_status;

  /** FmpcSolver::Configuration (FmpcSolver.h:57-89) as a POD; print_level is host-side only and lives in the mirrors. */
  typedef struct
  {
    int horizon_steps; /* :63 (fixed at create(); set_config rejects a different value) */
    int max_iter; /* :66 */
    double kkt_error_thre; /* :69 */
    int check_nan; /* :72 */
    int init_complementary_variable; /* :75 */
    int update_barrier_eps; /* :78 */
    int break_if_llt_fails; /* :81 */
    int enable_line_search; /* :84 */
    int merit_const_scale_from_lagrange_multipliers; /* :87 */
    /** 1 (default): the kernel sequence of one solve is captured in a hipGraph at the first solve and replayed afterwards
        (one host launch per solve instead of ~6 per iteration); 0: plain stream launches. */
    int use_graph;
    /** 1: plain stream launches with a HIP-event pair around every kernel of the solve, summed per kernel class and read back
        with nmpc_hip_fmpc_last_solve_kernel_ms (bench.py's roofline leg).  Overrides use_graph.  0 (default): off. */
    int time_kernels;
  } nmpc_hip_fmpc_config;

  /** Trace columns.  TraceData (FmpcSolver.h:230-249) holds iter, kkt_error and four CPU timers; the timer slots carry the
      scalar decisions of the iteration instead (what the parity tests compare). */
  typedef enum
  {
    NMPC_HIP_FMPC_TRACE_ITER = 0,
    NMPC_HIP_FMPC_TRACE_KKT_ERROR = 1,
    NMPC_HIP_FMPC_TRACE_BARRIER_EPS = 2, /* barrier_eps_ of the iteration (FmpcSolver.hpp:370-392) */
    NMPC_HIP_FMPC_TRACE_ALPHA_S_MAX = 3, /* fraction-to-boundary step lengths (:713-742) */
    NMPC_HIP_FMPC_TRACE_ALPHA_NU_MAX = 4,
    NMPC_HIP_FMPC_TRACE_ALPHA_S = 5 /* after the merit line search (:748-792); = ALPHA_S_MAX when it is off */
  } nmpc_hip_fmpc_trace_col;

  typedef enum
  {
    NMPC_HIP_FMPC_FIELD_X = 0,
    NMPC_HIP_FMPC_FIELD_U = 1,
    NMPC_HIP_FMPC_FIELD_LAMBDA = 2,
    NMPC_HIP_FMPC_FIELD_S = 3,
    NMPC_HIP_FMPC_FIELD_NU = 4,
    NMPC_HIP_FMPC_FIELD_STATUS = 5, /* int */
    NMPC_HIP_FMPC_FIELD_ITERS = 6, /* int */
    NMPC_HIP_FMPC_FIELD_TRACE = 7,
    NMPC_HIP_FMPC_FIELD_GAIN_K = 8,
    NMPC_HIP_FMPC_FIELD_GAIN_k = 9,
    NMPC_HIP_FMPC_FIELD_GAIN_S = 10,
    NMPC_HIP_FMPC_FIELD_GAIN_P = 11,
    NMPC_HIP_FMPC_FIELD_BARRIER_EPS = 12, /* [B]: barrier_eps_ (FmpcSolver.h:414) */
    NMPC_HIP_FMPC_FIELD_DELTA_X = 13, /* delta_variable_ of the last iteration that reached the forward pass (FmpcSolver.h:402) */
    NMPC_HIP_FMPC_FIELD_DELTA_U = 14,
    NMPC_HIP_FMPC_FIELD_DELTA_LAMBDA = 15,
    NMPC_HIP_FMPC_FIELD_DELTA_S = 16,
    NMPC_HIP_FMPC_FIELD_DELTA_NU = 17,
    NMPC_HIP_FMPC_FIELD_MERIT = 18, /* [B][3]: merit_func_, merit_deriv_, merit_const_scale_ of the last line search (:417-423) */
    NMPC_HIP_FMPC_FIELD_PARTIALS = 19 /* diagnostic, [B][T+1][4]: per-timestep terms of the horizon reductions as the last kernels left
                                         them — KKT-error terms (calcKktError, :493-521), the two fraction-to-boundary candidates
                                         (:713-731), s . nu (:376-380) */
  } nmpc_hip_fmpc_field;

  typedef struct nmpc_hip_fmpc_solver * nmpc_hip_fmpc_handle;

  /** Fill cfg with the reference defaults (FmpcSolver.h:57-89). */
  int nmpc_hip_fmpc_default_config(nmpc_hip_fmpc_config * cfg);

  /** Registered FMPC problem types.  Replaces the template arguments FmpcSolver<StateDim, InputDim, IneqDim>
      (FmpcSolver.h:17-19). */
  int nmpc_hip_fmpc_model_count(void);
  int nmpc_hip_fmpc_model_name(int index, const char ** name);
  int nmpc_hip_fmpc_model_info(const char * model, int * state_dim, int * input_dim, int * ineq_dim, size_t * param_bytes);
  /** Copy the default-constructed problem object (a trivially-copyable blob of param_bytes) to out. */
  int nmpc_hip_fmpc_model_default_params(const char * model, void * out, size_t bytes);

  /** FmpcSolver::FmpcSolver(problem) (FmpcSolver.h:270) for `batch` instances with horizon `horizon_steps` on HIP device
      `device`.  All device buffers are allocated here and live until destroy; the variables and barrier_eps_ stay resident
      between solves (the warm start of the reference's callers, TestFmpcOscillator.cpp:193). */
  int nmpc_hip_fmpc_create(const char * model, int horizon_steps, int batch, int device, nmpc_hip_fmpc_handle * out);
  int nmpc_hip_fmpc_destroy(nmpc_hip_fmpc_handle h);

  /** FmpcSolver::config() (FmpcSolver.h:272-281). */
  int nmpc_hip_fmpc_set_config(nmpc_hip_fmpc_handle h, const nmpc_hip_fmpc_config * cfg);
  int nmpc_hip_fmpc_get_config(nmpc_hip_fmpc_handle h, nmpc_hip_fmpc_config * cfg);

  /** The problem object(s) the solver co-owns (FmpcSolver.h:393): one blob of param_bytes shared by every instance
      (per_instance = 0) or `batch` blobs back to back (per_instance = 1; dt() must be the same in all of them). */
  int nmpc_hip_fmpc_set_problem(nmpc_hip_fmpc_handle h, const void * params, size_t bytes, int per_instance);

  /** The `initial_variable` argument of solve (FmpcSolver.h:283) for every instance: HOST arrays in the layouts above, or
      DEVICE arrays of the same layouts when on_device != 0.  A NULL pointer leaves that part of the resident variable as
      it is.  barrier_eps [B] (may be NULL) sets barrier_eps_ (FmpcSolver.h:414; 1e-4 at create).  Host sources: synchronous.
      Device sources: asynchronous on the solver's own stream (ordered before the next solve on that stream). */
  int nmpc_hip_fmpc_set_variable(nmpc_hip_fmpc_handle h,
                                 const double * x,
                                 const double * u,
                                 const double * lambda,
                                 const double * s,
                                 const double * nu,
                                 const double * barrier_eps,
                                 int on_device);
  /** Variable::reset (FmpcSolver.hpp:42-69) on the resident variable of every instance. */
  int nmpc_hip_fmpc_reset_variable(nmpc_hip_fmpc_handle h, double x, double u, double lambda, double s, double nu);

  /** FmpcSolver::solve (FmpcSolver.h:283, FmpcSolver.hpp:156-255) for the whole batch from the resident variable (the
      `variable = solver.variable()` warm start of the callers is implicit).  HOST pointers: t [B] (NULL = all 0), x0 [B][N].
      Returns NMPC_HIP_ERR_RUNTIME if checkVariable() failed for some instance (those get status
      NMPC_HIP_FMPC_STATUS_INVALID_VARIABLE, the others are solved).  Synchronous. */
  int nmpc_hip_fmpc_solve(nmpc_hip_fmpc_handle h, const double * t, const double * x0);
  /** Same with DEVICE pointers (layouts above), asynchronous on `stream` (hipStream_t; NULL = the solver's own stream).
      checkVariable() failures are visible in STATUS only. */
  int nmpc_hip_fmpc_solve_device(nmpc_hip_fmpc_handle h, const double * d_t, const double * d_x0, void * stream);
  int nmpc_hip_fmpc_synchronize(nmpc_hip_fmpc_handle h);

  /** variable() / coeffList() / traceDataList() (FmpcSolver.h:286-301): copy one field to HOST memory (on_device = 0) or
      DEVICE memory in the boundary layout.  bytes must equal the field size. */
  int nmpc_hip_fmpc_get(nmpc_hip_fmpc_handle h, int field, void * out, size_t bytes, int on_device);
  int nmpc_hip_fmpc_field_bytes(nmpc_hip_fmpc_handle h, int field, size_t * bytes);

  /** computationDuration() (FmpcSolver.h:304-307): HIP-event time of the last solve [ms] (ingest of t / x0 + all kernels). */
  int nmpc_hip_fmpc_last_solve_ms(nmpc_hip_fmpc_handle h, float * ms);

  /** Kernel classes of nmpc_hip_fmpc_last_solve_kernel_ms, in this order. */
  typedef enum
  {
    NMPC_HIP_FMPC_KERNEL_BARRIER = 0,
    NMPC_HIP_FMPC_KERNEL_COEFF = 1,
    NMPC_HIP_FMPC_KERNEL_RICCATI = 2,
    NMPC_HIP_FMPC_KERNEL_DELTA = 3,
    NMPC_HIP_FMPC_KERNEL_STEP_LENGTH = 4,
    NMPC_HIP_FMPC_KERNEL_LINE_SEARCH = 5,
    NMPC_HIP_FMPC_KERNEL_UPDATE = 6, /* fmpc_update_kernel; or fmpc_tail_kernel, which does step length + update of an iteration and the
                                        barrier parameter + KKT-error terms of the next in one launch (fused-Riccati sequence without
                                        line search: classes 0, 1 and 4 then count the first iteration's launches only) */
    NMPC_HIP_FMPC_KERNEL_OTHER = 7, /* begin / init / check / finish */
    NMPC_HIP_FMPC_NKERNELS = 8
  } nmpc_hip_fmpc_kernel_class;

  /** The split of computationDuration() (FmpcSolver::ComputationDuration, FmpcSolver.h:252-287: coeff / backward / forward /
      update): HIP-event time [ms] and number of launches of each kernel class in the last solve.  Needs
      config.time_kernels = 1 (else NMPC_HIP_ERR_NOT_SOLVED).  ms / launches: arrays of NMPC_HIP_FMPC_NKERNELS. */
  int nmpc_hip_fmpc_last_solve_kernel_ms(nmpc_hip_fmpc_handle h, double * ms, int * launches);

  /** The reference's closed-loop caller patterns (TestFmpcOscillator.cpp:164-194, TestFmpcCartPole.cpp:340-366,408-414),
      batched and device-resident: n_ticks times { solve(t, x, resident variable); log; plant: `sim_substeps` steps of
      x <- stateEq(t, x, u, sim_dt), t += sim_dt } with u = u_list[0] (+ K_0 (x_list[0] - x) when use_feedback, the
      cart-pole test's inter-sample feedback, TestFmpcCartPole.cpp:347-351).  HOST pointers; logs may be NULL:
      x_log [B][n_ticks][N] (state handed to the solve of that tick), u0_log [B][n_ticks][M], status_log / iter_log
      [B][n_ticks] int, kkt_log [B][n_ticks] (kkt_error of the last iteration), x_final [B][N], t_final [B]. */
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
                            double * t_final);

  /** Names of the gfx950 kernels one iteration launches, comma separated (diagnostics for profiles / bench.py). */
  int nmpc_hip_fmpc_kernel_names(nmpc_hip_fmpc_handle h, const char ** names);

  /** Text of the last error raised on this thread. */
  const char * nmpc_hip_fmpc_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* NMPC_HIP_FMPC_H */
