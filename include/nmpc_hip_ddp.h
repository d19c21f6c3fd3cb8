/* C-ABI of the MI355X-native batched DDP solver (libnmpc_hip_ddp.so).
 *
 * This is the drop-in boundary for the hot path of the reference's nmpc_ddp::DDPSolver
 * (/root/reference/nmpc_ddp/include/nmpc_ddp/DDPSolver.h:255-308): a whole batch of independent
 * `solve()` calls crosses it at once and the entire optimisation loop (DDPSolver.hpp:115-123 -> procOnce
 * :143-340 -> backwardPass :342-534 / forwardPass :536-560, BoxQP.h:141-347) runs on the GPU.
 * The reference has no FFI layer of its own (it is a header-only C++ template library); each entry point
 * below cites the reference member it replaces.  Plain pointers and sizes only; no C++ or torch types.
 * Every function returns an int: 0 (NMPC_HIP_OK) or a negative nmpc_hip_status; nothing throws across
 * this boundary.  The host-side C++ mirror (include/nmpc_amd/DDPSolverBatch.hpp) turns the codes back into
 * the exception types the reference throws.
 *
 * Layouts (row-major in the order written; MM = max(input_dim_max, 1); doubles unless noted):
 *   x0      [B][N]                current_x of each instance                (DDPSolver.h:275)
 *   t0      [B]                   current_t of each instance (NULL = all 0) (DDPSolver.h:275)
 *   u_init  [B][T][MM]            initial_u_list, entries >= inputDim(t) ignored
 *   X       [B][T+1][N]           controlData().x_list                      (DDPSolver.h:116)
 *   U       [B][T][MM]            controlData().u_list                      (DDPSolver.h:119)
 *   COST    [B][T+1]              controlData().cost_list                   (DDPSolver.h:122)
 *   KFF     [B][T][MM]            k_list_                                   (DDPSolver.h:359)
 *   KFB     [B][T][N][MM]         K_list_; per step the m x N gain, column-major with leading dim MM,
 *                                 i.e. the memory image of Eigen's K_list_[t] when m == MM (DDPSolver.h:362)
 *   TRACE   [B][max_iter+1][NMPC_HIP_NTRACE]  traceDataList()               (DDPSolver.h:179-216,294)
 *   STATUS  [B] int               1 converged (solve() returned true), 0 max_iter exhausted, -1 failure
 *                                 (the retval of the last procOnce, DDPSolver.h:311-315, DDPSolver.hpp:140)
 *   ITERS   [B] int               traceDataList().back().iter
 */
#ifndef NMPC_HIP_DDP_H
#define NMPC_HIP_DDP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define NMPC_HIP_MAX_ALPHA 32
#define NMPC_HIP_NTRACE 12

  typedef enum
  {
    NMPC_HIP_OK = 0,
    NMPC_HIP_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference (DDPSolver.hpp:41-45) */
    NMPC_HIP_ERR_RUNTIME = -2, /* std::runtime_error in the reference (DDPSolver.hpp:46-58,391-414) */
    NMPC_HIP_ERR_UNKNOWN_MODEL = -3,
    NMPC_HIP_ERR_HIP = -4, /* a HIP runtime call failed; see nmpc_hip_ddp_last_error() */
    NMPC_HIP_ERR_NO_DEVICE = -5, /* no gfx950 device / HIP runtime unavailable: there is NO CPU fallback */
    NMPC_HIP_ERR_NOT_SOLVED = -6
  } nmpc_hip_status;

  /** DDPSolver::Configuration (DDPSolver.h:47-110) as a POD, plus BoxQP::Configuration (BoxQP.h:33-55).
      print_level is host-side only and lives in the C++/Python mirrors. */
  typedef struct
  {
    int with_input_constraint; /* DDPSolver.h:70 */
    int max_iter; /* :73 */
    int horizon_steps; /* :76 (fixed at create(); set_config rejects a different value) */
    int reg_type; /* :79 */
    double initial_lambda; /* :82 */
    double initial_dlambda; /* :85 */
    double lambda_factor; /* :88 */
    double lambda_min; /* :91 */
    double lambda_max; /* :94 */
    double k_rel_norm_thre; /* :97 */
    double lambda_thre; /* :100 */
    double cost_update_ratio_thre; /* :106 */
    double cost_update_thre; /* :109 */
    int n_alpha; /* alpha_list.size() (:103) */
    double alpha_list[NMPC_HIP_MAX_ALPHA];
    int use_state_eq_second_derivative; /* :67; non-zero is rejected with NMPC_HIP_ERR_RUNTIME (DDPSolver.hpp:391-414) */
    int qp_max_iter; /* BoxQP.h:39 */
    double qp_grad_thre; /* BoxQP.h:42 */
    double qp_rel_improve_thre; /* BoxQP.h:45 */
    double qp_step_factor; /* BoxQP.h:48 */
    double qp_min_step; /* BoxQP.h:51 */
    double qp_armijo_param; /* BoxQP.h:54 */
    int trace_level; /* 0: keep only the last trace row per instance; 1: full per-iteration trace */
    /* Line-search schedule of the kernels that can try several step sizes of alpha_list per forward pass (the quad kernel:
       cart-pole / bipedal up to 4096 instances).  Results are identical either way (the trials of DDPSolver.hpp:234-274 are
       independent and the first accepted one in list order is taken); only the time per iteration differs.
       0: automatic (parallel: the lane groups try the first four step sizes in the first pass, and the rollout of whichever
       is accepted is kept), 1: always parallel, 2: always sequential.  Box-constrained solves are always parallel. */
    int line_search_fan_out;
    /* Ragged-convergence schedule of a long solve (every DDPSolver object of the reference stops when IT converges,
       DDPSolver.hpp:115-123; a batch converges raggedly).  The solve is cut into resumable launches (iterations 1-16, 17-32, 33-48,
       49-64, 65-96, ... of those still running) with a device-side compaction between them, so that a persistent workgroup is not held
       by the one unconverged instance of its sixteen; no host round trip, results bit-identical to a single launch.
       0: automatic — on for solves QUEUED through nmpc_hip_ddp_solve_async with max_iter >= 64 (a caller that overlaps batches: the
       CUs a converged instance frees go to the batches behind it), off for the synchronous nmpc_hip_ddp_solve, for
       nmpc_hip_ddp_solve_device and for the ticks of nmpc_hip_ddp_mpc_run (max_iter is only a cap: a warm-started solve that
       converges within sixteen iterations must not pay the boundaries of the schedule, and a lone stream gains nothing from it);
       1: on wherever supported (the quad and two-wave kernels with a shared problem object; what a pool of handles over
       solve_device asks for), -1: off (one launch per solve). */
    int ragged_schedule;
  } nmpc_hip_ddp_config;

  /** Trace columns (TraceData, DDPSolver.h:179-216).  The three duration_* fields of the reference are
      per-instance CPU timers with no batched equivalent; their slots carry the discrete decisions the
      parity tests compare instead (alpha index, number of backward passes, number of forward passes). */
  typedef enum
  {
    NMPC_HIP_TRACE_ITER = 0,
    NMPC_HIP_TRACE_COST = 1,
    NMPC_HIP_TRACE_LAMBDA = 2,
    NMPC_HIP_TRACE_DLAMBDA = 3,
    NMPC_HIP_TRACE_ALPHA = 4,
    NMPC_HIP_TRACE_K_REL_NORM = 5,
    NMPC_HIP_TRACE_COST_UPDATE_ACTUAL = 6,
    NMPC_HIP_TRACE_COST_UPDATE_EXPECTED = 7,
    NMPC_HIP_TRACE_COST_UPDATE_RATIO = 8,
    NMPC_HIP_TRACE_ALPHA_IDX = 9,
    NMPC_HIP_TRACE_N_BACKWARD = 10,
    NMPC_HIP_TRACE_N_FORWARD = 11
  } nmpc_hip_trace_col;

  typedef enum
  {
    NMPC_HIP_FIELD_X = 0,
    NMPC_HIP_FIELD_U = 1,
    NMPC_HIP_FIELD_COST = 2,
    NMPC_HIP_FIELD_KFF = 3,
    NMPC_HIP_FIELD_KFB = 4,
    NMPC_HIP_FIELD_TRACE = 5,
    NMPC_HIP_FIELD_STATUS = 6, /* int */
    NMPC_HIP_FIELD_ITERS = 7, /* int */
    NMPC_HIP_FIELD_TRACE_LAST = 8, /* [B][NMPC_HIP_NTRACE]: last trace row of each instance */
    NMPC_HIP_FIELD_DV = 9, /* [B][2]: dV_ of the last backward pass (DDPSolver.h:374) */
    NMPC_HIP_FIELD_QP_RETVAL = 10, /* int [B][T]: BoxQP retval_ of the last backward pass (BoxQP.h:372) */
    NMPC_HIP_FIELD_QP_FREE_MASK = 11, /* unsigned [B][T]: bit i set <=> i in free_idxs_ (BoxQP.h:389) */
    NMPC_HIP_FIELD_INPUT_DIM = 12 /* int [B][T]: problem->inputDim(t0 + i dt) */
  } nmpc_hip_field;

  typedef struct nmpc_hip_ddp_solver * nmpc_hip_ddp_handle;

  /** Fill cfg with the reference defaults (DDPSolver::Configuration::Configuration, DDPSolver.h:50-60). */
  int nmpc_hip_ddp_default_config(nmpc_hip_ddp_config * cfg);

  /** Number of registered problem types, and their names / dimensions.
      Replaces the compile-time template arguments DDPSolver<StateDim, InputDim> (DDPSolver.h:23-25). */
  int nmpc_hip_ddp_model_count(void);
  int nmpc_hip_ddp_model_name(int index, const char ** name);
  int nmpc_hip_ddp_model_info(const char * model, int * state_dim, int * input_dim_max, int * dynamic_input,
                              size_t * param_bytes);
  /** Arithmetic type of a problem type: 8 = double (the reference's, DDPProblem.h:20-35), 4 = float (the "*_f32" problem
      types, BASELINE.json config 4).  The C-ABI exchanges doubles either way: inputs are rounded to the problem's type once
      at ingest, results widened at nmpc_hip_ddp_get.  fp32 problem types take the same calls as the double ones
      (with_input_constraint, set_model_params_batch, mpc_run) for n in {4, 8, 12}, m <= 4. */
  int nmpc_hip_ddp_model_scalar_bytes(const char * model, int * bytes);
  /** Copy the default-constructed problem object (a trivially-copyable blob of param_bytes) to out. */
  int nmpc_hip_ddp_model_default_params(const char * model, void * out, size_t bytes);

  /** DDPSolver::DDPSolver(problem) (DDPSolver.hpp:20-24) for a batch of `batch` instances with horizon
      `horizon_steps` on HIP device `device`.  All device buffers are allocated here and live until destroy,
      so consecutive solves (MPC warm start, DDPSolver.hpp:74-78) reuse them. */
  int nmpc_hip_ddp_create(const char * model, int horizon_steps, int batch, int device, nmpc_hip_ddp_handle * out);
  int nmpc_hip_ddp_destroy(nmpc_hip_ddp_handle h);

  /** DDPSolver::config() (DDPSolver.h:258-267). */
  int nmpc_hip_ddp_set_config(nmpc_hip_ddp_handle h, const nmpc_hip_ddp_config * cfg);
  int nmpc_hip_ddp_get_config(nmpc_hip_ddp_handle h, nmpc_hip_ddp_config * cfg);

  /** The problem object the solver co-owns (DDPSolver.h:332): overwrite it with a caller-built blob. */
  int nmpc_hip_ddp_set_model_params(nmpc_hip_ddp_handle h, const void * params, size_t bytes);

  /** Per-instance input limits (constant in time): lower[batch][MM], upper[batch][MM]; both NULL go back to the shared
      limits of nmpc_hip_ddp_set_input_limits (and to "no limits given" if that was never called).  Reference equivalent: a batch of DDPSolver objects, each with its own
      setInputLimitsFunc (DDPSolver.h:282-285). */
  int nmpc_hip_ddp_set_input_limits_batch(nmpc_hip_ddp_handle h, const double * lower, const double * upper);

  /** One problem object PER INSTANCE: params points to batch blobs of bytes_per_instance = param_bytes each (instance b at
      params + b * bytes_per_instance); NULL goes back to the shared object.  The reference equivalent is a batch of
      DDPSolver objects each constructed with its own problem (DDPSolver.hpp:20-24): different robots, weights or
      reference trajectories in one launch.  Restrictions: dt() and inputDim(t) must be the same for every instance (they
      define the shape of the batch); served by the model's default kernel (two-wavefront / wave-per-instance), a solve
      that would need the single-wavefront kernel reports NMPC_HIP_ERR_RUNTIME. */
  int nmpc_hip_ddp_set_model_params_batch(nmpc_hip_ddp_handle h, const void * params, size_t bytes_per_instance);

  /** problem->inputDim(t0 + i * dt) for i < horizon_steps, evaluated on the host from the handle's problem object:
      what DDPSolver::solve validates initial_u_list against (DDPSolver.hpp:46-58).  out has room for T ints. */
  int nmpc_hip_ddp_input_dims(nmpc_hip_ddp_handle h, double t0, int * out);

  /** DDPSolver::setInputLimitsFunc (DDPSolver.h:282-285) for limits that are constant in time, the form the reference's
      callers use (TestDDPCartPole.cpp:379-386, TestDDPVerticalMotion.cpp:262-270): lower[MM], upper[MM]; entries >=
      inputDim(t) are ignored.  Time-varying limits: nmpc_hip_ddp_set_input_limits_horizon. */
  int nmpc_hip_ddp_set_input_limits(nmpc_hip_ddp_handle h, const double * lower, const double * upper);

  /** DDPSolver::setInputLimitsFunc (DDPSolver.h:282-285) for limits that VARY in time: the reference evaluates
      input_limits_func_(current_t + i * dt) at every timestep of the backward pass (DDPSolver.hpp:470-472); here the caller
      samples the function for the next solve: lower / upper [T][MM] (per_instance = 0: every instance starts at the same
      current_t) or [batch][T][MM] (per_instance = 1).  Takes precedence over the constant limits; both NULL removes it.
      The table belongs to ONE solve's timesteps: nmpc_hip_ddp_mpc_run with more than one tick needs the longer table of
      nmpc_hip_ddp_set_input_limits_schedule. */
  int nmpc_hip_ddp_set_input_limits_horizon(nmpc_hip_ddp_handle h, const double * lower, const double * upper, int per_instance);

  /** The same for a run of solves that advance by one timestep each (nmpc_hip_ddp_mpc_run with the shift pattern, where the
      device advances current_t by dt per tick): `rows` >= T samples of the limits function, row j at current_t + j dt of the
      FIRST solve; tick k uses rows [k, k + T).  lower / upper [rows][MM] or [batch][rows][MM].  A run of n_ticks needs
      rows >= T + n_ticks - 1.  nmpc_hip_ddp_set_input_limits_horizon is this call with rows = T. */
  int nmpc_hip_ddp_set_input_limits_schedule(nmpc_hip_ddp_handle h, const double * lower, const double * upper, int rows, int per_instance);

  /** DDPSolver::solve (DDPSolver.h:275, DDPSolver.hpp:26-141) for the whole batch, HOST pointers:
      H2D copy, device solve, synchronise.  Results stay on the device until nmpc_hip_ddp_get. */
  int nmpc_hip_ddp_solve(nmpc_hip_ddp_handle h, const double * t0, const double * x0, const double * u_init);

  /** nmpc_hip_ddp_solve without the final synchronise: the inputs are staged (the host arrays may be reused when the call
      returns), the solve is queued on the handle's stream.  nmpc_hip_ddp_synchronize or nmpc_hip_ddp_get waits for it.  What a
      pool of handles (DDPSolverPool in include/nmpc_amd/DDPSolverBatch.hpp) overlaps consecutive batches with. */
  int nmpc_hip_ddp_solve_async(nmpc_hip_ddp_handle h, const double * t0, const double * x0, const double * u_init);

  /** Same with DEVICE pointers (reference layouts above, resident in HBM), asynchronous on `stream`
      (a hipStream_t, NULL = the solver's own stream).  Nothing is copied over PCIe. */
  int nmpc_hip_ddp_solve_device(nmpc_hip_ddp_handle h,
                                const double * d_t0,
                                const double * d_x0,
                                const double * d_u_init,
                                void * stream);
  int nmpc_hip_ddp_synchronize(nmpc_hip_ddp_handle h);

  /** controlData() / traceDataList() and friends (DDPSolver.h:288-297): copy one result field to HOST memory
      in the reference layout.  bytes must equal the field size. */
  int nmpc_hip_ddp_get(nmpc_hip_ddp_handle h, int field, void * out, size_t bytes);
  /** Same into DEVICE memory (e.g. the send buffer of the final RCCL gather), asynchronous on `stream`. */
  int nmpc_hip_ddp_get_device(nmpc_hip_ddp_handle h, int field, void * d_out, size_t bytes, void * stream);
  int nmpc_hip_ddp_field_bytes(nmpc_hip_ddp_handle h, int field, size_t * bytes);

  /** computationDuration() (DDPSolver.h:300-303): HIP-event time of the last device solve [ms]
      (ingest + solve kernel), and of the solve kernel alone. */
  int nmpc_hip_ddp_last_solve_ms(nmpc_hip_ddp_handle h, float * total_ms, float * kernel_ms);

  /** The split of computationDuration() (DDPSolver::ComputationDuration, DDPSolver.h:219-247: derivative / backward / forward):
      the solve kernel's HIP-event time of the last solve divided by the shader-clock shares of its phases, as counted by the
      wave that ran longest — backward passes (the linearisation is fused into them: derivative + backward + Q + reg + gain of
      the reference), forward passes (initial rollout + line-search rollouts), the rest (accept / lambda logic, write-out). */
  int nmpc_hip_ddp_last_solve_phases(nmpc_hip_ddp_handle h, double * backward_ms, double * forward_ms, double * other_ms);

  /** Accumulated HIP-event times over every device solve since create / the last reset: number of solves and
      the sums of (ingest + kernel) and of the solve kernel alone [ms].  Events are recorded on the launch stream
      and harvested lazily, so a sequence of asynchronous solves is timed without host synchronisation in
      between (bench.py's roofline leg). */
  int nmpc_hip_ddp_timing_stats(nmpc_hip_ddp_handle h,
                                int reset,
                                long long * n_solves,
                                double * total_ms_sum,
                                double * kernel_ms_sum);

  /** Options of the device-resident receding-horizon loop nmpc_hip_ddp_mpc_run. */
  typedef struct nmpc_hip_ddp_mpc_options
  {
    int n_ticks; /**< number of solve -> advance rounds */
    /** 1: "shift" caller pattern (TestDDPBipedal.cpp:243-268, TestDDPVerticalMotion.cpp:290-326,
        TestDDPCentroidalMotion.cpp:307-347): next x = x_list[1], u_list shifted by one with the last entry repeated
        (zeros when the input dimension changes at the end of the horizon), t += dt.
        0: "plant" caller pattern (TestDDPCartPole.cpp:323-346,388-403): u_list[0] (clamped to the input limits if
        clamp_u0) drives sim_substeps steps of the problem's stateEq(t, x, u, sim_dt), u_list is reused unshifted. */
    int shift_warm_start;
    /** > 0: config().max_iter for every solve after the first one (capped at the handle's config().max_iter, which
        sized the trace buffers) */
    int max_iter_after_first;
    int sim_substeps;
    double sim_dt;
    int clamp_u0;
  } nmpc_hip_ddp_mpc_options;

  /** shift pattern, one tick, no max_iter change. */
  int nmpc_hip_ddp_mpc_default_options(nmpc_hip_ddp_mpc_options * opt);

  /** The reference's receding-horizon caller loops, batched and device-resident: n_ticks times { solve; log; advance
      (t, x, u_list) on the device } without a host round trip in between.  Inputs as nmpc_hip_ddp_solve (HOST
      pointers).  Every log pointer is a HOST array or NULL:
        t_log[B][n_ticks], x_log[B][n_ticks][n] (state handed to the solve of that tick), u0_log[B][n_ticks][MM]
        (first input of the solution; clamped in the plant pattern), iter_log / status_log / m0_log[B][n_ticks]
        (iterations, status, input dimension of the first timestep), x_final[B][n], t_final[B] (after the last advance).
      Afterwards the handle holds the results of the LAST solve (nmpc_hip_ddp_get).  config().max_iter is restored.
      The plant pattern needs a problem type with the 4-argument stateEq (else NMPC_HIP_ERR_INVALID_ARGUMENT). */
  int nmpc_hip_ddp_mpc_run(nmpc_hip_ddp_handle h,
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
                           double * t_final);

  /** Name of the gfx950 kernel the next solve of this handle launches (as rocprofv3 --kernel-trace lists it, without
      template arguments): "ddp_solve_tpi2w_kernel" (master + helper wavefront per 64 instances, LDS-staged; chosen
      when the model's LDS records fit) or "ddp_solve_tpi_kernel" (one wavefront per 64 instances; also forced by the
      environment variable NMPC_HIP_DDP_KERNEL=1w).  No reference counterpart: diagnostics for profiles / bench.py. */
  int nmpc_hip_ddp_kernel_name(nmpc_hip_ddp_handle h, const char ** name);

  /** The same for a handle of `batch` instances with this handle's problem type, Configuration and kernel choice: what a sharding
      caller asks to learn which family the WHOLE batch would run on (nmpc_hip_ddp_set_dispatch_batch does that in one call). */
  int nmpc_hip_ddp_kernel_name_for_batch(nmpc_hip_ddp_handle h, int batch, const char ** name);

  /** Pin the kernel family of this handle: "auto" (default: chosen per solve from the problem's shape, the batch size and the
      Configuration), or one of "1w", "2w", "quad", "wpi", "tile64", "tile32" (or a name nmpc_hip_ddp_kernel_name reports).  A family
      the problem's shape has no instantiation of is ignored (the automatic choice among the remaining ones applies).  Within one
      family results are bit-reproducible across batch sizes and shardings; ACROSS families the discrete decisions agree and the
      values agree to ~1e-13 relative in fp64 (INTEGRATION.md "what is reproducible").  The environment variable NMPC_HIP_DDP_KERNEL is
      a developer override that is read once, when the handle is created, and sets the same field. */
  int nmpc_hip_ddp_set_kernel(nmpc_hip_ddp_handle h, const char * name);

  /** The batch size the kernel family is chosen FOR (0: the handle's own batch, the default).  A shard of a larger solve sets the
      size of the whole batch so that it runs on the family the unsharded solve would run on and returns its bits — in fp32 the
      family depends on the batch size (DDPSolverSharded.hpp and bench.py --global-batch do this). */
  int nmpc_hip_ddp_set_dispatch_batch(nmpc_hip_ddp_handle h, int batch);

  /** Kernel launches the LAST solve of this handle was cut into: 1 = one whole-solve launch; more = the ragged-convergence
      schedule (nmpc_hip_ddp_config::ragged_schedule: resumable launches with a device-side compaction between them).  No reference
      counterpart: diagnostics. */
  int nmpc_hip_ddp_last_solve_launches(nmpc_hip_ddp_handle h, int * launches);

  /** A QUEUE of n_instances problems (host arrays, reference layouts: t0 [N] or NULL, x0 [N][n], u_init [N][T][m]) through the
      handle's B slots, N >> B: every instance is solved to ITS convergence (or its max_iter-th iteration) — DDPSolver::solve of
      DDPSolver.hpp:26-141 once per instance, as a caller of the reference runs many problems through a few solver objects — and the
      slot of an instance that has finished takes the next one of the queue at the next round boundary (`span` iterations, 0: 8;
      include/nmpc_amd/hip/stream_schedule.hpp).  Every instance returns the bits of its lone solve on this kernel family.
      Kernel families with resumable launches only (n <= 4, one input, fp64, shared problem object and limits); blocks until the
      queue has drained.  Results: nmpc_hip_ddp_stream_get (X, U, COST, STATUS, ITERS, TRACE_LAST, DV; arrays of N instances). */
  int nmpc_hip_ddp_solve_stream(nmpc_hip_ddp_handle h, int n_instances, const double * t0, const double * x0, const double * u_init, int span);
  int nmpc_hip_ddp_stream_get(nmpc_hip_ddp_handle h, int field, void * out, size_t bytes);
  /** Rounds of the last streamed solve and its device time (HIP events, input staging excluded) [msec]. */
  int nmpc_hip_ddp_last_stream_stats(nmpc_hip_ddp_handle h, int * rounds, float * ms);

  /** Ask the HIP runtime for at least n hardware queues (GPU_MAX_HW_QUEUES; its default is 4).  Streams of one process are
      multiplexed onto them, and kernels of streams that share a queue run one after the other: a pool of handles overlaps only as
      many batches as there are queues.  The runtime reads the variable when it initialises, so this has to come before the first HIP
      call of the process (by anyone: torch, another library): *took_effect = 1 when the runtime is not up yet (the variable is raised
      to n if it was unset or smaller) or was started with a value >= n; 0 when it is already running with fewer — the caller can then
      say so instead of silently overlapping less.  The library never edits the environment on its own (it did, at load, until round 5).
      No reference counterpart (one solver, one thread there); DDPSolverPool's constructors call it with their handle count. */
  int nmpc_hip_ddp_request_hw_queues(int n, int * took_effect);

  /** Text of the last error raised on this thread (HIP error string or argument description). */
  const char * nmpc_hip_ddp_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* NMPC_HIP_DDP_H */
