// Host-side C++ mirror of the reference's nmpc_ddp::DDPSolver<StateDim, InputDim>
// (/root/reference/nmpc_ddp/include/nmpc_ddp/DDPSolver.h:23-376) for a BATCH of independent problem instances
// solved on one MI355X.  Same member names, argument meaning and error behaviour (the exception types of
// DDPSolver.hpp:41-58,391-414), with a leading batch index where the reference has one instance.
//
// This header is plain C++17 (no HIP, no Eigen): everything numeric happens behind the C-ABI of
// <nmpc_hip_ddp.h> in libnmpc_hip_ddp.so.  The problem TYPE must have been compiled into a gfx950 code object
// and registered (NMPC_AMD_REGISTER_PROBLEM, <nmpc_amd/hip/model_registry.hpp>); the problem OBJECT the user
// passes here (parameters, cost weights, dt) is copied to the solver at every solve(), so mutating
// `problem->param_` between solves behaves as with the reference's shared_ptr.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <nmpc_amd/DDPProblem.hpp>
#include <nmpc_hip_ddp.h>

namespace nmpc_amd
{
/** \brief Batched DDP solver.
    \tparam Problem problem class derived from nmpc_amd::DDPProblem<StateDim, InputDim[, MaxInputDim]> with a
            `static constexpr const char * kName` under which its kernels are registered */
template<class Problem>
class DDPSolverBatch
{
public:
  static constexpr int StateDim = Problem::kStateDim;
  static constexpr int InputDimMax = Problem::kInputDimMax;
  static constexpr int MM = InputDimMax > 0 ? InputDimMax : 1;

  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using InputStateDimMatrix = typename Problem::InputStateDimMatrix;

  /*! \brief Configuration (DDPSolver::Configuration, DDPSolver.h:47-110; defaults come from the library). */
  struct Configuration
  {
    Configuration()
    {
      nmpc_hip_ddp_config c;
      nmpc_hip_ddp_default_config(&c);
      with_input_constraint = c.with_input_constraint != 0;
      max_iter = c.max_iter;
      horizon_steps = c.horizon_steps;
      reg_type = c.reg_type;
      initial_lambda = c.initial_lambda;
      initial_dlambda = c.initial_dlambda;
      lambda_factor = c.lambda_factor;
      lambda_min = c.lambda_min;
      lambda_max = c.lambda_max;
      k_rel_norm_thre = c.k_rel_norm_thre;
      lambda_thre = c.lambda_thre;
      alpha_list.assign(c.alpha_list, c.alpha_list + c.n_alpha);
      cost_update_ratio_thre = c.cost_update_ratio_thre;
      cost_update_thre = c.cost_update_thre;
    }

    int print_level = 1;
    bool use_state_eq_second_derivative = false;
    bool with_input_constraint = false;
    int max_iter = 500;
    int horizon_steps = 100;
    int reg_type = 1;
    double initial_lambda = 1e-4;
    double initial_dlambda = 1.0;
    double lambda_factor = 1.6;
    double lambda_min = 1e-6;
    double lambda_max = 1e10;
    double k_rel_norm_thre = 1e-4;
    double lambda_thre = 1e-5;
    std::vector<double> alpha_list;
    double cost_update_ratio_thre = 0;
    double cost_update_thre = 1e-7;
    //! 1: keep the per-iteration trace of every instance on the device (traceDataList); 0: last row only
    int trace_level = 1;
    //! line-search schedule (nmpc_hip_ddp_config::line_search_fan_out): 0 automatic, 1 step-size parallel, 2 sequential
    int line_search_fan_out = 0;
    //! ragged-convergence schedule of long solves (nmpc_hip_ddp_config::ragged_schedule): 0 automatic, 1 on, -1 off
    int ragged_schedule = 0;
  };

  /*! \brief Control data of one instance (DDPSolver::ControlData, DDPSolver.h:113-123). */
  struct ControlData
  {
    std::vector<StateDimVector> x_list;
    std::vector<InputDimVector> u_list;
    std::vector<double> cost_list;
  };

  /*! \brief Data to trace optimization loop (DDPSolver::TraceData, DDPSolver.h:179-216).  The duration_*
      members have no per-instance meaning on the GPU and stay 0; the three integers are the discrete decisions
      of the iteration. */
  struct TraceData
  {
    int iter = 0;
    double cost = 0;
    double lambda = 0;
    double dlambda = 0;
    double alpha = 0;
    double k_rel_norm = 0;
    double cost_update_actual = 0;
    double cost_update_expected = 0;
    double cost_update_ratio = 0;
    double duration_derivative = 0;
    double duration_backward = 0;
    double duration_forward = 0;
    int alpha_idx = -1;
    int n_backward = 0;
    int n_forward = 0;
  };

  /*! \brief Data of computation duration (DDPSolver.h:219-247) for the whole batch [msec]. */
  struct ComputationDuration
  {
    double solve = 0; //!< H2D + layout conversion + solve kernel (HIP events)
    double setup = 0; //!< solve - opt
    double opt = 0; //!< solve kernel alone
    //! `opt` split by the shader-clock shares of the kernel's phases (nmpc_hip_ddp_last_solve_phases).  The linearisation is
    //! fused into the backward sweep: derivative / Q / reg / gain of DDPSolver.h:219-247 are parts of `backward` and stay 0.
    double derivative = 0, backward = 0, forward = 0, Q = 0, reg = 0, gain = 0;
  };

public:
  /** \brief Constructor.
      \param problem DDP problem
      \param batch_size number of independent instances per solve()
      \param device HIP device index */
  DDPSolverBatch(const std::shared_ptr<Problem> & problem, int batch_size, int device = 0)
  : problem_(problem), batch_size_(batch_size), device_(device)
  {
    static_assert(std::is_trivially_copyable<Problem>::value, "the problem object is passed to the GPU by value");
    if(batch_size <= 0)
    {
      throw std::invalid_argument("batch_size must be positive: " + std::to_string(batch_size));
    }
  }

  ~DDPSolverBatch()
  {
    nmpc_hip_ddp_destroy(handle_);
  }

  DDPSolverBatch(const DDPSolverBatch &) = delete;
  DDPSolverBatch & operator=(const DDPSolverBatch &) = delete;

  /** \brief Accessor to configuration. */
  inline Configuration & config()
  {
    return config_;
  }
  inline const Configuration & config() const
  {
    return config_;
  }

  inline int batchSize() const
  {
    return batch_size_;
  }

  /** \brief Set function to return input limits (lower, upper), DDPSolver.h:282-285.
      The reference evaluates it at every timestep of the backward pass, input_limits_func_(current_t + i * dt)
      (DDPSolver.hpp:470-472).  solve() samples it there — for every instance's own current_t — and hands the device
      the constant pair when the samples agree, the sampled table otherwise (nmpc_hip_ddp_set_input_limits_horizon).
      Each returned vector must have the problem's input dimension (the capacity, when it is dynamic). */
  inline void setInputLimitsFunc(const std::function<std::array<InputDimVector, 2>(double)> & input_limits_func)
  {
    input_limits_func_ = input_limits_func;
    has_limits_ = true;
    limits_from_func_ = true;
  }

  /** \brief Input limits that are constant in time, without a function object. */
  inline void setInputLimits(const InputDimVector & lower, const InputDimVector & upper)
  {
    limits_from_func_ = false;
    has_limits_ = true;
    storeConstantLimits(lower, upper);
    horizon_limits_active_ = false;
    horizon_limits_dirty_ = true;
  }

  /** \brief Per-instance input limits (constant in time): limits[b] = {lower, upper} of instance b — a batch of
      DDPSolver objects each with its own setInputLimitsFunc (DDPSolver.h:282-285).  An empty vector goes back to the
      shared limits. */
  inline void setInputLimitsBatch(const std::vector<std::array<InputDimVector, 2>> & limits)
  {
    if(!limits.empty() && static_cast<int>(limits.size()) != batch_size_)
    {
      throw std::invalid_argument("limits batch should be " + std::to_string(batch_size_) + " but "
                                  + std::to_string(limits.size()) + ".");
    }
    limits_batch_lo_.assign(limits.size() * MM, -INFINITY);
    limits_batch_up_.assign(limits.size() * MM, INFINITY);
    for(size_t b = 0; b < limits.size(); b++)
    {
      for(int a = 0; a < MM && a < limits[b][0].size() && a < limits[b][1].size(); a++)
      {
        limits_batch_lo_[b * MM + a] = limits[b][0][a];
        limits_batch_up_[b * MM + a] = limits[b][1][a];
      }
    }
    limits_batch_dirty_ = true;
  }

  /** \brief One problem object per instance — the batch then behaves like `batch` DDPSolver objects each constructed
      with its own problem (DDPSolver.hpp:20-24).  An empty vector goes back to the shared problem.  dt() and
      inputDim(t) must be those of the shared problem. */
  inline void setProblemBatch(const std::vector<Problem> & problems)
  {
    if(!problems.empty() && static_cast<int>(problems.size()) != batch_size_)
    {
      throw std::invalid_argument("problem batch should be " + std::to_string(batch_size_) + " but "
                                  + std::to_string(problems.size()) + ".");
    }
    problem_batch_ = problems;
    problem_batch_dirty_ = true;
  }

  /** \brief Solve optimization for every instance of the batch (DDPSolver::solve, DDPSolver.h:275).
      \param current_t current time of each instance [sec] (size batch)
      \param current_x current state of each instance (size batch)
      \param initial_u_list initial input sequence of each instance (size batch x horizon_steps)
      \return per instance, whether the process finished successfully (converged) */
  std::vector<bool> solve(const std::vector<double> & current_t,
                          const std::vector<StateDimVector> & current_x,
                          const std::vector<std::vector<InputDimVector>> & initial_u_list)
  {
    const size_t B = static_cast<size_t>(batch_size_);
    std::vector<double> x0, u0;
    sampleInputLimits(current_t);
    packInputs(current_t, current_x, initial_u_list, x0, u0);
    check(nmpc_hip_ddp_solve(handle_, current_t.data(), x0.data(), u0.data()));
    fetched_ = false;
    fetchResults();
    std::vector<bool> ok(B);
    int n_fail = 0;
    for(size_t b = 0; b < B; b++)
    {
      ok[b] = status_[b] == 1;
      n_fail += status_[b] < 0 ? 1 : 0;
    }
    if(n_fail > 0 && config_.print_level >= 1)
    {
      std::cout << "[DDP] Failure due to large lambda in " << n_fail << " of " << B << " instances." << std::endl;
    }
    return ok;
  }

  /** \brief Results of solveStream(): controlData() and the verdict of every instance of the queue. */
  struct StreamResult
  {
    std::vector<ControlData> control_data; //!< per instance (DDPSolver::controlData)
    std::vector<int> status; //!< 1 converged (solve() would return true), 0 max_iter exhausted, -1 failure due to large lambda
    std::vector<int> iters; //!< iterations of each instance (traceDataList().back().iter)
    int rounds = 0;
    double device_ms = 0;
  };

  /** \brief A QUEUE of instances through the solver's batch_size slots (nmpc_hip_ddp_solve_stream): any number of problems, each
      solved to ITS convergence as DDPSolver::solve does (DDPSolver.hpp:26-141, 115-123); the slot of an instance that has finished
      takes the next one of the queue after at most `span` (0: 8) further iterations.  Every instance returns the bits of its lone
      solve on the same kernel family.  State dimension <= 4, one input, fp64; shared problem object and constant limits. */
  StreamResult solveStream(const std::vector<double> & current_t,
                           const std::vector<StateDimVector> & current_x,
                           const std::vector<std::vector<InputDimVector>> & initial_u_list,
                           int span = 0)
  {
    const int T = config_.horizon_steps;
    const size_t N = current_x.size();
    if(current_t.size() != N || initial_u_list.size() != N || N == 0)
    {
      throw std::invalid_argument("current_t / current_x / initial_u_list should have one entry per instance of the queue.");
    }
    if(!current_t.empty())
    {
      sampleInputLimits(std::vector<double>(static_cast<size_t>(batch_size_), current_t[0]));
    }
    ensureHandle();
    pushState();
    std::vector<double> x0(N * StateDim), u0(N * T * MM, 0.0);
    for(size_t b = 0; b < N; b++)
    {
      if(static_cast<int>(initial_u_list[b].size()) != T)
      {
        throw std::invalid_argument("initial_u_list length should be " + std::to_string(T) + " but " + std::to_string(initial_u_list[b].size()) + ".");
      }
      for(int j = 0; j < StateDim; j++)
      {
        x0[b * StateDim + j] = current_x[b][j];
      }
      for(int i = 0; i < T; i++)
      {
        for(int a = 0; a < static_cast<int>(initial_u_list[b][i].size()) && a < MM; a++)
        {
          u0[(b * T + i) * MM + a] = initial_u_list[b][i][a];
        }
      }
    }
    check(nmpc_hip_ddp_solve_stream(handle_, static_cast<int>(N), current_t.data(), x0.data(), u0.data(), span));
    std::vector<double> X(N * (T + 1) * StateDim), U(N * T * MM), C(N * (T + 1));
    StreamResult r;
    r.status.resize(N);
    r.iters.resize(N);
    check(nmpc_hip_ddp_stream_get(handle_, NMPC_HIP_FIELD_X, X.data(), X.size() * sizeof(double)));
    check(nmpc_hip_ddp_stream_get(handle_, NMPC_HIP_FIELD_U, U.data(), U.size() * sizeof(double)));
    check(nmpc_hip_ddp_stream_get(handle_, NMPC_HIP_FIELD_COST, C.data(), C.size() * sizeof(double)));
    check(nmpc_hip_ddp_stream_get(handle_, NMPC_HIP_FIELD_STATUS, r.status.data(), N * sizeof(int)));
    check(nmpc_hip_ddp_stream_get(handle_, NMPC_HIP_FIELD_ITERS, r.iters.data(), N * sizeof(int)));
    float ms = 0;
    check(nmpc_hip_ddp_last_stream_stats(handle_, &r.rounds, &ms));
    r.device_ms = ms;
    r.control_data.resize(N);
    for(size_t b = 0; b < N; b++)
    {
      ControlData & cd = r.control_data[b];
      cd.x_list.resize(static_cast<size_t>(T + 1));
      cd.u_list.resize(static_cast<size_t>(T));
      cd.cost_list.resize(T + 1);
      for(int i = 0; i <= T; i++)
      {
        for(int j = 0; j < StateDim; j++)
        {
          cd.x_list[static_cast<size_t>(i)][j] = X[(b * (T + 1) + i) * StateDim + j];
        }
        cd.cost_list[i] = C[b * (T + 1) + i];
      }
      for(int i = 0; i < T; i++)
      {
        cd.u_list[static_cast<size_t>(i)].resize(MM);
        for(int a = 0; a < MM; a++)
        {
          cd.u_list[static_cast<size_t>(i)][a] = U[(b * T + i) * MM + a];
        }
      }
    }
    return r;
  }

  /** \brief solve() without waiting for the device: the inputs are validated and staged (the arguments may be reused when the call
      returns), the solve is queued on the handle's stream; wait() blocks until it is done and fetches the results, after which the
      accessors (controlData(b), traceDataList(b), ...) hold them.  What DDPSolverPool overlaps consecutive batches with.
      A solve still in flight on this solver is waited for first and its results are fetched: the accessors hold THEM until the
      new solve is waited for (queueing over an unfetched solve would overwrite its results on the device unread). */
  void solveAsync(const std::vector<double> & current_t,
                  const std::vector<StateDimVector> & current_x,
                  const std::vector<std::vector<InputDimVector>> & initial_u_list)
  {
    if(in_flight_)
    {
      wait();
    }
    std::vector<double> x0, u0;
    sampleInputLimits(current_t);
    packInputs(current_t, current_x, initial_u_list, x0, u0);
    check(nmpc_hip_ddp_solve_async(handle_, current_t.data(), x0.data(), u0.data()));
    fetched_ = false;
    in_flight_ = true;
  }

  /** \brief Wait for the solve queued by solveAsync() and fetch its results.
      \return per instance, whether the process finished successfully (converged) — what solve() returns */
  std::vector<bool> wait()
  {
    if(!in_flight_ && !fetched_)
    {
      throw std::runtime_error("wait(): no solve has been queued");
    }
    if(!fetched_)
    {
      fetchResults(); // (nmpc_hip_ddp_get synchronises with the handle's stream)
    }
    in_flight_ = false;
    std::vector<bool> ok(static_cast<size_t>(batch_size_));
    for(size_t b = 0; b < ok.size(); b++)
    {
      ok[b] = status_[b] == 1;
    }
    return ok;
  }

  /** \brief Pin the kernel family of this solver: "auto" (the default), "1w", "2w", "quad", "wpi", "tile64", "tile32"
      (nmpc_hip_ddp_set_kernel).  Results are bit-reproducible across batch sizes and shardings within one family. */
  void setKernel(const std::string & name)
  {
    kernel_ = name;
    if(handle_)
    {
      check(nmpc_hip_ddp_set_kernel(handle_, kernel_.c_str()));
    }
  }

  /** \brief The batch size the kernel family is chosen for: a solver that holds a shard of a larger batch sets the whole batch's
      size and returns the unsharded solve's bits (nmpc_hip_ddp_set_dispatch_batch); 0: this solver's own batch size. */
  void setDispatchBatch(int batch)
  {
    dispatch_batch_ = batch;
    if(handle_)
    {
      check(nmpc_hip_ddp_set_dispatch_batch(handle_, dispatch_batch_));
    }
  }

  /** \brief Name of the gfx950 kernel the next solve launches (nmpc_hip_ddp_kernel_name). */
  std::string kernelName()
  {
    ensureHandle();
    pushState();
    const char * name = nullptr;
    check(nmpc_hip_ddp_kernel_name(handle_, &name));
    return name ? name : "";
  }

  /** \brief Whether a solve queued by solveAsync() has not been waited for yet. */
  inline bool inFlight() const
  {
    return in_flight_;
  }

  int lastSolveLaunches() const
  {
    int n = 0;
    check(nmpc_hip_ddp_last_solve_launches(handle_, &n));
    return n;
  }

  /*! \brief Per-tick log of mpcRun(): the columns the reference's closed-loop tests dump (TestDDPBipedal.cpp:251-262). */
  struct MpcLog
  {
    int n_ticks = 0;
    std::vector<double> t; //!< [batch][n_ticks]
    std::vector<double> x; //!< [batch][n_ticks][StateDim] state handed to the solve of the tick
    std::vector<double> u0; //!< [batch][n_ticks][MM] first input of the solution (clamped in the plant pattern)
    std::vector<int> iters, status, m0; //!< [batch][n_ticks]
    std::vector<double> x_final; //!< [batch][StateDim] state after the last advance
    std::vector<double> t_final; //!< [batch]
  };

  /** \brief The reference's receding-horizon caller loops, batched and device-resident (nmpc_hip_ddp_mpc_run):
      n_ticks times { solve; advance (t, x, u_list) on the device }.
      \param shift_warm_start true: the loop of TestDDPBipedal.cpp:243-268 (next x = x_list[1], u_list shifted);
                               false: the plant loop of TestDDPCartPole.cpp:323-346,388-403 (u_list[0], clamped to the
                               input limits if clamp_u0, drives sim_substeps steps of stateEq(t, x, u, sim_dt))
      \param max_iter_after_first if > 0, config().max_iter of every solve after the first one
      Afterwards controlData(b) etc. hold the results of the last solve. */
  MpcLog mpcRun(const std::vector<double> & current_t,
                const std::vector<StateDimVector> & current_x,
                const std::vector<std::vector<InputDimVector>> & initial_u_list,
                int n_ticks,
                bool shift_warm_start = true,
                int max_iter_after_first = 0,
                int sim_substeps = 0,
                double sim_dt = 0.0,
                bool clamp_u0 = true)
  {
    const size_t B = static_cast<size_t>(batch_size_);
    std::vector<double> x0, u0;
    sampleInputLimits(current_t, shift_warm_start ? n_ticks - 1 : 0);
    packInputs(current_t, current_x, initial_u_list, x0, u0);
    nmpc_hip_ddp_mpc_options opt;
    check(nmpc_hip_ddp_mpc_default_options(&opt));
    opt.n_ticks = n_ticks;
    opt.shift_warm_start = shift_warm_start ? 1 : 0;
    opt.max_iter_after_first = max_iter_after_first;
    opt.sim_substeps = sim_substeps;
    opt.sim_dt = sim_dt;
    opt.clamp_u0 = clamp_u0 ? 1 : 0;
    MpcLog log;
    log.n_ticks = n_ticks;
    const size_t nt = n_ticks > 0 ? static_cast<size_t>(n_ticks) : 0;
    log.t.resize(B * nt);
    log.x.resize(B * nt * StateDim);
    log.u0.resize(B * nt * MM);
    log.iters.resize(B * nt);
    log.status.resize(B * nt);
    log.m0.resize(B * nt);
    log.x_final.resize(B * StateDim);
    log.t_final.resize(B);
    check(nmpc_hip_ddp_mpc_run(handle_, current_t.data(), x0.data(), u0.data(), &opt, log.t.data(), log.x.data(),
                               log.u0.data(), log.iters.data(), log.status.data(), log.m0.data(), log.x_final.data(),
                               log.t_final.data()));
    fetched_ = false;
    fetchResults();
    return log;
  }

  /** \brief Const accessor to control data of instance b calculated by solve(). */
  inline const ControlData & controlData(int b) const
  {
    return control_data_.at(b);
  }

  /** \brief Const accessor to trace data list of instance b. */
  inline const std::vector<TraceData> & traceDataList(int b) const
  {
    return trace_data_list_.at(b);
  }

  /** \brief Feedforward terms k[0..N-1] and feedback gains K[0..N-1] of instance b (k_list_, K_list_). */
  inline const std::vector<InputDimVector> & kList(int b) const
  {
    return k_list_.at(b);
  }
  inline const std::vector<InputStateDimMatrix> & KList(int b) const
  {
    return K_list_.at(b);
  }

  /** \brief Status of instance b: 1 converged, 0 max_iter exhausted, -1 failure (procOnce retval, DDPSolver.h:311-315). */
  inline int status(int b) const
  {
    return status_.at(b);
  }

  /** \brief Const accessor to computation duration. */
  inline const ComputationDuration & computationDuration() const
  {
    return computation_duration_;
  }

  /** \brief Dump trace data list of instance b (same 12 columns as DDPSolver::dumpTraceDataList). */
  void dumpTraceDataList(int b, const std::string & file_path) const
  {
    std::ofstream ofs(file_path);
    ofs << "iter cost lambda dlambda alpha k_rel_norm cost_update_actual cost_update_expected cost_update_ratio "
           "duration_derivative duration_backward duration_forward"
        << std::endl;
    for(const auto & t : trace_data_list_.at(b))
    {
      ofs << t.iter << " " << t.cost << " " << t.lambda << " " << t.dlambda << " " << t.alpha << " " << t.k_rel_norm
          << " " << t.cost_update_actual << " " << t.cost_update_expected << " " << t.cost_update_ratio << " "
          << t.duration_derivative << " " << t.duration_backward << " " << t.duration_forward << std::endl;
    }
  }

protected:
  static void check(int rc)
  {
    if(rc == NMPC_HIP_OK)
    {
      return;
    }
    const std::string msg = nmpc_hip_ddp_last_error();
    if(rc == NMPC_HIP_ERR_INVALID_ARGUMENT || rc == NMPC_HIP_ERR_UNKNOWN_MODEL)
    {
      throw std::invalid_argument(msg);
    }
    throw std::runtime_error(msg);
  }

  void ensureHandle()
  {
    if(handle_ && handle_T_ == config_.horizon_steps)
    {
      return;
    }
    nmpc_hip_ddp_destroy(handle_);
    handle_ = nullptr;
    check(nmpc_hip_ddp_create(Problem::kName, config_.horizon_steps, batch_size_, device_, &handle_));
    handle_T_ = config_.horizon_steps;
    if(!kernel_.empty())
    {
      check(nmpc_hip_ddp_set_kernel(handle_, kernel_.c_str()));
    }
    if(dispatch_batch_ > 0)
    {
      check(nmpc_hip_ddp_set_dispatch_batch(handle_, dispatch_batch_));
    }
    problem_batch_dirty_ = !problem_batch_.empty(); // a new handle starts with the shared problem
    limits_batch_dirty_ = !limits_batch_lo_.empty();
  }

  void pushState()
  {
    check(nmpc_hip_ddp_set_model_params(handle_, problem_.get(), sizeof(Problem)));
    if(problem_batch_dirty_)
    {
      check(nmpc_hip_ddp_set_model_params_batch(handle_, problem_batch_.empty() ? nullptr : problem_batch_.data(),
                                                sizeof(Problem)));
      problem_batch_dirty_ = false;
    }
    if(limits_batch_dirty_)
    {
      const bool none = limits_batch_lo_.empty();
      check(nmpc_hip_ddp_set_input_limits_batch(handle_, none ? nullptr : limits_batch_lo_.data(),
                                                none ? nullptr : limits_batch_up_.data()));
      limits_batch_dirty_ = false;
    }
    nmpc_hip_ddp_config c;
    nmpc_hip_ddp_default_config(&c);
    c.with_input_constraint = config_.with_input_constraint ? 1 : 0;
    c.use_state_eq_second_derivative = config_.use_state_eq_second_derivative ? 1 : 0;
    c.max_iter = config_.max_iter;
    c.horizon_steps = config_.horizon_steps;
    c.reg_type = config_.reg_type;
    c.initial_lambda = config_.initial_lambda;
    c.initial_dlambda = config_.initial_dlambda;
    c.lambda_factor = config_.lambda_factor;
    c.lambda_min = config_.lambda_min;
    c.lambda_max = config_.lambda_max;
    c.k_rel_norm_thre = config_.k_rel_norm_thre;
    c.lambda_thre = config_.lambda_thre;
    c.cost_update_ratio_thre = config_.cost_update_ratio_thre;
    c.cost_update_thre = config_.cost_update_thre;
    c.trace_level = config_.trace_level;
    c.line_search_fan_out = config_.line_search_fan_out;
    c.ragged_schedule = config_.ragged_schedule;
    if(config_.alpha_list.empty() || config_.alpha_list.size() > NMPC_HIP_MAX_ALPHA)
    {
      throw std::invalid_argument("alpha_list size must be in [1, 32]");
    }
    c.n_alpha = static_cast<int>(config_.alpha_list.size());
    for(int i = 0; i < c.n_alpha; i++)
    {
      c.alpha_list[i] = config_.alpha_list[i];
    }
    check(nmpc_hip_ddp_set_config(handle_, &c));
    if(has_limits_ && !horizon_limits_active_)
    {
      check(nmpc_hip_ddp_set_input_limits(handle_, lower_, upper_));
    }
    if(horizon_limits_dirty_)
    {
      if(horizon_limits_active_)
      {
        check(nmpc_hip_ddp_set_input_limits_schedule(handle_, horizon_lo_.data(), horizon_up_.data(), horizon_rows_,
                                                     horizon_per_instance_ ? 1 : 0));
      }
      else
      {
        check(nmpc_hip_ddp_set_input_limits_horizon(handle_, nullptr, nullptr, 0));
      }
      horizon_limits_dirty_ = false;
    }
  }

  inline void storeConstantLimits(const InputDimVector & lower, const InputDimVector & upper)
  {
    if(lower.size() != upper.size() || lower.size() > MM || (lower.size() == 0 && MM > 0 && Problem::kInputDimMax > 0))
    {
      throw std::invalid_argument("input limits should have the input dimension " + std::to_string(Problem::kInputDimMax)
                                  + " but " + std::to_string(lower.size()) + ".");
    }
    for(int i = 0; i < MM; i++)
    {
      lower_[i] = i < lower.size() ? lower[i] : -INFINITY;
      upper_[i] = i < upper.size() ? upper[i] : INFINITY;
    }
  }

  /** Sample input_limits_func_ where the reference's backward pass evaluates it (DDPSolver.hpp:470-472). */
  /** \param extra_rows further timesteps to sample beyond the horizon: a device-resident shift loop of extra_rows + 1 ticks
      starts every tick one timestep later (nmpc_hip_ddp_set_input_limits_schedule) */
  inline void sampleInputLimits(const std::vector<double> & current_t, int extra_rows = 0)
  {
    const size_t B = static_cast<size_t>(batch_size_);
    if(!limits_from_func_ || current_t.size() != B) // (a wrong batch size is reported by packInputs)
    {
      return;
    }
    const int T = config_.horizon_steps + (extra_rows > 0 ? extra_rows : 0);
    horizon_rows_ = T;
    const double dt = problem_->dt();
    bool same_t0 = true;
    for(size_t b = 1; b < B && b < current_t.size(); b++)
    {
      same_t0 = same_t0 && current_t[b] == current_t[0];
    }
    const size_t tables = same_t0 ? 1 : B;
    horizon_lo_.assign(tables * T * MM, -INFINITY);
    horizon_up_.assign(tables * T * MM, INFINITY);
    bool constant = true;
    for(size_t tb = 0; tb < tables; tb++)
    {
      for(int i = 0; i < T; i++)
      {
        const std::array<InputDimVector, 2> lim = input_limits_func_(current_t[tb] + i * dt);
        if(lim[0].size() != lim[1].size() || lim[0].size() > MM)
        {
          throw std::invalid_argument("input_limits_func should return vectors of the input dimension.");
        }
        for(int a = 0; a < lim[0].size(); a++)
        {
          horizon_lo_[(tb * T + i) * MM + a] = lim[0][a];
          horizon_up_[(tb * T + i) * MM + a] = lim[1][a];
          constant = constant && lim[0][a] == horizon_lo_[a] && lim[1][a] == horizon_up_[a];
        }
        if(tb == 0 && i == 0)
        {
          storeConstantLimits(lim[0], lim[1]);
        }
      }
    }
    const bool was_active = horizon_limits_active_;
    horizon_limits_active_ = !constant;
    horizon_per_instance_ = !same_t0;
    horizon_limits_dirty_ = horizon_limits_active_ || was_active;
  }

  void fetchResults()
  {
    const int T = config_.horizon_steps;
    const size_t B = static_cast<size_t>(batch_size_);
    std::vector<double> X(B * (T + 1) * StateDim), U(B * T * MM), C(B * (T + 1)), k(B * T * MM), K(B * T * StateDim * MM);
    std::vector<int> dims(B * T), iters(B);
    status_.resize(B);
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_X, X.data(), X.size() * sizeof(double)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_U, U.data(), U.size() * sizeof(double)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_COST, C.data(), C.size() * sizeof(double)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_KFF, k.data(), k.size() * sizeof(double)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_KFB, K.data(), K.size() * sizeof(double)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_INPUT_DIM, dims.data(), dims.size() * sizeof(int)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_STATUS, status_.data(), B * sizeof(int)));
    check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_ITERS, iters.data(), B * sizeof(int)));
    std::vector<double> trace;
    const size_t rows = static_cast<size_t>(config_.max_iter) + 1;
    if(config_.trace_level >= 1)
    {
      trace.resize(B * rows * NMPC_HIP_NTRACE);
      check(nmpc_hip_ddp_get(handle_, NMPC_HIP_FIELD_TRACE, trace.data(), trace.size() * sizeof(double)));
    }
    control_data_.assign(B, ControlData());
    k_list_.assign(B, {});
    K_list_.assign(B, {});
    trace_data_list_.assign(B, {});
    for(size_t b = 0; b < B; b++)
    {
      ControlData & cd = control_data_[b];
      cd.x_list.resize(T + 1);
      cd.u_list.resize(T);
      cd.cost_list.assign(C.begin() + b * (T + 1), C.begin() + (b + 1) * (T + 1));
      k_list_[b].resize(T);
      K_list_[b].resize(T);
      for(int i = 0; i <= T; i++)
      {
        for(int j = 0; j < StateDim; j++)
        {
          cd.x_list[i][j] = X[(b * (T + 1) + i) * StateDim + j];
        }
      }
      for(int i = 0; i < T; i++)
      {
        const int m = dims[b * T + i];
        cd.u_list[i].resize(m);
        k_list_[b][i].resize(m);
        K_list_[b][i].resize(m, StateDim);
        for(int a = 0; a < m; a++)
        {
          cd.u_list[i][a] = U[(b * T + i) * MM + a];
          k_list_[b][i][a] = k[(b * T + i) * MM + a];
          for(int c = 0; c < StateDim; c++)
          {
            K_list_[b][i](a, c) = K[((b * T + i) * StateDim + c) * MM + a];
          }
        }
      }
      if(config_.trace_level >= 1)
      {
        for(int r = 0; r <= iters[b]; r++)
        {
          const double * t = &trace[(b * rows + r) * NMPC_HIP_NTRACE];
          TraceData td;
          td.iter = static_cast<int>(t[NMPC_HIP_TRACE_ITER]);
          td.cost = t[NMPC_HIP_TRACE_COST];
          td.lambda = t[NMPC_HIP_TRACE_LAMBDA];
          td.dlambda = t[NMPC_HIP_TRACE_DLAMBDA];
          td.alpha = t[NMPC_HIP_TRACE_ALPHA];
          td.k_rel_norm = t[NMPC_HIP_TRACE_K_REL_NORM];
          td.cost_update_actual = t[NMPC_HIP_TRACE_COST_UPDATE_ACTUAL];
          td.cost_update_expected = t[NMPC_HIP_TRACE_COST_UPDATE_EXPECTED];
          td.cost_update_ratio = t[NMPC_HIP_TRACE_COST_UPDATE_RATIO];
          td.alpha_idx = static_cast<int>(t[NMPC_HIP_TRACE_ALPHA_IDX]);
          td.n_backward = static_cast<int>(t[NMPC_HIP_TRACE_N_BACKWARD]);
          td.n_forward = static_cast<int>(t[NMPC_HIP_TRACE_N_FORWARD]);
          trace_data_list_[b].push_back(td);
        }
      }
    }
    float total_ms = 0, kernel_ms = 0;
    check(nmpc_hip_ddp_last_solve_ms(handle_, &total_ms, &kernel_ms));
    computation_duration_.solve = total_ms;
    computation_duration_.opt = kernel_ms;
    computation_duration_.setup = total_ms - kernel_ms;
    double other_ms = 0;
    check(nmpc_hip_ddp_last_solve_phases(handle_, &computation_duration_.backward, &computation_duration_.forward, &other_ms));
    fetched_ = true;
  }

protected:
  /** Validate (DDPSolver.hpp:41-58) and pack current_x / initial_u_list into the C-ABI's padded arrays. */
  void packInputs(const std::vector<double> & current_t,
                  const std::vector<StateDimVector> & current_x,
                  const std::vector<std::vector<InputDimVector>> & initial_u_list,
                  std::vector<double> & x0,
                  std::vector<double> & u0)
  {
    const int T = config_.horizon_steps;
    const size_t B = static_cast<size_t>(batch_size_);
    if(current_t.size() != B || current_x.size() != B || initial_u_list.size() != B)
    {
      throw std::invalid_argument("batch of current_t / current_x / initial_u_list should be " + std::to_string(B) + ".");
    }
    ensureHandle();
    pushState();
    x0.assign(B * StateDim, 0.0);
    u0.assign(B * T * MM, 0.0);
    std::vector<int> dims(T);
    for(size_t b = 0; b < B; b++)
    {
      // Check initial_u_list    DDPSolver.hpp:41-58
      if(static_cast<int>(initial_u_list[b].size()) != T)
      {
        throw std::invalid_argument("initial_u_list length should be " + std::to_string(T) + " but "
                                    + std::to_string(initial_u_list[b].size()) + ".");
      }
      check(nmpc_hip_ddp_input_dims(handle_, current_t[b], dims.data()));
      for(int i = 0; i < T; i++)
      {
        const InputDimVector & u = initial_u_list[b][i];
        if(u.size() != dims[i])
        {
          const double t = current_t[b] + i * problem_->dt();
          throw std::runtime_error("initial_u dimension should be " + std::to_string(dims[i]) + " but "
                                   + std::to_string(u.size()) + ". i: " + std::to_string(i)
                                   + ", time: " + std::to_string(t));
        }
        for(int a = 0; a < dims[i]; a++)
        {
          u0[(b * T + i) * MM + a] = u[a];
        }
      }
      for(int j = 0; j < StateDim; j++)
      {
        x0[b * StateDim + j] = current_x[b][j];
      }
    }
  }

protected:
  Configuration config_;
  std::shared_ptr<Problem> problem_;
  int batch_size_ = 0;
  int device_ = 0;
  nmpc_hip_ddp_handle handle_ = nullptr;
  int handle_T_ = -1;
  bool has_limits_ = false;
  bool limits_from_func_ = false;
  std::function<std::array<InputDimVector, 2>(double)> input_limits_func_;
  std::vector<double> horizon_lo_, horizon_up_; //!< sampled time-varying limits [1 or batch][horizon_rows_][MM]
  int horizon_rows_ = 0;
  bool horizon_limits_active_ = false, horizon_per_instance_ = false, horizon_limits_dirty_ = false;
  double lower_[MM];
  double upper_[MM];
  bool fetched_ = false;
  std::string kernel_; //!< setKernel(): empty = automatic
  int dispatch_batch_ = 0;
  bool in_flight_ = false; //!< a solve queued by solveAsync() has not been waited for
  std::vector<Problem> problem_batch_;
  bool problem_batch_dirty_ = false;
  std::vector<double> limits_batch_lo_, limits_batch_up_;
  bool limits_batch_dirty_ = false;
  std::vector<ControlData> control_data_;
  std::vector<std::vector<TraceData>> trace_data_list_;
  std::vector<std::vector<InputDimVector>> k_list_;
  std::vector<std::vector<InputStateDimMatrix>> K_list_;
  std::vector<int> status_;
  ComputationDuration computation_duration_;
};
/** \brief Several DDPSolverBatch objects of the same problem and batch size, each with its own handle and stream: consecutive
    batches are queued round-robin and overlap on the device (the C++ counterpart of nmpc_amd.DDPSolverPool, nmpc_amd/ddp.py).

    A batch that is solved to convergence ends with a tail — a few instances that run for hundreds of iterations (every DDPSolver
    object of the reference runs its own loop to ITS end, DDPSolver.hpp:115-123).  With the next batches already queued on other
    streams their workgroups take the CUs the converged instances have vacated; with the ragged-convergence schedule
    (Configuration::ragged_schedule; automatic = on for the queued solves of a pool) a finished instance vacates its slot within sixteen iterations
    instead of when the slowest of its workgroup is done.  The results of a batch are those of the handle it ran on — bit-identical
    to a lone DDPSolverBatch.

        DDPSolverPool<DDPProblemCartPole> pool(problem, 4096, 4);
        pool.config().max_iter = 500;                       // one Configuration for all handles
        for(const auto & batch : batches) {
          auto & solver = pool.submit(batch.t, batch.x, batch.u);   // waits for (and returns) the handle's previous batch first
          ...
        }
        pool.waitAll();                                      // then pool.solver(k).controlData(b) ... */
template<class Problem>
class DDPSolverPool
{
public:
  using Solver = DDPSolverBatch<Problem>;
  using StateDimVector = typename Solver::StateDimVector;
  using InputDimVector = typename Solver::InputDimVector;

  DDPSolverPool(const std::shared_ptr<Problem> & problem, int batch_size, int n_handles = 4, int device = 0)
  {
    if(n_handles < 1)
    {
      throw std::invalid_argument("n_handles should be positive");
    }
    // one hardware queue per handle, or the streams that share one do not overlap (nmpc_hip_ddp_request_hw_queues: effective
    // only before the first HIP call of the process — construct the pool first, or export GPU_MAX_HW_QUEUES)
    int took_effect = 0;
    nmpc_hip_ddp_request_hw_queues(std::min(std::max(n_handles, 4), 64), &took_effect);
    hw_queues_ok_ = took_effect != 0;
    for(int k = 0; k < n_handles; k++)
    {
      solvers_.emplace_back(new Solver(problem, batch_size, device));
    }
  }

  /** \brief Whether the HIP runtime has (or will have) one hardware queue per handle; false: it was already initialised with
      fewer, and fewer batches overlap than the pool has handles. */
  inline bool hwQueuesOk() const
  {
    return hw_queues_ok_;
  }

  /** \brief The Configuration of every handle (the first one's object; copied to the others at every submit()). */
  inline typename Solver::Configuration & config()
  {
    return solvers_[0]->config();
  }

  inline int size() const
  {
    return static_cast<int>(solvers_.size());
  }

  inline Solver & solver(int k)
  {
    return *solvers_.at(static_cast<size_t>(k));
  }

  /** \brief Queue one batch on the next handle (round-robin).  If that handle still has a batch in flight it is waited for
      first — read its results (the returned solver's accessors hold them until the new batch is waited for) or call wait(k)
      before submitting n_handles further batches.
      \return the solver the batch was queued on */
  Solver & submit(const std::vector<double> & current_t,
                  const std::vector<StateDimVector> & current_x,
                  const std::vector<std::vector<InputDimVector>> & initial_u_list)
  {
    Solver & s = *solvers_[static_cast<size_t>(next_)];
    next_ = (next_ + 1) % size();
    if(&s != solvers_[0].get())
    {
      s.config() = solvers_[0]->config();
    }
    s.solveAsync(current_t, current_x, initial_u_list);
    return s;
  }

  /** \brief Wait for the batch queued on handle k and fetch its results. */
  std::vector<bool> wait(int k)
  {
    return solver(k).wait();
  }

  /** \brief Wait for every handle that has a batch in flight. */
  void waitAll()
  {
    for(auto & s : solvers_)
    {
      if(s->inFlight())
      {
        s->wait();
      }
    }
  }

protected:
  std::vector<std::unique_ptr<Solver>> solvers_;
  int next_ = 0;
  bool hw_queues_ok_ = false;
};

} // namespace nmpc_amd
