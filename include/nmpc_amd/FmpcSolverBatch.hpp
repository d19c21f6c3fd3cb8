// Host-side C++ mirror of the reference's nmpc_fmpc::FmpcSolver<StateDim, InputDim, IneqDim>
// (/root/reference/nmpc_fmpc/include/nmpc_fmpc/FmpcSolver.h:17-427) for a BATCH of independent problem instances solved on one
// MI355X.  Same member names, argument meaning and error behaviour (the exception types of FmpcSolver.hpp:285-354), with a
// leading batch index where the reference has one instance.
//
// Plain C++17 (no HIP, no Eigen): everything numeric happens behind the C-ABI of <nmpc_hip_fmpc.h> in libnmpc_hip_ddp.so.  The
// problem TYPE must have been compiled into a gfx950 code object and registered (NMPC_AMD_REGISTER_FMPC_PROBLEM,
// <nmpc_amd/hip/fmpc_ops.hpp>); the problem OBJECT passed here is copied to the solver at every solve(), so mutating it
// between solves behaves as with the reference's shared_ptr.
#pragma once

#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <nmpc_amd/FmpcProblem.hpp>
#include <nmpc_hip_fmpc.h>

namespace nmpc_amd
{
/** \brief Batched FMPC solver.
    \tparam Problem problem class derived from nmpc_amd::FmpcProblem<StateDim, InputDim, IneqDim> with a
            `static constexpr const char * kName` under which its kernels are registered */
template<class Problem>
class FmpcSolverBatch
{
public:
  static constexpr int StateDim = Problem::kStateDim;
  static constexpr int InputDim = Problem::kInputDimMax;
  static constexpr int IneqDim = Problem::kIneqDim;

  using StateDimVector = typename Problem::StateDimVector;
  using InputDimVector = typename Problem::InputDimVector;
  using IneqDimVector = typename Problem::IneqDimVector;
  using InputStateDimMatrix = typename Problem::InputStateDimMatrix;
  using StateStateDimMatrix = typename Problem::StateStateDimMatrix;

  /*! \brief Configuration (FmpcSolver::Configuration, FmpcSolver.h:57-89; defaults come from the library). */
  struct Configuration
  {
    Configuration()
    {
      nmpc_hip_fmpc_config c;
      nmpc_hip_fmpc_default_config(&c);
      horizon_steps = c.horizon_steps;
      max_iter = c.max_iter;
      kkt_error_thre = c.kkt_error_thre;
      check_nan = c.check_nan != 0;
      init_complementary_variable = c.init_complementary_variable != 0;
      update_barrier_eps = c.update_barrier_eps != 0;
      break_if_llt_fails = c.break_if_llt_fails != 0;
      enable_line_search = c.enable_line_search != 0;
      merit_const_scale_from_lagrange_multipliers = c.merit_const_scale_from_lagrange_multipliers != 0;
      use_graph = c.use_graph != 0;
    }

    int print_level = 1;
    int horizon_steps = 100;
    int max_iter = 10;
    double kkt_error_thre = 1e-4;
    bool check_nan = true;
    bool init_complementary_variable = false;
    bool update_barrier_eps = true;
    bool break_if_llt_fails = false;
    bool enable_line_search = false;
    bool merit_const_scale_from_lagrange_multipliers = false;
    //! replay the kernel sequence of a solve from a hipGraph (nmpc_hip_fmpc_config::use_graph)
    bool use_graph = true;
  };

  /*! \brief Result status (FmpcSolver::Status, FmpcSolver.h:92-114). */
  enum class Status
  {
    Uninitialized = 0,
    Succeeded = 1,
    ErrorInForward = 2,
    ErrorInBackward = 3,
    ErrorInUpdate = 4,
    MaxIterationReached = 5,
    IterationContinued = 6
  };

  /*! \brief Optimization variables of ONE instance (FmpcSolver::Variable, FmpcSolver.h:117-158). */
  struct Variable
  {
    Variable(int _horizon_steps = 0) : horizon_steps(_horizon_steps)
    {
      x_list.resize(horizon_steps + 1);
      u_list.resize(horizon_steps);
      lambda_list.resize(horizon_steps + 1);
      s_list.resize(horizon_steps);
      nu_list.resize(horizon_steps);
    }

    /** \brief Reset variables (FmpcSolver.hpp:42-69). */
    void reset(double _x, double _u, double _lambda, double _s, double _nu)
    {
      for(auto & x : x_list)
      {
        x.setConstant(_x);
      }
      for(auto & u : u_list)
      {
        u.setConstant(_u);
      }
      for(auto & lambda : lambda_list)
      {
        lambda.setConstant(_lambda);
      }
      for(auto & s : s_list)
      {
        s.setConstant(_s);
      }
      for(auto & nu : nu_list)
      {
        nu.setConstant(_nu);
      }
    }

    int horizon_steps;
    std::vector<StateDimVector> x_list;
    std::vector<InputDimVector> u_list;
    std::vector<StateDimVector> lambda_list;
    std::vector<IneqDimVector> s_list;
    std::vector<IneqDimVector> nu_list;
    int print_level = 1;
  };

  /*! \brief Data to trace optimization loop (FmpcSolver::TraceData, FmpcSolver.h:230-249).  The duration_* members are CPU
      timers of one instance in the reference and stay 0; the scalars of the iteration take their place. */
  struct TraceData
  {
    int iter = 0;
    double kkt_error = 0;
    double duration_coeff = 0;
    double duration_backward = 0;
    double duration_forward = 0;
    double duration_update = 0;
    double barrier_eps = 0;
    double alpha_s_max = 0;
    double alpha_nu_max = 0;
    double alpha_s = 0;
  };

  /*! \brief Data of computation duration (FmpcSolver.h:252-287) for the whole batch [msec]: only `solve` is measured. */
  struct ComputationDuration
  {
    double solve = 0;
    double setup = 0;
    double opt = 0;
    double coeff = 0;
    double backward = 0;
    double forward = 0;
    double update = 0;
  };

public:
  /** \brief Constructor (FmpcSolver.h:270).
      \param problem FMPC problem
      \param batch number of instances
      \param horizon_steps number of steps in horizon (fixed for the lifetime of the solver)
      \param device HIP device index */
  FmpcSolverBatch(const std::shared_ptr<Problem> & problem, int batch, int horizon_steps = 100, int device = 0)
  : problem_(problem), batch_(batch)
  {
    config_.horizon_steps = horizon_steps;
    check(nmpc_hip_fmpc_create(Problem::kName, horizon_steps, batch, device, &handle_));
  }

  ~FmpcSolverBatch()
  {
    nmpc_hip_fmpc_destroy(handle_);
  }

  FmpcSolverBatch(const FmpcSolverBatch &) = delete;
  FmpcSolverBatch & operator=(const FmpcSolverBatch &) = delete;

  /** \brief Accessor to configuration. */
  inline Configuration & config()
  {
    return config_;
  }

  inline const Configuration & config() const
  {
    return config_;
  }

  inline int batch() const
  {
    return batch_;
  }

  /** \brief One problem object per instance (a batch of FmpcSolver objects each constructed with its own problem). */
  void setProblems(const std::vector<Problem> & problems)
  {
    if(static_cast<int>(problems.size()) != batch_)
    {
      throw std::invalid_argument("[FMPC] problems length should be " + std::to_string(batch_) + ".");
    }
    check(nmpc_hip_fmpc_set_problem(handle_, problems.data(), problems.size() * sizeof(Problem), 1));
    own_problems_ = true;
  }

  /** \brief Solve optimization for every instance (FmpcSolver::solve, FmpcSolver.h:283).
      \param current_t current time [sec] of each instance
      \param current_x current state of each instance
      \param initial_variable initial guess of each instance; empty: continue from the resident variables (the
             `variable = solver.variable()` of the reference's callers without the copy)
      \return result status of each instance */
  std::vector<Status> solve(const std::vector<double> & current_t,
                            const std::vector<StateDimVector> & current_x,
                            const std::vector<Variable> & initial_variable = {})
  {
    if(static_cast<int>(current_t.size()) != batch_ || static_cast<int>(current_x.size()) != batch_)
    {
      throw std::invalid_argument("[FMPC] current_t / current_x length should be " + std::to_string(batch_) + ".");
    }
    pushConfig();
    if(!own_problems_)
    {
      check(nmpc_hip_fmpc_set_problem(handle_, problem_.get(), sizeof(Problem), 0));
    }
    if(!initial_variable.empty())
    {
      setVariable(initial_variable);
    }
    std::vector<double> x0(static_cast<size_t>(batch_) * StateDim);
    for(int b = 0; b < batch_; b++)
    {
      for(int i = 0; i < StateDim; i++)
      {
        x0[static_cast<size_t>(b) * StateDim + i] = current_x[b][i];
      }
    }
    check(nmpc_hip_fmpc_solve(handle_, current_t.data(), x0.data()));
    std::vector<int> st(batch_);
    check(nmpc_hip_fmpc_get(handle_, NMPC_HIP_FMPC_FIELD_STATUS, st.data(), st.size() * sizeof(int), 0));
    std::vector<Status> out(batch_);
    for(int b = 0; b < batch_; b++)
    {
      out[b] = static_cast<Status>(st[b]);
    }
    return out;
  }

  /** \brief Upload the initial guess of every instance; checkVariable's length tests (FmpcSolver.hpp:287-311). */
  void setVariable(const std::vector<Variable> & variable)
  {
    const int T = config_.horizon_steps;
    if(static_cast<int>(variable.size()) != batch_)
    {
      throw std::invalid_argument("[FMPC] variable length should be " + std::to_string(batch_) + ".");
    }
    std::vector<double> x(static_cast<size_t>(batch_) * (T + 1) * StateDim), u(static_cast<size_t>(batch_) * T * InputDim),
        lambda(x.size()), s(static_cast<size_t>(batch_) * T * IneqDim), nu(s.size());
    for(int b = 0; b < batch_; b++)
    {
      const Variable & v = variable[b];
      checkLength("x_list", v.x_list.size(), T + 1);
      checkLength("u_list", v.u_list.size(), T);
      checkLength("lambda_list", v.lambda_list.size(), T + 1);
      checkLength("s_list", v.s_list.size(), T);
      checkLength("nu_list", v.nu_list.size(), T);
      for(int i = 0; i <= T; i++)
      {
        for(int e = 0; e < StateDim; e++)
        {
          x[(static_cast<size_t>(b) * (T + 1) + i) * StateDim + e] = v.x_list[i][e];
          lambda[(static_cast<size_t>(b) * (T + 1) + i) * StateDim + e] = v.lambda_list[i][e];
        }
      }
      for(int i = 0; i < T; i++)
      {
        for(int e = 0; e < InputDim; e++)
        {
          u[(static_cast<size_t>(b) * T + i) * InputDim + e] = v.u_list[i][e];
        }
        for(int e = 0; e < IneqDim; e++)
        {
          s[(static_cast<size_t>(b) * T + i) * IneqDim + e] = v.s_list[i][e];
          nu[(static_cast<size_t>(b) * T + i) * IneqDim + e] = v.nu_list[i][e];
        }
      }
    }
    check(nmpc_hip_fmpc_set_variable(handle_, x.data(), u.empty() ? nullptr : u.data(), lambda.data(), s.empty() ? nullptr : s.data(),
                                     nu.empty() ? nullptr : nu.data(), nullptr, 0));
  }

  /** \brief Optimization variables of every instance (FmpcSolver::variable, FmpcSolver.h:286-289). */
  std::vector<Variable> variable() const
  {
    const int T = config_.horizon_steps;
    std::vector<double> x = getField(NMPC_HIP_FMPC_FIELD_X), u = getField(NMPC_HIP_FMPC_FIELD_U),
                        lambda = getField(NMPC_HIP_FMPC_FIELD_LAMBDA), s = getField(NMPC_HIP_FMPC_FIELD_S),
                        nu = getField(NMPC_HIP_FMPC_FIELD_NU);
    std::vector<Variable> out(batch_, Variable(T));
    for(int b = 0; b < batch_; b++)
    {
      for(int i = 0; i <= T; i++)
      {
        for(int e = 0; e < StateDim; e++)
        {
          out[b].x_list[i][e] = x[(static_cast<size_t>(b) * (T + 1) + i) * StateDim + e];
          out[b].lambda_list[i][e] = lambda[(static_cast<size_t>(b) * (T + 1) + i) * StateDim + e];
        }
      }
      for(int i = 0; i < T; i++)
      {
        for(int e = 0; e < InputDim; e++)
        {
          out[b].u_list[i][e] = u[(static_cast<size_t>(b) * T + i) * InputDim + e];
        }
        for(int e = 0; e < IneqDim; e++)
        {
          out[b].s_list[i][e] = s[(static_cast<size_t>(b) * T + i) * IneqDim + e];
          out[b].nu_list[i][e] = nu[(static_cast<size_t>(b) * T + i) * IneqDim + e];
        }
      }
    }
    return out;
  }

  /** \brief Feedback gain K of timestep `step` for every instance (coeffList()[step].K, FmpcSolver.h:217,292-295; the
      cart-pole caller uses coeffList().front().K, TestFmpcCartPole.cpp:350). */
  std::vector<InputStateDimMatrix> feedbackGain(int step = 0) const
  {
    const int T = config_.horizon_steps;
    const std::vector<double> K = getField(NMPC_HIP_FMPC_FIELD_GAIN_K);
    std::vector<InputStateDimMatrix> out(batch_);
    for(int b = 0; b < batch_; b++)
    {
      for(int e = 0; e < InputDim * StateDim; e++)
      {
        out[b].data()[e] = K[(static_cast<size_t>(b) * T + step) * InputDim * StateDim + e];
      }
    }
    return out;
  }

  /** \brief Trace data list of one instance (FmpcSolver::traceDataList, FmpcSolver.h:298-301). */
  std::vector<TraceData> traceDataList(int instance) const
  {
    const std::vector<double> tr = getField(NMPC_HIP_FMPC_FIELD_TRACE);
    std::vector<int> iters(batch_);
    check(nmpc_hip_fmpc_get(handle_, NMPC_HIP_FMPC_FIELD_ITERS, iters.data(), iters.size() * sizeof(int), 0));
    std::vector<TraceData> out(iters.at(instance));
    for(int k = 0; k < iters[instance]; k++)
    {
      const double * row = &tr[(static_cast<size_t>(instance) * config_.max_iter + k) * NMPC_HIP_FMPC_NTRACE];
      out[k].iter = static_cast<int>(row[NMPC_HIP_FMPC_TRACE_ITER]);
      out[k].kkt_error = row[NMPC_HIP_FMPC_TRACE_KKT_ERROR];
      out[k].barrier_eps = row[NMPC_HIP_FMPC_TRACE_BARRIER_EPS];
      out[k].alpha_s_max = row[NMPC_HIP_FMPC_TRACE_ALPHA_S_MAX];
      out[k].alpha_nu_max = row[NMPC_HIP_FMPC_TRACE_ALPHA_NU_MAX];
      out[k].alpha_s = row[NMPC_HIP_FMPC_TRACE_ALPHA_S];
    }
    return out;
  }

  /** \brief Computation duration of the last solve (FmpcSolver::computationDuration, FmpcSolver.h:304-307). */
  ComputationDuration computationDuration() const
  {
    ComputationDuration d;
    float ms = 0;
    check(nmpc_hip_fmpc_last_solve_ms(handle_, &ms));
    d.solve = ms;
    d.opt = ms;
    return d;
  }

  /** \brief Dump trace data list of one instance (FmpcSolver::dumpTraceDataList, FmpcSolver.hpp:257-283). */
  void dumpTraceDataList(const std::string & file_path, int instance = 0) const
  {
    std::ofstream ofs(file_path);
    ofs << "iter kkt_error duration_coeff duration_backward duration_forward duration_update" << std::endl;
    for(const auto & trace_data : traceDataList(instance))
    {
      ofs << trace_data.iter << " " << trace_data.kkt_error << " " << trace_data.duration_coeff << " "
          << trace_data.duration_backward << " " << trace_data.duration_forward << " " << trace_data.duration_update << std::endl;
    }
  }

  /** \brief The C-ABI handle (device-pointer entry points, nmpc_hip_fmpc_mpc_run). */
  nmpc_hip_fmpc_handle handle() const
  {
    return handle_;
  }

protected:
  static void check(int rc)
  {
    if(rc == NMPC_HIP_OK)
    {
      return;
    }
    const std::string msg = nmpc_hip_fmpc_last_error();
    if(rc == NMPC_HIP_ERR_INVALID_ARGUMENT || rc == NMPC_HIP_ERR_UNKNOWN_MODEL)
    {
      throw std::invalid_argument(msg);
    }
    throw std::runtime_error(msg);
  }

  void checkLength(const char * name, size_t have, int want) const
  {
    if(static_cast<int>(have) != want)
    {
      throw std::invalid_argument(std::string("[FMPC] ") + name + " length should be " + std::to_string(want) + " but "
                                  + std::to_string(have) + ".");
    }
  }

  void pushConfig()
  {
    nmpc_hip_fmpc_config c;
    nmpc_hip_fmpc_default_config(&c);
    c.horizon_steps = config_.horizon_steps;
    c.max_iter = config_.max_iter;
    c.kkt_error_thre = config_.kkt_error_thre;
    c.check_nan = config_.check_nan;
    c.init_complementary_variable = config_.init_complementary_variable;
    c.update_barrier_eps = config_.update_barrier_eps;
    c.break_if_llt_fails = config_.break_if_llt_fails;
    c.enable_line_search = config_.enable_line_search;
    c.merit_const_scale_from_lagrange_multipliers = config_.merit_const_scale_from_lagrange_multipliers;
    c.use_graph = config_.use_graph;
    c.time_kernels = 0;
    nmpc_hip_fmpc_config cur;
    check(nmpc_hip_fmpc_get_config(handle_, &cur));
    if(cur.max_iter != c.max_iter || cur.kkt_error_thre != c.kkt_error_thre || cur.check_nan != c.check_nan
       || cur.init_complementary_variable != c.init_complementary_variable || cur.update_barrier_eps != c.update_barrier_eps
       || cur.break_if_llt_fails != c.break_if_llt_fails || cur.enable_line_search != c.enable_line_search
       || cur.merit_const_scale_from_lagrange_multipliers != c.merit_const_scale_from_lagrange_multipliers
       || cur.use_graph != c.use_graph || cur.horizon_steps != c.horizon_steps)
    {
      check(nmpc_hip_fmpc_set_config(handle_, &c));
    }
  }

  std::vector<double> getField(int field) const
  {
    size_t bytes = 0;
    check(nmpc_hip_fmpc_field_bytes(handle_, field, &bytes));
    std::vector<double> out(bytes / sizeof(double));
    if(bytes > 0)
    {
      check(nmpc_hip_fmpc_get(handle_, field, out.data(), bytes, 0));
    }
    return out;
  }

protected:
  std::shared_ptr<Problem> problem_;
  int batch_ = 0;
  Configuration config_;
  nmpc_hip_fmpc_handle handle_ = nullptr;
  bool own_problems_ = false;
};
} // namespace nmpc_amd
