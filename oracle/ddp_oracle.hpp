// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement ("oracle") of the reference's DDP hot path, used only by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / reported baseline.
// Nothing under nmpc_amd/ or include/ may include, link or call this file.
//
// PARITY STATUS: the reference (header-only C++ on Eigen3) cannot be built in this image (Eigen3 is
// not installed and must not be stubbed), so this restatement is pinned by
//   (1) the reference's own known-answer vectors for BoxQP (nmpc_ddp/tests/src/TestBoxQP.cpp:35-98),
//   (2) its finite-difference derivative checks (TestDDPCartPole.cpp:609-649,
//       TestDDPCentroidalMotion.cpp:367-411),
//   (3) every EXPECT_LT of its closed-loop MPC tests (TestDDPBipedal.cpp:162-279,
//       TestDDPVerticalMotion.cpp:236-347, TestDDPCentroidalMotion.cpp:239-365,
//       TestDDPCartPole.cpp:336,351-354),
//   (4) an independent NumPy/SciPy restatement (oracle/ddp_numpy.py).
// Solver internals (k, K, dV, per-iteration cost, alpha index, iteration count) have NO golden vectors in
// the reference: for those, "parity unpinned" beyond (3)+(4).
//
// What is restated (all citations relative to /root/reference/nmpc_ddp/include/nmpc_ddp/):
//   DDPSolver::Configuration defaults        DDPSolver.h:47-110
//   DDPSolver::solve                         DDPSolver.hpp:26-141
//   DDPSolver::procOnce                      DDPSolver.hpp:143-340
//   DDPSolver::backwardPass                  DDPSolver.hpp:342-534
//   DDPSolver::forwardPass                   DDPSolver.hpp:536-560
//   BoxQP::solve                             BoxQP.h:141-347 (defaults BoxQP.h:33-55)
// Eigen semantics that matter are reproduced by hand: column-major storage,
// triple products evaluated left to right through a temporary, unblocked lower LLT that fails iff a
// pivot is <= 0 (NaN pivots pass), llt.solve = forward then backward substitution.  Reductions are
// summed in ascending index order (Eigen's own order is implementation-defined, so bit parity with an
// Eigen build is not a meaningful target; tolerance + exact indices is).
// Precision: the header is written against `Real`.  Included as is, Real = double in namespace `oracle` (the
// reference's arithmetic, DDPProblem.h:20-35).  oracle_capi.cpp includes it a second time with ORACLE_F32 defined:
// Real = float in namespace `oracle_f32` — the same statements instantiated in single precision, which is what the fp32
// HIP path (BASELINE.json config 4) is compared with ("the oracle instantiated in fp32").
#if defined(ORACLE_F32)
#  ifdef ORACLE_DDP_ORACLE_F32_HPP
#    error "ddp_oracle.hpp (fp32) included twice"
#  endif
#  define ORACLE_DDP_ORACLE_F32_HPP
#  define ORACLE_NS oracle_f32
#  define ORACLE_REAL float
#else
#  ifdef ORACLE_DDP_ORACLE_F64_HPP
#    error "ddp_oracle.hpp (fp64) included twice"
#  endif
#  define ORACLE_DDP_ORACLE_F64_HPP
#  define ORACLE_NS oracle
#  define ORACLE_REAL double
#endif


#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace ORACLE_NS
{
using Real = ORACLE_REAL;
// ---------------------------------------------------------------------------------------------------
// small dense helpers, column-major, explicit leading dimension = row count
// ---------------------------------------------------------------------------------------------------

// C(r x c) = A^T * B  with A (k x r), B (k x c)
inline void mulAtB(const Real * A, const Real * B, Real * C, int r, int k, int c)
{
  for(int j = 0; j < c; j++)
  {
    for(int i = 0; i < r; i++)
    {
      Real s = 0;
      for(int p = 0; p < k; p++)
      {
        s += A[p + i * k] * B[p + j * k];
      }
      C[i + j * r] = s;
    }
  }
}

// C(r x c) = A * B  with A (r x k), B (k x c)
inline void mulAB(const Real * A, const Real * B, Real * C, int r, int k, int c)
{
  for(int j = 0; j < c; j++)
  {
    for(int i = 0; i < r; i++)
    {
      Real s = 0;
      for(int p = 0; p < k; p++)
      {
        s += A[i + p * r] * B[p + j * k];
      }
      C[i + j * r] = s;
    }
  }
}

/** Unblocked lower Cholesky in place (Eigen::internal::llt_inplace<Lower>::unblocked).
    \return -1 on success, otherwise the index of the failing pivot (pivot <= 0; NaN passes) */
inline int lltInPlace(Real * A, int n)
{
  for(int k = 0; k < n; k++)
  {
    Real x = A[k + k * n];
    for(int j = 0; j < k; j++)
    {
      x -= A[k + j * n] * A[k + j * n];
    }
    if(x <= 0)
    {
      return k;
    }
    x = std::sqrt(x);
    A[k + k * n] = x;
    for(int i = k + 1; i < n; i++)
    {
      Real s = A[i + k * n];
      for(int j = 0; j < k; j++)
      {
        s -= A[i + j * n] * A[k + j * n];
      }
      A[i + k * n] = s / x;
    }
  }
  return -1;
}

/** Solve L L^T X = B in place, B is (n x c) column-major, L the lower factor stored in A. */
inline void lltSolveInPlace(const Real * L, int n, Real * B, int c)
{
  for(int col = 0; col < c; col++)
  {
    Real * b = B + col * n;
    for(int i = 0; i < n; i++)
    {
      Real s = b[i];
      for(int j = 0; j < i; j++)
      {
        s -= L[i + j * n] * b[j];
      }
      b[i] = s / L[i + i * n];
    }
    for(int i = n - 1; i >= 0; i--)
    {
      Real s = b[i];
      for(int j = i + 1; j < n; j++)
      {
        s -= L[j + i * n] * b[j];
      }
      b[i] = s / L[i + i * n];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// BoxQP  (BoxQP.h:141-347)
// ---------------------------------------------------------------------------------------------------
struct BoxQPConfig
{
  int max_iter = 500; // BoxQP.h:39
  Real grad_thre = 1e-8; // BoxQP.h:42
  Real rel_improve_thre = 1e-8; // BoxQP.h:45
  Real step_factor = 0.6; // BoxQP.h:48
  Real min_step = 1e-22; // BoxQP.h:51
  Real armijo_param = 0.1; // BoxQP.h:54
};

/** Projected-Newton box QP.  After solve(): x, retval, free_idxs, llt_free (lower factor of H[free,free],
    dimension n_llt x n_llt — may be stale exactly as in the reference when the loop exits before
    refactorising), iter, factorization_num. */
struct BoxQP
{
  BoxQPConfig config;
  int retval = 0;
  int iter = 0;
  int factorization_num = 0;
  int total_step_num = 0;
  std::vector<Real> x;
  std::vector<int> free_idxs;
  std::vector<Real> llt_free;
  int n_llt = 0;

  static Real objective(int m, const Real * H, const Real * g, const Real * x)
  {
    // x.dot(g) + 0.5 * x.dot(H * x)      BoxQP.h:149,297,303
    Real xg = 0;
    for(int i = 0; i < m; i++)
    {
      xg += x[i] * g[i];
    }
    Real xHx = 0;
    for(int i = 0; i < m; i++)
    {
      Real hx = 0;
      for(int j = 0; j < m; j++)
      {
        hx += H[i + j * m] * x[j];
      }
      xHx += x[i] * hx;
    }
    return xg + Real(0.5) * xHx;
  }

  void solve(int m,
             const Real * H,
             const Real * g,
             const Real * lower,
             const Real * upper,
             const Real * initial_x)
  {
    x.assign(m, Real(0));
    for(int i = 0; i < m; i++)
    {
      // initial_x.cwiseMin(upper).cwiseMax(lower)    BoxQP.h:148
      x[i] = std::max(std::min(initial_x ? initial_x[i] : Real(0), upper[i]), lower[i]);
    }
    Real obj = objective(m, H, g, x.data());
    Real old_obj = obj;

    retval = 0;
    factorization_num = 0;
    total_step_num = 0;
    std::vector<Real> grad(m, Real(0));
    std::vector<char> clamped(m, 0), old_clamped(m, 0);
    std::vector<int> clamped_idxs;
    std::vector<Real> search_dir(m), x_cand(m), rhs;
    free_idxs.clear();
    for(iter = 1;; iter++)
    {
      // relative improvement    BoxQP.h:176-181
      if(iter > 1 && (old_obj - obj) < config.rel_improve_thre * std::abs(old_obj))
      {
        retval = 4;
        break;
      }
      old_obj = obj;

      // gradient    BoxQP.h:184
      for(int i = 0; i < m; i++)
      {
        Real hx = 0;
        for(int j = 0; j < m; j++)
        {
          hx += H[i + j * m] * x[j];
        }
        grad[i] = g[i] + hx;
      }

      // clamped set, exact == compare    BoxQP.h:187-206
      old_clamped = clamped;
      clamped_idxs.clear();
      free_idxs.clear();
      bool all_clamped = true;
      for(int i = 0; i < m; i++)
      {
        clamped[i] = ((x[i] == lower[i] && grad[i] > 0) || (x[i] == upper[i] && grad[i] < 0)) ? 1 : 0;
        if(clamped[i])
        {
          clamped_idxs.push_back(i);
        }
        else
        {
          free_idxs.push_back(i);
          all_clamped = false;
        }
      }
      if(all_clamped)
      {
        retval = 6; // BoxQP.h:209-213
        break;
      }

      // factorise the free block iff the clamped set changed    BoxQP.h:216-241
      int nf = static_cast<int>(free_idxs.size());
      int nc = static_cast<int>(clamped_idxs.size());
      if(iter == 1 || clamped != old_clamped)
      {
        llt_free.assign(static_cast<size_t>(nf) * nf, Real(0));
        n_llt = nf;
        for(int i = 0; i < nf; i++)
        {
          for(int j = 0; j < nf; j++)
          {
            llt_free[i + j * nf] = H[free_idxs[i] + free_idxs[j] * m];
          }
        }
        if(lltInPlace(llt_free.data(), nf) >= 0)
        {
          retval = -1;
          break;
        }
        factorization_num++;
      }

      // free gradient norm    BoxQP.h:244-253
      Real grad_norm = 0;
      for(int i = 0; i < nf; i++)
      {
        grad_norm += grad[free_idxs[i]] * grad[free_idxs[i]];
      }
      if(grad_norm < config.grad_thre * config.grad_thre)
      {
        retval = 5;
        break;
      }

      // Newton direction on the free dims    BoxQP.h:256-279
      rhs.assign(nf, Real(0));
      for(int i = 0; i < nf; i++)
      {
        Real s = 0;
        for(int j = 0; j < nc; j++)
        {
          s += H[free_idxs[i] + clamped_idxs[j] * m] * x[clamped_idxs[j]];
        }
        rhs[i] = g[free_idxs[i]] + s;
      }
      lltSolveInPlace(llt_free.data(), nf, rhs.data(), 1);
      std::fill(search_dir.begin(), search_dir.end(), Real(0));
      for(int i = 0; i < nf; i++)
      {
        search_dir[free_idxs[i]] = -1 * rhs[i] - x[free_idxs[i]];
      }

      // descent check    BoxQP.h:282-291
      Real sdg = 0;
      for(int i = 0; i < m; i++)
      {
        sdg += search_dir[i] * grad[i];
      }
      if(sdg > Real(1e-10))
      {
        retval = -2;
        break;
      }

      // Armijo line search with projection    BoxQP.h:294-309
      Real step = 1;
      for(int i = 0; i < m; i++)
      {
        x_cand[i] = std::max(std::min(x[i] + step * search_dir[i], upper[i]), lower[i]);
      }
      Real obj_cand = objective(m, H, g, x_cand.data());
      while((obj_cand - old_obj) / (step * sdg) < config.armijo_param)
      {
        step = step * config.step_factor;
        total_step_num++;
        for(int i = 0; i < m; i++)
        {
          x_cand[i] = std::max(std::min(x[i] + step * search_dir[i], upper[i]), lower[i]);
        }
        obj_cand = objective(m, H, g, x_cand.data());
        if(step < config.min_step)
        {
          retval = 2; // only leaves the inner loop (BoxQP.h:304-308)
          break;
        }
      }

      // accept    BoxQP.h:328-329
      x = x_cand;
      obj = obj_cand;

      if(iter == config.max_iter)
      {
        retval = 1; // BoxQP.h:332-336
        break;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// DDP solver
// ---------------------------------------------------------------------------------------------------
struct Config
{
  // DDPSolver.h:47-110
  bool with_input_constraint = false;
  int max_iter = 500;
  int horizon_steps = 100;
  int reg_type = 1;
  Real initial_lambda = 1e-4;
  Real initial_dlambda = 1.0;
  Real lambda_factor = 1.6;
  Real lambda_min = 1e-6;
  Real lambda_max = 1e10;
  Real k_rel_norm_thre = 1e-4;
  Real lambda_thre = 1e-5;
  std::vector<Real> alpha_list;
  Real cost_update_ratio_thre = 0;
  Real cost_update_thre = 1e-7;

  Config()
  {
    // alpha_list[i] = std::pow(10, LinSpaced(11, 0, -3)[i])    DDPSolver.h:50-60
    // Eigen's LinSpaced (no flip since |high| >= |low|): low + i*step for i < size-1, exactly high at the end.
    const int list_size = 11;
    const Real low = 0, high = -3;
    const Real step = (high - low) / (list_size - 1);
    alpha_list.resize(list_size);
    for(int i = 0; i < list_size; i++)
    {
      Real e = (i == list_size - 1) ? high : (low + i * step);
      alpha_list[i] = std::pow(10, e);
    }
  }
};

/** One row of DDPSolver::TraceData (DDPSolver.h:179-216) plus three integer diagnostics. */
struct TraceRow
{
  int iter = 0;
  Real cost = 0;
  Real lambda = 0;
  Real dlambda = 0;
  Real alpha = 0;
  Real k_rel_norm = 0;
  Real cost_update_actual = 0;
  Real cost_update_expected = 0;
  Real cost_update_ratio = 0;
  // extras (not in the reference struct): discrete decisions the GPU path must reproduce exactly
  int alpha_idx = -1; // index into alpha_list of the last trial (-1 if the line search was not reached)
  int n_backward = 0; // number of backwardPass() calls in this iteration
  int n_forward = 0; // number of forwardPass() calls in this iteration
};

/** Restatement of DDPSolver<StateDim, InputDim>.  N = StateDim, MMAX = largest input dimension the model
    can return from inputDim(t) (the reference's Eigen::Dynamic case).  Model is a plain struct exposing the
    nine DDPProblem methods on raw arrays (see oracle/models.hpp). */
template<class Model>
class DDPSolver
{
public:
  static constexpr int N = Model::N;
  static constexpr int MMAX = Model::MMAX;

  struct Deriv
  {
    int m = 0;
    Real Fx[N * N];
    Real Fu[N * (MMAX > 0 ? MMAX : 1)];
    Real Lx[N];
    Real Lu[MMAX > 0 ? MMAX : 1];
    Real Lxx[N * N];
    Real Luu[MMAX > 0 ? MMAX * MMAX : 1];
    Real Lxu[N * (MMAX > 0 ? MMAX : 1)];
  };

  struct ControlData
  {
    std::vector<Real> x; // (T+1) * N
    std::vector<Real> u; // T * MMAX (first m_i entries of each row valid)
    std::vector<Real> cost; // T+1
  };

  explicit DDPSolver(const Model & model) : model_(model) {}

  Config & config()
  {
    return config_;
  }

  /** Input limits, constant in time: lower[MMAX], upper[MMAX] (entries beyond inputDim(t) ignored).
      Mirrors setInputLimitsFunc for the only form the reference tests use (TestDDPCartPole.cpp:379-386,
      TestDDPVerticalMotion.cpp:262-270). */
  void setInputLimits(const Real * lower, const Real * upper)
  {
    lower_.assign(lower, lower + MMAX);
    upper_.assign(upper, upper + MMAX);
    limits_per_step_ = false;
  }

  /** Input limits that vary in time: setInputLimitsFunc (DDPSolver.h:282-285) with the function sampled where the solver
      evaluates it, input_limits_func_(current_t + i * dt) for i < horizon_steps (DDPSolver.hpp:470-472):
      lower[T][MMAX], upper[T][MMAX]. */
  void setInputLimitsPerStep(const Real * lower, const Real * upper, int T)
  {
    lower_.assign(lower, lower + static_cast<size_t>(T) * MMAX);
    upper_.assign(upper, upper + static_cast<size_t>(T) * MMAX);
    limits_per_step_ = true;
  }

  const ControlData & controlData() const
  {
    return control_;
  }
  const std::vector<TraceRow> & traceDataList() const
  {
    return trace_;
  }
  const std::vector<Real> & kList() const
  {
    return k_list_;
  }
  const std::vector<Real> & KList() const
  {
    return K_list_;
  }
  const std::vector<int> & inputDimList() const
  {
    return m_list_;
  }
  int status() const
  {
    return status_;
  }
  const Real * dV() const
  {
    return dV_;
  }
  /** BoxQP return codes / free sets of the LAST backward pass (per step), for index-parity checks. */
  const std::vector<int> & qpRetvalList() const
  {
    return qp_retval_;
  }
  const std::vector<unsigned> & qpFreeMaskList() const
  {
    return qp_free_mask_;
  }

  /** DDPSolver::solve (DDPSolver.hpp:26-141).  u_init is T rows of MMAX Reals.
      \return retval == 1 */
  bool solve(Real current_t, const Real * current_x, const Real * u_init)
  {
    const int T = config_.horizon_steps;
    current_t_ = current_t;
    lambda_ = config_.initial_lambda; // :37
    dlambda_ = config_.initial_dlambda; // :38

    // per-step input dimension (the reference validates initial_u_list sizes against it, :46-58)
    m_list_.resize(T);
    for(int i = 0; i < T; i++)
    {
      Real t = current_t_ + i * model_.dt;
      m_list_[i] = model_.inputDim(t);
      if(m_list_[i] < 0 || m_list_[i] > MMAX)
      {
        throw std::runtime_error("inputDim(t) out of range");
      }
    }

    cand_.x.assign(static_cast<size_t>(T + 1) * N, Real(0));
    cand_.u.assign(static_cast<size_t>(T) * (MMAX > 0 ? MMAX : 1), Real(0));
    cand_.cost.assign(T + 1, Real(0));
    deriv_.resize(T);
    k_list_.assign(static_cast<size_t>(T) * (MMAX > 0 ? MMAX : 1), Real(0));
    K_list_.assign(static_cast<size_t>(T) * (MMAX > 0 ? MMAX : 1) * N, Real(0));
    qp_retval_.assign(T, 0);
    qp_free_mask_.assign(T, 0u);

    // initial rollout    :83-95
    control_.u.assign(u_init, u_init + static_cast<size_t>(T) * (MMAX > 0 ? MMAX : 1));
    control_.x.assign(static_cast<size_t>(T + 1) * N, Real(0));
    control_.cost.assign(T + 1, Real(0));
    for(int j = 0; j < N; j++)
    {
      control_.x[j] = current_x[j];
    }
    for(int i = 0; i < T; i++)
    {
      Real t = current_t_ + i * model_.dt;
      model_.stateEq(t, &control_.x[i * N], &control_.u[i * MM()], m_list_[i], &control_.x[(i + 1) * N]);
      control_.cost[i] = model_.runningCost(t, &control_.x[i * N], &control_.u[i * MM()], m_list_[i]);
    }
    Real terminal_t = current_t_ + T * model_.dt;
    control_.cost[T] = model_.terminalCost(terminal_t, &control_.x[T * N]);

    // trace[0]    :98-104
    trace_.clear();
    TraceRow row0;
    row0.iter = 0;
    row0.cost = sum(control_.cost);
    row0.lambda = lambda_;
    row0.dlambda = dlambda_;
    trace_.push_back(row0);

    // optimisation loop    :115-123
    int retval = 0;
    for(int iter = 1; iter <= config_.max_iter; iter++)
    {
      retval = procOnce(iter);
      if(retval != 0)
      {
        break;
      }
    }
    status_ = retval;
    return retval == 1; // :140
  }

protected:
  static constexpr int MM()
  {
    return MMAX > 0 ? MMAX : 1;
  }

  static Real sum(const std::vector<Real> & v)
  {
    Real s = 0;
    for(Real e : v)
    {
      s += e;
    }
    return s;
  }

  /** DDPSolver::procOnce (DDPSolver.hpp:143-340): 0 continue, 1 terminate, -1 failure. */
  int procOnce(int iter)
  {
    const int T = config_.horizon_steps;
    trace_.push_back(TraceRow());
    TraceRow & tr = trace_.back();
    tr.iter = iter;

    // Step 1: linearise along the current trajectory (recomputed every iteration)    :157-185
    for(int i = 0; i < T; i++)
    {
      Deriv & d = deriv_[i];
      Real t = current_t_ + i * model_.dt;
      d.m = m_list_[i];
      const Real * x = &control_.x[i * N];
      const Real * u = &control_.u[i * MM()];
      model_.calcStateEqDeriv(t, x, u, d.m, d.Fx, d.Fu);
      model_.calcRunningCostDeriv(t, x, u, d.m, d.Lx, d.Lu, d.Lxx, d.Luu, d.Lxu);
    }
    Real terminal_t = current_t_ + T * model_.dt;
    model_.calcTerminalCostDeriv(terminal_t, &control_.x[T * N], last_Vx_, last_Vxx_);

    // Step 2: backward pass with regularisation retries    :188-214
    tr.n_backward = 1;
    while(!backwardPass())
    {
      dlambda_ = std::max(dlambda_ * config_.lambda_factor, config_.lambda_factor);
      lambda_ = std::max(lambda_ * dlambda_, config_.lambda_min);
      if(lambda_ > config_.lambda_max)
      {
        return -1;
      }
      tr.n_backward++;
    }

    // small-gradient termination, evaluated before the line search    :217-231
    Real k_rel_norm = 0;
    for(int i = 0; i < T; i++)
    {
      int m = m_list_[i];
      Real kn = 0, un = 0;
      for(int a = 0; a < m; a++)
      {
        kn += k_list_[i * MM() + a] * k_list_[i * MM() + a];
        un += control_.u[i * MM() + a] * control_.u[i * MM() + a];
      }
      k_rel_norm = std::max(k_rel_norm, std::sqrt(kn) / (std::sqrt(un) + Real(1)));
    }
    tr.k_rel_norm = k_rel_norm;
    if(k_rel_norm < config_.k_rel_norm_thre && lambda_ < config_.lambda_thre)
    {
      return 1;
    }

    // Step 3: backtracking line search    :234-274
    bool forward_pass_success = false;
    Real cost_update_actual = 0;
    Real alpha = 0;
    Real cost_update_expected = 0;
    Real cost_update_ratio = 0;
    for(size_t ai = 0; ai < config_.alpha_list.size(); ai++)
    {
      alpha = config_.alpha_list[ai];
      forwardPass(alpha);
      tr.n_forward++;
      tr.alpha_idx = static_cast<int>(ai);

      cost_update_actual = sum(control_.cost) - sum(cand_.cost);
      cost_update_expected = -1 * alpha * (dV_[0] + alpha * dV_[1]);
      cost_update_ratio = cost_update_actual / cost_update_expected;
      if(cost_update_expected < 0)
      {
        cost_update_ratio = (cost_update_actual >= 0 ? 1 : -1); // :251-259
      }
      if(cost_update_ratio > config_.cost_update_ratio_thre)
      {
        forward_pass_success = true;
        break;
      }
    }
    tr.alpha = alpha;
    tr.cost_update_actual = cost_update_actual;
    tr.cost_update_expected = cost_update_expected;
    tr.cost_update_ratio = cost_update_ratio;

    // Step 4: accept / reject and the lambda schedule    :280-337
    int retval = 0;
    if(forward_pass_success)
    {
      control_.x = cand_.x;
      control_.u = cand_.u;
      control_.cost = cand_.cost;
      if(cost_update_actual < config_.cost_update_thre)
      {
        retval = 1;
      }
      dlambda_ = std::min(dlambda_ / config_.lambda_factor, 1 / config_.lambda_factor);
      if(lambda_ >= config_.lambda_min)
      {
        lambda_ *= dlambda_;
      }
      else
      {
        lambda_ = 0;
      }
    }
    else
    {
      dlambda_ = std::max(dlambda_ * config_.lambda_factor, config_.lambda_factor);
      lambda_ = std::max(lambda_ * dlambda_, config_.lambda_min);
      if(lambda_ > config_.lambda_max)
      {
        retval = -1;
      }
    }
    tr.cost = sum(control_.cost);
    tr.lambda = lambda_;
    tr.dlambda = dlambda_;
    return retval;
  }

  /** DDPSolver::backwardPass (DDPSolver.hpp:342-534). */
  bool backwardPass()
  {
    const int T = config_.horizon_steps;
    Real Vx[N], Vxx[N * N], Vxx_reg[N * N];
    Real Qu[MM()] = {0}, Qx[N], Qux[MM() * N] = {0}, Quu[MM() * MM()] = {0}, Qxx[N * N], Qux_reg[MM() * N] = {0},
           Quu_F[MM() * MM()] = {0};
    constexpr int PMAX = (N > MM() ? N : MM()) * (N > MM() ? N : MM());
    Real FuT_V[MM() * N], FxT_V[N * N], prod[PMAX];
    Real k[MM()] = {0}, K[MM() * N] = {0};
    Real tmp_m[MM()];

    for(int j = 0; j < N; j++)
    {
      Vx[j] = last_Vx_[j];
    }
    for(int j = 0; j < N * N; j++)
    {
      Vxx[j] = last_Vxx_[j];
    }
    dV_[0] = 0;
    dV_[1] = 0;

    for(int i = T - 1; i >= 0; i--)
    {
      Real t = current_t_ + i * model_.dt;
      const Deriv & d = deriv_[i];
      const int m = d.m;

      // Q terms    :386-408
      // Qu = Lu + Fu^T Vx
      for(int a = 0; a < m; a++)
      {
        Real s = 0;
        for(int r = 0; r < N; r++)
        {
          s += d.Fu[r + a * N] * Vx[r];
        }
        Qu[a] = d.Lu[a] + s;
      }
      // Qx = Lx + Fx^T Vx
      for(int a = 0; a < N; a++)
      {
        Real s = 0;
        for(int r = 0; r < N; r++)
        {
          s += d.Fx[r + a * N] * Vx[r];
        }
        Qx[a] = d.Lx[a] + s;
      }
      // Qux = Lxu^T + (Fu^T Vxx) Fx
      mulAtB(d.Fu, Vxx, FuT_V, m, N, N);
      mulAB(FuT_V, d.Fx, prod, m, N, N);
      for(int c = 0; c < N; c++)
      {
        for(int a = 0; a < m; a++)
        {
          Qux[a + c * m] = d.Lxu[c + a * N] + prod[a + c * m];
        }
      }
      // Quu = Luu + (Fu^T Vxx) Fu
      mulAB(FuT_V, d.Fu, prod, m, N, m);
      for(int e = 0; e < m * m; e++)
      {
        Quu[e] = d.Luu[e] + prod[e];
      }
      // Qxx = Lxx + (Fx^T Vxx) Fx
      mulAtB(d.Fx, Vxx, FxT_V, N, N, N);
      mulAB(FxT_V, d.Fx, prod, N, N, N);
      for(int e = 0; e < N * N; e++)
      {
        Qxx[e] = d.Lxx[e] + prod[e];
      }

      // regularisation    :421-441
      for(int e = 0; e < N * N; e++)
      {
        Vxx_reg[e] = Vxx[e];
      }
      if(config_.reg_type == 2)
      {
        for(int j = 0; j < N; j++)
        {
          Vxx_reg[j + j * N] += lambda_;
        }
      }
      mulAtB(d.Fu, Vxx_reg, FuT_V, m, N, N);
      mulAB(FuT_V, d.Fx, prod, m, N, N);
      for(int c = 0; c < N; c++)
      {
        for(int a = 0; a < m; a++)
        {
          Qux_reg[a + c * m] = d.Lxu[c + a * N] + prod[a + c * m];
        }
      }
      mulAB(FuT_V, d.Fu, prod, m, N, m);
      for(int e = 0; e < m * m; e++)
      {
        Quu_F[e] = d.Luu[e] + prod[e];
      }
      if(config_.reg_type == 1)
      {
        for(int a = 0; a < m; a++)
        {
          Quu_F[a + a * m] += lambda_;
        }
      }

      // gains    :448-517
      if(m > 0)
      {
        if(config_.with_input_constraint)
        {
          // warm start from k[i+1] when the dimension matches    :452-467
          Real initial_k[MM()];
          for(int a = 0; a < m; a++)
          {
            initial_k[a] = 0;
          }
          if(i != T - 1 && m_list_[i + 1] == m)
          {
            for(int a = 0; a < m; a++)
            {
              initial_k[a] = k_list_[(i + 1) * MM() + a];
            }
          }
          Real lo[MM()], up[MM()];
          (void)t; // input_limits_func_(t): the constant pair, or row i of the table sampled at current_t + i dt    :470-472
          const size_t lim_at = limits_per_step_ ? static_cast<size_t>(i) * MMAX : 0;
          for(int a = 0; a < m; a++)
          {
            lo[a] = lower_.at(lim_at + a) - control_.u[i * MM() + a];
            up[a] = upper_.at(lim_at + a) - control_.u[i * MM() + a];
          }
          BoxQP qp;
          qp.solve(m, Quu_F, Qu, lo, up, initial_k);
          qp_retval_[i] = qp.retval;
          unsigned mask = 0;
          for(int idx : qp.free_idxs)
          {
            mask |= (1u << idx);
          }
          qp_free_mask_[i] = mask;
          if(qp.retval < 0)
          {
            return false; // :473-480
          }
          for(int a = 0; a < m; a++)
          {
            k[a] = qp.x[a];
          }
          // K: free rows = -llt_free.solve(Qux_reg[free, :]), clamped rows 0    :482-496
          for(int e = 0; e < m * N; e++)
          {
            K[e] = 0;
          }
          int nf = static_cast<int>(qp.free_idxs.size());
          if(nf > 0)
          {
            std::vector<Real> Kf(static_cast<size_t>(nf) * N);
            for(int c = 0; c < N; c++)
            {
              for(int j = 0; j < nf; j++)
              {
                Kf[j + c * nf] = Qux_reg[qp.free_idxs[j] + c * m];
              }
            }
            lltSolveInPlace(qp.llt_free.data(), nf, Kf.data(), N);
            for(int c = 0; c < N; c++)
            {
              for(int j = 0; j < nf; j++)
              {
                K[qp.free_idxs[j] + c * m] = -1 * Kf[j + c * nf];
              }
            }
          }
        }
        else
        {
          // LLT(Quu_F); k = -solve(Qu); K = -solve(Qux_reg)    :500-510
          Real L[MM() * MM()];
          for(int e = 0; e < m * m; e++)
          {
            L[e] = Quu_F[e];
          }
          if(lltInPlace(L, m) >= 0)
          {
            return false;
          }
          for(int a = 0; a < m; a++)
          {
            k[a] = Qu[a];
          }
          lltSolveInPlace(L, m, k, 1);
          for(int a = 0; a < m; a++)
          {
            k[a] = -1 * k[a];
          }
          for(int e = 0; e < m * N; e++)
          {
            K[e] = Qux_reg[e];
          }
          lltSolveInPlace(L, m, K, N);
          for(int e = 0; e < m * N; e++)
          {
            K[e] = -1 * K[e];
          }
        }
      }

      // value update with the UNregularised Quu, Qux    :522-526
      // dV += [k.Qu, 0.5 k.(Quu k)]
      {
        Real kQu = 0;
        for(int a = 0; a < m; a++)
        {
          kQu += k[a] * Qu[a];
        }
        Real kQuuk = 0;
        for(int a = 0; a < m; a++)
        {
          Real s = 0;
          for(int b = 0; b < m; b++)
          {
            s += Quu[a + b * m] * k[b];
          }
          tmp_m[a] = s;
        }
        for(int a = 0; a < m; a++)
        {
          kQuuk += k[a] * tmp_m[a];
        }
        dV_[0] += kQu;
        dV_[1] += Real(0.5) * kQuuk;
      }
      // K^T Quu  (N x m), shared by both updates
      Real KtQuu[N * MM()];
      mulAtB(K, Quu, KtQuu, N, m, m);
      // Vx = Qx + (K^T Quu) k + K^T Qu + Qux^T k
      for(int r = 0; r < N; r++)
      {
        Real s1 = 0, s2 = 0, s3 = 0;
        for(int a = 0; a < m; a++)
        {
          s1 += KtQuu[r + a * N] * k[a];
        }
        for(int a = 0; a < m; a++)
        {
          s2 += K[a + r * m] * Qu[a];
        }
        for(int a = 0; a < m; a++)
        {
          s3 += Qux[a + r * m] * k[a];
        }
        Vx[r] = ((Qx[r] + s1) + s2) + s3;
      }
      // Vxx = Qxx + (K^T Quu) K + K^T Qux + Qux^T K, then symmetrise
      for(int c = 0; c < N; c++)
      {
        for(int r = 0; r < N; r++)
        {
          Real s1 = 0, s2 = 0, s3 = 0;
          for(int a = 0; a < m; a++)
          {
            s1 += KtQuu[r + a * N] * K[a + c * m];
          }
          for(int a = 0; a < m; a++)
          {
            s2 += K[a + r * m] * Qux[a + c * m];
          }
          for(int a = 0; a < m; a++)
          {
            s3 += Qux[a + r * m] * K[a + c * m];
          }
          prod[r + c * N] = ((Qxx[r + c * N] + s1) + s2) + s3;
        }
      }
      for(int c = 0; c < N; c++)
      {
        for(int r = 0; r < N; r++)
        {
          Vxx[r + c * N] = Real(0.5) * (prod[r + c * N] + prod[c + r * N]);
        }
      }

      // save gains    :529-530
      for(int a = 0; a < m; a++)
      {
        k_list_[i * MM() + a] = k[a];
      }
      for(int a = m; a < MM(); a++)
      {
        k_list_[i * MM() + a] = 0;
      }
      // K stored per step as (MMAX x N) column-major with leading dimension m_i compacted to MMAX rows
      for(int c = 0; c < N; c++)
      {
        for(int a = 0; a < MM(); a++)
        {
          K_list_[(static_cast<size_t>(i) * N + c) * MM() + a] = (a < m) ? K[a + c * m] : Real(0);
        }
      }
    }
    return true;
  }

  /** DDPSolver::forwardPass (DDPSolver.hpp:536-560). */
  void forwardPass(Real alpha)
  {
    const int T = config_.horizon_steps;
    for(int j = 0; j < N; j++)
    {
      cand_.x[j] = control_.x[j];
    }
    for(int i = 0; i < T; i++)
    {
      const int m = m_list_[i];
      // u' = u + alpha k + K (x' - x)    :545-546
      Real dx[N];
      for(int j = 0; j < N; j++)
      {
        dx[j] = cand_.x[i * N + j] - control_.x[i * N + j];
      }
      for(int a = 0; a < m; a++)
      {
        Real s = 0;
        for(int c = 0; c < N; c++)
        {
          s += K_list_[(static_cast<size_t>(i) * N + c) * MM() + a] * dx[c];
        }
        cand_.u[i * MM() + a] = (control_.u[i * MM() + a] + alpha * k_list_[i * MM() + a]) + s;
      }
      Real t = current_t_ + i * model_.dt;
      model_.stateEq(t, &cand_.x[i * N], &cand_.u[i * MM()], m, &cand_.x[(i + 1) * N]);
      cand_.cost[i] = model_.runningCost(t, &cand_.x[i * N], &cand_.u[i * MM()], m);
    }
    Real terminal_t = current_t_ + T * model_.dt;
    cand_.cost[T] = model_.terminalCost(terminal_t, &cand_.x[T * N]);
  }

public:
  /** DDPSolver::dumpTraceDataList (DDPSolver.hpp:562-598): same 12 column names, durations written as 0. */
  void dumpTraceDataList(const std::string & file_path) const
  {
    FILE * fp = std::fopen(file_path.c_str(), "w");
    if(!fp)
    {
      throw std::runtime_error("cannot open " + file_path);
    }
    std::fprintf(fp, "iter cost lambda dlambda alpha k_rel_norm cost_update_actual cost_update_expected "
                     "cost_update_ratio duration_derivative duration_backward duration_forward\n");
    for(const TraceRow & r : trace_)
    {
      std::fprintf(fp, "%d %g %g %g %g %g %g %g %g 0 0 0\n", r.iter, r.cost, r.lambda, r.dlambda, r.alpha,
                   r.k_rel_norm, r.cost_update_actual, r.cost_update_expected, r.cost_update_ratio);
    }
    std::fclose(fp);
  }

protected:
  Model model_;
  Config config_;
  std::vector<TraceRow> trace_;
  std::vector<Real> lower_, upper_;
  bool limits_per_step_ = false;
  Real current_t_ = 0;
  Real lambda_ = 0;
  Real dlambda_ = 0;
  ControlData control_, cand_;
  std::vector<Real> k_list_, K_list_;
  std::vector<int> m_list_;
  std::vector<Deriv> deriv_;
  Real last_Vx_[N];
  Real last_Vxx_[N * N];
  Real dV_[2] = {0, 0};
  int status_ = 0;
  std::vector<int> qp_retval_;
  std::vector<unsigned> qp_free_mask_;
};
} // namespace ORACLE_NS
#undef ORACLE_NS
#undef ORACLE_REAL
