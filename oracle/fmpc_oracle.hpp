// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement ("oracle") of the reference's FMPC solver, used only by tests/ and by bench.py's
// cpu_baseline leg as the checker / reported baseline.  Nothing under nmpc_amd/ or include/ may include, link or call it.
//
// PARITY STATUS: the reference (header-only C++ on Eigen3) cannot be built in this image (Eigen3 is not installed and must
// not be stubbed), so this restatement is pinned by
//   (1) the reference's analytical-vs-numerical checks of l1NormDirectionalDeriv (nmpc_fmpc/tests/src/TestMathUtils.cpp:7-70),
//   (2) its finite-difference derivative checks (TestFmpcOscillator.cpp:207-266, TestFmpcCartPole.cpp:625-693),
//   (3) every EXPECT of its closed-loop MPC tests (TestFmpcOscillator.cpp:137-205: status Succeeded / MaxIterationReached
//       at every tick, g(x, u0) <= 0 at every tick, |x| < 1e-2 at the end; TestFmpcCartPole.cpp:362,377-380).
// Solver internals (k, K, s, P, kkt_error per iteration, step lengths) have NO golden vectors in the reference: for those,
// "parity unpinned" beyond (3).
//
// What is restated (citations relative to /root/reference/nmpc_fmpc/include/nmpc_fmpc/):
//   FmpcSolver::Configuration defaults      FmpcSolver.h:57-89
//   FmpcSolver::Status                      FmpcSolver.h:92-114
//   FmpcSolver::Variable::reset             FmpcSolver.hpp:42-69
//   FmpcSolver::solve                       FmpcSolver.hpp:156-255
//   FmpcSolver::checkVariable               FmpcSolver.hpp:285-354
//   FmpcSolver::procOnce                    FmpcSolver.hpp:356-491
//   FmpcSolver::calcKktError                FmpcSolver.hpp:493-520
//   FmpcSolver::backwardPass                FmpcSolver.hpp:522-665
//   FmpcSolver::forwardPass                 FmpcSolver.hpp:667-708
//   FmpcSolver::updateVariables             FmpcSolver.hpp:710-838
//   FmpcSolver::setupMeritFunc              FmpcSolver.hpp:840-936
//   FmpcSolver::calcMeritFunc               FmpcSolver.hpp:938-981
//   l1NormDirectionalDeriv                  MathUtils.h:17-38
// Third-party algorithm behind the path: Eigen::LDLT (Eigen3, version not pinned by the reference: CMakeLists.txt:28
// `find_package(Eigen3 REQUIRED)`), used for G at FmpcSolver.hpp:581-586.  Restated from its published algorithm (Eigen
// 3.4 Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked and LDLT::_solve_impl): symmetric diagonal pivoting on the largest
// |diagonal|, D's pseudo-inverse in the solve (|d| <= DBL_MIN gives 0).  info() != Success needs an exactly zero pivot
// followed by a non-zero one; that branch (FullPivLU fallback, :588-602) is restated as a full-pivot Gaussian elimination.
// Reductions are summed in ascending index order (Eigen's order is implementation-defined; tolerance + exact statuses /
// iteration counts is the parity target, as for the DDP oracle).
// Fixed dimensions only (the reference's Eigen::Dynamic InputDim / IneqDim, FmpcSolver.hpp:211-218, is not restated: neither
// of its tests uses it).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle_fmpc
{
/** l1NormDirectionalDeriv (MathUtils.h:17-38): jac is (out_dim x in_dim) column-major. */
inline double l1NormDirectionalDeriv(const double * func, const double * jac, const double * dir, int out_dim, int in_dim)
{
  double deriv = 0.0;
  for(int i = 0; i < out_dim; i++)
  {
    double d = 0;
    for(int j = 0; j < in_dim; j++)
    {
      d += jac[i + j * out_dim] * dir[j];
    }
    if(func[i] > 0)
    {
      deriv += d;
    }
    else if(func[i] < 0)
    {
      deriv += -1 * d;
    }
    else
    {
      deriv += std::abs(d);
    }
  }
  return deriv;
}

/** Eigen::LDLT<Matrix, Lower> of an n x n matrix (column-major, n <= 8): in-place factorisation with symmetric diagonal
    pivoting.  Returns false where Eigen's info() would be NumericalIssue. */
struct Ldlt
{
  int n = 0;
  double a[64]; // L (unit lower, strictly-lower part) and D (diagonal)
  int tr[8]; // transpositions: step k swapped k <-> tr[k]

  bool compute(const double * G, int n_)
  {
    n = n_;
    for(int i = 0; i < n * n; i++)
    {
      a[i] = G[i];
    }
    if(n <= 1)
    {
      if(n == 1)
      {
        tr[0] = 0;
      }
      return true;
    }
    bool found_zero_pivot = false;
    bool ret = true;
    auto A = [&](int i, int j) -> double & { return a[i + j * n]; };
    for(int k = 0; k < n; k++)
    {
      // biggest diagonal element of the remaining block
      int p = k;
      double big = std::abs(A(k, k));
      for(int i = k + 1; i < n; i++)
      {
        if(std::abs(A(i, i)) > big)
        {
          big = std::abs(A(i, i));
          p = i;
        }
      }
      tr[k] = p;
      if(p != k)
      {
        // symmetric swap of rows / columns k and p, lower triangle only
        for(int j = 0; j < k; j++)
        {
          std::swap(A(k, j), A(p, j));
        }
        for(int i = p + 1; i < n; i++)
        {
          std::swap(A(i, k), A(i, p));
        }
        std::swap(A(k, k), A(p, p));
        for(int i = k + 1; i < p; i++)
        {
          std::swap(A(i, k), A(p, i));
        }
      }
      const int rs = n - k - 1;
      if(k > 0)
      {
        double temp[8];
        double acc = 0;
        for(int j = 0; j < k; j++)
        {
          temp[j] = A(j, j) * A(k, j);
          acc += A(k, j) * temp[j];
        }
        A(k, k) -= acc;
        for(int i = k + 1; i < n; i++)
        {
          double s = 0;
          for(int j = 0; j < k; j++)
          {
            s += A(i, j) * temp[j];
          }
          A(i, k) -= s;
        }
      }
      const double akk = A(k, k);
      const bool pivot_is_valid = std::abs(akk) > 0.0;
      if(k == 0 && !pivot_is_valid)
      {
        // the entire diagonal is zero: success iff the strictly lower part is zero too
        for(int j = 0; j < n; j++)
        {
          tr[j] = j;
          for(int i = j + 1; i < n; i++)
          {
            ret = ret && (A(i, j) == 0.0);
          }
        }
        return ret;
      }
      if(rs > 0 && pivot_is_valid)
      {
        for(int i = k + 1; i < n; i++)
        {
          A(i, k) /= akk;
        }
      }
      else if(rs > 0)
      {
        for(int i = k + 1; i < n; i++)
        {
          ret = ret && (A(i, k) == 0.0);
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

  /** x = G^-1 b for c right-hand sides (b: n x c column-major, in place). */
  void solveInPlace(double * b, int c) const
  {
    auto A = [&](int i, int j) { return a[i + j * n]; };
    for(int col = 0; col < c; col++)
    {
      double * x = b + col * n;
      for(int k = 0; k < n; k++)
      {
        std::swap(x[k], x[tr[k]]);
      }
      for(int i = 0; i < n; i++)
      {
        for(int j = 0; j < i; j++)
        {
          x[i] -= A(i, j) * x[j];
        }
      }
      for(int i = 0; i < n; i++)
      {
        if(std::abs(A(i, i)) > DBL_MIN)
        {
          x[i] /= A(i, i);
        }
        else
        {
          x[i] = 0;
        }
      }
      for(int i = n - 1; i >= 0; i--)
      {
        for(int j = i + 1; j < n; j++)
        {
          x[i] -= A(j, i) * x[j];
        }
      }
      for(int k = n - 1; k >= 0; k--)
      {
        std::swap(x[k], x[tr[k]]);
      }
    }
  }
};

/** Full-pivot Gaussian elimination, the role of Eigen::FullPivLU at FmpcSolver.hpp:599-601 (reached only when LDLT reports
    NumericalIssue).  b: n x c, in place; rank-deficient directions get 0. */
inline void fullPivLuSolveInPlace(const double * G, int n, double * b, int c)
{
  double a[64];
  int colperm[8];
  for(int i = 0; i < n * n; i++)
  {
    a[i] = G[i];
  }
  for(int i = 0; i < n; i++)
  {
    colperm[i] = i;
  }
  auto A = [&](int i, int j) -> double & { return a[i + j * n]; };
  int rank = 0;
  for(int k = 0; k < n; k++)
  {
    int pr = k, pc = k;
    double big = 0;
    for(int j = k; j < n; j++)
    {
      for(int i = k; i < n; i++)
      {
        if(std::abs(A(i, j)) > big)
        {
          big = std::abs(A(i, j));
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
    for(int j = 0; j < n; j++)
    {
      std::swap(A(k, j), A(pr, j));
    }
    for(int col = 0; col < c; col++)
    {
      std::swap(b[k + col * n], b[pr + col * n]);
    }
    for(int i = 0; i < n; i++)
    {
      std::swap(A(i, k), A(i, pc));
    }
    std::swap(colperm[k], colperm[pc]);
    for(int i = k + 1; i < n; i++)
    {
      const double f = A(i, k) / A(k, k);
      for(int j = k + 1; j < n; j++)
      {
        A(i, j) -= f * A(k, j);
      }
      for(int col = 0; col < c; col++)
      {
        b[i + col * n] -= f * b[k + col * n];
      }
    }
  }
  for(int col = 0; col < c; col++)
  {
    double y[8];
    for(int i = n - 1; i >= 0; i--)
    {
      if(i >= rank)
      {
        y[i] = 0;
        continue;
      }
      double s = b[i + col * n];
      for(int j = i + 1; j < rank; j++)
      {
        s -= A(i, j) * y[j];
      }
      y[i] = s / A(i, i);
    }
    for(int i = 0; i < n; i++)
    {
      b[colperm[i] + col * n] = y[i];
    }
  }
}

/** FmpcSolver::Configuration (FmpcSolver.h:57-89). */
struct Config
{
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
};

/** FmpcSolver::Status (FmpcSolver.h:92-114). */
enum Status
{
  Uninitialized = 0,
  Succeeded = 1,
  ErrorInForward = 2,
  ErrorInBackward = 3,
  ErrorInUpdate = 4,
  MaxIterationReached = 5,
  IterationContinued = 6
};

/** FmpcSolver::Variable (FmpcSolver.h:117-158) on flat arrays: x [T+1][N], u [T][M], lambda [T+1][N], s [T][G], nu [T][G]. */
struct Variable
{
  int T = 0, N = 0, M = 0, G = 0;
  std::vector<double> x, u, lambda, s, nu;

  Variable() {}
  Variable(int T_, int N_, int M_, int G_) : T(T_), N(N_), M(M_), G(G_)
  {
    x.assign(static_cast<size_t>(T + 1) * N, 0.0);
    u.assign(static_cast<size_t>(T) * M, 0.0);
    lambda.assign(static_cast<size_t>(T + 1) * N, 0.0);
    s.assign(static_cast<size_t>(T) * G, 0.0);
    nu.assign(static_cast<size_t>(T) * G, 0.0);
  }

  /** Variable::reset (FmpcSolver.hpp:42-69). */
  void reset(double x_, double u_, double lambda_, double s_, double nu_)
  {
    std::fill(x.begin(), x.end(), x_);
    std::fill(u.begin(), u.end(), u_);
    std::fill(lambda.begin(), lambda.end(), lambda_);
    std::fill(s.begin(), s.end(), s_);
    std::fill(nu.begin(), nu.end(), nu_);
  }

  /** Variable::containsNaN (FmpcSolver.hpp:71-96). */
  bool containsNaN() const
  {
    for(const auto * v : {&x, &u, &lambda, &s, &nu})
    {
      for(double e : *v)
      {
        if(std::isnan(e) || std::isinf(e))
        {
          return true;
        }
      }
    }
    return false;
  }
};

struct TraceRow
{
  int iter = 0;
  double kkt_error = 0;
  // no golden counterpart in the reference's TraceData (its other fields are CPU timers): the discrete / scalar decisions
  // of the iteration, for the parity tests
  double barrier_eps = 0;
  double alpha_s_max = 0;
  double alpha_nu_max = 0;
  double alpha_s = 0;
};

/** nmpc_fmpc::FmpcSolver<N, M, G> on flat column-major arrays. */
template<class Model>
class FmpcSolver
{
public:
  static constexpr int N = Model::N;
  static constexpr int M = Model::M;
  static constexpr int G = Model::G;

  /** Coefficient (FmpcSolver.h:161-230); all matrices column-major. */
  struct Coefficient
  {
    double A[N * N], B[N * M], C[G * N], D[G * M];
    double Lx[N], Lu[M], Lxx[N * N], Luu[M * M], Lxu[N * M];
    double x_bar[N], g_bar[G], Lx_bar[N], Lu_bar[M];
    double k[M], K[M * N], s[N], P[N * N];
    bool terminal = false;

    /** Coefficient::containsNaN (FmpcSolver.hpp:136-154).  The terminal coefficient holds only Lx, Lxx, Lx_bar, s, P
        (FmpcSolver.hpp:126-134); its other members are empty matrices there. */
    bool containsNaN() const
    {
      auto bad = [](const double * p, int n) {
        for(int i = 0; i < n; i++)
        {
          if(std::isnan(p[i]) || std::isinf(p[i]))
          {
            return true;
          }
        }
        return false;
      };
      if(terminal)
      {
        return bad(Lx, N) || bad(Lxx, N * N) || bad(Lx_bar, N) || bad(s, N) || bad(P, N * N);
      }
      return bad(A, N * N) || bad(B, N * M) || bad(C, G * N) || bad(D, G * M) || bad(Lx, N) || bad(Lu, M) || bad(Lxx, N * N)
             || bad(Luu, M * M) || bad(Lxu, N * M) || bad(x_bar, N) || bad(g_bar, G) || bad(Lx_bar, N) || bad(Lu_bar, M)
             || bad(k, M) || bad(K, M * N) || bad(s, N) || bad(P, N * N);
    }
  };

  explicit FmpcSolver(const Model & problem) : problem_(problem) {}

  Config & config()
  {
    return config_;
  }
  const Variable & variable() const
  {
    return variable_;
  }
  const Variable & deltaVariable() const
  {
    return delta_variable_;
  }
  const std::vector<Coefficient> & coeffList() const
  {
    return coeff_list_;
  }
  const std::vector<TraceRow> & traceDataList() const
  {
    return trace_data_list_;
  }
  double & barrierEps()
  {
    return barrier_eps_;
  }
  Model & problem()
  {
    return problem_;
  }

  /** FmpcSolver::solve (FmpcSolver.hpp:156-255). */
  Status solve(double current_t, const double * current_x, const Variable & initial_variable)
  {
    const int T = config_.horizon_steps;
    current_t_ = current_t;
    std::copy(current_x, current_x + N, current_x_);
    variable_ = initial_variable;

    if(config_.init_complementary_variable) // :170-187
    {
      constexpr double initial_barrier_eps = 1e-4;
      constexpr double complementary_variable_margin_rate = 1e-2;
      constexpr double complementary_variable_min = 1e-2;
      barrier_eps_ = initial_barrier_eps;
      for(int i = 0; i < T; i++)
      {
        const double t = current_t_ + i * problem_.dt;
        double g[G > 0 ? G : 1];
        problem_.ineqConst(t, &variable_.x[i * N], &variable_.u[i * M], g);
        for(int j = 0; j < G; j++)
        {
          const double sj = (1.0 + complementary_variable_margin_rate) * std::max(-1 * g[j], complementary_variable_min);
          variable_.s[i * G + j] = sj;
          variable_.nu[i * G + j] =
              (1.0 + complementary_variable_margin_rate) * std::max(barrier_eps_ * (1.0 / sj), complementary_variable_min);
        }
      }
    }

    checkVariable(); // :190

    if(delta_variable_.T != T) // :193-197
    {
      delta_variable_ = Variable(T, N, M, G);
    }
    coeff_list_.resize(T + 1); // :211-218 (fixed dimensions: existing elements are preserved)
    for(int i = 0; i <= T; i++)
    {
      coeff_list_[i].terminal = (i == T);
    }
    trace_data_list_.clear(); // :225

    Status status = Uninitialized; // :233-246
    for(int iter = 1; iter <= config_.max_iter; iter++)
    {
      status = procOnce(iter);
      if(status != IterationContinued)
      {
        break;
      }
    }
    if(status == IterationContinued)
    {
      status = MaxIterationReached;
    }
    return status;
  }

  /** FmpcSolver::checkVariable (FmpcSolver.hpp:285-354). */
  void checkVariable() const
  {
    const int T = config_.horizon_steps;
    if(static_cast<int>(variable_.x.size()) != (T + 1) * N)
    {
      throw std::invalid_argument("[FMPC] x_list length should be " + std::to_string(T + 1) + ".");
    }
    if(static_cast<int>(variable_.u.size()) != T * M)
    {
      throw std::invalid_argument("[FMPC] u_list length should be " + std::to_string(T) + ".");
    }
    if(static_cast<int>(variable_.lambda.size()) != (T + 1) * N)
    {
      throw std::invalid_argument("[FMPC] lambda_list length should be " + std::to_string(T + 1) + ".");
    }
    if(static_cast<int>(variable_.s.size()) != T * G)
    {
      throw std::invalid_argument("[FMPC] s_list length should be " + std::to_string(T) + ".");
    }
    if(static_cast<int>(variable_.nu.size()) != T * G)
    {
      throw std::invalid_argument("[FMPC] nu_list length should be " + std::to_string(T) + ".");
    }
    for(int i = 0; i < T * G; i++)
    {
      if(variable_.s[i] < 0)
      {
        throw std::runtime_error("[FMPC] s_list[i] must be non-negative. i: " + std::to_string(i / G));
      }
      if(variable_.nu[i] < 0)
      {
        throw std::runtime_error("[FMPC] nu_list[i] must be non-negative. i: " + std::to_string(i / G));
      }
    }
  }

  /** FmpcSolver::procOnce (FmpcSolver.hpp:356-491). */
  Status procOnce(int iter)
  {
    const int T = config_.horizon_steps;
    trace_data_list_.emplace_back();
    trace_data_list_.back().iter = iter;

    if(config_.update_barrier_eps) // :370-392, (19.19) in Nocedal & Wright
    {
      double s_nu_ave = 0.0;
      int total_ineq_dim = 0;
      for(int i = 0; i < T; i++)
      {
        double dot = 0;
        for(int j = 0; j < G; j++)
        {
          dot += variable_.s[i * G + j] * variable_.nu[i * G + j];
        }
        s_nu_ave += dot;
        total_ineq_dim += G;
      }
      s_nu_ave /= total_ineq_dim;
      const double sigma = 0.5;
      constexpr double barrier_eps_min = 1e-8;
      constexpr double barrier_eps_max = 1e6;
      barrier_eps_ = std::clamp(sigma * s_nu_ave, barrier_eps_min, barrier_eps_max);
    }
    trace_data_list_.back().barrier_eps = barrier_eps_;

    // Step 1: coefficients of the linearised KKT condition (:394-441)
    {
      const double dt = problem_.dt;
      for(int i = 0; i < T; i++)
      {
        Coefficient & c = coeff_list_[i];
        const double t = current_t_ + i * dt;
        const double * x = &variable_.x[i * N];
        const double * next_x = &variable_.x[(i + 1) * N];
        const double * u = &variable_.u[i * M];
        const double * lambda = &variable_.lambda[i * N];
        const double * next_lambda = &variable_.lambda[(i + 1) * N];
        const double * s = &variable_.s[i * G];
        const double * nu = &variable_.nu[i * G];

        problem_.calcStateEqDeriv(t, x, u, c.A, c.B);
        problem_.calcIneqConstDeriv(t, x, u, c.C, c.D);
        problem_.calcRunningCostDeriv(t, x, u, c.Lx, c.Lu, c.Lxx, c.Luu, c.Lxu);

        double f[N], g[G > 0 ? G : 1];
        problem_.stateEq(t, x, u, f);
        problem_.ineqConst(t, x, u, g);
        for(int a = 0; a < N; a++)
        {
          c.x_bar[a] = f[a] - next_x[a]; // (2.23c)
        }
        for(int a = 0; a < G; a++)
        {
          c.g_bar[a] = g[a] + s[a]; // (2.23d)
        }
        for(int a = 0; a < N; a++) // (2.25b): -lambda + dt Lx + A^T next_lambda + C^T nu, summed left to right
        {
          double at = 0, ct = 0;
          for(int r = 0; r < N; r++)
          {
            at += c.A[r + a * N] * next_lambda[r];
          }
          for(int r = 0; r < G; r++)
          {
            ct += c.C[r + a * G] * nu[r];
          }
          c.Lx_bar[a] = ((-1 * lambda[a] + dt * c.Lx[a]) + at) + ct;
        }
        for(int a = 0; a < M; a++) // (2.25c)
        {
          double bt = 0, dtn = 0;
          for(int r = 0; r < N; r++)
          {
            bt += c.B[r + a * N] * next_lambda[r];
          }
          for(int r = 0; r < G; r++)
          {
            dtn += c.D[r + a * G] * nu[r];
          }
          c.Lu_bar[a] = (dt * c.Lu[a] + bt) + dtn;
        }
      }
      {
        Coefficient & c = coeff_list_[T];
        const double terminal_t = current_t_ + T * dt;
        problem_.calcTerminalCostDeriv(terminal_t, &variable_.x[T * N], c.Lx, c.Lxx);
        for(int a = 0; a < N; a++)
        {
          c.Lx_bar[a] = c.Lx[a] - variable_.lambda[T * N + a]; // (2.25a)
        }
      }
    }

    const double kkt_error = calcKktError(0.0); // :443-449
    trace_data_list_.back().kkt_error = kkt_error;
    if(kkt_error <= config_.kkt_error_thre)
    {
      return Succeeded;
    }
    if(!backwardPass()) // :451-463
    {
      return ErrorInBackward;
    }
    if(!forwardPass()) // :465-477
    {
      return ErrorInForward;
    }
    if(!updateVariables()) // :479-490
    {
      return ErrorInUpdate;
    }
    return IterationContinued;
  }

  /** FmpcSolver::calcKktError (FmpcSolver.hpp:493-520). */
  double calcKktError(double barrier_eps) const
  {
    const int T = config_.horizon_steps;
    double kkt_error = 0;
    for(int a = 0; a < N; a++)
    {
      const double e = current_x_[a] - variable_.x[a];
      kkt_error += e * e;
    }
    auto sq = [](const double * p, int n) {
      double s = 0;
      for(int i = 0; i < n; i++)
      {
        s += p[i] * p[i];
      }
      return s;
    };
    for(int i = 0; i < T; i++)
    {
      const Coefficient & c = coeff_list_[i];
      kkt_error += sq(c.x_bar, N);
      kkt_error += sq(c.g_bar, G);
      kkt_error += sq(c.Lx_bar, N);
      kkt_error += sq(c.Lu_bar, M);
      double comp = 0;
      for(int j = 0; j < G; j++)
      {
        const double e = std::max(variable_.s[i * G + j] * variable_.nu[i * G + j] - barrier_eps, 0.0);
        comp += e * e;
      }
      kkt_error += comp;
    }
    kkt_error += sq(coeff_list_[T].Lx_bar, N);
    return std::sqrt(kkt_error);
  }

  /** FmpcSolver::backwardPass (FmpcSolver.hpp:522-665). */
  bool backwardPass()
  {
    const int T = config_.horizon_steps;
    const double dt = problem_.dt;
    double s[N], P[N * N];
    {
      Coefficient & tc = coeff_list_[T];
      for(int a = 0; a < N; a++)
      {
        s[a] = -1 * tc.Lx_bar[a]; // (2.34)
      }
      std::copy(tc.Lxx, tc.Lxx + N * N, P); // (2.34)
      std::copy(s, s + N, tc.s);
      std::copy(P, P + N * N, tc.P);
    }
    for(int i = T - 1; i >= 0; i--)
    {
      Coefficient & c = coeff_list_[i];
      const double * sv = &variable_.s[i * G];
      const double * nuv = &variable_.nu[i * G];

      // pre-process (:562-580)
      double nu_s[G > 0 ? G : 1], tilde_sub[G > 0 ? G : 1];
      for(int j = 0; j < G; j++)
      {
        nu_s[j] = nuv[j] / sv[j];
        tilde_sub[j] = (nu_s[j] * c.g_bar[j] - nuv[j]) + barrier_eps_ * (1.0 / sv[j]);
      }
      double Qxx[N * N], Quu[M * M > 0 ? M * M : 1], Qxu[N * M > 0 ? N * M : 1], Lx_t[N], Lu_t[M > 0 ? M : 1];
      // C^T diag(nu_s) C evaluated left to right: (C^T diag) first, then times C
      for(int b = 0; b < N; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int j = 0; j < G; j++)
          {
            acc += (c.C[j + a * G] * nu_s[j]) * c.C[j + b * G];
          }
          Qxx[a + b * N] = dt * c.Lxx[a + b * N] + acc; // (2.28c)
        }
      }
      for(int b = 0; b < M; b++)
      {
        for(int a = 0; a < M; a++)
        {
          double acc = 0;
          for(int j = 0; j < G; j++)
          {
            acc += (c.D[j + a * G] * nu_s[j]) * c.D[j + b * G];
          }
          Quu[a + b * M] = dt * c.Luu[a + b * M] + acc; // (2.28e)
        }
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int j = 0; j < G; j++)
          {
            acc += (c.C[j + a * G] * nu_s[j]) * c.D[j + b * G];
          }
          Qxu[a + b * N] = dt * c.Lxu[a + b * N] + acc; // (2.28d)
        }
      }
      for(int a = 0; a < N; a++)
      {
        double acc = 0;
        for(int j = 0; j < G; j++)
        {
          acc += c.C[j + a * G] * tilde_sub[j];
        }
        Lx_t[a] = c.Lx_bar[a] + acc; // (2.28f)
      }
      for(int a = 0; a < M; a++)
      {
        double acc = 0;
        for(int j = 0; j < G; j++)
        {
          acc += c.D[j + a * G] * tilde_sub[j];
        }
        Lu_t[a] = c.Lu_bar[a] + acc; // (2.28g)
      }
      // A^T P (n x n), then times A / B; B^T P (m x n) times B
      double AtP[N * N], BtP[N * M > 0 ? N * M : 1];
      for(int b = 0; b < N; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += c.A[r + a * N] * P[r + b * N];
          }
          AtP[a + b * N] = acc;
        }
        for(int a = 0; a < M; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += c.B[r + a * N] * P[r + b * N];
          }
          BtP[a + b * M] = acc;
        }
      }
      double F[N * N], H[N * M > 0 ? N * M : 1], Gm[M * M > 0 ? M * M : 1];
      for(int b = 0; b < N; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += AtP[a + r * N] * c.A[r + b * N];
          }
          F[a + b * N] = Qxx[a + b * N] + acc; // (2.35b)
        }
      }
      for(int b = 0; b < M; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += AtP[a + r * N] * c.B[r + b * N];
          }
          H[a + b * N] = Qxu[a + b * N] + acc; // (2.35c)
        }
        for(int a = 0; a < M; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += BtP[a + r * M] * c.B[r + b * N];
          }
          Gm[a + b * M] = Quu[a + b * M] + acc; // (2.35d)
        }
      }

      // gains (:582-617)
      double Px_s[N]; // P x_bar - s
      for(int a = 0; a < N; a++)
      {
        double acc = 0;
        for(int r = 0; r < N; r++)
        {
          acc += P[a + r * N] * c.x_bar[r];
        }
        Px_s[a] = acc - s[a];
      }
      double k[M > 0 ? M : 1], K[M * N > 0 ? M * N : 1];
      if(M > 0)
      {
        for(int a = 0; a < M; a++)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += c.B[r + a * N] * Px_s[r];
          }
          k[a] = acc + Lu_t[a];
        }
        for(int b = 0; b < N; b++)
        {
          for(int a = 0; a < M; a++)
          {
            K[a + b * M] = H[b + a * N]; // H^T
          }
        }
        Ldlt ldlt;
        if(ldlt.compute(Gm, M))
        {
          ldlt.solveInPlace(k, 1);
          ldlt.solveInPlace(K, N);
        }
        else
        {
          if(config_.break_if_llt_fails)
          {
            return false;
          }
          fullPivLuSolveInPlace(Gm, M, k, 1);
          fullPivLuSolveInPlace(Gm, M, K, N);
        }
        for(int a = 0; a < M; a++)
        {
          k[a] = -1 * k[a]; // (2.35e)
        }
        for(int a = 0; a < M * N; a++)
        {
          K[a] = -1 * K[a]; // (2.35e)
        }
      }

      // post-process (:620-631)
      double s_new[N], P_new[N * N];
      for(int a = 0; a < N; a++)
      {
        double at = 0, hk = 0;
        for(int r = 0; r < N; r++)
        {
          at += c.A[r + a * N] * (-1 * Px_s[r]); // A^T (s - P x_bar)
        }
        for(int r = 0; r < M; r++)
        {
          hk += H[a + r * N] * k[r];
        }
        s_new[a] = (at - Lx_t[a]) - hk; // (2.35a)
      }
      // K^T G K left to right: (K^T G) (n x m), then times K
      double KtG[N * M > 0 ? N * M : 1];
      for(int b = 0; b < M; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int r = 0; r < M; r++)
          {
            acc += K[r + a * M] * Gm[r + b * M];
          }
          KtG[a + b * N] = acc;
        }
      }
      for(int b = 0; b < N; b++)
      {
        for(int a = 0; a < N; a++)
        {
          double acc = 0;
          for(int r = 0; r < M; r++)
          {
            acc += KtG[a + r * N] * K[r + b * M];
          }
          P_new[a + b * N] = F[a + b * N] - acc; // (2.35a)
        }
      }
      for(int b = 0; b < N; b++)
      {
        for(int a = 0; a < N; a++)
        {
          P[a + b * N] = 0.5 * (P_new[a + b * N] + P_new[b + a * N]); // enforce symmetric (:627-629)
        }
      }
      std::copy(s_new, s_new + N, s);

      std::copy(k, k + M, c.k); // :634-637
      std::copy(K, K + M * N, c.K);
      std::copy(s, s + N, c.s);
      std::copy(P, P + N * N, c.P);
    }

    if(config_.check_nan) // :640-653
    {
      for(const auto & c : coeff_list_)
      {
        if(c.containsNaN())
        {
          return false;
        }
      }
    }
    return true;
  }

  /** FmpcSolver::forwardPass (FmpcSolver.hpp:667-708). */
  bool forwardPass()
  {
    const int T = config_.horizon_steps;
    Variable & d = delta_variable_;
    for(int a = 0; a < N; a++)
    {
      d.x[a] = current_x_[a] - variable_.x[a];
    }
    for(int i = 0; i < T + 1; i++)
    {
      const Coefficient & c = coeff_list_[i];
      const double * dx = &d.x[i * N];
      for(int a = 0; a < N; a++) // (2.33)
      {
        double acc = 0;
        for(int r = 0; r < N; r++)
        {
          acc += c.P[a + r * N] * dx[r];
        }
        d.lambda[i * N + a] = acc - c.s[a];
      }
      if(i < T)
      {
        double * du = &d.u[i * M];
        for(int a = 0; a < M; a++) // (2.36)
        {
          double acc = 0;
          for(int r = 0; r < N; r++)
          {
            acc += c.K[a + r * M] * dx[r];
          }
          du[a] = acc + c.k[a];
        }
        for(int a = 0; a < N; a++) // (2.26b)
        {
          double ax = 0, bu = 0;
          for(int r = 0; r < N; r++)
          {
            ax += c.A[a + r * N] * dx[r];
          }
          for(int r = 0; r < M; r++)
          {
            bu += c.B[a + r * N] * du[r];
          }
          d.x[(i + 1) * N + a] = (ax + bu) + c.x_bar[a];
        }
      }
    }
    for(int i = 0; i < T; i++)
    {
      const Coefficient & c = coeff_list_[i];
      const double * dx = &d.x[i * N];
      const double * du = &d.u[i * M];
      for(int j = 0; j < G; j++)
      {
        double cx = 0, du_ = 0;
        for(int r = 0; r < N; r++)
        {
          cx += c.C[j + r * G] * dx[r];
        }
        for(int r = 0; r < M; r++)
        {
          du_ += c.D[j + r * G] * du[r];
        }
        const double ds = -1 * ((cx + du_) + c.g_bar[j]); // (2.27a)
        d.s[i * G + j] = ds;
        const double sv = variable_.s[i * G + j];
        const double nuv = variable_.nu[i * G + j];
        d.nu[i * G + j] = -1 * (nuv * (ds + sv) - barrier_eps_) / sv; // (2.27b)
      }
    }
    if(config_.check_nan && d.containsNaN())
    {
      return false;
    }
    return true;
  }

  /** FmpcSolver::updateVariables (FmpcSolver.hpp:710-838). */
  bool updateVariables()
  {
    const int T = config_.horizon_steps;
    const Variable & d = delta_variable_;
    double alpha_s_max = 1.0;
    double alpha_nu_max = 1.0;
    {
      constexpr double margin_ratio = 0.995;
      for(int i = 0; i < T * G; i++) // (19.9) in Nocedal & Wright
      {
        if(d.s[i] < 0)
        {
          alpha_s_max = std::min(alpha_s_max, -1 * margin_ratio * variable_.s[i] / d.s[i]);
        }
        if(d.nu[i] < 0)
        {
          alpha_nu_max = std::min(alpha_nu_max, -1 * margin_ratio * variable_.nu[i] / d.nu[i]);
        }
      }
      if(!(alpha_s_max > 0.0 && alpha_s_max <= 1.0 && alpha_nu_max > 0.0 && alpha_nu_max <= 1.0))
      {
        return false;
      }
    }
    double alpha_s = alpha_s_max;
    const double alpha_nu = alpha_nu_max;
    if(config_.enable_line_search) // :748-792
    {
      setupMeritFunc();
      constexpr double armijo_scale = 1e-3;
      constexpr double alpha_s_update_ratio = 0.5;
      constexpr double alpha_s_min = 1e-10;
      Variable ls = variable_;
      while(true)
      {
        if(alpha_s < alpha_s_min)
        {
          break;
        }
        for(size_t i = 0; i < ls.x.size(); i++)
        {
          ls.x[i] = variable_.x[i] + alpha_s * d.x[i];
        }
        for(size_t i = 0; i < ls.u.size(); i++)
        {
          ls.u[i] = variable_.u[i] + alpha_s * d.u[i];
        }
        for(size_t i = 0; i < ls.s.size(); i++)
        {
          ls.s[i] = variable_.s[i] + alpha_s * d.s[i];
        }
        const double merit_func_new = calcMeritFunc(ls);
        if(merit_func_new < merit_func_ + armijo_scale * alpha_s * merit_deriv_)
        {
          break;
        }
        alpha_s *= alpha_s_update_ratio;
      }
    }
    trace_data_list_.back().alpha_s_max = alpha_s_max;
    trace_data_list_.back().alpha_nu_max = alpha_nu_max;
    trace_data_list_.back().alpha_s = alpha_s;

    for(size_t i = 0; i < variable_.x.size(); i++) // :801-835
    {
      variable_.x[i] += alpha_s * d.x[i];
      variable_.lambda[i] += alpha_nu * d.lambda[i];
    }
    for(size_t i = 0; i < variable_.u.size(); i++)
    {
      variable_.u[i] += alpha_s * d.u[i];
    }
    // `min_positive_value` of the reference is numeric_limits<double>::lowest() (:812), i.e. -DBL_MAX: the clamp below it
    // never changes a finite value.  Restated as written.
    constexpr double min_positive_value = std::numeric_limits<double>::lowest();
    for(int i = 0; i < T; i++)
    {
      bool s_neg = false, nu_neg = false;
      for(int j = 0; j < G; j++)
      {
        variable_.s[i * G + j] += alpha_s * d.s[i * G + j];
        variable_.nu[i * G + j] += alpha_nu * d.nu[i * G + j];
        s_neg = s_neg || variable_.s[i * G + j] < 0;
        nu_neg = nu_neg || variable_.nu[i * G + j] < 0;
      }
      for(int j = 0; j < G; j++)
      {
        if(s_neg)
        {
          variable_.s[i * G + j] = std::max(variable_.s[i * G + j], min_positive_value);
        }
        if(nu_neg)
        {
          variable_.nu[i * G + j] = std::max(variable_.nu[i * G + j], min_positive_value);
        }
      }
    }
    return true;
  }

  /** FmpcSolver::setupMeritFunc (FmpcSolver.hpp:840-936). */
  void setupMeritFunc()
  {
    const int T = config_.horizon_steps;
    const double dt = problem_.dt;
    const Variable & d = delta_variable_;
    double merit_func_obj = 0.0, merit_func_const = 0.0, merit_deriv_obj = 0.0, merit_deriv_const = 0.0;
    double neg_I[N * N] = {}, I_g[G * G > 0 ? G * G : 1] = {};
    for(int a = 0; a < N; a++)
    {
      neg_I[a + a * N] = -1;
    }
    for(int a = 0; a < G; a++)
    {
      I_g[a + a * G] = 1;
    }
    auto l1 = [](const double * p, int n) {
      double s = 0;
      for(int i = 0; i < n; i++)
      {
        s += std::abs(p[i]);
      }
      return s;
    };
    auto dot = [](const double * p, const double * q, int n) {
      double s = 0;
      for(int i = 0; i < n; i++)
      {
        s += p[i] * q[i];
      }
      return s;
    };
    {
      double cf[N];
      for(int a = 0; a < N; a++)
      {
        cf[a] = current_x_[a] - variable_.x[a];
      }
      merit_func_const += l1(cf, N);
      merit_deriv_const += l1NormDirectionalDeriv(cf, neg_I, &d.x[0], N, N);
    }
    for(int i = 0; i < T; i++)
    {
      const double t = current_t_ + i * dt;
      const double * x = &variable_.x[i * N];
      const double * u = &variable_.u[i * M];
      const double * s = &variable_.s[i * G];
      const double * next_x = &variable_.x[(i + 1) * N];
      const double * dx = &d.x[i * N];
      const double * du = &d.u[i * M];
      const double * ds = &d.s[i * G];
      const double * dnx = &d.x[(i + 1) * N];
      const Coefficient & c = coeff_list_[i];

      merit_func_obj += problem_.runningCost(t, x, u) * dt;
      merit_deriv_obj += (dot(c.Lx, dx, N) + dot(c.Lu, du, M)) * dt;

      double logsum = 0, invdot = 0;
      for(int j = 0; j < G; j++)
      {
        logsum += std::log(s[j]);
        invdot += (1.0 / s[j]) * ds[j];
      }
      merit_func_obj += -1 * barrier_eps_ * logsum;
      merit_deriv_obj += -1 * barrier_eps_ * invdot;

      {
        double f[N], cf[N];
        problem_.stateEq(t, x, u, f);
        for(int a = 0; a < N; a++)
        {
          cf[a] = f[a] - next_x[a];
        }
        merit_func_const += l1(cf, N);
        merit_deriv_const += l1NormDirectionalDeriv(cf, c.A, dx, N, N);
        merit_deriv_const += l1NormDirectionalDeriv(cf, c.B, du, N, M);
        merit_deriv_const += l1NormDirectionalDeriv(cf, neg_I, dnx, N, N);
      }
      {
        double g[G > 0 ? G : 1], cf[G > 0 ? G : 1];
        problem_.ineqConst(t, x, u, g);
        for(int a = 0; a < G; a++)
        {
          cf[a] = g[a] + s[a];
        }
        merit_func_const += l1(cf, G);
        merit_deriv_const += l1NormDirectionalDeriv(cf, c.C, dx, G, N);
        merit_deriv_const += l1NormDirectionalDeriv(cf, c.D, du, G, M);
        merit_deriv_const += l1NormDirectionalDeriv(cf, I_g, ds, G, G);
      }
    }
    {
      const double terminal_t = current_t_ + T * dt;
      merit_func_obj += problem_.terminalCost(terminal_t, &variable_.x[T * N]);
      merit_deriv_obj += dot(coeff_list_[T].Lx, &d.x[T * N], N);
    }

    constexpr double merit_const_scale_min = 1e-3;
    if(config_.merit_const_scale_from_lagrange_multipliers) // (18.32) in Nocedal & Wright
    {
      merit_const_scale_ = merit_const_scale_min;
      for(double v : variable_.lambda)
      {
        merit_const_scale_ = std::max(merit_const_scale_, std::abs(v));
      }
      for(double v : variable_.nu)
      {
        merit_const_scale_ = std::max(merit_const_scale_, std::abs(v));
      }
    }
    else // (18.33)
    {
      constexpr double rho = 0.5;
      merit_const_scale_ = std::max(merit_deriv_obj / ((1.0 - rho) * merit_func_const), merit_const_scale_min);
    }
    merit_func_ = merit_func_obj + merit_const_scale_ * merit_func_const;
    merit_deriv_ = merit_deriv_obj + merit_const_scale_ * merit_deriv_const;
  }

  /** FmpcSolver::calcMeritFunc (FmpcSolver.hpp:938-981). */
  double calcMeritFunc(const Variable & v) const
  {
    const int T = config_.horizon_steps;
    const double dt = problem_.dt;
    double merit_func_obj = 0.0, merit_func_const = 0.0;
    for(int a = 0; a < N; a++)
    {
      merit_func_const += std::abs(current_x_[a] - v.x[a]);
    }
    for(int i = 0; i < T; i++)
    {
      const double t = current_t_ + i * dt;
      const double * x = &v.x[i * N];
      const double * u = &v.u[i * M];
      const double * s = &v.s[i * G];
      merit_func_obj += problem_.runningCost(t, x, u) * dt;
      double logsum = 0;
      for(int j = 0; j < G; j++)
      {
        logsum += std::log(s[j]);
      }
      merit_func_obj += -1 * barrier_eps_ * logsum;
      double f[N], g[G > 0 ? G : 1];
      problem_.stateEq(t, x, u, f);
      double c1 = 0;
      for(int a = 0; a < N; a++)
      {
        c1 += std::abs(f[a] - v.x[(i + 1) * N + a]);
      }
      merit_func_const += c1;
      problem_.ineqConst(t, x, u, g);
      double c2 = 0;
      for(int a = 0; a < G; a++)
      {
        c2 += std::abs(g[a] + s[a]);
      }
      merit_func_const += c2;
    }
    merit_func_obj += problem_.terminalCost(current_t_ + T * dt, &v.x[T * N]);
    return merit_func_obj + merit_const_scale_ * merit_func_const;
  }

  double meritFunc() const
  {
    return merit_func_;
  }
  double meritDeriv() const
  {
    return merit_deriv_;
  }
  double meritConstScale() const
  {
    return merit_const_scale_;
  }

protected:
  Model problem_;
  Config config_;
  double current_t_ = 0;
  double current_x_[N] = {};
  Variable variable_;
  Variable delta_variable_;
  std::vector<Coefficient> coeff_list_;
  std::vector<TraceRow> trace_data_list_;
  double barrier_eps_ = 1e-4; // FmpcSolver.h:414
  double merit_const_scale_ = 0.0;
  double merit_func_ = 0.0;
  double merit_deriv_ = 0.0;
};
} // namespace oracle_fmpc
