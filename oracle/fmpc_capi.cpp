// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/fmpc_oracle.hpp header).
//
// ctypes-loadable C entry points over the CPU FMPC oracle: single solve (with every intermediate the parity tests compare),
// threaded batch solve (also the cpu_baseline leg of bench.py --workload fmpc, kind "port"), model evaluation, the LDLT
// restatement and l1NormDirectionalDeriv.
#include "fmpc_models.hpp"
#include "fmpc_oracle.hpp"

#include <atomic>
#include <cstring>
#include <thread>

#include <pthread.h>
#include <sched.h>

namespace
{
using namespace oracle_fmpc;

template<class F>
int dispatch(const char * name, F && f)
{
  const std::string s(name);
  if(s == Oscillator::kName)
  {
    return f(Oscillator());
  }
  if(s == CartPole::kName)
  {
    return f(CartPole());
  }
  if(s == PointMass::kName)
  {
    return f(PointMass());
  }
  return -100;
}

/** Every model is a struct of doubles: the parameter blob is its memory image. */
template<class M>
M fromParams(const double * params)
{
  static_assert(sizeof(M) % sizeof(double) == 0, "model structs hold doubles only");
  M m;
  if(params)
  {
    std::memcpy(static_cast<void *>(&m), params, sizeof(M));
  }
  return m;
}
} // namespace

extern "C"
{
  struct oracle_fmpc_config
  {
    int horizon_steps;
    int max_iter;
    double kkt_error_thre;
    int check_nan;
    int init_complementary_variable;
    int update_barrier_eps;
    int break_if_llt_fails;
    int enable_line_search;
    int merit_const_scale_from_lagrange_multipliers;
  };

  enum
  {
    ORACLE_FMPC_NTRACE = 6 // iter, kkt_error, barrier_eps, alpha_s_max, alpha_nu_max, alpha_s
  };

  void oracle_fmpc_default_config(oracle_fmpc_config * c)
  {
    Config d;
    c->horizon_steps = d.horizon_steps;
    c->max_iter = d.max_iter;
    c->kkt_error_thre = d.kkt_error_thre;
    c->check_nan = d.check_nan;
    c->init_complementary_variable = d.init_complementary_variable;
    c->update_barrier_eps = d.update_barrier_eps;
    c->break_if_llt_fails = d.break_if_llt_fails;
    c->enable_line_search = d.enable_line_search;
    c->merit_const_scale_from_lagrange_multipliers = d.merit_const_scale_from_lagrange_multipliers;
  }

  int oracle_fmpc_model_info(const char * model, int * n, int * m, int * g, int * param_doubles)
  {
    return dispatch(model, [&](auto mdl) {
      using M = decltype(mdl);
      *n = M::N;
      *m = M::M;
      *g = M::G;
      *param_doubles = static_cast<int>(sizeof(M) / sizeof(double));
      return 0;
    });
  }

  int oracle_fmpc_default_params(const char * model, double * out)
  {
    return dispatch(model, [&](auto mdl) {
      std::memcpy(out, &mdl, sizeof(mdl));
      return 0;
    });
  }
}

namespace
{
Config toConfig(const oracle_fmpc_config * c)
{
  Config cfg;
  cfg.print_level = 0;
  cfg.horizon_steps = c->horizon_steps;
  cfg.max_iter = c->max_iter;
  cfg.kkt_error_thre = c->kkt_error_thre;
  cfg.check_nan = c->check_nan != 0;
  cfg.init_complementary_variable = c->init_complementary_variable != 0;
  cfg.update_barrier_eps = c->update_barrier_eps != 0;
  cfg.break_if_llt_fails = c->break_if_llt_fails != 0;
  cfg.enable_line_search = c->enable_line_search != 0;
  cfg.merit_const_scale_from_lagrange_multipliers = c->merit_const_scale_from_lagrange_multipliers != 0;
  return cfg;
}

/** One solve on flat arrays.  Returns the Status, or -1 (std::invalid_argument) / -2 (std::runtime_error) where the
    reference throws (FmpcSolver.hpp:285-354). */
template<class M>
int solveOne(const M & mdl,
             const Config & cfg,
             double current_t,
             const double * current_x,
             double * x,
             double * u,
             double * lambda,
             double * s,
             double * nu,
             double * barrier_eps,
             int * iters,
             double * trace,
             double * gain_k,
             double * gain_K,
             double * gain_s,
             double * gain_P,
             double * delta)
{
  const int T = cfg.horizon_steps;
  constexpr int N = M::N, MM = M::M, G = M::G;
  FmpcSolver<M> solver(mdl);
  solver.config() = cfg;
  solver.barrierEps() = *barrier_eps;
  Variable v(T, N, MM, G);
  std::copy(x, x + (T + 1) * N, v.x.begin());
  std::copy(u, u + T * MM, v.u.begin());
  std::copy(lambda, lambda + (T + 1) * N, v.lambda.begin());
  std::copy(s, s + T * G, v.s.begin());
  std::copy(nu, nu + T * G, v.nu.begin());
  int status;
  try
  {
    status = solver.solve(current_t, current_x, v);
  }
  catch(const std::invalid_argument &)
  {
    return -1;
  }
  catch(const std::runtime_error &)
  {
    return -2;
  }
  const Variable & r = solver.variable();
  std::copy(r.x.begin(), r.x.end(), x);
  std::copy(r.u.begin(), r.u.end(), u);
  std::copy(r.lambda.begin(), r.lambda.end(), lambda);
  std::copy(r.s.begin(), r.s.end(), s);
  std::copy(r.nu.begin(), r.nu.end(), nu);
  *barrier_eps = solver.barrierEps();
  const auto & tr = solver.traceDataList();
  if(iters)
  {
    *iters = tr.empty() ? 0 : tr.back().iter;
  }
  if(trace)
  {
    for(int i = 0; i < cfg.max_iter; i++)
    {
      double * row = trace + i * ORACLE_FMPC_NTRACE;
      if(i < static_cast<int>(tr.size()))
      {
        row[0] = tr[i].iter;
        row[1] = tr[i].kkt_error;
        row[2] = tr[i].barrier_eps;
        row[3] = tr[i].alpha_s_max;
        row[4] = tr[i].alpha_nu_max;
        row[5] = tr[i].alpha_s;
      }
      else
      {
        std::fill(row, row + ORACLE_FMPC_NTRACE, 0.0);
      }
    }
  }
  const auto & cl = solver.coeffList();
  for(int i = 0; i <= T; i++)
  {
    if(i < T && gain_k)
    {
      std::copy(cl[i].k, cl[i].k + MM, gain_k + i * MM);
    }
    if(i < T && gain_K)
    {
      std::copy(cl[i].K, cl[i].K + MM * N, gain_K + i * MM * N);
    }
    if(gain_s)
    {
      std::copy(cl[i].s, cl[i].s + N, gain_s + i * N);
    }
    if(gain_P)
    {
      std::copy(cl[i].P, cl[i].P + N * N, gain_P + i * N * N);
    }
  }
  if(delta) // [dx (T+1)N | du TM | dlambda (T+1)N | ds TG | dnu TG] of the last iteration that reached the forward pass
  {
    const Variable & d = solver.deltaVariable();
    double * p = delta;
    for(const auto * vec : {&d.x, &d.u, &d.lambda, &d.s, &d.nu})
    {
      p = std::copy(vec->begin(), vec->end(), p);
    }
  }
  return status;
}
} // namespace

extern "C"
{
  /** FmpcSolver::solve for one instance; variables in / out.  Optional outputs may be NULL. */
  int oracle_fmpc_solve(const char * model,
                        const oracle_fmpc_config * c,
                        const double * params,
                        double current_t,
                        const double * current_x,
                        double * x,
                        double * u,
                        double * lambda,
                        double * s,
                        double * nu,
                        double * barrier_eps,
                        int * iters,
                        double * trace,
                        double * gain_k,
                        double * gain_K,
                        double * gain_s,
                        double * gain_P,
                        double * delta)
  {
    return dispatch(model, [&](auto mdl) {
      using M = decltype(mdl);
      return solveOne(fromParams<M>(params), toConfig(c), current_t, current_x, x, u, lambda, s, nu, barrier_eps, iters, trace,
                      gain_k, gain_K, gain_s, gain_P, delta);
    });
  }

  /** A batch of independent solves on n_threads CPU threads (dynamic chunks, threads pinned to the allowed CPUs in order).
      params: one blob (per_instance_params = 0) or one per instance.  Arrays are batch-major: x [B][T+1][N], ...;
      trace [B][max_iter][ORACLE_FMPC_NTRACE]; gain_K0 [B][M*N] (K of the first step, the caller's feedback gain,
      TestFmpcCartPole.cpp:350) may be NULL. */
  int oracle_fmpc_solve_batch(const char * model,
                              const oracle_fmpc_config * c,
                              const double * params,
                              int per_instance_params,
                              int batch,
                              const double * current_t,
                              const double * current_x,
                              double * x,
                              double * u,
                              double * lambda,
                              double * s,
                              double * nu,
                              double * barrier_eps,
                              int * status,
                              int * iters,
                              double * trace,
                              double * gain_K0,
                              int n_threads)
  {
    return dispatch(model, [&](auto mdl0) {
      using M = decltype(mdl0);
      constexpr int N = M::N, MM = M::M, G = M::G;
      const Config cfg = toConfig(c);
      const int T = cfg.horizon_steps;
      const int pd = static_cast<int>(sizeof(M) / sizeof(double));
      std::atomic<int> next(0);
      const int chunk = std::max(1, std::min(16, batch / std::max(1, n_threads * 8)));
      cpu_set_t allowed;
      CPU_ZERO(&allowed);
      sched_getaffinity(0, sizeof(allowed), &allowed);
      std::vector<int> cpus;
      for(int i = 0; i < CPU_SETSIZE; i++)
      {
        if(CPU_ISSET(i, &allowed))
        {
          cpus.push_back(i);
        }
      }
      auto work = [&](int tid) {
        if(n_threads > 1 && !cpus.empty())
        {
          cpu_set_t one;
          CPU_ZERO(&one);
          CPU_SET(cpus[tid % cpus.size()], &one);
          pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
        }
        std::vector<double> K(static_cast<size_t>(T) * MM * N);
        for(;;)
        {
          const int b0 = next.fetch_add(chunk);
          if(b0 >= batch)
          {
            break;
          }
          for(int b = b0; b < std::min(batch, b0 + chunk); b++)
          {
            const M mdl = fromParams<M>(params ? params + (per_instance_params ? static_cast<size_t>(b) * pd : 0) : nullptr);
            status[b] = solveOne(mdl, cfg, current_t ? current_t[b] : 0.0, current_x + static_cast<size_t>(b) * N,
                                 x + static_cast<size_t>(b) * (T + 1) * N, u + static_cast<size_t>(b) * T * MM,
                                 lambda + static_cast<size_t>(b) * (T + 1) * N, s + static_cast<size_t>(b) * T * G,
                                 nu + static_cast<size_t>(b) * T * G, barrier_eps + b, iters ? iters + b : nullptr,
                                 trace ? trace + static_cast<size_t>(b) * cfg.max_iter * ORACLE_FMPC_NTRACE : nullptr, nullptr,
                                 gain_K0 ? K.data() : nullptr, nullptr, nullptr, nullptr);
            if(gain_K0)
            {
              std::copy(K.begin(), K.begin() + MM * N, gain_K0 + static_cast<size_t>(b) * MM * N);
            }
          }
        }
      };
      if(n_threads <= 1)
      {
        work(0);
      }
      else
      {
        std::vector<std::thread> th;
        for(int i = 0; i < n_threads; i++)
        {
          th.emplace_back(work, i);
        }
        for(auto & t : th)
        {
          t.join();
        }
      }
      return 0;
    });
  }

  /** Model evaluation for the derivative checks and the plant step of the closed-loop tests.  Every output may be NULL.
      step_dt > 0: f = stateEq(t, x, u, step_dt) (TestFmpcOscillator.cpp:27, TestFmpcCartPole.cpp:73), else stateEq(t, x, u). */
  int oracle_fmpc_eval(const char * model,
                       const double * params,
                       double t,
                       const double * x,
                       const double * u,
                       double step_dt,
                       double * f,
                       double * g,
                       double * costs, // [running, terminal]
                       double * A,
                       double * B,
                       double * C,
                       double * D,
                       double * Lx,
                       double * Lu,
                       double * Lxx,
                       double * Luu,
                       double * Lxu,
                       double * Vx,
                       double * Vxx)
  {
    return dispatch(model, [&](auto mdl0) {
      using M = decltype(mdl0);
      const M mdl = fromParams<M>(params);
      if(f)
      {
        if(step_dt > 0)
        {
          mdl.stateEqDt(t, x, u, step_dt, f);
        }
        else
        {
          mdl.stateEq(t, x, u, f);
        }
      }
      if(g)
      {
        mdl.ineqConst(t, x, u, g);
      }
      if(costs)
      {
        costs[0] = mdl.runningCost(t, x, u);
        costs[1] = mdl.terminalCost(t, x);
      }
      if(A && B)
      {
        mdl.calcStateEqDeriv(t, x, u, A, B);
      }
      if(C && D)
      {
        mdl.calcIneqConstDeriv(t, x, u, C, D);
      }
      if(Lx && Lu && Lxx && Luu && Lxu)
      {
        mdl.calcRunningCostDeriv(t, x, u, Lx, Lu, Lxx, Luu, Lxu);
      }
      if(Vx && Vxx)
      {
        mdl.calcTerminalCostDeriv(t, x, Vx, Vxx);
      }
      return 0;
    });
  }

  double oracle_fmpc_l1_dir_deriv(const double * func, const double * jac, const double * dir, int out_dim, int in_dim)
  {
    return l1NormDirectionalDeriv(func, jac, dir, out_dim, in_dim);
  }

  /** x = G^-1 b through the LDLT restatement (b: n x c column-major, in place).  Returns 1 if info() == Success. */
  int oracle_fmpc_ldlt_solve(const double * G, int n, double * b, int c, int use_lu)
  {
    if(n > 8)
    {
      return -1;
    }
    if(use_lu)
    {
      fullPivLuSolveInPlace(G, n, b, c);
      return 1;
    }
    Ldlt l;
    const bool ok = l.compute(G, n);
    l.solveInPlace(b, c);
    return ok ? 1 : 0;
  }
}
