// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/ddp_oracle.hpp header).
//
// ctypes-loadable C entry points over the CPU oracle: single solve, threaded batch solve (also the
// cpu_baseline leg of bench.py, kind "port"), BoxQP, model evaluation, and ROS-free restatements of the
// reference's closed-loop MPC test loops.
#include "ddp_oracle.hpp"
#include "models.hpp"
#include "models_builder.hpp"
// the same statements in single precision: namespace oracle_f32 (BASELINE.json config 4 is specified in fp32)
#define ORACLE_F32
#include "ddp_oracle.hpp"
#include "model_cartpole.hpp"
#include "models_builder.hpp"
#undef ORACLE_F32

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <type_traits>

#include <pthread.h>
#include <sched.h>

namespace
{
using namespace oracle;

template<class F>
int dispatch(const char * name, F && f)
{
  std::string s(name);
  if(s == "cartpole")
  {
    return f(CartPole());
  }
  if(s == "bipedal")
  {
    return f(Bipedal());
  }
  if(s == "vertical")
  {
    return f(VerticalMotion());
  }
  if(s == "centroidal")
  {
    return f(CentroidalMotion());
  }
  if(s == "quadrotor")
  {
    return f(Quadrotor());
  }
  if(s == "manipulator")
  {
    return f(Manipulator());
  }
  if(s == "planar_vtol")
  {
    return f(PlanarVtol());
  }
  if(s == "cartpole_f32")
  {
    return f(oracle_f32::CartPole());
  }
  if(s == "quadrotor_f32")
  {
    return f(oracle_f32::Quadrotor());
  }
  if(s == "manipulator_f32")
  {
    return f(oracle_f32::Manipulator());
  }
  return -100;
}

/** The solver instantiation that matches the model's arithmetic type. */
template<class M>
using SolverOf = std::conditional_t<std::is_same<typename M::Real, float>::value, oracle_f32::DDPSolver<M>, oracle::DDPSolver<M>>;

/** The C entry points exchange doubles whatever the model computes in: inputs are rounded to the model's Real once,
    outputs widened (exact). */
template<class R>
std::vector<R> toReal(const double * p, size_t n)
{
  std::vector<R> v(n);
  for(size_t i = 0; i < n; i++)
  {
    v[i] = static_cast<R>(p[i]);
  }
  return v;
}
template<class R>
void toDouble(double * dst, const R * src, size_t n)
{
  for(size_t i = 0; i < n; i++)
  {
    dst[i] = static_cast<double>(src[i]);
  }
}
template<class R>
void toDouble(double * dst, const std::vector<R> & v)
{
  toDouble(dst, v.data(), v.size());
}
} // namespace

extern "C"
{
  struct oracle_config
  {
    int with_input_constraint;
    int max_iter;
    int horizon_steps;
    int reg_type;
    double initial_lambda;
    double initial_dlambda;
    double lambda_factor;
    double lambda_min;
    double lambda_max;
    double k_rel_norm_thre;
    double lambda_thre;
    double cost_update_ratio_thre;
    double cost_update_thre;
    int n_alpha;
    double alpha_list[32];
  };

  enum
  {
    ORACLE_NTRACE = 12
  };

  /** Fill cfg with the reference defaults (DDPSolver.h:47-110). */
  void oracle_default_config(oracle_config * c)
  {
    Config d;
    c->with_input_constraint = d.with_input_constraint;
    c->max_iter = d.max_iter;
    c->horizon_steps = d.horizon_steps;
    c->reg_type = d.reg_type;
    c->initial_lambda = d.initial_lambda;
    c->initial_dlambda = d.initial_dlambda;
    c->lambda_factor = d.lambda_factor;
    c->lambda_min = d.lambda_min;
    c->lambda_max = d.lambda_max;
    c->k_rel_norm_thre = d.k_rel_norm_thre;
    c->lambda_thre = d.lambda_thre;
    c->cost_update_ratio_thre = d.cost_update_ratio_thre;
    c->cost_update_thre = d.cost_update_thre;
    c->n_alpha = static_cast<int>(d.alpha_list.size());
    for(int i = 0; i < 32; i++)
    {
      c->alpha_list[i] = i < c->n_alpha ? d.alpha_list[i] : 0.0;
    }
  }

} // extern "C"

namespace
{
  template<class Cfg>
  void toConfig(const oracle_config * c, Cfg & d)
  {
    d.with_input_constraint = c->with_input_constraint != 0;
    d.max_iter = c->max_iter;
    d.horizon_steps = c->horizon_steps;
    d.reg_type = c->reg_type;
    d.initial_lambda = c->initial_lambda;
    d.initial_dlambda = c->initial_dlambda;
    d.lambda_factor = c->lambda_factor;
    d.lambda_min = c->lambda_min;
    d.lambda_max = c->lambda_max;
    d.k_rel_norm_thre = c->k_rel_norm_thre;
    d.lambda_thre = c->lambda_thre;
    d.cost_update_ratio_thre = c->cost_update_ratio_thre;
    d.cost_update_thre = c->cost_update_thre;
    d.alpha_list.resize(c->n_alpha);
    for(int i = 0; i < c->n_alpha; i++)
    {
      d.alpha_list[i] = static_cast<typename std::remove_reference<decltype(d.alpha_list[0])>::type>(c->alpha_list[i]);
    }
  }
} // namespace

extern "C"
{

  int oracle_model_dims(const char * model, int * n, int * mmax, int * nparam)
  {
    return dispatch(model,
                    [&](auto m)
                    {
                      using M = decltype(m);
                      *n = M::N;
                      *mmax = M::MMAX;
                      *nparam = M::NPARAM;
                      return 0;
                    });
  }

  /** Per-step input dimension for a horizon starting at t0. */
  int oracle_input_dims(const char * model, const double * params, double t0, int T, int * m_list)
  {
    return dispatch(model,
                    [&](auto m)
                    {
                      if(params)
                      {
                        m.setParams(params);
                      }
                      for(int i = 0; i < T; i++)
                      {
                        m_list[i] = m.inputDim(t0 + i * m.dt);
                      }
                      return 0;
                    });
  }

  /** Evaluate every DDPProblem method once.  Any output pointer may be NULL. */
  int oracle_model_eval(const char * model,
                        const double * params,
                        double t,
                        const double * x,
                        const double * u,
                        double * xn,
                        double * running_cost,
                        double * terminal_cost,
                        double * Fx,
                        double * Fu,
                        double * Lx,
                        double * Lu,
                        double * Lxx,
                        double * Luu,
                        double * Lxu,
                        double * Vx,
                        double * Vxx,
                        int * m_out)
  {
    return dispatch(model,
                    [&](auto m)
                    {
                      using M = decltype(m);
                      if(params)
                      {
                        m.setParams(params);
                      }
                      using R = typename M::Real;
                      constexpr int N = M::N;
                      constexpr int MM = M::MMAX > 0 ? M::MMAX : 1;
                      const R tr = static_cast<R>(t);
                      int mi = m.inputDim(tr);
                      if(m_out)
                      {
                        *m_out = mi;
                      }
                      const std::vector<R> xr = toReal<R>(x, N);
                      const std::vector<R> ur = u ? toReal<R>(u, MM) : std::vector<R>(MM, R(0));
                      if(xn)
                      {
                        R o[N];
                        m.stateEq(tr, xr.data(), ur.data(), mi, o);
                        toDouble(xn, o, N);
                      }
                      if(running_cost)
                      {
                        *running_cost = m.runningCost(tr, xr.data(), ur.data(), mi);
                      }
                      if(terminal_cost)
                      {
                        *terminal_cost = m.terminalCost(tr, xr.data());
                      }
                      if(Fx && Fu)
                      {
                        R fx[N * N], fu[N * MM];
                        m.calcStateEqDeriv(tr, xr.data(), ur.data(), mi, fx, fu);
                        toDouble(Fx, fx, N * N);
                        toDouble(Fu, fu, static_cast<size_t>(N) * mi);
                      }
                      if(Lx && Lu && Lxx && Luu && Lxu)
                      {
                        R lx[N], lu[MM], lxx[N * N], luu[MM * MM], lxu[N * MM];
                        m.calcRunningCostDeriv(tr, xr.data(), ur.data(), mi, lx, lu, lxx, luu, lxu);
                        toDouble(Lx, lx, N);
                        toDouble(Lu, lu, mi);
                        toDouble(Lxx, lxx, N * N);
                        toDouble(Luu, luu, static_cast<size_t>(mi) * mi);
                        toDouble(Lxu, lxu, static_cast<size_t>(N) * mi);
                      }
                      if(Vx && Vxx)
                      {
                        R vx[N], vxx[N * N];
                        m.calcTerminalCostDeriv(tr, xr.data(), vx, vxx);
                        toDouble(Vx, vx, N);
                        toDouble(Vxx, vxx, N * N);
                      }
                      (void)sizeof(M);
                      return 0;
                    });
  }

  /** BoxQP::solve (BoxQP.h:141-347).  H column-major m x m.  free_idxs has room for m ints. */
  int oracle_boxqp_solve(int m,
                         const double * H,
                         const double * g,
                         const double * lower,
                         const double * upper,
                         const double * initial_x,
                         double * x_out,
                         int * retval,
                         int * free_idxs,
                         int * n_free,
                         int * iter,
                         int * factorization_num)
  {
    BoxQP qp;
    qp.solve(m, H, g, lower, upper, initial_x);
    for(int i = 0; i < m; i++)
    {
      x_out[i] = qp.x[i];
    }
    *retval = qp.retval;
    *n_free = static_cast<int>(qp.free_idxs.size());
    for(int i = 0; i < *n_free; i++)
    {
      free_idxs[i] = qp.free_idxs[i];
    }
    if(iter)
    {
      *iter = qp.iter;
    }
    if(factorization_num)
    {
      *factorization_num = qp.factorization_num;
    }
    return 0;
  }

  /** One DDPSolver::solve.  Layouts (MM = max(MMAX,1)):
        x0[N], u_init[T][MM], X[T+1][N], U[T][MM], cost[T+1], k[T][MM], K[T][N][MM] (per step column-major
        MM x N), trace[(max_iter+1)][12] = {iter,cost,lambda,dlambda,alpha,k_rel_norm,cost_update_actual,
        cost_update_expected,cost_update_ratio,alpha_idx,n_backward,n_forward}.
      Any output pointer may be NULL.  Returns 0, or <0 on misuse. */
  int oracle_ddp_solve(const char * model,
                       const double * params,
                       const oracle_config * cfg,
                       double t0,
                       const double * x0,
                       const double * u_init,
                       const double * lower,
                       const double * upper,
                       double * X,
                       double * U,
                       double * cost,
                       double * k,
                       double * K,
                       double * trace,
                       int * n_trace,
                       int * status,
                       double * dV,
                       int * qp_retval,
                       unsigned * qp_free_mask,
                       int limits_per_step /* 1: lower / upper are [T][MM] tables (time-varying limits), 0: [MM] */)
  {
    return dispatch(model,
                    [&](auto m)
                    {
                      using M = decltype(m);
                      if(params)
                      {
                        m.setParams(params);
                      }
                      using R = typename M::Real;
                      constexpr int N = M::N;
                      constexpr int MM = M::MMAX > 0 ? M::MMAX : 1;
                      SolverOf<M> solver(m);
                      toConfig(cfg, solver.config());
                      if(cfg->with_input_constraint)
                      {
                        if(!lower || !upper)
                        {
                          return -2;
                        }
                        if(limits_per_step)
                        {
                          const size_t nl = static_cast<size_t>(cfg->horizon_steps) * MM;
                          solver.setInputLimitsPerStep(toReal<R>(lower, nl).data(), toReal<R>(upper, nl).data(), cfg->horizon_steps);
                        }
                        else
                        {
                          solver.setInputLimits(toReal<R>(lower, MM).data(), toReal<R>(upper, MM).data());
                        }
                      }
                      solver.solve(static_cast<R>(t0), toReal<R>(x0, N).data(),
                                   toReal<R>(u_init, static_cast<size_t>(cfg->horizon_steps) * MM).data());
                      const auto & cd = solver.controlData();
                      if(X)
                      {
                        toDouble(X, cd.x);
                      }
                      if(U)
                      {
                        toDouble(U, cd.u);
                      }
                      if(cost)
                      {
                        toDouble(cost, cd.cost);
                      }
                      if(k)
                      {
                        toDouble(k, solver.kList());
                      }
                      if(K)
                      {
                        toDouble(K, solver.KList());
                      }
                      const auto & tr = solver.traceDataList();
                      if(n_trace)
                      {
                        *n_trace = static_cast<int>(tr.size());
                      }
                      if(trace)
                      {
                        for(size_t r = 0; r < tr.size(); r++)
                        {
                          double * o = trace + r * ORACLE_NTRACE;
                          o[0] = tr[r].iter;
                          o[1] = tr[r].cost;
                          o[2] = tr[r].lambda;
                          o[3] = tr[r].dlambda;
                          o[4] = tr[r].alpha;
                          o[5] = tr[r].k_rel_norm;
                          o[6] = tr[r].cost_update_actual;
                          o[7] = tr[r].cost_update_expected;
                          o[8] = tr[r].cost_update_ratio;
                          o[9] = tr[r].alpha_idx;
                          o[10] = tr[r].n_backward;
                          o[11] = tr[r].n_forward;
                        }
                      }
                      if(status)
                      {
                        *status = solver.status();
                      }
                      if(dV)
                      {
                        dV[0] = solver.dV()[0];
                        dV[1] = solver.dV()[1];
                      }
                      if(qp_retval)
                      {
                        std::memcpy(qp_retval, solver.qpRetvalList().data(), solver.qpRetvalList().size() * sizeof(int));
                      }
                      if(qp_free_mask)
                      {
                        std::memcpy(qp_free_mask, solver.qpFreeMaskList().data(),
                                    solver.qpFreeMaskList().size() * sizeof(unsigned));
                      }
                      return 0;
                    });
  }

  /** B independent solves on n_threads std::threads (one solver object per thread, re-used from instance to instance as
      an MPC caller re-uses its DDPSolver), instances handed out dynamically in small chunks, threads pinned.  Batch layouts are the single-solve layouts with a
      leading [B].  trace_last[B][12] receives the last trace row; iters[B] its iter field.  Returns the
      wall time of the threaded region in seconds through *seconds. */
  int oracle_ddp_solve_batch(const char * model,
                             const double * params,
                             const oracle_config * cfg,
                             int B,
                             const double * t0,
                             const double * x0,
                             const double * u_init,
                             const double * lower,
                             const double * upper,
                             int n_threads,
                             double * X,
                             double * U,
                             double * cost,
                             double * k,
                             double * K,
                             int * status,
                             int * iters,
                             double * trace_last,
                             int * alpha_idx_hist, /* [B][max_iter] or NULL */
                             long long * total_iters,
                             double * seconds,
                             int limits_per_step /* 0: [MM]; 1: one [T][MM] table for all; 2: [B][T][MM], one per instance */)
  {
    return dispatch(
        model,
        [&](auto m)
        {
          using M = decltype(m);
          constexpr int N = M::N;
          constexpr int MM = M::MMAX > 0 ? M::MMAX : 1;
          if(params)
          {
            m.setParams(params);
          }
          const int T = cfg->horizon_steps;
          if(n_threads < 1)
          {
            n_threads = 1;
          }
          std::vector<long long> it_count(n_threads, 0);
          // instances are handed out in chunks from a shared counter (solves differ in length: a static partition leaves
          // threads idle at the end); threads are pinned to the CPUs this process may run on, one each, in order
          std::atomic<int> next_chunk(0);
          const int chunk = std::max(1, std::min(16, B / (n_threads * 8)));
          std::vector<int> cpus;
          {
            cpu_set_t set;
            CPU_ZERO(&set);
            if(sched_getaffinity(0, sizeof(set), &set) == 0)
            {
              for(int c = 0; c < CPU_SETSIZE; c++)
              {
                if(CPU_ISSET(c, &set))
                {
                  cpus.push_back(c);
                }
              }
            }
          }
          auto worker = [&](int tid)
          {
            if(n_threads > 1 && !cpus.empty())
            {
              cpu_set_t one;
              CPU_ZERO(&one);
              CPU_SET(cpus[tid % cpus.size()], &one);
              pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
            }
            using R = typename M::Real;
            SolverOf<M> solver(m);
            toConfig(cfg, solver.config());
            const size_t nl = static_cast<size_t>(T) * MM;
            if(cfg->with_input_constraint && limits_per_step == 0)
            {
              solver.setInputLimits(toReal<R>(lower, MM).data(), toReal<R>(upper, MM).data());
            }
            if(cfg->with_input_constraint && limits_per_step == 1)
            {
              solver.setInputLimitsPerStep(toReal<R>(lower, nl).data(), toReal<R>(upper, nl).data(), T);
            }
            std::vector<R> x0r(N), u0r(static_cast<size_t>(T) * MM);
            for(;;)
            {
            const int b0 = next_chunk.fetch_add(chunk);
            if(b0 >= B)
            {
              break;
            }
            const int b1 = std::min(B, b0 + chunk);
            for(int b = b0; b < b1; b++)
            {
              for(int j = 0; j < N; j++)
              {
                x0r[j] = static_cast<R>(x0[static_cast<size_t>(b) * N + j]);
              }
              for(size_t e = 0; e < u0r.size(); e++)
              {
                u0r[e] = static_cast<R>(u_init[static_cast<size_t>(b) * T * MM + e]);
              }
              if(cfg->with_input_constraint && limits_per_step == 2)
              {
                solver.setInputLimitsPerStep(toReal<R>(lower + static_cast<size_t>(b) * nl, nl).data(),
                                             toReal<R>(upper + static_cast<size_t>(b) * nl, nl).data(), T);
              }
              solver.solve(static_cast<R>(t0 ? t0[b] : 0.0), x0r.data(), u0r.data());
              const auto & cd = solver.controlData();
              if(X)
              {
                toDouble(X + static_cast<size_t>(b) * (T + 1) * N, cd.x);
              }
              if(U)
              {
                toDouble(U + static_cast<size_t>(b) * T * MM, cd.u);
              }
              if(cost)
              {
                toDouble(cost + static_cast<size_t>(b) * (T + 1), cd.cost);
              }
              if(k)
              {
                toDouble(k + static_cast<size_t>(b) * T * MM, solver.kList());
              }
              if(K)
              {
                toDouble(K + static_cast<size_t>(b) * T * MM * N, solver.KList());
              }
              const auto & tr = solver.traceDataList();
              const auto & last = tr.back();
              it_count[tid] += last.iter;
              if(status)
              {
                status[b] = solver.status();
              }
              if(iters)
              {
                iters[b] = last.iter;
              }
              if(trace_last)
              {
                double * o = trace_last + static_cast<size_t>(b) * ORACLE_NTRACE;
                o[0] = last.iter;
                o[1] = last.cost;
                o[2] = last.lambda;
                o[3] = last.dlambda;
                o[4] = last.alpha;
                o[5] = last.k_rel_norm;
                o[6] = last.cost_update_actual;
                o[7] = last.cost_update_expected;
                o[8] = last.cost_update_ratio;
                o[9] = last.alpha_idx;
                o[10] = last.n_backward;
                o[11] = last.n_forward;
              }
              if(alpha_idx_hist)
              {
                int * h = alpha_idx_hist + static_cast<size_t>(b) * cfg->max_iter;
                for(int i = 0; i < cfg->max_iter; i++)
                {
                  h[i] = -2;
                }
                for(size_t r = 1; r < tr.size(); r++)
                {
                  h[r - 1] = tr[r].alpha_idx;
                }
              }
            }
            }
          };
          auto t_start = std::chrono::steady_clock::now();
          if(n_threads == 1)
          {
            worker(0);
          }
          else
          {
            std::vector<std::thread> th;
            for(int i = 0; i < n_threads; i++)
            {
              th.emplace_back(worker, i);
            }
            for(auto & t : th)
            {
              t.join();
            }
          }
          auto t_end = std::chrono::steady_clock::now();
          if(seconds)
          {
            *seconds = std::chrono::duration<double>(t_end - t_start).count();
          }
          if(total_iters)
          {
            long long s = 0;
            for(long long c : it_count)
            {
              s += c;
            }
            *total_iters = s;
          }
          return 0;
        });
  }

  /** ROS-free restatement of the reference's receding-horizon loops
        TestDDPBipedal.cpp:243-268, TestDDPVerticalMotion.cpp:290-326, TestDDPCentroidalMotion.cpp:307-347
      (shift_warm_start = 1: next x = x_list[1]; u_list shifted by one, last entry repeated or zero-filled
      when the terminal input dimension changes) and TestDDPCartPole.cpp:323-346,388-403 (shift_warm_start =
      0: u_list reused unshifted, plant integrated with sim_substeps Euler steps of sim_dt under the clamped
      u[0]; only valid for model "cartpole").
      Per tick outputs: t_log[n], x_log[n][N] (state handed to solve), u0_log[n][MM], iter_log[n], m0_log[n].
      x_final[N] = state after the last tick. */
  int oracle_mpc_run(const char * model,
                     const double * params,
                     const oracle_config * cfg,
                     int max_iter_after_first,
                     double t0,
                     const double * x0,
                     int n_ticks,
                     int shift_warm_start,
                     int sim_substeps,
                     double sim_dt,
                     const double * lower,
                     const double * upper,
                     double * t_log,
                     double * x_log,
                     double * u0_log,
                     int * iter_log,
                     int * m0_log,
                     double * x_final,
                     double * t_final)
  {
    return dispatch(
        model,
        [&](auto m)
        {
          using M = decltype(m);
          constexpr int N = M::N;
          constexpr int MM = M::MMAX > 0 ? M::MMAX : 1;
          if(params)
          {
            m.setParams(params);
          }
          const int T = cfg->horizon_steps;
          if constexpr(!std::is_same<typename M::Real, double>::value)
          {
            return -3; // the closed-loop restatements exist for the reference's own (fp64) models
          }
          else
          {
          DDPSolver<M> solver(m);
          toConfig(cfg, solver.config());
          if(lower && upper)
          {
            solver.setInputLimits(lower, upper);
          }
          double current_t = t0;
          double x[N];
          for(int j = 0; j < N; j++)
          {
            x[j] = x0[j];
          }
          std::vector<double> u_list(static_cast<size_t>(T) * MM, 0.0);
          for(int tick = 0; tick < n_ticks; tick++)
          {
            solver.solve(current_t, x, u_list.data());
            if(tick == 0 && max_iter_after_first > 0)
            {
              solver.config().max_iter = max_iter_after_first;
            }
            const auto & cd = solver.controlData();
            const auto & ml = solver.inputDimList();
            if(t_log)
            {
              t_log[tick] = current_t;
            }
            if(x_log)
            {
              for(int j = 0; j < N; j++)
              {
                x_log[static_cast<size_t>(tick) * N + j] = cd.x[j];
              }
            }
            if(u0_log)
            {
              for(int a = 0; a < MM; a++)
              {
                u0_log[static_cast<size_t>(tick) * MM + a] = (a < ml[0]) ? cd.u[a] : 0.0;
              }
            }
            if(iter_log)
            {
              iter_log[tick] = solver.traceDataList().back().iter;
            }
            if(m0_log)
            {
              m0_log[tick] = ml[0];
            }
            if(shift_warm_start)
            {
              for(int j = 0; j < N; j++)
              {
                x[j] = cd.x[N + j];
              }
              // erase(begin); push_back(back) or zeros when the terminal dimension differs
              std::vector<double> nu(static_cast<size_t>(T) * MM, 0.0);
              for(int i = 0; i + 1 < T; i++)
              {
                for(int a = 0; a < MM; a++)
                {
                  nu[static_cast<size_t>(i) * MM + a] = cd.u[static_cast<size_t>(i + 1) * MM + a];
                }
              }
              double terminal_t = current_t + T * m.dt;
              int term_m = m.inputDim(terminal_t);
              int last_m = ml[T - 1];
              if(last_m == term_m)
              {
                for(int a = 0; a < MM; a++)
                {
                  nu[static_cast<size_t>(T - 1) * MM + a] = cd.u[static_cast<size_t>(T - 1) * MM + a];
                }
              }
              u_list = nu;
              current_t += m.dt;
            }
            else
            {
              // cart-pole: clamp u[0] to the limits, integrate the plant, reuse u_list as is
              if constexpr(std::is_same<M, CartPole>::value)
              {
                double u0[1] = {cd.u[0]};
                if(lower && upper)
                {
                  u0[0] = std::min(std::max(u0[0], lower[0]), upper[0]);
                }
                if(u0_log)
                {
                  u0_log[static_cast<size_t>(tick) * MM] = u0[0];
                }
                for(int s = 0; s < sim_substeps; s++)
                {
                  double xn[N];
                  m.stateEqDt(current_t, x, u0, sim_dt, xn);
                  for(int j = 0; j < N; j++)
                  {
                    x[j] = xn[j];
                  }
                  current_t += sim_dt;
                }
                u_list = cd.u;
              }
              else
              {
                return -3;
              }
            }
          }
          if(x_final)
          {
            for(int j = 0; j < N; j++)
            {
              x_final[j] = x[j];
            }
          }
          if(t_final)
          {
            *t_final = current_t;
          }
          return 0;
          }
        });
  }
} // extern "C"
