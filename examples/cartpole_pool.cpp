// Consecutive batches solved to convergence on a pool of handles (nmpc_amd::DDPSolverPool, plain C++ over the C-ABI): the tail of one
// batch — the few instances that iterate for hundreds of iterations (DDPSolver.hpp:115-123: every solver runs its own loop to the
// end) — overlaps with the next batches, and with the ragged-convergence schedule (Configuration::ragged_schedule) a converged
// instance vacates its slot within sixteen iterations.  Every batch's results are checked bit for bit against a lone solver's.
//   g++ -std=c++17 -O2 -Iinclude examples/cartpole_pool.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp -Wl,-rpath,$PWD/nmpc_amd/lib
//       -o /tmp/cartpole_pool && /tmp/cartpole_pool [batch] [n_batches] [n_handles]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include <nmpc_amd/DDPSolverBatch.hpp>
#include <nmpc_amd/models/CartPole.hpp>

using Problem = nmpc_amd::DDPProblemCartPole;
using Solver = nmpc_amd::DDPSolverBatch<Problem>;

struct Batch
{
  std::vector<double> t;
  std::vector<Problem::StateDimVector> x;
  std::vector<std::vector<Problem::InputDimVector>> u;
};

// splitmix64 -> U[0, 1): the generator of nmpc_amd/workloads.py
static double uniform01(unsigned long long & state)
{
  state += 0x9E3779B97F4A7C15ull;
  unsigned long long z = state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
}

static Batch makeBatch(int B, int T, unsigned long long seed)
{
  Batch b;
  b.t.assign(B, 0.0);
  b.x.resize(B);
  b.u.resize(B);
  const double lo[4] = {-1.0, -M_PI, -1.0, -1.0}, hi[4] = {1.0, M_PI, 1.0, 1.0};
  Problem::InputDimVector zero;
  zero.setZero();
  for(int i = 0; i < B; i++)
  {
    for(int j = 0; j < 4; j++)
    {
      b.x[i][j] = lo[j] + (hi[j] - lo[j]) * uniform01(seed);
    }
    b.u[i].assign(T, zero);
  }
  return b;
}

static bool sameResults(const Solver & a, const Solver & b, int B)
{
  for(int i = 0; i < B; i++)
  {
    const auto & ca = a.controlData(i);
    const auto & cb = b.controlData(i);
    if(a.status(i) != b.status(i) || a.traceDataList(i).size() != b.traceDataList(i).size()
       || std::memcmp(ca.cost_list.data(), cb.cost_list.data(), ca.cost_list.size() * sizeof(double)) != 0)
    {
      return false;
    }
    for(size_t k = 0; k < ca.x_list.size(); k++)
    {
      for(int j = 0; j < 4; j++)
      {
        if(std::memcmp(&ca.x_list[k][j], &cb.x_list[k][j], sizeof(double)) != 0)
        {
          return false;
        }
      }
    }
    for(size_t k = 0; k < ca.u_list.size(); k++)
    {
      if(std::memcmp(&ca.u_list[k][0], &cb.u_list[k][0], sizeof(double)) != 0 || a.KList(i)[k](0, 1) != b.KList(i)[k](0, 1))
      {
        return false;
      }
    }
  }
  return true;
}

int main(int argc, char ** argv)
{
  const int B = argc > 1 ? std::atoi(argv[1]) : 4096;
  const int n_batches = argc > 2 ? std::atoi(argv[2]) : 8;
  const int n_handles = argc > 3 ? std::atoi(argv[3]) : 8;
  const int T = 100;
  auto problem = std::make_shared<Problem>(0.01);

  std::vector<Batch> batches;
  for(int k = 0; k < n_batches; k++)
  {
    batches.push_back(makeBatch(B, T, 1234 + 7919ull * k));
  }

  // reference results: every batch on a lone solver, one whole-solve launch each
  Solver lone(problem, B);
  lone.config().horizon_steps = T;
  lone.config().print_level = 0;
  lone.config().ragged_schedule = -1;
  lone.solve(batches[0].t, batches[0].x, batches[0].u); // (warm-up: library load, allocations)
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::unique_ptr<Solver>> refs;
  for(int k = 0; k < n_batches; k++)
  {
    refs.emplace_back(new Solver(problem, B));
    refs.back()->config() = lone.config();
    refs.back()->solve(batches[k].t, batches[k].x, batches[k].u);
  }
  const double lone_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("lone solver, whole-solve launches: %d batches of %d in %.1f ms (launches per solve: %d)\n", n_batches, B, lone_ms,
              refs.back()->lastSolveLaunches());

  int bad = 0;
  for(int ragged : {-1, 0})
  {
    nmpc_amd::DDPSolverPool<Problem> pool(problem, B, n_handles);
    pool.config().horizon_steps = T;
    pool.config().print_level = 0;
    pool.config().ragged_schedule = ragged;
    for(int k = 0; k < n_handles; k++) // warm-up: every handle allocates on its first solve
    {
      pool.submit(batches[0].t, batches[0].x, batches[0].u);
    }
    pool.waitAll();
    std::vector<int> where(n_batches, -1);
    t0 = std::chrono::steady_clock::now();
    for(int k = 0; k < n_batches; k++)
    {
      const int h = k % n_handles;
      // submit() waits for the handle's previous batch and fetches it before the new one is queued: the returned solver's accessors
      // hold the PREVIOUS batch until this one is waited for
      Solver & s = pool.submit(batches[k].t, batches[k].x, batches[k].u);
      if(k >= n_handles)
      {
        const bool same = s.inFlight() && sameResults(s, *refs[k - n_handles], B);
        if(!same)
        {
          std::printf("  batch %d (handle %d) differs from the lone solver's\n", k - n_handles, h);
        }
        bad += same ? 0 : 1;
      }
      where[k] = h;
    }
    pool.waitAll();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for(int k = std::max(0, n_batches - n_handles); k < n_batches; k++)
    {
      const bool same = sameResults(pool.solver(where[k]), *refs[k], B);
      if(!same)
      {
        std::printf("  batch %d (handle %d) differs from the lone solver's\n", k, where[k]);
      }
      bad += same ? 0 : 1;
    }
    std::printf("pool of %d handles, ragged_schedule %2d: %d batches in %.1f ms (launches per solve: %d), batches differing from the "
                "lone solver: %d\n", n_handles, ragged, n_batches, ms, pool.solver(0).lastSolveLaunches(), bad);
  }
  return bad == 0 ? 0 : 1;
}
