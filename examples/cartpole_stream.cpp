// A queue of cart-pole problems through the slots of ONE solver (nmpc_amd::DDPSolverBatch::solveStream, plain C++ over the C-ABI): every
// problem is solved to ITS convergence (DDPSolver.hpp:115-123) and the slot of a finished one takes the next of the queue.  Every
// instance is checked bit for bit against the same problems solved chunk by chunk with solve().
//   g++ -std=c++17 -O2 -Iinclude examples/cartpole_stream.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp -Wl,-rpath,$PWD/nmpc_amd/lib
//       -o /tmp/cartpole_stream && /tmp/cartpole_stream [instances] [slots]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include <nmpc_amd/DDPSolverBatch.hpp>
#include <nmpc_amd/models/CartPole.hpp>

using Problem = nmpc_amd::DDPProblemCartPole;
using Solver = nmpc_amd::DDPSolverBatch<Problem>;

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

int main(int argc, char ** argv)
{
  const int N = argc > 1 ? std::atoi(argv[1]) : 3000;
  const int S = argc > 2 ? std::atoi(argv[2]) : 512;
  const int T = 100;
  auto problem = std::make_shared<Problem>(0.01);
  std::vector<double> t(N, 0.0);
  std::vector<Problem::StateDimVector> x(N);
  Problem::InputDimVector zero;
  zero.setZero();
  std::vector<std::vector<Problem::InputDimVector>> u(N, std::vector<Problem::InputDimVector>(T, zero));
  unsigned long long seed = 4321;
  const double lo[4] = {-1.0, -M_PI, -1.0, -1.0}, hi[4] = {1.0, M_PI, 1.0, 1.0};
  for(int i = 0; i < N; i++)
  {
    for(int j = 0; j < 4; j++)
    {
      x[i][j] = lo[j] + (hi[j] - lo[j]) * uniform01(seed);
    }
  }

  Solver solver(problem, S);
  solver.config().horizon_steps = T;
  solver.config().print_level = 0;
  solver.config().max_iter = 120;
  solver.config().trace_level = 0;
  solver.setKernel("quad");
  const auto r = solver.solveStream(t, x, u);

  Solver lone(problem, S);
  lone.config() = solver.config();
  lone.config().ragged_schedule = -1;
  lone.setKernel("quad");
  int bad = 0;
  long long its = 0;
  for(int base = 0; base < N; base += S)
  {
    // (the last chunk is padded with copies of its first instance: solve() takes whole batches)
    std::vector<double> tc(S, 0.0);
    std::vector<Problem::StateDimVector> xc(S);
    std::vector<std::vector<Problem::InputDimVector>> uc(S, std::vector<Problem::InputDimVector>(T, zero));
    for(int k = 0; k < S; k++)
    {
      xc[k] = x[base + k < N ? base + k : base];
    }
    lone.solve(tc, xc, uc);
    for(int k = 0; k < S && base + k < N; k++)
    {
      const auto & a = r.control_data[base + k];
      const auto & b = lone.controlData(k);
      bool same = r.status[base + k] == lone.status(k) && std::memcmp(a.cost_list.data(), b.cost_list.data(), a.cost_list.size() * sizeof(double)) == 0;
      for(size_t i = 0; same && i < a.x_list.size(); i++)
      {
        for(int j = 0; j < 4; j++)
        {
          same = same && std::memcmp(&a.x_list[i][j], &b.x_list[i][j], sizeof(double)) == 0;
        }
      }
      for(size_t i = 0; same && i < a.u_list.size(); i++)
      {
        same = same && std::memcmp(&a.u_list[i][0], &b.u_list[i][0], sizeof(double)) == 0;
      }
      bad += same ? 0 : 1;
      its += r.iters[base + k];
    }
  }
  std::printf("%d instances through %d slots: %d rounds, %.2f ms on the device, %lld instance-iterations, instances differing from their lone "
              "solves: %d\n", N, S, r.rounds, r.device_ms, its, bad);
  std::printf(bad == 0 ? "STREAM_OK\n" : "STREAM_BAD\n");
  return bad == 0 ? 0 : 1;
}
