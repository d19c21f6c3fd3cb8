// The reference's closed-loop FMPC test (nmpc_fmpc/tests/src/TestFmpcOscillator.cpp:137-205) through the C++ mirror
// nmpc_amd::FmpcSolverBatch, for a batch of initial states at once.  Host code only (g++, no HIP): the problem type is already
// compiled into libnmpc_hip_ddp.so.  Prints one line per instance; exits non-zero if a bound of the reference's test fails.
//
//   g++ -std=c++17 -O2 -Iinclude examples/fmpc_oscillator_mpc.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp -Wl,-rpath,$PWD/nmpc_amd/lib
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include <nmpc_amd/FmpcSolverBatch.hpp>
#include <nmpc_amd/models/FmpcOscillator.hpp>

using Problem = nmpc_amd::FmpcProblemOscillator;
using Solver = nmpc_amd::FmpcSolverBatch<Problem>;
using Variable = Solver::Variable;
using Status = Solver::Status;

int main(int argc, char ** argv)
{
  const int batch = argc > 1 ? std::atoi(argv[1]) : 8;
  const double end_t = argc > 2 ? std::atof(argv[2]) : 10.0;
  const double horizon_dt = 0.01; // [sec]
  const double horizon_duration = 4.0; // [sec]
  const int horizon_steps = static_cast<int>(horizon_duration / horizon_dt);

  auto fmpc_problem = std::make_shared<Problem>(horizon_dt);
  auto fmpc_solver = std::make_shared<Solver>(fmpc_problem, batch, horizon_steps);
  fmpc_solver->config().max_iter = 3;
  std::vector<Variable> variable(batch, Variable(horizon_steps));
  for(auto & v : variable)
  {
    v.reset(0.0, 0.0, 0.0, 1e0, 1e0);
  }

  const double sim_dt = 0.005; // [sec]
  std::vector<double> current_t(batch, 0.0);
  std::vector<Problem::StateDimVector> current_x(batch);
  for(int b = 0; b < batch; b++)
  {
    current_x[b][0] = 0.02 * b; // instance 0: the reference's initial state (0, 1)
    current_x[b][1] = 1.0;
  }

  int failures = 0;
  bool first_iter = true;
  while(current_t[0] < end_t)
  {
    // the first solve uploads the initial guess; later ones continue from the resident variables
    const auto status = first_iter ? fmpc_solver->solve(current_t, current_x, variable) : fmpc_solver->solve(current_t, current_x);
    if(first_iter)
    {
      first_iter = false;
      fmpc_solver->dumpTraceDataList("/tmp/TestFmpcOscillatorTraceData.txt");
    }
    const auto result = fmpc_solver->variable();
    for(int b = 0; b < batch; b++)
    {
      if(!(status[b] == Status::Succeeded || status[b] == Status::MaxIterationReached))
      {
        failures++;
      }
      const Problem::InputDimVector current_u = result[b].u_list[0];
      const Problem::IneqDimVector current_g = fmpc_problem->ineqConst(current_t[b], current_x[b], current_u);
      for(int j = 0; j < 3; j++)
      {
        if(b == 0 && !(current_g[j] <= 0))
        {
          failures++;
          std::printf("inequality constraint %d violated at t = %g: %g\n", j, current_t[b], current_g[j]);
        }
      }
      current_x[b] = fmpc_problem->stateEq(current_t[b], current_x[b], current_u, sim_dt);
      current_t[b] += sim_dt;
    }
  }

  for(int b = 0; b < batch; b++)
  {
    std::printf("instance %d final x = (%.6e, %.6e)\n", b, current_x[b][0], current_x[b][1]);
    if(end_t >= 10.0 && !(std::abs(current_x[b][0]) < 1e-2 && std::abs(current_x[b][1]) < 1e-2))
    {
      failures++;
    }
  }
  std::printf("failures %d solve_ms %.3f\n", failures, fmpc_solver->computationDuration().solve);
  return failures == 0 ? 0 : 1;
}
