// Host-side use of the batched solver from plain C++ (compiled with g++, no HIP headers needed): the calling
// pattern of the reference's TestDDPCartPole (nmpc_ddp/tests/src/TestDDPCartPole.cpp:266-306, 388-403) with a
// batch axis.  Build + run:
//   g++ -std=c++17 -O2 -Iinclude examples/cartpole_batch.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp
//       -Wl,-rpath,$PWD/nmpc_amd/lib -o /tmp/cartpole_batch && /tmp/cartpole_batch
#include <cmath>
#include <cstdio>
#include <memory>

#include <nmpc_amd/DDPSolverBatch.hpp>
#include <nmpc_amd/models/CartPole.hpp>

int main(int argc, char ** argv)
{
  using Problem = nmpc_amd::DDPProblemCartPole;
  using Solver = nmpc_amd::DDPSolverBatch<Problem>;
  const int batch = argc > 1 ? std::atoi(argv[1]) : 8;

  // Instantiate problem (parameters can be edited exactly as in the reference test, :274-289)
  auto ddp_problem = std::make_shared<Problem>(0.01);
  ddp_problem->cost_weight_.running_u[0] = 0.001;

  // Instantiate solver
  auto ddp_solver = std::make_shared<Solver>(ddp_problem, batch);
  ddp_solver->config().horizon_steps = 100;
  ddp_solver->config().print_level = 0;

  // Instance 0 is the reference's swing-up start (:306); the others start from perturbed angles
  std::vector<double> current_t(batch, 0.0);
  std::vector<Problem::StateDimVector> current_x(batch);
  std::vector<std::vector<Problem::InputDimVector>> initial_u_list(batch);
  for(int b = 0; b < batch; b++)
  {
    current_x[b][0] = 0.0;
    current_x[b][1] = M_PI - 0.25 * b;
    current_x[b][2] = 0.0;
    current_x[b][3] = 0.0;
    Problem::InputDimVector zero;
    zero.setZero();
    initial_u_list[b].assign(ddp_solver->config().horizon_steps, zero);
  }

  const std::vector<bool> ok = ddp_solver->solve(current_t, current_x, initial_u_list);
  for(int b = 0; b < batch; b++)
  {
    const auto & trace = ddp_solver->traceDataList(b);
    const auto & cd = ddp_solver->controlData(b);
    double cost = 0;
    for(double c : cd.cost_list)
    {
      cost += c;
    }
    std::printf("instance %d converged %d iter %d cost %.12e u0 %.12e theta_end %.6e\n", b, static_cast<int>(ok[b]),
                trace.back().iter, cost, cd.u_list[0][0], cd.x_list.back()[1]);
  }
  ddp_solver->dumpTraceDataList(0, "/tmp/CartPoleBatchTraceData.txt");
  std::printf("solve %.3f ms (kernel %.3f ms)\n", ddp_solver->computationDuration().solve,
              ddp_solver->computationDuration().opt);

  // misuse raises the same exception types as the reference (DDPSolver.hpp:41-45)
  initial_u_list[0].pop_back();
  try
  {
    ddp_solver->solve(current_t, current_x, initial_u_list);
    std::printf("ERROR: no exception\n");
    return 1;
  }
  catch(const std::invalid_argument & e)
  {
    std::printf("invalid_argument: %s\n", e.what());
  }
  return 0;
}
