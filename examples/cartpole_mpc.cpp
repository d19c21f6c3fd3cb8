// Closed-loop swing-up of a batch of cart-poles, device-resident: the reference's MPC test
// (nmpc_ddp/tests/src/TestDDPCartPole.cpp:236-403 with the parameters of tests/test/TestDDPCartPole.test:14-26:
// horizon 2 s at dt 0.01, +-15 N input box, max_iter 3, MPC every 4 ms, plant integrated at 2 ms, 10 s) with a batch
// axis, through DDPSolverBatch::mpcRun — one call, no host round trip between the 2500 solves.  Build + run:
//   g++ -std=c++17 -O2 -Iinclude examples/cartpole_mpc.cpp -Lnmpc_amd/lib -lnmpc_hip_ddp
//       -Wl,-rpath,$PWD/nmpc_amd/lib -o /tmp/cartpole_mpc && /tmp/cartpole_mpc 64
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include <nmpc_amd/DDPSolverBatch.hpp>
#include <nmpc_amd/models/CartPole.hpp>

int main(int argc, char ** argv)
{
  using Problem = nmpc_amd::DDPProblemCartPole;
  using Solver = nmpc_amd::DDPSolverBatch<Problem>;
  const int batch = argc > 1 ? std::atoi(argv[1]) : 8;
  const int n_ticks = argc > 2 ? std::atoi(argv[2]) : 2500;

  auto ddp_problem = std::make_shared<Problem>(0.01);
  ddp_problem->cost_weight_.running_u[0] = 0.01; // TestDDPCartPole.test:23
  auto ddp_solver = std::make_shared<Solver>(ddp_problem, batch);
  ddp_solver->config().horizon_steps = 200;
  ddp_solver->config().max_iter = 3;
  ddp_solver->config().with_input_constraint = true;
  ddp_solver->config().print_level = 0;
  ddp_solver->setInputLimitsFunc(
      [](double) -> std::array<Problem::InputDimVector, 2>
      {
        std::array<Problem::InputDimVector, 2> limits;
        limits[0][0] = -15.0; // TestDDPCartPole.cpp:379-386
        limits[1][0] = 15.0;
        return limits;
      });

  std::vector<double> current_t(batch, 0.0);
  std::vector<Problem::StateDimVector> current_x(batch);
  std::vector<std::vector<Problem::InputDimVector>> initial_u_list(batch);
  for(int b = 0; b < batch; b++)
  {
    current_x[b][0] = 0.05 * b; // instance 0 is the reference's start (0, pi, 0, 0), :308
    current_x[b][1] = M_PI - 0.01 * b;
    current_x[b][2] = 0.0;
    current_x[b][3] = 0.0;
    Problem::InputDimVector zero;
    zero.setZero();
    initial_u_list[b].assign(ddp_solver->config().horizon_steps, zero);
  }

  const Solver::MpcLog log = ddp_solver->mpcRun(current_t, current_x, initial_u_list, n_ticks,
                                                /* shift_warm_start */ false, /* max_iter_after_first */ 0,
                                                /* sim_substeps */ 2, /* sim_dt */ 0.002, /* clamp_u0 */ true);
  for(int b = 0; b < batch; b++)
  {
    double max_abs_u = 0, max_abs_pos = 0;
    for(int k = 0; k < n_ticks; k++)
    {
      max_abs_u = std::fmax(max_abs_u, std::fabs(log.u0[static_cast<size_t>(b) * n_ticks + k]));
      max_abs_pos = std::fmax(max_abs_pos, std::fabs(log.x[(static_cast<size_t>(b) * n_ticks + k) * 4]));
    }
    std::printf("instance %d t_final %.3f x_final %.9e %.9e %.9e %.9e max|u0| %.6f max|pos| %.6f\n", b, log.t_final[b],
                log.x_final[b * 4 + 0], log.x_final[b * 4 + 1], log.x_final[b * 4 + 2], log.x_final[b * 4 + 3], max_abs_u,
                max_abs_pos);
  }
  return 0;
}
