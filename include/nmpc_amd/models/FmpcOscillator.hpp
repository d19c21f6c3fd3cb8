// Van der Pol oscillator problem for the MI355X FMPC solver.
// Same model as the reference's test problem FmpcProblemOscillator (nmpc_fmpc/tests/src/TestFmpcOscillator.cpp:18-135;
// https://web.casadi.org/docs/#a-simple-test-problem): state [x0, x1], input [u], three inequality rows
// (x1 >= -0.05, -1 <= u <= 0.9).
#pragma once

#include <nmpc_amd/FmpcProblem.hpp>

namespace nmpc_amd
{
class FmpcProblemOscillator : public FmpcProblem<2, 1, 3>
{
public:
  static constexpr const char * kName = "fmpc_oscillator";

  NMPC_HD explicit FmpcProblemOscillator(double dt = 0.01) : FmpcProblem(dt) {}

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    return stateEq(t, x, u, dt_);
  }

  /** The plant step of the reference's test (TestFmpcOscillator.cpp:27-37). */
  NMPC_HD StateDimVector stateEq(double, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u,
                                 double dt) const
  {
    StateDimVector x_next;
    x_next[0] = x[0] + dt * ((1.0 - x[1] * x[1]) * x[0] - x[1] + u[0]);
    x_next[1] = x[1] + dt * x[0];
    return x_next;
  }

  NMPC_HD double runningCost(double, // t
                             const StateDimVector & x,
                             const InputDimVector & u) const
  {
    return 0.5 * ((x[0] * x[0] + x[1] * x[1]) + u[0] * u[0]);
  }

  NMPC_HD double terminalCost(double, // t
                              const StateDimVector & // x
  ) const
  {
    return 0;
  }

  NMPC_HD IneqDimVector ineqConst(double, // t
                                  const StateDimVector & x,
                                  const InputDimVector & u) const
  {
    IneqDimVector g;
    g[0] = -1 * x[1] - 0.05;
    g[1] = -1 * u[0] - 1.0;
    g[2] = u[0] - 0.9;
    return g;
  }

  NMPC_HD void calcStateEqDeriv(double, // t
                                const StateDimVector & x,
                                const InputDimVector &, // u
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    state_eq_deriv_x.setZero();
    state_eq_deriv_x(0, 0) = 1.0 - x[1] * x[1];
    state_eq_deriv_x(0, 1) = -2 * x[0] * x[1] - 1.0;
    state_eq_deriv_x(1, 0) = 1;
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.addToDiagonal(1.0);

    state_eq_deriv_u.setZero();
    state_eq_deriv_u(0, 0) = 1;
    state_eq_deriv_u *= dt_;
  }

  NMPC_HD void calcRunningCostDeriv(double, // t
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & running_cost_deriv_x,
                                    InputDimVector & running_cost_deriv_u,
                                    StateStateDimMatrix & running_cost_deriv_xx,
                                    InputInputDimMatrix & running_cost_deriv_uu,
                                    StateInputDimMatrix & running_cost_deriv_xu) const
  {
    running_cost_deriv_x = x;
    running_cost_deriv_u = u;
    running_cost_deriv_xx.setIdentity();
    running_cost_deriv_uu.setIdentity();
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double, // t
                                     const StateDimVector &, // x
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_x.setZero();
    terminal_cost_deriv_xx.setZero();
  }

  NMPC_HD void calcIneqConstDeriv(double, // t
                                  const StateDimVector &, // x
                                  const InputDimVector &, // u
                                  IneqStateDimMatrix & ineq_const_deriv_x,
                                  IneqInputDimMatrix & ineq_const_deriv_u) const
  {
    ineq_const_deriv_x.setZero();
    ineq_const_deriv_x(0, 1) = -1;

    ineq_const_deriv_u.setZero();
    ineq_const_deriv_u(1, 0) = -1;
    ineq_const_deriv_u(2, 0) = 1;
  }
};
} // namespace nmpc_amd
