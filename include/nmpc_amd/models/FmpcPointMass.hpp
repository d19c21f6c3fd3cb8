// Planar point mass with quadratic drag and box limits on its two inputs, for the MI355X FMPC solver.
// No reference counterpart: both problems of the reference's FMPC tests have ONE input, so the matrix G of the Riccati step
// (FmpcSolver.hpp:577) is 1 x 1 there; this problem makes it a full 2 x 2 matrix (the running cost couples the inputs), which
// exercises the pivoted LDLT of the gain solve (FmpcSolver.hpp:581-586).
// State [px, py, vx, vy], input [fx, fy], inequality rows |fx| <= u_max[0], |fy| <= u_max[1].
#pragma once

#include <nmpc_amd/FmpcProblem.hpp>

namespace nmpc_amd
{
class FmpcProblemPointMass : public FmpcProblem<4, 2, 4>
{
public:
  static constexpr const char * kName = "fmpc_pointmass";

  NMPC_HD explicit FmpcProblemPointMass(double dt = 0.02) : FmpcProblem(dt) {}

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    return stateEq(t, x, u, dt_);
  }

  NMPC_HD StateDimVector stateEq(double, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u,
                                 double dt) const
  {
    const double speed = sqrt(x[2] * x[2] + x[3] * x[3] + 1e-6);
    StateDimVector x_next;
    x_next[0] = x[0] + dt * x[2];
    x_next[1] = x[1] + dt * x[3];
    x_next[2] = x[2] + dt * ((u[0] - drag_ * speed * x[2]) / mass_);
    x_next[3] = x[3] + dt * ((u[1] - drag_ * speed * x[3]) / mass_);
    return x_next;
  }

  NMPC_HD double runningCost(double, // t
                             const StateDimVector & x,
                             const InputDimVector & u) const
  {
    const double ex = x[0] - target_[0], ey = x[1] - target_[1];
    return 0.5 * (w_pos_ * (ex * ex + ey * ey) + w_vel_ * (x[2] * x[2] + x[3] * x[3]) + w_u_ * (u[0] * u[0] + u[1] * u[1]))
           + w_u_cross_ * (u[0] * u[1]);
  }

  NMPC_HD double terminalCost(double, // t
                              const StateDimVector & x) const
  {
    const double ex = x[0] - target_[0], ey = x[1] - target_[1];
    return 0.5 * w_term_ * ((ex * ex + ey * ey) + (x[2] * x[2] + x[3] * x[3]));
  }

  NMPC_HD IneqDimVector ineqConst(double, // t
                                  const StateDimVector &, // x
                                  const InputDimVector & u) const
  {
    IneqDimVector g;
    g[0] = -1 * u[0] - u_max_[0];
    g[1] = u[0] - u_max_[0];
    g[2] = -1 * u[1] - u_max_[1];
    g[3] = u[1] - u_max_[1];
    return g;
  }

  NMPC_HD void calcStateEqDeriv(double, // t
                                const StateDimVector & x,
                                const InputDimVector &, // u
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const double speed = sqrt(x[2] * x[2] + x[3] * x[3] + 1e-6);
    const double c = drag_ / mass_;
    state_eq_deriv_x.setZero();
    state_eq_deriv_x(0, 2) = dt_;
    state_eq_deriv_x(1, 3) = dt_;
    state_eq_deriv_x(2, 2) = -dt_ * c * (speed + x[2] * x[2] / speed);
    state_eq_deriv_x(2, 3) = -dt_ * c * (x[2] * x[3] / speed);
    state_eq_deriv_x(3, 2) = -dt_ * c * (x[2] * x[3] / speed);
    state_eq_deriv_x(3, 3) = -dt_ * c * (speed + x[3] * x[3] / speed);
    state_eq_deriv_x.addToDiagonal(1.0);

    state_eq_deriv_u.setZero();
    state_eq_deriv_u(2, 0) = dt_ / mass_;
    state_eq_deriv_u(3, 1) = dt_ / mass_;
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
    running_cost_deriv_x[0] = w_pos_ * (x[0] - target_[0]);
    running_cost_deriv_x[1] = w_pos_ * (x[1] - target_[1]);
    running_cost_deriv_x[2] = w_vel_ * x[2];
    running_cost_deriv_x[3] = w_vel_ * x[3];
    running_cost_deriv_xx.setZero();
    running_cost_deriv_xx(0, 0) = w_pos_;
    running_cost_deriv_xx(1, 1) = w_pos_;
    running_cost_deriv_xx(2, 2) = w_vel_;
    running_cost_deriv_xx(3, 3) = w_vel_;
    running_cost_deriv_u[0] = w_u_ * u[0] + w_u_cross_ * u[1];
    running_cost_deriv_u[1] = w_u_ * u[1] + w_u_cross_ * u[0];
    running_cost_deriv_uu(0, 0) = w_u_;
    running_cost_deriv_uu(1, 0) = w_u_cross_;
    running_cost_deriv_uu(0, 1) = w_u_cross_;
    running_cost_deriv_uu(1, 1) = w_u_;
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double, // t
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_x[0] = w_term_ * (x[0] - target_[0]);
    terminal_cost_deriv_x[1] = w_term_ * (x[1] - target_[1]);
    terminal_cost_deriv_x[2] = w_term_ * x[2];
    terminal_cost_deriv_x[3] = w_term_ * x[3];
    terminal_cost_deriv_xx.setZero();
    terminal_cost_deriv_xx.addToDiagonal(w_term_);
  }

  NMPC_HD void calcIneqConstDeriv(double, // t
                                  const StateDimVector &, // x
                                  const InputDimVector &, // u
                                  IneqStateDimMatrix & ineq_const_deriv_x,
                                  IneqInputDimMatrix & ineq_const_deriv_u) const
  {
    ineq_const_deriv_x.setZero();
    ineq_const_deriv_u.setZero();
    ineq_const_deriv_u(0, 0) = -1;
    ineq_const_deriv_u(1, 0) = 1;
    ineq_const_deriv_u(2, 1) = -1;
    ineq_const_deriv_u(3, 1) = 1;
  }

public:
  double mass_ = 1.5; // [kg]
  double drag_ = 0.3; // [kg/m]
  double target_[2] = {1.0, -0.5}; // [m]
  double w_pos_ = 2.0, w_vel_ = 0.2, w_u_ = 0.05, w_u_cross_ = 0.02, w_term_ = 5.0;
  double u_max_[2] = {2.0, 1.0}; // [N]
};
} // namespace nmpc_amd
