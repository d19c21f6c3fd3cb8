// Cart-pole problem with input and position limits for the MI355X FMPC solver.
// Same model as the reference's test problem FmpcProblemCartPole (nmpc_fmpc/tests/src/TestFmpcCartPole.cpp:32-267):
// state [pos, theta, vel, omega], input [force], explicit-Euler dynamics, quadratic costs, four inequality rows
// (|force| <= u_max, |pos| <= x_max).
#pragma once

#include <nmpc_amd/FmpcProblem.hpp>

namespace nmpc_amd
{
class FmpcProblemCartPole : public FmpcProblem<4, 1, 4>
{
public:
  struct Param
  {
    double cart_mass = 1.0; // [kg]
    double pole_mass = 0.5; // [kg]
    double pole_length = 2.0; // [m]
  };

  struct CostWeight
  {
    double running_x[4] = {0.1, 1.0, 0.01, 0.1};
    double running_u[1] = {0.001};
    double terminal_x[4] = {0.1, 1.0, 0.01, 0.1};
  };

  static constexpr const char * kName = "fmpc_cartpole";
  static constexpr double g_ = 9.80665; // [m/s^2]

  NMPC_HD explicit FmpcProblemCartPole(double dt = 0.01) : FmpcProblem(dt) {}

  /** Reference position of the cart (the reference's ref_pos_func_; its test returns a constant between service calls,
      TestFmpcCartPole.cpp:393-406). */
  NMPC_HD double refPos(double /* t */) const
  {
    return ref_pos_;
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    return stateEq(t, x, u, dt_);
  }

  /** The plant step of the reference's test (TestFmpcCartPole.cpp:73-103). */
  NMPC_HD StateDimVector stateEq(double, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u,
                                 double dt) const
  {
    const double theta = x[1];
    const double vel = x[2];
    const double omega = x[3];
    const double f = u[0];
    const double m1 = param_.cart_mass;
    const double m2 = param_.pole_mass;
    const double l = param_.pole_length;

    double sin_theta, cos_theta;
    sincos(theta, sin_theta, cos_theta);
    const double omega2 = omega * omega;
    const double denom = m1 + m2 * (sin_theta * sin_theta);

    StateDimVector x_next;
    x_next[0] = x[0] + dt * vel;
    x_next[1] = x[1] + dt * omega;
    x_next[2] = x[2] + dt * ((f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta) / denom);
    x_next[3] = x[3]
                + dt
                      * ((f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta)
                         / (l * denom));
    return x_next;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    double cost_x = 0;
    for(int i = 0; i < 4; i++)
    {
      const double e = x[i] - (i == 0 ? refPos(t) : 0.0);
      cost_x += cost_weight_.running_x[i] * (e * e);
    }
    return 0.5 * cost_x + 0.5 * (cost_weight_.running_u[0] * (u[0] * u[0]));
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    double cost_x = 0;
    for(int i = 0; i < 4; i++)
    {
      const double e = x[i] - (i == 0 ? refPos(t) : 0.0);
      cost_x += cost_weight_.terminal_x[i] * (e * e);
    }
    return 0.5 * cost_x;
  }

  NMPC_HD IneqDimVector ineqConst(double, // t
                                  const StateDimVector & x,
                                  const InputDimVector & u) const
  {
    const double u_min = -1 * u_max_;
    const double x_min = -1 * x_max_;
    IneqDimVector g;
    g[0] = -1 * u[0] + u_min;
    g[1] = u[0] - u_max_;
    g[2] = -1 * x[0] + x_min;
    g[3] = x[0] - x_max_;
    return g;
  }

  NMPC_HD void calcStateEqDeriv(double, // t
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const double theta = x[1];
    const double omega = x[3];
    const double f = u[0];
    const double m1 = param_.cart_mass;
    const double m2 = param_.pole_mass;
    const double l = param_.pole_length;

    double sin_theta, cos_theta;
    sincos(theta, sin_theta, cos_theta);
    const double omega2 = omega * omega;
    const double sin2 = sin_theta * sin_theta;
    const double denom = m1 + m2 * sin2;
    const double denom2 = denom * denom;
    const double acc_num = f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta;
    const double alp_num = f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta;
    const double ddenom = 2 * m2 * sin_theta * cos_theta;

    state_eq_deriv_x.setZero();
    state_eq_deriv_x(0, 2) = 1;
    state_eq_deriv_x(1, 3) = 1;
    state_eq_deriv_x(2, 1) =
        ((-1 * m2 * l * omega2 * cos_theta + m2 * g_ * (1 - 2 * sin2)) * denom + -1 * acc_num * ddenom) / denom2;
    state_eq_deriv_x(2, 3) = (-2 * m2 * l * omega * sin_theta) / denom;
    state_eq_deriv_x(3, 1) =
        ((-1 * f * sin_theta + -1 * m2 * l * omega2 * (1 - 2 * sin2) + g_ * (m1 + m2) * cos_theta) * denom
         + -1 * alp_num * ddenom)
        / (l * denom2);
    state_eq_deriv_x(3, 3) = (-2 * m2 * l * omega * sin_theta * cos_theta) / (l * denom);
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.addToDiagonal(1.0);

    state_eq_deriv_u.setZero();
    state_eq_deriv_u[2] = 1 / denom;
    state_eq_deriv_u[3] = cos_theta / (l * denom);
    state_eq_deriv_u *= dt_;
  }

  NMPC_HD void calcRunningCostDeriv(double t,
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & running_cost_deriv_x,
                                    InputDimVector & running_cost_deriv_u,
                                    StateStateDimMatrix & running_cost_deriv_xx,
                                    InputInputDimMatrix & running_cost_deriv_uu,
                                    StateInputDimMatrix & running_cost_deriv_xu) const
  {
    running_cost_deriv_xx.setZero();
    for(int i = 0; i < 4; i++)
    {
      running_cost_deriv_x[i] = cost_weight_.running_x[i] * (x[i] - (i == 0 ? refPos(t) : 0.0));
      running_cost_deriv_xx(i, i) = cost_weight_.running_x[i];
    }
    running_cost_deriv_u[0] = cost_weight_.running_u[0] * u[0];
    running_cost_deriv_uu(0, 0) = cost_weight_.running_u[0];
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_xx.setZero();
    for(int i = 0; i < 4; i++)
    {
      terminal_cost_deriv_x[i] = cost_weight_.terminal_x[i] * (x[i] - (i == 0 ? refPos(t) : 0.0));
      terminal_cost_deriv_xx(i, i) = cost_weight_.terminal_x[i];
    }
  }

  NMPC_HD void calcIneqConstDeriv(double, // t
                                  const StateDimVector &, // x
                                  const InputDimVector &, // u
                                  IneqStateDimMatrix & ineq_const_deriv_x,
                                  IneqInputDimMatrix & ineq_const_deriv_u) const
  {
    ineq_const_deriv_x.setZero();
    ineq_const_deriv_x(2, 0) = -1;
    ineq_const_deriv_x(3, 0) = 1;

    ineq_const_deriv_u.setZero();
    ineq_const_deriv_u(0, 0) = -1;
    ineq_const_deriv_u(1, 0) = 1;
  }

public:
  Param param_;
  CostWeight cost_weight_;
  double ref_pos_ = 0.0; // [m]
  double u_max_ = 15.0; // [N] (constexpr in the reference, TestFmpcCartPole.cpp:122)
  double x_max_ = 20.0; // [m] (TestFmpcCartPole.cpp:124)
};
} // namespace nmpc_amd
