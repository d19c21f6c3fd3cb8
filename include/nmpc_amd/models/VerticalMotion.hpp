// Vertical-motion problem with a time-varying INPUT DIMENSION for the MI355X DDP solver.
// Same model as the reference's test problem DDPProblemVerticalMotion
// (nmpc_ddp/tests/src/TestDDPVerticalMotion.cpp:31-234; ref_pos schedule :246-259): state [pos_z, vel_z],
// input = one force per contact, with 1, 2 or 0 contacts depending on t.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
class DDPProblemVerticalMotion : public DDPProblem<2, Dynamic, 2>
{
public:
  struct CostWeight
  {
    double running_x[2] = {1.0, 1e-3};
    double running_u = 1e-4;
    double terminal_x[2] = {1.0, 1e-3};
  };

  static constexpr const char * kName = "vertical";
  static constexpr double g_ = 9.80665; // [m/s^2]

  NMPC_HD explicit DDPProblemVerticalMotion(double dt = 0.01) : DDPProblem(dt) {}

  NMPC_HD double refPos(double t) const
  {
    t += 1e-6;
    return (t < ref_switch_t_) ? 1.0 : 0.0; // [m]
  }

  NMPC_HD int inputDim(double t) const
  {
    t += 1e-6;
    if(2.0 < t && t < 3.0)
    {
      return 2;
    }
    if(4.5 < t && t < 5.0)
    {
      return 0;
    }
    return 1;
  }

  NMPC_HD StateDimVector stateEq(double, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u) const
  {
    StateDimVector x_next;
    x_next[0] = x[0] + dt_ * x[1];
    x_next[1] = x[1] + dt_ * (u.sum() / mass_ - g_);
    return x_next;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const double e0 = x[0] - refPos(t);
    const double e1 = x[1];
    const double cost_x = 0.5 * (cost_weight_.running_x[0] * (e0 * e0) + cost_weight_.running_x[1] * (e1 * e1));
    const double cost_u = 0.5 * cost_weight_.running_u * u.squaredNorm();
    return cost_x + cost_u;
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    const double e0 = x[0] - refPos(t);
    const double e1 = x[1];
    return 0.5 * (cost_weight_.terminal_x[0] * (e0 * e0) + cost_weight_.terminal_x[1] * (e1 * e1));
  }

  NMPC_HD void calcStateEqDeriv(double, // t
                                const StateDimVector &, // x
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    state_eq_deriv_x.setZero();
    state_eq_deriv_x(0, 1) = 1;
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.addToDiagonal(1.0);

    state_eq_deriv_u.resize(2, u.size());
    state_eq_deriv_u.setZero();
    for(int i = 0; i < u.size(); i++)
    {
      state_eq_deriv_u(1, i) = (1.0 / mass_) * dt_;
    }
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
    running_cost_deriv_x[0] = cost_weight_.running_x[0] * (x[0] - refPos(t));
    running_cost_deriv_x[1] = cost_weight_.running_x[1] * x[1];
    running_cost_deriv_xx.setZero();
    running_cost_deriv_xx(0, 0) = cost_weight_.running_x[0];
    running_cost_deriv_xx(1, 1) = cost_weight_.running_x[1];
    running_cost_deriv_xu.resize(2, u.size());
    running_cost_deriv_xu.setZero();

    running_cost_deriv_u.resize(u.size());
    running_cost_deriv_uu.resize(u.size(), u.size());
    running_cost_deriv_uu.setZero();
    for(int i = 0; i < u.size(); i++)
    {
      running_cost_deriv_u[i] = cost_weight_.running_u * u[i];
      running_cost_deriv_uu(i, i) = 1.0 * cost_weight_.running_u;
    }
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_x[0] = cost_weight_.terminal_x[0] * (x[0] - refPos(t));
    terminal_cost_deriv_x[1] = cost_weight_.terminal_x[1] * x[1];
    terminal_cost_deriv_xx.setZero();
    terminal_cost_deriv_xx(0, 0) = cost_weight_.terminal_x[0];
    terminal_cost_deriv_xx(1, 1) = cost_weight_.terminal_x[1];
  }

public:
  CostWeight cost_weight_;
  double mass_ = 1.0; // [kg]
  double ref_switch_t_ = 8.0; // [sec] reference height drops from 1 m to 0 m here
};
} // namespace nmpc_amd
