// Planar VTOL (bicopter) problem, n = 6, m = 2, for the MI355X DDP solver.
// The reference has no such model; this one is defined by this project to put a shape with 5 <= n <= 8
// through the matrix-core path (ddp_kernels_tile64.hpp) — the reference's template takes any StateDim / InputDim
// (nmpc_ddp/include/nmpc_ddp/DDPSolver.h:23-25):
//   state  x = [px, pz, theta, vx, vz, omega]        input  u = thrust of the left / right rotor
//   vx' = -(u0 + u1) / m sin(theta),  vz' = (u0 + u1) / m cos(theta) - g,  omega' = arm (u1 - u0) / J;  explicit Euler
//   quadratic costs around hover at ref_pos.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
class DDPProblemPlanarVtol : public DDPProblem<6, 2>
{
public:
  static constexpr const char * kName = "planar_vtol";
  static constexpr double g_ = 9.80665; // [m/s^2]

  NMPC_HD explicit DDPProblemPlanarVtol(double dt = 0.02) : DDPProblem(dt) {}

  NMPC_HD double hoverThrust() const
  {
    return mass_ * g_ / 2;
  }
  NMPC_HD double stateWeight(int i) const
  {
    return i < 2 ? w_pos_ : (i == 2 ? w_ang_ : (i < 5 ? w_vel_ : w_omega_));
  }
  NMPC_HD double stateError(const StateDimVector & x, int i) const
  {
    return i < 2 ? x[i] - ref_pos_[i] : x[i];
  }

  NMPC_HD StateDimVector stateEq(double, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u) const
  {
    double s, c;
    sincosFast(x[2], s, c);
    const double accel = (u[0] + u[1]) / mass_;
    StateDimVector x_next;
    x_next[0] = x[0] + dt_ * x[3];
    x_next[1] = x[1] + dt_ * x[4];
    x_next[2] = x[2] + dt_ * x[5];
    x_next[3] = x[3] + dt_ * (-accel * s);
    x_next[4] = x[4] + dt_ * (accel * c - g_);
    x_next[5] = x[5] + dt_ * (arm_ * (u[1] - u[0]) / inertia_);
    return x_next;
  }

  NMPC_HD double runningCost(double, const StateDimVector & x, const InputDimVector & u) const
  {
    double cost_x = 0;
    for(int i = 0; i < 6; i++)
    {
      const double e = stateError(x, i);
      cost_x += stateWeight(i) * (e * e);
    }
    double cost_u = 0;
    for(int a = 0; a < 2; a++)
    {
      const double e = u[a] - hoverThrust();
      cost_u += e * e;
    }
    return 0.5 * cost_x + 0.5 * w_u_ * cost_u;
  }

  NMPC_HD double terminalCost(double, const StateDimVector & x) const
  {
    double cost_x = 0;
    for(int i = 0; i < 6; i++)
    {
      const double e = stateError(x, i);
      cost_x += (wt_scale_ * stateWeight(i)) * (e * e);
    }
    return 0.5 * cost_x;
  }

  NMPC_HD void calcStateEqDeriv(double, // t
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    double s, c;
    sincosFast(x[2], s, c);
    const double accel = (u[0] + u[1]) / mass_;
    state_eq_deriv_x.setIdentity();
    state_eq_deriv_x(0, 3) = dt_;
    state_eq_deriv_x(1, 4) = dt_;
    state_eq_deriv_x(2, 5) = dt_;
    state_eq_deriv_x(3, 2) = dt_ * (-accel * c);
    state_eq_deriv_x(4, 2) = dt_ * (-accel * s);
    state_eq_deriv_u.setZero();
    for(int a = 0; a < 2; a++)
    {
      state_eq_deriv_u(3, a) = dt_ * (-s / mass_);
      state_eq_deriv_u(4, a) = dt_ * (c / mass_);
    }
    state_eq_deriv_u(5, 0) = dt_ * (-arm_ / inertia_);
    state_eq_deriv_u(5, 1) = dt_ * (arm_ / inertia_);
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
    running_cost_deriv_xx.setZero();
    for(int i = 0; i < 6; i++)
    {
      running_cost_deriv_x[i] = stateWeight(i) * stateError(x, i);
      running_cost_deriv_xx(i, i) = stateWeight(i);
    }
    running_cost_deriv_uu.setZero();
    for(int a = 0; a < 2; a++)
    {
      running_cost_deriv_u[a] = w_u_ * (u[a] - hoverThrust());
      running_cost_deriv_uu(a, a) = w_u_;
    }
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double, // t
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_xx.setZero();
    for(int i = 0; i < 6; i++)
    {
      terminal_cost_deriv_x[i] = (wt_scale_ * stateWeight(i)) * stateError(x, i);
      terminal_cost_deriv_xx(i, i) = wt_scale_ * stateWeight(i);
    }
  }

public:
  double mass_ = 1.0; // [kg]
  double inertia_ = 0.02; // [kg m^2]
  double arm_ = 0.25; // [m]
  double w_pos_ = 1.0, w_ang_ = 0.5, w_vel_ = 0.1, w_omega_ = 0.05;
  double w_u_ = 0.01;
  double wt_scale_ = 10.0; // terminal weight = wt_scale * running weight
  double ref_pos_[2] = {0.0, 1.0}; // [m]
};
} // namespace nmpc_amd
