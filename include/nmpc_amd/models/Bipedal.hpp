// Bipedal CoM-ZMP problem (linear time-varying) for the MI355X DDP solver.
// Same model and schedules as the reference's test problem DDPProblemBipedal and its TestCase1
// (nmpc_ddp/tests/src/TestDDPBipedal.cpp:16-144 model, :171-225 ref_zmp / omega^2 schedules):
// state [CoM_pos, CoM_vel], input [ZMP].
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
class DDPProblemBipedal : public DDPProblem<2, 1>
{
public:
  struct CostWeight
  {
    double running_vel = 1e-14;
    double running_zmp = 1e-1;
    double terminal_pos = 1e2;
    double terminal_vel = 1.0;
  };

  static constexpr const char * kName = "bipedal";

  NMPC_HD explicit DDPProblemBipedal(double dt = 0.01) : DDPProblem(dt) {}

  /** Min-jerk blend from (0,0) to (1,1) and its second derivative. */
  NMPC_HD static double minJerk(double t)
  {
    const double t3 = t * t * t;
    return 6 * (t3 * t * t) + -15 * (t3 * t) + 10 * t3;
  }
  NMPC_HD static double minJerkSecondDeriv(double t)
  {
    return 120 * (t * t * t) + -180 * (t * t) + 60 * t;
  }

  /** Reference ZMP: zero at both ends, alternating +-0.15 m every second in between. */
  NMPC_HD double refZmp(double t) const
  {
    t += 1e-6; // keeps switching instants on the same side as the reference's epsilon_t
    if(t <= 1.5 || t >= end_t_ - 1.5)
    {
      return 0.0;
    }
    const int step_idx = static_cast<int>(floor((t - 1.0) / 1.0));
    return (step_idx % 2 == 0) ? 0.15 : -0.15;
  }

  /** omega^2 = (z'' + g) / z for a CoM height that drops 1.0 -> 0.3 m over [7,8] s and rises back over [12,13] s. */
  NMPC_HD double omega2(double t) const
  {
    t += 1e-6;
    constexpr double z_high = 1.0;
    constexpr double z_low = 0.3;
    constexpr double g = 9.80665;
    double z = z_high;
    double zdd = 0.0;
    if(t >= 7.0 && t < 8.0)
    {
      z = (z_low - z_high) * minJerk(t - 7.0) + z_high;
      zdd = (z_low - z_high) * minJerkSecondDeriv(t - 7.0);
    }
    else if(t >= 8.0 && t < 12.0)
    {
      z = z_low;
    }
    else if(t >= 12.0 && t < 13.0)
    {
      z = (z_high - z_low) * minJerk(t - 12.0) + z_low;
      zdd = (z_high - z_low) * minJerkSecondDeriv(t - 12.0);
    }
    return (zdd + g) / z;
  }

  // A(t), B(t) of the discretised dynamics from omega^2(t).  The two share ONE evaluation of the schedule wherever both are needed
  // (stateEq, calcStateEqDeriv): omega2() is two polynomials, three range tests and an fp64 division — evaluated per matrix, as A(t)
  // and B(t) each do, the compiler kept both copies, ~45 of the ~290 instructions of a rollout timestep on the quad kernel's master
  // wave (round 6, BASELINE config 3).  Element for element the same expressions: the same bits.
  NMPC_HD StateStateDimMatrix Aof(double w2) const
  {
    StateStateDimMatrix A;
    A(0, 0) = 1 + 0.5 * dt_ * dt_ * w2;
    A(0, 1) = dt_;
    A(1, 0) = dt_ * w2;
    A(1, 1) = 1;
    return A;
  }

  NMPC_HD StateInputDimMatrix Bof(double w2) const
  {
    StateInputDimMatrix B;
    B(0, 0) = -0.5 * dt_ * dt_ * w2;
    B(1, 0) = -1 * dt_ * w2;
    return B;
  }

  NMPC_HD StateStateDimMatrix A(double t) const
  {
    return Aof(omega2(t));
  }

  NMPC_HD StateInputDimMatrix B(double t) const
  {
    return Bof(omega2(t));
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const double w2 = omega2(t);
    const StateStateDimMatrix a = Aof(w2);
    const StateInputDimMatrix b = Bof(w2);
    StateDimVector x_next;
    x_next[0] = (a(0, 0) * x[0] + a(0, 1) * x[1]) + b(0, 0) * u[0];
    x_next[1] = (a(1, 0) * x[0] + a(1, 1) * x[1]) + b(1, 0) * u[0];
    return x_next;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const double zmp_err = u[0] - refZmp(t);
    return cost_weight_.running_vel * 0.5 * (x[1] * x[1]) + cost_weight_.running_zmp * 0.5 * (zmp_err * zmp_err);
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    const double pos_err = x[0] - refZmp(t);
    return cost_weight_.terminal_pos * 0.5 * (pos_err * pos_err) + cost_weight_.terminal_vel * 0.5 * (x[1] * x[1]);
  }

  NMPC_HD void calcStateEqDeriv(double t,
                                const StateDimVector &, // x
                                const InputDimVector &, // u
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const double w2 = omega2(t);
    state_eq_deriv_x = Aof(w2);
    state_eq_deriv_u = Bof(w2);
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
    running_cost_deriv_x[0] = 0;
    running_cost_deriv_x[1] = cost_weight_.running_vel * x[1];
    running_cost_deriv_u[0] = cost_weight_.running_zmp * (u[0] - refZmp(t));
    running_cost_deriv_xx.setZero();
    running_cost_deriv_xx(1, 1) = cost_weight_.running_vel;
    running_cost_deriv_uu(0, 0) = cost_weight_.running_zmp;
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_x[0] = cost_weight_.terminal_pos * (x[0] - refZmp(t));
    terminal_cost_deriv_x[1] = cost_weight_.terminal_vel * x[1];
    terminal_cost_deriv_xx.setZero();
    terminal_cost_deriv_xx(0, 0) = cost_weight_.terminal_pos;
    terminal_cost_deriv_xx(1, 1) = cost_weight_.terminal_vel;
  }

public:
  CostWeight cost_weight_;
  double end_t_ = 20.0; // [sec] end of the walking schedule
};
} // namespace nmpc_amd
