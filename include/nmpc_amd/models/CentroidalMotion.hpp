// Centroidal-motion problem (bilinear dynamics, input dimension 16 or 0) for the MI355X DDP solver.
// Same model and schedules as the reference's test problem DDPProblemCentroidalMotion
// (nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp:24-204 model, :206-237 stance construction, :249-281
// stance / reference schedules): state [CoM(3), linear momentum(3), angular momentum(3)], input = force
// scale along each friction-pyramid ridge of each contact vertex.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
class DDPProblemCentroidalMotion : public DDPProblem<9, Dynamic, 16>
{
public:
  static constexpr const char * kName = "centroidal";
  static constexpr int kRidgeNum = 16;

  /** Contact vertices and force directions (friction-pyramid ridges), one column per input. */
  struct StanceData
  {
    int num = 0;
    double vertices[3][kRidgeNum];
    double ridges[3][kRidgeNum];
  };

  NMPC_HD explicit DDPProblemCentroidalMotion(double dt = 0.03) : DDPProblem(dt) {}

  /** Rectangular foot [min_x, min_y, max_x, max_y] on the ground: 4 vertices x 4 ridges tilted by
      atan(0.5) around the vertical. */
  NMPC_HD static StanceData makeStanceDataFromRect(const double * rect)
  {
    StanceData s;
    s.num = kRidgeNum;
    const double vertex_x[4] = {rect[0], rect[0], rect[2], rect[2]};
    const double vertex_y[4] = {rect[1], rect[3], rect[3], rect[1]};
    // The four ridge directions (0.5 cos(theta), 0.5 sin(theta), 1) / |.| with theta = 2 pi ri / 4 do not depend on the foot:
    // they are the constants the reference's expression evaluates to in double (tests/test_host_cpu.py checks them against
    // libm).  Evaluated per call — every stateEq and every derivative builds its stance — they were sixteen cos / sin /
    // sqrt / divisions of the device math library: ~12 k of the 15 k cycles of a rollout timestep on the tile kernel.
    constexpr double kRidgeDir[4][3] = {{0x1.c9f25c5bfedd9p-2, 0x0p+0, 0x1.c9f25c5bfedd9p-1},
                                        {0x1.f924f9f58fd97p-56, 0x1.c9f25c5bfedd9p-2, 0x1.c9f25c5bfedd9p-1},
                                        {-0x1.c9f25c5bfedd9p-2, 0x1.f924f9f58fd97p-55, 0x1.c9f25c5bfedd9p-1},
                                        {-0x1.7adbbb782be31p-54, -0x1.c9f25c5bfedd9p-2, 0x1.c9f25c5bfedd9p-1}};
    for(int vi = 0; vi < 4; vi++)
    {
      for(int ri = 0; ri < 4; ri++)
      {
        const int col = vi * 4 + ri;
        s.vertices[0][col] = vertex_x[vi];
        s.vertices[1][col] = vertex_y[vi];
        s.vertices[2][col] = 0.0;
        s.ridges[0][col] = kRidgeDir[ri][0];
        s.ridges[1][col] = kRidgeDir[ri][1];
        s.ridges[2][col] = kRidgeDir[ri][2];
      }
    }
    return s;
  }

  /** Stance schedule: first foothold, flight phase (no input), second foothold. */
  NMPC_HD StanceData refStance(double t) const
  {
    t += 1e-6;
    if(t < flight_start_t_)
    {
      const double rect1[4] = {-0.1, -0.1, 0.1, 0.1};
      return makeStanceDataFromRect(rect1);
    }
    if(t < flight_end_t_)
    {
      StanceData s;
      s.num = 0;
      return s;
    }
    return makeStanceDataFromRect(second_rect_);
  }

  NMPC_HD void refPos(double t, double * pos) const
  {
    t += 1e-6;
    pos[0] = (t < ref_switch_t_) ? 0.0 : 0.5;
    pos[1] = 0.0;
    pos[2] = 1.0;
  }

  NMPC_HD int inputDim(double t) const
  {
    t += 1e-6;
    return (t >= flight_start_t_ && t < flight_end_t_) ? 0 : kRidgeNum;
  }

  NMPC_HD double weight(int i) const
  {
    return (i >= 3 && i < 6) ? weight_lin_ : weight_pos_ang_;
  }

  NMPC_HD static void cross(const double * a, const double * b, double * c)
  {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const StanceData stance = refStance(t);
    double x_dot[9];
    for(int c = 0; c < 3; c++)
    {
      x_dot[c] = x[3 + c] / mass_;
      double force = 0;
      for(int i = 0; i < u.size(); i++)
      {
        force += stance.ridges[c][i] * u[i];
      }
      x_dot[3 + c] = force - mass_ * (c == 2 ? g_ : 0.0);
      x_dot[6 + c] = 0;
    }
    for(int i = 0; i < u.size(); i++)
    {
      const double arm[3] = {stance.vertices[0][i] - x[0], stance.vertices[1][i] - x[1], stance.vertices[2][i] - x[2]};
      const double ridge[3] = {stance.ridges[0][i], stance.ridges[1][i], stance.ridges[2][i]};
      double moment[3];
      cross(arm, ridge, moment);
      for(int c = 0; c < 3; c++)
      {
        x_dot[6 + c] += u[i] * moment[c];
      }
    }
    StateDimVector x_next;
    for(int i = 0; i < 9; i++)
    {
      x_next[i] = x[i] + dt_ * x_dot[i];
    }
    return x_next;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    double ref_pos[3];
    refPos(t, ref_pos);
    double cost_x = 0;
    for(int i = 0; i < 9; i++)
    {
      const double e = (i < 3) ? x[i] - ref_pos[i] : x[i];
      cost_x += weight(i) * (e * e);
    }
    return 0.5 * cost_x + 0.5 * running_u_ * u.squaredNorm();
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    double ref_pos[3];
    refPos(t, ref_pos);
    double cost_x = 0;
    for(int i = 0; i < 9; i++)
    {
      const double e = (i < 3) ? x[i] - ref_pos[i] : x[i];
      cost_x += weight(i) * (e * e);
    }
    return 0.5 * cost_x;
  }

  NMPC_HD void calcStateEqDeriv(double t,
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const StanceData stance = refStance(t);

    double force[3] = {0, 0, 0};
    for(int c = 0; c < 3; c++)
    {
      for(int i = 0; i < u.size(); i++)
      {
        force[c] += stance.ridges[c][i] * u[i];
      }
    }
    state_eq_deriv_x.setZero();
    for(int c = 0; c < 3; c++)
    {
      state_eq_deriv_x(c, 3 + c) = 1 / mass_;
    }
    // d(angular momentum rate)/d(CoM) = skew(total force)
    state_eq_deriv_x(6, 1) = -force[2];
    state_eq_deriv_x(6, 2) = force[1];
    state_eq_deriv_x(7, 0) = force[2];
    state_eq_deriv_x(7, 2) = -force[0];
    state_eq_deriv_x(8, 0) = -force[1];
    state_eq_deriv_x(8, 1) = force[0];
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.addToDiagonal(1.0);

    state_eq_deriv_u.resize(9, u.size());
    state_eq_deriv_u.setZero();
    for(int i = 0; i < u.size(); i++)
    {
      const double arm[3] = {stance.vertices[0][i] - x[0], stance.vertices[1][i] - x[1], stance.vertices[2][i] - x[2]};
      const double ridge[3] = {stance.ridges[0][i], stance.ridges[1][i], stance.ridges[2][i]};
      double moment[3];
      cross(arm, ridge, moment);
      for(int c = 0; c < 3; c++)
      {
        state_eq_deriv_u(3 + c, i) = ridge[c] * dt_;
        state_eq_deriv_u(6 + c, i) = moment[c] * dt_;
      }
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
    double ref_pos[3];
    refPos(t, ref_pos);
    running_cost_deriv_xx.setZero();
    for(int i = 0; i < 9; i++)
    {
      const double e = (i < 3) ? x[i] - ref_pos[i] : x[i];
      running_cost_deriv_x[i] = weight(i) * e;
      running_cost_deriv_xx(i, i) = weight(i);
    }
    running_cost_deriv_u.resize(u.size());
    running_cost_deriv_uu.resize(u.size(), u.size());
    running_cost_deriv_uu.setZero();
    for(int i = 0; i < u.size(); i++)
    {
      running_cost_deriv_u[i] = running_u_ * u[i];
      running_cost_deriv_uu(i, i) = 1.0 * running_u_;
    }
    running_cost_deriv_xu.resize(9, u.size());
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    double ref_pos[3];
    refPos(t, ref_pos);
    terminal_cost_deriv_xx.setZero();
    for(int i = 0; i < 9; i++)
    {
      const double e = (i < 3) ? x[i] - ref_pos[i] : x[i];
      terminal_cost_deriv_x[i] = weight(i) * e;
      terminal_cost_deriv_xx(i, i) = weight(i);
    }
  }

public:
  static constexpr double g_ = 9.80665; // [m/s^2]
  double running_u_ = 1e-6;
  double mass_ = 100.0; // [kg]
  double flight_start_t_ = 1.4; // [sec]
  double flight_end_t_ = 1.6; // [sec]
  double ref_switch_t_ = 1.5; // [sec]
  double weight_pos_ang_ = 1.0; // running = terminal weight on CoM position and angular momentum
  double weight_lin_ = 0.0; // running = terminal weight on linear momentum
  double second_rect_[4] = {0.4, -0.1, 0.6, 0.1};
};
} // namespace nmpc_amd
