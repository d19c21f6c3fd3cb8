// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/fmpc_oracle.hpp header).
//
// The two problems the reference's FMPC tests define, restated on flat column-major arrays:
//   Oscillator   FmpcProblemOscillator  nmpc_fmpc/tests/src/TestFmpcOscillator.cpp:18-135 (Van der Pol, n=2 m=1 g=3)
//   CartPole     FmpcProblemCartPole    nmpc_fmpc/tests/src/TestFmpcCartPole.cpp:32-267   (n=4 m=1 g=4)
// plus one problem of this repo with two inputs (so that the pivoted LDLT of G is exercised beyond 1 x 1):
//   PointMass    planar point mass with velocity-dependent drag, n=4 m=2 g=4 (no reference counterpart)
#pragma once

#include <cmath>

namespace oracle_fmpc
{
struct Oscillator
{
  static constexpr int N = 2, M = 1, G = 3;
  static constexpr const char * kName = "fmpc_oscillator";
  double dt = 0.01;

  void stateEqDt(double, const double * x, const double * u, double step, double * out) const // :27-37
  {
    const double x_dot0 = (1.0 - std::pow(x[1], 2)) * x[0] - x[1] + u[0];
    const double x_dot1 = x[0];
    out[0] = x[0] + step * x_dot0;
    out[1] = x[1] + step * x_dot1;
  }
  void stateEq(double t, const double * x, const double * u, double * out) const // :22-25
  {
    stateEqDt(t, x, u, dt, out);
  }
  double runningCost(double, const double * x, const double * u) const // :39-44
  {
    return 0.5 * ((x[0] * x[0] + x[1] * x[1]) + u[0] * u[0]);
  }
  double terminalCost(double, const double *) const // :46-51
  {
    return 0;
  }
  void ineqConst(double, const double * x, const double * u, double * g) const // :53-62
  {
    g[0] = -1 * x[1] - 0.05;
    g[1] = -1 * u[0] - 1.0;
    g[2] = u[0] - 0.9;
  }
  void calcStateEqDeriv(double, const double * x, const double *, double * A, double * B) const // :64-80
  {
    for(int i = 0; i < 4; i++)
    {
      A[i] = 0;
    }
    A[0 + 0 * 2] = 1.0 - std::pow(x[1], 2);
    A[0 + 1 * 2] = -2 * x[0] * x[1] - 1.0;
    A[1 + 0 * 2] = 1;
    for(int i = 0; i < 4; i++)
    {
      A[i] *= dt;
    }
    A[0] += 1;
    A[3] += 1;
    B[0] = 1 * dt;
    B[1] = 0 * dt;
  }
  void calcRunningCostDeriv(double, const double * x, const double * u, double * Lx, double * Lu, double * Lxx, double * Luu,
                            double * Lxu) const // :82-105
  {
    Lx[0] = x[0];
    Lx[1] = x[1];
    Lu[0] = u[0];
    Lxx[0] = 1;
    Lxx[1] = 0;
    Lxx[2] = 0;
    Lxx[3] = 1;
    Luu[0] = 1;
    Lxu[0] = 0;
    Lxu[1] = 0;
  }
  void calcTerminalCostDeriv(double, const double *, double * Lx, double * Lxx) const // :107-121
  {
    Lx[0] = Lx[1] = 0;
    Lxx[0] = Lxx[1] = Lxx[2] = Lxx[3] = 0;
  }
  void calcIneqConstDeriv(double, const double *, const double *, double * C, double * D) const // :123-134
  {
    for(int i = 0; i < 6; i++)
    {
      C[i] = 0;
    }
    C[0 + 1 * 3] = -1;
    D[0] = 0;
    D[1] = -1;
    D[2] = 1;
  }
};

struct CartPole
{
  static constexpr int N = 4, M = 1, G = 4;
  static constexpr const char * kName = "fmpc_cartpole";
  static constexpr double g_ = 9.80665;
  double dt = 0.01;
  double cart_mass = 1.0, pole_mass = 0.5, pole_length = 2.0; // :36-43
  double running_x[4] = {0.1, 1.0, 0.01, 0.1}; // :45-57
  double running_u[1] = {0.001};
  double terminal_x[4] = {0.1, 1.0, 0.01, 0.1};
  double ref_pos = 0.0; // ref_pos_func_ (:259): the test's getRefPos returns a constant between service calls (:393-406)
  double u_max = 15.0, x_max = 20.0; // :122-125 (constexpr there)

  void stateEqDt(double, const double * x, const double * u, double step, double * out) const // :73-103
  {
    const double theta = x[1], vel = x[2], omega = x[3], f = u[0];
    const double m1 = cart_mass, m2 = pole_mass, l = pole_length;
    const double sin_theta = std::sin(theta), cos_theta = std::cos(theta);
    const double omega2 = std::pow(omega, 2);
    const double denom = m1 + m2 * std::pow(sin_theta, 2);
    double x_dot[4];
    x_dot[0] = vel;
    x_dot[1] = omega;
    x_dot[2] = (f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta) / denom;
    x_dot[3] = (f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta) / (l * denom);
    for(int i = 0; i < 4; i++)
    {
      out[i] = x[i] + step * x_dot[i];
    }
  }
  void stateEq(double t, const double * x, const double * u, double * out) const // :68-71
  {
    stateEqDt(t, x, u, dt, out);
  }
  double runningCost(double, const double * x, const double * u) const // :105-110
  {
    double cx = 0;
    for(int i = 0; i < 4; i++)
    {
      const double e = x[i] - (i == 0 ? ref_pos : 0.0);
      cx += running_x[i] * (e * e);
    }
    return 0.5 * cx + 0.5 * (running_u[0] * (u[0] * u[0]));
  }
  double terminalCost(double, const double * x) const // :112-117
  {
    double cx = 0;
    for(int i = 0; i < 4; i++)
    {
      const double e = x[i] - (i == 0 ? ref_pos : 0.0);
      cx += terminal_x[i] * (e * e);
    }
    return 0.5 * cx;
  }
  void ineqConst(double, const double * x, const double * u, double * g) const // :119-133
  {
    const double u_min = -1 * u_max, x_min = -1 * x_max;
    g[0] = -1 * u[0] + u_min;
    g[1] = u[0] - u_max;
    g[2] = -1 * x[0] + x_min;
    g[3] = x[0] - x_max;
  }
  void calcStateEqDeriv(double, const double * x, const double * u, double * A, double * B) const // :135-178
  {
    const double theta = x[1], omega = x[3], f = u[0];
    const double m1 = cart_mass, m2 = pole_mass, l = pole_length;
    const double sin_theta = std::sin(theta), cos_theta = std::cos(theta);
    const double omega2 = std::pow(omega, 2);
    const double denom = m1 + m2 * std::pow(sin_theta, 2);
    for(int i = 0; i < 16; i++)
    {
      A[i] = 0;
    }
    auto a = [&](int r, int c) -> double & { return A[r + c * 4]; };
    a(0, 2) = 1;
    a(1, 3) = 1;
    a(2, 1) = ((-1 * m2 * l * omega2 * cos_theta + m2 * g_ * (1 - 2 * std::pow(sin_theta, 2))) * denom
               + -1 * (f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta) * (2 * m2 * sin_theta * cos_theta))
              / std::pow(denom, 2);
    a(2, 3) = (-2 * m2 * l * omega * sin_theta) / denom;
    a(3, 1) = ((-1 * f * sin_theta + -1 * m2 * l * omega2 * (1 - 2 * std::pow(sin_theta, 2)) + g_ * (m1 + m2) * cos_theta)
                   * denom
               + -1 * (f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta)
                     * (2 * m2 * sin_theta * cos_theta))
              / (l * std::pow(denom, 2));
    a(3, 3) = (-2 * m2 * l * omega * sin_theta * cos_theta) / (l * denom);
    for(int i = 0; i < 16; i++)
    {
      A[i] *= dt;
    }
    for(int i = 0; i < 4; i++)
    {
      a(i, i) += 1.0;
    }
    B[0] = 0;
    B[1] = 0;
    B[2] = (1 / denom) * dt;
    B[3] = (cos_theta / (l * denom)) * dt;
  }
  void calcRunningCostDeriv(double, const double * x, const double * u, double * Lx, double * Lu, double * Lxx, double * Luu,
                            double * Lxu) const // :195-214
  {
    for(int i = 0; i < 16; i++)
    {
      Lxx[i] = 0;
    }
    for(int i = 0; i < 4; i++)
    {
      Lx[i] = running_x[i] * (x[i] - (i == 0 ? ref_pos : 0.0));
      Lxx[i + i * 4] = running_x[i];
      Lxu[i] = 0;
    }
    Lu[0] = running_u[0] * u[0];
    Luu[0] = running_u[0];
  }
  void calcTerminalCostDeriv(double, const double * x, double * Lx, double * Lxx) const // :227-238
  {
    for(int i = 0; i < 16; i++)
    {
      Lxx[i] = 0;
    }
    for(int i = 0; i < 4; i++)
    {
      Lx[i] = terminal_x[i] * (x[i] - (i == 0 ? ref_pos : 0.0));
      Lxx[i + i * 4] = terminal_x[i];
    }
  }
  void calcIneqConstDeriv(double, const double *, const double *, double * C, double * D) const // :240-254
  {
    for(int i = 0; i < 16; i++)
    {
      C[i] = 0;
    }
    C[2 + 0 * 4] = -1;
    C[3 + 0 * 4] = 1;
    D[0] = -1;
    D[1] = 1;
    D[2] = 0;
    D[3] = 0;
  }
};

/** Planar point mass, state [px, py, vx, vy], input [fx, fy]; quadratic drag couples the axes so that the dynamics are
    nonlinear; box limits on both inputs (g = 4).  The running cost couples the two inputs (Luu is full), so that G of the
    Riccati step is a full 2 x 2 matrix. */
struct PointMass
{
  static constexpr int N = 4, M = 2, G = 4;
  static constexpr const char * kName = "fmpc_pointmass";
  double dt = 0.02;
  double mass = 1.5, drag = 0.3;
  double target[2] = {1.0, -0.5};
  double w_pos = 2.0, w_vel = 0.2, w_u = 0.05, w_u_cross = 0.02, w_term = 5.0;
  double u_max[2] = {2.0, 1.0};

  void stateEqDt(double, const double * x, const double * u, double step, double * out) const
  {
    const double speed = std::sqrt(x[2] * x[2] + x[3] * x[3] + 1e-6);
    out[0] = x[0] + step * x[2];
    out[1] = x[1] + step * x[3];
    out[2] = x[2] + step * ((u[0] - drag * speed * x[2]) / mass);
    out[3] = x[3] + step * ((u[1] - drag * speed * x[3]) / mass);
  }
  void stateEq(double t, const double * x, const double * u, double * out) const
  {
    stateEqDt(t, x, u, dt, out);
  }
  double runningCost(double, const double * x, const double * u) const
  {
    const double ex = x[0] - target[0], ey = x[1] - target[1];
    return 0.5 * (w_pos * (ex * ex + ey * ey) + w_vel * (x[2] * x[2] + x[3] * x[3]) + w_u * (u[0] * u[0] + u[1] * u[1]))
           + w_u_cross * (u[0] * u[1]);
  }
  double terminalCost(double, const double * x) const
  {
    const double ex = x[0] - target[0], ey = x[1] - target[1];
    return 0.5 * w_term * ((ex * ex + ey * ey) + (x[2] * x[2] + x[3] * x[3]));
  }
  void ineqConst(double, const double *, const double * u, double * g) const
  {
    g[0] = -1 * u[0] - u_max[0];
    g[1] = u[0] - u_max[0];
    g[2] = -1 * u[1] - u_max[1];
    g[3] = u[1] - u_max[1];
  }
  void calcStateEqDeriv(double, const double * x, const double *, double * A, double * B) const
  {
    const double speed = std::sqrt(x[2] * x[2] + x[3] * x[3] + 1e-6);
    const double c = drag / mass;
    for(int i = 0; i < 16; i++)
    {
      A[i] = 0;
    }
    auto a = [&](int r, int col) -> double & { return A[r + col * 4]; };
    a(0, 2) = dt;
    a(1, 3) = dt;
    a(2, 2) = -dt * c * (speed + x[2] * x[2] / speed);
    a(2, 3) = -dt * c * (x[2] * x[3] / speed);
    a(3, 2) = -dt * c * (x[2] * x[3] / speed);
    a(3, 3) = -dt * c * (speed + x[3] * x[3] / speed);
    for(int i = 0; i < 4; i++)
    {
      a(i, i) += 1.0;
    }
    for(int i = 0; i < 8; i++)
    {
      B[i] = 0;
    }
    B[2 + 0 * 4] = dt / mass;
    B[3 + 1 * 4] = dt / mass;
  }
  void calcRunningCostDeriv(double, const double * x, const double * u, double * Lx, double * Lu, double * Lxx, double * Luu,
                            double * Lxu) const
  {
    for(int i = 0; i < 16; i++)
    {
      Lxx[i] = 0;
    }
    for(int i = 0; i < 8; i++)
    {
      Lxu[i] = 0;
    }
    Lx[0] = w_pos * (x[0] - target[0]);
    Lx[1] = w_pos * (x[1] - target[1]);
    Lx[2] = w_vel * x[2];
    Lx[3] = w_vel * x[3];
    Lxx[0] = w_pos;
    Lxx[5] = w_pos;
    Lxx[10] = w_vel;
    Lxx[15] = w_vel;
    Lu[0] = w_u * u[0] + w_u_cross * u[1];
    Lu[1] = w_u * u[1] + w_u_cross * u[0];
    Luu[0] = w_u;
    Luu[1] = w_u_cross;
    Luu[2] = w_u_cross;
    Luu[3] = w_u;
  }
  void calcTerminalCostDeriv(double, const double * x, double * Lx, double * Lxx) const
  {
    for(int i = 0; i < 16; i++)
    {
      Lxx[i] = 0;
    }
    Lx[0] = w_term * (x[0] - target[0]);
    Lx[1] = w_term * (x[1] - target[1]);
    Lx[2] = w_term * x[2];
    Lx[3] = w_term * x[3];
    for(int i = 0; i < 4; i++)
    {
      Lxx[i + i * 4] = w_term;
    }
  }
  void calcIneqConstDeriv(double, const double *, const double *, double * C, double * D) const
  {
    for(int i = 0; i < 16; i++)
    {
      C[i] = 0;
    }
    for(int i = 0; i < 8; i++)
    {
      D[i] = 0;
    }
    D[0 + 0 * 4] = -1;
    D[1 + 0 * 4] = 1;
    D[2 + 1 * 4] = -1;
    D[3 + 1 * 4] = 1;
  }
};
} // namespace oracle_fmpc
