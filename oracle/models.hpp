// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/ddp_oracle.hpp header).
//
// CPU restatements of the DDPProblem subclasses that the reference's tests define, on raw column-major
// arrays.  Each struct exposes the DDPProblem interface (nmpc_ddp/include/nmpc_ddp/DDPProblem.h:99-198):
//   inputDim(t), stateEq, runningCost, terminalCost, calcStateEqDeriv (1st order), calcRunningCostDeriv
//   (2nd order), calcTerminalCostDeriv (2nd order).
// Time-varying references that the reference passes as std::function are closed-form functions of t here,
// with the same epsilon_t = 1e-6 offset.
//
// Quadrotor and Manipulator are builder-defined (no reference model exists, SURVEY.md §8 d C4/C5); the
// oracle versions below are the independent CPU statement the HIP path is checked against.
#pragma once

#include "model_cartpole.hpp" // CartPole (written against Real: also instantiated in float)
#include <cmath>

namespace oracle
{

// ---------------------------------------------------------------------------------------------------
// Bipedal CoM-ZMP (LTV): state [com_pos, com_vel], input [zmp].  nmpc_ddp/tests/src/TestDDPBipedal.cpp:16-144
// ref_zmp(t) and omega2(t) are the schedules of TestCase1 (:171-225).
// ---------------------------------------------------------------------------------------------------
struct Bipedal
{
  using Real = double;
  static constexpr int N = 2;
  static constexpr int MMAX = 1;
  static constexpr int NPARAM = 6;

  double dt = 0.01;
  double running_vel = 1e-14; // :24
  double running_zmp = 1e-1; // :25
  double terminal_pos = 1e2; // :26
  double terminal_vel = 1.0; // :27
  double end_t = 20.0; // :167

  void setParams(const double * p)
  {
    dt = p[0];
    running_vel = p[1];
    running_zmp = p[2];
    terminal_pos = p[3];
    terminal_vel = p[4];
    end_t = p[5];
  }

  static double minJerk(double t) // :151-154
  {
    return 6 * std::pow(t, 5) + -15 * std::pow(t, 4) + 10 * std::pow(t, 3);
  }
  static double minJerkSecondDeriv(double t) // :156-159
  {
    return 120 * std::pow(t, 3) + -180 * std::pow(t, 2) + 60 * t;
  }

  double refZmp(double t) const // :171-191
  {
    t += 1e-6;
    if(t <= 1.5 || t >= end_t - 1.5)
    {
      return 0.0;
    }
    if(static_cast<int>(std::floor((t - 1.0) / 1.0)) % 2 == 0)
    {
      return 0.15;
    }
    return -0.15;
  }

  double omega2(double t) const // :192-225
  {
    t += 1e-6;
    const double z_high = 1.0, z_low = 0.3;
    double z = 0.0, acc = 0.0;
    if(t < 7.0)
    {
      z = z_high;
    }
    else if(t < 8.0)
    {
      double scale = z_low - z_high;
      z = scale * minJerk(t - 7.0) + z_high;
      acc = scale * minJerkSecondDeriv(t - 7.0);
    }
    else if(t < 12.0)
    {
      z = z_low;
    }
    else if(t < 13.0)
    {
      double scale = z_high - z_low;
      z = scale * minJerk(t - 12.0) + z_low;
      acc = scale * minJerkSecondDeriv(t - 12.0);
    }
    else
    {
      z = z_high;
    }
    return (acc + 9.80665) / z;
  }

  int inputDim(double) const
  {
    return 1;
  }

  // A(t), B(t)    :124-138
  void AB(double t, double * A, double * B) const
  {
    const double w2 = omega2(t);
    A[0] = 1 + 0.5 * dt * dt * w2; // (0,0)
    A[2] = dt; // (0,1)
    A[1] = dt * w2; // (1,0)
    A[3] = 1; // (1,1)
    B[0] = -0.5 * dt * dt * w2;
    B[1] = -1 * dt * w2;
  }

  void stateEq(double t, const double * x, const double * u, int, double * xn) const // :38-41
  {
    double A[4], B[2];
    AB(t, A, B);
    for(int r = 0; r < 2; r++)
    {
      xn[r] = (A[r] * x[0] + A[r + 2] * x[1]) + B[r] * u[0];
    }
  }

  double runningCost(double t, const double * x, const double * u, int) const // :43-47
  {
    double dz = u[0] - refZmp(t);
    return running_vel * 0.5 * (x[1] * x[1]) + running_zmp * 0.5 * (dz * dz);
  }

  double terminalCost(double t, const double * x) const // :49-53
  {
    double dp = x[0] - refZmp(t);
    return terminal_pos * 0.5 * (dp * dp) + terminal_vel * 0.5 * (x[1] * x[1]);
  }

  void calcStateEqDeriv(double t, const double *, const double *, int, double * Fx, double * Fu) const // :55-63
  {
    AB(t, Fx, Fu);
  }

  void calcRunningCostDeriv(double t,
                            const double * x,
                            const double * u,
                            int,
                            double * Lx,
                            double * Lu,
                            double * Lxx,
                            double * Luu,
                            double * Lxu) const // :79-103
  {
    Lx[0] = 0;
    Lx[1] = running_vel * x[1];
    Lu[0] = running_zmp * (u[0] - refZmp(t));
    Lxx[0] = 0;
    Lxx[1] = 0;
    Lxx[2] = 0;
    Lxx[3] = running_vel;
    Luu[0] = running_zmp;
    Lxu[0] = 0;
    Lxu[1] = 0;
  }

  void calcTerminalCostDeriv(double t, const double * x, double * Vx, double * Vxx) const // :105-121
  {
    Vx[0] = terminal_pos * (x[0] - refZmp(t));
    Vx[1] = terminal_vel * x[1];
    Vxx[0] = terminal_pos;
    Vxx[1] = 0;
    Vxx[2] = 0;
    Vxx[3] = terminal_vel;
  }
};

// ---------------------------------------------------------------------------------------------------
// Vertical motion: state [pos_z, vel_z], input [force_z ...] with nu(t) in {1, 2, 0}.
// nmpc_ddp/tests/src/TestDDPVerticalMotion.cpp:31-234; ref_pos schedule :246-259.
// (use_smooth_abs_ is false in the reference, :233, so only the quadratic input cost is restated.)
// ---------------------------------------------------------------------------------------------------
struct VerticalMotion
{
  using Real = double;
  static constexpr int N = 2;
  static constexpr int MMAX = 2;
  static constexpr int NPARAM = 8;

  double dt = 0.01;
  double running_x[2] = {1.0, 1e-3}; // :39
  double running_u = 1e-4; // :40
  double terminal_x[2] = {1.0, 1e-3}; // :41
  double mass = 1.0; // :228
  double ref_switch_t = 8.0; // :250
  static constexpr double g = 9.80665; // :225

  void setParams(const double * p)
  {
    dt = p[0];
    running_x[0] = p[1];
    running_x[1] = p[2];
    running_u = p[3];
    terminal_x[0] = p[4];
    terminal_x[1] = p[5];
    mass = p[6];
    ref_switch_t = p[7];
  }

  double refPos(double t) const // :246-259
  {
    t += 1e-6;
    return t < ref_switch_t ? 1.0 : 0.0;
  }

  int inputDim(double t) const // :58-75
  {
    t += 1e-6;
    if(2.0 < t && t < 3.0)
    {
      return 2;
    }
    else if(4.5 < t && t < 5.0)
    {
      return 0;
    }
    return 1;
  }

  void stateEq(double, const double * x, const double * u, int m, double * xn) const // :77-84
  {
    double usum = 0;
    for(int a = 0; a < m; a++)
    {
      usum += u[a];
    }
    xn[0] = x[0] + dt * x[1];
    xn[1] = x[1] + dt * (usum / mass - g);
  }

  double runningCost(double t, const double * x, const double * u, int m) const // :86-100
  {
    double d0 = x[0] - refPos(t), d1 = x[1] - 0;
    double cost_x = 0.5 * (running_x[0] * (d0 * d0) + running_x[1] * (d1 * d1));
    double un = 0;
    for(int a = 0; a < m; a++)
    {
      un += u[a] * u[a];
    }
    double cost_u = 0.5 * running_u * un;
    return cost_x + cost_u;
  }

  double terminalCost(double t, const double * x) const // :102-107
  {
    double d0 = x[0] - refPos(t), d1 = x[1] - 0;
    return 0.5 * (terminal_x[0] * (d0 * d0) + terminal_x[1] * (d1 * d1));
  }

  void calcStateEqDeriv(double, const double *, const double *, int m, double * Fx, double * Fu) const // :109-122
  {
    Fx[0] = 0 * dt + 1.0;
    Fx[1] = 0 * dt;
    Fx[2] = 1 * dt;
    Fx[3] = 0 * dt + 1.0;
    for(int a = 0; a < m; a++)
    {
      Fu[0 + a * 2] = 0 * dt;
      Fu[1 + a * 2] = (1.0 / mass) * dt;
    }
  }

  void calcRunningCostDeriv(double t,
                            const double * x,
                            const double * u,
                            int m,
                            double * Lx,
                            double * Lu,
                            double * Lxx,
                            double * Luu,
                            double * Lxu) const // :171-199
  {
    Lx[0] = running_x[0] * (x[0] - refPos(t));
    Lx[1] = running_x[1] * (x[1] - 0);
    Lxx[0] = running_x[0];
    Lxx[1] = 0;
    Lxx[2] = 0;
    Lxx[3] = running_x[1];
    for(int e = 0; e < 2 * m; e++)
    {
      Lxu[e] = 0;
    }
    for(int a = 0; a < m; a++)
    {
      Lu[a] = running_u * u[a];
    }
    for(int e = 0; e < m * m; e++)
    {
      Luu[e] = 0;
    }
    for(int a = 0; a < m; a++)
    {
      Luu[a + a * m] = 1.0 * running_u;
    }
  }

  void calcTerminalCostDeriv(double t, const double * x, double * Vx, double * Vxx) const // :211-222
  {
    Vx[0] = terminal_x[0] * (x[0] - refPos(t));
    Vx[1] = terminal_x[1] * (x[1] - 0);
    Vxx[0] = terminal_x[0];
    Vxx[1] = 0;
    Vxx[2] = 0;
    Vxx[3] = terminal_x[1];
  }
};

// ---------------------------------------------------------------------------------------------------
// Centroidal motion: state [com(3), linear momentum(3), angular momentum(3)], input = 16 ridge force
// scales or 0 in flight.  nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp:24-237; stance/ref schedules :249-281.
// ---------------------------------------------------------------------------------------------------
struct CentroidalMotion
{
  using Real = double;
  static constexpr int N = 9;
  static constexpr int MMAX = 16;
  static constexpr int NPARAM = 12;

  double dt = 0.03;
  double running_x[9] = {1, 1, 1, 0, 0, 0, 1, 1, 1}; // :43
  double running_u = 1e-6; // :44
  double terminal_x[9] = {1, 1, 1, 0, 0, 0, 1, 1, 1}; // :45
  double mass = 100.0; // :203
  // schedule of SolveMpc (:249-281); CheckDerivative uses a constant stance / reference (set flight_t0 > all t)
  double flight_t0 = 1.4, flight_t1 = 1.6; // :254,258
  double rect1[4] = {-0.1, -0.1, 0.1, 0.1}; // :256
  double rect2[4] = {0.4, -0.1, 0.6, 0.1}; // :267
  double ref_switch_t = 1.5; // :274
  static constexpr double gz = 9.80665; // :199

  void setParams(const double * p)
  {
    dt = p[0];
    running_u = p[1];
    mass = p[2];
    flight_t0 = p[3];
    flight_t1 = p[4];
    ref_switch_t = p[5];
    // p[6] : weight on com / angular momentum (1.0), p[7] : weight on linear momentum (0.0)
    for(int i = 0; i < 9; i++)
    {
      double w = (i >= 3 && i < 6) ? p[7] : p[6];
      running_x[i] = w;
      terminal_x[i] = w;
    }
    rect2[0] = p[8];
    rect2[1] = p[9];
    rect2[2] = p[10];
    rect2[3] = p[11];
  }

  /** makeStanceDataFromRect (:206-237): 4 vertices x 4 pyramid ridges.  Returns the number of columns. */
  static int stanceFromRect(const double * r, double * vertices /*3x16*/, double * ridges /*3x16*/)
  {
    const double vx[4] = {r[0], r[0], r[2], r[2]};
    const double vy[4] = {r[1], r[3], r[3], r[1]};
    double rid[4][3];
    for(int i = 0; i < 4; i++)
    {
      double theta = 2 * M_PI * (static_cast<double>(i) / 4);
      double v[3] = {0.5 * std::cos(theta), 0.5 * std::sin(theta), 1};
      double nrm = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
      for(int c = 0; c < 3; c++)
      {
        rid[i][c] = v[c] / nrm;
      }
    }
    int col = 0;
    for(int vi = 0; vi < 4; vi++)
    {
      for(int ri = 0; ri < 4; ri++)
      {
        vertices[0 + col * 3] = vx[vi];
        vertices[1 + col * 3] = vy[vi];
        vertices[2 + col * 3] = 0.0;
        for(int c = 0; c < 3; c++)
        {
          ridges[c + col * 3] = rid[ri][c];
        }
        col++;
      }
    }
    return 16;
  }

  int stance(double t, double * vertices, double * ridges) const // :249-270
  {
    t += 1e-6;
    if(t < flight_t0)
    {
      return stanceFromRect(rect1, vertices, ridges);
    }
    else if(t < flight_t1)
    {
      return 0;
    }
    return stanceFromRect(rect2, vertices, ridges);
  }

  void refPos(double t, double * r) const // :271-281
  {
    t += 1e-6;
    r[0] = (t < ref_switch_t) ? 0.0 : 0.5;
    r[1] = 0.0;
    r[2] = 1.0;
  }

  int inputDim(double t) const // :64-68
  {
    double V[48], R[48];
    return stance(t, V, R);
  }

  static void cross(const double * a, const double * b, double * c)
  {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
  }

  void stateEq(double t, const double * x, const double * u, int m, double * xn) const // :70-93
  {
    double V[48], R[48];
    stance(t, V, R);
    double xd[9];
    for(int c = 0; c < 3; c++)
    {
      xd[c] = x[3 + c] / mass;
    }
    for(int c = 0; c < 3; c++)
    {
      double s = 0;
      for(int a = 0; a < m; a++)
      {
        s += R[c + a * 3] * u[a];
      }
      xd[3 + c] = s - mass * (c == 2 ? gz : 0.0);
    }
    xd[6] = xd[7] = xd[8] = 0;
    for(int a = 0; a < m; a++)
    {
      double d[3] = {V[0 + a * 3] - x[0], V[1 + a * 3] - x[1], V[2 + a * 3] - x[2]};
      double cr[3];
      cross(d, &R[a * 3], cr);
      for(int c = 0; c < 3; c++)
      {
        xd[6 + c] += u[a] * cr[c];
      }
    }
    for(int i = 0; i < 9; i++)
    {
      xn[i] = x[i] + dt * xd[i];
    }
  }

  double runningCost(double t, const double * x, const double * u, int m) const // :95-100
  {
    double r[3];
    refPos(t, r);
    double s = 0;
    for(int i = 0; i < 9; i++)
    {
      double d = (i < 3) ? x[i] - r[i] : x[i];
      s += running_x[i] * (d * d);
    }
    double un = 0;
    for(int a = 0; a < m; a++)
    {
      un += u[a] * u[a];
    }
    return 0.5 * s + 0.5 * running_u * un;
  }

  double terminalCost(double t, const double * x) const // :102-107
  {
    double r[3];
    refPos(t, r);
    double s = 0;
    for(int i = 0; i < 9; i++)
    {
      double d = (i < 3) ? x[i] - r[i] : x[i];
      s += terminal_x[i] * (d * d);
    }
    return 0.5 * s;
  }

  void calcStateEqDeriv(double t, const double * x, const double * u, int m, double * Fx, double * Fu) const // :109-136
  {
    double V[48], R[48];
    stance(t, V, R);
    for(int e = 0; e < 81; e++)
    {
      Fx[e] = 0;
    }
    for(int c = 0; c < 3; c++)
    {
      Fx[c + (3 + c) * 9] = 1 / mass;
    }
    // block(6,0) = crossMat(ridges * u)
    double f[3] = {0, 0, 0};
    for(int c = 0; c < 3; c++)
    {
      double s = 0;
      for(int a = 0; a < m; a++)
      {
        s += R[c + a * 3] * u[a];
      }
      f[c] = s;
    }
    Fx[6 + 0 * 9] = 0;
    Fx[6 + 1 * 9] = -f[2];
    Fx[6 + 2 * 9] = f[1];
    Fx[7 + 0 * 9] = f[2];
    Fx[7 + 1 * 9] = 0;
    Fx[7 + 2 * 9] = -f[0];
    Fx[8 + 0 * 9] = -f[1];
    Fx[8 + 1 * 9] = f[0];
    Fx[8 + 2 * 9] = 0;
    for(int e = 0; e < 81; e++)
    {
      Fx[e] *= dt;
    }
    for(int i = 0; i < 9; i++)
    {
      Fx[i + i * 9] += 1.0;
    }
    for(int e = 0; e < 9 * m; e++)
    {
      Fu[e] = 0;
    }
    for(int a = 0; a < m; a++)
    {
      for(int c = 0; c < 3; c++)
      {
        Fu[3 + c + a * 9] = R[c + a * 3];
      }
      double d[3] = {V[0 + a * 3] - x[0], V[1 + a * 3] - x[1], V[2 + a * 3] - x[2]};
      double cr[3];
      cross(d, &R[a * 3], cr);
      for(int c = 0; c < 3; c++)
      {
        Fu[6 + c + a * 9] = cr[c];
      }
    }
    for(int e = 0; e < 9 * m; e++)
    {
      Fu[e] *= dt;
    }
  }

  void calcRunningCostDeriv(double t,
                            const double * x,
                            const double * u,
                            int m,
                            double * Lx,
                            double * Lu,
                            double * Lxx,
                            double * Luu,
                            double * Lxu) const // :150-178
  {
    double r[3];
    refPos(t, r);
    for(int i = 0; i < 9; i++)
    {
      double d = (i < 3) ? x[i] - r[i] : x[i];
      Lx[i] = running_x[i] * d;
    }
    for(int a = 0; a < m; a++)
    {
      Lu[a] = running_u * u[a];
    }
    for(int e = 0; e < 81; e++)
    {
      Lxx[e] = 0;
    }
    for(int i = 0; i < 9; i++)
    {
      Lxx[i + i * 9] = running_x[i];
    }
    for(int e = 0; e < m * m; e++)
    {
      Luu[e] = 0;
    }
    for(int a = 0; a < m; a++)
    {
      Luu[a + a * m] = 1.0 * running_u;
    }
    for(int e = 0; e < 9 * m; e++)
    {
      Lxu[e] = 0;
    }
  }

  void calcTerminalCostDeriv(double t, const double * x, double * Vx, double * Vxx) const // :180-196
  {
    double r[3];
    refPos(t, r);
    for(int i = 0; i < 9; i++)
    {
      double d = (i < 3) ? x[i] - r[i] : x[i];
      Vx[i] = terminal_x[i] * d;
    }
    for(int e = 0; e < 81; e++)
    {
      Vxx[e] = 0;
    }
    for(int i = 0; i < 9; i++)
    {
      Vxx[i + i * 9] = terminal_x[i];
    }
  }
};
} // namespace oracle
