// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/ddp_oracle.hpp header).
//
// The cart-pole restatement of oracle/models.hpp, written against `Real` like ddp_oracle.hpp and models_builder.hpp:
// Real = double in namespace `oracle` (what the reference computes in and what the pins of tests/test_oracle_pins.py hold);
// with ORACLE_F32 defined, Real = float in namespace `oracle_f32` — the float instantiation the fp32 problem type
// "cartpole_f32" (include/nmpc_amd/models/CartPole.hpp, fp32 tile kernel at n = 4, m = 1) is compared with.
#if defined(ORACLE_F32)
#  ifdef ORACLE_MODEL_CARTPOLE_F32_HPP
#    define ORACLE_MODEL_CARTPOLE_SKIP
#  else
#    define ORACLE_MODEL_CARTPOLE_F32_HPP
#    define ORACLE_NS oracle_f32
#    define ORACLE_REAL float
#  endif
#else
#  ifdef ORACLE_MODEL_CARTPOLE_F64_HPP
#    define ORACLE_MODEL_CARTPOLE_SKIP
#  else
#    define ORACLE_MODEL_CARTPOLE_F64_HPP
#    define ORACLE_NS oracle
#    define ORACLE_REAL double
#  endif
#endif
#ifndef ORACLE_MODEL_CARTPOLE_SKIP

#include <cmath>

namespace ORACLE_NS
{
using Real = ORACLE_REAL;
// ---------------------------------------------------------------------------------------------------
// Cart-pole: state [pos, theta, vel, omega], input [force].  nmpc_ddp/tests/src/TestDDPCartPole.cpp:28-234
// ---------------------------------------------------------------------------------------------------
struct CartPole
{
  using Real = ORACLE_NS::Real;
  static constexpr int N = 4;
  static constexpr int MMAX = 1;
  static constexpr int NPARAM = 14;

  Real dt = 0.01;
  Real cart_mass = 1.0; // :35
  Real pole_mass = 0.5; // :36
  Real pole_length = 2.0; // :37
  Real running_x[4] = {0.1, 1.0, 0.01, 0.1}; // :44
  Real running_u = 0.001; // :45  (launch file overrides to 0.01, tests/test/TestDDPCartPole.test:23)
  Real terminal_x[4] = {0.1, 1.0, 0.01, 0.1}; // :46
  Real ref_pos = 0.0; // getRefPos with target_pos_ = NaN  (:363-376)
  static constexpr Real g = Real(9.80665); // :230

  void setParams(const double * p)
  {
    dt = p[0];
    cart_mass = p[1];
    pole_mass = p[2];
    pole_length = p[3];
    for(int i = 0; i < 4; i++)
    {
      running_x[i] = p[4 + i];
    }
    running_u = p[8];
    for(int i = 0; i < 4; i++)
    {
      terminal_x[i] = p[9 + i];
    }
    ref_pos = p[13];
  }

  int inputDim(Real) const
  {
    return 1;
  }

  // :63-98 (explicit Euler)
  void stateEq(Real t, const Real * x, const Real * u, int, Real * xn) const
  {
    stateEqDt(t, x, u, dt, xn);
  }

  void stateEqDt(Real, const Real * x, const Real * u, Real step, Real * xn) const
  {
    const Real theta = x[1], vel = x[2], omega = x[3], f = u[0];
    const Real m1 = cart_mass, m2 = pole_mass, l = pole_length;
    const Real s = std::sin(theta), c = std::cos(theta);
    const Real omega2 = omega * omega;
    const Real denom = m1 + m2 * (s * s);
    Real xd[4];
    xd[0] = vel;
    xd[1] = omega;
    xd[2] = (f - m2 * l * omega2 * s + m2 * g * s * c) / denom;
    xd[3] = (f * c - m2 * l * omega2 * s * c + g * (m1 + m2) * s) / (l * denom);
    for(int i = 0; i < 4; i++)
    {
      xn[i] = x[i] + step * xd[i];
    }
  }

  // :100-105
  Real runningCost(Real, const Real * x, const Real * u, int) const
  {
    const Real ref[4] = {ref_pos, 0, 0, 0};
    Real sx = 0;
    for(int i = 0; i < 4; i++)
    {
      Real d = x[i] - ref[i];
      sx += running_x[i] * (d * d);
    }
    return Real(0.5) * sx + Real(0.5) * (running_u * (u[0] * u[0]));
  }

  // :107-112
  Real terminalCost(Real, const Real * x) const
  {
    const Real ref[4] = {ref_pos, 0, 0, 0};
    Real sx = 0;
    for(int i = 0; i < 4; i++)
    {
      Real d = x[i] - ref[i];
      sx += terminal_x[i] * (d * d);
    }
    return Real(0.5) * sx;
  }

  // :114-159
  void calcStateEqDeriv(Real, const Real * x, const Real * u, int, Real * Fx, Real * Fu) const
  {
    const Real theta = x[1], omega = x[3], f = u[0];
    const Real m1 = cart_mass, m2 = pole_mass, l = pole_length;
    const Real s = std::sin(theta), c = std::cos(theta);
    const Real omega2 = omega * omega;
    const Real s2 = s * s;
    const Real denom = m1 + m2 * s2;
    const Real denom2 = denom * denom;
    for(int e = 0; e < 16; e++)
    {
      Fx[e] = 0;
    }
    auto A = [&](int r, int col) -> Real & { return Fx[r + col * 4]; };
    A(0, 2) = 1;
    A(1, 3) = 1;
    A(2, 1) = ((-1 * m2 * l * omega2 * c + m2 * g * (1 - 2 * s2)) * denom
               + -1 * (f - m2 * l * omega2 * s + m2 * g * s * c) * (2 * m2 * s * c))
              / denom2;
    A(2, 3) = (-2 * m2 * l * omega * s) / denom;
    A(3, 1) = ((-1 * f * s + -1 * m2 * l * omega2 * (1 - 2 * s2) + g * (m1 + m2) * c) * denom
               + -1 * (f * c - m2 * l * omega2 * s * c + g * (m1 + m2) * s) * (2 * m2 * s * c))
              / (l * denom2);
    A(3, 3) = (-2 * m2 * l * omega * s * c) / (l * denom);
    for(int e = 0; e < 16; e++)
    {
      Fx[e] *= dt;
    }
    for(int i = 0; i < 4; i++)
    {
      Fx[i + i * 4] += Real(1);
    }
    Fu[0] = 0;
    Fu[1] = 0;
    Fu[2] = 1 / denom;
    Fu[3] = c / (l * denom);
    for(int i = 0; i < 4; i++)
    {
      Fu[i] *= dt;
    }
  }

  // :187-205
  void calcRunningCostDeriv(Real,
                            const Real * x,
                            const Real * u,
                            int,
                            Real * Lx,
                            Real * Lu,
                            Real * Lxx,
                            Real * Luu,
                            Real * Lxu) const
  {
    const Real ref[4] = {ref_pos, 0, 0, 0};
    for(int i = 0; i < 4; i++)
    {
      Lx[i] = running_x[i] * (x[i] - ref[i]);
    }
    Lu[0] = running_u * u[0];
    for(int e = 0; e < 16; e++)
    {
      Lxx[e] = 0;
    }
    for(int i = 0; i < 4; i++)
    {
      Lxx[i + i * 4] = running_x[i];
    }
    Luu[0] = running_u;
    for(int i = 0; i < 4; i++)
    {
      Lxu[i] = 0;
    }
  }

  // :217-227
  void calcTerminalCostDeriv(Real, const Real * x, Real * Vx, Real * Vxx) const
  {
    const Real ref[4] = {ref_pos, 0, 0, 0};
    for(int i = 0; i < 4; i++)
    {
      Vx[i] = terminal_x[i] * (x[i] - ref[i]);
    }
    for(int e = 0; e < 16; e++)
    {
      Vxx[e] = 0;
    }
    for(int i = 0; i < 4; i++)
    {
      Vxx[i + i * 4] = terminal_x[i];
    }
  }
};
} // namespace ORACLE_NS
#undef ORACLE_NS
#undef ORACLE_REAL
#endif
#undef ORACLE_MODEL_CARTPOLE_SKIP
