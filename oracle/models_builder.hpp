// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/ddp_oracle.hpp header).
//
// Builder-defined models for BASELINE.json configs 4 and 5.  The reference has NO quadrotor or manipulator
// model (SURVEY.md §8 d, C4/C5), so these are defined by this project (math stated in DESIGN.md §Models) and
// written twice, independently: here for the CPU oracle and in include/nmpc_amd/models/ for the HIP path.
// Their Jacobians are pinned by finite-difference checks in tests/ (same method as the reference's
// CheckDerivative tests, TestDDPCartPole.cpp:609-649).
// Precision: written against `Real` like ddp_oracle.hpp — Real = double in namespace `oracle`; with ORACLE_F32 defined
// Real = float in namespace `oracle_f32` (the fp32 instantiation BASELINE.json config 4 is compared with).
#if defined(ORACLE_F32)
#  ifdef ORACLE_MODELS_BUILDER_F32_HPP
#    error "models_builder.hpp (fp32) included twice"
#  endif
#  define ORACLE_MODELS_BUILDER_F32_HPP
#  define ORACLE_NS oracle_f32
#  define ORACLE_REAL float
#else
#  ifdef ORACLE_MODELS_BUILDER_F64_HPP
#    error "models_builder.hpp (fp64) included twice"
#  endif
#  define ORACLE_MODELS_BUILDER_F64_HPP
#  define ORACLE_NS oracle
#  define ORACLE_REAL double
#endif


#include <cmath>

namespace ORACLE_NS
{
using Real = ORACLE_REAL;
// ---------------------------------------------------------------------------------------------------
// Quadrotor: x = [p(3), rpy(3), v(3) world, w(3) body], u = 4 rotor thrusts, explicit Euler.
// ---------------------------------------------------------------------------------------------------
struct Quadrotor
{
  using Real = ORACLE_NS::Real;
  static constexpr int N = 12;
  static constexpr int MMAX = 4;
  static constexpr int NPARAM = 18;

  Real dt = 0.02;
  Real mass = 1.0;
  Real J[3] = {0.01, 0.01, 0.02};
  Real arm = 0.2;
  Real yaw_coef = 0.05;
  Real w_pos = 1.0, w_rpy = 0.5, w_vel = 0.1, w_omega = 0.05; // running state weights
  Real w_u = 0.01;
  Real wt_scale = 10.0; // terminal weights = wt_scale * running weights
  Real ref_pos[3] = {0, 0, 1.0};
  Real reserved[3] = {0, 0, 0};
  static constexpr Real g = 9.80665;

  void setParams(const double * p)
  {
    dt = p[0];
    mass = p[1];
    J[0] = p[2];
    J[1] = p[3];
    J[2] = p[4];
    arm = p[5];
    yaw_coef = p[6];
    w_pos = p[7];
    w_rpy = p[8];
    w_vel = p[9];
    w_omega = p[10];
    w_u = p[11];
    wt_scale = p[12];
    ref_pos[0] = p[13];
    ref_pos[1] = p[14];
    ref_pos[2] = p[15];
  }

  int inputDim(Real) const
  {
    return 4;
  }

  Real hoverThrust() const
  {
    return mass * g / 4;
  }

  void weights(Real * w) const
  {
    for(int i = 0; i < 3; i++)
    {
      w[i] = w_pos;
      w[3 + i] = w_rpy;
      w[6 + i] = w_vel;
      w[9 + i] = w_omega;
    }
  }

  void xdot(const Real * x, const Real * u, Real * xd) const
  {
    const Real sph = std::sin(x[3]), cph = std::cos(x[3]);
    const Real sth = std::sin(x[4]), cth = std::cos(x[4]);
    const Real sps = std::sin(x[5]), cps = std::cos(x[5]);
    const Real tth = sth / cth;
    const Real p = x[9], q = x[10], r = x[11];
    const Real F = ((u[0] + u[1]) + u[2]) + u[3];
    xd[0] = x[6];
    xd[1] = x[7];
    xd[2] = x[8];
    xd[3] = p + sph * tth * q + cph * tth * r;
    xd[4] = cph * q - sph * r;
    xd[5] = (sph * q + cph * r) / cth;
    const Real b0 = cph * sth * cps + sph * sps;
    const Real b1 = cph * sth * sps - sph * cps;
    const Real b2 = cph * cth;
    xd[6] = F / mass * b0;
    xd[7] = F / mass * b1;
    xd[8] = F / mass * b2 - g;
    const Real tx = arm * (u[1] - u[3]);
    const Real ty = arm * (u[2] - u[0]);
    const Real tz = yaw_coef * (((u[0] - u[1]) + u[2]) - u[3]);
    xd[9] = (tx - (J[2] - J[1]) * q * r) / J[0];
    xd[10] = (ty - (J[0] - J[2]) * p * r) / J[1];
    xd[11] = (tz - (J[1] - J[0]) * p * q) / J[2];
  }

  void stateEq(Real, const Real * x, const Real * u, int, Real * xn) const
  {
    Real xd[12];
    xdot(x, u, xd);
    for(int i = 0; i < 12; i++)
    {
      xn[i] = x[i] + dt * xd[i];
    }
  }

  Real runningCost(Real, const Real * x, const Real * u, int) const
  {
    Real w[12];
    weights(w);
    Real s = 0;
    for(int i = 0; i < 12; i++)
    {
      Real d = (i < 3) ? x[i] - ref_pos[i] : x[i];
      s += w[i] * (d * d);
    }
    Real su = 0;
    const Real fh = hoverThrust();
    for(int a = 0; a < 4; a++)
    {
      Real d = u[a] - fh;
      su += d * d;
    }
    return 0.5 * s + 0.5 * w_u * su;
  }

  Real terminalCost(Real, const Real * x) const
  {
    Real w[12];
    weights(w);
    Real s = 0;
    for(int i = 0; i < 12; i++)
    {
      Real d = (i < 3) ? x[i] - ref_pos[i] : x[i];
      s += (wt_scale * w[i]) * (d * d);
    }
    return 0.5 * s;
  }

  void calcStateEqDeriv(Real, const Real * x, const Real * u, int, Real * Fx, Real * Fu) const
  {
    const Real sph = std::sin(x[3]), cph = std::cos(x[3]);
    const Real sth = std::sin(x[4]), cth = std::cos(x[4]);
    const Real sps = std::sin(x[5]), cps = std::cos(x[5]);
    const Real tth = sth / cth;
    const Real p = x[9], q = x[10], r = x[11];
    const Real F = ((u[0] + u[1]) + u[2]) + u[3];
    Real A[144];
    for(int e = 0; e < 144; e++)
    {
      A[e] = 0;
    }
    auto a = [&](int row, int col) -> Real & { return A[row + col * 12]; };
    a(0, 6) = 1;
    a(1, 7) = 1;
    a(2, 8) = 1;
    // rpy kinematics
    a(3, 3) = cph * tth * q - sph * tth * r;
    a(3, 4) = (sph * q + cph * r) / (cth * cth);
    a(3, 9) = 1;
    a(3, 10) = sph * tth;
    a(3, 11) = cph * tth;
    a(4, 3) = -sph * q - cph * r;
    a(4, 10) = cph;
    a(4, 11) = -sph;
    a(5, 3) = (cph * q - sph * r) / cth;
    a(5, 4) = (sph * q + cph * r) * sth / (cth * cth);
    a(5, 10) = sph / cth;
    a(5, 11) = cph / cth;
    // thrust direction
    const Real fm = F / mass;
    a(6, 3) = fm * (-sph * sth * cps + cph * sps);
    a(7, 3) = fm * (-sph * sth * sps - cph * cps);
    a(8, 3) = fm * (-sph * cth);
    a(6, 4) = fm * (cph * cth * cps);
    a(7, 4) = fm * (cph * cth * sps);
    a(8, 4) = fm * (-cph * sth);
    a(6, 5) = fm * (-cph * sth * sps + sph * cps);
    a(7, 5) = fm * (cph * sth * cps + sph * sps);
    // body rates
    a(9, 10) = -(J[2] - J[1]) * r / J[0];
    a(9, 11) = -(J[2] - J[1]) * q / J[0];
    a(10, 9) = -(J[0] - J[2]) * r / J[1];
    a(10, 11) = -(J[0] - J[2]) * p / J[1];
    a(11, 9) = -(J[1] - J[0]) * q / J[2];
    a(11, 10) = -(J[1] - J[0]) * p / J[2];
    for(int e = 0; e < 144; e++)
    {
      Fx[e] = dt * A[e];
    }
    for(int i = 0; i < 12; i++)
    {
      Fx[i + i * 12] += 1.0;
    }
    const Real b0 = cph * sth * cps + sph * sps;
    const Real b1 = cph * sth * sps - sph * cps;
    const Real b2 = cph * cth;
    for(int e = 0; e < 48; e++)
    {
      Fu[e] = 0;
    }
    for(int c = 0; c < 4; c++)
    {
      Fu[6 + c * 12] = dt * (b0 / mass);
      Fu[7 + c * 12] = dt * (b1 / mass);
      Fu[8 + c * 12] = dt * (b2 / mass);
    }
    Fu[9 + 1 * 12] = dt * (arm / J[0]);
    Fu[9 + 3 * 12] = dt * (-arm / J[0]);
    Fu[10 + 2 * 12] = dt * (arm / J[1]);
    Fu[10 + 0 * 12] = dt * (-arm / J[1]);
    Fu[11 + 0 * 12] = dt * (yaw_coef / J[2]);
    Fu[11 + 1 * 12] = dt * (-yaw_coef / J[2]);
    Fu[11 + 2 * 12] = dt * (yaw_coef / J[2]);
    Fu[11 + 3 * 12] = dt * (-yaw_coef / J[2]);
  }

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
    Real w[12];
    weights(w);
    for(int i = 0; i < 12; i++)
    {
      Real d = (i < 3) ? x[i] - ref_pos[i] : x[i];
      Lx[i] = w[i] * d;
    }
    const Real fh = hoverThrust();
    for(int a = 0; a < 4; a++)
    {
      Lu[a] = w_u * (u[a] - fh);
    }
    for(int e = 0; e < 144; e++)
    {
      Lxx[e] = 0;
    }
    for(int i = 0; i < 12; i++)
    {
      Lxx[i + i * 12] = w[i];
    }
    for(int e = 0; e < 16; e++)
    {
      Luu[e] = 0;
    }
    for(int a = 0; a < 4; a++)
    {
      Luu[a + a * 4] = w_u;
    }
    for(int e = 0; e < 48; e++)
    {
      Lxu[e] = 0;
    }
  }

  void calcTerminalCostDeriv(Real, const Real * x, Real * Vx, Real * Vxx) const
  {
    Real w[12];
    weights(w);
    for(int i = 0; i < 12; i++)
    {
      Real d = (i < 3) ? x[i] - ref_pos[i] : x[i];
      Vx[i] = (wt_scale * w[i]) * d;
    }
    for(int e = 0; e < 144; e++)
    {
      Vxx[e] = 0;
    }
    for(int i = 0; i < 12; i++)
    {
      Vxx[i + i * 12] = wt_scale * w[i];
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// Planar VTOL (bicopter), n = 6, m = 2: x = [px, pz, th, vx, vz, om], u = left / right rotor thrust, explicit Euler.
//   vx' = -(u0 + u1) / mass * sin(th),  vz' = (u0 + u1) / mass * cos(th) - g,  om' = arm * (u1 - u0) / inertia
// A builder-defined shape with 5 <= n <= 8 (the reference's template takes any StateDim / InputDim, DDPSolver.h:23-25).
// ---------------------------------------------------------------------------------------------------
struct PlanarVtol
{
  using Real = ORACLE_NS::Real;
  static constexpr int N = 6;
  static constexpr int MMAX = 2;
  static constexpr int NPARAM = 12;

  Real dt = 0.02;
  Real mass = 1.0;
  Real inertia = 0.02;
  Real arm = 0.25;
  Real w_pos = 1.0, w_ang = 0.5, w_vel = 0.1, w_omega = 0.05;
  Real w_u = 0.01;
  Real wt_scale = 10.0;
  Real ref_pos[2] = {0, 1.0};
  static constexpr Real g = 9.80665;

  void setParams(const double * p)
  {
    dt = p[0];
    mass = p[1];
    inertia = p[2];
    arm = p[3];
    w_pos = p[4];
    w_ang = p[5];
    w_vel = p[6];
    w_omega = p[7];
    w_u = p[8];
    wt_scale = p[9];
    ref_pos[0] = p[10];
    ref_pos[1] = p[11];
  }
  int inputDim(Real) const
  {
    return 2;
  }
  Real hover() const
  {
    return mass * g / 2;
  }
  void weights(Real * w) const
  {
    w[0] = w_pos;
    w[1] = w_pos;
    w[2] = w_ang;
    w[3] = w_vel;
    w[4] = w_vel;
    w[5] = w_omega;
  }
  Real err(const Real * x, int i) const
  {
    return (i < 2) ? x[i] - ref_pos[i] : x[i];
  }

  void stateEq(Real, const Real * x, const Real * u, int, Real * xn) const
  {
    const Real sth = std::sin(x[2]), cth = std::cos(x[2]);
    const Real a = (u[0] + u[1]) / mass;
    const Real xd[6] = {x[3], x[4], x[5], -a * sth, a * cth - g, arm * (u[1] - u[0]) / inertia};
    for(int i = 0; i < 6; i++)
    {
      xn[i] = x[i] + dt * xd[i];
    }
  }
  Real runningCost(Real, const Real * x, const Real * u, int) const
  {
    Real w[6];
    weights(w);
    Real s = 0;
    for(int i = 0; i < 6; i++)
    {
      const Real d = err(x, i);
      s += w[i] * (d * d);
    }
    Real su = 0;
    for(int a = 0; a < 2; a++)
    {
      const Real d = u[a] - hover();
      su += d * d;
    }
    return 0.5 * s + 0.5 * w_u * su;
  }
  Real terminalCost(Real, const Real * x) const
  {
    Real w[6];
    weights(w);
    Real s = 0;
    for(int i = 0; i < 6; i++)
    {
      const Real d = err(x, i);
      s += (wt_scale * w[i]) * (d * d);
    }
    return 0.5 * s;
  }
  void calcStateEqDeriv(Real, const Real * x, const Real * u, int, Real * Fx, Real * Fu) const
  {
    const Real sth = std::sin(x[2]), cth = std::cos(x[2]);
    const Real a = (u[0] + u[1]) / mass;
    for(int e = 0; e < 36; e++)
    {
      Fx[e] = 0;
    }
    for(int i = 0; i < 6; i++)
    {
      Fx[i + i * 6] = 1.0;
    }
    Fx[0 + 3 * 6] = dt;
    Fx[1 + 4 * 6] = dt;
    Fx[2 + 5 * 6] = dt;
    Fx[3 + 2 * 6] = dt * (-a * cth);
    Fx[4 + 2 * 6] = dt * (-a * sth);
    for(int e = 0; e < 12; e++)
    {
      Fu[e] = 0;
    }
    for(int c = 0; c < 2; c++)
    {
      Fu[3 + c * 6] = dt * (-sth / mass);
      Fu[4 + c * 6] = dt * (cth / mass);
    }
    Fu[5 + 0 * 6] = dt * (-arm / inertia);
    Fu[5 + 1 * 6] = dt * (arm / inertia);
  }
  void calcRunningCostDeriv(Real, const Real * x, const Real * u, int, Real * Lx, Real * Lu, Real * Lxx, Real * Luu, Real * Lxu) const
  {
    Real w[6];
    weights(w);
    for(int e = 0; e < 36; e++)
    {
      Lxx[e] = 0;
    }
    for(int i = 0; i < 6; i++)
    {
      Lx[i] = w[i] * err(x, i);
      Lxx[i + i * 6] = w[i];
    }
    for(int e = 0; e < 4; e++)
    {
      Luu[e] = 0;
    }
    for(int a = 0; a < 2; a++)
    {
      Lu[a] = w_u * (u[a] - hover());
      Luu[a + a * 2] = w_u;
    }
    for(int e = 0; e < 12; e++)
    {
      Lxu[e] = 0;
    }
  }
  void calcTerminalCostDeriv(Real, const Real * x, Real * Vx, Real * Vxx) const
  {
    Real w[6];
    weights(w);
    for(int e = 0; e < 36; e++)
    {
      Vxx[e] = 0;
    }
    for(int i = 0; i < 6; i++)
    {
      Vx[i] = (wt_scale * w[i]) * err(x, i);
      Vxx[i + i * 6] = wt_scale * w[i];
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// Manipulator (synthetic 7-DoF joint-space chain): x = [q(7), qd(7)], u = joint torques.
//   r_j   = u_j - damping * qd_j - grav_j * sin(S_j),   S_j = q_0 + ... + q_j
//   qdd_i = sum_j W_ij(q) r_j,   W_ij = w_diag * [i == j] + w_off * cos(q_i - q_j)   (dense, state dependent)
//   explicit Euler: q+ = q + dt qd, qd+ = qd + dt qdd
// ---------------------------------------------------------------------------------------------------
struct Manipulator
{
  using Real = ORACLE_NS::Real;
  static constexpr int N = 14;
  static constexpr int MMAX = 7;
  static constexpr int NPARAM = 12;
  static constexpr int NJ = 7;

  Real dt = 0.01;
  Real w_diag = 2.0;
  Real w_off = 0.15;
  Real damping = 0.5;
  Real grav_scale = 4.0; // grav_j = grav_scale * (NJ - j) / NJ
  Real wq = 1.0, wv = 0.05, wu = 0.002;
  Real wt_scale = 20.0;
  Real q_ref_scale = 0.3; // q_ref_j = q_ref_scale * (j odd ? -1 : 1)
  Real reserved[2] = {0, 0};

  void setParams(const double * p)
  {
    dt = p[0];
    w_diag = p[1];
    w_off = p[2];
    damping = p[3];
    grav_scale = p[4];
    wq = p[5];
    wv = p[6];
    wu = p[7];
    wt_scale = p[8];
    q_ref_scale = p[9];
  }

  int inputDim(Real) const
  {
    return 7;
  }

  Real grav(int j) const
  {
    return grav_scale * static_cast<Real>(NJ - j) / NJ;
  }
  Real qRef(int j) const
  {
    return (j % 2 == 1) ? -q_ref_scale : q_ref_scale;
  }

  void residual(const Real * x, const Real * u, Real * r, Real * S) const
  {
    Real acc = 0;
    for(int j = 0; j < NJ; j++)
    {
      acc += x[j];
      S[j] = acc;
      r[j] = (u[j] - damping * x[NJ + j]) - grav(j) * std::sin(acc);
    }
  }

  void stateEq(Real, const Real * x, const Real * u, int, Real * xn) const
  {
    Real r[NJ], S[NJ];
    residual(x, u, r, S);
    for(int i = 0; i < NJ; i++)
    {
      Real s = 0;
      for(int j = 0; j < NJ; j++)
      {
        Real W = (i == j ? w_diag : 0.0) + w_off * std::cos(x[i] - x[j]);
        s += W * r[j];
      }
      xn[i] = x[i] + dt * x[NJ + i];
      xn[NJ + i] = x[NJ + i] + dt * s;
    }
  }

  Real runningCost(Real, const Real * x, const Real * u, int) const
  {
    Real s = 0;
    for(int j = 0; j < NJ; j++)
    {
      Real d = x[j] - qRef(j);
      s += wq * (d * d);
    }
    for(int j = 0; j < NJ; j++)
    {
      s += wv * (x[NJ + j] * x[NJ + j]);
    }
    Real su = 0;
    for(int j = 0; j < NJ; j++)
    {
      su += u[j] * u[j];
    }
    return 0.5 * s + 0.5 * wu * su;
  }

  Real terminalCost(Real, const Real * x) const
  {
    Real s = 0;
    for(int j = 0; j < NJ; j++)
    {
      Real d = x[j] - qRef(j);
      s += (wt_scale * wq) * (d * d);
    }
    for(int j = 0; j < NJ; j++)
    {
      s += (wt_scale * wv) * (x[NJ + j] * x[NJ + j]);
    }
    return 0.5 * s;
  }

  void calcStateEqDeriv(Real, const Real * x, const Real * u, int, Real * Fx, Real * Fu) const
  {
    Real r[NJ], S[NJ];
    residual(x, u, r, S);
    Real W[NJ * NJ]; // W(i,j) at i + j*NJ
    for(int i = 0; i < NJ; i++)
    {
      for(int j = 0; j < NJ; j++)
      {
        W[i + j * NJ] = (i == j ? w_diag : 0.0) + w_off * std::cos(x[i] - x[j]);
      }
    }
    for(int e = 0; e < N * N; e++)
    {
      Fx[e] = 0;
    }
    for(int i = 0; i < N; i++)
    {
      Fx[i + i * N] = 1.0;
    }
    for(int i = 0; i < NJ; i++)
    {
      Fx[i + (NJ + i) * N] = dt; // dq+/dqd
    }
    for(int i = 0; i < NJ; i++)
    {
      // d qdd_i / d q_l
      Real self = 0;
      for(int j = 0; j < NJ; j++)
      {
        self += std::sin(x[i] - x[j]) * r[j];
      }
      for(int l = 0; l < NJ; l++)
      {
        Real dW = w_off * std::sin(x[i] - x[l]) * r[l];
        if(l == i)
        {
          dW += -w_off * self;
        }
        Real dG = 0;
        for(int j = l; j < NJ; j++)
        {
          dG += W[i + j * NJ] * (grav(j) * std::cos(S[j]));
        }
        Fx[(NJ + i) + l * N] += dt * (dW - dG);
      }
      // d qdd_i / d qd_l
      for(int l = 0; l < NJ; l++)
      {
        Fx[(NJ + i) + (NJ + l) * N] += dt * (-W[i + l * NJ] * damping);
      }
    }
    for(int e = 0; e < N * NJ; e++)
    {
      Fu[e] = 0;
    }
    for(int i = 0; i < NJ; i++)
    {
      for(int l = 0; l < NJ; l++)
      {
        Fu[(NJ + i) + l * N] = dt * W[i + l * NJ];
      }
    }
  }

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
    for(int j = 0; j < NJ; j++)
    {
      Lx[j] = wq * (x[j] - qRef(j));
      Lx[NJ + j] = wv * x[NJ + j];
      Lu[j] = wu * u[j];
    }
    for(int e = 0; e < N * N; e++)
    {
      Lxx[e] = 0;
    }
    for(int j = 0; j < NJ; j++)
    {
      Lxx[j + j * N] = wq;
      Lxx[(NJ + j) + (NJ + j) * N] = wv;
    }
    for(int e = 0; e < NJ * NJ; e++)
    {
      Luu[e] = 0;
    }
    for(int j = 0; j < NJ; j++)
    {
      Luu[j + j * NJ] = wu;
    }
    for(int e = 0; e < N * NJ; e++)
    {
      Lxu[e] = 0;
    }
  }

  void calcTerminalCostDeriv(Real, const Real * x, Real * Vx, Real * Vxx) const
  {
    for(int j = 0; j < NJ; j++)
    {
      Vx[j] = (wt_scale * wq) * (x[j] - qRef(j));
      Vx[NJ + j] = (wt_scale * wv) * x[NJ + j];
    }
    for(int e = 0; e < N * N; e++)
    {
      Vxx[e] = 0;
    }
    for(int j = 0; j < NJ; j++)
    {
      Vxx[j + j * N] = wt_scale * wq;
      Vxx[(NJ + j) + (NJ + j) * N] = wt_scale * wv;
    }
  }
};
} // namespace ORACLE_NS
#undef ORACLE_NS
#undef ORACLE_REAL
