// The jet gyrostat of JetGyrostatEigenStyle.hpp once more, entry by entry on scalars (element access only: no views, no
// initialisers, no products).  It is the yardstick of tests/cpp/test_eigen_style_port.cpp: what linalg.hpp's Eigen-subset syntax
// evaluates to has to be these operations in this order, bit for bit, on the host and on gfx950.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace conformance
{
class JetGyrostatPlain : public nmpc_amd::DDPProblem<9, nmpc_amd::Dynamic, 8>
{
public:
  NMPC_HD explicit JetGyrostatPlain(double dt = 0.05) : DDPProblem(dt) {}

  using DDPProblem::inputDim;

  NMPC_HD static int jetCount(double t)
  {
    t += 1e-6;
    return t < 1.0 ? 8 : (t < 1.5 ? 0 : 4);
  }
  NMPC_HD int inputDim(double t) const
  {
    return jetCount(t);
  }

  //! torque arm of jet k of a set of `count`: mount x axis
  NMPC_HD static void torqueArm(int k, int count, double (&arm)[3])
  {
    const double phi = 0.25 * M_PI * k + 0.1;
    const double mount[3] = {0.6 * cos(phi), 0.6 * sin(phi), (k % 2 == 0 ? 0.3 : -0.3)};
    double axis[3] = {-sin(phi), cos(phi), 0.4 * (count == 4 ? 1.0 : -1.0)};
    double len2 = 0;
    for(int a = 0; a < 3; a++)
    {
      len2 += axis[a] * axis[a];
    }
    const double len = sqrt(len2);
    for(int a = 0; a < 3; a++)
    {
      axis[a] = axis[a] / len;
    }
    arm[0] = mount[1] * axis[2] - mount[2] * axis[1];
    arm[1] = mount[2] * axis[0] - mount[0] * axis[2];
    arm[2] = mount[0] * axis[1] - mount[1] * axis[0];
  }

  NMPC_HD static void referenceAt(double t, double (&ref)[9])
  {
    for(int j = 0; j < 9; j++)
    {
      ref[j] = 0.0;
    }
    ref[1] = 0.2 * sin(0.5 * t);
    ref[4] = 0.1 * cos(0.5 * t);
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const int count = jetCount(t);
    const double r[3] = {x[0], x[1], x[2]}, w[3] = {x[3], x[4], x[5]}, h[3] = {x[6], x[7], x[8]};
    double rate[9];
    const double rxw[3] = {r[1] * w[2] - r[2] * w[1], r[2] * w[0] - r[0] * w[2], r[0] * w[1] - r[1] * w[0]};
    double stored[3], jets[3] = {0.0, 0.0, 0.0};
    for(int a = 0; a < 3; a++)
    {
      rate[a] = w[a] + rxw[a] * 0.5;
      stored[a] = inertia(a) * w[a] + h[a];
    }
    for(int k = 0; k < count; k++)
    {
      double arm[3];
      torqueArm(k, count, arm);
      for(int a = 0; a < 3; a++)
      {
        jets[a] += arm[a] * u[k];
      }
    }
    const double gyro[3] = {w[1] * stored[2] - w[2] * stored[1], w[2] * stored[0] - w[0] * stored[2],
                            w[0] * stored[1] - w[1] * stored[0]};
    for(int a = 0; a < 3; a++)
    {
      rate[3 + a] = (jets[a] - gyro[a]) / inertia(a);
      rate[6 + a] = kWheelGain * w[a] - kWheelLeak * h[a];
    }
    StateDimVector next;
    for(int j = 0; j < 9; j++)
    {
      next[j] = x[j] + dt_ * rate[j];
    }
    return next;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    double ref[9], state_term = 0, thrust_term = 0;
    referenceAt(t, ref);
    for(int j = 0; j < 9; j++)
    {
      const double miss = x[j] - ref[j];
      state_term += runningWeight(j) * (miss * miss);
    }
    for(int k = 0; k < u.size(); k++)
    {
      thrust_term += u[k] * u[k];
    }
    return 0.5 * state_term + 0.5 * kThrustWeight * thrust_term;
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    double ref[9], state_term = 0;
    referenceAt(t, ref);
    for(int j = 0; j < 9; j++)
    {
      const double miss = x[j] - ref[j];
      state_term += terminalWeight(j) * (miss * miss);
    }
    return 0.5 * state_term;
  }

  NMPC_HD void calcStateEqDeriv(double t,
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & fx,
                                StateInputDimMatrix & fu) const
  {
    const int count = jetCount(t);
    const double r[3] = {x[0], x[1], x[2]}, w[3] = {x[3], x[4], x[5]}, h[3] = {x[6], x[7], x[8]};
    double stored[3], inv_inertia[3];
    for(int a = 0; a < 3; a++)
    {
      stored[a] = inertia(a) * w[a] + h[a];
      inv_inertia[a] = 1.0 / inertia(a);
    }
    // [v]x, row by row
    const double Sr[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
    const double Sw[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    const double Ss[3][3] = {{0, -stored[2], stored[1]}, {stored[2], 0, -stored[0]}, {-stored[1], stored[0], 0}};
    for(int c = 0; c < 9; c++)
    {
      for(int i = 0; i < 9; i++)
      {
        fx(i, c) = 0.0;
      }
    }
    for(int i = 0; i < 3; i++)
    {
      for(int c = 0; c < 3; c++)
      {
        fx(i, c) = Sw[i][c] * -0.5;
        fx(i, 3 + c) = (i == c ? 1.0 : 0.0) + Sr[i][c] * 0.5;
        fx(3 + i, 3 + c) = inv_inertia[i] * (Ss[i][c] - Sw[i][c] * inertia(c));
        fx(3 + i, 6 + c) = inv_inertia[i] * -Sw[i][c];
      }
      fx(6 + i, 3 + i) = kWheelGain;
      fx(6 + i, 6 + i) = -kWheelLeak;
    }
    for(int c = 0; c < 9; c++)
    {
      for(int i = 0; i < 9; i++)
      {
        fx(i, c) *= dt_;
      }
    }
    for(int j = 0; j < 9; j++)
    {
      fx(j, j) += 1.0;
    }
    fu.resize(9, u.size());
    for(int k = 0; k < count; k++)
    {
      double arm[3];
      torqueArm(k, count, arm);
      for(int i = 0; i < 9; i++)
      {
        fu(i, k) = 0.0;
      }
      for(int a = 0; a < 3; a++)
      {
        fu(3 + a, k) = (inv_inertia[a] * arm[a]) * dt_;
      }
    }
  }

  NMPC_HD void calcRunningCostDeriv(double t,
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & lx,
                                    InputDimVector & lu,
                                    StateStateDimMatrix & lxx,
                                    InputInputDimMatrix & luu,
                                    StateInputDimMatrix & lxu) const
  {
    const int m = u.size();
    double ref[9];
    referenceAt(t, ref);
    lu.resize(m);
    luu.resize(m, m);
    lxu.resize(9, m);
    for(int j = 0; j < 9; j++)
    {
      lx[j] = runningWeight(j) * (x[j] - ref[j]);
      for(int i = 0; i < 9; i++)
      {
        lxx(i, j) = (i == j) ? runningWeight(j) : 0.0;
      }
      for(int k = 0; k < m; k++)
      {
        lxu(j, k) = 0.0;
      }
    }
    for(int k = 0; k < m; k++)
    {
      lu[k] = kThrustWeight * u[k];
      for(int l = 0; l < m; l++)
      {
        luu(l, k) = (l == k) ? kThrustWeight : 0.0;
      }
    }
  }

  NMPC_HD void calcTerminalCostDeriv(double t, const StateDimVector & x, StateDimVector & vx, StateStateDimMatrix & vxx) const
  {
    double ref[9];
    referenceAt(t, ref);
    for(int j = 0; j < 9; j++)
    {
      vx[j] = terminalWeight(j) * (x[j] - ref[j]);
      for(int i = 0; i < 9; i++)
      {
        vxx(i, j) = (i == j) ? terminalWeight(j) : 0.0;
      }
    }
  }

  static constexpr double kWheelGain = 0.3, kWheelLeak = 0.8, kThrustWeight = 1e-3;
  // (functions rather than static arrays: device code may not index a host-side constant array at run time)
  NMPC_HD static double inertia(int a)
  {
    return a == 0 ? 2.4 : (a == 1 ? 3.1 : 1.7);
  }
  NMPC_HD static double runningWeight(int j)
  {
    return j < 3 ? 4.0 : (j < 6 ? (j == 4 ? 0.25 : 0.5) : 0.01);
  }
  NMPC_HD static double terminalWeight(int j)
  {
    return j < 3 ? 40.0 : (j < 6 ? 2.0 : 0.1);
  }
};
} // namespace conformance
