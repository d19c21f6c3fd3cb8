// 7-DoF joint-space manipulator problem (n = 14, m = 7) for the MI355X DDP solver — BASELINE.json config 5.
// The reference has no manipulator model; this synthetic chain is defined by this project
// so that Fx and Fu are dense and state dependent:
//   state x = [q(7), qd(7)], input u = joint torques
//   r_j   = u_j - damping qd_j - grav_j sin(q_0 + ... + q_j)          (torque left after damping and gravity)
//   qdd_i = sum_j W_ij(q) r_j,  W_ij = w_diag [i == j] + w_off cos(q_i - q_j)   (configuration-dependent coupling)
//   explicit Euler: q+ = q + dt qd, qd+ = qd + dt qdd
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
/** \tparam Real double ("manipulator": the reference's arithmetic) or float ("manipulator_f32": the fp64 tile kernel's float
    instantiation — the fp32 shape with seven inputs that ddp_kernels_tile32.hpp does not take). */
template<class Real>
class DDPProblemManipulatorT : public DDPProblemT<Real, 14, 7>
{
  using Base = DDPProblemT<Real, 14, 7>;

public:
  using typename Base::InputDimVector;
  using typename Base::InputInputDimMatrix;
  using typename Base::StateDimVector;
  using typename Base::StateInputDimMatrix;
  using typename Base::StateStateDimMatrix;
  using Base::dt_;
  static constexpr const char * kName = sizeof(Real) == 4 ? "manipulator_f32" : "manipulator";
  static constexpr int kJoints = 7;

  NMPC_HD explicit DDPProblemManipulatorT(Real dt = Real(0.01)) : Base(dt) {}

  NMPC_HD Real gravityGain(int j) const
  {
    return grav_scale_ * static_cast<Real>(kJoints - j) / kJoints;
  }
  NMPC_HD Real refAngle(int j) const
  {
    return (j % 2 == 1) ? -q_ref_scale_ : q_ref_scale_;
  }
  /** sin / cos of every joint-angle difference q_i - q_j and of the cumulative angles q_0 + ... + q_j, evaluated once per
      model call from the seven pairs (sin q_i, cos q_i) by the angle-sum identities: 7 sincosFast calls and ~110
      multiply-adds instead of 28 calls (a call is 33 instructions in double) — the Jacobian alone refers to these values
      ~400 times, and a rollout of the line search is little else.  The results differ from sin / cos of the rounded
      difference by a rounding error of the same size as that one's own (a few 1e-16 absolute). */
  struct Trig
  {
    Real cd[kJoints][kJoints], sd[kJoints][kJoints]; // cos / sin (q_i - q_j)
    Real sc[kJoints], cc[kJoints]; // sin / cos (q_0 + ... + q_j)
    NMPC_HD explicit Trig(const StateDimVector & x)
    {
      Real s[kJoints], c[kJoints];
      for(int i = 0; i < kJoints; i++)
      {
        sincosFast(x[i], s[i], c[i]);
      }
      for(int i = 0; i < kJoints; i++)
      {
        cd[i][i] = Real(1);
        sd[i][i] = Real(0);
        for(int j = i + 1; j < kJoints; j++)
        {
          const Real cij = c[i] * c[j] + s[i] * s[j];
          const Real sij = s[i] * c[j] - c[i] * s[j];
          cd[i][j] = cij;
          cd[j][i] = cij;
          sd[i][j] = sij;
          sd[j][i] = -sij;
        }
      }
      sc[0] = s[0];
      cc[0] = c[0];
      for(int j = 1; j < kJoints; j++)
      {
        sc[j] = sc[j - 1] * c[j] + cc[j - 1] * s[j];
        cc[j] = cc[j - 1] * c[j] - sc[j - 1] * s[j];
      }
    }
  };
  NMPC_HD Real coupling(const Trig & g, int i, int j) const
  {
    return (i == j ? w_diag_ : Real(0)) + w_off_ * g.cd[i][j];
  }

  /** Net joint torques r. */
  NMPC_HD void netTorque(const Trig & g, const StateDimVector & x, const InputDimVector & u, Real * r) const
  {
    for(int j = 0; j < kJoints; j++)
    {
      r[j] = (u[j] - damping_ * x[kJoints + j]) - gravityGain(j) * g.sc[j];
    }
  }

  NMPC_HD StateDimVector stateEq(Real, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u) const
  {
    const Trig g(x);
    Real r[kJoints];
    netTorque(g, x, u, r);
    StateDimVector x_next;
    for(int i = 0; i < kJoints; i++)
    {
      Real acc = 0;
      for(int j = 0; j < kJoints; j++)
      {
        acc += coupling(g, i, j) * r[j];
      }
      x_next[i] = x[i] + dt_ * x[kJoints + i];
      x_next[kJoints + i] = x[kJoints + i] + dt_ * acc;
    }
    return x_next;
  }

  NMPC_HD Real runningCost(Real, const StateDimVector & x, const InputDimVector & u) const
  {
    Real cost_x = 0;
    for(int j = 0; j < kJoints; j++)
    {
      const Real e = x[j] - refAngle(j);
      cost_x += wq_ * (e * e);
    }
    for(int j = 0; j < kJoints; j++)
    {
      cost_x += wv_ * (x[kJoints + j] * x[kJoints + j]);
    }
    return Real(0.5) * cost_x + Real(0.5) * wu_ * u.squaredNorm();
  }

  NMPC_HD Real terminalCost(Real, const StateDimVector & x) const
  {
    Real cost_x = 0;
    for(int j = 0; j < kJoints; j++)
    {
      const Real e = x[j] - refAngle(j);
      cost_x += (wt_scale_ * wq_) * (e * e);
    }
    for(int j = 0; j < kJoints; j++)
    {
      cost_x += (wt_scale_ * wv_) * (x[kJoints + j] * x[kJoints + j]);
    }
    return Real(0.5) * cost_x;
  }

  NMPC_HD void calcStateEqDeriv(Real, // t
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const Trig g(x);
    Real r[kJoints];
    netTorque(g, x, u, r);

    state_eq_deriv_x.setIdentity();
    state_eq_deriv_u.setZero();
    for(int i = 0; i < kJoints; i++)
    {
      state_eq_deriv_x(i, kJoints + i) = dt_;

      // sum_j sin(q_i - q_j) r_j : derivative of row i of W w.r.t. its own angle
      Real own = 0;
      for(int j = 0; j < kJoints; j++)
      {
        own += g.sd[i][j] * r[j];
      }
      for(int l = 0; l < kJoints; l++)
      {
        Real d_coupling = w_off_ * g.sd[i][l] * r[l];
        if(l == i)
        {
          d_coupling += -w_off_ * own;
        }
        // gravity torque of joint j depends on q_l for every l <= j
        Real d_gravity = 0;
        for(int j = l; j < kJoints; j++)
        {
          d_gravity += coupling(g, i, j) * (gravityGain(j) * g.cc[j]);
        }
        state_eq_deriv_x(kJoints + i, l) += dt_ * (d_coupling - d_gravity);
        state_eq_deriv_x(kJoints + i, kJoints + l) += dt_ * (-coupling(g, i, l) * damping_);
        state_eq_deriv_u(kJoints + i, l) = dt_ * coupling(g, i, l);
      }
    }
  }

  NMPC_HD void calcRunningCostDeriv(Real, // t
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & running_cost_deriv_x,
                                    InputDimVector & running_cost_deriv_u,
                                    StateStateDimMatrix & running_cost_deriv_xx,
                                    InputInputDimMatrix & running_cost_deriv_uu,
                                    StateInputDimMatrix & running_cost_deriv_xu) const
  {
    running_cost_deriv_xx.setZero();
    running_cost_deriv_uu.setZero();
    for(int j = 0; j < kJoints; j++)
    {
      running_cost_deriv_x[j] = wq_ * (x[j] - refAngle(j));
      running_cost_deriv_x[kJoints + j] = wv_ * x[kJoints + j];
      running_cost_deriv_xx(j, j) = wq_;
      running_cost_deriv_xx(kJoints + j, kJoints + j) = wv_;
      running_cost_deriv_u[j] = wu_ * u[j];
      running_cost_deriv_uu(j, j) = wu_;
    }
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(Real, // t
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_xx.setZero();
    for(int j = 0; j < kJoints; j++)
    {
      terminal_cost_deriv_x[j] = (wt_scale_ * wq_) * (x[j] - refAngle(j));
      terminal_cost_deriv_x[kJoints + j] = (wt_scale_ * wv_) * x[kJoints + j];
      terminal_cost_deriv_xx(j, j) = wt_scale_ * wq_;
      terminal_cost_deriv_xx(kJoints + j, kJoints + j) = wt_scale_ * wv_;
    }
  }

public:
  Real w_diag_ = Real(2.0);
  Real w_off_ = Real(0.15);
  Real damping_ = Real(0.5);
  Real grav_scale_ = Real(4.0);
  Real wq_ = Real(1.0), wv_ = Real(0.05), wu_ = Real(0.002);
  Real wt_scale_ = Real(20.0);
  Real q_ref_scale_ = Real(0.3);
};
using DDPProblemManipulator = DDPProblemManipulatorT<double>;
using DDPProblemManipulatorF32 = DDPProblemManipulatorT<float>;
} // namespace nmpc_amd
