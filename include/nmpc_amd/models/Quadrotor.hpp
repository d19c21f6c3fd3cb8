// Quadrotor problem (n = 12, m = 4) for the MI355X DDP solver — BASELINE.json config 4.
// The reference has no quadrotor model; this one is defined by this project:
//   state  x = [position(3), roll-pitch-yaw(3), world velocity(3), body rates(3)]
//   input  u = thrust of the four rotors (plus configuration, rotor i on the +x, +y, -x, -y arm)
//   rigid-body dynamics with ZYX Euler kinematics, explicit Euler step; quadratic costs around hover.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
/** \tparam Real Real: the reference's arithmetic, registered as "quadrotor"; float: BASELINE.json config 4 ("fp32"),
    registered as "quadrotor_f32".  The statements are the same; every constant is converted to Real once. */
template<class Real>
class DDPProblemQuadrotorT : public DDPProblemT<Real, 12, 4>
{
  using Base = DDPProblemT<Real, 12, 4>;
  using Base::dt_;

public:
  using typename Base::InputDimVector;
  using typename Base::InputInputDimMatrix;
  using typename Base::StateDimVector;
  using typename Base::StateInputDimMatrix;
  using typename Base::StateStateDimMatrix;
  static constexpr const char * kName = sizeof(Real) == 8 ? "quadrotor" : "quadrotor_f32";
  static constexpr Real g_ = Real(9.80665); // [m/s^2]

  NMPC_HD explicit DDPProblemQuadrotorT(Real dt = Real(0.02)) : Base(dt) {}

  NMPC_HD Real hoverThrust() const
  {
    return mass_ * g_ / 4;
  }

  NMPC_HD Real stateWeight(int i) const
  {
    return i < 3 ? w_pos_ : (i < 6 ? w_rpy_ : (i < 9 ? w_vel_ : w_omega_));
  }

  NMPC_HD Real stateError(const StateDimVector & x, int i) const
  {
    return i < 3 ? x[i] - ref_pos_[i] : x[i];
  }

  /** Trigonometry of the attitude shared by the dynamics and its Jacobian. */
  struct Trig
  {
    Real sr, cr, sp, cp, sy, cy, tp;
    NMPC_HD explicit Trig(const StateDimVector & x)
    {
      // attitude angles are physical (|angle| << 2^27 rad): sincosFast, ~130 cycles per pair on gfx950 instead of
      // ~640 for the math library's sin + cos (linalg.hpp)
      sincosFast(x[3], sr, cr);
      sincosFast(x[4], sp, cp);
      sincosFast(x[5], sy, cy);
      tp = sp / cp;
    }
  };

  NMPC_HD StateDimVector stateEq(Real, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u) const
  {
    const Trig g(x);
    const Real p = x[9], q = x[10], r = x[11];
    const Real thrust = ((u[0] + u[1]) + u[2]) + u[3];
    // body z axis expressed in the world frame
    const Real bz[3] = {g.cr * g.sp * g.cy + g.sr * g.sy, g.cr * g.sp * g.sy - g.sr * g.cy, g.cr * g.cp};
    const Real torque[3] = {arm_ * (u[1] - u[3]), arm_ * (u[2] - u[0]), yaw_coef_ * (((u[0] - u[1]) + u[2]) - u[3])};

    Real x_dot[12];
    x_dot[0] = x[6];
    x_dot[1] = x[7];
    x_dot[2] = x[8];
    x_dot[3] = p + g.sr * g.tp * q + g.cr * g.tp * r;
    x_dot[4] = g.cr * q - g.sr * r;
    x_dot[5] = (g.sr * q + g.cr * r) / g.cp;
    x_dot[6] = thrust / mass_ * bz[0];
    x_dot[7] = thrust / mass_ * bz[1];
    x_dot[8] = thrust / mass_ * bz[2] - g_;
    x_dot[9] = (torque[0] - (inertia_[2] - inertia_[1]) * q * r) / inertia_[0];
    x_dot[10] = (torque[1] - (inertia_[0] - inertia_[2]) * p * r) / inertia_[1];
    x_dot[11] = (torque[2] - (inertia_[1] - inertia_[0]) * p * q) / inertia_[2];

    StateDimVector x_next;
    for(int i = 0; i < 12; i++)
    {
      x_next[i] = x[i] + dt_ * x_dot[i];
    }
    return x_next;
  }

  NMPC_HD Real runningCost(Real, const StateDimVector & x, const InputDimVector & u) const
  {
    Real cost_x = 0;
    for(int i = 0; i < 12; i++)
    {
      const Real e = stateError(x, i);
      cost_x += stateWeight(i) * (e * e);
    }
    Real cost_u = 0;
    for(int i = 0; i < 4; i++)
    {
      const Real e = u[i] - hoverThrust();
      cost_u += e * e;
    }
    return Real(0.5) * cost_x + Real(0.5) * w_u_ * cost_u;
  }

  NMPC_HD Real terminalCost(Real, const StateDimVector & x) const
  {
    Real cost_x = 0;
    for(int i = 0; i < 12; i++)
    {
      const Real e = stateError(x, i);
      cost_x += (wt_scale_ * stateWeight(i)) * (e * e);
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
    const Real p = x[9], q = x[10], r = x[11];
    const Real thrust = ((u[0] + u[1]) + u[2]) + u[3];
    const Real acc = thrust / mass_;
    const Real cp2 = g.cp * g.cp;

    StateStateDimMatrix & A = state_eq_deriv_x; // filled with d(x_dot)/dx, then scaled
    A.setZero();
    A(0, 6) = 1;
    A(1, 7) = 1;
    A(2, 8) = 1;
    // Euler-angle kinematics
    A(3, 3) = g.cr * g.tp * q - g.sr * g.tp * r;
    A(3, 4) = (g.sr * q + g.cr * r) / cp2;
    A(3, 9) = 1;
    A(3, 10) = g.sr * g.tp;
    A(3, 11) = g.cr * g.tp;
    A(4, 3) = -g.sr * q - g.cr * r;
    A(4, 10) = g.cr;
    A(4, 11) = -g.sr;
    A(5, 3) = (g.cr * q - g.sr * r) / g.cp;
    A(5, 4) = (g.sr * q + g.cr * r) * g.sp / cp2;
    A(5, 10) = g.sr / g.cp;
    A(5, 11) = g.cr / g.cp;
    // thrust direction w.r.t. roll, pitch, yaw
    A(6, 3) = acc * (-g.sr * g.sp * g.cy + g.cr * g.sy);
    A(7, 3) = acc * (-g.sr * g.sp * g.sy - g.cr * g.cy);
    A(8, 3) = acc * (-g.sr * g.cp);
    A(6, 4) = acc * (g.cr * g.cp * g.cy);
    A(7, 4) = acc * (g.cr * g.cp * g.sy);
    A(8, 4) = acc * (-g.cr * g.sp);
    A(6, 5) = acc * (-g.cr * g.sp * g.sy + g.sr * g.cy);
    A(7, 5) = acc * (g.cr * g.sp * g.cy + g.sr * g.sy);
    // gyroscopic coupling
    A(9, 10) = -(inertia_[2] - inertia_[1]) * r / inertia_[0];
    A(9, 11) = -(inertia_[2] - inertia_[1]) * q / inertia_[0];
    A(10, 9) = -(inertia_[0] - inertia_[2]) * r / inertia_[1];
    A(10, 11) = -(inertia_[0] - inertia_[2]) * p / inertia_[1];
    A(11, 9) = -(inertia_[1] - inertia_[0]) * q / inertia_[2];
    A(11, 10) = -(inertia_[1] - inertia_[0]) * p / inertia_[2];
    A *= dt_;
    A.addToDiagonal(Real(1));

    const Real bz[3] = {g.cr * g.sp * g.cy + g.sr * g.sy, g.cr * g.sp * g.sy - g.sr * g.cy, g.cr * g.cp};
    StateInputDimMatrix & Bm = state_eq_deriv_u;
    Bm.setZero();
    for(int i = 0; i < 4; i++)
    {
      Bm(6, i) = dt_ * (bz[0] / mass_);
      Bm(7, i) = dt_ * (bz[1] / mass_);
      Bm(8, i) = dt_ * (bz[2] / mass_);
      Bm(11, i) = dt_ * ((i % 2 == 0 ? yaw_coef_ : -yaw_coef_) / inertia_[2]);
    }
    Bm(9, 1) = dt_ * (arm_ / inertia_[0]);
    Bm(9, 3) = dt_ * (-arm_ / inertia_[0]);
    Bm(10, 2) = dt_ * (arm_ / inertia_[1]);
    Bm(10, 0) = dt_ * (-arm_ / inertia_[1]);
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
    for(int i = 0; i < 12; i++)
    {
      running_cost_deriv_x[i] = stateWeight(i) * stateError(x, i);
      running_cost_deriv_xx(i, i) = stateWeight(i);
    }
    running_cost_deriv_uu.setZero();
    for(int i = 0; i < 4; i++)
    {
      running_cost_deriv_u[i] = w_u_ * (u[i] - hoverThrust());
      running_cost_deriv_uu(i, i) = w_u_;
    }
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(Real, // t
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_xx.setZero();
    for(int i = 0; i < 12; i++)
    {
      terminal_cost_deriv_x[i] = (wt_scale_ * stateWeight(i)) * stateError(x, i);
      terminal_cost_deriv_xx(i, i) = wt_scale_ * stateWeight(i);
    }
  }

public:
  Real mass_ = Real(1.0); // [kg]
  Real inertia_[3] = {Real(0.01), Real(0.01), Real(0.02)}; // [kg m^2]
  Real arm_ = Real(0.2); // [m]
  Real yaw_coef_ = Real(0.05); // rotor drag torque per unit thrust [m]
  Real w_pos_ = Real(1.0), w_rpy_ = Real(0.5), w_vel_ = Real(0.1), w_omega_ = Real(0.05);
  Real w_u_ = Real(0.01);
  Real wt_scale_ = Real(10.0); // terminal weight = wt_scale * running weight
  Real ref_pos_[3] = {Real(0.0), Real(0.0), Real(1.0)}; // [m]
};

using DDPProblemQuadrotor = DDPProblemQuadrotorT<double>;
using DDPProblemQuadrotorF32 = DDPProblemQuadrotorT<float>;
} // namespace nmpc_amd
