// Cart-pole swing-up problem for the MI355X DDP solver.
// Same model as the reference's test problem DDPProblemCartPole
// (nmpc_ddp/tests/src/TestDDPCartPole.cpp:28-234): state [pos, theta, vel, omega], input [force],
// explicit-Euler dynamics, quadratic running / terminal costs.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
/** \tparam Real Real: the reference's arithmetic, registered as "cartpole"; float: "cartpole_f32", the fp32 tile kernel's
    n = 4, m = 1 shape (include/nmpc_amd/hip/ddp_kernels_tile32.hpp).  The statements are the same; every constant is converted
    to Real once. */
template<class Real>
class DDPProblemCartPoleT : public DDPProblemT<Real, 4, 1>
{
  using Base = DDPProblemT<Real, 4, 1>;
  using Base::dt_;

public:
  using typename Base::InputDimVector;
  using typename Base::InputInputDimMatrix;
  using typename Base::StateDimVector;
  using typename Base::StateInputDimMatrix;
  using typename Base::StateStateDimMatrix;
  struct Param
  {
    Real cart_mass = Real(1.0); // [kg]
    Real pole_mass = Real(0.5); // [kg]
    Real pole_length = Real(2.0); // [m]
  };

  struct CostWeight
  {
    Real running_x[4] = {Real(0.1), Real(1.0), Real(0.01), Real(0.1)};
    Real running_u[1] = {Real(0.001)};
    Real terminal_x[4] = {Real(0.1), Real(1.0), Real(0.01), Real(0.1)};
  };

  static constexpr const char * kName = sizeof(Real) == 8 ? "cartpole" : "cartpole_f32";
  static constexpr Real g_ = Real(9.80665); // [m/s^2]

  NMPC_HD explicit DDPProblemCartPoleT(Real dt = Real(0.01)) : Base(dt) {}

  /** Reference position of the cart (the reference's ref_pos_func_; its test uses a constant,
      TestDDPCartPole.cpp:363-376). */
  NMPC_HD Real refPos(Real /* t */) const
  {
    return ref_pos_;
  }

  NMPC_HD StateDimVector stateEq(Real, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u) const
  {
    return step<false>(x, u, dt_);
  }

  /** The plant step of the reference's test (TestDDPCartPole.cpp:63-98: stateEq with an explicit dt), used by the
      receding-horizon driver's plant pattern.  Full-range sin / cos: a simulated pole may wind up arbitrarily far, where the
      solver's restricted-range sincosFast would give NaN (its rollouts of diverging line-search candidates are rejected
      either way, the plant's state is kept). */
  NMPC_HD StateDimVector stateEq(Real, // t
                                 const StateDimVector & x,
                                 const InputDimVector & u,
                                 Real dt) const
  {
    return step<true>(x, u, dt);
  }

  template<bool kFullRange>
  NMPC_HD StateDimVector step(const StateDimVector & x, const InputDimVector & u, Real dt) const
  {
    const Real theta = x[1];
    const Real vel = x[2];
    const Real omega = x[3];
    const Real f = u[0];
    const Real m1 = param_.cart_mass;
    const Real m2 = param_.pole_mass;
    const Real l = param_.pole_length;

    Real sin_theta, cos_theta;
    if constexpr(kFullRange)
    {
      sincos(theta, sin_theta, cos_theta);
    }
    else
    {
      sincosFast(theta, sin_theta, cos_theta); // |theta| < 2^27 rad, NaN beyond (linalg.hpp)
    }
    const Real omega2 = omega * omega;
    const Real denom = m1 + m2 * (sin_theta * sin_theta);
    // one reciprocal instead of the two divisions of the textbook form (an fp64 divide costs ~10 FMAs on gfx950)
    const Real inv_denom = recipFast(denom); // denom >= cart mass > 0
    const Real inv_l = 1 / l;

    StateDimVector x_next;
    x_next[0] = x[0] + dt * vel;
    x_next[1] = x[1] + dt * omega;
    x_next[2] = x[2] + dt * ((f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta) * inv_denom);
    x_next[3] = x[3]
                + dt
                      * ((f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta)
                         * (inv_denom * inv_l));
    return x_next;
  }

  NMPC_HD Real runningCost(Real t, const StateDimVector & x, const InputDimVector & u) const
  {
    Real cost_x = 0;
    for(int i = 0; i < 4; i++)
    {
      const Real e = x[i] - (i == 0 ? refPos(t) : Real(0));
      cost_x += cost_weight_.running_x[i] * (e * e);
    }
    return Real(0.5) * cost_x + Real(0.5) * (cost_weight_.running_u[0] * (u[0] * u[0]));
  }

  NMPC_HD Real terminalCost(Real t, const StateDimVector & x) const
  {
    Real cost_x = 0;
    for(int i = 0; i < 4; i++)
    {
      const Real e = x[i] - (i == 0 ? refPos(t) : Real(0));
      cost_x += cost_weight_.terminal_x[i] * (e * e);
    }
    return Real(0.5) * cost_x;
  }

  NMPC_HD void calcStateEqDeriv(Real, // t
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const Real theta = x[1];
    const Real omega = x[3];
    const Real f = u[0];
    const Real m1 = param_.cart_mass;
    const Real m2 = param_.pole_mass;
    const Real l = param_.pole_length;

    Real sin_theta, cos_theta;
    sincosFast(theta, sin_theta, cos_theta); // |theta| < 2^27 rad, NaN beyond (linalg.hpp)
    const Real omega2 = omega * omega;
    const Real sin2 = sin_theta * sin_theta;
    const Real denom = m1 + m2 * sin2;
    const Real inv_denom = recipFast(denom); // denom >= cart mass > 0
    const Real inv_denom_sq = inv_denom * inv_denom;
    const Real inv_l = 1 / l;
    // numerators of the two accelerations and d(denom)/d(theta)
    const Real acc_num = f - m2 * l * omega2 * sin_theta + m2 * g_ * sin_theta * cos_theta;
    const Real alp_num = f * cos_theta - m2 * l * omega2 * sin_theta * cos_theta + g_ * (m1 + m2) * sin_theta;
    const Real ddenom = 2 * m2 * sin_theta * cos_theta;

    state_eq_deriv_x.setZero();
    state_eq_deriv_x(0, 2) = 1;
    state_eq_deriv_x(1, 3) = 1;
    state_eq_deriv_x(2, 1) =
        ((-1 * m2 * l * omega2 * cos_theta + m2 * g_ * (1 - 2 * sin2)) * denom + -1 * acc_num * ddenom) * inv_denom_sq;
    state_eq_deriv_x(2, 3) = (-2 * m2 * l * omega * sin_theta) * inv_denom;
    state_eq_deriv_x(3, 1) = ((-1 * f * sin_theta + -1 * m2 * l * omega2 * (1 - 2 * sin2) + g_ * (m1 + m2) * cos_theta)
                                  * denom
                              + -1 * alp_num * ddenom)
                             * (inv_denom_sq * inv_l);
    state_eq_deriv_x(3, 3) = (-2 * m2 * l * omega * sin_theta * cos_theta) * (inv_denom * inv_l);
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.addToDiagonal(Real(1));

    state_eq_deriv_u.setZero();
    state_eq_deriv_u[2] = inv_denom;
    state_eq_deriv_u[3] = cos_theta * (inv_denom * inv_l);
    state_eq_deriv_u *= dt_;
  }

  NMPC_HD void calcRunningCostDeriv(Real t,
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & running_cost_deriv_x,
                                    InputDimVector & running_cost_deriv_u,
                                    StateStateDimMatrix & running_cost_deriv_xx,
                                    InputInputDimMatrix & running_cost_deriv_uu,
                                    StateInputDimMatrix & running_cost_deriv_xu) const
  {
    running_cost_deriv_xx.setZero();
    for(int i = 0; i < 4; i++)
    {
      running_cost_deriv_x[i] = cost_weight_.running_x[i] * (x[i] - (i == 0 ? refPos(t) : Real(0)));
      running_cost_deriv_xx(i, i) = cost_weight_.running_x[i];
    }
    running_cost_deriv_u[0] = cost_weight_.running_u[0] * u[0];
    running_cost_deriv_uu(0, 0) = cost_weight_.running_u[0];
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(Real t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    terminal_cost_deriv_xx.setZero();
    for(int i = 0; i < 4; i++)
    {
      terminal_cost_deriv_x[i] = cost_weight_.terminal_x[i] * (x[i] - (i == 0 ? refPos(t) : Real(0)));
      terminal_cost_deriv_xx(i, i) = cost_weight_.terminal_x[i];
    }
  }

public:
  Param param_;
  CostWeight cost_weight_;
  Real ref_pos_ = Real(0); // [m]
};

using DDPProblemCartPole = DDPProblemCartPoleT<double>;
using DDPProblemCartPoleF32 = DDPProblemCartPoleT<float>;
} // namespace nmpc_amd
