// Conformance fixture for the Eigen-subset syntax of include/nmpc_amd/linalg.hpp: a problem class written the
// way a user of nmpc_ddp writes one — comma initialisers, segment<3> / head / tail views, block<3, 3>, middleRows<3>, col(),
// cross, asDiagonal, cwiseAbs2 / cwiseProduct, dot / squaredNorm, normalize, a matrix with a run-time number of columns and a
// run-time input dimension.  The problem itself is this repository's own (it exists only for this test): a gyrostat — a rigid
// body with a momentum wheel cluster — steered by a set of cold-gas jets, of which a time-dependent subset is armed.
//
//   state  x = [ r (3): attitude error vector | w (3): body rate | h (3): wheel-cluster momentum ]
//   input  u = thrust of each armed jet (inputDim(t) of them; none during the coast window)
//   r' = w + r x w / 2
//   w' = J^-1 (G u - w x (J w + h)),    G.col(k) = mount_k x axis_k  (torque arm of jet k)
//   h' = wheel_gain w - wheel_leak h
//   explicit Euler, x+ = x + dt x'
//
// tests/cpp/JetGyrostatPlain.hpp states the same arithmetic entry by entry on scalars; tests/cpp/test_eigen_style_port.cpp
// demands the two agree bit for bit, on the host and in a gfx950 kernel.
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace conformance
{
using nmpc_amd::Dynamic;
using nmpc_amd::Matrix;
using Vector3d = Matrix<double, 3, 1>;
using Matrix3d = Matrix<double, 3, 3>;
constexpr int kJetCapacity = 8;
using Matrix3Xd = Matrix<double, 3, kJetCapacity, false, true>; // three rows, a run-time number of columns

//! v -> [v]x with [v]x a = v x a
NMPC_HD Matrix3d skew(const Vector3d & v)
{
  Matrix3d s;
  s << 0, -v.z(), v.y(), v.z(), 0, -v.x(), -v.y(), v.x(), 0;
  return s;
}

class JetGyrostatEigenStyle : public nmpc_amd::DDPProblem<9, Dynamic, kJetCapacity>
{
public:
  struct JetSet
  {
    Matrix3Xd mounts; // where each armed jet sits (body frame)
    Matrix3Xd axes; // unit thrust direction of each armed jet
  };

  NMPC_HD explicit JetGyrostatEigenStyle(double dt = 0.05) : DDPProblem(dt)
  {
    inertia_ << 2.4, 3.1, 1.7;
    running_weight_ << Vector3d::Constant(4.0), Vector3d(0.5, 0.25, 0.5), Vector3d::Constant(0.01);
    terminal_weight_ << Vector3d::Constant(40.0), Vector3d::Constant(2.0), Vector3d::Constant(0.1);
  }

  //! the jets armed at time t: a ring of eight early on, none while coasting, the four of the +z deck afterwards
  NMPC_HD JetSet armedJets(double t) const
  {
    t += 1e-6;
    JetSet jets;
    const int count = t < 1.0 ? 8 : (t < 1.5 ? 0 : 4);
    jets.mounts.resize(3, count);
    jets.axes.resize(3, count);
    for(int k = 0; k < count; k++)
    {
      const double phi = 0.25 * M_PI * k + 0.1;
      Vector3d mount, axis;
      mount << 0.6 * cos(phi), 0.6 * sin(phi), (k % 2 == 0 ? 0.3 : -0.3);
      axis << -sin(phi), cos(phi), 0.4 * (count == 4 ? 1.0 : -1.0);
      axis.normalize();
      jets.mounts.col(k) = mount;
      jets.axes.col(k) = axis;
    }
    return jets;
  }

  //! attitude / rate reference: a slow nod about the body y axis
  NMPC_HD StateDimVector reference(double t) const
  {
    StateDimVector ref;
    ref << Vector3d(0.0, 0.2 * sin(0.5 * t), 0.0), Vector3d(0.0, 0.1 * cos(0.5 * t), 0.0), Vector3d::Zero();
    return ref;
  }

  using DDPProblem::inputDim;

  NMPC_HD int inputDim(double t) const
  {
    return static_cast<int>(armedJets(t).axes.cols());
  }

  NMPC_HD Matrix3Xd torqueArms(const JetSet & jets) const
  {
    Matrix3Xd arms;
    arms.resize(3, jets.axes.cols());
    for(int k = 0; k < jets.axes.cols(); k++)
    {
      arms.col(k) = jets.mounts.col(k).cross(jets.axes.col(k));
    }
    return arms;
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const JetSet jets = armedJets(t);
    const Matrix3Xd arms = torqueArms(jets);
    const auto r = x.segment<3>(0);
    const auto w = x.segment<3>(3);
    const auto h = x.tail<3>();

    StateDimVector x_dot;
    auto r_dot = x_dot.head<3>();
    auto w_dot = x_dot.segment<3>(3);
    auto h_dot = x_dot.tail<3>();
    r_dot = w + r.cross(w) * 0.5;
    const Vector3d stored = inertia_.cwiseProduct(w) + h;
    Vector3d torque = arms * u - w.cross(stored);
    for(int a = 0; a < 3; a++)
    {
      torque[a] = torque[a] / inertia_[a];
    }
    w_dot = torque;
    h_dot = wheel_gain_ * w - wheel_leak_ * h;
    return x + dt_ * x_dot;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const StateDimVector miss = x - reference(t);
    return 0.5 * running_weight_.dot(miss.cwiseAbs2()) + 0.5 * thrust_weight_ * u.squaredNorm();
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    const StateDimVector miss = x - reference(t);
    return 0.5 * terminal_weight_.dot(miss.cwiseAbs2());
  }

  NMPC_HD void calcStateEqDeriv(double t,
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const JetSet jets = armedJets(t);
    const Matrix3Xd arms = torqueArms(jets);
    const auto r = x.segment<3>(0);
    const auto w = x.segment<3>(3);
    const auto h = x.tail<3>();
    const Vector3d stored = inertia_.cwiseProduct(w) + h;

    Matrix3d inertia_mat, inv_inertia_mat;
    inertia_mat = inertia_.asDiagonal();
    Vector3d inv_inertia;
    inv_inertia << 1.0 / inertia_[0], 1.0 / inertia_[1], 1.0 / inertia_[2];
    inv_inertia_mat = inv_inertia.asDiagonal();

    state_eq_deriv_x.setZero();
    state_eq_deriv_x.block<3, 3>(0, 0) = skew(w) * -0.5; // d(r x w / 2) / dr
    state_eq_deriv_x.block<3, 3>(0, 3) = Matrix3d::Identity() + skew(r) * 0.5;
    state_eq_deriv_x.block<3, 3>(3, 3) = inv_inertia_mat * (skew(stored) - skew(w) * inertia_mat);
    state_eq_deriv_x.block<3, 3>(3, 6) = inv_inertia_mat * -skew(w);
    state_eq_deriv_x.block<3, 3>(6, 3).diagonal().setConstant(wheel_gain_);
    state_eq_deriv_x.block<3, 3>(6, 6).diagonal().setConstant(-wheel_leak_);
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.diagonal().array() += 1.0;

    state_eq_deriv_u.resize(9, u.size());
    state_eq_deriv_u.setZero();
    state_eq_deriv_u.middleRows<3>(3) = dt_ * (inv_inertia_mat * arms);
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
    const StateDimVector miss = x - reference(t);
    running_cost_deriv_x = running_weight_.cwiseProduct(miss);
    running_cost_deriv_u = thrust_weight_ * u;
    running_cost_deriv_xx = running_weight_.asDiagonal();
    running_cost_deriv_uu.resize(u.size(), u.size());
    running_cost_deriv_uu.setZero();
    running_cost_deriv_uu.diagonal().setConstant(thrust_weight_);
    running_cost_deriv_xu.resize(9, u.size());
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    const StateDimVector miss = x - reference(t);
    terminal_cost_deriv_x = terminal_weight_.cwiseProduct(miss);
    terminal_cost_deriv_xx = terminal_weight_.asDiagonal();
  }

public:
  Vector3d inertia_; // principal moments of the body
  double wheel_gain_ = 0.3;
  double wheel_leak_ = 0.8;
  double thrust_weight_ = 1e-3;
  StateDimVector running_weight_;
  StateDimVector terminal_weight_;
};
} // namespace conformance
