// Problem functor API of the MI355X DDP solver.
//
// Keeps the interface of the reference's abstract class nmpc_ddp::DDPProblem<StateDim, InputDim>
// (nmpc_ddp/include/nmpc_ddp/DDPProblem.h:15-203) method for method — same names, same argument order, same
// meaning — so a user model carries over by
//   * deriving from nmpc_amd::DDPProblem<StateDim, InputDim[, MaxInputDim]> instead of nmpc_ddp::DDPProblem,
//   * marking the methods NMPC_HD (host + device, non-virtual: the solver is instantiated on the concrete type),
//   * replacing Eigen::Ref<> out-parameters by plain references to the types below,
//   * keeping every member trivially copyable (no std::function: a time-varying reference is written as a
//     function of t, exactly what the std::function bodies in the reference's tests are).
//
// Methods a problem must provide (DDPProblem.h:99-198; the overloads the solver never calls — the 2nd-order
// dynamics overload :139-146, which the reference solver rejects with std::runtime_error at
// DDPSolver.hpp:391-414, and the 1st-order cost overloads :155-159, :185-187 — are optional here):
//
//   StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const;
//   double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const;
//   double terminalCost(double t, const StateDimVector & x) const;
//   void calcStateEqDeriv(double t, const StateDimVector & x, const InputDimVector & u,
//                         StateStateDimMatrix & state_eq_deriv_x, StateInputDimMatrix & state_eq_deriv_u) const;
//   void calcRunningCostDeriv(double t, const StateDimVector & x, const InputDimVector & u,
//                             StateDimVector & running_cost_deriv_x, InputDimVector & running_cost_deriv_u,
//                             StateStateDimMatrix & running_cost_deriv_xx, InputInputDimMatrix & running_cost_deriv_uu,
//                             StateInputDimMatrix & running_cost_deriv_xu) const;
//   void calcTerminalCostDeriv(double t, const StateDimVector & x, StateDimVector & terminal_cost_deriv_x,
//                              StateStateDimMatrix & terminal_cost_deriv_xx) const;
//   int inputDim(double t) const;     // only if InputDim == nmpc_amd::Dynamic
#pragma once

#include <nmpc_amd/linalg.hpp>

namespace nmpc_amd
{
/** \brief DDP problem, arithmetic type as a template parameter.
    The reference computes in double throughout (DDPProblem.h:20-35); BASELINE.json's config 4 narrows the same
    interface to fp32, so the scalar type is the first template parameter here and nmpc_amd::DDPProblem below fixes
    it to double (the reference's signature).
    \tparam ScalarT double or float
    \tparam StateDim state dimension (fixed only)
    \tparam InputDim input dimension (fixed, or nmpc_amd::Dynamic)
    \tparam MaxInputDim capacity of the input dimension when InputDim is Dynamic (ignored otherwise) */
template<class ScalarT, int StateDim, int InputDim, int MaxInputDim = InputDim>
class DDPProblemT
{
  static_assert(StateDim > 0, "[DDP] Template param StateDim should be positive.");
  static_assert(InputDim >= 0 || InputDim == Dynamic,
                "[DDP] Template param InputDim should be non-negative or nmpc_amd::Dynamic.");
  static_assert(InputDim != Dynamic || MaxInputDim >= 0, "[DDP] Dynamic input dimension needs MaxInputDim.");

public:
  using Scalar = ScalarT;
  static constexpr int kStateDim = StateDim;
  static constexpr bool kDynamicInput = (InputDim == Dynamic);
  static constexpr int kInputDimMax = kDynamicInput ? MaxInputDim : InputDim;

  /** \brief Type of vector of state dimension. */
  using StateDimVector = Matrix<Scalar, StateDim, 1>;
  /** \brief Type of vector of input dimension. */
  using InputDimVector = Matrix<Scalar, kInputDimMax, 1, kDynamicInput, false>;
  /** \brief Type of matrix of state x state dimension. */
  using StateStateDimMatrix = Matrix<Scalar, StateDim, StateDim>;
  /** \brief Type of matrix of input x input dimension. */
  using InputInputDimMatrix = Matrix<Scalar, kInputDimMax, kInputDimMax, kDynamicInput, kDynamicInput>;
  /** \brief Type of matrix of state x input dimension. */
  using StateInputDimMatrix = Matrix<Scalar, StateDim, kInputDimMax, false, kDynamicInput>;
  /** \brief Type of matrix of input x state dimension. */
  using InputStateDimMatrix = Matrix<Scalar, kInputDimMax, StateDim, kDynamicInput, false>;

  /** \brief Constructor.
      \param dt discretization timestep [sec] */
  NMPC_HD explicit DDPProblemT(Scalar dt) : dt_(dt) {}

  /** \brief Gets the state dimension. */
  NMPC_HD static constexpr int stateDim()
  {
    return StateDim;
  }

  /** \brief Gets the input dimension (capacity when the dimension is dynamic; the reference throws there,
      DDPProblem.h:61-68 — device code cannot, so callers must use inputDim(t)). */
  NMPC_HD static constexpr int inputDim()
  {
    return kInputDimMax;
  }

  /** \brief Gets the input dimension at time t.  Must be shadowed by the problem if InputDim is Dynamic. */
  NMPC_HD int inputDim(Scalar) const
  {
    return kInputDimMax;
  }

  /** \brief Gets the discretization timestep [sec]. */
  NMPC_HD Scalar dt() const
  {
    return dt_;
  }

  Scalar dt_ = 0;
};

/** \brief DDP problem in the reference's arithmetic (double): nmpc_ddp::DDPProblem<StateDim, InputDim>
    (DDPProblem.h:15-203). */
template<int StateDim, int InputDim, int MaxInputDim = InputDim>
class DDPProblem : public DDPProblemT<double, StateDim, InputDim, MaxInputDim>
{
public:
  NMPC_HD explicit DDPProblem(double dt) : DDPProblemT<double, StateDim, InputDim, MaxInputDim>(dt) {}
};
} // namespace nmpc_amd
