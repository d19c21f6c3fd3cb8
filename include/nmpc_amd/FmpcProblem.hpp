// Problem functor API of the MI355X FMPC solver.
//
// Keeps the interface of the reference's abstract class nmpc_fmpc::FmpcProblem<StateDim, InputDim, IneqDim>
// (nmpc_fmpc/include/nmpc_fmpc/FmpcProblem.h:15-132), which is nmpc_ddp::DDPProblem<StateDim, InputDim> plus the inequality
// constraints g(t, x, u) <= 0 and their first derivatives.  A user problem carries over as described in DDPProblem.hpp
// (derive from nmpc_amd::FmpcProblem, NMPC_HD non-virtual methods, plain references instead of Eigen::Ref, trivially copyable
// members).  In addition to the DDPProblem methods a problem provides (FmpcProblem.h:94-109):
//
//   IneqDimVector ineqConst(double t, const StateDimVector & x, const InputDimVector & u) const;
//   void calcIneqConstDeriv(double t, const StateDimVector & x, const InputDimVector & u,
//                           IneqStateDimMatrix & ineq_const_deriv_x, IneqInputDimMatrix & ineq_const_deriv_u) const;
//
// The second-order overload of calcStateEqDeriv is private and throws in the reference (FmpcProblem.h:113-131): it does not
// exist here.  Fixed dimensions only: the reference's Eigen::Dynamic InputDim / IneqDim (FmpcProblem.h:9-10,
// FmpcSolver.hpp:211-218) is not offered by this build (neither of the reference's FMPC problems uses it).
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace nmpc_amd
{
/** \brief Fast MPC problem.
    \tparam StateDim state dimension
    \tparam InputDim input dimension
    \tparam IneqDim inequality dimension */
template<int StateDim, int InputDim, int IneqDim>
class FmpcProblem : public DDPProblem<StateDim, InputDim>
{
  static_assert(InputDim >= 0, "[FMPC] Template param InputDim should be non-negative (dynamic dimensions are not offered).");
  static_assert(IneqDim >= 0, "[FMPC] Template param IneqDim should be non-negative (dynamic dimensions are not offered).");

public:
  static constexpr int kIneqDim = IneqDim;

  /** \brief Type of vector of inequality dimension. */
  using IneqDimVector = Matrix<double, IneqDim, 1>;
  /** \brief Type of matrix of inequality x state dimension. */
  using IneqStateDimMatrix = Matrix<double, IneqDim, StateDim>;
  /** \brief Type of matrix of inequality x input dimension. */
  using IneqInputDimMatrix = Matrix<double, IneqDim, InputDim>;

  /** \brief Constructor.
      \param dt discretization timestep [sec] */
  NMPC_HD explicit FmpcProblem(double dt) : DDPProblem<StateDim, InputDim>(dt) {}

  /** \brief Gets the inequality dimension. */
  NMPC_HD static constexpr int ineqDim()
  {
    return IneqDim;
  }

  /** \brief Gets the inequality dimension at time t (FmpcProblem.h:76-87). */
  NMPC_HD int ineqDim(double) const
  {
    return IneqDim;
  }
};
} // namespace nmpc_amd
