// This file is DERIVED from nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp of isri-aist/NMPC (a statement-for-statement port of one
// of its test problem classes, kept as a conformance fixture for the Eigen-subset syntax; nothing else in this repository is).
// The original is distributed under the BSD 2-Clause License, whose notice is retained here as its terms require:
//
//   BSD 2-Clause License
//   Copyright (c) 2022, AIST-CNRS JRL.  All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
//   conditions are met:
//   1. Redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer.
//   2. Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following
//      disclaimer in the documentation and/or other materials provided with the distribution.
//
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING,
//   BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT
//   SHALL THE COPYRIGHT HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL
//   DAMAGES (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS
//   INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING
//   NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
//
// DDPProblemCentroidalMotion written against nmpc_amd::DDPProblem the way the reference writes it against
// nmpc_ddp::DDPProblem (nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp:24-210): the same statements in the same order, in
// the Eigen-subset syntax of include/nmpc_amd/linalg.hpp.  What has to change when a reference problem class is ported:
//   * Eigen::Ref<T> parameters become T & (and `const Eigen::Ref<const Vector3d> & com = ...` becomes `const auto com = ...`),
//   * the methods are NMPC_HD and non-virtual,
//   * std::function members (ref_stance_func_, ref_pos_func_) become member functions of t (no heap on the device),
//   * Eigen::Matrix3Xd becomes a matrix with a column capacity (here 16 ridges).
// tests/test_host_cpu.py::test_eigen_style_port_matches_the_shipped_model compares it with
// include/nmpc_amd/models/CentroidalMotion.hpp bit for bit (host) and compiles it for gfx950 (device).
#pragma once

#include <nmpc_amd/DDPProblem.hpp>

namespace port
{
using nmpc_amd::Dynamic;
using nmpc_amd::Matrix;
using Vector3d = Matrix<double, 3, 1>;
using Matrix3d = Matrix<double, 3, 3>;
using Matrix3Xd = Matrix<double, 3, 16, false, true>; // Eigen::Matrix3Xd with a capacity

NMPC_HD Matrix3d crossMat(const Vector3d & vec) // nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp:16-22
{
  Matrix3d mat;
  mat << 0, -vec.z(), vec.y(), vec.z(), 0, -vec.x(), -vec.y(), vec.x(), 0;
  return mat;
}

class DDPProblemCentroidalMotion : public nmpc_amd::DDPProblem<9, Dynamic, 16>
{
public:
  struct StanceData
  {
    //! Contact vertices
    Matrix3Xd vertices_mat;
    //! Force direction (i.e., friction pyramid ridge); the number of columns is the same as that of vertices_mat
    Matrix3Xd ridges_mat;
  };

  struct CostWeight
  {
    NMPC_HD CostWeight()
    {
      running_x << Vector3d::Constant(1.0), Vector3d::Constant(0.0), Vector3d::Constant(1.0);
      running_u = 1e-6;
      terminal_x << Vector3d::Constant(1.0), Vector3d::Constant(0.0), Vector3d::Constant(1.0);
    }

    StateDimVector running_x;
    double running_u;
    StateDimVector terminal_x;
  };

public:
  NMPC_HD explicit DDPProblemCentroidalMotion(double dt = 0.03, const CostWeight & cost_weight = CostWeight())
  : DDPProblem(dt), cost_weight_(cost_weight)
  {
  }

  // ---- the reference's std::function members (TestDDPCentroidalMotion.cpp:248-281) as functions of t
  NMPC_HD static StanceData makeStanceDataFromRect(double min_x, double min_y, double max_x, double max_y) // :212-246
  {
    Vector3d vertex_list[4];
    vertex_list[0] << min_x, min_y, 0.0;
    vertex_list[1] << min_x, max_y, 0.0;
    vertex_list[2] << max_x, max_y, 0.0;
    vertex_list[3] << max_x, min_y, 0.0;

    Vector3d ridge_list[4];
    for(int i = 0; i < 4; i++)
    {
      double theta = 2 * M_PI * (static_cast<double>(i) / 4);
      ridge_list[i] << 0.5 * cos(theta), 0.5 * sin(theta), 1;
      ridge_list[i].normalize();
    }

    StanceData stance_data;
    stance_data.vertices_mat.resize(3, 16);
    stance_data.ridges_mat.resize(3, 16);
    int col_idx = 0;
    for(const auto & vertex : vertex_list)
    {
      for(const auto & ridge : ridge_list)
      {
        stance_data.vertices_mat.col(col_idx) = vertex;
        stance_data.ridges_mat.col(col_idx) = ridge;
        col_idx++;
      }
    }
    return stance_data;
  }
  NMPC_HD StanceData ref_stance_func_(double t) const
  {
    t += 1e-6;
    if(t < 1.4)
    {
      return makeStanceDataFromRect(-0.1, -0.1, 0.1, 0.1);
    }
    else if(t < 1.6)
    {
      StanceData stance_data;
      stance_data.vertices_mat.resize(3, 0);
      stance_data.ridges_mat.resize(3, 0);
      return stance_data;
    }
    else
    {
      return makeStanceDataFromRect(0.4, -0.1, 0.6, 0.1);
    }
  }
  NMPC_HD Vector3d ref_pos_func_(double t) const
  {
    t += 1e-6;
    Vector3d ref_pos;
    if(t < 1.5)
    {
      ref_pos << 0.0, 0.0, 1.0;
    }
    else
    {
      ref_pos << 0.5, 0.0, 1.0;
    }
    return ref_pos;
  }

  using DDPProblem::inputDim;

  NMPC_HD int inputDim(double t) const
  {
    const StanceData & stance_data = ref_stance_func_(t);
    return static_cast<int>(stance_data.vertices_mat.cols());
  }

  NMPC_HD StateDimVector stateEq(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    const StanceData & stance_data = ref_stance_func_(t);
    const Matrix3Xd & vertices_mat = stance_data.vertices_mat;
    const Matrix3Xd & ridges_mat = stance_data.ridges_mat;

    const auto com = x.segment<3>(0);
    const auto linear_momentum = x.segment<3>(3);
    // const auto angular_momentum = x.segment<3>(6);

    StateDimVector x_dot;
    auto com_dot = x_dot.segment<3>(0);
    auto linear_momentum_dot = x_dot.segment<3>(3);
    auto angular_momentum_dot = x_dot.segment<3>(6);
    com_dot = linear_momentum / mass_;
    linear_momentum_dot = ridges_mat * u - mass_ * g_;
    angular_momentum_dot.setZero();
    for(int i = 0; i < u.size(); i++)
    {
      angular_momentum_dot += u[i] * (vertices_mat.col(i) - com).cross(ridges_mat.col(i));
    }

    return x + dt_ * x_dot;
  }

  NMPC_HD double runningCost(double t, const StateDimVector & x, const InputDimVector & u) const
  {
    StateDimVector x_diff;
    x_diff << x.head<3>() - ref_pos_func_(t), x.tail<6>();
    return 0.5 * cost_weight_.running_x.dot(x_diff.cwiseAbs2()) + 0.5 * cost_weight_.running_u * u.squaredNorm();
  }

  NMPC_HD double terminalCost(double t, const StateDimVector & x) const
  {
    StateDimVector x_diff;
    x_diff << x.head<3>() - ref_pos_func_(t), x.tail<6>();
    return 0.5 * cost_weight_.terminal_x.dot(x_diff.cwiseAbs2());
  }

  NMPC_HD void calcStateEqDeriv(double t,
                                const StateDimVector & x,
                                const InputDimVector & u,
                                StateStateDimMatrix & state_eq_deriv_x,
                                StateInputDimMatrix & state_eq_deriv_u) const
  {
    const StanceData & stance_data = ref_stance_func_(t);
    const Matrix3Xd & vertices_mat = stance_data.vertices_mat;
    const Matrix3Xd & ridges_mat = stance_data.ridges_mat;

    const auto com = x.segment<3>(0);

    state_eq_deriv_x.setZero();
    state_eq_deriv_x.block<3, 3>(0, 3).diagonal().setConstant(1 / mass_);
    state_eq_deriv_x.block<3, 3>(6, 0) = crossMat(ridges_mat * u);
    state_eq_deriv_x *= dt_;
    state_eq_deriv_x.diagonal().array() += 1.0;

    state_eq_deriv_u.resize(9, u.size()); // (Eigen::Ref arrives sized; a capacity matrix is told its extent)
    state_eq_deriv_u.setZero();
    state_eq_deriv_u.middleRows<3>(3) = ridges_mat;
    for(int i = 0; i < u.size(); i++)
    {
      state_eq_deriv_u.middleRows<3>(6).col(i) = (vertices_mat.col(i) - com).cross(ridges_mat.col(i));
    }
    state_eq_deriv_u *= dt_;
  }

  NMPC_HD void calcRunningCostDeriv(double t,
                                    const StateDimVector & x,
                                    const InputDimVector & u,
                                    StateDimVector & running_cost_deriv_x,
                                    InputDimVector & running_cost_deriv_u) const
  {
    StateDimVector x_diff;
    x_diff << x.head<3>() - ref_pos_func_(t), x.tail<6>();
    running_cost_deriv_x = cost_weight_.running_x.cwiseProduct(x_diff);
    running_cost_deriv_u = cost_weight_.running_u * u;
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
    calcRunningCostDeriv(t, x, u, running_cost_deriv_x, running_cost_deriv_u);

    running_cost_deriv_xx = cost_weight_.running_x.asDiagonal();
    running_cost_deriv_uu.resize(u.size(), u.size());
    running_cost_deriv_uu.setIdentity();
    running_cost_deriv_uu *= cost_weight_.running_u;
    running_cost_deriv_xu.resize(9, u.size());
    running_cost_deriv_xu.setZero();
  }

  NMPC_HD void calcTerminalCostDeriv(double t, const StateDimVector & x, StateDimVector & terminal_cost_deriv_x) const
  {
    StateDimVector x_diff;
    x_diff << x.head<3>() - ref_pos_func_(t), x.tail<6>();
    terminal_cost_deriv_x = cost_weight_.terminal_x.cwiseProduct(x_diff);
  }

  NMPC_HD void calcTerminalCostDeriv(double t,
                                     const StateDimVector & x,
                                     StateDimVector & terminal_cost_deriv_x,
                                     StateStateDimMatrix & terminal_cost_deriv_xx) const
  {
    calcTerminalCostDeriv(t, x, terminal_cost_deriv_x);
    terminal_cost_deriv_xx = cost_weight_.terminal_x.asDiagonal();
  }

public:
  static constexpr const char * kName = "centroidal_eigen_style";
  const Vector3d g_ = Vector3d(0, 0, 9.80665); // [m/s^2]
  CostWeight cost_weight_;
  double mass_ = 100.0; // [kg]
};
} // namespace port
