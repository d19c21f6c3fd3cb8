// Small fixed-capacity dense types usable from host and gfx950 device code.
//
// The reference writes its problem classes against Eigen fixed / dynamic small matrices
// (nmpc_ddp/include/nmpc_ddp/DDPProblem.h:20-35).  Eigen is a host-only dependency the HIP path cannot use,
// so problem functors are written against this header instead: column-major like Eigen, compile-time
// CAPACITY, and — where the reference uses Eigen::Dynamic for the input dimension — a run-time extent that
// never exceeds the capacity (no heap, so an instance lives entirely in registers / LDS).
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#  define NMPC_HD __host__ __device__ __forceinline__
#else
#  define NMPC_HD inline
#endif

namespace nmpc_amd
{
/** sin and cos of the same angle in one call (one argument reduction instead of two: on gfx950 an fp64 sin or
    cos costs ~320 cycles per wavefront and the fused sincos ~340, profiles/ubench_r01.txt). */
NMPC_HD void sincos(double x, double & s, double & c)
{
  ::sincos(x, &s, &c);
}

//! Marker for a run-time input dimension (the reference's Eigen::Dynamic).
constexpr int Dynamic = -1;

namespace detail
{
template<bool DynRows, bool DynCols>
struct Extents
{
  NMPC_HD Extents(int, int) {}
  NMPC_HD void set(int, int) {}
};
template<>
struct Extents<true, false>
{
  int r_;
  NMPC_HD Extents(int r, int) : r_(r) {}
  NMPC_HD void set(int r, int)
  {
    r_ = r;
  }
};
template<>
struct Extents<false, true>
{
  int c_;
  NMPC_HD Extents(int, int c) : c_(c) {}
  NMPC_HD void set(int, int c)
  {
    c_ = c;
  }
};
template<>
struct Extents<true, true>
{
  int r_, c_;
  NMPC_HD Extents(int r, int c) : r_(r), c_(c) {}
  NMPC_HD void set(int r, int c)
  {
    r_ = r;
    c_ = c;
  }
};
} // namespace detail

/** Column-major matrix with capacity RMAX x CMAX (leading dimension RMAX) and optional run-time extents.
    \tparam DynRows rows() is a run-time value <= RMAX
    \tparam DynCols cols() is a run-time value <= CMAX */
template<class Scalar, int RMAX, int CMAX, bool DynRows = false, bool DynCols = false>
class Matrix : private detail::Extents<DynRows, DynCols>
{
  using Ext = detail::Extents<DynRows, DynCols>;

public:
  static constexpr int kRowsMax = RMAX;
  static constexpr int kColsMax = CMAX;
  static constexpr int kCapacity = (RMAX * CMAX > 0) ? RMAX * CMAX : 1;

  NMPC_HD Matrix() : Ext(RMAX, CMAX) {}
  //! Run-time sized constructor (vector: Matrix(n); matrix: Matrix(r, c)).
  NMPC_HD explicit Matrix(int r, int c = CMAX) : Ext(r, c) {}

  NMPC_HD int rows() const
  {
    if constexpr(DynRows)
    {
      return this->r_;
    }
    else
    {
      return RMAX;
    }
  }
  NMPC_HD int cols() const
  {
    if constexpr(DynCols)
    {
      return this->c_;
    }
    else
    {
      return CMAX;
    }
  }
  NMPC_HD int size() const
  {
    return rows() * cols();
  }
  NMPC_HD void resize(int r, int c = CMAX)
  {
    Ext::set(r, c);
  }

  NMPC_HD Scalar & operator()(int r, int c)
  {
    return d_[r + c * RMAX];
  }
  NMPC_HD const Scalar & operator()(int r, int c) const
  {
    return d_[r + c * RMAX];
  }
  //! Vector access (column vectors only).
  NMPC_HD Scalar & operator[](int i)
  {
    return d_[i];
  }
  NMPC_HD const Scalar & operator[](int i) const
  {
    return d_[i];
  }
  NMPC_HD Scalar * data()
  {
    return d_;
  }
  NMPC_HD const Scalar * data() const
  {
    return d_;
  }

  NMPC_HD Matrix & setConstant(Scalar v)
  {
#pragma unroll
    for(int i = 0; i < kCapacity; i++)
    {
      d_[i] = v;
    }
    return *this;
  }
  NMPC_HD Matrix & setZero()
  {
    return setConstant(Scalar(0));
  }
  NMPC_HD Matrix & setIdentity()
  {
    setZero();
    constexpr int kDiag = RMAX < CMAX ? RMAX : CMAX;
#pragma unroll
    for(int i = 0; i < kDiag; i++)
    {
      d_[i + i * RMAX] = Scalar(1);
    }
    return *this;
  }
  //! diag += v   (Eigen: m.diagonal().array() += v)
  NMPC_HD Matrix & addToDiagonal(Scalar v)
  {
    constexpr int kDiag = RMAX < CMAX ? RMAX : CMAX;
#pragma unroll
    for(int i = 0; i < kDiag; i++)
    {
      if(i < rows() && i < cols())
      {
        d_[i + i * RMAX] += v;
      }
    }
    return *this;
  }
  NMPC_HD Matrix & operator*=(Scalar s)
  {
#pragma unroll
    for(int i = 0; i < kCapacity; i++)
    {
      d_[i] *= s;
    }
    return *this;
  }
  NMPC_HD Matrix & operator+=(const Matrix & o)
  {
#pragma unroll
    for(int i = 0; i < kCapacity; i++)
    {
      d_[i] += o.d_[i];
    }
    return *this;
  }
  NMPC_HD Matrix operator+(const Matrix & o) const
  {
    Matrix r(*this);
    r += o;
    return r;
  }
  NMPC_HD Matrix operator-(const Matrix & o) const
  {
    Matrix r(*this);
#pragma unroll
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] -= o.d_[i];
    }
    return r;
  }
  NMPC_HD friend Matrix operator*(Scalar s, const Matrix & m)
  {
    Matrix r(m);
    r *= s;
    return r;
  }

  // ---- reductions over the valid extent (ascending index order) ----
  NMPC_HD Scalar sum() const
  {
    Scalar s = 0;
    for(int c = 0; c < cols(); c++)
    {
      for(int r = 0; r < rows(); r++)
      {
        s += (*this)(r, c);
      }
    }
    return s;
  }
  NMPC_HD Scalar dot(const Matrix & o) const
  {
    Scalar s = 0;
    for(int c = 0; c < cols(); c++)
    {
      for(int r = 0; r < rows(); r++)
      {
        s += (*this)(r, c) * o(r, c);
      }
    }
    return s;
  }
  NMPC_HD Scalar squaredNorm() const
  {
    return dot(*this);
  }
  NMPC_HD Scalar norm() const
  {
    return sqrt(squaredNorm());
  }
  NMPC_HD Matrix cwiseProduct(const Matrix & o) const
  {
    Matrix r(*this);
#pragma unroll
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] *= o.d_[i];
    }
    return r;
  }
  NMPC_HD Matrix cwiseAbs2() const
  {
    return cwiseProduct(*this);
  }

private:
  Scalar d_[kCapacity];
};

template<class Scalar, int N>
using Vector = Matrix<Scalar, N, 1>;
} // namespace nmpc_amd
