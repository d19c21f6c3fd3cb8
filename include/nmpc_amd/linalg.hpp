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
// full unrolling of the small fixed-trip loops below: clang / hipcc spelling, GCC spelling for host-only builds
#if defined(__clang__)
#  define NMPC_UNROLL _Pragma("unroll")
#elif defined(__GNUC__)
#  define NMPC_UNROLL _Pragma("GCC unroll 64")
#else
#  define NMPC_UNROLL
#endif

namespace nmpc_amd
{
/** sin and cos of the same angle, restricted range: |x| < 2^27 rad (2.1e7 revolutions); outside (and for
    NaN / Inf) both results are NaN — loud, never a silently wrong value.  Branch-free, ~32 instructions:
    four-term Cody-Waite reduction by pi/2 with FMAs (three 26-bit pieces + a 53-bit tail = 131 bits of pi/2; the
    products k * piece are exact for |k| < 2^27), then the degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].
    Measured against long-double references: <= 1.5 ulp for |x| <= 1e3, <= 2.5 ulp up to 1e8
    (tests/test_host_cpu.py).  On gfx950 the device math library's sin / cos cost ~320 cycles per wavefront each
    and its sincos ~340 (profiles/); this costs ~130.  Problem functors whose angles are physical (joint, pole,
    attitude angles) should use this; sincos() below is the full-range version. */
NMPC_HD void sincosFast(double x, double & s, double & c)
{
  constexpr double kTwoOverPi = 6.36619772367581382433e-01;
  constexpr double kP1 = 0x1.921fb50000000p+0; // pi/2, bits 1..26
  constexpr double kP2 = 0x1.110b460000000p-26; // bits 27..52
  constexpr double kP3 = 0x1.1a62630000000p-54; // bits 53..78
  constexpr double kP4 = 0x1.8a2e03707344ap-81; // remainder
  const double k = rint(x * kTwoOverPi);
  double r = fma(-k, kP1, x);
  r = fma(-k, kP2, r);
  r = fma(-k, kP3, r);
  r = fma(-k, kP4, r);
  const double z = r * r;
  // sin(r) = r + r z (S1 + z (S2 + ... )),  cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double cr = fma(z, fma(z, pc, -0.5), 1.0);
  // quadrant: x = r + k pi/2; out of range => NaN
  const bool in_range = fabs(x) < 134217728.0;
  const int q = static_cast<int>(k) & 3;
  const double s0 = (q & 1) ? cr : sr;
  const double c0 = (q & 1) ? sr : cr;
  const double nan = __builtin_nan("");
  s = in_range ? ((q & 2) ? -s0 : s0) : nan;
  c = in_range ? (((q + 1) & 2) ? -c0 : c0) : nan;
}

/** sin and cos of the same angle, full range: sincosFast inside |x| < 2^27, the math library beyond. */
NMPC_HD void sincos(double x, double & s, double & c)
{
  if(__builtin_expect(fabs(x) < 134217728.0, 1))
  {
    sincosFast(x, s, c);
  }
  else
  {
    ::sincos(x, &s, &c);
  }
}

/** 1 / x for a finite, normal-range x != 0 (a pivot, a mass-matrix determinant, a norm + 1): on gfx950 the hardware
    reciprocal estimate plus two Newton steps — 5 instructions; identical to the IEEE divide on all 4.2 M arguments of
    scripts/ubench_recip.hip (magnitudes 2^-40 .. 2^41, measured on MI355X) — instead of the 11-instruction correctly-rounded IEEE divide sequence (~72 cycles for a lone
    wavefront).  Zero, infinity and NaN give NaN (the IEEE divide would give Inf / 0 / NaN): callers test their
    argument first where that matters (the pivot test of ldltInPlace does).  On the host: the plain divide. */
NMPC_HD double recipFast(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

/** Single-precision sin and cos of the same angle for |x| < 2^15 rad (NaN beyond, like the double version): three-term
    Cody-Waite reduction by pi/2 (8 + 11 + 24 bits, k * piece exact for |k| < 2^16) with FMAs, then the classic degree-7 /
    degree-8 minimax kernels on [-pi/4, pi/4] (<= 2 ulp there).  ~20 instructions, branch-free; used by the fp32 problem
    types (BASELINE.json config 4). */
NMPC_HD void sincosFast(float x, float & s, float & c)
{
  constexpr float kTwoOverPi = 0.636619772367581343f;
  constexpr float kP1 = 1.5703125f; // pi/2, leading 8 bits
  constexpr float kP2 = 4.837512969970703125e-4f;
  constexpr float kP3 = 7.54978995489188216e-8f;
  const float k = rintf(x * kTwoOverPi);
  float r = fmaf(-k, kP1, x);
  r = fmaf(-k, kP2, r);
  r = fmaf(-k, kP3, r);
  const float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float sr = fmaf(r * z, ps, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
  const bool in_range = fabsf(x) < 32768.0f;
  const int q = static_cast<int>(k) & 3;
  const float s0 = (q & 1) ? cr : sr;
  const float c0 = (q & 1) ? sr : cr;
  const float nan = __builtin_nanf("");
  s = in_range ? ((q & 2) ? -s0 : s0) : nan;
  c = in_range ? (((q + 1) & 2) ? -c0 : c0) : nan;
}

/** Single precision 1 / x for a finite, normal-range x != 0: the hardware estimate (1 ulp) plus one Newton step on
    gfx950, the plain divide on the host. */
NMPC_HD float recipFast(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
#else
  return 1.0f / x;
#endif
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
    NMPC_UNROLL
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
    NMPC_UNROLL
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
    NMPC_UNROLL
    for(int i = 0; i < kDiag; i++)
    {
      if(i < rows() && i < cols())
      {
        d_[i + i * RMAX] += v;
      }
    }
    return *this;
  }
  /** Scaling leaves STRUCTURAL zeros (entries the compiler knows to be literal 0, e.g. after setZero()) untouched
      instead of turning them into the run-time value 0 * s, which IEEE arithmetic does not allow the compiler to
      fold.  This keeps the reference's idiom `m.setZero(); m(i,j) = ...; m *= dt_; m.diagonal() += 1`
      (TestDDPCartPole.cpp:136-153) sparse for the solver kernels (ddp_kernels.hpp, macc()); exact for finite s. */
  NMPC_HD Matrix & operator*=(Scalar s)
  {
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      if(!(__builtin_constant_p(d_[i]) && d_[i] == Scalar(0)))
      {
        d_[i] *= s;
      }
    }
    return *this;
  }
  NMPC_HD Matrix & operator+=(const Matrix & o)
  {
    NMPC_UNROLL
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
    NMPC_UNROLL
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
    NMPC_UNROLL
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
