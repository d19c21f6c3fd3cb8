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
#if defined(__HIP_DEVICE_COMPILE__)
  // A lone wavefront pays ~4.5 cycles for EVERY instruction it issues, selects and integer logic included, and ~8.75 for
  // one that depends on the instruction before it (scripts/ubench_issue_cost.hip): out of range => NaN is one select on
  // the argument's high word (everything below propagates it) instead of four on the results, the quadrant signs are
  // XORs of shifted bits of k, and the two polynomials are written interleaved (two independent chains: the compiler keeps the order).  Same values as the
  // host branch below, bit for bit.
  const double xe = __hiloint2double((fabs(x) < 134217728.0) ? __double2hiint(x) : 0x7ff80000, __double2loint(x));
  const double k = rint(xe * kTwoOverPi);
  double r = fma(-k, kP1, xe);
  r = fma(-k, kP2, r);
  r = fma(-k, kP3, r);
  r = fma(-k, kP4, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  double pc = -1.13596475577881948265e-11;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  pc = fma(pc, z, 2.08757232129817482790e-09);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double rz = r * z;
  const double ch = fma(z, pc, -0.5);
  const double sr = fma(rz, ps, r);
  const double cr = fma(z, ch, 1.0);
  const int ki = static_cast<int>(k);
  const bool odd = (ki & 1) != 0;
  const double s0 = odd ? cr : sr;
  const double c0 = odd ? sr : cr;
  const int sign_s = static_cast<int>((static_cast<unsigned>(ki) << 30) & 0x80000000u); // q & 2
  const int sign_c = static_cast<int>((static_cast<unsigned>(ki + 1) << 30) & 0x80000000u); // (q + 1) & 2
  s = __hiloint2double(__double2hiint(s0) ^ sign_s, __double2loint(s0));
  c = __hiloint2double(__double2hiint(c0) ^ sign_c, __double2loint(c0));
  // The results leave as opaque values: which products of the CALLER's expressions the compiler contracts into FMAs
  // depends on what it sees its operands are made of (a select of negations, an integer detour, ...), and it decided
  // differently from kernel to kernel once the signs became XORs — the lane mappings then disagree in the last bit
  // (tests/test_gpu_parity.py).  Behind the (empty) statement every kernel sees the same two plain registers.
  asm("" : "+v"(s), "+v"(c));
#else
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
#endif
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

/** sin and cos of the same angle in float, full range (the plant step of the receding-horizon driver). */
NMPC_HD void sincos(float x, float & s, float & c)
{
  ::sincosf(x, &s, &c);
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

template<class Scalar, int RMAX, int CMAX, bool DynRows = false, bool DynCols = false>
class Matrix;

// ---------------------------------------------------------------------------------------------------------------
// The subset of Eigen's block / initialiser syntax the reference's problem classes are written in
// (nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp:64-136, TestDDPCartPole.cpp:100-227): segment / head / tail / block /
// middleRows / col / diagonal views, `m << a, b, c`, asDiagonal(), cross, transpose, products, cwiseMin / cwiseMax,
// Constant / Zero / Identity — evaluated eagerly on the small fixed-capacity types above (no expression templates: every
// operator returns a value, which after inlining is what Eigen's lazy products compile to for these sizes).  With it a
// reference problem body carries over statement for statement (the syntax is exercised by tests/cpp/JetGyrostatEigenStyle.hpp).
// ---------------------------------------------------------------------------------------------------------------
/** Writable view of R x (up to CMAX) entries of a column-major matrix: Eigen's Block / Ref for fixed R. */
template<class Scalar, int R, int CMAX>
class Block
{
public:
  NMPC_HD Block(Scalar * p, int ld, int cols = CMAX) : p_(p), ld_(ld), cols_(cols) {}
  NMPC_HD int rows() const
  {
    return R;
  }
  NMPC_HD int cols() const
  {
    return cols_;
  }
  NMPC_HD int size() const
  {
    return R * cols_;
  }
  NMPC_HD Scalar & operator()(int r, int c) const
  {
    return p_[r + c * ld_];
  }
  NMPC_HD Scalar & operator[](int i) const // vectors: a column (CMAX == 1) or a row (R == 1)
  {
    return (CMAX == 1) ? p_[i] : p_[i * ld_];
  }
  template<bool DR, bool DC>
  NMPC_HD const Block & operator=(const Matrix<Scalar, R, CMAX, DR, DC> & m) const
  {
    for(int c = 0; c < cols_; c++)
    {
      NMPC_UNROLL
      for(int r = 0; r < R; r++)
      {
        (*this)(r, c) = m(r, c);
      }
    }
    return *this;
  }
  NMPC_HD const Block & operator=(const Block & o) const
  {
    for(int c = 0; c < cols_; c++)
    {
      NMPC_UNROLL
      for(int r = 0; r < R; r++)
      {
        (*this)(r, c) = o(r, c);
      }
    }
    return *this;
  }
  template<bool DR, bool DC>
  NMPC_HD const Block & operator+=(const Matrix<Scalar, R, CMAX, DR, DC> & m) const
  {
    for(int c = 0; c < cols_; c++)
    {
      NMPC_UNROLL
      for(int r = 0; r < R; r++)
      {
        (*this)(r, c) += m(r, c);
      }
    }
    return *this;
  }
  template<bool DR, bool DC>
  NMPC_HD const Block & operator-=(const Matrix<Scalar, R, CMAX, DR, DC> & m) const
  {
    for(int c = 0; c < cols_; c++)
    {
      NMPC_UNROLL
      for(int r = 0; r < R; r++)
      {
        (*this)(r, c) -= m(r, c);
      }
    }
    return *this;
  }
  NMPC_HD const Block & setConstant(Scalar v) const
  {
    for(int c = 0; c < cols_; c++)
    {
      NMPC_UNROLL
      for(int r = 0; r < R; r++)
      {
        (*this)(r, c) = v;
      }
    }
    return *this;
  }
  NMPC_HD const Block & setZero() const
  {
    return setConstant(Scalar(0));
  }
  NMPC_HD Block<Scalar, R, 1> col(int c) const
  {
    return Block<Scalar, R, 1>(p_ + c * ld_, ld_);
  }
  /** The main diagonal as a strided vector view (setConstant, array() += v). */
  struct Diagonal
  {
    Scalar * p;
    int stride, n;
    NMPC_HD const Diagonal & setConstant(Scalar v) const
    {
      for(int i = 0; i < n; i++)
      {
        p[i * stride] = v;
      }
      return *this;
    }
    NMPC_HD const Diagonal & array() const
    {
      return *this;
    }
    NMPC_HD const Diagonal & operator+=(Scalar v) const
    {
      for(int i = 0; i < n; i++)
      {
        p[i * stride] += v;
      }
      return *this;
    }
    NMPC_HD Scalar & operator[](int i) const
    {
      return p[i * stride];
    }
  };
  NMPC_HD Diagonal diagonal() const
  {
    return Diagonal{p_, ld_ + 1, R < cols_ ? R : cols_};
  }
  /** The viewed entries as a value; arithmetic on a view goes through it. */
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> eval() const;
  NMPC_HD operator Matrix<Scalar, R, CMAX, false, false>() const;
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> operator-(const Matrix<Scalar, R, CMAX, false, false> & o) const;
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> operator+(const Matrix<Scalar, R, CMAX, false, false> & o) const;
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> operator*(Scalar v) const;
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> operator/(Scalar v) const;
  NMPC_HD Matrix<Scalar, R, CMAX, false, false> cross(const Matrix<Scalar, R, CMAX, false, false> & o) const;
  NMPC_HD Scalar dot(const Matrix<Scalar, R, CMAX, false, false> & o) const;

private:
  Scalar * p_;
  int ld_, cols_;
};

/** vec.asDiagonal(): assignable to a square matrix. */
template<class Scalar, int N>
struct DiagonalWrapper
{
  Scalar d[N > 0 ? N : 1];
};

/** `m << a, b, c;`  Scalars and vectors / matrices fill the target in Eigen's order: vectors entry by entry, matrices row by
    row with blocks placed left to right (only what the reference's problem classes use: scalars into anything, vectors into
    vectors). */
template<class Target>
class CommaInitializer
{
public:
  using Scalar = typename Target::ScalarType;
  NMPC_HD CommaInitializer(Target & t, int at) : t_(t), at_(at) {}
  NMPC_HD CommaInitializer & operator,(Scalar v)
  {
    put(v);
    return *this;
  }
  template<int RM, bool DR>
  NMPC_HD CommaInitializer & operator,(const Matrix<Scalar, RM, 1, DR, false> & v)
  {
    for(int i = 0; i < v.rows(); i++)
    {
      put(v[i]);
    }
    return *this;
  }
  template<int K>
  NMPC_HD CommaInitializer & operator,(const Block<Scalar, K, 1> & v)
  {
    for(int i = 0; i < K; i++)
    {
      put(v[i]);
    }
    return *this;
  }
  NMPC_HD void put(Scalar v)
  {
    if(Target::kColsMax == 1)
    {
      t_[at_] = v;
    }
    else
    {
      t_(at_ / t_.cols(), at_ % t_.cols()) = v; // row by row, as Eigen fills a matrix
    }
    at_++;
  }

private:
  Target & t_;
  int at_;
};

/** Column-major matrix with capacity RMAX x CMAX (leading dimension RMAX) and optional run-time extents.
    \tparam DynRows rows() is a run-time value <= RMAX
    \tparam DynCols cols() is a run-time value <= CMAX */
template<class Scalar, int RMAX, int CMAX, bool DynRows, bool DynCols>
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

  // ---- the Eigen subset (see Block above) ----
  using ScalarType = Scalar;
  //! Vector3(x, y, z)
  NMPC_HD Matrix(Scalar a, Scalar b, Scalar c) : Ext(RMAX, CMAX)
  {
    static_assert(RMAX == 3 && CMAX == 1, "three coefficients: a 3-vector");
    d_[0] = a;
    d_[1] = b;
    d_[2] = c;
  }
  NMPC_HD Scalar x() const
  {
    return d_[0];
  }
  NMPC_HD Scalar y() const
  {
    return d_[1];
  }
  NMPC_HD Scalar z() const
  {
    return d_[2];
  }
  NMPC_HD static Matrix Constant(Scalar v)
  {
    Matrix m;
    m.setConstant(v);
    return m;
  }
  NMPC_HD static Matrix Zero()
  {
    return Constant(Scalar(0));
  }
  NMPC_HD static Matrix Identity()
  {
    Matrix m;
    m.setIdentity();
    return m;
  }
  NMPC_HD CommaInitializer<Matrix> operator<<(Scalar v)
  {
    CommaInitializer<Matrix> ci(*this, 0);
    ci.put(v);
    return ci;
  }
  template<int RM, bool DR>
  NMPC_HD CommaInitializer<Matrix> operator<<(const Matrix<Scalar, RM, 1, DR, false> & v)
  {
    CommaInitializer<Matrix> ci(*this, 0);
    ci, v;
    return ci;
  }
  template<int K>
  NMPC_HD CommaInitializer<Matrix> operator<<(const Block<Scalar, K, 1> & v)
  {
    CommaInitializer<Matrix> ci(*this, 0);
    ci, v;
    return ci;
  }
  NMPC_HD Matrix & operator=(const DiagonalWrapper<Scalar, (RMAX < CMAX ? RMAX : CMAX)> & dw)
  {
    setZero();
    constexpr int kDiag = RMAX < CMAX ? RMAX : CMAX;
    NMPC_UNROLL
    for(int i = 0; i < kDiag; i++)
    {
      d_[i + i * RMAX] = dw.d[i];
    }
    return *this;
  }
  NMPC_HD DiagonalWrapper<Scalar, RMAX> asDiagonal() const
  {
    static_assert(CMAX == 1, "asDiagonal() of a vector");
    DiagonalWrapper<Scalar, RMAX> dw;
    NMPC_UNROLL
    for(int i = 0; i < RMAX; i++)
    {
      dw.d[i] = d_[i];
    }
    return dw;
  }
  // vector pieces: values from a const object, writable views from a mutable one
  template<int K>
  NMPC_HD Matrix<Scalar, K, 1> segment(int at) const
  {
    Matrix<Scalar, K, 1> r;
    NMPC_UNROLL
    for(int i = 0; i < K; i++)
    {
      r[i] = d_[at + i];
    }
    return r;
  }
  template<int K>
  NMPC_HD Block<Scalar, K, 1> segment(int at)
  {
    return Block<Scalar, K, 1>(d_ + at, RMAX);
  }
  template<int K>
  NMPC_HD Matrix<Scalar, K, 1> head() const
  {
    return this->template segment<K>(0);
  }
  template<int K>
  NMPC_HD Block<Scalar, K, 1> head()
  {
    return this->template segment<K>(0);
  }
  template<int K>
  NMPC_HD Matrix<Scalar, K, 1> tail() const
  {
    return this->template segment<K>(rows() - K);
  }
  template<int K>
  NMPC_HD Block<Scalar, K, 1> tail()
  {
    return this->template segment<K>(rows() - K);
  }
  // matrix pieces
  NMPC_HD Matrix<Scalar, RMAX, 1, DynRows, false> col(int c) const
  {
    Matrix<Scalar, RMAX, 1, DynRows, false> r(rows());
    NMPC_UNROLL
    for(int i = 0; i < RMAX; i++)
    {
      r[i] = d_[i + c * RMAX];
    }
    return r;
  }
  NMPC_HD Block<Scalar, RMAX, 1> col(int c)
  {
    static_assert(!DynRows, "column views of matrices with a fixed number of rows");
    return Block<Scalar, RMAX, 1>(d_ + c * RMAX, RMAX);
  }
  template<int BR, int BC>
  NMPC_HD Matrix<Scalar, BR, BC> block(int r0, int c0) const
  {
    Matrix<Scalar, BR, BC> r;
    NMPC_UNROLL
    for(int c = 0; c < BC; c++)
    {
      NMPC_UNROLL
      for(int i = 0; i < BR; i++)
      {
        r(i, c) = (*this)(r0 + i, c0 + c);
      }
    }
    return r;
  }
  template<int BR, int BC>
  NMPC_HD Block<Scalar, BR, BC> block(int r0, int c0)
  {
    return Block<Scalar, BR, BC>(d_ + r0 + c0 * RMAX, RMAX);
  }
  template<int BR>
  NMPC_HD Block<Scalar, BR, CMAX> middleRows(int r0)
  {
    return Block<Scalar, BR, CMAX>(d_ + r0, RMAX, cols());
  }
  NMPC_HD typename Block<Scalar, RMAX, CMAX>::Diagonal diagonal()
  {
    return Block<Scalar, RMAX, CMAX>(d_, RMAX, cols()).diagonal();
  }
  NMPC_HD Matrix<Scalar, CMAX, RMAX, DynCols, DynRows> transpose() const
  {
    Matrix<Scalar, CMAX, RMAX, DynCols, DynRows> r(cols(), rows());
    NMPC_UNROLL
    for(int c = 0; c < CMAX; c++)
    {
      NMPC_UNROLL
      for(int i = 0; i < RMAX; i++)
      {
        r(c, i) = (*this)(i, c);
      }
    }
    return r;
  }
  NMPC_HD Matrix cross(const Matrix & o) const
  {
    static_assert(RMAX == 3 && CMAX == 1, "cross product of 3-vectors");
    Matrix r;
    r[0] = d_[1] * o.d_[2] - d_[2] * o.d_[1];
    r[1] = d_[2] * o.d_[0] - d_[0] * o.d_[2];
    r[2] = d_[0] * o.d_[1] - d_[1] * o.d_[0];
    return r;
  }
  NMPC_HD void normalize()
  {
    const Scalar n = norm();
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      d_[i] = d_[i] / n;
    }
  }
  NMPC_HD Matrix cwiseMin(const Matrix & o) const
  {
    Matrix r(*this);
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] = o.d_[i] < r.d_[i] ? o.d_[i] : r.d_[i];
    }
    return r;
  }
  NMPC_HD Matrix cwiseMax(const Matrix & o) const
  {
    Matrix r(*this);
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] = o.d_[i] > r.d_[i] ? o.d_[i] : r.d_[i];
    }
    return r;
  }
  NMPC_HD Matrix operator-() const
  {
    Matrix r(*this);
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] = -r.d_[i];
    }
    return r;
  }
  NMPC_HD Matrix operator*(Scalar s) const
  {
    Matrix r(*this);
    r *= s;
    return r;
  }
  NMPC_HD Matrix operator/(Scalar s) const
  {
    Matrix r(*this);
    NMPC_UNROLL
    for(int i = 0; i < kCapacity; i++)
    {
      r.d_[i] = r.d_[i] / s;
    }
    return r;
  }
  /** Matrix product, contraction in ascending index order over this->cols() (run-time when the inner dimension is). */
  template<int C2, bool DR2, bool DC2>
  NMPC_HD Matrix<Scalar, RMAX, C2, DynRows, DC2> operator*(const Matrix<Scalar, CMAX, C2, DR2, DC2> & o) const
  {
    Matrix<Scalar, RMAX, C2, DynRows, DC2> r(rows(), o.cols());
    NMPC_UNROLL
    for(int c = 0; c < C2; c++)
    {
      NMPC_UNROLL
      for(int i = 0; i < RMAX; i++)
      {
        Scalar acc = 0;
        for(int k = 0; k < cols(); k++)
        {
          acc += (*this)(i, k) * o(k, c);
        }
        r(i, c) = acc;
      }
    }
    return r;
  }

private:
  Scalar d_[kCapacity];
};

template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::eval() const
{
  Matrix<Scalar, R, CMAX, false, false> m;
  for(int c = 0; c < CMAX; c++)
  {
    NMPC_UNROLL
    for(int r = 0; r < R; r++)
    {
      m(r, c) = (c < cols_) ? (*this)(r, c) : Scalar(0);
    }
  }
  return m;
}
template<class Scalar, int R, int CMAX>
NMPC_HD Block<Scalar, R, CMAX>::operator Matrix<Scalar, R, CMAX, false, false>() const
{
  return eval();
}
template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::operator-(const Matrix<Scalar, R, CMAX, false, false> & o) const
{
  return eval() - o;
}
template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::operator+(const Matrix<Scalar, R, CMAX, false, false> & o) const
{
  return eval() + o;
}
template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::operator*(Scalar v) const
{
  return eval() * v;
}
template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::operator/(Scalar v) const
{
  return eval() / v;
}
template<class Scalar, int R, int CMAX>
NMPC_HD Matrix<Scalar, R, CMAX, false, false> Block<Scalar, R, CMAX>::cross(const Matrix<Scalar, R, CMAX, false, false> & o) const
{
  return eval().cross(o);
}
template<class Scalar, int R, int CMAX>
NMPC_HD Scalar Block<Scalar, R, CMAX>::dot(const Matrix<Scalar, R, CMAX, false, false> & o) const
{
  return eval().dot(o);
}

template<class Scalar, int N>
using Vector = Matrix<Scalar, N, 1>;
} // namespace nmpc_amd
