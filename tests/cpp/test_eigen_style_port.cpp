// A problem class in Eigen block / initialiser syntax (JetGyrostatEigenStyle.hpp) against the same arithmetic written entry by
// entry on scalars (JetGyrostatPlain.hpp): every method of the DDPProblem interface, bit for bit, across the jet schedule
// (eight, zero and four inputs).
//   g++ -std=c++17 -O2 -ffp-contract=off -Iinclude tests/cpp/test_eigen_style_port.cpp && ./a.out
//   hipcc --offload-arch=gfx950 -DDEVICE_COMPILE_CHECK -c ...   (the same functors in a __global__ kernel)
#include <cmath>
#include <cstdio>
#include <cstring>

#include "JetGyrostatEigenStyle.hpp"
#include "JetGyrostatPlain.hpp"

#if defined(DEVICE_COMPILE_CHECK)
__global__ void eval_both(const double * xin, const double * uin, double * out)
{
  conformance::JetGyrostatEigenStyle a;
  conformance::JetGyrostatPlain b;
  using P = conformance::JetGyrostatEigenStyle;
  P::StateDimVector x;
  P::InputDimVector u(8);
  for(int i = 0; i < 9; i++) x[i] = xin[i];
  for(int i = 0; i < 8; i++) u[i] = uin[i];
  const double t = 0.03 * threadIdx.x;
  u.resize(a.inputDim(t));
  P::StateStateDimMatrix fx, fx2;
  P::StateInputDimMatrix fu, fu2;
  a.calcStateEqDeriv(t, x, u, fx, fu);
  b.calcStateEqDeriv(t, x, u, fx2, fu2);
  double d = fabs(a.runningCost(t, x, u) - b.runningCost(t, x, u)) + fabs(a.terminalCost(t, x) - b.terminalCost(t, x));
  const auto xa = a.stateEq(t, x, u), xb = b.stateEq(t, x, u);
  for(int i = 0; i < 9; i++)
  {
    d += fabs(xa[i] - xb[i]);
    for(int j = 0; j < 9; j++)
    {
      d += fabs(fx(i, j) - fx2(i, j));
    }
    for(int j = 0; j < u.size(); j++)
    {
      d += fabs(fu(i, j) - fu2(i, j));
    }
  }
  out[threadIdx.x] = d;
}
int main()
{
  double hx[9] = {0.3, -0.2, 0.15, 0.5, -0.7, 0.3, 0.05, 0.25, -0.15}, hu[8], hout[100];
  for(int i = 0; i < 8; i++) hu[i] = 0.4 * i - 1.3;
  double *dx, *du, *dout;
  hipMalloc(&dx, sizeof(hx));
  hipMalloc(&du, sizeof(hu));
  hipMalloc(&dout, sizeof(hout));
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipMemcpy(du, hu, sizeof(hu), hipMemcpyHostToDevice);
  eval_both<<<1, 100>>>(dx, du, dout); // t = 0 .. 2.97 s: the ring of eight, the coast window, the deck of four
  if(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double worst = 0;
  for(int i = 0; i < 100; i++) worst = hout[i] > worst ? hout[i] : worst;
  std::printf("device: largest difference between the Eigen-style class and the scalar one over 100 times: %g\n", worst);
  if(worst == 0) std::printf("EIGEN_STYLE_PORT_DEVICE_OK\n");
  return worst != 0;
}
#else
template<class M1, class M2>
static int diff(const char * what, const M1 & a, const M2 & b, int rows, int cols)
{
  int bad = 0;
  for(int c = 0; c < cols; c++)
    for(int r = 0; r < rows; r++)
    {
      const double va = a(r, c), vb = b(r, c);
      if(std::memcmp(&va, &vb, sizeof(double)) != 0 && !(va == 0 && vb == 0))
      {
        if(bad < 3) std::printf("  %s(%d,%d): %.17g vs %.17g\n", what, r, c, va, vb);
        bad++;
      }
    }
  return bad;
}

int main()
{
  conformance::JetGyrostatEigenStyle a;
  conformance::JetGyrostatPlain b;
  using P = conformance::JetGyrostatEigenStyle;
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return ((s >> 11) * (1.0 / 9007199254740992.0)) * 2 - 1; };
  int bad = 0, evals = 0;
  for(double t = 0.0; t < 3.0; t += 0.07)
  {
    P::StateDimVector x;
    for(int i = 0; i < 9; i++) x[i] = rnd() * (i < 3 ? 0.4 : 1.5);
    const int m = a.inputDim(t);
    bad += (m != b.inputDim(t));
    P::InputDimVector u(m);
    for(int i = 0; i < 8; i++) u[i] = (i < m) ? 2.0 * rnd() : 0.0;
    const auto xa = a.stateEq(t, x, u), xb = b.stateEq(t, x, u);
    bad += diff("stateEq", xa, xb, 9, 1);
    const double ca = a.runningCost(t, x, u), cb = b.runningCost(t, x, u), ta = a.terminalCost(t, x), tb = b.terminalCost(t, x);
    bad += (ca != cb) + (ta != tb);
    P::StateStateDimMatrix fxa, fxb, lxxa, lxxb, vxxa, vxxb;
    P::StateInputDimMatrix fua(9, m), fub(9, m), lxua(9, m), lxub(9, m);
    P::StateDimVector lxa, lxb, vxa, vxb;
    P::InputDimVector lua(m), lub(m);
    P::InputInputDimMatrix luua(m, m), luub(m, m);
    a.calcStateEqDeriv(t, x, u, fxa, fua);
    b.calcStateEqDeriv(t, x, u, fxb, fub);
    bad += diff("Fx", fxa, fxb, 9, 9) + diff("Fu", fua, fub, 9, m);
    a.calcRunningCostDeriv(t, x, u, lxa, lua, lxxa, luua, lxua);
    b.calcRunningCostDeriv(t, x, u, lxb, lub, lxxb, luub, lxub);
    bad += diff("Lx", lxa, lxb, 9, 1) + diff("Lu", lua, lub, m, 1) + diff("Lxx", lxxa, lxxb, 9, 9) + diff("Luu", luua, luub, m, m)
           + diff("Lxu", lxua, lxub, 9, m);
    a.calcTerminalCostDeriv(t, x, vxa, vxxa);
    b.calcTerminalCostDeriv(t, x, vxb, vxxb);
    bad += diff("Vx", vxa, vxb, 9, 1) + diff("Vxx", vxxa, vxxb, 9, 9);
    evals++;
  }
  std::printf("%d evaluation points, %d differing entries\n", evals, bad);
  if(bad == 0) std::printf("EIGEN_STYLE_PORT_OK\n");
  return bad != 0;
}
#endif
