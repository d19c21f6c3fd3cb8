// Accuracy of nmpc_amd::sincosFast / sincos (include/nmpc_amd/linalg.hpp) against long-double references.
// Built and run by tests/test_host_cpu.py (host compile of the same NMPC_HD code the kernels inline).
#include <cmath>
#include <cstdio>
#include <random>

#include <nmpc_amd/linalg.hpp>

static double ulpErr(double got, long double want)
{
  const double w = static_cast<double>(want);
  double u = std::nextafter(std::fabs(w), INFINITY) - std::fabs(w);
  if(u == 0)
  {
    u = 5e-324;
  }
  return static_cast<double>(std::fabs(static_cast<long double>(got) - want) / u);
}

int main()
{
  std::mt19937_64 gen(1);
  int fail = 0;
  const double ranges[4] = {4.0, 50.0, 1e3, 1e8};
  const double limits[4] = {1.6, 1.6, 1.6, 2.6};
  for(int ri = 0; ri < 4; ri++)
  {
    std::uniform_real_distribution<double> dist(-ranges[ri], ranges[ri]);
    double max_s = 0, max_c = 0;
    for(int i = 0; i < 500000; i++)
    {
      const double x = dist(gen);
      double s, c, s2, c2;
      nmpc_amd::sincosFast(x, s, c);
      nmpc_amd::sincos(x, s2, c2);
      if(s != s2 || c != c2)
      {
        fail++;
      }
      max_s = std::fmax(max_s, ulpErr(s, sinl(static_cast<long double>(x))));
      max_c = std::fmax(max_c, ulpErr(c, cosl(static_cast<long double>(x))));
    }
    std::printf("range +-%g max ulp sin %.3f cos %.3f\n", ranges[ri], max_s, max_c);
    if(max_s > limits[ri] || max_c > limits[ri])
    {
      fail++;
    }
  }
  double s, c;
  nmpc_amd::sincosFast(M_PI, s, c); // sin(pi_double) must keep full relative accuracy (cancellation in the reduction)
  if(s != 1.2246467991473532e-16 || c != -1.0)
  {
    fail++;
  }
  nmpc_amd::sincosFast(0.0, s, c);
  if(s != 0.0 || c != 1.0)
  {
    fail++;
  }
  nmpc_amd::sincosFast(134217728.0, s, c); // outside the documented range: loud NaN
  if(!std::isnan(s) || !std::isnan(c))
  {
    fail++;
  }
  nmpc_amd::sincosFast(NAN, s, c);
  if(!std::isnan(s))
  {
    fail++;
  }
  nmpc_amd::sincos(1e9, s, c); // the full-range version falls back to libm
  if(std::fabs(s - std::sin(1e9)) > 1e-15 || std::fabs(c - std::cos(1e9)) > 1e-15)
  {
    fail++;
  }
  std::printf(fail == 0 ? "SINCOS_OK\n" : "SINCOS_FAIL %d\n", fail);
  return fail == 0 ? 0 : 1;
}
