// Accuracy of nmpc_amd::recipFast on gfx950 against the correctly-rounded divide, over arguments spanning the
// magnitudes the solver feeds it (pivots, determinants, norms + 1).  Build + run: see scripts/profile_round.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <nmpc_amd/linalg.hpp>

__global__ void recip_k(const double * x, double * fast, double * exact, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
  {
    fast[i] = nmpc_amd::recipFast(x[i]);
    exact[i] = 1.0 / x[i];
  }
}

int main()
{
  const int n = 1 << 22;
  double * h = new double[n];
  uint64_t s = 1234;
  for(int i = 0; i < n; i++)
  {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = (z >> 11) * (1.0 / 9007199254740992.0); // [0,1)
    const int e = static_cast<int>((z & 0xff) % 81) - 40; // 2^-40 .. 2^40
    h[i] = std::ldexp(1.0 + u, e) * ((z & 0x100) ? -1.0 : 1.0);
  }
  double *dx, *df, *de;
  (void)hipMalloc(&dx, n * sizeof(double));
  (void)hipMalloc(&df, n * sizeof(double));
  (void)hipMalloc(&de, n * sizeof(double));
  (void)hipMemcpy(dx, h, n * sizeof(double), hipMemcpyHostToDevice);
  recip_k<<<n / 256, 256>>>(dx, df, de, n);
  double * f = new double[n];
  double * e = new double[n];
  (void)hipMemcpy(f, df, n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipMemcpy(e, de, n * sizeof(double), hipMemcpyDeviceToHost);
  long long max_ulp = 0, n_diff = 0;
  for(int i = 0; i < n; i++)
  {
    int64_t a, b;
    std::memcpy(&a, &f[i], 8);
    std::memcpy(&b, &e[i], 8);
    const long long d = std::llabs(a - b);
    if(d > max_ulp) max_ulp = d;
    if(d) n_diff++;
  }
  std::printf("recipFast vs 1/x on gfx950: %d arguments in +-[2^-40, 2^41): max error %lld ulp, %lld (%.3f %%) differ\n", n, max_ulp,
              n_diff, 100.0 * n_diff / n);
  return max_ulp <= 2 ? 0 : 1;
}
