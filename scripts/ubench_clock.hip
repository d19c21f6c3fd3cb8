// Microbenchmark: shader clock (s_memtime ticks vs wall), fp64 FMA issue/latency in cycles, fp64 division,
// sin/cos cost — the per-instruction prices the DDP kernel design is budgeted with.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while(0)
template<int MODE>
__global__ void k(double * out, long long * cyc, int iters)
{
  const int lane = threadIdx.x;
  double a[8];
  for(int j = 0; j < 8; j++) a[j] = 1.0 + lane * 1e-3 + j;
  const double m = 1.0000001, c = 1e-9;
  long long t0 = __builtin_readcyclecounter();
  for(int i = 0; i < iters; i++)
  {
    if(MODE == 0) { for(int j = 0; j < 8; j++) a[j] = fma(a[j], m, c); }            // 8 independent fma
    if(MODE == 1) { a[0] = fma(a[0], m, c); }                                         // dependent fma
    if(MODE == 2) { for(int j = 0; j < 8; j++) a[j] = m / a[j] + 1.0; }               // 8 independent div
    if(MODE == 3) { for(int j = 0; j < 4; j++) a[j] = sin(a[j]) + 2.0; }              // 4 sin
    if(MODE == 4) { for(int j = 0; j < 4; j++) { double s, co; sincos(a[j], &s, &co); a[j] = s + co + 2.0; } } // 4 sincos
    if(MODE == 5) { for(int j = 0; j < 8; j++) a[j] = sqrt(a[j]) + 1.0; }             // 8 sqrt
    if(MODE == 6) { for(int j = 0; j < 8; j++) a[j] = a[j] * m + c * a[(j + 1) & 7]; } // mul+fma mix
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0; for(int j = 0; j < 8; j++) s += a[j];
  out[blockIdx.x * 64 + lane] = s;
  if(lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template<int MODE> int run(const char * name, int per_iter, double * d, long long * dc, int blocks, int iters)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 64>>>(d, dc, iters / 10); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<MODE><<<blocks, 64>>>(d, dc, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long cyc; CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
  printf("%-22s blocks %4d: %8.3f ms, %12lld ticks -> %.1f MHz tick rate, %.2f ns/op, %.2f ticks/op\n", name, blocks, ms, cyc,
         cyc / (ms * 1e3), ms * 1e6 / ((double)iters * per_iter), (double)cyc / ((double)iters * per_iter));
  return 0;
}
int main()
{
  double * d; long long * dc; CK(hipMalloc(&d, 4096 * 64 * 8)); CK(hipMalloc(&dc, 8));
  for(int blocks : {1, 64, 1024})
  {
    run<0>("fma x8 indep", 8, d, dc, blocks, 2000000);
    run<1>("fma dependent", 1, d, dc, blocks, 2000000);
    run<6>("mul+fma mix x8", 8, d, dc, blocks, 2000000);
    run<2>("div x8 indep", 8, d, dc, blocks, 200000);
    run<5>("sqrt x8 indep", 8, d, dc, blocks, 200000);
    run<3>("sin x4", 4, d, dc, blocks, 100000);
    run<4>("sincos x4", 4, d, dc, blocks, 100000);
  }
  return 0;
}
