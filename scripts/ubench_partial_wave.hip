// Microbenchmark: does a wave64 fp64 FMA stream get cheaper when only 16 / 32 lanes are active?
// (decides whether spreading instances over more, partially filled waves can pay off)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void fma_chain(double * out, int active, int iters)
{
  const int lane = threadIdx.x & 63;
  if(lane >= active) return;
  double a0 = lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  for(int i = 0; i < iters; i++)
  {
    a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
    a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void fma_dep(double * out, int active, int iters)
{
  const int lane = threadIdx.x & 63;
  if(lane >= active) return;
  double a0 = lane * 1e-3;
  const double m = 1.0000001, c = 1e-9;
  for(int i = 0; i < iters; i++) { a0 = fma(a0, m, c); }
  out[blockIdx.x * 64 + lane] = a0;
}
int main()
{
  double * d; hipMalloc(&d, 1024 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 100000;
  for(int blocks : {1, 64, 256, 1024})
    for(int active : {64, 32, 16, 1})
    {
      float ms, ms2;
      fma_chain<<<blocks, 64>>>(d, active, 1000);
      hipDeviceSynchronize();
      hipEventRecord(e0); fma_chain<<<blocks, 64>>>(d, active, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      hipEventRecord(e0); fma_dep<<<blocks, 64>>>(d, active, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms2, e0, e1);
      printf("blocks %4d active %2d : 8-way ILP %.3f ns/fma-instr (%.2f cyc@2.4GHz) | dependent %.3f ns/fma (%.2f cyc)\n", blocks, active,
             ms * 1e6 / (8.0 * iters), ms * 1e6 / (8.0 * iters) * 2.4, ms2 * 1e6 / iters, ms2 * 1e6 / iters * 2.4);
    }
  return 0;
}
