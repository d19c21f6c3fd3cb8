// fp64 FMA latency vs issue interval for a LONE wavefront on gfx950: C interleaved dependent chains, each 64 FMAs
// long per loop trip (unrolled), C = 1, 2, 3, 4, 6, 8.  cycles per FMA = max(issue interval, latency / C).
// Also: the raw accuracy-irrelevant cost of v_rcp_f64 and of a dependent v_cndmask pair inside a chain.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int C>
__global__ void chain_k(double * out, long long * cyc, int trips, double m, double c)
{
  double a[C];
  for(int j = 0; j < C; j++) a[j] = threadIdx.x * 1e-3 + j;
  const long long t0 = __builtin_readcyclecounter();
  for(int t = 0; t < trips; t++)
  {
#pragma unroll
    for(int u = 0; u < 64; u++)
    {
#pragma unroll
      for(int j = 0; j < C; j++) a[j] = fma(a[j], m, c);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for(int j = 0; j < C; j++) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// every operand a distinct per-lane VGPR pair (no SGPR / literal operands): what the solver's matrix code issues
template<int C>
__global__ void chain3_k(double * out, long long * cyc, int trips, const double * in)
{
  double a[C], b[C], c[C];
  for(int j = 0; j < C; j++)
  {
    a[j] = in[threadIdx.x + 64 * j];
    b[j] = in[threadIdx.x + 64 * (j + 8)];
    c[j] = in[threadIdx.x + 64 * (j + 16)];
  }
  const long long t0 = __builtin_readcyclecounter();
  for(int t = 0; t < trips; t++)
  {
#pragma unroll
    for(int u = 0; u < 64; u++)
    {
#pragma unroll
      for(int j = 0; j < C; j++) a[j] = fma(a[j], b[j], c[j]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for(int j = 0; j < C; j++) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void rcp_chain_k(double * out, long long * cyc, int trips)
{
  double a = 1.5 + threadIdx.x * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for(int t = 0; t < trips; t++)
  {
#pragma unroll
    for(int u = 0; u < 64; u++) a = __builtin_amdgcn_rcp(a) + 0.5;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void sel_chain_k(double * out, long long * cyc, int trips, double thr)
{
  double a = 1.5 + threadIdx.x * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for(int t = 0; t < trips; t++)
  {
#pragma unroll
    for(int u = 0; u < 64; u++) a = (a < thr) ? a * 1.0000001 : thr - a;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if(threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int C>
void run(double * d, long long * dc)
{
  const int trips = 2000;
  chain_k<C><<<1, 64>>>(d, dc, trips, 0.999999, 1e-7);
  (void)hipDeviceSynchronize();
  chain_k<C><<<1, 64>>>(d, dc, trips, 0.999999, 1e-7);
  (void)hipDeviceSynchronize();
  long long c;
  (void)hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
  std::printf("fma, %d dependent chain(s) interleaved: %6.2f cycles per FMA, %6.2f per chain step\n", C,
              double(c) / (double(trips) * 64 * C), double(c) / (double(trips) * 64));
}

template<int C>
void run3(double * d, long long * dc, const double * in)
{
  const int trips = 2000;
  chain3_k<C><<<1, 64>>>(d, dc, trips, in);
  (void)hipDeviceSynchronize();
  long long c;
  (void)hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
  std::printf("fma with three VGPR operands, %d chain(s): %6.2f cycles per FMA\n", C, double(c) / (double(trips) * 64 * C));
}

int main()
{
  double * d;
  long long * dc;
  (void)hipMalloc(&d, 64 * sizeof(double));
  (void)hipMalloc(&dc, sizeof(long long));
  double * in;
  (void)hipMalloc(&in, 64 * 24 * sizeof(double));
  {
    double h[64 * 24];
    for(int i = 0; i < 64 * 24; i++) h[i] = (i < 64 * 8) ? 1.0 + 1e-3 * i : ((i < 64 * 16) ? 0.999999 : 1e-7);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  }
  run3<1>(d, dc, in);
  run3<2>(d, dc, in);
  run3<4>(d, dc, in);
  run3<8>(d, dc, in);
  run<1>(d, dc);
  run<2>(d, dc);
  run<3>(d, dc);
  run<4>(d, dc);
  run<6>(d, dc);
  run<8>(d, dc);
  long long c;
  rcp_chain_k<<<1, 64>>>(d, dc, 2000);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
  std::printf("dependent v_rcp_f64 + v_add_f64: %6.2f cycles per pair\n", double(c) / (2000.0 * 64));
  sel_chain_k<<<1, 64>>>(d, dc, 2000, 1e300);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
  std::printf("dependent compare + mul + sub + select: %6.2f cycles per step\n", double(c) / (2000.0 * 64));
  return 0;
}
