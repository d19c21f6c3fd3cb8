// Issue interval and dependent latency of v_mfma_f64_4x4x4_4b_f64, a quad_perm DPP move pair and ds_bpermute_b32 for a
// lone wavefront on gfx950 (what the quad kernel's Riccati step is made of), timed with HIP events over long loops.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma4_rate.hip -o scripts/ubench_mfma4_rate && scripts/ubench_mfma4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 20000;
__global__ void k_mfma_indep(double * out)
{
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for(int i = 0; i < kIters; i++)
  {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void k_mfma_dep(double * out)
{
  double a = 1e-3, d = 1.0 + threadIdx.x * 1e-4;
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.5, 0, 0, 0);
    }
  }
  out[threadIdx.x] = d;
}
__global__ void k_mfma_valu_dep(double * out)
{
  // MFMA result consumed by a VALU FMA which feeds the next MFMA: the matrix core <-> VALU round trip
  double a = 1e-3, d = 1.0 + threadIdx.x * 1e-4;
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.5, 0, 0, 0);
      d = fma(d, 0.999, 0.001);
    }
  }
  out[threadIdx.x] = d;
}
__global__ void k_dpp_dep(double * out)
{
  double d = 1.0 + threadIdx.x * 1e-4;
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(d), 0x55, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(d), 0x55, 0xf, 0xf, true);
      d = fma(__hiloint2double(hi, lo), 0.999, 0.001);
    }
  }
  out[threadIdx.x] = d;
}
__global__ void k_bperm_dep(double * out)
{
  double d = 1.0 + threadIdx.x * 1e-4;
  const int src = (((threadIdx.x % 4) * 16 + (threadIdx.x / 4 % 4) * 4 + threadIdx.x / 16) % 64) * 4; // 4 x 4 transpose
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(d));
      const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(d));
      d = fma(__hiloint2double(hi, lo), 0.999, 0.001);
    }
  }
  out[threadIdx.x] = d;
}
__global__ void k_fma_dep(double * out)
{
  double d = 1.0 + threadIdx.x * 1e-4;
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      d = fma(d, 0.999, 0.001);
    }
  }
  out[threadIdx.x] = d;
}
/** Does a lone wave issue independent VALU work while its MFMA occupies the matrix core?  Per unrolled element: one MFMA of a
    dependent chain (24 cycles alone) + NV FMAs of chains that do not touch it. */
template<int NV>
__global__ void k_mfma_dep_plus_valu(double * out)
{
  double a = 1e-3, d = 1.0 + threadIdx.x * 1e-4;
  double f[6] = {1.0, 1.1, 1.2, 1.3, 1.4, 1.5};
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.5, 0, 0, 0);
#pragma unroll
      for(int v = 0; v < NV; v++)
      {
        f[v % 6] = fma(f[v % 6], 0.999, 0.001);
      }
    }
  }
  out[threadIdx.x] = d + f[0] + f[1] + f[2] + f[3] + f[4] + f[5];
}
/** The same with independent MFMAs (issue-limited, 13.5 cycles alone). */
template<int NV>
__global__ void k_mfma_indep_plus_valu(double * out)
{
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double f[6] = {1.0, 1.1, 1.2, 1.3, 1.4, 1.5};
  for(int i = 0; i < kIters; i++)
  {
#pragma unroll
    for(int r = 0; r < 8; r++)
    {
      c[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[r], 0, 0, 0);
#pragma unroll
      for(int v = 0; v < NV; v++)
      {
        f[v % 6] = fma(f[v % 6], 0.999, 0.001);
      }
    }
  }
  out[threadIdx.x] = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7] + f[0] + f[1] + f[2] + f[3] + f[4] + f[5];
}
template<class K>
double timeIt(K kernel, double * out)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  kernel<<<1, 64>>>(out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kernel<<<1, 64>>>(out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (static_cast<double>(kIters) * 8); // ns per unrolled element
}
int main()
{
  double * out;
  (void)hipMalloc(&out, 512);
  const double fma_ns = timeIt(k_fma_dep, out);
  const double ns_per_cycle = fma_ns / 4.5; // a dependent fp64 FMA of a lone wave issues every ~4.5 cycles (ubench_latency)
  std::printf("reference: dependent v_fma_f64 %.2f ns each (taken as 4.5 shader cycles)\n", fma_ns);
  std::printf("v_mfma_f64_4x4x4, 8 independent accumulators: %.2f ns = %.1f cycles each\n", timeIt(k_mfma_indep, out),
              timeIt(k_mfma_indep, out) / ns_per_cycle);
  std::printf("v_mfma_f64_4x4x4, dependent through B: %.2f ns = %.1f cycles each\n", timeIt(k_mfma_dep, out),
              timeIt(k_mfma_dep, out) / ns_per_cycle);
  const double mv = timeIt(k_mfma_valu_dep, out);
  std::printf("MFMA -> v_fma_f64 -> MFMA round trip: %.2f ns = %.1f cycles per pair\n", mv, mv / ns_per_cycle);
  const double dp = timeIt(k_dpp_dep, out);
  std::printf("2 x v_mov_b32_dpp quad_perm + dependent v_fma_f64: %.2f ns = %.1f cycles per group\n", dp, dp / ns_per_cycle);
  const double bp = timeIt(k_bperm_dep, out);
  std::printf("2 x ds_bpermute_b32 + dependent v_fma_f64: %.2f ns = %.1f cycles per group\n", bp, bp / ns_per_cycle);
  std::printf("MFMA of a dependent chain + NV independent v_fma_f64 per MFMA (cycles per element):");
  std::printf(" NV=0 %.1f", timeIt(k_mfma_dep_plus_valu<0>, out) / ns_per_cycle);
  std::printf(" NV=2 %.1f", timeIt(k_mfma_dep_plus_valu<2>, out) / ns_per_cycle);
  std::printf(" NV=4 %.1f", timeIt(k_mfma_dep_plus_valu<4>, out) / ns_per_cycle);
  std::printf(" NV=6 %.1f", timeIt(k_mfma_dep_plus_valu<6>, out) / ns_per_cycle);
  std::printf(" NV=8 %.1f\n", timeIt(k_mfma_dep_plus_valu<8>, out) / ns_per_cycle);
  std::printf("independent MFMAs + NV independent v_fma_f64 per MFMA (cycles per element):");
  std::printf(" NV=0 %.1f", timeIt(k_mfma_indep_plus_valu<0>, out) / ns_per_cycle);
  std::printf(" NV=1 %.1f", timeIt(k_mfma_indep_plus_valu<1>, out) / ns_per_cycle);
  std::printf(" NV=2 %.1f", timeIt(k_mfma_indep_plus_valu<2>, out) / ns_per_cycle);
  std::printf(" NV=3 %.1f", timeIt(k_mfma_indep_plus_valu<3>, out) / ns_per_cycle);
  std::printf(" NV=4 %.1f\n", timeIt(k_mfma_indep_plus_valu<4>, out) / ns_per_cycle);
  return 0;
}
