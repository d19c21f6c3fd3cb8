// v_mfma_f64_16x16x4_f64 on gfx950: issue interval, dependent latencies, the natural-layout identity, and what two
// wavefronts on ONE SIMD share (matrix wave beside matrix wave, matrix wave beside fp64 VALU wave) — the numbers the fp64
// tile kernel (include/nmpc_amd/hip/ddp_kernels_tile64.hpp) is laid out on.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma_f64_16.hip -o scripts/ubench_mfma_f64_16 && scripts/ubench_mfma_f64_16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kIters = 4000;

// role: 0 idle, 1 independent MFMAs, 2 dependent MFMAs (same accumulator), 3 MFMA result as next B operand,
//       4 independent v_fma_f64 (8 chains), 5 dependent v_fma_f64 chain, 6 ds_read_b64 dependent chain,
//       7 MFMA -> 4 VALU fma on the result -> MFMA (dependent round trip)
__device__ __forceinline__ double run_role(int role, double seed, double * lds)
{
  double a = seed * 1e-3 + 1e-3, b = 1.0 + seed * 1e-4;
  if(role == 1)
  {
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for(int i = 0; i < kIters; i++)
    {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
  }
  if(role == 2)
  {
    v4d c0 = {0, 0, 0, 0};
    for(int i = 0; i < kIters; i++)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      }
    }
    return c0[0] + c0[1] + c0[2] + c0[3];
  }
  if(role == 3)
  {
    v4d c0 = {b, b, b, b};
    const v4d z = {0.5, 0.5, 0.5, 0.5};
    for(int i = 0; i < kIters; i++)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[r], z, 0, 0, 0);
      }
    }
    return c0[0] + c0[1] + c0[2] + c0[3];
  }
  if(role == 4)
  {
    double c[8];
    for(int r = 0; r < 8; r++)
    {
      c[r] = b + r;
    }
    for(int i = 0; i < kIters * 4; i++)
    {
#pragma unroll
      for(int r = 0; r < 8; r++)
      {
        c[r] = fma(c[r], 0.999, a);
      }
    }
    double s = 0;
    for(int r = 0; r < 8; r++)
    {
      s += c[r];
    }
    return s;
  }
  if(role == 5)
  {
    double c = b;
    for(int i = 0; i < kIters * 4; i++)
    {
#pragma unroll
      for(int r = 0; r < 8; r++)
      {
        c = fma(c, 0.999, a);
      }
    }
    return c;
  }
  if(role == 6)
  {
    int at = static_cast<int>(threadIdx.x) & 63;
    const int * idx = reinterpret_cast<const int *>(lds);
    for(int i = 0; i < kIters * 4; i++)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        at = idx[2 * at];
      }
    }
    return at;
  }
  if(role == 7)
  {
    v4d c0 = {b, b, b, b};
    const v4d z = {0.5, 0.5, 0.5, 0.5};
    for(int i = 0; i < kIters; i++)
    {
#pragma unroll
      for(int r = 0; r < 4; r++)
      {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c0[0], z, 0, 0, 0);
#pragma unroll
        for(int e = 0; e < 4; e++)
        {
          c0[e] = fma(c0[e], 0.999, 0.001);
        }
      }
    }
    return c0[0] + c0[1] + c0[2] + c0[3];
  }
  return 0;
}

/** Workgroup of 8 waves: wave w runs on SIMD w % 4.  roles[w] says what wave w does; every wave reports its own ticks. */
__global__ void k_roles(const int * roles, double * out, unsigned long long * ticks)
{
  __shared__ double lds[128];
  const int wave = threadIdx.x >> 6;
  if(threadIdx.x < 64)
  {
    reinterpret_cast<int *>(lds)[2 * threadIdx.x] = (threadIdx.x * 17 + 5) & 63;
  }
  __syncthreads();
  const int role = roles[wave];
  const unsigned long long t0 = __builtin_readcyclecounter();
  const double v = run_role(role, static_cast<double>(threadIdx.x & 63), lds);
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v;
  if((threadIdx.x & 63) == 0)
  {
    ticks[wave] = t1 - t0;
  }
}

__global__ void k_layout(const double * X, const double * Y, double * D)
{
  // natural layout: register r of lane (q = lane / 16, j = lane % 16) holds M[4 r + q][j]
  const int l = threadIdx.x, q = l >> 4, j = l & 15;
  v4d x, y, c = {0, 0, 0, 0};
  for(int r = 0; r < 4; r++)
  {
    x[r] = X[(4 * r + q) * 16 + j];
    y[r] = Y[(4 * r + q) * 16 + j];
  }
  for(int s = 0; s < 4; s++)
  {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], y[s], c, 0, 0, 0);
  }
  for(int r = 0; r < 4; r++)
  {
    D[(4 * r + q) * 16 + j] = c[r];
  }
}

static const char * kNames[] = {"idle", "mfma indep", "mfma same-acc", "mfma via B", "fma64 indep x8", "fma64 dep", "ds_read dep", "mfma->4fma->mfma"};
static double perOp(int role, unsigned long long t)
{
  const double n = (role >= 1 && role <= 3) ? kIters * 4.0 : (role == 4 ? kIters * 32.0 : (role == 5 ? kIters * 32.0 : (role == 6 ? kIters * 16.0 : kIters * 4.0)));
  return t / n;
}
int main()
{
  int * d_roles;
  double * d_out;
  unsigned long long * d_ticks;
  hipMalloc(&d_roles, 8 * sizeof(int));
  hipMalloc(&d_out, 512 * sizeof(double));
  hipMalloc(&d_ticks, 8 * sizeof(unsigned long long));
  auto run = [&](const char * label, const int (&roles)[8])
  {
    hipMemcpy(d_roles, roles, sizeof(roles), hipMemcpyHostToDevice);
    unsigned long long t[8];
    for(int rep = 0; rep < 2; rep++)
    {
      k_roles<<<1, 512>>>(d_roles, d_out, d_ticks);
      hipDeviceSynchronize();
    }
    hipMemcpy(t, d_ticks, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-44s", label);
    for(int w = 0; w < 8; w++)
    {
      if(roles[w] != 0)
      {
        printf("  w%d(simd %d) %s: %.1f ticks/op", w, w % 4, kNames[roles[w]], perOp(roles[w], t[w]));
      }
    }
    printf("\n");
  };
  run("lone wave, independent MFMAs", {1, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, same accumulator", {2, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, result as next B", {3, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, mfma -> 4 fma -> mfma", {7, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, 8 independent fma64 chains", {4, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, dependent fma64", {5, 0, 0, 0, 0, 0, 0, 0});
  run("lone wave, dependent ds_read_b32", {6, 0, 0, 0, 0, 0, 0, 0});
  run("two MFMA waves, same SIMD", {1, 0, 0, 0, 1, 0, 0, 0});
  run("two MFMA waves, different SIMDs", {1, 1, 0, 0, 0, 0, 0, 0});
  run("MFMA wave + fma64 wave, same SIMD", {1, 0, 0, 0, 4, 0, 0, 0});
  run("MFMA wave + fma64 wave, different SIMDs", {1, 4, 0, 0, 0, 0, 0, 0});
  run("dependent MFMA wave + fma64 wave, same SIMD", {3, 0, 0, 0, 4, 0, 0, 0});
  run("dependent MFMA + dependent MFMA, same SIMD", {3, 0, 0, 0, 3, 0, 0, 0});
  run("dependent MFMA + dependent fma64, same SIMD", {3, 0, 0, 0, 5, 0, 0, 0});
  run("two fma64 waves, same SIMD", {4, 0, 0, 0, 4, 0, 0, 0});
  run("two dependent fma64 waves, same SIMD", {5, 0, 0, 0, 5, 0, 0, 0});
  run("mfma->fma->mfma + same, same SIMD", {7, 0, 0, 0, 7, 0, 0, 0});
  run("ds_read dep + MFMA, same SIMD", {1, 0, 0, 0, 6, 0, 0, 0});

  // natural-layout identity
  double hX[256], hY[256], hD[256];
  for(int i = 0; i < 256; i++)
  {
    hX[i] = 1 + (rand() % 97) * 0.013;
    hY[i] = 2 + (rand() % 89) * 0.017;
  }
  double *dX, *dY, *dD;
  hipMalloc(&dX, 2048);
  hipMalloc(&dY, 2048);
  hipMalloc(&dD, 2048);
  hipMemcpy(dX, hX, 2048, hipMemcpyHostToDevice);
  hipMemcpy(dY, hY, 2048, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dX, dY, dD);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  int exact = 0;
  for(int i = 0; i < 16; i++)
  {
    for(int j = 0; j < 16; j++)
    {
      double s = 0;
      for(int k = 0; k < 16; k++)
      {
        s = fma(hX[k * 16 + i], hY[k * 16 + j], s);
      }
      exact += (s == hD[i * 16 + j]);
    }
  }
  printf("natural layout: sum_s mfma(X.reg[s], Y.reg[s]) == X^T Y (ascending-row fma chain) on %d / 256 entries\n", exact);
  return 0;
}
