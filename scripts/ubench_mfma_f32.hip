// Operand layout, summation order and issue / dependent-issue intervals of v_mfma_f32_16x16x4_f32 on gfx950, plus the
// latencies the fp32 tile kernel (ddp_kernels_tile32.hpp) is designed around (v_rcp_f32, ds_bpermute_b32, LDS b128 read).
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma_f32.hip -o scripts/ubench_mfma_f32 && scripts/ubench_mfma_f32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float * A, const float * B, const float * C, float * Dout)
{
  const int l = threadIdx.x;
  v4f c = {C[4 * l + 0], C[4 * l + 1], C[4 * l + 2], C[4 * l + 3]};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[l], B[l], c, 0, 0, 0);
  for(int r = 0; r < 4; r++)
  {
    Dout[4 * l + r] = d[r];
  }
}

// "natural layout" identity: with X, Y held as D-layout register quadruples, mmaNat(X, Y) = sum_s mfma(X[s], Y[s]) = X^T Y
__global__ void nat_kernel(const float * X, const float * Y, float * Z)
{
  const int l = threadIdx.x;
  const v4f x = {X[4 * l + 0], X[4 * l + 1], X[4 * l + 2], X[4 * l + 3]};
  const v4f y = {Y[4 * l + 0], Y[4 * l + 1], Y[4 * l + 2], Y[4 * l + 3]};
  v4f acc = {0, 0, 0, 0};
  for(int s = 0; s < 4; s++)
  {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], y[s], acc, 0, 0, 0);
  }
  for(int r = 0; r < 4; r++)
  {
    Z[4 * l + r] = acc[r];
  }
}

template<int MODE>
__global__ void rate_kernel(float * out, long long * cyc, int n)
{
  const int l = threadIdx.x;
  float a = 1.0f + l * 1e-3f, b = 0.5f + l * 1e-3f;
  v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  __shared__ float lds[4096];
  for(int i = l; i < 4096; i += 64)
  {
    lds[i] = 1.0f + i * 1e-4f;
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for(int i = 0; i < n; i++)
  {
    if(MODE == 0) // independent accumulators: issue interval
    {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
    }
    else if(MODE == 1) // same accumulator (a product's k-slices): dependent through C
    {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    }
    else if(MODE == 2) // result feeds the next product's B operand (natural-layout chaining)
    {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc0[0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc0[1], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc0[2], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc0[3], acc0, 0, 0, 0);
    }
    else if(MODE == 3) // v_rcp_f32 chain
    {
      a = __builtin_amdgcn_rcpf(a);
      a = __builtin_amdgcn_rcpf(a);
      a = __builtin_amdgcn_rcpf(a);
      a = __builtin_amdgcn_rcpf(a);
    }
    else if(MODE == 4) // ds_bpermute_b32 chain
    {
      int v = __float_as_int(a);
      v = __builtin_amdgcn_ds_bpermute(((l + 16) & 63) * 4, v);
      v = __builtin_amdgcn_ds_bpermute(((l + 16) & 63) * 4, v);
      v = __builtin_amdgcn_ds_bpermute(((l + 16) & 63) * 4, v);
      v = __builtin_amdgcn_ds_bpermute(((l + 16) & 63) * 4, v);
      a = __int_as_float(v);
    }
    else if(MODE == 5) // dependent LDS b128 reads (address from the previous value)
    {
      int at = (__float_as_int(a) & 0xff) * 4;
      for(int q = 0; q < 4; q++)
      {
        const v4f v = *reinterpret_cast<const v4f *>(&lds[at]);
        at = (__float_as_int(v[0]) & 0xff) * 4;
      }
      a = __int_as_float(at | 0x3f800000);
    }
    else if(MODE == 6) // fp32 FMA chain
    {
      a = fmaf(a, b, 0.25f);
      a = fmaf(a, b, 0.25f);
      a = fmaf(a, b, 0.25f);
      a = fmaf(a, b, 0.25f);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[l] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + a;
  if(l == 0)
  {
    cyc[0] = t1 - t0;
  }
}

template<int MODE>
double rate(const char * what, float * dout, long long * dc)
{
  const int n = 4096;
  long long hc = 0;
  rate_kernel<MODE><<<1, 64>>>(dout, dc, n);
  rate_kernel<MODE><<<1, 64>>>(dout, dc, n);
  hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  const double per = double(hc) / (4.0 * n);
  printf("%-70s %7.1f shader-clock ticks each\n", what, per);
  return per;
}

int main()
{
  float hA[64], hB[64], hC[256], hD[256];
  float *dA, *dB, *dC, *dD;
  long long * dc;
  hipMalloc(&dA, 256);
  hipMalloc(&dB, 256);
  hipMalloc(&dC, 1024);
  hipMalloc(&dD, 1024);
  hipMalloc(&dc, 16);
  // ---- layout by probing with unit operands: A = e_la, B = e_lb  ->  which D entries become 1
  int a_i[64], a_k[64], b_k[64], b_j[64], d_i[256], d_j[256];
  for(int i = 0; i < 64; i++)
  {
    a_i[i] = a_k[i] = b_k[i] = b_j[i] = -1;
  }
  for(int i = 0; i < 256; i++)
  {
    d_i[i] = d_j[i] = -1;
    hC[i] = 0;
  }
  hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
  // hypothesis: A[i][k] at lane i + 16 k, B[k][j] at lane j + 16 k, D[i][j] at lane j + 16 (i / 4), register i % 4
  int ok = 0, total = 0;
  for(int la = 0; la < 64; la++)
  {
    for(int lb = (la / 16) * 16; lb < (la / 16) * 16 + 16; lb++) // same k group, else the product is zero under the hypothesis
    {
      for(int i = 0; i < 64; i++)
      {
        hA[i] = (i == la);
        hB[i] = (i == lb);
      }
      hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice);
      hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
      layout_kernel<<<1, 64>>>(dA, dB, dC, dD);
      hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
      const int i = la % 16, j = lb % 16;
      const int want = 4 * (j + 16 * (i / 4)) + (i % 4);
      int good = 1;
      for(int e = 0; e < 256; e++)
      {
        good &= (hD[e] == (e == want ? 1.0f : 0.0f));
      }
      ok += good;
      total++;
    }
  }
  printf("v_mfma_f32_16x16x4_f32 layout  A[i][k] lane i+16k | B[k][j] lane j+16k | C/D[i][j] lane j+16(i/4) reg i%%4 : %d / %d unit probes\n", ok,
         total);
  // cross-group probe: different k groups must give zero
  {
    for(int i = 0; i < 64; i++)
    {
      hA[i] = (i == 3);
      hB[i] = (i == 16 + 5);
    }
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    float s = 0;
    for(int e = 0; e < 256; e++)
    {
      s += fabsf(hD[e]);
    }
    printf("operands in different k groups contribute %g (expected 0)\n", s);
  }
  // ---- summation order: random operands against fma chains in k ascending from C, and other candidates
  {
    srand(7);
    for(int i = 0; i < 64; i++)
    {
      hA[i] = 1 + (rand() % 97) * 0.0137f;
      hB[i] = 2 + (rand() % 89) * 0.0211f;
    }
    for(int i = 0; i < 256; i++)
    {
      hC[i] = 0.3f + (rand() % 83) * 0.0171f;
    }
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int eq_asc = 0, eq_desc = 0, eq_pair = 0;
    float err = 0;
    for(int i = 0; i < 16; i++)
      for(int j = 0; j < 16; j++)
      {
        const int e = 4 * (j + 16 * (i / 4)) + (i % 4);
        float s = hC[e];
        for(int k = 0; k < 4; k++)
        {
          s = fmaf(hA[i + 16 * k], hB[j + 16 * k], s);
        }
        float sd = hC[e];
        for(int k = 3; k >= 0; k--)
        {
          sd = fmaf(hA[i + 16 * k], hB[j + 16 * k], sd);
        }
        // exact sum of the four products, rounded once with C (what a fused dot-product unit would do)
        double ex = hC[e];
        for(int k = 0; k < 4; k++)
        {
          ex += double(hA[i + 16 * k]) * double(hB[j + 16 * k]);
        }
        eq_asc += (s == hD[e]);
        eq_desc += (sd == hD[e]);
        eq_pair += (float(ex) == hD[e]);
        err = fmaxf(err, fabsf(s - hD[e]) / fabsf(s));
      }
    printf("summation: equal to ascending-k fma chain %d / 256, descending %d / 256, exact-sum-rounded-once %d / 256 (max rel diff vs "
           "ascending %.3g)\n", eq_asc, eq_desc, eq_pair, err);
  }
  // ---- natural-layout identity  sum_s mfma(X[s], Y[s]) = X^T Y
  {
    float hX[256], hY[256], hZ[256];
    float X[16][16], Y[16][16];
    for(int i = 0; i < 16; i++)
      for(int j = 0; j < 16; j++)
      {
        X[i][j] = (rand() % 19 - 9) * 0.25f;
        Y[i][j] = (rand() % 17 - 8) * 0.5f;
        hX[4 * (j + 16 * (i / 4)) + (i % 4)] = X[i][j];
        hY[4 * (j + 16 * (i / 4)) + (i % 4)] = Y[i][j];
      }
    float *dX, *dY, *dZ;
    hipMalloc(&dX, 1024);
    hipMalloc(&dY, 1024);
    hipMalloc(&dZ, 1024);
    hipMemcpy(dX, hX, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dY, hY, 1024, hipMemcpyHostToDevice);
    nat_kernel<<<1, 64>>>(dX, dY, dZ);
    hipMemcpy(hZ, dZ, 1024, hipMemcpyDeviceToHost);
    int good = 0;
    for(int i = 0; i < 16; i++)
      for(int j = 0; j < 16; j++)
      {
        float s = 0; // small integers / quarters: exact in any order
        for(int k = 0; k < 16; k++)
        {
          s += X[k][i] * Y[k][j];
        }
        good += (s == hZ[4 * (j + 16 * (i / 4)) + (i % 4)]);
      }
    printf("natural layout: sum_s mfma(X.reg[s], Y.reg[s]) == X^T Y on %d / 256 entries\n", good);
  }
  float * dout;
  hipMalloc(&dout, 256);
  rate<0>("v_mfma_f32_16x16x4_f32, four independent accumulators", dout, dc);
  rate<1>("v_mfma_f32_16x16x4_f32, same accumulator (k-slices of one product)", dout, dc);
  rate<2>("v_mfma_f32_16x16x4_f32, result register as next B operand", dout, dc);
  rate<3>("v_rcp_f32 dependent chain", dout, dc);
  rate<4>("ds_bpermute_b32 dependent chain", dout, dc);
  rate<5>("ds_read_b128 dependent chain", dout, dc);
  rate<6>("v_fma_f32 dependent chain", dout, dc);
  return 0;
}
