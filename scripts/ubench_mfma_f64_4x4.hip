// Operand layout, summation order and latency of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per
// wavefront, one per 16-lane row), and of the 64-bit row broadcast (DPP row_newbcast) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma_f64_4x4.hip -o /tmp/ubench_mfma4 && /tmp/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
__global__ void k(const double * A, const double * B, const double * C, double * Dout, double * bc, long long * cyc)
{
  const int l = threadIdx.x;
  const double a = A[l], b = B[l], c = C[l];
  Dout[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  // row broadcast of lane 1 of each 16-lane row
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x151, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x151, 0xf, 0xf, false);
  bc[l] = __hiloint2double(hi, lo);
  // latency of a dependent chain of 64 MFMAs
  double d = c;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for(int r = 0; r < 64; r++)
  {
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, c, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  // dependent chain D -> A operand
  double e = c;
#pragma unroll
  for(int r = 0; r < 64; r++)
  {
    e = __builtin_amdgcn_mfma_f64_4x4x4f64(e, b, c, 0, 0, 0);
  }
  const long long t2 = __builtin_readcyclecounter();
  if(l == 0)
  {
    cyc[0] = t1 - t0;
    cyc[1] = t2 - t1;
  }
  Dout[64 + l] = d + e;
}
int main()
{
  double hA[64], hB[64], hC[64], hD[128], hbc[64];
  long long hc[2];
  for(int i = 0; i < 64; i++)
  {
    hA[i] = 1 + (rand() % 97) * 0.0137;
    hB[i] = 2 + (rand() % 89) * 0.0211;
    hC[i] = 0.3 + (rand() % 83) * 0.0171;
  }
  double *dA, *dB, *dC, *dD, *dbc;
  long long * dc;
  hipMalloc(&dA, 512);
  hipMalloc(&dB, 512);
  hipMalloc(&dC, 512);
  hipMalloc(&dD, 1024);
  hipMalloc(&dbc, 512);
  hipMalloc(&dc, 16);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC, dD, dbc, dc);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(hbc, dbc, 512, hipMemcpyDeviceToHost);
  hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  // hypotheses for (A, B, D) lane maps inside a 16-lane row: index = p + 4 * q; "rm": (row, col) = (q, p); "cm": (p, q)
  const char * names[2] = {"row-major (lane = 4*row + col)", "col-major (lane = row + 4*col)"};
  for(int ha = 0; ha < 2; ha++)
    for(int hb = 0; hb < 2; hb++)
      for(int hd = 0; hd < 2; hd++)
      {
        int exact = 0;
        double err = 0;
        for(int blk = 0; blk < 4; blk++)
          for(int i = 0; i < 4; i++)
            for(int j = 0; j < 4; j++)
            {
              const int ld = 16 * blk + (hd == 0 ? 4 * i + j : i + 4 * j);
              double s = hC[ld];
              for(int kk = 0; kk < 4; kk++)
              {
                const double av = hA[16 * blk + (ha == 0 ? 4 * i + kk : i + 4 * kk)];
                const double bv = hB[16 * blk + (hb == 0 ? 4 * kk + j : kk + 4 * j)];
                s = fma(av, bv, s);
              }
              exact += (s == hD[ld]);
              err = fmax(err, fabs(s - hD[ld]));
            }
        if(err < 1e-9)
        {
          printf("A %s | B %s | D,C %s : max err %g, bitwise equal to the ascending-k fma chain from C: %d / 64\n",
                 names[ha], names[hb], names[hd], err, exact);
        }
      }
  int ok = 0;
  for(int l = 0; l < 64; l++)
  {
    ok += (hbc[l] == hA[(l / 16) * 16 + 1]);
  }
  printf("DPP row_newbcast:1 on both halves of a double: %d / 64 lanes hold lane 1 of their row\n", ok);
  printf("dependent chain of 64 MFMA 4x4x4 f64: D->B %.1f cycles each, D->A %.1f cycles each\n", hc[0] / 64.0, hc[1] / 64.0);
  return 0;
}
