// Operand layout and summation order of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per
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
  // lane map: block = (lane / 4) % 4;  A[i][k] at lane i + 4 block + 16 k;  B[k][j] at j + 4 block + 16 k;
  // C/D[i][j] at j + 4 block + 16 i   (found by probing with unit operands)
  {
    int exact = 0;
    double err = 0;
    for(int blk = 0; blk < 4; blk++)
      for(int i = 0; i < 4; i++)
        for(int j = 0; j < 4; j++)
        {
          const int ld = j + 4 * blk + 16 * i;
          double s = hC[ld];
          for(int kk = 0; kk < 4; kk++)
          {
            s = fma(hA[i + 4 * blk + 16 * kk], hB[j + 4 * blk + 16 * kk], s);
          }
          exact += (s == hD[ld]);
          err = fmax(err, fabs(s - hD[ld]));
        }
    printf("A[i][k] lane i+4b+16k | B[k][j] lane j+4b+16k | C/D[i][j] lane j+4b+16i : max err %g, bitwise equal to the "
           "ascending-k fma chain from C: %d / 64\n", err, exact);
  }
  int ok = 0;
  for(int l = 0; l < 64; l++)
  {
    ok += (hbc[l] == hA[(l / 16) * 16 + 1]);
  }
  printf("DPP row_newbcast:1 on both halves of a double: %d / 64 lanes hold lane 1 of their row\n", ok);
  return 0;
}
