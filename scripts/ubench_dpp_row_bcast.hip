#include <hip/hip_runtime.h>
__global__ void k(double * out, const double * in)
{
  double v = in[threadIdx.x];
  int lo = __double2loint(v), hi = __double2hiint(v);
  int blo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + 5, 0xf, 0xf, false);
  int bhi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + 5, 0xf, 0xf, false);
  out[threadIdx.x] = __hiloint2double(bhi, blo);
}
int main()
{
  double *in, *out;
  hipMalloc(&in, 64 * 8); hipMalloc(&out, 64 * 8);
  double h[64]; for(int i = 0; i < 64; i++) h[i] = i;
  hipMemcpy(in, h, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(out, in);
  hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
  for(int i = 0; i < 64; i++) printf("%g ", h[i]);
  printf("\n");
}
