// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS kernel family's access pattern
// (8 bytes per lane, 512-byte segments per wave instruction), as MI355X_MICROARCH.md §HBM asks before an absolute
// HBM byte count is trusted.  Known traffic: read_k reads BYTES, write_k writes BYTES.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while(0)
__global__ void read_k(const double * __restrict__ in, double * out, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for(; i < n; i += stride) s += in[i];
  if(s == 1.2345e300) out[0] = s;
}
__global__ void write_k(double * out, size_t n, double v)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for(; i < n; i += stride) out[i] = v;
}
int main()
{
  const size_t bytes = (size_t)1 << 30, n = bytes / 8;
  double *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int rep = 0; rep < 3; rep++)
  {
    float ms;
    CK(hipEventRecord(e0)); read_k<<<2048, 256>>>(a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("read  1 GiB (8 B/lane): %.3f ms = %.1f GB/s\n", ms, bytes / ms * 1e-6);
    CK(hipEventRecord(e0)); write_k<<<2048, 256>>>(b, n, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("write 1 GiB (8 B/lane): %.3f ms = %.1f GB/s\n", ms, bytes / ms * 1e-6);
  }
  return 0;
}
