// Issue cost (cycles per instruction, one wave64 alone on its SIMD, and two waves sharing a SIMD) of the fp64 DPP forms against the
// plain fp64 / 32-bit instructions they would replace: v_mov_b64_dpp, v_fmac_f64_dpp (row_newbcast, the only DPP control of the DP ALU),
// a pair of v_mov_b32_dpp, v_fma_f64, v_mul_f64, v_mov_b32.   hipcc --offload-arch=gfx950 -O3 scripts/ubench_dpp64.hip -o ubench_dpp64
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template<int KIND>
__global__ void k(double * out, const double * in, long long * ticks)
{
  double a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, x = in[64 + threadIdx.x % 64];
  double b0 = a0, b1 = a1, b2 = a2, b3 = a3;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for(int it = 0; it < 16; it++)
  {
    if(KIND == 0)
    {
      REP64(asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));)
    }
    else if(KIND == 1)
    {
      REP64(asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    }
    else if(KIND == 2)
    {
      REP64(asm volatile("v_fmac_f64_dpp %0, -%4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, -%5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, -%6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, -%7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(x));)
    }
    else if(KIND == 3)
    { // four 64-bit broadcasts as pairs of 32-bit DPP moves
      REP64(asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(reinterpret_cast<int &>(b0)), "+v"(reinterpret_cast<int &>(b1)), "+v"(reinterpret_cast<int &>(b2)), "+v"(reinterpret_cast<int &>(b3)) : "v"(__double2loint(a4)), "v"(__double2loint(a5)), "v"(__double2loint(a6)), "v"(__double2loint(a7)));)
    }
    else if(KIND == 4)
    {
      REP64(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));)
    }
    else if(KIND == 5)
    { // 32-bit moves
      REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "+v"(reinterpret_cast<int &>(b0)), "+v"(reinterpret_cast<int &>(b1)), "+v"(reinterpret_cast<int &>(b2)), "+v"(reinterpret_cast<int &>(b3)) : "v"(__double2loint(a4)), "v"(__double2loint(a5)), "v"(__double2loint(a6)), "v"(__double2loint(a7)));)
    }
    else if(KIND == 6)
    { // the forward pair as the batch has it: mul, s_nop 1, mov_b64_dpp, mul, fma, fma (one (i, J) step), four independent ones
      REP64(asm volatile("v_mul_f64 %0, %4, %8\n v_mul_f64 %1, %5, %8\n s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b64_dpp %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f64 %2, %0, %8\n v_mul_f64 %3, %1, %8\n v_fma_f64 %4, -%2, %8, %4\n v_fma_f64 %5, -%3, %8, %5\n v_fma_f64 %6, -%0, %8, %6\n v_fma_f64 %7, -%1, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    }
    else if(KIND == 7)
    { // the same with 32-bit DPP pairs instead of v_mov_b64_dpp (compiler-scheduled, builtins)
#pragma unroll
      for(int u = 0; u < 64; u++)
      {
        double t0_ = a4 * x, t1_ = a5 * x;
        t0_ = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(t0_), 0x153, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(t0_), 0x153, 0xf, 0xf, true));
        t1_ = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(t1_), 0x153, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(t1_), 0x153, 0xf, 0xf, true));
        const double p0 = t0_ * x, p1 = t1_ * x;
        a4 = __builtin_fma(-p0, x, a4);
        a5 = __builtin_fma(-p1, x, a5);
        a6 = __builtin_fma(-t0_, x, a6);
        a7 = __builtin_fma(-t1_, x, a7);
        asm volatile("" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3;
  if(threadIdx.x % 64 == 0)
  {
    ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  }
}
template<int KIND>
void run(const char * name, int per_rep, int waves)
{
  double *in, *out;
  long long * ticks;
  hipMalloc(&in, 128 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 64 * 8);
  double h[128]; for(int i = 0; i < 128; i++) h[i] = 1.0 + 1e-9 * i;
  hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
  k<KIND><<<1, 64 * waves>>>(out, in, ticks);
  k<KIND><<<1, 64 * waves>>>(out, in, ticks);
  long long t[64];
  hipMemcpy(t, ticks, 64 * 8, hipMemcpyDeviceToHost);
  const double n = 16.0 * 64 * per_rep;
  printf("%-58s %d wave(s) in the workgroup (%d per SIMD): %.2f shader-clock ticks per instruction (wave 0)\n", name, waves, (waves + 3) / 4, t[0] / n);
}
int main()
{
  for(int waves : {1, 8})
  {
    if(waves == 1) { run<0>("v_fma_f64", 4, 1); run<4>("v_mul_f64", 4, 1); run<1>("v_mov_b64_dpp row_newbcast", 4, 1); run<2>("v_fmac_f64_dpp row_newbcast", 4, 1); run<3>("v_mov_b32_dpp row_newbcast", 8, 1); run<5>("v_mov_b32", 4, 1); run<6>("forward step with v_mov_b64_dpp (12 instr incl. 2 s_nop)", 12, 1); run<7>("forward step with 32-bit DPP pairs (13 instr incl. 1 s_nop)", 13, 1); }
    else { run<0>("v_fma_f64", 4, 8); run<4>("v_mul_f64", 4, 8); run<1>("v_mov_b64_dpp row_newbcast", 4, 8); run<2>("v_fmac_f64_dpp row_newbcast", 4, 8); run<3>("v_mov_b32_dpp row_newbcast", 8, 8); run<5>("v_mov_b32", 4, 8); run<6>("forward step with v_mov_b64_dpp (12 instr incl. 2 s_nop)", 12, 8); run<7>("forward step with 32-bit DPP pairs (13 instr incl. 1 s_nop)", 13, 8); }
  }
  return 0;
}
