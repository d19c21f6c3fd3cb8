// Issue cost (shader cycles per instruction, s_memtime) of the instruction kinds the latency-bound kernels are made of, for a
// LONE wavefront on gfx950: 32 independent copies of one instruction per loop trip, written as inline assembly so that the
// compiler neither merges nor reorders them.  A wave issues in order: what a kernel with one wave per SIMD pays per
// instruction is this number, not the throughput figure of a full machine.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_issue_cost.hip -o scripts/ubench_issue_cost && scripts/ubench_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kTrips = 4000;
constexpr int kPer = 32;

#define REP4(X) X X X X
#define REP32(X) REP4(REP4(X)) REP4(REP4(X))

enum Kind
{
  kFma,
  kMul,
  kAdd,
  kMax,
  kCnd,
  kDpp,
  kMov32,
  kMov64,
  kXor,
  kCmp,
  kRcp,
  kDsRead,
  kDsRead2,
  kDsRead128,
  kDsWrite,
  kDsReadSpread,
  kDsReadBcast,
  kDsWriteSame,
  kDsMix,
  kDsWrite2,
  kDsWrite2st64,
  kDsWrite128,
  kGlobalLoad,
  kGlobalLoadX4,
  kGlobalStore,
  kDsWrite16,
  kDsWriteMirror,
  kGlobalLoad16,
  kGlobalLoadMirror,
  kDsRead16,
  kFmaDep,
  kCndDep,
  kDppDep,
  kMfma,
  kSaluAdd,
  kNop0
};

template<int KIND>
__global__ void issue_k(double * out, long long * cyc, double seed, double * gbuf)
{
  extern __shared__ double lds_all[];
  double * lds = lds_all + (threadIdx.x / 64) * 4096;
  double a = seed + threadIdx.x * 1e-3, b = 0.999, c = 1e-3;
  double r[8] = {a, a + 1, a + 2, a + 3, a + 4, a + 5, a + 6, a + 7};
  int ia = threadIdx.x, ib = 7;
  for(int i = threadIdx.x % 64; i < 4096; i += 64)
  {
    lds[i] = i;
  }
  __syncthreads();
  const unsigned addr_same = 64;
  const unsigned wave_base = (threadIdx.x / 64) * 4096 * 8;
  const unsigned addr_lane = wave_base + (threadIdx.x % 64) * 8;
  const unsigned addr_bcast16 = wave_base + ((threadIdx.x % 64) / 16) * 8 + (((threadIdx.x % 64) / 4) % 4) * 6272; // 16 distinct addresses per wave
  const unsigned addr_same12 = wave_base + (((threadIdx.x % 64) / 16 == 0) ? (threadIdx.x % 64) * 8 : 4096 + (((threadIdx.x % 64) / 4) % 4) * 8 * 49);
  const unsigned addr_spread = wave_base + (threadIdx.x % 16) * 8 + ((threadIdx.x % 64) / 16) * 6272; // the quad kernel's pattern
  const unsigned addr_lane16 = wave_base + (threadIdx.x % 64) * 16;
  double * gptr = gbuf + (threadIdx.x / 64) * 4096 + (threadIdx.x % 64);
  double * gptr16 = gbuf + (threadIdx.x / 64) * 4096 + (threadIdx.x % 64) * 2;
  double l0 = 0, l1 = 0;
  double4 l4;
  l4.x = l4.y = l4.z = l4.w = 0; // (unused unless kDsRead128)
  const long long t0 = __builtin_readcyclecounter();
  for(int t = 0; t < kTrips; t++)
  {
    if constexpr(KIND == kFma)
    {
      REP4(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(b), "v"(c));)
    }
    else if constexpr(KIND == kMul)
    {
      REP4(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                        "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(b));)
    }
    else if constexpr(KIND == kAdd)
    {
      REP4(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                        "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(c));)
    }
    else if constexpr(KIND == kMax)
    {
      REP4(asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                        "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(c));)
    }
    else if constexpr(KIND == kCnd)
    {
      int * q = reinterpret_cast<int *>(r);
      REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                        : "v"(ib)
                        : "vcc");)
    }
    else if constexpr(KIND == kDpp)
    {
      int * q = reinterpret_cast<int *>(r);
      REP4(asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                        : "v"(ia));)
    }
    else if constexpr(KIND == kMov32)
    {
      int * q = reinterpret_cast<int *>(r);
      REP4(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                        : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                        : "v"(ia));)
    }
    else if constexpr(KIND == kMov64)
    {
      REP4(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(b));)
    }
    else if constexpr(KIND == kXor)
    {
      int * q = reinterpret_cast<int *>(r);
      REP4(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                        : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                        : "v"(ib));)
    }
    else if constexpr(KIND == kCmp)
    {
      REP4(asm volatile("v_cmp_nge_f64 vcc, %0, %1\n v_cmp_nge_f64 vcc, %2, %3\n v_cmp_nge_f64 vcc, %4, %5\n v_cmp_nge_f64 vcc, %6, %7\n"
                        "v_cmp_nge_f64 vcc, %1, %0\n v_cmp_nge_f64 vcc, %3, %2\n v_cmp_nge_f64 vcc, %5, %4\n v_cmp_nge_f64 vcc, %7, %6\n"
                        :
                        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
                        : "vcc");)
    }
    else if constexpr(KIND == kRcp)
    {
      REP4(asm volatile("v_rcp_f64 %0, %8\n v_rcp_f64 %1, %8\n v_rcp_f64 %2, %8\n v_rcp_f64 %3, %8\n v_rcp_f64 %4, %8\n v_rcp_f64 %5, %8\n v_rcp_f64 %6, %8\n v_rcp_f64 %7, %8\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(b));)
    }
    else if constexpr(KIND == kDsMix)
    {
      // the quad step's LDS traffic: 8 operand reads and 2 gain writes, four times
      REP4(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                        "ds_write_b64 %9, %10 offset:4096\n"
                        "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                        "ds_write_b64 %9, %10 offset:4608\n"
                        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                        : "v"(addr_spread), "v"(addr_lane), "v"(b)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsWrite2 || KIND == kDsWrite2st64)
    {
      if constexpr(KIND == kDsWrite2)
      {
        REP4(asm volatile("ds_write2_b64 %8, %0, %1 offset0:0 offset1:1\n ds_write2_b64 %8, %2, %3 offset0:64 offset1:65\n ds_write2_b64 %8, %4, %5 offset0:128 offset1:129\n ds_write2_b64 %8, %6, %7 offset0:192 offset1:193\n"
                          "ds_write2_b64 %8, %0, %1 offset0:2 offset1:3\n ds_write2_b64 %8, %2, %3 offset0:66 offset1:67\n ds_write2_b64 %8, %4, %5 offset0:130 offset1:131\n ds_write2_b64 %8, %6, %7 offset0:194 offset1:195\n"
                          :
                          : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(addr_lane16)
                          : "memory");)
      }
      else
      {
        REP4(asm volatile("ds_write2st64_b64 %8, %0, %1 offset0:0 offset1:1\n ds_write2st64_b64 %8, %2, %3 offset0:2 offset1:3\n ds_write2st64_b64 %8, %4, %5 offset0:4 offset1:5\n ds_write2st64_b64 %8, %6, %7 offset0:6 offset1:7\n"
                          "ds_write2st64_b64 %8, %0, %1 offset0:8 offset1:9\n ds_write2st64_b64 %8, %2, %3 offset0:10 offset1:11\n ds_write2st64_b64 %8, %4, %5 offset0:12 offset1:13\n ds_write2st64_b64 %8, %6, %7 offset0:14 offset1:15\n"
                          :
                          : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(addr_lane)
                          : "memory");)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsWrite128)
    {
      typedef double v2d __attribute__((ext_vector_type(2)));
      v2d q[4];
      for(int j = 0; j < 4; j++)
      {
        q[j][0] = r[2 * j];
        q[j][1] = r[2 * j + 1];
      }
      REP4(asm volatile("ds_write_b128 %4, %0\n ds_write_b128 %4, %1 offset:1024\n ds_write_b128 %4, %2 offset:2048\n ds_write_b128 %4, %3 offset:3072\n"
                        "ds_write_b128 %4, %0 offset:4096\n ds_write_b128 %4, %1 offset:5120\n ds_write_b128 %4, %2 offset:6144\n ds_write_b128 %4, %3 offset:7168\n"
                        :
                        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(addr_lane16)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kGlobalLoad)
    {
      REP4(asm volatile("global_load_dwordx2 %0, %8, off\n global_load_dwordx2 %1, %8, off offset:512\n global_load_dwordx2 %2, %8, off offset:1024\n global_load_dwordx2 %3, %8, off offset:1536\n"
                        "global_load_dwordx2 %4, %8, off offset:2048\n global_load_dwordx2 %5, %8, off offset:2560\n global_load_dwordx2 %6, %8, off offset:3072\n global_load_dwordx2 %7, %8, off offset:3584\n"
                        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                        : "v"(gptr)
                        : "memory");)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kGlobalLoadX4)
    {
      double2 * q = reinterpret_cast<double2 *>(r);
      REP4(asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:1024\n global_load_dwordx4 %2, %4, off offset:2048\n global_load_dwordx4 %3, %4, off offset:3072\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(gptr16)
                        : "memory");
           asm volatile("global_load_dwordx4 %0, %4, off offset:512\n global_load_dwordx4 %1, %4, off offset:1536\n global_load_dwordx4 %2, %4, off offset:2560\n global_load_dwordx4 %3, %4, off offset:3584\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(gptr16)
                        : "memory");)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kGlobalStore)
    {
      REP4(asm volatile("global_store_dwordx2 %8, %0, off\n global_store_dwordx2 %8, %1, off offset:512\n global_store_dwordx2 %8, %2, off offset:1024\n global_store_dwordx2 %8, %3, off offset:1536\n"
                        "global_store_dwordx2 %8, %4, off offset:2048\n global_store_dwordx2 %8, %5, off offset:2560\n global_store_dwordx2 %8, %6, off offset:3072\n global_store_dwordx2 %8, %7, off offset:3584\n"
                        :
                        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(gptr)
                        : "memory");)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsWrite16 || KIND == kDsWriteMirror)
    {
      // the forward master: 16 instances in lanes 0..15, lanes 16..63 mirror them (same addresses, same values) or are off
      const unsigned ad = wave_base + (threadIdx.x % 16) * 8;
      if(KIND == kDsWriteMirror || threadIdx.x % 64 < 16)
      {
        REP4(asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n"
                          "ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n"
                          :
                          : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(ad)
                          : "memory");)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    else if constexpr(KIND == kDsRead16)
    {
      const unsigned ad = wave_base + (threadIdx.x % 16) * 8;
      if(threadIdx.x % 64 < 16)
      {
        REP4(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                          "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                          : "v"(ad)
                          : "memory");)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    else if constexpr(KIND == kGlobalLoad16 || KIND == kGlobalLoadMirror)
    {
      double * gp = gbuf + (threadIdx.x / 64) * 4096 + (threadIdx.x % 16);
      if(KIND == kGlobalLoadMirror || threadIdx.x % 64 < 16)
      {
        REP4(asm volatile("global_load_dwordx2 %0, %8, off\n global_load_dwordx2 %1, %8, off offset:512\n global_load_dwordx2 %2, %8, off offset:1024\n global_load_dwordx2 %3, %8, off offset:1536\n"
                          "global_load_dwordx2 %4, %8, off offset:2048\n global_load_dwordx2 %5, %8, off offset:2560\n global_load_dwordx2 %6, %8, off offset:3072\n global_load_dwordx2 %7, %8, off offset:3584\n"
                          : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                          : "v"(gp)
                          : "memory");)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    else if constexpr(KIND == kDsWriteSame)
    {
      REP4(asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n"
                        "ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n"
                        :
                        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(addr_same12)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsRead || KIND == kDsReadSpread || KIND == kDsReadBcast)
    {
      const unsigned ad = (KIND == kDsRead) ? addr_lane : (KIND == kDsReadSpread ? addr_spread : addr_bcast16);
      REP4(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                        "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                        : "v"(ad)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsRead2)
    {
      double2 * q = reinterpret_cast<double2 *>(r);
      REP4(asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:16\n ds_read2_b64 %1, %4 offset0:32 offset1:48\n"
                        "ds_read2_b64 %2, %4 offset0:64 offset1:80\n ds_read2_b64 %3, %4 offset0:96 offset1:112\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(addr_lane)
                        : "memory");
           asm volatile("ds_read2_b64 %0, %4 offset0:1 offset1:17\n ds_read2_b64 %1, %4 offset0:33 offset1:49\n"
                        "ds_read2_b64 %2, %4 offset0:65 offset1:81\n ds_read2_b64 %3, %4 offset0:97 offset1:113\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(addr_lane)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsRead128)
    {
      double2 * q = reinterpret_cast<double2 *>(r);
      const unsigned ad = wave_base + (threadIdx.x % 64) * 16;
      REP4(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(ad)
                        : "memory");
           asm volatile("ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                        : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3])
                        : "v"(ad)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kDsWrite)
    {
      REP4(asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n"
                        "ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n"
                        :
                        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(addr_lane)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    else if constexpr(KIND == kFmaDep)
    {
      REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[0]) : "v"(b), "v"(c));)
    }
    else if constexpr(KIND == kCndDep)
    {
      int * q = reinterpret_cast<int *>(r);
      REP32(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[0]) : "v"(ib) : "vcc");)
    }
    else if constexpr(KIND == kDppDep)
    {
      int * q = reinterpret_cast<int *>(r);
      REP32(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(q[0]));)
    }
    else if constexpr(KIND == kMfma)
    {
      REP4(asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n"
                        "v_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n v_mfma_f64_4x4x4_4b_f64 %4, %8, %9, %4\n v_mfma_f64_4x4x4_4b_f64 %5, %8, %9, %5\n"
                        "v_mfma_f64_4x4x4_4b_f64 %6, %8, %9, %6\n v_mfma_f64_4x4x4_4b_f64 %7, %8, %9, %7\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                        : "v"(b), "v"(c));)
    }
    else if constexpr(KIND == kSaluAdd)
    {
      REP32(asm volatile("s_add_u32 %0, %0, 1" : "+s"(ib));)
    }
    else if constexpr(KIND == kNop0)
    {
      REP32(asm volatile("s_nop 0");)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + l0 + l1 + l4.x + ib + lds[threadIdx.x];
  if(threadIdx.x == 0)
  {
    cyc[0] = t1 - t0;
  }
}

template<int KIND>
double run(const char * name, double loop_overhead, int waves = 1)
{
  double * out;
  long long * cyc;
  std::printf("%-62s ", name);
  std::fflush(stdout);
  (void)hipMalloc(&out, 256 * 8);
  double * gbuf;
  (void)hipMalloc(&gbuf, 4 * 4096 * 8);
  (void)hipMemset(gbuf, 0, 4 * 4096 * 8);
  (void)hipMalloc(&cyc, 8);
  const size_t lds_bytes = 4 * 4096 * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(issue_k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
  issue_k<KIND><<<1, 64 * waves, lds_bytes>>>(out, cyc, 1.0, gbuf);
  issue_k<KIND><<<1, 64 * waves, lds_bytes>>>(out, cyc, 1.0, gbuf);
  (void)hipDeviceSynchronize();
  long long h = 0;
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (static_cast<double>(h) / kTrips - loop_overhead) / kPer;
  std::printf("%6.2f cycles\n", per);
  std::fflush(stdout);
  (void)hipFree(out);
  (void)hipFree(gbuf);
  (void)hipFree(cyc);
  return per;
}

int main()
{
  std::printf("cycles per instruction, lone wave64, 32 per loop trip (s_memtime ticks; the s_nop row is the floor: one issue slot every ~4.2)\n");
  run<kNop0>("s_nop 0", 0);
  run<kFma>("v_fma_f64, 8 independent chains", 0);
  run<kFmaDep>("v_fma_f64, one dependent chain", 0);
  run<kMul>("v_mul_f64", 0);
  run<kAdd>("v_add_f64", 0);
  run<kMax>("v_max_f64", 0);
  run<kRcp>("v_rcp_f64", 0);
  run<kCmp>("v_cmp_nge_f64 -> vcc", 0);
  run<kCnd>("v_cndmask_b32, independent", 0);
  run<kCndDep>("v_cndmask_b32, dependent", 0);
  run<kDpp>("v_mov_b32_dpp quad_perm, independent", 0);
  run<kDppDep>("v_mov_b32_dpp quad_perm, dependent", 0);
  run<kMov32>("v_mov_b32", 0);
  run<kMov64>("v_mov_b64", 0);
  run<kXor>("v_xor_b32", 0);
  run<kMfma>("v_mfma_f64_4x4x4_4b_f64, 8 independent accumulators", 0);
  run<kDsRead>("ds_read_b64, lane-contiguous (+ one lgkmcnt(0) per 32)", 0);
  run<kDsReadSpread>("ds_read_b64, 4 x 16 contiguous doubles 6272 B apart", 0);
  run<kDsRead2>("ds_read2_b64 (two doubles per lane per instruction)", 0);
  run<kDsRead128>("ds_read_b128", 0);
  run<kDsWrite>("ds_write_b64, lane-contiguous", 0);
  run<kDsReadBcast>("ds_read_b64, 16 distinct addresses per wave (row broadcast)", 0);
  run<kDsWriteSame>("ds_write_b64, 16 lanes their own slot + 4 x 12 lanes one slot", 0);
  run<kDsMix>("8 ds_read_b64 + 2 ds_write_b64 interleaved (per instruction)", 0);
  run<kDsWrite2>("ds_write2_b64 (two adjacent doubles per lane)", 0);
  run<kDsWrite2st64>("ds_write2st64_b64 (two doubles 512 B apart per lane)", 0);
  run<kDsWrite128>("ds_write_b128", 0);
  run<kGlobalLoad>("global_load_dwordx2, coalesced, L2 hit (+ one vmcnt(0) per 32)", 0);
  run<kGlobalLoadX4>("global_load_dwordx4, coalesced, L2 hit", 0);
  run<kGlobalStore>("global_store_dwordx2, coalesced", 0);
  run<kDsWriteMirror>("ds_write_b64, lanes 16..63 mirror lanes 0..15 (same address, same value)", 0);
  run<kDsWrite16>("ds_write_b64, lanes 0..15 only (exec mask)", 0);
  run<kDsRead16>("ds_read_b64, lanes 0..15 only", 0);
  run<kGlobalLoadMirror>("global_load_dwordx2, lanes 16..63 mirror lanes 0..15", 0);
  run<kGlobalLoad16>("global_load_dwordx2, lanes 0..15 only", 0);
  std::printf("--- two waves of one workgroup (the forward pass: master + helper)\n");
  run<kDsWrite>("ds_write_b64", 0, 2);
  run<kDsWrite2st64>("ds_write2st64_b64", 0, 2);
  run<kDsWrite128>("ds_write_b128", 0, 2);
  run<kDsRead>("ds_read_b64", 0, 2);
  run<kDsRead128>("ds_read_b128", 0, 2);
  run<kGlobalLoad>("global_load_dwordx2", 0, 2);
  std::printf("--- four waves of one workgroup doing the same, each in its own 32 KB of LDS\n");
  run<kFma>("v_fma_f64, 8 independent chains", 0, 4);
  run<kMfma>("v_mfma_f64_4x4x4_4b_f64", 0, 4);
  run<kDsRead>("ds_read_b64, lane-contiguous", 0, 4);
  run<kDsReadSpread>("ds_read_b64, 4 x 16 contiguous doubles 6272 B apart", 0, 4);
  run<kDsReadBcast>("ds_read_b64, 16 distinct addresses per wave", 0, 4);
  run<kDsRead2>("ds_read2_b64", 0, 4);
  run<kDsRead128>("ds_read_b128", 0, 4);
  run<kDsWrite>("ds_write_b64, lane-contiguous", 0, 4);
  run<kDsWriteSame>("ds_write_b64, 16 lanes own slot + 4 x 12 lanes one slot", 0, 4);
  run<kDsMix>("8 ds_read_b64 + 2 ds_write_b64 interleaved", 0, 4);
  return 0;
}
