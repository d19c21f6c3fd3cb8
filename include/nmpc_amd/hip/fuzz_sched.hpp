// Wave-timing fuzz for the test builds (-DNMPC_AMD_FUZZ_SCHED=<seed>): every workgroup barrier / publish point of every kernel
// family calls fuzzSched() in front of and behind it, and each wave then sleeps a pseudo-random time of its own.
//
// Why: the kernels of this library hand data from wave to wave through LDS and HBM behind hand-placed `s_waitcnt` / `s_barrier`
// pairs.  A barrier that is MISSING shows only when one wave happens to run far enough ahead of another — on an idle chip the waves
// of a workgroup run almost in lock-step and a repetition soak (scripts/determinism_soak.py) exercises one timing over and over.
// Under the fuzz build a wave that leaves a barrier may stay behind by up to ~16 k cycles while its neighbours run on into the next
// phase, so an ordering that only a barrier rules out actually happens within a few launches, and the bit-for-bit / oracle parity
// tests catch what it breaks (tests/test_gpu_fuzz_sched.py; profiles/r05_fuzz_*.txt holds the experiment that re-opens a race
// this code base once shipped).
//
// The sleep length is a hash of (compile-time seed, the shader clock when the wave arrives, workgroup, wave, call site): different
// per wave and per launch, uniform within a wave (s_sleep is a scalar instruction).  It touches no data: results of a fuzz build
// are bit-identical to those of the product build — that is the test.  In product builds fuzzSched() is empty.
#pragma once

#include <hip/hip_runtime.h>

namespace nmpc_amd
{
#ifdef NMPC_AMD_FUZZ_SCHED
__device__ __forceinline__ void fuzzSched(unsigned site)
{
  unsigned t = static_cast<unsigned>(__builtin_amdgcn_s_memtime());
  unsigned wave = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  unsigned h = (static_cast<unsigned>(NMPC_AMD_FUZZ_SCHED) + 1u) * 0x9E3779B9u;
  h ^= t * 0x85EBCA6Bu;
  h ^= (blockIdx.x * 8u + wave + 1u) * 0xC2B2AE35u;
  h ^= (site + 1u) * 0x27D4EB2Fu;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  h = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(h)));
  // three wave in four: 0 .. 15 short naps (<= 1 k cycles: shuffles who arrives first); one in four: up to 255 more (~16 k cycles:
  // longer than any phase between two barriers of these kernels, so a wave really is a whole phase behind its neighbours)
  unsigned n = (h & 15u) + (((h >> 8) & 255u) & (0u - static_cast<unsigned>(((h >> 4) & 3u) == 0u)));
  // The nap loop is ONE inline-assembly statement: a loop written in C++ would split every basic block a barrier sits in, and with
  // it the compiler's floating-point contraction (an a * b in front of the barrier and the + c behind it fuse in the product build
  // and would not here) — the fuzz build has to differ from the product build in timing only, not in a rounding.
  asm volatile("s_cmp_eq_u32 %0, 0\n\t"
               "s_cbranch_scc1 2f\n"
               "1:\n\t"
               "s_sleep 1\n\t"
               "s_sub_u32 %0, %0, 1\n\t"
               "s_cmp_lg_u32 %0, 0\n\t"
               "s_cbranch_scc1 1b\n"
               "2:"
               : "+s"(n)
               :
               : "scc");
}
#else
__device__ __forceinline__ void fuzzSched(unsigned) {}
#endif

/** __syncthreads() of the kernels that use it bare, with the fuzz points around it. */
__device__ __forceinline__ void syncThreadsFuzzed(unsigned site)
{
  fuzzSched(site);
  __syncthreads();
  fuzzSched(site + 0x8000u);
}
} // namespace nmpc_amd
