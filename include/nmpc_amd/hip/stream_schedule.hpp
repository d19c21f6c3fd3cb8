// Streamed solves: a queue of N >> B instances through the B slots of ONE handle (reference: every DDPSolver
// object runs its own loop and stops when IT converges, DDPSolver.hpp:115-123 — a caller with many problems runs them through as
// many solver objects as it has cores and hands a finished object the next problem).
//
// The ragged-convergence schedule (ragged_schedule.hpp) frees the workgroups that a batch's converged instances held; this is its
// successor for a caller with MORE instances than slots: the freed slots are REFILLED.  A round is
//   solve    a resumable launch (DeviceBuffers::stream_mode 2): every slot whose instance still iterates runs at most `span`
//            further iterations — numbered per instance (row 4 of the parked state), so each one stops at ITS max_iter-th;
//   extract  the instances that finished in the round: X / U / cost of the half `sel` names, status, iteration count, last trace
//            row and dV go to the caller-facing output arrays (reference layouts, indexed by instance), the slot is free;
//   compact  ragged_compact_kernel / ragged_swap_kernel: the instances in mid-solve swapped into a dense prefix [0, n_run);
//   refill   the slots from the next multiple of 64 on take the next instances of the queue (inputs converted into the slot's
//            tile-major rows); the NEXT round's launch starts the workgroups of that region with the initial rollout
//            (DDPSolver.hpp:83-95) instead of resuming them — which is why the region starts on a workgroup boundary of both
//            kernel families.
// Everything is queued on the handle's stream; the kernels read the prefix length, the queue cursor and the count of finished
// instances from device memory, the host looks at that count every few rounds.  An instance's iterations are the same instructions
// on the same values whichever slot it sits in and whoever its neighbours are (the property the ragged schedule rests on:
// tests/test_gpu_ragged.py), so every instance returns the bits of its lone solve (tests/test_gpu_stream.py).
#pragma once

#include <hip/hip_runtime.h>

#include <nmpc_amd/hip/ddp_kernels.hpp>

namespace nmpc_amd
{
namespace hip
{
/** Device words of a streamed solve. */
enum StreamWord
{
  kSwPrefix = 0, //!< slots [0, prefix) hold instances (the next solve launch's *n_active)
  kSwRun, //!< ... of which [0, run) are in mid-solve after the compaction (ragged_compact_kernel writes it: n_active[1])
  kSwFirst, //!< first slot of the region filled last (a multiple of 64)
  kSwCursor, //!< instances handed out so far
  kSwDone, //!< instances extracted so far
  kSwTotal, //!< N
  kSwTaken, //!< instances the last fill took (the next plan moves the cursor past them)
  kSwCount
};

/** Caller-facing arrays of a streamed solve (device memory, reference layouts, indexed by instance). */
struct StreamArrays
{
  const double * t0; //!< [N] or nullptr
  const double * x0; //!< [N][n]
  const double * u_init; //!< [N][T][MM]
  double * X; //!< [N][T+1][n]
  double * U; //!< [N][T][MM]
  double * cost; //!< [N][T+1]
  double * trace_last; //!< [N][NMPC_HIP_NTRACE]
  double * dV; //!< [N][2]
  int * status; //!< [N]
  int * iters; //!< [N]
};

__global__ void stream_begin_kernel(int * w, int n_total)
{
  if(threadIdx.x == 0 && blockIdx.x == 0)
  {
    w[kSwPrefix] = 0;
    w[kSwRun] = 0;
    w[kSwFirst] = 0;
    w[kSwCursor] = 0;
    w[kSwDone] = 0;
    w[kSwTotal] = n_total;
    w[kSwTaken] = 0;
  }
}

/** The instances that finished in the round leave their slots.  One workgroup per slot of the prefix; thread = row.
    \param id  [Bp] instance in the slot, -1: none */
__global__ __launch_bounds__(256) void stream_extract_kernel(const DeviceBuffers buf, StreamArrays out, int * __restrict__ id,
                                                             int * __restrict__ w, int n, int mm)
{
  const int p = blockIdx.x;
  if(p >= w[kSwPrefix])
  {
    return;
  }
  const int q = id[p];
  const size_t tile = static_cast<size_t>(p) >> 6, ln = static_cast<size_t>(p) & 63;
  if(q < 0 || buf.resume[(tile * kResumeRows + 3) * 64 + ln] != 0.0)
  {
    return; // empty, or still iterating
  }
  const int T = buf.T;
  const size_t rows_x = static_cast<size_t>(T + 1) * n, rows_u = static_cast<size_t>(T) * mm, rows_c = static_cast<size_t>(T + 1);
  const int sel = buf.sel[p];
  const double * Xs = buf.X + ((tile * 2 + sel) * rows_x) * 64 + ln;
  const double * Us = buf.U + ((tile * 2 + sel) * rows_u) * 64 + ln;
  const double * Cs = buf.cost + ((tile * 2 + sel) * rows_c) * 64 + ln;
  for(size_t r = threadIdx.x; r < rows_x; r += 256)
  {
    out.X[static_cast<size_t>(q) * rows_x + r] = Xs[r * 64];
  }
  for(size_t r = threadIdx.x; r < rows_u; r += 256)
  {
    out.U[static_cast<size_t>(q) * rows_u + r] = Us[r * 64];
  }
  for(size_t r = threadIdx.x; r < rows_c; r += 256)
  {
    out.cost[static_cast<size_t>(q) * rows_c + r] = Cs[r * 64];
  }
  if(threadIdx.x < NMPC_HIP_NTRACE)
  {
    out.trace_last[static_cast<size_t>(q) * NMPC_HIP_NTRACE + threadIdx.x] = buf.trace_last[(tile * NMPC_HIP_NTRACE + threadIdx.x) * 64 + ln];
  }
  if(threadIdx.x < 2)
  {
    out.dV[static_cast<size_t>(q) * 2 + threadIdx.x] = buf.dV[(tile * 2 + threadIdx.x) * 64 + ln];
  }
  if(threadIdx.x == 0)
  {
    out.status[q] = buf.status[p];
    out.iters[q] = buf.iters[p];
    id[p] = -1;
    atomicAdd(&w[kSwDone], 1);
  }
}

/** After the compaction: where the next instances go (and the cursor past the ones the previous fill took).  One thread. */
__global__ void stream_plan_kernel(int * w, int n_slots)
{
  if(threadIdx.x == 0 && blockIdx.x == 0)
  {
    w[kSwCursor] += w[kSwTaken];
    const int run = w[kSwRun];
    const int first = (run + 63) & ~63;
    const int room = n_slots > first ? n_slots - first : 0;
    const int left = w[kSwTotal] - w[kSwCursor];
    const int take = left < room ? left : room;
    w[kSwFirst] = first;
    w[kSwPrefix] = take > 0 ? first + take : run;
    w[kSwTaken] = take;
  }
}

/** The slots behind the dense prefix: [first, first + take) take the instances cursor .. cursor + take - 1 — current_t, current_x
    and initial_u_list into the slot's tile-major rows (half 0 of U; `sel` 0), the parked state says "running, no iteration yet" —
    the others are marked empty.  grid = (slots, row blocks); thread = row. */
__global__ __launch_bounds__(256) void stream_fill_kernel(const DeviceBuffers buf, StreamArrays in, int * __restrict__ id,
                                                          const int * __restrict__ w, int n, int mm, double * t0_slots, double * x0_slots)
{
  const int p = blockIdx.x;
  const int run = w[kSwRun], first = w[kSwFirst], prefix = w[kSwPrefix], cursor = w[kSwCursor];
  if(p < run)
  {
    return; // in mid-solve
  }
  const size_t tile = static_cast<size_t>(p) >> 6, ln = static_cast<size_t>(p) & 63;
  const bool fill = w[kSwTaken] > 0 && p >= first && p < prefix;
  if(!fill)
  {
    if(threadIdx.x == 0 && blockIdx.y == 0)
    {
      buf.resume[(tile * kResumeRows + 3) * 64 + ln] = 0.0;
      id[p] = -1;
    }
    return;
  }
  const int q = cursor + (p - first);
  const int T = buf.T;
  const size_t rows_u = static_cast<size_t>(T) * mm;
  double * U0 = buf.U + ((tile * 2 + 0) * rows_u) * 64 + ln;
  for(size_t r = static_cast<size_t>(blockIdx.y) * 256 + threadIdx.x; r < rows_u; r += static_cast<size_t>(gridDim.y) * 256)
  {
    U0[r * 64] = in.u_init[static_cast<size_t>(q) * rows_u + r];
  }
  if(blockIdx.y == 0)
  {
    if(threadIdx.x < n)
    {
      x0_slots[(tile * n + threadIdx.x) * 64 + ln] = in.x0[static_cast<size_t>(q) * n + threadIdx.x];
    }
    if(threadIdx.x == 0)
    {
      t0_slots[tile * 64 + ln] = in.t0 ? in.t0[q] : 0.0;
      buf.sel[p] = 0;
      buf.status[p] = 0;
      buf.iters[p] = 0;
      buf.resume[(tile * kResumeRows + 3) * 64 + ln] = 1.0;
      buf.resume[(tile * kResumeRows + 4) * 64 + ln] = 0.0;
      id[p] = q;
    }
  }
}

} // namespace hip
} // namespace nmpc_amd
