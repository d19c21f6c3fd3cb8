// gfx950 device code of the receding-horizon (MPC) driver: what the reference's callers do BETWEEN two solve() calls,
// executed on the device so that thousands of closed-loop rollouts advance without a host round trip.
//
// The reference has two caller patterns:
//   shift pattern   TestDDPBipedal.cpp:243-268, TestDDPVerticalMotion.cpp:290-326, TestDDPCentroidalMotion.cpp:307-347
//                   current_x = controlData().x_list[1]; current_u_list = controlData().u_list with the first entry
//                   erased and the last one repeated (or a zero vector of the terminal dimension when the input
//                   dimension changes at the end of the horizon); current_t += dt
//   plant pattern   TestDDPCartPole.cpp:323-346,388-403
//                   current_u = controlData().u_list[0] clamped to the input limits; the plant is integrated with
//                   ddp_problem->stateEq(t, x, u, sim_dt) (the problem's 4-argument overload) for several sim steps;
//                   initial_u_list = controlData().u_list (not shifted)
// One lane per instance, same tile-major arrays as the solver kernels (ddp_kernels.hpp); the kernel rewrites the solver's
// own input buffers (t0, x0, half 0 of U), so the next solve is launched without any ingest.
#pragma once

#include <type_traits>
#include <utility>

#include <nmpc_amd/hip/ddp_kernels.hpp>

#include <nmpc_amd/hip/mpc_args.hpp>

namespace nmpc_amd
{
namespace hip
{

/** Does the problem offer the plant step stateEq(t, x, u, dt) (TestDDPCartPole.cpp:63-98 has it)? */
template<class Problem, class = void>
struct HasPlantStep : std::false_type
{
};
template<class Problem>
struct HasPlantStep<Problem,
                    std::void_t<decltype(std::declval<const Problem &>().stateEq(
                        0.0,
                        std::declval<const typename Problem::StateDimVector &>(),
                        std::declval<const typename Problem::InputDimVector &>(),
                        0.0))>> : std::true_type
{
};

/** Rows 0 ... n - 1 of one lane's column (stride kLanesPerBlock) from src to dst, sixteen rows requested before the first is stored —
    as `dst[r] = src[r]` per trip the copy was a round trip to L2 per row, T of them in a row per tick (the store may alias the next
    load as far as the compiler knows).  dst may be the column src starts MM rows into (the in-place shift of the warm start): every
    row is read before a batch that reaches it is stored, ascending. */
template<class S>
__device__ __forceinline__ void copyColumn(S * dst, const S * src, size_t n)
{
  constexpr size_t LW = kLanesPerBlock;
  constexpr int kBatch = 16;
  for(size_t r0 = 0; r0 < n; r0 += kBatch)
  {
    S v[kBatch];
#pragma unroll
    for(int k = 0; k < kBatch; k++)
    {
      v[k] = src[(r0 + k < n ? r0 + k : n - 1) * LW];
    }
#pragma unroll
    for(int k = 0; k < kBatch; k++)
    {
      if(r0 + k < n)
      {
        dst[(r0 + k) * LW] = v[k];
      }
    }
  }
}

/** \tparam S element type of the handle's arrays = Problem::Scalar (the logs are double whatever it is: the C-ABI side) */
template<class Problem, class S = typename Problem::Scalar>
__global__ __launch_bounds__(kLanesPerBlock) void mpc_advance_kernel(const Problem shared_problem,
                                                                     const DeviceBuffersT<S> buf,
                                                                     const MpcAdvanceArgs args)
{
  constexpr int N = Problem::kStateDim;
  constexpr int M = Problem::kInputDimMax;
  constexpr int MM = (M > 0) ? M : 1;
  constexpr size_t LW = kLanesPerBlock;
  const int b = blockIdx.x * kLanesPerBlock + threadIdx.x;
  if(b >= buf.B)
  {
    return;
  }
  const Problem problem = instanceProblem(shared_problem, buf, b); // its own object if the batch has per-instance ones
  const size_t tile = static_cast<size_t>(b) / LW, lane = static_cast<size_t>(b) % LW;
  const int T = buf.T;
  const size_t rows_x = static_cast<size_t>(T + 1) * N, rows_u = static_cast<size_t>(T) * MM;
  const int sel = buf.sel[b];
  const S * Xs = buf.X + ((tile * 2 + sel) * rows_x) * LW + lane; // control_data_.x_list, row r at Xs[r * 64]
  const S * Us = buf.U + ((tile * 2 + sel) * rows_u) * LW + lane; // control_data_.u_list
  S * U0 = buf.U + ((tile * 2 + 0) * rows_u) * LW + lane; // initial_u_list of the next solve
  S * x0 = static_cast<S *>(args.x0) + (tile * N) * LW + lane;
  // current_t lives in DOUBLE whatever the problem's scalar: an fp32 handle's t0 array holds its rounding, the exact value is carried
  // from tick to tick in the handle's own per-instance array args.t_exact (written below; until round 5 it rode in the caller's
  // optional time log) — accumulated in float, sim_dt = 0.01 at t ~ 100 s is 1e-3 relative per step off.  Double
  // handles: the same additions as before, bit for bit.
  const double t_exact = (args.tick == 0) ? static_cast<double>(static_cast<S *>(args.t0)[b]) : args.t_exact[b];
  const S t = static_cast<S>(t_exact);
  const int m0 = buf.input_dim[(tile * T + 0) * LW + lane];
  const size_t log_at = static_cast<size_t>(b) * args.n_ticks + args.tick;

  typename Problem::StateDimVector x;
  typename Problem::InputDimVector u0;
  u0.resize(m0);
  for(int j = 0; j < N; j++)
  {
    x[j] = Xs[static_cast<size_t>(j) * LW];
  }
  for(int a = 0; a < MM; a++)
  {
    u0[a] = (a < m0) ? Us[static_cast<size_t>(a) * LW] : S(0);
  }
  if(!args.shift_warm_start && args.clamp_u0)
  {
    for(int a = 0; a < MM; a++)
    {
      if(a < m0)
      {
        u0[a] = fmin(fmax(u0[a], static_cast<S>(inputLimitLo(buf, b, 0, a))), static_cast<S>(inputLimitHi(buf, b, 0, a))); // cwiseMax(lower).cwiseMin(upper), :394
      }
    }
  }
  if(args.t_log)
  {
    args.t_log[log_at] = t_exact;
  }
  if(args.x_log)
  {
    for(int j = 0; j < N; j++)
    {
      args.x_log[log_at * N + j] = x[j];
    }
  }
  if(args.u0_log)
  {
    for(int a = 0; a < MM; a++)
    {
      args.u0_log[log_at * MM + a] = u0[a];
    }
  }
  if(args.iter_log)
  {
    args.iter_log[log_at] = buf.iters[b];
  }
  if(args.status_log)
  {
    args.status_log[log_at] = buf.status[b];
  }
  if(args.m0_log)
  {
    args.m0_log[log_at] = m0;
  }

  double t_next = t_exact;
  if(args.shift_warm_start)
  {
    for(int j = 0; j < N; j++)
    {
      x0[static_cast<size_t>(j) * LW] = Xs[static_cast<size_t>(N + j) * LW]; // x_list[1]
    }
    // erase(begin()); push_back(back()) — or a zero vector when the terminal input dimension differs.  When sel == 0
    // source and destination are the same column: row i + 1 is read before row i is written, ascending i.
    copyColumn(U0, Us + static_cast<size_t>(MM) * LW, static_cast<size_t>(T - 1) * MM);
    int last_m = MM, term_m = MM;
    if constexpr(Problem::kDynamicInput)
    {
      last_m = buf.input_dim[(tile * T + (T - 1)) * LW + lane];
      term_m = problem.inputDim(t + T * problem.dt());
    }
    for(int a = 0; a < MM; a++)
    {
      const S keep = Us[(static_cast<size_t>(T - 1) * MM + a) * LW];
      U0[(static_cast<size_t>(T - 1) * MM + a) * LW] = (last_m == term_m) ? keep : S(0);
    }
    t_next = t_exact + static_cast<double>(problem.dt());
  }
  else
  {
    if constexpr(HasPlantStep<Problem>::value)
    {
      for(int s = 0; s < args.sim_substeps; s++)
      {
        x = problem.stateEq(static_cast<S>(t_next), x, u0, static_cast<S>(args.sim_dt));
        t_next += args.sim_dt;
      }
    }
    for(int j = 0; j < N; j++)
    {
      x0[static_cast<size_t>(j) * LW] = x[j];
    }
    if(sel != 0)
    {
      copyColumn(U0, Us, rows_u);
    }
  }
  static_cast<S *>(args.t0)[b] = static_cast<S>(t_next);
  args.t_exact[b] = t_next;
}
} // namespace hip
} // namespace nmpc_amd
