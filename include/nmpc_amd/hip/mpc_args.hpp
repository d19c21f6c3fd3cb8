// Arguments of the receding-horizon advance step (mpc_kernels.hpp), shared with the type-erased model operations.
#pragma once

namespace nmpc_amd
{
namespace hip
{
/** Arguments of one advance step (tick).  Log pointers are batch-major device arrays [B][n_ticks][...] or nullptr. */
struct MpcAdvanceArgs
{
  int tick;
  int n_ticks;
  int shift_warm_start; //!< 1: shift pattern, 0: plant pattern
  int sim_substeps; //!< plant pattern: number of stateEq(t, x, u, sim_dt) steps per tick
  double sim_dt;
  int clamp_u0; //!< plant pattern: clamp u[0] to the handle's input limits first
  void * t0; //!< [Bp]            solver input: start time of the next solve   } elements of the problem's Scalar
  void * x0; //!< [tile][N][64]   solver input: initial state of the next solve } (double, or float: fp32 problem types)
  double * t_exact; //!< [Bp] current_t of the next tick in double (owned by the handle; never NULL)
  double * t_log; //!< [B][n_ticks]
  double * x_log; //!< [B][n_ticks][N]    state handed to the solve of this tick
  double * u0_log; //!< [B][n_ticks][MM]  first input of the solution (clamped in the plant pattern)
  int * iter_log; //!< [B][n_ticks]
  int * status_log; //!< [B][n_ticks]
  int * m0_log; //!< [B][n_ticks]         input dimension of the first timestep
};
} // namespace hip
} // namespace nmpc_amd
