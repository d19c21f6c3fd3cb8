// Registry that connects a problem TYPE (a functor class derived from nmpc_amd::DDPProblem, compiled into
// gfx950 code together with the solver kernel) to the NAME the C-ABI of include/nmpc_hip_ddp.h uses.
//
// A user model is added without touching the library: write the problem header, then in one .hip file
//     #include <nmpc_amd/hip/model_registry.hpp>
//     #include "MyProblem.hpp"
//     NMPC_AMD_REGISTER_PROBLEM(MyProblem);          // needs: static constexpr const char * kName
// compile it with hipcc --offload-arch=gfx950 and link (or dlopen) it next to libnmpc_hip_ddp.so.
#pragma once

#include <atomic>

#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include <nmpc_amd/hip/model_ops.hpp>
#include <nmpc_amd/hip/ddp_kernels.hpp>
#include <nmpc_amd/hip/ddp_kernels_2w.hpp>
#include <nmpc_amd/hip/ddp_kernels_quad.hpp>
#include <nmpc_amd/hip/ddp_kernels_wpi.hpp>
#include <nmpc_amd/hip/ddp_kernels_tile64.hpp>
#include <nmpc_amd/hip/mpc_kernels.hpp>

namespace nmpc_amd
{
namespace hip
{
template<class Problem>
struct ModelOpsFor
{
  static void defaultParams(void * out)
  {
    new(out) Problem();
  }
  /** Lane mapping: the 2-wave (master + helper, LDS-staged) kernel whenever its records fit in LDS, else the
      single-wave kernel.  NMPC_HIP_DDP_KERNEL=1w forces the single-wave kernel (A/B measurements, tests). */
  //! both record layouts (the box-constrained one is the larger: it carries the input limits) have to fit
  static constexpr bool kTwoWaveFits = PairSolver<Problem, false>::kFits && PairSolver<Problem, true>::kFits;
  static bool useTwoWave()
  {
    return kTwoWaveFits && !launchKnobs().kernelIs("1w");
  }
  /** Wave-per-instance (matrix-core) kernel: the shapes whose blocks fill a 16 x 16 tile; unconstrained solves only
      (checked at launch). */
  static constexpr bool kWpiShape =
      Problem::kStateDim >= 9 && Problem::kStateDim <= 16 && Problem::kInputDimMax >= 1 && Problem::kInputDimMax <= 16;
  //! box-constrained solves on the wave-per-instance kernel: register-path shapes only (static m <= 8)
  static constexpr bool kWpiBoxQP = kWpiShape && !Problem::kDynamicInput && Problem::kInputDimMax <= 8;
  static bool useWpi(bool constrained)
  {
    const LaunchKnobs knobs = launchKnobs();
    return kWpiShape && (!constrained || kWpiBoxQP) && !knobs.kernelIs("1w") && knobs.have_workspace != 0;
  }
  /** fp64 tile kernel (ddp_kernels_tile64.hpp: groups of up to 32 instances per workgroup, derivatives LDS-resident, backward
      pass on v_mfma_f64_16x16x4 in natural layout, BoxQP included): 5 <= n <= 15, m <= 16, static or inputDim(t) (kTile64Big: m > 8 or run-time m, gains in natural layout).  It
      replaces the wave-per-instance kernel on these shapes; NMPC_HIP_DDP_KERNEL=wpi / 1w select the older kernels (A/B). */
  static constexpr bool kTile64Shape = Problem::kStateDim >= 5 && Problem::kStateDim <= 15 && Problem::kInputDimMax >= 1
                                       && Problem::kInputDimMax <= 16;
  //! m > 8 or inputDim(t) (the reference's centroidal-motion problem, 16 / 0): the gains are computed in natural layout across
  //! the wave (TileSolver64::stepGainsNatural, round 4); unconstrained solves only — their BoxQP stays on the lane kernel
  static constexpr bool kTile64Big = kTile64Shape && (Problem::kDynamicInput || Problem::kInputDimMax > 8);
  //! Below this batch the wave-per-instance kernel is still the (marginally) faster one where both exist (n >= 9).  Until round
  //! 4 the threshold was 1025: a group's sweep could not go faster than its model wave linearises ONE timestep per pass
  //! (33 k cycles for the manipulator, whatever the group size).  The model wave now linearises a CHUNK of timesteps per pass
  //! — as many (slot, timestep) pairs as its 64 lanes and the record area hold (backwardSweepModel) — and the two kernels
  //! are level up to 512 instances, the tile kernel ahead from there (scripts/tile64_chunk_ab.py,
  //! profiles/r04_tile64_chunk_ab.txt: manipulator 1.38 against 1.37 ms at 256, 1.41 / 1.42 at 512, 1.45 / 1.57 at 1024,
  //! 1.82 / 3.04 at 2048).
  //! Round 4, second half (the later step sizes ride along with the first, slot-major ring): ahead from 64 instances on
  //! (profiles/r04*_tile64_chunk_ab.txt: manipulator tile64 / wave-per-instance 1.07 at 16 instances, 0.96 at 64, 0.91 at 256,
  //! 0.31 at 8192; quadrotor 0.96 / 0.89 / 0.81 / 0.40).
  static constexpr int kTile64MinBatch = 64;
  //! box-constrained solves: round 3's threshold (the QP dominates their timestep; small batches have not been re-measured)
  static constexpr int kTile64MinBatchBoxQP = 1025;
  //! ... and for m > 4 (the QP grows with m^3; measured at 8192 instances only: profiles/r05_constrained_tile64_ab.txt)
  static constexpr int kTile64MinBatchBoxQPWide = 4096;
  //! Small models: the lane kernels keep a timestep's blocks in registers (n (n + m) <= 48 — planar VTOL: 70 - 80 scratch
  //! instructions; the quadrotor's 192 make 1500) and hold 64 instances per wavefront, several wavefronts per SIMD, so their
  //! time barely grows with the batch, while the tile kernel's (<= 35 instances per CU and round) grows linearly.  Measured
  //! (scripts/lane_vs_tile_ab.py, profiles/r05_lane_vs_tile_ab.txt; planar VTOL, T 60, 6 iterations; tile / lane ms):
  //! unconstrained 0.94 / 0.98 at 1024, 1.42 / 1.00 at 2048, 2.87 / 1.04 at 8192, 11.0 / 1.62 at 32768; box 3.78 / 4.81 at
  //! 4096, 6.04 / 4.74 at 8192, 23.2 / 5.75 at 32768.  Above these batches such shapes go back to the lane kernels.
  static constexpr bool kLaneKeepsUp =
      !Problem::kDynamicInput && Problem::kStateDim * (Problem::kStateDim + Problem::kInputDimMax) <= 48;
  static constexpr int kTile64MaxBatchSmall = 1024;
  static constexpr int kTile64MaxBatchSmallBoxQP = 6143;
  static bool useTile64(bool constrained, int batch)
  {
    const LaunchKnobs knobs = launchKnobs();
    // (without the per-instance workspace — the allocation failed at create — the gain records and the candidate scratch have
    // nowhere to live: the lane kernels, which need none, take the solve)
    if(!kTile64Shape || knobs.kernelIs("1w") || knobs.kernelIs("wpi") || (knobs.kernelIs("2w") && kTwoWaveFits) || knobs.have_workspace == 0)
    {
      return false;
    }
    batch = knobs.batchFor(batch);
    if(kTile64Big)
    {
      return !constrained; // (every batch size: the wave-per-instance kernel takes these gains through LDS, 3 - 4 x slower)
    }
    if(knobs.kernelIs("tile64"))
    {
      return true;
    }
    if(kLaneKeepsUp && batch > (constrained ? kTile64MaxBatchSmallBoxQP : kTile64MaxBatchSmall))
    {
      return false;
    }
    if(kWpiShape && batch < (constrained ? kTile64MinBatchBoxQP : kTile64MinBatch))
    {
      return false;
    }
    // Box-constrained solves: every lane of a wave runs the BoxQP of its instance (boxQPMasked), a matrix wave of the tile
    // kernel one after the other for its up to five instances.  Measured (scripts/constrained_tile64_ab.py, 8192 instances, 4
    // iterations): quadrotor (m = 4) tile 7.2 ms against 11.1 ms on the wave-per-instance kernel, manipulator (m = 7) 23.7
    // against 17.1 — the QP grows with m^3 and the wave-per-instance kernel hides it behind more waves per SIMD.  So the tile
    // kernel takes the constrained solves up to m = 4 (and all of 5 <= n <= 8, where no other matrix-core kernel exists).
    // Round 5: the QPs of a matrix wave's five slots are solved together, lane = slot (TileSolver64::qpBatch) — manipulator box 24.3 ->
    // 12.2 ms (wave-per-instance kernel 15.1), quadrotor box 7.2 -> 4.8 (11.1): m > 4 goes to the tile kernel on full chips too.
    return !(constrained && kWpiBoxQP && Problem::kInputDimMax > 4 && batch < kTile64MinBatchBoxQPWide);
  }
  /** Where k_list_ / K_list_ are after a solve: the tile kernel leaves instance-major records in the workspace. */
  static int gainLayoutOf(int batch, int constrained)
  {
    return useTile64(constrained != 0, batch) ? 1 : 0;
  }
  /** One problem object per instance: every kernel family but the single-wavefront lane kernel has the instantiation. */
  static int ownProblemsSupported(int batch, int constrained)
  {
    const int padded = (batch + kLanesPerBlock - 1) / kLanesPerBlock * kLanesPerBlock;
    return (useTile64(constrained != 0, batch) || useWpi(constrained != 0) || useQuad(padded, true) || useTwoWave()) ? 1 : 0;
  }
  /** Resumable launches (the ragged-convergence schedule): the quad kernel with the step-size-parallel line search and the
      two-wave kernel, one problem object for all instances. */
  static int resumableSupported(int batch, const nmpc_hip_ddp_config & cfg, int own_problems)
  {
    if(own_problems)
    {
      return 0;
    }
    const int padded = (batch + kLanesPerBlock - 1) / kLanesPerBlock * kLanesPerBlock;
    const bool con = cfg.with_input_constraint != 0;
    if constexpr(kTile64Shape)
    {
      if(useTile64(con, batch))
      {
        return 0;
      }
    }
    if constexpr(kWpiShape)
    {
      if(useWpi(con))
      {
        return 0;
      }
    }
    if constexpr(kQuadShape)
    {
      if(useQuad(padded, false))
      {
        const bool fan = cfg.line_search_fan_out == 1 || (cfg.line_search_fan_out == 0 && cfg.max_iter > fanOutAutoMaxIter());
        return (con || fan) ? 1 : 0;
      }
    }
    return (kTwoWaveFits && useTwoWave()) ? 1 : 0;
  }
  static size_t wpiWorkspaceDoubles(int T)
  {
    if constexpr(kWpiShape && kTile64Shape)
    {
      const size_t a = WaveSolver<Problem>::workspaceDoubles(T), b = TileSolver64<Problem>::workspaceDoubles(T);
      return a > b ? a : b;
    }
    else if constexpr(kWpiShape)
    {
      return WaveSolver<Problem>::workspaceDoubles(T);
    }
    else if constexpr(kTile64Shape)
    {
      return TileSolver64<Problem>::workspaceDoubles(T);
    }
    else if constexpr(QuadSolver<Problem, false>::kShape)
    {
      // fan-out scratch of the quad kernel's line search (PairSolver::FanDest): three more candidate trajectories.
      // NMPC_HIP_DDP_FAN_SCRATCH=0: none (A/B measurements, tests of the path taken when the allocation fails)
      if(launchKnobs().fan_scratch == 0)
      {
        return 0;
      }
      return 3 * (static_cast<size_t>(T + 1) * (Problem::kStateDim + 1) + static_cast<size_t>(T) * Problem::kInputDimMax);
    }
    else
    {
      return 0;
    }
  }
  /** Quad kernel (matrix-core backward pass, 16 instances per workgroup): n <= 4, one input.  It wins while its
      workgroups fit on the chip in one round (one per CU: 16 * 256 instances); larger batches go to the 2-wave kernel,
      whose 64-instance workgroups keep the latency flat up to 16384 instances.  NMPC_HIP_DDP_KERNEL=quad / 2w force. */
  static constexpr bool kQuadShape = QuadSolver<Problem, false>::kShape;
  static constexpr int kQuadMaxBatch = 4096;
  //! Configuration::line_search_fan_out = 0 (automatic): solves with max_iter above this use the step-size-parallel search.
  //! -1 = always: since the lane groups fan out from the first pass on and an accepted rollout is adopted from the fan-out
  //! scratch (PairSolver::adoptFanOut), the parallel search is the faster one in the nominal regime too.
  //! NMPC_HIP_DDP_FAN_AUTO=<max_iter> overrides (A/B measurements).
  static constexpr int kQuadFanOutAutoMaxIter = -1;
  static int fanOutAutoMaxIter()
  {
    const LaunchKnobs knobs = launchKnobs();
    return knobs.has_fan_auto ? knobs.fan_auto : kQuadFanOutAutoMaxIter;
  }
  static bool useQuad(int batch_padded, bool own)
  {
    const LaunchKnobs knobs = launchKnobs();
    (void)own; // per-instance problem objects have their own instantiation of the quad kernel
    if(!kQuadShape || knobs.kernelIs("1w") || knobs.kernelIs("2w"))
    {
      return false;
    }
    if(knobs.dispatch_batch > 0)
    {
      batch_padded = (knobs.dispatch_batch + kLanesPerBlock - 1) / kLanesPerBlock * kLanesPerBlock;
    }
    return batch_padded <= kQuadMaxBatch || knobs.kernelIs("quad");
  }
  static const char * kernelName(int batch, const nmpc_hip_ddp_config & cfg)
  {
    const int constrained = cfg.with_input_constraint != 0 ? 1 : 0;
    const int padded = (batch + kLanesPerBlock - 1) / kLanesPerBlock * kLanesPerBlock;
    if(useQuad(padded, false))
    {
      return "ddp_solve_quad_kernel";
    }
    if(useTile64(constrained != 0, batch))
    {
      return "ddp_solve_tile64_kernel";
    }
    if(useWpi(constrained != 0))
    {
      return "ddp_solve_wpi_kernel";
    }
    return useTwoWave() ? "ddp_solve_tpi2w_kernel" : "ddp_solve_tpi_kernel";
  }
  static hipError_t launchSolve(const void * params,
                                const nmpc_hip_ddp_config & cfg,
                                const DeviceBuffers & buf,
                                hipStream_t stream)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    const int grid = buf.Bp / kLanesPerBlock;
    const bool own = buf.params_batch != nullptr; // per-instance problem objects: separate instantiations (kOwnProblem)
    const bool con = cfg.with_input_constraint != 0;
    if(buf.iter_end > 0 && !resumableSupported(buf.B, cfg, own ? 1 : 0))
    {
      return hipErrorNotSupported; // (capi.hip asks resumable_supported first)
    }
    if constexpr(kTile64Shape)
    {
      if(useTile64(con, buf.B))
      {
        if(buf.wpi_ws == nullptr)
        {
          return hipErrorOutOfMemory; // the gain records live in the workspace (ModelOps::wpi_workspace_doubles)
        }
        if constexpr(!kTile64Big)
        {
          if(con && own)
          {
            return launchTile64<Problem, true, true>(problem, cfg, buf, stream);
          }
          if(con)
          {
            return launchTile64<Problem, true, false>(problem, cfg, buf, stream);
          }
        }
        if(own)
        {
          return launchTile64<Problem, false, true>(problem, cfg, buf, stream);
        }
        return launchTile64<Problem, false, false>(problem, cfg, buf, stream);
      }
    }
    if constexpr(kWpiShape)
    {
      if(useWpi(con) && buf.wpi_ws != nullptr)
      {
        constexpr size_t wpi_lds = WaveSolver<Problem, false>::kLdsBytes; // same layout with and without BoxQP
        const dim3 g(buf.B), blk(kLanesPerBlock);
        if constexpr(kWpiBoxQP)
        {
          if(con && own)
          {
            hipLaunchKernelGGL((ddp_solve_wpi_kernel<Problem, true, true>), g, blk, wpi_lds, stream, problem, cfg, buf);
            return hipGetLastError();
          }
          if(con)
          {
            hipLaunchKernelGGL((ddp_solve_wpi_kernel<Problem, true, false>), g, blk, wpi_lds, stream, problem, cfg, buf);
            return hipGetLastError();
          }
        }
        if(own)
        {
          hipLaunchKernelGGL((ddp_solve_wpi_kernel<Problem, false, true>), g, blk, wpi_lds, stream, problem, cfg, buf);
        }
        else
        {
          hipLaunchKernelGGL((ddp_solve_wpi_kernel<Problem, false, false>), g, blk, wpi_lds, stream, problem, cfg, buf);
        }
        return hipGetLastError();
      }
    }
    if constexpr(kQuadShape)
    {
      if(useQuad(buf.Bp, own))
      {
        constexpr size_t quad_lds = QuadSolver<Problem, false>::kLdsBytes;
        const dim3 g(buf.Bp / kQuadInstances), blk(kQuadWaves * 64);
        // > 64 KB of dynamic LDS has to be requested per kernel and device (once: remembered per device ordinal)
        static std::atomic<bool> requested[64] = {}; // (several host threads may launch at once; the setup is idempotent)
        int dev = 0;
        if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        {
          return hipErrorInvalidDevice;
        }
        if(!requested[dev].load(std::memory_order_acquire))
        {
          const void * variants[8] = {reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, false, false, true, true>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, true, false, true, true>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, false, false>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, true, false>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, false, true>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, true, true>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, false, false, true>),
                                      reinterpret_cast<const void *>(&ddp_solve_quad_kernel<Problem, false, true, true>)};
          for(const void * fn : variants)
          {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(quad_lds));
            if(e != hipSuccess)
            {
              return e;
            }
          }
          requested[dev].store(true, std::memory_order_release);
        }
        // step-size-parallel line search for unconstrained solves: on request, or (0 = automatic) for long solves
        const bool fan = cfg.line_search_fan_out == 1 || (cfg.line_search_fan_out == 0 && cfg.max_iter > fanOutAutoMaxIter());
        if(buf.iter_end > 0)
        {
          // a resumable launch (resumableSupported() has said yes: shared problem object, fan-out line search)
          if(own || !(con || fan))
          {
            return hipErrorNotSupported;
          }
          if(con)
          {
            hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, true, false, true, true>), g, blk, quad_lds, stream, problem, cfg, buf);
          }
          else
          {
            hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, false, false, true, true>), g, blk, quad_lds, stream, problem, cfg, buf);
          }
        }
        else if(con && own)
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, true, true>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        else if(con)
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, true, false>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        else if(fan && own)
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, false, true, true>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        else if(fan)
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, false, false, true>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        else if(own)
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, false, true>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        else
        {
          hipLaunchKernelGGL((ddp_solve_quad_kernel<Problem, false, false>), g, blk, quad_lds, stream, problem, cfg, buf);
        }
        return hipGetLastError();
      }
    }
    if(useTwoWave())
    {
      if constexpr(kTwoWaveFits)
      {
        // the box-constrained record also carries the input limits (PairSolver::kBwdRec): LDS is sized per instantiation
        constexpr size_t lds_bytes = PairSolver<Problem, false>::kLdsBytes;
        constexpr size_t lds_bytes_con = PairSolver<Problem, true>::kLdsBytes;
        static_assert(lds_bytes <= 64 * 1024 && lds_bytes_con <= 64 * 1024,
                      "kTwoWaveFits keeps the records of both layouts within the default dynamic LDS limit");
        const dim3 g(grid), blk(2 * kLanesPerBlock);
        if(buf.iter_end > 0)
        {
          if(own)
          {
            return hipErrorNotSupported;
          }
          if(con)
          {
            hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, true, false, true>), g, blk, lds_bytes_con, stream, problem, cfg, buf);
          }
          else
          {
            hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, false, false, true>), g, blk, lds_bytes, stream, problem, cfg, buf);
          }
        }
        else if(con && own)
        {
          hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, true, true>), g, blk, lds_bytes_con, stream, problem, cfg, buf);
        }
        else if(con)
        {
          hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, true, false>), g, blk, lds_bytes_con, stream, problem, cfg, buf);
        }
        else if(own)
        {
          hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, false, true>), g, blk, lds_bytes, stream, problem, cfg, buf);
        }
        else
        {
          hipLaunchKernelGGL((ddp_solve_tpi2w_kernel<Problem, false, false>), g, blk, lds_bytes, stream, problem, cfg, buf);
        }
        return hipGetLastError();
      }
    }
    if(own)
    {
      return hipErrorNotSupported; // the single-wavefront kernel has no per-instance-problem instantiation
    }
    if(con)
    {
      hipLaunchKernelGGL((ddp_solve_tpi_kernel<Problem, true>), dim3(grid), dim3(kLanesPerBlock), 0, stream, problem, cfg,
                         buf);
    }
    else
    {
      hipLaunchKernelGGL((ddp_solve_tpi_kernel<Problem, false>), dim3(grid), dim3(kLanesPerBlock), 0, stream, problem,
                         cfg, buf);
    }
    return hipGetLastError();
  }
  static hipError_t launchMpcAdvance(const void * params,
                                     const DeviceBuffers & buf,
                                     const MpcAdvanceArgs & args,
                                     hipStream_t stream)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    hipLaunchKernelGGL((mpc_advance_kernel<Problem>), dim3(buf.Bp / kLanesPerBlock), dim3(kLanesPerBlock), 0, stream,
                       problem, buf, args);
    return hipGetLastError();
  }
  static void inputDims(const void * params, double t0, int T, int * out)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    for(int i = 0; i < T; i++)
    {
      if constexpr(Problem::kDynamicInput)
      {
        out[i] = problem.inputDim(t0 + i * problem.dt());
      }
      else
      {
        out[i] = Problem::kInputDimMax;
      }
    }
  }
  static double dt(const void * params)
  {
    Problem problem;
    std::memcpy(static_cast<void *>(&problem), params, sizeof(Problem));
    return problem.dt();
  }
  static ModelOps make()
  {
    static_assert(std::is_trivially_copyable<Problem>::value,
                  "a DDP problem must be trivially copyable (no std::function / heap members): it is passed to "
                  "the GPU by value");
    static_assert(std::is_default_constructible<Problem>::value, "a DDP problem must be default constructible");
    ModelOps ops;
    ops.name = Problem::kName;
    ops.state_dim = Problem::kStateDim;
    ops.input_dim_max = Problem::kInputDimMax;
    ops.dynamic_input = Problem::kDynamicInput ? 1 : 0;
    ops.param_bytes = sizeof(Problem);
    ops.default_params = &defaultParams;
    ops.launch_solve = &launchSolve;
    ops.input_dims = &inputDims;
    ops.dt = &dt;
    ops.kernel_name = &kernelName;
    ops.launch_mpc_advance = &launchMpcAdvance;
    ops.has_plant_step = HasPlantStep<Problem>::value ? 1 : 0;
    ops.wpi_workspace_doubles = &wpiWorkspaceDoubles;
    ops.scalar_bytes = static_cast<int>(sizeof(typename Problem::Scalar));
    ops.gain_layout = 0;
    ops.gain_layout_of = &gainLayoutOf;
    ops.own_problems_supported = &ownProblemsSupported;
    ops.resumable_supported = &resumableSupported;
    static_assert(sizeof(typename Problem::Scalar) == 8, "these kernel families compute in double; fp32 problem types register "
                                                         "through ddp_kernels_tile32.hpp");
    return ops;
  }
};
} // namespace hip
} // namespace nmpc_amd

#define NMPC_AMD_REGISTER_PROBLEM(ProblemType) NMPC_AMD_REGISTER_PROBLEM_WITH(ProblemType, nmpc_amd::hip::ModelOpsFor<ProblemType>)
