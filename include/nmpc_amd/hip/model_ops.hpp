// Type-erased description of one registered problem type: what the C-ABI (nmpc_amd/csrc/capi.hip) knows about a problem
// class compiled into gfx950 code.  Kernel families fill it: model_registry.hpp (fp64 lane / two-wave / quad / wave-per-
// instance kernels), ddp_kernels_tile32.hpp (fp32 tile kernel).
#pragma once

#include <cstddef>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include <nmpc_amd/hip/ddp_kernels.hpp>
#include <nmpc_amd/hip/mpc_args.hpp>

namespace nmpc_amd
{
namespace hip
{
/** What picks a kernel family and its launch schedule besides the problem's shape, the batch size and the Configuration — per
    HANDLE, fixed when the handle is created or through the C-ABI (nmpc_hip_ddp_set_kernel, nmpc_hip_ddp_set_dispatch_batch), never
    read from the environment on the launch path.  The environment variables (NMPC_HIP_DDP_KERNEL, NMPC_HIP_DDP_TILE64_GROUP / _CHUNK /
    _PAIR / _ADOPT / _WIDE, NMPC_HIP_DDP_FAN_SCRATCH, NMPC_HIP_DDP_FAN_AUTO) are developer overrides for A/B measurements and tests:
    fromEnvironment() reads them ONCE, when a handle is created. */
struct LaunchKnobs
{
  char kernel[16] = ""; //!< "" (automatic), "1w", "2w", "quad", "wpi", "tile64", "tile32"
  int tile64_group = 0; //!< > 0: at most this many instances per group of the tile kernel
  int tile64_chunk = 0; //!< > 0: at most this many timesteps per pass of its model code
  int tile64_pair = 1, tile64_adopt = 1, tile64_wide = 1; //!< 0: that part of its line-search schedule off (A/B)
  int fan_scratch = 1; //!< 0: the quad kernel's fan-out scratch is not allocated
  int fan_auto = 0; //!< valid if has_fan_auto: ModelOpsFor::fanOutAutoMaxIter()
  int has_fan_auto = 0;
  int have_workspace = 1; //!< the handle's per-instance workspace was allocated (0: the kernels that need it are not chosen)
  //! > 0: the batch size the kernel family is chosen FOR — a shard of a larger solve takes the family the whole batch would get,
  //! so that its results are the unsharded solve's bit for bit (families differ in the last bits; DDPSolverSharded, bench.py)
  int dispatch_batch = 0;

  bool kernelIs(const char * name) const
  {
    return std::strcmp(kernel, name) == 0;
  }
  int batchFor(int batch) const
  {
    return dispatch_batch > 0 ? dispatch_batch : batch;
  }
  static LaunchKnobs fromEnvironment()
  {
    LaunchKnobs k;
    if(const char * e = std::getenv("NMPC_HIP_DDP_KERNEL"))
    {
      std::strncpy(k.kernel, e, sizeof(k.kernel) - 1);
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_TILE64_GROUP"))
    {
      k.tile64_group = std::atoi(e) & 0xffff;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_TILE64_CHUNK"))
    {
      k.tile64_chunk = std::atoi(e) & 0x1fff;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_TILE64_PAIR"))
    {
      k.tile64_pair = std::atoi(e) != 0;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_TILE64_ADOPT"))
    {
      k.tile64_adopt = std::atoi(e) != 0;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_TILE64_WIDE"))
    {
      k.tile64_wide = std::atoi(e) != 0;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_FAN_SCRATCH"))
    {
      k.fan_scratch = std::strcmp(e, "0") != 0;
    }
    if(const char * e = std::getenv("NMPC_HIP_DDP_FAN_AUTO"))
    {
      k.fan_auto = std::atoi(e);
      k.has_fan_auto = 1;
    }
    return k;
  }
};
/** The knobs of the handle whose operation is running on this thread (set by the C-ABI entry points around every ModelOps call:
    ScopedKnobs); outside of one — the registry's own queries — the environment as it is now. */
inline thread_local const LaunchKnobs * g_launch_knobs = nullptr;
inline LaunchKnobs launchKnobs()
{
  return g_launch_knobs ? *g_launch_knobs : LaunchKnobs::fromEnvironment();
}
struct ScopedKnobs
{
  const LaunchKnobs * saved;
  explicit ScopedKnobs(const LaunchKnobs * k) : saved(g_launch_knobs)
  {
    g_launch_knobs = k;
  }
  ~ScopedKnobs()
  {
    g_launch_knobs = saved;
  }
  ScopedKnobs(const ScopedKnobs &) = delete;
  ScopedKnobs & operator=(const ScopedKnobs &) = delete;
};

/** Type-erased operations of one registered problem type. */
struct ModelOps
{
  const char * name;
  int state_dim;
  int input_dim_max;
  int dynamic_input;
  size_t param_bytes;
  //! placement-constructs a default problem object into out
  void (*default_params)(void * out);
  //! launches the solve kernel; params points to a host copy of the problem object
  hipError_t (*launch_solve)(const void * params,
                             const nmpc_hip_ddp_config & cfg,
                             const DeviceBuffers & buf,
                             hipStream_t stream);
  //! host-side inputDim(t0 + i dt) for i < T (validation of initial_u_list, DDPSolver.hpp:46-58)
  void (*input_dims)(const void * params, double t0, int T, int * out);
  //! dt() of the problem object
  double (*dt)(const void * params);
  //! name of the kernel launch_solve launches for a batch of `batch` instances under the Configuration `cfg` (lane mapping, see
  //! launchSolve: with / without input constraints; the fp32 types also look at cost_update_thre), the handle's LaunchKnobs included
  const char * (*kernel_name)(int batch, const nmpc_hip_ddp_config & cfg);
  //! launches the receding-horizon advance step (mpc_kernels.hpp) between two solves
  hipError_t (*launch_mpc_advance)(const void * params,
                                   const DeviceBuffers & buf,
                                   const MpcAdvanceArgs & args,
                                   hipStream_t stream);
  //! 1 if the problem has the plant step stateEq(t, x, u, dt) the plant pattern integrates with
  int has_plant_step;
  //! elements (of the problem's scalar type) of per-instance workspace the model's kernel needs for horizon T (0: none)
  size_t (*wpi_workspace_doubles)(int T);
  //! sizeof(Problem::Scalar): 8 for the reference's arithmetic, 4 for the fp32 problem types (every Scalar device array
  //! of the handle has this element size; the C-ABI exchanges doubles either way)
  int scalar_bytes;
  //! 0: k_list_ / K_list_ live in the handle's tile-major kff / Kfb arrays; 1: in the workspace, instance-major records
  //! [B][T][MM + MM * N] (k_i, then K_i column-major) as the fp32 tile kernel writes them
  int gain_layout;
  //! the same per launch, for problem types whose kernel families differ in it (nullptr: gain_layout): what a solve of `batch`
  //! instances with / without input constraints leaves behind
  int (*gain_layout_of)(int batch, int constrained) = nullptr;
  //! 1 if the kernel launch_solve picks for such a batch has an instantiation with one problem object per instance
  //! (nmpc_hip_ddp_set_model_params_batch is refused otherwise — at set time, not at the first solve); nullptr: it has
  int (*own_problems_supported)(int batch, int constrained) = nullptr;
  //! 1 if the kernel launch_solve picks for such a solve has a RESUMABLE instantiation (DeviceBuffers::iter_end > 0: a launch runs
  //! iterations iter_begin .. iter_end of the dense prefix and parks the solver state): what the ragged-convergence schedule of
  //! capi.hip needs; nullptr / 0: whole solves only
  int (*resumable_supported)(int batch, const nmpc_hip_ddp_config & cfg, int own_problems) = nullptr;
};

} // namespace hip
} // namespace nmpc_amd

extern "C" int nmpc_hip_ddp_register_model(const nmpc_amd::hip::ModelOps * ops);

/** Registers the problem type under ProblemType::kName with the operations `OpsMaker::make()` returns. */
#define NMPC_AMD_REGISTER_PROBLEM_WITH(ProblemType, OpsMaker)                                         \
  namespace                                                                                           \
  {                                                                                                   \
  struct ProblemType##Registrar                                                                       \
  {                                                                                                   \
    ProblemType##Registrar()                                                                          \
    {                                                                                                 \
      static const nmpc_amd::hip::ModelOps ops = OpsMaker::make();                                    \
      nmpc_hip_ddp_register_model(&ops);                                                              \
    }                                                                                                 \
  } g_##ProblemType##_registrar;                                                                      \
  }
