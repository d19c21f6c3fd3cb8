// Type-erased description of one registered FMPC problem type: what the C-ABI (nmpc_amd/csrc/fmpc_capi.hip) knows about a
// problem class compiled into gfx950 code (the FMPC counterpart of model_ops.hpp).
#pragma once

#include <cstddef>
#include <cstdlib>
#include <new>
#include <type_traits>

#include <hip/hip_runtime.h>

#include <nmpc_amd/hip/fmpc_kernels.hpp>

namespace nmpc_amd
{
namespace hip
{
/** Which Riccati kernel a batch of B instances of an (N states, M inputs) problem type runs: fmpc_riccati_quad_kernel (sixteen
    lanes per instance on the fp64 matrix cores) for N <= 4, M = 1 while its 16-instance workgroups are at most two per CU — the
    regime in which the chip is otherwise empty; fmpc_riccati_kernel (one lane per instance) otherwise: it issues half as many
    instructions per instance and wins once the batch fills the chip by itself (measured on MI355X, cart-pole, T = 200, quad /
    lane ms per 5-iteration solve: 1.5 / 2.8 at 2048 instances, 2.2 / 3.6 at 4096, 4.5 / 4.7 at 8192, 8.9 / 7.2 at 16384,
    37.8 / 29.0 at 65536).  The environment variable NMPC_HIP_FMPC_RICCATI=quad|lane forces one (A/B measurements, tests). */
/** \param force FmpcBuffers::riccati_force: 0 automatic, 1 the matrix-core kernel, 2 the lane kernel (a handle's setting, taken
    from NMPC_HIP_FMPC_RICCATI = quad / lane once, when the handle is created: fmpcRiccatiForceFromEnvironment) */
inline int fmpcRiccatiForceFromEnvironment()
{
  const char * force = getenv("NMPC_HIP_FMPC_RICCATI");
  return (force && force[0] == 'q') ? 1 : ((force && force[0] == 'l') ? 2 : ((force && force[0] == 'f') ? 3 : 0));
}
inline int fmpcComputeUnits()
{
  static int n_cu = 0; // of the current device at first use (the handles of one process sit on like devices)
  if(n_cu == 0)
  {
    int device = 0;
    if(hipGetDevice(&device) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess
       || n_cu <= 0)
    {
      n_cu = 256;
    }
  }
  return n_cu;
}
inline bool fmpcUseQuadRiccati(int N, int M, int B, int force)
{
  if(!(N <= 4 && M == 1))
  {
    return false;
  }
  if(force != 0)
  {
    return force == 1 || force == 3;
  }
  return (B + 15) / 16 <= 2 * fmpcComputeUnits();
}
/** The fused kernel (fmpc_riccati_fused_kernel: coefficient records computed by producer waves into the staging LDS, never in HBM)
    where the matrix-core kernel runs with at most ONE workgroup per CU — its three staging slots are 93 KB of LDS, the unfused kernel's
    61 KB let two workgroups share a CU — unless the handle pins one (riccati_force 1 / 3: NMPC_HIP_FMPC_RICCATI = quad / fused). */
inline bool fmpcUseFusedRiccati(int B, int force)
{
  if(force != 0)
  {
    return force == 3;
  }
  return (B + 15) / 16 <= fmpcComputeUnits();
}

struct FmpcOps
{
  const char * name;
  int state_dim;
  int input_dim;
  int ineq_dim;
  size_t param_bytes;
  int coef_stride; //!< doubles per timestep of FmpcBuffers::coef
  int gain_stride; //!< doubles per timestep of FmpcBuffers::gain
  int gain_offset_k, gain_offset_K, gain_offset_s, gain_offset_P;
  //! placement-constructs a default problem object into out
  void (*default_params)(void * out);
  //! dt() of a problem object
  double (*dt)(const void * params);
  hipError_t (*launch_init_complementary)(const FmpcBuffers & buf, hipStream_t stream);
  hipError_t (*launch_coeff)(const FmpcBuffers & buf, hipStream_t stream);
  hipError_t (*launch_riccati)(const FmpcBuffers & buf, int iter, hipStream_t stream);
  hipError_t (*launch_delta)(const FmpcBuffers & buf, hipStream_t stream);
  //! fmpc_tail_kernel: step length + update of iteration iter and the head of iteration iter + 1 (barrier parameter, KKT-error terms,
  //! terminal record).  \return hipErrorNotSupported where the sequence does not apply (tail_applies)
  hipError_t (*launch_tail)(const FmpcBuffers & buf, int iter, int last, hipStream_t stream);
  //! whether an iteration of this handle ends in fmpc_tail_kernel: the fused Riccati kernel is the one launched (its producer waves
  //! take over the records' NaN verdict), no line search between step length and update, not switched off (FmpcBuffers::fuse_tail)
  bool (*tail_applies)(const FmpcBuffers & buf);
  hipError_t (*launch_line_search)(const FmpcBuffers & buf, int iter, hipStream_t stream);
  hipError_t (*launch_plant)(const FmpcBuffers & buf,
                             double * x_plant,
                             double * t_plant,
                             double sim_dt,
                             int substeps,
                             int use_feedback,
                             hipStream_t stream);
};

template<class Problem>
struct FmpcOpsOf
{
  static constexpr int N = Problem::kStateDim, M = Problem::kInputDimMax, G = Problem::kIneqDim;
  static_assert(std::is_trivially_copyable<Problem>::value, "[FMPC] a problem object must be trivially copyable");
  static_assert(!Problem::kDynamicInput, "[FMPC] dynamic input dimensions are not offered");

  static unsigned blocks(size_t threads, unsigned block)
  {
    return static_cast<unsigned>((threads + block - 1) / block);
  }

  static FmpcOps make()
  {
    using GL = fmpc::GainLayout<N, M>;
    FmpcOps o{};
    o.name = Problem::kName;
    o.state_dim = N;
    o.input_dim = M;
    o.ineq_dim = G;
    o.param_bytes = sizeof(Problem);
    o.coef_stride = fmpc::CoefLayout<N, M>::kStride;
    o.gain_stride = GL::kStride;
    o.gain_offset_k = GL::k;
    o.gain_offset_K = GL::K;
    o.gain_offset_s = GL::S;
    o.gain_offset_P = GL::P;
    o.default_params = [](void * out) { new(out) Problem(); };
    o.dt = [](const void * params) { return static_cast<const Problem *>(params)->dt(); };
    o.launch_init_complementary = [](const FmpcBuffers & buf, hipStream_t stream) {
      hipLaunchKernelGGL(fmpc_init_complementary_kernel<Problem>, dim3(blocks(static_cast<size_t>(buf.B) * buf.T, 256)), dim3(256),
                         0, stream, buf);
      return hipGetLastError();
    };
    o.launch_coeff = [](const FmpcBuffers & buf, hipStream_t stream) {
      if constexpr(N <= 4 && M == 1)
      {
        if(fmpcUseQuadRiccati(N, M, buf.B, buf.riccati_force) && fmpcUseFusedRiccati(buf.B, buf.riccati_force))
        {
          // (the records themselves are computed by the Riccati kernel's producer wave)
          hipLaunchKernelGGL((fmpc_coeff_kernel<Problem, false>), dim3(blocks(static_cast<size_t>(buf.B) * (buf.T + 1), 256)), dim3(256),
                             0, stream, buf);
          return hipGetLastError();
        }
      }
      hipLaunchKernelGGL((fmpc_coeff_kernel<Problem, true>), dim3(blocks(static_cast<size_t>(buf.B) * (buf.T + 1), 256)), dim3(256), 0,
                         stream, buf);
      return hipGetLastError();
    };
    o.launch_riccati = [](const FmpcBuffers & buf, int iter, hipStream_t stream) {
      if constexpr(N <= 4 && M == 1)
      {
        if(fmpcUseQuadRiccati(N, M, buf.B, buf.riccati_force))
        {
          if(fmpcUseFusedRiccati(buf.B, buf.riccati_force))
          {
            hipLaunchKernelGGL((fmpc_riccati_fused_kernel<Problem>), dim3(blocks(buf.B, 16)), dim3(384), 0, stream, buf, iter);
          }
          else
          {
            hipLaunchKernelGGL((fmpc_riccati_quad_kernel<N>), dim3(blocks(buf.B, 16)), dim3(256), 0, stream, buf, iter);
          }
          return hipGetLastError();
        }
      }
      hipLaunchKernelGGL((fmpc_riccati_kernel<N, M>), dim3(blocks(buf.B, 64)), dim3(64), 0, stream, buf, iter);
      return hipGetLastError();
    };
    o.launch_delta = [](const FmpcBuffers & buf, hipStream_t stream) {
      hipLaunchKernelGGL(fmpc_delta_kernel<Problem>, dim3(blocks(static_cast<size_t>(buf.B) * (buf.T + 1), 256)), dim3(256), 0,
                         stream, buf);
      return hipGetLastError();
    };
    o.tail_applies = [](const FmpcBuffers & buf) {
      if constexpr(N <= 4 && M == 1)
      {
        return buf.fuse_tail != 0 && !buf.enable_line_search && fmpc::tailFits(buf) && fmpcUseQuadRiccati(N, M, buf.B, buf.riccati_force)
               && fmpcUseFusedRiccati(buf.B, buf.riccati_force);
      }
      return false;
    };
    o.launch_tail = [](const FmpcBuffers & buf, int iter, int last, hipStream_t stream) {
      if constexpr(N <= 4 && M == 1)
      {
        const unsigned dot_bytes = fmpc::tailDotBytes(buf.T);
        hipLaunchKernelGGL(fmpc_tail_kernel<Problem>, dim3(blocks(buf.B, 16)), dim3(16 * fmpc::tailSlices(buf.T)), dot_bytes, stream, buf,
                           iter, last, dot_bytes != 0 ? 1 : 0);
        return hipGetLastError();
      }
      return hipErrorNotSupported;
    };
    o.launch_line_search = [](const FmpcBuffers & buf, int iter, hipStream_t stream) {
      hipLaunchKernelGGL(fmpc_line_search_kernel<Problem>, dim3(blocks(buf.B, 64)), dim3(64), 0, stream, buf, iter);
      return hipGetLastError();
    };
    o.launch_plant = [](const FmpcBuffers & buf, double * x_plant, double * t_plant, double sim_dt, int substeps, int use_feedback,
                        hipStream_t stream) {
      hipLaunchKernelGGL(fmpc_plant_kernel<Problem>, dim3(blocks(buf.B, 64)), dim3(64), 0, stream, buf, x_plant, t_plant, sim_dt,
                         substeps, use_feedback);
      return hipGetLastError();
    };
    return o;
  }
};
} // namespace hip
} // namespace nmpc_amd

extern "C" int nmpc_hip_fmpc_register_model(const nmpc_amd::hip::FmpcOps * ops);

/** Registers the FMPC problem type under ProblemType::kName. */
#define NMPC_AMD_REGISTER_FMPC_PROBLEM(ProblemType)                                    \
  namespace                                                                            \
  {                                                                                    \
  struct ProblemType##FmpcRegistrar                                                    \
  {                                                                                    \
    ProblemType##FmpcRegistrar()                                                       \
    {                                                                                  \
      static const nmpc_amd::hip::FmpcOps ops = nmpc_amd::hip::FmpcOpsOf<ProblemType>::make(); \
      nmpc_hip_fmpc_register_model(&ops);                                              \
    }                                                                                  \
  } g_##ProblemType##_fmpc_registrar;                                                  \
  }
