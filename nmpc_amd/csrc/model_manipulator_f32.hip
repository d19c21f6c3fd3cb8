// The manipulator problem (n 14, m 7) in fp32: a shape the fp32 tile kernel (ddp_kernels_tile32.hpp: m <= 4, n in {4, 8, 12}) does
// not take — served by the fp64 tile kernel's float instantiation (ddp_kernels_tile64.hpp).  Registered as "manipulator_f32".
#include <nmpc_amd/hip/ddp_kernels_tile32.hpp>

#include <nmpc_amd/models/Manipulator.hpp>

using nmpc_amd::DDPProblemManipulatorF32;

NMPC_AMD_REGISTER_PROBLEM_TILE64_FLOAT(DDPProblemManipulatorF32);
