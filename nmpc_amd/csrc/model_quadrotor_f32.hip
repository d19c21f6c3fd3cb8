// BASELINE.json config 4 as specified: the quadrotor problem in fp32, served by the fp32 tile kernel
// (include/nmpc_amd/hip/ddp_kernels_tile32.hpp).  Registered as "quadrotor_f32"; "quadrotor" is the same problem in the
// reference's arithmetic (double) on the wave-per-instance kernel.
#include <nmpc_amd/hip/ddp_kernels_tile32.hpp>

#include <nmpc_amd/models/Quadrotor.hpp>

using nmpc_amd::DDPProblemQuadrotorF32;

NMPC_AMD_REGISTER_PROBLEM_TILE32(DDPProblemQuadrotorF32);
