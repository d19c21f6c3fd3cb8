// The cart-pole problem in fp32, served by the fp32 tile kernel at its n = 4, m = 1 shape
// (include/nmpc_amd/hip/ddp_kernels_tile32.hpp).  Registered as "cartpole_f32"; "cartpole" is the same problem in the
// reference's arithmetic (double) on the quad / two-wave kernels (builtin_models.hip).
#include <nmpc_amd/hip/ddp_kernels_tile32.hpp>

#include <nmpc_amd/models/CartPole.hpp>

using nmpc_amd::DDPProblemCartPoleF32;

NMPC_AMD_REGISTER_PROBLEM_TILE32(DDPProblemCartPoleF32);
