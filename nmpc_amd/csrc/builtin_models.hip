// Problem types compiled into libnmpc_hip_ddp.so: the small models the reference's tests define
// (nmpc_ddp/tests/src/TestDDPCartPole.cpp, TestDDPBipedal.cpp, TestDDPVerticalMotion.cpp) restated against the
// nmpc_amd::DDPProblem functor API.  Each registration instantiates the solve kernels for that type (gfx950 code
// objects).  The large models have one translation unit each (model_*.hip) so that they compile in parallel.
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/Bipedal.hpp>
#include <nmpc_amd/models/CartPole.hpp>
#include <nmpc_amd/models/VerticalMotion.hpp>

using nmpc_amd::DDPProblemBipedal;
using nmpc_amd::DDPProblemCartPole;
using nmpc_amd::DDPProblemVerticalMotion;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemCartPole);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemBipedal);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemVerticalMotion);
