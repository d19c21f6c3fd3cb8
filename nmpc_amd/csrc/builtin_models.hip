// Problem types compiled into libnmpc_hip_ddp.so: the four models the reference's tests define
// (nmpc_ddp/tests/src/TestDDP*.cpp) restated against the nmpc_amd::DDPProblem functor API.
// Each registration instantiates the solve kernel for that type (gfx950 code object).
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/Bipedal.hpp>
#include <nmpc_amd/models/CartPole.hpp>
#include <nmpc_amd/models/CentroidalMotion.hpp>
#include <nmpc_amd/models/VerticalMotion.hpp>

using nmpc_amd::DDPProblemBipedal;
using nmpc_amd::DDPProblemCartPole;
using nmpc_amd::DDPProblemCentroidalMotion;
using nmpc_amd::DDPProblemVerticalMotion;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemCartPole);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemBipedal);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemVerticalMotion);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemCentroidalMotion);
