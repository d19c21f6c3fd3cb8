// Problem types for BASELINE.json configs 4 and 5 (no reference model exists for them).
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/Manipulator.hpp>
#include <nmpc_amd/models/Quadrotor.hpp>

using nmpc_amd::DDPProblemManipulator;
using nmpc_amd::DDPProblemQuadrotor;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemQuadrotor);
NMPC_AMD_REGISTER_PROBLEM(DDPProblemManipulator);
