// Problem type for BASELINE.json config 5 (no reference model exists for it).
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/Manipulator.hpp>

using nmpc_amd::DDPProblemManipulator;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemManipulator);
