// Problem type for BASELINE.json config 4 (no reference model exists for it).
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/Quadrotor.hpp>

using nmpc_amd::DDPProblemQuadrotor;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemQuadrotor);
