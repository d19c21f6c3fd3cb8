// Centroidal-motion problem of the reference's tests (nmpc_ddp/tests/src/TestDDPCentroidalMotion.cpp), n = 9,
// input dimension 16 / 0 along the horizon.
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/CentroidalMotion.hpp>

using nmpc_amd::DDPProblemCentroidalMotion;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemCentroidalMotion);
