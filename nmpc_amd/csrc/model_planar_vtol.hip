// Builder-defined n = 6, m = 2 problem type: the 5 <= n <= 8 shapes on the fp64 tile kernel.
#include <nmpc_amd/hip/model_registry.hpp>

#include <nmpc_amd/models/PlanarVtol.hpp>

using nmpc_amd::DDPProblemPlanarVtol;

NMPC_AMD_REGISTER_PROBLEM(DDPProblemPlanarVtol);
