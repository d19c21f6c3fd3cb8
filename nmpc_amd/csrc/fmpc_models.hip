// gfx950 code objects of the FMPC problem types shipped with the library: the two problems of the reference's FMPC tests
// (nmpc_fmpc/tests/src/TestFmpcOscillator.cpp:18-135, TestFmpcCartPole.cpp:32-267) and a two-input point mass.
#include <nmpc_amd/hip/fmpc_ops.hpp>
#include <nmpc_amd/models/FmpcCartPole.hpp>
#include <nmpc_amd/models/FmpcOscillator.hpp>
#include <nmpc_amd/models/FmpcPointMass.hpp>

using nmpc_amd::FmpcProblemCartPole;
using nmpc_amd::FmpcProblemOscillator;
using nmpc_amd::FmpcProblemPointMass;

NMPC_AMD_REGISTER_FMPC_PROBLEM(FmpcProblemOscillator)
NMPC_AMD_REGISTER_FMPC_PROBLEM(FmpcProblemCartPole)
NMPC_AMD_REGISTER_FMPC_PROBLEM(FmpcProblemPointMass)
