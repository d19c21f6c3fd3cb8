"""nmpc_amd — MI355X-native batched DDP solver behind the nmpc_ddp problem / solver API.

Host-side mirror of the reference interface (`DDPSolverBatch`, `Configuration`, problem handles); all numerics
run in libnmpc_hip_ddp.so (hand-written HIP for gfx950) through the C-ABI of include/nmpc_hip_ddp.h.
"""
import os as _os

# (see nmpc_amd/csrc/capi.hip: streams share GPU_MAX_HW_QUEUES hardware queues, default 4; a pool of handles overlaps as many
# batches as there are queues.  Has to be in the environment before the HIP runtime initialises — import nmpc_amd before torch
# touches the device, or set it yourself.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .ddp import (ComputationDuration, Configuration, ControlData, DDPSolverBatch, DDPSolverPool, MpcLog,  # noqa: F401
                  TraceData)
from .models import (DDPProblemBipedal, DDPProblemCartPole, DDPProblemCartPoleF32, DDPProblemCentroidalMotion,  # noqa: F401
                     DDPProblemManipulator, DDPProblemManipulatorF32, DDPProblemQuadrotor, DDPProblemVerticalMotion, make_problem)
