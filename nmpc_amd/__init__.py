"""nmpc_amd — MI355X-native batched DDP solver behind the nmpc_ddp problem / solver API.

Host-side mirror of the reference interface (`DDPSolverBatch`, `Configuration`, problem handles); all numerics
run in libnmpc_hip_ddp.so (hand-written HIP for gfx950) through the C-ABI of include/nmpc_hip_ddp.h.
"""
from .ddp import (ComputationDuration, Configuration, ControlData, DDPSolverBatch, DDPSolverPool, MpcLog,  # noqa: F401
                  StreamResult, TraceData, request_hw_queues)
from .models import (DDPProblemBipedal, DDPProblemCartPole, DDPProblemCartPoleF32, DDPProblemCentroidalMotion,  # noqa: F401
                     DDPProblemManipulator, DDPProblemManipulatorF32, DDPProblemQuadrotor, DDPProblemVerticalMotion, make_problem)
