"""Host-side handles of the problem types compiled into libnmpc_hip_ddp.so.

Each class mirrors the memory image of the C++ problem object (include/nmpc_amd/models/*.hpp) as a ctypes
structure of doubles, so that a Python caller can build / modify a problem exactly like a C++ caller mutates
`ddp_problem->param_` / `cost_weight_` in the reference's tests (TestDDPCartPole.cpp:274-289).  No math lives
here: the functors run on the GPU.
"""
from __future__ import annotations

import ctypes as C

from . import _capi


class _Problem:
    """Base: `name` is the registry name, `blob` the ctypes structure sent through set_model_params."""

    name: str = ""
    _Blob = None

    def __init__(self, **kw):
        L = _capi.load()
        self.blob = self._Blob()
        _capi.check(L.nmpc_hip_ddp_model_default_params(self.name.encode(), C.byref(self.blob),
                                                        C.sizeof(self.blob)))
        for k, v in kw.items():
            self.set(k, v)

    def set(self, key, value):
        cur = getattr(self.blob, key)
        if hasattr(cur, "__len__"):
            if len(value) != len(cur):
                raise ValueError(f"{key} expects {len(cur)} values")
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(self.blob, key, value)
        return self

    def get(self, key):
        cur = getattr(self.blob, key)
        return list(cur) if hasattr(cur, "__len__") else cur

    def dt(self) -> float:
        return self.blob.dt

    @classmethod
    def scalar_bytes(cls) -> int:
        """8: the problem computes in double (the reference's arithmetic); 4: in float."""
        out = C.c_int()
        _capi.check(_capi.load().nmpc_hip_ddp_model_scalar_bytes(cls.name.encode(), C.byref(out)))
        return out.value

    @classmethod
    def dims(cls):
        n, m, dyn, nbytes = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        _capi.check(_capi.load().nmpc_hip_ddp_model_info(cls.name.encode(), C.byref(n), C.byref(m), C.byref(dyn),
                                                         C.byref(nbytes)))
        return n.value, m.value, bool(dyn.value), nbytes.value


class DDPProblemCartPole(_Problem):
    """include/nmpc_amd/models/CartPole.hpp (reference: TestDDPCartPole.cpp:28-234)."""

    name = "cartpole"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("cart_mass", C.c_double), ("pole_mass", C.c_double),
                    ("pole_length", C.c_double), ("running_x", C.c_double * 4), ("running_u", C.c_double * 1),
                    ("terminal_x", C.c_double * 4), ("ref_pos", C.c_double)]


class DDPProblemCartPoleF32(_Problem):
    """The same problem type instantiated in float (DDPProblemCartPoleT<float>), served by the fp32 tile kernel
    (include/nmpc_amd/hip/ddp_kernels_tile32.hpp) at its n = 4, m = 1 shape."""

    name = "cartpole_f32"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_float), ("cart_mass", C.c_float), ("pole_mass", C.c_float),
                    ("pole_length", C.c_float), ("running_x", C.c_float * 4), ("running_u", C.c_float * 1),
                    ("terminal_x", C.c_float * 4), ("ref_pos", C.c_float)]


class DDPProblemBipedal(_Problem):
    """include/nmpc_amd/models/Bipedal.hpp (reference: TestDDPBipedal.cpp:16-144, :171-225)."""

    name = "bipedal"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("running_vel", C.c_double), ("running_zmp", C.c_double),
                    ("terminal_pos", C.c_double), ("terminal_vel", C.c_double), ("end_t", C.c_double)]


class DDPProblemVerticalMotion(_Problem):
    """include/nmpc_amd/models/VerticalMotion.hpp (reference: TestDDPVerticalMotion.cpp:31-234)."""

    name = "vertical"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("running_x", C.c_double * 2), ("running_u", C.c_double),
                    ("terminal_x", C.c_double * 2), ("mass", C.c_double), ("ref_switch_t", C.c_double)]


class DDPProblemCentroidalMotion(_Problem):
    """include/nmpc_amd/models/CentroidalMotion.hpp (reference: TestDDPCentroidalMotion.cpp:24-281)."""

    name = "centroidal"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("running_u", C.c_double), ("mass", C.c_double),
                    ("flight_start_t", C.c_double), ("flight_end_t", C.c_double), ("ref_switch_t", C.c_double),
                    ("weight_pos_ang", C.c_double), ("weight_lin", C.c_double), ("second_rect", C.c_double * 4)]


class DDPProblemQuadrotor(_Problem):
    """include/nmpc_amd/models/Quadrotor.hpp (builder-defined; BASELINE.json config 4)."""

    name = "quadrotor"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("inertia", C.c_double * 3), ("arm", C.c_double),
                    ("yaw_coef", C.c_double), ("w_pos", C.c_double), ("w_rpy", C.c_double), ("w_vel", C.c_double),
                    ("w_omega", C.c_double), ("w_u", C.c_double), ("wt_scale", C.c_double),
                    ("ref_pos", C.c_double * 3)]


class DDPProblemQuadrotorF32(_Problem):
    """The same problem type instantiated in float (DDPProblemQuadrotorT<float>): BASELINE.json config 4 as specified
    ("fp32"), served by the fp32 tile kernel (include/nmpc_amd/hip/ddp_kernels_tile32.hpp) or, chosen per launch, by the
    tile kernel's float instantiation (ddp_kernels_tile64.hpp; ModelOpsTile32::useTile64Float, kernelName() tells)."""

    name = "quadrotor_f32"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_float), ("mass", C.c_float), ("inertia", C.c_float * 3), ("arm", C.c_float),
                    ("yaw_coef", C.c_float), ("w_pos", C.c_float), ("w_rpy", C.c_float), ("w_vel", C.c_float),
                    ("w_omega", C.c_float), ("w_u", C.c_float), ("wt_scale", C.c_float),
                    ("ref_pos", C.c_float * 3)]


class DDPProblemManipulator(_Problem):
    """include/nmpc_amd/models/Manipulator.hpp (builder-defined; BASELINE.json config 5)."""

    name = "manipulator"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("w_diag", C.c_double), ("w_off", C.c_double), ("damping", C.c_double),
                    ("grav_scale", C.c_double), ("wq", C.c_double), ("wv", C.c_double), ("wu", C.c_double),
                    ("wt_scale", C.c_double), ("q_ref_scale", C.c_double)]


class DDPProblemManipulatorF32(_Problem):
    """The manipulator in float (DDPProblemManipulatorT<float>): an fp32 shape with seven inputs — the fp64 tile kernel's float
    instantiation (the fp32 tile kernel takes m <= 4, n in {4, 8, 12})."""

    name = "manipulator_f32"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_float), ("w_diag", C.c_float), ("w_off", C.c_float), ("damping", C.c_float),
                    ("grav_scale", C.c_float), ("wq", C.c_float), ("wv", C.c_float), ("wu", C.c_float),
                    ("wt_scale", C.c_float), ("q_ref_scale", C.c_float)]


class DDPProblemPlanarVtol(_Problem):
    """include/nmpc_amd/models/PlanarVtol.hpp (builder-defined n = 6, m = 2: the 5 <= n <= 8 shapes on the fp64 tile kernel)."""

    name = "planar_vtol"

    class _Blob(C.Structure):
        _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("inertia", C.c_double), ("arm", C.c_double),
                    ("w_pos", C.c_double), ("w_ang", C.c_double), ("w_vel", C.c_double), ("w_omega", C.c_double),
                    ("w_u", C.c_double), ("wt_scale", C.c_double), ("ref_pos", C.c_double * 2)]


PROBLEMS = {c.name: c for c in (DDPProblemCartPole, DDPProblemCartPoleF32, DDPProblemBipedal, DDPProblemVerticalMotion,
                                DDPProblemCentroidalMotion, DDPProblemQuadrotor, DDPProblemQuadrotorF32,
                                DDPProblemManipulator, DDPProblemManipulatorF32, DDPProblemPlanarVtol)}


def make_problem(name: str, **kw) -> _Problem:
    return PROBLEMS[name](**kw)
