"""rdhip -- host-side mirror of RobotDynamics.jl's batched dynamics / Jacobian operators over librdhip (HIP, gfx950).

    import rdhip as RD
    model  = RD.Satellite()                                   # RigidBody{QuatRotation}
    dmodel = RD.DiscretizedDynamics(model, RD.RK4)            # DiscretizedDynamics{RK4}(model)
    Z = RD.PackedTrajectory(13, 6, data, t, dt)               # dense Vector{KnotPoint}
    J = RD.alloc_jacobians(dmodel, len(Z), data); Y = ...
    RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dmodel, J, Y, Z)   # jacobian!(sig, diff, model, J, y, z) for all k
"""
from ._lib import Context, NoDeviceError, RdhipError, device_count, last_kernel_name, lib, tune, LIB_PATH, HEADER_PATH, EXPORTS  # noqa: F401
from .core import *  # noqa: F401,F403
from .core import _gettimeinfo  # noqa: F401
from . import parallel  # noqa: F401
