"""Shared test helpers: golden-fixture loading and oracle model construction from fixture metadata."""
import glob
import os

import numpy as np

from oracle import rd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RULES = {"euler": O.EULER, "rk2": O.RK2, "rk3": O.RK3, "rk4": O.RK4}


def dyn_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "dyn_*.npz")))


def reference_fixtures():
    """ref_*.npz: outputs of the REAL RobotDynamics.jl on the committed inputs (tests/golden/make_golden_reference.jl; needs a Julia install, so
    none ship with the build image).  When present they take precedence over the torch-derived values in every test that loads a fixture."""
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "ref_*.npz")))


def load(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ref = os.path.join(GOLDEN, "ref_" + name + ".npz")
    if os.path.exists(ref):                       # reference-generated values win, key by key (inputs are identical by construction)
        r = dict(np.load(ref))
        for k in ("Z", "t", "dt", "X", "Xbar", "X0"):
            if k in r and k in d:
                assert np.array_equal(r[k], d[k], equal_nan=True), f"ref_{name}.npz was generated from different inputs"
        d.update({k: v for k, v in r.items() if not (np.issubdtype(v.dtype, np.floating) and np.isnan(v).all())})
        d["_reference_derived"] = np.asarray(True)
    return d


def model_spec(name, d):
    """(kind, params, frame) for a dyn_* fixture -- the same tuple feeds the oracle and the HIP library."""
    if name.startswith("cartpole"):
        return O.CARTPOLE, list(d["params"]), O.FRAME_WORLD
    if name.startswith("di"):
        return O.DOUBLE_INTEGRATOR, [int(d["params"][0])], O.FRAME_WORLD
    if name.startswith("linear"):
        A, B = d["A"], d["B"]
        return O.LINEAR, np.concatenate([[A.shape[0], B.shape[1]], A.flatten(order="F"), B.flatten(order="F")]), O.FRAME_WORLD
    if name.startswith(("satellite_", "body_", "satfull_")):
        return O.SATELLITE, np.concatenate([[float(d["mass"])], d["J"].flatten(order="F")]), int(d["frame"])
    if name.startswith("quadrotor"):
        return (O.QUADROTOR, np.concatenate([[float(d["mass"])], d["J"].flatten(order="F"), d["gravity"], [float(d["L"]), float(d["kf"]), float(d["km"])]]),
                int(d["frame"]))
    if name.startswith("satpair"):
        return O.SATELLITE_PAIR, np.concatenate([d["J1"].flatten(order="F"), d["J2"].flatten(order="F")]), O.FRAME_WORLD
    if name.startswith("liesat"):
        return O.LIE_SATELLITE, list(d["Jd"]), O.FRAME_WORLD
    raise KeyError(name)


def oracle_model(name, d):
    kind, params, frame = model_spec(name, d)
    return O.Model(kind, params, frame=frame, rot=int(d["rot"]) if "rot" in d else O.ROT_QUAT)      # rot: RigidBody{MRP|RodriguesParam} fixtures (make_golden_rot3.py)


def errmap_oracle_model(d):
    """the oracle's bare LieState for an errmap_* fixture (tests/golden/make_golden_round6.py)"""
    return O.lie_state(int(d["rot"]), [int(v) for v in d["P"]])


def three_parameter(name):
    """dyn_* fixtures of RigidBody{MRP} / RigidBody{RodriguesParam} (no user-defined Jacobian in the reference: src/rigidbody.jl:260)"""
    return "_mrp_" in name or "_rp_" in name


def rand_rigid(rng, N, m, ulo=0.0, uhi=1.0):
    """rand(model) for a RigidBody{QuatRotation}: r,v,w ~ U[0,1), q uniform on S^3 (src/rigidbody.jl:50-59)."""
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = np.concatenate([rng.random((N, 3)), q, rng.random((N, 6))], axis=1)
    return np.concatenate([X, ulo + (uhi - ulo) * rng.random((N, m))], axis=1)

from examples.custom_models import CARTPOLE_SRC, SATELLITE_SRC, TV_PENDULUM_SRC, TV_SATELLITE_SRC  # noqa: E402,F401
