#!/usr/bin/env python3
"""Generate tests/golden/{dyn,err,nab,exp}_*_{mrp,rp}*.npz -- "restatement-derived" golden vectors for RigidBody{MRP} / RigidBody{RodriguesParam}
and LieState{MRP|RodriguesParam,P} (src/rigidbody.jl:80-107,213-236; src/liestate.jl:262-298; test/rigidbody_test.jl:79-98; test/liestate.jl:56-68).

Independent of oracle/rd_oracle.cpp and of the kernels: nothing here uses the closed forms (1 - p.p) I + 2 (skew(p) + p p') or
I + skew(g) + g g'.  Kinematics and the error-state blocks come from their DEFINITIONS, differentiated by torch:
    kinematics(R, w)      = d/d(eps) params(R * exp-map-free small rotation [1, w eps / 2])        at eps = 0
    grad-differential(R)  = d/d(delta) params(R * R_type(delta))                                    at delta = 0     (Rotations.jl)
with params(q) = v / (1 + w) for MRP, v / w for RodriguesParam, and R_type(delta) the unit quaternion of delta in the same parameters.
A rotation acts on a vector through its unit quaternion.  state_diff = Rodrigues parameters of R0^-1 R (CayleyMap), through scipy.
Round 4 (second stage, own random stream, so that the files of the first stage stay bit-identical):
    grad^2-differential(R, b) = the Hessian of delta -> b . params(R * R_type(delta))                at delta = 0     (nab_*.npz; src/liestate.jl:300-320)
    error-state expansion     = G(x_{k+1})' d/d(delta, u) f_d(x_k (+) delta, u_k)                    at delta = 0     (exp_*_{mrp,rp}_*.npz)
(the quaternion fixtures of make_golden.py hold the same two definitions, where they reproduce Rotations.jl's -(q.b) I.)

Run (build container only; torch + scipy, no GPU):  python tests/golden/make_golden_rot3.py"""
import os
import sys

import numpy as np
import torch
from scipy.spatial.transform import Rotation as SciRot
from torch.func import hessian, jacfwd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (qmul, qconj, qrot, rk_step, evaluate_case, quadrotor_wrench)

MRP, RP = 1, 2


def to_quat(rot, p):
    n = p @ p
    if rot == MRP:
        return torch.cat([(1 - n).reshape(1), 2 * p]) / (1 + n)
    return torch.cat([torch.ones(1), p]) / torch.sqrt(1 + n)


def from_quat(rot, q):
    return q[1:] / (1 + q[0]) if rot == MRP else q[1:] / q[0]


def kinematics(rot, p, w):
    return jacfwd(lambda eps: from_quat(rot, MG.qmul(to_quat(rot, p), torch.cat([torch.ones(1), w * eps / 2]))))(torch.zeros(()))


def rigid_body3(rot, mass, J, wrench_kind, m, body_frame, quad=None):
    J = torch.as_tensor(J)
    Jinv = torch.linalg.inv(J)

    def f(x, u):
        p, v, w = x[3:6], x[6:9], x[9:12]
        q = to_quat(rot, p)
        if wrench_kind == "sat":
            F, T = MG.qrot(q, u[0:3]), u[3:6]
        else:
            mq, g, L, kf, km = quad
            Fi = torch.clamp(kf * u, min=0.0)
            F = mq * torch.as_tensor(g) + MG.qrot(q, torch.stack([torch.zeros(()), torch.zeros(()), Fi.sum()]))
            Mi = km * u
            T = torch.stack([L * (Fi[1] - Fi[3]), L * (Fi[2] - Fi[0]), Mi[0] - Mi[1] + Mi[2] - Mi[3]])
        pd = kinematics(rot, p, w)
        if not body_frame:
            rd, vd = v, F / mass
        else:
            rd = MG.qrot(q, v)
            vd = MG.qrot(MG.qconj(q), F / mass) - torch.linalg.cross(w, v)
        wd = Jinv @ (T - torch.linalg.cross(w, J @ w))
        return torch.cat([rd, pd, vd, wd])
    return f, 12, m


def lie_layout(P):
    blocks, r = [], 0
    for i, p in enumerate(P):
        if p:
            blocks.append(("v", r, p))
        r += p
        if i < len(P) - 1:
            blocks.append(("r", r, 3))
            r += 3
    return blocks, r


def errstate_case(rot, P, X, X0):
    blocks, n = lie_layout(P)
    e = n

    def comp(x, d):
        out, c = [], 0
        for b in blocks:
            if b[0] == "v":
                out.append(x[b[1]:b[1] + b[2]] + d[c:c + b[2]]); c += b[2]
            else:
                out.append(from_quat(rot, MG.qmul(to_quat(rot, x[b[1]:b[1] + 3]), to_quat(rot, d[c:c + 3])))); c += 3
        return torch.cat(out)
    Xt = torch.as_tensor(X)
    G = torch.stack([jacfwd(lambda d: comp(Xt[k], d))(torch.zeros(e)) for k in range(Xt.shape[0])]).numpy()
    dX = np.zeros((X.shape[0], e))
    for k in range(X.shape[0]):
        c = 0
        for b in blocks:
            if b[0] == "v":
                dX[k, c:c + b[2]] = X[k, b[1]:b[1] + b[2]] - X0[k, b[1]:b[1] + b[2]]; c += b[2]
            else:
                q = to_quat(rot, torch.as_tensor(X[k, b[1]:b[1] + 3])).numpy(); q0 = to_quat(rot, torch.as_tensor(X0[k, b[1]:b[1] + 3])).numpy()
                r = SciRot.from_quat(np.roll(q0, -1)).inv() * SciRot.from_quat(np.roll(q, -1))
                qq = np.roll(r.as_quat(), 1)
                dX[k, c:c + 3] = qq[1:] / qq[0]; c += 3
    return {"G": G, "dX": dX}


def main():
    rng = np.random.default_rng(20260930)
    N = 10
    t = np.arange(N) * 0.1
    dt = np.full(N, 0.1)
    dt_mixed = np.array([0.1, 0.01, 0.05, 0.2, 0.1] * 2)
    Jbody = np.diag([2.0, 3.0, 1.0])
    Jfull = np.array([[2.0, 0.1, -0.2], [0.1, 3.0, 0.3], [-0.2, 0.3, 1.5]])
    Jq = np.diag([0.0023, 0.0023, 0.004])
    trim = 9.81 * 0.5 / 4
    quad = (0.5, [0, 0, -9.81], 0.1750, 1.0, 0.0245)
    nfix = 0
    for rot, rtag in ((MRP, "mrp"), (RP, "rp")):
        for frame, ftag in ((0, "world"), (1, "body")):
            # rand(RigidBody{R}): r, v, w, u ~ U[0,1); the attitude as the parameters of a uniformly random rotation, kept inside the unit ball / moderate
            def rand_state(m, ulo, uhi):
                q = rng.standard_normal((N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 0] < 0] *= -1
                p = q[:, 1:] / (1 + q[:, :1]) if rot == MRP else q[:, 1:] / np.maximum(q[:, :1], 0.3)
                return np.concatenate([rng.random((N, 3)), p, rng.random((N, 6)), ulo + (uhi - ulo) * rng.random((N, m))], axis=1)
            cases = {f"satellite_{rtag}_{ftag}": (rigid_body3(rot, 2.0, Jbody, "sat", 6, frame), rand_state(6, 0.0, 1.0), dt_mixed, {"mass": 2.0, "J": Jbody}),
                     f"satfull_{rtag}_{ftag}": (rigid_body3(rot, 1.7, Jfull, "sat", 6, frame), rand_state(6, 0.0, 1.0), dt, {"mass": 1.7, "J": Jfull})}
            Zq = rand_state(4, 0.5 * trim, 1.5 * trim); Zq[0, 12] = -0.3
            cases[f"quadrotor_{rtag}_{ftag}"] = (rigid_body3(rot, 0.5, Jq, "quad", 4, frame, quad), Zq, dt,
                                                 {"mass": 0.5, "J": Jq, "gravity": [0, 0, -9.81], "L": 0.1750, "kf": 1.0, "km": 0.0245})
            for name, (fnm, Z, dts, meta) in cases.items():
                f, n, m = fnm
                res = MG.evaluate_case(f, n, m, Z, dts)
                res.update({"Z": Z, "dt": dts, "t": t, "frame": np.asarray(frame), "rot": np.asarray(rot)})
                res.update({k: np.asarray(v) for k, v in meta.items()})
                np.savez_compressed(os.path.join(HERE, f"dyn_{name}.npz"), **res); nfix += 1
        for tag, P in (("rigid", (3, 6)), ("p323", (3, 2, 3))):      # RigidBody{R} ; test/liestate.jl:56-68
            _, n = lie_layout(P)
            X, X0 = rng.random((N, n)), rng.random((N, n))
            res = errstate_case(rot, P, X, X0)
            res.update({"X": X, "X0": X0, "P": np.asarray(P), "rot": np.asarray(rot)})
            np.savez_compressed(os.path.join(HERE, f"err_{tag}_{rtag}.npz"), **res); nfix += 1
    print("wrote", nfix, "fixtures to", HERE)


def compose3(rot, P):
    blocks, n = lie_layout(P)

    def comp(x, d):
        out, c = [], 0
        for b in blocks:
            if b[0] == "v":
                out.append(x[b[1]:b[1] + b[2]] + d[c:c + b[2]]); c += b[2]
            else:
                out.append(from_quat(rot, MG.qmul(to_quat(rot, x[b[1]:b[1] + 3]), to_quat(rot, d[c:c + 3])))); c += 3
        return torch.cat(out)
    return comp, n


def second_stage():
    """nab_* and exp_* fixtures on the inputs of the first stage's files (loaded, not regenerated)."""
    rng = np.random.default_rng(20261001)
    nfix = 0
    for rot, rtag in ((MRP, "mrp"), (RP, "rp")):
        for tag, P in (("rigid", (3, 6)), ("p323", (3, 2, 3))):
            comp, n = compose3(rot, P)
            X = np.load(os.path.join(HERE, f"err_{tag}_{rtag}.npz"))["X"]
            Xbar = rng.standard_normal(X.shape)
            Xt, Xb = torch.as_tensor(X), torch.as_tensor(Xbar)
            dG = torch.stack([hessian(lambda d: Xb[k] @ comp(Xt[k], d))(torch.zeros(n)) for k in range(X.shape[0])]).numpy()
            np.savez_compressed(os.path.join(HERE, f"nab_{tag}_{rtag}.npz"), X=X, Xbar=Xbar, dG=dG, P=np.asarray(P), rot=np.asarray(rot)); nfix += 1
        comp, n = compose3(rot, (3, 6))
        e = n
        for ftag in ("world", "body"):
            for kind in ("satellite", "satfull", "quadrotor"):
                d = np.load(os.path.join(HERE, f"dyn_{kind}_{rtag}_{ftag}.npz"))
                Z, dts, frame = d["Z"], d["dt"], int(d["frame"])
                if kind == "quadrotor":
                    quad = (float(d["mass"]), [float(g) for g in d["gravity"]], float(d["L"]), float(d["kf"]), float(d["km"]))
                    f, _, m = rigid_body3(rot, float(d["mass"]), d["J"], "quad", 4, frame, quad)
                else:
                    f, _, m = rigid_body3(rot, float(d["mass"]), d["J"], "sat", 6, frame)
                Zt, dtt = torch.as_tensor(Z), torch.as_tensor(dts)
                out = {}
                for rule in ("rk3", "rk4"):
                    step = MG.rk_step(f, rule)
                    Es, errs = [], []
                    for k in range(Z.shape[0] - 1):
                        x, u, h = Zt[k, :n], Zt[k, n:], dtt[k]
                        errs.append(step(x, u, h) - Zt[k + 1, :n])
                        G2 = jacfwd(lambda dd: comp(Zt[k + 1, :n], dd))(torch.zeros(e))
                        Jdu = jacfwd(lambda du: step(comp(x, du[:e]), u + du[e:], h))(torch.zeros(e + m))
                        Es.append(G2.T @ Jdu)
                    out["E_" + rule] = torch.stack(Es).numpy()
                    out["err_" + rule] = torch.stack(errs).numpy()
                np.savez_compressed(os.path.join(HERE, f"exp_{kind}_{rtag}_{ftag}.npz"), **out); nfix += 1
    print("second stage: wrote", nfix, "fixtures to", HERE)


if __name__ == "__main__":
    torch.set_default_dtype(torch.float64)
    if "--second-stage-only" not in sys.argv:
        main()
    second_stage()
