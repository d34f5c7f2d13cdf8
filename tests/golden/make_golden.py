#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- "restatement-derived" golden vectors.

Julia is not installed in the build image, and RobotDynamics.jl ships no stored numeric vectors,
so golden values cannot come from the reference itself.  This script is a THIRD, independent
restatement of the math (after oracle/rd_oracle.cpp and the HIP kernels): plain torch fp64
formulas differentiated by torch.func.jacfwd / hessian, quaternion error maps taken from their
mathematical definition (q (x) cayley(phi)) rather than the L(q)H closed form, and scipy's
Rotation class for the Cayley state difference.  The oracle is checked against these files in
tests/test_oracle_golden.py; the HIP path is checked against both.

Reference formulas followed (file:line in /root/reference):
  cartpole          examples/cartpole.jl:15-34
  rigid body        src/rigidbody.jl:213-236, wrenches examples/single_satellite.jl:18-27,
                    test/quadrotor.jl:56-96, test/rigidbody_test.jl:26-31,53-54
  satellite pair    examples/satellite_pair.jl:14-38
  lie satellite     test/liemodel.jl:14-22
  double integrator test/double_integrator.jl:97-107
  RK steps          src/integration.jl:73-83,130-135,280-286 ; RK2 = explicit midpoint (v0.3)
  error state       src/liestate.jl:210-320, src/statevectortype.jl:144-155

Run (in the build container only; needs torch + scipy, no GPU):  python tests/golden/make_golden.py
"""
import os

import numpy as np
import torch
from scipy.spatial.transform import Rotation as SciRot
from torch.func import hessian, jacfwd

torch.set_default_dtype(torch.float64)
HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- quaternion algebra (scalar first)
def qmul(a, b):
    aw, ax, ay, az = a[0], a[1], a[2], a[3]
    bw, bx, by, bz = b[0], b[1], b[2], b[3]
    return torch.stack([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def qconj(q):
    return torch.stack([q[0], -q[1], -q[2], -q[3]])


def qrot(q, r):
    """q*r for a possibly non-unit QuatRotation: vector part of q (x) [0;r] (x) conj(q)."""
    p = torch.cat([torch.zeros(1), r])
    return qmul(qmul(q, p), qconj(q))[1:]


def kinematics(q, w):
    return 0.5 * qmul(q, torch.cat([torch.zeros(1), w]))


# ---------------------------------------------------------------- continuous dynamics
def cartpole(p):
    mc, mp, l, g = p

    def f(x, u):
        s, c = torch.sin(x[1]), torch.cos(x[1])
        qd = x[2:4]
        H = torch.stack([torch.stack([torch.as_tensor(mc + mp), mp * l * c]),
                         torch.stack([mp * l * c, torch.as_tensor(mp * l ** 2)])])
        Cqd = torch.stack([-mp * qd[1] * l * s * qd[1], torch.zeros(())])
        G = torch.stack([torch.zeros(()), mp * g * l * s])
        Bu = torch.stack([u[0], torch.zeros(())])
        qdd = -torch.linalg.solve(H, Cqd + G - Bu)
        return torch.cat([qd, qdd])
    return f, 4, 1


def double_integrator(D):
    def f(x, u):
        return torch.cat([x[D:], u])
    return f, 2 * D, D


def linear(A, B):
    A, B = torch.as_tensor(A), torch.as_tensor(B)

    def f(x, u):
        return A @ x + B @ u
    return f, A.shape[0], B.shape[1]


def rigid_body(mass, J, wrench, m, body_frame):
    J = torch.as_tensor(J)
    Jinv = torch.linalg.inv(J)

    def f(x, u):
        q, v, w = x[3:7], x[7:10], x[10:13]
        F, T = wrench(x, u)
        qd = kinematics(q, w)
        if not body_frame:
            rd, vd = v, F / mass
        else:
            rd = qrot(q, v)
            vd = qrot(qconj(q), F / mass) - torch.linalg.cross(w, v)
        wd = Jinv @ (T - torch.linalg.cross(w, J @ w))
        return torch.cat([rd, qd, vd, wd])
    return f, 13, m


def satellite_wrench(x, u):
    return qrot(x[3:7], u[0:3]), u[3:6]


def quadrotor_wrench(mass, g, L, kf, km):
    g = torch.as_tensor(g)

    def wrench(x, u):
        Fi = torch.clamp(kf * u, min=0.0)
        F = mass * g + qrot(x[3:7], torch.stack([torch.zeros(()), torch.zeros(()), Fi.sum()]))
        Mi = km * u
        T = torch.stack([L * (Fi[1] - Fi[3]), L * (Fi[2] - Fi[0]), Mi[0] - Mi[1] + Mi[2] - Mi[3]])
        return F, T
    return wrench


def satellite_pair(J1, J2):
    J1, J2 = torch.as_tensor(J1), torch.as_tensor(J2)

    def f(x, u):
        out = []
        for b, J in enumerate((J1, J2)):
            q, w, ub = x[7 * b:7 * b + 4], x[7 * b + 4:7 * b + 7], u[3 * b:3 * b + 3]
            out += [kinematics(q, w), torch.linalg.solve(J, ub - torch.linalg.cross(w, J @ w))]
        return torch.cat(out)
    return f, 14, 6


def lie_satellite(Jd):
    Jd = torch.as_tensor(Jd)

    def f(x, u):
        w, q = x[0:3], x[3:7] / torch.linalg.norm(x[3:7])
        return torch.cat([(u - torch.linalg.cross(w, Jd * w)) / Jd, kinematics(q, w)])
    return f, 7, 3


# ---------------------------------------------------------------- explicit RK steps
def rk_step(f, rule):
    def step(x, u, h):
        if rule == "euler":
            return x + h * f(x, u)
        if rule == "rk2":
            k1 = f(x, u) * h
            return x + f(x + k1 / 2, u) * h
        if rule == "rk3":
            k1 = f(x, u) * h
            k2 = f(x + k1 / 2, u) * h
            k3 = f(x - k1 + 2 * k2, u) * h
            return x + (k1 + 4 * k2 + k3) / 6
        k1 = f(x, u) * h
        k2 = f(x + k1 / 2, u) * h
        k3 = f(x + k2 / 2, u) * h
        k4 = f(x + k3, u) * h
        return x + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return step


RULES = ("euler", "rk2", "rk3", "rk4")


def evaluate_case(f, n, m, Z, dt):
    """Z: (N, n+m). Returns dict with xdot, Jc and per-rule xn, Jd (Jacobians wrt z=[x;u], shape (N,n,n+m))."""
    Zt, dtt = torch.as_tensor(Z), torch.as_tensor(dt)
    N = Zt.shape[0]
    out = {}
    fz = lambda z: f(z[:n], z[n:])
    # plain per-knot loops (no vmap: batching rules of linalg.solve under vmap(jacfwd) gave wrong
    # derivatives for knots > 0 when this script was first written)
    out["xdot"] = torch.stack([fz(Zt[k]) for k in range(N)]).numpy()
    out["Jc"] = torch.stack([jacfwd(fz)(Zt[k]) for k in range(N)]).numpy()
    for rule in RULES:
        step = rk_step(f, rule)
        g = lambda z, h: step(z[:n], z[n:], h)
        out["xn_" + rule] = torch.stack([g(Zt[k], dtt[k]) for k in range(N)]).numpy()
        out["Jd_" + rule] = torch.stack([jacfwd(g, argnums=0)(Zt[k], dtt[k]) for k in range(N)]).numpy()
    return out


# ---------------------------------------------------------------- error-state maps from their definitions
def cayley(phi):
    return torch.cat([torch.ones(1), phi]) / torch.sqrt(1 + phi @ phi)


def lie_layout(P, rp=4):
    """Return list of ('v', start, len) / ('r', start) blocks for LieState{Quat,P} (0-based)."""
    blocks, r = [], 0
    for i, p in enumerate(P):
        if p:
            blocks.append(("v", r, p))
        r += p
        if i < len(P) - 1:
            blocks.append(("r", r, rp))
            r += rp
    return blocks, r


def compose(P):
    """x (+) delta: vector parts add, rotations q <- normalize(q) (x) cayley(delta_q)."""
    blocks, n = lie_layout(P)

    def comp(x, d):
        out, c = [], 0
        for b in blocks:
            if b[0] == "v":
                out.append(x[b[1]:b[1] + b[2]] + d[c:c + b[2]])
                c += b[2]
            else:
                q = x[b[1]:b[1] + 4]
                out.append(qmul(q / torch.linalg.norm(q), cayley(d[c:c + 3])))
                c += 3
        return torch.cat(out)
    return comp, n


def errstate_case(P, X, Xbar, X0):
    comp, n = compose(P)
    e = sum(P) + 3 * (len(P) - 1)
    Xt, Xb = torch.as_tensor(X), torch.as_tensor(Xbar)
    z = torch.zeros(e)
    G = torch.stack([jacfwd(lambda d: comp(Xt[k], d))(z) for k in range(Xt.shape[0])]).numpy()
    dG = torch.stack([hessian(lambda d: Xb[k] @ comp(Xt[k], d))(z) for k in range(Xt.shape[0])]).numpy()
    # state_diff via scipy: Rodrigues parameters of q0^-1 q  (test/rbstate.jl:157)
    blocks, _ = lie_layout(P)
    dX = np.zeros((X.shape[0], e))
    for k in range(X.shape[0]):
        c = 0
        for b in blocks:
            if b[0] == "v":
                dX[k, c:c + b[2]] = X[k, b[1]:b[1] + b[2]] - X0[k, b[1]:b[1] + b[2]]
                c += b[2]
            else:
                q, q0 = X[k, b[1]:b[1] + 4], X0[k, b[1]:b[1] + 4]
                r = SciRot.from_quat(np.roll(q0, -1)).inv() * SciRot.from_quat(np.roll(q, -1))   # scipy: scalar last
                qq = np.roll(r.as_quat(), 1)
                dX[k, c:c + 3] = qq[1:] / qq[0]
                c += 3
    return {"G": G, "dG": dG, "dX": dX}


def expansion_case(f, n, m, P, Z, dt):
    """Error-state expansion of consecutive knot points from the DEFINITIONS (no L(q)H closed form, no chain through A):
         E_k = G(x_{k+1})' * d/d(delta, u) [ f_d(x_k (+) delta, u_k) ]  at delta = 0,
    with G(x) = d/d(delta) [x (+) delta] at 0 (unit quaternions, so x (+) 0 = x).  Euclidean models (P = None): E_k = d f_d / d z.
    Also the dynamics error f_d(z_k) - x_{k+1} (reference src/discrete_dynamics.jl:134-135)."""
    Zt, dtt = torch.as_tensor(Z), torch.as_tensor(dt)
    N = Zt.shape[0]
    out = {}
    if P is not None:
        comp, _ = compose(P)
        e = sum(P) + 3 * (len(P) - 1)
    for rule in ("rk3", "rk4"):
        step = rk_step(f, rule)
        Es, errs = [], []
        for k in range(N - 1):
            x, u, h = Zt[k, :n], Zt[k, n:], dtt[k]
            errs.append(step(x, u, h) - Zt[k + 1, :n])
            if P is None:
                Es.append(jacfwd(lambda z: step(z[:n], z[n:], h))(Zt[k]))
                continue
            z0 = torch.zeros(e)
            G2 = jacfwd(lambda d: comp(Zt[k + 1, :n], d))(z0)
            Jdu = jacfwd(lambda du: step(comp(x, du[:e]), u + du[e:], h))(torch.zeros(e + m))
            Es.append(G2.T @ Jdu)
        out["E_" + rule] = torch.stack(Es).numpy()
        out["err_" + rule] = torch.stack(errs).numpy()
    return out


# ---------------------------------------------------------------- input generators (rand(model))
def rand_rigid(rng, N, m, ulo=0.0, uhi=1.0):
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = np.concatenate([rng.random((N, 3)), q, rng.random((N, 6))], axis=1)
    U = ulo + (uhi - ulo) * rng.random((N, m))
    return np.concatenate([X, U], axis=1)


def main():
    rng = np.random.default_rng(20260929)
    N = 12
    cases = {}

    def add(name, fnm, Z, dt, meta):
        f, n, m = fnm
        res = evaluate_case(f, n, m, Z, dt)
        res.update({"Z": Z, "dt": dt, "t": np.arange(N) * 0.1})
        res.update({k: np.asarray(v) for k, v in meta.items()})
        cases[name] = res

    dt = np.full(N, 0.1)
    dt_mixed = np.array([0.1, 0.01, 0.05, 0.2] * 3)
    add("cartpole", cartpole((1.0, 0.2, 0.5, 9.81)), rng.random((N, 5)), dt_mixed, {"params": [1.0, 0.2, 0.5, 9.81]})
    # deterministic input of test/implicit_dynamics_test.jl:38-44: x = 0, u = 1, dt = 0.01
    Zd = np.zeros((N, 5)); Zd[:, 4] = 1.0
    add("cartpole_x0u1", cartpole((1.0, 0.2, 0.5, 9.81)), Zd, np.full(N, 0.01), {"params": [1.0, 0.2, 0.5, 9.81]})
    for D in (1, 2, 3):
        add(f"di{D}", double_integrator(D), rng.random((N, 3 * D)), dt, {"params": [D]})
    A = rng.standard_normal((4, 4)); B = rng.standard_normal((4, 2))
    add("linear42", linear(A, B), rng.random((N, 6)), dt, {"A": A, "B": B})
    Jsat = np.eye(3)
    Jbody = np.diag([2.0, 3.0, 1.0])
    Jfull = np.array([[2.0, 0.1, -0.2], [0.1, 3.0, 0.3], [-0.2, 0.3, 1.5]])
    for frame in (0, 1):
        tag = "world" if frame == 0 else "body"
        add(f"satellite_{tag}", rigid_body(1.0, Jsat, satellite_wrench, 6, frame), rand_rigid(rng, N, 6), dt, {"mass": 1.0, "J": Jsat, "frame": frame})
        add(f"body_{tag}", rigid_body(2.0, Jbody, satellite_wrench, 6, frame), rand_rigid(rng, N, 6), dt_mixed, {"mass": 2.0, "J": Jbody, "frame": frame})
        add(f"satfull_{tag}", rigid_body(1.7, Jfull, satellite_wrench, 6, frame), rand_rigid(rng, N, 6), dt, {"mass": 1.7, "J": Jfull, "frame": frame})
        trim = 9.81 * 0.5 / 4
        Jq = np.diag([0.0023, 0.0023, 0.004])
        Zq = rand_rigid(rng, N, 4, 0.5 * trim, 1.5 * trim)
        Zq[0, 13] = -0.3   # one negative rotor command: exercises max(0, kf*u)
        add(f"quadrotor_{tag}", rigid_body(0.5, Jq, quadrotor_wrench(0.5, [0, 0, -9.81], 0.1750, 1.0, 0.0245), 4, frame), Zq, dt,
            {"mass": 0.5, "J": Jq, "frame": frame, "gravity": [0, 0, -9.81], "L": 0.1750, "kf": 1.0, "km": 0.0245})
    # satellite pair: x = [q1, w1, q2, w2]
    q1 = rng.standard_normal((N, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.standard_normal((N, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    Zp = np.concatenate([q1, rng.random((N, 3)), q2, rng.random((N, 3)), rng.random((N, 6))], axis=1)
    add("satpair", satellite_pair(np.eye(3), 2 * np.eye(3)), Zp, dt, {"J1": np.eye(3), "J2": 2 * np.eye(3)})
    ql = rng.standard_normal((N, 4)); ql /= np.linalg.norm(ql, axis=1, keepdims=True)
    add("liesat", lie_satellite([1.0, 1.0, 1.0]), np.concatenate([rng.random((N, 3)), ql, rng.random((N, 3))], axis=1), dt, {"Jd": [1.0, 1.0, 1.0]})

    for name, res in cases.items():
        np.savez_compressed(os.path.join(HERE, f"dyn_{name}.npz"), **res)

    # error-state expansion / dynamics error along the same knot sequences (unit quaternions), from the definitions
    specs = {"cartpole": (cartpole((1.0, 0.2, 0.5, 9.81)), None), "satellite_world": (rigid_body(1.0, Jsat, satellite_wrench, 6, 0), (3, 6)),
             "body_body": (rigid_body(2.0, Jbody, satellite_wrench, 6, 1), (3, 6)), "satfull_world": (rigid_body(1.7, Jfull, satellite_wrench, 6, 0), (3, 6)),
             "quadrotor_world": (rigid_body(0.5, np.diag([0.0023, 0.0023, 0.004]), quadrotor_wrench(0.5, [0, 0, -9.81], 0.1750, 1.0, 0.0245), 4, 0), (3, 6)),
             "satpair": (satellite_pair(np.eye(3), 2 * np.eye(3)), (0, 3, 3)), "liesat": (lie_satellite([1.0, 1.0, 1.0]), (3, 0))}
    for name, (fnm, P) in specs.items():
        f, n, m = fnm
        res = expansion_case(f, n, m, P, cases[name]["Z"], cases[name]["dt"])
        np.savez_compressed(os.path.join(HERE, f"exp_{name}.npz"), **res)

    # error-state fixtures: rigid body (3,6), satellite pair (0,3,3), test/liestate.jl's (3,2,3) with NON-unit
    # quaternions (x = rand(n), test/liestate.jl:50-51), lie satellite (3,0)
    for tag, P in (("rigid", (3, 6)), ("satpair", (0, 3, 3)), ("p323", (3, 2, 3)), ("liesat", (3, 0))):
        _, n = lie_layout(P)
        X, Xb, X0 = rng.random((N, n)), rng.random((N, n)), rng.random((N, n))
        if tag != "p323":   # unit quaternions as produced by rand(model)
            for b in lie_layout(P)[0]:
                if b[0] == "r":
                    for A_ in (X, X0):
                        A_[:, b[1]:b[1] + 4] = rng.standard_normal((N, 4))
                        A_[:, b[1]:b[1] + 4] /= np.linalg.norm(A_[:, b[1]:b[1] + 4], axis=1, keepdims=True)
        res = errstate_case(P, X, Xb, X0)
        res.update({"X": X, "Xbar": Xb, "X0": X0, "P": np.asarray(P)})
        np.savez_compressed(os.path.join(HERE, f"err_{tag}.npz"), **res)
    print("wrote", len(cases) + 4 + len(specs), "fixtures to", HERE)


if __name__ == "__main__":
    main()
