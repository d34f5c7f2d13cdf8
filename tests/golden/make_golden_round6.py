#!/usr/bin/env python3
"""Round-6 fixtures: state_diff with a non-default error map (src/liestate.jl:26-32,216-217: `Rotations.rotation_error(R, R0, errmap)` per rotation).
  errmap_rigid.npz      LieState{QuatRotation,(3,6)}      -- the RigidBody state (src/rigidbody.jl:48)
  errmap_p323.npz       LieState{QuatRotation,(3,2,3)}    -- test/liestate.jl:33-51
  errmap_p323_mrp.npz   LieState{MRP,(3,2,3)}
  errmap_rigid_rp.npz   LieState{RodriguesParam,(3,6)}    -- test/liestate.jl:56-68 uses RodriguesParam
Keys: X, X0 (N x n), P, rot (0 quat / 1 MRP / 2 RodriguesParam), dX_cayley, dX_exponential, dX_mrp, dX_quatvec, dX_identity (three-parameter states only).

Expected values are derived from the GEOMETRY of the relative rotation (scipy: angle phi in [0, pi] and axis n of R0^-1 R), not from quaternion algebra, so
they share no arithmetic with the kernels or the oracle.  The reference's maps act on the quaternion dq = q0^-1 (x) q WITHOUT canonicalising its sign
(Rotations.jl 1.x), so each value comes in two branches, selected by s = sign(scalar(dq)) = sign(q0 . q) (a dot product of unit 4-vectors):
                          s > 0 (dq = [cos(phi/2); n sin(phi/2)])      s < 0 (dq = -[cos(phi/2); n sin(phi/2)])
  CayleyMap        v / w            n tan(phi/2)                              the same
  ExponentialMap   atan(|v|,w)/|v| v   (phi/2) n                              (phi/2 - pi) n          (the "long way round" half rotation vector)
  MRPMap           2 v / (1 + w)    2 n tan(phi/4)                           -2 n cot(phi/4)          (the shadow set)
  QuatVecMap       sign(w) v        n sin(phi/2)                              the same
  IdentityMap      params(R0 \\ R)  MRP states: half the MRPMap value; RodriguesParam states: the CayleyMap value; QuatRotation: the reference throws
Vector parts: x - x0.  Run:  python tests/golden/make_golden_round6.py"""
import os

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))


def unit_quat(kind, p):
    """the unit quaternion [w x y z] Rotations.jl converts a state's rotation to (QuatRotation(w,x,y,z) normalises; RodriguesParam g -> [1; g] / |.|;
    MRP p -> [1 - p.p; 2 p] / (1 + p.p))"""
    if kind == 0:
        return p / np.linalg.norm(p, axis=1, keepdims=True)
    if kind == 2:
        q = np.concatenate([np.ones((len(p), 1)), p], axis=1)
        return q / np.linalg.norm(q, axis=1, keepdims=True)
    n2 = (p * p).sum(axis=1, keepdims=True)
    return np.concatenate([1 - n2, 2 * p], axis=1) / (1 + n2)


def expected(kind, ra, rb):
    """rotation parts for every map: ra, rb = the rotation parameters of x and x0 (N x 4 or N x 3)"""
    qa, qb = unit_quat(kind, ra), unit_quat(kind, rb)
    s = np.sign((qa * qb).sum(axis=1))[:, None]                      # sign of the scalar part of conj(q0) (x) q
    rel = R.from_quat(np.roll(qb, -1, axis=1)).inv() * R.from_quat(np.roll(qa, -1, axis=1))     # scipy is scalar-last; R0^-1 R
    rv = rel.as_rotvec()
    phi = np.linalg.norm(rv, axis=1, keepdims=True)
    n = rv / phi
    out = {"cayley": n * np.tan(phi / 2), "quatvec": n * np.sin(phi / 2),
           "exponential": np.where(s > 0, (phi / 2) * n, (phi / 2 - np.pi) * n),
           "mrp": np.where(s > 0, 2 * n * np.tan(phi / 4), -2 * n / np.tan(phi / 4))}
    if kind == 1:
        out["identity"] = out["mrp"] / 2
    elif kind == 2:
        out["identity"] = out["cayley"]
    return out


def make(name, kind, P, rng, N=24):
    rp = 4 if kind == 0 else 3
    nrot = len(P) - 1
    n = sum(P) + rp * nrot
    X, X0 = rng.random((N, n)) - 0.3, rng.random((N, n)) - 0.3
    rows, cols, r, c = [], [], 0, 0
    for v, pv in enumerate(P):
        r += pv; c += pv
        if v < nrot:
            rows.append(r); cols.append(c); r += rp; c += 3
    for r0 in rows:                                                    # rotations: uniform attitudes (quaternions of both signs); the last six knot points a
        q = rng.standard_normal((N, 4)); q0 = rng.standard_normal((N, 4))   # tiny relative rotation (ExponentialMap's series branch, |v| <= 1e-6)
        if kind == 0:
            X[:, r0:r0 + 4], X0[:, r0:r0 + 4] = q, q0                 # NOT normalised: the QuatRotation constructor does that (src/liestate.jl:216)
            small = R.from_quat(np.roll(q0[-6:] / np.linalg.norm(q0[-6:], axis=1, keepdims=True), -1, axis=1)) * R.from_rotvec(1e-7 * rng.standard_normal((6, 3)))
            X[-6:, r0:r0 + 4] = np.roll(small.as_quat(), 1, axis=1) * (1 + rng.random((6, 1)))
        else:
            conv = (lambda rr: rr.as_mrp()) if kind == 1 else (lambda rr: np.tan(rr.magnitude() / 2)[:, None] * rr.as_rotvec() / rr.magnitude()[:, None])
            ra = R.from_quat(q / np.linalg.norm(q, axis=1, keepdims=True)); rb = R.from_quat(q0 / np.linalg.norm(q0, axis=1, keepdims=True))
            pa, pb = conv(ra), conv(rb)
            if kind == 1:                                              # half of the MRPs from the shadow set (|p| > 1: negative scalar part after conversion)
                flip = rng.random(N) < 0.5
                pa[flip] = -pa[flip] / (pa[flip] ** 2).sum(axis=1, keepdims=True)
            small = rb[-6:] * R.from_rotvec(1e-7 * rng.standard_normal((6, 3)))
            pa[-6:] = conv(small)
            X[:, r0:r0 + 3], X0[:, r0:r0 + 3] = pa, pb
    e = sum(P) + 3 * nrot
    out = {"X": X, "X0": X0, "P": np.asarray(P), "rot": np.asarray(kind)}
    maps = ["cayley", "exponential", "mrp", "quatvec"] + (["identity"] if kind != 0 else [])
    for mp in maps:
        d = np.zeros((N, e))
        r = c = 0
        for v, pv in enumerate(P):
            d[:, c:c + pv] = X[:, r:r + pv] - X0[:, r:r + pv]
            r += pv; c += pv
            if v < nrot:
                d[:, c:c + 3] = expected(kind, X[:, r:r + rp], X0[:, r:r + rp])[mp]
                r += rp; c += 3
        out["dX_" + mp] = d
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    return name


def main():
    rng = np.random.default_rng(606)
    names = [make("errmap_rigid", 0, (3, 6), rng), make("errmap_p323", 0, (3, 2, 3), rng), make("errmap_p323_mrp", 1, (3, 2, 3), rng), make("errmap_rigid_rp", 2, (3, 6), rng)]
    print("wrote", ", ".join(n + ".npz" for n in names))


if __name__ == "__main__":
    main()
