"""ctypes front-end of the CPU oracle (oracle/rd_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (robotdynamics.jl_amd/rdhip) must never import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librd_oracle.so")

# enums -- numerically identical to include/rdhip.h
CARTPOLE, DOUBLE_INTEGRATOR, LINEAR, SATELLITE, QUADROTOR, SATELLITE_PAIR, LIE_SATELLITE = range(7)
TV_PENDULUM, TV_SATELLITE = 8, 9     # explicitly time-dependent test models (oracle only; the product runs them as user-defined models)
LIE_STATE = 10                       # a bare LieState{R,P} (src/liestate.jl:75-83): the error-state maps only
ROT_QUAT, ROT_MRP, ROT_RP = range(3)
FRAME_WORLD, FRAME_BODY = range(2)
EULER, RK2, RK3, RK4 = range(4)
CONTINUOUS = -1
FORWARD_AD, FINITE_DIFF, USER = range(3)
STATIC_RETURN, INPLACE = range(2)


def build(force=False):
    """Compile the oracle with g++ (no GPU needed)."""
    src = os.path.join(_HERE, "rd_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/librd_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_FAST_PATH = os.path.join(_HERE, "_build", "librd_oracle_fast.so")


def timed_jacobian_throughput(model, rule, Z, t, dt, threads, budget_s):
    """CPU-baseline helper for bench.py: knot-point Jacobians/s of the host-tuned build of this same oracle
    (-O3 -march=native, built on the machine it runs on), C entry point called directly on preallocated buffers."""
    import hashlib
    import time
    # -march=native: (re)build when the binary is missing, older than the source, or was built on a different CPU
    src = os.path.join(_HERE, "rd_oracle.cpp")
    try:
        cpu = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        cpu = []
    stamp = hashlib.sha1("".join(cpu).encode()).hexdigest()
    stamp_file = os.path.join(_HERE, "_build", "fast.stamp")
    fresh = (os.path.exists(_FAST_PATH) and os.path.getmtime(_FAST_PATH) >= os.path.getmtime(src)
             and os.path.exists(stamp_file) and open(stamp_file).read() == stamp)
    if not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/librd_oracle_fast.so"], stdout=subprocess.DEVNULL)
        open(stamp_file, "w").write(stamp)
    L = C.CDLL(_FAST_PATH)
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    N = Z.shape[0]
    t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (N,)))
    dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (N,)))
    J = np.empty((N, model.n + model.m, model.n))
    Y = np.empty((N, model.n))
    L.oracle_set_num_threads(int(threads))
    call = lambda: L.oracle_jacobian(model._buf, int(rule), FORWARD_AD, STATIC_RETURN, N, _dp(Z), _dp(t), _dp(dt), _dp(J), _dp(Y))
    call()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        call()
        done += N
    return done / (time.perf_counter() - t0), done


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Model:
    def __init__(self, kind, params=(), rot=ROT_QUAT, frame=FRAME_WORLD):
        L = lib()
        self._buf = C.create_string_buffer(L.oracle_sizeof_model())
        p = np.ascontiguousarray(np.asarray(params, dtype=np.float64))
        rc = L.oracle_model_init(self._buf, int(kind), int(rot), int(frame), _dp(p), int(p.size))
        if rc:
            raise ValueError("oracle_model_init failed")
        self.kind, self.rot, self.frame, self.params = kind, rot, frame, p
        n, m, p_, e = (C.c_int(), C.c_int(), C.c_int(), C.c_int())
        L.oracle_dims(self._buf, C.byref(n), C.byref(m), C.byref(p_), C.byref(e))
        self.n, self.m, self.p, self.e = n.value, m.value, p_.value, e.value

    # ---- integer queries -------------------------------------------------------------
    def dims(self):
        return (self.n, self.m, self.p)

    def liestate(self):
        rot, nP = C.c_int(), C.c_int()
        P = np.zeros(8, dtype=np.int32)
        lib().oracle_model_liestate(self._buf, C.byref(rot), C.byref(nP), _ip(P))
        return (rot.value, tuple(int(v) for v in P[: nP.value])) if nP.value else None

    def gen_inds(self):
        first = np.zeros(5, dtype=np.int32)
        ln = np.zeros(5, dtype=np.int32)
        rc = lib().oracle_gen_inds(self._buf, _ip(first), _ip(ln))
        if rc:
            raise ValueError("gen_inds: not a RigidBody")
        return {k: range(int(f), int(f + l)) for k, f, l in zip("rqvwu", first, ln)}

    # ---- batched evaluations ---------------------------------------------------------
    def _prep(self, Z, t, dt):
        Z = np.ascontiguousarray(Z, dtype=np.float64)   # shape (N, n+m): row k = knot k  (== col-major (n+m) x N)
        assert Z.ndim == 2 and Z.shape[1] == self.n + self.m
        N = Z.shape[0]
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (N,)))
        dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (N,)))
        return Z, t, dt, N

    def dynamics(self, Z, t, dt):
        Z, t, dt, N = self._prep(Z, t, dt)
        out = np.empty((N, self.n))
        rc = lib().oracle_dynamics(self._buf, N, _dp(Z), _dp(t), _dp(dt), _dp(out))
        if rc:
            raise RuntimeError(f"oracle_dynamics rc={rc}")
        return out

    def discrete_dynamics(self, rule, Z, t, dt, sig=STATIC_RETURN):
        Z, t, dt, N = self._prep(Z, t, dt)
        out = np.empty((N, self.n))
        rc = lib().oracle_discrete_dynamics(self._buf, int(rule), int(sig), N, _dp(Z), _dp(t), _dp(dt), _dp(out))
        if rc:
            raise RuntimeError(f"oracle_discrete_dynamics rc={rc}")
        return out

    def jacobian(self, rule, diff, Z, t, dt, sig=STATIC_RETURN, want_y=True):
        """Returns (J, Y); J[k] is the n x (n+m) Jacobian of knot k (J stored column-major per knot,
        returned as an (N, n, n+m) array view), Y[k] = f(z_k)."""
        Z, t, dt, N = self._prep(Z, t, dt)
        n, nz = self.n, self.n + self.m
        J = np.empty((N, nz, n))     # column-major n x nz  == C-order (nz, n)
        Y = np.empty((N, n)) if want_y else None
        rc = lib().oracle_jacobian(self._buf, int(rule), int(diff), int(sig), N, _dp(Z), _dp(t), _dp(dt), _dp(J), _dp(Y))
        if rc:
            raise NotImplementedError(f"oracle_jacobian rc={rc}")
        return J.transpose(0, 2, 1), Y

    def dynamics_error(self, rule, Z, t, dt):
        """dynamics_error(model, z2, z1) = discrete_dynamics(model, z1) - state(z2) for consecutive pairs of one trajectory
        (reference src/discrete_dynamics.jl:134-135).  Z (N, n+m) -> (N-1, n)."""
        Z = np.asarray(Z, dtype=np.float64)
        return self.discrete_dynamics(rule, Z[:-1], np.asarray(t)[:-1], np.asarray(dt)[:-1]) - Z[1:, :self.n]

    def dynamics_error_jacobian(self, rule, diff, Z, t, dt):
        """(J2, J1, e): J1 = [A B] at z1, J2 = [-I 0] (reference src/discrete_dynamics.jl:170-177)."""
        Z = np.asarray(Z, dtype=np.float64)
        J1, y = self.jacobian(rule, diff, Z[:-1], np.asarray(t)[:-1], np.asarray(dt)[:-1])
        J2 = np.zeros_like(J1)
        J2[:, np.arange(self.n), np.arange(self.n)] = -1.0
        return J2, J1, y - Z[1:, :self.n]

    def error_expansion(self, rule, Z, t, dt):
        """[G2' A G1 | G2' B] for consecutive pairs (TrajectoryOptimization.jl error_expansion!: D.A = G2'A G1, D.B = G2'B), built from
        this oracle's own ForwardAD Jacobian and errstate_jacobian (reference src/liestate.jl:262-298).  Z (N, n+m) -> (N-1, e, e+m)."""
        Z = np.asarray(Z, dtype=np.float64)
        J, y = self.jacobian(rule, FORWARD_AD, Z[:-1], np.asarray(t)[:-1], np.asarray(dt)[:-1])
        n = self.n
        if self.liestate() is None:          # EuclideanState: G = I
            return J, y
        G = self.errstate_jacobian(Z[:, :n])
        A, B = J[:, :, :n], J[:, :, n:]
        G1, G2 = G[:-1], G[1:]
        Ab = np.einsum("kia,kij,kjb->kab", G2, A, G1)
        Bb = np.einsum("kia,kij->kaj", G2, B)
        return np.concatenate([Ab, Bb], axis=2), y

    def implicit_midpoint(self, Z, t, dt, newton_iters=10, tol=1e-12, want_jacobian=True):
        """DiscretizedDynamics{ImplicitMidpoint}: Newton solve for x_{k+1} (reference src/integration.jl:422-463) and the
        implicit-function-theorem Jacobian -J2[:,1:n] \\ J1 (src/integration.jl:517-538).  Returns (Xn (N,n), J (N,n,n+m) or None, iterations (N,))."""
        Z, t, dt, N = self._prep(Z, t, dt)
        Xn = np.empty((N, self.n))
        J = np.empty((N, self.n + self.m, self.n)) if want_jacobian else None
        its = np.zeros(N, dtype=np.int32)
        rc = lib().oracle_implicit_midpoint(self._buf, N, _dp(Z), _dp(t), _dp(dt), int(newton_iters), C.c_double(tol), _dp(Xn), _dp(J), _ip(its))
        if rc:
            raise RuntimeError(f"rc={rc}")
        return Xn, (None if J is None else J.transpose(0, 2, 1)), its

    def implicit_midpoint_error(self, Z, t, dt):
        """dynamics_error / dynamics_error_jacobian! of ImplicitMidpoint for consecutive pairs (src/integration.jl:631-690): (e, J2, J1)."""
        Z, t, dt, N = self._prep(Z, t, dt)
        nz = self.n + self.m
        E = np.empty((N - 1, self.n)); J2 = np.empty((N - 1, nz, self.n)); J1 = np.empty((N - 1, nz, self.n))
        rc = lib().oracle_implicit_midpoint_error(self._buf, N, _dp(Z), _dp(t), _dp(dt), _dp(E), _dp(J2), _dp(J1))
        if rc:
            raise RuntimeError(f"rc={rc}")
        return E, J2.transpose(0, 2, 1), J1.transpose(0, 2, 1)

    def errstate_jacobian(self, X, clear=True, G0=None):
        """G0: optional (N, n, e) initial content (the reference does not clear G)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        N, ldx = X.shape
        if G0 is None:
            Gc = np.zeros((N, self.e, self.n))
        else:
            Gc = np.ascontiguousarray(np.asarray(G0, dtype=np.float64).transpose(0, 2, 1))
        rc = lib().oracle_errstate_jacobian(self._buf, N, _dp(X), ldx, int(clear), _dp(Gc))
        if rc:
            raise NotImplementedError(f"rc={rc}")
        return Gc.transpose(0, 2, 1)

    def state_diff(self, X, X0, errmap=0):
        """errmap: 0 CayleyMap (the reference's default, src/liestate.jl:27) 1 ExponentialMap 2 MRPMap 3 QuatVecMap 4 IdentityMap"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        N = X.shape[0]
        out = np.empty((N, self.e))
        rc = lib().oracle_state_diff_map(self._buf, N, _dp(X), X.shape[1], _dp(X0), X0.shape[1], int(errmap), _dp(out))
        if rc:
            raise NotImplementedError(f"rc={rc}")
        return out

    def nabla_errstate_jacobian(self, X, Xbar, clear=True):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xb = np.ascontiguousarray(Xbar, dtype=np.float64)
        N = X.shape[0]
        out = np.zeros((N, self.e, self.e))
        rc = lib().oracle_nabla_errstate_jacobian(self._buf, N, _dp(X), X.shape[1], _dp(Xb), Xb.shape[1], int(clear), _dp(out))
        if rc:
            raise NotImplementedError(f"rc={rc}")
        return out.transpose(0, 2, 1)

    def wrenches(self, Z, t):
        """wrenches(model, x, u, t) = [forces; moments] per knot point (src/rigidbody.jl:244-248): (N, 6)."""
        Z, t, _, N = self._prep(Z, t, 0.0)
        out = np.empty((N, 6))
        rc = lib().oracle_wrenches(self._buf, N, _dp(Z), _dp(t), _dp(out))
        if rc:
            raise NotImplementedError(f"rc={rc}")
        return out

    def wrench_jacobian(self, Z, t):
        """d wrenches / d z by forward duals (what src/rigidbody.jl:315-320 is written to return): (N, 6, n+m)."""
        Z, t, _, N = self._prep(Z, t, 0.0)
        out = np.empty((N, self.n + self.m, 6))
        rc = lib().oracle_wrench_jacobian(self._buf, N, _dp(Z), _dp(t), _dp(out))
        if rc:
            raise NotImplementedError(f"rc={rc}")
        return out.transpose(0, 2, 1)

    def rollout(self, rule, Z, t, dt, X0=None, sig=STATIC_RETURN):
        """Z: (M, K, n+m) trajectories; returns a copy with states k>=1 overwritten by the simulation."""
        Z = np.array(Z, dtype=np.float64, order="C")
        M, K, nz = Z.shape
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (M, K)))
        dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (M, K)))
        X0 = None if X0 is None else np.ascontiguousarray(X0, dtype=np.float64)
        rc = lib().oracle_rollout(self._buf, int(rule), int(sig), K, M, _dp(Z), _dp(t), _dp(dt), _dp(X0))
        if rc:
            raise RuntimeError(f"rc={rc}")
        return Z


def liestate_query(rot, P):
    """(length, errstate_dim, [first rot_inds], [first vec_inds]) for LieState{rot,P} (1-based)."""
    P = np.asarray(P, dtype=np.int32)
    ln, ed = C.c_int(), C.c_int()
    rf = np.zeros(8, dtype=np.int32)
    vf = np.zeros(8, dtype=np.int32)
    lib().oracle_liestate_query(int(rot), _ip(P), int(P.size), C.byref(ln), C.byref(ed), _ip(rf), _ip(vf))
    return ln.value, ed.value, [int(v) for v in rf[: P.size - 1]], [int(v) for v in vf[: P.size]]


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


# ---- standard model constructors (parameters of the reference's examples/tests) -----------
def cartpole():            # examples/cartpole.jl:12
    return Model(CARTPOLE, [1.0, 0.2, 0.5, 9.81])


def double_integrator(D):  # test/double_integrator.jl:97
    return Model(DOUBLE_INTEGRATOR, [D])


def linear(A, B):
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    n, m = B.shape
    return Model(LINEAR, np.concatenate([[n, m], A.flatten(order="F"), B.flatten(order="F")]))


def satellite(mass=1.0, J=np.eye(3), frame=FRAME_WORLD, rot=ROT_QUAT):   # examples/single_satellite.jl:33-35 ; rot: RigidBody{R}, test/rigidbody_test.jl:79-98
    return Model(SATELLITE, np.concatenate([[mass], np.asarray(J, float).flatten(order="F")]), frame=frame, rot=rot)


def body(frame=FRAME_WORLD):   # test/rigidbody_test.jl:53-54  (mass 2, J = diag(2,3,1))
    return satellite(2.0, np.diag([2.0, 3.0, 1.0]), frame)


def quadrotor(frame=FRAME_WORLD, rot=ROT_QUAT):   # test/quadrotor.jl:36-46
    J = np.diag([0.0023, 0.0023, 0.004])
    return Model(QUADROTOR, np.concatenate([[0.5], J.flatten(order="F"), [0, 0, -9.81], [0.1750, 1.0, 0.0245]]), frame=frame, rot=rot)


def satellite_pair(J1=np.eye(3), J2=2 * np.eye(3)):   # examples/satellite_pair.jl:42-44
    return Model(SATELLITE_PAIR, np.concatenate([np.asarray(J1, float).flatten(order="F"), np.asarray(J2, float).flatten(order="F")]))


def lie_satellite(Jd=(1.0, 1.0, 1.0)):   # test/liemodel.jl:5
    return Model(LIE_SATELLITE, list(Jd))


def lie_state(rot, P):   # LieState{R,P} (test/liestate.jl:33-68): state_diff / errstate_jacobian / nabla_errstate_jacobian of any partition
    return Model(LIE_STATE, [len(P)] + [int(v) for v in P], rot=rot)


def tv_pendulum(a=9.81, b=2.0, w=3.0, c=0.1):   # test model: thdd = -(a + b cos wt) sin th - c thd + u + b sin wt
    return Model(TV_PENDULUM, [a, b, w, c])


def tv_satellite(mass=2.0, J=np.diag([2.0, 3.0, 1.0]), fd=0.7, td=0.4, w=2.5, frame=FRAME_WORLD):   # test model: the `Body` satellite + a time-dependent disturbance wrench
    return Model(TV_SATELLITE, np.concatenate([[mass], np.asarray(J, float).flatten(order="F"), [fd, td, w]]), frame=frame)
