"""Host-side mirror of RobotDynamics.jl's operator interface for the batched hot path.

Julia is not available in the build image, so the host layer above the C ABI is written in Python; it keeps the
reference's names, argument order and error behaviour so that callers (and the parity tests) read like the
reference's own code.  Julia's trailing `!` becomes a trailing underscore (`jacobian!` -> `jacobian_`).
Every function here ends in ONE call into librdhip (include/rdhip.h) -- there is no Python or CPU compute path.

Array conventions (row k = knot point k, i.e. the C-order view of Julia's column-major arrays):
    Z.data  (N, n+m)        == Julia (n+m) x N         t, dt  (N,) float64
    Y / Xn  (N, n)          == Julia n x N
    J       (N, n, n+m) view whose memory is (N, n+m, n) C-order == Julia n x (n+m) x N   (see `alloc_jacobians`)
    G       (N, n, e)   view whose memory is (N, e, n)   C-order == Julia n x e x N
numpy arrays are host memory (the library stages them, call is synchronous); torch CUDA tensors are device
memory (call is asynchronous on torch's current stream).

Reference citations are file:line in RobotDynamics.jl v0.4.8.
"""
import ctypes as C

import numpy as np

from . import _lib as L

try:  # torch is plumbing (device memory + streams), never the compute path
    import torch
except Exception:  # pragma: no cover
    torch = None


# =============================================================================================================
# traits: FunctionSignature / DiffMethod / QuadratureRule / StateVectorType / FunctionInputs
# (src/functionbase.jl:78-80, :109-120, :172-206 ; src/statevectortype.jl:63,73 ; src/discretized_dynamics.jl:16-99)
# =============================================================================================================
class _Trait:
    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self))

    def __repr__(self):
        return type(self).__name__ + "()"


class FunctionSignature(_Trait):
    pass


class InPlace(FunctionSignature):
    pass


class StaticReturn(FunctionSignature):
    pass


class DiffMethod(_Trait):
    pass


class ForwardAD(DiffMethod):
    code = L.FORWARD_AD


class FiniteDifference(DiffMethod):
    code = L.FINITE_DIFF


class UserDefined(DiffMethod):
    code = L.USER_DEFINED


class ImplicitFunctionTheorem(DiffMethod):
    """ImplicitFunctionTheorem(diff): the DiffMethod of implicit models (src/functionbase.jl:109-120, src/discretized_dynamics.jl:230-251):
    solve for the next state, then J = -J2[:,1:n] \\ J1 with the continuous Jacobians from `diff` (ForwardAD is built)."""

    def __init__(self, diff):
        if isinstance(diff, ImplicitFunctionTheorem):   # src/functionbase.jl:113-116
            raise ValueError("DiffMethod for ImplicitFunctionTheorem cannot also be ImplicitFunctionTheorem.")
        self.diff = diff


class StateVectorType(_Trait):
    pass


class EuclideanState(StateVectorType):
    pass


class RotationState(StateVectorType):
    pass


class FunctionInputs(_Trait):
    pass


class StateControl(FunctionInputs):
    pass


class StateOnly(FunctionInputs):
    pass


class ControlOnly(FunctionInputs):
    pass


class QuadratureRule:
    code = None

    def __init__(self, n=None, m=None):   # RK4(n, m) in the reference allocates caches; nothing to allocate here
        pass


class Explicit(QuadratureRule):
    pass


class Euler(Explicit):
    code = L.EULER


class RK2(Explicit):       # v0.3 explicit midpoint
    code = L.RK2


class RK3(Explicit):
    code = L.RK3


class RK4(Explicit):
    code = L.RK4


class Implicit(QuadratureRule):
    pass


class ImplicitMidpoint(Implicit):   # src/integration.jl:620-629: x1 + h f((x1+x2)/2, u1, t + h/2) - x2 = 0, Newton solve
    code = L.IMPLICIT_MIDPOINT


def _rule_code(Q):
    if isinstance(Q, type) and issubclass(Q, QuadratureRule) or isinstance(Q, QuadratureRule):
        if Q.code is None:
            raise NotImplementedError(f"{Q} has no batched kernel")
        return Q.code
    raise ValueError(f"not a QuadratureRule: {Q!r}")


# rotation parameterisations (Rotations.jl types): the class is the tag the kernels are selected by; an instance is a host-side
# value with the conversions the state helpers need (quaternion <-> MRP p = v/(1+w) <-> Rodrigues parameter g = v/w)
class _Rotation:
    def params(self):
        return self._p.copy()

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(float(v)) for v in self._p)})"


class QuatRotation(_Rotation):
    """QuatRotation(w, x, y, z, normalize=True) or QuatRotation(other_rotation)."""
    code, nparams = L.ROT_QUAT, 4

    def __init__(self, *a):
        if len(a) == 1 and isinstance(a[0], _Rotation):
            self._p = a[0].quat()
            return
        normalize = True
        if len(a) == 5:
            a, normalize = a[:4], bool(a[4])
        elif len(a) == 2 and np.ndim(a[0]) == 1:
            a, normalize = tuple(a[0]), bool(a[1])
        elif len(a) == 1:
            a = tuple(a[0])
        if len(a) != 4:
            raise TypeError("QuatRotation(w, x, y, z[, normalize])")
        q = np.array(a, dtype=np.float64)
        self._p = q / np.linalg.norm(q) if normalize else q

    def quat(self):
        return self._p.copy()


UnitQuaternion = QuatRotation   # pre-1.0 Rotations.jl name


class MRP(_Rotation):
    code, nparams = L.ROT_MRP, 3

    def __init__(self, *a):
        if len(a) == 1 and isinstance(a[0], _Rotation):
            q = a[0].quat(); q = q / np.linalg.norm(q)
            self._p = q[1:] / (1.0 + q[0])
            return
        a = tuple(a[0]) if len(a) == 1 else a
        if len(a) != 3:
            raise TypeError("MRP(x, y, z)")
        self._p = np.array(a, dtype=np.float64)

    def quat(self):
        n2 = float(self._p @ self._p)
        return np.concatenate([[1.0 - n2], 2.0 * self._p]) / (1.0 + n2)


class RodriguesParam(_Rotation):
    code, nparams = L.ROT_RP, 3

    def __init__(self, *a):
        if len(a) == 1 and isinstance(a[0], _Rotation):
            q = a[0].quat()
            self._p = q[1:] / q[0]
            return
        a = tuple(a[0]) if len(a) == 1 else a
        if len(a) != 3:
            raise TypeError("RodriguesParam(x, y, z)")
        self._p = np.array(a, dtype=np.float64)

    def quat(self):
        return np.concatenate([[1.0], self._p]) / np.sqrt(1.0 + float(self._p @ self._p))


def params(R):   # Rotations.params(::Type) as extended at src/liestate.jl:42-46; Rotations.params(rotation) for a value
    return R.params() if isinstance(R, _Rotation) else R.nparams


# =============================================================================================================
# array plumbing
# =============================================================================================================
def _is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


class _Arg:
    """A caller array resolved to (pointer, memspace, dtype code)."""
    __slots__ = ("ptr", "space", "dtype", "device", "keep")

    def __init__(self, a, name, want_dtype=None):
        if a is None:
            self.ptr, self.space, self.dtype, self.device, self.keep = None, None, None, None, None
            return
        if _is_torch(a):
            # logical views (transposes) are fine as long as the underlying memory is the dense slab
            if a.dtype not in (torch.float64, torch.float32):
                raise ValueError(f"{name}: dtype must be float64 or float32")
            self.ptr = a.data_ptr()
            self.space = L.DEVICE if a.is_cuda else L.HOST
            self.dtype = L.F64 if a.dtype == torch.float64 else L.F32
            self.device = a.device.index if a.is_cuda else None
        elif isinstance(a, np.ndarray):
            if a.dtype not in (np.float64, np.float32):
                raise ValueError(f"{name}: dtype must be float64 or float32")
            self.ptr = a.ctypes.data
            self.space, self.device = L.HOST, None
            self.dtype = L.F64 if a.dtype == np.float64 else L.F32
        else:
            raise ValueError(f"{name}: expected a numpy array or torch tensor, got {type(a)}")
        self.keep = a
        if want_dtype is not None and self.dtype != want_dtype:
            raise ValueError(f"{name}: dtype does not match the trajectory data")


def _dense(a, shape, name, transposed_last2=False):
    """Check that `a` has logical `shape` and that its memory is the dense slab the C ABI expects.
    transposed_last2: logical (N, r, c) must be stored as C-order (N, c, r) (Julia column-major r x c x N)."""
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")   # DimensionMismatch
    if any(s == 0 for s in shape):
        return a
    if _is_torch(a):
        st = tuple(a.stride())
        item = 1
    else:
        st = tuple(s // a.itemsize for s in a.strides)
        item = 1
    nd = len(shape)
    if transposed_last2:
        N, r, c = shape
        want = (r * c, 1, r)
    else:
        want = []
        acc = item
        for s in reversed(shape):
            want.append(acc)
            acc *= s
        want = tuple(reversed(want))
    for i in range(nd):
        if shape[i] > 1 and st[i] != want[i]:
            hint = " (allocate it with rdhip.alloc_jacobians / alloc_errstate_jacobians)" if transposed_last2 else " (must be C-contiguous)"
            raise ValueError(f"{name}: memory layout {st} does not match the packed layout {want}{hint}")
    return a


def _resolve_ctx(args):
    """Pick memspace/device/stream from the data arguments; all must live in the same space."""
    spaces = {a.space for a in args if a.ptr is not None}
    if len(spaces) != 1:
        raise ValueError("all arrays of one call must be either host (numpy) or device (torch.cuda) arrays")
    space = spaces.pop()
    if space == L.DEVICE:
        devs = {a.device for a in args if a.ptr is not None}
        if len(devs) != 1:
            raise ValueError("arrays live on different devices")
        dev = devs.pop()
        stream = torch.cuda.current_stream(dev).cuda_stream
        return L.context_for(dev, stream), space
    dev = torch.cuda.current_device() if (torch is not None and torch.cuda.is_available()) else 0
    return L.context_for(dev, None), space


def _empty_like_data(data, shape):
    if _is_torch(data):
        return torch.empty(shape, dtype=data.dtype, device=data.device)
    return np.empty(shape, dtype=data.dtype)


def alloc_jacobians(model, N, like=None, dtype=None):
    """(N, n, n+m) Jacobian stack with the reference's column-major-per-knot memory (src/jacobian.jl:26-37)."""
    n, m = state_dim(model), control_dim(model)
    base = _alloc((N, n + m, n), like, dtype)
    return base.transpose(1, 2) if _is_torch(base) else base.transpose(0, 2, 1)


def alloc_errstate_jacobians(model, N, like=None, dtype=None):
    n, e = state_dim(model), errstate_dim(model)
    base = _alloc((N, e, n), like, dtype, zero=True)
    return base.transpose(1, 2) if _is_torch(base) else base.transpose(0, 2, 1)


def _alloc(shape, like, dtype, zero=False):
    if like is not None and _is_torch(like):
        f = torch.zeros if zero else torch.empty
        return f(shape, dtype=dtype or like.dtype, device=like.device)
    f = np.zeros if zero else np.empty
    return f(shape, dtype=dtype or (like.dtype if like is not None else np.float64))


# =============================================================================================================
# knot points and packed trajectories  (src/knotpoint.jl, src/trajectories.jl)
# =============================================================================================================
class AbstractKnotPoint:
    """z = [x; u], t, dt.  dt == 0 marks a terminal knot point whose control reads as zeros
    (src/knotpoint.jl:67,123)."""

    def __init__(self, *args):
        # KnotPoint(x, u, t, dt) | KnotPoint(n, m, z, t, dt)     (src/knotpoint.jl:154-183)
        if len(args) == 4:
            x, u, t, dt = args
            x, u = np.asarray(x), np.asarray(u)
            both32 = x.dtype == np.float32 and u.dtype == np.float32
            z = np.concatenate([x, u]).astype(np.float32 if both32 else np.float64)
            n, m = x.shape[0], u.shape[0]
        elif len(args) == 5:
            n, m, z, t, dt = args
            z = z if isinstance(z, np.ndarray) else np.asarray(z, dtype=np.float64)
        elif len(args) == 3:
            raise TypeError("KnotPoint(x, u, dt) is the v0.3 constructor; v0.4 needs KnotPoint(x, u, t, dt) (src/knotpoint.jl:178-183)")
        else:
            raise TypeError("KnotPoint(x, u, t, dt) or KnotPoint(n, m, z, t, dt)")
        assert n > 0 and m > 0, "state and control dimensions must be positive"
        assert n + m == len(z), "n + m must equal length(z)"
        self.z, self.t, self.dt, self.n, self.m = z, float(t), float(dt), int(n), int(m)

    # AbstractVector interface
    def __len__(self):
        return self.n + self.m

    def __getitem__(self, i):
        return self.z[i]

    def __setitem__(self, i, v):
        self.z[i] = v

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.z, dtype=dtype)


class KnotPoint(AbstractKnotPoint):
    pass


class StaticKnotPoint(AbstractKnotPoint):
    """StaticKnotPoint(z, v): same t, dt, new data (src/knotpoint.jl:225-227)."""

    def __init__(self, *args):
        if len(args) == 2 and isinstance(args[0], AbstractKnotPoint):
            z, v = args
            super().__init__(z.n, z.m, np.asarray(v), z.t, z.dt)
        elif len(args) == 1 and isinstance(args[0], AbstractKnotPoint):
            z = args[0]
            super().__init__(z.n, z.m, z.z, z.t, z.dt)
        else:
            super().__init__(*args)


def is_terminal(z):
    return z.dt == 0.0


def state(z):
    return z.z[: z.n]


def control(z):
    return np.zeros(z.m, dtype=z.z.dtype) if is_terminal(z) else z.z[z.n:]


def getdata(z):
    return z.data if isinstance(z, PackedTrajectory) else z.z


def getparams(z):
    return {"t": z.t, "dt": z.dt}


def time(z):
    return z.t


def timestep(z):
    return z.dt


def setdata_(z, v):
    z.z[:] = v


def setstate_(z, x):
    z.z[: z.n] = x


def setcontrol_(z, u):
    z.z[z.n:] = u


def settime_(z, t):
    z.t = float(t)


def settimestep_(z, dt):
    z.dt = float(dt)


class PackedTrajectory:
    """Dense replacement for SampledTrajectory = Vector{KnotPoint} (src/trajectories.jl:40-50): one (N, n+m)
    array + t, dt vectors.  `Z[k]` is a KnotPoint VIEW into row k (host trajectories)."""

    def __init__(self, n, m, data, t, dt):
        self.n, self.m = int(n), int(m)
        N = data.shape[0]
        _dense(data, (N, n + m), "data")
        self.data = data
        if _is_torch(data):
            self.t = torch.as_tensor(t, dtype=torch.float64, device=data.device).expand(N).contiguous()
            self.dt = torch.as_tensor(dt, dtype=torch.float64, device=data.device).expand(N).contiguous()
        else:
            self.t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (N,)))
            self.dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (N,)))

    # ---- constructors mirroring SampledTrajectory{n,m}(X, U; dt=, tf=) (src/trajectories.jl:92-109) --------------
    @classmethod
    def from_states_controls(cls, X, U, dt=None, tf=None, t0=0.0, dtype=np.float64):
        X, U = np.asarray(X, dtype=dtype), np.asarray(U, dtype=dtype)
        N, n = X.shape
        m = U.shape[1]
        times = _gettimeinfo(t0=t0, tf=tf, dt=dt, N=N)
        dts = np.empty(N)
        dts[:-1] = np.diff(times)
        data = np.zeros((N, n + m), dtype=dtype)
        data[:, :n] = X
        if U.shape[0] == N:          # terminal control present: dt = Inf (first-order hold convention)
            data[:, n:] = U
            dts[-1] = np.inf
        elif U.shape[0] == N - 1:
            data[:-1, n:] = U
            dts[-1] = 0.0
        else:
            raise ValueError("U must have N or N-1 rows")
        return cls(n, m, data, times, dts)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, k):
        if _is_torch(self.data):
            raise TypeError("indexing knot points needs a host trajectory; use .cpu() first")
        if k < 0:
            k += len(self)
        return KnotPoint(self.n, self.m, self.data[k], self.t[k], self.dt[k])

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def to(self, device):
        if torch is None:
            raise RuntimeError("torch is required for device trajectories")
        d = torch.as_tensor(np.array(self.data) if not _is_torch(self.data) else self.data).to(device)
        return PackedTrajectory(self.n, self.m, d, torch.as_tensor(np.array(self.t) if not _is_torch(self.t) else self.t).to(device),
                                torch.as_tensor(np.array(self.dt) if not _is_torch(self.dt) else self.dt).to(device))

    def cpu(self):
        if not _is_torch(self.data):
            return self
        return PackedTrajectory(self.n, self.m, self.data.cpu().numpy(), self.t.cpu().numpy(), self.dt.cpu().numpy())


SampledTrajectory = PackedTrajectory
Traj = PackedTrajectory   # v0.3 name (src/deprecate.jl:6)


def _gettimeinfo(t0=0.0, tf=None, dt=None, N=0):   # src/trajectories.jl:355-400
    if dt is not None and np.ndim(dt) > 0:
        dt = np.asarray(dt, dtype=np.float64)
        assert np.all(dt > 0), "All time steps must be positive"
        if N == 0:
            N = len(dt) + 1
        assert len(dt) == N - 1, f"Time step vector must have length of N-1, got {len(dt)} with N={N}."
        # (the reference returns [t0; cumsum(dt)] here, src/trajectories.jl:399 -- times that do NOT start from t0 when t0 != 0, which its own assertion on
        # the elapsed time two lines above contradicts; this mirror returns t0 + cumsum(dt), the times the docstring of gettimeinfo describes.  For
        # t0 = 0, the only case the reference's callers and tests use, the two agree bit for bit.)
        return np.concatenate([[t0], t0 + np.cumsum(dt)])
    missing = (tf is None) + (dt is None) + (N == 0)
    if missing > 1:
        raise ValueError("Must specify at least two of the following: dt, tf, N")
    if N == 0:
        N = int(round(tf / dt + 1))
        dt = (tf - t0) / (N - 1)
    if tf is not None and dt is None:
        dt = tf / (N - 1)
    if dt is not None and tf is None:
        tf = dt * (N - 1)
    assert np.isclose(tf, dt * (N - 1)), "Inconsistent time step and final time."
    return np.linspace(t0, tf, N)


def states(Z):
    return Z.data[:, : Z.n]


def controls(Z):
    return Z.data[:, Z.n:]


def gettimes(Z):
    return Z.t


def has_terminal_control(Z):
    return float(Z.dt[-1]) != 0.0


# =============================================================================================================
# models  (AbstractFunction > AbstractModel > ContinuousDynamics / DiscreteDynamics)
# =============================================================================================================
class AbstractFunction:
    pass


class AbstractModel(AbstractFunction):
    pass


class ContinuousDynamics(AbstractModel):
    """A registered continuous model.  Subclasses define `_kind`, `_params()`; user-defined Python dynamics cannot
    run on the GPU (no kernel DSL by design), so only the registry models are evaluable."""
    _kind = None
    rotation = QuatRotation
    velocity_frame = "world"

    def _params(self):
        raise NotImplementedError

    @property
    def _handle(self):
        h = getattr(self, "_h", None)
        if h is None:
            if self._kind is None:
                raise NotImplementedError(f"User-defined static return function not implemented for {type(self).__name__}")   # src/functionbase.jl:231-232
            p = np.ascontiguousarray(np.asarray(self._params(), dtype=np.float64))
            h = C.c_void_p()
            frame = L.FRAME_BODY if self.velocity_frame == "body" else L.FRAME_WORLD
            L.check(L.lib().rdhip_model_create(self._kind, self.rotation.code, frame, p.ctypes.data_as(C.POINTER(C.c_double)), p.size, C.byref(h)))
            self._h = h
        return h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None:
            try:
                L.lib().rdhip_model_destroy(h)
            except Exception:
                pass

    def _dims4(self):
        n, m, p, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().rdhip_model_dims(self._handle, C.byref(n), C.byref(m), C.byref(p), C.byref(e)))
        return n.value, m.value, p.value, e.value


class DiscreteDynamics(AbstractModel):
    pass


class _DiscretizedMeta(type):
    def __getitem__(cls, Q):          # DiscretizedDynamics[RK4](model)  ==  DiscretizedDynamics{RK4}(model)
        return lambda model: cls(model, Q)


class DiscretizedDynamics(DiscreteDynamics, metaclass=_DiscretizedMeta):
    """DiscretizedDynamics(continuous_dynamics, integrator) (src/discretized_dynamics.jl:169-184)."""

    def __init__(self, continuous_dynamics, integrator):
        if not isinstance(continuous_dynamics, ContinuousDynamics):
            raise TypeError("continuous_dynamics must be a ContinuousDynamics model")
        self.continuous_dynamics = continuous_dynamics
        self.integrator = integrator() if isinstance(integrator, type) else integrator
        _rule_code(self.integrator)

    def __copy__(self):               # Base.copy(model) is shallow (src/discretized_dynamics.jl:194-196)
        return DiscretizedDynamics(self.continuous_dynamics, self.integrator)


def integration(model):
    return model.integrator


class CustomModel(ContinuousDynamics):
    """A user-defined ContinuousDynamics model -- the reference's `dynamics(model, x, u)` method (src/dynamics.jl:23-45) -- given as
    the body of a HIP device function written on a generic scalar type S:

        pendulum = RD.CustomModel(2, 1, params=(9.81, 1.0, 0.1), dynamics=
            "xdot[0] = x[1];"
            "xdot[1] = (u[0] - x[1] * p[2]) / (p[1] * p[1]) - sin_(x[0]) * (p[0] / p[1]);")

    `x`, `u`, `xdot` are arrays of S, `p[i]` are the parameters (type T = float/double), `t` (type T) is the time the dynamics are
    evaluated at -- `dynamics(model, x, u, t)`: the knot point's time, or the stage times t, t + h/2, t + h inside the Runge-Kutta rules
    (src/integration.jl:130-135,280-286); use sin_, cos_, sqrt_, max0_ and the arithmetic operators (they are defined for T and for
    forward duals, csrc/scalar.hpp).  The library JIT-compiles its kernel
    templates for this model with hipRTC on first use (a few seconds per (operation, rule, dtype)), after which every operator
    of the compiled-in models works: dynamics, discrete_dynamics, jacobian! with ForwardAD / FiniteDifference / UserDefined,
    rollout!, dynamics_error, and -- with `liestate=(R, P)` as in LieState{R,P} -- the error-state operators."""
    _kind = L.MODEL_CUSTOM
    _OPS = {"evaluate": 0, "rollout": 1, "ForwardAD": 2, "FiniteDifference": 3, "UserDefined": 4, "error_expansion": 12,
            "implicit_solve": 13, "implicit_jacobian": 16}

    def __init__(self, n, m, dynamics, params=(), preamble="", liestate=None, jacobian=None):
        self.n, self.m = int(n), int(m)
        self.dynamics_source, self.preamble = str(dynamics), str(preamble)
        # `jacobian`: the user's own jacobian!(model, J, xdot, x, u, t) (src/dynamics.jl:115-116) as the body of
        #     void jac(const T* x, const T* u, T t, T* J) const        -- J: n x (n+m), column-major (J[i + n * j]), zero on entry
        # in plain T arithmetic (sin, cos, sqrt, ...: the HIP device functions).  UserDefined() then evaluates THESE formulas: directly for the
        # continuous model, and through the integrators' chain rule on the stage Jacobians (src/integration.jl:85-93, :149-177, :302-337) for
        # DiscretizedDynamics of models with n + m <= 10.  Without it (and for bigger discretized models) UserDefined() is the matrix-free chain
        # rule on forward duals of `dynamics`.
        self.jacobian_source = None if jacobian is None else str(jacobian)
        self.params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).ravel())
        if liestate is not None:
            R, P = (liestate.R, liestate.P) if isinstance(liestate, LieState) else liestate
            self.rotation, self._P = R, tuple(int(v) for v in P)
        else:
            self._P = ()

    def _params(self):
        return self.params

    @property
    def _handle(self):
        h = getattr(self, "_h", None)
        if h is None:
            h = C.c_void_p()
            P = (C.c_int * max(len(self._P), 1))(*self._P)
            p = self.params
            L.check(L.lib().rdhip_model_create_custom(self.n, self.m, self.dynamics_source.encode(), self.preamble.encode(),
                                                      p.ctypes.data_as(C.POINTER(C.c_double)), p.size, self.rotation.code, P, len(self._P), C.byref(h)))
            if self.jacobian_source is not None:
                L.check(L.lib().rdhip_model_set_user_jacobian(h, self.jacobian_source.encode()))
            self._h = h
        return h

    def balance_columns(self):
        """n+m > 10: cost-balance the Jacobian columns over the compute waves (each column is JIT-compiled alone and its code size is
        the cost; ~0.5 s per column).  Call before the first Jacobian."""
        L.check(L.lib().rdhip_model_balance_columns(self._handle))
        return self

    def set_cascade(self, state_block, control_block):
        """Declare the block structure of the dynamics (what the compiled-in rigid bodies state as w | q | r, v, src/rigidbody.jl:213-236): the time
        derivative of block b reads blocks 0..b only; control_block[j] = the first block whose derivative reads control j.  The model then gets
        the kernels that exploit it (FiniteDifference along the cascade, rollout_ with one wavefront per block, block-by-block Newton for
        ImplicitMidpoint); the declaration is verified against the Jacobian pattern on first GPU use (ValueError names the offending entry)."""
        sb = np.ascontiguousarray(np.asarray(state_block, dtype=np.int32).ravel())
        cb = np.ascontiguousarray(np.asarray(control_block, dtype=np.int32).ravel())
        if sb.size != self.n or cb.size != self.m:
            raise ValueError(f"set_cascade: need {self.n} state blocks and {self.m} control blocks")
        nb = int(max(sb.max(initial=0), cb.max(initial=0))) + 1
        L.check(L.lib().rdhip_model_set_cascade(self._handle, nb, sb.ctypes.data_as(C.POINTER(C.c_int)), cb.ctypes.data_as(C.POINTER(C.c_int))))
        self.cascade = (tuple(int(v) for v in sb), tuple(int(v) for v in cb))
        return self

    def precompile(self, what="ForwardAD", rule=RK4, dtype=np.float64):
        """JIT-compile one kernel now (works without a GPU); raises ValueError with the compiler log if the source does not compile."""
        code = L.CONTINUOUS if rule is None else _rule_code(rule() if isinstance(rule, type) else rule)
        L.check(L.lib().rdhip_model_precompile(self._handle, self._OPS[what], code, L.F64 if np.dtype(dtype) == np.float64 else L.F32))


class CustomDiscreteModel(DiscreteDynamics):
    """A user-defined DiscreteDynamics model -- the reference's `discrete_dynamics(model, x, u, t, dt)` method on a `DiscreteDynamics` subtype
    (src/discrete_dynamics.jl:1-108; test/discrete_dynamics_test.jl:5-63 is its discrete double integrator) -- given as the body of

        template <class S> void step(const S* x, const S* u, T t, T h, S* xn) const      # p[i]: the parameters; t, h: time and time step

        di = RD.CustomDiscreteModel(4, 2, step="xn[0] = x[0] + x[2] * h; xn[1] = x[1] + x[3] * h; xn[2] = x[2] + u[0] * h; xn[3] = x[3] + u[1] * h;",
                                    jacobian="J[0] = 1; J[8] = h; J[5] = 1; J[13] = h; J[10] = 1; J[18] = h; J[15] = 1; J[23] = h;")

    and, optionally, its own jacobian!(model, J, xn, x, u, t, dt) (src/discrete_dynamics.jl:111-113) as the body of
    `void jac(const T* x, const T* u, T t, T h, T* J) const` (J: n x (n+m) column-major, zero on entry), which is what UserDefined() evaluates
    (NotImplementedError without it, like the reference).  evaluate / discrete_dynamics, jacobian! with ForwardAD (the default, as `@autodiff`
    gives a DiscreteDynamics model) / FiniteDifference / UserDefined, rollout!, dynamics_error and dynamics_error_jacobian! (J2 = [-I 0]) and, with
    `liestate`, the error-state operators work as for a DiscretizedDynamics model; there is no quadrature rule and no `dynamics`."""
    _kind = L.MODEL_CUSTOM
    _OPS = {"evaluate": 0, "rollout": 18, "ForwardAD": 2, "FiniteDifference": 3, "UserDefined": 19, "error_expansion": 12}
    rotation = QuatRotation
    velocity_frame = "world"

    def __init__(self, n, m, step, params=(), preamble="", liestate=None, jacobian=None):
        self.n, self.m = int(n), int(m)
        self.step_source, self.preamble = str(step), str(preamble)
        self.jacobian_source = None if jacobian is None else str(jacobian)
        self.params = np.ascontiguousarray(np.asarray(params, dtype=np.float64).ravel())
        if liestate is not None:
            R, P = (liestate.R, liestate.P) if isinstance(liestate, LieState) else liestate
            self.rotation, self._P = R, tuple(int(v) for v in P)
        else:
            self._P = ()

    def _params(self):
        return self.params

    @property
    def _handle(self):
        h = getattr(self, "_h", None)
        if h is None:
            h = C.c_void_p()
            P = (C.c_int * max(len(self._P), 1))(*self._P)
            p = self.params
            jac = None if self.jacobian_source is None else self.jacobian_source.encode()
            L.check(L.lib().rdhip_model_create_custom_discrete(self.n, self.m, self.step_source.encode(), jac, self.preamble.encode(),
                                                               p.ctypes.data_as(C.POINTER(C.c_double)), p.size, self.rotation.code, P, len(self._P), C.byref(h)))
            self._h = h
        return h

    __del__ = ContinuousDynamics.__del__
    _dims4 = ContinuousDynamics._dims4

    def precompile(self, what="ForwardAD", dtype=np.float64):
        """JIT-compile one kernel now (works without a GPU); raises ValueError with the compiler log if the source does not compile."""
        L.check(L.lib().rdhip_model_precompile(self._handle, self._OPS[what], L.DISCRETE, L.F64 if np.dtype(dtype) == np.float64 else L.F32))


class LieGroupModel(ContinuousDynamics):
    pass


def rigid_body_source(forces, moments, bodyframe=False):
    """The HIP body of `dynamics(model::RigidBody{QuatRotation}, x, u, t)` (src/rigidbody.jl:213-236) around user wrenches -- the reference's
    extension points `forces(model, x, u)` (world frame) and `moments(model, x, u)` (body frame), src/rigidbody.jl:244-258, given as code that fills
    `F[0..2]` / `tau[0..2]` and may read `q` (the attitude quaternion, 4 values), `r`, `v`, `w`, `x`, `u`, `t` and the user parameters `P[i]`
    (= p[19 + i]).  Parameter layout of the generated model: p[0] = mass, p[1..9] = inertia (column-major), p[10..18] = its inverse."""
    lin = ("for (int i = 0; i < 3; ++i) { xdot[i] = v[i]; xdot[7 + i] = F[i] * im; }" if not bodyframe else
           "quat_rotate(q, v, xdot);\n    { S Fm[3] = {F[0] * im, F[1] * im, F[2] * im}; S a[3]; quat_rotate_inv(q, Fm, a); S cv[3]; cross3(w, v, cv);\n"
           "      for (int i = 0; i < 3; ++i) xdot[7 + i] = a[i] - cv[i]; }")
    return f"""
    const S* r = x; const S* q = x + 3; const S* v = x + 7; const S* w = x + 10; const T* P = p + 19;
    (void)r; (void)P;
    S F[3], tau[3];
    {{ {forces} }}
    {{ {moments} }}
    quat_kinematics(q, w, xdot + 3);
    const T im = T(1) / p[0];
    {lin}
    S Jw[3]; mat3vec(p + 1, w, Jw);
    S c[3]; cross3(w, Jw, c);
    const S tt[3] = {{tau[0] - c[0], tau[1] - c[1], tau[2] - c[2]}};
    mat3vec(p + 10, tt, xdot + 10);
"""


def wrench_source(forces, moments):
    """The HIP body of a 13-output helper model whose first six outputs are wrenches(model, x, u, t) = [forces; moments] (src/rigidbody.jl:244-248) of a
    CustomRigidBody (same snippets, same names as rigid_body_source); the other outputs are zero."""
    return f"""
    const S* r = x; const S* q = x + 3; const S* v = x + 7; const S* w = x + 10; const T* P = p + 19;
    (void)r; (void)P; (void)q; (void)v; (void)w;
    S F[3], tau[3];
    {{ {forces} }}
    {{ {moments} }}
    for (int i = 0; i < 3; ++i) {{ xdot[i] = F[i]; xdot[3 + i] = tau[i]; }}
    for (int i = 6; i < 13; ++i) xdot[i] = S(T(0));
"""


def CustomRigidBody(m, forces, moments, mass=1.0, J=None, params=(), bodyframe=False, preamble="", cascade=True, force_only_controls=()):
    """A user-defined RigidBody{QuatRotation} for the GPU: what `struct MyBody <: RigidBody{R}` + `forces` / `moments` methods are in the reference
    (src/rigidbody.jl:244-258; examples/single_satellite.jl:18-27), with the wrenches given as HIP source (see rigid_body_source).  `m` controls,
    LieState{QuatRotation,(3,6)}.  cascade=True declares the block structure w | q | r, v (valid when the moments read neither q nor r, v and the
    forces read neither r nor v, as for the reference's satellite and quadrotor; verified on first GPU use -- pass cascade=False otherwise);
    `force_only_controls`: controls that enter the forces only (they then belong to the r, v block)."""
    Jm = np.eye(3) if J is None else np.asarray(J, dtype=np.float64)
    p = np.concatenate([[float(mass)], Jm.flatten(order="F"), np.linalg.inv(Jm).flatten(order="F"), np.asarray(params, dtype=np.float64).ravel()])
    model = CustomModel(13, int(m), rigid_body_source(forces, moments, bodyframe), params=p, preamble=preamble, liestate=(QuatRotation, (3, 6)))
    model.velocity_frame = "body" if bodyframe else "world"
    model._wrench_parts = (forces, moments, p, preamble, int(m))       # wrenches / wrench_jacobian of a custom body: a helper model built from the same snippets
    if cascade:
        fo = {int(j) for j in force_only_controls}
        model.set_cascade([2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 0, 0, 0], [2 if j in fo else 0 for j in range(int(m))])
    return model


class Cartpole(ContinuousDynamics):          # examples/cartpole.jl:5-12 ; test/cartpole_model.jl:2-9
    _kind = L.MODEL_CARTPOLE

    def __init__(self, mc=1.0, mp=0.2, l=0.5, g=9.81):
        self.mc, self.mp, self.l, self.g = mc, mp, l, g

    def _params(self):
        return [self.mc, self.mp, self.l, self.g]


class DoubleIntegrator(ContinuousDynamics):  # test/double_integrator.jl:97-99  (DI{D})
    _kind = L.MODEL_DOUBLE_INTEGRATOR

    def __init__(self, D):
        self.D = int(D)

    def _params(self):
        return [self.D]


class LinearModel(ContinuousDynamics):       # xdot = A x + B u
    _kind = L.MODEL_LINEAR

    def __init__(self, A, B):
        self.A, self.B = np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64)

    def _params(self):
        n, m = self.B.shape
        return np.concatenate([[n, m], self.A.flatten(order="F"), self.B.flatten(order="F")])


class RigidBody(LieGroupModel):
    """abstract type RigidBody{R} <: LieGroupModel (src/rigidbody.jl:46); LieState = (R, (3, 6)) (:48)."""

    def __init__(self, R=QuatRotation, bodyframe=False):
        self.rotation = R
        self.velocity_frame = "body" if bodyframe else "world"


class Satellite(RigidBody):                  # examples/single_satellite.jl:7-35 ; `Body` of test/rigidbody_test.jl:23-54
    _kind = L.MODEL_SATELLITE

    def __init__(self, mass=1.0, J=None, R=QuatRotation, bodyframe=False):
        super().__init__(R, bodyframe)
        self.mass = float(mass)
        self.J = np.eye(3) if J is None else np.asarray(J, dtype=np.float64)

    def _params(self):
        return np.concatenate([[self.mass], self.J.flatten(order="F")])


def Body(R=QuatRotation, bodyframe=False):   # test/rigidbody_test.jl:53-54: mass 2, J = Diagonal(2, 3, 1)
    return Satellite(2.0, np.diag([2.0, 3.0, 1.0]), R, bodyframe)


class Quadrotor(RigidBody):                  # test/quadrotor.jl:21-46
    _kind = L.MODEL_QUADROTOR

    def __init__(self, R=QuatRotation, mass=0.5, J=None, gravity=(0.0, 0.0, -9.81), motor_dist=0.1750, kf=1.0, km=0.0245, bodyframe=False):
        super().__init__(R, bodyframe)
        self.mass = float(mass)
        self.J = np.diag([0.0023, 0.0023, 0.004]) if J is None else np.asarray(J, dtype=np.float64)
        self.gravity = np.asarray(gravity, dtype=np.float64)
        self.motor_dist, self.kf, self.km = float(motor_dist), float(kf), float(km)

    def _params(self):
        return np.concatenate([[self.mass], self.J.flatten(order="F"), self.gravity, [self.motor_dist, self.kf, self.km]])


class SatellitePair(LieGroupModel):          # examples/satellite_pair.jl:8-44
    _kind = L.MODEL_SATELLITE_PAIR

    def __init__(self, J1=None, J2=None, R=QuatRotation):
        self.rotation = R
        self.J1 = np.eye(3) if J1 is None else np.asarray(J1, dtype=np.float64)
        self.J2 = 2 * np.eye(3) if J2 is None else np.asarray(J2, dtype=np.float64)

    def _params(self):
        return np.concatenate([self.J1.flatten(order="F"), self.J2.flatten(order="F")])


class LieSatellite(LieGroupModel):           # `Satellite` of test/liemodel.jl:1-22 (x = [w; q])
    _kind = L.MODEL_LIE_SATELLITE

    def __init__(self, Jd=(1.0, 1.0, 1.0), R=QuatRotation):
        self.rotation = R
        self.Jd = np.asarray(Jd, dtype=np.float64)

    def _params(self):
        return self.Jd


# ---- trait queries ------------------------------------------------------------------------------------------------
def _cm(fun):
    if isinstance(fun, LieState):
        return fun._model()
    return fun.continuous_dynamics if isinstance(fun, DiscretizedDynamics) else fun


def state_dim(fun):
    if isinstance(fun, (AbstractKnotPoint, PackedTrajectory)):
        return fun.n
    if isinstance(fun, LieState):
        return len(fun)
    if not isinstance(fun, AbstractFunction):
        raise TypeError(f"state_dim needs to be implemented for {type(fun)}")
    return _cm(fun)._dims4()[0]


def control_dim(fun):
    if isinstance(fun, (AbstractKnotPoint, PackedTrajectory)):
        return fun.m
    return _cm(fun)._dims4()[1]


def output_dim(fun):
    return _cm(fun)._dims4()[2]


def errstate_dim(fun):
    if isinstance(fun, LieState):
        return fun.errstate_dim()
    if isinstance(fun, AbstractKnotPoint):     # src/functionbase.jl:134
        return fun.n
    return _cm(fun)._dims4()[3]


def dims(fun):
    if isinstance(fun, AbstractKnotPoint):
        return (fun.n, fun.m)
    if isinstance(fun, PackedTrajectory):      # dims(Z) = (n per knot, m per knot, N)  src/trajectories.jl:131
        N = len(fun)
        return ([fun.n] * N, [fun.m] * N, N)
    return _cm(fun)._dims4()[:3]


def jacobian_width(fun):                       # src/functionbase.jl:135
    return errstate_dim(fun) + control_dim(fun)


def functioninputs(fun):
    return StateControl()


def input_dim(fun):
    return state_dim(fun) + control_dim(fun)


def default_diffmethod(fun):                   # src/functionbase.jl:121 ; src/discretized_dynamics.jl:190,250-251
    if isinstance(fun, DiscretizedDynamics):
        if isinstance(fun.integrator, Implicit):
            return ImplicitFunctionTheorem(default_diffmethod(fun.continuous_dynamics) if False else ForwardAD())
        return ForwardAD()
    if isinstance(fun, CustomDiscreteModel):
        return ForwardAD()                         # what `@autodiff struct ... <: DiscreteDynamics` generates (test/discrete_dynamics_test.jl:71-75)
    return UserDefined()


def setnewtoniters(model, iters):              # src/discretized_dynamics.jl:227-228
    cm = _cm(model)
    cm._newton = (int(iters), getattr(cm, "_newton", (10, 1e-12))[1])
    L.check(L.lib().rdhip_model_set_newton(cm._handle, cm._newton[0], cm._newton[1]))


def setnewtontolerance(model, tol):            # src/discretized_dynamics.jl:225-226
    cm = _cm(model)
    cm._newton = (getattr(cm, "_newton", (10, 1e-12))[0], float(tol))
    L.check(L.lib().rdhip_model_set_newton(cm._handle, cm._newton[0], cm._newton[1]))


def default_signature(fun):
    return StaticReturn()


def statevectortype(fun):
    flag = C.c_int()
    L.check(L.lib().rdhip_model_statevectortype(_cm(fun)._handle, C.byref(flag)))
    return RotationState() if flag.value else EuclideanState()


def rotation_type(model):
    return _cm(model).rotation


def velocity_frame(model):
    return _cm(model).velocity_frame


def gen_inds(model):
    """1-based index ranges (r, q, v, w, u) of a RigidBody state (src/rigidbody.jl:80-94)."""
    first = (C.c_int * 5)()
    ln = (C.c_int * 5)()
    L.check(L.lib().rdhip_model_gen_inds(_cm(model)._handle, first, ln))
    return {k: range(first[i], first[i] + ln[i]) for i, k in enumerate(("r", "q", "v", "w", "u"))}


def wrench_sparsity(model):
    """wrench_sparsity(model) (src/rigidbody.jl:322-347): the 2 x 5 boolean block pattern [dF/dr dF/dq dF/dv dF/dw dF/du; dM/d...]."""
    js = (C.c_int * 10)()
    L.check(L.lib().rdhip_model_wrench_sparsity(_cm(model)._handle, js))
    return np.array(list(js), dtype=bool).reshape(5, 2).T


def _custom_wrench_call(model, Z, jac):
    """wrenches / wrench_jacobian! of a CustomRigidBody (the reference's default works for any RigidBody that defines forces and moments,
    src/rigidbody.jl:244-248, :315-320): the two snippets compiled once more as a helper model whose first six outputs are [F; M]; its `dynamics` is the
    wrench, the first six rows of its continuous ForwardAD Jacobian d wrenches / d z."""
    if not hasattr(model, "_wrench_parts"):
        raise NotImplementedError("wrenches are defined for RigidBody models (Satellite, Quadrotor, CustomRigidBody)")
    if not hasattr(model, "_wrench_helper"):
        forces, moments, p, preamble, m = model._wrench_parts
        model._wrench_helper = CustomModel(13, m, wrench_source(forces, moments), params=p, preamble=preamble)
    helper = model._wrench_helper
    Z, single = _as_traj(model, _split_args(Z))
    n, m = _check_traj(model, Z)
    N = len(Z)
    Z = PackedTrajectory(n, m, Z.data, Z.t, Z.dt * 0 + 1)       # wrenches(model, x, u, t) takes the control as given: no terminal-knot rule (as rdhip_wrenches)
    if not jac:
        Xd = _empty_like_data(Z.data, (N, n))
        _eval_into(helper, Xd, Z)
        return (Xd[:, :6].clone() if _is_torch(Xd) else np.ascontiguousarray(Xd[:, :6])), single
    J = alloc_jacobians(helper, N, Z.data)
    jacobian_(StaticReturn(), ForwardAD(), helper, J, None, Z)
    return J[:, :6, :], single


def _wrench_call(model, Z, out_shape, jac):
    Z, single = _as_traj(model, _split_args(Z))
    n, m = _check_traj(model, Z)
    N = len(Z)
    out = _empty_like_data(Z.data, (N,) + out_shape(n, m))
    az, at = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t")
    ao = _Arg(out, "output", az.dtype)
    ctx, space = _resolve_ctx([az, at, ao])
    fn = L.lib().rdhip_wrench_jacobian if jac else L.lib().rdhip_wrenches
    L.check(fn(ctx.handle, _cm(model)._handle, az.dtype, N, az.ptr, at.ptr, ao.ptr, space), ctx.handle)
    return out, single


def wrenches(model, *args):
    """wrenches(model, x, u, t) = [forces; moments] (src/rigidbody.jl:244-248): one knot point (x, u[, t]) / KnotPoint, or every knot point of a
    PackedTrajectory -> (N, 6)."""
    if isinstance(_cm(model), CustomModel):
        W, single = _custom_wrench_call(_cm(model), args, False)
        return W[0] if single else W
    W, single = _wrench_call(model, args, lambda n, m: (6,), False)
    return W[0] if single else W


def wrench_jacobian(model, *args):
    """d wrenches / d z (what wrench_jacobian!(F, model, z) of src/rigidbody.jl:315-320 is written to return): (6, n+m) for one knot point,
    (N, 6, n+m) views for a PackedTrajectory (stored (N, n+m, 6): each slab column-major 6 x (n+m) like the reference's F)."""
    if isinstance(_cm(model), CustomModel):
        F, single = _custom_wrench_call(_cm(model), args, True)
        return F[0] if single else F
    F, single = _wrench_call(model, args, lambda n, m: (n + m, 6), True)
    F = F.transpose(1, 2) if _is_torch(F) else F.transpose(0, 2, 1)
    return F[0] if single else F


def _ix(r):     # 1-based Julia range -> numpy slice
    return slice(r.start - 1, r.stop - 1)


def position(model, x):               # src/rigidbody.jl:97
    return np.asarray(x)[_ix(gen_inds(model)["r"])]


def linear_velocity(model, x):        # src/rigidbody.jl:98
    return np.asarray(x)[_ix(gen_inds(model)["v"])]


def angular_velocity(model, x):       # src/rigidbody.jl:99
    return np.asarray(x)[_ix(gen_inds(model)["w"])]


def orientation(model, x, renorm=False):   # src/rigidbody.jl:101-108
    R = rotation_type(model)
    q = np.asarray(x, dtype=np.float64)[_ix(gen_inds(model)["q"])]
    return R(*q, renorm) if R is QuatRotation else R(*q)


def flipquat(model, x):               # src/rigidbody.jl:115-118
    if rotation_type(model) is not QuatRotation:
        raise TypeError("flipquat is defined for RigidBody{<:QuatRotation}")
    y = np.array(x, copy=True)
    y[3:7] = -y[3:7]
    return y


def parse_state(model, x, renorm=False):   # src/rigidbody.jl:127-133
    return position(model, x), orientation(model, x, renorm), linear_velocity(model, x), angular_velocity(model, x)


def build_state(model, *args):        # src/rigidbody.jl:161-199
    """build_state(model, r, q, v, w) with q a rotation value (converted to the model's rotation type) or a parameter vector of the
    model's rotation (taken as is, like the reference: no renormalisation); build_state(model, x) with a 13-vector [r; quat; v; w]."""
    R = rotation_type(model)
    if len(args) == 1:
        x = np.asarray(args[0], dtype=np.float64)
        if x.shape != (13,):
            raise ValueError("build_state(model, x): x is [r; q; v; w] with a quaternion (13 entries)")
        r, q, v, w = x[0:3], QuatRotation(*x[3:7], False), x[7:10], x[10:13]
    elif len(args) == 4:
        r, q, v, w = args
    else:
        raise TypeError("build_state(model, r, q, v, w) or build_state(model, x)")
    if isinstance(q, _Rotation):
        qp = q.params() if type(q) is R else R(q).params()
    else:
        qp = np.asarray(q, dtype=np.float64)
        if qp.shape == (4,) and R is not QuatRotation:      # a quaternion given as a vector (RBState(r, q_, v, w)): convert
            qp = R(QuatRotation(*qp, False)).params()
        if qp.shape != (R.nparams,):
            raise AssertionError(f"length(q) == {R.nparams}")
    return np.concatenate([np.asarray(r, dtype=np.float64), qp, np.asarray(v, dtype=np.float64), np.asarray(w, dtype=np.float64)])


def fill_state(model, x, q, v, w):    # src/rigidbody.jl:201-207
    nq = rotation_type(model).nparams
    return np.array([x] * 3 + [q] * nq + [v] * 3 + [w] * 3)


def vec_states(s, x):                 # src/liestate.jl:142-151
    s = s if isinstance(s, LieState) else LieState(s)
    x = np.asarray(x)
    return tuple(x[_ix(vec_inds(s, i))] for i in range(1, len(s.P) + 1))


def rot_states(s, x):                 # src/liestate.jl:160-173 (quaternions are NOT renormalised)
    s = s if isinstance(s, LieState) else LieState(s)
    x = np.asarray(x, dtype=np.float64)
    return tuple(s.R(*x[_ix(rot_inds(s, i))], False) if s.R is QuatRotation else s.R(*x[_ix(rot_inds(s, i))]) for i in range(1, len(s.P)))


class LieState:
    """LieState{R,P}: partition of a state vector into vector parts P and rotations R (src/liestate.jl:75-131)."""

    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], AbstractFunction):
            rot, nP, P = C.c_int(), C.c_int(), (C.c_int * 8)()
            L.check(L.lib().rdhip_model_liestate(_cm(args[0])._handle, C.byref(rot), C.byref(nP), P))
            if nP.value == 0:
                raise TypeError(f"LieState is not defined for the Euclidean model {type(args[0]).__name__}")
            self.R = {L.ROT_QUAT: QuatRotation, L.ROT_MRP: MRP, L.ROT_RP: RodriguesParam}[rot.value]
            self.P = tuple(P[i] for i in range(nP.value))
        else:
            self.R = args[0]
            self.P = tuple(args[1]) if len(args) == 2 and not isinstance(args[1], int) else tuple(int(p) for p in args[1:])
        self._q = None

    def _query(self):
        if self._q is None:
            P = (C.c_int * len(self.P))(*self.P)
            ln, ed = C.c_int(), C.c_int()
            rf, vf = (C.c_int * 8)(), (C.c_int * 8)()
            L.check(L.lib().rdhip_liestate_query(self.R.code, P, len(self.P), C.byref(ln), C.byref(ed), rf, vf))
            self._q = (ln.value, ed.value, list(rf)[: len(self.P) - 1], list(vf)[: len(self.P)])
        return self._q

    def __len__(self):
        return self._query()[0]

    def errstate_dim(self):
        return self._query()[1]

    def _model(self):
        """A state-only native model carrying this partition: lets state_diff(s::LieState, x, x0) (src/liestate.jl:210-260) run batched."""
        m = getattr(self, "_m", None)
        if m is None:
            m = self._m = CustomModel(len(self), 1, "/* state-only LieState carrier */ xdot[0] = x[0];", liestate=(self.R, self.P))
        return m

    def __eq__(self, other):
        return isinstance(other, LieState) and self.R is other.R and self.P == other.P

    def __hash__(self):
        return hash((self.R, self.P))

    def __repr__(self):
        return f"LieState{{{self.R.__name__},{self.P}}}()"


def QuatState(n, Q):
    """QuatState(n, Q): LieState with quaternions starting at the 1-based indices Q (src/liestate.jl:99-113)."""
    Q = list(Q)
    P = [Q[0] - 1] + [Q[i] - Q[i - 1] - 4 for i in range(1, len(Q))] + [n - (Q[-1] + 4) + 1]
    return LieState(QuatRotation, tuple(P))


def num_rotations(s):
    return len(s.P) - 1


def rot_inds(s, i):      # 1-based ranges like Julia (src/liestate.jl:127)
    f = s._query()[2][i - 1]
    return range(f, f + s.R.nparams)


def vec_inds(s, i):      # src/liestate.jl:128-129
    f = s._query()[3][i - 1]
    return range(f, f + s.P[i - 1])


def inds(s, i):          # src/liestate.jl:130
    return vec_inds(s, 1 + i // 2) if i % 2 == 1 else rot_inds(s, i // 2)


# ---- rand / zeros / fill generators (src/functionbase.jl:271-278, src/rigidbody.jl:50-70, src/liestate.jl:175-205) ------
def rand(model, rng=None, dtype=np.float64):
    rng = np.random.default_rng() if rng is None else rng
    n, m = state_dim(model), control_dim(model)
    x = rng.random(n)
    if isinstance(statevectortype(model), RotationState):
        s = LieState(model)
        for i in range(1, num_rotations(s) + 1):
            ri = rot_inds(s, i)
            if s.R is QuatRotation:   # rand(QuatRotation): normalised 4-D Gaussian
                q = rng.standard_normal(4)
                x[ri.start - 1: ri.stop - 1] = q / np.linalg.norm(q)
    return x.astype(dtype), rng.random(m).astype(dtype)


def zeros(model, dtype=np.float64):
    n, m = state_dim(model), control_dim(model)
    x = np.zeros(n, dtype=dtype)
    if isinstance(statevectortype(model), RotationState):
        s = LieState(model)
        if s.R is QuatRotation:
            for i in range(1, num_rotations(s) + 1):
                x[rot_inds(s, i).start - 1] = 1.0   # one(QuatRotation) = [1,0,0,0]
    u = np.zeros(m, dtype=dtype)
    cm = _cm(model)
    if isinstance(cm, Quadrotor):       # hover controls, test/quadrotor.jl:133-137
        u[:] = -cm.mass * cm.gravity[2] / 4
    return x, u


def fill(model, v):
    return np.full(state_dim(model), v), np.full(control_dim(model), v)


def trim_controls(model):               # test/quadrotor.jl:52-54
    return np.full(4, -model.gravity[2] * model.mass / 4.0)


# =============================================================================================================
# DynamicsJacobian (src/jacobian.jl)
# =============================================================================================================
class DynamicsJacobian:
    """n x (n+m) Jacobian stored column-major, with A / B views (src/jacobian.jl:26-48,74-75)."""

    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], AbstractFunction):
            n, m = state_dim(args[0]), control_dim(args[0])
        elif len(args) == 1:
            F = np.asarray(args[0], dtype=np.float64)
            n, m = F.shape[0], F.shape[1] - F.shape[0]
        else:
            n, m = args
        self.data = np.zeros((n, n + m), order="F")
        if len(args) == 1 and not isinstance(args[0], AbstractFunction):
            self.data[:] = args[0]
        self.A = self.data[:, :n]
        self.B = self.data[:, n:]

    @property
    def shape(self):
        return self.data.shape

    def __getitem__(self, i):
        return self.data[i]

    def __setitem__(self, i, v):
        self.data[i] = v

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)


def get_A(D):
    return D.A


def get_B(D):
    return D.B


def get_data(A):
    return A.data if isinstance(A, DynamicsJacobian) else A


# =============================================================================================================
# the batched operators
# =============================================================================================================
def _model_rule(fun):
    if isinstance(fun, DiscretizedDynamics):
        return fun.continuous_dynamics, _rule_code(fun.integrator)
    if isinstance(fun, ContinuousDynamics):
        return fun, L.CONTINUOUS
    if isinstance(fun, CustomDiscreteModel):
        return fun, L.DISCRETE
    raise NotImplementedError(f"no batched kernels for {type(fun).__name__}")


def _as_traj(fun, z):
    """Accept a PackedTrajectory, a single knot point, or (x, u, t, dt)."""
    if isinstance(z, PackedTrajectory):
        return z, False
    if isinstance(z, AbstractKnotPoint):
        data = np.ascontiguousarray(np.asarray(z.z)[None, :])
        if data.dtype not in (np.float64, np.float32):
            data = data.astype(np.float64)
        return PackedTrajectory(z.n, z.m, data, [z.t], [z.dt]), True
    raise TypeError(f"expected a KnotPoint or PackedTrajectory, got {type(z)}")


def _check_traj(fun, Z):
    n, m = state_dim(fun), control_dim(fun)
    if (Z.n, Z.m) != (n, m):
        raise ValueError(f"trajectory has (n,m)=({Z.n},{Z.m}) but the model has ({n},{m})")
    return n, m


def _eval_into(fun, Y, Z):
    model, rule = _model_rule(fun)
    n, m = _check_traj(fun, Z)
    N = len(Z)
    _dense(Y, (N, n), "output")
    az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
    ay = _Arg(Y, "output", az.dtype)
    ctx, space = _resolve_ctx([az, at, adt, ay])
    lib = L.lib()
    if rule == L.CONTINUOUS:
        rc = lib.rdhip_dynamics(ctx.handle, model._handle, az.dtype, N, az.ptr, at.ptr, adt.ptr, ay.ptr, space)
    else:
        rc = lib.rdhip_discrete_dynamics(ctx.handle, model._handle, rule, az.dtype, N, az.ptr, at.ptr, adt.ptr, ay.ptr, space)
    L.check(rc, ctx.handle)
    return Y


def _split_args(args):
    """(z,) | (x, u) | (x, u, t) | (x, u, t, dt) -> a one-knot PackedTrajectory."""
    if len(args) == 1:
        return args[0]
    x, u = np.asarray(args[0], dtype=np.float64), np.asarray(args[1], dtype=np.float64)
    t = args[2] if len(args) > 2 else 0.0
    dt = args[3] if len(args) > 3 else np.inf    # continuous dynamics never reads dt; Inf is "not terminal" (src/trajectories.jl:104)
    return KnotPoint(len(x), len(u), np.concatenate([x, u]), t, dt)


def evaluate(fun, *args):
    """evaluate(fun, z): src/functionbase.jl:214-218 -> dynamics (src/dynamics.jl:110) or discrete_dynamics
    (src/discrete_dynamics.jl:65).  z may be one knot point or a whole PackedTrajectory."""
    Z, single = _as_traj(fun, _split_args(args))
    Y = _empty_like_data(Z.data, (len(Z), state_dim(fun)))
    _eval_into(fun, Y, Z)
    return Y[0] if single else Y


def evaluate_(fun, y, *args):
    """evaluate!(fun, y, z): src/functionbase.jl:215."""
    Z, single = _as_traj(fun, _split_args(args))
    if single:
        y[...] = evaluate(fun, Z)[0]
        return None
    _eval_into(fun, y, Z)
    return None


def dynamics(model, *args):
    """dynamics(model, z) / (model, x, u[, t]): src/dynamics.jl:81-83."""
    if not isinstance(model, ContinuousDynamics):
        raise TypeError("dynamics needs a ContinuousDynamics model")
    return evaluate(model, *args)


def dynamics_(*args):
    """dynamics!(model, xdot, z) / dynamics!(sig, model, xdot, z): src/dynamics.jl:95-107."""
    if isinstance(args[0], FunctionSignature):
        args = args[1:]
    model, xdot = args[0], args[1]
    return evaluate_(model, xdot, *args[2:])


def discrete_dynamics(*args):
    """discrete_dynamics(dmodel, z) / (dmodel, x, u, t, dt): src/discrete_dynamics.jl:80-83.
    v0.3 spelling discrete_dynamics(RK3, model, z) is accepted."""
    if isinstance(args[0], type) and issubclass(args[0], QuadratureRule) or isinstance(args[0], QuadratureRule):
        fun, rest = DiscretizedDynamics(args[1], args[0]), args[2:]
    else:
        fun, rest = args[0], args[1:]
    if not isinstance(fun, DiscreteDynamics):
        raise TypeError("discrete_dynamics needs a DiscreteDynamics model (wrap it: DiscretizedDynamics(model, RK4))")
    return evaluate(fun, *rest)


def discrete_dynamics_(*args):
    """discrete_dynamics!(dmodel, xn, z) / (sig, dmodel, xn, z): src/discrete_dynamics.jl:94-108."""
    if isinstance(args[0], FunctionSignature):
        args = args[1:]
    if isinstance(args[0], type) and issubclass(args[0], QuadratureRule):   # v0.3: discrete_dynamics!(Q, xn, model, Z)
        Q, xn, model = args[0], args[1], args[2]
        return evaluate_(DiscretizedDynamics(model, Q), xn, *args[3:])
    return evaluate_(args[0], args[1], *args[2:])


def jacobian_(*args):
    """jacobian!(sig, diff, fun, J, y, z): src/functionbase.jl:242 (generated methods src/jacobian_gen.jl:485-529,
    :643-685; chain rule src/discretized_dynamics.jl:208-219).  Also the v0.3 form jacobian!(J, model, z).
    Batched: J (N,n,n+m) from alloc_jacobians, y (N,n) or None, z a PackedTrajectory.  Per-knot: J (n,n+m), y (n)."""
    if len(args) == 3:    # v0.3: jacobian!(D, model, z)
        J, fun, z = args
        sig, diff, y = default_signature(fun), default_diffmethod(fun), None
    elif len(args) == 6:
        sig, diff, fun, J, y, z = args
    else:
        raise TypeError("jacobian_(sig, diff, fun, J, y, z)  or  jacobian_(J, model, z)")
    if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
        raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
    model, rule = _model_rule(fun)
    if rule == L.IMPLICIT_MIDPOINT:
        # src/discretized_dynamics.jl:243-249: "Must call jacobian! on an ImplicitDynamicsModel with the ImplicitFunctionTheorem diff method"
        if not isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError(f"Must call jacobian! on an implicit model with the ImplicitFunctionTheorem diff method. Got {type(diff).__name__}.")
        if not isinstance(diff.diff, ForwardAD):
            raise NotImplementedError("ImplicitFunctionTheorem is built over ForwardAD continuous Jacobians only")
        diff = diff.diff
    elif isinstance(diff, ImplicitFunctionTheorem):
        raise ValueError("ImplicitFunctionTheorem needs an implicit integrator")
    Z, single = _as_traj(fun, z)
    n, m = _check_traj(fun, Z)
    N = len(Z)
    if single:
        Jb = alloc_jacobians(fun, 1, Z.data)
        yb = np.empty((1, n), dtype=Z.data.dtype)
        jacobian_(sig, diff, fun, Jb, yb, Z)
        get_data(J)[...] = Jb[0]
        if y is not None:
            y[...] = yb[0]
        return None
    _dense(J, (N, n, n + m), "J", transposed_last2=True)
    az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
    aj = _Arg(J, "J", az.dtype)
    ay = _Arg(y, "y", az.dtype)
    if y is not None:
        _dense(y, (N, n), "y")
    ctx, space = _resolve_ctx([az, at, adt, aj, ay])
    L.check(L.lib().rdhip_jacobian(ctx.handle, model._handle, rule, diff.code, az.dtype, N, az.ptr, at.ptr, adt.ptr, aj.ptr, ay.ptr, space), ctx.handle)
    return None


_shard_ctx = {}


def jacobian_sharded_(sig, diff, fun, J, y, Z, devices):
    """jacobian!(sig, diff, fun, J, y, Z) for host (numpy) arrays with the knot points cut into len(devices) contiguous blocks, block i on
    GPU devices[i] of THIS process, all blocks concurrently (rdhip_jacobian_sharded: one context and one host thread per entry; a device
    may be listed more than once).  The stacked J / y are complete on return.  The one-process alternative to rdhip.parallel for
    hosts without torch.distributed."""
    if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
        raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
    model, rule = _model_rule(fun)
    if rule == L.IMPLICIT_MIDPOINT:
        if not isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError(f"Must call jacobian! on an implicit model with the ImplicitFunctionTheorem diff method. Got {type(diff).__name__}.")
        diff = diff.diff
    if not isinstance(Z, PackedTrajectory) or _is_torch(Z.data):
        raise TypeError("jacobian_sharded_ takes a host (numpy) PackedTrajectory")
    n, m = _check_traj(fun, Z)
    N = len(Z)
    _dense(J, (N, n, n + m), "J", transposed_last2=True)
    az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
    aj, ay = _Arg(J, "J", az.dtype), _Arg(y, "y", az.dtype)
    if y is not None:
        _dense(y, (N, n), "y")
    if any(a.space != L.HOST for a in (az, at, adt, aj) if a.ptr is not None) or (ay.ptr is not None and ay.space != L.HOST):
        raise ValueError("jacobian_sharded_ takes host arrays")
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("no devices")
    ctxs = []
    for i, d in enumerate(devices):
        key = (d, devices[:i].count(d))
        if key not in _shard_ctx:
            _shard_ctx[key] = L.Context(d)
        ctxs.append(_shard_ctx[key])
    arr = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    rc = L.lib().rdhip_jacobian_sharded(arr, len(ctxs), model._handle, rule, diff.code, az.dtype, N, az.ptr, at.ptr, adt.ptr, aj.ptr, ay.ptr)
    L.check(rc, None)
    return None


class PreparedJacobian:
    """jacobian!(sig, diff, fun, J, y, Z) with the argument checking done ONCE: calling the object re-issues the same batched
    call on the same arrays (their contents may change, their memory may not).  For solver loops that evaluate the same packed
    trajectory every iteration: the per-call host cost drops from ~15 us of Python checks to the bare C call (~5 us; a Julia
    `ccall` host pays only that anyway).  Device arrays: launched on torch's current stream at call time."""

    def __init__(self, sig, diff, fun, J, y, Z, ctx=None):
        """ctx: an rdhip Context that OWNS its stream (rdhip.Context(device)) to issue the call on instead of torch's current stream -- e.g. one with
        launch overlap switched on (Context.set_overlap); the caller orders other streams against it with Context.fence() / synchronize()."""
        if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
            raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
        if not isinstance(Z, PackedTrajectory):
            raise TypeError("PreparedJacobian needs a PackedTrajectory")
        self._own_ctx = ctx
        model, rule = _model_rule(fun)
        if rule == L.IMPLICIT_MIDPOINT:                      # src/integration.jl:517-538: implicit rules differentiate through the implicit function theorem
            if not isinstance(diff, ImplicitFunctionTheorem):
                raise ValueError(f"Must call jacobian! on an implicit model with the ImplicitFunctionTheorem diff method. Got {type(diff).__name__}.")
            diff = diff.diff
        elif isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError("ImplicitFunctionTheorem needs an implicit integrator")
        n, m = _check_traj(fun, Z)
        N = len(Z)
        _dense(J, (N, n, n + m), "J", transposed_last2=True)
        az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
        aj, ay = _Arg(J, "J", az.dtype), _Arg(y, "y", az.dtype)
        if y is not None:
            _dense(y, (N, n), "y")
        ctx, space = _resolve_ctx([az, at, adt, aj, ay])
        self._keep = (fun, model, J, y, Z)                  # keep the arrays (and the native model handle) alive
        self._space, self._dev = space, az.device
        self._fn = L.lib().rdhip_jacobian
        self._args = (model._handle, rule, diff.code, az.dtype, N, az.ptr, at.ptr, adt.ptr, aj.ptr, ay.ptr, space)
        self._ctx = ctx

    def __call__(self):
        if self._own_ctx is not None and self._space == L.DEVICE:
            ctx = self._own_ctx
        else:
            ctx = L.context_for(self._dev, torch.cuda.current_stream(self._dev).cuda_stream) if self._space == L.DEVICE else self._ctx
        rc = self._fn(ctx.handle, *self._args)
        if rc:
            L.check(rc, ctx.handle)


def prepare_jacobian(sig, diff, fun, J, y, Z, ctx=None):
    return PreparedJacobian(sig, diff, fun, J, y, Z, ctx)


class PreparedJacobianMulti:
    """jacobian!(sig, diff, fun, Js[p], ys[p], Zs[p]) for several INDEPENDENT packed trajectories in ONE kernel launch (rdhip_jacobian_multi):
    what `for p in problems, k in 1:N; jacobian!(sig, diff, model, J[p][k], y, Z[p][k]); end` does for the per-knot loops of several
    TrajectoryOptimization problems / shooting segments (src/functionbase.jl:242).  A trajectory of a few thousand knot points is launch-bound
    on its own; sixteen 4 096-knot trajectories enumerated by one persistent tile loop cost what one 65 536-knot batch costs.  Argument
    checking happens once, calling the object re-issues the launch (array contents may change, their memory may not)."""

    def __init__(self, sig, diff, fun, Js, ys, Zs):
        if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
            raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
        model, rule = _model_rule(fun)
        if rule == L.IMPLICIT_MIDPOINT:
            if not isinstance(diff, ImplicitFunctionTheorem):
                raise ValueError(f"Must call jacobian! on an implicit model with the ImplicitFunctionTheorem diff method. Got {type(diff).__name__}.")
            diff = diff.diff
        elif isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError("ImplicitFunctionTheorem needs an implicit integrator")
        Zs, Js = list(Zs), list(Js)
        ys = [None] * len(Zs) if ys is None else list(ys)
        if not (len(Zs) == len(Js) == len(ys)):
            raise ValueError("Js, ys and Zs must list one entry per trajectory")
        nb = len(Zs)
        args, dtype = [], None
        Ns = (C.c_longlong * max(nb, 1))()
        tabs = [(C.c_void_p * max(nb, 1))() for _ in range(5)]
        for b, (Z, J, y) in enumerate(zip(Zs, Js, ys)):
            if not isinstance(Z, PackedTrajectory):
                raise TypeError("every trajectory must be a PackedTrajectory")
            n, m = _check_traj(fun, Z)
            N = len(Z)
            _dense(J, (N, n, n + m), f"Js[{b}]", transposed_last2=True)
            az, at, adt = _Arg(Z.data, f"Zs[{b}].data"), _Arg(Z.t, f"Zs[{b}].t"), _Arg(Z.dt, f"Zs[{b}].dt")
            dtype = az.dtype if dtype is None else dtype
            if az.dtype != dtype:
                raise ValueError("all trajectories must share one dtype")
            aj, ay = _Arg(J, f"Js[{b}]", dtype), _Arg(y, f"ys[{b}]", dtype)
            if y is not None:
                _dense(y, (N, n), f"ys[{b}]")
            args += [az, at, adt, aj, ay]
            Ns[b] = N
            for tab, a in zip(tabs, (az, at, adt, aj, ay)):
                tab[b] = a.ptr
        ctx, space = _resolve_ctx(args) if args else (None, L.HOST)
        self._keep = (fun, model, Js, ys, Zs, Ns, tabs)
        self._space, self._dev, self._ctx, self._nb = space, (args[0].device if args else None), ctx, nb
        self._args = (model._handle, rule, diff.code, dtype if dtype is not None else L.F64, nb, Ns, *tabs, space)

    def __call__(self):
        if self._nb == 0:
            return
        ctx = L.context_for(self._dev, torch.cuda.current_stream(self._dev).cuda_stream) if self._space == L.DEVICE else self._ctx
        rc = L.lib().rdhip_jacobian_multi(ctx.handle, *self._args)
        if rc:
            L.check(rc, ctx.handle)
        self._last_ctx = ctx

    @property
    def launches(self):
        """kernel launches the last call issued (rdhip_ctx_last_multi_launches): 1 = fused, 3 = gathered into one stack (no batch-table kernel for this diff method), otherwise one per run of batches"""
        ctx = getattr(self, "_last_ctx", None)
        if ctx is None:
            return 0
        k = C.c_int(0)
        L.check(L.lib().rdhip_ctx_last_multi_launches(ctx.handle, C.byref(k)), ctx.handle)
        return k.value


def prepare_jacobian_multi(sig, diff, fun, Js, ys, Zs):
    return PreparedJacobianMulti(sig, diff, fun, Js, ys, Zs)


def jacobian_multi_(sig, diff, fun, Js, ys, Zs):
    """One launch for the Jacobians of several independent packed trajectories: see PreparedJacobianMulti.  Returns the number of kernel
    launches the call issued (1 = fused)."""
    call = PreparedJacobianMulti(sig, diff, fun, Js, ys, Zs)
    call()
    return call.launches


def discrete_jacobian_(Q, J, model, z):
    """v0.3 discrete_jacobian!(Q, D, model, z) == jacobian!(sig, ForwardAD(), DiscretizedDynamics{Q}(model), D, y, z)."""
    fun = DiscretizedDynamics(model, Q)
    return jacobian_(default_signature(fun), ForwardAD(), fun, J, None, z)


def _pair_traj(fun, z2, z1):
    """(z2, z1) knot points -> the two-knot PackedTrajectory [z1, z2]."""
    if not (isinstance(z2, AbstractKnotPoint) and isinstance(z1, AbstractKnotPoint)):
        raise TypeError("expected two knot points (z2, z1)")
    data = np.ascontiguousarray(np.stack([np.asarray(z1.z, dtype=np.float64), np.asarray(z2.z, dtype=np.float64)]))
    return PackedTrajectory(z1.n, z1.m, data, [z1.t, z2.t], [z1.dt, z2.dt])


def dynamics_error(fun, *args):
    """dynamics_error(model, z2, z1) = discrete_dynamics(model, z1) - state(z2): src/discrete_dynamics.jl:134-135.
    dynamics_error(model, Z) with a PackedTrajectory returns the (N-1, n) errors of all consecutive pairs (Z[k], Z[k+1])."""
    model, rule = _model_rule(fun)
    if rule == L.CONTINUOUS:
        raise TypeError("dynamics_error needs a DiscreteDynamics model")
    single = len(args) == 2
    Z = _pair_traj(fun, *args) if single else args[0]
    n, m = _check_traj(fun, Z)
    N = len(Z)
    E = _empty_like_data(Z.data, (max(N - 1, 0), n))
    if N >= 2:
        az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
        ae = _Arg(E, "E", az.dtype)
        ctx, space = _resolve_ctx([az, at, adt, ae])
        L.check(L.lib().rdhip_dynamics_error(ctx.handle, model._handle, rule, az.dtype, N, az.ptr, at.ptr, adt.ptr, ae.ptr, space), ctx.handle)
    return E[0] if single else E


def dynamics_error_(fun, y2, y1, *args):
    """dynamics_error!(model, y2, y1, z2, z1): src/discrete_dynamics.jl:146-156 (y1 is scratch in the reference; unused here)."""
    y2[...] = dynamics_error(fun, *args)
    return None


def dynamics_error_jacobian_(sig, diff, fun, J2, J1, y2, y1, *args):
    """dynamics_error_jacobian!(sig, diff, model, J2, J1, y2, y1, z2, z1): src/discrete_dynamics.jl:158-201.  Explicit rules:
    J1 = [A B] at z1, J2 = [-I 0].  Batched form: pass ONE PackedTrajectory instead of (z2, z1); J2, J1 are (N-1, n, n+m) stacks
    from alloc_jacobians (either may be None), y2 (N-1, n) or None receives the error, y1 is ignored."""
    if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
        raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
    model, rule = _model_rule(fun)
    if rule == L.CONTINUOUS:
        raise TypeError("dynamics_error_jacobian! needs a DiscreteDynamics model")
    if isinstance(diff, ImplicitFunctionTheorem):      # the implicit model's diff method wraps the one used for the continuous Jacobians
        diff = diff.diff
    if len(args) == 2:
        Z = _pair_traj(fun, *args)
        n, m = _check_traj(fun, Z)
        J2b, J1b = alloc_jacobians(fun, 1, Z.data), alloc_jacobians(fun, 1, Z.data)
        yb = np.empty((1, n))
        dynamics_error_jacobian_(sig, diff, fun, J2b, J1b, yb, None, Z)
        if J2 is not None:
            get_data(J2)[...] = J2b[0]
        if J1 is not None:
            get_data(J1)[...] = J1b[0]
        if y2 is not None:
            y2[...] = yb[0]
        return None
    Z = args[0]
    n, m = _check_traj(fun, Z)
    N = len(Z)
    P = max(N - 1, 0)
    for name, Jx in (("J2", J2), ("J1", J1)):
        if Jx is not None:
            _dense(Jx, (P, n, n + m), name, transposed_last2=True)
    if y2 is not None:
        _dense(y2, (P, n), "y2")
    if N < 2:
        return None
    az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
    a2, a1, ay = _Arg(J2, "J2", az.dtype), _Arg(J1, "J1", az.dtype), _Arg(y2, "y2", az.dtype)
    ctx, space = _resolve_ctx([az, at, adt, a2, a1, ay])
    L.check(L.lib().rdhip_dynamics_error_jacobian(ctx.handle, model._handle, rule, diff.code, az.dtype, N, az.ptr, at.ptr, adt.ptr,
                                                  a2.ptr, a1.ptr, ay.ptr, space), ctx.handle)
    return None


def alloc_error_expansions(model, N, like=None, dtype=None):
    """(N-1, e, e+m) stack of error-state expansions [G2'AG1 | G2'B] with column-major slabs (like alloc_jacobians)."""
    e, m = errstate_dim(model), control_dim(model)
    base = _alloc((max(N - 1, 0), e + m, e), like, dtype)
    return base.transpose(1, 2) if _is_torch(base) else base.transpose(0, 2, 1)


def error_expansion_(fun, E, y, Z):
    """Batched error-state expansion along a trajectory: E[k] = [G(x_{k+1})' A_k G(x_k) | G(x_{k+1})' B_k], k = 0..N-2, with
    [A_k B_k] = jacobian!(sig, ForwardAD(), DiscretizedDynamics{Q}(model), ., ., Z[k]) and G = errstate_jacobian!.  This is what
    TrajectoryOptimization.jl's `error_expansion!` assembles per knot point from this path's outputs; for RigidBody models it is
    fused into the Jacobian kernel.  E from alloc_error_expansions; y (N-1, n) or None receives f(z_k)."""
    model, rule = _model_rule(fun)
    if rule == L.CONTINUOUS:
        raise TypeError("error_expansion! needs a DiscreteDynamics model")
    n, m = _check_traj(fun, Z)
    e = errstate_dim(fun)
    N = len(Z)
    P = max(N - 1, 0)
    _dense(E, (P, e, e + m), "E", transposed_last2=True)
    if y is not None:
        _dense(y, (P, n), "y")
    if N < 2:
        return None
    az, at, adt = _Arg(Z.data, "Z.data"), _Arg(Z.t, "Z.t"), _Arg(Z.dt, "Z.dt")
    ae, ay = _Arg(E, "E", az.dtype), _Arg(y, "y", az.dtype)
    ctx, space = _resolve_ctx([az, at, adt, ae, ay])
    L.check(L.lib().rdhip_error_expansion(ctx.handle, model._handle, rule, az.dtype, N, az.ptr, at.ptr, adt.ptr, ae.ptr, ay.ptr, space), ctx.handle)
    return None


def _states_of(model, x):
    """(array, ld, N, single): states from a PackedTrajectory / KnotPoint / (N, >=n) array / single vector."""
    n = state_dim(model)
    if isinstance(x, PackedTrajectory):
        return x.data, x.n + x.m, len(x), False
    if isinstance(x, AbstractKnotPoint):
        a = np.ascontiguousarray(np.asarray(x.z, dtype=np.float64)[None, :n])
        return a, n, 1, True
    if (isinstance(x, np.ndarray) or _is_torch(x)) and x.ndim == 2:
        # rows = states; a row stride larger than the row length is a leading dimension (e.g. Z.data[:, :n])
        st = tuple(x.stride()) if _is_torch(x) else tuple(v // x.itemsize for v in x.strides)
        if x.shape[1] < n:
            raise ValueError(f"states have length {x.shape[1]}, model needs {n}")
        if x.shape[1] > 1 and st[1] != 1 or (x.shape[0] > 1 and st[0] < x.shape[1]):
            raise ValueError(f"x: memory layout {st} is not a row-major state stack")
        return x, (st[0] if x.shape[0] > 1 else x.shape[1]), x.shape[0], False
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64)[None, :])
    if a.shape[1] < n:
        raise ValueError(f"state has length {a.shape[1]}, model needs {n}")
    return a, a.shape[1], 1, True


def errstate_jacobian_(model, G, x, clear=None):
    """errstate_jacobian!(model, G, z|x): src/statevectortype.jl:119-122.  Like the reference, G is NOT cleared for
    RotationState models unless clear=True (src/liestate.jl:293-297); Euclidean models write G = I.
    A caller that zeroes G and then calls (the reference's own pattern, src/trajectories.jl:428-433) should pass clear=True and drop the
    zeroing: one dense write of whole lines (63 us at 262 144 quadrotor knots) instead of a memset plus 131 us of read-modify-write."""
    X, ldx, N, single = _states_of(model, x)
    cm = _cm(model)
    n, e = state_dim(model), errstate_dim(model)
    clear = False if clear is None else bool(clear)
    if single:
        Gb = np.ascontiguousarray(np.asarray(get_data(G), dtype=X.dtype).T[None])      # (1, e, n)
        errstate_jacobian_(model, Gb.transpose(0, 2, 1), X, clear)
        get_data(G)[...] = Gb[0].T
        return None
    _dense(G, (N, n, e), "G", transposed_last2=True)
    ax = _Arg(X, "x")
    ag = _Arg(G, "G", ax.dtype)
    ctx, space = _resolve_ctx([ax, ag])
    L.check(L.lib().rdhip_errstate_jacobian(ctx.handle, cm._handle, ax.dtype, N, ax.ptr, ldx, int(clear), ag.ptr, space), ctx.handle)
    return None


def state_diff_jacobian_(G, model, z):
    """v0.3 state_diff_jacobian!(G, model, z) (deprecated alias, src/deprecate.jl:1) -- note the argument order."""
    return errstate_jacobian_(model, G, z)


class ErrorMap(_Trait):
    """Rotations.ErrorMap (third party): the optional last argument of state_diff (src/liestate.jl:26-32)."""


class CayleyMap(ErrorMap):           # the reference's default (src/liestate.jl:27,31): v / w
    code = L.ERRMAP_CAYLEY


class ExponentialMap(ErrorMap):      # 0.5 logm(dq): half the rotation vector
    code = L.ERRMAP_EXPONENTIAL


class MRPMap(ErrorMap):              # 2 v / (1 + w)
    code = L.ERRMAP_MRP


class QuatVecMap(ErrorMap):          # sign(w) v
    code = L.ERRMAP_QUATVEC


class IdentityMap(ErrorMap):         # params(R0 \ R); three-parameter rotations only (a quaternion has four: the reference throws)
    code = L.ERRMAP_IDENTITY


def _errmap_code(errmap):
    if errmap is None:
        return L.ERRMAP_CAYLEY
    if isinstance(errmap, type) and issubclass(errmap, ErrorMap):
        errmap = errmap()
    if not isinstance(errmap, ErrorMap) or not hasattr(errmap, "code"):
        raise TypeError("errmap must be one of CayleyMap, ExponentialMap, MRPMap, QuatVecMap, IdentityMap")
    return errmap.code


def state_diff(model, x, x0, errmap=None):
    """state_diff(model, x, x0[, errmap]): src/statevectortype.jl:104, src/liestate.jl:26-28 (default CayleyMap())."""
    X, ldx, N, single = _states_of(model, x)
    X0, ldx0, N0, _ = _states_of(model, x0)
    if N != N0:
        raise ValueError("x and x0 hold different numbers of states")
    dx = _empty_like_data(X, (N, errstate_dim(model)))
    state_diff_(model, dx, X, X0, errmap)
    return dx[0] if single else dx


def state_diff_(model, dx, x, x0, errmap=None):
    """state_diff!(model, dx, x, x0[, errmap]): src/statevectortype.jl:102, src/liestate.jl:30-32."""
    code = _errmap_code(errmap)
    X, ldx, N, single = _states_of(model, x)
    X0, ldx0, _, _ = _states_of(model, x0)
    e = errstate_dim(model)
    if single and not (hasattr(dx, "ndim") and dx.ndim == 2):
        tmp = np.empty((1, e), dtype=X.dtype)
        state_diff_(model, tmp, X, X0, errmap)
        dx[...] = tmp[0]
        return None
    _dense(dx, (N, e), "dx")
    ax, a0 = _Arg(X, "x"), _Arg(X0, "x0")
    ad = _Arg(dx, "dx", ax.dtype)
    if a0.dtype != ax.dtype:
        raise ValueError("x and x0 have different dtypes")
    ctx, space = _resolve_ctx([ax, a0, ad])
    L.check(L.lib().rdhip_state_diff_map(ctx.handle, _cm(model)._handle, ax.dtype, N, ax.ptr, ldx, a0.ptr, ldx0, code, ad.ptr, space), ctx.handle)
    return None


def nabla_errstate_jacobian_(model, dG, x, xbar, clear=None):
    """nabla-errstate_jacobian!(model, nablaG, x, xbar): src/statevectortype.jl:137-141, src/liestate.jl:300-320.
    Only the 3x3 rotation blocks are written unless clear=True (Euclidean models: nablaG .= 0)."""
    X, ldx, N, single = _states_of(model, x)
    Xb, ldxb, _, _ = _states_of(model, xbar)
    e = errstate_dim(model)
    clear = False if clear is None else bool(clear)
    if single:
        Hb = np.ascontiguousarray(np.asarray(get_data(dG), dtype=X.dtype).T[None])
        nabla_errstate_jacobian_(model, Hb.transpose(0, 2, 1), X, Xb, clear)
        get_data(dG)[...] = Hb[0].T
        return None
    _dense(dG, (N, e, e), "nablaG", transposed_last2=True)
    ax, ab = _Arg(X, "x"), _Arg(Xb, "xbar")
    ah = _Arg(dG, "nablaG", ax.dtype)
    ctx, space = _resolve_ctx([ax, ab, ah])
    L.check(L.lib().rdhip_nabla_errstate_jacobian(ctx.handle, _cm(model)._handle, ax.dtype, N, ax.ptr, ldx, ab.ptr, ldxb, int(clear), ah.ptr, space), ctx.handle)
    return None


grad_errstate_jacobian_ = nabla_errstate_jacobian_


def nabla2_differential_(dG, model, x, dx):
    """v0.3 nabla^2 differential!(nablaG, model, x, dx) (src/deprecate.jl:2)."""
    return nabla_errstate_jacobian_(model, dG, x, dx)


def propagate_dynamics_(sig, model, z2, z1):
    """propagate_dynamics!(sig, model, z2, z1): src/discrete_dynamics.jl:217-235."""
    setstate_(z2, discrete_dynamics(model, z1))
    return None


def _check_stack(data3, t2, dt2, nz):
    """The dict form {"data", "t", "dt"} of a trajectory stack goes straight to the C ABI, which reads t / dt as `const double*` and
    indexes `data` as a dense slab: enforce what PackedTrajectory enforces (float64 times, C-contiguous arrays of matching shapes)."""
    if len(data3.shape) != 3 or data3.shape[2] != nz:
        raise ValueError(f"trajectory stack: data must be (.., .., n+m = {nz}), got {tuple(data3.shape)}")
    _dense(data3, tuple(data3.shape), "Z.data")
    lead = tuple(data3.shape[:2])
    for name, a in (("Z.t", t2), ("Z.dt", dt2)):
        if _is_torch(a) != _is_torch(data3) and not (isinstance(a, np.ndarray) and isinstance(data3, np.ndarray)):
            raise ValueError(f"{name}: must be the same kind of array as Z.data")
        is64 = (a.dtype == torch.float64) if _is_torch(a) else (getattr(a, "dtype", None) == np.float64)
        if not is64:
            raise ValueError(f"{name}: times are always float64 (KnotPoint.t / .dt are Float64, src/knotpoint.jl:150-151), got {getattr(a, 'dtype', type(a))}")
        _dense(a, lead, name)


def rollout_(sig, model, Z, x0=None, time_major=False):
    """rollout!(sig, model, Z, x0): src/trajectories.jl:436-441.  Z is ONE PackedTrajectory (K knots) or a
    (M, K, n+m) stack of trajectories that share the kernel launch (sequential in k, parallel over trajectories); with
    time_major=True the stack is (K, M, n+m) / (K, M): knot k of all trajectories contiguous (explicit rules)."""
    cm, rule = _model_rule(model)
    if rule == L.CONTINUOUS:
        raise TypeError("rollout! needs a DiscreteDynamics model")
    if isinstance(Z, PackedTrajectory):
        _check_traj(model, Z)
        K, M = len(Z), 1
        data3, t2, dt2 = Z.data, Z.t, Z.dt
    else:                       # {"data": (M, K, n+m), "t": (M, K), "dt": (M, K)}
        data3, t2, dt2 = Z["data"], Z["t"], Z["dt"]
        _check_stack(data3, t2, dt2, state_dim(model) + control_dim(model))
        M, K = (data3.shape[1], data3.shape[0]) if time_major else (data3.shape[0], data3.shape[1])
    x0a = x0
    if x0a is not None and not _is_torch(x0a):
        x0a = np.ascontiguousarray(np.asarray(x0a, dtype=data3.dtype if isinstance(data3, np.ndarray) else np.float64).reshape(M, -1))
    if x0a is not None and tuple(x0a.shape) != (M, state_dim(model)) and tuple(x0a.shape) != (state_dim(model),):
        raise ValueError("x0 has the wrong shape")
    az, at, adt, a0 = _Arg(data3, "Z.data"), _Arg(t2, "Z.t"), _Arg(dt2, "Z.dt"), _Arg(x0a, "x0")
    if a0.ptr is not None and a0.dtype != az.dtype:
        raise ValueError("x0 dtype does not match the trajectory")
    ctx, space = _resolve_ctx([az, at, adt, a0])
    L.check(L.lib().rdhip_rollout_layout(ctx.handle, cm._handle, rule, az.dtype, K, M, 1 if time_major and M > 1 else 0, az.ptr, at.ptr, adt.ptr, a0.ptr, space), ctx.handle)
    return None


def rollout_split_(sig, model, X, U, t, dt, x0=None, time_major=False):
    """rollout!(sig, model, Z, x0) with states and controls in arrays of their own (rdhip_rollout_split; src/trajectories.jl:436-441): X (M, K, n) receives
    every simulated state (knot 0 = x0, or what X holds there when x0 is None), U (M, K-1, m) is read, t / dt (M, K); with time_major=True the arrays are
    (K, M, n), (K-1, M, m), (K, M).  Same arithmetic as rollout_; a big batch moves its algorithmic bytes (the packed stack's rollout moves 1.9 x)."""
    cm, rule = _model_rule(model)
    if rule == L.CONTINUOUS:
        raise TypeError("rollout! needs a DiscreteDynamics model")
    n, m = state_dim(model), control_dim(model)
    if X.ndim != 3 or U.ndim != 3 or X.shape[2] != n or U.shape[2] != m:
        raise ValueError(f"X must be (.., .., {n}) and U (.., .., {m})")
    K, M = (X.shape[0], X.shape[1]) if time_major else (X.shape[1], X.shape[0])
    want_u = (max(K - 1, 0), M, m) if time_major else (M, max(K - 1, 0), m)
    if tuple(U.shape) != want_u:
        raise ValueError(f"U has shape {tuple(U.shape)}, expected {want_u}")
    if tuple(t.shape) != tuple(X.shape[:2]) or tuple(dt.shape) != tuple(X.shape[:2]):
        raise ValueError("t and dt must have one entry per knot point of X")
    for a, name in ((X, "X"), (U, "U"), (t, "t"), (dt, "dt")):
        if not (a.is_contiguous() if _is_torch(a) else a.flags["C_CONTIGUOUS"]):
            raise ValueError(f"{name} must be contiguous")
    x0a = x0
    if x0a is not None and not _is_torch(x0a):
        x0a = np.ascontiguousarray(np.asarray(x0a, dtype=X.dtype if isinstance(X, np.ndarray) else np.float64).reshape(M, -1))
    if x0a is not None and tuple(x0a.shape) != (M, n):
        raise ValueError("x0 has the wrong shape")
    ax, au, at, adt, a0 = _Arg(X, "X"), _Arg(U, "U", None), _Arg(t, "t"), _Arg(dt, "dt"), _Arg(x0a, "x0")
    if au.ptr is not None and au.dtype != ax.dtype or (a0.ptr is not None and a0.dtype != ax.dtype):
        raise ValueError("X, U and x0 must share one dtype")
    ctx, space = _resolve_ctx([ax, au, at, adt, a0])
    L.check(L.lib().rdhip_rollout_split(ctx.handle, cm._handle, rule, ax.dtype, K, M, 1 if time_major else 0, ax.ptr, au.ptr, at.ptr, adt.ptr, a0.ptr, space), ctx.handle)
    return None


def rollout_jacobian_(sig, diff, model, J, Z, x0=None, time_major=False):
    """rollout!(sig, model, Z, x0) followed by jacobian!(sig, diff, model, J[k], y, Z[k]) at every knot point of the simulated
    trajectories, in ONE call (the forward pass + linearisation of a shooting method).  Z = {"data", "t", "dt"}: a stack of M
    trajectories of K knot points, (M, K, n+m) / (M, K) or, with time_major=True, (K, M, n+m) / (K, M).  J: (K*M, n, n+m) from
    alloc_jacobians, in the knot order of Z.  Equivalent to rollout_ + jacobian_ on the flattened stack."""
    if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
        raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
    cm, rule = _model_rule(model)
    if rule == L.CONTINUOUS:
        raise TypeError("rollout! needs a DiscreteDynamics model")
    if rule == L.IMPLICIT_MIDPOINT:
        if not isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError(f"Must call jacobian! on an implicit model with the ImplicitFunctionTheorem diff method. Got {type(diff).__name__}.")
        diff = diff.diff
    elif isinstance(diff, ImplicitFunctionTheorem):
        raise ValueError("ImplicitFunctionTheorem needs an implicit integrator")
    data3, t2, dt2 = Z["data"], Z["t"], Z["dt"]
    n, m = state_dim(model), control_dim(model)
    _check_stack(data3, t2, dt2, n + m)
    K, M = (data3.shape[0], data3.shape[1]) if time_major else (data3.shape[1], data3.shape[0])
    x0a = x0
    if x0a is not None and not _is_torch(x0a):
        x0a = np.ascontiguousarray(np.asarray(x0a, dtype=data3.dtype if isinstance(data3, np.ndarray) else np.float64).reshape(M, -1))
    if x0a is not None and tuple(x0a.shape) != (M, n):
        raise ValueError("x0 has the wrong shape")
    _dense(J, (K * M, n, n + m), "J", transposed_last2=True)
    az, at, adt, a0 = _Arg(data3, "Z.data"), _Arg(t2, "Z.t"), _Arg(dt2, "Z.dt"), _Arg(x0a, "x0")
    aj = _Arg(J, "J", az.dtype)
    if a0.ptr is not None and a0.dtype != az.dtype:
        raise ValueError("x0 dtype does not match the trajectory")
    ctx, space = _resolve_ctx([az, at, adt, a0, aj])
    L.check(L.lib().rdhip_rollout_jacobian(ctx.handle, cm._handle, rule, diff.code, az.dtype, K, M, 1 if time_major else 0,
                                           az.ptr, at.ptr, adt.ptr, a0.ptr, aj.ptr, space), ctx.handle)
    k = C.c_int(0)
    L.check(L.lib().rdhip_ctx_last_rollout_jacobian_launches(ctx.handle, C.byref(k)), ctx.handle)
    return k.value          # kernel launches issued: 1 = rollout and Jacobians overlapped in one launch, 2 (or more) = one after the other


class PreparedRolloutJacobianMulti:
    """rollout! + jacobian! along the rollouts for several MODELS in one call (rdhip_rollout_jacobian_multi): the mixed-model Monte-Carlo batch.
    models / Js / Zs / x0s list one entry per model (Zs[p] = {"data", "t", "dt"} stacks as for rollout_jacobian_).  Argument checking happens once;
    calling the object re-issues the call (array contents may change, their memory may not)."""

    def __init__(self, sig, diff, models, Js, Zs, x0s=None, time_major=False, ctx=None):
        """ctx: an rdhip Context that owns its stream to issue the call on instead of torch's current stream (see PreparedJacobian)."""
        if not isinstance(sig, FunctionSignature) or not isinstance(diff, DiffMethod):
            raise TypeError("sig must be a FunctionSignature and diff a DiffMethod")
        self._own_ctx = ctx
        nm = len(models)
        if not (len(Js) == len(Zs) == nm) or (x0s is not None and len(x0s) != nm):
            raise ValueError("models, Js, Zs (and x0s) must list one entry per model")
        if nm > 4:
            raise ValueError("at most four models per call")
        handles, rules = (C.c_void_p * max(nm, 1))(), (C.c_int * max(nm, 1))()
        Ks, Ms = (C.c_longlong * max(nm, 1))(), (C.c_longlong * max(nm, 1))()
        tabs = [(C.c_void_p * max(nm, 1))() for _ in range(5)]
        args, dtype = [], None
        for p, (model, J, Z) in enumerate(zip(models, Js, Zs)):
            cm, rule = _model_rule(model)
            if rule in (L.CONTINUOUS, L.IMPLICIT_MIDPOINT):
                raise TypeError("rollout_jacobian_multi_ takes explicit DiscreteDynamics models")
            data3, t2, dt2 = Z["data"], Z["t"], Z["dt"]
            n, m = state_dim(model), control_dim(model)
            _check_stack(data3, t2, dt2, n + m)
            K, M = (data3.shape[0], data3.shape[1]) if time_major else (data3.shape[1], data3.shape[0])
            x0 = None if x0s is None else x0s[p]
            if x0 is not None and tuple(x0.shape) != (M, n):
                raise ValueError(f"x0s[{p}] has the wrong shape")
            _dense(J, (K * M, n, n + m), f"Js[{p}]", transposed_last2=True)
            az, at, adt, a0 = _Arg(data3, f"Zs[{p}].data"), _Arg(t2, f"Zs[{p}].t"), _Arg(dt2, f"Zs[{p}].dt"), _Arg(x0, f"x0s[{p}]")
            dtype = az.dtype if dtype is None else dtype
            if az.dtype != dtype or (a0.ptr is not None and a0.dtype != dtype):
                raise ValueError("all stacks must share one dtype")
            aj = _Arg(J, f"Js[{p}]", dtype)
            args += [az, at, adt, a0, aj]
            handles[p], rules[p], Ks[p], Ms[p] = cm._handle, rule, K, M
            for tab, a in zip(tabs, (az, at, adt, a0, aj)):
                tab[p] = a.ptr
        if isinstance(diff, ImplicitFunctionTheorem):
            raise ValueError("ImplicitFunctionTheorem needs an implicit integrator")
        ctx, space = _resolve_ctx(args) if args else (None, L.HOST)
        self._keep = (models, Js, Zs, x0s, handles, rules, Ks, Ms, tabs)
        self._space, self._dev, self._ctx, self._nm = space, (args[0].device if args else None), ctx, nm
        self._args = (nm, handles, rules, diff.code, dtype if dtype is not None else L.F64, Ks, Ms, 1 if time_major else 0, *tabs, space)

    def __call__(self):
        if self._nm == 0:
            return
        if self._own_ctx is not None and self._space == L.DEVICE:
            ctx = self._own_ctx
        else:
            ctx = L.context_for(self._dev, torch.cuda.current_stream(self._dev).cuda_stream) if self._space == L.DEVICE else self._ctx
        rc = L.lib().rdhip_rollout_jacobian_multi(ctx.handle, *self._args)
        if rc:
            L.check(rc, ctx.handle)
        self._last_ctx = ctx

    @property
    def launches(self):
        """kernel launches the last call issued ON THE CONTEXT'S STREAM (rdhip_ctx_last_rollout_jacobian_launches): 1 = the whole call was one launch
        (csrc/mixed.hpp, or the fused launch of the model that stayed on the context's stream while the others ran on internal streams)"""
        ctx = getattr(self, "_last_ctx", None)
        if ctx is None:
            return 0
        k = C.c_int(0)
        L.check(L.lib().rdhip_ctx_last_rollout_jacobian_launches(ctx.handle, C.byref(k)), ctx.handle)
        return k.value


def rollout_jacobian_multi_(sig, diff, models, Js, Zs, x0s=None, time_major=False):
    """One call for the rollouts and the Jacobians along them of several models (see PreparedRolloutJacobianMulti)."""
    PreparedRolloutJacobianMulti(sig, diff, models, Js, Zs, x0s, time_major)()
    return None


def maxdiff(a, b):   # src/utils.jl:16-22
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))
