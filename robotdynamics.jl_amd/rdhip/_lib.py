"""ctypes binding of librdhip.so (include/rdhip.h).  No torch types cross this boundary: only raw pointers and sizes.

The library is built in-tree (robotdynamics.jl_amd/lib/librdhip.so) by `make -C robotdynamics.jl_amd/csrc` or
`__graft_entry__.build()`.  There is no CPU fallback anywhere in this package: if the library is missing, or no GPU
is present, the compute entry points raise.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("RDHIP_LIB") or os.path.join(_PKG, "lib", "librdhip.so")      # RDHIP_LIB: A/B builds of the library (tuning)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "rdhip.h")

# rdhip_status
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_NOT_IMPLEMENTED, ERR_HIP, ERR_NO_DEVICE, ERR_DIM_MISMATCH = 0, -1, -2, -3, -4, -5, -6
# rdhip_model_kind
MODEL_CARTPOLE, MODEL_DOUBLE_INTEGRATOR, MODEL_LINEAR, MODEL_SATELLITE, MODEL_QUADROTOR, MODEL_SATELLITE_PAIR, MODEL_LIE_SATELLITE, MODEL_CUSTOM = range(8)
ROT_QUAT, ROT_MRP, ROT_RP = range(3)
FRAME_WORLD, FRAME_BODY = range(2)
ERRMAP_CAYLEY, ERRMAP_EXPONENTIAL, ERRMAP_MRP, ERRMAP_QUATVEC, ERRMAP_IDENTITY = range(5)
CONTINUOUS, EULER, RK2, RK3, RK4, IMPLICIT_MIDPOINT, DISCRETE = -1, 0, 1, 2, 3, 4, 5
FORWARD_AD, FINITE_DIFF, USER_DEFINED = range(3)
F64, F32 = range(2)
DEVICE, HOST = range(2)
STREAM_OWN, STREAM_BORROW = range(2)
GATHER_NONE, GATHER_ROOT, GATHER_ALL = range(3)
COMM_ID_BYTES = 128


class RdhipError(RuntimeError):
    """A librdhip call failed (HIP error, missing library, no GPU)."""

    def __init__(self, code, msg):
        super().__init__(f"librdhip status {code}: {msg}")
        self.code = code


class NoDeviceError(RdhipError):
    pass


_lib = None

_P = C.c_void_p
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)
_LL = C.c_longlong
_SIGS = {
    "rdhip_version": (C.c_int, []),
    "rdhip_device_count": (C.c_int, [_IP]),
    "rdhip_ctx_create": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "rdhip_ctx_destroy": (C.c_int, [_P]),
    "rdhip_ctx_synchronize": (C.c_int, [_P]),
    "rdhip_ctx_set_overlap": (C.c_int, [_P, C.c_int]),
    "rdhip_ctx_fence": (C.c_int, [_P]),
    "rdhip_ctx_stream": (_P, [_P]),
    "rdhip_last_error": (C.c_char_p, [_P]),
    "rdhip_last_kernel_name": (C.c_int, [C.c_char_p, C.c_int]),
    "rdhip_timer_start": (C.c_int, [_P]),
    "rdhip_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "rdhip_malloc": (C.c_int, [_P, C.POINTER(_P), C.c_ulonglong]),
    "rdhip_free": (C.c_int, [_P, _P]),
    "rdhip_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_ulonglong]),
    "rdhip_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_ulonglong]),
    "rdhip_model_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _DP, C.c_int, C.POINTER(_P)]),
    "rdhip_model_destroy": (C.c_int, [_P]),
    "rdhip_model_create_custom": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, _DP, C.c_int, C.c_int, _IP, C.c_int, C.POINTER(_P)]),
    "rdhip_model_precompile": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "rdhip_model_set_newton": (C.c_int, [_P, C.c_int, C.c_double]),
    "rdhip_model_balance_columns": (C.c_int, [_P]),
    "rdhip_model_set_cascade": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rdhip_model_create_custom_discrete": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, _DP, C.c_int, C.c_int, _IP, C.c_int, C.POINTER(_P)]),
    "rdhip_model_set_user_jacobian": (C.c_int, [_P, C.c_char_p]),
    "rdhip_model_is_discrete": (C.c_int, [_P, _IP]),
    "rdhip_model_dims": (C.c_int, [_P, _IP, _IP, _IP, _IP]),
    "rdhip_model_statevectortype": (C.c_int, [_P, _IP]),
    "rdhip_model_liestate": (C.c_int, [_P, _IP, _IP, _IP]),
    "rdhip_model_gen_inds": (C.c_int, [_P, _IP, _IP]),
    "rdhip_liestate_query": (C.c_int, [C.c_int, _IP, C.c_int, _IP, _IP, _IP, _IP]),
    "rdhip_dynamics": (C.c_int, [_P, _P, C.c_int, _LL, _P, _P, _P, _P, C.c_int]),
    "rdhip_discrete_dynamics": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _P, _P, _P, _P, C.c_int]),
    "rdhip_jacobian": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _LL, _P, _P, _P, _P, _P, C.c_int]),
    "rdhip_jacobian_multi": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_LL), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.c_int]),
    "rdhip_ctx_last_multi_launches": (C.c_int, [_P, _IP]),
    "rdhip_dynamics_error": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _P, _P, _P, _P, C.c_int]),
    "rdhip_dynamics_error_jacobian": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _LL, _P, _P, _P, _P, _P, _P, C.c_int]),
    "rdhip_error_expansion": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _P, _P, _P, _P, _P, C.c_int]),
    "rdhip_errstate_jacobian": (C.c_int, [_P, _P, C.c_int, _LL, _P, C.c_int, C.c_int, _P, C.c_int]),
    "rdhip_tune": (C.c_int, [C.c_char_p]),
    "rdhip_state_diff": (C.c_int, [_P, _P, C.c_int, _LL, _P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "rdhip_state_diff_map": (C.c_int, [_P, _P, C.c_int, _LL, _P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int]),
    "rdhip_nabla_errstate_jacobian": (C.c_int, [_P, _P, C.c_int, _LL, _P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int]),
    "rdhip_graph_begin_capture": (C.c_int, [_P]),
    "rdhip_graph_end_capture": (C.c_int, [_P, C.POINTER(_P)]),
    "rdhip_graph_launch": (C.c_int, [_P, _P]),
    "rdhip_graph_destroy": (C.c_int, [_P]),
    "rdhip_shard_range": (C.c_int, [_LL, C.c_int, C.c_int, C.POINTER(_LL), C.POINTER(_LL)]),
    "rdhip_jacobian_sharded": (C.c_int, [C.POINTER(_P), C.c_int, _P, C.c_int, C.c_int, C.c_int, _LL, _P, _P, _P, _P, _P]),
    "rdhip_memcpy_peer": (C.c_int, [_P, _P, _P, _P, C.c_ulonglong]),
    "rdhip_comm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "rdhip_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rdhip_comm_create_rank": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.POINTER(_P)]),
    "rdhip_comm_destroy": (C.c_int, [_P]),
    "rdhip_comm_size": (C.c_int, [_P, _IP, _IP]),
    "rdhip_comm_rank": (C.c_int, [_P, C.c_int, _IP, C.POINTER(_P)]),
    "rdhip_gather": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_ulonglong), C.POINTER(_P)]),
    "rdhip_jacobian_sharded_gather": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _LL, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.c_int, C.c_int, C.POINTER(_P)]),
    "rdhip_errstate_jacobian_sharded_gather": (C.c_int, [_P, _P, C.c_int, _LL, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(_P)]),
    "rdhip_wrenches": (C.c_int, [_P, _P, C.c_int, _LL, _P, _P, _P, C.c_int]),
    "rdhip_wrench_jacobian": (C.c_int, [_P, _P, C.c_int, _LL, _P, _P, _P, C.c_int]),
    "rdhip_model_wrench_sparsity": (C.c_int, [_P, _IP]),
    "rdhip_rollout": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _LL, _P, _P, _P, _P, C.c_int]),
    "rdhip_rollout_layout": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _LL, C.c_int, _P, _P, _P, _P, C.c_int]),
    "rdhip_ctx_last_rollout_jacobian_launches": (C.c_int, [_P, _IP]),
    "rdhip_rollout_jacobian_multi": (C.c_int, [_P, C.c_int, C.POINTER(_P), _IP, C.c_int, C.c_int, C.POINTER(_LL), C.POINTER(_LL), C.c_int,
                                               C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.c_int]),
    "rdhip_rollout_split": (C.c_int, [_P, _P, C.c_int, C.c_int, _LL, _LL, C.c_int, _P, _P, _P, _P, _P, C.c_int]),
    "rdhip_rollout_jacobian": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _LL, _LL, C.c_int, _P, _P, _P, _P, _P, C.c_int]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load librdhip.so; raises RdhipError if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RdhipError(ERR_HIP, f"{LIB_PATH} not found: build it with `make -C robotdynamics.jl_amd/csrc` "
                                      "(or __graft_entry__.build()); this package has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("RDHIP_LIB"):        # an A/B build of an OLDER library (tools/ab_lib.py): entry points added since are simply absent
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, ctx_handle=None):
    if rc == OK:
        return
    msg = lib().rdhip_last_error(ctx_handle)
    msg = msg.decode() if msg else ""
    if rc == ERR_NOT_IMPLEMENTED:
        raise NotImplementedError(msg)            # Julia: NotImplementedError (src/utils.jl:1-8)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc in (ERR_INVALID_ARG, ERR_DIM_MISMATCH):
        raise ValueError(msg)                      # Julia: ArgumentError / DimensionMismatch
    if rc == ERR_NO_DEVICE:
        raise NoDeviceError(rc, msg)
    raise RdhipError(rc, msg)


def last_kernel_name():
    """Demangled name of the kernel this host thread launched last through the library ('' before the first launch)."""
    buf = C.create_string_buffer(1024)
    check(lib().rdhip_last_kernel_name(buf, 1024))
    return buf.value.decode()


def tune(pairs):
    """rdhip_tune: merge "key=value,..." into the dispatcher's tuning knobs ("reset": back to RDHIP_TUNE's); include/rdhip.h lists the keys."""
    check(lib().rdhip_tune(pairs.encode() if isinstance(pairs, str) else pairs))


def device_count():
    n = C.c_int(0)
    rc = lib().rdhip_device_count(C.byref(n))
    return n.value if rc == OK else 0


class Context:
    """One HIP stream on one device (rdhip_ctx).  Not thread-safe; use one per host thread."""

    def __init__(self, device=0, stream=None):
        """stream=None: the context owns a fresh non-blocking stream; otherwise an integer hipStream_t
        (e.g. torch.cuda.current_stream().cuda_stream; 0 = the default stream) that is borrowed."""
        h = _P()
        mode = STREAM_OWN if stream is None else STREAM_BORROW
        check(lib().rdhip_ctx_create(int(device), _P(stream or 0), mode, C.byref(h)))
        self.handle, self.device = h, int(device)

    def synchronize(self):
        check(lib().rdhip_ctx_synchronize(self.handle), self.handle)

    def set_overlap(self, on=True):
        """Launch overlap (contexts that own their stream): consecutive independent rdhip_jacobian calls alternate between two streams."""
        check(lib().rdhip_ctx_set_overlap(self.handle, 1 if on else 0), self.handle)

    def fence(self):
        """Order the context's stream behind everything issued so far (overlap mode keeps part of it on an internal stream)."""
        check(lib().rdhip_ctx_fence(self.handle), self.handle)

    @property
    def stream(self):
        return lib().rdhip_ctx_stream(self.handle) or 0

    def timer_start(self):
        check(lib().rdhip_timer_start(self.handle), self.handle)

    def timer_stop(self):
        ms = C.c_float()
        check(lib().rdhip_timer_stop(self.handle, C.byref(ms)), self.handle)
        return ms.value

    def begin_capture(self):
        """Record the following device-memory calls on this context into a HIP graph instead of executing them."""
        check(lib().rdhip_graph_begin_capture(self.handle), self.handle)

    def end_capture(self):
        g = _P()
        check(lib().rdhip_graph_end_capture(self.handle, C.byref(g)), self.handle)
        return Graph(g)

    def launch(self, graph):
        check(lib().rdhip_graph_launch(self.handle, graph.handle), self.handle)

    def close(self):
        if getattr(self, "handle", None):
            lib().rdhip_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """A captured sequence of calls (rdhip_graph); replay with Context.launch(graph)."""

    def __init__(self, handle):
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            lib().rdhip_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context_for(device, stream):
    """Cached context per (device, borrowed stream); stream=None -> a context with its own stream."""
    key = (int(device), stream)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = Context(device, stream)
    return ctx
