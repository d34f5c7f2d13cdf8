"""Why does a Python host see less of the early-part gain than a C++ host?  Same library call, varying (memory source, stream, data)."""
import os, sys, json, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd"), os.path.join(ROOT, "tools")]
import rdhip as RD
from rdhip import _lib as L
import bench_configs as BC
lib = L.lib()
N = 65536
rng = np.random.default_rng(3)
Zn = BC.rigid_inputs(N, 6, rng)
t, dt = np.arange(N) * 0.1, np.full(N, 0.1)
sat = RD.Satellite()
dev = torch.device("cuda:0")

def run(tag, ctx, zp, tp, dp, Jp, Yp, sync, ev):
    def call(i):
        rc = lib.rdhip_jacobian(ctx.handle, sat._handle, L.RK4, L.FORWARD_AD, L.F64, N, zp, tp, dp, Jp[i % 4], Yp[i % 4], L.DEVICE)
        assert rc == 0
    for i in range(100): call(i)
    sync()
    for rep in range(2):
        us = ev(lambda: [call(i) for i in range(400)]) * 1e3 / 400
        print(tag, round(us, 2), flush=True)

def torch_ev(stream):
    def ev(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    return ev
def own_ev(ctx):
    def ev(fn):
        assert lib.rdhip_timer_start(ctx.handle) == 0
        fn()
        ms = C.c_float(0)
        assert lib.rdhip_timer_stop(ctx.handle, C.byref(ms)) == 0
        return ms.value
    return ev

# (a) torch memory, torch's current (null) stream
Z = torch.as_tensor(Zn).to(dev); tt = torch.as_tensor(t).to(dev); dd = torch.as_tensor(dt).to(dev)
Js = [torch.empty((N, 19, 13), dtype=torch.float64, device=dev) for _ in range(4)]
Ys = [torch.empty((N, 13), dtype=torch.float64, device=dev) for _ in range(4)]
print("J ptr mod 4096:", [j.data_ptr() % 4096 for j in Js], "Y:", [y.data_ptr() % 4096 for y in Ys])
ctx0 = L.context_for(0, torch.cuda.current_stream(dev).cuda_stream)
run("torch mem, null stream ", ctx0, Z.data_ptr(), tt.data_ptr(), dd.data_ptr(), [j.data_ptr() for j in Js], [y.data_ptr() for y in Ys], torch.cuda.synchronize, torch_ev(torch.cuda.current_stream(dev)))
# (b) torch memory, the library's own stream
ctx1 = L.context_for(0, None)
run("torch mem, own stream  ", ctx1, Z.data_ptr(), tt.data_ptr(), dd.data_ptr(), [j.data_ptr() for j in Js], [y.data_ptr() for y in Ys], lambda: lib.rdhip_ctx_synchronize(ctx1.handle), own_ev(ctx1))
# (c) rdhip_malloc memory, own stream
def dmalloc(nbytes):
    p = C.c_void_p(); assert lib.rdhip_malloc(ctx1.handle, C.byref(p), C.c_ulonglong(nbytes)) == 0; return p.value
dZ, dT, dD = dmalloc(N * 19 * 8), dmalloc(N * 8), dmalloc(N * 8)
for d, a in ((dZ, Zn), (dT, t), (dD, dt)):
    a = np.ascontiguousarray(a); assert lib.rdhip_memcpy_h2d(ctx1.handle, C.c_void_p(d), a.ctypes.data_as(C.c_void_p), C.c_ulonglong(a.nbytes)) == 0
lib.rdhip_ctx_synchronize(ctx1.handle)
dJ = [dmalloc(N * 247 * 8) for _ in range(4)]; dY = [dmalloc(N * 13 * 8) for _ in range(4)]
print("J ptr mod 4096:", [j % 4096 for j in dJ])
run("hip mem,   own stream  ", ctx1, dZ, dT, dD, dJ, dY, lambda: lib.rdhip_ctx_synchronize(ctx1.handle), own_ev(ctx1))
run("hip mem,   null stream ", ctx0, dZ, dT, dD, dJ, dY, torch.cuda.synchronize, torch_ev(torch.cuda.current_stream(dev)))
del Js, Ys
torch.cuda.empty_cache()
run("hip mem, own stream, torch pool emptied", ctx1, dZ, dT, dD, dJ, dY, lambda: lib.rdhip_ctx_synchronize(ctx1.handle), own_ev(ctx1))
