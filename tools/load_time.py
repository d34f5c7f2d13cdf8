#!/usr/bin/env python3
"""tools/load_time.py -- what loading librdhip.so costs a host process (the library is ~90 MB: one code object per (model family, rule, dtype, diff method)):
dlopen, the first context, the first launch of a kernel (HIP loads a code object when its first kernel is launched), a second launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
t0 = time.perf_counter()
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
t1 = time.perf_counter()
import rdhip as RD
from rdhip import _lib as L
lib = L.lib()
t2 = time.perf_counter()
ctx = L.Context(0)
t3 = time.perf_counter()
N = 4096
rng = np.random.default_rng(0)
q = rng.standard_normal((N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
Zn = np.concatenate([rng.random((N, 3)), q, rng.random((N, 12))], axis=1)
dm = RD.DiscretizedDynamics(RD.Satellite(), RD.RK4)
Z = RD.PackedTrajectory(13, 6, Zn, np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
J = RD.alloc_jacobians(dm, N, Z.data)
torch.cuda.synchronize()
t4 = time.perf_counter()
RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, None, Z); torch.cuda.synchronize()
t5 = time.perf_counter()
RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), dm, J, None, Z); torch.cuda.synchronize()
t6 = time.perf_counter()
Jc = RD.alloc_jacobians(RD.DiscretizedDynamics(RD.Cartpole(), RD.RK4), N, Z.data)
Zc = RD.PackedTrajectory(4, 1, rng.random((N, 5)), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
t7 = time.perf_counter()
RD.jacobian_(RD.StaticReturn(), RD.ForwardAD(), RD.DiscretizedDynamics(RD.Cartpole(), RD.RK4), Jc, None, Zc); torch.cuda.synchronize()
t8 = time.perf_counter()
print(f"librdhip.so {os.path.getsize(L.LIB_PATH) / 1e6:.1f} MB: torch + HIP runtime up {t1 - t0:.2f} s | dlopen + symbol table {1e3 * (t2 - t1):.1f} ms | first context {1e3 * (t3 - t2):.1f} ms | "
      f"first satellite launch {1e3 * (t5 - t4):.1f} ms, second {1e3 * (t6 - t5):.2f} ms | first launch of another family (cartpole) {1e3 * (t8 - t7):.1f} ms")
