#!/usr/bin/env python3
"""tools/host_overhead.py -- per-call host cost of the C ABI vs the Python mirror (what a Julia `ccall` host would / would not pay).
Launches the N = 4096 cartpole RK4 Jacobian (a ~9 us kernel) back to back: (a) through rdhip.jacobian_ (argument checking in
Python), (b) through a bare ctypes call with pre-resolved pointers, (c) an empty torch op for scale."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rdhip as RD  # noqa: E402
from rdhip import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = RD.Cartpole()
dm = RD.DiscretizedDynamics(model, RD.RK4)
rng = np.random.default_rng(0)
Z = RD.PackedTrajectory(4, 1, rng.random((N, 5)), np.arange(N) * 0.1, np.full(N, 0.1)).to("cuda:0")
J = RD.alloc_jacobians(dm, N, Z.data)
Y = torch.empty((N, 4), dtype=torch.float64, device="cuda:0")
sig, diff = RD.StaticReturn(), RD.ForwardAD()
reps = 2000


def timeit(f):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_issue / reps * 1e6, (time.perf_counter() - t0) / reps * 1e6


a = timeit(lambda: RD.jacobian_(sig, diff, dm, J, Y, Z))
ctx = L.context_for(0, torch.cuda.current_stream(0).cuda_stream)
lib = L.lib()
args = (ctx.handle, model._handle, L.RK4, L.FORWARD_AD, L.F64, N, Z.data.data_ptr(), Z.t.data_ptr(), Z.dt.data_ptr(), J.data_ptr(), Y.data_ptr(), L.DEVICE)
b = timeit(lambda: lib.rdhip_jacobian(*args))
pj = RD.prepare_jacobian(sig, diff, dm, J, Y, Z)
p = timeit(pj)
print(f"N={N}: rdhip.prepare_jacobian(...)()   host issue {p[0]:.2f} us/call, sustained {p[1]:.2f} us/call")
x = torch.zeros(1, device="cuda:0")
c = timeit(lambda: x.add_(1))
print(f"N={N}: rdhip.jacobian_ (Python mirror) host issue {a[0]:.2f} us/call, sustained {a[1]:.2f} us/call")
print(f"N={N}: bare C call via ctypes          host issue {b[0]:.2f} us/call, sustained {b[1]:.2f} us/call")
print(f"       torch x.add_(1)                 host issue {c[0]:.2f} us/call, sustained {c[1]:.2f} us/call")
