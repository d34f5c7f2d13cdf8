import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotdynamics.jl_amd")]
import rdhip as RD
from oracle import rd_oracle as O
from tests import helpers as H
M, K = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
model, om = RD.Satellite(), O.satellite()
dm = RD.DiscretizedDynamics(model, RD.RK4)
sig, AD = RD.StaticReturn(), RD.ForwardAD()
data = H.rand_rigid(rng, M * K, 6).reshape(M, K, 19)
t = np.broadcast_to(np.arange(K) * 0.02, (M, K)).copy(); dt = np.full((M, K), 0.02); dt[:, -1] = 0
ref = om.rollout(O.RK4, data, t, dt)
st1 = {"data": torch.as_tensor(np.ascontiguousarray(data.transpose(1, 0, 2))).cuda(), "t": torch.as_tensor(np.ascontiguousarray(t.T)).cuda(), "dt": torch.as_tensor(np.ascontiguousarray(dt.T)).cuda()}
J1 = RD.alloc_jacobians(dm, K * M, st1["data"])
print("launches", RD.rollout_jacobian_(sig, AD, dm, J1, st1, None, time_major=True))
st2 = {"data": torch.as_tensor(data).cuda(), "t": torch.as_tensor(t).cuda(), "dt": torch.as_tensor(dt).cuda()}
RD.rollout_(sig, dm, st2, None)
a = st1["data"].cpu().numpy().transpose(1, 0, 2); b = st2["data"].cpu().numpy()
print("fused vs oracle", np.abs(a - ref).max(), "two-call vs oracle", np.abs(b - ref).max(), "fused vs two-call", np.abs(a - b).max())
d = np.abs(a - b)
idx = np.argwhere(d > 0)
print("mismatching entries", len(idx), "of", d.size, "first", idx[:10].tolist())
if len(idx):
    print("by knot:", np.unique(idx[:, 1], return_counts=True))
    print("by comp:", np.unique(idx[:, 2], return_counts=True))
