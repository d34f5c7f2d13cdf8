#!/usr/bin/env python3
"""Round-5 fixtures: the inputs (and definition-derived expected values) of the quantities tests/golden/make_golden_reference.jl's discrete_case,
user_jacobian_case and nan_case evaluate with the real RobotDynamics.jl.  Independent of the oracle and of the kernels (imports neither):
  discrete_di.npz       the reference's discrete double integrator (test/discrete_dynamics_test.jl:5-63): x+ and the closed-form Jacobian of :52-63,
                        dynamics_error = x+(z1) - x2 and its Jacobians J1 = d x+ / d z1, J2 = [-I 0] (:65-73)
  userjac_cartpole.npz  what UserDefined() must return for the cartpole's hand-written jacobian! (test/cartpole_model.jl:57-96): the exact
                        Jacobian of the dynamics (continuous) and of each explicit rule's step (the integrators' chain rule on stage Jacobians
                        IS the derivative of the step, src/integration.jl:149-177, 302-337) -- torch forward-mode on the definitions
  nan_satellite.npz     a satellite batch with one NaN-filled knot point (src/trajectories.jl:83) and dt = Inf on the last (src/trajectories.jl:104-108):
                        expected = the clean batch's values everywhere else, NaN at the poisoned knot
Run:  python tests/golden/make_golden_round5.py"""
import os

import numpy as np
import torch
from torch.func import jacfwd

import make_golden as G

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


def main():
    rng = np.random.default_rng(505)
    # --- discrete double integrator
    N = 20
    Z = rng.random((N, 6)); t = np.arange(N) * 0.1; dt = 0.05 + 0.1 * rng.random(N)
    h = dt[:, None]
    xn = np.concatenate([Z[:, 0:2] + Z[:, 2:4] * h, Z[:, 2:4] + Z[:, 4:6] * h], axis=1)
    J = np.zeros((N, 4, 6))
    for i in range(4):
        J[:, i, i] = 1.0
        J[:, i, i + 2] = dt
    J2 = np.zeros((N - 1, 4, 6)); J2[:, range(4), range(4)] = -1.0
    np.savez(os.path.join(HERE, "discrete_di.npz"), Z=Z, t=t, dt=dt, xn=xn, J_user=J, J_ad=J, e=xn[:-1] - Z[1:, :4], J1=J[:-1], J2=J2)
    # --- cartpole: user Jacobian = the Jacobian
    p = [1.0, 0.2, 0.5, 9.81]
    f, n, m = G.cartpole(p)
    N = 16
    Z = rng.random((N, 5)) - 0.3; t = np.arange(N) * 0.1; dt = np.full(N, 0.1)
    ev = G.evaluate_case(f, n, m, Z, dt)
    out = {"Z": Z, "t": t, "dt": dt, "params": np.asarray(p), "Jc_user": ev["Jc"]}
    for rule in G.RULES:
        out["Jd_user_" + rule] = ev["Jd_" + rule]
    np.savez(os.path.join(HERE, "userjac_cartpole.npz"), **out)
    # --- NaN knot point / dt = Inf
    mass, Jm = 2.0, np.diag([2.0, 3.0, 1.0])
    f, n, m = G.rigid_body(mass, torch.as_tensor(Jm), G.satellite_wrench, 6, False)
    N = 12
    Zc = G.rand_rigid(rng, N, 6); t = np.arange(N) * 0.05; dt = np.full(N, 0.05)
    ev = G.evaluate_case(f, n, m, Zc, dt)
    Z = Zc.copy(); Z[5, :] = np.nan                       # a knot point as Traj(n, m, dt, N) creates it: NaN-filled
    dti = dt.copy(); dti[-1] = np.inf
    xn, Jd = ev["xn_rk4"].copy(), ev["Jd_rk4"].copy()
    xn[5] = np.nan; Jd[5] = np.nan
    xn[-1] = np.nan; Jd[-1] = np.nan                      # x + Inf * f: not specified here (the reference's value is whatever IEEE gives: Inf / NaN); masked by the tests
    np.savez(os.path.join(HERE, "nan_satellite.npz"), Z=Z, t=t, dt=dti, mass=np.asarray(mass), J=Jm, frame=np.asarray(0), xn_rk4=xn, Jd_rk4=Jd,
             poisoned=np.asarray([5]), inf_dt=np.asarray([N - 1]))
    print("wrote discrete_di.npz, userjac_cartpole.npz, nan_satellite.npz")


if __name__ == "__main__":
    main()
