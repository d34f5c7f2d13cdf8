"""Host-side state helpers of the rdhip mirror (no GPU, no kernels): gen_inds / parse_state / build_state / fill_state / orientation
(reference src/rigidbody.jl:80-207, pinned by test/rigidbody_test.jl:88-125) and vec_states / rot_states (src/liestate.jl:142-173, pinned
by examples/satellite_pair.jl usage in test/liestate.jl).  The rotation conversions are checked against scipy's rotation algebra."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as SR

import rdhip as RD


def rand_quat(rng):
    q = rng.standard_normal(4)
    return q / np.linalg.norm(q)


def test_gen_inds_and_getters_like_rigidbody_test():
    rng = np.random.default_rng(0)
    model = RD.Body()
    x = np.concatenate([rng.random(3), rand_quat(rng), rng.random(6)])
    inds = RD.gen_inds(model)
    assert list(inds["q"]) == [4, 5, 6, 7] and list(inds["u"]) == list(range(14, 20)) and list(inds["r"]) == [1, 2, 3]   # test/rigidbody_test.jl:88-92
    model2 = RD.Body(RD.MRP)
    i2 = RD.gen_inds(model2)
    assert list(i2["q"]) == [4, 5, 6] and list(i2["v"]) == [7, 8, 9] and list(i2["u"]) == list(range(13, 19))           # :94-98
    r, q, v, w = RD.parse_state(model, x)                                                                                # :101-106
    assert np.array_equal(r, x[0:3]) and np.array_equal(RD.params(q), x[3:7]) and np.array_equal(v, x[7:10]) and np.array_equal(w, x[10:13])
    assert isinstance(RD.orientation(model, x), RD.QuatRotation)
    np.testing.assert_array_equal(RD.position(model, x), x[0:3])
    np.testing.assert_array_equal(RD.linear_velocity(model, x), x[7:10])
    np.testing.assert_array_equal(RD.angular_velocity(model, x), x[10:13])
    # parse_state does not renormalise unless asked to (src/rigidbody.jl:101-104)
    x2 = x.copy(); x2[3:7] *= 2.0
    np.testing.assert_array_equal(RD.params(RD.orientation(model, x2)), x2[3:7])
    np.testing.assert_allclose(RD.params(RD.orientation(model, x2, True)), x[3:7], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(RD.flipquat(model, x), np.concatenate([x[:3], -x[3:7], x[7:]]))


def test_build_state_and_fill_state_like_rigidbody_test():
    rng = np.random.default_rng(1)
    model, model2, model3 = RD.Body(), RD.Body(RD.MRP), RD.Body(RD.RodriguesParam)
    r, qv, v, w = rng.random(3), rand_quat(rng), rng.random(3), rng.random(3)
    x = np.concatenate([r, qv, v, w])
    q = RD.QuatRotation(*qv)
    np.testing.assert_allclose(RD.build_state(model, r, q, v, w), x, rtol=0, atol=1e-15)                 # test/rigidbody_test.jl:109
    np.testing.assert_allclose(RD.build_state(model, r, RD.MRP(q), v, w), x, rtol=0, atol=1e-15)         # :110-111
    np.testing.assert_allclose(RD.build_state(model, x), x, rtol=0, atol=0)                              # :112-114
    g = RD.params(RD.MRP(q))
    x2 = np.concatenate([r, g, v, w])
    np.testing.assert_allclose(RD.build_state(model2, r, q, v, w), x2, rtol=0, atol=1e-15)               # :116-120
    np.testing.assert_allclose(RD.build_state(model2, r, RD.MRP(q), v, w), x2, rtol=0, atol=0)
    np.testing.assert_allclose(RD.build_state(model2, r, g, v, w), x2, rtol=0, atol=0)
    np.testing.assert_allclose(RD.build_state(model2, r, qv, v, w), x2, rtol=0, atol=1e-15)              # a quaternion given as a vector
    with pytest.raises(AssertionError):
        RD.build_state(model, r, g, v, w)                                                                # @assert length(q) == 4
    # conversions against scipy (scalar-first quaternion here, scalar-last there)
    sr = SR.from_quat(np.concatenate([qv[1:], qv[:1]]))
    np.testing.assert_allclose(SR.from_mrp(g).as_matrix(), sr.as_matrix(), rtol=0, atol=1e-13)     # same rotation (v/(1+w) may be the shadow MRP set)
    if qv[0] > 0:
        np.testing.assert_allclose(g, sr.as_mrp(), rtol=0, atol=1e-14)
    gr = RD.params(RD.RodriguesParam(q))
    np.testing.assert_allclose(gr, qv[1:] / qv[0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(RD.build_state(model3, r, q, v, w), np.concatenate([r, gr, v, w]), rtol=0, atol=1e-15)
    for back in (RD.QuatRotation(RD.MRP(q)), RD.QuatRotation(RD.RodriguesParam(q))):
        qb = RD.params(back)
        np.testing.assert_allclose(qb * np.sign(qb[0] * qv[0]), qv, rtol=0, atol=1e-14)
    f = RD.fill_state(model, 1, 2, 3, 4)                                                                # :122-130
    assert f.shape == (13,) and f.dtype.kind == "i"
    assert RD.fill_state(model2, 1, 2, 3, 4).shape == (12,)
    assert RD.fill_state(model, 1, 2, 3, 4.0).dtype == np.float64
    assert np.all(RD.position(model, f) == 1) and np.all(RD.params(RD.orientation(model, f)) == 2)
    assert np.all(RD.linear_velocity(model, f) == 3) and np.all(RD.angular_velocity(model, f) == 4)


def test_vec_states_and_rot_states_of_a_liestate():
    rng = np.random.default_rng(2)
    model = RD.SatellitePair()
    x = np.concatenate([rand_quat(rng), rng.random(3), rand_quat(rng), rng.random(3)])
    s = RD.LieState(model)
    vs, rs = RD.vec_states(s, x), RD.rot_states(s, x)
    assert [len(v) for v in vs] == [0, 3, 3]                                                            # P = (0, 3, 3)
    np.testing.assert_array_equal(vs[1], x[4:7]); np.testing.assert_array_equal(vs[2], x[11:14])
    np.testing.assert_array_equal(RD.params(rs[0]), x[0:4]); np.testing.assert_array_equal(RD.params(rs[1]), x[7:11])
    assert len(RD.vec_states(model, x)) == 3 and len(RD.rot_states(model, x)) == 2
    s2 = RD.LieState(RD.MRP, (0, 3, 3))
    y = rng.random(12)
    r2 = RD.rot_states(s2, y)
    assert isinstance(r2[0], RD.MRP)
    np.testing.assert_array_equal(RD.params(r2[1]), y[6:9])
    np.testing.assert_array_equal(RD.vec_states(s2, y)[2], y[9:12])
