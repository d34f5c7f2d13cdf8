"""CPU side of the round-6 fixtures (tests/golden/make_golden_round6.py): the oracle's error maps (oracle/rd_oracle.cpp errmap_inverse, restating
Rotations.jl's inverse maps as src/liestate.jl:216-217 calls them) against values derived from the angle and axis of the relative rotation."""
import numpy as np
import pytest

from oracle import rd_oracle as O
from tests import helpers as H

MAPS = {"cayley": 0, "exponential": 1, "mrp": 2, "quatvec": 3, "identity": 4}
CASES = ["errmap_rigid", "errmap_p323", "errmap_p323_mrp", "errmap_rigid_rp"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_error_maps_against_rotation_geometry(case):
    d = H.load(case)
    M = H.errmap_oracle_model(d)
    for name, code in MAPS.items():
        key = "dX_" + name
        if key not in d:                                  # IdentityMap of a quaternion state: the reference throws, the oracle reports -3
            with pytest.raises(NotImplementedError):
                M.state_diff(d["X"], d["X0"], errmap=code)
            continue
        got = M.state_diff(d["X"], d["X0"], errmap=code)
        np.testing.assert_allclose(got, d[key], rtol=1e-10, atol=1e-10, err_msg=f"{case} {name}")


def test_error_maps_agree_to_first_order():
    """every map is v (half the rotation angle times the axis) to first order: the tiny-rotation knot points of a fixture agree across maps to 1e-13"""
    d = H.load("errmap_rigid")
    small = slice(-6, None)
    for name in ("exponential", "mrp", "quatvec"):
        np.testing.assert_allclose(d["dX_" + name][small], d["dX_cayley"][small], rtol=0, atol=1e-13)


def test_default_map_is_cayley():
    d = H.load("errmap_p323")
    M = H.errmap_oracle_model(d)
    assert np.array_equal(M.state_diff(d["X"], d["X0"]), M.state_diff(d["X"], d["X0"], errmap=0))
