"""Pins the oracle's (and the product host-side) SE(3)/quaternion primitives against
scipy.spatial.transform.Rotation -- the reference's own test oracle (tests/test_spatialmath.py:4,108,
158,209,423,488,515,552) -- through the committed golden vectors, plus the algebraic identities the
reference tests (reversed quaternion product, inverse)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

import oracle.spatialmath as osm
import optas_amd.spatialmath as psm


def _quat_close(a, b, tol=1e-12):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


@pytest.mark.parametrize("sm", [osm, psm], ids=["oracle", "product-host"])
class TestAgainstScipyGolden:
    def test_rot_xyz(self, sm, golden_sm):
        for i, th in enumerate(golden_sm["theta"]):
            assert np.allclose(sm.rotx(th), golden_sm["rotx"][i], atol=1e-14, rtol=0)
            assert np.allclose(sm.roty(th), golden_sm["roty"][i], atol=1e-14, rtol=0)
            assert np.allclose(sm.rotz(th), golden_sm["rotz"][i], atol=1e-14, rtol=0)

    def test_angvec2r(self, sm, golden_sm):
        for i in range(len(golden_sm["theta"])):
            R = sm.angvec2r(golden_sm["theta"][i], golden_sm["axis"][i])
            assert np.allclose(R, golden_sm["angvec2r"][i], atol=1e-14, rtol=0)

    def test_rpy2r(self, sm, golden_sm):
        for i in range(len(golden_sm["rpy"])):
            assert np.allclose(sm.rpy2r(golden_sm["rpy"][i]), golden_sm["rpy2r"][i], atol=1e-14, rtol=0)
        with pytest.raises(ValueError):
            sm.rpy2r([0.1, 0.2, 0.3], opt="bad")

    def test_quaternion_fromrpy_fromangvec(self, sm, golden_sm):
        for i in range(len(golden_sm["rpy"])):
            assert _quat_close(sm.Quaternion.fromrpy(golden_sm["rpy"][i]).getquat(), golden_sm["quat_fromrpy"][i])
            q = sm.Quaternion.fromangvec(golden_sm["theta"][i], golden_sm["axis"][i]).getquat()
            assert _quat_close(q, golden_sm["quat_fromangvec"][i])

    def test_quaternion_product_is_reversed(self, sm, golden_sm):
        # reference pin: (a*b) rotates like R(b) R(a)  (tests/test_spatialmath.py:415-423)
        qs = golden_sm["quat_fromrpy"]
        for i in range(0, len(qs) - 1, 2):
            a, b = sm.Quaternion.fromvec(qs[i]), sm.Quaternion.fromvec(qs[i + 1])
            ab = (a * b).getquat()
            ref = (Rot.from_quat(qs[i + 1]) * Rot.from_quat(qs[i])).as_quat()
            assert _quat_close(ab, ref)

    def test_quaternion_inverse_and_getrpy(self, sm, golden_sm):
        for i in range(len(golden_sm["rpy"])):
            q = sm.Quaternion.fromvec(golden_sm["quat_fromrpy"][i])
            e = (q * q.inv()).getquat()
            assert np.allclose(e, [0, 0, 0, 1], atol=1e-14)
            assert np.allclose(q.getrpy(), golden_sm["rpy"][i], atol=1e-12)

    def test_transform_helpers(self, sm):
        R = sm.rpy2r([0.3, -0.2, 0.5])
        T = sm.rt2tr(R, [1.0, 2.0, 3.0])
        assert np.allclose(sm.invt(T) @ T, np.eye(4), atol=1e-15)
        assert np.allclose(sm.r2t(R)[:3, :3], R) and np.allclose(sm.r2t(R)[:3, 3], 0)
        assert np.allclose(sm.skew([1.0, 2.0, 3.0]) @ np.array([0.5, -1.0, 2.0]), np.cross([1.0, 2.0, 3.0], [0.5, -1.0, 2.0]))
        assert np.isclose(np.linalg.norm(sm.unit([3.0, 4.0, 0.0])), 1.0)
        with pytest.raises(ValueError):
            sm.skew([1.0, 2.0])
