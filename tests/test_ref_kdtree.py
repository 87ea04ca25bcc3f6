"""Pins the detector's nearest-neighbour rejection rule against the REFERENCE's own kd-tree:
oracle/_ref/libref_kdtree.so is compiled by oracle/Makefile from /root/reference/src/vslam/feature/kdtree/KDTree.cpp
as it lies (STL only), behind a shim that restates the call site FeatureDetectorShiTomasiBucketing.cpp:127-130,
168-185.  The oracle's brute-force rule (orc_nn_accept; select_kernel evaluates the same expression in double, and
the whole-table GPU tests hold it to the oracle) must take the same accept / reject decision for every candidate."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_kdtree.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (no reference checkout)")


def _ref_accept(pts, q, min_dist):
    L = C.CDLL(REF)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    q = np.ascontiguousarray(q, np.float64).reshape(-1, 2)
    acc = np.zeros(len(q), np.uint8)
    nn = np.zeros((len(q), 2), np.float64)
    rc = L.ref_kdtree_accept(pts.ctypes.data_as(C.c_void_p), len(pts), q.ctypes.data_as(C.c_void_p), len(q),
                             C.c_double(min_dist), acc.ctypes.data_as(C.c_void_p), nn.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return acc.astype(bool), nn


def _orc_accept(oracle, pts, q, min_dist):
    L = oracle.lib()
    L.orc_nn_accept.restype = C.c_int
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    return np.array([bool(L.orc_nn_accept(pts.ctypes.data_as(C.c_void_p), len(pts), C.c_double(x), C.c_double(y),
                                          C.c_double(min_dist))) for x, y in np.reshape(q, (-1, 2))])


def _check(oracle, pts, q, min_dist):
    ref, nn = _ref_accept(pts, q, min_dist)
    got = _orc_accept(oracle, pts, q, min_dist)
    assert np.array_equal(ref, got), "accept/reject differs at %s" % np.nonzero(ref != got)[0][:8]
    pts = np.reshape(pts, (-1, 2))
    if len(pts):  # the tree's nearest point is AT the exact minimum squared distance
        q = np.reshape(q, (-1, 2))
        d2 = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1).min(1)
        assert np.array_equal(((q - nn) ** 2).sum(-1), d2)
    return ref


def test_empty_tree_accepts_everything(oracle):
    acc = _check(oracle, np.zeros((0, 2)), [[1.0, 2.0], [700.5, 13.25]], 10.0)
    assert acc.all()


@pytest.mark.parametrize("seed", range(6))
def test_random_sets(oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 700))
    w, h = 752, 480
    # features as the tracker leaves them: float-valued doubles, some outside the image (:127 adds them all)
    pts = np.stack([rng.uniform(-30, w + 30, n), rng.uniform(-30, h + 30, n)], 1).astype(np.float32).astype(np.float64)
    q = np.stack([rng.integers(0, w, 1500), rng.integers(0, h, 1500)], 1).astype(np.float64)  # integer corners
    md = float(rng.choice([1.0, 2.0, 7.5, 10.0, 12.0]))
    acc = _check(oracle, pts, q, md)
    assert acc.any() and (~acc).any()


def test_ties_duplicates_and_exact_threshold(oracle):
    # lattice features and lattice corners: distances hit the threshold exactly (3-4-5 triangles, axis offsets),
    # many equidistant nearest neighbours, duplicated points, a single point, collinear points
    g = np.arange(0, 60, 5, dtype=np.float64)
    pts = np.stack(np.meshgrid(g, g), -1).reshape(-1, 2)
    pts = np.concatenate([pts, pts[:40], [[-5.0, -5.0], [1000.0, 1000.0]]])
    q = np.stack(np.meshgrid(np.arange(-6, 70, dtype=np.float64), np.arange(-6, 70, dtype=np.float64)), -1).reshape(-1, 2)
    for md in (0.0, 1.0, 2.0, 2.5, np.sqrt(2.0), 3.0, 5.0):
        _check(oracle, pts, q, float(md))
    _check(oracle, [[10.0, 10.0]], [[10.0, 10.0], [13.0, 14.0], [13.0, 14.000000001], [10.0, 15.0]], 5.0)
    line = np.stack([np.arange(50, dtype=np.float64), np.zeros(50)], 1)
    _check(oracle, line, q, 3.0)
    _check(oracle, np.repeat([[7.0, 7.0]], 33, axis=0), q, 4.0)


def test_detector_candidates_on_real_tables(oracle, synth):
    """Candidates and tables as the detector sees them: the corners of a second detection pass against the
    tracked features of the first."""
    w, h = 320, 240
    frames, _ = synth.sequence(41, w, h, 2)
    det = oracle.Detector(3, 3, 150, 6.0)
    tab = oracle.FeatureTable(400)
    det.update_detection(tab, frames[0])
    pts = tab.xy[:tab.n.value].copy() + 0.37                      # as after a track step: sub-pixel positions
    det2 = oracle.Detector(3, 3, 150, 1.0)
    tab2 = oracle.FeatureTable(400)
    det2.update_detection(tab2, frames[1])
    _check(oracle, pts, tab2.xy[:tab2.n.value], 6.0)
