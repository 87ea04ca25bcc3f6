#!/usr/bin/env python3
"""Generates tests/golden/lk_seq_outliers.json: for every parametrised case of
tests/test_gpu_parity.py::test_lk_points, the indices of the random points whose track under exact integer sums
(ORC_ACC_EXACT, what the HIP kernel computes bit for bit) differs by more than 0.01 px from the track under the
reference's accumulation order (OpenCV scalar float32 raster order, ORC_ACC_FLOAT_SEQ), with both positions.
The GPU test asserts that the set of points beyond the tolerance is EXACTLY this set, so it cannot grow.
CPU only (the oracle in both modes); run from the repo root:  python tests/golden/make_lk_outliers.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [(752, 480, 21, 3, 0.01), (640, 360, 15, 3, 0.01), (480, 320, 24, 4, 0.1), (400, 300, 31, 2, 0.03),
         (320, 240, 9, 2, 0.01)]
TOL_PX = 0.01


def case_points(w, h):
    rng = np.random.default_rng(5)
    n = 600
    pts = np.stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)], 1).astype(np.float32)
    pts[:50] = np.round(pts[:50])  # integer positions: zero fractional weights
    return pts


def main():
    O.build()
    synth = load_synth()
    out = {}
    for (w, h, win, ml, eps) in CASES:
        frames, _ = synth.sequence(11, w, h, 2)
        pts = case_points(w, h)
        a, sa, _ = O.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, eps, acc_mode=O.ACC_EXACT)
        b, sb, _ = O.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, 30, eps, acc_mode=O.ACC_FLOAT_SEQ)
        assert np.array_equal(sa, sb), "status differs between the two summation orders"
        d = np.abs(a - b).max(1)
        idx = [int(i) for i in np.nonzero((sa == 1) & (d > TOL_PX))[0]]
        out["%dx%d_win%d_ml%d_eps%g" % (w, h, win, ml, eps)] = {
            "tracked": int(sa.sum()), "outliers": idx,
            "exact_xy": [[float(v) for v in a[i]] for i in idx], "float_seq_xy": [[float(v) for v in b[i]] for i in idx],
            "max_dev_px_of_the_rest": float(d[(sa == 1) & (d <= TOL_PX)].max())}
        print(w, h, win, ml, eps, "tracked", int(sa.sum()), "outliers", idx)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lk_seq_outliers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
