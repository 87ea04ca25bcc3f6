#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/*.npz from the CPU oracle.

PARITY UNPINNED: the reference (eryeden/vi-slam) holds no golden vectors for this path (its tests
are placeholders, test/sample/test_sample1.cpp:38-45) and its arithmetic lives in OpenCV 3.4.9,
which is installed neither here nor on the GPU box.  These fixtures therefore pin the build's own
oracle (oracle/vsf_oracle.c, a restatement of SURVEY.md Appendix A) on seeded synthetic inputs, so
that later edits to the oracle or to the kernels cannot drift silently.  A fixture is data only:
input frames, parameters, and the expected outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from _pkg import load_synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (w, h, seed, n_frames, detector (rows, cols, max, min_dist), tracker (iter, eps, win, lvl, back))
    "small_a": (160, 120, 101, 4, (2, 2, 60, 6.0), (30, 0.01, 21, 3, 1.0)),
    "small_b": (200, 136, 202, 4, (3, 4, 96, 2.0), (30, 0.01, 15, 3, 1.0)),
    "small_c": (176, 144, 303, 3, (2, 3, 60, 0.5), (20, 0.1, 9, 2, 0.5)),
}
CAP = 256


def run_case(w, h, seed, n_frames, det_args, trk_args):
    synth = load_synth()
    frames, _ = synth.sequence(seed, w, h, n_frames)
    out = {"frames": frames, "det_args": np.array(det_args, np.float64),
           "trk_args": np.array(trk_args, np.float64)}
    # per-stage building blocks on frame 0 / the pair (0, 1)
    win, ml = int(trk_args[2]), int(trk_args[3])
    pyr = O.build_pyramid(frames[0], win, ml)
    for l, lv in enumerate(pyr):
        out["pyr%d" % l] = lv
    out["scharr0"] = O.scharr(pyr[-1])
    rects, origin = O.generate_grid(w, h, det_args[1], det_args[0])
    out["grid_rects"], out["grid_origin"] = rects, origin
    out["mineig_cell0"] = O.corner_min_eigen_val(frames[0], rects[0])
    out["gftt_cell0"] = O.good_features_to_track(frames[0], 25, 0.1, det_args[3], roi=rects[0])
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-3, w + 3, 80), rng.uniform(-3, h + 3, 80)], 1).astype(np.float32)
    out["lk_pts"] = pts
    for mode, tag in ((O.ACC_FLOAT_SEQ, "seq"), (O.ACC_EXACT, "exact")):
        nxt, st, err = O.calc_optical_flow_pyr_lk(frames[0], frames[1], pts, win, ml, int(trk_args[0]),
                                                  trk_args[1], acc_mode=mode)
        out["lk_next_" + tag], out["lk_status_" + tag] = nxt, st
        if mode == O.ACC_FLOAT_SEQ:
            out["lk_err_seq"] = err
    # the frontend sequence: Track then UpdateDetection per frame, both accumulation modes
    for mode, tag in ((O.ACC_FLOAT_SEQ, "seq"), (O.ACC_EXACT, "exact")):
        det = O.Detector(*det_args)
        trk = O.Tracker(*trk_args, mode)
        tab = O.FeatureTable(CAP)
        for k in range(n_frames):
            rc, st = trk.track(tab, frames[k])
            assert rc == 0
            n = tab.n.value
            out["%s_trk%d_ids" % (tag, k)] = tab.ids[:n].copy()
            out["%s_trk%d_xy" % (tag, k)] = tab.xy[:n].copy()
            out["%s_trk%d_stats" % (tag, k)] = np.array(st, np.int32)
            assert det.update_detection(tab, frames[k]) == 0
            n = tab.n.value
            out["%s_det%d_ids" % (tag, k)] = tab.ids[:n].copy()
            out["%s_det%d_xy" % (tag, k)] = tab.xy[:n].copy()
            out["%s_det%d_age" % (tag, k)] = tab.age[:n].copy()
            out["%s_det%d_maxidx" % (tag, k)] = np.array([det.max_feature_index], np.uint64)
    # the same sequence with the ANMS detector (FeatureDetectorANMS) in place of the bucketing one
    amax, adist = int(det_args[2]), max(float(det_args[3]), 4.0)
    out["anms_args"] = np.array([amax, adist], np.float64)
    det = O.AnmsDetector(amax, adist)
    trk = O.Tracker(*trk_args, O.ACC_EXACT)
    tab = O.FeatureTable(CAP * 2)
    for k in range(n_frames):
        rc, st = trk.track(tab, frames[k])
        assert rc == 0
        n = tab.n.value
        out["anms_mask%d" % k] = np.packbits(det.build_mask(w, h, tab.xy[:n]))
        assert det.update_detection(tab, frames[k]) in (0, 1)
        n = tab.n.value
        out["anms_det%d_ids" % k] = tab.ids[:n].copy()
        out["anms_det%d_xy" % k] = tab.xy[:n].copy()
        out["anms_det%d_age" % k] = tab.age[:n].copy()
        out["anms_det%d_maxidx" % k] = np.array([det.max_feature_index], np.uint64)
    out["anms_kp0"] = O.good_features_to_track(frames[0], 2 * amax, 0.001, 1.0,
                                               mask=det.build_mask(w, h, np.zeros((0, 2))), block_size=4)
    # the LSSD SE(2) patch tracker (FeatureTrackerLSSDLucasKanade, default Parameter): pyramid, point
    # tracking in both summation orders, and the frontend sequence with the bucketing detector
    lp = O.lssd_build_pyramid(frames[0], 3)
    for l in range(1, 4):
        out["lssd_pyr%d" % l] = lp[l]
    kp = O.good_features_to_track(frames[0], 60, 0.01, 5.0)
    lpts = np.concatenate([kp + 0.25, pts[:20]]).astype(np.float32)
    out["lssd_pts"] = lpts
    for mode, tag in ((O.LSSD_SEQ, "seq"), (O.LSSD_TREE, "tree")):
        nxt, ok = O.lssd_track_points(frames[0], frames[1], lpts, 3, 5, mode)
        out["lssd_next_" + tag], out["lssd_valid_" + tag] = nxt, ok
    det = O.Detector(*det_args)
    trk = O.LssdTracker(mode=O.LSSD_TREE)
    tab = O.FeatureTable(CAP)
    for k in range(n_frames):
        assert trk.track(tab, frames[k]) == 0
        n = tab.n.value
        out["lssd_trk%d_ids" % k] = tab.ids[:n].copy()
        out["lssd_trk%d_xy" % k] = tab.xy[:n].copy()
        assert det.update_detection(tab, frames[k]) == 0
        n = tab.n.value
        out["lssd_det%d_ids" % k] = tab.ids[:n].copy()
        out["lssd_det%d_xy" % k] = tab.xy[:n].copy()
        out["lssd_det%d_age" % k] = tab.age[:n].copy()
    return out


def main():
    O.build()
    for name, args in CASES.items():
        data = run_case(*args)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(name, "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
