#!/usr/bin/env python3
"""C1 from disk: runs the GPU frontend path over an EuRoC sequence (mav0/cam0 + basalt calibration),
the way example_feature_detection_and_tracking.cpp drives the reference, and prints one JSON line with
the disk-to-tracks rate (PNG decode + host->device copy + step; never bench.py's `value`).
  usage: python tools/run_euroc.py --dataset /path/to/V1_01_easy --calib calib.json [--mask m.png] [--tracker lssd]
         python tools/run_euroc.py --synthetic 200      (writes a synthetic 752x480 sequence to a temp dir first)"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset")
    ap.add_argument("--calib")
    ap.add_argument("--mask", default="")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--tracker", choices=("lk", "lssd"), default="lk")
    args = ap.parse_args()
    from _pkg import load_package, load_synth
    vsf = load_package()
    euroc = importlib.import_module("vi_slam_amd.euroc")
    tmp = None
    if args.synthetic:
        from test_euroc_ingest import make_dataset
        tmp = tempfile.TemporaryDirectory()
        frames, _ = load_synth().sequence(1, 752, 480, args.synthetic)
        args.dataset = os.path.join(tmp.name, "SYNTH")
        args.calib = make_dataset(args.dataset, frames)
    dp = euroc.EurocKimeraDataProvider(args.dataset, args.calib, args.mask)
    w, h = dp.frame_size()
    det = vsf.FeatureDetectorShiTomasiBucketing(4, 4, 300, 2.0, w, h, capacity=1024)
    if args.tracker == "lssd":
        trk = vsf.FeatureTrackerLSSDLucasKanade(vsf.FeatureTrackerLSSDLucasKanade.Parameter(), w, h, capacity=1024)
    else:
        trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 21, 3, 1.0, w, h, capacity=1024)
    db = vsf.FeatureDatabase(1024)
    n, t_decode, t_gpu, feats = 0, 0.0, 0.0, 0
    t0 = time.perf_counter()
    while True:
        a = time.perf_counter()
        inp = dp.GetInput()
        b = time.perf_counter()
        if inp is None:
            break
        trk.Track(db, inp.frame_)
        det.UpdateDetection(db, inp.frame_, inp.mask_)
        c = time.perf_counter()
        t_decode += b - a
        t_gpu += c - b
        feats += db.n
        n += 1
    dt = time.perf_counter() - t0
    # the same loop through vsf_step on ONE context: one frame upload, one table round trip, one synchronisation
    step_ms = None
    if args.tracker == "lk":
        dp2 = euroc.EurocKimeraDataProvider(args.dataset, args.calib, args.mask)
        ctx = vsf.Context(vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=2.0, klt_max_iter=30,
                                             klt_eps=0.01, klt_win=21, klt_max_level=3, backtrack_thr=1.0, width=w,
                                             height=h, capacity=1024))
        db2 = vsf.FeatureDatabase(1024)
        t_step, m = 0.0, 0
        while True:
            inp = dp2.GetInput()
            if inp is None:
                break
            a = time.perf_counter()
            db2.n, _ = ctx.step(inp.frame_, db2.ids, db2.xy, db2.age, db2.n, inp.mask_)
            t_step += time.perf_counter() - a
            m += 1
        step_ms = 1e3 * t_step / max(m, 1)
        assert db2.as_dict() == db.as_dict(), "vsf_step differs from Track + UpdateDetection"
        ctx.close()
    print(json.dumps({"workload": "C1: EuRoC cam0 from disk, %dx%d, %d frames, single stream, host API" % (w, h, n),
                      "frames_per_s": n / dt, "decode_ms_per_frame": 1e3 * t_decode / max(n, 1),
                      "track_detect_ms_per_frame": 1e3 * t_gpu / max(n, 1), "vsf_step_ms_per_frame": step_ms,
                      "mean_features": feats / max(n, 1),
                      "tracker": args.tracker}))


if __name__ == "__main__":
    main()
