"""Cliff check at the other BASELINE configs (parity for them is in tests/test_gpu_batch.py::test_full_size_configs):
C3 1280x720 / 4 levels / 1000 features, C4 3840x2160 / 5 levels / 4000 features.  Not a bench line."""
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
from _pkg import load_package
vsf = load_package()
dev = torch.device("cuda", 0)
for name, (w, h, levels, maxn, grid, B) in {"C3": (1280, 720, 4, 1000, (4, 4), 256), "C4": (3840, 2160, 5, 4000, (8, 8), 32)}.items():
    bench.W, bench.H = w, h
    frames = bench.synth_frames(B, 4, dev, 0)
    torch.cuda.synchronize()
    for tracker in ("lk", "lssd", "lk+anms"):
        if tracker == "lk+anms" and 2 * maxn > 2048:
            continue  # anms_detection_features <= 2048
        p = vsf.default_params(grid_rows=grid[0], grid_cols=grid[1], max_features=maxn, min_distance=10.0, klt_win=21,
                               klt_max_level=levels - 1, width=w, height=h, streams=B,
                               tracker_kind=vsf.TRACKER_LSSD if tracker == "lssd" else vsf.TRACKER_LK,
                               detector_kind=vsf.DETECTOR_ANMS if tracker.endswith("anms") else vsf.DETECTOR_BUCKETING,
                               anms_detection_features=2 * maxn if tracker.endswith("anms") else 600)
        ctx = vsf.Context(p)
        order = [0, 1, 2, 3, 2, 1]
        fs = w * h
        for i in range(3):
            ctx.batch_step(frames.data_ptr() + order[i % 6] * B * fs, fs, w, 0)
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        K = 10
        for i in range(3, 3 + K):
            ctx.batch_step(frames.data_ptr() + order[i % 6] * B * fs, fs, w, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ctx.profile_enable(False)
        pr = ctx.profile_read()
        _, _, _, n, st = ctx.batch_download(0, min(B, 32))
        assert (ctx.batch_status() == 0).all()
        print(json.dumps({"config": name, "tracker": tracker, "streams": B, "frames_per_s": B * K / dt, "ms_per_step": dt / K * 1e3,
                          "mean_features": float(n.mean()), "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items()}}))
        ctx.close()
    del frames
    torch.cuda.empty_cache()
