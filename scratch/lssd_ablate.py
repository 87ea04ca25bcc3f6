"""LSSD kernel time against the iteration limit (set-up vs iterations)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _pkg import load_package
import bench
vsf = load_package()
B = 256
cfg = bench.CONFIGS["C2"]; W, H = cfg["w"], cfg["h"]
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, 6, W, H, dev, 0, 0.06)
def mk(it):
    return vsf.Context(vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=10.0, width=W, height=H,
                       streams=B, tracker_kind=vsf.TRACKER_LSSD, lssd_max_iterations=it))
fp = lambda t: frames.data_ptr() + t * B * W * H
out = {}
for it in (0, 1, 2, 5):
    c = mk(it)
    for t in range(3):
        c.batch_step(fp(t), W * H, W, 0)
    c.profile_enable(True)
    c.batch_step(fp(3), W * H, W, 0)
    torch.cuda.synchronize()
    c.profile_enable(False)
    pr = c.profile_read()
    ids, xy, age, n, st = c.batch_download(0, 8)
    out["iters_%d" % it] = {"lssd_ms": pr["lk"][0], "mean_n": float(n.mean()), "tracked": float(st[:, 0].mean()), "kept": float(st[:, 3].mean())}
    c.close()
print(json.dumps(out, indent=1))
