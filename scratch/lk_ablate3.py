"""LK kernel time against the iteration limit on a fixed table (uncached: the tables come from the host), to split the
kernel's time into level set-up, first iteration (+ J staging) and marginal iterations."""
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
    return vsf.Context(vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=10.0, klt_win=21, klt_max_level=3,
                       klt_max_iter=it, klt_eps=0.01, backtrack_thr=1.0, width=W, height=H, streams=B))
base = mk(30)
fp = lambda t: frames.data_ptr() + t * B * W * H
for t in range(4):
    base.batch_step(fp(t), W * H, W, 0)
ids, xy, age, n, _ = base.batch_download()
out = {"mean_features": float(n.mean())}
for it in (0, 1, 2, 3, 30):
    c = mk(it)
    c.batch_upload(ids, xy, age, n)
    c.batch_track(fp(3), W * H, W, 0)
    ms = []
    for rep in range(4):
        c.batch_upload(ids, xy, age, n)
        c.reset_tracker(); c.batch_track(fp(3), W * H, W, 0)
        c.batch_upload(ids, xy, age, n)
        c.profile_enable(True)
        c.batch_track(fp(4), W * H, W, 0)
        torch.cuda.synchronize()
        c.profile_enable(False)
        ms.append(c.profile_read()["lk"][0])
    out["max_iter_%d" % it] = min(ms)
    c.close()
print(json.dumps(out))
