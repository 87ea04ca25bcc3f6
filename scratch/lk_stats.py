"""Per-level iteration histogram and event counts of lk_fb_kernel on the bench workload (instrumented build,
-DVSF_LK_STATS -> scratch/lib/libvsf_hip_stats.so).  usage: python scratch/lk_stats.py [streams] [occlusion]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from _pkg import load_package
import bench

vsf = load_package()
vsf.LIB_PATH = os.path.join(ROOT, "scratch", "lib", "libvsf_hip_stats.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
occ = float(sys.argv[2]) if len(sys.argv) > 2 else 0.06
cfg = bench.CONFIGS["C2"]
W, H = cfg["w"], cfg["h"]
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, 6, W, H, dev, 0, occ)
p = vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=10.0, klt_win=21, klt_max_level=3,
                       klt_max_iter=30, klt_eps=0.01, backtrack_thr=1.0, width=W, height=H, streams=B)
ctx = vsf.Context(p)
order = list(range(6)) + list(range(4, 0, -1))
L = vsf.lib()
buf = (C.c_ulonglong * 512)()
for i in range(4):
    ctx.batch_step(frames.data_ptr() + order[i % len(order)] * B * W * H, W * H, W, 0)
L.vsf_debug_lk_stats(buf, 1)
steps = 10
for i in range(4, 4 + steps):
    ctx.batch_step(frames.data_ptr() + order[i % len(order)] * B * W * H, W * H, W, 0)
L.vsf_debug_lk_stats(buf, 0)
s = np.array(list(buf), np.int64)
out = {"streams": B, "occlusion": occ, "steps": steps, "levels": []}
names = ["entered", "passed_min_eig", "iterations", "j_stagings", "interior_setups", "end_eps", "end_oscillation",
         "end_max_count", "end_left_image", "cached_setups"]
for lv in range(4):
    d = {n: int(s[lv * 16 + k]) for k, n in enumerate(names)}
    d["mean_iterations"] = d["iterations"] / max(1, d["passed_min_eig"])
    d["histogram"] = [int(x) for x in s[128 + lv * 32:128 + lv * 32 + 32]]
    out["levels"].append(d)
tracks = out["levels"][0]["entered"]
out["point_tracks"] = tracks
out["iterations_per_track"] = sum(l["iterations"] for l in out["levels"]) / max(1, tracks)
print(json.dumps(out))
