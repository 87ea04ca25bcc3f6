"""Per-stage cycle counts of score_kernel's tiles (instrumented build, scratch/experiments/score_stage_timing.patch)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _pkg import load_package
import bench
vsf = load_package()
vsf.LIB_PATH = os.path.join(ROOT, "scratch", "lib", "libvsf_hip_stats.so")
B = 1024
cfg = bench.CONFIGS["C2"]; W, H = cfg["w"], cfg["h"]
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, 6, W, H, dev, 0, 0.06)
order = list(range(6)) + list(range(4, 0, -1))
fp = lambda i: frames.data_ptr() + order[i % len(order)] * B * W * H
c = vsf.Context(vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=10.0, klt_win=21, klt_max_level=3,
                width=W, height=H, streams=B))
L = vsf.lib(); buf = (C.c_ulonglong * 512)()
for i in range(300): c.batch_step(fp(i), W * H, W, 0)
L.vsf_debug_lk_stats(buf, 1)
for i in range(300, 310): c.batch_step(fp(i), W * H, W, 0)
L.vsf_debug_lk_stats(buf, 0)
s = np.array(list(buf), np.float64)
n = s[415]
names = ["stage image (global load -> LDS)", "covariance", "box + lambda", "candidates", "slot atomic", "key stores"]
print(json.dumps({"tiles": int(n), "mean_cycles_100MHz_ticks": {nm: round(s[400 + i] / n, 2) for i, nm in enumerate(names)}}, indent=1))
