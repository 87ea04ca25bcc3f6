"""select_kernel time against the LDS key capacity (vsf_debug_set_cand_cap): is it occupancy-bound?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _pkg import load_package
import bench
vsf = load_package()
B = 1024
cfg = bench.CONFIGS["C2"]; W, H = cfg["w"], cfg["h"]
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, 6, W, H, dev, 0, 0.06)
order = list(range(6)) + list(range(4, 0, -1))
fp = lambda i: frames.data_ptr() + order[i % len(order)] * B * W * H
for cap in (0, 2048, 1024, 512):
    c = vsf.Context(vsf.default_params(grid_rows=4, grid_cols=4, max_features=300, min_distance=10.0, klt_win=21, klt_max_level=3,
                    width=W, height=H, streams=B))
    if cap: c.debug_set_cand_cap(cap)
    for i in range(300): c.batch_step(fp(i), W * H, W, 0)
    c.profile_enable(True)
    for i in range(300, 320): c.batch_step(fp(i), W * H, W, 0)
    torch.cuda.synchronize(); c.profile_enable(False)
    pr = c.profile_read()
    print(cap, {k: round(v[0], 4) for k, v in pr.items() if k in ("select", "candidates", "score")}, flush=True)
    c.close()
