"""Width iterations and rounds of anms_ssc_kernel on the bench workloads (instrumented build -DVSF_LK_STATS)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _pkg import load_package
import bench
vsf = load_package()
STATS = "--nostats" not in sys.argv
if STATS:
    vsf.LIB_PATH = os.path.join(ROOT, "scratch", "lib", "libvsf_hip_stats.so")
dev = torch.device("cuda", 0)
for name, B in (("C2", 256), ("C4", 16)):
    cfg = bench.CONFIGS[name]; W, H = cfg["w"], cfg["h"]
    print("synth", name, flush=True)
    frames = bench.synth_frames(B, 6, W, H, dev, 0, 0.06)
    print("ctx", flush=True)
    order = list(range(6)) + list(range(4, 0, -1))
    fp = lambda i: frames.data_ptr() + order[i % len(order)] * B * W * H
    c = vsf.Context(vsf.default_params(grid_rows=cfg["rows"], grid_cols=cfg["cols"], max_features=cfg["maxn"], min_distance=cfg["mind"],
                    klt_win=21, klt_max_level=cfg["levels"] - 1, width=W, height=H, streams=B, detector_kind=vsf.DETECTOR_ANMS,
                    tracker_kind=vsf.TRACKER_LSSD if "--lssd" in sys.argv else vsf.TRACKER_LK))
    L = vsf.lib(); buf = (C.c_ulonglong * 512)()
    for i in range(40):
        c.batch_step(fp(i), W * H, W, 0)
        if i < 12 or i % 10 == 0:
            torch.cuda.synchronize(); print("step", i, flush=True)
    torch.cuda.synchronize()
    if STATS: L.vsf_debug_lk_stats(buf, 1)
    for i in range(40, 50): c.batch_step(fp(i), W * H, W, 0)
    torch.cuda.synchronize()
    if STATS: L.vsf_debug_lk_stats(buf, 0)
    s = np.array(list(buf), np.float64)
    calls = max(s[430], 1)
    print(json.dumps({"config": name, "streams": B, "searches": int(s[430]), "keypoints_per_search": s[434] / calls, "need_per_search": s[435] / calls,
                      "widths_per_search": s[431] / calls, "dense_fraction": s[433] / max(s[431], 1), "rounds_per_width": s[432] / max(s[431], 1),
                      "max_widths": s[444], "max_rounds": s[445], "search_ticks_mean": s[446] / calls, "search_ticks_max": s[447],
                      "runaway": {"rounds": s[440], "widths": s[441], "claim_probes": s[442], "owner_probes": s[443]}}), flush=True)
    c.close(); del frames
