"""Experiment: the C2 job (1024 streams) as N contexts of 1024/N streams, each on its own HIP stream, against one
context of 1024: do the HBM-bound and latency-bound kernels of one context run under the issue-bound ones of another?
  python scratch/two_ctx.py [--parts 1 2 4] [--steps 40] [--preroll 300]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from _pkg import load_package

vsf = load_package()


def run(parts, frames, cfg, steps, preroll, B, T):
    W, H = cfg["w"], cfg["h"]
    fstride = H * W
    nb = B // parts
    ctxs, streams = [], []
    for k in range(parts):
        p = vsf.default_params(grid_rows=cfg["rows"], grid_cols=cfg["cols"], max_features=cfg["maxn"],
                               min_distance=cfg["mind"], klt_win=cfg["win"], klt_max_level=cfg["levels"] - 1,
                               klt_max_iter=30, klt_eps=0.01, backtrack_thr=1.0, width=W, height=H, streams=nb,
                               device_id=0, detector_kind=vsf.DETECTOR_BUCKETING, tracker_kind=vsf.TRACKER_LK)
        ctxs.append(vsf.Context(p))
        streams.append(torch.cuda.Stream())
    order = list(range(T)) + list(range(T - 2, 0, -1))

    def step(i):
        t = order[i % len(order)]
        for k in range(parts):
            ctxs[k].batch_step(frames.data_ptr() + (t * B + k * nb) * fstride, fstride, W, streams[k].cuda_stream)

    i = 0
    for _ in range(preroll + 3):
        step(i)
        i += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(i)
        i += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = [c.batch_download(0, 8)[3].mean() for c in ctxs]
    for c in ctxs:
        assert (c.batch_status() == 0).all()
        c.close()
    return {"parts": parts, "streams_per_context": nb, "ms_per_step": dt / steps * 1e3, "frames_per_s": B * steps / dt,
            "mean_features": float(sum(n) / len(n))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--preroll", type=int, default=300)
    a = ap.parse_args()
    cfg = bench.CONFIGS["C2"]
    B, T = cfg["streams"], 6
    dev = torch.device("cuda", 0)
    frames = bench.synth_frames(B, T, cfg["w"], cfg["h"], dev, seed=0, occlusion=0.06)
    torch.cuda.synchronize()
    for parts in a.parts:
        print(json.dumps(run(parts, frames, cfg, a.steps, a.preroll, B, T)), flush=True)


if __name__ == "__main__":
    main()
