#!/usr/bin/env python3
"""bench.py -- frames/s of one frontend step (pyramid + KLT track fwd/bwd/verify + grid detect)
on synthetic 752x480 frames resident in HBM, BASELINE.json config C2, one process per GPU.

A "step" = vsf_batch_step over B independent streams (B frames).  N GPUs => N*B streams (weak
scaling, no data-path collective); the only collective is the final RCCL all_gather of the packed
track records.  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline     -- dominant kernel (LK forward+backward+verify): algorithmic bytes / launch divided by
                  its average launch duration, measured with HIP events on the launch stream inside
                  the timed region (vsf_profile_*), against 8 TB/s HBM.
  cpu_baseline -- the CPU oracle ("restated reference CPU path", kind "port") timed on this host on
                  a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

W, H = 752, 480
LEVELS = 4
WIN = 21
MAX_FEATURES = 300
GRID = (4, 4)
MIN_DIST = 10.0
HBM_PEAK = 8.0e12
VALU_PEAK = 256 * 4 * 2.4e9 / 2  # wave64 vector instructions per second (SIMD-32: 2 cycles each)


def pyramid_bytes(w, h, levels):
    tot, sizes = 0, []
    for _ in range(levels):
        sizes.append(w * h)
        tot += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return tot, sizes


def synth_frames(B, T, device, seed):
    """[T][B][H][W] uint8: per stream a random blob texture under smooth sub-pixel translation and
    small rotation, plus +-2 grey levels of noise (same recipe as vi-slam_amd/synth.py, rendered on
    the device with torch so that thousands of distinct frames exist in HBM before timing)."""
    import torch.nn.functional as F
    g = torch.Generator(device=device)
    g.manual_seed(0xC0FFEE + seed)
    M = 48
    out = torch.empty((T, B, H, W), dtype=torch.uint8, device=device)

    def gauss_kernel(sigma):
        r = int(3 * sigma + 0.5)
        x = torch.arange(-r, r + 1, device=device, dtype=torch.float32)
        k = torch.exp(-x * x / (2 * sigma * sigma))
        return (k / k.sum()), r

    chunk = 32
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        scene = torch.zeros((nb, 1, H + 2 * M, W + 2 * M), device=device)
        for sigma, amp in ((1.6, 420.0), (3.5, 600.0), (7.0, 500.0)):
            noise = torch.randn((nb, 1, H + 2 * M, W + 2 * M), device=device, generator=g)
            k, r = gauss_kernel(sigma)
            noise = F.conv2d(F.pad(noise, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
            noise = F.conv2d(F.pad(noise, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
            scene += amp * noise / (1.0 + sigma)
        scene = 128.0 + scene
        # smooth motion per stream: velocity <= 3 px/frame, <= 0.2 deg/frame
        v = (torch.rand((nb, 2), device=device, generator=g) * 2 - 1) * 2.5
        om = (torch.rand((nb,), device=device, generator=g) * 2 - 1) * 0.15
        ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                                torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
        cx, cy = (W - 1) * 0.5, (H - 1) * 0.5
        for t in range(T):
            th = torch.deg2rad(om * t).view(nb, 1, 1)
            c, s = torch.cos(th), torch.sin(th)
            tx, ty = (v[:, 0] * t).view(nb, 1, 1), (v[:, 1] * t).view(nb, 1, 1)
            u = c * (xs - cx) - s * (ys - cy) + cx + tx + M
            w_ = s * (xs - cx) + c * (ys - cy) + cy + ty + M
            gx = u / (W + 2 * M - 1) * 2 - 1
            gy = w_ / (H + 2 * M - 1) * 2 - 1
            grid = torch.stack([gx, gy], -1)
            fr = F.grid_sample(scene, grid, mode="bilinear", padding_mode="border", align_corners=True)
            fr = fr + torch.randint(-2, 3, fr.shape, device=device, generator=g)
            out[t, b0:b0 + nb] = fr.squeeze(1).round().clamp(0, 255).to(torch.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1024, help="independent streams (frames per step) per GPU")
    ap.add_argument("--seq", type=int, default=6, help="distinct frames per stream held in HBM")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--detector", choices=("bucketing", "anms"), default="bucketing",
                    help="FeatureDetectorShiTomasiBucketing (the headline configuration) or FeatureDetectorANMS")
    ap.add_argument("--tracker", choices=("lk", "lssd"), default="lk",
                    help="FeatureTrackerLucasKanade (the headline configuration) or FeatureTrackerLSSDLucasKanade")
    ap.add_argument("--cpu-streams", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=100)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group(backend="nccl", device_id=device)

    from _pkg import load_package
    vsf = load_package()
    import importlib
    mg = importlib.import_module("vi_slam_amd.multigpu")

    B, T = args.streams, max(2, args.seq)
    frames = synth_frames(B, T, device, seed=rank)
    torch.cuda.synchronize()

    p = vsf.default_params(grid_rows=GRID[0], grid_cols=GRID[1], max_features=MAX_FEATURES,
                           min_distance=MIN_DIST, klt_win=WIN, klt_max_level=LEVELS - 1, klt_max_iter=30,
                           klt_eps=0.01, backtrack_thr=1.0, width=W, height=H, streams=B,
                           device_id=local_rank,
                           detector_kind=vsf.DETECTOR_ANMS if args.detector == "anms" else vsf.DETECTOR_BUCKETING,
                           tracker_kind=vsf.TRACKER_LSSD if args.tracker == "lssd" else vsf.TRACKER_LK)
    ctx = vsf.Context(p)
    cap = ctx.capacity
    stream = torch.cuda.current_stream().cuda_stream
    fstride = H * W
    records = torch.empty((B * cap * 24,), dtype=torch.uint8, device=device)

    # ping-pong through the T resident frames so the motion stays continuous
    order = list(range(T)) + list(range(T - 2, 0, -1))

    def frame_ptr(i):
        t = order[i % len(order)]
        return frames.data_ptr() + t * B * fstride

    step_i = 0
    for _ in range(args.warmup):
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    if args.warmup > 0:  # the gather's first call sets up RCCL channels: keep that out of the timed region
        ctx.batch_export_tracks(records.data_ptr(), stream)
        mg.gather_tracks(records, world, dist)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    ctx.batch_export_tracks(records.data_ptr(), stream)
    gathered = mg.gather_tracks(records, world, dist)  # the only collective: RCCL all_gather of track records
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ids, xy, age, n, stats = ctx.batch_download(0, min(B, 64))
    status = ctx.batch_status()
    assert (status == 0).all(), "device status reports a capacity overflow"
    mean_feats = float(n.mean())
    kept_frac = float((stats[:, 3] / np.maximum(stats[:, 0], 1)).mean())

    if rank == 0:
        total_frames = world * B * args.steps
        value = total_frames / dt
        P, _ = pyramid_bytes(W, H, LEVELS)
        P1 = P - W * H
        alg_step = 2 * W * H + P1 + 2 * P + 24 * MAX_FEATURES  # 1,806,360 B at C2
        alg_lk = 2 * P + 24 * MAX_FEATURES                     # the LK kernel's share per frame
        kernel_name = "lk_fb_kernel"
        if args.tracker == "lssd":  # two u8 frames + two sets of u16 levels 1..3 + the table
            p16 = sum(2 * (W >> l) * (H >> l) for l in range(1, 4))
            alg_lk = 2 * (W * H + p16) + 24 * MAX_FEATURES
            alg_step = 2 * W * H + W * H + p16 + alg_lk
            kernel_name = "lssd_fb_kernel"
        lk_ms, lk_n = prof["lk"]
        lk_avg = (lk_ms / lk_n) * 1e-3 if lk_n else float("nan")
        achieved = alg_lk * B / lk_avg / 1e9 if lk_n else 0.0
        traffic = None
        valu = None
        tpath = None
        if args.tracker == "lk":
            tpath = os.path.join(ROOT, "profiles", "lk_traffic.json")
        if tpath and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("streams") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
                    if tj.get("valu_insts_per_launch") and lk_n:
                        # the kernel is bound by vector-instruction issue, not bytes: committed SQ_INSTS_VALU count
                        # per launch over the live launch time, against one wave64 VALU instruction per 2 cycles
                        # per SIMD (4 SIMDs x 256 CUs x 2.4 GHz, MI355X_MICROARCH.md)
                        rate = tj["valu_insts_per_launch"] / lk_avg
                        valu = {"insts_per_launch": tj["valu_insts_per_launch"], "achieved_ginst_s": rate / 1e9,
                                "peak_ginst_s": VALU_PEAK / 1e9, "frac": rate / VALU_PEAK}
            except Exception:
                traffic = None
        line = {
            "metric": "frames/s detect+KLT-track @752x480" if args.tracker == "lk"
                      else "frames/s detect+LSSD-track @752x480",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 (f32 2x2 solve)",
            "data": "synthetic",
            "config": {"workload": "C2: synthetic 752x480 8-bit frames, %s, "
                                   "%s / 300 features, %d streams per GPU resident in HBM"
                                   % ("4-level pyramid, 21x21 LK window" if args.tracker == "lk"
                                      else "LSSD SE(2) tracker (Pattern24, 3+1 u16 levels, 5 iterations)",
                                      "4x4 grid" if args.detector == "bucketing" else "ANMS detector", B),
                       "streams_per_gpu": B, "frames_per_step": world * B,
                       "mean_features_per_stream": mean_feats, "verified_fraction": kept_frac},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic,
                         "avg_launch_ms": lk_avg * 1e3, "algorithmic_bytes_per_launch": alg_lk * B,
                         "valu_issue": valu},
            "step_hbm_frac": alg_step * value / world / HBM_PEAK,
            "kernel_ms": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()},
        }
        if not args.no_cpu and args.detector == "bucketing" and args.tracker == "lk":
            line["cpu_baseline"] = cpu_baseline(frames, order, args)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def cpu_baseline(frames, order, args):
    """Times the CPU oracle (OpenMP over streams) on the first `cpu_streams` streams of the same
    workload for `cpu_steps` steps; rank 0 at N=1 leg only, bounded to a few seconds."""
    from oracle import oracle as O
    O.build()
    S = min(args.cpu_streams, frames.shape[1])
    host = frames[:, :S].cpu().numpy()  # [T][S][H][W]
    base = O.StreamsBaseline(S, W, H, MAX_FEATURES + 64, (GRID[0], GRID[1], MAX_FEATURES, MIN_DIST),
                             (30, 0.01, WIN, LEVELS - 1, 1.0), omp=True)
    base.step(np.ascontiguousarray(host[order[0]]))  # first frame: detect only (untimed warm-up)
    t0 = time.perf_counter()
    for i in range(1, args.cpu_steps + 1):
        rc = base.step(np.ascontiguousarray(host[order[i % len(order)]]))
        assert rc == 0
    dt = time.perf_counter() - t0
    threads = base.threads
    base.close()
    return {"value": S * args.cpu_steps / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d streams x %d steps of the same C2 workload, CPU oracle (restated reference CPU "
                      "path, gcc -O3, OpenMP over streams), %.1f s" % (S, args.cpu_steps, dt)}


if __name__ == "__main__":
    main()
