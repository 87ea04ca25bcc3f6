#!/usr/bin/env python3
"""bench.py -- frames/s of one frontend step (pyramid + KLT track fwd/bwd/verify + grid detect)
on synthetic frames resident in HBM, one process per GPU.  Default workload: BASELINE.json config C2.

A "step" = vsf_batch_step over B independent streams (B frames).  N GPUs => N*B streams (weak
scaling, no data-path collective); the only collective is the final RCCL all_gather of the packed
track records.  `--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, before
anything touches the GPU; under torchrun it checks that WORLD_SIZE == N.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- dominant kernel (LK forward+backward+verify): algorithmic bytes / launch divided by
                  its average launch duration, measured with HIP events on the launch stream inside
                  the timed region (vsf_profile_*), against 8 TB/s HBM (`frac`, the contract's figure);
                  `bound` names the resource the counters show the kernel is limited by ("valu": vector
                  instruction issue), `valu_issue` states that in cycles of the shader
                  clock measured in this run under the step's load (vsf_debug_sclk_probe); `traffic` and
                  the instruction count come from profiles/lk_traffic.json and are emitted only when the
                  hash of vi-slam_amd/csrc/* stamped there matches the sources the library was built
                  from (null otherwise); `stream_copy_GBps` is a device copy measured in the same job.
  cpu_baseline -- the CPU oracle ("restated reference CPU path", kind "port") timed on this host on
                  a bounded sample of the same workload: all cores (OpenMP over streams) and one thread.
  configs      -- short runs (few steps, no CPU leg) of the other BASELINE.json configurations in the
                  same process: value, ms_per_step, roofline.frac and kernel_ms of C3, C4, C5.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hashlib  # noqa: E402

HBM_PEAK = 8.0e12
N_SIMD = 256 * 4


def csrc_sha16():
    """Hash of the kernel sources (the same recipe as tools/update_traffic.py): ties a committed counter file to
    the code it was measured on."""
    d = os.path.join(ROOT, "vi-slam_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]

# sizes, grids (cols x rows), feature counts and batch sizes of BASELINE.json's configs.
# cpu = (streams, steps) of the all-cores leg and of the one-thread leg, sized for ~5-15 s each.
CONFIGS = {
    "C2": dict(w=752, h=480, levels=4, win=21, rows=4, cols=4, maxn=300, mind=10.0, streams=1024,
               cpu=((64, 100), (8, 40)), name="C2: synthetic 752x480 8-bit frames"),
    "C3": dict(w=1280, h=720, levels=4, win=21, rows=5, cols=8, maxn=1000, mind=8.0, streams=256,
               cpu=((32, 40), (4, 16)), name="C3: synthetic 1280x720 8-bit frames"),
    "C4": dict(w=3840, h=2160, levels=5, win=21, rows=10, cols=16, maxn=4000, mind=12.0, streams=32,
               cpu=((16, 8), (2, 4)), name="C4: synthetic 3840x2160 8-bit frames"),
    "C5": dict(w=752, h=480, levels=4, win=21, rows=4, cols=4, maxn=300, mind=10.0, streams=8,
               cpu=((8, 200), (2, 60)), name="C5: 64 independent 752x480 sequences over 8 GPUs = 8 streams per GPU"),
}


def pyramid_bytes(w, h, levels):
    tot = 0
    for _ in range(levels):
        tot += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return tot


def synth_frames(B, T, W, H, device, seed, occlusion):
    """[T][B][H][W] uint8: per stream a random blob texture under smooth sub-pixel translation and
    small rotation, plus +-2 grey levels of noise (same recipe as vi-slam_amd/synth.py, rendered on
    the device with torch so that thousands of distinct frames exist in HBM before timing).
    `occlusion` > 0 pastes independently placed 32x32 blocks of a foreign texture over that fraction of
    every frame (moving occluders): the features under them fail the forward-backward check, so every
    step loses a few per cent of its tracks and the detector has real work in (nearly) every cell."""
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

    def texture(nb, hh, ww):
        scene = torch.zeros((nb, 1, hh, ww), device=device)
        for sigma, amp in ((1.6, 420.0), (3.5, 600.0), (7.0, 500.0)):
            noise = torch.randn((nb, 1, hh, ww), device=device, generator=g)
            k, r = gauss_kernel(sigma)
            noise = F.conv2d(F.pad(noise, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
            noise = F.conv2d(F.pad(noise, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
            scene += amp * noise / (1.0 + sigma)
        return 128.0 + scene

    chunk = max(1, min(32, (64 << 20) // (W * H)))
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    cx, cy = (W - 1) * 0.5, (H - 1) * 0.5
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        scene = texture(nb, H + 2 * M, W + 2 * M)
        alt = texture(nb, H, W) if occlusion > 0 else None
        # smooth motion per stream: velocity <= 3 px/frame, <= 0.2 deg/frame
        v = (torch.rand((nb, 2), device=device, generator=g) * 2 - 1) * 2.5
        om = (torch.rand((nb,), device=device, generator=g) * 2 - 1) * 0.15
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
            if occlusion > 0:
                gh, gw = (H + 31) // 32, (W + 31) // 32
                occ = (torch.rand((nb, 1, gh, gw), device=device, generator=g) < occlusion).float()
                occ = F.interpolate(occ, scale_factor=32, mode="nearest")[:, :, :H, :W] > 0.5
                sh = torch.randint(0, 1 << 20, (2,), generator=g, device=device).tolist()
                fr = torch.where(occ, torch.roll(alt, (sh[0] % H, sh[1] % W), (2, 3)), fr)
            fr = fr + torch.randint(-2, 3, fr.shape, device=device, generator=g)
            out[t, b0:b0 + nb] = fr.squeeze(1).round().clamp(0, 255).to(torch.uint8)
    return out


def spawn_ranks(n):
    """`--gpus N` without a launcher: N fresh processes, one per GPU, started before this process has
    touched the GPU (device_count() does not initialise it); rank 0 prints the line."""
    ndev = torch.cuda.device_count()
    if ndev < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible\n" % (n, ndev))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:  # a rank died: the others would wait for it at the next barrier
                    q.terminate()
    sys.exit(rc)


def stream_copy_gbps(device):
    """Device-to-device copy bandwidth (read + write bytes) measured in this job, beside the vendor peak."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.zero_()
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def algorithmic_bytes(cfg, tracker):
    """per-frame bytes of the whole step and of the tracker kernel."""
    W, H, LEVELS, MAXN = cfg["w"], cfg["h"], cfg["levels"], cfg["maxn"]
    P = pyramid_bytes(W, H, LEVELS)
    P1 = P - W * H
    alg_step = 2 * W * H + P1 + 2 * P + 24 * MAXN  # 1,806,360 B at C2
    alg_lk = 2 * P + 24 * MAXN
    if tracker == "lssd":  # two u8 frames + two sets of u16 levels 1..3 + the table
        p16 = sum(2 * (W >> l) * (H >> l) for l in range(1, 4))
        alg_lk = 2 * (W * H + p16) + 24 * MAXN
        alg_step = 2 * W * H + W * H + p16 + alg_lk
    return alg_step, alg_lk


def run_workload(vsf, mg, name, args, env, steps, warmup, preroll, streams=0, keep_frames=False, probe=False):
    """One configuration: synthetic frames into HBM, pre-roll, warm-up, EXACTLY `steps` timed steps between
    barrier + synchronize pairs, max over ranks.  Returns the measured figures (and the frames when asked)."""
    device, local_rank, rank, world, dist = env
    cfg = CONFIGS[name]
    W, H, LEVELS, WIN, MAXN = cfg["w"], cfg["h"], cfg["levels"], cfg["win"], cfg["maxn"]
    B, T = (streams or cfg["streams"]), max(2, args.seq)
    frames = synth_frames(B, T, W, H, device, seed=rank, occlusion=args.occlusion)
    torch.cuda.synchronize()
    p = vsf.default_params(grid_rows=cfg["rows"], grid_cols=cfg["cols"], max_features=MAXN,
                           min_distance=cfg["mind"], klt_win=WIN, klt_max_level=LEVELS - 1, klt_max_iter=30,
                           klt_eps=0.01, backtrack_thr=1.0, width=W, height=H, streams=B,
                           device_id=local_rank,
                           detector_kind=vsf.DETECTOR_ANMS if args.detector == "anms" else vsf.DETECTOR_BUCKETING,
                           tracker_kind=vsf.TRACKER_LSSD if args.tracker == "lssd" else vsf.TRACKER_LK,
                           anms_detection_min_distance=args.anms_min_distance,
                           anms_use_harris=1 if args.anms_harris else 0)
    ctx = vsf.Context(p)
    cap = ctx.capacity
    stream = torch.cuda.current_stream().cuda_stream
    fstride = H * W
    # the only collective: all-gather of the 24-byte track records, through the library's own RCCL entry
    # (vsf_gather_tracks) when it can be set up on every rank, through torch.distributed otherwise
    native = None
    if not args.torch_gather:
        native = mg.NativeGather(vsf, ctx, world, rank, local_rank, dist, device)
        if not native.ok:
            if rank == 0:
                sys.stderr.write("bench.py: native gather unavailable (%s); using torch.distributed\n" % native.error)
            native = None
    records = None if native is not None else torch.empty((B * cap * 24,), dtype=torch.uint8, device=device)

    def gather():
        if native is not None:
            return native.gather(stream)
        ctx.batch_export_tracks(records.data_ptr(), stream)
        return mg.gather_tracks(records, world, dist)

    # ping-pong through the T resident frames so the motion stays continuous
    order = list(range(T)) + list(range(T - 2, 0, -1))

    def frame_ptr(i):
        t = order[i % len(order)]
        return frames.data_ptr() + t * B * fstride

    step_i = 0
    for _ in range(max(0, preroll)):  # workload set-up: out of the start-up transient (see --preroll)
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    for _ in range(warmup):
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    if warmup > 0:  # the gather's first call sets up RCCL channels: keep that out of the timed region
        gather()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # In the timed region only the tracker kernel -- the roofline's kernel -- sits between HIP events: an event pair
    # drains the stream before and after its kernel (about 8 us of idle GPU each), and timing all eight kernel
    # classes there cost the C2 step 3-5 %.  The other classes are timed over the steps that FOLLOW the timed region.
    ctx.profile_enable(True, kernels=[vsf.KERNEL_NAMES.index("lk")])
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    gather()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof_lk = ctx.profile_read()
    nb = max(3, min(steps, 10))
    ctx.profile_enable(True)
    for _ in range(nb):  # untimed: the per-class breakdown
        ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
        step_i += 1
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    prof["lk_all_events"] = prof["lk"]   # the tracker kernel with every class timed around it (cross-check)
    prof["lk"] = prof_lk["lk"]
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ids, xy, age, n, stats = ctx.batch_download(0, min(B, 64))
    status = ctx.batch_status()
    assert (status == 0).all(), "device status reports a capacity overflow"
    res = {"name": name, "cfg": cfg, "B": B, "dt": dt, "steps": steps, "prof": prof,
           "mean_feats": float(n.mean()), "kept_frac": float((stats[:, 3] / np.maximum(stats[:, 0], 1)).mean()),
           "gather": "vsf_gather_tracks (RCCL opened by libvsf_hip.so)" if native is not None
                     else "torch.distributed all_gather", "lk_cache_bytes": int(ctx.info().lk_cache_bytes),
           "order": order}
    if probe:
        # shader clock under the step's load: queue more (untimed) steps, then spin a one-lane probe on a stream of
        # its own for about half of what is queued; afterwards the same probe on the idle GPU
        ms = dt / steps * 1e3
        extra = int(min(400, max(10, 120.0 / max(ms, 0.05))))
        for _ in range(extra):
            ctx.batch_step(frame_ptr(step_i), fstride, W, stream)
            step_i += 1
        try:
            res["sclk_mhz_load"], res["realtime_mhz"] = ctx.sclk_probe(int(max(2000, min(60000, 0.5 * extra * ms * 1e3))))
            torch.cuda.synchronize()
            res["sclk_mhz_idle"], _ = ctx.sclk_probe(5000)
        except Exception as e:  # noqa: BLE001 -- diagnostic only
            res["sclk_error"] = str(e)
        torch.cuda.synchronize()
    if native is not None:
        native.close()
    ctx.close()
    if keep_frames:
        res["frames"] = frames
    else:
        del frames
        torch.cuda.empty_cache()
    return res


def summarize(res, world, tracker):
    """The figures of one configuration: value, ms per step, the tracker kernel's contract roofline, kernel times."""
    cfg, B = res["cfg"], res["B"]
    alg_step, alg_lk = algorithmic_bytes(cfg, tracker)
    value = world * B * res["steps"] / res["dt"]
    lk_ms, lk_n = res["prof"]["lk"]
    lk_avg = (lk_ms / lk_n) * 1e-3 if lk_n else float("nan")
    achieved = alg_lk * B / lk_avg / 1e9 if lk_n else 0.0
    return {"value": value, "ms_per_step": res["dt"] / res["steps"] * 1e3, "lk_avg_s": lk_avg, "achieved": achieved,
            "alg_lk": alg_lk, "alg_step": alg_step,
            "step_hbm_frac": alg_step * value / world / HBM_PEAK,
            "kernel_ms": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in res["prof"].items()}}
    # (kernel_ms["lk"]: HIP events inside the timed region; the other classes -- and "lk_all_events" -- over the
    # untimed steps right after it, with every class between events)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preroll", type=int, default=300,
                    help="untimed steps BEFORE the warm-up that take the feature tables out of their start-up transient: "
                         "right after the first detection most grid cells are full and the reference's early break skips "
                         "them, so the detector's share grows over the first few hundred steps (C2: score kernel 0.76 ms "
                         "per 1024 frames at steps 3-23, 1.25 ms from step 300 on); part of the workload set-up, like the "
                         "synthetic frames")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2", help="BASELINE.json workload (SURVEY 8(d))")
    ap.add_argument("--streams", type=int, default=0, help="independent streams (frames per step) per GPU; 0 = the config's")
    ap.add_argument("--seq", type=int, default=6, help="distinct frames per stream held in HBM")
    ap.add_argument("--occlusion", type=float, default=0.06,
                    help="fraction of every frame covered by moving occluder blocks (0 = pure smooth motion)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--torch-gather", action="store_true",
                    help="run the collective through torch.distributed instead of the library's vsf_gather_tracks")
    ap.add_argument("--detector", choices=("bucketing", "anms"), default="bucketing",
                    help="FeatureDetectorShiTomasiBucketing (the headline configuration) or FeatureDetectorANMS")
    ap.add_argument("--anms-min-distance", type=float, default=1.0,
                    help="with --detector anms: detection_min_feature_distance_ (above 1: cv's greedy rule)")
    ap.add_argument("--anms-harris", action="store_true", help="with --detector anms: Harris response")
    ap.add_argument("--tracker", choices=("lk", "lssd"), default="lk",
                    help="FeatureTrackerLucasKanade (the headline configuration) or FeatureTrackerLSSDLucasKanade")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE %d\n" % (args.gpus, world))
        sys.exit(2)
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
    env = (device, local_rank, rank, world, dist)

    headline = args.detector == "bucketing" and args.tracker == "lk"
    want_cpu = not args.no_cpu and world == 1 and headline
    res = run_workload(vsf, mg, args.config, args, env, args.steps, args.warmup, args.preroll, streams=args.streams,
                       keep_frames=want_cpu, probe=True)
    sm = summarize(res, world, args.tracker)
    cfg, B = res["cfg"], res["B"]
    W, H, LEVELS, WIN, MAXN = cfg["w"], cfg["h"], cfg["levels"], cfg["win"], cfg["maxn"]

    # short runs of the other configurations (same kernels, same measurement, fewer steps, no CPU leg); every rank
    # takes part, rank 0 reports.  With several GPUs only C5 is added (the one configuration defined over 8 GPUs).
    extras = {}
    if headline and not args.no_configs and args.config == "C2" and not args.streams:
        for name in (("C3", "C4", "C5") if world == 1 else ("C5",)):
            r = run_workload(vsf, mg, name, args, env, steps=(40 if name == "C5" else 10), warmup=2,
                             preroll=args.preroll)
            m = summarize(r, world, args.tracker)
            extras[name] = {"workload": r["cfg"]["name"], "streams_per_gpu": r["B"], "value": m["value"],
                            "unit": "frames/s", "steps": r["steps"], "ms_per_step": m["ms_per_step"],
                            "roofline_frac": m["achieved"] / (HBM_PEAK / 1e9), "step_hbm_frac": m["step_hbm_frac"],
                            "verified_fraction": r["kept_frac"], "kernel_ms": m["kernel_ms"]}

    if rank == 0:
        lk_avg = sm["lk_avg_s"]
        kernel_name = "lssd_fb_kernel" if args.tracker == "lssd" else "lk_fb_kernel"
        traffic, valu, traffic_note = None, None, None
        sclk = res.get("sclk_mhz_load")
        tpath = os.path.join(ROOT, "profiles", "lk_traffic.json")
        if args.tracker == "lk":
            try:
                tj = json.load(open(tpath))
            except (OSError, ValueError) as e:
                tj, traffic_note = None, "profiles/lk_traffic.json unreadable: %s" % e
            if tj is not None:
                if tj.get("csrc_sha16") != csrc_sha16():
                    traffic_note = ("profiles/lk_traffic.json was measured on other kernel sources (csrc_sha16 %s, built %s): "
                                    "counter figures withheld" % (tj.get("csrc_sha16"), csrc_sha16()))
                elif tj.get("streams") != B or tj.get("config_name", "C2") != args.config:
                    traffic_note = "profiles/lk_traffic.json holds %s / %s streams" % (tj.get("config_name"), tj.get("streams"))
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
                    insts = tj.get("valu_insts_per_launch")
                    if insts and lk_avg == lk_avg:
                        valu = {"insts_per_launch": insts, "achieved_ginst_s": insts / lk_avg / 1e9}
                        if sclk:
                            # wave64 vector instructions issued per SIMD and shader cycle, at the clock measured in
                            # THIS run under the step's load; the guide's issue peak is one per 2 cycles per SIMD,
                            # a SIMD-16 pipe passes a wave64 integer instruction in 4
                            cyc = N_SIMD * sclk * 1e6 * lk_avg / insts
                            valu.update({"sclk_mhz_under_load": sclk, "cycles_per_inst_per_simd": cyc,
                                         "peak_ginst_s_2cycle": N_SIMD * sclk * 1e6 / 2 / 1e9,
                                         "frac_of_2cycle_peak": 2.0 / cyc, "frac_of_4cycle_issue": 4.0 / cyc})
        line = {
            "metric": "frames/s detect+KLT-track @%dx%d" % (W, H) if args.tracker == "lk"
                      else "frames/s detect+LSSD-track @%dx%d" % (W, H),
            "value": sm["value"],
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": sm["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 (f32 2x2 solve)",
            "data": "synthetic",
            "config": {"workload": "%s, %s, %s / %d features, %d streams per GPU resident in HBM, "
                                   "%.0f%% of every frame under moving occluders"
                                   % (cfg["name"],
                                      "%d-level pyramid, %dx%d LK window" % (LEVELS, WIN, WIN) if args.tracker == "lk"
                                      else "LSSD SE(2) tracker (Pattern24, 3+1 u16 levels, 5 iterations)",
                                      "%dx%d grid" % (cfg["cols"], cfg["rows"]) if args.detector == "bucketing"
                                      else "ANMS detector", MAXN, B, 100 * args.occlusion),
                       "name": args.config, "streams_per_gpu": B, "frames_per_step": world * B,
                       "mean_features_per_stream": res["mean_feats"], "verified_fraction": res["kept_frac"],
                       "preroll_steps": max(0, args.preroll), "collective": res["gather"],
                       "lk_template_cache_bytes": res["lk_cache_bytes"]},
            # `frac` is the contract's figure: algorithmic bytes over the kernel's time against the HBM peak.  The
            # kernel is NOT limited by HBM: `bound` names what the counters show (vector instruction issue).
            "roofline": {"bound": "valu" if args.tracker == "lk" else "hbm", "kernel": kernel_name,
                         "achieved": sm["achieved"], "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": sm["achieved"] / (HBM_PEAK / 1e9), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "avg_launch_ms": lk_avg * 1e3, "algorithmic_bytes_per_launch": sm["alg_lk"] * B,
                         "valu_issue": valu, "sclk_mhz_under_load": sclk, "sclk_mhz_idle": res.get("sclk_mhz_idle"),
                         "stream_copy_GBps": stream_copy_gbps(device)},
            "step_hbm_frac": sm["step_hbm_frac"],
            "kernel_ms": sm["kernel_ms"],
            "kernel_ms_note": "lk: HIP events on the launch stream inside the timed region (the roofline's kernel); the "
                              "other classes, and lk_all_events, over the untimed steps right after it with every "
                              "class between events (an event pair drains the stream: all eight classes inside the "
                              "timed region cost the step 3-5 %)",
        }
        if extras:
            line["configs"] = extras
        if want_cpu:
            line["cpu_baseline"] = cpu_baseline(res["frames"], res["order"], cfg)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(frames, order, cfg):
    """Times the CPU oracle on the first streams of the same workload: one leg with OpenMP over streams on
    all host cores, one leg on a single thread; rank 0 at N=1 only, bounded."""
    from oracle import oracle as O
    O.build()
    W, H = cfg["w"], cfg["h"]

    def leg(S, steps, threads):
        S = min(S, frames.shape[1])
        host = frames[:, :S].cpu().numpy()  # [T][S][H][W]
        base = O.StreamsBaseline(S, W, H, cfg["maxn"] + 64, (cfg["rows"], cfg["cols"], cfg["maxn"], cfg["mind"]),
                                 (30, 0.01, cfg["win"], cfg["levels"] - 1, 1.0), omp=True, threads=threads)
        base.step(np.ascontiguousarray(host[order[0]]))  # first frame: detect only (untimed warm-up)
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            rc = base.step(np.ascontiguousarray(host[order[i % len(order)]]))
            assert rc == 0
        dt = time.perf_counter() - t0
        used = base.threads
        base.close()
        return S * steps / dt, used, S, steps, dt

    (sa, ka), (s1, k1) = cfg["cpu"]
    va, ta, sa, ka, da = leg(sa, ka, 0)
    v1, t1, s1, k1, d1 = leg(s1, k1, 1)
    # ONE stream on all cores: threads over the rows of the pyramid / derivative / response images and over the
    # points of the LK call, the way OpenCV's parallel_for_ spends the cores of the reference's single camera
    # (the fastest of a few team sizes: on a host shared with other jobs the full team is not the quickest)
    vl, tl, sl, kl, dl = max((leg(1, 2 * k1, t) for t in (4, 8, 0)), key=lambda r: r[0])
    return {"value": va, "unit": "frames/s", "cores": ta, "kind": "port",
            "single_thread": {"value": v1, "cores": t1,
                              "sample": "%d streams x %d steps, %.1f s" % (s1, k1, d1)},
            "single_stream": {"value": vl, "ms_per_frame": 1e3 / vl, "cores": tl,
                              "sample": "1 stream x %d steps, OpenMP over rows and points, %.1f s" % (kl, dl)},
            "sample": "%d streams x %d steps of the same workload, CPU oracle (restated reference CPU path, "
                      "gcc -O3 -march=x86-64-v3 -ffp-contract=off, OpenMP over streams), %.1f s; timed from the second frame "
                      "on, i.e. including the cheaper start-up transient the GPU line's pre-roll skips" % (sa, ka, da)}


if __name__ == "__main__":
    main()
