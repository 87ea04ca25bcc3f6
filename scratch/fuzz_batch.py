"""Randomised parity sweep of the BATCHED device API (vsf_batch_step on B streams) against per-stream oracles.
usage: python scratch/fuzz_batch.py <seconds> [seed]"""
import sys, time, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from _pkg import load_package, load_synth
from oracle import oracle as O
O.build()
vsf, synth = load_package(), load_synth()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
ncase = nbad = nrej = 0
while time.time() < t_end:
    w = int(rng.integers(16, 100)) * 4 if rng.random() < 0.6 else int(rng.integers(64, 400))
    h = int(rng.integers(48, 300))
    B = int(rng.integers(2, 12)); T = int(rng.integers(2, 5))
    lssd = rng.random() < 0.5; anms = rng.random() < 0.4
    maxn = int(rng.integers(10, 150)); mind = float(rng.choice([1.0, 2.0, 3.7, 6.0, 10.0]))
    rows, cols = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    win = int(rng.choice([9, 15, 21, 31])); lvl = int(rng.integers(0, 4))
    llev = int(rng.integers(0, 4))
    if min(w, h) >> max(lvl, llev) < 16:
        continue
    desc = dict(w=w, h=h, B=B, T=T, lssd=lssd, anms=anms, maxn=maxn, mind=mind, grid=(rows, cols), win=win, lvl=lvl, llev=llev)
    try:
        kw = dict(width=w, height=h, streams=B, max_features=maxn, min_distance=mind, klt_win=win, klt_max_level=lvl,
                  tracker_kind=vsf.TRACKER_LSSD if lssd else vsf.TRACKER_LK, lssd_levels=llev)
        if anms:
            kw.update(detector_kind=vsf.DETECTOR_ANMS, anms_detection_features=2 * maxn)
            if w <= 48 or h <= 48:
                continue
        else:
            kw.update(grid_rows=rows, grid_cols=cols)
        try:
            ctx = vsf.Context(vsf.default_params(**kw))
        except vsf.VsfError as e:
            if e.code == -4:
                nrej += 1
                continue
            raise
        frames = np.stack([synth.sequence(int(rng.integers(0, 100000)), w, h, T)[0] for _ in range(B)], 1)
        dev = torch.from_numpy(frames).cuda()
        dets = [O.AnmsDetector(maxn, mind, detection_feature_number=2 * maxn) if anms else O.Detector(rows, cols, maxn, mind) for _ in range(B)]
        trks = [O.LssdTracker(0.09, llev, 5, O.LSSD_TREE) if lssd else O.Tracker(30, 0.01, win, lvl, 1.0, O.ACC_EXACT) for _ in range(B)]
        tabs = [O.FeatureTable(ctx.capacity) for _ in range(B)]
        ok = True
        for t in range(T):
            ctx.batch_step(dev[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
            ids, xy, age, n, _ = ctx.batch_download()
            for b in range(B):
                trks[b].track(tabs[b], frames[t, b]); dets[b].update_detection(tabs[b], frames[t, b])
                got = {int(i): (float(q[0]), float(q[1]), int(a)) for i, q, a in zip(ids[b, :n[b]], xy[b, :n[b]], age[b, :n[b]])}
                if got != tabs[b].as_dict():
                    ok = False; print("MISMATCH step", t, "stream", b, desc, len(got), len(tabs[b].as_dict()), flush=True); break
            if not ok:
                break
        if (ctx.batch_status() != 0).any():
            print("STATUS", desc, ctx.batch_status(), flush=True); ok = False
        ncase += 1; nbad += 0 if ok else 1
        ctx.close()
    except Exception as e:
        print("EXC", desc, repr(e), flush=True); traceback.print_exc(); nbad += 1
    if ncase % 25 == 0:
        print("progress", ncase, "cases", nbad, "bad", flush=True)
print("done", ncase, "cases", nbad, "bad", nrej, "refused geometries")
