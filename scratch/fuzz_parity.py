"""Randomised parity sweep (GPU vs oracle) over sizes, parameters and seeds; prints every mismatch.
usage: python scratch/fuzz_parity.py <seconds> [seed]"""
import sys, time, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from _pkg import load_package, load_synth
from oracle import oracle as O
O.build()
vsf, synth = load_package(), load_synth()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
ncase = nbad = nrej = 0
kinds = {"lk+bucket": 0, "lssd+bucket": 0, "lk+anms": 0, "lssd+anms": 0}
while time.time() < t_end:
    small = len(sys.argv) > 3 and sys.argv[3] == "small"   # tiny frames, extreme parameters
    if small:
        w = int(rng.integers(24, 80)); h = int(rng.integers(24, 64))
    else:
        w = int(rng.integers(64, 420)); h = int(rng.integers(48, 320))
    if rng.random() < 0.3:
        w = (w // 4) * 4
    seed = int(rng.integers(0, 1 << 30))
    nfr = int(rng.integers(2, 5))
    kind = list(kinds)[int(rng.integers(0, 4))]
    maxn = int(rng.integers(10, 200)); mind = float(rng.choice([0.5, 1.0, 2.0, 3.7, 6.0, 10.0, 15.5]))
    rows, cols = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    win = int(rng.choice([5, 9, 15, 21, 25, 31])); lvl = int(rng.integers(0, 4)); it = int(rng.choice([1, 5, 30])); eps = float(rng.choice([0.01, 0.1]))
    back = float(rng.choice([0.5, 1.0, 2.0]))
    llev = int(rng.integers(0, 4)); lit = int(rng.choice([1, 5, 8])); ldist = float(rng.choice([0.04, 0.09, 1.0]))
    if small:
        maxn = int(rng.choice([0, 1, 2, 7, 40])); mind = float(rng.choice([0.5, 1.0, 5.0, 30.0]))
        win = int(rng.choice([3, 5, 9, 15, 21])); lvl = int(rng.integers(0, 6)); llev = int(rng.integers(0, 5))
        if win >= min(w, h) or (min(w, h) >> llev) < 1:
            continue
    elif min(w, h) >> max(lvl, llev) < 16:
        continue
    desc = dict(kind=kind, w=w, h=h, seed=seed, nfr=nfr, maxn=maxn, mind=mind, grid=(rows, cols), lk=(it, eps, win, lvl, back), lssd=(llev, lit, ldist))
    try:
        frames, _ = synth.sequence(seed % 100000, w, h, nfr)
        if rng.random() < 0.3:
            frames = np.clip(frames.astype(np.int32) + rng.integers(-12, 13, frames.shape), 0, 255).astype(np.uint8)
        cap = 1024
        if kind.endswith("anms"):
            if mind < 1.0 or int(mind) > 31:
                continue
            P = vsf.FeatureDetectorANMS.Parameter(); P.max_feature_number_ = maxn; P.min_feature_distance_ = mind
            P.detection_feature_number_ = 2 * maxn; P.detection_ignore_edge_ = int(rng.choice([0, 5, 20]))
            if w <= 2 * P.detection_ignore_edge_ + 8 or h <= 2 * P.detection_ignore_edge_ + 8:
                continue
            # the rest of the Parameter (round 3): Harris response, cv's minDistance above 1, block size, quality
            P.detection_use_harris_corner_ = bool(rng.random() < 0.35)
            P.detection_k_ = float(rng.choice([0.04, 0.06, 0.15]))
            P.detection_min_feature_distance_ = float(rng.choice([1.0, 1.0, 0.0, 1.3, 2.0, 4.5, 9.0, 31.0]))
            P.detection_block_size_ = int(rng.choice([3, 4, 4]))
            P.detection_quality_level_ = float(rng.choice([0.001, 0.001, 0.05, 0.0]))
            desc["anms"] = (P.detection_use_harris_corner_, P.detection_k_, P.detection_min_feature_distance_,
                            P.detection_block_size_, P.detection_quality_level_, P.detection_ignore_edge_)
            det = vsf.FeatureDetectorANMS(P, w, h, capacity=cap)
            odet = O.AnmsDetector(maxn, mind, detection_feature_number=2 * maxn,
                                  detection_ignore_edge=P.detection_ignore_edge_,
                                  detection_use_harris_corner=int(P.detection_use_harris_corner_),
                                  detection_k=P.detection_k_,
                                  detection_min_feature_distance=P.detection_min_feature_distance_,
                                  detection_block_size=P.detection_block_size_,
                                  detection_quality_level=P.detection_quality_level_)
        else:
            det = vsf.FeatureDetectorShiTomasiBucketing(rows, cols, maxn, mind, w, h, capacity=cap)
            odet = O.Detector(rows, cols, maxn, mind)
        if kind.startswith("lssd"):
            P2 = vsf.FeatureTrackerLSSDLucasKanade.Parameter(); P2.optical_flow_levels_ = llev; P2.optical_flow_max_iterations_ = lit
            P2.optical_flow_max_recovered_dist_ = ldist
            trk = vsf.FeatureTrackerLSSDLucasKanade(P2, w, h, capacity=cap)
            otrk = O.LssdTracker(ldist, llev, lit, O.LSSD_TREE)
        else:
            trk = vsf.FeatureTrackerLucasKanade(it, eps, win, lvl, back, w, h, capacity=cap)
            otrk = O.Tracker(it, eps, win, lvl, back, O.ACC_EXACT)
        db, T = vsf.FeatureDatabase(cap), O.FeatureTable(cap)
        mask = None
        if rng.random() < 0.3:
            mask = (rng.random((h, w)) < 0.8).astype(np.uint8) * int(rng.choice([1, 255, 7]))
        ok = True
        for k in range(nfr):
            trk.Track(db, frames[k]); otrk.track(T, frames[k])
            if db.as_dict() != T.as_dict():
                ok = False; print("MISMATCH after track", k, desc, flush=True); break
            det.UpdateDetection(db, frames[k], mask); odet.update_detection(T, frames[k], mask)
            if db.as_dict() != T.as_dict():
                ok = False; print("MISMATCH after detect", k, desc, len(db.as_dict()), len(T.as_dict()), flush=True); break
        ncase += 1; kinds[kind] += 1
        nbad += 0 if ok else 1
        trk.ctx.close(); det.ctx.close()
    except vsf.VsfError as e:
        # a grid whose rounded cells leave the image: cv::Mat::operator()(Rect) throws in the reference, the
        # oracle returns -2, the library refuses the context (VSF_E_GEOMETRY)
        if e.code == -4 and kind.endswith("bucket"):
            # refused at creation when ANY cell leaves the image (the oracle / reference only when the loop reaches it)
            nrej += 1
        elif e.code == -1 and kind.endswith("anms") and maxn == 0:
            nrej += 1   # anms_detection_features must be 1..2048
        else:
            print("EXC", desc, repr(e), flush=True)
            nbad += 1
    except Exception as e:
        print("EXC", desc, repr(e), flush=True)
        traceback.print_exc()
        nbad += 1
    if ncase % 50 == 0:
        print("progress", ncase, "cases", nbad, "bad", kinds, flush=True)
print("done", ncase, "cases", nbad, "bad", nrej, "geometries refused by both", kinds)
