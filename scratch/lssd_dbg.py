import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from _pkg import load_package, load_synth
from oracle import oracle
oracle.build()
vsf, synth = load_package(), load_synth()
w, h, maxn = 752, 480, 300
frames, _ = synth.sequence(41 + w, w, h, 6)
P = vsf.FeatureTrackerLSSDLucasKanade.Parameter()
trk = vsf.FeatureTrackerLSSDLucasKanade(P, w, h, capacity=512)
det = vsf.FeatureDetectorShiTomasiBucketing(3, 3, maxn, 6.0, w, h, capacity=512)
db = vsf.FeatureDatabase(512)
otrk = oracle.LssdTracker(0.09, 3, 5, oracle.LSSD_TREE)
odet = oracle.Detector(3, 3, maxn, 6.0)
T = oracle.FeatureTable(512)
for t, fr in enumerate(frames):
    trk.Track(db, fr); otrk.track(T, fr)
    print(t, "track", db.n, T.n.value, np.array_equal(db.xy[:db.n], T.xy[:db.n]))
    pre = db.xy[:db.n].copy()
    det.UpdateDetection(db, fr); odet.update_detection(T, fr)
    print(t, "det", db.n, T.n.value)
    if db.n != T.n.value:
        a = {int(i): tuple(x) for i, x in zip(db.ids[:db.n], db.xy[:db.n])}
        b = {int(i): tuple(x) for i, x in zip(T.ids[:T.n.value], T.xy[:T.n.value])}
        sa = set(a.values()); sb = set(b.values())
        print("only gpu", sorted(sa - sb)); print("only oracle", sorted(sb - sa))
        for q in sorted(sb - sa) + sorted(sa - sb):
            d = np.hypot(pre[:, 0] - q[0], pre[:, 1] - q[1]); j = d.argmin()
            print(q, "nearest tracked", pre[j].tolist(), d[j], "cell", int(q[0] // (w / 3)), int(q[1] // (h / 3)))
        break
