import sys, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from _pkg import load_package
from oracle import oracle
oracle.build()
vsf = load_package()
rng = np.random.default_rng(8)
mats = []; tags = []
for scale in ([1, 1, 1], [1, 5, 30], [30, 5, 1], [5, 30, 1], [5, 1, 30], [1, 30, 5], [30, 1, 5]):
    for _ in range(300):
        J = (rng.normal(size=(24, 3)) * np.array(scale)).astype(np.float32)
        mats.append((J.T @ J).astype(np.float32)); tags.append("spd%s" % scale)
for v in (1.0, 2.5, 0.0):
    mats.append(np.eye(3, dtype=np.float32) * v); tags.append("eye%g" % v)
for dd in ([1, 1, 2], [2, 1, 1], [1, 2, 2], [0, 3, 0], [-4, 1, 2]):
    mats.append(np.diag(np.array(dd, np.float32))); tags.append("diag%s" % dd)
sym = rng.normal(size=(200, 3, 3)).astype(np.float32)
for m in sym + sym.transpose(0, 2, 1):
    mats.append(m); tags.append("indef")
H = np.stack(mats).astype(np.float32)
ctx = vsf.Context(vsf.default_params(tracker_kind=vsf.TRACKER_LSSD, width=64, height=48))
got = ctx.debug_lssd_inverse3(H)
ref = np.zeros_like(H)
fn = oracle.lib().orc_lssd_debug_inverse3
for i in range(len(H)):
    fn(H[i].ctypes.data_as(C.c_void_p), ref[i].ctypes.data_as(C.c_void_p))
bad = ~((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref)))
idx = np.where(bad.any(axis=(1, 2)))[0]
print("mismatching matrices:", len(idx), "of", len(H))
from collections import Counter
print(Counter(tags[i] for i in idx))
for i in idx[:4]:
    print(tags[i], "\nH=", H[i], "\ngot=", got[i], "\nref=", ref[i], "\ndiff=", got[i] - ref[i])
