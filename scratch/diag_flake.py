"""Run-to-run determinism under perturbed scheduling: the same batch sequence repeated, with and without the LK cache,
while another stream keeps the GPU busy with unrelated work of varying size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_package, load_synth
vsf = load_package(); synth = load_synth()
def scenario(w, h, B, T, maxn, rows, cols, R, noise_stream):
    frames = np.stack([synth.sequence(80 + s % 7, w, h, T)[0] for s in range(B)], 1)
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    junk = torch.randn(2048, 2048, device="cuda")
    def run(limit, r):
        ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B, max_features=maxn, grid_rows=rows, grid_cols=cols,
                                             min_distance=5.0, lk_cache_max_bytes=limit))
        outs = []
        for t in range(T):
            if noise_stream is not None:
                with torch.cuda.stream(noise_stream):
                    for _ in range(r % 5):
                        junk2 = junk @ junk
            ctx.batch_step(d[t].data_ptr(), h * w, w, torch.cuda.current_stream().cuda_stream)
        outs = [np.array(a).copy() for a in ctx.batch_download()]
        torch.cuda.synchronize()
        ctx.close()
        return outs
    ref = run(-1, 0)
    nbad = 0
    for r in range(R):
        l = (0, -1, 1 << 20)[r % 3]
        o = run(l, r)
        nn = ref[3]
        assert np.array_equal(nn, o[3])
        for k, (a, b) in enumerate(zip(ref[:3], o[:3])):
            a = np.concatenate([a[s, :nn[s]] for s in range(len(nn))])
            b = np.concatenate([b[s, :nn[s]] for s in range(len(nn))])
            if not np.array_equal(a, b):
                idx = np.argwhere(a != b)
                nbad += 1
                print("  run", r, "limit", l, "array", k, "ndiff", len(idx), "first", idx[:4].tolist(),
                      [(a[tuple(i)], b[tuple(i)]) for i in idx[:4]], flush=True)
    print("scenario", w, h, B, "runs", R, "bad", nbad, "n min", int(ref[3].min()), flush=True)
ns = torch.cuda.Stream()
scenario(320, 240, 4, 4, 100, 2, 2, 300, None)
scenario(320, 240, 4, 4, 100, 2, 2, 300, ns)
scenario(752, 480, 64, 5, 300, 4, 6, 40, ns)
scenario(752, 480, 256, 4, 300, 4, 6, 12, ns)
print("done")
