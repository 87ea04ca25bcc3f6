"""The timed CPU baseline (oracle built with OpenMP, bench.py's cpu_baseline leg) gives the same tables however it
spends its threads: over streams, over the rows and points of one stream, or on one thread -- and the same as the
plain build the parity tests use."""
import numpy as np


def _run(oracle, frames, S, threads):
    T, _, h, w = frames.shape
    base = oracle.StreamsBaseline(S, w, h, 200, (2, 3, 120, 6.0), (30, 0.01, 21, 3, 1.0), omp=True, threads=threads)
    for t in range(T):
        assert base.step(np.ascontiguousarray(frames[t, :S])) == 0
    out = [(int(base.n[s]), base.ids[s, :base.n[s]].copy(), base.xy[s, :base.n[s]].copy(),
            base.age[s, :base.n[s]].copy()) for s in range(S)]
    used = base.threads
    base.close()
    return out, used


def test_baseline_threading_modes_agree(oracle, synth):
    w, h, S = 376, 240, 3
    frames = np.stack([synth.sequence(60 + s, w, h, 4)[0] for s in range(S)], 1)   # [T][S][h][w]
    over_streams, _ = _run(oracle, frames, S, 0)
    one_thread, used1 = _run(oracle, frames, S, 1)
    assert used1 == 1
    for s in range(S):
        inner, used = _run(oracle, frames[:, s:s + 1], 1, 0)    # ONE stream: threads over rows and points
        assert used >= 1
        for got in (over_streams[s], one_thread[s]):
            assert got[0] == inner[0][0] > 50
            assert all(np.array_equal(a, b) for a, b in zip(got[1:], inner[0][1:]))
    # the plain (single-threaded, no OpenMP) build the parity tests check the GPU against
    det, trk, tab = oracle.Detector(2, 3, 120, 6.0), oracle.Tracker(30, 0.01, 21, 3, 1.0), oracle.FeatureTable(200)
    for t in range(frames.shape[0]):
        trk.track(tab, frames[t, 0])
        det.update_detection(tab, frames[t, 0])
    n = tab.n.value
    assert n == over_streams[0][0] and np.array_equal(tab.xy[:n], over_streams[0][2])
    assert np.array_equal(tab.ids[:n], over_streams[0][1])
