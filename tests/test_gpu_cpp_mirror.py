"""The C++ host mirror (vi-slam_amd/host/vsf_frontend.hpp: same class names and call pattern as the
reference's example_feature_detection_and_tracking.cpp) against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "frontend_mirror_main")


def test_mirror_binary_is_built():
    assert os.path.exists(EXE), "run __graft_entry__.build()"


def _oracle_detector(oracle, det_args):
    if det_args[0]:
        return oracle.Detector(*det_args)
    kw = dict(detection_use_harris_corner=1, detection_k=0.06, detection_min_feature_distance=3.5) if det_args[1] else {}
    return oracle.AnmsDetector(det_args[2], det_args[3], **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("det_args", [(3, 3, 180, 5.0), (0, 0, 120, 9.0), (0, 1, 120, 9.0)])
def test_cpp_mirror_matches_oracle(oracle, synth, tmp_path, det_args):
    w, h, n = 400, 300, 5
    frames, _ = synth.sequence(77, w, h, n)
    raw, out = str(tmp_path / "f.raw"), str(tmp_path / "o.txt")
    frames.tofile(raw)
    trk_args = (30, 0.01, 21, 3, 1.0)
    subprocess.check_call([EXE, raw, str(w), str(h), str(n), out] + [str(v) for v in det_args + trk_args], timeout=120)
    got, cur = [], None
    for line in open(out):
        p = line.split()
        if p[0] == "frame":
            cur = {"maxidx": int(p[3]), "tab": {}}
            got.append(cur)
        else:
            cur["tab"][int(p[0])] = (float(p[1]), float(p[2]), int(p[3]))
    det = _oracle_detector(oracle, det_args)
    trk, tab = oracle.Tracker(*trk_args, oracle.ACC_EXACT), oracle.FeatureTable(1024)
    for k in range(n):
        if k == 0:
            det.update_detection(tab, frames[0])
        else:
            if k % 2 == 0:
                det.update_detection(tab, frames[k])
            trk.track_pair(tab, frames[k - 1], frames[k])
        assert got[k]["tab"] == tab.as_dict(), "frame %d" % k
        assert got[k]["maxidx"] == det.max_feature_index
    assert len(got[-1]["tab"]) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("det_args", [(3, 3, 180, 5.0), (0, 0, 120, 9.0)])
def test_cpp_mirror_lssd_tracker_matches_oracle(oracle, synth, tmp_path, det_args):
    """FeatureTrackerLSSDLucasKanade through the C++ mirror (unordered databases: the table order
    changes between calls, the tracker's per-id state is matched by id)."""
    w, h, n = 400, 300, 6
    frames, _ = synth.sequence(78, w, h, n)
    raw, out = str(tmp_path / "f.raw"), str(tmp_path / "o.txt")
    frames.tofile(raw)
    subprocess.check_call([EXE, raw, str(w), str(h), str(n), out] + [str(v) for v in det_args + (0, 0, 0, 0, 0)],
                          timeout=120)
    got, cur = [], None
    for line in open(out):
        p = line.split()
        if p[0] == "frame":
            cur = {"maxidx": int(p[3]), "tab": {}}
            got.append(cur)
        else:
            cur["tab"][int(p[0])] = (float(p[1]), float(p[2]), int(p[3]))
    det = _oracle_detector(oracle, det_args)
    trk, tab = oracle.LssdTracker(mode=oracle.LSSD_TREE), oracle.FeatureTable(1024)
    for k in range(n):
        trk.track(tab, frames[k])
        if k % 2 == 0:
            det.update_detection(tab, frames[k])
        assert got[k]["tab"] == tab.as_dict(), "frame %d" % k
        assert got[k]["maxidx"] == det.max_feature_index
    assert len(got[-1]["tab"]) > 20


@pytest.mark.gpu
def test_abi_walk_from_cpp(synth, tmp_path):
    """tests/cpp/abi_exercise_main.cpp: every kind of C ABI entry from a C++ host, refused parameters and error
    paths included; it checks return codes and invariants (results are the other tests' business)."""
    exe = os.path.join(ROOT, "tests", "cpp", "abi_exercise_main")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    w, h, n = 400, 300, 5
    frames, _ = synth.sequence(79, w, h, n)
    raw = str(tmp_path / "f.raw")
    frames.tofile(raw)
    p = subprocess.run([exe, raw, str(w), str(h), str(n)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "abi exercise ok" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]
