"""C1 end to end: an EuRoC-format directory on disk -> EurocKimeraDataProvider -> detector + tracker on
the GPU, driven exactly like the reference's example_feature_detection_and_tracking.cpp:33-58, against
the CPU oracle fed with the raw frames the dataset was written from."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from test_euroc_ingest import make_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "euroc_example_main")


def _oracle_run(oracle, frames):
    det = oracle.Detector(4, 4, 300, 2.0)
    trk, tab = oracle.Tracker(30, 0.01, 15, 3, 1.0, oracle.ACC_EXACT), oracle.FeatureTable(1024)
    out, counter = [], 0
    for k in range(len(frames)):
        if k == 0:
            det.update_detection(tab, frames[0])
        else:
            if counter % 10 == 0:
                det.update_detection(tab, frames[k])
            counter += 1
            trk.track_pair(tab, frames[k - 1], frames[k])
        out.append(tab.as_dict())
    return out


def test_example_binary_is_built():
    assert os.path.exists(EXE), "run __graft_entry__.build()"


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n", [(376, 240, 13), (752, 480, 12)])  # the second: C1 at its own size (EuRoC cam0)
def test_cpp_example_from_disk_matches_oracle(oracle, synth, tmp_path, w, h, n):
    frames, _ = synth.sequence(91, w, h, n)
    root = str(tmp_path / "V1_01_easy")
    calib = make_dataset(root, frames)
    out = str(tmp_path / "o.txt")
    subprocess.check_call([EXE, root, calib, out], timeout=120)
    got, cur = [], None
    for line in open(out):
        p = line.split()
        if p[0] == "frame":
            cur = {"t": float(p[1]), "tab": {}}
            got.append(cur)
        else:
            cur["tab"][int(p[0])] = (float(p[1]), float(p[2]), int(p[3]))
    exp = _oracle_run(oracle, frames)
    assert len(got) == n
    for k in range(n):
        assert abs(got[k]["t"] - (1403715273262142976 + k * 50000000) * 1e-9) < 1e-6
        assert got[k]["tab"] == exp[k], "frame %d" % k
    assert len(got[-1]["tab"]) > 100


@pytest.mark.gpu
def test_python_provider_drives_the_gpu_path(vsf, oracle, synth, tmp_path):
    euroc = importlib.import_module("vi_slam_amd.euroc")
    w, h, n = 320, 240, 12
    frames, _ = synth.sequence(92, w, h, n)
    root = str(tmp_path / "MH_01_easy")
    calib = make_dataset(root, frames)
    dp = euroc.EurocKimeraDataProvider(root, calib)
    det = vsf.FeatureDetectorShiTomasiBucketing(4, 4, 300, 2.0, w, h, capacity=1024)
    trk = vsf.FeatureTrackerLucasKanade(30, 0.01, 15, 3, 1.0, w, h, capacity=1024)
    db = vsf.FeatureDatabase(1024)
    exp = _oracle_run(oracle, frames)
    prev, counter, k = None, 0, 0
    while True:
        inp = dp.GetInput()
        if inp is None:
            break
        if prev is None:
            det.UpdateDetection(db, inp.frame_)
        else:
            if counter % 10 == 0:
                det.UpdateDetection(db, inp.frame_)
            counter += 1
            trk.Track(db, prev.frame_, inp.frame_)
        assert db.as_dict() == exp[k], "frame %d" % k
        prev = inp
        k += 1
    assert k == n
