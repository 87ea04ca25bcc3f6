"""The host-side file readers (PNG / csv / yaml / json, vi-slam_amd/csrc/vsf_io.cpp) under AddressSanitizer and
UBSan: the data set as written, then a few thousand mutated PNGs and text files.  CPU only; the driver and the
library source are compiled together with g++ -fsanitize=address,undefined (tests/cpp/io_sanitize_main.cpp)."""
import os
import subprocess

import numpy as np

from test_euroc_ingest import _png, make_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_io_readers_under_sanitizers(synth, tmp_path):
    exe = str(tmp_path / "io_sanitize_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "io_sanitize_main.cpp"),
                           os.path.join(ROOT, "vi-slam_amd", "csrc", "vsf_io.cpp"), "-o", exe, "-lz"])
    w, h = 96, 64
    frames, _ = synth.sequence(5, w, h, 4)
    root = str(tmp_path / "V1_01_easy")
    calib = make_dataset(root, frames)
    mask = np.zeros((h, w), np.uint8)
    mask[:, 40:] = 200
    mpath = str(tmp_path / "mask.png")
    open(mpath, "wb").write(_png.encode(mask))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None) if False else None   # (the harness's preload stays as it is)
    for seed in (1, 2):
        p = subprocess.run([exe, root, calib, mpath, str(seed), "3000"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        assert "frames 4 size 96x64" in p.stdout and "mutations:" in p.stdout
