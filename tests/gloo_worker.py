"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 gloo group on the CPU.
Each rank steps ITS shard of independent streams (compute = the CPU oracle, standing in for the
GPU), packs fixed-size track records, and takes part in the single all_gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package, load_synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

TOTAL_STREAMS, CAP, W, H, FRAMES = 4, 96, 160, 120, 3


def step_stream(stream):
    synth = load_synth()
    frames, _ = synth.sequence(500 + stream, W, H, FRAMES)
    det, trk, tab = O.Detector(2, 2, 40, 6.0), O.Tracker(30, 0.01, 21, 3, 1.0), O.FeatureTable(CAP)
    for f in frames:
        trk.track(tab, f)
        det.update_detection(tab, f)
    return tab


def main():
    load_package()
    import importlib
    mg = importlib.import_module("vi_slam_amd.multigpu")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    mine = mg.shard_streams(TOTAL_STREAMS, world, rank)
    rec = np.zeros((len(mine), CAP), mg.RECORD_DTYPE)
    for local, s in enumerate(mine):
        tab = step_stream(s)
        n = tab.n.value
        rec[local, :n]["id"] = tab.ids[:n]
        rec[local, :n]["x"] = tab.xy[:n, 0]
        rec[local, :n]["y"] = tab.xy[:n, 1]
        rec[local, :n]["age"] = tab.age[:n]
        rec[local, :n]["valid"] = 1
    t = torch.frombuffer(bytearray(rec.tobytes()), dtype=torch.uint8)
    dist.barrier()
    gathered = mg.gather_tracks(t, world, dist)
    allt = mg.unpack_tracks(gathered, world, len(mine), CAP)
    ok = sorted(allt) == list(range(TOTAL_STREAMS))
    for s in range(TOTAL_STREAMS):   # every rank checks the full gathered result against a local recompute
        exp = step_stream(s).as_dict()
        got = allt[s]
        ok = ok and got.keys() == exp.keys() and all(
            got[i] == (float(np.float32(exp[i][0])), float(np.float32(exp[i][1])), exp[i][2]) for i in exp)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    open(os.environ["GLOO_TEST_OUT"] + ".%d" % rank, "w").write("ok" if flag.item() == 1 else "fail")


if __name__ == "__main__":
    main()
