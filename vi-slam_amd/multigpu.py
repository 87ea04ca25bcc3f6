"""Multi-GPU host logic: independent streams are dealt round-robin to ranks
(stream s -> rank s mod world), every rank keeps its streams' state resident, and the only
collective is one all_gather of fixed-size track records at the end of a reporting interval
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)."""
import numpy as np

# layout written by vsf_batch_export_tracks (vi-slam_amd/csrc/vsf_kernels.hpp: TrackRecord), 24 bytes
RECORD_DTYPE = np.dtype([("id", "<u8"), ("x", "<f4"), ("y", "<f4"), ("age", "<u4"), ("valid", "<u4")])
assert RECORD_DTYPE.itemsize == 24


def shard_streams(total_streams, world, rank):
    """Global stream indices owned by `rank`."""
    return list(range(rank, total_streams, world))


def owner_of(stream, world):
    return stream % world


def gather_tracks(records, world, dist=None):
    """records: uint8 tensor [streams_per_rank * capacity * 24] on this rank's device.
    Returns the list of every rank's records (one collective; payload is tiny, latency bound)."""
    if world == 1 or dist is None:
        return [records]
    import torch
    out = [torch.empty_like(records) for _ in range(world)]
    dist.all_gather(out, records)
    return out


def unpack_tracks(gathered, world, streams_per_rank, capacity):
    """-> {global stream index: {id: (x, y, age)}} from the gathered byte tensors."""
    res = {}
    for rank, t in enumerate(gathered):
        a = np.frombuffer(t.cpu().numpy().tobytes(), dtype=RECORD_DTYPE).reshape(streams_per_rank, capacity)
        for local in range(streams_per_rank):
            rows = a[local][a[local]["valid"] == 1]
            res[rank + local * world] = {int(r["id"]): (float(r["x"]), float(r["y"]), int(r["age"])) for r in rows}
    return res
