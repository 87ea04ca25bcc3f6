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


class NativeGather:
    """The same collective through the library itself (vsf_comm_* / vsf_gather_tracks, include/vsf.h): RCCL opened by
    libvsf_hip.so, no torch in the data path.  The 128-byte unique id travels over the caller's existing process
    group (any backend); every rank must construct it at the same point.  `ok` is False on every rank when any rank
    could not set it up (the caller then keeps gather_tracks above)."""

    def __init__(self, vsf, ctx, world, rank, device_index, dist=None, device=None):
        import torch
        self.ctx, self.world, self.comm, self.ok, self.error = ctx, world, None, True, ""
        self.recv = torch.empty((world * ctx.streams * ctx.capacity * RECORD_DTYPE.itemsize,), dtype=torch.uint8,
                                device=device if device is not None else "cuda")
        if world == 1:
            return  # a world of one: export only, no communicator, no librccl
        ident = torch.zeros(vsf.COMM_ID_BYTES, dtype=torch.uint8)
        flag = 1
        try:
            if rank == 0:
                ident = torch.frombuffer(bytearray(vsf.comm_unique_id()), dtype=torch.uint8).clone()
        except Exception as e:  # noqa: BLE001 -- reported, and the torch path takes over
            flag, self.error = 0, str(e)
        dev = self.recv.device if dist.get_backend() == "nccl" else torch.device("cpu")
        ident = ident.to(dev)
        dist.broadcast(ident, src=0)
        good = torch.tensor([flag], device=dev)
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        if int(good.item()) == 1:
            try:
                self.comm = vsf.Comm(bytes(ident.cpu().numpy().tobytes()), rank, world, device_index)
            except Exception as e:  # noqa: BLE001
                flag, self.error = 0, str(e)
            good = torch.tensor([flag], device=dev)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
        self.ok = int(good.item()) == 1
        if not self.ok and self.comm is not None:
            self.comm.close()
            self.comm = None

    def gather(self, stream=0):
        """Enqueues export + all-gather on `stream`; returns the list of per-rank byte tensors (views of one buffer)."""
        self.ctx.gather_tracks(self.comm, self.recv.data_ptr(), stream)
        return list(self.recv.view(self.world, -1))

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def unpack_tracks(gathered, world, streams_per_rank, capacity):
    """-> {global stream index: {id: (x, y, age)}} from the gathered byte tensors."""
    res = {}
    for rank, t in enumerate(gathered):
        a = np.frombuffer(t.cpu().numpy().tobytes(), dtype=RECORD_DTYPE).reshape(streams_per_rank, capacity)
        for local in range(streams_per_rank):
            rows = a[local][a[local]["valid"] == 1]
            res[rank + local * world] = {int(r["id"]): (float(r["x"]), float(r["y"]), int(r["age"])) for r in rows}
    return res
