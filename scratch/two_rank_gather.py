"""Two ranks on ONE GPU through the library's own RCCL entry (vsf_comm_* / vsf_gather_tracks): does RCCL accept
two ranks on one device here, and if it does, is the all-gather right?  Each rank is a fresh process; the unique id
travels through a file.   usage: python scratch/two_rank_gather.py            (parent; starts the ranks)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = os.path.join(ROOT, "gpurun_out", "two_rank")


def rank_main(rank):
    import numpy as np, torch
    from _pkg import load_package, load_synth
    vsf, synth = load_package(), load_synth()
    w, h, B = 160, 120, 3
    dev = rank if torch.cuda.device_count() >= 2 else 0   # two GPUs visible: the real thing
    torch.cuda.set_device(dev)
    frames = np.stack([synth.sequence(300 + 10 * rank + s, w, h, 2)[0] for s in range(B)], 1)
    ctx = vsf.Context(vsf.default_params(width=w, height=h, streams=B, max_features=40, grid_rows=2, grid_cols=2,
                                         min_distance=4.0, klt_win=9, klt_max_level=1, device_id=dev))
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    for t in range(2):
        ctx.batch_step(d[t].data_ptr(), w * h, w, 0)
    torch.cuda.synchronize()
    mine = torch.zeros(B * ctx.capacity * 24, dtype=torch.uint8, device="cuda")
    ctx.batch_export_tracks(mine.data_ptr(), 0)
    torch.cuda.synchronize()
    np.save(os.path.join(TMP, "mine%d.npy" % rank), mine.cpu().numpy())
    idf = os.path.join(TMP, "id.bin")
    if rank == 0:
        open(idf + ".tmp", "wb").write(bytes(vsf.comm_unique_id()))
        os.rename(idf + ".tmp", idf)
    else:
        while not os.path.exists(idf):
            time.sleep(0.05)
    ident = open(idf, "rb").read()
    print("rank", rank, "creating communicator", flush=True)
    comm = vsf.Comm(ident, rank, 2, dev)
    print("rank", rank, "communicator up", flush=True)
    recv = torch.zeros(2 * B * ctx.capacity * 24, dtype=torch.uint8, device="cuda")
    ctx.gather_tracks(comm, recv.data_ptr(), 0)
    torch.cuda.synchronize()
    np.save(os.path.join(TMP, "recv%d.npy" % rank), recv.cpu().numpy())
    comm.close()
    ctx.close()
    print("rank", rank, "done", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        rank_main(int(sys.argv[1]))
        sys.exit(0)
    import numpy as np
    os.makedirs(TMP, exist_ok=True)
    for f in os.listdir(TMP):
        os.remove(os.path.join(TMP, f))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)], env=env) for r in range(2)]
    t0 = time.time()
    rcs = [None, None]
    while time.time() - t0 < 60 and any(rc is None for rc in rcs):
        rcs = [p.poll() for p in ps]
        time.sleep(0.2)
    for p in ps:
        if p.poll() is None:
            p.kill()   # exactly the processes started above
    print("exit codes", rcs)
    if rcs == [0, 0]:
        m = [np.load(os.path.join(TMP, "mine%d.npy" % r)) for r in range(2)]
        want = np.concatenate(m)
        for r in range(2):
            got = np.load(os.path.join(TMP, "recv%d.npy" % r))
            print("rank", r, "gathered == [rank 0's records, rank 1's records]:", bool(np.array_equal(got, want)),
                  "valid records", int((want.view(np.uint32).reshape(-1, 6)[:, 5] == 1).sum()))
    else:
        print("two ranks on one GPU: not possible here (RCCL refused or timed out)")
