"""N>1 host path on the CPU: world_size 2, gloo, 127.0.0.1.  Streams are sharded round-robin with
no data-path collective; the one all_gather of 24-byte track records must reassemble every
stream's table on every rank."""
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_and_owner(vsf):
    import importlib
    mg = importlib.import_module("vi_slam_amd.multigpu")
    assert mg.shard_streams(64, 8, 3) == list(range(3, 64, 8))
    assert sorted(sum((mg.shard_streams(10, 4, r) for r in range(4)), [])) == list(range(10))
    assert all(mg.owner_of(s, 8) == s % 8 for s in range(64))


def test_world2_gather(tmp_path):
    port = _free_port()
    out = str(tmp_path / "res")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_TEST_OUT=out, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "gloo_worker.py")], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    for rank in range(2):
        assert open(out + ".%d" % rank).read() == "ok"
