"""World-size-2 gloo test of bench.py's multi-process harness (batch sharding, barrier,
max-over-ranks) on CPU.  The hot path has no collective, so this is all the N>1
logic there is."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_batch_partitions_exactly():
    sys.path.insert(0, ROOT)
    import bench
    for gb in (8, 4096, 65536, 65539, 7):
        for world in (1, 2, 3, 8):
            pieces = [bench.shard_batch(gb, r, world) for r in range(world)]
            assert sum(c for _, c in pieces) == gb
            pos = 0
            for s, c in pieces:
                assert s == pos
                pos += c
            assert max(c for _, c in pieces) - min(c for _, c in pieces) <= 1


def test_two_rank_gloo_harness():
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-dist"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["selftest"] and res["ok"] and res["world"] == 2
    assert res["counts"] == [4098, 4097] and res["starts"] == [0, 4098]
    assert res["max_s"] >= 0.02


def test_self_launch_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2 --selftest-dist` with RANK / WORLD_SIZE unset starts its own two ranks (children,
    before torch or HIP is touched in the parent) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-dist"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["selftest"] and res["ok"] and res["world"] == 2
    assert res["counts"] == [4098, 4097]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ)
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "WORLD_SIZE=2 but --gpus 4" in (out.stderr + out.stdout)
