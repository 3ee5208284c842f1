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


def test_global_dataset_is_indexed_by_transform_not_by_rank():
    """bench.make_host_block: local transform s of a rank whose slice starts at gstart holds global transform gstart + s,
    for every rank (no per-rank seed), so a slice can be checked against the global dataset."""
    sys.path.insert(0, ROOT)
    import numpy
    import bench
    shape, dtname, batch, seed = (64,), "complex64", 200, 11
    blk, _, _, whole = bench.make_host_block(shape, dtname, batch, seed, 0)
    assert blk == 64
    for world in (2, 3):
        for rank in range(world):
            gstart, count = bench.shard_batch(batch * world, rank, world)
            b2, re, im, local = bench.make_host_block(shape, dtname, batch, seed, gstart)
            assert b2 == blk
            for s in (0, 1, blk - 1, blk, count - 1):
                g = gstart + s
                assert numpy.array_equal(local[s % blk], whole[g % blk])
                assert numpy.array_equal(local[s % blk], bench.global_item(shape, dtname, batch, seed, g))
                assert numpy.array_equal(re[s % blk] + 1j * im[s % blk], local[s % blk])


def test_visible_gpus_does_not_need_the_runtime(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3")
    assert bench.visible_gpus() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpus()
    assert n is None or n >= 0
    assert "torch" not in bench.self_launch.__code__.co_names      # the parent only spawns children
