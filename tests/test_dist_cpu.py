"""CPU, world_size 2, gloo: the N>1 path of the benchmark -- round-robin pair sharding + the pose all-gather --
is correct by construction (the GPU tier only ever runs N=1 here; the driver runs N=2/4/8)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lv_slam_amd import dist as D


def test_shard_is_a_partition():
    for n in (1, 7, 271, 4541):
        for w in (1, 2, 4, 8):
            owned = [D.shard_pairs(n, r, w) for r in range(w)]
            flat = sorted(i for o in owned for i in o)
            assert flat == list(range(n))
            assert all(i % w == r for r, o in enumerate(owned) for i in o)
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def _fake_result(pid):
    rng = np.random.default_rng(pid)
    F = np.eye(4, dtype=np.float32)
    F[:3, 3] = rng.normal(size=3)
    return F, float(rng.normal()), int(rng.integers(2, 9)), True


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = D.shard_pairs(n_total, rank, world)
        cap = D.shard_capacity(n_total, world)
        res = [_fake_result(p) for p in mine]
        rec = D.pack_records(np.stack([r[0] for r in res]) if res else np.zeros((0, 4, 4), np.float32),
                             [r[1] for r in res], [r[2] for r in res], [r[3] for r in res], mine, capacity=cap)
        g = D.gather_records(rec)
        got = D.unpack_records(g)
        ok = sorted(got) == list(range(n_total))
        for pid, r in got.items():
            F, s, it, c = _fake_result(pid)
            ok = ok and np.array_equal(r["final"], F) and abs(r["score"] - np.float32(s)) < 1e-6 and r["iterations"] == it and r["converged"] == c
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _run_world(world, n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(out) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("n_total", [6, 7, 4541])      # 4541 = BASELINE config 4 (uneven shards: 2271 + 2270)
def test_gather_world2_gloo(n_total):
    _run_world(2, n_total)


@pytest.mark.parametrize("n_total", [13, 4541])        # 4541 over eight ranks: five shards of 568, three of 567 with a padding row each
def test_gather_world8_gloo(n_total):
    """The node the scaling run uses has eight GPUs (BASELINE config 4: 4,541 pairs round-robin over 8): eight ranks, uneven shards, padding
    rows on three of them, every record back bit-identical on every rank (independence anchor: scan_matching_odom_nodelet.cpp:240-250)."""
    cap = D.shard_capacity(n_total, 8)
    sizes = [len(D.shard_pairs(n_total, r, 8)) for r in range(8)]
    assert sum(sizes) == n_total and max(sizes) == cap and (n_total != 4541 or (sizes.count(568) == 5 and sizes.count(567) == 3))
    _run_world(8, n_total)


def test_records_roundtrip_single_process():
    F = np.stack([_fake_result(p)[0] for p in range(3)])
    rec = D.pack_records(F, [1.0, 2.0, 3.0], [3, 4, 5], [1, 0, 1], [10, 11, 12], capacity=5)
    assert rec.shape == (5, D.REC_WORDS) and rec.element_size() * D.REC_WORDS == 96 and rec.dtype == torch.int32
    got = D.unpack_records(D.gather_records(rec))
    assert sorted(got) == [10, 11, 12] and got[11]["converged"] is False and np.array_equal(got[12]["final"], F[2])


def test_pair_ids_beyond_float_precision_survive():
    """pair ids travel as int32 words, not as float32 values (exact beyond 2^24)."""
    ids = [16777217, 16777219, 2000000001]
    F = np.stack([np.eye(4, dtype=np.float32)] * 3)
    got = D.unpack_records(D.pack_records(F, [0.5, 1.5, 2.5], [1, 2, 3], [1, 1, 0], ids))
    assert sorted(got) == ids and got[2000000001]["iterations"] == 3 and got[16777217]["score"] == 0.5
