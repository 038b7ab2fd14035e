"""N > 1 path on CPU: two gloo ranks shard a seeded batch by contiguous ranges (no data-path collective),
each evaluates its shard (plaintext look-ups stand in for the GPU PBS here), and the union equals the
single-rank result; the timing reduction used by bench.py takes the MAX over ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bmi_amd.shard import reduce_max, shard_range, barrier


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 84, 8192, 8193):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    msgs = rng.integers(-8, 8, 1001)                 # the whole job's batch (same seed on every rank)
    table = np.random.default_rng(99).integers(-8, 8, 16)
    a, b = shard_range(msgs.size, rank, world)
    mine = table[msgs[a:b] + 8]                      # this rank's share of the look-ups
    barrier()
    elapsed = reduce_max(0.5 + rank)                 # MAX over ranks, as bench.py does
    np.save(os.path.join(out_dir, f"r{rank}.npy"), mine)
    if rank == 0:
        np.save(os.path.join(out_dir, "elapsed.npy"), np.array([elapsed]))
    barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")])
    msgs = np.random.default_rng(1234).integers(-8, 8, 1001)
    table = np.random.default_rng(99).integers(-8, 8, 16)
    assert np.array_equal(got, table[msgs + 8])
    assert float(np.load(tmp_path / "elapsed.npy")[0]) == 1.5
