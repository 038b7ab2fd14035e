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


# ----------------------------------------------------------------------------------------------------------------
# The sharded level executor (bmi_amd/executor.py) under two gloo ranks.  The GPU engine is replaced by a plaintext
# stand-in with the same call surface (a "ciphertext" is a row whose last word is the message at delta_log = 1: the executor
# keeps constants in units of Delta / 2 for the half-scale tables of Circuit.lut_neg), so
# what is tested here is the host logic: padded level regions, per-rank row ranges, the in-place all-gather and the
# threshold below which levels are computed redundantly.  The same executor runs on GPUs in tests/test_gpu_inverse.py.
class _PlainParams:
    big = 3


class _PlainEngine:
    P = _PlainParams()
    modulus = 1 << 64
    device = "cpu"

    def __init__(self):
        self.tables = []

    def torch_device(self):
        return torch.device("cpu")

    def delta_log(self, msg_bits=4):
        return 1

    def lut_register(self, table, msg_bits, out_delta_log):
        self.tables.append((int(msg_bits), [int(v) << int(out_delta_log) for v in table]))   # out_delta_log 0: a lut_neg table
        return len(self.tables) - 1

    def reserve(self, n):
        pass

    def lincomb(self, store, rp, ix, cf, cs, count, out, stream=0):
        for r in range(count):
            acc = torch.zeros(self.P.big, dtype=torch.int64)
            for e in range(int(rp[r]), int(rp[r + 1])):
                acc += int(cf[e]) * store[int(ix[e])]
            acc[-1] += int(cs[r])
            out[r] = acc

    def scatter_rows(self, src, count, store, rows, stream=0):
        for r in range(count):
            store[int(rows[r])] = src[r]

    def pbs(self, d_in, ids, count, d_out, stream=0):
        for r in range(count):
            p, table = self.tables[int(ids[r])]
            x = int(d_in[r, -1])
            assert x % 2 == 0                     # whole multiples of Delta reach a look-up
            x >>= 1
            m = x >> (4 - p)
            half = 1 << (p - 1)
            d_out[r] = 0
            if m >= half:          # the other half of the torus (Circuit.lut_odd inputs): negacyclic wrap-around
                d_out[r, -1] = -table[m - 2 * half + half]
            elif m < -half:
                d_out[r, -1] = -table[m + 2 * half + half]
            else:
                d_out[r, -1] = table[m + half]


def _trace_small():
    from bmi_amd.main import trace_inverse
    return trace_inverse(2, 8, 4, 2, False, False)


def _exec_worker(rank, world, port, out_dir, threshold):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmi_amd.executor import Executor
    circ = _trace_small()
    ex = Executor(circ, _PlainEngine(), shard_threshold=threshold)
    inputs = np.load(os.path.join(out_dir, "inputs.npy"))
    cts = np.zeros((circ.n_inputs, 3), np.uint64)
    cts[:, -1] = (2 * inputs.astype(np.int64)).view(np.uint64)
    out = ex.run(cts)
    assert not (out[:, -1].view(np.int64) % 2).any()
    np.save(os.path.join(out_dir, f"out{rank}.npy"), out[:, -1].view(np.int64) // 2)
    np.save(os.path.join(out_dir, f"sharded{rank}.npy"), np.array([ex.sharded_levels, len(ex.levels)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_executor_matches_simulation(tmp_path):
    circ = _trace_small()
    rng = np.random.default_rng(5)
    inputs = np.array([rng.integers(lo, hi + 1) for lo, hi in zip(circ.leaf_lo[:circ.n_inputs], circ.leaf_hi[:circ.n_inputs])])
    want = np.array(circ.simulate(list(inputs)))
    np.save(tmp_path / "inputs.npy", inputs)
    for threshold in (1, 16, None):   # 1: every level is split (odd widths exercise the padding); 16: narrow levels stay
        # replicated; None: the default - levels re-packed for 2 x 256 per round, only levels wider than a round are split
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_exec_worker, args=(2, port, str(tmp_path), threshold), nprocs=2, join=True)
        for r in range(2):
            assert np.array_equal(np.load(tmp_path / f"out{r}.npy"), want), (threshold, r)
        sharded, total = np.load(tmp_path / "sharded0.npy")
        if threshold is None:
            assert sharded == 0      # no level of this small circuit needs more than one kernel round
        else:
            assert (sharded == total) if threshold == 1 else (0 < sharded < total)
