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


# ----------------------------------------------------------------------------------------------------------------
# The same executor on a circuit whose DEFAULT schedule shards (BASELINE config 4: levels up to 3,072 wide), on 2 and on 8
# ranks - the launch shape of the driver's scaling run - and the reference's one-call form `run()` on two ranks (rank 0 holds the
# secret keys: it encrypts and decrypts, ciphertexts and result travel by broadcast).  The stand-in engine is vectorised
# (one tensor operation per level and stage), so that a quarter of a million look-ups per rank run in seconds.
class _VecEngine(_PlainEngine):
    """the call surface of tfhe.Engine on plaintext rows (last word = 2 x message: delta_log 1), whole levels at a time"""
    q_bits = 65
    bsk_precision = 48

    class P(_PlainParams):
        N = 1024

    def __init__(self, secret=True):
        super().__init__()
        self.secret = secret
        self._tab = None

    def _tables(self):
        if self._tab is None or self._tab[0].shape[0] != len(self.tables):
            T = torch.zeros((len(self.tables), 32), dtype=torch.int64)
            P = torch.zeros(len(self.tables), dtype=torch.int64)
            for j, (p, t) in enumerate(self.tables):
                T[j, : len(t)] = torch.tensor(t, dtype=torch.int64)
                P[j] = p
            self._tab = (T, P)
        return self._tab

    def lincomb(self, store, rp, ix, cf, cs, count, out, stream=0):
        rp = rp[: count + 1].to(torch.int64)
        e0, e1 = int(rp[0]), int(rp[count])
        seg = torch.repeat_interleave(torch.arange(count), rp[1:] - rp[:-1])
        acc = torch.zeros((count, self.P.big), dtype=torch.int64)
        if e1 > e0:
            acc.index_add_(0, seg, store[ix[e0:e1].to(torch.int64)] * cf[e0:e1].to(torch.int64)[:, None])
        acc[:, -1] += cs[:count].to(torch.int64)
        out[:count] = acc

    def scatter_rows(self, src, count, store, rows, stream=0):
        store[rows[:count].to(torch.int64)] = src[:count]

    def pbs(self, d_in, ids, count, d_out, stream=0):
        T, P = self._tables()
        ids = ids[:count].to(torch.int64)
        x = d_in[:count, -1]
        assert not bool((x % 2 != 0).any())      # whole multiples of Delta reach a look-up
        p = P[ids]
        m = (x >> 1) >> (4 - p)
        half = torch.ones_like(p) << (p - 1)
        hi, lo = m >= half, m < -half            # the other half of the torus (Circuit.lut_odd inputs): negacyclic wrap-around
        idx = torch.where(hi, m - half, torch.where(lo, m + 3 * half, m + half))
        val = T[ids, idx]
        d_out[:count] = 0
        d_out[:count, -1] = torch.where(hi | lo, -val, val)

    # what EncryptedMatrixInversion.run() needs beyond the executor's calls
    def encrypt(self, flat, delta_log):
        if not self.secret:
            raise RuntimeError("evaluation-only context: it holds no secret key")
        ct = np.zeros((len(flat), self.P.big), np.uint64)
        ct[:, -1] = (2 * np.asarray(flat, np.int64)).view(np.uint64)
        return ct

    def decrypt(self, cts, delta_log):
        if not self.secret:
            raise RuntimeError("evaluation-only context: it holds no secret key")
        v = np.asarray(cts, np.uint64)[:, -1].view(np.int64)
        assert not (v % 2).any()
        return v // 2


def _config4_program():
    from bmi_amd.main import compile_inverse
    prog, _ = compile_inverse(4, 40, 16)
    return prog


def _wide_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from bmi_amd.executor import Executor
    prog = _config4_program()
    ex = Executor(prog, _VecEngine())                      # default sharding: levels re-packed for world x 256 per round
    inputs = np.load(os.path.join(out_dir, "inputs.npy"))
    cts = np.zeros((prog.n_inputs, 3), np.uint64)
    cts[:, -1] = (2 * inputs.astype(np.int64)).view(np.uint64)
    out = ex.run(cts)
    np.save(os.path.join(out_dir, f"out{rank}.npy"), out[:, -1].view(np.int64) // 2)
    widths = [w for w, *_ in ex.levels]
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([ex.sharded_levels, len(ex.levels), ex.world, max(widths), ex.shard_threshold]))
    dist.barrier()
    dist.destroy_process_group()


def _random_inputs(prog, seed):
    """a real matrix through the quantiser, so that the circuit's interval claims hold"""
    from bmi_amd.qfloat_matrix_inversion import float_matrix_to_qfloat_arrays
    M = np.random.default_rng(seed).normal(0, 100, (4, 4))
    q, s = float_matrix_to_qfloat_arrays(M, 40, 16, 2)
    return np.concatenate([np.asarray(q, np.int64).reshape(-1), np.asarray(s, np.int64)])


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_executor_on_a_circuit_whose_default_schedule_shards(tmp_path, world):
    """BASELINE config 4 (227 k look-ups, levels up to 3,072 wide before re-packing) under the DEFAULT sharding on 2 and 8 gloo
    ranks: some levels are split (uneven widths: the last rank's share is padded), the rest are replicated, every rank ends with
    the plaintext circuit's integers"""
    prog = _config4_program()
    inputs = _random_inputs(prog, 7)
    want = np.array(prog.simulate(inputs))
    np.save(tmp_path / "inputs.npy", inputs)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_wide_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"out{r}.npy"), want), r
    sharded, total, seen_world, widest, threshold = np.load(tmp_path / "meta0.npy")
    assert seen_world == world and 0 < sharded < total and widest >= threshold
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / f"meta{r}.npy"), np.load(tmp_path / "meta0.npy"))


def _run_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from bmi_amd.main import EncryptedMatrixInversion
    emi = EncryptedMatrixInversion(4, None, 2, 40, 16, False, False, engine=_VecEngine(secret=(rank == 0)))
    emi.error_budget = {}                      # (the stand-in engine has no noise: skip the budget of a real parameter set)
    M = np.load(os.path.join(out_dir, "M.npy"))
    inv = emi.run(M)
    np.save(os.path.join(out_dir, f"inv{rank}.npy"), inv)
    dist.barrier()
    dist.destroy_process_group()


def test_one_call_run_on_two_ranks(tmp_path):
    """EncryptedMatrixInversion.run() (reference main.py:93-116) with torch.distributed on two ranks: only rank 0 can encrypt and
    decrypt (the other rank's engine refuses, as an evaluation-only context does); both return the same inverse, equal to the
    single-process plaintext evaluation of the circuit"""
    from bmi_amd.main import EncryptedMatrixInversion
    M = np.random.default_rng(3).normal(0, 100, (4, 4))
    np.save(tmp_path / "M.npy", M)
    want = EncryptedMatrixInversion(4, None, 2, 40, 16, False, False, engine=_VecEngine()).run(M, simulate=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_run_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"inv{r}.npy"), want), r


def test_batched_executor_and_run_many_match_the_single_runs():
    """Executor(batch=B): B input vectors through one walk of the levels (every level B times wider, replica b on its own copy
    of the store) == B single runs; EncryptedMatrixInversion.run_many == run() matrix by matrix.  (Stand-in engine on plaintext
    rows; the GPU twin is tests/test_gpu_inverse.py::test_batched_inverses_on_the_gpu.)"""
    from bmi_amd.executor import Executor
    from bmi_amd.main import EncryptedMatrixInversion, compile_inverse
    prog, _ = compile_inverse(2, 20, 8)
    from bmi_amd.qfloat_matrix_inversion import float_matrix_to_qfloat_arrays
    rng = np.random.default_rng(11)
    B = 3
    inputs = []
    for _ in range(B):
        q, s = float_matrix_to_qfloat_arrays(rng.normal(0, 10, (2, 2)), 20, 8, 2)
        inputs.append(np.concatenate([np.asarray(q, np.int64).reshape(-1), np.asarray(s, np.int64)]))
    cts = np.zeros((B, prog.n_inputs, 3), np.uint64)
    for b in range(B):
        cts[b, :, -1] = (2 * inputs[b]).view(np.uint64)
    ex = Executor(prog, _VecEngine(), batch=B)
    one = Executor(prog, _VecEngine())
    # (the batched executor re-packs the levels for B replicas per throughput-kernel round: same depth, same look-ups)
    assert len(ex.levels) == len(one.levels) and sum(w for w, *_ in ex.levels) == B * sum(w for w, *_ in one.levels)
    assert all(w % B == 0 for w, *_ in ex.levels) and ex.n_rows % B == 0
    out = ex.run(cts)
    assert out.shape == (B, prog.n_outputs, 3)
    for b in range(B):
        want = np.array(prog.simulate(inputs[b]))
        assert np.array_equal(out[b, :, -1].view(np.int64) // 2, want), b
        assert np.array_equal(one.run(cts[b])[:, -1].view(np.int64) // 2, want), b
    assert not np.array_equal(out[0], out[1])
    emi = EncryptedMatrixInversion(2, None, 2, 20, 8, False, False, engine=_VecEngine())
    emi.error_budget = {}
    Ms = [rng.normal(0, 10, (2, 2)) for _ in range(4)]
    many = emi.run_many(Ms)
    for M, inv in zip(Ms, many):
        assert np.array_equal(inv, emi.run(M)) and np.allclose(inv @ M, np.eye(2), atol=0.1)


def _batched_exec_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmi_amd.executor import Executor
    circ = _trace_small()
    ex = Executor(circ, _PlainEngine(), shard_threshold=16, batch=3)
    inputs = np.load(os.path.join(out_dir, "inputs3.npy"))
    cts = np.zeros((3, circ.n_inputs, 3), np.uint64)
    cts[:, :, -1] = (2 * inputs.astype(np.int64)).view(np.uint64)
    out = ex.run(cts)
    np.save(os.path.join(out_dir, f"out3_{rank}.npy"), out[:, :, -1].view(np.int64) // 2)
    np.save(os.path.join(out_dir, f"sharded3_{rank}.npy"), np.array([ex.sharded_levels, len(ex.levels)]))
    dist.barrier()
    dist.destroy_process_group()


def test_batched_executor_on_two_ranks(tmp_path):
    """batch and sharding together: three input vectors per walk, the (three times wider) levels split across two gloo ranks"""
    circ = _trace_small()
    rng = np.random.default_rng(9)
    inputs = np.array([[rng.integers(lo, hi + 1) for lo, hi in zip(circ.leaf_lo[:circ.n_inputs], circ.leaf_hi[:circ.n_inputs])] for _ in range(3)])
    np.save(tmp_path / "inputs3.npy", inputs)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_batched_exec_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"out3_{r}.npy")
        for b in range(3):
            assert np.array_equal(got[b], np.array(circ.simulate(list(inputs[b])))), (r, b)
    sharded, total = np.load(tmp_path / "sharded3_0.npy")
    assert 0 < sharded <= total
