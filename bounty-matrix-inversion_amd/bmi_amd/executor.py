"""GPU executor: runs a compiled `program.Program` level by level through the C ABI.

Per level:  bmi_lincomb_batch (forms the level's PBS inputs from the device-resident leaf store) ->
bmi_pbs_batch (keyswitch -> blind rotation -> extraction, into a contiguous level buffer) -> bmi_scatter_rows (level
buffer -> the store rows assigned to the outputs).  All index arrays are built once, vectorised, from the program's CSR
arrays and stay on the device; a run is 3 launches per level plus the keyswitch's on one stream, with no host
synchronisation until the outputs are read back.  PyTorch is used only as the device allocator.

Store rows are recycled: a leaf's row returns to the free list at the level of its last consumer (liveness is known
at compile time), so the store holds the live set, not every look-up ever made (8x8 inverse: 2.6 M look-ups, a few
hundred thousand rows).

Several GPUs: every rank holds the same evaluation keys (Engine.keygen_shared, or seeded test keys; checked at construction) and the whole store and walks the same level
list.  A level at least `shard_threshold` wide is split into contiguous row ranges (shard.shard_range on a padded
width): each rank forms and bootstraps ITS rows only, one all-gather (RCCL) on the level buffer completes it
everywhere, then every rank scatters the whole level into its store; this is the path's only exchange step.  Narrower
levels are computed redundantly by every rank - cheaper than any transfer, since a level below ~256 ciphertexts
costs one latency-kernel round whatever its width.

Several INPUTS at once (`batch`): the same program evaluated on `batch` independent input vectors in one walk of the level list -
every level is `batch` times wider (replica b of a node reads replica b's rows: the store is `batch` stacked copies of the
one-input store), so the levels of one latency-kernel round become throughput-kernel launches: the serving form (a 3x3 inverse
is 319 rounds of <= 256 look-ups alone, 69 k look-ups at the throughput kernel's rate in a batch).
"""
from __future__ import annotations

import numpy as np

from .circuit import Circuit
from .program import Program
from .program import ROUND as Program_ROUND, WIDE_ROUND as Program_WIDE_ROUND


def _torus(v, delta_log, q):
    """signed integers -> v * 2^delta_log mod q, as int64 bit patterns"""
    out = np.empty(len(v), np.int64)
    for i, x in enumerate(v):
        t = (int(x) << delta_log) % q
        out[i] = t - (1 << 64) if t >= (1 << 63) else t
    return out


def assign_rows(prog: Program, recycle=True):
    """Store row of every leaf (inputs first).  With recycling, the rows of leaves whose last consumer is level t are
    handed to the outputs of level t (the level's inputs are copied out by its lincomb before its PBS writes).
    Returns (row per leaf, number of rows)."""
    n_in, nn = prog.n_inputs, prog.n_nodes
    n_leaves = n_in + nn
    row = np.empty(n_leaves, np.int64)
    row[:n_in] = np.arange(n_in)
    order, counts = prog.level_order()
    if not recycle:
        row[n_in + order] = n_in + np.arange(nn)
        return row, n_leaves
    depth = prog.depth
    # last level that reads each leaf: terms visited in level order, so the last write per leaf is the latest level
    lens = np.diff(prog.node_ptr)
    term_level = np.repeat(prog.node_level, lens)
    by_level = np.argsort(term_level, kind="stable")
    last_use = np.zeros(n_leaves, np.int64)
    last_use[prog.term_leaf[by_level]] = term_level[by_level]
    last_use[prog.out_leaf] = depth + 1          # outputs stay until the end
    dying = np.argsort(last_use, kind="stable")  # leaves grouped by the level that frees them
    dead_counts = np.bincount(last_use, minlength=depth + 2)
    dead_start = np.concatenate([[0], np.cumsum(dead_counts)])
    free = np.empty(n_leaves, np.int64)
    nfree, top, pos = 0, n_in, 0
    for t in range(1, depth + 1):
        d = dying[dead_start[t]: dead_start[t + 1]]
        if d.size:
            free[nfree: nfree + d.size] = row[d]
            nfree += d.size
        k = int(counts[t - 1])
        take = min(k, nfree)
        nodes = order[pos: pos + k]
        pos += k
        if take:
            row[n_in + nodes[:take]] = free[nfree - take: nfree]
            nfree -= take
        if k > take:
            row[n_in + nodes[take:]] = top + np.arange(k - take)
            top += k - take
    return row, top


class Executor:
    def __init__(self, circuit, engine, group=None, shard_threshold=None, recycle=True, batch=1):
        """circuit: a program.Program (or a circuit.Circuit, frozen here); group: a torch.distributed process group (None:
        the default group when initialised with more than one rank, otherwise single-GPU execution); shard_threshold:
        narrowest level that is split across the ranks - None (default) = every level wider than one latency-kernel
        round (257 ciphertexts), and the program's levels are then re-packed for `world` x 256 ciphertexts per round
        (Program.rescheduled): G GPUs working on one level are one machine with G x 256 workgroup slots; recycle: reuse
        store rows after a leaf's last consumer; batch: number of independent input vectors one run() evaluates (module docstring)."""
        import torch
        self.torch = torch
        prog = Program.from_circuit(circuit) if isinstance(circuit, Circuit) else circuit
        self.c = self.prog = prog
        self.eng = engine
        self.dev = engine.torch_device() if hasattr(engine, "torch_device") else torch.device("cuda", engine.device)
        self.on_gpu = self.dev.type == "cuda"
        self.dist, self.group, self.rank, self.world = None, group, 0, 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self.dist, self.rank, self.world = dist, dist.get_rank(group), dist.get_world_size(group)
            # every rank bootstraps its part of a split level with ITS evaluation keys: they must be one key set (seeded keygen on
            # every rank, or Engine.keygen_shared / EncryptedMatrixInversion.keygen).  Independent keys would gather ciphertexts
            # under different keys and decrypt to garbage without any error - refused here instead.
            if hasattr(engine, "eval_key_fingerprint"):
                fp = engine.eval_key_fingerprint()
                on_gpu = dist.get_backend(group) == "nccl"
                lo = torch.tensor([fp], dtype=torch.int64, device=self.dev if on_gpu else "cpu")
                hi = lo.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
                if int(lo.item()) != int(hi.item()):
                    raise RuntimeError("the ranks of this group hold different evaluation keys: generate one key set and share it "
                                       "(Engine.keygen_shared, or EncryptedMatrixInversion.keygen under torch.distributed)")
        if shard_threshold is None:
            shard_threshold = Program_ROUND + 1
            if self.world > 1:
                prog = prog.rescheduled(Program_ROUND * self.world, Program_WIDE_ROUND * self.world)
                self.c = self.prog = prog
        self.shard_threshold = max(int(shard_threshold), 1)
        if int(batch) > 1:
            # `batch` replicas of every level run side by side: a level is cheapest when replicas x width fills whole rounds of the
            # throughput kernel (1,024 ciphertexts per round and GPU), so the program's levels are re-packed for that width per
            # replica (same depth; 3x3 at 8 matrices per walk: 7 % fewer rounds, 2x2: 17 %)
            per = max(Program_WIDE_ROUND * self.world // int(batch), 1)
            prog = prog.rescheduled(per, per)
            self.c = self.prog = prog
        self.big = engine.P.big
        self.batch = B = int(batch)
        if B < 1:
            raise ValueError("batch must be at least 1")
        n_in = prog.n_inputs
        order, counts = prog.level_order()
        self.row_of, self.n_rows1 = assign_rows(prog, recycle)
        self.n_rows = self.n_rows1 * B
        self.delta_log = engine.delta_log(prog.msg_bits)
        q, dl = engine.modulus, self.delta_log
        # a lut_neg table (-+1) is registered at half the output scale: its ciphertexts carry (bit - 1/2) Delta, and every
        # consumer's constant below makes up for the missing Delta / 2 (Circuit.lut_neg, Program.half_unit_consts)
        lut_ids = np.asarray([engine.lut_register(prog.lut_tab[j, : 1 << int(prog.lut_p[j])].astype(np.int64),
                                                  int(prog.lut_p[j]), dl - 1 if prog.lut_half[j] else dl)
                              for j in range(prog.lut_p.size)], np.int32)
        to_dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.dev)  # noqa: E731

        # every level's rows in level order: one set of flat arrays for the whole program, sliced per level
        consts = prog.half_unit_consts(prog.node_ptr, prog.term_leaf, prog.term_coef, prog.node_const)   # units of Delta / 2
        shift = np.zeros(order.size, np.int64)      # store-row offset of a node's replica
        if B > 1:
            # replica b of every node right after replica b - 1 of its level: per level the node list tiled B times
            lvl_start = np.concatenate([[0], np.cumsum(counts)])
            tiled = [np.tile(order[lvl_start[t]: lvl_start[t + 1]], B) for t in range(len(counts))]
            shift = np.concatenate([np.repeat(np.arange(B, dtype=np.int64) * self.n_rows1, int(counts[t])) for t in range(len(counts))]) \
                if len(counts) else shift
            order = np.concatenate(tiled) if tiled else order
            counts = counts * B
        lens = np.diff(prog.node_ptr)[order]
        starts = prog.node_ptr[:-1][order]
        tot = int(lens.sum())
        off = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens) + np.repeat(starts, lens)
        self.d_idx = to_dev(self.row_of[prog.term_leaf[off]] + np.repeat(shift, lens) if tot else [0], np.int32)
        self.d_coef = to_dev(prog.term_coef[off] if tot else [0], np.int64)
        consts = consts[order]
        uniq, inv = np.unique(consts, return_inverse=True)
        self.d_const = to_dev(_torus(uniq, dl - 1, q)[inv] if consts.size else [0], np.int64)
        self.d_ids = to_dev(lut_ids[prog.node_lut[order]] if order.size else [0], np.int32)
        self.d_rows = to_dev(self.row_of[n_in + order] + shift if order.size else [0], np.int32)
        # row_ptr of all nodes in level order, absolute term offsets: a level (or a rank's part of it) is a slice
        self.d_rp = to_dev(np.concatenate([[0], np.cumsum(lens)]), np.int32)
        self.levels = []          # (width, position of the level's first node, padded width)
        pos = 0
        for w in counts.tolist():
            padded = -(-w // self.world) * self.world if (self.world > 1 and w >= self.shard_threshold) else w
            self.levels.append((w, pos, padded))
            pos += w
        o_len = np.diff(prog.out_ptr)
        o_rows = self.row_of[prog.out_leaf] if prog.out_leaf.size else np.zeros(0, np.int64)
        o_const = _torus(prog.half_unit_consts(prog.out_ptr, prog.out_leaf, prog.out_coef, prog.out_const), dl - 1, q)
        self.out_csr = (to_dev(np.concatenate([[0], np.cumsum(np.tile(o_len, B))]), np.int32),
                        to_dev(np.concatenate([o_rows + b * self.n_rows1 for b in range(B)]) if o_rows.size else [0], np.int32),
                        to_dev(np.tile(prog.out_coef, B) if prog.out_coef.size else [0], np.int64),
                        to_dev(np.tile(o_const, B), np.int64))
        self.n_out = prog.n_outputs * B
        self.max_width = max((p for *_, p in self.levels), default=1)
        self.sharded_levels = sum(1 for w, *_ in self.levels if self.world > 1 and w >= self.shard_threshold)
        self.store = torch.zeros((max(self.n_rows, 1), self.big), dtype=torch.int64, device=self.dev)
        self.tmp = torch.zeros((max(self.max_width, 1), self.big), dtype=torch.int64, device=self.dev)
        self.lvl = torch.zeros((max(self.max_width, 1), self.big), dtype=torch.int64, device=self.dev)
        self.out = torch.zeros((max(self.n_out, 1), self.big), dtype=torch.int64, device=self.dev)
        engine.reserve(self.max_width)

    def store_bytes(self):
        return int(self.store.numel() + self.tmp.numel() + self.lvl.numel()) * 8

    def run(self, ct_inputs):
        """ct_inputs: (n_inputs, k*N+1) uint64 ciphertexts (host) -> (n_outputs, k*N+1) uint64 (host); with batch > 1:
        (batch, n_inputs, k*N+1) -> (batch, n_outputs, k*N+1)"""
        torch = self.torch
        n_in = self.prog.n_inputs
        ct = np.ascontiguousarray(ct_inputs, dtype=np.uint64).reshape(self.batch, n_in, self.big)
        import contextlib
        with (torch.cuda.device(self.dev) if self.on_gpu else contextlib.nullcontext()):
            stream = torch.cuda.current_stream().cuda_stream if self.on_gpu else 0
            self.store.view(self.batch, -1, self.big)[:, :n_in].copy_(torch.from_numpy(ct.view(np.int64)), non_blocking=False)
            eng, tmp, lvl = self.eng, self.tmp, self.lvl
            for width, pos, padded in self.levels:
                lo, hi = 0, width
                sharded = self.world > 1 and width >= self.shard_threshold
                if sharded:
                    per = padded // self.world
                    lo = min(self.rank * per, width)
                    hi = min(lo + per, width)
                if hi > lo:
                    eng.lincomb(self.store, self.d_rp[pos + lo:], self.d_idx, self.d_coef, self.d_const[pos + lo:], hi - lo, tmp[lo:], stream)
                    eng.pbs(tmp[lo:], self.d_ids[pos + lo:], hi - lo, lvl[lo:], stream)
                if sharded:
                    per = padded // self.world
                    pad_from = max(hi, self.rank * per)       # this rank's gather region is [rank * per, (rank + 1) * per)
                    if pad_from < (self.rank + 1) * per:      # rows past the level's width: they travel as zeros, never scattered
                        lvl[pad_from: (self.rank + 1) * per].zero_()
                    self._all_gather_rows(lvl[:padded], padded // self.world)
                eng.scatter_rows(lvl, width, self.store, self.d_rows[pos:], stream)
            rp, ix, cf, cs = self.out_csr
            eng.lincomb(self.store, rp, ix, cf, cs, self.n_out, self.out, stream)
            if self.on_gpu:
                torch.cuda.synchronize(self.dev)
            res = self.out[: self.n_out].cpu().numpy().view(np.uint64)
            return res if self.batch == 1 else res.reshape(self.batch, -1, self.big)

    def _all_gather_rows(self, region, per):
        """in-place all-gather of a level buffer: rank r contributed rows [r*per, (r+1)*per) (rows of the last rank past
        the level's width are padding: gathered, never scattered)"""
        dist, torch = self.dist, self.torch
        mine = region[self.rank * per: (self.rank + 1) * per]
        if dist.get_backend(self.group) == "nccl":
            # RCCL, in place (rank r's part already sits at its offset), ordered after this level's kernels: they were
            # launched on torch's current stream, which the collective synchronises with.  Exercised by
            # tests/test_gpu_inverse.py::test_two_gpu_rccl_sharded_inverse, which needs two GPUs (the build's test boxes
            # have one: there the gloo branch below runs).
            dist.all_gather_into_tensor(region, mine, group=self.group)
            return
        # other backends (gloo rehearsal): stage through the host
        if self.on_gpu:
            torch.cuda.synchronize(self.dev)
        parts = [torch.empty((per, self.big), dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(parts, mine.cpu().contiguous(), group=self.group)
        for r, part in enumerate(parts):
            if r != self.rank:
                region[r * per: (r + 1) * per].copy_(part)
