"""GPU executor: runs a traced `circuit.Circuit` level by level through the C ABI.

Per ASAP level:  bmi_lincomb_batch (forms every PBS input of the level from the device-resident leaf store)
followed by bmi_pbs_batch (keyswitch -> blind rotation -> extraction) writing the level's outputs straight
into the store.  All index arrays are built once (at `compile`, the analogue of the reference's
`compiler.compile`, main.py:66) and stay on the device; a run is ~2 launches per level on one stream with
no host synchronisation until the outputs are read back.  PyTorch is used only as the device allocator.

Several GPUs: every rank holds the keys (seeded keygen) and the whole leaf store and walks the
same level list.  A level at least `shard_threshold` wide is split into contiguous row ranges (shard.shard_range
on a padded width): each rank bootstraps its range in place and one all-gather (RCCL, in place on the level's
store region) completes the region everywhere; this is the path's only exchange step.  Narrower levels are
computed redundantly by every rank - cheaper than any transfer, since a level below ~256 ciphertexts costs one
latency-kernel round whatever its width.
"""
from __future__ import annotations

import numpy as np

from .circuit import MSG_BITS


def _torus(v, delta_log, q):
    """signed integer -> v * 2^delta_log mod q, as an int64 bit pattern"""
    t = (int(v) << delta_log) % q
    return t - (1 << 64) if t >= (1 << 63) else t


class Executor:
    def __init__(self, circuit, engine, group=None, shard_threshold=1024):
        """group: a torch.distributed process group (None: the default group when initialised with more than one
        rank, otherwise single-GPU execution); shard_threshold: narrowest level that is split across the ranks."""
        import torch
        self.torch = torch
        self.c, self.eng = circuit, engine
        self.dev = engine.torch_device() if hasattr(engine, "torch_device") else torch.device("cuda", engine.device)
        self.on_gpu = self.dev.type == "cuda"
        self.dist, self.group, self.rank, self.world = None, group, 0, 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            self.dist, self.rank, self.world = dist, dist.get_rank(group), dist.get_world_size(group)
        self.shard_threshold = max(int(shard_threshold), 1)
        P = engine.P
        self.big = P.big
        levels = circuit.levels()
        n_in = circuit.n_inputs
        # renumber leaves so that every level's outputs are contiguous rows of the store; a sharded level's region
        # is padded to a multiple of the world size so that the in-place all-gather has equal parts
        new_id = {i: i for i in range(n_in)}
        nxt = n_in
        self.level_rows = []
        for lv in levels:
            for j, ni in enumerate(lv):
                new_id[circuit.nodes[ni][3]] = nxt + j
            rows = len(lv)
            if self.world > 1 and rows >= self.shard_threshold:
                rows = -(-rows // self.world) * self.world
            self.level_rows.append(rows)
            nxt += rows
        self.n_leaves = nxt
        self.delta_log = engine.delta_log(getattr(circuit, "msg_bits", MSG_BITS))
        q, dl = engine.modulus, self.delta_log
        lut_ids = [engine.lut_register(np.array(tab, dtype=np.int64), p, dl) for p, tab in circuit.luts]

        def csr(rows):
            rp, ix, cf, cs = [0], [], [], []
            for terms, const in rows:
                for leaf, coef in terms:
                    ix.append(new_id[leaf])
                    cf.append(coef)
                rp.append(len(ix))
                cs.append(_torus(const, dl, q))
            t = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(self.dev)  # noqa: E731
            return (t(rp, np.int32), t(ix if ix else [0], np.int32), t(cf if cf else [0], np.int64), t(cs, np.int64))

        self.levels = []
        base = n_in
        for lv, padded in zip(levels, self.level_rows):
            rows = [(circuit.nodes[ni][0], circuit.nodes[ni][1]) for ni in lv]
            ids = torch.from_numpy(np.asarray([lut_ids[circuit.nodes[ni][2]] for ni in lv], dtype=np.int32)).to(self.dev)
            self.levels.append((len(lv), base, csr(rows), ids, padded))
            base += padded
        self.out_csr = csr(circuit.outputs)
        self.n_out = len(circuit.outputs)
        self.max_width = max((w for w, *_ in self.levels), default=1)
        self.sharded_levels = sum(1 for w, _, _, _, padded in self.levels if self.world > 1 and w >= self.shard_threshold)
        self.store = torch.zeros((self.n_leaves, self.big), dtype=torch.int64, device=self.dev)
        self.tmp = torch.zeros((max(self.max_width, 1), self.big), dtype=torch.int64, device=self.dev)
        self.out = torch.zeros((max(self.n_out, 1), self.big), dtype=torch.int64, device=self.dev)
        engine.reserve(self.max_width)

    def run(self, ct_inputs):
        """ct_inputs: (n_inputs, k*N+1) uint64 ciphertexts (host) -> (n_outputs, k*N+1) uint64 (host)"""
        torch = self.torch
        ct = np.ascontiguousarray(ct_inputs, dtype=np.uint64).reshape(self.c.n_inputs, self.big)
        import contextlib
        with (torch.cuda.device(self.dev) if self.on_gpu else contextlib.nullcontext()):
            stream = torch.cuda.current_stream().cuda_stream if self.on_gpu else 0
            self.store[: self.c.n_inputs].copy_(torch.from_numpy(ct.view(np.int64)), non_blocking=False)
            from .shard import shard_range
            for width, base, (rp, ix, cf, cs), ids, padded in self.levels:
                # every rank forms all of the level's PBS inputs (a few hundred bytes of index data per row)
                self.eng.lincomb(self.store, rp, ix, cf, cs, width, self.tmp, stream)
                if self.world > 1 and width >= self.shard_threshold:
                    per = padded // self.world
                    lo = min(self.rank * per, width)
                    hi = min(lo + per, width)
                    if hi > lo:
                        self.eng.pbs(self.tmp[lo:hi], ids[lo:hi], hi - lo, self.store[base + lo: base + hi], stream)
                    self._all_gather_rows(self.store[base: base + padded], per)
                else:
                    self.eng.pbs(self.tmp, ids, width, self.store[base: base + width], stream)
            rp, ix, cf, cs = self.out_csr
            self.eng.lincomb(self.store, rp, ix, cf, cs, self.n_out, self.out, stream)
            if self.on_gpu:
                torch.cuda.synchronize(self.dev)
            return self.out[: self.n_out].cpu().numpy().view(np.uint64)

    def _all_gather_rows(self, region, per):
        """in-place all-gather of a level's store region: rank r contributed rows [r*per, (r+1)*per)"""
        dist, torch = self.dist, self.torch
        mine = region[self.rank * per: (self.rank + 1) * per]
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(region, mine, group=self.group)   # RCCL, in place, on the compute stream order
            return
        # other backends (gloo rehearsal): stage through the host
        if self.on_gpu:
            torch.cuda.synchronize(self.dev)
        parts = [torch.empty((per, self.big), dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(parts, mine.cpu().contiguous(), group=self.group)
        for r, part in enumerate(parts):
            if r != self.rank:
                region[r * per: (r + 1) * per].copy_(part)
