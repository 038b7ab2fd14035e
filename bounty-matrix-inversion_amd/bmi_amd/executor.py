"""GPU executor: runs a traced `circuit.Circuit` level by level through the C ABI.

Per ASAP level:  bmi_lincomb_batch (forms every PBS input of the level from the device-resident leaf store)
followed by bmi_pbs_batch (keyswitch -> blind rotation -> extraction) writing the level's outputs straight
into the store.  All index arrays are built once (at `compile`, the analogue of the reference's
`compiler.compile`, main.py:66) and stay on the device; a run is ~2 launches per level on one stream with
no host synchronisation until the outputs are read back.  PyTorch is used only as the device allocator.
"""
from __future__ import annotations

import numpy as np

from .circuit import MSG_BITS


def _torus(v, delta_log, q):
    """signed integer -> v * 2^delta_log mod q, as an int64 bit pattern"""
    t = (int(v) << delta_log) % q
    return t - (1 << 64) if t >= (1 << 63) else t


class Executor:
    def __init__(self, circuit, engine):
        import torch
        self.torch = torch
        self.c, self.eng = circuit, engine
        self.dev = torch.device("cuda", engine.device)
        P = engine.P
        self.big = P.big
        levels = circuit.levels()
        n_in = circuit.n_inputs
        # renumber leaves so that every level's outputs are contiguous rows of the store
        new_id = {i: i for i in range(n_in)}
        nxt = n_in
        for lv in levels:
            for ni in lv:
                new_id[circuit.nodes[ni][3]] = nxt
                nxt += 1
        self.n_leaves = nxt
        self.delta_log = engine.delta_log(MSG_BITS)
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
        for lv in levels:
            rows = [(circuit.nodes[ni][0], circuit.nodes[ni][1]) for ni in lv]
            ids = torch.from_numpy(np.asarray([lut_ids[circuit.nodes[ni][2]] for ni in lv], dtype=np.int32)).to(self.dev)
            self.levels.append((len(lv), base, csr(rows), ids))
            base += len(lv)
        self.out_csr = csr(circuit.outputs)
        self.n_out = len(circuit.outputs)
        self.max_width = max((w for w, *_ in self.levels), default=1)
        self.store = torch.zeros((self.n_leaves, self.big), dtype=torch.int64, device=self.dev)
        self.tmp = torch.zeros((max(self.max_width, 1), self.big), dtype=torch.int64, device=self.dev)
        self.out = torch.zeros((max(self.n_out, 1), self.big), dtype=torch.int64, device=self.dev)
        engine.reserve(self.max_width)

    def run(self, ct_inputs):
        """ct_inputs: (n_inputs, k*N+1) uint64 ciphertexts (host) -> (n_outputs, k*N+1) uint64 (host)"""
        torch = self.torch
        ct = np.ascontiguousarray(ct_inputs, dtype=np.uint64).reshape(self.c.n_inputs, self.big)
        with torch.cuda.device(self.dev):
            stream = torch.cuda.current_stream().cuda_stream
            self.store[: self.c.n_inputs].copy_(torch.from_numpy(ct.view(np.int64)), non_blocking=False)
            row_bytes = self.big * 8
            sp = self.store.data_ptr()
            for width, base, (rp, ix, cf, cs), ids in self.levels:
                self.eng.lincomb(self.store, rp, ix, cf, cs, width, self.tmp, stream)
                self.eng.pbs(self.tmp, ids, width, sp + base * row_bytes, stream)
            rp, ix, cf, cs = self.out_csr
            self.eng.lincomb(self.store, rp, ix, cf, cs, self.n_out, self.out, stream)
            torch.cuda.synchronize(self.dev)
            return self.out[: self.n_out].cpu().numpy().view(np.uint64)
