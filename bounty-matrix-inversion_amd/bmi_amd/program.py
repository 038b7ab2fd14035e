"""Program: a traced circuit frozen into flat arrays — what `compile` produces, what the executor runs and what the
on-disk cache stores.

The reference pays for compilation on every start (`fhe.Compiler(...).compile`, matrix_inversion/main.py:53-66; its
README counts those seconds in the published totals) and softens it with Concrete's key cache
(qfloat_matrix_inversion.py:997-998).  Here the traced PBS graph of a configuration is a pure function of
(n, len, ints, base, true_division, tensorize), so it is traced once (`circuit.Circuit`, Python), then

  * pruned: look-ups no output depends on are dropped (dead-code elimination),
  * scheduled: every look-up gets its level (same depth as ASAP, width-aware: `schedule_levels`),
  * frozen into CSR arrays (terms of every PBS input, tables, outputs, interval claims),
  * written to `<cache dir>/<key>.npz`; every later start is a load of that file (< 1 s even for the 8x8 inverse).

`Program.simulate` is the vectorised plaintext evaluator (the analogue of `circuit.simulate`, main.py:107): one numpy
pass per level instead of one Python step per look-up; it checks the same interval claims as `Circuit.simulate`.
"""
from __future__ import annotations

import hashlib
import heapq
import os
import time

import numpy as np

from .circuit import Circuit, RangeError

FORMAT = 4
ROUND = Circuit.ROUND
WIDE_ROUND = Circuit.WIDE_ROUND


def cache_dir():
    d = os.environ.get("BMI_CACHE_DIR")
    if not d:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cache")
    return d


def shipped_dir():
    """read-only programs that travel with the package (compact form, Program.save_compact): the BASELINE configurations whose
    trace takes minutes (8 x 8, 10 x 10).  A file is only used when its name carries the current tracer fingerprint."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "programs")


def _tracer_fingerprint():
    """hash of the modules that define what a trace looks like: a cached program is only valid for the code that made it"""
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    # inverse_circuit.py too: it declares the inputs (order, intervals), picks the message width and the division radix of a
    # configuration (the API wrapper, main.py, is NOT part of it)
    for name in ("circuit.py", "base_p_arrays.py", "qfloat.py", "qfloat_matrix_inversion.py", "program.py", "inverse_circuit.py"):
        with open(os.path.join(here, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def schedule_levels(n_in, node_ptr, term_leaf, asap, round_=ROUND, wide_round=WIDE_ROUND):
    """Width-aware list schedule (same depth as ASAP).  A level costs one latency-kernel round per started `round_`
    ciphertexts up to two rounds and one throughput-kernel round per started `wide_round` beyond that, so nodes with
    slack (ALAP later than ASAP) are moved out of levels that would otherwise spill into one more round: levels are
    filled in order, critical nodes (ALAP = this level) first, then ready nodes by ALAP while the rounds the critical
    ones need anyway have room.  Returns the level (1-based) of every node."""
    nn = len(asap)
    if nn == 0:
        return np.zeros(0, np.int32)
    ptr = node_ptr.tolist()
    tl = term_leaf.tolist()
    asap_l = asap.tolist()
    depth = max(asap_l)
    preds = [[t - n_in for t in tl[ptr[i]:ptr[i + 1]] if t >= n_in] for i in range(nn)]
    succs = [[] for _ in range(nn)]
    for i, ps in enumerate(preds):
        for q in ps:
            succs[q].append(i)
    alap = [depth] * nn
    for i in np.argsort(-asap, kind="stable").tolist():   # successors (larger ASAP) are final before their producers
        a = alap[i]
        for sc in succs[i]:
            if alap[sc] - 1 < a:
                a = alap[sc] - 1
        alap[i] = a
    indeg = [len(ps) for ps in preds]
    bucket, keys = {}, []
    for i in range(nn):
        if indeg[i] == 0:
            b = bucket.get(alap[i])
            if b is None:
                b = bucket[alap[i]] = []
                heapq.heappush(keys, alap[i])
            b.append(i)
    level = [0] * nn
    done, t = 0, 0
    while done < nn:
        t += 1
        must = len(bucket.get(t, ()))
        cap = max(1, -(-must // round_)) * round_ if must <= 2 * round_ else -(-must // wide_round) * wide_round
        chosen = []
        while keys and len(chosen) < cap:
            a = keys[0]
            b = bucket[a]
            room = cap - len(chosen)
            if len(b) <= room:
                chosen.extend(b)
                del bucket[a]
                heapq.heappop(keys)
            else:
                chosen.extend(b[-room:])
                del b[-room:]
        for i in chosen:
            level[i] = t
            for sc in succs[i]:
                indeg[sc] -= 1
                if indeg[sc] == 0:
                    a = alap[sc]
                    b = bucket.get(a)
                    if b is None:
                        b = bucket[a] = []
                        heapq.heappush(keys, a)
                    b.append(sc)
        done += len(chosen)
    assert t == depth, (t, depth)
    return np.asarray(level, np.int32)


def prune_nodes(n_in, node_ptr, term_leaf, out_leaf):
    """live mask of the look-ups an output depends on (Python twin of bmi_circuit_prune)"""
    nn = node_ptr.size - 1
    live = [False] * nn
    for t in out_leaf.tolist():
        if t >= n_in:
            live[t - n_in] = True
    ptr, tl = node_ptr.tolist(), term_leaf.tolist()
    for i in range(nn - 1, -1, -1):     # creation order is topological: consumers come after producers
        if live[i]:
            for t in tl[ptr[i]:ptr[i + 1]]:
                if t >= n_in:
                    live[t - n_in] = True
    return np.asarray(live, bool)


class Program:
    ARRAYS = ("in_lo", "in_hi", "node_ptr", "term_leaf", "term_coef", "node_const", "node_lut", "node_level",
              "node_wide", "node_half", "node_lo", "node_hi", "lut_p", "lut_tab", "lut_half", "out_ptr", "out_leaf", "out_coef", "out_const",
              "claim_ptr", "claim_leaf", "claim_coef", "claim_const", "claim_lo", "claim_hi")

    def __init__(self, msg_bits, arrays, meta=None):
        self.msg_bits = int(msg_bits)
        for k in self.ARRAYS:
            setattr(self, k, arrays[k])
        self.meta = dict(meta or {})
        self.n_inputs = int(self.in_lo.size)
        self.n_nodes = int(self.node_lut.size)
        self.depth = int(self.node_level.max()) if self.n_nodes else 0
        self._order = None

    # ---- what the host layer reads off a circuit ----------------------------------------------------------------
    @property
    def leaf_lo(self):
        return self.in_lo

    @property
    def leaf_hi(self):
        return self.in_hi

    @property
    def n_outputs(self):
        return int(self.out_const.size)

    def half_unit_consts(self, ptr, leaf, coef, const):
        """Constants of the linear combinations (ptr, leaf, coef, const) in units of Delta / 2: a lut_neg leaf's ciphertext
        carries (bit - 1/2) Delta, so a consumer reading it with coefficient c owes c Delta / 2 (Circuit.lut_neg)."""
        ln = np.diff(ptr)
        seg = np.repeat(np.arange(ln.size), ln)
        is_half = np.zeros(self.n_inputs + self.n_nodes, bool)
        is_half[self.n_inputs:] = self.node_half
        owed = np.bincount(seg, weights=np.asarray(coef, np.int64) * is_half[leaf], minlength=ln.size).astype(np.int64)
        return 2 * np.asarray(const, np.int64) + owed

    def level_order(self):
        """(order, counts): node indices sorted by level (stable), number of nodes per level 1..depth"""
        if self._order is None:
            order = np.argsort(self.node_level, kind="stable")
            counts = np.bincount(self.node_level, minlength=self.depth + 1)[1:]
            self._order = (order, counts)
        return self._order

    def level_widths(self):
        return self.level_order()[1]

    def summary(self):
        w = self.level_widths()
        return {"inputs": self.n_inputs, "pbs": self.n_nodes, "depth": self.depth, "luts": int(self.lut_p.size),
                "max_width": int(w.max()) if w.size else 0, "mean_width": float(w.mean()) if w.size else 0.0,
                "median_width": float(np.median(w)) if w.size else 0.0, **{k: v for k, v in self.meta.items()
                                                                           if isinstance(v, (int, float, str))}}

    def failure_probability(self, params, bsk_precision=None, unroll=False, hw_small=None, hw_big=None):
        """Probability that an encrypted evaluation under the parameter set `params` (tfhe.Params, or an Engine) decodes a wrong
        integer somewhere because of noise: the sum over the look-ups of the Gaussian tail at each look-up's own margin (its
        linear combination's amplification included), plus the outputs' decryption tails (error_budget.py).  What Concrete's
        `p_error` / `global_p_error` bound for the reference (main.py:53-66)."""
        from . import error_budget
        if hasattr(params, "P") and hasattr(params, "bsk_precision"):        # an Engine
            eng = params
            return error_budget.failure_probability(self, eng.P, eng.bsk_precision if eng.q_bits == 65 else None,
                                                    getattr(eng, "unroll", 1) == 2, hw_small, hw_big)
        return error_budget.failure_probability(self, params, bsk_precision, unroll, hw_small, hw_big)

    # ---- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def from_circuit(cls, c: Circuit, meta=None, prune=True, native=True):
        """Freezes a traced circuit: flatten -> prune -> renumber -> schedule.  native=True runs the two graph passes in
        the library (bmi_circuit_prune / bmi_circuit_schedule, C++); native=False the Python twins in this module."""
        from itertools import chain
        n_in, nn = c.n_inputs, len(c.nodes)
        n_leaves = n_in + nn
        if nn and (c.nodes[0][3] != n_in or c.nodes[-1][3] != n_leaves - 1):
            raise ValueError("circuit leaves are not numbered inputs-then-nodes")

        def flatten(rows):
            """rows of (terms, const) -> CSR (ptr, leaf, coef, const)"""
            lens = np.fromiter((len(r[0]) for r in rows), np.int64, len(rows))
            flat = np.fromiter(chain.from_iterable(chain.from_iterable(r[0] for r in rows)), np.int64, 2 * int(lens.sum()))
            const = np.fromiter((r[1] for r in rows), np.int64, len(rows))
            return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), flat[0::2].astype(np.int32), flat[1::2].copy(), const

        node_ptr, term_leaf, term_coef, node_const = flatten(c.nodes)
        node_lut = np.fromiter((r[2] for r in c.nodes), np.int64, nn)
        out_ptr, out_leaf, out_coef, out_const = flatten(c.outputs)
        claim_ptr, claim_leaf, claim_coef, claim_const = flatten([cl[0] for cl in c.claims])
        claim_lo = np.fromiter((cl[1] for cl in c.claims), np.int64, len(c.claims))
        claim_hi = np.fromiter((cl[2] for cl in c.claims), np.int64, len(c.claims))
        if not prune:
            live = np.ones(nn, bool)
        elif native:
            from . import tfhe
            live = tfhe.circuit_prune(n_in, node_ptr, term_leaf, out_leaf)
        else:
            live = prune_nodes(n_in, node_ptr, term_leaf, out_leaf)
        keep = np.flatnonzero(live)
        new_leaf = np.full(n_leaves, -1, np.int64)
        new_leaf[:n_in] = np.arange(n_in)
        new_leaf[n_in + keep] = n_in + np.arange(keep.size)

        def take_rows(ptr, leaf, coef, rows):
            """the CSR rows `rows`, leaves renumbered"""
            lens = np.diff(ptr)[rows]
            tot = int(lens.sum())
            off = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens) + np.repeat(ptr[:-1][rows], lens)
            return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), new_leaf[leaf[off]].astype(np.int32), coef[off]

        k_ptr, k_leaf, k_coef = take_rows(node_ptr, term_leaf, term_coef, keep)
        o_ptr, o_leaf, o_coef = take_rows(out_ptr, out_leaf, out_coef, np.arange(out_const.size))
        # a claim about a value nothing depends on goes with it
        claim_lens = np.diff(claim_ptr)
        dead_terms = np.bincount(np.repeat(np.arange(claim_const.size), claim_lens),
                                 weights=(new_leaf[claim_leaf] < 0), minlength=claim_const.size)
        claim_rows = np.flatnonzero(dead_terms == 0)
        c_ptr, c_leaf, c_coef = take_rows(claim_ptr, claim_leaf, claim_coef, claim_rows)
        if native:
            from . import tfhe
            node_level = tfhe.circuit_schedule(n_in, k_ptr, k_leaf, ROUND, WIDE_ROUND)
        else:
            asap = np.asarray(c.leaf_level, np.int32)[n_in + keep]
            node_level = schedule_levels(n_in, k_ptr, k_leaf, asap)
        leaf_lo, leaf_hi = np.asarray(c.leaf_lo, np.int64), np.asarray(c.leaf_hi, np.int64)
        wide = np.zeros(n_leaves, bool)
        if c.wide_leaves:
            wide[np.fromiter(c.wide_leaves, np.int64, len(c.wide_leaves))] = True
        halfl = np.zeros(n_leaves, bool)        # Circuit.lut_neg leaves: the ciphertext carries (bit - 1/2) * Delta
        if getattr(c, "half_leaves", None):
            halfl[np.fromiter(c.half_leaves, np.int64, len(c.half_leaves))] = True
        width = 1 << c.msg_bits
        lut_tab = np.zeros((max(len(c.luts), 1), width), np.int16)
        for j, (p, tab) in enumerate(c.luts):
            lut_tab[j, : len(tab)] = tab
        k_lut = node_lut[keep]
        used_luts = np.unique(k_lut)
        lut_map = np.full(max(len(c.luts), 1), -1, np.int64)
        lut_map[used_luts] = np.arange(used_luts.size)
        arrays = dict(
            in_lo=leaf_lo[:n_in].copy(), in_hi=leaf_hi[:n_in].copy(), node_ptr=k_ptr, term_leaf=k_leaf,
            term_coef=k_coef.astype(np.int32), node_const=node_const[keep], node_lut=lut_map[k_lut].astype(np.int32),
            node_level=node_level, node_wide=wide[n_in + keep], node_half=halfl[n_in + keep],
            lut_half=np.asarray([j in getattr(c, "half_luts", ()) for j in used_luts.tolist()], bool),
            node_lo=leaf_lo[n_in + keep].astype(np.int16), node_hi=leaf_hi[n_in + keep].astype(np.int16),
            lut_p=np.asarray([c.luts[j][0] for j in used_luts.tolist()], np.int8), lut_tab=lut_tab[used_luts],
            out_ptr=o_ptr, out_leaf=o_leaf, out_coef=o_coef, out_const=out_const,
            claim_ptr=c_ptr, claim_leaf=c_leaf, claim_coef=c_coef, claim_const=claim_const[claim_rows],
            claim_lo=claim_lo[claim_rows], claim_hi=claim_hi[claim_rows])
        m = dict(meta or {})
        m.update(traced_pbs=nn, pruned_pbs=int(nn - keep.size), cse_hits=int(c.stats.get("cse_hits", 0)),
                 const_folds=int(c.stats.get("const_folds", 0)))
        return cls(c.msg_bits, arrays, m)

    def rescheduled(self, round_, wide_round):
        """The same program with its levels re-packed for another round capacity (G GPUs sharing every level run
        G x 256 ciphertexts per latency-kernel round): one call of the C++ scheduling pass, no re-tracing."""
        from . import tfhe
        arrays = {k: getattr(self, k) for k in self.ARRAYS}
        arrays["node_level"] = tfhe.circuit_schedule(self.n_inputs, self.node_ptr, self.term_leaf, round_, wide_round)
        return Program(self.msg_bits, arrays, dict(self.meta, round=int(round_), wide_round=int(wide_round)))

    # ---- disk ---------------------------------------------------------------------------------------------------------
    def save(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        meta_keys = sorted(self.meta)
        np.savez_compressed(tmp, __format=np.int64(FORMAT), __msg_bits=np.int64(self.msg_bits),
                            __meta_keys=np.asarray(meta_keys), __meta_vals=np.asarray([repr(self.meta[k]) for k in meta_keys]),
                            **{k: getattr(self, k) for k in self.ARRAYS})
        os.replace(tmp, path)     # atomic: several ranks may compile the same configuration at once

    # compact form (shipped programs): index arrays as differences, then LZMA - 16 MB of a zlib .npz become ~3 MB
    _DELTA = ("node_ptr", "out_ptr", "claim_ptr")

    def save_compact(self, path):
        import io
        import lzma
        os.makedirs(os.path.dirname(path), exist_ok=True)
        arrays = {k: getattr(self, k) for k in self.ARRAYS}
        for k in self._DELTA:
            arrays[k] = np.diff(arrays[k], prepend=0).astype(np.int32)
        # a term's leaf as its distance back from the node that reads it (small and repetitive), same for claims / outputs
        n_in = self.n_inputs
        owner = n_in + np.repeat(np.arange(self.n_nodes, dtype=np.int64), np.diff(self.node_ptr))
        arrays["term_leaf"] = (owner - self.term_leaf).astype(np.int32)
        arrays["claim_leaf"] = np.diff(self.claim_leaf.astype(np.int64), prepend=0).astype(np.int32)
        meta_keys = sorted(self.meta)
        buf = io.BytesIO()
        np.savez(buf, __format=np.int64(FORMAT), __msg_bits=np.int64(self.msg_bits), __meta_keys=np.asarray(meta_keys),
                 __meta_vals=np.asarray([repr(self.meta[k]) for k in meta_keys]), **arrays)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(lzma.compress(buf.getvalue(), preset=6))
        os.replace(tmp, path)

    @classmethod
    def load_compact(cls, path):
        import ast
        import io
        import lzma
        with open(path, "rb") as f:
            z = np.load(io.BytesIO(lzma.decompress(f.read())), allow_pickle=False)
        if int(z["__format"]) != FORMAT:
            raise ValueError("shipped program has another format")
        arrays = {k: z[k] for k in cls.ARRAYS}
        for k in cls._DELTA:
            arrays[k] = np.cumsum(arrays[k].astype(np.int64))
        n_in = int(arrays["in_lo"].size)
        owner = n_in + np.repeat(np.arange(arrays["node_lut"].size, dtype=np.int64), np.diff(arrays["node_ptr"]))
        arrays["term_leaf"] = (owner - arrays["term_leaf"]).astype(np.int32)
        arrays["claim_leaf"] = np.cumsum(arrays["claim_leaf"].astype(np.int64)).astype(np.int32)
        meta = {str(k): ast.literal_eval(str(v)) for k, v in zip(z["__meta_keys"], z["__meta_vals"])}
        return cls(int(z["__msg_bits"]), arrays, meta)

    @classmethod
    def load(cls, path):
        import ast
        with np.load(path, allow_pickle=False) as z:
            if int(z["__format"]) != FORMAT:
                raise ValueError("cached program has another format")
            meta = {str(k): ast.literal_eval(str(v)) for k, v in zip(z["__meta_keys"], z["__meta_vals"])}
            return cls(int(z["__msg_bits"]), {k: z[k] for k in cls.ARRAYS}, meta)

    # ---- plaintext evaluation ---------------------------------------------------------------------------------------
    def simulate(self, inputs, check=True):
        x = np.asarray([int(v) for v in inputs], np.int64)
        if x.size != self.n_inputs:
            raise ValueError("wrong number of inputs")
        if check:
            bad = np.flatnonzero((x < self.in_lo) | (x > self.in_hi))
            if bad.size:
                i = int(bad[0])
                raise RangeError(f"input {i} = {int(x[i])} outside its declared interval [{int(self.in_lo[i])}, {int(self.in_hi[i])}]")
        n_in = self.n_inputs
        val = np.zeros(n_in + self.n_nodes, np.int64)
        val[:n_in] = x
        order, counts = self.level_order()
        half_space = 1 << (self.msg_bits - 1)
        lens = np.diff(self.node_ptr)
        term_coef = self.term_coef.astype(np.int64)
        pos = 0
        for w in counts.tolist():
            nodes = order[pos: pos + w]
            pos += w
            ln = lens[nodes]
            starts = self.node_ptr[nodes]
            # gather the terms of these nodes: offsets start[i] + (0 .. len[i] - 1)
            tot = int(ln.sum())
            seg = np.repeat(np.arange(w), ln)
            off = np.arange(tot) - np.repeat(np.cumsum(ln) - ln, ln) + np.repeat(starts, ln)
            contrib = term_coef[off] * val[self.term_leaf[off]]
            xin = self.node_const[nodes] + np.bincount(seg, weights=contrib, minlength=w).astype(np.int64)
            p = self.lut_p[self.node_lut[nodes]].astype(np.int64)
            scale = np.left_shift(1, self.msg_bits - p)
            widef = self.node_wide[nodes]
            if check:
                ok = np.where(widef, (xin > -2 * half_space) & (xin < 2 * half_space), (xin >= -half_space) & (xin < half_space))
                if not ok.all():
                    raise RangeError(f"PBS input {int(xin[np.flatnonzero(~ok)[0]])} outside the message space")
                if (xin % scale).any():
                    raise RangeError("PBS input not a multiple of its scale")
            m = xin // scale
            hp = np.left_shift(1, p - 1)
            hi_wrap = widef & (m >= half_space)       # the other half of the torus: negacyclic wrap-around
            lo_wrap = widef & (m < -half_space)
            mm = np.where(hi_wrap, m - 2 * half_space, np.where(lo_wrap, m + 2 * half_space, m))
            out = self.lut_tab[self.node_lut[nodes], mm + hp].astype(np.int64)
            out = np.where(hi_wrap | lo_wrap, -out, out)
            out = np.where(self.node_half[nodes], (out + 1) // 2, out)      # lut_neg: the table holds (bit - 1/2) * 2
            if check and ((out < self.node_lo[nodes]) | (out > self.node_hi[nodes])).any():
                raise RangeError("look-up output outside its interval")
            val[n_in + nodes] = out

        def rows(ptr, leaf, coef, const):
            ln = np.diff(ptr)
            seg = np.repeat(np.arange(ln.size), ln)
            return const + np.bincount(seg, weights=coef * val[leaf], minlength=ln.size).astype(np.int64)

        if check and self.claim_const.size:
            cv = rows(self.claim_ptr, self.claim_leaf, self.claim_coef, self.claim_const)
            bad = np.flatnonzero((cv < self.claim_lo) | (cv > self.claim_hi))
            if bad.size:
                i = int(bad[0])
                raise RangeError(f"interval claim [{int(self.claim_lo[i])}, {int(self.claim_hi[i])}] violated by value {int(cv[i])}")
        return [int(v) for v in rows(self.out_ptr, self.out_leaf, self.out_coef, self.out_const)]


def estimated_evaluate_ms(widths, gpus=1):
    """Cost model of one evaluation (measured per-level costs on one MI355X with the plain 49-bit kernels): a
    level up to 512 ciphertexts wide runs ceil(width / 256) rounds of the latency kernel (3.8 ms each with its keyswitch and
    linear combinations: one workgroup per ciphertext, 256 CUs), a wider one the throughput kernel (9.8 ms per started 1,024
    ciphertexts, 104 PBS per ms once the chip is full).  Used to choose between circuit variants (division radix) and for the
    multi-GPU estimate; what matters is the ratio of a round to a wide level, which the unrolled kernels share.  gpus > 1:
    every level wider than one round is split evenly (each rank bootstraps width / gpus rows; the all-gather of a few MB per
    level over xGMI is not modelled) - an ESTIMATE, no multi-GPU box was available to measure it."""
    w = np.ceil(np.asarray(widths, np.float64) / (gpus if gpus > 1 else 1))
    if gpus > 1:
        w = np.where(np.asarray(widths) <= 256, np.asarray(widths, np.float64), w)
    lat = 3.8 * np.ceil(w / 256)
    tp = np.maximum(9.8 * np.ceil(w / 1024), w / 104.0)
    return float(np.where(w <= 512, lat, tp).sum())


def compile_cached(key_fields, build, cache=True):
    """key_fields: dict naming the configuration; build(): -> Program (slow path: trace + freeze).
    Returns (program, info) with info = {"cached": bool, "seconds": load or build time, "path": file}."""
    t0 = time.time()
    path = None
    if cache:
        key = hashlib.sha256(repr((FORMAT, _tracer_fingerprint(), sorted(key_fields.items()))).encode()).hexdigest()[:24]
        name = "_".join(f"{k}{v}" for k, v in sorted(key_fields.items()) if not isinstance(v, bool) or v)
        path = os.path.join(cache_dir(), f"{name}_{key}.npz".replace(" ", ""))
        if os.path.exists(path):
            try:
                prog = Program.load(path)
                return prog, {"cached": True, "seconds": time.time() - t0, "path": path}
            except Exception:
                pass    # unreadable / stale file: rebuild below and overwrite it
        shipped = os.path.join(shipped_dir(), f"{name}_{key}.prog.xz".replace(" ", ""))
        if os.path.exists(shipped):      # traced by THIS source tree (the key carries its fingerprint) and committed with it
            try:
                prog = Program.load_compact(shipped)
                return prog, {"cached": True, "shipped": True, "seconds": time.time() - t0, "path": shipped}
            except Exception:
                pass
    prog = build()
    if path is not None:
        try:
            prog.save(path)
        except OSError:
            path = None     # read-only tree: stay uncached
    return prog, {"cached": False, "seconds": time.time() - t0, "path": path}
