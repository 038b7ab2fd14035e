"""EncryptedMatrixInversion — the drop-in counterpart of the reference's API wrapper
(matrix_inversion/main.py:17-116): same constructor arguments and the same
quantize / encrypt / evaluate / decrypt / dequantize / run methods (+ keygen, which the reference reaches
as `circuit.keygen()`, main.py:177).  `evaluate` runs every PBS on the MI355X through libbmi_tfhe.so and
raises if the library or the GPU is missing; `run(..., simulate=True)` evaluates the traced circuit in
plaintext, exactly like the reference's `circuit.simulate` branch (main.py:106-109)."""
from __future__ import annotations

import time
from typing import Tuple

import numpy as np

from .circuit import Circuit, MSG_BITS   # noqa: F401
from .program import Program, compile_cached   # noqa: F401
from .qfloat import QFloat   # noqa: F401
from .qfloat_matrix_inversion import float_matrix_to_qfloat_arrays, qfloat_and_signs_arrays_to_float_matrix
# the circuit of a configuration (tracing, division radix, compiled + cached program) lives in inverse_circuit.py - the module the
# tracer fingerprint covers; re-exported here under the names importers use
from .inverse_circuit import (estimated_evaluate_ms, message_bits_for, trace_inverse, _trace_inverse,   # noqa: F401
                              compile_inverse)


class EncryptedMatrixInversion:
    shape: Tuple[int, int]

    def __init__(self, n, sampler=None, qfloat_base=2, qfloat_len=32, qfloat_ints=16, true_division=False,
                 tensorize=False, engine=None, device=0, shard_threshold=None, cache=True, unroll=False, q_bits=None,
                 params=None, p_error=None):
        """The reference's seven arguments (main.py:17-36), then: engine / device (the GPU context to use) and
        shard_threshold (with torch.distributed initialised on several ranks, levels at least this wide are split
        across the ranks' GPUs; None = every level wider than one kernel round, levels re-packed for the rank count,
        see executor.py); unroll (when this object creates the engine, 4-bit look-ups): bootstrap-key unrolling, two LWE
        coefficients per blind-rotation step (bmi_set_bsk_unroll; key noise 2^-41 so that the look-up margin of the default set
        is kept) - 2.6 ms per level instead of 3.6; q_bits (when this object creates the engine): the ciphertext modulus, None =
        the library's default = tfhe.TORUS64 = 2^64, the torus concrete-python computes on (its default set: Bg = 2^10,
        bootstrap key at 48 bits of precision; with unroll=True the unrolled torus kernel, key noise unchanged), 49 = the prime field
        2^49 - 720895 (with unroll=True the engine with the shortest single bootstrap: 0.83 s for the 3x3 against 1.26 s); params (when
        this object creates the engine): a named parameter set of the library ("secure128_torus", "secure128": the 128-bit-secure
        sets, include/bmi_tfhe.h) or a tfhe.Params - what `fhe.Compiler.compile`'s parameter optimiser chooses for the reference
        (main.py:53-66); it overrides q_bits; p_error (when this object creates the engine and params is not given): the bound on
        the probability that noise makes the whole evaluation wrong (Concrete's `global_p_error`, 1e-5 there) - the first
        parameter set of the chosen modulus whose error budget for THIS circuit meets it is taken (error_budget.choose_params:
        N = 1024 for small circuits, N = 2048 where the look-up count needs the wider margin, e.g. the 8x8 inverse); None keeps
        the north-star set and reports its budget in `self.error_budget`."""
        self.shape = (n, n)
        self.qfloat_base, self.qfloat_len, self.qfloat_ints = qfloat_base, qfloat_len, qfloat_ints
        self.true_division, self.tensorize = true_division, tensorize
        # The reference calls sampler() 100 times to let Concrete measure value ranges (main.py:41-47);
        # here ranges are derived by interval analysis, so the sampler is only validated.
        if sampler is not None:
            s = sampler()
            assert isinstance(s, np.ndarray) and np.issubdtype(s.dtype, np.floating) and s.shape == self.shape
        t0 = time.time()
        self.program, self.compile_info = compile_inverse(n, qfloat_len, qfloat_ints, qfloat_base, true_division,
                                                          tensorize, cache=cache)
        self.circuit = self.program      # what the reference calls the compiled circuit
        self.msg_bits = self.program.msg_bits
        self.trace_seconds = time.time() - t0
        self.engine = engine
        self.device = device
        self.shard_threshold = shard_threshold
        self.unroll = bool(unroll)
        self.q_bits = q_bits
        self.params = params
        self.p_error = p_error
        self.error_budget = None      # filled when the engine exists: Program.failure_probability under its parameters
        if engine is not None and params is not None:
            raise ValueError("pass an engine or a parameter set, not both (the engine already has its parameters)")
        if engine is not None and q_bits is not None and engine.q_bits != q_bits:
            raise ValueError(f"q_bits={q_bits} but the engine passed in computes on q_bits={engine.q_bits}")
        if self.unroll and engine is not None and getattr(engine, "unroll", 1) != 2:
            raise ValueError("unroll=True with an engine that is not in unrolled mode: call engine.set_bsk_unroll(2) before its "
                             "keygen, or let this object create the engine")
        self._exec = None

    # ---- key generation / engine -------------------------------------------------------------------
    def _engine(self):
        if self.engine is None:
            from . import tfhe  # raises BmiError when libbmi_tfhe.so or the GPU is missing: no CPU fallback
            # 4-bit look-ups: the north-star set (N = 1024); 5-bit ones (bases other than 2): N = 2048, same margin
            qb = self.q_bits
            if self.params is not None:    # a named set of the library, or a tfhe.Params
                params = tfhe.preset_params(self.params) if isinstance(self.params, str) else self.params
            elif self.p_error is not None:  # the first set of this modulus whose error budget for this circuit meets p_error
                from . import error_budget
                params, self.error_budget = error_budget.choose_params(self.program, self.p_error, q_bits=qb, unroll=self.unroll)
            elif self.msg_bits <= 4:
                params = tfhe.default_params(q_bits=qb)
            elif self.msg_bits <= 6 and qb in (None, tfhe.TORUS64, 49):
                # 5-bit look-ups: N = 2048, 6-bit ones: N = 4096 - one more message bit per doubling of N at the same margin; on the
                # library's default modulus (the 2^64 torus: k_blind_rotate_w_t64f / k_blind_rotate_q_t64f) unless q_bits says 49
                params = tfhe.default_params(log_N=self.msg_bits + 6, **({} if qb is None else dict(q_bits=qb)))
            else:
                raise ValueError("look-ups wider than 4 bits need N >= 2048, which exists on the 2^64 torus and the 49-bit field; "
                                 "wider than 6 bits: no parameter set")
            if self.unroll:
                if self.msg_bits > 4 or params.log_N != 10:
                    raise ValueError("bootstrap-key unrolling exists at N = 1024 (4-bit look-ups) only")
                if params.q_bits == 49 and self.params is None:     # three products per step: key noise 2^-41 keeps the default set's output noise
                    params = tfhe.default_params(q_bits=49, glwe_noise=2.0 ** -41)
                elif params.q_bits != tfhe.TORUS64:
                    raise ValueError("bootstrap-key unrolling exists on the 49-bit field and on the 2^64 torus")
            self.engine = tfhe.Engine(params, device=self.device)
            if self.unroll:
                self.engine.set_bsk_unroll(2)
        if self.engine.P.N < (1 << (self.msg_bits + 6)):
            raise ValueError(f"{self.msg_bits}-bit look-ups need a parameter set with N >= {1 << (self.msg_bits + 6)} "
                             f"(this engine has N = {self.engine.P.N})")
        if self.error_budget is None:
            self.error_budget = self.program.failure_probability(self.engine)
            if self.p_error is not None and self.error_budget["p_fail"] > self.p_error:
                raise ValueError(f"this circuit's failure probability under the engine's parameters is {self.error_budget['p_fail']:.2e} "
                                 f"> p_error = {self.p_error:g}: pass a wider parameter set (e.g. log_N=11) or let this object choose")
        return self.engine

    def keygen(self, seed=None):
        """circuit.keygen() (main.py:177).  seed=None: CSPRNG keys; an integer: the test-only deterministic key set.
        With torch.distributed initialised on several ranks (the sharded executor) rank 0 generates the set and the others
        receive its evaluation keys (Engine.keygen_shared): independent CSPRNG keys per rank would make the ranks' parts of a
        level undecryptable garbage to each other.  Only rank 0 can then encrypt / decrypt."""
        eng = self._engine()
        try:
            import torch.distributed as dist
            shared = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:
            shared = False
        if shared:
            eng.keygen_shared(seed)
        else:
            eng.keygen(seed)

    def _executor(self, batch=1):
        """the executor of this circuit for `batch` input matrices per run (one per batch size, built on first use)"""
        if self._exec is None:
            self._exec = {}
        if batch not in self._exec:
            from .executor import Executor
            self._exec[batch] = Executor(self.circuit, self._engine(), shard_threshold=self.shard_threshold, batch=batch)
        return self._exec[batch]

    # ---- the reference's six methods -----------------------------------------------------------------
    def quantize(self, matrix: np.ndarray):
        return float_matrix_to_qfloat_arrays(matrix, self.qfloat_len, self.qfloat_ints, self.qfloat_base)

    def _flat_inputs(self, quantized_matrix, qfloats_signs):
        q = np.asarray(quantized_matrix, dtype=np.int64)
        s = np.asarray(qfloats_signs, dtype=np.int64)
        n2 = self.shape[0] * self.shape[1]
        if q.shape != (n2, self.qfloat_len) or s.shape != (n2,):
            raise ValueError("quantized matrix / signs have the wrong shape")
        flat = np.concatenate([q.reshape(-1), s])
        c = self.circuit
        for i, v in enumerate(flat):
            if not (c.leaf_lo[i] <= v <= c.leaf_hi[i]):
                raise ValueError(f"input {i} = {v} outside the interval [{c.leaf_lo[i]}, {c.leaf_hi[i]}] the circuit "
                                 "was traced for (matrix entry too large for qfloat_ints?)")
        return flat

    def encrypt(self, quantized_matrix: np.ndarray, qfloats_signs: np.ndarray) -> np.ndarray:
        flat = self._flat_inputs(quantized_matrix, qfloats_signs)
        return self._engine().encrypt(flat, self._engine().delta_log(self.msg_bits))

    def evaluate(self, encrypted_quantized_matrix: np.ndarray) -> np.ndarray:
        return self._executor().run(encrypted_quantized_matrix)

    def evaluate_many(self, encrypted_quantized_matrices) -> np.ndarray:
        """B encrypted matrices (a sequence of encrypt() results, or one (B, inputs, words) array) through the circuit in ONE walk of
        its levels: every level is B times wider, so the look-ups run on the throughput kernel instead of B x depth rounds of the
        latency kernel (executor.py, `batch`) - the serving form.  Returns (B, outputs, words); decrypt() each."""
        enc = np.ascontiguousarray(np.stack([np.asarray(e) for e in encrypted_quantized_matrices]), dtype=np.uint64)
        if enc.ndim != 3:
            raise ValueError("expected B encrypted matrices of shape (inputs, words)")
        return self._executor(enc.shape[0]).run(enc).reshape(enc.shape[0], -1, enc.shape[2])

    def run_many(self, matrices, validate=True):
        """run() for a batch of matrices on one GPU: quantize, encrypt, ONE batched encrypted evaluation (evaluate_many), decrypt,
        dequantize - the same inverses as run() matrix by matrix"""
        if self._dist() is not None:
            raise NotImplementedError("run_many is the one-GPU serving form; under torch.distributed call run() per matrix")
        qs = []
        for m in matrices:
            assert np.issubdtype(m.dtype, np.floating) and m.shape == self.shape
            q = self.quantize(m)
            if validate:
                self.simulate(*q)
            qs.append(q)
        res = self.evaluate_many([self.encrypt(q, s_) for q, s_ in qs])
        return [self.dequantize(self.decrypt(r)) for r in res]

    def decrypt(self, encrypted_quantized_inverted_matrix: np.ndarray) -> np.ndarray:
        n2 = self.shape[0] * self.shape[1]
        m = self._engine().decrypt(encrypted_quantized_inverted_matrix, self._engine().delta_log(self.msg_bits))
        return m.reshape(n2, self.qfloat_len + 1)

    def dequantize(self, quantized_inverted_matrix: np.ndarray) -> np.ndarray:
        return qfloat_and_signs_arrays_to_float_matrix(quantized_inverted_matrix, self.qfloat_ints, self.qfloat_base)

    def simulate(self, quantized_matrix, qfloats_signs) -> np.ndarray:
        n2 = self.shape[0] * self.shape[1]
        flat = self._flat_inputs(quantized_matrix, qfloats_signs)
        return np.array(self.circuit.simulate(flat), dtype=np.int64).reshape(n2, self.qfloat_len + 1)

    @staticmethod
    def _dist():
        """torch.distributed when it is initialised on more than one rank, else None"""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                return dist
        except Exception:
            pass
        return None

    def _broadcast(self, dist, arr, src=0):
        """a numpy array from rank `src` to every rank (through the GPU when the backend is RCCL)"""
        import torch
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a)
        on_gpu = dist.get_backend() == "nccl"
        if on_gpu:
            t = t.to(torch.device("cuda", self._engine().device))
        dist.broadcast(t, src=src)
        out = t.cpu().numpy() if on_gpu else t.numpy()
        return out.view(a.dtype)

    def run(self, matrix: np.ndarray, simulate=False, validate=True) -> np.ndarray:
        """The reference's one-call form (main.py:93-116).  validate (encrypted runs): the caller of run() holds the
        plaintext, so the compiled program is first evaluated in plaintext with every interval claim checked
        (program.Program.simulate: 0.07 s for 3x3, 2 s for 8x8) - an input the circuit was not traced for (a singular
        matrix whose reciprocal overflows its format, an entry beyond the leading-digit range) raises RangeError here
        instead of decrypting to garbage silently, which is what an encrypted evaluation outside its ranges does (on
        Concrete as well: its circuits are only defined on the ranges their inputset showed).  With torch.distributed
        initialised on several ranks every rank calls run() with the same matrix: rank 0 (the holder of the secret keys after
        keygen()) encrypts and decrypts, ciphertexts and result are broadcast, every rank returns the same inverse."""
        assert np.issubdtype(matrix.dtype, np.floating)
        assert matrix.shape == self.shape
        quantized_matrix, qfloats_signs = self.quantize(matrix)
        if not simulate:
            if validate:
                self.simulate(quantized_matrix, qfloats_signs)
            dist = self._dist()
            if dist is None:
                enc = self.encrypt(quantized_matrix, qfloats_signs)
                enc_inv = self.evaluate(enc)
                quantized_inverted_matrix = self.decrypt(enc_inv)
            else:
                # several ranks (the sharded executor): keygen() left the secret keys on rank 0 only, so rank 0 encrypts and
                # decrypts and the ciphertexts / the result travel by broadcast; every rank walks the levels and returns the
                # same matrix
                eng = self._engine()
                n_ct = self.circuit.n_inputs
                enc = (self.encrypt(quantized_matrix, qfloats_signs) if dist.get_rank() == 0
                       else np.zeros((n_ct, eng.P.big), np.uint64))
                enc = self._broadcast(dist, enc)
                enc_inv = self.evaluate(enc)
                n2 = self.shape[0] * self.shape[1]
                quantized_inverted_matrix = (self.decrypt(enc_inv) if dist.get_rank() == 0
                                             else np.zeros((n2, self.qfloat_len + 1), np.int64))
                quantized_inverted_matrix = self._broadcast(dist, quantized_inverted_matrix)
        else:
            quantized_inverted_matrix = self.simulate(quantized_matrix, qfloats_signs)
        inverted_matrix = self.dequantize(quantized_inverted_matrix)
        assert np.issubdtype(inverted_matrix.dtype, np.floating)
        assert inverted_matrix.shape == self.shape
        return inverted_matrix
