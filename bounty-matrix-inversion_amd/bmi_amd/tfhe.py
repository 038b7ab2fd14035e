"""ctypes binding of the C ABI in include/bmi_tfhe.h (libbmi_tfhe.so).

This is the binding a maintainer of the reference would add in place of concrete-python's Circuit
object (reference call sites: matrix_inversion/main.py:53-86,177).  It fails loudly when the HIP
library is missing: the PBS path has no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbmi_tfhe.so")
Q = 0xFFFFFFFF00000001  # Goldilocks modulus (q_bits = 64)
TORUS64 = 65                      # bmi_params.q_bits value for q = 2^64 exactly (BMI_Q_TORUS64, Concrete's torus)
MODULUS = {64: 0xFFFFFFFF00000001, 49: 562949952700417, TORUS64: 1 << 64}


class Params(C.Structure):
    _fields_ = [("n", C.c_uint32), ("log_N", C.c_uint32), ("k", C.c_uint32), ("bs_levels", C.c_uint32),
                ("bs_base_log", C.c_uint32), ("ks_levels", C.c_uint32), ("ks_base_log", C.c_uint32),
                ("q_bits", C.c_uint32), ("lwe_noise", C.c_double), ("glwe_noise", C.c_double)]

    @property
    def N(self):
        return 1 << self.log_N

    @property
    def big(self):
        return self.k * self.N + 1

    @property
    def small(self):
        return self.n + 1


class BmiError(RuntimeError):
    pass


_lib = None

_U64P = C.POINTER(C.c_uint64)
_SIGS = {
    "bmi_default_params": [C.POINTER(Params)],
    "bmi_default_params_for": [C.c_uint32, C.POINTER(Params)],
    "bmi_preset_params": [C.c_char_p, C.POINTER(Params)],
    "bmi_ctx_create": [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)],
    "bmi_get_params": [C.c_void_p, C.POINTER(Params)],
    "bmi_keygen": [C.c_void_p],
    "bmi_keygen_insecure_deterministic": [C.c_void_p, C.c_uint64],
    "bmi_export_keys": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "bmi_encrypt": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p],
    "bmi_decrypt": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p],
    "bmi_phase": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_lut_register": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)],
    "bmi_lut_get": [C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_pbs_batch": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p],
    "bmi_keyswitch_batch": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p],
    "bmi_blind_rotate_batch": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p],
    "bmi_lincomb_batch": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                          C.c_void_p, C.c_void_p],
    "bmi_scatter_rows": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p],
    "bmi_pbs_batch_host": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_keyswitch_batch_host": [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_blind_rotate_batch_host": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_negacyclic_mul_host": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
    "bmi_fft_margin_host": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_double)],
    "bmi_sync": [C.c_void_p, C.c_void_p],
    "bmi_reserve": [C.c_void_p, C.c_uint32],
    "bmi_import_keys": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "bmi_keygen_from_secret": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64],
    "bmi_torus64_to_field": [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p],
    "bmi_field_to_torus64": [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p],
    "bmi_set_kernel_variant": [C.c_void_p, C.c_int],
    "bmi_set_keyswitch_variant": [C.c_void_p, C.c_int],
    "bmi_set_bsk_precision": [C.c_void_p, C.c_uint32],
    "bmi_get_bsk_precision": [C.c_void_p, C.POINTER(C.c_uint32)],
    "bmi_set_bsk_unroll": [C.c_void_p, C.c_uint32],
    "bmi_import_bsk_unrolled": [C.c_void_p, C.c_void_p],
    "bmi_export_bsk_unrolled": [C.c_void_p, C.c_void_p],
    "bmi_circuit_prune": [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p],
    "bmi_circuit_schedule": [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                             C.POINTER(C.c_int32)],
    "bmi_key_bytes": [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
}


def circuit_prune(n_in, node_ptr, term_leaf, out_leaf):
    """live mask of the look-ups an output depends on (bmi_circuit_prune; context-free, CPU)"""
    node_ptr = np.ascontiguousarray(node_ptr, np.int64)
    term_leaf = np.ascontiguousarray(term_leaf, np.int32)
    out_leaf = np.ascontiguousarray(out_leaf, np.int32)
    live = np.zeros(max(node_ptr.size - 1, 0), np.uint8)
    rc = load_library().bmi_circuit_prune(int(n_in), live.size, _ptr(node_ptr), _ptr(term_leaf), _ptr(out_leaf),
                                          out_leaf.size, _ptr(live))
    if rc != 0:
        raise BmiError(f"bmi_circuit_prune failed ({rc}): malformed circuit arrays")
    return live.astype(bool)


def circuit_schedule(n_in, node_ptr, term_leaf, round_, wide_round):
    """width-aware level (1-based) of every look-up (bmi_circuit_schedule; context-free, CPU)"""
    node_ptr = np.ascontiguousarray(node_ptr, np.int64)
    term_leaf = np.ascontiguousarray(term_leaf, np.int32)
    level = np.zeros(max(node_ptr.size - 1, 0), np.int32)
    depth = C.c_int32(0)
    rc = load_library().bmi_circuit_schedule(int(n_in), level.size, _ptr(node_ptr), _ptr(term_leaf), int(round_),
                                             int(wide_round), _ptr(level), C.byref(depth))
    if rc != 0:
        raise BmiError(f"bmi_circuit_schedule failed ({rc}): malformed circuit arrays")
    return level


def torus64_to_field(ct, q_bits):
    """modulus switch round(x * q / 2^64) of u64 torus words; context-free (bmi_torus64_to_field)"""
    a = np.ascontiguousarray(ct, dtype=np.uint64)
    out = np.empty_like(a)
    if load_library().bmi_torus64_to_field(int(q_bits), _ptr(a), a.size, _ptr(out)):
        raise BmiError("bmi_torus64_to_field failed")
    return out


def field_to_torus64(ct, q_bits):
    """round(x * 2^64 / q) mod 2^64 of words mod q; context-free (bmi_field_to_torus64)"""
    a = np.ascontiguousarray(ct, dtype=np.uint64)
    out = np.empty_like(a)
    if load_library().bmi_field_to_torus64(int(q_bits), _ptr(a), a.size, _ptr(out)):
        raise BmiError("bmi_field_to_torus64 failed (words must be reduced mod q)")
    return out


def _load_hip_runtime():
    """libbmi_tfhe.so is linked without a HIP runtime of its own (-no-hip-rt): it binds to the one already
    in the process.  PyTorch-ROCm bundles its own libamdhip64.so, and two HIP runtimes in one process cannot
    both own the GPU, so prefer torch's copy when torch is installed; otherwise use the system ROCm."""
    cands = []
    try:
        import torch  # noqa: F401  (plumbing only: device memory / streams / torch.distributed)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    errs = []
    for c in cands:
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:
            errs.append(f"{c}: {e}")
    raise BmiError("no HIP runtime (libamdhip64.so) could be loaded: " + "; ".join(errs))


def load_library():
    """Loads libbmi_tfhe.so; raises BmiError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BmiError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C bounty-matrix-inversion_amd/csrc). There is no CPU fallback for the PBS path.")
        _load_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        lib.bmi_ctx_destroy.argtypes = [C.c_void_p]
        lib.bmi_ctx_destroy.restype = None
        lib.bmi_last_error.argtypes = [C.c_void_p]
        lib.bmi_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def default_params(q_bits=None, **kw):
    """North-star parameter set; q_bits = 64 (Goldilocks) or 49 (f64 kernels), None = the library's default."""
    P = Params()
    lib = load_library()
    rc = lib.bmi_default_params(C.byref(P)) if q_bits is None else lib.bmi_default_params_for(int(q_bits), C.byref(P))
    if rc != 0:
        raise BmiError("unsupported q_bits (64, 49 or TORUS64 = 65)")
    for k, v in kw.items():
        setattr(P, k, v)
    return P


def preset_params(name, **kw):
    """named parameter set of the library: "north_star", "north_star_torus64", "north_star_goldilocks", "secure128"
    (n 742, N 2048: the 128-bit-secure set on the 49-bit field) and "secure128_torus" (the same on q = 2^64, Concrete's
    modulus, l = 3 x 10 bits); include/bmi_tfhe.h"""
    P = Params()
    if load_library().bmi_preset_params(name.encode(), C.byref(P)) != 0:
        raise BmiError(f"unknown parameter preset {name!r}")
    for k, v in kw.items():
        setattr(P, k, v)
    return P


def _ptr(a):
    """host numpy array / torch tensor (host or device) / int -> raw address"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class Engine:
    """One context = crypto parameters + keys + one GPU (reference analogue: a compiled fhe.Circuit)."""

    def __init__(self, params=None, device=0):
        self.lib = load_library()
        self.P = params if params is not None else default_params()
        h = C.c_void_p()
        rc = self.lib.bmi_ctx_create(C.byref(self.P), int(device), C.byref(h))
        if rc != 0:
            raise BmiError(f"bmi_ctx_create failed ({rc}): {self.lib.bmi_last_error(None).decode()}")
        self.h = h
        self.device = device
        self._luts = {}
        self.q_bits = self.P.q_bits or 64
        self.modulus = MODULUS[self.q_bits]
        self.log_q = 49 if self.q_bits == 49 else 64     # bits of the torus messages are scaled on

    def delta_log(self, msg_bits=4):
        """scaling exponent of a signed msg_bits-bit message space: log_q - 1 - msg_bits (59 / 44 for 4 bits)"""
        return self.log_q - 1 - msg_bits

    def close(self):
        if getattr(self, "h", None):
            self.lib.bmi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise BmiError(f"{what} failed ({rc}): {self.lib.bmi_last_error(self.h).decode()}")

    # ---- keys
    def keygen(self, seed=None):
        """seed=None: production key generation, all randomness from the library's CSPRNG (ChaCha20 keyed by getrandom).
        An integer seed selects bmi_keygen_insecure_deterministic: reproducible keys AND reproducible encryptions for
        the oracle parity tests and for replicating one key set on every rank of a benchmark - not cryptographic."""
        if seed is None:
            self._ck(self.lib.bmi_keygen(self.h), "bmi_keygen")
        else:
            self._ck(self.lib.bmi_keygen_insecure_deterministic(self.h, C.c_uint64(seed)), "bmi_keygen_insecure_deterministic")

    def keygen_shared(self, seed=None, group=None, src=0, share_secret=False):
        """One set of EVALUATION keys on every rank of a torch.distributed group WITHOUT the seeded (test-only) generator: rank
        `src` generates the key set (CSPRNG when seed is None) and broadcasts the bootstrap and keyswitch keys (about 100 MB at
        the north-star set; the unrolled key too in unrolled mode); the other ranks import them as evaluation-only contexts -
        the secret keys stay on `src`, the only rank that can encrypt / decrypt (share_secret=True replicates them as well, over
        whatever transport the group uses: only for tests and trusted single-node groups).  With a single process this is
        keygen(seed).  The sharded executor needs the same evaluation keys on every GPU and checks it (executor.py)."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self.keygen(seed)
        rank = dist.get_rank(group)
        unrolled = getattr(self, "unroll", 1) == 2     # the unrolled bootstrap key travels with the set
        P, rows = self.P, (self.P.k + 1) * self.P.bs_levels
        if rank == src:
            self.keygen(seed)
            sk_small, sk_big, bsk, ksk = self.export_keys()
            parts = ([sk_small, sk_big] if share_secret else []) + [bsk, ksk] + ([self.export_bsk_unrolled()] if unrolled else [])
        else:
            parts = ([np.zeros(P.n, np.uint64), np.zeros(P.k * P.N, np.uint64)] if share_secret else []) + \
                    [np.zeros((P.n, rows, P.k + 1, P.N), np.uint64), np.zeros((P.k * P.N, P.ks_levels, P.n + 1), np.uint64)]
            if unrolled:
                parts.append(np.zeros(self.unrolled_key_shape(), np.uint64))
        on_gpu = dist.get_backend(group) == "nccl"
        for a in parts:
            t = torch.from_numpy(a.view(np.int64))
            if on_gpu:
                t = t.to(torch.device("cuda", self.device))
            dist.broadcast(t, src=src, group=group)
            if on_gpu:
                a.view(np.int64)[...] = t.cpu().numpy()
        if rank != src:
            sec = parts[:2] if share_secret else [None, None]
            ev = parts[2:] if share_secret else parts
            self.import_keys(sec[0], sec[1], ev[0], ev[1])
            if unrolled:
                self.import_bsk_unrolled(ev[2])

    def eval_key_fingerprint(self):
        """64-bit fingerprint of the evaluation keys this context holds (bootstrap, keyswitch and, in unrolled mode, the unrolled
        key): equal on two contexts iff they bootstrap alike.  Used by the sharded executor to refuse ranks with different keys."""
        import hashlib
        _, _, bsk, ksk = self.export_keys(secret=False)
        h = hashlib.blake2b(digest_size=8)
        for a in (bsk, ksk) + ((self.export_bsk_unrolled(),) if getattr(self, "unroll", 1) == 2 else ()):
            h.update(np.ascontiguousarray(a).view(np.uint8).data)
        return int.from_bytes(h.digest(), "little") >> 1      # fits a signed 64-bit tensor element

    def export_keys(self, secret=True):
        """(sk_small, sk_big, bsk, ksk), standard domain; secret=False returns (None, None, bsk, ksk) and also works on
        an evaluation-only context"""
        P = self.P
        rows = (P.k + 1) * P.bs_levels
        sk_small = np.zeros(P.n, np.uint64) if secret else None
        sk_big = np.zeros(P.k * P.N, np.uint64) if secret else None
        bsk = np.zeros((P.n, rows, P.k + 1, P.N), np.uint64)
        ksk = np.zeros((P.k * P.N, P.ks_levels, P.n + 1), np.uint64)
        self._ck(self.lib.bmi_export_keys(self.h, _ptr(sk_small), _ptr(sk_big), _ptr(bsk), _ptr(ksk)), "bmi_export_keys")
        return sk_small, sk_big, bsk, ksk

    def import_keys(self, sk_small, sk_big, bsk, ksk):
        """loads a key set (layout of export_keys); sk_small = sk_big = None makes this context evaluation-only"""
        P = self.P
        rows = (P.k + 1) * P.bs_levels
        bsk = np.ascontiguousarray(bsk, dtype=np.uint64)
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        if bsk.size != P.n * rows * (P.k + 1) * P.N or ksk.size != P.k * P.N * P.ks_levels * (P.n + 1):
            raise BmiError("key arrays do not match this context's parameters")
        if (sk_small is None) != (sk_big is None):
            raise BmiError("pass both secret keys or neither")
        if sk_small is not None:
            sk_small = np.ascontiguousarray(sk_small, dtype=np.uint64)
            sk_big = np.ascontiguousarray(sk_big, dtype=np.uint64)
            if sk_small.size != P.n or sk_big.size != P.k * P.N:
                raise BmiError("secret key arrays do not match this context's parameters")
        self._ck(self.lib.bmi_import_keys(self.h, _ptr(sk_small), _ptr(sk_big), _ptr(bsk), _ptr(ksk)), "bmi_import_keys")

    def keygen_from_secret(self, sk_small, sk_big, seed=0):
        """evaluation keys for binary secret keys made elsewhere (e.g. by a Concrete client); seed=0: CSPRNG masks and
        noise, seed != 0: deterministic test vectors (not cryptographic)"""
        sk_small = np.ascontiguousarray(sk_small, dtype=np.uint64)
        sk_big = np.ascontiguousarray(sk_big, dtype=np.uint64)
        if sk_small.size != self.P.n or sk_big.size != self.P.k * self.P.N:
            raise BmiError("secret key arrays do not match this context's parameters")
        self._ck(self.lib.bmi_keygen_from_secret(self.h, _ptr(sk_small), _ptr(sk_big), int(seed)), "bmi_keygen_from_secret")

    def from_torus64(self, ct):
        """ciphertext words on the 2^64 torus (Concrete's representation) -> words mod q (same shape)"""
        return torus64_to_field(ct, self.q_bits)

    def to_torus64(self, ct):
        """ciphertext words mod q -> words on the 2^64 torus (same shape)"""
        return field_to_torus64(ct, self.q_bits)

    _PARAM_FIELDS = ("n", "log_N", "k", "bs_levels", "bs_base_log", "ks_levels", "ks_base_log", "q_bits", "lwe_noise", "glwe_noise")

    def save_keys(self, path, secret=True):
        """Key file (numpy .npz): the parameter set and the standard-domain keys.  secret=False writes the evaluation
        keys only - what a server needs (the reference's analogue: Concrete's key cache, qfloat_matrix_inversion.py:997)."""
        sk_small, sk_big, bsk, ksk = self.export_keys(secret=secret)
        arrays = {"bsk": bsk, "ksk": ksk, "params": np.array([float(getattr(self.P, f)) for f in self._PARAM_FIELDS])}
        if secret:
            arrays.update(sk_small=sk_small, sk_big=sk_big)
        if getattr(self, "unroll", 1) == 2:          # the unrolled bootstrap key travels with the set
            arrays["bsk_unrolled"] = self.export_bsk_unrolled()
        with open(path, "wb") as f:
            np.savez(f, **arrays)

    def load_keys(self, path):
        """loads a key file written by save_keys; the file's parameter set must equal this context's"""
        with np.load(path) as z:
            mine = [float(getattr(self.P, f)) for f in self._PARAM_FIELDS]
            mine[7] = float(self.q_bits)
            theirs = list(z["params"])
            if theirs[7] == 0:
                theirs[7] = 64.0
            if mine != theirs:
                raise BmiError(f"key file parameters {theirs} differ from this context's {mine}")
            has_secret = "sk_small" in z.files
            self.import_keys(z["sk_small"] if has_secret else None, z["sk_big"] if has_secret else None, z["bsk"], z["ksk"])
            if "bsk_unrolled" in z.files:            # written by a context in unrolled mode: this one switches to it too
                self.set_bsk_unroll(2)
                self.import_bsk_unrolled(z["bsk_unrolled"])
            elif getattr(self, "unroll", 1) == 2 and has_secret:
                self.set_bsk_unroll(2)               # no unrolled key in the file: derived from the secret keys just loaded
        return has_secret

    def key_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.bmi_key_bytes(self.h, C.byref(a), C.byref(b)), "bmi_key_bytes")
        return a.value, b.value

    # ---- encrypt / decrypt (host buffers)
    def encrypt(self, msgs, delta_log):
        msgs = np.ascontiguousarray(msgs, dtype=np.int64).reshape(-1)
        out = np.zeros((msgs.size, self.P.big), np.uint64)
        self._ck(self.lib.bmi_encrypt(self.h, _ptr(msgs), msgs.size, delta_log, _ptr(out)), "bmi_encrypt")
        return out

    def decrypt(self, cts, delta_log):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.P.big)
        out = np.zeros(cts.shape[0], np.int64)
        self._ck(self.lib.bmi_decrypt(self.h, _ptr(cts), cts.shape[0], delta_log, _ptr(out)), "bmi_decrypt")
        return out

    def phase(self, cts):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.P.big)
        out = np.zeros(cts.shape[0], np.uint64)
        self._ck(self.lib.bmi_phase(self.h, _ptr(cts), cts.shape[0], _ptr(out)), "bmi_phase")
        return out

    # ---- LUTs
    def lut_register(self, table, msg_bits, out_delta_log):
        table = np.ascontiguousarray(table, dtype=np.int64).reshape(-1)
        if table.size != 1 << msg_bits:
            raise ValueError("table must have 2^msg_bits entries")
        key = (table.tobytes(), msg_bits, out_delta_log)
        if key in self._luts:
            return self._luts[key]
        lid = C.c_uint32()
        self._ck(self.lib.bmi_lut_register(self.h, _ptr(table), msg_bits, out_delta_log, C.byref(lid)), "bmi_lut_register")
        self._luts[key] = lid.value
        return lid.value

    def lut_get(self, lut_id):
        tv = np.zeros(self.P.N, np.uint64)
        self._ck(self.lib.bmi_lut_get(self.h, lut_id, _ptr(tv)), "bmi_lut_get")
        return tv

    def set_kernel_variant(self, v):
        """blind rotation: 0 auto, 1 pair kernel exchanging per level, 2 latency kernel (two wavefronts per transform on
        the 49-bit field), 3 pair kernel exchanging per CMUX, 4 latency kernel with one wavefront per transform, 5 (2^64 torus,
        48-bit key in base 2^10) pair kernel with the exact limb products carried by the f64 complex FFT, 6 its latency form"""
        self._ck(self.lib.bmi_set_kernel_variant(self.h, int(v)), "bmi_set_kernel_variant")

    def set_bsk_precision(self, bits):
        """2^64 torus, before keygen: 64 = exact key (three limbs); 48 = key rounded to 48 bits (two limbs, 2/3 of the work; the
        default at Bg = 2^10, the torus parameter set); 42 = rounded to 42 bits (two limbs at Bg = 2^15, noisier)"""
        self._ck(self.lib.bmi_set_bsk_precision(self.h, int(bits)), "bmi_set_bsk_precision")

    @property
    def bsk_precision(self):
        """bits of precision the bootstrap key is stored at (64 = exact; the prime fields always)"""
        v = C.c_uint32()
        self._ck(self.lib.bmi_get_bsk_precision(self.h, C.byref(v)), "bmi_get_bsk_precision")
        return v.value

    def set_bsk_unroll(self, factor):
        """49-bit field, N = 1024 (or N = 2048 with l <= 2), or the 2^64 torus at its default set: 1 = CGGI's blind rotation (default), 2 = two LWE coefficients per step with an unrolled
        bootstrap key (generated by the next keygen, or at once from the secret keys already held); include/bmi_tfhe.h"""
        self._ck(self.lib.bmi_set_bsk_unroll(self.h, int(factor)), "bmi_set_bsk_unroll")
        self.unroll = int(factor)

    def unrolled_key_shape(self):
        P = self.P
        return ((P.n + 1) // 2, 3, (P.k + 1) * P.bs_levels, P.k + 1, P.N)

    def export_bsk_unrolled(self):
        bsk3 = np.zeros(self.unrolled_key_shape(), np.uint64)
        self._ck(self.lib.bmi_export_bsk_unrolled(self.h, _ptr(bsk3)), "bmi_export_bsk_unrolled")
        return bsk3

    def import_bsk_unrolled(self, bsk3):
        """the unrolled bootstrap key of the key set this context holds (layout of export_bsk_unrolled)"""
        bsk3 = np.ascontiguousarray(bsk3, dtype=np.uint64)
        if bsk3.size != int(np.prod(self.unrolled_key_shape())):
            raise BmiError("unrolled key array does not match this context's parameters")
        self._ck(self.lib.bmi_import_bsk_unrolled(self.h, _ptr(bsk3)), "bmi_import_bsk_unrolled")

    def set_keyswitch_variant(self, v):
        """keyswitch: 0 auto (int8 matrix-core product), 1 scalar kernel"""
        self._ck(self.lib.bmi_set_keyswitch_variant(self.h, int(v)), "bmi_set_keyswitch_variant")

    # ---- hot path, host buffers (numpy in / numpy out)
    def pbs_host(self, cts, lut_ids):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.P.big)
        ids = np.ascontiguousarray(lut_ids, dtype=np.uint32).reshape(-1)
        assert ids.size == cts.shape[0]
        out = np.zeros_like(cts)
        self._ck(self.lib.bmi_pbs_batch_host(self.h, _ptr(cts), _ptr(ids), cts.shape[0], _ptr(out)), "bmi_pbs_batch_host")
        return out

    def keyswitch_host(self, cts):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.P.big)
        out = np.zeros((cts.shape[0], self.P.small), np.uint64)
        self._ck(self.lib.bmi_keyswitch_batch_host(self.h, _ptr(cts), cts.shape[0], _ptr(out)), "bmi_keyswitch_batch_host")
        return out

    def blind_rotate_host(self, small, lut_ids):
        small = np.ascontiguousarray(small, dtype=np.uint64).reshape(-1, self.P.small)
        ids = np.ascontiguousarray(lut_ids, dtype=np.uint32).reshape(-1)
        out = np.zeros((small.shape[0], self.P.big), np.uint64)
        self._ck(self.lib.bmi_blind_rotate_batch_host(self.h, _ptr(small), _ptr(ids), small.shape[0], _ptr(out)),
                 "bmi_blind_rotate_batch_host")
        return out

    def fft_margin_host(self, small, lut_ids):
        """test hook (2^64 torus, 48-bit key in base 2^10): blind rotation by the floating-point-transform wave-pair kernel ->
        (outputs, largest distance of a limb sum from the integer it was rounded to)"""
        small = np.ascontiguousarray(small, dtype=np.uint64).reshape(-1, self.P.small)
        ids = np.ascontiguousarray(lut_ids, dtype=np.uint32).reshape(-1)
        out = np.zeros((small.shape[0], self.P.big), np.uint64)
        dist = C.c_double(0.0)
        self._ck(self.lib.bmi_fft_margin_host(self.h, _ptr(small), _ptr(ids), small.shape[0], _ptr(out), C.byref(dist)),
                 "bmi_fft_margin_host")
        return out, float(dist.value)

    def negacyclic_mul_host(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.P.N)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, self.P.N)
        c = np.zeros_like(a)
        self._ck(self.lib.bmi_negacyclic_mul_host(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(c)), "bmi_negacyclic_mul_host")
        return c

    # ---- hot path, device buffers (torch int64 tensors viewed as uint64 words; `stream` = raw hipStream_t)
    def pbs(self, d_in, d_lut_ids, count, d_out, stream=0):
        self._ck(self.lib.bmi_pbs_batch(self.h, _ptr(d_in), _ptr(d_lut_ids), count, _ptr(d_out), C.c_void_p(stream)),
                 "bmi_pbs_batch")

    def keyswitch(self, d_in, count, d_small, stream=0):
        self._ck(self.lib.bmi_keyswitch_batch(self.h, _ptr(d_in), count, _ptr(d_small), C.c_void_p(stream)),
                 "bmi_keyswitch_batch")

    def blind_rotate(self, d_small, d_lut_ids, count, d_out, stream=0):
        self._ck(self.lib.bmi_blind_rotate_batch(self.h, _ptr(d_small), _ptr(d_lut_ids), count, _ptr(d_out),
                                                 C.c_void_p(stream)), "bmi_blind_rotate_batch")

    def lincomb(self, d_store, d_row_ptr, d_idx, d_coef, d_const, count, d_out, stream=0):
        self._ck(self.lib.bmi_lincomb_batch(self.h, _ptr(d_store), _ptr(d_row_ptr), _ptr(d_idx), _ptr(d_coef),
                                            _ptr(d_const), count, _ptr(d_out), C.c_void_p(stream)), "bmi_lincomb_batch")

    def scatter_rows(self, d_src, count, d_store, d_rows, stream=0):
        self._ck(self.lib.bmi_scatter_rows(self.h, _ptr(d_src), count, _ptr(d_store), _ptr(d_rows), C.c_void_p(stream)),
                 "bmi_scatter_rows")

    def reserve(self, max_count):
        self._ck(self.lib.bmi_reserve(self.h, int(max_count)), "bmi_reserve")

    def sync(self, stream=0):
        self._ck(self.lib.bmi_sync(self.h, C.c_void_p(stream)), "bmi_sync")
