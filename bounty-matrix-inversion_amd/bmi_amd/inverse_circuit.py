"""The PBS circuit of qfloat_matrix_inverse for one configuration: tracing, the choice of the division radix, and the compiled,
cached program (the analogue of fhe.Compiler(...).compile, reference matrix_inversion/main.py:53-66).  Kept apart from the API
wrapper (main.py) because this module is part of the tracer fingerprint (program._tracer_fingerprint): editing it invalidates the
cached and the shipped programs, editing the wrapper does not."""
from __future__ import annotations

from .circuit import Circuit, MSG_BITS
from .program import Program, compile_cached
from .program import estimated_evaluate_ms as _estimate_widths
from .qfloat import QFloat
from .qfloat_matrix_inversion import qfloat_matrix_inverse


def estimated_evaluate_ms(circuit):
    """Cost model of one evaluation on one MI355X (program.estimated_evaluate_ms: rounds of the latency kernel for levels up to
    512 ciphertexts wide, the throughput kernel beyond)."""
    widths = circuit.level_widths() if isinstance(circuit, Program) else [len(lv) for lv in circuit.levels()]
    return _estimate_widths(widths)


def message_bits_for(qfloat_base):
    """Message bits of the look-ups a configuration needs: 4 for base 2 (every site restructured to fit);
    other bases keep the reference's digit sums, whose leading-digit sums span up to 39 values (base 3): 5 bits with the
    odd-function look-ups, i.e. the N = 2048 parameter sets (what Concrete solves by choosing wider parameters itself)."""
    return 4 if qfloat_base == 2 else 5


def trace_inverse(n, qfloat_len, qfloat_ints, qfloat_base=2, true_division=False, tensorize=False, division_bits=None,
                  msg_bits=None):
    """Builds the PBS circuit of qfloat_matrix_inverse (the analogue of fhe.Compiler(...).compile, main.py:53-66).
    Inputs are declared in the order: all n^2 * len digits (row-major), then the n^2 signs.
    Digit intervals: leading digit [0, 2*base - 1] (from_float does not reduce it), others
    [0, base - 1]; signs [-1, 1].

    division_bits: quotient bits per step of the binary divisions (2 or 3, base_p_arrays._division_radix); None
    traces both for n <= 4 and keeps the circuit the cost model above estimates faster (a step costs the same look-up
    levels at either radix - three or four of borrow look-ahead and one selection - so radix 8 is a third shallower, but
    2.3x wider per step; larger matrices are bound by throughput, where its 13 % more look-ups lose: radix 4)."""
    from . import base_p_arrays as bpa
    if division_bits is None:
        cands = (2, 3) if (n <= 4 and qfloat_base == 2) else (2,)
        best = None
        for bits in cands:
            cir = trace_inverse(n, qfloat_len, qfloat_ints, qfloat_base, true_division, tensorize, bits, msg_bits)
            est = estimated_evaluate_ms(cir) if len(cands) > 1 else 0.0
            if best is None or est < best[0]:
                best = (est, cir)
        return best[1]
    saved = bpa.DIVISION_BITS
    bpa.DIVISION_BITS = int(division_bits)
    try:
        return _trace_inverse(n, qfloat_len, qfloat_ints, qfloat_base, true_division, tensorize,
                              msg_bits or message_bits_for(qfloat_base))
    finally:
        bpa.DIVISION_BITS = saved


def _trace_inverse(n, qfloat_len, qfloat_ints, qfloat_base, true_division, tensorize, msg_bits=MSG_BITS):
    c = Circuit(msg_bits=msg_bits)
    top = 2 * qfloat_base - 1
    arrays = [[c.input(0, top if j == 0 else qfloat_base - 1) for j in range(qfloat_len)] for _ in range(n * n)]
    signs = [c.input(-1, 1) for _ in range(n * n)]
    QFloat.reset_stats()
    out = qfloat_matrix_inverse(arrays, signs, n, qfloat_len, qfloat_ints, qfloat_base, true_division, tensorize)
    c.set_outputs([x for row in out for x in row])
    return c


def compile_inverse(n, qfloat_len, qfloat_ints, qfloat_base=2, true_division=False, tensorize=False, division_bits=None,
                    cache=True):
    """The compiled program of qfloat_matrix_inverse for one configuration: traced, pruned, scheduled and frozen into
    arrays (program.Program) once, then loaded from the on-disk cache (BMI_CACHE_DIR, default <package>/cache) on every
    later start.  The analogue of fhe.Compiler(...).compile (main.py:53-66), whose seconds the reference pays on
    every run.  Returns (program, info); info = {"cached", "seconds", "path"}."""
    key = dict(kind="inverse", n=int(n), len=int(qfloat_len), ints=int(qfloat_ints), base=int(qfloat_base),
               truediv=bool(true_division), tensorize=bool(tensorize), divbits=division_bits or 0,
               msg=message_bits_for(qfloat_base))

    def build():
        from . import base_p_arrays as bpa
        cands = (division_bits,) if division_bits else ((2, 3) if (n <= 4 and qfloat_base == 2) else (2,))
        best = None
        for bits in cands:
            cir = trace_inverse(n, qfloat_len, qfloat_ints, qfloat_base, true_division, tensorize, bits)
            stats = {"additions": QFloat.ADDITIONS, "multiplications": QFloat.MULTIPLICATION, "divisions": QFloat.DIVISION}
            prog = Program.from_circuit(cir, meta=dict(key, division_bits=bits, **stats))
            est = estimated_evaluate_ms(prog)
            if best is None or est < best[0]:
                best = (est, prog)
        del bpa
        return best[1]

    return compile_cached(key, build, cache=cache)
