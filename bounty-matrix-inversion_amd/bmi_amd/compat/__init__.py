"""`concrete.fhe`-compatible front end: the reference's code runs AS WRITTEN on this stack.

The reference drives Concrete through five call sites - `fhe.Compiler(fn, {...}).compile(inputset[, configuration])`
(matrix_inversion/main.py:53-66, qfloat_matrix_inversion.py:989-1004), `circuit.keygen()` (main.py:177), `circuit.encrypt`
(main.py:73-76), `circuit.run` (main.py:78-81), `circuit.decrypt` (main.py:83-86), `circuit.simulate` (main.py:107) - and
through the tracer's operator surface inside the traced function (`qfloat.py:11` imports `Tracer`; SURVEY.md section 8b lists
the operators).  `bmi_amd/compat/concrete/fhe` provides exactly that surface on top of this package's circuit IR, compiler
passes, executor and C-ABI engine: `compile` measures the value ranges on the inputset (what Concrete's compiler does), builds
the table look-ups, picks the parameter set whose error budget meets `p_error` / `global_p_error` (default 1e-5, Concrete's;
`bmi_amd/error_budget.py`), and `encrypt / run / decrypt` work on LWE ciphertexts with every look-up a programmable bootstrap on
the MI355X.  There is no CPU fallback for `run`: without the library or a GPU it raises (`circuit.simulate` is the plaintext
path, as in Concrete).

Use: put this directory ahead of any other `concrete` on the import path,

    import bmi_amd.compat; bmi_amd.compat.install()          # or PYTHONPATH=<pkg>/bmi_amd/compat:...
    from concrete import fhe

Supported surface (anything else raises at trace time, never silently):
  Compiler(fn, {"x": "encrypted", ...}).compile(inputset, configuration=None, verbose=False)
  Configuration(p_error=, global_p_error=, security_level=128 | None, ...)   other Concrete options are accepted and ignored
  Circuit.keygen / encrypt / run / decrypt / simulate / encrypt_run_decrypt; PublicArguments, PublicResult
  fhe.zeros, fhe.ones, fhe.univariate, tracing.tracer.Tracer
  on traced values: + - * (ct/ct and ct/const), unary -, // and % by constants, < <= > >= == != (vs constants and ct/ct),
  & | ^, np.abs, np.sign, np.sum(axis), np.concatenate, reshape / flatten, basic and slice indexing, in-place slice
  assignment, .size / .shape, len().
Environment: BMI_COMPAT_BACKEND=simulate makes `run` evaluate in plaintext (fixture generation in a container without a GPU);
BMI_COMPAT_KEY_SEED=<int> selects the seeded TEST-ONLY key generator (default: CSPRNG keys); BMI_COMPAT_Q_BITS=49 | 65 the
ciphertext modulus (default 65: q = 2^64, Concrete's own)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    """puts this directory first on sys.path so that `from concrete import fhe` resolves to the compatible front end, and the
    package root so that the front end finds `bmi_amd`"""
    pkg_root = os.path.dirname(os.path.dirname(HERE))
    for p in (pkg_root, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    stale = [m for m in sys.modules if m == "concrete" or m.startswith("concrete.")]
    for m in stale:
        if not getattr(sys.modules[m], "__file__", "").startswith(HERE):
            del sys.modules[m]
