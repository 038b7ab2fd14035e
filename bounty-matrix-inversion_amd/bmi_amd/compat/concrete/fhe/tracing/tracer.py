class Tracer:
    """Base class of everything encrypted in the shim: the reference tests `isinstance(x, Tracer)` (qfloat.py:11)."""
