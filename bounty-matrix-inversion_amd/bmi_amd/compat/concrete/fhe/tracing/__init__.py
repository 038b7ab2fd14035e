from . import tracer  # noqa: F401
