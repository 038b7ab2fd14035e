class Tracer:  # never instantiated in plaintext mode
    pass
