class VectorQuantize:  # placeholder: never instantiated on the hot path
    pass
