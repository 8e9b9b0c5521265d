"""Minimal stand-in for the third-party `munch` package (absent here), used ONLY by
tests/golden/make_golden.py to import the reference in the build container."""


class Munch(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
