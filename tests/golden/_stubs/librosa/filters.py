def mel(*a, **k): raise NotImplementedError
