"""Stub: reports that no CUDA device exists so the reference takes its CPU path."""
def is_available():
    return False
gpus = []
def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
