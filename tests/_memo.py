"""Memoisation of the CPU oracle calls inside the GPU suite. Several tests compare DIFFERENT kernel switches against the
SAME fp64 oracle result (a double backward through torch on the host: 5 - 15 s each); the driver gives the whole GPU suite
a fixed wall-clock budget, so an oracle result is computed once per distinct input and reused. The key is a digest of
every argument (tensors by shape, dtype and two checksums), so a changed input can never pick up a stale result."""
import functools

import numpy as np
import torch


def _digest(obj):
    if torch.is_tensor(obj):
        t = obj.detach().double().reshape(-1)
        w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 251.0
        return ("T", tuple(obj.shape), str(obj.dtype), float(t.sum()), float((t * w).sum()))
    if isinstance(obj, np.ndarray):
        return _digest(torch.as_tensor(obj))
    if isinstance(obj, dict):
        return ("D",) + tuple((str(k), _digest(v)) for k, v in sorted(obj.items(), key=lambda kv: str(kv[0])))
    if isinstance(obj, (list, tuple)):
        return ("L",) + tuple(_digest(v) for v in obj)
    return ("R", repr(obj))


def memo_oracle(fn):
    cache = {}

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        key = (_digest(args), _digest(kwargs))
        if key not in cache:
            cache[key] = fn(*args, **kwargs)
        return cache[key]

    return wrapped
