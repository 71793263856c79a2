"""Per-tensor digest used by the full-size golden fixture (same function as make_golden.py)."""
import numpy as np


def digest(a):
    a = np.asarray(a, np.float64).ravel()
    d = np.zeros(18, np.float64)
    d[0] = np.sqrt((a * a).sum())
    d[1] = a.sum()
    k = min(8, a.size)
    d[2:2 + k] = a[:k]
    d[10:10 + k] = a[-k:]
    return d
