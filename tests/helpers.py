"""Comparison helpers shared by the parity tests."""
import numpy as np


def bit_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def count_diff(a, b):
    """number of elements that differ, NaN == NaN, -0 == +0 NOT equal unless bitwise (reported separately)"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype.kind == "f":
        both_nan = np.isnan(a) & np.isnan(b)
        return int(((a != b) & ~both_nan).sum())
    return int((a != b).sum())


def ulp_diff(a, b):
    """max distance in units of the last place between two float32 arrays (NaN==NaN -> 0)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d = np.where(np.isnan(a) & np.isnan(b), 0, d)
    return d


def assert_dict_bit_equal(got, want, prefix=""):
    bad = []
    for k, w in want.items():
        g = got[k]
        if not bit_equal(np.asarray(g), np.asarray(w)):
            bad.append((prefix + k, count_diff(g, w), np.asarray(w).size))
    assert not bad, f"bit mismatches (name, n_diff, n): {bad}"
