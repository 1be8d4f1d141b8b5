"""ctypes binding of oracle/c/coocc_oracle.c (checker only).

Functions mirror the `mmdet3d.ops` call shapes of the reference so they can be
injected into the unmodified reference modules (oracle/refshim.py) and compared
against the HIP path in tests/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcoocc_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "c", "coocc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fps(xyz, m):
    """xyz [B,N,3] f32 -> [B,m] int32 (furthest_point_sample.py:15-35)."""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    out = np.zeros((b, m), np.int32)
    lib().oracle_fps(b, n, m, _p(xyz), _p(out))
    return out


def ball_query(min_radius, max_radius, nsample, xyz, center_xyz):
    """(ball_query.py:14-40) xyz [B,N,3], center_xyz [B,M,3] -> [B,M,nsample] int32."""
    xyz, center_xyz = _f32(xyz), _f32(center_xyz)
    b, n, _ = xyz.shape
    m = center_xyz.shape[1]
    out = np.zeros((b, m, nsample), np.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(min_radius), ctypes.c_float(max_radius),
                            nsample, _p(center_xyz), _p(xyz), _p(out))
    return out


def bev_pool_forward(x, geom, lengths, starts, b, d, h, w):
    """bev_pool_ext.bev_pool_forward (bev_pool.cpp:22-47): sorted x [n,c] -> [b,d,h,w,c]."""
    x = _f32(x)
    geom = np.ascontiguousarray(geom, np.int32)
    lengths = np.ascontiguousarray(lengths, np.int32)
    starts = np.ascontiguousarray(starts, np.int32)
    n, c = x.shape
    out = np.zeros((b, d, h, w, c), np.float32)
    lib().oracle_bev_pool_forward(b, d, h, w, n, c, len(starts), _p(x), _p(geom), _p(starts),
                                  _p(lengths), _p(out))
    return out


def knn_topk(q, key, K):
    """Canonical (d^2, index)-ordered K nearest keys per query row. q [nq,3], key [nk,3]."""
    q, key = _f32(q), _f32(key)
    val = np.zeros((q.shape[0], K), np.float32)
    idx = np.zeros((q.shape[0], K), np.int64)
    rc = lib().oracle_knn_topk(q.shape[0], key.shape[0], K, _p(q), _p(key), _p(val), _p(idx))
    if rc != 0:
        raise ValueError("K larger than the number of keys")
    return val, idx


def knn_assign(val, nn, group, nq, dist_thresh):
    """bifuser_n.py:104-125 with last-writer-wins; returns [K,nq] int64."""
    val = _f32(val)
    nn = np.ascontiguousarray(nn, np.int64)
    group = np.ascontiguousarray(group, np.int32)
    nc, K = val.shape
    out = np.zeros((K, nq), np.int64)
    lib().oracle_knn_assign(nc, K, group.shape[1], nq, ctypes.c_float(dist_thresh), _p(val), _p(nn),
                            _p(group), _p(out))
    return out
