"""The C-ABI library loads without a GPU and exports exactly what include/coocc_hip.h declares;
the ctypes table in co_occ_amd/_lib.py agrees with the header (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from co_occ_amd import _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "coocc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct.*?\}\s*\w+;", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:const char\*|int64_t|size_t|int)\s+(coocc_\w+)\s*\(([^)]*)\)\s*;", src):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libcoocc_hip.so not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    fns = header_functions()
    assert len(fns) >= 30
    for name in fns:
        assert hasattr(lib, name), "missing export " + name


def test_ctypes_table_matches_header():
    fns = header_functions()
    assert set(fns) == set(_lib.SIGNATURES), set(fns) ^ set(_lib.SIGNATURES)
    for name, nargs in fns.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_conv_desc_layout_matches_header():
    src = open(os.path.join(ROOT, "include", "coocc_hip.h")).read()
    body = re.search(r"typedef struct coocc_conv_desc \{(.*?)\} coocc_conv_desc;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ptr = "*" in decl
        parts = decl.replace("*", " ").split()
        for n in " ".join(parts[2 if parts[0] == "const" else 1:]).split(","):
            names.append((n.strip(), ptr))
    got = [(n.rstrip("_"), t is ctypes.c_void_p) for n, t in _lib.ConvDesc._fields_]
    assert [n for n, _ in names] == [n for n, _ in got]
    assert [p for _, p in names] == [p for _, p in got]


def test_abi_version_and_error_path():
    lib = _lib.load()
    assert lib.coocc_abi_version() == 1
    # argument validation happens before any launch: exercising it needs no GPU
    rc = lib.coocc_knn_topk(0, 0, 1, None, None, None, None, None)
    assert rc == -1 and b"knn_topk" in lib.coocc_last_error()


def test_round3_entry_points_validate_before_launching():
    """The entry points added in round 3 reject bad arguments with COOCC_EINVAL and a message naming them -- before any launch,
    so this needs no GPU: the fused fine branch (ratio, null pointers), the FPS pair (sizes; grids above 640 buckets report 2 =
    'not taken' without touching the error text), the search driver (null descriptor), the cached-geometry pooling sums."""
    lib = _lib.load()
    i32 = (ctypes.c_int * 3)(8, 8, 4)
    one = ctypes.c_void_p(16)              # a non-null, 16-byte aligned dummy address: validation never dereferences device pointers
    rc = lib.coocc_fine_fused(one, 4, 4, 2, one, 6, 4, 4, one, one, 10, None, 3, i32, one, one, one, 1e-5, one, one, one, one, 1e-5, one, one,
                              17, one, one, None)
    assert rc == -1 and b"fine_fused" in lib.coocc_last_error()
    rc = lib.coocc_fine_fused(None, 4, 4, 2, one, 6, 4, 4, one, one, 10, None, 2, i32, one, one, one, 1e-5, one, one, one, one, 1e-5, one, one,
                              17, one, one, None)
    assert rc == -1 and b"null" in lib.coocc_last_error()
    rc = lib.coocc_fine_fused(one, 4, 4, 2, one, 6, 4, 4, one, one, 10, None, 2, (ctypes.c_int * 3)(9, 8, 4), one, one, one, 1e-5, one, one, one,
                              one, 1e-5, one, one, 17, one, one, None)
    assert rc == -1 and b"final_occ_size" in lib.coocc_last_error()
    assert lib.coocc_fps_voxels_pair(one, 100, one, one, one, 100, one, one, 1 << 20, 200, 200, 16, 64, None) == 2      # too many buckets: not taken
    rc = lib.coocc_fps_voxels_pair(one, 0, one, one, one, 100, one, one, 1 << 20, 100, 100, 8, 64, None)
    assert rc == -1 and b"fps_voxels_pair" in lib.coocc_last_error()
    rc = lib.coocc_fuser_search(None, None, None)
    assert rc == -1 and b"fuser_search" in lib.coocc_last_error()
    rc = lib.coocc_lift_splat_reuse(None, one, 6, 112, 16, 44, 128, 1, 100, 100, 8, one, 128, one, 1 << 30, None)
    assert rc == -1 and b"lift_splat_reuse" in lib.coocc_last_error()


def _wfrag_index(chunk, ngroups, n, kk):
    """csrc/conv3d.hip wfrag_index: [chunk][128-col group][wn][q][lane = 32h + li][4]."""
    g, wn, li, q, h, e = n >> 7, (n >> 5) & 3, n & 31, kk >> 3, (kk >> 2) & 1, kk & 3
    return ((chunk * ngroups + g) * 4 + wn) * 1024 + q * 256 + (h * 32 + li) * 4 + e


def test_pack_weights_host_layout():
    """coocc_conv_pack_weights is host code: fragment-major blocks of 128 columns x 32 k."""
    lib = _lib.load()
    Cout, Cin, taps = 133, 40, 27
    w = np.arange(Cout * Cin * taps, dtype=np.float32).reshape(Cout, Cin, taps)
    n = lib.coocc_conv_pack_weights(w.ctypes.data, Cout, Cin, taps, 0, None)
    assert n == taps * 2 * 256 * 32                       # ceil(40/32) chunks, Cout padded to 256
    packed = np.zeros(n, np.float32)
    lib.coocc_conv_pack_weights(w.ctypes.data, Cout, Cin, taps, 0, packed.ctypes.data)
    for t, n_, c in [(0, 0, 0), (26, 4, 39), (13, 2, 31), (5, 132, 32), (7, 127, 17), (9, 128, 3)]:
        assert packed[_wfrag_index((c // 32) * taps + t, 2, n_, c % 32)] == w[n_, c, t]      # chunk = kc*taps + tap
    assert np.count_nonzero(packed) == np.count_nonzero(w)   # everything else is zero padding
    wl = np.arange(5 * 3 * 8, dtype=np.float32).reshape(5, 3 * 8)                  # Linear(C*K -> Cout), K=3, C=8
    n = lib.coocc_conv_pack_weights(wl.ctypes.data, 5, 8, 3, 1, None)
    packed = np.zeros(n, np.float32)
    lib.coocc_conv_pack_weights(wl.ctypes.data, 5, 8, 3, 1, packed.ctypes.data)
    assert packed[_wfrag_index(2, 1, 4, 7)] == wl[4, 2 * 8 + 7]
