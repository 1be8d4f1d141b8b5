"""K1-K5 on the GPU through the C ABI vs the oracle (bit-exact: integer / index work)."""
import numpy as np
import pytest
import torch

import co_occ_amd as pkg
from co_occ_amd import _lib
from co_occ_amd._lib import call, ptr
from oracle import cases, native, ref_cpu
import test_oracle as kat

pytestmark = pytest.mark.gpu
I32, F32 = torch.int32, torch.float32


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.to(dtype) if dtype else t


def test_fps_upstream_kat(dev):
    idx = pkg.furthest_point_sample(T(kat.FPS_XYZ, dev), 3)
    assert idx.dtype == I32 and np.array_equal(idx.cpu().numpy(), kat.FPS_EXPECTED)


@pytest.mark.parametrize("n,m,hi", [(5, 5, 3), (37, 20, 4), (700, 300, 5), (1024, 64, 6), (3000, 512, 6),
                                    (9000, 2048, 12), (15000, 2048, 16), (30000, 256, 40)])
def test_fps_ties_match_reference_block_reduction(dev, n, m, hi):
    rng = np.random.default_rng(n)
    pts = rng.integers(0, hi, (2, n, 3)).astype(np.float32)           # integer grid: ties everywhere
    got = pkg.furthest_point_sample(T(pts, dev), m).cpu().numpy()
    assert np.array_equal(got, native.fps(pts, m))


@pytest.mark.parametrize("grid,p,m", [((40, 40, 4), 0.7, 2048), ((100, 100, 8), 0.65, 2048), ((100, 100, 8), 0.12, 2048),
                                      ((33, 21, 5), 0.5, 700), ((200, 200, 16), 0.3, 512), ((7, 5, 3), 1.0, 105)])
def test_fps_voxels_bucketed_equals_reference_rule(dev, grid, p, m):
    """The pruned voxel kernel must select exactly what the reference block reduction selects."""
    X, Y, Z = grid
    g = torch.Generator().manual_seed(X * 7 + Z)
    lin = torch.nonzero(torch.rand(X * Y * Z, generator=g) < p)[:, 0].int()
    n = lin.numel()
    m = min(m, n)
    xyz = torch.stack([lin // (Y * Z), (lin // Z) % Y, lin % Z], 1).float()
    want = native.fps(xyz[None].numpy(), m)[0]
    from co_occ_amd.fuser import _fps_voxels
    got = _fps_voxels(lin.to(dev), grid, m)[0].cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(pkg.furthest_point_sample(xyz[None].contiguous().to(dev), m)[0].cpu().numpy(), want)


def test_fps_voxels_pair_equals_two_single_calls(dev):
    """coocc_fps_voxels_pair (both search directions of BiFuser_N as ONE launch of two workgroups, distance table in registers)
    against two coocc_fps_voxels calls and the reference rule; a grid above 640 buckets reports 2 and launches nothing."""
    from co_occ_amd import _lib
    from co_occ_amd.fuser import _fps_voxels
    X, Y, Z, m = 100, 100, 8, 2048
    g = torch.Generator().manual_seed(77)
    lins = [torch.nonzero(torch.rand(X * Y * Z, generator=g) < p)[:, 0].int().to(dev) for p in (0.12, 0.65)]
    lib = _lib.load()
    wsb = int(lib.coocc_fps_voxels_ws(X, Y, Z))
    ws = [torch.empty(wsb, device=dev, dtype=torch.uint8) for _ in range(2)]
    out = [torch.full((m,), -7, device=dev, dtype=I32) for _ in range(2)]
    rc = call("coocc_fps_voxels_pair", ptr(lins[0]), lins[0].numel(), ptr(out[0]), ptr(ws[0]), ptr(lins[1]), lins[1].numel(), ptr(out[1]),
              ptr(ws[1]), wsb, X, Y, Z, m)
    assert rc in (0, None)
    for lin, o in zip(lins, out):
        single = _fps_voxels(lin, (X, Y, Z), m)[0]
        assert torch.equal(o, single)
        xyz = torch.stack([lin // (Y * Z), (lin // Z) % Y, lin % Z], 1).float().cpu()
        assert np.array_equal(o.cpu().numpy(), native.fps(xyz[None].numpy(), m)[0])
    big = lib.coocc_fps_voxels_pair(ptr(lins[0]), lins[0].numel(), ptr(out[0]), ptr(ws[0]), ptr(lins[1]), lins[1].numel(), ptr(out[1]),
                                    ptr(ws[1]), wsb, 200, 200, 16, m, None)
    assert big == 2


def test_fps_float_coordinates(dev):
    rng = np.random.default_rng(7)
    pts = (rng.integers(-400, 400, (1, 5000, 3)) / 8.0).astype(np.float32)   # exactly representable
    assert np.array_equal(pkg.furthest_point_sample(T(pts, dev), 128).cpu().numpy(), native.fps(pts, 128))


def test_ball_query_upstream_kat(dev):
    xyz, new = T(kat.BQ_XYZ, dev), T(kat.BQ_NEW, dev)
    assert np.array_equal(pkg.ball_query(0, 0.2, 5, xyz, new).cpu().numpy(), kat.BQ_EXPECTED_02)
    assert np.array_equal(pkg.ball_query(0.2, 0.4, 5, xyz, new).cpu().numpy(), kat.BQ_EXPECTED_DILATED)


@pytest.mark.parametrize("n,m,ns,r", [(50, 7, 5, 2.0), (3000, 130, 200, 6.0), (9000, 2048, 200, 6.0), (200, 10, 300, 50.0)])
def test_ball_query_random(dev, n, m, ns, r):
    rng = np.random.default_rng(n + m)
    xyz = rng.integers(0, 30, (2, n, 3)).astype(np.float32)
    ctr = np.concatenate([xyz[:, : m - 2], rng.integers(100, 120, (2, 2, 3)).astype(np.float32)], 1)   # 2 empty balls
    got = pkg.ball_query(0, r, ns, T(xyz, dev), T(ctr, dev)).cpu().numpy()
    assert np.array_equal(got, native.ball_query(0, r, ns, xyz, ctr))


@pytest.mark.parametrize("K", [1, 2, 3, 4, 8])
def test_knn_topk(dev, K):
    rng = np.random.default_rng(K)
    q = rng.integers(0, 40, (300, 3)).astype(np.float32)
    k = rng.integers(0, 40, (5000, 3)).astype(np.float32)
    val = torch.empty(300, K, device=dev, dtype=F32)
    idx = torch.empty(300, K, device=dev, dtype=I32)
    call("coocc_knn_topk", 300, 5000, K, ptr(T(q, dev)), ptr(T(k, dev)), ptr(val), ptr(idx))
    v, i = native.knn_topk(q, k, K)
    assert np.array_equal(idx.cpu().numpy(), i.astype(np.int32))
    assert np.array_equal(val.cpu().numpy(), v)


def test_knn_assign_last_writer(dev):
    rng = np.random.default_rng(3)
    nc, K, ns, nq = 256, 3, 50, 4000
    val = (rng.random((nc, K)) * 20).astype(np.float32)
    nn = rng.integers(0, 9999, (nc, K)).astype(np.int32)
    group = rng.integers(0, nq, (nc, ns)).astype(np.int32)
    winner = torch.empty(K, nq, device=dev, dtype=I32)
    out = torch.empty(K, nq, device=dev, dtype=I32)
    call("coocc_knn_assign", nc, K, ns, nq, 13.3, ptr(T(val, dev)), ptr(T(nn, dev)), ptr(T(group, dev)), ptr(winner), ptr(out))
    assert np.array_equal(out.cpu().numpy(), native.knn_assign(val, nn, group, nq, 13.3).astype(np.int32))


def test_knn_assign_with_ball_query_padding(dev):
    """Groups as ball_query writes them: short groups padded with their FIRST member (ball_query_cuda.cu:47-52) -- the kernel
    skips those repeats (idempotent atomicMax), the result is the reference's."""
    rng = np.random.default_rng(9)
    nc, K, ns, nq = 300, 2, 40, 2500
    val = (rng.random((nc, K)) * 20).astype(np.float32)
    nn = rng.integers(0, 9999, (nc, K)).astype(np.int32)
    group = np.zeros((nc, ns), np.int32)
    for c in range(nc):
        k = int(rng.integers(0, ns + 1))                 # members found; 0: the row stays all zeros (query 0), as the CUDA op leaves it
        if k:
            mem = rng.choice(nq, size=k, replace=False).astype(np.int32)
            group[c, :k] = mem
            group[c, k:] = mem[0]
    winner = torch.empty(K, nq, device=dev, dtype=I32)
    out = torch.empty(K, nq, device=dev, dtype=I32)
    call("coocc_knn_assign", nc, K, ns, nq, 13.3, ptr(T(val, dev)), ptr(T(nn, dev)), ptr(T(group, dev)), ptr(winner), ptr(out))
    assert np.array_equal(out.cpu().numpy(), native.knn_assign(val, nn, group, nq, 13.3).astype(np.int32))


@pytest.mark.parametrize("name", sorted(cases.FUSER_CASES) + sorted(cases.ORACLE_ONLY_CASES))
def test_fps_nn_fast_module_method(dev, name):
    c = dict(cases.FUSER_CASES, **cases.ORACLE_ONLY_CASES)[name]
    img, pts = cases.fuser_inputs(c)
    qi, ki = ref_cpu.voxel_nonzero(pts), ref_cpu.voxel_nonzero(img)
    f = pkg.BiFuser_N(c["C"], c["C"], c["knum"])
    kw = dict(fps_num=2048, radius=6, max_cluster_samples=200, dist_thresh=13.3, num=c["knum"])
    for q, k in ((qi, ki), (ki, qi)):
        if q.shape[0] <= 2048 and c["knum"] > 1:
            with pytest.raises(IndexError):
                f.fps_NN_fast(q.to(dev), k.to(dev), **kw)
            continue
        got = f.fps_NN_fast(q.to(dev), k.to(dev), **kw)
        want = ref_cpu.fps_nn_fast(q, k, **kw)
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)


def test_knn_parts_vs_golden(dev, golden):
    g = golden("knn_parts_k4")
    c = cases.FUSER_CASES["fuser_k4"]
    img, pts = cases.fuser_inputs(c)
    q = ref_cpu.voxel_nonzero(pts)[:, 1:].float().contiguous().to(dev)
    k = ref_cpu.voxel_nonzero(img)[:, 1:].float().contiguous().to(dev)
    ridx = pkg.furthest_point_sample(q[None].contiguous(), 2048)[0]
    assert np.array_equal(ridx.cpu().numpy(), g["repr_idx"])
    rq = q[ridx.long()].contiguous()
    grp = pkg.ball_query(0, 6, 200, q[None].contiguous(), rq[None].contiguous())[0]
    assert np.array_equal(grp.cpu().numpy(), g["group"])
    val = torch.empty(2048, 4, device=dev, dtype=F32)
    idx = torch.empty(2048, 4, device=dev, dtype=I32)
    call("coocc_knn_topk", 2048, k.shape[0], 4, ptr(rq), ptr(k), ptr(val), ptr(idx))
    assert np.array_equal(idx.cpu().numpy(), g["nn"]) and np.array_equal(val.cpu().numpy(), g["val"])


@pytest.mark.parametrize("shape", [(1, 16, 7, 5, 3), (2, 128, 20, 10, 4), (1, 130, 9, 9, 2)])
def test_layout_and_nonzero(dev, shape):
    B, C, X, Y, Z = shape
    g = torch.Generator().manual_seed(C)
    img = torch.randn(shape, generator=g) * (torch.rand(B, 1, X, Y, Z, generator=g) < 0.6)
    pts = torch.relu(torch.randn(shape, generator=g)) * (torch.rand(B, 1, X, Y, Z, generator=g) < 0.2)
    V = X * Y * Z
    cat4 = torch.empty(B * V, 4 * C, device=dev)
    flags = torch.empty(2, B * V, device=dev, dtype=torch.uint8)
    call("coocc_fuser_prepare", ptr(img.to(dev)), ptr(pts.to(dev)), ptr(cat4), ptr(flags[0]), ptr(flags[1]), B, C, V)
    want = torch.cat([img.permute(0, 2, 3, 4, 1), pts.permute(0, 2, 3, 4, 1), torch.zeros(B, X, Y, Z, 2 * C)], -1).reshape(B * V, 4 * C)
    assert torch.equal(cat4.cpu(), want)
    for i, vol in enumerate((img, pts)):
        nz = torch.nonzero(vol.sum(1))
        lin = torch.empty(B * V, device=dev, dtype=I32)
        cnt = torch.empty(1, device=dev, dtype=I32)
        ws = torch.empty(B * V // 1024 + 2, device=dev, dtype=I32)
        call("coocc_compact_flags", ptr(flags[i]), B * V, ptr(lin), ptr(cnt), ptr(ws), ws.numel() * 4)
        n = int(cnt.item())
        assert n == nz.shape[0]
        bxyz = torch.empty(n, 4, device=dev, dtype=torch.int64)
        xyz = torch.empty(n, 3, device=dev)
        call("coocc_lin_to_coords", ptr(lin[:n]), n, X, Y, Z, ptr(xyz), ptr(bxyz))
        assert torch.equal(bxyz.cpu(), nz) and torch.equal(xyz.cpu(), nz[:, 1:].float())
    # generic transposes round-trip
    rows = torch.empty(B * V, C, device=dev)
    call("coocc_ncdhw_to_ndhwc", ptr(img.to(dev)), ptr(rows), B, C, V, C, 0)
    assert torch.equal(rows.cpu(), img.permute(0, 2, 3, 4, 1).reshape(B * V, C))
    back = torch.empty(B, C, V, device=dev)
    call("coocc_ndhwc_to_ncdhw", ptr(rows), ptr(back), B, C, V, C, 0)
    assert torch.equal(back.cpu().view(shape), img)


def test_compaction_property_full_size(dev):
    """BASELINE-size property: 200x200x16 grid, sortedness + count."""
    total = 200 * 200 * 16
    g = torch.Generator().manual_seed(1)
    flags = (torch.rand(total, generator=g) < 0.65).to(torch.uint8).to(dev)
    lin = torch.empty(total, device=dev, dtype=I32)
    cnt = torch.empty(1, device=dev, dtype=I32)
    ws = torch.empty(total // 1024 + 2, device=dev, dtype=I32)
    call("coocc_compact_flags", ptr(flags), total, ptr(lin), ptr(cnt), ptr(ws), ws.numel() * 4)
    n = int(cnt.item())
    assert n == int(flags.sum().item())
    l = lin[:n].long()
    assert bool((l[1:] > l[:-1]).all()) and bool(flags[l].all())


@pytest.mark.parametrize("total,p,offset", [(1, 1.0, 0), (1023, 0.5, 0), (1024, 0.0, 0), (1025, 1.0, 0), (80000, 0.12, 0),
                                            (80000, 0.65, 4), (80000, 0.65, 16), (131072, 0.5, 0), (131073, 0.5, 0), (80001, 0.3, 1),
                                            (640000, 0.65, 0), (1500007, 0.02, 0), (2200001, 0.3, 0)])
def test_compaction_forms_equal_nonzero(dev, monkeypatch, total, p, offset):
    """The compaction (torch.nonzero's replacement) in its three forms -- ONE launch (every earlier block's flags counted by the
    block itself; up to 131072 flags, 16-byte aligned), two launches (counts + write with the prefix summed in the block; up to
    2 M flags), and the count / scan / write form of rounds 1-4 (above, or COOCC_COMPACT_SCAN=1) -- gives nonzero()'s list, count
    and inverse map:
    empty / full / ragged tails, totals on both sides of the one-launch limit, an unaligned flag pointer, flag values > 1."""
    g = torch.Generator().manual_seed(total + offset)
    buf = torch.zeros(total + 32, dtype=torch.uint8)
    buf[offset:offset + total] = (torch.rand(total, generator=g) < p).to(torch.uint8) * torch.randint(1, 255, (total,), generator=g).to(torch.uint8)
    buf[offset + total:] = 7                       # bytes past the end must not be counted
    dbuf = buf.to(dev)
    flags = dbuf[offset:offset + total]
    want = torch.nonzero(buf[offset:offset + total])[:, 0].int()
    want_map = torch.full((total,), -1, dtype=I32)
    want_map[want.long()] = torch.arange(want.numel(), dtype=I32)
    for scan in ("0", "1"):
        monkeypatch.setenv("COOCC_COMPACT_SCAN", scan)
        for with_map in (False, True):
            lin = torch.full((total,), -7, device=dev, dtype=I32)
            cnt = torch.full((1,), -7, device=dev, dtype=I32)
            fmap = torch.full((total,), -7, device=dev, dtype=I32)
            ws = torch.empty(total // 1024 + 2, device=dev, dtype=I32)
            call("coocc_compact_flags_ex", ptr(flags), total, ptr(lin), ptr(cnt), ptr(fmap if with_map else None), ptr(ws),
                 ws.numel() * 4)
            n = int(cnt.item())
            assert n == want.numel()
            assert torch.equal(lin[:n].cpu(), want) and bool((lin[n:] == -7).all())
            if with_map:
                assert torch.equal(fmap.cpu(), want_map)


@pytest.mark.parametrize("grid,pq,pk,K,m", [((40, 40, 4), 0.45, 0.7, 4, 2048), ((100, 100, 8), 0.12, 0.65, 2, 2048),
                                            ((100, 100, 8), 0.65, 0.12, 2, 2048), ((64, 40, 4), 0.65, 0.003, 3, 512),
                                            ((33, 21, 5), 0.5, 0.5, 8, 300), ((50, 50, 1), 0.3, 0.02, 2, 256),
                                            ((128, 128, 10), 0.1, 0.6, 2, 2048)])
def test_grid_topk_and_ball_query_equal_brute_force(dev, grid, pq, pk, K, m):
    """K3 / K4 on a voxel grid (index maps + sorted offset table / window walk) == the brute-force kernels == the oracle,
    bit for bit: dense and sparse key sets (the 0.3 % case leaves most representatives to the brute-force fallback),
    K up to 8, thin grids, the OpenOccupancy grid."""
    from co_occ_amd.fuser import offset_table
    X, Y, Z = grid
    g = torch.Generator().manual_seed(X * 13 + Z + K)
    lin_q = torch.nonzero(torch.rand(X * Y * Z, generator=g) < pq)[:, 0].int()
    lin_k = torch.nonzero(torch.rand(X * Y * Z, generator=g) < pk)[:, 0].int()
    Q, Nk = lin_q.numel(), lin_k.numel()
    assert Nk >= K
    m = min(m, Q)
    xyz = lambda l: torch.stack([l // (Y * Z), (l // Z) % Y, l % Z], 1).float().contiguous()
    q, k = xyz(lin_q).to(dev), xyz(lin_k).to(dev)
    rep = torch.randperm(Q, generator=g)[:m].int().to(dev)
    rq = q[rep.long()].contiguous()
    lq, lk = lin_q.to(dev), lin_k.to(dev)
    maps = torch.empty(2, X * Y * Z, device=dev, dtype=I32)
    call("coocc_voxel_index_map", ptr(lq), Q, X * Y * Z, ptr(maps[0]))
    call("coocc_voxel_index_map", ptr(lk), Nk, X * Y * Z, ptr(maps[1]))
    inv = torch.full((X * Y * Z,), -1, dtype=I32)
    inv[lin_k.long()] = torch.arange(Nk, dtype=I32)
    assert torch.equal(maps[1].cpu(), inv)
    # top-K
    val_b, idx_b = torch.empty(m, K, device=dev, dtype=F32), torch.empty(m, K, device=dev, dtype=I32)
    call("coocc_knn_topk", m, Nk, K, ptr(rq), ptr(k), ptr(val_b), ptr(idx_b))
    off = offset_table(Z, dev)
    val_g, idx_g = torch.empty(m, K, device=dev, dtype=F32), torch.empty(m, K, device=dev, dtype=I32)
    call("coocc_knn_topk_voxels", m, Nk, K, X, Y, Z, ptr(rep), ptr(lq), ptr(maps[1]), ptr(off), off.numel(), ptr(rq), ptr(k),
         ptr(val_g), ptr(idx_g))
    assert torch.equal(idx_g, idx_b) and torch.equal(val_g, val_b)
    o_val, o_idx = native.knn_topk(rq.cpu().numpy(), k.cpu().numpy(), K)
    assert np.array_equal(idx_g.cpu().numpy(), o_idx) and np.array_equal(val_g.cpu().numpy(), o_val)
    # ball query (radius 6, 200 samples: the fuser's call; and a small cap that truncates)
    for radius, ns in ((6.0, 200), (3.0, 7)):
        grp_b = pkg.ball_query(0, radius, ns, q[None].contiguous(), rq[None].contiguous())[0]
        grp_g = torch.empty(m, ns, device=dev, dtype=I32)
        call("coocc_ball_query_voxels", m, 0.0, radius, ns, X, Y, Z, ptr(rep), ptr(lq), ptr(maps[0]), ptr(grp_g))
        assert torch.equal(grp_g, grp_b)


@pytest.mark.parametrize("shape", [(1, 32, 11, 9, 5), (2, 128, 20, 13, 4), (1, 128, 33, 31, 3), (1, 48, 7, 5, 3)])
@pytest.mark.parametrize("img_form,pts_form", [("ncdhw", "ncdhw"), ("rows", "ncdhw"), ("inplace", "ncdhw"), ("rows", "rows"),
                                               ("inplace", "wide")])
def test_fuser_prepare_rows_every_source_form(dev, shape, img_form, pts_form):
    """coocc_fuser_prepare_rows (vector fast path for C % 32 == 0, generic kernel otherwise) against the NCDHW prologue
    coocc_fuser_prepare: same concat rows, same non-empty flags (incl. all-zero rows and rows whose entries cancel exactly),
    for NCDHW sources, dense rows, rows that already sit in their slot, and rows inside a wider buffer; V is never a multiple
    of the 256-voxel workgroup."""
    B, C, X, Y, Z = shape
    g = torch.Generator().manual_seed(C + X)
    img = torch.randn(shape, generator=g) * (torch.rand(B, 1, X, Y, Z, generator=g) < 0.6)
    pts = torch.relu(torch.randn(shape, generator=g)) * (torch.rand(B, 1, X, Y, Z, generator=g) < 0.2)
    img[0, :, 1, 1, 1] = 0.
    img[0, 0, 1, 1, 1], img[0, C - 1, 1, 1, 1] = 3.5, -3.5          # non-zero row with a zero sum: flag 0 as upstream's sum(1)
    V = X * Y * Z
    want = torch.empty(B * V, 4 * C, device=dev)
    wflags = torch.empty(2, B * V, device=dev, dtype=torch.uint8)
    call("coocc_fuser_prepare", ptr(img.to(dev)), ptr(pts.to(dev)), ptr(want), ptr(wflags[0]), ptr(wflags[1]), B, C, V)
    rows_of = lambda t: t.permute(0, 2, 3, 4, 1).reshape(B * V, C).contiguous().to(dev)
    cat4 = torch.full((B * V, 4 * C), 9.0, device=dev)
    flags = torch.full((2, B * V), 7, device=dev, dtype=torch.uint8)
    keep = []

    def source(t, form, slot):
        if form == "ncdhw":
            s = t.to(dev).contiguous(); keep.append(s)
            return ptr(s), 0, 0
        if form == "rows":
            s = rows_of(t); keep.append(s)
            return ptr(s), 1, C
        if form == "inplace":
            cat4[:, slot * C:(slot + 1) * C] = rows_of(t)
            return _lib.c_void_p(cat4.data_ptr() + 4 * slot * C), 1, 4 * C
        wide = torch.full((B * V, C + 8), 5.0, device=dev)            # rows at column offset 4 of a wider buffer
        wide[:, 4:4 + C] = rows_of(t); keep.append(wide)
        return _lib.c_void_p(wide.data_ptr() + 16), 1, C + 8
    a = source(img, img_form, 0)
    b = source(pts, pts_form, 1)
    call("coocc_fuser_prepare_rows", *a, *b, ptr(cat4), ptr(flags[0]), ptr(flags[1]), B, C, V)
    assert torch.equal(cat4, want)
    assert torch.equal(flags, wflags)
    assert int(wflags[0, (1 * Y + 1) * Z + 1]) == 0
