"""Host-side plumbing shared by the module mirrors: voxel-row views, weight packing
(eval-mode BN folded into a per-channel scale/bias), and the implicit-GEMM launcher.

Dense volumes travel between modules as channels-last rows ([B*X*Y*Z, C] fp32); to callers
they look like the reference's [B,C,X,Y,Z] tensors (zero-copy permuted views), so the module
surface stays drop-in while no layout conversion happens between our own modules.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr

_F32 = torch.float32


from ._lib import TIMER, KernelTimer  # noqa: E402,F401


TILE_HINT = int(__import__("os").environ.get("COOCC_CONV_TILE", "0"))   # 0 auto | 128 | 160 (tuning knob)
CONV_V2 = int(__import__("os").environ.get("COOCC_CONV_V2", "1"))       # mirrors csrc/conv3d.hip
# Winograd F(m x m,3x3) over (x,y) for the 3x3x3 stride-1 convs (csrc/winograd.hip): 0 off | 1 on for layers
# with at least WINO_MIN_ROWS output rows (the transforms cost two extra HBM passes and (m+2)^2/9 x the
# weight bytes; the small deep layers are weight-bandwidth-bound and gain nothing)
WINO = int(__import__("os").environ.get("COOCC_WINO", "1"))
WINO_MIN_ROWS = int(__import__("os").environ.get("COOCC_WINO_MIN_ROWS", "8192"))
CONV_PERSIST = __import__("os").environ.get("COOCC_CONV_PERSIST", "1") != "0"   # persistent short-K grouped GEMM (k_conv2p)
ZTRIM = __import__("os").environ.get("COOCC_ZTRIM", "1") != "0"   # drop z taps that only see padding (Z = 1, 2 grids)
WINO_TILE = int(__import__("os").environ.get("COOCC_WINO_TILE", "4"))   # F(4x4,3x3) where X, Y >= 8, else F(2x2,3x3)
# "f32": exact-fp32 MFMA (the parity path, default).  "bf16": the geometric convolutions of C0-C3 round their operands to
# bf16 in LDS and run v_mfma_f32_32x32x16_bf16 (fp32 accumulate / BN epilogue / storage; direct form, no Winograd): the
# reduced-precision path of the OpenOccupancy config (configs[4]).  Set per process or assign core.CONV_DTYPE.
CONV_DTYPE = __import__("os").environ.get("COOCC_CONV_DTYPE", "f32")
# How the fp32 convolutions of CONV_DTYPE == "f32" are evaluated:  "h2" (default): operands split into two f16 halves
# (hi + lo * 2^-11), three v_mfma_f32_32x32x16_f16 per step, fp32 accumulation (csrc/gemm_h2.hip) -- fp32-accurate (measured
# error vs fp64 is half that of the fp32-MFMA chain) at up to 5.3x the fp32-MFMA rate; "f32": v_mfma_f32_32x32x2_f32 everywhere.
CONV_ENGINE = __import__("os").environ.get("COOCC_CONV_ENGINE", "h2")
# operand scale of the Winograd-domain V in the h2 engine (keeps B^T d B inside the f16 range: |activation| < 65504 / (amp * scale),
# amp = 100 for F(4x4), 25 for F(3x3), 4 for F(2x2))
H2_WINO_SCALE = {2: 0.5, 3: 0.25, 4: 0.125}
H2_DIRECT = __import__("os").environ.get("COOCC_H2_DIRECT", "1") != "0"     # stride-1 3x3xkz layers outside the Winograd path
# 1: split-K layers of the split-f16 engine reduce in-kernel (arrival counters; the last workgroup of a tile sums the slabs in slice
# order and runs the epilogue): no k_conv_reduce launch, same bits.  MEASURED SLOWER and off by default: the last-arriving
# workgroup of a tile reads splitk (up to 32) slabs with 16 tiles' worth of parallelism where the second-pass kernel spreads the
# same reads over the whole chip -- k_gemm_h2z<1,true> 24 -> 168 us with device-scope loads (200 us with __threadfence()), against
# a 5 us k_conv_reduce launch (profiles/r4_dense_stage_kernels.txt).  The second pass instead writes the H2 twin.
INKERNEL_REDUCE = __import__("os").environ.get("COOCC_INKERNEL_REDUCE", "0") != "0"
# Direct layers below this many flops stay on the fp32-MFMA kernels.  1e9 through round 5 ("the input split + split-K reduce launches cost
# more than the faster GEMM saves"); since the producers write the H2 twins in their epilogues there is no input split left, and the
# stride-2 1x1x1 downsamples, the 1 250- / 169-row laterals and pyramid-top output / OccHead layers (0.09-0.66 GF, 20-43 us each as
# fp32-MFMA launches at 27-160 workgroups) take 15-25 us on the split-f16 kernels: dense stage 3.669 -> 3.575 ms
# (profiles/r6_h2_min_flops.txt: 1e9 / 3e8 / 1e8 / 5e7 / 0).  Below 5e7 only the 64 -> 4 soft-weight head is left, whose N pads to 128.
H2_DIRECT_MIN_FLOPS = float(__import__("os").environ.get("COOCC_H2_MIN_FLOPS", "5e7"))


# Independent branches of ONE sample's dense stage on side streams (round 6; OFF by default).  The stage is a chain of ~100 dependent
# launches, a third of which occupy a fraction of the chip (the 25x25x2 / 13x13x1 pyramid levels, the image branch of the fine head:
# 27-512 workgroups).  With COOCC_BRANCHES=1 what does not depend on the chain is forked onto a side stream and joined where it is
# consumed -- the render block, the image branch of the fine head, pyramid levels 1-3 of the FPN / OccHead convolutions beside
# level 0's -- so a captured graph has parallel branches.  Same kernels on the same operands: same bits (the graph / serving /
# co-runner tests pass either way).  MEASURED (profiles/r6_graph_branches.txt): one graph alone 3.619 -> 3.528 ms per replay
# (-2.5 %: the overlapped kernels slow each other, 3.61 -> 4.29 ms of kernel time), but with several graphs in flight -- the
# serving loop -- every branch is one more hardware queue per replay and the loop collapses: 2 / 3 / 4 graphs in flight 335 / 316 /
# 351 -> 155 / 261 / 254 samples/s, pipeline 274 -> 188.  The other samples' graphs already fill the gaps the branches aim at.
BRANCHES = __import__("os").environ.get("COOCC_BRANCHES", "0") != "0"
_branch_streams = {}


class Fork:
    """``with Fork(i) as b: <launches on side stream i of the current stream>`` ... ``b.join()`` (on the forking stream) before
    anything reads the branch's results.  Inactive (the body runs in place) when ``BRANCHES`` is off or kernel timing is on."""

    def __init__(self, idx, device=None, enable=True):
        self.on = bool(BRANCHES and enable and not TIMER.enabled)
        self.idx, self.device, self._ctx, self.main, self.side = idx, device, None, None, None

    def __enter__(self):
        if self.on:
            dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
            self.main = torch.cuda.current_stream(dev)
            key = (dev.index, self.main.cuda_stream, self.idx)
            side = _branch_streams.get(key)
            if side is None:
                side = _branch_streams[key] = torch.cuda.Stream(device=dev)
            self.side = side
            side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def join(self, *tensors):
        """The forking stream waits for the branch; ``tensors``: results allocated inside the branch that the forking stream goes
        on to use (recorded on it, so the caching allocator does not hand their memory out while it still reads them)."""
        if self.on and self.side is not None:
            cur = torch.cuda.current_stream(self.side.device)
            cur.wait_stream(self.side)
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
            self.side = None


def branch_streams_of(device, stream):
    """The side streams ``Fork`` made for ``stream`` (a captured graph pins their scratch buffers too)."""
    return [s for (d, h, _), s in _branch_streams.items() if d == device.index and h == stream.cuda_stream]


def conv_kernel_name(M, Cout, table, hint=0, iters=1 << 30, one_by_one=False):
    """Mirror of the tile-configuration rule in csrc/conv3d.hip (coocc_conv_fwd).  iters = taps * ceil(Cin/32);
    one_by_one: a 1x1x1 stride-1 layer (takes the persistent kernel when it has more than 768 tiles)."""
    if (one_by_one and CONV_PERSIST and CONV_V2 and not table and Cout > 64 and M >= 8192 and iters <= 24
            and -(-M // 128) * -(-Cout // 128) > 768):
        return "k_conv2p<1x1>"
    hint = hint or TILE_HINT
    if Cout <= 32:
        t = "128,32,32,32"
    elif Cout <= 64:
        t = "128,64,32,64"
    elif M < 8192:
        if M >= 512 and not table and CONV_V2:
            return "k_conv2<128>"
        t = "64,128,32,64"
    else:
        def util(bm):
            tiles = -(-M // bm) * (-(-Cout // 128))
            return tiles * bm / float(-(-tiles // 512) * 512)
        big = hint == 160 or (hint == 0 and util(160) > util(128) * 1.02)
        if iters <= 24 and hint != 160:
            big = False        # short K: 128-row tiles at 3 workgroups per CU
        if not table and CONV_V2:
            return "k_conv2<%d>" % (160 if big else 128)      # software-pipelined large-layer kernel
        t = "160,128,160,32" if big else "128,128,64,64"
    return "k_conv<%s,%s>" % (t, "table" if table else "geom")


class Rows:
    """A dense voxel volume as channels-last rows: t[B*X*Y*Z, stride], C channels at `coff`."""

    __slots__ = ("t", "B", "X", "Y", "Z", "C", "coff", "h16", "h2", "persistent", "aux")

    def __init__(self, t, B, X, Y, Z, C, coff=0, persistent=False):
        self.t, self.B, self.X, self.Y, self.Z, self.C, self.coff = t, B, X, Y, Z, C, coff
        # rows of a buffer that this package's kernels REWRITE in place through raw pointers (sample slots of the serving loop,
        # ``out=`` targets): torch's version counter does not see those writes, so reference-layout views of such rows carry no
        # back-reference (``as_ncdhw`` / ``to_rows``) and a later consumer converts / re-splits instead of trusting a cached twin
        self.persistent = persistent
        self.aux = None       # by-products a producer made from these rows for a known consumer (OccHead: the fine branch's Q rows)
        self.h16 = None       # f16 twin [B*V, C] written by the producing convolution's epilogue (CONV_DTYPE == "f16")
        # H2 twin [B*V, C] (split-f16 operand rows, csrc/h2_rows.h) of these rows: written by the PRODUCER's epilogue when the
        # consumer is a split-f16 layer outside the Winograd path (``conv_rows(twin_for=...)``), else made once by
        # ``h2_rows`` and shared by every consumer.  Any in-place writer of ``t`` must refresh or drop it.
        self.h2 = None

    @property
    def stride(self):
        return self.t.shape[1]

    @property
    def V(self):
        return self.X * self.Y * self.Z

    def data(self):
        """ctypes pointer to channel 0 of row 0."""
        return ctypes.c_void_p(self.t.data_ptr() + 4 * self.coff)

    def as_ncdhw(self):
        """Reference-layout view [B,C,X,Y,Z] (no copy).  The view remembers these Rows (and the tensor version they were
        valid at), so handing it to the next module of this package finds the 16-bit twins again (``to_rows``)."""
        v = self.t.view(self.B, self.X, self.Y, self.Z, self.stride)
        if self.coff or self.stride != self.C:
            v = v[..., self.coff:self.coff + self.C]
        v = v.permute(0, 4, 1, 2, 3)
        if not self.persistent:
            v._coocc_rows = (self, v._version)
        return v


def to_rows(x):
    """[B,C,X,Y,Z] tensor (any strides) -> Rows, converting with the HIP transpose if needed."""
    if isinstance(x, Rows):
        return x
    back = getattr(x, "_coocc_rows", None)
    if back is not None:
        # the zero-copy view of Rows this package produced, untouched since (any in-place torch op bumps the version counter
        # the view shares with its base): the same Rows, 16-bit twins included
        r, ver = back
        if ver == x._version and x.dim() == 5 and tuple(x.shape) == (r.B, r.C, r.X, r.Y, r.Z):
            return r
    if x.dim() != 5:
        raise ValueError("expected a [B,C,X,Y,Z] tensor")
    if not x.is_cuda:
        raise _lib.CooccError("co_occ_amd modules run on the GPU only (no CPU fallback)")
    B, C, X, Y, Z = x.shape
    x = x.float()
    cl = x.permute(0, 2, 3, 4, 1)
    if cl.is_contiguous():
        return Rows(cl.reshape(B * X * Y * Z, C), B, X, Y, Z, C)
    x = x.contiguous()
    out = torch.empty(B * X * Y * Z, C, device=x.device, dtype=_F32)
    call("coocc_ncdhw_to_ndhwc", ptr(x), ptr(out), B, C, X * Y * Z, C, 0)
    return Rows(out, B, X, Y, Z, C)


def rows_to_ncdhw_contiguous(r):
    """Materialise the reference memory layout (only needed by callers that insist on it)."""
    out = torch.empty(r.B, r.C, r.X, r.Y, r.Z, device=r.t.device, dtype=_F32)
    call("coocc_ndhwc_to_ncdhw", ptr(r.t), ptr(out), r.B, r.C, r.V, r.stride, r.coff)
    return out


# ------------------------------------------------------------------ weights
def fold_bn(bn, conv_bias=None):
    """Eval-mode BatchNorm -> (scale, bias) in fp64, returned fp32."""
    var = bn.running_var.double()
    scale = torch.rsqrt(var + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight.double()
    bias = -bn.running_mean.double() * scale
    if bn.bias is not None:
        bias = bias + bn.bias.double()
    if conv_bias is not None:
        bias = bias + conv_bias.double() * scale
    return scale.float(), bias.float()


class PackedConv:
    """A conv/linear layer in the layout coocc_conv_fwd consumes (+ folded norm)."""

    def __init__(self, weight, bn=None, bias=None, ksize=1, stride=1, pad=0, tap_major=False, taps=None):
        w = weight.detach().float().cpu().contiguous()
        self.Cout = w.shape[0]
        if taps is None:
            taps = ksize ** 3
        if tap_major:
            self.Cin = w.shape[1] // taps
        else:
            self.Cin = w.shape[1]
            w = w.reshape(self.Cout, self.Cin, -1)
            assert w.shape[2] == taps, "weight does not have ksize^3 taps"
        self.taps, self.ksize, self.stride, self.pad = taps, ksize, stride, pad
        # raw weights kept on the host for the lazily built Winograd packs
        self._w_raw = w if (ksize == 3 and stride == 1 and pad == 1 and not tap_major and taps == 27) else None
        # cubic 3x3x3 weights also kept for the z-trimmed packs (conv_rows: taps that only ever see z padding)
        self._w_cube = w.view(self.Cout, self.Cin, 3, 3, 3) if (ksize == 3 and not tap_major and taps == 27) else None
        self._ztrim = {}
        self._wino = {}
        self._bf16 = {}
        # [Cout, Cin, taps] on the host: source of the bf16 / f16 packs (tap-major weights are [Cout, taps * Cin])
        self._w_taps = w if not tap_major else w.view(self.Cout, taps, self.Cin).permute(0, 2, 1).contiguous()
        self.wino_tile = None        # per-layer override of WINO_TILE (2 | 3 | 4)
        lib = _lib.load()
        n = lib.coocc_conv_pack_weights(ctypes.c_void_p(w.data_ptr()), self.Cout, self.Cin, taps, int(tap_major), None)
        packed = torch.empty(n, dtype=_F32)
        lib.coocc_conv_pack_weights(ctypes.c_void_p(w.data_ptr()), self.Cout, self.Cin, taps, int(tap_major),
                                    ctypes.c_void_p(packed.data_ptr()))
        dev = weight.device
        self.w = packed.to(dev)
        if bn is not None:
            s, b = fold_bn(bn, bias)
            self.scale, self.bias = s.to(dev).contiguous(), b.to(dev).contiguous()
        else:
            self.scale = None
            self.bias = bias.detach().float().to(dev).contiguous() if bias is not None else None


    def ztrim_pack(self, lo, hi):
        """Pack of the z taps lo..hi only (3 x 3 x (hi-lo+1) kernel): the other z taps read nothing but padding
        for every output voxel of the grid this is called for, so dropping them is exact."""
        if (lo, hi) not in self._ztrim:
            w = self._w_cube[:, :, :, :, lo:hi + 1].contiguous().view(self.Cout, self.Cin, -1)
            lib = _lib.load()
            n = lib.coocc_conv_pack_weights(ctypes.c_void_p(w.data_ptr()), self.Cout, self.Cin, w.shape[2], 0, None)
            packed = torch.empty(n, dtype=_F32)
            lib.coocc_conv_pack_weights(ctypes.c_void_p(w.data_ptr()), self.Cout, self.Cin, w.shape[2], 0,
                                        ctypes.c_void_p(packed.data_ptr()))
            self._ztrim[(lo, hi)] = packed.to(self.w.device)
        return self._ztrim[(lo, hi)]

    def bf16_pack(self, ztrim=None):
        """bf16 pack of k_conv_bf16w (fragment-major, RNE), for all taps or, with
        ``ztrim=(lo, hi)``, for the z taps lo..hi of a 3x3x3 kernel only.  None when the layer cannot take that kernel."""
        if self._w_taps is None or self.Cin % 64:
            return None
        if ztrim not in self._bf16:
            w = self._w_taps
            if ztrim is not None:
                w = self._w_cube[:, :, :, :, ztrim[0]:ztrim[1] + 1].reshape(self.Cout, self.Cin, -1)
            taps = w.shape[2]
            npad = -(-self.Cout // 128) * 128
            wp = torch.zeros(npad, self.Cin, taps, dtype=_F32)
            wp[:self.Cout] = w
            # fragment-major: [(chunk, tap)][Npad/32][4 k-steps][2 lane halves][32 columns][8 k] (k = 16 s + 8 h + e)
            pack = wp.view(npad // 32, 32, self.Cin // 64, 4, 2, 8, taps).permute(2, 6, 0, 3, 4, 1, 5)
            self._bf16[ztrim] = pack.contiguous().to(torch.bfloat16).to(self.w.device)
        return self._bf16[ztrim]

    @staticmethod
    def _h2_layout(w64):
        """[Npad, Cin, taps] fp64 -> H2 pack [(chunk, tap)][Npad/32][2 k16 steps][hi | lo][64 lanes][8 f16] (csrc/gemm_h2.hip):
        lane l of step s holds k = 32 chunk + 16 s + 8 (l >> 5) + 0..7 of column 32 nt + (l & 31)."""
        npad, cin, taps = w64.shape
        if w64.numel() and float(w64.abs().max()) >= 32768.0:
            raise _lib.CooccRangeError("split-f16 engine: a weight of magnitude %.3g does not fit the f16 operand range (|w| < 32768); "
                                  "set COOCC_CONV_ENGINE=f32 for this model" % float(w64.abs().max()))
        hi = w64.to(torch.float16)
        lo = ((w64 - hi.double()) * 2048.0).to(torch.float16)
        planes = torch.stack([hi, lo], 0)                                     # [pl, n, c, t]
        v = planes.view(2, npad // 32, 32, cin // 32, 2, 2, 8, taps)            # pl, nt, li, chunk, s, hf, e, t
        return v.permute(3, 7, 1, 4, 0, 5, 2, 6).contiguous()                  # chunk, t, nt, s, pl, hf, li, e

    def h2_pack(self, ztrim=None):
        """H2 (split-f16) pack of the direct form: all taps or, with ``ztrim=(lo, hi)``, the z taps lo..hi of a 3x3x3 kernel.
        None when the layer cannot take the h2 kernel (Cin % 32)."""
        if self._w_taps is None or self.Cin % 32:
            return None
        key = ("h2", ztrim)
        if key not in self._bf16:
            w = self._w_taps
            if ztrim is not None:
                w = self._w_cube[:, :, :, :, ztrim[0]:ztrim[1] + 1].reshape(self.Cout, self.Cin, -1)
            npad = -(-self.Cout // 128) * 128
            wp = torch.zeros(npad, self.Cin, w.shape[2], dtype=torch.float64)
            wp[:self.Cout] = w.double()
            self._bf16[key] = self._h2_layout(wp).to(self.w.device)
        return self._bf16[key]

    def h1_pack(self, ztrim=None):
        """One-term f16 pack (mfma_dtype 4): [(Cin/64 chunk, tap)][Npad/32][4 k16 steps][64 lanes][8 f16], RNE; all taps or the z
        taps lo..hi of a 3x3x3 kernel.  None when Cin % 64."""
        if self._w_taps is None or self.Cin % 64:
            return None
        key = ("h1", ztrim)
        if key not in self._bf16:
            w = self._w_taps
            if ztrim is not None:
                w = self._w_cube[:, :, :, :, ztrim[0]:ztrim[1] + 1].reshape(self.Cout, self.Cin, -1)
            taps = w.shape[2]
            npad = -(-self.Cout // 128) * 128
            wp = torch.zeros(npad, self.Cin, taps, dtype=_F32)
            wp[:self.Cout] = w
            # nt, li, chunk, s, hf, e, t -> chunk, t, nt, s, hf, li, e        (k = 64 chunk + 16 s + 8 hf + e)
            pack = wp.view(npad // 32, 32, self.Cin // 64, 4, 2, 8, taps).permute(2, 6, 0, 3, 4, 1, 5)
            self._bf16[key] = pack.contiguous().to(torch.float16).to(self.w.device)
        return self._bf16[key]

    def wino_h2_pack(self, tile):
        """(tile+2)^2 H2 packs of U[p][dz] = (G g G^T)[xi][eta][dz] (taps = 3), split from the fp64 products."""
        key = ("h2", tile)
        if key not in self._wino:
            G = self._wino_G(tile)
            n2 = G.shape[0] ** 2
            w = self._w_raw.double().view(self.Cout, self.Cin, 3, 3, 3)
            U = torch.einsum("pa,qb,ncabz->pqncz", G, G, w).reshape(n2, self.Cout, self.Cin, 3)
            npad = -(-self.Cout // 128) * 128
            Up = torch.zeros(n2, npad, self.Cin, 3, dtype=torch.float64)
            Up[:, :self.Cout] = U
            self._wino[key] = torch.stack([self._h2_layout(Up[p]) for p in range(n2)], 0).to(self.w.device)
        return self._wino[key]

    @staticmethod
    def _wino_G(tile):
        if tile == 2:
            return torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)
        if tile == 3:
            return torch.tensor([[1, 0, 0], [-2 / 9, 2 / 9, -2 / 9], [1 / 9, 2 / 9, 4 / 9], [-8 / 9, -4 / 9, -2 / 9], [0, 0, 1]],
                                dtype=torch.float64)
        return torch.tensor([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                             [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], dtype=torch.float64)

    def wino_pack(self, tile):
        """(tile+2)^2 packs (one per transform point p = (tile+2)*xi + eta) of U[p][dz] = (G g G^T)[xi][eta][dz],
        taps = 3 (z).  G: F(2,3) / F(4,3) Toom-Cook matrices, products in fp64."""
        if tile not in self._wino:
            if tile == 2:
                G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)
            elif tile == 3:
                # Toom-Cook points (0, -1, 2, 1/2, inf), matching Wino<5>
                G = torch.tensor([[1, 0, 0], [-2 / 9, 2 / 9, -2 / 9], [1 / 9, 2 / 9, 4 / 9], [-8 / 9, -4 / 9, -2 / 9], [0, 0, 1]],
                                 dtype=torch.float64)
            else:
                # Toom-Cook points (0, 1, -1, 1/2, -2, inf), matching Wino<6> in csrc/winograd.hip
                G = torch.tensor([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                                  [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], dtype=torch.float64)
            n2 = G.shape[0] ** 2
            w = self._w_raw.double().view(self.Cout, self.Cin, 3, 3, 3)                 # [n, c, kx, ky, kz]
            U = torch.einsum("pa,qb,ncabz->pqncz", G, G, w).reshape(n2, self.Cout, self.Cin, 3).float().contiguous()
            lib = _lib.load()
            n = lib.coocc_conv_pack_weights(ctypes.c_void_p(U[0].data_ptr()), self.Cout, self.Cin, 3, 0, None)
            packed = torch.empty(n2, n, dtype=_F32)
            for p in range(n2):
                lib.coocc_conv_pack_weights(ctypes.c_void_p(U[p].data_ptr()), self.Cout, self.Cin, 3, 0,
                                            ctypes.c_void_p(packed[p].data_ptr()))
            self._wino[tile] = packed.to(self.w.device)
        return self._wino[tile]


_ws_cache = {}
_wino_ws = {}
_bf16_ws = {}
# bf16 path: 1 = round the activations to bf16 in memory once per layer and run k_conv_bf16w (operands staged by
# global_load_lds); 0 = k_conv_bf16 (fp32 operands rounded inside the K loop)
BF16_PRECONVERT = __import__("os").environ.get("COOCC_BF16_PRECONVERT", "1") != "0"
ZSHARE = __import__("os").environ.get("COOCC_BF16_ZSHARE", "1") != "0"       # read by the library too (k_conv_bf16z)


def _bf16_buffer(device, n):
    key = (device.index, _lib.stream(device).value)
    t = _bf16_ws.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(n, device=device, dtype=torch.bfloat16)
        _bf16_ws[key] = t
    return t


def _wino_buffer(device, kind, nfloats):
    key = (device.index, kind, torch.cuda.current_stream(device).cuda_stream)
    t = _wino_ws.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.zeros(nfloats, device=device, dtype=_F32)     # padded rows stay zero
        _wino_ws[key] = t
    return t


def _lcm(a, b):
    import math
    return a * b // math.gcd(a, b)


def h2_capable(pc):
    """The split-f16 engine takes this layer's Winograd-domain GEMM (inference packs only: the training path re-packs its
    weights on the device every step and stays on the fp32-MFMA kernels)."""
    return CONV_ENGINE == "h2" and pc.Cin % 32 == 0 and hasattr(pc, "wino_h2_pack")


def wino_plan(x, pc, M, res_mode):
    """None, or (tile, points, Tx, Ty, rows, G, tile_hint) for the Winograd path of this layer."""
    return _wino_plan_geom(x.B, x.X, x.Y, x.Z, pc, M, res_mode)


def _wino_plan_geom(B, X, Y, Z, pc, M, res_mode):
    if not WINO or pc._w_raw is None or M < WINO_MIN_ROWS or res_mode not in (0, 1) or pc.Cin % 4:
        return None
    tile = pc.wino_tile or WINO_TILE
    if tile not in (2, 3, 4) or min(X, Y) < 2 * tile:
        tile = 2
    Tx, Ty = -(-X // tile), -(-Y // tile)
    rows = B * Tx * Ty * Z
    # group rows: a multiple of the GEMM's M tile and of Z.  640 = lcm(128, 160) leaves the tile choice to the
    # kernel; when that would pad much more than a 128-row granule (50x50x4: 676 -> 1280 vs 768), pin 128-row tiles
    g640, g128 = _lcm(640, Z), _lcm(128, Z)
    G640, G128 = -(-rows // g640) * g640, -(-rows // g128) * g128
    G, hint = (G640, 0) if G640 <= 1.05 * G128 else (G128, 128)
    if h2_capable(pc):
        g256 = _lcm(256, Z)           # 256-row tiles of the persistent split-f16 GEMM (k_gemm_h2p)
        G, hint = -(-rows // g256) * g256, 0
    pts = (tile + 2) ** 2
    if pts * G >= 1 << 31:
        return None     # row indices are 32-bit
    return tile, pts, Tx, Ty, rows, G, hint


def wino_eligible(x, pc, M, res_mode):
    return wino_plan(x, pc, M, res_mode) is not None


def scratch(device, kind, nfloats):
    """Per-stream scratch buffer (uninitialised), grown on demand."""
    key = (device.index, "s:" + kind, torch.cuda.current_stream(device).cuda_stream)
    t = _wino_ws.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(nfloats, device=device, dtype=_F32)
        _wino_ws[key] = t
    return t


_sem_ws = {}


def tile_sem(device, n=4096):
    """Per-stream arrival counters of the in-kernel split-K reduction (coocc_conv_desc.tile_sem): zero on entry, left zero by
    every launch, so one buffer serves every layer issued on the stream."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _sem_ws.get(key)
    if t is None:
        t = _sem_ws[key] = torch.zeros(n, device=device, dtype=torch.int32)
    return t


def stream_scratch(device, stream):
    """Every per-stream scratch tensor of ``stream`` (Winograd V / M, split-K slabs, H2 inputs, arrival counters): a captured
    hipGraph holds raw pointers into them and must keep them alive (``graph.DenseGraph``)."""
    h = stream.cuda_stream
    out = [t for k, t in _wino_ws.items() if k[0] == device.index and k[-1] == h]
    out += [t for k, t in _ws_cache.items() if k[0] == device.index and k[-1] == h]
    out += [t for k, t in _bf16_ws.items() if k[0] == device.index and k[-1] == h]
    out += [t for k, t in _sem_ws.items() if k[0] == device.index and k[-1] == h]
    return out


def h2_rows(x):
    """The H2 twin of Rows ``x`` (whole rows: coff 0): the producer's, or one conversion pass whose result every later consumer
    of the same Rows shares (a block's output feeds the strided conv1, the 1x1x1 downsample and the FPN lateral)."""
    if x.h2 is None:
        assert x.C % 32 == 0
        xh = torch.empty(x.B * x.V, x.C, device=x.t.device, dtype=_F32)
        call("coocc_rows_to_h2", x.data(), x.stride, x.B * x.V, x.C, 1.0, ptr(xh))
        x.h2 = xh
    return x.h2


def check_h2_overflow(reset=True):
    """Raise if a kernel of the split-f16 engine wrote a 16-bit operand outside its guarded range since the last check
    (csrc/h2_rows.h h2_guard: |v| >= 32768 after the writer's scale, or NaN).  Call after the stream(s) have been synchronised
    -- the detector does at its own host reads (the fine-branch count, the metrics) and ``serving`` when a result is fetched."""
    fault = _lib.load().coocc_device_fault(1 if reset else 0)
    if fault:
        raise _lib.CooccError("libcoocc_hip: a kernel met an index outside the buffer it addresses and skipped the access (fault code %d: "
                              "%s) -- the outputs of this sample are not trustworthy" % (
                                  fault, {1: "coocc_sparse_tap_sum read a voxel -> ordinal map entry past the rows it was sized for "
                                             "(a stale or corrupted map)"}.get(fault, "unknown")))
    if _lib.load().coocc_h2_overflow(1 if reset else 0):
        raise _lib.CooccRangeError(
            "split-f16 engine: an activation left the f16 operand range (|x| >= 32768, or %g for the F(4x4) Winograd layers whose "
            "transform amplifies by up to 100 at scale 1/8; NaN counts) -- the outputs of this sample are not trustworthy.  "
            "Set COOCC_CONV_ENGINE=f32 (exact-fp32 MFMA kernels, no range limit; ~1.8x slower) or rescale the inputs "
            "(INTEGRATION.md, 'Operand range of the split-f16 engine')." % (32768.0 / (100 * H2_WINO_SCALE[4])))


def route(B, X, Y, Z, pc, rm=0, splitk=0):
    """Which kernel family ``conv_rows`` takes for layer ``pc`` on an input grid [B, X, Y, Z]: "wino" | "h2" | "f16" | "other".
    One rule for the dispatch itself and for the producers that decide whether to write an H2 twin for their consumer."""
    Xo, Yo, Zo = out_dim(X, pc.ksize, pc.stride, pc.pad), out_dim(Y, pc.ksize, pc.stride, pc.pad), out_dim(Z, pc.ksize, pc.stride, pc.pad)
    M = B * Xo * Yo * Zo
    bf16, f16 = CONV_DTYPE == "bf16", CONV_DTYPE == "f16"
    if not (bf16 or f16) and _wino_plan_geom(B, X, Y, Z, pc, M, rm) is not None:
        return "wino"
    taps = pc.taps
    if ZTRIM and pc._w_cube is not None:
        ok = [kz for kz in range(3) if any(0 <= zo * pc.stride - pc.pad + kz < Z for zo in range(Zo))]
        if ok[-1] - ok[0] + 1 < 3:
            taps = 9 * (ok[-1] - ok[0] + 1)
    if f16 and pc.Cin % 64 == 0 and pc._w_taps is not None and rm in (0, 1) and splitk in (0, 1):
        return "f16"
    if (not bf16 and not f16 and CONV_ENGINE == "h2" and H2_DIRECT and pc.Cin % 32 == 0 and pc._w_taps is not None
            and rm in (0, 1) and 2.0 * M * pc.Cin * pc.Cout * taps >= H2_DIRECT_MIN_FLOPS):
        return "h2"
    return "other"


def takes_h2(rows, consumers):
    """True when one of the layers ``consumers`` (PackedConv or None) reads ``rows`` (a Rows, or a (B, X, Y, Z) grid) as H2 rows."""
    g = (rows.B, rows.X, rows.Y, rows.Z) if isinstance(rows, Rows) else rows
    return any(pc is not None and route(*g, pc) == "h2" for pc in consumers)


def conv_rows_wino(x, pc, out, relu, res, plan, in_ranges=None, twin=False):
    """3x3x3 stride-1 conv as Winograd F(m x m,3x3) over (x,y) + direct z taps: input transform, one grouped
    GEMM launch (one weight pack per transform point), output transform with the epilogue.
    ``in_ranges``: [(channel offset, count), ...] inside x's rows whose concatenation is the layer's input (sum = pc.Cin);
    default: the first pc.Cin channels of x."""
    dev = x.t.device
    tile, pts, Tx, Ty, rows, G, hint = plan
    V = _wino_buffer(dev, "V", pts * G * pc.Cin)
    Mb = _wino_buffer(dev, "M", pts * G * pc.Cout)
    h2 = h2_capable(pc) and all(c % 32 == 0 for _, c in (in_ranges or []))
    wp = pc.wino_h2_pack(tile) if h2 else pc.wino_pack(tile)
    vscale = H2_WINO_SCALE[tile]
    sdev = getattr(pc, "operand_scale_dev", None) if h2 else None         # training's dgrad: {scale, 1 / scale} of the gradient rows (device)
    with TIMER.region("k_wino_in", 4.0 * x.V * pc.Cin + 4.0 * pts * rows * pc.Cin):
        if in_ranges is None and not h2:
            call("coocc_wino_input", x.data(), x.stride, x.B, x.X, x.Y, x.Z, pc.Cin, tile, ptr(V), G)
        else:
            assert in_ranges is None or sum(c for _, c in in_ranges) == pc.Cin
            voff = 0
            for coff, cr in (in_ranges or [(0, pc.Cin)]):
                src = _lib.DevPtr(x.t.data_ptr() + 4 * (x.coff + coff))
                src._keep = x.t
                dst = _lib.DevPtr(V.data_ptr() + 4 * voff)        # H2 rows: 128 bytes per 32-channel chunk = 4 bytes per channel too
                dst._keep = V
                if h2:
                    call("coocc_wino_input_h2_ex", src, x.stride, x.B, x.X, x.Y, x.Z, cr, tile, dst, pc.Cin, G, vscale, ptr(sdev))
                else:
                    call("coocc_wino_input_strided", src, x.stride, x.B, x.X, x.Y, x.Z, cr, tile, dst, pc.Cin, G)
                voff += cr
    d = ConvDesc()
    ws = workspace(dev)
    d.in_, d.w, d.out = ptr(V), ptr(wp), ptr(Mb)
    d.scale = d.bias = d.res = d.gather = d.out_rows = None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = pts * G, pc.Cin, pc.Cout, 3
    d.in_stride, d.out_stride, d.res_stride = pc.Cin, pc.Cout, 0
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = pts * G // x.Z, 1, 1, x.Z, 1, 1, x.Z
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.kx, d.ky, d.kz, d.px, d.py, d.pz = 1, 1, 3, 0, 0, 1
    d.wgroup_rows = G
    d.relu, d.res_mode, d.splitk, d.tile_hint = 0, 0, 1, (hint or TILE_HINT)
    kname = conv_kernel_name(pts * G, pc.Cout, False, hint, 3 * -(-pc.Cin // 32))
    if (CONV_PERSIST and CONV_V2 and kname == "k_conv2<128>" and 3 * -(-pc.Cin // 32) <= 24
            and (pts * G // 128) * -(-pc.Cout // 128) > 768):
        kname = "k_conv2p"          # mirror of the dispatch in coocc_conv_fwd: persistent workgroups, >= 2 tiles each
    if h2:
        d.mfma_dtype, d.alpha, kname = 3, 1.0 / vscale, "k_gemm_h2z"
        if sdev is not None:
            inv = _lib.DevPtr(sdev.data_ptr() + 4)
            inv._keep = sdev
            d.alpha_dev = inv
    with TIMER.region(kname + " wino%d" % tile, 2.0 * pts * rows * pc.Cin * pc.Cout * 3):
        _lib.conv_fwd(d, V.device)
    tw = None
    if twin and h2 and pc.Cout % 32 == 0 and out.coff == 0 and out.stride % 4 == 0:
        tw = out.h2 = torch.empty(out.B * out.V, pc.Cout, device=dev, dtype=_F32)
    with TIMER.region("k_wino_out", 4.0 * pts * rows * pc.Cout + 4.0 * x.V * pc.Cout * (2 if tw is not None else 1)):
        call("coocc_wino_output_ex", ptr(Mb), G, x.B, x.X, x.Y, x.Z, pc.Cout, tile, out.data(), out.stride, ptr(pc.scale),
             ptr(pc.bias), res.data() if res is not None else None, res.stride if res is not None else 0, int(relu), ptr(tw))
    return out


def workspace(device, nfloats=64 << 20):
    """Split-K scratch (256 MB by default), one per device, reused by every launch on the stream."""
    key = (device.index, nfloats, _lib.stream(device).value)   # one per stream: samples overlap
    if key not in _ws_cache:
        _ws_cache[key] = torch.empty(nfloats, device=device, dtype=_F32)
    return _ws_cache[key]


def out_dim(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def conv_rows(x, pc, relu=True, res=None, res_mode=0, out=None, splitk=0, twin_for=()):
    """out = epi(conv(x)) on Rows.  res: Rows added before ReLU (res_mode 1).  ``twin_for``: the layers (PackedConv) that
    read the result next -- when one of them takes the split-f16 direct path the epilogue writes the H2 twin it needs
    (``out.h2``) next to the fp32 rows, which replaces that layer's conversion pass."""
    Xo, Yo, Zo = (out_dim(x.X, pc.ksize, pc.stride, pc.pad), out_dim(x.Y, pc.ksize, pc.stride, pc.pad),
                  out_dim(x.Z, pc.ksize, pc.stride, pc.pad))
    M = x.B * Xo * Yo * Zo
    if out is None:
        out = Rows(torch.empty(M, pc.Cout, device=x.t.device, dtype=_F32), x.B, Xo, Yo, Zo, pc.Cout)
    else:
        out.h2 = out.h16 = None        # a caller-provided buffer is overwritten: whatever twins it carried are stale
    assert x.C == pc.Cin, "channel mismatch: %d vs %d" % (x.C, pc.Cin)
    rm = res_mode or (1 if res is not None else 0)
    bf16 = CONV_DTYPE == "bf16"
    f16 = CONV_DTYPE == "f16"
    plan = None if (bf16 or f16) else wino_plan(x, pc, M, rm)
    twin = bool(twin_for) and CONV_ENGINE == "h2" and not (bf16 or f16) and pc.Cout % 32 == 0 and takes_h2(out, twin_for)
    if plan is not None:
        return conv_rows_wino(x, pc, out, relu, res, plan, twin=twin)
    ws = workspace(x.t.device)
    d = ConvDesc()
    d.in_, d.w, d.out = x.data(), ptr(pc.w), out.data()
    d.scale, d.bias = ptr(pc.scale), ptr(pc.bias)
    d.res = res.data() if res is not None else None
    d.gather = None
    d.out_rows = None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = M, pc.Cin, pc.Cout, pc.taps
    d.in_stride, d.out_stride = x.stride, out.stride
    d.res_stride = res.stride if res is not None else 0
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = x.B, x.X, x.Y, x.Z, Xo, Yo, Zo
    d.ksize, d.stride, d.pad = pc.ksize, pc.stride, pc.pad
    d.relu, d.res_mode, d.splitk = int(relu), (res_mode or (1 if res is not None else 0)), splitk
    d.tile_hint = TILE_HINT
    d.mfma_dtype = 1 if bf16 else 0
    taps = pc.taps
    trim = None
    if ZTRIM and pc._w_cube is not None:
        # z taps that are in range for at least one output z; on thin grids (Z = 1, 2) the rest only multiply padding
        ok = [kz for kz in range(3) if any(0 <= zo * pc.stride - pc.pad + kz < x.Z for zo in range(Zo))]
        lo, hi = ok[0], ok[-1]
        if hi - lo + 1 < 3:
            trim = (lo, hi)
            d.w = ptr(pc.ztrim_pack(lo, hi))
            d.kx, d.ky, d.kz, d.px, d.py, d.pz = 3, 3, hi - lo + 1, pc.pad, pc.pad, pc.pad - lo
            d.taps = taps = 9 * (hi - lo + 1)
    same = pc.stride == 1 and (Xo, Yo, Zo) == (x.X, x.Y, x.Z)
    if f16 and pc.Cin % 64 == 0 and pc._w_taps is not None and rm in (0, 1) and splitk in (0, 1) and out.coff == 0 and out.stride == pc.Cout:
        # configs[4]'s reduced-precision path: ONE v_mfma_f32_32x32x16_f16 per step on f16 operands that live in HBM as f16 rows
        # (2 bytes per element) -- written by the PRODUCER's epilogue (out16; Rows.h16) or, for inputs that came from a non-conv
        # kernel, by one conversion pass -- fp32 accumulate / BN / residual / ReLU; the fp32 rows are written as well (the
        # resampling / mixing / fine-branch kernels and the residual adds read those).  Direct form (no Winograd: its transform
        # constants cost bits f16 does not have).
        xh = x.h16
        if xh is None:
            xh = torch.empty(x.B * x.V, pc.Cin, device=x.t.device, dtype=torch.float16)
            call("coocc_rows_to_f16", x.data(), x.stride, x.B * x.V, pc.Cin, ptr(xh))
        out.h16 = torch.empty(M, pc.Cout, device=x.t.device, dtype=torch.float16) if pc.Cout % 4 == 0 else None
        d.in_, d.in_stride, d.w, d.mfma_dtype, d.alpha, d.splitk = ptr(xh), pc.Cin, ptr(pc.h1_pack(trim)), 4, 1.0, 1
        if out.h16 is not None:
            d.out16, d.out16_stride = ptr(out.h16), pc.Cout
        with TIMER.region("k_gemm_h1z" if (same and taps > 1) else "k_gemm_h1w", 2.0 * M * pc.Cin * pc.Cout * taps):
            _lib.conv_fwd(d, pc.w.device)
        return out
    if (not bf16 and not f16 and CONV_ENGINE == "h2" and H2_DIRECT and pc.Cin % 32 == 0 and pc._w_taps is not None
            and rm in (0, 1) and 2.0 * M * pc.Cin * pc.Cout * taps >= H2_DIRECT_MIN_FLOPS):
        # fp32-accurate split-f16 GEMM (csrc/gemm_h2.hip) for the layers the Winograd path leaves out (small grids, strided,
        # 1x1x1): the input rows are split into H2 rows once per layer, stride-1 "same" layers share one LDS image per 3 z taps
        # (k_gemm_h2z), the rest fetch one image per (chunk, tap) (k_gemm_h2w); the epilogue (folded BN, residual, ReLU) is the usual one
        if x.coff == 0 and x.C == pc.Cin:
            xh = h2_rows(x)                 # the producer's twin, or one conversion shared by every consumer of x
        else:
            xh = scratch(x.t.device, "h2in", x.B * x.V * pc.Cin)
            call("coocc_rows_to_h2", x.data(), x.stride, x.B * x.V, pc.Cin, 1.0, ptr(xh))
        d.in_, d.in_stride, d.w, d.mfma_dtype, d.alpha = ptr(xh), pc.Cin, ptr(pc.h2_pack(trim)), 3, 1.0
        if INKERNEL_REDUCE:
            sem = tile_sem(x.t.device)      # split-K layers reduce in-kernel (last workgroup of a tile): no k_conv_reduce launch
            d.tile_sem, d.tile_sem_ints = ptr(sem), sem.numel()
        if twin and out.coff == 0 and out.stride % 4 == 0 and (res is None or res.stride % 4 == 0):
            out.h2 = torch.empty(M, pc.Cout, device=x.t.device, dtype=_F32)
            d.out_h2_twin = ptr(out.h2)
        with TIMER.region("k_gemm_h2z direct" if (same and taps > 1) else "k_gemm_h2w", 2.0 * M * pc.Cin * pc.Cout * taps):
            _lib.conv_fwd(d, pc.w.device)
        return out
    if bf16 and BF16_PRECONVERT and pc.Cin % 64 == 0 and pc._w_taps is not None and taps > 1:    # 1x1x1: HBM-bound either way
        # operands bf16 in memory (k_conv_bf16w): the activations are rounded once per layer into a scratch buffer, the
        # weights once per pack
        xb = _bf16_buffer(x.t.device, x.B * x.V * pc.Cin)
        call("coocc_rows_to_bf16", x.data(), x.stride, x.B * x.V, pc.Cin, ptr(xb))
        wb = pc.bf16_pack(trim)
        d.in_, d.in_stride, d.w, d.mfma_dtype = ptr(xb), pc.Cin, ptr(wb), 2
    if not TIMER.enabled:
        _lib.conv_fwd(d, pc.w.device)
        return out
    if d.mfma_dtype == 2:      # mirror of the dispatch in coocc_conv_fwd: stride-1 "same" layers share the tile between z taps
        kname = "k_conv_bf16z" if (pc.stride == 1 and (Xo, Yo, Zo) == (x.X, x.Y, x.Z) and ZSHARE) else "k_conv_bf16w"
    else:
        kname = "k_conv_bf16" if bf16 else conv_kernel_name(M, pc.Cout, False, 0, taps * -(-pc.Cin // 32),
                                                         pc.ksize == 1 and pc.stride == 1 and pc.pad == 0)
    with TIMER.region(kname, 2.0 * M * pc.Cin * pc.Cout * taps):
        _lib.conv_fwd(d, pc.w.device)
    return out


def linear_rows(x2d, pc, relu=False, out=None, out_coff=0, in_coff=0, in_C=None):
    """Row-wise nn.Linear on a [n, stride] tensor via the same kernel (taps = 1)."""
    n = x2d.shape[0]
    Cin = in_C if in_C is not None else pc.Cin
    if out is None:
        out = torch.empty(n, pc.Cout, device=x2d.device, dtype=_F32)
    if n == 0:
        return out
    ws = workspace(x2d.device)
    d = ConvDesc()
    d.in_ = ctypes.c_void_p(x2d.data_ptr() + 4 * in_coff)
    d.w = ptr(pc.w)
    d.out = ctypes.c_void_p(out.data_ptr() + 4 * out_coff)
    d.scale, d.bias = ptr(pc.scale), ptr(pc.bias)
    d.res = None
    d.gather = None
    d.out_rows = None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = n, Cin, pc.Cout, 1
    d.in_stride, d.out_stride, d.res_stride = x2d.shape[1], out.shape[1], 0
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = 1, n, 1, 1, n, 1, 1
    d.ksize, d.stride, d.pad = 1, 1, 0
    d.relu, d.res_mode, d.splitk = int(relu), 0, 0
    d.tile_hint = TILE_HINT
    with TIMER.region(conv_kernel_name(n, pc.Cout, False, 0, -(-Cin // 32), True), 2.0 * n * Cin * pc.Cout):
        _lib.conv_fwd(d, x2d.device)
    return out


def linear_rows_h2(xh, n, Cin, pc, relu=False, out=None, out_coff=0, out_h2=False):
    """Row-wise nn.Linear on the split-f16 engine: ``xh`` = H2 rows [n, Cin] (``rows_to_h2`` or a producer's ``out_h2``
    epilogue).  ``out_h2``: the result is written as H2 rows too (a float32-typed [n, Cout] tensor holding them)."""
    dev = xh.device
    if out is None:
        out = torch.empty(n, pc.Cout, device=dev, dtype=_F32)
    ws = workspace(dev)
    d = ConvDesc()
    d.in_, d.w = ptr(xh), ptr(pc.h2_pack())
    d.out = ctypes.c_void_p(out.data_ptr() + 4 * out_coff)
    d.scale, d.bias = ptr(pc.scale), ptr(pc.bias)
    d.res = d.gather = d.out_rows = None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = n, Cin, pc.Cout, 1
    d.in_stride, d.out_stride, d.res_stride = Cin, out.shape[1], 0
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = 1, n, 1, 1, n, 1, 1
    d.ksize, d.stride, d.pad = 1, 1, 0
    d.relu, d.res_mode, d.splitk = int(relu), 0, 1
    d.mfma_dtype, d.alpha, d.out_h2 = 3, 1.0, int(out_h2)
    with TIMER.region("k_gemm_h2w linear", 2.0 * n * Cin * pc.Cout):
        _lib.conv_fwd(d, dev)
    return out


def rows_to_h2(x2d, C=None, coff=0, name="h2rows"):
    """H2 copy of the first C channels (from ``coff``) of a [n, stride] fp32 tensor (per-stream scratch)."""
    n = x2d.shape[0]
    C = C if C is not None else x2d.shape[1]
    xh = scratch(x2d.device, name, n * C)
    src = _lib.DevPtr(x2d.data_ptr() + 4 * coff)
    src._keep = x2d
    call("coocc_rows_to_h2", src, x2d.shape[1], n, C, 1.0, ptr(xh))
    return xh


def g1_h2_capable(pc, C):
    """True when ``gather_conv_rows`` runs this pack on the split-f16 engine (and therefore reads H2 source rows)."""
    return bool(CONV_ENGINE == "h2" and H2_DIRECT and C % 32 == 0 and h2_capable(pc) and pc.h2_pack() is not None)


def g1_sources_h2(cat4, C, pc):
    """The H2 operand rows of BOTH G1 gather GEMMs from one conversion launch: columns [0, 2C) of the concat rows (img | pts)
    as H2 rows [V, 2C] -- a 32-channel chunk is 128 contiguous bytes, so each slot is a column range of the wider rows
    (row stride 2C, the pts slot C * 4 bytes in).  None when the layer is not on the split-f16 engine."""
    if not g1_h2_capable(pc, C) or cat4.shape[1] < 2 * C:
        return None
    return rows_to_h2(cat4, 2 * C, 0, name="g1rows2")


def gather_conv_rows(src, src_coff, pc, gather, out_rows, dst, dst_coff, gate_coff, C, relu=True, count_dev=None, src_h2=None):
    """GSFusion G1: dst[out_rows[m], dst_coff:+C] = relu(sum_k W_k . src[gather[k,m], src_coff:+C] + b)
    * dst[out_rows[m], gate_coff:+C]   (bifuser_n.py:138-169).  src/dst: [rows, stride] tensors.
    ``count_dev``: int32 device tensor holding the number of rows; ``gather`` [K, cap] / ``out_rows`` [cap] are then
    capacity-sized buffers and nothing about the launch depends on the count (hipGraph replay).
    ``src_h2``: (H2 rows of ``src``'s columns [c0, c0 + n), c0, n) made by the caller (``g1_sources_h2``: one conversion for
    both gather GEMMs); without it the source slot is converted here."""
    K, M = gather.shape
    if M == 0:
        return
    ws = workspace(src.device)
    d = ConvDesc()
    d.in_ = ctypes.c_void_p(src.data_ptr() + 4 * src_coff)
    d.w = ptr(pc.w)
    d.out = ctypes.c_void_p(dst.data_ptr() + 4 * dst_coff)
    d.scale, d.bias = None, ptr(pc.bias)
    d.res = ctypes.c_void_p(dst.data_ptr() + 4 * gate_coff)
    d.gather = ptr(gather, torch.int32)
    d.out_rows = ptr(out_rows, torch.int32)
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = M, C, pc.Cout, K
    d.in_stride, d.out_stride, d.res_stride = src.shape[1], dst.shape[1], dst.shape[1]
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = 1, src.shape[0], 1, 1, 1, 1, 1     # Xi = number of source rows
    d.ksize, d.stride, d.pad = 1, 1, 0
    d.relu, d.res_mode, d.splitk = int(relu), 2, 1
    d.tile_hint = TILE_HINT
    if count_dev is not None:
        d.M_dev, d.gather_stride = ptr(count_dev, torch.int32), M
    kname = conv_kernel_name(M, pc.Cout, True)
    if g1_h2_capable(pc, C):
        # split-f16 engine: the source slot is converted once ([rows, C] H2 rows: 13 us for 80 k rows), the row-table kernel
        # gathers 128-byte chunks of it (k_gemm_h2w<TABLE>); 118 -> ~40 us per call at configs[1]
        if src_h2 is not None and src_h2[1] <= src_coff and src_coff + C <= src_h2[1] + src_h2[2] and (src_coff - src_h2[1]) % 32 == 0:
            sh, c0, n = src_h2
            hp = _lib.DevPtr(sh.data_ptr() + 4 * (src_coff - c0))
            hp._keep = sh
            d.in_, d.in_stride = hp, n
        else:
            sh = rows_to_h2(src, C, src_coff, name="g1rows")
            d.in_, d.in_stride = ptr(sh), C
        d.w, d.mfma_dtype, d.alpha, kname = ptr(pc.h2_pack()), 3, 1.0, "k_gemm_h2w"
    with TIMER.region(kname, 2.0 * M * C * pc.Cout * K):
        _lib.conv_fwd(d, src.device)


class PackCache:
    """Re-pack lazily when any source parameter/buffer changed (torch `_version` counters).  Writes that bypass the
    version counter (``p.data.copy_()``, EMA / weight-surgery hooks) are not seen: ``PackCache(owner)`` registers a
    ``load_state_dict`` post-hook on the owning module that drops the packs, and ``invalidate()`` (or
    ``co_occ_amd.invalidate_packs(model)``) does so explicitly after any other in-place surgery."""

    def __init__(self, owner=None):
        self._key = None
        self._val = None
        self._slots = None
        if owner is not None and hasattr(owner, "register_load_state_dict_post_hook"):
            owner.register_load_state_dict_post_hook(self._on_load_state_dict)

    def _on_load_state_dict(self, module, incompatible_keys):
        self.invalidate()

    def __getstate__(self):
        # pickling a module (torch.save(model), multiprocessing spawn) carries the hook = this object: without its device packs
        return dict(_key=None, _val=None, _slots=None)

    def invalidate(self):
        self._key = self._val = None
        self._slots = None

    def get_modules(self, modules, build, buffers=True):
        """``get`` over every parameter (and buffer) under ``modules`` without walking the module tree per call: the
        (``_parameters`` / ``_buffers`` dict, name) slots are listed once and read by direct lookups afterwards, so replaced
        Parameter objects, ``.to()`` moves and in-place writes are all seen; adding or removing SUBMODULES after the first
        call is not (``invalidate()``).  The per-call tree walk was ~0.5 ms of host time per sample."""
        slots = getattr(self, "_slots", None)
        if slots is None:
            slots = []
            for mod in modules:
                for m in mod.modules():
                    slots += [(m._parameters, k) for k in m._parameters]
                    if buffers:
                        slots += [(m._buffers, k) for k in m._buffers]
            self._slots = slots
        key = []
        for d, k in slots:
            t = d.get(k)
            if t is not None:
                key.append(t.data_ptr())
                key.append(t._version)
        if key != self._key:
            self._val = build()
            self._key = key
        return self._val

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors if t is not None)
        if key != self._key:
            self._val = build()
            self._key = key
        return self._val


_PACK_EPOCH = [0]          # bumped by ``invalidate_packs``: part of every ``WeightWatch`` fingerprint


class WeightWatch:
    """Fingerprint of every parameter and buffer under ``module``: (``data_ptr``, ``_version``) per tensor + the global
    pack epoch.  A captured hipGraph bakes RAW pointers to the weight packs / folded-BN constants that were current at capture
    time (``PackCache``); the eager path re-packs when a version counter moves, a replay cannot.  Whoever owns a captured
    graph keeps one of these and re-captures when ``changed()`` (optimizer steps between ``train()`` and ``eval()``,
    ``load_state_dict``, ``.to()`` / ``.half()``, ``invalidate_packs``).  The (dict, name) slots are listed once, like
    ``PackCache.get_modules``; one check is ~50 us of host time for the whole detector."""

    def __init__(self, module):
        self._slots = []
        for m in module.modules():
            self._slots += [(m._parameters, k) for k in m._parameters]
            self._slots += [(m._buffers, k) for k in m._buffers]
        self.key = self.snapshot()

    def snapshot(self):
        key = [_PACK_EPOCH[0]]
        for d, k in self._slots:
            t = d.get(k)
            if t is not None:
                key.append(t.data_ptr())
                key.append(t._version)
        return key

    def changed(self):
        return self.snapshot() != self.key

    def refresh(self):
        self.key = self.snapshot()


def invalidate_packs(module):
    """Drop every cached weight pack / folded-BN constant under ``module`` (call after writing parameters through
    ``.data`` or any other path that does not bump the tensors' version counters).  Captured serving graphs notice through
    the pack epoch and re-capture at their next submit."""
    _PACK_EPOCH[0] += 1
    n = 0
    for m in module.modules():
        pc = getattr(m, "_packs", None)
        if isinstance(pc, PackCache):
            pc.invalidate()
            n += 1
    return n
