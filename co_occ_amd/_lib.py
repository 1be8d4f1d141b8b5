"""ctypes binding of libcoocc_hip.so (include/coocc_hip.h).

The HIP library is the product: there is NO CPU or eager-PyTorch fallback.  Importing this
module without a built library, or calling an op with CPU tensors, raises immediately.
"""
import ctypes
import os
import threading
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcoocc_hip.so")

c_int, c_float, c_void_p, c_size_t, c_int64 = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_int64)


class ConvDesc(ctypes.Structure):
    """struct coocc_conv_desc (include/coocc_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("in_", "w", "out", "scale", "bias", "res", "gather", "out_rows", "ws")] + \
               [("ws_floats", c_int64)] + \
               [(n, c_int) for n in ("M", "Cin", "Cout", "taps", "in_stride", "out_stride", "res_stride",
                                     "B", "Xi", "Yi", "Zi", "Xo", "Yo", "Zo", "ksize", "stride", "pad",
                                     "relu", "res_mode", "splitk", "tile_hint", "kx", "ky", "kz", "px", "py", "pz",
                                     "wgroup_rows", "mfma_dtype")] + [("alpha", ctypes.c_float), ("M_dev", c_void_p), ("gather_stride", c_int), ("out16", c_void_p), ("out16_stride", c_int), ("out_h2", c_int),
                                                                   ("out_h2_twin", c_void_p), ("tile_sem", c_void_p), ("tile_sem_ints", c_int),
                                                                  ("alpha_dev", c_void_p)]


class SearchDesc(ctypes.Structure):
    """struct coocc_search_desc (include/coocc_hip.h)."""
    _fields_ = [("cat4", c_void_p), ("pts", c_void_p), ("pts_rows", c_int), ("pts_stride", c_int)] + \
               [(n, c_int) for n in ("C", "X", "Y", "Z", "K", "fps_num", "max_cluster")] + \
               [("radius", ctypes.c_float), ("dist_thresh", ctypes.c_float), ("offsets", c_void_p), ("noff", c_int)] + \
               [(n, c_void_p) for n in ("lin", "counts", "near_img", "near_pts", "rows", "rows_p", "ws")] + \
               [("ws_bytes", ctypes.c_size_t), ("counts_host", c_void_p)]


P, I, F, Z, L = c_void_p, c_int, c_float, c_size_t, c_int64
# name -> (restype, argtypes); mirrors include/coocc_hip.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    "coocc_last_error": (ctypes.c_char_p, []),
    "coocc_abi_version": (I, []),
    "coocc_device_cu_count": (I, [P]),
    "coocc_stream_create_cu_mask": (I, [P, I, P]),
    "coocc_stream_destroy": (I, [P]),
    "coocc_event_create": (I, [I, P]),
    "coocc_event_destroy": (I, [P]),
    "coocc_event_record": (I, [P, P]),
    "coocc_stream_wait_event": (I, [P, P]),
    "coocc_event_synchronize": (I, [P]),
    "coocc_event_query": (I, [P]),
    "coocc_event_elapsed_ms": (I, [P, P, P]),
    "coocc_ncdhw_to_ndhwc": (I, [P, P, I, I, I, I, I, P]),
    "coocc_rows_to_bf16": (I, [P, I, L, I, P, P]),
    "coocc_rows_to_h2": (I, [P, I, L, I, F, P, P]),
    "coocc_rows_to_h2_ex": (I, [P, I, L, I, F, P, P, P]),
    "coocc_rows_to_h2_gather": (I, [P, I, P, L, P, I, F, P, P]),
    "coocc_rows_to_f16": (I, [P, I, L, I, P, P]),
    "coocc_wino_input_h2": (I, [P, I, I, I, I, I, I, I, P, I, L, F, P]),
    "coocc_wino_input_h2_ex": (I, [P, I, I, I, I, I, I, I, P, I, L, F, P, P]),
    "coocc_ndhwc_to_ncdhw": (I, [P, P, I, I, I, I, I, P]),
    "coocc_fuser_prepare": (I, [P, P, P, P, P, I, I, I, P]),
    "coocc_fuser_prepare_rows": (I, [P, I, I, P, I, I, P, P, P, I, I, I, P]),
    "coocc_compact_flags": (I, [P, I, P, P, P, Z, P]),
    "coocc_compact_flags_ex": (I, [P, I, P, P, P, P, Z, P]),
    "coocc_lin_to_coords": (I, [P, I, I, I, I, P, P, P]),
    "coocc_furthest_point_sampling": (I, [I, I, I, P, P, P, P]),
    "coocc_fps_voxels_ws": (Z, [I, I, I]),
    "coocc_fps_voxels": (I, [P, I, I, I, I, I, P, P, Z, P]),
    "coocc_fps_voxels_pair": (I, [P, I, P, P, P, I, P, P, Z, I, I, I, I, P]),
    "coocc_ball_query": (I, [I, I, I, F, F, I, P, P, P, P]),
    "coocc_knn_topk": (I, [I, I, I, P, P, P, P, P]),
    "coocc_voxel_index_map": (I, [P, I, I, P, P]),
    "coocc_voxel_index_map_dev": (I, [P, I, P, I, P, P]),
    "coocc_fine_sample_voxel_dev": (I, [P, I, I, I, I, P, I, P, I, P, P, P, I, P]),
    "coocc_fine_sample_img_dev": (I, [P, I, I, I, I, P, P, L, P, P, I, I, P]),
    "coocc_fine_mlp_pre_dev": (I, [P, I, P, I, L, P, I, P, P, P, F, P, P, P, P, F, P, P, I, P, P]),
    "coocc_fine_fused": (I, [P, I, I, I, P, I, I, I, P, P, I, P, I, P, P, P, P, F, P, P, P, P, F, P, P, I, P, P, P]),
    "coocc_fine2_pack": (I, [P, P, I, P, P, P, P, P, P, P, P, P, P]),
    "coocc_fine2_h2": (I, [P, I, I, I, I, P, I, I, I, P, P, I, P, P, P, P, F, F, I, P, P, P, P]),
    "coocc_fine_sample_img_lin": (I, [P, I, I, I, I, P, P, I, I, I, P, P, I, I, P]),
    "coocc_scatter_fine_dev": (I, [P, L, P, I, I, I, P, P, I, I, I, F, P]),
    "coocc_scatter_fine_grouped": (I, [P, I, I, P, I, P, I, I, I, I, P, F, P, P]),
    "coocc_ball_query_voxels": (I, [I, F, F, I, I, I, I, P, P, P, P, P]),
    "coocc_knn_topk_voxels": (I, [I, I, I, I, I, I, P, P, P, P, I, P, P, P, P, P]),
    "coocc_knn_assign": (I, [I, I, I, I, F, P, P, P, P, P, P]),
    "coocc_knn_threshold": (I, [I, F, P, P, P, P]),
    "coocc_index_rows_i32": (I, [P, I, P, I, P, P]),
    "coocc_conv_pack_weights": (L, [P, I, I, I, I, P]),
    "coocc_conv_fwd": (I, [ctypes.POINTER(ConvDesc), P]),
    "coocc_fuser_search_ws": (Z, [ctypes.POINTER(SearchDesc)]),
    "coocc_fuser_search": (I, [ctypes.POINTER(SearchDesc), P, P]),
    "coocc_upsample_add_trilinear": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "coocc_upsample_add_trilinear_ex": (I, [P, P, I, I, I, I, I, I, I, I, P, P]),
    "coocc_occhead_mix": (I, [P, P, I, P, P, I, I, P]),
    "coocc_occhead_mix_ex": (I, [P, P, I, P, P, I, I, P, P]),
    "coocc_argmax_flags": (I, [P, I, I, I, I, P, P]),
    "coocc_fine_sample_voxel": (I, [P, I, I, I, I, P, I, I, P, P, P, I, P]),
    "coocc_fine_sample_img": (I, [P, I, I, I, I, P, P, L, P, I, I, P]),
    "coocc_fine_mlp_pre": (I, [P, I, P, I, L, P, P, P, F, P, P, P, P, F, P, P, I, P, P]),
    "coocc_fine_mlp": (I, [P, I, P, I, L, P, P, P, P, F, P, P, P, P, F, P, P, I, P, P]),
    "coocc_groupnorm_rows": (I, [P, L, I, I, I, P, P, F, I, P]),
    "coocc_groupnorm_nhwc": (I, [P, I, I, I, I, P, P, F, I, P]),
    "coocc_scatter_fine": (I, [P, L, I, I, P, P, I, I, I, F, P]),
    "coocc_camera_mats": (I, [P, P, P, P, P, P, I, I, I, I, P, P]),
    "coocc_get_geometry": (I, [P, P, P, P, I, I, I, I, P, P]),
    "coocc_bev_pool_forward": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P]),
    "coocc_bev_pool_backward": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P]),
    "coocc_voxel_pool_ws": (Z, [I, I]),
    "coocc_voxel_pool": (I, [P, P, I, I, I, P, I, I, I, I, P, I, P, Z, I, P]),
    "coocc_lift_splat": (I, [P, P, P, I, I, I, I, I, I, P, I, I, I, I, P, I, P, Z, I, P]),
    "coocc_lift_splat_cams": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, I, I, I, I, P, I, P, Z, I, P]),
    "coocc_lift_splat_reuse": (I, [P, P, I, I, I, I, I, I, I, I, I, P, I, P, Z, P]),
    "coocc_bev_pool_coords": (I, [P, P, I, I, I, I, I, I, P, I, P, Z, I, P]),
    "coocc_render_nearest": (I, [P, I, I, I, P, P, I, I, I, I, P, I, P, P]),
    "coocc_render_nearest_cams": (I, [P, I, I, I, P, P, P, P, P, I, I, I, I, P, I, P, P]),
    "coocc_render_activate_table": (I, [P, I, P]),
    "coocc_upsample_maps": (I, [P, I, I, I, I, P, P, P]),
    "coocc_volume_sampling": (I, [P, I, I, I, I, P, I, P, P, P, P]),
    "coocc_raw2outputs": (I, [P, P, I, I, I, F, F, P, P, P, P]),
    "coocc_render_losses": (I, [P, P, P, P, L, I, P, P, Z, P]),
    "coocc_conv_pack_weights_dev": (L, [P, I, I, I, I, P, P]),
    "coocc_wino_pack_weights_dev": (L, [P, I, I, I, I, P, P]),
    "coocc_wino_pack_weights_h2_dev": (L, [P, I, I, I, I, P, P]),
    "coocc_conv_pack_weights_h2_dev": (L, [P, I, I, I, I, P, P]),
    "coocc_wino_gradout": (I, [P, I, I, I, I, I, I, I, P, L, P]),
    "coocc_wino_ztap_table": (I, [L, I, P, P]),
    "coocc_wino_wgrad": (I, [P, P, L, I, I, I, I, P, P, I, P, L, P]),
    "coocc_rows_to_kh2": (I, [P, I, L, L, I, F, P, P, P]),
    "coocc_wino_operand_kh2": (I, [I, P, I, I, I, I, I, I, I, P, L, F, P, P]),
    "coocc_conv_wgrad_h2": (I, [P, P, L, I, I, F, P, P, I, P, L, P]),
    "coocc_wino_wgrad_h2": (I, [P, P, L, I, I, I, I, F, P, P, I, P, L, P]),
    "coocc_conv_tap_table": (I, [I, I, I, I, I, I, I, I, I, I, I, P, P]),
    "coocc_conv_epilogue_bwd": (I, [P, I, P, I, P, I, I, I, P, I, P, I, I, P, I, P, L, P]),
    "coocc_conv_epilogue_bwd_ex": (I, [P, I, P, I, P, I, I, I, P, I, P, I, I, P, I, P, L, P, P, F, P]),
    "coocc_conv_wgrad": (I, [P, I, I, P, I, P, I, I, I, I, P, I, P, L, P]),
    "coocc_gather_rows": (I, [P, I, P, I, I, P, I, P]),
    "coocc_scatter_add_rows": (I, [P, I, P, I, I, P, I, P]),
    "coocc_voxel_pool_bwd": (I, [P, I, P, I, I, I, P, I, I, I, I, P, P]),
    "coocc_lift_splat_bwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P, I, I, I, I, P, P, P]),
    "coocc_render_nearest_bwd": (I, [P, I, I, I, P, P, I, I, I, I, P, P, P, P]),
    "coocc_upsample_maps_bwd": (I, [P, P, I, I, I, I, P, P]),
    "coocc_render_losses_bwd": (I, [P, P, P, P, L, I, P, P, P, P, P]),
    "coocc_upsample_trilinear_bwd": (I, [P, P, I, I, I, I, I, I, I, I, I, P]),
    "coocc_wino_input": (I, [P, I, I, I, I, I, I, I, P, L, P]),
    "coocc_wino_input_strided": (I, [P, I, I, I, I, I, I, I, P, I, L, P]),
    "coocc_sparse_tap_sum": (I, [P, P, I, I, I, I, I, P, P, I, I, P]),
    "coocc_wino_output": (I, [P, L, I, I, I, I, I, I, P, I, P, P, P, I, I, P]),
    "coocc_wino_output_ex": (I, [P, L, I, I, I, I, I, I, P, I, P, P, P, I, I, P, P]),
    "coocc_h2_overflow": (I, [I]),
    "coocc_device_fault": (I, [I]),
    "coocc_render_heads_h2": (I, [P, I, I, I, P, P, P, P, P, P, I, P, P, P, I, P]),
    "coocc_projection_params": (I, [P, P, P, P, P, P, I, P, P, P]),
    "coocc_occhead_mix_bwd": (I, [P, P, I, P, P, P, P, I, I, P]),
    "coocc_fine_sample_voxel_bwd": (I, [P, I, I, I, I, I, P, L, P, P, P]),
    "coocc_groupnorm_rows_bwd": (I, [P, P, P, L, I, I, I, P, F, I, P, P, P, P]),
    "coocc_fine_sample_img_bwd": (I, [P, I, I, I, I, I, P, P, L, P, P]),
    "coocc_groupnorm_nhwc_bwd": (I, [P, P, P, I, I, I, I, P, F, I, P, P, P, P]),
    "coocc_voxelize_ws": (Z, [I]),
    "coocc_voxelize_hard": (I, [P, I, I, P, P, I, I, P, P, P, P, P, Z, P]),
    "coocc_vfe_mean": (I, [P, P, I, I, I, I, P, I, P]),
    "coocc_sparse_index_map": (I, [P, I, I, I, I, P, P]),
    "coocc_sparse_conv_table": (I, [P, I, I, I, I, I, I, I, P, P, P]),
    "coocc_sparse_down_flags": (I, [P, I, I, I, I, I, I, I, P, P]),
    "coocc_sparse_lin_to_coors": (I, [P, I, I, I, I, P, P, P]),
    "coocc_bn_stats": (I, [P, I, I, I, P, P, P, Z, P]),
    "coocc_bn_apply": (I, [P, I, I, P, P, P, P, F, P, I, P, P]),
    "coocc_bn_backward": (I, [P, P, P, I, I, P, P, P, F, I, P, P, P, P, P, Z, P]),
    "coocc_bn_backward_sums": (I, [P, P, P, I, I, P, P, F, I, P, P, P, Z, P]),
    "coocc_bn_backward_dx": (I, [P, P, P, I, I, P, P, P, F, I, P, P, ctypes.c_double, P, P, P]),
    "coocc_predict_labels": (I, [P, L, L, L, L, I, I, I, I, I, I, I, P, P]),
    "coocc_eval_semantic": (I, [P, L, L, L, L, I, I, I, I, P, P, I, I, I, I, I, P, P]),
}

_lib = None


class KernelTimer:
    """Optional per-launch HIP-event timing on the launch stream (bench.py's roofline numbers).
    Disabled by default.  ``enabled`` = 1 (True): only the regions that declare their work (the MFMA GEMMs, the
    Winograd transforms, the render pair) record an event pair; 2: every C-ABI call does (bench.py --kernel-table).
    Each pair costs host time and a barrier packet on the stream, so level 2 perturbs the throughput it measures
    (r50: 90.7 samples/s untimed, 81 at level 2)."""

    def __init__(self):
        self.enabled = False
        self.only = None          # level 1: optional tuple of tag prefixes to restrict the timed regions further
        self.records = []

    class _Region:
        def __init__(self, owner, tag, work):
            self.o, self.tag, self.work = owner, tag, work

        def __enter__(self):
            lvl = int(self.o.enabled)
            self.on = lvl >= 2 or (lvl == 1 and self.work > 0 and (self.o.only is None or self.tag.startswith(self.o.only)))
            if self.on:
                self.s = torch.cuda.Event(enable_timing=True)
                self.e = torch.cuda.Event(enable_timing=True)
                self.s.record()
            return self

        def __exit__(self, *a):
            if self.on:
                self.e.record()
                self.o.records.append((self.tag, self.work, self.s, self.e))
            return False

    def region(self, tag, work=0.0):
        return KernelTimer._Region(self, tag, work)

    def summary(self):
        """{tag: dict(launches, ms, work)} -- call after torch.cuda.synchronize()."""
        out = {}
        for tag, work, s, e in self.records:
            d = out.setdefault(tag, dict(launches=0, ms=0.0, work=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["work"] += work
        return out

    def reset(self):
        self.records = []


TIMER = KernelTimer()



def load():
    """Load libcoocc_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "co_occ_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    if os.environ.get("COOCC_FPS_THREADS"):        # tuning knob of the voxel FPS kernel (256 | 512 | 1024)
        lib.coocc_fps_voxels_set_threads(int(os.environ["COOCC_FPS_THREADS"]))
    _lib = lib
    return lib


class CooccError(RuntimeError):
    """Any error of the HIP library.  ``code``: the C ABI's COOCC_E* value (include/coocc_hip.h:31-34) when the library
    returned one, else None."""
    code = None


class CooccArgError(CooccError):
    """COOCC_EINVAL / COOCC_ENOMEM: the library REFUSED the call (an argument or shape an entry point does not take, a
    workspace that is too small) before launching anything -- nothing ran, the caller may take another route."""


class CooccRangeError(CooccError):
    """The split-f16 engine's range guard fired (``core.check_h2_overflow``): kernels ran and their result is not trustworthy.
    Never a reason to fall back silently."""


def check(rc):
    if rc != 0:
        cls = CooccArgError if rc in (-1, -3) else CooccError          # COOCC_EINVAL, COOCC_ENOMEM | COOCC_EHIP and the rest
        e = cls("libcoocc_hip: %s (code %d)" % (load().coocc_last_error().decode(), rc))
        e.code = rc
        raise e


def _device_of(args):
    for a in args:
        t = getattr(a, "_keep", None)
        if t is not None:
            return t.device
    return None


_fns = {}


def call(name, *args):
    """Call an int-returning entry point on the torch HIP stream OF THE ARGUMENTS' DEVICE (the first device pointer
    decides; a module living on cuda:1 while cuda:0 is current launches on cuda:1's current stream under a device guard)."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    dev = _device_of(args)
    if dev is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            with TIMER.region(name):
                check(fn(*args, stream(dev)))
        return
    if not TIMER.enabled:                  # the common case: ~250 of these per sample, keep the host path short
        rc = fn(*args, stream(dev))
        if rc:
            check(rc)
        return
    with TIMER.region(name):
        check(fn(*args, stream(dev)))


def conv_fwd(desc, device):
    """coocc_conv_fwd on ``device``'s current stream (device guard when it is not the current device)."""
    fn = load().coocc_conv_fwd
    if device.index != torch.cuda.current_device():
        with torch.cuda.device(device):
            check(fn(ctypes.byref(desc), stream(device)))
    else:
        check(fn(ctypes.byref(desc), stream(device)))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device=None):
    """The current torch HIP stream of ``device`` (default: the current device) as a raw handle."""
    if _raw_stream is not None:        # no Stream object per launch
        idx = device.index if device is not None and device.index is not None else torch.cuda.current_device()
        return c_void_p(_raw_stream(idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class DevPtr(ctypes.c_void_p):
    """c_void_p that keeps its tensor alive until the foreign call has been issued, so that
    temporaries such as ``ptr(x.contiguous())`` cannot be recycled by the caching allocator
    while later arguments of the same call are still being built."""


def ptr(t, dtype=None, strided=False):
    """Device pointer of a contiguous HIP tensor (None -> NULL).  ``strided=True`` is for entry points that
    take element strides explicitly."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise CooccError("co_occ_amd ops run on the GPU only; got a %s tensor (no CPU fallback)" % t.device)
    if dtype is not None and t.dtype != dtype:
        raise CooccError("expected %s, got %s" % (dtype, t.dtype))
    if not strided and not t.is_contiguous():
        raise CooccError("tensor must be contiguous")
    p = DevPtr(t.data_ptr())
    p._keep = t
    return p


_blocked = threading.local()


def host_read(t):
    """Device->host read of a small tensor: the synchronisation points of the path (voxel counts of the index search, the
    fine-branch count).  The time the calling thread spent blocked is accumulated per thread (``blocked_seconds``) so that
    ``bench.py --diag`` can separate issue time from waiting."""
    t0 = time.perf_counter()
    v = t.tolist()
    _blocked.s = getattr(_blocked, "s", 0.0) + time.perf_counter() - t0
    return v


def blocked_seconds():
    return getattr(_blocked, "s", 0.0)


def host_f32(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])


def host_i32(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])
