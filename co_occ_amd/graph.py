"""Serving loop over captured hipGraphs: one host call per stage instead of ~250 launches per sample.

The reference's ``simple_test`` (coocc_ray.py:520-627) is a Python call tree; so was this package's hot path through
round 2, and with the convolutions on the f16 matrix cores (csrc/gemm_h2.hip) the HOST became the bottleneck: ~6 ms of
ctypes launches per sample against ~5 ms of GPU work.  HIP's answer is stream capture: the dense stage of a sample --
G1 gather GEMMs -> con_enc -> CustomResNet3D -> FPN3D -> OccHead (coarse + cascade fine, scattered into the dense grid)
-> render -- is issued ONCE under ``torch.cuda.graph`` and replayed with one ``hipGraphLaunch`` per sample.  What makes the
stage capturable:

* every data-dependent size stays on the device: the non-empty voxel counts of the fuser (``SearchSlot.counts``) and the
  foreground count of the fine branch are read by the kernels themselves (``coocc_conv_desc.M_dev``, the ``*_dev`` entry
  points), grids and buffers are sized for the worst case and surplus workgroups leave at once;
* everything the stage reads sits at fixed addresses: a ``SearchSlot`` per sample in flight (concat rows, voxel lists,
  neighbour tables -- written by the eager search stage on its own stream) and the sample's own tensors.

The index search (K1-K5: FPS chains, top-K, ball query) stays eager on prefetch streams, one sample ahead, exactly as
before: it is latency-bound on two CUs and hidden under the previous sample's dense stage.  Outputs of a replay are the
graph's static tensors (``DenseGraph.out``): valid until the same graph is replayed again.

``tests/test_gpu_graph.py`` checks a replay bit for bit against the eager path on the same sample.
"""
import torch

from . import _lib, core
from .fuser import SearchSlot


NATIVE_SEARCH = __import__("os").environ.get("COOCC_NATIVE_SEARCH", "1") != "0"     # 0: the Python-issued search stage


class DenseGraph:
    """The dense stage of one sample slot as a hipGraph.  ``inputs``: dict(pts [1,C,X,Y,Z], gemo, img_feats, transform) of the
    sample bound to this slot; the slot's concat buffer must have been filled by ``search_into_slot`` at least once before
    capture (warm-up) and before every replay."""

    def __init__(self, model, slot, inputs, stream, render=True):
        self.model, self.slot, self.inputs, self.stream, self.render = model, slot, inputs, stream, render
        self.graph = None
        self.out = None

    def _run(self):
        m, s = self.model, self.inputs
        vf = m.occ_fuser.forward_static(self.slot)
        gemo, cam_geo = s.get("gemo"), None
        if self.render and s.get("cams") is not None:
            # P1 per sample (ViewTransformerLSSBEVDepth.py:117-150 via coocc_ray.py:186): the frustum geometry the render
            # block samples comes from the camera matrices inside the stage -- the 39 constants per camera by one tiny kernel,
            # the per-sample chain in the ray kernel itself -- not from a pre-baked [N,D,H,W,3] tensor
            cam_geo = m.img_view_transformer._camera_mats(*s["cams"])
            mats = cam_geo[0].reshape(-1, cam_geo[0].shape[-1])
            cam_geo = (mats,) + tuple(cam_geo[1:])
        return m.decode(vf, gemo, s["img_feats"], s["transform"], self.render, static=True, cam_geo=cam_geo)

    def capture(self, warmup=2):
        """Eager warm-up on the capture stream (weight packs, per-stream scratch), then the capture itself."""
        assert not core.TIMER.enabled, "kernel timing regions cannot be captured"
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(warmup):
                self._run()
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
            self.out = self._run()
        # The captured launches hold RAW pointers into the stream's scratch buffers (Winograd V / M, split-K slabs, arrival
        # counters): keep those tensors alive for as long as the graph exists.  A later eager call on the same stream that
        # needs a larger buffer REPLACES the dict entry (core.scratch / _wino_buffer) -- without this reference the old tensor
        # would be freed, handed to someone else by the caching allocator, and every replay would scribble over it.
        dev = self.slot.cat4.device
        self._pinned = core.stream_scratch(dev, self.stream)
        for side in core.branch_streams_of(dev, self.stream):          # the stage's forked branches (core.Fork) have their own scratch
            self._pinned += core.stream_scratch(dev, side)
        return self

    def fits(self, counts):
        """Host-side check of a sample's voxel counts against the capacities the graph was captured with."""
        Ni, Np = counts
        from . import fuser
        split = fuser.SPLIT_C0 or self.slot.V > fuser.DENSE_C0_MAX_VOXELS
        cap = self.model.occ_fuser.c0_capacity(self.slot.V) if split else self.slot.V   # the dense form of con_enc.0 has no capacity
        return 0 < Np <= cap and Ni > 0

    def replay(self):
        self.graph.replay()
        return self.out


def search_into_slot(model, slot, depth, ctx, cams, pts):
    """Pooling (fused Lift (x) Splat into slot 0 of the slot's concat rows) + index search of one sample on the CURRENT
    stream, outputs in the slot's static buffers.  Returns the ``SearchResult`` (events, host-side counts)."""
    img = model.img_view_transformer.lift_splat(depth, ctx, cams=cams, out=slot.img_rows())
    if NATIVE_SEARCH:
        return model.occ_fuser.search_native(pts, slot)
    return model.occ_fuser.search(img, pts, slot=slot)


def make_slot(model, grid, device):
    f = model.occ_fuser
    return SearchSlot(f.in_channels, f.knum, grid, device)
