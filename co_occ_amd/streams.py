"""CU-partitioned HIP streams (MI355X: 256 CUs).

The two FPS searches of a sample are single-workgroup, latency-bound chains; when their workgroup
shares a CU with convolution waves their time doubles (measured).  `partition()` creates,
per host thread / pipeline slot, one stream masked to the last `reserved` CUs (FPS kernels only) and
one stream masked to all the other CUs (everything else), via hipExtStreamCreateWithCUMask.
"""
import ctypes

import torch

from . import _lib

_active = {}      # main stream handle -> list of FPS streams


class Partition:
    def __init__(self, main, fps, make_main):
        self.main, self.fps = main, fps
        self._make_main, self._sides = make_main, {}

    def side(self, which):
        """Extra stream with the same CU mask as `main` (helper streams of the pipeline)."""
        if which not in self._sides:
            self._sides[which] = self._make_main()
        return self._sides[which]


def partition(device, reserved=2, nfps=2):
    """Returns Partition(main=ExternalStream on CUs [0, n-reserved), fps=[ExternalStream on the
    reserved CUs] * nfps).  Run the pipeline under ``torch.cuda.stream(p.main)``."""
    lib = _lib.load()
    with torch.cuda.device(device):
        n = ctypes.c_int(0)
        _lib.check(lib.coocc_device_cu_count(ctypes.byref(n)))
        ncu = n.value
        nwords = (ncu + 31) // 32

        def make(lo, hi):
            words = [0] * nwords
            for cu in range(lo, hi):
                words[cu // 32] |= 1 << (cu % 32)
            arr = (ctypes.c_uint32 * nwords)(*words)
            out = ctypes.c_void_p(0)
            _lib.check(lib.coocc_stream_create_cu_mask(arr, nwords, ctypes.byref(out)))
            return torch.cuda.ExternalStream(out.value, device=device)
        main = make(0, ncu - reserved)
        per = max(1, reserved // nfps)
        fps = [make(ncu - reserved + min(i * per, reserved - per), ncu - reserved + min(i * per, reserved - per) + per)
               for i in range(nfps)]
    p = Partition(main, fps, lambda: make(0, ncu - reserved))
    _active[main.cuda_stream] = p
    return p


class _FpsOnly:
    """Registry entry of a plain (unmasked) stream whose FPS chains go to reserved-CU streams: no masked helper streams."""

    def __init__(self, fps):
        self.fps = fps

    def side(self, which):
        return None


def attach_fps(stream, fps_streams):
    """``fps_stream_for(stream)`` -> one of ``fps_streams`` from now on; ``stream`` itself stays a plain stream."""
    _active[stream.cuda_stream] = _FpsOnly(list(fps_streams))


def side_stream_for(current, which):
    """Helper stream for a pipeline on `current`: masked like it when partitioned, else None."""
    p = _active.get(current.cuda_stream)
    return p.side(which) if p is not None else None


def fps_stream_for(current, which):
    """The reserved-CU stream paired with `current` (None when the pipeline is not partitioned)."""
    p = _active.get(current.cuda_stream)
    if p is None:
        return None
    return p.fps[which % len(p.fps)]
