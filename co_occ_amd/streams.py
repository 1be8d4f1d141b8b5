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

# 1 (default): ordering between this package's streams uses DeviceEvent (no system-scope fence per record); 0: torch.cuda.Event
LIGHT_EVENTS = __import__("os").environ.get("COOCC_LIGHT_EVENTS", "1") != "0"


class DeviceEvent:
    """An event for ordering streams of ONE device (``coocc_event_*``: hipEventDisableSystemFence), with the part of
    ``torch.cuda.Event``'s interface this package uses: ``record(stream=None)``, ``wait(stream=None)``, ``synchronize()``,
    ``query()``, ``elapsed_time(end)`` (``timing=True``).  A default HIP / torch event writes back and invalidates the caches at
    every record so that the host and other devices see the stream's writes; the serving loop records ~8 per sample only to order
    its own streams, and with three dense stages sharing the chip those fences were 24 % of the throughput
    (profiles/r6_serving_probe_events.txt).  NOT a hand-over to the host or to another device: synchronise the stream, or copy
    device -> host on it (stream-ordered), for that."""

    __slots__ = ("h", "_lib")

    def __init__(self, timing=False, blocking=False):
        self._lib = _lib.load()
        out = ctypes.c_void_p(0)
        _lib.check(self._lib.coocc_event_create((1 if timing else 0) | (2 if blocking else 0), ctypes.byref(out)))
        self.h = out

    @staticmethod
    def _s(stream):
        if stream is None:
            stream = torch.cuda.current_stream()
        return ctypes.c_void_p(stream.cuda_stream)

    def record(self, stream=None):
        _lib.check(self._lib.coocc_event_record(self.h, self._s(stream)))

    def wait(self, stream=None):
        _lib.check(self._lib.coocc_stream_wait_event(self._s(stream), self.h))

    def synchronize(self):
        _lib.check(self._lib.coocc_event_synchronize(self.h))

    def query(self):
        rc = self._lib.coocc_event_query(self.h)
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def elapsed_time(self, end):
        ms = ctypes.c_float(0.0)
        _lib.check(self._lib.coocc_event_elapsed_ms(self.h, end.h, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h is not None and h.value:
            try:
                self._lib.coocc_event_destroy(h)
            except Exception:
                pass


def new_event(timing=False, blocking=False):
    """The event type this package orders its streams with (``DeviceEvent``; ``torch.cuda.Event`` under COOCC_LIGHT_EVENTS=0).
    Both have ``record(stream)`` / ``wait(stream)`` / ``synchronize()`` / ``query()`` / ``elapsed_time()``."""
    return DeviceEvent(timing, blocking) if LIGHT_EVENTS else torch.cuda.Event(enable_timing=timing, blocking=blocking)


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
