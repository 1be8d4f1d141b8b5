"""Serving loop of the hot path: frames in, occupancy + rendered maps out, one ``hipGraphLaunch`` per dense stage.

The reference serves one sample per call (``COOCC_Ray.simple_test``, coocc_ray.py:520-656, driven by
``custom_single_gpu_test``, P/coocc/apis/test.py:22-60).  On MI355X a sample is ~4 ms of GPU work behind ~250 launches, so the
product keeps SEVERAL samples in flight instead (round 3 measured this loop inside ``bench.py``; it now lives here, behind the
reference's call):

* a frame is bound to one of ``slots`` sample slots.  The slot's ``SearchSlot`` (concat rows, voxel lists, device-side counts,
  neighbour tables) and the few inputs the DENSE stage reads (2-D image features, camera matrices) sit at fixed addresses;
  ``submit(frame)`` COPIES the new frame's tensors there -- there is no per-frame capture;
* pooling (fused Lift (x) Splat) + the index search (K1-K5, one C-ABI call, ``coocc_fuser_search``) run EAGERLY on the slot's
  prefetch stream from a helper thread -- they read the frame's own tensors, have one device->host read, and are
  latency-bound on two CUs, hidden under the dense stages of earlier frames;
* the dense stage (G1 -> con_enc -> CustomResNet3D -> FPN3D -> OccHead coarse + fine -> render) is the slot's captured
  ``DenseGraph``: one launch on one of ``dense_streams`` streams.  A frame whose LiDAR sweep is denser than the captured
  capacity takes the eager dense stage (``eager_fallbacks``).

``Ticket.result()`` hands back the graph's output tensors (valid until the slot is reused ``slots`` submits later; ask for
``copy=True`` to own them) after checking the split-f16 engine's range guard.  Everything here is stream-ordered: results are
bit-identical to sequential eager calls (tests/test_gpu_serving.py).
"""
import collections
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch

from . import core, graph as cg, streams as cstreams_

CAM_KEYS = ("rots", "trans", "intrins", "post_rots", "post_trans", "bda")


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class Ticket:
    """One submitted frame.  ``result()`` issues every dense stage up to this frame (if the caller has not pumped them yet),
    optionally waits for it, and returns the output dict."""

    __slots__ = ("pipe", "index", "slot", "frame", "search", "out", "done", "copy", "events", "fallback", "ready", "pts_vol", "stream")

    def __init__(self, pipe, index, slot, frame, copy):
        self.pipe, self.index, self.slot, self.frame, self.copy = pipe, index, slot, frame, copy
        self.search = self.out = self.done = self.events = self.pts_vol = self.stream = None
        self.fallback = False
        # whatever produced the frame's tensors on the submitting thread's stream (the upstream encoders) is waited for by the
        # prefetch stream that reads them
        self.ready = cstreams_.new_event()      # device-scope ordering only (streams.DeviceEvent): no cache writeback per record
        self.ready.record(torch.cuda.current_stream(pipe.dev))

    def result(self, wait=True):
        self.pipe._issue_through(self.index)
        if wait:
            # a device-scope event (streams.DeviceEvent): once it has fired the replay's kernels have completed and released their
            # writes at device scope -- any kernel or device -> host copy issued from now on sees them
            self.done.synchronize()
            core.check_h2_overflow()
        return self.out


# DIAGNOSTIC (profiles/r6_serving_probe_events.txt): bits of the search stage left out of every submitted frame -- 1: the copies of the frame's
# dense-stage inputs into the slot, 2: the pooling (fused lift-splat), 4: the index search (K1-K5).  The outputs are then NOT the frame's.
_DIAG_SKIP = int(os.environ.get("COOCC_SERVING_DIAG_SKIP", "0"))
SLOT_WAIT_HOST = os.environ.get("COOCC_SLOT_WAIT", "host") != "device"     # see ServingPipeline._search


def _clone_frame_item(v):
    if torch.is_tensor(v):
        return v.detach().clone()
    if isinstance(v, (list, tuple)):
        return type(v)(_clone_frame_item(x) for x in v)
    return v


class ServingPipeline:
    """``model``: a ``COOCC_Ray`` in eval mode.  ``example``: one frame (dict, see ``submit``) that fixes the shapes; it is
    used for the warm-up and the capture of every slot.  ``after_done(out, event)`` (optional) is called on the issuing thread with the completion event of the frame's replay
    (what a collective should wait for ON THE HOST: ``dist.GatherThread``); ``after_replay(out)`` (optional) runs on the dense stream right after
    a frame's dense stage has been issued (bench.py issues its RCCL all-gather there)."""

    def __init__(self, model, example, slots=6, dense_streams=3, ahead=0, render=None, search_priority=0, after_replay=None,
                 time_dense=False, after_done=None):
        assert not model.training, "ServingPipeline serves the eval-mode (folded-BN) path"
        self.model = model
        pts = example.get("pts")
        if pts is None:            # raw LiDAR points: the producer (voxelisation + VFE + sparse encoder) runs in the search stage
            with torch.no_grad():
                pts = model.extract_pts_feat(example["points"])[0]
        self.dev = dev = pts.device
        _, _, X, Y, Z = pts.shape
        self.grid = (X, Y, Z)
        self.render = (X >= 100 and Y >= 100 and Z >= 8 and model.use_rendering) if render is None else render
        self.n = n = max(1, int(slots))
        self.ndense = max(1, min(int(dense_streams), max(1, n - 1)))
        # searches in flight ahead of the dense stage: at most slots - dense streams (a slot is rewritten only after the replay
        # that read it finished); `ahead` limits it further (concurrent searches contend with each other and with the graphs)
        most = max(1, n - self.ndense) if n > 1 else 1
        self.ahead = most if ahead <= 0 else max(1, min(int(ahead), most))
        self.after_replay, self.after_done, self.time_dense = after_replay, after_done, time_dense
        # (CU-masked streams -- the FPS chains on reserved CUs, the dense graphs masked away from them -- were built and measured in
        # round 6: 270 -> 44-64 samples/s, profiles/r6_reserve_cus_*.txt; removed.)
        self.dense_streams = [torch.cuda.Stream(device=dev) for _ in range(self.ndense)]
        self.search_streams = [torch.cuda.Stream(device=dev, priority=search_priority) for _ in range(n)]
        self.slots = [cg.make_slot(model, self.grid, dev) for _ in range(n)]
        self.static = [self._make_static(example) for _ in range(n)]
        self.slot_done = [None] * n            # event: the last replay that read slot k
        self.graphs = [None] * n
        self.tpool = ThreadPoolExecutor(self.ahead)
        self.fallbacks = 0
        self.dense_ev = []
        # COOCC_SERVING_TRACE=1: host timestamps of every stage of every ticket (tools/serving_trace.py prints them next to the
        # device-side event times of `time_dense`)
        self.trace = [] if os.environ.get("COOCC_SERVING_TRACE", "0") == "1" else None
        self._lock = threading.Lock()
        self._pending = collections.deque()    # tickets whose search has not been dispatched yet (their slot is still in use)
        self._queue = collections.deque()      # tickets with a dispatched search, dense stage not issued yet
        self._submitted = self._issued = 0
        self._issued_of_slot = [0] * n
        self._dispatched_of_slot = [0] * n
        # a private copy of the example frame for re-captures: the pipeline does not keep the caller's first frame alive, and
        # a caller that rewrites its tensors in place cannot change what a later re-capture warms up on
        self._last_sr = [None] * n
        self._example = {k: _clone_frame_item(v) for k, v in example.items()}
        self.recaptures = 0
        self._capture(example)
        # the captured launches hold raw pointers to the weight packs that were current just now: re-capture when any
        # parameter / buffer of the model has been rewritten or replaced since (checked at every submit)
        self._watch = core.WeightWatch(model)

    # ------------------------------------------------------------------ static inputs of the dense stage
    def _make_static(self, fr):
        dev = self.dev
        cams = fr.get("cams")
        tr = fr["transform"]
        size = tr[-1]
        size = tuple(int(v[0]) if torch.is_tensor(v) else int(v) for v in size)       # image size: fixed per pipeline
        st = dict(img_feats=[torch.empty_like(fr["img_feats"][0], device=dev)],
                  cams=tuple(torch.empty_like(t, device=dev) for t in cams) if cams is not None else None,
                  transform=tuple(torch.empty_like(t, device=dev) if torch.is_tensor(t) else t for t in tr[:-1]) + (size,),
                  gemo=None)
        if cams is None:                       # geometry tensor instead of camera matrices (render reads it)
            st["gemo"] = torch.empty_like(fr["gemo"], device=dev) if fr.get("gemo") is not None else None
        return st

    def _copy_in(self, k, fr):
        """The new frame's dense-stage inputs -> slot k's static tensors (on the current = the slot's prefetch stream)."""
        st = self.static[k]
        st["img_feats"][0].copy_(fr["img_feats"][0], non_blocking=True)
        if st["cams"] is not None:
            for d, s in zip(st["cams"], fr["cams"]):
                d.copy_(s, non_blocking=True)
        for d, s in zip(st["transform"][:-1], fr["transform"][:-1]):
            if torch.is_tensor(d):
                d.copy_(s, non_blocking=True)
        if st["gemo"] is not None:
            st["gemo"].copy_(fr["gemo"], non_blocking=True)

    # ------------------------------------------------------------------ search stage (helper threads)
    def _mark(self, t, tag):
        if self.trace is not None:
            self.trace.append((t.index, tag, time.perf_counter()))

    def _search(self, t):
        k, fr = t.slot, t.frame
        torch.cuda.set_device(self.dev)
        self._mark(t, "search_begin")
        st = self.search_streams[k]
        with torch.cuda.stream(st), torch.no_grad():
            t.ready.wait(st)
            if self.slot_done[k] is not None:
                # the replay that read this slot last.  Waited for by THIS HOST THREAD (a sleeping wait), not by the stream: a
                # second queue waiting for an event of a dense stream that has not fired yet cost the loop 15-20 % with three
                # dense stages in flight -- 274 against 339 samples/s without the search stage, whichever queue waits and
                # whatever the event's fence flags; a record nobody waits for on the device is free
                # (profiles/r6_serving_probe_events.txt).  The thread would block on the stage's count read anyway.
                if SLOT_WAIT_HOST:
                    self.slot_done[k].synchronize()
                else:
                    self.slot_done[k].wait(st)
            if self.time_dense:
                e0 = cstreams_.new_event(timing=True)
                e0.record()
            for v in _tensors_of(fr):
                if v.is_cuda:
                    v.record_stream(st)            # allocated on the caller's stream, read by this one
            diag = _DIAG_SKIP if t.index >= 0 else 0          # (never while capturing: the slots must be filled once)
            if not diag & 1:
                self._copy_in(k, fr)
            pts = fr.get("pts")
            if pts is None:        # coocc_ray.py:215-234: points -> Voxelization -> HardSimpleVFE -> SparseLiDAREnc8x -> [1,C,X,Y,Z]
                pts = self.model.extract_pts_feat(fr["points"])[0]
            t.pts_vol = pts
            self._mark(t, "search_native")
            slot = self.slots[k]
            if diag & 6 and fr.get("depth") is not None and self._last_sr[k] is not None:
                # DIAGNOSTIC ONLY (COOCC_SERVING_DIAG_SKIP, tools/pipeline_gap.py): leave out the pooling (2) and / or the index search
                # (4) of this frame and replay the slot's previous contents -- results are those of the slot's LAST searched frame
                if not diag & 2:
                    self.model.img_view_transformer.lift_splat(fr["depth"], fr["ctx"], cams=fr["cams"], out=slot.img_rows())
                if not diag & 4:
                    sr = self.model.occ_fuser.search_native(pts, slot)
                else:
                    sr = self._last_sr[k]
                    sr.done_main = cstreams_.new_event()
                    sr.done_main.record()
                    sr.done_side = sr.done_main
            elif fr.get("depth") is not None:
                sr = cg.search_into_slot(self.model, slot, fr["depth"], fr["ctx"], fr["cams"], pts)
                self._last_sr[k] = sr
            else:
                # an already-pooled camera volume: into slot 0 of the concat rows, then the same search
                vol = fr["img_voxel_feats"]
                slot.img_rows().as_ncdhw().copy_(vol)
                sr = self.model.occ_fuser.search_native(pts, slot) if cg.NATIVE_SEARCH else \
                    self.model.occ_fuser.search(slot.img_rows().as_ncdhw(), pts, slot=slot)
            if self.time_dense:
                if sr.done_side is not None:
                    sr.done_side.wait(st)
                e1 = cstreams_.new_event(timing=True)
                e1.record()
                t.events = (e0, e1)
        self._mark(t, "search_end")
        return sr

    # ------------------------------------------------------------------ capture
    def _capture(self, example):
        for k in range(self.n):
            t = Ticket(self, -1, k, example, False)
            sr = self._search(t)
            ds = self.dense_streams[k % self.ndense]
            sr.done_main.wait(ds)
            if sr.done_side is not None:
                sr.done_side.wait(ds)
            self.graphs[k] = cg.DenseGraph(self.model, self.slots[k], self.static[k], ds, render=self.render).capture()
        torch.cuda.synchronize(self.dev)
        core.check_h2_overflow()

    # ------------------------------------------------------------------ the loop
    def submit(self, frame, copy=False):
        """``frame``: dict(depth [N,D,fH,fW], ctx [N,C,fH,fW], cams=(rots, trans, intrins, post_rots, post_trans, bda),
        pts [1,C,X,Y,Z], img_feats=[[1,N,512,fH,fW]], transform=img_inputs[1:]) -- or ``img_voxel_feats`` [1,C,X,Y,Z] instead
        of depth / ctx, and / or raw LiDAR ``points`` [n,F] instead of ``pts`` (the LiDAR producer then runs in the search stage).  Shapes as the example's.  Returns a ``Ticket``.  The frame's tensors are read by the prefetched
        search: do not overwrite them before ``Ticket.result()``.  If the model's weights changed since the capture (optimizer
        step, ``load_state_dict``, ``.to()``, ``invalidate_packs``) everything in flight is drained and the slots are captured
        again first -- result tensors of earlier tickets must have been consumed by then."""
        if self._watch.changed():
            self._recapture()
        with self._lock:
            i = self._submitted
            self._submitted += 1
            t = Ticket(self, i, i % self.n, frame, copy)
            self._pending.append(t)
        self._pump()
        return t

    def _pump(self, block_until=None):
        """Dispatch the searches whose slot is free and issue the dense stages whose search has finished (in submit order).
        ``block_until``: ticket index whose dense stage must have been issued on return."""
        while True:
            with self._lock:
                # a ticket's search may start once its slot's previous occupant (n submits earlier) has had its replay ISSUED
                # (the search then waits for that replay's event on the device), and at most `ahead` searches are in flight
                while self._pending and len(self._queue) < self.ahead:
                    t = self._pending[0]
                    if self._dispatched_of_slot[t.slot] > self._issued_of_slot[t.slot]:
                        break
                    self._pending.popleft()
                    self._dispatched_of_slot[t.slot] += 1
                    self._mark(t, "dispatch")
                    t.search = self.tpool.submit(self._search, t)
                    self._queue.append(t)
                head = self._queue[0] if self._queue else None
            if head is None:
                return
            must = block_until is not None and self._issued <= block_until
            if not (must or head.search.done()):
                return
            sr = head.search.result()              # blocks only when the caller asked for this ticket
            self._mark(head, "issue_begin")
            self._issue(head, sr)
            self._mark(head, "issue_end")
            with self._lock:
                self._queue.popleft()
                self._issued += 1
                self._issued_of_slot[head.slot] += 1

    def _issue_through(self, index):
        if self._issued <= index:
            self._pump(block_until=index)
        assert self._issued > index

    def _issue(self, t, sr):
        k = t.slot
        ds = self.dense_streams[k % self.ndense]
        sr.done_main.wait(ds)
        if sr.done_side is not None:
            sr.done_side.wait(ds)
        with torch.cuda.stream(ds), torch.no_grad():
            if self.time_dense:
                e0 = cstreams_.new_event(timing=True)
                e0.record()
            g = self.graphs[k]
            if g.fits(sr.counts):
                out = g.replay()
            else:                                  # a sweep denser than the captured capacity: eager dense stage
                self.fallbacks += 1
                t.fallback = True
                st = self.static[k]
                vf = self.model.occ_fuser(self.slots[k].img_rows().as_ncdhw(), t.pts_vol, search=sr)
                cam_geo = None
                if self.render and st["cams"] is not None:
                    cam_geo = self.model.img_view_transformer._camera_mats(*st["cams"])
                    cam_geo = (cam_geo[0].reshape(-1, cam_geo[0].shape[-1]),) + tuple(cam_geo[1:])
                out = self.model.decode(vf, st["gemo"], st["img_feats"], st["transform"], self.render, cam_geo=cam_geo)
            if self.time_dense:
                e1 = cstreams_.new_event(timing=True)
                e1.record()
                self.dense_ev.append((e0, e1, t.index, t.events))
            if t.copy:                             # own copies of the tensors (lists -- the capacity-sized fine outputs -- by reference)
                out = {kk: (v.clone() if torch.is_tensor(v) else (v.as_ncdhw().clone() if isinstance(v, core.Rows) else v))
                       for kk, v in out.items()}
            if self.after_replay is not None:
                self.after_replay(out)
            ev = cstreams_.new_event(blocking=True)
            ev.record()
        self.slot_done[k] = ev
        if self.after_done is not None:
            self.after_done(out, ev)           # e.g. hand (outputs, completion event) to the thread that issues the rank's collectives
        t.out, t.done, t.stream = out, ev, ds
        t.frame = t.pts_vol = None                 # the search has consumed the frame's tensors (stream-ordered before `ev`)

    def _recapture(self):
        """The model's weights moved under the captured graphs: finish what is in flight (those frames were submitted under the
        old weights and the eager prefetch of a pending one may already have run), drop the graphs and capture again."""
        assert not self.model.training, "ServingPipeline serves the eval-mode (folded-BN) path"
        self.drain()
        torch.cuda.synchronize(self.dev)
        self.graphs = [None] * self.n
        self._capture(self._example)
        self._watch.refresh()
        self.recaptures += 1

    def drain(self):
        """Issue everything submitted so far and wait for it."""
        if self._submitted:
            self._issue_through(self._submitted - 1)
        for ds in self.dense_streams:
            ds.synchronize()
        core.check_h2_overflow()

    def run(self, frames, nsteps, collect=None):
        """Bench / test driver: frame i = frames[i % len(frames)], `ahead` submits in flight.  ``collect(i, out)`` is called in
        order right after frame i's dense stage has been issued (stream-ordered, no host wait)."""
        tickets = collections.deque()
        nxt = 0
        for i in range(nsteps):
            while nxt < nsteps and nxt <= i + self.ahead:
                tickets.append(self.submit(frames[nxt % len(frames)]))
                nxt += 1
            t = tickets.popleft()
            out = t.result(wait=False)
            if collect is not None:
                with torch.cuda.stream(self.dense_streams[t.slot % self.ndense]):
                    collect(i, out)
        for ds in self.dense_streams:
            ds.synchronize()
        core.check_h2_overflow()

    def close(self):
        self.tpool.shutdown(wait=True)
