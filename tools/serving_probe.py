"""ServingPipeline alone (no bench.py around it): frames through model.serving() with the knobs of profiles/r6_pipeline_gap.txt.

    python tools/serving_probe.py [slots] [dense_streams] [steps]        env: COOCC_SERVING_DIAG_SKIP, COOCC_SERVING_PROBE=a,b,...
      probe flags:  recnowait / dummywait -- an event is recorded after every replay but the search stream does not wait for it
                                  (nobody / an idle stream does); with DIAG_SKIP=7
                    nowait   -- _issue does not make the dense stream wait for the search's event (only valid with DIAG_SKIP=7)
                    noev     -- no event record after a replay (slot_done reuses the previous event; only valid with DIAG_SKIP=7)
                    prequeue -- every search is dispatched and finished before the first replay is issued (host-side ordering)
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import serving

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
flags = set(filter(None, os.environ.get("COOCC_SERVING_PROBE", "").split(",")))
dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
samples = [bench.make_inputs("r50", 1234 + i, dev, model) for i in range(slots)]
frames = [bench.frame_of(s) for s in samples]
gp = model.serving(frames[0], slots=slots, dense_streams=nds)
if flags & {"nowait", "noev", "recnowait", "dummywait", "hostsync"}:
    orig = gp._issue
    keep = []
    dummy = torch.cuda.Stream(device=dev)

    def _issue(t, sr):
        k = t.slot
        ds = gp.dense_streams[k % gp.ndense]
        if "nowait" not in flags:
            sr.done_main.wait(ds)
        with torch.cuda.stream(ds), torch.no_grad():
            out = gp.graphs[k].replay()
            from co_occ_amd import streams as cstreams
            if gp.slot_done[k] is None:
                ev = cstreams.new_event()
                ev.record()
            elif "noev" in flags:
                ev = gp.slot_done[k]                      # no record at all
            elif "recnowait" in flags:
                ev2 = cstreams.new_event()
                ev2.record()                              # a record nobody waits for; the search keeps waiting for the first event
                keep.append(ev2)
                del keep[:-64]
                ev = gp.slot_done[k]
            elif "dummywait" in flags:
                ev2 = cstreams.new_event()
                ev2.record()
                ev2.wait(dummy)                           # an idle third stream waits for it, not the search stream
                keep.append(ev2)
                del keep[:-64]
                ev = gp.slot_done[k]
            else:
                ev = cstreams.new_event()
                ev.record()
        gp.slot_done[k] = ev
        t.out, t.done = out, ev
        t.frame = t.pts_vol = None
    gp._issue = _issue
gp.run(frames, 2 * slots)
torch.cuda.synchronize()
res = []
for w in range(3):
    t0 = time.perf_counter()
    gp.run(frames, steps)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
print("serving alone: slots %d dense streams %d DIAG_SKIP=%s probe=%s: %s ms per sample -> %.1f samples/s" % (
    slots, nds, os.environ.get("COOCC_SERVING_DIAG_SKIP", "0"), ",".join(sorted(flags)) or "-", ["%.3f" % r for r in res], 1e3 / sorted(res)[1]),
    flush=True)
gp.close()
