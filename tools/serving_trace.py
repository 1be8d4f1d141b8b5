"""Host and device timeline of the serving loop, ticket by ticket (no profiler attached).

COOCC_SERVING_TRACE=1 python tools/serving_trace.py [config] [frames] [slots] [dense_streams] [ahead]
Device times from HIP events (time_dense), host times from perf_counter, both relative to one synchronised origin.
"""
import os
import sys
import time

os.environ["COOCC_SERVING_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "r50"
    nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    nds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    ahead = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dev = torch.device("cuda:0")
    model, _ = bench.build_model(cfg, dev)
    samples = [bench.make_inputs(cfg, 1234 + i, dev, model) for i in range(slots)]
    frames = [bench.frame_of(x) for x in samples]
    gp = model.serving(frames[0], slots=slots, dense_streams=nds, ahead=ahead)
    gp.run(frames, 2 * slots)
    gp.time_dense = True
    gp.run(frames, slots)
    torch.cuda.synchronize()
    gp.trace.clear()
    gp.dense_ev.clear()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    base.synchronize()
    h0 = time.perf_counter()
    first = gp._submitted
    gp.run(frames, nfr)
    torch.cuda.synchronize()
    wall = time.perf_counter() - h0
    print("%s: %d frames, slots %d, dense streams %d, searches ahead %d: %.3f ms per frame" % (cfg, nfr, slots, nds, gp.ahead, 1e3 * wall / nfr))
    host = {}
    for idx, tag, ts in gp.trace:
        host.setdefault(idx, {})[tag] = 1e3 * (ts - h0)
    print("ticket | host: dispatch  search_begin  native_call  search_end  issue_begin  issue_end | device: search start..end   dense start..end")
    for e0, e1, idx, sev in gp.dense_ev:
        h = host.get(idx, {})
        g = lambda k: "%8.2f" % h[k] if k in h else "    --  "
        s0, s1 = (base.elapsed_time(sev[0]), base.elapsed_time(sev[1])) if sev else (float("nan"),) * 2
        print("%5d  | %s %s %s %s %s %s | %8.2f .. %8.2f   %8.2f .. %8.2f  (%.2f)" % (
            idx - first, g("dispatch"), g("search_begin"), g("search_native"), g("search_end"), g("issue_begin"), g("issue_end"),
            s0, s1, base.elapsed_time(e0), base.elapsed_time(e1), e0.elapsed_time(e1)))


if __name__ == "__main__":
    main()
