"""Where the pipelined step's time goes: occupancy of the GPU over the timed region of a bench run, from a rocprofv3 kernel trace.

usage: python tools/pipeline_gaps.py <kernel_trace.csv> [window_ms]
Takes the LAST window_ms (default 60) of the trace that still contains k_gemm_h2z launches (the timed steps; the CPU baseline and
the eager roofline passes come after / before), and prints: wall time with no kernel resident, with only the single-workgroup FPS
kernel resident, with exactly 1 / 2 / 3+ "wide" kernels resident; the same per queue; and the ten largest idle gaps with the
kernels on either side.
"""
import csv
import sys


def main():
    path = sys.argv[1]
    window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    # the steady state of the pipeline: the densest stretch of k_gemm_h2z<3, false launches
    big = [s for s, e, n, q in rows if "k_gemm_h2z<3, false" in n]
    if not big:
        print("no k_gemm_h2z launches in the trace")
        return
    best, lo = 0, 0
    j = 0
    for i, s in enumerate(big):
        while big[i] - big[j] > window:
            j += 1
        if i - j + 1 > best:
            best, lo = i - j + 1, big[j]
    hi = lo + window
    if len(sys.argv) > 3:
        timeline(path, lo + int(window / 2), 10e6, sys.argv[3])
    sel = [(max(s, lo), min(e, hi), n, q) for s, e, n, q in rows if e > lo and s < hi]
    print("window %.1f ms, %d kernel launches, %d Winograd GEMMs (= %.1f samples at 13 per sample -> %.3f ms per sample)"
          % (window / 1e6, len(sel), best, best / 13.0, window / 1e6 / (best / 13.0)))
    ev = []
    for s, e, n, q in sel:
        narrow = "k_fps_voxels" in n
        ev.append((s, 1, narrow))
        ev.append((e, -1, narrow))
    ev.sort()
    wide = fps = 0
    t = lo
    hist = {}
    for ts, d, narrow in ev:
        key = "idle" if wide == 0 and fps == 0 else "fps only" if wide == 0 else "%d wide" % min(wide, 4)
        hist[key] = hist.get(key, 0) + ts - t
        t = ts
        if narrow:
            fps += d
        else:
            wide += d
    hist["idle"] = hist.get("idle", 0) + hi - t
    for k in ("idle", "fps only", "1 wide", "2 wide", "3 wide", "4 wide"):
        print("  %-9s %6.2f %%" % (k + ("+" if k == "4 wide" else ""), 100.0 * hist.get(k, 0) / window))
    # sum of kernel time by name inside the window, per sample
    per = {}
    for s, e, n, q in sel:
        per[n[:60]] = per.get(n[:60], 0) + e - s
    nsamp = best / 13.0
    tot = sum(per.values())
    print("kernel time per sample: %.3f ms (sum over all queues)" % (tot / 1e6 / nsamp))
    for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-62s %7.1f us" % (n, v / 1e3 / nsamp))
    # per queue busy
    qs = {}
    for s, e, n, q in sel:
        qs.setdefault(q, []).append((s, e))
    for q, iv in sorted(qs.items()):
        iv.sort()
        busy, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print("  queue %-4s busy %5.1f %%  launches %d" % (q, 100.0 * busy / window, len(iv)))


def timeline(path, lo, span, out):
    """every launch that starts inside [lo, lo + span): start (us from lo), queue, duration, name"""
    with open(path) as f, open(out, "w") as g:
        rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                       r.get("Workgroup_Size", ""), r.get("Grid_Size", "")) for r in csv.DictReader(f))
        for s, e, n, q, wg, grid in rows:
            if lo <= s < lo + span:
                g.write("%9.1f  q%-3s %8.1f us  grid %-9s %s\n" % ((s - lo) / 1e3, q, (e - s) / 1e3, grid, n[:70]))


if __name__ == "__main__":
    main()
