#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, as the
MI355X guide prescribes) -> JSON keyed by the kernel names bench.py uses.
    python tools/make_traffic.py FETCH.csv WRITE.csv [config] > profiles/r2_traffic.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced reads as 64 B, so the raw
value is doubled (guide, section HBM)."""
import collections
import csv
import json
import sys

ALIASES = {   # bench.py region name -> kernel symbol prefix in the rocprofv3 trace
    "k_conv2<160,wg> wino": "void k_conv2<160, 2, true, 2, false>",
    "k_conv2p wino": "void k_conv2p<true, false>",
    "k_conv2<128,wg> wino": "void k_conv2<128, 1, true, 3, false>",
    "k_conv2<128>": "void k_conv2<128, 3, false, 2, false>",
    "k_conv2<160>": "void k_conv2<160, 2, false, 2, false>",
    "k_conv2p": "void k_conv2p<false, true>",
    "k_conv_bf16": "k_conv_bf16",
    # the trace carries all template arguments ("void k_gemm_h2z<3, false, 0, 3>"): prefixes end BEFORE the next argument
    "k_gemm_h2z wino": "void k_gemm_h2z<3, false,",
    "k_gemm_h2z direct": "void k_gemm_h2z<3, true,",
    "k_gemm_h2w": "void k_gemm_h2w<false,",
    "k_gemm_h2w table": "void k_gemm_h2w<true,",
}


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        a = agg[r["Kernel_Name"].split("(")[0]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    cfg = sys.argv[3] if len(sys.argv) > 3 else "r50"

    def per_launch(sym):
        f = [(c, v) for k, (c, v) in fetch.items() if k.startswith(sym)]
        w = [(c, v) for k, (c, v) in write.items() if k.startswith(sym)]
        if not f or not w:
            return None
        fc, fv = sum(c for c, _ in f), sum(v for _, v in f)
        wc, wv = sum(c for c, _ in w), sum(v for _, v in w)
        return dict(fetch_KB_raw_per_launch=round(fv / fc, 1), write_KB_per_launch=round(wv / wc, 1),
                    bytes_per_launch=int(2 * 1024 * fv / fc + 1024 * wv / wc), launches_in_pass=fc)
    out = {}
    for alias, sym in ALIASES.items():
        d = per_launch(sym)
        if d:
            out[alias] = dict(d, symbol=sym)
    rays, ups = per_launch("void k_render_rays_geo") or per_launch("void k_render_nearest"), per_launch("k_upsample_maps")
    if rays and ups:
        out["k_render_nearest+k_upsample_maps"] = {cfg: dict(bytes_per_launch=rays["bytes_per_launch"] + ups["bytes_per_launch"],
                                                             rays=rays, upsample=ups)}
    # the pooling call (roofline_pool): its four launches together, per call
    parts = [per_launch("void " + sym) or per_launch(sym) for sym in ("k_seg_hist", "k_scan_local", "k_csr_fill", "k_pool_sum_seg")]
    if all(parts):            # round 5: the ray-segment form
        out["coocc_lift_splat_cams"] = {cfg: dict(bytes_per_launch=sum(d["bytes_per_launch"] for d in parts),
                                                   seg_hist=parts[0], scan_local=parts[1], csr_fill=parts[2], pool_sum_seg=parts[3])}
    else:
        parts = [per_launch("void " + sym) or per_launch(sym) for sym in ("k_keys_hist", "k_scan_local", "k_csr_fill", "k_pool_sum_csr<true>")]
        if all(parts):
            out["coocc_lift_splat_cams"] = {cfg: dict(bytes_per_launch=sum(d["bytes_per_launch"] for d in parts),
                                                       keys_hist=parts[0], scan_local=parts[1], csr_fill=parts[2], pool_sum_csr=parts[3])}
    for sym in ("k_wino_in_h2<6>", "k_wino_in_h2<4>", "k_rows_to_h2", "k_wino_in<6>", "k_wino_out<6>", "k_wino_in<4>", "k_wino_out<4>", "k_pool_sum_csr<true>", "k_key_hist", "k_fuser_prepare_rows",
                "k_fine_mlp<true>", "k_fine_sample_img_grp", "k_fine_sample_voxel_r2", "k_fine2_h2", "k_gemm_h2p", "k_pool_sum_seg", "k_seg_hist"):
        d = per_launch("void " + sym) or per_launch(sym)
        if d:
            out[sym] = d
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
