#!/usr/bin/env python3
"""Aggregate the rocprofv3 counter passes written by tools/profile_round.sh into per-kernel HBM-side traffic and
matrix-pipe utilisation (run on the GPU box right after the passes; the CSVs are too large to travel).

    python tools/pmc_traffic.py gpurun_out/prof_r02 r02  ->  gpurun_out/prof_r02/r02_hbm_traffic.json

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KB by the L2's
fabric-side request counters (Infinity-Cache hits included); on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, i.e. HALF the bytes - it is doubled here.  Per launch: bytes = 2 * FETCH_SIZE + WRITE_SIZE.
Matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_pass(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = defaultdict(lambda: defaultdict(float))   # (kernel, dispatch) -> counter -> value
    dur = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (r["Kernel_Name"], r["Dispatch_Id"])
                rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
                dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return rows, dur


def short(name):
    name = name.replace("void pn::", "").replace("pn::", "")
    return name[:name.index("(")] if "(" in name else name


def main(out_dir, tag):
    big = lambda k: ("gemm_nt" in k or "gemm_tn" in k)
    res = {"per_kernel": {}, "note": __doc__.split("Conventions")[1].strip()}
    fetch, _ = read_pass(os.path.join(out_dir, "pmc_FETCH_SIZE"))
    write, _ = read_pass(os.path.join(out_dir, "pmc_WRITE_SIZE"))
    agg = defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
    for (k, _), c in fetch.items():
        if big(k):
            agg[short(k)]["launches"] += 1
            agg[short(k)]["fetch_kb"] += c.get("FETCH_SIZE", 0.0)
    wl = defaultdict(int)
    for (k, _), c in write.items():
        if big(k):
            agg[short(k)]["write_kb"] += c.get("WRITE_SIZE", 0.0)
            wl[short(k)] += 1
    sq, dur = read_pass(os.path.join(out_dir, "pmc_SQ"))
    util = defaultdict(lambda: {"n": 0, "busy": 0.0, "gui": 0.0, "sec": 0.0, "valu": 0.0, "mfma": 0.0})
    for key, c in sq.items():
        k = key[0]
        if big(k):
            u = util[short(k)]
            u["n"] += 1
            u["busy"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            u["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
            u["valu"] += c.get("SQ_INSTS_VALU", 0.0)
            u["mfma"] += c.get("SQ_INSTS_MFMA", 0.0)
            u["sec"] += dur[key]
    # full pair-grid launches, classified PER DISPATCH (the passes replay the same launch sequence, so Dispatch_Id pairs the
    # FETCH_SIZE pass with the WRITE_SIZE pass): a kernel name also covers the small row-MLP launches and the 1/32-grid
    # chunks of the top layer's dh GEMM, which must not dilute the per-launch figure
    wr = {key: c.get("WRITE_SIZE", 0.0) for key, c in write.items()}
    full = defaultdict(list)
    for key, c in fetch.items():
        k = key[0]
        if big(k) and "bf16x3" not in k:
            b = (2 * c.get("FETCH_SIZE", 0.0) + wr.get(key, 0.0)) * 1024.0
            if b >= 100e9:
                full[short(k)].append(b)
    total_bytes = sum(sum(v) for v in full.values())
    total_launch = sum(len(v) for v in full.values())
    res["full_grid_launches"] = {k: {"launches": len(v), "hbm_side_GB_per_launch": sum(v) / len(v) / 1e9}
                                 for k, v in sorted(full.items())}
    for k, a in sorted(agg.items()):
        n = max(a["launches"], 1)
        per = (2 * a["fetch_kb"] / n + a["write_kb"] / max(wl[k], 1)) * 1024.0
        e = {"launches": a["launches"], "fetch_kb_per_launch": a["fetch_kb"] / n,
             "write_kb_per_launch": a["write_kb"] / max(wl[k], 1), "hbm_side_bytes_per_launch": per}
        u = util.get(k)
        if u and u["gui"] > 0:
            e["mfma_busy"] = u["busy"] / (u["gui"] / 8.0 * 1024.0)
            e["clock_ghz"] = u["gui"] / 8.0 / u["sec"] / 1e9 if u["sec"] > 0 else None
            e["valu_insts_per_mfma"] = u["valu"] / u["mfma"] if u["mfma"] > 0 else None
            e["avg_ms"] = u["sec"] / u["n"] * 1e3
        res["per_kernel"][k] = e
    # the HBM-bound streaming passes of the step (bench.py `stages`): measured fabric-side bytes of their pair-grid launches
    # (>= 1 ms; the same kernels also run on the small row-MLP tensors) next to the algorithmic bytes bench.py prices them with
    h, Rr = 3072.0, 256.0 * 32102.0
    stage_kernels = {
        "k_dz_apply<1>": ("dz in place, top layer", Rr * (8 * h + 4)),
        "k_dz_apply<0>": ("dz in place, inner layer", Rr * 12 * h),
        "k_dz_apply_bf16<1, 4>": ("dz in place as bf16, top layer (bf16 backward)", Rr * (6 * h + 4)),
        "k_dz_apply_bf16<0, 4>": ("dz in place as bf16, inner layer (bf16 backward)", Rr * 10 * h),
        "k_bn_bwd_stats<1, 0>": ("BatchNorm-backward statistics, top layer", Rr * (4 * h + 4)),
        "k_bn_bwd_stats<0, 0>": ("BatchNorm-backward statistics, inner layer", Rr * 8 * h),
        "k_pair_mask_reduce_fused": ("layer-1 masked reduction", Rr * 4 * h),
        "k_rowdot_rows_reg": ("row-dot logits", Rr * (4 * h + 4)),
    }
    _, fdur = read_pass(os.path.join(out_dir, "pmc_FETCH_SIZE"))
    st = {}
    for sub, (what, alg) in stage_kernels.items():
        xs = []
        for key, c in fetch.items():
            if sub in key[0] and fdur.get(key, 0.0) >= 1e-3:
                xs.append((2 * c.get("FETCH_SIZE", 0.0) + wr.get(key, 0.0)) * 1024.0)
        if xs:
            m = sum(xs) / len(xs)
            st[sub] = {"stage": what, "launches": len(xs), "hbm_side_GB_per_launch": m / 1e9,
                       "algorithmic_GB_per_launch": alg / 1e9, "ratio": m / alg}
    res["stages"] = st
    res["stages_note"] = ("pair-grid launches (>= 1 ms) of the streaming passes at the bench shape (B = 256, N_L = 32102, h = 3072): "
                          "2 * FETCH_SIZE + WRITE_SIZE per launch against the algorithmic bytes of bench.py's `stages` block")
    res["bytes_per_launch"] = total_bytes / total_launch if total_launch else None
    res["bytes_per_launch_definition"] = ("mean over the full-pair-grid f32 GEMM launches (dispatches moving >= 100 GB) of "
                                          "2 * FETCH_SIZE + WRITE_SIZE; algorithmic bytes of such a launch: 202 GB")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protnote_amd.build import csrc_hash

    res["csrc_hash"] = csrc_hash()  # bench.py marks `traffic_stale` when the kernels have changed since
    path = os.path.join(out_dir, f"{tag}_hbm_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(path, res["bytes_per_launch"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02")
