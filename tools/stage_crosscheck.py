#!/usr/bin/env python3
"""Cross-check bench.py's event-timed `stages` and GEMM kinds against the rocprofv3 kernel trace of the SAME run
(tools/profile_round.sh writes both: <dir>/bench_stats.json and <dir>/stats/p_kernel_trace.csv).

    python tools/stage_crosscheck.py gpurun_out/prof_r04 r04   ->   profiles/r04_stage_crosscheck.json

A stage is one or more kernels per step; launches of a kernel that also serves small tensors (k_dz_apply<0> and
k_bn_bwd_stats<0, 0> run in the row MLPs too) are told apart by duration: the pair-grid launches are the ones above
1 ms.  The trace covers warm-up + timed steps of both math modes of that command (f32, then bf16x3); streaming passes
are the same code in both, so all of their launches are averaged.  Event timings bracket the launches of a stage on
the stream, so for stages of a few tens of microseconds they carry the launch gaps that the per-dispatch durations do not."""
import csv
import json
import os
import sys
from collections import defaultdict


def main(d, tag):
    bench = json.loads(open(os.path.join(d, "bench_stats.json")).read().strip().splitlines()[-1])
    per = defaultdict(list)
    with open(os.path.join(d, "stats", "p_kernel_trace.csv"), newline="") as fh:
        for r in csv.DictReader(fh):
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)

    def ms(sub, big=None):
        xs = [x for k, v in per.items() if sub in k for x in v]
        if big is True:
            xs = [x for x in xs if x > 1.0]
        return sum(xs) / max(len(xs), 1), len(xs)

    # stage name prefix -> list of (kernel substring, pair-grid launches only?, launches of that kernel per stage launch)
    recipe = {
        "K2 ": [("k_onehot_ids", None), ("k_conv1_gather", None), ("k_ncl_to_nlc", None)],
        "K6 ": [("k_pool(", None)],
        "K13/K14": [("k_loss(", None)],
        "K16 ": [("k_sumsq", None), ("k_adam", None)],
        "BatchNorm-backward": None,  # two shapes per step, see below
        "dz in place": None,
        "layer-1 masked": [("k_pair_mask_reduce_fused", None)],
        "row-dot": [("k_rowdot_rows_reg", None)],
        "conv operand staging": [("k_conv_stage_act", None)],
    }
    out = {"source": f"{d}/bench_stats.json (hipEvent timings inside the timed region) vs {d}/stats/p_kernel_trace.csv "
                     "(rocprofv3 --kernel-trace of the same process)", "stages": {}}
    for name, st in bench["stages"].items():
        ev = st["ms_per_launch"]
        key = next((k for k in recipe if name.startswith(k)), None)
        if key is None:
            continue
        if recipe[key] is None:
            if key.startswith("dz"):
                top, n1 = ms("k_dz_apply<1>", True)
                inner, n0 = ms("k_dz_apply<0>", True)
            else:
                top, n1 = ms("k_bn_bwd_stats<1, 0>", True)
                inner, n0 = ms("k_bn_bwd_stats<0, 0>", True)
            tr = (top + inner) / 2  # one top-layer and one inner-layer launch per step
            parts = {"top_layer_ms": top, "inner_layer_ms": inner, "launches": [n1, n0]}
        else:
            tr, parts = 0.0, {}
            for sub, big in recipe[key]:
                m, n = ms(sub, big)
                tr += m
                parts[sub] = {"avg_ms": m, "launches": n}
        nbytes = st["algorithmic_bytes_per_launch"]
        out["stages"][name] = {
            "event_ms_per_launch": ev, "trace_ms_per_launch": tr, "trace_over_event": tr / ev if ev else None,
            "event_TBps": st["achieved_TBps"], "trace_TBps": nbytes / (tr * 1e-3) / 1e12 if tr else None, "kernels": parts}
    # the GEMM family: per-kind event totals vs the trace's per-kernel averages (full-grid launches: > 500 ms in f32)
    fam = {}
    for sub in ("gemm_nt_dma_kernel<0, 0", "gemm_nt_dma_kernel<1, 0", "gemm_nt_dma_kernel<2, 0", "gemm_tn_fast_kernel<1",
                "gemm_tn_fast_kernel<2"):
        xs = [x for k, v in per.items() if sub in k for x in v if x > 500.0]
        if xs:
            fam[sub + ">"] = {"full_grid_launches": len(xs), "avg_ms": sum(xs) / len(xs),
                              "tflops": 2 * 256 * 32102 * 3072.0 * 3072.0 / (sum(xs) / len(xs) * 1e-3) / 1e12}
    out["f32_family_full_grid_launches"] = fam
    out["bench_line"] = {k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}
    out["bench_line"]["roofline"] = {k: bench["roofline"][k] for k in ("achieved", "frac", "avg_ms_per_launch", "launches")}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_stage_crosscheck.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["stages"].items():
        print(f"{k[:48]:50s} event {v['event_ms_per_launch']:9.4f} ms  trace {v['trace_ms_per_launch']:9.4f} ms  "
              f"ratio {v['trace_over_event']:.3f}  {v['event_TBps']:.2f} / {v['trace_TBps']:.2f} TB/s")
    for k, v in fam.items():
        print(k, v)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r04")
