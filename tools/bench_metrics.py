"""Evaluation-metric pass at evaluation-set scale on one MI355X: N proteins x 32 102 labels of resident scores ->
exact per-label AP + micro AP (sort-based) and the 50-threshold binned estimate; HIP-event timings, algorithmic bytes
(5 B/pair read by the sort's first pass ... see DESIGN.md) and a bounded CPU sample of the same work beside it.
    python tools/bench_metrics.py [N]        # default 50 000
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protnote_amd.utils.evaluation import DeviceAveragePrecision, DeviceBinnedAUPRC

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
NL, B = 32102, 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
acc = DeviceAveragePrecision(NL, N, dev)
binned = DeviceBinnedAUPRC(NL, dev, threshold=50)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


t_app = t_bin = 0.0
sample_p, sample_y = [], []
for i in range(0, N, B):
    b = min(B, N - i)
    logits = torch.randn(b, NL, device=dev, generator=g) * 3 - 4
    y = torch.rand(b, NL, device=dev, generator=g) < torch.sigmoid(logits - 1)
    p = torch.sigmoid(logits)
    e0 = ev(); acc.update(p, y); e1 = ev(); binned.update(p, y); e2 = ev()
    torch.cuda.synchronize()
    t_app += e0.elapsed_time(e1); t_bin += e1.elapsed_time(e2)
    sample_p.append(p[:, :64].cpu()); sample_y.append(y[:, :64].cpu())
for _ in range(2):
    e0 = ev(); out = acc.compute(micro=False); e1 = ev(); out_m = acc.compute(micro=True); e2 = ev()
    bo = binned.compute(); e3 = ev()
    torch.cuda.synchronize()
t_label, t_both, t_binc = e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)
pairs = N * NL
# CPU beside it: sklearn-equivalent numpy AP on 64 labels (single thread), scaled to the label set
from oracle import metrics_oracle as MO
sp, sy = torch.cat(sample_p).numpy(), torch.cat(sample_y).numpy()
t0 = time.time()
cpu = [MO.average_precision_fast(sp[:, j], sy[:, j]) for j in range(64)]
t_cpu = (time.time() - t0) * NL / 64
np.testing.assert_allclose(out["ap_per_label"][:64].cpu().numpy(), cpu, rtol=1e-12)
print(json.dumps({
    "pairs": pairs, "N": N, "N_L": NL,
    "append_ms_total": round(t_app, 1), "append_GBps": round(pairs * 10 / t_app / 1e6, 1),
    "binned_update_ms_total": round(t_bin, 1), "binned_update_GBps": round(pairs * 5 / t_bin / 1e6, 1),
    "exact_per_label_ms": round(t_label, 1), "exact_per_label_plus_micro_ms": round(t_both, 1),
    "binned_compute_ms": round(t_binc, 2),
    "per_label_pairs_per_s": round(pairs / t_label * 1e3), "cpu_per_label_s_1thread_scaled": round(t_cpu, 1),
    "map_macro": out["map_macro"], "map_micro": out_m["map_micro"], "binned_macro": bo["map_macro"],
    "binned_micro": bo["map_micro"], "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
