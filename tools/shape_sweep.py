"""Off-the-bench-shape sweep: GEMM-engine throughput of a train step (fwd + bwd) and an eval forward
of the full-width head for B in {8, 32, 100, 256} x label rows in {5134, 10268, 32102, 64204} - the reference ships
per-GPU batches of 8 and 32 (configs/base_config.yaml:5-9), EC / GO tables with 1 or 2 descriptions - plus shapes that
leave the fast paths: B % 32 != 0, a ragged label count from in-batch sampling (collators.py:93-96), a ragged last batch.
Per configuration and kernel kind: launches, ms, TFLOP/s, fraction of the 157.3 TFLOP/s f32-MFMA peak.
    python tools/shape_sweep.py out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from protnote_amd import _lib
from protnote_amd.utils.losses import BCEWithLogitsLoss

# PN_SWEEP_MODE=amp: the same sweep (a subset of the shapes) in the full AMP class - bf16x3 base arithmetic, forward_math and
# backward_math = bf16 - priced against the dense bf16 peak
AMP = os.environ.get("PN_SWEEP_MODE", "") == "amp"
PEAK = bench.BF16_MFMA_PEAK_TFLOPS if AMP else bench.F32_MFMA_PEAK_TFLOPS


def table(prof, steps):
    prof = bench.gemm_kinds(prof)
    tot_ms = sum(v[1] for v in prof.values())
    tot_fl = sum(v[2] for v in prof.values())
    kinds = {}
    for k, (cnt, ms, fl) in sorted(prof.items()):
        if ms <= 0:
            continue
        kinds[bench.KIND_NAMES.get(k, str(k))] = {"launches_per_step": cnt / steps, "ms_per_step": round(ms / steps, 3),
                                                  "tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / PEAK, 3),
                                                  "share_of_gemm_time": round(ms / tot_ms, 3)}
    return {"gemm_tflops": round(tot_fl / tot_ms / 1e9, 1), "gemm_frac_of_peak": round(tot_fl / tot_ms / 1e9 / PEAK, 3),
            "gemm_ms_per_step": round(tot_ms / steps, 2), "kinds": kinds}


def main():
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, seed=1)
    model.label_embedding_noising_alpha = 0.0
    if AMP:
        model.math_mode, model.forward_math, model.backward_math = "bf16x3", "bf16", "bf16"
    gen = torch.Generator().manual_seed(3)
    P_all = torch.randn(300, 1100, generator=gen).to(dev)
    lab_all = torch.randn(64204, 1024, generator=gen).to(dev)
    shapes = [(B, NL, "") for B in (8, 32, 100, 256) for NL in (5134, 10268, 32102, 64204)]
    shapes += [(20, 32102, "B % 32 != 0 (ragged last batch of an epoch)"), (250, 32102, "B % 32 != 0"),
               (32, 1789, "ragged N_L (in-batch label sampling)"), (8, 333, "tiny grid"), (300, 5134, "B > 256")]
    if AMP:
        shapes = [(B, NL, "") for B in (8, 32, 256) for NL in (5134, 32102)] + [(8, 64204, "GO x 2 descriptions at the reference's batch"),
                                                                                   (20, 32102, "B % 32 != 0"), (8, 333, "tiny grid")]
    out = {"peak_tflops": PEAK, "mode": "amp (bf16x3 base, forward_math = backward_math = bf16)" if AMP else "f32", "note": "whole GEMM engine (pair grid + row MLPs; the encoder is not run: sequence_embeddings "
                                        "are given), hipEvent-timed per launch", "configs": []}
    for B, NL, note in shapes:
        P_f, lab = P_all[:B].contiguous(), lab_all[:NL].contiguous()
        y = (torch.rand(B, NL, generator=gen) < 2e-3).float().to(dev)
        row = {"B": B, "label_rows": NL, "pair_rows": B * NL, "note": note}
        need_gb = 2 * (B * NL + 300000) * 3072 * 4 / 1e9
        if need_gb > 60:
            # a big activation store is about to be allocated: hand every cached block (the previous case's store and
            # workspaces, whatever their size) back to the driver FIRST.  Round 4 freed only after big cases, so 250 x 32 102
            # (190 GB) came after 20 x 32 102 (23 GB, not freed) and died of fragmentation - the record held an OOM string
            # where DESIGN quoted a number.
            import protnote_amd

            model.__dict__.pop("_pn_train_save", None)
            protnote_amd.free_workspaces()
            torch.cuda.empty_cache()
        try:
            if need_gb < 230:
                model.train()
                steps = 2 if B * NL > 2e6 else 4
                for it in range(steps + 1):
                    if it == 1:
                        torch.cuda.synchronize()
                        _lib.prof_begin()
                        t0 = time.time()
                    for q in model.parameters():
                        q.grad = None
                    logits, _ = model(sequence_embeddings=P_f, label_embeddings=lab)
                    BCEWithLogitsLoss()(logits, y).backward()
                torch.cuda.synchronize()
                dt = (time.time() - t0) / steps
                row["train"] = {"ms_per_step": round(dt * 1e3, 2), "pairs_per_s": round(B * NL / dt), **table(_lib.prof_end(), steps)}
                del logits
            else:
                row["train"] = {"skipped": f"activation store {need_gb:.0f} GB does not fit one GPU"}
            model.eval()
            steps = 2 if B * NL > 2e6 else 4
            with torch.no_grad():
                for it in range(steps + 1):
                    if it == 1:
                        torch.cuda.synchronize()
                        _lib.prof_begin()
                        t0 = time.time()
                    model(sequence_embeddings=P_f, label_embeddings=lab)
                torch.cuda.synchronize()
                dt = (time.time() - t0) / steps
            row["eval"] = {"ms_per_forward": round(dt * 1e3, 2), "pairs_per_s": round(B * NL / dt), **table(_lib.prof_end(), steps)}
        except Exception as e:  # noqa: BLE001
            row["error"] = str(e)[:300]
        out["configs"].append(row)
        tr = row.get("train", {})
        print(f"B={B:4d} rows={NL:6d} train {tr.get('gemm_frac_of_peak')} ({tr.get('ms_per_step')} ms)  "
              f"eval {row.get('eval', {}).get('gemm_frac_of_peak')} ({row.get('eval', {}).get('ms_per_forward')} ms)  {note} {row.get('error', '')}",
              flush=True)
        if need_gb > 60:  # hand the big activation store back before the next (differently sized) one is allocated
            import protnote_amd

            model.__dict__.pop("_pn_train_save", None)
            protnote_amd.free_workspaces()
            torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
