#!/usr/bin/env python3
"""Per-kind GEMM timings of the AMP-class train step (bf16x3 forward + bf16 backward) at the bench shape, for A/B builds of a
compile-time experiment macro:

    python -m protnote_amd.build && python tools/tn_tasks_ab.py out_a.json
    PN_EXTRA_HIPCC_FLAGS=-DPN_TN_TASKS_T=1 python -m protnote_amd.build && PN_EXTRA_HIPCC_FLAGS=-DPN_TN_TASKS_T=1 python tools/tn_tasks_ab.py out_b.json

The record carries the binary's source hash (which covers the extra flags) and the flags, so a stale binary cannot stand in for
the experiment (round 4's "transposed region tasks" null result was very likely measured on an unchanged .so)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from protnote_amd import _lib, build
from protnote_amd.models.ProtNoteTrainer import train_step
from protnote_amd.models.train_path import head_parameters
from protnote_amd.utils.losses import get_loss
from protnote_amd.utils.optim import FusedClipAdam


def main():
    dev = torch.device("cuda:0")
    steps = int(os.environ.get("PN_STEPS", "6"))
    model = bench.build_model(dev).train()
    model.math_mode, model.backward_math = os.environ.get("PN_AB_FORWARD", "bf16x3"), os.environ.get("PN_AB_BACKWARD", "bf16")
    model.forward_math = os.environ.get("PN_AB_FORWARD_MATH", "same")
    if "PN_AB_MFMA16" in os.environ:  # run-time A/B of the one-product NT kernels' MFMA shape (round 6)
        _lib.check(_lib.lib().pn_set_bf16_mfma16(int(os.environ["PN_AB_MFMA16"])))
    opt = FusedClipAdam(list(head_parameters(model)), lr=3e-4, max_norm=1.0)
    loss_fn = get_loss({"params": {"LOSS_FN": "BCE"}}, bce_pos_weight=torch.tensor(1.0))
    batch = bench.synthetic_batch(256, 512, 32102, dev, seed=1000)
    for _ in range(2):
        loss = train_step(model, loss_fn, opt, batch)
    torch.cuda.synchronize()
    _lib.prof_begin()
    t0 = time.time()
    for _ in range(steps):
        loss = train_step(model, loss_fn, opt, batch)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    prof = _lib.prof_end()
    # a checksum of every gradient-updated weight after the timed steps: A/B variants that claim bit-identity must agree on it
    checksum = float(opt.flat_w.double().sum().item())
    out = {"build_hash": _lib.build_hash(), "flat_w_checksum": repr(checksum),
           "modes": {"math": model.math_mode, "forward_math": model.forward_math, "backward_math": model.backward_math}, "csrc_hash_now": build.csrc_hash(), "extra_flags": build._extra(), "bf16_mfma16": os.environ.get("PN_AB_MFMA16", "default (1)"),
           "steps": steps, "ms_per_step": dt * 1e3, "final_loss": float(loss),
           "per_launch_ms": {bench.KIND_NAMES.get(k, str(k)): round(v[1] / v[0], 3) for k, v in sorted(bench.gemm_kinds(prof).items())
                             if v[0] > 0 and v[2] / v[0] > 1e12},
           "kernels": bench.kernel_table(prof)}
    assert out["build_hash"] == out["csrc_hash_now"]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
