#!/usr/bin/env python3
"""What the vendor library reaches on THIS box for the plain bf16 GEMMs of the AMP-class modes' shapes (torch.matmul ->
hipBLASLt / rocBLAS, bf16 x bf16 -> bf16 with f32 accumulation), next to which the hand-written kernels' per-launch rates
(bench.py's `kernels` tables, kinds "[bf16, one product ...]") can be read.  A yardstick only: nothing in the product calls
a library GEMM for these shapes (their epilogues - BatchNorm column partials, row dots, bf16 re-materialisation - are fused).

    python tools/bf16_gemm_yardstick.py out.json            # random normal operands (the DVFS guide: never zeros)

Shapes (h = 3072):
  nt  C[M][h] = A[M][h] W[h][h]^T      M = 524288 (eval chunk: 2048 labels x 256 proteins), 262144 (train chunk)
  tn  dW[h][h] = dz[M][h]^T h[M][h]    M = 2^21 rows of the 8.2 M-row pair grid (K = M: the streaming weight gradient)
"""
import json
import os
import sys
import time

import torch


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return {"median_ms": ms[len(ms) // 2], "min_ms": ms[0], "max_ms": ms[-1]}


def main():
    dev = torch.device("cuda:0")
    h = 3072
    g = torch.Generator(device=dev).manual_seed(7)
    out = {"torch": torch.__version__, "hip": torch.version.hip, "prefer_hipblaslt": os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"),
           "device": torch.cuda.get_device_name(0), "cases": {}}
    W = (torch.randn(h, h, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    for M in (524288, 262144):
        A = torch.randn(M, h, device=dev, generator=g).to(torch.bfloat16)
        C = torch.empty(M, h, device=dev, dtype=torch.bfloat16)
        r = timed(lambda: torch.matmul(A, W.t(), out=C), 20)
        r["tflops"] = 2.0 * M * h * h / (r["median_ms"] * 1e-3) / 1e12
        out["cases"][f"nt M={M}"] = r
        del A, C
    M = 1 << 21
    dz = torch.randn(M, h, device=dev, generator=g).to(torch.bfloat16)
    hh = torch.randn(M, h, device=dev, generator=g).to(torch.bfloat16)
    dW = torch.empty(h, h, device=dev, dtype=torch.bfloat16)
    r = timed(lambda: torch.matmul(dz.t(), hh, out=dW), 10)
    r["tflops"] = 2.0 * M * h * h / (r["median_ms"] * 1e-3) / 1e12
    out["cases"][f"tn M={M}"] = r
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
