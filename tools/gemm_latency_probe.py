"""What the LDS-DMA of the f32 all-DMA NT loop costs, and why: the same GEMM (M = 524288, N = K = 3072) with
  (a) real operands (A streams from HBM, W from L2 / Infinity Cache),
  (b) lda = 0: every tile row of A is the same 12 KB row -> A always hits in L2 (no HBM latency, same LDS-side work),
  (c) lda = ldw = 0: both operands from one cache line set.
If (b)/(c) run at the no-DMA ceiling (150 TFLOP/s) the loss is memory-latency tails (deeper prefetch
helps); if they stay at (a)'s rate it is the LDS-side cost of the DMA itself (it does not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protnote_amd import _lib as L

M, N, K = 262144 * 2, 3072, 3072
dev = "cuda"
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
for name, lda, ldw in (("real", K, K), ("lda=0", 0, K), ("lda=ldw=0", 0, 0), ("real", K, K)):
    def run():
        L.check(L.lib().pn_gemm_nt(L.ptr(A), lda, L.ptr(W), ldw, L.ptr(C), N, M, N, K, None, None, None, None, None, 0, None, 0,
                                   L.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(name, round(ms, 2), "ms", round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1), "TF", flush=True)
