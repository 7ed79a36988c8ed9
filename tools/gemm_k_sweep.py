"""Per-tile vs per-slab overhead of the f32 LDS-DMA NT kernel: time C[M,3072] = A[M,K] W[3072,K]^T for several K at fixed
M and fit t = a + b K (a: launch + prologue + epilogue per tile wave, b: the slab loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protnote_amd import _lib as L

M, N = 262144 * 2, 3072
dev = "cuda"
res = []
for K in (768, 1536, 3072, 6144):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    def run():
        L.check(L.lib().pn_gemm_nt(L.ptr(A), K, L.ptr(W), K, L.ptr(C), N, M, N, K, None, None, None, None, None, 0, None, 0,
                                   L.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    res.append((K, ms, tf))
    print(K, round(ms, 2), "ms", round(tf, 1), "TF", flush=True)
    del A, W, C
(k1, t1, _), (k2, t2, _) = res[1], res[3]
b = (t2 - t1) / (k2 - k1)
a = t1 - b * k1
tiles_per_cu = (M // 256) * (N // 256) / 256
print("fit: per-launch constant a = %.3f ms (%.1f us per tile), slope -> %.1f TF asymptote" % (a, a / tiles_per_cu * 1e3, 2.0 * M * N / (b * 1e-3) / 1e12))
