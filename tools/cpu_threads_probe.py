import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for n in (16, 32, 64, 128):
    os.environ["PN_CPU_THREADS"] = str(n)
    torch.set_num_threads(n)
    t = time.time()
    r = bench.cpu_baseline()
    print(n, "threads:", r["value"], "pairs/s", time.time() - t, flush=True)
