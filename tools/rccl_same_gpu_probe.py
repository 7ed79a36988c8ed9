"""Probe: can two RCCL ranks share ONE MI355X (this pool leases single GPUs)?  Prints what happens."""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok -> {t[0].item()} backend {dist.get_backend()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)

if __name__ == "__main__":
    mp.spawn(worker, args=(2,), nprocs=2, join=True)
