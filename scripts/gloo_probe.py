"""Does a gloo all_reduce / all_gather of CUDA tensors between two processes that share one GPU complete at the sizes
the bench model exchanges?  (torchrun --nproc-per-node 2 scripts/gloo_probe.py)"""
import os, time, faulthandler
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
for n in (1000, 100_000, 1_000_000, 3_000_000, 10_000_000):
    for rep in range(3):
        x = torch.full((n,), float(rank + 1), device=dev)
        y = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)  # work queued before the collective
        t0 = time.perf_counter()
        dist.all_reduce(x)
        torch.cuda.synchronize()
        ok = bool((x == 3.0).all())
        if rank == 0:
            print(f"all_reduce n={n} rep={rep}: {1e3 * (time.perf_counter() - t0):.1f} ms ok={ok}", flush=True)
parts = [torch.empty(8, 10, 256, device=dev) for _ in range(2)]
dist.all_gather(parts, torch.ones(8, 10, 256, device=dev) * rank)
if rank == 0:
    print("all_gather ok", float(parts[1].mean()), flush=True)
dist.barrier()
dist.destroy_process_group()
