import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29512")
torch.cuda.set_device(0)
tables = [torch.zeros(n, dtype=torch.int64, device="cuda") for n in (750, 1800, 500, 1200, 101, 94, 900, 900)]
flat = torch.empty(sum(t.numel() for t in tables), dtype=torch.int64, device="cuda")
parts = list(flat.split([t.numel() for t in tables]))
for _ in range(5):
    torch._foreach_copy_(parts, tables); dist.all_reduce(flat); torch.cuda.synchronize()
for name, fn in (("copy", lambda: torch._foreach_copy_(parts, tables)), ("allreduce", lambda: dist.all_reduce(flat)),
                 ("both", lambda: (torch._foreach_copy_(parts, tables), dist.all_reduce(flat)))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn(); torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
