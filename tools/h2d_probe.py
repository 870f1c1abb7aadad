"""Does a small host->device copy block the host behind queued GPU work?  (pageable vs pinned, ROCm)"""
import time

import numpy as np
import torch

x = torch.randn(8192, 8192, device="cuda")


def busy():
    for _ in range(30):
        x @ x


def probe(name, fn):
    torch.cuda.synchronize()
    busy()
    t = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-44s host %.2f ms (queued work then took %.1f ms more)" % (name, dt * 1e3, (time.perf_counter() - t2) * 1e3), flush=True)
    return out


busy(); torch.cuda.synchronize()
pinned = torch.empty(1024, dtype=torch.int64).pin_memory()
arr = np.arange(1024)
probe("nothing", lambda: None)
probe("torch.tensor(list, device=cuda)", lambda: torch.tensor([1.0, 2.0, 3.0, 4.0], device="cuda"))
probe("as_tensor(np).to(cuda, non_blocking=True)", lambda: torch.as_tensor(arr).to("cuda", non_blocking=True))
probe("pinned.to(cuda, non_blocking=True)", lambda: pinned.to("cuda", non_blocking=True))
probe("torch.full((4,), 1.0, device=cuda)", lambda: torch.full((4,), 1.0, device="cuda"))
probe("torch.zeros(()).requires_grad", lambda: torch.zeros((), device="cuda", requires_grad=True))
probe("torch.arange(device=cuda)", lambda: torch.arange(100, device="cuda"))
