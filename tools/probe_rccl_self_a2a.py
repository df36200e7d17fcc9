#!/usr/bin/env python3
"""How many bytes does `dist.all_to_all_single` move on an RCCL group of ONE rank?  (Observed on
RCCL 2.26.6 / ROCm 7.0: a 1.16 GB self-exchange returned with only its first 553.75 MiB copied.)"""
import os

import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
MiB = 1 << 20
for dtype in (torch.uint8, torch.bfloat16, torch.float32):
    for mib in (128, 256, 512, 553, 554, 600, 1024, 1107, 1108, 2048, 3000):
        n = mib * MiB // torch.empty((), dtype=dtype).element_size()
        src = torch.randint(1, 100, (n,), device=dev, dtype=torch.int32).to(dtype) if dtype != torch.uint8 else torch.randint(1, 100, (n,), device=dev, dtype=torch.uint8)
        dst = torch.zeros_like(src)
        dist.all_to_all_single(dst, src)
        torch.cuda.synchronize()
        bad = (dst != src).nonzero()
        first = int(bad[0]) if bad.numel() else -1
        print(f"{str(dtype):16s} {mib:5d} MiB: {'ok' if first < 0 else f'WRONG from element {first} = byte {first * src.element_size()} = {first * src.element_size() / MiB:.2f} MiB; {bad.numel()} bad'}", flush=True)
        del src, dst, bad
dist.destroy_process_group()
