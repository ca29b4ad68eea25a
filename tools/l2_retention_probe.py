#!/usr/bin/env python3
"""Do the per-XCD L2s keep data across kernel boundaries?  The same elementwise kernel (same grid -> same workgroup-to-XCD
mapping) is run repeatedly in place on buffers of several sizes; run under `rocprofv3 --pmc FETCH_SIZE` and look at the
bytes each launch fetches: ~0 for a buffer that fits the L2s would mean a consumer launched with the producer's block
mapping finds its data in its own XCD's L2."""
import torch
for mb in (2, 8, 24, 64, 512):
    x = torch.ones(mb * 1024 * 1024 // 4, device="cuda")
    for _ in range(6):
        x.mul_(1.0001)
    torch.cuda.synchronize()
    print("buffer %d MB done" % mb, flush=True)
