#!/usr/bin/env python3
"""What read bandwidth does a plain streaming kernel reach on this box?  (ceiling for the row R2C pass, whose 268 MB
input map is read exactly once per reconstruction.)  Alternates between buffers so that nothing is served by the
256 MB infinity cache; also shows the same loop over ONE buffer (cache-assisted) for contrast."""
import torch

N = 8192
bufs = [torch.randn(N, N, device="cuda") for _ in range(3)]
out = torch.empty(N, N, device="cuda")


def timed(fn, reps=30):
    for i in range(5):
        fn(i)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps):
        fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


nb = 4 * N * N
for name, fn, moved in [("sum, 3 buffers in turn", lambda i: bufs[i % 3].sum(), nb), ("sum, one buffer", lambda i: bufs[0].sum(), nb),
                        ("abs-max, 3 buffers", lambda i: bufs[i % 3].abs().max(), None),
                        ("copy, 3 buffers in turn (read + write)", lambda i: out.copy_(bufs[i % 3]), 2 * nb)]:
    dt = timed(fn)
    if moved:
        print("%-44s %.1f us  %.2f TB/s" % (name, dt * 1e6, moved / dt / 1e12), flush=True)
