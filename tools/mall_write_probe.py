"""Does data WRITTEN by a kernel stay in the 256 MB infinity cache?  Effective rates of (a) repeated fills of one buffer, (b) a
read-modify-write sweep, (c) write-then-read pairs, for buffer sizes below and above the cache size."""
import time, torch
dev = "cuda"
def rate(fn, nbytes, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12
for mb in (32, 64, 96, 128, 192, 256, 512, 2048):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    a.fill_(1.0); b.fill_(2.0)
    w = rate(lambda: a.fill_(3.0), mb * 2 ** 20)
    rmw = rate(lambda: a.mul_(1.0000001), 2 * mb * 2 ** 20)
    def wr():
        a.fill_(1.5)            # write a
        torch.add(a, 1.0, out=b)  # read a, write b
    wtr = rate(wr, 3 * mb * 2 ** 20)
    rd = rate(lambda: torch.sum(a), mb * 2 ** 20)
    print("%5d MB: fill %.2f TB/s, in-place multiply %.2f TB/s (r+w), fill + add-to-other %.2f TB/s (w+r+w), sum %.2f TB/s" % (mb, w, rmw, wtr, rd), flush=True)
