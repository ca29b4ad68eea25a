"""4096^2 windowed Gaussian Monte Carlo (oa_mc_run_windowed) in a loop, for rocprofv3 --kernel-trace --stats."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from orphics_amd import mc, maps                     # noqa: E402
sys.path.insert(0, 'tools')
import config_bench as cb                            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape, g, th, ml, beam, noise, q = cb.setup(N, 0.5, False)
tot = (th.lCl("TT", ml) * beam ** 2 + noise)[:, :N // 2 + 1]
taper, w2 = maps.get_taper(shape, g)
drv = mc.GaussianN0MonteCarlo(q, tot, np.linspace(20, 3500, 20), mean_field=True, window=taper)
drv.run_local(range(8))
torch.cuda.synchronize()
t0 = time.perf_counter()
drv.run_local(range(8, 72))
torch.cuda.synchronize()
print("windowed %d^2: %.1f us per realisation" % (N, (time.perf_counter() - t0) / 64 * 1e6))
