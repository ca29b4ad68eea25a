#!/usr/bin/env python3
"""List VGPRs / spills / scratch / occupancy of every kernel in a HIP source (hipcc -Rpass-analysis).
usage: python tools/kernel_resources.py orphics_amd/csrc/fft.hip [--scratch-only]"""
import re
import subprocess
import sys

src = sys.argv[1]
only = "--scratch-only" in sys.argv
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src,
                      "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
name, rec = None, {}
keys = {"VGPRs": r"VGPRs: (\d+)", "spill": r"VGPRs Spill: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        rec[name] = {}
    for k, pat in keys.items():
        m = re.search(pat, line)
        if m and name:
            rec[name][k] = int(m.group(1))
for n, r in rec.items():
    if only and not r.get("scratch"):
        continue
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    print("%-100s %s" % (d[:100], r))
