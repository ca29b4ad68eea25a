#!/usr/bin/env python3
"""Which kernels wait for their own global loads one at a time?  Compiles a .hip file to gfx950 assembly and prints, per kernel, the
sequence of global loads (L), stores (S), atomics (A), `s_waitcnt vmcnt` (w), LDS reads / writes (r / W), barriers (|) and branches (b),
run-length encoded -- `L16 w1` is sixteen reads in flight together, `L1 w1 L1 w1 ...` is one dependent trip to memory per element --
sorted by the number of load-immediately-followed-by-wait pairs.
    python3 tools/isa_loads.py orphics_amd/csrc/fft.hip [substring of a mangled kernel name ...]"""
import itertools, re, subprocess, sys, tempfile, os

def main():
    src = sys.argv[1]
    keys = sys.argv[2:]
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read().split("\n")
    start = {l.split(":")[0]: i for i, l in enumerate(s) if l.startswith("_Z") and "; @" in l}
    rows = []
    for name, i0 in start.items():
        if keys and not any(k in name for k in keys):
            continue
        seq = []
        for l in s[i0 + 1:]:
            t = l.strip()
            if t.startswith("s_endpgm"):
                break
            if not t or t.startswith((";", ".")):
                continue
            op = t.split()[0]
            if "load" in op and not op.startswith(("ds_", "s_", "scratch")): seq.append("L")
            elif "store" in op and not op.startswith(("ds_", "scratch")): seq.append("S")
            elif op.startswith("scratch_"): seq.append("x")
            elif "atomic" in op and not op.startswith("ds_"): seq.append("A")
            elif op == "s_waitcnt" and "vmcnt" in t: seq.append("w")
            elif op.startswith(("ds_read", "ds_load")): seq.append("r")
            elif op.startswith(("ds_write", "ds_store")): seq.append("W")
            elif op == "s_barrier": seq.append("|")
            elif op.startswith("s_cbranch"): seq.append("b")
        flat = "".join(c for c in seq if c in "Lw")
        rows.append((len(re.findall("Lw", flat)), flat.count("L"), name, "".join(k + str(len(list(g))) for k, g in itertools.groupby(seq))))
    rows.sort(reverse=True)
    names = subprocess.run(["c++filt"], input="\n".join(r[2] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for (pairs, loads, _, comp), nm in zip(rows, names):
        print("%3d of %3d loads followed by a wait  %s" % (pairs, loads, nm[:150]))
        if keys:
            print("    " + comp)

if __name__ == "__main__":
    main()
