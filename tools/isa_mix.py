#!/usr/bin/env python3
"""Instruction mix of one kernel from a hipcc -S listing (static counts; loops are counted once).
usage: python tools/isa_mix.py file.s 'substring of the mangled kernel name' [--dump out.s]"""
import collections
import re
import sys


def kernel_body(path, needle):
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") and needle in l and l.startswith("_Z") and "@" not in l.split(":")[0]:
            start = i
            break
        if re.match(r"^_Z\S*:\s*; @", l) and needle in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return lines[start], body


def main():
    path, needle = sys.argv[1], sys.argv[2]
    name, body = kernel_body(path, needle)
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
    ops = collections.Counter()
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        ops[s.split()[0]] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        if op.startswith("v_pk_"):
            cls["valu_pk"] += n
        elif op.startswith("v_"):
            cls["valu_other"] += n
        elif op.startswith("ds_"):
            cls["lds"] += n
        elif op.startswith("s_waitcnt"):
            cls["s_waitcnt"] += n
        elif op.startswith("s_barrier"):
            cls["s_barrier"] += n
        elif op.startswith("s_"):
            cls["salu"] += n
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls["vmem"] += n
        else:
            cls["other"] += n
    print(name[:120])
    print(dict(cls))
    for op, n in ops.most_common(40):
        print("  %-28s %d" % (op, n))


if __name__ == "__main__":
    main()
