#!/usr/bin/env python3
"""Does giving the HBM-bound row R2C pass and the latency-bound small launches DISJOINT sets of CUs raise throughput?
Each lane (forked estimator handle) runs stage 0 (row R2C) on an 'R' stream and stages 1..5 on an 'S' stream, chained by
events; the streams are created with hipExtStreamCreateWithCUMask.  usage: python tools/cumask_probe.py [lanes] [steps]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench

hip = ctypes.CDLL("libamdhip64.so")
P_ = ctypes.c_void_p


def hipcheck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


def masked_stream(words):
    s = P_()
    arr = (ctypes.c_uint32 * len(words))(*words)
    hipcheck(hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr), "hipExtStreamCreateWithCUMask")
    return s


def plain_stream():
    s = P_()
    hipcheck(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), "hipStreamCreateWithFlags")   # non-blocking
    return s


def event():
    e = P_()
    hipcheck(hip.hipEventCreateWithFlags(ctypes.byref(e), 2), "hipEventCreateWithFlags")      # disable timing
    return e


def run(lanes, steps, rmask, smask, label):
    from orphics_amd._lib import check
    from orphics_amd.engine import _ptr
    N, res = 8192, 0.5
    Pp = bench.build_pipeline(N, res, "f32", torch)
    tm = bench.make_maps(Pp, torch, 1234)
    q = Pp["q"]
    qs = [q] + [q.fork() for _ in range(lanes - 1)]
    norm = Pp["geom"].area / float(N * N) ** 2
    for e in qs:
        e.bind_bins(Pp["ids"], Pp["nids"], norm)
    plans = []
    for e in qs:
        en = e._bind_bins()
        plans.append((en.lib, en.plan))
    torch.cuda.synchronize()
    R = [masked_stream(rmask) if rmask else plain_stream() for _ in range(lanes)]
    S = [masked_stream(smask) if smask else plain_stream() for _ in range(lanes)]
    evR = [event() for _ in range(lanes)]
    evS = [event() for _ in range(lanes)]
    maps = [_ptr(t) for t in tm]

    def lane_loop(j, count):
        # one host thread per lane (ctypes releases the GIL inside the calls): the probe must not be issue-bound
        lib, plan = plans[j]
        for it in range(count):
            m = maps[(it + j) & 1]
            if it:
                hipcheck(hip.hipStreamWaitEvent(R[j], evS[j], 0), "wait")     # the lane's previous column pass 1 has read tA
            check(lib.oa_qe_tt_stage(plan, 0, m, R[j]))
            hipcheck(hip.hipEventRecord(evR[j], R[j]), "record")
            hipcheck(hip.hipStreamWaitEvent(S[j], evR[j], 0), "wait")
            check(lib.oa_qe_tt_stage(plan, 1, m, S[j]))
            hipcheck(hip.hipEventRecord(evS[j], S[j]), "record")
            for k in (2, 3, 4, 5):
                check(lib.oa_qe_tt_stage(plan, k, m, S[j]))

    import threading

    def all_lanes(count):
        th = [threading.Thread(target=lane_loop, args=(j, count)) for j in range(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
    all_lanes(5)
    t0 = time.perf_counter()
    all_lanes(steps // lanes)
    dt = time.perf_counter() - t0
    steps = (steps // lanes) * lanes
    print("%-34s lanes %d: %.0f recon/s (%.1f us per reconstruction)" % (label, lanes, steps / dt, dt / steps * 1e6), flush=True)
    del qs, Pp, tm


if __name__ == "__main__":
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    full = [0xFFFFFFFF] * 8
    cfgs = {"plain": (None, None, "plain streams (no masks)"),
            "192": ([0x00FFFFFF] * 8, [0xFF000000] * 8, "R: 24 of every 32 CUs, S: 8"),
            "224": ([0x0FFFFFFF] * 8, [0xF0000000] * 8, "R: 28 of every 32 CUs, S: 4"),
            "160": ([0x000FFFFF] * 8, [0xFFF00000] * 8, "R: 20 of every 32 CUs, S: 12"),
            "xcd": ([0xFFFFFFFF] * 6 + [0, 0], [0, 0, 0, 0, 0, 0, 0xFFFFFFFF, 0xFFFFFFFF], "R: words 0-5, S: words 6-7"),
            "Rall": (full, [0xFF000000] * 8, "R: all CUs, S: 8 of every 32")}
    for key, (rm, sm, label) in cfgs.items():
        if which in ("all", key):
            run(lanes, steps, rm, sm, label)
