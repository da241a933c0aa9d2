#!/usr/bin/env python3
"""Measurement aid (GPU box): HBM WRITE-only bandwidth (a fill kernel, 8 bytes per lane, coalesced) next to the copy calibration —
the headline kernel writes 238 B per env-step and reads 6: how far is its write rate from what the memory system takes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
for gb in (1.5, 4.0):
    n = int(gb * 1e9 / 8)
    x = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.25), 8 * n), ("zero_ (write only)", lambda: x.zero_(), 8 * n),
                             ("copy (read + write)", lambda: y.copy_(x), 16 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[3]
        print("%.1f GB buffer, %-20s %7.1f GB/s (%.3f ms)" % (gb, name, nbytes / t / 1e6, t), flush=True)
    del x, y
    torch.cuda.empty_cache()
