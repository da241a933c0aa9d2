#!/usr/bin/env python3
"""Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: copies 1 GiB with 8-byte-per-lane accesses (known byte count)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd import _native as N
n = 1 << 27  # 128 Mi doubles = 1 GiB
a = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
L = N.lib()
for _ in range(3):
    N.check(L.ssg_debug_copy8(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, None), None, "copy8")
torch.cuda.synchronize()
print("copied", n * 8, "bytes x3")
