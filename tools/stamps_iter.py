#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS -DSSG_STAMPS_ITER build): when role 3 starts each of the first steps of a K-step
launch (cycles after the workgroup's first wave started), mean over tiles / workgroups."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n, nb, epw = 65536, 8, 256
K = int(os.environ.get("K", "20"))
vec = ShipVecEnv(n, n_maps=64, n_beams=nb)
L = N.lib()
nw = 4 * n // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 205 + K * 6)
vec.reset_tensor()
vec.rollout_tensor(acts[:205])
for r in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    vec.rollout_tensor(acts[205 + K * r: 205 + K * (r + 1)]); e1.record(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, 4, epw // 64, 16)
    t0 = b[..., 8].min(axis=(1, 2))[:, None]            # first wave of the workgroup
    r3 = b[:, 3, :, :]                                   # role 3 stamps [wg][tile][slot]
    m = (r3 - t0[:, :, None]).mean(axis=(0, 1))
    wgspan = (b[..., 10].max(axis=(1, 2)) - b[..., 8].min(axis=(1, 2)))
    print("launch %d: %.1f us by events | barrier0 %6.0f | step starts %s | k=12 %6.0f k=16 %6.0f | last step %6.0f | role 3 end %6.0f | wg span mean %6.0f max %6.0f min-start-skew %6.0f" % (
        r, e0.elapsed_time(e1) * 1e3, m[9], " ".join("%6.0f" % v for v in m[:8]), m[12], m[13], m[11], m[10], wgspan.mean(), wgspan.max(),
        (b[..., 8].min(axis=(1, 2)).max() - b[..., 8].min())))
