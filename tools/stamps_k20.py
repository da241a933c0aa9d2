#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS build): the driver's bench sequence (reset, 5 warm-up steps, five 20-step launches)
with the stamps of each launch's second-to-last step: which role is late in the post-reset transient, where all envs
are in the same phase of their first episodes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n, nb, epw = 65536, 8, 256
vec = ShipVecEnv(n, n_maps=64, n_beams=nb)
L = N.lib()
nw = 4 * n // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 5 + 20 * 12)
vec.reset_tensor()
vec.rollout_tensor(acts[:5])
wpr = epw // 64
for r in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    vec.rollout_tensor(acts[5 + 20 * r: 25 + 20 * r]); e1.record(); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, 4, wpr, 16)
    t0 = b[:, 3, :, 0][:, None, :, None]
    a = (b[..., :8] - t0).mean(axis=(0, 2))
    done = float(vec.done.double().mean()) if hasattr(vec, "done") else -1
    print("launch %2d (steps %3d-%3d) %.1f us | role3: pose %5.0f goals %5.0f B %5.0f end %5.0f | lidar lo: pose-copied %5.0f B %5.0f query-end %5.0f (%5.0f) | hi: %5.0f (%5.0f) | obs: B %5.0f rows-end %5.0f (%5.0f)" % (
        r, 5 + 20 * r, 25 + 20 * r, e0.elapsed_time(e1) * 1e3, a[3, 2], a[3, 3], a[3, 4], a[3, 5],
        a[0, 0], a[0, 1], a[0, 2], a[0, 2] - a[0, 1], a[1, 2], a[1, 2] - a[1, 1], a[2, 2], a[2, 3], a[2, 3] - a[2, 2]))
