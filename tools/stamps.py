#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS build): average per-wave cycles between the s_memtime stamps of the step kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = 65536
vec = ShipVecEnv(n, n_maps=64, n_beams=8)
L = N.lib()
buf = torch.zeros((n // 64, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 300)
vec.reset_tensor()
vec.rollout_tensor(acts[:200])
names = ["loads+prebank compute", "wait DMA + barrier", "prev goal+cull+queue", "lidar passes", "lidar gather", "SAT", "goals", "tail compute"]
acc = np.zeros(8)
span = 0.0
for k in range(200, 300):
    vec.step_tensor(acts[k]); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64)
    d = np.diff(b[:, :9], axis=1)
    acc += d.mean(axis=0)
    span += (b[:, 8].max() - b[:, 0].min())
acc /= 100
for nme, v in zip(names, acc):
    print("%-24s %8.0f cycles" % (nme, v))
print("total stamped (per wave) %8.0f cycles;  first stamp0 -> last stamp8 over the chip: %.0f" % (acc.sum(), span / 100))
