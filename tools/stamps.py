#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS build): average per-wave cycles between the s_memtime stamps of the step kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = int(os.environ.get("SSG_N", "65536"))
vec = ShipVecEnv(n, n_maps=64, n_beams=8)
L = N.lib()
epw = 64 if n <= 64*256 else (128 if n <= 128*256 else 256)
nw = 2 * n // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 300)
vec.reset_tensor()
vec.rollout_tensor(acts[:200])
namesA = ["A loads+sincos+action+bb", "A wait DMA+barrier1", "A prevgoal+cull+queue", "A lidar passes", "A wait barrier2"]
namesB = ["B loads+integrate+sincos+bb", "B wait DMA+barrier1", "B SAT", "B goals", "B nearest+reward+stats", "B wait barrier2", "B merge+obs assembly"]
accA = np.zeros(5); accB = np.zeros(7)
wpg = 2 * epw // 64
for k in range(200, 300):
    vec.step_tensor(acts[k]); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, wpg, 16)
    a_, b_ = b[:, :wpg // 2, :], b[:, wpg // 2:, :]
    accA += np.diff(a_[..., :6], axis=-1).mean(axis=(0, 1))
    accB += np.diff(b_[..., :8], axis=-1).mean(axis=(0, 1))
for nme, v in zip(namesA, accA / 100):
    print("%-30s %8.0f cycles" % (nme, v))
print("A total %.0f" % (accA.sum() / 100))
for nme, v in zip(namesB, accB / 100):
    print("%-30s %8.0f cycles" % (nme, v))
print("B total %.0f" % (accB.sum() / 100))
