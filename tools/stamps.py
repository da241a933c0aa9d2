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
epw = 64 if n <= 64 * 256 else (128 if n <= 128 * 256 else 256)
nw = 4 * ((n + epw - 1) // epw) * epw // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 300)
vec.reset_tensor()
vec.rollout_tensor(acts[:200])
names = {
    0: ["loads+sincos+force", "wait DMA+barrier1", "bb+cull+queue", "lidar passes", "wait barrier2"],
    1: ["loads+sincos", "wait DMA+barrier1", "bb+cull+queue", "lidar passes", "wait barrier2"],
    2: ["loads+integrate+sincos", "wait DMA+barrier1", "bb+SAT+publish", "-", "wait barrier2"],
    3: ["loads+integrate+sincos", "wait DMA+barrier1", "bb+goals+nearest", "prev nearest goal", "wait barrier2", "tail"],
}
wpr = epw // 64
acc = {r: np.zeros(len(v)) for r, v in names.items()}
fused = int(os.environ.get("SSG_STAMP_FUSED", "1"))
for k in range(200, 300):
    if fused:
        vec.rollout_tensor(acts[k - 50:k])   # 50 fused steps; the stamps left behind are the last iteration's
    else:
        vec.step_tensor(acts[k])
    torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, 4, wpr, 16)
    for r in range(4):
        m = len(names[r])
        acc[r] += np.diff(b[:, r, :, :m + 1], axis=-1).mean(axis=(0, 1))
b3 = b[:, 3, :, :]
print("role 3 goals fine: 2->10 bb+consts %.0f | 10->11 near tests+queue %.0f | 11->12 pair passes %.0f | 12->3 gw read+nearest goal %.0f" % (
    (b3[..., 10] - b3[..., 2]).mean(), (b3[..., 11] - b3[..., 10]).mean(), (b3[..., 12] - b3[..., 11]).mean(), (b3[..., 3] - b3[..., 12]).mean()))
print("role 3 post: 6->13 lidar results + obs tile + obs stores %.0f | 13->14 outputs + state stores issued %.0f | 14->15 store drain (vmcnt 0) %.0f" % (
    (b3[..., 13] - b3[..., 6]).mean(), (b3[..., 14] - b3[..., 13]).mean(), (b3[..., 15] - b3[..., 14]).mean()))
for r in range(4):
    print("role %d" % r)
    for nme, v in zip(names[r], acc[r] / 100):
        print("   %-26s %8.0f cycles" % (nme, v))
    print("   total %.0f" % (acc[r].sum() / 100))
