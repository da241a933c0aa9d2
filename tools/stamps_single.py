#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS build, see tools/build_variant.sh): s_memtime timeline of single-step launches
(ssg_step, K = 1) of the step kernel, in cycles after the workgroup's earliest wave start; mean over waves / launches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = int(os.environ.get("SSG_N", "65536")); nb = int(os.environ.get("SSG_NB", "8"))
mode = os.environ.get("SSG_MODE", "bank")  # "fresh_device": per-env rings of worlds (RING records each), the bank gathered from L2 / HBM
vec = ShipVecEnv(n, n_maps=64, n_beams=nb, n_ships=int(os.environ.get("SSG_SHIPS", "1")), map_mode=mode, ring=int(os.environ.get("RING", "32")))
L = N.lib()
epw = 64 if n <= 64 * 256 else (128 if n <= 128 * 256 else 256)
nw = 4 * ((n + epw - 1) // epw) * epw // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 400)
vec.reset_tensor()
for k in range(200): vec.step_tensor(acts[k])
wpr = epw // 64
R = 60
acc = np.zeros((4, 12)); span = 0.0
for r in range(R):
    vec.step_tensor(acts[200 + r]); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, 4, wpr, 16)[..., :12]
    t0 = b[..., 8].min(axis=(1, 2))[:, None, None, None]
    acc += (b - t0).mean(axis=(0, 2))
    span += (b[..., 10].max(axis=(1, 2)) - b[..., 8].min(axis=(1, 2))).mean()
acc /= R
print("workgroup span (first wave start -> last wave end): %.0f cycles" % (span / R))
sp = (b[..., 10].max(axis=(1, 2)) - b[..., 8].min(axis=(1, 2)))
st = b[..., 8].min(axis=(1, 2)); en = b[..., 10].max(axis=(1, 2))
print("last launch: span per workgroup min %.0f median %.0f p99 %.0f max %.0f | first start -> last end over the whole grid: %.0f cycles | start skew (max - min of the workgroups' first-wave starts) %.0f" % (
    sp.min(), np.median(sp), np.percentile(sp, 99), sp.max(), en.max() - st.min(), st.max() - st.min()))
for role, nm in ((0, "lidar lo"), (1, "lidar hi"), (2, "observer"), (3, "body")):
    a = acc[role]
    print("role %d %-9s start %6.0f | after barrier 0 %6.0f | stamps %s | end %6.0f" % (
        role, nm, a[8], a[9], " ".join("%6.0f" % v for v in a[:8]), a[10]))
