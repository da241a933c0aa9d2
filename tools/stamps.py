#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS build: `make -C ship_sim_gym_amd/csrc EXTRA=-DSSG_STAMPS`): per-wave s_memtime
stamps of the second-to-last step of fused launches of the pipelined step kernel, averaged over waves and launches.
All stamps of a workgroup come from one CU's clock, so differences ACROSS roles are meaningful too."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = int(os.environ.get("SSG_N", "65536"))
nb = int(os.environ.get("SSG_NB", "8"))
mode = os.environ.get("SSG_MODE", "bank")  # or fresh_device: per-env records gathered from L2
vec = ShipVecEnv(n, n_maps=64, n_beams=nb, map_mode=mode, ring=64) if mode != "bank" else ShipVecEnv(n, n_maps=64, n_beams=nb)
L = N.lib()
epw = 64 if n <= 64 * 256 else (128 if n <= 128 * 256 else 256)
NR = 6 if epw <= 128 else 4  # waves per tile (csrc: tile_roles)
nw = NR * ((n + epw - 1) // epw) * epw // 64
buf = torch.zeros((nw, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 400)
vec.reset_tensor()
vec.rollout_tensor(acts[:200])
wpr = epw // 64
R = 40
acc = np.zeros((NR, 8))
for r in range(R):
    vec.rollout_tensor(acts[200 + r: 250 + r])  # 50 fused steps; stamps are those of step 48
    torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, NR, wpr, 16)  # [workgroup][role][tile][stamp]
    t0 = b[:, 3, :, 0][:, None, :, None]                             # role 3's loop-top stamp of the same tile
    acc += (b[..., :8] - t0).mean(axis=(0, 2))
acc /= R
names3 = ["iter start", "pose ready (acks of pose k-1 in)", "pose published", "goals done, done word written (arrives at B)",
          "B complete", "statistics / reset done (iteration end)"]
print("all times in cycles after role 3's iteration start (mean over tiles / workgroups / %d launches)" % R)
print("role 3 (body):")
for i, nme in enumerate(names3):
    print("   %-46s %8.0f   (+%.0f)" % (nme, acc[3, i], acc[3, i] - (acc[3, i - 1] if i else 0)))
for role, nm in ((0, "lidar lo"), (1, "lidar hi")):
    print("role %d (%s): pose copied %.0f | arrives at B %.0f | B complete %.0f | next step's query done %.0f  (query = %.0f cycles)" % (
        role, nm, acc[role, 0], acc[role, 3], acc[role, 1], acc[role, 2], acc[role, 2] - acc[role, 1]))
print("role 2 (observer): pose copied %.0f | first-step narrowphase slot %.0f (= %.0f cycles) | B complete %.0f | outputs + obs rows written %.0f (= %.0f cycles)" % (
    acc[2, 0], acc[2, 1], acc[2, 1] - acc[2, 0], acc[2, 2], acc[2, 3], acc[2, 3] - acc[2, 2]))
if NR == 6:
    for role in (4, 5):
        print("role %d (collide_ship, bank hull %d): pose seen %.0f | arrives at B %.0f (= %.0f cycles) | B complete %.0f" % (
            role, role - 4, acc[role, 0], acc[role, 1], acc[role, 1] - acc[role, 0], acc[role, 2]))
