#!/usr/bin/env python3
"""Diagnostic (needs a -DSSG_STAMPS -DSSG_STAMPS_ITER build): how the 1024 tiles' finishing times of a K-step launch are
distributed (the launch ends with its slowest tile)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n, nb, epw = 65536, 8, 256
K = int(os.environ.get("K", "20"))
vec = ShipVecEnv(n, n_maps=64, n_beams=nb)
L = N.lib()
buf = torch.zeros((4 * n // 64, 16), dtype=torch.int64, device="cuda")
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
acts = vec.random_actions(12345, 0, 205 + K * 4)
vec.reset_tensor(); vec.rollout_tensor(acts[:205])
ends = []
for r in range(4):
    torch.cuda.synchronize()
    vec.rollout_tensor(acts[205 + K * r: 205 + K * (r + 1)]); torch.cuda.synchronize()
    b = buf.cpu().numpy().astype(np.int64).reshape(-1, 4, epw // 64, 16)   # [wg][role][tile][slot]
    start = b[..., 8].min()                        # first wave of the launch (all CUs share one clock domain? see spread)
    wg_start = b[..., 8].min(axis=(1, 2)) - start
    end3 = b[:, 3, :, 10] - b[..., 8].min(axis=(1, 2))[:, None]          # role 3 end, relative to its workgroup's start
    endo = b[:, 2, :, 10] - b[..., 8].min(axis=(1, 2))[:, None]          # observer end
    per_step = (b[:, 3, :, 11] - b[:, 3, :, 0]) / (K - 1)                # role 3: mean cycles per step of this tile
    ends.append(endo)
    print("launch %d: wg start spread %d..%d | role-3 end mean %.0f sd %.0f min %.0f max %.0f | observer end mean %.0f max %.0f | cycles/step per tile: mean %.0f sd %.0f min %.0f max %.0f" % (
        r, wg_start.min(), wg_start.max(), end3.mean(), end3.std(), end3.min(), end3.max(), endo.mean(), endo.max(),
        per_step.mean(), per_step.std(), per_step.min(), per_step.max()))
    xcd = np.arange(end3.shape[0]) % 8
    print("    per XCD (blockIdx %% 8) mean role-3 end:", " ".join("%.0f" % end3[xcd == x].mean() for x in range(8)))
e = np.stack(ends)  # [launch][wg][tile]
print("correlation of a tile's finishing time between consecutive launches: %.2f" % np.corrcoef(e[1].ravel(), e[2].ravel())[0, 1])
