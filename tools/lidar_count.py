#!/usr/bin/env python3
"""Diagnostic (needs tools/build_variant.sh lcount -DSSG_STAMPS -DSSG_LIDAR_COUNT): of the (beam, hull) pairs that survive the
box-against-box cull of the lidar, how many are hits?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv
n, nb = 65536, int(os.environ.get("SSG_NB", "8"))
vec = ShipVecEnv(n, n_maps=64, n_beams=nb)
L = N.lib()
buf = torch.zeros(65536 + 16, dtype=torch.int64, device="cuda")  # (the stamp area of a -DSSG_STAMPS build, then the counters)
acts = vec.random_actions(12345, 0, 600)
vec.reset_tensor(); vec.rollout_tensor(acts[:300]); torch.cuda.synchronize()
L.ssg_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
L.ssg_debug_set_stamp_buffer(vec._h, C.c_void_p(buf.data_ptr()))
vec.rollout_tensor(acts[300:600]); torch.cuda.synchronize()
pairs, hits, passes, inside = [int(v) for v in buf[65536:65540].cpu().tolist()]
steps = 300
print("per env-step: %.2f surviving pairs of %d (%.1f %%), %.2f hits (%.1f %% of the survivors), origin inside a hull %.4f; passes per tile-step %.2f (lane use %.1f %%)" % (
    pairs / (n * steps), 2 * nb, 100.0 * pairs / (n * steps * 2 * nb), hits / (n * steps), 100.0 * hits / max(pairs, 1), inside / (n * steps),
    passes / (n / 64 * steps), 100.0 * pairs / max(passes * 64, 1)))
