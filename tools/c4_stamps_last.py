#!/usr/bin/env python3
"""Development aid (run with SSG_DYN_STOP=-1): phase stamps of the lanes that COMPUTED in one steady-state launch of the full
dyn step (the memo's misses: what the launch waits for)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import numpy as np
from ship_sim_gym_amd.vec_env import ShipVecEnv
from ship_sim_gym_amd import _native as N
n = int(os.environ.get("N", "65536"))
vec = ShipVecEnv(n, n_beams=10, n_maps=64, n_ships=4, dyn_memo=os.environ.get("MEMO", "1") != "0")
W = int(os.environ.get("W", "150"))
acts = vec.random_actions(12345, 0, W + 8)
vec.reset_tensor()
vec.rollout_tensor(acts[:W])
off, es, nc, stride = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
N.check(N.lib().ssg_state_field(vec._h, N.F_TRAFFIC, C.byref(off), C.byref(es), C.byref(nc), C.byref(stride)), vec._h, "f")
npad = stride.value // 8
DC_ARB = 27 + 8 * 6
cols = vec.state[off.value: off.value + (DC_ARB + 4 * 54 + 2) * npad * 8].view(torch.float64).view(-1, npad)
names = ["load+pos", "(unused)", "collide", "prestep", "vel+solver", "writeback"]
for rep in range(6):
    cols[DC_ARB + 200: DC_ARB + 216].zero_()
    vec.step_tensor(acts[W + rep])
    st = cols[DC_ARB + 200: DC_ARB + 216, :n].cpu().numpy()
    hd = cols[DC_ARB + 180: DC_ARB + 183, :n].cpu().numpy()
    sel = st[5] > 0
    if not sel.any():
        print("launch %d: no lane computed" % rep); continue
    i = int(np.argmax(np.where(sel, st[5], 0)))
    s = st[:, i]
    mp = int(vec.field(N.F_MAP_ID)[i]); age = int(vec.field(N.F_STEP_COUNT)[i])
    print("launch %d: %d lanes computed; slowest lane env %d (map %d, age %d): head: counters %.0f entry %.0f row %.0f arbiters %.0f | position + LDS %.0f | broadphase %.0f | memo: hash+probe %.0f verify %.0f mates' hit path %.0f | collide %.0f | ageing %.0f prestep %.0f | solver %.0f | writeback %.0f | total %.0f cycles; n_act %d gjk %d epa %d queries %d"
          % (rep, sel.sum(), i, mp, age, hd[0, i], hd[1, i] - hd[0, i], hd[2, i] - hd[1, i], s[15] - hd[2, i], s[0] - s[15], s[10] - s[0], s[12] - s[10], s[13] - s[12], s[11] - s[13], s[2] - s[11], s[14] - s[2], s[3] - s[14], s[4] - s[3], s[5] - s[4], s[5], s[6], s[7], s[8], s[9]))
print(vec.dyn_memo_stats())
