#!/usr/bin/env python3
"""Development aid: per-phase s_memtime stamps of the config-4 full dyn step for UNIFORM waves — every env on the same
map, all reset together, so that all 64 lanes of a wave walk the same path (no divergence): the length of ONE env's
chain, step by step after a reset.  Run with SSG_DYN_STOP=-1 (and a -DSSG_DYN_PROFILE variant for the categories)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import numpy as np
from ship_sim_gym_amd.vec_env import ShipVecEnv
from ship_sim_gym_amd import _native as N
n = int(os.environ.get("N", "4096"))
vec = ShipVecEnv(n, n_beams=10, n_maps=1, n_ships=4)
acts = torch.zeros((12, n), dtype=torch.int32, device=vec.device) + 1  # rudder only: the player stays at the spawn point
vec.reset_tensor()
off, es, nc, stride = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
N.check(N.lib().ssg_state_field(vec._h, N.F_TRAFFIC, C.byref(off), C.byref(es), C.byref(nc), C.byref(stride)), vec._h, "f")
npad = stride.value // 8
DC_ARB = 27 + 8 * 6
cols = vec.state[off.value: off.value + (DC_ARB + 4 * 54 + 2) * npad * 8].view(torch.float64).view(-1, npad)
names = ["load+pos", "player hit", "collide", "prestep", "vel+solver", "writeback"]
names_p = ["loops/rejects/shapes", "gjk", "epa", "closest+edges+clip", "push", "bank staging"]
for k in range(12):
    cols[DC_ARB + 200: DC_ARB + 210].zero_()
    vec.step_tensor(acts[k])
    torch.cuda.synchronize()
    st = cols[DC_ARB + 200: DC_ARB + 210, :n].cpu().numpy()
    fl = vec.field(N.F_DYN_FLAGS).cpu().numpy()
    sel = st[5] > 0
    if not sel.any():
        print("step %2d: nothing queued (rest %.2f)" % (k, ((fl & 4) != 0).mean()))
        continue
    d = np.diff(np.vstack([np.zeros(sel.sum()), st[:6][:, sel]]), axis=0)
    line = " ".join("%s %6.0f" % (nm, np.median(d[i])) for i, nm in enumerate(names))
    print("step %2d: queued %5d | %s | total %7.0f | n_act %s gjk %.1f epa %.1f queries %.1f" % (
        k, sel.sum(), line, np.median(st[5][sel]), np.bincount(st[6][sel].astype(int)).tolist(), st[7][sel].mean(), st[8][sel].mean(), st[9][sel].mean()))
    pr = cols[DC_ARB + 180: DC_ARB + 186, :n].cpu().numpy()
    if pr[:, sel].max() > 0:
        print("         collide by category: " + " ".join("%s %6.0f" % (nm, np.median(pr[i][sel])) for i, nm in enumerate(names_p)))
