#!/usr/bin/env python3
"""Development aid: per-phase s_memtime stamps of the config-4 full dyn step (run with SSG_DYN_STOP=-1)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import numpy as np
from ship_sim_gym_amd.vec_env import ShipVecEnv
from ship_sim_gym_amd import _native as N
n = int(os.environ.get("N", "4096"))
vec = ShipVecEnv(n, n_beams=10, n_maps=int(os.environ.get("MAPS", "64")), n_ships=4, map_mode=os.environ.get("MAP_MODE", "bank"),
                 ring=int(os.environ.get("RING", "8")), dyn_memo=os.environ.get("MEMO", "1") != "0")
acts = vec.random_actions(12345, 0, 120)
vec.reset_tensor()
vec.rollout_tensor(acts)
off, es, nc, stride = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
N.check(N.lib().ssg_state_field(vec._h, N.F_TRAFFIC, C.byref(off), C.byref(es), C.byref(nc), C.byref(stride)), vec._h, "f")
npad = stride.value // 8
DC_ARB = 27 + 8 * 6
cols = vec.state[off.value: off.value + (DC_ARB + 4 * 54 + 2) * npad * 8].view(torch.float64).view(-1, npad)
st = cols[DC_ARB + 200: DC_ARB + 210, :n].cpu().numpy()
fl = vec.field(N.F_DYN_FLAGS).cpu().numpy()
sel = st[5] > 0
print("envs with stamps", sel.sum(), "of", n)
names = ["load+pos", "player hit", "collide", "prestep", "vel+solver", "writeback"]
prev = np.zeros(sel.sum())
for i, nm in enumerate(names):
    cur = st[i][sel]
    print("%-12s median %8.0f  p90 %8.0f  max %8.0f cycles" % (nm, np.median(cur - prev), np.percentile(cur - prev, 90), (cur - prev).max()))
    prev = cur
print("total median %8.0f max %8.0f ; n_act hist" % (np.median(st[5][sel]), st[5][sel].max()), np.bincount(st[6][sel].astype(int)))
print("per env: gjk iterations mean %.2f max %d | epa iterations mean %.2f max %d | narrowphase queries mean %.2f max %d" % (
    st[7][sel].mean(), st[7][sel].max(), st[8][sel].mean(), st[8][sel].max(), st[9][sel].mean(), st[9][sel].max()))
# (per-wave figures: see the wave table below — waves are 48 envs of the SORTED queue, not 64 neighbours of the batch)

pr = cols[DC_ARB + 180: DC_ARB + 186, :n].cpu().numpy()
if pr[1][sel].max() > 0:
    names_p = ["loops / rejects / shapes", "gjk", "epa", "closest + edges + clip", "push", "broadphase (54 box tests)"]
    print("collide phase by category (SSG_DYN_PROFILE build), median cycles per wave:")
    heavy_ = st[5][sel] >= np.percentile(st[5][sel], 97)
    for i, nm in enumerate(names_p):
        print("   %-26s %8.0f   slowest 3 %% of the waves: %8.0f" % (nm, np.median(pr[i][sel]), np.median(pr[i][sel][heavy_])))
    for i, nm in enumerate(names):
        cur = st[i][sel][heavy_]; prv = st[i - 1][sel][heavy_] if i else 0
        print("   phase %-12s in the slowest 3 %%: median %8.0f" % (nm, np.median(cur - prv)))
# do the iteration counts depend on the episode's age (would age-binned waves be more homogeneous)?
age = vec.field(N.F_STEP_COUNT).cpu().numpy()
tot = st[7] + 2.5 * st[8]  # rough cost: an EPA iteration ~2.5 GJK iterations
for lo, hi in ((0, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 2000)):
    m = sel & (age >= lo) & (age < hi)
    if m.sum():
        print("age [%3d,%4d): %6d envs | gjk mean %.2f p90 %.0f max %d | epa mean %.2f p90 %.0f max %d | queries mean %.2f" % (
            lo, hi, m.sum(), st[7][m].mean(), np.percentile(st[7][m], 90), st[7][m].max(), st[8][m].mean(),
            np.percentile(st[8][m], 90), st[8][m].max(), st[9][m].mean()))
q = fl & 4
print("resting %.3f of envs; resting by age:" % (q != 0).mean(), [round(float(((q != 0) & (age >= lo) & (age < hi)).sum() / max(1, ((age >= lo) & (age < hi)).sum())), 2) for lo, hi in ((0, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 2000))])

# per wave (envs that share the same final stamp were lanes of one wave): what makes the slow waves slow?
tot = st[5]
keys, inv = np.unique(tot[sel], return_inverse=True)
rows = []
for w in range(len(keys)):
    m = inv == w
    rows.append((keys[w], m.sum(), st[6][sel][m].max(), st[9][sel][m].max(), st[7][sel][m].max(), st[8][sel][m].max(),
                 len(np.unique(st[7][sel][m] * 100 + st[8][sel][m] * 10 + st[9][sel][m])),
                 st[0][sel][m][0], (st[2] - st[1])[sel][m][0], (st[4] - st[3])[sel][m][0]))
rows.sort()
print("waves: %d; total cycles p10 %.0f median %.0f p90 %.0f max %.0f" % (len(rows), rows[len(rows) // 10][0], rows[len(rows) // 2][0], rows[len(rows) * 9 // 10][0], rows[-1][0]))
print("slowest / fastest waves: total, lanes, max n_act, max queries, max gjk it, max epa it, distinct (gjk,epa,q) signatures, load, collide, solver")
for r in rows[-8:] + rows[:4]:
    print("   ", " ".join("%7.0f" % v for v in r))

ty = cols[DC_ARB + 188: DC_ARB + 193, :n].cpu().numpy()
if ty[:, sel].max() > 0:
    nm = ["goal-bank", "goal-goal", "ship-bank", "goal-ship", "ship-ship"]
    print("narrowphase trips by pair type (cycles per wave): median / p90 / max and share of waves that enter the type")
    for i in range(5):
        v = ty[i][sel]
        print("   %-10s %8.0f %8.0f %8.0f   entered by %.2f of the waves" % (nm[i], np.median(v), np.percentile(v, 90), v.max(), (v > 0).mean()))
    tc = cols[DC_ARB + 210: DC_ARB + 215, :n].cpu().numpy()
    trips = tc[:, sel].sum(axis=0)
    types = (tc[:, sel] > 0).sum(axis=0)
    print("   narrowphase trips per wave: mean %.2f p90 %.0f max %.0f | pair types per wave: mean %.2f | trips by type (mean): %s" % (
        trips.mean(), np.percentile(trips, 90), trips.max(), types.mean(), ", ".join("%s %.2f" % (nm[i], tc[i][sel].mean()) for i in range(5))))
    print("   if every type took ONE trip (a lane per query): trips per wave mean %.2f; cycles per trip by type (mean): %s" % (
        types.mean(), ", ".join("%s %.0f" % (nm[i], (ty[i][sel].sum() / max(1.0, tc[i][sel].sum()))) for i in range(5))))
    heavy = st[5][sel] >= np.percentile(st[5][sel], 95)
    print("   slowest 5 %%: trips mean %.2f, types mean %.2f" % (trips[heavy].mean(), types[heavy].mean()))
    print("   slowest 5 %% of the waves: " + ", ".join("%s %.0f" % (nm[i], ty[i][sel][heavy].mean()) for i in range(5)))

# which (age, world) make a wave slow?  Only the envs stepped by the LAST full step (not at rest now) carry stamps of one and
# the same launch; their age at that step = step_count now (the step kernel that followed incremented it) - 1.
mp = vec.field(N.F_MAP_ID).cpu().numpy() if hasattr(N, "F_MAP_ID") else None
lastm = sel & (q == 0)
print("envs in the last full step: %d" % lastm.sum())
keys, inv = np.unique(tot[lastm], return_inverse=True)
rw = []
for w in range(len(keys)):
    m = np.nonzero(lastm)[0][inv == w]
    rw.append((keys[w], len(m), age[m].min(), age[m].max(), (mp[m].min() if mp is not None else -1), (mp[m].max() if mp is not None else -1),
               st[9][m].max(), st[7][m].max(), st[8][m].max(), st[6][m].max(), (st[2] - st[1])[m][0], (st[4] - st[3])[m][0]))
rw.sort()
print("last step's waves: %d; slowest 14 and fastest 3: total, lanes, age min..max (now), map min..max, queries, gjk, epa, n_act, collide, solver" % len(rw))
for r in rw[-14:] + rw[:3]:
    print("   ", " ".join("%7.0f" % v for v in r))
for lo, hi in ((1, 2), (2, 3), (3, 4), (4, 5), (5, 7), (7, 9), (9, 2000)):
    v = [r[0] for r in rw if r[2] >= lo and r[3] < hi]
    if v:
        print("   waves with all lanes at age(now) in [%d,%d): %4d, total cycles median %.0f max %.0f" % (lo, hi, len(v), np.median(v), max(v)))
