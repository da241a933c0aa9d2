#!/usr/bin/env python3
"""Development aid: which states does the memo of the full dyn step still LEARN in steady state?  Runs config 4, then reads the
table out of the state blob and lists the entries born after the warm-up."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from ship_sim_gym_amd import _native as N
from ship_sim_gym_amd.vec_env import ShipVecEnv

n, W, K = int(os.environ.get("N", "65536")), int(os.environ.get("W", "60")), int(os.environ.get("K", "200"))
vec = ShipVecEnv(n, n_beams=10, n_maps=64, n_ships=4)
acts = vec.random_actions(12345, 0, K + W)
vec.reset_tensor()
vec.rollout_tensor(acts[:W])
vec.rollout_tensor(acts[W:])
torch.cuda.synchronize()
o_, es, nc, st = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
N.check(N.lib().ssg_state_field(vec._h, N.F_DYN_MEMO_STATS, C.byref(o_), C.byref(es), C.byref(nc), C.byref(st)), vec._h, "field")
STRIDE, ENTRIES, KEY, VAL = 244, 1 << 14, 4, 4 + 102
off = o_.value + 256 * 16 * 8
tab = vec.state[off: off + ENTRIES * STRIDE * 8].view(torch.int64).view(ENTRIES, STRIDE).cpu().numpy()
used = tab[:, 0] != 0
born = tab[:, 2]
print("entries in use %d, ready %d; born histogram by 20 launches:" % (used.sum(), (tab[:, 0] == tab[:, 1])[used].sum()))
print(np.bincount((born[used] // 20).astype(int)))
late = np.where(used & (born > W + 5))[0]
f = tab.view(np.float64)
rows = []
for i in late:
    hdr = int(tab[i, KEY]) & 0xFFFFFFFFFFFFFFFF
    m, incl, nl = hdr & 0xFF, (hdr >> 8) & 0xFF, (hdr >> 16) & 0xFF
    live = int(tab[i, KEY + 1]) & 0xFFFFFFFFFFFFFFFF
    vh = int(tab[i, VAL])
    ships = f[i, KEY + 2: KEY + 29].reshape(3, 9)
    rows.append((int(born[i]), m, incl, nl, live, vh & 1, (vh >> 8) & 0xFF, (vh >> 16) & 0xFF, ships))
rows.sort(key=lambda r: (r[1], r[0]))
for r in rows[:120]:
    s = r[8]
    print("born %4d map %2d incl %02x n_live %d live %014x changed %d n_out %d n_aged %d | ship0 (%.6f %.6f a %.3e vb %.2e %.2e) ship1 (%.4f %.4f vb %.2e) ship2 (%.4f %.4f vb %.2e)" % (
        r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], s[0, 0], s[0, 1], s[0, 2], s[0, 6], s[0, 7], s[1, 0], s[1, 1], max(abs(s[1, 6]), abs(s[1, 7])), s[2, 0], s[2, 1], max(abs(s[2, 6]), abs(s[2, 7]))))
print("late entries: %d; by map: %r" % (len(rows), np.bincount([r[1] for r in rows], minlength=64).tolist()))
print(vec.dyn_memo_stats())
# which key / value words differ between consecutive late entries of one record?
names = ["hdr", "live"] + ["ship%d.%s" % (k, f) for k in range(3) for f in ("x", "y", "a", "vx", "vy", "w", "vbx", "vby", "wb")] + ["pad"] + \
        ["goal%d.%s" % (g, f) for g in range(6) for f in ("x", "y", "vx", "vy", "vbx", "vby", "w", "wb")] + \
        ["arb%d.%s" % (a, f) for a in range(4) for f in ("ids", "jn0", "jn1", "jt0", "jt1", "pad")]
for m in sorted(set(r[1] for r in rows)):
    idx = [i for i in late if (int(tab[i, KEY]) & 0xFF) == m and ((int(tab[i, KEY]) >> 8) & 0xFF) != 0]
    idx.sort(key=lambda i: born[i])
    print("map %d: %d late entries with a participating goal" % (m, len(idx)))
    for a, b in list(zip(idx, idx[1:]))[:6]:
        d = [names[j] for j in range(102) if tab[a, KEY + j] != tab[b, KEY + j]]
        vals = [(names[j], float(f[a, KEY + j]), float(f[b, KEY + j])) for j in range(2, 102) if tab[a, KEY + j] != tab[b, KEY + j] and not names[j].endswith("ids")]
        print("   born %d -> %d: differ in %r" % (born[a], born[b], d))
        for nme, x, y in vals[:12]:
            print("        %-10s %.17g -> %.17g" % (nme, x, y))
