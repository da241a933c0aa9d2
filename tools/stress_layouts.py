#!/usr/bin/env python3
"""Race hunt (GPU box): random (env count, beam count, bank size / gathered, launch length) — a fused trajectory rollout against
the same steps launched one by one on a second handle, bitwise, slot by slot.  Any ordering bug in the tile's hand-overs (four or
six wave roles) shows up as a mismatch sooner or later.  SECONDS = how long to keep going (default 240)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
rng = random.Random(int(os.environ.get("SEED", "2026")))
t_end = time.time() + float(os.environ.get("SECONDS", "240"))
cases = steps = c4_cases = ring_cases = 0
while time.time() < t_end:
    n = rng.choice([1, 63, 64, 65, 200, 1000, 4096, 5000, 16384, 16385, 20000, 32768, 40000, 65536])
    nb = rng.choice([1, 2, 4, 5, 8, 10, 12, 16])
    kw = rng.choice([{"n_maps": 64}, {"n_maps": 64, "bank_in_global": True}, {"n_maps": 150}, {"n_maps": 7}])
    K = rng.choice([1, 2, 3, 7, 20, 50, 100, 101, 130])
    if n >= 32768 and K > 50: K = 50
    ships = 4 if rng.random() < float(os.environ.get("C4_SHARE", "0.25")) else 1
    if ships == 4:  # config 4: the memoised full step (fused API = its loop of two launches per step) against single steps WITHOUT the memo
        n = min(n, 20000); K = min(K, 50); kw = dict(kw); kw.pop("bank_in_global", None)
        a = ShipVecEnv(n, n_beams=nb, n_ships=4, **kw); b = ShipVecEnv(n, n_beams=nb, n_ships=4, dyn_memo=False, **kw)
    elif rng.random() < float(os.environ.get("RING_SHARE", "0.15")):  # a brand-new world per episode: rings refilled between launches
        n = min(n, 20000); ring = rng.choice([2, 3, 8, 33, 128]); ships = 0
        if n * ring > 600000: ring = 8
        kw = {"map_mode": "fresh_device", "ring": ring, "map_seed": rng.randrange(1 << 20)}
        a = ShipVecEnv(n, n_beams=nb, **kw); b = ShipVecEnv(n, n_beams=nb, **kw)
    else:
        a = ShipVecEnv(n, n_beams=nb, **kw); b = ShipVecEnv(n, n_beams=nb, **kw)
    a.reset_tensor(); b.reset_tensor()
    warm = rng.choice([0, 5, 33])
    acts = a.random_actions(rng.randrange(1 << 30), 0, warm + K)
    if warm:
        a.rollout_tensor(acts[:warm]); b.rollout_tensor(acts[:warm])
    to, tr, td, tf = a.rollout_tensor(acts[warm:], trajectory=True)
    for k in range(K):
        o, r, d, f = b.step_tensor(acts[warm + k])
        if not (torch.equal(to[k], o) and torch.equal(tr[k], r) and torch.equal(td[k], d) and torch.equal(tf[k], f)):
            print("MISMATCH n=%d nb=%d ships=%d %s K=%d warm=%d at step %d geometry %s" % (n, nb, ships, kw, K, warm, k, a.launch_geometry()))
            sys.exit(1)
    from ship_sim_gym_amd import _native as NN
    fids = list(range(12)) + ([NN.F_TRAFFIC, NN.F_GOAL_BODIES, NN.F_DYN_FLAGS] if ships == 4 else [])
    for fid in fids:
        if not torch.equal(a.field(fid), b.field(fid)):
            print("STATE MISMATCH field %d n=%d nb=%d ships=%d %s K=%d" % (fid, n, nb, ships, kw, K)); sys.exit(1)
    cases += 1; steps += K; c4_cases += ships == 4; ring_cases += ships == 0
    a.close(); b.close()
print("stress: %d random cases (%d of config 4: memo on, fused API, against memo off, single steps; %d on per-env rings of fresh worlds), %d fused steps compared slot by slot with single-step launches, all bitwise equal" % (cases, c4_cases, ring_cases, steps))
