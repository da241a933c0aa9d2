#!/usr/bin/env python3
"""How much does each NAMED, UNVERIFIABLE assumption of the oracle matter?  (VERDICT r1 "next" item 2.)

pymunk 5.4.0 / Chipmunk2D cannot be installed here, so the oracle's physics is "parity unpinned" (oracle/ssg_oracle.h).
The assumptions it rests on are each behind a switch (ORA_VAR_*, ssg_oracle.h / ora_config.thrust_px0).  This tool
runs the oracle with ONE switch flipped against the baseline on the same bank and Philox action stream and reports
what fraction of the observable stream (observations at the 1e-5 tolerance of BASELINE.json, rewards, done flags)
changes, plus the census of player pairs on which the SAT predicate (what the oracle and the HIP kernels evaluate for
`collide_ship`) and the restated cpCollide (GJK/EPA + ContactPoints: `begin` fires iff it pushes >= 1 contact) disagree.

    python tools/assumption_sensitivity.py [--c3-envs 16384 --c3-steps 700 --c4-envs 8192 --c4-steps 1300] > table.md

CPU only (test infrastructure).  Results are committed in DESIGN.md §3 and asserted loosely by
tests/test_assumptions.py on a smaller run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ATOL = 1e-5


def run_pair(O, cfg_base, cfg_var, polys, goals, n, K, threads, chunk=50, census=False):
    """Step a baseline batch and a variant batch side by side; returns the divergence statistics."""
    a = O.Batch(n, cfg_base, polys, goals)
    b = O.Batch(n, cfg_var, polys, goals)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    diverged_at = np.full(n, -1, dtype=np.int64)
    obs_mis = rew_mis = done_mis = 0
    max_obs = 0.0
    k = 0
    while k < K:
        kk = min(chunk, K - k)
        acts = O.fill_actions(12345, k, kk, 0, n)
        for j in range(kk):
            oa, ra, da = a.step(acts[j], n_threads=threads)
            ob, rb, db = b.step(acts[j], n_threads=threads)
            dob = np.abs(oa - ob).max(axis=1)
            bad_o = dob > ATOL
            bad_r = ra != rb
            bad_d = da != db
            obs_mis += int(bad_o.sum()); rew_mis += int(bad_r.sum()); done_mis += int(bad_d.sum())
            fin = dob[np.isfinite(dob)]
            if fin.size:
                max_obs = max(max_obs, float(fin.max()))
            newly = (diverged_at < 0) & (bad_o | bad_r | bad_d)
            diverged_at[newly] = k + j
        k += kk
    total = n * K
    exposure = np.where(diverged_at >= 0, diverged_at + 1, K).sum()  # env-steps lived before the first difference
    out = {"env_steps": total, "obs_mismatch_frac": obs_mis / total, "reward_mismatch_frac": rew_mis / total,
           "done_mismatch_frac": done_mis / total, "envs_diverged_frac": float((diverged_at >= 0).mean()),
           "first_divergence_per_env_step": float((diverged_at >= 0).sum() / max(exposure, 1)),
           "max_abs_obs_diff_where_finite": max_obs}
    if census:
        out["census"] = b.counters()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3-envs", type=int, default=16384)
    ap.add_argument("--c3-steps", type=int, default=700)
    ap.add_argument("--c4-envs", type=int, default=8192)
    ap.add_argument("--c4-steps", type=int, default=1300)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from oracle import oracle as O
    from ship_sim_gym_amd import worldgen  # host geometry via libshipsim's ssg_host_* (no GPU needed)
    threads = args.threads or O.max_threads()
    recs, polys, goals = worldgen.build_bank(64, (600, 600))
    results = {}
    t0 = time.time()

    def cfg(n_traffic, n_beams, **kw):
        return O.default_config(n_beams=n_beams, n_traffic=n_traffic, **kw)

    # ---- config 3 (1 ship, 8 beams): assumptions that touch the player path -------------------------------------------
    n, K = args.c3_envs, args.c3_steps
    base3 = cfg(0, 8)
    results["c3"] = {
        "CHECK_SAT (census only; behaviour = baseline)": run_pair(O, base3, cfg(0, 8, variant=O.VAR_CHECK_SAT), polys, goals, n, K, threads, census=True),
        "TOUCH: touching does not count (`<` instead of `<=`)": run_pair(O, base3, cfg(0, 8, variant=O.VAR_TOUCH_STRICT), polys, goals, n, K, threads),
        "PLAYER colliding from cpCollide's contact count instead of SAT": run_pair(O, base3, cfg(0, 8, variant=O.VAR_PLAYER_CPCOLLIDE), polys, goals, n, K, threads),
        "THRUST_PX0 = bb centre (10, 22.5) instead of (0, 0)": run_pair(O, base3, cfg(0, 8, thrust_px0=10.0, thrust_py0=22.5), polys, goals, n, K, threads),
    }
    # ---- config 4 (4 ships, 10 beams): the solver's assumptions ------------------------------------------------------------
    n, K = args.c4_envs, args.c4_steps
    base4 = cfg(3, 10)
    results["c4"] = {
        "CHECK_SAT (census only; behaviour = baseline)": run_pair(O, base4, cfg(3, 10, variant=O.VAR_CHECK_SAT), polys, goals, n, K, threads, census=True),
        "TOUCH: touching does not count": run_pair(O, base4, cfg(3, 10, variant=O.VAR_TOUCH_STRICT), polys, goals, n, K, threads),
        "ORDER: solver walks the arbiter list in reverse": run_pair(O, base4, cfg(3, 10, variant=O.VAR_ORDER_REVERSED), polys, goals, n, K, threads),
        "ORDER: poly-poly pairs collided with a/b exchanged": run_pair(O, base4, cfg(3, 10, variant=O.VAR_SWAP_AB), polys, goals, n, K, threads),
        "GJK-ID: warm start from the cached collision id": run_pair(O, base4, cfg(3, 10, variant=O.VAR_GJK_WARM), polys, goals, n, K, threads),
        "PLAYER colliding from cpCollide's contact count instead of SAT": run_pair(O, base4, cfg(3, 10, variant=O.VAR_PLAYER_CPCOLLIDE), polys, goals, n, K, threads),
    }
    results["meta"] = {"threads": threads, "seconds": time.time() - t0, "atol": ATOL,
                       "c3": "%d envs x %d steps, 8 beams, 64-map bank, Philox seed 12345" % (args.c3_envs, args.c3_steps),
                       "c4": "%d envs x %d steps, 4 ships, 10 beams" % (args.c4_envs, args.c4_steps)}
    if args.json:
        json.dump(results, open(args.json, "w"), indent=1)
    for cname in ("c3", "c4"):
        print("\n### %s — %s\n" % (cname.upper(), results["meta"][cname]))
        print("| assumption flipped | obs rows changed (>1e-5) | rewards changed | done flags changed | envs that ever diverged | first divergence per env-step |")
        print("|---|---|---|---|---|---|")
        for name, r in results[cname].items():
            print("| %s | %.3g | %.3g | %.3g | %.3g | %.3g |" % (name, r["obs_mismatch_frac"], r["reward_mismatch_frac"],
                                                             r["done_mismatch_frac"], r["envs_diverged_frac"],
                                                             r["first_divergence_per_env_step"]))
            if "census" in r:
                c = r["census"]
                print("| &nbsp;&nbsp;census over %d env-steps | player pairs past the AABB test: %d | SAT != cpCollide(player, other): %d | "
                      "SAT != cpCollide(other, player): %d | pairs with abs(distance) < 1e-9: %d | |"
                      % (r["env_steps"], c["checked"], c["disagree_ab"], c["disagree_ba"], c["near_zero"]))
    print("\n(%.0f s on %d threads)" % (results["meta"]["seconds"], threads))


if __name__ == "__main__":
    main()
