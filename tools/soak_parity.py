#!/usr/bin/env python3
"""Long randomized parity soak on the GPU box (not part of the test suite): the HIP path against the oracle over
millions of env-steps with auto-reset, configs 3 and 4, single-step launches and (round 2) 100-step fused launches of the
pipelined kernel at the full 65 536 envs.  Round 1: 12 M + 20 M env-steps, 780 k episode ends, rewards / done flags
identical, max |obs - oracle| 1.1e-10; round 2 (pipelined kernel): the same, plus 26 M env-steps fused."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from ship_sim_gym_amd.vec_env import ShipVecEnv
from helpers import oracle_cfg


def soak(n, K, seed=777, **kw):
    vec = ShipVecEnv(n, n_maps=kw.pop("n_maps", 64), **kw)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts = vec.random_actions(seed, 0, K)
    ah = acts.cpu().numpy()
    worst, t0, ndone = 0.0, time.time(), 0
    for k in range(K):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(ah[k], auto_reset=True, n_threads=O.max_threads())
        assert np.array_equal(done.cpu().numpy(), r_done), ("done", k)
        assert np.array_equal(rew.cpu().numpy(), r_rew), ("reward", k)
        worst = max(worst, float(np.abs(obs.cpu().numpy() - r_obs).max()))
        ndone += int(r_done.sum())
    print(kw, "n=%d K=%d: max |obs-oracle| %.3e, %d episode ends, %.1f s" % (n, K, worst, ndone, time.time() - t0), flush=True)
    assert worst <= 1e-9
    vec.close()


def soak_fused(n, chunks, seed=99, **kw):
    """ssg_rollout in launches of 100 fused steps against the oracle stepped 100 times: the last step's outputs and the
    body state after every launch."""
    from ship_sim_gym_amd import _native as N
    vec = ShipVecEnv(n, n_maps=64, **kw)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts = vec.random_actions(seed, 0, 100 * chunks)
    ah = acts.cpu().numpy()
    worst, t0 = 0.0, time.time()
    for c in range(chunks):
        obs, rew, done, flags = vec.rollout_tensor(acts[100 * c: 100 * c + 100])
        for k in range(100 * c, 100 * c + 100):
            r_obs, r_rew, r_done = ob.step(ah[k], auto_reset=True, n_threads=O.max_threads())
        assert np.array_equal(done.cpu().numpy(), r_done), ("done", c)
        assert np.array_equal(rew.cpu().numpy(), r_rew), ("reward", c)
        worst = max(worst, float(np.abs(obs.cpu().numpy() - r_obs).max()))
        pk = ob.peek_all()
        worst = max(worst, float(np.abs(vec.field(N.F_X).cpu().numpy() - pk[:, 0]).max()), float(np.abs(vec.field(N.F_W).cpu().numpy() - pk[:, 5]).max()))
        assert np.array_equal(vec.field(N.F_STEP_COUNT).cpu().numpy(), pk[:, 7].astype(np.int32)), ("step_count", c)
    print(kw, "fused: n=%d, %d launches of 100 steps: max |obs/state - oracle| %.3e, %.1f s" % (n, chunks, worst, time.time() - t0), flush=True)
    assert worst <= 1e-9
    vec.close()


def soak_traj(n, chunks, seed=2026, **kw):
    """ssg_rollout_traj (round 3): EVERY step of launches of 100 fused steps against the oracle, at the full env count."""
    vec = ShipVecEnv(n, n_maps=kw.pop("n_maps", 64), **kw)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    worst, t0, ndone = 0.0, time.time(), 0
    for c in range(chunks):
        acts = vec.random_actions(seed, 100 * c, 100)
        ah = acts.cpu().numpy()
        to, tr, td, tf = [t.cpu().numpy() for t in vec.rollout_tensor(acts, trajectory=True)]
        for k in range(100):
            r_obs, r_rew, r_done = ob.step(ah[k], auto_reset=True, n_threads=O.max_threads())
            assert np.array_equal(td[k], r_done), ("done", c, k)
            assert np.array_equal(tr[k], r_rew), ("reward", c, k)
            worst = max(worst, float(np.abs(to[k] - r_obs).max()))
            ndone += int(r_done.sum())
    print(kw, "trajectory: n=%d, %d launches of 100 steps, every step: max |obs - oracle| %.3e, %d episode ends, %.1f s" % (
        n, chunks, worst, ndone, time.time() - t0), flush=True)
    assert worst <= 1e-9
    vec.close()


if __name__ == "__main__":
    if "--long" in sys.argv:  # several minutes: the round's extended soak (profiles/r3/soak_parity.txt)
        from ship_sim_gym_amd.config import GameConfig

        class Train(GameConfig):  # train/stable_baselines/ppo.py:65-69
            SPEED = 30
            BOUNDS = (1000, 1000)

        soak_traj(65536, 20, n_beams=8)                               # 131 M env-steps, every fused step
        soak_traj(65536, 6, seed=7, n_beams=10)                       # the default 10-beam lidar
        soak(16384, 3000, seed=11, n_beams=10, n_ships=4)             # 49 M env-steps of config 4
        soak(16384, 1500, seed=12, n_beams=8, game_config=Train, n_maps=32)  # the training configuration (episodes of ~6 steps)
        soak(4096, 2500, seed=13, n_beams=16)                         # 16 beams: the bank gathered from L2, record heads in LDS
        sys.exit(0)
    if "--extra" in sys.argv:  # round 5: other seeds, and the layouts the round added
        soak_traj(65536, 10, seed=35, n_beams=8)                      # the headline layout, another action stream
        soak_traj(65536, 6, seed=31, n_beams=8, n_maps=120)           # a bank too large for the LDS: gathered on 256-env workgroups
        soak(20000, 1500, seed=32, n_beams=10)                        # 128-env workgroups, six wave roles
        soak(32768, 800, seed=33, n_beams=8)
        soak(4096, 4000, seed=36, n_beams=10)                         # BASELINE configs[1]: 64-env workgroups, six wave roles
        soak(16384, 2000, seed=34, n_beams=10, n_ships=4)             # config 4, another action stream
        sys.exit(0)
    soak_traj(65536, 4, n_beams=8)
    soak_fused(65536, 4, n_beams=8)
    soak(8192, 1500, n_beams=8)
    soak(8192, 1200, n_beams=10, n_ships=4)
    soak(16384, 600, seed=4, n_beams=10, n_ships=4)
