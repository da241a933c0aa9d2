"""Shared helpers for the parity tests: run the HIP path and the oracle on the same bank and action stream."""
import numpy as np


def oracle_cfg(O, vec):
    """ora_config equivalent of a ShipVecEnv's ssg_config."""
    c = vec.cfg
    import math
    return O.default_config(width=c.width, height=c.height, dt=c.dt, space_damping=0.4, max_steps=c.max_steps,
                            history=c.history, n_beams=c.n_beams, lidar_spread_deg=c.lidar_spread_deg,
                            lidar_dist=c.lidar_dist, n_goals=c.n_goals, goal_radius=c.goal_radius,
                            spawn_x=c.spawn_x, spawn_y=c.spawn_y, thrust_px0=c.thrust_px0, thrust_py0=c.thrust_py0,
                            n_traffic=(3 if getattr(vec, "n_ships", 1) > 1 else 0))


def run_pair(O, N, vec, K, seed=12345, check_every=1, atol=1e-5):
    """Step `vec` (HIP, bank mode) and an oracle Batch K times on the same Philox actions; compare every output.
    Returns (max_abs_obs_err, n_done_total)."""
    import torch
    n = vec.num_envs
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals,
                 map_ids=(vec.env_id_base + np.arange(n)) % vec.n_maps)
    o_ref = ob.reset()
    o_gpu = vec.reset_tensor().cpu().numpy()
    np.testing.assert_array_equal(o_gpu, o_ref)
    acts = vec.random_actions(seed, 0, K)
    acts_h = acts.cpu().numpy()
    np.testing.assert_array_equal(acts_h, O.fill_actions(seed, 0, K, vec.env_id_base, n))
    max_err, n_done = 0.0, 0
    for k in range(K):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=True)
        if k % check_every == 0 or k == K - 1:
            g_obs, g_rew, g_done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            np.testing.assert_array_equal(g_done, r_done, err_msg="done flags differ at step %d" % k)
            np.testing.assert_array_equal(g_rew, r_rew, err_msg="rewards differ at step %d" % k)
            err = float(np.max(np.abs(g_obs - r_obs)))
            assert err <= atol, "obs differ by %g at step %d" % (err, k)
            max_err = max(max_err, err)
            n_done += int(r_done.sum())
    return max_err, n_done
