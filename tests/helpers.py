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


class OracleSample:
    """The oracle on a random SAMPLE of a large batch's envs (BASELINE sizes: 65 536 .. 1 048 576 envs).  Envs are independent
    and everything an env sees is keyed by its GLOBAL id — bank record (env_id_base + e) mod n_maps, Philox action stream — so
    the oracle world of env e stepped with column e of the batch's action tensor must reproduce row e of every output, whatever
    the batch size: reward / done bit-exact, observations within `atol`, every step."""

    def __init__(self, O, vec, m=2048, seed=0):
        n = vec.num_envs
        rng = np.random.RandomState(seed)
        self.idx = np.sort(rng.choice(n, size=min(m, n), replace=False))
        # (the first and the last env, a wave boundary and a workgroup boundary are always in)
        self.idx = np.unique(np.concatenate([self.idx, [0, n - 1, min(n - 1, 63), min(n - 1, 64), min(n - 1, 255), min(n - 1, 256)]]))
        self.ob = O.Batch(len(self.idx), oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals,
                          map_ids=(vec.env_id_base + self.idx) % vec.n_maps)
        self.worst, self.n_done, self.steps = 0.0, 0, 0
        import torch
        self.tidx = torch.as_tensor(self.idx, device=vec.device)

    def reset(self, obs_gpu):
        np.testing.assert_array_equal(obs_gpu[self.tidx].cpu().numpy(), self.ob.reset())

    def step(self, acts_k, obs, rew, done, atol=1e-5, n_threads=8):
        """acts_k: this step's [n] action row (device tensor); obs / rew / done: the batch's outputs of the step."""
        a = acts_k[self.tidx].cpu().numpy()
        r_obs, r_rew, r_done = self.ob.step(a, auto_reset=True, n_threads=n_threads)
        np.testing.assert_array_equal(done[self.tidx].cpu().numpy(), r_done, err_msg="sampled envs: done differs at step %d" % self.steps)
        np.testing.assert_array_equal(rew[self.tidx].cpu().numpy(), r_rew, err_msg="sampled envs: reward differs at step %d" % self.steps)
        err = float(np.max(np.abs(obs[self.tidx].cpu().numpy() - r_obs)))
        assert err <= atol, "sampled envs: obs differ by %g at step %d" % (err, self.steps)
        self.worst = max(self.worst, err)
        self.n_done += int(r_done.sum())
        self.steps += 1
