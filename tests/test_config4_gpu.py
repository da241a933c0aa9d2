"""Config 4 (BASELINE configs[3]: 65 536 envs x 4 ships, curriculum maps): the HIP dyn kernel (traffic ships, dynamic
goal bodies, Chipmunk contact solver) + the DYN step kernel against the oracle's restatement (oracle/ssg_dynamics.c)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    assert torch.cuda.is_available()
    from oracle import oracle as O
    from ship_sim_gym_amd import _native as N
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    return torch, O, N, ShipVecEnv


def _dyn_state(N, vec):
    t = vec.field(N.F_TRAFFIC).cpu().numpy()          # [27, n]
    g = vec.field(N.F_GOAL_BODIES).cpu().numpy()      # [48, n]
    n = t.shape[1]
    traffic = t.reshape(3, 9, n).transpose(2, 0, 1)[:, :, :6]                      # [n, 3, (x,y,a,vx,vy,w)]
    goals = g.reshape(6, 8, n).transpose(2, 0, 1)[:, :5, :4]                       # [n, 5, (x,y,vx,vy)]
    return traffic, goals


def _oracle_dyn(ob, idx):
    tr, go = [], []
    for i in idx:
        d = ob.peek_dyn(int(i))
        tr.append(d["traffic"]); go.append(d["goals"])
    return np.stack(tr), np.stack(go)


def _compare_dyn(N, vec, ob, just_reset=None, atol=1e-9):
    """Traffic / goal body columns against the oracle's bodies.  The body COLUMNS of an auto-reset env lag its reset by ONE
    step: the step kernel only flags the env (dyn flag bit 1) and the next dyn step rebuilds its bodies (ShipGame.reset +
    add_default_traffic) and writes every column.  `just_reset` = the envs reset by the step just taken are left out."""
    idx = np.arange(vec.num_envs)
    keep = np.ones(vec.num_envs, dtype=bool) if just_reset is None else ~just_reset.astype(bool)
    t_g, g_g = _dyn_state(N, vec)
    t_o, g_o = _oracle_dyn(ob, idx)
    np.testing.assert_allclose(t_g[keep], t_o[keep], atol=atol, rtol=0)
    mask = vec.field(N.F_GOAL_MASK).cpu().numpy()
    for g in range(5):                                   # bodies of goals that left the space are not maintained
        alive = (mask >> g & 1).astype(bool) & keep
        np.testing.assert_allclose(g_g[alive, g], g_o[alive, g], atol=atol, rtol=0)


# (banks of more than 64 records: the full dyn step's waves no longer sit on ONE record each — its per-lane-planes variant)
@pytest.mark.parametrize("n,nb,K,n_maps", [(4096, 10, 160, 64), (777, 8, 120, 64), (2048, 10, 100, 96),
                                           (3000, 8, 60, 1)])  # one world: after the reset ONE bucket of the dyn queue holds every env
def test_config4_parity(n, nb, K, n_maps):
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    vec = ShipVecEnv(n, n_beams=nb, n_maps=n_maps, n_ships=4)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    _compare_dyn(N, vec, ob, atol=0)
    acts = vec.random_actions(4242, 0, K)
    acts_h = acts.cpu().numpy()
    max_err, n_done, n_col = 0.0, 0, 0
    r_done = np.zeros(n, dtype=np.uint8)
    for k in range(K):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=True, n_threads=8)
        np.testing.assert_array_equal(done.cpu().numpy(), r_done, err_msg="done differs at step %d" % k)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_rew, err_msg="reward differs at step %d" % k)
        err = float(np.max(np.abs(obs.cpu().numpy() - r_obs)))
        assert err <= 1e-5, "obs differ by %g at step %d" % (err, k)     # BASELINE tolerance
        max_err = max(max_err, err)
        n_done += int(r_done.sum())
        n_col += int(((flags.cpu().numpy() & N.EV_COLLIDING) != 0).sum())
        if k in (0, 1, 2, 7, 40, K - 1):
            _compare_dyn(N, vec, ob, just_reset=r_done)
    assert n_done > n // 4 and n_col > 0
    assert max_err <= 1e-9
    vec.close()


@pytest.mark.parametrize("n,nb,K,n_maps", [(4096, 10, 150, 64), (1500, 8, 90, 96)])
def test_config4_rollout_matches_oracle_every_step(n, nb, K, n_maps):
    """ssg_rollout_traj on config 4 (the loop of train/random.py:14-27 with add_default_traffic): every step of two
    back-to-back trajectory rollouts is compared with the oracle, the bodies at the end, and the whole thing bit for bit with
    the same steps taken one ssg_step at a time."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    vec = ShipVecEnv(n, n_beams=nb, n_maps=n_maps, n_ships=4)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts = vec.random_actions(777, 0, K)
    acts_h = acts.cpu().numpy()
    K1 = K // 3                                              # two calls
    parts = [vec.rollout_tensor(acts[:K1], trajectory=True), None]
    parts[0] = [t.clone() for t in parts[0]]
    parts[1] = [t.clone() for t in vec.rollout_tensor(acts[K1:], trajectory=True)]
    g_obs, g_rew, g_done, g_flags = [torch.cat([a, b]).cpu().numpy() for a, b in zip(*parts)]
    worst, n_done, n_goal = 0.0, 0, 0
    r_done = np.zeros(n, dtype=np.uint8)
    for k in range(K):
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=True, n_threads=8)
        np.testing.assert_array_equal(g_done[k], r_done, err_msg="done differs at rollout step %d" % k)
        np.testing.assert_array_equal(g_rew[k], r_rew, err_msg="reward differs at rollout step %d" % k)
        err = float(np.max(np.abs(g_obs[k] - r_obs)))
        assert err <= 1e-5, "obs differ by %g at rollout step %d" % (err, k)
        worst = max(worst, err)
        n_done += int(r_done.sum()); n_goal += int((r_rew == 1.0).sum())
    assert worst <= 1e-9 and n_done > n // 4 and n_goal > 0
    _compare_dyn(N, vec, ob, just_reset=r_done)
    # the same steps, one launch sequence per step
    b = ShipVecEnv(n, n_beams=nb, n_maps=n_maps, n_ships=4)
    b.reset_tensor()
    for k in range(K):
        o, r, d, f = b.step_tensor(acts[k])
        assert torch.equal(o, torch.from_numpy(g_obs[k]).to(o.device)), "step %d" % k
        assert torch.equal(d, torch.from_numpy(g_done[k]).to(d.device)) and torch.equal(f, torch.from_numpy(g_flags[k]).to(f.device))
    keep = torch.from_numpy(~r_done.astype(bool)).to(b.device)
    assert torch.equal(b.field(N.F_TRAFFIC)[:, keep], vec.field(N.F_TRAFFIC)[:, keep])
    assert torch.equal(b.field(N.F_X), vec.field(N.F_X)) and torch.equal(b.field(N.F_GOAL_MASK), vec.field(N.F_GOAL_MASK))
    vec.close(); b.close()


def test_config4_solver_scenarios():
    """Scenarios that exercise what random rollouts rarely reach: ship-ship impact with friction, a goal shoved by a
    moving ship, a ship driven into a bank, the player running into parked traffic."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    n = 256
    vec = ShipVecEnv(n, n_maps=16, n_ships=4)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    vec.reset_tensor(); ob.reset()
    vec.step_tensor(torch.ones(n, dtype=torch.int32, device=vec.device)); ob.step(np.ones(n, dtype=np.int32))
    rng = np.random.RandomState(7)
    T = vec.field(N.F_TRAFFIC)                            # [27, n] view into the state blob
    poke = np.zeros((n, 3, 6))
    for e in range(n):
        kind = e % 4
        if kind == 0:      # ship 2 thrown at ship 3, off-centre, spinning
            poke[e, 1] = [370 + rng.uniform(-5, 5), 352 + rng.uniform(-10, 25), rng.uniform(-.3, .3), 12, rng.uniform(-2, 2), rng.uniform(-.05, .05)]
            poke[e, 0] = [100, 200, 0, 0, 0, 0]; poke[e, 2] = [400, 350, 0, 0, 0, 0]
        elif kind == 1:    # ship 3 driven into the right bank
            poke[e, 2] = [430 + rng.uniform(0, 15), 300 + rng.uniform(-50, 50), rng.uniform(-1, 1), 15, rng.uniform(-3, 3), rng.uniform(-.1, .1)]
            poke[e, 0] = [100, 200, 0, 0, 0, 0]; poke[e, 1] = [300, 200, 0, 0, 0, 0]
        elif kind == 2:    # ship 2 sweeping up the fairway through the goals
            poke[e, 1] = [rng.uniform(270, 320), 60, rng.uniform(-.2, .2), rng.uniform(-1, 1), 14, 0]
            poke[e, 0] = [100, 200, 0, 0, 0, 0]; poke[e, 2] = [400, 350, 0, 0, 0, 0]
        else:              # ship 1 parked in front of the player
            poke[e, 0] = [295 + rng.uniform(-8, 8), 90 + rng.uniform(0, 30), rng.uniform(-.5, .5), 0, 0, 0]
            poke[e, 1] = [300, 200, 0, 0, 0, 0]; poke[e, 2] = [400, 350, 0, 0, 0, 0]
    for e in range(n):
        for k in range(3):
            ob.poke_traffic(e, k, *poke[e, k])
    cols = np.zeros((27, n))
    for k in range(3):
        cols[9 * k: 9 * k + 6] = poke[:, k].T
    T.copy_(torch.from_numpy(cols).to(vec.device))
    vec.wake_dynamics()
    # the live arbiters of ship 1 (resting on the left bank) refer to its old pose in both implementations alike
    K = 60
    acts = (np.arange(n)[None, :] % 4 == 3) * 0 + np.where(np.arange(n)[None, :] % 4 == 3, 0, 1) * np.ones((K, 1), dtype=np.int64)
    acts = acts.astype(np.int32)                          # kind 3 drives forward, the others only move the rudder
    n_col = 0
    struck = np.zeros(n, dtype=bool)
    r_done = np.zeros(n, dtype=np.uint8)
    for k in range(K):
        a = torch.from_numpy(acts[k]).to(vec.device)
        obs, rew, done, flags = vec.step_tensor(a)
        r_obs, r_rew, r_done = ob.step(acts[k], auto_reset=True)
        np.testing.assert_array_equal(done.cpu().numpy(), r_done, err_msg="done differs at step %d" % k)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_rew)
        assert float(np.max(np.abs(obs.cpu().numpy() - r_obs))) <= 1e-9
        _compare_dyn(N, vec, ob, just_reset=r_done, atol=1e-8)
        n_col += int(((flags.cpu().numpy() & N.EV_COLLIDING) != 0)[3::4].sum())
        struck |= np.abs(_dyn_state(N, vec)[0][:, 2, 3:]).max(axis=1) > 0      # ship 3 acquired a real velocity
    assert n_col >= n // 8                                # the player did run into the parked ships
    assert struck[0::4].sum() >= n // 16                  # ship-ship impacts happened (kind 0)
    vec.close()


def test_config4_full_size_properties():
    """BASELINE configs[3] at full size (65 536 envs x 4 ships, 10 beams): ~2 050 randomly chosen envs against the oracle at
    every step (reward / done exact, observations 1e-5; bodies at the end), plus properties of the whole batch."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import OracleSample
    n = 65536
    vec = ShipVecEnv(n, n_beams=10, n_maps=64, n_ships=4)
    smp = OracleSample(O, vec, 2048, seed=4)
    smp.reset(vec.reset_tensor())
    acts = vec.random_actions(99, 0, 80)
    for k in range(80):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        smp.step(acts[k], obs, rew, done, atol=1e-5)
    assert smp.n_done > 500 and smp.worst <= 1e-9
    # the sampled envs' bodies (traffic ships, goal circles) where the oracle's are; envs reset by the last step excluded
    t_all, g_all = _dyn_state(N, vec)
    t_o, g_o = _oracle_dyn(smp.ob, np.arange(len(smp.idx)))
    keep = ~smp.ob.done.astype(bool)
    np.testing.assert_allclose(t_all[smp.idx][keep], t_o[keep], atol=1e-8, rtol=0)
    mask = vec.field(N.F_GOAL_MASK).cpu().numpy()[smp.idx]
    for g in range(5):
        alive = (mask >> g & 1).astype(bool) & keep
        np.testing.assert_allclose(g_all[smp.idx][alive, g], g_o[alive, g], atol=1e-8, rtol=0)
    assert torch.isfinite(obs).all()
    t, g = _dyn_state(N, vec)
    assert np.isfinite(t).all() and np.isfinite(g).all()
    # envs that share a map and an action history are bit-identical (64 maps, env e starts on map e % 64, and the
    # Philox stream differs per env, so compare the traffic only: it never depends on the actions before a contact)
    assert np.abs(t[:, :, 3:]).max() < 50
    # no real velocity ever appears on a ship that only rests against its bank
    st = vec.stats()
    assert st["episodes"] > n // 2
    vec.close()


def test_config4_rest_bit_and_poke_at_rest():
    """The dyn step skips envs whose bodies are at a fixed point of cpSpaceStep (rest bit).  The skip must be invisible:
    parity continues while most envs rest, and a caller that writes the traffic columns wakes the env (wake_dynamics)."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    n = 512
    vec = ShipVecEnv(n, n_maps=32, n_ships=4, auto_reset=True)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    vec.reset_tensor(); ob.reset()
    ones = torch.ones(n, dtype=torch.int32, device=vec.device)   # rudder only: nobody dies, everything settles
    for k in range(14):
        obs, rew, done, flags = vec.step_tensor(ones)
        r_obs, r_rew, r_done = ob.step(np.ones(n, dtype=np.int32))
        np.testing.assert_array_equal(obs.cpu().numpy(), r_obs) if k < 2 else None
    fl = vec.field(N.F_DYN_FLAGS).cpu().numpy()
    assert ((fl & 4) != 0).mean() > 0.8                           # most spaces have reached their fixed point
    _compare_dyn(N, vec, ob, atol=1e-9)
    # throw ship 2 of every 3rd env at ship 3 while the env rests: the hash of the body columns no longer matches
    T = vec.field(N.F_TRAFFIC)
    cols = T.clone()
    idx = np.arange(0, n, 3)
    cols[9 + 0, idx] = 372.0; cols[9 + 1, idx] = 360.0; cols[9 + 3, idx] = 11.0; cols[9 + 5, idx] = 0.03
    T.copy_(cols)
    vec.wake_dynamics()                                        # the contract for writing body columns (ssg_dyn_invalidate)
    for e in idx:
        cur = ob.peek_dyn(int(e))["traffic"][1]
        ob.poke_traffic(int(e), 1, 372.0, 360.0, cur[2], 11.0, cur[4], 0.03)
    r_done = np.zeros(n, dtype=np.uint8)
    for k in range(25):
        obs, rew, done, flags = vec.step_tensor(ones)
        r_obs, r_rew, r_done = ob.step(np.ones(n, dtype=np.int32))
        np.testing.assert_array_equal(done.cpu().numpy(), r_done)
        assert float(np.max(np.abs(obs.cpu().numpy() - r_obs))) <= 1e-9
        _compare_dyn(N, vec, ob, just_reset=r_done, atol=1e-8)
    t, _ = _dyn_state(N, vec)
    assert (np.abs(t[idx, 2, 0] - 400.0) > 1e-3).mean() > 0.9     # ship 3 was really pushed in the poked envs
    vec.close()


def test_config4_player_moved_next_to_parked_ship():
    """Which envs the dyn kernels visit is decided at the end of a step from the player state the step kernel holds in
    registers: a caller that moves the PLAYER of a resting env (SSG_F_X .. SSG_F_VY) next to a parked ship must announce it
    (wake_dynamics / ssg_dyn_invalidate, include/shipsim.h) — then collide_ship (game.py:232-241) fires as in the oracle;
    masked form: only the written envs lose their rest bit."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    n = 512
    vec = ShipVecEnv(n, n_maps=32, n_ships=4, auto_reset=True)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    vec.reset_tensor(); ob.reset()
    ones = torch.ones(n, dtype=torch.int32, device=vec.device)   # rudder only: nobody moves, every space settles
    for k in range(14):
        vec.step_tensor(ones)
        ob.step(np.ones(n, dtype=np.int32))
    assert ((vec.field(N.F_DYN_FLAGS).cpu().numpy() & 4) != 0).mean() > 0.8
    F = 6 + vec.cfg.n_beams
    for rnd, masked in enumerate((False, True)):
        # every 2nd env's player onto ship 2 (parked at (300, 200), hull 15 x 30), every 5th well clear of everything
        hit = np.arange(rnd, n, 2); far = np.setdiff1d(np.arange(0, n, 5), hit)
        X, Y, VX, VY = vec.field(N.F_X), vec.field(N.F_Y), vec.field(N.F_VX), vec.field(N.F_VY)
        hi, fi = torch.as_tensor(hit, device=vec.device), torch.as_tensor(far, device=vec.device)
        X[hi] = 292.0; Y[hi] = 190.0; VX[hi] = 0.0; VY[hi] = 0.0
        X[fi] = 350.0; Y[fi] = 60.0; VX[fi] = 0.0; VY[fi] = 0.0
        for e in hit: ob.poke_player(int(e), 292.0, 190.0)
        for e in far: ob.poke_player(int(e), 350.0, 60.0)
        if masked:
            m = torch.zeros(n, dtype=torch.uint8, device=vec.device); m[hi] = 1; m[fi] = 1
            vec.wake_dynamics(m)
        else:
            vec.wake_dynamics()
        for k in range(3):
            obs, rew, done, flags = vec.step_tensor(ones)
            r_obs, r_rew, r_done = ob.step(np.ones(n, dtype=np.int32))
            g_done, g_flags = done.cpu().numpy(), flags.cpu().numpy()
            np.testing.assert_array_equal(g_done, r_done, err_msg="round %d step %d" % (rnd, k))
            np.testing.assert_array_equal(rew.cpu().numpy(), r_rew)
            if k == 0:
                assert r_done[hit].all() and ((g_flags[hit] & N.EV_COLLIDING) != 0).all() and not r_done[far].any()
                # the newest frame agrees everywhere; the OLDER frame of a moved env is whatever its columns held (the oracle's
                # deque still holds the frame from before the move)
                np.testing.assert_allclose(obs.cpu().numpy()[:, F:], r_obs[:, F:], atol=1e-9, rtol=0)
            else:
                np.testing.assert_allclose(obs.cpu().numpy(), r_obs, atol=1e-9, rtol=0)
    vec.close()


def test_config4_curriculum_maps():
    """BASELINE configs[3] in full: 4 ships AND curriculum maps — a lesson change installs the next river width's bank
    (wider banks: traffic ship 1 starts deeper inside the left one) and resets; parity on the new bank."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    from ship_sim_gym_amd.curriculum import CurriculumMaps
    n = 1024
    vec = ShipVecEnv(n, n_maps=16, n_ships=4)
    cm = CurriculumMaps(vec, widths=(0.5, 0.8), conditions=(0.0,), repeat_condition=0, n_maps=16)
    vec.reset_tensor()
    acts = vec.random_actions(7, 0, 130)
    for k in range(12):                                           # lesson 0 runs long enough for spaces to come to rest
        vec.step_tensor(acts[k])
    assert ((vec.field(N.F_DYN_FLAGS) & 4) != 0).double().mean() > 0.3
    assert cm.progress(1.0) is not None and cm.width_frac == 0.8
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts_h = acts.cpu().numpy()
    moved3 = False
    r_done = np.zeros(n, dtype=np.uint8)
    for k in range(12, 130):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=True, n_threads=8)
        np.testing.assert_array_equal(done.cpu().numpy(), r_done, err_msg="done differs at step %d" % k)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_rew)
        assert float(np.max(np.abs(obs.cpu().numpy() - r_obs))) <= 1e-9
        if k % 10 == 0:
            _compare_dyn(N, vec, ob, just_reset=r_done, atol=1e-8)
            moved3 |= bool((_dyn_state(N, vec)[0][:, 0, 0] > 170.0).any())
    assert moved3                                                 # ship 1 is pushed out of a left bank wider than lesson 0's 150
    vec.close()


@pytest.mark.parametrize("hist", [1, 3])
def test_config4_history_sizes(hist):
    """Config 4 with EnvConfig.HISTORY_SIZE other than 2 (per-step launches + the frame-shift kernel for H > 2)."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import run_pair
    from ship_sim_gym_amd.config import EnvConfig

    class E(EnvConfig):
        HISTORY_SIZE = hist
    vec = ShipVecEnv(300, env_config=E, n_maps=8, n_ships=4)
    assert vec.observation_space.shape == (16 * hist,)
    err, n_done = run_pair(O, N, vec, K=80)
    assert err <= 1e-9 and n_done > 30
    vec.close()


def test_config4_single_env_facade_fresh_mode():
    """The reference's usage pattern for config 4 — `env = ShipEnv(...); env.reset(); env.game.add_default_traffic()` —
    on the facade (fresh map mode, one brand-new world per reset from the global RNG streams) against an oracle World."""
    import random
    torch, O, N, ShipVecEnv = _mods()
    from ship_sim_gym_amd.ship_env import ShipEnv
    from ship_sim_gym_amd import worldgen
    for seed in (3, 4):
        random.seed(seed); np.random.seed(seed)
        env = ShipEnv(n_ships=4)
        o = env.reset()
        env.game.add_default_traffic()                           # a no-op here: every reset of an n_ships=4 env adds them
        random.seed(seed); np.random.seed(seed)
        worldgen.generate_world((600, 600))                      # construction consumes one world (game.py:58)
        _, polys, goals = worldgen.generate_world((600, 600))
        w = O.World(O.default_config(n_traffic=3))
        np.testing.assert_array_equal(o, w.reset(polys[0], polys[1], goals))
        assert [(s.x, s.y) for s in env.game.ships] == [(100.0, 200.0), (300.0, 200.0), (400.0, 350.0)]
        rng = np.random.RandomState(seed + 100)
        for t in range(300):
            a = int(rng.randint(3))
            o, r, d, info = env.step(a)
            ro, rr, rd = w.step(a)
            assert r == rr and d == rd
            np.testing.assert_allclose(o, ro, rtol=0, atol=1e-9)
            pd = w.peek_dyn()
            got = np.array([(s.x, s.y, s.angle) for s in env.game.ships])
            np.testing.assert_allclose(got, pd["traffic"][:, :3], rtol=0, atol=1e-8)
            if d:
                break
        g = env.game.goals                                       # moving goal bodies are reported from body.position
        assert len(g) == bin(int(w.peek()["alive_mask"])).count("1")
        env.close()
    with pytest.raises(N.ShipSimError):
        e1 = ShipEnv()
        try:
            e1.game.add_default_traffic()                        # a 1-ship env cannot grow traffic after the fact
        finally:
            e1.close()


def test_config4_shard_equivalence():
    """SURVEY §8e for config 4: stepping N envs on one handle == the same envs split over two handles (per-env action
    stream and map assignment are keyed by the GLOBAL env id), bit for bit, bodies included."""
    torch, O, N, ShipVecEnv = _mods()
    n, K = 640, 60
    whole = ShipVecEnv(n, n_maps=16, n_ships=4)
    a = ShipVecEnv(256, n_maps=16, n_ships=4, env_id_base=0)
    b = ShipVecEnv(n - 256, n_maps=16, n_ships=4, env_id_base=256)
    ow = whole.reset_tensor().clone()
    assert torch.equal(ow[:256], a.reset_tensor()) and torch.equal(ow[256:], b.reset_tensor())
    aw, aa, ab = whole.random_actions(5, 0, K), a.random_actions(5, 0, K), b.random_actions(5, 0, K)
    assert torch.equal(aw[:, :256], aa) and torch.equal(aw[:, 256:], ab)
    for k in range(K):
        o, r, d, f = whole.step_tensor(aw[k])
        o1, r1, d1, f1 = a.step_tensor(aa[k])
        o2, r2, d2, f2 = b.step_tensor(ab[k])
        assert torch.equal(o[:256], o1) and torch.equal(o[256:], o2)
        assert torch.equal(r[:256], r1) and torch.equal(r[256:], r2) and torch.equal(d[:256], d1) and torch.equal(d[256:], d2)
    tw = whole.field(N.F_TRAFFIC)
    assert torch.equal(tw[:, :256], a.field(N.F_TRAFFIC)) and torch.equal(tw[:, 256:], b.field(N.F_TRAFFIC))
    for v in (whole, a, b):
        v.close()


def test_config4_goal_override_moves_the_goal_bodies():
    """ShipEnv.reset(goals=...) with n_ships=4: the dynamic goal BODIES (not only the map record and the reset
    observation) must sit on the overridden centres, so the physics and the nearest-goal report agree."""
    import random
    from ship_gym.ship_env import ShipEnv
    random.seed(5); np.random.seed(5)
    e = ShipEnv(n_ships=4)
    goals = [[300, 100], [300, 170], [300, 240], [300, 310], [300, 380]]
    o = e.reset(spawn_point=(295, 30), goals=goals)
    assert (o[20], o[21]) == (300, 100)
    assert [(g.x, g.y) for g in e.game.goals] == [tuple(map(float, g)) for g in goals]   # body.position of each goal
    got = 0
    for _ in range(14):
        o, r, d, _ = e.step(0)
        got += int(r == 1.0)
        c = e.game.closest_goal()
        if c is None or d:
            break
        assert (c.x, c.y) == (o[20], o[21])
    assert got >= 2      # the ship sails up x = 295..315 and collects the overridden goals, not the generated ones
    e.close()


@pytest.mark.parametrize("n,n_maps,K", [(4096, 64, 140), (1000, 3, 90), (3000, 96, 110)])  # (96 records: the per-lane-planes variant of the full step, memo on)
def test_config4_memo_is_invisible(n, n_maps, K):
    """The memo of the full cpSpaceStep (SSG_F_DYN_MEMO_STATS: envs that share a bank record replay the same body states, so
    a state's step is computed once and looked up afterwards) must change NOTHING: every output of every step, and every body /
    arbiter / flag column at the end, bit for bit equal to a handle that computes every step (SSG_FLAG_DYN_MEMO_OFF) — and
    equal to the oracle.  The table really answers (most look-ups hit once it is warm)."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    a = ShipVecEnv(n, n_beams=10, n_maps=n_maps, n_ships=4)
    b = ShipVecEnv(n, n_beams=10, n_maps=n_maps, n_ships=4, dyn_memo=False)
    ob = O.Batch(n, oracle_cfg(O, a), a.bank_polys, a.bank_goals, map_ids=np.arange(n) % a.n_maps)
    oa = a.reset_tensor().clone()
    assert torch.equal(oa, b.reset_tensor())
    np.testing.assert_array_equal(oa.cpu().numpy(), ob.reset())
    acts = a.random_actions(31, 0, K)
    acts_h = acts.cpu().numpy()
    worst = 0.0
    for k in range(K):
        o1, r1, d1, f1 = a.step_tensor(acts[k])
        o2, r2, d2, f2 = b.step_tensor(acts[k])
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(f1, f2), "step %d" % k
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=True, n_threads=8)
        np.testing.assert_array_equal(d1.cpu().numpy(), r_done); np.testing.assert_array_equal(r1.cpu().numpy(), r_rew)
        worst = max(worst, float(np.max(np.abs(o1.cpu().numpy() - r_obs))))
    assert worst <= 1e-9
    for fid in (N.F_X, N.F_Y, N.F_ANGLE, N.F_LIDAR, N.F_STEP_COUNT, N.F_MAP_ID, N.F_GOAL_MASK, N.F_TRAFFIC, N.F_GOAL_BODIES, N.F_DYN_FLAGS):
        assert torch.equal(a.field(fid), b.field(fid)), fid
    # the arbiter columns, the live masks and the row-major shadow too: the whole dyn region up to the queue / memo areas
    import ctypes as C
    off = {}
    for name, fid in (("traffic", N.F_TRAFFIC), ("flags", N.F_DYN_FLAGS)):
        o_, es, nc, st = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
        N.check(N.lib().ssg_state_field(a._h, fid, C.byref(o_), C.byref(es), C.byref(nc), C.byref(st)), a._h, "field")
        off[name] = o_.value
    assert torch.equal(a.state[off["traffic"]: off["flags"]], b.state[off["traffic"]: off["flags"]])   # body, arbiter, live, u32 columns
    _compare_dyn(N, a, ob, just_reset=r_done)
    sa, sb = a.dyn_memo_stats(), b.dyn_memo_stats()
    assert sb["hits"] == 0 and sb["stored"] == 0
    assert sa["hits"] > sa["computed"] and sa["stored"] > 0, sa   # (the first ~9 steps after the full reset all compute: every env is new)
    print("memo: %r" % sa)
    a.close(); b.close()


@pytest.mark.parametrize("n,n_maps,memo", [(2048, 16, True), (1500, 96, True), (1024, 1, False)])
def test_config4_masked_reset_joins_the_queue(n, n_maps, memo):
    """The RLlib flow (train/rllib/ppo.py:21-44 over ShipEnv.reset, ship_env.py:171-184): no in-kernel auto-reset, the caller
    resets the done envs with ONE masked ssg_reset after every step.  Those envs join the queue the step kernel left for the
    next full cpSpaceStep (no rebuild of the queue from the per-env flags: ssg_debug_dyn_counters), and every step — terminal
    observations, reset observations, rewards, done flags, the bodies — matches the oracle stepped the same way.  A SECOND
    masked reset between two steps (every tenth step here) falls back to the rebuild and stays exact."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    K = 120
    vec = ShipVecEnv(n, n_beams=10, n_maps=n_maps, n_ships=4, auto_reset=False, dyn_memo=memo)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts = vec.random_actions(555, 0, K)
    acts_h = acts.cpu().numpy()
    n_done, worst, doubles = 0, 0.0, 0

    def masked_reset(done_t, done_h):
        ids = vec.field(N.F_MAP_ID).clone()
        ids = torch.where(done_t != 0, (ids + 1) % vec.n_maps, ids).to(torch.int32).contiguous()
        o2 = vec.reset_tensor(mask=done_t.clone(), map_ids=ids)
        r2 = ob.auto_reset_done()
        np.testing.assert_array_equal(o2.cpu().numpy()[done_h != 0], r2[done_h != 0])

    for k in range(K):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=False, n_threads=8)
        np.testing.assert_array_equal(done.cpu().numpy(), r_done, err_msg="done differs at step %d" % k)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_rew)
        worst = max(worst, float(np.max(np.abs(obs.cpu().numpy() - r_obs))))       # terminal observations of the done envs
        if r_done.any():
            masked_reset(done, r_done)
            n_done += int(r_done.sum())
            if k % 10 == 3:                                                        # ... and once more: onto the record after that
                masked_reset(done, r_done)
                doubles += 1
        if k in (1, 5, 40, K - 1):
            _compare_dyn(N, vec, ob)                                               # (a masked reset rebuilds the bodies at once)
    assert worst <= 1e-9 and n_done > n // 2 and doubles >= 3
    steps, rebuilds = vec.dyn_counters()
    if n_maps <= 64:
        assert steps == K and rebuilds == 1 + doubles, (steps, rebuilds, doubles)  # the first step, and the steps after a double reset
    else:  # the per-lane-planes kernel cannot tell a stale queue entry from the new one: every reset rebuilds the queue
        assert steps == K and rebuilds >= K // 2, (steps, rebuilds)
    vec.close()


@pytest.mark.parametrize("mode", ["bank96", "fresh_device"])
def test_config4_masked_reset_of_queued_envs_never_steps_an_env_twice(mode):
    """Handles whose full cpSpaceStep runs the per-lane-planes kernel (banks of more than 64 records, map_ring) at a size where
    only a part of the queue's waves is resident at once: a masked host reset moves QUEUED envs to another record between two
    steps.  An env must still be stepped exactly once per step — the queue is rebuilt instead of appended to — so the run equals
    one whose queue is rebuilt from the per-env flags before every step (ssg_dyn_invalidate with an all-zero mask changes no
    state, only forces the rebuild), bit for bit.  (The oracle comparison of this flow is the test above, at 1 500 envs.)"""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    n, K = 32768, 24
    kw = dict(n_maps=96) if mode == "bank96" else dict(map_mode="fresh_device", ring=8, map_seed=77)
    va = ShipVecEnv(n, n_beams=10, n_ships=4, auto_reset=(mode == "fresh_device"), **kw)
    vb = ShipVecEnv(n, n_beams=10, n_ships=4, auto_reset=(mode == "fresh_device"), **kw)
    assert torch.equal(va.reset_tensor(), vb.reset_tensor())
    acts = va.random_actions(909, 0, K)
    zero = torch.zeros(n, dtype=torch.uint8, device=va.device)
    gen = torch.Generator(device="cpu").manual_seed(5)
    for k in range(K):
        oa = [t.clone() for t in va.step_tensor(acts[k])]
        vb.wake_dynamics(mask=zero)                       # b: the queue is rebuilt from the flags before every step
        ob_ = [t.clone() for t in vb.step_tensor(acts[k])]
        for x, y in zip(oa, ob_):
            assert torch.equal(x, y), "step %d" % k
        # reset a quarter of the envs — queued or not, done or not — onto their next record
        mask = (torch.rand(n, generator=gen) < 0.25).to(torch.uint8).to(va.device)
        if mode == "bank96":
            ids = va.field(N.F_MAP_ID).clone()
            ids = torch.where(mask != 0, (ids + 1) % va.n_maps, ids).to(torch.int32).contiguous()
            ra, rb = va.reset_tensor(mask=mask, map_ids=ids).clone(), vb.reset_tensor(mask=mask.clone(), map_ids=ids.clone()).clone()
        else:
            ra, rb = va.reset_tensor(mask=mask).clone(), vb.reset_tensor(mask=mask.clone()).clone()
        assert torch.equal(ra, rb)
    for fid in (N.F_X, N.F_Y, N.F_ANGLE, N.F_TRAFFIC, N.F_GOAL_BODIES, N.F_MAP_ID, N.F_GOAL_MASK):
        assert torch.equal(va.field(fid), vb.field(fid)), fid
    va.close(); vb.close()


@pytest.mark.parametrize("n_ships,nb", [(4, 10), (1, 8)])
def test_terminal_obs_side_buffer_is_the_rllib_flow_without_a_reset_launch(n_ships, nb):
    """ssg_set_terminal_obs (train/rllib/ppo.py:21-44 over ship_env.py:136-156,171-184): with in-kernel auto-reset, ONE step hands out
    both observations of an episode's end — the reset observation in the env's row of `obs`, the terminal observation in its row of
    `term_obs` — against the oracle stepped WITHOUT auto-reset (terminal rows) and then reset (reset rows).  Single steps and fused
    overwrite-mode rollouts; rows of envs that are not done are never touched."""
    torch, O, N, ShipVecEnv = _mods()
    from helpers import oracle_cfg
    n, K = 1200, 150
    vec = ShipVecEnv(n, n_beams=nb, n_maps=16, n_ships=n_ships)
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    term = vec.enable_terminal_obs()
    term.fill_(-7.0)
    acts = vec.random_actions(808, 0, K)
    acts_h = acts.cpu().numpy()
    n_done, worst = 0, 0.0
    for k in range(K):
        if n_ships == 1 and k % 3 == 2:   # a fused overwrite-mode rollout of one step is the same launch path as K > 1
            obs, rew, done, flags = vec.rollout_tensor(acts[k:k + 1])
        else:
            obs, rew, done, flags = vec.step_tensor(acts[k])
        r_obs, r_rew, r_done = ob.step(acts_h[k], auto_reset=False, n_threads=8)      # terminal observations on done rows
        d = r_done != 0
        np.testing.assert_array_equal(done.cpu().numpy(), r_done)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_rew)
        o_h, t_h = obs.cpu().numpy(), term.cpu().numpy()
        worst = max(worst, float(np.max(np.abs(o_h[~d] - r_obs[~d]))) if (~d).any() else 0.0)
        if d.any():
            worst = max(worst, float(np.max(np.abs(t_h[d] - r_obs[d]))))                # the episode's last observation
            r2 = ob.auto_reset_done()
            np.testing.assert_array_equal(o_h[d], r2[d])                              # ... and the next episode's first
            n_done += int(d.sum())
        assert np.all(t_h[~d] == -7.0)
        term.fill_(-7.0)
    assert n_done > n // 2 and worst <= 1e-9, (n_done, worst)
    # fused steps (1 ship): the rows of the LAST step's done envs, and of envs done at earlier steps of the launch (their rows stay
    # until the env ends another episode)
    if n_ships == 1:
        a2 = vec.random_actions(809, 0, 40)
        vec.rollout_tensor(a2)
        assert (term != -7.0).any()
    vec.enable_terminal_obs(False)
    assert vec.term_obs is None
    vec.close()


def test_config4_snapshot_restore_of_the_state_blob():
    """The caller owns the state blob (include/shipsim.h): a copy taken between two steps and copied back later, followed by
    ssg_dyn_invalidate (the queue of the next full cpSpaceStep lives in the blob, its live counter set is named by the handle),
    replays the same steps bit for bit — bodies, arbiters, memo table and all."""
    torch, O, N, ShipVecEnv = _mods()
    n = 1536
    vec = ShipVecEnv(n, n_beams=10, n_maps=16, n_ships=4)
    vec.reset_tensor()
    acts = vec.random_actions(2718, 0, 31 + 45)
    for k in range(31):                                            # (an odd count: the queue's other counter set is the live one)
        vec.step_tensor(acts[k])
    snap = vec.state.clone()
    first = [tuple(t.clone() for t in vec.step_tensor(acts[31 + k])) for k in range(45)]
    end1 = vec.state.clone()
    vec.state.copy_(snap)
    vec.wake_dynamics()
    for k in range(45):
        o, r, d, f = vec.step_tensor(acts[31 + k])
        assert torch.equal(o, first[k][0]) and torch.equal(r, first[k][1]) and torch.equal(d, first[k][2]) and torch.equal(f, first[k][3]), k
    got = {fid: vec.field(fid).clone() for fid in (N.F_X, N.F_Y, N.F_ANGLE, N.F_LIDAR, N.F_STEP_COUNT, N.F_MAP_ID, N.F_GOAL_MASK, N.F_TRAFFIC, N.F_GOAL_BODIES)}
    vec.state.copy_(end1)
    for fid, a in got.items():
        assert torch.equal(a, vec.field(fid)), fid
    vec.close()


def test_config4_memo_generations_wrap_around():
    """Every bank change starts a new GENERATION of the memo tables (entries of the old bank count as empty, no memset); the
    generation counter has 255 values and a wrap zeroes the tables.  Re-installing the bank 300 times — past the wrap — with steps in
    between leaves memo on and memo off bit for bit equal, and both equal to a handle that never re-installed anything (the
    bank's contents did not change: only rest bits and table generations did)."""
    torch, O, N, ShipVecEnv = _mods()
    n = 384
    a = ShipVecEnv(n, n_beams=8, n_maps=4, n_ships=4)
    b = ShipVecEnv(n, n_beams=8, n_maps=4, n_ships=4, dyn_memo=False)
    c = ShipVecEnv(n, n_beams=8, n_maps=4, n_ships=4)
    for v in (a, b, c):
        v.reset_tensor()
    acts = a.random_actions(17, 0, 300)
    for k in range(300):
        a.set_bank(a.bank); b.set_bank(b.bank)
        oa, ra, da, fa = a.step_tensor(acts[k])
        ob_, rb, db, fb = b.step_tensor(acts[k])
        oc, rc, dc, fc = c.step_tensor(acts[k])
        assert torch.equal(oa, ob_) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(fa, fb), k
        assert torch.equal(oa, oc) and torch.equal(da, dc), k
    for fid in (N.F_TRAFFIC, N.F_GOAL_BODIES, N.F_X, N.F_GOAL_MASK):
        assert torch.equal(a.field(fid), b.field(fid)) and torch.equal(a.field(fid), c.field(fid)), fid
    sa, sc = a.dyn_memo_stats(), c.dyn_memo_stats()
    assert sc["hits"] > sc["computed"] and sa["stored"] > 250          # every generation starts empty and learns again
    for v in (a, b, c):
        v.close()


def test_config4_refuses_hip_graph_capture():
    """A config-4 step's kernel arguments carry host state that changes from call to call (which counter set holds the queue, the
    launch number the memo stamps its entries with): captured in a HIP graph, every replay would repeat the captured step's — so
    ssg_step refuses a capturing stream.  A 1-ship handle on a shared bank launches with constant arguments: capture + replay
    equals plain launches."""
    torch, O, N, ShipVecEnv = _mods()
    from ship_sim_gym_amd._native import ShipSimError
    n = 512
    for ships in (1, 4):
        a = ShipVecEnv(n, n_beams=10, n_ships=ships); b = ShipVecEnv(n, n_beams=10, n_ships=ships)
        a.reset_tensor(); b.reset_tensor()
        acts = a.random_actions(5, 0, 40)
        static_act = torch.zeros(n, dtype=torch.int32, device="cuda")
        static_act.copy_(acts[0]); a.step_tensor(static_act); b.step_tensor(acts[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        refused = False
        with torch.cuda.stream(s):
            static_act.copy_(acts[1])
            g.capture_begin()
            try:
                a.step_tensor(static_act)
            except ShipSimError as ex:
                refused = True
                assert "capturing" in str(ex)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        assert refused == (ships == 4)
        for k in range(1, 40):
            static_act.copy_(acts[k])
            if ships == 1:
                g.replay()
            else:
                a.step_tensor(static_act)  # (the refused capture left the handle as it was: plain launches go on)
            ob, rb, db, fb = b.step_tensor(acts[k])
            assert torch.equal(a.obs, ob) and torch.equal(a.reward, rb) and torch.equal(a.done, db), (ships, k)
        a.close(); b.close()
