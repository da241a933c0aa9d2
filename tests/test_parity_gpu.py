"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical banks and action streams.

Bar (BASELINE.json north_star): observations within 1e-5 absolute, done / collision / reward values bit-exact.
The oracle itself is "parity unpinned" at the pymunk boundary (oracle/ssg_oracle.h).
"""
import numpy as np
import pytest

from helpers import OracleSample, oracle_cfg, run_pair

pytestmark = pytest.mark.gpu

ATOL = 1e-5  # north_star tolerance


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch


def _vec(n, **kw):
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    return ShipVecEnv(n, **kw)


def test_default_config_parity(torch_cuda, oracle, native):
    """BASELINE configs[1] shape at reduced N: default map/bounds/SPEED, 10 beams, history 2."""
    vec = _vec(1024, n_maps=64)
    err, n_done = run_pair(oracle, native, vec, K=400)
    assert n_done > 100  # the run exercised collisions / out-of-bounds / auto-reset
    assert err <= ATOL
    print("max |obs - oracle| = %.3e over %d episode ends" % (err, n_done))


def test_c3_eight_beams_parity(torch_cuda, oracle, native):
    """BASELINE configs[2] shape at reduced N: 8-beam lidar."""
    vec = _vec(2048, n_maps=64, n_beams=8)
    err, n_done = run_pair(oracle, native, vec, K=300)
    assert n_done > 100 and err <= ATOL


def test_training_config_parity(torch_cuda, oracle, native):
    """train/stable_baselines/ppo.py:65-69: SPEED 30, BOUNDS 1000^2 (dt = 3.0000000000000004)."""
    from ship_sim_gym_amd.config import GameConfig

    class GC(GameConfig):
        SPEED = 30
        BOUNDS = (1000, 1000)

    vec = _vec(512, game_config=GC, n_maps=16)
    err, n_done = run_pair(oracle, native, vec, K=200)
    assert n_done > 500 and err <= ATOL  # episodes are ~6 steps long at this speed


def test_ragged_sizes_and_bank_in_global(torch_cuda, oracle, native):
    """N not a multiple of the wavefront / workgroup; LDS-staged bank and global-gather bank must agree bitwise."""
    for n in (1, 63, 65, 257, 1000):
        a = _vec(n, n_maps=5)
        b = _vec(n, n_maps=5, bank_in_global=True)
        a.reset_tensor(); b.reset_tensor()
        acts = a.random_actions(7, 0, 120)
        for k in range(120):
            oa = [t.clone() for t in a.step_tensor(acts[k])]
            ob = b.step_tensor(acts[k])
            for x, y in zip(oa, ob):
                assert torch_cuda.equal(x, y)
        err, _ = run_pair(oracle, native, _vec(n, n_maps=5), K=60)
        assert err <= ATOL


def test_exact_and_one_division_lidar_agree(torch_cuda, oracle, native):
    """SSG_FLAG_EXACT_LIDAR (cpPolyShapeSegmentQuery plane by plane) vs the default one-division-per-beam evaluation:
    same hit/miss decisions and hit points, so identical observations; both against the oracle."""
    torch = torch_cuda
    a = _vec(8192, n_maps=64, n_beams=10)
    b = _vec(8192, n_maps=64, n_beams=10, exact_lidar=True)
    a.reset_tensor(); b.reset_tensor()
    acts = a.random_actions(11, 0, 300)
    worst = 0.0
    for k in range(300):
        oa, ra, da, fa = a.step_tensor(acts[k])
        ob, rb, db, fb = b.step_tensor(acts[k])
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(fa, fb)
        worst = max(worst, float((oa - ob).abs().max()))
    assert worst <= 1e-9, worst  # equal up to the vertex-grazing case (none expected in 2.4e7 rays)
    err, _ = run_pair(oracle, native, _vec(1024, n_maps=64, exact_lidar=True), K=200)
    assert err <= ATOL


def test_event_flags_bit_exact(torch_cuda, oracle, native):
    """colliding / goal_reached (ShipGame attributes, game.py:190-191,240,254) against the oracle's."""
    vec = _vec(512, n_maps=32)
    n = vec.num_envs
    ob = oracle.Batch(n, oracle_cfg(oracle, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    ob.reset(); vec.reset_tensor()
    acts = vec.random_actions(99, 0, 300)
    acts_h = acts.cpu().numpy()
    seen = {"col": 0, "goal": 0}
    for k in range(300):
        _, _, done, flags = vec.step_tensor(acts[k])
        # oracle without auto-reset first, to read its flags, then reset the done ones by stepping the batch API
        ob.step(acts_h[k], auto_reset=False)
        pk = ob.peek_all()
        f = flags.cpu().numpy()
        np.testing.assert_array_equal((f & native.EV_COLLIDING) != 0, pk[:, 9] != 0)
        np.testing.assert_array_equal((f & native.EV_GOAL_REACHED) != 0, pk[:, 10] != 0)
        seen["col"] += int((pk[:, 9] != 0).sum()); seen["goal"] += int((pk[:, 10] != 0).sum())
        d = done.cpu().numpy()
        np.testing.assert_array_equal(d, ob.done)
        # bring the oracle's done envs onto the next map like the kernel's auto-reset did
        for e in np.nonzero(d)[0]:
            m = (int(pk[e, 11]) + 1) % vec.n_maps
            w = oracle.lib().ora_world_at(ob._p, int(e))
            import ctypes as C
            oracle.lib().ora_batch_reset(w, 1, C.byref(ob.cfg), C.byref(ob.bank),
                                         np.asarray([m], dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32)), None)
    assert seen["col"] > 20 and seen["goal"] > 20


def test_state_columns_match_oracle(torch_cuda, oracle, native):
    """Hidden state too (velocities, angular velocity, step counters, goal masks), not only what obs shows."""
    vec = _vec(300, n_maps=8)
    n = vec.num_envs
    ob = oracle.Batch(n, oracle_cfg(oracle, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    ob.reset(); vec.reset_tensor()
    acts = vec.random_actions(5, 0, 150)
    acts_h = acts.cpu().numpy()
    for k in range(150):
        vec.step_tensor(acts[k]); r_obs = ob.step(acts_h[k])[0]
    pk = ob.peek_all()
    N = native
    # the sticky lidar readings kept between launches (single-step launches store them from the lidar waves: hits and
    # fresh episodes only) are the newest frame's readings
    F = r_obs.shape[1] // 2
    lid = vec.field(N.F_LIDAR).cpu().numpy()  # [n_beams, n]
    np.testing.assert_allclose(lid.T, r_obs[:, F + 6:], rtol=0, atol=ATOL)
    for fid, col in ((N.F_X, 0), (N.F_Y, 1), (N.F_VX, 2), (N.F_VY, 3), (N.F_ANGLE, 4), (N.F_W, 5), (N.F_CUM_REWARD, 12)):
        np.testing.assert_allclose(vec.field(fid).cpu().numpy(), pk[:, col], rtol=0, atol=ATOL)
    np.testing.assert_array_equal(vec.field(N.F_RUDDER).cpu().numpy(), pk[:, 6].astype(np.int32))
    np.testing.assert_array_equal(vec.field(N.F_STEP_COUNT).cpu().numpy(), pk[:, 7].astype(np.int32))
    np.testing.assert_array_equal(vec.field(N.F_MAP_ID).cpu().numpy(), pk[:, 11].astype(np.int32))
    np.testing.assert_array_equal(vec.field(N.F_GOAL_MASK).cpu().numpy() & 0x7F, pk[:, 13].astype(np.uint8))


def test_fused_rollout_equals_single_steps(torch_cuda, native):
    """ssg_rollout (K steps fused in one launch, state in registers, other roles re-reading it from L2) must leave
    exactly the state and outputs that K separate ssg_step launches leave."""
    torch = torch_cuda
    for n, nb in ((1000, 10), (65536, 8)):
        a = _vec(n, n_maps=64, n_beams=nb)
        b = _vec(n, n_maps=64, n_beams=nb)
        a.reset_tensor(); b.reset_tensor()
        acts = a.random_actions(77, 0, 150)
        for k in range(150):
            a.step_tensor(acts[k])
        b.rollout_tensor(acts)
        torch.cuda.synchronize()
        assert torch.equal(a.state, b.state)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
        assert torch.equal(a.flags, b.flags)
        assert a.stats() == b.stats()


def test_fused_trajectory_every_step_matches_oracle(torch_cuda, oracle, native):
    """The kernel the headline times, checked step by step: ssg_rollout_traj keeps the (obs, reward, done, flags) of EVERY
    step of its fused launches (what train/random.py:14-27 consumes); each of the 2 x 100 fused steps + a ragged third
    launch of 37 is compared with the oracle stepping the same envs on the same Philox actions."""
    torch = torch_cuda
    vec = _vec(2048, n_maps=64, n_beams=8)
    n, K = vec.num_envs, 237
    ob = oracle.Batch(n, oracle_cfg(oracle, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    np.testing.assert_array_equal(vec.reset_tensor().cpu().numpy(), ob.reset())
    acts = vec.random_actions(4242, 0, K)
    acts_h = acts.cpu().numpy()
    to, tr, td, tf = vec.rollout_tensor(acts, trajectory=True)
    assert tuple(to.shape) == (K, n, vec.states_history) and tuple(tr.shape) == tuple(td.shape) == tuple(tf.shape) == (K, n)
    g_obs, g_rew, g_done, g_flags = to.cpu().numpy(), tr.cpu().numpy(), td.cpu().numpy(), tf.cpu().numpy()
    worst, n_done = 0.0, 0
    for k in range(K):
        ob.step(acts_h[k], auto_reset=False)
        pk = ob.peek_all()  # the oracle's colliding / goal_reached attributes of this step
        np.testing.assert_array_equal((g_flags[k] & native.EV_COLLIDING) != 0, pk[:, 9] != 0, err_msg="colliding, fused step %d" % k)
        np.testing.assert_array_equal((g_flags[k] & native.EV_GOAL_REACHED) != 0, pk[:, 10] != 0, err_msg="goal, fused step %d" % k)
        np.testing.assert_array_equal(g_done[k], ob.done, err_msg="done flags differ at fused step %d" % k)
        np.testing.assert_array_equal(g_rew[k], ob.reward, err_msg="rewards differ at fused step %d" % k)
        r_obs = ob.auto_reset_done()  # VecEnv semantics: the done envs move to the next bank record; rows = reset obs
        err = float(np.max(np.abs(g_obs[k] - r_obs)))
        assert err <= ATOL, "obs differ by %g at fused step %d" % (err, k)
        worst = max(worst, err)
        n_done += int(ob.done.sum())
    assert n_done > 300
    print("fused trajectory: max |obs - oracle| = %.3e over %d steps, %d episode ends" % (worst, K, n_done))
    # the trajectory's last step is what the overwrite mode leaves, and the state columns agree bitwise
    b = _vec(2048, n_maps=64, n_beams=8)
    b.reset_tensor()
    b.rollout_tensor(acts)
    torch.cuda.synchronize()
    assert torch.equal(b.obs, to[K - 1]) and torch.equal(b.reward, tr[K - 1]) and torch.equal(b.done, td[K - 1])
    assert torch.equal(b.flags, tf[K - 1]) and torch.equal(b.state, vec.state)


def test_trajectory_mode_other_paths_and_strides(torch_cuda, native):
    """ssg_rollout_traj on the paths that launch once per step (history 3, config 4), with a caller-provided buffer that
    is longer than K, and through the raw ABI with a stride wider than n_envs: every slot equals the single-step outputs."""
    torch = torch_cuda
    import ctypes as C
    from ship_sim_gym_amd.config import EnvConfig

    class E3(EnvConfig):
        HISTORY_SIZE = 3

    for kw in (dict(n_beams=8), dict(env_config=E3), dict(n_ships=4)):
        a, b = _vec(700, n_maps=16, **kw), _vec(700, n_maps=16, **kw)
        a.reset_tensor(); b.reset_tensor()
        K = 130
        acts = a.random_actions(31, 0, K)
        n, D = a.num_envs, a.states_history
        out = (torch.full((K + 3, n, D), 7.0, dtype=torch.float64, device=a.device), torch.full((K + 3, n), 7.0, dtype=torch.float64, device=a.device),
               torch.full((K + 3, n), 7, dtype=torch.uint8, device=a.device), torch.full((K + 3, n), 7, dtype=torch.uint8, device=a.device))
        to, tr, td, tf = b.rollout_tensor(acts, trajectory=True, out=out)
        for k in range(K):
            o, r, d, f = a.step_tensor(acts[k])
            assert torch.equal(o, to[k]) and torch.equal(r, tr[k]) and torch.equal(d, td[k]) and torch.equal(f, tf[k]), (kw, k)
        if "n_ships" in kw:  # (config 4's work queue is filled in scheduling order: compare the fields, not the raw blob)
            for fid in (native.F_X, native.F_W, native.F_LIDAR, native.F_RUDDER, native.F_STEP_COUNT, native.F_MAP_ID, native.F_GOAL_MASK,
                        native.F_STATS, native.F_TRAFFIC, native.F_GOAL_BODIES, native.F_DYN_FLAGS, native.F_EPISODES):
                assert torch.equal(a.field(fid), b.field(fid)), fid
        else:
            assert torch.equal(a.state, b.state)
        assert all(bool((t[K:] == 7).all()) for t in out)  # nothing written past step K-1
    # raw ABI, stride 1000 > n_envs 700: this handle's shard of a wider [K][1000] layout; the gap columns stay untouched
    a, b = _vec(700, n_maps=16, n_beams=8), _vec(700, n_maps=16, n_beams=8)
    a.reset_tensor(); b.reset_tensor()
    K, S, D = 120, 1000, a.states_history
    acts = a.random_actions(32, 0, K)
    to = torch.full((K, S, D), 9.0, dtype=torch.float64, device=a.device)
    tr = torch.full((K, S), 9.0, dtype=torch.float64, device=a.device)
    td = torch.full((K, S), 9, dtype=torch.uint8, device=a.device)
    vp = lambda t: C.c_void_p(t.data_ptr())
    L = native.lib()
    native.check(L.ssg_rollout_traj(b._h, vp(acts), K, vp(to), vp(tr), vp(td), None, S, b._stream()), b._h, "traj")
    for k in range(K):
        o, r, d, _ = a.step_tensor(acts[k])
        assert torch.equal(o, to[k, :700]) and torch.equal(r, tr[k, :700]) and torch.equal(d, td[k, :700])
    assert bool((to[:, 700:] == 9).all()) and bool((tr[:, 700:] == 9).all()) and bool((td[:, 700:] == 9).all())
    # overlapping slots are refused
    rc = L.ssg_rollout_traj(b._h, vp(acts), K, vp(to), vp(tr), vp(td), None, 699, b._stream())
    assert rc == -1 and b"step_stride_envs" in L.ssg_last_error(b._h)


def test_trajectory_buffer_set_reused(torch_cuda, native):
    """A caller's buffer set handed to rollout_tensor again (bench.py rotates a few): the second and later calls take the
    lean path (cached pointers, one ctypes call) — same results as a twin env stepped one launch at a time, for different
    K, including K longer than the buffers (falls back to the checked path, which refuses)."""
    torch = torch_cuda
    a, b = _vec(1500, n_maps=16, n_beams=8), _vec(1500, n_maps=16, n_beams=8)
    a.reset_tensor(); b.reset_tensor()
    n, D, cap = a.num_envs, a.states_history, 40
    out = (torch.empty((cap, n, D), dtype=torch.float64, device=a.device), torch.empty((cap, n), dtype=torch.float64, device=a.device),
           torch.empty((cap, n), dtype=torch.uint8, device=a.device), torch.empty((cap, n), dtype=torch.uint8, device=a.device))
    acts = a.random_actions(77, 0, 200)
    k0 = 0
    for i, K in enumerate((40, 20, 40, 7, 33)):
        to, tr, td, tf = b.rollout_tensor(acts[k0: k0 + K], trajectory=True, out=out)
        assert to.shape[0] == K and to.data_ptr() == out[0].data_ptr()
        if i >= 1:
            plan = b._traj_plans[out[0].data_ptr()]
            assert not any(isinstance(v, torch.Tensor) for v in plan)  # pointers and shapes only: a caller's `del` frees its buffers
        for k in range(K):
            o, r, d, f = a.step_tensor(acts[k0 + k])
            assert torch.equal(o, to[k]) and torch.equal(r, tr[k]) and torch.equal(d, td[k]) and torch.equal(f, tf[k]), (i, k)
        k0 += K
    assert torch.equal(a.state, b.state)
    with pytest.raises(AssertionError):
        b.rollout_tensor(acts[:41], trajectory=True, out=out)
    # a different reward buffer beside the known obs buffer is not mistaken for the cached set
    out2 = (out[0], torch.full((cap, n), 5.0, dtype=torch.float64, device=a.device), out[2], out[3])
    keep = out[1].clone()
    b.rollout_tensor(acts[:3], trajectory=True, out=out2)
    assert torch.equal(out[1], keep) and not bool((out2[1][:3] == 5.0).any())
    b.clear_traj_cache()
    assert "_traj_plans" not in b.__dict__
    b.close()
    assert "_traj_plans" not in b.__dict__


def test_shard_equivalence(torch_cuda, native):
    """N envs on one handle == the same envs split over two handles (env_id_base keyed), bitwise."""
    n = 1024
    full = _vec(n, n_maps=16)
    lo = _vec(n // 2, n_maps=16, env_id_base=0)
    hi = _vec(n // 2, n_maps=16, env_id_base=n // 2)
    full.reset_tensor(); lo.reset_tensor(); hi.reset_tensor()
    af, al, ah = full.random_actions(3, 0, 200), lo.random_actions(3, 0, 200), hi.random_actions(3, 0, 200)
    assert torch_cuda.equal(af[:, :n // 2], al) and torch_cuda.equal(af[:, n // 2:], ah)
    for k in range(200):
        of, rf, df, _ = full.step_tensor(af[k])
        ol, rl, dl, _ = lo.step_tensor(al[k])
        oh, rh, dh, _ = hi.step_tensor(ah[k])
        assert torch_cuda.equal(of[:n // 2], ol) and torch_cuda.equal(of[n // 2:], oh)
        assert torch_cuda.equal(df[:n // 2], dl) and torch_cuda.equal(df[n // 2:], dh)
        assert torch_cuda.equal(rf[:n // 2], rl) and torch_cuda.equal(rf[n // 2:], rh)


def test_full_size_properties(torch_cuda, native):
    """BASELINE configs[2] at full size (65 536 envs, 8 beams): size-independent properties instead of the oracle."""
    torch = torch_cuda
    n, K = 65536, 200
    vec = _vec(n, n_maps=64, n_beams=8)
    ref = _vec(n, n_maps=64, n_beams=8, bank_in_global=True)
    F = vec.n_states
    obs0 = vec.reset_tensor().clone(); ref.reset_tensor()
    assert torch.all(obs0[:, :F] == -1)
    acts = vec.random_actions(2024, 0, K)
    prev = obs0
    ep_done = 0
    for k in range(K):
        obs, rew, done, flags = vec.step_tensor(acts[k])
        o2, r2, d2, f2 = ref.step_tensor(acts[k])
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2) and torch.equal(flags, f2)
        cont = done == 0
        # history is oldest-first: this step's old frame is last step's new frame (ship_env.py:113,181)
        assert torch.equal(obs[cont][:, :F], prev[cont][:, F:])
        # a reset env reports the reset observation: (-1)*F then the spawn frame
        rs = done != 0
        if rs.any():
            assert torch.all(obs[rs][:, :F] == -1)
            assert torch.all(obs[rs][:, F] == vec.cfg.spawn_x) and torch.all(obs[rs][:, F + 1] == vec.cfg.spawn_y)
        # frame agrees with the state columns; rudder stays on the 5-value lattice
        assert torch.equal(obs[:, F], vec.field(native.F_X)) and torch.equal(obs[:, F + 1], vec.field(native.F_Y))
        rud = vec.field(native.F_RUDDER)
        assert int(rud.abs().max()) <= 10 and torch.all(rud % 5 == 0)
        # rewards take only the reference's three values (ship_env.py:62-77)
        assert torch.all((rew == 1.0) | (rew == -1.0) | (rew == -0.01))
        assert torch.equal(rew == 1.0, (flags & native.EV_GOAL_REACHED) != 0)
        # lidar readings are -1 (never hit yet) or within (0, lidar_dist + eps]
        lid = obs[:, F + 6:]
        assert torch.all((lid == -1) | ((lid >= 0) & (lid <= 100.0 + 1e-9)))
        prev = obs.clone()
        ep_done += int(done.sum())
    st = vec.stats()
    assert st["episodes"] == ep_done and ep_done > n // 2


def test_full_size_fused_trajectory_properties(torch_cuda, oracle, native):
    """The benchmark's own launch shape — BASELINE configs[2] at full size, 100 fused steps per launch, trajectory outputs —
    through size-independent properties of EVERY step: history chaining between consecutive slots, reset rows, the three
    reward values, flag / reward consistency, lidar ranges; slot by slot equal to a second handle that gathers its bank
    from L2 instead of staging it in LDS (an independent instantiation of the kernel); and, for ~2 050 randomly chosen envs
    of the 65 536, every step of the trajectory against the ORACLE (reward / done exact, observations within 1e-5)."""
    torch = torch_cuda
    n, K = 65536, 200
    vec = _vec(n, n_maps=64, n_beams=8)
    ref = _vec(n, n_maps=64, n_beams=8, bank_in_global=True)
    F = vec.n_states
    smp = OracleSample(oracle, vec, 2048, seed=65536)
    obs0 = vec.reset_tensor().clone(); ref.reset_tensor()
    smp.reset(obs0)
    acts = vec.random_actions(2025, 0, K)
    to, tr, td, tf = vec.rollout_tensor(acts, trajectory=True)
    ro, rr, rd, rf = ref.rollout_tensor(acts, trajectory=True)
    assert torch.equal(to, ro) and torch.equal(tr, rr) and torch.equal(td, rd) and torch.equal(tf, rf)
    prev = obs0
    for k in range(K):
        obs, rew, done, flags = to[k], tr[k], td[k], tf[k]
        smp.step(acts[k], obs, rew, done, atol=ATOL)
        cont = done == 0
        assert torch.equal(obs[cont][:, :F], prev[cont][:, F:]), k        # oldest-first history (ship_env.py:113,181)
        rs = ~cont
        if rs.any():
            assert torch.all(obs[rs][:, :F] == -1) and torch.all(obs[rs][:, F] == vec.cfg.spawn_x) and torch.all(obs[rs][:, F + 1] == vec.cfg.spawn_y)
        assert torch.all((rew == 1.0) | (rew == -1.0) | (rew == -0.01))
        assert torch.equal(rew == 1.0, (flags & native.EV_GOAL_REACHED) != 0)
        assert torch.equal(done != 0, (flags & (native.EV_COLLIDING | native.EV_OUT_OF_BOUNDS | native.EV_MAX_STEPS | native.EV_NO_GOALS_LEFT)) != 0)
        lid = obs[:, F + 6:]
        assert torch.all((lid == -1) | ((lid >= 0) & (lid <= 100.0 + 1e-9)))
        prev = obs
    st = vec.stats()
    assert st["episodes"] == int(td.sum()) and st["episodes"] > n // 2
    assert torch.equal(to[K - 1][:, F], vec.field(native.F_X)) and torch.equal(to[K - 1][:, F + 1], vec.field(native.F_Y))
    assert smp.n_done > 1000 and smp.worst <= 1e-9
    print("configs[2] at 65 536 envs: %d sampled envs x %d fused steps vs the oracle, %d episode ends, max |obs - oracle| = %.3e"
          % (len(smp.idx), K, smp.n_done, smp.worst))


def test_action_tensors_are_checked(torch_cuda):
    """step_tensor / rollout_tensor hand the actions' ADDRESS to the library: a tensor of another dtype, size or device would be
    read as int32 [N] all the same (wrong steps, or a read past its end), so the host side refuses it."""
    torch = torch_cuda
    vec = _vec(300, n_maps=8)
    vec.reset_tensor()
    good = torch.zeros(300, dtype=torch.int32, device="cuda")
    vec.step_tensor(good)
    for bad in (torch.zeros(300, dtype=torch.int64, device="cuda"), torch.zeros(299, dtype=torch.int32, device="cuda"),
                torch.zeros(300, dtype=torch.int32), torch.zeros((300, 2), dtype=torch.int32, device="cuda")[:, 0]):
        with pytest.raises(ValueError):
            vec.step_tensor(bad)
    vec.reset_tensor(mask=torch.zeros(300, dtype=torch.uint8, device="cuda"))
    vec.reset_tensor(mask=torch.zeros(300, dtype=torch.bool, device="cuda"), map_ids=torch.zeros(300, dtype=torch.int32, device="cuda"))
    for bad in (torch.zeros(300, dtype=torch.int64, device="cuda"), torch.zeros(299, dtype=torch.uint8, device="cuda"), torch.zeros(300, dtype=torch.uint8)):
        with pytest.raises(ValueError):
            vec.reset_tensor(mask=bad)
    with pytest.raises(ValueError):
        vec.reset_tensor(map_ids=torch.zeros(300, dtype=torch.int64, device="cuda"))
    vec.rollout_tensor(torch.zeros((5, 300), dtype=torch.int32, device="cuda"))
    for bad in (torch.zeros((5, 300), dtype=torch.int64, device="cuda"), torch.zeros((5, 301), dtype=torch.int32, device="cuda"),
                torch.zeros(300, dtype=torch.int32, device="cuda"), torch.zeros((5, 600), dtype=torch.int32, device="cuda")[:, ::2]):
        with pytest.raises(ValueError):
            vec.rollout_tensor(bad)
    vec.close()


def test_every_workgroup_layout_against_the_oracle(torch_cuda, oracle, native):
    """The step kernel's workgroup layouts — 64 and 128 envs with SIX wave roles (what batches of <= 32 768 envs get:
    collide_ship on waves of its own), 256 envs with four — each stepped against the oracle at a size the oracle handles (SSG_BLOCK
    forces the layout): single steps, then a fused trajectory rollout compared slot by slot; staged and gathered bank."""
    import os
    torch = torch_cuda
    for blk, kw in (("64", {}), ("128", {}), ("256", {}), ("64", {"bank_in_global": True}), ("128", {"bank_in_global": True}),
                    ("256", {"bank_in_global": True})):
        os.environ["SSG_BLOCK"] = blk
        try:
            for nb in (8, 10):
                vec = _vec(1200, n_maps=64, n_beams=nb, **kw)
                assert vec.launch_geometry()[0] == int(blk)
                err, n_done = run_pair(oracle, native, vec, K=150)
                assert err <= 1e-9 and n_done > 30
                # fused trajectory: every slot against single steps of a second handle on the same layout
                ref = _vec(1200, n_maps=64, n_beams=nb, **kw)
                vec.reset_tensor(); ref.reset_tensor()
                acts = vec.random_actions(77, 0, 60)
                to, tr, td, tf = vec.rollout_tensor(acts, trajectory=True)
                for k in range(60):
                    o, r, d, f = ref.step_tensor(acts[k])
                    assert torch.equal(to[k], o) and torch.equal(tr[k], r) and torch.equal(td[k], d) and torch.equal(tf[k], f), (blk, nb, k)
                vec.close(); ref.close()
        finally:
            del os.environ["SSG_BLOCK"]


def test_full_size_bank_too_large_for_the_lds(torch_cuda, oracle, native):
    """65 536 envs on a bank of 120 records: it fits the LDS only beside 64-env workgroups (four rounds per launch), so
    ssg_set_map_bank gathers it from L2 on 256-env workgroups.  ~2 050 sampled envs against the oracle, every fused step; and
    the whole batch, slot by slot, against the same bank staged beside 64-env workgroups (SSG_BLOCK forces that layout)."""
    import os
    torch = torch_cuda
    n, K = 65536, 120
    vec = _vec(n, n_maps=120, n_beams=8)
    assert vec.launch_geometry()[:2] == (256, False)
    os.environ["SSG_BLOCK"] = "64"
    try:
        ref = _vec(n, n_maps=120, n_beams=8)
        assert ref.launch_geometry()[:2] == (64, True)
    finally:
        del os.environ["SSG_BLOCK"]
    smp = OracleSample(oracle, vec, 2048, seed=120)
    smp.reset(vec.reset_tensor().clone()); ref.reset_tensor()
    acts = vec.random_actions(2026, 0, K)
    to, tr, td, tf = vec.rollout_tensor(acts, trajectory=True)
    ro, rr, rd, rf = ref.rollout_tensor(acts, trajectory=True)
    assert torch.equal(to, ro) and torch.equal(tr, rr) and torch.equal(td, rd) and torch.equal(tf, rf)
    for k in range(K):
        smp.step(acts[k], to[k], tr[k], td[k], atol=ATOL)
    assert smp.n_done > 500 and smp.worst <= 1e-9
    assert int(vec.field(native.F_MAP_ID).max()) < 120
    vec.close(); ref.close()


def test_trajectory_rewards_add_up_to_the_in_kernel_episode_statistics(torch_cuda, native):
    """What train/random.py:14-27 does with the per-step tuples — accumulate `total_reward` until `done` — done on the
    trajectory tensors of a fused rollout must give exactly the episode statistics the step kernel accumulates itself
    (integer hundredths: returns are sums of {1, -1, -0.01})."""
    torch = torch_cuda
    n, K = 20000, 300
    vec = _vec(n, n_maps=64, n_beams=8)
    vec.reset_tensor()
    acts = vec.random_actions(55, 0, K)
    to, tr, td, tf = vec.rollout_tensor(acts, trajectory=True)
    cents = torch.round(tr * 100.0).to(torch.int64)                     # [K, n] rewards in hundredths
    running = torch.zeros(n, dtype=torch.int64, device=vec.device)
    total_ret, total_len, episodes = 0, 0, 0
    length = torch.zeros(n, dtype=torch.int64, device=vec.device)
    for k in range(K):
        running += cents[k]
        length += 1
        d = td[k] != 0
        total_ret += int(running[d].sum()); total_len += int(length[d].sum()); episodes += int(d.sum())
        running[d] = 0; length[d] = 0
    st = vec.stats()
    assert episodes == st["episodes"] and total_len == st["sum_length"]
    assert total_ret == int(round(st["sum_return"] * 100.0))
    assert int((tf & native.EV_GOAL_REACHED != 0).sum()) == st["goals_hit"]


def test_single_env_fresh_mode_matches_oracle(torch_cuda, oracle, native):
    """configs[0] analogue: the ShipEnv facade in reference-exact 'fresh' map mode against an oracle World fed the
    same RNG streams; seeds python random and numpy as SURVEY App. B-12/17 prescribes."""
    import random
    from ship_sim_gym_amd.ship_env import ShipEnv
    from ship_sim_gym_amd import worldgen
    for seed in (0, 1, 2):
        random.seed(seed); np.random.seed(seed)
        env = ShipEnv()
        o = env.reset()
        random.seed(seed); np.random.seed(seed)
        # oracle side: construction consumes one world, reset() another (game.py:58, App. B-17)
        worldgen.generate_world((600, 600))
        _, polys, goals = worldgen.generate_world((600, 600))
        w = oracle.World()
        ro = w.reset(polys[0], polys[1], goals)
        np.testing.assert_array_equal(o, ro)
        rng = np.random.RandomState(seed + 100)
        for t in range(400):
            a = int(rng.randint(3))
            o, r, d, info = env.step(a)
            ro, rr, rd = w.step(a)
            assert r == rr and d == rd and info == {}
            np.testing.assert_allclose(o, ro, rtol=0, atol=ATOL)
            assert env.game.colliding == bool(w.peek()["colliding"])
            if d:
                break
        env.close()


def test_vec_env_protocols(torch_cuda, native):
    """stable-baselines VecEnv and RLlib VectorEnv surfaces + the ship_gym alias package."""
    from ship_gym.ship_env import ShipEnv as AliasEnv          # what the reference's scripts import
    from ship_gym.config import EnvConfig, GameConfig
    from ship_sim_gym_amd.ship_env import ShipEnv
    assert AliasEnv is ShipEnv and GameConfig.BOUNDS == (600, 600) and EnvConfig.HISTORY_SIZE == 2
    v = _vec(8, n_maps=4)
    assert v.num_envs == 8 and v.action_space.n == 3 and v.observation_space.shape == (32,)
    assert v.observation_space.dtype == np.uint8 and v.reward_range == (-1, 1)      # ship_env.py:18-20,48
    o = v.reset()
    assert o.shape == (8, 32) and o.dtype == np.float64 and np.all(o[:, :16] == -1)
    v.step_async(np.zeros(8, dtype=np.int64))
    o, r, d, infos = v.step_wait()
    assert o.shape == (8, 32) and r.shape == (8,) and d.dtype == bool and infos == [{}] * 8
    with pytest.raises(AssertionError):
        v.step(np.full(8, 3))                                   # Discrete(3): action 3 is rejected (ship_env.py:143)
    obs_l, rew_l, done_l, infos = v.vector_step([0] * 8)        # RLlib VectorEnv
    assert len(obs_l) == 8 and obs_l[0].shape == (32,) and len(v.vector_reset()) == 8
    o1 = v.reset_at(3)
    assert o1.shape == (32,) and np.all(o1[:16] == -1) and o1[16] == 300
    assert [h.index for h in v.get_unwrapped()] == list(range(8)) and v.get_attr('num_envs', indices=[2]) == [8]
    assert v.env_method('seed', 11, indices=[0, 1]) == [[11], [11]]
    assert v.seed(7) == [7]
    # auto-reset returns the reset observation: run until some env is done
    seen = False
    for _ in range(400):
        o, r, d, _ = v.step(np.zeros(8, dtype=np.int64))
        if d.any():
            seen = True
            assert np.all(o[d][:, :16] == -1) and np.all(o[d][:, 16] == 300)
            break
    assert seen
    v.close()
    with pytest.raises(ValueError):
        class E(EnvConfig):
            HISTORY_SIZE = 0
        _vec(4, env_config=E)


def test_numpy_protocol_is_the_tensor_path_through_pinned_blocks(torch_cuda, native):
    """step_async LAUNCHES (pinned actions -> device, ssg_step, one device -> host copy of the packed obs | reward | done | flags
    block, side stream), step_wait waits on the event and hands out numpy views of a rotating pinned block: every step equals the
    tensor API's on a twin env, an array handed out stays intact for host_slots - 1 further steps, tensor-API calls between
    numpy steps are ordered with them, and copy_host_outputs=True returns arrays nobody rewrites."""
    import torch
    n = 4096
    a_env, b_env = _vec(n, n_maps=8), _vec(n, n_maps=8)
    c_env = _vec(n, n_maps=8, copy_host_outputs=True, host_slots=2)
    assert a_env.obs.data_ptr() == a_env._out_blob.data_ptr() and a_env.obs.is_contiguous()
    np.testing.assert_array_equal(a_env.reset(), b_env.reset_tensor().cpu().numpy())
    c_env.reset()
    acts = b_env.random_actions(2024, 0, 60)
    acts_h = acts.cpu().numpy()
    held, kept = [], []
    for k in range(60):
        a_env.step_async(acts_h[k].astype(np.int64))
        want = [t.cpu().numpy() for t in b_env.step_tensor(acts[k])]      # (the other env steps while a's step is in flight)
        o, r, d, infos = a_env.step_wait()
        np.testing.assert_array_equal(o, want[0]); np.testing.assert_array_equal(r, want[1])
        np.testing.assert_array_equal(d, want[2].astype(bool))
        assert d.dtype == np.bool_ and len(infos) == n and infos[0] == {}
        held.append((o, want[0]))
        for oo, ww in held[-(a_env.host_slots - 1):]:                     # views of the last host_slots - 1 steps are intact
            np.testing.assert_array_equal(oo, ww)
        oc, rc, dc, _ = c_env.step(acts_h[k])
        kept.append((oc, want[0]))
        if k == 30:                                                       # a tensor-API call between two numpy steps
            m = torch.zeros(n, dtype=torch.uint8, device=a_env.device); m[:100] = 1
            ids = (a_env.field(native.F_MAP_ID) + 1) % a_env.n_maps
            a_env.reset_tensor(mask=m, map_ids=ids.to(torch.int32).contiguous())
            b_env.reset_tensor(mask=m.clone(), map_ids=(b_env.field(native.F_MAP_ID) + 1).remainder(b_env.n_maps).to(torch.int32).contiguous())
            c_env.reset_tensor(mask=m.clone(), map_ids=(c_env.field(native.F_MAP_ID) + 1).remainder(c_env.n_maps).to(torch.int32).contiguous())
    for oc, ww in kept:                                                   # fresh arrays: all 60 still what they were
        np.testing.assert_array_equal(oc, ww)
    with pytest.raises(native.ShipSimError):
        a_env.step_wait()                                                 # no step in flight
    a_env.close(); b_env.close(); c_env.close()


def test_terminal_obs_is_refused_where_it_cannot_be_served(torch_cuda, native):
    """ssg_set_terminal_obs needs the in-kernel auto-reset (otherwise a done env's row of `obs` already IS its terminal observation) and
    history <= 2 (longer histories are assembled by the frame-shift kernel); a trajectory rollout keeps every step in its own slot and
    never writes the side buffer."""
    import torch
    from ship_sim_gym_amd.config import EnvConfig

    class H3(EnvConfig):
        HISTORY_SIZE = 3

    v = _vec(64, n_maps=4, auto_reset=False)
    with pytest.raises(native.ShipSimError):
        v.enable_terminal_obs()
    v.close()
    v = _vec(64, n_maps=4, env_config=H3)
    with pytest.raises(native.ShipSimError):
        v.enable_terminal_obs()
    v.close()
    v = _vec(512, n_maps=4)
    term = v.enable_terminal_obs()
    term.fill_(-7.0)
    v.reset_tensor()
    to, tr, td, tf = v.rollout_tensor(v.random_actions(3, 0, 150), trajectory=True)
    assert td.sum() > 0 and bool((term == -7.0).all())          # episodes ended, the side buffer was not touched
    v.rollout_tensor(v.random_actions(4, 0, 150))                # overwrite mode: it is
    assert bool((term != -7.0).any())
    v.close()


@pytest.mark.parametrize("hist", [2, 3])
def test_rllib_flow_terminal_obs_and_single_reset(torch_cuda, native, hist):
    """RLlib VectorEnv flow (train/rllib/ppo.py:21-24,43): vector_step returns the TERMINAL observation of a done env,
    reset_at(i) is the single reset.  The trajectory must equal the SB-protocol env's (in-kernel auto-reset) step for
    step: same rewards / dones, same observations except on done rows, where SB reports the reset observation that
    RLlib gets from reset_at.  HISTORY_SIZE 3 takes the frame-shift route: the steps after a reset must show the
    reference's [-1]*(H-1) frames + spawn frame history, not the finished episode's frames."""
    from ship_sim_gym_amd.config import EnvConfig

    class E(EnvConfig):
        HISTORY_SIZE = hist

    sb = _vec(64, n_maps=8, env_config=E)
    rl = _vec(64, n_maps=8, rllib=True, env_config=E)
    assert rl.rllib and rl.auto_reset == (hist <= 2) and (rl.term_obs is not None) == (hist <= 2)  # (history <= 2: no reset launch at all)
    o_sb, o_rl = sb.reset(), np.stack(rl.vector_reset())
    np.testing.assert_array_equal(o_sb, o_rl)
    acts = sb.random_actions(77, 0, 300).cpu().numpy()
    n_done = 0
    for k in range(300):
        o_sb, r_sb, d_sb, _ = sb.step(acts[k])
        o_l, r_l, d_l, _ = rl.vector_step(list(acts[k]))
        o_rl = np.stack(o_l)
        np.testing.assert_array_equal(np.asarray(r_l), r_sb)
        np.testing.assert_array_equal(np.asarray(d_l), d_sb)
        np.testing.assert_array_equal(o_rl[~d_sb], o_sb[~d_sb])
        for i in np.nonzero(d_sb)[0]:
            n_done += 1
            assert not np.all(o_rl[i][16 * (hist - 2):16 * (hist - 1)] == -1)  # terminal observation: a real previous frame
            np.testing.assert_array_equal(rl.reset_at(int(i)), o_sb[i])   # the one reset == SB's auto-reset observation
    assert n_done > 30
    np.testing.assert_array_equal(sb.field(native.F_MAP_ID).cpu().numpy(), rl.field(native.F_MAP_ID).cpu().numpy())
    sb.close(); rl.close()


def test_from_env_fns_and_step_after_done(torch_cuda, native):
    """`SubprocVecEnv([make_env() for i in range(n)])` (train/stable_baselines/ppo.py:54-76,122-123) -> one batched env;
    the ShipEnv facade refuses to step a finished episode."""
    from ship_gym.config import EnvConfig, GameConfig
    from ship_gym.ship_env import ShipEnv
    from ship_sim_gym_amd.vec_env import ShipVecEnv

    class GC(GameConfig):
        SPEED = 30
        BOUNDS = (1000, 1000)

    def make_env():
        def _init():
            return ShipEnv(GC, EnvConfig)
        return _init

    v = ShipVecEnv.from_env_fns([make_env() for _ in range(12)], n_maps=4)
    assert v.num_envs == 12 and v.bounds == (1000, 1000) and v.cfg.dt == 30 * 0.1 and v.game_config is GC
    o = v.reset()
    assert o.shape == (12, 32) and np.all(o[:, 16] == 500)
    v.close()
    e = ShipEnv(GC, EnvConfig)
    e.reset()
    done = False
    for _ in range(1000):
        _, _, done, _ = e.step(0)
        if done:
            break
    assert done
    with pytest.raises(native.ShipSimError):
        e.step(0)
    e.reset()
    e.step(0)
    e.close()


def test_curriculum_maps_switch_banks(torch_cuda, oracle, native):
    """BASELINE configs[3]'s "curriculum maps": a lesson change installs the next width's bank and resets; the HIP
    path on the new bank still matches the oracle on the same bank."""
    from ship_sim_gym_amd.curriculum import CurriculumMaps
    vec = _vec(512, n_maps=8)
    cm = CurriculumMaps(vec, widths=(0.5, 0.7), conditions=(0.0,), repeat_condition=0, n_maps=8)
    assert cm.width_frac == 0.5 and cm.progress(-1.0) is None
    obs = cm.progress(1.0)
    assert obs is not None and cm.width_frac == 0.7 and cm.progress(5.0) is None
    # wider banks: left-bank hull reaches further into the river than at width 0.5
    assert float(vec.bank[:, 4].max()) > 150.0 + 1e-9
    err, n_done = run_pair(oracle, native, vec, K=150)
    assert err <= ATOL and n_done > 50


def test_history_sizes_parity(torch_cuda, oracle, native):
    """EnvConfig.HISTORY_SIZE other than the default 2 (ship_env.py:44-47,180-181): 1, 3 and 5 frames."""
    from ship_sim_gym_amd.config import EnvConfig
    for hist in (1, 3, 5):
        class E(EnvConfig):
            HISTORY_SIZE = hist
        vec = _vec(300, env_config=E, n_maps=8)
        assert vec.observation_space.shape == (16 * hist,)
        err, n_done = run_pair(oracle, native, vec, K=150)
        assert err <= ATOL and n_done > 30
        # the fused-rollout entry point takes the same per-step route for history > 2
        a, b = _vec(300, env_config=E, n_maps=8), _vec(300, env_config=E, n_maps=8)
        a.reset_tensor(); b.reset_tensor()
        acts = a.random_actions(5, 0, 60)
        for k in range(60):
            a.step_tensor(acts[k])
        b.rollout_tensor(acts)
        assert torch_cuda.equal(a.obs, b.obs) and torch_cuda.equal(a.state, b.state)


def test_edge_configurations(torch_cuda, oracle, native):
    """Edges: a one-map bank, a one-step episode limit, 16 beams (gathers the bank from L2: does not fit LDS at 256
    envs per workgroup), a single beam, bounds 1000 with the RLlib script's SPEED 40, FIX_COLLISION_REWARD."""
    from ship_sim_gym_amd.config import EnvConfig, GameConfig

    class E1(EnvConfig):
        MAX_STEPS = 1
    v = _vec(130, env_config=E1, n_maps=1)
    err, n_done = run_pair(oracle, native, v, K=5)
    assert n_done == 130 * 5 and err <= ATOL          # every step ends an episode; auto-reset onto the same single map

    for nb in (1, 16):
        v = _vec(70000 if nb == 16 else 200, n_maps=64, n_beams=nb)
        acts = v.random_actions(3, 0, 40)
        v.reset_tensor(); v.rollout_tensor(acts)
        small = _vec(200, n_maps=64, n_beams=nb)
        err, n_done = run_pair(oracle, native, small, K=80)
        assert err <= ATOL
        # the big run and the small run agree on the envs they share (same global ids, same actions)
        small2 = _vec(200, n_maps=64, n_beams=nb)
        small2.reset_tensor(); small2.rollout_tensor(small2.random_actions(3, 0, 40))
        assert torch_cuda.equal(v.obs[:200], small2.obs)

    class G(GameConfig):
        SPEED = 40
        BOUNDS = (1000, 1000)
    v = _vec(256, game_config=G, n_maps=8)              # train/rllib/ppo.py:12-16
    err, n_done = run_pair(oracle, native, v, K=100)
    assert err <= ATOL and n_done > 500

    v = _vec(512, n_maps=16, fix_collision_reward=True)
    v.reset_tensor()
    acts = v.random_actions(8, 0, 300)
    hit = 0
    for k in range(300):
        _, rew, done, flags = v.step_tensor(acts[k])
        col = (flags & native.EV_COLLIDING) != 0
        goal = (flags & native.EV_GOAL_REACHED) != 0
        assert torch_cuda.all(rew[col & ~goal] == -1.0)
        hit += int((col & ~goal).sum())
    assert hit > 20


def test_million_envs_smoke(torch_cuda, oracle, native):
    """BASELINE configs[4] size on ONE device (1 048 576 envs, 10 beams): runs, stays finite, statistics add up, and ~2 050
    randomly chosen envs of the million agree with the oracle at every step."""
    torch = torch_cuda
    v = _vec(1 << 20, n_maps=64, n_beams=10)
    smp = OracleSample(oracle, v, 2048, seed=1 << 20)
    smp.reset(v.reset_tensor())
    acts = v.random_actions(1, 0, 60)
    eps = 0
    for k in range(60):
        obs, rew, done, flags = v.step_tensor(acts[k])
        smp.step(acts[k], obs, rew, done, atol=ATOL)
        eps += int(done.sum())
    assert bool(torch.isfinite(obs).all()) and v.stats()["episodes"] == eps and eps > 1000
    assert smp.n_done > 100 and smp.worst <= 1e-9
    v.close()


@pytest.mark.parametrize("rank", [0, 5, 7])
def test_configs4_rank_share_against_the_oracle(torch_cuda, oracle, native, rank):
    """BASELINE configs[4] as a rank of the 8-GPU job sees it: 131 072 envs, 10 beams, global env ids
    [rank * 131 072, (rank + 1) * 131 072) — fused trajectory rollout (two launches), every step of ~2 050 sampled envs against
    oracle worlds built from their GLOBAL ids (bank record and Philox stream)."""
    torch = torch_cuda
    n, K = 131072, 120
    v = _vec(n, n_maps=64, n_beams=10, env_id_base=rank * n)
    smp = OracleSample(oracle, v, 2048, seed=rank)
    smp.reset(v.reset_tensor())
    acts = v.random_actions(12345, 0, K)
    np.testing.assert_array_equal(acts[:, smp.tidx[:8]].cpu().numpy(),
                                  np.stack([oracle.fill_actions(12345, 0, K, rank * n + int(e), 1)[:, 0] for e in smp.idx[:8]], axis=1))
    to, tr, td, tf = v.rollout_tensor(acts, trajectory=True)
    for k in range(K):
        smp.step(acts[k], to[k], tr[k], td[k], atol=ATOL)
    assert smp.n_done > 500 and smp.worst <= 1e-9
    v.close()


@pytest.mark.parametrize("n_ships", [1, 4])
def test_reference_random_rollout_configuration(torch_cuda, oracle, native, n_ships):
    """train/random.py:4-7 — the reference's own random-action rollout: GameConfig.SPEED = 1 (dt = 0.1, damping 0.4^0.1 per
    step), default bounds and EnvConfig: small steps, long episodes that mostly end on MAX_STEPS.  Single steps and a fused
    trajectory rollout against the oracle, with and without add_default_traffic."""
    torch = torch_cuda
    from ship_sim_gym_amd.config import EnvConfig, GameConfig

    class G(GameConfig):
        SPEED = 1

    v = _vec(700, game_config=G, env_config=EnvConfig, n_maps=16, n_ships=n_ships)
    assert abs(v.cfg.dt - 0.1) < 1e-15 and v.cfg.max_steps == 1000
    err, n_done = run_pair(oracle, native, v, K=1100, seed=4242, check_every=1)
    assert err <= (1e-9 if n_ships == 1 else 1e-8)
    st = v.stats()
    assert n_done >= 700 and st["episodes"] == n_done          # every env ended at least once (MAX_STEPS at step 1000 at the latest)
    v.close()
    # the same stream as one fused trajectory rollout (1-ship: 100-step launches; 4 ships: per-step launch sequences)
    w = _vec(700, game_config=G, env_config=EnvConfig, n_maps=16, n_ships=n_ships)
    smp = OracleSample(oracle, w, 700, seed=1)
    smp.reset(w.reset_tensor())
    acts = w.random_actions(4242, 0, 1050)
    to, tr, td, tf = w.rollout_tensor(acts, trajectory=True)
    for k in range(1050):
        smp.step(acts[k], to[k], tr[k], td[k], atol=ATOL, n_threads=8)
    assert smp.n_done >= 700 and smp.worst <= 1e-8
    w.close()


def _ppo_mod():
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ppo_torch", os.path.join(root, "train", "ppo_torch.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def test_trainer_glue_runs_end_to_end(torch_cuda, native):
    """a GPU-resident PPO loop (train/ppo_torch.py) drives ShipVecEnv through the zero-copy
    tensor API for a few updates: finite losses/returns, episodes accumulate, policy-in-the-loop stepping works — and the rollout
    step captured as ONE HIP graph {policy forward + sampling + ssg_step + buffer writes} replays bit for bit what the eager loop
    launches kernel by kernel (train/stable_baselines/ppo.py:84-100,122-123 is the loop being replaced): every rollout buffer of
    every update, the env state at the end, and the trained parameters."""
    import torch
    mod = _ppo_mod()
    lines = []
    hist, ref = mod.train(envs=1024, updates=3, horizon=32, log=lines.append, mode="eager", return_details=True)
    assert len(hist) == 3 and all(np.isfinite(h[1]) and np.isfinite(h[3]) for h in hist)
    assert "env-steps/s" in lines[-1]
    assert ref["snapshots"][0]["done"].sum() > 0                       # episodes ended (and were reset in-kernel) inside the rollouts
    hist_g, got = mod.train(envs=1024, updates=3, horizon=32, log=lines.append, mode="graph", return_details=True)
    assert hist_g == hist
    for u in range(3):
        for k, v in ref["snapshots"][u].items():
            assert torch.equal(v, got["snapshots"][u][k]), (u, k)
    for name, col in ref["final_state"][0].items():
        assert torch.equal(col, got["final_state"][0][name]), name
    for a, b in zip(ref["params"], got["params"]):
        assert torch.equal(a, b)


def test_trainer_glue_ping_pong_halves(torch_cuda, native):
    """The two-half-batch ping-pong (half A's env step on one stream while half B's policy forward runs on another, one HIP graph
    each): the halves are shards of the same batch (global env ids), so replaying the actions it took through ONE unsplit env
    reproduces its observations, rewards and dones exactly."""
    import torch
    mod = _ppo_mod()
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    n, H = 2048, 24
    hist, got = mod.train(envs=n, updates=1, horizon=H, log=lambda s: None, mode="pingpong", return_details=True)
    snap = got["snapshots"][0]
    env = ShipVecEnv(n, n_maps=64)
    obs = env.reset_tensor()
    scale = float(max(env.bounds))
    for t in range(H):
        assert torch.equal((obs / scale).float(), snap["obs"][t]), t
        obs, rew, done, _ = env.step_tensor(snap["act"][t].to(torch.int32))
        assert torch.equal(rew.float(), snap["rew"][t]) and torch.equal(done.float(), snap["done"][t]), t
    env.close()


def test_device_bank_generation_matches_host_geometry(torch_cuda, oracle, native):
    """records generated on the device (ssg_generate_bank) are, bit for bit, what the host path
    builds from the same raw polygons and goal draws; the draws respect gen_river_poly's ranges; the env steps on a
    device-generated bank exactly like the oracle fed the same polygons and goals."""
    from ship_sim_gym_amd import worldgen
    vec = _vec(512, n_maps=32)
    raw = vec.regenerate_bank(seed=99, return_raw=True).cpu().numpy()
    bank = vec.bank.cpu().numpy()
    polys = raw[:, :48].reshape(32, 2, 12, 2)
    goals = np.zeros((32, 5, 2))
    for m in range(32):
        bare = worldgen.build_record(polys[m, 0], polys[m, 1], np.zeros((0, 2)), (300.0, 25.0))
        for i in range(5):
            y, u, fb = raw[m, 48 + 3 * i: 51 + 3 * i]
            hit, lo, hi = worldgen.goal_x_range(bare, 600.0, y)
            goals[m, i] = [lo + (hi - lo) * u if hit else fb, y]
            assert 100 * (i + 1) - 20 <= y <= 100 * (i + 1) + 20 and 0 <= u < 1
        rec = worldgen.build_record(polys[m, 0], polys[m, 1], goals[m], (300.0, 25.0))
        np.testing.assert_array_equal(bank[m], rec)
        # game_map.py:22-73 ranges: left bank x in [0, 150], right in [450, 600]; corners appended last
        assert np.all((polys[m, 0, :10, 0] >= 0) & (polys[m, 0, :10, 0] <= 150))
        assert np.all((polys[m, 1, :10, 0] >= 450) & (polys[m, 1, :10, 0] <= 600))
        assert polys[m, 0, 10:].tolist() == [[0, 600], [0, 0]] and polys[m, 1, 10:].tolist() == [[600, 600], [600, 0]]
    assert len({bank[m].tobytes() for m in range(32)}) == 32  # all maps differ
    other = _vec(64, n_maps=32)
    other.regenerate_bank(seed=100)
    assert not torch_cuda.equal(other.bank, vec.bank)
    vec.bank_polys, vec.bank_goals = polys, goals
    err, n_done = run_pair(oracle, native, vec, K=150)
    assert err <= ATOL and n_done > 50


def test_two_handles_with_different_bank_sizes_interleave(torch_cuda, native):
    """The dynamic-LDS cap is a property of the kernel, not of a handle: a small-bank handle created after a
    big-bank one must not break the big one's launches."""
    big = _vec(65536, n_maps=64, n_beams=8)
    big.reset_tensor(); big.step_tensor(big.random_actions(1, 0, 1)[0])
    small = _vec(65536, n_maps=2, n_beams=8)
    small.reset_tensor(); small.step_tensor(small.random_actions(1, 0, 1)[0])
    ref = _vec(65536, n_maps=64, n_beams=8)
    ref.reset_tensor()
    acts = ref.random_actions(1, 0, 3)
    for k in range(3):
        ref.step_tensor(acts[k])
    for k in range(1, 3):
        big.step_tensor(acts[k])
    torch_cuda.cuda.synchronize()
    assert torch_cuda.equal(big.obs, ref.obs)


@pytest.mark.gpu
def test_rgb_array_frames():
    """ShipGame.render / get_screen (game.py:133-138,197-229) rasterised on the GPU: colours at known world points."""
    import torch
    from ship_sim_gym_amd import _native as N
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    vec = ShipVecEnv(8, n_maps=8, n_ships=4)
    vec.reset_tensor()
    img = vec.get_screen(3).cpu().numpy()                       # [x][y][rgb], screen y down (pygame.surfarray.array3d)
    assert img.shape == (600, 600, 3) and img.dtype == np.uint8

    def at(x, y):
        return tuple(int(v) for v in img[int(x), int(600 - y)])

    assert at(300, 25) == (255, 255, 0)                          # yellow marker at the player's position (game.py:229)
    assert at(305, 60) == (255, 255, 255)                        # the player's hull, white (game.py:275)
    assert at(405, 370) == (0, 0, 0) and at(307, 210) == (0, 0, 0)   # traffic ships, black (game.py:284-286)
    assert at(2, 300) == (139, 69, 19) and at(598, 300) == (139, 69, 19)   # banks (models.py:181)
    assert at(250, 140) in ((0, 0, 200), (0, 255, 0))            # open water (or a goal / a free beam's end marker)
    g = vec.bank_goals[3 % 8]
    seen_green = sum(at(gx, gy) == (0, 255, 0) for gx, gy in g)
    assert seen_green >= 3                                       # goals are green discs (game.py:88)
    plain = vec.get_screen(3, debug=False).cpu().numpy()         # GameConfig.DEBUG off: blue screen + the yellow marker
    cols = {tuple(c) for c in plain.reshape(-1, 3)}
    assert cols == {(0, 0, 200), (255, 255, 0)}
    small = vec.render(mode='rgb_array', env=3)
    assert small.shape == (600, 600, 3) and tuple(small[600 - 25, 300]) == (255, 255, 0)   # [row = y][col = x]
    half = vec.get_screen(3, width=300, height=300).cpu().numpy()
    assert tuple(half[150, 300 - 13]) == (255, 255, 0)
    vec.close()


@pytest.mark.parametrize("n_ships", [1, 4])
def test_abi_garbage_blob_and_shrinking_bank(torch_cuda, native, n_ships):
    """A C caller may bind device memory that was never zeroed: the first full ssg_reset after ssg_bind_state must
    start from clean counters / config-4 columns.  Record indices can never leave the bank: ids given to ssg_reset
    are folded modulo n_maps, and installing a smaller bank folds the ids already stored."""
    import ctypes as C
    torch = torch_cuda
    n = 700
    ref = _vec(n, n_maps=8, n_ships=n_ships)
    dirty = _vec(n, n_maps=8, n_ships=n_ships)
    dirty.state.fill_(0xAB)                                             # garbage everywhere, incl. dyn_count
    native.check(native.lib().ssg_bind_state(dirty._h, C.c_void_p(dirty.state.data_ptr())), dirty._h, "rebind")
    ref.reset_tensor(); dirty.reset_tensor()
    acts = ref.random_actions(31, 0, 120)
    for k in range(120):
        a = [t.clone() for t in ref.step_tensor(acts[k])]
        b = dirty.step_tensor(acts[k])
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert ref.stats() == dirty.stats() and ref.stats()["episodes"] > 0
    # ids beyond the bank at reset time are folded, not used as they are
    ids = torch.full((n,), 8 + 3, dtype=torch.int32, device=ref.device)
    ref.reset_tensor(map_ids=ids)
    assert torch.all(ref.field(native.F_MAP_ID) == 3)
    # shrink the bank from 8 to 3 maps without resetting: stored ids must be folded before the next step reads them
    dirty.reset_tensor(map_ids=torch.arange(n, dtype=torch.int32, device=ref.device) % 8)
    small = dirty.bank[:3].clone()
    dirty.set_bank(small)
    dirty.step_tensor(acts[0])
    torch.cuda.synchronize()
    assert int(dirty.field(native.F_MAP_ID).max()) < 3
    ref.close(); dirty.close()


def test_lds_fit_fallbacks_are_exercised_and_exact(torch_cuda, oracle, native):
    """ssg_set_map_bank stages the bank in the CU's LDS at the workgroup size preferred for the env count when it fits there;
    when it only fits a smaller workgroup it weighs that against gathering the bank from L2 on a larger one (a smaller
    workgroup = more workgroups than the chip holds at once = launches of several rounds), and gathers when nothing fits:
    every one of those layouts must step exactly like the others."""
    torch = torch_cuda
    seen = set()
    for n, nb, n_maps, want in ((65536, 8, 64, (256, True)), (65536, 10, 64, (256, True)), (65536, 16, 64, (128, True)),
                                (65536, 16, 100, (128, False)), (65536, 16, 140, (128, False)), (65536, 10, 140, (256, False)),
                                (65536, 8, 100, (256, False)), (20000, 10, 64, (128, True)), (20000, 16, 100, (128, False)),
                                (4096, 10, 64, (64, True)), (4096, 16, 100, (64, True)), (4096, 10, 200, (64, False))):
        v = _vec(n, n_beams=nb, n_maps=n_maps)
        epw, lds, nbytes = v.launch_geometry()
        assert (epw, lds) == want and nbytes <= 160 * 1024, (n, nb, n_maps, epw, lds, nbytes)
        seen.add((epw, lds))
        v.close()
    assert seen == {(256, True), (128, True), (64, True), (64, False), (128, False), (256, False)}
    # exactness of each layout at a size the oracle handles: same n_beams / n_maps, fewer envs, geometry forced by SSG_BLOCK
    import os
    for nb, n_maps, blk in ((16, 64, "128"), (16, 100, "64"), (16, 140, "64"), (16, 140, "128"), (10, 140, "256"), (10, 64, "256")):
        os.environ["SSG_BLOCK"] = blk
        try:
            v = _vec(1500, n_beams=nb, n_maps=n_maps)
            assert v.launch_geometry()[0] <= int(blk)
            err, n_done = run_pair(oracle, native, v, K=120)
            assert err <= ATOL and n_done > 20
            v.close()
        finally:
            del os.environ["SSG_BLOCK"]


def test_randomised_configuration_sweep(torch_cuda, oracle, native):
    """Twenty seeded random configurations — beam count 1..16, history 1..3, SPEED, BOUNDS, river width, bank size, env
    count (ragged), episode length limit, config 4 — stepped against the oracle with single-step launches AND fused
    launches; every output of every step (single) / the final outputs (fused) must match: done / reward bit-exact,
    observations within the 1e-5 tolerance."""
    import random
    from ship_sim_gym_amd.config import EnvConfig, GameConfig
    from ship_sim_gym_amd import worldgen
    rng = random.Random(20261003)
    worst = 0.0
    for it in range(20):
        nb = rng.choice([1, 2, 3, 5, 7, 8, 9, 10, 12, 16])
        hist = rng.choice([1, 2, 2, 2, 3])
        speed, bounds = rng.choice([(10, (600, 600)), (30, (1000, 1000)), (40, (1000, 1000)), (20, (600, 600))])
        n = rng.choice([1, 65, 200, 257, 700, 1025])
        n_maps = rng.choice([1, 3, 16, 64])
        n_ships = 4 if (hist <= 2 and nb <= 10 and rng.random() < 0.25) else 1
        wf = rng.choice([0.5, 0.6, 0.7])

        class G(GameConfig):
            SPEED = speed
            BOUNDS = bounds

        class E(EnvConfig):
            HISTORY_SIZE = hist
            MAX_STEPS = rng.choice([7, 50, 1000])

        def make():
            bank, polys, goals = worldgen.build_bank(n_maps, bounds, width_frac=wf, seed=500 + it)
            v = _vec(n, game_config=G, env_config=E, n_beams=nb, bank=bank, n_ships=n_ships)
            v.bank_polys, v.bank_goals = polys, goals
            return v
        cfgtxt = "it=%d nb=%d hist=%d speed=%d n=%d maps=%d ships=%d wf=%.1f" % (it, nb, hist, speed, n, n_maps, n_ships, wf)
        v = make()
        K = 60
        try:
            err, n_done = run_pair(oracle, native, v, K=K, seed=it)
        except AssertionError as ex:
            raise AssertionError(cfgtxt + ": " + str(ex))
        worst = max(worst, err)
        # fused launches from a fresh handle must end where the single steps ended
        w = make()
        w.reset_tensor()
        w.rollout_tensor(w.random_actions(it, 0, K))
        torch_cuda.cuda.synchronize()
        assert torch_cuda.equal(w.obs, v.obs) and torch_cuda.equal(w.reward, v.reward) and torch_cuda.equal(w.done, v.done), cfgtxt
        if n_ships == 1:
            if not torch_cuda.equal(w.state, v.state):
                d = (w.state != v.state).nonzero().flatten().cpu().numpy()
                raise AssertionError("%s: state blobs differ at %d bytes, first offsets %r" % (cfgtxt, len(d), d[:6].tolist()))
        else:  # (the blob also holds the dyn work queue, whose ORDER depends on the order of the workgroups' atomics)
            for fid in (native.F_X, native.F_Y, native.F_VX, native.F_VY, native.F_ANGLE, native.F_W, native.F_LIDAR, native.F_RUDDER,
                        native.F_STEP_COUNT, native.F_MAP_ID, native.F_GOAL_MASK, native.F_TRAFFIC, native.F_GOAL_BODIES, native.F_DYN_FLAGS):
                assert torch_cuda.equal(w.field(fid), v.field(fid)), (cfgtxt, fid)
        v.close(); w.close()
    assert worst <= ATOL
