"""Control-flow goldens: streams recorded by EXECUTING the reference's own Python (ship_gym/ship_env.py, game.py, models.py,
imported unmodified from /root/reference in the build container) under the test-only stand-ins of tests/golden/shims, whose
physics primitives are the CPU oracle's (tests/golden/make_golden_controlflow.py -> tests/golden/ref_controlflow.npz).

They pin the PYTHON LAYER of the path by execution — observation layout and history deque, sticky lidar and its loop order,
the reward overwrite, is_done's order, action decoding, closest_goal, gen_goal_path's RNG call order — NOT Chipmunk2D's
arithmetic (the stand-in's physics IS the oracle: parity stays "unpinned" at the pymunk boundary).  Replayed here on
(a) the oracle's C world (whose control flow is a restatement of that Python), (b) the product's host-side world
generation from the same seeds, and (c, -m gpu) the HIP ShipEnv facade."""
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "ref_controlflow.npz")
ATOL = 1e-5  # north_star tolerance (HIP path); the oracle replay is held to 1e-9


def _streams():
    z = np.load(PATH)
    names = sorted({k.rsplit("/", 1)[0] for k in z.files if "/" in k})
    return z, names


# every assertion message below carries this: a failure here is a failure of the Python layer's restatement, and a pass says
# nothing about Chipmunk2D (the recorded physics is the oracle's own)
CAVEAT = " [control-flow golden: python layer only, physics = oracle stand-in, not Chipmunk parity evidence]"


def _cfg_of(z, name):
    speed, bw, bh, max_steps, hist = z[name + "/config"]
    return float(speed), (int(bw), int(bh)), int(max_steps), int(hist)


def test_fixture_shape_and_coverage():
    z, names = _streams()
    scope = str(z["__scope__"])  # the fixture says itself what it pins
    assert "PYTHON LAYER ONLY" in scope and "NOT Chipmunk parity evidence" in scope
    assert len(names) == 52  # ten scenarios (two of them with HISTORY_SIZE 1 / 3) x seeds 0..3, and six x seeds 0..1 with add_default_traffic()
    assert sum("_traffic/" in n for n in names) == 12
    seen = {"collision": 0, "goal": 0, "max_steps": 0, "oob": 0, "episodes": 0, "steps": 0}
    for n in names:
        done, rew = z[n + "/done"], z[n + "/reward"]
        col, goal = z[n + "/colliding"], z[n + "/goal_reached"]
        speed, bounds, max_steps, hist = _cfg_of(z, n)
        obs = z[n + "/obs"]
        assert obs.shape == (len(done), 16 * hist) and set(np.unique(rew)) <= {-1.0, -0.01, 1.0}
        seen["collision"] += int(col.sum()); seen["goal"] += int(goal.sum()); seen["episodes"] += int(done.sum())
        seen["steps"] += len(done)
        x, y = obs[:, 16 * (hist - 1)], obs[:, 16 * (hist - 1) + 1]  # the newest frame
        seen["oob"] += int(((x < 0) | (x > bounds[0]) | (y < 0) | (y > bounds[1])).sum())
        # the reward-overwrite quirk of determine_reward, as the reference itself produced it: an in-bounds collision that
        # reaches no goal is rewarded -0.01, not -1 (ship_env.py:66-77)
        inb = (x >= 0) & (x <= bounds[0]) & (y >= 0) & (y <= bounds[1])
        assert np.all(rew[col & ~goal & inb] == -0.01)
        assert np.all(done[col])
        starts = z[n + "/episode_start"]
        lens = np.diff(np.append(starts, len(done)))
        seen["max_steps"] += int((lens == max_steps).sum())
    assert seen["collision"] >= 20 and seen["goal"] >= 10 and seen["max_steps"] >= 1 and seen["steps"] > 4000, seen


def test_oracle_world_replays_the_reference_streams(oracle):
    """The oracle's C world on the worlds the reference generated: reset observation and every step's (obs, reward, done)."""
    z, names = _streams()
    worst = 0.0
    for n in names:
        speed, bounds, max_steps, hist = _cfg_of(z, n)
        cfg = oracle.default_config(width=float(bounds[0]), height=float(bounds[1]), dt=speed * 0.1, max_steps=max_steps, history=hist,
                                    n_traffic=3 if "_traffic/" in n else 0)
        w = oracle.World(cfg)
        acts, starts = z[n + "/actions"], list(z[n + "/episode_start"])
        polys, goals, reset_obs = z[n + "/polys"], z[n + "/goals"], z[n + "/reset_obs"]
        ep = -1
        for k, a in enumerate(acts):
            if ep + 1 < len(starts) and starts[ep + 1] == k:
                ep += 1
                o0 = w.reset(polys[ep][0], polys[ep][1], goals[ep])
                np.testing.assert_allclose(o0, reset_obs[ep], rtol=0, atol=1e-9, err_msg="%s reset %d" % (n, ep) + CAVEAT)
            o, r, d = w.step(int(a))
            err = float(np.max(np.abs(o - z[n + "/obs"][k])))
            assert err <= 1e-9, "%s step %d: obs differ by %g" % (n, k, err) + CAVEAT
            assert r == z[n + "/reward"][k] and d == bool(z[n + "/done"][k]), "%s step %d" % (n, k) + CAVEAT
            pk = w.peek()
            assert bool(pk["colliding"]) == bool(z[n + "/colliding"][k]) and bool(pk["goal_reached"]) == bool(z[n + "/goal_reached"][k])
            worst = max(worst, err)
    print("oracle vs executed reference Python: max |obs diff| = %.3e" % worst)


def test_host_worldgen_draws_the_worlds_the_reference_drew(native):
    """ship_sim_gym_amd.worldgen from the same (random, np.random) seeds: the river polygons and goal paths of every reset
    of every stream — i.e. the same RNG call order as ShipGame.__init__ / reset / gen_level / gen_goal_path executed."""
    from ship_sim_gym_amd import worldgen
    z, names = _streams()
    n_worlds = 0
    for n in names:
        speed, bounds, max_steps, hist = _cfg_of(z, n)
        seed = int(n.rsplit("seed", 1)[1])
        random.seed(seed)
        np.random.seed(seed)
        worldgen.generate_world(bounds)  # ShipGame.__init__ ends with reset(): the constructor's world
        for ep in range(len(z[n + "/polys"])):
            rec, polys, goals = worldgen.generate_world(bounds)
            np.testing.assert_array_equal(polys, z[n + "/polys"][ep], err_msg="%s world %d polygons" % (n, ep) + CAVEAT)
            np.testing.assert_allclose(goals, z[n + "/goals"][ep], rtol=0, atol=1e-9, err_msg="%s world %d goals" % (n, ep) + CAVEAT)
            n_worlds += 1
    assert n_worlds > 100


@pytest.mark.gpu
def test_hip_ship_env_facade_replays_the_reference_streams(native):
    """The HIP path behind the reference-shaped ShipEnv facade (map_mode 'fresh': worlds drawn from the global RNG streams
    in the reference's order), same seeds, same actions: observations within the north_star tolerance, rewards and done
    flags exact, colliding / goal_reached attributes exact."""
    from ship_sim_gym_amd.config import EnvConfig, GameConfig
    from ship_sim_gym_amd.ship_env import ShipEnv
    z, names = _streams()
    worst, n_steps = 0.0, 0
    for n in names:
        speed, bounds, max_steps, hist = _cfg_of(z, n)

        class G(GameConfig):
            SPEED = speed
            BOUNDS = bounds

        class E(EnvConfig):
            MAX_STEPS = max_steps
            HISTORY_SIZE = hist

        seed = int(n.rsplit("seed", 1)[1])
        random.seed(seed)
        np.random.seed(seed)
        env = ShipEnv(G, E, n_ships=4) if "_traffic/" in n else ShipEnv(G, E)
        acts, starts = z[n + "/actions"], list(z[n + "/episode_start"])
        ep = -1
        for k, a in enumerate(acts):
            if ep + 1 < len(starts) and starts[ep + 1] == k:
                ep += 1
                o0 = env.reset()
                env.game.add_default_traffic() if "_traffic/" in n else None  # (a no-op on the HIP path: n_ships=4 adds it at every reset)
                np.testing.assert_allclose(o0, z[n + "/reset_obs"][ep], rtol=0, atol=ATOL, err_msg="%s reset %d" % (n, ep) + CAVEAT)
            o, r, d, _ = env.step(int(a))
            err = float(np.max(np.abs(o - z[n + "/obs"][k])))
            assert err <= ATOL, "%s step %d: obs differ by %g" % (n, k, err) + CAVEAT
            assert r == z[n + "/reward"][k] and d == bool(z[n + "/done"][k]), "%s step %d" % (n, k) + CAVEAT
            assert env.game.colliding == bool(z[n + "/colliding"][k]) and env.game.goal_reached == bool(z[n + "/goal_reached"][k])
            worst = max(worst, err)
            n_steps += 1
        env.close()
    print("HIP facade vs executed reference Python: max |obs diff| = %.3e over %d steps" % (worst, n_steps))
