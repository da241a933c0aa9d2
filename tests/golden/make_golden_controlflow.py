#!/usr/bin/env python3
"""Generates tests/golden/ref_controlflow.npz by EXECUTING the reference's own Python — ship_gym/ship_env.py, game.py and
models.py, imported unmodified from /root/reference — under the test-only stand-ins of tests/golden/shims (pymunk / pygame
/ gym; physics primitives answered by the CPU oracle).  Build container only: /root/reference does not exist on the GPU
box, and only the resulting DATA file is committed.

What the streams pin (by execution of the real reference code, not by reading it): the Python layer —
ShipEnv.step / reset / determine_reward / is_done / __add_states (ship_env.py:62-156,171-184), ShipGame.reset /
gen_goal_path incl. its RNG call order and `except` fallback / handle_discrete_action / update / closest_goal and the
collision callbacks (game.py:140-153,185-195,232-349), LiDAR.query and Ship (models.py:39-76,87-146).  What they do
NOT pin: Chipmunk2D's arithmetic, which under the stand-in is the oracle's own restatement ("parity unpinned").

Scenarios: the eight of tools/capture_pymunk_golden.py and two with HISTORY_SIZE 1 / 3, x seeds 0..3 (random.seed(s);
np.random.seed(s)) without traffic,
and six of them x seeds 0..1 with `env.game.add_default_traffic()` after every reset (BASELINE configs[3]).
Per stream: the worlds the reference generated (river polygons handed to PolyEnv, goal centres), the reset observation
of every episode, and (obs, reward, done) of every step; a done step is followed by env.reset() as a trainer would.
"""
import io
import os
import random
import sys
from contextlib import redirect_stdout

import numpy as np

SCOPE = ("scope: PYTHON LAYER ONLY. Streams recorded by executing the reference's ship_env.py / game.py / models.py under stand-in pymunk / pygame / gym modules whose physics IS this repository's CPU oracle (oracle/ssg_oracle.c, ssg_dynamics.c). They pin action decoding, reward / done order, the history deque, lidar stickiness, RNG call order. They are CIRCULAR for every Chipmunk2D computation (integrator, segment queries, narrowphase, the config-4 contact solver) and are NOT Chipmunk parity evidence.")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("SHIP_SIM_GYM", "/root/reference")


def scenarios():
    fwd = [0] * 60
    yield "all_forward", {}, fwd
    yield "left_then_forward", {}, [1, 1] + fwd
    yield "right_clamp", {}, [2] * 5 + fwd
    yield "into_left_bank", {}, [1, 1] + [0] * 120
    yield "rudder_only_max_steps", dict(max_steps=150), [1, 2] * 80
    yield "zigzag", {}, ([1] * 3 + [0] * 6 + [2] * 6 + [0] * 6) * 8
    rng = np.random.RandomState(6)
    yield "random_actions", {}, [int(a) for a in rng.randint(0, 3, size=400)]
    yield "training_config", dict(speed=30, bounds=(1000, 1000)), fwd + [1, 0, 0, 2, 0, 0] * 10
    # EnvConfig.HISTORY_SIZE other than the default 2: the deque of ship_env.py:50,100-113 (history of -1 after a reset)
    yield "history_1", dict(history=1), ([1] * 2 + [0] * 7 + [2] * 4 + [0] * 7) * 6
    yield "history_3", dict(history=3), [2, 2] + [0] * 100


def run(ShipEnv, GameConfig, EnvConfig, opts, actions, seed, traffic=False):
    class G(GameConfig):
        DEBUG = False
        FPS = 100000
        SPEED = opts.get("speed", 10)
        BOUNDS = opts.get("bounds", (600, 600))

    class E(EnvConfig):
        MAX_STEPS = opts.get("max_steps", EnvConfig.MAX_STEPS)
        HISTORY_SIZE = opts.get("history", EnvConfig.HISTORY_SIZE)

    random.seed(seed)
    np.random.seed(seed)
    with redirect_stdout(io.StringIO()):
        env = ShipEnv(G, E)  # ShipGame.__init__ ends with reset(): one world's worth of RNG is consumed here (App. B-17)
    out = {"obs": [], "reward": [], "done": [], "polys": [], "goals": [], "reset_obs": [], "episode_start": [], "colliding": [],
           "goal_reached": []}

    def do_reset():
        o = env.reset()
        if traffic:  # config 4: the reference expects its caller to add the traffic after every reset (game.py:279-286)
            env.game.add_default_traffic()
        out["reset_obs"].append(np.asarray(o, dtype=np.float64))
        out["episode_start"].append(len(out["obs"]))
        out["polys"].append(np.asarray(env.game.level.poly_list, dtype=np.float64))  # the raw 12-gons handed to PolyEnv
        out["goals"].append(np.asarray([[g.body.position.x, g.body.position.y] for g in env.game.goals], dtype=np.float64))

    do_reset()
    for a in actions:
        o, r, d, _ = env.step(a)
        out["obs"].append(np.asarray(o, dtype=np.float64))
        out["reward"].append(float(r))
        out["done"].append(bool(d))
        out["colliding"].append(bool(env.game.colliding))
        out["goal_reached"].append(bool(env.game.goal_reached))
        if d:
            do_reset()
    res = {k: np.asarray(v) for k, v in out.items()}
    res["actions"] = np.asarray(actions, dtype=np.int32)
    res["config"] = np.asarray([G.SPEED, G.BOUNDS[0], G.BOUNDS[1], E.MAX_STEPS, E.HISTORY_SIZE], dtype=np.float64)
    return res


def main():
    sys.path[:] = [p for p in sys.path if os.path.realpath(p or ".") != os.path.realpath(ROOT)]  # the repo has a `ship_gym` alias
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REFERENCE)
    sys.path.append(ROOT)
    import pymunk
    assert "standin" in pymunk.version, "the genuine pymunk is installed: use tools/capture_pymunk_golden.py instead"
    from ship_gym.ship_env import ShipEnv
    from ship_gym.config import EnvConfig, GameConfig
    import ship_gym.ship_env as se
    assert os.path.realpath(se.__file__).startswith(os.path.realpath(REFERENCE)), se.__file__
    data = {}
    n_steps = n_done = 0
    for name, opts, actions in scenarios():
        for seed in range(4):
            r = run(ShipEnv, GameConfig, EnvConfig, opts, actions, seed)
            for k, v in r.items():
                data["%s/seed%d/%s" % (name, seed, k)] = v
            n_steps += len(actions)
            n_done += int(r["done"].sum())
    # config 4: the same Python with env.game.add_default_traffic() after every reset; the stand-in's Space.step then runs the
    # oracle's full cpSpaceStep (contact solver for the traffic ships and goal bodies) on a shadow world
    for name, opts, actions in scenarios():
        if name in ("rudder_only_max_steps", "training_config", "history_1", "history_3"):
            continue
        for seed in range(2):
            r = run(ShipEnv, GameConfig, EnvConfig, opts, actions, seed, traffic=True)
            for k, v in r.items():
                data["%s_traffic/seed%d/%s" % (name, seed, k)] = v
            n_steps += len(actions)
            n_done += int(r["done"].sum())
    out = os.path.join(HERE, "ref_controlflow.npz")
    n_arrays = len(data)
    data["__scope__"] = np.array(SCOPE)  # what these streams do and do not pin, carried inside the fixture
    np.savez_compressed(out, **data)
    print("wrote %s: %d streams, %d steps, %d episode ends, %d bytes" % (out, n_arrays // 11, n_steps, n_done, os.path.getsize(out)))


if __name__ == "__main__":
    main()
