#!/usr/bin/env python3
"""Capture golden step streams from the GENUINE reference (pymunk 5.4.0 / Chipmunk2D, pygame, gym) — the missing pin.

The oracle (oracle/) restates Chipmunk2D from its published behaviour because pymunk is neither under the reference
tree nor installed in the build image ("PARITY UNPINNED", oracle/ssg_oracle.h).  On any machine where

    pip install pymunk==5.4.0 pygame==1.9.4 gym==0.10.9 numpy      (requirements.txt:29,59,75,78 of the reference)

works, this script runs the reference's ShipEnv UNMODIFIED and writes tests/golden/pymunk_streams.npz:

    python tools/capture_pymunk_golden.py --reference /path/to/ship-sim-gym

tests/test_pymunk_golden.py then replays every stream on the oracle (and, with a GPU, on the HIP path through the
fresh-mode ShipEnv facade) and requires |obs - golden| <= 1e-5, rewards and done flags exact.  Any failure localises
to one of the named assumptions (oracle/ssg_dynamics.c ORDER / GJK-ID / PLAYER).

Scenarios (SURVEY.md §8c "G2"): all-forward, left-then-forward, right clamp, drive into the left bank, onto the first
goal, rudder-only to MAX_STEPS, out-of-bounds exit, the training config (SPEED 30, BOUNDS 1000), and for config 4 the
same streams after `env.game.add_default_traffic()` (which the reference expects its caller to invoke after reset).
Only data is written: seeds, action lists, observations, rewards, done flags, and the world the reference generated
(river polygons, goal positions) so that a replay does not depend on reproducing the RNG streams.
"""
import argparse
import os
import random
import sys

import numpy as np


def scenarios():
    fwd = [0] * 60
    yield "all_forward", dict(seed=0), fwd
    yield "left_then_forward", dict(seed=1), [1, 1] + fwd
    yield "right_clamp", dict(seed=2), [2] * 5 + fwd
    yield "into_left_bank", dict(seed=3), [1, 1] + [0] * 120
    yield "rudder_only_max_steps", dict(seed=4), [1, 2] * 520
    yield "zigzag", dict(seed=5), ([1] * 3 + [0] * 6 + [2] * 6 + [0] * 6) * 8
    rng = np.random.RandomState(6)
    yield "random_actions", dict(seed=6), [int(a) for a in rng.randint(0, 3, size=400)]
    yield "training_config", dict(seed=7, speed=30, bounds=(1000, 1000)), fwd + [1, 0, 0, 2, 0, 0] * 10


def run(ref_env_cls, game_cfg_cls, env_cfg_cls, name, opts, actions, traffic):
    class G(game_cfg_cls):
        DEBUG = False
        FPS = 100000
        SPEED = opts.get("speed", 10)
        BOUNDS = opts.get("bounds", (600, 600))
    seed = opts["seed"]
    random.seed(seed)
    np.random.seed(seed)
    env = ref_env_cls(G, env_cfg_cls)      # ShipGame.__init__ ends with reset(): consumes one world (App. B-17)
    out = {"obs": [], "reward": [], "done": [], "polys": [], "goals": [], "reset_obs": [], "episode_start": []}
    # (the world has to be read BEFORE traffic can disturb the goals: positions right after ShipGame.reset)

    def do_reset():
        o = env.reset()
        if traffic:
            env.game.add_default_traffic()
        out["reset_obs"].append(np.asarray(o, dtype=np.float64))
        out["episode_start"].append(len(out["obs"]))
        # the raw 12-vertex river polygons handed to PolyEnv (models.py:163; pm.Poly hulls them, as the oracle does)
        out["polys"].append(np.asarray(env.game.level.poly_list, dtype=np.float64))
        out["goals"].append(np.asarray([[g.body.position.x, g.body.position.y] for g in env.game.goals], dtype=np.float64))

    do_reset()
    for a in actions:
        o, r, d, _ = env.step(a)
        out["obs"].append(np.asarray(o, dtype=np.float64))
        out["reward"].append(float(r))
        out["done"].append(bool(d))
        if d:
            do_reset()
    return {"%s%s/%s" % (name, "_traffic" if traffic else "", k): np.asarray(v)
            for k, v in out.items()} | {"%s%s/actions" % (name, "_traffic" if traffic else ""): np.asarray(actions, dtype=np.int32),
                                        "%s%s/seed" % (name, "_traffic" if traffic else ""): np.asarray(seed)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("SHIP_SIM_GYM", "/root/reference"))
    default_out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pymunk_streams.npz")
    ap.add_argument("--out", default=default_out)
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise THIS SCRIPT, not the reference's physics: run every scenario (first --dry-steps actions) under the "
                         "test-only stand-in pymunk / pygame / gym of tests/golden/shims (physics = the CPU oracle) and write to "
                         "--out, which must not be the golden path.  The output is NOT a capture and pins nothing.")
    ap.add_argument("--dry-steps", type=int, default=40)
    args = ap.parse_args()
    os.environ.setdefault("SDL_VIDEODRIVER", "dummy")
    if args.dry_run:
        if os.path.abspath(args.out) == os.path.abspath(default_out):
            sys.exit("--dry-run never writes the golden file: pass --out elsewhere")
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        sys.path.insert(0, os.path.join(root, "tests", "golden", "shims"))
        sys.path.insert(0, root)
    try:
        import pymunk  # noqa: F401
        import pygame  # noqa: F401
        import gym  # noqa: F401
    except ImportError as e:
        sys.exit("the genuine reference cannot run here: %s (see the module docstring)" % e)
    sys.path.insert(0, args.reference)
    from ship_gym.ship_env import ShipEnv
    from ship_gym.config import EnvConfig, GameConfig
    import pymunk as pm
    standin = "standin" in str(pm.version)
    if standin and not args.dry_run:
        sys.exit("the pymunk on sys.path is the test-only stand-in (physics = the oracle): refusing to write a 'capture' from it")
    data = {"meta/pymunk_version": np.asarray(pm.version), "meta/chipmunk_version": np.asarray(pm.chipmunk_version),
            "meta/dry_run": np.asarray(bool(args.dry_run))}
    for traffic in (False, True):
        for name, opts, actions in scenarios():
            if args.dry_run:
                actions = actions[:max(1, args.dry_steps)]
            data.update(run(ShipEnv, GameConfig, EnvConfig, name, opts, actions, traffic))
    np.savez_compressed(args.out, **data)
    print("%swrote %s with %d arrays (pymunk %s)" % ("DRY RUN, NOT A CAPTURE: " if args.dry_run else "", args.out, len(data), pm.version))


if __name__ == "__main__":
    main()
