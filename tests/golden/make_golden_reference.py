#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference modules that import in the build container.

Only three reference modules import without pymunk/pygame/gym:
``ship_gym/game_map.py``, ``ship_gym/config.py`` and ``ship_gym/curriculum.py``.  This script
imports them *from /root/reference* (never copied into this repo), drives them with fixed seeds
and writes inputs + expected outputs as data fixtures:

* ``ref_maps.npz``        — ``gen_river_poly`` polygons (game_map.py:22-73) for seeds 0..15,
                            bounds 600^2 / 1000^2, width_frac 0.5/0.6/0.7, plus the next
                            ``random.random()`` draw after each call (pins RNG consumption).
* ``ref_config.json``     — the class-attribute defaults of config.py:8-24.
* ``ref_curriculum.json`` — ``Curriculum.progress`` traces (curriculum.py:23-50).

Run in the build container only:  python tests/golden/make_golden_reference.py
The GPU box never has /root/reference; tests read the fixtures, not this script's imports.
"""
import importlib.util
import json
import os
import random

import numpy as np

REF = "/root/reference/ship_gym"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("_ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    game_map = _load("game_map")
    config = _load("config")
    curriculum = _load("curriculum")

    # ---- maps -------------------------------------------------------------------------------
    seeds, bounds_l, fracs, polys, nxt = [], [], [], [], []
    for bounds in ((600, 600), (1000, 1000)):
        for frac in (0.5, 0.6, 0.7):
            for seed in range(16):
                random.seed(seed)
                if frac == 0.5:
                    p = game_map.gen_river_poly(bounds)  # default-argument call, as game.py:66
                else:
                    p = game_map.gen_river_poly(bounds, width_frac=frac)
                assert len(p) == 2 and len(p[0]) == 12 and len(p[1]) == 12
                seeds.append(seed)
                bounds_l.append(bounds)
                fracs.append(frac)
                polys.append(np.asarray(p, dtype=np.float64))
                nxt.append(random.random())
    # two back-to-back calls on one stream (construction + reset consume RNG twice, App. B-17)
    random.seed(1234)
    chain = [np.asarray(game_map.gen_river_poly((600, 600)), dtype=np.float64) for _ in range(3)]
    np.savez_compressed(
        os.path.join(HERE, "ref_maps.npz"),
        seeds=np.asarray(seeds, dtype=np.int64),
        bounds=np.asarray(bounds_l, dtype=np.float64),
        width_frac=np.asarray(fracs, dtype=np.float64),
        polys=np.stack(polys),  # [case, bank(2), vertex(12), xy(2)]
        next_random=np.asarray(nxt, dtype=np.float64),
        chain_seed=np.int64(1234),
        chain=np.stack(chain),
    )

    # ---- config -----------------------------------------------------------------------------
    cfg = {
        "LidarConfig": {k: getattr(config.LidarConfig, k) for k in ("N_BEAMS", "DISTANCE", "ANGULAR_SPREAD")},
        "EnvConfig": {k: getattr(config.EnvConfig, k) for k in ("HISTORY_SIZE", "MAX_STEPS")},
        "GameConfig": {
            "DEBUG": config.GameConfig.DEBUG,
            "FPS": config.GameConfig.FPS,
            "SPEED": config.GameConfig.SPEED,
            "BOUNDS": list(config.GameConfig.BOUNDS),
        },
        "EnvConfig.LIDAR_CONFIG_is_LidarConfig": config.EnvConfig.LIDAR_CONFIG is config.LidarConfig,
    }
    with open(os.path.join(HERE, "ref_config.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)

    # ---- curriculum -------------------------------------------------------------------------
    traces = []
    rng = np.random.RandomState(7)
    for values, conditions, repeat in (
        ([0.5, 0.6, 0.7], [0.0, 1.0], 1),
        ([1, 2, 3, 4], [10, 20, 30], 0),
        ([3.5, 2.5], [5.0], 3),
    ):
        c = curriculum.Curriculum(values, conditions, repeat_condition=repeat)
        vals = [float(v) for v in rng.uniform(-5, 40, size=40)]
        steps = []
        for v in vals:
            ret = c.progress(v)
            try:  # __float__ hands back values[lesson] unconverted (curriculum.py:37-38): ints raise
                as_float = float(c)
            except TypeError:
                as_float = "TypeError"
            steps.append({"val": v, "ret": bool(ret), "lesson": c.lesson, "repeat_reached": c.repeat_reached,
                          "as_float": as_float, "as_int": int(c)})
        traces.append({"values": values, "conditions": conditions, "repeat_condition": repeat, "steps": steps})
    with open(os.path.join(HERE, "ref_curriculum.json"), "w") as f:
        json.dump(traces, f, indent=1)

    print("wrote ref_maps.npz (%d cases), ref_config.json, ref_curriculum.json" % len(seeds))


if __name__ == "__main__":
    main()
