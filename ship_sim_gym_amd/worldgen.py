"""Reset-time world generation on the host: river banks + goal path, in the reference's RNG call order.

Reference: ShipGame.reset -> gen_level -> game_map.gen_river_poly (game.py:60-71,260-277) and gen_goal_path
(game.py:300-330).  Three RNG streams are involved: python ``random`` (bank vertices, goal y
jitter, fallback x jitter) and numpy's global RandomState (goal x).  The geometry the reference got from pymunk
(hulling by pm.Poly, fat segment queries by Space.segment_query) comes from libshipsim's host entry points.
"""
import ctypes as C
import random

import numpy as np

from . import _native as N
from . import game_map

Y_JITTER = 20      # game.py:314
X_JITTER = 50      # game.py:313


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def build_record(left, right, goals, spawn):
    """Pack one map-bank record (SSG_MAP_STRIDE doubles): hull planes, AABBs, goals, reset-obs goal."""
    left = np.ascontiguousarray(left, dtype=np.float64).reshape(-1, 2)
    right = np.ascontiguousarray(right, dtype=np.float64).reshape(-1, 2)
    goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 2)
    rec = np.zeros(N.MAP_STRIDE, dtype=np.float64)
    N.check(N.lib().ssg_host_build_map(_dp(left), len(left), _dp(right), len(right), _dp(goals), len(goals),
                                        float(spawn[0]), float(spawn[1]), _dp(rec)), None, "ssg_host_build_map")
    return rec


def goal_x_range(rec, width, y):
    lo, hi, hit = C.c_double(), C.c_double(), C.c_int()
    N.check(N.lib().ssg_host_goal_x_range(_dp(rec), float(width), float(y), C.byref(lo), C.byref(hi), C.byref(hit)),
            None, "ssg_host_goal_x_range")
    return bool(hit.value), lo.value, hi.value


def gen_goal_path(rec, bounds, n_goals, rng=random, np_rng=np.random):
    """gen_goal_path (game.py:300-330) against the hulls of ``rec``; returns [n_goals, 2] goal centres."""
    y_delta = bounds[1] / (n_goals + 1)
    x_middle = bounds[0] / 2
    goals = []
    for i in range(1, n_goals + 1):
        y = y_delta * i + rng.randint(-Y_JITTER, Y_JITTER)
        hit, lo, hi = goal_x_range(rec, bounds[0], y)
        if hit:
            x = np_rng.uniform(lo, hi)
        else:  # the reference's `except Exception` branch (empty hit list -> IndexError)
            x = x_middle * i + rng.randint(-X_JITTER, X_JITTER)
        goals.append([x, y])
    return np.asarray(goals, dtype=np.float64)


def generate_world(bounds, n_goals=5, width_frac=0.5, spawn=None, rng=random, np_rng=np.random):
    """One ShipGame.reset worth of world: (record, polys[2,12,2], goals[n,2]).  Consumes ``rng``/``np_rng`` exactly
    as the reference consumes ``random``/``np.random``."""
    if spawn is None:
        spawn = (bounds[0] / 2, 25)  # game.py:274
    polys = np.asarray(game_map.gen_river_poly(bounds, width_frac=width_frac, rng=rng), dtype=np.float64)
    bare = build_record(polys[0], polys[1], np.zeros((0, 2)), spawn)
    goals = gen_goal_path(bare, bounds, n_goals, rng=rng, np_rng=np_rng)
    return build_record(polys[0], polys[1], goals, spawn), polys, goals


def build_bank(n_maps, bounds, n_goals=5, width_frac=0.5, seed=1000, spawn=None):
    """The benchmark / training map bank of SURVEY.md §8d: map m is generated with ``random.seed(seed+m)`` and
    ``np.random.seed(seed+m)`` (private generator objects, the global streams are left untouched).
    Returns (records[n_maps, MAP_STRIDE], polys[n_maps,2,12,2], goals[n_maps,n_goals,2])."""
    recs, polys, goals = [], [], []
    for m in range(n_maps):
        r, p, g = generate_world(bounds, n_goals=n_goals, width_frac=width_frac, spawn=spawn,
                                 rng=random.Random(seed + m), np_rng=np.random.RandomState(seed + m))
        recs.append(r)
        polys.append(p)
        goals.append(g)
    return np.stack(recs), np.stack(polys), np.stack(goals)
