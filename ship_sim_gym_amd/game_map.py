"""River-bank polygon generator with the reference's RNG call sequence (ship_gym/game_map.py:22-73).

Draw order per bank vertex: ``random.gauss`` for x, then ``random.gauss`` for y, rejected (both redrawn) while x
falls outside [x_min, x_max], at most ``max_tries`` draws; the last draw is kept even if still invalid.  Note the
reference's ``x_middle = x_min + (x_max - x_min)`` is simply ``x_max`` (game_map.py:48), so bank vertices crowd
the outer edge of the left bank's range and the right edge of the map.  Seeding python ``random`` identically
therefore yields identical polygons (pinned by tests/golden/ref_maps.npz, generated from the real function).
"""
import random

N_SEGMENTS = 10
Y_START = -100


def _bank(n_segments, x_min, x_max, y_delta, rng, y_jitter=20, x_jitter=50, max_tries=1000):
    centre = x_min + (x_max - x_min)
    out = []
    for i in range(1, n_segments + 1):
        tries = 0
        while True:
            x = rng.gauss(centre, x_jitter)
            y = Y_START + rng.gauss(y_delta * i, y_jitter)
            tries += 1
            if x_min <= x <= x_max or tries >= max_tries:
                break
        out.append([x, y])
    return out


def gen_river_poly(bounds, N=N_SEGMENTS, width_frac=0.5, rng=random):
    """Two 12-vertex polygons [left, right]: N jittered points plus the two map corners of that side."""
    w, h = bounds[0], bounds[1]
    y_delta = (h * 1.2 - Y_START) / N
    bank_width = width_frac * w / 2
    left = _bank(N, 0, bank_width, y_delta, rng)
    left += [[0, h], [0, 0]]
    right = _bank(N, w - bank_width, w, y_delta, rng)
    right += [[w, h], [w, 0]]
    return [left, right]
