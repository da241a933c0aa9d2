"""Lesson counter with the reference's semantics (ship_gym/curriculum.py:23-50).

The reference never imports its Curriculum; BASELINE config 4 ("curriculum maps") is its only consumer, where it
selects which map bank (river width) the envs reset onto.  Behaviour reproduced, including the quirks:
``progress(val)`` needs MORE than ``repeat_condition`` successes (``>``, so repeat_condition=1 means two) and
``__float__`` returns ``values[lesson]`` unconverted (an int value makes ``float(c)`` raise TypeError).
"""


class Curriculum(object):
    def __init__(self, values, conditions, repeat_condition=1):
        self.values = values
        self.conditions = conditions
        self.repeat_condition = repeat_condition
        self.lesson = 0
        self.repeat_reached = 0

    def __float__(self):
        return self.values[self.lesson]

    def __int__(self):
        return int(self.values[self.lesson])

    @property
    def value(self):
        return self.values[self.lesson]

    def progress(self, val):
        """Advance when ``val`` beat the current lesson's threshold often enough; True iff the lesson changed."""
        if self.lesson >= len(self.conditions):
            return False
        if not (val > self.conditions[self.lesson]):
            return False
        self.repeat_reached += 1
        if self.repeat_reached > self.repeat_condition:
            self.lesson += 1
            self.repeat_reached = 0
            return True
        return False


class CurriculumMaps(object):
    """BASELINE config 4's "curriculum maps": one map bank per lesson, the lesson's value being the river-bank
    `width_frac` handed to gen_river_poly (game_map.py:22; wider banks = narrower river).

    `progress(val)` feeds the reference's Curriculum; when the lesson changes the new lesson's bank is installed in
    the env (on every rank: rank 0's bank is broadcast over RCCL) and every env is reset onto it, since a reference
    env only meets a new width at its next reset()."""

    def __init__(self, vec, widths=(0.5, 0.6, 0.7), conditions=(0.0, 1.0), repeat_condition=1, n_maps=None, seed=1000):
        from . import worldgen
        self.vec = vec
        self.curriculum = Curriculum(list(widths), list(conditions), repeat_condition=repeat_condition)
        n_maps = n_maps if n_maps is not None else vec.n_maps
        self.banks = [worldgen.build_bank(n_maps, vec.bounds, n_goals=vec.cfg.n_goals, width_frac=w, seed=seed + 100000 * i)
                      for i, w in enumerate(widths)]
        self._install()

    def _install(self):
        from . import sharding
        recs, polys, goals = self.banks[self.curriculum.lesson]
        self.vec.bank_polys, self.vec.bank_goals = polys, goals
        self.vec.set_bank(recs)
        sharding.broadcast_bank(self.vec, src=0)

    @property
    def width_frac(self):
        return float(self.curriculum)

    def progress(self, val):
        """Returns the reset observations (device tensor) when the lesson advanced, else None."""
        if self.curriculum.progress(val):
            self._install()
            return self.vec.reset_tensor()
        return None
