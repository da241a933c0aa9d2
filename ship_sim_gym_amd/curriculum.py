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
