"""pymunk.pygame_util stand-in (only named by GameConfig.DEBUG drawing, which the goldens never enable)."""


class DrawOptions(object):
    def __init__(self, surface):
        self.flags = 0
