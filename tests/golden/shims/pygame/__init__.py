"""pygame stand-in, test infrastructure only (see ../README.md): a window that draws nothing and a clock that never sleeps."""
QUIT, KEYDOWN = 12, 2
K_ESCAPE, K_q, K_w, K_s, K_a, K_d = 27, 113, 119, 115, 97, 100


class _Surface(object):
    def __init__(self, size):
        self.size = size

    def fill(self, color):
        pass


class _Display(object):
    @staticmethod
    def set_mode(size, *a, **k):
        return _Surface(size)

    @staticmethod
    def set_caption(title):
        pass

    @staticmethod
    def flip():
        pass


class _Clock(object):
    def tick(self, fps=0):
        return 0


class _Time(object):
    Clock = _Clock


class _Key(object):
    @staticmethod
    def set_repeat(*a):
        pass


class _Event(object):
    @staticmethod
    def get():
        return []


class _Color(object):
    THECOLORS = {"white": (255, 255, 255, 255), "black": (0, 0, 0, 255), "green": (0, 255, 0, 255), "red": (255, 0, 0, 255)}


class _Draw(object):
    @staticmethod
    def circle(*a, **k):
        pass


class _Surfarray(object):
    @staticmethod
    def array3d(surface):
        raise NotImplementedError


display, time, key, event, color, draw, surfarray = _Display, _Time, _Key, _Event, _Color, _Draw, _Surfarray


def init():
    return (0, 0)
