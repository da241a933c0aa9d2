"""Stand-in for pymunk 5.4.0, test infrastructure only (see ../README.md): the subset of the API that the reference's
ship_gym/{models,game}.py call, answered by the CPU oracle's restatement of Chipmunk2D through ctypes.

Semantics stated from memory of pymunk 5.4.0 / Chipmunk 7.0.x (unverifiable here):
  * Shape.bb is the cached AABB: zeros until the shape is added to a space (cpSpaceAddShape updates it) or a step ran;
  * Space.step: cpBodyUpdatePosition for every dynamic body, shape caches, collision detection with `begin`
    callbacks on first touch, cpBodyUpdateVelocity (damping^dt), forces cleared;
  * a `begin` callback returning False suppresses the pair's response; space.remove inside a callback is deferred;
  * the PLAYER assumption of the oracle: contact response of the player is not simulated (every such contact ends
    the episode).
  * with traffic ships in the space (ShipGame.add_default_traffic, config 4) Space.step hands the whole scene to a shadow
    oracle world — the reference's Python still drives everything (actions, lidar, reward, observation, resets, the
    callbacks), the oracle's restated cpSpaceStep incl. its contact solver moves the traffic ships and goal bodies.
"""
import ctypes as C
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)  # (appended: the repo's own `ship_gym` alias package must not shadow the reference's)
from oracle import oracle as _O  # noqa: E402

version = "5.4.0-standin"
chipmunk_version = "oracle restatement"


class Vec2d(object):
    __slots__ = ("x", "y")

    def __init__(self, x=0.0, y=None):
        if y is None:
            x, y = x[0], x[1]
        self.x, self.y = x, y

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __len__(self):
        return 2

    def __iter__(self):
        return iter((self.x, self.y))

    def __add__(self, o):
        return Vec2d(self.x + o[0], self.y + o[1])

    def __sub__(self, o):
        return Vec2d(self.x - o[0], self.y - o[1])

    def __mul__(self, s):
        return Vec2d(self.x * s, self.y * s)

    __rmul__ = __mul__

    def __neg__(self):
        return Vec2d(-self.x, -self.y)

    def get_distance(self, other):
        return math.sqrt((self.x - other[0]) ** 2 + (self.y - other[1]) ** 2)

    def __repr__(self):
        return "Vec2d(%r, %r)" % (self.x, self.y)


def _v2(p):
    return _O.V2(float(p[0]), float(p[1]))


def moment_for_poly(mass, vertices, offset=(0, 0), radius=0):
    return _O.moment_for_poly(mass, [(float(x), float(y)) for x, y in vertices])


def moment_for_circle(mass, inner_radius, outer_radius, offset=(0, 0)):
    return mass * (0.5 * (inner_radius * inner_radius + outer_radius * outer_radius) + (offset[0] ** 2 + offset[1] ** 2))


class Transform(object):
    @staticmethod
    def identity():
        return Transform()


class ShapeFilter(object):
    ALL_MASKS = 0xFFFFFFFF
    ALL_CATEGORIES = 0xFFFFFFFF

    def __init__(self, group=0, categories=0xFFFFFFFF, mask=0xFFFFFFFF):
        self.group, self.categories, self.mask = group, categories, mask

    def rejects(self, other):  # cpShapeFilterReject
        return ((self.group != 0 and self.group == other.group) or (self.categories & other.mask) == 0 or
                (other.categories & self.mask) == 0)


class SpaceDebugDrawOptions(object):
    DRAW_SHAPES = 1


class Body(object):
    DYNAMIC, KINEMATIC, STATIC = 0, 1, 2

    def __init__(self, mass=0, moment=0, body_type=0):
        self.body_type = body_type
        self._b = _O.Body()
        self._b.rot = _O.V2(1.0, 0.0)
        if body_type == Body.DYNAMIC:
            self._b.m_inv = 1.0 / mass
            self._b.i_inv = 1.0 / moment
        self.mass, self.moment = mass, moment
        self.center_of_gravity = Vec2d(0.0, 0.0)
        self.shapes = []
        self.space = None

    @property
    def position(self):
        return Vec2d(self._b.p.x, self._b.p.y)

    @position.setter
    def position(self, p):
        self._b.p = _v2(p)

    @property
    def angle(self):
        return self._b.a

    @property
    def velocity(self):
        return Vec2d(self._b.v.x, self._b.v.y)

    @property
    def angular_velocity(self):
        return self._b.w

    def apply_force_at_local_point(self, force, point):
        _O.lib().ora_body_apply_force_at_local_point(C.byref(self._b), _v2(force), _v2(point))


class BB(object):
    def __init__(self, left=0.0, bottom=0.0, right=0.0, top=0.0):
        self.left, self.bottom, self.right, self.top = left, bottom, right, top

    def center(self):  # cpBBCenter = cpvlerp((l, b), (r, t), 0.5)
        return Vec2d(self.left * 0.5 + self.right * 0.5, self.bottom * 0.5 + self.top * 0.5)

    def merge(self, o):
        return BB(min(self.left, o.left), min(self.bottom, o.bottom), max(self.right, o.right), max(self.top, o.top))


class SegmentQueryInfo(object):
    def __init__(self, shape, point, normal, alpha):
        self.shape, self.point, self.normal, self.alpha = shape, point, normal, alpha


class Shape(object):
    def __init__(self, body):
        self.body = body
        self.friction, self.elasticity, self.color = 0.0, 0.0, None
        self.collision_type = 0
        self.filter = ShapeFilter()
        self.space = None
        if body is not None:
            body.shapes.append(self)


class Poly(Shape):
    def __init__(self, body, vertices, transform=None, radius=0):
        Shape.__init__(self, body)
        self._p = _O.Poly()
        v = [(float(x), float(y)) for x, y in vertices]
        self._verts = v
        import numpy as np
        a = np.ascontiguousarray(v, dtype=np.float64)
        _O.lib().ora_poly_init(C.byref(self._p), len(v), a.ctypes.data_as(C.POINTER(C.c_double)))  # hulls its input

    def _cache(self):  # cpPolyShapeCacheData
        b = self.body._b
        _O.lib().ora_poly_update(C.byref(self._p), b.p, b.rot)

    @property
    def bb(self):
        return BB(self._p.bb_l, self._p.bb_b, self._p.bb_r, self._p.bb_t)

    def update(self, transform):
        self._cache()
        return self.bb

    def _segment(self, start, end, radius):
        info = _O.SegInfo()
        hit = _O.lib().ora_poly_segment_query(C.byref(self._p), _v2(start), _v2(end), float(radius), C.byref(info))
        return hit, info

    def segment_query(self, start, end, radius=0):
        hit, info = self._segment(start, end, radius)
        return SegmentQueryInfo(self if hit else None, Vec2d(info.point.x, info.point.y),
                                Vec2d(info.normal.x, info.normal.y), info.alpha)


class Circle(Shape):
    def __init__(self, body, radius, offset=(0, 0)):
        Shape.__init__(self, body)
        self.radius, self.offset = float(radius), Vec2d(offset)

    def _center(self):
        b = self.body._b
        return _O.V2(b.rot.x * self.offset.x - b.rot.y * self.offset.y + b.p.x, b.rot.y * self.offset.x + b.rot.x * self.offset.y + b.p.y)

    def _cache(self):
        pass

    @property
    def bb(self):
        c = self._center()
        return BB(c.x - self.radius, c.y - self.radius, c.x + self.radius, c.y + self.radius)

    def _segment(self, start, end, radius):
        info = _O.SegInfo()
        hit = _O.lib().ora_circle_segment_query(self._center(), self.radius, _v2(start), _v2(end), float(radius), C.byref(info))
        return hit, info


class Arbiter(object):
    def __init__(self, a, b):
        self.shapes = (a, b)


class CollisionHandler(object):
    def __init__(self):
        self.begin = None


class Space(object):
    def __init__(self):
        self.damping = 1.0
        self.gravity = Vec2d(0, 0)
        self.bodies, self.shapes = [], []
        self._handlers = {}
        self._touching = set()
        self._in_step = False
        self._deferred = []

    def add(self, *objs):
        for o in objs:
            if isinstance(o, Body):
                if o.body_type == Body.DYNAMIC:
                    self.bodies.append(o)
                o.space = self
            else:
                self.shapes.append(o)
                o.space = self
                o._cache()  # cpSpaceAddShape: cpShapeUpdate

    def remove(self, *objs):
        if self._in_step:  # pymunk defers removals requested from inside a callback to the end of the step
            self._deferred.append(objs)
            return
        for o in objs:
            if isinstance(o, Body):
                if o in self.bodies:
                    self.bodies.remove(o)
            elif o in self.shapes:
                self.shapes.remove(o)

    def add_collision_handler(self, type_a, type_b):
        return self._handlers.setdefault((type_a, type_b), CollisionHandler())

    def segment_query(self, start, end, radius, shape_filter):
        out = []
        for s in self.shapes:
            if s.filter.rejects(shape_filter):
                continue
            hit, info = s._segment(start, end, radius)
            if hit:
                out.append(SegmentQueryInfo(s, Vec2d(info.point.x, info.point.y), Vec2d(info.normal.x, info.normal.y), info.alpha))
        return out

    @staticmethod
    def _touch(a, b):
        """Chipmunk reports a contact iff the shapes' closed sets intersect (after the cpBBIntersects reject): the oracle's
        narrowphase predicates."""
        L = _O.lib()
        if isinstance(a, Poly) and isinstance(b, Poly):
            return bool(L.ora_polys_collide(C.byref(a._p), C.byref(b._p)))
        if isinstance(a, Poly) and isinstance(b, Circle):
            return bool(L.ora_circle_poly_collide(b._center(), b.radius, C.byref(a._p)))
        if isinstance(a, Circle) and isinstance(b, Poly):
            return bool(L.ora_circle_poly_collide(a._center(), a.radius, C.byref(b._p)))
        raise NotImplementedError("circle-circle handlers are not on the reference's path")

    def step(self, dt):
        L = _O.lib()
        if any(isinstance(s, Poly) and s.collision_type == 1 and s.body.body_type == Body.DYNAMIC for s in self.shapes):
            return self._step_with_traffic(dt)
        for b in self.bodies:            # cpBodyUpdatePosition
            L.ora_body_update_position(C.byref(b._b), float(dt))
        for s in self.shapes:            # cpShapeUpdateFunc (static shapes keep their cache)
            if s.body.body_type == Body.DYNAMIC:
                s._cache()
        self._in_step = True
        now = set()
        for (ta, tb), h in list(self._handlers.items()):
            for a in list(self.shapes):
                if a.collision_type != ta:
                    continue
                for b in list(self.shapes):
                    if b is a or b.collision_type != tb:
                        continue
                    if not self._touch(a, b):
                        continue
                    key = (id(a), id(b))
                    now.add(key)
                    if key not in self._touching and h.begin is not None:
                        h.begin(Arbiter(a, b), self, None)
        self._touching = now
        self._in_step = False
        for objs in self._deferred:
            self.remove(*objs)
        self._deferred = []
        damping = math.pow(self.damping, dt)
        for b in self.bodies:            # cpBodyUpdateVelocity, forces cleared
            L.ora_body_update_velocity(C.byref(b._b), damping, float(dt))

    def debug_draw(self, options):
        pass

    # ---- config 4: traffic ships in the space -> the oracle's full cpSpaceStep on a shadow world ----
    def _build_shadow(self, dt):
        import numpy as np
        banks = [s for s in self.shapes if isinstance(s, Poly) and s.body.body_type == Body.STATIC]
        goals = [s for s in self.shapes if isinstance(s, Circle)]
        player = [s for s in self.shapes if isinstance(s, Poly) and s.collision_type == 0]
        traffic = [s for s in self.shapes if isinstance(s, Poly) and s.collision_type == 1 and s.body.body_type == Body.DYNAMIC]
        assert len(banks) == 2 and len(player) == 1 and len(traffic) == 3 and all(len(b._verts) == 12 for b in banks)
        cfg = _O.default_config(dt=float(dt), space_damping=float(self.damping), n_goals=len(goals), goal_radius=goals[0].radius,
                                n_traffic=3)
        w = _O.World(cfg)
        w.reset(np.asarray(banks[0]._verts), np.asarray(banks[1]._verts), np.asarray([[g.body._b.p.x, g.body._b.p.y] for g in goals]))
        dyn = w.peek_dyn()
        for k, t in enumerate(traffic):  # the oracle's add_default_traffic must be the reference's (game.py:279-286)
            assert (t.body._b.p.x, t.body._b.p.y) == (dyn["traffic"][k, 0], dyn["traffic"][k, 1]), "traffic ship %d" % k
            assert abs(1.0 / t.body._b.i_inv - t.body.moment) < 1e-9
        self._shadow = (w, banks, goals, player[0], traffic, (1 << len(goals)) - 1)

    def _step_with_traffic(self, dt):
        L = _O.lib()
        if not hasattr(self, "_shadow"):
            self._build_shadow(dt)
        w, banks, goals, player, traffic, alive0 = self._shadow
        L.ora_world_set_ship(w._p, C.byref(player.body._b))
        L.ora_world_space_step(w._p)
        L.ora_world_get_ship(w._p, C.byref(player.body._b))
        player._cache()
        pk, dyn = w.peek(), w.peek_dyn()
        for k, t in enumerate(traffic):
            b = t.body._b
            b.p, b.a = _O.V2(dyn["traffic"][k, 0], dyn["traffic"][k, 1]), dyn["traffic"][k, 2]
            b.rot = _O.V2(math.cos(b.a), math.sin(b.a))
            b.v, b.w = _O.V2(dyn["traffic"][k, 3], dyn["traffic"][k, 4]), dyn["traffic"][k, 5]
            t._cache()
        for g, s in enumerate(goals):
            s.body._b.p = _O.V2(dyn["goals"][g, 0], dyn["goals"][g, 1])
            s.body._b.v = _O.V2(dyn["goals"][g, 2], dyn["goals"][g, 3])
        # the begin callbacks, as cpSpaceStep would have fired them
        self._in_step = True
        if pk["colliding"] and (0, 1) in self._handlers and self._handlers[(0, 1)].begin is not None:
            self._handlers[(0, 1)].begin(Arbiter(player, banks[0]), self, None)
        alive = int(pk["alive_mask"])
        for g, s in enumerate(goals):
            if (alive0 >> g) & 1 and not (alive >> g) & 1 and (0, 2) in self._handlers and self._handlers[(0, 2)].begin is not None:
                self._handlers[(0, 2)].begin(Arbiter(player, s), self, None)
        self._in_step = False
        for objs in self._deferred:
            self.remove(*objs)
        self._deferred = []
        self._shadow = (w, banks, goals, player, traffic, alive)
