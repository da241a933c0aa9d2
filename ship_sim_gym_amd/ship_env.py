"""Single-env view with the reference's ``ShipEnv`` surface (ship_gym/ship_env.py:16-184), backed by one lane of
the HIP path in reference-exact ``fresh`` map mode: every ``reset()`` draws a brand-new river and goal path from
the global ``random`` / ``np.random`` streams in the reference's call order, so identical seeds give identical
worlds.  ``env.game`` offers the attribute reach-through the reference's tests use (tests/test_ship_env.py:22-36).

Stepping after ``done`` without ``reset()`` is outside the contract: the reference would keep simulating contact
response with Chipmunk's impulse solver, which this path does not carry for the player — so
``step()`` on a finished episode RAISES instead of silently returning states the reference would not produce.
"""
import numpy as np

from . import _native as N
from .vec_env import ShipVecEnv

try:  # pragma: no cover
    from gym import Env as _GymEnv  # type: ignore
except Exception:
    _GymEnv = object

DEFAULT_STATE_VAL = -1
STEP_PENALTY = -0.01


class _Body(object):
    def __init__(self, env):
        self._env = env

    @property
    def angle(self):
        return float(self._env._vec.field(N.F_ANGLE)[0].item())

    @property
    def position(self):
        v = self._env._vec
        return (float(v.field(N.F_X)[0].item()), float(v.field(N.F_Y)[0].item()))


class _Goal(object):
    def __init__(self, x, y):
        self.x, self.y = float(x), float(y)

    def __repr__(self):
        return "Goal(%s, %s)" % (self.x, self.y)


class _Traffic(object):
    def __init__(self, x, y, angle):
        self.x, self.y, self.angle = float(x), float(y), float(angle)


class _Player(object):
    def __init__(self, env):
        self._env = env
        self.body = _Body(env)

    @property
    def x(self):
        return self.body.position[0]

    @property
    def y(self):
        return self.body.position[1]

    @property
    def rudder_angle(self):
        return int(self._env._vec.field(N.F_RUDDER)[0].item())


class _GameView(object):
    """Read-only stand-in for ShipGame's attributes (game.py:21-58): player, goals, colliding, goal_reached, bounds."""

    def __init__(self, env):
        self._env = env
        self.player = _Player(env)
        self.bounds = env._vec.bounds

    @property
    def goals(self):
        v = self._env._vec
        mask = int(v.field(N.F_GOAL_MASK)[0].item())
        g = v.worlds[0][1]
        if v.n_ships > 1:  # config 4: goals are dynamic bodies, their position is body.position (game.py:343)
            b = v.field(N.F_GOAL_BODIES)[:, 0].cpu().numpy().reshape(N.MAX_GOALS, 8)
            return [_Goal(b[i, 0], b[i, 1]) for i in range(len(g)) if mask & (1 << i)]
        return [_Goal(g[i, 0], g[i, 1]) for i in range(len(g)) if mask & (1 << i)]

    @property
    def ships(self):
        """ShipGame.ships (game.py:284-286): the traffic ships of a config-4 env as (x, y, angle) views."""
        v = self._env._vec
        if v.n_ships <= 1:
            return []
        t = v.field(N.F_TRAFFIC)[:, 0].cpu().numpy().reshape(N.N_TRAFFIC, 9)
        return [_Traffic(*t[k, :3]) for k in range(N.N_TRAFFIC)]

    def add_default_traffic(self):
        """ShipGame.add_default_traffic (game.py:279-286).  The state layout of the HIP path is fixed at construction:
        build the env with ``n_ships=4`` and the three traffic ships are (re-)added by every reset, which is what a
        caller of the reference does by hand after each ``reset()``; this call is then a no-op."""
        if self._env._vec.n_ships <= 1:
            raise N.ShipSimError("add_default_traffic: construct ShipEnv / ShipVecEnv with n_ships=4")

    @property
    def colliding(self):
        return bool(self._env._last_flags & N.EV_COLLIDING)

    @property
    def goal_reached(self):
        return bool(self._env._last_flags & N.EV_GOAL_REACHED)

    def closest_goal(self):
        goals = self.goals
        if not goals:
            return None
        px, py = self.player.body.position
        best, bd = goals[0], np.hypot(goals[0].x - px, goals[0].y - py)
        for g in goals[1:]:
            d = np.hypot(g.x - px, g.y - py)
            if d < bd:
                best, bd = g, d
        return best


class ShipEnv(_GymEnv):
    metadata = {'render.modes': ['human', 'rgb_array']}
    reward_range = (-1, 1)

    def __init__(self, game_config=None, env_config=None, device="cuda:0", **kw):
        self._vec = ShipVecEnv(1, game_config, env_config, device=device, map_mode="fresh", auto_reset=False, **kw)
        self.game_config = self._vec.game_config   # what ShipVecEnv.from_env_fns reads back from a probe env
        self._ctor_kw = dict(kw)
        self._done = False
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.env_config = env_config
        self.n_states = self._vec.n_states
        self.states_history = self._vec.states_history
        self.game = _GameView(self)
        self.last_action = None
        self.reward = 0
        self.cumulative_reward = 0
        self.step_count = 0
        self.episodes_count = -1  # ship_env.py:33
        self._last_flags = 0
        self.states = None

    def seed(self, seed=None):
        np.random.seed(seed)  # ship_env.py:57-59
        return [seed]

    def reset(self, spawn_point=None, goals=None):
        """ShipEnv.reset (ship_env.py:171-184).  Extensions for scenario tests, mirroring what the reference's own
        (older-API) tests did with `reset(spawn_point=...)` and `game.add_goal(x, y)` (tests/test_ship_env.py:26-38):
        `goals` replaces the generated goal path with the given N_GOALS centres, `spawn_point` moves the ship."""
        if goals is None and spawn_point is None:
            obs = self._vec.reset()[0]
        else:
            obs = self._reset_with_override(spawn_point, goals)
        self._done = False
        self.last_action = None
        self.reward = 0
        self.cumulative_reward = 0
        self.step_count = 0
        self.episodes_count += 1
        self._last_flags = 0
        self.states = obs
        return obs

    def _reset_with_override(self, spawn_point, goals):
        """reset() onto a world whose goal path / spawn point the caller dictates.  The world is still drawn first (same
        RNG consumption as a plain reset); the override goes into the map record BEFORE ssg_reset, so everything the
        reset derives from it — the reset observation's nearest goal and, in config 4, the dynamic goal bodies and the
        previous-frame goal columns — sees the overridden goals."""
        import torch
        from . import worldgen
        v = self._vec
        v._fresh_world(0)
        polys, g0 = v.worlds[0]
        g = np.asarray(goals if goals is not None else g0, dtype=np.float64).reshape(v.cfg.n_goals, 2)
        sp = (float(spawn_point[0]), float(spawn_point[1])) if spawn_point is not None else (v.cfg.spawn_x, v.cfg.spawn_y)
        v.worlds[0] = (polys, g)
        v.bank_host[0] = worldgen.build_record(polys[0], polys[1], g, sp)
        v.bank.copy_(torch.from_numpy(v.bank_host))
        obs = v.reset_tensor().cpu().numpy()[0].copy()
        if spawn_point is not None:  # the reset kernel spawns at the configured point: move the body and its frame
            v.field(N.F_X)[0] = sp[0]
            v.field(N.F_Y)[0] = sp[1]
            F = self.n_states
            obs[-F + 0], obs[-F + 1] = sp
            v.obs[0].copy_(torch.from_numpy(obs))
        torch.cuda.synchronize()
        return obs

    def step(self, action):
        assert self.action_space.contains(action), "%r (%s) invalid" % (action, type(action))  # ship_env.py:143
        if self._done:
            raise N.ShipSimError("ShipEnv.step() called after done=True: call reset() first (the player's contact "
                                 "response after a collision is not simulated, DESIGN.md §2)")
        obs, rew, done, _ = self._vec.step(np.asarray([action]))
        self._done = bool(done[0])
        self._last_flags = int(self._vec.flags[0].item())
        self.last_action = action
        self.reward = float(rew[0])
        self.cumulative_reward += self.reward
        self.step_count += 1
        self.states = obs[0]
        return obs[0], self.reward, bool(done[0]), {}

    def render(self, mode='human', close=False):
        import sys
        if mode == 'rgb_array':  # metadata['render.modes'] (ship_env.py:18); the reference leaves it unimplemented
            return self._vec.render(mode='rgb_array', env=0)
        if self.last_action is not None:  # ship_env.py:165-168
            sys.stdout.write('action=%s, cumm_reward=%s' % (self.last_action, self.cumulative_reward))

    def close(self):
        self._vec.close()
