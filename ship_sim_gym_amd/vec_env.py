"""ShipVecEnv — the batched, MI355X-resident ShipEnv.

Keeps the reference's gym surface (ship_gym/ship_env.py:16-184: ``action_space = Discrete(3)``,
``observation_space = Box(0, max(bounds), (n_states*H,), uint8)``, ``reset()``, ``step()``) but for N envs at
once, and speaks the two vector-env protocols the reference's trainers use:

* stable-baselines 2.2.0 ``VecEnv`` (train/stable_baselines/ppo.py:122-123 builds a SubprocVecEnv): ``num_envs``,
  ``reset() -> obs[N,D]``, ``step_async(actions)``, ``step_wait() -> (obs, rews, dones, infos)``, ``step``,
  ``close``; a done env is reset inside the step and its returned observation is the reset observation.
* RLlib 0.6.0 ``VectorEnv`` (train/rllib/ppo.py:21-24,43): ``vector_reset()``, ``reset_at(i)``,
  ``vector_step(actions)``, ``get_unwrapped()``.

State for all envs lives in one caller-owned torch-ROCm byte tensor (struct-of-arrays columns, see
include/shipsim.h); PyTorch is only the device-buffer container and stream provider.  All compute goes through
libshipsim.so; there is no CPU path.

Two map modes:
* ``"bank"`` (default): a fixed bank of ``n_maps`` pre-generated worlds lives in HBM (and is staged in LDS by
  the kernel); env e starts on map ``(env_id_base+e) % n_maps`` and every in-kernel auto-reset moves it to the
  next map.  No host involvement per step.
* ``"fresh"``: reference-exact resets — every reset draws a brand-new world from the global ``random`` /
  ``np.random`` streams on the host (game.py:260-277), one bank slot per env.  Needs a host round trip on done.
* ``"fresh_device"``: a brand-new world for EVERY episode of every env, like the reference (ShipGame.reset generates a
  river and a goal path at every reset), but at batch scale and without the host: env e owns a ring of ``ring``
  bank records, episode p lives in record ``e*ring + p % ring`` and is drawn on the device from a Philox stream keyed
  by (map_seed, global env id, p) (ssg_config.map_ring / ssg_refill_worlds).  The in-kernel auto-reset moves an env
  to its next record; the library refills the rings between launches.  Not seed-compatible with the reference's
  Mersenne-Twister draws (use ``"fresh"`` for that); same geometry code as the host path, bit for bit.
"""
import ctypes as C
import inspect
import math
import warnings

import numpy as np

from . import _native as N
from . import config as cfgmod
from . import spaces, worldgen


def _torch():
    import torch
    return torch


def _trainer_bases():
    """Base classes for ShipVecEnv: the trainers' own abstract vector-env classes when they are importable, so that
    PPO2's `isinstance(env, VecEnv)` (stable-baselines, train/stable_baselines/ppo.py:88,122-123) and RLlib's
    `isinstance(env, VectorEnv)` (train/rllib/ppo.py:21-24,43) gates accept the batched env; `object` otherwise."""
    bases = []
    for mod, name in (("stable_baselines.common.vec_env", "VecEnv"), ("stable_baselines3.common.vec_env", "VecEnv"),
                      ("ray.rllib.env.vector_env", "VectorEnv")):
        if mod.startswith("stable_baselines3") and bases:
            continue  # one VecEnv flavour is enough
        try:
            cls = getattr(__import__(mod, fromlist=[name]), name)
        except Exception:
            continue
        if isinstance(cls, type) and cls not in bases:
            bases.append(cls)
    return tuple(bases) or (object,)


_BASES = _trainer_bases()


class EnvHandle(object):
    """One env of a ShipVecEnv, for the per-env calls of the trainers' APIs (`env_method`, `get_attr`, `set_attr`,
    RLlib's `get_unwrapped()`): attribute reads fall through to the batch, `reset()`/`seed()`/`render()` act on
    this env only."""

    def __init__(self, vec, index):
        object.__setattr__(self, "_vec", vec)
        object.__setattr__(self, "index", int(index))
        object.__setattr__(self, "_attrs", {})

    def __getattr__(self, name):
        attrs = object.__getattribute__(self, "_attrs")
        if name in attrs:
            return attrs[name]
        return getattr(object.__getattribute__(self, "_vec"), name)

    def __setattr__(self, name, value):
        self._attrs[name] = value

    def reset(self):
        return self._vec.reset_at(self.index)

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def render(self, mode='human', close=False):
        return self._vec.render(mode=mode, close=close, env=self.index)

    def close(self):
        return None


class ShipVecEnv(*_BASES):
    metadata = {'render.modes': ['human', 'rgb_array']}  # ship_env.py:18
    reward_range = (-1, 1)                                # ship_env.py:20

    def __init__(self, num_envs, game_config=None, env_config=None, device="cuda:0", map_mode="bank", n_maps=64,
                 map_seed=1000, width_frac=0.5, env_id_base=0, auto_reset=True, n_beams=None, bank=None,
                 fix_collision_reward=False, bank_in_global=False, exact_lidar=False, n_ships=1, rllib=False, ring=32,
                 dyn_memo=True, host_slots=4, copy_host_outputs=False):
        torch = _torch()
        if not torch.cuda.is_available():
            raise N.ShipSimError("ShipVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        game_config = game_config if game_config is not None else cfgmod.GameConfig
        env_config = env_config if env_config is not None else cfgmod.EnvConfig
        if env_config.HISTORY_SIZE < 1:  # ship_env.py:46-47
            raise ValueError("history_size must be greater than zero")
        self.num_envs = int(num_envs)
        self._i32 = torch.int32
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self._dev_index)  # (always with its index: tensors report `cuda:N`)
        self.map_mode = map_mode
        self.bounds = tuple(game_config.BOUNDS)
        self.width_frac = float(width_frac)
        # rllib=True: the RLlib VectorEnv flow — vector_step returns the TERMINAL observation of a done env and the
        # caller's reset_at(i) is the one reset it gets (no in-kernel auto-reset underneath)
        self.rllib = bool(rllib)
        # The RLlib flow WITHOUT a reset launch (ssg_set_terminal_obs, history <= 2, worlds that live on the device): the step
        # kernel resets a done env itself — its row of `obs` is then the reset observation reset_at(i) hands out — and also
        # stores the env's TERMINAL observation in `term_obs`, which vector_step reports.  Otherwise (history > 2, host-drawn
        # worlds) the rllib flow runs without in-kernel reset and resets its done envs with ONE masked ssg_reset per step.
        self._rllib_fused = self.rllib and env_config.HISTORY_SIZE <= 2 and map_mode in ("bank", "fresh_device")
        self.auto_reset = bool(auto_reset) and (not self.rllib or self._rllib_fused)
        self.env_id_base = int(env_id_base)
        self.game_config, self.env_config = game_config, env_config

        c = N.default_config()
        c.device_id = self.device.index if self.device.index is not None else torch.cuda.current_device()
        c.n_envs = self.num_envs
        c.env_id_base = self.env_id_base
        if n_beams is None:
            if getattr(env_config, "USE_LIDAR_CONFIG", False):
                lc = env_config.LIDAR_CONFIG
                c.n_beams, c.lidar_spread_deg, c.lidar_dist = lc.N_BEAMS, lc.ANGULAR_SPREAD, lc.DISTANCE
            else:  # the reference ignores LidarConfig: LiDAR() defaults (models.py:29,150)
                c.n_beams = cfgmod.LIDAR_DEFAULT_N_BEAMS
        else:
            c.n_beams = int(n_beams)
        c.history = int(env_config.HISTORY_SIZE)
        c.max_steps = int(env_config.MAX_STEPS)
        c.n_goals = cfgmod.N_GOALS
        c.width, c.height = float(self.bounds[0]), float(self.bounds[1])
        c.dt = game_config.SPEED * cfgmod.BASE_DT                 # game.py:194: speed * base_dt
        c.damping_pow_dt = math.pow(cfgmod.SPACE_DAMPING, c.dt)   # cpSpaceStep: pow(space.damping, dt)
        c.spawn_x, c.spawn_y = self.bounds[0] / 2, 25.0           # game.py:274
        flags = 0
        if self.auto_reset and map_mode in ("bank", "fresh_device"):
            flags |= N.FLAG_AUTO_RESET
        if fix_collision_reward:
            flags |= N.FLAG_FIX_COLLISION_REWARD
        if bank_in_global or map_mode == "fresh":
            flags |= N.FLAG_BANK_IN_GLOBAL
        if exact_lidar:
            flags |= N.FLAG_EXACT_LIDAR
        if not dyn_memo:  # config 4 measurement aid: every queued env computes its cpSpaceStep (no memo table look-ups)
            flags |= N.FLAG_DYN_MEMO_OFF
        c.flags = flags
        # n_ships = 4: BASELINE configs[3] — env.game.add_default_traffic() (game.py:279-286) after every reset: three
        # traffic ships, dynamic goal bodies and Chipmunk's contact solver (csrc/shipsim_dynamics.hip)
        c.n_ships = int(n_ships)
        self.n_ships = int(n_ships)
        self.ring = int(ring) if map_mode == "fresh_device" else 0
        c.map_ring = self.ring
        self.cfg = c
        self.n_states = 6 + c.n_beams                              # ship_env.py:43
        self.states_history = self.n_states * c.history            # ship_env.py:44
        self.action_space = spaces.Discrete(3)                     # ship_env.py:19
        self.observation_space = spaces.Box(low=0, high=max(self.bounds), shape=(self.states_history,),
                                            dtype=np.uint8)       # ship_env.py:48 (declared uint8; obs are float64)

        L = N.lib()
        self._h = C.c_void_p()
        N.check(L.ssg_create(C.byref(c), C.byref(self._h)), None, "ssg_create")
        nbytes = C.c_size_t()
        N.check(L.ssg_state_nbytes(self._h, C.byref(nbytes)), self._h, "ssg_state_nbytes")
        # obs | reward | done | flags are views of ONE device block (256-byte aligned sections), so that the numpy protocols
        # (SB VecEnv / RLlib VectorEnv) bring a whole step to the host with ONE device -> host copy into a pinned block
        n_, D_ = self.num_envs, self.states_history
        up = lambda v: (v + 255) // 256 * 256
        self._o_obs, self._o_rew = 0, up(n_ * D_ * 8)
        self._o_done = self._o_rew + up(n_ * 8)
        self._o_flags = self._o_done + up(n_)
        self._out_nbytes = self._o_flags + up(n_)
        self.host_slots = max(2, min(8, int(host_slots)))  # (ssg_step_host has eight completion-event slots)
        self.copy_host_outputs = bool(copy_host_outputs)
        with torch.cuda.device(self.device):
            self.state = torch.zeros(nbytes.value, dtype=torch.uint8, device=self.device)
            self._out_blob = torch.zeros(self._out_nbytes, dtype=torch.uint8, device=self.device)
            self.obs, self.reward, self.done, self.flags = self._blob_views(self._out_blob)
            self._actions = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self._host = None  # pinned host side of the numpy protocols, made by the first step_async (the tensor API never needs it)
        self.term_obs = None
        N.check(L.ssg_bind_state(self._h, C.c_void_p(self.state.data_ptr())), self._h, "ssg_bind_state")

        # ---- map bank ----
        if map_mode == "bank":
            if bank is None:
                bank, self.bank_polys, self.bank_goals = worldgen.build_bank(
                    n_maps, self.bounds, n_goals=c.n_goals, width_frac=self.width_frac, seed=map_seed)
            else:
                self.bank_polys = self.bank_goals = None
            self.set_bank(bank)
        elif map_mode == "fresh":
            # reference-exact: ShipGame.__init__ ends with self.reset() (game.py:58), so constructing an env already
            # consumes one world's worth of RNG; the user's reset() draws another (App. B-17).
            self.bank_host = np.zeros((self.num_envs, N.MAP_STRIDE), dtype=np.float64)
            self.worlds = [None] * self.num_envs
            for e in range(self.num_envs):
                self._fresh_world(e)
            self.set_bank(self.bank_host)
        elif map_mode == "fresh_device":
            with torch.cuda.device(self.device):
                bank = torch.zeros((self.num_envs * self.ring, N.MAP_STRIDE), dtype=torch.float64, device=self.device)
            self.bank_polys = self.bank_goals = None
            self.map_seed = int(map_seed)
            self.set_bank(bank)
            self.refill_worlds(self.map_seed)  # fills every ring: episodes 0 .. ring-1 of every env
        else:
            raise ValueError("map_mode must be 'bank', 'fresh' or 'fresh_device'")
        if self._rllib_fused:
            self.enable_terminal_obs()
        self._pending = None
        self._closed = False
        self._handles = {}
        self._await_reset = np.zeros(self.num_envs, dtype=bool)  # rllib flow: done envs already re-initialised
        self._reset_obs_h = None
        self._call_base_ctors()

    def _call_base_ctors(self):
        """The trainers' base-class constructors.  stable-baselines declares VecEnv.__init__(num_envs, observation_space,
        action_space), ray >= 1.x VectorEnv.__init__(observation_space, action_space, num_envs), ray 0.6 (the reference's
        pin) none: each is called with the KEYWORDS its signature names, never positionally, and our own attributes are
        re-asserted afterwards so that no base can have swapped them."""
        mine = {"num_envs": self.num_envs, "observation_space": self.observation_space, "action_space": self.action_space}
        # (the bases of ShipVecEnv itself, not of type(self): for a subclass the latter would list ShipVecEnv and re-enter
        # this constructor)
        for base in ShipVecEnv.__mro__[1:]:
            if base is object or "__init__" not in vars(base):
                continue
            try:
                params = inspect.signature(base.__init__).parameters
                base.__init__(self, **{k: v for k, v in mine.items() if k in params})
            except TypeError as ex:  # a signature this dispatch does not know: say so, keep our own attributes
                warnings.warn("ShipVecEnv: %s.__init__ not called (%s)" % (base.__name__, ex))
        self.num_envs, self.observation_space, self.action_space = mine["num_envs"], mine["observation_space"], mine["action_space"]

    @classmethod
    def from_env_fns(cls, env_fns, **kw):
        """SubprocVecEnv-shaped constructor (train/stable_baselines/ppo.py:122-123: ``SubprocVecEnv([make_env() for i in
        range(num_cpu)])``): one batched env of ``len(env_fns)`` envs instead of one OS process per env.  The first
        thunk is called once to learn the (game_config, env_config) the caller's ``ShipEnv(...)`` line passes."""
        env_fns = list(env_fns)
        if not env_fns:
            raise ValueError("need at least one env thunk")
        probe = env_fns[0]()
        gc, ec = getattr(probe, "game_config", None), getattr(probe, "env_config", None)
        extra = dict(getattr(probe, "_ctor_kw", {}))
        try:
            probe.close()
        except Exception:
            pass
        extra.update(kw)
        return cls(len(env_fns), gc, ec, **extra)

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _blob_views(self, blob):
        """(obs [N, D] f64, reward [N] f64, done [N] u8, flags [N] u8) views of an output block (device or pinned host)."""
        torch = _torch()
        n, D = self.num_envs, self.states_history
        return (blob[self._o_obs: self._o_obs + n * D * 8].view(torch.float64).view(n, D),
                blob[self._o_rew: self._o_rew + n * 8].view(torch.float64),
                blob[self._o_done: self._o_done + n], blob[self._o_flags: self._o_flags + n])

    def _host_side(self):
        """The host half of the numpy protocols, made once: `host_slots` pinned output blocks (rotated, so that the arrays of
        one step stay valid while the next steps run), a pinned action buffer and a side stream (the completion events live
        in the handle: ssg_step_host / ssg_wait_host)."""
        if self._host is None:
            torch = _torch()
            with torch.cuda.device(self.device):
                blocks = [torch.empty(self._out_nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(self.host_slots)]
                acts = torch.empty(self.num_envs, dtype=torch.int32, pin_memory=True)
                self._host = {"blocks": blocks,
                              "np": [tuple(v.numpy() for v in self._blob_views(b)) for b in blocks],
                              "acts": acts, "acts_np": acts.numpy(), "acts_ptr": acts.data_ptr(),
                              "block_ptrs": [b.data_ptr() for b in blocks], "stream": torch.cuda.Stream(device=self.device),
                              "step_host": N.lib().ssg_step_host, "wait_host": N.lib().ssg_wait_host, "slot": 0, "inflight": None,
                              "infos": [{} for _ in range(self.num_envs)]}
        return self._host

    def enable_terminal_obs(self, on=True):
        """ssg_set_terminal_obs: from now on every step ALSO stores the terminal observation of each env it auto-resets into that
        env's row of `self.term_obs` ([N, D] float64 device tensor; rows of envs that were not done keep whatever they held).  A
        GPU-resident caller gets both observations of an episode's end from one launch: `obs` (reset observation) and
        `torch.where(done[:, None] != 0, term_obs, obs)` (what RLlib's vector_step reports).  Needs in-kernel auto-reset and
        history <= 2."""
        torch = _torch()
        if on and self.term_obs is None:
            with torch.cuda.device(self.device):
                self.term_obs = torch.zeros((self.num_envs, self.states_history), dtype=torch.float64, device=self.device)
        N.check(N.lib().ssg_set_terminal_obs(self._h, C.c_void_p(self.term_obs.data_ptr()) if on else None), self._h, "ssg_set_terminal_obs")
        if not on:
            self.term_obs = None
        return self.term_obs

    def _fresh_world(self, e):
        rec, polys, goals = worldgen.generate_world(self.bounds, n_goals=self.cfg.n_goals, width_frac=self.width_frac)
        self.bank_host[e] = rec
        self.worlds[e] = (polys, goals)

    def set_bank(self, bank):
        """Install a map bank (numpy [M, MAP_STRIDE] or a device tensor).  A curriculum lesson change calls this."""
        torch = _torch()
        if isinstance(bank, np.ndarray):
            bank = torch.from_numpy(np.ascontiguousarray(bank, dtype=np.float64)).to(self.device)
        if not bank.is_cuda:
            bank = bank.to(self.device)
        assert bank.dtype == torch.float64 and bank.dim() == 2 and bank.shape[1] == N.MAP_STRIDE and bank.is_contiguous()
        assert bank.device == self.device, "set_bank: the bank lives on %s, the env on %s" % (bank.device, self.device)
        self.bank = bank
        self.n_maps = int(bank.shape[0])
        with torch.cuda.device(self.device):
            N.check(N.lib().ssg_set_map_bank(self._h, C.c_void_p(bank.data_ptr()), self.n_maps), self._h,
                    "ssg_set_map_bank")

    def regenerate_bank(self, seed, width_frac=None, n_maps=None, return_raw=False):
        """Refresh the map bank ON THE DEVICE (no host geometry, no upload): n_maps brand-new rivers and goal paths
        from a Philox stream keyed by (seed, map index).  Not seed-compatible with the reference's python/numpy RNG —
        use the host `worldgen` path for that.  Envs keep their map ids; the new geometry applies from the next step
        on, so callers normally follow with reset_tensor()."""
        torch = _torch()
        n_maps = self.n_maps if n_maps is None else int(n_maps)
        wf = self.width_frac if width_frac is None else float(width_frac)
        with torch.cuda.device(self.device):
            bank = self.bank if n_maps == self.n_maps else torch.empty((n_maps, N.MAP_STRIDE), dtype=torch.float64,
                                                                      device=self.device)
            raw = torch.empty((n_maps, 48 + 3 * self.cfg.n_goals), dtype=torch.float64, device=self.device) if return_raw else None
            N.check(N.lib().ssg_generate_bank(self._h, int(seed), wf, C.c_void_p(bank.data_ptr()), n_maps,
                                              C.c_void_p(raw.data_ptr()) if raw is not None else None, self._stream()),
                    self._h, "ssg_generate_bank")
        self.bank_polys = self.bank_goals = None  # host copies no longer describe the bank
        if bank is not self.bank:
            self.set_bank(bank)
        return raw

    def refill_worlds(self, seed=None, width_frac=None, return_raw=False):
        """map_mode="fresh_device": draw every world the rings are missing (ssg_refill_worlds) and make (seed, width_frac)
        the source of the automatic refills from now on — a curriculum lesson change passes the new width here.  With
        return_raw, returns [num_envs*ring, 48 + 3*n_goals]: rows of the slots drawn by THIS call hold the raw polygons
        and goal draws (rows of other slots are NaN)."""
        torch = _torch()
        if self.map_mode != "fresh_device":
            raise N.ShipSimError("refill_worlds needs map_mode='fresh_device'")
        seed = self.map_seed if seed is None else int(seed)
        self.map_seed = seed
        if width_frac is not None:
            self.width_frac = float(width_frac)
        with torch.cuda.device(self.device):
            raw = None
            if return_raw:
                raw = torch.full((self.num_envs * self.ring, 48 + 3 * self.cfg.n_goals), float("nan"), dtype=torch.float64,
                                 device=self.device)
            N.check(N.lib().ssg_refill_worlds(self._h, seed, self.width_frac, C.c_void_p(raw.data_ptr()) if raw is not None else None,
                                              self._stream()), self._h, "ssg_refill_worlds")
        return raw

    def launch_geometry(self):
        """(envs per workgroup, bank staged in LDS?, dynamic LDS bytes per workgroup) of the step kernel for this handle."""
        epw, lds, nb = C.c_int(), C.c_int(), C.c_size_t()
        N.check(N.lib().ssg_debug_launch_geometry(self._h, C.byref(epw), C.byref(lds), C.byref(nb)), self._h, "geometry")
        return epw.value, bool(lds.value), nb.value

    def field(self, fid):
        """Typed torch view [n_columns, num_envs] (or [num_envs]) into the state blob.  Config 4 (n_ships = 4): after WRITING
        any column through such a view — the player's pose as much as the traffic / goal bodies — call wake_dynamics():
        which envs the dyn kernels visit next step was decided from the state the last step ended with (ssg_dyn_invalidate)."""
        torch = _torch()
        off, es, nc, stride = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
        N.check(N.lib().ssg_state_field(self._h, fid, C.byref(off), C.byref(es), C.byref(nc), C.byref(stride)), self._h,
                "ssg_state_field")
        dt = {8: torch.float64, 4: torch.int32, 1: torch.uint8}[es.value]
        if fid == N.F_STATS:
            return self.state[off.value: off.value + 8 * nc.value].view(torch.int64).view(-1, 4)
        if fid == N.F_DYN_MEMO_STATS:
            return self.state[off.value: off.value + 8 * nc.value].view(torch.int64).view(-1, 16)
        n_pad = stride.value // es.value
        v = self.state[off.value: off.value + nc.value * stride.value].view(dt).view(nc.value, n_pad)[:, :self.num_envs]
        return v[0] if nc.value == 1 else v

    def _mask_ptr(self, mask, what):
        """A per-env byte mask as a C pointer (None = all envs): uint8 or bool, one element per env, on this env's device."""
        if mask is None:
            return None
        torch = _torch()
        if (mask.dtype not in (torch.uint8, torch.bool) or mask.numel() != self.num_envs or not mask.is_cuda or mask.device != self.device
                or not mask.is_contiguous()):
            raise ValueError("%s: mask must be a contiguous uint8 / bool tensor of %d elements on %s (got %s %s on %s)"
                             % (what, self.num_envs, self.device, mask.dtype, tuple(mask.shape), mask.device))
        return C.c_void_p(mask.data_ptr())

    def _ids_ptr(self, ids, what):
        if ids is None:
            return None
        torch = _torch()
        if ids.dtype != torch.int32 or ids.numel() != self.num_envs or ids.device != self.device or not ids.is_contiguous():
            raise ValueError("%s: map_ids must be a contiguous int32 tensor of %d elements on %s (got %s %s on %s)"
                             % (what, self.num_envs, self.device, ids.dtype, tuple(ids.shape), ids.device))
        return C.c_void_p(ids.data_ptr())

    def wake_dynamics(self, mask=None):
        """Config 4: call after writing ANY state column through field() — the traffic / goal-body columns (envs whose
        bodies had come to rest are otherwise not stepped; the ships' rotation columns, which the step kernel's collide_ship turns
        their hulls with, are recomputed from the angles — for every env) and the player's own columns (which envs the next full
        cpSpaceStep visits was decided from the state the last step ended with): ssg_dyn_invalidate.  Also after restoring or
        copying the state blob.  mask: uint8 device tensor [num_envs]
        (envs whose rest bit is cleared) or None = all; the next step's queue is rebuilt from the columns either way."""
        mp = self._mask_ptr(mask, "wake_dynamics")
        with _torch().cuda.device(self.device):
            N.check(N.lib().ssg_dyn_invalidate(self._h, mp, self._stream()), self._h, "ssg_dyn_invalidate")

    def field_stats_tensor(self):
        """int64 device tensor [4]: sum_return*100, sum_length, episodes, goals_hit of this handle (slots summed)."""
        return self.field(N.F_STATS).sum(dim=0)

    def kernel_times(self, enable):
        """Config 4 measurement aid (ssg_debug_kernel_times): returns (full cpSpaceStep us, step kernel us, steps) averaged over
        the steps since the previous call, and switches the per-launch HIP events on or off for the calls that follow."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint64()
        N.check(N.lib().ssg_debug_kernel_times(self._h, int(bool(enable)), C.byref(a), C.byref(b), C.byref(n)), self._h, "ssg_debug_kernel_times")
        return float(a.value), float(b.value), int(n.value)

    def dyn_counters(self):
        """Config 4: (launches of the full cpSpaceStep, how many of them rebuilt their queue from the per-env flags first)."""
        a, b = C.c_uint64(), C.c_uint64()
        N.check(N.lib().ssg_debug_dyn_counters(self._h, C.byref(a), C.byref(b)), self._h, "ssg_debug_dyn_counters")
        return int(a.value), int(b.value)

    def dyn_memo_stats(self):
        """Config 4: how the full cpSpaceStep of the queued envs was served so far — looked up in the memo table (`hits`),
        computed (`computed`), results stored (`stored`); SSG_F_DYN_MEMO_STATS."""
        s = self.field(N.F_DYN_MEMO_STATS).sum(dim=0).cpu().numpy()
        return {"hits": int(s[0]), "computed": int(s[1]), "stored": int(s[2]),
                "ship_x_bank_narrowphase": {"hits": int(s[3]), "computed": int(s[4])}}

    def stats(self):
        """Per-handle episode counters accumulated in-kernel: sum_return, sum_length, episodes, goals_hit."""
        s = self.field_stats_tensor().cpu().numpy()
        return {"sum_return": float(s[0]) / 100.0, "sum_length": int(s[1]), "episodes": int(s[2]), "goals_hit": int(s[3])}

    # ------------------------------------------------------------------------------------------------
    # tensor API (zero-copy; what a GPU-resident policy should use)
    # ------------------------------------------------------------------------------------------------
    def reset_tensor(self, mask=None, map_ids=None):
        torch = _torch()
        mp, ip = self._mask_ptr(mask, "reset_tensor"), self._ids_ptr(map_ids, "reset_tensor")
        with torch.cuda.device(self.device):
            if self.map_mode == "fresh" and map_ids is None:
                ids = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
                ip = C.c_void_p(ids.data_ptr())
            N.check(N.lib().ssg_reset(self._h, mp, ip, C.c_void_p(self.obs.data_ptr()), self._stream()), self._h,
                    "ssg_reset")
            if self.map_mode == "fresh" and map_ids is None:
                torch.cuda.current_stream(self.device).synchronize()  # keep `ids` alive until the kernel ran
        return self.obs

    def _out_ptrs(self):
        """The four output buffers as C pointers (they are allocated once; cached: these calls sit on the launch path)."""
        p = self.__dict__.get("_out_ptr_cache")
        if p is None or p[0] != (self.obs.data_ptr(), self.reward.data_ptr()):
            p = ((self.obs.data_ptr(), self.reward.data_ptr()),
                 C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                 C.c_void_p(self.flags.data_ptr()))
            self._out_ptr_cache = p
        return p[1:]

    def step_tensor(self, actions):
        """actions: int32 device tensor [N].  Returns (obs, reward, done, flags) device tensors (reused buffers).
        This is the policy-in-the-loop path: one launch per call, so the host side is kept to one ctypes call with plain
        integers (the cached output pointers, the actions' address, the current stream's handle)."""
        # (a tensor of another dtype / size would be read as int32 [N] all the same: wrong steps, or a read past its end)
        if (actions.dtype is not self._i32 or actions.numel() != self.num_envs or actions.device != self.device
                or not actions.is_contiguous()):
            raise ValueError("step_tensor: actions must be a contiguous int32 device tensor of %d elements (got %s %s on %s)"
                             % (self.num_envs, actions.dtype, tuple(actions.shape), actions.device))
        hot = self.__dict__.get("_hot")
        if hot is None or hot[0] != (self.obs.data_ptr(), self.reward.data_ptr()):
            torch = _torch()
            hot = self._hot = ((self.obs.data_ptr(), self.reward.data_ptr()), N.lib().ssg_step, torch.cuda.current_stream,
                               self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(), self.flags.data_ptr())
        _, fn, cur_stream, o, r, d, f = hot
        rc = fn(self._h, actions.data_ptr(), o, r, d, f, cur_stream(self.device).cuda_stream)
        if rc:
            torch = _torch()
            if torch.cuda.current_device() != self._dev_index:  # called with another device current: switch and retry
                with torch.cuda.device(self.device):
                    rc = fn(self._h, actions.data_ptr(), o, r, d, f, cur_stream(self.device).cuda_stream)
            if rc:
                N.check(rc, self._h, "ssg_step")
        return self.obs, self.reward, self.done, self.flags

    def rollout_tensor(self, actions_kn, trajectory=False, out=None):
        """K back-to-back steps from a pre-generated int32 [K, N] action tensor (random-action throughput run).

        trajectory=False: every step rewrites the env's reused [N, ...] buffers; returns the LAST step's
        (obs, reward, done, flags).  trajectory=True (ssg_rollout_traj): every step's outputs are kept, as the reference's
        rollout loop sees them (train/random.py:14-27) — returns (obs [K, N, D], reward [K, N], done [K, N], flags [K, N])
        device tensors; `out` = a tuple of four such preallocated tensors (first dimension >= K, contiguous) to write
        into instead of allocating.  self.obs / reward / done / flags are left untouched in trajectory mode."""
        if (actions_kn.dtype is not self._i32 or actions_kn.dim() != 2 or actions_kn.shape[1] != self.num_envs
                or actions_kn.device != self.device or not actions_kn.is_contiguous()):
            raise ValueError("rollout_tensor: actions must be a contiguous int32 device tensor [K, %d] (got %s %s on %s)"
                             % (self.num_envs, actions_kn.dtype, tuple(actions_kn.shape), actions_kn.device))
        if trajectory and out is not None:
            # a buffer set seen before: one ctypes call with plain integers, like step_tensor — a 20-step launch is ~140 us, and
            # everything the host does before the launch is GPU idle time inside a caller's timed region.  The cache holds
            # POINTERS and shapes only, never the tensors (a caller's `del bufs` really frees them), and a hit needs all four
            # buffers to sit where they sat when the plan was made, with the shapes checked then.
            po = out[0].data_ptr()
            plan = self.__dict__.setdefault("_traj_plans", {}).get(po)
            if plan is not None:
                K = actions_kn.shape[0]
                fn, cur_stream, pr, pd, pf, cap, sig = plan
                # (a hit needs the four buffers to be what was checked when the plan was made: addresses AND dtype / shape /
                # strides / device of each — the caching allocator readily hands a freed buffer's address to another tensor)
                if (K <= cap and out[1].data_ptr() == pr and out[2].data_ptr() == pd and out[3].data_ptr() == pf
                        and tuple((t.dtype, tuple(t.shape), t.stride(), t.device.index) for t in out) == sig):
                    if fn(self._h, actions_kn.data_ptr(), K, po, pr, pd, pf, self.num_envs, cur_stream(self.device).cuda_stream) == 0:
                        return out[0][:K], out[1][:K], out[2][:K], out[3][:K]  # (sliced after the launch: the GPU is already busy)
                    # (an error — e.g. another device is current: the checked path below repeats the call and reports)
        torch = _torch()
        K = int(actions_kn.shape[0])
        if not trajectory:
            o, r, d, f = self._out_ptrs()
            if torch.cuda.current_device() == self._dev_index:
                rc = N.lib().ssg_rollout(self._h, C.c_void_p(actions_kn.data_ptr()), K, o, r, d, f, self._stream())
            else:
                with torch.cuda.device(self.device):
                    rc = N.lib().ssg_rollout(self._h, C.c_void_p(actions_kn.data_ptr()), K, o, r, d, f, self._stream())
            if rc:
                N.check(rc, self._h, "ssg_rollout")
            return self.obs, self.reward, self.done, self.flags
        n, D = self.num_envs, self.states_history
        with torch.cuda.device(self.device):
            caller_out = out is not None
            if out is None:
                out = (torch.empty((K, n, D), dtype=torch.float64, device=self.device),
                       torch.empty((K, n), dtype=torch.float64, device=self.device),
                       torch.empty((K, n), dtype=torch.uint8, device=self.device),
                       torch.empty((K, n), dtype=torch.uint8, device=self.device))
            to, tr, td, tf = out
            assert to.dtype == torch.float64 and tr.dtype == torch.float64 and td.dtype == torch.uint8 and tf.dtype == torch.uint8
            assert tuple(to.shape[1:]) == (n, D) and all(tuple(t.shape[1:]) == (n,) for t in (tr, td, tf))
            assert all(t.is_contiguous() and t.shape[0] >= K and t.device == self.device for t in out)
            rc = N.lib().ssg_rollout_traj(self._h, C.c_void_p(actions_kn.data_ptr()), K, C.c_void_p(to.data_ptr()),
                                          C.c_void_p(tr.data_ptr()), C.c_void_p(td.data_ptr()), C.c_void_p(tf.data_ptr()),
                                          n, self._stream())
        if rc:
            N.check(rc, self._h, "ssg_rollout_traj")
        if caller_out:
            plans = self.__dict__.setdefault("_traj_plans", {})
            if len(plans) >= 8 and to.data_ptr() not in plans:  # (a handful of rotating buffer sets at most)
                plans.pop(next(iter(plans)))
            plans[to.data_ptr()] = (N.lib().ssg_rollout_traj, torch.cuda.current_stream, tr.data_ptr(), td.data_ptr(), tf.data_ptr(),
                                    min(int(t.shape[0]) for t in out),
                                    tuple((t.dtype, tuple(t.shape), t.stride(), t.device.index) for t in out))
        return to[:K], tr[:K], td[:K], tf[:K]

    def clear_traj_cache(self):
        """Forget the cached launch plans of rollout_tensor(trajectory=True, out=...).  They hold no tensors (pointers and
        shapes only), so this is never needed to free memory; it exists for callers that recycle addresses deliberately."""
        self.__dict__.pop("_traj_plans", None)

    def random_actions(self, seed, step0, K):
        """int32 [K, N] Philox action stream keyed by (seed, step, global env id), generated on the device."""
        torch = _torch()
        with torch.cuda.device(self.device):
            out = torch.empty((K, self.num_envs), dtype=torch.int32, device=self.device)
            N.check(N.lib().ssg_fill_actions(self._h, int(seed), int(step0), int(K), C.c_void_p(out.data_ptr()),
                                             self._stream()), self._h, "ssg_fill_actions")
        return out

    # ------------------------------------------------------------------------------------------------
    # stable-baselines VecEnv protocol (numpy in / numpy out)
    # ------------------------------------------------------------------------------------------------
    def reset(self):
        if self.map_mode == "fresh":
            for e in range(self.num_envs):
                self._fresh_world(e)
            self.bank.copy_(_torch().from_numpy(self.bank_host))
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        """VecEnv.step_async: checks the actions, then LAUNCHES the step — actions host -> device from a pinned buffer, ssg_step,
        and ONE device -> host copy of the packed obs | reward | done | flags block into a pinned host block, all on a side
        stream — and returns at once: the step runs while the caller does whatever it does between step_async and step_wait."""
        a = np.asarray(actions)
        # ship_env.py:143 `assert self.action_space.contains(action)` for the whole batch in one range check
        ok = a.dtype.kind in "iu" and a.size == self.num_envs and bool(np.all((a >= 0) & (a < self.action_space.n)))
        assert ok, "%r (%s) invalid" % (a, a.dtype)
        self._pending = a.astype(np.int32).reshape(self.num_envs)
        if self._closed:
            return
        hs = self._host_side()
        np.copyto(hs["acts_np"], self._pending)
        slot = hs["slot"]
        torch = _torch()
        io = hs["stream"]
        io.wait_stream(torch.cuda.current_stream(self.device))  # after whatever the tensor API queued (a reset, say)
        # ONE foreign call: actions host -> device, ssg_step, the packed block device -> host, the slot's completion event
        rc = hs["step_host"](self._h, hs["acts_ptr"], self._actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                             self.done.data_ptr(), self.flags.data_ptr(), self._out_blob.data_ptr(), hs["block_ptrs"][slot],
                             self._out_nbytes, slot, io.cuda_stream)
        if rc:
            with torch.cuda.device(self.device):  # (called with another device current: switch and repeat, then report)
                rc = hs["step_host"](self._h, hs["acts_ptr"], self._actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                     self.done.data_ptr(), self.flags.data_ptr(), self._out_blob.data_ptr(), hs["block_ptrs"][slot],
                                     self._out_nbytes, slot, io.cuda_stream)
            if rc:
                N.check(rc, self._h, "ssg_step_host")
        hs["inflight"] = slot
        hs["slot"] = (slot + 1) % self.host_slots

    def step_wait(self):
        """VecEnv.step_wait: waits for the event behind step_async's device -> host copy and returns numpy views of that pinned
        block — obs [N, D] float64, rewards [N], dones [N] bool — plus a reused list of N empty info dicts.  The observation
        array is a view of one of `host_slots` rotating blocks: it stays valid until `host_slots - 1` further steps have been
        taken (stable-baselines' runners copy it into their own buffer at once: `self.obs[:] = obs`); rewards and dones are
        fresh small arrays (runners keep them in lists).  `copy_host_outputs=True` returns a fresh observation array too."""
        torch = _torch()
        hs = self._host_side()
        slot = hs["inflight"]
        if slot is None:
            raise N.ShipSimError("step_wait without a step_async")
        hs["inflight"] = None
        # (the step is COMPLETE when this returns — a host-side wait on the slot's event — so later tensor-API calls on any stream see it)
        N.check(hs["wait_host"](self._h, slot), self._h, "ssg_wait_host")
        obs_h, rew_h, done_u8, _ = hs["np"][slot]
        done_h = done_u8.view(np.bool_).copy()
        if self.map_mode == "fresh" and self.auto_reset and done_h.any():
            # host-side auto-reset with brand-new worlds (reference-exact resets)
            for e in np.nonzero(done_h)[0]:
                self._fresh_world(int(e))
            self.bank.copy_(torch.from_numpy(self.bank_host))
            mask = torch.from_numpy(done_h.astype(np.uint8)).to(self.device)
            ids = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
            self.reset_tensor(mask=mask, map_ids=ids)
            obs_h[done_h] = self.obs[mask.bool()].cpu().numpy()  # (the reset observations replace the terminal ones)
            torch.cuda.current_stream(self.device).synchronize()
        return (obs_h.copy() if self.copy_host_outputs else obs_h), rew_h.copy(), done_h, hs["infos"]

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.__dict__.pop("_traj_plans", None)
        hs = self.__dict__.get("_host")
        if hs is not None:
            try:
                hs["stream"].synchronize()  # a step_async still in flight reads and writes the buffers about to go
            except Exception:
                pass
            self._host = None
        if not self._closed and self._h:
            N.lib().ssg_destroy(self._h)
            self._h = None
            self._closed = True

    def seed(self, seed=None):
        """ShipEnv.seed seeds numpy's global generator only (ship_env.py:52-60)."""
        np.random.seed(seed)
        return [seed]

    def get_screen(self, env=0, width=None, height=None, debug=True):
        """ShipGame.render + get_screen (game.py:133-138,197-229) for one env: uint8 [width, height, 3] like
        pygame.surfarray.array3d (x first, screen y down), rasterised on the GPU (csrc/shipsim_render.hip)."""
        torch = _torch()
        width = int(width or self.bounds[0])
        height = int(height or self.bounds[1])
        with torch.cuda.device(self.device):
            img = torch.empty((width, height, 3), dtype=torch.uint8, device=self.device)
            N.check(N.lib().ssg_render(self._h, int(env), width, height, C.c_void_p(img.data_ptr()), 1 if debug else 0,
                                       self._stream()), self._h, "ssg_render")
        return img

    def render(self, mode='human', close=False, env=0):
        """ShipEnv.render (ship_env.py:158-168) only writes a text line (done by the ShipEnv facade); 'rgb_array'
        returns the frame the reference's screen would hold, as a numpy array [H, W, 3]."""
        if mode == 'rgb_array':
            return self.get_screen(env).permute(1, 0, 2).cpu().numpy()
        return None

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, (int, np.integer)):
            return [int(indices)]
        return [int(i) for i in indices]

    def env(self, index):
        """Per-env handle (cached): what `env_method` / `get_attr` / `set_attr` / `get_unwrapped` act on."""
        index = int(index)
        if not 0 <= index < self.num_envs:
            raise IndexError(index)
        h = self._handles.get(index)
        if h is None:
            h = self._handles[index] = EnvHandle(self, index)
        return h

    def get_attr(self, attr_name, indices=None):
        """VecEnv.get_attr: the attribute of each selected env (per-env values set by set_attr win over the batch's)."""
        return [getattr(self.env(i), attr_name) for i in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        """VecEnv.set_attr: stored per env; the batched physics configuration itself is fixed at construction."""
        for i in self._indices(indices):
            setattr(self.env(i), attr_name, value)

    def env_method(self, method_name, *method_args, **method_kwargs):
        """VecEnv.env_method: call `method_name` on each selected env handle (reset / seed / render / close, or any
        batch method that takes no env index) and return the list of results."""
        indices = method_kwargs.pop("indices", None)
        return [getattr(self.env(i), method_name)(*method_args, **method_kwargs) for i in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [self.render(mode='rgb_array', env=i) for i in range(self.num_envs)]

    @property
    def unwrapped(self):
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    # RLlib VectorEnv protocol
    # ------------------------------------------------------------------------------------------------
    def vector_reset(self):
        self._await_reset[:] = False
        return list(self.reset())

    def reset_at(self, index):
        """VectorEnv.reset_at: the ONE reset of env `index` (game.py:260-277).  In the rllib flow vector_step has
        already re-initialised every env it reported done, in one masked launch, and cached the reset observations;
        this call hands that observation out instead of resetting (and advancing the map) a second time."""
        torch = _torch()
        index = int(index)
        if self._await_reset[index]:
            self._await_reset[index] = False
            return self._reset_obs_h[index].copy()
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        mask[index] = 1
        if self.map_mode == "fresh":
            self._fresh_world(int(index))
            self.bank.copy_(torch.from_numpy(self.bank_host))
            ids = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
            self.reset_tensor(mask=mask, map_ids=ids)
        elif self.map_mode == "fresh_device":
            self.reset_tensor(mask=mask)  # the reset kernel moves the env to the next world of its ring
        else:
            ids = self.field(N.F_MAP_ID).clone()
            ids[index] = (ids[index] + 1) % self.n_maps
            self.reset_tensor(mask=mask, map_ids=ids.contiguous())
        torch.cuda.current_stream(self.device).synchronize()
        return self.obs[index].cpu().numpy()

    def _reset_done_envs(self, done_h):
        """rllib flow: re-initialise all done envs with ONE masked reset launch (next map of the bank, exactly the
        sequence the in-kernel auto-reset follows) into a side buffer, leaving self.obs = the terminal observations."""
        torch = _torch()
        mask = torch.from_numpy(done_h.astype(np.uint8)).to(self.device)
        if self.map_mode == "fresh":
            for e in np.nonzero(done_h)[0]:
                self._fresh_world(int(e))
            self.bank.copy_(torch.from_numpy(self.bank_host))
            ids = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
        elif self.map_mode == "fresh_device":
            ids = None  # the reset kernel moves each env to the next world of its ring
        else:
            ids = self.field(N.F_MAP_ID).clone()
            ids = torch.where(mask != 0, (ids + 1) % self.n_maps, ids).to(torch.int32).contiguous()
        side = torch.empty_like(self.obs)
        with torch.cuda.device(self.device):
            N.check(N.lib().ssg_reset(self._h, C.c_void_p(mask.data_ptr()), C.c_void_p(ids.data_ptr()) if ids is not None else None,
                                      C.c_void_p(side.data_ptr()), self._stream()), self._h, "ssg_reset")
        torch.cuda.current_stream(self.device).synchronize()
        if self._reset_obs_h is None:
            self._reset_obs_h = np.empty((self.num_envs, self.states_history), dtype=np.float64)
        idx = np.nonzero(done_h)[0]
        self._reset_obs_h[idx] = side[torch.from_numpy(idx).to(self.device)].cpu().numpy()
        self._await_reset |= done_h

    def vector_step(self, actions):
        """VectorEnv.vector_step.  Returns (obs, rewards, dones, infos) as SEQUENCES of per-env items: numpy arrays [N, D] /
        [N] / [N] — RLlib's own adapter only enumerates them (`dict(enumerate(self.new_obs))`), and a per-row `list()` of 65 536
        rows costs more than the step — and the reused list of info dicts.  The observation array is a fresh copy: RLlib's
        sample builders keep the rows by reference."""
        if not self.rllib:
            obs, rew, done, infos = self.step(np.asarray(actions))
            return (obs if self.copy_host_outputs else obs.copy()), rew, done, infos
        if self._await_reset.any():
            raise N.ShipSimError("vector_step: envs %r were reported done and have not been reset_at()"
                                 % (np.nonzero(self._await_reset)[0][:8].tolist(),))
        obs, rew, done, infos = self.step(np.asarray(actions))
        obs = obs if self.copy_host_outputs else obs.copy()
        if done.any():
            if self._rllib_fused:
                # the step kernel has already reset the done envs: their rows of `obs` are the reset observations (kept for
                # reset_at), their terminal observations are in term_obs — no reset launch
                torch = _torch()
                idx = np.nonzero(done)[0]
                if self._reset_obs_h is None:
                    self._reset_obs_h = np.empty((self.num_envs, self.states_history), dtype=np.float64)
                self._reset_obs_h[idx] = obs[idx]
                obs[idx] = self.term_obs[torch.from_numpy(idx).to(self.device)].cpu().numpy()
                self._await_reset |= done
            else:  # no in-kernel reset underneath: obs rows of done envs are terminal; ONE masked reset for all of them
                self._reset_done_envs(done)
        return obs, rew, done, infos

    def get_unwrapped(self):
        """VectorEnv.get_unwrapped: the underlying envs, as per-env handles."""
        return [self.env(i) for i in range(self.num_envs)]
