"""ctypes binding of libshipsim.so (include/shipsim.h) — the only way the Python host reaches the HIP path.

There is no CPU fallback: if the shared library is missing this module raises at import of the symbol table,
and every compute entry point returns an error (raised as ShipSimError) when no MI355X/HIP device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSG_LIB_PATH: development override (tools/build_variant.sh builds diagnostic variants next to the product library)
LIB_PATH = os.environ.get("SSG_LIB_PATH") or os.path.join(_HERE, "libshipsim.so")

ABI_VERSION = 8
MAX_BEAMS, MAX_GOALS, MAX_HULL, SHIP_VERTS, N_TRAFFIC = 16, 6, 12, 5, 3
MAP_STRIDE = 145
MAP_OFF_COUNTS, MAP_OFF_AABB, MAP_OFF_GOALS, MAP_OFF_SPAWN_GOAL, MAP_OFF_PLANES, PLANE_DOUBLES = 0, 2, 10, 22, 24, 5
FLAG_AUTO_RESET, FLAG_FIX_COLLISION_REWARD, FLAG_BANK_IN_GLOBAL, FLAG_EXACT_LIDAR, FLAG_DYN_MEMO_OFF = 0x1, 0x2, 0x4, 0x8, 0x10
EV_COLLIDING, EV_GOAL_REACHED, EV_OUT_OF_BOUNDS, EV_MAX_STEPS, EV_NO_GOALS_LEFT = 0x1, 0x2, 0x4, 0x8, 0x10
(F_X, F_Y, F_VX, F_VY, F_ANGLE, F_W, F_CUM_REWARD, F_LIDAR, F_RUDDER, F_STEP_COUNT, F_MAP_ID, F_GOAL_MASK,
 F_STATS, F_TRAFFIC, F_GOAL_BODIES, F_DYN_FLAGS, F_EPISODES, F_DYN_MEMO_STATS) = range(18)

# every symbol include/shipsim.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = (
    "ssg_abi_version", "ssg_strerror", "ssg_last_error", "ssg_create", "ssg_destroy", "ssg_default_config",
    "ssg_config_set_ship", "ssg_state_nbytes", "ssg_state_field", "ssg_bind_state", "ssg_set_map_bank", "ssg_reset",
    "ssg_step", "ssg_rollout", "ssg_fill_actions", "ssg_host_convex_hull", "ssg_host_moment_for_poly", "ssg_host_goal_x_range",
    "ssg_host_build_map", "ssg_host_segment_query", "ssg_debug_copy8", "ssg_generate_bank", "ssg_render", "ssg_dyn_invalidate",
    "ssg_init_state", "ssg_refill_worlds", "ssg_debug_launch_geometry", "ssg_rollout_traj", "ssg_debug_dyn_counters", "ssg_debug_kernel_times",
    "ssg_debug_clock_probe", "ssg_debug_launch_clock", "ssg_set_terminal_obs", "ssg_step_host", "ssg_wait_host",
)


class ShipSimError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32), ("device_id", C.c_int32), ("n_envs", C.c_int32),
        ("env_id_base", C.c_int64),
        ("n_beams", C.c_int32), ("history", C.c_int32), ("max_steps", C.c_int32), ("n_goals", C.c_int32),
        ("lidar_spread_deg", C.c_double), ("lidar_dist", C.c_double), ("goal_radius", C.c_double),
        ("width", C.c_double), ("height", C.c_double), ("dt", C.c_double), ("damping_pow_dt", C.c_double),
        ("spawn_x", C.c_double), ("spawn_y", C.c_double),
        ("ship_hull", C.c_double * (2 * SHIP_VERTS)), ("ship_normals", C.c_double * (2 * SHIP_VERTS)),
        ("ship_m_inv", C.c_double), ("ship_i_inv", C.c_double), ("force_y", C.c_double),
        ("thrust_px0", C.c_double), ("thrust_py0", C.c_double),
        ("rudder_step", C.c_int32), ("rudder_max", C.c_int32),
        ("n_ships", C.c_int32), ("map_ring", C.c_int32),
    ]


_lib = None


def lib():
    """Load libshipsim.so (built in-tree by __graft_entry__.build() / csrc/Makefile).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ShipSimError(
            "libshipsim.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ship_sim_gym_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # libshipsim.so needs libamdhip64.so.7.  PyTorch-ROCm bundles its own copy of that runtime (same SONAME) and
    # is the owner of the device memory and streams we are handed, so torch must be imported FIRST: the dynamic
    # loader then binds our HIP calls to the runtime already in the process instead of mapping a second one from
    # /opt/rocm (two HIP runtimes in one process cannot share streams and fail at the first hipGetDevice).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, dp, i32p, u8p, szp, ip = (C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_int))
    L.ssg_abi_version.restype = C.c_int
    L.ssg_strerror.restype = C.c_char_p
    L.ssg_strerror.argtypes = [C.c_int]
    L.ssg_last_error.restype = C.c_char_p
    L.ssg_last_error.argtypes = [vp]
    L.ssg_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.ssg_destroy.argtypes = [vp]
    L.ssg_default_config.argtypes = [C.POINTER(Config)]
    L.ssg_config_set_ship.argtypes = [C.POINTER(Config), C.c_double, C.c_double, C.c_double]
    L.ssg_state_nbytes.argtypes = [vp, szp]
    L.ssg_state_field.argtypes = [vp, C.c_int, szp, ip, ip, szp]
    L.ssg_bind_state.argtypes = [vp, vp]
    L.ssg_init_state.argtypes = [vp, vp]
    L.ssg_debug_launch_geometry.argtypes = [vp, ip, ip, szp]
    L.ssg_debug_kernel_times.argtypes = [vp, C.c_int, dp, dp, C.POINTER(C.c_uint64)]
    L.ssg_debug_dyn_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ssg_refill_worlds.argtypes = [vp, C.c_uint64, C.c_double, vp, vp]
    L.ssg_set_map_bank.argtypes = [vp, vp, C.c_int]
    L.ssg_reset.argtypes = [vp, vp, vp, vp, vp]
    L.ssg_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.ssg_rollout.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
    L.ssg_rollout_traj.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int64, vp]
    L.ssg_fill_actions.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp]
    L.ssg_generate_bank.argtypes = [vp, C.c_uint64, C.c_double, vp, C.c_int, vp, vp]
    L.ssg_debug_copy8.argtypes = [vp, vp, C.c_size_t, vp]
    L.ssg_debug_clock_probe.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ssg_debug_launch_clock.argtypes = [vp, vp]
    L.ssg_set_terminal_obs.argtypes = [vp, vp]
    L.ssg_step_host.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.ssg_wait_host.argtypes = [vp, C.c_int]
    L.ssg_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_uint32, vp]
    L.ssg_dyn_invalidate.argtypes = [vp, vp, vp]
    L.ssg_host_convex_hull.argtypes = [C.c_int, dp, dp, ip]
    L.ssg_host_moment_for_poly.argtypes = [C.c_double, C.c_int, dp, dp]
    L.ssg_host_goal_x_range.argtypes = [dp, C.c_double, C.c_double, dp, dp, ip]
    L.ssg_host_build_map.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp]
    L.ssg_host_segment_query.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, ip,
                                         dp, dp, dp]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = a declared entry point is not exported
        if getattr(L, name).restype is C.c_int and name not in ("ssg_abi_version",):
            pass
    if L.ssg_abi_version() != ABI_VERSION:
        raise ShipSimError("libshipsim.so ABI %d != binding ABI %d" % (L.ssg_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc, handle=None, what=""):
    if rc != 0:
        L = lib()
        msg = L.ssg_last_error(handle).decode() if handle is not None else L.ssg_last_error(None).decode()
        raise ShipSimError("%s failed: %s (%s)" % (what or "libshipsim call", L.ssg_strerror(rc).decode(), msg))


def default_config():
    c = Config()
    check(lib().ssg_default_config(C.byref(c)), None, "ssg_default_config")
    return c
