"""CPU-side checks of the C ABI (include/shipsim.h): the library loads, exports every declared symbol, reports
errors instead of crashing, describes its state layout, and its host geometry agrees with the oracle.
No compute entry point is exercised here (no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "shipsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(native):
    declared = _declared_symbols()
    assert len(declared) >= 19
    L = native.lib()
    for name in declared:
        assert hasattr(L, name), "include/shipsim.h declares %s but libshipsim.so does not export it" % name
    assert sorted(native.EXPORTS) == declared  # the Python binding covers the whole header
    assert L.ssg_abi_version() == native.ABI_VERSION


def test_header_constants_match_binding(native):
    text = open(os.path.join(ROOT, "include", "shipsim.h")).read()
    consts = dict(re.findall(r"#define\s+(SSG_[A-Z_]+)\s+(0x[0-9a-fA-F]+u?|\d+)", text))
    val = lambda k: int(consts[k].rstrip("u"), 0)
    assert val("SSG_MAP_STRIDE") == native.MAP_STRIDE and val("SSG_PLANE_DOUBLES") == native.PLANE_DOUBLES
    assert val("SSG_MAP_OFF_PLANES") == native.MAP_OFF_PLANES and val("SSG_MAP_OFF_GOALS") == native.MAP_OFF_GOALS
    assert val("SSG_MAP_OFF_SPAWN_GOAL") == native.MAP_OFF_SPAWN_GOAL
    assert val("SSG_MAX_BEAMS") == native.MAX_BEAMS and val("SSG_ABI_VERSION") == native.ABI_VERSION
    assert val("SSG_FLAG_AUTO_RESET") == native.FLAG_AUTO_RESET and val("SSG_FLAG_EXACT_LIDAR") == native.FLAG_EXACT_LIDAR
    assert val("SSG_EV_COLLIDING") == native.EV_COLLIDING and val("SSG_EV_NO_GOALS_LEFT") == native.EV_NO_GOALS_LEFT
    assert val("SSG_MAP_STRIDE") % 2 == 1  # odd 8-byte stride: LDS bank spreading (see the header)
    assert native.MAP_OFF_PLANES + 2 * 12 * native.PLANE_DOUBLES <= native.MAP_STRIDE


def test_default_config_is_the_reference_configuration(native, oracle):
    c = native.default_config()
    assert c.struct_size == C.sizeof(native.Config)
    assert (c.n_beams, c.history, c.max_steps, c.n_goals) == (10, 2, 1000, 5)          # models.py:29, config.py:15-16
    assert (c.width, c.height, c.dt) == (600.0, 600.0, 1.0)                            # config.py:23-24
    assert c.damping_pow_dt == 0.4 and (c.spawn_x, c.spawn_y) == (300.0, 25.0)         # game.py:270,274
    assert (c.force_y, c.goal_radius, c.lidar_dist, c.lidar_spread_deg) == (100.0, 5.0, 100.0, 90.0)
    assert (c.rudder_step, c.rudder_max, c.thrust_px0, c.thrust_py0) == (5, 10, 0.0, 0.0)
    assert list(c.ship_hull) == [0, 0, 20, 0, 20, 30, 10, 45, 0, 30]                   # cpConvexHull order
    assert c.ship_m_inv == 0.2 and c.ship_i_inv == 1.0 / 3087.5                         # cpMomentForPoly
    ship = oracle.make_poly([(0, 0), (0, 30), (10, 45), (20, 30), (20, 0)])
    for i in range(5):
        assert (c.ship_normals[2 * i], c.ship_normals[2 * i + 1]) == (ship.ln[i].x, ship.ln[i].y)
    # traffic-ship shapes of add_default_traffic (game.py:279-286) through ssg_config_set_ship
    L = native.lib()
    for (ws, hs), moment in (((1, 1), 433.3333333333333), ((1.5, 2), 1448.9583333333333), ((1, 3), 2600.0)):
        native.check(L.ssg_config_set_ship(C.byref(c), ws, hs, 5.0))
        assert abs(1.0 / c.ship_i_inv - moment) < 1e-9


def test_create_validates_and_reports(native):
    L = native.lib()
    h = C.c_void_p()
    c = native.default_config()
    c.history = 0
    rc = L.ssg_create(C.byref(c), C.byref(h))
    assert rc < 0 and b"history_size must be greater than zero" in L.ssg_last_error(None)  # ship_env.py:46-47
    for field, bad in (("n_beams", 0), ("n_beams", 17), ("n_envs", 0), ("n_goals", 7), ("struct_size", 4), ("history", 9)):
        c = native.default_config()
        setattr(c, field, bad)
        assert L.ssg_create(C.byref(c), C.byref(h)) < 0, field
    assert L.ssg_strerror(0) == b"ok" and L.ssg_strerror(-3) != L.ssg_strerror(-1)


def test_state_layout_and_unbound_errors(native):
    L = native.lib()
    c = native.default_config()
    c.n_envs = 1000
    c.n_beams = 8
    h = C.c_void_p()
    native.check(L.ssg_create(C.byref(c), C.byref(h)))
    nbytes = C.c_size_t()
    native.check(L.ssg_state_nbytes(h, C.byref(nbytes)), h)
    n_pad = 1024
    assert nbytes.value == 256 * 4 * 8 + (7 + 8) * n_pad * 8 + 5 * n_pad * 4 + n_pad  # 5 i32 columns: rudder, step, map, episodes, worlds drawn
    spans = []
    for fid, es, nc in ((native.F_X, 8, 1), (native.F_W, 8, 1), (native.F_CUM_REWARD, 8, 1), (native.F_LIDAR, 8, 8),
                        (native.F_RUDDER, 4, 1), (native.F_MAP_ID, 4, 1), (native.F_GOAL_MASK, 1, 1)):
        off, e, n, stride = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
        native.check(L.ssg_state_field(h, fid, C.byref(off), C.byref(e), C.byref(n), C.byref(stride)), h)
        assert (e.value, n.value) == (es, nc) and stride.value == n_pad * es and off.value % es == 0
        spans.append((off.value, off.value + nc * stride.value))
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= nbytes.value
    # compute entry points before binding: an error code and a message, never a crash, never a silent CPU path
    assert L.ssg_step(h, None, None, None, None, None, None) == -3
    assert b"ssg_bind_state" in L.ssg_last_error(h)
    assert L.ssg_reset(h, None, None, None, None) == -3
    assert L.ssg_bind_state(h, C.c_void_p(0x1008)) < 0  # misaligned blob
    # the host-array step (ssg_step_host / ssg_wait_host) and the round-6 setters validate before they touch a device
    assert L.ssg_step_host(h, None, None, None, None, None, None, None, None, 0, 0, None) == -1
    assert L.ssg_step_host(h, C.c_void_p(0x1000), C.c_void_p(0x1000), None, None, None, None, C.c_void_p(0x1000), C.c_void_p(0x1000), 64, 8, None) == -1  # slot 0..7
    assert L.ssg_wait_host(h, 0) == -1 and b"no ssg_step_host" in L.ssg_last_error(h)  # nothing was issued into that slot
    assert L.ssg_wait_host(h, -1) == -1 and L.ssg_wait_host(h, 8) == -1
    assert L.ssg_set_terminal_obs(h, None) == 0 and L.ssg_debug_launch_clock(h, None) == 0
    assert L.ssg_debug_clock_probe(None, 8, 100, None) == -1
    L.ssg_destroy(h)
    for flags, hist, code, text in ((0, 2, -1, b"SSG_FLAG_AUTO_RESET"), (native.FLAG_AUTO_RESET, 3, -4, b"history > 2")):
        c2 = native.default_config()
        c2.n_envs, c2.flags, c2.history = 64, flags, hist
        h2 = C.c_void_p()
        native.check(L.ssg_create(C.byref(c2), C.byref(h2)))
        assert L.ssg_set_terminal_obs(h2, C.c_void_p(0x1000)) == code and text in L.ssg_last_error(h2)  # refused; NULL always passes
        assert L.ssg_set_terminal_obs(h2, None) == 0
        L.ssg_destroy(h2)


def test_host_geometry_agrees_with_oracle(native, oracle):
    """Two independent implementations (Andrew monotone chain in C++ vs QuickHull in C) of what pm.Poly / the fat
    segment queries of gen_goal_path did for the reference, on the reference's own maps."""
    from ship_sim_gym_amd import worldgen
    d = np.load(os.path.join(ROOT, "tests", "golden", "ref_maps.npz"))
    L = native.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(0, 96, 5):
        polys = d["polys"][i]
        W, H = d["bounds"][i]
        rec = worldgen.build_record(polys[0], polys[1], np.zeros((0, 2)), (W / 2, 25))
        w = oracle.World(oracle.default_config(width=float(W), height=float(H)))
        w.set_banks_only(polys[0], polys[1])
        for s in range(2):
            hull = oracle.convex_hull(polys[s])
            n = int(rec[s])
            assert n == len(hull)
            PD = native.PLANE_DOUBLES
            pl = rec[native.MAP_OFF_PLANES + s * 12 * PD: native.MAP_OFF_PLANES + s * 12 * PD + PD * n].reshape(n, PD)
            np.testing.assert_array_equal(pl[:, :2], hull)
            op = oracle.make_poly(polys[s])
            np.testing.assert_array_equal(pl[:, 2], [op.ln[j].x for j in range(n)])
            np.testing.assert_array_equal(pl[:, 3], [op.ln[j].y for j in range(n)])
            np.testing.assert_array_equal(rec[2 + 4 * s: 6 + 4 * s], [op.bb_l, op.bb_b, op.bb_r, op.bb_t])
            # thin and fat segment queries
            rng = np.random.RandomState(i * 2 + s)
            for _ in range(40):
                a = rng.uniform([0, 0], [W, H]); b = rng.uniform([0, 0], [W, H]); r = float(rng.choice([0.0, 10.0]))
                hit, px, py, al = C.c_int(), C.c_double(), C.c_double(), C.c_double()
                native.check(L.ssg_host_segment_query(dp(rec), s, a[0], a[1], b[0], b[1], r, C.byref(hit), C.byref(px),
                                                      C.byref(py), C.byref(al)))
                oh, opt, _, oal = oracle.segment_query(op, a, b, r)
                assert bool(hit.value) == oh and al.value == oal and (px.value, py.value) == opt
        for y in np.linspace(0.1 * H, 0.9 * H, 7):
            assert worldgen.goal_x_range(rec, W, y) == w.goal_x_range(y)
        out = np.zeros((12, 2)); cnt = C.c_int()
        native.check(L.ssg_host_convex_hull(12, dp(np.ascontiguousarray(polys[0])), dp(out), C.byref(cnt)))
        np.testing.assert_array_equal(out[:cnt.value], oracle.convex_hull(polys[0]))
    m = C.c_double()
    native.check(L.ssg_host_moment_for_poly(5.0, 5, dp(np.array([(0, 0), (0, 30), (10, 45), (20, 30), (20, 0)], dtype=float)), C.byref(m)))
    assert m.value == 3087.5


def test_worldgen_reset_sequence_matches_oracle_and_rng_order(native, oracle):
    """generate_world consumes python `random` and numpy's generator exactly as ShipGame.reset does (game.py:66,
    320,325): same draws fed to the oracle's own goal placement give the same goals; the record's spawn goal is
    closest_goal from (W/2, 25)."""
    import random
    from ship_sim_gym_amd import worldgen, game_map
    for seed in range(6):
        random.seed(seed); np.random.seed(seed)
        rec, polys, goals = worldgen.generate_world((600, 600))
        after = (random.random(), np.random.random_sample())
        random.seed(seed); np.random.seed(seed)
        p2 = np.asarray(game_map.gen_river_poly((600, 600)))
        w = oracle.World(); w.set_banks_only(p2[0], p2[1])
        g2 = []
        for i in range(1, 6):
            y = 100.0 * i + random.randint(-20, 20)
            ok, lo, hi = w.goal_x_range(y)
            assert ok
            g2.append([np.random.uniform(lo, hi), y])
        assert after == (random.random(), np.random.random_sample())
        np.testing.assert_array_equal(polys, p2)
        np.testing.assert_array_equal(goals, np.asarray(g2))
        obs0 = w.reset(p2[0], p2[1], goals)
        assert (rec[native.MAP_OFF_SPAWN_GOAL], rec[native.MAP_OFF_SPAWN_GOAL + 1]) == (obs0[20], obs0[21])
        np.testing.assert_array_equal(rec[native.MAP_OFF_GOALS:native.MAP_OFF_GOALS + 10], goals.reshape(-1))


def test_every_entry_point_cites_what_it_replaces():
    """include/shipsim.h: every exported function's comment names the reference interface it replaces (file:line), or says
    that the reference has no counterpart (memory binding, diagnostics)."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "shipsim.h")).read()
    decls = [(m.start(), m.group(1)) for m in re.finditer(r'^\s*(?:int|const char\s*\*)\s+(ssg_\w+)\s*\(', src, flags=re.M)]
    assert len(decls) >= 26
    section = src.index("Memory binding (caller-owned device memory)")
    assert re.search(r'\w+\.py:\d+', src[section: section + 900])          # the section's own comment cites the reference's state
    prev, missing = 0, []
    for pos, name in decls:
        chunk = src[prev:pos]
        ok = re.search(r'\w+\.py:\d+', chunk) or re.search(r'no reference counterpart|Replaces nothing', chunk, flags=re.I)
        if not ok and name not in ("ssg_state_field", "ssg_bind_state", "ssg_abi_version", "ssg_strerror", "ssg_last_error"):
            missing.append(name)
        prev = pos
    assert not missing, missing


def test_create_limits_and_config4_blob_layout(native):
    """Host-side contract of the round-5 additions, no GPU needed: `map_ring` up to 128 with the 32-bit record-offset guard, and the
    config-4 state blob carrying the memo tables (SSG_F_DYN_MEMO_STATS is where ssg_state_field says, after the queue)."""
    L = native.lib()

    def create(**kw):
        c = native.default_config()
        for k, v in kw.items():
            setattr(c, k, v)
        h = C.c_void_p()
        rc = L.ssg_create(C.byref(c), C.byref(h))
        return rc, h

    rc, h = create(n_envs=4096, map_ring=128, flags=native.FLAG_AUTO_RESET)
    assert rc == 0
    L.ssg_destroy(h)
    assert create(n_envs=4096, map_ring=129)[0] < 0                          # map_ring must be in 2..128
    assert create(n_envs=200000, map_ring=128)[0] < 0                        # 200 000 * 128 * 145 doubles: offsets would not fit 32 bits
    assert b"2^31" in L.ssg_last_error(None)
    rc, h1 = create(n_envs=1024, n_ships=1)
    rc4, h4 = create(n_envs=1024, n_ships=4)
    assert rc == 0 and rc4 == 0
    n1, n4 = C.c_size_t(), C.c_size_t()
    L.ssg_state_nbytes(h1, C.byref(n1)); L.ssg_state_nbytes(h4, C.byref(n4))
    assert n4.value - n1.value > 30 << 20                                    # the memo tables (~35 MB) live in the caller's blob
    off, es, nc, st = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
    assert L.ssg_state_field(h4, native.F_DYN_MEMO_STATS, C.byref(off), C.byref(es), C.byref(nc), C.byref(st)) == 0
    assert es.value == 8 and nc.value == 256 * 16 and off.value % 256 == 0 and off.value + 8 * nc.value < n4.value
    assert L.ssg_state_field(h1, native.F_DYN_MEMO_STATS, C.byref(off), C.byref(es), C.byref(nc), C.byref(st)) < 0   # 1-ship handles have none
    a, b = C.c_uint64(7), C.c_uint64(7)
    assert L.ssg_debug_dyn_counters(h4, C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (0, 0)
    L.ssg_destroy(h1); L.ssg_destroy(h4)
