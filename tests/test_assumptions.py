"""The oracle's named, unverifiable assumptions (pymunk is absent: "parity unpinned"), each behind a switch, quantified
on a small run here and on > 10 M env-steps per config in profiles/r2/assumption_sensitivity.md (tools/
assumption_sensitivity.py).  What these tests pin: (a) the SAT predicate the oracle and the HIP kernels use for
`collide_ship` agrees with the restated cpCollide (GJK/EPA + ContactPoints: Chipmunk fires `begin` iff it pushes >= 1
contact) on every player pair, in both a/b orders; (b) which assumptions the observable stream is insensitive to
(touching `<=` vs `<`, GJK cold vs warm, solver order) and which it is not (THRUST_PX0, and — rarely — which poly
of a poly-poly pair is "a")."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def bank(native):
    from ship_sim_gym_amd import worldgen
    return worldgen.build_bank(64, (600, 600))


def _pair(oracle, bank, n_traffic, n_beams, n, K, census=False, **over):
    import assumption_sensitivity as A
    _, polys, goals = bank
    base = oracle.default_config(n_beams=n_beams, n_traffic=n_traffic)
    var = oracle.default_config(n_beams=n_beams, n_traffic=n_traffic, **over)
    return A.run_pair(oracle, base, var, polys, goals, n, K, oracle.max_threads(), census=census)


@pytest.mark.parametrize("n_traffic,n_beams,n,K", [(0, 8, 2048, 300), (3, 10, 768, 300)])
def test_sat_predicate_agrees_with_restated_cpcollide_on_player_pairs(oracle, bank, n_traffic, n_beams, n, K):
    r = _pair(oracle, bank, n_traffic, n_beams, n, K, census=True, variant=oracle.VAR_CHECK_SAT)
    c = r["census"]
    assert c["checked"] > 10000                                   # the census saw real near-bank / near-traffic pairs
    assert c["disagree_ab"] == 0 and c["disagree_ba"] == 0        # begin() fires exactly when SAT says "touching"
    assert r["obs_mismatch_frac"] == 0 and r["done_mismatch_frac"] == 0   # the census itself changes nothing
    # and driving `colliding` from cpCollide's contact count gives the same stream
    r2 = _pair(oracle, bank, n_traffic, n_beams, n, K, variant=oracle.VAR_PLAYER_CPCOLLIDE)
    assert r2["obs_mismatch_frac"] == 0 and r2["done_mismatch_frac"] == 0 and r2["reward_mismatch_frac"] == 0


def test_insensitive_assumptions(oracle, bank):
    for var in (oracle.VAR_TOUCH_STRICT, oracle.VAR_GJK_WARM, oracle.VAR_ORDER_REVERSED):
        r = _pair(oracle, bank, 3, 10, 768, 300, variant=var)
        assert r["obs_mismatch_frac"] == 0 and r["done_mismatch_frac"] == 0 and r["reward_mismatch_frac"] == 0, var
    r = _pair(oracle, bank, 0, 8, 2048, 300, variant=oracle.VAR_TOUCH_STRICT)
    assert r["obs_mismatch_frac"] == 0 and r["done_mismatch_frac"] == 0


def test_sensitive_assumptions_are_the_documented_ones(oracle, bank):
    # which poly is "a" in a poly-poly pair: rare but real (ship 1 leaving the left bank)
    r = _pair(oracle, bank, 3, 10, 2048, 400, variant=oracle.VAR_SWAP_AB)
    assert r["first_divergence_per_env_step"] < 1e-4
    # THRUST_PX0: the highest-risk assumption changes almost every trajectory
    r = _pair(oracle, bank, 0, 8, 1024, 200, thrust_px0=10.0, thrust_py0=22.5)
    assert r["envs_diverged_frac"] > 0.5


def test_committed_table_matches_the_tool_output_format():
    p = os.path.join(ROOT, "profiles", "r2", "assumption_sensitivity.json")
    import json
    j = json.load(open(p))
    for cfg in ("c3", "c4"):
        census = [v["census"] for k, v in j[cfg].items() if "census" in v][0]
        assert census["disagree_ab"] == 0 and census["disagree_ba"] == 0 and census["checked"] > 500000
        assert [v for v in j[cfg].values()][0]["env_steps"] >= 10_000_000
