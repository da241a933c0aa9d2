"""Replays golden streams captured from the genuine pymunk reference (tools/capture_pymunk_golden.py) on the oracle.

No such capture exists yet: pymunk 5.4.0 / pygame / gym are not installable in the build image (no network), so the
physics parity is UNPINNED (oracle/ssg_oracle.h) and this test skips, saying so.  Dropping a capture made elsewhere
into tests/golden/pymunk_streams.npz turns it into the pin: observations within BASELINE's 1e-5, reward / done exact."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pymunk_streams.npz")


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="parity unpinned: no capture from the genuine pymunk reference "
                    "(run tools/capture_pymunk_golden.py where pymunk==5.4.0 installs)")
def test_oracle_matches_pymunk_capture():
    from oracle import oracle as O
    z = np.load(GOLDEN, allow_pickle=True)
    assert not bool(z["meta/dry_run"]) if "meta/dry_run" in z.files else True, "a --dry-run output is not a capture"
    assert "standin" not in str(z["meta/pymunk_version"]), "recorded under the stand-in pymunk: not a capture"
    _replay(O, z)


def _replay(O, z):
    names = sorted({k.split("/")[0] for k in z.files if not k.startswith("meta/")})
    assert names
    for name in names:
        traffic = name.endswith("_traffic")
        obs, rew, done = z[name + "/obs"], z[name + "/reward"], z[name + "/done"]
        acts, starts = z[name + "/actions"], list(z[name + "/episode_start"])
        nb = obs.shape[1] // 2 - 6
        first = np.asarray(z[name + "/reset_obs"][0], dtype=np.float64)
        bounds = (1000.0, 1000.0) if "training" in name else (600.0, 600.0)
        cfg = O.default_config(width=bounds[0], height=bounds[1], dt=(30 if "training" in name else 10) * 0.1, n_beams=nb,
                               n_traffic=3 if traffic else 0)
        w = O.World(cfg)
        ep = -1
        for k, a in enumerate(acts):
            if ep + 1 < len(starts) and starts[ep + 1] == k:
                ep += 1
                polys = np.asarray(z[name + "/polys"][ep], dtype=np.float64).reshape(2, 12, 2)
                o0 = w.reset(polys[0], polys[1], z[name + "/goals"][ep])
                np.testing.assert_allclose(o0, z[name + "/reset_obs"][ep], atol=1e-5, rtol=0, err_msg=name + " reset obs")
            o, r, d = w.step(int(a))
            np.testing.assert_allclose(o, obs[k], atol=1e-5, rtol=0, err_msg="%s step %d" % (name, k))
            assert r == rew[k] and d == bool(done[k]), "%s step %d: reward/done" % (name, k)
        assert first.shape[0] == obs.shape[1]


REFERENCE = os.environ.get("SHIP_SIM_GYM", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "ship_gym")), reason="build container only: the capture script runs the "
                    "reference's own Python, which does not travel to the GPU box")
def test_capture_script_dry_run(tmp_path):
    """The capture script itself must not rot while no machine with pymunk is at hand: --dry-run drives every scenario of it
    (with and without add_default_traffic) through the reference's unmodified ShipEnv under the stand-in modules, and the
    file it writes has the layout the replay above reads — replayed here on the oracle (trivially equal: the stand-in's
    physics IS the oracle; this checks the script and the replay code, and is no parity evidence).  It refuses to write the
    golden path, and a non-dry run refuses the stand-in."""
    import subprocess
    import sys
    from oracle import oracle as O
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "capture_pymunk_golden.py")
    out = str(tmp_path / "dry.npz")
    env = dict(os.environ, SDL_VIDEODRIVER="dummy")
    r = subprocess.run([sys.executable, tool, "--reference", REFERENCE, "--dry-run", "--dry-steps", "25", "--out", out],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "DRY RUN, NOT A CAPTURE" in r.stdout
    z = np.load(out, allow_pickle=True)
    assert bool(z["meta/dry_run"]) and "standin" in str(z["meta/pymunk_version"])
    names = sorted({k.split("/")[0] for k in z.files if not k.startswith("meta/")})
    assert len(names) == 16 and sum(n.endswith("_traffic") for n in names) == 8
    _replay(O, z)
    r2 = subprocess.run([sys.executable, tool, "--reference", REFERENCE, "--dry-run"], capture_output=True, text=True, env=env, timeout=60)
    assert r2.returncode != 0 and "never writes the golden file" in (r2.stderr + r2.stdout)
    assert not os.path.exists(GOLDEN)
