"""The N>1 PRODUCT path on hardware: two OS processes, each running sharding.make_sharded_env on the HIP path
(tests/shard_worker.py), against one unsharded HIP run — bit for bit, including the all-reduced episode counters.
With one visible device both ranks share cuda:0 and the bank broadcast goes over gloo through host memory; with
>= 2 devices the ranks use RCCL ("nccl").  Also: bench.py's own N-rank launcher on the GPU box."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(tmp_path, world, total, K, n_ships):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), str(world), str(port),
                               str(total), str(K), str(tmp_path), str(n_ships)]) for r in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world, "shard workers failed: %r" % rcs
    return [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("n_ships,total,K", [(1, 1000, 150), (4, 384, 60)])
def test_two_process_sharded_env_equals_unsharded(tmp_path, native, n_ships, total, K):
    import torch
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    r = _run_ranks(tmp_path, 2, total, K, n_ships)
    # with >= 2 devices the ranks MUST have taken the RCCL branch (one process per GPU), and every rank must have seen the
    # same world: a rank that fell back, or a group of the wrong size, fails here rather than passing on the gloo proxy
    want_backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    for rk in r:
        assert int(rk["seen_world"][0]) == 2 and str(rk["seen_backend"]) == want_backend and int(rk["backend"][0]) == (want_backend == "nccl"), \
            (int(rk["seen_world"][0]), str(rk["seen_backend"]), int(rk["n_dev"][0]))
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, total // 2, total // 2, total)
    full = ShipVecEnv(total, n_maps=16, map_seed=1000, n_ships=n_ships)
    np.testing.assert_array_equal(r[1]["bank"], full.bank.cpu().numpy())  # the broadcast delivered rank 0's bank
    np.testing.assert_array_equal(r[0]["bank"], r[1]["bank"])
    obs0 = full.reset_tensor().cpu().numpy().copy()
    np.testing.assert_array_equal(np.concatenate([r[0]["obs0"], r[1]["obs0"]]), obs0)
    acts = full.random_actions(4242, 0, 2 * K)
    for k in range(K):
        o, rew, d, f = full.step_tensor(acts[k])
        np.testing.assert_array_equal(np.concatenate([r[0]["rews"][k], r[1]["rews"][k]]), rew.cpu().numpy())
        np.testing.assert_array_equal(np.concatenate([r[0]["dones"][k], r[1]["dones"][k]]), d.cpu().numpy())
    o, rew, d, f = full.rollout_tensor(acts[K:])
    torch.cuda.synchronize()
    for key, t in (("obs", o), ("rew", rew), ("done", d), ("flags", f)):
        np.testing.assert_array_equal(np.concatenate([r[0][key], r[1][key]]), t.cpu().numpy(), err_msg=key)
    np.testing.assert_array_equal(np.concatenate([r[0]["x"], r[1]["x"]]), full.field(native.F_X).cpu().numpy())
    st = full.stats()
    want = np.array([st["sum_return"], st["sum_length"], st["episodes"], st["goals_hit"]])
    assert st["episodes"] > 20
    np.testing.assert_array_equal(r[0]["glob"], r[1]["glob"])              # every rank holds the job-wide counters
    np.testing.assert_allclose(r[0]["glob"], want, rtol=0, atol=1e-9)      # ... and they are the unsharded run's
    np.testing.assert_allclose(r[0]["local"] + r[1]["local"], want, rtol=0, atol=1e-9)
    assert r[0]["local"][2] > 0 and r[1]["local"][2] > 0


def test_bench_launcher_on_this_box(tmp_path):
    """`python bench.py --gpus 2` (no torchrun around it) must start 2 ranks itself.  On a box with >= 2 devices
    the line says n_gpus 2; on a 1-GPU box it must FAIL LOUDLY (non-zero exit, no JSON line) rather than print N=1."""
    import torch
    n_dev = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "20",
                        "--repeats", "2", "--envs-per-gpu", "4096", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if n_dev >= 2:
        assert p.returncode == 0, p.stderr[-2000:]
        j = json.loads(lines[-1])
        assert j["n_gpus"] == 2 and j["config"]["total_envs"] == 8192 and j["scaling"] == "weak"
    else:
        assert p.returncode != 0 and not lines, (p.returncode, p.stdout[-500:])
        assert "needs device" in p.stderr


def test_bench_multi_rank_control_flow_on_one_device(tmp_path):
    """The N>1 branch of bench.py (barriers, MAX over ranks, the rank table, the c5_full leg) exercised on whatever this
    box has: two ranks share device 0 over gloo (SSG_BENCH_SHARE_DEVICE / SSG_BENCH_BACKEND, test-only overrides).  The
    line says so in `data`; the timings mean nothing, the structure must be the one the driver's 8-GPU run prints."""
    env = dict(os.environ, SSG_BENCH_SHARE_DEVICE="1", SSG_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--repeats", "1", "--envs-per-gpu", "4096", "--c5-full", "--no-cpu-baseline"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                   # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "TEST RUN" in j["data"]
    assert j["config"]["total_envs"] == 8192
    ranks = j["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["env_id_base"] for r in ranks] == [0, 4096]
    c5 = j["other_configs"]["c5_full"]
    assert c5["n_gpus"] == 2 and c5["total_envs"] == 2 * 131072 and c5["env_steps_per_s"] > 0
    assert "cpu_baseline" not in j or j["cpu_baseline"] is None or j["n_gpus"] == 2
