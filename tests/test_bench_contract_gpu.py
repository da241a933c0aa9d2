"""The driver's contract for bench.py (one JSON line on stdout at N = 1), checked on a small, quick run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_carries_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--envs-per-gpu", "8192", "--no-other-configs"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["higher_is_better"] is True
    assert j["unit"] == "env-steps/s" and j["dtype"] == "f64" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    # value = envs x steps / the timed region, whole job
    assert abs(j["value"] - 8192 * 20 / (j["ms_per_step"] * 20 * 1e-3)) <= 1e-6 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_bytes_per_env_step"] == 675 and r["avg_launch_us"] > 0
    # what binds, without fractions above 1: the 675 B formula over the wall time `value` uses, and the bytes the FUSED API
    # must move (outputs 234 + action 4 + the state columns once per 20-step launch)
    for k in ("frac_wall", "fused_compulsory_bytes_per_env_step", "frac_fused_compulsory", "measured_copy_GBps"):
        assert k in r, k
    assert "frac_of_measured_copy" not in r
    assert abs(r["frac_wall"] - 675 * j["value"] / 1e9 / 8000.0) < 1e-9
    assert abs(r["fused_compulsory_bytes_per_env_step"] - (234 + 4 + 2 * ((7 + 8) * 8 + 17) / 20.0)) < 1e-9
    assert 0 < r["frac_fused_compulsory"] < 1 and r["frac_wall"] <= r["frac"] * 1.0001
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["single_thread"]["cores"] == 1
    assert j["single_step_launch_us"] > 0 and len(j["single_step_launch_us_repeats"]) >= 5
    # the timed repeats: spread, the shader clock the step kernel recorded in each, and what preceded them (all untimed, all declared)
    assert len(j["repeats_ms"]) == j["repeats"] == len(j["repeats_shader_clock_ghz"]) == len(j["repeats_event_ms"])
    assert j["repeats_min_ms"] <= j["repeats_median_ms"] <= j["repeats_max_ms"]
    assert abs(j["repeats_median_ms"] - j["ms_per_step"] * 20) < 1e-9 and j["value_min"] <= j["value"] <= j["value_max"]
    assert all(0.5 < g < 3.0 for g in j["repeats_shader_clock_ghz"]), j["repeats_shader_clock_ghz"]   # GHz, from s_memtime / s_memrealtime
    assert j["preconditioning_ms"] >= 300.0 and j["burn_in_steps"] == 1000 and "scratch" in j["preconditioning"]
