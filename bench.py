#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched ShipEnv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1, launched by torch.distributed.run); every rank steps
its own shard of envs with NO data-path collective (envs are independent) after one RCCL broadcast
of the map bank from rank 0.  A "step" is one ssg_step pass (one kernel launch) over the rank's whole env batch
with a pre-generated random action vector already resident in HBM; obs / reward / done are written to HBM every
step and done envs are auto-reset in-kernel.  Workload at N=1: BASELINE.json configs[2] — 65 536 parallel envs,
1 ship, 8-beam lidar, default 600x600 map bank (64 maps), SPEED 10 — the configuration the ">= 10 M env-steps/s
on one MI355X" target is quoted on.  Weak scaling: every rank owns 65 536 envs.

Rank 0 prints ONE JSON line (see the driver contract) with two extra objects:
  roofline     — algorithmic HBM bytes per launch (675 B/env-step at S=1, nb=8, H=2) divided by the
                 kernel's average launch duration measured with HIP events on the launch stream, against 8 TB/s.
  cpu_baseline — the CPU oracle ("port": our C restatement of the reference path, NOT pymunk) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
N_BEAMS = 8
N_MAPS = int(os.environ.get("SSG_BENCH_MAPS", "64"))  # BASELINE workload: 64; override only for experiments
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(n_ships, nb, hist):
    """SURVEY.md §8(d) per env-step figure (not to be redefined)."""
    return 96 * n_ships + 8 + 8 + 2 + 80 + 4 + 4 + 16 * nb + 8 * (6 + nb) + 8 * hist * (6 + nb) + 9


def cpu_baseline(vec, seconds_target=6.0):
    """Time the CPU oracle on a bounded sample of the same workload: same bank, same Philox action stream."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_cfg
    from oracle import oracle as O
    threads = O.max_threads()
    n = 16384
    ob = O.Batch(n, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n) % vec.n_maps)
    ob.reset()
    t0 = time.perf_counter()
    ob.rollout(12345, 0, 20, n_threads=threads)  # calibration
    per_step = (time.perf_counter() - t0) / 20
    K = max(20, min(100000, int(seconds_target / max(per_step, 1e-9))))
    ob.reset()
    t0 = time.perf_counter()
    done_steps = ob.rollout(12345, 0, K, n_threads=threads)
    dt = time.perf_counter() - t0
    return {"value": done_steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d steps, 8-beam lidar, same 64-map bank and Philox action stream, OpenMP over envs, "
                      "%.1f s" % (n, K, dt)}


def measured_copy_gbps(dev, stream_ptr=None):
    """SURVEY.md §8(d) "measured roofline": our own 8-byte-per-lane device-to-device copy (the step kernel's access
    width) over 2 x 1 GiB on the same GPU in the same run; returns (read + write) GB/s, best of 5."""
    import ctypes as C
    import torch
    from ship_sim_gym_amd import _native as N
    nd = (1 << 30) // 8
    a = torch.ones(nd, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    L = N.lib()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = 0.0
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N.check(L.ssg_debug_copy8(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nd, sp), None, "copy8")
        e1.record()
        torch.cuda.synchronize()
        if it > 0:
            best = max(best, 2.0 * nd * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true", help="skip the informational one-launch-per-step timing")
    ap.add_argument("--workload", choices=("c3", "c4"), default="c3",
                    help="c3 (default, the BASELINE metric's config): 1 ship, 8 beams; c4: BASELINE configs[3], 4 ships "
                         "(traffic + dynamic goals + contact solver), 10 beams — informational, not the headline line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    from ship_sim_gym_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    n = args.envs_per_gpu
    c4 = args.workload == "c4"
    n_beams = 10 if c4 else N_BEAMS
    vec = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=n_beams,
                     env_id_base=rank * n, exact_lidar=bool(int(os.environ.get("SSG_EXACT_LIDAR", "0"))),
                     n_ships=4 if c4 else 1)
    if os.environ.get("SSG_ABLATE"):  # timing-only development aid (needs a -DSSG_ABLATION build)
        import ctypes as C
        vec.cfg.flags |= int(os.environ["SSG_ABLATE"], 0) << 16
        from ship_sim_gym_amd import _native as N
        N.lib().ssg_destroy(vec._h)
        N.check(N.lib().ssg_create(C.byref(vec.cfg), C.byref(vec._h)), None, "ssg_create")
        N.check(N.lib().ssg_bind_state(vec._h, C.c_void_p(vec.state.data_ptr())), vec._h, "bind")
        vec.set_bank(vec.bank)
    if use_dist:
        sharding.broadcast_bank(vec, src=0)  # RCCL broadcast of the map bank over xGMI; the only collective on the path

    K, W = args.steps, args.warmup
    acts = vec.random_actions(12345, 0, K + W)  # [K+W, n] int32, generated on device before the timed region
    vec.reset_tensor()
    vec.rollout_tensor(acts[:W]) if W > 0 else None
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    vec.rollout_tensor(acts[W:])  # K launches of the step kernel on torch's current stream
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    # for information: the policy-in-the-loop path, one ssg_step launch per step (not the headline number)
    single_us = None
    if world == 1 and not args.no_single_step:
        ks = min(K, 500)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(ks):
            vec.step_tensor(acts[W + k])
        e1.record()
        torch.cuda.synchronize()
        single_us = e0.elapsed_time(e1) * 1e3 / ks
    if use_dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank defines the job's time
        wall = float(t.item())

    if rank == 0:
        total_steps = float(n) * world * K
        B = algorithmic_bytes(4 if c4 else 1, n_beams, 2)
        # ssg_rollout fuses SSG_ROLLOUT_STEPS_PER_LAUNCH (100) steps into each launch of the step kernel:
        # algorithmic bytes per launch = B * n * steps_per_launch, launch duration = HIP-event time / launches
        # (config 4 has no fused rollout: one "launch" below is one step = dyn classify + dyn step + step kernel)
        spl = 1 if c4 else int(os.environ.get("SSG_FUSE", "100"))
        n_launch = (K + spl - 1) // spl
        launch_s = ev_ms * 1e-3 / n_launch
        achieved = B * n * (K / n_launch) / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and not c4:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")  # rocprofv3 PMC, same command (profiles/)
            except Exception:
                traffic = None
        copy_gbps = measured_copy_gbps(dev)
        out = {
            "metric": "env steps/sec (batched ShipEnv)", "value": total_steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3]: 65536 parallel envs per GPU x 4 ships (3 traffic ships, dynamic "
                                    "goal bodies, Chipmunk contact solver), 10-beam lidar, 64-map bank, random Philox "
                                    "actions, auto-reset in-kernel") if c4 else
                                   ("BASELINE configs[2]: 65536 parallel envs per GPU, 1 ship, 8-beam lidar, 64-map "
                                    "bank (600x600, SPEED 10), random Philox actions, auto-reset in-kernel"),
                       "envs_per_gpu": n, "total_envs": n * world, "n_beams": n_beams, "history": 2,
                       "parallelism": "env-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps,
                         "kernel": ("ssg::dyn_classify_kernel + ssg::dyn_step_kernel + ssg::step_kernel<10, 256, true, false, true>"
                                    if c4 else "ssg::step_kernel<8, 256, true, false, false>"),
                         "algorithmic_bytes_per_env_step": B,
                         "steps_per_launch": K / n_launch, "avg_launch_us": launch_s * 1e6,
                         "us_per_step_in_launch": launch_s * 1e6 * n_launch / K},
            "single_step_launch_us": single_us,
        }
        if world == 1 and not args.no_cpu_baseline and not c4:
            out["cpu_baseline"] = cpu_baseline(vec)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
