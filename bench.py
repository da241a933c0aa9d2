#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched ShipEnv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no torch.distributed environment (RANK unset) this process is only a
LAUNCHER: it touches no GPU, starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
(the same command line the driver uses) and exits with the child's code.  A rank started by torch.distributed.run
initialises RCCL (backend "nccl"), refuses to run when WORLD_SIZE != --gpus or when the node shows fewer devices
than ranks, and steps its own shard of envs with NO data-path collective (envs are independent)
after one RCCL broadcast of the map bank from rank 0.  Weak scaling: every rank owns 65 536 envs; `n_gpus` in the
JSON line is the RCCL world size the ranks saw.

A "step" is one pass of the hot path over the rank's whole env batch (ssg_rollout_traj: fused launches of
ssg::step_kernel) with a pre-generated random action tensor already resident in HBM.  The rollout runs in TRAJECTORY
mode: every step's obs / reward / done / flags land in their own slot of [K, N, ...] tensors (what the reference's rollout
loop consumes, train/random.py:14-27; SURVEY.md 8d: "obs/reward/done materialised in HBM every step"), so the
233 B per env-step of outputs physically reach HBM instead of being rewritten in place inside the 256 MiB Infinity
Cache; done envs are auto-reset in-kernel.  Workload at N=1: BASELINE.json configs[2] — 65 536 parallel envs, 1 ship,
8-beam lidar, default 600x600 map bank (64 maps), SPEED 10 — the configuration the ">= 10 M env-steps/s on one
MI355X" target is quoted on.  With --gpus 8 rank 0 also times BASELINE configs[4] (131 072 envs per rank x 8 =
1 048 576 envs, 10 beams) as `other_configs.c5_full` (all ranks step, MAX over ranks).

TEST-ONLY overrides (tests/test_sharding_gpu.py; never set by the driver): SSG_BENCH_SHARE_DEVICE=1 puts every rank
on device 0 and SSG_BENCH_BACKEND=gloo replaces RCCL, so the N>1 control flow can be exercised on a 1-GPU box; the
line's `data` field then says "TEST RUN ... timings meaningless".

Timing: the timed env is reset and stepped `--burn-in-steps` (default 1 000, `burn_in_steps`) untimed random-action steps, so that
its episodes are no longer phase-locked; then an untimed, time-based device conditioning (`--precondition-ms`, default 300: the same K-step rollout on a SCRATCH env and
scratch buffers, bracketed by synchronize exactly like a timed repeat and repeated until the time is up, reported as
`preconditioning_ms`; every trajectory buffer set is written once beforehand), then W untimed warm-up steps, then the K-step
rollout is timed `--repeats` (default 5) times (`repeats_min_ms` / `repeats_median_ms` / `repeats_max_ms`; the shader clock the
step kernel recorded during each repeat in `repeats_shader_clock_ghz`: no probe kernel runs between repeats), every repeat
bracketed by barrier + torch.cuda.synchronize() on both sides (a rank's interval runs from the opening barrier + synchronize
to its own closing synchronize; the closing barrier follows) and reduced with MAX over ranks; `value` is the
MEDIAN repeat, all repeats are listed in `repeats_ms`.

Rank 0 prints ONE JSON line (see the driver contract) with extra objects:
  roofline      — `achieved` / `frac`: algorithmic HBM bytes per launch (675 B/env-step at S=1, nb=8,
                  H=2; formula not redefined) divided by the step kernel's average launch duration in the median
                  repeat, measured live with HIP events on the launch stream, against 8 TB/s.  `traffic`,
                  `hbm_measured` and `valu_issue_frac` come from the rocprofv3 PMC run committed under profiles/
                  (`counters_source`): PMC counters cannot be read from inside this process, so they are DERIVED from
                  that stored profile and reported only when the profile was taken on this very kernel source
                  (sha256 of csrc/ + include/ recorded in the profile == the tree's); otherwise they are null.
                  `bound` is what those counters say binds the kernel, not a label.
  other_configs — informational (N=1 only, outside the timed region): BASELINE configs[1] (4 096 envs, 10 beams),
                  configs[3] (65 536 envs x 4 ships), the per-GPU share of configs[4], configs[2] with a brand-new world
                  per episode (map_mode="fresh_device"), and the one-launch-per-step (policy-in-the-loop) path.
  config.ranks  — per rank: device, HIP ordinal, the RCCL world size it saw, its env range and the HSA / HIP / NCCL / RCCL /
                  rendezvous environment it ran under (a rank that cannot join the job prints the same to stderr, exit 4).
  cpu_baseline  — the CPU oracle ("port": our C restatement of the reference path, NOT pymunk) timed on this box's
                  host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import hashlib

ENVS_PER_GPU = 65536
N_BEAMS = 8
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0  # 256 CUs x 4 SIMDs, one FP64 wave64 VALU instruction per 4 cycles at 2.4 GHz
TRAJ_RING_BYTES = 48 << 30            # at most this much HBM for the bench's trajectory buffers
N_MAPS = int(os.environ.get("SSG_BENCH_MAPS", "64"))  # BASELINE workload: 64; override only for experiments
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(n_ships, nb, hist):
    """SURVEY.md §8(d) per env-step figure (not to be redefined)."""
    return 96 * n_ships + 8 + 8 + 2 + 80 + 4 + 4 + 16 * nb + 8 * (6 + nb) + 8 * hist * (6 + nb) + 9


def source_sha():
    """sha256 over the kernel / ABI sources: ties a stored PMC profile to the code it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ship_sim_gym_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "shipsim.h"), "rb").read())
    return h.hexdigest()[:16]


def traj_buffers(vec, K, n_bufs):
    """n_bufs sets of trajectory tensors (obs [K,N,D], reward / done / flags [K,N]) in HBM."""
    import torch
    n, D, dev = vec.num_envs, vec.states_history, vec.device
    return [(torch.empty((K, n, D), dtype=torch.float64, device=dev), torch.empty((K, n), dtype=torch.float64, device=dev),
             torch.empty((K, n), dtype=torch.uint8, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev))
            for _ in range(n_bufs)]


def traj_bytes_per_step(n, D):
    return n * (8 * D + 8 + 1 + 1)


def state_bytes_per_env(nb):
    """The state columns one launch reads (first step) and writes back (last step): 7 + nb f64, 4 i32, the goal mask."""
    return 2 * ((7 + nb) * 8 + 4 * 4 + 1)


def fused_compulsory_bytes(nb, D, steps_in_launch):
    """Compulsory HBM bytes per env-step of the FUSED rollout API (ssg_rollout_traj): every step's outputs (obs row, reward,
    done, flags) and its action, plus the state columns once per launch.  The honest HBM denominator of a kernel that keeps
    the state in registers across the fused steps (the 675 B of SURVEY 8d charge every step a state read + write and a
    re-read of the previous frame)."""
    return traj_bytes_per_step(1, D) + 4 + state_bytes_per_env(nb) / float(steps_in_launch)


def steps_per_launch_cfg():
    return max(1, int(os.environ.get("SSG_FUSE", "100")))  # SSG_ROLLOUT_STEPS_PER_LAUNCH (include/shipsim.h)


def cpu_baseline(vec, seconds_target=10.0):
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
    # "1 thread and os.cpu_count() threads" — the scalar figure on a ~3 s sample of the same stream
    n1 = 1024
    o1 = O.Batch(n1, oracle_cfg(O, vec), vec.bank_polys, vec.bank_goals, map_ids=np.arange(n1) % vec.n_maps)
    o1.reset()
    t1 = time.perf_counter()
    o1.rollout(12345, 0, 50, n_threads=1)
    K1 = max(50, min(100000, int(3.0 / max((time.perf_counter() - t1) / 50, 1e-9))))
    o1.reset()
    t1 = time.perf_counter()
    s1 = o1.rollout(12345, 0, K1, n_threads=1)
    d1 = time.perf_counter() - t1
    return {"value": done_steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d steps, 8-beam lidar, same 64-map bank and Philox action stream, OpenMP over envs, "
                      "%.1f s" % (n, K, dt),
            "single_thread": {"value": s1 / d1, "cores": 1, "sample": "%d envs x %d steps, %.1f s" % (n1, K1, d1)}}


def measured_copy_gbps(dev):
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


def event_time_rollout(vec, acts, reps=3):
    """HIP-event time (ms, median of `reps`) of one trajectory-mode ssg_rollout_traj over `acts` on torch's current stream."""
    import torch
    ts = []
    out = traj_buffers(vec, int(acts.shape[0]), 1)[0]
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        vec.rollout_tensor(acts, trajectory=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    del out
    return sorted(ts)[len(ts) // 2]


def side_config(dev, n, n_beams, n_ships, K, W, map_mode="bank", ring=32, dyn_memo=True, kernel_split=False, n_maps=N_MAPS, single_step=False):
    """Informational timing of another BASELINE config on this GPU (outside the headline's timed region)."""
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    vec = ShipVecEnv(n, device=dev, map_mode=map_mode, n_maps=n_maps, map_seed=1000, n_beams=n_beams, n_ships=n_ships, ring=ring,
                     dyn_memo=dyn_memo)
    acts = vec.random_actions(12345, 0, K + W)
    vec.reset_tensor()
    vec.rollout_tensor(acts[:W])
    ms = event_time_rollout(vec, acts[W:])
    B = algorithmic_bytes(n_ships, n_beams, 2)
    us = ms * 1e3 / K
    sps = n * K / (ms * 1e-3)
    out = {"envs": n, "outputs": "trajectory [K,N,...]", "n_beams": n_beams, "n_ships": n_ships, "map_mode": map_mode, "steps": K, "us_per_step": us, "env_steps_per_s": sps,
           "algorithmic_bytes_per_env_step": B, "achieved_GBps": sps * B / 1e9, "frac": sps * B / 1e9 / HBM_PEAK_GBPS}
    epw, in_lds, lds_bytes = vec.launch_geometry()
    out["launch"] = {"envs_per_workgroup": epw, "bank_in_lds": bool(in_lds), "lds_bytes": lds_bytes}
    if n_ships > 1:
        out["dyn_memo"] = bool(dyn_memo and map_mode == "bank")
        if out["dyn_memo"]:
            st = vec.dyn_memo_stats()
            out["memo_lookups"] = {k: st[k] for k in ("hits", "computed", "stored")}
    if single_step:
        # the policy-in-the-loop path at this size: one ssg_step launch per step, median of 5 repeats of 500 back-to-back steps
        import torch
        a1 = vec.random_actions(777, 0, 500)
        rows = [a1[k] for k in range(500)]
        reps = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for a in rows:
                vec.step_tensor(a)
            e1.record()
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) * 1e3 / 500)
        out["single_step_launch_us"] = sorted(reps)[2]
    if kernel_split and n_ships > 1:
        # the two launches of a config-4 step, by HIP events around each (a separate short run: the events and the wait at the end
        # of the call are not part of the figure above)
        vec.kernel_times(True)
        vec.rollout_tensor(acts[W: W + min(K, 100)])
        d_us, s_us, cnt = vec.kernel_times(False)
        out["kernel_split_us"] = {"dyn_step_kernel": d_us, "step_kernel_DYN": s_us, "steps": cnt,
                                  "method": "HIP events around each of the two launches of a step, separate %d-step run" % cnt}
    vec.close()
    del acts
    import torch
    torch.cuda.empty_cache()
    return out


def c4_policy_in_the_loop(dev, n, host_reset, ks=300, terminal_obs=False):
    """Config 4 stepped one ssg_step at a time (what a trainer does), us per step (median of 5 x `ks` back-to-back steps):
    host_reset = False: VecEnv semantics, done envs are reset inside the step kernel; host_reset = True: the RLlib flow on the
    device — no in-kernel auto-reset, ONE masked ssg_reset(mask = done) after every step (its envs join the queue of the next
    full cpSpaceStep), no host synchronisation in the loop."""
    import torch
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    vec = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=10, n_ships=4, auto_reset=not host_reset)
    if terminal_obs:  # the RLlib flow without a reset launch: in-kernel reset + the terminal observations in a side buffer
        vec.enable_terminal_obs()
    a1 = vec.random_actions(4242, 0, ks)
    rows = [a1[k] for k in range(ks)]
    vec.reset_tensor()
    reps = []
    for r in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for a in rows:
            obs, rew, done, flags = vec.step_tensor(a)
            if host_reset:
                vec.reset_tensor(mask=done)   # (default record of each env; the mask is the done tensor the step just wrote)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            reps.append(e0.elapsed_time(e1) * 1e3 / ks)
    steps, rebuilds = vec.dyn_counters()
    vec.close()
    torch.cuda.empty_cache()
    return {"envs": n, "us_per_step": sorted(reps)[len(reps) // 2], "repeats_us": reps,
            "launch": "one ssg_step per step" + (" + one masked ssg_reset" if host_reset else "") +
                      (" (ssg_set_terminal_obs: reset observation in obs, terminal observation in a side buffer)" if terminal_obs else ""),
            "full_cpSpaceSteps": steps, "queue_rebuilds": rebuilds}


def vecenv_numpy_path(dev, n, n_beams, ks=300):
    """The SB / RLlib-facing numpy protocol (train/stable_baselines/ppo.py:122-123: `env.step(actions)` on host arrays): per step,
    actions host -> device, ONE ssg_step, ONE device -> host copy of the packed obs | reward | done | flags block into pinned memory
    (ShipVecEnv.step_async / step_wait).  Wall time per step over `ks` steps (median of 5), next to the PCIe bound of the same
    bytes: the block's device -> host copy and the actions' host -> device copy timed alone on this box."""
    import numpy as np
    import torch
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    vec = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=n_beams)
    acts = vec.random_actions(31337, 0, ks).cpu().numpy().astype(np.int64)  # what a host-side policy hands over
    rows = [acts[k] for k in range(ks)]
    vec.reset()
    for a in rows[:20]:
        vec.step(a)
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        for a in rows:
            vec.step_async(a)
            vec.step_wait()
        reps.append((time.perf_counter() - t0) / ks)
    per_step = sorted(reps)[2]
    # the PCIe bound of exactly these bytes: the two copies alone, HIP events, pinned memory, best of 7
    hs = vec._host_side()
    blk, hact = hs["blocks"][0], hs["acts"]
    best_d2h = best_h2d = 1e9
    for _ in range(7):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(); blk.copy_(vec._out_blob, non_blocking=True); e1.record(); vec._actions.copy_(hact, non_blocking=True); e2.record()
        torch.cuda.synchronize()
        best_d2h, best_h2d = min(best_d2h, e0.elapsed_time(e1) * 1e-3), min(best_h2d, e1.elapsed_time(e2) * 1e-3)
    bytes_down = n * (8 * vec.states_history + 8 + 1 + 1)
    bound = n / (best_d2h + best_h2d)
    out = {"envs": n, "n_beams": n_beams, "steps": ks, "us_per_step": per_step * 1e6, "env_steps_per_s": n / per_step,
           "repeats_us": [r * 1e6 for r in reps],
           "bytes_per_env_step": {"device_to_host": bytes_down / n, "host_to_device": 4},
           "pcie_bound": {"env_steps_per_s": bound, "d2h_us": best_d2h * 1e6, "h2d_us": best_h2d * 1e6,
                          "d2h_GBps": vec._out_nbytes / best_d2h / 1e9,
                          "method": "the packed output block's device -> host copy and the actions' host -> device copy alone, "
                                    "pinned memory, HIP events, best of 7"},
           "frac_of_pcie_bound": (n / per_step) / bound,
           "path": "ShipVecEnv.step_async (pinned actions -> H2D, ssg_step, one D2H of the packed block, side stream) + step_wait "
                   "(event wait, numpy views of the pinned block)"}
    vec.close()
    # the RLlib VectorEnv flow on the same bytes (train/rllib/ppo.py:21-44): vector_step reports terminal observations (a fresh
    # observation array per step: RLlib's sample builders keep rows by reference), every done env gets its reset_at(i) — one Python
    # call per done env, as RLlib's own adapter makes them; no reset launch underneath (ssg_set_terminal_obs)
    rl = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=n_beams, rllib=True)
    rl.vector_reset()
    kr = min(ks, 100)
    for a in rows[:5]:
        _, _, d, _ = rl.vector_step(a)
        for i in np.nonzero(d)[0]:
            rl.reset_at(int(i))
    t0 = time.perf_counter()
    n_resets = 0
    for a in rows[:kr]:
        _, _, d, _ = rl.vector_step(a)
        idx = np.nonzero(d)[0]
        n_resets += len(idx)
        for i in idx:
            rl.reset_at(int(i))
    per_rl = (time.perf_counter() - t0) / kr
    out["rllib_vector_step"] = {"us_per_step": per_rl * 1e6, "env_steps_per_s": n / per_rl, "steps": kr, "reset_at_calls_per_step": n_resets / kr,
                                "path": "ShipVecEnv(rllib=True).vector_step (terminal observations, a fresh obs array per step) + one reset_at(i) "
                                        "per done env; no reset launch (ssg_set_terminal_obs)"}
    rl.close()
    torch.cuda.empty_cache()
    return out


def ppo_end_to_end(dev, n, single_step_us, updates=4, horizon=32):
    """An end-to-end training rate (the reference prints steps per minute, train/stable_baselines/ppo.py:86-96): train/ppo_torch.py —
    a GPU-resident PPO loop, one env step per policy forward — at `n` envs, with its rollout step launched eagerly, captured as ONE
    HIP graph {policy forward + sampling + ssg_step + buffer writes}, and as two half-batch graphs ping-ponging on two streams.
    `env_share_of_rollout` = the env's own one-launch-per-step time (single_step_launch_us) over the rollout's time per step."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("ppo_torch", os.path.join(ROOT, "train", "ppo_torch.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = {"envs": n, "horizon": horizon, "updates": updates, "policy": "MLP 28-64-64-(3+1), fp32, Adam, 2 epochs x 4 minibatches per update",
           "env_single_step_launch_us": single_step_us}
    for mode in ("eager", "graph", "pingpong"):
        mod.train(envs=n, updates=1, horizon=horizon, log=lambda s: None, mode=mode, device=str(dev))  # (first-use costs: hipBLASLt, allocator)
        hist, det = mod.train(envs=n, updates=updates, horizon=horizon, log=lambda s: None, mode=mode, device=str(dev), return_details="timing")
        out[mode] = {"rollout_env_steps_per_s": det["rollout_env_steps_per_s"], "rollout_us_per_step": det["rollout_us_per_step"],
                     "training_env_steps_per_s": det["training_env_steps_per_s"],
                     "env_share_of_rollout": (single_step_us / det["rollout_us_per_step"]) if single_step_us else None}
        del det
        torch.cuda.empty_cache()
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


ENV_PREFIXES = ("HSA_", "HIP_", "ROCR_", "NCCL_", "RCCL_", "GPU_", "MASTER_", "TORCH_NCCL", "CUDA_VISIBLE")
ENV_NAMES = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "SSG_BENCH_SHARE_DEVICE", "SSG_BENCH_BACKEND")


def comm_env(env=None):
    """The environment variables that decide how a rank finds its GPU and its peers (HSA / HIP / ROCR / NCCL / RCCL /
    rendezvous): recorded per rank in the JSON line and printed when a rank fails, so that a failed multi-GPU run can be
    diagnosed from its tail."""
    env = os.environ if env is None else env
    return {k: env[k] for k in sorted(env) if k.startswith(ENV_PREFIXES) or k in ENV_NAMES}


def rank_command(args, argv, port=None):
    """(command line, environment) of the N-rank child job: what the driver itself runs for N > 1."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    argv = [a for a in argv if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port() if port is None else port), os.path.abspath(__file__)] + argv
    return cmd, env


def launch_ranks(args, argv):
    """--gpus N > 1 without a torch.distributed environment: start N ranks as a CHILD job and return its exit code.
    This process never touches a GPU (no HIP call, no torch.cuda query), so nothing GPU-initialised is re-executed."""
    cmd, env = rank_command(args, argv)
    if args.dry_launch:  # what WOULD be started, as one JSON line; nothing runs, no GPU is touched
        print(json.dumps({"dry_launch": True, "n_ranks": args.gpus, "cmd": cmd, "env": comm_env(env),
                          "refusals": {"2": "WORLD_SIZE != --gpus", "3": "fewer HIP devices than ranks"}}))
        return 0
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.write("bench.py: comm env of the ranks: %s\n" % json.dumps(comm_env(env)))
    return subprocess.call(cmd, env=env)


def launch_clock(vec, buf_row):
    """Ask the step kernel to record the shader clock of its own launches (ssg_debug_launch_clock: the first wave of workgroup 0
    stores how far s_memtime and the constant 100 MHz s_memrealtime advanced over its lifetime) into buf_row (2 x int64 on the
    device), or stop (None).  A host-side setting: nothing is launched around the timed region — a separate probe kernel between two
    repeats, however light, leaves the GPU idle long enough for the next launch to start slower (tools/bench_probe_when.sh: 8.0
    against 9.2 G env-steps/s in the driver's form), and one wide enough to load every CU with FP64 work pulls the chip into its
    power-limited clocks (2.31 -> 2.04 GHz)."""
    import ctypes as C
    from ship_sim_gym_amd import _native as N
    N.check(N.lib().ssg_debug_launch_clock(vec._h, C.c_void_p(buf_row.data_ptr()) if buf_row is not None else None), vec._h,
            "ssg_debug_launch_clock")


def bracketed_rollout(vec, a, out, use_dist):
    """ONE repeat: exactly a.shape[0] trajectory-mode steps bracketed by barrier + synchronize on both sides.  Returns (wall
    seconds of this rank, HIP-event ms).  The device conditioning runs this very function on a scratch env, so the timed repeats
    continue the launch / wait pattern the GPU is already in."""
    import torch
    import torch.distributed as dist
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    vec.rollout_tensor(a, trajectory=True, out=out)  # exactly K steps: ceil(K / steps_per_launch) launches of the step kernel
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()  # this rank's K steps are complete; the job's time is the MAX over ranks (below), so the
    if use_dist:              # closing barrier's own latency (an RCCL all-reduce: tens of us against a 150 us region at
        dist.barrier()        # K = 20) is bracketing, not stepping, and stays outside the interval
    return t1 - t0, ev0.elapsed_time(ev1)


def precondition(pvec, pacts, pouts, min_ms, use_dist, dev):
    """Time-based device conditioning BEFORE the warm-up steps: the same K-step rollout, bracketed exactly like a timed repeat
    (bracketed_rollout), on a SCRATCH env and scratch trajectory buffers, repeated for at least `min_ms` of wall time — so that the
    warm-up steps and the timed repeats start on a GPU that is already in the power / clock state this launch-and-wait pattern
    settles in.  (Two things measured in round 6, tools/coldbuf_probe.py and tools/bench_probe_when.sh: after an idle stretch the
    first launches run below the settled rate — BENCH_r05's repeats fell 206 -> 158 us — and after a stretch of back-to-back
    launches with no waits in between, or an FP64 burst on every CU, they do too: the chip is then at its power-limited clocks
    and takes two or three 150-us repeats to come back.)  Nothing it touches is read or written by the timed region.  The number
    of rounds is agreed between the ranks (their barriers must pair up)."""
    import torch
    import torch.distributed as dist
    if min_ms <= 0:
        return 0.0, 0
    pvec.reset_tensor()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rounds = 0
    while True:  # chunks of 16 bracketed rollouts until EVERY rank has run for min_ms (the ranks' barriers must pair up)
        for _ in range(16):
            bracketed_rollout(pvec, pacts, pouts[rounds % len(pouts)], use_dist)  # (rotating over as many sets as the repeats do)
            rounds += 1
        more = (time.perf_counter() - t0) * 1e3 < min_ms
        if use_dist:
            t = torch.tensor([1 if more else 0], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            more = bool(int(t.item()))
        if not more:
            break
    return (time.perf_counter() - t0) * 1e3, rounds


def timed_rollouts(vec, K, W, R, use_dist, dev, precondition_ms=0.0, pvec=None, burn_in=0):
    """Device conditioning (untimed, scratch env and buffers), W untimed warm-up steps, then R timed repeats of exactly K
    trajectory-mode steps, each bracketed by barrier + synchronize on both sides; returns (wall seconds per repeat, MAX over ranks;
    HIP-event ms per repeat; buffer sets; the shader clock GHz the step kernel itself recorded in every repeat; preconditioning info)."""
    import torch
    import torch.distributed as dist
    acts = vec.random_actions(12345, 0, W + K * R)  # int32 [W + R*K, n], generated on device before any timed region
    per_set = traj_bytes_per_step(vec.num_envs, vec.states_history) * K
    n_bufs = max(1, min(R, int(TRAJ_RING_BYTES // per_set)))
    bufs = traj_buffers(vec, K, n_bufs)  # distinct slots per repeat: a repeat never rewrites lines still in the Infinity Cache
    for set_ in bufs:                     # pre-touch: no repeat is the first writer of its buffer set's pages,
        for t in set_:                    # nor the first call that hands this set to rollout_tensor (its checked slow path)
            t.zero_()
        vec.rollout_tensor(acts[:1], trajectory=True, out=set_)  # (before the reset below: the env starts over afterwards)
    wbuf = bufs[0]
    if n_bufs > 1:  # the warm-up steps get a set of their own: no repeat finds lines of ITS set still in the Infinity Cache
        wbuf = traj_buffers(vec, K, 1)[0]
        for t in wbuf:
            t.zero_()
        vec.rollout_tensor(acts[:1], trajectory=True, out=wbuf)
    clk = torch.zeros((R, 2), dtype=torch.int64, device=dev)  # the step kernel's own clock stamps, one row per timed repeat
    # The interpreter's cyclic garbage collector stays out of the conditioning and the timed repeats: collected and switched off HERE,
    # before anything is conditioned.  (Round 6 first did this between the conditioning and the warm-up steps: a gc.collect() with
    # torch loaded takes tens of milliseconds, the GPU idled through it, and every timed repeat then ran at the 2.13 GHz of a GPU
    # that has just woken up instead of the 2.39 GHz the conditioning had brought it to — 8.7 instead of 9.3 G env-steps/s.)
    import gc
    gc.collect()
    gc.disable()
    vec.reset_tensor()
    # Burn-in of the timed env's STATE (untimed, declared as `burn_in_steps`): a freshly reset batch is phase-locked — every env
    # starts its first episode in the same step, so steps 5-25 (no ship within lidar range of a bank yet) are ~4 % cheaper and
    # steps 25-45 (every ship reaches the first goal and the banks at once) ~5 % dearer than the stationary mix a long rollout
    # runs in (tools/phase_probe.py; it is what made the second of the driver form's five repeats the slowest in every run).
    # `burn_in` random-action steps (~16 episodes per env at the default 1 000) desynchronise the episodes first.
    for b0 in range(0, burn_in, 200):  # (whole 100-step launches: a kernel trace of this command averages over equal launches)
        vec.rollout_tensor(vec.random_actions(777, b0, min(200, burn_in - b0)))
    pre = {"preconditioning_ms": 0.0, "preconditioning_launches": 0}
    if precondition_ms > 0 and pvec is not None:
        pacts = pvec.random_actions(4321, 0, K)
        pouts = traj_buffers(pvec, K, n_bufs) if n_bufs > 1 else [bufs[0]]  # (a 30 GB set is not allocated twice)
        for set_ in pouts:
            for t in set_:
                t.zero_()
        ms, nl = precondition(pvec, pacts, pouts, precondition_ms, use_dist, dev)
        pre = {"preconditioning_ms": ms, "preconditioning_launches": nl,
               "preconditioning": "%d rollouts of %d steps of the same kernel on a scratch env and scratch trajectory buffers, each "
                                  "bracketed by synchronize like a timed repeat, untimed, before the warm-up steps" % (nl, K)}
    # warm-up: the same kernel in the same output mode, bracketed like a repeat (chunks of <= K steps into a set of their own)
    for w0 in range(0, W, K):
        bracketed_rollout(vec, acts[w0: min(W, w0 + K)], wbuf, use_dist)
    walls, evs = [], []
    for r in range(R):
        launch_clock(vec, clk[r])
        w, e = bracketed_rollout(vec, acts[W + r * K: W + (r + 1) * K], bufs[r % n_bufs], use_dist)
        walls.append(w); evs.append(e)
    gc.enable()
    launch_clock(vec, None)
    ch = clk.cpu().double()
    clocks = [float(ch[r, 0] / max(float(ch[r, 1]), 1.0)) * 0.1 for r in range(R)]
    if use_dist:
        t = torch.tensor(walls, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank defines the job's time, repeat by repeat
        walls = [float(v) for v in t.tolist()]
    del bufs, acts, wbuf
    torch.cuda.empty_cache()
    return walls, evs, n_bufs, clocks, pre


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of the K-step rollout; value = the median")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--precondition-ms", type=float, default=300.0,
                    help="untimed device conditioning before the warm-up steps: the same kernel on a scratch env and scratch buffers "
                         "for at least this many ms of wall time (reported as preconditioning_ms; 0 = none)")
    ap.add_argument("--burn-in-steps", type=int, default=1000,
                    help="untimed random-action steps of the timed env after its reset and before the device conditioning and the "
                         "warm-up steps, so that the timed steps run on a stationary mix of episode phases instead of a phase-locked "
                         "batch (reported as burn_in_steps; 0 = time the steps right after the reset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true", help="skip the informational one-launch-per-step timing")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informational C2 / C4 timings")
    ap.add_argument("--c5-full", action="store_true", help="also time BASELINE configs[4]'s per-rank share (131 072 envs, 10 beams) "
                    "on every rank and report other_configs.c5_full (automatic when --gpus 8)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1: print the N-rank child command line and its "
                    "communication environment as one JSON line and exit without starting anything")
    ap.add_argument("--workload", choices=("c3", "c4"), default="c3",
                    help="c3 (default, the BASELINE metric's config): 1 ship, 8 beams; c4: BASELINE configs[3], 4 ships "
                         "(traffic + dynamic goals + contact solver), 10 beams — informational, not the headline line")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.repeats < 1:
        ap.error("--gpus/--steps/--repeats must be >= 1 and --warmup >= 0")

    in_dist_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_dist_env:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from ship_sim_gym_amd.vec_env import ShipVecEnv
    from ship_sim_gym_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1")) if in_dist_env else 1
    rank = int(os.environ.get("RANK", "0")) if in_dist_env else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if in_dist_env else 0
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a line for the wrong N\n"
                         % (args.gpus, world))
        sys.exit(2)
    n_dev = torch.cuda.device_count()
    # TEST-ONLY overrides (tests/test_sharding_gpu.py on a 1-GPU box): all ranks on device 0 and gloo instead of RCCL, to
    # exercise the N > 1 control flow — barriers, MAX over ranks, the rank table, configs[4] — where no second GPU exists.
    # The numbers of such a run mean nothing and the JSON line says so.
    share_dev = os.environ.get("SSG_BENCH_SHARE_DEVICE") == "1"
    backend = os.environ.get("SSG_BENCH_BACKEND", "nccl")
    if share_dev:
        local_rank = 0
    if n_dev < (local_rank + 1) or (in_dist_env and not share_dev and n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))):
        sys.stderr.write("bench.py: rank %d/%d needs device %d but this node shows %d HIP device(s): one process per "
                         "GPU, no oversubscription\n" % (rank, world, local_rank, n_dev))
        sys.exit(3)
    use_dist = in_dist_env
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
            else:
                dist.init_process_group(backend=backend)
            world = dist.get_world_size()
            if world != args.gpus:
                raise RuntimeError("the process group has %d ranks, --gpus says %d" % (world, args.gpus))
            # one small collective before anything is timed: a rank that cannot reach its peers fails HERE, with its environment
            probe = torch.ones(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce over %d ranks returned %r" % (world, probe.item()))
        except Exception as ex:
            sys.stderr.write("bench.py: rank %d/%d (device %d of %d, backend %s) could not join the job: %r\n"
                             "bench.py: comm env: %s\n" % (rank, args.gpus, local_rank, n_dev, backend, ex, json.dumps(comm_env())))
            sys.exit(4)

    n = args.envs_per_gpu
    c4 = args.workload == "c4"
    n_beams = 10 if c4 else N_BEAMS
    vec = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=n_beams,
                     env_id_base=rank * n, exact_lidar=bool(int(os.environ.get("SSG_EXACT_LIDAR", "0"))),
                     bank_in_global=bool(int(os.environ.get("SSG_BENCH_BANK_IN_GLOBAL", "0"))),  # experiments only
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

    K, W, R = args.steps, args.warmup, args.repeats
    # the device-copy calibration runs BEFORE the timed region (it also brings the GPU out of its idle clocks; it is not a
    # step of the hot path and touches none of its buffers)
    copy_gbps = measured_copy_gbps(dev) if rank == 0 else None
    # scratch env for the untimed device conditioning: same kernel instantiation, same bank, its own state and buffers
    pvec = None
    if args.precondition_ms > 0:
        pvec = ShipVecEnv(n, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=n_beams, env_id_base=rank * n,
                          n_ships=4 if c4 else 1)
    walls, evs, n_bufs, clocks, pre = timed_rollouts(vec, K, W, R, use_dist, dev, args.precondition_ms, pvec, max(0, args.burn_in_steps))
    if pvec is not None:
        pvec.close()
        del pvec
    order = sorted(range(R), key=lambda i: walls[i])
    med = order[R // 2]
    wall, ev_ms = walls[med], evs[med]

    ranks_info = [{"rank": rank, "device": torch.cuda.get_device_name(dev), "hip_device": local_rank, "rccl_world_size": world,
                   "backend": (backend if use_dist else None), "envs": n, "env_id_base": rank * n, "env": comm_env()}]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_info[0])
        ranks_info = gathered

    # BASELINE configs[4] — 1 048 576 envs over 8 GPUs (131 072 per rank, 10 beams): timed by every rank when the job is that
    # size (or on request), reported by rank 0 next to the headline
    c5_full = None
    if (world == 8 or args.c5_full) and not c4:
        vec5 = ShipVecEnv(131072, device=dev, map_mode="bank", n_maps=N_MAPS, map_seed=1000, n_beams=10, env_id_base=rank * 131072)
        if use_dist:
            sharding.broadcast_bank(vec5, src=0)
        k5 = min(K, 500)
        w5, e5 = timed_rollouts(vec5, k5, min(W, 100), 3, use_dist, dev)[:2]
        m5 = sorted(w5)[1]
        B5 = algorithmic_bytes(1, 10, 2)
        c5_full = {"workload": "BASELINE configs[4]: %d envs = 131072 per rank x %d ranks, 1 ship, 10-beam lidar, trajectory outputs"
                               % (131072 * world, world), "total_envs": 131072 * world, "n_gpus": world, "steps": k5,
                   "env_steps_per_s": 131072.0 * world * k5 / m5, "ms_per_step": m5 * 1e3 / k5,
                   "algorithmic_bytes_per_env_step": B5, "achieved_GBps_per_gpu": 131072.0 * k5 / m5 * B5 / 1e9,
                   "frac_per_gpu": 131072.0 * k5 / m5 * B5 / 1e9 / HBM_PEAK_GBPS}
        vec5.close()
        del vec5
        torch.cuda.empty_cache()

    if rank == 0:
        total_steps = float(n) * world * K
        B = algorithmic_bytes(4 if c4 else 1, n_beams, 2)
        # ssg_rollout_traj fuses SSG_ROLLOUT_STEPS_PER_LAUNCH (100) steps into each launch of the step kernel:
        # algorithmic bytes per launch = B * n * steps_per_launch, launch duration = HIP-event time / launches
        # (config 4 has no fused rollout: one "launch" below is one step = dyn kernels + step kernel)
        spl = 1 if c4 else steps_per_launch_cfg()
        n_launch = (K + spl - 1) // spl
        launch_s = ev_ms * 1e-3 / n_launch
        steps_in_launch = K / n_launch
        achieved = B * n * steps_in_launch / launch_s / 1e9
        env_steps_per_s_kernel = n * steps_in_launch / launch_s
        # PMC-derived figures: only from a stored profile of THIS kernel source, in THIS output mode
        roof_pmc = {"traffic": None, "traffic_bytes_per_env_step": None, "hbm_measured": None, "valu_issue_frac": None,
                    "lds_pipe_busy_frac": None, "shader_clock_ghz_profiled": None, "valu_issue_frac_at_profiled_clock": None,
                    "counters_source": None,
                    "counters_note": "PMC counters cannot be read in-process; no stored rocprofv3 profile of this kernel source "
                                     "(sha %s) in trajectory mode under profiles/" % source_sha()}
        bound = "unknown (no PMC profile of this build)"
        tpath = os.path.join(ROOT, "profiles", "counters_latest.json")
        if os.path.exists(tpath) and not c4:
            try:
                tj = json.load(open(tpath))
                if tj.get("source_sha") == source_sha() and tj.get("mode") == "trajectory":
                    bpe = float(tj["hbm_bytes_per_env_step"])
                    hbm_gbps = bpe * env_steps_per_s_kernel / 1e9
                    valu = float(tj["valu_wave_insts_per_env_step"]) * env_steps_per_s_kernel / VALU_ISSUE_PEAK
                    # LDS pipe: busy cycles per env-step (summed over the CUs) x this run's rate / (256 CUs x 2.4 GHz)
                    lds = (float(tj["lds_active_cycles_per_env_step"]) * env_steps_per_s_kernel / (256 * 2.4e9)
                           if "lds_active_cycles_per_env_step" in tj else None)
                    roof_pmc = {"traffic": bpe * n * steps_in_launch, "traffic_bytes_per_env_step": bpe,
                                "hbm_measured": {"GBps": hbm_gbps, "frac": hbm_gbps / HBM_PEAK_GBPS,
                                                 "note": "stored-profile HBM bytes per env-step x this run's kernel rate"},
                                "valu_issue_frac": valu, "lds_pipe_busy_frac": lds,
                                # the shader clock the profiled launches actually ran at (SQ_WAVE_CYCLES / waves / duration:
                                # FP64-heavy kernels do not hold the 2.4 GHz boost ceiling the peaks above are priced at), and
                                # the VALU issue fraction against 1024 SIMDs x THAT clock / 4
                                "shader_clock_ghz_profiled": tj.get("shader_clock_ghz"),
                                "valu_issue_frac_at_profiled_clock": (valu * 2.4 / float(tj["shader_clock_ghz"])
                                                                      if tj.get("shader_clock_ghz") else None),
                                "counters_source": tj.get("source"),
                                "counters_note": "derived from the stored rocprofv3 PMC profile of this kernel source (sha %s), "
                                                 "not measured by this run" % source_sha()}
                    cands = {"hbm": hbm_gbps / HBM_PEAK_GBPS, "valu_issue": valu, "lds_pipe": float(lds or 0.0)}
                    bound = max(cands, key=cands.get)
            except Exception:
                pass
        out = {
            "metric": "env steps/sec (batched ShipEnv)", "value": total_steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not share_dev else "synthetic (TEST RUN: ranks share one device over %s; timings meaningless)" % backend,
            "config": {"workload": ("BASELINE configs[3]: 65536 parallel envs per GPU x 4 ships (3 traffic ships, dynamic "
                                    "goal bodies, Chipmunk contact solver), 10-beam lidar, 64-map bank, random Philox "
                                    "actions, auto-reset in-kernel") if c4 else
                                   ("BASELINE configs[2]: 65536 parallel envs per GPU, 1 ship, 8-beam lidar, 64-map "
                                    "bank (600x600, SPEED 10), random Philox actions, auto-reset in-kernel"),
                       "outputs": "trajectory: every step's obs/reward/done/flags kept in HBM as [K, N, ...] tensors "
                                  "(ssg_rollout_traj), %d B per env-step, %d buffer set(s) of %.2f GB rotated over the repeats"
                                  % (traj_bytes_per_step(1, vec.states_history), n_bufs, traj_bytes_per_step(n, vec.states_history) * K / 1e9),
                       "envs_per_gpu": n, "total_envs": n * world, "n_beams": n_beams, "history": 2,
                       "parallelism": "env-sharded x%d, one process per GPU (RCCL world size %d), no data-path "
                                      "collective" % (world, world),
                       "ranks": ranks_info},
            "repeats": R, "repeats_ms": [w * 1e3 for w in walls], "timing": "median of %d repeats of the K-step rollout" % R,
            # spread of the timed repeats (value = the median one), the shader clock DURING each (recorded by the step kernel itself:
            # s_memtime over the 100 MHz reference counter, rank 0) and the untimed device conditioning that preceded the warm-up
            "repeats_min_ms": min(walls) * 1e3, "repeats_median_ms": wall * 1e3, "repeats_max_ms": max(walls) * 1e3,
            "repeats_spread": (max(walls) - min(walls)) / wall,
            "value_min": total_steps / max(walls), "value_max": total_steps / min(walls),
            "repeats_event_ms": evs, "repeats_shader_clock_ghz": clocks,
            "preconditioning_ms": pre["preconditioning_ms"], "preconditioning": pre.get("preconditioning"),
            "burn_in_steps": max(0, args.burn_in_steps),
            "burn_in": "untimed random-action steps of the timed env between its reset and the warm-up steps: the timed steps run on a "
                       "stationary mix of episode phases, not on a batch whose envs all started their first episode together",
            "roofline": dict({"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBPS,
                              "frac_note": "SURVEY 8d's algorithmic bytes per env-step (single-step API with persistent state) over "
                                           "the live launch time (HIP events); it charges every step a state read + write and a "
                                           "re-read of the previous frame, which a fused launch keeps in registers, so it can "
                                           "exceed what the kernel moves: frac_wall is the same formula over the wall time "
                                           "`value` uses, frac_fused_compulsory the bytes the fused API must move",
                              # the same formula over the driver-visible wall time (value = envs x steps / wall)
                              "frac_wall": B * (total_steps / world / wall) / 1e9 / HBM_PEAK_GBPS,
                              # what the fused API has to move: outputs + action every step, the state once per launch
                              "fused_compulsory_bytes_per_env_step": (None if c4 else fused_compulsory_bytes(n_beams, vec.states_history, steps_in_launch)),
                              "frac_fused_compulsory": (None if c4 else fused_compulsory_bytes(n_beams, vec.states_history, steps_in_launch)
                                                        * env_steps_per_s_kernel / 1e9 / HBM_PEAK_GBPS),
                              "measured_copy_GBps": copy_gbps,
                              "kernel": ("ssg::dyn_* kernels + ssg::step_kernel<10, 256, true, false, true>"
                                         if c4 else "ssg::step_kernel<8, 256, true, false, false>"),
                              "algorithmic_bytes_per_env_step": B,
                              "output_bytes_per_env_step": traj_bytes_per_step(1, vec.states_history),
                              "steps_per_launch": steps_in_launch, "avg_launch_us": launch_s * 1e6,
                              "us_per_step_in_launch": launch_s * 1e6 / steps_in_launch}, **roof_pmc),
        }
        other = {}
        if world == 1 and not args.no_single_step:
            # the policy-in-the-loop path: one ssg_step launch per step; median of 7 repeats of `ks` back-to-back steps
            # (500 whatever --steps says: every repeat starts from an idle, synchronized GPU and pays ~80 us for it once —
            # 0.8 us per step over 100 steps, 0.16 over 500; a trainer's stream of steps has no such restarts)
            ks = 500
            a1 = vec.random_actions(777, 0, ks)
            rows = [a1[k] for k in range(ks)]  # (the row views are made outside the timed loop: it times launches, not slicing)
            step = vec.step_tensor
            reps = []
            for _ in range(7):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for a in rows:
                    step(a)
                e1.record()
                torch.cuda.synchronize()
                reps.append(e0.elapsed_time(e1) * 1e3 / ks)
            out["single_step_launch_us"] = sorted(reps)[len(reps) // 2]
            out["single_step_launch_us_repeats"] = reps
        else:
            out["single_step_launch_us"] = None
        if world == 1 and not args.no_other_configs and not c4:
            try:
                other["c2_4096_envs_10_beams"] = side_config(dev, 4096, 10, 1, 1000, 200, single_step=True)
                # BASELINE configs[3].  Bank mode (the benchmark workload of SURVEY 8d: envs share the 64 records): cpSpaceStep of the
                # traffic ships / goal bodies is MEMOISED (SSG_F_DYN_MEMO_STATS) — labelled, and next to it the same run with every
                # step computed, and with a brand-new world per episode (no env shares a world: nothing to memoise)
                other["c4_65536_envs_x4_ships_10_beams"] = side_config(dev, 65536, 10, 4, 200, 200, kernel_split=True)
                other["c4_65536_envs_x4_ships_10_beams"]["note"] = ("bank mode: envs on one bank record replay the same body states, "
                                                                    "so a state's cpSpaceStep is computed once and looked up afterwards "
                                                                    "(verified word for word); c4_memo_off / c4_fresh_world_per_episode "
                                                                    "are the figures without that sharing")
                other["c4_memo_off"] = side_config(dev, 65536, 10, 4, 200, 200, dyn_memo=False, kernel_split=True)
                # (rings of 128 worlds per env, 9.7 GB of the 288, as for c3_fresh_world_per_episode: one refill per 127 steps)
                other["c4_fresh_world_per_episode"] = side_config(dev, 65536, 10, 4, 254, 127, map_mode="fresh_device", ring=128)
                other["c4_fresh_world_per_episode"]["ring"] = 128
                other["c4_single_step_auto_reset"] = c4_policy_in_the_loop(dev, 65536, host_reset=False)
                other["c4_host_masked_reset"] = c4_policy_in_the_loop(dev, 65536, host_reset=True)
                # the RLlib flow as ShipVecEnv(rllib=True) runs it since round 6: no reset launch (ssg_set_terminal_obs)
                other["c4_rllib_flow_terminal_obs"] = c4_policy_in_the_loop(dev, 65536, host_reset=False, terminal_obs=True)
                # the numpy protocol stable-baselines / RLlib callers use, next to its PCIe bound
                other["vecenv_numpy_path"] = {"65536_envs_8_beams": vecenv_numpy_path(dev, 65536, 8),
                                              "4096_envs_10_beams": vecenv_numpy_path(dev, 4096, 10)}
                # an end-to-end PPO loop over the zero-copy API: eager / HIP graph / two-half-batch ping-pong
                other["ppo_torch_end_to_end"] = ppo_end_to_end(dev, n, out.get("single_step_launch_us"))
                # (the same loop at BASELINE configs[1]'s size, where a rollout step is launch-bound and the graph pays)
                other["ppo_torch_end_to_end_4096_envs"] = ppo_end_to_end(dev, 4096, None, updates=6, horizon=64)
                other["c5_share_131072_envs_10_beams"] = side_config(dev, 131072, 10, 1, 500, 100)
                # the headline's envs on a bank of 120 records, which the LDS only holds beside 64-env workgroups (four rounds per
                # launch): gathered from L2 on 256-env workgroups instead (ssg_set_map_bank weighs the two)
                other["c3_bank_of_120_records"] = side_config(dev, 65536, 8, 1, 500, 100, n_maps=120)
                # a brand-new world per episode, drawn on the device (map_mode="fresh_device", ring of 128 worlds per env: up to 127
                # steps between refills; 9.7 GB of the 288).  The world generator is throughput-bound — ~1 630 new worlds per step at
                # ~780 wave-instructions each — so its share (~2.2 us per step) does not depend on the ring
                other["c3_fresh_world_per_episode"] = side_config(dev, 65536, 8, 1, 635, 127, map_mode="fresh_device", ring=128)
                other["c3_fresh_world_per_episode"]["ring"] = 128
                # the headline workload with every step overwriting the same [N, ...] rows (ssg_rollout): outputs stay in cache
                vec.reset_tensor()
                ao = vec.random_actions(999, 0, 600)
                vec.rollout_tensor(ao[:100])
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record(); vec.rollout_tensor(ao[100:]); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ms = sorted(ts)[1]
                other["c3_overwrite_outputs"] = {"envs": n, "steps": 500, "us_per_step": ms * 1e3 / 500, "env_steps_per_s": n * 500 / (ms * 1e-3),
                                                 "outputs": "every step rewrites the same [N, ...] rows (ssg_rollout)"}
            except Exception as ex:  # informational only: never lose the headline line
                other["error"] = repr(ex)
        if c5_full is not None:
            other["c5_full"] = c5_full
        out["other_configs"] = other or None
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
