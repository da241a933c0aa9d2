#!/usr/bin/env python3
"""Development aid (GPU box): does the gathered step kernel's extra time come from WHERE its records lie?  The same 65 536 envs
on (a) a shared 64-record bank gathered from L2, (b) one record per env, records DENSE (bank mode on 65 536 / 262 144 device-drawn
records: env e walks e, e+1, ...), (c) per-env rings of R records (fresh_device: record e*R + p, a 9.7 GB footprint at R = 128).
Run under rocprofv3 --kernel-trace for the step kernel's own time (tools/gather_locality.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
n, K, W = 65536, int(os.environ.get("K", "600")), 100
which = os.environ.get("WHICH", "all")


def timed(vec, label):
    acts = vec.random_actions(12345, 0, K + W)
    vec.reset_tensor(); vec.rollout_tensor(acts[:W]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); vec.rollout_tensor(acts[W:]); e1.record(); torch.cuda.synchronize()
    print("%-60s %.2f us per step" % (label, e0.elapsed_time(e1) * 1e3 / K), flush=True)


if which in ("all", "shared"):
    v = ShipVecEnv(n, n_beams=8, n_maps=64, bank_in_global=True); timed(v, "shared 64-record bank, gathered from L2"); v.close(); del v
if which in ("all", "ring"):
    v = ShipVecEnv(n, n_beams=8, map_mode="fresh_device", ring=128); timed(v, "per-env rings of 128 (fresh_device, refills included)")
    bank = v.bank.view(n, 128, -1)
    dense1 = bank[:, 0, :].contiguous(); dense4 = bank[:, :4, :].reshape(n * 4, -1).contiguous()
    sparse = v.bank
    v.close(); del v
    for lab, b in (("bank mode, 65 536 dense records (76 MB)", dense1), ("bank mode, 262 144 dense records (304 MB)", dense4)):
        v = ShipVecEnv(n, n_beams=8, map_mode="bank", bank=b); timed(v, lab); v.close(); del v
