#!/usr/bin/env python3
"""Diagnostic: the policy-in-the-loop path (one ssg_step launch per step).  Prints the host-side time per call and the
HIP-event time per call; run under `rocprofv3 --kernel-trace --stats` to get the kernel's own duration beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
n = int(os.environ.get("SSG_N", "65536")); nb = int(os.environ.get("SSG_NB", "8"))
vec = ShipVecEnv(n, n_maps=64, n_beams=nb, bank_in_global=bool(int(os.environ.get("SSG_BIG", "0"))))  # SSG_BIG=1: bank gathered from L2, no LDS staging
if os.environ.get("SSG_ABLATE"):  # timing-only (needs a -DSSG_ABLATION build)
    import ctypes as C
    from ship_sim_gym_amd import _native as N
    vec.cfg.flags |= int(os.environ["SSG_ABLATE"], 0) << 16
    N.lib().ssg_destroy(vec._h)
    N.check(N.lib().ssg_create(C.byref(vec.cfg), C.byref(vec._h)), None, "ssg_create")
    N.check(N.lib().ssg_bind_state(vec._h, C.c_void_p(vec.state.data_ptr())), vec._h, "bind")
    vec.set_bank(vec.bank)
acts = vec.random_actions(12345, 0, 1200)
vec.reset_tensor()
for k in range(200): vec.step_tensor(acts[k])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for k in range(200, 1200): vec.step_tensor(acts[k])
t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.2f us/call | events %.2f us/step | wall incl. sync %.2f us/step" % ((t1 - t0) * 1e3, e0.elapsed_time(e1), (t2 - t0) * 1e3))
