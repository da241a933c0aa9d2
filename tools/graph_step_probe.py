#!/usr/bin/env python3
"""Development aid (GPU box): can a policy-in-the-loop caller capture ssg_step in a HIP graph (torch.cuda.graph) and replay it?
One step per replay, actions read from a static buffer; outputs compared with un-captured launches; time per replayed step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
for n, ships in ((4096, 1), (65536, 1), (4096, 4)):
    a = ShipVecEnv(n, n_beams=10, n_ships=ships); b = ShipVecEnv(n, n_beams=10, n_ships=ships)
    a.reset_tensor(); b.reset_tensor()
    K = 300
    acts = a.random_actions(5, 0, K)
    static_act = torch.zeros(n, dtype=torch.int32, device="cuda")
    static_act.copy_(acts[0]); a.step_tensor(static_act); b.step_tensor(acts[0])  # warm-up outside the capture (prepares the kernels)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        static_act.copy_(acts[1])
        with torch.cuda.graph(g, stream=s):
            a.step_tensor(static_act)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    # the capture itself does not run the step: replay it for step 1, then steps 2..K-1
    ok = True
    for k in range(1, K):
        static_act.copy_(acts[k]); g.replay()
        ob, rb, db, fb = b.step_tensor(acts[k])
        if k % 37 == 0 or k == K - 1:
            ok &= bool(torch.equal(a.obs, ob) and torch.equal(a.reward, rb) and torch.equal(a.done, db))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(500): g.replay()
    e1.record(); torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) * 1e3 / 500
    e0.record()
    for k in range(500): b.step_tensor(static_act)
    e1.record(); torch.cuda.synchronize()
    print("%d envs x %d ship(s): graph replay == plain launches: %s; %.2f us per replayed step, %.2f us per plain step_tensor call" % (
        n, ships, ok, t_graph, e0.elapsed_time(e1) * 1e3 / 500), flush=True)
    a.close(); b.close()
