import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ship_sim_gym_amd.vec_env import ShipVecEnv
for n in (65536, 131072, 4096):
    vec = ShipVecEnv(n, n_beams=10, n_maps=64)
    acts = vec.random_actions(1, 0, 1100)
    vec.reset_tensor(); vec.rollout_tensor(acts[:100]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); vec.rollout_tensor(acts[100:]); e1.record(); torch.cuda.synchronize()
    print(n, "x10 beams: %.2f us/step" % (e0.elapsed_time(e1) * 1e3 / 1000))
    vec.close()
