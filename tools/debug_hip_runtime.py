"""Debug helper: which HIP runtime(s) end up in the process, and does hipGetDevice work from libshipsim?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch cuda available", torch.cuda.is_available())
x = torch.zeros(4, device="cuda:0")
def maps():
    s = set()
    for line in open("/proc/self/maps"):
        if "amdhip" in line or "hsa-runtime" in line:
            s.add(line.split()[-1])
    return sorted(s)
print("after torch:", maps())
from ship_sim_gym_amd import _native as N
L = N.lib()
print("after libshipsim:", maps())
hip = C.CDLL("libamdhip64.so.7")
d = C.c_int(-1)
print("hipGetDevice via soname:", hip.hipGetDevice(C.byref(d)), d.value)
from ship_sim_gym_amd.vec_env import ShipVecEnv
v = ShipVecEnv(256, n_maps=4)
try:
    v.reset_tensor(); torch.cuda.synchronize(); print("reset ok")
except Exception as e:
    print("reset failed:", e)
a = v.random_actions(1, 0, 2); torch.cuda.synchronize(); print("fill ok")
try:
    v.reset_tensor(); torch.cuda.synchronize(); print("reset ok (2nd)")
except Exception as e:
    print("reset failed (2nd):", e)
