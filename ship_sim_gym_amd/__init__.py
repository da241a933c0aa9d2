"""ship_sim_gym_amd — MI355X-native batched replacement for ship-sim-gym's ShipEnv.step()/reset() hot path.

Python here is the thin host: configuration, reset-time world generation in the reference's RNG order, and the
gym / VecEnv / VectorEnv facades.  All stepping runs in hand-written gfx950 HIP kernels behind the C ABI of
``libshipsim.so`` (include/shipsim.h), reached through ctypes; PyTorch-ROCm tensors are only the device buffers.
"""
from .config import EnvConfig, GameConfig, LidarConfig  # noqa: F401
from .curriculum import Curriculum  # noqa: F401

__all__ = ["EnvConfig", "GameConfig", "LidarConfig", "Curriculum", "ShipEnv", "ShipVecEnv"]


def __getattr__(name):  # lazy: importing the package must not require torch or the built library
    if name == "ShipVecEnv":
        from .vec_env import ShipVecEnv
        return ShipVecEnv
    if name == "ShipEnv":
        from .ship_env import ShipEnv
        return ShipEnv
    raise AttributeError(name)
