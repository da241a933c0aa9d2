"""Multi-GPU: env batches sharded across ranks, one process per GPU, RCCL over xGMI only for the map bank.

Envs are fully independent (each reference ShipEnv owns a private ShipGame, ship_env.py:32; the reference's only
parallelism is one OS process per env, train/stable_baselines/ppo.py:122-123), so rank r owns the contiguous
global env range [r*N/W, (r+1)*N/W) and steps it with no per-step communication.  The only exchange is a
broadcast of the map bank from rank 0 at start-up and at each curriculum lesson change (a few hundred KB:
latency-bound), plus an optional all-reduce of the four episode counters for logging / Curriculum.progress.
The Philox action stream and the default map assignment are keyed by GLOBAL env id, so a sharded run is bitwise
the union of its shards (tests/test_parity_gpu.py::test_shard_equivalence, tests/test_sharding_cpu.py).
"""


def shard_range(total_envs, rank, world_size):
    """Contiguous [lo, hi) of global env ids owned by `rank`; remainders go to the lowest ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _through_host(t, group):
    """gloo carries CPU tensors: a device tensor is staged through host memory (two ranks sharing ONE GPU — the
    1-GPU test box — cannot form an RCCL communicator; on a real node the backend is nccl = RCCL over xGMI)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def broadcast_tensor(t, src=0, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _through_host(t, group):
            h = t.cpu()
            dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=group)
    return t


def broadcast_bank(vec, src=0, group=None):
    """Make every rank's bank identical to rank `src`'s (RCCL broadcast on device tensors; gloo on CPU tensors)."""
    import torch.distributed as dist
    broadcast_tensor(vec.bank, src=src, group=group)
    if dist.is_available() and dist.is_initialized() and dist.get_rank(group) != src:
        vec.bank_polys = vec.bank_goals = None  # host copies described the overwritten bank
        vec.set_bank(vec.bank)                  # same tensor, new contents: re-collide resting bodies (config 4)
    return vec.bank


def all_reduce_stats(stats_tensor, group=None):
    """Sum {sum_return, sum_length, episodes, goals_hit} over ranks (feeds Curriculum.progress, curriculum.py:40-50)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _through_host(stats_tensor, group):
            h = stats_tensor.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            stats_tensor.copy_(h)
        else:
            dist.all_reduce(stats_tensor, op=dist.ReduceOp.SUM, group=group)
    return stats_tensor


def global_stats(vec, group=None):
    """Episode counters of the WHOLE sharded job (every rank gets the same dict): the per-handle int64 counters the
    step kernel accumulates, summed on the device and all-reduced over ranks."""
    s = vec.field_stats_tensor()
    all_reduce_stats(s, group=group)
    v = s.cpu().numpy()
    return {"sum_return": float(v[0]) / 100.0, "sum_length": int(v[1]), "episodes": int(v[2]), "goals_hit": int(v[3])}


def make_sharded_env(total_envs, rank=None, world_size=None, device=None, **kw):
    """ShipVecEnv over this rank's shard of `total_envs` global envs; the bank is broadcast from rank 0."""
    import os
    import torch
    from .vec_env import ShipVecEnv
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    lo, hi = shard_range(total_envs, rank, world_size)
    vec = ShipVecEnv(hi - lo, device=device, env_id_base=lo, **kw)
    broadcast_bank(vec, src=0)
    return vec
