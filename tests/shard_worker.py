"""Worker of tests/test_sharding_gpu.py: ONE rank of a sharded job on the PRODUCT path.

    python tests/shard_worker.py <rank> <world> <port> <total_envs> <K> <out_dir> <n_ships>

Runs sharding.make_sharded_env (ShipVecEnv over this rank's env range on the HIP path; map bank broadcast from
rank 0), K single steps + a fused rollout on the global Philox action stream, then all-reduces the episode
counters.  Backend: nccl (= RCCL) when the node shows at least `world` devices, else gloo with both ranks on
cuda:0 and the bank staged through host memory (sharding._through_host) — the 1-GPU test box.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, total, K, out_dir, n_ships = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                                                     int(sys.argv[5]), sys.argv[6], int(sys.argv[7]))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from ship_sim_gym_amd import sharding
    n_dev = torch.cuda.device_count()
    use_rccl = n_dev >= world
    dev = torch.device("cuda", rank if use_rccl else 0)
    torch.cuda.set_device(dev)
    if use_rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank but 0 starts from a DIFFERENT bank (other seed): only the broadcast can make the shards agree
    vec = sharding.make_sharded_env(total, rank=rank, world_size=world, device=dev, n_maps=16,
                                    map_seed=1000 if rank == 0 else 555 + rank, n_ships=n_ships)
    lo, hi = sharding.shard_range(total, rank, world)
    assert vec.env_id_base == lo and vec.num_envs == hi - lo
    obs0 = vec.reset_tensor().cpu().numpy().copy()
    acts = vec.random_actions(4242, 0, 2 * K)
    rews, dones = [], []
    for k in range(K):
        o, r, d, f = vec.step_tensor(acts[k])
        rews.append(r.cpu().numpy().copy())
        dones.append(d.cpu().numpy().copy())
    o, r, d, f = vec.rollout_tensor(acts[K:])  # fused launches on the same handle
    torch.cuda.synchronize()
    local = vec.stats()
    glob = sharding.global_stats(vec)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, obs0=obs0, obs=o.cpu().numpy(), rew=r.cpu().numpy(),
             done=d.cpu().numpy(), flags=f.cpu().numpy(), rews=np.stack(rews), dones=np.stack(dones),
             bank=vec.bank.cpu().numpy(), x=vec.field(0).cpu().numpy(),
             local=np.array([local["sum_return"], local["sum_length"], local["episodes"], local["goals_hit"]]),
             glob=np.array([glob["sum_return"], glob["sum_length"], glob["episodes"], glob["goals_hit"]]),
             backend=np.array([1 if use_rccl else 0]), n_dev=np.array([n_dev]),
             seen_world=np.array([dist.get_world_size()]), seen_backend=np.array(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()
    vec.close()


if __name__ == "__main__":
    main()
