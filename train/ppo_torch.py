#!/usr/bin/env python3
"""Trainer glue demo: a minimal PPO loop in plain PyTorch driving ShipVecEnv with zero-copy
device tensors — the role SubprocVecEnv + PPO2 play in the reference's train/stable_baselines/ppo.py:84-123, without
stable-baselines (absent from this image).  The env side is the only point: observations, rewards and dones never
leave the GPU; the policy is a small MLP in fp32.

    python train/ppo_torch.py --envs 4096 --updates 20 [--mode eager|graph|pingpong]

Three ways to run the rollout loop (the reference's `model.learn` -> runner.run(): one `env.step(actions)` per policy
forward, train/stable_baselines/ppo.py:84-100,122-123) — same arithmetic, same results bit for bit:

* ``eager``    — every rollout step launches the policy forward, the action sampling, ``ssg_step`` and the buffer writes one
                 kernel at a time from Python (a dozen launches and their host overhead per step);
* ``graph``    — that whole step — policy forward + sampling + ``ssg_step`` + buffer writes — is captured ONCE as a HIP graph
                 (a 1-ship handle on a shared bank launches with constant arguments, include/shipsim.h) and replayed per step:
                 one host call per rollout step;
* ``pingpong`` — the batch is split into two halves (two ShipVecEnv shards, global env ids and so results unchanged), each
                 with its own graph on its own stream: half A's env step runs while half B's policy forward does.

The sampling noise of a whole rollout is drawn in one call before it (uniforms [horizon, envs], inverse-CDF sampling inside
the step), so a captured step holds no random-number generator state and replays exactly what the eager loop computes.

This is NOT a re-implementation of the reference's trainers (out of scope); it exists to show the batched env plugs
into a GPU-resident training loop and to give `tests/` and `bench.py` an end-to-end caller.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ship_gym.config import EnvConfig, GameConfig  # noqa: E402  (the reference's import lines, via the alias package)
from ship_sim_gym_amd.vec_env import ShipVecEnv  # noqa: E402


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, n_actions, hidden=64):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.pi = nn.Linear(hidden, n_actions)
        self.v = nn.Linear(hidden, 1)

    def forward(self, x):
        h = self.body(x)
        return self.pi(h), self.v(h).squeeze(-1)


def normalise(obs, scale):
    """obs is float64 with -1 for "nothing yet"; positions/lidar in map units, rudder in degrees, angle in rad."""
    return (obs / scale).float()


class Shard(object):
    """One env shard of the rollout and everything a rollout step of it reads or writes, at fixed addresses (what a captured
    step needs): the env's own output tensors, the rollout buffers [horizon, n, ...] and a device-side step index."""

    def __init__(self, env, net, scale, horizon):
        self.env, self.net, self.scale, self.horizon = env, net, scale, horizon
        n, D, dev = env.num_envs, env.states_history, env.device
        self.n = n
        f32 = dict(dtype=torch.float32, device=dev)
        self.buf_obs = torch.zeros((horizon, n, D), **f32)
        self.buf_act = torch.zeros((horizon, n), dtype=torch.int64, device=dev)
        self.buf_logp, self.buf_val = torch.zeros((horizon, n), **f32), torch.zeros((horizon, n), **f32)
        self.buf_rew, self.buf_done = torch.zeros((horizon, n), **f32), torch.zeros((horizon, n), **f32)
        self.noise = torch.zeros((horizon, n), **f32)           # uniforms of the whole rollout, drawn before it
        self.t = torch.zeros(1, dtype=torch.int64, device=dev)  # the step index, advanced by the step itself
        self.act_i32 = torch.zeros(n, dtype=torch.int32, device=dev)
        self.graph = None
        self.stream = None

    def step(self):
        """ONE rollout step of this shard: policy forward on the env's current observation, inverse-CDF sampling with this
        step's uniforms, ssg_step (the env rewrites its obs / reward / done in place; done envs are reset in-kernel), and the
        rollout buffers' rows of this step.  Every address it touches is fixed, so it can run eagerly or be captured."""
        env, t = self.env, self.t
        x = normalise(env.obs, self.scale)
        with torch.no_grad():
            logits, val = self.net(x)
            logp_all = torch.log_softmax(logits, dim=-1)
            cdf = logp_all.exp().cumsum(dim=-1)
            u = self.noise.index_select(0, t)[0]
            act = (u.unsqueeze(-1) > cdf[:, :-1]).sum(dim=-1)                 # in 0 .. n_actions - 1
            logp = logp_all.gather(-1, act.unsqueeze(-1)).squeeze(-1)
        self.act_i32.copy_(act)
        self.buf_obs.index_copy_(0, t, x.unsqueeze(0))
        self.buf_act.index_copy_(0, t, act.unsqueeze(0))
        self.buf_logp.index_copy_(0, t, logp.unsqueeze(0))
        self.buf_val.index_copy_(0, t, val.unsqueeze(0))
        _, rew, done, _ = env.step_tensor(self.act_i32)
        self.buf_rew.index_copy_(0, t, rew.float().unsqueeze(0))
        self.buf_done.index_copy_(0, t, done.float().unsqueeze(0))
        t.add_(1)

    def capture(self, stream):
        """Capture step() as a HIP graph on `stream` (after one eager step on a scratch copy of nothing: the library's kernels are
        prepared by the env's reset + the warm-up step the caller ran).  The capture itself executes nothing."""
        self.stream = stream
        g = torch.cuda.CUDAGraph()
        stream.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                self.step()
        torch.cuda.current_stream(self.env.device).wait_stream(stream)
        self.graph = g


def env_columns(env):
    """The env's body / episode state columns (clones): what two runs that stepped the same envs the same way must agree on.  (Not the
    whole state blob: its reset counters also count the extra reset a graph run does after its warm-up step.)"""
    from ship_sim_gym_amd import _native as N
    return {name: env.field(getattr(N, name)).clone() for name in
            ("F_X", "F_Y", "F_VX", "F_VY", "F_ANGLE", "F_W", "F_LIDAR", "F_RUDDER", "F_STEP_COUNT", "F_MAP_ID", "F_GOAL_MASK", "F_CUM_REWARD")}


def make_shards(envs, mode, net, device, horizon, n_maps=64, env_kw=None):
    """The rollout's env shards: one ShipVecEnv, or (pingpong) two halves that together are the same batch — global env ids,
    map assignment and therefore every result are those of the unsplit batch (ship_sim_gym_amd/sharding.py)."""
    env_kw = dict(env_kw or {})
    sizes = [envs] if mode != "pingpong" else [envs - envs // 2, envs // 2]
    shards, base = [], 0
    for n in sizes:
        env = ShipVecEnv(n, GameConfig, EnvConfig, device=device, n_maps=n_maps, env_id_base=base, **env_kw)  # was: SubprocVecEnv([make_env()]*n)
        scale = torch.full((env.states_history,), float(max(env.bounds)), dtype=torch.float64, device=device)
        shards.append(Shard(env, net, scale, horizon))
        base += n
    return shards


def rollout(shards, horizon, mode, gen):
    """`horizon` policy-in-the-loop steps of every shard; returns the rollout buffers concatenated over the shards (env axis)."""
    full = torch.rand((horizon, sum(sh.n for sh in shards)), generator=gen, device=shards[0].noise.device)
    base = 0
    for sh in shards:  # (one draw for the whole batch, split by env: a split batch samples with the unsplit batch's uniforms)
        sh.noise.copy_(full[:, base: base + sh.n])
        sh.t.zero_()
        base += sh.n
    if mode == "eager":
        for _ in range(horizon):
            shards[0].step()
    elif mode == "graph":
        g = shards[0].graph
        for _ in range(horizon):
            g.replay()
    else:  # pingpong: both graphs every step, each on its own stream — A's env step overlaps B's policy forward
        cur = torch.cuda.current_stream(shards[0].env.device)
        for sh in shards:
            sh.stream.wait_stream(cur)
        for _ in range(horizon):
            for sh in shards:
                with torch.cuda.stream(sh.stream):
                    sh.graph.replay()
        for sh in shards:
            cur.wait_stream(sh.stream)
    cat = (lambda name: torch.cat([getattr(sh, name) for sh in shards], dim=1)) if len(shards) > 1 else (lambda name: getattr(shards[0], name))
    return {k: cat("buf_" + k) for k in ("obs", "act", "logp", "val", "rew", "done")}


def train(envs=4096, updates=20, horizon=64, epochs=2, minibatches=4, lr=3e-4, gamma=0.99, lam=0.95, clip=0.2,
          device="cuda:0", seed=0, log=print, mode="eager", return_details=False, env_kw=None):
    assert mode in ("eager", "graph", "pingpong")
    torch.manual_seed(seed)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 1)
    probe = ShipVecEnv(1, GameConfig, EnvConfig, device=device, n_maps=1)
    D, A = probe.states_history, probe.action_space.n
    probe.close()
    net = ActorCritic(D, A).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    shards = make_shards(envs, mode, net, device, horizon, env_kw=env_kw)
    for sh in shards:
        sh.env.reset_tensor()
    if mode != "eager":
        # one eager warm-up step per shard OUTSIDE the capture (prepares the library's kernels and hipBLASLt's workspaces), then the
        # envs start over; the capture itself runs nothing
        for sh in shards:
            sh.step()
            sh.env.reset_tensor()
            sh.t.zero_()
        torch.cuda.synchronize()
        for sh in shards:
            sh.capture(torch.cuda.Stream(device=dev))
        torch.cuda.synchronize()
    history, t_roll, t_all0 = [], 0.0, time.perf_counter()
    snapshots = []
    for u in range(updates):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = rollout(shards, horizon, mode, gen)
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - t0
        if return_details is True:
            snapshots.append({k: v.clone() for k, v in b.items()})
        with torch.no_grad():
            last_val = torch.cat([net(normalise(sh.env.obs, sh.scale))[1] for sh in shards])
        adv = torch.zeros(envs, device=dev)
        advs, rets = [None] * horizon, [None] * horizon
        nxt = last_val
        for t in reversed(range(horizon)):           # GAE; a done env's next obs belongs to a fresh episode (auto-reset)
            nonterm = 1.0 - b["done"][t]
            delta = b["rew"][t] + gamma * nxt * nonterm - b["val"][t]
            adv = delta + gamma * lam * nonterm * adv
            advs[t], rets[t] = adv, adv + b["val"][t]
            nxt = b["val"][t]
        b_obs, b_act = b["obs"].reshape(horizon * envs, D), b["act"].reshape(-1)
        b_logp, b_adv, b_ret = b["logp"].reshape(-1), torch.cat(advs), torch.cat(rets)
        b_adv = (b_adv - b_adv.mean()) / (b_adv.std() + 1e-8)
        n = b_obs.shape[0]
        for _ in range(epochs):
            perm = torch.randperm(n, device=dev, generator=gen)
            for mb in perm.chunk(minibatches):
                logits, val = net(b_obs[mb])
                dist = torch.distributions.Categorical(logits=logits)
                ratio = torch.exp(dist.log_prob(b_act[mb]) - b_logp[mb])
                pg = -torch.min(ratio * b_adv[mb], torch.clamp(ratio, 1 - clip, 1 + clip) * b_adv[mb]).mean()
                loss = pg + 0.5 * (val - b_ret[mb]).pow(2).mean() - 0.01 * dist.entropy().mean()
                opt.zero_grad(); loss.backward(); opt.step()
        st = {k: sum(sh.env.stats()[k] for sh in shards) for k in ("sum_return", "episodes", "goals_hit")}
        mean_ret = st["sum_return"] / max(st["episodes"], 1)
        goals_per_ep = st["goals_hit"] / max(st["episodes"], 1)
        history.append((u, mean_ret, goals_per_ep, float(b["rew"].mean())))
        log("update %3d  episodes %8d  mean return so far %+.3f  goals/episode %.3f  mean step reward %+.4f" % (
            u, st["episodes"], mean_ret, goals_per_ep, history[-1][3]))
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all0
    steps = envs * horizon * updates
    log("rollout (%s): %.1f M env-steps/s with the policy in the loop; whole training loop (rollout + PPO update): %.1f M env-steps/s" % (
        mode, steps / max(t_roll, 1e-9) / 1e6, steps / max(t_all, 1e-9) / 1e6))
    details = {"mode": mode, "rollout_env_steps_per_s": steps / max(t_roll, 1e-9), "training_env_steps_per_s": steps / max(t_all, 1e-9),
               "rollout_us_per_step": t_roll * 1e6 / (horizon * updates), "rollout_seconds": t_roll, "total_seconds": t_all,
               "snapshots": snapshots, "final_state": [env_columns(sh.env) for sh in shards] if return_details is True else None,
               "params": [p.detach().clone() for p in net.parameters()] if return_details is True else None}
    for sh in shards:
        sh.graph = None
        sh.env.close()
    return (history, details) if return_details else history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=20)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--mode", choices=("eager", "graph", "pingpong"), default="graph")
    a = ap.parse_args()
    train(envs=a.envs, updates=a.updates, horizon=a.horizon, mode=a.mode)
