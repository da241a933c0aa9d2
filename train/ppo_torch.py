#!/usr/bin/env python3
"""Trainer glue demo: a minimal PPO loop in plain PyTorch driving ShipVecEnv with zero-copy
device tensors — the role SubprocVecEnv + PPO2 play in the reference's train/stable_baselines/ppo.py:84-123, without
stable-baselines (absent from this image).  The env side is the only point: observations, rewards and dones never
leave the GPU; the policy is a small MLP in fp32.

    python train/ppo_torch.py --envs 4096 --updates 20

This is NOT a re-implementation of the reference's trainers (out of scope); it exists to show the batched env plugs
into a GPU-resident training loop and to give `tests/` an end-to-end caller.
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


def train(envs=4096, updates=20, horizon=64, epochs=2, minibatches=4, lr=3e-4, gamma=0.99, lam=0.95, clip=0.2,
          device="cuda:0", seed=0, log=print):
    torch.manual_seed(seed)
    game_config = GameConfig
    env = ShipVecEnv(envs, game_config, EnvConfig, device=device, n_maps=64)   # was: SubprocVecEnv([make_env()]*n)
    D, A = env.states_history, env.action_space.n
    scale = torch.full((D,), float(max(env.bounds)), dtype=torch.float64, device=device)
    net = ActorCritic(D, A).to(device)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    obs = env.reset_tensor().clone()
    history = []
    t_env = 0.0
    for u in range(updates):
        buf_obs, buf_act, buf_logp, buf_val, buf_rew, buf_done = [], [], [], [], [], []
        for t in range(horizon):
            x = normalise(obs, scale)
            with torch.no_grad():
                logits, val = net(x)
                dist = torch.distributions.Categorical(logits=logits)
                act = dist.sample()
            t0 = time.perf_counter()
            nobs, rew, done, _ = env.step_tensor(act.to(torch.int32))
            t_env += time.perf_counter() - t0
            buf_obs.append(x); buf_act.append(act); buf_logp.append(dist.log_prob(act)); buf_val.append(val)
            buf_rew.append(rew.float().clone()); buf_done.append(done.float().clone())
            obs = nobs.clone()
        with torch.no_grad():
            _, last_val = net(normalise(obs, scale))
        adv = torch.zeros(envs, device=device)
        advs, rets = [None] * horizon, [None] * horizon
        nxt = last_val
        for t in reversed(range(horizon)):           # GAE; a done env's next obs belongs to a fresh episode (auto-reset)
            nonterm = 1.0 - buf_done[t]
            delta = buf_rew[t] + gamma * nxt * nonterm - buf_val[t]
            adv = delta + gamma * lam * nonterm * adv
            advs[t], rets[t] = adv, adv + buf_val[t]
            nxt = buf_val[t]
        b_obs, b_act = torch.cat(buf_obs), torch.cat(buf_act)
        b_logp, b_adv, b_ret = torch.cat(buf_logp), torch.cat(advs), torch.cat(rets)
        b_adv = (b_adv - b_adv.mean()) / (b_adv.std() + 1e-8)
        n = b_obs.shape[0]
        for _ in range(epochs):
            perm = torch.randperm(n, device=device)
            for mb in perm.chunk(minibatches):
                logits, val = net(b_obs[mb])
                dist = torch.distributions.Categorical(logits=logits)
                ratio = torch.exp(dist.log_prob(b_act[mb]) - b_logp[mb])
                pg = -torch.min(ratio * b_adv[mb], torch.clamp(ratio, 1 - clip, 1 + clip) * b_adv[mb]).mean()
                loss = pg + 0.5 * (val - b_ret[mb]).pow(2).mean() - 0.01 * dist.entropy().mean()
                opt.zero_grad(); loss.backward(); opt.step()
        st = env.stats()
        mean_ret = st["sum_return"] / max(st["episodes"], 1)
        goals_per_ep = st["goals_hit"] / max(st["episodes"], 1)
        history.append((u, mean_ret, goals_per_ep, float(torch.cat(buf_rew).mean())))
        log("update %3d  episodes %8d  mean return so far %+.3f  goals/episode %.3f  mean step reward %+.4f" % (
            u, st["episodes"], mean_ret, goals_per_ep, history[-1][3]))
    torch.cuda.synchronize()
    log("env stepping: %.1f M env-steps/s inside the training loop (launch-to-launch, policy in the loop)" % (
        envs * horizon * updates / max(t_env, 1e-9) / 1e6))
    env.close()
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=20)
    ap.add_argument("--horizon", type=int, default=64)
    a = ap.parse_args()
    train(envs=a.envs, updates=a.updates, horizon=a.horizon)
