"""A compact PPO loop around the device-resident environments.

Everything stays on the GPU: observations are read in place by the policy, actions go back as a flat int tensor
(`PCBENV_ACTION_FLAT`), trajectories live in `[T, B, ...]` tensors.  Data parallel = one process per GPU, envs
sharded by global index (no env-side communication); the two exchanges are the advantage standardisation
(`distributed.normalize_advantages`: RCCL all-gather over xGMI, or the 3-scalar all-reduce) and the gradient
all-reduce of the (tiny) policy.  RLlib's hyper-parameters are not in the reference tree -> parity unpinned;
this is judged on learning-curve sanity only (tools/ppo_sanity.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.distributed as dist

from .distributed import allreduce_mean_, broadcast_module, normalize_advantages


@dataclass
class PPOConfig:
    rollout_steps: int = 10
    epochs: int = 4
    minibatches: int = 4
    gamma: float = 1.0
    lam: float = 0.95
    clip: float = 0.2
    vf_coef: float = 0.5
    ent_coef: float = 0.01
    lr: float = 3e-4
    adv_mode: str = "all_gather"


def _allreduce_grads(model):
    allreduce_mean_([p.grad for p in model.parameters() if p.grad is not None])


class PPOTrainer:
    def __init__(self, env, policy, cfg: PPOConfig = PPOConfig(), obs_keys=("grid", "pin_grid", "component_grid", "placement_mask", "action_mask")):
        assert env.auto_reset, "create the environment with auto_reset=True"
        self.env, self.policy, self.cfg, self.obs_keys = env, policy, cfg, obs_keys
        broadcast_module(policy)  # data parallel: identical weights and BatchNorm statistics on every rank
        self.opt = torch.optim.Adam(policy.parameters(), lr=cfg.lr)
        self.returns: List[float] = []
        self._ep_ret = torch.zeros(env.num_envs, dtype=torch.float64, device=env.device)
        # Trajectory layout: with at least rollout_steps + 1 slots the environment writes the observation of step t
        # straight into slot t + 1 of its [S, B, ...] tensors and the update reads them in place -- no observation
        # byte is copied.  (With fewer slots every observation is copied once per step, as in round 1.)
        self.in_place = getattr(env, "num_slots", 1) >= cfg.rollout_steps + 1
        self.timing = {"collect_s": 0.0, "update_s": 0.0, "env_steps": 0}

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        env, T, B = self.env, self.cfg.rollout_steps, self.env.num_envs
        if self.in_place:
            if env.slot != 0:  # the observation the last rollout ended on becomes slot 0 of this one (1 / T of the bytes)
                for k in self.obs_keys:
                    env.traj[k][0].copy_(env.traj[k][env.slot])
                env.select_slot(0)
            buf = {k: env.traj[k][:T] for k in self.obs_keys}  # views
        else:
            buf = {k: torch.zeros((T,) + tuple(env.obs[k].shape), dtype=env.obs[k].dtype, device=env.device) for k in self.obs_keys}
        act = torch.zeros((T, B), dtype=torch.int64, device=env.device)
        logp = torch.zeros((T, B), device=env.device)
        val = torch.zeros((T + 1, B), device=env.device)
        rew = torch.zeros((T, B), device=env.device)
        done = torch.zeros((T, B), device=env.device)
        self.policy.eval()
        finished = []
        for t in range(T):
            obs = env.obs  # slot t
            if not self.in_place:
                for k in self.obs_keys:
                    buf[k][t].copy_(obs[k])
            logits, v = self.policy(obs)
            d = torch.distributions.Categorical(logits=logits)
            a = d.sample()
            act[t], logp[t], val[t] = a, d.log_prob(a), v
            if self.in_place:
                env.select_slot(t + 1)
            env.step(a.to(torch.int32))
            rew[t], done[t] = env.reward.float(), env.done.float()
            self._ep_ret += env.reward
            fin = env.done.bool()
            if fin.any():
                finished.append(self._ep_ret[fin].clone())
                self._ep_ret[fin] = 0
        _, val[T] = self.policy(env.obs)
        if finished:
            self.returns.append(float(torch.cat(finished).mean()))
        adv = torch.zeros((T, B), device=env.device)
        last = torch.zeros(B, device=env.device)
        for t in reversed(range(T)):  # GAE; auto-reset: the value after a terminal step belongs to the next episode
            nonterm = 1.0 - done[t]
            delta = rew[t] + self.cfg.gamma * val[t + 1] * nonterm - val[t]
            last = delta + self.cfg.gamma * self.cfg.lam * nonterm * last
            adv[t] = last
        return {"obs": buf, "act": act, "logp": logp, "adv": adv, "ret": adv + val[:T]}

    def update(self, batch) -> Dict[str, float]:
        T, B = batch["act"].shape
        N = T * B
        flat_obs = {k: v.reshape((N,) + tuple(v.shape[2:])) for k, v in batch["obs"].items()}
        act, logp0, ret = batch["act"].reshape(N), batch["logp"].reshape(N), batch["ret"].reshape(N)
        adv = normalize_advantages(batch["adv"].reshape(N), self.cfg.adv_mode)  # global mean / std over all ranks
        # BatchNorm: the rollout computed logp0 with the running statistics, so the surrogate ratio must use them too
        # (its first-epoch value is then exactly 1); the statistics themselves are refreshed once per iteration from
        # a RANDOM minibatch in training mode (a leading slice would be the first timesteps of every environment: mostly
        # empty boards), averaged over the ranks.
        with torch.no_grad():
            self.policy.train()
            ridx = torch.randperm(N, device=act.device)[:max(1, N // self.cfg.minibatches)]
            self.policy({k: o[ridx] for k, o in flat_obs.items()})
            allreduce_mean_([b for b in self.policy.buffers() if b.is_floating_point()])
        self.policy.eval()
        stats = {}
        for _ in range(self.cfg.epochs):
            perm = torch.randperm(N, device=act.device)
            for idx in perm.chunk(self.cfg.minibatches):
                logits, v = self.policy({k: o[idx] for k, o in flat_obs.items()})
                d = torch.distributions.Categorical(logits=logits)
                ratio = torch.exp(d.log_prob(act[idx]) - logp0[idx])
                pg = -torch.min(ratio * adv[idx], torch.clamp(ratio, 1 - self.cfg.clip, 1 + self.cfg.clip) * adv[idx]).mean()
                vf = 0.5 * (v - ret[idx]).pow(2).mean()
                ent = d.entropy().mean()
                loss = pg + self.cfg.vf_coef * vf - self.cfg.ent_coef * ent
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                _allreduce_grads(self.policy)
                torch.nn.utils.clip_grad_norm_(self.policy.parameters(), 1.0)
                self.opt.step()
                stats = {"loss": float(loss.detach()), "pg": float(pg.detach()), "vf": float(vf.detach()), "entropy": float(ent.detach())}
        return stats

    def train(self, iterations: int, log=None):
        import time
        for it in range(iterations):
            torch.cuda.synchronize(self.env.device)
            t0 = time.perf_counter()
            batch = self.collect()
            torch.cuda.synchronize(self.env.device)
            t1 = time.perf_counter()
            stats = self.update(batch)
            torch.cuda.synchronize(self.env.device)
            t2 = time.perf_counter()
            self.timing["collect_s"] += t1 - t0
            self.timing["update_s"] += t2 - t1
            self.timing["env_steps"] += self.cfg.rollout_steps * self.env.num_envs
            if log:
                log(it, self.returns[-1] if self.returns else float("nan"), stats)
        return self.returns
