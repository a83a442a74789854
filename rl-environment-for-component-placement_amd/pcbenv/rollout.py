"""On-device rollout driver: the counterpart of the reference's `simulate()` loop
(`agent/random/random_policy_square.py:25-58`) and of RLlib's sampler loop around `env.step`.

Trajectories stay on the device as `[T, B, ...]` tensors.  With `policy=None` every step is ONE kernel launch
(`pcbenv_step_sampled`: uniform draw over the legal actions + transition, plus the in-launch reset when the
environment was created with `auto_reset=True`); with a policy callable the action comes from
`policy(obs) -> int tensor [B] (flat) or [B, 3]`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Sequence

import torch


@dataclass
class Trajectory:
    actions: torch.Tensor      # [T, B, 3] int32 (orientation, x, y)
    rewards: torch.Tensor      # [T, B] float64
    dones: torch.Tensor        # [T, B] uint8
    info: Optional[torch.Tensor]  # [T, B, 2] float64 (wirelength, num_intersections; NaN when absent) or None
    obs: Dict[str, torch.Tensor]  # requested observation keys, [T, B, ...] (observation BEFORE the step)

    def episode_returns(self):
        """List of per-environment lists of completed-episode returns (sum of rewards up to each done)."""
        r, d = self.rewards.cpu(), self.dones.cpu().bool()
        out = []
        for b in range(r.shape[1]):
            acc, eps = 0.0, []
            for t in range(r.shape[0]):
                acc += float(r[t, b])
                if d[t, b]:
                    eps.append(acc)
                    acc = 0.0
            out.append(eps)
        return out


def collect(env, num_steps: int, policy: Optional[Callable] = None, t0: int = 0,
            store_obs: Sequence[str] = ()) -> Trajectory:
    B, dev = env.num_envs, env.device
    actions = torch.zeros((num_steps, B, 3), dtype=torch.int32, device=dev)
    rewards = torch.zeros((num_steps, B), dtype=torch.float64, device=dev)
    dones = torch.zeros((num_steps, B), dtype=torch.uint8, device=dev)
    info = torch.zeros((num_steps, B, 2), dtype=torch.float64, device=dev) if env.info else None
    obs = {k: torch.zeros((num_steps,) + tuple(env.obs[k].shape), dtype=env.obs[k].dtype, device=dev) for k in store_obs}
    H, W = env.cfg.height, env.cfg.width
    for t in range(num_steps):
        for k in store_obs:
            obs[k][t].copy_(env.obs[k])
        if policy is None:
            env.rollout_step(t0 + t, out=actions[t])
        else:
            a = policy(env.obs)
            if a.dim() == 1:  # flat -> tuple for the record (utils/environment/env_wrappers.py:80-98)
                a = a.to(torch.int32)
                actions[t, :, 0] = a // (H * W)
                actions[t, :, 1] = (a % (H * W)) // W
                actions[t, :, 2] = a % W
                env.step(a)
            else:
                actions[t].copy_(a)
                env.step(actions[t])
        rewards[t].copy_(env.reward)
        dones[t].copy_(env.done)
        if info is not None:
            info[t].copy_(env.info_raw)
        if not env.auto_reset:
            env.reset_done()
    return Trajectory(actions, rewards, dones, info, obs)


def masked_logits(logits: torch.Tensor, action_mask: torch.Tensor) -> torch.Tensor:
    """`logits += max(log(action_mask), float32.min)` -- how every reference model masks its logits
    (`agent/models/square_model.py:137-139`); `action_mask` is the flat uint8 / float mask."""
    m = action_mask.reshape(logits.shape).to(logits.dtype)
    return logits + torch.clamp(torch.log(m), min=torch.finfo(logits.dtype).min)


def sample_masked_categorical(logits: torch.Tensor, action_mask: torch.Tensor, generator=None) -> torch.Tensor:
    """Flat action per environment from masked logits (never an illegal action while one legal exists)."""
    probs = torch.softmax(masked_logits(logits.float(), action_mask), dim=-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1).to(torch.int32)
