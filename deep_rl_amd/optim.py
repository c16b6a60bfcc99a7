"""clip_grad_norm_ + Adam fused in one HIP kernel (mi_clip_adam), presented as a torch Optimizer.

Replaces ``nn.utils.clip_grad_norm_(agent.parameters(), max_grad_norm); optimizer.step()`` (ppo.py:191-192) and
keeps the reference's ``optimizer.param_groups[0]["lr"] = new_lr`` annealing idiom (ppo.py:107-108).
"""
import torch

from . import _native as N


class ClipAdam(torch.optim.Optimizer):
    """Also serves dqn.py's plain `optim.Adam(q_network.parameters(), lr)` (dqn.py:68): max_grad_norm = inf disables clipping."""

    def __init__(self, agent, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=float("inf")):
        self.agent = agent
        flat = agent.flat
        super().__init__([torch.nn.Parameter(flat, requires_grad=False)], dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm))
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.step_count = 0

    @torch.no_grad()
    def step(self, grads):
        """grads: flat f32 device tensor (already all-reduced across ranks if world_size > 1)."""
        g = self.param_groups[0]
        self.step_count += 1
        flat = self.agent.flat
        N.check(N.lib().mi_clip_adam(N.ptr(flat), N.ptr(grads), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq), flat.numel(),
                                     self.step_count, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                                     float(g["max_grad_norm"]), N.ptr(self.grad_norm), N.stream_ptr(flat.device)), "mi_clip_adam")


class Adam(torch.optim.Optimizer):
    """optim.Adam over a flat fp32 device buffer (sac.py:108,117,122): one mi_adam launch per step, no clipping."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        flat = getattr(flat, "flat", flat)
        self.flat = flat
        super().__init__([torch.nn.Parameter(flat, requires_grad=False)], dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0

    @torch.no_grad()
    def step(self, grads):
        g = self.param_groups[0]
        self.step_count += 1
        N.check(N.lib().mi_adam(N.ptr(self.flat), N.ptr(grads), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq), self.flat.numel(), self.step_count,
                                float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], N.stream_ptr(self.flat.device)), "mi_adam")
