"""PolicyTrainer — the training step of flare/runners/policy_trainer.py:66-102 rebuilt for one-process-per-GPU data
parallelism on MI355X:

    reference (single device, eager)                         here
    ------------------------------------------------------   -------------------------------------------------------
    zero_grad                                                not needed: gradients are overwritten, not accumulated
    loss, metrics = network.forward(batch)  (5 .item())      normalize kernels + ONE vita_compute_loss_f32 call,
    loss.backward()                                          forward and backward; no host sync
    (no distributed code)                                    3 bucketed RCCL all-reduces started by events recorded
                                                             inside the backward, on a side stream
    clip_grad_norm_(1.0)         pass over all params        vita_sumsq_f32 (one read of the flat gradient)
    optimizer.step()  AdamW      pass over p, g, m, v        }
    lr_scheduler.step() cosine   host                        } ONE fused kernel: x 1/N, clip, AdamW, EMA
    ema.step()                   pass over p, ema            }
    loss.item(), grad_norm.item()  2 host syncs              metrics stay on the device until asked for

Orchestration around it (logging, checkpoints cadence, rollouts, wandb) is out of scope.
"""
from __future__ import annotations

import time
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .parallel import BucketReducer, world_info
from .policy import VitaPolicy, ema_decay, lr_lambda


class PolicyTrainer:
    def __init__(self, config, policy: VitaPolicy, device, num_training_steps: Optional[int] = None, group=None,
                 overlap: bool = True, max_bucket_elems: Optional[int] = 8 << 20):
        self.config, self.device = config, torch.device(device)
        self.network = policy.to(self.device)
        self.flat, self.flat_grad = policy.flatten_parameters()
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.ema = self.flat.clone() if config.use_ema else None                    # base_policy.py:45-49
        self.rank, self.world = world_info(group)
        self.update_ranges = policy.trainable_ranges()    # frozen modules are skipped by the optimizer, as torch skips grad=None
        self.reducer = BucketReducer(self.flat_grad, policy.bucket_ranges, group, max_bucket_elems)
        self.overlap = overlap
        self.events = None
        if self.world > 1:
            self.events = [torch.cuda.Event() for _ in range(3)]
            for e in self.events:   # torch creates the hipEvent_t lazily, on the first record: the C side needs the handle
                e.record(torch.cuda.current_stream(self.device))
        self.num_training_steps = num_training_steps or config.train.steps
        self.step_counter = 0
        self.losses = torch.zeros(max(L.LOSS_COUNT, L.MF_LOSS_COUNT), dtype=torch.float32, device=self.device)
        self.gnorm_sq = torch.zeros((), dtype=torch.float32, device=self.device)
        self.update_s = 0.0
        if self.world > 1:
            self.broadcast_parameters(group)

    def broadcast_parameters(self, group=None):
        """Replicas start identical (rank 0's initialisation), as DistributedDataParallel does."""
        import torch.distributed as dist
        dist.broadcast(self.flat, src=0, group=group)
        if self.ema is not None:
            self.ema.copy_(self.flat)

    def current_lr(self) -> float:
        c = self.config
        return float(c.optimizer_lr) * lr_lambda(c.scheduler_name, self.step_counter, c.scheduler_warmup_steps, self.num_training_steps)

    def train_step(self, batch, *, t=None, pair_i=None, pair_j=None):
        """policy_trainer.py:66-102 for this rank's shard of the global batch.  Enqueues the whole step and returns without
        a host sync; `metrics()` fetches the numbers."""
        t0 = time.perf_counter()
        net, c = self.network, self.config
        batch = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        net.train()
        batch = net.normalize_inputs(batch)
        batch = net.normalize_targets(batch)
        net.loss_and_grads(batch, t=t, pair_i=pair_i, pair_j=pair_j, grad_ready_events=self.events if self.overlap else None,
                           losses=self.losses)
        for i in range(len(self.reducer.segments)):
            self.reducer.reduce(i, self.events[i] if (self.events and self.overlap) else None)
        scale = self.reducer.finish()
        ops.sumsq(self.flat_grad, out=self.gnorm_sq)
        step = self.step_counter + 1
        for a, b in self.update_ranges:
            ops.adamw_ema_step(self.flat[a:b], self.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b],
                               self.ema[a:b] if self.ema is not None else None, step=step, lr=self.current_lr(),
                               betas=tuple(c.optimizer_betas), eps=c.optimizer_eps, weight_decay=c.optimizer_weight_decay,
                               gnorm_sq=self.gnorm_sq, max_norm=float(c.train.grad_clip_norm), grad_scale=scale,
                               ema_decay=ema_decay(step) if self.ema is not None else 0.0)
        self.step_counter = step
        self.update_s = time.perf_counter() - t0
        return self.losses

    def metrics(self) -> dict:
        """One device->host copy: the metric keys of vita_policy.py:132-186 plus grad_norm / lr (policy_trainer.py:97-99).
        Under data parallelism the loss values are this rank's shard means."""
        vals = self.losses.cpu().tolist()
        names = L.MF_LOSS_NAMES if getattr(self.network, "mean_flow", False) else L.LOSS_NAMES
        out = {k: vals[i] for i, k in enumerate(names)}
        out["grad_norm"] = float(self.gnorm_sq.sqrt().item()) / self.world
        out["lr"] = self.current_lr()
        out["update_s"] = self.update_s
        return out

    # -- checkpoints: the reference's two files (policy_trainer.py:261-274) ------------------------------------
    def save_checkpoint(self, directory: str):
        import os
        self.network.save_pretrained(directory)
        torch.save({"optimizer": {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()},
                    "ema": self.ema.cpu() if self.ema is not None else None, "step": self.step_counter},
                   os.path.join(directory, "training_state.pt"))

    def load_checkpoint(self, directory: str):
        import os
        self.network.load_pretrained(directory)
        st = torch.load(os.path.join(directory, "training_state.pt"), weights_only=True)
        self.exp_avg.copy_(st["optimizer"]["exp_avg"])
        self.exp_avg_sq.copy_(st["optimizer"]["exp_avg_sq"])
        if self.ema is not None and st["ema"] is not None:
            self.ema.copy_(st["ema"])
        self.step_counter = int(st["step"])

    def ema_state_dict(self) -> dict:
        """EMA weights under the reference's state-dict keys (what validate_online evaluates)."""
        out, off = {}, 0
        for k, p in self.network.hot_path_named_parameters():
            out[k] = self.ema[off:off + p.numel()].view_as(p)
            off += p.numel()
        return out
