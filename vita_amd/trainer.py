"""PolicyTrainer — the training step of flare/runners/policy_trainer.py:66-102 rebuilt for one-process-per-GPU data
parallelism on MI355X:

    reference (single device, eager)                         here
    ------------------------------------------------------   -------------------------------------------------------
    zero_grad                                                not needed: gradients are overwritten, not accumulated
    loss, metrics = network.forward(batch)  (5 .item())      normalize kernels + ONE vita_compute_loss_f32 call,
    loss.backward()                                          forward and backward; no host sync
    (no distributed code)                                    4 + L bucketed RCCL all-reduces started by events recorded
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
                 overlap: bool = True, max_bucket_elems: Optional[int] = 8 << 20, rehearse_collectives: bool = False):
        self.config, self.device = config, torch.device(device)
        self.network = policy.to(self.device)
        self.flat, self.flat_grad = policy.flatten_parameters()
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.ema = self.flat.clone() if config.use_ema else None                    # base_policy.py:45-49
        self.rank, self.world = world_info(group)
        self.update_ranges = policy.trainable_ranges()    # frozen modules are skipped by the optimizer, as torch skips grad=None
        self.plan = policy.reduce_plan()              # [(lo, hi, event index)] in the order the backward makes them final
        self.reducer = BucketReducer(self.flat_grad, [(lo, hi) for lo, hi, _ in self.plan], group, max_bucket_elems,
                                     rehearse=rehearse_collectives)
        self.overlap = overlap
        self.events = None
        # a trainable observer's segment is the plan's last entry; its event is recorded by the policy, not by the fused step
        self.observer_segment = getattr(policy, "observer_range", None) is not None
        if self.reducer.active:
            self.events = [torch.cuda.Event() for _ in range(1 + max(ev for _, _, ev in self.plan))]
            for e in self.events:   # torch creates the hipEvent_t lazily, on the first record: the C side needs the handle
                e.record(torch.cuda.current_stream(self.device))
        self.num_training_steps = num_training_steps or config.train.steps
        self.step_counter = 0
        self.losses = torch.zeros(max(L.LOSS_COUNT, L.MF_LOSS_COUNT), dtype=torch.float32, device=self.device)
        self.gnorm_sq = torch.zeros((), dtype=torch.float32, device=self.device)
        self.update_s = 0.0
        if self.reducer.active:
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
        if not net.training:
            net.train()
        batch = net.normalize_inputs(batch)
        batch = net.normalize_targets(batch)
        ev = self.events if (self.events and self.overlap) else None
        fused_ev, obs_ev = (ev[:-1], ev[-1]) if (ev and self.observer_segment) else (ev, None)
        net.loss_and_grads(batch, t=t, pair_i=pair_i, pair_j=pair_j, grad_ready_events=fused_ev, losses=self.losses, observer_event=obs_ev)
        for i, (_, _, ev) in enumerate(self.plan):
            self.reducer.reduce(i, self.events[ev] if (self.events and self.overlap) else None)
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
        net.mark_weights_updated()      # the fused optimizer wrote the parameters through raw pointers: no torch version bump
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

    # -- checkpoints: the reference's two files, in the reference's schema (policy_trainer.py:261-311) ----------------
    def _flat_slices(self):
        """{id(parameter): (offset, numel)} of the hot-path parameters inside the flat buffers."""
        out, off = {}, 0
        for _, p in self.network.flat_named_parameters():
            out[id(p)] = (off, p.numel())
            off += p.numel()
        return out

    def training_state(self) -> dict:
        """`training_state.pt` as the reference writes it (policy_trainer.py:268-274): `optimizer` = torch.optim.AdamW
        .state_dict() over `network.parameters()` in module order (per-parameter step / exp_avg / exp_avg_sq cut from the flat
        moment buffers; parameters that never got a gradient -- the frozen Normalize statistics, frozen modules -- carry no
        state, as in torch), `lr_scheduler` = LambdaLR.state_dict(), `ema` = diffusers EMAModel.state_dict() keys
        (shadow_params over all parameters, optimization_step, decay, power, ...), `step`."""
        net, step = self.network, self.step_counter
        params = list(net.parameters())
        slices = self._flat_slices()
        trainable = [(a, b) for a, b in self.update_ranges]
        opt = net.get_optimizer()
        sch = net.get_scheduler(opt, self.num_training_steps)
        if step > 0:
            for p in params:
                sl = slices.get(id(p))
                if sl is None or not any(a <= sl[0] < b for a, b in trainable):
                    continue
                off, n = sl
                opt.state[p] = {"step": torch.tensor(float(step)),
                                "exp_avg": self.exp_avg[off:off + n].view_as(p).detach().clone().cpu(),
                                "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).detach().clone().cpu()}
        lr = self.current_lr()
        for g in opt.param_groups:
            g["lr"] = lr
        sch.last_epoch, sch._step_count, sch._last_lr = step, step + 1, [lr]
        ema = None
        if self.ema is not None:
            shadow = []
            for p in params:
                sl = slices.get(id(p))
                shadow.append((self.ema[sl[0]:sl[0] + sl[1]].view_as(p) if sl is not None else p.data).detach().clone().cpu())
            ema = {"decay": 0.9999, "min_decay": 0.0, "optimization_step": step, "update_after_step": 0, "use_ema_warmup": False,
                   "inv_gamma": 1.0, "power": float(self.config.ema_power), "shadow_params": shadow}
        return {"optimizer": opt.state_dict(), "lr_scheduler": sch.state_dict(), "ema": ema, "step": step}

    def save_checkpoint(self, directory: str):
        import os
        self.network.save_pretrained(directory)
        torch.save(self.training_state(), os.path.join(directory, "training_state.pt"))

    def load_training_state(self, st: dict):
        """Inverse of training_state(); also takes a file the reference wrote.  Parameter i of the file is parameter i of
        `network.parameters()`; a file with MORE parameters (the reference's ResNet18 observer sits between the Normalize
        statistics and obs_encoder, vita_policy.py:43-50,77) is aligned from the end, where the hot-path modules are.  Every
        moment tensor's shape is checked against its parameter before anything is written."""
        net = self.network
        params = list(net.parameters())
        slices = self._flat_slices()
        opt_sd = st["optimizer"]
        if "param_groups" not in opt_sd or "state" not in opt_sd:
            raise L.VitaHipError("training_state.pt: 'optimizer' is not a torch optimizer state_dict (round-1 native files are not supported)")
        ids = [i for g in opt_sd["param_groups"] for i in g["params"]]
        n_file, n_here = len(ids), len(params)
        if n_file < n_here:
            raise L.VitaHipError(f"training_state.pt covers {n_file} parameters, this policy has {n_here}")
        n_head = next((i for i, p in enumerate(params) if id(p) in slices), n_here)   # parameters in front of the hot path
        def file_index(i):
            return ids[i] if i < n_head and n_file == n_here else ids[i + (n_file - n_here)]
        todo = []
        for i, p in enumerate(params):
            sl = slices.get(id(p))
            if sl is None:
                continue
            ent = opt_sd["state"].get(file_index(i))
            if ent is None:
                continue
            for key in ("exp_avg", "exp_avg_sq"):
                if tuple(ent[key].shape) != tuple(p.shape):
                    raise L.VitaHipError(f"training_state.pt: {key} of parameter {i} has shape {tuple(ent[key].shape)}, expected {tuple(p.shape)}")
            todo.append((sl, ent))
        ema_sd = st.get("ema")
        shadows = None
        if self.ema is not None and ema_sd:
            sp = ema_sd["shadow_params"]
            if len(sp) != n_file:
                raise L.VitaHipError(f"training_state.pt: ema.shadow_params has {len(sp)} entries, optimizer covers {n_file}")
            shadows = []
            for i, p in enumerate(params):
                sl = slices.get(id(p))
                if sl is None:
                    continue
                t = sp[i if (i < n_head and n_file == n_here) else i + (n_file - n_here)]
                if tuple(t.shape) != tuple(p.shape):
                    raise L.VitaHipError(f"training_state.pt: ema shadow {i} has shape {tuple(t.shape)}, expected {tuple(p.shape)}")
                shadows.append((sl, t))
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for (off, n), ent in todo:
            self.exp_avg[off:off + n].copy_(ent["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(ent["exp_avg_sq"].reshape(-1))
        if shadows is not None:
            for (off, n), t in shadows:
                self.ema[off:off + n].copy_(t.reshape(-1))
        self.step_counter = int(st.get("step", 0))

    def load_checkpoint(self, directory: str):
        """policy_trainer.py:276-311.  The file is read with torch.load(weights_only=True): nothing in it is executed.  A
        reference file that holds non-tensor objects the safe loader refuses (the reference itself loads with
        weights_only=False) raises here instead of being unpickled."""
        import os
        self.network.load_pretrained(directory)
        st = torch.load(os.path.join(directory, "training_state.pt"), map_location="cpu", weights_only=True)
        self.load_training_state(st)
        self.network.mark_weights_updated()
        return self.step_counter

    def ema_state_dict(self) -> dict:
        """EMA weights under the reference's state-dict keys (what validate_online evaluates)."""
        out, off = {}, 0
        for k, p in self.network.flat_named_parameters():
            out[k] = self.ema[off:off + p.numel()].view_as(p)
            off += p.numel()
        return out
