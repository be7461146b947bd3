"""Configuration for the `policy=vita` hot path without Hydra/OmegaConf (neither is in this image).

`Config` gives attribute access over nested dicts, enough for every field the reference reads
(vita_policy.py:41-97,153,162; action_ae.py:337-383; base_policy.py:22-30; policy_trainer.py).  A real
OmegaConf DictConfig works too: the policy only uses attribute access and `.get`.

The defaults below restate flare/configs/default_policy.yaml, policy/vita.yaml and task/*.yaml (values only:
the reference's files are read, never forked — `load_reference_config` parses them in place when the
reference tree is present).
"""
from __future__ import annotations

import copy
import os
from typing import Any, Mapping


class Config(dict):
    """dict with attribute access, recursively."""

    def __init__(self, d: Mapping[str, Any] | None = None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config(v) if isinstance(v, Mapping) and not isinstance(v, Config) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def merge(self, other: Mapping[str, Any]) -> "Config":
        for k, v in other.items():
            if isinstance(v, Mapping) and isinstance(self.get(k), Config):
                self[k].merge(v)
            else:
                self[k] = v
        return self

    def set_path(self, dotted: str, value) -> "Config":
        """`cfg.set_path("policy.vita.decode_flow_latents", False)` — the CLI override form of README.md:167-180."""
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, Config())
        node[parts[-1]] = value
        return self


# flare/configs/task/*.yaml: fps (the dataset's frame rate: delta timestamps = index / fps, dataset_utils.py:85-92)
TASK_FPS = {"pusht": 10, "hook_package": 8.333333333, "thread_needle": 8.333333333, "robomimic_square": 20}

# flare/configs/task/*.yaml:3-4: (resize_shape, crop_shape) of the observer
TASK_IMAGE_SHAPES = {"pusht": ([96, 96], [84, 84]), "hook_package": ([240, 320], [224, 308]), "thread_needle": ([240, 320], [224, 308]),
                     "robomimic_square": ([256, 256], [224, 224])}

# flare/configs/task/*.yaml: (image_keys, state_dim, action_dim)
TASKS = {
    "pusht": (["observation.image"], 2, 2),                                             # pusht.yaml:27-33
    "hook_package": (["observation.images.zed_cam_left"], 21, 21),                      # hook_package.yaml:27-32
    "thread_needle": (["observation.images.zed_cam_left"], 21, 21),                     # thread_needle.yaml:27-32
    "robomimic_square": (["observation.images.agentview_image"], 43, 7),                # robomimic_square.yaml:22-27
}


def default_config(task: str = "hook_package") -> Config:
    if task not in TASKS:
        raise ValueError(f"unknown task {task!r}; known: {sorted(TASKS)}")
    image_keys, state_dim, action_dim = TASKS[task]
    return Config({
        # default_policy.yaml:15-51
        "seed": 42, "device": "cuda:0",
        "resize_shape": TASK_IMAGE_SHAPES[task][0], "crop_shape": TASK_IMAGE_SHAPES[task][1],
        "optimizer_lr": 1e-4, "optimizer_lr_backbone": 1e-5, "optimizer_betas": [0.95, 0.999], "optimizer_eps": 1e-8,
        "optimizer_weight_decay": 1e-6, "scheduler_name": "cosine", "scheduler_warmup_steps": 500,
        "use_ema": True, "ema_power": 0.75,
        "train": {"steps": 100000, "batch_size": 128, "grad_clip_norm": 1.0},
        "val": {"batch_size": 512, "eval_n_envs": 25},
        "task": {"name": task, "image_keys": image_keys, "state_key": "observation.state", "action_key": "action",
                 "state_dim": state_dim, "action_dim": action_dim, "fps": TASK_FPS[task]},
        # policy/vita.yaml:3-57
        "policy": {
            "name": "vita", "runner": "policy_trainer", "action_horizon": 8, "pred_horizon": 16, "obs_horizon": 1,
            "observer": {"name": "resnet18", "tokenize": False},
            "vita": {"decode_flow_latents": True, "consistency_weight": 1.0, "enc_contrastive_weight": 0.0,
                     "flow_contrastive_weight": 0.0, "latent_dim": 512},
            "flow_matcher": {"name": "exact", "sigma": 0.0, "num_sampling_steps": 6},
            "flow_net": {"name": "simple", "hidden_dim": 512, "num_layers": 4, "mlp_ratio": 4, "dropout": 0.0},
            "action_ae": {"kl_weight": 0.0, "use_variational": False, "enc_recon_weight": 0.5, "flow_recon_weight": 0.5,
                          "recon_loss_type": "l1", "freeze_encoder": False, "freeze_decoder": False,
                          "net": {"encoder_type": "cnn", "decoder_type": "simple", "enc_hidden_dim": 512,
                                  "dec_hidden_dim": 512, "latent_dim": 512, "num_heads": 8, "mlp_ratio": 4, "dropout": 0.0,
                                  "num_layers": 4}},
        },
    })


def load_reference_config(flare_root: str, task: str = "hook_package") -> Config:
    """Compose default_policy.yaml + task/<task>.yaml + policy/vita.yaml from a reference checkout (plain YAML,
    safe loader; `${policy.vita.latent_dim}` is the only interpolation the hot path reads)."""
    import yaml
    cfgdir = os.path.join(flare_root, "configs")
    cfg = Config()
    for rel in ("default_policy.yaml", os.path.join("task", task + ".yaml"), os.path.join("policy", "vita.yaml")):
        with open(os.path.join(cfgdir, rel)) as f:
            d = yaml.safe_load(f) or {}
        d.pop("defaults", None)
        cfg.merge(d)
    net = cfg.policy.action_ae.net
    if isinstance(net.get("latent_dim"), str):
        net["latent_dim"] = cfg.policy.vita.latent_dim
    for k in ("optimizer_lr", "optimizer_eps", "optimizer_weight_decay", "optimizer_lr_backbone"):
        if isinstance(cfg.get(k), str):   # YAML 1.1 reads "1e-4" as a string
            cfg[k] = float(cfg[k])
    return cfg
