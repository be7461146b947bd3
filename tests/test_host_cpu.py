"""Host logic that needs no GPU: config, registry, state-dict keys (pinned to the reference's own modules through the
golden fixture), flow-matcher table, OT pairing, schedules, sharding; and that the product path refuses to run
without the GPU instead of falling back."""
import math
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle import vita_oracle as O
from vita_amd import _lib
from vita_amd.config import Config, default_config, TASKS
from vita_amd.factory import get_policy_class, get_runner_class
from vita_amd.flow import FLOW_MATCHER_CLASSES, exact_ot_pairing, get_flow_matcher
from vita_amd.parallel import shard_range
from vita_amd.policy import VitaPolicy, ema_decay, lr_lambda


def stats_for(cfg):
    s, a = cfg.task.state_dim, cfg.task.action_dim
    return {cfg.task.state_key: {"mean": [0.1] * s, "std": [0.9] * s}, cfg.task.action_key: {"mean": [0.2] * a, "std": [1.1] * a},
            cfg.task.image_keys[0]: {"mean": [[[0.485]], [[0.456]], [[0.406]]], "std": [[[0.229]], [[0.224]], [[0.225]]]}}


def small_cfg(encoder="cnn"):
    cfg = default_config("robomimic_square")
    cfg.policy.vita.latent_dim = 32
    cfg.policy.action_ae.net.merge({"latent_dim": 32, "enc_hidden_dim": 32, "dec_hidden_dim": 32, "encoder_type": encoder,
                                    "num_layers": 4 if encoder == "cnn" else 2})
    cfg.policy.flow_net.merge({"hidden_dim": 32, "num_layers": 2})
    return cfg


def test_config_attribute_access_and_overrides():
    cfg = default_config("hook_package")
    assert cfg.policy.flow_matcher.num_sampling_steps == 6 and cfg.policy.action_ae.net.latent_dim == 512
    assert cfg.task.action_dim == 21 and cfg.train.batch_size == 128 and tuple(cfg.optimizer_betas) == (0.95, 0.999)
    cfg.set_path("policy.vita.decode_flow_latents", False)      # README.md:167-180 style override
    assert cfg.policy.vita.decode_flow_latents is False
    with pytest.raises(ValueError):
        default_config("no_such_task")
    assert isinstance(Config({"a": {"b": 1}}).a, Config)


def test_registry_keeps_the_reference_names():
    assert get_policy_class("vita") is VitaPolicy                # vita_policy.py:36
    assert get_runner_class("policy_trainer").__name__ == "PolicyTrainer"
    with pytest.raises(ValueError):
        get_policy_class("act")


@pytest.mark.parametrize("tag,encoder", [("tiny_A7_cnn", "cnn"), ("tiny_A7_simple", "simple")])
def test_state_dict_keys_and_shapes_match_the_reference_modules(golden_dir, tag, encoder):
    """tiny.safetensors holds state dicts dumped from the reference's own SimpleFlowNet / CNNActionEncoder /
    SimpleActionEncoder / SimpleActionDecoder (tools/gen_golden.py): same keys, same shapes."""
    ref = {}
    with safe_open(os.path.join(golden_dir, "tiny.safetensors"), framework="pt") as f:
        for k in f.keys():
            if k.startswith(tag + "/sd/"):
                ref[k[len(tag) + 4:]] = tuple(f.get_tensor(k).shape)
    cfg = small_cfg(encoder)
    pol = VitaPolicy(cfg, stats_for(cfg))
    mine = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    hot = {k: v for k, v in mine.items() if k.startswith(("flow_net.", "action_encoder.", "action_decoder."))}
    ref_hot = {k: v for k, v in ref.items() if not k.startswith("obs_encoder.")}
    assert set(hot) == set(ref_hot)
    for k in ref_hot:
        if "time_embed" in k:      # the fixture uses time_embed_dim=16; the policy the reference's fixed 256
            continue
        assert hot[k] == ref_hot[k], k
    assert mine["obs_encoder.weight"] == (32, 512 + 43) and "normalize_inputs.buffer_observation_state.mean" in mine
    assert "normalize_targets.buffer_action.std" in mine and "unnormalize_outputs.buffer_action.mean" in mine
    assert sum(p.numel() for _, p in pol.hot_path_named_parameters()) == sum(
        v.numel() for k, v in pol.state_dict().items() if not k.startswith(("normalize", "unnormalize")))


def test_default_policy_parameter_count_matches_survey():
    cfg = default_config("hook_package")
    pol = VitaPolicy(cfg, stats_for(cfg))
    by = lambda pre: sum(p.numel() for k, p in pol.named_parameters() if k.startswith(pre))  # noqa: E731
    assert by("flow_net.") == 12_865_024 and by("action_encoder.") == 4_250_624 and by("action_decoder.") == 2_536_272
    # init quirk kept on purpose: modulators are Xavier, not zero
    assert pol.flow_net.layers[0].time_modulator[1].weight.abs().mean() > 1e-3
    assert float(pol.flow_net.time_embed[1].bias.abs().max()) == 0.0


def test_constructor_errors_mirror_the_reference():
    cfg = small_cfg()
    cfg.policy.flow_net.name = "unet"
    with pytest.raises(ValueError, match="Unsupported flow_net"):
        VitaPolicy(cfg, stats_for(cfg))
    for ok in ("simple", "simple_flow_net"):
        cfg.policy.flow_net.name = ok
        VitaPolicy(cfg, stats_for(cfg))
    cfg.policy.action_ae.recon_loss_type = "huber"
    with pytest.raises(ValueError, match="Unsupported recon_loss_type"):
        VitaPolicy(cfg, stats_for(cfg))
    cfg.policy.action_ae.recon_loss_type = "l1"
    cfg.policy.action_ae.net.encoder_type = "mamba"
    with pytest.raises(ValueError, match="Unknown encoder type"):
        VitaPolicy(cfg, stats_for(cfg))
    cfg.policy.action_ae.net.encoder_type = "cnn"
    cfg.policy.flow_matcher.name = "nope"
    with pytest.raises(ValueError, match="Invalid flow matcher name"):
        VitaPolicy(cfg, stats_for(cfg))


def test_flow_matcher_table():
    assert set(FLOW_MATCHER_CLASSES) == {"conditional", "target", "schrodinger", "exact", "mean", "improved_mean", "consistency"}
    assert get_flow_matcher("exact", sigma=0.0, num_sampling_steps=6).pairing == "exact"
    assert get_flow_matcher("conditional").num_sampling_steps == 6
    from vita_amd import _lib as L
    tg = get_flow_matcher("target", sigma=0.0, num_sampling_steps=6)                        # flow_matchers.py:87-89
    assert tg.kind == L.FM_TARGET and tg.pairing == "identity" and tg.needs_noise
    assert get_flow_matcher("exact", sigma=0.1).needs_noise and not get_flow_matcher("exact", sigma=0.0).needs_noise
    sb = get_flow_matcher("schrodinger", sigma=0.5)                                         # :92-94
    assert sb.kind == L.FM_SCHRODINGER and sb.pairing == "exact" and sb.needs_noise and get_flow_matcher("schrodinger").sigma == 1.0
    with pytest.raises(ValueError):
        get_flow_matcher("schrodinger", sigma=0.0)       # vita.yaml's sigma: 0.0 is rejected by torchcfm's bridge matcher too
    # sampler-only matchers: constructor arguments as in the reference, training refused
    m = get_flow_matcher("mean", sigma=0.0, num_sampling_steps=1, flow_ratio=0.25)          # MeanFlow swallows **kwargs (:56)
    assert not m.inference_only and m.flow_ratio == 0.25 and not m.use_imf      # MeanFlow training is built (fused MeanFlow step)
    im = get_flow_matcher("improved_mean")
    assert im.use_imf and not im.inference_only and not get_flow_matcher("mean", dispersive_loss_weight=0.1).inference_only
    tt, rr = m.sample_t_r(64, "cpu")                                                          # mean_flow_matcher.py:87-107
    assert tt.shape == (64,) and bool((tt >= rr).all()) and int((tt == rr).sum()) >= 16 and float(tt.max()) < 1 and float(rr.min()) > 0
    with pytest.raises(NotImplementedError):
        m.compute_loss(None, torch.zeros(2, 4))
    with pytest.raises(ValueError):
        m.sample(None, (2, 4), "cpu")                                                         # needs `start` (:205-206)
    c = get_flow_matcher("consistency", eps=0.02)
    assert c.inference_only and c.eps == 0.02 and c.sigma_var == 1.0 and abs(c.sigma_t(0.25) - 0.75) < 1e-12
    with pytest.raises(TypeError):
        get_flow_matcher("consistency", sigma=0.0)                                            # no such argument in the reference either


def test_exact_ot_pairing_is_a_sample_from_the_optimal_permutation():
    g = torch.Generator().manual_seed(0)
    B = 16
    x1 = torch.randn(B, 8, generator=g)
    perm = torch.randperm(B, generator=g)
    x0 = x1[perm] + 1e-3 * torch.randn(B, 8, generator=g)     # x0[k] is next to x1[perm[k]]: the optimal plan is obvious
    np.random.seed(0)
    i, j = exact_ot_pairing(x0, x1)
    assert i.dtype == torch.int64 and i.shape == (B,) and int(i.min()) >= 0 and int(i.max()) < B
    assert torch.equal(j, perm[i])
    assert len(set(i.tolist())) < B          # drawn with replacement: repeats are expected


def test_schedules_match_the_oracle_restatement():
    for s in (0, 1, 250, 499, 500, 501, 50_000, 99_999, 100_000):
        assert lr_lambda("cosine", s, 500, 100_000) == O.cosine_with_warmup_lambda(s, 500, 100_000)
    assert lr_lambda("constant", 3, 500, 1000) == 1.0
    with pytest.raises(ValueError):
        lr_lambda("linear", 0, 1, 2)
    assert ema_decay(1) == 0.0 and abs(ema_decay(2) - 2 / 11) < 1e-12 and ema_decay(10 ** 9) == 0.9999
    for s in (1, 2, 17, 100000):
        assert ema_decay(s) == O.ema_decay(s)


@pytest.mark.parametrize("n,world", [(4096, 8), (4096, 3), (5, 8), (0, 2), (1, 1)])
def test_shard_range_covers_rows_exactly_once(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and b >= a and d >= c
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_no_cpu_fallback():
    """The product path must fail loudly, not compute on the CPU."""
    cfg = small_cfg()
    pol = VitaPolicy(cfg, stats_for(cfg))
    batch = {"observation.state": torch.zeros(2, 1, 43), "observation.features": torch.zeros(2, 1, 512),
             "action": torch.zeros(2, 16, 7)}
    with pytest.raises(_lib.VitaHipError):
        pol.generate_actions(batch)
    with pytest.raises(_lib.VitaHipError):
        pol.forward(batch)
    with pytest.raises(_lib.VitaHipError):
        pol.flatten_parameters()
    with pytest.raises(KeyError):
        pol.generate_actions({"observation.state": torch.zeros(2, 1, 43)})


def test_task_table_matches_baseline_dims():
    # O = 512*cams + state_dim
    for task, O_ in (("pusht", 514), ("hook_package", 533), ("robomimic_square", 555), ("thread_needle", 533)):
        cams, s, _ = TASKS[task]
        assert 512 * len(cams) + s == O_


def test_oracle_matcher_paths_are_consistent():
    """torchcfm's three Gaussian paths as restated in the oracle (UNPINNED): at sigma = 0 the target path's ut equals
    x1 - eps exactly in real arithmetic; the bridge's ut reduces to x1 - x0 at t = 1/2; conditional with sigma adds sigma*eps."""
    from oracle import vita_oracle as O
    g = torch.Generator().manual_seed(0)
    x0, x1, eps = (torch.randn(6, 5, generator=g, dtype=torch.float64) for _ in range(3))
    t = torch.rand(6, generator=g, dtype=torch.float64) * 0.8 + 0.1
    _, xt, ut = O.cfm_sample_location_and_conditional_flow(x0, x1, t, 0.0, eps, "target")
    assert torch.allclose(ut, x1 - eps, atol=1e-12) and torch.allclose(xt, t[:, None] * x1 + (1 - t[:, None]) * eps, atol=1e-12)
    half = torch.full((6,), 0.5, dtype=torch.float64)
    _, xt, ut = O.cfm_sample_location_and_conditional_flow(x0, x1, half, 0.7, eps, "schrodinger")
    assert torch.allclose(ut, x1 - x0, atol=1e-12) and torch.allclose(xt, 0.5 * (x0 + x1) + 0.35 * eps, atol=1e-12)
    _, xt, _ = O.cfm_sample_location_and_conditional_flow(x0, x1, t, 0.3, eps, "conditional")
    assert torch.allclose(xt, t[:, None] * x1 + (1 - t[:, None]) * x0 + 0.3 * eps, atol=1e-12)


def test_bench_self_launches_one_process_per_gpu():
    """`python bench.py --gpus N` without a launcher (the driver's command form) starts N rank processes itself with the
    torch.distributed.run environment; --dry-launch makes each print that environment and exit before touching a GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-launch"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(ln) for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("{")]
    assert sorted(d["rank"] for d in lines) == [0, 1, 2, 3] and all(d["local_rank"] == d["rank"] and d["world"] == 4 for d in lines)
    assert len({d["pid"] for d in lines}) == 4 and len({d["ppid"] for d in lines}) == 1
    assert len({d["master_port"] for d in lines}) == 1 and all(d["master_addr"] == "127.0.0.1" for d in lines)
    assert [json.loads(ln)["rank"] for ln in r.stdout.splitlines() if ln.startswith("{")] == [0]      # only rank 0 on stdout
    # under a launcher (WORLD_SIZE set) nothing is spawned, and a mismatch with --gpus is an error
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_reduce_plan_covers_the_flat_buffer_once_in_firing_order():
    """The data-parallel reduce plan: every hot-path parameter in exactly one segment, one contiguous segment per event, in the
    order in which vita_compute_loss_f32 records the events under grad_ready_layers: decoder (0), one per flow-net layer's fc1 / fc2
    gradients (3 + l), encoders (2), time path and modulators (1), outer projections and final norm (3 + L)."""
    import torch
    from vita_amd.config import default_config
    from vita_amd.policy import VitaPolicy
    cfg = default_config("pusht")
    stats = {"observation.state": {"mean": [0.0, 0.0], "std": [1.0, 1.0]}, "action": {"mean": [0.0, 0.0], "std": [1.0, 1.0]}}
    pol = VitaPolicy(cfg, stats)
    plan = pol.reduce_plan()
    n = sum(p.numel() for _, p in pol.hot_path_named_parameters())
    cover = torch.zeros(n, dtype=torch.int32)
    for lo, hi, ev in plan:
        cover[lo:hi] += 1
    assert bool((cover == 1).all())
    L_ = cfg.policy.flow_net.num_layers
    evs = [ev for _, _, ev in plan]
    assert evs == [0] + [3 + l for l in range(L_)] + [2, 1, 3 + L_]
    sizes = {ev: hi - lo for lo, hi, ev in plan}
    assert sizes[3 + L_] == 2 * (H_ := cfg.policy.flow_net.hidden_dim) * cfg.policy.action_ae.net.latent_dim + H_ + cfg.policy.action_ae.net.latent_dim + 2 * H_
    H = cfg.policy.flow_net.hidden_dim
    per_layer = 2 * H * 4 * H + 4 * H + H          # fc1 (4H x H) + fc2 (H x 4H) + their biases
    assert all(hi - lo == per_layer for lo, hi, ev in plan if 3 <= ev < 3 + L_)
