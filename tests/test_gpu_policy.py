"""GPU: the VitaPolicy / PolicyTrainer host mirror against the oracle — generate_actions, select_action's queue
semantics (vita_policy.py:210-220), compute_loss through torch autograd (loss.backward() as policy_trainer.py:82),
validate's metric keys, and whole training steps (normalize -> fused fwd+bwd -> clip -> AdamW -> EMA) against the
same step restated with the oracle + torch autograd on the CPU."""
import math

import pytest
import torch

from oracle import vita_oracle as O
from vita_amd import _lib as L
from vita_amd.config import default_config
from vita_amd.policy import VitaPolicy, ema_decay
from vita_amd.trainer import PolicyTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def small_cfg(encoder="cnn", matcher="conditional"):
    cfg = default_config("robomimic_square")          # O = 555, A = 7
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 48, "dec_hidden_dim": 48, "encoder_type": encoder,
                                    "num_layers": 4 if encoder == "cnn" else 2})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 3})
    cfg.policy.flow_matcher.name = matcher
    cfg.scheduler_name = "constant"
    cfg.optimizer_lr = 1e-3
    cfg.optimizer_eps = 1e-3      # keeps the Adam direction well-conditioned where a gradient is ~0 (plumbing test, see below)
    return cfg


def make_stats(cfg, g):
    s, a = cfg.task.state_dim, cfg.task.action_dim
    return {cfg.task.state_key: {"mean": torch.randn(s, generator=g).tolist(), "std": (torch.rand(s, generator=g) + 0.5).tolist()},
            cfg.task.action_key: {"mean": torch.randn(a, generator=g).tolist(), "std": (torch.rand(a, generator=g) + 0.5).tolist()}}


def make_policy(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    pol = VitaPolicy(cfg, make_stats(cfg, g))
    with torch.no_grad():     # biases / modulators away from their zero init so every term is exercised
        for k, p in pol.named_parameters():
            if p.requires_grad and k.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return pol


def make_batch(cfg, B, g, with_action=True):
    b = {cfg.task.state_key: torch.randn(B, 1, cfg.task.state_dim, generator=g),
         "observation.features": torch.relu(torch.randn(B, 1, 512, generator=g))}
    if with_action:
        b[cfg.task.action_key] = torch.randn(B, cfg.policy.pred_horizon, cfg.task.action_dim, generator=g)
    return b


def hot_sd(pol):
    return {k: p.detach().cpu().clone() for k, p in pol.hot_path_named_parameters()}


def oracle_features(pol, batch, cfg, normalize_targets=True):
    st = pol.stats
    sk, ak = cfg.task.state_key, cfg.task.action_key
    state = O.normalize(batch[sk], torch.tensor(st[sk]["mean"]), torch.tensor(st[sk]["std"]))
    feats = torch.cat([state.flatten(1), batch["observation.features"].flatten(1)], dim=1)
    acts = None
    if ak in batch:
        acts = batch[ak]
        if normalize_targets:
            acts = O.normalize(acts, torch.tensor(st[ak]["mean"]), torch.tensor(st[ak]["std"]))
    return feats, acts


@pytest.mark.parametrize("encoder", ["cnn", "simple"])
def test_generate_actions_and_select_action_queue(encoder):
    cfg = small_cfg(encoder)
    pol = make_policy(cfg).to(DEV)
    g = torch.Generator().manual_seed(1)
    B = 5
    sd = hot_sd(pol)
    T, A, st = cfg.policy.pred_horizon, cfg.task.action_dim, pol.stats
    calls = []
    orig = pol.generate_actions
    pol.generate_actions = lambda b: (calls.append(1), orig(b))[1]
    pol.reset()
    expected_queue, n_solves = [], 0
    for step in range(17):
        obs = make_batch(cfg, B, g, with_action=False)                 # env observations: no obs_horizon dim
        obs = {k: v[:, 0] for k, v in obs.items()}
        if step % cfg.policy.action_horizon == 0:                      # the queue is empty on calls 1, 9, 17
            feats, _ = oracle_features(pol, {k: v.unsqueeze(1) for k, v in obs.items()}, cfg)
            ref = O.generate_actions(sd, feats, 6, T, A)[:, :cfg.policy.action_horizon]
            ref = O.unnormalize(ref, torch.tensor(st["action"]["mean"]), torch.tensor(st["action"]["std"]))
            expected_queue = list(ref.transpose(0, 1))
            n_solves += 1
        a = pol.select_action({k: v.to(DEV) for k, v in obs.items()})
        assert a.shape == (B, A) and len(calls) == n_solves, f"call {step + 1}: {len(calls)} solves, expected {n_solves}"
        assert nerr(a, expected_queue.pop(0)) <= 1e-5
    assert n_solves == 3 and not pol.training


def test_compute_loss_autograd_bridge_matches_oracle_autograd():
    cfg = small_cfg()
    pol = make_policy(cfg).to(DEV)
    g = torch.Generator().manual_seed(2)
    B = 11
    batch = make_batch(cfg, B, g)
    t = torch.rand(B, generator=g)
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    pol.train()
    opt = pol.get_optimizer()
    opt.zero_grad()
    nb = pol.normalize_targets(pol.normalize_inputs(dbatch))
    loss, metrics = pol.compute_loss(nb, t=t.to(DEV))
    assert set(metrics) == {"loss", "flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss"}
    (loss * 2.0).backward()                                             # a non-trivial upstream factor
    feats, acts = oracle_features(pol, batch, cfg)
    sd = {k: v.double().requires_grad_(True) for k, v in hot_sd(pol).items()}
    rloss, rm = O.compute_loss(sd, feats.double(), acts.double(), t.double(), 6)
    (rloss * 2.0).backward()
    assert abs(loss.item() - rloss.item()) <= 1e-5 * abs(rloss.item())
    for k in ("flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss"):
        assert abs(metrics[k] - rm[k].item()) <= 1e-5 * abs(rm[k].item()), k
    for k, p in pol.hot_path_named_parameters():
        assert p.grad is not None and nerr(p.grad, sd[k].grad) <= 2e-5, k
    for k, p in pol.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k            # frozen statistics are left alone, as in the reference
    # no_grad: forward only, same value
    with torch.no_grad():
        l2, _ = pol.compute_loss(nb, t=t.to(DEV))
    assert abs(l2.item() - loss.item()) <= 1e-6 * abs(loss.item())


def test_exact_matcher_draws_a_valid_pairing_and_trains():
    cfg = small_cfg(matcher="exact")
    pol = make_policy(cfg).to(DEV)
    g = torch.Generator().manual_seed(3)
    batch = {k: v.to(DEV) for k, v in make_batch(cfg, 16, g).items()}
    pol.train()
    loss, metrics = pol.forward(batch)
    loss.backward()
    assert math.isfinite(loss.item()) and all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in pol.hot_path_named_parameters())


def test_validate_reports_the_reference_meter():
    cfg = small_cfg()
    pol = make_policy(cfg).to(DEV)
    g = torch.Generator().manual_seed(4)
    batch = {k: v.to(DEV) for k, v in make_batch(cfg, 8, g).items()}
    m = pol.validate(batch)
    for k in ("action_mse", "loss", "time_per_chunk_ms", "time_per_action_ms", "chunks_per_sec", "flow_loss"):
        assert k in m and math.isfinite(m[k]), k
    assert abs(m["chunks_per_sec"] - 1000 / m["time_per_chunk_ms"]) < 1e-6 * m["chunks_per_sec"]


def test_trainer_steps_match_the_reference_step_restated_on_cpu(tmp_path):
    """3 steps of train_step vs: normalize -> oracle.compute_loss -> autograd -> clip_grad_norm_(1.0) -> AdamW -> EMA.
    Adam's eps is raised to 1e-3 in this config so that an element whose gradient is ~0 does not turn fp32 rounding into a
    sign flip of its update; the fused kernel's own arithmetic is checked at the real eps in test_gpu_train.py."""
    cfg = small_cfg()
    pol = make_policy(cfg)
    ref = {k: v.double() for k, v in hot_sd(pol).items()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    ema = {k: v.clone() for k, v in ref.items()}
    tr = PolicyTrainer(cfg, pol, DEV, num_training_steps=100)
    assert tr.world == 1 and [e - s for s, e in pol.bucket_ranges] == [
        sum(p.numel() for k, p in pol.hot_path_named_parameters() if k.startswith(pre)) for pre in VitaPolicy.BUCKET_PREFIXES]
    g = torch.Generator().manual_seed(5)
    B = 24
    for step in range(1, 4):
        batch = make_batch(cfg, B, g)
        t = torch.rand(B, generator=g)
        tr.train_step({k: v.clone() for k, v in batch.items()}, t=t.to(DEV))
        got = tr.metrics()
        feats, acts = oracle_features(pol, batch, cfg)
        leaves = {k: v.clone().requires_grad_(True) for k, v in ref.items()}
        loss, _ = O.compute_loss(leaves, feats.double(), acts.double(), t.double(), 6)
        loss.backward()
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in leaves.values()))
        coef = min(1.0, cfg.train.grad_clip_norm / (gn.item() + 1e-6))
        for k in ref:
            O.adamw_step(ref[k], leaves[k].grad * coef, m[k], v2[k], step, cfg.optimizer_lr, *cfg.optimizer_betas, cfg.optimizer_eps,
                         cfg.optimizer_weight_decay)
            d = ema_decay(step)
            ema[k].sub_((1 - d) * (ema[k] - ref[k]))
        assert abs(got["loss"] - loss.item()) <= 1e-5 * abs(loss.item()), f"step {step}"
        assert abs(got["grad_norm"] - gn.item()) <= 1e-4 * gn.item(), f"step {step}"
    worst = max((nerr(p, ref[k]), k) for k, p in pol.hot_path_named_parameters())
    assert worst[0] <= 2e-5, worst
    es = tr.ema_state_dict()
    worst = max((nerr(es[k], ema[k]), k) for k in ema)
    assert worst[0] <= 2e-5, worst
    # checkpoint round trip: model.safetensors + training_state.pt (policy_trainer.py:261-274)
    tr.save_checkpoint(str(tmp_path))
    pol2 = make_policy(cfg, seed=9)
    tr2 = PolicyTrainer(cfg, pol2, DEV, num_training_steps=100)
    tr2.load_checkpoint(str(tmp_path))
    assert tr2.step_counter == 3 and torch.equal(tr2.flat, tr.flat) and torch.equal(tr2.exp_avg_sq, tr.exp_avg_sq) and torch.equal(tr2.ema, tr.ema)
    sd1, sd2 = pol.state_dict(), pol2.state_dict()
    assert set(sd1) == set(sd2) and all(torch.equal(sd1[k], sd2[k]) for k in sd1)


def test_training_state_is_the_reference_schema(tmp_path):
    """training_state.pt as policy_trainer.py:268-274 writes it: torch's own AdamW / LambdaLR load it, the EMA entry has
    diffusers' EMAModel keys; and a file written by the REFERENCE, whose parameter list also holds the ResNet observer's
    tensors between the Normalize statistics and obs_encoder (vita_policy.py:43-50,77), resumes here (aligned from the end)."""
    cfg = small_cfg()
    pol = make_policy(cfg)
    tr = PolicyTrainer(cfg, pol, DEV, num_training_steps=100)
    g = torch.Generator().manual_seed(6)
    batch = make_batch(cfg, 16, g)
    for _ in range(2):
        tr.train_step({k: v.clone() for k, v in batch.items()}, t=torch.rand(16, generator=g).to(DEV))
    tr.save_checkpoint(str(tmp_path))
    st = torch.load(str(tmp_path / "training_state.pt"), weights_only=True)
    assert set(st) == {"optimizer", "lr_scheduler", "ema", "step"} and st["step"] == 2
    # the reference's own loaders (policy_trainer.py:299-305): torch.optim.AdamW / LambdaLR .load_state_dict
    opt = pol.get_optimizer()
    opt.load_state_dict(st["optimizer"])
    sch = pol.get_scheduler(opt, 100)
    sch.load_state_dict(st["lr_scheduler"])
    assert sch.last_epoch == 2 and abs(opt.param_groups[0]["lr"] - tr.current_lr()) < 1e-12
    params = list(pol.parameters())
    n_norm = sum(1 for p in params if not p.requires_grad)
    assert len(st["optimizer"]["param_groups"][0]["params"]) == len(params)
    assert sorted(st["optimizer"]["state"]) == list(range(n_norm, len(params)))       # frozen statistics carry no state
    off = {id(p): o for o, p in zip(__import__("itertools").accumulate([0] + [q.numel() for _, q in pol.hot_path_named_parameters()]),
                                    [q for _, q in pol.hot_path_named_parameters()])}
    for i, p in enumerate(params):
        if p.requires_grad:
            ent = opt.state[p]
            assert float(ent["step"]) == 2.0 and ent["exp_avg"].shape == p.shape
            assert torch.equal(ent["exp_avg"].reshape(-1).cpu(), tr.exp_avg[off[id(p)]:off[id(p)] + p.numel()].cpu())
    ema = st["ema"]
    assert {"decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power", "shadow_params"} <= set(ema)
    assert ema["optimization_step"] == 2 and len(ema["shadow_params"]) == len(params)
    # a reference-written file: 3 extra (observer) parameters after the Normalize statistics, every index shifted
    ref_st = {"optimizer": {"state": {}, "param_groups": [dict(st["optimizer"]["param_groups"][0])]}, "lr_scheduler": st["lr_scheduler"],
              "ema": dict(ema), "step": 2}
    extra = [torch.randn(4, 3), torch.randn(7), torch.randn(2, 2, 2)]
    ref_st["optimizer"]["param_groups"][0]["params"] = list(range(len(params) + 3))
    for i, ent in st["optimizer"]["state"].items():
        ref_st["optimizer"]["state"][i + 3] = ent
    for j, e in enumerate(extra):
        ref_st["optimizer"]["state"][n_norm + j] = {"step": torch.tensor(2.0), "exp_avg": e, "exp_avg_sq": e * e}
    ref_st["ema"]["shadow_params"] = ema["shadow_params"][:n_norm] + extra + ema["shadow_params"][n_norm:]
    pol2 = make_policy(cfg, seed=9)
    tr2 = PolicyTrainer(cfg, pol2, DEV, num_training_steps=100)
    tr2.load_training_state(ref_st)
    assert tr2.step_counter == 2 and torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.exp_avg_sq, tr.exp_avg_sq) and torch.equal(tr2.ema, tr.ema)
    # a file whose moments do not fit is refused before anything is written
    bad = {"optimizer": {"state": {k: dict(v) for k, v in st["optimizer"]["state"].items()}, "param_groups": st["optimizer"]["param_groups"]},
           "lr_scheduler": None, "ema": None, "step": 1}
    first = min(bad["optimizer"]["state"])
    bad["optimizer"]["state"][first]["exp_avg"] = torch.zeros(3)
    before = tr2.exp_avg.clone()
    with pytest.raises(L.VitaHipError):
        tr2.load_training_state(bad)
    assert torch.equal(tr2.exp_avg, before) and tr2.step_counter == 2


def test_bf16_weight_cache_follows_the_native_optimizer():
    """The fused optimizer kernel writes the parameters through raw pointers (no torch version bump): generate_actions with
    gemm_dtype = bf16 after PolicyTrainer.train_step must use bf16 copies of the UPDATED weights."""
    cfg = small_cfg()
    cfg.policy.gemm_dtype = "bf16"
    cfg.optimizer_lr = 5e-2            # a step large enough that stale weights miss the tolerance by far
    cfg.scheduler_name = "constant"
    pol = make_policy(cfg)
    tr = PolicyTrainer(cfg, pol, DEV, num_training_steps=100)
    g = torch.Generator().manual_seed(8)
    batch = make_batch(cfg, 16, g)
    nb = pol.normalize_inputs({k: v.to(DEV) for k, v in batch.items()})
    pol.eval()
    a0 = pol.generate_actions(nb).clone()          # packs the bf16 copies of the initial weights
    for _ in range(3):
        tr.train_step({k: v.clone() for k, v in batch.items()}, t=torch.rand(16, generator=g).to(DEV))
    pol.eval()
    a1 = pol.generate_actions(nb)
    feats, _ = oracle_features(pol, batch, cfg)
    ref = O.generate_actions(hot_sd(pol), feats, cfg.policy.flow_matcher.num_sampling_steps, cfg.policy.pred_horizon, cfg.task.action_dim)
    assert nerr(a0.reshape(ref.shape), ref) > 0.1, "the step did not move the weights enough for this test to mean anything"
    assert nerr(a1.reshape(ref.shape), ref) <= 2e-2


def test_freeze_flags_and_contrastive_terms_at_policy_level():
    """action_ae.freeze_encoder (vita_policy.py:119,153) with the contrastive weights on: metric keys as the reference emits
    them, frozen module without gradient under the autograd bridge, and untouched bit-for-bit by the native trainer."""
    cfg = small_cfg()
    cfg.policy.action_ae.freeze_encoder = True
    cfg.policy.vita.enc_contrastive_weight = 0.25
    cfg.policy.vita.flow_contrastive_weight = 0.25          # inactive: the sampler branch is gated off by the freeze
    pol = make_policy(cfg).to(DEV)
    g = torch.Generator().manual_seed(12)
    B = 13
    batch = make_batch(cfg, B, g)
    t = torch.rand(B, generator=g)
    nb = pol.normalize_targets(pol.normalize_inputs({k: v.to(DEV) for k, v in batch.items()}))
    pol.train()
    loss, metrics = pol.compute_loss(nb, t=t.to(DEV))
    assert set(metrics) == {"loss", "flow_loss", "enc_contrastive_loss", "enc_action_recon_loss"}
    loss.backward()
    feats, acts = oracle_features(pol, batch, cfg)
    sd = {k: v.double().requires_grad_(True) for k, v in hot_sd(pol).items()}
    rloss, rm = O.compute_loss(sd, feats.double(), acts.double(), t.double(), 6, freeze_encoder=True, enc_contrastive_weight=0.25,
                               flow_contrastive_weight=0.25)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-5 * abs(rloss.item())
    assert abs(metrics["enc_contrastive_loss"] - rm["enc_contrastive_loss"].item()) <= 1e-5 * rm["enc_contrastive_loss"].item()
    for k, p in pol.hot_path_named_parameters():
        if k.startswith("action_encoder."):
            assert p.grad is None and sd[k].grad is None, k
        else:
            assert nerr(p.grad, sd[k].grad) <= 2e-5, k
    # native trainer: the frozen encoder is skipped by the fused optimizer (no weight decay, no moment update)
    pol2 = make_policy(cfg)
    tr = PolicyTrainer(cfg, pol2, DEV, num_training_steps=10)
    before = {k: p.detach().clone() for k, p in pol2.hot_path_named_parameters()}
    tr.train_step({k: v.clone() for k, v in batch.items()}, t=t.to(DEV))
    m = tr.metrics()
    assert abs(m["loss"] - rloss.item()) <= 1e-5 * abs(rloss.item()) and m["flow_contrastive_loss"] == 0.0
    for k, p in pol2.hot_path_named_parameters():
        same = torch.equal(p.detach(), before[k])
        assert same == k.startswith("action_encoder."), k


@pytest.mark.parametrize("matcher", ["mean", "consistency"])
def test_alternate_matchers_generate_actions(matcher):
    """policy.flow_matcher.name = mean / consistency (vita_policy.py:53,84-97): generate_actions runs obs_encoder -> the
    matcher's sampler -> decoder on the HIP path and matches the oracle (MeanFlow training: tests/test_gpu_meanflow_train.py)."""
    cfg = small_cfg(matcher=matcher)
    if matcher == "mean":
        cfg.policy.flow_net.name = "simple_mean_flow_net"
        cfg.policy.flow_matcher.num_sampling_steps = 1
    else:
        cfg.policy.flow_matcher.pop("sigma")     # vita.yaml's torchcfm key; the Consistency matcher's constructor rejects it
        cfg.policy.flow_matcher.merge({"num_sampling_steps": 3, "sigma_var": 0.0})
    pol = make_policy(cfg)
    with torch.no_grad():      # the reference zero-initialises the MeanFlow heads: give them values
        for k, p in pol.named_parameters():
            if "output_layer" in k:
                p.copy_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(3)))
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(8)
    B = 9
    batch = make_batch(cfg, B, g, with_action=False)
    nb = pol.normalize_inputs({k: v.to(DEV) for k, v in batch.items()})
    acts = pol.generate_actions(nb)
    sd = hot_sd(pol)
    feats, _ = oracle_features(pol, batch, cfg)
    z0 = O.obs_encode(sd, feats)
    z = O.mean_flow_sample(sd, z0, 1) if matcher == "mean" else O.consistency_sample(sd, z0, 3, sigma_var=0.0)
    ref = O.simple_action_decode(sd, z, cfg.policy.pred_horizon, cfg.task.action_dim)
    assert acts.shape == ref.shape and nerr(acts, ref) <= 1e-5
    if matcher == "consistency":    # its training loss is not built (and cannot run on 2-D latents in the reference either)
        with pytest.raises(NotImplementedError):
            pol.compute_loss(nb)
    if matcher == "consistency":   # default sigma_var = 1.0: stochastic sampler, finite and different from the ODE branch
        cfg2 = small_cfg(matcher=matcher)
        cfg2.policy.flow_matcher.pop("sigma")
        pol2 = make_policy(cfg2).to(DEV)
        a2 = pol2.generate_actions(nb)
        assert torch.isfinite(a2).all()


def test_full_size_training_step_properties_b4096():
    """hook_package dims at B = 4096 per GPU (the large-batch data-parallel regime): finite, deterministic, and the
    gradient of a batch equals the mean of the gradients of its two halves (what the all-reduce relies on)."""
    from vita_amd import ops
    B = 4096
    sd = {k: v.to(DEV) for k, v in O.make_state_dict(533, 21, seed=0).items()}
    g = torch.Generator().manual_seed(6)
    obs, acts, t = torch.randn(B, 533, generator=g).to(DEV), torch.randn(B, 16, 21, generator=g).to(DEV), torch.rand(B, generator=g).to(DEV)
    pw = ops.PolicyHandle(sd, DEV, strict=True)
    cfgl = ops.loss_config(6)

    def run(sl, cfg=cfgl):
        gsd = {k: torch.empty_like(v) for k, v in sd.items()}
        gw = ops.PolicyHandle(gsd, DEV, strict=True)
        losses, _ = ops.compute_loss(pw, cfg, obs[sl], acts[sl], t[sl], grads=gw)
        return losses.clone(), gsd
    l_all, g_all = run(slice(0, B))
    l_again, g_again = run(slice(0, B))
    assert torch.isfinite(l_all[:5]).all() and torch.equal(l_all, l_again)
    assert all(torch.equal(g_all[k], g_again[k]) for k in sd)
    # additivity, with the smooth (l2) reconstruction loss: half batches select other tile shapes (another k-summation
    # order -> last-bit differences in the forward), and at a kink a last-bit difference flips a derivative: under l1 one
    # sign(d) flip at d ~ 0 moves a 1.4 M-term random-sign sum by ~1e-3.  The conv encoder's ReLUs have the same kink
    # (one flipped mask bit among 16.7 M pre-activations changes one of the 32768 terms of a weight-gradient row), so the
    # conv weights are held to 5e-3 and everything behind smooth activations to 2e-5.
    cfg2 = ops.loss_config(6, recon="l2")
    l_all, g_all = run(slice(0, B), cfg2)
    l_a, g_a = run(slice(0, B // 2), cfg2)
    l_b, g_b = run(slice(B // 2, B), cfg2)
    assert abs((l_a[0] + l_b[0]).item() / 2 - l_all[0].item()) <= 1e-5 * abs(l_all[0].item())
    worst = max((nerr((g_a[k] + g_b[k]) * 0.5, g_all[k]), k) for k in sd if not k.startswith("action_encoder.encoder."))
    assert worst[0] <= 2e-5, worst
    worst = max((nerr((g_a[k] + g_b[k]) * 0.5, g_all[k]), k) for k in sd if k.startswith("action_encoder.encoder."))
    assert worst[0] <= 5e-3, worst


def test_forked_child_of_a_gpu_process_fails_loudly():
    """HIP cannot be used in a fork()ed child of a process that initialised it; calls through the ctypes binding used to hang
    there (round 1).  The binding now marks such a child and raises before any HIP call."""
    import os
    torch.zeros(1, device=DEV)          # this process holds a HIP context
    L.load()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                        # child: no GPU call may happen here
        code = 1
        try:
            from vita_amd import ops
            try:
                ops.sumsq  # noqa: B018
                L.load()
            except L.VitaHipError as e:
                code = 0 if "fork" in str(e) else 2
        finally:
            os.write(w, bytes([code]))
            os._exit(0)
    os.close(w)
    got = os.read(r, 1)
    os.waitpid(pid, 0)
    os.close(r)
    assert got == b"\x00", got
