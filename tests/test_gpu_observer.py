"""VitaPolicy(observer="resnet18") (vita_policy.py:43-50): camera frames in; the ResNet18 observer runs as a torch module in front of
the fused step and trains through d loss / d features (vita_compute_loss_f32's d_obs_features).
The chain rule across the boundary is checked against the same policy fed the observer's features as a leaf; the losses against the
fp64 oracle on those features.  The backbone's arithmetic is torch's (MIOpen): the observer's own logic is pinned on the CPU
(tests/test_observers_cpu.py); the ResNet18 definition is UNPINNED (torchvision absent)."""
import pytest
import torch

from oracle import vita_oracle as O
from vita_amd.config import default_config
from vita_amd.observers import ResNetObserver
from vita_amd.policy import VitaPolicy

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cfg():
    cfg = default_config("pusht")        # one 96 x 96 camera, crop 84 x 84; state 2, action 2
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 48, "dec_hidden_dim": 48, "num_layers": 4})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 2})
    cfg.policy.flow_matcher.name = "conditional"
    return cfg


def _stats(cfg):
    s, a = cfg.task.state_dim, cfg.task.action_dim
    return {cfg.task.state_key: {"mean": [0.1] * s, "std": [1.5] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [2.0] * a},
            cfg.task.image_keys[0]: {"mean": [[[0.485]], [[0.456]], [[0.406]]], "std": [[[0.229]], [[0.224]], [[0.225]]]}}


def _batch(cfg, B, seed=0, action=True):
    g = torch.Generator().manual_seed(seed)
    b = {cfg.task.state_key: torch.randn(B, 1, cfg.task.state_dim, generator=g).to(DEV),
         cfg.task.image_keys[0]: torch.rand(B, 1, 3, 96, 96, generator=g).to(DEV)}
    if action:
        b[cfg.task.action_key] = torch.randn(B, 16, cfg.task.action_dim, generator=g).to(DEV)
    return b


def _policy(cfg):
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, _stats(cfg), observer="resnet18").to(DEV)
    assert isinstance(pol.observer, ResNetObserver) and pol.observer.crop_shape == (84, 84) and pol.observer.resize_shape == (96, 96)
    return pol


class _Fixed(torch.nn.Module):
    def __init__(self, feats):
        super().__init__()
        self.feats = feats

    def forward(self, batch):
        return self.feats


def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_state_dict_has_the_references_observer_keys():
    pol = _policy(_cfg())
    keys = set(pol.state_dict())
    for k in ("observer.backbone.conv1.weight", "observer.backbone.bn1.running_var", "observer.backbone.layer4.1.bn2.bias",
              "observer.backbone.layer3.0.downsample.0.weight", "obs_encoder.weight"):
        assert k in keys, k
    # AdamW over policy.parameters() (vita_policy.py:222-229) holds the 20 convolutions
    opt = pol.get_optimizer()
    ids = {id(p) for grp in opt.param_groups for p in grp["params"]}
    assert all(id(p) in ids for p in pol.observer.parameters()) and sum(1 for _ in pol.observer.parameters()) == 20


def test_train_step_through_the_backbone():
    cfg = _cfg()
    pol = _policy(cfg)
    pol.train()
    B = 12
    batch = _batch(cfg, B)
    t = torch.rand(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    nb = pol.normalize_targets(pol.normalize_inputs(batch))
    torch.manual_seed(21)                                    # the crop draw
    loss, metrics = pol.compute_loss(nb, t=t)
    loss.backward()
    got = {k: p.grad.clone() for k, p in pol.observer.named_parameters()}
    assert all(g.abs().max() > 0 for g in got.values())
    # the same features as a leaf of the same fused step: d loss / d features, then torch's backward of the observer alone
    for p in pol.parameters():
        p.grad = None
    torch.manual_seed(21)
    feats = pol.observer(nb)
    assert feats.shape == (B, 2 + 512) and torch.equal(feats[:, :2], nb[cfg.task.state_key].flatten(1))
    leaf = feats.detach().clone().requires_grad_(True)
    obs, pol.observer = pol.observer, _Fixed(leaf)
    try:
        loss2, _ = pol.compute_loss(nb, t=t)
        loss2.backward()
    finally:
        pol.observer = obs
    assert abs(loss2.item() - loss.item()) <= 1e-6 * abs(loss.item())
    feats.backward(leaf.grad)
    for k, p in pol.observer.named_parameters():
        assert nerr(got[k], p.grad) <= 1e-4, k                # MIOpen's weight-gradient reductions are not run-to-run bitwise
    # the losses and d loss / d features against the fp64 oracle on those features
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in pol.state_dict().items()
          if k.startswith(("obs_encoder.", "flow_net.", "action_encoder.", "action_decoder."))}
    f64 = feats.detach().cpu().double().requires_grad_(True)
    rloss, rm = O.compute_loss(sd, f64, nb[cfg.task.action_key].cpu().double(), t.cpu().double(), 6)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-5 * abs(rloss.item())
    for k in ("flow_loss", "consistency_loss", "flow_action_recon_loss", "enc_action_recon_loss"):
        assert abs(metrics[k] - rm[k].item()) <= 1e-5 * abs(rm[k].item()), k
    assert nerr(leaf.grad, f64.grad) <= 2e-5


def test_native_trainer_trains_the_observer():
    """The native trainer (flat buffers, fused clip + AdamW + EMA, the data-parallel path) with a trainable backbone, as the reference
    trains it in the same AdamW group (vita_policy.py:113-114, 222-229): loss_and_grads hands d loss / d features to torch's backward
    of the observer alone and the backbone's gradients land in the observer's range of the flat gradient buffer.  Checked against
    the drop-in path (compute_loss(...).backward()) on the same crop and the same t; then whole steps against the same step made of
    torch's clip_grad_norm_ + AdamW on a twin policy."""
    from vita_amd.trainer import PolicyTrainer
    cfg = _cfg()
    cfg.scheduler_name = "constant"
    cfg.optimizer_lr = 1e-3
    cfg.optimizer_eps = 1e-3      # keeps the Adam direction well-conditioned where a gradient is ~0 (MIOpen's reductions are not bitwise)
    cfg.use_ema = True
    B = 12
    batch = _batch(cfg, B)
    t = torch.rand(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    twin = _policy(cfg)                                       # same seed: same weights
    twin.train()
    pol = _policy(cfg)
    tr = PolicyTrainer(cfg, pol, DEV, num_training_steps=100)
    assert pol.observer_range is not None and pol.observer_range[1] == tr.flat.numel()
    n_obs = sum(p.numel() for p in pol.observer.parameters() if p.requires_grad)
    assert pol.observer_range[1] - pol.observer_range[0] == n_obs and tr.update_ranges[-1][1] == tr.flat.numel()
    assert tr.plan[-1][:2] == pol.observer_range
    # gradients of one call against the drop-in path
    pol.train()
    nb = pol.normalize_targets(pol.normalize_inputs(batch))
    torch.manual_seed(21)
    pol.loss_and_grads(nb, t=t)
    torch.manual_seed(21)
    loss, _ = twin.compute_loss(twin.normalize_targets(twin.normalize_inputs(batch)), t=t)
    loss.backward()
    for (k, p), (_, q) in zip(pol.named_parameters(), twin.named_parameters()):
        if q.grad is not None:
            assert nerr(p.grad, q.grad) <= 1e-4, k            # MIOpen's weight-gradient reductions are not run-to-run bitwise
    # two whole steps: normalise -> fused step + observer backward -> clip -> AdamW (+ EMA) on the flat buffers
    opt = torch.optim.AdamW(twin.parameters(), lr=1e-3, betas=tuple(cfg.optimizer_betas), eps=cfg.optimizer_eps,
                            weight_decay=cfg.optimizer_weight_decay)
    w0 = pol.observer.backbone.conv1.weight.detach().clone()
    for step in range(2):
        torch.manual_seed(30 + step)
        tr.train_step(batch, t=t)
        opt.zero_grad()
        torch.manual_seed(30 + step)
        loss, _ = twin.compute_loss(twin.normalize_targets(twin.normalize_inputs(batch)), t=t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), cfg.train.grad_clip_norm)
        opt.step()
    assert not torch.equal(w0, pol.observer.backbone.conv1.weight)
    m = tr.metrics()
    assert abs(m["loss"] - loss.item()) <= 1e-4 * abs(loss.item())
    for (k, p), (_, q) in zip(pol.named_parameters(), twin.named_parameters()):
        if q.requires_grad:
            assert nerr(p, q) <= 2e-4, k
    # the checkpoint covers the backbone's moments and EMA copies too
    st = tr.training_state()
    assert len(st["optimizer"]["state"]) == sum(1 for p in pol.parameters() if p.requires_grad)
    assert len(st["ema"]["shadow_params"]) == len(list(pol.parameters()))


def test_generate_and_select_action_from_frames():
    cfg = _cfg()
    pol = _policy(cfg)
    pol.eval()
    B = 5
    batch = _batch(cfg, B, seed=2, action=False)
    nb = pol.normalize_inputs(batch)
    torch.manual_seed(4)
    with torch.no_grad():
        feats = pol.observer(nb)
    torch.manual_seed(4)
    acts = pol.generate_actions(nb)
    assert acts.shape == (B, 16, 2) and torch.isfinite(acts).all()
    sd = {k: v.detach().cpu().double() for k, v in pol.state_dict().items() if k.startswith(("obs_encoder.", "flow_net.", "action_decoder."))}
    want = O.generate_actions(sd, feats.cpu().double(), 6, 16, 2)
    assert nerr(acts, want) <= 2e-5
    pol.reset()
    a = pol.select_action({k: v[:, 0] for k, v in batch.items()})
    assert a.shape == (B, 2) and torch.isfinite(a).all()


# ---- the backbone's elementwise tail: vita_channel_affine_relu_f32 / _bwd_f32 --------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 5, 7, 10), (2, 8, 4, 6), (4, 64, 56, 77), (1, 3, 1, 1), (0, 4, 2, 2)])
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_channel_affine_relu_is_bitwise_the_torch_ops(shape, with_res, relu):
    from vita_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + 2 * with_res + relu)
    x = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
    res = torch.randn(shape, generator=g).to(DEV).requires_grad_(True) if with_res else None
    scale = (torch.rand(shape[1], generator=g) + 0.5).to(DEV)
    shift = torch.randn(shape[1], generator=g).to(DEV)
    dy = torch.randn(shape, generator=g).to(DEV)
    y = ops.channel_affine_relu(x, scale, shift, res, relu)
    y.backward(dy)
    got = (y.detach().clone(), x.grad.clone(), res.grad.clone() if with_res else None)
    x.grad = None
    if with_res:
        res.grad = None
    want = x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    if with_res:
        want = want + res
    if relu:
        want = torch.relu(want)
    want.backward(dy)
    assert torch.equal(got[0], want.detach()) and torch.equal(got[1], x.grad)
    if with_res:
        assert torch.equal(got[2], res.grad)


def test_channel_affine_relu_propagates_nan_and_rejects_bad_shapes():
    from vita_amd import ops
    from vita_amd._lib import VitaHipError
    x = torch.tensor([[[[float("nan"), -1.0, 2.0, -0.5]]]], device=DEV)
    y = ops.channel_affine_relu(x, torch.ones(1, device=DEV), torch.zeros(1, device=DEV), None, True)
    assert torch.isnan(y[0, 0, 0, 0]) and y[0, 0, 0, 1:].tolist() == [0.0, 2.0, 0.0]
    with pytest.raises(VitaHipError):
        ops.channel_affine_relu(torch.zeros(2, 3, 4, 4, device=DEV), torch.ones(4, device=DEV), torch.zeros(4, device=DEV))
    with pytest.raises(VitaHipError):
        ops.channel_affine_relu(torch.zeros(2, 3, 4, 4, device=DEV), torch.ones(3, device=DEV), torch.zeros(3, device=DEV),
                                torch.zeros(2, 3, 4, 5, device=DEV))


def test_backbone_with_the_fused_tail_equals_the_torch_ops(monkeypatch):
    from observer_fixture import seeded_backbone
    bb = seeded_backbone().to(DEV)
    x = torch.rand(6, 3, 84, 84, generator=torch.Generator().manual_seed(0)).to(DEV)
    dy = None
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VITA_OBSERVER_TORCH_EPILOGUE", mode)
        for p in bb.parameters():
            p.grad = None
        y = bb(x)["feature_map"]
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(DEV) if dy is None else dy
        y.backward(dy)
        outs[mode] = (y.detach().clone(), {k: p.grad.clone() for k, p in bb.named_parameters()})
    # each fused pass is bitwise the torch ops (above); across the network MIOpen may pick another convolution solver per call
    assert nerr(outs["0"][0], outs["1"][0]) <= 1e-5
    for k in outs["0"][1]:
        assert nerr(outs["0"][1][k], outs["1"][1][k]) <= 1e-4, k      # MIOpen's weight-gradient reductions are not run-to-run bitwise


@pytest.mark.parametrize("shape", [(2, 3, 9, 11), (3, 8, 42, 42), (1, 64, 112, 154), (2, 4, 1, 1), (2, 5, 2, 7), (0, 4, 6, 6),
                                   (2, 16, 112, 112), (3, 2, 7, 4), (1, 3, 5, 8)])      # widths divisible by four: the four-column backward
def test_stem_tail_is_bitwise_norm_relu_maxpool(shape):
    from vita_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(DEV).requires_grad_(True)
    scale = (torch.rand(shape[1], generator=g) + 0.5).to(DEV)
    shift = (0.3 * torch.randn(shape[1], generator=g)).to(DEV)
    y = ops.affine_relu_maxpool(x, scale, shift)
    dy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(dy)
    got = (y.detach().clone(), x.grad.clone())
    x.grad = None
    want = F.max_pool2d(torch.relu(x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)), 3, 2, 1)
    want.backward(dy)
    assert got[0].shape == want.shape and torch.equal(got[0], want.detach())
    assert torch.equal(got[1], x.grad)


def test_stem_tail_splits_more_planes_than_one_launch_takes(monkeypatch):
    from vita_amd import ops
    import torch.nn.functional as F
    monkeypatch.setattr(ops._AffineReLUMaxPool, "MAX_PLANES", 7)        # 3 channels: two images per launch
    x = torch.randn(5, 3, 8, 10, generator=torch.Generator().manual_seed(0)).to(DEV).requires_grad_(True)
    scale, shift = torch.full((3,), 1.25, device=DEV), torch.tensor([0.1, -0.2, 0.0], device=DEV)
    y = ops.affine_relu_maxpool(x, scale, shift)
    y.sum().backward()
    got = x.grad.clone()
    x.grad = None
    want = F.max_pool2d(torch.relu(x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)), 3, 2, 1)
    want.sum().backward()
    assert torch.equal(y.detach(), want.detach()) and torch.equal(got, x.grad)


@pytest.mark.parametrize("W", [12, 13])
def test_stem_tail_nan_and_nonpositive_windows_follow_torch(W):
    from vita_amd import ops
    import torch.nn.functional as F
    x = torch.randn(2, 3, 10, W, generator=torch.Generator().manual_seed(W)).to(DEV)
    x[0, 0, 5, 5] = float("nan")
    x[1, 1] = -x[1, 1].abs()             # a plane whose every window has maximum 0 after the ReLU: no gradient anywhere
    x.requires_grad_(True)
    scale, shift = torch.tensor([1.5, 0.75, 1.0], device=DEV), torch.zeros(3, device=DEV)
    y = ops.affine_relu_maxpool(x, scale, shift)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    y.backward(dy)
    got = (y.detach().clone(), x.grad.clone())
    x.grad = None
    want = F.max_pool2d(torch.relu(x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)), 3, 2, 1)
    want.backward(dy)
    assert torch.isnan(want).any() and float(x.grad[1, 1].abs().max()) == 0.0
    assert torch.equal(torch.nan_to_num(got[0], nan=77.0), torch.nan_to_num(want.detach(), nan=77.0))
    assert torch.equal(torch.nan_to_num(got[1], nan=77.0), torch.nan_to_num(x.grad, nan=77.0))
