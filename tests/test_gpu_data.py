"""Dataset -> device batches on a replay buffer resident in HBM (vita_gather_window_f32 / vita_gather_window_u8_nchw_f32 through
vita_amd.data.DeviceReplayDataset): bit-exact against the reference's own AVAlohaDataset.__getitem__ + default_collate
(tests/golden/dataset.safetensors), against the oracle on a larger ragged buffer, the loader's batching, and a batch going
straight into VitaPolicy.compute_loss."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle import data_oracle as DO
from vita_amd.config import default_config
from vita_amd.data import DeviceBatchLoader, DeviceReplayDataset, EpisodeAwareSampler, policy_delta_timestamps
from vita_amd.policy import VitaPolicy

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _check(batch, want):
    assert set(batch) == set(want), set(batch) ^ set(want)
    for k, v in want.items():
        got = batch[k].cpu().numpy()
        v = v.astype(bool) if got.dtype == bool else v
        assert got.shape == v.shape and got.dtype == v.dtype and np.array_equal(got, v), k


def test_device_batches_are_the_reference_items_bit_for_bit(golden_dir):
    g = {}
    with safe_open(os.path.join(golden_dir, "dataset.safetensors"), framework="pt") as f:
        for k in f.keys():
            g[k] = f.get_tensor(k)
    meta = json.load(open(os.path.join(golden_dir, "dataset.json")))
    arrays = {k[len("buffer/"):]: v.numpy() for k, v in g.items() if k.startswith("buffer/")}
    fps = 10
    ds = DeviceReplayDataset(arrays, meta["lengths"], {k: [d / fps for d in v] for k, v in meta["delta_indices"].items()}, fps,
                             image_keys=meta["image_keys"], device=DEV)
    assert ds.delta_indices == meta["delta_indices"] and len(ds) == sum(meta["lengths"])
    batch = ds.batch(g["batch/indices"])
    _check(batch, {k[len("batch/out/"):]: v.numpy() for k, v in g.items() if k.startswith("batch/out/")})
    assert batch["observation.images.cam"].dtype == torch.float32 and batch["action_is_pad"].dtype == torch.bool


def test_item_extras_and_index_bounds(golden_dir):
    """image_transforms, 'task_index' and the 'task' strings of AVAlohaDataset.__getitem__ (av_aloha_dataset.py:264-266, 362-370)
    against the reference's own items; and the index bounds of numpy / zarr indexing: -1 is the last frame, N raises."""
    g = {}
    with safe_open(os.path.join(golden_dir, "dataset.safetensors"), framework="pt") as f:
        for k in f.keys():
            g[k] = f.get_tensor(k)
    meta = json.load(open(os.path.join(golden_dir, "dataset.json")))
    arrays = {k[len("buffer/"):]: v.numpy() for k, v in g.items() if k.startswith("buffer/")}
    arrays["task_index"] = g["extras/task_index"].numpy()
    fps = 10
    tasks = {int(k): v for k, v in meta["extras/tasks"].items()}
    ds = DeviceReplayDataset(arrays, meta["lengths"], {k: [d / fps for d in v] for k, v in meta["delta_indices"].items()}, fps,
                             image_keys=meta["image_keys"], device=DEV, image_transforms=lambda x: x.flip(-1) * 0.5 + 0.25, tasks=tasks)
    batch = ds.batch(g["batch/indices"])
    assert batch.pop("task") == meta["extras/task"]
    _check(batch, {k[len("extras/out/"):]: v.numpy() for k, v in g.items() if k.startswith("extras/out/")})
    N = len(ds)
    last, wrapped = ds.batch([N - 1]), ds.batch([-1])
    for k in last:
        assert last[k] == wrapped[k] if k == "task" else torch.equal(last[k], wrapped[k]), k
    with pytest.raises(IndexError):
        ds.batch([0, N])
    with pytest.raises(IndexError):
        ds.batch([-N - 1])
    # device-side index tensors: the kernels keep the reads inside the buffer and report through the error word
    ok = ds.batch(torch.tensor([-1, 3], device=DEV), check_indices=True)
    assert torch.equal(ok["action"][0], last["action"][0])
    with pytest.raises(IndexError):
        ds.batch(torch.tensor([1, N + 5], device=DEV), check_indices=True)
    unchecked = ds.batch(torch.tensor([5, N + 5], device=DEV))        # no fault, the row is flagged as padding
    assert bool(unchecked["observation.state_is_pad"][1].all()) and not bool(unchecked["observation.state_is_pad"][0].any())
    ds._err.zero_()


def test_large_ragged_buffer_against_the_oracle():
    rng = np.random.default_rng(3)
    lengths = rng.integers(1, 60, size=70).tolist() + [1, 2, 300]
    N = sum(lengths)
    arrays = {"episode_index": np.repeat(np.arange(len(lengths)), lengths).astype(np.int64),
              "observation.state": rng.standard_normal((N, 21)).astype(np.float32),
              "observation.features": rng.standard_normal((N, 1024)).astype(np.float32),     # precomputed camera latents
              "action": rng.standard_normal((N, 21)).astype(np.float32),
              "cam": rng.integers(0, 256, size=(N, 20, 28, 3), dtype=np.uint8),
              "depth": rng.integers(0, 256, size=(N, 9, 7, 1), dtype=np.uint8)}
    delta = {"observation.state": [-1, 0], "observation.features": [-1, 0], "action": list(range(-1, 15)), "cam": [-3, 0, 2], "depth": [0]}
    fps = 25
    ds = DeviceReplayDataset(arrays, lengths, {k: [d / fps for d in v] for k, v in delta.items()}, fps, image_keys=("cam", "depth"), device=DEV)
    idx = rng.integers(0, N, size=257)
    idx[:4] = [0, N - 1, lengths[0] - 1, lengths[0]]
    ends = np.cumsum(lengths)
    want = DO.batch(arrays, np.concatenate([[0], ends[:-1]]), ends, delta, ("cam", "depth"), idx.tolist())
    _check(ds.batch(torch.from_numpy(idx)), want)
    assert ds.batch([]).get("action").shape == (0, 16, 21)


def test_loader_batches_follow_the_sampler():
    lengths = [12, 30, 7]
    N = sum(lengths)
    arrays = {"s": np.arange(N, dtype=np.float32)[:, None].repeat(3, 1), "a": np.arange(N, dtype=np.float32)[:, None]}
    ds = DeviceReplayDataset(arrays, lengths, {"s": [0.0], "a": [0.0, 0.1]}, 10, device=DEV)
    torch.manual_seed(5)
    loader = DeviceBatchLoader(ds, batch_size=8, drop_n_last_frames=4, is_training=True)
    got = [b["s"][:, 0, 0].long().cpu().tolist() for b in loader]
    torch.manual_seed(5)
    order = list(EpisodeAwareSampler(ds.episode_data_index, drop_n_last_frames=4, shuffle=True))
    assert [i for b in got for i in b] == order and len(got) == len(loader) == -(-len(order) // 8)
    assert all(len(b) == 8 for b in got[:-1]) and len(got[-1]) == len(order) - 8 * (len(got) - 1)        # drop_last=False
    ev = DeviceBatchLoader(ds, batch_size=16, drop_n_last_frames=0, is_training=False)
    assert [i for b in ev for i in b["s"][:, 0, 0].long().cpu().tolist()] == list(range(N))


def test_a_device_batch_trains_the_policy():
    cfg = default_config("pusht")
    S, A = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats).to(DEV)
    rng = np.random.default_rng(0)
    lengths = [40, 55, 33]
    N = sum(lengths)
    arrays = {cfg.task.state_key: rng.standard_normal((N, S)).astype(np.float32),
              "observation.features": np.maximum(rng.standard_normal((N, 512 * len(cfg.task.image_keys))), 0).astype(np.float32),
              cfg.task.action_key: rng.standard_normal((N, A)).astype(np.float32)}
    dts = {k: v for k, v in policy_delta_timestamps(pol, cfg.task).items() if k in arrays}
    dts["observation.features"] = dts[cfg.task.state_key]
    ds = DeviceReplayDataset(arrays, lengths, dts, cfg.task.fps, device=DEV)
    drop = cfg.policy.pred_horizon - cfg.policy.action_horizon - cfg.policy.obs_horizon + 1      # dataset_utils.py:107-110
    loader = DeviceBatchLoader(ds, batch_size=32, drop_n_last_frames=drop, is_training=True)
    batch = next(iter(loader))
    assert tuple(batch[cfg.task.action_key].shape) == (32, cfg.policy.pred_horizon, A)
    pol.train()
    loss, metrics = pol.forward(batch)
    loss.backward()
    assert torch.isfinite(loss) and pol.obs_encoder.weight.grad is not None and "flow_loss" in metrics


def test_frames_from_the_replay_buffer_train_the_policy_through_the_observer():
    """uint8 frames resident in HBM -> one gather launch (float32 / 255, channels first) -> normalise -> ResNet18 observer -> the fused
    step; three optimizer steps of the drop-in trainer loop (policy_trainer.py:76-94) move the backbone and lower the loss."""
    cfg = default_config("pusht")
    cfg.policy.flow_matcher.name = "conditional"
    S, A, cam = cfg.task.state_dim, cfg.task.action_dim, cfg.task.image_keys[0]
    stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A},
             cam: {"mean": [[[0.5]], [[0.5]], [[0.5]]], "std": [[[0.25]], [[0.25]], [[0.25]]]}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats, observer="resnet18").to(DEV)
    rng = np.random.default_rng(1)
    lengths = [30, 26]
    N = sum(lengths)
    arrays = {cfg.task.state_key: rng.standard_normal((N, S)).astype(np.float32),
              cam: rng.integers(0, 256, (N, 96, 96, 3), dtype=np.uint8),
              cfg.task.action_key: rng.standard_normal((N, A)).astype(np.float32)}
    dts = {k: v for k, v in policy_delta_timestamps(pol, cfg.task).items() if k in arrays}
    assert cam in dts
    ds = DeviceReplayDataset(arrays, lengths, dts, cfg.task.fps, image_keys=(cam,), device=DEV)
    loader = DeviceBatchLoader(ds, batch_size=16, drop_n_last_frames=8, is_training=True)
    batch = next(iter(loader))
    assert tuple(batch[cam].shape) == (16, 1, 3, 96, 96) and batch[cam].dtype == torch.float32 and float(batch[cam].max()) <= 1.0
    pol.train()
    opt = pol.get_optimizer()
    w0 = pol.observer.backbone.conv1.weight.detach().clone()
    first = None
    for _ in range(3):
        opt.zero_grad()
        torch.manual_seed(5)                       # same crop and same draws: the loss on this batch must go down
        loss, _ = pol.forward(batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(pol.parameters(), cfg.train.grad_clip_norm)
        opt.step()
        first = loss.item() if first is None else first
    assert torch.isfinite(loss) and loss.item() < first
    assert not torch.equal(w0, pol.observer.backbone.conv1.weight)
