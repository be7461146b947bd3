#!/usr/bin/env python3
"""tests/golden/dataset.safetensors (+ dataset.json): the reference's EpisodeAwareSampler and AVAlohaDataset.__getitem__ +
default_collate on a synthetic in-memory replay buffer (build container only; only the fixture travels).

zarr, lerobot, torchvision, omegaconf and the gym_av_aloha package are not importable here: they are stubbed for IMPORT
RESOLUTION ONLY (empty stand-ins whose functionality is never called); the code that runs is the reference's own sampler and its
own _get_query_indices / __getitem__, on an AVAlohaDataset instance created without __init__ (which would open a zarr store)."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
from safetensors.torch import save_file

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError("stand-in for an absent library: must not be used")


never = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in for an absent library: must not be called"))  # noqa: E731
stub("omegaconf", DictConfig=_Absent, OmegaConf=_Absent)
pkg = stub("gym_av_aloha", __file__=os.path.join(REF, "gym-av-aloha", "gym_av_aloha", "__init__.py"))
stub("gym_av_aloha.common")
stub("gym_av_aloha.common.replay_buffer", ReplayBuffer=_Absent)
stub("lerobot"); stub("lerobot.common"); stub("lerobot.common.datasets")
stub("lerobot.common.datasets.lerobot_dataset", LeRobotDatasetMetadata=_Absent, LeRobotDataset=_Absent)
stub("lerobot.common.datasets.utils", check_delta_timestamps=never, get_delta_indices=never, get_episode_data_index=never,
     check_timestamps_sync=never)
stub("lerobot.common.datasets.compute_stats", aggregate_stats=never)
stub("torchvision"); stub("torchvision.transforms", Resize=_Absent)
avd = load_by_path("ref_av_aloha_dataset", os.path.join(REF, "gym-av-aloha", "gym_av_aloha", "datasets", "av_aloha_dataset.py"))
stub("gym_av_aloha.datasets")
stub("gym_av_aloha.datasets.av_aloha_dataset", AVAlohaDataset=avd.AVAlohaDataset, AVAlohaDatasetMeta=getattr(avd, "AVAlohaDatasetMeta", _Absent))
du = load_by_path("ref_dataset_utils", os.path.join(REF, "flare", "utils", "dataset_utils.py"))


def main():
    out, meta = {}, {}
    rng = np.random.default_rng(0)
    lengths = [10, 25, 5, 20]
    N = sum(lengths)
    ends = np.cumsum(lengths)
    edi = {"from": torch.LongTensor(np.concatenate([[0], ends[:-1]])), "to": torch.LongTensor(ends)}
    arrays = {"episode_index": np.repeat(np.arange(len(lengths)), lengths).astype(np.int64),
              "observation.state": rng.standard_normal((N, 7)).astype(np.float32),
              "action": rng.standard_normal((N, 6)).astype(np.float32),
              "observation.images.cam": rng.integers(0, 256, size=(N, 12, 16, 3), dtype=np.uint8)}
    delta_indices = {"observation.images.cam": [-1, 0], "observation.state": [-1, 0], "action": list(range(16))}
    # ---- sampler
    cases = {"all": dict(), "drop_last": dict(drop_n_last_frames=7), "drop_both_subset": dict(drop_n_first_frames=2, drop_n_last_frames=3,
                                                                                              episode_indices_to_use=[0, 3])}
    for tag, kw in cases.items():
        s = du.EpisodeAwareSampler(edi, shuffle=False, **kw)
        out[f"sampler/{tag}/order"] = torch.tensor(list(iter(s)), dtype=torch.int64)
        assert len(s) == len(out[f"sampler/{tag}/order"])
        s = du.EpisodeAwareSampler(edi, shuffle=True, **kw)
        torch.manual_seed(1234)
        out[f"sampler/{tag}/shuffled_seed1234"] = torch.tensor(list(iter(s)), dtype=torch.int64)
        meta[f"sampler/{tag}"] = kw
    # cycle: two passes of a shuffling sampler draw two orders
    s = du.EpisodeAwareSampler(edi, shuffle=True, drop_n_last_frames=7)
    torch.manual_seed(7)
    it = du.cycle(s)
    out["sampler/cycle_seed7_two_passes"] = torch.tensor([next(it) for _ in range(2 * len(s))], dtype=torch.int64)
    # ---- dataset items through the reference's own methods
    ds = object.__new__(avd.AVAlohaDataset)
    ds.episode_data_index = edi
    ds.delta_indices = delta_indices
    ds.replay_buffer = arrays
    ds.image_transforms = None
    ds.meta = types.SimpleNamespace(image_keys=["observation.images.cam"], video_keys=[], camera_keys=["observation.images.cam"], tasks={},
                                    features={}, stats={}, fps=10)
    indices = [0, 1, 9, 10, 11, 20, 33, 34, 35, 39, 40, 45, 58, 59, 5, 5]
    items = [ds[i] for i in indices]
    batch = torch.utils.data.default_collate(items)
    out["batch/indices"] = torch.tensor(indices, dtype=torch.int64)
    for k, v in batch.items():
        out[f"batch/out/{k}"] = (v.to(torch.uint8) if v.dtype == torch.bool else v).contiguous()
    for k, v in arrays.items():
        out[f"buffer/{k}"] = torch.from_numpy(v)
    # ---- the item extras (av_aloha_dataset.py:264-266, 362-370): a 'task_index' feature adds delta_timestamps['task_index'] = [0]
    # (what __init__ does), image_transforms runs on every camera key, and the task's string joins the item
    arrays2 = dict(arrays, task_index=rng.integers(0, 3, size=N).astype(np.int64))
    ds2 = object.__new__(avd.AVAlohaDataset)
    ds2.episode_data_index = edi
    ds2.delta_indices = dict(delta_indices, task_index=[0])
    ds2.replay_buffer = arrays2
    ds2.image_transforms = lambda x: x.flip(-1) * 0.5 + 0.25       # any callable on the (T, C, H, W) float frames of one item
    tasks = {0: "hook the package", 1: "thread the needle", 2: "push the T"}
    ds2.meta = types.SimpleNamespace(image_keys=["observation.images.cam"], video_keys=[], camera_keys=["observation.images.cam"], tasks=tasks,
                                     features={"task_index": {}}, stats={}, fps=10)
    batch2 = torch.utils.data.default_collate([ds2[i] for i in indices])
    meta["extras/tasks"] = {str(k): v for k, v in tasks.items()}
    meta["extras/task"] = list(batch2.pop("task"))                  # default_collate keeps the strings as a list
    for k, v in batch2.items():
        out[f"extras/out/{k}"] = (v.to(torch.uint8) if v.dtype == torch.bool else v).contiguous()
    out["extras/task_index"] = torch.from_numpy(arrays2["task_index"])
    meta["lengths"], meta["delta_indices"], meta["image_keys"] = lengths, delta_indices, ["observation.images.cam"]
    save_file(out, os.path.join(REPO, "tests", "golden", "dataset.safetensors"))
    json.dump(meta, open(os.path.join(REPO, "tests", "golden", "dataset.json"), "w"), indent=1)
    print("ok", len(out), "tensors")


if __name__ == "__main__":
    main()
