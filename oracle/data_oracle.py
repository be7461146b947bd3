"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's dataset -> batch path, SURVEY.md 8(f) #3.  Only tests/ import
it.  Pinned by tests/golden/dataset.safetensors: outputs of the reference's own EpisodeAwareSampler (flare/utils/dataset_utils.py:
15-55) and of AVAlohaDataset._get_query_indices / __getitem__ (gym_av_aloha/datasets/av_aloha_dataset.py:307-320, 349-372) +
torch's default_collate on a synthetic in-memory replay buffer (tools/gen_golden_data.py; zarr, lerobot and torchvision are absent
here and were stubbed for import resolution only — get_delta_indices / get_episode_data_index are restated, UNPINNED: two lines each)."""
from __future__ import annotations

import numpy as np


def episode_aware_indices(ep_from, ep_to, episode_indices_to_use=None, drop_n_first_frames=0, drop_n_last_frames=0):
    """dataset_utils.py:35-43."""
    out = []
    for ep, (lo, hi) in enumerate(zip(ep_from, ep_to)):
        if episode_indices_to_use is None or ep in episode_indices_to_use:
            out.extend(range(int(lo) + drop_n_first_frames, int(hi) - drop_n_last_frames))
    return out


def query_indices(idx, lo, hi, delta):
    """av_aloha_dataset.py:307-320 for one key: clamped source frames and the padding flags."""
    q = [max(lo, min(hi - 1, idx + d)) for d in delta]
    pad = [(idx + d < lo) or (idx + d >= hi) for d in delta]
    return q, pad


def get_item(arrays, ep_from, ep_to, delta_indices, image_keys, idx, image_transforms=None, tasks=None):
    """av_aloha_dataset.py:349-372; image_transforms: a callable on one item's (T, C, H, W) float array (:362-365); tasks: the
    {task_index: str} table, used when delta_indices holds 'task_index' (:264-266, 367-370)."""
    ep = int(arrays["episode_index"][idx])
    lo, hi = int(ep_from[ep]), int(ep_to[ep])
    item = {"episode_index": np.int64(ep)}
    for key, delta in delta_indices.items():
        q, pad = query_indices(idx, lo, hi, delta)
        item[f"{key}_is_pad"] = np.asarray(pad, dtype=bool)
        val = arrays[key][q]
        if key in image_keys:
            val = val.astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
        if key in image_keys and image_transforms is not None:
            val = image_transforms(val)
        item[key] = val
    if "task_index" in item and tasks is not None:
        item["task"] = tasks[int(item["task_index"].item())]
    return item


def collate(items):
    """torch.utils.data.default_collate on dicts of arrays: stack along a new first axis."""
    return {k: ([it[k] for it in items] if isinstance(items[0][k], str) else np.stack([it[k] for it in items])) for k in items[0]}


def batch(arrays, ep_from, ep_to, delta_indices, image_keys, indices, image_transforms=None, tasks=None):
    return collate([get_item(arrays, ep_from, ep_to, delta_indices, image_keys, int(i), image_transforms, tasks) for i in indices])
