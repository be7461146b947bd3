"""Dataset -> batch path on the CPU: the oracle's restatement and vita_amd.data's sampler / index helpers
against tests/golden/dataset.safetensors — outputs of the reference's own EpisodeAwareSampler (dataset_utils.py:15-55), cycle
(:58-68) and AVAlohaDataset.__getitem__ (av_aloha_dataset.py:307-372) + default_collate on a synthetic replay buffer
(tools/gen_golden_data.py)."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle import data_oracle as DO
from vita_amd.data import EpisodeAwareSampler, cycle, get_delta_indices, get_episode_data_index


@pytest.fixture(scope="module")
def gold(golden_dir):
    out = {}
    with safe_open(os.path.join(golden_dir, "dataset.safetensors"), framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
    meta = json.load(open(os.path.join(golden_dir, "dataset.json")))
    return out, meta


@pytest.mark.parametrize("tag", ["all", "drop_last", "drop_both_subset"])
def test_episode_aware_sampler_matches_the_reference(gold, tag):
    g, meta = gold
    edi = get_episode_data_index(meta["lengths"])
    kw = meta[f"sampler/{tag}"]
    s = EpisodeAwareSampler(edi, shuffle=False, **kw)
    assert list(s) == g[f"sampler/{tag}/order"].tolist() and len(s) == len(g[f"sampler/{tag}/order"])
    assert DO.episode_aware_indices(edi["from"], edi["to"], **kw) == g[f"sampler/{tag}/order"].tolist()
    s = EpisodeAwareSampler(edi, shuffle=True, **kw)
    torch.manual_seed(1234)        # the reference draws torch.randperm(len(indices)) per pass
    assert list(s) == g[f"sampler/{tag}/shuffled_seed1234"].tolist()


def test_cycle_reshuffles_every_pass(gold):
    g, meta = gold
    s = EpisodeAwareSampler(get_episode_data_index(meta["lengths"]), shuffle=True, drop_n_last_frames=7)
    torch.manual_seed(7)
    it = cycle(s)
    got = [next(it) for _ in range(2 * len(s))]
    assert got == g["sampler/cycle_seed7_two_passes"].tolist()
    assert got[:len(s)] != got[len(s):] and sorted(got[:len(s)]) == sorted(got[len(s):])


def test_oracle_batch_matches_the_reference_items(gold):
    g, meta = gold
    arrays = {k[len("buffer/"):]: v.numpy() for k, v in g.items() if k.startswith("buffer/")}
    edi = get_episode_data_index(meta["lengths"])
    b = DO.batch(arrays, edi["from"].numpy(), edi["to"].numpy(), meta["delta_indices"], meta["image_keys"], g["batch/indices"].tolist())
    want = {k[len("batch/out/"):]: v.numpy() for k, v in g.items() if k.startswith("batch/out/")}
    assert set(b) == set(want)
    for k, v in want.items():
        got = b[k].astype(np.uint8) if b[k].dtype == bool else b[k]
        assert got.shape == v.shape and np.array_equal(got, v), k        # bit-exact: gathers, and uint8 / 255 in float32


def test_index_helpers():
    assert get_delta_indices({"a": [-0.1, 0.0, 0.1, 1.5]}, 10) == {"a": [-1, 0, 1, 15]}
    assert get_delta_indices({"a": [i / 25 for i in range(-1, 16)]}, 25) == {"a": list(range(-1, 16))}
    edi = get_episode_data_index([3, 1, 4])
    assert edi["from"].tolist() == [0, 3, 4] and edi["to"].tolist() == [3, 4, 8]


def test_oracle_item_extras_match_the_reference(gold):
    """av_aloha_dataset.py:264-266, 362-370: 'task_index' joins the window keys, image_transforms runs on every camera key of an
    item, the task's string joins the item; default_collate keeps the strings as a list."""
    g, meta = gold
    arrays = {k[len("buffer/"):]: v.numpy() for k, v in g.items() if k.startswith("buffer/")}
    arrays["task_index"] = g["extras/task_index"].numpy()
    edi = get_episode_data_index(meta["lengths"])
    tasks = {int(k): v for k, v in meta["extras/tasks"].items()}
    b = DO.batch(arrays, edi["from"].numpy(), edi["to"].numpy(), dict(meta["delta_indices"], task_index=[0]), meta["image_keys"],
                 g["batch/indices"].tolist(), image_transforms=lambda x: x[..., ::-1] * np.float32(0.5) + np.float32(0.25), tasks=tasks)
    assert b.pop("task") == meta["extras/task"]
    want = {k[len("extras/out/"):]: v.numpy() for k, v in g.items() if k.startswith("extras/out/")}
    assert set(b) == set(want)
    for k, v in want.items():
        got = b[k].astype(np.uint8) if b[k].dtype == bool else b[k]
        assert got.shape == v.shape and np.array_equal(got, v), k
