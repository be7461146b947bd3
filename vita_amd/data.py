"""Dataset -> device batches, rebuilt for a GPU with 288 GB of HBM.

The reference feeds training from the host: `create_dataloader` (flare/utils/dataset_utils.py:105-138) wraps `AVAlohaDataset`
(gym_av_aloha/datasets/av_aloha_dataset.py:234-372, a zarr replay buffer) in a torch DataLoader with 16 workers and an
`EpisodeAwareSampler`; every item is a window of frames around an index, clamped to the frame's episode, images converted to
float32 / 255 channels-first; default_collate stacks the items and the batch crosses PCIe.

Here the replay buffer is moved to the device once (uint8 images stay uint8: a quarter of the bytes) and a batch is ONE gather
launch per key (`vita_gather_window_f32` / `vita_gather_window_u8_nchw_f32`, include/vita_hip.h): no worker processes, no
collate, no host-to-device copy per step.  The sampler, the index helpers and the batch's keys / dtypes / shapes are the
reference's, so `VitaPolicy.compute_loss(batch)` takes the result unchanged.  Reading the zarr files themselves is outside this
module (zarr is not part of this build): `DeviceReplayDataset` takes the arrays of the replay buffer however they were read.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib as L


class EpisodeAwareSampler:
    """dataset_utils.py:15-55: the frame indices of the chosen episodes, minus `drop_n_first_frames` / `drop_n_last_frames` at
    the ends of each; shuffled per pass with torch.randperm(len(indices)) on the global generator, the sampler's own draw
    (pinned in isolation by tests/golden/dataset.safetensors).  NOT pinned: the order a seeded reference RUN sees -- its
    torch DataLoader draws a base seed from the global generator before the sampler's randperm in every __iter__
    (_BaseDataLoaderIter), which DeviceBatchLoader(match_dataloader_rng=True) restates from torch's public source."""

    def __init__(self, episode_data_index: Mapping[str, torch.Tensor], episode_indices_to_use: Optional[Sequence[int]] = None,
                 drop_n_first_frames: int = 0, drop_n_last_frames: int = 0, shuffle: bool = False):
        starts, ends = episode_data_index["from"], episode_data_index["to"]
        if len(starts) != len(ends):
            raise ValueError("episode_data_index['from'] and ['to'] differ in length")       # zip(strict=True), :37
        use = None if episode_indices_to_use is None else set(int(e) for e in episode_indices_to_use)
        self.indices: List[int] = []
        for ep, (lo, hi) in enumerate(zip(starts, ends)):
            if use is None or ep in use:
                self.indices.extend(range(int(lo) + drop_n_first_frames, int(hi) - drop_n_last_frames))
        self.shuffle = shuffle

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            for i in torch.randperm(len(self.indices)):
                yield self.indices[i]
        else:
            yield from self.indices

    def __len__(self) -> int:
        return len(self.indices)


def cycle(iterable):
    """dataset_utils.py:58-68: itertools.cycle without caching the items (a new pass re-iterates, so a shuffling sampler draws
    a new order every epoch)."""
    iterator = iter(iterable)
    while True:
        try:
            yield next(iterator)
        except StopIteration:
            iterator = iter(iterable)


def get_delta_indices(delta_timestamps: Mapping[str, Sequence[float]], fps: float) -> Dict[str, List[int]]:
    """lerobot get_delta_indices (lerobot/common/datasets/utils.py:572-577): round(delta * fps) per key."""
    return {k: [round(d * fps) for d in v] for k, v in delta_timestamps.items()}


def get_episode_data_index(episode_lengths: Sequence[int]) -> Dict[str, torch.Tensor]:
    """lerobot get_episode_data_index (utils.py:455-466) for all episodes: cumulative frame ranges [from, to)."""
    ends = np.cumsum(np.asarray(list(episode_lengths), dtype=np.int64))
    return {"from": torch.from_numpy(np.concatenate([[0], ends[:-1]]).astype(np.int64)), "to": torch.from_numpy(ends)}


def policy_delta_timestamps(policy, task_cfg) -> Dict[str, List[float]]:
    """create_dataset's window definition (dataset_utils.py:85-92): observation indices for the image keys and the state, action
    indices for the action, in seconds."""
    fps = task_cfg.fps
    obs = [i / fps for i in policy.get_observation_indices()]
    out = {k: list(obs) for k in task_cfg.image_keys}
    out[task_cfg.state_key] = list(obs)
    out[task_cfg.action_key] = [i / fps for i in policy.get_action_indices()]
    return out


class DeviceReplayDataset:
    """AVAlohaDataset on a replay buffer held in device memory.

    arrays: {key: (N, ...) array} — float rows (state, action, precomputed camera latents ...) or uint8 images (N, H, W, C), as
        the zarr replay buffer stores them; 'episode_index' (N) is taken from it or rebuilt from `episode_lengths`.
    delta_timestamps / fps: as AVAlohaDataset takes them (av_aloha_dataset.py:239, 284-286); a key without an entry is not part of
        the batches.
    image_keys: the keys converted as images (:354-356).
    image_transforms: NULL or a callable applied to every image key of every ITEM, (T, C, H, W) float frames in, as :362-365
        (per item, so a random transform draws per item as in the reference; it runs on device tensors).
    tasks: {task_index: str} (meta.tasks); with a 'task_index' array in the replay buffer the batch also carries 'task_index'
        (B, 1), 'task_index_is_pad' and 'task' = the B strings, as :264-266 and :367-370 + default_collate give them."""

    def __init__(self, arrays: Mapping[str, Union[np.ndarray, torch.Tensor]], episode_lengths: Sequence[int],
                 delta_timestamps: Mapping[str, Sequence[float]], fps: float, image_keys: Sequence[str] = (), device="cuda:0",
                 image_transforms=None, tasks: Optional[Mapping[int, str]] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.VitaHipError("DeviceReplayDataset keeps the replay buffer in GPU memory: pass a cuda device")
        self.episode_data_index = get_episode_data_index(episode_lengths)
        self.num_frames = int(self.episode_data_index["to"][-1]) if len(episode_lengths) else 0
        self.delta_indices = get_delta_indices(delta_timestamps, fps)
        self.image_keys = tuple(image_keys)
        ep = arrays.get("episode_index")
        if ep is None:
            ep = np.repeat(np.arange(len(episode_lengths), dtype=np.int64), np.asarray(list(episode_lengths), dtype=np.int64))
        self._ep_index = torch.as_tensor(np.asarray(ep), dtype=torch.int64).to(self.device).contiguous()
        if self._ep_index.numel() != self.num_frames:
            raise L.VitaHipError(f"episode_index has {self._ep_index.numel()} frames, the episode lengths sum to {self.num_frames}")
        self._from = self.episode_data_index["from"].to(self.device)
        self._to = self.episode_data_index["to"].to(self.device)
        self._delta = {k: torch.tensor(v, dtype=torch.int64, device=self.device) for k, v in self.delta_indices.items()}
        self._data: Dict[str, torch.Tensor] = {}
        for k in self.delta_indices:
            if k not in arrays:
                raise L.VitaHipError(f"delta_timestamps names '{k}', which the replay buffer does not hold")
            t = torch.as_tensor(np.asarray(arrays[k])) if not isinstance(arrays[k], torch.Tensor) else arrays[k]
            if t.shape[0] != self.num_frames:
                raise L.VitaHipError(f"'{k}' has {t.shape[0]} frames, expected {self.num_frames}")
            if k in self.image_keys:
                if t.dtype != torch.uint8 or t.dim() != 4:
                    raise L.VitaHipError(f"image key '{k}' must be uint8 (N, H, W, C), got {t.dtype} {tuple(t.shape)}")
            else:
                t = t.to(torch.float32).reshape(self.num_frames, -1) if t.dim() != 2 else t.to(torch.float32)
            self._data[k] = t.to(self.device).contiguous()
        self._shapes = {k: tuple(np.asarray(arrays[k]).shape[1:]) if not isinstance(arrays[k], torch.Tensor) else tuple(arrays[k].shape[1:])
                        for k in self.delta_indices}
        self.image_transforms = image_transforms
        self.tasks = None if tasks is None else {int(k): v for k, v in tasks.items()}
        self._task_index = None
        if "task_index" in arrays:         # av_aloha_dataset.py:264-266: the feature adds delta_timestamps['task_index'] = [0]
            ti = torch.as_tensor(np.asarray(arrays["task_index"]), dtype=torch.int64).reshape(-1)
            if ti.numel() != self.num_frames:
                raise L.VitaHipError(f"'task_index' has {ti.numel()} frames, expected {self.num_frames}")
            self._task_index = ti.to(self.device)
            self._task_index_host = ti.cpu()
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __len__(self) -> int:
        return self.num_frames

    def resident_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._data.values()) + self._ep_index.numel() * 8

    def batch(self, indices: Union[Sequence[int], torch.Tensor], check_indices: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """The collated batch of `[dataset[i] for i in indices]` (av_aloha_dataset.py:349-372 + default_collate): per key a
        (B, T, ...) tensor, `{key}_is_pad` (B, T) bool, `episode_index` (B).

        Index bounds as numpy / zarr indexing has them: a negative index wraps, anything else outside [0, N) raises IndexError.
        Host-side indices (lists, CPU tensors) are checked before any launch; for a device tensor the kernels keep every read
        inside the buffer and set an error word, which `check_indices=True` reads back (a synchronisation) and raises from;
        the default checks host-side indices only."""
        lib = L.load()
        on_device = isinstance(indices, torch.Tensor) and indices.is_cuda
        idx_host = None if on_device else torch.as_tensor(indices, dtype=torch.int64).reshape(-1)
        if idx_host is not None and idx_host.numel() and check_indices is not False:
            lo_, hi_ = int(idx_host.min()), int(idx_host.max())
            if lo_ < -self.num_frames or hi_ >= self.num_frames:
                raise IndexError(f"index {lo_ if lo_ < -self.num_frames else hi_} is out of bounds for a replay buffer of {self.num_frames} frames")
        idx = (indices if on_device else idx_host).to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
        B = idx.numel()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N, E = self.num_frames, int(self._from.numel())
        if B and N == 0:
            raise IndexError("the replay buffer is empty")
        widx = torch.where(idx < 0, idx + N, idx).clamp_(0, max(N - 1, 0)) if B else idx       # torch-side reads stay in bounds too
        out: Dict[str, torch.Tensor] = {"episode_index": self._ep_index[widx]}
        err_ptr = self._err.data_ptr()
        for k, src in self._data.items():
            d = self._delta[k]
            T = d.numel()
            pad = torch.empty((B, T), dtype=torch.uint8, device=self.device)
            if k in self.image_keys:
                _, H, W, Cc = src.shape
                o = torch.empty((B, T, Cc, H, W), dtype=torch.float32, device=self.device)
                L.check(lib.vita_gather_window_u8_nchw_f32(src.data_ptr(), H, W, Cc, idx.data_ptr(), B, d.data_ptr(), T, self._ep_index.data_ptr(),
                                                           self._from.data_ptr(), self._to.data_ptr(), N, E, o.data_ptr(), pad.data_ptr(), err_ptr,
                                                           stream),
                        "vita_gather_window_u8_nchw_f32")
            else:
                R = src.shape[1]
                o = torch.empty((B, T) + self._shapes[k], dtype=torch.float32, device=self.device)
                L.check(lib.vita_gather_window_f32(src.data_ptr(), src.stride(0), R, idx.data_ptr(), B, d.data_ptr(), T, self._ep_index.data_ptr(),
                                                   self._from.data_ptr(), self._to.data_ptr(), N, E, o.data_ptr(), pad.data_ptr(), err_ptr, stream),
                        "vita_gather_window_f32")
            if k in self.image_keys and self.image_transforms is not None and B:      # :362-365, per item
                o = torch.stack([self.image_transforms(o[b]) for b in range(B)])
            out[k] = o
            out[f"{k}_is_pad"] = pad.bool()
        if self._task_index is not None:                                             # :264-266, 367-370
            out["task_index"] = self._task_index[widx].view(B, 1)
            out["task_index_is_pad"] = torch.zeros((B, 1), dtype=torch.bool, device=self.device)
            if self.tasks is not None:
                ti = self._task_index_host[idx_host % max(N, 1)] if idx_host is not None else out["task_index"].view(-1).cpu()
                out["task"] = [self.tasks[int(t)] for t in ti.tolist()]
        if check_indices and on_device and B:
            if int(self._err.item()) != 0:
                self._err.zero_()
                raise IndexError(f"an index of the batch is out of bounds for a replay buffer of {N} frames")
        return out


class DeviceBatchLoader:
    """create_dataloader (dataset_utils.py:105-138) without the DataLoader: batches of `batch_size` indices drawn from the
    sampler (an EpisodeAwareSampler when drop_n_last_frames > 0, else all frames, shuffled when training), last batch kept
    (drop_last=False), each assembled on the device by DeviceReplayDataset.batch."""

    def __init__(self, dataset: DeviceReplayDataset, batch_size: int, drop_n_last_frames: int = 0, is_training: bool = True,
                 match_dataloader_rng: bool = False):
        """match_dataloader_rng: consume the global generator as torch's DataLoader does around the sampler, so that a seeded
        run draws the frame order a seeded reference run draws: every __iter__ first draws the workers' base seed
        (`torch.empty((), dtype=int64).random_()`, _BaseDataLoaderIter.__init__), and a shuffling loader WITHOUT a sampler
        (RandomSampler) seeds a private generator from a second such draw and permutes with it.  Restated from torch's
        public source; no fixture of the reference covers the loader's order: parity UNPINNED (the sampler alone is pinned)."""
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.match_dataloader_rng = bool(match_dataloader_rng)
        if drop_n_last_frames > 0:
            self.sampler = EpisodeAwareSampler(dataset.episode_data_index, drop_n_last_frames=drop_n_last_frames, shuffle=is_training)
        else:
            self.sampler = _AllFrames(len(dataset), shuffle=is_training)

    def __len__(self) -> int:
        return (len(self.sampler) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        buf: List[int] = []
        if self.match_dataloader_rng:
            torch.empty((), dtype=torch.int64).random_()              # the DataLoader iterator's base seed
            if isinstance(self.sampler, _AllFrames):
                self.sampler.private_generator = True
        for i in self.sampler:
            buf.append(int(i))
            if len(buf) == self.batch_size:
                yield self.dataset.batch(buf)
                buf = []
        if buf:
            yield self.dataset.batch(buf)


class _AllFrames:
    """DataLoader(shuffle=...) without a sampler: RandomSampler draws torch.randperm(n) per pass, SequentialSampler range(n)."""

    def __init__(self, n: int, shuffle: bool):
        self.n, self.shuffle = n, shuffle
        self.private_generator = False      # RandomSampler.__iter__: a generator seeded from the global one, then randperm with it

    def __iter__(self):
        if not self.shuffle:
            return iter(range(self.n))
        if self.private_generator:
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            return iter(torch.randperm(self.n, generator=g).tolist())
        return iter(torch.randperm(self.n).tolist())

    def __len__(self):
        return self.n
