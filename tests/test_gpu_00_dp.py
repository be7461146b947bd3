"""GPU: the data-parallel train_step with two ranks (both on cuda:0, gloo backend — RCCL refuses two ranks on one
device, and the box has one GPU; the trainer's event / side-stream / bucket logic is the same code for any backend).
With gloo the reducer stages each bucket through pinned host memory itself (ProcessGroupGloo's own CUDA-tensor path hung
intermittently here with two ranks on one device: 6 of 13 runs, always inside dist.all_reduce of the second step).
Two ranks each train on half of a global batch; the result must equal one process training on the whole batch:
equal shard sizes => mean of the shard-mean gradients == global-batch-mean gradient.

Process hygiene: the file name sorts first among the GPU tests so that the parent has not initialised HIP when the ranks
are started, and every helper process (ranks AND the result manager) is created with the "spawn" method: a fork()ed copy
of a process that holds a HIP context hung intermittently on the GPU box."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from vita_amd.config import default_config
    cfg = default_config("robomimic_square")
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 48, "dec_hidden_dim": 48})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 2})
    cfg.policy.flow_matcher.name = "conditional"
    cfg.scheduler_name = "constant"
    cfg.optimizer_lr = 1e-3
    cfg.optimizer_eps = 1e-3
    return cfg


def _policy(cfg):
    from vita_amd.policy import VitaPolicy
    torch.manual_seed(0)
    s, a = cfg.task.state_dim, cfg.task.action_dim
    stats = {cfg.task.state_key: {"mean": [0.1] * s, "std": [0.9] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [1.2] * a}}
    return VitaPolicy(cfg, stats)


def _batches(cfg, B, steps):
    g = torch.Generator().manual_seed(7)
    out = []
    for _ in range(steps):
        out.append(({cfg.task.state_key: torch.randn(B, 1, cfg.task.state_dim, generator=g),
                     "observation.features": torch.relu(torch.randn(B, 1, 512, generator=g)),
                     cfg.task.action_key: torch.randn(B, 16, cfg.task.action_dim, generator=g)}, torch.rand(B, generator=g)))
    return out


STEPS, B = 3, 32


def _worker(rank, world, port, out):
    import faulthandler
    dbg = os.environ.get("VITA_DP_DEBUG_DIR")
    if dbg:   # where a rank is stuck, should it ever hang: stacks of all threads after 40 s
        faulthandler.dump_traceback_later(40, repeat=False, file=open(os.path.join(dbg, f"dp_rank{rank}.txt"), "w"), exit=True)
    import torch.distributed as dist
    from vita_amd.parallel import shard_range
    from vita_amd.trainer import PolicyTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _cfg()
        pol = _policy(cfg)
        if rank == 1:                      # replicas must start from rank 0's weights: perturb, the trainer broadcasts
            with torch.no_grad():
                for p in pol.parameters():
                    if p.requires_grad:
                        p.add_(1.0)
        tr = PolicyTrainer(cfg, pol, "cuda:0", num_training_steps=100, max_bucket_elems=20000)
        assert tr.world == world and tr.events is not None
        lo, hi = shard_range(B, rank, world)
        for batch, t in _batches(cfg, B, STEPS):
            tr.train_step({k: v[lo:hi] for k, v in batch.items()}, t=t[lo:hi].to("cuda:0"))
        torch.cuda.synchronize()
        out[rank] = (tr.flat.cpu(), tr.ema.cpu(), tr.metrics()["grad_norm"])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_equals_single_process_on_the_global_batch():
    from vita_amd.trainer import PolicyTrainer
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.start_processes(_worker, args=(world, _free_port(), out), nprocs=world, join=True, start_method="spawn")
    assert set(out.keys()) == {0, 1}
    p0, e0, gn0 = out[0]
    p1, e1, gn1 = out[1]
    mgr.shutdown()
    d = (p0 - p1).abs()
    assert torch.equal(p0, p1) and torch.equal(e0, e1), f"replicas diverged: max {d.max().item():g} at {d.argmax().item()} of {d.numel()}, {int((d > 0).sum())} differ"
    cfg = _cfg()
    tr = PolicyTrainer(cfg, _policy(cfg), "cuda:0", num_training_steps=100)
    for batch, t in _batches(cfg, B, STEPS):
        tr.train_step(batch, t=t.to("cuda:0"))
    torch.cuda.synchronize()
    ref = tr.flat.cpu()
    err = ((p0 - ref).abs().max() / ref.abs().max()).item()
    assert err <= 2e-5, f"data-parallel parameters differ from the single-process run: {err:g}"
    assert abs(gn0 - tr.metrics()["grad_norm"]) <= 1e-4 * gn0


def _rccl_worker(rank, port, out):
    import faulthandler
    dbg = os.environ.get("VITA_DP_DEBUG_DIR")
    if dbg:
        faulthandler.dump_traceback_later(60, repeat=False, file=open(os.path.join(dbg, "dp_rccl_rank0.txt"), "w"), exit=True)
    import torch.distributed as dist
    from vita_amd.trainer import PolicyTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # as bench.py does for N > 1
    try:
        cfg = _cfg()
        tr = PolicyTrainer(cfg, _policy(cfg), dev, num_training_steps=100, max_bucket_elems=20000, rehearse_collectives=True)
        assert tr.reducer.active and not tr.reducer.host_staged and tr.events is not None and tr.reducer.stream is not None
        for batch, t in _batches(cfg, B, STEPS):
            tr.train_step(batch, t=t.to(dev))
        dist.barrier()
        torch.cuda.synchronize()
        out["rccl"] = (tr.flat.cpu(), tr.ema.cpu(), tr.metrics()["grad_norm"])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_branch_one_rank_rehearsal_equals_the_plain_trainer():
    """The RCCL ("nccl") branch of BucketReducer / PolicyTrainer — in-place device all-reduce on the side stream behind the
    per-layer gradient-ready events, parameter broadcast, barrier — executed for real in a one-rank group (the box has one
    GPU and RCCL refuses two ranks on one device).  A sum over one rank is the identity, so three steps must leave exactly the
    parameters, EMA and gradient norm of the trainer without a process group."""
    from vita_amd.trainer import PolicyTrainer
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.start_processes(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True, start_method="spawn")
    p0, e0, gn0 = out["rccl"]
    mgr.shutdown()
    cfg = _cfg()
    tr = PolicyTrainer(cfg, _policy(cfg), "cuda:0", num_training_steps=100)
    assert not tr.reducer.active
    for batch, t in _batches(cfg, B, STEPS):
        tr.train_step(batch, t=t.to("cuda:0"))
    torch.cuda.synchronize()
    assert torch.equal(p0, tr.flat.cpu()) and torch.equal(e0, tr.ema.cpu())
    assert gn0 == tr.metrics()["grad_norm"]


def _obs_setup():
    """pusht dims with the ResNet18 observer on 96 x 96 frames (tests/test_gpu_observer.py's policy), a batch of 8."""
    from vita_amd.config import default_config
    from vita_amd.policy import VitaPolicy
    cfg = default_config("pusht")
    cfg.policy.vita.latent_dim = 64
    cfg.policy.action_ae.net.merge({"latent_dim": 64, "enc_hidden_dim": 48, "dec_hidden_dim": 48, "num_layers": 4})
    cfg.policy.flow_net.merge({"hidden_dim": 64, "num_layers": 2})
    cfg.policy.flow_matcher.name = "conditional"
    cfg.scheduler_name = "constant"
    cfg.optimizer_lr = 1e-3
    cfg.optimizer_eps = 1e-3
    s, a, cam = cfg.task.state_dim, cfg.task.action_dim, cfg.task.image_keys[0]
    stats = {cfg.task.state_key: {"mean": [0.1] * s, "std": [1.5] * s}, cfg.task.action_key: {"mean": [0.0] * a, "std": [2.0] * a},
             cam: {"mean": [[[0.485]], [[0.456]], [[0.406]]], "std": [[[0.229]], [[0.224]], [[0.225]]]}}
    torch.manual_seed(0)
    pol = VitaPolicy(cfg, stats, observer="resnet18")
    g = torch.Generator().manual_seed(3)
    batch = {cfg.task.state_key: torch.randn(8, 1, s, generator=g), cam: torch.rand(8, 1, 3, 96, 96, generator=g),
             cfg.task.action_key: torch.randn(8, 16, a, generator=g)}
    return cfg, pol, batch, torch.rand(8, generator=g)


def _obs_steps(tr, batch, t, dev):
    for step in range(2):
        torch.manual_seed(40 + step)            # the observer's crop draw
        tr.train_step(batch, t=t.to(dev))
    torch.cuda.synchronize()
    return tr.flat.cpu(), tr.metrics()["grad_norm"]


def _rccl_observer_worker(rank, port, out):
    import torch.distributed as dist
    from vita_amd.trainer import PolicyTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg, pol, batch, t = _obs_setup()
        tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100, rehearse_collectives=True)
        assert tr.reducer.active and tr.observer_segment and tr.plan[-1][:2] == pol.observer_range
        assert len(tr.events) == 4 + cfg.policy.flow_net.num_layers + 1      # the fused step's events + the observer's
        out["obs"] = _obs_steps(tr, batch, t, dev)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_rehearsal_with_a_trainable_observer():
    """The data-parallel step with the ResNet18 observer in the flat buffers: the fused step's 4 + L events, then the observer's
    segment behind torch's backward of the backbone (its event recorded by the policy), all-reduced on the side stream in a
    one-rank RCCL group.  Against the trainer without a process group (MIOpen's weight gradients are not run-to-run bitwise:
    2e-4 on the parameters after two steps)."""
    from vita_amd.trainer import PolicyTrainer
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.start_processes(_rccl_observer_worker, args=(_free_port(), out), nprocs=1, join=True, start_method="spawn")
    p0, gn0 = out["obs"]
    mgr.shutdown()
    cfg, pol, batch, t = _obs_setup()
    tr = PolicyTrainer(cfg, pol, "cuda:0", num_training_steps=100)
    ref, gn = _obs_steps(tr, batch, t, "cuda:0")
    err = ((p0 - ref).abs().max() / ref.abs().max()).item()
    assert err <= 2e-4, err
    assert abs(gn0 - gn) <= 1e-3 * gn


def _mf_setup():
    cfg = _cfg()
    cfg.policy.flow_matcher.name = "mean"
    cfg.policy.flow_net.name = "simple_mean_flow_net"
    cfg.policy.flow_matcher.num_sampling_steps = 2
    pol = _policy(cfg)
    with torch.no_grad():      # the reference zero-initialises the MeanFlow heads: give them values
        for k, p in pol.named_parameters():
            if "output_layer" in k:
                p.copy_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(3)))
    return cfg, pol


def _mf_steps(tr, cfg, dev):
    g = torch.Generator().manual_seed(11)
    for batch, t in _batches(cfg, B, STEPS):
        r = t * torch.rand(B, generator=g)
        tr.train_step(batch, t=t.to(dev), pair_i=r.to(dev))
    torch.cuda.synchronize()
    return tr.flat.cpu(), tr.ema.cpu(), tr.metrics()["grad_norm"]


def _rccl_meanflow_worker(rank, port, out):
    import torch.distributed as dist
    from vita_amd.trainer import PolicyTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg, pol = _mf_setup()
        tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100, max_bucket_elems=20000, rehearse_collectives=True)
        L_ = cfg.policy.flow_net.num_layers
        assert tr.reducer.active and len(tr.events) == 4 + L_                     # per-layer events for the MeanFlow step too
        assert [ev for _, _, ev in tr.plan] == [0] + [3 + l for l in range(L_)] + [1, 3 + L_, 2]   # the order they fire in
        assert tr.plan[0][0] == 0 and sorted((lo, hi) for lo, hi, _ in tr.plan)[-1][1] == tr.flat.numel()
        out["mf"] = _mf_steps(tr, cfg, dev)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_rehearsal_of_the_meanflow_step_with_per_layer_events():
    """vita_meanflow_loss_config.grad_ready_layers (round 3): the MeanFlow step records 4 + L events — decoder, each hidden layer,
    the time / interval embeddings, input layer + heads, encoders — and the reducer all-reduces one contiguous segment behind each.
    One-rank RCCL group: three steps must leave exactly the parameters of the trainer without a process group (which passes no
    events and runs the weight gradients in the single-event order)."""
    from vita_amd.trainer import PolicyTrainer
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.start_processes(_rccl_meanflow_worker, args=(_free_port(), out), nprocs=1, join=True, start_method="spawn")
    p0, e0, gn0 = out["mf"]
    mgr.shutdown()
    cfg, pol = _mf_setup()
    tr = PolicyTrainer(cfg, pol, "cuda:0", num_training_steps=100)
    p1, e1, gn1 = _mf_steps(tr, cfg, "cuda:0")
    assert torch.equal(p0, p1) and torch.equal(e0, e1) and gn0 == gn1
