#!/usr/bin/env python3
"""bench.py — ODE action-chunks/sec of the VITA hot path on MI355X.

A "step" is one pass of `VitaPolicy.generate_actions` from observer features on
(obs_encoder -> K-step Euler ODE over SimpleFlowNet -> action decoder, vita_policy.py:191-208)
over one batch of synthetic hook_package-shaped input already resident in HBM: B=4096 rows per
GPU, O=533, D=H=512, L=4, R=4, K=6, T*A=336, fp32 (BASELINE.json configs[1]).  With N>1 every
rank runs its own 4096-row shard (rows are independent: no collective on the data path) and the
value is the sum over ranks divided by the slowest rank's time (weak scaling).

Contract: python bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic FLOPs (2*M*N*K, x-dependent GEMMs only; SURVEY.md 8(d) / BASELINE.md section 4)
def flops_per_chunk(O, D, H, FF, L, K, HD, NL, OUT):
    f_step = 2 * D * H + L * (2 * H * FF + 2 * FF * H) + 2 * H * D
    f_dec = 2 * D * HD + NL * (2 * 2 * HD * HD) + 2 * HD * OUT
    return 2 * O * D + K * f_step + f_dec


PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
TASKS = {"hook_package": (533, 21), "robomimic_square": (555, 7), "pusht": (514, 2), "thread_needle": (533, 21)}


def self_launch(n: int) -> int:
    """One child process per GPU with the torch.distributed.run environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*);
    rank 0's stdout is relayed, the other ranks' goes to stderr.  Returns the first non-zero exit code, else 0."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for pr in procs:
        c = pr.wait()
        rc = rc or c
    return rc


def cpu_baseline(args, sd, obs, K, A, fpc):
    """the oracle (CPU restatement of the reference, torch CPU fp32; kind "port") on this box's host
    cores: hook_package inference on a bounded row sample, pusht B=64 inference and one full pusht train step (forward,
    autograd backward, clip 1.0, AdamW) -- each at all cores of this job's share and at 1 thread, 3 warm-ups + the median of
    up to 10 timed iterations, the whole leg inside --cpu-seconds."""
    import statistics
    from oracle import vita_oracle as O   # the checker, timed here as the reported CPU baseline only
    from vita_amd.synthetic import make_state_dict
    ncores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    ncores = max(1, min(ncores, os.cpu_count() or 1))
    deadline = time.perf_counter() + args.cpu_seconds

    def med(fn, warm=3, iters=10):
        ts = []
        for i in range(warm + iters):
            if ts and time.perf_counter() > deadline:
                break
            t0 = time.perf_counter()
            fn()
            if i >= warm or time.perf_counter() > deadline:
                ts.append(time.perf_counter() - t0)
        return statistics.median(ts), len(ts)

    psd = make_state_dict(514, 2, seed=0)
    g = torch.Generator().manual_seed(1)
    pobs, pact = torch.randn(64, 514, generator=g), torch.randn(64, 16, 2, generator=g)
    pt = torch.rand(64, generator=torch.Generator().manual_seed(2))
    f_pusht = flops_per_chunk(514, 512, 512, 2048, 4, K, 512, 4, 32)

    def pusht_train():
        leaf = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        loss, _ = O.compute_loss(leaf, pobs, pact, pt, K)
        loss.backward()
        grads = [p.grad for p in leaf.values()]
        gn = torch.sqrt(sum((g_.double() ** 2).sum() for g_ in grads))
        coef = min(1.0, 1.0 / (gn.item() + 1e-6))
        with torch.no_grad():
            for p_ in leaf.values():
                O.adamw_step(p_, p_.grad * coef, torch.zeros_like(p_), torch.zeros_like(p_), 1, 1e-4)

    out = {"unit": "action-chunks/s", "kind": "port", "cores": ncores, "cpu_model": cpu_model(), "torch": torch.__version__,
           "host_cpus_visible": os.cpu_count(), "protocol": "3 warm-ups + median of <= 10 iterations per figure, time-boxed"}
    legs = {}
    for nthreads in (ncores, 1):
        torch.set_num_threads(nthreads)
        rows = args.cpu_rows if nthreads > 1 else max(64, args.cpu_rows // 4)
        cobs = obs[:rows].cpu()
        with torch.no_grad():
            dt, n = med(lambda: O.generate_actions(sd, cobs, K, 16, A))
            hp = {"chunks_per_s": round(rows / dt, 1), "rows": rows, "iters": n, "gflops": round(rows * fpc / dt / 1e9, 1)}
            dt, n = med(lambda: O.generate_actions(psd, pobs, K, 16, 2))
            pi = {"chunks_per_s": round(64 / dt, 1), "ms": round(dt * 1e3, 2), "iters": n, "gflops": round(64 * f_pusht / dt / 1e9, 1)}
        dt, n = med(pusht_train, warm=1, iters=10)
        ptn = {"samples_per_s": round(64 / dt, 1), "ms_per_step": round(dt * 1e3, 1), "iters": n}
        legs[f"{nthreads}_threads"] = {"hook_package_inference": hp, "pusht_b64_inference": pi, "pusht_b64_train_step": ptn}
    torch.set_num_threads(ncores)
    out["value"] = legs[f"{ncores}_threads"]["hook_package_inference"]["chunks_per_s"]
    out["sample"] = (f"hook_package generate_actions, {legs[f'{ncores}_threads']['hook_package_inference']['rows']} of the {obs.shape[0]} rows "
                     f"(same weights / inputs, K={K}), {ncores} threads, median of {legs[f'{ncores}_threads']['hook_package_inference']['iters']}")
    out["configs"] = legs
    return out


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--ode-steps", type=int, default=6)
    ap.add_argument("--task", default="hook_package", choices=sorted(TASKS))
    ap.add_argument("--hidden", type=int, default=512, help="flow-net hidden_dim (BASELINE configs[4]: 1024 / 2048 = the wide velocity net)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-batch", type=int, default=128, help="training rows per GPU (default_policy.yaml:34)")
    ap.add_argument("--train-steps", type=int, default=0, help="timed training steps (0 = same as --steps)")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-large-batch", type=int, default=4096, help="N=1 only: side figures of the training step at this batch, fp32 and bf16 operands (0 = skip)")
    ap.add_argument("--cpu-rows", type=int, default=1024, help="rows of the hook_package CPU-baseline sample at all cores")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="time box of the whole cpu_baseline leg")
    ap.add_argument("--dry-launch", action="store_true", help="ranks print their launch environment as JSON and exit (no GPU touched): "
                                                              "the CPU-side test of the N>1 self-launch")
    ap.add_argument("--rehearse-dist", action="store_true", help="N=1 only: initialise the RCCL process group with one rank and issue "
                                                                 "every collective of the N>1 path anyway (a rehearsal of that path on a one-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver may call it: start N fresh rank processes (never re-exec this one; nothing
        # here has touched the GPU yet), relay rank 0's line, fail if any rank fails
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master_addr": os.environ.get("MASTER_ADDR"),
                          "master_port": os.environ.get("MASTER_PORT"), "pid": os.getpid(), "ppid": os.getppid()}), flush=True)
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.rehearse_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:        # --rehearse-dist started by hand
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from vita_amd import ops, _lib
    from vita_amd.synthetic import make_state_dict
    lib = _lib.load()

    obs_dim, A = TASKS[args.task]
    B, K = args.batch, args.ode_steps
    sd = make_state_dict(obs_dim, A, hidden_dim=args.hidden, seed=0)
    fw, dw = ops.FlowNetHandle(sd, dev), ops.DecoderHandle(sd, dev)
    ow, ob = sd["obs_encoder.weight"].to(dev), sd["obs_encoder.bias"].to(dev)
    g = torch.Generator().manual_seed(1 + rank)
    obs = torch.randn(B, obs_dim, generator=g).to(dev)
    out = torch.empty(B, dw.OUT, device=dev)

    fpc = flops_per_chunk(obs_dim, fw.D, fw.H, fw.FF, fw.L, K, dw.HD, dw.w.NL, dw.OUT)

    def step():
        ops.generate_actions(ow, ob, fw, dw, obs, K, out=out)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed * 1e3 / args.steps
    chunks_per_s = world * B * args.steps / elapsed

    # ---- strong scaling (BASELINE configs[3]): ONE global batch of `B` rows cut into contiguous row shards
    # (vita_amd.parallel.shard_range), no collective on the data path.  N > 1: every rank runs its shard, max over ranks.
    # N == 1: the shard sizes of N = 2 / 4 / 8 timed on this GPU -- what each rank of such a job runs (rows are independent, so
    # this is the job's time up to box-to-box clock differences).
    from vita_amd.parallel import shard_range

    def time_rows(rows, iters):
        o_, y_ = obs[:rows].contiguous(), torch.empty(rows, dw.OUT, device=dev)
        for _ in range(3):
            ops.generate_actions(ow, ob, fw, dw, o_, K, out=y_)
        fence()
        t0_ = time.perf_counter()
        for _ in range(iters):
            ops.generate_actions(ow, ob, fw, dw, o_, K, out=y_)
        fence()
        return (time.perf_counter() - t0_) / iters

    strong = None
    if world > 1:
        lo, hi = shard_range(B, rank, world)
        ts_ = torch.tensor([time_rows(hi - lo, args.steps)], device=dev, dtype=torch.float64)
        dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
        strong = {"global_batch": B, "rows_per_gpu": hi - lo, "ms_per_step": round(ts_.item() * 1e3, 4),
                  "chunks_per_s": round(B / ts_.item(), 1),
                  "efficiency_vs_one_gpu_same_run": round((ms_per_step * 1e-3) / (world * ts_.item()), 4),
                  "note": "one 4096-row batch sharded by rows over the ranks, no collective; efficiency = (time of all 4096 rows on one GPU, "
                          "this run's weak leg) / (N x time of the shard)"}
    else:
        strong = {"global_batch": B, "predicted_from_one_gpu": True, "shards": {}}
        for n_ in (2, 4, 8):
            rows = shard_range(B, 0, n_)[1]
            t_ = time_rows(rows, max(5, args.steps))
            strong["shards"][f"{n_}_gpus"] = {"rows_per_gpu": rows, "ms_per_step": round(t_ * 1e3, 4), "chunks_per_s": round(B / t_, 1),
                                              "efficiency_vs_one_gpu": round((ms_per_step * 1e-3) / (n_ * t_), 4)}

    # latent-only sampler (B,D)->(B,D): "the fused ODE sampler" of the north star.  At chip-filling batches it is ONE persistent
    # launch per solve (+ a few tiny time-path launches): HIP events on the launch stream around n solves
    x0 = torch.randn(B, fw.D, generator=g).to(dev)
    for _ in range(2):
        ops.ode_sample(fw, x0, K)
    torch.cuda.synchronize(dev)
    n_lat = max(3, args.steps // 2)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n_lat):
        ops.ode_sample(fw, x0, K)
    ev1.record()
    ev1.synchronize()
    lat_ms = ev0.elapsed_time(ev1) / n_lat
    f_step = 2 * fw.D * fw.H + fw.L * 4 * fw.H * fw.FF + 2 * fw.H * fw.D
    lat_tflops = B * K * f_step / (lat_ms * 1e-3) / 1e12
    import ctypes
    persistent = bool(lib.vita_ode_sample_takes_persistent(ctypes.byref(fw.w), B))
    sampler_status = ops.ode_sample_status(fw, B, K, dev)

    # ---- serving-size latency (rank 0): generate_actions at eval_n_envs rows (default_policy.yaml:48), the weights-resident
    # launch against the per-launch path
    serving = None
    if rank == 0 and args.hidden == 512:
        SB = 25
        sobs = obs[:SB].contiguous()
        sout = torch.empty(SB, dw.OUT, device=dev)
        serving = {"batch": SB, "ode_steps": K, "what": "generate_actions, obs features -> actions, one call per solve, back to back",
                   "takes_resident_launch": int(lib.vita_ode_sample_takes_persistent(ctypes.byref(fw.w), SB)) == 2}
        for name, on in (("resident", True), ("per_launch", False)):
            prev = ops.set_persistent_sampler(on)
            for _ in range(10):
                ops.generate_actions(ow, ob, fw, dw, sobs, K, out=sout)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(200):
                ops.generate_actions(ow, ob, fw, dw, sobs, K, out=sout)
            torch.cuda.synchronize(dev)
            serving[f"ms_per_solve_{name}"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
            ops.set_persistent_sampler(prev)

    # ---- bf16-operand variant of the same path (side figure: NOT the headline, the reference is strict fp32)
    bf16 = None
    if rank == 0:
        bw = ops.Bf16Weights(sd, dev)
        outb = torch.empty_like(out)
        for _ in range(3):
            ops.generate_actions_bf16(ob, fw, dw, bw, obs, K, out=outb)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nb = max(5, args.steps)
        for _ in range(nb):
            ops.generate_actions_bf16(ob, fw, dw, bw, obs, K, out=outb)
        torch.cuda.synchronize(dev)
        bms = (time.perf_counter() - t0) * 1e3 / nb
        step()
        torch.cuda.synchronize(dev)
        err = ((outb - out).abs().max() / out.abs().max()).item()
        bf16 = {"value": round(B / (bms * 1e-3), 1), "unit": "action-chunks/s", "ms_per_step": round(bms, 4), "dtype": "bf16 operands, f32 accumulate",
                "tflops": round(B * fpc / (bms * 1e-3) / 1e12, 2), "frac_of_bf16_mfma_peak": round(B * fpc / (bms * 1e-3) / 1e12 / 2500.0, 4),
                "normalised_max_error_vs_f32_path": float(f"{err:.3g}"), "stated_tolerance": 2e-2}

    # ---- training leg: data-parallel train_step (normalize -> fused fwd+bwd -> bucketed all-reduce -> clip+AdamW+EMA)
    def train_leg(TB, gemm_dtype, nts, all_ranks, group=None):
        from vita_amd.config import default_config
        from vita_amd.policy import VitaPolicy
        from vita_amd.trainer import PolicyTrainer
        cfg = default_config(args.task)
        cfg.policy.flow_matcher.name = "conditional"     # identity pairing for timing; the OT draw is host work
        cfg.policy.gemm_dtype = gemm_dtype
        S = cfg.task.state_dim
        stats = {cfg.task.state_key: {"mean": [0.0] * S, "std": [1.0] * S}, cfg.task.action_key: {"mean": [0.0] * A, "std": [1.0] * A}}
        torch.manual_seed(0)
        pol = VitaPolicy(cfg, stats)
        pol.load_state_dict({k: v for k, v in sd.items()}, strict=False)
        tr = PolicyTrainer(cfg, pol, dev, num_training_steps=100000, group=group, rehearse_collectives=args.rehearse_dist and all_ranks)
        gt = torch.Generator().manual_seed(100 + rank)
        ncam = len(cfg.task.image_keys)
        tbatch = {cfg.task.state_key: torch.randn(TB, 1, S, generator=gt).to(dev),
                  "observation.features": torch.relu(torch.randn(TB, 1, 512 * ncam, generator=gt)).to(dev),
                  cfg.task.action_key: torch.randn(TB, 16, A, generator=gt).to(dev)}
        tts = torch.rand(64, TB, generator=gt).to(dev)
        tr.step_counter = 1000                           # past the lr warm-up, so the update is not a no-op
        for i in range(3 if TB <= 1024 else 6):          # (the large-batch legs' first steps run on cold workspace pages)
            tr.train_step(tbatch, t=tts[i % 64])
        fence() if all_ranks else torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(nts):
            tr.train_step(tbatch, t=tts[i % 64])
        fence() if all_ranks else torch.cuda.synchronize(dev)
        tel = time.perf_counter() - t0
        w_ = world if all_ranks else 1
        if dist is not None and all_ranks:
            tt = torch.tensor([tel], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tel = tt.item()
        m = tr.metrics()
        f_train = 3 * (2 * obs_dim * fw.D + 19_734_528 + (2 * fw.D * fw.H + fw.L * 4 * fw.H * fw.FF + 2 * fw.H * fw.D + 7_864_320)
                       + K * (2 * fw.D * fw.H + fw.L * 4 * fw.H * fw.FF + 2 * fw.H * fw.D)
                       + 2 * (2 * fw.D * dw.HD + dw.w.NL * 4 * dw.HD * dw.HD + 2 * dw.HD * dw.OUT)) if args.task == "hook_package" else None
        res = {"metric": "train samples/sec", "value": round(w_ * TB * nts / tel, 1), "unit": "samples/s",
               "ms_per_step": round(tel * 1e3 / nts, 4), "steps": nts, "batch_per_gpu": TB, "global_batch": TB * w_,
               "dtype": "f32" if gemm_dtype == "f32" else "bf16 GEMM operands (calls >= 2^31 FLOP), f32 accumulate / master weights / gradients",
               "loss": round(m["loss"], 6), "grad_norm": round(m["grad_norm"], 6),
               "grad_bytes_allreduced_per_step": int(tr.flat_grad.numel() * 4) if tr.reducer.active else 0,
               "parallelism": f"data-parallel x{w_}, {len(tr.plan)} event-driven gradient buckets over RCCL (decoder, encoders, flow net rest, one per layer)" if w_ > 1 else "single GPU",
               "includes": "normalize, fused forward+backward (1 FM eval + K Euler steps + encoder + 2 decodes), grad-norm clip, AdamW, EMA; observer excluded (camera latents are the input)",
               "algorithmic_flops_per_sample_fwd_bwd": f_train}
        if f_train:
            res["tflops"] = round(res["value"] * f_train / 1e12 / w_, 2)
            res["frac_of_f32_mfma_peak"] = round(res["tflops"] / PEAK_F32_MFMA_TFLOPS, 4)
        del tr, pol
        return res

    train = train_large = train_large_bf16 = None
    if not args.no_train and args.hidden != 512:
        print("bench.py: the training leg uses the default policy dims; skipped for --hidden != 512", file=sys.stderr)
    if not args.no_train and args.hidden == 512:
        try:
            train = train_leg(args.train_batch, "f32", args.train_steps or args.steps, True)
        except Exception as e:          # N > 1: the training leg is a side figure of the line; `value` must survive it
            if not multi:
                raise
            train = {"metric": "train samples/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        if multi and train.get("value") is not None:
            # the same step without the collective (each rank alone in a one-rank group): what the all-reduce leaves exposed
            try:
                solo_groups = [dist.new_group([r]) for r in range(world)]
                solo = train_leg(args.train_batch, "f32", args.train_steps or args.steps, False, group=solo_groups[rank])
                train["ms_per_step_without_allreduce"] = solo["ms_per_step"]
                train["exposed_communication_ms_per_step"] = round(train["ms_per_step"] - solo["ms_per_step"], 4)
            except Exception as e:      # a side figure must never cost the scaling line
                train["exposed_communication_ms_per_step"] = None
                train["exposed_communication_error"] = repr(e)[:200]
        if multi and args.train_large_batch > 0:
            # BASELINE configs[2]: bf16-operand training, data parallel — at a per-GPU batch whose GEMMs take the bf16 kernels (the
            # default 128 rows per GPU stay fp32: none of their GEMMs reaches the 2^31-FLOP switch)
            try:
                train_large_bf16 = train_leg(args.train_large_batch, "bf16", 10, True)
            except Exception as e:
                train_large_bf16 = {"metric": "train samples/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        if world == 1 and args.train_large_batch > 0:
            # side figures at a batch that fills the chip: fp32, and BASELINE configs[2]'s bf16-operand training
            train_large = train_leg(args.train_large_batch, "f32", 10, False)
            train_large_bf16 = train_leg(args.train_large_batch, "bf16", 10, False)
            train_large_bf16["loss_rel_diff_vs_f32"] = float(f"{abs(train_large_bf16['loss'] - train_large['loss']) / abs(train_large['loss']):.3g}")

    # ---- per-kernel roofline: HIP events around the dominant GEMM classes, on the launch stream
    def time_linear(M, N, Kd, reps=20, **kw):
        x = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) / Kd ** 0.5
        bias = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        extra = {}
        if kw.get("ln"):
            extra = dict(pro=_lib.PRO_LN_MOD, ln_eps=1e-6, pro_mul=torch.randn(Kd, device=dev) * 0.1,
                         pro_add=torch.randn(Kd, device=dev) * 0.1)
        if kw.get("gated"):
            extra.update(gate=torch.randn(N, device=dev), resid=y)
        for _ in range(3):
            ops.linear(x, w, bias, kw.get("act", 0), out=y, **extra)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.linear(x, w, bias, kw.get("act", 0), out=y, **extra)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    kernels = {}
    dom = None
    if rank == 0:
        for name, (M, N, Kd, kw) in {
            "fc1": (B, fw.FF, fw.H, dict(act=_lib.ACT_GELU_TANH)),
            "fc2": (B, fw.H, fw.FF, dict(gated=True)),
            "in_proj": (B, fw.H, fw.D, dict()),
        }.items():
            dt = time_linear(M, N, Kd, **kw)
            cfg_ = (lib.vita_debug_last_gemm_config() or b"").decode() or "not the LDS-DMA family"   # what the call dispatched
            epi = {"fc1": "bias+gelu_tanh", "fc2": "bias+gate+resid", "in_proj": "bias"}[name]
            label = f"{name} {cfg_} {M}x{N}x{Kd} {epi}"
            fl = 2.0 * M * N * Kd
            kernels[label] = {"avg_us": round(dt * 1e6, 2), "tflops": round(fl / dt / 1e12, 2),
                              "frac_of_f32_mfma_peak": round(fl / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
            if name == "fc1":
                dom = label

    # ---- HBM-bound kernels (loss / elementwise / optimizer): achieved GB/s of algorithmic bytes against the 8 TB/s HBM3E peak
    hbm = {}
    if rank == 0:
        def time_op(fn, reps=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps
        R, Dl, Hh = 7 * B, fw.D, fw.H          # rows of one stacked training pass (1 + K evaluations)
        xa, xb = torch.randn(R, Dl, device=dev), torch.randn(R, Dl, device=dev)
        tt = torch.rand(R, device=dev)
        one = torch.ones((), device=dev)
        hh, dyh = torch.randn(R, Hh, device=dev), torch.randn(R, Hh, device=dev)
        mul, add = torch.randn(Hh, device=dev) * 0.1, torch.randn(Hh, device=dev) * 0.1
        nparam = 19_925_328
        pp, gg, mm, vv, ee = (torch.randn(nparam, device=dev) * 0.01 for _ in range(5))
        vv.abs_()
        gsq = ops.sumsq(gg)
        cases = {
            f"mse_loss_fwd {R}x{Dl}": (lambda: ops.mse_loss_fwd(xa, xb), 8 * R * Dl),
            f"mse_loss_bwd {R}x{Dl} (one gradient)": (lambda: ops.mse_loss_bwd(xa, xb, one, need_b=False), 12 * R * Dl),
            f"cfm_interpolate {R}x{Dl}": (lambda: ops.cfm_interpolate(xa, xb, tt), 16 * R * Dl),
            f"layernorm_modulate fwd {R}x{Hh}": (lambda: ops.layernorm_modulate(hh, mul, add, 1e-6), 8 * R * Hh),
            f"layernorm_modulate bwd {R}x{Hh} (dx + dmul/dadd rows)": (lambda: ops.layernorm_modulate_bwd(hh, dyh, mul, 1e-6, _lib.PRO_LN_MOD), 24 * R * Hh),
            f"colsum {R}x{Hh}": (lambda: ops.colsum(hh), 4 * R * Hh),
            f"sumsq {nparam}": (lambda: ops.sumsq(gg), 4 * nparam),
            f"adamw_ema_step {nparam} (clip + AdamW + EMA, one pass)": (
                lambda: ops.adamw_ema_step(pp, gg, mm, vv, ee, step=10, lr=1e-4, gnorm_sq=gsq, max_norm=1.0, ema_decay=0.5), 36 * nparam),
        }
        for name, (fn, nbytes) in cases.items():
            dt_ = time_op(fn)
            hbm[name] = {"avg_us": round(dt_ * 1e6, 2), "algorithmic_bytes": nbytes, "GB_per_s": round(nbytes / dt_ / 1e9, 1),
                         "frac_of_hbm_peak": round(nbytes / dt_ / 8e12, 4)}
        del xa, xb, hh, dyh, pp, gg, mm, vv, ee
        # the observer's elementwise tail at the stem's shape for 64 frames of 224 x 308 (csrc/vita_observer.hip)
        sx = torch.randn(64, 64, 112, 154, device=dev)
        ssc, ssh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
        n_el = sx.numel()
        with torch.no_grad():
            for name, fn, nbytes in (
                    ("observer frozen-norm + ReLU 64x64x112x154", lambda: ops.channel_affine_relu(sx, ssc, ssh, None, True), 8 * n_el),
                    ("observer frozen-norm + ReLU + maxpool3s2 64x64x112x154 (in + pooled out + 1 B/window)",
                     lambda: ops.affine_relu_maxpool(sx, ssc, ssh), 4 * n_el + 5 * (n_el // 4))):
                dt_ = time_op(fn)
                hbm[name] = {"avg_us": round(dt_ * 1e6, 2), "algorithmic_bytes": nbytes, "GB_per_s": round(nbytes / dt_ / 1e9, 1),
                             "frac_of_hbm_peak": round(nbytes / dt_ / 8e12, 4)}
        del sx

    # HBM bytes per fc1 launch from the PMC passes (tools/profile.sh -> profiles/traffic.json): a committed measurement, not
    # taken in this run, so it is reported only for the shape it was taken on, with its source named
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            want = ["persistent", B, K, fw.H] if persistent else [B, fw.FF, fw.H]
            if tj.get("shape") == want:
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), f"{tj.get('source')} (PMC passes on {tj.get('kernel')}, shape {tj.get('shape')}, not this run)"
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, sd, obs, K, A, fpc)

    if rank == 0:
        if persistent:
            # the dominant kernel IS the solve: one persistent launch runs all K steps (K * (2 + 2L) GEMM phases)
            roofline = {"bound": "mfma",
                        "kernel": f"ode_persistent_kernel<false> B={B} K={K}: {K * (2 + 2 * fw.L)} phases, fc1 tiles <256,128,32,3>, input/fc2/output "
                                  f"projection tiles <64,128,64,3,KG=4>, teams of {fw.FF // 128} workgroups per 256 rows",
                        "achieved": round(lat_tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(lat_tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "flops_per_launch": float(B) * K * f_step, "avg_launch_us": round(lat_ms * 1e3, 2), "sampler_status": sampler_status,
                        "timing": "HIP events on the launch stream around back-to-back solves; one solve = the phase-table kernel, a memset, the "
                                  "persistent launch and the time path's ~8 tiny launches (all inside the interval); flops = the algorithmic "
                                  "x-dependent GEMM flops of SURVEY 8(d): B * K * F_step"}
        else:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": kernels[dom]["frac_of_f32_mfma_peak"], "traffic": traffic, "traffic_source": traffic_src,
                        "flops_per_launch": 2.0 * B * fw.FF * fw.H, "avg_launch_us": kernels[dom]["avg_us"],
                        "timing": "HIP events around 20 back-to-back launches on the launch stream"}
        line = {
            "metric": "ODE action-chunks/sec (bs 4096)",
            "value": round(chunks_per_s, 1),
            "unit": "action-chunks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.task} generate_actions: obs features (B,{obs_dim}) -> actions (B,16,{A})",
                       "batch_per_gpu": B, "global_batch": B * world, "ode_steps": K,
                       "latent_dim": fw.D, "hidden_dim": fw.H, "layers": fw.L, "mlp_ratio": fw.FF // fw.H,
                       "parallelism": f"batch-sharded x{world}, no collective"},
            "flops_per_chunk": fpc,
            "whole_path_tflops": round(chunks_per_s * fpc / 1e12 / world, 2),
            "whole_path_frac_of_f32_mfma_peak": round(chunks_per_s * fpc / 1e12 / world / PEAK_F32_MFMA_TFLOPS, 4),
            "latent_only": {"ms_per_solve": round(lat_ms, 4), "chunks_per_s": round(B / (lat_ms * 1e-3), 1),
                            "tflops": round(lat_tflops, 2),
                            "frac_of_f32_mfma_peak": round(lat_tflops / PEAK_F32_MFMA_TFLOPS, 4)},
            "roofline": roofline,
            "kernels": kernels,
            "hbm_bound_kernels": {"peak_GB_per_s": 8000, "note": "wrapper calls incl. their output allocation; algorithmic bytes = reads + writes "
                                  "of the operands once", "kernels": hbm},
            "train": train,
            "train_large_batch": train_large,
            "train_large_batch_bf16": train_large_bf16,
            "bf16_variant": bf16,
            "serving_latency": serving,
            "strong_scaling": strong,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
