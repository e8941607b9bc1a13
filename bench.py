#!/usr/bin/env python3
"""Headline benchmark: masked-pretraining samples/s on synthetic (neurons x timebins x modalities)
batches through the drop-in path (MultiModalTrainer step -> MultiModal.forward -> HIP engine ->
backward -> fused AdamW), BASELINE.json configs[1]: 1 session, spike + behaviour, d_model=256,
5+5 layers, T=100 bins per modality.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--dtype fp32|bf16]

N>1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL); every rank
processes its own batch of B samples (weak scaling) and gradients are mean-all-reduced in buckets
overlapped with backward.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}     # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("MMFM_BENCH_BATCH", "1024")), help="samples per GPU per step")
    ap.add_argument("--dtype", default=os.environ.get("MMFM_DTYPE", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="batches start in pinned HOST memory and cross PCIe inside the timed region (the PCIe-inclusive rate quoted in "
                         "DESIGN.md; never the headline `value`, which is measured with inputs resident in HBM)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--host-profile", default=None, help="write a cProfile summary of the timed loop's host side to this file (diagnostic)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short fp32 / exact-masker / multi-session / config-5 legs")
    ap.add_argument("--cpu-steps", type=int, default=12, help="CPU-baseline steps at B=16 (~1 s each on 16 threads: a 10-15 s bounded sample)")
    return ap.parse_args()


def flops_per_sample_fwd(cfg, channels, T):
    """MACs of one forward per sample (x2 = FLOP; a train step is 3x forward)."""
    H, I = cfg.hidden, cfg.inter
    M = len(channels)
    L = M * T
    mac = 0
    for n in channels:
        mac += 2 * T * (n * 2 * n + 2 * n * H) + T * H * n
    mac += cfg.n_enc * (L * (4 * H * H + 2 * H * I) + 2 * L * L * H)
    mac += L * H * H
    mac += cfg.n_dec * (L * (8 * H * H + 2 * H * I) + 4 * L * L * H)
    return 2 * mac


def kernel_profile(engine, plan, reps=3):
    """Per-launch timing of every entry of the step plan with HIP events on the launch stream.  The plan is replayed IN
    ORDER (forward, then backward), one event between consecutive launches, so every kernel sees the cache state it sees in
    the real step (timing each entry in a warm back-to-back loop read 8 % low against rocprofv3 on the GEMMs); the first pass
    is untimed."""
    st = torch.cuda.current_stream().cuda_stream
    entries = list(plan["fwd"]) + [e for _, seg in plan["bwd"] for e in seg]
    n = len(entries)
    acc = [0.0] * n
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for i, (fn, args, keep) in enumerate(entries):
            evs[i].record()
            fn(*args, st)
        evs[n].record()
        torch.cuda.synchronize()
        if rep:
            for i in range(n):
                acc[i] += evs[i].elapsed_time(evs[i + 1])
    rows = []
    for i, (fn, args, keep) in enumerate(entries):
        ms = acc[i] / reps
        name, flops, sub, nbytes = fn.__name__, 0.0, None, 0.0          # nbytes = ALGORITHMIC bytes (operands read once + results written once)
        if name == "mmfm_gemm":
            d = keep[0]
            flops = 2.0 * d.M * d.N * d.K
            sub = "x.W^T" if (d.a_kcontig and d.b_kcontig) else ("dY.W" if d.a_kcontig else "dY^T.X")
            nbytes = 2.0 * (d.M * d.K + d.N * d.K) + (4.0 if d.c_f32 else 2.0) * d.M * d.N * max(1, d.splits)
        elif name == "mmfm_gemm_pair":          # two weight gradients in one launch: one launch of the GEMM family
            name = "mmfm_gemm"
            flops = sum(2.0 * d.M * d.N * d.K for d in keep[:2])
            nbytes = sum(2.0 * (d.M * d.K + d.N * d.K) + 4.0 * d.M * d.N * max(1, d.splits) for d in keep[:2])
            sub = "dY^T.X"
        elif name == "mmfm_rowgemm":
            d = keep[0]
            flops = 2.0 * d.R * d.N * d.K
            sub = "row dX (+LN bwd)" if d.ln_bwd else ("row LN+x.W^T" if d.ln else ("row x.W^T" if d.bias else "row dY.W"))
            nbytes = 2.0 * d.R * (d.K + d.N) + 2.0 * d.N * d.K + (2.0 * d.R * d.N if d.residual else 0) + (2.0 * d.R * d.K if d.xhat else 0) \
                + (2.0 * d.R * d.N if d.ln_bwd else 0)
        elif name in ("mmfm_mlp_fwd", "mmfm_mlp_bwd"):
            d = keep[0]
            flops = 4.0 * d.R * 256 * 512           # algorithmic: two products each way (the backward's recompute of up() is not counted)
            sub = "row MLP fwd" if name.endswith("fwd") else "row MLP bwd (dX chain)"
            nbytes = 2.0 * d.R * 256 * (3 if name.endswith("fwd") else 4) + (2.0 * d.R * 512 * 2 if name.endswith("bwd") else 0)
            if name.endswith("bwd") and not d.dx:   # front half only (t1, g, du): ONE algorithmic product; d(x_hat) + LayerNorm backward is the
                flops = 2.0 * d.R * 256 * 512       # mmfm_rowgemm launch behind it, which counts its own flops and bytes (it re-reads du)
                sub = "row MLP bwd front half (t1, g, du)"
                nbytes = 2.0 * d.R * 256 * 3 + 2.0 * d.R * 512 * 2
        elif name in ("mmfm_attn_fwd", "mmfm_attn_bwd"):
            d = keep[0]
            flops = (4.0 if name.endswith("fwd") else 10.0) * d.B * d.heads * d.Lq * d.Lk * d.dh
            hd = d.heads * d.dh
            nbytes = 2.0 * d.B * hd * ((d.Lq + 2 * d.Lk + d.Lq) if name.endswith("fwd") else (3 * d.Lq + 2 * d.Lk + d.Lq + 2 * d.Lk)) \
                + (d.B * d.heads * ((d.Lq + 31) // 32) * ((d.Lk + 31) // 32) * 128 * (2 if name.endswith("fwd") else 1) if d.keepbits else 0)
        rows.append((name, ms, flops, sub, nbytes))
    agg, subs = {}, {}
    for name, ms, fl, sub, nb in rows:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fl; a[3] += nb
        if sub:
            b = subs.setdefault(sub, [0, 0.0, 0.0, 0.0])
            b[0] += 1; b[1] += ms; b[2] += fl; b[3] += nb
    return agg, subs


def cpu_baseline(steps, B=16):
    """The CPU oracle (oracle/mm_oracle.py, pinned against the reference's fixtures) on the host cores:
    the same synthetic step (forward + autograd backward + AdamW), dropout as configured, B=16
    (the reference's batch size).  A reported baseline, not the target."""
    from oracle import mm_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MMFM_CPU_THREADS", "16"))))     # the 1-GPU box's CPU share is 16
    torch.set_num_threads(cores)
    mk = dict(force_active=True, mode="temporal", ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0, max_timespan=1,
              channels=None, timesteps=None, mask_regions=["all"], target_regions=["all"], n_mask_regions=1, causal_zero=True)

    def run(cfg, n):
        sd = O.init_state_dict(cfg, seed=42)
        tr = O.OracleTrainer(sd, cfg, dict(mk), total_steps=1000)
        objs = O.objective_schedule(n + 1)
        batches = [O.synth_batch(B, 100, 668, 2, seed=s) for s in range(n + 1)]
        tr.step(batches[0], objs[0])                       # warm-up
        t0 = time.perf_counter()
        for s in range(1, n + 1):
            tr.step(batches[s], objs[s])
        return time.perf_counter() - t0

    dt = run(O.OracleCfg(), steps)
    n0 = max(2, steps // 3)
    dt0 = run(O.OracleCfg(embed_dropout=0.0, dropout=0.0), n0)      # the CPU step is RNG-dominated with dropout on
    return dict(value=round(B * steps / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} train steps (fwd+bwd+AdamW) at B={B}, T=100, 668+2 channels, fp32 torch-CPU oracle, dropout as configured, "
                       f"{dt:.1f} s of CPU work (+ {n0} steps with dropout 0: {dt0:.1f} s)",
                value_no_dropout=round(B * n0 / dt0, 3))


def src_sha():
    """Hash of the kernel sources: profiles/pmc_traffic.json records it, so a traffic figure measured on an older library is not
    attached to this run's roofline line."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "multi_modal_foundation_model_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def timed_leg(step, warm, steps):
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def leg_trainer(dev, dtype, B, exact_masker, sessions=None, steps=8, warm=3):
    """A short timed leg of the same trainer step with its own model: other precision (fp32 parity), the default masker stream,
    or multi-session batches (BASELINE configs[2]: neurons right-padded with -1 to 668, one session per batch)."""
    from torch.optim.lr_scheduler import OneCycleLR
    from multi_modal_foundation_model_amd.builders import build_model, load_config
    from multi_modal_foundation_model_amd.optim import make_optimizer
    from multi_modal_foundation_model_amd.synthetic import synth_batch
    from trainer.make import make_multimodal_trainer
    import numpy as np
    cfg = load_config()
    model = build_model(cfg.model, 668, 2, seed=cfg.seed)
    model.compute_dtype = dtype
    model.engine_seed = 99
    cfg["training"]["exact_masker_stream"] = bool(exact_masker)      # default trainer behaviour: token-mask-only stream (trainer/base.py)
    model = model.to(dev).train()
    opt = make_optimizer(model, lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.wd, eps=cfg.optimizer.eps)
    sch = OneCycleLR(optimizer=opt, total_steps=1000, max_lr=cfg.optimizer.lr, pct_start=cfg.optimizer.warmup_pct, div_factor=cfg.optimizer.div_factor)

    class Acc:
        device = dev
    tr = make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=opt, log_dir="/tmp", accelerator=Acc(),
                                 lr_scheduler=sch, avail_mod=["ap", "behavior"], config=cfg,
                                 modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]), mixed_training=True, num_neurons=[668])
    pool = []
    n_pool = 4 if sessions is None else sessions
    rng = np.random.default_rng(2024)
    for i in range(n_pool):
        b = synth_batch(B, 100, 668, 2, seed=7000 + i)
        if sessions is not None:                          # session i has n_i neurons; the loader pads the rest with -1 (loader/base.py:407-425)
            n_i = int(rng.integers(300, 669))
            b["spikes_data"][:, :, n_i:] = -1.0
            b["space_attn_mask"][:, n_i:] = 0
            b["eid"] = [f"session{i}"] * B
        pool.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()})

    def step(i):
        tr._sample_modes()
        out = tr._forward_model_outputs(dict(pool[i % len(pool)]), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        return out.loss
    dt = timed_leg(step, warm, steps)
    del model, opt, tr, pool
    torch.cuda.empty_cache()
    return dt


def leg_config5(dev, B=256, steps=4, warm=2):
    """BASELINE configs[4] as SURVEY.md §8d instantiates it: H=512, I=1024, dh=64, T=200, ap + behavior + lfp (L=600), bf16, dropout on."""
    import numpy as np
    from multi_modal_foundation_model_amd.builders import build_model_mods, make_optimizer, model_config
    mods = [("ap", 668), ("behavior", 2), ("lfp", 128)]
    model = build_model_mods(model_config(H=512, heads=8, inter=1024, max_F=200, n_modality=3), mods, seed=42)
    model.loss_mod["lfp"] = "mse"
    model.compute_dtype = "bf16"
    model = model.to(dev).train()
    opt, sch = make_optimizer(model, 1000)
    g = torch.Generator().manual_seed(0)
    T = 200
    attn = torch.ones(B, T, dtype=torch.int64, device=dev)
    ts = torch.arange(T, dtype=torch.int64)[None].repeat(B, 1).to(dev)
    base = {}
    for i, (name, n) in enumerate(mods):
        x = (torch.poisson(torch.full((B, T, n), 0.3), generator=g) if name == "ap" else torch.randn(B, T, n, generator=g)).to(dev)
        idx = torch.tensor(i, device=dev)
        base[name] = dict(inputs_modality=idx, targets_modality=idx, inputs_attn_mask=attn, inputs_timestamp=ts, targets_timestamp=ts,
                          masking_mode=None, inputs=x, targets=x,
                          eval_mask=torch.full((1, 1, 1), 1 if name == "ap" else 0, dtype=torch.int64, device=dev).expand(B, T, n))
        if name == "ap":
            base[name]["inputs_regions"] = np.full((B, n), "XX")

    def step(i):
        out = model({m: dict(x) for m, x in base.items()})
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        return out.loss
    dt = timed_leg(step, warm, steps)
    fl = 3 * flops_per_sample_fwd(model._engine.cfg, [n for _, n in mods], T) * B
    del model, opt
    torch.cuda.empty_cache()
    return dt, fl


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    try:
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    except AttributeError:
        pass
    # rehearsal switch for a one-GPU box: every rank on cuda:0 and gloo instead of RCCL (which needs one GPU per rank), so the
    # N > 1 control flow of this script (barriers, max-over-ranks timing, rank-0 JSON) can be exercised before an 8-GPU run
    rehearse = os.environ.get("MMFM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from multi_modal_foundation_model_amd.builders import build_model, load_config
    from torch.optim.lr_scheduler import OneCycleLR
    from multi_modal_foundation_model_amd.ddp import DataParallelModel
    from multi_modal_foundation_model_amd.optim import make_optimizer
    from multi_modal_foundation_model_amd.synthetic import synth_batch
    from trainer.make import make_multimodal_trainer

    cfg = load_config()
    B, T, n_ap, n_beh = a.batch, 100, 668, 2
    model = build_model(cfg.model, n_ap, n_beh, seed=cfg.seed)          # same init on every rank (set_seed(42) upstream)
    model.compute_dtype = a.dtype
    model.engine_seed = 1234 + rank
    model = model.to(dev)       # the trainer (default config) switches the masker to its token-mask-only host stream (trainer/base.py)
    if world > 1:
        model = DataParallelModel(model)
    total = max(1000, a.steps + a.warmup + 1)
    opt = make_optimizer(model, lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.wd, eps=cfg.optimizer.eps)
    sch = OneCycleLR(optimizer=opt, total_steps=total, max_lr=cfg.optimizer.lr, pct_start=cfg.optimizer.warmup_pct,
                     div_factor=cfg.optimizer.div_factor)

    class Acc:
        device = dev
    tr = make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=opt, log_dir="/tmp", accelerator=Acc(),
                                 lr_scheduler=sch, avail_mod=["ap", "behavior"], config=cfg,
                                 modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]), mixed_training=True,
                                 num_neurons=[n_ap])
    # synthetic batches, resident in HBM before the timed region (pool of 4 per rank, cycled)
    pool = []
    for i in range(4):
        b = synth_batch(B, T, n_ap, n_beh, seed=1000 * rank + i)
        if a.host_inputs:
            pool.append({k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
        else:
            pool.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    random.seed(42)                                   # same objective on every rank, different data/masks
    torch.manual_seed(4242 + rank)
    model.train()

    def step(i):
        tr._sample_modes()
        out = tr._forward_model_outputs(dict(pool[i % len(pool)]), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
        out.loss.backward()
        opt.step()
        sch.step()
        opt.zero_grad()
        return out.loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"model built ({a.dtype}, B={B}/GPU, world={world}); warm-up {a.warmup} steps")
    for i in range(a.warmup):
        loss = step(i)
    barrier()
    if rank == 0:
        log(f"timing {a.steps} steps")
    prof = None
    if a.host_profile and rank == 0:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    if prof is not None:
        prof.disable()
        import io, pstats
        buf_ = io.StringIO()
        pstats.Stats(prof, stream=buf_).sort_stats("cumulative").print_stats(70)
        with open(a.host_profile, "w") as fh:
            fh.write(buf_.getvalue())
    dt_host = time.perf_counter() - t0         # the host thread is done enqueueing: close to dt = the step is host- (dispatch-) bound
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    last_loss = loss.item()

    inner = getattr(model, "module", model)
    eng = inner._engine
    res = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        fl_step = 3 * flops_per_sample_fwd(eng.cfg, [n_ap, n_beh], T) * B
        res = dict(metric="pretrain samples/sec (T=100 bins/modality, d_model=256, 5+5 layers, spike+behaviour masked pretraining step)",
                   value=round(value, 2), unit="samples/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.dtype, data="synthetic",
                   host_enqueue_ms_per_step=round(dt_host / a.steps * 1e3, 3),
                   config=dict(workload="BASELINE.json configs[1]: 1 session, ap(668 neurons)+behaviour(2), d_model=256, heads=8, mlp=512, "
                                        "5 enc + 5 dec layers, T=100 (L=200 tokens), dropout 0.4/0.2, AdamW+OneCycleLR, mixed objectives",
                               per_gpu_batch=B, global_batch=B * world, seq_len=200, parallelism=f"dp{world}"),
                   samples_per_sec_per_gpu=round(value / world, 2), final_loss=round(last_loss, 5),
                   model_tflops_per_gpu=round(fl_step / (dt / a.steps) / 1e12, 2),
                   frac_of_mfma_peak_whole_step=round(fl_step / (dt / a.steps) / 1e12 / PEAK_TFLOPS[a.dtype], 4))
        res["inputs"] = "host (pinned), PCIe inside the timed region" if a.host_inputs else "resident in HBM"
        # the trainer's DEFAULT for mask_type 'embd' (trainer/base.py): token-mask-only host stream; `value_exact_masker` below is the
        # same step under training.exact_masker_stream (the reference's CPU generator walk, host-bound)
        res["masker_token_mask_only"] = bool(getattr(model.masker, "token_mask_only", False))
        res["fused_mask"] = eng._fused_mask(B * 200)
        log(f"{ms:.2f} ms/step, {value:.1f} samples/s")
        if not a.no_kernel_profile:
            log("per-kernel HIP-event profile")
            agg, subs = kernel_profile(eng, eng._last)
            tot = sum(v[1] for v in agg.values())
            top = sorted(agg.items(), key=lambda kv: -kv[1][1])
            res["kernel_breakdown_ms"] = {k: round(v[1], 3) for k, v in top[:8]}
            res["gemm_layouts"] = {k: dict(launches=v[0], ms=round(v[1], 3), tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                                           gbs_algorithmic=round(v[3] / (v[1] * 1e-3) / 1e9, 1)) for k, v in subs.items()}
            res["kernel_time_sum_ms"] = round(tot, 3)
            # Kernel families of the step, each against the roof that bounds it (arithmetic intensity against the
            # ridge 2500 TF / 8 TB/s = 312 flop/B decides between MFMA and HBM; attention is bound by vector-instruction issue).
            # `traffic` = HBM bytes by PMC (profiles/pmc_traffic.json, only when taken on THIS library) where available, else the
            # algorithmic bytes (operands read once, results written once).
            tj = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("src_sha") != src_sha():
                    tj = None
            RIDGE = PEAK_TFLOPS[a.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)

            def family(label, names, sub_names=None, valu_slots=None):
                if sub_names is not None:
                    parts = [subs[n] for n in sub_names if n in subs]
                else:
                    parts = [agg[n] for n in names if n in agg]
                if not parts:
                    return None
                n, ms, fl, nb = (sum(x[i] for x in parts) for i in range(4))
                pmc = None
                if tj is not None and sub_names is None and all(k in tj for k in names if k in agg):
                    pmc = sum(tj[k] * agg[k][0] for k in names if k in agg)
                byts = nb                      # `hbm_gbs` / `hbm_frac` are ALGORITHMIC bytes over time; the PMC figure rides along as `traffic_*`
                tf, gbs = fl / (ms * 1e-3) / 1e12, byts / (ms * 1e-3) / 1e9
                out = dict(family=label, launches_per_step=n, ms_per_step=round(ms, 3), tflops=round(tf, 1), mfma_frac=round(tf / PEAK_TFLOPS[a.dtype], 4),
                           hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4), bytes_per_launch=int(byts / n),
                           bytes_source="algorithmic", flop_per_byte=round(fl / max(byts, 1.0), 1))
                if pmc is not None:
                    out.update(traffic_bytes_per_launch=int(pmc / n), traffic_over_algorithmic=round(pmc / max(nb, 1.0), 3))
                out["bound"] = "mfma" if fl / max(byts, 1.0) >= RIDGE else "hbm"
                if valu_slots is not None:
                    # issue floor: vector issue slots per score element x elements, one slot = 4 cycles of one of the chip's 1,024 SIMDs
                    floor_ms = valu_slots / (1024 * 2.4e9 / 4) * 1e3
                    out.update(bound="valu-issue", valu_issue_floor_ms=round(floor_ms, 3), valu_issue_frac=round(floor_ms / ms, 4))
                return out

            FAM = ("mmfm_gemm", "mmfm_rowgemm", "mmfm_mlp_fwd", "mmfm_mlp_bwd")
            res["family_ms"] = {n: round(agg[n][1], 3) for n in FAM if n in agg}
            att_slots = 0.0
            for fn, args, keep in list(eng._last["fwd"]) + [e for _, seg in eng._last["bwd"] for e in seg]:
                if fn.__name__ in ("mmfm_attn_fwd", "mmfm_attn_bwd"):
                    d = keep[0]
                    # measured instruction mix of the dh = 32 keep-bit kernels (csrc/attention_fast.hip): 6 issue slots per score element in
                    # the forward (fma, exp x2, add, select, half a pack + the MFMAs' issue share) + 3.6 for its decision in the generator
                    # (114 instructions per 32-decision word), 9 in the backward; 64 elements per slot
                    per = (6.0 + (3.6 if (d.keepbits and d.drop_p.p > 0) else 0.0)) if fn.__name__.endswith("fwd") else 9.0
                    att_slots += per * d.B * d.heads * d.Lq * d.Lk / 64.0
            fams = [family("dense linears (x.W^T, dY.W, row-owner LN / MLP kernels)", FAM, sub_names=[k for k in subs if k != "dY^T.X"]),
                    family("weight gradients (dY^T.X, streaming split-K)", ("mmfm_gemm",), sub_names=["dY^T.X"]),
                    family("attention (dh 32 keep-bit kernels)", ("mmfm_attn_fwd", "mmfm_attn_bwd"), valu_slots=att_slots)]
            res["roofline_families"] = [f for f in fams if f]
            fam = [agg[n] for n in FAM if n in agg]
            v = [sum(x[i] for x in fam) for i in range(4)]
            k = "dense linears: " + "+".join(n for n in FAM if n in agg)
            traffic = None
            if tj is not None and all(n in tj for n in FAM if n in agg):
                traffic = int(sum(tj[n] * agg[n][0] for n in FAM if n in agg) / max(1, v[0]))
            byts = v[3]                        # achieved = ALGORITHMIC bytes per launch / average launch time; `traffic` = HBM bytes by PMC, beside it
            tf, gbs = v[2] / (v[1] * 1e-3) / 1e12, byts / (v[1] * 1e-3) / 1e9
            hbm_bound = v[2] / max(byts, 1.0) < RIDGE
            # the dominant family against ITS roof: HBM when its arithmetic intensity sits under the ridge (it does: ~140 flop/B)
            res["roofline"] = dict(bound="hbm" if hbm_bound else "mfma", kernel=k, launches_per_step=v[0],
                                   achieved=round(gbs if hbm_bound else tf, 2), peak=HBM_PEAK_GBS if hbm_bound else PEAK_TFLOPS[a.dtype],
                                   unit="GB/s" if hbm_bound else "TFLOP/s",
                                   frac=round((gbs / HBM_PEAK_GBS) if hbm_bound else (tf / PEAK_TFLOPS[a.dtype]), 4),
                                   hbm_frac=round(gbs / HBM_PEAK_GBS, 4), mfma_frac=round(tf / PEAK_TFLOPS[a.dtype], 4),
                                   traffic=traffic, avg_launch_ms=round(v[1] / v[0], 4), algorithmic_flops_per_launch=v[2] / v[0],
                                   algorithmic_bytes_per_launch=v[3] / v[0], flop_per_byte=round(v[2] / max(byts, 1.0), 1),
                                   ridge_flop_per_byte=round(RIDGE, 1))
        if world == 1 and not a.no_extra_legs:
            try:
                log("extra legs: exact masker stream, fp32 parity mode, multi-session, config 5")
                dt = leg_trainer(dev, a.dtype, B, exact_masker=True, steps=6)
                res["value_exact_masker"] = round(B / dt, 2)
                Bf = min(B, 256)
                dt = leg_trainer(dev, "fp32", Bf, exact_masker=False, steps=4, warm=2)
                fl32 = 3 * flops_per_sample_fwd(eng.cfg, [n_ap, n_beh], T) * Bf
                res["fp32_parity"] = dict(samples_per_sec=round(Bf / dt, 2), ms_per_step=round(dt * 1e3, 3), per_gpu_batch=Bf,
                                          frac_of_fp32_mfma_peak=round(fl32 / dt / 1e12 / PEAK_TFLOPS["fp32"], 4))
                dt = leg_trainer(dev, a.dtype, B, exact_masker=False, sessions=40, steps=8)
                dt5, fl5 = leg_config5(dev, B=min(B, 256))
                res["configs"] = dict(
                    multi_session=dict(samples_per_sec=round(B / dt, 2), ms_per_step=round(dt * 1e3, 3), sessions=40,
                                       note="BASELINE configs[2]: one session per batch, 300-668 neurons right-padded with -1 to 668"),
                    config5=dict(samples_per_sec=round(min(B, 256) / dt5, 2), ms_per_step=round(dt5 * 1e3, 3), per_gpu_batch=min(B, 256),
                                 model_tflops=round(fl5 / dt5 / 1e12, 1),
                                 note="BASELINE configs[4] on one GPU: H=512, I=1024, dh=64, T=200, ap+behavior+lfp (L=600), bf16"))
            except Exception as e:                      # the extra legs never take the headline down with them
                res["extra_legs_error"] = repr(e)[:300]
        if world == 1 and not a.no_cpu_baseline:
            log("CPU baseline (oracle on host cores)")
            res["cpu_baseline"] = cpu_baseline(a.cpu_steps)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
