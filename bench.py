#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec of the full train2 step (encoders + decoder +
discriminator + all losses + grad clip + Adam, D phase and G phase) on 4 s @ 22.05 kHz
piano/violin pairs, B=8 clips per GPU (BASELINE.json configs[1]; weak scaling to configs[2]).

    python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  Inputs are synthetic and resident in HBM before the timed
region.  `roofline` is measured live with device events around every launch of the dominant
GEMM kernel; `cpu_baseline` times the CPU oracle (a port of the reference step) on the host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

CLIP_SECONDS = 4.0
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def granted_cores():
    """Host cores this process may actually use: the scheduler affinity mask, cut by the cgroup CPU quota when one is set
    (the GPU box hands a one-GPU job a share of a 64-core part; os.cpu_count() reports the whole machine)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    n = min(aff, quota) if quota else aff
    return max(1, n), aff, quota


def _set_cpu_threads():
    """Threads for the CPU legs: every granted core, at most 32 (the oracle's conv kernels stop scaling there)."""
    n, aff, quota = granted_cores()
    torch.set_num_threads(max(1, min(n, 32)))
    return {"cores": torch.get_num_threads(), "affinity_cores": aff, "cgroup_quota_cores": quota, "os_cpu_count": os.cpu_count()}


def cpu_baseline(B=8, S=2, warm=3, timed=10, budget_s=75.0):
    """the oracle's restatement of the SAME step (D phase + G phase + clip + Adam, dropout on) in fp32 on the
    host cores at the benchmarked batch (B=8, S=2), 3 warm-up + 10 timed steps, median; and the front-end oracle
    (STFT + CQT + z-score + sectioning of the B clips), so the rate is quoted with and without it.  Bounded: stops timing
    early when the budget is spent (the number of steps actually timed is reported)."""
    from oracle import cqt_oracle as CO
    from oracle import frontend_oracle as FO
    from oracle import seeded_params as sp
    from oracle.train_step import OracleTrainer
    cores = _set_cpu_threads()               # the affinity / quota actually granted, not an assumed 16
    ot = OracleTrainer(p_drop=0.1)
    x, labels = sp.seeded_input(B, S), sp.balanced_labels(B)
    times, t_start = [], time.perf_counter()
    for it in range(warm + timed):
        t0 = time.perf_counter()
        ot.step(x, labels)
        dt = time.perf_counter() - t0
        if it >= warm:
            times.append(dt)
        print(f"[cpu_baseline] step {it}: {dt:.2f} s", file=sys.stderr, flush=True)
        if time.perf_counter() - t_start > budget_s and len(times) >= 3:
            break
    t_med, t_min = sorted(times)[len(times) // 2], min(times)
    # front end of the same B clips (numpy, one thread: librosa's CQT recursion restated step by step)
    waves = [FO.synth_waveform(i, "piano" if i < B // 2 else "violin", seconds=CLIP_SECONDS) for i in range(B)]
    fe = []
    for rep in range(2):
        t0 = time.perf_counter()
        for w in waves:
            spec = np.concatenate([FO.stft(w), CO.get_cqt(w)], axis=2)
            FO.overlap_windows(spec)
        fe.append(time.perf_counter() - t0)
    t_fe = min(fe)
    print(f"[cpu_baseline] front end of {B} clips: {t_fe:.2f} s", file=sys.stderr, flush=True)
    return {"value": B * CLIP_SECONDS / (t_med + t_fe), "unit": "audio-seconds/sec", **cores, "kind": "port",
            "cpu_model": _cpu_model(), "value_model_only": B * CLIP_SECONDS / t_med, "s_per_step_median": t_med, "s_per_step_min": t_min,
            "s_front_end": t_fe, "steps_timed": len(times),
            "sample": f"oracle fp32 step (D+G phases, clip, Adam, dropout 0.1) at the benchmarked B={B} S={S}: {warm} warm-up + {len(times)} timed steps, "
                      f"median {t_med:.2f} s (min {t_min:.2f}); + oracle STFT/CQT front end of the {B} clips {t_fe:.2f} s (single-thread numpy); "
                      "`value` includes the front end as the GPU step does, `value_model_only` does not"}


def _newest_traffic_file():
    """profiles/r<NN>/*pmc_traffic*.json of the highest round (written by tools/pmc_traffic.sh + .py from rocprofv3 --pmc passes)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*", "*pmc_traffic*.json")):
        m = re.search(r"profiles[/\\]r(\d+)", f)
        key = (int(m.group(1)) if m else -1, os.path.getmtime(f))
        if best is None or key > best[0]:
            best = (key, f)
    return best[1] if best else None


def kernel_roofline(trainer, x, labels, dtype_name):
    """Eager (un-graphed) steps with device events around every GEMM launch (ops.PROFILE) and around the HBM-bound kernel
    families (_lib.PROFILE_CALLS); aggregates per kernel configuration.  `roofline` = the GEMM configuration with the
    largest total time; `roofline.kernels[]` = the streaming kernels against the 8 TB/s HBM peak."""
    from ast_amd import _lib, ops
    ops.PROFILE, _lib.PROFILE_CALLS = [], []
    was = trainer.cfg.use_graph
    trainer.cfg.use_graph = False
    nsteps = 2
    for _ in range(nsteps):
        trainer.step(x, labels)
    torch.cuda.synchronize()
    trainer.cfg.use_graph = was
    recs, ops.PROFILE = ops.PROFILE, None
    calls, _lib.PROFILE_CALLS = _lib.PROFILE_CALLS, None
    agg = {}
    for name, flops, nbytes, e0, e1 in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += flops; a[3] += nbytes
    total_t = sum(a[1] for a in agg.values())
    # HBM-bound kernel families (norm passes, recon loss, Adam, pack / flush, layout): they compete for "dominant kernel" with the
    # GEMM configurations on total time per step (a streaming family has no flops: it is priced against HBM only)
    fam = {}
    for cname, nbytes, e0, e1 in calls:
        a = fam.setdefault(cname, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += nbytes
    name, (cnt, t, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
    fam_top = max(fam.items(), key=lambda kv: kv[1][1]) if fam else None
    if fam_top is not None and fam_top[1][1] > t:
        name, (cnt, t, by), fl = fam_top[0].replace("ast_", "") + "_kernel", fam_top[1], 0.0
    tf = fl / t / 1e12
    gbs = by / t / 1e9
    peak = PEAK_TFLOPS[dtype_name]
    intensity = fl / max(by, 1.0)
    balance = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    if intensity >= balance:
        roof = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak}
    else:
        roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    traffic, traffic_src = None, None
    pmc = _newest_traffic_file()
    if pmc:
        # HBM bytes per launch from rocprofv3 PMC passes over this same workload (tools/pmc_traffic.sh + .py: separate
        # FETCH_SIZE / WRITE_SIZE runs, KiB units, gfx950 x2 fetch correction); counters cannot be read in-process
        table = json.load(open(pmc))
        meta = table.get("_meta", {})
        rec = table.get(name)
        rel = os.path.relpath(pmc, ROOT)
        if rec:
            traffic = rec["traffic_bytes"]
            traffic_src = f"{rel} (git {meta.get('git_head', 'unknown')}; 2*FETCH_SIZE + WRITE_SIZE, KiB->bytes, mean per launch)"
        else:
            traffic_src = f"{rel} holds no entry for kernel '{name}' (kernel names changed since it was collected: re-run tools/pmc_traffic.sh)"
            print(f"warning: {traffic_src}", file=sys.stderr)
    roof.update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "kernel": name, "launches_per_step": cnt // nsteps, "avg_launch_us": t / cnt * 1e6,
                 "algorithmic_flops_per_launch": fl / cnt, "algorithmic_bytes_per_launch": by / cnt,
                 "tflops": tf, "frac_of_mfma_peak": tf / peak, "gbs": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                 "gemm_time_per_step_ms": total_t / nsteps * 1e3,
                 "all_gemm_tflops": sum(a[2] for a in agg.values()) / total_t / 1e12})
    # the three largest GEMM configurations by time, for the record
    roof["gemm_top"] = [{"kernel": k, "launches_per_step": v[0] // nsteps, "avg_launch_us": v[1] / v[0] * 1e6, "tflops": v[2] / v[1] / 1e12,
                         "frac_of_mfma_peak": v[2] / v[1] / 1e12 / peak, "gbs": v[3] / v[1] / 1e9}
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:5]]
    # HBM-bound kernel families: algorithmic bytes (operands + results once) / event time, against 8 TB/s
    roof["kernels"] = [{"kernel": k.replace("ast_", ""), "bound": "hbm", "launches_per_step": v[0] // nsteps, "time_per_step_us": v[1] / nsteps * 1e6,
                        "achieved": v[2] / v[1] / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": v[2] / v[1] / 1e9 / PEAK_HBM_GBS}
                       for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]) if v[1] > 0]
    return roof


def ar_decode_bench(tr, x, labels, S, iters=20, cpu_reps=3, eager=True):
    """BASELINE configs[3]: the reference's process_audio (evaluation_style_transfer.py:135-159) for a batch of clips:
    eval-mode content encoder + autoregressive new_decoder generation of S sections + overlap-average + iSTFT ->
    waveforms, as one replayed hipGraph (and eagerly), with the reference's recompute loop and with the KV-cached decode;
    next to it the CPU oracle's timing of the same pipeline ("frames/sec vs CPU").  Class embeddings come from one
    style-encoder pass beforehand, as the reference precomputes them."""
    from ast_amd import infer
    for m in (tr.style, tr.content, tr.decoder):
        m.eval()
    with torch.no_grad():
        _, ce = tr.style(x, labels)
    cls = ce[labels.to(x.device)].contiguous()
    res = {}
    for mode in ("recompute", "kv_cache"):
        type(tr.decoder).decode_mode = mode
        for name, use_graph in ((("graph", True), ("eager", False)) if eager else (("graph", True),)):
            sess = infer.StyleTransferSession(tr.content, tr.decoder, use_graph=use_graph)
            for _ in range(3):
                sess(x, cls)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                sess(x, cls)
            torch.cuda.synchronize()
            res[mode, name] = (time.perf_counter() - t0) / iters
    type(tr.decoder).decode_mode = "recompute"
    for m in (tr.style, tr.content, tr.decoder):
        m.train()
    B = x.shape[0]
    clip_s = {1: 3.0, 2: 4.0, 3: 8.0, 4: 10.0}[S]
    frames = B * (191 * (S - 1) + 287)
    dt = min(res["recompute", "graph"], res["kv_cache", "graph"])
    out = {"ms_per_batch": dt * 1e3, "ms_per_batch_recompute_graph": res["recompute", "graph"] * 1e3, "ms_per_batch_kv_cache_graph": res["kv_cache", "graph"] * 1e3,
           **({"ms_per_batch_recompute_eager": res["recompute", "eager"] * 1e3, "ms_per_batch_kv_cache_eager": res["kv_cache", "eager"] * 1e3} if eager else {}),
           "stft_frames_per_s": frames / dt, "audio_seconds_per_s": B * clip_s / dt, "batch": B, "sections": S,
           "note": "content encoder + autoregressive decoder (new_decoder.py:272-319) + overlap-average + iSTFT to waveforms, one hipGraph; "
                   "recompute = the reference's O(S^2) loop, kv_cache = per-layer cached K/V (same results)"}
    out["cpu"] = ar_decode_cpu(B, S, frames, clip_s, reps=cpu_reps)
    out["speedup_vs_cpu"] = out["stft_frames_per_s"] / out["cpu"]["stft_frames_per_s"]
    return out


def ar_decode_cpu(B, S, frames, clip_s, reps=3):
    """The oracle's eval-mode pipeline on the host cores: content encoder + AR decode + overlap-average + iSTFT."""
    from oracle import ast_oracle as O
    from oracle import frontend_oracle as FO
    from oracle import layout as OL
    from oracle import seeded_params as sp
    cores = _set_cpu_threads()
    sds = {t: OL.seeded_model_state(t, requires_grad=False) for t in ("content", "decoder")}
    cfg = O.Cfg(training=False)
    x = sp.seeded_input(B, S)
    cls = sp.seeded_normal((B, 256), 5)
    times = []
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            content = O.content_encoder_forward(sds["content"], x, cfg)
            out = O.decoder_forward(sds["decoder"], content, cls, cfg, target_length=S).numpy()
            for b in range(B):
                FO.istft(FO.sections_to_spectrogram(out[b], 191 * (S - 1) + 287, 96))
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"s_per_batch": t, "stft_frames_per_s": frames / t, "audio_seconds_per_s": B * clip_s / t, **cores,
            "cpu_model": _cpu_model(), "kind": "port", "sample": f"oracle eval pipeline, B={B} S={S}, median of {reps}"}


def mixed_bench(args, tr, dev, rank, world, barrier, steps=None, warmup=None):
    """BASELINE configs[4]: variable-length curriculum batches.  Seven length buckets (2..8 s); every step takes one bucket's
    batch of B clips (resident waveforms -> STFT + CQT front end -> step); each bucket owns its front-end buffers and its
    captured graph, so after the first visit every step is a replay.  value = true audio seconds / wall time."""
    from ast_amd import train
    from ast_amd.dataloader import sections_for_samples
    lengths = [2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    buckets = []
    for i, sec in enumerate(lengths):
        waves, x, mean, std, labels = train.synthetic_waveform_batch(args.batch, sec, dev, seed=1000 + 17 * i + rank)
        assert x.shape[1] == sections_for_samples(waves.shape[1])
        buckets.append((sec, waves, x, mean, std, labels))
    cq = (torch.zeros(2, 84, device=dev), torch.full((2, 84), 0.25, device=dev))

    def run(k):
        sec, waves, x, mean, std, labels = buckets[k % len(buckets)]
        tr.set_frontend(waves, mean, std, *cq)
        tr.step(x, labels)
        return sec
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    g0 = len(tr._graphs)
    for k in range(max(warmup, len(buckets))):         # every bucket captured once before timing
        run(k)
    barrier()
    t0 = time.perf_counter()
    secs = sum(run(k) for k in range(steps))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return {"metric": "audio-seconds/sec/node (train step, mixed 2-8 s clips, bucketed by length)", "value": world * args.batch * secs / dt,
            "unit": "audio-seconds/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[4]: clips of {lengths} s (S = {[b[2].shape[1] for b in buckets]}), batch={args.batch} per step and GPU, "
                                   "one length bucket per step in rotation, STFT + CQT front end inside the step, full train2 step",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "hip_graph": tr.cfg.use_graph,
                       "graphs_captured": len(tr._graphs) - g0}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (4 piano + 4 violin)")
    ap.add_argument("--sections", type=int, default=2, help="S=2 <=> 4 s clips")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cqt", action="store_true", help="front end runs the STFT only; the 84 CQT bins of x stay synthetic")
    ap.add_argument("--no-frontend", action="store_true", help="feed a resident model-ready x instead of running the STFT kernel each step")
    ap.add_argument("--infer", action="store_true", help="also time the autoregressive decode (BASELINE configs[3]) and add it to the JSON")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]: a stream of 2..8 s clips, one length bucket per step (S = 1,1,2,2,2,3,3), audio-seconds counted "
                         "on the true clip lengths; replaces the fixed 4 s workload")
    ap.add_argument("--per-rank-bn", action="store_true",
                    help="N > 1: per-rank BatchNorm statistics and batch-coupled losses (valid DDP, NOT the single-process loss; multi-stream "
                         "step) instead of the default loss-matched mode")
    ap.add_argument("--loss-matched", action="store_true",
                    help="N > 1: sync-BN + gathered batch-coupled losses (global-batch semantics, eager) instead of per-rank statistics")
    ap.add_argument("--decoder", default="new", choices=["new", "simple"],
                    help="new_decoder.Decoder (north star) or SimpleDecoder_TransformerOnly.Decoder (SURVEY 8(f)1, 182 M parameters)")
    ap.add_argument("--single-stream", action="store_true", help="capture the three encoder branches on one stream (A/B of the fork/join capture)")
    ap.add_argument("--no-overlap-d", action="store_true", help="run the discriminator phase in line instead of beside the decoder forward (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-objects of the default run: autoregressive_decode (configs[3], B=8 and B=1), parity_mode (f32) and mixed (configs[4])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("AST_DIST_BACKEND", "nccl")      # "gloo" only for rehearsing ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if os.environ.get("AST_ONE_GPU"):          # rehearsal: all ranks on cuda:0
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    import ast_amd
    from ast_amd import train
    ast_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    # N > 1 defaults to the loss-matched mode (sync-BN + gathered batch-coupled losses: the step of the single-process reference on
    # the global batch, north_star "loss-matched"); --per-rank-bn selects the throughput mode
    args.loss_matched = bool(args.loss_matched or (world > 1 and not args.per_rank_bn))
    tr = train.Trainer(train.TrainConfig(use_graph=not args.no_graph, loss_matched=args.loss_matched, decoder=args.decoder, multi_stream=not args.single_stream, overlap_d=not args.no_overlap_d), device=dev, rank=rank, world=world)
    clip_seconds = {1: 3.0, 2: CLIP_SECONDS, 3: 8.0, 4: 10.0}[args.sections]   # clip length that yields S sections
    if args.no_frontend:
        x, labels = train.synthetic_batch(args.batch, args.sections, dev, seed=1000 + rank)
    else:
        # waveforms resident in HBM; every step runs STFT + z-score + sectioning into x[..., :513]
        waves, x, mean, std, labels = train.synthetic_waveform_batch(args.batch, clip_seconds, dev, seed=1000 + rank)
        assert x.shape[1] == args.sections
        if args.no_cqt:
            tr.set_frontend(waves, mean, std)
        else:
            # ... and the CQT of the same waveforms (get_CQT) + z-score + sectioning into x[..., 513:]
            tr.set_frontend(waves, mean, std, torch.zeros(2, 84, device=dev), torch.full((2, 84), 0.25, device=dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mixed:
        res = mixed_bench(args, tr, dev, rank, world, barrier)
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            dist.destroy_process_group()
        return

    for _ in range(args.warmup):
        tr.step(x, labels)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(x, labels)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    losses = {k: float(v) for k, v in tr.losses.items()}
    ms = dt / args.steps * 1e3
    value = world * args.batch * clip_seconds / (dt / args.steps)

    out = {"metric": "audio-seconds/sec/node (train step, 4s@22.05kHz pairs)", "value": value, "unit": "audio-seconds/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"configs[1]: batch={args.batch} {clip_seconds:g} s clips per GPU (S={args.sections}, x=(B,S,2,287,597)), full train2 step: "
                                  + ("" if args.no_frontend else ("STFT front-end from resident waveforms (CQT bins synthetic), " if args.no_cqt else "STFT + CQT front-end from resident waveforms, "))
                                  + ("encoders+decoder+discriminator" if args.decoder == "new" else "encoders + SimpleDecoder_TransformerOnly (SURVEY 8(f)1, 182 M parameters) + discriminator")
                                  + ", all losses, D and G phases, grad clip, Adam",
                      "global_batch": world * args.batch, "parallelism": f"dp{world}", "hip_graph": tr.cfg.use_graph,
                      "grad_allreduce": ("bf16" if tr._wire_dtype == torch.bfloat16 else "f32") if world > 1 else None,
                      "collectives": (("captured inside the step's graph" if tr._dist_in_graph else ("eager step" if tr._matched else "eager, between three graphs")) if tr.cfg.use_graph else "eager") if world > 1 else None,
                      "grad_exchange": ("three buckets (decoder, content, style) all-reduced while the backward pass goes on" if (tr._matched and tr.cfg.bucketed) else "one all-reduce per optimiser group after backward") if world > 1 else None,
                      "dp_semantics": ("global-batch (sync-BN + gathered losses: loss-matched to the single-process step)" if args.loss_matched and world > 1 else "per-rank BN and batch-coupled losses")},
           "losses": losses}
    extras = rank == 0 and world == 1 and args.decoder == "new" and not args.no_extras
    if rank == 0 and world == 1 and args.infer and args.decoder == "new":
        out["autoregressive_decode"] = ar_decode_bench(tr, x, labels, args.sections)
    elif extras:
        # BASELINE configs[3] in the driver's line: B = 8 and B = 1, graph replays only, the CPU leg at 2 reps
        out["autoregressive_decode"] = ar_decode_bench(tr, x, labels, args.sections, cpu_reps=2, eager=False)
        b1 = ar_decode_bench(tr, x[:1].contiguous(), labels[:1], args.sections, cpu_reps=2, eager=False)
        out["autoregressive_decode"]["batch_1"] = {k: b1[k] for k in ("ms_per_batch", "ms_per_batch_recompute_graph", "ms_per_batch_kv_cache_graph",
                                                                        "stft_frames_per_s", "audio_seconds_per_s", "batch", "sections", "cpu", "speedup_vs_cpu")}
    if rank == 0 and world == 1:
        if not args.no_roofline:
            out["roofline"] = kernel_roofline(tr, x, labels, args.dtype)
            # whole step against the MFMA peak: SURVEY 8(d)'s 41.74 GFLOP per section forward + backward (3x convention)
            step_flops = 41.74e9 * args.batch * args.sections
            out["roofline"]["step_tflops"] = step_flops / (ms * 1e-3) / 1e12
            out["roofline"]["step_frac_of_mfma_peak"] = out["roofline"]["step_tflops"] / PEAK_TFLOPS[args.dtype]
        if not args.no_cpu_baseline and args.decoder == "new":
            out["cpu_baseline"] = cpu_baseline()
    if extras and args.dtype == "bf16" and not args.no_frontend and not args.no_cqt:
        # the 1e-3-contract mode (f32 storage, exact v_mfma_f32_16x16x4_f32: tests/test_gpu_bench_config.py f32 bounds) on the
        # same workload, so that its throughput is in the driver's line next to the bf16 figure
        ast_amd.set_compute_dtype(torch.float32)
        tr32 = train.Trainer(train.TrainConfig(use_graph=not args.no_graph), device=dev)
        tr32.set_frontend(waves, mean, std, torch.zeros(2, 84, device=dev), torch.full((2, 84), 0.25, device=dev))
        x32 = x.clone()
        for _ in range(5):
            tr32.step(x32, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr32.step(x32, labels)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / 20
        out["parity_mode"] = {"dtype": "f32", "ms_per_step": dt32 * 1e3, "value": args.batch * clip_seconds / dt32, "unit": "audio-seconds/sec", "steps": 20, "warmup": 5,
                              "step_tflops": 41.74e9 * args.batch * args.sections / dt32 / 1e12,
                              "step_frac_of_mfma_peak": 41.74e9 * args.batch * args.sections / dt32 / 1e12 / PEAK_TFLOPS["f32"],
                              "note": "same workload and step in the f32 compute mode (f32 storage, f32 MFMA): the mode that meets the 1e-3 rel "
                                      "parity contract against the oracle; bf16 deviates by its storage rounding (DESIGN 2)"}
        del tr32, x32
        ast_amd.set_compute_dtype(torch.bfloat16)
    if extras and not args.no_frontend:
        # BASELINE configs[4] in the driver's line: the mixed-length stream on the same trainer (7 more graphs), short
        m = mixed_bench(args, tr, dev, rank, world, barrier, steps=21, warmup=7)
        out["mixed"] = {k: m[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config")}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
