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

import torch
import torch.distributed as dist

CLIP_SECONDS = 4.0
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def cpu_baseline(sample_B=2, S=2, steps=2):
    """The oracle's restatement of the same step (D phase + G phase + clip + Adam) in fp32 on the host
    cores, on a bounded sample (B=2 instead of 8: per-clip cost is batch-linear on CPU)."""
    from oracle import ast_oracle as O
    from oracle import layout as OL
    from oracle import seeded_params as sp
    # the GPU box gives one job a 16-core share whatever os.cpu_count() says: more threads only thrash
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sds = {t: OL.seeded_model_state(t) for t in ("style", "content", "decoder", "disc")}
    gparams = [v for t in ("style", "content", "decoder") for v in sds[t].values() if v.requires_grad]
    dparams = [v for v in sds["disc"].values() if v.requires_grad]
    og, od = torch.optim.Adam(gparams, lr=1e-4), torch.optim.Adam(dparams, lr=1e-4)
    cfg = O.Cfg(training=True, p_drop=0.1)
    x = sp.seeded_input(sample_B, S)
    labels = sp.balanced_labels(sample_B)
    y = x[..., :513]
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        style, cls = O.style_encoder_forward(sds["style"], x, labels, cfg)
        content = O.content_encoder_forward(sds["content"], x, cfg)
        od.zero_grad()
        d_loss, _ = O.adversarial_loss(sds["disc"], style.detach(), cls.detach(), content.detach(), labels, True)
        d_loss.backward()
        torch.nn.utils.clip_grad_norm_(dparams, 1.0)
        od.step()
        og.zero_grad()
        out = O.decoder_forward(sds["decoder"], content, cls[labels], cfg, y=y)
        total = (O.comprehensive_loss(out, y)["total_loss"] + O.infonce_loss(style, labels) + O.margin_loss(cls)
                 + O.disentanglement_loss(style, content.mean(1))
                 + O.adversarial_loss(sds["disc"], style, cls, content, labels, False)[1])
        total.backward()
        torch.nn.utils.clip_grad_norm_(gparams, 1.0)
        og.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {it}: {time.perf_counter() - t0:.2f} s", file=sys.stderr, flush=True)
    t = sorted(times)[len(times) // 2]
    return {"value": sample_B * CLIP_SECONDS / t, "unit": "audio-seconds/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 step (D+G phases, clip, Adam; STFT/CQT front end NOT included, the GPU step includes it), B={sample_B} S={S}, median of {steps} steps after 1 warm-up, {t:.2f} s/step"}


def kernel_roofline(trainer, x, labels, dtype_name):
    """Eager (un-graphed) steps with device events around every GEMM launch; aggregates per kernel
    configuration and reports the one with the largest total time."""
    from ast_amd import ops
    ops.PROFILE = []
    was = trainer.cfg.use_graph
    trainer.cfg.use_graph = False
    nsteps = 2
    for _ in range(nsteps):
        trainer.step(x, labels)
    torch.cuda.synchronize()
    trainer.cfg.use_graph = was
    recs, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for name, flops, nbytes, e0, e1 in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += flops; a[3] += nbytes
    total_t = sum(a[1] for a in agg.values())
    name, (cnt, t, fl, by) = max(agg.items(), key=lambda kv: kv[1][1])
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
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "d_pmc_traffic.json")
    if os.path.exists(pmc):
        # HBM bytes per launch from rocprofv3 PMC passes over this same workload (tools/pmc_traffic.sh + .py: separate
        # FETCH_SIZE / WRITE_SIZE runs, KiB units, gfx950 x2 fetch correction); counters cannot be read in-process
        rec = json.load(open(pmc)).get(name)
        if rec:
            traffic, traffic_src = rec["traffic_bytes"], "profiles/r01/d_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, KiB->bytes, mean per launch)"
    roof.update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "kernel": name, "launches_per_step": cnt // nsteps, "avg_launch_us": t / cnt * 1e6,
                 "algorithmic_flops_per_launch": fl / cnt, "algorithmic_bytes_per_launch": by / cnt,
                 "tflops": tf, "frac_of_mfma_peak": tf / peak, "gbs": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                 "gemm_time_per_step_ms": total_t / nsteps * 1e3,
                 "all_gemm_tflops": sum(a[2] for a in agg.values()) / total_t / 1e12})
    return roof


def ar_decode_bench(tr, x, labels, S, iters=20):
    """BASELINE configs[3]: the reference's process_audio (evaluation_style_transfer.py:135-159) for a batch of clips:
    eval-mode content encoder + autoregressive new_decoder generation of S sections + overlap-average + iSTFT ->
    waveforms, as one replayed hipGraph (and eagerly, for comparison).  Class embeddings come from one style-encoder
    pass beforehand, as the reference precomputes them."""
    from ast_amd import infer
    for m in (tr.style, tr.content, tr.decoder):
        m.eval()
    with torch.no_grad():
        _, ce = tr.style(x, labels)
    cls = ce[labels.to(x.device)].contiguous()
    res = {}
    for name, use_graph in (("graph", True), ("eager", False)):
        sess = infer.StyleTransferSession(tr.content, tr.decoder, use_graph=use_graph)
        for _ in range(3):
            sess(x, cls)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            sess(x, cls)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / iters
    for m in (tr.style, tr.content, tr.decoder):
        m.train()
    B = x.shape[0]
    frames = B * (191 * (S - 1) + 287)
    dt = res["graph"]
    return {"ms_per_batch": dt * 1e3, "ms_per_batch_eager": res["eager"] * 1e3, "stft_frames_per_s": frames / dt,
            "audio_seconds_per_s": B * CLIP_SECONDS / dt,
            "note": "content encoder + O(S^2) AR decoder loop (new_decoder.py:272-319) + overlap-average + iSTFT to waveforms, one hipGraph"}


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
    ap.add_argument("--loss-matched", action="store_true",
                    help="N > 1: sync-BN + gathered batch-coupled losses (global-batch semantics, eager) instead of per-rank statistics")
    ap.add_argument("--decoder", default="new", choices=["new", "simple"],
                    help="new_decoder.Decoder (north star) or SimpleDecoder_TransformerOnly.Decoder (SURVEY 8(f)1, 182 M parameters)")
    ap.add_argument("--single-stream", action="store_true", help="capture the three encoder branches on one stream (A/B of the fork/join capture)")
    ap.add_argument("--no-overlap-d", action="store_true", help="run the discriminator phase in line instead of beside the decoder forward (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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
                      "dp_semantics": ("global-batch (sync-BN + gathered losses)" if args.loss_matched and world > 1 else "per-rank BN and batch-coupled losses")},
           "losses": losses}
    if rank == 0 and world == 1 and args.infer and args.decoder == "new":
        out["autoregressive_decode"] = ar_decode_bench(tr, x, labels, args.sections)
    if rank == 0 and world == 1:
        if not args.no_roofline:
            out["roofline"] = kernel_roofline(tr, x, labels, args.dtype)
        if not args.no_cpu_baseline and args.decoder == "new":
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
