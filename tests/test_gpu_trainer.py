"""The train step must give the same losses / parameters whether it runs eagerly on one stream,
on three concurrent streams, or as a replayed hipGraph (dropout off => deterministic up to the
summation order of f32 atomics)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import ast_amd
    from ast_amd import train


def _run(use_graph, multi_stream, steps=3, segmented=False):
    ast_amd.set_compute_dtype(torch.float32)
    tr = train.Trainer(train.TrainConfig(use_graph=use_graph, multi_stream=multi_stream, dropout=False, segmented=segmented), seed=7)
    x, labels = train.synthetic_batch(4, 1, "cuda:0", seed=3)
    hist = []
    for _ in range(steps):
        out = tr.step(x, labels)
        hist.append({k: float(v) for k, v in out.items()})
    torch.cuda.synchronize()
    return hist, float(tr.G.flat_p.double().sum()), float(tr.G.flat_p.double().abs().sum()), tr


def _noise_tolerances(run_a, run_b, factor=8.0, cap=5e-2):
    """Per (step, key) relative tolerance for comparing two execution MODES of the same step, from the measured difference of
    two IDENTICAL runs (run_a, run_b: loss histories of the same mode and seed).  The only run-to-run noise is the summation
    order of f32 atomics; it is amplified by Adam's first update (lr * sign(g) for every parameter: a gradient near zero flips),
    so everything behind an optimiser step -- all of steps 1.., and adv_g / total of step 0, which sit behind the
    discriminator's update -- carries it.  tol = max(floor, factor x measured noise), with tight floors where no optimiser
    step lies in between (3e-4) and 2e-3 / 5e-3 behind one; a noise level that would need more than `cap` fails the test."""
    tols = []
    for i, (a, b) in enumerate(zip(run_a, run_b)):
        rel = {k: abs(a[k] - b[k]) / max(abs(a[k]), 1e-5) for k in a}
        step_noise = max(rel.values())
        t = {}
        for k in a:
            floor = 5e-3 if i > 0 else (2e-3 if k in ("adv_g", "total") else 3e-4)
            behind_update = i > 0 or k in ("adv_g", "total")
            t[k] = max(floor, factor * max(rel[k], 0.5 * step_noise if behind_update else 0.0))
            assert t[k] <= cap, f"two identical runs differ by {rel[k]:.2e} at step {i} on {k}: noise, not a tolerance question"
            if k == "hsic" and i > 0:
                # HSIC is DISCONTINUOUS in the parameters: its kernel width is the median of the pairwise embedding distances
                # (losses.py:170-171), and once the parameters differ in the last bits the median can sit on the neighbouring pair
                # (measured: graph 4.41e-3 against eager 3.98e-3 at step 5 while two eager runs agreed to 1e-4).  Its value at the
                # steps behind an optimiser update is bounded loosely; `total`, which contains it, stays at the tight bound.
                t[k] = max(t[k], 0.15)
        tols.append(t)
    return tols


def _assert_close_hist(hist, ref_hist, tols, what):
    for i, (h, r) in enumerate(zip(hist, ref_hist)):
        for k in r:
            assert math.isclose(h[k], r[k], rel_tol=tols[i][k], abs_tol=1e-5), (what, i, k, h[k], r[k], tols[i][k])


def test_trainer_modes_agree():
    ref_hist, ref_sum, ref_abs, tr0 = _run(False, False)
    ref2_hist = _run(False, False)[0]                              # the same mode again: the run-to-run noise floor
    tols = _noise_tolerances(ref_hist, ref2_hist)
    print("tolerances from two identical eager runs:", [{k: f"{v:.1e}" for k, v in t.items()} for t in tols])
    assert all(math.isfinite(v) for h in ref_hist for v in h.values())
    assert ref_hist[0]["total"] != ref_hist[-1]["total"]           # the optimiser actually moves the weights
    assert int(tr0.G.step) == len(ref_hist) and int(tr0.D.step) == len(ref_hist)
    for mode in ((False, True, 3, False), (True, True, 3, False), (True, True, 3, True)):   # last: the 3-graph form DP uses
        hist, s, a, _ = _run(*mode)
        _assert_close_hist(hist, ref_hist, tols, mode)
        assert math.isclose(a, ref_abs, rel_tol=1e-6), (mode, a, ref_abs)


def test_weight_gradient_scheduling_modes_agree():
    """Where the convolution weight gradients are LAUNCHED must not change the step: inside each layer's backward node (the
    reference order), deferred to the banks' end-of-backward flush (the default), pooled over all banks' flush streams, on their
    own stream beside the data-gradient chain, and with the slab flush -- eager and as a captured graph."""
    from ast_amd import config
    saved = (config.wgrad_defer, config.wgrad_defer_pool, config.wgrad_slabs, os.environ.get("AST_WGRAD_STREAM"))
    try:
        config.wgrad_defer, config.wgrad_defer_pool, config.wgrad_slabs = False, False, 0
        ref_hist, _, ref_abs, _ = _run(False, True)
        tols = _noise_tolerances(ref_hist, _run(False, True)[0])
        for name, defer, pool, slabs, wstream in (("deferred", True, False, 0, None), ("pooled", True, True, 0, None),
                                                   ("deferred+slabs", True, False, 128, None), ("side stream", False, False, 0, "1")):
            config.wgrad_defer, config.wgrad_defer_pool, config.wgrad_slabs = defer, pool, slabs
            os.environ.pop("AST_WGRAD_STREAM", None)
            if wstream:
                os.environ["AST_WGRAD_STREAM"] = wstream
            for use_graph in (False, True):
                hist, _, a, _ = _run(use_graph, True)
                _assert_close_hist(hist, ref_hist, tols, (name, use_graph))
                assert math.isclose(a, ref_abs, rel_tol=1e-6), (name, use_graph, a, ref_abs)
    finally:
        config.wgrad_defer, config.wgrad_defer_pool, config.wgrad_slabs = saved[:3]
        os.environ.pop("AST_WGRAD_STREAM", None)
        if saved[3] is not None:
            os.environ["AST_WGRAD_STREAM"] = saved[3]


def test_loss_goes_down_on_a_fixed_batch():
    ast_amd.set_compute_dtype(torch.float32)
    cfg = train.TrainConfig(use_graph=True, dropout=False, lr_g=2e-4, use_adv=False, use_hsic=False, use_nce=False)
    tr = train.Trainer(cfg, seed=11)
    x, labels = train.synthetic_batch(2, 1, "cuda:0", seed=5)
    first = float(tr.step(x, labels)["rec"])
    for _ in range(25):
        last = float(tr.step(x, labels)["rec"])
    assert last < first, (first, last)


def test_frontend_in_step_matches_prefilled_input():
    """step() with the STFT front-end attached == step() on an x whose STFT bins were filled beforehand."""
    from ast_amd import utilityFunctions as U
    ast_amd.set_compute_dtype(torch.float32)
    waves, x, mean, std, labels = train.synthetic_waveform_batch(2, 4.0, "cuda:0", seed=9)
    assert x.shape == (2, 2, 2, 287, 597)
    x_ref = x.clone()
    U.stft_sections(waves, mean, std, n_sections=2, F_total=597, out=x_ref)
    assert float((x_ref[..., :513] - x[..., :513]).abs().max()) > 0 and torch.equal(x_ref[..., 513:], x[..., 513:])
    a = train.Trainer(train.TrainConfig(use_graph=False, dropout=False), seed=3)
    ra = {k: float(v) for k, v in a.step(x_ref, labels).items()}
    b = train.Trainer(train.TrainConfig(use_graph=True, dropout=False), seed=3)
    b.set_frontend(waves, mean, std)
    rb = {k: float(v) for k, v in b.step(x.clone(), labels).items()}
    # adv_g (and through it total) is evaluated after the discriminator's Adam step, whose first update is +-lr per
    # parameter whatever the gradient's size: atomic-order noise in near-zero gradients flips signs, and the spread
    # between two identical eager runs is already 3e-4 (tools/noise.py)
    for k in ra:
        tol = 2e-3 if k in ("adv_g", "total") else 2e-4
        assert math.isclose(ra[k], rb[k], rel_tol=tol, abs_tol=1e-6), (k, ra[k], rb[k])


def test_loss_matched_data_parallel_two_ranks_one_gpu():
    """SURVEY 8(e)(2)+(3): two ranks (gloo, both on cuda:0) with sync-BN and gathered batch-coupled losses reproduce
    the single-process step on the global batch; without them (the default graph mode's semantics) InfoNCE differs."""
    import json, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dp_match.py")
    out = subprocess.run([sys.executable, tool, "--batch", "8", "--sections", "1", "--port", "29561"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    a, b = r["losses_single"], r["losses_dp"]
    for k in a:
        tol = 2e-3 if k in ("adv_g", "total") else 2e-4          # adv_g: behind the sign-sensitive first Adam step of D
        assert math.isclose(a[k], b[k], rel_tol=tol, abs_tol=1e-6), (k, a[k], b[k])
    assert r["grad_cos"] > 0.9999 and r["grad_rel_err"] < 2e-2, r
    assert r["bn_running_mean_err"] < 1e-6, r


def test_inference_session_graph_matches_eager_and_reference_tail():
    """evaluation_style_transfer.py:135-159 as one hipGraph: same output as the eager module calls, and the waveform
    tail (overlap-average + iSTFT kernels) equals the oracle's sections_to_spectrogram + istft of the same sections."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import frontend_oracle as FO
    from ast_amd import infer
    from ast_amd.content_encoder import ContentEncoder
    from ast_amd.new_decoder import Decoder
    ast_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(5)
    ce, de = ContentEncoder().cuda(), Decoder().cuda()
    with torch.no_grad():
        for name, p in de.named_parameters():
            if p.dim() == 1 and "weight" in name:
                p.fill_(1.0)                     # a fresh reference decoder has zero gammas and outputs 0
    x = torch.randn(2, 3, 2, 287, 597, device="cuda")
    cls = torch.randn(2, 256, device="cuda")
    eager = infer.StyleTransferSession(ce, de, use_graph=False)
    graph = infer.StyleTransferSession(ce, de, use_graph=True)
    w0, o0 = eager(x, cls)
    w1, o1 = graph(x, cls)
    w2, _ = graph(x.clone(), cls.clone())              # replay with fresh buffers
    assert o0.shape == (2, 3, 2, 287, 513) and w0.shape == (2, 256 * (191 * 2 + 287 - 1))
    assert float(o0.abs().max()) > 0
    def rel(a, b):
        return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    assert rel(o1, o0) < 1e-5 and rel(w1, w0) < 1e-5 and rel(w2, w1) < 1e-5      # scale-relative: f32 atomic order differs
    for b in range(2):
        spec = FO.sections_to_spectrogram(o0[b].cpu().numpy(), 191 * 2 + 287, 96)
        wave = FO.istft(spec)
        err = float((torch.from_numpy(wave) - w0[b].cpu()).abs().max()) / (float(abs(wave).max()) + 1e-12)
        assert err < 1e-4, err


def test_trainer_with_simple_decoder_graph_matches_eager():
    """the same train step around SimpleDecoder_TransformerOnly.Decoder (182 M parameters, 0.73 GB of
    gradients), replayed as a hipGraph, equals the eager step."""
    ast_amd.set_compute_dtype(torch.float32)
    x, labels = train.synthetic_batch(2, 2, "cuda:0", seed=4)
    res = []
    for use_graph in (False, True):
        tr = train.Trainer(train.TrainConfig(use_graph=use_graph, dropout=False, decoder="simple"), seed=5)
        assert tr.G.n > 180_000_000
        h = [{k: float(v) for k, v in tr.step(x, labels).items()} for _ in range(2)]
        res.append(h)
        del tr
        torch.cuda.empty_cache()
    for a, b in zip(*res):
        for k in a:
            assert math.isfinite(a[k]) and math.isclose(a[k], b[k], rel_tol=5e-3, abs_tol=1e-5), (k, a[k], b[k])
    assert res[0][1]["rec"] != res[0][0]["rec"]


def _seq_losses(tr, batches, labels):
    hist = []
    for x in batches:
        hist.append({k: float(v) for k, v in tr.step(x, labels).items()})
    torch.cuda.synchronize()
    return hist


def test_graph_replays_read_each_new_batch():
    """A replayed step must see the values of the batch it is given (the captured graph contains the input layout
    conversion; nothing is cached across steps): graph vs eager over three DISTINCT batches."""
    ast_amd.set_compute_dtype(torch.float32)
    labels = train.synthetic_batch(4, 1, "cuda:0", seed=3)[1]
    batches = [train.synthetic_batch(4, 1, "cuda:0", seed=s)[0] for s in (3, 4, 5)]
    batches[2] = 3.0 * batches[2]                                     # make the third one unmistakably different
    eager = _seq_losses(train.Trainer(train.TrainConfig(use_graph=False, multi_stream=False, dropout=False), seed=7), batches, labels)
    eager2 = _seq_losses(train.Trainer(train.TrainConfig(use_graph=False, multi_stream=False, dropout=False), seed=7), batches, labels)
    graph = _seq_losses(train.Trainer(train.TrainConfig(use_graph=True, dropout=False), seed=7), batches, labels)
    assert abs(eager[2]["rec"] - eager[0]["rec"]) > 0.1 * abs(eager[0]["rec"])        # the batches do differ
    _assert_close_hist(graph, eager, _noise_tolerances(eager, eager2), "graph vs eager")     # bounds from the measured run-to-run noise


def test_inference_session_reads_each_new_clip():
    from ast_amd import infer
    from ast_amd.content_encoder import ContentEncoder
    from ast_amd.new_decoder import Decoder
    from oracle import seeded_params as sp
    ast_amd.set_compute_dtype(torch.float32)
    ce, de = ContentEncoder(), Decoder()
    ce.load_state_dict(sp.seeded_state_dict(ce.state_dict(), tag="content"))      # non-degenerate parameters: a fresh
    de.load_state_dict(sp.seeded_state_dict(de.state_dict(), tag="decoder"))      # decoder outputs 0
    ce, de = ce.cuda().train(), de.cuda().train()
    cls = sp.seeded_normal((2, 256), 77).cuda()
    # eval mode normalises with the BatchNorm running statistics: make them those of real activations (one training
    # forward with momentum 1), otherwise the seeded running stats let the activations explode and the output barely
    # depends on the clip
    x0 = sp.seeded_input(2, 2, seed=9).cuda()
    bns = [m for mod in (ce, de) for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        de(ce(x0), cls, y=x0[..., :513])
    for m in bns:
        m.momentum = 0.1
    eager = infer.StyleTransferSession(ce, de, use_graph=False)
    graph = infer.StyleTransferSession(ce, de, use_graph=True)
    outs = []
    for i, seed in enumerate((1, 2)):
        x = (1.0 + 2.0 * i) * sp.seeded_input(2, 2, seed=seed).cuda()
        w_e, o_e = eager(x, cls)
        w_g, o_g = graph(x, cls)
        err = float((o_g - o_e).abs().max()) / float(o_e.abs().max())
        assert err < 1e-4, (seed, err)
        outs.append(o_e.clone())
    # different clips, different outputs (a graph that kept reading the first clip would fail the line above)
    diff = float((outs[0] - outs[1]).abs().max()) / float(outs[0].abs().max())
    assert diff > 1e-2, diff


def test_train_eval_train_on_one_module():
    """An eval / no-grad forward between two training steps (validation, an inference session) must leave the
    training-side weight-gradient staging intact."""
    ast_amd.set_compute_dtype(torch.float32)
    torch.manual_seed(2)
    enc = ast_amd.ContentEncoder().cuda().train()
    x = torch.randn(2, 1, 2, 287, 597, device="cuda")

    def train_pass():
        for p in enc.parameters():
            p.grad = None
        enc(x).square().mean().backward()
        torch.cuda.synchronize()
        return float(enc.cnn[0].conv1.weight_orig.grad.norm())
    g0 = train_pass()
    enc.eval()
    with torch.no_grad():
        enc(x)
    enc.train()
    g1 = train_pass()                     # raised "weights were prepared in eval/no-grad mode" before the fix
    assert g0 > 0 and math.isfinite(g1)
    # (the values differ slightly: BN running stats are not used in training mode, but the spectral-norm u, v advanced)
    assert math.isclose(g0, g1, rel_tol=0.2)


def test_set_frontend_per_batch_under_graph():
    """set_frontend() with a NEW batch of waveforms between replayed steps: the trainer owns the front-end buffers, so
    the captured graph reads the new clips (eager and graph agree step by step)."""
    ast_amd.set_compute_dtype(torch.float32)
    w1, x, mean, std, labels = train.synthetic_waveform_batch(2, 4.0, "cuda:0", seed=9)
    w2 = train.synthetic_waveform_batch(2, 4.0, "cuda:0", seed=21)[0]
    res = []
    for use_graph in (False, False, True):
        tr = train.Trainer(train.TrainConfig(use_graph=use_graph, dropout=False), seed=3)
        h = []
        for w in (w1, w2, 0.5 * w1):
            tr.set_frontend(w, mean, std)
            h.append({k: float(v) for k, v in tr.step(x.clone(), labels).items()})
        res.append(h)
        if use_graph:
            assert len(tr._graphs) == 1               # same buffers, one capture
    assert abs(res[0][1]["rec"] - res[0][0]["rec"]) > 1e-3
    _assert_close_hist(res[2], res[0], _noise_tolerances(res[0], res[1]), "graph vs eager")


def test_mixed_length_stream_vs_oracle():
    """BASELINE configs[4]: a stream of clips of different lengths (2 s, 4 s, 7 s -> S = 1, 2, 3) through ONE Trainer: each
    length bucket gets its own front-end buffers and captured graph (the second visit of a bucket replays), and every
    step's loss scalars follow the CPU oracle's trainer fed the oracle front end of the same waveforms."""
    import numpy as np
    from oracle import cqt_oracle as CO
    from oracle import frontend_oracle as FO
    from oracle import seeded_params as sp
    from oracle.train_step import OracleTrainer
    from ast_amd.dataloader import sections_for_samples
    ast_amd.set_compute_dtype(torch.float32)
    B = 4              # HSIC's kernel width is the MEDIAN of the 2B x 2B pairwise distances: at B = 2 a two-pair knife edge (see below)
    g = np.random.default_rng(5)
    mean, std = (0.01 * g.standard_normal((2, 513))).astype(np.float32), (0.5 + g.random((2, 513))).astype(np.float32)
    cmean, cstd = np.zeros((2, 84), np.float32), np.full((2, 84), 0.25, np.float32)
    labels = sp.balanced_labels(B)

    def batch(seconds, seed):
        waves = np.stack([FO.synth_waveform(seed + i, "piano" if i < B // 2 else "violin", seconds=seconds) for i in range(B)])
        xs = []
        for w in waves:
            spec = np.concatenate([FO.normalize(FO.stft(w), mean, std), FO.normalize(CO.get_cqt(w), cmean, cstd)], axis=2)
            xs.append(FO.overlap_windows(spec))
        return waves.astype(np.float32), torch.from_numpy(np.stack(xs).astype(np.float32))

    tr = train.Trainer(train.TrainConfig(use_graph=True, dropout=False), seed=7)
    for tag, m in (("style", tr.style), ("content", tr.content), ("decoder", tr.decoder), ("disc", tr.disc)):
        m.load_state_dict({k: v.to("cuda") for k, v in sp.seeded_state_dict(m.state_dict(), tag=tag).items()})
    ot = OracleTrainer()
    dev = lambda a: torch.from_numpy(a).cuda()          # noqa: E731
    seq = [(2.0, 10), (4.0, 20), (7.0, 30), (4.0, 40), (2.0, 50)]
    for i, (seconds, seed) in enumerate(seq):
        waves, x_ref = batch(seconds, seed)
        S = sections_for_samples(waves.shape[1])
        assert x_ref.shape == (B, S, 2, 287, 597) and S == {2.0: 1, 4.0: 2, 7.0: 3}[seconds]
        tr.set_frontend(dev(waves), dev(mean), dev(std), dev(cmean), dev(cstd))
        got = {k: float(v) for k, v in tr.step(torch.zeros((B, S, 2, 287, 597), device="cuda"), labels).items()}
        ref = ot.step(x_ref, labels)
        for k in got:
            # steps after the first carry the optimisers' state: f32 summation-order noise compounds (as test_trainer_modes_agree).
            # HSIC's kernel width is an ORDER statistic (median) of the pairwise distances: once the parameters differ in the last
            # bits the median can pick a neighbouring pair.  At B = 2 (16 distances, round 2) that was a jump of up to 9e-2 and the
            # band was 25e-2 -- set while a CQT race was still corrupting these very steps (fixed: DESIGN 8.11); at B = 4 the
            # median sits among 64 distances and neighbouring order statistics are closer (measured 4e-4 .. 1e-3); 0.15 covers a flip.
            tol = (0.15 if k == "hsic" else 2e-2) if i > 0 else 2e-3
            print(f"step {i} {k}: {got[k]:.6f} vs {ref[k]:.6f} ({abs(got[k] - ref[k]) / max(abs(ref[k]), 1e-9):.2e})")
            assert math.isclose(got[k], ref[k], rel_tol=tol, abs_tol=2e-4), (i, seconds, k, got[k], ref[k])
    assert len(tr._graphs) == 3                            # one capture per length bucket; revisits replay


def ops_sync_active():
    from ast_amd import ops
    return ops._SyncBN.active


def test_collectives_inside_the_captured_step_one_rank():
    """Data parallel, default mode: the two gradient all-reduces are RCCL calls captured INSIDE the step's graph (world > 1
    then replays one graph, like a single GPU).  One-rank rehearsal on this box: a 1-rank NCCL group, the collective path
    forced on (AST_FORCE_COLLECTIVES=1): the probe must accept, the step must be ONE graph, and three steps must give the
    losses of a plain single-GPU trainer (an all-reduce over one rank is the identity)."""
    import torch.distributed as dist
    ast_amd.set_compute_dtype(torch.bfloat16)
    x, labels = train.synthetic_batch(2, 2, "cuda:0")
    ref = train.Trainer(train.TrainConfig(dropout=False), device="cuda:0")
    ref_losses = [float(ref.step(x, labels)["total"]) for _ in range(3)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", AST_FORCE_COLLECTIVES="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        tr = train.Trainer(train.TrainConfig(dropout=False), device="cuda:0")
        assert tr._dist and tr._force_coll
        losses = [float(tr.step(x, labels)["total"]) for _ in range(3)]
        assert tr._dist_in_graph is True, "the probe refused all-reduce capture on this stack"
        (graphs, _, _), = tr._graphs.values()
        assert len(graphs) == 1
        # bf16 at B = 2: the margin term moves by ~1e-2 with the summation order of the statistics atomics alone (two plain
        # trainers differ by that much), and the collective path rounds the gradient to the bf16 wire
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 1e-2 * max(1.0, abs(b)), (losses, ref_losses)
        # The LOSS-MATCHED mode on the same one-rank group (TrainConfig.loss_matched): the BatchNorm statistics all-reduces
        # (forward and backward), the embedding gathers and the three-bucket gradient exchange -- ~70 small collectives -- are all
        # captured inside ONE graph of a single-stream step; over one rank they are identities, so the losses must again follow the
        # plain trainer.
        trm = train.Trainer(train.TrainConfig(dropout=False, loss_matched=True), device="cuda:0")
        assert trm._matched and not trm.cfg.multi_stream and ops_sync_active()
        lm = [float(trm.step(x, labels)["total"]) for _ in range(3)]
        assert trm._dist_in_graph is True
        (graphs_m, _, _), = trm._graphs.values()
        assert len(graphs_m) == 1
        for a, b in zip(lm, ref_losses):
            assert abs(a - b) <= 1e-2 * max(1.0, abs(b)), (lm, ref_losses)
    finally:
        from ast_amd import ops as _ops
        _ops.set_sync_bn(1)
        os.environ.pop("AST_FORCE_COLLECTIVES", None)
        dist.destroy_process_group()


def test_lr_schedule_and_ramped_weights_replay_one_graph():
    """A 10-step linear LR warm-up, a ramped adversarial / HSIC weight and a clip norm that changes half way (the reconstructed
    train2 step: scheduler.step(), lambda_adv(t) -- SURVEY 3.1, README.md:144-150) under hipGraph replay: ONE capture, and the
    losses follow the eager trainer given the same schedule step by step.  The step scalars live on the device
    (Trainer.hyper, ast_adam_dev / ast_weighted_sum read them at run time)."""
    ast_amd.set_compute_dtype(torch.float32)
    x, labels = train.synthetic_batch(4, 1, "cuda:0", seed=3)

    def run(use_graph, schedule=True):
        tr = train.Trainer(train.TrainConfig(use_graph=use_graph, dropout=False), seed=7)
        hist = []
        for i in range(10):
            if schedule:
                tr.cfg.lr_g, tr.cfg.lr_d = 2e-4 * (i + 1) / 10, 1e-4 * (i + 1) / 10
                tr.cfg.w_adv, tr.cfg.w_hsic = 0.2 * i, 1.0 + 0.1 * i
                tr.cfg.max_grad_norm = 1.0 if i < 5 else 0.25
            hist.append({k: float(v) for k, v in tr.step(x, labels).items()})
        torch.cuda.synchronize()
        return hist, tr

    eager, _ = run(False)
    eager2, _ = run(False)
    graph, trg = run(True)
    assert len(trg._graphs) == 1, f"{len(trg._graphs)} captures for one shape: a step scalar is still baked into the graph"
    _assert_close_hist(graph, eager, _noise_tolerances(eager, eager2), "scheduled graph vs scheduled eager")
    const, _ = run(True, schedule=False)
    # ... and the schedule is really applied: w_adv = 0 at step 0 (total without the adversarial term) and the later steps move
    assert abs(graph[0]["total"] - const[0]["total"]) > 1e-3 * abs(const[0]["total"])
    assert abs(graph[9]["rec"] - const[9]["rec"]) > 1e-5 * abs(const[9]["rec"])
    w = trg.hyper.cpu()
    assert abs(float(w[train.H_LR_G]) - 2e-4) < 1e-9 and abs(float(w[train.H_W_ADV]) - 1.8) < 1e-6 and abs(float(w[train.H_LR_G + 1]) - 0.25) < 1e-7


def test_graph_replay_survives_an_eval_pass_between_steps():
    """train step (captured) -> eval-mode forward of the same modules (a validation pass: the weight banks build their eval
    tables) -> train steps (REPLAYED).  The captured graph bakes the addresses of the banks' descriptor / tile tables in; an eval
    build used to replace and free them, and the next replay read whatever the allocator had put there (a GPU memory fault in
    bench.py's default run).  The tables are content-addressed and never freed now (layers.WeightBank._const_dev)."""
    ast_amd.set_compute_dtype(torch.float32)
    x, labels = train.synthetic_batch(4, 1, "cuda:0", seed=3)

    def run(use_graph):
        tr = train.Trainer(train.TrainConfig(use_graph=use_graph, dropout=False), seed=7)
        hist = [{k: float(v) for k, v in tr.step(x, labels).items()}]
        for m in (tr.style, tr.content, tr.decoder):
            m.eval()
        with torch.no_grad():
            se, ce = tr.style(x, labels)
            co = tr.content(x)
            tr.decoder(co, ce[labels.to(x.device)].contiguous())
        for m in (tr.style, tr.content, tr.decoder):
            m.train()
        junk = [torch.full((1 << 12,), 0x7fffffff, dtype=torch.int32, device="cuda:0") for _ in range(256)]   # land on whatever was freed
        for _ in range(2):
            hist.append({k: float(v) for k, v in tr.step(x, labels).items()})
        torch.cuda.synchronize()
        del junk
        return hist, tr

    eager, _ = run(False)
    eager2, _ = run(False)
    graph, trg = run(True)
    assert len(trg._graphs) == 1
    _assert_close_hist(graph, eager, _noise_tolerances(eager, eager2), "graph vs eager around an eval pass")
