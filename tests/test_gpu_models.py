"""Model-level parity of the HIP-backed drop-in modules against the CPU oracle
and the golden fixtures captured from the reference (config 1: B=2,S=2 and
B=4,S=1), forward + backward + mutated buffers + autoregressive decode.

Tolerance (north_star: 1e-3 rel fp32): the f32 path is checked at 1e-3 of the
tensor scale for outputs/embeddings and loss scalars; gradients at 1e-2 (the
reference-vs-oracle fp32 re-association noise on gradients is itself 2e-3, see
tests/test_oracle_golden.py).  The bf16 path (throughput mode) is checked at
loss-scalar level 2e-2 and relative-L2 5e-2 on outputs.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import ast_amd
    from ast_amd import config
from oracle import ast_oracle as O
from oracle import layout as OL
from oracle import seeded_params as sp

DEV = "cuda"


def rel_err(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / max(b.abs().max().item(), 1e-12))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / max(b.norm().item(), 1e-12))


def build_models():
    ms = {}
    for tag, ctor in (("style", ast_amd.StyleEncoder), ("content", ast_amd.ContentEncoder),
                      ("decoder", ast_amd.Decoder), ("disc", ast_amd.Discriminator)):
        m = ctor()
        m.load_state_dict(sp.seeded_state_dict(m.state_dict(), tag=tag))
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        ms[tag] = m.to(DEV).train()
    return ms


def hip_step(ms, x, labels):
    y = x[..., :513]
    style, cls = ms["style"](x, labels)
    content = ms["content"](x)
    out = ms["decoder"](content, cls[labels.to(DEV)], y=y)
    rec = ast_amd.compute_comprehensive_loss(out, y)
    nce = ast_amd.infoNCE_loss(style, labels)
    mar = ast_amd.margin_loss(cls)
    hs = ast_amd.disentanglement_loss(style, content.mean(1))
    d_loss, g_loss = ast_amd.adversarial_loss(style, cls, content, ms["disc"], labels, False)
    total = rec["total_loss"] + nce + mar + hs + g_loss
    total.backward()
    return dict(style=style, cls=cls, content=content, out=out, rec=rec, nce=nce, mar=mar, hs=hs, d_loss=d_loss,
                g_loss=g_loss, total=total)


def oracle_step(B, S):
    sds = {t: OL.seeded_model_state(t) for t in ("style", "content", "decoder", "disc")}
    cfg = O.Cfg(training=True, p_drop=0.0)
    x = sp.seeded_input(B, S)
    labels = sp.balanced_labels(B)
    y = x[..., :513]
    style, cls = O.style_encoder_forward(sds["style"], x, labels, cfg)
    content = O.content_encoder_forward(sds["content"], x, cfg)
    out = O.decoder_forward(sds["decoder"], content, cls[labels], cfg, y=y)
    rec = O.comprehensive_loss(out, y)
    total = (rec["total_loss"] + O.infonce_loss(style, labels) + O.margin_loss(cls)
             + O.disentanglement_loss(style, content.mean(1))
             + O.adversarial_loss(sds["disc"], style, cls, content, labels, False)[1])
    total.backward()
    return dict(sds=sds, style=style, cls=cls, content=content, out=out, rec=rec, total=total)


@pytest.mark.parametrize("name,B,S", [("b2s2", 2, 2), ("b4s1", 4, 1)])
def test_full_step_f32_vs_golden_and_oracle(golden_dir, name, B, S):
    config.set_compute_dtype(torch.float32)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    ms = build_models()
    x = sp.seeded_input(B, S).to(DEV)
    labels = sp.balanced_labels(B)
    r = hip_step(ms, x, labels)
    # ---- forward against the reference's golden vectors (1e-3 of scale)
    assert rel_err(r["style"], g["style_emb"]) < 1e-3
    assert rel_err(r["cls"], g["class_emb"]) < 1e-3
    assert rel_err(r["content"], g["content_emb"]) < 1e-3
    out = r["out"].detach()
    assert rel_err(out[:, :, :, ::11, ::13], g["out_sub"]) < 1e-3
    assert rel_err(out.abs().sum(dim=(3, 4)), g["out_abs_sum"]) < 1e-3
    for k in ("total_loss", "mse_loss", "mag_loss", "phase_loss", "temporal_loss", "spectral_loss"):
        assert math.isclose(float(r["rec"][k]), float(g["rec_" + k]), rel_tol=1e-3, abs_tol=1e-6), k
    for key, val in (("infonce", r["nce"]), ("margin", r["mar"]), ("hsic", r["hs"]), ("adv_d", r["d_loss"]),
                     ("adv_g", r["g_loss"]), ("total", r["total"])):
        assert math.isclose(float(val), float(g["loss_" + key]), rel_tol=1e-3, abs_tol=1e-6), key
    # ---- buffers mutated by the training forward
    st, dec = ms["style"], ms["decoder"]
    assert rel_err(st.cnn.net[0].bn1.running_mean, g["bn_rm_style_b0_bn1"]) < 1e-3
    assert rel_err(st.cnn.net[0].bn1.running_var, g["bn_rv_style_b0_bn1"]) < 1e-3
    assert rel_err(dec.conv_encoder[1].running_mean, g["bn_rm_dec_ce1"]) < 1e-3
    assert rel_err(dec.conv_decoder[10].running_var, g["bn_rv_dec_cd10"]) < 1e-3
    assert rel_err(st.cnn.net[0].conv1.weight_u, g["sn_u_style_b0_conv1"]) < 1e-4
    assert rel_err(st.cnn.net[5].conv2.weight_v, g["sn_v_style_b5_conv2"]) < 1e-4
    assert rel_err(dec.conv_decoder[3].weight_u, g["sn_u_dec_cd3"]) < 1e-4
    assert rel_err(dec.conv_decoder[3].weight_v, g["sn_v_dec_cd3"]) < 1e-4
    assert int(st.cnn.net[0].bn1.num_batches_tracked) == 1
    # ---- gradients: per-parameter norms + raw slices from the reference
    for tag in ("style", "content", "decoder"):
        params = dict(ms[tag].named_parameters())
        bad = []
        for k, v in zip(g[f"gradnorm_keys_{tag}"], g[f"gradnorm_vals_{tag}"]):
            k = str(k)
            gr = params[k].grad
            got = 0.0 if gr is None else float(gr.norm())
            if v < 1e-3:          # incl. biases in front of a norm layer: identically 0 here, rounding noise there
                ok = got < 2e-3
            else:
                ok = math.isclose(got, v, rel_tol=1e-2)
            if not ok:
                bad.append((k, got, float(v)))
        assert not bad, (tag, bad[:8])
    assert rel_err(st.cnn.net[0].conv1.weight_orig.grad, g["grad_style_conv1_0"]) < 1e-2
    assert rel_err(st.cnn.proj.weight.grad[:8], g["grad_style_proj_w"]) < 1e-2
    assert rel_err(ms["content"].cnn[5].conv2.weight_orig.grad[:4, :4], g["grad_content_b5_conv2"]) < 1e-2
    assert rel_err(dec.conv_decoder[3].weight_orig.grad[:8, :8], g["grad_dec_convT3"]) < 1e-2
    assert rel_err(dec.start_token.grad, g["grad_dec_start_token"]) < 1e-2
    # ---- and against the on-box oracle, full tensors
    o = oracle_step(B, S)
    assert rel_err(r["out"], o["out"]) < 1e-3
    assert math.isclose(float(r["total"]), float(o["total"]), rel_tol=1e-3)
    for tag in ("style", "content", "decoder"):
        rows, num, den = [], 0.0, 0.0
        for k, p in ms[tag].named_parameters():
            ref = o["sds"][tag][k].grad
            if ref is None or p.grad is None:
                continue
            num += float((p.grad.double().cpu() - ref.double()).pow(2).sum())
            den += float(ref.double().pow(2).sum())
            if float(ref.norm()) >= 1e-3:
                rows.append((rel_l2(p.grad, ref), k, float(ref.norm())))
        rows.sort(reverse=True)
        # whole-model gradient.  Run to run (f32 atomic order) the style encoder's value is bimodal -- ~1e-3 in most
        # runs, 5.4-5.8e-3 when one element of the reference's wrapped-phase term (new_decoder.py:377-383) lands on the
        # other side of +-pi (tools/grad_noise.py: 1.4, 1.3, 1.6, 5.8, 0.8, 5.4 e-3 in six runs); content/decoder < 1e-3
        assert math.sqrt(num / den) < 1e-2, (tag, math.sqrt(num / den))
        assert rows[0][0] < 3e-2, (tag, rows[:4])                            # worst single parameter

    if name == "b2s2":   # eval-mode autoregressive decode (config 4 plumbing), after the one training step
        gi = np.load(os.path.join(golden_dir, "infer_b2s2.npz"))
        for m in ms.values():
            m.eval()
        with torch.no_grad():
            se, ce_ = ms["style"](x, labels)
            co = ms["content"](x)
            ar = ms["decoder"](co, ce_[labels.to(DEV)])
        assert rel_err(se, gi["style_emb"]) < 1e-3 and rel_err(co, gi["content_emb"]) < 1e-3
        assert rel_err(ar[:, :, :, ::11, ::13], gi["out_sub"]) < 1e-3


def _surrogate(style, content, out, dev):
    """Smooth (linear) functional of the three model outputs with fixed random weights."""
    gen = torch.Generator().manual_seed(5)
    ws = torch.randn(style.shape, generator=gen).to(dev)
    wc = torch.randn(content.shape, generator=gen).to(dev)
    wo = torch.randn(out.shape, generator=gen).to(dev)
    return (style * ws).sum() + (content * wc).sum() + (out * wo).sum() / 50.0


def test_full_step_bf16_vs_oracle():
    """bf16 MFMA path (throughput mode).  Forward and loss scalars are compared on the real
    losses.  Gradients are compared on a smooth surrogate: the reference's wrapped-phase term
    (new_decoder.py:377-383) is discontinuous at +-pi and weights by 1/|z|^2, so it amplifies
    the 0.3-0.9 % forward rounding of bf16 into O(20 %) gradient changes that say nothing about
    the backward kernels (the f32 test above covers the real loss end to end)."""
    config.set_compute_dtype(torch.bfloat16)
    try:
        B, S = 2, 2
        ms = build_models()
        x = sp.seeded_input(B, S).to(DEV)
        labels = sp.balanced_labels(B)
        y = x[..., :513]
        style, cls = ms["style"](x, labels)
        content = ms["content"](x)
        out = ms["decoder"](content, cls[labels.to(DEV)], y=y)
        rec = ast_amd.compute_comprehensive_loss(out, y)
        _surrogate(style, content, out, DEV).backward()

        sds = {t: OL.seeded_model_state(t) for t in ("style", "content", "decoder")}
        cfg = O.Cfg(training=True, p_drop=0.0)
        xc = x.cpu()
        so, co_ = O.style_encoder_forward(sds["style"], xc, labels, cfg)
        cn = O.content_encoder_forward(sds["content"], xc, cfg)
        oo = O.decoder_forward(sds["decoder"], cn, co_[labels], cfg, y=xc[..., :513])
        reco = O.comprehensive_loss(oo, xc[..., :513])
        _surrogate(so, cn, oo, "cpu").backward()

        assert rel_l2(out, oo) < 3e-2
        assert rel_l2(style, so) < 2e-2 and rel_l2(content, cn) < 2e-2
        for k in ("total_loss", "mse_loss", "mag_loss", "phase_loss", "temporal_loss", "spectral_loss"):
            assert math.isclose(float(rec[k]), float(reco[k]), rel_tol=2e-2), k
        for tag in ("style", "content", "decoder"):
            num = den = 0.0
            for k, p in ms[tag].named_parameters():
                ref = sds[tag][k].grad
                if ref is None or p.grad is None:
                    continue
                num += float((p.grad.double().cpu() - ref.double()).pow(2).sum())
                den += float(ref.double().pow(2).sum())
            assert math.sqrt(num / den) < 0.15, (tag, math.sqrt(num / den))   # bf16 storage of activations AND gradients, 12 conv+BN layers each way
    finally:
        config.set_compute_dtype(torch.float32)


def test_error_surface():
    config.set_compute_dtype(torch.float32)
    dec = ast_amd.Decoder().to(DEV).train()
    c = torch.randn(2, 2, 256, device=DEV)
    with pytest.raises(ValueError):
        dec(c, torch.randn(2, 256, device=DEV), y=torch.randn(2, 2, 287, 513, device=DEV))
    # fresh decoder outputs exactly zero (new_decoder.py:134-143)
    out = dec(c, torch.randn(2, 256, device=DEV), y=torch.randn(2, 2, 2, 287, 513, device=DEV))
    assert out.shape == (2, 2, 2, 287, 513) and float(out.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ast_amd.StyleEncoder()(torch.randn(1, 1, 2, 287, 597))        # CPU tensors: no fallback


@pytest.mark.parametrize("B,S", [(2, 3), (6, 1)])
def test_step_f32_vs_oracle_other_shapes(B, S):
    """BASELINE configs[4] shapes (variable clip length: S = 1..3 sections; an odd number of rows per class): forward,
    every loss scalar and the whole-model gradient against the on-box oracle (no golden file for these shapes)."""
    config.set_compute_dtype(torch.float32)
    ms = build_models()
    x = sp.seeded_input(B, S).to(DEV)
    labels = sp.balanced_labels(B)
    r = hip_step(ms, x, labels)
    o = oracle_step(B, S)
    assert rel_err(r["style"], o["style"]) < 1e-3 and rel_err(r["content"], o["content"]) < 1e-3
    assert rel_err(r["out"], o["out"]) < 1e-3
    for k in ("total_loss", "mse_loss", "mag_loss", "phase_loss", "temporal_loss", "spectral_loss"):
        assert math.isclose(float(r["rec"][k]), float(o["rec"][k]), rel_tol=1e-3, abs_tol=1e-6), k
    assert math.isclose(float(r["total"]), float(o["total"]), rel_tol=1e-3)
    for tag in ("style", "content", "decoder"):
        num = den = 0.0
        for k, p in ms[tag].named_parameters():
            ref = o["sds"][tag][k].grad
            if ref is None or p.grad is None:
                continue
            num += float((p.grad.double().cpu() - ref.double()).pow(2).sum())
            den += float(ref.double().pow(2).sum())
        assert math.sqrt(num / den) < 1e-2, (tag, math.sqrt(num / den))


def test_simple_decoder_f32_vs_golden(golden_dir):
    """SimpleDecoder_TransformerOnly.Decoder against the real reference's fixtures: teacher-forced
    output, the 1.0-MSE-weight loss, gradient norms and strided samples of the two 301 MB weight gradients, and the
    eval-mode autoregressive decode."""
    from ast_amd import SimpleDecoder_TransformerOnly as SD
    g = np.load(os.path.join(golden_dir, "simple_b2s2.npz"), allow_pickle=False)
    ast_amd.set_compute_dtype(torch.float32)
    m = SD.Decoder()
    m.load_state_dict(sp.seeded_state_dict(m.state_dict(), tag="simple_decoder"))
    assert sp.layout_digest(m.state_dict()) == bytes(g["layout_digest"]).decode()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    m = m.to(DEV).train()
    B, S = 2, 2
    content, cls = sp.seeded_normal((B, S, 256), 4101).to(DEV), sp.seeded_normal((B, 256), 4102).to(DEV)
    y = sp.seeded_input(B, S, seed=4103, F=513).to(DEV)
    assert rel_err(m.encode_input(y), torch.from_numpy(g["y_emb"])) < 1e-3
    out = m(content, cls, y=y)
    rec = SD.compute_comprehensive_loss(out, y)
    rec["total_loss"].backward()
    torch.cuda.synchronize()
    assert rel_err(out[:, :, :, ::11, ::13], torch.from_numpy(g["out_sub"])) < 1e-3
    assert math.isclose(float(out.abs().sum()), float(g["out_abs"]), rel_tol=1e-3)
    for k in ("total_loss", "mse_loss", "mag_loss", "phase_loss", "temporal_loss", "spectral_loss"):
        assert math.isclose(float(rec[k]), float(g["rec_" + k]), rel_tol=1e-3, abs_tol=1e-6), k
    worst = 0.0
    for k, p in m.named_parameters():
        ref = float(g["gn/" + k])
        if ref > 1e-6:
            worst = max(worst, abs(float(p.grad.norm()) - ref) / ref)
    assert worst < 3e-2, worst
    assert rel_err(m.stft_to_embedding.weight.grad[::17, ::9973], torch.from_numpy(g["gw_in_sample"])) < 5e-3
    assert rel_err(m.embedding_to_stft.weight.grad[::9973, ::17], torch.from_numpy(g["gw_out_sample"])) < 5e-3
    assert rel_err(m.embedding_to_stft.bias.grad[::9973], torch.from_numpy(g["gb_out_sample"])) < 5e-3
    m.eval()
    with torch.no_grad():
        inf = m(content, cls)
    assert rel_err(inf[:, :, :, ::11, ::13], torch.from_numpy(g["infer_sub"])) < 1e-3


@pytest.mark.parametrize("S", [2, 4])
def test_kv_cached_decode_equals_recompute(S):
    """the KV-cached autoregressive decode (one new token per step against per-layer cached K/V) gives the
    reference loop's result (new_decoder.py:294-314 recomputes the whole stack over all generated tokens every step)."""
    config.set_compute_dtype(torch.float32)
    ms = build_models()
    for m in ms.values():
        m.eval()
    B = 3
    content = sp.seeded_normal((B, S, 256), 991).to(DEV)
    cls = sp.seeded_normal((B, 256), 992).to(DEV)
    dec = ms["decoder"]
    with torch.no_grad():
        dec.decode_mode = "recompute"
        ref = dec(content, cls, target_length=S)
        dec.decode_mode = "kv_cache"
        try:
            got = dec(content, cls, target_length=S)
        finally:
            dec.decode_mode = "recompute"
    assert ref.shape == (B, S, 2, 287, 513) and float(ref.abs().max()) > 0
    assert rel_err(got, ref) < 1e-5
    # and against the oracle's autoregressive decode
    sd = OL.seeded_model_state("decoder", requires_grad=False)
    oo = O.decoder_forward(sd, content.cpu(), cls.cpu(), O.Cfg(training=False), target_length=S)
    assert rel_err(got, oo) < 1e-3


def test_reference_imports_resolve_to_the_dropin(golden_dir):
    """with audio-style-transfer_amd/dropin first on sys.path the reference's own import lines
    (evaluation_style_transfer.py:10-17) bind this implementation, and the step through those names reproduces the
    reference's golden losses (tools/dropin_step.py, a fresh interpreter so the module names are really the shims')."""
    import json, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "dropin_step.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    g = np.load(os.path.join(golden_dir, "model_b2s2.npz"))
    assert r["module"] == "ast_amd.style_encoder" and r["stft_shape"] == [2, 87, 513]
    assert math.isclose(r["total"], float(g["loss_total"]), rel_tol=1e-3)
    assert math.isclose(r["rec"], float(g["rec_total_loss"]), rel_tol=1e-3)
    assert math.isclose(r["adv_d"], float(g["loss_adv_d"]), rel_tol=1e-3)
    assert r["grad"] > 0
