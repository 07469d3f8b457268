"""Pins the CPU oracle (oracle/) against fixtures captured from the real
reference modules by tools/make_golden.py.  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ast_oracle as O
from oracle import frontend_oracle as FO
from oracle import layout as L
from oracle import seeded_params as sp

torch.set_num_threads(min(8, os.cpu_count() or 1))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = np.abs(b).max() if b.size else 0.0
    # fp32 re-association noise grows with the tensor's scale: atol is relative to it
    assert np.allclose(a, b, rtol=rtol, atol=atol + rtol * scale), f"max abs err {err}, ref scale {scale}"


def _bias_before_norm(k):
    import re
    return bool(re.search(r"(conv1|conv2|downsample\.0|conv_encoder\.(0|3|6|9)|spatial_projection\.0|"
                          r"conv_decoder\.(0|3|6|9))\.bias$", k))


def test_layout_digests(golden_dir):
    g = _load(golden_dir, "model_b2s2.npz")
    for tag in ("style", "content", "decoder", "disc"):
        assert sp.layout_digest(L.LAYOUTS[tag]()) == str(g[f"digest_{tag}"]), tag


def _oracle_step(B, S):
    sds = {t: L.seeded_model_state(t) for t in ("style", "content", "decoder", "disc")}
    cfg = O.Cfg(training=True, p_drop=0.0)
    x = sp.seeded_input(B, S)
    labels = sp.balanced_labels(B)
    y = x[..., :513]
    style, cls = O.style_encoder_forward(sds["style"], x, labels, cfg)
    content = O.content_encoder_forward(sds["content"], x, cfg)
    out = O.decoder_forward(sds["decoder"], content, cls[labels], cfg, y=y)
    rec = O.comprehensive_loss(out, y)
    nce = O.infonce_loss(style, labels)
    mar = O.margin_loss(cls)
    hs = O.disentanglement_loss(style, content.mean(1))
    cc = O.disentanglement_loss(style, content.mean(1), use_hsic=False)
    d_loss, g_loss = O.adversarial_loss(sds["disc"], style, cls, content, labels, False)
    total = rec["total_loss"] + nce + mar + hs + g_loss
    total.backward()
    return dict(sds=sds, x=x, labels=labels, style=style, cls=cls, content=content, out=out, rec=rec,
                nce=nce, mar=mar, hs=hs, cc=cc, d_loss=d_loss, g_loss=g_loss, total=total)


@pytest.mark.parametrize("name,B,S", [("b2s2", 2, 2), ("b4s1", 4, 1)])
def test_full_step_matches_reference(golden_dir, name, B, S):
    g = _load(golden_dir, f"model_{name}.npz")
    r = _oracle_step(B, S)
    _close(r["style"].detach(), g["style_emb"], 2e-4, 2e-5)
    _close(r["cls"].detach(), g["class_emb"], 2e-4, 2e-5)
    _close(r["content"].detach(), g["content_emb"], 2e-4, 2e-5)
    out = r["out"].detach()
    _close(out[:, :, :, ::11, ::13], g["out_sub"], 2e-4, 2e-5)
    _close(out.sum(dim=(3, 4)), g["out_sum"], 1e-3, 1e-2)
    _close(out.abs().sum(dim=(3, 4)), g["out_abs_sum"], 1e-4, 1e-2)
    for k, v in r["rec"].items():
        assert math.isclose(float(v), float(g["rec_" + k]), rel_tol=1e-4, abs_tol=1e-6), k
    for key, val in (("infonce", r["nce"]), ("margin", r["mar"]), ("hsic", r["hs"]), ("crosscov", r["cc"]),
                     ("adv_d", r["d_loss"]), ("adv_g", r["g_loss"]), ("total", r["total"])):
        assert math.isclose(float(val), float(g["loss_" + key]), rel_tol=2e-4, abs_tol=1e-6), key
    # gradients: per-parameter norms for all three generator-side models
    for tag in ("style", "content", "decoder"):
        keys = [str(k) for k in g[f"gradnorm_keys_{tag}"]]
        vals = g[f"gradnorm_vals_{tag}"]
        sd = r["sds"][tag]
        for k, v in zip(keys, vals):
            gr = sd[k].grad
            got = -1.0 if gr is None else float(gr.norm())
            if v < 0:
                assert gr is None or got == 0.0, k
            elif _bias_before_norm(k):
                # d(loss)/d(bias) of a conv that feeds Batch/InstanceNorm is exactly 0 in
                # real arithmetic; both sides hold rounding noise only
                assert got < 1e-3 and v < 1e-3, (tag, k, got, v)
            else:
                assert math.isclose(got, v, rel_tol=5e-3, abs_tol=1e-6), (tag, k, got, v)
    sds = r["sds"]
    _close(sds["style"]["cnn.net.0.conv1.weight_orig"].grad, g["grad_style_conv1_0"], 5e-3, 1e-6)
    _close(sds["style"]["cnn.proj.weight"].grad[:8], g["grad_style_proj_w"], 5e-3, 1e-6)
    _close(sds["content"]["cnn.5.conv2.weight_orig"].grad[:4, :4], g["grad_content_b5_conv2"], 5e-3, 1e-7)
    _close(sds["decoder"]["conv_decoder.3.weight_orig"].grad[:8, :8], g["grad_dec_convT3"], 5e-3, 1e-7)
    _close(sds["decoder"]["start_token"].grad, g["grad_dec_start_token"], 5e-3, 1e-7)
    # buffers mutated by one training forward
    _close(sds["style"]["cnn.net.0.bn1.running_mean"], g["bn_rm_style_b0_bn1"])
    _close(sds["style"]["cnn.net.0.bn1.running_var"], g["bn_rv_style_b0_bn1"])
    _close(sds["decoder"]["conv_encoder.1.running_mean"], g["bn_rm_dec_ce1"])
    _close(sds["decoder"]["conv_decoder.10.running_var"], g["bn_rv_dec_cd10"])
    _close(sds["style"]["cnn.net.0.conv1.weight_u"], g["sn_u_style_b0_conv1"])
    _close(sds["style"]["cnn.net.5.conv2.weight_v"], g["sn_v_style_b5_conv2"])
    _close(sds["decoder"]["conv_decoder.3.weight_u"], g["sn_u_dec_cd3"])
    _close(sds["decoder"]["conv_decoder.3.weight_v"], g["sn_v_dec_cd3"])
    assert int(sds["style"]["cnn.net.0.bn1.num_batches_tracked"]) == 1

    if name == "b2s2":
        gi = _load(golden_dir, "infer_b2s2.npz")
        cfg = O.Cfg(training=False)
        with torch.no_grad():
            st, cl = O.style_encoder_forward(sds["style"], r["x"], r["labels"], cfg)
            co = O.content_encoder_forward(sds["content"], r["x"], cfg)
            ar = O.decoder_forward(sds["decoder"], co, cl[r["labels"]], cfg)
        _close(st, gi["style_emb"], 2e-4, 2e-5)
        _close(co, gi["content_emb"], 2e-4, 2e-5)
        _close(ar[:, :, :, ::11, ::13], gi["out_sub"], 2e-4, 2e-5)


@pytest.mark.parametrize("B", [8, 16])
def test_losses_and_gradients(golden_dir, B):
    g = _load(golden_dir, "losses.npz")
    rng = np.random.default_rng([77, B])
    style = torch.tensor(rng.standard_normal((B, 256)).astype(np.float32), requires_grad=True)
    content = torch.tensor(rng.standard_normal((B, 3, 256)).astype(np.float32), requires_grad=True)
    labels = sp.balanced_labels(B)
    disc = L.seeded_model_state("disc")
    cls = torch.stack([style[labels == 0].mean(0), style[labels == 1].mean(0)])
    cm = content.mean(1)
    d_loss, g_loss = O.adversarial_loss(disc, style, cls, content, labels, False)
    vals = {"infonce": O.infonce_loss(style, labels), "margin": O.margin_loss(cls),
            "hsic": O.disentanglement_loss(style, cm), "crosscov": O.disentanglement_loss(style, cm, False),
            "adv_d": d_loss, "adv_g": g_loss}
    for k, v in vals.items():
        assert math.isclose(float(v), float(g[f"B{B}_{k}"]), rel_tol=1e-4, abs_tol=1e-7), (k, float(v))
        gs, gc = torch.autograd.grad(v, [style, content], retain_graph=True, allow_unused=True)
        if gs is not None:
            _close(gs, g[f"B{B}_{k}_dstyle"], 2e-3, 1e-7)
        if gc is not None:
            _close(gc, g[f"B{B}_{k}_dcontent"], 2e-3, 1e-7)


def test_known_answers(golden_dir):
    g = _load(golden_dir, "losses.npz")
    # test_correctness.ipynb cell 9: identical embeddings, B=16 -> ln 15
    v = float(O.infonce_loss(torch.ones(16, 256), sp.balanced_labels(16)))
    assert math.isclose(v, math.log(15), rel_tol=1e-5)
    assert math.isclose(float(g["kat_infonce_identical_B16"]), math.log(15), rel_tol=1e-5)
    # uniform logits (all-zero discriminator): D = 2.5 ln2, G = -ln2 (cell 9: 1.7329 / -0.6931)
    disc = {k: torch.zeros_like(v) for k, v in L.discriminator_layout().items()}
    e = torch.randn(4, 256)
    d, gl = O.adversarial_loss(disc, e, e[:2], e, sp.balanced_labels(4), False)
    assert math.isclose(float(d), 2.5 * math.log(2), rel_tol=1e-5)
    assert math.isclose(float(gl), -math.log(2), rel_tol=1e-5)
    for B in (2, 4, 8, 32):
        m = 2 * B
        assert O.hsic_sigma_rank(B) == m * m // 2 - 1


def test_frontend(golden_dir):
    g = _load(golden_dir, "frontend.npz")
    w = FO.synth_waveform(0, "piano", 4.0)
    st = FO.stft(w)
    assert tuple(g["stft_shape"]) == st.shape == (2, 345, 513)
    _close(st[:, :6], g["stft_piano0_head"], 1e-4, 2e-5)
    _close(st[:, ::23, ::17], g["stft_piano0_sub"], 1e-4, 2e-5)
    _close(st.sum(axis=1), g["stft_piano0_sum"], 1e-3, 2e-3)
    _close(FO.stft(FO.synth_waveform(1, "violin", 4.0))[:, ::23, ::17], g["stft_violin1_sub"], 1e-4, 2e-5)
    rec = FO.istft(st)
    assert rec.shape[0] == int(g["istft_len"])
    _close(rec[::97], g["istft_piano0_sub"], 1e-4, 1e-6)
    for s, T, n in zip(g["win_secs"], g["win_frames"], g["win_nsec"]):
        assert len(FO.section_starts(int(T))) == int(n), (s, T, n)
    for s in (4, 6):
        T = 1 + (s * 22050) // 256
        spec = np.arange(2 * T * 3, dtype=np.float32).reshape(2, T, 3)
        win = FO.overlap_windows(spec)
        _close(win, g[f"windows_{s}s"], 0, 0)
        _close(FO.sections_to_spectrogram(win, T), g[f"recon_{s}s"], 1e-6, 1e-6)


def test_simple_decoder_oracle_vs_reference(golden_dir):
    """SimpleDecoder_TransformerOnly.Decoder: oracle restatement against the real reference's
    teacher-forced step (output, loss, gradients incl. strided samples of the two 301 MB weight gradients) and its
    eval-mode autoregressive decode."""
    g = _load(golden_dir, "simple_b2s2.npz")
    sd = L.seeded_model_state("simple_decoder")
    assert sp.layout_digest(L.LAYOUTS["simple_decoder"]()) == bytes(g["layout_digest"]).decode()
    B, S = 2, 2
    content, cls = sp.seeded_normal((B, S, 256), 4101), sp.seeded_normal((B, 256), 4102)
    y = sp.seeded_input(B, S, seed=4103, F=513)
    out = O.simple_decoder_forward(sd, content, cls, O.Cfg(training=True, p_drop=0.0), y=y)
    rec = O.comprehensive_loss(out, y, mse_weight=1.0)          # SimpleDecoder_TransformerOnly.py:194
    rec["total_loss"].backward()
    _close(out[:, :, :, ::11, ::13].detach().numpy(), g["out_sub"])
    assert math.isclose(float(out.abs().sum()), float(g["out_abs"]), rel_tol=1e-4)
    for k in ("total_loss", "mse_loss", "mag_loss", "phase_loss", "temporal_loss", "spectral_loss"):
        assert math.isclose(float(rec[k]), float(g["rec_" + k]), rel_tol=1e-4, abs_tol=1e-6), k
    for k, v in sd.items():
        if ("gn/" + k) in g.files:
            assert math.isclose(float(v.grad.norm()), float(g["gn/" + k]), rel_tol=5e-3, abs_tol=1e-6), k
    _close(sd["stft_to_embedding.weight"].grad[::17, ::9973].numpy(), g["gw_in_sample"], rtol=5e-3)
    _close(sd["embedding_to_stft.weight"].grad[::9973, ::17].numpy(), g["gw_out_sample"], rtol=5e-3)
    _close(sd["embedding_to_stft.bias"].grad[::9973].numpy(), g["gb_out_sample"], rtol=5e-3)
    with torch.no_grad():
        inf = O.simple_decoder_forward(sd, content, cls, O.Cfg(training=False, p_drop=0.0))
    _close(inf[:, :, :, ::11, ::13].numpy(), g["infer_sub"])
