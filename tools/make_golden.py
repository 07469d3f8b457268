#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference modules
(/root/reference, read-only) on CPU with the seeded parameter recipe of
oracle/seeded_params.py.

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/make_golden.py

Only *data* is written (inputs are re-derivable from seeds; outputs are small
sub-samples, sums, loss scalars and gradient norms).  No reference source is
copied anywhere.  `utilityFunctions.py` imports torchaudio/librosa at module
scope (neither is installed); empty placeholder modules are registered so the
torch-only functions (get_STFT, get_overlap_windows, sections2spectrogram,
inverse_STFT) can be executed -- get_CQT/load_audio are never called and CQT
stays parity-unpinned.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn as nn

from oracle import seeded_params as sp
from oracle import frontend_oracle as fo

torch.manual_seed(0)
torch.set_num_threads(8)
OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

import style_encoder as r_se          # noqa: E402  (reference)
import content_encoder as r_ce        # noqa: E402
import new_decoder as r_dec           # noqa: E402
import discriminator as r_disc        # noqa: E402
import losses as r_losses             # noqa: E402


def no_dropout(m: nn.Module):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0


def build(tag):
    ctor = {"style": r_se.StyleEncoder, "content": r_ce.ContentEncoder,
            "decoder": r_dec.Decoder, "disc": r_disc.Discriminator}[tag]
    m = ctor()
    sd = m.state_dict()
    m.load_state_dict(sp.seeded_state_dict(sd, tag=tag))
    no_dropout(m)
    return m, sp.layout_digest(sd)


def sub(t):
    """Small deterministic sub-sample of a (B,S,2,287,513) tensor."""
    return t[:, :, :, ::11, ::13].detach().numpy().copy()


def grad_norms(m):
    return {k: float(p.grad.norm()) if p.grad is not None else -1.0 for k, p in m.named_parameters()}


def run_model_config(B, S, name):
    style, dg_s = build("style")
    content, dg_c = build("content")
    dec, dg_d = build("decoder")
    disc, dg_k = build("disc")
    for m in (style, content, dec, disc):
        m.train()
    x = sp.seeded_input(B, S)
    labels = sp.balanced_labels(B)
    y = x[..., :513]

    style_emb, class_emb = style(x, labels)
    content_emb = content(x)
    out = dec(content_emb, class_emb[labels], y=y)
    rec = r_dec.compute_comprehensive_loss(out, y)
    nce = r_losses.infoNCE_loss(style_emb, labels)
    mar = r_losses.margin_loss(class_emb)
    hs = r_losses.disentanglement_loss(style_emb, content_emb.mean(1))
    cc = r_losses.disentanglement_loss(style_emb, content_emb.mean(1), use_hsic=False)
    d_loss, g_loss = r_losses.adversarial_loss(style_emb, class_emb, content_emb, disc, labels,
                                               compute_for_discriminator=False)
    total = rec["total_loss"] + nce + mar + hs + g_loss
    total.backward()

    g = {"digest_style": dg_s, "digest_content": dg_c, "digest_decoder": dg_d, "digest_disc": dg_k,
         "B": B, "S": S,
         "style_emb": style_emb.detach().numpy(), "class_emb": class_emb.detach().numpy(),
         "content_emb": content_emb.detach().numpy(),
         "out_sub": sub(out), "out_sum": out.detach().sum(dim=(3, 4)).numpy(),
         "out_abs_sum": out.detach().abs().sum(dim=(3, 4)).numpy(),
         "loss_infonce": float(nce), "loss_margin": float(mar), "loss_hsic": float(hs),
         "loss_crosscov": float(cc), "loss_adv_d": float(d_loss), "loss_adv_g": float(g_loss),
         "loss_total": float(total)}
    for k, v in rec.items():
        g["rec_" + k] = float(v)
    for tag, m in (("style", style), ("content", content), ("decoder", dec)):
        gn = grad_norms(m)
        g[f"gradnorm_keys_{tag}"] = np.array(sorted(gn.keys()))
        g[f"gradnorm_vals_{tag}"] = np.array([gn[k] for k in sorted(gn.keys())], dtype=np.float64)
    # a few raw gradient slices
    g["grad_style_conv1_0"] = style.cnn.net[0].conv1.weight_orig.grad.numpy()
    g["grad_style_proj_w"] = style.cnn.proj.weight.grad.numpy()[:8]
    g["grad_content_b5_conv2"] = content.cnn[5].conv2.weight_orig.grad.numpy()[:4, :4]
    g["grad_dec_convT3"] = dec.conv_decoder[3].weight_orig.grad.numpy()[:8, :8]
    g["grad_dec_start_token"] = dec.start_token.grad.numpy()
    # buffers mutated by one training forward
    g["bn_rm_style_b0_bn1"] = style.cnn.net[0].bn1.running_mean.numpy()
    g["bn_rv_style_b0_bn1"] = style.cnn.net[0].bn1.running_var.numpy()
    g["bn_rm_dec_ce1"] = dec.conv_encoder[1].running_mean.numpy()
    g["bn_rv_dec_cd10"] = dec.conv_decoder[10].running_var.numpy()
    g["sn_u_style_b0_conv1"] = style.cnn.net[0].conv1.weight_u.numpy()
    g["sn_v_style_b5_conv2"] = style.cnn.net[5].conv2.weight_v.numpy()
    g["sn_u_dec_cd3"] = dec.conv_decoder[3].weight_u.numpy()
    g["sn_v_dec_cd3"] = dec.conv_decoder[3].weight_v.numpy()
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **g)
    print(name, "total", float(total), {k: float(v) for k, v in rec.items()})

    if name == "b2s2":
        # eval-mode autoregressive decode with the (now once-updated) buffers
        for m in (style, content, dec):
            m.eval()
        with torch.no_grad():
            se, ce_ = style(x, labels)
            co = content(x)
            ar = dec(co, ce_[labels])
        np.savez_compressed(os.path.join(OUT, "infer_b2s2.npz"),
                            style_emb=se.numpy(), content_emb=co.numpy(), out_sub=sub(ar),
                            out_sum=ar.sum(dim=(3, 4)).numpy())
        print("infer ok", float(ar.abs().mean()))


def run_simple_decoder(B=2, S=2):
    """SimpleDecoder_TransformerOnly.Decoder: teacher-forced forward + loss + backward, eval-mode
    autoregressive decode.  Inputs are seeded (content / class embeddings as N(0,1) rows, y as seeded_input)."""
    import SimpleDecoder_TransformerOnly as r_simple          # noqa: E402  (reference)
    m = r_simple.Decoder()
    sd = m.state_dict()
    m.load_state_dict(sp.seeded_state_dict(sd, tag="simple_decoder"))
    no_dropout(m)
    g = {"layout_digest": np.frombuffer(sp.layout_digest(sd).encode(), dtype=np.uint8)}
    content = sp.seeded_normal((B, S, 256), 4101)
    cls = sp.seeded_normal((B, 256), 4102)
    y = sp.seeded_input(B, S, seed=4103, F=513)
    m.train()
    emb = m.encode_input(y)
    out = m(content, cls, y=y)
    rec = r_simple.compute_comprehensive_loss(out, y)
    rec["total_loss"].backward()
    g["y_emb"] = emb.detach().numpy()
    g["out_sub"], g["out_sum"], g["out_abs"] = sub(out), float(out.sum()), float(out.abs().sum())
    for k, v in rec.items():
        g["rec_" + k] = float(v)
    for k, p in m.named_parameters():
        g["gn/" + k] = float(p.grad.norm())
    # a strided sample of the two big gradients (the full tensors are 301 MB each)
    g["gw_in_sample"] = m.stft_to_embedding.weight.grad[::17, ::9973].numpy().copy()
    g["gw_out_sample"] = m.embedding_to_stft.weight.grad[::9973, ::17].numpy().copy()
    g["gb_out_sample"] = m.embedding_to_stft.bias.grad[::9973].numpy().copy()
    m.eval()
    with torch.no_grad():
        inf = m(content, cls)
    g["infer_sub"], g["infer_sum"] = sub(inf), float(inf.sum())
    np.savez_compressed(os.path.join(OUT, f"simple_b{B}s{S}.npz"), **g)
    print("simple decoder: loss", float(rec["total_loss"]), "out_abs", g["out_abs"])


def run_losses():
    g = {}
    disc, _ = build("disc")
    for B in (8, 16):
        rng = np.random.default_rng([77, B])
        style = torch.tensor(rng.standard_normal((B, 256)).astype(np.float32), requires_grad=True)
        content = torch.tensor(rng.standard_normal((B, 3, 256)).astype(np.float32), requires_grad=True)
        labels = sp.balanced_labels(B)
        cls = torch.stack([style[labels == 0].mean(0), style[labels == 1].mean(0)])
        cm = content.mean(1)
        vals = {
            "infonce": r_losses.infoNCE_loss(style, labels),
            "margin": r_losses.margin_loss(cls),
            "hsic": r_losses.disentanglement_loss(style, cm),
            "crosscov": r_losses.disentanglement_loss(style, cm, use_hsic=False),
        }
        d_loss, g_loss = r_losses.adversarial_loss(style, cls, content, disc, labels, False)
        vals["adv_d"], vals["adv_g"] = d_loss, g_loss
        for k, v in vals.items():
            gs, gc = torch.autograd.grad(v, [style, content], retain_graph=True, allow_unused=True)
            g[f"B{B}_{k}"] = float(v)
            g[f"B{B}_{k}_dstyle"] = np.zeros((B, 256), np.float32) if gs is None else gs.numpy()
            g[f"B{B}_{k}_dcontent"] = np.zeros((B, 3, 256), np.float32) if gc is None else gc.numpy()
    # analytic known answers recorded in test_correctness.ipynb cell 9
    e = torch.ones(16, 256)
    g["kat_infonce_identical_B16"] = float(r_losses.infoNCE_loss(e, sp.balanced_labels(16)))
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **g)
    print("losses ok", g["kat_infonce_identical_B16"])


def run_frontend():
    for name in ("torchaudio", "librosa", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import utilityFunctions as uf   # reference
    g = {}
    w = fo.synth_waveform(0, "piano", 4.0)
    st = uf.get_STFT(torch.from_numpy(w)[None])
    g["stft_shape"] = np.array(st.shape)
    g["stft_piano0_head"] = st[:, :6].contiguous().numpy()
    g["stft_piano0_sub"] = st[:, ::23, ::17].contiguous().numpy()
    g["stft_piano0_sum"] = st.sum(dim=1).numpy()
    wv = fo.synth_waveform(1, "violin", 4.0)
    stv = uf.get_STFT(torch.from_numpy(wv))
    g["stft_violin1_sub"] = stv[:, ::23, ::17].contiguous().numpy()
    rec = uf.inverse_STFT(st.contiguous())
    g["istft_len"] = rec.shape[0]
    g["istft_piano0_sub"] = rec[::97].numpy()
    # window counts / tail rule for 2..10 s
    secs, frames, nsec = [], [], []
    for s in (2, 3, 4, 5, 6, 7, 8, 10):
        T = 1 + (s * 22050) // 256
        spec = torch.arange(2 * T * 3, dtype=torch.float32).view(2, T, 3)
        win = uf.get_overlap_windows(spec)
        secs.append(s); frames.append(T); nsec.append(win.shape[0])
        if s in (4, 6):
            g[f"windows_{s}s"] = win.numpy()
            g[f"recon_{s}s"] = uf.sections2spectrogram(win, T).numpy()
    g["win_secs"], g["win_frames"], g["win_nsec"] = np.array(secs), np.array(frames), np.array(nsec)
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), **g)
    print("frontend ok", dict(zip(secs, nsec)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["frontend", "losses", "b2s2", "b4s1"]
    if "frontend" in which:
        run_frontend()
    if "losses" in which:
        run_losses()
    if "b2s2" in which:
        run_model_config(2, 2, "b2s2")
    if "b4s1" in which:
        run_model_config(4, 1, "b4s1")
    if "simple" in which:
        run_simple_decoder(2, 2)
    assert not os.path.exists(os.path.join(REF, "__pycache__")), "bytecode leaked into the reference tree"
