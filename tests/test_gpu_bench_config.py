"""The configuration bench.py times -- train.Trainer, hipGraph replay, three encoder streams + side stream, D phase +
both Adam steps, B=8 clips of S=2 sections (BASELINE configs[1]) -- against the CPU oracle's step on the same seeded
parameters and input.  f32 compute mode at the north-star tolerance (1e-3 rel on loss scalars), bf16 (the mode the
driver's number is quoted in) at tolerances set from the measured per-term errors (tools/bf16_grad_ablation.py,
DESIGN.md section 2).  Dropout off (torch's RNG stream is not part of the contract)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import ast_amd
    from ast_amd import train
from oracle import seeded_params as sp
from oracle.train_step import OracleTrainer

B, S = 8, 2
_ORACLE = {}


def oracle_reference(use_hsic=True):
    """One oracle step at B=8,S=2 (a few seconds of CPU), shared by the dtype cases."""
    if use_hsic not in _ORACLE:
        ot = OracleTrainer()
        x, labels = sp.seeded_input(B, S), sp.balanced_labels(B)
        r = _ORACLE[use_hsic] = {}
        r["losses"] = ot.step(x, labels, apply_g=False, use=OracleTrainer.TERMS if use_hsic else ("rec", "nce", "margin", "adv_g"))
        r["rec_parts"] = {k: float(v) for k, v in ot.rec_parts.items()}
        r["grads"] = ot.raw_grads
        r["gnorm"] = ot.gnorm
        r["disc_after"] = {k: v.detach().clone() for k, v in ot.sds["disc"].items() if v.requires_grad}
    return _ORACLE[use_hsic]


def seeded_trainer(dtype, use_graph=True, use_hsic=True):
    ast_amd.set_compute_dtype(dtype)
    tr = train.Trainer(train.TrainConfig(use_graph=use_graph, dropout=False, keep_grads=True, use_hsic=use_hsic), seed=7)
    for tag, m in (("style", tr.style), ("content", tr.content), ("decoder", tr.decoder), ("disc", tr.disc)):
        # in place: the parameters are views of the trainer's flat buffers
        m.load_state_dict({k: v.to("cuda") for k, v in sp.seeded_state_dict(m.state_dict(), tag=tag).items()})
    return tr


def grad_errors(tr, ref_grads):
    """relative L2 error of the whole-model generator gradient, per model (the gradient Adam consumes: flat_g)."""
    g = tr.last_grad_g
    errs, off = {}, 0
    views = {}
    for p in tr.G.params:
        k = p.numel()
        views[id(p)] = g[off:off + k].view(p.shape)
        off += (k + 3) // 4 * 4
    for tag, m in (("style", tr.style), ("content", tr.content), ("decoder", tr.decoder)):
        num = den = 0.0
        for name, p in m.named_parameters():
            ref = ref_grads[tag][name]
            if ref is None:
                continue
            d = views[id(p)].double().cpu() - ref.double()
            num += float(d.pow(2).sum()); den += float(ref.double().pow(2).sum())
        errs[tag] = math.sqrt(num / den)
    return errs


# tolerances: loss scalars (relative; "*" = every scalar not named), whole-model gradient relative L2 per model
CASES = {
    # f32 = the parity mode: north_star's 1e-3 on every loss scalar (measured 1e-6 .. 5e-6), gradients 1e-2 (measured 2e-3,
    # 2e-4, 4e-4; the oracle-vs-reference re-association noise on gradients is itself 2e-3, tests/test_oracle_golden.py)
    "f32": (torch.float32, {"*": 1e-3}, {"style": 1e-2, "content": 1e-2, "decoder": 1e-2}),
    # bf16 = the throughput mode bench.py times.  Its deviation is that of bf16 STORAGE of activations and activation
    # gradients: the CPU oracle with nothing but that rounding emulated (oracle Cfg.act_dtype) is off by the same amounts
    # (profiles/r02/bf16_ablation.txt: style 0.25 / content 0.11 / decoder 0.09 on the full loss, dominated by the margin
    # and HSIC terms, whose gradients point along the DIFFERENCE of nearby embeddings).  Tolerances = 2x the measured values.
    # Measured at this configuration (GPUTEST log line "[bench-config parity] bf16"): rec 1.4e-4, nce 6e-5, adv_g 2e-4,
    # adv_d 3.8e-4, total 1.9e-3 (it contains the margin term), hsic 3.2e-2 (a 7e-3-sized statistic of embedding
    # differences); gradients style 0.25, content 0.053, decoder 0.055.
    # Round 3 (profiles/r03/bf16_stage_ablation_*.txt): NO stage is responsible -- rounding only the INPUT image to bf16 already
    # gives 0.13 on the margin gradient, every block adds 0.05-0.17, the weights 0.17: the margin / HSIC gradients are differences of
    # nearly equal per-image contributions (||c0 - c1|| = 1.3 against ||c|| = 16 with these seeded parameters), so f32 storage of
    # a few tensors cannot repair them.  Bounds = 2x the values measured over five runs of round 3: style 0.241-0.247, decoder
    # 0.043-0.044; content is BIMODAL, 0.053 (three runs) or 0.164 (one run): HSIC's kernel width is the median of the pairwise
    # embedding distances (losses.py:170-171), and bf16 noise can move the median to the neighbouring pair -- the HSIC value then
    # moves from 1.9-2.4 % to 3.4 % off and its gradient with it.  The second case below takes HSIC out and bounds content tightly.
    "bf16": (torch.bfloat16, {"*": 2e-3, "total": 6e-3, "hsic": 0.07}, {"style": 0.5, "content": 0.33, "decoder": 0.09}),
    "bf16-nohsic": (torch.bfloat16, {"*": 2e-3, "total": 6e-3}, {"style": 0.5, "content": 0.2, "decoder": 0.09}),
}


@pytest.mark.parametrize("mode", ["f32", "bf16", "bf16-nohsic"])
def test_benchmarked_configuration_vs_oracle(mode):
    dtype, tol, gtol = CASES[mode]
    use_hsic = mode != "bf16-nohsic"
    ref = oracle_reference(use_hsic)
    try:
        tr = seeded_trainer(dtype, use_hsic=use_hsic)
        x, labels = sp.seeded_input(B, S).cuda(), sp.balanced_labels(B)
        out = {k: float(v) for k, v in tr.step(x, labels).items()}       # capture + first replay: step 1
        torch.cuda.synchronize()
        assert int(tr.G.step) == 1 and int(tr.D.step) == 1              # capture warm-ups were rolled back
        print(f"[bench-config parity] {mode}: loss rel err " + ", ".join(
            f"{k} {abs(out[k] - ref['losses'][k]) / abs(ref['losses'][k]):.1e}" for k in out))
        for k in ("rec", "nce", "hsic", "adv_d", "adv_g", "total"):
            if k not in out:
                continue
            t = tol.get(k, tol["*"])
            assert math.isclose(out[k], ref["losses"][k], rel_tol=t, abs_tol=1e-5), (mode, k, out[k], ref["losses"][k])
        # the discriminator after ITS optimiser step.  The first Adam update is +-lr per parameter (sign of the gradient):
        # an element only differs where the gradient's sign does -- near-zero gradients -- so all but a few per cent agree
        # (a lost contribution to D's weight gradient, as the same-destination race of round 1 produced, flips ~30 %)
        bad = tot = 0
        for name, p in tr.disc.named_parameters():
            d = (p.detach().cpu() - ref["disc_after"][name]).abs()
            bad += int((d > 1e-5).sum()); tot += d.numel()
        print(f"[bench-config parity] {mode}: discriminator parameters differing after the D step: {bad}/{tot}")
        assert bad <= (0.02 if mode == "f32" else 0.10) * tot, (mode, bad, tot)
        errs = grad_errors(tr, ref["grads"])
        print(f"[bench-config parity] {mode}: losses {out}  grad rel-L2 {errs}")
        for tag, e in errs.items():
            assert e < gtol[tag], (mode, tag, e)
    finally:
        ast_amd.set_compute_dtype(torch.float32)
