"""Token op lists (ast_tok_program, csrc/tokprog.hip, ast_amd/tokprog.py: one autograd node per transformer STACK) against
the per-operator path they replace, in both execution modes (config.tok_programs / AST_TOK_PROGRAMS: 1 = one launch per
op, the default; 2 = persistent program launches, experimental) (ast_skinny_gemm / ast_attn_* / ast_add_drop_ln_*, themselves checked against the oracle in
test_gpu_ops.py / test_gpu_models.py): same outputs, same input and parameter gradients, to f32 re-association noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import ast_amd
    from ast_amd import config, tokprog
from oracle import seeded_params as sp

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(b.norm().item(), 1e-12))


def _model(ctor, tag, p_drop):
    m = ctor()
    m.load_state_dict(sp.seeded_state_dict(m.state_dict(), tag=tag))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p_drop
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = p_drop
    return m.to(DEV).train()


def _run(m, fn, use_programs):
    old = config.tok_programs
    config.tok_programs = int(use_programs)
    try:
        for p in m.parameters():
            p.grad = None
        outs, ins = fn(m)
        loss = sum((o * torch.linspace(-1.0, 1.0, o.numel(), device=DEV).view_as(o)).sum() for o in outs)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        return [o.detach().clone() for o in outs], [i.grad.detach().clone() for i in ins], grads
    finally:
        config.tok_programs = old


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,L", [(8, 3), (8, 2), (2, 2), (16, 4), (4, 1), (7, 8)])
def test_encoder_stacks_match_per_operator_path(B, L, mode):
    """The TransformerEncoder stacks of StyleEncoder (CLS token: L = S + 1) and ContentEncoder (L = S) on a random token
    sequence, dropout off: output, input gradient and every parameter gradient."""
    from ast_amd.style_encoder import _module_bank
    config.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(3)
    seq0 = torch.randn(B, L, 256, generator=g).to(DEV)
    for ctor, tag in ((ast_amd.StyleEncoder, "style"), (ast_amd.ContentEncoder, "content")):
        m = _model(ctor, tag, 0.0)

        def fn(mm):
            seq = seq0.clone().requires_grad_(True)
            _module_bank(mm).prepare(True)
            assert tokprog.encoder_stack_ok(seq, mm._layers)
            if config.tok_programs:
                out = tokprog.encoder_stack(seq, mm._layers, True, 0)
            else:
                out = seq
                for lyr in mm._layers:
                    out = lyr(out, True)
            return [out], [seq]
        o1, i1, g1 = _run(m, fn, mode)
        o0, i0, g0 = _run(m, fn, 0)
        assert rel_l2(o1[0], o0[0]) < 1e-5
        assert rel_l2(i1[0], i0[0]) < 2e-4
        assert set(g1) == set(g0) and len(g0) >= 4 * 12
        for k in g0:
            assert rel_l2(g1[k], g0[k]) < 2e-4, k
    tokprog.check_status()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,S", [(8, 2), (2, 4), (16, 3)])
def test_decoder_stack_matches_per_operator_path(B, S, mode):
    """The TransformerDecoder stack (pre-norm, causal self-attention, cross-attention over the 2 S memory rows) with
    gradients to the target embeddings, the memory and every parameter."""
    config.set_compute_dtype(torch.float32)
    m = _model(ast_amd.Decoder, "decoder", 0.0)
    g = torch.Generator().manual_seed(5)
    tgt0 = torch.randn(B, S, 256, generator=g).to(DEV)
    mem0 = torch.randn(B, 2 * S, 256, generator=g).to(DEV)

    def fn(mm):
        tgt, mem = tgt0.clone().requires_grad_(True), mem0.clone().requires_grad_(True)
        mm._prepare()
        return [mm._stack(tgt, mem)], [tgt, mem]
    o1, i1, g1 = _run(m, fn, mode)
    o0, i0, g0 = _run(m, fn, 0)
    assert rel_l2(o1[0], o0[0]) < 1e-5
    for a, b in zip(i1, i0):
        assert rel_l2(a, b) < 2e-4
    assert set(g1) == set(g0) and g0
    for k in g0:
        assert rel_l2(g1[k], g0[k]) < 1e-3, k
    tokprog.check_status()


@pytest.mark.parametrize("mode", [1, 2])
def test_stacks_with_dropout_match_per_operator_path(mode):
    """Dropout on (attention probabilities, residual branches, FFN hidden units): the op lists draw their masks from the same
    (seed, call number, element index) streams as the per-operator kernels, so with the call counter rewound the two paths
    must agree -- outputs AND gradients -- which also proves that the backward ops use the masks their forward drew."""
    from ast_amd import ops
    from ast_amd.style_encoder import _module_bank
    config.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(9)
    B, S = 8, 2
    tgt0 = torch.randn(B, S, 256, generator=g).to(DEV)
    mem0 = torch.randn(B, 2 * S, 256, generator=g).to(DEV)
    seq0 = torch.randn(B, S + 1, 256, generator=g).to(DEV)
    ops._DropState.counter = torch.full((1,), 37, dtype=torch.int64, device=DEV)       # a step counter in mid-training

    dec = _model(ast_amd.Decoder, "decoder", 0.1)
    enc = _model(ast_amd.StyleEncoder, "style", 0.1)

    def dec_fn(mm):
        ops._DropState.calls = 1000
        tgt, mem = tgt0.clone().requires_grad_(True), mem0.clone().requires_grad_(True)
        mm._prepare()
        return [mm._stack(tgt, mem)], [tgt, mem]

    def enc_fn(mm):
        ops._DropState.calls = 2000
        seq = seq0.clone().requires_grad_(True)
        _module_bank(mm).prepare(True)
        if config.tok_programs > 0:
            out = tokprog.encoder_stack(seq, mm._layers, True, 0)
        else:
            out = seq
            for lyr in mm._layers:
                out = lyr(out, True)
        return [out], [seq]
    for m, fn in ((dec, dec_fn), (enc, enc_fn)):
        o1, i1, g1 = _run(m, fn, mode)
        o0, i0, g0 = _run(m, fn, 0)
        assert rel_l2(o1[0], o0[0]) < 1e-5
        for a, b in zip(i1, i0):
            assert rel_l2(a, b) < 2e-4
        assert set(g1) == set(g0) and g0
        for k in g0:
            assert rel_l2(g1[k], g0[k]) < 2e-4, k
    # and the masks were really there: the same call with dropout off gives another output
    off = _model(ast_amd.Decoder, "decoder", 0.0)
    o_off, _, _ = _run(off, dec_fn, mode)
    o_on, _, _ = _run(dec, dec_fn, mode)
    assert rel_l2(o_on[0], o_off[0]) > 1e-2
    tokprog.check_status()
