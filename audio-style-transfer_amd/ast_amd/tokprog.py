"""Token programs: the transformer stacks of the model run as a few launches of libast_hip's program kernel
(include/ast_hip.h: ast_tok_program, csrc/tokprog.hip) instead of one launch per operator.

The reference builds its stacks from nn.TransformerEncoderLayer (post-norm, ReLU; style_encoder.py:181-191,
content_encoder.py:24-26) and nn.TransformerDecoderLayer(norm_first=True) (new_decoder.py:49-51,111-119).  On the <= 64
token rows of this model every operator of a layer is a 2-9 us launch-latency-bound kernel and a dependent node of the
replayed graph; a layer is 7 (encoder) / 13 (decoder) of them forward and as many backward.  Here the host writes the
same operator sequence as a list of ops over one activation buffer and the device walks it with grid barriers.

Autograd: one Function per STACK.  Its backward runs the reverse program (data gradients, LayerNorm parameter
gradients in place) and hands the (dy, x) pairs of the linears to the bank's batched weight-gradient launch, exactly
as LinearFn / FFNFn do; gradients that two consumers of one tensor used to leave to the autograd engine's add are
merged by the `addend` operand of the GEMM that produces the second one.
"""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import check, lib, ptr, stream

GEMM, ATTN_FWD, ATTN_BWD, ADLN_FWD, ADLN_BWD = 1, 2, 3, 4, 5
RELU, NO_BARRIER, CAUSAL = 1, 2, 4
G_WORKGROUPS = 32          # every CU of one XCD


class TokOp(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("flags", ctypes.c_int32), ("i", ctypes.c_int32 * 8), ("p", ctypes.c_float),
                ("eps", ctypes.c_float), ("seed", ctypes.c_uint64), ("inp", ctypes.c_void_p * 7), ("out", ctypes.c_void_p * 5)]


def _a(t):
    """device address of a tensor / (tensor, element offset) / raw int / None"""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if isinstance(t, tuple):
        return t[0].data_ptr() + 4 * t[1]
    return t.data_ptr()


def _op(kind, flags, ints, inp, out, p=0.0, eps=0.0, seed=0):
    o = TokOp()
    o.type, o.flags, o.p, o.eps, o.seed = kind, flags, p, eps, seed
    for k, v in enumerate(ints):
        o.i[k] = int(v)
    for k, v in enumerate(inp):
        o.inp[k] = _a(v)
    for k, v in enumerate(out):
        o.out[k] = _a(v)
    return o


def gemm(x, w, bias, y, M, N, K, ldx, ldw, ldy, relu=False, mul_mask=None, addend=None, drop_mask=None, p=0.0, seed=0, flags=0):
    return _op(GEMM, flags | (RELU if relu else 0), (M, N, K, ldx, ldw, ldy), (x, w, bias, mul_mask, addend), (y, drop_mask), p=p, seed=seed)


def _next_seed():
    ops._DropState.calls += 1
    return ops._DropState.seed + 7919 * ops._DropState.calls


_sync = {}


def _sync_for(device, xcd):
    key = (str(device), xcd, torch.cuda.current_stream(device).cuda_stream)
    s = _sync.get(key)
    if s is None:
        s = _sync[key] = (torch.zeros(32, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32, device=device))
    return s


def check_status():
    """Raise if any program launch so far gave up on a grid barrier (synchronises; call after warm-up, not per step)."""
    for (dev, xcd, _), (_, status) in _sync.items():
        st = int(status.item())
        if st != 0:
            status.zero_()
            why = {1: "a grid barrier timed out", 3: "fewer than G workgroups landed on the target XCD"}.get(
                st, "the workgroups were not placed on one XCD (blockIdx % 8 != XCC_ID)")
            raise RuntimeError(f"ast_tok_program: {why} on {dev} (xcd {xcd}, status {st}); results are invalid")


def run(oplist, device, xcd, chunk_ends=None, persistent=None):
    """Execute the ops: one kernel per op (config.tok_programs == 1), or persistent launches of at most ast_tok_max_ops()
    ops (== 2; `chunk_ends` = indices after which a launch may end)."""
    if not oplist:
        return
    from . import config
    if persistent is None:
        persistent = config.tok_programs == 2
    ctr = ops._DropState.counter
    if ctr is None or ctr.device != device:
        ctr = ops._DropState.counter = torch.zeros(1, dtype=torch.int64, device=device)
    if not persistent:
        cap = lib().ast_tok_max_ops()
        for start in range(0, len(oplist), cap):
            part = oplist[start:start + cap]
            arr = (TokOp * len(part))(*part)
            check(lib().ast_tok_program(arr, len(part), 0, 0, None, None, ptr(ctr), stream()), "ast_tok_program")
        return
    sync, status = _sync_for(device, xcd)
    cap = lib().ast_tok_max_ops()
    ends = sorted(set(chunk_ends or range(1, len(oplist) + 1)) | {len(oplist)})
    start = 0
    while start < len(oplist):
        fit = [e for e in ends if start < e <= start + cap]
        end = max(fit) if fit else min(start + cap, len(oplist))
        n = end - start
        arr = (TokOp * n)(*oplist[start:end])
        check(lib().ast_tok_program(arr, n, G_WORKGROUPS, xcd, ptr(sync), ptr(status), ptr(ctr), stream()), "ast_tok_program")
        start = end


# ------------------------------------------------------------------------------------------------------------------
def _pw_ok(pw, rows):
    return ops._skinny_ok(pw, rows) and pw.Co % 64 == 0 and pw.Ci % 64 == 0 and pw.s_co % 4 == 0


def encoder_stack_ok(x, layers):
    if x.dtype != torch.float32 or x.dim() != 3 or not layers:
        return False
    B, L, d = x.shape
    rows = B * L
    if d != 256 or rows > 64 or L > 8:
        return False
    for lyr in layers:
        a = lyr.attn
        if a.cross or d // a.h > 64 or not all(_pw_ok(pw, rows) for pw in (a.qkv, a.out, lyr.ff1, lyr.ff2)):
            return False
        if lyr.l.norm1.weight.numel() != d or lyr.l.norm1.weight.dtype != torch.float32:
            return False
    return True


class _Carver:
    """Views of one flat f32 allocation (one allocator call per stack instead of ~15 per layer)."""

    def __init__(self, device):
        self.device, self.sizes = device, []

    def want(self, *shape):
        n = 1
        for s in shape:
            n *= s
        n = (n + 3) // 4 * 4                      # keep every view 16-byte aligned
        self.sizes.append((shape, n))
        return len(self.sizes) - 1

    def alloc(self):
        total = sum(n for _, n in self.sizes)
        self.buf = torch.empty(total, dtype=torch.float32, device=self.device)
        self.views, off = [], 0
        for shape, n in self.sizes:
            m = 1
            for s in shape:
                m *= s
            self.views.append(self.buf[off:off + m].view(*shape))
            off += n
        return self.views


class EncoderStackFn(torch.autograd.Function):
    """x (B,L,256) f32 -> the output of `layers` (layers.EncoderLayer objects, nn.TransformerEncoderLayer defaults:
    post-norm, ReLU FFN; style_encoder.py:181-187)."""

    @staticmethod
    def forward(ctx, x, anchor, layers, training, xcd):
        x = x.contiguous()
        B, L, d = x.shape
        rows = B * L
        dev = x.device
        cv = _Carver(dev)
        slots = []
        for lyr in layers:
            a, l = lyr.attn, lyr.l
            F = lyr.ff1.Co
            p_attn = float(l.self_attn.dropout) if training else 0.0
            p1 = float(l.dropout1.p) if training else 0.0
            pf = float(l.dropout.p) if training else 0.0
            p2 = float(l.dropout2.p) if training else 0.0
            s = dict(qkv=cv.want(rows, 3 * d), probs=cv.want(B, a.h, L, L), o=cv.want(rows, d), a=cv.want(rows, d),
                     s1=cv.want(rows, d), x1=cv.want(rows, d), st1=cv.want(2, rows), h=cv.want(rows, F), hmask=cv.want(rows, F),
                     f=cv.want(rows, d), s2=cv.want(rows, d), y=cv.want(rows, d), st2=cv.want(2, rows),
                     m1=cv.want(rows, d) if p1 > 0 else None, m2=cv.want(rows, d) if p2 > 0 else None,
                     p=(p_attn, p1, pf, p2))
            slots.append(s)
        v = cv.alloc()
        prog, ends = [], []
        xin = x
        seeds = []
        for lyr, s in zip(layers, slots):
            a, l = lyr.attn, lyr.l
            dh = d // a.h
            p_attn, p1, pf, p2 = s["p"]
            sd = [_next_seed() if pp > 0 else 0 for pp in (p_attn, p1, pf, p2)]
            seeds.append(sd)
            qkv, o, aa, x1, h, f, y = v[s["qkv"]], v[s["o"]], v[s["a"]], v[s["x1"]], v[s["h"]], v[s["f"]], v[s["y"]]
            W = lambda pw: pw.weight.data_ptr() + 4 * pw.w_off
            Bs = lambda pw: pw.bias.data_ptr() + 4 * pw.b_off
            prog.append(gemm(xin, W(a.qkv), Bs(a.qkv), qkv, rows, 3 * d, d, d, a.qkv.s_co, 3 * d))
            prog.append(_op(ATTN_FWD, 0, (B, a.h, L, L, dh, 3 * d, 3 * d, d), (qkv, (qkv, d), (qkv, 2 * d)), (o, v[s["probs"]]),
                            p=p_attn, seed=sd[0]))
            prog.append(gemm(o, W(a.out), Bs(a.out), aa, rows, d, d, d, a.out.s_co, d))
            st1 = v[s["st1"]]
            prog.append(_op(ADLN_FWD, 0, (rows, d), (xin, aa, l.norm1.weight, l.norm1.bias),
                            (v[s["m1"]] if s["m1"] is not None else None, v[s["s1"]], x1, st1[0], st1[1]), p=p1, eps=float(l.norm1.eps), seed=sd[1]))
            prog.append(gemm(x1, W(lyr.ff1), Bs(lyr.ff1), h, rows, lyr.ff1.Co, d, d, lyr.ff1.s_co, lyr.ff1.Co, relu=True,
                             drop_mask=v[s["hmask"]], p=pf, seed=sd[2]))
            prog.append(gemm(h, W(lyr.ff2), Bs(lyr.ff2), f, rows, d, lyr.ff2.Ci, lyr.ff2.Ci, lyr.ff2.s_co, d))
            st2 = v[s["st2"]]
            prog.append(_op(ADLN_FWD, 0, (rows, d), (x1, f, l.norm2.weight, l.norm2.bias),
                            (v[s["m2"]] if s["m2"] is not None else None, v[s["s2"]], y, st2[0], st2[1]), p=p2, eps=float(l.norm2.eps), seed=sd[3]))
            ends.append(len(prog))
            xin = y
        run(prog, dev, xcd, ends)
        ctx.layers, ctx.slots, ctx.views, ctx.buf, ctx.x, ctx.seeds = layers, slots, v, cv.buf, x, seeds
        ctx.dims, ctx.xcd = (B, L, d, rows), xcd
        return v[slots[-1]["y"]].view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        layers, slots, v, x, seeds = ctx.layers, ctx.slots, ctx.views, ctx.x, ctx.seeds
        B, L, d, rows = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        cv = _Carver(dev)
        gs = []
        for lyr in layers:
            F = lyr.ff1.Co
            gs.append(dict(ds2=cv.want(rows, d), df=cv.want(rows, d), dh=cv.want(rows, F), dx1=cv.want(rows, d), ds1=cv.want(rows, d),
                           da=cv.want(rows, d), do=cv.want(rows, d), dqkv=cv.want(rows, 3 * d), dx=cv.want(rows, d)))
        g = cv.alloc()
        prog, ends = [], []
        dcur = dy
        for li in range(len(layers) - 1, -1, -1):
            lyr, s, t, sd = layers[li], slots[li], gs[li], seeds[li]
            a, l = lyr.attn, lyr.l
            dh_ = d // a.h
            p_attn = s["p"][0]
            xin = x if li == 0 else v[slots[li - 1]["y"]]
            st1, st2 = v[s["st1"]], v[s["st2"]]
            G = lambda k: g[t[k]]
            ag = ops.acc_grad
            prog.append(_op(ADLN_BWD, 0, (rows, d), (dcur, None, v[s["s2"]], l.norm2.weight, st2[0], st2[1], v[s["m2"]] if s["m2"] is not None else None),
                            (G("ds2"), G("df"), ag(l.norm2.weight), ag(l.norm2.bias))))
            f1, f2 = lyr.ff1, lyr.ff2
            prog.append(gemm(G("df"), f2.wb, None, G("dh"), rows, f2.Ci, f2.Cop, f2.Cop, f2.Cop, f2.Cip, mul_mask=v[s["hmask"]]))
            prog.append(gemm(G("dh"), f1.wb, None, G("dx1"), rows, f1.Ci, f1.Cop, f1.Cop, f1.Cop, f1.Cip, addend=G("ds2")))
            prog.append(_op(ADLN_BWD, 0, (rows, d), (G("dx1"), None, v[s["s1"]], l.norm1.weight, st1[0], st1[1], v[s["m1"]] if s["m1"] is not None else None),
                            (G("ds1"), G("da"), ag(l.norm1.weight), ag(l.norm1.bias))))
            prog.append(gemm(G("da"), a.out.wb, None, G("do"), rows, a.out.Ci, a.out.Cop, a.out.Cop, a.out.Cop, a.out.Cip))
            qkv, dqkv = v[s["qkv"]], G("dqkv")
            prog.append(_op(ATTN_BWD, 0, (B, a.h, L, L, dh_, 3 * d, 3 * d, d), (G("do"), qkv, (qkv, d), (qkv, 2 * d), v[s["probs"]]),
                            (dqkv, (dqkv, d), (dqkv, 2 * d)), p=p_attn, seed=sd[0]))
            prog.append(gemm(dqkv, a.qkv.wb, None, G("dx"), rows, a.qkv.Ci, a.qkv.Cop, a.qkv.Cop, a.qkv.Cop, a.qkv.Cip, addend=G("ds1")))
            ends.append(len(prog))
            # weight gradients: the bank's one batched launch after the backward pass (LinearFn / FFNFn do the same)
            f2.bank.defer_linear_wgrad(f2, G("df"), v[s["h"]])
            f1.bank.defer_linear_wgrad(f1, G("dh"), v[s["x1"]])
            a.out.bank.defer_linear_wgrad(a.out, G("da"), v[s["o"]])
            a.qkv.bank.defer_linear_wgrad(a.qkv, dqkv, xin.reshape(rows, d))
            dcur = G("dx")
        run(prog, dev, ctx.xcd, ends)
        return dcur.view(B, L, d), None, None, None, None


def encoder_stack(x, layers, training, xcd=0):
    return EncoderStackFn.apply(x, layers[0].ff1.weight, layers, bool(training), int(xcd))


# ------------------------------------------------------------------------------------------------------------------
def decoder_stack_ok(x, memory, layers):
    if x.dtype != torch.float32 or memory.dtype != torch.float32 or x.dim() != 3 or memory.dim() != 3 or not layers:
        return False
    B, L, d = x.shape
    Lm = memory.shape[1]
    if d != 256 or memory.shape[0] != B or memory.shape[2] != d or B * L > 64 or B * Lm > 64 or L > 8 or Lm > 8:
        return False
    for lyr in layers:
        if d // lyr.sa.h > 64 or d // lyr.ca.h > 64:
            return False
        if not all(_pw_ok(pw, B * L) for pw in (lyr.sa.qkv, lyr.sa.out, lyr.ca.q, lyr.ca.out, lyr.ff1, lyr.ff2)) or not _pw_ok(lyr.ca.kv, B * Lm):
            return False
    return True


class DecoderStackFn(torch.autograd.Function):
    """x (B,L,256), memory (B,Lm,256) f32 -> the output of `layers` (layers.DecoderLayer objects:
    nn.TransformerDecoderLayer(norm_first=True), causal self-attention; new_decoder.py:49-51,111-119)."""

    @staticmethod
    def forward(ctx, x, memory, anchor, layers, training, xcd):
        x, memory = x.contiguous(), memory.contiguous()
        B, L, d = x.shape
        Lm = memory.shape[1]
        rows, mrows = B * L, B * Lm
        dev = x.device
        cv = _Carver(dev)
        slots = []
        for lyr in layers:
            l = lyr.l
            F = lyr.ff1.Co
            ps = (float(l.self_attn.dropout), float(l.dropout1.p), float(l.multihead_attn.dropout), float(l.dropout2.p), float(l.dropout.p),
                  float(l.dropout3.p)) if training else (0.0,) * 6
            s = dict(h1=cv.want(rows, d), st0=cv.want(2, rows), qkv=cv.want(rows, 3 * d), probs=cv.want(B, lyr.sa.h, L, L), o=cv.want(rows, d),
                     a=cv.want(rows, d), x2=cv.want(rows, d), h2=cv.want(rows, d), st1=cv.want(2, rows), q=cv.want(rows, d),
                     kvm=cv.want(mrows, 2 * d), probs2=cv.want(B, lyr.ca.h, L, Lm), o2=cv.want(rows, d), a2=cv.want(rows, d),
                     x3=cv.want(rows, d), h3=cv.want(rows, d), st2=cv.want(2, rows), f1=cv.want(rows, F), hmask=cv.want(rows, F),
                     f2=cv.want(rows, d), x4=cv.want(rows, d),
                     m1=cv.want(rows, d) if ps[1] > 0 else None, m2=cv.want(rows, d) if ps[3] > 0 else None,
                     m3=cv.want(rows, d) if ps[5] > 0 else None, p=ps)
            slots.append(s)
        v = cv.alloc()
        W = lambda pw: pw.weight.data_ptr() + 4 * pw.w_off
        Bs = lambda pw: pw.bias.data_ptr() + 4 * pw.b_off
        prog, ends, seeds = [], [], []
        xin = x
        for lyr, s in zip(layers, slots):
            l, sa, ca = lyr.l, lyr.sa, lyr.ca
            ps = s["p"]
            sd = [_next_seed() if pp > 0 else 0 for pp in ps]
            seeds.append(sd)
            V = lambda k: v[s[k]] if s[k] is not None else None
            st0, st1, st2 = V("st0"), V("st1"), V("st2")
            prog.append(_op(ADLN_FWD, 0, (rows, d), (None, xin, l.norm1.weight, l.norm1.bias), (None, None, V("h1"), st0[0], st0[1]),
                            eps=float(l.norm1.eps)))
            # the K | V projection of the memory does not depend on the self-attention block: no barrier before it
            prog.append(gemm(V("h1"), W(sa.qkv), Bs(sa.qkv), V("qkv"), rows, 3 * d, d, d, sa.qkv.s_co, 3 * d, flags=NO_BARRIER))
            prog.append(gemm(memory, W(ca.kv), Bs(ca.kv), V("kvm"), mrows, 2 * d, d, d, ca.kv.s_co, 2 * d))
            qkv = V("qkv")
            prog.append(_op(ATTN_FWD, CAUSAL, (B, sa.h, L, L, d // sa.h, 3 * d, 3 * d, d), (qkv, (qkv, d), (qkv, 2 * d)), (V("o"), V("probs")),
                            p=ps[0], seed=sd[0]))
            prog.append(gemm(V("o"), W(sa.out), Bs(sa.out), V("a"), rows, d, d, d, sa.out.s_co, d))
            prog.append(_op(ADLN_FWD, 0, (rows, d), (xin, V("a"), l.norm2.weight, l.norm2.bias), (V("m1"), V("x2"), V("h2"), st1[0], st1[1]),
                            p=ps[1], eps=float(l.norm2.eps), seed=sd[1]))
            prog.append(gemm(V("h2"), W(ca.q), Bs(ca.q), V("q"), rows, d, d, d, ca.q.s_co, d))
            kvm = V("kvm")
            prog.append(_op(ATTN_FWD, 0, (B, ca.h, L, Lm, d // ca.h, d, 2 * d, d), (V("q"), kvm, (kvm, d)), (V("o2"), V("probs2")),
                            p=ps[2], seed=sd[2]))
            prog.append(gemm(V("o2"), W(ca.out), Bs(ca.out), V("a2"), rows, d, d, d, ca.out.s_co, d))
            prog.append(_op(ADLN_FWD, 0, (rows, d), (V("x2"), V("a2"), l.norm3.weight, l.norm3.bias), (V("m2"), V("x3"), V("h3"), st2[0], st2[1]),
                            p=ps[3], eps=float(l.norm3.eps), seed=sd[3]))
            prog.append(gemm(V("h3"), W(lyr.ff1), Bs(lyr.ff1), V("f1"), rows, lyr.ff1.Co, d, d, lyr.ff1.s_co, lyr.ff1.Co, relu=True,
                             drop_mask=V("hmask"), p=ps[4], seed=sd[4]))
            prog.append(gemm(V("f1"), W(lyr.ff2), Bs(lyr.ff2), V("f2"), rows, d, lyr.ff2.Ci, lyr.ff2.Ci, lyr.ff2.s_co, d))
            prog.append(_op(ADLN_FWD, 0, (rows, d), (V("x3"), V("f2"), None, None), (V("m3"), V("x4"), None, None, None), p=ps[5], seed=sd[5]))
            ends.append(len(prog))
            xin = V("x4")
        run(prog, dev, xcd, ends)
        ctx.layers, ctx.slots, ctx.views, ctx.buf, ctx.x, ctx.memory, ctx.seeds = layers, slots, v, cv.buf, x, memory, seeds
        ctx.dims, ctx.xcd = (B, L, Lm, d, rows, mrows), xcd
        return v[slots[-1]["x4"]].view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        layers, slots, v, x, memory, seeds = ctx.layers, ctx.slots, ctx.views, ctx.x, ctx.memory, ctx.seeds
        B, L, Lm, d, rows, mrows = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        cv = _Carver(dev)
        gs = []
        for lyr in layers:
            F = lyr.ff1.Co
            gs.append(dict(df2=cv.want(rows, d), dF1=cv.want(rows, F), dh3=cv.want(rows, d), dx3=cv.want(rows, d), da2=cv.want(rows, d),
                           do2=cv.want(rows, d), dq=cv.want(rows, d), dkvm=cv.want(mrows, 2 * d), dh2=cv.want(rows, d), dx2=cv.want(rows, d),
                           da=cv.want(rows, d), do=cv.want(rows, d), dqkv=cv.want(rows, 3 * d), dh1=cv.want(rows, d), dx=cv.want(rows, d)))
        dmem_i = cv.want(mrows, d)
        g = cv.alloc()
        dmem = g[dmem_i]
        prog, ends = [], []
        dcur = dy
        ag = ops.acc_grad
        first = True
        for li in range(len(layers) - 1, -1, -1):
            lyr, s, t, sd = layers[li], slots[li], gs[li], seeds[li]
            l, sa, ca = lyr.l, lyr.sa, lyr.ca
            ps = s["p"]
            xin = x if li == 0 else v[slots[li - 1]["x4"]]
            V = lambda k: v[s[k]] if s[k] is not None else None
            G = lambda k: g[t[k]]
            st0, st1, st2 = V("st0"), V("st1"), V("st2")
            f1, f2 = lyr.ff1, lyr.ff2
            if V("m3") is not None:
                prog.append(_op(ADLN_BWD, 0, (rows, d), (None, dcur, None, None, None, None, V("m3")), (None, G("df2"), None, None)))
                df2 = G("df2")
            else:
                df2 = dcur
            prog.append(gemm(df2, f2.wb, None, G("dF1"), rows, f2.Ci, f2.Cop, f2.Cop, f2.Cop, f2.Cip, mul_mask=V("hmask")))
            prog.append(gemm(G("dF1"), f1.wb, None, G("dh3"), rows, f1.Ci, f1.Cop, f1.Cop, f1.Cop, f1.Cip))
            prog.append(_op(ADLN_BWD, 0, (rows, d), (G("dh3"), dcur, V("x3"), l.norm3.weight, st2[0], st2[1], V("m2")),
                            (G("dx3"), G("da2"), ag(l.norm3.weight), ag(l.norm3.bias))))
            prog.append(gemm(G("da2"), ca.out.wb, None, G("do2"), rows, ca.out.Ci, ca.out.Cop, ca.out.Cop, ca.out.Cop, ca.out.Cip))
            kvm, dkvm = V("kvm"), G("dkvm")
            prog.append(_op(ATTN_BWD, 0, (B, ca.h, L, Lm, d // ca.h, d, 2 * d, d), (G("do2"), V("q"), kvm, (kvm, d), V("probs2")),
                            (G("dq"), dkvm, (dkvm, d)), p=ps[2], seed=sd[2]))
            # d(memory) of this layer (summed over the layers through `addend`, in place) beside d(h2): independent
            prog.append(gemm(dkvm, ca.kv.wb, None, dmem, mrows, ca.kv.Ci, ca.kv.Cop, ca.kv.Cop, ca.kv.Cop, ca.kv.Cip,
                             addend=None if first else dmem, flags=NO_BARRIER))
            first = False
            prog.append(gemm(G("dq"), ca.q.wb, None, G("dh2"), rows, ca.q.Ci, ca.q.Cop, ca.q.Cop, ca.q.Cop, ca.q.Cip))
            prog.append(_op(ADLN_BWD, 0, (rows, d), (G("dh2"), G("dx3"), V("x2"), l.norm2.weight, st1[0], st1[1], V("m1")),
                            (G("dx2"), G("da"), ag(l.norm2.weight), ag(l.norm2.bias))))
            prog.append(gemm(G("da"), sa.out.wb, None, G("do"), rows, sa.out.Ci, sa.out.Cop, sa.out.Cop, sa.out.Cop, sa.out.Cip))
            qkv, dqkv = V("qkv"), G("dqkv")
            prog.append(_op(ATTN_BWD, 0, (B, sa.h, L, L, d // sa.h, 3 * d, 3 * d, d), (G("do"), qkv, (qkv, d), (qkv, 2 * d), V("probs")),
                            (dqkv, (dqkv, d), (dqkv, 2 * d)), p=ps[0], seed=sd[0]))
            prog.append(gemm(dqkv, sa.qkv.wb, None, G("dh1"), rows, sa.qkv.Ci, sa.qkv.Cop, sa.qkv.Cop, sa.qkv.Cop, sa.qkv.Cip))
            prog.append(_op(ADLN_BWD, 0, (rows, d), (G("dh1"), G("dx2"), xin, l.norm1.weight, st0[0], st0[1], None),
                            (G("dx"), None, ag(l.norm1.weight), ag(l.norm1.bias))))
            ends.append(len(prog))
            f2.bank.defer_linear_wgrad(f2, df2, V("f1"))
            f1.bank.defer_linear_wgrad(f1, G("dF1"), V("h3"))
            ca.out.bank.defer_linear_wgrad(ca.out, G("da2"), V("o2"))
            ca.kv.bank.defer_linear_wgrad(ca.kv, dkvm, memory.reshape(mrows, d))
            ca.q.bank.defer_linear_wgrad(ca.q, G("dq"), V("h2"))
            sa.out.bank.defer_linear_wgrad(sa.out, G("da"), V("o"))
            sa.qkv.bank.defer_linear_wgrad(sa.qkv, dqkv, V("h1"))
            dcur = G("dx")
        run(prog, dev, ctx.xcd, ends)
        return dcur.view(B, L, d), dmem.view(B, Lm, d), None, None, None, None


def decoder_stack(x, memory, layers, training, xcd=2):
    return DecoderStackFn.apply(x, memory, layers[0].ff1.weight, layers, bool(training), int(xcd))
