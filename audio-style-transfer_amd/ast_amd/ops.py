"""autograd.Function wrappers over the libast_hip C-ABI.

Conventions: image activations are NHWC torch tensors (N,H,W,Cp) in the
compute dtype (config.compute_dtype), Cp = channels padded to a multiple of 8;
token tensors (rows, d) are always f32.  Parameter gradients are accumulated by
the kernels straight into ``param.grad`` (allocated on demand) and the Functions
return None for them -- there is no second pass over the 31 M parameters.
"""
from __future__ import annotations

import functools
import math

import torch

from . import _lib, config
from ._lib import Gather, check, dcode, lib, ptr, stream


@functools.lru_cache(maxsize=None)
def const_tensor(values, dtype, device):
    """Small device constant, created once (outside any hipGraph capture)."""
    return torch.tensor(values, dtype=dtype, device=device)


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def _tap(dh, dw, wt):
    return (dh + 64) | ((dw + 64) << 8) | (wt << 16)


def _mk(N, Hs, Ws, Cs, Hm, Wm, sh, sw, oh, ow, Hd, Wd, Cd, dsh, dsw, doh, dow, taps, wtaps):
    g = Gather(N=N, Hs=Hs, Ws=Ws, Cs=Cs, Hm=Hm, Wm=Wm, sh=sh, sw=sw, oh=oh, ow=ow, Hd=Hd, Wd=Wd, Cd=Cd,
               dsh=dsh, dsw=dsw, doh=doh, dow=dow, ntaps=len(taps), wtaps=wtaps)
    for i, (dh, dw, wt) in enumerate(taps):
        g.tap[i] = _tap(dh, dw, wt)
    return g


@functools.lru_cache(maxsize=None)
def gather_direct(N, H, W, Cs, Cd, k, stride, pad):
    """Conv2d forward geometry (also ConvTranspose2d backward-data): dst pixel
    (ho,wo) reads src (ho*s - p + kh, wo*s - p + kw)."""
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    taps = tuple((kh, kw, kh * k + kw) for kh in range(k) for kw in range(k))
    return _mk(N, H, W, Cs, Ho, Wo, stride, stride, -pad, -pad, Ho, Wo, Cd, 1, 1, 0, 0, taps, k * k), (Ho, Wo)


@functools.lru_cache(maxsize=None)
def gathers_transposed(N, Hs, Ws, Cs, Hd, Wd, Cd, k, stride, pad):
    """Conv2d backward-data / ConvTranspose2d forward: dst pixel hd receives
    src pixel hs through tap kh iff hd = hs*s - p + kh.  One launch per residue
    class (hd mod s, wd mod s) so no MFMA work is spent on structural zeros."""
    out = []
    for ph in range(stride):
        th = [((ph + pad - kh) // stride, kh) for kh in range(k) if (ph + pad - kh) % stride == 0]
        Hm = (Hd - ph + stride - 1) // stride
        for pw in range(stride):
            tw = [((pw + pad - kw) // stride, kw) for kw in range(k) if (pw + pad - kw) % stride == 0]
            Wm = (Wd - pw + stride - 1) // stride
            if Hm <= 0 or Wm <= 0:
                continue
            taps = tuple((dh, dw, kh * k + kw) for dh, kh in th for dw, kw in tw)
            out.append(_mk(N, Hs, Ws, Cs, Hm, Wm, 1, 1, 0, 0, Hd, Wd, Cd, stride, stride, ph, pw, taps, k * k))
    return tuple(out)


PROFILE = None     # bench.py sets this to a list: (kernel config, algorithmic flops, bytes, start event, end event)


def _gemm_cost(g, esize, wgrad=False):
    """Algorithmic work of one launch: 2*M*N*K flops; operands + result once (HBM)."""
    M = g.N * g.Hm * g.Wm
    K = g.ntaps * g.Cs
    flops = 2.0 * M * g.Cd * K
    if wgrad:
        nbytes = (M * g.Cd + g.N * g.Hs * g.Ws * g.Cs) * esize + g.Cd * K * 4
    else:
        src_pix = min(g.N * g.Hs * g.Ws, M * max(g.ntaps, 1))
        nbytes = (src_pix * g.Cs + M * g.Cd + g.Cd * K) * esize
    return flops, nbytes


def _igemm_config(g, dt):
    """Tile plan the library picks for this geometry (reporting only)."""
    import ctypes
    out = (ctypes.c_int32 * 5)()
    check(lib().ast_igemm_plan(g, dt, ctypes.byref(out)), "ast_igemm_plan")
    if out[2] < 0:                                      # patch kernel: (tile rows) x 128 pixels x channels, slab bytes per pixel
        return f"patch{-out[0]}r{'(rows,' + str(-out[3]) + 'frag)' if out[3] < 0 else ''},{out[1]}ch,slab{-out[2]}B"
    return f"{out[0]}x{out[1]},k{out[2] * 16}B" + (f",split{out[3]}" if out[3] > 1 else "") + (f",kg{out[4]}" if out[4] > 1 else "")


_ws_cache = {}


def _ws_need(g, dt):
    key = (id(g), dt)          # Gather objects are interned by the lru-cached geometry builders
    v = _ws_cache.get(key)
    if v is None:
        v = int(lib().ast_igemm_ws_floats(g, dt))
        if v < 0:
            check(-1, "ast_igemm_ws_floats")
        _ws_cache[key] = v
    return v


STAT_SLOTS = 64


def stat_slots(C):
    """Rows of the statistics slot table a GEMM epilogue spreads its atomics over: 64, fewer for wide layers so that
    C x slots <= 4096 (the consumer reduces the whole table in every workgroup: ast_bn_apply_fwd / _bwd)."""
    s = STAT_SLOTS
    while s > 8 and s * C > 4096:
        s //= 2
    return s


def _slot_flags(slots):
    return 0 if slots == 64 else ((slots.bit_length() - 1) << 8)


class _StatArena:
    """Zero-initialised f32 statistics tables ([slots][C][2] BatchNorm sums, [N][C][2] InstanceNorm sums, [..][C][3]
    backward sums) for the GEMM epilogues and reduction passes to ADD into.  With the finalize folded into the apply passes
    nobody can hand a table back clean (its readers are thousands of workgroups), so every table is a fresh region of one
    arena, used once, and the arena is cleared by ONE memset at the start of a train step (stat_arena_reset: Trainer) -- a
    region is never reused within a step, whatever stream its producer and consumer run on."""
    CAP = 1 << 24                    # floats
    by_device = {}


def stat_table(n, device):
    device = torch.device(device)
    a = _StatArena.by_device.get(device)
    if a is None:
        a = _StatArena.by_device[device] = [torch.zeros(_StatArena.CAP, dtype=torch.float32, device=device), 0, 0]
    n64 = (int(n) + 63) // 64 * 64
    if a[1] + n64 > _StatArena.CAP:
        # callers that never reset (modules used directly, outside a Trainer): clear everything once the arena is used up
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("statistics arena exhausted inside a hipGraph capture: call ops.stat_arena_reset() at the start of the step")
        torch.cuda.synchronize(device)
        a[0].zero_()
        a[1] = 0
    t = a[0][a[1]:a[1] + int(n)]
    a[1] += n64
    a[2] = max(a[2], a[1])
    return t


def stat_arena_reset(device):
    """Clear every table handed out so far (one memset on the current stream) and start over.  Call where every stream that
    used a table has been joined into the current one: the start of a train step.  With the separate finalize launches
    (config.fused_finalize off) every table comes back clean from its finalize kernel, and only the allocation rewinds."""
    a = _StatArena.by_device.get(torch.device(device))
    if a is None or a[2] == 0:
        return
    if config.fused_finalize or _SyncBN.active:
        a[0][:a[2]].zero_()
    a[1] = 0


def stats_fusable(g, dt):
    """True when ast_igemm can add the BatchNorm statistics of its output in the epilogue (plans that do not split K)."""
    return _ws_need(g, dt) == 0


class BNLink:
    """Hand-off between a BatchNorm2d(+ReLU) layer and the convolution that consumes its output: that convolution's
    data-gradient GEMM produces the layer's dy and can add the layer's backward sums in its epilogue (ast_igemm_bn)."""
    __slots__ = ("x", "scale", "shift", "relu", "table", "filled", "slots")

    def __init__(self):
        self.x = self.scale = self.shift = self.table = None
        self.relu, self.filled, self.slots = True, False, STAT_SLOTS


def in_stats_fusable(g, dt):
    """True when ast_igemm can add per-IMAGE statistics of its output (flags bits 3 + 6): gathered or patch kernel, no split-K."""
    import ctypes
    out = (ctypes.c_int32 * 5)()
    check(lib().ast_igemm_plan(g, dt, ctypes.byref(out)), "ast_igemm_plan")
    return out[2] != 0 and _ws_need(g, dt) == 0


def _igemm(src, wgt, bias, dst, g, flags=0, stats=None, bn=None, per_image=False):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    need = _ws_need(g, dcode(src.dtype))
    if bn is not None:                          # [slots][Cd][3]: BatchNorm-backward sums of the dy this GEMM produces
        assert need == 0 and flags == 0
        check(lib().ast_igemm_bn(ptr(src), ptr(wgt), ptr(bias), ptr(dst), g, dcode(src.dtype), 16 | (0 if bn.relu else 32) | _slot_flags(bn.slots), ptr(bn.table),
                                 bn.table.numel(), ptr(bn.x), ptr(bn.scale), ptr(bn.shift), stream()), "ast_igemm_bn")
    else:
        if stats is not None:                   # [slots][Cd][2] table filled by the epilogue (flags bit 3); never with split-K
            assert need == 0 and flags == 0
            ws, need, flags = stats, stats.numel(), (8 | 64 if per_image else 8 | _slot_flags(stats.numel() // (2 * g.Cd)))
        else:
            ws = _clean_scratch(need, src.device) if need > 0 else None    # persistent, handed back zeroed by the finish pass
            flags |= 4 if need > 0 else 0
        check(lib().ast_igemm(ptr(src), ptr(wgt), ptr(bias), ptr(dst), g, dcode(src.dtype), flags, ptr(ws), need,
                              stream()), "ast_igemm")
    if PROFILE is not None:
        e1.record()
        fl, by = _gemm_cost(g, src.element_size())
        dt = "bf16" if src.dtype == torch.bfloat16 else "f32"
        PROFILE.append((f"igemm_kernel<{dt},{_igemm_config(g, dcode(src.dtype))}>", fl, by, e0, e1))


class _WgradStream:
    """Weight gradients are leaves of the backward graph: with a stream set here (Trainer, AST_WGRAD_STREAM=1) the convolution
    weight-gradient launches go to that stream -- it waits for the node's stream, nothing waits for it until the bank's flush --
    so the data-gradient chain does not carry them."""
    stream = None
    dirty = False          # launches since the last join (one join per backward pass: redundant graph edges are not free)


def _wgrad(dy, src, dwp, g, replicas=1, pw=None):
    if pw is not None and config.wgrad_defer and PROFILE is None:
        pw.bank.defer_conv_wgrad(dy, src, dwp, g, replicas, pw)       # launched by the bank's end-of-backward flush, on its stream
        return
    W = _WgradStream.stream
    if W is not None and PROFILE is None:
        _WgradStream.dirty = True
        W.wait_stream(torch.cuda.current_stream())
        dy.record_stream(W)
        src.record_stream(W)
        with torch.cuda.stream(W):
            _wgrad_launch(dy, src, dwp, g, replicas, pw)
        return
    _wgrad_launch(dy, src, dwp, g, replicas, pw)


def _wgrad_launch(dy, src, dwp, g, replicas=1, pw=None):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if pw is not None and pw.slab:                 # dwp holds `replicas` copies: one per pixel slice, plain stores; the bank sums them
        import ctypes
        slices = ctypes.c_int32(0)
        check(lib().ast_wgrad_slab(ptr(dy), ptr(src), ptr(dwp), g, dcode(src.dtype), int(replicas), ctypes.byref(slices), stream()), "ast_wgrad_slab")
        pw.bank.note_slab(pw, int(slices.value))
    elif replicas > 1:                             # dwp holds `replicas` zeroed copies; the bank's flush sums them
        check(lib().ast_wgrad_rep(ptr(dy), ptr(src), ptr(dwp), g, dcode(src.dtype), int(replicas), stream()), "ast_wgrad_rep")
    else:
        check(lib().ast_wgrad(ptr(dy), ptr(src), ptr(dwp), g, dcode(src.dtype), stream()), "ast_wgrad")
    if PROFILE is not None:
        e1.record()
        fl, by = _gemm_cost(g, src.element_size(), wgrad=True)
        halo = g.ntaps >= 2 and g.Cd <= 32
        PROFILE.append((f"wgrad{'_halo' if halo else ''}_kernel<{'bf16' if src.dtype == torch.bfloat16 else 'f32'},Cd{min(64, pad8(g.Cd) if g.Cd <= 16 else (32 if g.Cd <= 32 else 64))}>",
                        fl, by, e0, e1))


def acc_grad(p: torch.Tensor) -> torch.Tensor:
    """param.grad to accumulate into (zeros on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class PackedWeight:
    """One GEMM weight of a model: the f32 master parameter (PyTorch layout),
    optional spectral-norm buffers and its two packed images (see WeightBank)."""
    __slots__ = ("weight", "u", "v", "bias", "Co", "Ci", "KK", "s_co", "s_ci", "Cop", "Cip", "dtype", "wf", "wb",
                 "sigma", "scratch", "bias_pad", "w_off", "b_off", "transposed", "gtmp", "dwp", "bank", "replicas", "slab")

    def __init__(self):
        self.wf = self.wb = self.dwp = self.bank = None
        self.replicas = 1
        self.slab = False

    def bias_ptr_tensor(self):
        if self.bias is None:
            return None
        if self.bias_pad is None:
            return self.bias if self.b_off == 0 and self.Co == self.bias.numel() else self.bias[self.b_off:self.b_off + self.Co]
        self.bias_pad[:self.Co].copy_(self.bias.detach()[self.b_off:self.b_off + self.Co])
        return self.bias_pad

    def add_weight_grad(self, dwp, from_wb):
        g = acc_grad(self.weight)
        sn = self.u is not None
        check(lib().ast_weight_grad_unpack(
            ptr(dwp), int(from_wb), self.weight.data_ptr() + 4 * self.w_off, ptr(self.u), ptr(self.v), ptr(self.sigma),
            g.data_ptr() + 4 * self.w_off, self.Co, self.Ci, self.KK, self.s_co, self.s_ci, self.Cop, self.Cip,
            ptr(self.gtmp), stream()), "ast_weight_grad_unpack")

    def add_bias_grad(self, dy2d):
        """dy2d: (rows, Cop) contiguous."""
        if self.bias is None:
            return
        g = acc_grad(self.bias)
        check(lib().ast_colsum_acc(ptr(dy2d), dy2d.numel() // self.Cop, self.Cop, self.Co, g.data_ptr() + 4 * self.b_off,
                                   dcode(dy2d.dtype), stream()), "ast_colsum_acc")


# ---------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------
class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d on NHWC (style_encoder.py:50-67, new_decoder.py:29-61)."""

    @staticmethod
    def forward(ctx, x, weight, pw: PackedWeight, k, stride, pad, bias_grad, stats=None, link=None):
        N, H, W, Cs = x.shape
        g, (Ho, Wo) = gather_direct(N, H, W, Cs, pw.Cop, k, stride, pad)
        y = torch.empty((N, Ho, Wo, pw.Cop), dtype=x.dtype, device=x.device)
        _igemm(x, pw.wf, pw.bias_ptr_tensor(), y, g, stats=stats)
        ctx.save_for_backward(x)
        ctx.pw, ctx.geom, ctx.args, ctx.bias_grad, ctx.link = pw, g, (k, stride, pad), bias_grad, link
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pw, (k, stride, pad) = ctx.pw, ctx.args
        dy = dy.contiguous()
        N, H, W, Cs = x.shape
        if pw.dwp is None:
            raise RuntimeError("weights were prepared in eval/no-grad mode; run the forward in training mode before backward")
        _wgrad(dy, x, pw.dwp, ctx.geom, pw.replicas, pw)           # into the per-model staging arena; unpacked once after backward
        pw.bank.request_flush()
        if ctx.bias_grad:
            pw.add_bias_grad(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gs = gathers_transposed(N, dy.shape[1], dy.shape[2], pw.Cop, H, W, Cs, k, stride, pad)
            link = _bn_link_ready(ctx.link, gs, x)
            for g in gs:
                _igemm(dy, pw.wb, None, dx, g, bn=link)
            if link is not None:
                link.filled = True
        return dx, None, None, None, None, None, None, None, None


def _bn_link_ready(link, geoms, x):
    """The BNLink of a conv input, armed with a zeroed [64][C][3] slot table, when every data-gradient launch can take the
    fused sums (no split-K plan); else None and the BatchNorm backward runs its own pass."""
    if link is None or link.x is None or not config.fused_bn_bwd or _SyncBN.active:
        return None
    if not all(stats_fusable(g, dcode(x.dtype)) for g in geoms):
        return None
    link.slots = stat_slots(x.shape[3])
    link.table = stat_table(link.slots * x.shape[3] * 3, x.device)
    return link



class ResHeadFn(torch.autograd.Function):
    """The two convolutions that read a ResBlock's input -- conv1 (3x3) and the 1x1 shortcut, both with the block's
    stride (style_encoder.py:50, 64-70) -- as ONE autograd node: the second data gradient is accumulated into the
    first one's result by the GEMM epilogue (flags bit 0) instead of an ATen add over the activation map."""

    @staticmethod
    def forward(ctx, x, w1, wd, pw1: PackedWeight, pwd: PackedWeight, stride, stats, in_stats=None):
        N, H, W, Cs = x.shape
        g1, (Ho, Wo) = gather_direct(N, H, W, Cs, pw1.Cop, 3, stride, 1)
        gd, (Hd, Wd) = gather_direct(N, H, W, Cs, pwd.Cop, 1, stride, 0)
        assert (Ho, Wo) == (Hd, Wd)
        c1 = torch.empty((N, Ho, Wo, pw1.Cop), dtype=x.dtype, device=x.device)
        idn = torch.empty((N, Ho, Wo, pwd.Cop), dtype=x.dtype, device=x.device)
        _igemm(x, pw1.wf, pw1.bias_ptr_tensor(), c1, g1, stats=stats)
        _igemm(x, pwd.wf, pwd.bias_ptr_tensor(), idn, gd, stats=in_stats, per_image=in_stats is not None)   # [N][C][2] InstanceNorm sums
        ctx.save_for_backward(x)
        ctx.pws, ctx.geoms, ctx.stride = (pw1, pwd), (g1, gd), stride
        return c1, idn

    @staticmethod
    def backward(ctx, dc1, didn):
        (x,) = ctx.saved_tensors
        (pw1, pwd), (g1, gd), stride = ctx.pws, ctx.geoms, ctx.stride
        dc1, didn = dc1.contiguous(), didn.contiguous()
        N, H, W, Cs = x.shape
        if pw1.dwp is None or pwd.dwp is None:
            raise RuntimeError("weights were prepared in eval/no-grad mode; run the forward in training mode before backward")
        _wgrad(dc1, x, pw1.dwp, g1, pw1.replicas, pw1)
        _wgrad(didn, x, pwd.dwp, gd, pwd.replicas, pwd)
        pw1.bank.request_flush()                 # both convs feed a normalisation: their biases carry no gradient
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            Ho, Wo = dc1.shape[1], dc1.shape[2]
            for g in gathers_transposed(N, Ho, Wo, pw1.Cop, H, W, Cs, 3, stride, 1):
                _igemm(dc1, pw1.wb, None, dx, g)
            for g in gathers_transposed(N, Ho, Wo, pwd.Cop, H, W, Cs, 1, stride, 0):
                if g.ntaps > 0:                  # a 1x1 stride-s conv reaches one output-parity class only
                    _igemm(didn, pwd.wb, None, dx, g, flags=1)
        return dx, None, None, None, None, None, None, None


class ConvT2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d on NHWC (new_decoder.py:72-96)."""

    @staticmethod
    def forward(ctx, x, weight, pw: PackedWeight, k, stride, pad, out_pad, bias_grad, stats=None, link=None):
        N, H, W, Cs = x.shape
        Ho = (H - 1) * stride - 2 * pad + k + out_pad
        Wo = (W - 1) * stride - 2 * pad + k + out_pad
        y = torch.empty((N, Ho, Wo, pw.Cop), dtype=x.dtype, device=x.device)
        b = pw.bias_ptr_tensor()
        for g in gathers_transposed(N, H, W, Cs, Ho, Wo, pw.Cop, k, stride, pad):
            _igemm(x, pw.wf, b, y, g, stats=stats)        # the parity classes partition the output: their statistics add up
        ctx.save_for_backward(x)
        ctx.pw, ctx.args, ctx.bias_grad, ctx.link = pw, (k, stride, pad, Ho, Wo), bias_grad, link
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pw, (k, stride, pad, Ho, Wo) = ctx.pw, ctx.args
        dy = dy.contiguous()
        N, H, W, Cs = x.shape
        # x pixel (h,w) meets dy pixel (h*s - p + kh, ...): the direct geometry with dy as source
        g, (Hx, Wx) = gather_direct(N, Ho, Wo, pw.Cop, Cs, k, stride, pad)
        assert (Hx, Wx) == (H, W), "ConvTranspose2d geometry mismatch"
        if pw.dwp is None:
            raise RuntimeError("weights were prepared in eval/no-grad mode; run the forward in training mode before backward")
        _wgrad(x, dy, pw.dwp, g, pw.replicas, pw)
        pw.bank.request_flush()
        if ctx.bias_grad:
            pw.add_bias_grad(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            link = _bn_link_ready(ctx.link, (g,), x)
            _igemm(dy, pw.wb, None, dx, g, bn=link)
            if link is not None:
                link.filled = True
        return dx, None, None, None, None, None, None, None, None, None


SKINNY_MAX_ROWS = 64


class LinearFn(torch.autograd.Function):
    """nn.Linear on (rows, in) f32 token tensors; optional fused ReLU.  rows <= 64 (every linear of
    the model at B=8: B*(S+1) <= 40 token rows) takes the wave-per-column path of csrc/skinny.hip and
    writes dW/db straight into the parameter gradients; larger row counts use the MFMA igemm."""

    @staticmethod
    def forward(ctx, x, weight, pw: PackedWeight, relu):
        x = x.contiguous()
        rows = x.shape[0]
        assert x.shape[1] == pw.Cip and x.dtype == pw.dtype, (x.shape, pw.Cip, x.dtype, pw.dtype)
        ctx.skinny = rows <= SKINNY_MAX_ROWS and pw.KK == 1 and pw.u is None and pw.Ci == pw.Cip
        y = torch.empty((rows, pw.Cop), dtype=x.dtype, device=x.device)
        if ctx.skinny:
            if pw.Cop != pw.Co:
                y.zero_()
            b = None if pw.bias is None else pw.bias.data_ptr() + 4 * pw.b_off
            check(lib().ast_skinny_gemm(ptr(x), pw.weight.data_ptr() + 4 * pw.w_off, b, ptr(y), rows, pw.Co, pw.Ci, pw.s_co,
                                        pw.Cop, int(relu), stream()), "ast_skinny_gemm")
        else:
            g, _ = gather_direct(rows, 1, 1, pw.Cip, pw.Cop, 1, 1, 0)
            _igemm(x, pw.wf, pw.bias_ptr_tensor(), y, g, 2 if relu else 0)
        ctx.save_for_backward(x, y if relu else None)
        ctx.pw, ctx.relu = pw, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        pw = ctx.pw
        dy = dy.contiguous()
        rows = x.shape[0]
        if ctx.relu:
            dz = torch.empty_like(dy)
            check(lib().ast_relu_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), dcode(dy.dtype), stream()), "ast_relu_bwd")
            dy = dz
        dx = None
        if ctx.skinny:
            if pw.bank is not None:
                pw.bank.defer_linear_wgrad(pw, dy, x)      # all dW/db of the model in ONE launch after backward
            else:
                gw = acc_grad(pw.weight)
                gb = None if pw.bias is None else acc_grad(pw.bias).data_ptr() + 4 * pw.b_off
                check(lib().ast_linear_wgrad(ptr(dy), ptr(x), gw.data_ptr() + 4 * pw.w_off, gb, rows, pw.Co, pw.Ci, pw.Cop, pw.s_co,
                                             stream()), "ast_linear_wgrad")
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)       # dx[m][k] = sum_n dy[m][n] Wt[k][n], Wt = packed [Ci][Cop]
                check(lib().ast_skinny_gemm(ptr(dy), ptr(pw.wb), None, ptr(dx), rows, pw.Ci, pw.Cop, pw.Cop, pw.Cip, 0, stream()),
                      "ast_skinny_gemm")
            return dx, None, None, None
        g, _ = gather_direct(rows, 1, 1, pw.Cip, pw.Cop, 1, 1, 0)
        dwp = torch.zeros((pw.Cop, 1, pw.Cip), dtype=torch.float32, device=x.device)
        _wgrad(dy, x, dwp, g)
        pw.add_weight_grad(dwp, 0)
        pw.add_bias_grad(dy)
        if ctx.needs_input_grad[0]:
            gb, _ = gather_direct(rows, 1, 1, pw.Cop, pw.Cip, 1, 1, 0)
            dx = torch.empty_like(x)
            _igemm(dy, pw.wb, None, dx, gb)
        return dx, None, None, None


class BigLinearFn(torch.autograd.Function):
    """nn.Linear with one HUGE dimension on <= 64 token rows, straight on the f32 parameter (no packed copies: the
    weight is 301 MB) -- SimpleDecoder_TransformerOnly.py:16-17.  in_features huge: ast_bigk_gemm (the input is data,
    no input gradient exists on the reference's path); out_features huge: ast_skinny_gemm forward, ast_bign_dgrad
    backward.  Weight/bias gradients: ast_linear_wgrad into the parameter gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        rows, K = x.shape
        N = weight.shape[0]
        assert weight.shape[1] == K and x.dtype == torch.float32 and weight.is_contiguous()
        if rows > SKINNY_MAX_ROWS:
            raise RuntimeError(f"BigLinearFn: {rows} token rows > {SKINNY_MAX_ROWS}")
        y = torch.empty((rows, N), dtype=torch.float32, device=x.device)
        ctx.big_in = K > N
        if ctx.big_in:
            check(lib().ast_bigk_gemm(ptr(x), ptr(weight), ptr(bias), ptr(y), rows, N, K, N, stream()), "ast_bigk_gemm")
        else:
            check(lib().ast_skinny_gemm(ptr(x), ptr(weight), ptr(bias), ptr(y), rows, N, K, K, N, 0, stream()), "ast_skinny_gemm")
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b = ctx.weight, ctx.bias
        dy = dy.contiguous()
        rows, K = x.shape
        N = w.shape[0]
        check(lib().ast_linear_wgrad(ptr(dy), ptr(x), ptr(acc_grad(w)), ptr(acc_grad(b)) if b is not None else None, rows, N, K, N, K,
                                     stream()), "ast_linear_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.big_in:
                raise NotImplementedError("BigLinearFn: input gradient of the huge-input linear (its input is data on the reference's path)")
            dx = torch.empty_like(x)
            check(lib().ast_bign_dgrad(ptr(dy), ptr(w), ptr(dx), rows, N, K, N, stream()), "ast_bign_dgrad")
        return dx, None, None


class FFNFn(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x)))) on <= 64 token rows in two launches forward and two backward
    (transformer.py _ff_block as style_encoder.py:181-187 / new_decoder.py:111-118 use it): the dropout mask is
    drawn in linear1's epilogue and stored combined with the ReLU mask; backward applies it in the epilogue of
    dh = dy W2.  Weight gradients go to the bank's batched launch."""

    @staticmethod
    def forward(ctx, x, w1, w2, pw1: PackedWeight, pw2: PackedWeight, p):
        x = x.contiguous()
        rows = x.shape[0]
        h = torch.empty((rows, pw1.Cop), dtype=x.dtype, device=x.device)
        y = torch.empty((rows, pw2.Cop), dtype=x.dtype, device=x.device)
        mask, seed, ctr = None, 0, None
        if p > 0.0:
            if _DropState.counter is None or _DropState.counter.device != x.device:
                _DropState.counter = torch.zeros(1, dtype=torch.int64, device=x.device)
            _DropState.calls += 1
            seed, ctr = _DropState.seed + 7919 * _DropState.calls, _DropState.counter
            mask = torch.empty_like(h)
        check(lib().ast_skinny_gemm_ex(ptr(x), pw1.weight.data_ptr() + 4 * pw1.w_off, pw1.bias.data_ptr() + 4 * pw1.b_off, ptr(h), rows,
                                       pw1.Co, pw1.Ci, pw1.s_co, pw1.Cop, 1, None, ptr(mask), float(p), seed, ptr(ctr), stream()),
              "ast_skinny_gemm_ex")
        check(lib().ast_skinny_gemm(ptr(h), pw2.weight.data_ptr() + 4 * pw2.w_off, pw2.bias.data_ptr() + 4 * pw2.b_off, ptr(y), rows,
                                    pw2.Co, pw2.Ci, pw2.s_co, pw2.Cop, 0, stream()), "ast_skinny_gemm")
        ctx.save_for_backward(x, h, mask)
        ctx.pw1, ctx.pw2 = pw1, pw2
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, mask = ctx.saved_tensors
        pw1, pw2 = ctx.pw1, ctx.pw2
        dy = dy.contiguous()
        rows = x.shape[0]
        pw2.bank.defer_linear_wgrad(pw2, dy, h)
        dh = torch.empty_like(h)
        if mask is None:                       # eval-mode autograd: ReLU mask only
            check(lib().ast_skinny_gemm(ptr(dy), ptr(pw2.wb), None, ptr(dh), rows, pw2.Ci, pw2.Cop, pw2.Cop, pw2.Cip, 0, stream()),
                  "ast_skinny_gemm")
            dz = torch.empty_like(dh)
            check(lib().ast_relu_bwd(ptr(dh), ptr(h), ptr(dz), dh.numel(), dcode(dh.dtype), stream()), "ast_relu_bwd")
            dh = dz
        else:
            check(lib().ast_skinny_gemm_ex(ptr(dy), ptr(pw2.wb), None, ptr(dh), rows, pw2.Ci, pw2.Cop, pw2.Cop, pw2.Cip, 0, ptr(mask), None,
                                           0.0, 0, None, stream()), "ast_skinny_gemm_ex")
        pw1.bank.defer_linear_wgrad(pw1, dh, x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib().ast_skinny_gemm(ptr(dh), ptr(pw1.wb), None, ptr(dx), rows, pw1.Ci, pw1.Cop, pw1.Cop, pw1.Cip, 0, stream()),
                  "ast_skinny_gemm")
        return dx, None, None, None, None, None


def _skinny_ok(pw, rows):
    return (rows <= SKINNY_MAX_ROWS and pw.KK == 1 and pw.u is None and pw.Ci == pw.Cip and pw.Co == pw.Cop and pw.bank is not None
            and pw.bias is not None)


def ffn(x2d, pw1, pw2, p, training):
    """linear2(dropout(relu(linear1(x2d))))."""
    rows = x2d.shape[0]
    if _skinny_ok(pw1, rows) and _skinny_ok(pw2, rows) and x2d.dtype == torch.float32:
        return FFNFn.apply(x2d, pw1.weight, pw2.weight, pw1, pw2, float(p) if training else 0.0)
    h = dropout(LinearFn.apply(x2d, pw1.weight, pw1, True), p, training)
    h = h if pw1.Cop == pw1.Co else h[:, :pw1.Co]
    y = LinearFn.apply(h, pw2.weight, pw2, False)
    return y if pw2.Cop == pw2.Co else y[:, :pw2.Co]


# ---------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------
_scratch = {}


def _clean_scratch(n, device, tag=None):
    """Persistent zero-initialised f32 scratch of n floats.  Every kernel pair that uses it (statistics ->
    finalize, split-K GEMM -> finish) hands it back zeroed, so no launch ever needs a memset.  Work on one
    stream is serial, so one buffer per (size, stream) is enough."""
    key = (n, device, torch.cuda.current_stream(device).cuda_stream, tag)    # streams run concurrently: one scratch each
    t = _scratch.get(key)
    if t is None:
        t = _scratch[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return t


def _stats(x):
    N, H, W, C = x.shape
    sums = stat_table(N * C * 2, x.device)
    check(lib().ast_chan_stats(ptr(x), ptr(sums), N, H * W, C, dcode(x.dtype), 1, stream()), "ast_chan_stats")
    return sums


class _SyncBN:
    """Cross-rank BatchNorm statistics: with world > 1 the per-image sums of every BatchNorm2d are
    all-reduced before the finalize, forward and backward, so the normalisation is over all B*S images of the global
    batch as in the single-process reference.  The exchange is a clone + all-reduce + finalize on the current stream: with the
    NCCL (RCCL) backend it is captured into the step's hipGraph like the gradient all-reduce (Trainer, loss-matched mode: one
    stream, so that every collective forks from and joins into the capture's ORIGIN stream -- ast_amd/streams.py); with gloo it
    runs eagerly.  The default data-parallel mode keeps per-rank statistics."""
    world = 1
    group = None
    active = False      # world > 1, or a one-rank group whose collectives are forced on (the single-GPU rehearsal of this path)


def set_sync_bn(world=1, group=None, force=False):
    _SyncBN.world, _SyncBN.group = int(world), group
    _SyncBN.active = _SyncBN.world > 1 or bool(force)


def _global_sums(sums, n_floats):
    """all-reduced copy of the first n_floats of a statistics scratch; the scratch itself is handed back zeroed."""
    import torch.distributed as dist
    g = sums[:n_floats].clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=_SyncBN.group)
    return g


def _bn_batch_stats(x, gamma, beta, bn, stats=None):
    """(mean, rstd, scale, shift) of a training-mode BatchNorm2d over the local or (sync-BN) global batch.
    stats: the [64][C][2] slot table the producing conv's epilogue filled (then no statistics pass runs)."""
    N, H, W, C = x.shape
    rows = stats.numel() // (2 * C) if stats is not None else N
    sums = stats if stats is not None else _stats(x)
    pixels = N * H * W
    if _SyncBN.active:
        g = _global_sums(sums, rows * C * 2)
        sums[:rows * C * 2].zero_()
        out = torch.empty((4, C), dtype=torch.float32, device=x.device)
        check(lib().ast_norm_finalize(ptr(g), 0, ptr(bn.num_batches_tracked), rows, 0, C, gamma.numel(), 0, ptr(gamma),
                                      ptr(beta), ptr(bn.running_mean), ptr(bn.running_var), 0, bn.eps, ptr(out[0]), ptr(out[1]),
                                      ptr(out[2]), ptr(out[3]), pixels * _SyncBN.world, stream()), "ast_norm_finalize")
        return out[0], out[1], out[2], out[3]
    return _finalize(sums, rows, H * W, C, gamma.numel(), False, gamma, beta, bn.running_mean, bn.running_var, False, bn.eps, x.device,
                     nbt=bn.num_batches_tracked, count=pixels)


def _finalize(sums, N, HW, C, Creal, instance, gamma, beta, rm, rv, eval_mode, eps, dev, nbt=None, count=0):
    n = N * C if instance else C
    out = torch.empty((4, n), dtype=torch.float32, device=dev)
    mean, rstd, scale, shift = out[0], out[1], out[2], out[3]
    check(lib().ast_norm_finalize(ptr(sums), 1, ptr(nbt), N, HW, C, Creal, int(instance), ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                                  int(eval_mode), eps, ptr(mean), ptr(rstd), ptr(scale), ptr(shift), count, stream()),
          "ast_norm_finalize")
    return mean, rstd, scale, shift


class BatchNormActFn(torch.autograd.Function):
    """relu?(BatchNorm2d(x)) with batch statistics (training) or running stats (eval)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, training, relu, stats=None, link=None):
        N, H, W, C = x.shape
        Creal = gamma.numel()
        ctx.link = None
        ctx.fused = bool(training and config.fused_finalize and not _SyncBN.active)
        if ctx.fused:
            # statistics finalize + apply in ONE launch (ast_bn_apply_fwd): the table is reduced by every workgroup
            tab = stats if stats is not None else _stats(x)
            rows = tab.numel() // (2 * C)
            out = torch.empty((4, C), dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            check(lib().ast_bn_apply_fwd(ptr(x), None, ptr(y), ptr(tab), rows, N * H * W, None, ptr(gamma), ptr(beta), ptr(bn.running_mean),
                                         ptr(bn.running_var), ptr(bn.num_batches_tracked), bn.eps, None, None, 0.0, ptr(out), None, N, H * W, C,
                                         Creal, int(relu), dcode(x.dtype), stream()), "ast_bn_apply_fwd")
            mean, rstd, scale, shift = out[0], out[1], out[2], out[3]
            if link is not None:
                link.x, link.scale, link.shift, link.relu = x, scale, shift, bool(relu)
                ctx.link = link
            ctx.save_for_backward(x, None, mean, rstd, scale, shift)
            ctx.gamma, ctx.beta, ctx.relu, ctx.training = gamma, beta, relu, training
            return y
        if training:
            mean, rstd, scale, shift = _bn_batch_stats(x, gamma, beta, bn, stats)
            if link is not None:                # the consumer conv's data gradient may add this layer's backward sums
                link.x, link.scale, link.shift, link.relu = x, scale, shift, bool(relu)
                ctx.link = link
        else:
            mean, rstd, scale, shift = _finalize(None, N, H * W, C, Creal, False, gamma, beta, bn.running_mean,
                                                 bn.running_var, True, bn.eps, x.device)
        y = torch.empty_like(x)
        check(lib().ast_affine_act(ptr(x), ptr(scale), ptr(shift), None, None, None, ptr(y), N, H * W, C, int(relu),
                                   dcode(x.dtype), stream()), "ast_affine_act")
        ctx.save_for_backward(x, y, mean, rstd, scale, shift)
        ctx.gamma, ctx.beta, ctx.relu, ctx.training = gamma, beta, relu, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, scale, shift = ctx.saved_tensors
        pre = config.bn_mask_from_preact and ctx.relu          # ReLU mask recomputed from x: y is not read
        if not ctx.training:
            raise RuntimeError("BatchNormActFn: backward in eval mode is not on the reference's path")
        dy = dy.contiguous()
        N, H, W, C = x.shape
        gamma, beta = ctx.gamma, ctx.beta
        link = ctx.link
        if ctx.fused:
            if link is not None and link.filled:        # the data-gradient GEMM that produced dy added the sums to the slot table
                link.filled = False
                tab3, rows = link.table, link.slots
            else:
                tab3, rows = stat_table(N * C * 3, x.device), N
                check(lib().ast_norm_bwd_sums_pre(ptr(dy), None, ptr(x), None, ptr(tab3), N, H * W, C, int(ctx.relu), dcode(x.dtype), 1,
                                                  ptr(scale), ptr(shift), None, None, stream()), "ast_norm_bwd_sums")
            dx = torch.empty_like(x)
            check(lib().ast_bn_apply_bwd(ptr(dy), ptr(x), None, ptr(dx), None, ptr(tab3), rows, N * H * W, ptr(gamma), ptr(mean), ptr(rstd),
                                         ptr(acc_grad(gamma)), ptr(acc_grad(beta)), None, None, None, None, None, ptr(scale), ptr(shift), None, None,
                                         N, H * W, C, gamma.numel(), int(ctx.relu), dcode(x.dtype), stream()), "ast_bn_apply_bwd")
            return dx, None, None, None, None, None, None, None
        k1 = torch.empty((C, 3), dtype=torch.float32, device=x.device)
        if link is not None and link.filled:
            # the data-gradient GEMM that produced dy already added (sum dz, sum dz*x) to the slot table: no pass over dy, x
            link.filled = False
            check(lib().ast_norm_bwd_finalize_n(ptr(link.table), 1, link.slots, 0, C, gamma.numel(), ptr(gamma), ptr(mean), ptr(rstd),
                                                ptr(acc_grad(gamma)), ptr(acc_grad(beta)), ptr(k1), None, None, None, None, None, None,
                                                N * H * W, stream()), "ast_norm_bwd_finalize")
            gsum = None
        else:
            sums3 = _clean_scratch(N * C * 3, x.device)
            check(lib().ast_norm_bwd_sums_pre(ptr(dy), ptr(y), ptr(x), None, ptr(sums3), N, H * W, C, int(ctx.relu),
                                              dcode(x.dtype), 1, ptr(scale) if pre else None, ptr(shift) if pre else None, None, None,
                                              stream()), "ast_norm_bwd_sums")
            gsum = _global_sums(sums3, N * C * 3) if _SyncBN.active else None
            # gamma/beta gradients come from the LOCAL sums (the gradient all-reduce averages them over ranks)
            check(lib().ast_norm_bwd_finalize(ptr(sums3), 1, N, H * W, C, gamma.numel(), ptr(gamma), ptr(mean), ptr(rstd),
                                              ptr(acc_grad(gamma)), ptr(acc_grad(beta)), ptr(k1),
                                              None, None, None, None, None, None, stream()), "ast_norm_bwd_finalize")
        if gsum is not None:        # dx coefficients from the global sums and the global pixel count
            check(lib().ast_norm_bwd_finalize(ptr(gsum), 0, N, H * W * _SyncBN.world, C, gamma.numel(), ptr(gamma), ptr(mean), ptr(rstd),
                                              None, None, ptr(k1), None, None, None, None, None, None, stream()), "ast_norm_bwd_finalize")
        dx = torch.empty_like(x)
        check(lib().ast_norm_bwd_apply_pre(ptr(dy), ptr(y), ptr(x), None, ptr(k1), None, ptr(dx), None, N, H * W, C,
                                           int(ctx.relu), dcode(x.dtype), ptr(scale) if pre else None, ptr(shift) if pre else None,
                                           None, None, stream()), "ast_norm_bwd_apply")
        return dx, None, None, None, None, None, None, None


class ResTailFn(torch.autograd.Function):
    """relu(BatchNorm2d(c2) + InstanceNorm2d(ds)) -- the ResBlock tail
    (style_encoder.py:76-83) as one elementwise pass over both branches."""

    @staticmethod
    def forward(ctx, c2, ds, g1, b1, g2, b2, bn, inn, training, stats=None, in_stats=None):
        N, H, W, C = c2.shape
        Creal = g1.numel()
        ctx.fused = bool(training and config.fused_finalize and not _SyncBN.active)
        if ctx.fused:
            tab1 = stats if stats is not None else _stats(c2)
            tab2 = in_stats if in_stats is not None else _stats(ds)
            rows1 = tab1.numel() // (2 * C)
            out1 = torch.empty((4, C), dtype=torch.float32, device=c2.device)
            out2 = torch.empty((4, N * C), dtype=torch.float32, device=c2.device)
            y = torch.empty_like(c2)
            check(lib().ast_bn_apply_fwd(ptr(c2), ptr(ds), ptr(y), ptr(tab1), rows1, N * H * W, ptr(tab2), ptr(g1), ptr(b1), ptr(bn.running_mean),
                                         ptr(bn.running_var), ptr(bn.num_batches_tracked), bn.eps, ptr(g2), ptr(b2), inn.eps, ptr(out1), ptr(out2),
                                         N, H * W, C, Creal, 1, dcode(c2.dtype), stream()), "ast_bn_apply_fwd")
            ctx.save_for_backward(c2, ds, None, out1[0], out1[1], out2[0], out2[1], out1[2], out1[3], out2[2], out2[3])
            ctx.params, ctx.training = (g1, b1, g2, b2), training
            return y
        if training:
            m1, r1, s1, f1 = _bn_batch_stats(c2, g1, b1, bn, stats)
        else:
            m1, r1, s1, f1 = _finalize(None, N, H * W, C, Creal, False, g1, b1, bn.running_mean, bn.running_var,
                                       True, bn.eps, c2.device)
        # InstanceNorm sums of the shortcut: from the shortcut convolution's epilogue when it could add them, else one pass
        m2, r2, s2, f2 = _finalize(in_stats if in_stats is not None else _stats(ds), N, H * W, C, Creal, True, g2, b2, None, None, False,
                                   inn.eps, c2.device)
        y = torch.empty_like(c2)
        check(lib().ast_affine_act(ptr(c2), ptr(s1), ptr(f1), ptr(ds), ptr(s2), ptr(f2), ptr(y), N, H * W, C, 1,
                                   dcode(c2.dtype), stream()), "ast_affine_act")
        ctx.save_for_backward(c2, ds, y, m1, r1, m2, r2, s1, f1, s2, f2)
        ctx.params, ctx.training = (g1, b1, g2, b2), training
        return y

    @staticmethod
    def backward(ctx, dy):
        c2, ds, y, m1, r1, m2, r2, s1, f1, s2, f2 = ctx.saved_tensors
        pre = config.bn_mask_from_preact
        if not ctx.training:
            raise RuntimeError("ResTailFn: backward in eval mode is not on the reference's path")
        g1, b1, g2, b2 = ctx.params
        dy = dy.contiguous()
        N, H, W, C = c2.shape
        dev = c2.device
        if ctx.fused:
            tab3 = stat_table(N * C * 3, dev)
            check(lib().ast_norm_bwd_sums_pre(ptr(dy), None, ptr(c2), ptr(ds), ptr(tab3), N, H * W, C, 1, dcode(c2.dtype), 1,
                                              ptr(s1), ptr(f1), ptr(s2), ptr(f2), stream()), "ast_norm_bwd_sums")
            dc2, dds = torch.empty_like(c2), torch.empty_like(ds)
            check(lib().ast_bn_apply_bwd(ptr(dy), ptr(c2), ptr(ds), ptr(dc2), ptr(dds), ptr(tab3), N, N * H * W, ptr(g1), ptr(m1), ptr(r1),
                                         ptr(acc_grad(g1)), ptr(acc_grad(b1)), ptr(g2), ptr(m2), ptr(r2), ptr(acc_grad(g2)), ptr(acc_grad(b2)),
                                         ptr(s1), ptr(f1), ptr(s2), ptr(f2), N, H * W, C, g1.numel(), 1, dcode(c2.dtype), stream()), "ast_bn_apply_bwd")
            return dc2, dds, None, None, None, None, None, None, None, None, None
        sums3 = _clean_scratch(N * C * 3, dev)
        coef = (ptr(s1), ptr(f1), ptr(s2), ptr(f2)) if pre else (None, None, None, None)
        check(lib().ast_norm_bwd_sums_pre(ptr(dy), ptr(y), ptr(c2), ptr(ds), ptr(sums3), N, H * W, C, 1, dcode(c2.dtype),
                                          1, *coef, stream()), "ast_norm_bwd_sums")
        k1 = torch.empty((C, 3), dtype=torch.float32, device=dev)
        k2 = torch.empty((N, C, 3), dtype=torch.float32, device=dev)
        gsum = _global_sums(sums3, N * C * 3) if _SyncBN.active else None
        check(lib().ast_norm_bwd_finalize(ptr(sums3), 1, N, H * W, C, g1.numel(), ptr(g1), ptr(m1), ptr(r1),
                                          ptr(acc_grad(g1)), ptr(acc_grad(b1)), ptr(k1), ptr(g2), ptr(m2), ptr(r2),
                                          ptr(acc_grad(g2)), ptr(acc_grad(b2)), ptr(k2), stream()),
              "ast_norm_bwd_finalize")
        if gsum is not None:        # batch branch only: the InstanceNorm shortcut is per image, hence local
            check(lib().ast_norm_bwd_finalize(ptr(gsum), 0, N, H * W * _SyncBN.world, C, g1.numel(), ptr(g1), ptr(m1), ptr(r1),
                                              None, None, ptr(k1), None, None, None, None, None, None, stream()), "ast_norm_bwd_finalize")
        dc2, dds = torch.empty_like(c2), torch.empty_like(ds)
        check(lib().ast_norm_bwd_apply_pre(ptr(dy), ptr(y), ptr(c2), ptr(ds), ptr(k1), ptr(k2), ptr(dc2), ptr(dds), N, H * W,
                                           C, 1, dcode(c2.dtype), *coef, stream()), "ast_norm_bwd_apply")
        return dc2, dds, None, None, None, None, None, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        shape = x.shape
        x2 = x.contiguous().view(-1, shape[-1])
        rows, D = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        check(lib().ast_layernorm_fwd(ptr(x2), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, D, eps,
                                      dcode(x2.dtype), stream()), "ast_layernorm_fwd")
        ctx.save_for_backward(x2, mean, rstd)
        ctx.gamma, ctx.beta, ctx.shape = gamma, beta, shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        rows, D = x2.shape
        dy2 = dy.contiguous().view(rows, D)
        dx = torch.empty_like(x2)
        check(lib().ast_layernorm_bwd(ptr(dy2), ptr(x2), ptr(ctx.gamma), ptr(mean), ptr(rstd), ptr(dx),
                                      ptr(acc_grad(ctx.gamma)), ptr(acc_grad(ctx.beta)), rows, D, dcode(x2.dtype),
                                      stream()), "ast_layernorm_bwd")
        return dx.view(ctx.shape), None, None, None


class AddDropLNFn(torch.autograd.Function):
    """One launch for `norm(x + dropout(sub))` on f32 token rows (ast_add_drop_ln_fwd/bwd).

    Returns (s, y) with s = x + dropout(sub) and y = LayerNorm(s); with `ln` None only s.  The post-norm encoder
    layer (style_encoder.py:181-187) uses y, the pre-norm decoder layer (new_decoder.py:111-118) carries s on as its
    residual stream and feeds y to the next sub-layer.  gamma/beta gradients are accumulated in place."""

    @staticmethod
    def forward(ctx, x, sub, ln, p):
        ctx.set_materialize_grads(False)
        shape = sub.shape
        D = shape[-1]
        sub2 = sub.contiguous().view(-1, D)
        x2 = x.contiguous().view(-1, D) if x is not None else None
        rows = sub2.shape[0]
        s = torch.empty_like(sub2)
        mask = torch.empty_like(sub2) if p > 0.0 else None
        seed = 0
        if p > 0.0:
            if _DropState.counter is None or _DropState.counter.device != sub.device:
                _DropState.counter = torch.zeros(1, dtype=torch.int64, device=sub.device)
            _DropState.calls += 1
            seed = _DropState.seed + 7919 * _DropState.calls
        y = mean = rstd = None
        if ln is not None:
            y = torch.empty_like(sub2)
            mean = torch.empty(rows, dtype=torch.float32, device=sub.device)
            rstd = torch.empty_like(mean)
        check(lib().ast_add_drop_ln_fwd(ptr(x2), ptr(sub2), ptr(mask), ptr(s), ptr(ln.weight) if ln is not None else None,
                                        ptr(ln.bias) if ln is not None else None, ptr(y), ptr(mean), ptr(rstd), rows, D,
                                        float(ln.eps) if ln is not None else 0.0, float(p), seed,
                                        ptr(_DropState.counter) if p > 0.0 else None, stream()), "ast_add_drop_ln_fwd")
        ctx.save_for_backward(s, mean, rstd, mask)
        ctx.ln, ctx.shape, ctx.has_x = ln, shape, x is not None
        if ln is None:
            return s.view(shape)
        return s.view(shape), y.view(shape)

    @staticmethod
    def backward(ctx, ds_ext, dy=None):
        s, mean, rstd, mask = ctx.saved_tensors
        rows, D = s.shape
        if ds_ext is None and dy is None:
            return None, None, None, None
        ds2 = ds_ext.contiguous().view(rows, D) if ds_ext is not None else None
        dy2 = dy.contiguous().view(rows, D) if dy is not None else None
        want_dx = ctx.has_x and ctx.needs_input_grad[0]
        # without a LayerNorm term the residual gradient passes straight through
        dx = None if (not want_dx or dy2 is None) else torch.empty_like(s)
        if dy2 is None and mask is None:
            return (ds_ext if want_dx else None), ds_ext, None, None
        dsub = torch.empty_like(s)
        ln = ctx.ln
        check(lib().ast_add_drop_ln_bwd(ptr(dy2), ptr(ds2), ptr(s), ptr(ln.weight) if dy2 is not None else None, ptr(mean), ptr(rstd),
                                        ptr(mask), ptr(dx), ptr(dsub), ptr(acc_grad(ln.weight)) if dy2 is not None else None,
                                        ptr(acc_grad(ln.bias)) if dy2 is not None else None, rows, D, stream()), "ast_add_drop_ln_bwd")
        if want_dx and dx is None:
            dx = ds2
        return (dx.view(ctx.shape) if want_dx else None), dsub.view(ctx.shape), None, None


def add_drop_ln(x, sub, ln, p, training):
    """(s, y) = (x + dropout(sub), LayerNorm(s)); y is None when ln is None."""
    out = AddDropLNFn.apply(x, sub, ln, float(p) if training else 0.0)
    return (out, None) if ln is None else out


class RowMixFn(torch.autograd.Function):
    """Y = A X for a small constant coefficient matrix A (ast_rowmix); backward is A^T dY."""

    @staticmethod
    def forward(ctx, x, A, At):
        x = x.contiguous()
        R_out, R_in = A.shape
        assert x.shape[0] == R_in and x.dtype == torch.float32
        y = torch.empty((R_out, x.shape[1]), dtype=torch.float32, device=x.device)
        check(lib().ast_rowmix(ptr(A), ptr(x), ptr(y), R_out, R_in, x.shape[1], stream()), "ast_rowmix")
        ctx.At = At
        return y

    @staticmethod
    def backward(ctx, dy):
        At = ctx.At
        dy = dy.contiguous()
        dx = torch.empty((At.shape[0], dy.shape[1]), dtype=torch.float32, device=dy.device)
        check(lib().ast_rowmix(ptr(At), ptr(dy), ptr(dx), At.shape[0], At.shape[1], dy.shape[1], stream()), "ast_rowmix")
        return dx, None, None


@functools.lru_cache(maxsize=256)
def _mix_matrices(kind, key, device):
    """(A, A^T) as device constants.  kind 'proto': rows = present classes ascending, A[c][i] = 1/count_c for label_i == c;
    'gather': A[b][c] = 1 for class index c of row b; 'mean': key = (B, S), A[b][b*S + s] = 1/S."""
    if kind == "proto":
        classes = sorted(set(key))
        A = torch.zeros(len(classes), len(key))
        for r, cid in enumerate(classes):
            idx = [i for i, v in enumerate(key) if v == cid]
            A[r, idx] = 1.0 / len(idx)
    elif kind == "gather":
        classes = sorted(set(key))
        A = torch.zeros(len(key), len(classes))
        for b, v in enumerate(key):
            A[b, classes.index(v)] = 1.0
    else:
        B, S = key
        A = torch.zeros(B, B * S)
        for b in range(B):
            A[b, b * S:(b + 1) * S] = 1.0 / S
    return A.to(device).contiguous(), A.t().contiguous().to(device)


def mul_const(x, c):
    """x * c elementwise for small f32 vectors (no autograd): one ast_mul launch."""
    y = torch.empty_like(x)
    check(lib().ast_mul(ptr(x.contiguous()), ptr(c), ptr(y), x.numel(), 0, stream()), "ast_mul")
    return y


def class_means(emb, labels_host):
    """style_encoder.py:243-253: mean embedding per PRESENT class id, ascending."""
    A, At = _mix_matrices("proto", tuple(int(v) for v in labels_host.tolist()), emb.device)
    return RowMixFn.apply(emb, A, At)


def class_rows(class_emb, labels_host):
    """class_emb[row of label_b] for every batch row b (labels as present-class ranks)."""
    A, At = _mix_matrices("gather", tuple(int(v) for v in labels_host.tolist()), class_emb.device)
    return RowMixFn.apply(class_emb, A, At)


def mean_over_sections(x):
    """(B,S,d).mean(dim=1)."""
    B, S, d = x.shape
    if S == 1:
        return x.reshape(B, d)
    A, At = _mix_matrices("mean", (B, S), x.device)
    return RowMixFn.apply(x.reshape(B * S, d), A, At)


# ---------------------------------------------------------------------------
# pooling / resampling / layout
# ---------------------------------------------------------------------------
class AdaptivePoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        N, H, W, C = x.shape
        y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
        check(lib().ast_adaptive_pool_fwd(ptr(x), ptr(y), N, H, W, C, Ho, Wo, dcode(x.dtype), stream()), "pool_fwd")
        ctx.dims = (N, H, W, C, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, Ho, Wo = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        check(lib().ast_adaptive_pool_bwd(ptr(dy), ptr(dx), N, H, W, C, Ho, Wo, dcode(dy.dtype), stream()), "pool_bwd")
        return dx, None, None


class BilinearToNCHWFn(torch.autograd.Function):
    """nn.Upsample(size, 'bilinear', align_corners=False) (new_decoder.py:99): NHWC(Cp) -> NCHW f32."""

    @staticmethod
    def forward(ctx, x, C, Ho, Wo):
        N, H, W, Cp = x.shape
        y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        check(lib().ast_bilinear_fwd(ptr(x), ptr(y), N, C, Cp, H, W, Ho, Wo, dcode(x.dtype), stream()), "bilinear_fwd")
        ctx.dims, ctx.dtype = (N, C, Cp, H, W, Ho, Wo), x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, Cp, H, W, Ho, Wo = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, Cp), dtype=ctx.dtype, device=dy.device)
        check(lib().ast_bilinear_bwd(ptr(dy), ptr(dx), N, C, Cp, H, W, Ho, Wo, dcode(ctx.dtype), stream()), "bilinear_bwd")
        return dx, None, None, None


def nchw_to_nhwc(x4, dtype, Cp=None):
    """(N,C,H,W) f32 view with unit W stride -> NHWC `dtype`, channels padded (no grad: model inputs)."""
    N, Cc, H, W = x4.shape
    assert x4.dtype == torch.float32 and x4.stride(3) == 1, "expects f32 with contiguous last dim"
    Cp = Cp or pad8(Cc)
    y = torch.empty((N, H, W, Cp), dtype=dtype, device=x4.device)
    check(lib().ast_nchw_to_nhwc(ptr(x4), ptr(y), N, Cc, H, W, x4.stride(0), x4.stride(1), x4.stride(2), Cp, dcode(dtype),
                                 stream()), "ast_nchw_to_nhwc")
    return y


class _SharedInput:
    """The NHWC image of the step's input, shared by the modules that read the same x (both encoders).

    Set by `shared_nhwc(x)` for the duration of ONE forward pass and dropped at its end: the conversion kernel is
    launched inside the step that uses it (so a hipGraph capture records it and the image lives in the graph's pool),
    and nothing survives into the next step -- a later step on new values of the same tensor object converts again."""
    x = None
    y = None
    dtype = None


class shared_nhwc:
    """with shared_nhwc(x5, dtype): ...   -- encoders called inside find the (B*S,T,F,Cp) image of x5 ready."""

    def __init__(self, x5, dtype):
        self.x5, self.dtype = x5, dtype

    def __enter__(self):
        B, S, C, T, F = self.x5.shape
        _SharedInput.x, _SharedInput.dtype = self.x5, self.dtype
        _SharedInput.y = nchw_to_nhwc(self.x5.view(B * S, C, T, F), self.dtype)
        return _SharedInput.y

    def __exit__(self, *exc):
        _SharedInput.x = _SharedInput.y = _SharedInput.dtype = None
        return False


def cached_nhwc(x5, dtype):
    """(B,S,C,T,F) f32 -> (B*S,T,F,Cp) NHWC: the image a surrounding `shared_nhwc(x5)` scope prepared, else a fresh
    conversion (no state is kept between calls)."""
    if _SharedInput.x is x5 and _SharedInput.dtype == dtype:
        return _SharedInput.y
    B, S, C, T, F = x5.shape
    return nchw_to_nhwc(x5.view(B * S, C, T, F), dtype)


class CastFn(torch.autograd.Function):
    """dtype cast between the image dtype and f32 token tensors."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        check(lib().ast_cast(ptr(x), dcode(x.dtype), ptr(y), dcode(dtype), x.numel(), stream()), "ast_cast")
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == ctx.src:
            return dy, None
        dy = dy.contiguous()
        dx = torch.empty(dy.shape, dtype=ctx.src, device=dy.device)
        check(lib().ast_cast(ptr(dy), dcode(dy.dtype), ptr(dx), dcode(ctx.src), dy.numel(), stream()), "ast_cast")
        return dx, None


# ---------------------------------------------------------------------------
# attention core + dropout
# ---------------------------------------------------------------------------
class AttnCoreFn(torch.autograd.Function):
    """softmax(QK^T/sqrt(dh) + causal) V for <=16 tokens.  q:(B*Lq, ldq) k,v views of (B*Lk, ldk).
    p_drop > 0: the attention-probability dropout is drawn inside the kernels (same values forward and backward)."""

    @staticmethod
    def forward(ctx, q, kv, B, H, Lq, Lk, dh, k_off, v_off, causal, p_drop):
        d = H * dh
        o = torch.empty((B * Lq, d), dtype=torch.float32, device=q.device)
        probs = torch.empty((B, H, Lq, Lk), dtype=torch.float32, device=q.device)
        ldq, ldk = q.stride(0), kv.stride(0)
        seed, ctr = 0, None
        if p_drop > 0.0:
            if _DropState.counter is None or _DropState.counter.device != q.device:
                _DropState.counter = torch.zeros(1, dtype=torch.int64, device=q.device)
            _DropState.calls += 1
            seed, ctr = _DropState.seed + 7919 * _DropState.calls, _DropState.counter
        check(lib().ast_attn_fwd_p(q.data_ptr(), kv.data_ptr() + 4 * k_off, kv.data_ptr() + 4 * v_off, ptr(o), ptr(probs),
                                   B, H, Lq, Lk, dh, ldq, ldk, d, int(causal), None, float(p_drop), seed, ptr(ctr), stream()), "ast_attn_fwd")
        ctx.save_for_backward(q, kv, probs)
        ctx.dims = (B, H, Lq, Lk, dh, k_off, v_off)
        ctx.drop = (float(p_drop), seed, ctr)
        ctx.same = q.data_ptr() == kv.data_ptr()
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, probs = ctx.saved_tensors
        B, H, Lq, Lk, dh, k_off, v_off = ctx.dims
        p_drop, seed, ctr = ctx.drop
        do = do.contiguous()
        d = H * dh
        dq = torch.empty_like(q)                    # the kernel writes every q / k / v column of every row
        dkv = dq if ctx.same else torch.empty_like(kv)
        check(lib().ast_attn_bwd_p(ptr(do), q.data_ptr(), kv.data_ptr() + 4 * k_off, kv.data_ptr() + 4 * v_off, ptr(probs),
                                   dq.data_ptr(), dkv.data_ptr() + 4 * k_off, dkv.data_ptr() + 4 * v_off, B, H, Lq, Lk, dh,
                                   q.stride(0), kv.stride(0), d, None, p_drop, seed, ptr(ctr), stream()), "ast_attn_bwd")
        return dq, (None if ctx.same else dkv), None, None, None, None, None, None, None, None, None


class _DropState:
    seed = 0x5EED
    counter = None   # device int64, bumped once per training step by the trainer (hipGraph-replay safe)
    calls = 0


def dropout_mask(shape, p, device):
    if _DropState.counter is None or _DropState.counter.device != device:
        _DropState.counter = torch.zeros(1, dtype=torch.int64, device=device)
    _DropState.calls += 1
    m = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib().ast_dropout_mask(ptr(m), m.numel(), p, _DropState.seed + 7919 * _DropState.calls, ptr(_DropState.counter),
                                 stream()), "ast_dropout_mask")
    return m


class MulMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().ast_mul(ptr(x), ptr(mask), ptr(y), x.numel(), dcode(x.dtype), stream()), "ast_mul")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().ast_mul(ptr(dy), ptr(mask), ptr(dx), dy.numel(), dcode(dy.dtype), stream()), "ast_mul")
        return dx, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on f32 token tensors: mask generation and application in one launch."""

    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        if _DropState.counter is None or _DropState.counter.device != x.device:
            _DropState.counter = torch.zeros(1, dtype=torch.int64, device=x.device)
        _DropState.calls += 1
        y, mask = torch.empty_like(x), torch.empty_like(x)
        check(lib().ast_dropout_fwd(ptr(x), ptr(y), ptr(mask), x.numel(), p, _DropState.seed + 7919 * _DropState.calls,
                                    ptr(_DropState.counter), stream()), "ast_dropout_fwd")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().ast_mul(ptr(dy), ptr(mask), ptr(dx), dy.numel(), dcode(dy.dtype), stream()), "ast_mul")
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    if x.dtype != torch.float32:
        return MulMaskFn.apply(x, dropout_mask(x.shape, p, x.device))
    return DropoutFn.apply(x, float(p))


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
def _scaled(grad_saved, gout):
    y = torch.empty_like(grad_saved)
    check(lib().ast_scale(ptr(grad_saved), ptr(gout.contiguous().float()), 1.0, ptr(y), y.numel(), 0, stream()), "ast_scale")
    return y


class WeightedSumFn(torch.autograd.Function):
    """total = sum_i weights[widx_i] * term_i for scalar loss terms, the weights read from DEVICE memory at run time
    (ast_weighted_sum): one launch forward, one backward, and a captured hipGraph follows a ramped loss weight without being
    captured again.  widx_i < 0: weight 1."""

    @staticmethod
    def forward(ctx, weights, widx, *terms):
        import ctypes as C
        n = len(terms)
        ts = [t.detach().reshape(1) for t in terms]
        assert all(t.dtype == torch.float32 and t.is_cuda for t in ts) and len(widx) == n
        out = torch.empty(1, dtype=torch.float32, device=weights.device)
        parr = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        iarr = (C.c_int32 * n)(*[int(i) for i in widx])
        check(lib().ast_weighted_sum(parr, iarr, n, ptr(weights), ptr(out), stream()), "ast_weighted_sum")
        ctx.weights, ctx.widx, ctx.shapes = weights, tuple(int(i) for i in widx), [t.shape for t in terms]
        return out[0]

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        n = len(ctx.widx)
        grads = torch.empty(n, dtype=torch.float32, device=g.device)
        iarr = (C.c_int32 * n)(*ctx.widx)
        check(lib().ast_weighted_sum_bwd(ptr(g.contiguous().reshape(1)), iarr, n, ptr(ctx.weights), ptr(grads), stream()), "ast_weighted_sum_bwd")
        return (None, None) + tuple(grads[i].view(shp) for i, shp in enumerate(ctx.shapes))


def weighted_sum(weights, pairs):
    """pairs: [(index into `weights` or -1, scalar tensor), ...] -> sum of weights[index] * tensor."""
    return WeightedSumFn.apply(weights, tuple(i for i, _ in pairs), *[t for _, t in pairs])


class ReconTotalFn(torch.autograd.Function):
    """compute_comprehensive_loss (new_decoder.py:348-420) in one pass plus a finishing launch: returns (total, raw sums[5],
    reported means[5] = inv * sums) and keeps d total / d out computed in the same pass."""

    @staticmethod
    def forward(ctx, out, tgt, coefs, inv):
        import ctypes as C
        B, S, Cc, T, Fq = out.shape
        assert Cc == 2 and out.is_contiguous() and out.dtype == torch.float32
        assert tgt.shape == out.shape and tgt.dtype == torch.float32 and tgt.stride(4) == 1
        ld = tgt.stride(3)
        assert tgt.stride(2) == T * ld and tgt.stride(1) == 2 * T * ld and tgt.stride(0) == S * 2 * T * ld, \
            "target must be a [..., :F] slice of a contiguous tensor"
        ws = torch.empty(64 * 5, dtype=torch.float32, device=out.device)          # AST_RECON_SLOTS rows of partial sums
        res = torch.empty(11, dtype=torch.float32, device=out.device)
        grad = torch.empty_like(out)
        c5, i5 = (C.c_float * 5)(*[float(c) for c in coefs]), (C.c_float * 5)(*[float(c) for c in inv])
        check(lib().ast_recon_loss_total(ptr(out), ptr(tgt), ld, B, S, T, Fq, c5, i5, ptr(ws), ptr(res), ptr(grad), stream()),
              "ast_recon_loss_total")
        ctx.save_for_backward(grad)
        total, sums, parts = res[5], res[:5], res[6:]
        ctx.mark_non_differentiable(sums, parts)
        return total, sums, parts

    @staticmethod
    def backward(ctx, gtotal, gsums, gparts):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, gtotal), None, None, None


class InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, labels32, temperature):
        emb = emb.contiguous()
        B, D = emb.shape
        loss = torch.empty(1, dtype=torch.float32, device=emb.device)
        demb = torch.empty_like(emb)
        ws = torch.empty(2 * B * B + B, dtype=torch.float32, device=emb.device)
        check(lib().ast_infonce(ptr(emb), ptr(labels32), B, D, temperature, ptr(loss), ptr(demb), ptr(ws), stream()),
              "ast_infonce")
        ctx.save_for_backward(demb)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (demb,) = ctx.saved_tensors
        return _scaled(demb, g), None, None


class MarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, margin):
        cls = cls.contiguous()
        Cn, D = cls.shape
        loss = torch.empty(1, dtype=torch.float32, device=cls.device)
        dcls = torch.empty_like(cls)
        check(lib().ast_margin(ptr(cls), Cn, D, margin, ptr(loss), ptr(dcls), stream()), "ast_margin")
        ctx.save_for_backward(dcls)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _scaled(d, g), None


class HSICFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, c):
        s, c = s.contiguous(), c.contiguous()
        B, D = s.shape
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        ds, dc = torch.empty_like(s), torch.empty_like(c)
        ws = torch.empty(6 * B * B + 2 * B + 8, dtype=torch.float32, device=s.device)
        check(lib().ast_hsic(ptr(s), ptr(c), B, D, ptr(loss), ptr(ds), ptr(dc), ptr(ws), stream()), "ast_hsic")
        ctx.save_for_backward(ds, dc)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ds, dc = ctx.saved_tensors
        return _scaled(ds, g), _scaled(dc, g)


class CrossCovFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, c):
        s, c = s.contiguous(), c.contiguous()
        B, D = s.shape
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        ds, dc = torch.empty_like(s), torch.empty_like(c)
        ws = torch.empty(2 * D + 2 * B * B, dtype=torch.float32, device=s.device)
        check(lib().ast_crosscov(ptr(s), ptr(c), B, D, ptr(loss), ptr(ds), ptr(dc), ptr(ws), stream()), "ast_crosscov")
        ctx.save_for_backward(ds, dc)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ds, dc = ctx.saved_tensors
        return _scaled(ds, g), _scaled(dc, g)


class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target32):
        logits = logits.contiguous()
        R, Cn = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        check(lib().ast_cross_entropy(ptr(logits), ptr(target32), R, Cn, ptr(loss), ptr(dl), stream()), "ast_cross_entropy")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return _scaled(dl, g), None


class SoftmaxEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        logits = logits.contiguous()
        R, Cn = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        check(lib().ast_softmax_entropy(ptr(logits), R, Cn, ptr(loss), ptr(dl), stream()), "ast_softmax_entropy")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return _scaled(dl, g)
