"""Host-side building blocks shared by the drop-in modules.

torch.nn classes (Conv2d+spectral_norm, BatchNorm2d, Linear, MultiheadAttention,
Transformer*Layer ...) are used ONLY as parameter containers, so construction,
default initialisation, ``state_dict`` keys/shapes and checkpoint loading are
identical to the reference by construction; their ``forward`` is never called --
all arithmetic goes through libast_hip (ops.py).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch
import torch.nn as nn

from . import config, ops, streams
from ._lib import LinWg, WeightDesc, check, dcode, lib, ptr, stream
from .ops import PackedWeight, pad8


class WeightBank:
    """All GEMM weights of one model.  ``prepare(training)`` runs, in three
    launches for the whole model, the spectral-norm power iteration
    (torch spectral_norm.py:92-114), sigma, and the packing of W/sigma into the
    [rows][tap][channel] images the MFMA kernels read."""

    def __init__(self):
        self.specs = []
        self.entries: list[PackedWeight] = []
        self._key = None
        self._keys = {}
        self._flush_pending = False
        self._flush_stream = None
        self._lin_deferred = []       # (PackedWeight, dy, x) of this backward pass, weight gradients computed in one launch
        self._pinned = []             # pointer tables referenced by captured memcpy nodes must outlive the graph
        self._pin_pool = None         # pinned host memory for those tables, allocated outside capture
        self._pin_used = 0
        self._conv_dirty = False
        self._slab_recs = []          # (PackedWeight, pixel slices) of this backward pass's slab-mode weight gradients
        self._conv_deferred = []      # config.wgrad_defer: (dy, src, dwp, geometry, replicas, PackedWeight) launched by _flush
        self._defer_streams = []
        self._pool_flushed = False    # _DeferPool: this bank's flush kernels were already launched (helper bank)
        self._wait_lane = None        # _DeferPool: the helper's stream carries some of this bank's launches
        self.d_train = self.d_eval = None
        self._dev_consts = {}         # content -> device tensor (descriptor tables, tile lists): see _const_dev
        self._packed = {}             # (entry index, dtype, size) -> the entry's packed images and scratch
        self.hold = False     # True: packed images are current (several forwards between optimiser steps)

    def add(self, weight, kind, dtype_fn, u=None, v=None, bias=None, rows=None):
        """kind: 'conv' (Co,Ci,k,k) | 'convT' (Ci,Co,k,k) | 'linear' (Co,Ci); rows=(r0,r1) selects output rows
        of a linear weight (packed q/k/v projections).  dtype_fn() gives the packed dtype at build time."""
        self.specs.append((weight, kind, dtype_fn, u, v, bias, rows))
        self.entries.append(PackedWeight())
        return self.entries[-1]

    def _const_dev(self, payload: bytes, dtype, dev):
        """Device copy of a small constant table, keyed by CONTENT and never freed.  A captured hipGraph bakes the ADDRESSES of
        the bank's descriptor / tile / dtype tables into its kernel nodes: rebuilding the bank (an eval pass between two
        training steps, a compute-dtype switch and back) must hand the same bytes back at the same address, and must not free
        what an older graph still reads -- a replay after such a rebuild read a freed tile list and faulted the GPU."""
        key = (payload, dtype, str(dev))
        t = self._dev_consts.get(key)
        if t is None:
            t = self._dev_consts[key] = torch.frombuffer(bytearray(payload), dtype=dtype).to(dev)
        return t

    def _signature(self, training):
        return (training,) + tuple((w.data_ptr(), w.device, fn(), (None if w.grad is None else w.grad.data_ptr()) if training else 0)
                                   for w, _, fn, *_ in self.specs)

    def _build(self, training):
        dev = self.specs[0][0].device
        descs, dts = [], []
        self.max_co = self.max_cols = self.max_packed = 1
        # packed f32 gradient staging for every conv / conv-transpose weight: ONE arena, zeroed by the
        # pack kernel of each training forward, unpacked by ONE batched launch pair after backward
        # Small weights belong to the pixel-rich layers, whose weight-gradient kernels end with ~85 workgroups adding one tile
        # each into the same addresses (f32 atomics serialise: 14 us of a 40 us launch).  They get WGRAD_REPLICAS copies of
        # their staging; ast_wgrad_rep spreads the workgroups over them and the flush sums the copies.
        # With config.wgrad_slabs the small weights get that many copies instead and every pixel slice of the weight-gradient launch
        # STORES into its own (ast_wgrad_slab); ast_slab_sum adds them into copy 0 in _flush, and the descriptor shows ONE copy.
        sizes, reps, slabs = [], [], []
        for (w, kind, *_rest) in self.specs:
            if kind == "linear":
                sizes.append(0); reps.append(1); slabs.append(False)
            else:
                co, ci = (w.shape[0], w.shape[1]) if kind == "conv" else (w.shape[1], w.shape[0])
                one = pad8(co) * w.shape[2] * w.shape[3] * pad8(ci)
                small = one <= (1 << 18)
                if small and config.wgrad_slabs > 1:
                    r, sl = config.wgrad_slabs, True
                else:
                    r, sl = (config.wgrad_replicas if (config.wgrad_replicas > 1 and small) else 1), False
                sizes.append(one * r); reps.append(r); slabs.append(sl)
        if training and (getattr(self, "dw_arena", None) is None or self.dw_arena.device != dev or self.dw_arena.numel() < max(1, sum(sizes))):
            self.dw_arena = torch.zeros(max(1, sum(sizes)), dtype=torch.float32, device=dev)
        off = 0
        for e, (w, kind, dtype_fn, u, v, bias, rows), sz, rep, slab in zip(self.entries, self.specs, sizes, reps, slabs):
            dt = dtype_fn()
            if kind == "conv":
                Co, Ci, k, _ = w.shape
                KK, s_co, s_ci, w_off, b_off = k * k, Ci * k * k, k * k, 0, 0
            elif kind == "convT":
                Ci, Co, k, _ = w.shape
                KK, s_co, s_ci, w_off, b_off = k * k, k * k, Co * k * k, 0, 0
            else:
                Co, Ci = w.shape
                r0, r1 = rows if rows else (0, Co)
                Co, KK, s_co, s_ci, w_off, b_off = r1 - r0, 1, Ci, 1, r0 * Ci, r0
            n = pad8(Co) * KK * pad8(Ci)
            if e.wf is None or e.wf.device != dev or e.wf.dtype != dt or e.wf.numel() != n:
                # one set of packed images per (entry, dtype), kept for the bank's lifetime: graphs captured in the other
                # compute dtype keep reading theirs
                pk = (len(descs), dt, n, str(dev))
                if pk not in self._packed:
                    self._packed[pk] = (torch.empty(n, dtype=dt, device=dev), torch.empty(n, dtype=dt, device=dev),
                                        torch.ones(1, dtype=torch.float32, device=dev),
                                        torch.zeros(int(lib().ast_sn_scratch_floats(Co, Ci * KK)), dtype=torch.float32, device=dev),
                                        torch.zeros(1, dtype=torch.float32, device=dev),
                                        torch.zeros(pad8(Co), dtype=torch.float32, device=dev) if (bias is not None and Co != pad8(Co)) else None)
                e.wf, e.wb, e.sigma, e.scratch, e.gtmp, e.bias_pad = self._packed[pk]
            e.weight, e.u, e.v, e.bias, e.bank = w, u, v, bias, self
            e.Co, e.Ci, e.KK, e.s_co, e.s_ci, e.w_off, e.b_off = Co, Ci, KK, s_co, s_ci, w_off, b_off
            e.Cop, e.Cip, e.dtype = pad8(Co), pad8(Ci), dt
            # training-only state (gradient staging slice) is left alone by an eval build: a validation pass or an
            # inference session between two training steps must not invalidate the training descriptors
            grad_ptr = None
            if training:
                e.dwp, e.replicas, e.slab = None, rep, slab
                if sz:
                    e.dwp = self.dw_arena[off:off + sz]
                    off += sz
                    grad_ptr = ops.acc_grad(w).data_ptr() + 4 * w_off
            descs.append(WeightDesc(w=w.data_ptr() + 4 * w_off, u=ptr(u), v=ptr(v), sigma=ptr(e.sigma), scratch=ptr(e.scratch),
                                    wf=ptr(e.wf), wb=ptr(e.wb), Co=Co, Ci=Ci, KK=KK, s_co=s_co, s_ci=s_ci, Cop=e.Cop, Cip=e.Cip,
                                    power_iter=1 if training else 0, dwp=ptr(e.dwp), grad=grad_ptr, inner=ptr(e.gtmp),
                                    dwp_from_wb=1 if kind == "convT" else 0, dwp_replicas=(1 if slab else rep) if training else 1))
            dts.append(dcode(dt))
            self.max_co = max(self.max_co, Co)
            self.max_cols = max(self.max_cols, Ci * KK)
            self.max_packed = max(self.max_packed, n)
        # 32x32-channel tiles of every weight for the LDS-tiled pack / flush kernels
        tl = []
        for wi, e in enumerate(self.entries):
            for co0 in range(0, e.Cop, 32):
                for ci0 in range(0, e.Cip, 32):
                    tl += [wi, co0, ci0, 0]
        if self._pin_pool is None:
            self._pin_pool = torch.empty(256 * 1024, dtype=torch.uint8, pin_memory=True)
        import array
        self.ntiles = len(tl) // 4
        self.d_tiles = self._const_dev(array.array("i", tl).tobytes(), torch.int32, dev)
        arr = (WeightDesc * len(descs))(*descs)
        dev_descs = self._const_dev(bytes(arr), torch.uint8, dev)
        if training:
            self.d_train = dev_descs
        else:
            self.d_eval = dev_descs
        self.d_dtypes = self._const_dev(array.array("i", dts).tobytes(), torch.int32, dev)
        self._keys[training] = self._signature(training)

    def prepare(self, training: bool):
        if self.hold:
            return
        training = bool(training) and torch.is_grad_enabled()
        if self._keys.get(training) != self._signature(training):
            self._build(training)
        d = self.d_train if training else self.d_eval
        from . import _lib as _L
        if _L.PROFILE_CALLS is not None:       # algorithmic bytes of the pack: read the f32 masters, write both packed images
            _L.NEXT_BYTES = sum(e.Co * e.Ci * e.KK * 4 + 2 * e.wf.numel() * e.wf.element_size() for e in self.entries)
        check(lib().ast_weights_prepare_t(ptr(d), ptr(self.d_dtypes), len(self.entries), self.max_co, self.max_cols,
                                          ptr(self.d_tiles), self.ntiles, stream()), "ast_weights_prepare_t")

    # ---- batched weight-gradient unpack, once per backward pass ----------------------
    def defer_linear_wgrad(self, pw, dy, x):
        self._lin_deferred.append((pw, dy, x))
        self.request_flush(conv=False)

    def defer_conv_wgrad(self, dy, src, dwp, g, replicas, pw):
        self._conv_deferred.append((dy, src, dwp, g, replicas, pw))
        if not any(b is self for b in _DeferPool.banks):
            _DeferPool.banks.append(self)

    def note_slab(self, pw, slices):
        if any(p is pw for p, _ in self._slab_recs):
            raise NotImplementedError("a convolution weight was used twice in one backward pass: its slab-mode gradient does not "
                                      "accumulate (set AST_WGRAD_SLABS=0 for the atomic replicas)")
        self._slab_recs.append((pw, slices))

    def _sum_slabs(self):
        """copy 0 <- sum of the slices' copies, for all slab-mode weights of this backward pass, 48 weights per launch"""
        recs, self._slab_recs = self._slab_recs, []
        for i in range(0, len(recs), 48):
            part = recs[i:i + 48]
            n = len(part)
            bases = (C.c_void_p * n)(*[pw.dwp.data_ptr() for pw, _ in part])
            sizes = (C.c_int64 * n)(*[pw.dwp.numel() // pw.replicas for pw, _ in part])
            slabs = (C.c_int32 * n)(*[sl for _, sl in part])
            check(lib().ast_slab_sum(bases, sizes, slabs, n, stream()), "ast_slab_sum")

    def _flush_conv_kernels(self, side):
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if self._slab_recs:
                self._sum_slabs()
            check(lib().ast_weight_grads_flush_t(ptr(self.d_train), ptr(self.d_tiles), self.ntiles, stream()),
                  "ast_weight_grads_flush_t")

    def _side_stream(self, origin):
        """This bank's flush stream, forked from `origin` once per parallel_flush() scope."""
        if self._flush_stream is None:
            self._flush_stream = torch.cuda.Stream(device=self.d_tiles.device)
        if not any(s is self._flush_stream for s in _ParallelFlush.pending):
            streams.fork(self._flush_stream, origin)
            _ParallelFlush.pending.append(self._flush_stream)
        return self._flush_stream

    def _launch_deferred(self, origin):
        """The deferred convolution weight gradients of this backward pass (config.wgrad_defer), longest first, spread over
        config.wgrad_defer_streams streams: the current (flush) stream and helpers forked from `origin` -- the stream the flush
        was forked from -- that the flush stream then waits for (sibling waits; a stream never waits on its own child, streams.py)."""
        items, self._conv_deferred = self._conv_deferred, []
        cur = torch.cuda.current_stream()
        k = max(1, min(config.wgrad_defer_streams, len(items)))
        if k > 1 and cur != origin:
            while len(self._defer_streams) < k - 1:
                self._defer_streams.append(torch.cuda.Stream(device=self.d_tiles.device))
            lanes = [cur] + self._defer_streams[:k - 1]
            for st in lanes[1:]:
                streams.fork(st, origin)
        else:
            lanes = [cur]
        cost = lambda it: float(it[3].N) * it[3].Hm * it[3].Wm * max(it[3].Cd, 32) * max(it[3].ntaps * it[3].Cs, 64)
        load = [0.0] * len(lanes)
        for it in sorted(items, key=cost, reverse=True):
            i = load.index(min(load))
            load[i] += cost(it)
            dy, src, dwp, g, replicas, pw = it
            dy.record_stream(lanes[i])
            src.record_stream(lanes[i])
            with torch.cuda.stream(lanes[i]):
                ops._wgrad_launch(dy, src, dwp, g, replicas, pw)
        for st in lanes[1:]:
            cur.wait_stream(st)

    def request_flush(self, conv=True):
        self._conv_dirty = self._conv_dirty or conv
        if not self._flush_pending:
            self._flush_pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self._flush)

    def _flush(self):
        self._flush_pending = False
        if ops._WgradStream.stream is not None and ops._WgradStream.dirty:      # the weight-gradient launches of this backward pass
            ops._WgradStream.dirty = False
            streams.join(torch.cuda.current_stream(), ops._WgradStream.stream)
        if self._conv_dirty:
            self._conv_dirty = False
            from . import _lib as _L
            if _L.PROFILE_CALLS is not None:   # read the packed f32 staging + the master (spectral-norm term), add into the gradient
                _L.NEXT_BYTES = sum((e.dwp.numel() * 4 if e.dwp is not None else 0) + 3 * e.Co * e.Ci * e.KK * 4 for e in self.entries if e.dwp is not None)
            side, origin = None, torch.cuda.current_stream()
            if _ParallelFlush.active:
                if config.wgrad_defer_pool and self._conv_deferred and len(_DeferPool.banks) > 1:
                    _DeferPool.launch_all(origin)
                # the banks' flushes touch disjoint, persistent buffers (staging arena, masters, flat gradient): inside
                # parallel_flush() each bank's pair of kernels runs on its own stream and the context joins them
                side = self._side_stream(origin)
            if self._pool_flushed:                  # _DeferPool already ran this bank's flush on its stream (it is the helper)
                self._pool_flushed = False
            else:
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    if self._conv_deferred:
                        self._launch_deferred(origin)
                    if self._wait_lane is not None:     # launches of this bank that _DeferPool gave to the helper's stream
                        torch.cuda.current_stream().wait_stream(self._wait_lane)
                        self._wait_lane = None
                self._flush_conv_kernels(side)
        if self._lin_deferred:
            items, self._lin_deferred = self._lin_deferred, []
            recs, max_tiles = [], 1
            for pw, dy, x in items:
                gw = ops.acc_grad(pw.weight)
                gb = 0 if pw.bias is None else ops.acc_grad(pw.bias).data_ptr() + 4 * pw.b_off
                recs.append(LinWg(dy=dy.data_ptr(), x=x.data_ptr(), dW=gw.data_ptr() + 4 * pw.w_off, db=gb or None, M=x.shape[0],
                                  N=pw.Co, K=pw.Ci, lddy=pw.Cop, ldw=pw.s_co, p0=0, p1=0, p2=0))
                max_tiles = max(max_tiles, ((pw.Ci + 63) // 64) * ((pw.Co + 63) // 64))
            # The kernel adds into dW / db with plain read-modify-write, one workgroup per (record, tile): two records with the
            # SAME destination (a module applied more than once per backward pass -- the discriminator sees style, content
            # and class embeddings, losses.py:89-104) must not share a launch.  Round r holds the r-th use of every weight;
            # rounds run back to back on the stream, so the sum order is fixed (bit-reproducible).
            rounds, seen = [], {}
            for rec in recs:
                r = seen.get(rec.dW, 0)
                seen[rec.dW] = r + 1
                if r == len(rounds):
                    rounds.append([])
                rounds[r].append(rec)
            # records go to the kernel BY VALUE (kernel arguments): no device table, no H2D copy node in the captured step;
            # the operand tensors are kept alive for graph replays
            if torch.cuda.is_current_stream_capturing():
                self._pinned.append(items)
            for rr in rounds:
                arr = (LinWg * len(rr))(*rr)
                check(lib().ast_linear_wgrad_batched_host(C.addressof(arr), len(rr), max_tiles, stream()), "ast_linear_wgrad_batched_host")


class _ParallelFlush:
    active = False
    pending = []


class _DeferPool:
    """config.wgrad_defer_pool: balance the deferred weight gradients of a backward pass over the banks' flush streams.  The banks
    differ in size (decoder < encoders) and the flush streams share the GPU unevenly, so the last bank used to run alone for a
    quarter of the phase.  The SMALLEST bank's stream is the helper: it runs its own launches and its own flush, then takes over
    the cheapest launches of the larger banks, whose flush streams wait for it -- waits in ONE direction only (the helper never
    waits for a donor: mutual waits between side streams crash the capture, streams.py), no barrier through the origin."""
    banks = []
    cost = staticmethod(lambda it: float(it[3].N) * it[3].Hm * it[3].Wm * max(it[3].Cd, 32) * max(it[3].ntaps * it[3].Cs, 64))

    @staticmethod
    def _run(lane, items):
        for dy, src, dwp, g, replicas, pw in sorted(items, key=_DeferPool.cost, reverse=True):
            dy.record_stream(lane)
            src.record_stream(lane)
            with torch.cuda.stream(lane):
                ops._wgrad_launch(dy, src, dwp, g, replicas, pw)

    @staticmethod
    def launch_all(origin):
        banks, _DeferPool.banks = [b for b in _DeferPool.banks if b._conv_deferred], []
        cost = _DeferPool.cost
        tot = [sum(cost(it) for it in b._conv_deferred) for b in banks]
        h = tot.index(min(tot))
        target = sum(tot) / len(banks)
        room = max(0.0, target - tot[h]) * config.wgrad_defer_lend
        lent = []
        for bi, b in enumerate(banks):
            keep = list(b._conv_deferred)
            if bi != h and room > 0:
                give = min(tot[bi] - target, room * (tot[bi] - target) / max(1e-9, sum(max(0.0, t - target) for i, t in enumerate(tot) if i != h)))
                keep.sort(key=cost)                                   # cheapest first: the donor keeps (and starts with) its long launches
                while keep and give >= cost(keep[0]):
                    give -= cost(keep[0])
                    lent.append(keep.pop(0))
                if len(keep) < len(b._conv_deferred):
                    b._wait_lane = banks[h]._side_stream(origin)
            b._conv_deferred = []
            _DeferPool._run(b._side_stream(origin), keep)
        helper = banks[h]
        if lent:
            helper._flush_conv_kernels(helper._side_stream(origin))   # the helper's own gradients are complete: flush, then help
            helper._pool_flushed = True
            _DeferPool._run(helper._side_stream(origin), lent)


@contextlib.contextmanager
def parallel_flush():
    """Within this context the end-of-backward gradient flushes of different WeightBanks run on per-bank streams; leaving it
    makes the current stream wait for all of them.  (The three generator banks' flushes were 0.24 ms back to back at the
    end of the step's longest chain.)"""
    if os.environ.get("AST_PARALLEL_FLUSH", "1") == "0":
        yield
        return
    _ParallelFlush.active, _ParallelFlush.pending = True, []
    _DeferPool.banks = []
    try:
        yield
    finally:
        _ParallelFlush.active = False
        cur = torch.cuda.current_stream()
        for s in _ParallelFlush.pending:
            streams.join(cur, s)
        _ParallelFlush.pending = []


def img_dtype():
    return config.compute_dtype


def tok_dtype():
    return torch.float32


# ---- thin functional wrappers --------------------------------------------------------
def _link_of(x):
    """BNLink left on a BatchNorm(+ReLU) output by bn_act / conv_bn_act / convT_bn_act (None otherwise)."""
    return getattr(x, "_bn_link", None)


def conv(x, pw, k, stride, pad, bias_grad):
    return ops.Conv2dFn.apply(x, pw.weight, pw, k, stride, pad, bias_grad, None, _link_of(x))


def convT(x, pw, k, stride, pad, out_pad, bias_grad):
    return ops.ConvT2dFn.apply(x, pw.weight, pw, k, stride, pad, out_pad, bias_grad, None, _link_of(x))


def _bn_apply(y, bn, training, relu, stats):
    link = ops.BNLink() if (training and config.fused_bn_bwd) else None
    out = ops.BatchNormActFn.apply(y, bn.weight, bn.bias, bn, training, relu, stats, link)
    if link is not None:
        out._bn_link = link              # read by the conv that consumes `out` (conv / convT / conv_with_stats)
    return out


def linear(x2d, pw, relu=False):
    y = ops.LinearFn.apply(x2d, pw.weight, pw, relu)
    return y if pw.Cop == pw.Co else y[:, :pw.Co]


def bn_act(x, bn: nn.BatchNorm2d, training, relu=True, stats=None):
    return _bn_apply(x, bn, training, relu, stats)


def conv_with_stats(x, pw, k, stride, pad, training):
    """conv(x) plus, in training mode, the [64][C][2] slot table of its output's per-channel (sum, sum of squares)
    filled by the GEMM epilogue (None when the plan splits K or fusion is switched off: the consumer then runs the
    separate statistics pass).  The table must be consumed by the very next normalisation on this stream."""
    stats = None
    if training and config.fused_bn_stats:
        N, H, W, Cs = x.shape
        g, _ = ops.gather_direct(N, H, W, Cs, pw.Cop, k, stride, pad)
        if ops.stats_fusable(g, ops.dcode(x.dtype)):
            stats = ops.stat_table(ops.stat_slots(pw.Cop) * pw.Cop * 2, x.device)
    return ops.Conv2dFn.apply(x, pw.weight, pw, k, stride, pad, False, stats, _link_of(x)), stats


def convT_bn_act(x, pw, k, stride, pad, out_pad, bn: nn.BatchNorm2d, training, relu=True):
    """relu?(BatchNorm2d(conv_transpose(x))), statistics from the epilogues of the output-parity-class GEMMs."""
    stats = None
    if training and config.fused_bn_stats:
        N, H, W, Cs = x.shape
        Ho = (H - 1) * stride - 2 * pad + k + out_pad
        Wo = (W - 1) * stride - 2 * pad + k + out_pad
        gs = ops.gathers_transposed(N, H, W, Cs, Ho, Wo, pw.Cop, k, stride, pad)
        if all(ops.stats_fusable(g, ops.dcode(x.dtype)) for g in gs):
            stats = ops.stat_table(ops.stat_slots(pw.Cop) * pw.Cop * 2, x.device)
    y = ops.ConvT2dFn.apply(x, pw.weight, pw, k, stride, pad, out_pad, False, stats, _link_of(x))
    return _bn_apply(y, bn, training, relu, stats)


def res_head(x, p1, pd, stride, training):
    """(conv1(x), shortcut_conv(x), BatchNorm statistics table of conv1's output or None, per-image InstanceNorm sums of the
    shortcut's output or None) -- see ops.ResHeadFn."""
    stats = None
    if training and config.fused_bn_stats:
        N, H, W, Cs = x.shape
        g, _ = ops.gather_direct(N, H, W, Cs, p1.Cop, 3, stride, 1)
        if ops.stats_fusable(g, ops.dcode(x.dtype)):
            stats = ops.stat_table(ops.stat_slots(p1.Cop) * p1.Cop * 2, x.device)
    in_stats = None
    if training and config.fused_bn_stats:
        N, H, W, Cs = x.shape
        gd, _ = ops.gather_direct(N, H, W, Cs, pd.Cop, 1, stride, 0)
        if ops.in_stats_fusable(gd, ops.dcode(x.dtype)):
            in_stats = ops.stat_table(N * pd.Cop * 2, x.device)
    c1, idn = ops.ResHeadFn.apply(x, p1.weight, pd.weight, p1, pd, stride, stats, in_stats)
    return c1, idn, stats, in_stats


def conv_bn_act(x, pw, k, stride, pad, bn: nn.BatchNorm2d, training, relu=True):
    """relu?(BatchNorm2d(conv(x))) -- the conv bias has no effect through a batch norm and carries no gradient."""
    y, stats = conv_with_stats(x, pw, k, stride, pad, training)
    return _bn_apply(y, bn, training, relu, stats)


def layer_norm(x, ln: nn.LayerNorm):
    return ops.LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)


class MHA:
    """Bank entries + forward for one nn.MultiheadAttention container."""

    def __init__(self, bank: WeightBank, m: nn.MultiheadAttention, cross: bool):
        d = m.embed_dim
        self.m, self.d, self.h, self.cross = m, d, m.num_heads, cross
        if cross:
            self.q = bank.add(m.in_proj_weight, "linear", tok_dtype, bias=m.in_proj_bias, rows=(0, d))
            self.kv = bank.add(m.in_proj_weight, "linear", tok_dtype, bias=m.in_proj_bias, rows=(d, 3 * d))
        else:
            self.qkv = bank.add(m.in_proj_weight, "linear", tok_dtype, bias=m.in_proj_bias)
        self.out = bank.add(m.out_proj.weight, "linear", tok_dtype, bias=m.out_proj.bias)

    def __call__(self, x, mem, training, p_drop, causal=False):
        """x: (B,Lq,d), mem: (B,Lk,d) or None for self-attention."""
        B, Lq, d = x.shape
        dh = d // self.h
        if self.cross:
            Lk = mem.shape[1]
            q = linear(x.reshape(B * Lq, d), self.q)
            kv = linear(mem.reshape(B * Lk, d), self.kv)
            k_off, v_off = 0, d
        else:
            Lk = Lq
            q = kv = linear(x.reshape(B * Lq, d), self.qkv)
            k_off, v_off = d, 2 * d
        o = ops.AttnCoreFn.apply(q, kv, B, self.h, Lq, Lk, dh, k_off, v_off, causal, float(p_drop) if training else 0.0)
        return linear(o, self.out).view(B, Lq, d)


class EncoderLayer:
    """nn.TransformerEncoderLayer defaults: post-norm, ReLU FFN (style_encoder.py:181-187)."""

    def __init__(self, bank, layer: nn.TransformerEncoderLayer):
        self.l = layer
        self.attn = MHA(bank, layer.self_attn, cross=False)
        self.ff1 = bank.add(layer.linear1.weight, "linear", tok_dtype, bias=layer.linear1.bias)
        self.ff2 = bank.add(layer.linear2.weight, "linear", tok_dtype, bias=layer.linear2.bias)

    def __call__(self, x, training):
        l, p = self.l, self.l.dropout.p
        B, L, d = x.shape
        if not config.fused_tokens:
            x = layer_norm(x + ops.dropout(self.attn(x, None, training, l.self_attn.dropout), l.dropout1.p, training), l.norm1)
            h = ops.dropout(linear(x.reshape(B * L, d), self.ff1, relu=True), p, training)
            h = ops.dropout(linear(h, self.ff2), l.dropout2.p, training).view(B, L, d)
            return layer_norm(x + h, l.norm2)
        _, x = ops.add_drop_ln(x, self.attn(x, None, training, l.self_attn.dropout), l.norm1, l.dropout1.p, training)
        h = ops.ffn(x.reshape(B * L, d), self.ff1, self.ff2, p, training)
        _, y = ops.add_drop_ln(x, h.view(B, L, d), l.norm2, l.dropout2.p, training)
        return y


class DecoderLayer:
    """nn.TransformerDecoderLayer(norm_first=True) (new_decoder.py:111-118)."""

    def __init__(self, bank, layer: nn.TransformerDecoderLayer):
        self.l = layer
        self.sa = MHA(bank, layer.self_attn, cross=False)
        self.ca = MHA(bank, layer.multihead_attn, cross=True)
        self.ff1 = bank.add(layer.linear1.weight, "linear", tok_dtype, bias=layer.linear1.bias)
        self.ff2 = bank.add(layer.linear2.weight, "linear", tok_dtype, bias=layer.linear2.bias)

    def __call__(self, x, memory, training):
        l = self.l
        B, L, d = x.shape
        if not config.fused_tokens:
            h = layer_norm(x, l.norm1)
            x = x + ops.dropout(self.sa(h, None, training, l.self_attn.dropout, causal=True), l.dropout1.p, training)
            h = layer_norm(x, l.norm2)
            x = x + ops.dropout(self.ca(h, memory, training, l.multihead_attn.dropout), l.dropout2.p, training)
            h = layer_norm(x, l.norm3)
            h = ops.dropout(linear(h.reshape(B * L, d), self.ff1, relu=True), l.dropout.p, training)
            h = ops.dropout(linear(h, self.ff2), l.dropout3.p, training).view(B, L, d)
            return x + h
        h = layer_norm(x, l.norm1)
        x, h = ops.add_drop_ln(x, self.sa(h, None, training, l.self_attn.dropout, causal=True), l.norm2, l.dropout1.p, training)
        x, h = ops.add_drop_ln(x, self.ca(h, memory, training, l.multihead_attn.dropout), l.norm3, l.dropout2.p, training)
        h = ops.ffn(h.reshape(B * L, d), self.ff1, self.ff2, l.dropout.p, training)
        x, _ = ops.add_drop_ln(x, h.view(B, L, d), None, l.dropout3.p, training)
        return x

    # ---- incremental (KV-cached) decoding: eval / no-grad only -----------------------------------
    def memory_kv(self, memory):
        """K | V projections of the cross-attention memory, (B*Lm, 2d): computed once per generated sequence."""
        B, Lm, d = memory.shape
        return linear(memory.reshape(B * Lm, d), self.ca.kv), Lm

    def step(self, x_t, kv_self, mem_kv):
        """One new token x_t (B,1,d) through the layer.  kv_self: cached (B, t, 2d) K|V of the earlier tokens of THIS layer (or
        None); returns (y_t, kv_self grown by one).  Causality makes the earlier tokens' K/V independent of later ones, so
        this equals row t of the full-sequence layer (new_decoder.py:294-314 recomputes all rows every step)."""
        l = self.l
        B, _, d = x_t.shape
        dh = d // self.sa.h
        h = layer_norm(x_t, l.norm1).reshape(B, d)
        qkv = linear(h, self.sa.qkv)                                   # (B, 3d): q | k | v of the new token
        new_kv = qkv[:, d:].reshape(B, 1, 2 * d)
        kv_self = new_kv if kv_self is None else torch.cat([kv_self, new_kv], dim=1)
        t1 = kv_self.shape[1]
        q = qkv[:, :d].contiguous()
        o = ops.AttnCoreFn.apply(q, kv_self.reshape(B * t1, 2 * d), B, self.sa.h, 1, t1, dh, 0, d, False, 0.0)
        x = x_t.reshape(B, d) + linear(o, self.sa.out)
        h = layer_norm(x, l.norm2)
        kvm, Lm = mem_kv
        o = ops.AttnCoreFn.apply(linear(h, self.ca.q), kvm, B, self.ca.h, 1, Lm, dh, 0, d, False, 0.0)
        x = x + linear(o, self.ca.out)
        h = layer_norm(x, l.norm3)
        x = x + linear(linear(h, self.ff1, relu=True), self.ff2)
        return x.view(B, 1, d), kv_self
