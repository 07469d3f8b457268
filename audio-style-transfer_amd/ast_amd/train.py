"""Train-step harness: the step `train2.ipynb` runs (reconstructed -- the notebook
is not in the reference checkout) over the drop-in modules.

    encoders -> D phase (adversarial_loss on detached embeddings, Adam on D)
             -> G phase (decoder teacher forcing + recon/InfoNCE/margin/HSIC/
                adversarial-generator losses, grad-norm clip, Adam on encoders+decoder)

MI355X-first choices: parameters and gradients of each optimiser group live in
ONE flat f32 buffer (views handed to the modules), so zero_grad is one memset,
clipping is one reduction launch, Adam is one launch and the data-parallel
exchange is one RCCL all-reduce per group; the whole step (hundreds of small
launches) is captured into a hipGraph and replayed.
"""
from __future__ import annotations

import dataclasses
import os

import torch
import torch.distributed as dist

from . import config, ops, streams
from ._lib import check, lib, ptr, stream
from .content_encoder import ContentEncoder
from .discriminator import Discriminator
from .losses import adversarial_loss, disentanglement_loss, infoNCE_loss, margin_loss
from .new_decoder import Decoder, compute_comprehensive_loss
from .parallel import gather_rows, global_labels
from .style_encoder import StyleEncoder, _module_bank, class_prototypes, initialize_weights
from . import layers as layers_mod


# layout of Trainer.hyper (device-side step scalars): [lr, max_norm] pairs as ast_adam_dev reads them, then the loss weights
H_LR_G, H_LR_D, H_W_REC, H_W_NCE, H_W_MARGIN, H_W_HSIC, H_W_ADV = 0, 2, 4, 5, 6, 7, 8


@dataclasses.dataclass
class TrainConfig:
    lr_g: float = 1e-4
    lr_d: float = 1e-4
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    max_grad_norm: float = 1.0
    # loss weights (unknown in the reference; 1.0 = plain sum) and curriculum gates (README.md:144-150)
    w_rec: float = 1.0
    w_nce: float = 1.0
    w_margin: float = 1.0
    w_hsic: float = 1.0
    w_adv: float = 1.0
    use_hsic: bool = True
    use_nce: bool = True
    use_adv: bool = True
    use_graph: bool = True
    segmented: bool = False       # capture the step as 3 graphs (what world > 1 uses); for testing on one GPU
    multi_stream: bool = True     # style encoder / content encoder / decoder target-encoder on separate HIP streams
    dropout: bool = True          # nn.Dropout(0.1) as constructed by the reference
    # world > 1 only: reproduce the single-process step on the GLOBAL batch (SURVEY 8(e)(2)+(3)): BatchNorm statistics
    # all-reduced per layer, style/content embeddings all-gathered for InfoNCE / HSIC / class prototypes / margin.
    # Runs on ONE stream (every collective then forks from and joins into the capture's origin stream: streams.py) and, over
    # RCCL, as ONE captured hipGraph with all ~70 small collectives inside; over gloo eagerly.  The gradient exchange is
    # bucketed by model (decoder, content encoder, style encoder) and overlapped with the rest of the backward pass.
    loss_matched: bool = False
    bucketed: bool = True         # loss-matched mode: per-model gradient buckets, all-reduced while the backward pass goes on
    decoder: str = "new"          # "new" = new_decoder.Decoder (north star); "simple" = SimpleDecoder_TransformerOnly.Decoder (8(f)1)
    grad_wire: str = "auto"       # dtype of the gradient all-reduce: "f32", "bf16", or "auto" = bf16 in the bf16 compute
                                  # mode over RCCL, else f32 (env AST_GRAD_WIRE overrides)
    overlap_d: bool = True        # one GPU: discriminator phase on its own stream beside the decoder forward
    keep_grads: bool = False      # eager mode: keep a copy of the (all-reduced) generator gradient for tests


def curriculum_gates(progress: float):
    """README.md:144-150: 0-20 % recon only, 20-40 % + disentanglement, 40-60 % + contrastive, 60-100 % + adversarial."""
    return dict(use_hsic=progress >= 0.2, use_nce=progress >= 0.4, use_adv=progress >= 0.6)


class FlatGroup:
    """Parameters of several modules re-seated as views of one flat buffer, with a flat
    gradient buffer, Adam moments and device-side step counter."""

    def __init__(self, modules, device):
        self.params = [p for m in modules for p in m.parameters()]
        n = sum((p.numel() + 3) // 4 * 4 for p in self.params)      # every view starts 16-byte aligned (float4 kernels)
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        self.step = torch.zeros(1, dtype=torch.int64, device=device)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=device)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += (k + 3) // 4 * 4
        # [offset, length) of every module's parameters inside the flat buffers (the modules' parameters are contiguous)
        self.slices, off = [], 0
        for m in modules:
            k = sum((p.numel() + 3) // 4 * 4 for p in m.parameters())
            self.slices.append((off, k))
            off += k
        self._pending = []

    def zero_grad(self):
        self.flat_g.zero_()

    def all_reduce(self, world, wire_dtype=torch.float32, force=False):
        """Mean of the flat gradient over ranks: ONE collective per group.  wire_dtype=bf16 halves the bytes on the
        xGMI ring (124 MB -> 62 MB for encoders+decoder; the all-reduce sits on the critical path between backward and
        Adam): cast kernel -> all-reduce -> cast back + scale.  Used in the bf16 compute mode only."""

        def scale(t, s):
            check(lib().ast_scale(ptr(t), None, s, ptr(t), t.numel(), 0, stream()), "ast_scale")
        if world <= 1 and not force:
            return
        # force (tests): a one-rank group runs the SAME calls as world > 1 -- cast, collective, cast back, scale
        if wire_dtype == torch.bfloat16 and self.n >= (1 << 20):
            if getattr(self, "_wire", None) is None:
                self._wire = torch.empty(self.n, dtype=torch.bfloat16, device=self.flat_g.device)
            check(lib().ast_cast(ptr(self.flat_g), 0, ptr(self._wire), 1, self.n, stream()), "ast_cast")
            dist.all_reduce(self._wire, op=dist.ReduceOp.SUM)
            check(lib().ast_cast(ptr(self._wire), 1, ptr(self.flat_g), 0, self.n, stream()), "ast_cast")
            if world > 1:
                scale(self.flat_g, 1.0 / world)
            return
        dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
        if world > 1:
            scale(self.flat_g, 1.0 / world)

    def all_reduce_bucket(self, index, world, wire_dtype=torch.float32, force=False):
        """Start the SUM all-reduce of module `index`'s slice of the flat gradient (async: the collective runs on the process
        group's stream while the caller goes on with the backward pass); finish_buckets() waits and takes the mean."""
        if world <= 1 and not force:
            return
        off, k = self.slices[index]
        seg = self.flat_g[off:off + k]
        if wire_dtype == torch.bfloat16 and k >= (1 << 20):
            if getattr(self, "_wire", None) is None:
                self._wire = torch.empty(self.n, dtype=torch.bfloat16, device=self.flat_g.device)
            wire = self._wire[off:off + k]
            check(lib().ast_cast(ptr(seg), 0, ptr(wire), 1, k, stream()), "ast_cast")
            self._pending.append((dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True), seg, wire))
        else:
            self._pending.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), seg, None))

    def finish_buckets(self, world):
        pending, self._pending = self._pending, []
        for work, seg, wire in pending:
            work.wait()                                # the current stream waits for the collective (capturable)
            if wire is not None:
                check(lib().ast_cast(ptr(wire), 1, ptr(seg), 0, seg.numel(), stream()), "ast_cast")
        if pending and world > 1:
            check(lib().ast_scale(ptr(self.flat_g), None, 1.0 / world, ptr(self.flat_g), self.n, 0, stream()), "ast_scale")

    def adam(self, hyper, betas, eps, clip=True):
        """hyper: DEVICE tensor [lr, max_norm], read by the kernel at run time (a replayed graph follows an LR schedule).
        clip=False skips the gradient-norm pass altogether (structural: part of the graph key)."""
        if clip:
            self.gnorm_sq.zero_()
            check(lib().ast_sumsq(ptr(self.flat_g), self.n, ptr(self.gnorm_sq), stream()), "ast_sumsq")
        check(lib().ast_counter_incr(ptr(self.step), stream()), "ast_counter_incr")
        check(lib().ast_adam_dev(ptr(self.flat_p), ptr(self.flat_g), ptr(self.m), ptr(self.v), self.n, ptr(hyper), betas[0], betas[1], eps,
                                 0.0, ptr(self.step), ptr(self.gnorm_sq) if clip else None, stream()), "ast_adam_dev")


class Trainer:
    def __init__(self, cfg: TrainConfig = None, device="cuda:0", rank=0, world=1, seed=1234, init="unit_gammas"):
        self.cfg = cfg or TrainConfig()
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        torch.manual_seed(seed)                      # identical replicas on every rank
        ops._DropState.seed = 0x5EED + 1000003 * int(rank)   # ... but independent dropout masks: ranks hold different clips
        self.style, self.content = StyleEncoder(), ContentEncoder()
        self._simple = self.cfg.decoder == "simple"
        if self._simple:
            from . import SimpleDecoder_TransformerOnly as SD
            self.decoder, self._rec_loss = SD.Decoder(), SD.compute_comprehensive_loss
        else:
            self.decoder, self._rec_loss = Decoder(), compute_comprehensive_loss
        self.disc = Discriminator()
        if init in ("reference", "unit_gammas"):
            # NOT the reference's initial state: a freshly built reference decoder has all BN/LN gammas = 0 and
            # outputs exactly 0 (new_decoder.py:134-143).  "unit_gammas" (the default; "reference" is the
            # old name of the same thing) gives the decoder gammas their conventional value 1 so the step does real
            # work; init="as_constructed" keeps the reference's zeros.
            with torch.no_grad():
                for name, p in self.decoder.named_parameters():
                    if p.dim() == 1 and "weight" in name:
                        p.fill_(1.0)
        if not self.cfg.dropout:
            for m in (self.style, self.content, self.decoder):
                for mod in m.modules():
                    if isinstance(mod, torch.nn.Dropout):
                        mod.p = 0.0
                    if isinstance(mod, torch.nn.MultiheadAttention):
                        mod.dropout = 0.0
        for m in (self.style, self.content, self.decoder, self.disc):
            m.to(self.device).train()
        self.G = FlatGroup([self.style, self.content, self.decoder], self.device)
        self.D = FlatGroup([self.disc], self.device)
        # Step scalars live on the DEVICE (learning rates, clip norm, loss weights): the kernels read them at run time, so an
        # LR schedule or a ramped adversarial weight (scheduler.step(), lambda_adv(t)) changes values between
        # replays of ONE captured graph.  step() pushes cfg's current values before every step when they changed.
        self.hyper = torch.zeros(16, dtype=torch.float32, device=self.device)
        self._hyper_sent = None
        wire = os.environ.get("AST_GRAD_WIRE", self.cfg.grad_wire)
        if wire == "auto":
            nccl = dist.is_initialized() and dist.get_backend() == "nccl"
            wire = "bf16" if (config.compute_dtype == torch.bfloat16 and nccl) else "f32"
        self._wire_dtype = torch.bfloat16 if wire == "bf16" else torch.float32
        self._force_coll = os.environ.get("AST_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()
        self._matched = bool(self.cfg.loss_matched and (world > 1 or self._force_coll))
        if self._matched:
            ops.set_sync_bn(world, force=self._force_coll)
            self.cfg = dataclasses.replace(self.cfg, multi_stream=False, overlap_d=False)
        # Data parallel, default mode: the two gradient all-reduces are RCCL calls INSIDE the captured step (stream-ordered,
        # capturable through torch's NCCL process group), so world > 1 replays ONE graph with the discriminator phase
        # overlapped, exactly as one GPU does, instead of three graphs with eager collectives between them.  A probe
        # capture of a small all-reduce decides (any failure -> the three-graph form); AST_DIST_IN_GRAPH=0 turns it off.
        self._dist = world > 1 or self._force_coll
        self._dist_in_graph = None                   # decided at the first graph step (probe)
        self._replicas_checked = False
        self._glob = None
        self._stream_d = None
        self._pre_zeroed = False
        self._stream_fe = None
        self._graphs = {}
        self._streams = None
        self._y_emb = None
        self._carry = None
        self._parts = None
        self._frontend = None
        self._frontend_cqt = None
        self._fe_bufs = {}
        self._static = None
        self.losses = {}

    # ------------------------------------------------------------------ one eager step
    def _forward_backward(self, x, labels_host, defer_d=False):
        c = self.cfg
        if self._pre_zeroed:                         # _prepare_beside_frontend cleared both gradient buffers on a branch stream
            self._pre_zeroed = False
        else:
            self.G.zero_grad()
            self.D.zero_grad()
        if ops._DropState.counter is None or ops._DropState.counter.device != self.device:
            ops._DropState.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        check(lib().ast_counter_incr(ptr(ops._DropState.counter), stream()), "ast_counter_incr")
        y = x[..., :513]
        with ops.shared_nhwc(x, config.compute_dtype):      # one input conversion for both encoders, inside this step
            style_emb, class_emb, content_emb, y_emb = self._encoders(x, y, labels_host)
        self._y_emb = y_emb
        if defer_d:
            return y, style_emb, class_emb, content_emb, None
        d_loss = self._d_phase(style_emb, class_emb, content_emb, labels_host)
        return y, style_emb, class_emb, content_emb, d_loss

    def _encoders(self, x, y, labels_host):
        c = self.cfg
        y_emb = None
        if c.multi_stream:
            # three mutually independent CNN branches: run them concurrently (forward here; autograd replays
            # each node's backward on the stream of its forward, so backward overlaps the same way)
            main = torch.cuda.current_stream()
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
            s1, s2, s3 = self._streams
            for st in self._streams:                 # fork after the shared input conversion (shared_nhwc, on main)
                streams.fork(st, main)
                if ops._SharedInput.y is not None:
                    ops._SharedInput.y.record_stream(st)
            order = os.environ.get("AST_BRANCH_ORDER", "ysc")   # creation order = reverse backward priority; y first measured 0.03 ms better
            for b in order:
                if b == "s":
                    with torch.cuda.stream(s1):
                        style_emb, class_emb = self.style(x, labels_host)
                elif b == "c":
                    with torch.cuda.stream(s2):
                        content_emb = self.content(x)
                else:
                    with torch.cuda.stream(s3):
                        if not self._simple:
                            y_emb = self.decoder.encode_target(y)
            for st in self._streams:
                streams.join(main, st)
            for t in (style_emb, class_emb, content_emb, y_emb):
                if t is not None:
                    t.record_stream(main)
        elif self._matched:
            style_emb, _ = self.style(x, None)
            content_emb = self.content(x)
            # Stage boundaries of the bucketed backward pass (_backward_bucketed): everything downstream consumes VIEWS of the
            # encoders' outputs.  backward(inputs=[t]) EXECUTES t's grad_fn (to fire its retain-grad hook) and then frees that
            # node's saved tensors; for the raw outputs that node is the encoder's last layer, which the next stage still has to
            # run.  A view's grad_fn is a ViewBackward: no kernel, nothing saved.
            self._enc_raw = (style_emb, content_emb)
            style_emb, content_emb = style_emb.view_as(style_emb), content_emb.view_as(content_emb)
            style_g = gather_rows(style_emb, self.rank, self.world, force=self._force_coll)
            labels_g = global_labels(labels_host, self.world)
            class_emb = class_prototypes(style_g, labels_g)            # prototypes over the global batch
            self._glob = (style_g, labels_g)
        else:
            style_emb, class_emb = self.style(x, labels_host)
            content_emb = self.content(x)
        return style_emb, class_emb, content_emb, y_emb

    def _d_phase(self, style_emb, class_emb, content_emb, labels_host):
        """Discriminator step on the detached embeddings (losses.py:69-79 compute_for_discriminator=True)."""
        bank_d = _module_bank(self.disc)
        bank_d.prepare(True)
        bank_d.hold = True
        d_loss, _ = adversarial_loss(style_emb.detach(), class_emb.detach(), content_emb.detach(), self.disc, labels_host, True)
        d_loss.backward()
        bank_d.hold = False
        return d_loss

    def _aux_losses(self, labels_host, style_emb, class_emb, content_emb):
        """The embedding losses that need neither the decoder nor the discriminator: margin, InfoNCE, HSIC."""
        c = self.cfg
        parts = {}
        terms = [(H_W_MARGIN, margin_loss(class_emb))]            # (index of the weight in self.hyper, loss term)
        style_b, labels_b, content_b = style_emb, labels_host, ops.mean_over_sections(content_emb)
        if self._matched:                    # batch-coupled terms on the gathered global batch
            style_b, labels_b = self._glob
            content_b = gather_rows(content_b, self.rank, self.world, force=self._force_coll)
        if c.use_nce:
            nce = infoNCE_loss(style_b, labels_b)
            terms.append((H_W_NCE, nce))
            parts["nce"] = nce.detach()
        if c.use_hsic:
            hs = disentanglement_loss(style_b, content_b)
            terms.append((H_W_HSIC, hs))
            parts["hsic"] = hs.detach()
        # ONE scalar leaves this stream: summed here (weights read from the device), not on the main stream.  Handing the three
        # terms to the main stream's total one by one gave every term its own cross-stream edge in the captured graph (forward
        # and backward); hipGraphLaunch then took 4.8 instead of 3.9 ms of host time and the replayed step 7.1 instead of
        # 6.2 ms -- with identical kernels (profiles/r03/bisect*.txt).
        return [(-1, ops.weighted_sum(self.hyper, terms))], parts

    def _g_phase(self, x, y, labels_host, style_emb, class_emb, content_emb, before_adv=None, aux=None, side=None, mid=None):
        c = self.cfg
        cls_rows = ops.class_rows(class_emb, labels_host)
        if self._simple:
            out = self.decoder(content_emb, cls_rows, y=y)
        else:
            out = self.decoder(content_emb, cls_rows, y=y, y_embeddings=self._y_emb)
        rec = self._rec_loss(out, y)
        terms = [(H_W_REC, rec["total_loss"])]
        parts = {"rec": rec["total_loss"].detach()}
        def adv_term():
            bank_d = _module_bank(self.disc)
            bank_d.prepare(True)          # D weights changed in the D phase
            bank_d.hold = True
            _, g = adversarial_loss(style_emb, class_emb, content_emb, self.disc, labels_host, False)
            bank_d.hold = False
            return g

        g_adv = None
        if mid is not None:
            mid()                         # (data parallel: D's gradient exchange + optimiser step, behind the decoder forward)
        if side is not None and aux is None:
            # The embedding losses and the generator's adversarial term run on the side stream (behind the D phase and D's
            # Adam step, which the adversarial term needs anyway).  They are created AFTER the decoder's nodes, so the
            # autograd engine runs their backward BEFORE the decoder's -- on the side stream, beside it.
            with torch.cuda.stream(side):
                aux = self._aux_losses(labels_host, style_emb, class_emb, content_emb)
                if c.use_adv:
                    g_adv = adv_term()
            for _, t in aux[0]:
                t.record_stream(torch.cuda.current_stream())
            if g_adv is not None:
                g_adv.record_stream(torch.cuda.current_stream())
        if before_adv is not None:
            before_adv()                  # join the side stream
        aux_terms, aux_parts = aux if aux is not None else self._aux_losses(labels_host, style_emb, class_emb, content_emb)
        terms += aux_terms
        parts.update(aux_parts)
        if c.use_adv:
            if g_adv is None:
                g_adv = adv_term()
            terms.append((H_W_ADV, g_adv))
            parts["adv_g"] = g_adv.detach()
        # total = w_rec rec + w_margin margin + w_nce nce + w_hsic hsic + w_adv adv_g, weights read from the device: one launch
        total = ops.weighted_sum(self.hyper, terms)
        if self._matched and c.bucketed and self._dist and not self._simple:
            self._backward_bucketed(total, style_emb, content_emb)
        else:
            if os.environ.get("AST_WGRAD_STREAM", "0") != "0" and not self._matched:
                if getattr(self, "_wstream", None) is None:
                    self._wstream = torch.cuda.Stream(device=self.device)
                ops._WgradStream.stream = self._wstream
            try:
                with layers_mod.parallel_flush():     # the three generator banks' gradient flushes side by side
                    total.backward()
            finally:
                ops._WgradStream.stream = None
        parts["total"] = total.detach()
        return parts

    def _backward_bucketed(self, total, style_emb, content_emb):
        """Backward pass in three stages, one gradient bucket each (loss-matched data parallel, one stream):
          1. from the total down to the encoders' OUTPUTS and into every decoder parameter (the decoder's weight bank is flushed
             by the end-of-backward callback of this stage) -> the decoder's slice of the flat gradient starts its all-reduce;
          2. content encoder, from the gradient of its output -> its slice starts;  3. style encoder -> its slice.
        The collectives run on the process group's stream while the next stage's kernels run on ours; G's optimiser step waits
        for all three (FlatGroup.finish_buckets).  G's modules are [style, content, decoder]: slices 0, 1, 2."""
        G = self.G
        mods = (self.style, self.content, self.decoder)
        par = [list(m.parameters()) for m in mods]
        for t in (style_emb, content_emb):
            t.retain_grad()
        raw_s, raw_c = self._enc_raw                  # style_emb / content_emb are views of these (see _encoders)
        torch.autograd.backward(total, inputs=[style_emb, content_emb] + par[2])
        G.all_reduce_bucket(2, self.world, self._wire_dtype, force=self._force_coll)
        g_s, g_c = style_emb.grad, content_emb.grad
        torch.autograd.backward([raw_c], [g_c], inputs=par[1])
        G.all_reduce_bucket(1, self.world, self._wire_dtype, force=self._force_coll)
        torch.autograd.backward([raw_s], [g_s], inputs=par[0])
        G.all_reduce_bucket(0, self.world, self._wire_dtype, force=self._force_coll)
        self._enc_raw = None

    # The step as three segments; the data-parallel gradient all-reduces sit between them.
    def _seg_a(self, x, labels_host):
        ops.stat_arena_reset(self.device)            # every BatchNorm / InstanceNorm statistics table of the step: ONE memset
        self._run_frontend(x)
        self._carry = self._forward_backward(x, labels_host)          # zero grads, encoders fwd, D-phase fwd+bwd

    def _seg_b(self, x, labels_host):
        c = self.cfg
        y, style_emb, class_emb, content_emb, d_loss = self._carry
        if c.multi_stream and c.overlap_d:
            # D's Adam step, the embedding losses and the generator's adversarial term beside the decoder (see _step_overlapped)
            main = torch.cuda.current_stream()
            if self._stream_d is None:
                self._stream_d = torch.cuda.Stream(device=self.device)
            sd = self._stream_d
            streams.fork(sd, main)
            with torch.cuda.stream(sd):
                self.D.adam(self.hyper[H_LR_D:H_LR_D + 2], c.betas, c.eps, c.max_grad_norm > 0)
                self.D.zero_grad()
            self._parts = self._g_phase(x, y, labels_host, style_emb, class_emb, content_emb, before_adv=lambda: streams.join(main, sd), side=sd)
        else:
            self.D.adam(self.hyper[H_LR_D:H_LR_D + 2], c.betas, c.eps, c.max_grad_norm > 0)
            self.D.zero_grad()
            self._parts = self._g_phase(x, y, labels_host, style_emb, class_emb, content_emb)
        self._parts["adv_d"] = d_loss.detach()
        self._carry = None

    def _seg_c(self, x, labels_host):
        c = self.cfg
        self.G.adam(self.hyper[H_LR_G:H_LR_G + 2], c.betas, c.eps, c.max_grad_norm > 0)

    def _step_overlapped(self, x, labels_host):
        """One GPU, one graph: the discriminator phase (D forward/backward + D's Adam) runs on its own stream beside the
        decoder forward and the D-independent losses; the streams join before the generator's adversarial term, which
        needs the updated discriminator (same arithmetic and order of updates as _step_body)."""
        c = self.cfg
        ops.stat_arena_reset(self.device)            # every BatchNorm / InstanceNorm statistics table of the step: ONE memset
        held = self._prepare_beside_frontend()
        try:
            return self._step_overlapped_held(x, labels_host)
        finally:
            self._pre_zeroed = False
            for b in held:
                b.hold = False

    def _step_overlapped_held(self, x, labels_host):
        c = self.cfg
        self._run_frontend(x)
        y, style_emb, class_emb, content_emb, _ = self._forward_backward(x, labels_host, defer_d=True)
        main = torch.cuda.current_stream()
        if self._stream_d is None:
            self._stream_d = torch.cuda.Stream(device=self.device)
        sd = self._stream_d
        streams.fork(sd, main)
        def d_update():
            self.D.adam(self.hyper[H_LR_D:H_LR_D + 2], c.betas, c.eps, c.max_grad_norm > 0)
            self.D.zero_grad()
        mid = None
        with torch.cuda.stream(sd):
            d_loss = self._d_phase(style_emb, class_emb, content_emb, labels_host)
            if not self._dist:
                d_update()
        if self._dist:
            # Data parallel: D's gradient all-reduce is issued from the capture's ORIGIN stream, after the decoder forward has been
            # enqueued there (mid-way through _g_phase), and the side stream goes on from it.  RCCL's internal stream forks from
            # and joins into the stream the collective is called on; called on the forked side stream that join would be "a forked
            # stream waits on its own child" -- the pattern that crashes hipStreamEndCapture (streams.py).
            def mid():
                streams.join(main, sd)
                self.D.all_reduce(self.world, force=self._force_coll)
                streams.fork(sd, main)
                with torch.cuda.stream(sd):
                    d_update()
        d_loss.record_stream(main)
        # margin / InfoNCE / HSIC go to the side stream too, but are CREATED after the decoder's nodes (inside _g_phase):
        # created before them, their backward ran after the decoder's and held the three encoder branches back (+0.9 ms)
        self._parts = self._g_phase(x, y, labels_host, style_emb, class_emb, content_emb, before_adv=lambda: streams.join(main, sd), side=sd, mid=mid)
        self._parts["adv_d"] = d_loss.detach()
        if self._dist:
            self.G.all_reduce(self.world, self._wire_dtype, force=self._force_coll)
        if c.keep_grads:
            self.last_grad_g = self.G.flat_g.clone()
        self.G.adam(self.hyper[H_LR_G:H_LR_G + 2], c.betas, c.eps, c.max_grad_norm > 0)
        return self._parts

    def _prepare_beside_frontend(self):
        """Spectral-norm iteration + weight packing of the three generator banks (5 launches each, ~0.1 ms on the step's
        longest chain: tools/graph_critical_path.py) depend on the weights only: they go to the branch streams BEFORE the
        front end runs on the main stream, and the modules' own prepare() calls are held for this step."""
        if not self.cfg.multi_stream or os.environ.get("AST_EARLY_PREPARE", "1") == "0":
            return []
        main = torch.cuda.current_stream()
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        held = []
        st0 = self._streams[0]
        streams.fork(st0, main)
        with torch.cuda.stream(st0):                 # gradients are first touched in the backward pass, after the branches join
            self.G.zero_grad()
            self.D.zero_grad()
        self._pre_zeroed = True
        for st, mod in zip(self._streams, (self.style, self.content, self.decoder)):
            bank = _module_bank(mod)
            if bank.hold:
                continue
            streams.fork(st, main)
            with torch.cuda.stream(st):
                bank.prepare(True)
            bank.hold = True
            held.append(bank)
        return held

    def _one_graph(self):
        """The step as ONE captured graph: a single GPU, or data parallel with the collectives inside the graph."""
        if self.cfg.segmented:
            return False
        if not self._dist:
            return self.world == 1
        # data parallel: the collectives must be capturable; the loss-matched mode is a single-stream step (its sync-BN and gather
        # collectives cannot sit BETWEEN three graphs: without in-graph collectives it runs eagerly -- see step())
        return bool(self._dist_in_graph) and ((self.cfg.multi_stream and self.cfg.overlap_d) or self._matched)

    def _probe_collective_capture(self):
        """Can this process group's all-reduce be captured into a hipGraph and replayed?  Decided once, by doing it on
        a small tensor; every rank runs the same collectives, and the ranks agree on the outcome (MIN) at the end."""
        if os.environ.get("AST_DIST_IN_GRAPH", "1") == "0" or not self.cfg.use_graph or dist.get_backend() != "nccl":
            return False
        ok = 1.0
        try:
            t = torch.ones(4096, device=self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_reduce(t)                       # communicator set-up happens outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # (the process group's watchdog thread polls events meanwhile)
                dist.all_reduce(t)
            t.fill_(1.0)
            g.replay()
            torch.cuda.synchronize()
            if abs(float(t[0]) - dist.get_world_size()) > 1e-3:
                ok = 0.0
        except Exception as e:                           # noqa: BLE001 -- any failure means "use the three-graph form"
            print(f"[ast_amd] all-reduce inside a captured graph is not available ({type(e).__name__}: {e}); using three graphs", flush=True)
            ok = 0.0
            try:
                torch.cuda.synchronize()
            except Exception:                            # noqa: BLE001
                pass
        flag = torch.tensor([ok], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def _step_body(self, x, labels_host):
        if self._one_graph() and self.cfg.multi_stream and self.cfg.overlap_d:
            return self._step_overlapped(x, labels_host)
        self._seg_a(x, labels_host)
        if self._dist:
            self.D.all_reduce(self.world, force=self._force_coll)
        self._seg_b(x, labels_host)
        if self._dist:
            if self.G._pending:                # the bucketed exchange is already in flight (loss-matched mode)
                self.G.finish_buckets(self.world)
            else:
                self.G.all_reduce(self.world, self._wire_dtype, force=self._force_coll)
        if self.cfg.keep_grads:
            self.last_grad_g = self.G.flat_g.clone()
        self._seg_c(x, labels_host)
        return self._parts

    # ------------------------------------------------------------------ public API
    def set_frontend(self, waves, mean, std, cqt_mean=None, cqt_std=None):
        """Make the STFT front-end part of the step: every step() first runs the fused STFT + z-score +
        sectioning kernel (utilityFunctions.py:12-37,240-263; dataloader.py:9-13) from these device-resident
        waveforms straight into bins [0,513) of x (the collate layout of dataloader.py:123-147).

        The Trainer OWNS its front-end buffers: calling this again with tensors of the same shapes (a new batch of
        waveforms, other statistics) copies into them, so captured graphs -- which bake the buffer addresses in --
        read the new values on their next replay.  New shapes get new buffers, and the graph cache key (which holds
        the buffer addresses) makes step() capture again."""
        def own(slot, t):
            # one persistent buffer per (slot, shape): a stream of mixed clip lengths (BASELINE configs[4]) alternates between
            # a few waveform shapes, and each keeps its buffers -- and with them its captured graph -- across the switches
            t = t.detach()
            key = (slot, tuple(t.shape), t.dtype, str(t.device))
            cur = self._fe_bufs.get(key)
            if cur is None:
                cur = self._fe_bufs[key] = t.contiguous().clone()
            elif cur.data_ptr() != t.data_ptr():
                cur.copy_(t)
            return cur
        self._frontend = (own("waves", waves), own("mean", mean), own("std", std))
        self._frontend_cqt = None if cqt_mean is None else (own("cqt_mean", cqt_mean), own("cqt_std", cqt_std))

    def _run_frontend(self, x):
        if self._frontend is None:
            return
        from .utilityFunctions import stft_sections
        waves, mean, std = self._frontend
        side = None
        if self._frontend_cqt is not None:
            from .cqt import cqt_sections
            if self.cfg.multi_stream and self._streams is not None and os.environ.get("AST_FRONTEND_FORK", "0") != "0":
                # the CQT (five dependent decimations, then the octave kernel: ~0.1 ms of small launches) and the STFT write
                # disjoint bins of x: side by side.  Opt-in (AST_FRONTEND_FORK=1): no measurable gain on the step
                if self._stream_fe is None:
                    self._stream_fe = torch.cuda.Stream(device=self.device)
                main, side = torch.cuda.current_stream(), self._stream_fe
                streams.fork(side, main)
                with torch.cuda.stream(side):
                    cqt_sections(waves, x, *self._frontend_cqt)
            else:
                cqt_sections(waves, x, *self._frontend_cqt)
        stft_sections(waves, mean, std, n_sections=x.shape[1], F_total=x.shape[-1], out=x)
        if side is not None:
            streams.join(torch.cuda.current_stream(), side)

    def step(self, x: torch.Tensor, labels_host: torch.Tensor):
        """x: (B,S,2,287,597) f32 on the device; labels on the HOST (balanced [0]*B/2+[1]*B/2 as
        dataloader.py:143-146 builds them).  Returns a dict of detached device scalars."""
        assert not labels_host.is_cuda, "pass labels on the host: avoids a device sync per step"
        self._sync_hyper()
        if self._dist and self._dist_in_graph is None:
            self._dist_in_graph = self._probe_collective_capture()
        if not self.cfg.use_graph or (self._matched and not self._dist_in_graph):
            self.losses = self._step_body(x, labels_host)
            return self.losses
        segmented = not self._one_graph()
        key = self._graph_key(x, labels_host, segmented)
        if key not in self._graphs:
            self._capture(key, x, labels_host, segmented)
            self._check_tok_programs(after_capture=True)
        graphs, static_x, outs = self._graphs[key]
        if static_x.data_ptr() != x.data_ptr():
            static_x.copy_(x)
        if not segmented:
            graphs[0].replay()
            if self._dist and self.world > 1 and not self._replicas_checked:
                self._check_replicas_after_first_replay()
        else:                                   # data parallel, three-graph form: the two flat-gradient all-reduces run between replays
            graphs[0].replay()
            if self._dist:
                self.D.all_reduce(self.world, force=self._force_coll)
            graphs[1].replay()
            if self._dist:
                self.G.all_reduce(self.world, self._wire_dtype, force=self._force_coll)
            graphs[2].replay()
        self.losses = outs
        self._check_tok_programs()
        return outs

    def _sync_hyper(self):
        """Push cfg's current step scalars to the device when they differ from what is there (one tiny launch whose values
        travel as kernel arguments: stream-ordered behind the previous step, nothing for the host to wait for)."""
        import ctypes
        c = self.cfg
        vals = [0.0] * 9
        vals[H_LR_G], vals[H_LR_G + 1], vals[H_LR_D], vals[H_LR_D + 1] = c.lr_g, c.max_grad_norm, c.lr_d, c.max_grad_norm
        vals[H_W_REC], vals[H_W_NCE], vals[H_W_MARGIN], vals[H_W_HSIC], vals[H_W_ADV] = c.w_rec, c.w_nce, c.w_margin, c.w_hsic, c.w_adv
        vals = tuple(float(v) for v in vals)
        if vals == self._hyper_sent:
            return
        arr = (ctypes.c_float * len(vals))(*vals)
        check(lib().ast_set_values(ptr(self.hyper), arr, len(vals), stream()), "ast_set_values")
        self._hyper_sent = vals

    def _check_tok_programs(self, after_capture=False):
        """Persistent token programs (AST_TOK_PROGRAMS=2, opt-in) depend on G workgroups landing on one XCD; a launch that did
        not get them reports it in a device status word.  Checked after every capture's warm-up steps and every 256 steps:
        on failure the trainer falls back to one launch per op (mode 1) and captures again."""
        if config.tok_programs != 2:
            return
        self._tok_checked = getattr(self, "_tok_checked", 0) + 1
        if not after_capture and self._tok_checked % 256:
            return
        from . import tokprog
        try:
            tokprog.check_status()
        except RuntimeError as e:
            print(f"[ast_amd] {e}; falling back to per-op token launches (AST_TOK_PROGRAMS=1)", flush=True)
            config.tok_programs = 1
            self._graphs.clear()

    def _check_replicas_after_first_replay(self):
        """Once, after the first replay with the collectives inside the graph: every rank must hold the same parameters
        (identical replicas + the MEAN gradient = identical updates).  A captured all-reduce that silently did nothing
        would leave them different; then the ranks resynchronise from rank 0 and continue in the three-graph form."""
        self._replicas_checked = True
        chk = torch.stack([self.G.flat_p.double().sum(), self.D.flat_p.double().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if bool(((hi - lo).abs() <= 1e-9 * (1.0 + hi.abs())).all()):
            return
        if self.rank == 0:
            print("[ast_amd] replicas diverged after a step with the collectives inside the graph: resynchronising and "
                  "falling back to the three-graph form", flush=True)
        for grp in (self.G, self.D):
            for t in (grp.flat_p, grp.m, grp.v, grp.step):
                dist.broadcast(t, src=0)
        self._dist_in_graph = False
        self._graphs.clear()

    def sync_buffers(self):
        """Data parallel, default mode: every rank's BatchNorm running statistics follow its own shard.  Average them
        over ranks (and keep rank 0's integer counters) so that a checkpoint reflects the global batch stream."""
        if self.world <= 1:
            return
        for m in (self.style, self.content, self.decoder, self.disc):
            for mod in m.modules():
                if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.running_mean is not None:
                    for t in (mod.running_mean, mod.running_var):
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                        t.div_(self.world)

    def save_checkpoint(self, path, **extra):
        """The reference's four-state_dict .pth (evaluation_style_transfer.py:246-252), written by rank 0 after
        sync_buffers()."""
        from .checkpoint import save_checkpoint
        self.sync_buffers()
        if self.rank == 0:
            save_checkpoint(path, self.content, self.style, self.decoder, self.disc, **extra)

    def _graph_key(self, x, labels_host, segmented):
        """Everything a captured step bakes in: shapes, labels (host-side class layout), dtype, curriculum gates, Adam's
        betas / eps (kernel arguments), whether the gradient norm is clipped at all, and the front-end buffer addresses.
        Changing any of them between steps captures a new graph instead of silently replaying the old values.  The
        learning rates, the clip norm and the loss weights are NOT in the key: the kernels read them from self.hyper."""
        c = self.cfg
        fe = tuple(t.data_ptr() for t in (self._frontend or ())) + tuple(t.data_ptr() for t in (self._frontend_cqt or ()))
        return (tuple(x.shape), tuple(labels_host.tolist()), config.compute_dtype, c.use_nce, c.use_hsic, c.use_adv, segmented,
                tuple(c.betas), c.eps, c.max_grad_norm > 0, fe)

    def _mutable_state(self):
        ts = [self.G.flat_p, self.G.m, self.G.v, self.G.step, self.D.flat_p, self.D.m, self.D.v, self.D.step]
        for m in (self.style, self.content, self.decoder, self.disc):
            ts += list(m.buffers())            # BN running stats / num_batches_tracked, spectral-norm u and v
        if ops._DropState.counter is not None:
            ts.append(ops._DropState.counter)
        return ts

    def _capture(self, key, x, labels_host, segmented=False):
        """Warm-up (allocations, weight banks, constants) + capture.  Warm-up steps are real steps, so every
        piece of mutable training state is snapshotted first and restored afterwards: capturing is side-effect free."""
        if ops._DropState.counter is None or ops._DropState.counter.device != self.device:
            ops._DropState.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        state = self._mutable_state()
        saved = [t.clone() for t in state]
        static_x = x.clone()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # warm-up: builds weight banks, const tensors, caches
                self._step_body(static_x, labels_host)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not segmented:
            dot = os.environ.get("AST_GRAPH_DOT")    # tools/graph_critical_path.py: the captured step's nodes and edges
            gph = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, **({"capture_error_mode": "thread_local"} if self._dist else {})), streams.capture_origin():
                outs = self._step_body(static_x, labels_host)
            if dot:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
                rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(gph.raw_cuda_graph()), dot.encode(), 1)      # 1 = verbose
                if rc != 0:
                    raise RuntimeError(f"hipGraphDebugDotPrint failed ({rc})")
                gph.instantiate()
            graphs = [gph]
        else:
            # three graphs over ONE memory pool: the autograd graph built while capturing segment A is walked
            # while capturing segment B, so B must see A's activations at the same addresses
            graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
            pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(graphs[0], pool=pool), streams.capture_origin():
                self._seg_a(static_x, labels_host)
            with torch.cuda.graph(graphs[1], pool=pool), streams.capture_origin():
                self._seg_b(static_x, labels_host)
            with torch.cuda.graph(graphs[2], pool=pool), streams.capture_origin():
                self._seg_c(static_x, labels_host)
            outs = self._parts
        with torch.no_grad():
            for t, sv in zip(state, saved):
                t.copy_(sv)
        self._graphs[key] = (graphs, static_x, outs)


def synthetic_waveform_batch(B, seconds, device, seed=1000, sr=22050):
    """Synthetic clips for the front-end-inclusive step: B mono waveforms (first half piano-like decaying
    partials, second half violin-like sustained harmonics with vibrato), RMS 0.07, plus N(0,1) initial values for
    the 84 z-scored CQT bins (overwritten by the device CQT when the front end is given cqt statistics) and per-bin statistics."""
    g = torch.Generator().manual_seed(seed)
    n = int(round(seconds * sr))
    t = torch.arange(n, dtype=torch.float32) / sr
    waves = []
    for i in range(B):
        f0 = 110.0 * 2 ** (torch.randint(0, 36, (1,), generator=g).item() / 12)
        y = torch.zeros(n)
        if i < B // 2:
            for h in range(1, 7):
                y += torch.exp(-3.0 * h * (t % 0.5)) * torch.sin(2 * torch.pi * f0 * h * t) / h
        else:
            for h in range(1, 9):
                y += torch.sin(2 * torch.pi * f0 * h * t + 0.3 * h * torch.sin(2 * torch.pi * 5.5 * t)) / h
        y += 0.01 * torch.randn(n, generator=g)
        waves.append(0.07 * y / y.pow(2).mean().sqrt())
    waves = torch.stack(waves).to(device)
    T = 1 + n // 256
    from .utilityFunctions import section_starts
    S = len(section_starts(T))
    x = torch.randn((B, S, 2, 287, 597), generator=g, dtype=torch.float32).to(device)     # bins 513.. are replaced by get_CQT when Trainer.set_frontend gets cqt stats
    mean = (0.01 * torch.randn(2, 513, generator=g)).to(device)
    std = (0.5 + torch.rand(2, 513, generator=g)).to(device)
    labels = torch.cat([torch.zeros(B // 2, dtype=torch.long), torch.ones(B - B // 2, dtype=torch.long)])
    return waves, x, mean, std, labels


def synthetic_batch(B, S, device, seed=1000):
    """Model-ready synthetic batch: x ~ N(0,1) (what z-scored spectrograms look like and what the
    reference's own smoke tests feed, test_correctness.ipynb cell 6), balanced labels."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, S, 2, 287, 597), generator=g, dtype=torch.float32)
    labels = torch.cat([torch.zeros(B // 2, dtype=torch.long), torch.ones(B - B // 2, dtype=torch.long)])
    return x.to(device), labels
