"""Run-time configuration of the HIP path."""
import torch

# storage / MFMA operand dtype of the image (CNN) activations: torch.float32 (exact f32
# MFMA, parity mode) or torch.bfloat16 (bf16 MFMA with f32 accumulation, throughput mode).
# Parameters, statistics, token tensors and losses are always f32.
compute_dtype = torch.float32


def set_compute_dtype(dtype):
    global compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    compute_dtype = dtype


# Token-path fusions (residual + dropout + LayerNorm in one launch, the FFN in two).  The per-op path is kept as
# the reference the fused kernels are tested against; AST_FUSED_TOKENS=0 selects it for A/B timing.
import os as _os
fused_tokens = _os.environ.get("AST_FUSED_TOKENS", "1") != "0"
# Transformer stacks as token op lists (ast_amd/tokprog.py, ast_tok_program): 0 = off (one autograd node per operator),
# 1 = one autograd node per STACK, its ops launched one kernel each (the engine's gradient sums, the separate ReLU-backward and
# LayerNorm launches disappear into fused epilogues), 2 = the same op lists walked by ONE persistent launch per few layers
# with grid barriers.  Both are parity-green (tests/test_gpu_tokprog.py) and neither is faster on MI355X (stacks
# forward + backward 347 / 401 / 509 us, the whole step unchanged), so the default stays 0.
tok_programs = int(_os.environ.get("AST_TOK_PROGRAMS", "0"))

# BatchNorm batch statistics accumulated in the producing conv GEMM's epilogue instead of a separate pass over its
# output (AST_FUSED_BN_STATS=0 keeps the separate pass: the reference path for tests and A/B timing).
fused_bn_stats = _os.environ.get("AST_FUSED_BN_STATS", "1") != "0"

# BatchNorm/ResBlock-tail backward: recompute the ReLU mask from the pre-activation (x, scale, shift) instead of
# reading the activation output (AST_BN_MASK_FROM_PREACT=0 reads y: the reference path for A/B timing).
bn_mask_from_preact = _os.environ.get("AST_BN_MASK_FROM_PREACT", "1") != "0"

# BatchNorm backward sums accumulated in the epilogue of the data-gradient GEMM that produces the layer's dy
# (AST_FUSED_BN_BWD=0 keeps the separate pass over dy and x).
fused_bn_bwd = _os.environ.get("AST_FUSED_BN_BWD", "1") != "0"

# copies of the weight-gradient staging of small (pixel-rich) conv layers that ast_wgrad_rep spreads its atomics over
wgrad_replicas = int(_os.environ.get("AST_WGRAD_REPLICAS", "8"))

# BatchNorm / InstanceNorm statistics finalize folded into the apply passes (ast_bn_apply_fwd / _bwd): every workgroup of the
# apply pass reduces the statistics table itself, so the ~78 single-wave finalize launches of a step disappear from its
# dependency chain (AST_FUSED_FINALIZE=0 keeps norm_finalize + affine_act: the reference path for tests and A/B timing).
# Opt-in: measured on one box against the separate launches, three runs each: 6.35 / 6.29 / 6.37 ms per step fused against
# 6.23 / 6.28 / 6.35 separate (profiles/r03/bisect6.txt) -- the table reduction in front of every apply workgroup costs what the
# finalize launch cost, the 78 nodes it removes are worth ~0.1 ms of dispatch, and the sum is a wash.
fused_finalize = _os.environ.get("AST_FUSED_FINALIZE", "0") != "0"

# Slab flush of the pixel-rich conv layers' weight gradients (ast_wgrad_slab + ast_slab_sum): every pixel slice of a launch stores
# its partial dW into its own copy (plain stores) and one launch per model sums the copies at the end of the backward pass, instead
# of f32 atomics into 8 replicas.  Value = copies kept per small weight (also the cap on a launch's pixel slices); 0 = replicas.
# Opt-in: isolated launches gain 2-5 us per layer (profiles/r03/wg_rows_layers.txt), the step does not -- 6.34 / 6.24 / 6.29 ms with
# 128 slabs against 6.27 / 6.24 / 6.27 with replicas (ab_slabs_step.txt), 6.16 / 6.15 / 6.20 against 6.15 / 6.13 / 6.10 with the
# deferred launches (ab_defer_flush.txt): the summing pass and the extra 19-25 MB of partial tiles per layer cost what the atomics did.
wgrad_slabs = int(_os.environ.get("AST_WGRAD_SLABS", "0"))

# Convolution weight gradients deferred to the bank's end-of-backward flush (they are leaves of the backward graph): the data-
# gradient chain runs without them, and the banks' weight gradients run on the flush streams afterwards, longest first.
# Same box, three runs each: 6.24 / 6.25 / 6.20 ms per step inline against 5.97 / 6.11 / 6.08 deferred (profiles/r03/ab_defer.txt).
# On a side stream BESIDE the chain instead (AST_WGRAD_STREAM=1) the step takes 7.4-7.5 ms: a CU-filling kernel next to a chain of
# short dependent kernels delays every one of them (profiles/r03/ab_wstream.txt).  More than one stream per bank for the deferred
# launches loses as well (2: 6.01 / 6.21 / 6.13, 3: 6.35, 4: 6.43; profiles/r03/ab_defer_streams.txt).
wgrad_defer = _os.environ.get("AST_WGRAD_DEFER", "1") != "0"
wgrad_defer_streams = int(_os.environ.get("AST_WGRAD_DEFER_STREAMS", "1"))   # streams per bank that share its deferred launches
# Balance the deferred launches over the banks' flush streams (the smallest bank's stream takes over the larger banks' cheapest
# launches after its own flush; one-directional waits).  Opt-in: 6.06 / 6.07 / 6.02 ms without against 6.19-6.38 with it
# (profiles/r03/ab_defer_lend.txt; an earlier form with a barrier through the origin stream: 6.16 -> 6.20, ab_defer_pool.txt) --
# on paper the last bank runs alone for a quarter of the phase, in the replay every extra cross-stream wait costs more than it buys.
wgrad_defer_pool = _os.environ.get("AST_WGRAD_DEFER_POOL", "0") != "0"
wgrad_defer_lend = float(_os.environ.get("AST_WGRAD_DEFER_LEND", "1.0"))       # fraction of the helper's spare capacity that is used
