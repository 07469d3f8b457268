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
