"""Drop-in for the reference's discriminator.py on libast_hip."""
import torch
import torch.nn as nn

from . import layers as L
from .style_encoder import _module_bank

NUM_INSTRUMENT_CLASSES = 2


class Discriminator(nn.Module):
    """discriminator.py:14-28: MLP 256 -> 128 -> 128 -> 2 logits (`net.0/2/4`)."""

    def __init__(self, input_dim: int = 256, hidden_dim: int = 128):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, NUM_INSTRUMENT_CLASSES))

    def register(self, bank):
        self._pw = [bank.add(self.net[i].weight, "linear", L.tok_dtype, bias=self.net[i].bias) for i in (0, 2, 4)]

    def forward(self, emb: torch.Tensor) -> torch.Tensor:
        bank = _module_bank(self)
        bank.prepare(self.training)
        shape = emb.shape
        h = L.linear(emb.reshape(-1, shape[-1]), self._pw[0], relu=True)
        h = L.linear(h, self._pw[1], relu=True)
        return L.linear(h, self._pw[2]).reshape(*shape[:-1], NUM_INSTRUMENT_CLASSES)
