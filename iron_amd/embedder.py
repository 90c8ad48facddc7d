"""Positional encoding, mirroring the reference's models/embedder.py surface (get_embedder).

The HIP kernels compute the encoding in registers (iron_amd/csrc/mlp_core.h: head_fill); this
module only exists so that callers of `get_embedder` keep working.  It is a thin torch expression
(device-agnostic tensor glue, not a compute path of the renderer).
"""
from __future__ import annotations

import torch


class Embedder:
    """models/embedder.py:6-36: cat[x, sin(x*2^0), cos(x*2^0), ..., sin(x*2^(L-1)), cos(x*2^(L-1))]."""

    def __init__(self, input_dims: int, num_freqs: int, include_input: bool = True):
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.include_input = include_input
        self.freq_bands = 2.0 ** torch.linspace(0.0, float(num_freqs - 1), num_freqs)
        self.out_dim = (input_dims if include_input else 0) + 2 * input_dims * num_freqs

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        parts = [inputs] if self.include_input else []
        for freq in self.freq_bands:
            parts.append(torch.sin(inputs * freq))
            parts.append(torch.cos(inputs * freq))
        return torch.cat(parts, -1)


def get_embedder(multires: int, input_dims: int = 3):
    """models/embedder.py:39-54: returns (embed_fn, out_dim)."""
    obj = Embedder(input_dims=input_dims, num_freqs=multires)
    return (lambda x, eo=obj: eo.embed(x)), obj.out_dim
