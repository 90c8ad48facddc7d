"""Positional encoding, mirroring the reference's models/embedder.py surface (get_embedder).

The HIP kernels compute the encoding in registers (iron_amd/csrc/mlp_core.h: head_fill); this
module only exists so that callers of `get_embedder` keep working.  It is a thin torch expression
(device-agnostic tensor glue, not a compute path of the renderer).
"""
from __future__ import annotations

import torch


class Embedder:
    """models/embedder.py:6-36, configured by the same keyword dictionary (include_input, input_dims, max_freq_log2,
    num_freqs, log_sampling, periodic_fns): cat[x, p(x f_0) for p in periodic_fns, ..., p(x f_{N-1})] with f_i spaced in log2
    (log_sampling) or linearly between 1 and 2^max_freq_log2."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        cfg = self.kwargs
        n_freqs, top = cfg["num_freqs"], cfg["max_freq_log2"]
        if cfg["log_sampling"]:
            self.freq_bands = 2.0 ** torch.linspace(0.0, top, n_freqs)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** top, n_freqs)
        self.periodic_fns = list(cfg["periodic_fns"])
        self.include_input = bool(cfg["include_input"])
        width = cfg["input_dims"]
        self.out_dim = (width if self.include_input else 0) + width * n_freqs * len(self.periodic_fns)

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        parts = [inputs] if self.include_input else []
        for freq in self.freq_bands:
            parts.extend(fn(inputs * freq) for fn in self.periodic_fns)
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    """models/embedder.py:39-54: returns (embed_fn, out_dim) of the NeRF encoding with `multires` octaves."""
    obj = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires, log_sampling=True,
                   periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=obj: eo.embed(x)), obj.out_dim
