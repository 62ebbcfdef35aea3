"""Parameter containers with the reference's names for the EPA transformer
block (reference ``models/modules/transformer.py:8-74,130-201``).

These modules hold the (fp32, trainable-shaped) parameters so checkpoints with
the reference key layout load unchanged
(``cp_blocks_*.transformer.{attn1.to_q,...}``, ``pe.freq_bands``); the arithmetic
runs in ``engine.run_epa`` on the HIP kernels, not in ``forward`` methods here.
"""
import torch
import torch.nn as nn


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        inner = int(dim * mult)
        last = nn.Linear(inner, dim)
        nn.init.zeros_(last.weight)          # reference transformer.py:29-30
        nn.init.zeros_(last.bias)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(0.0), last)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Linear(inner, query_dim)
        nn.init.zeros_(self.to_out.weight)   # reference transformer.py:54-55
        nn.init.zeros_(self.to_out.bias)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(dim, context_dim or dim, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)


class SphericalPE(nn.Module):
    """Frequency bands of reference transformer.py:165-183 (logscale)."""

    def __init__(self, N_freqs):
        super().__init__()
        self.N_freqs = N_freqs
        base = 2 if N_freqs <= 80 else 5000 ** (1 / (N_freqs / 2.5))
        self.register_buffer("freq_bands", base ** torch.linspace(0, N_freqs - 1, N_freqs))
