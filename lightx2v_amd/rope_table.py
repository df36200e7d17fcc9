"""3-axis RoPE table for Wan latent grids, in the real (cos || sin) layout the HIP kernels read.

Restates lightx2v/models/schedulers/wan/scheduler.py:31-48,77-114: head_dim/2 complex slots are
split temporal | height | width as (c - 2(c//3), c//3, c//3); axis `a` with `n_a` slots uses
theta = pos_a * 10000^(-2i/(2 n_a)), i = 0..n_a-1; tokens are ordered f-major, then h, then w; the
same row serves every head.  Sequence-parallel ranks take contiguous chunks of the zero-padded
table (zero cos AND sin, so padded tokens rotate to exactly 0, scheduler.py:96-113).
"""

from __future__ import annotations

from typing import Tuple

import torch


def axis_slots(head_dim: int) -> Tuple[int, int, int]:
    c = head_dim // 2
    return c - 2 * (c // 3), c // 3, c // 3


def _axis_angles(n_pos: int, n_slots: int) -> torch.Tensor:
    """fp32 [n_pos, n_slots] angles, evaluated in fp32 like the reference's `rope_params`."""
    dim = 2 * n_slots
    inv = 1.0 / torch.pow(torch.tensor(10000.0, dtype=torch.float32), torch.arange(0, dim, 2, dtype=torch.float32) / dim)
    return torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)


def build_rope_table(grid: Tuple[int, int, int], head_dim: int = 128, device="cpu") -> torch.Tensor:
    """fp32 [f*h*w, head_dim]: columns [0, head_dim/2) = cos, [head_dim/2, head_dim) = sin."""
    f, h, w = grid
    nt, nh, nw = axis_slots(head_dim)
    at = _axis_angles(f, nt).view(f, 1, 1, nt).expand(f, h, w, nt)
    ah = _axis_angles(h, nh).view(1, h, 1, nh).expand(f, h, w, nh)
    aw = _axis_angles(w, nw).view(1, 1, w, nw).expand(f, h, w, nw)
    ang = torch.cat([at, ah, aw], dim=-1).reshape(f * h * w, head_dim // 2)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous().to(device)


def padded_len(seq_len: int, world_size: int) -> int:
    return seq_len + (world_size - seq_len % world_size) % world_size


def shard_rope_table(table: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    """Zero-pad rows to a multiple of world_size and return this rank's contiguous chunk."""
    s = table.shape[0]
    total = padded_len(s, world_size)
    if total != s:
        table = torch.cat([table, table.new_zeros(total - s, table.shape[1])], dim=0)
    per = total // world_size
    return table[rank * per:(rank + 1) * per].contiguous()


def build_hunyuan_rope_table(grid: Tuple[int, int, int], head_dim: int = 128, rope_dim_list=(16, 56, 56), theta: float = 256.0, device="cpu") -> torch.Tensor:
    """HunyuanVideo-1.5's table in the same [tokens, head_dim] = cos || sin layout
    (lightx2v/models/schedulers/hunyuan_video/scheduler.py:123-142 over
    posemb_layers.py:169-283): axis `a` owns rope_dim_list[a]/2 slots with angle
    pos_a * theta^(-2j / rope_dim_list[a]); tokens ordered t-major, then h, then w."""
    if sum(rope_dim_list) != head_dim:
        raise ValueError(f"sum(rope_dim_list)={sum(rope_dim_list)} must equal head_dim={head_dim}")
    parts = []
    for a, (n, dim) in enumerate(zip(grid, rope_dim_list)):
        inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        shape = [1, 1, 1, dim // 2]
        shape[a] = n
        ang = torch.outer(torch.arange(n, dtype=torch.float32), inv).view(shape)
        parts.append(ang.expand(grid[0], grid[1], grid[2], dim // 2))
    ang = torch.cat(parts, dim=-1).reshape(grid[0] * grid[1] * grid[2], head_dim // 2)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous().to(device)
