"""`cuRoPE2D`-shaped drop-in over the HIP rope kernel (`sta_rope2d_inplace_dtype`).

The reference picks its rotary embedding with an import switch (pos_embed/pos_embed.py:106-108):

    try:
        from .curope import cuRoPE2D
        RoPE2D = cuRoPE2D
    except ImportError: ...   # slow pytorch version

`.curope` is the package pos_embed/curope/__init__.py, which exports `cuRoPE2D` from curope2d.py:32-40; that module
wraps the compiled extension's one function `rope_2d(tokens, positions, base, fwd)` (curope.cpp:49-65).  This file
mirrors those three names one for one, so either of these makes the reference use the gfx950 kernel unchanged:

  * point the switch at it:   `from vista_slam_amd.curope_compat import cuRoPE2D`   (one-line edit of pos_embed.py:107), or
  * leave pos_embed.py alone and install this module AS the extension the wrapper imports (curope2d.py:6-9):
        import sys, vista_slam_amd.curope_compat as cc; sys.modules["curope"] = cc
    (`import curope as _kernels` then resolves to `cc`, whose `rope_2d` has the extension's signature).

Same contract as the CUDA kernel (kernels.cu:84-108): tokens (B, N, H, D) with stride(3) == 1 and stride(2) == D (the
transposed view of the attention layer's (B, H, N, D) q / k), positions (B, N, 2) int64 contiguous, fp16 / fp32 / fp64
tokens, rotation evaluated in fp32, IN PLACE; `fwd = -F0` is the backward pass (curope2d.py:24-29).
"""
import torch

from .sta_frontend import rope2d_inplace


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """curope.rope_2d (curope.cpp:49-65).  GPU tensors only: this library has no CPU path (the reference's rope_2d_cpu,
    curope.cpp:21-47, is restated only in the test infrastructure, as the checker)."""
    if not tokens.is_cuda:
        raise RuntimeError("vista_slam_amd.curope_compat.rope_2d: CPU tensors are not served (no CPU path in this library)")
    rope2d_inplace(tokens, positions, float(base), float(fwd))


class cuRoPE2D_func(torch.autograd.Function):
    """curope2d.py:12-29: in-place rotation forward, inverse rotation of the incoming gradient backward."""

    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base = base
        ctx.saved_F0 = F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        rope_2d(grad_res, positions, base, -F0)
        ctx.mark_dirty(grad_res)
        return grad_res, None, None, None


class cuRoPE2D(torch.nn.Module):
    """curope2d.py:32-40: `forward(tokens (B,H,N,D), positions (B,N,2))` rotates `tokens` in place and returns it."""

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
