"""Inference-only stand-in for the reference's `curope` package over the HIP rope kernel (`sta_rope2d_inplace_dtype`).

The reference selects its rotary embedding at import time (pos_embed/pos_embed.py:106-108): if `from .curope import cuRoPE2D`
succeeds, `RoPE2D = cuRoPE2D`, otherwise a slow PyTorch class.  Two names are interface there:

  * the module class `cuRoPE2D(freq, F0)` with `forward(tokens (B,H,N,D), positions (B,N,2)) -> tokens`, rotating IN PLACE
    (the attention layers hand it strided views of the qkv tensor, sta_blocks.py:132-137), and
  * the compiled extension's one function `rope_2d(tokens (B,N,H,D), positions, base, fwd)` (curope.cpp:49-65).

Both are provided here, so either of these makes the reference's own PyTorch layers run the gfx950 kernel:

    from vista_slam_amd.curope_compat import cuRoPE2D                    # one-line edit of pos_embed.py:107
    import sys, vista_slam_amd.curope_compat as cc; sys.modules["curope"] = cc      # or no edit: be the extension (curope2d.py:6-9)

Kernel contract (kernels.cu:84-108): tokens with stride(3) == 1 and stride(2) == D, int64 contiguous positions, fp16 / fp32 /
fp64 storage, rotation evaluated in fp32.  This frontend is inference-only (training is out of scope): there
is no autograd wrapper; calling the module on tensors that require grad raises instead of silently dropping the gradient.
"""
import torch

from .sta_frontend import rope2d_inplace


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """Signature of the extension function (curope.cpp:49-65); `fwd < 0` applies the inverse rotation.  GPU tensors only:
    the library has no CPU path (the reference's CPU loop, curope.cpp:21-47, is restated only in the test oracle)."""
    if not tokens.is_cuda:
        raise RuntimeError("vista_slam_amd.curope_compat.rope_2d: CPU tensors are not served (no CPU path in this library)")
    rope2d_inplace(tokens, positions, float(base), float(fwd))


class cuRoPE2D(torch.nn.Module):
    """`RoPE2D` drop-in for inference: rotates the (B,H,N,D) view it is given in place and returns that same tensor."""

    def __init__(self, freq: float = 100.0, F0: float = 1.0):
        super().__init__()
        self.base, self.F0 = float(freq), float(F0)

    @torch.no_grad()
    def forward(self, tokens: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        if tokens.requires_grad:
            raise RuntimeError("vista_slam_amd.curope_compat.cuRoPE2D is inference-only (no backward pass); run under torch.no_grad()")
        # the kernel walks (B, N, H, D); the layer's (B, H, N, D) view transposed is exactly that, with no copy
        rope_2d(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
