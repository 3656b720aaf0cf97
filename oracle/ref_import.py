"""Import the reference STA model from /root/reference (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE - never imported by the product path.  Used only by
`oracle/gen_golden.py` (fixture generation) and `oracle/check_oracle_vs_ref.py`.
The reference Python never travels to the GPU box; only the `.npz` fixtures do.

xformers is absent here and `vista_slam/sta_model/blocks/sta_blocks.py:22` imports it
unconditionally, so a math-equivalent stub of `memory_efficient_attention`
(= softmax(Q K^T * scale) V on (B,N,H,K) tensors, what `sta_blocks.py:143` computes)
is injected into `sys.modules` first.
"""
import sys
import types

REF_ROOT = "/root/reference"


def _install_xformers_stub():
    import torch  # noqa: F401
    if "xformers" in sys.modules:
        return
    xf = types.ModuleType("xformers")
    xo = types.ModuleType("xformers.ops")

    def memory_efficient_attention(q, k, v, scale=None, p=0.0, attn_bias=None):
        assert p == 0.0 and attn_bias is None
        q_, k_, v_ = (t.permute(0, 2, 1, 3) for t in (q, k, v))
        if scale is None:
            scale = q.shape[-1] ** -0.5
        a = (q_ @ k_.transpose(-1, -2)) * scale
        a = a.softmax(dim=-1)
        return (a @ v_).permute(0, 2, 1, 3)

    xo.memory_efficient_attention = memory_efficient_attention
    xf.ops = xo
    sys.modules["xformers"] = xf
    sys.modules["xformers.ops"] = xo


def load_reference_model(cfg, state):
    """Build the reference model for `cfg` (vista_slam_amd.weights.STAConfig) and load
    `state` (name -> numpy array) with strict=True."""
    import os
    import torch
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    _install_xformers_stub()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # `vista_slam_amd.install_as_reference()` (the zero-edit drop-in path, exercised by other tests of the same process) registers
    # STAND-IN modules under the reference's package name: drop every `vista_slam*` module that does not come from the reference tree
    for name in [n for n, m in list(sys.modules.items()) if n == "vista_slam" or n.startswith("vista_slam.")]:
        origin = getattr(sys.modules[name], "__file__", None) or ""
        if not origin.startswith(REF_ROOT):
            del sys.modules[name]
    from vista_slam.sta_model.sta_model import SymmetricTwoViewAssociation as STA
    model = STA(enc_embed_dim=cfg.enc_embed_dim, enc_depth=cfg.enc_depth,
                enc_num_heads=cfg.enc_num_heads, dec_embed_dim=cfg.dec_embed_dim,
                dec_depth=cfg.dec_depth, dec_num_heads=cfg.dec_num_heads)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model
