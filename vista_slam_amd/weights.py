"""Procedural STA weights + state_dict schema.

The reference ships no checkpoint (`pretrains/README.md:1-4`), so parity fixtures, the GPU
tests and the bench all use weights produced by this deterministic, platform-independent
counter-hash generator.  The schema (names, shapes, order) mirrors the reference
`SymmetricTwoViewAssociation.state_dict()` (`vista_slam/sta_model/sta_model.py:33-75`,
`heads/dpt_block.py:264-410`, `heads/pose_head.py:7-36`) so the generated dict loads into
the reference with `strict=True` (done in `oracle/gen_golden.py`, this container only).

Pure numpy; no torch needed.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, asdict
from typing import Dict, Iterator, List, Tuple

import numpy as np


@dataclass(frozen=True)
class STAConfig:
    """Constructor defaults of the reference model (`sta_model.py:33-52`)."""
    patch_size: int = 16
    enc_embed_dim: int = 1024
    enc_depth: int = 24
    enc_num_heads: int = 16
    dec_embed_dim: int = 768
    dec_depth: int = 12
    dec_num_heads: int = 12
    mlp_ratio: int = 4
    rope_base: float = 100.0
    ln_eps: float = 1e-6
    # DPT head constants (`heads/dpt_head.py:98-117`, `dpt_block.py:281-283`)
    dpt_feature_dim: int = 256
    dpt_last_dim: int = 128
    dpt_layer_dims: Tuple[int, int, int, int] = (96, 192, 384, 768)
    pose_hidden: int = 512

    @property
    def hooks(self) -> Tuple[int, int, int, int]:
        l2 = self.dec_depth
        return (0, l2 * 2 // 4 + 1, l2 * 3 // 4 + 1, l2 + 1)   # dpt_head.py:112

    def as_dict(self):
        return asdict(self)


FULL = STAConfig()
# Smallest config that still exercises every code path (head_dim must stay 64; dec_depth>9
# is asserted by the reference, dpt_head.py:102).
TINY = STAConfig(enc_embed_dim=128, enc_depth=2, enc_num_heads=2,
                 dec_embed_dim=128, dec_depth=10, dec_num_heads=2)


def schema(cfg: STAConfig) -> List[Tuple[str, Tuple[int, ...], str, int]]:
    """Ordered (name, shape, kind, fan_in) list == reference state_dict order.

    kind in {'w','b','ln_w','ln_b','tok'}; aliased keys (`scratch.layerK_rn` ==
    `scratch.layer_rn.{K-1}`, dpt_block.py:70-75) appear twice with identical values.
    """
    E, D = cfg.enc_embed_dim, cfg.dec_embed_dim
    P = cfg.patch_size
    out: List[Tuple[str, Tuple[int, ...], str, int]] = []

    def lin(name, o, i):
        out.append((f"{name}.weight", (o, i), "w", i))
        out.append((f"{name}.bias", (o,), "b", i))

    def ln(name, c):
        out.append((f"{name}.weight", (c,), "ln_w", c))
        out.append((f"{name}.bias", (c,), "ln_b", c))

    def conv(name, o, i, k, bias=True, transposed=False):
        shape = (i, o, k, k) if transposed else (o, i, k, k)
        fan = i * k * k if not transposed else i  # effective contraction per output
        out.append((f"{name}.weight", shape, "w", fan))
        if bias:
            out.append((f"{name}.bias", (o,), "b", fan))

    out.append(("init_pose_token", (1, 1, D), "tok", D))
    conv("patch_embed.proj", E, 3, P)
    for i in range(cfg.enc_depth):
        p = f"enc_blocks.{i}"
        ln(f"{p}.norm1", E)
        lin(f"{p}.attn.qkv", 3 * E, E)
        lin(f"{p}.attn.proj", E, E)
        ln(f"{p}.norm2", E)
        lin(f"{p}.mlp.fc1", cfg.mlp_ratio * E, E)
        lin(f"{p}.mlp.fc2", E, cfg.mlp_ratio * E)
    ln("enc_norm", E)
    lin("decoder_embed", D, E)
    for i in range(cfg.dec_depth):
        p = f"dec_block.{i}"
        ln(f"{p}.norm1", D)
        lin(f"{p}.attn.qkv", 3 * D, D)
        lin(f"{p}.attn.proj", D, D)
        for n in ("projq", "projk", "projv", "proj"):
            lin(f"{p}.cross_attn.{n}", D, D)
        ln(f"{p}.norm2", D)
        ln(f"{p}.norm3", D)
        lin(f"{p}.mlp.fc1", cfg.mlp_ratio * D, D)
        lin(f"{p}.mlp.fc2", D, cfg.mlp_ratio * D)
        ln(f"{p}.norm_y", D)
    ln("dec_norm", D)
    F = cfg.dpt_feature_dim
    dp = "downstream_head_pts.dpt"
    for k, c in enumerate(cfg.dpt_layer_dims):
        conv(f"{dp}.scratch.layer{k + 1}_rn", F, c, 3, bias=False)
    for k, c in enumerate(cfg.dpt_layer_dims):
        conv(f"{dp}.scratch.layer_rn.{k}", F, c, 3, bias=False)
    for r in (1, 2, 3, 4):
        conv(f"{dp}.scratch.refinenet{r}.out_conv", F, F, 1)
        for u in (1, 2):
            conv(f"{dp}.scratch.refinenet{r}.resConfUnit{u}.conv1", F, F, 3)
            conv(f"{dp}.scratch.refinenet{r}.resConfUnit{u}.conv2", F, F, 3)
    L = cfg.dpt_last_dim
    conv(f"{dp}.head.0", F // 2, F, 3)
    conv(f"{dp}.head.2", L, F // 2, 3)
    conv(f"{dp}.head.4", 4, L, 1)
    l0, l1, l2, l3 = cfg.dpt_layer_dims
    conv(f"{dp}.act_postprocess.0.0", l0, E, 1)
    conv(f"{dp}.act_postprocess.0.1", l0, l0, 4, transposed=True)
    conv(f"{dp}.act_postprocess.1.0", l1, D, 1)
    conv(f"{dp}.act_postprocess.1.1", l1, l1, 2, transposed=True)
    conv(f"{dp}.act_postprocess.2.0", l2, D, 1)
    conv(f"{dp}.act_postprocess.3.0", l3, D, 1)
    conv(f"{dp}.act_postprocess.3.1", l3, l3, 3)
    Hd = cfg.pose_hidden
    lin("head_pose_s.mlp.0", Hd, D)
    lin("head_pose_s.mlp.2", Hd, Hd)
    lin("head_pose_s.mlp.4", Hd, Hd)
    lin("head_pose_s.fc_t", 3, Hd)
    lin("head_pose_s.fc_conf.0", 1, Hd)
    lin("head_pose_s.fc_rot", 9, Hd)
    return out


def _alias_of(name: str) -> str:
    """`scratch.layer_rn.{k}` shares storage with `scratch.layer{k+1}_rn`."""
    marker = ".scratch.layer_rn."
    if marker in name:
        head, tail = name.split(marker)
        k, rest = tail.split(".", 1)
        return f"{head}.scratch.layer{int(k) + 1}_rn.{rest}"
    return name


def hash_uniform(seed: int, n: int, offset: int = 0, out: np.ndarray = None) -> np.ndarray:
    """n floats in [-1, 1): murmur3 finaliser over a 32-bit counter (exact on any platform).

    `out` (float32, >= n elements) lets callers reuse one scratch buffer: first-touch page
    faults dominate generation time on sandboxed hosts."""
    out = np.empty(n, np.float32) if out is None else out[:n]
    CH = 1 << 18                      # cache-sized chunks: ~10x faster than one big pass
    base = np.arange(CH, dtype=np.uint32)
    t = np.empty(CH, np.uint32)
    s32 = np.uint32(seed & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for lo in range(0, n, CH):
            m = min(CH, n - lo)
            x = base[:m] + np.uint32((offset + lo) & 0xFFFFFFFF)   # wraps mod 2^32
            x *= np.uint32(0x9E3779B1)
            x += s32
            tt = t[:m]
            np.right_shift(x, np.uint32(16), out=tt); x ^= tt
            x *= np.uint32(0x85EBCA6B)
            np.right_shift(x, np.uint32(13), out=tt); x ^= tt
            x *= np.uint32(0xC2B2AE35)
            np.right_shift(x, np.uint32(16), out=tt); x ^= tt
            x >>= np.uint32(8)
            o = out[lo:lo + m]
            o[:] = x                                  # exact: < 2^24
            o *= np.float32(2.0 / (1 << 24))
            o -= np.float32(1.0)
    return out


def _name_seed(seed: int, name: str) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


OUTLIER_DPT_SCALE = {1: 300.0, 2: 3.0e4}     # outlier level -> scale of the DPT head's feature maps (undone by head.4)
OUTLIER_MASSIVE = (80.0, -60.0, 50.0)         # "massive activation" channels added to the residual streams


def _outlier(cfg: STAConfig, seed: int, level: int, src: str, kind: str, a: np.ndarray) -> None:
    """Trained-checkpoint statistics the plain U(+-1/sqrt(fan_in)) weights do not have, in place:
    * LayerNorm gains with a heavy tail: ~3 % of the channels of every norm scaled by 3..10;
    * massive activations: three channels of the encoder / decoder residual streams carry +80 / -60 / +50 from the second
      block on (bias of mlp.fc2), i.e. 50-100x the other channels - every later LayerNorm, GEMM operand split and the DPT
      hooks see them;
    * DPT head (no normalisation layers, dpt_block.py:264-324, use_bn=False): the four act_postprocess 1x1 convolutions are
      scaled by OUTLIER_DPT_SCALE[level] and head.4 by its inverse, so every feature map of the head is 300x larger
      (level 1: up to 6.5e3 - past the +-448 an e4m3 correction byte could carry, which is why the f16mx arithmetic keeps
      its ACTIVATION bytes in e5m2) or 3e4x larger (level 2: past the +-65504 of the fp16 planes) while the fp32 reference's outputs keep their scale."""
    if kind == "ln_w":
        n = a.size
        u = hash_uniform(_name_seed(seed, src + "/outlier"), 2 * n)
        pick = u[:n] > np.float32(0.94)                 # u in [-1, 1): 3 % of the channels
        a[pick] *= (np.float32(6.5) + np.float32(3.5) * u[n:][pick])
        return
    E, D = cfg.enc_embed_dim, cfg.dec_embed_dim
    for blk, C in ((f"enc_blocks.{min(1, cfg.enc_depth - 1)}.mlp.fc2.bias", E), (f"dec_block.{min(1, cfg.dec_depth - 1)}.mlp.fc2.bias", D)):
        if src == blk:
            for ch, v in zip((7 % C, C // 3, (2 * C) // 3 + 5), OUTLIER_MASSIVE):
                a[ch] += np.float32(v)
            return
    dp = "downstream_head_pts.dpt."
    if src.startswith(dp + "act_postprocess.") and src.split(".")[-2] == "0" and src.split(".")[-3] in "0123":
        a *= np.float32(OUTLIER_DPT_SCALE[level])
    elif src == dp + "head.4.weight":
        a *= np.float32(1.0 / OUTLIER_DPT_SCALE[level])


def generate(cfg: STAConfig = FULL, seed: int = 43, qk_gain: float = 1.0,
             reuse_buffer: bool = False, outlier: int = 0) -> Iterator[Tuple[str, np.ndarray]]:
    """Yield (name, float32 array) in reference state_dict order.

    reuse_buffer=True yields views of ONE scratch buffer (valid only until the next
    iteration) - the streaming path used by `STAFrontend.load_procedural`.

    qk_gain > 1 multiplies every Q/K projection (weights and biases) so attention becomes
    peaky ("sharp" set): with default-scale weights a wrong RoPE/softmax
    hides under the 1e-3 bar.

    outlier > 0 adds trained-checkpoint-like range statistics (`_outlier`): 1 = heavy-tailed LayerNorm gains, massive
    activation channels, DPT feature maps up to 6.5e3; 2 = DPT feature maps past the fp16 range as well.
    """
    E, D = cfg.enc_embed_dim, cfg.dec_embed_dim
    sch = schema(cfg)
    scratch = np.empty(max(int(np.prod(s[1])) for s in sch), np.float32) if reuse_buffer else None
    for name, shape, kind, fan in sch:
        src = _alias_of(name)
        n = int(np.prod(shape))
        a = hash_uniform(_name_seed(seed, src), n, out=scratch)
        if kind == "w":
            a *= np.float32(1.0 / np.sqrt(fan))
        elif kind == "b":
            a *= np.float32(0.5 / np.sqrt(fan))
        elif kind == "ln_w":
            a *= np.float32(0.2)
            a += np.float32(1.0)
        elif kind == "ln_b":
            a *= np.float32(0.1)
        else:  # pose token
            a *= np.float32(0.035)
        a = a.reshape(shape)
        if qk_gain != 1.0:
            g = np.float32(qk_gain)
            if src.endswith("attn.qkv.weight") or src.endswith("attn.qkv.bias"):
                C = E if src.startswith("enc_blocks") else D
                a[: 2 * C] *= g
            elif ".cross_attn.projq." in src or ".cross_attn.projk." in src:
                a *= g
        if outlier:
            _outlier(cfg, seed, outlier, src, kind, a.reshape(-1) if kind in ("ln_w", "b") else a)
        yield name, a


def state_dict(cfg: STAConfig = FULL, seed: int = 43, qk_gain: float = 1.0, outlier: int = 0) -> Dict[str, np.ndarray]:
    return dict(generate(cfg, seed, qk_gain, outlier=outlier))


def state_dict_fingerprint(sd) -> str:
    """sha256 over the (name, shape, fp32 bytes) of every tensor of a state_dict in sorted key order - numpy arrays or torch
    tensors.  The real-checkpoint fixtures (oracle/gen_golden.py --checkpoint) carry the fingerprint of the weights they were
    generated with; the GPU test compares it with the checkpoint file it is handed, so fixtures and file cannot be mixed up."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k]
        if hasattr(v, "detach"):
            v = v.detach().to("cpu").float().numpy()
        a = np.ascontiguousarray(v, dtype=np.float32)
        h.update(k.encode()); h.update(str(tuple(a.shape)).encode()); h.update(a.tobytes())
    return h.hexdigest()


def synth_images(n: int, H: int, W: int, seed: int = 43, tag: int = 0) -> np.ndarray:
    """n synthetic RGB images, uint8 uniform[0,255] -> normalised (x/255-0.5)/0.5, NCHW fp32.

    Same distribution as the reference `ImgNorm` (`vista_slam/utils/image.py:13`);
    seed 43 == reference `random_seed` (`configs/default.yaml:20`).
    """
    cnt = n * 3 * H * W
    u = hash_uniform(_name_seed(seed, f"image/{tag}"), cnt)
    u8 = np.floor((u + np.float32(1.0)) * np.float32(128.0)).clip(0, 255)
    img = (u8 / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return img.reshape(n, 3, H, W).astype(np.float32)


def synth_images_u8(n: int, H: int, W: int, seed: int = 43, tag: int = 0) -> np.ndarray:
    """The uint8 HWC frames [n,H,W,3] whose ImgNorm is exactly `synth_images(n, H, W, seed, tag)`."""
    cnt = n * 3 * H * W
    u = hash_uniform(_name_seed(seed, f"image/{tag}"), cnt)
    u8 = np.floor((u + np.float32(1.0)) * np.float32(128.0)).clip(0, 255).astype(np.uint8)
    return np.ascontiguousarray(u8.reshape(n, 3, H, W).transpose(0, 2, 3, 1))


def synth_frames_u8(H: int, W: int, seed: int = 43, tag: int = 0) -> np.ndarray:
    """A camera-like uint8 HWC frame [H,W,3] for the input-step tests (f3): integer-only arithmetic (triangle waves of
    several periods + sharp edges + hash noise), so the bytes are identical on every platform."""
    y, x = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")

    def tri(v, p):
        return np.abs(v % (2 * p) - p) * 255 // p
    noise = synth_images_u8(1, H, W, seed, tag=1000 + tag)[0].astype(np.int64)
    out = np.empty((H, W, 3), np.int64)
    for c in range(3):
        wave = tri(x * (c + 2) + y, 191) * 2 + tri(y * (3 - c) + x // 2, 113)
        edge = (((x // 37) + (y // 29) + c) % 2) * 255                     # checkerboard: ringing-prone edges
        out[..., c] = (wave + edge + noise[..., c]) // 5
    return np.clip(out, 0, 255).astype(np.uint8)


def smooth_images(n: int, H: int, W: int, seed: int = 43, tag: int = 0) -> np.ndarray:
    """Low-frequency synthetic images (sums of a few sinusoids) - a second, structured input
    distribution for parity tests (white noise excites every patch identically)."""
    rng = hash_uniform(_name_seed(seed, f"smooth/{tag}"), n * 3 * 6 * 4).reshape(n, 3, 6, 4)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32) / H, np.arange(W, dtype=np.float32) / W,
                         indexing="ij")
    img = np.zeros((n, 3, H, W), np.float32)
    for k in range(6):
        fy = (rng[:, :, k, 0] * 6.0)[:, :, None, None]
        fx = (rng[:, :, k, 1] * 6.0)[:, :, None, None]
        ph = (rng[:, :, k, 2] * np.pi)[:, :, None, None]
        am = (rng[:, :, k, 3] * 0.4)[:, :, None, None]
        img += am * np.sin(2 * np.pi * (fy * yy + fx * xx) + ph)
    return np.clip(img, -1.0, 1.0).astype(np.float32)
