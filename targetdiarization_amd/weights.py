"""Weight catalogue, deterministic recipe weights and the TDXW blob format.

* `mossformer2_param_shapes` restates the state_dict layout of the reference's
  MossFormer2 (reference: look2hear/models/mossformer2.py:532-561 and the sub-modules it
  builds; key list verified against the imported reference by oracle/make_goldens.py).
* `recipe_state_dict` fills that layout from a counter-based PRNG (Philox keyed by the
  tensor *name*), so the 223 MB of fp32 weights never have to be committed or shipped:
  the oracle, the tests and bench.py all regenerate bit-identical weights.
  No trained checkpoint ships with the reference (checkpoints/ holds only .gitkeep).
* `pack_blob` serialises {name: tensor} into the flat "TDXW" container that the C-ABI
  (`tdx_mf2_create`, include/tdx.h) parses by tensor name.  A real checkpoint
  (base_model.py:52-64 format: {"model_name", "state_dict"}) goes through the same packer.
"""
from __future__ import annotations

import struct
import zlib
from collections import OrderedDict

import numpy as np
import torch

_PFX = "mask_net.mdl.intra_mdl.mossformerM."


def mossformer2_param_shapes(num_blocks: int = 24, d: int = 512, kernel_size: int = 16,
                             num_spks: int = 2, qk_dim: int = 128, expansion: int = 4,
                             inner: int = 256, rot_dim: int = 32) -> "OrderedDict[str, tuple]":
    """Ordered {state_dict key: shape}; same key order as the reference's nn.Module."""
    hid = d * expansion
    s = OrderedDict()
    s["enc.conv1d.weight"] = (d, 1, kernel_size)
    s["mask_net.norm.weight"] = (d,)
    s["mask_net.norm.bias"] = (d,)
    s["mask_net.conv1d_encoder.weight"] = (d, d, 1)
    s["mask_net.pos_enc.scale"] = (1,)
    s["mask_net.pos_enc.inv_freq"] = (d // 2,)
    for l in range(num_blocks):
        p = f"{_PFX}fsmn.{l}."
        s[p + "conv1.0.weight"] = (inner, d, 1)
        s[p + "conv1.0.bias"] = (inner,)
        s[p + "conv1.1.weight"] = (1,)
        s[p + "norm1.weight"] = (inner,)
        s[p + "norm1.bias"] = (inner,)
        for br in ("to_u", "to_v"):
            q = p + f"gated_fsmn.{br}.mdl."
            s[q + "0.weight"] = (inner,)
            s[q + "0.bias"] = (inner,)
            s[q + "1.weight"] = (inner, inner)
            s[q + "1.bias"] = (inner,)
            s[q + "3.sequential.1.conv.weight"] = (inner, 1, 17)
        q = p + "gated_fsmn.fsmn."
        s[q + "linear.weight"] = (inner, inner)
        s[q + "linear.bias"] = (inner,)
        s[q + "project.weight"] = (inner, inner)
        s[q + "conv.conv1.weight"] = (inner, 1, 39, 1)
        s[q + "conv.norm1.weight"] = (inner,)
        s[q + "conv.norm1.bias"] = (inner,)
        s[q + "conv.prelu1.weight"] = (inner,)
        s[q + "conv.conv2.weight"] = (inner, 2, 39, 1)
        s[q + "conv.norm2.weight"] = (inner,)
        s[q + "conv.norm2.bias"] = (inner,)
        s[q + "conv.prelu2.weight"] = (inner,)
        s[p + "norm2.weight"] = (inner,)
        s[p + "norm2.bias"] = (inner,)
        s[p + "conv2.weight"] = (d, inner, 1)
        s[p + "conv2.bias"] = (d,)
    for l in range(num_blocks):
        p = f"{_PFX}layers.{l}."
        s[p + "rotary_pos_emb.freqs"] = (rot_dim // 2,)
        for nm, (i, o) in (("to_hidden", (d, hid)), ("to_qk", (d, qk_dim))):
            s[p + nm + ".mdl.0.g"] = (1,)
            s[p + nm + ".mdl.1.weight"] = (o, i)
            s[p + nm + ".mdl.1.bias"] = (o,)
            s[p + nm + ".mdl.3.sequential.1.conv.weight"] = (o, 1, 17)
        s[p + "qk_offset_scale.gamma"] = (4, qk_dim)
        s[p + "qk_offset_scale.beta"] = (4, qk_dim)
        s[p + "to_out.mdl.0.g"] = (1,)
        s[p + "to_out.mdl.1.weight"] = (d, 2 * d)
        s[p + "to_out.mdl.1.bias"] = (d,)
        s[p + "to_out.mdl.3.sequential.1.conv.weight"] = (d, 1, 17)
    s["mask_net.mdl.intra_mdl.norm.weight"] = (d,)
    s["mask_net.mdl.intra_mdl.norm.bias"] = (d,)
    s["mask_net.mdl.intra_norm.weight"] = (d,)
    s["mask_net.mdl.intra_norm.bias"] = (d,)
    s["mask_net.conv1d_out.weight"] = (d * num_spks, d, 1)
    s["mask_net.conv1d_out.bias"] = (d * num_spks,)
    s["mask_net.conv1_decoder.weight"] = (d, d, 1)
    s["mask_net.prelu.weight"] = (1,)
    s["mask_net.output.0.weight"] = (d, d, 1)
    s["mask_net.output.0.bias"] = (d,)
    s["mask_net.output_gate.0.weight"] = (d, d, 1)
    s["mask_net.output_gate.0.bias"] = (d,)
    s["dec.weight"] = (d, 1, kernel_size)
    return s


def philox_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [-1, 1): top 24 bits of Philox4x64 raw words, keyed by
    (seed, crc32(name)).  `random_raw` is the documented stable bit stream."""
    key = np.array([seed & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())], dtype=np.uint64)
    raw = np.random.Philox(key=key).random_raw(n)
    f = (raw >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (f * np.float32(2.0) - np.float32(1.0)).astype(np.float32)


def _recipe_tensor(name: str, shape: tuple, seed: int) -> torch.Tensor:
    n = int(np.prod(shape))
    if name.endswith("pos_enc.inv_freq"):
        dim = shape[0] * 2
        return 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    if name.endswith("rotary_pos_emb.freqs"):
        dim = shape[0] * 2
        return 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    u = torch.from_numpy(philox_uniform(name, n, seed)).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = any(t in name for t in (".norm.", ".norm1.", ".norm2.", "intra_norm.", ".mdl.0.weight", ".mdl.0.bias"))
    if name.endswith(".g") or name.endswith("pos_enc.scale"):
        return 1.0 + 0.2 * u
    if "prelu" in name or name.endswith("conv1.1.weight"):
        return 0.25 + 0.1 * u
    if name.endswith("qk_offset_scale.gamma"):
        return 2.5 + 1.0 * u      # large enough that the relu^2 quadratic branch matters
    if name.endswith("qk_offset_scale.beta"):
        return 0.3 * u
    if is_norm:
        return (1.0 + 0.2 * u) if leaf == "weight" else (0.1 * u)
    if leaf == "bias":
        return 0.1 * u
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return u * float(1.0 / np.sqrt(fan_in))


def recipe_state_dict(seed: int = 0, num_blocks: int = 24, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 weights for the MossFormer2 layout (see module docstring)."""
    out = OrderedDict()
    for name, shape in mossformer2_param_shapes(num_blocks=num_blocks, **kw).items():
        # the reference shares ONE RotaryEmbedding across layers -> identical tensors
        out[name] = _recipe_tensor(name, shape, seed).to(torch.float32).contiguous()
    return out


def paraformer_encoder_param_shapes(num_blocks: int = 50, d: int = 512, d_in: int = 560, ffn: int = 2048,
                                    ksize: int = 11) -> "OrderedDict[str, tuple]":
    """funasr SANMEncoder state_dict layout [upstream-recall, SURVEY Appendix B.4]: one
    `encoders0` layer (560 -> 512) + num_blocks-1 `encoders` layers + after_norm."""
    s = OrderedDict()

    def layer(p, din):
        s[p + "self_attn.linear_out.weight"] = (d, d)
        s[p + "self_attn.linear_out.bias"] = (d,)
        s[p + "self_attn.linear_q_k_v.weight"] = (3 * d, din)
        s[p + "self_attn.linear_q_k_v.bias"] = (3 * d,)
        s[p + "self_attn.fsmn_block.weight"] = (d, 1, ksize)
        s[p + "feed_forward.w_1.weight"] = (ffn, d)
        s[p + "feed_forward.w_1.bias"] = (ffn,)
        s[p + "feed_forward.w_2.weight"] = (d, ffn)
        s[p + "feed_forward.w_2.bias"] = (d,)
        s[p + "norm1.weight"] = (din,)
        s[p + "norm1.bias"] = (din,)
        s[p + "norm2.weight"] = (d,)
        s[p + "norm2.bias"] = (d,)

    layer("encoder.encoders0.0.", d_in)
    for i in range(num_blocks - 1):
        layer(f"encoder.encoders.{i}.", d)
    s["encoder.after_norm.weight"] = (d,)
    s["encoder.after_norm.bias"] = (d,)
    return s


def recipe_paraformer_state_dict(seed: int = 0, num_blocks: int = 50) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for name, shape in paraformer_encoder_param_shapes(num_blocks).items():
        n = int(np.prod(shape))
        u = torch.from_numpy(philox_uniform("pf:" + name, n, seed)).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if ".norm" in name or "after_norm" in name:
            t = (1.0 + 0.2 * u) if leaf == "weight" else 0.1 * u
        elif leaf == "bias":
            t = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            t = u * float(1.0 / np.sqrt(fan_in))
        out[name] = t.to(torch.float32).contiguous()
    return out


def paraformer_decoder_param_shapes(num_blocks: int = 16, d: int = 512, ffn: int = 2048, ksize: int = 11,
                                    vocab: int = 8404) -> "OrderedDict[str, tuple]":
    """funasr Paraformer (CifPredictorV2 + ParaformerSANMDecoder) state_dict layout [upstream-recall, SURVEY Appendix B.4]:
    predictor conv/linear, `decoders.{i}` (FFN -> FSMN memory -> cross attention), one FFN-only `decoders3.0`, after_norm,
    output_layer.  (`decoder.embed` is unused by the non-autoregressive forward and not part of the recipe.)"""
    s = OrderedDict()
    s["predictor.cif_conv1d.weight"] = (d, d, 3)
    s["predictor.cif_conv1d.bias"] = (d,)
    s["predictor.cif_output.weight"] = (1, d)
    s["predictor.cif_output.bias"] = (1,)

    def ff(p):
        s[p + "feed_forward.w_1.weight"] = (ffn, d)
        s[p + "feed_forward.w_1.bias"] = (ffn,)
        s[p + "feed_forward.w_2.weight"] = (d, ffn)
        s[p + "feed_forward.norm.weight"] = (ffn,)
        s[p + "feed_forward.norm.bias"] = (ffn,)
        s[p + "norm1.weight"] = (d,)
        s[p + "norm1.bias"] = (d,)

    for i in range(num_blocks):
        p = f"decoder.decoders.{i}."
        ff(p)
        s[p + "self_attn.fsmn_block.weight"] = (d, 1, ksize)
        s[p + "norm2.weight"] = (d,); s[p + "norm2.bias"] = (d,)
        s[p + "src_attn.linear_q.weight"] = (d, d); s[p + "src_attn.linear_q.bias"] = (d,)
        s[p + "src_attn.linear_k_v.weight"] = (2 * d, d); s[p + "src_attn.linear_k_v.bias"] = (2 * d,)
        s[p + "src_attn.linear_out.weight"] = (d, d); s[p + "src_attn.linear_out.bias"] = (d,)
        s[p + "norm3.weight"] = (d,); s[p + "norm3.bias"] = (d,)
    ff("decoder.decoders3.0.")
    s["decoder.after_norm.weight"] = (d,)
    s["decoder.after_norm.bias"] = (d,)
    s["decoder.output_layer.weight"] = (vocab, d)
    s["decoder.output_layer.bias"] = (vocab,)
    return s


def recipe_paraformer_decoder_state_dict(seed: int = 0, num_blocks: int = 16, vocab: int = 8404) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for name, shape in paraformer_decoder_param_shapes(num_blocks, vocab=vocab).items():
        n = int(np.prod(shape))
        u = torch.from_numpy(philox_uniform("pfd:" + name, n, seed)).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if "norm" in name:
            t = (1.0 + 0.2 * u) if leaf == "weight" else 0.1 * u
        elif name == "predictor.cif_output.bias":
            t = torch.full(shape, -1.0)                      # sigmoid around 0.27: one token per ~4 encoder frames
        elif leaf == "bias":
            t = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            t = u * float(np.sqrt(3.0 / fan_in))
        out[name] = t.to(torch.float32).contiguous()
    return out


def eres2netv2_param_shapes(m: int = 64, feat_dim: int = 80, emb: int = 192, base_width: int = 24, scale: int = 4,
                            expansion: int = 4, num_blocks=(3, 4, 6, 3)) -> "OrderedDict[str, tuple]":
    """3D-Speaker ERes2NetV2 state_dict layout [upstream-recall, SURVEY Appendix B.3]."""
    s = OrderedDict()

    def bn(p, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[p + leaf] = (c,)

    def aff(p, c, r=4):
        inter = c // r
        s[p + "local_att.0.weight"] = (inter, 2 * c, 1, 1); s[p + "local_att.0.bias"] = (inter,)
        bn(p + "local_att.1.", inter)
        s[p + "local_att.3.weight"] = (c, inter, 1, 1); s[p + "local_att.3.bias"] = (c,)
        bn(p + "local_att.4.", c)

    s["conv1.weight"] = (m, 1, 3, 3); bn("bn1.", m)
    in_planes = m
    for li, (nb, stride) in enumerate(zip(num_blocks, (1, 2, 2, 2)), start=1):
        planes = m * (2 ** (li - 1))
        width = int(np.floor(planes * (base_width / 64.0)))
        for i in range(nb):
            p = f"layer{li}.{i}."
            st = stride if i == 0 else 1
            s[p + "conv1.weight"] = (width * scale, in_planes, 1, 1); bn(p + "bn1.", width * scale)
            for j in range(scale):
                s[p + f"convs.{j}.weight"] = (width, width, 3, 3); bn(p + f"bns.{j}.", width)
            s[p + "conv3.weight"] = (planes * expansion, width * scale, 1, 1); bn(p + "bn3.", planes * expansion)
            if st != 1 or in_planes != planes * expansion:
                s[p + "shortcut.0.weight"] = (planes * expansion, in_planes, 1, 1); bn(p + "shortcut.1.", planes * expansion)
            if li >= 3:
                for j in range(scale - 1):
                    aff(p + f"fuse_models.{j}.", width)
            in_planes = planes * expansion
    s["layer3_ds.weight"] = (m * 8 * expansion, m * 4 * expansion, 3, 3)
    aff("fuse34.", m * 8 * expansion)
    s["seg_1.weight"] = (emb, (feat_dim // 8) * m * 8 * expansion * 2)
    s["seg_1.bias"] = (emb,)
    return s


def recipe_eres2netv2_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for name, shape in eres2netv2_param_shapes().items():
        n = int(np.prod(shape))
        u = torch.from_numpy(philox_uniform("sv:" + name, n, seed)).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 1.0 + 0.3 * u
        elif leaf == "running_mean":
            t = 0.1 * u
        elif len(shape) == 1 and leaf == "weight":
            t = 1.0 + 0.2 * u                      # BatchNorm gamma
        elif leaf == "bias":
            t = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            t = u * float(np.sqrt(3.0 / fan_in))    # variance-preserving so 16 bottlenecks stay O(1)
        out[name] = t.to(torch.float32).contiguous()
    return out


def recipe_wave(name: str, batch: int, n: int, seed: int = 0, amp: float = 0.1) -> np.ndarray:
    """Deterministic synthetic input waveforms [batch, n] float32 in [-amp, amp)."""
    return (philox_uniform(f"wave:{name}", batch * n, seed) * np.float32(amp)).reshape(batch, n)


# ---------------------------------------------------------------------------------------
# TDXW blob: magic(8) | u32 n | n x { u16 name_len | name | u8 ndim | u32 dims[ndim] |
#            u64 data_offset (bytes from start of data section) } | pad to 64 | data (f32 LE)
# ---------------------------------------------------------------------------------------
TDXW_MAGIC = b"TDXW0001"


def pack_blob(state_dict) -> bytes:
    head = bytearray()
    head += TDXW_MAGIC
    items = [(k, v) for k, v in state_dict.items()]
    head += struct.pack("<I", len(items))
    datas = []
    off = 0
    for k, v in items:
        a = np.ascontiguousarray(v.detach().cpu().to(torch.float32).numpy() if isinstance(v, torch.Tensor)
                                 else np.asarray(v, dtype=np.float32))
        kb = k.encode()
        head += struct.pack("<H", len(kb)) + kb
        head += struct.pack("<B", a.ndim)
        for dim in a.shape:
            head += struct.pack("<I", dim)
        head += struct.pack("<Q", off)
        datas.append(a.tobytes())
        off += (a.nbytes + 63) // 64 * 64
    pad = (-len(head)) % 64
    head += b"\0" * pad
    body = bytearray()
    for dbytes in datas:
        body += dbytes
        body += b"\0" * ((-len(dbytes)) % 64)
    return bytes(head) + bytes(body)


# ---------------------------------------------------------------------------------------
# MDX-Net denoiser body (KUIELab ConvTDFNet / TFC-TDF v2; the network inside the UVR-MDX-NET *.onnx files that
# AudioProcessor.init_mdx_model hands to onnxruntime, AudioProcessor.py:224-241).  [upstream-recall]: module tree of
# kuielab/mdx-net `ConvTDFNet` (first_conv, encoding_blocks.{i}.{tfc.H.{j}, tdf}, ds.{i}, bottleneck_block, us.{i},
# decoding_blocks.{i}, final_conv), BatchNorm2d norms (the rmsprop-trained public models), no checkpoint in the reference tree.
# ---------------------------------------------------------------------------------------
def mdx_param_shapes(L: int = 11, l: int = 3, g: int = 32, k: int = 3, bn: int = 8, bias: bool = False, dim_c: int = 4,
                     dim_f: int = 3072) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    n = L // 2

    def norm(p, c):
        s[p + "weight"] = (c,); s[p + "bias"] = (c,); s[p + "running_mean"] = (c,); s[p + "running_var"] = (c,)

    def tfc_tdf(p, c, f):
        for j in range(l):
            s[f"{p}tfc.H.{j}.0.weight"] = (c, c, k, k); s[f"{p}tfc.H.{j}.0.bias"] = (c,)
            norm(f"{p}tfc.H.{j}.1.", c)
        s[p + "tdf.0.weight"] = (f // bn, f)
        if bias:
            s[p + "tdf.0.bias"] = (f // bn,)
        norm(p + "tdf.1.", c)
        s[p + "tdf.3.weight"] = (f, f // bn)
        if bias:
            s[p + "tdf.3.bias"] = (f,)
        norm(p + "tdf.4.", c)
    s["first_conv.0.weight"] = (g, dim_c, 1, 1); s["first_conv.0.bias"] = (g,)
    norm("first_conv.1.", g)
    f, c = dim_f, g
    for i in range(n):
        tfc_tdf(f"encoding_blocks.{i}.", c, f)
        s[f"ds.{i}.0.weight"] = (c + g, c, 2, 2); s[f"ds.{i}.0.bias"] = (c + g,)
        norm(f"ds.{i}.1.", c + g)
        f //= 2; c += g
    tfc_tdf("bottleneck_block.", c, f)
    for i in range(n):
        s[f"us.{i}.0.weight"] = (c, c - g, 2, 2); s[f"us.{i}.0.bias"] = (c - g,)      # ConvTranspose2d: [in, out, kh, kw]
        norm(f"us.{i}.1.", c - g)
        f *= 2; c -= g
        tfc_tdf(f"decoding_blocks.{i}.", c, f)
    s["final_conv.0.weight"] = (dim_c, c, 1, 1); s["final_conv.0.bias"] = (dim_c,)
    return s


def recipe_mdx_state_dict(seed: int = 0, **kw) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for name, shape in mdx_param_shapes(**kw).items():
        u = torch.from_numpy(philox_uniform("mdx:" + name, int(np.prod(shape)), seed)).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 1.0 + 0.3 * u
        elif leaf == "running_mean":
            t = 0.1 * u
        elif len(shape) == 1 and leaf == "weight":
            t = 1.0 + 0.2 * u                      # BatchNorm gamma
        elif leaf == "bias":
            t = 0.1 * u
        elif name.startswith("us."):               # ConvTranspose2d [in, out, 2, 2]: every output pixel sees `in` inputs
            t = u * float(np.sqrt(3.0 / shape[0]))
        else:
            fan_in = int(np.prod(shape[1:]))
            t = u * float(np.sqrt(3.0 / fan_in))    # keeps the activations O(1) through the ~35 layers (x + tdf(x), x * skip included)
        out[name] = t.to(torch.float32).contiguous()
    return out


# ---------------------------------------------------------------------------------------
# CT-Transformer punctuation model (funasr CTTransformer: Embedding -> SANM encoder -> Linear) [upstream-recall]
# ---------------------------------------------------------------------------------------
def punc_param_shapes(num_blocks: int = 4, vocab: int = 272727, npunc: int = 6, d: int = 256, ffn: int = 1024, ksize: int = 11) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    s["embed.weight"] = (vocab, d)

    def layer(p):
        s[p + "self_attn.linear_q_k_v.weight"] = (3 * d, d); s[p + "self_attn.linear_q_k_v.bias"] = (3 * d,)
        s[p + "self_attn.fsmn_block.weight"] = (d, 1, ksize)
        s[p + "self_attn.linear_out.weight"] = (d, d); s[p + "self_attn.linear_out.bias"] = (d,)
        s[p + "feed_forward.w_1.weight"] = (ffn, d); s[p + "feed_forward.w_1.bias"] = (ffn,)
        s[p + "feed_forward.w_2.weight"] = (d, ffn); s[p + "feed_forward.w_2.bias"] = (d,)
        s[p + "norm1.weight"] = (d,); s[p + "norm1.bias"] = (d,); s[p + "norm2.weight"] = (d,); s[p + "norm2.bias"] = (d,)
    layer("encoder.encoders0.0.")
    for i in range(num_blocks - 1):
        layer(f"encoder.encoders.{i}.")
    s["encoder.after_norm.weight"] = (d,); s["encoder.after_norm.bias"] = (d,)
    s["decoder.weight"] = (npunc, d); s["decoder.bias"] = (npunc,)
    return s


def recipe_punc_state_dict(seed: int = 0, num_blocks: int = 4, vocab: int = 4096, npunc: int = 6) -> "OrderedDict[str, torch.Tensor]":
    """(a small vocabulary by default: the real table is 272 727 x 256 = 279 MB)"""
    out = OrderedDict()
    for name, shape in punc_param_shapes(num_blocks, vocab, npunc).items():
        u = torch.from_numpy(philox_uniform("punc:" + name, int(np.prod(shape)), seed)).reshape(shape)
        leaf = name.rsplit(".", 1)[-1]
        if name == "embed.weight":
            t = 0.5 * u
        elif name.startswith("decoder."):
            t = u * (2.0 if leaf == "weight" else 0.5)          # spread logits: every class gets chosen somewhere ...
            if leaf == "bias":
                t = t.clone(); t[0] = -30.0                     # ... except <unk> (a trained model does not emit it; funasr would print it literally)
        elif ".norm" in name or "after_norm" in name:
            t = (1.0 + 0.2 * u) if leaf == "weight" else 0.1 * u
        elif leaf == "bias":
            t = 0.1 * u
        else:
            t = u * float(1.0 / np.sqrt(int(np.prod(shape[1:]))))
        out[name] = t.to(torch.float32).contiguous()
    return out
