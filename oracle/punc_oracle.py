"""CT-Transformer punctuation model, CPU restatement (test infrastructure; checker only).

The reference calls `self.punc.inference(input=text)` (ASRProcessor.punctuation_restore, ASRProcessor.py:880-897) on a funasr
`CTTransformer` (model dir `iic/punc_ct-transformer_zh-cn-common-vocab272727-pytorch`); funasr is third-party, unpinned and absent,
no weights in the tree => PARITY UNPINNED.  Restated from the published funasr code [upstream-recall]:
    punc_forward: x = Embedding(ids); h = SANMEncoder(x) (input_layer "pe": x * sqrt(d) + sinusoidal PE, encoders0 + encoders, after_norm);
                  y = Linear(h)                                   config: embed_unit = att_unit = 256, 8 heads, FFN 1024, 4 blocks, kernel 11
    inference:    words -> ids -> mini-sentences of `split_size` (20) words; each window is the cached tail of the previous one + the new
                  words; argmax punctuation per word; every window but the last is cut after its last period / question mark (or, beyond
                  200 cached words, at the last comma, which becomes a period) and the rest is carried over; ASCII words are joined with
                  spaces, get ASCII punctuation and a capital letter after a sentence end; the text ends with a period.
`inference` here mirrors targetdiarization_amd/punctuation.py's contract on the ORACLE's logits, for end-to-end checks."""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12


def sinusoidal_pe(T, depth, dtype):
    pos = torch.arange(1, T + 1, dtype=dtype)
    inc = math.log(10000.0) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth // 2, dtype=dtype) * (-inc))
    st = pos[:, None] * inv[None, :]
    return torch.cat((torch.sin(st), torch.cos(st)), dim=1)


def _attention(x, sd, p, heads, ksize):
    B, T, _ = x.shape
    qkv = F.linear(x, sd[p + "linear_q_k_v.weight"], sd[p + "linear_q_k_v.bias"])
    n = qkv.shape[-1] // 3
    q, k, v = torch.split(qkv, n, dim=-1)
    dk = n // heads
    qh = q.reshape(B, T, heads, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(B, T, heads, dk).transpose(1, 2)
    vh = v.reshape(B, T, heads, dk).transpose(1, 2)
    att = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)), dim=-1)
    ctx = torch.matmul(att, vh).transpose(1, 2).reshape(B, T, n)
    out = F.linear(ctx, sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])
    left = (ksize - 1) // 2
    mem = F.conv1d(F.pad(v.transpose(1, 2), (left, ksize - 1 - left)), sd[p + "fsmn_block.weight"], groups=n).transpose(1, 2) + v
    return out + mem


def _layer(x, sd, p, heads, ksize):
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    x = x + _attention(h, sd, p + "self_attn.", heads, ksize)
    h = F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    return x + F.linear(F.relu(F.linear(h, sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"])),
                        sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])


def punc_forward(ids: torch.Tensor, sd, heads: int = 8, ksize: int = 11):
    """ids int64 [B, T] -> logits [B, T, npunc]"""
    num_blocks = 2 + max([int(k.split("encoders.")[1].split(".")[0]) for k in sd if ".encoders." in k] + [-1])
    emb = sd["embed.weight"]
    d = emb.shape[1]
    x = emb[ids] * (d ** 0.5) + sinusoidal_pe(ids.shape[1], d, emb.dtype)[None]
    x = _layer(x, sd, "encoder.encoders0.0.", heads, ksize)
    for i in range(num_blocks - 1):
        x = _layer(x, sd, f"encoder.encoders.{i}.", heads, ksize)
    x = F.layer_norm(x, (d,), sd["encoder.after_norm.weight"], sd["encoder.after_norm.bias"], LN_EPS)
    return F.linear(x, sd["decoder.weight"], sd["decoder.bias"])
