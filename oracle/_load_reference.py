"""Test infrastructure (container-only): import the reference's MossFormer2 by path.

Only used by oracle/make_goldens.py *in the build container* (where /root/reference
exists) to validate the CPU restatement and to mint golden fixtures.  Nothing here is
imported by the product path, the -m gpu tests, smoke() or bench.py; /root/reference
does not exist on the GPU box.

Procedure = SURVEY.md Appendix C:
 * `rotary_embedding_torch` (third-party, un-vendored, unpinned: requirements.txt:28) is
   absent, so a stand-in module exposing RotaryEmbedding(dim).rotate_queries_or_keys is
   inserted in sys.modules.  The stand-in restates the published lucidrains algorithm
   (SURVEY Appendix B.1): freqs = theta^-(arange(0,dim,2)/dim), interleaved-pair rotation
   of the first `dim` features with angle pos*freqs.  => rotary parity is pinned to this
   stand-in, not to the real package ("rotary parity unpinned").
 * the six MossFormer2 source files are loaded as a synthetic package so that
   look2hear/models/__init__.py (which eagerly imports absent deps) is never executed.
"""
import importlib.util
import sys
import types

import torch
from torch import nn

REF_MODELS = "/root/reference/look2hear/models"


class _RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self.dim = dim

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        pos = torch.arange(n, device=t.device).type(self.freqs.dtype)
        ang = torch.einsum("i,j->ij", pos, self.freqs)          # [n, dim/2]
        ang = ang.repeat_interleave(2, dim=-1)                   # [n, dim] pairwise
        rot, rest = t[..., : self.dim], t[..., self.dim:]
        x1 = rot[..., 0::2]
        x2 = rot[..., 1::2]
        half = torch.stack((-x2, x1), dim=-1).reshape(rot.shape)
        rot = rot * ang.cos() + half * ang.sin()
        return torch.cat((rot, rest), dim=-1)


def load_reference_package():
    if "l2h" in sys.modules:
        return sys.modules["l2h"]
    ret = types.ModuleType("rotary_embedding_torch")
    ret.RotaryEmbedding = _RotaryEmbedding
    sys.modules["rotary_embedding_torch"] = ret
    pkg = types.ModuleType("l2h")
    pkg.__path__ = [REF_MODELS]
    sys.modules["l2h"] = pkg
    for name in ("base_model", "layer_norm", "fsmn", "conv_module", "mossformer_block", "mossformer2"):
        spec = importlib.util.spec_from_file_location(f"l2h.{name}", f"{REF_MODELS}/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"l2h.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    return pkg


def build_reference_mossformer2(**kw):
    pkg = load_reference_package()
    return pkg.mossformer2.MossFormer2(**kw).eval()
