"""CPU tests: the oracle (oracle/mossformer2_oracle.py) against the fixtures the REFERENCE
produced (oracle/make_goldens.py), plus the host-side pieces of the path."""
import json
import os
import wave as wavmod
import zlib

import numpy as np
import pytest
import torch

from oracle import mossformer2_oracle as orc
from targetdiarization_amd.weights import (mossformer2_param_shapes, pack_blob, philox_uniform,
                                           recipe_state_dict, recipe_wave)


def rel_l2(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm())


def test_recipe_is_deterministic_and_complete(sd24):
    shapes = mossformer2_param_shapes()
    assert len(shapes) == 1099                                  # 1 099 state_dict tensors
    assert sum(int(np.prod(s)) for s in shapes.values()) == 55735410 + 256 + 23 * 16   # params + inv_freq buffer + 23 aliases of the shared rotary freqs
    again = recipe_state_dict(seed=0, num_blocks=24)
    for k in ("enc.conv1d.weight", "dec.weight", orc.PFX + "layers.23.to_out.mdl.1.weight"):
        assert torch.equal(sd24[k], again[k])
    assert not torch.equal(sd24["enc.conv1d.weight"], recipe_state_dict(seed=1, num_blocks=1)["enc.conv1d.weight"])


@pytest.mark.parametrize("tag,B,T", [("1x16000", 1, 16000), ("2x8000", 2, 8000)])
def test_oracle_matches_reference_outputs(gold, sd24, tag, B, T):
    fx = np.load(os.path.join(gold, "g1_mossformer2_24blk_seed0.npz"))
    x = torch.from_numpy(recipe_wave(f"g1:{tag}", B, T))
    out = orc.mossformer2_forward(x, sd24)
    assert out.shape == (B, 2, T)
    assert rel_l2(out, fx[tag]) < 3e-5


def test_oracle_taps_match_reference(gold, sd24):
    fx = np.load(os.path.join(gold, "g1_taps_1x4000_seed0.npz"))
    x = torch.from_numpy(recipe_wave("g1:taps", 1, 4000))
    taps = {}
    out = orc.mossformer2_forward(x, sd24, taps=taps)
    assert rel_l2(out, fx["out"]) < 3e-5
    idx = torch.as_tensor(fx["tokens"])
    for k in ("enc", "after_flash0", "after_fsmn0", "after_stack", "mask"):
        t = taps[k]
        if k == "enc":
            t = t.permute(0, 2, 1)
        if k == "mask":
            t = t.permute(0, 1, 3, 2)
        assert rel_l2(t.index_select(-2, idx), fx[k]) < 3e-5, k
        assert abs(float(t.double().sum()) - float(fx[k + "_sum"])) / float(fx[k + "_abssum"]) < 1e-5, k


def test_oracle_cal_attention_and_ddn(gold, sd24):
    fx = np.load(os.path.join(gold, "g2_cal_attention_2x600.npz"))
    B, S, E = 2, 600, 64
    u_ = lambda nm, *shape: torch.from_numpy(philox_uniform("g2:" + nm, int(np.prod(shape)))).reshape(shape)
    qq, lq, qk, lk = (u_(n, B, S, 128) for n in ("qq", "lq", "qk", "lk"))
    v, u = u_("v", B, S, E), u_("u", B, S, E)
    av, au = orc.cal_attention(qq, lq, qk, lk, v, u, sd24[orc.PFX + "layers.0.rotary_pos_emb.freqs"])
    assert rel_l2(av, fx["att_v"]) < 1e-5 and rel_l2(au, fx["att_u"]) < 1e-5
    fx3 = np.load(os.path.join(gold, "g3_dilated_dense_net_2x150.npz"))
    p = u_("ddn", 2, 150, 256)
    o3 = orc.dilated_dense_net(p, sd24, orc.PFX + "fsmn.0.gated_fsmn.fsmn.conv.")
    assert rel_l2(o3, fx3["out"]) < 1e-5


def test_window_plan(gold):
    g4 = json.load(open(os.path.join(gold, "g4_window_plan.json")))
    for n, plan in g4.items():
        got = orc.window_plan(int(n))
        assert [list(t) for t in got] == plan
        # windows tile [0,n) exactly, each in (0, 240000]
        assert got[0][0] == 0 and got[-1][1] == int(n)
        for (a, b), (c, d) in zip(got, got[1:]):
            assert b == c
        assert all(0 < e - s <= 240000 for s, e in got)


def test_cosine_similarity_edges():
    a = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    assert orc.cosine_similarity(np.zeros(3), a) == 1.0            # TargetASR.py:145-146
    assert orc.cosine_similarity(a, -a) == 0.0                     # clipped below
    assert abs(orc.cosine_similarity(a, 2 * a) - 1.0) < 1e-7
    assert isinstance(orc.cosine_similarity(a, a), float)


def test_assets_identity(gold):
    rep = json.load(open(os.path.join(gold, "pin_report.json")))
    for fn in ("chat_mix.wav", "female_a.wav"):
        with wavmod.open(os.path.join(gold, fn), "rb") as w:
            raw = w.readframes(w.getnframes())
            assert w.getframerate() == 16000 and w.getnchannels() == 1
            assert w.getnframes() == rep[f"asset_{fn}"]["n"]
            assert zlib.crc32(raw) == rep[f"asset_{fn}"]["crc32"]


def test_blob_roundtrip(sd2):
    import struct
    blob = pack_blob(sd2)
    assert blob[:8] == b"TDXW0001"
    (n,) = struct.unpack_from("<I", blob, 8)
    assert n == len(sd2)
