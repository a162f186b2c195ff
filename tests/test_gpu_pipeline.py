"""-m gpu: the whole hot path on the reference's config-1 inputs (assets/chat_mix.wav +
assets/female_a.wav; SURVEY §8d config 1 = the hot-path subset of infer()): H1 on the mix as ONE
8.665 s window (S = 17 328), H2 on the target clip and both separated streams + cosine scores,
H3 encoder on both streams — every stage against its oracle.  Reduced depths (2-block separator,
2-block ASR encoder) keep the CPU oracle side to seconds; the full-depth models are covered by
test_gpu_mossformer2 / test_gpu_paraformer."""
import os
import wave as wavmod

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(gold, fn):
    with wavmod.open(os.path.join(gold, fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_config1_hot_path_subset(gold, sd2):
    from oracle import eres2netv2_oracle as eo
    from oracle import frontend_oracle as fo
    from oracle import mossformer2_oracle as orc
    from oracle import paraformer_oracle as po
    from targetdiarization_amd.loudness import integrated_loudness
    from targetdiarization_amd.pipeline import HotPath
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_state_dict
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    assert len(mix) == 138634 and len(tgt) == 30768
    spk_sd, asr_sd = recipe_eres2netv2_state_dict(0), recipe_paraformer_state_dict(0, 2)
    hp = HotPath(sd2, spk_sd, asr_sd, cuda_device=0)
    tgt_emb = hp.spk.get_speaker_embedding(tgt)
    res = hp.run([mix], target_embedding=tgt_emb)

    # ---- oracle chain (fp32 separator like the reference's CPU path; fp64 for the rest)
    y = orc.mossformer2_forward(torch.from_numpy(mix)[None], sd2)[0].numpy()
    s1, s2 = y[0], y[1]
    if round(integrated_loudness(s1, 16000), 1) < round(integrated_loudness(s2, 16000), 1):
        s1, s2 = s2, s1
    a, b = res["streams"][0]
    assert rel(a, s1) < 1e-4 and rel(b, s2) < 1e-4
    spk64 = {k: v.double() for k, v in spk_sd.items()}
    asr64 = {k: v.double() for k, v in asr_sd.items()}
    emb_ref = lambda w: eo.eres2netv2_forward(fo.sv_features(torch.from_numpy(w).double())[None], spk64)[0].numpy()
    e_t, e1, e2 = emb_ref(tgt), emb_ref(s1), emb_ref(s2)
    assert rel(tgt_emb, e_t) < 1e-4
    assert rel(res["embeddings"][0], e1) < 1e-3 and rel(res["embeddings"][1], e2) < 1e-3
    for i, e in enumerate((e1, e2)):
        assert abs(res["scores"][i] - orc.cosine_similarity(e, e_t)) < 1e-3
    for i, s in enumerate((s1, s2)):
        feats = fo.asr_features(torch.from_numpy(s).double(), torch.zeros(560, dtype=torch.float64), torch.ones(560, dtype=torch.float64))
        enc = po.sanm_encoder_forward(feats[None], asr64)[0].numpy()
        assert res["encoder"][i].shape == enc.shape
        assert rel(res["encoder"][i], enc) < 1e-3


def test_multi_utterance_batching_is_order_invariant(sd2):
    """three utterances of different lengths (windows 160000/160000/80001, 50000, 160000):
    results of the batched run equal per-utterance runs."""
    from targetdiarization_amd.pipeline import HotPath
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_wave
    hp = HotPath(sd2, recipe_eres2netv2_state_dict(0), None, cuda_device=0)
    utts = [recipe_wave("u0", 1, 400001)[0], recipe_wave("u1", 1, 50000)[0], recipe_wave("u2", 1, 160000)[0]]
    full = hp.run(utts)
    for i, u in enumerate(utts):
        one = hp.run([u])
        assert rel(full["streams"][i][0], one["streams"][0][0]) < 1e-5
        assert rel(full["embeddings"][2 * i], one["embeddings"][0]) < 1e-4
