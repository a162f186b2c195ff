"""-m gpu: TargetDiarization.infer (N1) end to end on the reference's config-1 assets with
synthetic segmentation plug-ins (the CAM++/pyannote/FSMN-VAD models are third-party and absent):
checks the contract of the return value and that the batched hot loops give the same decisions as
the reference's per-segment loop restated with the oracles' scorer."""
import os
import wave as wavmod

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(gold, fn):
    with wavmod.open(os.path.join(gold, fn), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0


def test_infer_contract_and_decisions(gold, sd2):
    from oracle import mossformer2_oracle as orc
    from targetdiarization_amd import intervals as iv
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_state_dict
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
    od = [(0.0, 3.0, "SPEAKER_00"), (2.4, 5.5, "SPEAKER_01"), (5.5, 8.6, "SPEAKER_00")]
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0),
                           asr_state_dict=recipe_paraformer_state_dict(0, 2), sd_pipeline=lambda a: sd_rows,
                           od_pipeline=lambda a: od, target_similarity_threshold=0.0)
    target_spk, results, target_audio = td.infer(mix, tgt)
    assert target_spk in ("0", "1")
    assert results and all(set(r) >= {"speaker", "timerange", "text", "type", "score"} for r in results)
    assert all("audio" not in r for r in results)
    assert [r["timerange"][0] for r in results] == sorted(r["timerange"][0] for r in results)
    assert {r["type"] for r in results} <= {"single", "overlap"}
    assert any(r["type"] == "overlap" for r in results)            # 2.4-3.0 s is a 0.6 s overlap (>= 0.4 s)
    assert target_audio is not None and target_audio.dtype == np.float32
    assert abs(len(target_audio) / 16000 - results[-1]["timerange"][1]) < 0.05
    # decision of hot loop A restated per segment (reference semantics :581-600) with the host scorer
    audio = td.audio_preprocess(mix)
    sd_res = iv.sd_result_parser(sd_rows)
    od_res = td.od_result_parser(od, sd_result=sd_res)
    refined, omap = iv.apply_od_result(sd_res, od_res)
    single = iv.subtract_overlap(refined, omap)
    temb = td.hp.spk.get_speaker_embedding(td.audio_preprocess(tgt))
    means = {}
    for spk, rs in single.items():
        sc = [orc.cosine_similarity(temb, td.hp.spk.get_speaker_embedding(td.split_audio_by_time(audio, s, e))) for (s, e) in rs]
        means[spk] = sum(sc) / len(sc)
    assert target_spk == max(means, key=means.get)


def test_infer_single_and_no_target(gold, sd2):
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict
    mix = _load(gold, "chat_mix.wav")
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0))
    spk, res, aud = td.infer(mix, None, is_single=True, output_target_audio=False)
    assert spk == "" and aud is None and len(res) == 1 and res[0]["speaker"] == "0" and res[0]["score"] == -1.0
    spk2, res2, aud2 = td.infer(mix, None)        # no target file: longest speaker becomes the target
    assert spk2 == "0" and aud2 is not None


def test_target_asr_and_asr_processor_mirrors(gold):
    """the reference-named host classes route through the same C-ABI paths"""
    from targetdiarization_amd.asr_processor import ASRProcessor
    from targetdiarization_amd.target_asr import TargetASR
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_state_dict
    tgt = _load(gold, "female_a.wav")
    t = TargetASR(cuda_device=0, spk_state_dict=recipe_eres2netv2_state_dict(0))
    e1 = t.get_speaker_embedding(tgt)
    e2 = t.get_speaker_embeddings([tgt, tgt[:16000]])
    assert e1.shape == (192,) and e1.dtype == np.float32 and np.allclose(e1, e2[0], rtol=1e-5, atol=1e-6)
    assert 0.0 <= t.cosine_similarity(e1, e2[1]) <= 1.0 and abs(t.cosine_similarity(e1, e1) - 1.0) < 1e-6
    seen = {}

    def decoder(enc):
        seen["shape"] = tuple(enc.shape)
        return {"text": "a b", "timestamp": [[0, 500], [500, 900]]}
    ap = ASRProcessor(is_asr=True, cuda_device=0, asr_state_dict=recipe_paraformer_state_dict(0, 2), decoder=decoder)
    res = ap.asr_detection(tgt, asr_engine="paraformer", no_punc=True)
    frames = 1 + (len(tgt) - 400) // 160
    assert seen["shape"] == ((frames + 5) // 6, 512)
    assert res[0]["timestamp"] == [("a", [0.0, 0.5]), ("b", [0.5, 0.9])] and res[0]["language"] == "en" and res[0]["key"] == "clip_0"
    assert ap.asr_detection(tgt, output_text_only=True) == "a b"


def test_infer_with_device_decoder_and_denoiser(gold, sd2):
    """the full chain of infer(): audio_preprocess (loudness -> denoise_vocal with an identity MDX body -> loudness, :166-182), separation
    of the overlap, target selection, and the ASR forward down to tokens through the device CIF + NAR decoder (N2)"""
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import (recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict,
                                               recipe_paraformer_state_dict)
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    asr_sd = dict(recipe_paraformer_state_dict(0, 2)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
    od = [(0.0, 3.0, "SPEAKER_00"), (2.4, 5.5, "SPEAKER_01"), (5.5, 8.6, "SPEAKER_00")]
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0), asr_state_dict=asr_sd,
                           sd_pipeline=lambda a: sd_rows, od_pipeline=lambda a: od, mdx_model=lambda spec: spec,
                           mdx_weights_file="mdx/weights/Kim_Vocal_2.onnx", token_list=[f"w{i}|" for i in range(8404)])
    assert td.hp.ap.is_denoise_vocal and td.hp.dec is not None
    pre = td.audio_preprocess(mix)
    assert pre.shape == mix.shape and pre.dtype == np.float32 and np.isfinite(pre).all()
    spk, res, aud = td.infer(mix, tgt)
    assert spk in ("0", "1") and res
    texts = [r["text"] for r in res]
    # recipe weights: arbitrary tokens, but tokens; Latin-letter tokens are joined with a leading space each, like the reference (:813-814)
    assert any(t for t in texts) and all(t == "" or t.startswith(" w") for t in texts)


def test_infer_with_the_device_mdx_body_and_punctuation(gold, sd2):
    """every neural stage of infer() on the device: audio_preprocess with the ConvTDFNet denoiser body (N3, `mdx_state_dict`; small geometry),
    separation, embeddings, Paraformer encoder + CIF + NAR decoder (N2) and the CT-Transformer punctuation restorer (N3, `punc_state_dict`):
    the result texts are the decoder's tokens with punctuation marks from the device model's own inference() on the same text"""
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import (recipe_eres2netv2_state_dict, recipe_mdx_state_dict, recipe_paraformer_decoder_state_dict,
                                               recipe_paraformer_state_dict, recipe_punc_state_dict)
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    asr_sd = dict(recipe_paraformer_state_dict(0, 2)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
    od = [(0.0, 3.0, "SPEAKER_00"), (2.4, 5.5, "SPEAKER_01"), (5.5, 8.6, "SPEAKER_00")]
    geo = dict(L=3, l=1, g=32, bn=8, dim_f=3072)
    chars = [chr(0x4e00 + i) for i in range(8404)]                      # one CJK ideograph per token id
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0), asr_state_dict=asr_sd,
                           sd_pipeline=lambda a: sd_rows, od_pipeline=lambda a: od, token_list=chars,
                           mdx_state_dict=recipe_mdx_state_dict(2, **geo), mdx_args=dict(geo, dim_t=256, max_blocks_per_launch=2),
                           mdx_weights_file="mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx", punc_state_dict=recipe_punc_state_dict(0, vocab=512))
    assert td.hp.ap.is_denoise_vocal and td.hp.dec is not None and td.punctuation is not None
    pre = td.audio_preprocess(mix)
    assert pre.shape == mix.shape and np.isfinite(pre).all()
    assert np.abs(pre - mix).max() > 1e-4                               # the denoiser body ran (recipe weights: it changes the audio)
    spk, res, aud = td.infer(mix, tgt)
    assert spk in ("0", "1") and res and aud is not None and np.isfinite(aud).all()
    marks = set("，。？、")
    texts = [r["text"] for r in res if r["text"]]
    assert texts
    for t in texts:
        bare = "".join(ch for ch in t if ch not in marks)
        assert bare and all(0x4e00 <= ord(ch) < 0x4e00 + 8404 for ch in bare)
        assert t[-1] in "。？"                                           # CTTransformer.inference ends a text with a sentence end
        assert t == td.punctuation(bare)


def test_serving_shell_over_the_device_model(gold, sd2):
    """N4: POST /diarization/infer with the config-1 assets as multipart uploads gives the same decisions as a direct infer()"""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient
    from targetdiarization_amd.server import create_app
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict
    sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
    od = [(0.0, 3.0, "SPEAKER_00"), (2.4, 5.5, "SPEAKER_01"), (5.5, 8.6, "SPEAKER_00")]
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0), sd_pipeline=lambda a: sd_rows,
                           od_pipeline=lambda a: od, target_similarity_threshold=0.0)
    spk, res, aud = td.infer(_load(gold, "chat_mix.wav"), _load(gold, "female_a.wav"))
    c = TestClient(create_app(td))
    with open(os.path.join(gold, "chat_mix.wav"), "rb") as f, open(os.path.join(gold, "female_a.wav"), "rb") as g:
        j = c.post("/diarization/infer", files={"audio_file": ("m.wav", f.read(), "audio/wav"), "target_file": ("t.wav", g.read(), "audio/wav")}).json()
    assert j["success"], j["error"]
    d = j["data"]
    assert d["target_speaker_id"] == spk and len(d["results"]) == len(res)
    assert [(r["speaker"], r["type"]) for r in d["results"]] == [(r["speaker"], r["type"]) for r in res]
    assert np.allclose([r["timerange"] for r in d["results"]], [r["timerange"] for r in res])
    import base64
    pcm = np.frombuffer(base64.b64decode(d["target_audio_base64"]), dtype=np.int16)
    assert pcm.shape[0] == aud.shape[0] and np.abs(pcm.astype(np.float32) / 32767.0 - aud).max() < 1e-3


def test_streaming_session_on_the_device(gold, sd2):
    """N4: TargetDiarizationStream over the device hot path — the config-1 mix in 1 s chunks, target clip given; every released
    buffer went through audio_preprocess(stream_mode) = MossFormer2, ERes2NetV2 for the speaker label, Paraformer + CIF + decoder"""
    from targetdiarization_amd.target_diarization_stream import TargetDiarizationStream
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict, recipe_paraformer_state_dict
    asr_sd = dict(recipe_paraformer_state_dict(0, 2)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 2))
    s = TargetDiarizationStream(max_buffer_duration=3.0, cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0),
                                asr_state_dict=asr_sd, od_pipeline=lambda a: [])
    mix, tgt = _load(gold, "chat_mix.wav"), _load(gold, "female_a.wav")
    chunks = [(mix[i:i + 16000] * 32767).astype(np.int16) for i in range(0, mix.shape[0] - 8000, 16000)]
    out = list(s.infer_stream(iter(chunks), target_file=tgt, output_target_audio=True))
    assert len(out) >= 2 and s.target_embedding is not None and s.target_embedding.shape == (192,)
    t_prev = 0.0
    for spk, res, audio in out:
        assert spk == "1" and len(res) == 1
        r = res[0]
        assert set(r) >= {"speaker", "timerange", "text", "type"} and r["speaker"] in ("0", "1") and r["type"] == "single" and r["text"]
        assert r["timerange"][0] >= t_prev
        t_prev = r["timerange"][0]


def test_infer_without_target_clip_uses_get_target_embedding(gold, sd2):
    """N1: infer(mix, None) -> sd_result_to_target_embedding (:551-578) -> TargetASR.get_target_embedding (VAD pieces re-joined,
    loudness control, 30 s cap, one embedding) on the device embedder; list inputs go through the batched launch + HDBSCAN + mean"""
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict
    mix = _load(gold, "chat_mix.wav")
    sd_rows = {"text": [[0.0, 3.0, 0], [2.4, 5.5, 1], [5.5, 8.6, 0]]}
    vad = lambda a: [[0.1, round(a.shape[0] / 16000.0 - 0.1, 3)]] if a.shape[0] > 6400 else []
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0), sd_pipeline=lambda a: sd_rows, vad=vad)
    spk, res, aud = td.infer(mix, None)
    assert spk == "0" and res and aud is not None            # speaker 0 has the longer total duration
    # the embedding the orchestrator used = get_target_embedding of speaker 0's first non-overlap piece (the reference's overwrite quirk)
    piece = td.split_audio_by_time(mix, 0.0, 3.0)
    want = td.hp.spk.get_speaker_embedding(td.audio_loudness_control(piece[1600:int(2.9 * 16000)]))
    _, got = td.sd_result_to_target_embedding(mix, {"0": [(0.0, 3.0), (5.5, 8.6)], "1": [(2.4, 5.5)]}, [])
    assert got.shape == (192,) and np.allclose(got, want, rtol=1e-4, atol=1e-6)
    # list input: per-clip embeddings in ONE bucketed launch sequence, clustering, mean
    clips = [mix[i * 16000:(i + 2) * 16000].copy() for i in range(0, 6, 1)]
    embs = td.hp.spk.get_speaker_embeddings([c for c in clips])
    from targetdiarization_amd.target_asr import target_embedding_from_audio
    out = target_embedding_from_audio([f"c{i}" for i in range(len(clips))], td.hp.spk.get_speaker_embeddings, vad, td.audio_loudness_control,
                                      read_audio=lambda p: clips[int(p[1:])], is_preprocess=False, is_cluster=False, output_embedding_list=True)
    assert len(out) == len(clips) and np.allclose(np.stack(out), embs, rtol=1e-4, atol=1e-6)
    mean = target_embedding_from_audio([f"c{i}" for i in range(len(clips))], td.hp.spk.get_speaker_embeddings, vad, td.audio_loudness_control,
                                       read_audio=lambda p: clips[int(p[1:])], is_preprocess=False, is_cluster=True, output_embedding_list=False)
    assert mean.shape == (192,) and np.isfinite(mean).all()


def test_infer_accepts_wav_paths_and_other_rates(gold, sd2):
    """the reference's infer() takes paths / file objects at any rate (TargetDiarization.py:98-121): a .wav path gives the result
    of the array call, and an 8 kHz copy goes through the device resampler"""
    from targetdiarization_amd.target_diarization import TargetDiarization
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict
    td = TargetDiarization(cuda_device=0, sep_state_dict=sd2, spk_state_dict=recipe_eres2netv2_state_dict(0))
    mix = _load(gold, "chat_mix.wav")
    spk_a, res_a, aud_a = td.infer(mix, None)
    spk_p, res_p, aud_p = td.infer(os.path.join(gold, "chat_mix.wav"), None)
    assert spk_a == spk_p and [r["timerange"] for r in res_a] == [r["timerange"] for r in res_p]
    assert np.abs(aud_a - aud_p).max() < 1e-4 * max(np.abs(aud_a).max(), 1e-9) + 1e-6
    spk_8, res_8, aud_8 = td.infer(mix[::2].copy(), None, sampling_rate=8000)
    assert abs(res_8[-1]["timerange"][1] - res_a[-1]["timerange"][1]) < 0.01 and aud_8.shape[0] == aud_a.shape[0]
