"""Mint tests/golden/n1_target_embedding.json / n1_asr_audio.json by EXECUTING the reference's own methods in the
build container (same arrangement as make_goldens_n1.py).  Test infrastructure.

  * TargetASR.get_target_embedding            TargetASR.py:166-258
  * TargetDiarization.sd_result_to_target_embedding   TargetDiarization.py:551-578
  * TargetDiarization.sd_result_to_asr_audio  TargetDiarization.py:716-820 (timestamp AND no-timestamp branches)

The classes are created without __init__; absent third-party imports are empty placeholder modules.  Everything the
methods call on other objects is a small deterministic stand-in defined HERE and re-implemented identically in
tests/test_n1_target_embedding.py, so that the goldens pin the reference's CONTROL FLOW (VAD split / re-join, the 0.4 s
and 400-sample filters, the 30 s cap, merge / longest / separate / auto, the NaN skip, the "> 2 embeddings" clustering
gate, the noise drop, list vs mean vs zero fallback; speaker choice, overlap subtraction; item assembly, per-speaker
timelines, token-to-chunk assignment by 0.1 s rounded ranges, the language rule, the one-item-per-speaker branch):
    embed(audio)        -> e[0] = rms, e[1] = seconds/100, e[2] = 1 (NaN when the clip's first sample is exactly 7.0)
    vad(audio)          -> runs of |x| > 1e-4 at 10 ms resolution, seconds rounded to 3
    loudness_control(a) -> a scaled to rms 0.05 (clips >= 0.4 s with rms > 0)
    hdbscan.HDBSCAN     -> sklearn.cluster.HDBSCAN (same published algorithm; the labels themselves are "parity
                           unpinned" against the hdbscan package, which is absent)
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SR = 16000


# ---------------------------------------------------------------- the stand-ins (mirrored in the test)
def embed(audio):
    a = np.asarray(audio, dtype=np.float64).reshape(-1)
    e = np.zeros(192, dtype=np.float32)
    e[0] = np.sqrt(np.mean(a * a)) if a.size else 0.0
    e[1] = a.size / SR / 100.0
    e[2] = 1.0
    if a.size and a[0] == 7.0:
        e[5] = np.nan
    return e


def vad(audio):
    a = np.abs(np.asarray(audio, dtype=np.float32).reshape(-1))
    n = a.shape[0] // 160
    act = a[: n * 160].reshape(n, 160).max(axis=1) > 1e-4 if n else np.zeros(0, bool)
    out, start = [], None
    for i, v in enumerate(list(act) + [False]):
        if v and start is None:
            start = i
        if not v and start is not None:
            out.append([round(start * 0.01, 3), round(i * 0.01, 3)]); start = None
    return out


def loudness_control(audio, sampling_rate=SR):
    if audio.shape[0] / sampling_rate < 0.4:
        return audio
    r = float(np.sqrt(np.mean(audio.astype(np.float64) ** 2)))
    return audio if r == 0.0 else (audio * (0.05 / r)).astype(np.float32)


def split_audio_by_time(audio_data, sampling_rate, start_time, end_time):
    s = max(0, int(start_time * sampling_rate)); e = min(int(end_time * sampling_rate), audio_data.shape[0])
    return audio_data.copy()[s:e]


def asr_stub(audio, with_timestamp, language):
    """one token per 0.5 s of the timeline: ("t<k>", [0.5k, 0.5k+0.3])"""
    n = int(audio.shape[0] / SR / 0.5)
    stamps = [(f"t{k}", [round(0.5 * k, 3), round(0.5 * k + 0.3, 3)]) for k in range(n)]
    res = {"text": " ".join(t for t, _ in stamps), "language": language}
    if with_timestamp:
        res["timestamp"] = stamps
    return res


def separate_stub(clip):
    """stands for TargetASR.multi_speakers_separate_asr(is_output_asr=False): by clip length, two / one / no results"""
    d = round(clip.shape[0] / SR, 3)
    k = int(clip.shape[0]) % 3
    if k == 0:
        return []
    res = [{"timerange": [0.1, round(d - 0.1, 3)], "text": "", "score": 0.9, "sampling_rate": SR, "audio": (clip * 0.5).astype(np.float32)}]
    if k == 2:
        res.append({"timerange": [0.2, d], "text": "", "score": 0.1, "sampling_rate": SR, "audio": (clip * 0.25).astype(np.float32)})
    return res


def synth(seed, segments):
    """audio from [(seconds, level)]: level 0 = silence, else uniform noise of that amplitude"""
    rng = np.random.default_rng(seed)
    return np.concatenate([(rng.uniform(-1, 1, int(s * SR)) * lv).astype(np.float32) for s, lv in segments])


# ---------------------------------------------------------------- loading the reference classes
def load_reference():
    from sklearn.cluster import HDBSCAN
    for name in ("modelscope", "pyannote", "pyannote.audio", "dotenv", "silero_vad", "hdbscan", "AudioProcessor", "ASRProcessor"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pyannote.audio"].Pipeline = object
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None
    sys.modules["silero_vad"].load_silero_vad = sys.modules["silero_vad"].get_speech_timestamps = lambda *a, **k: None
    sys.modules["modelscope"].pipeline = lambda *a, **k: None
    sys.modules["hdbscan"].HDBSCAN = HDBSCAN
    sys.modules["AudioProcessor"].AudioProcessor = object
    sys.modules["ASRProcessor"].ASRProcessor = object
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    tasr_mod = load("TargetASR", "/root/reference/TargetASR.py")
    td_mod = load("ref_td_b", "/root/reference/TargetDiarization.py")
    return tasr_mod.TargetASR, td_mod.TargetDiarization


class _AP:
    verbose_log = False
    split_audio_by_time = staticmethod(lambda audio_data, sampling_rate, start_time, end_time: split_audio_by_time(audio_data, sampling_rate, start_time, end_time))
    audio_loudness_control = staticmethod(lambda audio_data, sampling_rate, target_loudness=-23.0: loudness_control(audio_data, sampling_rate))

    @staticmethod
    def combine_audio_chunks(audio_data_list):      # AudioProcessor.py:1116-1120 (three lines, restated: the class itself needs its whole import list)
        return audio_data_list[0] if len(audio_data_list) == 1 else np.concatenate(audio_data_list)


class _ASRP:
    def __init__(self, with_timestamp=True, language="zh"):
        self.with_timestamp, self.language = with_timestamp, language

    def vad_detection(self, wav_file, **kw):
        return vad(wav_file)

    def asr_detection(self, wav_file, asr_engine="paraformer", **kw):
        return [asr_stub(wav_file, self.with_timestamp, self.language)]

    def punctuation_restore(self, text):
        return text + "." if text else text


def make_tasr(TargetASR, **asrp_kw):
    t = object.__new__(TargetASR)
    t.verbose_log = False
    t.ap = _AP()
    t.asrp = _ASRP(**asrp_kw)
    t.embedding = {"eres2netv2_large": lambda wav, output_emb=True: {"embs": embed(wav)[None]}}
    return t


def rle_sum(a):
    """compact fingerprint of an audio array: length, sum, sum of squares, first / last sample"""
    a = np.asarray(a, dtype=np.float64)
    return [int(a.shape[0]), float(a.sum()), float((a * a).sum()), float(a[0]) if a.size else 0.0, float(a[-1]) if a.size else 0.0]


def main():
    TargetASR, TargetDiarization = load_reference()
    out = {"get_target_embedding": [], "sd_to_target_embedding": []}
    # ---- get_target_embedding: specs are lists of segment lists (one per input audio); None = ndarray input
    specs = [
        ("one clip, speech-silence-speech", [[(1.0, 0.1), (0.5, 0), (1.2, 0.2)]], False),
        ("one clip, silence only", [[(1.0, 0)]], False),
        ("one clip, 40 s (30 s cap)", [[(40.0, 0.1)]], False),
        ("one clip, 0.3 s (below the 0.4 s filter)", [[(0.3, 0.1)]], False),
        ("five clips, two groups + an outlier", [[(1.0, 0.1)], [(1.1, 0.1)], [(1.0, 0.3)], [(1.05, 0.3)], [(1.0, 0.9)]], True),
        ("three clips + a short one + silence", [[(1.0, 0.1)], [(0.2, 0.1)], [(1.0, 0.11)], [(0.6, 0)], [(1.0, 0.5)]], True),
        ("two clips (no clustering)", [[(1.0, 0.1)], [(2.0, 0.3)]], True),
        ("four clips, one with a NaN embedding", [[(1.0, 0.1)], [(1.0, 0.1)], [(1.0, 0.12)], [(1.0, 0.3)]], True),
    ]
    for si, (name, clips, as_list) in enumerate(specs):
        audios = [synth(100 + si * 10 + j, segs) for j, segs in enumerate(clips)]
        if "NaN" in name:
            audios[3][0] = 7.0
        for pre in (True, False):
            for ait in ("separate", "merge", "longest", "auto"):
                for clus in (True, False):
                    for as_emb_list in (True, False):
                        t = make_tasr(TargetASR)
                        # (the reference reads list inputs from files through ap.read_audio / audio_resample: give it the arrays)
                        if as_list:
                            store = {f"f{j}": a for j, a in enumerate(audios)}
                            t.ap.read_audio = lambda file_path, _s=store: (_s[file_path].copy(), SR)
                            t.ap.audio_resample = lambda audio_data, orig_sr, target_sr: (audio_data, target_sr)
                            arg = list(store)
                        else:
                            arg = audios[0]
                        r = t.get_target_embedding(target_audio=arg, is_preprocess=pre, is_cluster=clus, audio_input_type=ait,
                                                   output_embedding_list=as_emb_list)
                        r = [np.asarray(e, dtype=np.float64)[:3].tolist() for e in r] if isinstance(r, list) else np.asarray(r, dtype=np.float64)[:3].tolist()
                        out["get_target_embedding"].append({"spec": si, "clips": clips, "as_list": as_list, "nan_clip": 3 if "NaN" in name else -1,
                                                            "pre": pre, "type": ait, "cluster": clus, "list": as_emb_list, "out": r})
    # ---- sd_result_to_target_embedding
    td = object.__new__(TargetDiarization)
    td.ap = _AP()
    td.tasr = make_tasr(TargetASR)
    audio = synth(7, [(2.0, 0.1), (1.0, 0), (3.0, 0.3), (0.5, 0), (2.5, 0.2), (1.0, 0.05)])
    ref = td

    def refine(raw):
        """(sd_result, overlap_map) the way infer() builds them: the reference's own get_speaker_overlap + apply_od_result"""
        if not raw:
            return {}, []
        sd, omap = ref.apply_od_result({k: [tuple(x) for x in v] for k, v in raw.items()}, ref.get_speaker_overlap(raw))
        return sd, omap
    raw_cases = [
        ({}, ""),
        ({"0": [(0.0, 2.0), (6.5, 9.0)], "1": [(3.0, 6.0)]}, ""),
        ({"0": [(0.0, 2.0)], "1": [(3.0, 6.0), (6.5, 9.0)]}, ""),
        ({"0": [(0.0, 2.0), (6.5, 9.0)], "1": [(3.0, 6.0)]}, "1"),
        ({"0": [(0.0, 2.0), (6.5, 9.0)], "1": [(3.0, 6.0)]}, "7"),
        ({"0": [(0.0, 2.5), (6.5, 9.0)], "1": [(2.0, 6.0)]}, ""),
        ({"0": [(0.0, 0.3)], "1": [(0.5, 0.7)]}, ""),
        ({"0": [(3.0, 4.5)], "1": [(3.0, 4.5)]}, ""),
        ({"0": [(0.0, 4.0), (6.0, 9.5)], "1": [(3.0, 7.0)]}, ""),
    ]
    for raw, tspk in raw_cases:
        sd, omap = refine(raw)
        spk, emb = ref.sd_result_to_target_embedding(audio_data=audio.copy(), sampling_rate=SR, sd_result={k: list(v) for k, v in sd.items()},
                                                     overlap_map=[list(m) for m in omap], target_spk=tspk)
        out["sd_to_target_embedding"].append({"sd": sd, "omap": omap, "target_spk": tspk, "spk": spk, "emb": np.asarray(emb, dtype=np.float64)[:3].tolist()})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "n1_target_embedding.json"), "w"))
    print({k: len(v) for k, v in out.items()})

    # ---- sd_result_to_asr_audio: both branches
    cases = []
    audio = synth(11, [(12.0, 0.1)])
    raws = [
        {"0": [(0.0, 3.0), (6.0, 9.5)], "1": [(3.2, 5.8)]},
        {"0": [(0.0, 4.0), (6.0, 9.5)], "1": [(3.0, 7.0)]},
        {"0": [(0.0, 4.0)], "1": [(3.0, 7.013)], "2": [(6.5, 9.0)]},
        {"0": [(0.0, 4.0), (5.0, 7.7)], "1": [(3.0, 5.6), (7.0, 11.0)]},
        {},
    ]
    for raw in raws:
        sd, omap = refine(raw)
        for tspk, have_emb in (("0", True), ("1", True), ("", False), ("0", False)):
            if have_emb and len(sd) > 2:
                continue        # (the reference picks the noise speaker as list(set(...))[0]: with two candidates the choice follows the str hash seed)
            for with_ts in (True, False):
                for lang in ("zh", "en"):
                    td = object.__new__(TargetDiarization)
                    td.ap = _AP()
                    td.asr_engine = "paraformer"
                    td.tasr = make_tasr(TargetASR, with_timestamp=with_ts, language=lang)
                    td.tasr.multi_speakers_separate_asr = lambda asr_audio, target_embedding=None, threshold=0.0, is_output_asr=False, more_args={}: separate_stub(asr_audio)
                    emb = embed(audio) if have_emb else None
                    res = td.sd_result_to_asr_audio(audio_data=audio.copy(), sampling_rate=SR, sd_result={k: [tuple(x) for x in v] for k, v in sd.items()},
                                                    overlap_map=[list(m) for m in omap], target_spk=tspk, target_embedding=emb)
                    items = [{"speaker": r["speaker"], "timerange": [float(x) for x in r["timerange"]], "text": r["text"], "type": r["type"],
                              "audio": rle_sum(r["audio"])} for r in res]
                    cases.append({"sd": sd, "omap": omap, "target_spk": tspk, "have_emb": have_emb, "with_ts": with_ts, "lang": lang, "out": items})
    json.dump({"asr_audio": cases}, open(os.path.join(ROOT, "tests", "golden", "n1_asr_audio.json"), "w"))
    print({"asr_audio": len(cases)})


if __name__ == "__main__":
    main()
