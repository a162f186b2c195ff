"""TargetDiarization — the orchestrator on top of the MI355X hot path.

Keeps the reference's constructor / `infer()` surface (TargetDiarization.py:26-41, :98-163) and the
order of its steps, with two differences that are the point of the rewrite:
  * every model call is batched: hot loop A (one embedding per non-overlap segment,
    :581-600) and hot loop C (recheck, :603-629) become ONE bucketed embedding launch + ONE
    cosine launch; hot loop B (separate every overlap segment, :760-775 ->
    TargetASR.multi_speakers_separate_asr :571-655) batches all windows of all overlap segments;
  * the third-party stages that are NOT kernels of the path are plug-ins, not hard imports:
      sd_pipeline(audio) -> {"text": [[start, end, label], ...]}        (CAM++ modelscope pipeline, :73,:126)
      od_pipeline(audio) -> [(start, end, "SPEAKER_xx"), ...]           (pyannote itertracks, :84,:132)
      vad(audio)         -> [[start, end], ...] in seconds               (FSMN-VAD, ASRProcessor.py:742)
      decoder(encoder_out[T',512]) -> (text, [(token, [s, e]), ...][, language])   (CIF + NAR decoder)
      punctuation(text)  -> text                                          (CT-Transformer, ASRProcessor.py:880-897)
    Defaults: one segment per utterance / no overlap detector / whole clip is speech / the device decoder / text unchanged.
The MDX denoiser body runs on the device when `mdx_state_dict` is given (mdx.ConvTDFNetBody, or any `mdx_model` callable); Apollo
restoration and non-.wav file decoding are outside the path.
"""
from __future__ import annotations

import math
import threading
from typing import Callable, Optional

import numpy as np

from . import intervals as iv
from .loudness import integrated_loudness
from .pipeline import HotPath


def _whole_clip_vad(audio: np.ndarray):
    return [[0.0, round(audio.shape[0] / 16000.0, 3)]] if audio.shape[0] else []


class TargetDiarization:
    def __init__(self, diarization_pipeline_dir: str = "iic/speech_campplus_speaker-diarization_common",
                 od_model_dir: str = "pyannote/speaker-diarization-3.1", mdx_weights_file: str = "mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx",
                 embedding_model_dir: str = "iic/speech_eres2netv2w24s4ep4_sv_zh-cn_16k-common",
                 vad_model_dir: str = "iic/speech_fsmn_vad_zh-cn-16k-common-pytorch",
                 asr_model_dir: str = "iic/speech_paraformer-large-vad-punc_asr_nat-zh-cn-16k-common-vocab8404-pytorch",
                 separater_weights_folder: str = "checkpoints/mossformer2-finetune", restorer_weights_folder: str = "JusperLee/Apollo",
                 asr_engine: str = "paraformer", pyannote_clustering_threshold: float = 0.0, target_similarity_threshold: float = 0.0,
                 cuda_device: int = 0, verbose_log: bool = False, *args,
                 sep_state_dict=None, spk_state_dict=None, asr_state_dict=None,
                 sd_pipeline: Optional[Callable] = None, od_pipeline: Optional[Callable] = None,
                 vad: Optional[Callable] = None, decoder: Optional[Callable] = None, mdx_model: Optional[Callable] = None, token_list=None,
                 punctuation: Optional[Callable] = None, mdx_state_dict=None, mdx_args=None,
                 punc_state_dict=None, punc_vocab=None, **kwargs):
        self.target_similarity_threshold = target_similarity_threshold
        self.asr_engine = asr_engine
        self.cuda_device = cuda_device
        self.verbose_log = verbose_log
        self.sd_pipeline = sd_pipeline
        self.od_pipeline = od_pipeline
        self.vad = vad or _whole_clip_vad
        self.decoder = decoder
        self.token_list = token_list          # funasr's tokens.json (absent here): ids -> text; None -> "<id>" placeholders
        self.punctuation = punctuation        # CT-Transformer punctuation restorer (ASRProcessor.punctuation_restore :880-897, third-party): text -> text
        if punctuation is None and punc_state_dict is not None:      # the device CT-Transformer (punctuation.py, tdx_punc_*)
            from .punctuation import CTTransformer
            self.punctuation = CTTransformer(punc_state_dict, device=f"cuda:{cuda_device}", vocab=punc_vocab)
        self.hp = HotPath(sep_state_dict, spk_state_dict, asr_state_dict, cuda_device=cuda_device, mdx_model=mdx_model,
                          mdx_weights_file=mdx_weights_file, mdx_state_dict=mdx_state_dict, mdx_args=mdx_args)
        # One model serves every request of the reference's server (main.py:42): REST handlers and WebSocket worker threads call
        # into it concurrently.  Each model object of the hot path serialises its own calls (_lib.HandleGuard); this lock keeps a
        # whole infer() — and, in stream mode, the processing of one released buffer — together, so that concurrent requests
        # interleave at request / buffer granularity and every result equals the one a lone caller gets.
        self.gpu_lock = threading.RLock()

    # ---- small DSP helpers kept from AudioProcessor ------------------------------------------
    @staticmethod
    def split_audio_by_time(audio, start_time, end_time, sampling_rate=16000):
        """AudioProcessor.split_audio_by_time (:740-747): clamped to the clip"""
        return audio[max(0, int(start_time * sampling_rate)):min(int(end_time * sampling_rate), audio.shape[0])].copy()

    @staticmethod
    def audio_loudness_control(audio, sampling_rate=16000, target_loudness=-23.0):
        """AudioProcessor.audio_loudness_control (:417-429): skip clips < 0.4 s, else gain to -23 LUFS"""
        if audio.shape[0] / sampling_rate < 0.4:
            return audio
        loud = round(integrated_loudness(audio, sampling_rate), 1)
        if not np.isfinite(loud):
            return audio
        return (audio * (10.0 ** ((target_loudness - loud) / 20.0))).astype(np.float32)

    def audio_preprocess(self, audio, sampling_rate: int = 16000, stream_mode: bool = False):
        """:166-182: mono -> float32 -> 16 kHz -> -23 LUFS -> denoise_vocal (or, in stream mode, the louder separated stream) ->
        -23 LUFS again; a failure inside the chain is printed and the audio of the last good step returned, like the reference"""
        audio = np.asarray(audio)
        if audio.ndim > 1:
            audio = audio.mean(axis=-1)
        if audio.dtype == np.int16:
            audio = audio.astype(np.float32) / 32768.0
        audio = audio.astype(np.float32)
        try:
            if sampling_rate != 16000:
                audio, sampling_rate = self.hp.ap.audio_resample(audio, sampling_rate, 16000)
            audio = self.audio_loudness_control(audio)
            if stream_mode:
                audio, _ = self.hp.ap.separate_speaker(audio, sampling_rate)
            elif self.hp.ap.is_denoise_vocal:
                audio = self.hp.ap.denoise_vocal(audio, sampling_rate)[: audio.shape[0]]
            else:
                return audio              # (denoiser off: the reference's fast_mode falls back to noisereduce, a third-party CPU package)
            audio = self.audio_loudness_control(audio)
        except Exception as e:
            print(f"Failed in func audio_preprocess: {e}")
        return audio

    # ---- segmentation parsers ----------------------------------------------------------------
    def od_result_parser(self, od_result, sd_result={}, is_single=False, output_overlap=True):
        """:230-246 with pyannote's itertracks replaced by (start, end, "SPEAKER_xx") tuples"""
        result = {}
        if not od_result:
            return result
        for start, end, label in od_result:
            key = "0" if is_single else str(int(str(label).split("_")[-1]))
            result.setdefault(key, []).append((round(start, 3), round(end, 3)))
        if is_single:
            result["0"] = iv.merge_timeranges(result["0"])
        if sd_result:
            result = iv.sd_key_matcher(sd_result, result)
        if output_overlap:
            result = iv.get_speaker_overlap(result)
        return result

    # ---- hot loop A, batched -----------------------------------------------------------------
    def target_embedding_to_target_spk(self, target_embedding, audio, sd_result, overlap_map):
        """:581-600: mean cosine score per speaker over its non-overlap segments, argmax."""
        if not sd_result:
            return ""
        sd_single = iv.subtract_overlap(sd_result, overlap_map)
        clips, owner = [], []
        for spk, ranges in sd_single.items():
            for (s, e) in ranges:
                c = self.split_audio_by_time(audio, s, e)
                if c.shape[0] >= 400 + 8 * 160:          # ERes2NetV2 needs >= 9 fbank frames
                    clips.append(c); owner.append(spk)
        if not clips:
            return ""
        scores = self.hp.spk.cosine_scores(self.hp.spk.get_speaker_embeddings(clips), target_embedding)
        best, best_spk = -1.0, ""
        for spk in sd_single:                            # first maximum in key order, like the reference's stable sort
            sc = [float(scores[i]) for i in range(len(clips)) if owner[i] == spk]
            if sc and sum(sc) / len(sc) > best:
                best, best_spk = sum(sc) / len(sc), spk
        return best_spk

    def _read_16k(self, src):
        """read_audio + mono + float32 + 16 kHz (what get_target_embedding does with a path, TargetASR.py:172-176)"""
        wav, sr = self.read_audio(src)
        a = np.asarray(wav)
        if a.ndim > 1:
            a = a.mean(axis=-1)
        a = a.astype(np.float32) / 32768.0 if a.dtype == np.int16 else a.astype(np.float32)
        if sr != 16000:
            a, _ = self.hp.ap.audio_resample(a, sr, 16000)
        return a

    def get_target_embedding(self, target_audio, is_preprocess: bool = True, is_cluster: bool = True, audio_input_type: str = "separate",
                             output_embedding_list: bool = True):
        """TargetASR.get_target_embedding (TargetASR.py:166-258) over the device embedder: VAD pieces re-joined, loudness control,
        the clip selection rule, ONE bucketed embedding launch for all clips, HDBSCAN(min_cluster_size=2) on the host, mean"""
        from .target_asr import target_embedding_from_audio
        return target_embedding_from_audio(target_audio, self.hp.spk.get_speaker_embeddings, self.vad, self.audio_loudness_control,
                                           read_audio=self._read_16k,
                                           is_preprocess=is_preprocess, is_cluster=is_cluster, audio_input_type=audio_input_type,
                                           output_embedding_list=output_embedding_list, verbose_log=self.verbose_log)

    def sd_result_to_target_embedding(self, audio, sd_result, overlap_map, target_spk: str = ""):
        """:551-578: no segmentation -> the clip itself; else the speaker with the longest total duration (first wins a tie) unless
        `target_spk` names one; its non-overlap ranges of >= 0.4 s concatenated (the whole clip if none) -> get_target_embedding"""
        if not sd_result:
            return "", self.get_target_embedding(audio, output_embedding_list=False)
        if not target_spk or target_spk not in sd_result:
            target_spk = list(sd_result)[0]
            best = sum(r[1] - r[0] for r in sd_result[target_spk])
            for spk, rs in sd_result.items():
                d = sum(r[1] - r[0] for r in rs)
                if d > best:
                    target_spk, best = spk, d
        if overlap_map:
            sd_result = iv.subtract_overlap(sd_result, overlap_map)
        # Quirk kept (:568-573): the reference assigns every cut back to `audio_data`, so the second and later ranges are cut out of
        # the PREVIOUS piece, not out of the recording (usually empty: a later range starts past the first piece's end); when no
        # range qualifies the embedding comes from whatever `audio_data` holds then — the whole recording.
        parts, cur = [], audio
        for (s, e) in sd_result[target_spk]:
            if e - s < 0.4:
                continue
            cur = self.split_audio_by_time(cur, s, e)
            parts.append(cur)
        src = np.concatenate(parts, axis=0) if parts else audio
        return target_spk, self.get_target_embedding(src, output_embedding_list=False)

    # ---- hot loop B, batched: separate every overlap segment of the target ---------------------
    def _separate_overlaps(self, audio, ranges, target_embedding):
        """TargetASR.multi_speakers_separate_asr(is_output_asr=False, threshold=0.0) for a list of
        segments: returns per segment [(timerange, audio) for target, then noise]."""
        clips = [self.split_audio_by_time(audio, s, e) for (s, e) in ranges]
        ok = [i for i, c in enumerate(clips) if c.shape[0] >= 400 + 8 * 160 and self.vad(c)]
        out = [[] for _ in clips]
        if not ok:
            return out
        seps = self.hp.separate([clips[i] for i in ok])
        embs = self.hp.spk.get_speaker_embeddings([s for pair in seps for s in pair])
        scores = self.hp.spk.cosine_scores(embs, target_embedding)
        for j, i in enumerate(ok):
            s1, s2 = float(scores[2 * j]), float(scores[2 * j + 1])
            a, b = seps[j]
            tgt, noise = (a, b) if s1 > s2 else (b, a)
            for stream in (tgt, noise):
                v = self.vad(stream)
                if v:
                    out[i].append(([v[0][0], v[-1][1]], stream))
        return out

    def sd_result_to_asr_audio(self, audio, sd_result, overlap_map, target_spk, target_embedding):
        """:716-820"""
        items = []
        if not sd_result:
            return items
        if overlap_map:
            single = iv.subtract_overlap(sd_result, overlap_map)
            overlap = iv.subtract_overlap(sd_result, overlap_map, reverse_output=True)
        else:
            single, overlap = sd_result, {}
        for spk, rs in single.items():
            for r in rs:
                items.append({"speaker": spk, "timerange": r, "text": "", "type": "single", "audio": self.split_audio_by_time(audio, r[0], r[1])})
        if not target_spk or target_embedding is None:
            for spk, rs in overlap.items():
                for r in rs:
                    items.append({"speaker": spk, "timerange": r, "text": "", "type": "overlap", "audio": self.split_audio_by_time(audio, r[0], r[1])})
        else:
            noise_spks = [k for k in sd_result if k != target_spk]
            for spk, rs in overlap.items():
                if spk in noise_spks:
                    continue
                for r, res in zip(rs, self._separate_overlaps(audio, rs, target_embedding)):
                    if not res:
                        continue
                    tr, ta = res[0]
                    items.append({"speaker": spk, "timerange": [round(r[0] + tr[0], 3), round(r[0] + tr[1], 3)], "text": "",
                                  "type": "overlap", "audio": self.audio_loudness_control(ta)})
                    if noise_spks and len(res) > 1:
                        nr, na = res[1]
                        items.append({"speaker": noise_spks[0], "timerange": [round(r[0] + nr[0], 3), round(r[0] + nr[1], 3)], "text": "",
                                      "type": "overlap", "audio": na})
        if not items:
            return items
        items.sort(key=lambda x: x["timerange"][0])
        # ---- H3 per speaker on the silence-padded timeline (:783-818)
        out = []
        spks = list(dict.fromkeys(it["speaker"] for it in items))
        timelines = [self.combine_audio_chunks(items, spk) for spk in spks]
        lines = [t for t in timelines if t is not None]
        encs, dres = [], None
        if self.hp.asr is not None and lines:
            if self.decoder is None and self.hp.dec is not None:       # the device CIF + NAR decoder (N2)
                encs, dres = self.hp.encode_device(lines, decode=True)
            else:
                encs = self.hp.encode_streams(lines)
        from .asr_processor import ASRProcessor
        punc = self.punctuation if self.punctuation is not None else (lambda t: t)
        k = 0
        for spk, tl in zip(spks, timelines):
            if tl is None:
                continue
            text, stamps, lang = "", [], []
            recognised = self.hp.asr is not None and (dres is not None or self.decoder is not None)
            if not recognised:                       # no recogniser loaded (asr_detection prints and returns nothing): the chunks keep empty texts
                out.extend(it for it in items if it["speaker"] == spk)
                continue
            if self.hp.asr is not None:
                enc = encs[k]
                if dres is not None:
                    tok = lambda i: self.token_list[i] if self.token_list is not None and i < len(self.token_list) else f"<{i}>"
                    stamps = [(tok(i), [round(a / 1000.0, 3), round(b / 1000.0, 3)]) for seg in dres[k] for i, (a, b) in zip(seg["token_ids"], seg["timestamp"])]
                    text = " ".join(t for t, _ in stamps)                 # (asr_detection's text: space-separated tokens, ASRProcessor.py:427-437)
                elif self.decoder is not None:
                    text, stamps, *lang = self.decoder(enc)             # (an optional third value: the recogniser's language tag)
                k += 1
            if not stamps:
                # :787-797 the recogniser returned no timestamps: ONE item per speaker over the whole span of the items, carrying the
                # speaker's timeline (the reference computes the punctuated text here and then stores the raw one — kept)
                out.append({"speaker": spk, "timerange": [items[0]["timerange"][0], items[-1]["timerange"][1]], "text": text, "type": "single",
                            "audio": tl})
                continue
            # :798-818 tokens go to the chunk whose range, widened to 0.1 s, contains their START; CJK languages join without a space
            joiner = "" if (lang[0] if lang else ASRProcessor.detect_language(text)) in ("zh", "ja", "ko", "yue") else " "
            for it in items:
                if it["speaker"] != spk:
                    continue
                lo, hi = math.floor(it["timerange"][0] * 10) / 10, math.ceil(it["timerange"][1] * 10) / 10
                it["text"] = punc("".join(joiner + tok for tok, (s, e) in stamps if lo <= s <= hi))
                out.append(it)
        out.sort(key=lambda x: x["timerange"][0])
        return out

    @staticmethod
    def combine_audio_chunks(items, speaker, sampling_rate=16000):
        """:822-838: the speaker's chunks on a silence-padded timeline"""
        parts, cursor = [], 0.0
        for it in items:
            if it["speaker"] == speaker:
                if cursor < it["timerange"][0]:
                    parts.append(np.zeros(int((it["timerange"][0] - cursor) * sampling_rate), dtype=np.float32))
                parts.append(it["audio"])
                cursor = it["timerange"][1]
        return np.concatenate(parts, axis=0) if parts else None

    # ---- hot loop C, batched -----------------------------------------------------------------
    def recheck_target_speaker(self, result, target_spk, target_embedding):
        """:603-629 (method "recheck_target"): score -1 = unchecked"""
        if not result:
            return []
        for it in result:
            it["score"] = -1.0
        if target_embedding is None or not self.target_similarity_threshold:
            return result
        idx = [i for i, it in enumerate(result) if it["speaker"] == target_spk and it.get("audio") is not None
               and it["audio"].shape[0] >= 400 + 8 * 160]
        if not idx:
            return result
        scores = self.hp.spk.cosine_scores(self.hp.spk.get_speaker_embeddings([result[i]["audio"] for i in idx]), target_embedding)
        for i, sc in zip(idx, scores):
            result[i]["score"] = round(float(sc), 3)
            if sc < self.target_similarity_threshold:
                result[i]["speaker"] = "-1"
        return result

    @staticmethod
    def asr_audio_parser(asr_result, target_spk, output_target_audio=True):
        """:841-873"""
        if not asr_result:
            return [], None
        result = []
        if not output_target_audio:
            for it in asr_result:
                it.pop("audio", None); result.append(it)
            return result, None
        asr_result.sort(key=lambda x: x["timerange"][0])
        parts, cursor = [], 0.0
        for it in asr_result:
            if it["speaker"] == target_spk:
                n = int((it["timerange"][0] - cursor) * 16000)
                if n > 0:
                    parts.append(np.zeros(n, dtype=np.float32))
                parts.append(it["audio"].astype(np.float32))
                cursor = it["timerange"][1]
            it.pop("audio", None)
            result.append(it)
        if cursor < asr_result[-1]["timerange"][1]:
            parts.append(np.zeros(int((asr_result[-1]["timerange"][1] - cursor) * 16000), dtype=np.float32))
        return result, (np.concatenate(parts, axis=0) if parts else None)

    # ---- main entry (:98-163) ------------------------------------------------------------------
    @staticmethod
    def read_audio(src, sampling_rate: int = 16000):
        """AudioProcessor.read_audio (:432-470) for the formats the stdlib decodes: a numpy array passes through with the caller's
        rate; a path / bytes / file object must be PCM16 .wav (the reference hands everything else to audioread / ffmpeg)"""
        if isinstance(src, np.ndarray):
            return src, sampling_rate
        import io
        import wave
        f = io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src
        with wave.open(f, "rb") as w:
            if w.getsampwidth() != 2:
                raise ValueError("read_audio(): only 16-bit PCM .wav is decoded here")
            x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
            ch, sr = w.getnchannels(), w.getframerate()
        return (x.reshape(-1, ch) if ch > 1 else x), sr

    def infer(self, wav_file, target_file=None, sampling_rate: int = 16000, is_single: bool = False, output_target_audio: bool = True):
        with self.gpu_lock:
            return self._infer(wav_file, target_file, sampling_rate, is_single, output_target_audio)

    def _infer(self, wav_file, target_file, sampling_rate, is_single, output_target_audio):
        wav, sr = self.read_audio(wav_file, sampling_rate)
        audio = self.audio_preprocess(wav, sr)
        target_embedding = None
        if target_file is not None:
            tw, tsr = self.read_audio(target_file, sampling_rate)
            tgt = self.audio_preprocess(tw, tsr)
            v = self.vad(tgt)
            if v:
                tgt = self.split_audio_by_time(tgt, v[0][0], v[-1][1])
                target_embedding = self.hp.spk.get_speaker_embedding(tgt)
        sd_result = None
        od_raw = None
        if audio.shape[0] / 16000 >= 30.0 or self.od_pipeline is None:
            rows = self.sd_pipeline(audio) if self.sd_pipeline is not None else {"text": [[0.0, round(audio.shape[0] / 16000, 3), 0]]}
            sd_result = iv.sd_result_parser(rows, is_single=is_single, combine_timerange=False)
        if not sd_result and self.od_pipeline is not None:
            od_raw = self.od_pipeline(audio)
            sd_result = self.od_result_parser(od_raw, is_single=is_single, output_overlap=False)
        overlap_map, target_spk = [], ""
        if not is_single:
            if od_raw is None and self.od_pipeline is not None:
                od_raw = self.od_pipeline(audio)
            od_result = self.od_result_parser(od_raw, sd_result=sd_result)
            sd_result, overlap_map = iv.apply_od_result(sd_result, od_result)
            if target_embedding is not None:
                target_spk = self.target_embedding_to_target_spk(target_embedding, audio, sd_result, overlap_map)
            else:
                target_spk, target_embedding = self.sd_result_to_target_embedding(audio, sd_result, overlap_map)
        asr_result = self.sd_result_to_asr_audio(audio, sd_result, overlap_map, target_spk, target_embedding)
        asr_result = self.recheck_target_speaker(asr_result, target_spk, target_embedding)
        asr_result, target_audio = self.asr_audio_parser(asr_result, target_spk, output_target_audio)
        return target_spk, asr_result, target_audio
