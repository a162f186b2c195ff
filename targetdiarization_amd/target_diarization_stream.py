"""TargetDiarizationStream (TargetDiarizationStream.py:13-258): the streaming session of the reference — VAD-buffer
router, per-buffer overlap detection, separation / speaker check / ASR of each released buffer — on the MI355X hot path.

The arithmetic stages are the base class's (HotPath: MossFormer2, ERes2NetV2, Paraformer, BS.1770 loudness); the third-party
detectors are plug-ins with the reference's result formats:
    stream_vad(audio) -> [[start_s, end_s], ...]     silero-VAD `get_speech_timestamps(..., threshold 0.5, min_silence 100 ms)` (:132-133)
    vad(audio)        -> [[start_s, end_s], ...]     FunASR FSMN-VAD (`tasr.asrp.vad_detection`, :58, :135, :209)
    od_pipeline(audio)-> [(start, end, "SPEAKER_xx")] pyannote overlap detection (:179-184)
Defaults treat the whole clip as one speech range (no detector installed), like the base class.  Everything else — the five
buffer rules, the loudness gate, the bootstrap of the target embedding from the first released buffer, the choice of the longest
text, the "1"/"0" speaker labels and the running clock — follows the reference line by line; `prompt` (`use_asr_prompt`) is
accepted and recorded but the NAR decoder has no prompt input."""
from __future__ import annotations

import copy
import re
from typing import Callable, Generator, Optional

import numpy as np

from .loudness import integrated_loudness
from .target_diarization import TargetDiarization, _whole_clip_vad


class TargetDiarizationStream(TargetDiarization):
    def __init__(self, is_vad_buffer: bool = True, use_asr_prompt: bool = False, similarity_threshold: float = 0.4, vad_min_silence: float = 0.3,
                 max_buffer_duration: float = 30.0, loudness_diff_threshold: float = 12.0, *args, stream_vad: Optional[Callable] = None, **kwargs):
        super().__init__(*args, **kwargs)
        # configuration (the reference's attribute names: callers read and tune them between sessions)
        for name, value in (("is_vad_buffer", is_vad_buffer), ("use_asr_prompt", use_asr_prompt), ("similarity_threshold", similarity_threshold),
                            ("max_buffer_duration", max_buffer_duration), ("vad_min_silence", vad_min_silence),
                            ("loudness_diff_threshold", loudness_diff_threshold), ("stream_vad", stream_vad or _whole_clip_vad)):
            setattr(self, name, value)
        self._reset_session()

    def _reset_session(self):
        # session state (the reference keeps it on the model object, TargetDiarizationStream.py:27-34)
        self.vad_buffer, self.current_buffer_duration = [], 0.0
        self.current_time, self.prev_asr_text = 0.0, ""
        self.target_embedding, self.system_loudness_diff = None, 0.0

    def session(self) -> "TargetDiarizationStream":
        """A per-connection view of this model: shares the hot path, the plug-ins, the configuration and `gpu_lock`, owns its
        session state (buffer, clock, target embedding, loudness reference).  The reference keeps that state on the one shared
        model, so two WebSocket clients overwrite each other's buffers; the server here opens one session per connection.
        `model.infer_stream(...)` itself still works like the reference (state on the object: one stream at a time)."""
        s = copy.copy(self)
        s._reset_session()
        return s

    # ---- helpers --------------------------------------------------------------------------------
    def clear_vad_buffer(self):
        del self.vad_buffer[:]
        self.current_buffer_duration = 0.0

    def meter_loudness(self, audio: np.ndarray) -> float:
        """AudioProcessor.meter_loudness (:1123-1127): BS.1770 integrated loudness, rounded to 0.1 LU"""
        return round(float(integrated_loudness(np.asarray(audio, dtype=np.float32), 16000)), 1)

    def chunk_preprocess(self, audio_data: np.ndarray, sampling_rate: int) -> np.ndarray:
        """:37-41: mono -> float32 -> 16 kHz"""
        a = np.asarray(audio_data)
        if a.ndim > 1:
            a = a.mean(axis=-1)
        a = a.astype(np.float32) / 32768.0 if a.dtype == np.int16 else a.astype(np.float32)
        if sampling_rate != 16000:
            a, _ = self.hp.ap.audio_resample(a, sampling_rate, 16000)
        return a

    def is_same_person(self, a: np.ndarray, b: np.ndarray, threshold: float) -> bool:
        """TargetASR.is_same_person: cosine score (clipped to [0, 1], 1.0 for a zero vector) >= threshold"""
        return float(self.hp.spk.cosine_scores(np.asarray(a)[None], np.asarray(b))[0]) >= threshold

    def _embedding(self, audio: np.ndarray):
        if audio.shape[0] < 400 + 8 * 160:                   # ERes2NetV2 needs >= 9 fbank frames
            return np.zeros(192, dtype=np.float32)
        return self.hp.spk.get_speaker_embedding(audio)

    # ---- main method (:44-78) -------------------------------------------------------------------
    def infer_stream(self, audio_stream_generator: Generator, target_file=None, sampling_rate: int = 16000, is_single: bool = False,
                     output_target_audio: bool = False):
        self.current_time = 0.0
        self.clear_vad_buffer()
        if target_file is not None:
            if not isinstance(target_file, np.ndarray):
                raise ValueError("infer_stream(): pass the target clip as numpy audio (file decoding is outside the MI355X hot path)")
            target = target_file.copy()
            if target.shape[0] / sampling_rate >= 1.0:
                with self.gpu_lock:
                    self.system_loudness_diff = self.meter_loudness(self.chunk_preprocess(target, sampling_rate)) + 23.0
                    target = self.audio_preprocess(target, sampling_rate, stream_mode=True)
                    v = self.vad(target)
                    if v:
                        if v[-1][1] - v[0][0] < 4.0:
                            print("WARNING: The valid speaking duration of target audio is less than 4s. This may cause a bad result.")
                        target = self.split_audio_by_time(target, v[0][0], v[-1][1])
                    self.target_embedding = self._embedding(target)
        # (the lock is held while ONE chunk / buffer is processed, never while waiting for the next chunk or while the consumer
        # holds a yielded result)
        try:
            for pcm_chunk in audio_stream_generator:
                with self.gpu_lock:
                    pcm_chunk = self.chunk_preprocess(pcm_chunk, sampling_rate)
                    results = list(self.process_vad_chunk(pcm_chunk, is_single))
                for result in results:
                    asr_result, target_audio = self.asr_audio_parser([result], "1", output_target_audio)
                    yield "1", asr_result, target_audio
        finally:
            if self.vad_buffer:
                combined = np.concatenate(self.vad_buffer)
                with self.gpu_lock:
                    results = list(self.process_single_chunk(combined, is_single))
                self.clear_vad_buffer()
                for result in results:
                    asr_result, target_audio = self.asr_audio_parser([result], "1", output_target_audio)
                    yield "1", asr_result, target_audio

    # ---- VAD buffer router (:81-110) ------------------------------------------------------------
    def process_vad_chunk(self, pcm_chunk: np.ndarray, is_single: bool):
        if pcm_chunk is None or not pcm_chunk.shape[0]:
            return
        # loudness gate (armed once a target level is known): a chunk far below it enters the buffer as near-silence
        gated = self.system_loudness_diff != 0.0 and \
            self.meter_loudness(pcm_chunk) < -23.0 + self.system_loudness_diff - self.loudness_diff_threshold
        if gated:
            pcm_chunk = np.full_like(pcm_chunk, fill_value=0.00001, dtype=np.float32)
        self.vad_buffer.append(pcm_chunk)
        self.current_buffer_duration += round(pcm_chunk.shape[0] / 16000, 3)
        if not self.is_vad_buffer:              # no buffering: every audible chunk on its own (a gated one stays in the buffer, :95-96)
            release = None if gated else self.vad_buffer[-1]
        else:
            release = None if self.should_wait_for_next_chunk(is_silence=gated) else np.concatenate(self.vad_buffer)
        if release is None:
            return
        yield from self.process_single_chunk(release, is_single)
        self.clear_vad_buffer()

    # ---- the five rules (:113-171) --------------------------------------------------------------
    def should_wait_for_next_chunk(self, is_silence: bool = False) -> bool:
        def silence_gap(audio, vad_result):
            if not vad_result:
                return True
            return len(audio) / 16000 - vad_result[-1][-1] >= self.vad_min_silence
        if self.current_buffer_duration >= self.max_buffer_duration:          # rule 1: buffer length
            return False
        if not self.vad_buffer:
            return True
        combined = np.concatenate(self.vad_buffer)
        vad_result = self.stream_vad(combined)                                # rule 2: a silent chunk releases a buffer that ended in silence
        chunk_vad = self.vad(self.vad_buffer[-1])
        if is_silence:
            return not silence_gap(combined, vad_result)
        if not chunk_vad:                                                     # rule 3: no speech in this chunk: blank it, keep waiting
            self.vad_buffer[-1] = np.full_like(self.vad_buffer[-1], fill_value=0.00001, dtype=np.float32)
            return True
        if silence_gap(combined, vad_result):
            return False
        if len(self.vad_buffer) > 1:                                          # rule 4: a change of speaker releases the buffer
            prev = self._embedding(np.concatenate(self.vad_buffer[:-1]))
            cur = self._embedding(self.vad_buffer[-1])
            return self.is_same_person(prev, cur, self.similarity_threshold)
        return True                                                           # rule 5

    # ---- one released buffer (:174-258) ----------------------------------------------------------
    def process_single_chunk(self, pcm_chunk: np.ndarray, is_single: bool):
        od_raw = self.od_pipeline(pcm_chunk) if self.od_pipeline is not None else []
        od_result = self.od_result_parser(od_raw, is_single=is_single, output_overlap=True)
        result = self.asr_audio_streaming(pcm_chunk, is_overlap=bool(od_result))
        if result is not None:
            self.prev_asr_text = result["text"]
            yield result

    def _asr_text(self, audios):
        """Paraformer on each clip (HotPath H3 + device CIF / NAR decoder, or the `decoder` plug-in): list of texts"""
        if self.hp.asr is None:
            return ["" for _ in audios]
        if self.decoder is None and self.hp.dec is not None:
            _, dres = self.hp.encode_device(audios, decode=True)
            tok = lambda i: self.token_list[i] if self.token_list is not None and i < len(self.token_list) else f"<{i}>"
            return ["".join(tok(i) for seg in d for i in seg["token_ids"]) for d in dres]
        encs = self.hp.encode_streams(audios)
        return [self.decoder(e)[0] if self.decoder is not None else "" for e in encs]

    def asr_audio_streaming(self, audio_data: np.ndarray, is_overlap: bool = False, is_output_audio: bool = True):
        def remove_punc(text):
            return re.sub(r"[^\w\s]", "", text).lower().strip() if text else text
        duration = round(audio_data.shape[0] / 16000, 3)
        if duration < 0.4:
            return None
        self.current_time += duration                       # (the reference advances the clock first: the range below starts at the buffer's END, :208,:239)
        if self.target_embedding is None:                   # bootstrap: the first released buffer defines the target
            self.system_loudness_diff = self.meter_loudness(audio_data) + 23.0
            audio_data = self.audio_preprocess(audio_data, 16000, stream_mode=True)
            self.target_embedding = self._embedding(audio_data)
            is_overlap = False
        else:
            audio_data = self.audio_preprocess(audio_data, 16000, stream_mode=True)
        if self.meter_loudness(audio_data) < -23.0 + self.system_loudness_diff - self.loudness_diff_threshold:
            return None
        vad_result = self.vad(audio_data)
        if not vad_result:
            return None
        if is_overlap:                                      # TargetASR.multi_speakers_separate_asr: both separated streams, target first
            clips = [{"audio": a} for _, a in self._separate_overlaps(audio_data, [(0.0, duration)], self.target_embedding)[0]]
        else:                                               # single_speaker_asr
            clips = [{"audio": audio_data}]
        if not clips:
            return None
        for c, t in zip(clips, self._asr_text([c["audio"] for c in clips])):
            c["text"] = t
        if len(clips) > 1:
            clips = sorted(clips, key=lambda x: len(remove_punc(x["text"])), reverse=True)
        text = clips[0]["text"].strip()
        if not text:
            return None
        segment_audio = clips[0]["audio"] if is_overlap else audio_data
        is_target = self.is_same_person(self._embedding(segment_audio), self.target_embedding, self.similarity_threshold)
        return {"speaker": "1" if is_target else "0",
                "timerange": [self.current_time + vad_result[0][0], self.current_time + vad_result[-1][-1]],
                "text": text, "type": "overlap" if is_overlap else "single", "audio": segment_audio if is_output_audio else None}
