"""ASRProcessor — the paraformer branch of the reference's `asr_detection` (ASRProcessor.py:373-442)
around the MI355X encoder.  The swap point is `self.asr['paraformer'].generate(input=wav, hotword=…)`
(:424): the neural forward runs through `tdx_pfenc_*` (SANM encoder) and, when the state dict carries the
`predictor.*` / `decoder.*` tensors, `tdx_pfdec_*` (CIF predictor + NAR SANM decoder): tokens are the argmax
ids, mapped to text through `token_list` (funasr's tokens.json, absent here: "<id>" placeholders by default).
A custom `decoder(encoder_out[T',512]) -> {"text": str, "timestamp": [[start_ms, end_ms], ...]}` plug-in overrides it.
The post-processing of the result dicts (:427-437: ms -> s, token/timestamp pairing) and the
print-and-degrade behaviour for missing engines (:375-389) are the reference's."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


class ASRProcessor:
    def __init__(self, is_asr: bool = False, fast_asr: bool = False, asr_model_dir=None, is_vad: bool = False, vad_model_dir: str = "",
                 is_punc: bool = False, punc_model_dir: str = "", is_timestamp: bool = False, timestamp_model_dir: str = "",
                 is_emotion: bool = False, emotion_model_dir: str = "", is_diarization: bool = False, diarization_model_dir: str = "",
                 is_asr_api: bool = False, api_config_path: str = "", verbose_log: bool = True, cuda_device: int = 0, ap=None,
                 *, asr_state_dict=None, decoder: Optional[Callable] = None, punctuation: Optional[Callable] = None, token_list=None,
                 punc_state_dict=None, punc_vocab=None):
        self.is_asr = is_asr
        self.verbose_log = verbose_log
        self.decoder = decoder
        self.punctuation = punctuation
        if punctuation is None and punc_state_dict is not None:      # ASRProcessor.py:261-268: load, or print and switch the feature off
            try:
                from .punctuation import CTTransformer
                self.punctuation = CTTransformer(punc_state_dict, device=f"cuda:{cuda_device}", vocab=punc_vocab)
            except Exception as e:
                print(f"Failed to load punctuation restorer model: {e}")
        self.is_punc = self.punctuation is not None
        self.token_list = token_list
        self.asr = {}
        self.nar_decoder = None
        if is_asr and asr_state_dict is not None:
            try:
                from .paraformer import ParaformerDecoder, ParaformerEncoder
                enc_sd = {k: v for k, v in asr_state_dict.items() if k.startswith("encoder.")}
                self.asr["paraformer"] = ParaformerEncoder(enc_sd, device=f"cuda:{cuda_device}")
                if decoder is None and any(k.startswith("decoder.decoders.") for k in asr_state_dict):
                    self.nar_decoder = ParaformerDecoder(asr_state_dict, device=f"cuda:{cuda_device}")
                    self.decoder = self._nar_decode
            except Exception as e:                       # ASRProcessor.py:215-264: print, feature off
                print(f"Load ASR model failed: {e}")
                self.is_asr = False

    def _nar_decode(self, enc_out):
        """encoder output [T',512] -> {"text", "timestamp"} through the device CIF predictor + NAR decoder"""
        r = self.nar_decoder.decode(enc_out[None])[0]
        toks = [self.token_list[i] if self.token_list is not None and i < len(self.token_list) else f"<{i}>" for i in r["token_ids"]]
        return {"text": " ".join(toks), "timestamp": r["timestamp"], "token_ids": r["token_ids"]}

    def punctuation_restore(self, text):
        """ASRProcessor.punctuation_restore (:880-897): str or list of str; no model / empty text / a failing model -> the text unchanged"""
        if self.punctuation is None:
            if self.verbose_log:
                print("Modelscope punctuation restorer model hasn't been loaded. Return original result.")
            return text
        if not text:
            return text
        try:
            if isinstance(text, str):
                return self.punctuation(text)
            return [self.punctuation(t) if t else t for t in text]
        except Exception as e:
            print(f"Failed in punctuation_restore: {e}")
            return text

    @staticmethod
    def detect_language(text: str) -> str:
        """ASRProcessor.py:1013-1031: "en" only if Latin letters outnumber CJK ideographs, else "zh" """
        chinese_count = sum(1 for ch in text if "\u4e00" <= ch <= "\u9fff")
        english_count = sum(1 for ch in text if "a" <= ch.lower() <= "z")
        return "en" if english_count > chinese_count else "zh"

    @staticmethod
    def paraformer_postprocess(result_list: list, no_punc: bool, punctuation_restore, detect_language) -> list:
        """ASRProcessor.py:427-440 verbatim in meaning: timestamps ms -> s (3 decimals), paired with the
        space-separated tokens (padded with "" when there are fewer tokens than stamps)."""
        for i, result in enumerate(result_list):
            if "timestamp" in result:
                texts = result["text"].split(" ")
                value = [[round(point / 1000, 3) for point in clip] for clip in result["timestamp"]]
                if len(texts) < len(value):
                    texts.extend([""] * (len(value) - len(texts)))
                result_list[i]["timestamp"] = [(texts[j], value[j]) for j in range(len(value))]
                if not no_punc:
                    result_list[i]["text"] = punctuation_restore(result_list[i]["text"])
            if "language" not in result:
                result_list[i]["language"] = detect_language(result["text"])
        return result_list

    def asr_detection(self, wav_file, language: str = "auto", prompt: str = "", asr_engine: str = "paraformer", no_punc: bool = False,
                      output_text_only: bool = False, output_raw_result: bool = False):
        result_list = []
        if not self.is_asr or not self.asr or self.decoder is None:
            print("ASR models haven't been loaded. Return empty result.")
            return "" if output_text_only else result_list
        if asr_engine.lower() != "paraformer":
            asr_engine = list(self.asr.keys())[0]
        if isinstance(wav_file, (str, bytes)):
            raise ValueError("asr_detection: pass 16 kHz float32 numpy audio (file decoding is outside the MI355X hot path)")
        import torch
        wavs = wav_file if isinstance(wav_file, list) else [wav_file]
        enc = self.asr["paraformer"]
        for k, w in enumerate(wavs):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(w, dtype=np.float32).reshape(1, -1)))
            out = enc(x.to(enc.device))[0]
            res = dict(self.decoder(out))
            res.setdefault("key", f"clip_{k}")
            result_list.append(res)
        if output_raw_result:
            return result_list
        result_list = self.paraformer_postprocess(result_list, no_punc, self.punctuation_restore, self.detect_language)
        if output_text_only:
            return "".join(r["text"] for r in result_list)
        return result_list
