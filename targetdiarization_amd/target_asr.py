"""TargetASR — the hot-path methods of the reference class of the same name (TargetASR.py) on the
MI355X path: `cosine_similarity` (:144-152, host arithmetic kept verbatim in meaning),
`get_speaker_embedding` (:155-163, the swap point `self.embedding[name](wav, output_emb=True)`),
and the batched form of `multi_speakers_separate_asr` (:571-655, via TargetDiarization).
Constructor keeps the reference's argument names (:28); model directories are accepted and
ignored — weights arrive as state dicts (`spk_state_dict`, …) because no checkpoint ships with
the reference.  Error convention: never raise out of a wrapper for a disabled
feature — print and degrade."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np


class TargetASR:
    def __init__(self, cuda_device: int = 0, embedding_model_dir: Union[str, list] = "iic/speech_eres2netv2w24s4ep4_sv_zh-cn_16k-common",
                 vad_model_dir: str = "iic/speech_fsmn_vad_zh-cn-16k-common-pytorch", diarization_model_dir: Optional[str] = None,
                 asr_model_dir: Union[str, list, None] = None, mdx_weights_file: Optional[str] = None,
                 separater_weights_folder: Optional[str] = None, restorer_weights_folder: Optional[str] = None, verbose_log: bool = False,
                 *, spk_state_dict=None):
        self.cuda_device = cuda_device
        self.verbose_log = verbose_log
        self.embedding = {}
        if spk_state_dict is not None:
            try:
                from .speaker import SpeakerEmbedder
                self.embedding["eres2netv2_large"] = SpeakerEmbedder(spk_state_dict, cuda_device=cuda_device)
            except Exception as e:                       # reference: init failure -> print, feature off (TargetASR.py:98-109)
                print(f"Load embedding model failed: {e}")

    # TargetASR.py:144-152
    @staticmethod
    def cosine_similarity(embedding_a: np.ndarray, embedding_b: np.ndarray) -> float:
        if np.all(embedding_a == 0.0) or np.all(embedding_b == 0.0):
            return 1.0
        norm_a = np.linalg.norm(embedding_a)
        norm_b = np.linalg.norm(embedding_b)
        similarity = np.dot(embedding_a, embedding_b) / (norm_a * norm_b)
        return float(max(0.0, min(similarity, 1.0)))

    # TargetASR.py:155-163
    def get_speaker_embedding(self, wav_file, embedding_model: str = "eres2netv2_large") -> np.ndarray:
        if embedding_model not in self.embedding:
            raise KeyError(f"embedding model {embedding_model!r} is not loaded")      # the reference raises KeyError here as well
        if isinstance(wav_file, np.ndarray):
            wav_file = wav_file.reshape(1, -1)
        elif isinstance(wav_file, str):
            wav_file = [wav_file]
        return self.embedding[embedding_model].get_speaker_embedding(wav_file)

    # TargetASR.py:491-505
    def is_same_person(self, existed_embeddings, target_embedding: np.ndarray, threshold: float = 0.4, verbose_result: bool = False):
        """cosine score of the MEAN of the known embeddings against the candidate, compared with the threshold"""
        known = [existed_embeddings] if isinstance(existed_embeddings, np.ndarray) else list(existed_embeddings)
        score = self.cosine_similarity(np.mean(known, axis=0), target_embedding)
        same = bool(score >= threshold)
        return {"is_same": same, "score": round(score, 3)} if verbose_result else same

    def get_speaker_embeddings(self, wavs, embedding_model: str = "eres2netv2_large") -> np.ndarray:
        """MI355X addition: one bucketed launch sequence for a list of clips (hot loops A/C)."""
        return self.embedding[embedding_model].get_speaker_embeddings(wavs)
