"""TargetASR — the hot-path methods of the reference class of the same name (TargetASR.py) on the
MI355X path: `cosine_similarity` (:144-152, host arithmetic kept verbatim in meaning),
`get_speaker_embedding` (:155-163, the swap point `self.embedding[name](wav, output_emb=True)`),
and the batched form of `multi_speakers_separate_asr` (:571-655, via TargetDiarization).
Constructor keeps the reference's argument names (:28); model directories are accepted and
ignored — weights arrive as state dicts (`spk_state_dict`, …) because no checkpoint ships with
the reference.  Error convention: never raise out of a wrapper for a disabled
feature — print and degrade."""
from __future__ import annotations

from typing import Callable, Optional, Union

import numpy as np


def _whole_clip(audio: np.ndarray):
    return [[0.0, round(audio.shape[0] / 16000.0, 3)]] if audio.shape[0] else []


def target_embedding_from_audio(target_audio, embed_many: Callable, vad: Callable, loudness_control: Callable, read_audio: Optional[Callable] = None,
                                is_preprocess: bool = True, is_cluster: bool = True, audio_input_type: str = "separate",
                                output_embedding_list: bool = True, verbose_log: bool = False):
    """TargetASR.get_target_embedding (TargetASR.py:166-258), step for step:
      1. inputs: one array, or a path / list of paths (read through `read_audio(path) -> 16 kHz mono float32`);
      2. is_preprocess: per input, FSMN-VAD ranges (`vad`) -> the speech pieces re-joined -> loudness control; an input without
         speech is dropped (printed);
      3. nothing left -> a 192-dim zero vector;
      4. audio_input_type: "merge" = all inputs concatenated, "longest" = the longest, "separate" = those >= 0.4 s, "auto" =
         longest if >= 3 s, else merge if <= 2 usable inputs, else separate; every clip capped at 30 s;
      5. one embedding per clip of >= 400 samples (`embed_many(list of clips) -> [n,192]`: ONE bucketed launch sequence here,
         one model call per clip in the reference), NaN embeddings skipped (printed);
      6. is_cluster and more than two embeddings: HDBSCAN(min_cluster_size=2, euclidean), noise (-1) dropped unless all is noise;
      7. output_embedding_list: the list; else zero vector / the single embedding / the mean."""
    from .clustering import hdbscan_labels
    sr = 16000
    if isinstance(target_audio, str):
        target_audio = [target_audio]
    if isinstance(target_audio, list):
        if read_audio is None:
            raise ValueError("get_target_embedding: path inputs need a read_audio(path) plug-in")
        audios = [np.asarray(read_audio(p), dtype=np.float32) for p in target_audio]
    else:
        audios = [np.asarray(target_audio).copy()]
    if is_preprocess:
        kept = []
        for a in audios:
            ranges = vad(a)
            if not ranges:
                print("Failed in func get_target_embedding: No VAD result.")
                continue
            pieces = [a[max(0, int(s * sr)):min(int(e * sr), a.shape[0])] for s, e in ranges]
            a = pieces[0].copy() if len(pieces) == 1 else np.concatenate(pieces)
            kept.append(loudness_control(a))
        audios = kept
    if not audios:
        print("Length of embedding_list shouldn't be zero. Return a 192-dim full-zero embedding.")
        return np.zeros([192], dtype=np.float32)
    longest = max(audios, key=lambda x: x.shape[0])
    normal = [a for a in audios if a.shape[0] >= int(sr * 0.4)]
    merged = audios[0] if len(audios) == 1 else np.concatenate(audios)
    if audio_input_type == "auto":
        audio_input_type = "longest" if longest.shape[0] >= 3.0 * sr else ("merge" if len(normal) <= 2 else "separate")
    audios = [merged] if audio_input_type == "merge" else [longest] if audio_input_type == "longest" else normal
    audios = [a[:30 * sr] for a in audios]
    clips = [a for a in audios if a.shape[0] >= 400]
    embs = np.asarray(embed_many(clips), dtype=np.float32).reshape(len(clips), -1) if clips else np.zeros((0, 192), np.float32)
    embedding_list = []
    for e in embs:
        if np.isnan(e).any():
            print("NaN value in embedding. Skip.")
            continue
        embedding_list.append(e)
    if verbose_log:
        print(f"Length of embedding_list before clustering: {len(embedding_list)}")
    if is_cluster and len(embedding_list) > 2:
        labels = hdbscan_labels(np.stack(embedding_list), min_cluster_size=2)
        valid = np.where(labels != -1)[0]
        if len(valid) > 0:
            embedding_list = [embedding_list[i] for i in valid]
    if output_embedding_list:
        return embedding_list
    if len(embedding_list) == 0:
        print("Length of embedding_list shouldn't be zero. Return a 192-dim full-zero embedding.")
        return np.zeros([192], dtype=np.float32)
    if len(embedding_list) == 1:
        return embedding_list[0]
    return np.mean(embedding_list, axis=0)


class TargetASR:
    def __init__(self, cuda_device: int = 0, embedding_model_dir: Union[str, list] = "iic/speech_eres2netv2w24s4ep4_sv_zh-cn_16k-common",
                 vad_model_dir: str = "iic/speech_fsmn_vad_zh-cn-16k-common-pytorch", diarization_model_dir: Optional[str] = None,
                 asr_model_dir: Union[str, list, None] = None, mdx_weights_file: Optional[str] = None,
                 separater_weights_folder: Optional[str] = None, restorer_weights_folder: Optional[str] = None, verbose_log: bool = False,
                 *, spk_state_dict=None, vad: Optional[Callable] = None, loudness_control: Optional[Callable] = None):
        """vad(audio) -> [[start_s, end_s], ...]: FunASR FSMN-VAD (`self.asrp.vad_detection`, third-party) as a plug-in, default = whole clip;
        loudness_control(audio) -> audio: AudioProcessor.audio_loudness_control (:417-429), default = the BS.1770 host meter"""
        self.cuda_device = cuda_device
        self.verbose_log = verbose_log
        self.vad = vad or _whole_clip
        self.loudness_control = loudness_control
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

    # TargetASR.py:166-258
    def get_target_embedding(self, target_audio, is_preprocess: bool = True, is_cluster: bool = True, embedding_model: str = "eres2netv2_large",
                             audio_input_type: str = "separate", output_embedding_list: bool = True):
        if embedding_model not in self.embedding:
            raise KeyError(f"embedding model {embedding_model!r} is not loaded")
        emb = self.embedding[embedding_model]
        lc = self.loudness_control
        if lc is None:
            from .target_diarization import TargetDiarization
            lc = TargetDiarization.audio_loudness_control
        return target_embedding_from_audio(target_audio, emb.get_speaker_embeddings, self.vad, lc, read_audio=getattr(emb, "_read_wav", None),
                                           is_preprocess=is_preprocess, is_cluster=is_cluster, audio_input_type=audio_input_type,
                                           output_embedding_list=output_embedding_list, verbose_log=self.verbose_log)

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
