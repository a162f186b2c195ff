"""Hot-path pipeline: separation -> speaker embeddings + cosine scoring -> ASR encoder, for
batches of utterances, sharded across GPUs by utterance with one all-gather of embeddings.

This is the slice of TargetDiarization.infer that is neural per-frame work (SURVEY.md §3.2
steps 4-6: hot loops A/B/C), restructured so that every model call is batched:
  * hot loop B (`multi_speakers_separate_asr` -> `separate_speaker`, TargetASR.py:571-655):
    all 10 s windows of all utterances, grouped by length, one launch sequence per group;
  * hot loops A/C (`get_speaker_embedding` + `cosine_similarity` per segment,
    TargetDiarization.py:581-629): all streams of equal length per launch, one scoring launch;
  * H3: the Paraformer encoder over <= 30 s segments of each stream (funasr's VAD segmentation
    and the CIF/decoder are outside the path — SURVEY §8f N2).
Multi-GPU (BASELINE config 5): utterance i -> rank i % P, full weight replica per rank, no
collective on the data path except `gather_embeddings` (RCCL all-gather over xGMI of each
rank's [n_i*2,192] block, zero-padded to the largest n_i).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n: int, rank: int, world: int):
    """static round-robin partition of n utterances"""
    return list(range(rank, n, world))


def gather_embeddings(local: torch.Tensor, n_total: int, rank: int, world: int, streams: int = 2) -> torch.Tensor:
    """local [n_i*streams, D] (utterances rank, rank+P, ... in order) -> [n_total*streams, D] in
    utterance order on every rank.  One all-gather of equal-size (zero-padded) blocks."""
    D = local.shape[1]
    if world == 1:
        return local
    max_n = (n_total + world - 1) // world
    buf = torch.zeros(max_n * streams, D, dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty(world * max_n * streams, D, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, max_n, streams, D)
    full = torch.empty(n_total, streams, D, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        full[idx] = out[r, : len(idx)]
    return full.view(n_total * streams, D)


class HotPath:
    def __init__(self, sep_state_dict, spk_state_dict=None, asr_state_dict=None, cuda_device: int = 0,
                 asr_segment: int = 480000, cmvn_shift=None, cmvn_scale=None):
        from .audio_processor import AudioProcessor
        self.device = torch.device(f"cuda:{cuda_device}")
        self.ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sep_state_dict, cuda_device=cuda_device, verbose_log=False)
        if not self.ap.is_separate_audio:
            from ._lib import TdxError
            raise TdxError("separator failed to initialise")
        self.spk = None
        self.asr = None
        if spk_state_dict is not None:
            from .speaker import SpeakerEmbedder
            self.spk = SpeakerEmbedder(spk_state_dict, cuda_device)
        if asr_state_dict is not None:
            from .paraformer import ParaformerEncoder
            self.asr = ParaformerEncoder(asr_state_dict, self.device, cmvn_shift=cmvn_shift, cmvn_scale=cmvn_scale)
        self.asr_segment = asr_segment

    # ---- H1 -------------------------------------------------------------------------------
    def separate(self, utts):
        """list of 1-D float32 arrays -> list of (spk1, spk2); windows of ALL utterances batched."""
        plans = [self.ap.window_plan(len(u), 160000) for u in utts]
        wins, owner = [], []
        for ui, (u, plan) in enumerate(zip(utts, plans)):
            for (s, e) in plan:
                wins.append(u[s:e].astype(np.float32, copy=True)); owner.append(ui)
        outs = self.ap.separate_windows_device(wins)
        pairs, k = [], 0
        for plan in plans:
            pairs.append(torch.cat(outs[k:k + len(plan)], dim=1))       # [2, n] on the device
            k += len(plan)
        return self.ap.louder_first(pairs)

    # ---- H2 -------------------------------------------------------------------------------
    def embed_streams(self, streams):
        return self.spk.get_speaker_embeddings(streams)

    # ---- H3 -------------------------------------------------------------------------------
    def encode_streams(self, streams):
        """Paraformer encoder outputs per stream: list of [T_i,512] arrays (segments <= 30 s,
        equal-length segments batched)."""
        segs, owner = [], []
        for si, s in enumerate(streams):
            for a in range(0, len(s), self.asr_segment):
                seg = s[a:a + self.asr_segment]
                if len(seg) >= 400:
                    segs.append(seg); owner.append(si)
        outs = [None] * len(segs)
        by_len = {}
        for i, s in enumerate(segs):
            by_len.setdefault(len(s), []).append(i)
        for n, idxs in by_len.items():
            for c in range(0, len(idxs), 16):
                chunk = idxs[c:c + 16]
                x = torch.from_numpy(np.stack([segs[i] for i in chunk]).astype(np.float32, copy=False)).to(self.device)
                y = self.asr(x).cpu().numpy()
                for j, i in enumerate(chunk):
                    outs[i] = y[j]
        res = [[] for _ in streams]
        for i, si in enumerate(owner):
            res[si].append(outs[i])
        return [np.concatenate(r, axis=0) if r else np.zeros((0, 512), np.float32) for r in res]

    # ---- whole path over a shard ------------------------------------------------------------
    def run(self, utts, target_embedding=None, rank: int = 0, world: int = 1, n_total: int | None = None, with_asr: bool = True):
        """utts: THIS rank's utterances (utterance i of the job lives on rank i % world).
        Returns dict with separated streams, all-gathered embeddings [n_total*2,192] (utterance
        order), cosine scores vs `target_embedding`, and encoder outputs of the local streams."""
        n_total = n_total if n_total is not None else len(utts)
        sep = self.separate(utts)
        out = {"streams": sep}
        if self.spk is not None:
            flat = [s for pair in sep for s in pair]
            local = torch.from_numpy(self.embed_streams(flat)).to(self.device) if flat else torch.zeros(0, 192, device=self.device)
            allemb = gather_embeddings(local, n_total, rank, world)
            out["embeddings"] = allemb.cpu().numpy()
            if target_embedding is not None:
                out["scores"] = self.spk.cosine_scores(out["embeddings"], target_embedding)
        if self.asr is not None and with_asr:
            out["encoder"] = self.encode_streams([s for pair in sep for s in pair])
        return out
