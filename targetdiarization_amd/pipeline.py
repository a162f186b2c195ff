"""Hot-path pipeline: separation -> speaker embeddings + cosine scoring -> ASR encoder, for
batches of utterances, sharded across GPUs by utterance with one all-gather of embeddings.

This is the slice of TargetDiarization.infer that is neural per-frame work (SURVEY.md §3.2
steps 4-6: hot loops A/B/C), restructured so that every model call is batched:
  * hot loop B (`multi_speakers_separate_asr` -> `separate_speaker`, TargetASR.py:571-655):
    all 10 s windows of all utterances, grouped by length, one launch sequence per group;
  * hot loops A/C (`get_speaker_embedding` + `cosine_similarity` per segment,
    TargetDiarization.py:581-629): all streams of equal length per launch, one scoring launch;
  * H3: the Paraformer encoder over <= 30 s segments of each stream (funasr's VAD segmentation
    is outside the path; the CIF predictor / decoder is `paraformer.ParaformerDecoder`, N2).
The separated streams stay ON THE DEVICE between the stages (H1 -> loudness swap -> H2 -> H3):
the reference crosses host<->device around every model call (AudioProcessor.py:940-946); here the
utterances go up once and the results come down once.
Multi-GPU (BASELINE config 5): utterance i -> rank i % P, full weight replica per rank, no
collective on the data path except `gather_embeddings` (RCCL all-gather over xGMI of each
rank's [n_i*2,192] block, zero-padded to the largest n_i).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n: int, rank: int, world: int):
    """static round-robin partition of n utterances"""
    return list(range(rank, n, world))


def cluster_embeddings(embeddings, min_cluster_size: int = 2):
    """The consumer of the gathered [n_total*k,192] block (BASELINE config 5: "all-gather of embeddings ... for clustering"): HDBSCAN*
    (min_cluster_size 2, euclidean — the reference's setting for speaker embeddings, TargetASR.py:238) on the host, labels [n] with -1
    = noise.  One D2H of n*768 bytes; O(n^2) float64 arithmetic for the few hundred utterances of a step (clustering.py)."""
    from .clustering import hdbscan_labels
    e = embeddings.detach().cpu().numpy() if isinstance(embeddings, torch.Tensor) else np.asarray(embeddings)
    return hdbscan_labels(e, min_cluster_size=min_cluster_size)


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
        n_r = len(shard_indices(n_total, r, world))
        full[r::world] = out[r, :n_r]          # (a strided slice, not an index list: no host-blocking index upload)
    return full.view(n_total * streams, D)


class HotPath:
    def __init__(self, sep_state_dict, spk_state_dict=None, asr_state_dict=None, cuda_device: int = 0,
                 asr_segment: int = 480000, cmvn_shift=None, cmvn_scale=None, windows_per_launch: int = 32,
                 asr_rows_per_launch: int = 32768, mdx_model=None, mdx_weights_file: str = "mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx",
                 mdx_state_dict=None, mdx_args=None):
        from .audio_processor import AudioProcessor
        self.device = torch.device(f"cuda:{cuda_device}")
        self.ap = AudioProcessor(is_separate_audio=True, separater_state_dict=sep_state_dict, cuda_device=cuda_device, verbose_log=False,
                                 is_denoise_vocal=mdx_model is not None or mdx_state_dict is not None, mdx_model=mdx_model,
                                 mdx_weights_file=mdx_weights_file, quality=3, mdx_state_dict=mdx_state_dict, mdx_args=mdx_args)
        if not self.ap.is_separate_audio:
            from ._lib import TdxError
            raise TdxError("separator failed to initialise")
        self.spk = None
        self.asr = None
        if spk_state_dict is not None:
            from .speaker import SpeakerEmbedder
            self.spk = SpeakerEmbedder(spk_state_dict, cuda_device)
        self.dec = None
        if asr_state_dict is not None:
            from .paraformer import ParaformerDecoder, ParaformerEncoder
            enc_sd = {k: v for k, v in asr_state_dict.items() if k.startswith("encoder.")}
            self.asr = ParaformerEncoder(enc_sd, self.device, cmvn_shift=cmvn_shift, cmvn_scale=cmvn_scale)
            if any(k.startswith("decoder.decoders.") for k in asr_state_dict):       # CIF predictor + NAR decoder (N2)
                self.dec = ParaformerDecoder(asr_state_dict, self.device)
        self.asr_segment = asr_segment
        self.windows_per_launch = windows_per_launch
        self.asr_rows_per_launch = asr_rows_per_launch
        self.overlap_seconds = float(os.environ.get("TDX_OVERLAP_SECONDS", "1e9"))     # run(): stage overlap on side streams up to this much audio per call
        # the two side streams of run() live as long as the HotPath (a fresh stream per call would make every call's buffers belong to
        # another stream of the allocator's pool); created on first use
        self._side_target = None
        self._side_asr = None

    def _side_stream(self, which: str):
        name = "_side_" + which
        if getattr(self, name) is None:
            setattr(self, name, torch.cuda.Stream(self.device))
        return getattr(self, name)

    def _dev(self, a):
        """host array or tensor -> 1-D float32 device tensor (the one H2D of the path)"""
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32).reshape(-1)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)).to(self.device)

    # ---- H1 -------------------------------------------------------------------------------
    def separate_device(self, utts):
        """list of 1-D waveforms (device tensors or host arrays) -> list of [2,n] DEVICE tensors, louder stream
        first (AudioProcessor.py:949-952); the windows of ALL utterances are batched by length."""
        utts = [self._dev(u) for u in utts]
        plans = [self.ap.window_plan(int(u.shape[0]), 160000) for u in utts]
        wins = [u[s:e] for u, plan in zip(utts, plans) for (s, e) in plan]
        outs = self.ap.separate_windows_device(wins, self.windows_per_launch)
        pairs, k = [], 0
        for plan in plans:
            pairs.append(outs[k] if len(plan) == 1 else torch.cat(outs[k:k + len(plan)], dim=1))       # [2, n]
            k += len(plan)
        return self.ap.louder_first_device(pairs)

    def separate(self, utts):
        """host form: list of (spk1, spk2) numpy arrays"""
        res = []
        for p in self.separate_device(utts):
            h = p.cpu().numpy()
            res.append((h[0], h[1]))
        return res

    # ---- H2 -------------------------------------------------------------------------------
    def embed_streams(self, streams):
        return self.spk.get_speaker_embeddings(streams)

    # ---- H3 -------------------------------------------------------------------------------
    def encode_device(self, streams, decode: bool = False):
        """Paraformer encoder outputs per stream: list of [T_i,512] DEVICE tensors (segments <= 30 s,
        equal-length segments batched, <= asr_rows_per_launch LFR frames per launch sequence).
        decode=True (needs the decoder weights): also the CIF + NAR decoder results of every segment, as
        (encoder outputs, [per stream: list of {"token_ids","scores","timestamp"} per segment, timestamps offset to the stream])."""
        streams = [self._dev(s) for s in streams]
        segs, owner = [], []
        for si, s in enumerate(streams):
            for a in range(0, int(s.shape[0]), self.asr_segment):
                seg = s[a:a + self.asr_segment]
                if seg.shape[0] >= 400:
                    segs.append(seg); owner.append(si)
        outs = [None] * len(segs)
        decs = [None] * len(segs)
        if decode and self.dec is None:
            from ._lib import TdxError
            raise TdxError("decode=True needs the predictor/decoder tensors in asr_state_dict")
        by_len = {}
        for i, s in enumerate(segs):
            by_len.setdefault(int(s.shape[0]), []).append(i)
        for n, idxs in by_len.items():
            rows = ((1 + (n - 400) // 160) + 5) // 6
            step = max(1, self.asr_rows_per_launch // max(rows, 1))
            for c in range(0, len(idxs), step):
                chunk = idxs[c:c + step]
                y = self.asr(torch.stack([segs[i] for i in chunk]))
                d = self.dec.decode(y) if decode else None
                for j, i in enumerate(chunk):
                    outs[i] = y[j]
                    if decode:
                        decs[i] = d[j]
        res = [[] for _ in streams]
        dres = [[] for _ in streams]
        for i, si in enumerate(owner):
            res[si].append(outs[i])
            if decode:
                off_ms = len(dres[si]) * self.asr_segment / 16.0            # segment start inside the stream, ms
                r = dict(decs[i]); r["timestamp"] = [[a + off_ms, b + off_ms] for a, b in r["timestamp"]]
                dres[si].append(r)
        enc = [(r[0] if len(r) == 1 else torch.cat(r, dim=0)) if r else torch.zeros(0, 512, device=self.device) for r in res]
        return (enc, dres) if decode else enc

    def encode_streams(self, streams):
        """host form: list of [T_i,512] numpy arrays"""
        return [t.cpu().numpy() for t in self.encode_device(streams)]

    # ---- whole path over a shard ------------------------------------------------------------
    def run(self, utts, target_embedding=None, rank: int = 0, world: int = 1, n_total: int | None = None, with_asr: bool = True,
            to_host: bool = True, embed_segment: int | None = None, target_clip=None, cluster: bool = False,
            punctuation=None, token_list=None):
        """utts: THIS rank's utterances (utterance i of the job lives on rank i % world), host arrays or
        device tensors.  Returns dict with the separated streams, the all-gathered embeddings [n_total*k,192]
        (utterance order), cosine scores vs `target_embedding`, and encoder outputs of the local streams.
        embed_segment: None = one embedding per separated stream (k = 2 per utterance); an integer cuts every
        stream into pieces of that many samples (the last piece keeps the remainder if it has >= 9 fbank frames)
        and embeds each piece — the per-window scoring of a long recording (BASELINE configs[2]/[3]); the all-gather
        then needs every utterance to produce the same number of pieces.
        to_host=False leaves every result on the device (the benchmark's resident-in-HBM boundary).
        cluster=True: HDBSCAN labels of the gathered embeddings (`cluster_embeddings`, host) as out["cluster_labels"] — read back after
        every launch of the step has been queued, so the host arithmetic runs under the tail of the device work.
        target_clip: the target speaker's sample as a waveform instead of `target_embedding` (TargetDiarization.infer has both clips
        at hand, :98-121): its embedding is computed on a side stream WHILE the mix is separated.
        The independent stages overlap on HIP streams: target embedding || separation, then speaker embeddings || Paraformer
        (one clip per call: −8 ms of 50; 1800 s per call: −1 %, the stages fill each other's tails; `overlap_seconds` /
        env TDX_OVERLAP_SECONDS bounds the audio per call up to which this is done, default: always)."""
        n_total = n_total if n_total is not None else len(utts)
        main = torch.cuda.current_stream(self.device)
        overlap = sum(int(u.shape[0]) for u in utts) <= self.overlap_seconds * 16000
        side0 = None
        if target_clip is not None and self.spk is not None:
            tclip = self._dev(target_clip)
            if overlap:
                side0 = self._side_stream("target")
                side0.wait_stream(main)
                with torch.cuda.stream(side0):
                    target_embedding = self.spk.embed_device([tclip])[0]
                target_embedding.record_stream(main)
            else:
                target_embedding = self.spk.embed_device([tclip])[0]
        sep = self.separate_device(utts)
        sep_done = torch.cuda.Event()
        sep_done.record(main)                  # what the ASR side stream waits for (NOT the embedding launches queued after it)
        if side0 is not None:
            main.wait_stream(side0)
        out = {"streams": sep}
        flat = [p[k] for p in sep for k in (0, 1)]
        if self.spk is not None:
            if embed_segment is None:
                clips, per = flat, 2
            else:
                clips = []
                for s in flat:
                    n = int(s.shape[0])
                    cuts = [(a, min(a + embed_segment, n)) for a in range(0, n, embed_segment)]
                    cuts = [(a, b) for (a, b) in cuts if b - a >= 400 + 8 * 160]
                    clips.extend(s[a:b] for (a, b) in cuts)
                per = (len(clips) // max(len(sep), 1)) if sep else 2
                if world > 1 and len(clips) != per * len(sep):
                    from ._lib import TdxError
                    raise TdxError("embed_segment with world > 1 needs utterances of equal length")
            local = self.spk.embed_device(clips) if clips else torch.zeros(0, 192, device=self.device)
            allemb = gather_embeddings(local, n_total, rank, world, streams=per)
            out["embeddings"] = allemb
            if target_clip is not None:
                out["target_embedding"] = target_embedding
            if target_embedding is not None:
                from . import ops
                tgt = target_embedding if isinstance(target_embedding, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(target_embedding, dtype=np.float32))
                out["scores"] = ops.cosine_scores(allemb, tgt.to(self.device, torch.float32)) if allemb.shape[0] else torch.zeros(0, device=self.device)
        if self.asr is not None and with_asr:
            # (issued after the embedding launches above: the decoder reads its token counts back, and that host wait then
            # overlaps embeddings already queued on the main stream)
            # (world > 1: no overlap — the multi-rank path has not run on hardware yet (the 8-GPU node is the driver's), so it keeps
            # the plainest stream picture: one compute stream next to RCCL's own; the overlap is worth <= 1 % at 200-utterance steps)
            side = None
            if overlap and world == 1:
                side = self._side_stream("asr")
                side.wait_event(sep_done)
            with torch.cuda.stream(side if side is not None else main):
                if self.dec is not None:
                    out["encoder"], out["asr"] = self.encode_device(flat, decode=True)      # tokens + timestamps per <= 30 s segment
                else:
                    out["encoder"] = self.encode_device(flat)
            if side is not None:
                for t in out["encoder"]:
                    t.record_stream(main)
                main.wait_stream(side)
        if cluster and "embeddings" in out:
            out["cluster_labels"] = cluster_embeddings(out["embeddings"])
        if to_host:                                   # the one D2H of the path
            # every result is copied into page-locked host memory with non-blocking copies on the main stream and the host waits ONCE
            # (pageable `.cpu()` copies are staged and block per tensor).  to_host="results": what infer() hands to its caller — the
            # separated streams, embeddings, scores, tokens; the encoder outputs (an intermediate once the device decoder has produced
            # the tokens) stay on the device.
            keep = []

            def down(t):
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                keep.append(h)
                return h
            streams_h = [down(p) for p in sep]
            small = {k: down(out[k]) for k in ("embeddings", "scores", "target_embedding") if k in out and isinstance(out[k], torch.Tensor)}
            enc_h = [down(t) for t in out["encoder"]] if "encoder" in out and to_host != "results" else None
            main.synchronize()
            out["streams"] = [(h[0].numpy(), h[1].numpy()) for h in streams_h]
            for k, h in small.items():
                out[k] = h.numpy()
            if enc_h is not None:
                out["encoder"] = [h.numpy() for h in enc_h]
            out["d2h_bytes"] = sum(h.numel() * h.element_size() for h in keep)
        if punctuation is not None and "asr" in out:
            # CT-Transformer punctuation of every recognised segment (ASRProcessor.punctuation_restore per chunk text, TargetDiarization.py:816):
            # all segments of all streams in lock step through the device model (punctuation.CTTransformer.inference_batch)
            tl = token_list
            texts = ["".join((tl[i] if tl is not None and i < len(tl) else f" <{i}>") for i in seg["token_ids"]) for stream in out["asr"] for seg in stream]
            res = punctuation.inference_batch(texts) if hasattr(punctuation, "inference_batch") else [(punctuation(t), None) for t in texts]
            k = 0
            for stream in out["asr"]:
                for seg in stream:
                    seg["text"] = res[k][0]; k += 1
        return out
