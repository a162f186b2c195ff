"""AudioProcessor — drop-in for the hot-path slice of the reference's AudioProcessor
(/root/reference/AudioProcessor.py): same constructor signature (:125-130), same device
convention (:205-221), same never-raise/degrade error convention (:171-202, :886-888) and
the same `separate_speaker` contract (:885-956).  What differs is *how* it runs: all windows
of equal length are batched into ONE device launch sequence (the reference loops with
batch = 1), on the MI355X-native MossFormer2 (libtdx.so).

`denoise_vocal` (:601-713) keeps the reference's outer 15 s / 1 s-margin chunker and the block plan of
`process_audio_chunk` (:605-643) around the device STFT / iSTFT (`frontend.BlockSTFT` = ConvTDFNet.stft/.istft
:82-120); the MDX net BODY (a third-party ONNX model in the reference: no weights, no onnxruntime here) is either the device
implementation of its architecture (`mdx_state_dict` -> `mdx.ConvTDFNetBody`, csrc/mdx.hip; parity unpinned) or any plug-in
`mdx_model(spec[n,4,dim_f,dim_t]) -> spec` on device tensors.  Resampling runs on the device polyphase resampler
(`ops.resample_poly`; the reference calls librosa: parity unpinned).

Out of scope here: enhancer, Apollo restorer, noisereduce (fast_mode), file I/O — those flags are
accepted and degrade to "module skipped" exactly like the reference does when a package is disabled.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from .loudness import integrated_loudness
from .separator import MossFormer2Separator


class AudioProcessor:
    def __init__(self,
                 is_denoise_vocal: bool = False, mdx_weights_file: str = "mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx",
                 is_enhance_vocal: bool = False, enhancer_weights_folder: str = "resemble_enhance/model_repo/enhancer_stage2",
                 is_separate_audio: bool = False, separater_weights_folder: str = "look2hear/checkpoints/TFGNet-Noise",
                 is_restore_audio: bool = False, restorer_weights_folder: str = "JusperLee/Apollo",
                 verbose_log: bool = True, cuda_device: int = 0, quality: int = 2,
                 separater_state_dict=None, mdx_model=None, mdx_dim_f: int = 3072, mdx_n_fft: int = 6144, silero_vad=None,
                 mdx_state_dict=None, mdx_args=None):
        """`separater_state_dict` (extension): an in-memory state_dict instead of
        `<separater_weights_folder>/best_model.pth` — no checkpoint ships with the reference.
        `mdx_model` (extension): the MDX net body as a callable on device tensors, spec[n,4,dim_f,256] -> spec (the
        reference runs the ONNX file through onnxruntime, AudioProcessor.py:231-233,630); without it the denoiser is off,
        like a failed `init_mdx_model` (:171-176).  `mdx_dim_f` / `mdx_n_fft`: the ONNX metadata the reference reads (:234-237)."""
        if is_denoise_vocal and mdx_model is None and mdx_state_dict is not None:
            # the ConvTDFNet body on the device (mdx.ConvTDFNetBody, csrc/mdx.hip) from a state dict with the PyTorch module names;
            # `mdx_args`: its geometry (L, l, g, k, bn, dim_f, dim_t, max_blocks_per_launch).  A failure prints and leaves the denoiser
            # off, like a failed `init_mdx_model` (AudioProcessor.py:171-176)
            try:
                from .mdx import ConvTDFNetBody
                mdx_model = ConvTDFNetBody(mdx_state_dict, f"cuda:{0 if cuda_device is None else cuda_device}", **(mdx_args or {}))
            except Exception as e:
                print(f"Failed to init MDX model: {e}")
        self.is_denoise_vocal = bool(is_denoise_vocal and mdx_model is not None)
        self.mdx_model = mdx_model
        self.silero_vad = silero_vad            # low_gpu_ram plug-in: silero_vad(audio[n] f32 @16 kHz) -> [[start, end], ...] in samples (:903-905)
        self.mdx_net = None
        self._mdx_geom = (mdx_n_fft, {1: 256, 2: 1024, 3: 2048}.get(quality, 1024), mdx_dim_f)        # :225-240
        if is_denoise_vocal and mdx_model is None:
            print("Failed to init MDX model: the MDX net body is a third-party ONNX model; pass mdx_model=callable(spec)->spec")
        self.is_enhance_vocal = False
        self.is_restore_audio = False
        self.is_separate_audio = is_separate_audio
        self.mdx_weights_file = mdx_weights_file.replace("\\", "/")
        self.enhancer_weights_folder = enhancer_weights_folder.replace("\\", "/")
        self.separater_weights_folder = separater_weights_folder.replace("\\", "/").rstrip("/")
        self.restorer_weights_folder = restorer_weights_folder.replace("\\", "/")
        self.verbose_log = verbose_log
        self.cuda_device = cuda_device
        self.quality = quality
        self.separater = None
        self.get_device()
        if self.is_separate_audio and (separater_state_dict is not None or os.path.isdir(self.separater_weights_folder)):
            try:
                self.init_separater_model(separater_state_dict)
            except Exception as e:                       # AudioProcessor.py:189-193
                print(f"Failed to init separater model: {e}")
                self.is_separate_audio = False
        else:
            self.is_separate_audio = False

    # AudioProcessor.py:205-221 — None -> auto, -1 -> cpu, n -> cuda:n
    def get_device(self):
        if self.cuda_device is None:
            self.device = "cuda" if torch.cuda.is_available() else "cpu"
        elif self.cuda_device == -1:
            self.device = "cpu"
        else:
            self.device = f"cuda:{self.cuda_device}"
            try:
                torch.cuda.set_device(self.device)
            except Exception as e:
                print(f"Failed to set CUDA device: {e}")

    # AudioProcessor.py:268-274 (config.yaml's model: block only carries constructor kwargs;
    # the MI355X build supports the constructor defaults + num_blocks)
    def init_separater_model(self, state_dict=None):
        if str(self.device) == "cpu":
            raise _lib.TdxError("the MI355X build has no CPU separator (cuda_device=-1 is not supported)")
        dev = "cuda:0" if self.device == "cuda" else self.device
        if state_dict is not None:
            self.separater = MossFormer2Separator(state_dict, device=dev)
        else:
            # config.yaml's `model:` block minus `_target_` = constructor kwargs (AudioProcessor.py:269-271)
            model_args = {}
            cfg_path = f"{self.separater_weights_folder}/config.yaml"
            if os.path.exists(cfg_path):
                import yaml
                with open(cfg_path) as fh:
                    model_args = dict((yaml.safe_load(fh) or {}).get("model", {}) or {})
                model_args.pop("_target_", None)
            self.separater = MossFormer2Separator.from_pretrain(f"{self.separater_weights_folder}/best_model.pth", device=dev, **model_args)
        self.separater.eval()

    # AudioProcessor.py:1123-1127
    def meter_loudness(self, audio_data: np.ndarray, sampling_rate: int):
        return round(integrated_loudness(audio_data, sampling_rate), 1)

    @staticmethod
    def window_plan(n: int, window_size: int = 160000, start: int = 0):
        """[start,end) windows for one VAD frame — AudioProcessor.py:920-935."""
        starts, ends = [], []
        k = n // window_size
        if k == 0:
            starts.append(start); ends.append(start + n)
        else:
            for j in range(k):
                starts.append(start + j * window_size); ends.append(start + (j + 1) * window_size)
            if n % window_size > 0:
                if n % window_size > window_size / 2:
                    starts.append(ends[-1]); ends.append(start + n)
                else:
                    ends[-1] = start + n
        return list(zip(starts, ends))

    def separate_windows_device(self, windows, max_batch: int = 32):
        """Run the separator on a list of 1-D float32 windows (host arrays or device tensors); windows of
        equal length share one batched forward (bit-compatible with B=1: the model has no cross-sample op and
        the reference never pads a batch, mossformer_block.py:485).  Returns a list of [2,T] DEVICE tensors."""
        out = [None] * len(windows)
        by_len = {}
        for i, w in enumerate(windows):
            by_len.setdefault(int(w.shape[0]), []).append(i)
        dev = self.separater.device
        for T, idxs in by_len.items():
            for c in range(0, len(idxs), max_batch):         # bound the workspace: <= max_batch windows / launch
                chunk = idxs[c:c + max_batch]
                if isinstance(windows[chunk[0]], torch.Tensor):
                    x = torch.stack([windows[i] for i in chunk]).to(dev, torch.float32)
                else:
                    x = torch.from_numpy(np.stack([windows[i] for i in chunk]).astype(np.float32, copy=False)).to(dev)
                y = self.separater(x)
                for j, i in enumerate(chunk):
                    out[i] = y[j]
        return out

    def separate_windows(self, windows):
        """as above, host arrays"""
        return [t.cpu().numpy() for t in self.separate_windows_device(windows)]

    def louder_first_device(self, pairs):
        """The reference's LUFS compare & swap (AudioProcessor.py:949-952) for a list of device stream pairs
        [2,n]: one BS.1770 launch per distinct length (tdx_loudness), values rounded to 0.1 like meter_loudness
        (:1123-1127).  Only the [k,2] loudness values cross to the host (the rounding rule is Python's); the
        streams stay on the device.  Returns the list of [2,n] device tensors, louder stream first."""
        from . import ops
        swap = [False] * len(pairs)
        by_len = {}
        for i, p in enumerate(pairs):
            by_len.setdefault(int(p.shape[1]), []).append(i)
        for n, idxs in by_len.items():
            if n < 6400:
                continue                                      # shorter than one 400 ms gating block: the reference's ValueError path
            l = ops.loudness(torch.cat([pairs[i] for i in idxs], 0), 16000).cpu().numpy().reshape(len(idxs), 2)
            for k, i in enumerate(idxs):
                swap[i] = round(float(l[k, 0]), 1) < round(float(l[k, 1]), 1)
        return [p.flip(0) if swap[i] else p for i, p in enumerate(pairs)]

    def louder_first(self, pairs):
        """host form: list of (spk1, spk2) numpy arrays"""
        res = []
        for p in self.louder_first_device(pairs):
            h = p.cpu().numpy()
            res.append((h[0], h[1]))
        return res

    # ---- AudioProcessor.py:350-383 (pure numpy in the reference; torch here, same arithmetic) ----
    @staticmethod
    def mono_to_stereo(audio):
        return audio.reshape(-1, 1).repeat(1, 2) if audio.ndim == 1 else audio

    @staticmethod
    def audio_to_mono(audio):
        return audio if audio.ndim == 1 else audio.double().mean(dim=1).float()      # np.average with unit weights, float64 accumulate

    def audio_resample(self, audio_data, orig_sr: int, target_sr: int, output_audio_only: bool = False):
        """:549-569 on the device (`ops.resample_poly`); numpy in -> numpy out, device tensor in -> device tensor out"""
        from . import ops
        if target_sr == orig_sr:
            return audio_data if output_audio_only else (audio_data, target_sr)
        is_np = isinstance(audio_data, np.ndarray)
        x = torch.from_numpy(np.ascontiguousarray(audio_data, dtype=np.float32)).to(self._dev()) if is_np else audio_data
        y = ops.resample_poly(x if x.ndim == 1 else x.t().contiguous(), orig_sr, target_sr)
        y = y if y.ndim == 1 else y.t().contiguous()
        if is_np:
            y = y.cpu().numpy().astype(np.float32)
        return y if output_audio_only else (y, target_sr)

    def _dev(self):
        return torch.device("cuda:0" if self.device == "cuda" else self.device)

    def init_mdx_model(self):
        """:224-241: the block STFT geometry (hop by `quality`, dim_t = 2**8 frames)"""
        from .frontend import BlockSTFT
        n_fft, hop, dim_f = self._mdx_geom
        self.mdx_net = BlockSTFT(n_fft=n_fft, hop=hop, dim_f=dim_f, dim_t=256, device=self._dev())

    @staticmethod
    def mdx_segments(total_samples: int, chunk_size: int, margin_size: int):
        """The outer chunker of denoise_vocal (:666-706) as index arithmetic: [(start, end, start_trim, end_trim)] — segment
        [start, end) of the audio is processed and [start_trim, len - end_trim) of the result is kept (end_trim None = to the end);
        the trims are clamped to half the PROCESSED segment's length, which equals the segment's length (:693-699)."""
        if total_samples <= chunk_size:
            return [(0, total_samples, 0, None)]
        margin_size = min(margin_size, chunk_size)
        segs, cursor, k = [], 0, 0
        while cursor < total_samples:
            start = max(0, cursor - (0 if k == 0 else margin_size))
            chunk_end = cursor + chunk_size
            last = chunk_end >= total_samples
            end = total_samples if last else min(chunk_end + margin_size, total_samples)
            segs.append([start, end])
            k += 1
            cursor += chunk_size
            if last:
                break
        out = []
        for i, (a, b) in enumerate(segs):
            n = b - a
            out.append((a, b, 0 if i == 0 else min(margin_size, n // 2), None if i == len(segs) - 1 else min(margin_size, n // 2)))
        return out

    # AudioProcessor.py:601-713
    def denoise_vocal(self, audio_data: np.ndarray, sampling_rate: int = 16000, fast_mode: bool = False):
        if not self.is_denoise_vocal:
            fast_mode = True
        if self.verbose_log:
            print("\nRunning module: denoise_vocal")
            print("Using method: noisereduce" if fast_mode else f"Using method: MDX-Net\nUsing model: {self.mdx_weights_file}")
        if fast_mode:
            print("denoise_vocal: noisereduce (fast_mode) is a third-party CPU package outside the MI355X hot path; audio returned unchanged")
            return audio_data
        if self.mdx_net is None:
            self.init_mdx_model()
        net = self.mdx_net
        dev = self._dev()
        x = torch.from_numpy(np.ascontiguousarray(audio_data, dtype=np.float32)).to(dev)
        new_sr = sampling_rate
        if sampling_rate != 44100:
            x, new_sr = self.audio_resample(x, sampling_rate, 44100)
        is_mono = x.ndim == 1
        if is_mono:
            x = self.mono_to_stereo(x)
        total = x.shape[0]
        segs = self.mdx_segments(total, int(15.0 * new_sr), int(1.0 * new_sr))
        # ---- process_audio_chunk (:605-643) for ALL segments in one batched STFT -> net -> iSTFT
        trim = net.n_fft // 2
        gen = net.chunk_size - 2 * trim
        blocks, plan = [], []
        for (a, b, _, _) in segs:
            n = b - a
            pad = (gen - (n % gen)) % gen
            mix = torch.cat((torch.zeros(2, trim, device=dev), x[a:b].t(), torch.zeros(2, pad + trim, device=dev)), dim=1)
            blk = mix.unfold(1, net.chunk_size, gen).permute(1, 0, 2)            # [nb, 2, chunk], stride gen (:613-617)
            plan.append((n, pad, blk.shape[0]))
            blocks.append(blk)
        waves = torch.cat(blocks, 0).contiguous()
        spec = net.stft(waves)
        pred = self.mdx_model(spec)
        wav = net.istft(pred)                                                   # [nb_total, 2, chunk]
        inst = "inst" in os.path.basename(self.mdx_weights_file).lower()
        outs, k = [], 0
        for (a, b, t0, t1), (n, pad, nb) in zip(segs, plan):
            o = wav[k:k + nb, :, trim:net.chunk_size - trim].transpose(0, 1).reshape(2, -1)
            k += nb
            o = o[:, :o.shape[1] - pad] if pad > 0 else o[:, :0]                   # `[:, :-pad]` is EMPTY when pad == 0 (:637)
            o = o.t()
            if inst:
                if o.shape[0] != n:                                             # numpy: (n,2) - (0,2) cannot broadcast
                    raise ValueError(f"operands could not be broadcast together with shapes ({n},2) ({o.shape[0]},2) ")
                o = torch.clamp(x[a:b] - o, -1.0, 1.0)
            else:
                o = torch.clamp(o, -1.0, 1.0)
            outs.append(o[t0:(o.shape[0] - t1) if t1 is not None else None] if len(segs) > 1 else o)
        out = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        if is_mono:
            out = self.audio_to_mono(out)
        if new_sr != sampling_rate:
            out, _ = self.audio_resample(out, new_sr, sampling_rate)
        return out.cpu().numpy().astype(np.float32)

    # AudioProcessor.py:885-956
    def separate_speaker(self, audio_data: np.ndarray, sampling_rate: int = 16000, low_gpu_ram: bool = False):
        if not self.is_separate_audio:
            print("\nSkip module: separate_speaker")
            return audio_data, audio_data
        orig_sr = sampling_rate
        if sampling_rate != 16000:                                   # :889-891
            audio_data, sampling_rate = self.audio_resample(np.asarray(audio_data, dtype=np.float32), sampling_rate, 16000)
        window_size, is_vad = (16000, True) if low_gpu_ram else (160000, False)      # :892-897
        if self.verbose_log:
            print("\nRunning module: separate_speaker")
            print(f"Window size: {window_size}")
            print(f"Use VAD: {is_vad}")
        if is_vad:
            # :903-905 — silero VAD (third-party) is a plug-in: silero_vad(audio[n] f32) -> [[start, end], ...] in samples
            if self.silero_vad is None:
                print("separate_speaker: low_gpu_ram needs a silero_vad plug-in; treating the whole clip as one speech frame")
                vad_frames = [[0, audio_data.shape[0]]]
            else:
                vad_frames = [[int(a), int(b)] for a, b in self.silero_vad(audio_data)]
        else:
            vad_frames = [[0, audio_data.shape[0]]]
        # :908-948 as a plan: the output is [zeros up to the first frame][frame 0][zeros between frames][frame 1]... (nothing
        # after the last frame: the reference's output is shorter than its input when the VAD cuts the tail); every window of
        # every frame goes through the separator, equal lengths batched
        pieces, wins, filled = [], [], 0
        for i, (f0, f1) in enumerate(vad_frames):
            if f0 > filled:
                gap = f0 if i == 0 else f0 - vad_frames[i - 1][1]
                pieces.append(("zeros", gap)); filled += gap
            for (a, b) in self.window_plan(f1 - f0, window_size, f0):
                if b - a < 16:
                    # MossFormer2's encoder needs >= kernel_size samples (the reference would raise inside conv1d)
                    pieces.append(("zeros", b - a))
                else:
                    pieces.append(("win", len(wins))); wins.append(audio_data[a:b].astype(np.float32, copy=True))
                filled += b - a
        if not wins:
            return audio_data, audio_data
        max_batch = max(1, min(512, (32 * 160000) // max(len(w) for w in wins)))
        outs = self.separate_windows_device(wins, max_batch)
        dev = outs[0].device
        cat = torch.cat([outs[k] if kind == "win" else torch.zeros(2, k, device=dev) for kind, k in pieces], dim=1)
        spk1, spk2 = self.louder_first([cat])[0]                     # louder stream first, :949-952 (metered on the device)
        if orig_sr != sampling_rate:                                 # :953-955
            spk1 = self.audio_resample(spk1, sampling_rate, orig_sr, output_audio_only=True)
            spk2 = self.audio_resample(spk2, sampling_rate, orig_sr, output_audio_only=True)
        return spk1, spk2
