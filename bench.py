#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric: real-time factor of the full
infer() pipe, 16 kHz mono, 1/2/4/8 GPU).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic audio that is already resident in HBM
(utterances go up before the timed region; results stay on the device, DESIGN.md §5 gives the PCIe-inclusive
rate).  The pipe per step: MossFormer2 separation (H1, 10 s windows) -> louder-stream-first swap (BS.1770 on
the device) -> ERes2NetV2 embeddings of the separated streams + all-gather + cosine scores against a target
embedding (H2) -> Paraformer on <= 30 s segments of both streams (H3: SANM encoder, CIF predictor, NAR decoder,
argmax tokens + timestamps), punctuation = host pass-through.  24-block MossFormer2 / ERes2NetV2-w24s4ep4 /
50-layer encoder + 16-layer decoder at the reference's model sizes,
recipe (random-init) weights, fp32-accurate arithmetic.

  N = 1 (default)  BASELINE configs[3]: one 1800 s synthetic conversation = 180 windows, embeddings per 10 s
                   window of each stream, encoder on 120 x 30 s segments.
  N > 1            BASELINE configs[4]: the 1000 x 30 s utterance job, STRONG scaling: the job is processed in
                   batches of --utterances-per-step (200) utterances, one step = one batch, utterance i of a
                   batch on rank i % N, ONE all-gather (RCCL over xGMI) of the [n_i*2,192] embedding blocks per
                   step.  `python bench.py --gpus N` without a launcher starts the N ranks itself (children via
                   torch.distributed.run, before this process touches a GPU); under the driver's own
                   torch.distributed.run launch it finds RANK/WORLD_SIZE and just runs.
  --workload cfg2  BASELINE configs[1] (MossFormer2 only, 32 x 4 s) — the H1 kernel measurement of round 1.
  --workload cfg3  BASELINE configs[2] (600 s, H1 + H2 + cosine).
  --dry-run        N-rank plumbing on the CPU (gloo, kernels skipped, fabricated embeddings): tests only.

Output: ONE JSON line on rank 0 (driver contract) with
  roofline     — the dominant kernel (to_hidden+to_qk GEMM of MossFormer2, 24 launches per forward):
                 algorithmic FLOPs per launch / mean launch duration from HIP events recorded around that launch
                 on the forward's stream inside the timed region (tdx_mf2_profile_*).
  cpu_baseline — the oracle chain (CPU restatement of the reference: separator + fbank + ERes2NetV2 + SANM
                 encoder) timed on this host's cores on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
H3_PASSES = 3                    # f16 MFMA passes per fp32-accurate product (csrc/gemm_h3.hpp)
# Roofline of the dominant kernel: fp32-accurate products as 3 f16 MFMA passes over exponent-aligned split
# operands, so its ceiling in ALGORITHMIC FLOP/s is the dense f16 MFMA peak / 3 = 833.3 TFLOP/s.
PEAK_H3_TFLOPS = PEAK_F16_MFMA_TFLOPS / H3_PASSES
DTYPE = "f32 (split-f16 x3: fp32-accurate products from three f16 MFMA passes, fp32 accumulate)"
METRIC = "real-time factor (audio-sec/wall-sec) full infer() pipe, 16kHz mono, 1/2/4/8 GPU"
KERNEL = ("gemm_h3_kernel<TWOSEG, EpiHiddenConv> = to_hidden+to_qk: split-f16 x3 MFMA over pre-split planes; epilogue = ScaleNorm gain/bias, "
          "SiLU and the 17-tap depthwise convolution of v|u, written as K-major planes + fp32 u (TDX_FUSE_CONV=0: the round-2 pair of launches)")
FUSION_NOTE = ("since round 3 this launch also does the work of round 2's conv17<4> launch (SiLU + depthwise conv + split, VALU work in the epilogue, "
               "+6.7 % MFMA work for the 16 halo rows per 256-row tile): `achieved` divides the GEMM's algorithmic FLOPs by the time of BOTH; "
               "the two separate launches together ran 12.8 + 7.9 ms per 90-window launch of the default workload (profiles/r03_b_kernel_stats_cfg4_default_bench.csv) "
               "= 194 TFLOP/s on the same definition (DESIGN 4.1g, 5)")
WINDOW = 160000


def synth_mixtures(batch: int, n: int, seed: int) -> np.ndarray:
    """SURVEY.md §8d generator: s1+s2, s_k = 0.05*N(0,1) shaped by a 4 Hz raised-cosine AM envelope with
    random phase, clipped to [-1,1], f32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64) / 16000.0
    out = np.zeros((batch, n), dtype=np.float64)
    for b in range(batch):
        for _ in range(2):
            ph = rng.uniform(0, 2 * np.pi)
            env = 0.5 * (1.0 - np.cos(2 * np.pi * 4.0 * t + ph))
            out[b] += 0.05 * rng.standard_normal(n) * env
    return np.clip(out, -1.0, 1.0).astype(np.float32)


def synth_conversation(total_s: int, seed: int) -> np.ndarray:
    """ONE recording of two alternating "speakers" with 20 % overlap.  Speaker k = 0.05*N(0,1) shaped by a
    4 Hz raised-cosine AM envelope with its own random phase; turns of 4-8 s alternate, every turn runs 20 % of its length into the
    next speaker's turn (5 ms raised-cosine edges); clipped to [-1,1], f32.  Timing-neutral for the kernels (dense arithmetic)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = total_s * 16000
    t = np.arange(n, dtype=np.float64) / 16000.0
    out = np.zeros(n, dtype=np.float64)
    gates = np.zeros((2, n), dtype=np.float32)
    pos, k = 0.0, 0
    while pos < total_s:
        dur = rng.uniform(4.0, 8.0)
        a, b = int(pos * 16000), min(n, int((pos + 1.2 * dur) * 16000))
        gates[k, a:b] = 1.0
        pos += dur
        k ^= 1
    edge = 0.5 * (1.0 - np.cos(np.pi * np.arange(80) / 80.0)).astype(np.float32)
    for k in range(2):
        g = np.convolve(gates[k], edge / edge.sum(), mode="same")
        ph = rng.uniform(0, 2 * np.pi)
        env = 0.5 * (1.0 - np.cos(2 * np.pi * 4.0 * t + ph))
        out += 0.05 * rng.standard_normal(n) * env * g
    return np.clip(out, -1.0, 1.0).astype(np.float32)


def kernel_source_sha() -> str:
    """hash of the sources of the dominant kernel (gemm_h3.hpp + the launch site in mf2.hip): profiles/traffic.json records the hash
    its PMC passes were taken at, and the line only quotes those recorded counters while the hashes agree"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm_h3.hpp", "mf2.hip", "tdx_common.hpp"):
        h.update(open(os.path.join(ROOT, "targetdiarization_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def recorded_pmc(rows: int):
    """(traffic bytes for `rows` token rows, MFMA pipe utilisation, held clock in GHz, note) from the RECORDED PMC passes in
    profiles/traffic.json — not measured by this run; None when the file is absent or was recorded for other kernel sources"""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tf):
        return None, None, None, "no profiles/traffic.json"
    tj = json.load(open(tf))
    sha = tj.get("kernel_source_sha")
    if sha != kernel_source_sha():
        return None, None, None, f"profiles/traffic.json was recorded for kernel sources {sha}, the built library is {kernel_source_sha()}: not quoted"
    bpr = tj.get("gemm_to_hidden_hbm_bytes_per_token_row")
    return (bpr * rows if bpr else None), tj.get("mfma_pipe_utilisation"), tj.get("clock_GHz_held"), \
        f"recorded PMC passes ({tj.get('source')}): HBM-side bytes = (FETCH_SIZE x2 + WRITE_SIZE) per token row x rows of this run's full launch"


def host_cores() -> int:
    """CPU threads this process may actually use: min(affinity, cgroup quota), capped at 16
    (the GPU box gives a one-GPU job a 16-CPU share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def free_port() -> int:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# ----------------------------------------------------------------------------------------------
# CPU baselines (oracle = test infrastructure; only this leg of bench.py may import it)
# ----------------------------------------------------------------------------------------------
def cpu_baseline_h1(sd, wave_np, budget_windows: int):
    from oracle import mossformer2_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    with torch.no_grad():
        t0 = time.perf_counter()
        n_done = 0
        for i in range(budget_windows):
            orc.mossformer2_forward(torch.from_numpy(wave_np[i:i + 1]), sd)
            n_done += 1
            print(f"[bench] cpu_baseline window {n_done}: {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
            if time.perf_counter() - t0 > 20.0:
                break
        dt = time.perf_counter() - t0
    secs = n_done * wave_np.shape[1] / 16000.0
    return {"value": secs / dt, "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"{n_done} x [1,{wave_np.shape[1]}] windows of the same synthetic batch, B=1 per call "
                      f"(the reference's own batch size), oracle/mossformer2_oracle.py, torch CPU fp32, {dt:.1f}s wall"}


def cpu_baseline_pipe(sep_sd, spk_sd, asr_sd, window_np, with_asr: bool):
    """The oracle chain on ONE 10 s window of the workload, the way the reference runs it (batch 1 per call):
    MossFormer2 -> fbank + ERes2NetV2 on both streams + cosine -> (fbank/LFR + SANM encoder on both streams)."""
    from oracle import eres2netv2_oracle as eo
    from oracle import frontend_oracle as fo
    from oracle import mossformer2_oracle as orc
    from oracle import paraformer_oracle as po
    cores = host_cores()
    torch.set_num_threads(cores)
    stages = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        y = orc.mossformer2_forward(torch.from_numpy(window_np[None]), sep_sd)[0]
        t1 = time.perf_counter(); stages["separation"] = t1 - t0
        print(f"[bench] cpu_baseline separation: {t1 - t0:.1f}s", file=sys.stderr, flush=True)
        tgt = torch.randn(192)
        for k in range(2):
            e = eo.eres2netv2_forward(fo.sv_features(y[k])[None], spk_sd)[0]
            orc.cosine_similarity(e.numpy(), tgt.numpy())
        t2 = time.perf_counter(); stages["embedding"] = t2 - t1
        print(f"[bench] cpu_baseline embedding: {t2 - t1:.1f}s", file=sys.stderr, flush=True)
        if with_asr:
            for k in range(2):
                feats = fo.asr_features(y[k], torch.zeros(560), torch.ones(560))
                enc = po.sanm_encoder_forward(feats[None], asr_sd)
                po.paraformer_decode(enc, asr_sd)
        t3 = time.perf_counter(); stages["asr_encoder+decoder"] = t3 - t2
        dt = t3 - t0
    secs = window_np.shape[0] / 16000.0
    return {"value": secs / dt, "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"one [1,{window_np.shape[0]}] window of the same synthetic recording through the oracle chain "
                      f"(oracle/mossformer2_oracle.py -> frontend_oracle + eres2netv2_oracle on both streams + cosine"
                      + (" -> frontend_oracle + paraformer_oracle encoder + CIF + decoder on both streams" if with_asr else "")
                      + f"), batch 1 per call like the reference, torch CPU fp32, {dt:.1f}s wall",
            "stage_seconds": stages}


# ----------------------------------------------------------------------------------------------
# full pipe (cfg3 / cfg4 at one rank, cfg5 sharded)
# ----------------------------------------------------------------------------------------------
def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def pipe_bench(args):
    import torch.distributed as dist
    from targetdiarization_amd.pipeline import HotPath, shard_indices
    from targetdiarization_amd.weights import (recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict,
                                               recipe_paraformer_state_dict, recipe_state_dict)
    rank, local_rank, world = dist_env()
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    wl = args.workload
    with_asr = wl != "cfg3"
    sep_sd, spk_sd = recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0)
    asr_sd = None
    if with_asr:                                  # 50-layer SANM encoder + CIF predictor + 16-layer NAR decoder (vocabulary 8404)
        asr_sd = dict(recipe_paraformer_state_dict(0, 50)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 16))
    hp = HotPath(sep_sd, spk_sd, asr_sd, cuda_device=local_rank, windows_per_launch=args.windows_per_launch,
                 asr_rows_per_launch=args.asr_rows_per_launch)
    hp.spk.max_batch_frames = args.embed_frames_per_launch
    punc = token_list = None
    if with_asr and wl == "cfg4" and not args.no_punc:
        # configs[3] names the CT-Transformer punctuation stage: the device model (punctuation.CTTransformer, funasr's geometry and its
        # 272 727-word embedding table, recipe weights) restores the punctuation of every recognised 30 s segment inside the timed step;
        # token id k of the recogniser reads as the CJK ideograph U+4E00 + k, word w has the punctuation-vocabulary id 1 + (ord(w) - 0x4E00)
        from targetdiarization_amd.punctuation import CTTransformer
        from targetdiarization_amd.weights import recipe_punc_state_dict
        punc = CTTransformer(recipe_punc_state_dict(0, num_blocks=4, vocab=272727), device=f"cuda:{local_rank}", vocab=lambda w: 1 + (ord(w[0]) - 0x4E00) % 272726)
        token_list = [chr(0x4E00 + i) for i in range(8404)]
    target = torch.from_numpy(np.random.default_rng(5).standard_normal(192).astype(np.float32)).to(dev)

    if wl in ("cfg3", "cfg4"):
        total_s = 600 if wl == "cfg3" else 1800
        nwin = total_s // 10
        rec_np = synth_conversation(total_s, seed=3 if wl == "cfg3" else 4)                     # one long recording
        batches = [[torch.from_numpy(rec_np).to(dev)]]
        n_step_total = 1
        audio_s_per_step = float(total_s)
        embed_segment = WINDOW
        windows_local = nwin
        desc = (f"BASELINE {'configs[2]' if wl == 'cfg3' else 'configs[3]'}: {total_s} s synthetic conversation (two alternating speakers, 20 % overlap) = {nwin} x 10 s windows: "
                f"MossFormer2 ({args.windows_per_launch} windows per launch) -> loudness swap -> ERes2NetV2 on every 10 s window of both streams "
                "+ cosine vs a target embedding" + (" -> Paraformer (SANM encoder, CIF predictor, NAR decoder, argmax tokens + timestamps) on 30 s segments of both streams -> "
                   + ("CT-Transformer punctuation of every segment's text on the device (all segments in lock step)" if punc is not None else "punctuation pass-through") if with_asr else "")
                + "; recipe weights; input resident in HBM, results (separated streams, embeddings, scores, tokens) copied to the host inside the timed region")
        sample_window = rec_np[:WINDOW]
    else:   # cfg5
        n_job, per_step = args.utterances, args.utterances_per_step
        n_batches = max(1, n_job // per_step)
        n_used = min(n_batches, args.steps + args.warmup)
        batches = []
        for b in range(n_used):
            mine = shard_indices(per_step, rank, world)
            batches.append([torch.from_numpy(synth_mixtures(1, 3 * WINDOW, seed=5 + b * per_step + i)[0]).to(dev) for i in mine])
        n_step_total = per_step
        audio_s_per_step = per_step * 30.0
        embed_segment = None
        windows_local = 3 * len(batches[0])
        desc = (f"BASELINE configs[4]: the {n_job} x 30 s utterance job in batches of {per_step} utterances (one step = one batch, "
                f"utterance i of a batch on rank i % {world}; strong scaling: the batch does not grow with N): MossFormer2 (3 windows per "
                "utterance) -> loudness swap -> ERes2NetV2 on both streams -> all-gather of the [n_i*2,192] embedding blocks (RCCL) -> cosine "
                "scores + HDBSCAN clustering of the gathered embeddings on rank 0 -> Paraformer (SANM encoder, CIF predictor, NAR decoder) on both "
                "streams; recipe weights; inputs resident in HBM, results copied to the host inside the timed region")
        sample_window = None

    def step(k, inputs=None):
        utts = inputs if inputs is not None else batches[k % len(batches)]
        # to_host="results": the D2H of the results is INSIDE the timed region ("... to end of D2H")
        return hp.run(utts, target, rank, world, n_step_total, with_asr=with_asr, to_host="results", embed_segment=embed_segment,
                      cluster=(wl == "cfg5" and rank == 0), punctuation=punc, token_list=token_list)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        out = step(k)
        if rank == 0:
            print(f"[bench] warmup {k + 1}/{args.warmup}", file=sys.stderr, flush=True)
    sep = hp.ap.separater
    launches_per_step = 24 * ((windows_local + args.windows_per_launch - 1) // args.windows_per_launch)
    sep.profile_enable(launches_per_step * args.steps)
    barrier()
    t0 = time.perf_counter()
    step_s = []
    for k in range(args.steps):
        t1 = time.perf_counter()
        out = step(args.warmup + k)                 # (ends with the host's wait for the D2H copies)
        step_s.append(time.perf_counter() - t1)
    barrier()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_launches = sep.profile_collect()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert np.isfinite(out["embeddings"]).all() and np.isfinite(out["scores"]).all()
    assert out["embeddings"].shape[0] == out["scores"].shape[0]
    d2h_bytes = out.get("d2h_bytes", 0)
    # PCIe-inclusive figure (never `value`): the same step with its input coming from page-locked HOST memory (H2D + step + D2H)
    pcie = None
    if rank == 0 or use_dist:
        host_in = [u.cpu().pin_memory() for u in batches[0]]
        barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            step(0, inputs=[h.to(dev, non_blocking=True) for h in host_in])
        barrier()
        pcie_dt = (time.perf_counter() - t1) / 2
        pcie = {"ms_per_step": pcie_dt * 1e3, "value": audio_s_per_step / pcie_dt, "h2d_bytes_rank0": sum(h.numel() * 4 for h in host_in),
                "d2h_bytes_rank0": d2h_bytes, "note": "input uploaded from pinned host memory inside the timed step; 2 steps"}

    # one extra, untimed, instrumented step: per-stage times (events around the stages) and FLOP accounting
    stage_ms = None
    if rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        utts = batches[0]
        torch.cuda.synchronize(dev)
        ev[0].record()
        pairs = hp.separate_device(utts)
        ev[1].record()
        flat = [p[k] for p in pairs for k in (0, 1)]
        if embed_segment:
            clips = [s[a:a + embed_segment] for s in flat for a in range(0, int(s.shape[0]), embed_segment)]
        else:
            clips = flat
        from targetdiarization_amd import ops
        ops.cosine_scores(hp.spk.embed_device(clips), target)
        ev[2].record()
        dres = None
        if with_asr:
            _, dres = hp.encode_device(flat, decode=True)
        ev[3].record()
        torch.cuda.synchronize(dev)
        stage_ms = {"separation+loudness": ev[0].elapsed_time(ev[1]), "embedding+cosine": ev[1].elapsed_time(ev[2]),
                    "asr_encoder+decoder": ev[2].elapsed_time(ev[3]) if with_asr else 0.0}
        if punc is not None and dres is not None:             # host wall time: device forwards + the host's window / cache / text logic
            texts = ["".join(token_list[i] for i in seg["token_ids"]) for stream in dres for seg in stream]
            tp = time.perf_counter()
            pres = punc.inference_batch(texts)
            stage_ms["punctuation (wall)"] = (time.perf_counter() - tp) * 1e3
            stage_ms["punctuation_texts"] = len(texts)
            stage_ms["punctuation_words"] = sum(len(r[1]) for r in pres)
    if use_dist:
        dist.barrier()

    if rank == 0:
        value = audio_s_per_step * args.steps / dt
        print(f"[bench] gpu: {dt / args.steps * 1e3:.1f} ms/step, RTF {value:.1f}", file=sys.stderr, flush=True)
        S = (WINDOW - 16) // 8 + 1
        F = 1 + (WINDOW - 400) // 160
        # algorithmic FLOPs of the step on THIS rank (rank 0's shard; shards differ by at most one utterance)
        n_win_rank = windows_local
        fl = {"separation": sep.flops(1, WINDOW) * n_win_rank}
        if embed_segment:
            fl["embedding"] = hp.spk.model.flops(2 * n_win_rank, F)
            fl["asr_encoder"] = hp.asr.flops(2 * n_win_rank // 3, ((1 + (3 * WINDOW - 400) // 160) + 5) // 6) if with_asr else 0.0
        else:
            F30 = 1 + (3 * WINDOW - 400) // 160
            fl["embedding"] = hp.spk.model.flops(2 * n_win_rank // 3, F30)
            fl["asr_encoder"] = hp.asr.flops(2 * n_win_rank // 3, (F30 + 5) // 6) if with_asr else 0.0
        # dominant kernel: every timed launch covers min(windows_per_launch, remaining) windows; report per FULL launch
        rows_total = args.steps * n_win_rank * S              # token rows through the GEMM over the timed region (x 24 layers)
        gemm_flops_total = 2.0 * rows_total * 512 * 2176 * 24
        ach = gemm_flops_total / (gemm_ms * 1e-3) / 1e12 if gemm_launches else None
        full_rows = min(args.windows_per_launch, n_win_rank) * S
        traffic, mfma_util, clock_held, traffic_note = recorded_pmc(full_rows)
        n1_ref = None
        if wl == "cfg5" and world > 1:            # the same workload on ONE GPU, measured this round (the N = 1 default line is configs[3])
            f = os.path.join(ROOT, "profiles", "r03_bench_cfg5_n1.json")
            if os.path.exists(f):
                j = json.loads(open(f).read().strip().splitlines()[-1])
                n1_ref = {"value": j["value"], "ms_per_step": j["ms_per_step"], "source": "profiles/r03_bench_cfg5_n1.json (`bench.py --workload cfg5`, 1 x MI355X, this round)"}
        line = {
            "metric": METRIC, "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": float(np.median(step_s)) * 1e3, "higher_is_better": True,
            "scaling": "strong" if wl == "cfg5" else "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": desc, "audio_seconds_per_step": audio_s_per_step, "windows_per_launch": args.windows_per_launch,
                       "parallelism": f"{world} rank(s), one process per GPU, utterances sharded round-robin, full weight replica per rank"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_H3_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ach / PEAK_H3_TFLOPS) if ach else None, "traffic": traffic, "kernel": KERNEL, "fusion_note": FUSION_NOTE,
                         "peak_note": "algorithmic fp32-accurate FLOP/s ceiling of the kernel = dense f16 MFMA peak 2500 / 3 passes",
                         "mfma_pipe_executed_tflops": (ach * H3_PASSES) if ach else None,
                         "launches_timed": gemm_launches, "ms_per_launch": gemm_ms / max(gemm_launches, 1),
                         "algorithmic_flops_per_launch": 2.0 * full_rows * 512 * 2176,
                         "token_rows_per_full_launch": full_rows,
                         "pmc_recorded_mfma_pipe_utilisation": mfma_util,
                         "pmc_recorded_clock_GHz": clock_held,     # the chip is power-limited under this load: the peak above assumes 2.4 GHz
                         "frac_of_peak_at_recorded_clock": (ach / (PEAK_H3_TFLOPS * clock_held / 2.4)) if (ach and clock_held) else None,      # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x busy cycles), recorded pass
                         "traffic_source": "pmc_recorded (not measured by this run)" if traffic is not None else None,
                         "traffic_note": traffic_note,
                         "whole_path_tflops_per_gpu": sum(fl.values()) * args.steps / dt / 1e12,
                         "whole_path_frac": sum(fl.values()) * args.steps / dt / 1e12 / PEAK_H3_TFLOPS},
            "timed_region": "inputs resident in HBM -> all launches -> D2H of the results (streams, embeddings, scores, tokens) into pinned host memory",
            "pcie_inclusive": pcie,
            "same_workload_n1": n1_ref,
            "stage_ms_per_step_rank0": stage_ms,
            "peak_device_memory_gib_rank0": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "stage_tflops_rank0": {k: (fl[a] / (stage_ms[k] * 1e-3) / 1e12 if stage_ms[k] > 0 else None)
                                   for k, a in (("separation+loudness", "separation"), ("embedding+cosine", "embedding"), ("asr_encoder+decoder", "asr_encoder"))},
            "algorithmic_flops_per_step_rank0": fl,
        }
        if world == 1 and not args.no_cpu_baseline:
            if sample_window is None:
                sample_window = batches[0][0][:WINDOW].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline_pipe(sep_sd, spk_sd, asr_sd, sample_window, with_asr)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# cfg1: the reference's own CPU-runnable case (BASELINE configs[0]) — the hot-path subset of infer() on the two asset clips,
# host arrays in, host arrays out, ONE window per call (what a user of the drop-in API sees)
# ----------------------------------------------------------------------------------------------
def cfg1_bench(args):
    import wave as wavmod
    from targetdiarization_amd.pipeline import HotPath
    from targetdiarization_amd.weights import (recipe_eres2netv2_state_dict, recipe_paraformer_decoder_state_dict,
                                               recipe_paraformer_state_dict, recipe_state_dict)

    def load(fn):
        with wavmod.open(os.path.join(ROOT, "tests", "golden", fn), "rb") as w:
            return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
    mix, tgt = load("chat_mix.wav"), load("female_a.wav")
    torch.cuda.set_device(0)
    asr_sd = dict(recipe_paraformer_state_dict(0, 50)); asr_sd.update(recipe_paraformer_decoder_state_dict(0, 16))
    hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), asr_sd, cuda_device=0)

    def step():
        # H2 on the target clip || H1 (one 8.665 s window) -> H2 on both streams + cosine || H3 on both streams
        return hp.run([mix], target_clip=tgt)
    for _ in range(max(args.warmup, 3)):              # (graphs are captured at the second sighting of a shape)
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        hp.separate([mix])
    torch.cuda.synchronize()
    dsep = time.perf_counter() - t1
    secs = len(mix) / 16000.0
    print(json.dumps({"metric": METRIC + " — hot-path subset of BASELINE configs[0]", "value": secs * args.steps / dt, "unit": "audio-s/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": DTYPE, "data": "reference assets (chat_mix.wav 8.665 s + female_a.wav 1.923 s), recipe weights",
                      "config": {"workload": "BASELINE configs[0] hot-path subset: host arrays in/out, batch 1: MossFormer2 on the mix as ONE window (S = 17 328, "
                                             "HIP-graph replay) -> ERes2NetV2 on the target clip and both separated streams + cosine -> Paraformer encoder + CIF + "
                                             "decoder on both streams"},
                      "separation_only_ms": dsep / args.steps * 1e3, "separation_only_rtf": secs * args.steps / dsep,
                      "tokens_stream0": len(out["asr"][0][0]["token_ids"])}), flush=True)


# ----------------------------------------------------------------------------------------------
# cfg2: MossFormer2 only (BASELINE configs[1]) — H1 kernel measurement
# ----------------------------------------------------------------------------------------------
def cfg2_bench(args):
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_state_dict
    B, T = args.batch, int(round(args.seconds * 16000))
    sd = recipe_state_dict(seed=0, num_blocks=24)
    sep = MossFormer2Separator(sd, device=dev)
    wave_np = synth_mixtures(B, T, seed=2 + 1000 * rank)
    wav = torch.from_numpy(wave_np).to(dev)
    S = (T - 16) // 8 + 1
    for _ in range(args.warmup):
        out = sep(wav)
    torch.cuda.synchronize(dev)
    sep.profile_enable(24 * args.steps)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sep(wav)
    barrier()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_launches = sep.profile_collect()
    assert torch.isfinite(out).all()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = world * B * T / 16000.0 * args.steps / dt
    flops_step = sep.flops(B, T)
    if rank == 0:
        M = B * S
        gemm_flops = 2.0 * M * 512 * 2176
        ach = gemm_flops / (gemm_ms / max(gemm_launches, 1) * 1e-3) / 1e12 if gemm_launches else None
        traffic, mfma_util, clock_held, traffic_note = recorded_pmc(M)
        line = {
            "metric": "real-time factor (audio-sec/wall-sec), MossFormer2 separation only (sub-measurement of the full pipe), 16kHz mono",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: MossFormer2 separation only, batch of {B} synthetic "
                                   f"{args.seconds:g} s 16 kHz 2-speaker mixtures per GPU, 24 blocks, recipe weights",
                       "per_gpu_batch": B, "samples_per_window": T, "frames_per_window": S,
                       "parallelism": f"{world} independent replicas (windows sharded, no collective)"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_H3_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ach / PEAK_H3_TFLOPS) if ach else None, "traffic": traffic, "kernel": KERNEL, "fusion_note": FUSION_NOTE,
                         "peak_note": "algorithmic fp32-accurate FLOP/s ceiling of the kernel = dense f16 MFMA peak 2500 / 3 passes",
                         "mfma_pipe_executed_tflops": (ach * H3_PASSES) if ach else None,
                         "launches_timed": gemm_launches, "ms_per_launch": gemm_ms / max(gemm_launches, 1),
                         "algorithmic_flops_per_launch": gemm_flops,
                         "pmc_recorded_mfma_pipe_utilisation": mfma_util,
                         "pmc_recorded_clock_GHz": clock_held,     # the chip is power-limited under this load: the peak above assumes 2.4 GHz
                         "frac_of_peak_at_recorded_clock": (ach / (PEAK_H3_TFLOPS * clock_held / 2.4)) if (ach and clock_held) else None,
                         "traffic_source": "pmc_recorded (not measured by this run)" if traffic is not None else None, "traffic_note": traffic_note,
                         "whole_path_tflops_per_gpu": flops_step * args.steps / dt / 1e12,
                         "whole_path_frac": flops_step * args.steps / dt / 1e12 / PEAK_H3_TFLOPS},
            "algorithmic_flops_per_step_per_gpu": flops_step,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_h1(sd, wave_np, budget_windows=4)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# hbm: the HBM-bound kernels of the path at fixed shapes, for `rocprofv3 --kernel-trace --stats` (tools/hbm_table.py turns the
# per-kernel times into GB/s against the 8 TB/s spec: SURVEY §8d asks a1/a2, the encoder/decoder strided convolutions and the norms
# to be reported against the HBM roof).  Shapes: MossFormer2 at the config-2 size (32 x 4 s), denoise_vocal (identity net body:
# the MDX network is a third-party plug-in) on 60 s of 44.1 kHz stereo, fbank SV / ASR + LFR/CMVN on 64 x 10 s.
# ----------------------------------------------------------------------------------------------
def hbm_workload(args):
    from targetdiarization_amd.audio_processor import AudioProcessor
    from targetdiarization_amd.frontend import Fbank, lfr_cmvn
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.weights import recipe_state_dict
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sep = MossFormer2Separator(recipe_state_dict(0, 24), device=dev)
    wav = torch.from_numpy(synth_mixtures(32, 64000, seed=2)).to(dev)
    ap = AudioProcessor(is_denoise_vocal=True, mdx_weights_file="mdx/weights/UVR-MDX-NET-Inst_HQ_3.onnx", cuda_device=0, quality=3,
                        verbose_log=False, mdx_model=lambda spec: spec)
    rng = np.random.default_rng(9)
    stereo = (0.1 * rng.standard_normal((60 * 44100, 2))).astype(np.float32)
    fb_sv, fb_asr = Fbank("sv", dev), Fbank("asr", dev)
    w10 = torch.from_numpy(synth_mixtures(64, WINDOW, seed=7)).to(dev)
    shift, scale = torch.zeros(560, device=dev), torch.ones(560, device=dev)
    for it in range(args.warmup + args.steps):
        sep(wav)
        ap.denoise_vocal(stereo, 44100)
        fb_sv(w10)
        lfr_cmvn(fb_asr(w10), shift, scale)
        torch.cuda.synchronize()
    print(json.dumps({"workload": "hbm", "passes": args.warmup + args.steps,
                      "shapes": {"mossformer2": [32, 64000], "denoise_vocal": [60 * 44100, 2], "fbank": [64, WINDOW]}}), flush=True)


# ----------------------------------------------------------------------------------------------
# --dry-run: the N-rank plumbing on the CPU (gloo): sharding, the all-gather, the max-over-ranks clock and the
# JSON line, with fabricated embeddings instead of kernels (tests/test_bench_launch.py)
# ----------------------------------------------------------------------------------------------
def dry_run(args):
    import torch.distributed as dist
    from targetdiarization_amd.pipeline import gather_embeddings, shard_indices
    rank, _, world = dist_env()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    per_step = args.utterances_per_step

    def emb_of(i, stream):
        g = torch.Generator().manual_seed(1000 * i + stream)
        return torch.randn(192, generator=g)

    def step():
        mine = shard_indices(per_step, rank, world)
        local = torch.stack([emb_of(i, s) for i in mine for s in (0, 1)]) if mine else torch.zeros(0, 192)
        return gather_embeddings(local, per_step, rank, world)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ref = torch.stack([emb_of(i, s) for i in range(per_step) for s in (0, 1)])
    ok = bool(torch.equal(full, ref))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": per_step * 30.0 * args.steps / float(dt.item()), "unit": "audio-s/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(dt.item()) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "none", "data": "dry-run (no kernels)",
                          "config": {"workload": "DRY RUN: rank plumbing only (gloo, fabricated embeddings)"}, "dry_run": True,
                          "gather_ok": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def spawn_ranks(args) -> int:
    """`--gpus N` without a launcher: start the N ranks as CHILDREN (one process per GPU) through
    torch.distributed.run.  This process has not touched a GPU (importing torch does not initialise HIP) and never
    replaces itself: it waits for the children and returns their status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)       # warm (3 warm-up iterations), >= 10 timed
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "hbm"],
                    help="default: cfg4 (BASELINE configs[3], the full pipe on 1800 s) at one GPU, cfg5 (BASELINE configs[4], the "
                         "1000-utterance job, strong scaling) at more; cfg2 = MossFormer2 only; cfg3 = 600 s without the ASR encoder")
    ap.add_argument("--asr-rows-per-launch", type=int, default=65536, help="LFR frames per Paraformer launch sequence")
    ap.add_argument("--embed-frames-per-launch", type=int, default=180000, help="fbank frames per ERes2NetV2 launch sequence (180 x 10 s windows: the 360 clips of the default workload in two equal launch sequences; measured 120k / 180k / 360k: 550.8 / 538.7 / 532.0 ms per step at 154 / 179 / 255 GiB peak)")
    ap.add_argument("--windows-per-launch", type=int, default=90, help="10 s windows per MossFormer2 launch sequence (90 = 1.8 M token rows: the launches get larger, not more — DESIGN.md §5)")
    ap.add_argument("--utterances", type=int, default=1000, help="cfg5: utterances in the job")
    ap.add_argument("--utterances-per-step", type=int, default=200, help="cfg5: utterances per step (global batch, all ranks together)")
    ap.add_argument("--batch", type=int, default=32, help="cfg2: windows per GPU")
    ap.add_argument("--seconds", type=float, default=4.0, help="cfg2: window length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-punc", action="store_true", help="configs[3] without the CT-Transformer punctuation stage (the round-1/2 form of the workload: pass-through)")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing on the CPU (gloo), no kernels")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if args.dry_run:
        return dry_run(args)
    if args.workload is None:
        args.workload = "cfg4" if world == 1 else "cfg5"
    if args.workload == "cfg1":
        return cfg1_bench(args)
    if args.workload == "cfg2":
        return cfg2_bench(args)
    if args.workload == "hbm":
        return hbm_workload(args)
    return pipe_bench(args)


if __name__ == "__main__":
    main()
