#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric: real-time factor).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: MossFormer2 separation,
batch of 32 synthetic 4 s / 16 kHz two-speaker mixtures (recipe of SURVEY.md §8d config 2),
24-block model at the reference's constructor defaults, recipe (random-init) weights, fp32.
N>1: every rank runs the same per-GPU workload on its own mixtures (independent units, weak
scaling; no data-path collective in the separation-only configuration).

Output: ONE JSON line on rank 0 (see the driver contract), plus
  roofline     — dominant kernel (the to_hidden+to_qk fp32-MFMA GEMM, 24 launches/step):
                 achieved = algorithmic FLOPs per launch / mean launch duration measured with
                 HIP events recorded on the forward's stream inside the timed region.
  cpu_baseline — the oracle (CPU restatement of the reference, pinned to it by
                 tests/golden) timed on this host's cores on a bounded sample of the same
                 workload (N=1, rank 0 only).
"""
import argparse
import json
import os
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
# Roofline of the dominant kernel: it computes fp32-accurate products as 3 f16 MFMA passes over
# exponent-aligned split operands, so its ceiling in ALGORITHMIC FLOP/s is the dense f16 MFMA
# peak / 3 = 833.3 TFLOP/s.
PEAK_H3_TFLOPS = PEAK_F16_MFMA_TFLOPS / H3_PASSES


def synth_mixtures(batch: int, n: int, seed: int) -> np.ndarray:
    """SURVEY.md §8d config-2 generator: s1+s2, s_k = 0.05*N(0,1) shaped by a 4 Hz
    raised-cosine AM envelope with random phase, clipped to [-1,1], f32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64) / 16000.0
    out = np.zeros((batch, n), dtype=np.float64)
    for b in range(batch):
        for _ in range(2):
            ph = rng.uniform(0, 2 * np.pi)
            env = 0.5 * (1.0 - np.cos(2 * np.pi * 4.0 * t + ph))
            out[b] += 0.05 * rng.standard_normal(n) * env
    return np.clip(out, -1.0, 1.0).astype(np.float32)


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


def cpu_baseline(sd, wave_np, budget_windows: int):
    """Time the oracle on the host cores over `budget_windows` windows of the workload."""
    from oracle import mossformer2_oracle as orc          # checker / baseline leg only
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


def cfg5_bench(args):
    """BASELINE configs[4] (scaled: --utterances per GPU instead of 1000 in total): utterance i lives on rank
    i % P, each utterance = three 10 s windows through MossFormer2, ERes2NetV2 on both separated streams, ONE
    all-gather of the [n_i*2,192] embedding blocks (RCCL over xGMI).  Host arrays in, host arrays out (the
    boundary of `pipeline.HotPath.run`), so this number includes PCIe."""
    import torch.distributed as dist
    from targetdiarization_amd.pipeline import HotPath, shard_indices
    from targetdiarization_amd.weights import recipe_eres2netv2_state_dict, recipe_state_dict
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    hp = HotPath(recipe_state_dict(0, 24), recipe_eres2netv2_state_dict(0), None, cuda_device=local_rank)
    n_total = args.utterances * world
    mine = shard_indices(n_total, rank, world)
    utts = [synth_mixtures(1, 480000, seed=5 + i)[0] for i in mine]
    target = np.random.default_rng(5).standard_normal(192).astype(np.float32)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):        # same shapes as the timed steps: workspaces are sized on first use
        hp.run(utts, target, rank, world, n_total, with_asr=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = hp.run(utts, target, rank, world, n_total, with_asr=False)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert out["embeddings"].shape == (n_total * 2, 192) and np.isfinite(out["scores"]).all()
    if rank == 0:
        print(json.dumps({"metric": "real-time factor (audio-sec/wall-sec), hot path cfg5 (host arrays in/out)", "value": n_total * 30.0 * args.steps / dt,
                          "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[4] scaled: {n_total} x 30 s utterances, utterance i on rank i % {world}, "
                                                 "MossFormer2 (3 windows each) -> ERes2NetV2 on both streams -> all-gather of [n*2,192] embeddings "
                                                 "-> cosine scores; recipe weights",
                                     "utterances_per_gpu": args.utterances}}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def pipeline_bench(args):
    """BASELINE configs[2]/[3] at window granularity, device-resident: every 10 s window goes
    H1 -> (both separated streams) H2 + cosine [-> H3 encoder]; windows are independent units."""
    from targetdiarization_amd import ops
    from targetdiarization_amd.paraformer import ParaformerEncoder
    from targetdiarization_amd.separator import MossFormer2Separator
    from targetdiarization_amd.speaker import ERes2NetV2
    from targetdiarization_amd.weights import (recipe_eres2netv2_state_dict, recipe_paraformer_state_dict,
                                               recipe_state_dict)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    total_s = 600 if args.workload == "cfg3" else 1800
    T = 160000
    nwin = total_s // 10
    sep = MossFormer2Separator(recipe_state_dict(0, 24), device=dev)
    spk = ERes2NetV2(recipe_eres2netv2_state_dict(0), dev)
    asr = ParaformerEncoder(recipe_paraformer_state_dict(0, 50), dev) if args.workload == "cfg4" else None
    wav = torch.from_numpy(synth_mixtures(nwin, T, seed=3 if args.workload == "cfg3" else 4)).to(dev)
    target = torch.randn(192, device=dev)
    CH = 20          # windows per launch group

    ASR_CH = 180     # separated streams per encoder launch sequence (M = 180 x 167 LFR frames fills the 256 CUs)

    def step():
        t_sep = t_spk = t_asr = 0.0
        streams = []
        for c in range(0, nwin, CH):
            x = wav[c:c + CH]
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e2 = torch.cuda.Event(enable_timing=True)
            e0.record()
            y = sep(x).reshape(-1, T)                      # [2*CH, T] separated streams
            e1.record()
            emb = spk(y)
            sc = ops.cosine_scores(emb, target)
            e2.record()
            if asr is not None:
                streams.append(y)
            torch.cuda.synchronize()
            t_sep += e0.elapsed_time(e1); t_spk += e1.elapsed_time(e2)
        if asr is not None:                                # H3 on all separated streams, batched across the windows
            ys = torch.cat(streams)
            e2 = torch.cuda.Event(enable_timing=True); e3 = torch.cuda.Event(enable_timing=True)
            e2.record()
            for c in range(0, ys.shape[0], ASR_CH):
                enc = asr(ys[c:c + ASR_CH])
            e3.record()
            torch.cuda.synchronize()
            t_asr += e2.elapsed_time(e3)
        return t_sep, t_spk, t_asr

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = [0.0, 0.0, 0.0]
    for _ in range(args.steps):
        r = step()
        acc = [a + b for a, b in zip(acc, r)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    S = (T - 16) // 8 + 1
    F = 1 + (T - 400) // 160
    fl = {"sep": sep.flops(nwin, T), "spk": spk.flops(2 * nwin, F), "asr": asr.flops(2 * nwin, (F + 5) // 6) if asr else 0.0}
    line = {"metric": "real-time factor (audio-sec/wall-sec), hot path " + args.workload, "value": total_s * args.steps / dt,
            "unit": "audio-s/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE {'configs[2]' if args.workload == 'cfg3' else 'configs[3]'}: {total_s} s synthetic audio as "
                                   f"{nwin} x 10 s windows: MossFormer2 -> ERes2NetV2 on both streams + cosine"
                                   + (" -> Paraformer encoder on both streams" if asr else "") + ", recipe weights, device-resident"},
            "stage_ms_per_step": {"separation": acc[0] / args.steps, "embedding+cosine": acc[1] / args.steps, "asr_encoder": acc[2] / args.steps},
            "stage_tflops": {"separation": fl["sep"] / (acc[0] / args.steps * 1e-3) / 1e12,
                             "embedding": fl["spk"] / (acc[1] / args.steps * 1e-3) / 1e12,
                             "asr_encoder": (fl["asr"] / (acc[2] / args.steps * 1e-3) / 1e12) if asr else None},
            "algorithmic_flops_per_step": fl}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--utterances", type=int, default=16, help="cfg5: 30 s utterances per GPU (BASELINE config 5 has 1000 in total)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 (default, the headline config): separation only; cfg3: + ERes2NetV2 embeddings + cosine "
                         "on 10 min of audio; cfg4: + Paraformer encoder on 30 min.  cfg3/cfg4 are extra measurements.")
    args = ap.parse_args()
    if args.workload == "cfg5":
        return cfg5_bench(args)
    if args.workload != "cfg2":
        return pipeline_bench(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
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
    wav = torch.from_numpy(wave_np).to(dev)           # resident in HBM before timing
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

    audio_s_per_step = world * B * T / 16000.0
    value = audio_s_per_step * args.steps / dt
    flops_step = sep.flops(B, T)
    if rank == 0:
        print(f"[bench] gpu: {dt / args.steps * 1e3:.1f} ms/step, RTF {value:.1f}", file=sys.stderr, flush=True)
        M = B * S
        gemm_flops = 2.0 * M * 512 * 2176          # to_hidden+to_qk: [M,512] x [512,2176]
        ach = gemm_flops / (gemm_ms / max(gemm_launches, 1) * 1e-3) / 1e12 if gemm_launches else None
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get("gemm_to_hidden_hbm_bytes_per_launch")
        line = {
            "metric": "real-time factor (audio-sec/wall-sec), MossFormer2 separation, 16kHz mono",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: MossFormer2 separation only, batch of {B} synthetic "
                                   f"{args.seconds:g} s 16 kHz 2-speaker mixtures per GPU, 24 blocks, recipe weights",
                       "per_gpu_batch": B, "samples_per_window": T, "frames_per_window": S,
                       "parallelism": f"{world} independent replicas (windows sharded, no collective)"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_H3_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ach / PEAK_H3_TFLOPS) if ach else None, "traffic": traffic,
                         "kernel": "gemm_h3_kernel<to_hidden+to_qk: split-f16 x3 MFMA over pre-split planes, ScaleNorm gain/bias epilogue (SiLU applied by the consuming depthwise convolution)>",
                         "peak_note": "algorithmic fp32-accurate FLOP/s ceiling of the kernel = dense f16 MFMA peak 2500 / 3 passes; "
                                      "vs the fp32-input MFMA peak (157.3) the same number is frac_vs_f32_mfma_peak",
                         "frac_vs_f32_mfma_peak": (ach / PEAK_F32_MFMA_TFLOPS) if ach else None,
                         "mfma_pipe_executed_tflops": (ach * H3_PASSES) if ach else None,
                         "launches_timed": gemm_launches, "ms_per_launch": gemm_ms / max(gemm_launches, 1),
                         "algorithmic_flops_per_launch": gemm_flops,
                         "whole_path_tflops_per_gpu": flops_step * args.steps / dt / 1e12,
                         "whole_path_frac_vs_f32_mfma_peak": flops_step * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "algorithmic_flops_per_step_per_gpu": flops_step,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, wave_np, budget_windows=4)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
