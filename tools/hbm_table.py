#!/usr/bin/env python3
"""usage: tools/hbm_table.py <kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload hbm`> [out.json]
Per HBM-bound kernel of the path: algorithmic bytes per launch (from the fixed shapes of the `hbm` workload) / mean launch time ->
GB/s and the fraction of the 8 TB/s spec.  The GEMM-shaped launches (DFT, mel, x3 cores) are MFMA-bound and not listed."""
import csv
import json
import re
import sys

M, B, S, SP, T = 255968, 32, 7999, 8192, 64000          # MossFormer2 at the config-2 size
R, NFFT, DIMT, CHUNK = None, 6144, 256, 522240           # denoise_vocal blocks: R = rows (blocks x 2 channels), from the launch count
FB_B, FB_N = 64, 160000
FB_F = 1 + (FB_N - 400) // 160

# kernel-name pattern -> (label, algorithmic bytes per launch or a function of the mean launch count, reference lines)
KB = 1024
TABLE = [
    (r"encoder_kernel", "encoder Conv1d(1->512,k16,s8)+ReLU (+GroupNorm partials)", B * T * 4 + M * 2048, "mossformer2.py:157-210"),
    (r"gn_apply_kernel<0", "GroupNorm(1,512) apply", M * 4096, "mossformer2.py:487"),
    (r"gn_apply_kernel<1", "GroupNorm apply + skip + PReLU", M * 6144, "mossformer2.py:389-396,500"),
    (r"rowblock_stats_kernel", "GroupNorm statistics", M * 2048, "mossformer2.py:389"),
    (r"layernorm_kernel<512", "LayerNorm(512)", M * 4096, "mossformer2.py:320"),
    (r"layernorm_kernel<256", "CLayerNorm(256) + planes", M * 3072, "layer_norm.py:9-30"),
    (r"conv17_kernel<4", "conv17<4>: SiLU + depthwise k17 of v|u -> K-major planes", M * 16384, "conv_module.py:180-220"),
    (r"conv17_kernel<3", "conv17<3>: q/k depthwise k17 + OffsetScale + rotary -> 4 heads as planes", M * 512 + 4 * B * SP * 512, "mossformer_block.py:76-86,230-233"),
    (r"conv17_kernel<1", "conv17<1>: to_out depthwise k17 + residual -> x, x planes", M * 8192, "mossformer_block.py:219"),
    (r"conv17_kernel<0", "conv17<0>: FSMN to_u|to_v depthwise k17 -> uv, x_u planes", M * 5120, "mossformer_block.py:316-325"),
    (r"ddn_conv1_kernel", "DilatedDenseNet conv1 (k39) + IN partials", M * 2048, "fsmn.py:103-107"),
    (r"ddn_conv2_kernel", "DilatedDenseNet IN+PReLU + grouped conv2 (k39, dil 2)", M * 3072, "fsmn.py:108-111"),
    (r"fsmn_tail_kernel", "IN + PReLU + gate + CLayerNorm -> planes", M * 5120, "mossformer_block.py:323-324,422-423"),
    (r"kmajor_to_rows_kernel", "lin_k K-major -> row-major planes", B * SP * 1024, "-"),
    (r"xplanes_kernel", "x -> planes (once per forward)", M * 4096, "-"),
    (r"decoder_dot_kernel", "decoder ConvTranspose1d taps", 2 * M * 2048 + 2 * M * 64, "mossformer2.py:213-257"),
    (r"decoder_ola_kernel", "decoder overlap-add + pad/trim", 2 * M * 64 + B * 2 * T * 4, "mossformer2.py:583-588"),
    (r"stft_frame_kernel", "STFT reflect-pad framing (x -> frames[R,256,6144])", "stft", "AudioProcessor.py:82-99"),
    (r"istft_ola_kernel", "iSTFT window + overlap-add + envelope (frames -> y)", "stft", "AudioProcessor.py:101-120"),
    (r"resample_poly_kernel", "polyphase resampler", None, "AudioProcessor.py:549-569"),
    (r"fbank_frame_kernel", "fbank framing: DC, pre-emphasis, window, pad to 512", FB_B * FB_N * 4 + FB_B * FB_F * 2048, "torchaudio kaldi.fbank (third-party)"),
    (r"fbank_submean_kernel", "fbank per-utterance mean removal", 2 * FB_B * FB_F * 320, "modelscope SV front-end"),
    (r"lfr_cmvn_kernel", "LFR (7,6) + CMVN", FB_B * FB_F * 320 + FB_B * ((FB_F + 5) // 6) * 2240, "funasr WavFrontend"),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = []
    for pat, label, nbytes, ref in TABLE:
        hit = [r for r in rows if re.search(pat, r["Name"])]
        if not hit:
            continue
        calls = sum(int(r["Calls"]) for r in hit)
        total_ns = sum(float(r["TotalDurationNs"]) for r in hit)
        us = total_ns / calls / 1e3
        if nbytes == "stft":
            # the launch count fixes R: every denoise_vocal pass launches the kernel once over all R rows; frames are R x 256 x 6144 fp32
            nb = None
        else:
            nb = nbytes
        e = {"kernel": pat, "what": label, "reference": ref, "launches": calls, "us_per_launch": round(us, 2)}
        if nb:
            e.update({"algorithmic_bytes_per_launch": nb, "GBps": round(nb / us / 1e3, 1), "frac_of_8TBps": round(nb / us / 1e3 / 8000.0, 3)})
        out.append(e)
    # STFT rows: 60 s of 44.1 kHz stereo = 4 outer segments of <= 17 s -> blocks of 516 096 new samples: R rows counted from the frames size
    n = 60 * 44100
    seg = int(15.0 * 44100); margin = 44100
    nb_total = 0
    a = 0
    while a < n:
        b = min(a + seg, n)
        lo, hi = max(0, a - margin), min(n, b + margin)
        nb_total += -(-(hi - lo) // (CHUNK - NFFT))
        a = b
    Rrows = nb_total * 2
    for e in out:
        if e["kernel"] in ("stft_frame_kernel", "istft_ola_kernel"):
            nb = Rrows * DIMT * NFFT * 4 + Rrows * CHUNK * 4
            e.update({"algorithmic_bytes_per_launch": nb, "rows": Rrows, "GBps": round(nb / e["us_per_launch"] / 1e3, 1),
                      "frac_of_8TBps": round(nb / e["us_per_launch"] / 1e3 / 8000.0, 3)})
    res = {"source": sys.argv[1], "peak_GBps": 8000.0, "note": "algorithmic bytes (each operand once) / mean launch time; streaming copies reach ~6.3 TB/s on MI355X",
           "kernels": out}
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    for e in out:
        print(f'{e["kernel"]:28s} {e["launches"]:6d} x {e["us_per_launch"]:9.1f} us  {e.get("GBps", "-"):>8} GB/s  {e.get("frac_of_8TBps", "-")}')


if __name__ == "__main__":
    main()
