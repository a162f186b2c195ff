/* tdx.h — C-ABI of the MI355X-native TargetDiarization hot path (libtdx.so).
 *
 * The reference (ishine/TargetDiarization) is pure Python and has no FFI; its "plugin
 * boundary" for this path is four Python call sites.  Each entry point below replaces one
 * of them (reference file:line cited per function).  The Python host in
 * targetdiarization_amd/ binds these with ctypes (see INTEGRATION.md for the stub a
 * reference maintainer would add).
 *
 * Conventions
 *  - plain C types only; every pointer named *_dev is a DEVICE pointer owned by the caller
 *    (PyTorch-ROCm allocates; `tensor.data_ptr()`), every stream is a hipStream_t passed
 *    as void*.  The library owns only the opaque model handle (weights + tables).
 *  - every function returns 0 on success, non-zero TDX_E_* otherwise; no C++ exception
 *    crosses the boundary; tdx_last_error() returns a thread-local message.
 *  - all launches are asynchronous on the given stream; no hipMalloc/hipFree/sync inside
 *    a forward call (graph-capture safe).
 *  - threading: a handle is immutable after create (weights + tables); every *_forward /
 *    predict / decode call may be issued concurrently from several host threads on ONE handle
 *    provided each concurrent call has its OWN workspace and its OWN stream.  The only
 *    per-call state the library keeps is tdx_mf2's fork/join context (one side stream + three
 *    events PER CALLER STREAM, created on the first eager forward on that stream under an
 *    internal mutex; a forward on a stream that is being captured and has no context yet runs
 *    unforked: same results).  Two calls that share a workspace or a stream must be ordered
 *    by the caller (the Python host serialises per model object: _lib.HandleGuard).
 *    tdx_mf2_profile_* and tdx_mf2_enable_taps are single-caller diagnostics (profile slots
 *    are claimed under the same mutex; do not toggle them while forwards are in flight).
 */
#ifndef TDX_H
#define TDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDX_OK 0
#define TDX_E_INVALID 1   /* bad argument / shape */
#define TDX_E_BLOB 2      /* weight blob malformed or tensor missing */
#define TDX_E_HIP 3       /* HIP runtime error */
#define TDX_E_WORKSPACE 4 /* workspace too small */

/* library version string, e.g. "tdx 0.1.0 gfx950" */
const char* tdx_version(void);
/* thread-local description of the last non-zero status returned on this thread */
const char* tdx_last_error(void);

/* ------------------------------------------------------------------------------------
 * H1  MossFormer2 separator — replaces `self.separater(audio_data_tensor)`
 *     AudioProcessor.py:943  (model: look2hear/models/mossformer2.py:563-589)
 *     loader replaces BaseModel.from_pretrain  base_model.py:52-64
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_mf2 tdx_mf2;

typedef struct tdx_mf2_config {
    int32_t num_blocks;   /* 24  (mossformer2.py:535) */
    int32_t channels;     /* 512 (in_channels == out_channels, :533-534); must be 512 */
    int32_t kernel_size;  /* 16  (:536), stride = kernel_size/2 */
    int32_t num_spks;     /* 2   (:538); must be 2 */
    int32_t group_size;   /* 256 (mossformer_block.py:434) */
    int32_t reserved[3];
} tdx_mf2_config;

/* weights_blob: host memory in the TDXW container (targetdiarization_amd/weights.py:pack_blob)
 * holding the reference's state_dict tensors under their reference names. */
int tdx_mf2_create(const tdx_mf2_config* cfg, const void* weights_blob, size_t blob_bytes,
                   int device, tdx_mf2** out);
int tdx_mf2_destroy(tdx_mf2* h);
/* bytes of device workspace tdx_mf2_forward needs for a [B,T] batch (0 on bad shape) */
size_t tdx_mf2_workspace_bytes(const tdx_mf2* h, int B, int T);
/* wav_dev [B,T] f32 -> out_dev [B,2,T] f32.  All B windows must have the same T (the
 * reference never pads a batch: mossformer_block.py:485 passes no mask). */
int tdx_mf2_forward(tdx_mf2* h, const float* wav_dev, int B, int T, float* out_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);
/* algorithmic FLOPs (2*MAC) of one forward at [B,T] — SURVEY.md §8(d) accounting */
double tdx_mf2_flops(const tdx_mf2* h, int B, int T);

/* keep copies of the layer-0 intermediates and the mask for tdx_mf2_tap (costs workspace) */
int tdx_mf2_enable_taps(tdx_mf2* h, int on);

/* Live timing of the dominant kernel (the to_hidden+to_qk GEMM, one launch per block per
 * forward): after enable, each forward records a HIP event pair around that launch on the
 * forward's stream until max_records pairs are used; collect (after the caller synchronised
 * the stream) returns the summed elapsed ms and the number of launches, and re-arms. */
int tdx_mf2_profile_enable(tdx_mf2* h, int max_records);
int tdx_mf2_profile_collect(tdx_mf2* h, double* total_ms, int* launches);

/* test/diagnostic tap: copy a named intermediate of the LAST forward on (h, workspace)
 * into dst_dev (f32).  names: "enc","z","after_flash0","after_fsmn0","after_stack","mask".
 * Layouts are token-major ([B,S,C]; "mask" is [2,B,S,C]).  Returns element count via *n.
 * "headroom": [num_blocks][2] = the largest |f16| written under each layer's static plane scale for (v|u, lin_k) — the scales
 * are bounds derived from the weights (2^15 after scaling); 2^15 / value is the headroom the bound left (needs taps). */
int tdx_mf2_tap(tdx_mf2* h, const char* name, int B, int T, void* workspace_dev,
                float* dst_dev, size_t dst_elems, size_t* n, void* stream);

/* ------------------------------------------------------------------------------------
 * Stand-alone ops exported for parity tests of individual reference functions
 * ---------------------------------------------------------------------------------- */
/* FLASH_ShareA_FFConvM.cal_attention (mossformer_block.py:222-294), non-causal, no mask.
 * q/k inputs [B,S,128] are the four OffsetScale heads BEFORE rotary; v,u [B,S,E], E%64==0.
 * freqs_dev: 16 rotary frequencies.  att_v/att_u out [B,S,E]. */
int tdx_cal_attention(const float* quad_q_dev, const float* lin_q_dev, const float* quad_k_dev,
                      const float* lin_k_dev, const float* v_dev, const float* u_dev,
                      const float* freqs_dev, int B, int S, int E, float* att_v_dev,
                      float* att_u_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
size_t tdx_cal_attention_workspace_bytes(int B, int S, int E);

/* DilatedDenseNet.forward (fsmn.py:103-111) on token-major p[B,S,256] -> out[B,S,256].
 * w1[256,39], w2[256,2,39], in_g/in_b[2,256] (InstanceNorm affine), prelu[2,256]. */
int tdx_dilated_dense_net(const float* p_dev, int B, int S, const float* w1_dev,
                          const float* w2_dev, const float* in_g_dev, const float* in_b_dev,
                          const float* prelu_dev, float* out_dev, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
size_t tdx_dilated_dense_net_workspace_bytes(int B, int S);

/* C[M,N] = A[M,K] * W[N,K]^T (+bias[N]); fp32 MFMA; N%128==0, K%32==0.  Test hook for the
 * GEMM core every nn.Linear / 1x1 Conv1d on the path goes through. */
int tdx_linear(const float* a_dev, const float* w_dev, const float* bias_dev, int M, int N, int K,
               float* c_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * a2  Kaldi fbank front-end ("mel frontend") — replaces torchaudio.compliance.kaldi.fbank as
 *     called inside modelscope's ERes2NetV2 pipeline (reached from TargetASR.py:161) and
 *     funasr's WavFrontend (reached from ASRProcessor.py:424).  [third-party algorithm]
 *     mode 0 = speaker front-end: povey window, input in [-1,1], per-utterance mean removed
 *     mode 1 = ASR front-end: hamming window, input x32768 (LFR/CMVN: tdx_lfr_cmvn)
 *     wav_dev [B,N] -> feat_dev [B,F,80], F = tdx_fbank_frames(N) = 1 + (N-400)/160.
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_fbank tdx_fbank;
int tdx_fbank_create(int mode, int device, tdx_fbank** out);
int tdx_fbank_destroy(tdx_fbank* h);
int tdx_fbank_frames(int N);
size_t tdx_fbank_workspace_bytes(int B, int N);
int tdx_fbank_forward(tdx_fbank* h, const float* wav_dev, int B, int N, float* feat_dev,
                      void* workspace_dev, size_t workspace_bytes, void* stream);
/* funasr WavFrontend.apply_lfr(m=7,n=6) + apply_cmvn: feat [B,F,80] -> out [B,ceil(F/6),560],
 * out = (stacked + shift[560]) * scale[560] */
int tdx_lfr_cmvn(const float* feat_dev, int B, int F, const float* shift_dev, const float* scale_dev,
                 float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * a1  MDX block STFT / iSTFT — replaces ConvTDFNet.stft / .istft  AudioProcessor.py:82-120
 *     (torch.stft(n_fft, hop, hann periodic, center=True) on chunks of hop*(dim_t-1) samples,
 *     packed (L-re, L-im, R-re, R-im) x dim_f x dim_t; inverse zero-pads bins >= dim_f).
 *     x_dev [R, chunk] <-> spec_dev [R, 2, dim_f, dim_t], R = n_blocks * 2 channels.
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_stft tdx_stft;
int tdx_stft_create(int n_fft, int hop, int dim_f, int dim_t, int device, tdx_stft** out);
int tdx_stft_destroy(tdx_stft* h);
int tdx_stft_chunk_size(const tdx_stft* h);
size_t tdx_stft_workspace_bytes(const tdx_stft* h, int R);
int tdx_stft_forward(tdx_stft* h, const float* x_dev, int R, float* spec_dev, void* workspace_dev,
                     size_t workspace_bytes, void* stream);
int tdx_stft_inverse(tdx_stft* h, const float* spec_dev, int R, float* y_dev, void* workspace_dev,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * N3  MDX-Net denoiser body — replaces `self.mdx_model.run(None, {"input": mix_spec})[0]`  AudioProcessor.py:630
 *     (an onnxruntime session over UVR-MDX-NET-*.onnx, AudioProcessor.py:224-241: third-party; the network is KUIELab's
 *     ConvTDFNet / TFC-TDF v2: first 1x1 conv, n = num_blocks/2 encoder levels of [l x (3x3 conv + BN + ReLU) + TDF bottleneck
 *     Linear pair over frequency] with 2x2 stride-2 down convs, bottleneck block, n decoder levels with 2x2 transposed convs and
 *     multiplicative skips, final 1x1 conv).  blob: TDXW container with the PyTorch module names of that class (first_conv.*,
 *     encoding_blocks.{i}.tfc.H.{j}.*, encoding_blocks.{i}.tdf.*, ds.{i}.*, bottleneck_block.*, us.{i}.*, decoding_blocks.{i}.*,
 *     final_conv.*), BatchNorm2d in eval mode.  spec_dev [B,4,dim_f,dim_t] (tdx_stft_forward output) -> out_dev, same shape.
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_mdx tdx_mdx;
typedef struct tdx_mdx_config {
    int32_t num_blocks;   /* L = 11 (AudioProcessor.py:241) */
    int32_t l;            /* convolutions per TFC block (3) */
    int32_t g;            /* channel growth per level (32; multiple of 32) */
    int32_t k;            /* TFC kernel size; must be 3 */
    int32_t bn;           /* TDF bottleneck factor (8) */
    int32_t dim_f;        /* 3072 */
    int32_t dim_t;        /* frames per block: 256 (the reference passes dim_t = 8 meaning 2^8) */
    int32_t reserved;
} tdx_mdx_config;
int tdx_mdx_create(const tdx_mdx_config* cfg, const void* weights_blob, size_t blob_bytes, int device, tdx_mdx** out);
int tdx_mdx_destroy(tdx_mdx* h);
size_t tdx_mdx_workspace_bytes(const tdx_mdx* h, int B);
double tdx_mdx_flops(const tdx_mdx* h, int B);
int tdx_mdx_forward(tdx_mdx* h, const float* spec_dev, int B, float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * N3  CT-Transformer punctuation restorer — replaces the neural forward of `self.punc.inference(input=text)`
 *     ASRProcessor.punctuation_restore  ASRProcessor.py:880-897  (funasr CTTransformer.punc_forward: third-party):
 *     Embedding(vocab, 256) -> SANM encoder (num_blocks layers, d 256, 8 heads, FFN 1024, FSMN k = 11) -> Linear(256, npunc).
 *     blob: TDXW container with funasr's names (embed.weight, encoder.encoders0.0.*, encoder.encoders.{i}.*, encoder.after_norm.*,
 *     decoder.*).  ids_dev int32 [B,T] (T <= 1024) -> logits_dev [B,T,npunc].  lens_dev: null (every row has T tokens) or int32 [B]:
 *     row b holds lens[b] <= T tokens, the rest is padding — masked in the attention and zero in the FSMN memory, like funasr's
 *     padding mask, so that the valid positions equal the unbatched result (rows of many texts in one launch sequence); logits of
 *     padding positions are unspecified.  The mini-sentence windows, the sentence cache and the text assembly are host logic
 *     (targetdiarization_amd/punctuation.py).
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_punc tdx_punc;
int tdx_punc_create(int num_blocks, int vocab, int npunc, const void* weights_blob, size_t blob_bytes, int device, tdx_punc** out);
int tdx_punc_destroy(tdx_punc* h);
size_t tdx_punc_workspace_bytes(const tdx_punc* h, int B, int T);
int tdx_punc_forward(tdx_punc* h, const int* ids_dev, const int* lens_dev, int B, int T, float* logits_dev, void* workspace_dev, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------------------------
 * a12 Paraformer-large SANM encoder — replaces the encoder forward inside
 *     `self.asr['paraformer'].generate(input=wav, hotword=...)`  ASRProcessor.py:424
 *     (funasr SANMEncoder: third-party; 1+49 pre-LN layers, d=512, 4 heads, FSMN memory k=11).
 *     blob: TDXW container with funasr's names (encoder.encoders0.0.*, encoder.encoders.{i}.*,
 *     encoder.after_norm.*).  feats_dev [B,T,560] (tdx_lfr_cmvn output) -> out_dev [B,T,512].
 *     lens_host: NULL, or B lengths that must all equal T (callers bucket segments by length,
 *     as for the separator; no padding mask is implemented).
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_pfenc tdx_pfenc;
int tdx_pfenc_create(int num_blocks, const void* weights_blob, size_t blob_bytes, int device, tdx_pfenc** out);
int tdx_pfenc_destroy(tdx_pfenc* h);
size_t tdx_pfenc_workspace_bytes(const tdx_pfenc* h, int B, int T);
double tdx_pfenc_flops(const tdx_pfenc* h, int B, int T);
int tdx_pfenc_forward(tdx_pfenc* h, const float* feats_dev, const int* lens_host, int B, int T,
                      float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * N2  Paraformer CIF predictor + non-autoregressive SANM decoder — replaces the rest of
 *     `self.asr['paraformer'].generate(input=wav, hotword=...)` after the encoder  ASRProcessor.py:424
 *     (funasr CifPredictorV2 + ParaformerSANMDecoder: third-party; 16 layers FFN -> FSMN memory -> cross attention,
 *     one FFN-only layer, LayerNorm, vocabulary projection).  blob: TDXW container with funasr's names
 *     (predictor.cif_conv1d / cif_output, decoder.decoders.{i}.*, decoder.decoders3.0.*, decoder.after_norm,
 *     decoder.output_layer).  Two calls because the number of tokens is data dependent:
 *       predict: enc_dev [B,T,512] -> alphas_dev [B,T+1] (tail frame appended), emb_dev [B,T+1,512] (the fired frames
 *                first, zero beyond), counts_dev int32 [B] = floor(sum alphas), peaks_dev int32 [B,T+1] (frame of fire k, -1)
 *       decode : the first L rows of every utterance of emb_dev (L = max count, read back by the host), masked by
 *                counts_dev, with enc_dev as attention memory -> ids_dev int32 [B,L] (argmax), score_dev [B,L] (log-softmax
 *                at the argmax) or NULL.
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_pfdec tdx_pfdec;
int tdx_pfdec_create(int num_blocks, int vocab, const void* weights_blob, size_t blob_bytes, int device, tdx_pfdec** out);
int tdx_pfdec_destroy(tdx_pfdec* h);
size_t tdx_pfdec_predict_workspace_bytes(const tdx_pfdec* h, int B, int T);
int tdx_pfdec_predict(tdx_pfdec* h, const float* enc_dev, int B, int T, float* alphas_dev, float* emb_dev, int* counts_dev,
                      int* peaks_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
size_t tdx_pfdec_decode_workspace_bytes(const tdx_pfdec* h, int B, int L, int T);
int tdx_pfdec_decode(tdx_pfdec* h, const float* emb_dev, int emb_rows, const int* counts_dev, const float* enc_dev, int B, int L,
                     int T, int* ids_dev, float* score_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * a10 ERes2NetV2-w24s4ep4 speaker embedding — replaces
 *     `self.embedding[embedding_model](wav_file, output_emb=True)['embs']`  TargetASR.py:161
 *     (modelscope / 3D-Speaker ERes2NetV2: third-party).  blob: TDXW container with the
 *     3D-Speaker module names (conv1, bn1, layer{1..4}.{i}.*, layer3_ds, fuse34, seg_1).
 *     feat_dev [B,F,80] = tdx_fbank mode-0 output (mean-normalised fbank) -> emb_dev [B,192].
 *     All B utterances share F (bucket by length); F >= 9.
 * ---------------------------------------------------------------------------------- */
typedef struct tdx_eres2net tdx_eres2net;
int tdx_eres2net_create(const void* weights_blob, size_t blob_bytes, int device, tdx_eres2net** out);
int tdx_eres2net_destroy(tdx_eres2net* h);
size_t tdx_eres2net_workspace_bytes(const tdx_eres2net* h, int B, int F);
double tdx_eres2net_flops(const tdx_eres2net* h, int B, int F);
int tdx_eres2net_forward(tdx_eres2net* h, const float* feat_dev, int B, int F, float* emb_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * a11  cosine scoring — replaces TargetASR.cosine_similarity  TargetASR.py:144-152
 *      emb_dev [N,D] f32, ref_dev [D] f32 -> scores_dev [N] f32 (1.0 if either vector is
 *      all-zero, else cos clipped to [0,1]).
 * ---------------------------------------------------------------------------------- */
int tdx_cosine_scores(const float* emb_dev, const float* ref_dev, int N, int D, float* scores_dev,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * N3   integrated loudness (ITU-R BS.1770-4) — replaces AudioProcessor.meter_loudness
 *      AudioProcessor.py:1123-1127 (pyloudnorm.Meter(rate).integrated_loudness, third-party) on
 *      device-resident clips: wav_dev [B,N] f32 mono -> lufs_dev [B] f64 (-inf for silence).
 *      N must cover one 400 ms gating block (the reference raises ValueError otherwise).
 * ---------------------------------------------------------------------------------- */
size_t tdx_loudness_workspace_bytes(int B, long N, int rate);
int tdx_loudness(const float* wav_dev, int B, long N, int rate, double* lufs_dev, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * N3   rational polyphase resampler — replaces AudioProcessor.audio_resample  AudioProcessor.py:549-569
 *      (librosa.resample, third-party; parity unpinned: the filter is the caller's — the Python host designs a
 *      Kaiser-windowed sinc like scipy.signal.resample_poly).  x_dev [C][n_in] -> y_dev [C][n_out],
 *      n_out = ceil(n_in*up/down); h_dev: 2*half+1 taps with DC gain `up`; the filter delay is removed.
 * ---------------------------------------------------------------------------------- */
int tdx_resample_poly(const float* x_dev, long n_in, int C, int up, int down, const float* h_dev, int half,
                      float* y_dev, long n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDX_H */
