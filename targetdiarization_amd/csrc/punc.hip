// punc.hip — CT-Transformer punctuation restorer on MI355X: the neural forward behind
// `self.punc.inference(input=text)` (ASRProcessor.punctuation_restore, ASRProcessor.py:880-897; funasr CTTransformer: third-party,
// parity unpinned — oracle/punc_oracle.py restates the published punc_forward):
//     x = Embedding(ids) ; h = SANMEncoder(x) ; y = Linear(h)          (embed_unit = att_unit = 256, 8 heads, FFN 1024, 4 blocks,
//                                                                       FSMN memory k = 11, sinusoidal PE, pre-LN, eps 1e-12)
// The sequences are mini-sentences of <= a few hundred tokens (funasr splits the text in windows of 20 tokens and carries the
// unfinished sentence over), so every launch is tiny: the Linears run on the exact fp32 MFMA core (gemm.hpp), the rest are small
// one-purpose kernels.  ids_dev int32 [B,T] -> logits_dev [B,T,npunc] (argmax / the mini-sentence logic stay on the host: punctuation.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/tdx.h"
#include "gemm.hpp"
#include "devutil.hpp"
#include "tdx_common.hpp"

using namespace tdx;

namespace {

constexpr int PD = 256, PH = 8, PDK = 32, PFFN = 1024, PKS = 11, PMAXT = 1024;
inline size_t al(size_t n) { return (n + 63) / 64 * 64; }
inline int up(int n, int m) { return (n + m - 1) / m * m; }
#define LAUNCH_CHECK()                                    \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return tdx::fail_hip(e__, __FILE__, __LINE__); \
    } while (0)
#define TRY(x) do { int rc__ = (x); if (rc__ != TDX_OK) return rc__; } while (0)

// x[b,t,:] = emb[id] * sqrt(256) + PE(t + 1)        (SinusoidalPositionEncoder: [sin | cos] halves, positions from 1)
__global__ __launch_bounds__(256) void punc_embed_kernel(const int* __restrict__ ids, const float* __restrict__ emb, int vocab, float* __restrict__ x, int T) {
    const int row = blockIdx.x, d = threadIdx.x;           // 256 threads = 256 channels
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int t = row % T;
    const int j = d & 127;
    const float inv = expf((float)j * (-logf(10000.0f) / (float)(PD / 2 - 1)));
    const float ang = (float)(t + 1) * inv;
    x[(long)row * PD + d] = emb[(long)id * PD + d] * 16.0f + (d < 128 ? sinf(ang) : cosf(ang));
}
// LayerNorm over 256 channels, one wave per row
__global__ __launch_bounds__(256) void punc_ln_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y,
                                                       long rows, float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4 v = *reinterpret_cast<const float4*>(x + row * PD + 4 * lane);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / PD);
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    const float var = wave_sum(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) * (1.0f / PD);
    const float r = 1.0f / sqrtf(var + eps);
    const float4 gg = *reinterpret_cast<const float4*>(g + 4 * lane), bb = *reinterpret_cast<const float4*>(b + 4 * lane);
    *reinterpret_cast<float4*>(y + row * PD + 4 * lane) = make_float4(a0 * r * gg.x + bb.x, a1 * r * gg.y + bb.y, a2 * r * gg.z + bb.z, a3 * r * gg.w + bb.w);
}
// FSMN memory: mem[b,t,c] = v[b,t,c] + sum_j w[c][j] v[b,t + j - 5,c]   (depthwise over time, zero padded; v = qkv[:, 512:768])
__global__ __launch_bounds__(256) void punc_fsmn_kernel(const float* __restrict__ qkv, const float* __restrict__ wT /* [11][256] */, float* __restrict__ mem, int T,
                                                       const int* __restrict__ lens) {
    const int row = blockIdx.x, c = threadIdx.x;
    const int t = row % T;
    const int L = lens ? min(lens[row / T], T) : T;          // rows >= L of a batch entry are padding: they read as zero (funasr: inputs * mask)
    const long base = (long)(row - t) * (3 * PD) + 2 * PD + c;
    float acc = t < L ? qkv[base + (long)t * (3 * PD)] : 0.f;
#pragma unroll
    for (int j = 0; j < PKS; ++j) {
        const int tt = t + j - (PKS - 1) / 2;
        if (tt >= 0 && tt < L) acc = fmaf(wT[j * PD + c], qkv[base + (long)tt * (3 * PD)], acc);
    }
    mem[(long)row * PD + c] = acc;
}
// softmax(q k^T / sqrt(32)) v for 8 query rows of one (batch, head); T <= PMAXT
__global__ __launch_bounds__(256) void punc_attn_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int T, const int* __restrict__ lens) {
    __shared__ float qs[8][PDK];
    __shared__ float p[8][PMAXT];
    const int b = blockIdx.z, h = blockIdx.y, r0 = blockIdx.x * 8, tid = threadIdx.x;
    const float* base = qkv + (long)b * T * (3 * PD);
    const int L = lens ? min(lens[b], T) : T;                // keys >= L are padding (masked); L = 0: the context rows are written as zeros
    { const int r = tid >> 5, d = tid & 31; qs[r][d] = base[(long)min(r0 + r, T - 1) * (3 * PD) + h * PDK + d] * 0.17677669529663687f; }
    __syncthreads();
    for (int j = tid; j < L; j += 256) {
        const float* kp = base + (long)j * (3 * PD) + PD + h * PDK;
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll
        for (int d = 0; d < PDK; ++d) {
            const float kv = kp[d];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = fmaf(qs[r][d], kv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r][j] = acc[r];
    }
    __syncthreads();
    {
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            float* row = p[2 * w + rr];
            float mx = -INFINITY;
            for (int c = lane; c < L; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = 0.f;
            for (int c = lane; c < L; c += 64) { const float e = expf(row[c] - mx); row[c] = e; sum += e; }
            sum = wave_sum(sum);
            const float inv = L > 0 ? 1.0f / sum : 0.f;
            for (int c = lane; c < L; c += 64) row[c] *= inv;
        }
    }
    __syncthreads();
    {
        const int r = tid >> 5, d = tid & 31;
        const float* vp = base + 2 * PD + h * PDK + d;
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(p[r][j], vp[(long)j * (3 * PD)], acc);
        if (r0 + r < T) ctx[((long)b * T + r0 + r) * PD + h * PDK + d] = acc;
    }
}

struct EpiB { const float* b; float* out; long ld; int nreal;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { if (n < nreal) out[(long)m * (int)ld + n] = v + c; } };
struct EpiBRelu { const float* b; float* out; long ld;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)m * (int)ld + n] = fmaxf(v + c, 0.f); } };
struct EpiBRes { const float* b; const float* mem; float* x;       // x += v + b (+ mem)
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { const long i = (long)m * PD + n; return mem ? x[i] + mem[i] : x[i]; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c, float a) const { x[(long)m * PD + n] = a + (v + c); } };

template <class Epi>
int lin(const float* A, long lda, const float* W, int M, int N, int K, Epi e, hipStream_t st) {
    GemmArgs g = make_args(M, N, make_seg(A, lda, W, K, K));
    if (launch_gemm<false, false, false, false>(g, 1, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

struct PLayer { size_t Wqkv, bqkv, fsmnT, Wo, bo, W1, b1, W2, b2, n1g, n1b, n2g, n2b; };

}  // namespace

struct tdx_punc {
    int device = 0, L = 0, vocab = 0, npunc = 0;
    float* dev = nullptr;
    size_t emb = 0, ang = 0, anb = 0, Wd = 0, bd = 0;
    std::vector<PLayer> layers;
};

extern "C" {

int tdx_punc_create(int num_blocks, int vocab, int npunc, const void* blob, size_t blob_bytes, int device, tdx_punc** out) {
    if (!blob || !out || num_blocks < 1 || vocab < 1 || npunc < 1 || npunc > 128) return tdx::fail(TDX_E_INVALID, "tdx_punc_create: bad argument");
    tdx::Blob bl;
    if (!bl.parse(blob, blob_bytes)) return tdx::fail(TDX_E_BLOB, "tdx_punc_create: malformed TDXW blob");
    std::vector<float> host;
    bool ok = true; std::string missing;
    auto get = [&](const std::string& name, size_t n) -> const float* {
        const tdx::BlobTensor* t = bl.find(name);
        if (!t || t->numel != n) { ok = false; if (missing.empty()) missing = name; return nullptr; }
        return t->data;
    };
    auto push = [&](const float* p, size_t n, size_t npad = 0) -> size_t {
        size_t o = host.size(); host.resize(o + al(std::max(n, npad)), 0.f);
        if (p) memcpy(host.data() + o, p, n * sizeof(float));
        return o;
    };
    tdx_punc* h = new tdx_punc();
    h->L = num_blocks; h->vocab = vocab; h->npunc = npunc;
    h->emb = push(get("embed.weight", (size_t)vocab * PD), (size_t)vocab * PD);
    for (int l = 0; l < num_blocks && ok; ++l) {
        const std::string p = l == 0 ? "encoder.encoders0.0." : "encoder.encoders." + std::to_string(l - 1) + ".";
        PLayer w;
        w.Wqkv = push(get(p + "self_attn.linear_q_k_v.weight", (size_t)3 * PD * PD), (size_t)3 * PD * PD);
        w.bqkv = push(get(p + "self_attn.linear_q_k_v.bias", 3 * PD), 3 * PD);
        {   // fsmn_block.weight [256][1][11] -> tap-major [11][256]
            const float* fw = get(p + "self_attn.fsmn_block.weight", (size_t)PD * PKS);
            w.fsmnT = host.size(); host.resize(host.size() + al((size_t)PKS * PD), 0.f);
            if (fw) for (int c = 0; c < PD; ++c) for (int j = 0; j < PKS; ++j) host[w.fsmnT + (size_t)j * PD + c] = fw[(size_t)c * PKS + j];
        }
        w.Wo = push(get(p + "self_attn.linear_out.weight", (size_t)PD * PD), (size_t)PD * PD);
        w.bo = push(get(p + "self_attn.linear_out.bias", PD), PD);
        w.W1 = push(get(p + "feed_forward.w_1.weight", (size_t)PFFN * PD), (size_t)PFFN * PD);
        w.b1 = push(get(p + "feed_forward.w_1.bias", PFFN), PFFN);
        w.W2 = push(get(p + "feed_forward.w_2.weight", (size_t)PD * PFFN), (size_t)PD * PFFN);
        w.b2 = push(get(p + "feed_forward.w_2.bias", PD), PD);
        w.n1g = push(get(p + "norm1.weight", PD), PD); w.n1b = push(get(p + "norm1.bias", PD), PD);
        w.n2g = push(get(p + "norm2.weight", PD), PD); w.n2b = push(get(p + "norm2.bias", PD), PD);
        h->layers.push_back(w);
    }
    h->ang = push(get("encoder.after_norm.weight", PD), PD);
    h->anb = push(get("encoder.after_norm.bias", PD), PD);
    h->Wd = push(get("decoder.weight", (size_t)npunc * PD), (size_t)npunc * PD, (size_t)128 * PD);       // rows read in 128-row tiles
    h->bd = push(get("decoder.bias", npunc), npunc, 128);
    if (!ok) { delete h; return tdx::fail(TDX_E_BLOB, "tdx_punc_create: tensor missing or wrong size: " + missing); }
    { const std::string extra = bl.first_unused(); if (!extra.empty()) { delete h; return tdx::fail(TDX_E_BLOB, "tdx_punc_create: unexpected tensor in the blob: " + extra); } }
    tdx::DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    h->device = device;
    e = hipMalloc(&h->dev, host.size() * sizeof(float));
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    *out = h;
    return TDX_OK;
}

int tdx_punc_destroy(tdx_punc* h) {
    if (h) { if (h->dev) hipFree(h->dev); delete h; }
    return TDX_OK;
}

size_t tdx_punc_workspace_bytes(const tdx_punc* h, int B, int T) {
    if (!h || B < 1 || T < 1 || T > PMAXT) return 0;
    const size_t M = (size_t)B * T;
    return (al(M * PD) * 4 + al(M * 3 * PD) + al(M * PFFN) + 1024) * sizeof(float);
}

int tdx_punc_forward(tdx_punc* h, const int* ids, const int* lens, int B, int T, float* logits, void* ws_, size_t ws_bytes, void* stream) {
    if (!h || !ids || !logits || !ws_) return tdx::fail(TDX_E_INVALID, "tdx_punc_forward: null argument");
    if (B < 1 || T < 1 || T > PMAXT) return tdx::fail(TDX_E_INVALID, "tdx_punc_forward: need 1 <= T <= 1024");
    if (ws_bytes < tdx_punc_workspace_bytes(h, B, T)) return tdx::fail(TDX_E_WORKSPACE, "tdx_punc_forward: workspace too small");
    tdx::DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return tdx::fail_hip(guard.err, __FILE__, __LINE__);
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)B * T;
    float* ws = (float*)ws_;
    float* x = ws; float* hn = x + al(M * PD); float* mem = hn + al(M * PD); float* ctx = mem + al(M * PD);
    float* qkv = ctx + al(M * PD); float* ffn = qkv + al(M * 3 * PD);
    const float* d = h->dev;
    hipLaunchKernelGGL(punc_embed_kernel, dim3((unsigned)M), dim3(256), 0, st, ids, d + h->emb, h->vocab, x, T);
    LAUNCH_CHECK();
    for (const PLayer& w : h->layers) {
        hipLaunchKernelGGL(punc_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, d + w.n1g, d + w.n1b, hn, M, 1e-12f);
        LAUNCH_CHECK();
        TRY(lin(hn, PD, d + w.Wqkv, (int)M, 3 * PD, PD, EpiB{d + w.bqkv, qkv, 3 * PD, 3 * PD}, st));
        hipLaunchKernelGGL(punc_fsmn_kernel, dim3((unsigned)M), dim3(256), 0, st, qkv, d + w.fsmnT, mem, T, lens);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(punc_attn_kernel, dim3((T + 7) / 8, PH, B), dim3(256), 0, st, qkv, ctx, T, lens);
        LAUNCH_CHECK();
        TRY(lin(ctx, PD, d + w.Wo, (int)M, PD, PD, EpiBRes{d + w.bo, mem, x}, st));                 // x += att W_o + b_o + mem
        hipLaunchKernelGGL(punc_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, d + w.n2g, d + w.n2b, hn, M, 1e-12f);
        LAUNCH_CHECK();
        TRY(lin(hn, PD, d + w.W1, (int)M, PFFN, PD, EpiBRelu{d + w.b1, ffn, PFFN}, st));
        TRY(lin(ffn, PFFN, d + w.W2, (int)M, PD, PFFN, EpiBRes{d + w.b2, nullptr, x}, st));
    }
    hipLaunchKernelGGL(punc_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, d + h->ang, d + h->anb, hn, M, 1e-12f);
    LAUNCH_CHECK();
    TRY(lin(hn, PD, d + h->Wd, (int)M, 128, PD, EpiB{d + h->bd, logits, h->npunc, h->npunc}, st));
    return TDX_OK;
}

}  // extern "C"
