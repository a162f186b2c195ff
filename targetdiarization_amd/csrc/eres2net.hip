// eres2net.hip — ERes2NetV2-w24s4ep4 speaker-embedding extractor on MI355X.
// Replaces `self.embedding['eres2netv2_large'](wav, output_emb=True)['embs']` (TargetASR.py:161;
// modelscope / 3D-Speaker: third-party, parity unpinned — oracle/eres2netv2_oracle.py).
// feat [B,F,80] (fbank minus utterance mean, tdx_fbank mode 0) -> embedding [B,192].
//
// Layout: NHWC fp32, rows = (b, freq, time), channels contiguous and padded to a multiple of
// 32.  The convolutions are implicit GEMMs (CONV mode: one K segment per tap, the A-loader computes the
// shifted/strided input row and zero-fills the halo) — stage 1's 1x1 convolutions on the fp32-MFMA core (gemm.hpp),
// stages 2-4 on the split-f16 x3 core (gemm_h3.hpp) over planes with a static scale (ReLU20-bounded activations) —
// except the 3x3 chain convolutions of stages 1-2, which run on an LDS-tiled direct x3 convolution (chain_conv_x3_kernel).
// Eval-mode BatchNorm is folded into the weights on the host; ReLU20 / residual / Res2Net chain add / AFF gate are
// fused into the epilogues.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/tdx.h"
#include "gemm.hpp"
#include "gemm_h3.hpp"
#include "devutil.hpp"
#include "tdx_common.hpp"

using namespace tdx;

namespace {

constexpr int NSTAGE = 4, SCALE = 4, EMB = 192;
const int kBlocks[NSTAGE] = {3, 4, 6, 3};
const int kPlanes[NSTAGE] = {64, 128, 256, 512};
const int kStride[NSTAGE] = {1, 2, 2, 2};

inline size_t al(size_t n) { return (n + 63) / 64 * 64; }
inline int up(int n, int m) { return (n + m - 1) / m * m; }

#define LAUNCH_CHECK()                                    \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return tdx::fail_hip(e__, __FILE__, __LINE__); \
    } while (0)
#define TRY(x) do { int rc__ = (x); if (rc__ != TDX_OK) return rc__; } while (0)

__device__ __forceinline__ float relu20(float v) { return fminf(fmaxf(v, 0.f), 20.f); }

// stem: conv3x3(1->64, pad 1) + BN + ReLU on x[b, h=freq, w=time] = feat[b, w, h]
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ feat, const float* __restrict__ w9,   // [9][64]
                                                    const float* __restrict__ bias, float* __restrict__ out, int F, long rows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // (row, channel quad): 16 quads per row
    if (i >= rows * 16) return;
    const long m = i >> 4;
    const int c = (int)(i & 15) * 4;
    const int w = (int)(m % F), h = (int)((m / F) % 80), b = (int)(m / ((long)F * 80));
    float4 acc = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ih = h + t / 3 - 1, iw = w + t % 3 - 1;
        if (ih >= 0 && ih < 80 && iw >= 0 && iw < F) {
            const float x = feat[((long)b * F + iw) * 80 + ih];
            const float4 k = *reinterpret_cast<const float4*>(w9 + t * 64 + c);
            acc.x = fmaf(k.x, x, acc.x); acc.y = fmaf(k.y, x, acc.y); acc.z = fmaf(k.z, x, acc.z); acc.w = fmaf(k.w, x, acc.w);
        }
    }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    *reinterpret_cast<float4*>(out + m * 64 + c) = acc;
}

// TSTP over time: x [B, 10, T, 2048] -> stats[b][c*10+fq] (mean) , stats[b][20480 + c*10+fq] (std, unbiased)
__global__ __launch_bounds__(256) void tstp_kernel(const float* __restrict__ x, float* __restrict__ stats, int T, int C, int Fq) {
    const int fq = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0, q = 0.0;
        const float* p = x + (((long)b * Fq + fq) * T) * C + c;
        for (int t = 0; t < T; ++t) { const double v = p[(long)t * C]; s += v; q += v * v; }
        const double mean = s / T;
        double var = (q - s * mean) / (T - 1);
        if (var < 0) var = 0;
        stats[(long)b * 2 * C * Fq + (long)c * Fq + fq] = (float)mean;
        stats[(long)b * 2 * C * Fq + (long)C * Fq + (long)c * Fq + fq] = (float)sqrt(var + 1e-7);
    }
}

// seg_1: emb[b][n] = bias[n] + stats[b] . W[n]      (K = 40960)
__global__ __launch_bounds__(256) void seg_kernel(const float* __restrict__ stats, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ emb, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* s = stats + (long)b * K;
    const float* w = W + (long)n * K;
    float acc = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
        const float4 a = *reinterpret_cast<const float4*>(s + k);
        const float4 c = *reinterpret_cast<const float4*>(w + k);
        acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) emb[(long)b * gridDim.x + n] = (red[0] + red[1]) + (red[2] + red[3]) + bias[n];
}

// ---------------------------------------------------------------- epilogues
struct EpiRelu20 {      // clamp(v + b, 0, 20), columns < nreal
    const float* b; float* out; long ld; int nreal;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { if (n < nreal) out[(long)m * (int)ld + n] = relu20(v + c); }
};
struct EpiBiasG {       // v + b, columns < nreal
    const float* b; float* out; long ld; int nreal;
    __device__ float col(int, int n) const { return b ? b[n] : 0.f; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { if (n < nreal) out[(long)m * (int)ld + n] = v + c; }
};
struct EpiSlabZ {        // split over the taps: partial sums of batch z -> slab[z][m][n]
    float* slab; long ld; long strideZ;
    __device__ EpiNone col(int, int) const { return EpiNone{}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int z, int m, int n, float v, EpiNone, EpiNone) const { slab[(long)z * strideZ + (long)m * (int)ld + n] = v; }
};
// out[m][n] = sum_z slab[z][m][n] (+ bias[n])
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nz, long MN, int N, const float* __restrict__ bias, float* __restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    float4 a = *reinterpret_cast<const float4*>(slab + i);
    for (int z = 1; z < nz; ++z) {
        const float4 p = *reinterpret_cast<const float4*>(slab + (long)z * MN + i);
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + (int)(i % N)); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *reinterpret_cast<float4*>(out + i) = a;
}
struct EpiChain {       // Res2Net chain: sp = relu20(v+b) -> cat[:, coff+n]; next input sp + spx[i+1] -> spin
    const float* b; float* cat; long ldcat; int coff; int width; int wpad;
    const float* o1; long ldo1; int next_off; float* spin;      // spin == nullptr: no plain-add successor
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { return (spin && n < width) ? o1[(long)m * (int)ldo1 + next_off + n] : 0.f; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c, float nxt) const {
        if (n >= wpad) return;
        const float sp = relu20(v + c);
        if (n < width) cat[(long)m * (int)ldcat + coff + n] = sp;
        if (spin) spin[(long)m * wpad + n] = n < width ? sp + nxt : 0.f;
    }
};
struct EpiConv3 {       // relu20(v + b + residual)
    const float* b; const float* res; float* out; long ld;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { return res[(long)m * (int)ld + n]; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c, float r) const { out[(long)m * (int)ld + n] = relu20(v + c + r); }
};
struct EpiAffSilu {     // t = silu(v + b), columns < ipad
    const float* b; float* t; int ipad;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { if (n < ipad) t[(long)m * ipad + n] = siluf_acc(v + c); }
};
struct EpiAffGate {     // att = 1 + tanh(v+b); out = x*att + y*(2-att), columns < C
    const float* b; const float* x; long ldx; const float* y; long ldy; float* out; long ldo; int C;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float2 aux(int, int m, int n, EpiNone) const {
        return n < C ? make_float2(x[(long)m * (int)ldx + n], y[(long)m * (int)ldy + n]) : make_float2(0.f, 0.f);
    }
    __device__ void store(int, int m, int n, float v, EpiNone, float c, float2 xy) const {
        if (n >= C) return;
        const float att = 1.0f + tanhf(v + c);
        out[(long)m * (int)ldo + n] = xy.x * att + xy.y * (2.0f - att);
    }
};


// ---------------------------------------------------------------------------------------------------------
// Res2Net chain convolution of stages 1-2 (3x3, C -> C channels, stride 1, C = 24 / 48) as an LDS-tiled direct convolution on
// the split-f16 x3 MFMA scheme.  As an implicit GEMM on the 256 x 256 tiles of gemm_h3.hpp these launches fill 48 of 256 tile
// columns and re-read every input pixel nine times from L2 (stage 1 ran on a fp32-MFMA direct convolution until round 3); here
// a block stages the (TH + 2) x 34 pixel patch ONCE, splitting the fp32 input into hi/lo f16 with the static scale 2^9 on the
// way (inputs are sums of two ReLU20 outputs: no separate split pass, no planes in HBM), moves the weight planes one tap at a
// time through a ring of register slots into a double-buffered LDS tile and feeds v_mfma_f32_32x32x16_f16 (3 passes:
// hi.lo + lo.hi + hi.hi, the order of gemm_h3.hpp: results are bit-identical to the implicit GEMM) from LDS: wave w owns MT
// image rows of 32 pixels (= 32-row MFMA tiles) and all NT 32-column tiles.  LDS images: [pixel | n][channel / 8][hi 16 B |
// lo 16 B] with a pitch that is an odd multiple of 16 B (conflict-free ds_read_b128 over consecutive pixels / columns).
// Epilogue = EpiChain.  Measured (B = 60, F = 998): C = 48: 422 us per launch (implicit GEMM 0.9 ms + 60 us split pass), of
// which patch reads ~145, MFMA loop ~120, epilogue ~140 us, hardly overlapped — 1.26 GB per launch, the two memory phases each
// run at 3.4-5.5 TB/s; C = 24 (TH = 4, four blocks per CU): 703 us for 2.5 GB (fp32-MFMA direct convolution: 1112 us).
// With C = 96 (stage 3; 158 KB of LDS, one block per CU) the kernel lost to the implicit GEMM (0.50 vs 0.34 ms): not instantiated.
// ---------------------------------------------------------------------------------------------------------
struct ChainX3Args {
    const float* in; long ldin;                         // NHWC input [B*H*W][ldin], channels 0..CIN-1
    const unsigned char* wp; const float* ws; int wk;   // weight planes [n][wk / 8][2][8] f16 (wk = 9 * cinp), inverse row scales ws[n]
    int cinp;
    const float* b;
    float* cat; long ldcat; int coff;                   // sp -> cat[m*ldcat + coff + n], n < CIN
    const float* o1; long ldo1; int next_off; float* spin; int ldspin;      // spin (or null): spin[m*ldspin + n] = n < CIN ? sp + o1[m*ldo1 + next_off + n] : 0
    int H, W;
};
template <int CIN, int TH>
struct ChainX3Cfg {
    static constexpr int KS = (CIN + 15) / 16, CP = KS * 16, PPB = CP * 4 + 16, NT = (CIN + 31) / 32, NB = NT * 32, MT = TH / 4;
    static constexpr int PATCH = (TH + 2) * 34 * PPB, BT = NB * PPB, LDS = PATCH + 2 * BT;
    static constexpr int BQ = NB * (CP / 8) * 2;         // 16-B pieces of one tap's weight tile
    static constexpr int BPT = (BQ + 255) / 256;
};
template <int CIN, int TH>
__global__ __launch_bounds__(256) void chain_conv_x3_kernel(ChainX3Args a) {
    using C = ChainX3Cfg<CIN, TH>;
    constexpr int KS = C::KS, CP = C::CP, PPB = C::PPB, NT = C::NT, MT = C::MT;
    static_assert(PPB % 32 == 16 && TH % 4 == 0 && CIN % 4 == 0, "pitch = odd multiple of 16 B");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const patch = lds;
    unsigned char* const bt = lds + C::PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TH, b = blockIdx.z;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    // ---- weight tiles: global planes -> a ring of three register slots (16-B pieces (n, chunk, hi|lo)) -> the double-buffered LDS
    // tile.  Tap t + 4 is requested in iteration t and written to LDS in iteration t + 3: no global latency inside the tap loop.
    // (literal tap numbers, native vector type: the array stays in registers — HIP's uint4 struct is copied by memcpy, which
    // kept it in scratch)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 breg[3][C::BPT];
#define TDX_CHAIN_BLOAD(tap_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < C::BPT; ++i) {                                                               \
        const int t = tid + i * 256;                                                                                   \
        if (C::BQ % 256 == 0 || t < C::BQ) {                                                                           \
            const int n = t / (CP / 4), r = t - n * (CP / 4);       /* r = 2 * chunk + (lo ? 1 : 0) */                 \
            breg[(tap_) % 3][i] = *reinterpret_cast<const u32x4*>(a.wp + (long)n * a.wk * 4 + ((long)(tap_) * a.cinp / 8) * 32 + r * 16); \
        }                                                                                                              \
    }
#define TDX_CHAIN_BSTORE(tap_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < C::BPT; ++i) {                                                               \
        const int t = tid + i * 256;                                                                                   \
        if (C::BQ % 256 == 0 || t < C::BQ) {                                                                           \
            const int n = t / (CP / 4), r = t - n * (CP / 4);                                                          \
            *reinterpret_cast<u32x4*>(bt + ((tap_) & 1) * C::BT + n * PPB + r * 16) = breg[(tap_) % 3][i];             \
        }                                                                                                              \
    }
    TDX_CHAIN_BLOAD(0) TDX_CHAIN_BLOAD(1) TDX_CHAIN_BLOAD(2)
    // ---- the input patch (zero outside the image and for the channels >= CIN): every load of the thread in flight at once
    constexpr int SLOTS = (TH + 2) * 34 * (CP / 4), SPT = (SLOTS + 255) / 256;
    float4 sv[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const int t = tid + i * 256;
        const int p = t / (CP / 4), q = t - p * (CP / 4);
        const int py = p / 34, px = p - py * 34;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        sv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < SLOTS && 4 * q < CIN && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            sv[i] = *reinterpret_cast<const float4*>(a.in + (((long)b * a.H + iy) * a.W + ix) * a.ldin + 4 * q);
    }
    // ---- the epilogue's operands (bias, scale, the next chunk of conv1's output) do not depend on the accumulators: requested now
    float bias[NT], sc[NT], nx[NT][MT][16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 32 + l31;
        const bool real = c < CIN;
        bias[n] = real ? a.b[c] : 0.f;
        sc[n] = real ? a.ws[c] * (1.0f / 512.0f) : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = min(y0 + MT * wave + m, a.H - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int x = min(x0 + (i & 3) + 8 * (i >> 2) + 4 * h, a.W - 1);
                nx[n][m][i] = (a.spin && real) ? a.o1[(((long)b * a.H + y) * a.W + x) * a.ldo1 + a.next_off + c] : 0.f;
            }
        }
    }
    // ---- split the patch with the static scale 2^9 into the LDS image
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const int t = tid + i * 256;
        const int p = t / (CP / 4), q = t - p * (CP / 4);
        const float xs[4] = {sv[i].x * 512.0f, sv[i].y * 512.0f, sv[i].z * 512.0f, sv[i].w * 512.0f};
        h4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const _Float16 u = (_Float16)xs[j]; hi[j] = u; lo[j] = (_Float16)(xs[j] - (float)u); }
        unsigned char* d = patch + p * PPB + (q >> 1) * 32 + (q & 1) * 8;
        if (SLOTS % 256 == 0 || t < SLOTS) {
            *reinterpret_cast<h4*>(d) = hi;
            *reinterpret_cast<h4*>(d + 16) = lo;
        }
    }
    TDX_CHAIN_BSTORE(0) TDX_CHAIN_BLOAD(3)
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    auto tap_mfma = [&](const unsigned char* bb, const unsigned char* ab) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = *reinterpret_cast<const f16x8*>(ab + m * 34 * PPB + ks * 64);
                al[m] = *reinterpret_cast<const f16x8*>(ab + m * 34 * PPB + ks * 64 + 16);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bh[n] = *reinterpret_cast<const f16x8*>(bb + n * 32 * PPB + ks * 64);
                bl[n] = *reinterpret_cast<const f16x8*>(bb + n * 32 * PPB + ks * 64 + 16);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    };
#define TDX_CHAIN_TAP(tap_) tap_mfma(bt + ((tap_) & 1) * C::BT + l31 * PPB + h * 32, patch + ((MT * wave + (tap_) / 3) * 34 + l31 + (tap_) % 3) * PPB + h * 32);
    // (the LDS buffer a tap's successor is written to was last read in iteration tap - 1, before its barrier)
    TDX_CHAIN_BSTORE(1) TDX_CHAIN_BLOAD(4) TDX_CHAIN_TAP(0) __syncthreads();
    TDX_CHAIN_BSTORE(2) TDX_CHAIN_BLOAD(5) TDX_CHAIN_TAP(1) __syncthreads();
    TDX_CHAIN_BSTORE(3) TDX_CHAIN_BLOAD(6) TDX_CHAIN_TAP(2) __syncthreads();
    TDX_CHAIN_BSTORE(4) TDX_CHAIN_BLOAD(7) TDX_CHAIN_TAP(3) __syncthreads();
    TDX_CHAIN_BSTORE(5) TDX_CHAIN_BLOAD(8) TDX_CHAIN_TAP(4) __syncthreads();
    TDX_CHAIN_BSTORE(6) TDX_CHAIN_TAP(5) __syncthreads();
    TDX_CHAIN_BSTORE(7) TDX_CHAIN_TAP(6) __syncthreads();
    TDX_CHAIN_BSTORE(8) TDX_CHAIN_TAP(7) __syncthreads();
    TDX_CHAIN_TAP(8)
#undef TDX_CHAIN_TAP
#undef TDX_CHAIN_BSTORE
#undef TDX_CHAIN_BLOAD
    // ---- epilogue (EpiChain): D col = l31 (output channel of tile n), row = (r&3) + 8*(r>>2) + 4*h (pixel of the image row)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c = n * 32 + l31;
        const bool real = c < CIN;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = y0 + MT * wave + m;
            if (y >= a.H) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int x = x0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (x >= a.W) continue;
                const long mrow = ((long)b * a.H + y) * a.W + x;
                const float sp = relu20(acc[m][n][i] * sc[n] + bias[n]);
                if (real) a.cat[mrow * a.ldcat + a.coff + c] = sp;
                if (a.spin && c < a.ldspin) a.spin[mrow * a.ldspin + c] = real ? sp + nx[n][m][i] : 0.f;
            }
        }
    }
}
template <int CIN, int TH>
int launch_chain_x3(const ChainX3Args& a, int B, hipStream_t st) {
    using C = ChainX3Cfg<CIN, TH>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)chain_conv_x3_kernel<CIN, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (attr_rc != hipSuccess) return tdx::fail_hip(attr_rc, __FILE__, __LINE__);
    hipLaunchKernelGGL((chain_conv_x3_kernel<CIN, TH>), dim3((a.W + 31) / 32, (a.H + TH - 1) / TH, B), dim3(256), C::LDS, st, a);
    LAUNCH_CHECK();
    return TDX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// AFF of the Res2Net chain in stages 3-4 (C = 96 / 192 channels) as ONE launch: t = silu(W0 [x | y] + b0) (C/4 channels),
// att = 1 + tanh(W3 t + b3), out = x att + y (2 - att).  As two launches of the fp32-MFMA GEMM core the pair is bound by tile
// latency (K = 2C: six k steps per 128-row tile, 384 us per AFF for 20 MB of operands at B = 60); here a block owns 128 rows
// (wave w the rows 32 w ..), streams [x | y] and W0 in chunks of 32 k through double-buffered LDS tiles (next chunk in registers
// while the MFMAs of the current one run), keeps t in LDS for the second GEMM (W3 resident) and applies the gate from x and y.
// v_mfma_f32_32x32x2_f32 throughout: exact fp32 products like the GEMM core's.  LDS rows have a pitch of 4 mod 32 floats.
// ---------------------------------------------------------------------------------------------------------
struct AffArgs {
    const float* x; long ldx; const float* y; long ldy;
    const float* w0; int ld0; const float* b0;          // [ipad rows][2C] (pitch ld0), rows >= C/4 zero
    const float* w3; int ld3; const float* b3;          // [C rows][ipad] (pitch ld3), columns >= C/4 zero
    float* out; long ldo; long M;
};
template <int C>
struct AffCfg {
    static constexpr int IP = (C / 4 + 31) / 32 * 32, NT1 = IP / 32, NT2 = C / 32, NCH = 2 * C / 32;
    static constexpr int PA = 36, PT = IP + 4;
    static constexpr int AS = 128 * PA, WS = IP * PA;                       // floats per buffer
    static constexpr int TS = 128 * PT;
    static constexpr int REG1 = (2 * AS > TS ? 2 * AS : TS);                // the A ring, later t
    static constexpr int LDS = (REG1 + 2 * WS + C * PT) * 4;
    static constexpr int WPT = (IP * 8 + 255) / 256;                        // float4 of a W0 chunk per thread
};
template <int C>
__global__ __launch_bounds__(256, C == 96 ? 2 : 1) void aff_fused_kernel(AffArgs a) {
    using F = AffCfg<C>;
    constexpr int IP = F::IP, NT1 = F::NT1, NT2 = F::NT2, NCH = F::NCH, PA = F::PA, PT = F::PT;
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    float* const As = ldsf;                              // [2][128][PA]
    float* const Ws = ldsf + F::REG1;                    // [2][IP][PA]
    float* const W3s = Ws + 2 * F::WS;                   // [C][PT]
    float* const Ts = ldsf;                              // [128][PT] (after the first GEMM)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.x * 128;
    // ---- chunk kc of [x | y] (128 rows x 32 k) and of W0 (IP rows x 32 k): global -> registers -> LDS buffer
    f32x4 ra[4], rw[F::WPT];
    auto gload = [&](int kc) {
        const bool second = kc >= NCH / 2;
        const float* src = second ? a.y : a.x;
        const long ld = second ? a.ldy : a.ldx;
        const int k0 = (second ? kc - NCH / 2 : kc) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, r = idx >> 3, q = idx & 7;
            ra[i] = ldg4(src + (long)min(m0 + r, a.M - 1) * ld + k0 + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < F::WPT; ++i) {
            const int idx = tid + 256 * i, n = idx >> 3, q = idx & 7;
            if (IP * 8 % 256 == 0 || idx < IP * 8) rw[i] = ldg4(a.w0 + (long)n * a.ld0 + kc * 32 + 4 * q);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, r = idx >> 3, q = idx & 7;
            *reinterpret_cast<f32x4*>(As + buf * F::AS + r * PA + 4 * q) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < F::WPT; ++i) {
            const int idx = tid + 256 * i, n = idx >> 3, q = idx & 7;
            if (IP * 8 % 256 == 0 || idx < IP * 8) *reinterpret_cast<f32x4*>(Ws + buf * F::WS + n * PA + 4 * q) = rw[i];
        }
    };
    gload(0);
    // W3 resident: [C][IP] -> W3s[C][PT]
    for (int idx = tid; idx < C * (IP / 4); idx += 256) {
        const int n = idx / (IP / 4), q = idx - n * (IP / 4);
        *reinterpret_cast<f32x4*>(W3s + n * PT + 4 * q) = ldg4(a.w3 + (long)n * a.ld3 + 4 * q);
    }
    lstore(0);
    __syncthreads();
    f32x16 acc1[NT1];
#pragma unroll
    for (int n = 0; n < NT1; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[n][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < NCH; ++kc) {
        if (kc + 1 < NCH) gload(kc + 1);
        const float* const Ab = As + (kc & 1) * F::AS + (wave * 32 + l31) * PA + 4 * h;
        const float* const Wb = Ws + (kc & 1) * F::WS + l31 * PA + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + 8 * kk);
#pragma unroll
            for (int n = 0; n < NT1; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(Wb + n * 32 * PA + 8 * kk);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc1[n], 0, 0, 0);
            }
        }
        if (kc + 1 < NCH) lstore((kc + 1) & 1);       // (that buffer was last read in iteration kc - 1, before its barrier)
        __syncthreads();
    }
    // ---- t = silu(. + b0) -> Ts (the A ring is dead: every wave is past the last barrier); D col = l31, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
    for (int n = 0; n < NT1; ++n) {
        const float b0 = a.b0[n * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Ts[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * PT + n * 32 + l31] = siluf_acc(acc1[n][r] + b0);
    }
    // (each wave reads back only the rows it wrote: no barrier)
    f32x16 acc2[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
    {
        const float* const Tb = Ts + (wave * 32 + l31) * PT + 4 * h;
        const float* const Wb = W3s + l31 * PT + 4 * h;
#pragma unroll
        for (int kk = 0; kk < IP / 8; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(Tb + 8 * kk);
#pragma unroll
            for (int n = 0; n < NT2; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(Wb + n * 32 * PT + 8 * kk);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc2[n], 0, 0, 0);
            }
        }
    }
    // ---- gate: att = 1 + tanh(. + b3); out = x att + y (2 - att)
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int c = n * 32 + l31;
        const float b3 = a.b3[c];
        float xv[16], yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = min(m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, a.M - 1);
            xv[r] = a.x[m * a.ldx + c]; yv[r] = a.y[m * a.ldy + c];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { tdx::touch(xv[r]); tdx::touch(yv[r]); }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= a.M) continue;
            const float att = 1.0f + tanhf(acc2[n][r] + b3);
            a.out[m * a.ldo + c] = xv[r] * att + yv[r] * (2.0f - att);
        }
    }
}
template <int C>
int launch_aff_fused(const AffArgs& a, hipStream_t st) {
    using F = AffCfg<C>;
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)aff_fused_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS);
    if (attr_rc != hipSuccess) return tdx::fail_hip(attr_rc, __FILE__, __LINE__);
    hipLaunchKernelGGL((aff_fused_kernel<C>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), F::LDS, st, a);
    LAUNCH_CHECK();
    return TDX_OK;
}

// static-scale split (gemm_h3.hpp h3_split_rows_static) of the pixels a stride-2 1x1 convolution reads only: planes row
// (b, ho, wo) <- source pixel (b, 2 ho, 2 wo).  conv1 and the shortcut of a stage's first block then run as stride-1 GEMMs over
// a quarter of the rows; splitting every pixel of the previous stage moved 4x the bytes (9.8 GB at B = 60 for stage 2).
__global__ __launch_bounds__(256) void split_rows_static_s2_kernel(const float* __restrict__ x, int ld, unsigned char* __restrict__ planes,
                                                                  long R, int K, int Hin, int Win, int Ho, int Wo, float s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // (row, 8-value chunk)
    const int per = K / 8;
    if (i >= R * per) return;
    const long row = i / per;
    const int ch = (int)(i - row * per);
    const long b = row / ((long)Ho * Wo);
    const int r = (int)(row - b * Ho * Wo), ho = r / Wo, wo = r - ho * Wo;
    const float* src = x + ((b * Hin + 2 * ho) * Win + 2 * wo) * ld + ch * 8;
    const f32x4 a0 = ldg4(src), a1 = ldg4(src + 4);
    const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    h3_store_chunk(planes + row * (long)K * 4 + ch * 32, v, s);
}

struct ConvW {
    size_t w, b; int N, Npad, cin, cinp, taps;
    const unsigned char* hp = nullptr; const float* hs = nullptr;     // split-f16 planes [Npad][taps*cinp] + row scales (x3 core), if made
};
struct AffW { ConvW c0, c3; int C, inter, ipad; };
struct BlockW { ConvW conv1, convs[4], conv3, sc; bool has_sc; AffW aff[3]; int stride, cin, width, wpad, w4, cout; bool is_aff; };

template <class Epi>
int conv_gemm(const float* A, long lda, const float* dev, const ConvW& cw, int B, int Hin, int Win, int Hout, int Wout, int stride,
              Epi e, hipStream_t st) {
    const long M = (long)B * Hout * Wout;
    GemmArgs g = make_args((int)M, cw.Npad, make_seg(A, lda, dev + cw.w, (long)cw.taps * cw.cinp, cw.cinp));
    g.n_valid = up(cw.N, 32);
    g.cv_Hin = Hin; g.cv_Win = Win; g.cv_Hout = Hout; g.cv_Wout = Wout; g.cv_stride = stride; g.cv_ntaps = cw.taps; g.cv_cin = cw.cinp;
    if (launch_gemm<false, false, false, false, Epi, 0, true>(g, 1, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

// the same convolution on the split-f16 x3 core (gemm_h3.hpp, A_CONV): Ap = planes of the NHWC input with ONE scale
// (*inv_scale on the device) — activations here are ReLU20-bounded, and the taps of a 3x3 read different pixel rows into
// one accumulator row, so a per-row scale could not be used anyway
template <class Epi>
int conv_gemm_h3(const unsigned char* Ap, const float* inv_scale, const unsigned char* zero_row, const ConvW& cw, int B, int Hin, int Win,
                 int Hout, int Wout, int stride, Epi e, hipStream_t st) {
    const long M = (long)B * Hout * Wout;
    H3Args g{};
    g.seg[0] = h3_seg(Ap, inv_scale, 4L * cw.cinp, cw.hp, cw.hs, 4L * cw.taps * cw.cinp, cw.taps * cw.cinp);
    g.seg[0].sa_mul = 0;
    g.nseg = 1; g.M = (int)M; g.N = cw.Npad;
    g.cv_Hin = Hin; g.cv_Win = Win; g.cv_Hout = Hout; g.cv_Wout = Wout; g.cv_stride = stride; g.cv_ntaps = cw.taps; g.cv_cin = cw.cinp;
    g.zero_row = zero_row;
    if (launch_gemm_h3x<false, false, false, false, Epi, 0, true>(g, 1, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

// ... with the 9 taps split over 3 batches of 3 taps (few output rows: e.g. 72 tiles of 576 k-steps for two 8.7 s clips) into
// slab[3][M][N], then summed
constexpr long DS_SPLIT_ROWS = 4096;
constexpr int DS_SPLIT = 3;
inline int conv_gemm_h3_tapsplit(const unsigned char* Ap, const float* inv_scale, const unsigned char* zero_row, const ConvW& cw, int B, int Hin, int Win,
                                 int Hout, int Wout, int stride, float* slab, const float* bias, float* out, hipStream_t st) {
    const long M = (long)B * Hout * Wout;
    H3Args g{};
    const int tpb = cw.taps / DS_SPLIT;           // taps per batch
    g.seg[0] = h3_seg(Ap, inv_scale, 4L * cw.cinp, cw.hp, cw.hs, 4L * cw.taps * cw.cinp, tpb * cw.cinp);
    g.seg[0].sa_mul = 0;
    g.seg[0].strideB = 4L * tpb * cw.cinp;
    g.nseg = 1; g.M = (int)M; g.N = cw.Npad;
    g.cv_Hin = Hin; g.cv_Win = Win; g.cv_Hout = Hout; g.cv_Wout = Wout; g.cv_stride = stride; g.cv_ntaps = tpb; g.cv_cin = cw.cinp;
    g.cv_taps_total = cw.taps;
    g.zero_row = zero_row;
    if (launch_gemm_h3x<false, false, false, false, EpiSlabZ, 0, true>(g, DS_SPLIT, EpiSlabZ{slab, (long)cw.Npad, M * cw.Npad}, st) != hipSuccess)
        return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    const long MN = M * cw.Npad;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, st, slab, DS_SPLIT, MN, cw.Npad, bias, out);
    LAUNCH_CHECK();
    return TDX_OK;
}

}  // namespace

struct tdx_eres2net {
    int device = 0;
    float* dev; std::vector<BlockW> blocks; size_t stem_w, stem_b, seg_w, seg_b; ConvW ds; AffW fuse34;
    unsigned char* dev_planes = nullptr;      // x3 planes of the stage-3/4 and layer3_ds convolution weights
    size_t consts = 0;                        // dev + consts: {2^-10, 2^-9} inverse static scales, then 8 KB of zeros (halo row)
};

namespace {

// AFF(x, y): GEMM1 over the two K segments [x | y], GEMM2 with the gate epilogue
int run_aff(const tdx_eres2net* h, const AffW& a, const float* x, long ldx, const float* y, long ldy, float* tbuf, float* out, long ldo,
            long M, hipStream_t st) {
    if (a.C == 96 || a.C == 192) {      // the chain AFFs of stages 3-4: one fused launch
        AffArgs fa{x, ldx, y, ldy, h->dev + a.c0.w, a.c0.cinp, h->dev + a.c0.b, h->dev + a.c3.w, a.c3.cinp, h->dev + a.c3.b, out, ldo, M};
        return a.C == 96 ? launch_aff_fused<96>(fa, st) : launch_aff_fused<192>(fa, st);
    }
    {
        GemmSeg s0 = make_seg(x, ldx, h->dev + a.c0.w, 2L * a.C, a.C);
        GemmSeg s1 = make_seg(y, ldy, h->dev + a.c0.w + a.C, 2L * a.C, a.C);
        GemmArgs g = make_args((int)M, a.c0.Npad, s0);
        g.seg[1] = s1; g.nseg = 2;
        if (launch_gemm<false, false, false, false>(g, 1, EpiAffSilu{h->dev + a.c0.b, tbuf, a.ipad}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    GemmArgs g = make_args((int)M, a.c3.Npad, make_seg(tbuf, a.ipad, h->dev + a.c3.w, a.ipad, a.ipad));
    if (launch_gemm<false, false, false, false>(g, 1, EpiAffGate{h->dev + a.c3.b, x, ldx, y, ldy, out, ldo, a.C}, st) != hipSuccess)
        return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

// fuse34 (C = 2048, M >= 8192 rows) on the x3 core: GEMM1 as TWO K segments — x (a ReLU20 output) as planes with the static scale 2^10,
// y (layer3_ds: no activation) as planes with exact row scales — then t as planes with row scales for GEMM2 + the gate epilogue.
// Against the fp32-MFMA core (K = 4096: 105 TFLOP/s): 3.0 + 1.5 ms -> ~2.2 ms incl. the three split passes at B = 60.
int run_aff34_x3(const tdx_eres2net* h, const AffW& a, const float* x, const float* y, float* tbuf, float* out, long M,
                 unsigned char* xP, unsigned char* yP, float* ys, unsigned char* tP, float* ts, const float* inv20, hipStream_t st) {
    const int C = a.C, IP = a.ipad;
    if (tdx::launch_h3_split_rows_static(x, C, xP, M, C, 1024.0f, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    if (tdx::launch_h3_split_rows(y, C, yP, ys, M, C, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    {
        H3Args g{};
        g.seg[0] = h3_seg(xP, inv20, 4L * C, a.c0.hp, a.c0.hs, 4L * 2 * C, C);
        g.seg[0].sa_mul = 0;
        g.seg[1] = h3_seg(yP, ys, 4L * C, a.c0.hp + 4L * C, a.c0.hs, 4L * 2 * C, C);
        g.nseg = 2; g.M = (int)M; g.N = a.c0.Npad;
        if (launch_gemm_h3x<false, false, false, true>(g, 1, EpiAffSilu{h->dev + a.c0.b, tbuf, IP}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    if (tdx::launch_h3_split_rows(tbuf, IP, tP, ts, M, IP, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    H3Args g{};
    g.seg[0] = h3_seg(tP, ts, 4L * IP, a.c3.hp, a.c3.hs, 4L * IP, IP);
    g.nseg = 1; g.M = (int)M; g.N = a.c3.Npad;
    if (launch_gemm_h3<false>(g, 1, EpiAffGate{h->dev + a.c3.b, x, (long)C, y, (long)C, out, (long)C, C}, st) != hipSuccess)
        return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

struct Dims { int H[5], W[5]; };
inline Dims make_dims(int F) {
    Dims d; d.H[0] = 80; d.W[0] = F;
    for (int s = 0; s < NSTAGE; ++s) {
        d.H[s + 1] = kStride[s] == 1 ? d.H[s] : (d.H[s] - 1) / 2 + 1;
        d.W[s + 1] = kStride[s] == 1 ? d.W[s] : (d.W[s] - 1) / 2 + 1;
    }
    return d;
}

}  // namespace

extern "C" {

int tdx_eres2net_create(const void* blob, size_t blob_bytes, int device, tdx_eres2net** out) {
    if (!blob || !out) return tdx::fail(TDX_E_INVALID, "tdx_eres2net_create: null argument");
    tdx::Blob bl;
    if (!bl.parse(blob, blob_bytes)) return tdx::fail(TDX_E_BLOB, "tdx_eres2net_create: malformed TDXW blob");
    std::vector<float> host;
    bool ok = true; std::string missing;
    auto get = [&](const std::string& name, size_t n) -> const float* {
        const tdx::BlobTensor* t = bl.find(name);
        if (!t || t->numel != n) { ok = false; if (missing.empty()) missing = name; return nullptr; }
        return t->data;
    };
    // conv [N,Cin,kh,kw] (+bias) followed by eval BatchNorm `bn` ("" = none) -> [Npad][taps][cinp], bias[Npad]
    auto fold = [&](const std::string& wname, const std::string& bname, const std::string& bn, int N, int cin, int taps) -> ConvW {
        ConvW cw; cw.N = N; cw.Npad = up(N, 128); cw.cin = cin; cw.cinp = up(cin, 32); cw.taps = taps;
        const float* W = get(wname, (size_t)N * cin * taps);
        const float* cb = bname.empty() ? nullptr : get(bname, N);
        const float *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
        if (!bn.empty()) { g = get(bn + "weight", N); be = get(bn + "bias", N); mu = get(bn + "running_mean", N); var = get(bn + "running_var", N); }
        cw.w = host.size(); host.resize(host.size() + al((size_t)cw.Npad * taps * cw.cinp), 0.f);
        cw.b = host.size(); host.resize(host.size() + al(cw.Npad), 0.f);
        if (!ok) return cw;
        for (int n = 0; n < N; ++n) {
            const double sc = bn.empty() ? 1.0 : (double)g[n] / sqrt((double)var[n] + 1e-5);
            const double b0 = cb ? (double)cb[n] : 0.0;
            host[cw.b + n] = (float)(bn.empty() ? b0 : (b0 - (double)mu[n]) * sc + (double)be[n]);
            for (int c = 0; c < cin; ++c)
                for (int t = 0; t < taps; ++t)
                    host[cw.w + ((size_t)n * taps + t) * cw.cinp + c] = (float)((double)W[((size_t)n * cin + c) * taps + t] * sc);
        }
        return cw;
    };
    auto fold_aff = [&](const std::string& p, int C) -> AffW {
        AffW a; a.C = C; a.inter = C / 4; a.ipad = up(a.inter, 32);
        a.c0 = fold(p + "local_att.0.weight", p + "local_att.0.bias", p + "local_att.1.", a.inter, 2 * C, 1);
        a.c3 = fold(p + "local_att.3.weight", p + "local_att.3.bias", p + "local_att.4.", C, a.inter, 1);
        return a;
    };
    tdx_eres2net* h = new tdx_eres2net();
    // stem: [64,1,3,3] + bn1 -> w9[9][64], bias[64]
    {
        const float* W = get("conv1.weight", 64 * 9);
        const float *g = get("bn1.weight", 64), *be = get("bn1.bias", 64), *mu = get("bn1.running_mean", 64), *var = get("bn1.running_var", 64);
        h->stem_w = host.size(); host.resize(host.size() + al(9 * 64), 0.f);
        h->stem_b = host.size(); host.resize(host.size() + al(64), 0.f);
        if (ok) for (int n = 0; n < 64; ++n) {
            const double sc = (double)g[n] / sqrt((double)var[n] + 1e-5);
            host[h->stem_b + n] = (float)((double)be[n] - (double)mu[n] * sc);
            for (int t = 0; t < 9; ++t) host[h->stem_w + t * 64 + n] = (float)((double)W[n * 9 + t] * sc);
        }
    }
    int in_planes = 64;
    for (int s = 0; s < NSTAGE && ok; ++s) {
        const int planes = kPlanes[s], width = planes * 24 / 64, cout = planes * 4;
        for (int i = 0; i < kBlocks[s] && ok; ++i) {
            const std::string p = "layer" + std::to_string(s + 1) + "." + std::to_string(i) + ".";
            BlockW b;
            b.stride = i == 0 ? kStride[s] : 1; b.cin = in_planes; b.width = width; b.wpad = up(width, 32); b.w4 = width * SCALE;
            b.cout = cout; b.is_aff = s >= 2;
            b.conv1 = fold(p + "conv1.weight", "", p + "bn1.", b.w4, in_planes, 1);
            for (int j = 0; j < SCALE; ++j) b.convs[j] = fold(p + "convs." + std::to_string(j) + ".weight", "", p + "bns." + std::to_string(j) + ".", width, width, 9);
            b.conv3 = fold(p + "conv3.weight", "", p + "bn3.", cout, b.w4, 1);
            b.has_sc = (b.stride != 1) || (in_planes != cout);
            if (b.has_sc) b.sc = fold(p + "shortcut.0.weight", "", p + "shortcut.1.", cout, in_planes, 1);
            if (b.is_aff) for (int j = 0; j < SCALE - 1; ++j) b.aff[j] = fold_aff(p + "fuse_models." + std::to_string(j) + ".", width);
            h->blocks.push_back(b);
            in_planes = cout;
        }
    }
    if (ok) {
        h->ds = fold("layer3_ds.weight", "", "", 2048, 1024, 9);
        h->fuse34 = fold_aff("fuse34.", 2048);
        h->seg_w = host.size(); host.resize(host.size() + al((size_t)EMB * 40960), 0.f);
        const float* sw = get("seg_1.weight", (size_t)EMB * 40960);
        if (sw) memcpy(host.data() + h->seg_w, sw, (size_t)EMB * 40960 * sizeof(float));
        h->seg_b = host.size(); host.resize(host.size() + al(EMB), 0.f);
        const float* sb = get("seg_1.bias", EMB);
        if (sb) memcpy(host.data() + h->seg_b, sb, EMB * sizeof(float));
    }
    h->consts = host.size();
    host.resize(host.size() + al(64 + 2048), 0.f);
    host[h->consts] = 1.0f / 1024.0f;      // activations <= 20 (ReLU20): x * 2^10 < 2^15
    host[h->consts + 1] = 1.0f / 512.0f;   // sums of two such (Res2Net chain inputs, AFF outputs) <= 40: x * 2^9 < 2^15
    if (!ok) { delete h; return tdx::fail(TDX_E_BLOB, "tdx_eres2net_create: tensor missing or wrong size: " + missing); }
    tdx::DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    h->device = device;
    e = hipMalloc(&h->dev, host.size() * sizeof(float));
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    {   // x3 planes of the compute-heavy convolutions (stages 3-4: K >= 96 per tap; layer3_ds), split once
        std::vector<ConvW*> jobs;
        size_t bi = 0;
        for (int s = 0; s < NSTAGE; ++s)
            for (int i = 0; i < kBlocks[s]; ++i, ++bi) {
                BlockW& b = h->blocks[bi];
                if (s < 1) { for (int j = 0; j < SCALE; ++j) jobs.push_back(&b.convs[j]); continue; }     // stage 1: the chain convolutions only
                jobs.push_back(&b.conv1); if (b.has_sc) jobs.push_back(&b.sc);
                for (int j = 0; j < SCALE; ++j) jobs.push_back(&b.convs[j]);
                jobs.push_back(&b.conv3);
            }
        jobs.push_back(&h->ds);
        jobs.push_back(&h->fuse34.c0); jobs.push_back(&h->fuse34.c3);
        size_t bytes = 0;
        for (ConvW* c : jobs) bytes += (size_t)c->Npad * c->taps * c->cinp * 4 + al(c->Npad) * 4;
        e = hipMalloc(&h->dev_planes, bytes);
        if (e != hipSuccess) { hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
        unsigned char* q = h->dev_planes;
        for (ConvW* c : jobs) {
            const int K = c->taps * c->cinp;
            float* sc = (float*)(q + (size_t)c->Npad * K * 4);
            e = tdx::launch_h3_split_rows_long(h->dev + c->w, K, q, sc, c->Npad, K, nullptr);
            if (e != hipSuccess) { hipFree(h->dev_planes); hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
            c->hp = q; c->hs = sc;
            q += (size_t)c->Npad * K * 4 + al(c->Npad) * 4;
        }
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { hipFree(h->dev_planes); hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    }
    *out = h;
    return TDX_OK;
}

int tdx_eres2net_destroy(tdx_eres2net* h) {
    if (h) { if (h->dev_planes) hipFree(h->dev_planes); if (h->dev) hipFree(h->dev); delete h; }
    return TDX_OK;
}

namespace {
struct WsPlan { size_t P, keep3, o1, spin, tbuf, stats, hx, hcat, hin, slab, total; };
inline WsPlan ws_plan(const tdx_eres2net* h, int B, int F) {
    const Dims d = make_dims(F);
    WsPlan w{}; size_t mP = (size_t)B * 80 * F * 64, mo1 = 0, msp = 0, mt = 0, mhx = 0, mhc = 0, mhi = 0;
    int s = 0, cnt = 0;
    for (const BlockW& b : h->blocks) {
        const size_t rows = (size_t)B * d.H[s + 1] * d.W[s + 1];
        if (s >= 1) {      // planes of the x3 convolutions' inputs: block input, concatenated chain outputs, chain-conv input
            const size_t rows_in = cnt == 0 ? (size_t)B * d.H[s] * d.W[s] : rows;
            mhx = std::max(mhx, rows_in * b.cin); mhc = std::max(mhc, rows * b.w4); mhi = std::max(mhi, rows * b.wpad);
        }
        mP = std::max(mP, rows * b.cout); mo1 = std::max(mo1, rows * b.w4); msp = std::max(msp, rows * b.wpad);
        if (b.is_aff) mt = std::max(mt, rows * (size_t)b.aff[0].ipad);
        if (++cnt == kBlocks[s]) { cnt = 0; ++s; }
    }
    const size_t rows3 = (size_t)B * d.H[3] * d.W[3], rows4 = (size_t)B * d.H[4] * d.W[4];
    mt = std::max(mt, rows4 * 512);
    w.P = al(mP + 4096); w.keep3 = al(rows3 * 1024 + 4096); w.o1 = al(mo1 + 4096); w.spin = al(msp + 4096); w.tbuf = al(mt + 4096);
    w.stats = al((size_t)B * 40960);
    mhx = std::max(mhx, rows3 * 1024);
    w.hx = al(mhx + 4096); w.hcat = al(mhc + 4096); w.hin = al(mhi + 4096);
    w.slab = rows4 <= (size_t)DS_SPLIT_ROWS ? al(DS_SPLIT * rows4 * 2048 + 4096) : 0;      // tap-split partial sums of layer3_ds (few rows only)
    w.total = 3 * w.P + w.keep3 + 2 * w.o1 + 2 * w.spin + w.tbuf + w.stats + w.hx + w.hcat + w.hin + w.slab;
    return w;
}
}  // namespace

size_t tdx_eres2net_workspace_bytes(const tdx_eres2net* h, int B, int F) {
    if (!h || B < 1 || F < 9) return 0;
    return ws_plan(h, B, F).total * sizeof(float);
}

double tdx_eres2net_flops(const tdx_eres2net* h, int B, int F) {
    if (!h || F < 1) return 0.0;
    const Dims d = make_dims(F);
    double fl = 2.0 * 80 * F * 9 * 64;
    int s = 0, cnt = 0;
    for (const BlockW& b : h->blocks) {
        const double pos = (double)d.H[s + 1] * d.W[s + 1];
        fl += 2.0 * pos * ((double)b.cin * b.w4 + 4.0 * 9 * b.width * b.width + (double)b.w4 * b.cout + (b.has_sc ? (double)b.cin * b.cout : 0.0));
        if (b.is_aff) fl += 3.0 * 2.0 * pos * (2.0 * b.width * (b.width / 4) + (double)(b.width / 4) * b.width);
        if (++cnt == kBlocks[s]) { cnt = 0; ++s; }
    }
    const double pos4 = (double)d.H[4] * d.W[4];
    fl += 2.0 * pos4 * (9.0 * 1024 * 2048 + 4096.0 * 512 + 512.0 * 2048) + 2.0 * 40960 * EMB;
    return fl * B;
}

int tdx_eres2net_forward(tdx_eres2net* h, const float* feat, int B, int F, float* emb, void* ws_, size_t ws_bytes, void* stream) {
    if (!h || !feat || !emb || !ws_ || B < 1) return tdx::fail(TDX_E_INVALID, "tdx_eres2net_forward: bad argument");
    if (F < 9) return tdx::fail(TDX_E_INVALID, "tdx_eres2net_forward: need >= 9 fbank frames (TSTP needs >= 2 pooled frames)");
    const WsPlan wp = ws_plan(h, B, F);
    if (ws_bytes < wp.total * sizeof(float)) return tdx::fail(TDX_E_WORKSPACE, "tdx_eres2net_forward: workspace too small");
    tdx::DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return tdx::fail_hip(guard.err, __FILE__, __LINE__);
    hipStream_t st = (hipStream_t)stream;
    const Dims d = make_dims(F);
    const size_t rows1 = (size_t)B * 80 * F;
    float* ws = (float*)ws_;
    float* P[3] = {ws, ws + wp.P, ws + 2 * wp.P};
    float* keep3 = ws + 3 * wp.P;
    float* o1 = keep3 + wp.keep3; float* cat = o1 + wp.o1;
    float* spin[2] = {cat + wp.o1, cat + wp.o1 + wp.spin};
    float* tbuf = spin[1] + wp.spin; float* stats = tbuf + wp.tbuf;
    unsigned char* hx = (unsigned char*)(stats + wp.stats);
    unsigned char* hcat = hx + wp.hx * sizeof(float);
    unsigned char* hin = hcat + wp.hcat * sizeof(float);
    float* slab = reinterpret_cast<float*>(hin + wp.hin * sizeof(float));
    const float* inv20 = h->dev + h->consts;            // 2^-10
    const float* inv40 = h->dev + h->consts + 1;        // 2^-9
    const unsigned char* zero_row = (const unsigned char*)(h->dev + h->consts + 64);

    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((rows1 * 16 + 255) / 256)), dim3(256), 0, st, feat, h->dev + h->stem_w, h->dev + h->stem_b, P[0], F, (long)rows1);
    LAUNCH_CHECK();
    const float* x = P[0];
    int cur = 0;                 // index of the P buffer holding x, or -1 when x is keep3
    int s = 0, cnt = 0;
    for (const BlockW& b : h->blocks) {
        const int Hin = (cnt == 0) ? d.H[s] : d.H[s + 1], Win = (cnt == 0) ? d.W[s] : d.W[s + 1];
        const int Ho = d.H[s + 1], Wo = d.W[s + 1];
        const long M = (long)B * Ho * Wo;
        const bool last_of_stage3 = (s == 2) && (cnt == kBlocks[2] - 1);
        const int iy = cur < 0 ? 0 : (cur + 1) % 3, ir = cur < 0 ? 1 : (cur + 2) % 3;
        float* y = last_of_stage3 ? keep3 : P[iy];
        float* res = P[ir];
        const bool x3 = b.conv1.hp != nullptr;          // stages 3-4: the convolutions run on the split-f16 x3 core
        const float* resid = x;
        if (x3) {
            // block input (ReLU20 output, <= 20) as planes with the static scale 2^10, shared by conv1 and the shortcut
            int Hc = Hin, Wc = Win, sc_ = b.stride;            // geometry of the 1x1 convolutions over the planes
            if (b.stride == 2 && b.cin % 16 == 0) {            // only the pixels the stride-2 convolutions read, compacted
                const long R = (long)B * Ho * Wo;
                hipLaunchKernelGGL(split_rows_static_s2_kernel, dim3((unsigned)((R * (b.cin / 8) + 255) / 256)), dim3(256), 0, st,
                                   x, b.cin, hx, R, b.cin, Hin, Win, Ho, Wo, 1024.0f);
                LAUNCH_CHECK();
                Hc = Ho; Wc = Wo; sc_ = 1;
            } else if (tdx::launch_h3_split_rows_static(x, b.cin, hx, (long)B * Hin * Win, b.cin, 1024.0f, st) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
            TRY(conv_gemm_h3(hx, inv20, zero_row, b.conv1, B, Hc, Wc, Ho, Wo, sc_, EpiRelu20{h->dev + b.conv1.b, o1, b.w4, b.w4}, st));
            if (b.has_sc) {
                TRY(conv_gemm_h3(hx, inv20, zero_row, b.sc, B, Hc, Wc, Ho, Wo, sc_, EpiBiasG{h->dev + b.sc.b, res, b.cout, b.cout}, st));
                resid = res;
            }
        } else {
            // conv1 (1x1, stride) + bn1 + relu20 -> o1 [M, w4]
            TRY(conv_gemm(x, b.cin, h->dev, b.conv1, B, Hin, Win, Ho, Wo, b.stride, EpiRelu20{h->dev + b.conv1.b, o1, b.w4, b.w4}, st));
            if (b.has_sc) {
                TRY(conv_gemm(x, b.cin, h->dev, b.sc, B, Hin, Win, Ho, Wo, b.stride, EpiBiasG{h->dev + b.sc.b, res, b.cout, b.cout}, st));
                resid = res;
            }
        }
        // Res2Net chain: conv_j reads in_j (o1 chunk 0 in place, later spin[j&1]) and its epilogue
        // writes sp_j into cat and, for the plain-add stages, in_{j+1} = sp_j + spx[j+1] into spin[(j+1)&1]
        for (int j = 0; j < SCALE; ++j) {
            const float* in = j == 0 ? o1 : spin[j & 1];
            const long ldin = j == 0 ? b.w4 : b.wpad;
            const bool plain_next = (j + 1 < SCALE) && !b.is_aff;
            EpiChain e{h->dev + b.convs[j].b, cat, b.w4, j * b.width, b.width, b.wpad, o1, b.w4, (j + 1) * b.width,
                       plain_next ? spin[(j + 1) & 1] : nullptr};
            if (b.convs[j].hp && !b.is_aff && (b.width == 24 || b.width == 48)) {      // stages 1-2: LDS-tiled direct convolution (x3), fp32 input split while staged
                ChainX3Args ca{in, ldin, b.convs[j].hp, b.convs[j].hs, 9 * b.convs[j].cinp, b.convs[j].cinp, h->dev + b.convs[j].b, cat, b.w4, j * b.width,
                               o1, b.w4, (j + 1) * b.width, plain_next ? spin[(j + 1) & 1] : nullptr, b.wpad, Ho, Wo};
                if (b.width == 24) TRY((launch_chain_x3<24, 4>(ca, B, st)));
                else TRY((launch_chain_x3<48, 4>(ca, B, st)));
            } else if (x3) {     // chain input = sum of two ReLU20 outputs or an AFF blend of them (<= 40): static scale 2^9
                if (tdx::launch_h3_split_rows_static(in, ldin, hin, M, b.wpad, 512.0f, st) != hipSuccess)
                    return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
                TRY(conv_gemm_h3(hin, inv40, zero_row, b.convs[j], B, Ho, Wo, Ho, Wo, 1, e, st));
            } else {
                TRY(conv_gemm(in, ldin, h->dev, b.convs[j], B, Ho, Wo, Ho, Wo, 1, e, st));
            }
            if (b.is_aff && j + 1 < SCALE)      // in_{j+1} = AFF(sp_j, spx[j+1])
                TRY(run_aff(h, b.aff[j], cat + j * b.width, b.w4, o1 + (j + 1) * b.width, b.w4, tbuf, spin[(j + 1) & 1], b.wpad, M, st));
        }
        if (x3) {   // conv3 + bn3 + residual + relu20 -> y   (cat = ReLU20 outputs)
            if (tdx::launch_h3_split_rows_static(cat, b.w4, hcat, M, b.w4, 1024.0f, st) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
            TRY(conv_gemm_h3(hcat, inv20, zero_row, b.conv3, B, Ho, Wo, Ho, Wo, 1, EpiConv3{h->dev + b.conv3.b, resid, y, b.cout}, st));
        } else {
            GemmArgs g = make_args((int)M, b.conv3.Npad, make_seg(cat, b.w4, h->dev + b.conv3.w, b.conv3.cinp, b.conv3.cinp));
            if (launch_gemm<false, false, false, false>(g, 1, EpiConv3{h->dev + b.conv3.b, resid, y, b.cout}, st) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        }
        x = y;
        cur = last_of_stage3 ? -1 : iy;
        if (++cnt == kBlocks[s]) { cnt = 0; ++s; }
    }
    // layer3_ds (3x3 stride 2) on out3, fuse34 = AFF(out4, out3_ds), TSTP, seg_1
    const int i_ds = (cur + 1) % 3, i_fu = (cur + 2) % 3;
    const long M4 = (long)B * d.H[4] * d.W[4];
    if (tdx::launch_h3_split_rows_static(keep3, 1024, hx, (long)B * d.H[3] * d.W[3], 1024, 1024.0f, st) != hipSuccess)
        return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    if (wp.slab && h->ds.taps == 9 && h->ds.Npad == 2048)
        TRY(conv_gemm_h3_tapsplit(hx, inv20, zero_row, h->ds, B, d.H[3], d.W[3], d.H[4], d.W[4], 2, slab, nullptr, P[i_ds], st));
    else
        TRY(conv_gemm_h3(hx, inv20, zero_row, h->ds, B, d.H[3], d.W[3], d.H[4], d.W[4], 2, EpiBiasG{nullptr, P[i_ds], 2048, 2048}, st));
    const char* aff34_env = getenv("TDX_ERES_AFF34_ROWS");             // (tests force the x3 path at small sizes)
    const long aff34_min = aff34_env ? atol(aff34_env) : 8192;
    if (M4 >= aff34_min && 2 * M4 <= (long)B * 40960 && h->fuse34.c0.hp && h->fuse34.ipad == 512)      // (fewer rows: a handful of 256-row tiles with 256 k steps each — the fp32 core's smaller tiles win)
        TRY(run_aff34_x3(h, h->fuse34, x, P[i_ds], tbuf, P[i_fu], M4, hx, hcat, stats, hin, stats + M4, inv20, st));
    else
        TRY(run_aff(h, h->fuse34, x, 2048, P[i_ds], 2048, tbuf, P[i_fu], 2048, M4, st));
    hipLaunchKernelGGL(tstp_kernel, dim3(d.H[4], B), dim3(256), 0, st, P[i_fu], stats, d.W[4], 2048, d.H[4]);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_kernel, dim3(EMB, B), dim3(256), 0, st, stats, h->dev + h->seg_w, h->dev + h->seg_b, emb, 40960);
    LAUNCH_CHECK();
    return TDX_OK;
}

}  // extern "C"
