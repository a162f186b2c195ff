// mf2.hip — MossFormer2 separator on MI355X: weight import, workspace plan, launch sequence
// and the C-ABI entry points declared in include/tdx.h.
//
// Reference being replaced: MossFormer2.forward look2hear/models/mossformer2.py:563-589 and
// everything it calls.  Data layout: every activation is
// token-major fp32 [B*S, C]; the 24-layer stack runs as a fixed sequence of launches on
// the caller's stream (no allocation, no sync => hipGraph-capturable).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tdx.h"
#include "gemm.hpp"
#include "gemm_x6.hpp"
#include "gemm_h3.hpp"
#include "gemm_h3a.hpp"
#include "mf2_kernels.hpp"
#include "tdx_common.hpp"

using namespace tdx;

namespace {

constexpr int C = 512, HID = 2048, QK = 128, HQ = HID + QK, INNER = 256;

// ------------------------------------------------------------------ GEMM epilogues
// (concept in gemm.hpp: col()/row() fetch per-column / per-row constants once, store() writes)
struct Col2 { float a, b; };
template <bool ACT>
struct EpiHiddenT {  // silu(acc*rs[m]*g[n] + b[n])                      mossformer_block.py:89-102
    // ACT = false: the pre-activation is stored and the consuming depthwise convolution applies SiLU as it loads (Conv17Args::silu_in)
    const float* rs; const float* g; const float* b; float* out; long ld;
    __device__ static float f(float x) { return ACT ? siluf_acc(x) : x; }
    __device__ Col2 col(int, int n) const { return Col2{g[n], b[n]}; }
    __device__ float row(int, int m) const { return rs[m]; }
    __device__ void store(int, int m, int n, float v, float r, Col2 c) const { out[(long)m * (int)ld + n] = f(v * r * c.a + c.b); }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, float r, Col2 c) const { *p = f(v * r * c.a + c.b); }
    __device__ float rowmul(float r) const { return r; }              // (folded into the kernel's row / column scales:
    __device__ float colmul(Col2 c) const { return c.a; }             //  put_scaled() gets v * r * c.a)
    __device__ void put_scaled(float* p, float v, float, Col2 c) const { *p = f(v + c.b); }
};
using EpiHidden = EpiHiddenT<true>;
// the same (pre-activation stored) with the ScaleNorm row factor formed from per-segment sums of squares that the producer of
// the planes left behind: rs = 1 / max(sqrt(sum_j ss[j]) * C^-1/2, 1e-5)   (mossformer_block.py:52-54).
//   SHIFT: the token-shifted row = [half 0 of token s-1 (nothing at s = 0) | half 1 of token s]
//   else : the NSEG 128-channel segments of the gated attention output
template <int NSEG, bool SHIFT>
struct EpiHiddenSN {
    const float* ss; long M; int S; float cinv; const float* g; const float* b; float* out; long ld;
    __device__ Col2 col(int, int n) const { return Col2{g[n], b[n]}; }
    __device__ float row(int, int m) const {
        float s = 0.f;
        if constexpr (SHIFT) { s = ss[M + m]; if (m % S) s += ss[m - 1]; }
        else {
#pragma unroll
            for (int j = 0; j < NSEG; ++j) s += ss[(long)j * M + m];
        }
        return 1.0f / fmaxf(sqrtf(s) * cinv, 1e-5f);
    }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, float r, Col2 c) const { *p = v * r * c.a + c.b; }
    __device__ float rowmul(float r) const { return r; }
    __device__ float colmul(Col2 c) const { return c.a; }
    __device__ void put_scaled(float* p, float v, float, Col2 c) const { *p = v + c.b; }
    __device__ void store(int, int m, int n, float v, float r, Col2 c) const { out[(long)m * (int)ld + n] = v * r * c.a + c.b; }
};
// to_hidden + to_qk with the depthwise convolution of v|u FUSED into the epilogue (gemm_h3.hpp H3Conv): the v|u columns leave the
// launch as K-major planes (+ u in fp32) — y = silu(.) never goes through HBM and conv17<4> is gone; the to_qk columns leave as
// fp32 pre-activations for conv17<3>, like EpiHiddenSN<2, true>.                   mossformer_block.py:89-102, conv_module.py:180-220
struct EpiHiddenConv {
    const float* ss; long M; int S; float cinv; const float* g; const float* b; float* out; long ld;
    const float* cw; unsigned char* vuP; float sv; float* u32; int Sp;
    __device__ Col2 col(int, int n) const { return Col2{g[n], b[n]}; }
    __device__ float row(int, int m) const {
        float s = ss[M + m]; if (m % S) s += ss[m - 1];
        return 1.0f / fmaxf(sqrtf(s) * cinv, 1e-5f);
    }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ float rowmul(float r) const { return r; }
    __device__ float colmul(Col2 c) const { return c.a; }
    __device__ void put_scaled(float* p, float v, float, Col2 c) const { *p = v + c.b; }
    __device__ void put(float* p, float v, float r, Col2 c) const { *p = v * r * c.a + c.b; }
    __device__ void store(int, int m, int n, float v, float r, Col2 c) const { out[(long)m * (int)ld + n] = v * r * c.a + c.b; }
    __device__ float act(float v, Col2 c) const { return siluf_acc(v + c.b); }
    __device__ tdx::H3Conv conv() const { return tdx::H3Conv{cw, HID, HID, vuP, 4L * HID, sv, u32, HID / 2, (long)(HID / 2), S, Sp}; }
};
// pad rows S .. Sp-1 of every sample of the K-major v|u planes = 0 (the split-K lin_k^T [v|u] launch and the attention launch read
// whole 256-token groups); once per forward when the planes come from the fused epilogue, which only writes the S real rows
__global__ void zero_pad_rows_kernel(unsigned char* __restrict__ planes, int S, int Sp, long pitch) {
    const int b = blockIdx.y, r = S + blockIdx.x;
    uint4* p = reinterpret_cast<uint4*>(planes + ((long)b * Sp + r) * pitch);
    for (int i = threadIdx.x; i < pitch / 16; i += blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
struct EpiQuadSim {  // relu(acc/256)^2 with key mask                       mossformer_block.py:256-262
    float* A; int G; int S; float inv_g;
    __device__ bool col(int z, int n) const { return (z % G) * 256 + n < S; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int z, int m, int n, float v, EpiNone, bool keep) const {
        float s = fmaxf(v * inv_g, 0.f);
        A[((long)z * 256 + m) * 256 + n] = keep ? s * s : 0.f;
    }
};
struct EpiQuadSimPl {  // the same, written as row-major planes + row scales (PLOUT): the A operand of the attention GEMM's quadratic segment
    unsigned char* P; float* sc; int G; int S; float inv_g;
    __device__ bool col(int z, int n) const { return (z % G) * 256 + n < S; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float val(int, int, int, float v, EpiNone, bool keep) const { const float s = fmaxf(v * inv_g, 0.f); return keep ? s * s : 0.f; }
    __device__ tdx::H3PlOut plout(int) const { return tdx::H3PlOut{P, 1024, sc, nullptr, 0}; }
    __device__ long prow(int z, int m) const { return (long)z * 256 + m; }
};
struct EpiStore {    // plain store, per-batch stride
    float* out; long ld; long strideZ;
    __device__ EpiNone col(int, int) const { return EpiNone{}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int z, int m, int n, float v, EpiNone, EpiNone) const { out[(long)z * strideZ + (long)m * (int)ld + n] = v; }
};
struct EpiAttnGate { // o = (att_u*v)*sigmoid(att_v*u)                       mossformer_block.py:217
    const float* vu; float* o; float* att_v; float* att_u; int G; int S; int E;
    __device__ EpiNone col(int, int) const { return EpiNone{}; }
    __device__ long row(int z, int m) const {     // global token row, or -1 for group padding
        const int b = z / G, s = (z % G) * 256 + m;
        return s < S ? (long)b * S + s : -1L;
    }
    __device__ bool full(int z, int m0) const { return (z % G) * 256 + m0 + 256 <= S; }
    __device__ float2 aux(int, int, int c, long rw) const {       // the gate's v, u operands
        if (rw < 0 || att_v) return make_float2(0.f, 0.f);
        return make_float2(vu[rw * (2 * E) + c], vu[rw * (2 * E) + E + c]);
    }
    __device__ void store2(int, int, int c, float av, float au, long rw, EpiNone, float2 vu2) const {
        if (rw < 0) return;
        if (att_v) {  // stand-alone cal_attention: return the two attention outputs
            att_v[rw * E + c] = av;
            att_u[rw * E + c] = au;
        } else {
            o[rw * E + c] = (au * vu2.x) * sigmoidf_acc(av * vu2.y);
        }
    }
};
struct EpiAttnGatePlOut { // the gate o = (att_u*v)*sigmoid(att_v*u) with o written as ROW-major planes (PLOUT): one scale and one sum of
    // squares per (token, 128-channel segment) — the A operand of to_out with segmented row scales and its ScaleNorm statistics.
    // Gate operands:
    //   v  from the K-major split-f16 planes ((hi + lo) * inv; hi and lo of 32 columns share one 128-B line: lanes 2j / 2j+1
    //      (adjacent columns) share the loads — the even lane fetches the 4-byte (c, c+1) word of hi, the odd lane that of lo — and
    //      swap by DPP quad_perm [1,0,3,2] in val2_scaled()).  v enters the product linearly: the planes' ABSOLUTE precision
    //      (2^-40 of the layer's static bound) is an error 2^-40 * bound * |att_u| against |att_u| * |v|_typical — harmless.
    //   u  in fp32 (`u32`, written by conv17<4> next to the planes).  u sits INSIDE the sigmoid, multiplied by att_v: an absolute
    //      error du becomes an error att_v * du of the argument.  With large attention values (heavy-tailed checkpoints: outlier
    //      rows in to_hidden AND to_qk give att_v ~ 1e8) the planes' 2^-40 * bound flipped the gate of the 0.3 % of the elements
    //      whose argument is O(1): 6e-4 rel-L2 on the layer output where fp32 arithmetic has 2e-6
    //      (tests/test_gpu_mossformer2.py::test_static_scales_under_heavy_tailed_weights).  fp32's relative precision is what the
    //      reference has; the bytes read per output element are the same (one 4-byte word of v, one of u).
    // aux() returns the RAW words (the kernel issues it ahead of the stores: anything that consumed the loads there would wait for
    // them there); the lane swap and the unpacking happen in val2_scaled().
    const unsigned char* vuP; const float* inv; const float* u32; unsigned char* oP; float* os; float* oss; long M; int G; int S; int Sp; int E;
    __device__ float col(int, int) const { return inv[0]; }
    __device__ long row(int z, int m) const { const int b = z / G, s = (z % G) * 256 + m; return s < S ? (long)b * S + s : -1L; }
    __device__ bool full(int z, int m0) const { return (z % G) * 256 + m0 + 256 <= S; }
    __device__ int2 aux(int z, int m, int c, long) const {
        const int b = z / G, s = min((z % G) * 256 + m, S - 1);
        const int c2 = c & ~1;
        const unsigned char* p = vuP + (long)b * Sp * (8L * E) + (long)s * (8 * E) + (c2 >> 5) * 128 + (c2 & 31) * 2 + (c & 1) * 64;
        return make_int2(*reinterpret_cast<const int*>(p), __float_as_int(u32[((long)b * S + s) * E + c]));
    }
    // the plane scale k folded into the kernel's column scale of att_u, the sigmoid's -log2(e) into that of att_v:
    // val2_scaled() gets av' = -log2(e) * att_v and au' = k * att_u
    __device__ float2 pairmul(float k) const { return make_float2(-1.4426950408889634f, k); }
    __device__ float val2_scaled(int, int, int c, float av, float au, long rw, float, int2 w) const {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        // even lane: own = (hi[c], hi[c+1]), partner = (lo[c], lo[c+1]); odd lane: own = (lo[c-1], lo[c]), partner = (hi[c-1], hi[c]).
        // One byte permute builds (hi[c], lo[c]) as a packed f16 pair.
        const unsigned ov = (unsigned)__builtin_amdgcn_mov_dpp(w.x, 0xB1, 0xF, 0xF, true);
        const unsigned sel = (c & 1) ? 0x03020706u : 0x05040100u;        // v_perm_b32: bytes 0-3 = own word, 4-7 = partner word
        const h2 pv = __builtin_bit_cast(h2, __builtin_amdgcn_perm(ov, (unsigned)w.x, sel));
        const float vs = (float)pv[0] + (float)pv[1];                    // v / k
        const float t0 = au * vs;                                        // att_u * v
        const float t1 = av * __int_as_float(w.y);                       // -log2(e) * att_v * u
        const float o = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t1));
        return rw < 0 ? 0.f : o;
    }
    __device__ tdx::H3PlOut plout(int) const { return tdx::H3PlOut{oP, 4L * E, os, oss, M}; }
    __device__ long prow(int z, int m) const { return row(z, m); }
};
struct EpiBiasPrelu { // prelu_scalar(acc + b[n])                            mossformer_block.py:405-408
    const float* b; const float* a; float* out; long ld;
    __device__ Col2 col(int, int n) const { return Col2{b[n], a[0]}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, Col2 c) const {
        v += c.a;
        out[(long)m * (int)ld + n] = v >= 0.f ? v : c.b * v;
    }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, EpiNone, Col2 c) const { v += c.a; *p = v >= 0.f ? v : c.b * v; }
};
struct EpiBiasSilu { const float* b; float* out; long ld;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)m * (int)ld + n] = siluf_acc(v + c); }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, EpiNone, float c) const { *p = siluf_acc(v + c); } };
struct EpiBiasRelu { const float* b; float* out; long ld;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)m * (int)ld + n] = fmaxf(v + c, 0.f); }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, EpiNone, float c) const { *p = fmaxf(v + c, 0.f); } };
struct EpiBiasReluPl { const float* b; unsigned char* P; float* sc;   // relu(acc + b) as row-major planes (N = 256 = one tile: whole rows)
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float val(int, int, int, float v, EpiNone, float c) const { return fmaxf(v + c, 0.f); }
    __device__ tdx::H3PlOut plout(int) const { return tdx::H3PlOut{P, 1024, sc, nullptr, 0}; }
    __device__ long prow(int, int m) const { return m; } };
struct EpiBias { const float* b; float* out; long ld;   // b may be null
    __device__ float col(int, int n) const { return b ? b[n] : 0.f; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)m * (int)ld + n] = v + c; }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, EpiNone, float c) const { *p = v + c; } };
struct EpiBiasHalves { const float* b; float* out; long M;   // [M][2C] written as [2][M][C] (the two mask branches)
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)(n >> 9) * M * C + (long)m * C + (n & 511)] = v + c; } };
struct EpiBiasResidual { const float* b; float* x; long ld;   // x += acc + b   mossformer_block.py:424-425
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { return x[(long)m * (int)ld + n]; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c, float xo) const { x[(long)m * (int)ld + n] = xo + (v + c); } };
struct EpiBiasResidualPl {   // the same, and the new x rows also as planes with one scale + sum of squares per (row, 256-channel half):
    // the A operand and the ScaleNorm statistics of the next layer's to_hidden (PLOUT; N = 512 = two tiles = the two halves)
    const float* b; float* x; long ld; unsigned char* xp; float* xs; float* xss; long M;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { return x[(long)m * (int)ld + n]; }
    __device__ float val(int, int, int, float v, EpiNone, float c, float xo) const { return xo + (v + c); }
    __device__ float* rowout(int, long m, int n0) const { return x + m * ld + n0; }          // the fp32 x rows, from phase 2 (full lines)
    // (the launch moves 1.7 GB and sits at ~82 % of the streaming kernels' HBM rate: see epi_has_rowout in gemm_h3.hpp)
    __device__ tdx::H3PlOut plout(int) const { return tdx::H3PlOut{xp, 2048, xs, xss, M}; }
    __device__ long prow(int, int m) const { return m; } };
struct EpiPosEnc {   // z = acc + pe[s][n]*scale ; x = z                     mossformer2.py:490-496
    const float* pe; const float* scale; float* zout; float* x; int S;
    __device__ float col(int, int) const { return scale[0]; }
    __device__ int row(int, int m) const { return m % S; }
    __device__ float aux(int, int, int n, int s) const { return pe[(long)s * C + n]; }
    __device__ void store(int, int m, int n, float v, int, float sc, float pv) const {
        const float r = v + pv * sc;
        zout[(long)m * C + n] = r; x[(long)m * C + n] = r;
    }
};
struct EpiTanhSig {  // tanh(a+bt)*sigmoid(g+bg)                             mossformer2.py:510
    const float* bias; float* out; long strideZ;
    __device__ Col2 col(int, int c) const { return Col2{bias[c], bias[C + c]}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store2(int z, int m, int c, float a, float g, EpiNone, Col2 k) const {
        out[(long)z * strideZ + (long)m * C + c] = tanhf(a + k.a) * sigmoidf_acc(g + k.b);
    }
};
struct EpiMaskMul {  // mask = relu(acc); EM = E*mask                        mossformer2.py:513-518, :576
    const float* E; float* EM; float* mask; long strideZ;
    __device__ EpiNone col(int, int) const { return EpiNone{}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ float aux(int, int m, int n, EpiNone) const { return E[(long)m * C + n]; }
    __device__ void store(int z, int m, int n, float v, EpiNone, EpiNone, float e) const {
        v = fmaxf(v, 0.f);
        if (mask) mask[(long)z * strideZ + (long)m * C + n] = v;
        EM[(long)z * strideZ + (long)m * C + n] = v * e;
    }
};

// positional tables, fp32 angle product then sin/cos of the ROUNDED angle:
//   pe[s][c]  = sin(s*inv_freq[c]) c<256 ; cos(s*inv_freq[c-256])           mossformer_block.py:68-73
//   rot[s][i] = cos/sin(s*freqs[i])                                         rotary_embedding_torch
__global__ void tables_kernel(const float* __restrict__ inv_freq, const float* __restrict__ freqs, float* __restrict__ pe,
                              float* __restrict__ rc, float* __restrict__ rsn, int S) {
    const int s = blockIdx.x;
    const float t = (float)s;
    for (int c = threadIdx.x; c < 256; c += blockDim.x) {
        const float ang = __fmul_rn(t, inv_freq[c]);
        pe[(long)s * C + c] = sinf(ang);
        pe[(long)s * C + 256 + c] = cosf(ang);
    }
    if (threadIdx.x < 16) {
        const float ang = __fmul_rn(t, freqs[threadIdx.x]);
        rc[s * 16 + threadIdx.x] = cosf(ang);
        rsn[s * 16 + threadIdx.x] = sinf(ang);
    }
}

// stand-alone cal_attention input packing: rotary on the four heads, zero group padding,
// v|u concatenation.
__global__ void pack_heads_kernel(const float* __restrict__ src, const float* __restrict__ freqs, float* __restrict__ dst,
                                  int S, int Sp) {
    const int s = blockIdx.x, b = blockIdx.y, c = threadIdx.x;   // 128 threads
    float v = 0.f;
    if (s < S) {
        const float* row = src + ((long)b * S + s) * 128;
        v = row[c];
        if (c < 32) {
            const float ang = __fmul_rn((float)s, freqs[c >> 1]);
            const float cs = cosf(ang), sn = sinf(ang);
            const float other = row[c ^ 1];
            v = (c & 1) ? v * cs + other * sn : v * cs - other * sn;
        }
    }
    dst[((long)b * Sp + s) * 128 + c] = v;
}
__global__ void concat_vu_kernel(const float* __restrict__ v, const float* __restrict__ u, float* __restrict__ vu, long M, int E) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * 2 * E) return;
    const long m = i / (2 * E);
    const int c = (int)(i % (2 * E));
    vu[i] = c < E ? v[m * E + c] : u[m * E + c - E];
}

// diagnostics (taps): largest |f16| of a planes buffer — how close the values written under a layer's STATIC scale come to the
// f16 range (2^15 after scaling = the bound derived from the weights).  out holds the bits of a non-negative float (atomicMax on
// the bit pattern is then the float maximum).
__global__ void planes_absmax_kernel(const _Float16* __restrict__ p, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)p[i]));
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
__global__ void zero_words_kernel(unsigned* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// ------------------------------------------------------------------ model
struct H3W { const unsigned char* p; const float* s; };
struct LayerW {
    // FLASH
    const float *Whq, *ghq, *bhq, *cw_h, *cw_qk, *gamma, *beta, *Wo, *go, *bo, *cw_o;
    // FSMN
    const float *W1, *b1, *a1, *ln1g, *ln1b, *Wuv, *buv, *cw_uv, *Wl, *bl, *Wp, *w1T, *w2T, *ing, *inb, *pre, *ln2g, *ln2b, *W2, *b2;
    // nn.Linear weights as split-f16 planes + row scales (gemm_h3.hpp), made once at create
    H3W hWhq, hWo, hW1, hWuv, hWl, hWp, hW2;
    // static power-of-two scales of the K-major planes of v|u and lin_k (bounds from the weights, see tdx_mf2_create):
    // sv_* multiplies the value before the f16 split, st points at the device copy of {1/sv_vu, 1/sv_lk}
    float sv_vu, sv_lk;
    const float* st;
};

}  // namespace

constexpr long FORK_ROWS = 1L << 40;    // token rows up to which a FLASH layer forks its q/k-head branch onto the side stream (env TDX_FORK_ROWS):
                                        // −5.7 % at one window per call, −0.6 % at 30 windows (the branches fill each other's tile tails), so: always

struct tdx_mf2 {
    int device;
    int L;
    float* dev_weights;
    size_t n_weights;
    unsigned char* dev_planes;      // split-f16 planes + scales of every nn.Linear weight
    float* dev_static;              // [L][2] inverse static scales
    H3W hWenc, hWout, hWtg, hWdec1;
    std::vector<LayerW> layers;
    const float *encT, *gn1g, *gn1b, *Wenc, *pe_scale, *inv_freq, *rot_freqs, *lnfg, *lnfb, *gn2g, *gn2b, *prelu, *Wout, *bout,
        *Wtg, *btg, *Wdec1, *decT;
    int taps;
    // optional live profiling of the dominant kernel (to_hidden GEMM): event pairs recorded on
    // the forward's stream, read back by tdx_mf2_profile_collect after the caller synchronised.
    std::vector<hipEvent_t> ev0, ev1;
    size_t ev_used;
    // The q/k-head branch of a FLASH layer (conv17<3> -> similarity GEMM) runs on a side stream next to the v|u branch
    // (conv17<4> -> lin_k^T[v|u]); fork / join by events, so the pair is capturable in a HIP graph.  The side stream and its three
    // events belong to the CALLER'S stream, not to the handle: one context per caller stream (created on first use, kept for the
    // handle's life), so that forwards issued concurrently on different (workspace, stream) pairs share nothing mutable
    // (include/tdx.h threading contract).  `mu` guards the map and the profile event vectors.
    struct SideCtx { hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_heads = nullptr, ev_sim = nullptr; };
    std::mutex mu;
    std::map<hipStream_t, SideCtx> side_ctx;
    static constexpr size_t MAX_SIDE_CTX = 64;    // beyond this many distinct caller streams a forward runs unforked (same results)
    // returns nullptr when no context can be had (stream under capture without one, creation failure, cap reached): the forward then
    // issues both branches on the caller's stream — slower at one window per call, never wrong
    SideCtx* side_for(hipStream_t st) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = side_ctx.find(st);
        if (it != side_ctx.end()) return &it->second;
        if (side_ctx.size() >= MAX_SIDE_CTX) return nullptr;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (cs != hipStreamCaptureStatusNone) return nullptr;      // no stream / event creation inside a capture: capture after one eager call on that stream
        SideCtx c;
        if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_heads, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.ev_sim, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (c.ev_sim) hipEventDestroy(c.ev_sim);
            if (c.ev_heads) hipEventDestroy(c.ev_heads);
            if (c.ev_fork) hipEventDestroy(c.ev_fork);
            if (c.side) hipStreamDestroy(c.side);
            return nullptr;
        }
        return &(side_ctx[st] = c);      // (std::map: the address stays valid while other streams are added)
    }
};

namespace {

struct Plan {
    int B, T, S, G, Sp, splits, kchunk, nblk_enc, nblk_gn, nchunk1, nchunk2;
    long M;
    // offsets in floats
    size_t fs, xp, xs, xss, os, oss, zrow, E, z, x, rs, hid, vu, qk4, Abuf, slab, kvu, o, t, hraw, h, hp, hs, vuP, AbufP, Asc, qks, KvuP, kvus, uvpre, uv, f, p, c1, c2, pe, rc, rsn, stat,
        part, tap0, tap1, mask, hdr, total;
};

inline size_t al(size_t n) { return (n + 63) / 64 * 64; }

bool make_plan(const tdx_mf2* h, int B, int T, Plan& P) {
    if (B < 1 || T < 16) return false;
    P.B = B; P.T = T; P.S = (T - 16) / 8 + 1; P.G = (P.S + 255) / 256; P.Sp = P.G * 256; P.M = (long)B * P.S;
    // split-K for the linear-attention K^T[v|u] GEMM (16 N-tiles per sample, 3 blocks per CU
    // resident): aim for >= 1536 workgroups so that the 256 CUs stay full (slabs are 1 MB each)
    int sp = (1536 + 16 * B - 1) / (16 * B); if (sp < 1) sp = 1;
    int maxsp = (P.S + 511) / 512; if (sp > maxsp) sp = maxsp;
    P.splits = sp;
    P.kchunk = ((P.S + sp - 1) / sp + 31) / 32 * 32;
    P.splits = (P.S + P.kchunk - 1) / P.kchunk;
    P.nblk_enc = (P.S + ENC_TOK - 1) / ENC_TOK;
    P.nblk_gn = (P.S + 63) / 64;
    const int ts = ddn_ts(B, P.S);
    P.nchunk1 = (P.S + ts - 1) / ts;
    P.nchunk2 = 2 * ((P.S + 2 * ts - 1) / (2 * ts));
    const size_t M = (size_t)P.M;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += al(n); return o; };
    P.E = take(M * C); P.z = take(M * C); P.x = take(M * C); P.rs = take(M);
    P.hid = take(M * HQ); P.vu = take(M * HID);
    P.qk4 = take((size_t)4 * B * P.Sp * QK);
    P.Abuf = take((size_t)B * P.G * 65536);
    P.slab = take((size_t)B * P.splits * QK * HID);
    P.kvu = take((size_t)B * QK * HID);
    P.o = take(M * 1024); P.t = take(M * C);
    P.hraw = take(M * INNER); P.h = take(M * INNER);
    P.hp = take(M * 1024);      // split-f16 planes of the current GEMM's A operand (<= 1024 channels: 4 KB per row)
    P.hs = take(2 * M);         // and its row scales (2M rows of 512 in the output head)
    // attention on the x3 core: K-major planes of v|u (pad rows zero), row-major planes + row scales of the
    // relu^2 similarity, row scales of quad_q / lin_q / quad_k (their planes live in the qk4 slots), K-major
    // planes of Kvu with one scale per sample (+ its atomicMax word)
    P.vuP = take((size_t)B * P.Sp * HID);
    P.AbufP = take((size_t)B * P.G * 65536);
    P.Asc = take((size_t)B * P.Sp);
    P.qks = take((size_t)3 * B * P.Sp);
    P.KvuP = take((size_t)B * QK * HID);
    P.kvus = take((size_t)B + (size_t)B * (QK * HID / 1024) + 64);
    P.uvpre = take(M * C); P.uv = take(M * C); P.f = take(M * INNER); P.fs = take(M); P.p = take(M * INNER);
    P.xp = take(M * C);          // x as planes (2 KB per row) with per-half scales / sums of squares
    P.xs = take(2 * M); P.xss = take(2 * M);
    P.os = take(8 * M); P.oss = take(8 * M);      // per-(token, 128-channel segment) scales / sums of squares of the gated attention output
    P.zrow = take(1024);         // zero planes row (token shift at the first token of a sample)
    P.c1 = take(M * INNER); P.c2 = take(M * INNER);
    P.pe = take((size_t)P.S * C); P.rc = take((size_t)P.S * 16); P.rsn = take((size_t)P.S * 16);
    P.stat = take((size_t)B * 2 + (size_t)2 * B * 256 * 2 + 64);
    size_t npart = (size_t)B * (P.nblk_enc > P.nblk_gn ? P.nblk_enc : P.nblk_gn) * 2;
    size_t npart_in = (size_t)B * (P.nchunk1 > P.nchunk2 ? P.nchunk1 : P.nchunk2) * 256 * 2;
    if (npart_in > npart) npart = npart_in;
    P.part = take(npart * 2);   // doubles
    if (h->taps) { P.tap0 = take(M * C); P.tap1 = take(M * C); P.mask = take(2 * M * C); P.hdr = take((size_t)h->L * 2 + 64); }
    else { P.tap0 = P.tap1 = P.mask = P.hdr = 0; }
    P.total = off;
    return true;
}

#define LAUNCH_CHECK()                                    \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return tdx::fail_hip(e__, __FILE__, __LINE__); \
    } while (0)

inline dim3 rows4(long M) { return dim3((unsigned)((M + 3) / 4)); }

// conv17 launch helper
template <int MODE>
int launch_conv17(Conv17Args a, int B, hipStream_t st) {
    constexpr int TPT = (MODE == 2 || MODE == 3) ? 32 : 128;      // tokens per thread
    constexpr int TPT_SMALL = TPT / 4;                             // small problems: 4x the blocks (one window per call: 68-136 blocks otherwise)
    const int quads = a.C / 4;
    const int qb = quads >= 256 ? 256 : quads;      // 256, 128 or 32
    const int ty = 256 / qb;
    dim3 block(qb, ty);
    const int s_lim = MODE >= 2 ? a.Sp : a.S;
    const long nblk = (long)(quads / qb) * ((s_lim + ty * TPT - 1) / (ty * TPT)) * B;
    if (nblk < 512) {
        dim3 grid(quads / qb, (s_lim + ty * TPT_SMALL - 1) / (ty * TPT_SMALL), B);
        hipLaunchKernelGGL((conv17_kernel<MODE, TPT_SMALL>), grid, block, 0, st, a);
    } else {
        dim3 grid(quads / qb, (s_lim + ty * TPT - 1) / (ty * TPT), B);
        hipLaunchKernelGGL((conv17_kernel<MODE, TPT>), grid, block, 0, st, a);
    }
    LAUNCH_CHECK();
    return TDX_OK;
}

// quadratic + linear attention core shared by the model and tdx_cal_attention.
// qk4: [4][B][Sp][128] (quad_q, lin_q, quad_k, lin_k; rotary applied, pad rows zero);
// vu: [B*S][2E].  Produces o (gated) or att_v/att_u.
int attention_core(const float* qk4, const float* vu, int B, int S, int E, int splits, int kchunk, float* Abuf, float* slab,
                   float* kvu, float* o, float* att_v, float* att_u, hipStream_t st) {
    const int G = (S + 255) / 256, Sp = G * 256;
    const long hs = (long)B * Sp * QK;
    const float *quad_q = qk4, *lin_q = qk4 + hs, *quad_k = qk4 + 2 * hs, *lin_k = qk4 + 3 * hs;
    {   // A = relu(q k^T / 256)^2 per group                          mossformer_block.py:256-262
        GemmArgs g = make_args(256, 256, make_seg(quad_q, QK, quad_k, QK, QK, 256L * QK, 256L * QK));
        g.seg[0].zdiv = 1;
        EpiQuadSim e{Abuf, G, S, 1.0f / 256.0f};
        if (launch_gemm<false, false, false, false>(g, B * G, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    {   // slab[b][sp] = lin_k^T [v|u] over the split's token range      mossformer_block.py:286,289
        GemmSeg s = make_seg(lin_k, QK, vu, 2L * E, kchunk, (long)Sp * QK, (long)S * 2 * E);
        s.zdiv = splits; s.strideA2 = (long)kchunk * QK; s.strideB2 = (long)kchunk * 2 * E; s.kchunk = kchunk; s.ktotal = S;
        GemmArgs g = make_args(QK, 2 * E, s);
        EpiStore e{slab, 2L * E, (long)QK * 2 * E};
        if (launch_gemm<true, true, false, false>(g, B * splits, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        const long per = (long)QK * 2 * E;
        hipLaunchKernelGGL(kvu_reduce_kernel, dim3((unsigned)((per / 4 + 255) / 256), B), dim3(256), 0, st, slab, kvu, splits, per, (float)S);
        LAUNCH_CHECK();
    }
    {   // [A | lin_q] x [VU ; Kvu] with the gate epilogue              mossformer_block.py:269-294, :217
        GemmSeg s0 = make_seg(Abuf, 256, vu, 2L * E, 256, (long)G * 65536, (long)S * 2 * E);
        s0.zdiv = G; s0.strideA2 = 65536; s0.strideB2 = 256L * 2 * E; s0.kchunk = 256; s0.ktotal = S;
        GemmSeg s1 = make_seg(lin_q, QK, kvu, 2L * E, QK, (long)Sp * QK, (long)QK * 2 * E);
        s1.zdiv = G; s1.strideA2 = 256L * QK; s1.strideB2 = 0; s1.kchunk = 0; s1.ktotal = QK;
        GemmArgs g = make_args(256, E, s0);
        g.seg[1] = s1; g.nseg = 2; g.pair_off = E;
        EpiAttnGate e{vu, o, att_v, att_u, G, S, E};
        hipError_t r;
        if (E % 128 == 0) r = launch_gemm_x6<true, true, false>(g, B * G, e, st);       // split-bf16 x6 core (256-row group = one M tile)
        else r = launch_gemm<false, true, true, false>(g, B * G, e, st);                // fp32-MFMA core (test shapes with E % 128 != 0)
        if (r != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    return TDX_OK;
}

// A/B switch for the half-height two-blocks-per-CU attention kernel (gemm_h3a.hpp; bit-identical to the wide kernel).
// TDX_H3A=0 selects the wide 256-row kernel.
inline bool use_h3a() {
    static const bool on = [] { const char* e = getenv("TDX_H3A"); return e ? atoi(e) != 0 : true; }();
    return on;
}

// The same attention on the split-f16 x3 core (gemm_h3.hpp), E = 1024:
//   qkP : the four heads as planes in [4][B][Sp] x 512 B slots (conv17 MODE 3): quad_q, lin_q, quad_k row-major
//         with row scales qks[3][B*Sp]; lin_k K-major with the static scale st[1]
//   vuP : K-major planes [B][Sp][64][2][32] of v|u with the static scale st[0] (pad rows zero); vu: fp32 (gate)
int attention_core_h3(const unsigned char* qkP, const float* qks, const unsigned char* vuP, const float* vu, const float* st,
                      int B, int S, int E, int splits, int kchunk, float* Abuf, unsigned char* AbufP, float* Asc, float* slab, float* kvu,
                      unsigned char* KvuP, float* kvus, float* o, float* att_v, float* att_u, hipStream_t st_,
                      unsigned char* oP = nullptr, float* os = nullptr, float* oss = nullptr,
                      hipStream_t side = nullptr, hipEvent_t ev_heads = nullptr, hipEvent_t ev_sim = nullptr, const float* u32 = nullptr) {
    // side != nullptr: the similarity GEMM runs on `side` (which already holds the conv17<3> that wrote the heads; `ev_heads` was
    // recorded behind it), next to lin_k^T[v|u] on st_; st_ waits for the heads before it reads lin_k and for the similarity
    // (`ev_sim`) before the attention launch
    const int G = (S + 255) / 256, Sp = G * 256;
    const long hs = (long)B * Sp;                 // rows per head
    const unsigned char *quad_q = qkP, *lin_q = qkP + hs * 512, *quad_k = qkP + 2 * hs * 512, *lin_k = qkP + 3 * hs * 512;
    const float *sq = qks, *slq = qks + hs, *sk = qks + 2 * hs;
    {   // A = relu(q k^T / 256)^2 per group, then its rows as planes                 mossformer_block.py:256-262
        tdx::H3Args g{};
        g.seg[0] = tdx::h3_seg(quad_q, sq, 512, quad_k, sk, 512, QK);
        g.seg[0].strideA = 256L * 512; g.seg[0].strideB = 256L * 512; g.seg[0].strideSA = 256; g.seg[0].strideSB = 256;
        g.nseg = 1; g.M = 256; g.N = 256;
        EpiQuadSimPl e{AbufP, Asc, G, S, 1.0f / 256.0f};         // (planes straight from the epilogue: no fp32 similarity, no split pass)
        if (tdx::launch_gemm_h3x<false, false, false, false>(g, B * G, e, side ? side : st_) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        if (side) {
            if (hipEventRecord(ev_sim, side) != hipSuccess || hipStreamWaitEvent(st_, ev_heads, 0) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        }
    }
    {   // Kvu[b][d][ch] = (1/S) sum_t lin_k[t][d] vu[t][ch], split over token chunks         mossformer_block.py:286,289
        // M = 128 (d): the 128 rows of waves 4-7 do not exist and those waves only feed the ring; lin_k is the A operand — once
        // more as ROW-major planes (kmajor_to_rows_kernel; in the fp32 similarity buffer, which is dead after its split): the
        // transposing LDS reads of a K-major A operand bound this launch — v|u the K-major B operand with full 256-column tiles
        unsigned char* lin_kR = reinterpret_cast<unsigned char*>(Abuf);
        hipLaunchKernelGGL(kmajor_to_rows_kernel, dim3(Sp / 64, B), dim3(256), 0, st_, lin_k, lin_kR, Sp);
        LAUNCH_CHECK();
        tdx::H3Args g{};
        g.seg[0] = tdx::h3_seg(lin_kR, st + 1, 4L * Sp, vuP, st, 4L * 2 * E, kchunk);
        g.seg[0].sa_mul = 0; g.seg[0].sb_mul = 0;
        g.seg[0].zdiv = splits;
        g.seg[0].strideA = (long)Sp * 512; g.seg[0].strideA2 = (long)kchunk * 4;
        g.seg[0].strideB = (long)Sp * 4 * 2 * E; g.seg[0].strideB2 = (long)kchunk * 4 * 2 * E;
        g.seg[0].kchunk = kchunk; g.seg[0].ktotal = Sp;
        g.nseg = 1; g.M = QK; g.N = 2 * E;
        EpiStore e{slab, 2L * E, (long)QK * 2 * E};
        if (use_h3a() && tdx::h3a_fits<false>(g, false)) {      // 128-row tiles, two blocks per CU: every wave has rows
            if (tdx::launch_gemm_h3a<false>(g, B * splits, e, st_) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        } else if (tdx::launch_gemm_h3x<false, true, false, false>(g, B * splits, e, st_) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        const long per = (long)QK * 2 * E;
        const int nb = (int)((per / 4 + 255) / 256);
        float* bmax = kvus + B;          // [B][nb] block maxima
        hipLaunchKernelGGL(kvu_reduce_t_kernel, dim3(nb, B), dim3(256), 0, st_, slab, kvu, splits, per, (float)S, bmax);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(kvu_planes_kernel, dim3((unsigned)((128 * (2 * E / 8) + 255) / 256), B), dim3(256), 0, st_, kvu, bmax, nb, KvuP, kvus, 2 * E);
        LAUNCH_CHECK();
    }
    {   // [A | lin_q] x [VU ; Kvu] with the gate epilogue                                mossformer_block.py:269-294, :217
        tdx::H3Args g{};
        g.seg[0] = tdx::h3_seg(AbufP, Asc, 1024, vuP, st, 4L * 2 * E, 256);
        g.seg[0].sb_mul = 0; g.seg[0].zdiv = G;
        g.seg[0].strideA = (long)G * 256 * 1024; g.seg[0].strideA2 = 256L * 1024;
        g.seg[0].strideSA = (long)G * 256; g.seg[0].strideSA2 = 256;
        g.seg[0].strideB = (long)Sp * 4 * 2 * E; g.seg[0].strideB2 = 256L * 4 * 2 * E;
        g.seg[1] = tdx::h3_seg(lin_q, slq, 512, KvuP, kvus, 4L * 2 * E, QK);
        g.seg[1].sb_mul = 0; g.seg[1].zdiv = G;
        g.seg[1].strideA = (long)Sp * 512; g.seg[1].strideA2 = 256L * 512;
        g.seg[1].strideSA = Sp; g.seg[1].strideSA2 = 256;
        g.seg[1].strideB = (long)QK * 4 * 2 * E; g.seg[1].strideB2 = 0;
        g.seg[1].strideSB = 1; g.seg[1].strideSB2 = 0;
        g.nseg = 2; g.M = 256; g.N = E; g.pair_off = E;
        if (side && hipStreamWaitEvent(st_, ev_sim, 0) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        {   // (diagnostic, timing only, wrong results: TDX_H3_DEBUG & 2 = every group reads the first group's v|u rows — an L2-resident B operand)
            static const int dbg = [] { const char* e = getenv("TDX_H3_DEBUG"); return e ? atoi(e) : 0; }();
            if (dbg & 2) { g.seg[0].strideB = 0; g.seg[0].strideB2 = 0; }
        }
#ifdef TDX_H3_STAMPS
        // diagnostic build: the 4th config-2-sized attention launch records per-block time stamps into $TDX_H3_STAMPS (a file)
        static int stamp_calls = 0;
        static unsigned long long* stamp_buf = nullptr;
        const long stamp_blocks = 8L * ((B * G + 7) / 8) * (use_h3a() ? 16 : 8);
        const bool stamp_now = oP && getenv("TDX_H3_STAMPS") && B * G >= 900 && ++stamp_calls == 4;
        if (stamp_now) {
            hipMalloc((void**)&stamp_buf, stamp_blocks * 64);
            hipMemsetAsync(stamp_buf, 0, stamp_blocks * 64, st_);
            g.stamps = stamp_buf;
        }
        struct StampDump { bool on; long nb; unsigned long long* buf; hipStream_t s;
            ~StampDump() { if (!on) return; hipStreamSynchronize(s); std::vector<unsigned long long> h(nb * 8);
                hipMemcpy(h.data(), buf, nb * 64, hipMemcpyDeviceToHost); FILE* f = fopen(getenv("TDX_H3_STAMPS"), "wb");
                if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); } } } stamp_dump{stamp_now, stamp_blocks, stamp_buf, st_};
#endif
        if (oP) {             // the model: gate operands from the planes, o written as planes with per-segment scales / sums of squares
            if (use_h3a()) {  // half-height tiles, two blocks per CU (gemm_h3a.hpp): one block's epilogue under the other's MFMAs
                // the L2-resident lin_q x Kvu segment FIRST: the ring fills from cache hits while the first v|u rows (HBM) are on their way
                // (ring fill 5.1 -> 4.9 us, k loop 27.1 -> 26.1 us per half tile; config 2: 203.3 vs 205.2 ms).  Only the accumulation
                // order changes; TDX_H3A_SWAP=0 restores the wide kernel's order (then the results are bit-identical to it)
                static const bool swap = [] { const char* e = getenv("TDX_H3A_SWAP"); return e ? atoi(e) != 0 : true; }();
                if (swap) std::swap(g.seg[0], g.seg[1]);
                if (tdx::launch_gemm_h3a<true>(g, B * G, EpiAttnGatePlOut{vuP, st, u32, oP, os, oss, (long)B * S, G, S, Sp, E}, st_) != hipSuccess)
                    return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
                return TDX_OK;
            }
            if (tdx::launch_gemm_h3x<false, true, true, true>(g, B * G, EpiAttnGatePlOut{vuP, st, u32, oP, os, oss, (long)B * S, G, S, Sp, E}, st_) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
            return TDX_OK;
        }
        EpiAttnGate e{vu, o, att_v, att_u, G, S, E};
        if (tdx::launch_gemm_h3x<false, true, true, true>(g, B * G, e, st_) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    return TDX_OK;
}

int ddn_core(const float* p, int B, int S, const float* w1T, const float* w2T, const float* ing, const float* inb,
             const float* pre, float* c1, float* c2, float* stat1, float* stat2, double* part, hipStream_t st) {
    const int ts = ddn_ts(B, S);
    const int nchunk1 = (S + ts - 1) / ts, nchunk2 = 2 * ((S + 2 * ts - 1) / (2 * ts));
    hipLaunchKernelGGL(ddn_conv1_kernel, dim3(nchunk1, B), dim3(256), 0, st, p, w1T, c1, part, S, nchunk1, ts);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(in_finalize_kernel, dim3(B, 16), dim3(256), 0, st, part, stat1, nchunk1, S);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(ddn_conv2_kernel, dim3(nchunk2, B), dim3(256), 0, st, c1, p, stat1, ing, inb, pre, w2T, c2, part, S, nchunk2, ts);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(in_finalize_kernel, dim3(B, 16), dim3(256), 0, st, part, stat2, nchunk2, S);
    LAUNCH_CHECK();
    return TDX_OK;
}

// every nn.Linear / 1x1 Conv1d with a K-contiguous weight goes through the split-bf16 x6 kernel
// (gemm_x6.hpp); W must be readable up to the next multiple of 256 rows.
template <class Epi>
int linear_gemm(const float* A, long lda, const float* W, int M, int N, int K, Epi e, hipStream_t st) {
    GemmArgs g = make_args(M, N, make_seg(A, lda, W, K, K));
    if (launch_gemm_x6<false>(g, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}
// nn.Linear on the split-f16 x3 core (gemm_h3.hpp): A already in planes
template <class Epi>
int linear_h3(const unsigned char* Ap, const float* As, int M, const H3W& W, int N, int K, Epi e, hipStream_t st) {
    tdx::H3Args g{};
    g.seg[0] = tdx::h3_seg(Ap, As, 4L * K, W.p, W.s, 4L * K, K);
    g.nseg = 1; g.M = M; g.N = N;
    if (tdx::launch_gemm_h3<false>(g, 1, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}
// ... A in fp32: split pass first (rows whose producer does not own whole rows)
template <class Epi>
int split_linear_h3(const float* A, long lda, unsigned char* hp, float* hs, int M, const H3W& W, int N, int K, Epi e, hipStream_t st) {
    if (tdx::launch_h3_split_rows(A, lda, hp, hs, M, K, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return linear_h3(hp, hs, M, W, N, K, e, st);
}
template <class Epi>
int linear_gemm_f32(const float* A, long lda, const float* W, int M, int N, int K, Epi e, hipStream_t st) {
    GemmArgs g = make_args(M, N, make_seg(A, lda, W, K, K));
    if (launch_gemm<false, false, false, false>(g, 1, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    return TDX_OK;
}

#define TRY(x) do { int rc__ = (x); if (rc__ != TDX_OK) return rc__; } while (0)

// A/B switch: the depthwise convolution of v|u fused into the to_hidden epilogue (gemm_h3.hpp H3Conv).  TDX_FUSE_CONV=0 restores the
// separate conv17<4> pass (same arithmetic in the same order: bit-identical results).
inline bool fuse_conv() {
    static const bool on = [] { const char* e = getenv("TDX_FUSE_CONV"); return e ? atoi(e) != 0 : true; }();
    return on;
}
template <bool TWOSEG, class Epi>
hipError_t launch_linear_x3(tdx::H3Args g, Epi e, hipStream_t st) {
    return tdx::launch_gemm_h3x<false, false, false, TWOSEG>(g, 1, e, st);
}

}  // namespace

// =====================================================================================
extern "C" {

const char* tdx_version(void) { return "tdx 0.1.0 gfx950"; }
const char* tdx_last_error(void) { return tdx::last_error().c_str(); }

int tdx_mf2_create(const tdx_mf2_config* cfg, const void* blob, size_t blob_bytes, int device, tdx_mf2** out) {
    if (!cfg || !blob || !out) return tdx::fail(TDX_E_INVALID, "tdx_mf2_create: null argument");
    if (cfg->channels != C || cfg->num_spks != 2 || cfg->kernel_size != 16 || cfg->group_size != 256 || cfg->num_blocks < 1)
        return tdx::fail(TDX_E_INVALID, "tdx_mf2_create: unsupported config (need channels=512, spks=2, kernel=16, group=256)");
    tdx::Blob bl;
    if (!bl.parse(blob, blob_bytes)) return tdx::fail(TDX_E_BLOB, "tdx_mf2_create: malformed TDXW blob");
    const int L = cfg->num_blocks;
    std::vector<float> host;
    host.reserve(60u * 1000 * 1000);
    bool ok = true;
    std::string missing;
    auto get = [&](const std::string& name, size_t n) -> const float* {
        const tdx::BlobTensor* t = bl.find(name);
        if (!t || t->numel != n) { ok = false; if (missing.empty()) missing = name; return nullptr; }
        return t->data;
    };
    auto push = [&](const float* p, size_t n) -> size_t {
        size_t o = host.size();
        host.resize(o + al(n), 0.f);
        if (p) memcpy(host.data() + o, p, n * sizeof(float));
        return o;
    };
    // conv weight [Cc,1,k] -> tap-major [k][Cc]
    auto push_tapmajor = [&](const float* w, int Cc, int k) -> size_t {
        size_t o = host.size();
        host.resize(o + al((size_t)Cc * k), 0.f);
        if (w) for (int c = 0; c < Cc; ++c) for (int t = 0; t < k; ++t) host[o + (size_t)t * Cc + c] = w[(size_t)c * k + t];
        return o;
    };
    std::vector<float> stat_host((size_t)L * 2, 1.f);
    std::vector<float> sv_host((size_t)L * 2, 1.f);
    struct Off { size_t Whq, ghq, bhq, cw_h, cw_qk, gamma, beta, Wo, go, bo, cw_o, W1, b1, a1, ln1g, ln1b, Wuv, buv, cw_uv, Wl, bl, Wp, w1T, w2T, ing, inb, pre, ln2g, ln2b, W2, b2; };
    std::vector<Off> offs(L);
    const std::string PFX = "mask_net.mdl.intra_mdl.mossformerM.";
    for (int l = 0; l < L && ok; ++l) {
        Off& o = offs[l];
        const std::string p = PFX + "layers." + std::to_string(l) + ".";
        const float* Wh = get(p + "to_hidden.mdl.1.weight", (size_t)HID * C);
        const float* Wq = get(p + "to_qk.mdl.1.weight", (size_t)QK * C);
        const float* bh = get(p + "to_hidden.mdl.1.bias", HID);
        const float* bq = get(p + "to_qk.mdl.1.bias", QK);
        const float* gh = get(p + "to_hidden.mdl.0.g", 1);
        const float* gq = get(p + "to_qk.mdl.0.g", 1);
        const float* cwh = get(p + "to_hidden.mdl.3.sequential.1.conv.weight", (size_t)HID * 17);
        const float* cwq = get(p + "to_qk.mdl.3.sequential.1.conv.weight", (size_t)QK * 17);
        if (!ok) break;
        o.Whq = push(Wh, (size_t)HID * C); push(Wq, (size_t)QK * C);       // contiguous [2176][512] (HID*C is 64-aligned)
        host.resize(host.size() + (size_t)128 * C, 0.f);                     // zero rows up to 2304: the x6 GEMM reads W in 256-row tiles
        o.ghq = host.size(); host.resize(host.size() + al(HQ));
        for (int i = 0; i < HQ; ++i) host[o.ghq + i] = i < HID ? gh[0] : gq[0];
        o.bhq = host.size(); host.resize(host.size() + al(HQ));
        for (int i = 0; i < HQ; ++i) host[o.bhq + i] = i < HID ? bh[i] : bq[i - HID];
        o.cw_h = push_tapmajor(cwh, HID, 17);
        o.cw_qk = push_tapmajor(cwq, QK, 17);
        (void)get(p + "rotary_pos_emb.freqs", 16);      // one RotaryEmbedding object shared by all layers (mossformer_block.py:453): layer 0's copy is used
        const float* gam = get(p + "qk_offset_scale.gamma", 4 * QK);
        const float* bet = get(p + "qk_offset_scale.beta", 4 * QK);
        o.gamma = push(gam, 4 * QK);
        o.beta = push(bet, 4 * QK);
        if (!ok) break;
        {
            // Static bounds for the K-major planes (gemm_h3.hpp).  After ScaleNorm the (shifted) row has norm
            // <= sqrt(512), so |x_hat . W_n| <= sqrt(512) |g| ||W_n|| (Cauchy-Schwarz) and |silu(p)| <= |p|;
            // the depthwise conv + identity multiplies by at most 1 + sum_t |w_ct|; OffsetScale by |gamma|,
            // + |beta|; the rotary pair rotation by sqrt(2).
            auto ybound = [&](const float* W, const float* bias, int rows, float gs) {
                double best = 0;
                for (int n = 0; n < rows; ++n) {
                    double q2 = 0;
                    for (int k = 0; k < C; ++k) q2 += (double)W[(size_t)n * C + k] * W[(size_t)n * C + k];
                    best = std::max(best, std::sqrt(512.0) * std::fabs(gs) * std::sqrt(q2) + std::fabs(bias[n]));
                }
                return best;
            };
            auto wsum = [&](const float* cw, int rows) {
                double best = 0;
                for (int c = 0; c < rows; ++c) {
                    double a = 1.0;
                    for (int t = 0; t < 17; ++t) a += std::fabs(cw[(size_t)c * 17 + t]);
                    best = std::max(best, a);
                }
                return best;
            };
            const double vub = wsum(cwh, HID) * ybound(Wh, bh, HID, gh[0]);
            double gmax = 0, bmax = 0;
            for (int i = 0; i < QK; ++i) { gmax = std::max(gmax, (double)std::fabs(gam[3 * QK + i])); bmax = std::max(bmax, (double)std::fabs(bet[3 * QK + i])); }
            const double lkb = (wsum(cwq, QK) * ybound(Wq, bq, QK, gq[0]) * gmax + bmax) * std::sqrt(2.0);
            auto pow2scale = [](double bound) { int e = (int)std::floor(std::log2(std::max(bound, 1e-30))); e = std::max(-100, std::min(100, e)); return std::ldexp(1.0, 14 - e); };
            sv_host[l * 2] = (float)pow2scale(vub); sv_host[l * 2 + 1] = (float)pow2scale(lkb);
            stat_host[l * 2] = 1.0f / sv_host[l * 2]; stat_host[l * 2 + 1] = 1.0f / sv_host[l * 2 + 1];
        }
        o.Wo = push(get(p + "to_out.mdl.1.weight", (size_t)C * 1024), (size_t)C * 1024);
        const float* go = get(p + "to_out.mdl.0.g", 1);
        o.go = host.size(); host.resize(host.size() + al(C));
        if (go) for (int i = 0; i < C; ++i) host[o.go + i] = go[0];
        o.bo = push(get(p + "to_out.mdl.1.bias", C), C);
        o.cw_o = push_tapmajor(get(p + "to_out.mdl.3.sequential.1.conv.weight", (size_t)C * 17), C, 17);
        // FSMN
        const std::string q = PFX + "fsmn." + std::to_string(l) + ".";
        o.W1 = push(get(q + "conv1.0.weight", (size_t)INNER * C), (size_t)INNER * C);
        o.b1 = push(get(q + "conv1.0.bias", INNER), INNER);
        o.a1 = push(get(q + "conv1.1.weight", 1), 1);
        o.ln1g = push(get(q + "norm1.weight", INNER), INNER);
        o.ln1b = push(get(q + "norm1.bias", INNER), INNER);
        // fold the FFConvM LayerNorm affine into its Linear:  LN(h)W^T+b = hhat (W diag(g))^T + (b + W beta)
        o.Wuv = host.size(); host.resize(host.size() + al((size_t)C * INNER));
        o.buv = host.size(); host.resize(host.size() + al(C));
        for (int br = 0; br < 2 && ok; ++br) {
            const std::string r = q + "gated_fsmn." + (br == 0 ? "to_u" : "to_v") + ".mdl.";
            const float* lg = get(r + "0.weight", INNER);
            const float* lb = get(r + "0.bias", INNER);
            const float* W = get(r + "1.weight", (size_t)INNER * INNER);
            const float* bb = get(r + "1.bias", INNER);
            if (!ok) break;
            for (int n = 0; n < INNER; ++n) {
                double acc = bb[n];
                for (int k = 0; k < INNER; ++k) {
                    host[o.Wuv + ((size_t)br * INNER + n) * INNER + k] = W[(size_t)n * INNER + k] * lg[k];
                    acc += (double)W[(size_t)n * INNER + k] * (double)lb[k];
                }
                host[o.buv + br * INNER + n] = (float)acc;
            }
        }
        if (!ok) break;
        {
            const float* cu = get(q + "gated_fsmn.to_u.mdl.3.sequential.1.conv.weight", (size_t)INNER * 17);
            const float* cv = get(q + "gated_fsmn.to_v.mdl.3.sequential.1.conv.weight", (size_t)INNER * 17);
            o.cw_uv = host.size(); host.resize(host.size() + al((size_t)17 * C));
            if (cu && cv) for (int c = 0; c < INNER; ++c) for (int t = 0; t < 17; ++t) {
                host[o.cw_uv + (size_t)t * C + c] = cu[c * 17 + t];
                host[o.cw_uv + (size_t)t * C + INNER + c] = cv[c * 17 + t];
            }
        }
        const std::string f = q + "gated_fsmn.fsmn.";
        o.Wl = push(get(f + "linear.weight", (size_t)INNER * INNER), (size_t)INNER * INNER);
        o.bl = push(get(f + "linear.bias", INNER), INNER);
        o.Wp = push(get(f + "project.weight", (size_t)INNER * INNER), (size_t)INNER * INNER);
        o.w1T = push_tapmajor(get(f + "conv.conv1.weight", (size_t)INNER * 39), INNER, 39);
        {   // conv2.weight [256][2][39] -> [39][2][256]
            const float* w2 = get(f + "conv.conv2.weight", (size_t)INNER * 2 * 39);
            o.w2T = host.size(); host.resize(host.size() + al((size_t)39 * 2 * INNER));
            if (w2) for (int j = 0; j < INNER; ++j) for (int ic = 0; ic < 2; ++ic) for (int t = 0; t < 39; ++t)
                host[o.w2T + ((size_t)t * 2 + ic) * INNER + j] = w2[((size_t)j * 2 + ic) * 39 + t];
        }
        o.ing = push(get(f + "conv.norm1.weight", INNER), INNER); push(get(f + "conv.norm2.weight", INNER), INNER);
        o.inb = push(get(f + "conv.norm1.bias", INNER), INNER); push(get(f + "conv.norm2.bias", INNER), INNER);
        o.pre = push(get(f + "conv.prelu1.weight", INNER), INNER); push(get(f + "conv.prelu2.weight", INNER), INNER);
        o.ln2g = push(get(q + "norm2.weight", INNER), INNER);
        o.ln2b = push(get(q + "norm2.bias", INNER), INNER);
        o.W2 = push(get(q + "conv2.weight", (size_t)C * INNER), (size_t)C * INNER);
        o.b2 = push(get(q + "conv2.bias", C), C);
    }
    size_t encT = 0, gn1g = 0, gn1b = 0, Wenc = 0, pes = 0, invf = 0, rotf = 0, lnfg = 0, lnfb = 0, gn2g = 0, gn2b = 0, prelu = 0,
           Wout = 0, bout = 0, Wtg = 0, btg = 0, Wdec1 = 0, decT = 0;
    if (ok) {
        encT = push_tapmajor(get("enc.conv1d.weight", (size_t)C * 16), C, 16);
        gn1g = push(get("mask_net.norm.weight", C), C);
        gn1b = push(get("mask_net.norm.bias", C), C);
        Wenc = push(get("mask_net.conv1d_encoder.weight", (size_t)C * C), (size_t)C * C);
        pes = push(get("mask_net.pos_enc.scale", 1), 1);
        invf = push(get("mask_net.pos_enc.inv_freq", 256), 256);
        rotf = push(get(PFX + "layers.0.rotary_pos_emb.freqs", 16), 16);
        lnfg = push(get("mask_net.mdl.intra_mdl.norm.weight", C), C);
        lnfb = push(get("mask_net.mdl.intra_mdl.norm.bias", C), C);
        gn2g = push(get("mask_net.mdl.intra_norm.weight", C), C);
        gn2b = push(get("mask_net.mdl.intra_norm.bias", C), C);
        prelu = push(get("mask_net.prelu.weight", 1), 1);
        Wout = push(get("mask_net.conv1d_out.weight", (size_t)2 * C * C), (size_t)2 * C * C);
        bout = push(get("mask_net.conv1d_out.bias", 2 * C), 2 * C);
        Wtg = push(get("mask_net.output.0.weight", (size_t)C * C), (size_t)C * C);
        push(get("mask_net.output_gate.0.weight", (size_t)C * C), (size_t)C * C);
        btg = push(get("mask_net.output.0.bias", C), C);
        push(get("mask_net.output_gate.0.bias", C), C);
        Wdec1 = push(get("mask_net.conv1_decoder.weight", (size_t)C * C), (size_t)C * C);
        decT = push_tapmajor(get("dec.weight", (size_t)C * 16), C, 16);
    }
    if (!ok) return tdx::fail(TDX_E_BLOB, "tdx_mf2_create: tensor missing or wrong size: " + missing);
    {   // strict both ways, like load_state_dict(strict=True) at base_model.py:63
        const std::string extra = bl.first_unused();
        if (!extra.empty()) return tdx::fail(TDX_E_BLOB, "tdx_mf2_create: unexpected tensor in the blob: " + extra);
    }

    tdx::DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) return tdx::fail_hip(e, __FILE__, __LINE__);
    float* dev = nullptr;
    e = hipMalloc(&dev, host.size() * sizeof(float));
    if (e != hipSuccess) return tdx::fail_hip(e, __FILE__, __LINE__);
    e = hipMemcpy(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(dev); return tdx::fail_hip(e, __FILE__, __LINE__); }

    tdx_mf2* h = new tdx_mf2();
    h->dev_planes = nullptr; h->dev_static = nullptr;
    h->device = device; h->L = L; h->dev_weights = dev; h->n_weights = host.size(); h->taps = 0; h->ev_used = 0;
    h->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        const Off& o = offs[l]; LayerW& w = h->layers[l];
        w.Whq = dev + o.Whq; w.ghq = dev + o.ghq; w.bhq = dev + o.bhq; w.cw_h = dev + o.cw_h; w.cw_qk = dev + o.cw_qk;
        w.gamma = dev + o.gamma; w.beta = dev + o.beta; w.Wo = dev + o.Wo; w.go = dev + o.go; w.bo = dev + o.bo; w.cw_o = dev + o.cw_o;
        w.W1 = dev + o.W1; w.b1 = dev + o.b1; w.a1 = dev + o.a1; w.ln1g = dev + o.ln1g; w.ln1b = dev + o.ln1b;
        w.Wuv = dev + o.Wuv; w.buv = dev + o.buv; w.cw_uv = dev + o.cw_uv; w.Wl = dev + o.Wl; w.bl = dev + o.bl; w.Wp = dev + o.Wp;
        w.w1T = dev + o.w1T; w.w2T = dev + o.w2T; w.ing = dev + o.ing; w.inb = dev + o.inb; w.pre = dev + o.pre;
        w.ln2g = dev + o.ln2g; w.ln2b = dev + o.ln2b; w.W2 = dev + o.W2; w.b2 = dev + o.b2;
    }
    h->encT = dev + encT; h->gn1g = dev + gn1g; h->gn1b = dev + gn1b; h->Wenc = dev + Wenc; h->pe_scale = dev + pes;
    h->inv_freq = dev + invf; h->rot_freqs = dev + rotf; h->lnfg = dev + lnfg; h->lnfb = dev + lnfb; h->gn2g = dev + gn2g;
    h->gn2b = dev + gn2b; h->prelu = dev + prelu; h->Wout = dev + Wout; h->bout = dev + bout; h->Wtg = dev + Wtg; h->btg = dev + btg;
    h->Wdec1 = dev + Wdec1; h->decT = dev + decT;
    e = hipMalloc(&h->dev_static, stat_host.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->dev_static, stat_host.data(), stat_host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    for (int l = 0; l < L; ++l) { h->layers[l].sv_vu = sv_host[l * 2]; h->layers[l].sv_lk = sv_host[l * 2 + 1]; h->layers[l].st = h->dev_static + l * 2; }
    // ---- split every nn.Linear weight [N][K] into f16 planes + row scales, once
    {
        struct Job { const float* w; int N, K; H3W* dst; };
        std::vector<Job> jobs;
        for (int l = 0; l < L; ++l) {
            LayerW& w = h->layers[l];
            jobs.push_back({w.Whq, HQ, C, &w.hWhq}); jobs.push_back({w.Wo, C, 1024, &w.hWo}); jobs.push_back({w.W1, INNER, C, &w.hW1});
            jobs.push_back({w.Wuv, C, INNER, &w.hWuv}); jobs.push_back({w.Wl, INNER, INNER, &w.hWl});
            jobs.push_back({w.Wp, INNER, INNER, &w.hWp}); jobs.push_back({w.W2, C, INNER, &w.hW2});
        }
        jobs.push_back({h->Wenc, C, C, &h->hWenc}); jobs.push_back({h->Wout, 2 * C, C, &h->hWout});
        jobs.push_back({h->Wtg, 2 * C, C, &h->hWtg}); jobs.push_back({h->Wdec1, C, C, &h->hWdec1});
        size_t bytes = 0;
        for (const Job& j : jobs) bytes += (size_t)j.N * j.K * 4 + (size_t)al(j.N) * 4;
        e = hipMalloc(&h->dev_planes, bytes);
        if (e != hipSuccess) { hipFree(dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
        unsigned char* q = h->dev_planes;
        for (const Job& j : jobs) {
            float* sc = (float*)(q + (size_t)j.N * j.K * 4);
            e = tdx::launch_h3_split_rows(j.w, j.K, q, sc, j.N, j.K, nullptr);
            if (e != hipSuccess) { hipFree(h->dev_planes); hipFree(dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
            j.dst->p = q; j.dst->s = sc;
            q += (size_t)j.N * j.K * 4 + (size_t)al(j.N) * 4;
        }
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { hipFree(h->dev_planes); hipFree(dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    }
    *out = h;
    return TDX_OK;
}

int tdx_mf2_destroy(tdx_mf2* h) {
    if (!h) return TDX_OK;
    if (h->dev_weights) hipFree(h->dev_weights);
    if (h->dev_planes) hipFree(h->dev_planes);
    if (h->dev_static) hipFree(h->dev_static);
    for (auto e : h->ev0) hipEventDestroy(e);
    for (auto e : h->ev1) hipEventDestroy(e);
    for (auto& kv : h->side_ctx) {
        hipEventDestroy(kv.second.ev_fork); hipEventDestroy(kv.second.ev_heads); hipEventDestroy(kv.second.ev_sim);
        hipStreamDestroy(kv.second.side);
    }
    delete h;
    return TDX_OK;
}

int tdx_mf2_enable_taps(tdx_mf2* h, int on) {
    if (!h) return tdx::fail(TDX_E_INVALID, "null handle");
    h->taps = on ? 1 : 0;
    return TDX_OK;
}

int tdx_mf2_profile_enable(tdx_mf2* h, int max_records) {
    if (!h || max_records < 0) return tdx::fail(TDX_E_INVALID, "tdx_mf2_profile_enable: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto e : h->ev0) hipEventDestroy(e);
    for (auto e : h->ev1) hipEventDestroy(e);
    h->ev0.assign(max_records, nullptr); h->ev1.assign(max_records, nullptr); h->ev_used = 0;
    for (int i = 0; i < max_records; ++i) {
        if (hipEventCreate(&h->ev0[i]) != hipSuccess || hipEventCreate(&h->ev1[i]) != hipSuccess)
            return tdx::fail(TDX_E_HIP, "tdx_mf2_profile_enable: hipEventCreate failed");
    }
    return TDX_OK;
}

int tdx_mf2_profile_collect(tdx_mf2* h, double* total_ms, int* launches) {
    if (!h || !total_ms || !launches) return tdx::fail(TDX_E_INVALID, "tdx_mf2_profile_collect: null argument");
    std::lock_guard<std::mutex> lk(h->mu);
    double tot = 0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]);
        if (e != hipSuccess) return tdx::fail_hip(e, __FILE__, __LINE__);
        tot += ms;
    }
    *total_ms = tot; *launches = (int)h->ev_used; h->ev_used = 0;
    return TDX_OK;
}

size_t tdx_mf2_workspace_bytes(const tdx_mf2* h, int B, int T) {
    Plan P;
    if (!h || !make_plan(h, B, T, P)) return 0;
    return P.total * sizeof(float);
}

double tdx_mf2_flops(const tdx_mf2* h, int B, int T) {
    Plan P;
    if (!h || !make_plan(h, B, T, P)) return 0.0;
    const double S = P.S, Sp = P.Sp, Bd = B;
    // 2*MAC accounting, same terms as SURVEY.md Appendix D (torch FlopCounter convention)
    double per_layer = 2.0 * S * C * HQ                       // to_hidden + to_qk
                       + 2.0 * Sp * 256 * QK                   // quad sim
                       + 2.0 * Sp * 256 * HID                  // A [v|u]
                       + 2.0 * Sp * QK * HID * 2               // lin K^T[v|u] and lin_q K
                       + 2.0 * S * 1024 * C                    // to_out
                       + 2.0 * S * (HQ + C + C) * 17           // depthwise k=17
                       + 2.0 * S * (C * INNER * 2 + 4.0 * INNER * INNER)   // fsmn 1x1 + 4 linears
                       + 2.0 * S * INNER * 39 * 3;             // depthwise k=39 (conv1: 1 tap set, conv2: 2)
    double head = 2.0 * S * (C * C + C * 2 * C + 2.0 * (2 * C * C) + 2.0 * C * C) + 2.0 * S * C * 16 * 3;
    return Bd * (h->L * per_layer + head);
}

int tdx_mf2_forward(tdx_mf2* h, const float* wav, int B, int T, float* out, void* ws_, size_t ws_bytes, void* stream) {
    if (!h || !wav || !out || !ws_) return tdx::fail(TDX_E_INVALID, "tdx_mf2_forward: null argument");
    Plan P;
    if (!make_plan(h, B, T, P)) return tdx::fail(TDX_E_INVALID, "tdx_mf2_forward: need B>=1 and T>=16");
    if (ws_bytes < P.total * sizeof(float)) return tdx::fail(TDX_E_WORKSPACE, "tdx_mf2_forward: workspace too small");
    tdx::DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return tdx::fail_hip(guard.err, __FILE__, __LINE__);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)ws_;
    const int S = P.S, G = P.G, Sp = P.Sp;
    const long M = P.M;
    float *E = ws + P.E, *z = ws + P.z, *x = ws + P.x, *rs = ws + P.rs, *hid = ws + P.hid, *vu = ws + P.vu, *qk4 = ws + P.qk4,
          *Abuf = ws + P.Abuf, *slab = ws + P.slab, *kvu = ws + P.kvu, *o = ws + P.o, *t = ws + P.t, *hraw = ws + P.hraw,
          *hh = ws + P.h, *hs = ws + P.hs, *uvpre = ws + P.uvpre, *uv = ws + P.uv, *f = ws + P.f, *p = ws + P.p,
          *c1 = ws + P.c1, *c2 = ws + P.c2, *pe = ws + P.pe, *rc = ws + P.rc, *rsn = ws + P.rsn, *stat = ws + P.stat;
    double* part = (double*)(ws + P.part);
    unsigned char* hp = (unsigned char*)(ws + P.hp);
    unsigned char *vuP = (unsigned char*)(ws + P.vuP), *AbufP = (unsigned char*)(ws + P.AbufP), *KvuP = (unsigned char*)(ws + P.KvuP);
    float *Asc = ws + P.Asc, *qks = ws + P.qks, *kvus = ws + P.kvus;
    unsigned char *xp = (unsigned char*)(ws + P.xp), *oP = (unsigned char*)(ws + P.o), *zrow = (unsigned char*)(ws + P.zrow);
    float *xs = ws + P.xs, *xss = ws + P.xss, *os = ws + P.os, *oss = ws + P.oss;
    unsigned char* fP = (unsigned char*)(ws + P.f);     // f = relu(linear(x_u)) lives as planes only (1 KB per row, like the fp32 row)
    float* fs = ws + P.fs;
    float* gnstat = stat;               // [B][2]
    float* stat1 = stat + al(2 * B);    // [B][256][2]
    float* stat2 = stat1 + (size_t)B * 512;
    (void)G;

    // fork / join context of THIS caller stream (nullptr: run unforked)
    static const long fork_rows = [] { const char* e = getenv("TDX_FORK_ROWS"); return e ? atol(e) : FORK_ROWS; }();
    tdx_mf2::SideCtx* sc = M <= fork_rows ? h->side_for(st) : nullptr;

    hipLaunchKernelGGL(tables_kernel, dim3(S), dim3(256), 0, st, h->inv_freq, h->rot_freqs, pe, rc, rsn, S);
    LAUNCH_CHECK();
    if (fuse_conv() && Sp > S) {      // the fused epilogue writes the S real rows of the v|u planes only
        hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(Sp - S, B), dim3(256), 0, st, vuP, S, Sp, 4L * HID);
        LAUNCH_CHECK();
    }
    unsigned* hdr = h->taps ? reinterpret_cast<unsigned*>(ws + P.hdr) : nullptr;       // [L][2] largest scaled |f16| of v|u and lin_k (tap "headroom")
    if (hdr) { hipLaunchKernelGGL(zero_words_kernel, dim3((2 * h->L + 255) / 256), dim3(256), 0, st, hdr, 2 * h->L); LAUNCH_CHECK(); }
    // ---- encoder + GroupNorm + 1x1 conv + positional encoding   (mossformer2.py:573, :487-496)
    hipLaunchKernelGGL(encoder_kernel, dim3(P.nblk_enc, B), dim3(128), 0, st, wav, h->encT, E, part, T, S, P.nblk_enc);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, st, part, gnstat, P.nblk_enc, (double)S * C, 1e-8);
    LAUNCH_CHECK();
    hipLaunchKernelGGL((gn_apply_kernel<0>), dim3((unsigned)((M * 128 + 255) / 256)), dim3(256), 0, st, E, gnstat, h->gn1g, h->gn1b,
                       (const float*)nullptr, (const float*)nullptr, t, M, S);
    LAUNCH_CHECK();
    TRY(split_linear_h3(t, C, hp, hs, (int)M, h->hWenc, C, C, EpiPosEnc{pe, h->pe_scale, z, x, S}, st));
    // x as planes with per-half scales and sums of squares: inside the stack the producers of x keep them up to date
    hipLaunchKernelGGL(xplanes_kernel, dim3((unsigned)((2 * M + 3) / 4)), dim3(256), 0, st, x, xp, xs, xss, M, reinterpret_cast<float*>(zrow));
    LAUNCH_CHECK();

    for (int l = 0; l < h->L; ++l) {
        const LayerW& w = h->layers[l];
        // ================= FLASH_ShareA_FFConvM  (mossformer_block.py:191-220)
        // token shift + ScaleNorm: the GEMM reads the planes of x as TWO K segments — channels 0..255 from the row above
        // (zero row at the first token of a sample), channels 256..511 from the row itself — each with its own row scale;
        // the ScaleNorm factor comes from the halves' sums of squares (no pass over x)
        {
            // (live profiling is a single-caller diagnostic: the slot is claimed under the lock, the events are recorded on this stream)
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            {
                std::lock_guard<std::mutex> lk(h->mu);
                if (h->ev_used < h->ev0.size()) { pe0 = h->ev0[h->ev_used]; pe1 = h->ev1[h->ev_used]; h->ev_used++; }
            }
            const bool prof = pe0 != nullptr;
            if (prof) hipEventRecord(pe0, st);
            tdx::H3Args g{};
            g.seg[0] = tdx::h3_seg(xp, xs, 4L * C, w.hWhq.p, w.hWhq.s, 4L * C, C / 2);
            g.seg[0].a_shift = -1; g.seg[0].a_period = S; g.seg[0].a_zero = zrow;
            g.seg[1] = tdx::h3_seg(xp + 2 * C, xs + M, 4L * C, w.hWhq.p + 2 * C, w.hWhq.s, 4L * C, C / 2);
            g.nseg = 2; g.M = (int)M; g.N = HQ;
            if (fuse_conv()) {      // SiLU + depthwise conv of v|u in the epilogue: planes + fp32 u straight from the GEMM (H3Conv)
                EpiHiddenConv e{xss, M, S, 0.044194173824159216f, w.ghq, w.bhq, hid, HQ, w.cw_h, vuP, w.sv_vu, vu, Sp};
                const hipError_t rc = tdx::launch_gemm_h3x<false, false, false, true>(g, 1, e, st);
                if (rc != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
            } else {
                EpiHiddenSN<2, true> e{xss, M, S, 0.044194173824159216f, w.ghq, w.bhq, hid, HQ};       // (SiLU in conv17)
                if (launch_linear_x3<true>(g, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
            }
            if (prof) hipEventRecord(pe1, st);
        }
        // small forwards: the q/k-head branch (conv17<3> -> similarity) on the side stream, next to the v|u branch
        const bool fork = sc != nullptr;
        hipStream_t sq = fork ? sc->side : st;
        if (fork && (hipEventRecord(sc->ev_fork, st) != hipSuccess || hipStreamWaitEvent(sc->side, sc->ev_fork, 0) != hipSuccess))
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        {
            Conv17Args a{};
            a.in = hid; a.ld_in = HQ; a.col0 = 0; a.wT = w.cw_h; a.C = HID; a.out = vu; a.out_c0 = HID / 2; a.ld_out = HID / 2; a.S = S; a.Sp = Sp;
            a.hp = vuP; a.sv = w.sv_vu; a.silu_in = 1;
            if (!fuse_conv())
                TRY(launch_conv17<4>(a, B, st));      // v|u as K-major planes (GEMM operands; v also as the gate's operand) + u in fp32 (the gate's sigmoid)
            Conv17Args q{};
            q.in = hid; q.ld_in = HQ; q.col0 = HID; q.wT = w.cw_qk; q.C = QK; q.S = S; q.Sp = Sp; q.gamma = w.gamma; q.beta = w.beta;
            q.rot_cos = rc; q.rot_sin = rsn; q.head_stride = (long)B * Sp * QK;
            q.hp = (unsigned char*)qk4; q.hs = qks; q.sv = w.sv_lk; q.silu_in = 1;
            TRY(launch_conv17<3>(q, B, sq));          // the four heads as planes
            if (hdr) {
                hipLaunchKernelGGL(planes_absmax_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<const _Float16*>(vuP), (long)B * Sp * HID * 2, hdr + 2 * l);
                LAUNCH_CHECK();
                hipLaunchKernelGGL(planes_absmax_kernel, dim3(256), dim3(256), 0, sq, reinterpret_cast<const _Float16*>(qk4) + 3L * B * Sp * 256, (long)B * Sp * 256,
                                   hdr + 2 * l + 1);
                LAUNCH_CHECK();
            }
            if (fork && hipEventRecord(sc->ev_heads, sc->side) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        }
        TRY(attention_core_h3((const unsigned char*)qk4, qks, vuP, nullptr, w.st, B, S, 1024, P.splits, P.kchunk, Abuf, AbufP, Asc, slab, kvu, KvuP,
                              kvus, nullptr, nullptr, nullptr, st, oP, os, oss, fork ? sc->side : nullptr, fork ? sc->ev_heads : nullptr, fork ? sc->ev_sim : nullptr, vu));
        {   // to_out: A = the planes of o with one row scale per 128-channel segment; ScaleNorm from the segments' sums of squares
            tdx::H3Args g{};
            g.seg[0] = tdx::h3_seg(oP, os, 4L * 1024, w.hWo.p, w.hWo.s, 4L * 1024, 1024);
            g.seg[0].segk = 128; g.seg[0].strideSeg = M;
            g.nseg = 1; g.M = (int)M; g.N = C;
            EpiHiddenSN<8, false> e{oss, M, S, 0.03125f, w.go, w.bo, t, C};
            if (launch_linear_x3<false>(g, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        }
        {
            Conv17Args a{};
            a.in = t; a.ld_in = C; a.col0 = 0; a.wT = w.cw_o; a.C = C; a.out = x; a.ld_out = C; a.S = S; a.Sp = Sp; a.silu_in = 1;
            a.xp = xp; a.xs = xs; a.xs_stride = M;       // the new x also as planes (per-half scales): W1's A operand
            TRY(launch_conv17<1>(a, B, st));
        }
        if (h->taps && l == 0) hipMemcpyAsync(ws + P.tap0, x, M * C * sizeof(float), hipMemcpyDeviceToDevice, st);
        // ================= GatedFSMNBlockDilated  (mossformer_block.py:419-425)
        {
            tdx::H3Args g{};
            g.seg[0] = tdx::h3_seg(xp, xs, 4L * C, w.hW1.p, w.hW1.s, 4L * C, C);
            g.seg[0].segk = 256; g.seg[0].strideSeg = M;
            g.nseg = 1; g.M = (int)M; g.N = INNER;
            if (tdx::launch_gemm_h3x<false, false, false, false>(g, 1, EpiBiasPrelu{w.b1, w.a1, hraw, INNER}, st) != hipSuccess)
                return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        }
        hipLaunchKernelGGL((layernorm_kernel<INNER, true>), rows4(M), dim3(256), 0, st, hraw, w.ln1g, w.ln1b, hh, (float*)nullptr, M, 1e-5f,
                           hp, hs);
        LAUNCH_CHECK();
        TRY(linear_h3(hp, hs, (int)M, w.hWuv, C, INNER, EpiBiasSilu{w.buv, uvpre, C}, st));
        {
            Conv17Args a{};
            a.in = uvpre; a.ld_in = C; a.col0 = 0; a.wT = w.cw_uv; a.C = C; a.out = uv; a.ld_out = C; a.S = S; a.Sp = Sp;
            a.xp = hp; a.xs = hs;                          // x_u (channels 0..255) also as planes: the A operand of fsmn.linear
            TRY(launch_conv17<0>(a, B, st));
        }
        TRY(linear_h3(hp, hs, (int)M, w.hWl, INNER, INNER, EpiBiasReluPl{w.bl, fP, fs}, st));       // f = relu(.) as planes
        TRY(linear_h3(fP, fs, (int)M, w.hWp, INNER, INNER, EpiBias{nullptr, p, INNER}, st));
        TRY(ddn_core(p, B, S, w.w1T, w.w2T, w.ing, w.inb, w.pre, c1, c2, stat1, stat2, part, st));
        hipLaunchKernelGGL(fsmn_tail_kernel, rows4(M), dim3(256), 0, st, c2, stat2, w.ing + INNER, w.inb + INNER, w.pre + INNER, uv, hh,
                           w.ln2g, w.ln2b, hp, hs, M, S);
        LAUNCH_CHECK();
        TRY(linear_h3(hp, hs, (int)M, w.hW2, C, INNER, EpiBiasResidualPl{w.b2, x, C, xp, xs, xss, M}, st));    // x += ..., and its planes / statistics
        if (h->taps && l == 0) hipMemcpyAsync(ws + P.tap1, x, M * C * sizeof(float), hipMemcpyDeviceToDevice, st);
    }

    // ---- LayerNorm, GroupNorm + skip, PReLU, output head   (mossformer2.py:320, :389-396, :500-521)
    hipLaunchKernelGGL((layernorm_kernel<C, false>), rows4(M), dim3(256), 0, st, x, h->lnfg, h->lnfb, t, (float*)nullptr, M, 1e-6f);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(rowblock_stats_kernel, dim3(P.nblk_gn, B), dim3(256), 0, st, t, part, S, 64, P.nblk_gn);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, st, part, gnstat, P.nblk_gn, (double)S * C, 1e-8);
    LAUNCH_CHECK();
    float* r = uvpre;    // [M,512] scratch
    hipLaunchKernelGGL((gn_apply_kernel<1>), dim3((unsigned)((M * 128 + 255) / 256)), dim3(256), 0, st, t, gnstat, h->gn2g, h->gn2b, z,
                       h->prelu, r, M, S);
    LAUNCH_CHECK();
    float* r2 = o;       // [2][M][512]: the conv1d_out rows of the two speakers
    TRY(split_linear_h3(r, C, hp, hs, (int)M, h->hWout, 2 * C, C, EpiBiasHalves{h->bout, r2, M}, st));
    float* gate = vu;    // [2][M][512]
    {
        if (tdx::launch_h3_split_rows(r2, C, hp, hs, 2 * M, C, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        tdx::H3Args g{};
        g.seg[0] = tdx::h3_seg(hp, hs, 4L * C, h->hWtg.p, h->hWtg.s, 4L * C, C);
        g.seg[0].strideA = M * 4L * C; g.seg[0].strideSA = M;
        g.nseg = 1; g.M = (int)M; g.N = C; g.pair_off = C;
        if (tdx::launch_gemm_h3<true>(g, 2, EpiTanhSig{h->btg, gate, M * C}, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    float* EM = hid;     // [2][M][512]
    {
        if (tdx::launch_h3_split_rows(gate, C, hp, hs, 2 * M, C, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        tdx::H3Args g{};
        g.seg[0] = tdx::h3_seg(hp, hs, 4L * C, h->hWdec1.p, h->hWdec1.s, 4L * C, C);
        g.seg[0].strideA = M * 4L * C; g.seg[0].strideSA = M;
        g.nseg = 1; g.M = (int)M; g.N = C;
        EpiMaskMul e{E, EM, h->taps ? ws + P.mask : nullptr, M * C};
        if (tdx::launch_gemm_h3<false>(g, 2, e, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    float* D = t;        // [2][M][16]
    hipLaunchKernelGGL(decoder_dot_kernel, dim3((unsigned)((2 * M + 4 * DEC_ROWS - 1) / (4 * DEC_ROWS))), dim3(256), 0, st, EM, h->decT, D, 2 * M);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(decoder_ola_kernel, dim3((unsigned)(((long)B * 2 * T + 255) / 256)), dim3(256), 0, st, D, out, B, S, T);
    LAUNCH_CHECK();
    return TDX_OK;
}

int tdx_mf2_tap(tdx_mf2* h, const char* name, int B, int T, void* ws_, float* dst, size_t dst_elems, size_t* n, void* stream) {
    if (!h || !name || !ws_ || !dst) return tdx::fail(TDX_E_INVALID, "tdx_mf2_tap: null argument");
    Plan P;
    if (!make_plan(h, B, T, P)) return tdx::fail(TDX_E_INVALID, "tdx_mf2_tap: bad shape");
    float* ws = (float*)ws_;
    const size_t MC = (size_t)P.M * C;
    size_t off, cnt = MC;
    std::string s(name);
    if (s == "enc") off = P.E;
    else if (s == "z") off = P.z;
    else if (s == "after_stack") off = P.x;
    else if (!h->taps) return tdx::fail(TDX_E_INVALID, "tdx_mf2_tap: enable taps before forward for this tap");
    else if (s == "after_flash0") off = P.tap0;
    else if (s == "after_fsmn0") off = P.tap1;
    else if (s == "mask") { off = P.mask; cnt = 2 * MC; }
    else if (s == "headroom") { off = P.hdr; cnt = (size_t)h->L * 2; }     // per layer: largest |f16| written under the static scales of v|u, lin_k
    else return tdx::fail(TDX_E_INVALID, "tdx_mf2_tap: unknown tap " + s);
    if (n) *n = cnt;
    if (dst_elems < cnt) return tdx::fail(TDX_E_INVALID, "tdx_mf2_tap: destination too small");
    hipError_t e = hipMemcpyAsync(dst, ws + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return tdx::fail_hip(e, __FILE__, __LINE__);
    return TDX_OK;
}

// ------------------------------------------------------------------ stand-alone ops
static void attn_plan(int B, int S, int E, int& splits, int& kchunk, size_t& qk4, size_t& vu, size_t& Abuf, size_t& slab, size_t& kvu,
                      size_t& total) {
    const int G = (S + 255) / 256, Sp = G * 256;
    int sp = (1536 + (2 * E / 128) * B - 1) / ((2 * E / 128) * B); if (sp < 1) sp = 1;
    int maxsp = (S + 511) / 512; if (sp > maxsp) sp = maxsp;
    kchunk = ((S + sp - 1) / sp + 31) / 32 * 32;
    splits = (S + kchunk - 1) / kchunk;
    size_t off = 0;
    auto take = [&](size_t nn) { size_t o = off; off += al(nn); return o; };
    qk4 = take((size_t)4 * B * Sp * QK); vu = take((size_t)B * S * 2 * E); Abuf = take((size_t)B * G * 65536);
    slab = take((size_t)B * splits * QK * 2 * E); kvu = take((size_t)B * QK * 2 * E);
    if (E % 128 == 0) {   // split-f16 x3 path (as the model): planes of the heads, of v|u, of the similarity and of Kvu, and scales
        take((size_t)4 * B * Sp * QK); take((size_t)B * Sp * 2 * E); take((size_t)B * G * 65536); take((size_t)B * QK * 2 * E);
        take((size_t)4 * B * Sp); take((size_t)B + (size_t)B * (QK * 2 * E / 1024) + 64); take(64);
    }
    total = off;
}

size_t tdx_cal_attention_workspace_bytes(int B, int S, int E) {
    if (B < 1 || S < 1 || E < 64 || E % 64) return 0;
    int sp, kc; size_t a, b, c, d, e, tot;
    attn_plan(B, S, E, sp, kc, a, b, c, d, e, tot);
    return tot * sizeof(float);
}

int tdx_cal_attention(const float* quad_q, const float* lin_q, const float* quad_k, const float* lin_k, const float* v, const float* u,
                      const float* freqs, int B, int S, int E, float* att_v, float* att_u, void* ws_, size_t ws_bytes, void* stream) {
    if (!quad_q || !lin_q || !quad_k || !lin_k || !v || !u || !freqs || !att_v || !att_u || !ws_)
        return tdx::fail(TDX_E_INVALID, "tdx_cal_attention: null argument");
    if (B < 1 || S < 1 || E < 64 || E % 64) return tdx::fail(TDX_E_INVALID, "tdx_cal_attention: need E % 64 == 0");
    int splits, kchunk; size_t oq, ovu, oA, oslab, okvu, tot;
    attn_plan(B, S, E, splits, kchunk, oq, ovu, oA, oslab, okvu, tot);
    if (ws_bytes < tot * sizeof(float)) return tdx::fail(TDX_E_WORKSPACE, "tdx_cal_attention: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)ws_;
    const int G = (S + 255) / 256, Sp = G * 256;
    const long hs = (long)B * Sp * QK;
    const float* heads[4] = {quad_q, lin_q, quad_k, lin_k};
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(pack_heads_kernel, dim3(Sp, B), dim3(128), 0, st, heads[i], freqs, ws + oq + i * hs, S, Sp);
        LAUNCH_CHECK();
    }
    const long M = (long)B * S;
    hipLaunchKernelGGL(concat_vu_kernel, dim3((unsigned)((M * 2 * E + 255) / 256)), dim3(256), 0, st, v, u, ws + ovu, M, E);
    LAUNCH_CHECK();
    if (E % 128) return attention_core(ws + oq, ws + ovu, B, S, E, splits, kchunk, ws + oA, ws + oslab, ws + okvu, nullptr, att_v, att_u, st);
    // ---- the model's path (split-f16 x3 core): planes made here by the generic producers, with the exact maxima of
    // lin_k and v|u standing in for the model's static bounds
    float* p = ws + okvu + al((size_t)B * QK * 2 * E);
    unsigned char* qkP = (unsigned char*)p; p += al((size_t)4 * B * Sp * QK);
    unsigned char* vuP = (unsigned char*)p; p += al((size_t)B * Sp * 2 * E);
    unsigned char* AbufP = (unsigned char*)p; p += al((size_t)B * G * 65536);
    unsigned char* KvuP = (unsigned char*)p; p += al((size_t)B * QK * 2 * E);
    float* qks = p; p += al((size_t)4 * B * Sp);
    float* kvus = p; p += al((size_t)B + (size_t)B * (QK * 2 * E / 1024) + 64);
    float* stt = p;                       // [0..1] inverse scales of v|u, lin_k ; [2..3] their maxima (bits)
    float* Asc = qks + 3L * B * Sp;
    unsigned* mx = reinterpret_cast<unsigned*>(stt + 2);
    if (hipMemsetAsync(mx, 0, 2 * sizeof(unsigned), st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    hipLaunchKernelGGL(tdx::h3_absmax_kernel<0>, dim3(1024), dim3(256), 0, st, ws + ovu, M * 2 * E, mx);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(tdx::h3_absmax_kernel<0>, dim3(1024), dim3(256), 0, st, ws + oq + 3 * hs, hs, mx + 1);
    LAUNCH_CHECK();
    for (int i = 0; i < 3; ++i)
        if (tdx::launch_h3_split_rows(ws + oq + i * hs, QK, qkP + (size_t)i * hs * 4, qks + (long)i * B * Sp, (long)B * Sp, QK, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    {
        const long K = (long)B * Sp;
        long n = K * (QK / 8);
        hipLaunchKernelGGL(tdx::h3_split_kmajor_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws + oq + 3 * hs, (long)QK,
                           qkP + (size_t)3 * hs * 4, K, QK, 1.0f, mx + 1, stt + 1, 0, 0);
        LAUNCH_CHECK();
        n = K * (2 * E / 8);
        hipLaunchKernelGGL(tdx::h3_split_kmajor_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws + ovu, 2L * E, vuP, K, 2 * E, 1.0f,
                           mx, stt, S, Sp);
        LAUNCH_CHECK();
    }
    return attention_core_h3(qkP, qks, vuP, ws + ovu, stt, B, S, E, splits, kchunk, ws + oA, AbufP, Asc, ws + oslab, ws + okvu, KvuP, kvus,
                             nullptr, att_v, att_u, st);
}

size_t tdx_dilated_dense_net_workspace_bytes(int B, int S) {
    if (B < 1 || S < 1) return 0;
    const size_t M = (size_t)B * S;
    const int ts = ddn_ts(B, S);
    const int nchunk1 = (S + ts - 1) / ts, nchunk2 = 2 * ((S + 2 * ts - 1) / (2 * ts));
    const int nc = nchunk1 > nchunk2 ? nchunk1 : nchunk2;
    return (al(M * 256) * 2 + al((size_t)B * 512) * 2 + al(39 * 256) + al(39 * 512) + al((size_t)B * nc * 256 * 2 * 2)) * sizeof(float);
}

// weights arrive in the reference's native layouts here (w1[256][39], w2[256][2][39]); the
// tap-major transposition is done on the device side of this test hook by tiny kernels.
__global__ void transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wT, int Cc, int inner, int k) {
    // w[c][i][t] -> wT[t][i][c]
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cc * inner * k) return;
    const int t = idx % k, i = (idx / k) % inner, c = idx / (k * inner);
    wT[((long)t * inner + i) * Cc + c] = w[idx];
}

int tdx_dilated_dense_net(const float* p, int B, int S, const float* w1, const float* w2, const float* in_g, const float* in_b,
                          const float* prelu, float* out, void* ws_, size_t ws_bytes, void* stream) {
    if (!p || !w1 || !w2 || !in_g || !in_b || !prelu || !out || !ws_) return tdx::fail(TDX_E_INVALID, "tdx_dilated_dense_net: null argument");
    if (ws_bytes < tdx_dilated_dense_net_workspace_bytes(B, S)) return tdx::fail(TDX_E_WORKSPACE, "tdx_dilated_dense_net: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)ws_;
    const size_t M = (size_t)B * S;
    float* c1 = ws; float* c2 = c1 + al(M * 256);
    float* stat1 = c2 + al(M * 256); float* stat2 = stat1 + al((size_t)B * 512);
    float* w1T = stat2 + al((size_t)B * 512); float* w2T = w1T + al(39 * 256);
    double* part = (double*)(w2T + al(39 * 512));
    hipLaunchKernelGGL(transpose_w_kernel, dim3((256 * 39 + 255) / 256), dim3(256), 0, st, w1, w1T, 256, 1, 39);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(transpose_w_kernel, dim3((256 * 78 + 255) / 256), dim3(256), 0, st, w2, w2T, 256, 2, 39);
    LAUNCH_CHECK();
    TRY(ddn_core(p, B, S, w1T, w2T, in_g, in_b, prelu, c1, c2, stat1, stat2, part, st));
    hipLaunchKernelGGL(ddn_out_kernel, dim3((unsigned)((M * 256 + 255) / 256)), dim3(256), 0, st, c2, stat2, in_g + 256, in_b + 256,
                       prelu + 256, out, (long)M, S);
    LAUNCH_CHECK();
    return TDX_OK;
}

int tdx_linear(const float* a, const float* w, const float* bias, int M, int N, int K, float* c, void* stream) {
    if (!a || !w || !c) return tdx::fail(TDX_E_INVALID, "tdx_linear: null argument");
    if (M < 1 || N % 128 || K % 32 || N < 128 || K < 32) return tdx::fail(TDX_E_INVALID, "tdx_linear: need N%128==0, K%32==0");
    if (N % 256 == 0) return linear_gemm(a, K, w, M, N, K, EpiBias{bias, c, N}, (hipStream_t)stream);     // split-bf16 x6 core
    return linear_gemm_f32(a, K, w, M, N, K, EpiBias{bias, c, N}, (hipStream_t)stream);                   // fp32-MFMA core
}

int tdx_cosine_scores(const float* emb, const float* ref, int N, int D, float* scores, void* stream) {
    if (!emb || !ref || !scores || N < 0 || D < 1) return tdx::fail(TDX_E_INVALID, "tdx_cosine_scores: bad argument");
    if (N == 0) return TDX_OK;
    hipLaunchKernelGGL(cosine_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, ref, N, D, scores);
    LAUNCH_CHECK();
    return TDX_OK;
}

}  // extern "C"
