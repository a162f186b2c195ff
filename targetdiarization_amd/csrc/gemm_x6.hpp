// gemm_x6.hpp — fp32-accurate NT GEMM on the bf16 MFMA pipe ("split-bf16 x6").
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),  A,W fp32 in HBM, fp32 accumulate.
//
// Each fp32 operand is split on the fly into three bf16 planes  x = x_h + x_m + x_l
// (x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m): 3 x 8 = 24 mantissa bits, and
// bf16 keeps fp32's exponent range, so no scaling is needed) and the product is evaluated as
// the six MFMA passes  hh + hm + mh + mm + hl + lh  (the dropped ml/lm/ll terms are <= 2^-24
// relative, i.e. at the fp32 rounding level).  v_mfma_f32_32x32x16_bf16 runs at 16x the
// fp32-input MFMA rate, so six passes are 2.67x faster than v_mfma_f32_32x32x2_f32 at the
// same (measured, tests/test_gpu_mossformer2.py::test_linear*) accuracy.  SURVEY.md §7.2 names
// this scheme as the alternative to the fp32 MFMA.
//
// Tiling: 256 x 256 x 16 block tile, 8 waves as 2(M) x 4(N), each wave 128 x 64 = 4 x 2 MFMA
// tiles (128 accumulator VGPRs), one block per CU.  The big tile is what keeps the global
// load rate at ~11 B/clk/CU; the fp32 kernel's 128 x 128 tile would need 21 B/clk/CU here.
// LDS: 2 buffers x (A,B) x 3 planes x [256 rows][16 bf16], row pitch 48 B (conflict-free
// ds_read_b128: 48 r mod 256 is distinct for 16 consecutive rows) = 144 KB.
// Pipeline per k-tile (one barrier): MFMA phase on buffer t  ||  split + ds_write of tile
// t+1 (held in registers since the previous phase) into buffer t^1  ->  global loads of tile
// t+2 into the freed registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hpp"

namespace tdx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int X6_BM = 256, X6_BN = 256, X6_BK = 16, X6_THREADS = 512;
constexpr int X6_PITCH = 48;                 // bytes per LDS row: 16 bf16 + 16 B pad
constexpr int X6_PLANE = 256 * X6_PITCH;     // 12 288 B
constexpr int X6_OPER = 3 * X6_PLANE;        // hi, mid, lo planes of one operand tile
constexpr int X6_BUF = 2 * X6_OPER;          // A then B
constexpr int X6_LDS = 2 * X6_BUF;           // double buffered: 147 456 B

__device__ __forceinline__ void x6_split_store(unsigned char* base, f32x4 x, bool ok) {
    bf16x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = ok ? x[i] : 0.f;
        const __bf16 hi = (__bf16)v;
        const float r1 = v - (float)hi;
        const __bf16 mi = (__bf16)r1;
        const float r2 = r1 - (float)mi;
        h[i] = hi; m[i] = mi; l[i] = (__bf16)r2;
    }
    *reinterpret_cast<bf16x4*>(base) = h;
    *reinterpret_cast<bf16x4*>(base + X6_PLANE) = m;
    *reinterpret_cast<bf16x4*>(base + 2 * X6_PLANE) = l;
}

template <bool SHIFT, class Epi>
__global__ __launch_bounds__(X6_THREADS, 2) void gemm_x6_kernel(GemmArgs g, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    int bm, bn;
    {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int per = g.mp * g.gw, ngf = g.tiles_n / g.gw;
        const int p = i / per;
        int lm, n;
        if (p < ngf) { const int j = i - p * per; lm = j / g.gw; n = p * g.gw + (j - lm * g.gw); }
        else { const int rem = g.tiles_n - ngf * g.gw; const int j = i - ngf * per; lm = j / rem; n = ngf * g.gw + (j - lm * rem); }
        bm = x * g.mp + lm; bn = n;
        if (bm >= g.tiles_m) return;
    }
    const int m0 = bm * X6_BM, n0 = bn * X6_BN;
    const float* __restrict__ Ag = g.seg[0].A;
    const float* __restrict__ Bg = g.seg[0].B;
    const long lda = g.seg[0].lda, ldb = g.seg[0].ldb;
    const int nkt = g.seg[0].K / X6_BK;

    // ---- staging map: 256 rows x 4 float4 per operand tile, 2 rows per thread
    const int c4 = tid & 3, r0 = tid >> 2, r1 = r0 + 128;
    const int mA0 = m0 + r0, mA1 = m0 + r1;
    const bool okA0 = mA0 < g.M, okA1 = mA1 < g.M;
    const float* pa0 = Ag + (long)min(mA0, g.M - 1) * lda + c4 * 4;
    const float* pa1 = Ag + (long)min(mA1, g.M - 1) * lda + c4 * 4;
    const float *ps0 = pa0, *ps1 = pa1;
    bool oks0 = okA0, oks1 = okA1;
    if constexpr (SHIFT) {
        oks0 = okA0 && (mA0 % g.shift_S != 0); oks1 = okA1 && (mA1 % g.shift_S != 0);
        ps0 = Ag + (long)max(min(mA0, g.M - 1) - 1, 0) * lda + c4 * 4;
        ps1 = Ag + (long)max(min(mA1, g.M - 1) - 1, 0) * lda + c4 * 4;
    }
    const float* pb0 = Bg + (long)(n0 + r0) * ldb + c4 * 4;
    const float* pb1 = Bg + (long)(n0 + r1) * ldb + c4 * 4;
    const int st0 = r0 * X6_PITCH + c4 * 8, st1 = r1 * X6_PITCH + c4 * 8;     // byte offsets inside a plane

    f32x4 ra0, ra1, rb0, rb1;
    bool va0 = okA0, va1 = okA1;
#define X6_LOAD(kt)                                                                 \
    {                                                                               \
        const int k0_ = (kt) * X6_BK;                                               \
        if constexpr (SHIFT) {                                                      \
            const bool sh = k0_ < g.shift_k;                                        \
            ra0 = ldg4((sh ? ps0 : pa0) + k0_); va0 = sh ? oks0 : okA0;             \
            ra1 = ldg4((sh ? ps1 : pa1) + k0_); va1 = sh ? oks1 : okA1;             \
        } else { ra0 = ldg4(pa0 + k0_); ra1 = ldg4(pa1 + k0_); }                    \
        rb0 = ldg4(pb0 + k0_); rb1 = ldg4(pb1 + k0_);                               \
    }
#define X6_STORE(buf)                                                               \
    {                                                                               \
        unsigned char* b_ = lds + (buf) * X6_BUF;                                   \
        x6_split_store(b_ + st0, ra0, va0);                                         \
        x6_split_store(b_ + st1, ra1, va1);                                         \
        x6_split_store(b_ + X6_OPER + st0, rb0, true);                              \
        x6_split_store(b_ + X6_OPER + st1, rb1, true);                              \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes inside a plane)
    const int fa = (wm * 128 + l31) * X6_PITCH + h * 16;      // + tm * 32 rows
    const int fb = (wn * 64 + l31) * X6_PITCH + h * 16;       // + tn * 32 rows

    X6_LOAD(0)
    X6_STORE(0)
    if (nkt > 1) X6_LOAD(1)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* cur = lds + (kt & 1) * X6_BUF;
        const unsigned char* Ap = cur + fa;
        const unsigned char* Bp = cur + X6_OPER + fb;
        bf16x8 bh[2], bmid[2], bl[2], a[4];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            bh[tn] = *reinterpret_cast<const bf16x8*>(Bp + tn * 32 * X6_PITCH);
            bmid[tn] = *reinterpret_cast<const bf16x8*>(Bp + X6_PLANE + tn * 32 * X6_PITCH);
            bl[tn] = *reinterpret_cast<const bf16x8*>(Bp + 2 * X6_PLANE + tn * 32 * X6_PITCH);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + tm * 32 * X6_PITCH);
        // next tile: registers -> three bf16 planes in the other buffer, then refill the registers
        if (kt + 1 < nkt) {
            X6_STORE((kt + 1) & 1)
            if (kt + 2 < nkt) X6_LOAD(kt + 2)
        }
        // hi x {hi, mid, lo}
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bmid[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
        // mid x {hi, mid}
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + X6_PLANE + tm * 32 * X6_PITCH);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bmid[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
        // lo x hi
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + 2 * X6_PLANE + tm * 32 * X6_PITCH);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        __syncthreads();
    }
#undef X6_LOAD
#undef X6_STORE

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + l31;
        if (n >= g.N) continue;
        const auto cc = epi.col(0, n);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) epi.store(0, m, n, acc[tm][tn][r], epi.row(0, m), cc);
            }
        }
    }
}

// N may be any multiple of 64 <= ldw rows available: W must be readable for rows up to the next
// multiple of 256 (pad the weight buffer); columns >= N are not stored.
template <bool SHIFT, class Epi>
inline hipError_t launch_gemm_x6(GemmArgs g, Epi epi, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<SHIFT, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS);
        attr_set = true;
    }
    g.tiles_m = (g.M + X6_BM - 1) / X6_BM;
    g.tiles_n = (g.N + X6_BN - 1) / X6_BN;
    g.map_mode = 0;
    g.mp = (g.tiles_m + 7) / 8;
    long gw = (1536L * 1024) / (256L * g.seg[0].K * 4);
    if (gw < 1) gw = 1;
    if (gw > g.tiles_n) gw = g.tiles_n;
    g.gw = (int)gw;
    dim3 grid(8 * g.mp * g.tiles_n, 1, 1);
    hipLaunchKernelGGL((gemm_x6_kernel<SHIFT, Epi>), grid, dim3(X6_THREADS), X6_LDS, st, g, epi);
    return hipGetLastError();
}

}  // namespace tdx
