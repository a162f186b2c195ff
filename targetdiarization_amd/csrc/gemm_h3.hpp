// gemm_h3.hpp — fp32-accurate GEMM on the f16 MFMA pipe in THREE passes ("split-f16 x3").
//
//   C[z][m][n] = epilogue( sa[m] * sb[n] * sum_k A'[z][m][k] * B'[z][n][k] )
//
// Operands arrive PRE-SPLIT ("planes", produced once by the kernel that owns the row — see
// h3_split_rows_kernel and the fused producers in mf2_kernels.hpp; weights are split once
// when the model is created).  A row x[0..K) with max |x| = mu is stored as
//     e  = floor(log2 mu)                       per-row exponent (a power of two: exact)
//     xs = x * 2^(14-e)                         row max lands in [2^14, 2^15) < f16 max 65504
//     hi = f16(xs),  lo = f16(xs - hi)          2 x 11 = 22 mantissa bits for every element
//                                               within 2^-18 of the row max; smaller elements
//                                               keep an ABSOLUTE error <= 2^-40 * mu
//     scale = 2^(e-14)                          fp32, multiplied back in the epilogue
// and the product is  hi*lo' + lo*hi' + hi*hi'  (the dropped lo*lo' term is 2^-22 relative).
// Against fp64 this is MORE accurate than an fp32 GEMM (7.5e-8 vs 2.9e-7 rel-L2 at K=512,
// tools/h3_test.py), the dynamic-range loss of f16 being removed by the per-row exponent.
// Three passes of v_mfma_f32_32x32x16_f16 instead of six of the bf16 split (gemm_x6.hpp) or
// sixteen fp32-MFMA equivalents: the MFMA ceiling is 2500/3 = 833 TFLOP/s of fp32-grade work.
//
// Memory format of a split matrix X[R][K] (K a multiple of 32):
//     planes: [R][K/8][2][8] f16   (16 B of hi then 16 B of lo per 8 k) = 4 K bytes per row,
//             i.e. exactly the footprint of the fp32 row it replaces
//     scale : [R] fp32
// One K-step (32 k) of a row is one contiguous 128-B line.
//
// Tiling: 256 x 256 x 16 block tile, 8 waves as 2(M) x 4(N), wave tile 128 x 64 = 4 x 2 MFMA
// tiles (128 accumulator VGPRs), 1 block per CU.  No VALU work in the main loop: operands go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 16 rows x 64 B per wave
// instruction, 4 per wave per k-tile) into a ring of FOUR 32 KB stages; the DMA of tile t+3 is
// issued during the MFMAs of tile t and waited for with a COUNTED vmcnt (never 0 in steady
// state) before the raw s_barrier that opens stage t+2.  Fragments of tile t+1 are read from
// LDS during the MFMAs of tile t (A fragments into the registers the finished row of MFMA
// tiles just released, B fragments into a second register set), so a stage is
//     [vmcnt(4); s_barrier]  4 x { 6 MFMA ; 2-3 ds_read_b128 ; 1 LDS-DMA }
// with nothing exposed but the barrier itself.
// LDS image: row pitch 64 B (16 k x 2 planes x 2 B), the 16-B slot q of row r is stored at
// slot q ^ ((r>>2)&3); the swizzle is applied on the per-lane SOURCE address (the LDS-DMA
// destination is lane-linear) and on the fragment read.  With it the four 16-lane groups of a
// ds_read_b128 (rows {0-3,12-15,20-27}, {4-11,16-19,28-31} of a 32-row fragment) touch 16
// distinct 16-B bank slots: conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hpp"

namespace tdx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H3_BM = 256, H3_BN = 256, H3_BK = 16, H3_THREADS = 512;
constexpr int H3_ROWB = 64;                   // bytes per LDS row (16 k x 2 planes x 2 B)
constexpr int H3_OPER = 256 * H3_ROWB;        // 16 KB per operand tile
constexpr int H3_STAGE = 2 * H3_OPER;         // A then B
constexpr int H3_NBUF = 4;
constexpr int H3_LDS = H3_NBUF * H3_STAGE;    // 128 KB

struct H3Seg {
    const unsigned char* A;      // planes of the A rows  (row pitch lda bytes)
    const unsigned char* B;      // planes of the B rows ([N][K]: one row per output column)
    const float* sa;             // [M] row scales of A
    const float* sb;             // [N] row scales of B
    long lda, ldb;               // bytes
    // batch z -> z1 = z / zdiv, z2 = z % zdiv ; pointer += z1*stride + z2*stride2
    long strideA, strideB, strideA2, strideB2;          // bytes
    long strideSA, strideSB, strideSA2, strideSB2;      // floats
    int K;                       // multiple of 16
    int zdiv;
};

struct H3Args {
    H3Seg seg[2];
    int nseg;
    int M, N;                    // valid rows / columns (PAIRED: N = number of pair columns, multiple of 128)
    int tiles_m, tiles_n;
    int map_mode, mp, gw, batches;
    int pair_off;
};

inline H3Seg h3_seg(const void* A, const float* sa, long lda, const void* B, const float* sb, long ldb, int K) {
    H3Seg s{};
    s.A = (const unsigned char*)A; s.B = (const unsigned char*)B; s.sa = sa; s.sb = sb; s.lda = lda; s.ldb = ldb; s.K = K; s.zdiv = 1;
    return s;
}

__device__ __forceinline__ void h3_glds16(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// one pipeline stage: MFMAs of the tile whose fragments are in (ah, al, bh, bl); with NEXT the
// fragments of the following tile are read from stage buffer `nxt`; with ISSUE the wave's four
// LDS-DMA pieces of the tile three ahead are issued (src = gp[j] + goff, dst = dmad + j KiB).
template <bool FULLN, bool NEXT, bool ISSUE, int PH>
__device__ __forceinline__ void h3_stage(f32x16 (&acc)[4][2], f16x8 (&ah)[4], f16x8 (&al)[4], f16x8 (&bh)[2], f16x8 (&bl)[2],
                                         const unsigned char* nxt, int fa, int fb, int fo0, int fo1,
                                         const unsigned char* const (&gp)[4], long goff, unsigned char* dmad) {
    f16x8 nbh[2], nbl[2];
    if constexpr (NEXT) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            nbh[tn] = *reinterpret_cast<const f16x8*>(nxt + fb + tn * 128 * H3_ROWB + fo0);
            nbl[tn] = *reinterpret_cast<const f16x8*>(nxt + fb + tn * 128 * H3_ROWB + fo1);
        }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < (FULLN ? 2 : 1); ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            if constexpr (ISSUE) {
                if ((tm >> 1) == PH) h3_glds16(gp[(tm & 1) * 2 + tn] + goff, dmad + ((tm & 1) * 2 + tn) * 1024);
            }
        }
        if constexpr (ISSUE && !FULLN) {
            if ((tm >> 1) == PH) h3_glds16(gp[(tm & 1) * 2 + 1] + goff, dmad + ((tm & 1) * 2 + 1) * 1024);
        }
        if constexpr (NEXT) {
            ah[tm] = *reinterpret_cast<const f16x8*>(nxt + fa + tm * 32 * H3_ROWB + fo0);
            al[tm] = *reinterpret_cast<const f16x8*>(nxt + fa + tm * 32 * H3_ROWB + fo1);
        }
    }
    if constexpr (NEXT) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) { bh[tn] = nbh[tn]; bl[tn] = nbl[tn]; }
    }
    // pin the issue order (the scheduler would otherwise cluster the DMA and the reads at the end
    // of the stage, where both waves of a SIMD would stall on them together)
    if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const bool dma = ISSUE && (tm >> 1) == PH;
        if constexpr (FULLN) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
        }
        if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

#define H3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define H3_BARRIER()                                   \
    do {                                               \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)

// VARIANT != 0: timing-only diagnostics (wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads,
// 3 neither, 4 no barrier
template <bool PAIRED, class Epi, int VARIANT = 0>
__global__ __launch_bounds__(H3_THREADS, 2) void gemm_h3_kernel(H3Args g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    int z, bm, bn;
    {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        if (g.map_mode == 0) {
            z = blockIdx.y;
            const int per = g.mp * g.gw, ngf = g.tiles_n / g.gw;
            const int p = i / per;
            int lm, n;
            if (p < ngf) { const int j = i - p * per; lm = j / g.gw; n = p * g.gw + (j - lm * g.gw); }
            else { const int rem = g.tiles_n - ngf * g.gw; const int j = i - ngf * per; lm = j / rem; n = ngf * g.gw + (j - lm * rem); }
            bm = x * g.mp + lm; bn = n;
            if (bm >= g.tiles_m) return;
        } else {
            const int tpb = g.tiles_m * g.tiles_n;
            const int zb = i / tpb, tt = i - zb * tpb;
            z = zb * 8 + x;
            if (z >= g.batches) return;
            bm = tt / g.tiles_n; bn = tt - bm * g.tiles_n;
        }
    }
    const int m0 = bm * H3_BM;
    const int n0 = PAIRED ? bn * (H3_BN / 2) : bn * H3_BN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging role: waves 0-3 fetch the A tile, waves 4-7 the B tile; piece j of a wave = LDS rows
    // (wave&3)*64 + 16j .. +15, lane -> row +(lane>>2), LDS slot lane&3, source slot (lane&3)^((row>>2)&3)
    const bool stB = wave >= 4;
    const int srow = (wave & 3) * 64 + (lane >> 2);
    const int sdst = (stB ? H3_OPER : 0) + (wave & 3) * 64 * H3_ROWB;
    // fragment reads: per-lane slot offsets of the hi / lo plane (k = 8h .. 8h+7 of the 16-k tile)
    const int fl = (l31 >> 2) & 3;
    const int fo0 = ((2 * h) ^ fl) << 4, fo1 = ((2 * h + 1) ^ fl) << 4;
    const int fa = (wm * 128 + l31) * H3_ROWB;              // + tm * 32 rows
    const int fb = H3_OPER + (wn * 32 + l31) * H3_ROWB;     // + tn * 128 rows
    const bool full_n = PAIRED || (n0 + 128 < g.N);

    const int z1 = z / g.seg[0].zdiv, z2 = z - z1 * g.seg[0].zdiv;
    const int nkt = g.seg[0].K / H3_BK;
    const unsigned char* gp[4];
    if (!stB) {
        const unsigned char* Ag = g.seg[0].A + (long)z1 * g.seg[0].strideA + (long)z2 * g.seg[0].strideA2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = srow + 16 * j;
            const int q = (lane & 3) ^ ((r >> 2) & 3);
            gp[j] = Ag + (long)min(m0 + r, g.M - 1) * g.seg[0].lda + q * 16;
        }
    } else {
        const unsigned char* Bg = g.seg[0].B + (long)z1 * g.seg[0].strideB + (long)z2 * g.seg[0].strideB2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = srow + 16 * j;
            const int q = (lane & 3) ^ ((r >> 2) & 3);
            const int n = PAIRED ? (r < 128 ? n0 + r : g.pair_off + n0 + r - 128) : min(n0 + r, g.N - 1);
            gp[j] = Bg + (long)n * g.seg[0].ldb + q * 16;
        }
    }

    // ---- prologue: tiles 0..2 in flight, fragments of tile 0 in registers
#pragma unroll
    for (int p = 0; p < 3; ++p)
        if (p < nkt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + p * H3_ROWB, lds + p * H3_STAGE + sdst + j * 1024);
        }
    if (nkt >= 3) H3_WAIT_VM(8); else if (nkt == 2) H3_WAIT_VM(4); else H3_WAIT_VM(0);
    H3_BARRIER();
    f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        bh[tn] = *reinterpret_cast<const f16x8*>(lds + fb + tn * 128 * H3_ROWB + fo0);
        bl[tn] = *reinterpret_cast<const f16x8*>(lds + fb + tn * 128 * H3_ROWB + fo1);
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        ah[tm] = *reinterpret_cast<const f16x8*>(lds + fa + tm * 32 * H3_ROWB + fo0);
        al[tm] = *reinterpret_cast<const f16x8*>(lds + fa + tm * 32 * H3_ROWB + fo1);
    }

    // ---- stages.  At the top of stage t the wave's own pieces of tile t+1 must have landed
    // (tile t+2 may stay in flight: vmcnt(4)); the barrier then makes tile t+1 readable for
    // everyone and proves that buffer (t+3)&3 — read last during stage t-2 — is free.
#define H3_RUN(FULLN, PH)                                                                                              \
    {                                                                                                                  \
        int t = 0;                                                                                                     \
        for (; t + 3 < nkt; ++t) {                                                                                     \
            H3_WAIT_VM(4);                                                                                             \
            if (VARIANT != 4) H3_BARRIER();                                                                            \
            h3_stage<FULLN, VARIANT != 2 && VARIANT != 3, VARIANT != 1 && VARIANT != 3, PH>(                           \
                acc, ah, al, bh, bl, lds + ((t + 1) & 3) * H3_STAGE, fa, fb, fo0, fo1, gp, (long)(t + 3) * H3_ROWB,    \
                lds + ((t + 3) & 3) * H3_STAGE + sdst);                                                                \
        }                                                                                                              \
        for (; t + 1 < nkt; ++t) {                                                                                     \
            if (t + 2 < nkt) H3_WAIT_VM(4); else H3_WAIT_VM(0);                                                        \
            H3_BARRIER();                                                                                              \
            h3_stage<FULLN, true, false, PH>(acc, ah, al, bh, bl, lds + ((t + 1) & 3) * H3_STAGE, fa, fb, fo0, fo1, gp, 0, lds); \
        }                                                                                                              \
        h3_stage<FULLN, false, false, PH>(acc, ah, al, bh, bl, lds, fa, fb, fo0, fo1, gp, 0, lds);                     \
    }
    // the two waves of a SIMD (w and w+4) issue their DMA in different halves of the stage
    if (full_n) { if (stB) H3_RUN(true, 1) else H3_RUN(true, 0) }
    else { if (stB) H3_RUN(false, 1) else H3_RUN(false, 0) }
#undef H3_RUN

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h; scales of the LAST segment
    const float* sa = g.seg[0].sa + (long)z1 * g.seg[0].strideSA + (long)z2 * g.seg[0].strideSA2;
    const float* sb = g.seg[0].sb + (long)z1 * g.seg[0].strideSB + (long)z2 * g.seg[0].strideSB2;
    if constexpr (PAIRED) {
        const int c = n0 + wn * 32 + l31;
        const auto cc = epi.col(z, c);
        const float sc0 = sb[c], sc1 = sb[g.pair_off + c];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            decltype(epi.row(0, 0)) rw[16];
            float sr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1);
                rw[r] = epi.row(z, m);
                sr[r] = sa[m];
            }
            if constexpr (epi_has_aux<Epi>::value) {
                decltype(epi.aux(0, 0, 0, rw[0])) ax[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[r] = epi.aux(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1), c, rw[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < g.M) epi.store2(z, m, c, acc[tm][0][r] * (sr[r] * sc0), acc[tm][1][r] * (sr[r] * sc1), rw[r], cc, ax[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < g.M) epi.store2(z, m, c, acc[tm][0][r] * (sr[r] * sc0), acc[tm][1][r] * (sr[r] * sc1), rw[r], cc);
                }
            }
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + tn * 128 + wn * 32 + l31;
            if (n >= g.N) continue;
            const auto cc = epi.col(z, n);
            const float sc = sb[n];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                decltype(epi.row(0, 0)) rw[16];
                float sr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1);
                    rw[r] = epi.row(z, m);
                    sr[r] = sa[m];
                }
                if constexpr (epi_has_aux<Epi>::value) {
                    decltype(epi.aux(0, 0, 0, rw[0])) ax[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ax[r] = epi.aux(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1), n, rw[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (m < g.M) epi.store(z, m, n, acc[tm][tn][r] * (sr[r] * sc), rw[r], cc, ax[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (m < g.M) epi.store(z, m, n, acc[tm][tn][r] * (sr[r] * sc), rw[r], cc);
                    }
                }
            }
        }
    }
}

template <bool PAIRED, class Epi, int VARIANT = 0>
inline hipError_t launch_gemm_h3(H3Args g, int batches, Epi epi, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<PAIRED, Epi, VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS);
        attr_set = true;
    }
    g.tiles_m = (g.M + H3_BM - 1) / H3_BM;
    g.tiles_n = PAIRED ? g.N / (H3_BN / 2) : (g.N + H3_BN - 1) / H3_BN;
    g.batches = batches;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        long ktot = 0;
        for (int i = 0; i < g.nseg; ++i) ktot += g.seg[i].K;
        long gw = (1536L * 1024) / (256L * ktot * 4);      // weight slice of a group stays in the XCD's L2
        if (gw < 1) gw = 1;
        if (gw > g.tiles_n) gw = g.tiles_n;
        g.gw = (int)gw;
        grid = dim3(8 * g.mp * g.tiles_n, batches, 1);
    } else {
        g.map_mode = 1;
        g.mp = 0; g.gw = 1;
        grid = dim3(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    }
    hipLaunchKernelGGL((gemm_h3_kernel<PAIRED, Epi, VARIANT>), grid, dim3(H3_THREADS), H3_LDS, st, g, epi);
    return hipGetLastError();
}

// ---- split producers ---------------------------------------------------------------------------

// exponent-aligned scale of a row whose max |x| is mu: returns s = 2^(14-e) and writes inv = 2^(e-14)
__device__ __forceinline__ float h3_row_scale(float mu, float& inv) {
    int e = (int)((__float_as_uint(mu) >> 23) & 0xff) - 127;
    if (mu == 0.f) e = 14;
    e = max(-100, min(100, e));
    inv = __uint_as_float((unsigned)(127 - 14 + e) << 23);
    return __uint_as_float((unsigned)(127 + 14 - e) << 23);
}

// 8 consecutive k -> 16 B of hi + 16 B of lo
__device__ __forceinline__ void h3_store_chunk(unsigned char* dst, const float* x, float s) {
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float xs = x[i] * s;
        const _Float16 a = (_Float16)xs;
        hi[i] = a;
        lo[i] = (_Float16)(xs - (float)a);
    }
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 16) = lo;
}

// fused producers: a wave owns one row and lane l holds the 4 consecutive values of quad q
// (k = 4q .. 4q+3, q = i*64 + l).  Lanes 2j and 2j+1 exchange their quads; the even lane stores
// the hi half (16 B) of chunk j, the odd lane the lo half: consecutive lanes write consecutive
// 16 B.  All 64 lanes must be active.
__device__ __forceinline__ void h3_emit4(unsigned char* prow, int q, float4 v, float s) {
    float4 p;
    p.x = __shfl_xor(v.x, 1, 64); p.y = __shfl_xor(v.y, 1, 64); p.z = __shfl_xor(v.z, 1, 64); p.w = __shfl_xor(v.w, 1, 64);
    const bool odd = q & 1;
    const float4 a = odd ? p : v, b = odd ? v : p;
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float xs = x[i] * s;
        const _Float16 hi = (_Float16)xs;
        out[i] = odd ? (_Float16)(xs - (float)hi) : hi;
    }
    *reinterpret_cast<f16x8*>(prow + (q >> 1) * 32 + (odd ? 16 : 0)) = out;
}
__device__ __forceinline__ float h3_wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float h3_absmax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// generic producer: fp32 rows [R][K] (pitch ld floats, K % 8 == 0, K <= 2048) -> planes (pitch 4*Kp bytes,
// Kp = K rounded up to 16, the tail zero-filled) + scale.  One wave per row.
template <int UNUSED = 0>      // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void h3_split_rows_kernel(const float* __restrict__ x, long ld, unsigned char* __restrict__ planes,
                                                           float* __restrict__ scale, long R, int K, int Kp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + row * ld;
    float v[4][8];
    float mu = 0.f;
    const int nch = Kp / 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
        if (ch * 8 < K) {
            const f32x4 a = ldg4(xr + ch * 8), b = ldg4(xr + ch * 8 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[c][i] = a[i]; v[c][4 + i] = b[i]; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) mu = fmaxf(mu, fabsf(v[c][i]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mu = fmaxf(mu, __shfl_xor(mu, o, 64));
    float inv;
    const float s = h3_row_scale(mu, inv);
    unsigned char* pr = planes + row * (long)Kp * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) h3_store_chunk(pr + ch * 32, v[c], s);
    }
    if (lane == 0) scale[row] = inv;
}

inline hipError_t launch_h3_split_rows(const float* x, long ld, void* planes, float* scale, long R, int K, hipStream_t st) {
    const int Kp = (K + 15) / 16 * 16;
    hipLaunchKernelGGL(h3_split_rows_kernel<0>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, ld, (unsigned char*)planes, scale, R, K, Kp);
    return hipGetLastError();
}

}  // namespace tdx
