// gemm_h3a.hpp — the gated attention launch of FLASH ([A | lin_q] x [V|U ; Kvu], mossformer_block.py:269-294) on the split-f16
// x3 core as a 128 x (128 + 128) PAIRED tile with FOUR waves and TWO blocks per CU.
//
// Why: on the 256 x (128+128) / 8-wave kernel (gemm_h3.hpp) this launch spends ~45 % of its time in the gate + planes-out
// epilogue (K = 384 only: 24 k-tiles against ~60 VALU instructions per output element), and that kernel owns its CU alone,
// so the MFMA pipe idles through every epilogue and the VALU through every main loop.  Two independent 4-wave blocks per CU
// (72 KB ring each: three 24 KB stages) run one block's epilogue under the other block's MFMAs.  The tile keeps the full
// 128-column pair segment (the planes-out scale / sum of squares is per (row, 128-column segment), as to_out expects) and
// halves the rows instead: a 256-token group is two M tiles.  Wave tile 64 x (64 + 64) = 2 x 4 MFMA tiles (128 accumulator
// VGPRs, two waves per SIMD), LDS bytes read per MFMA as in the wide kernel; operand bytes fetched per flop are 1.5x.
//
// Modes: A row-major planes, B K-major planes (ds_read_b64_tr_b16 through inline asm, see h3_tr_read) and either
//   * TWOSEG + PAIRED + planes-out epilogue with aux() operands (EpiAttnGatePlOut concept): the attention launch, or
//   * one segment (optionally a split-K chunk: H3Seg::kchunk / ktotal), 256 consecutive columns per tile, store() epilogue:
//     lin_k^T [v|u] (M = 128: on the wide kernel half of the waves of a block have no rows and only feed the ring).
// Same k order, same MFMA shape, same rescale and the same per-element epilogue math as the wide kernel: bit-identical results.
#pragma once
#include "gemm_h3.hpp"

namespace tdx {

constexpr int H3A_THREADS = 256;
constexpr int H3A_A = 128 * H3_ROWB;           // 8 KB: 128 rows of row-major planes
constexpr int H3A_B = 16 * 1024;               // 16 KB: 16 k rows x [v block: hi 256 B | lo 256 B][u block: hi | lo]
constexpr int H3A_STAGE = H3A_A + H3A_B;       // 24 KB
constexpr int H3A_NBUF = 3;
constexpr int H3A_LDS = H3A_NBUF * H3A_STAGE;  // 72 KB
constexpr int H3A_EXTRA = 2048;
#ifndef H3A_PIN
#define H3A_PIN 0      // 1: sched_barrier after every MFMA group (reads interleaved as written): 28.6 vs 27.1 us per k loop — no better
#endif                // TWOSEG row factors 512 B | row scales 512 B | row() values 1 KB

template <bool TWOSEG, class Epi>
__global__ __launch_bounds__(H3A_THREADS, 2) void gemm_h3a_kernel(H3Args g, Epi epi) {
    constexpr bool GATE = epi_has_plout<Epi>::value;        // PAIRED tile + planes-out gate epilogue; otherwise 256 plain columns + store()
    static_assert(!GATE || (epi_has_aux<Epi>::value && epi_has_pairmul<Epi>::value && TWOSEG), "gate-style planes-out functor: the attention launch");
    static_assert(GATE || !TWOSEG, "store() epilogue: one segment");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    int z, bm, bn;
    {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;       // x = XCD: all tiles of a batch share one L2
        const int tpb = g.tiles_m * g.tiles_n;
        const int zb = i / tpb, tt = i - zb * tpb;
        z = zb * 8 + x;
        if (z >= g.batches) return;
        // column slices first.  (The other way round — the two M tiles of a column slice as neighbours, so that they stream the same
        // v|u rows at the same time and the second reader finds them in L2 — makes no difference: config 2 203.2 / 203.3 vs
        // 203.0 / 204.1 ms on one box, bit-identical, A/B by TDX_H3A_ORDER=1.)
        if (g.mp) { bn = tt / g.tiles_m; bm = tt - bn * g.tiles_m; }
        else { bm = tt / g.tiles_n; bn = tt - bm * g.tiles_n; }
    }
    const int m0 = bm * 128, n0 = bn * (GATE ? 128 : 256);       // (plain columns: the host sets pair_off = 128, the "u block" is columns n0 + 128 ..)
    H3_STAMP(0); H3_STAMP_HW();

    f32x16 acc[2][4];              // [tm][vu * 2 + cw]: rows wm*64 + tm*32 .., pair columns n0 + wn*64 + cw*32 + l31 of v (vu = 0) / u (1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage buffer (the images and swizzles of the wide kernel)
    const int fl = (l31 >> 2) & 3;
    const int a0 = (wm * 64 + l31) * H3_ROWB;
    const int pl0 = ((2 * h) ^ fl) << 4, pl1 = ((2 * h + 1) ^ fl) << 4;
    int bw[2];
    {
        const int q = (lane >> 2) & 3, p4 = lane & 3, nb = (lane >> 4) & 1;
#pragma unroll
        for (int cw = 0; cw < 2; ++cw) bw[cw] = H3A_A + (8 * h + q) * 1024 + (((2 * wn + cw) ^ q) << 6) + nb * 32 + p4 * 8;
    }

    // DMA: per stage every wave fetches two 1-KiB pieces of A (rows wave*32 + 16j ..) and four of B (k rows wave*4 + j)
    const int dstA = wave * 2048, dstB = H3A_A + wave * 4096;
    const unsigned char* gpA[2];
    const unsigned char* gpB[4];
    long stepB;
    auto setup = [&](const H3Seg& sg) {
        const int z1 = z / sg.zdiv, z2 = z - z1 * sg.zdiv;
        const unsigned char* Ag = sg.A + (long)z1 * sg.strideA + (long)z2 * sg.strideA2;
        const unsigned char* Bg = sg.B + (long)z1 * sg.strideB + (long)z2 * sg.strideB2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wave * 32 + (lane >> 2) + 16 * j;
            gpA[j] = Ag + (long)(m0 + r) * sg.lda + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
        const int ncol = (lane >> 5) ? g.pair_off + n0 : n0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = (lane & 15) ^ (j << 2);       // logical 16-B chunk (8 columns) of the 128-column block; (k & 3) = j
            gpB[j] = Bg + (long)(wave * 4 + j) * sg.ldb + (ncol / 128) * 512 + (ch >> 2) * 128 + ((lane >> 4) & 1) * 64 + (ch & 3) * 16;
        }
        stepB = 16 * sg.ldb;
    };

    const H3Seg& sg0 = g.seg[0];
    const H3Seg& s1 = g.seg[TWOSEG ? 1 : 0];          // the last segment: its scales are the epilogue's
    const int zl1 = z / s1.zdiv, zl2 = z - zl1 * s1.zdiv;
    using RowT = decltype(epi.row(0, 0));
    constexpr bool HAS_ROW = !std::is_empty<RowT>::value;
    float* rfl2 = reinterpret_cast<float*>(lds + H3A_LDS);                  // units of segment 0 -> units of segment 1, per row
    float* sal = reinterpret_cast<float*>(lds + H3A_LDS + 512);
    RowT* rwl = reinterpret_cast<RowT*>(lds + H3A_LDS + 1024);
    static_assert(sizeof(RowT) <= 8, "row() value: at most 8 bytes");
    if (tid < 128) {
        const int m = m0 + tid;
        const float* sa1 = s1.sa + (long)zl1 * s1.strideSA + (long)zl2 * s1.strideSA2;
        const float s1m = sa1[(long)m * s1.sa_mul];
        sal[tid] = s1m;
        if constexpr (HAS_ROW) rwl[tid] = epi.row(z, m);
        if constexpr (TWOSEG) {
            const float* sa0 = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)(z % sg0.zdiv) * sg0.strideSA2;
            rfl2[tid] = sa0[(long)m * sg0.sa_mul] / s1m;
        }
    }

    // k-tiles: segment 0 (a split-K chunk is clipped to the rows that exist: the host guarantees >= 3 tiles) [+ segment 1]
    const int nkt0 = (sg0.kchunk ? min(sg0.K, sg0.ktotal - (z % sg0.zdiv) * sg0.kchunk) : sg0.K) / H3_BK;
    const int nkt = TWOSEG ? nkt0 + s1.K / H3_BK : nkt0;
    float cf[4];                   // units of segment 0 -> units of segment 1, per column (loaded ahead of the ring's DMA)
    if constexpr (TWOSEG) {
        const float* sb0 = sg0.sb + (long)(z / sg0.zdiv) * sg0.strideSB + (long)(z % sg0.zdiv) * sg0.strideSB2;
        const float* sb1 = s1.sb + (long)zl1 * s1.strideSB + (long)zl2 * s1.strideSB2;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int c = ((tn >> 1) ? g.pair_off : 0) + n0 + wn * 64 + (tn & 1) * 32 + l31;
            cf[tn] = sb0[(long)c * sg0.sb_mul] / sb1[(long)c * s1.sb_mul];
        }
    }
    {
        setup(sg0);
        auto dma = [&](int ti) {      // prologue: this wave's six pieces of k-tile ti of segment 0
            unsigned char* st = lds + (ti % H3A_NBUF) * H3A_STAGE;
#pragma unroll
            for (int j = 0; j < 2; ++j) h3_glds16(gpA[j] + (long)ti * H3_ROWB, st + dstA + j * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) h3_glds16(gpB[j] + (long)ti * stepB, st + dstB + j * 1024);
        };
        dma(0); dma(1); dma(2);          // (K0 >= 48: host-checked)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
        H3_STAMP(1);
        f16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            ah[tm] = *reinterpret_cast<const f16x8*>(lds + a0 + tm * 2048 + pl0);
            al[tm] = *reinterpret_cast<const f16x8*>(lds + a0 + tm * 2048 + pl1);
        }
        // B fragments: transposing reads through inline asm (h3_tr_read: no compiler-inserted vmcnt(0)); raw[2*tn + plane]
        const unsigned lds0 = h3_lds_addr(lds);
        H3TrRaw raw[8];
        auto read_b = [&](unsigned sa, auto tn_c) {       // sa = LDS address of the stage buffer
            constexpr int tn = decltype(tn_c)::value;
            h3_tr_read<(tn >> 1) * 512>(raw[2 * tn], sa + bw[tn & 1]);
            h3_tr_read<(tn >> 1) * 512 + 256>(raw[2 * tn + 1], sa + bw[tn & 1]);
        };
        auto take_b = [&]() {                             // end of the window: wait, then the values may be used
            h3_tr_fence(raw);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) { bh[tn] = h3_tr_value(raw[2 * tn]); bl[tn] = h3_tr_value(raw[2 * tn + 1]); }
        };
        read_b(lds0, std::integral_constant<int, 0>{}); read_b(lds0, std::integral_constant<int, 1>{});
        read_b(lds0, std::integral_constant<int, 2>{}); read_b(lds0, std::integral_constant<int, 3>{});
        take_b();
        // one pipeline stage: at its top tile t+1 has landed for this wave (tile t+2 may stay in flight); the barrier makes it
        // readable for everyone and — every wave having retired its reads of tile t — frees buffer t % 3 for tile t+3
        int tile0 = 0;
        auto stage = [&](int t, auto next_c, auto issue_c) {
            constexpr bool NEXT = decltype(next_c)::value, ISSUE = decltype(issue_c)::value;
            if (NEXT && t + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            H3_BARRIER();
            const unsigned char* nst = lds + ((t + 1) % H3A_NBUF) * H3A_STAGE;
            const unsigned nsa = lds0 + ((t + 1) % H3A_NBUF) * H3A_STAGE;
            unsigned char* dst = lds + (t % H3A_NBUF) * H3A_STAGE;
            const long offA = (long)(t + 3 - tile0) * H3_ROWB, offB = (long)(t + 3 - tile0) * stepB;
            f16x8 nah[2], nal[2];
            if constexpr (NEXT) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    nah[tm] = *reinterpret_cast<const f16x8*>(nst + a0 + tm * 2048 + pl0);
                    nal[tm] = *reinterpret_cast<const f16x8*>(nst + a0 + tm * 2048 + pl1);
                }
            }
            auto group = [&](auto tn_c) {
                constexpr int tn = decltype(tn_c)::value;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    const int pj = tn * 2 + tm;                 // six of the eight slots carry one DMA piece each
                    if constexpr (ISSUE) {
                        if (pj < 2) h3_glds16(gpA[pj] + offA, dst + dstA + pj * 1024);
                        else if (pj < 6) h3_glds16(gpB[pj - 2] + offB, dst + dstB + (pj - 2) * 1024);
                    }
                }
                if constexpr (NEXT) read_b(nsa, tn_c);          // the next tile's fragment of this column tile (raw until take_b)
                if (H3A_PIN) __builtin_amdgcn_sched_barrier(0);    // keep the reads between the MFMA groups (the scheduler otherwise
                                                                // issues ~18 MFMAs first and all the reads at the end of the stage)
            };
            group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
            if constexpr (NEXT) {
                take_b();
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) { ah[tm] = nah[tm]; al[tm] = nal[tm]; }
            }
        };
        constexpr std::true_type Y{};
        constexpr std::false_type N{};
        int t = 0;
        if constexpr (TWOSEG) {
            for (; t + 3 < nkt0; ++t) stage(t, Y, Y);          // segment 0, its own tiles ahead
            setup(s1); tile0 = nkt0;
            for (; t < nkt0; ++t) stage(t, Y, Y);              // segment 0, the DMA already on segment 1 (K1 >= 48)
            // acc changes its scale domain between the segments (exact powers of two)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 rf = *reinterpret_cast<const f32x4*>(rfl2 + wm * 64 + tm * 32 + 8 * j + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) acc[tm][tn][4 * j + i] *= rf[i] * cf[tn];
                }
        }
        for (; t + 3 < nkt; ++t) stage(t, Y, Y);
        stage(t, Y, N); ++t;
        stage(t, Y, N); ++t;
        stage(t, N, N);
    }

    H3_STAMP(2);
    if (g.dbg & 1) {                    // (timing: no epilogue)
        float x = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) x += acc[0][0][r] + acc[1][1][r] + acc[0][2][r] + acc[1][3][r];
        if (x != 123.456f) return;
    }

    const float* sb = s1.sb + (long)zl1 * s1.strideSB + (long)zl2 * s1.strideSB2;
    const int sbm = s1.sb_mul;
    int lrow = wm * 64 + 4 * h, lcolb = wn * 64 + l31;
    asm volatile("" : "+v"(lrow), "+v"(lcolb));
    if constexpr (!GATE) {
        // ---- store() epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h of each 32 x 32 tile
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + (tn >> 1) * 128 + lcolb + (tn & 1) * 32;
            const auto cc = epi.col(z, n);
            const float sc = sb[(long)n * sbm];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = lrow + tm * 32 + (r & 3) + 8 * (r >> 2);
                    RowT rw{};
                    if constexpr (HAS_ROW) rw = rwl[lr];
                    epi.store(z, m0 + lr, n, acc[tm][tn][r] * (sal[lr] * sc), rw, cc);
                }
        }
    } else {
    // ---- epilogue (PLOUT, PAIRED, aux): phase 1 = gate values into the staging tile T [128][128] fp32 in the ring (32-float
    // blocks swapped on rows with bit 2 set), phase 2 = one tile row per half-wave: row max, scale, planes, scale / sum of squares
    float* const T = reinterpret_cast<float*>(lds);
    const H3PlOut po = epi.plout(z);
    __syncthreads();                         // every wave has retired its last fragment reads: the ring is free
    {
        decltype(epi.col(0, 0)) cc[2];
        float sc0[2], sc1[2];
#pragma unroll
        for (int cw = 0; cw < 2; ++cw) {
            const int c = n0 + lcolb + cw * 32;
            cc[cw] = epi.col(z, c);
            sc0[cw] = sb[(long)c * sbm]; sc1[cw] = sb[(long)(g.pair_off + c) * sbm];
        }
        touch(cc[0]); touch(cc[1]); touch(sc0[0]); touch(sc0[1]); touch(sc1[0]); touch(sc1[1]);
#pragma unroll
        for (int cw = 0; cw < 2; ++cw) { const float2 pm = epi.pairmul(cc[cw]); sc0[cw] *= pm.x; sc1[cw] *= pm.y; }
        // 8 half row blocks hb = (tm, cw, half); the aux operands run AH blocks ahead of the values
        constexpr int AH = H3_AUX_AHEAD;
        decltype(epi.aux(0, 0, 0, RowT{})) ax[AH + 1][8];
        auto fetch = [&](int hb) {
            const int tm = hb >> 2, cw = (hb >> 1) & 1;
            int lb = lrow + tm * 32;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = (hb & 1) * 8 + r8;
                const int lr = lb + (r & 3) + 8 * (r >> 2);
                ax[hb % (AH + 1)][r8] = epi.aux(z, m0 + lr, n0 + lcolb + cw * 32, rwl[lr]);
            }
        };
#pragma unroll
        for (int hb = 0; hb < AH; ++hb) fetch(hb);
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            __builtin_amdgcn_sched_barrier(0);
            if (hb + AH < 8) fetch(hb + AH);
            const int tm = hb >> 2, cw = (hb >> 1) & 1;
            RowT rw[8];
            float sr[8];
            int lb = lrow + tm * 32;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = (hb & 1) * 8 + r8;
                const int lr = lb + (r & 3) + 8 * (r >> 2);
                rw[r8] = rwl[lr];
                sr[r8] = sal[lr];
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) touch(ax[hb % (AH + 1)][r8]);
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = (hb & 1) * 8 + r8;
                const int lr = lb + (r & 3) + 8 * (r >> 2);
                const int lc = lcolb + cw * 32;
                const float v = epi.val2_scaled(z, m0 + lr, n0 + lc, acc[tm][cw][r] * (sr[r8] * sc0[cw]), acc[tm][2 + cw][r] * (sr[r8] * sc1[cw]),
                                                rw[r8], cc[cw], ax[hb % (AH + 1)][r8]);
                T[lr * 128 + (lc ^ (((lr >> 2) & 1) << 5))] = v;
            }
        }
    }
    __syncthreads();
    H3_STAMP(3);
    {
        const int hw = lane >> 5, l32 = lane & 31;
        unsigned char* const pbase = po.planes + (long)n0 * 4;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rl = wave * 32 + it * 2 + hw;
            const long pr = epi.prow(z, m0 + rl);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(T + rl * 128 + ((4 * l32) ^ (((rl >> 2) & 1) << 5)));
            const float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
            const float mu = h3_half_max(h3_absmax4(v));
            const float q2 = h3_half_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
            float inv;
            const float sc = h3_row_scale(mu, inv);
            h3_emit4_pred(pbase + (pr < 0 ? 0 : pr) * po.pitch, l32, v, sc, pr >= 0);
            if (l32 == 0 && pr >= 0) {
                po.scale[(long)bn * po.seg_stride + pr] = inv;
                if (po.ss) po.ss[(long)bn * po.seg_stride + pr] = q2;
            }
        }
    }
    }
    H3_STAMP(4);
}

// usable for this problem?  (the callers fall back to the wide kernel otherwise)
template <bool TWOSEG>
inline bool h3a_fits(const H3Args& g, bool gate) {
    if (g.nseg != (TWOSEG ? 2 : 1) || g.M % 128 || g.N % (gate ? 128 : 256)) return false;
    for (int i = 0; i < g.nseg; ++i) {
        const H3Seg& sg = g.seg[i];
        if (sg.K < 64 || sg.K % 16 || sg.segk || sg.a_shift || sg.a_period) return false;
        if (sg.kchunk) {           // split-K chunks: every chunk has at least three k-tiles
            if (TWOSEG || sg.kchunk % 16 || sg.ktotal % 16 || sg.K != sg.kchunk) return false;
            const int last = sg.ktotal - (sg.zdiv - 1) * sg.kchunk;
            if (last < 48) return false;
        }
    }
    return true;
}

template <bool TWOSEG, class Epi>
inline hipError_t launch_gemm_h3a(H3Args g, int batches, Epi epi, hipStream_t st) {
    constexpr bool GATE = epi_has_plout<Epi>::value;
    // (function-local static: set once, thread-safe — forwards may be issued from several host threads)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3a_kernel<TWOSEG, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, H3A_LDS + H3A_EXTRA);
    (void)attr_rc;
    if (batches < 1 || !h3a_fits<TWOSEG>(g, GATE)) return hipErrorInvalidValue;
    g.tiles_m = g.M / 128;
    g.tiles_n = g.N / (GATE ? 128 : 256);
    if (!GATE) g.pair_off = 128;
    g.batches = batches;
    static const int order = [] { const char* e = getenv("TDX_H3A_ORDER"); return e ? atoi(e) : 0; }();
    g.map_mode = 1; g.mp = order; g.gw = 1;
    static const int dbg = [] { const char* e = getenv("TDX_H3_DEBUG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    dim3 grid(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    hipLaunchKernelGGL((gemm_h3a_kernel<TWOSEG, Epi>), grid, dim3(H3A_THREADS), H3A_LDS + H3A_EXTRA, st, g, epi);
    return hipGetLastError();
}

}  // namespace tdx
