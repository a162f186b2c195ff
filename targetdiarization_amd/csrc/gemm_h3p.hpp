// gemm_h3p.hpp — the split-f16 x3 GEMM core with v_mfma_f32_16x16x32_f16 ("pair-stage" kernel).
//
// Same block tile (256 x 256), waves (8 as 2 x 4, wave tile 128 x 64), LDS images, LDS-DMA ring (four 32 KB slots of 16 k) and
// operand formats (row-major planes) as gemm_h3.hpp, but the MFMAs are 16 x 16 x 32: under real (toggling) data that shape
// delivers 1.08-1.16x the FLOP/s of 32 x 32 x 16 at equal cycles per FLOP (MI355X_MICROARCH.md; measured on this kernel's own
// loads as timing variant 6 of gemm_h3.hpp: to_hidden shape 1958 -> 1641 us, K = 2048 shape 1643 -> 1504 us).
//
// A K = 32 instruction needs 32 k-slots per operand.  The three products of the split (hi*lo' + lo*hi' + hi*hi') over 16 k
// are 48 slots = 1.5 instructions, so the loop works on PAIRS of 16-k tiles (32 k): per 16 x 16 output tile
//     X = hi(32 k), Y = lo(32 k) of A;  P = hi'(32 k), Q = lo'(32 k) of B:   acc += X*Q + Y*P + X*P      (3 MFMAs)
// Lane l of a fragment supplies row (l & 15) and k = 8*(l >> 4) .. +7: lane groups 0,1 read the pair's first tile (its two
// 8-k chunks), groups 2,3 the second — a per-lane constant offset of one ring slot, ONE ds_read_b128 per fragment, the same
// XOR swizzle as the wide kernel (conflict-free per 16-lane group).
// Pipeline: a pair owns two ring slots for its whole stage (A fragments are read just in time, one row block ahead: all
// fragments of a pair would need 96 VGPRs next to the 128 accumulators), so the ring holds two pairs: the DMA of pair p+1 is
// issued at the start of stage p and must land during it (one pair-stage = 96 MFMAs per wave of look-ahead; the wide kernel has
// three 24-MFMA stages).  One barrier per 32 k instead of one per 16 k.
//
// Modes: row-major planes both sides, TWOSEG (incl. the token-shifted first segment), segmented row scales, batches;
// store()/ptr() functors without aux().  Everything else stays on gemm_h3.hpp.
#pragma once
#include "gemm_h3.hpp"

namespace tdx {

template <bool TWOSEG, class Epi>
__global__ __launch_bounds__(H3_THREADS, 2) void gemm_h3p_kernel(H3Args g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
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
    const int m0 = bm * H3_BM, n0 = bn * H3_BN;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses relative to a pair's first slot: lane groups 2,3 read the second tile of the pair (one slot further)
    const int swz = (r16 >> 2) & 3, tsel = kg >> 1, ch = kg & 1;
    const int aoffh = tsel * H3_STAGE + (wm * 128 + r16) * H3_ROWB + (((2 * ch) ^ swz) << 4);
    const int aoffl = tsel * H3_STAGE + (wm * 128 + r16) * H3_ROWB + (((2 * ch + 1) ^ swz) << 4);
    const int boffh = tsel * H3_STAGE + H3_OPER + (wn * 32 + r16) * H3_ROWB + (((2 * ch) ^ swz) << 4);
    const int boffl = tsel * H3_STAGE + H3_OPER + (wn * 32 + r16) * H3_ROWB + (((2 * ch + 1) ^ swz) << 4);
    // DMA roles as in the wide kernel: waves 0-3 fetch A, waves 4-7 fetch B, four 1-KiB pieces per 16-k tile
    const bool stB = wave >= 4;
    const int sdst = (stB ? H3_OPER : 0) + (wave & 3) * 4096;
    const bool full_n = n0 + 128 < g.N;             // column blocks j = 2, 3 (the tile's second 128 columns) exist
    const bool wave_on = m0 + wm * 128 < g.M;

    auto setup = [&](const H3Seg& sg, const unsigned char* (&gp)[4]) {
        const int z1 = z / sg.zdiv, z2 = z - z1 * sg.zdiv;
        if (!stB) {
            const unsigned char* Ag = sg.A + (long)z1 * sg.strideA + (long)z2 * sg.strideA2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (wave & 3) * 64 + (lane >> 2) + 16 * j;
                const int mr = min(m0 + r, g.M - 1);
                const unsigned char* rowp = Ag + (long)(mr + sg.a_shift) * sg.lda;
                if (sg.a_period && (mr % sg.a_period) == 0) rowp = sg.a_zero;
                gp[j] = rowp + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
            }
        } else {
            const unsigned char* Bg = sg.B + (long)z1 * sg.strideB + (long)z2 * sg.strideB2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (wave & 3) * 64 + (lane >> 2) + 16 * j;
                gp[j] = Bg + (long)min(n0 + r, g.N - 1) * sg.ldb + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
            }
        }
    };

    const H3Seg& sg0 = g.seg[0];
    const H3Seg& sl = g.seg[TWOSEG ? 1 : 0];
    const int zl1 = z / sl.zdiv, zl2 = z - zl1 * sl.zdiv;
    using RowT = decltype(epi.row(0, 0));
    constexpr bool HAS_ROW = !std::is_empty<RowT>::value;
    float* rfl2 = reinterpret_cast<float*>(lds + H3_LDS);                   // TWOSEG row factors
    float* sal = reinterpret_cast<float*>(lds + H3_LDS + 1024);
    RowT* rwl = reinterpret_cast<RowT*>(lds + H3_LDS + 2048);
    float* rft = reinterpret_cast<float*>(lds + H3_LDS + 4096);             // segment boundary factors [boundary][256]
    if (tid < 256) {
        const int m = min(m0 + tid, g.M - 1);
        float sa_own = (sl.sa + (long)zl1 * sl.strideSA + (long)zl2 * sl.strideSA2 + (sl.segk ? (long)(sl.K / sl.segk - 1) * sl.strideSeg : 0L))[(long)m * sl.sa_mul];
        RowT rw_own{};
        if constexpr (HAS_ROW) rw_own = epi.row(z, m);
        if constexpr (epi_has_rowmul<Epi>::value) sa_own *= epi.rowmul(rw_own);
        sal[tid] = sa_own;
        if constexpr (HAS_ROW) rwl[tid] = rw_own;
        if constexpr (TWOSEG) {
            const H3Seg& s1 = g.seg[1];
            const float* sa0 = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)(z % sg0.zdiv) * sg0.strideSA2;
            const float* sa1 = s1.sa + (long)(z / s1.zdiv) * s1.strideSA + (long)(z % s1.zdiv) * s1.strideSA2;
            const bool zr = sg0.a_period && (m % sg0.a_period) == 0;
            const float f0 = zr ? 1.0f : sa0[(long)(m + sg0.a_shift) * sg0.sa_mul];
            rfl2[tid] = f0 / sa1[(long)m * s1.sa_mul];
        } else if (sg0.segk) {
            const float* sab = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)(z % sg0.zdiv) * sg0.strideSA2 + (long)m * sg0.sa_mul;
            const int nb = sg0.K / sg0.segk - 1;
            float prev = sab[0];
            for (int j = 0; j < nb; ++j) {
                const float nxt = sab[(long)(j + 1) * sg0.strideSeg];
                rft[j * 256 + tid] = prev / nxt;
                prev = nxt;
            }
        }
    }

    const int nkt0 = sg0.K / H3_BK;
    const int nkt = TWOSEG ? nkt0 + g.seg[1].K / H3_BK : nkt0;          // (even, >= 4: the host checks K % 32 == 0 per segment)
    const int np = nkt / 2;
    {
        const unsigned char* gp[4];
        setup(sg0, gp);
        int tile0 = 0;
        // this wave's pieces of 16-k tile `ti` (slot ti & 3)
        auto dma1 = [&](int ti, int j) { h3_glds16(gp[j] + (long)(ti - tile0) * H3_ROWB, lds + (ti & 3) * H3_STAGE + sdst + j * 1024); };
#pragma unroll
        for (int j = 0; j < 4; ++j) dma1(0, j);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma1(1, j);
        const int seg_stp = (!TWOSEG && sg0.segk) ? sg0.segk / (2 * H3_BK) : 0;       // boundaries in pair-stages (segk % 64 == 0)
        int seg_bnd = seg_stp ? seg_stp : 0x7fffffff, seg_j = 0;
        auto rescale_rows = [&](const float* rf, float c01, float c23) {
            // rf: 256 per-row factors of this tile; a lane's rows are wm*128 + i*16 + 4*kg + e
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(rf + wm * 128 + i * 16 + 4 * kg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[i][0][e] *= f[e] * c01; acc[i][1][e] *= f[e] * c01;
                    acc[i][2][e] *= f[e] * c23; acc[i][3][e] *= f[e] * c23;
                }
            }
        };
        for (int p = 0; p < np; ++p) {
            if constexpr (TWOSEG) {
                if (2 * p == nkt0) {           // acc changes its scale domain between the segments (exact powers of two)
                    const H3Seg& s1 = g.seg[1];
                    const float* sb0 = sg0.sb + (long)(z / sg0.zdiv) * sg0.strideSB + (long)(z % sg0.zdiv) * sg0.strideSB2;
                    const float* sb1 = s1.sb + (long)(z / s1.zdiv) * s1.strideSB + (long)(z % s1.zdiv) * s1.strideSB2;
                    if (sb0 != sb1 || sg0.sb_mul != s1.sb_mul) {
                        // per-column factors differ between the lane's four column blocks: apply them block by block
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = min(n0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + r16, g.N - 1);
                            const float cf = sb0[(long)c * sg0.sb_mul] / sb1[(long)c * s1.sb_mul];
#pragma unroll
                            for (int i = 0; i < 8; ++i)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][e] *= cf;
                        }
                    }
                    rescale_rows(rfl2, 1.0f, 1.0f);
                }
            } else {
                if (p == seg_bnd) { rescale_rows(rft + seg_j * 256, 1.0f, 1.0f); ++seg_j; seg_bnd += seg_stp; }
            }
            // pair p has landed for this wave (nothing else is in flight); the barrier makes it readable for everyone and — every
            // wave having retired its reads of pair p-1 — frees that pair's two slots for pair p+1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            H3_BARRIER();
            const unsigned char* base = lds + (p & 1) * 2 * H3_STAGE;
            const bool issue = p + 1 < np;
            if constexpr (TWOSEG) {
                if (issue && 2 * (p + 1) == nkt0) { setup(g.seg[1], gp); tile0 = nkt0; }
            }
            // the whole next pair is requested NOW (it has exactly this stage to land), before the fragment reads
            if (issue) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma1(2 * p + 2 + (i >> 2), i & 3);
            }
            if (wave_on) {
                f16x8 P[4], Q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < 2 || full_n) {
                        P[j] = *reinterpret_cast<const f16x8*>(base + boffh + (j >> 1) * 8192 + (j & 1) * 1024);
                        Q[j] = *reinterpret_cast<const f16x8*>(base + boffl + (j >> 1) * 8192 + (j & 1) * 1024);
                    }
                }
                // A fragments two row blocks ahead of the MFMAs that consume them
                f16x8 X[3], Y[3];
                X[0] = *reinterpret_cast<const f16x8*>(base + aoffh); Y[0] = *reinterpret_cast<const f16x8*>(base + aoffl);
                X[1] = *reinterpret_cast<const f16x8*>(base + aoffh + 1024); Y[1] = *reinterpret_cast<const f16x8*>(base + aoffl + 1024);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 6) {
                        X[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(base + aoffh + (i + 2) * 1024);
                        Y[(i + 2) % 3] = *reinterpret_cast<const f16x8*>(base + aoffl + (i + 2) * 1024);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j < 2 || full_n) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[i % 3], Q[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Y[i % 3], P[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[i % 3], P[j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue.  D of a 16x16 tile: col = lane & 15, row = 4*(lane >> 4) + e.  A lane owns rows wm*128 + i*16 + 4*kg + e
    // (i < 8, e < 4) and columns (j >> 1)*128 + wn*32 + (j & 1)*16 + r16 (j < 4).  Same waitcnt discipline as gemm_h3.hpp.
    const float* sb = sl.sb + (long)zl1 * sl.strideSB + (long)zl2 * sl.strideSB2;
    const int sbm = sl.sb_mul;
    const int lrow0 = wm * 128 + 4 * kg, lcol0 = wn * 32 + r16;
    auto row_of = [&](int lr) -> RowT { if constexpr (HAS_ROW) return rwl[lr]; else return RowT{}; };
    auto epilogue = [&](auto tag) {
        constexpr bool CHECK = decltype(tag)::value;
        int lrow = lrow0, lcol = lcol0;
        asm volatile("" : "+v"(lrow), "+v"(lcol));
        int nn[4];
        decltype(epi.col(0, 0)) cc[4];
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            nn[j] = n0 + (j >> 1) * 128 + (j & 1) * 16 + lcol;
            const int nc = min(nn[j], g.N - 1);
            cc[j] = epi.col(z, nc);
            sc[j] = sb[(long)nc * sbm];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { touch(cc[j]); touch(sc[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (nn[j] >= g.N) continue;
            if constexpr (epi_has_ptr<Epi>::value) {
                float* const p0 = epi.ptr(z, m0 + lrow, nn[j]);
                const long ldm = epi.ldm();
                float sct = sc[j];
                if constexpr (epi_has_rowmul<Epi>::value) sct *= epi.colmul(cc[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = i * 16 + e;
                        const RowT rw = row_of(lrow + k);
                        const float sr = sal[lrow + k];
                        if (!CHECK || m0 + lrow + k < g.M) {
                            if constexpr (epi_has_rowmul<Epi>::value) epi.put_scaled(p0 + k * ldm, acc[i][j][e] * (sr * sct), rw, cc[j]);
                            else epi.put(p0 + k * ldm, acc[i][j][e] * (sr * sct), rw, cc[j]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lr = lrow + i * 16 + e;
                        const RowT rw = row_of(lr);
                        const float sr = sal[lr];
                        if (!CHECK || m0 + lr < g.M) epi.store(z, m0 + lr, nn[j], acc[i][j][e] * (sr * sc[j]), rw, cc[j]);
                    }
                }
            }
        }
    };
    if (m0 + 256 <= g.M) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

template <bool TWOSEG, class Epi>
inline hipError_t launch_gemm_h3p(H3Args g, int batches, Epi epi, hipStream_t st) {
    static_assert(!epi_has_aux<Epi>::value && !epi_has_plout<Epi>::value, "pair-stage x3 kernel: store/ptr functors only");
    // (function-local static: set once, thread-safe — forwards may be issued from several host threads)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3p_kernel<TWOSEG, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS + H3_LDS_EXTRA);
    (void)attr_rc;
    g.tiles_m = (g.M + H3_BM - 1) / H3_BM;
    g.tiles_n = (g.N + H3_BN - 1) / H3_BN;
    g.batches = batches;
    long ktot = 0;
    for (int i = 0; i < g.nseg; ++i) {
        const H3Seg& sg = g.seg[i];
        ktot += sg.K;
        if (sg.K < 64 || sg.K % 32 || sg.kchunk) return hipErrorInvalidValue;
        if (sg.segk && (TWOSEG || sg.segk % 64 || sg.K % sg.segk || sg.K / sg.segk > 8)) return hipErrorInvalidValue;
        if (sg.a_period && !sg.a_zero) return hipErrorInvalidValue;
    }
    if ((TWOSEG ? 2 : 1) != g.nseg) return hipErrorInvalidValue;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        long gw = (1536L * 1024) / (256L * ktot * 4);
        if (gw < 1) gw = 1;
        if (gw > g.tiles_n) gw = g.tiles_n;
        g.gw = (int)gw;
        grid = dim3(8 * g.mp * g.tiles_n, batches, 1);
    } else {
        g.map_mode = 1;
        g.mp = 0; g.gw = 1;
        grid = dim3(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    }
    hipLaunchKernelGGL((gemm_h3p_kernel<TWOSEG, Epi>), grid, dim3(H3_THREADS), H3_LDS + H3_LDS_EXTRA, st, g, epi);
    return hipGetLastError();
}

}  // namespace tdx
