// gemm_h3n.hpp — the split-f16 x3 GEMM core (gemm_h3.hpp) as a 256 x 128 block tile with FOUR waves and TWO blocks per CU.
//
// Why: the 256 x 256 / 8-wave kernel owns its CU alone (128 KB ring), so nothing overlaps its epilogue (VALU + stores, 10-36 % of
// a launch) and its main loop is bound by per-stage latencies (LDS-DMA round trips, one barrier per 16 k), not by the MFMA pipe
// (42-66 % busy).  Two independent 4-wave blocks per CU (72 KB ring each: three 24 KB stages) interleave on the SIMDs: one
// block's epilogue / barrier waits run under the other's MFMAs.  Same wave tile (128 x 64 = 4 x 2 MFMA tiles, 128 accumulator
// VGPRs, two waves per SIMD), same LDS images, same operand formats (row-major planes), same epilogue functor concept.
// Price: 1.5x the operand bytes per flop (A 16 KB + B 8 KB per 24 MFMAs/wave instead of 16 + 16 per 48).
//
// Modes: row-major planes on both sides (nn.Linear), TWOSEG (two K segments with their own pointers / scales, incl. the token
// shift of the first), segmented row scales (H3Seg::segk), batches; epilogues: store()/ptr()/put()/rowmul() functors without
// aux() — everything the non-paired Linear launches of the model need.  K-major operands, PAIRED and PLOUT stay on the wide kernel.
#pragma once
#include "gemm_h3.hpp"

namespace tdx {

constexpr int H3N_THREADS = 256;
constexpr int H3N_A = 256 * H3_ROWB;           // 16 KB
constexpr int H3N_B = 128 * H3_ROWB;           // 8 KB
constexpr int H3N_STAGE = H3N_A + H3N_B;       // 24 KB
constexpr int H3N_NBUF = 3;
constexpr int H3N_LDS = H3N_NBUF * H3N_STAGE;  // 72 KB
constexpr int H3N_EXTRA = 7680;                // TWOSEG row factors 1 KB | row scales 1 KB | row() values 2 KB | segment exponent steps 3.5 KB

template <bool TWOSEG, class Epi, int MFMA_SHAPE = 0>
__global__ __launch_bounds__(H3N_THREADS, 2) void gemm_h3n_kernel(H3Args g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

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
    const int m0 = bm * 256, n0 = bn * 128;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage buffer (row-major planes, same swizzle as the wide kernel)
    const int fl = (l31 >> 2) & 3;
    const int a0 = (wm * 128 + l31) * H3_ROWB, b0 = H3N_A + (wn * 32 + l31) * H3_ROWB;
    const int pl0 = ((2 * h) ^ fl) << 4, pl1 = ((2 * h + 1) ^ fl) << 4;
    // DMA: every wave fetches four 1-KiB pieces of A (rows wave*64 + 16j ..) and two of B (rows wave*32 + 16j ..) per stage
    const int dstA = wave * 4096, dstB = H3N_A + wave * 2048;
    const bool full_n = n0 + 64 < g.N;              // the second column tile of the wave (tn = 1) has columns
    const bool wave_on = m0 + wm * 128 < g.M;

    auto setup = [&](const H3Seg& sg, const unsigned char* (&gp)[6]) {
        const int z1 = z / sg.zdiv, z2 = z - z1 * sg.zdiv;
        const unsigned char* Ag = sg.A + (long)z1 * sg.strideA + (long)z2 * sg.strideA2;
        const unsigned char* Bg = sg.B + (long)z1 * sg.strideB + (long)z2 * sg.strideB2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 64 + (lane >> 2) + 16 * j;
            const int mr = min(m0 + r, g.M - 1);
            const unsigned char* rowp = Ag + (long)(mr + sg.a_shift) * sg.lda;
            if (sg.a_period && (mr % sg.a_period) == 0) rowp = sg.a_zero;
            gp[j] = rowp + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wave * 32 + (lane >> 2) + 16 * j;
            gp[4 + j] = Bg + (long)min(n0 + r, g.N - 1) * sg.ldb + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
    };

    const H3Seg& sg0 = g.seg[0];
    const H3Seg& sl = g.seg[TWOSEG ? 1 : 0];
    const int zl1 = z / sl.zdiv, zl2 = z - zl1 * sl.zdiv;
    using RowT = decltype(epi.row(0, 0));
    constexpr bool HAS_ROW = !std::is_empty<RowT>::value;
    float* rfl2 = reinterpret_cast<float*>(lds + H3N_LDS);                  // TWOSEG row factors
    float* sal = reinterpret_cast<float*>(lds + H3N_LDS + 1024);
    RowT* rwl = reinterpret_cast<RowT*>(lds + H3N_LDS + 2048);
    short* segd = reinterpret_cast<short*>(lds + H3N_LDS + 4096);           // [boundary][256 rows] exponent steps
    {
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
            // the row scales are powers of two: a boundary factor sa_j / sa_{j+1} is kept as its exponent step (2 bytes)
            const float* sab = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)(z % sg0.zdiv) * sg0.strideSA2 + (long)m * sg0.sa_mul;
            const int nb = sg0.K / sg0.segk - 1;
            int prev = (int)((__float_as_uint(sab[0]) >> 23) & 0xff);
            for (int j = 0; j < nb; ++j) {
                const int nxt = (int)((__float_as_uint(sab[(long)(j + 1) * sg0.strideSeg]) >> 23) & 0xff);
                segd[j * 256 + tid] = (short)max(-126, min(127, prev - nxt));
                prev = nxt;
            }
        }
    }

    const int nkt0 = sg0.K / H3_BK;
    const int nkt = TWOSEG ? nkt0 + g.seg[1].K / H3_BK : nkt0;
    {
        const unsigned char* gp[6];
        setup(sg0, gp);
        auto dma = [&](int ti, int tile0) {      // this wave's six pieces of k-tile ti (ti - tile0 tiles into the current segment's pointers)
            unsigned char* st = lds + (ti % H3N_NBUF) * H3N_STAGE;
            const long off = (long)(ti - tile0) * H3_ROWB;
#pragma unroll
            for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + off, st + dstA + j * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) h3_glds16(gp[4 + j] + off, st + dstB + j * 1024);
        };
        // ---- prologue (nkt >= 4 guaranteed by the host: K >= 64)
        dma(0, 0); dma(1, 0); dma(2, 0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
        f16x8 ah[4], al[4], bh[2], bl[2];
        auto frags = [&](const unsigned char* st) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                bh[tn] = *reinterpret_cast<const f16x8*>(st + b0 + tn * 4096 + pl0);
                bl[tn] = *reinterpret_cast<const f16x8*>(st + b0 + tn * 4096 + pl1);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                ah[tm] = *reinterpret_cast<const f16x8*>(st + a0 + tm * 2048 + pl0);
                al[tm] = *reinterpret_cast<const f16x8*>(st + a0 + tm * 2048 + pl1);
            }
        };
        frags(lds);
        const int seg_stp = (!TWOSEG && sg0.segk) ? sg0.segk / H3_BK : 0;
        int seg_bnd = seg_stp ? seg_stp : 0x7fffffff, seg_j = 0;
        int tile0 = 0;
        for (int t = 0; t < nkt; ++t) {
            if constexpr (TWOSEG) {
                if (t + 3 == nkt0) { setup(g.seg[1], gp); tile0 = nkt0; }       // the DMA of this stage (tile t+3) is the second segment's first
                if (t == nkt0) {
                    // acc changes its scale domain between the segments (exact powers of two)
                    const H3Seg& s1 = g.seg[1];
                    const float* sb0 = sg0.sb + (long)(z / sg0.zdiv) * sg0.strideSB + (long)(z % sg0.zdiv) * sg0.strideSB2;
                    const float* sb1 = s1.sb + (long)(z / s1.zdiv) * s1.strideSB + (long)(z % s1.zdiv) * s1.strideSB2;
                    float cf[2];
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const int c = min(n0 + tn * 64 + wn * 32 + l31, g.N - 1);
                        cf[tn] = sb0[(long)c * sg0.sb_mul] / sb1[(long)c * s1.sb_mul];
                    }
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 rf = *reinterpret_cast<const f32x4*>(rfl2 + wm * 128 + tm * 32 + 8 * j + 4 * h);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { acc[tm][0][4 * j + i] *= rf[i] * cf[0]; acc[tm][1][4 * j + i] *= rf[i] * cf[1]; }
                        }
                }
            } else {
                if (t == seg_bnd) {
                    const short* sd = segd + seg_j * 256 + wm * 128 + 4 * h;
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const short4 d4 = *reinterpret_cast<const short4*>(sd + tm * 32 + 8 * j);
                            const float rf[4] = {__uint_as_float((unsigned)(127 + d4.x) << 23), __uint_as_float((unsigned)(127 + d4.y) << 23),
                                                 __uint_as_float((unsigned)(127 + d4.z) << 23), __uint_as_float((unsigned)(127 + d4.w) << 23)};
#pragma unroll
                            for (int i = 0; i < 4; ++i) { acc[tm][0][4 * j + i] *= rf[i]; acc[tm][1][4 * j + i] *= rf[i]; }
                        }
                    ++seg_j; seg_bnd += seg_stp;
                }
            }
            // tile t+1 has landed for this wave (tile t+2 may stay in flight); the barrier makes it readable for everyone and —
            // every wave having retired its reads of tile t — frees buffer t % 3 for tile t+3
            if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            H3_BARRIER();
            const bool next = t + 1 < nkt, issue = t + 3 < nkt;
            const unsigned char* nst = lds + ((t + 1) % H3N_NBUF) * H3N_STAGE;
            unsigned char* dst = lds + (t % H3N_NBUF) * H3N_STAGE;
            const long off = (long)(t + 3 - tile0) * H3_ROWB;
            if (wave_on) {
                f16x8 nbh[2], nbl[2];
                if (next) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        nbh[tn] = *reinterpret_cast<const f16x8*>(nst + b0 + tn * 4096 + pl0);
                        nbl[tn] = *reinterpret_cast<const f16x8*>(nst + b0 + tn * 4096 + pl1);
                    }
                }
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        if (tn == 0 || full_n) {
                            if constexpr (MFMA_SHAPE == 0) {
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                            } else {
                                // TIMING ONLY (wrong results): the same flops as six v_mfma_f32_16x16x32_f16 on the four 16x16 quarters
                                f32x4* q = reinterpret_cast<f32x4*>(&acc[tm][tn]);
                                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bl[tn], q[0], 0, 0, 0);
                                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], q[1], 0, 0, 0);
                                q[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bh[tn], q[2], 0, 0, 0);
                                q[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bl[tn], q[3], 0, 0, 0);
                                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], q[0], 0, 0, 0);
                                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bl[tn], q[1], 0, 0, 0);
                            }
                        }
                        const int pj = tm * 2 + tn;                 // six of the eight slots carry one DMA piece each
                        if (issue && pj < 6) {
                            if (pj < 4) h3_glds16(gp[pj] + off, dst + dstA + pj * 1024);
                            else h3_glds16(gp[pj] + off, dst + dstB + (pj - 4) * 1024);
                        }
                    }
                    if (next) {
                        ah[tm] = *reinterpret_cast<const f16x8*>(nst + a0 + tm * 2048 + pl0);
                        al[tm] = *reinterpret_cast<const f16x8*>(nst + a0 + tm * 2048 + pl1);
                    }
                }
                if (next) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) { bh[tn] = nbh[tn]; bl[tn] = nbl[tn]; }
                }
            } else if (issue) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + off, dst + dstA + j * 1024);
#pragma unroll
                for (int j = 0; j < 2; ++j) h3_glds16(gp[4 + j] + off, dst + dstB + j * 1024);
            }
        }
    }

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h (see gemm_h3.hpp for the waitcnt discipline)
    const float* sb = sl.sb + (long)zl1 * sl.strideSB + (long)zl2 * sl.strideSB2;
    const int sbm = sl.sb_mul;
    const int lrow0 = wm * 128 + 4 * h, lcol0 = wn * 32 + l31;
    auto row_of = [&](int lr) -> RowT { if constexpr (HAS_ROW) return rwl[lr]; else return RowT{}; };
    auto epilogue = [&](auto tag) {
        constexpr bool CHECK = decltype(tag)::value;
        int lrow = lrow0, lcol = lcol0;
        asm volatile("" : "+v"(lrow), "+v"(lcol));
        int nn[2];
        decltype(epi.col(0, 0)) cc[2];
        float sc[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            nn[tn] = n0 + tn * 64 + lcol;
            const int nc = min(nn[tn], g.N - 1);
            cc[tn] = epi.col(z, nc);
            sc[tn] = sb[(long)nc * sbm];
        }
        touch(cc[0]); touch(cc[1]); touch(sc[0]); touch(sc[1]);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            if (nn[tn] >= g.N) continue;
            if constexpr (epi_has_ptr<Epi>::value) {
                float* const p0 = epi.ptr(z, m0 + lrow, nn[tn]);
                const long ldm = epi.ldm();
                float sct = sc[tn];
                if constexpr (epi_has_rowmul<Epi>::value) sct *= epi.colmul(cc[tn]);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const RowT rw = row_of(lrow + k);
                        const float sr = sal[lrow + k];
                        if (!CHECK || m0 + lrow + k < g.M) {
                            if constexpr (epi_has_rowmul<Epi>::value) epi.put_scaled(p0 + k * ldm, acc[tm][tn][r] * (sr * sct), rw, cc[tn]);
                            else epi.put(p0 + k * ldm, acc[tm][tn][r] * (sr * sct), rw, cc[tn]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lr = lrow + tm * 32 + (r & 3) + 8 * (r >> 2);
                        const RowT rw = row_of(lr);
                        const float sr = sal[lr];
                        if (!CHECK || m0 + lr < g.M) epi.store(z, m0 + lr, nn[tn], acc[tm][tn][r] * (sr * sc[tn]), rw, cc[tn]);
                    }
                }
            }
        }
    };
    if (m0 + 256 <= g.M) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

template <bool TWOSEG, class Epi, int MFMA_SHAPE = 0>
inline hipError_t launch_gemm_h3n(H3Args g, int batches, Epi epi, hipStream_t st) {
    static_assert(!epi_has_aux<Epi>::value && !epi_has_plout<Epi>::value, "narrow x3 kernel: store/ptr functors only");
    // (function-local static: set once, thread-safe — forwards may be issued from several host threads)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3n_kernel<TWOSEG, Epi, MFMA_SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, H3N_LDS + H3N_EXTRA);
    (void)attr_rc;
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 127) / 128;
    g.batches = batches;
    long ktot = 0;
    for (int i = 0; i < g.nseg; ++i) {
        const H3Seg& sg = g.seg[i];
        ktot += sg.K;
        if (sg.K < 64 || sg.K % 16 || sg.kchunk) return hipErrorInvalidValue;
        if (sg.segk && (TWOSEG || sg.segk % 64 || sg.K % sg.segk || sg.K / sg.segk > 8)) return hipErrorInvalidValue;
        if (sg.a_period && !sg.a_zero) return hipErrorInvalidValue;
    }
    if ((TWOSEG ? 2 : 1) != g.nseg) return hipErrorInvalidValue;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        long gw = (1536L * 1024) / (128L * ktot * 4);      // weight slice of a group stays in the XCD's L2
        if (gw < 1) gw = 1;
        if (gw > g.tiles_n) gw = g.tiles_n;
        g.gw = (int)gw;
        grid = dim3(8 * g.mp * g.tiles_n, batches, 1);
    } else {
        g.map_mode = 1;
        g.mp = 0; g.gw = 1;
        grid = dim3(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    }
    hipLaunchKernelGGL((gemm_h3n_kernel<TWOSEG, Epi, MFMA_SHAPE>), grid, dim3(H3N_THREADS), H3N_LDS + H3N_EXTRA, st, g, epi);
    return hipGetLastError();
}

}  // namespace tdx
