// gemm_x6.hpp — fp32-accurate GEMM on the bf16 MFMA pipe ("split-bf16 x6").
//
//   C[z][m][n] = epilogue( sum_seg sum_k A_seg[z][m][k] * B_seg[z][k][n] ),  fp32 in HBM, fp32 accumulate.
//
// Each fp32 operand is split on the fly into three bf16 planes  x = x_h + x_m + x_l
// (x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m): 3 x 8 = 24 mantissa bits, and
// bf16 keeps fp32's exponent range, so no scaling is needed) and the product is evaluated as
// the six MFMA passes  hl + hm + hh + mm + mh + lh  (the dropped ml/lm/ll terms are <= 2^-24
// relative, i.e. at the fp32 rounding level).  v_mfma_f32_32x32x16_bf16 runs at 16x the
// fp32-input MFMA rate, so six passes are 2.67x faster than v_mfma_f32_32x32x2_f32 at the
// same accuracy (measured 3.5e-7 vs 4.1e-7 rel-L2 against fp64 at K=512).  SURVEY.md §7.2
// names this scheme as the alternative to the fp32 MFMA.
//
// Operand modes (same GemmArgs as gemm.hpp): A row-major K-contiguous (optionally with the
// token-shift loader); B either [N][K] K-contiguous (nn.Linear) or K-major [K][N] (attention
// values, transposed while staging: a thread takes two consecutive k-rows and writes packed
// (k,k+1) bf16 pairs); up to two K segments; PAIRED columns (gate epilogues); batched (z).
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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int X6_BM = 256, X6_BN = 256, X6_BK = 16, X6_THREADS = 512;
constexpr int X6_PITCH = 48;                 // bytes per LDS row: 16 bf16 + 16 B pad
constexpr int X6_PLANE = 256 * X6_PITCH;     // 12 288 B
constexpr int X6_OPER = 3 * X6_PLANE;        // hi, mid, lo planes of one operand tile
constexpr int X6_BUF = 2 * X6_OPER;          // A then B
constexpr int X6_LDS = 2 * X6_BUF;           // double buffered: 147 456 B

__device__ __forceinline__ void x6_split(float v, __bf16& hi, __bf16& mi, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;
    mi = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mi);
}

// 4 consecutive k of one row -> 8-byte stores into the three planes
__device__ __forceinline__ void x6_split_store(unsigned char* base, f32x4 x, bool ok) {
    bf16x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 a, b, c;
        x6_split(ok ? x[i] : 0.f, a, b, c);
        h[i] = a; m[i] = b; l[i] = c;
    }
    *reinterpret_cast<bf16x4*>(base) = h;
    *reinterpret_cast<bf16x4*>(base + X6_PLANE) = m;
    *reinterpret_cast<bf16x4*>(base + 2 * X6_PLANE) = l;
}

// K-major source: x0 = row k (4 consecutive n), x1 = row k+1 -> for each n a packed (k,k+1) pair
__device__ __forceinline__ void x6_split_store_t(unsigned char* base, f32x4 x0, f32x4 x1, bool ok0, bool ok1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 a0, b0, c0, a1, b1, c1;
        x6_split(ok0 ? x0[i] : 0.f, a0, b0, c0);
        x6_split(ok1 ? x1[i] : 0.f, a1, b1, c1);
        bf16x2 h = {a0, a1}, m = {b0, b1}, l = {c0, c1};
        unsigned char* p = base + i * X6_PITCH;
        *reinterpret_cast<bf16x2*>(p) = h;
        *reinterpret_cast<bf16x2*>(p + X6_PLANE) = m;
        *reinterpret_cast<bf16x2*>(p + 2 * X6_PLANE) = l;
    }
}

template <bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi>
__global__ __launch_bounds__(X6_THREADS, 2) void gemm_x6_kernel(GemmArgs g, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
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
    const int m0 = bm * X6_BM;
    const int n0 = PAIRED ? bn * (X6_BN / 2) : bn * X6_BN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging map.  K-contiguous operand: 256 rows x 4 float4, rows r0 and r0+128, chunk c4.
    const int c4 = tid & 3, r0 = tid >> 2, r1 = r0 + 128;
    const int st0 = r0 * X6_PITCH + c4 * 8, st1 = r1 * X6_PITCH + c4 * 8;
    // K-major B: k-pair kp (rows 2kp, 2kp+1) and column quad cq (n = 4cq .. 4cq+3)
    const int kp = tid & 7, cq = tid >> 3;
    const int stt = (cq * 4) * X6_PITCH + kp * 4;
    // fragment read offsets (bytes inside a plane): tile tn covers LDS rows tn*128 + wn*32 + l31
    const int fa = (wm * 128 + l31) * X6_PITCH + h * 16;      // + tm * 32 rows
    const int fb = (wn * 32 + l31) * X6_PITCH + h * 16;       // + tn * 128 rows

    const int mA0 = m0 + r0, mA1 = m0 + r1;
    const bool okA0 = mA0 < g.M, okA1 = mA1 < g.M;
    // last N tile of a width that is not a multiple of 256: its upper 128 columns (MFMA tile
    // tn = 1 of every wave) do not exist -> skip those MFMAs (block-uniform branch)
    const bool full_n = PAIRED || (n0 + 128 < g.N);

    for (int s = 0; s < g.nseg; ++s) {
        const bool s0 = s == 0;
        const int zdiv = s0 ? g.seg[0].zdiv : g.seg[1].zdiv;
        const int z1 = z / zdiv, z2 = z - z1 * zdiv;
        const float* __restrict__ Ag = (s0 ? g.seg[0].A : g.seg[1].A) + (long)z1 * (s0 ? g.seg[0].strideA : g.seg[1].strideA) +
                                       (long)z2 * (s0 ? g.seg[0].strideA2 : g.seg[1].strideA2);
        const float* __restrict__ Bg = (s0 ? g.seg[0].B : g.seg[1].B) + (long)z1 * (s0 ? g.seg[0].strideB : g.seg[1].strideB) +
                                       (long)z2 * (s0 ? g.seg[0].strideB2 : g.seg[1].strideB2);
        const long lda = s0 ? g.seg[0].lda : g.seg[1].lda, ldb = s0 ? g.seg[0].ldb : g.seg[1].ldb;
        const int K = s0 ? g.seg[0].K : g.seg[1].K;
        const int nkt = K / X6_BK;
        const int kvalid = min(K, max(0, (s0 ? g.seg[0].ktotal : g.seg[1].ktotal) - z2 * (s0 ? g.seg[0].kchunk : g.seg[1].kchunk)));

        const float* pa0 = Ag + (long)min(mA0, g.M - 1) * lda + c4 * 4;
        const float* pa1 = Ag + (long)min(mA1, g.M - 1) * lda + c4 * 4;
        const float *ps0 = pa0, *ps1 = pa1;
        bool oks0 = okA0, oks1 = okA1;
        if constexpr (SHIFT) {
            oks0 = okA0 && (mA0 % g.shift_S != 0); oks1 = okA1 && (mA1 % g.shift_S != 0);
            ps0 = Ag + (long)max(min(mA0, g.M - 1) - 1, 0) * lda + c4 * 4;
            ps1 = Ag + (long)max(min(mA1, g.M - 1) - 1, 0) * lda + c4 * 4;
        }
        const float *pb0, *pb1;
        if constexpr (!B_KMAJOR) {
            const int nA = PAIRED ? (r0 < 128 ? n0 + r0 : g.pair_off + n0 + r0 - 128) : n0 + r0;
            const int nB = PAIRED ? g.pair_off + n0 + r1 - 128 : n0 + r1;      // r1 >= 128 always
            pb0 = Bg + (long)nA * ldb + c4 * 4;
            pb1 = Bg + (long)nB * ldb + c4 * 4;
        } else {
            const int nl = cq * 4;
            const int n = PAIRED ? (nl < 128 ? n0 + nl : g.pair_off + n0 + nl - 128) : n0 + nl;
            pb0 = Bg + n;      // + k * ldb per tile
            pb1 = pb0;
        }

        f32x4 ra0, ra1, rb0, rb1;
        bool va0 = okA0, va1 = okA1, vb0 = true, vb1 = true;
#define X6_LOAD(kt)                                                                 \
        {                                                                           \
            const int k0_ = (kt) * X6_BK;                                           \
            if constexpr (SHIFT) {                                                  \
                const bool sh = k0_ < g.shift_k;                                    \
                ra0 = ldg4((sh ? ps0 : pa0) + k0_); va0 = sh ? oks0 : okA0;         \
                ra1 = ldg4((sh ? ps1 : pa1) + k0_); va1 = sh ? oks1 : okA1;         \
            } else { ra0 = ldg4(pa0 + k0_); ra1 = ldg4(pa1 + k0_); }                \
            if constexpr (!B_KMAJOR) { rb0 = ldg4(pb0 + k0_); rb1 = ldg4(pb1 + k0_); } \
            else {                                                                  \
                const int ka_ = k0_ + 2 * kp, kb_ = ka_ + 1;                        \
                vb0 = ka_ < kvalid; vb1 = kb_ < kvalid;                             \
                rb0 = ldg4(pb0 + (long)min(ka_, kvalid - 1) * ldb);                 \
                rb1 = ldg4(pb0 + (long)min(kb_, kvalid - 1) * ldb);                 \
            }                                                                       \
        }
#define X6_STORE(buf)                                                               \
        {                                                                           \
            unsigned char* b_ = lds + (buf) * X6_BUF;                               \
            x6_split_store(b_ + st0, ra0, va0);                                     \
            x6_split_store(b_ + st1, ra1, va1);                                     \
            if constexpr (!B_KMAJOR) {                                              \
                x6_split_store(b_ + X6_OPER + st0, rb0, true);                      \
                x6_split_store(b_ + X6_OPER + st1, rb1, true);                      \
            } else {                                                                \
                x6_split_store_t(b_ + X6_OPER + stt, rb0, rb1, vb0, vb1);           \
            }                                                                       \
        }

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
                bh[tn] = *reinterpret_cast<const bf16x8*>(Bp + tn * 128 * X6_PITCH);
                bmid[tn] = *reinterpret_cast<const bf16x8*>(Bp + X6_PLANE + tn * 128 * X6_PITCH);
                bl[tn] = *reinterpret_cast<const bf16x8*>(Bp + 2 * X6_PLANE + tn * 128 * X6_PITCH);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + tm * 32 * X6_PITCH);
            // next tile: registers -> three bf16 planes in the other buffer, then refill the registers
            if (kt + 1 < nkt) {
                X6_STORE((kt + 1) & 1)
                if (kt + 2 < nkt) X6_LOAD(kt + 2)
            }
            // hi x {lo, mid, hi}
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if (tn == 1 && !full_n) continue;
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bmid[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
            // mid x {mid, hi}
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + X6_PLANE + tm * 32 * X6_PITCH);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if (tn == 1 && !full_n) continue;
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bmid[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
            // lo x hi
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(Ap + 2 * X6_PLANE + tm * 32 * X6_PITCH);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if (tn == 1 && !full_n) continue;
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
            __syncthreads();
        }
#undef X6_LOAD
#undef X6_STORE
    }

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h.  Per-row constants and the
    // auxiliary loads of 16 rows are fetched before the first store of the batch.
    // (whole tiles take a branch-free path: per-row `if (m < M)` blocks cost one store round trip per store, see gemm_h3.hpp)
    auto epilogue = [&](auto tag) {
    constexpr bool CHECK = decltype(tag)::value;
    if constexpr (PAIRED) {
        const int c = n0 + wn * 32 + l31;
        auto cc = epi.col(z, c);
        touch(cc);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            decltype(epi.row(0, 0)) rw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rw[r] = epi.row(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1));
            if constexpr (epi_has_aux<Epi>::value) {
                decltype(epi.aux(0, 0, 0, rw[0])) ax[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[r] = epi.aux(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1), c, rw[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) { touch(rw[r]); touch(ax[r]); }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (!CHECK || m < g.M) epi.store2(z, m, c, acc[tm][0][r], acc[tm][1][r], rw[r], cc, ax[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) touch(rw[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (!CHECK || m < g.M) epi.store2(z, m, c, acc[tm][0][r], acc[tm][1][r], rw[r], cc);
                }
            }
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + tn * 128 + wn * 32 + l31;
            if (n >= g.N) continue;
            auto cc = epi.col(z, n);
            touch(cc);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                decltype(epi.row(0, 0)) rw[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rw[r] = epi.row(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1));
                if constexpr (epi_has_aux<Epi>::value) {
                    decltype(epi.aux(0, 0, 0, rw[0])) ax[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ax[r] = epi.aux(z, min(m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1), n, rw[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { touch(rw[r]); touch(ax[r]); }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (!CHECK || m < g.M) epi.store(z, m, n, acc[tm][tn][r], rw[r], cc, ax[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) touch(rw[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (!CHECK || m < g.M) epi.store(z, m, n, acc[tm][tn][r], rw[r], cc);
                    }
                }
            }
        }
    }
    };
    if (m0 + X6_BM <= g.M) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

// !PAIRED: N any value <= rows readable in 256-row tiles of W (pad the weight buffer), columns
// >= N are not stored.  PAIRED: N = number of pair columns, multiple of 128.
template <bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi>
inline hipError_t launch_gemm_x6(GemmArgs g, int batches, Epi epi, hipStream_t st) {
    // (function-local static: set once, thread-safe — forwards may be issued from several host threads)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<B_KMAJOR, PAIRED, SHIFT, Epi>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS);
    (void)attr_rc;
    g.tiles_m = (g.M + X6_BM - 1) / X6_BM;
    g.tiles_n = PAIRED ? g.N / (X6_BN / 2) : (g.N + X6_BN - 1) / X6_BN;
    g.batches = batches;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        long ktot = 0;
        for (int i = 0; i < g.nseg; ++i) ktot += g.seg[i].K;
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
    hipLaunchKernelGGL((gemm_x6_kernel<B_KMAJOR, PAIRED, SHIFT, Epi>), grid, dim3(X6_THREADS), X6_LDS, st, g, epi);
    return hipGetLastError();
}

// nn.Linear / 1x1 conv convenience form
template <bool SHIFT, class Epi>
inline hipError_t launch_gemm_x6(GemmArgs g, Epi epi, hipStream_t st) {
    return launch_gemm_x6<false, false, SHIFT, Epi>(g, 1, epi, st);
}

}  // namespace tdx
