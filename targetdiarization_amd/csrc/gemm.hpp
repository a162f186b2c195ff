// gemm.hpp — fp32 MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32, exact-f32 numerics).
//
// Every nn.Linear / 1x1 Conv1d / einsum on the MossFormer2 path (SURVEY.md Appendix D
// cheat-sheet) is an instance of this one template:
//     C[z][m][n] = epilogue( sum_seg sum_k A_seg[z][m][k] * B_seg[z][k][n] )
//  * A is row-major with K contiguous (activations [tokens, channels]) or K-major
//    (A_KMAJOR: memory is [K][M], used for lin_k^T).
//  * B is [N][K] with K contiguous (PyTorch Linear weight; "NT") or K-major [K][N]
//    (B_KMAJOR: values/keys indexed by token).
//  * up to two K segments (the attention GEMM concatenates [A_quad | lin_q] x [VU ; Kvu]).
//  * PAIRED: the block's 128 LDS columns are 64 columns at c0 and 64 columns at c0+pair_off,
//    so that one lane ends up with both members of a gated pair (att_v/att_u, tanh/sigmoid).
//
// Tiling: 128x128 block, 4 waves as 2(M)x2(N), each wave 64x64 = 2x2 MFMA tiles of 32x32
// (64 accumulator VGPRs), BK = 32.  LDS: K-contiguous operands are stored with a row pitch
// of 36 floats so that the ds_read_b128 fragment reads (lane = row, 4 consecutive k) are
// bank-conflict free (36*r mod 64 distinct for 16 consecutive rows); K-major operands are
// stored [k][128] and read with conflict-free ds_read_b32.  Global->LDS is register staged
// with the next tile's loads in flight during the MFMA phase; 36 KB LDS per block keeps
// 3-4 blocks per CU resident, which is what hides the barriers (fp32 MFMA: 64 cycles each).
//
// MFMA operand map (cdna_hip_programming.md §3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// A lane loads 4 consecutive k (k = 8*kc + 4*(l>>5) + j) for both operands, so register j
// of the fragment pairs k-values {8kc+j, 8kc+4+j} — the same permutation on A and B, hence
// a valid (re-ordered) K summation.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_PITCH = GEMM_BK + 4;   // K-contiguous operand row pitch (floats)

struct GemmSeg {
    const float* A;
    const float* B;
    long lda, ldb;        // leading dimensions (floats)
    long strideA, strideB;  // per-batch (grid.y) strides (floats)
    int K;                // multiple of 32 (K-major operands: rows >= kvalid read as zero)
    // batch index z = blockIdx.y is split as z1 = z / zdiv, z2 = z % zdiv:
    //   A += z1*strideA + z2*strideA2 ; B += z1*strideB + z2*strideB2
    //   valid k rows (K-major operands only) = clamp(ktotal - z2*kchunk, 0, K)
    int zdiv;
    long strideA2, strideB2;
    int kchunk, ktotal;
};

struct GemmArgs {
    GemmSeg seg[2];
    int nseg;
    int M;            // valid rows per batch (rows >= M: A reads zero, C not stored)
    int N;            // columns (multiple of 128; PAIRED: number of pair columns, multiple of 64)
    int tiles_m, tiles_n;
    // block -> tile mapping (both are XCD-aware: blocks with equal blockIdx.x % 8 share an XCD/L2)
    //  map_mode 0: XCD x owns M-panels [x*mp,(x+1)*mp); it sweeps them once per group of `gw`
    //              N-tiles, so the group's weight slice stays L2-resident while A panels stream.
    //  map_mode 1: batched small GEMMs, 1-D grid: XCD x owns batches z = 8*j + x, so all tiles
    //              of a batch (which share its operands) hit one L2.
    int map_mode, mp, gw, batches;
    // CONV (implicit GEMM, NHWC): output row m = (b, h, w) over [B, Hout, Wout]; K segment `tap`
    // reads input row (b, h*stride+dy, w*stride+dx) of A = [B*Hin*Win, lda] (zero outside), with
    // (dy,dx) = (tap/3-1, tap%3-1) for 9 taps or (0,0) for 1; weights B = [N][ntaps*cin].
    int cv_Hin, cv_Win, cv_Hout, cv_Wout, cv_stride, cv_ntaps, cv_cin;
    int n_valid;      // fp32 core, !PAIRED: columns >= n_valid are never stored by the epilogue -> their MFMA tiles are skipped
    int pair_off;     // PAIRED: column offset of the second member of a pair
    int shift_k;      // SHIFT: k < shift_k is read from row m-1 (zero when m % shift_S == 0)
    int shift_S;
};

// XCD-aware bijective remap of the linear block id (cdna_hip_programming.md §5 T1):
// blocks that share an XCD (id % 8 equal) get consecutive tile ids, so the N-tiles that
// re-read one A row-panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Epilogue concept:
//   struct E { Col col(int z,int n) const; Row row(int z,int m) const;
//              void store(int z,int m,int n,float v,Row,Col) const;            // plain
//              void store2(int z,int m,int c,float v0,float v1,Row,Col) const; // PAIRED }
struct EpiNone {};
// "this value is needed here": an empty asm that reads the registers of a loaded value, so that the compiler places its
// (counted) s_waitcnt in straight-line code ahead of the stores.  Left pending across the per-row / per-column branches of
// a store loop, a load makes the waitcnt pass re-wait vmcnt(0) at every block entry — and vmcnt counts stores.
template <class T> __device__ __forceinline__ void touch(T& v) {
    if constexpr (!std::is_empty<T>::value && sizeof(T) % 4 == 0) {
        float* w = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(T) / 4); ++i) asm volatile("" : "+v"(w[i]));
    }
}
// optional member  Aux aux(int z,int m,int n,Row) const : a value the epilogue has to LOAD per
// output element (residual, gate operands ...).  The kernels fetch it for 16 elements at a time
// before the first store of the batch, so the loads overlap instead of forming a load->store
// chain per element (stores may alias the loads as far as the compiler can tell).
template <class E, class = void> struct epi_has_aux { static constexpr bool value = false; };
template <class E> struct epi_has_aux<E, decltype((void)&E::aux, void())> { static constexpr bool value = true; };

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 sel4(bool ok, f32x4 v) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : zero;
}

// Per-thread staging of one operand tile: 4 x 16 B global loads from clamped (always
// in-bounds) addresses.  The validity mask is applied when the registers are written to LDS
// (one tile later), never at load time, so the loads stay in flight across the MFMA phase.
template <bool KMAJOR>
struct StageAddr {
    const float* p[4];   // address of k-tile 0
    unsigned ok;         // bit i: element i valid (K-contiguous: row guard; K-major: unused)
    int kr[4];           // K-major: k row within the tile
};

// VARIANT != 0 are timing-only diagnostic builds reachable through tdx_linear_variant
// (1: no global loads inside the k-loop, 2: no barriers, 3: loads as one burst at the tile start) — results are wrong by design.
template <bool A_KMAJOR, bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi, int VARIANT = 0, bool CONV = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(GemmArgs g, Epi epi) {
    constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_BK, P = GEMM_PITCH;
    constexpr int A_ELEMS = A_KMAJOR ? BK * BM : BM * P;
    constexpr int B_ELEMS = B_KMAJOR ? BK * BN : BN * P;
    __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];
    float* As = lds;
    float* Bs = lds + A_ELEMS;

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
    const int m0 = bm * BM;
    const int n0 = PAIRED ? bn * (BN / 2) : bn * BN;

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

    // LDS addresses of this thread's staging stores (loop invariant)
    const int i0 = tid, i1 = tid + 256, i2 = tid + 512, i3 = tid + 768;
#define TDX_ST_A(i) (A_KMAJOR ? ((i) >> 5) * BM + ((i) & 31) * 4 : ((i) >> 3) * P + ((i) & 7) * 4)
#define TDX_ST_B(i) (B_KMAJOR ? ((i) >> 5) * BN + ((i) & 31) * 4 : ((i) >> 3) * P + ((i) & 7) * 4)
    const int sa0 = TDX_ST_A(i0), sa1 = TDX_ST_A(i1), sa2 = TDX_ST_A(i2), sa3 = TDX_ST_A(i3);
    const int sb0 = TDX_ST_B(i0), sb1 = TDX_ST_B(i1), sb2 = TDX_ST_B(i2), sb3 = TDX_ST_B(i3);
    const int ml0 = wm * 64 + l31, ml1 = ml0 + 32;
    const int nl0 = wn * 32 + l31, nl1 = nl0 + 64;

    // CONV: decode this thread's four staging rows once
    int cvb0 = 0, cvb1 = 0, cvb2 = 0, cvb3 = 0, cvh0 = 0, cvh1 = 0, cvh2 = 0, cvh3 = 0, cvw0 = 0, cvw1 = 0, cvw2 = 0, cvw3 = 0;
    if constexpr (CONV) {
#define TDX_CV_DECODE(i, cvb, cvh, cvw)                                  \
        {                                                                \
            const int m = min(m0 + ((i) >> 3), g.M - 1);                 \
            const int hw = g.cv_Hout * g.cv_Wout;                        \
            const int b_ = m / hw, r_ = m - b_ * hw;                     \
            const int h_ = r_ / g.cv_Wout, w_ = r_ - h_ * g.cv_Wout;     \
            cvb = b_ * g.cv_Hin * g.cv_Win; cvh = h_ * g.cv_stride; cvw = w_ * g.cv_stride; \
        }
        TDX_CV_DECODE(i0, cvb0, cvh0, cvw0) TDX_CV_DECODE(i1, cvb1, cvh1, cvw1)
        TDX_CV_DECODE(i2, cvb2, cvh2, cvw2) TDX_CV_DECODE(i3, cvb3, cvh3, cvw3)
#undef TDX_CV_DECODE
    }
    const int nseg = CONV ? g.cv_ntaps : g.nseg;
    const bool nv0 = PAIRED || n0 + wn * 32 < g.n_valid, nv1 = PAIRED || n0 + 64 + wn * 32 < g.n_valid;

    for (int s = 0; s < nseg; ++s) {
        const bool s0 = CONV ? true : s == 0;
        const int zdiv = s0 ? g.seg[0].zdiv : g.seg[1].zdiv;
        const int z1 = z / zdiv, z2 = z - z1 * zdiv;
        const float* __restrict__ Ag = (s0 ? g.seg[0].A : g.seg[1].A) + (long)z1 * (s0 ? g.seg[0].strideA : g.seg[1].strideA) +
                                       (long)z2 * (s0 ? g.seg[0].strideA2 : g.seg[1].strideA2);
        const float* __restrict__ Bg = (s0 ? g.seg[0].B : g.seg[1].B) + (long)z1 * (s0 ? g.seg[0].strideB : g.seg[1].strideB) +
                                       (long)z2 * (s0 ? g.seg[0].strideB2 : g.seg[1].strideB2) + (CONV ? s * g.cv_cin : 0);
        const long lda = s0 ? g.seg[0].lda : g.seg[1].lda, ldb = s0 ? g.seg[0].ldb : g.seg[1].ldb;
        const int K = CONV ? g.cv_cin : (s0 ? g.seg[0].K : g.seg[1].K);
        const int nkt = K / BK;
        const int cdy = (CONV && g.cv_ntaps == 9) ? s / 3 - 1 : 0, cdx = (CONV && g.cv_ntaps == 9) ? s - (s / 3) * 3 - 1 : 0;
        const int kvalid = min(K, max(0, (s0 ? g.seg[0].ktotal : g.seg[1].ktotal) - z2 * (s0 ? g.seg[0].kchunk : g.seg[1].kchunk)));

        // ---- per-thread source addresses for k-tile 0 and validity ----
        const float *pa0, *pa1, *pa2, *pa3, *pb0, *pb1, *pb2, *pb3;
        const float *pas0 = nullptr, *pas1 = nullptr, *pas2 = nullptr, *pas3 = nullptr;   // SHIFT: shifted-row addresses
        bool oka0 = true, oka1 = true, oka2 = true, oka3 = true;          // row guards (K-contiguous A)
        bool oks0 = true, oks1 = true, oks2 = true, oks3 = true;          // SHIFT: validity of the shifted row
#define TDX_A_ADDR_CV(i, pa, oka, cvb, cvh, cvw)                                           \
        {                                                                                  \
            const int ih = cvh + cdy, iw = cvw + cdx;                                      \
            oka = (m0 + ((i) >> 3) < g.M) && ih >= 0 && ih < g.cv_Hin && iw >= 0 && iw < g.cv_Win; \
            const int src = oka ? cvb + ih * g.cv_Win + iw : 0;                            \
            pa = Ag + (long)src * lda + ((i) & 7) * 4;                                     \
        }
#define TDX_A_ADDR(i, pa, pas, oka, oks)                                                   \
        if constexpr (!A_KMAJOR) {                                                         \
            const int m = m0 + ((i) >> 3);                                                 \
            oka = m < g.M;                                                                 \
            const int mc = min(m, g.M - 1);                                                \
            pa = Ag + (long)mc * lda + ((i) & 7) * 4;                                      \
            if constexpr (SHIFT) {                                                         \
                oks = oka && (m % g.shift_S != 0);                                         \
                pas = Ag + (long)max(mc - 1, 0) * lda + ((i) & 7) * 4;                     \
            }                                                                              \
        } else {                                                                           \
            pa = Ag + m0 + ((i) & 31) * 4;                                                 \
        }
        if constexpr (CONV) {
            TDX_A_ADDR_CV(i0, pa0, oka0, cvb0, cvh0, cvw0) TDX_A_ADDR_CV(i1, pa1, oka1, cvb1, cvh1, cvw1)
            TDX_A_ADDR_CV(i2, pa2, oka2, cvb2, cvh2, cvw2) TDX_A_ADDR_CV(i3, pa3, oka3, cvb3, cvh3, cvw3)
        } else {
            TDX_A_ADDR(i0, pa0, pas0, oka0, oks0) TDX_A_ADDR(i1, pa1, pas1, oka1, oks1)
            TDX_A_ADDR(i2, pa2, pas2, oka2, oks2) TDX_A_ADDR(i3, pa3, pas3, oka3, oks3)
        }
#define TDX_B_ADDR(i, pb)                                                                  \
        if constexpr (!B_KMAJOR) {                                                         \
            const int row = (i) >> 3;                                                      \
            const int n = PAIRED ? ((row < BN / 2) ? n0 + row : g.pair_off + n0 + row - BN / 2) : n0 + row; \
            pb = Bg + (long)n * ldb + ((i) & 7) * 4;                                       \
        } else {                                                                           \
            const int nl = ((i) & 31) * 4;                                                 \
            const int n = PAIRED ? ((nl < BN / 2) ? n0 + nl : g.pair_off + n0 + nl - BN / 2) : n0 + nl; \
            pb = Bg + n;                                                                   \
        }
        TDX_B_ADDR(i0, pb0) TDX_B_ADDR(i1, pb1) TDX_B_ADDR(i2, pb2) TDX_B_ADDR(i3, pb3)
        const int kr0 = i0 >> 5, kr1 = i1 >> 5, kr2 = i2 >> 5, kr3 = i3 >> 5;   // K-major: k row in tile

        f32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
        bool va0 = true, va1 = true, va2 = true, va3 = true, vb0 = true, vb1 = true, vb2 = true, vb3 = true;
#define TDX_LOAD_A(kt, ra, va, pa, pas, oka, oks, kr)                                      \
        if constexpr (!A_KMAJOR) {                                                         \
            const int k0_ = (kt) * BK;                                                     \
            if constexpr (SHIFT) {                                                         \
                const bool sh = k0_ < g.shift_k;                                           \
                ra = ldg4((sh ? pas : pa) + k0_); va = sh ? oks : oka;                     \
            } else { ra = ldg4(pa + k0_); va = oka; }                                      \
        } else {                                                                           \
            const int k_ = (kt) * BK + kr;                                                 \
            va = k_ < kvalid;                                                              \
            ra = ldg4(pa + (long)min(k_, kvalid - 1) * lda);                               \
        }
#define TDX_LOAD_B(kt, rb, vb, pb, kr)                                                     \
        if constexpr (!B_KMAJOR) { rb = ldg4(pb + (kt) * BK); }                            \
        else {                                                                             \
            const int k_ = (kt) * BK + kr;                                                 \
            vb = k_ < kvalid;                                                              \
            rb = ldg4(pb + (long)min(k_, kvalid - 1) * ldb);                               \
        }
#define TDX_LOAD_TILE(kt)                                                                  \
        TDX_LOAD_A(kt, ra0, va0, pa0, pas0, oka0, oks0, kr0) TDX_LOAD_A(kt, ra1, va1, pa1, pas1, oka1, oks1, kr1) \
        TDX_LOAD_A(kt, ra2, va2, pa2, pas2, oka2, oks2, kr2) TDX_LOAD_A(kt, ra3, va3, pa3, pas3, oka3, oks3, kr3) \
        TDX_LOAD_B(kt, rb0, vb0, pb0, kr0) TDX_LOAD_B(kt, rb1, vb1, pb1, kr1)             \
        TDX_LOAD_B(kt, rb2, vb2, pb2, kr2) TDX_LOAD_B(kt, rb3, vb3, pb3, kr3)

#define TDX_LOAD_PART0(kt) TDX_LOAD_A(kt, ra0, va0, pa0, pas0, oka0, oks0, kr0) TDX_LOAD_B(kt, rb0, vb0, pb0, kr0)
#define TDX_LOAD_PART1(kt) TDX_LOAD_A(kt, ra1, va1, pa1, pas1, oka1, oks1, kr1) TDX_LOAD_B(kt, rb1, vb1, pb1, kr1)
#define TDX_LOAD_PART2(kt) TDX_LOAD_A(kt, ra2, va2, pa2, pas2, oka2, oks2, kr2) TDX_LOAD_B(kt, rb2, vb2, pb2, kr2)
#define TDX_LOAD_PART3(kt) TDX_LOAD_A(kt, ra3, va3, pa3, pas3, oka3, oks3, kr3) TDX_LOAD_B(kt, rb3, vb3, pb3, kr3)
        TDX_LOAD_TILE(0)
        for (int kt = 0; kt < nkt; ++kt) {
            if (VARIANT != 2) __syncthreads();          // previous tile fully consumed
            *reinterpret_cast<f32x4*>(As + sa0) = sel4(va0, ra0);
            *reinterpret_cast<f32x4*>(As + sa1) = sel4(va1, ra1);
            *reinterpret_cast<f32x4*>(As + sa2) = sel4(va2, ra2);
            *reinterpret_cast<f32x4*>(As + sa3) = sel4(va3, ra3);
            *reinterpret_cast<f32x4*>(Bs + sb0) = sel4(vb0, rb0);
            *reinterpret_cast<f32x4*>(Bs + sb1) = sel4(vb1, rb1);
            *reinterpret_cast<f32x4*>(Bs + sb2) = sel4(vb2, rb2);
            *reinterpret_cast<f32x4*>(Bs + sb3) = sel4(vb3, rb3);
            if (VARIANT != 2) __syncthreads();
            // next tile's loads are spread over the MFMA phase (2 x 16 B per lane after each group of
            // 16 MFMAs) instead of a 32 KB burst per block at the tile boundary
            const bool more = VARIANT != 1 && kt + 1 < nkt;
            if (VARIANT == 3 && more) { TDX_LOAD_TILE(kt + 1) }
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                f32x4 a0, a1, b0, b1;
                if constexpr (!A_KMAJOR) {
                    a0 = *reinterpret_cast<const f32x4*>(As + ml0 * P + kc * 8 + h * 4);
                    a1 = *reinterpret_cast<const f32x4*>(As + ml1 * P + kc * 8 + h * 4);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a0[j] = As[(kc * 8 + h * 4 + j) * BM + ml0];
                        a1[j] = As[(kc * 8 + h * 4 + j) * BM + ml1];
                    }
                }
                if constexpr (!B_KMAJOR) {
                    b0 = *reinterpret_cast<const f32x4*>(Bs + nl0 * P + kc * 8 + h * 4);
                    b1 = *reinterpret_cast<const f32x4*>(Bs + nl1 * P + kc * 8 + h * 4);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        b0[j] = Bs[(kc * 8 + h * 4 + j) * BN + nl0];
                        b1[j] = Bs[(kc * 8 + h * 4 + j) * BN + nl1];
                    }
                }
                // column tiles beyond N (narrow convolutions: N = 32 / 64 / 96 of the 128-wide tile) are skipped
                // (wave-uniform): their results are never stored
                if (nv0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc00, 0, 0, 0);
                        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc10, 0, 0, 0);
                    }
                }
                if (nv1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc01, 0, 0, 0);
                        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc11, 0, 0, 0);
                    }
                }
                if (VARIANT != 3 && more) {
                    const int ktn = VARIANT == 4 ? (kt & 1) : kt + 1;     // 4: re-read k-tiles 0/1 (L1/L2-hot) — timing only
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc == 0) { TDX_LOAD_PART0(ktn) }
                    else if (kc == 1) { TDX_LOAD_PART1(ktn) }
                    else if (kc == 2) { TDX_LOAD_PART2(ktn) }
                    else { TDX_LOAD_PART3(ktn) }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();   // before the next segment overwrites LDS
    }
#undef TDX_ST_A
#undef TDX_ST_B
#undef TDX_A_ADDR
#undef TDX_A_ADDR_CV
#undef TDX_B_ADDR
#undef TDX_LOAD_A
#undef TDX_LOAD_B
#undef TDX_LOAD_TILE
#undef TDX_LOAD_PART0
#undef TDX_LOAD_PART1
#undef TDX_LOAD_PART2
#undef TDX_LOAD_PART3

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h ----
    // per-column constants (bias, gain ...) are fetched once per lane; per-row constants and the
    // per-element auxiliary loads are fetched for 16 rows before the first store of the batch
    // (a tile whose rows all exist takes a branch-free path: per-row `if (m < M)` blocks make the compiler re-wait vmcnt(0)
    // at every block entry, and vmcnt counts stores — one store round trip per store)
    auto epilogue = [&](auto tag) {
    constexpr bool CHECK = decltype(tag)::value;
    auto c0 = epi.col(z, n0 + nl0);
    auto c1 = epi.col(z, PAIRED ? n0 + nl0 : n0 + nl1);
    touch(c0); touch(c1);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        decltype(epi.row(0, 0)) rw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1);
            rw[r] = epi.row(z, m);
        }
        if constexpr (epi_has_aux<Epi>::value) {
            decltype(epi.aux(0, 0, 0, rw[0])) ax0[16], ax1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1);
                ax0[r] = epi.aux(z, m, n0 + nl0, rw[r]);
                if constexpr (!PAIRED) ax1[r] = epi.aux(z, m, n0 + nl1, rw[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { touch(rw[r]); touch(ax0[r]); if constexpr (!PAIRED) touch(ax1[r]); }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v0 = tm == 0 ? acc00[r] : acc10[r];
                const float v1 = tm == 0 ? acc01[r] : acc11[r];
                if (!CHECK || m < g.M) {
                    if constexpr (PAIRED) {
                        epi.store2(z, m, n0 + nl0, v0, v1, rw[r], c0, ax0[r]);
                    } else {
                        epi.store(z, m, n0 + nl0, v0, rw[r], c0, ax0[r]);
                        epi.store(z, m, n0 + nl1, v1, rw[r], c1, ax1[r]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) touch(rw[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v0 = tm == 0 ? acc00[r] : acc10[r];
                const float v1 = tm == 0 ? acc01[r] : acc11[r];
                if (!CHECK || m < g.M) {
                    if constexpr (PAIRED) {
                        epi.store2(z, m, n0 + nl0, v0, v1, rw[r], c0);
                    } else {
                        epi.store(z, m, n0 + nl0, v0, rw[r], c0);
                        epi.store(z, m, n0 + nl1, v1, rw[r], c1);
                    }
                }
            }
        }
    }
    };
    if (m0 + GEMM_BM <= g.M) epilogue(std::false_type{}); else epilogue(std::true_type{});
}

template <bool A_KMAJOR, bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi, int VARIANT = 0, bool CONV = false>
inline hipError_t launch_gemm(GemmArgs g, int batches, Epi epi, hipStream_t st) {
    g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    g.tiles_n = PAIRED ? g.N / (GEMM_BN / 2) : g.N / GEMM_BN;
    g.batches = batches;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        // N-tiles per sweep: keep the weight slice (gw x 128 rows x Ktot floats) under ~1.5 MB of the 4 MB L2
        long ktot = 0;
        for (int i = 0; i < g.nseg; ++i) ktot += g.seg[i].K;
        if (CONV) ktot = (long)g.cv_ntaps * g.cv_cin;
        long gw = (1536L * 1024) / (128L * ktot * 4);
        if (gw < 1) gw = 1;
        if (gw > g.tiles_n) gw = g.tiles_n;
        g.gw = (int)gw;
        grid = dim3(8 * g.mp * g.tiles_n, batches, 1);
    } else {
        g.map_mode = 1;
        g.mp = 0; g.gw = 1;
        grid = dim3(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    }
    hipLaunchKernelGGL((gemm_f32_kernel<A_KMAJOR, B_KMAJOR, PAIRED, SHIFT, Epi, VARIANT, CONV>), grid, dim3(GEMM_THREADS), 0, st, g, epi);
    return hipGetLastError();
}

inline GemmSeg make_seg(const float* A, long lda, const float* B, long ldb, int K, long strideA = 0,
                        long strideB = 0, int kvalid = -1) {
    GemmSeg s;
    s.A = A; s.B = B; s.lda = lda; s.ldb = ldb; s.K = K; s.strideA = strideA; s.strideB = strideB;
    s.zdiv = 1; s.strideA2 = 0; s.strideB2 = 0; s.kchunk = 0; s.ktotal = kvalid < 0 ? K : kvalid;
    return s;
}

inline GemmArgs make_args(int M, int N, GemmSeg s0) {
    GemmArgs g{};
    g.seg[0] = s0; g.seg[1] = s0; g.nseg = 1; g.M = M; g.N = N; g.n_valid = N;
    g.pair_off = 0; g.shift_k = 0; g.shift_S = 1;
    return g;
}

}  // namespace tdx
