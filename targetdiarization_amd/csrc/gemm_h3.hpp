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
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, 4 per wave per
// k-tile) into a ring of FOUR 32 KB stages; the DMA of tile t+4 is issued during the MFMAs of
// tile t — into the buffer tile t itself just left for the registers — and waited for with a
// COUNTED vmcnt (never 0 in steady state) before the raw s_barrier that opens stage t+3: three
// stage times of latency tolerance.  The operands mostly hit L2, but a quarter of the requests
// go to the Infinity Cache / HBM (~3 us under load) and the vmcnt is in order, so the stage time
// settles at (miss latency) / (stages of look-ahead): 1.5 us with two stages of look-ahead.  Fragments of tile t+1 are read from LDS during the MFMAs of tile t (A
// fragments into the registers the finished row of MFMA tiles just released, B fragments into
// a second register set), so a stage is
//     [vmcnt(8); lgkmcnt(0); s_barrier]  4 x { 6 MFMA ; fragment reads ; 1 LDS-DMA }
// with nothing exposed but the barrier itself.
//
// Two operand formats, chosen per operand by A_TR / B_TR:
//  * row-major planes (above): one operand row = one matrix row, k contiguous.  LDS image: row
//    pitch 64 B (16 k x 2 planes x 2 B), the 16-B slot q of row r stored at slot q ^ ((r>>2)&3),
//    read with ds_read_b128 (conflict-free over its four 16-lane groups).
//  * K-major planes ("TR"): the matrix is stored [k][n/32][2 planes][32] f16 — k is the
//    slow index, as activations indexed by token are; hi and lo of 32 columns share one 128-B
//    line, so an element-wise consumer (the attention gate) reads full lines — with ONE scale for
//    the whole matrix (a static bound, see mf2.hip).  LDS image per k row: 1 KiB = [blk 0: hi 256 B | lo 256 B]
//    [blk 1: ...], the 64-B windows of a 256-B plane row permuted by window ^ (k&3); fragments
//    are gathered with ds_read_b64_tr_b16 (4 k x 16 n per 16-lane group, two per fragment),
//    conflict-free per 32-lane half.
// In both cases the swizzle is applied on the per-lane SOURCE address of the DMA (its LDS
// destination is lane-linear) and on the fragment read.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <typeinfo>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devutil.hpp"
#include "gemm.hpp"

namespace tdx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H3_BM = 256, H3_BN = 256, H3_BK = 16, H3_THREADS = 512;
constexpr int H3_ROWB = 64;                   // bytes per LDS row (16 k x 2 planes x 2 B)
constexpr int H3_OPER = 256 * H3_ROWB;        // 16 KB per operand tile
constexpr int H3_STAGE = 2 * H3_OPER;         // A then B
constexpr int H3_NBUF = 4;
constexpr int H3_AUX_AHEAD = 3;               // PAIRED epilogues: half row blocks of aux() operands requested ahead of the stores
constexpr int H3_LDS = H3_NBUF * H3_STAGE;    // 128 KB
constexpr int H3_LDS_EXTRA = 12288;           // behind the ring: TWOSEG row factors, row scales, row() values, segment factors (7 x 1 KB)

struct H3Seg {
    const unsigned char* A;      // planes of the A operand
    const unsigned char* B;      // planes of the B operand
    const float* sa;             // row scales of A   (element m * sa_mul: sa_mul = 0 -> one scale for all rows)
    const float* sb;             // column scales of B (element n * sb_mul)
    long lda, ldb;               // bytes: row-major planes: pitch of a matrix row; K-major: pitch of a k row
    // batch z -> z1 = z / zdiv, z2 = z % zdiv ; pointer += z1*stride + z2*stride2
    long strideA, strideB, strideA2, strideB2;          // bytes
    long strideSA, strideSB, strideSA2, strideSB2;      // floats
    int sa_mul, sb_mul;
    int K;                       // multiple of 16
    int zdiv;
    // split-K over a K-major operand: batch z2 covers k rows [z2*kchunk, z2*kchunk + K) clipped to ktotal
    // (kchunk = 0: unused); ktotal a multiple of 16
    int kchunk, ktotal;
    // segmented row scales (plain row-major mode, one segment): the A planes carry one row scale per K segment of `segk`
    // values (segk % 64 == 0, K % segk == 0; 0 = one scale per row): segment j's scales are sa + j*strideSeg.  The
    // accumulators change their scale domain at the segment boundaries (exact powers of two), like TWOSEG.
    int segk;
    long strideSeg;
    // row-major A: operand row of tile row m is m + a_shift; with a_period > 0 the rows with m % a_period == 0 read
    // a_zero (>= 4*K zero bytes) instead — the token shift of FLASH (first half of the channels from token s-1, zero at
    // the first token of a sample: mossformer_block.py:204-207) as the first K segment of a TWOSEG launch
    int a_shift, a_period;
    const unsigned char* a_zero;
};

struct H3Args {
    H3Seg seg[2];
    int nseg;
    int M, N;                    // valid rows / columns (PAIRED: N = number of pair columns, multiple of 128)
    int tiles_m, tiles_n;
    int map_mode, mp, gw, batches;
    int mc;                      // map_mode 0: M tiles per chunk of an XCD's range (0 = the whole range), see the block -> tile map
    int pair_off;
    // A_CONV (implicit GEMM over NHWC pixel rows): output row m = (b, h, w) of [B, Hout, Wout]; K segment `tap` (cv_cin
    // values) reads the planes row of input pixel (b, h*stride+dy, w*stride+dx), (dy,dx) = (tap/3-1, tap%3-1) for 9
    // taps or (0,0) for 1, or `zero_row` (>= 4*cv_cin zero bytes) outside the image; B = [N][ntaps*cv_cin] planes.
    int cv_Hin, cv_Win, cv_Hout, cv_Wout, cv_stride, cv_ntaps, cv_cin;
    int cv_taps_total;           // > 0: the taps are split over the batches (split-K for few output rows): batch z covers taps
                                 // [z*cv_ntaps, (z+1)*cv_ntaps) of cv_taps_total; seg[0].strideB = 4*cv_ntaps*cv_cin, K = cv_ntaps*cv_cin
    const unsigned char* zero_row;
    int dbg;                     // diagnostics (env TDX_H3_DEBUG, timing only, wrong results): 1 = no epilogue
    unsigned long long* stamps;  // diagnostics (builds with -DTDX_H3_STAMPS only): 8 x u64 per block, see H3_STAMP
};

// in-kernel time stamps (diagnostic builds): wave 0 lane 0 of every block records the 100 MHz wall clock at phase boundaries
#ifdef TDX_H3_STAMPS
#define H3_STAMP(i) do { if (g.stamps && threadIdx.x == 0) g.stamps[(long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define H3_STAMP_HW() do { if (g.stamps && threadIdx.x == 0) g.stamps[(long)blockIdx.x * 8 + 7] = \
    (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } while (0)
#else
#define H3_STAMP(i) do {} while (0)
#define H3_STAMP_HW() do {} while (0)
#endif

inline H3Seg h3_seg(const void* A, const float* sa, long lda, const void* B, const float* sb, long ldb, int K) {
    H3Seg s{};
    s.A = (const unsigned char*)A; s.B = (const unsigned char*)B; s.sa = sa; s.sb = sb; s.lda = lda; s.ldb = ldb; s.K = K; s.zdiv = 1;
    s.sa_mul = 1; s.sb_mul = 1;
    return s;
}

__device__ __forceinline__ void h3_glds16(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

typedef __fp16 h3_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// fragment (32 operand rows x 16 k, one plane) for lane (row l31, k = 8h .. 8h+7)
template <bool TR>
__device__ __forceinline__ f16x8 h3_frag(const unsigned char* p) {
    if constexpr (!TR) {
        return *reinterpret_cast<const f16x8*>(p);
    } else {
        // p = address of (k row 8h+q, 4 columns) for this lane; the second gather is 4 k rows (4 KiB) further
        const h3_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h3_fp16x4*)p);
        const h3_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h3_fp16x4*)(p + 4096));
        f16x8 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i] = (_Float16)lo[i]; r[4 + i] = (_Float16)hi[i]; }
        return r;
    }
}

// ds_read_b64_tr_b16 through INLINE ASM.  The compiler's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of the builtin form
// whenever an LDS-DMA is outstanding (it cannot tell that the read does not touch the DMA's destination; plain ds_read_b128
// loads are disambiguated, the builtin is not): the ring then waits, every stage, for the DMA it has just issued.  The asm
// form is invisible to that pass — and to its lgkmcnt bookkeeping: the raw halves must go through h3_tr_fence (an
// `s_waitcnt lgkmcnt(0)` tied to the registers) before anything reads them, and no instruction may touch them in between
// (tools/asm_tr_hazard.py checks the ISA).  LDS returns in order, so the compiler's own lgkmcnt counts stay conservative.
struct H3TrRaw { uint2 lo, hi; };         // k = 8h .. 8h+3 | 8h+4 .. 8h+7 of the lane's column
__device__ __forceinline__ unsigned h3_lds_addr(const unsigned char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
template <int OFF>
__device__ __forceinline__ void h3_tr_read(H3TrRaw& r, unsigned a) {   // a + OFF = LDS address of (k row 8h+q, 4 columns) for this lane
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.lo) : "v"(a), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.hi) : "v"(a), "n"(OFF + 4096) : "memory");
}
__device__ __forceinline__ f16x8 h3_tr_value(const H3TrRaw& r) {       // (after the fence)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f16x8, u4{r.lo.x, r.lo.y, r.hi.x, r.hi.y});
}
template <int N>
__device__ __forceinline__ void h3_tr_fence(H3TrRaw (&r)[N]) {
    static_assert(N == 2 || N == 4 || N == 8, "fence over 2, 4 or 8 raw fragments");
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0].lo), "+v"(r[0].hi), "+v"(r[1].lo), "+v"(r[1].hi), "+v"(r[2].lo), "+v"(r[2].hi), "+v"(r[3].lo), "+v"(r[3].hi),
                     "+v"(r[4].lo), "+v"(r[4].hi), "+v"(r[5].lo), "+v"(r[5].hi), "+v"(r[6].lo), "+v"(r[6].hi), "+v"(r[7].lo), "+v"(r[7].hi) :: "memory");
    else if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0].lo), "+v"(r[0].hi), "+v"(r[1].lo), "+v"(r[1].hi), "+v"(r[2].lo), "+v"(r[2].hi), "+v"(r[3].lo), "+v"(r[3].hi) :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0].lo), "+v"(r[0].hi), "+v"(r[1].lo), "+v"(r[1].hi) :: "memory");
}

// Per-lane fragment addressing inside a stage buffer.
//   row-major planes: A tile tm, plane pl at  a0 + tm*2048 + apl[pl];  B tile tn at  b0 + tn*8192 + bpl[pl]
//   K-major planes  : A tile tm, plane pl at  at[tm] + pl*256;          B tile tn at  b0 + tn*512 + pl*256
struct H3Frag { int a0, apl[2], at[4], b0, bpl[2]; };

template <bool A_TR>
__device__ __forceinline__ const unsigned char* h3_addr_a(const unsigned char* st, const H3Frag& f, int tm, int pl) {
    if constexpr (A_TR) return st + f.at[tm] + pl * 256;
    else return st + f.a0 + tm * 2048 + f.apl[pl];
}
template <bool B_TR>
__device__ __forceinline__ const unsigned char* h3_addr_b(const unsigned char* st, const H3Frag& f, int tn, int pl) {
    if constexpr (B_TR) return st + f.b0 + tn * 512 + pl * 256;
    else return st + f.b0 + tn * 8192 + f.bpl[pl];
}

// one pipeline stage: MFMAs of the tile whose fragments are in (ah, al, bh, bl); with NEXT the
// fragments of the following tile are read from stage buffer `nxt`; with ISSUE the wave's four
// LDS-DMA pieces of the tile three ahead are issued (src = gp[j] + goff, dst = dmad + j KiB).
// PH: the half of the stage in which this wave issues its DMA (the two waves of a SIMD differ).
template <bool A_TR, bool B_TR, bool FULLN, bool NEXT, bool ISSUE, int PH, bool M16 = false>
__device__ __forceinline__ void h3_stage(f32x16 (&acc)[4][2], f16x8 (&ah)[4], f16x8 (&al)[4], f16x8 (&bh)[2], f16x8 (&bl)[2],
                                         const unsigned char* nxt, const H3Frag& f,
                                         const unsigned char* const (&gp)[4], long goff, unsigned char* dmad) {
    // (K-major operands: the builtin transposing reads are preceded by a compiler-inserted s_waitcnt vmcnt(0) — see h3_tr_read;
    //  the asm form was measured here too: -2 % on the attention launch, +30 % on the M = 128 lin_k^T [v|u] launch, whose four
    //  computing waves lose the pinned issue order.  The model's attention launch runs on gemm_h3a.hpp, which uses the asm form.)
    f16x8 nbh[2], nbl[2];
    if constexpr (NEXT) {
#pragma unroll
        for (int tn = 0; tn < (FULLN ? 2 : 1); ++tn) {
            nbh[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(nxt, f, tn, 0));
            nbl[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(nxt, f, tn, 1));
        }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < (FULLN ? 2 : 1); ++tn) {
            if constexpr (!M16) {
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            } else {
                // VARIANT 6, TIMING ONLY (wrong results): the same flops, operand registers and accumulators as six
                // v_mfma_f32_16x16x32_f16 (16 x 16 x 32 = half the flops of 32 x 32 x 16) on the four 16 x 16 quarters
                f32x4* q = reinterpret_cast<f32x4*>(&acc[tm][tn]);
                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bl[tn], q[0], 0, 0, 0);
                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], q[1], 0, 0, 0);
                q[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bh[tn], q[2], 0, 0, 0);
                q[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bl[tn], q[3], 0, 0, 0);
                q[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[tm], bh[tn], q[0], 0, 0, 0);
                q[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[tm], bl[tn], q[1], 0, 0, 0);
            }
            if constexpr (ISSUE) {
                if ((tm >> 1) == PH) h3_glds16(gp[(tm & 1) * 2 + tn] + goff, dmad + ((tm & 1) * 2 + tn) * 1024);
            }
        }
        if constexpr (ISSUE && !FULLN) {
            if ((tm >> 1) == PH) h3_glds16(gp[(tm & 1) * 2 + 1] + goff, dmad + ((tm & 1) * 2 + 1) * 1024);
        }
        if constexpr (NEXT) {
            ah[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(nxt, f, tm, 0));
            al[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(nxt, f, tm, 1));
        }
    }
    if constexpr (NEXT) {
#pragma unroll
        for (int tn = 0; tn < (FULLN ? 2 : 1); ++tn) { bh[tn] = nbh[tn]; bl[tn] = nbl[tn]; }
    }
    // pin the issue order (the scheduler would otherwise cluster the DMA and the reads at the end
    // of the stage, where both waves of a SIMD would stall on them together)
    constexpr int RA = A_TR ? 4 : 2, RB = (B_TR ? 4 : 2) * (FULLN ? 2 : 1);
    if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, RB, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const bool dma = ISSUE && (tm >> 1) == PH;
        constexpr int NM = M16 ? 6 : 3;
        if constexpr (FULLN) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
        }
        if constexpr (NEXT) __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);
    }
}

// a wave whose 128 rows all lie beyond M (M <= 128 launches: lin_k^T [v|u]) only keeps feeding the ring
template <bool ISSUE>
__device__ __forceinline__ void h3_stage_dma_only(const unsigned char* const (&gp)[4], long goff, unsigned char* dmad) {
    if constexpr (ISSUE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + goff, dmad + j * 1024);
    }
}

#define H3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define H3_BARRIER()                                   \
    do {                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's fragment reads of the previous stage are done:    \
                                                              the buffer they came from may be refilled after the barrier */ \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)

// A_TR / B_TR: operand in K-major planes.  TWOSEG: two K segments (same operand formats, their own
// pointers and scales).  VARIANT != 0: timing-only diagnostics (wrong results): 1 no LDS-DMA in
// the loop, 2 no fragment reads, 3 neither, 4 no barrier, 9 no vmcnt wait in the steady loop.
// optional functor member full(z, m0): all 256 rows of the tile are real rows (row() >= 0 for the functors whose row() marks
// padding with -1) — lets the epilogue take its branch-free path
template <class E, class = void> struct epi_has_full { static constexpr bool value = false; };
template <class E> struct epi_has_full<E, decltype((void)&E::full, void())> { static constexpr bool value = true; };
// optional functor members ptr(z, m, n) -> float* (address of output element (m, n) of a single row-major output), ldm()
// (its row pitch in elements) and put(p, v, row, col) (= store() to that address): see the epilogue
template <class E, class = void> struct epi_has_ptr { static constexpr bool value = false; };
template <class E> struct epi_has_ptr<E, decltype((void)&E::ptr, void())> { static constexpr bool value = true; };
// optional scale folding: rowmul(row) is multiplied into the LDS-staged row scale, colmul(col) into the column scale (then
// put_scaled(p, v, row, col) receives v with both factors in); PAIRED: pairmul(col) -> float2 into the two column scales
// (then store2_scaled(...)).  Saves the per-element multiplies of functors whose output is scale-linear in v.
template <class E, class = void> struct epi_has_rowmul { static constexpr bool value = false; };
template <class E> struct epi_has_rowmul<E, decltype((void)&E::rowmul, void())> { static constexpr bool value = true; };
template <class E, class = void> struct epi_has_pairmul { static constexpr bool value = false; };
template <class E> struct epi_has_pairmul<E, decltype((void)&E::pairmul, void())> { static constexpr bool value = true; };
// ---- split helpers (used by the PLOUT epilogue below and by the producers at the end of this file) ----
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
// Each lane splits its OWN four values (packed converts: 3 VALU per value), then the lanes of a pair swap half of the result.
struct H3Pair { uint2 hi, lo; };      // 4 x f16 each
__device__ __forceinline__ H3Pair h3_split4(float4 v, float s) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const float xs[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    h4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const _Float16 t = (_Float16)xs[j]; a[j] = t; b[j] = (_Float16)(xs[j] - (float)t); }
    H3Pair r;
    r.hi = __builtin_bit_cast(uint2, a); r.lo = __builtin_bit_cast(uint2, b);
    return r;
}
__device__ __forceinline__ uint4 h3_pair16(int q, float4 v, float s) {
    // the 16 B this lane stores: even lane = hi of the chunk's 8 k (own 4, partner's 4), odd lane = lo (partner's 4, own 4)
    const H3Pair o = h3_split4(v, s);
    const bool odd = q & 1;
    const uint2 send = odd ? o.hi : o.lo;
    uint2 recv;
    recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);       // lanes 2j <-> 2j+1
    recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
    return odd ? make_uint4(recv.x, recv.y, o.lo.x, o.lo.y) : make_uint4(o.hi.x, o.hi.y, recv.x, recv.y);
}
__device__ __forceinline__ void h3_emit4(unsigned char* prow, int q, float4 v, float s) {
    *reinterpret_cast<uint4*>(prow + (q >> 1) * 32 + ((q & 1) ? 16 : 0)) = h3_pair16(q, v, s);
}
// the same with the store predicated per lane (the lane exchange runs for every lane): rows handled by half-waves
__device__ __forceinline__ void h3_emit4_pred(unsigned char* prow, int q, float4 v, float s, bool ok) {
    const uint4 w = h3_pair16(q, v, s);
    if (ok) *reinterpret_cast<uint4*>(prow + (q >> 1) * 32 + ((q & 1) ? 16 : 0)) = w;
}
__device__ __forceinline__ float h3_wave_max(float v) {
    v = h3_row16_max(v);
    return fmaxf(fmaxf(h3_lane(v, 0), h3_lane(v, 16)), fmaxf(h3_lane(v, 32), h3_lane(v, 48)));
}
__device__ __forceinline__ float h3_absmax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }


// PLANES-OUT epilogues ("PLOUT"): the functor's values are not stored as fp32 by the lanes that hold the accumulators but
// staged through the (then idle) LDS ring, re-read one tile row per wave and written as split-f16 ROW-MAJOR planes with an
// exact power-of-two scale per (row, tile-wide column segment) [+ the segment's sum of squares] — the A operand format of
// the next GEMM, without the fp32 round trip through HBM and the separate split pass.  Functor members:
//     H3PlOut plout(int z)                     destination of batch z
//     long prow(int z, int m)                  destination row of tile row m, or -1 (row skipped: group padding)
//     float val(z, m, n, v, row, col[, aux])   value of element (m, n) (may also store fp32 side outputs); non-PAIRED
//     float val2_scaled(z, m, c, av, au, row, col, aux)   PAIRED (with pairmul)
// Column segment j = tile column index: 256 columns (128 PAIRED) -> planes bytes [j*1024, (j+1)*1024) ([j*512, ...) PAIRED)
// of the destination row, scale[j*seg_stride + prow], ss[j*seg_stride + prow].
struct H3PlOut { unsigned char* planes; long pitch; float* scale; float* ss; long seg_stride; };
template <class E, class = void> struct epi_has_plout { static constexpr bool value = false; };
template <class E> struct epi_has_plout<E, decltype((void)&E::plout, void())> { static constexpr bool value = true; };
// optional PLOUT functor member rowout(z, prow, n0) -> float* : the same values ALSO as fp32, written in phase 2 from the staged tile
// (16 B per lane, full lines) instead of by store instructions inside val().  W2 per launch at config 2: stores inside val() 433 us,
// rowout 418 us; the residual added in phase 2 from full-line reads instead of aux() 440 us (and a build of that variant with a fully
// unrolled phase 2 ran the whole launch 4x slower, k loop included — cause not isolated; the shipped kernels miss the instruction
// cache on <= 0.03 % of their fetches, profiles/r02_pmc_icache.log)
template <class E, class = void> struct epi_has_rowout { static constexpr bool value = false; };
template <class E> struct epi_has_rowout<E, decltype((void)&E::rowout, void())> { static constexpr bool value = true; };
// FUSED DEPTHWISE CONVOLUTION epilogue ("CONV"): FFConvM = Linear -> SiLU -> y + dwconv17(y) along the tokens
// (conv_module.py:180-220) without the fp32 round trip of y through HBM.  The M tiles of such a launch OVERLAP: tile bm
// computes the GEMM rows [240 bm - 8, 240 bm + 248) and owns the 240 output rows in the middle, the 8 rows on either side
// are the convolution's halo (recomputed: 256 / 240 = +6.7 % MFMA work).  Per 128-column half of the tile:
//   phase 1  every lane: y = silu(acc * scales + bias) -> staging tile T[256][128] fp32 in the (then idle) ring
//   phase 2  wave w owns output rows 8 + 30 w .. + 29, lane l the column pair (2l, 2l+1): sliding 17-row window over T
//            (ds_read_b64, conflict-free), 17 packed FMAs per output row, then the split with the layer's STATIC scale and
//            stores of the K-major planes ([row][32-column group][hi 64 B | lo 64 B]: 4 B per lane = (c, c+1) of one plane,
//            64-B runs) and, for the columns >= u0, of the fp32 copy (8 B per lane, 512-B runs).
// Rows of another sample (the convolution zero-pads at the ends of a sample) and rows outside [0, M) are masked in a slow path
// that only the tiles containing a sample boundary take (one in S / 240).  Columns >= conv().ncols (to_qk) are written as
// plain fp32 by the owner rows.  Functor: the members of a ptr()/put_scaled() functor plus conv() -> H3Conv and act(x).
struct H3Conv {
    const float* w; int wld;                 // taps, tap-major [17][wld]
    int ncols;                               // columns [0, ncols) take the convolution (ncols % 256 == 0)
    unsigned char* planes; long pitch;       // K-major planes: row pitch in bytes (4 * channels); pad rows are the caller's
    float sv;                                // static scale multiplier (a bound: no row maximum)
    float* u32; int u0; long ld32;           // fp32 copy of the columns >= u0: u32[row * ld32 + (c - u0)], row = global row m
    int S, Sp;                               // rows per sample (conv zero padding at its ends); planes rows per sample
};
template <class E, class = void> struct epi_has_conv { static constexpr bool value = false; };
template <class E> struct epi_has_conv<E, decltype((void)&E::conv, void())> { static constexpr bool value = true; };
constexpr int H3_CONV_STEP = 240, H3_CONV_HALO = 8;
template <class T> __device__ __forceinline__ void h3_assume_row(const T&) {}
__device__ __forceinline__ void h3_assume_row(long rw) { __builtin_assume(rw >= 0); }

template <bool A_TR, bool B_TR, bool PAIRED, bool TWOSEG, class Epi, int VARIANT = 0, bool A_CONV = false>
__global__ __launch_bounds__(H3_THREADS, 2) void gemm_h3_kernel(H3Args g, Epi epi) {
    static_assert(!A_CONV || (!A_TR && !TWOSEG), "A_CONV: row-major A planes, one weight segment");
    constexpr bool PINGPONG = VARIANT == 5;
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
            // the XCD's M range in chunks of g.mc M tiles (0: one chunk): all N groups sweep a chunk before the next chunk starts, so
            // that the A rows a later group re-reads are still in the Infinity Cache (one sweep of the whole range evicts them)
            int i2 = i, mbase = 0, mcur = g.mp;
            if (g.mc > 0) { const int tpc = g.mc * g.tiles_n, c = i / tpc; i2 = i - c * tpc; mbase = c * g.mc; mcur = min(g.mc, g.mp - mbase); }
            const int per = mcur * g.gw, ngf = g.tiles_n / g.gw;
            const int p = i2 / per;
            int lm, n;
            if (p < ngf) { const int j = i2 - p * per; lm = j / g.gw; n = p * g.gw + (j - lm * g.gw); }
            else { const int rem = g.tiles_n - ngf * g.gw; const int j = i2 - ngf * per; lm = j / rem; n = ngf * g.gw + (j - lm * rem); }
            bm = x * g.mp + mbase + lm; bn = n;
            if (bm >= g.tiles_m) return;
        } else {
            const int tpb = g.tiles_m * g.tiles_n;
            const int zb = i / tpb, tt = i - zb * tpb;
            z = zb * 8 + x;
            if (z >= g.batches) return;
            bm = tt / g.tiles_n; bn = tt - bm * g.tiles_n;
        }
    }
    constexpr bool CONV = epi_has_conv<Epi>::value;
    static_assert(!CONV || (!PAIRED && !A_TR && !B_TR && !A_CONV && epi_has_ptr<Epi>::value && epi_has_rowmul<Epi>::value), "CONV: row-major Linear with a put_scaled() functor");
    const int m0 = CONV ? bm * H3_CONV_STEP - H3_CONV_HALO : bm * H3_BM;      // (CONV: overlapping tiles, m0 = -8 for the first)
    const int n0 = PAIRED ? bn * (H3_BN / 2) : bn * H3_BN;
    H3_STAMP(0); H3_STAMP_HW();

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging role: waves 0-3 fetch the A tile, waves 4-7 the B tile, four 1-KiB pieces each per stage.
    //   row-major planes: piece j = LDS rows (wave&3)*64 + 16j .. +15, lane -> row +(lane>>2), slot lane&3,
    //                     source slot (lane&3) ^ ((row>>2)&3)
    //   K-major planes  : piece j = k row (wave&3)*4 + j, lane -> blk lane>>5, plane (lane>>4)&1, slot lane&15,
    //                     source slot (lane&15) ^ (j<<2)         ((k&3) = j)
    const bool stB = wave >= 4;
    const int sdst = (stB ? H3_OPER : 0) + (wave & 3) * 4096;
    H3Frag f;
    {
        const int fl = (l31 >> 2) & 3;                       // row-major: slot swizzle of the lane's row
        const int q = (lane >> 2) & 3, p4 = lane & 3, nb = (lane >> 4) & 1;     // K-major gather roles
        f.a0 = (wm * 128 + l31) * H3_ROWB;
        f.apl[0] = ((2 * h) ^ fl) << 4; f.apl[1] = ((2 * h + 1) ^ fl) << 4;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) f.at[tm] = (8 * h + q) * 1024 + wm * 512 + (((tm ^ q) << 6) + nb * 32 + p4 * 8);
        if constexpr (B_TR) f.b0 = H3_OPER + (8 * h + q) * 1024 + (((wn ^ q) << 6) + nb * 32 + p4 * 8);
        else f.b0 = H3_OPER + (wn * 32 + l31) * H3_ROWB;
        f.bpl[0] = f.apl[0]; f.bpl[1] = f.apl[1];
    }
    const bool full_n = PAIRED || (n0 + 128 < g.N);
    const bool wave_on = m0 + wm * 128 < g.M;       // (wave-uniform) this wave's 128 rows exist

    // per-lane DMA source pointers of a segment's first k-tile (gp) and their advance per k-tile (gstep)
    auto setup = [&](const H3Seg& sg, const unsigned char* (&gp)[4], long& gstep) {
        const int z1 = z / sg.zdiv, z2 = z - z1 * sg.zdiv;
        if (!stB) {
            const unsigned char* Ag = sg.A + (long)z1 * sg.strideA + (long)z2 * sg.strideA2;
            if constexpr (!A_TR) {
                gstep = H3_ROWB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (wave & 3) * 64 + (lane >> 2) + 16 * j;
                    const int mr = min(max(m0 + r, 0), g.M - 1);
                    const unsigned char* rowp = Ag + (long)(mr + sg.a_shift) * sg.lda;
                    if (sg.a_period && (mr % sg.a_period) == 0) rowp = sg.a_zero;
                    gp[j] = rowp + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
                }
            } else {
                gstep = 16 * sg.lda;
                const int blk = min(m0 / 128 + (lane >> 5), (g.M - 1) / 128);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    {
                    const int ch = (lane & 15) ^ (j << 2);       // logical 16-B chunk (8 columns) of the 128-column block
                    gp[j] = Ag + (long)((wave & 3) * 4 + j) * sg.lda + blk * 512 + (ch >> 2) * 128 + ((lane >> 4) & 1) * 64 + (ch & 3) * 16;
                }
            }
        } else {
            const unsigned char* Bg = sg.B + (long)z1 * sg.strideB + (long)z2 * sg.strideB2;
            if constexpr (!B_TR) {
                gstep = H3_ROWB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (wave & 3) * 64 + (lane >> 2) + 16 * j;
                    const int n = PAIRED ? (r < 128 ? n0 + r : g.pair_off + n0 + r - 128) : min(n0 + r, g.N - 1);
                    gp[j] = Bg + (long)n * sg.ldb + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
                }
            } else {
                gstep = 16 * sg.ldb;
                const int ncol = PAIRED ? ((lane >> 5) ? g.pair_off + n0 : n0) : min(n0 + 128 * (lane >> 5), g.N - 128);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    {
                    const int ch = (lane & 15) ^ (j << 2);
                    gp[j] = Bg + (long)((wave & 3) * 4 + j) * sg.ldb + (ncol / 128) * 512 + (ch >> 2) * 128 + ((lane >> 4) & 1) * 64 + (ch & 3) * 16;
                }
            }
        }
    };

    // A_CONV: the A-loader waves' source pointers for one tap (k = 0 of the tap's cv_cin values)
    auto setup_conv = [&](int tap, const unsigned char* (&gp)[4]) {
        const int nt = g.cv_taps_total ? g.cv_taps_total : g.cv_ntaps;
        if (g.cv_taps_total) tap += z * g.cv_ntaps;
        const int dy = nt == 9 ? tap / 3 - 1 : 0, dx = nt == 9 ? tap - (tap / 3) * 3 - 1 : 0;
        const int hw = g.cv_Hout * g.cv_Wout;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave & 3) * 64 + (lane >> 2) + 16 * j;
            const int m = min(m0 + r, g.M - 1);
            const int b_ = m / hw, r_ = m - b_ * hw;
            const int h_ = r_ / g.cv_Wout, w_ = r_ - h_ * g.cv_Wout;
            const int ih = h_ * g.cv_stride + dy, iw = w_ * g.cv_stride + dx;
            const bool ok = ih >= 0 && ih < g.cv_Hin && iw >= 0 && iw < g.cv_Win;
            const unsigned char* row = ok ? g.seg[0].A + ((long)b_ * g.cv_Hin * g.cv_Win + (long)ih * g.cv_Win + iw) * g.seg[0].lda : g.zero_row;
            gp[j] = row + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
    };

    // the epilogue's row scales wait in LDS behind the ring (and the TWOSEG row factors): loaded here, ahead of the
    // prologue's DMA, they cost the epilogue a ds_read instead of one exposed global-load round trip per row block
    const H3Seg& sl = g.seg[TWOSEG ? 1 : 0];
    const int zl1 = z / sl.zdiv, zl2 = z - zl1 * sl.zdiv;
    using RowT = decltype(epi.row(0, 0));
    constexpr bool HAS_ROW = !std::is_empty<RowT>::value;
    float* sal = reinterpret_cast<float*>(lds + H3_LDS + 1024);
    RowT* rwl = reinterpret_cast<RowT*>(lds + H3_LDS + 2048);
    float sa_own = 0.f;
    RowT rw_own{};
    if (tid < 256) {
        sa_own = (sl.sa + (long)zl1 * sl.strideSA + (long)zl2 * sl.strideSA2 + (sl.segk ? (long)(sl.K / sl.segk - 1) * sl.strideSeg : 0L))[(long)min(max(m0 + tid, 0), g.M - 1) * sl.sa_mul];
        if constexpr (HAS_ROW) rw_own = epi.row(z, min(max(m0 + tid, 0), g.M - 1));
        if constexpr (epi_has_rowmul<Epi>::value) sa_own *= epi.rowmul(rw_own);
    }

    const H3Seg& sg0 = g.seg[0];
    const int zz2 = z % sg0.zdiv;
    const int nkt0 = (sg0.kchunk ? max(0, min(sg0.K, sg0.ktotal - zz2 * sg0.kchunk)) : sg0.K) / H3_BK;
    const int nkt = TWOSEG ? nkt0 + g.seg[1].K / H3_BK : nkt0;      // (TWOSEG: the host guarantees K0 >= 64, K1 >= 64, no kchunk)
    if (nkt > 0) {
        const unsigned char* gp[4];
        long gstep;
        setup(sg0, gp, gstep);
        if constexpr (A_CONV) { if (!stB) setup_conv(0, gp); }
        if constexpr (TWOSEG) {
            // acc changes its scale domain between the segments (sum/(sa0[m]*sb0[n]) -> units of sa1[m]*sb1[n]; powers of
            // two, exact).  The row factors wait in the 1 KiB of LDS behind the stage ring.
            const H3Seg& s1 = g.seg[1];
            const float* sa0 = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)zz2 * sg0.strideSA2;
            const float* sa1 = s1.sa + (long)(z / s1.zdiv) * s1.strideSA + (long)(z % s1.zdiv) * s1.strideSA2;
            if (tid < 256) {
                const int m = min(max(m0 + tid, 0), g.M - 1);
                const bool zr = sg0.a_period && (m % sg0.a_period) == 0;        // zero operand row: any finite factor will do
                const float f0 = zr ? 1.0f : sa0[(long)(m + sg0.a_shift) * sg0.sa_mul];
                reinterpret_cast<float*>(lds + H3_LDS)[tid] = f0 / sa1[(long)m * s1.sa_mul];
            }
        }

        if constexpr (!TWOSEG && !A_TR) {
            if (sg0.segk && tid < 256) {          // factors of the segment boundaries: units of segment j -> units of segment j+1
                const float* sab = sg0.sa + (long)(z / sg0.zdiv) * sg0.strideSA + (long)zz2 * sg0.strideSA2 + (long)min(m0 + tid, g.M - 1) * sg0.sa_mul;
                float* rft = reinterpret_cast<float*>(lds + H3_LDS + 4096);
                const int nb = sg0.K / sg0.segk - 1;          // (<= 7: host-checked)
                float sc[8];                                  // all segment scales requested at once: a load -> divide -> store chain per
#pragma unroll                                                // boundary cost the to_out tile 4-5 us of exposed latency
                for (int j = 0; j < 8; ++j) sc[j] = sab[(long)min(j, nb) * sg0.strideSeg];
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < nb) rft[j * 256 + tid] = sc[j] / sc[j + 1];
            }
        }

        // ---- prologue: tiles 0..3 in flight, fragments of tile 0 in registers
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < nkt0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + p * gstep, lds + p * H3_STAGE + sdst + j * 1024);
            }
        if (tid < 256) { sal[tid] = sa_own; if constexpr (HAS_ROW) rwl[tid] = rw_own; }
        if (nkt0 >= 4) H3_WAIT_VM(12); else if (nkt0 == 3) H3_WAIT_VM(8); else if (nkt0 == 2) H3_WAIT_VM(4); else H3_WAIT_VM(0);
        H3_BARRIER();
        H3_STAMP(1);
        f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            bh[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(lds, f, tn, 0));
            bl[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(lds, f, tn, 1));
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            ah[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(lds, f, tm, 0));
            al[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(lds, f, tm, 1));
        }

        // ---- stages.  At the top of stage t the wave's own pieces of tile t+1 must have landed
        // (tiles t+2, t+3 may stay in flight: vmcnt(8)); the barrier then makes tile t+1 readable for everyone and —
        // every wave having retired its reads of tile t (lgkmcnt(0) before the barrier) — frees buffer t&3 for tile t+4.
        // TWOSEG: ONE pipeline over both segments — the DMA switches to the second segment's operands three
        // stages before the MFMAs do, the accumulators are rescaled between stage nkt0-1 and stage nkt0.
#define H3_STEADY(FULLN, PH, LIMIT, TILE0)                                                                             \
            for (; t + 4 < (LIMIT); ++t) {                                                                             \
                if constexpr (!TWOSEG && !A_TR && !A_CONV) {                                                           \
                    if (t == seg_bnd) {   /* segmented row scales: the accumulators go to the next segment's units */  \
                        const float* rfl = reinterpret_cast<const float*>(lds + H3_LDS + 4096) + seg_j * 256;          \
                        _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                               \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
                                const f32x4 rf = *reinterpret_cast<const f32x4*>(rfl + wm * 128 + tm * 32 + 8 * j + 4 * h); \
                                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
                                    acc[tm][0][4 * j + i] *= rf[i];                                                    \
                                    acc[tm][1][4 * j + i] *= rf[i];                                                    \
                                }                                                                                      \
                            }                                                                                          \
                        ++seg_j; seg_bnd += seg_stp;                                                                   \
                    }                                                                                                  \
                }                                                                                                      \
                if (VARIANT != 9) H3_WAIT_VM(8);                                                                       \
                if (VARIANT != 4) H3_BARRIER();                                                                        \
                if (wave_on)                                                                                           \
                    h3_stage<A_TR, B_TR, FULLN, VARIANT != 2 && VARIANT != 3, VARIANT != 1 && VARIANT != 3, PH, VARIANT == 6>( \
                        acc, ah, al, bh, bl, lds + ((t + 1) & 3) * H3_STAGE, f, gp, (long)(t + 4 - (TILE0)) * gstep,   \
                        lds + ((t + 4) & 3) * H3_STAGE + sdst);                                                        \
                else                                                                                                   \
                    h3_stage_dma_only<VARIANT != 1 && VARIANT != 3>(gp, (long)(t + 4 - (TILE0)) * gstep,               \
                                                                    lds + ((t + 4) & 3) * H3_STAGE + sdst);            \
            }
#define H3_RUN(FULLN, PH)                                                                                              \
        {                                                                                                              \
            int t = 0;                                                                                                 \
            /* segmented row scales (segk >= 64, K % segk == 0: every boundary lies in the steady part of the loop) */ \
            const int seg_stp = sg0.segk ? sg0.segk / H3_BK : 0;                                                       \
            int seg_bnd = sg0.segk ? seg_stp : 0x7fffffff, seg_j = 0;                                                  \
            (void)seg_stp; (void)seg_bnd; (void)seg_j;                                                                 \
            if constexpr (TWOSEG) {                                                                                    \
                H3_STEADY(FULLN, PH, nkt0, 0)                                                                          \
                setup(g.seg[1], gp, gstep);                                                                            \
                H3_STEADY(FULLN, PH, nkt0 + 4, nkt0)                                                                   \
                {                                                                                                      \
                    const H3Seg& s1 = g.seg[1];                                                                        \
                    const float* sb0 = sg0.sb + (long)(z / sg0.zdiv) * sg0.strideSB + (long)zz2 * sg0.strideSB2;       \
                    const float* sb1 = s1.sb + (long)(z / s1.zdiv) * s1.strideSB + (long)(z % s1.zdiv) * s1.strideSB2; \
                    float cf[2] = {1.0f, 1.0f};                                                                        \
                    if (sb0 != sb1 || sg0.sb_mul != s1.sb_mul) {   /* (same weight rows in both segments: no column factor, no load) */ \
                        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) {                                             \
                            const int c = PAIRED ? (tn == 0 ? n0 + wn * 32 + l31 : g.pair_off + n0 + wn * 32 + l31)    \
                                                 : min(n0 + tn * 128 + wn * 32 + l31, g.N - 1);                        \
                            cf[tn] = sb0[(long)c * sg0.sb_mul] / sb1[(long)c * s1.sb_mul];                             \
                        }                                                                                              \
                    }                                                                                                  \
                    const float* rfl = reinterpret_cast<const float*>(lds + H3_LDS);                                   \
                    _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                                   \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
                            const f32x4 rf = *reinterpret_cast<const f32x4*>(rfl + wm * 128 + tm * 32 + 8 * j + 4 * h);\
                            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                                acc[tm][0][4 * j + i] *= rf[i] * cf[0];                                                \
                                acc[tm][1][4 * j + i] *= rf[i] * cf[1];                                                \
                            }                                                                                          \
                        }                                                                                              \
                }                                                                                                      \
                H3_STEADY(FULLN, PH, nkt, nkt0)                                                                        \
            } else if constexpr (A_CONV) {                                                                             \
                /* one pipeline over the taps: the A loaders re-aim at the next tap's pixel rows four stages before */ \
                /* the MFMAs get there; the weight rows (B) run straight through K = ntaps * cv_cin                  */ \
                const int nk = g.cv_cin / H3_BK;                                                                       \
                for (int tap = 0; tap < g.cv_ntaps; ++tap) {                                                           \
                    if (tap > 0 && !stB) setup_conv(tap, gp);                                                          \
                    const int tile0 = stB ? 0 : tap * nk;                                                              \
                    H3_STEADY(FULLN, PH, (tap + 1) * nk, tile0)                                                        \
                }                                                                                                      \
            } else {                                                                                                   \
                H3_STEADY(FULLN, PH, nkt, 0)                                                                           \
            }                                                                                                          \
            for (; t + 1 < nkt; ++t) {                                                                                 \
                if (t + 3 < nkt) H3_WAIT_VM(8); else if (t + 2 < nkt) H3_WAIT_VM(4); else H3_WAIT_VM(0);                                                    \
                H3_BARRIER();                                                                                          \
                if (wave_on) h3_stage<A_TR, B_TR, FULLN, true, false, PH>(acc, ah, al, bh, bl, lds + ((t + 1) & 3) * H3_STAGE, f, gp, 0, lds); \
            }                                                                                                          \
            if (wave_on) h3_stage<A_TR, B_TR, FULLN, false, false, PH>(acc, ah, al, bh, bl, lds, f, gp, 0, lds);       \
        }
        if constexpr (PINGPONG) {
            // ---- ping-pong main loop (plain mode: one segment, no convolution).  The two waves of a SIMD (w and w+4) never
            // mix MFMAs with loads: while group 0 (waves 0-3) issues the 24 MFMAs of tile t back to back, group 1 reads its
            // fragments of tile t and issues its LDS-DMA pieces; then they swap — two barriers per k-tile:
            //     G0:  MFMA(t)           | frags(t+1), DMA(t+4)
            //     G1:  frags(t), DMA(t+3) | MFMA(t)
            // Slot (t&3) is read by G0 in the second phase of stage t-1 and by G1 in the first phase of stage t: G0 refills it
            // (tile t+4) in the second phase of stage t, G1 one phase later (tile t+3 in the first phase of stage t).
            auto wait_keep = [&](int tiles) {          // at most `tiles` tiles (4 pieces each) of this wave stay in flight
                if (tiles <= 0) H3_WAIT_VM(0); else if (tiles == 1) H3_WAIT_VM(4); else if (tiles == 2) H3_WAIT_VM(8); else H3_WAIT_VM(12);
            };
            auto mma = [&](auto fulln) {
                constexpr bool FN = decltype(fulln)::value;
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < (FN ? 2 : 1); ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    }
            };
            auto frags = [&](const unsigned char* st, auto fulln) {
                constexpr bool FN = decltype(fulln)::value;
#pragma unroll
                for (int tn = 0; tn < (FN ? 2 : 1); ++tn) {
                    bh[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(st, f, tn, 0));
                    bl[tn] = h3_frag<B_TR>(h3_addr_b<B_TR>(st, f, tn, 1));
                }
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    ah[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(st, f, tm, 0));
                    al[tm] = h3_frag<A_TR>(h3_addr_a<A_TR>(st, f, tm, 1));
                }
            };
            auto dma = [&](int ti) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h3_glds16(gp[j] + (long)ti * gstep, lds + (ti & 3) * H3_STAGE + sdst + j * 1024);
            };
            auto run = [&](auto fulln) {
                for (int t = 0; t < nkt; ++t) {
                    if (!stB) {
                        const int last = min(t + 3, nkt - 1);
                        wait_keep(last - t);
                        H3_BARRIER();
                        if (wave_on) mma(fulln);
                        wait_keep(last - (t + 1));
                        H3_BARRIER();
                        if (t + 1 < nkt && wave_on) frags(lds + ((t + 1) & 3) * H3_STAGE, fulln);
                        if (t + 4 < nkt) dma(t + 4);
                    } else {
                        wait_keep(min(max(t + 2, 3), nkt - 1) - t);
                        H3_BARRIER();
                        if (t > 0 && wave_on) frags(lds + (t & 3) * H3_STAGE, fulln);
                        if (t >= 1 && t + 3 < nkt) dma(t + 3);
                        wait_keep(min(max(t + 3, 3), nkt - 1) - (t + 1));
                        H3_BARRIER();
                        if (wave_on) mma(fulln);
                    }
                }
            };
            if (full_n) run(std::true_type{}); else run(std::false_type{});
        } else {
        // the two waves of a SIMD (w and w+4) issue their DMA in different halves of the stage
        if (full_n) { if (stB) H3_RUN(true, 1) else H3_RUN(true, 0) }
        else { if (stB) H3_RUN(false, 1) else H3_RUN(false, 0) }
        }
#undef H3_RUN
#undef H3_STEADY
    } else {
        if (tid < 256) { sal[tid] = sa_own; if constexpr (HAS_ROW) rwl[tid] = rw_own; }
        __syncthreads();
    }

    // ---- epilogue: D col = l31, row = (r&3) + 8*(r>>2) + 4*h; scales of the LAST segment.
    // vmcnt retires in order and counts stores: any wait on a load issued after a store waits for that store's round
    // trip (~130 ns on an idle chip).  So (i) a tile whose 256 rows all exist takes a branch-free path — per-row
    // `if (m < M)` blocks make the compiler's waitcnt pass re-wait vmcnt(0) at every block entry, i.e. one store round
    // trip per store: 17 us per tile; (ii) row scales and the functor's row() values come from LDS (staged before the
    // prologue); (iii) column constants are fetched before the first store and aux() operands half a row block ahead.
    H3_STAMP(2);
    if (g.dbg & 1) {                    // (timing: no epilogue)
        float x = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) x += acc[0][0][r] + acc[1][1][r] + acc[2][0][r] + acc[3][1][r];
        if (x != 123.456f) return;
    }
    const float* sb = sl.sb + (long)zl1 * sl.strideSB + (long)zl2 * sl.strideSB2;
    const int sbm = sl.sb_mul;
    // this lane's first row and column inside the tile — laundered, so that the epilogue's index arithmetic is not hoisted
    // above the main loop (where it would be spilled: the loop runs at the register limit)
    // (and again inside each of the two paths: common subexpressions of the paths would be hoisted to their dominator, all
    // 128 row indices at once, and spilled there)
    const int lrow0 = wm * 128 + 4 * h, lcol0 = wn * 32 + l31;
    auto row_of = [&](int lr) -> RowT { if constexpr (HAS_ROW) return rwl[lr]; else return RowT{}; };
    auto epilogue = [&](auto tag) {
        if constexpr (!epi_has_plout<Epi>::value) {
        constexpr bool CHECK = decltype(tag)::value;
        int lrow = lrow0, lcol = lcol0;
        asm volatile("" : "+v"(lrow), "+v"(lcol));
        auto rowm = [&](int lr) { return CHECK ? min(m0 + lr, g.M - 1) : m0 + lr; };
        if constexpr (PAIRED) {
            const int c = n0 + lcol;
            auto cc = epi.col(z, c);
            float sc0 = sb[(long)c * sbm], sc1 = sb[(long)(g.pair_off + c) * sbm];
            touch(cc); touch(sc0); touch(sc1);
            if constexpr (epi_has_pairmul<Epi>::value) { const float2 pm = epi.pairmul(cc); sc0 *= pm.x; sc1 *= pm.y; }
            if constexpr (epi_has_aux<Epi>::value) {
                // 8 half row blocks; the aux operands (cold HBM reads) run H3_AUX_AHEAD blocks ahead of the stores: with one
                // block ahead a CU has 32 KB of requests in flight against 2-4 us of loaded HBM latency (= 2.7 TB/s chip-wide)
                constexpr int AH = H3_AUX_AHEAD;
                decltype(epi.aux(0, 0, 0, RowT{})) ax[AH + 1][8];
                auto fetch = [&](int hb) {
                    int lb = lrow + (hb >> 1) * 32;       // (laundered per block: index arithmetic stays inside its block)
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = (hb & 1) * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        ax[hb % (AH + 1)][r8] = epi.aux(z, rowm(lr), c, row_of(lr));
                    }
                };
#pragma unroll
                for (int hb = 0; hb < AH; ++hb) fetch(hb);
#pragma unroll
                for (int hb = 0; hb < 8; ++hb) {
                    __builtin_amdgcn_sched_barrier(0);      // (keeps the work of later blocks from being hoisted: registers)
                    if (hb + AH < 8) fetch(hb + AH);
                    RowT rw[8];
                    float sr[8];
                    int lb = lrow + (hb >> 1) * 32;
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = (hb & 1) * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        rw[r8] = row_of(lr);
                        sr[r8] = sal[lr];
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) touch(ax[hb % (AH + 1)][r8]);
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int tm = hb >> 1, r = (hb & 1) * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        if constexpr (!CHECK && epi_has_full<Epi>::value) h3_assume_row(rw[r8]);
                        if (!CHECK || m0 + lr < g.M) {
                            if constexpr (epi_has_pairmul<Epi>::value)
                                epi.store2_scaled(z, m0 + lr, c, acc[tm][0][r] * (sr[r8] * sc0), acc[tm][1][r] * (sr[r8] * sc1), rw[r8], cc, ax[hb % (AH + 1)][r8]);
                            else
                                epi.store2(z, m0 + lr, c, acc[tm][0][r] * (sr[r8] * sc0), acc[tm][1][r] * (sr[r8] * sc1), rw[r8], cc, ax[hb % (AH + 1)][r8]);
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
                        if constexpr (!CHECK && epi_has_full<Epi>::value) h3_assume_row(rw);
                        if (!CHECK || m0 + lr < g.M) epi.store2(z, m0 + lr, c, acc[tm][0][r] * (sr * sc0), acc[tm][1][r] * (sr * sc1), rw, cc);
                    }
                }
            }
        } else {
            int nn[2];
            decltype(epi.col(0, 0)) cc[2];
            float sc[2];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                nn[tn] = n0 + tn * 128 + lcol;
                const int nc = min(nn[tn], g.N - 1);
                cc[tn] = epi.col(z, nc);
                sc[tn] = sb[(long)nc * sbm];
            }
            touch(cc[0]); touch(cc[1]); touch(sc[0]); touch(sc[1]);
            if constexpr (epi_has_aux<Epi>::value) {
                // 16 half row blocks (tn, tm, half): the aux operands of block i+1 are requested before the stores of block i
                // (row values and row scales are LDS reads — lgkmcnt, not vmcnt — and are fetched where they are used)
                constexpr int AH = H3_AUX_AHEAD;
                decltype(epi.aux(0, 0, 0, RowT{})) ax[AH + 1][8];
                auto fetch = [&](int i, int set) {
                    const int tn = i >> 3, tm = (i >> 1) & 3, half = i & 1;
                    int lb = lrow + tm * 32;            // (laundered per block: index arithmetic stays inside its block)
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = half * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        ax[set][r8] = epi.aux(z, rowm(lr), min(nn[tn], g.N - 1), row_of(lr));
                    }
                };
#pragma unroll
                for (int i = 0; i < AH; ++i) fetch(i, i % (AH + 1));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_barrier(0);      // (keeps the work of later blocks from being hoisted: registers)
                    if (i + AH < 16) fetch(i + AH, (i + AH) % (AH + 1));
                    const int tn = i >> 3, tm = (i >> 1) & 3, half = i & 1;
                    RowT rw[8];
                    float sr[8];
                    int lb = lrow + tm * 32;
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = half * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        rw[r8] = row_of(lr);
                        sr[r8] = sal[lr];
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) touch(ax[i % (AH + 1)][r8]);
                    if (nn[tn] < g.N) {
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = half * 8 + r8;
                            const int lr = lb + (r & 3) + 8 * (r >> 2);
                            if constexpr (!CHECK && epi_has_full<Epi>::value) h3_assume_row(rw[r8]);
                            if (!CHECK || m0 + lr < g.M)
                                epi.store(z, m0 + lr, nn[tn], acc[tm][tn][r] * (sr[r8] * sc[tn]), rw[r8], cc[tn], ax[i % (AH + 1)][r8]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if (nn[tn] >= g.N) continue;
                    if constexpr (epi_has_ptr<Epi>::value) {
                        // functors with one row-major output: the element address is formed ONCE per lane and column block;
                        // the rows of the wave tile are compile-time multiples of the (uniform) row pitch away — one 64-bit
                        // add per store instead of a 64-bit multiply-add chain
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
                            if constexpr (!CHECK && epi_has_full<Epi>::value) h3_assume_row(rw);
                            if (!CHECK || m0 + lr < g.M) epi.store(z, m0 + lr, nn[tn], acc[tm][tn][r] * (sr * sc[tn]), rw, cc[tn]);
                        }
                    }
                    }
                }
            }
        }
        }
    };

    // ---- PLOUT epilogue (see H3PlOut): phase 1 = the usual per-lane epilogue math with the value written to the staging
    // tile T (fp32, in the ring: [256][128] PAIRED, [128][256] per half otherwise; 32-float blocks swapped on rows with
    // bit 2 set so that the two half-waves of a store hit different banks), phase 2 = one tile row per wave (two per wave
    // PAIRED): row max, scale, planes (16 B per lane, full lines), scale / sum of squares by lane 0.
    auto epilogue_pl = [&](auto tag) {
        if constexpr (epi_has_plout<Epi>::value) {
        constexpr bool CHECK = decltype(tag)::value;
        int lrow = lrow0, lcol = lcol0;
        asm volatile("" : "+v"(lrow), "+v"(lcol));
        float* const T = reinterpret_cast<float*>(lds);
        const H3PlOut po = epi.plout(z);
        auto rowm = [&](int lr) { return CHECK ? min(m0 + lr, g.M - 1) : m0 + lr; };
        __syncthreads();                         // every wave has retired its last fragment reads: the ring is free
        if constexpr (PAIRED) {
            static_assert(epi_has_aux<Epi>::value && epi_has_pairmul<Epi>::value, "PLOUT PAIRED: gate-style functor");
            const int c = n0 + lcol;
            auto cc = epi.col(z, c);
            float sc0 = sb[(long)c * sbm], sc1 = sb[(long)(g.pair_off + c) * sbm];
            touch(cc); touch(sc0); touch(sc1);
            { const float2 pm = epi.pairmul(cc); sc0 *= pm.x; sc1 *= pm.y; }
            constexpr int AH = H3_AUX_AHEAD;
            decltype(epi.aux(0, 0, 0, RowT{})) ax[AH + 1][8];
            auto fetch = [&](int hb) {
                int lb = lrow + (hb >> 1) * 32;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = (hb & 1) * 8 + r8;
                    const int lr = lb + (r & 3) + 8 * (r >> 2);
                    ax[hb % (AH + 1)][r8] = epi.aux(z, rowm(lr), c, row_of(lr));
                }
            };
#pragma unroll
            for (int hb = 0; hb < AH; ++hb) fetch(hb);
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                __builtin_amdgcn_sched_barrier(0);
                if (hb + AH < 8) fetch(hb + AH);
                RowT rw[8];
                float sr[8];
                int lb = lrow + (hb >> 1) * 32;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = (hb & 1) * 8 + r8;
                    const int lr = lb + (r & 3) + 8 * (r >> 2);
                    rw[r8] = row_of(lr);
                    sr[r8] = sal[lr];
                }
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) touch(ax[hb % (AH + 1)][r8]);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int tm = hb >> 1, r = (hb & 1) * 8 + r8;
                    const int lr = lb + (r & 3) + 8 * (r >> 2);
                    float v = epi.val2_scaled(z, m0 + lr, c, acc[tm][0][r] * (sr[r8] * sc0), acc[tm][1][r] * (sr[r8] * sc1), rw[r8], cc, ax[hb % (AH + 1)][r8]);
                    if (CHECK && m0 + lr >= g.M) v = 0.f;
                    T[lr * 128 + (lcol ^ (((lr >> 2) & 1) << 5))] = v;
                }
            }
            __syncthreads();
            H3_STAMP(3);
            const int hw = lane >> 5, l32 = lane & 31;
            unsigned char* const pbase = po.planes + (long)n0 * 4;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = wave * 32 + it * 2 + hw;
                const int m = m0 + rl;
                const long pr = (!CHECK || m < g.M) ? epi.prow(z, min(m, g.M - 1)) : -1L;
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
        } else {
            int nn[2];
            decltype(epi.col(0, 0)) cc[2];
            float sc[2];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                nn[tn] = n0 + tn * 128 + lcol;            // (PLOUT: N is a multiple of 256)
                cc[tn] = epi.col(z, nn[tn]);
                sc[tn] = sb[(long)nn[tn] * sbm];
            }
            touch(cc[0]); touch(cc[1]); touch(sc[0]); touch(sc[1]);
            unsigned char* const pbase = po.planes + (long)n0 * 4;
            auto phase2 = [&](int hf) {
                __syncthreads();
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int rl = wave * 16 + it;
                    const int lr = (rl >> 6) * 128 + (2 * hf + ((rl >> 5) & 1)) * 32 + (rl & 31);
                    const int m = m0 + lr;
                    const long pr = (!CHECK || m < g.M) ? epi.prow(z, min(m, g.M - 1)) : -1L;      // (wave-uniform)
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(T + rl * 256 + ((4 * lane) ^ (((rl >> 2) & 1) << 5)));
                    const float4 v = make_float4(v4[0], v4[1], v4[2], v4[3]);
                    const float mu = h3_wave_max(h3_absmax4(v));
                    float q2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    if (po.ss) q2 = h3_wave_sum(q2);
                    float inv;
                    const float s_ = h3_row_scale(mu, inv);
                    if (pr >= 0) {
                        h3_emit4(pbase + pr * po.pitch, lane, v, s_);
                        if constexpr (epi_has_rowout<Epi>::value) *reinterpret_cast<f32x4*>(epi.rowout(z, pr, n0) + 4 * lane) = v4;
                        if (lane == 0) {
                            po.scale[(long)bn * po.seg_stride + pr] = inv;
                            if (po.ss) po.ss[(long)bn * po.seg_stride + pr] = q2;
                        }
                    }
                }
            };
            if constexpr (epi_has_aux<Epi>::value) {
                // 16 half row blocks in the order (half tile hf, tn, tm & 1, half): the aux operands of block i+1 are
                // requested before the values of block i are formed
                constexpr int AH = H3_AUX_AHEAD;
                decltype(epi.aux(0, 0, 0, RowT{})) ax[AH + 1][8];
                auto fetch = [&](int i, int set) {
                    const int tn = (i >> 2) & 1, tm = 2 * (i >> 3) + ((i >> 1) & 1), half = i & 1;
                    int lb = lrow + tm * 32;
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = half * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        ax[set][r8] = epi.aux(z, rowm(lr), nn[tn], row_of(lr));
                    }
                };
#pragma unroll
                for (int i = 0; i < AH; ++i) fetch(i, i % (AH + 1));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 8) __syncthreads();             // phase 2 of the first half has read the tile
                    if (i + AH < 16) fetch(i + AH, (i + AH) % (AH + 1));
                    const int tn = (i >> 2) & 1, tmh = (i >> 1) & 1, tm = 2 * (i >> 3) + tmh, half = i & 1;
                    RowT rw[8];
                    float sr[8];
                    int lb = lrow + tm * 32;
                    asm volatile("" : "+v"(lb));
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = half * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        rw[r8] = row_of(lr);
                        sr[r8] = sal[lr];
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) touch(ax[i % (AH + 1)][r8]);
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = half * 8 + r8;
                        const int lr = lb + (r & 3) + 8 * (r >> 2);
                        float v = 0.f;
                        if (!CHECK || m0 + lr < g.M) v = epi.val(z, m0 + lr, nn[tn], acc[tm][tn][r] * (sr[r8] * sc[tn]), rw[r8], cc[tn], ax[i % (AH + 1)][r8]);
                        const int rl = lr - wm * 64 - (tm - tmh) * 32;          // (wm*128 + tm*32 + x) -> wm*64 + tmh*32 + x
                        T[rl * 256 + ((tn * 128 + lcol) ^ (((rl >> 2) & 1) << 5))] = v;
                    }
                    if (i == 7 || i == 15) phase2(i >> 3);
                }
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if (hf) __syncthreads();
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
                        for (int tmh = 0; tmh < 2; ++tmh) {
                            const int tm = 2 * hf + tmh;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int lr = lrow + tm * 32 + (r & 3) + 8 * (r >> 2);
                                const RowT rw = row_of(lr);
                                const float sr = sal[lr];
                                float v = 0.f;
                                if (!CHECK || m0 + lr < g.M) v = epi.val(z, m0 + lr, nn[tn], acc[tm][tn][r] * (sr * sc[tn]), rw, cc[tn]);
                                const int rl = lr - wm * 64 - hf * 64 + 0 * tmh;      // wm*128 + (2hf+tmh)*32 + x -> wm*64 + tmh*32 + x
                                T[rl * 256 + ((tn * 128 + lcol) ^ (((rl >> 2) & 1) << 5))] = v;
                            }
                        }
                    }
                    phase2(hf);
                }
            }
        }
        }
    };

    // ---- CONV epilogue (see H3Conv)
    auto epilogue_conv = [&]() {
        if constexpr (CONV) {
        int lrow = lrow0, lcol = lcol0;
        asm volatile("" : "+v"(lrow), "+v"(lcol));
        const H3Conv cv = epi.conv();
        int nn[2];
        decltype(epi.col(0, 0)) cc[2];
        float sc[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            nn[tn] = n0 + tn * 128 + lcol;
            const int nc = min(nn[tn], g.N - 1);
            cc[tn] = epi.col(z, nc);
            sc[tn] = sb[(long)nc * sbm] * epi.colmul(cc[tn]);
        }
        touch(cc[0]); touch(cc[1]); touch(sc[0]); touch(sc[1]);
        if (n0 >= cv.ncols) {
            // plain columns (to_qk: pre-activation for conv17<3>): the owner rows 8 .. 247 of the tile (every row has one owner)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                if (nn[tn] >= g.N) continue;
                float* const p0 = epi.ptr(z, m0 + lrow, nn[tn]);
                const long ldm = epi.ldm();
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const int lr = lrow + k;
                        const RowT rw = row_of(lr);
                        const float sr = sal[lr];
                        if (lr >= H3_CONV_HALO && lr < H3_CONV_HALO + H3_CONV_STEP && m0 + lr < g.M)
                            epi.put_scaled(p0 + k * ldm, acc[tm][tn][r] * (sr * sc[tn]), rw, cc[tn]);
                    }
                }
            }
            return;
        }
        float* const T = reinterpret_cast<float*>(lds);
        int* const sid = reinterpret_cast<int*>(lds + H3_LDS + 4096);         // sample index of tile row r, -1 outside [0, M)
        // fast tiles: all 256 rows exist and belong to one sample (wave-uniform)
        const bool fast = m0 >= 0 && m0 + 255 < g.M && (m0 / cv.S) == ((m0 + 255) / cv.S);
        if (!fast && tid < 256) { const int m = m0 + tid; sid[tid] = (m >= 0 && m < g.M) ? m / cv.S : -1; }
        typedef float v2f __attribute__((ext_vector_type(2)));
        constexpr int KT = 17, HALF = 8, RW = 30;
        const int r_first = H3_CONV_HALO + RW * wave;                      // this wave's output rows r_first .. r_first + 29
        __syncthreads();                             // every wave has retired its last fragment reads: the ring is free
        auto half = [&](auto tnc) {
            constexpr int tn = decltype(tnc)::value;     // (two instances: the first half's accumulators die after its phase 1)
            // ---- phase 1: y = act(acc * scales + bias) -> T[row][128]
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lr = lrow + tm * 32 + 8 * j;                 // rows lr .. lr + 3: one 16-B read of their scales
                    const f32x4 sr4 = *reinterpret_cast<const f32x4*>(sal + lr);
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[(lr + i) * 128 + lcol] = (g.dbg & 16) ? acc[tm][tn][4 * j + i] : epi.act(acc[tm][tn][4 * j + i] * (sr4[i] * sc[tn]), cc[tn]);
                }
            }
            // taps of this lane's column pair (issued before the barrier: their latency hides under it)
            const int colp = n0 + tn * 128 + 2 * lane;
            v2f w[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) w[t] = *reinterpret_cast<const v2f*>(cv.w + (long)t * cv.wld + colp);
            __syncthreads();
            H3_STAMP(tn == 0 ? 3 : 6);               // (diagnostic builds: phase 1 of this half done)
            const float* const Tc = T + 2 * lane;
            // per-lane byte offsets inside a planes row / an fp32 row; the row bases are wave-uniform (scalar + 32-bit offset stores)
            const unsigned poff = (unsigned)((colp >> 5) * 128 + (colp & 31) * 2);
            const bool has32 = cv.u32 && colp >= cv.u0;
            const unsigned uoff = (unsigned)(colp - cv.u0) * 4u;
            auto emit = [&](v2f o, long prow, long m) {
                // split with the static scale: (c, c+1) of the hi plane = one dword, of the lo plane another
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const v2f xs = o * v2f{cv.sv, cv.sv};
                const h2 hi = h2{(_Float16)xs[0], (_Float16)xs[1]};
                const h2 lo = h2{(_Float16)(xs[0] - (float)hi[0]), (_Float16)(xs[1] - (float)hi[1])};
                unsigned char* const rowp = cv.planes + prow * cv.pitch;           // uniform
                if (g.dbg & 4) { if (xs[0] == 123.456f) *reinterpret_cast<h2*>(rowp + poff) = lo; return; }      // (timing: no stores)
                *reinterpret_cast<h2*>(rowp + poff) = hi;
                *reinterpret_cast<h2*>(rowp + 64 + poff) = lo;
                if (has32) *reinterpret_cast<v2f*>(reinterpret_cast<unsigned char*>(cv.u32 + m * cv.ld32) + uoff) = o;
            };
            if (fast) {
                const int b0 = m0 / cv.S;
                const long prow0 = (long)b0 * cv.Sp + (m0 - b0 * cv.S);     // planes row of tile row 0
                // the wave's 30 output rows in three runs of 10: a 26-row window in registers, static indices, no shifting
                constexpr int RUN = 10;
#pragma unroll 1
                for (int r0 = r_first; r0 < r_first + RW; r0 += RUN) {
                    v2f win[RUN + KT - 1];
#pragma unroll
                    for (int i = 0; i < RUN + KT - 1; ++i) win[i] = *reinterpret_cast<const v2f*>(Tc + (r0 - HALF + i) * 128);
#pragma unroll
                    for (int i = 0; i < RUN; ++i) {
                        v2f o = win[HALF + i];
                        if (!(g.dbg & 8)) {                                   // (timing: no taps)
#pragma unroll
                            for (int t = 0; t < KT; ++t) o = __builtin_elementwise_fma(w[t], win[i + t], o);
                        }
                        emit(o, prow0 + r0 + i, (long)m0 + r0 + i);
                    }
                }
            } else {
                // slow tiles (a sample boundary or an end of the row range inside the tile): taps of another sample count as zero
#pragma unroll 1
                for (int lr = r_first; lr < r_first + RW; ++lr) {
                    const int so = sid[lr];
                    if (so < 0) continue;                   // (wave-uniform)
                    v2f o = *reinterpret_cast<const v2f*>(Tc + lr * 128);
#pragma unroll
                    for (int t = 0; t < KT; ++t) {          // (unrolled: w[] must stay in registers)
                        const int rt = lr + t - HALF;       // 0 .. 255 by construction
                        if (sid[rt] == so) o = __builtin_elementwise_fma(w[t], *reinterpret_cast<const v2f*>(Tc + rt * 128), o);
                    }
                    const long m = (long)m0 + lr;
                    emit(o, (long)so * cv.Sp + (m - (long)so * cv.S), m);
                }
            }
        };
        if (g.dbg & 32) return;                      // (timing: no conv epilogue at all)
        half(std::integral_constant<int, 0>{});
        H3_STAMP(5);                                 // (wave 0's convolution of the first half done)
        __syncthreads();                             // the convolution of the first half has read T
        half(std::integral_constant<int, 1>{});
        }
    };
    if constexpr (CONV) { epilogue_conv(); H3_STAMP(4); return; }
    bool whole = m0 + 256 <= g.M;
    if constexpr (epi_has_full<Epi>::value) whole = whole && epi.full(z, m0);
    if constexpr (epi_has_plout<Epi>::value) { if (whole) epilogue_pl(std::false_type{}); else epilogue_pl(std::true_type{}); }
    else { if (whole) epilogue(std::false_type{}); else epilogue(std::true_type{}); }
    H3_STAMP(4);
}

template <bool A_TR, bool B_TR, bool PAIRED, bool TWOSEG, class Epi, int VARIANT = 0, bool A_CONV = false>
inline hipError_t launch_gemm_h3x(H3Args g, int batches, Epi epi, hipStream_t st) {
    // (function-local static: set once, thread-safe — forwards may be issued from several host threads)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3_kernel<A_TR, B_TR, PAIRED, TWOSEG, Epi, VARIANT, A_CONV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            H3_LDS + H3_LDS_EXTRA);
    (void)attr_rc;
    g.tiles_m = (g.M + H3_BM - 1) / H3_BM;
    if constexpr (epi_has_conv<Epi>::value) {        // overlapping M tiles: 240 output rows each (see H3Conv)
        g.tiles_m = (g.M + H3_CONV_STEP - 1) / H3_CONV_STEP;
        if (batches != 1 || g.seg[0].segk || g.seg[0].kchunk) return hipErrorInvalidValue;
    }
    g.tiles_n = PAIRED ? g.N / (H3_BN / 2) : (g.N + H3_BN - 1) / H3_BN;
    g.batches = batches;
    dim3 grid;
    if (g.tiles_m >= 16 && batches <= 4) {
        g.map_mode = 0;
        g.mp = (g.tiles_m + 7) / 8;
        long ktot = 0;
        for (int i = 0; i < g.nseg; ++i) ktot += g.seg[i].K;
        static const long gw_kb = [] { const char* e = getenv("TDX_H3_GW_KB"); return e ? atol(e) : 1536L; }();
        long gw = (gw_kb * 1024) / (256L * ktot * 4);      // weight slice of a group stays in the XCD's L2
        if (gw < 1) gw = 1;
        if (gw > g.tiles_n) gw = g.tiles_n;
        g.gw = (int)gw;
        // several N groups re-read the A rows: keep a chunk's rows (per XCD <= ~6 MB, 48 MB on the chip) inside the 256 MB Infinity Cache
        static const int mc_env = [] { const char* e = getenv("TDX_H3_MC"); return e ? atoi(e) : -1; }();
        g.mc = 0;
        if (g.gw < g.tiles_n) {
            long mc = mc_env >= 0 ? mc_env : (6L * 1024 * 1024) / (256L * ktot * 4);
            if (mc > 0 && mc < g.mp) g.mc = (int)mc;
        }
        grid = dim3(8 * g.mp * g.tiles_n, batches, 1);
    } else {
        g.map_mode = 1;
        g.mp = 0; g.gw = 1;
        grid = dim3(8 * ((batches + 7) / 8) * g.tiles_m * g.tiles_n, 1, 1);
    }
    if (TWOSEG && (g.seg[0].K < 64 || g.seg[1].K < 64 || g.seg[0].kchunk || g.seg[1].kchunk)) return hipErrorInvalidValue;
    for (int i = 0; i < g.nseg; ++i) {
        const H3Seg& sg = g.seg[i];
        if (sg.segk && (TWOSEG || A_TR || A_CONV || sg.segk % 64 || sg.K % sg.segk || sg.K / sg.segk > 8 || sg.kchunk)) return hipErrorInvalidValue;
        if ((sg.a_shift || sg.a_period) && (A_TR || A_CONV || (sg.a_period && !sg.a_zero))) return hipErrorInvalidValue;
    }
    if (epi_has_plout<Epi>::value && !PAIRED && g.N % 256) return hipErrorInvalidValue;
    if (A_CONV && (g.cv_cin < 64 || g.cv_cin % 16 || g.seg[0].K != g.cv_ntaps * g.cv_cin || !g.zero_row ||
                   batches != (g.cv_taps_total ? g.cv_taps_total / g.cv_ntaps : 1) || (g.cv_taps_total && g.cv_taps_total % g.cv_ntaps))) return hipErrorInvalidValue;
    static const int dbg = [] { const char* e = getenv("TDX_H3_DEBUG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
#ifdef TDX_H3_STAMPS
    // diagnostic build: the 4th launch of every instantiation with >= 512 blocks dumps its per-block stamps to $TDX_H3_STAMPS_DIR
    static int stamp_calls = 0;
    const long nblk = (long)grid.x * grid.y;
    if (!g.stamps && getenv("TDX_H3_STAMPS_DIR") && nblk >= 512 && ++stamp_calls == 4) {
        unsigned long long* buf = nullptr;
        hipMalloc((void**)&buf, nblk * 64);
        hipMemsetAsync(buf, 0, nblk * 64, st);
        g.stamps = buf;
        g.tiles_n = g.tiles_n;      // (stamps are indexed by blockIdx.x: 2-D grids record batch 0 only)
        hipLaunchKernelGGL((gemm_h3_kernel<A_TR, B_TR, PAIRED, TWOSEG, Epi, VARIANT, A_CONV>), grid, dim3(H3_THREADS), H3_LDS + H3_LDS_EXTRA, st, g, epi);
        hipStreamSynchronize(st);
        std::vector<unsigned long long> hbuf(nblk * 8);
        hipMemcpy(hbuf.data(), buf, nblk * 64, hipMemcpyDeviceToHost);
        long ktot = 0;
        for (int i = 0; i < g.nseg; ++i) ktot += g.seg[i].K;
        char fn[512];
        snprintf(fn, sizeof fn, "%s/h3_%s_M%d_N%d_K%ld_z%d.bin", getenv("TDX_H3_STAMPS_DIR"), typeid(Epi).name(), g.M, g.N, ktot, batches);
        if (FILE* f = fopen(fn, "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
        hipFree(buf);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((gemm_h3_kernel<A_TR, B_TR, PAIRED, TWOSEG, Epi, VARIANT, A_CONV>), grid, dim3(H3_THREADS), H3_LDS + H3_LDS_EXTRA, st, g, epi);
    return hipGetLastError();
}

// row-major planes on both sides, one segment (nn.Linear)
template <bool PAIRED, class Epi, int VARIANT = 0>
inline hipError_t launch_gemm_h3(H3Args g, int batches, Epi epi, hipStream_t st) {
    return launch_gemm_h3x<false, false, PAIRED, false, Epi, VARIANT>(g, batches, epi, st);
}

// ---- split producers ---------------------------------------------------------------------------

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

// row-major planes with ONE caller-chosen power-of-two scale s (max|x|*s < 65504) for the whole tensor: for tensors with a
// known bound (ReLU20 activations) and for operands that need one scale along M (the taps of an implicit-GEMM
// convolution read different pixel rows into the same accumulator row).  K % 8 == 0; columns K..Kp-1 are zero.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void h3_split_rows_static_kernel(const float* __restrict__ x, long ld, unsigned char* __restrict__ planes,
                                                                  long R, int K, int Kp, float s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // (row, 8-value chunk)
    const int per = Kp / 8;
    if (i >= R * per) return;
    const long row = i / per;
    const int ch = (int)(i - row * per);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (ch * 8 < K) {
        const f32x4 a = ldg4(x + row * ld + ch * 8), b = ldg4(x + row * ld + ch * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    }
    h3_store_chunk(planes + row * (long)Kp * 4 + ch * 32, v, s);
}
inline hipError_t launch_h3_split_rows_static(const float* x, long ld, void* planes, long R, int K, float s, hipStream_t st) {
    const int Kp = (K + 15) / 16 * 16;
    const long n = R * (Kp / 8);
    hipLaunchKernelGGL(h3_split_rows_static_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ld, (unsigned char*)planes, R, K, Kp, s);
    return hipGetLastError();
}

// long rows (weights with K > 2048, e.g. a 3x3 convolution over 1024 channels): one 256-thread block per row,
// two passes over the row (max, then split).  K % 8 == 0, K % 16 == 0.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void h3_split_rows_long_kernel(const float* __restrict__ x, long ld, unsigned char* __restrict__ planes,
                                                                float* __restrict__ scale, int K) {
    __shared__ float wmax[4];
    const long row = blockIdx.x;
    const float* xr = x + row * ld;
    float mu = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) mu = fmaxf(mu, h3_absmax4(*reinterpret_cast<const float4*>(xr + k)));
    mu = h3_wave_max(mu);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mu;
    __syncthreads();
    mu = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float inv;
    const float s = h3_row_scale(mu, inv);
    for (int ch = threadIdx.x; ch < K / 8; ch += 256) {
        const f32x4 a = ldg4(xr + ch * 8), b = ldg4(xr + ch * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        h3_store_chunk(planes + row * (long)K * 4 + ch * 32, v, s);
    }
    if (threadIdx.x == 0) scale[row] = inv;
}
inline hipError_t launch_h3_split_rows_long(const float* x, long ld, void* planes, float* scale, long R, int K, hipStream_t st) {
    hipLaunchKernelGGL(h3_split_rows_long_kernel<0>, dim3((unsigned)R), dim3(256), 0, st, x, ld, (unsigned char*)planes, scale, K);
    return hipGetLastError();
}

// K-major producer (diagnostics / stand-alone ops): fp32 [K][N] (pitch ld floats, N % 128 == 0) -> K-major planes
// [k][N/32][2][32] with one scale: s (a power of two chosen by the caller: max|x|*s < 65504), or, when mx is
// given, the exact exponent-aligned scale of the float whose bits are mx[0] (its inverse is written to inv[0]).
// With Sp > 0 the k rows are [b][Sp] and row (b, t) reads source row b*S + t, zero for t >= S (group padding).
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void h3_split_kmajor_kernel(const float* __restrict__ x, long ld, unsigned char* __restrict__ planes,
                                                             long K, int N, float s, const unsigned* __restrict__ mx,
                                                             float* __restrict__ inv, int S, int Sp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // one thread per 8 consecutive n of one k row
    const int per = N / 8;
    if (i >= K * per) return;
    const long k = i / per;
    const int n = (int)(i - k * per) * 8;
    if (mx) {
        float iv;
        s = h3_row_scale(__uint_as_float(mx[0]), iv);
        if (i == 0) inv[0] = iv;
    }
    long src = k;
    bool ok = true;
    if (Sp > 0) { const long b = k / Sp; const int t = (int)(k - b * Sp); ok = t < S; src = b * S + t; }
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (ok) { a = ldg4(x + src * ld + n); b4 = ldg4(x + src * ld + n + 4); }
    const float v[8] = {a[0], a[1], a[2], a[3], b4[0], b4[1], b4[2], b4[3]};
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * s;
        const _Float16 t = (_Float16)xs;
        hi[j] = t; lo[j] = (_Float16)(xs - (float)t);
    }
    unsigned char* dst = planes + k * (4L * N) + (n >> 5) * 128 + ((n & 31) >> 3) * 16;
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 64) = lo;
}

// max |x| over n floats as float bits (atomicMax; mx zeroed by the caller)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void h3_absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ mx) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    m = h3_wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(mx, __float_as_uint(m));
}

inline hipError_t launch_h3_split_rows(const float* x, long ld, void* planes, float* scale, long R, int K, hipStream_t st) {
    const int Kp = (K + 15) / 16 * 16;
    hipLaunchKernelGGL(h3_split_rows_kernel<0>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, ld, (unsigned char*)planes, scale, R, K, Kp);
    return hipGetLastError();
}

}  // namespace tdx
