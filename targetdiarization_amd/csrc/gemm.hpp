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

// Epilogue concept:  struct E { __device__ void operator()(int z, int m, int n, float v) const; }
// PAIRED epilogue:   struct E { __device__ void operator()(int z, int m, int c, float v0, float v1) const; }

template <bool A_KMAJOR, bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(GemmArgs g, Epi epi) {
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
    const int z = blockIdx.y;

    const int nwg = g.tiles_m * g.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int bm = t / g.tiles_n, bn = t % g.tiles_n;
    const int m0 = bm * BM;
    const int n0 = PAIRED ? bn * (BN / 2) : bn * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging registers: 4 float4 per operand per thread
    float4 ra[4], rb[4];

    for (int s = 0; s < g.nseg; ++s) {
        const GemmSeg sg = g.seg[s];
        const int z1 = z / sg.zdiv, z2 = z - z1 * sg.zdiv;
        const float* __restrict__ Ag = sg.A + (long)z1 * sg.strideA + (long)z2 * sg.strideA2;
        const float* __restrict__ Bg = sg.B + (long)z1 * sg.strideB + (long)z2 * sg.strideB2;
        const int nkt = sg.K / BK;
        const int kvalid = min(sg.K, max(0, sg.ktotal - z2 * sg.kchunk));

        auto load_tile = [&](int kt) {
            const int k0 = kt * BK;
            // ---- A ----
            if constexpr (!A_KMAJOR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + i * GEMM_THREADS;   // 0..1023
                    const int row = idx >> 3, c4 = idx & 7;
                    int m = m0 + row;
                    bool ok = m < g.M;
                    if constexpr (SHIFT) {
                        if (k0 < g.shift_k) {
                            ok = ok && (m % g.shift_S != 0);
                            m -= 1;
                        }
                    }
                    ra[i] = ok ? *reinterpret_cast<const float4*>(Ag + (long)m * sg.lda + k0 + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + i * GEMM_THREADS;
                    const int kr = idx >> 5, c4 = idx & 31;   // 32 k-rows x 32 float4
                    const int k = k0 + kr;
                    const bool ok = k < kvalid;
                    ra[i] = ok ? *reinterpret_cast<const float4*>(Ag + (long)k * sg.lda + m0 + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            // ---- B ----
            if constexpr (!B_KMAJOR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + i * GEMM_THREADS;
                    const int row = idx >> 3, c4 = idx & 7;
                    int n;
                    if constexpr (PAIRED) n = (row < BN / 2) ? n0 + row : g.pair_off + n0 + row - BN / 2;
                    else n = n0 + row;
                    rb[i] = *reinterpret_cast<const float4*>(Bg + (long)n * sg.ldb + k0 + c4 * 4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + i * GEMM_THREADS;
                    const int kr = idx >> 5, c4 = idx & 31;
                    const int k = k0 + kr;
                    const int nl = c4 * 4;
                    int n;
                    if constexpr (PAIRED) n = (nl < BN / 2) ? n0 + nl : g.pair_off + n0 + nl - BN / 2;
                    else n = n0 + nl;
                    const bool ok = k < kvalid;
                    rb[i] = ok ? *reinterpret_cast<const float4*>(Bg + (long)k * sg.ldb + n)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * GEMM_THREADS;
                if constexpr (!A_KMAJOR) *reinterpret_cast<float4*>(As + (idx >> 3) * P + (idx & 7) * 4) = ra[i];
                else *reinterpret_cast<float4*>(As + (idx >> 5) * BM + (idx & 31) * 4) = ra[i];
                if constexpr (!B_KMAJOR) *reinterpret_cast<float4*>(Bs + (idx >> 3) * P + (idx & 7) * 4) = rb[i];
                else *reinterpret_cast<float4*>(Bs + (idx >> 5) * BN + (idx & 31) * 4) = rb[i];
            }
        };

        load_tile(0);
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();          // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (kt + 1 < nkt) load_tile(kt + 1);   // in flight during the MFMA phase
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                float a[2][4], b[2][4];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int ml = wm * 64 + tm * 32 + l31;
                    if constexpr (!A_KMAJOR) {
                        const float4 v = *reinterpret_cast<const float4*>(As + ml * P + kc * 8 + h * 4);
                        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[tm][j] = As[(kc * 8 + h * 4 + j) * BM + ml];
                    }
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int nl = tn * 64 + wn * 32 + l31;
                    if constexpr (!B_KMAJOR) {
                        const float4 v = *reinterpret_cast<const float4*>(Bs + nl * P + kc * 8 + h * 4);
                        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) b[tn][j] = Bs[(kc * 8 + h * 4 + j) * BN + nl];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
            }
        }
        __syncthreads();   // before the next segment overwrites LDS
    }

    // ---- epilogue ----
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < g.M) {
                if constexpr (PAIRED) {
                    epi(z, m, n0 + wn * 32 + l31, acc[tm][0][r], acc[tm][1][r]);
                } else {
                    epi(z, m, n0 + wn * 32 + l31, acc[tm][0][r]);
                    epi(z, m, n0 + 64 + wn * 32 + l31, acc[tm][1][r]);
                }
            }
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, bool PAIRED, bool SHIFT, class Epi>
inline hipError_t launch_gemm(GemmArgs g, int batches, Epi epi, hipStream_t st) {
    g.tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
    g.tiles_n = PAIRED ? g.N / (GEMM_BN / 2) : g.N / GEMM_BN;
    dim3 grid(g.tiles_m * g.tiles_n, batches, 1);
    hipLaunchKernelGGL((gemm_f32_kernel<A_KMAJOR, B_KMAJOR, PAIRED, SHIFT, Epi>), grid, dim3(GEMM_THREADS), 0, st, g, epi);
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
    g.seg[0] = s0; g.seg[1] = s0; g.nseg = 1; g.M = M; g.N = N;
    g.pair_off = 0; g.shift_k = 0; g.shift_S = 1;
    return g;
}

}  // namespace tdx
