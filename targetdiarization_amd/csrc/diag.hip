// diag.hip — libtdx_diag.so: test hooks and timing diagnostics of the GEMM cores (declared in include/tdx_test.h).
// NOT part of the product library: libtdx.so exports only what include/tdx.h declares.  Used by tests/test_gpu_h3.py
// (numerics of the split-f16 x3 core in all operand modes) and by tools/ (timing variants, operand fill rates).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/tdx.h"
#include "../../include/tdx_test.h"
#include "gemm.hpp"
#include "gemm_x6.hpp"
#include "gemm_h3.hpp"
#include "tdx_common.hpp"

using namespace tdx;

namespace {
struct EpiStore {    // plain store, per-batch stride
    float* out; long ld; long strideZ;
    __device__ EpiNone col(int, int) const { return EpiNone{}; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int z, int m, int n, float v, EpiNone, EpiNone) const { out[(long)z * strideZ + (long)m * (int)ld + n] = v; }
};
struct EpiBias { const float* b; float* out; long ld;   // b may be null
    __device__ float col(int, int n) const { return b ? b[n] : 0.f; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { out[(long)m * (int)ld + n] = v + c; }
    __device__ float* ptr(int, int m, int n) const { return out + (long)m * (int)ld + n; }
    __device__ long ldm() const { return ld; }
    __device__ void put(float* p, float v, EpiNone, float c) const { *p = v + c; } };
}  // namespace

extern "C" {

const char* tdx_diag_last_error(void) { return tdx::last_error().c_str(); }

// timing-only diagnostic: see gemm.hpp VARIANT
int tdx_linear_variant(const float* a, const float* w, int M, int N, int K, float* c, int variant, void* stream) {
    GemmArgs g = make_args(M, N, make_seg(a, K, w, K, K));
    EpiStore e{c, (long)N, 0};
    hipError_t r;
    if (variant == 1) r = launch_gemm<false, false, false, false, EpiStore, 1>(g, 1, e, (hipStream_t)stream);
    else if (variant == 2) r = launch_gemm<false, false, false, false, EpiStore, 2>(g, 1, e, (hipStream_t)stream);
    else if (variant == 3) r = launch_gemm<false, false, false, false, EpiStore, 3>(g, 1, e, (hipStream_t)stream);
    else if (variant == 4) r = launch_gemm<false, false, false, false, EpiStore, 4>(g, 1, e, (hipStream_t)stream);
    else if (variant == 6) r = launch_gemm_x6<false>(g, e, (hipStream_t)stream);
    else r = launch_gemm<false, false, false, false, EpiStore, 0>(g, 1, e, (hipStream_t)stream);
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}

// diagnostics for the split-f16 x3 core: tests/test_gpu_h3.py, tools/h3_test.py
int tdx_h3_split_rows(const float* x, long ld, void* planes, float* scale, long R, int K, void* stream) {
    if (K % 8 || K > 2048) return tdx::fail(TDX_E_INVALID, "tdx_h3_split_rows: need K%8==0, K<=2048");
    hipError_t r = tdx::launch_h3_split_rows(x, ld, planes, scale, R, K, (hipStream_t)stream);
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}
int tdx_h3_split_kmajor(const float* x, long ld, void* planes, long K, int N, float s, void* stream) {
    if (N % 128) return tdx::fail(TDX_E_INVALID, "tdx_h3_split_kmajor: need N%128==0");
    const long n = K * (N / 8);
    hipLaunchKernelGGL(tdx::h3_split_kmajor_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld,
                       (unsigned char*)planes, K, N, s, (const unsigned*)nullptr, (float*)nullptr, 0, 0);
    hipError_t r = hipGetLastError();
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}
// mode bit 0: A in K-major planes [K][M] (sa = one scale), bit 1: B in K-major planes [K][N] (sb = one scale)
int tdx_h3_gemm_x(int mode, const void* pa, const float* sa, const void* pb, const float* sb, float* c, int M, int N, int K, void* stream) {
    if (K % 16) return tdx::fail(TDX_E_INVALID, "tdx_h3_gemm_x: need K%16==0");
    tdx::H3Args g{};
    const bool atr = mode & 1, btr = mode & 2;
    g.seg[0] = tdx::h3_seg(pa, sa, atr ? 4L * M : 4L * K, pb, sb, btr ? 4L * N : 4L * K, K);
    if (atr) g.seg[0].sa_mul = 0;
    if (btr) g.seg[0].sb_mul = 0;
    g.nseg = 1; g.M = M; g.N = N;
    EpiBias e{nullptr, c, N};
    hipError_t r;
    if (atr && btr) r = tdx::launch_gemm_h3x<true, true, false, false>(g, 1, e, (hipStream_t)stream);
    else if (atr) r = tdx::launch_gemm_h3x<true, false, false, false>(g, 1, e, (hipStream_t)stream);
    else if (btr) r = tdx::launch_gemm_h3x<false, true, false, false>(g, 1, e, (hipStream_t)stream);
    else r = tdx::launch_gemm_h3x<false, false, false, false>(g, 1, e, (hipStream_t)stream);
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}
int tdx_h3_gemm_variant(const void* pa, const float* sa, const void* pb, const float* sb, const float* bias, float* c, int M, int N, int K, int variant, void* stream) {
    tdx::H3Args g{};
    g.seg[0] = tdx::h3_seg(pa, sa, 4L * K, pb, sb, 4L * K, K);
    g.nseg = 1; g.M = M; g.N = N;
    EpiBias e{bias, c, N};
    hipError_t r;
    if (variant == 1) r = tdx::launch_gemm_h3<false, EpiBias, 1>(g, 1, e, (hipStream_t)stream);
    else if (variant == 2) r = tdx::launch_gemm_h3<false, EpiBias, 2>(g, 1, e, (hipStream_t)stream);
    else if (variant == 3) r = tdx::launch_gemm_h3<false, EpiBias, 3>(g, 1, e, (hipStream_t)stream);
    else if (variant == 4) r = tdx::launch_gemm_h3<false, EpiBias, 4>(g, 1, e, (hipStream_t)stream);
    else if (variant == 9) r = tdx::launch_gemm_h3<false, EpiBias, 9>(g, 1, e, (hipStream_t)stream);
    else if (variant == 5) r = tdx::launch_gemm_h3<false, EpiBias, 5>(g, 1, e, (hipStream_t)stream);
    else if (variant == 6) r = tdx::launch_gemm_h3<false, EpiBias, 6>(g, 1, e, (hipStream_t)stream);
    else r = tdx::launch_gemm_h3<false, EpiBias, 0>(g, 1, e, (hipStream_t)stream);
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}
int tdx_h3_gemm(const void* pa, const float* sa, const void* pb, const float* sb, const float* bias, float* c, int M, int N, int K, void* stream) {
    if (K % 16) return tdx::fail(TDX_E_INVALID, "tdx_h3_gemm: need K%16==0");
    tdx::H3Args g{};
    g.seg[0] = tdx::h3_seg(pa, sa, 4L * K, pb, sb, 4L * K, K);
    g.nseg = 1; g.M = M; g.N = N;
    hipError_t r = tdx::launch_gemm_h3<false>(g, 1, EpiBias{bias, c, N}, (hipStream_t)stream);
    return r == hipSuccess ? TDX_OK : tdx::fail_hip(r, __FILE__, __LINE__);
}

}  // extern "C"

// ---- diagnostic: per-CU operand fill rate, LDS-DMA vs register staging (tools/fill_bench.py) -------------
namespace {
template <int MODE>      // 0: global_load_lds_dwordx4 ; 1: global_load_dwordx4 + ds_write_b128
__global__ __launch_bounds__(512, 2) void fill_bench_kernel(const unsigned char* __restrict__ src, long bytes_per_block, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (long)blockIdx.x * bytes_per_block;
    const long span = bytes_per_block / 32768;        // 32 KB tiles in this block's slab
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* tile = base + (long)(it % span) * 32768;
        unsigned char* dst = lds + (it & 3) * 32768;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + (wave * 4 + j) * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + (wave * 4 + j) * 1024), 16, 0, 0);
            if ((it & 1) == 1) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        } else {
            tdx::f32x4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const tdx::f32x4*>(tile + (wave * 4 + j) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<tdx::f32x4*>(dst + (wave * 4 + j) * 1024 + lane * 16) = r[j];
            if ((it & 1) == 1) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<const float*>(lds)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}
// GEMM-shaped fill: every block streams k-tiles of an A slab and a B slab (256 rows each, `stride` bytes per row) shared with
// the other blocks of its XCD, either as the row-major planes deliver them (MODE 1: a wave instruction covers 16 rows x 64 B)
// or as contiguous 16 KB tiles of the same slabs (MODE 0: 1 KB per wave instruction).
template <int MODE>
__global__ __launch_bounds__(512, 2) void fill_bench2_kernel(const unsigned char* __restrict__ src, int stride, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const long slab = 256L * stride;
    const unsigned char* a = src + (long)(xcd * 4 + ((local >> 3) & 3)) * slab;
    const unsigned char* b = src + 32 * slab + (long)((local >> 2) & 1) * slab;
    const unsigned char* base = wave < 4 ? a : b;
    const int kts = stride / 64;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int kt = it % kts;
        unsigned char* dst = lds + (it & 3) * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = (wave & 3) * 4 + j;
            const unsigned char* g = MODE == 0 ? base + (long)kt * 16384 + slot * 1024 + lane * 16
                                               : base + (long)(slot * 16 + (lane >> 2)) * stride + kt * 64 + (lane & 3) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(dst + (wave * 4 + j) * 1024), 16, 0, 0);
        }
        if ((it & 1) == 1) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<const float*>(lds)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}
}  // namespace
// column-strip streaming (the K^T[V|U] operand pattern): block = (row range, strip); per k row it fetches `seg` contiguous bytes
// of a `pitch`-byte row; 32 KB per iteration.  pitch == seg: a contiguous stream.
namespace {
__global__ __launch_bounds__(512, 2) void fill_bench3_kernel(const unsigned char* __restrict__ src, long pitch, int seg, long rows_per_block, int strips, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x % strips;
    const long row0 = (long)(blockIdx.x / strips) * rows_per_block;
    const int per_row = seg / 1024;                  // 1 KB instructions per row
    const int rows_per_it = 32 / per_row;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned char* dst = lds + (it & 3) * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave * 4 + j;              // instruction 0..31 of this iteration
            const long r = row0 + ((long)it * rows_per_it + i / per_row) % rows_per_block;
            const unsigned char* g = src + r * pitch + (long)strip * seg + (i % per_row) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
        if ((it & 1) == 1) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<const float*>(lds)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}
}  // namespace
extern "C" int tdx_fill_bench3(const void* src, long pitch, int seg, long rows_per_block, int strips, int blocks, int iters, float* sink, void* stream) {
    static bool set = false;
    if (!set) { hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_bench3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); set = true; }
    hipLaunchKernelGGL(fill_bench3_kernel, dim3(blocks), dim3(512), 131072, (hipStream_t)stream, (const unsigned char*)src, pitch, seg, rows_per_block, strips, iters, sink);
    return hipGetLastError() == hipSuccess ? TDX_OK : TDX_E_HIP;
}
extern "C" int tdx_fill_bench2(int mode, const void* src, int stride, int blocks, int iters, float* sink, void* stream) {
    static bool set = false;
    if (!set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_bench2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_bench2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        set = true;
    }
    if (mode == 0) hipLaunchKernelGGL(fill_bench2_kernel<0>, dim3(blocks), dim3(512), 131072, (hipStream_t)stream, (const unsigned char*)src, stride, iters, sink);
    else hipLaunchKernelGGL(fill_bench2_kernel<1>, dim3(blocks), dim3(512), 131072, (hipStream_t)stream, (const unsigned char*)src, stride, iters, sink);
    return hipGetLastError() == hipSuccess ? TDX_OK : TDX_E_HIP;
}
extern "C" int tdx_fill_bench(int mode, const void* src, long bytes_per_block, int blocks, int iters, float* sink, void* stream) {
    static bool set = false;
    if (!set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_bench_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_bench_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        set = true;
    }
    if (mode == 0) hipLaunchKernelGGL(fill_bench_kernel<0>, dim3(blocks), dim3(512), 131072, (hipStream_t)stream, (const unsigned char*)src, bytes_per_block, iters, sink);
    else hipLaunchKernelGGL(fill_bench_kernel<1>, dim3(blocks), dim3(512), 131072, (hipStream_t)stream, (const unsigned char*)src, bytes_per_block, iters, sink);
    return hipGetLastError() == hipSuccess ? TDX_OK : TDX_E_HIP;
}
