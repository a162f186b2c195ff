// devutil.hpp — small device helpers shared by the translation units of libtdx.so.
#pragma once
#include <hip/hip_runtime.h>

namespace tdx {

// cross-lane helpers on DPP (VALU, a few cycles) instead of ds_bpermute (__shfl_xor: an LDS round trip of ~100 cycles per
// step, and the steps of a reduction depend on each other)
__device__ __forceinline__ int h3_laneid() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ float h3_dpp(float v, int ctrl_sel) {
    switch (ctrl_sel) {
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    default: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    }
}
__device__ __forceinline__ float h3_lane(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// max over the 16 lanes of each DPP row (every lane gets its row's result)
__device__ __forceinline__ float h3_row16_max(float v) {
    v = fmaxf(v, h3_dpp(v, 0)); v = fmaxf(v, h3_dpp(v, 1)); v = fmaxf(v, h3_dpp(v, 2)); v = fmaxf(v, h3_dpp(v, 3));
    return v;
}
__device__ __forceinline__ float h3_row16_sum(float v) {
    v += h3_dpp(v, 0); v += h3_dpp(v, 1); v += h3_dpp(v, 2); v += h3_dpp(v, 3);
    return v;
}
// over the 32 lanes of each half-wave (every lane gets its half's result)
__device__ __forceinline__ float h3_half_max(float v) {
    v = h3_row16_max(v);
    const float a = fmaxf(h3_lane(v, 0), h3_lane(v, 16)), b = fmaxf(h3_lane(v, 32), h3_lane(v, 48));
    return (h3_laneid() & 32) ? b : a;
}
__device__ __forceinline__ float h3_half_sum(float v) {
    v = h3_row16_sum(v);
    const float a = h3_lane(v, 0) + h3_lane(v, 16), b = h3_lane(v, 32) + h3_lane(v, 48);
    return (h3_laneid() & 32) ? b : a;
}
__device__ __forceinline__ float h3_wave_sum(float v) {
    v = h3_row16_sum(v);
    return (h3_lane(v, 0) + h3_lane(v, 16)) + (h3_lane(v, 32) + h3_lane(v, 48));
}

// sum over the 64 lanes of a (fully active) wave, the same value in every lane
__device__ __forceinline__ float wave_sum(float v) { return h3_wave_sum(v); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sigmoid / SiLU on the hardware transcendental units: exp(-x) = v_exp_f32(-x*log2e) (1 ulp),
// 1/(1+e) = v_rcp_f32 (1 ulp).  ~3e-7 relative, i.e. at the level of one fp32 rounding; used in
// GEMM epilogues where a libm-grade expf + IEEE divide would cost ~4x the instructions.
__device__ __forceinline__ float sigmoidf_acc(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float siluf_acc(float x) { return x * sigmoidf_acc(x); }

}  // namespace tdx
