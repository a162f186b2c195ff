// devutil.hpp — small device helpers shared by the translation units of libtdx.so.
#pragma once
#include <hip/hip_runtime.h>

namespace tdx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
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
