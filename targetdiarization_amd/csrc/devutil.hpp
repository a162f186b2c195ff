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
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float siluf_acc(float x) { return x / (1.0f + expf(-x)); }

}  // namespace tdx
