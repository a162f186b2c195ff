/* tdx_test.h — exports of libtdx_diag.so: test hooks and timing diagnostics of the GEMM cores.
 *
 * NOT part of the product boundary (include/tdx.h / libtdx.so).  libtdx_diag.so is a separate library built
 * from csrc/diag.hip over the same kernel headers; tests/test_gpu_h3.py checks the numerics of the split-f16 x3
 * core through it and tools/ time its variants.  Same conventions as tdx.h (device pointers, void* stream,
 * 0 = ok). */
#ifndef TDX_TEST_H
#define TDX_TEST_H

#ifdef __cplusplus
extern "C" {
#endif

const char* tdx_diag_last_error(void);

/* fp32 rows x[R][ld] -> split-f16 planes [R][K/8][2][8] + one power-of-two scale per row (gemm_h3.hpp) */
int tdx_h3_split_rows(const float* x_dev, long ld, void* planes_dev, float* scale_dev, long R, int K, void* stream);
/* fp32 x[K][ld] -> K-major planes [K][N/32][2][32] with the static scale s */
int tdx_h3_split_kmajor(const float* x_dev, long ld, void* planes_dev, long K, int N, float s, void* stream);
/* C[M,N] = A B^T on the x3 core; mode bit 0: A in K-major planes (sa = one scale), bit 1: B in K-major planes */
int tdx_h3_gemm_x(int mode, const void* pa, const float* sa, const void* pb, const float* sb, float* c_dev,
                  int M, int N, int K, void* stream);
/* row-major planes, bias epilogue */
int tdx_h3_gemm(const void* pa, const float* sa, const void* pb, const float* sb, const float* bias_dev, float* c_dev,
                int M, int N, int K, void* stream);
/* timing variants of the x3 main loop (gemm_h3.hpp VARIANT: 0 product, 1 no loads, 2 no fragment reads, 3 neither,
 * 4 no barrier, 5 ping-pong, 6 v_mfma_f32_16x16x32_f16 instead of 32x32x16 (same flops), 9 loads never waited for); only 0 and 5
 * produce correct results.  (The narrow two-blocks-per-CU kernel and the pair-stage 16x16x32 kernel of round 2 — variants 10 / 16 /
 * 20 — lost inside the model and were removed in round 3; their measurements are in DESIGN.md §4.1e.) */
int tdx_h3_gemm_variant(const void* pa, const float* sa, const void* pb, const float* sb, const float* bias_dev,
                        float* c_dev, int M, int N, int K, int variant, void* stream);
/* timing variants of the fp32-MFMA core (gemm.hpp VARIANT) and the x6 core (variant 6) */
int tdx_linear_variant(const float* a_dev, const float* w_dev, int M, int N, int K, float* c_dev, int variant, void* stream);

/* per-CU operand fill rates (tools/fill_bench*.py) */
int tdx_fill_bench(int mode, const void* src_dev, long bytes_per_block, int blocks, int iters, float* sink_dev, void* stream);
int tdx_fill_bench2(int mode, const void* src_dev, int stride, int blocks, int iters, float* sink_dev, void* stream);
int tdx_fill_bench3(const void* src_dev, long pitch, int seg, long rows_per_block, int strips, int blocks, int iters,
                    float* sink_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDX_TEST_H */
