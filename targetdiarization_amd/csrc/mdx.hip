// mdx.hip — MDX-Net denoiser BODY on MI355X: the network between ConvTDFNet.stft and .istft that the
// reference runs as an opaque ONNX file through onnxruntime (`self.mdx_model.run(None, {"input": mix_spec})`,
// AudioProcessor.py:224-241, :630).  Architecture = KUIELab ConvTDFNet / TFC-TDF v2 [upstream-recall, third-party: parity
// unpinned — oracle/mdx_oracle.py restates the published forward]:
//   first_conv 1x1 (4 -> g) + BN + ReLU, transpose to [B, g, T, F];  n = L/2 levels of { TFC_TDF ; Conv2d 2x2 s2 (c -> c+g) + BN + ReLU };
//   bottleneck TFC_TDF;  n levels of { ConvTranspose2d 2x2 s2 (c -> c-g) + BN + ReLU ; * skip ; TFC_TDF };  transpose back, final 1x1 (g -> 4).
//   TFC_TDF(x): x = l x [Conv2d 3x3 + BN + ReLU];  x + ReLU(BN(Linear_{f/bn -> f}(ReLU(BN(Linear_{f -> f/bn}(x))))))   (Linear over frequency).
//
// Layout: NHWC fp32, rows = (b, t, f) with f fastest, channels contiguous (g % 32 == 0).  Every convolution / Linear runs on the exact
// fp32 MFMA core (gemm.hpp) with eval-mode BatchNorm folded on the host:
//   3x3 convs        implicit GEMM (CONV mode, 9 taps), ReLU epilogue
//   TDF Linears      batched GEMMs, one batch per (b, t): C[f', c] = W[f', f] x X_bt[f, c] — the weight is the row-major A operand, the
//                    activations of one time step the K-major B operand as they lie in memory; the output rows are NHWC pixel rows again
//   2x2 s2 conv      space-to-depth gather + 1x1 GEMM;   2x2 s2 transposed conv: 1x1 GEMM whose epilogue scatters the four taps
//                    (depth-to-space), applies BN + ReLU and multiplies the skip connection
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/tdx.h"
#include "gemm.hpp"
#include "tdx_common.hpp"

using namespace tdx;

namespace {

inline size_t al(size_t n) { return (n + 63) / 64 * 64; }
inline int up(int n, int m) { return (n + m - 1) / m * m; }

#define LAUNCH_CHECK()                                    \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return tdx::fail_hip(e__, __FILE__, __LINE__); \
    } while (0)
#define TRY(x) do { int rc__ = (x); if (rc__ != TDX_OK) return rc__; } while (0)

struct GW { size_t w = 0, b = 0; int N = 0, Npad = 0, K = 0; };            // folded GEMM weights [Npad][K] + bias [Npad]
struct TdfW { size_t w1 = 0, w2 = 0, b1 = 0, b2 = 0, s1 = 0, t1 = 0, s2 = 0, t2 = 0; int f = 0, fb = 0, fbp = 0; };
struct BlockW { std::vector<GW> conv; TdfW tdf; int c = 0, f = 0; };

// ---------------------------------------------------------------- epilogues
struct EpiBiasRelu {     // relu(v + b[n]), columns < nreal
    const float* b; float* out; long ld; int nreal;
    __device__ float col(int, int n) const { return b[n]; }
    __device__ EpiNone row(int, int) const { return EpiNone{}; }
    __device__ void store(int, int m, int n, float v, EpiNone, float c) const { if (n < nreal) out[(long)m * (int)ld + n] = fmaxf(v + c, 0.f); }
};
struct Col2 { float s, t; };
struct EpiTdf1 {         // t[z][m][n] = relu((v + b1[m]) * s[n] + t[n])       (Linear over f, BatchNorm over channels)
    const float* b1; const float* s; const float* t; float* out; long ld; long strideZ; int nreal;
    __device__ Col2 col(int, int n) const { return Col2{s[n], t[n]}; }
    __device__ float row(int, int m) const { return b1[m]; }
    __device__ void store(int z, int m, int n, float v, float r, Col2 c) const {
        if (n < nreal) out[(long)z * strideZ + (long)m * (int)ld + n] = fmaxf((v + r) * c.s + c.t, 0.f);
    }
};
struct EpiTdf2 {         // x[z][m][n] += relu((v + b2[m]) * s[n] + t[n])
    const float* b2; const float* s; const float* t; float* x; long ld; long strideZ; int nreal;
    __device__ Col2 col(int, int n) const { return Col2{s[n], t[n]}; }
    __device__ float row(int, int m) const { return b2[m]; }
    __device__ float aux(int z, int m, int n, float) const { return n < nreal ? x[(long)z * strideZ + (long)m * (int)ld + n] : 0.f; }
    __device__ void store(int z, int m, int n, float v, float r, Col2 c, float xo) const {
        if (n < nreal) x[(long)z * strideZ + (long)m * (int)ld + n] = xo + fmaxf((v + r) * c.s + c.t, 0.f);
    }
};
struct EpiUp {           // transposed 2x2 stride-2 conv: row m = input pixel (b, t, f), column r = tap * cg + n ->
                         // out[(b, 2t + kh, 2f + kw)][n] = relu(v + bias[n]) * skip[...]        (tap = kh * 2 + kw)
    const float* b; const float* skip; float* out; int Tin, Fin, cg;
    __device__ float col(int, int r) const { return b[r]; }           // (bias repeated per tap on the host)
    __device__ long row(int, int m) const {                           // destination pixel of tap 0
        const int f = m % Fin, bt = m / Fin, t = bt % Tin, bb = bt / Tin;
        return ((long)(bb * 2 * Tin + 2 * t) * (2 * Fin) + 2 * f);
    }
    __device__ long dst(long d0, int r) const {
        const int tap = r / cg, n = r - tap * cg;
        return (d0 + (long)(tap >> 1) * (2 * Fin) + (tap & 1)) * cg + n;
    }
    __device__ float aux(int, int, int r, long d0) const { return r < 4 * cg ? skip[dst(d0, r)] : 0.f; }
    __device__ void store(int, int, int r, float v, long d0, float c, float sk) const { if (r < 4 * cg) out[dst(d0, r)] = fmaxf(v + c, 0.f) * sk; }
};

// first_conv: spec [B][4][F][T] -> x[(b, t, f)][g] = relu(W[g][4] . spec[b][:, f, t] + bias)   (BN folded; includes the transpose)
__global__ __launch_bounds__(256) void mdx_first_kernel(const float* __restrict__ spec, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int F, int T, int g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // pixel with t fastest: coalesced reads of spec
    if (i >= (long)B * F * T) return;
    const int t = (int)(i % T), f = (int)((i / T) % F), b = (int)(i / ((long)T * F));
    float x[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) x[ci] = spec[(((long)b * 4 + ci) * F + f) * T + t];
    float* o = out + (((long)b * T + t) * F + f) * g;
    for (int n = 0; n < g; n += 4) {
        float4 r;
        float* rr = reinterpret_cast<float*>(&r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wn = w + (n + j) * 4;
            rr[j] = fmaxf(bias[n + j] + wn[0] * x[0] + wn[1] * x[1] + wn[2] * x[2] + wn[3] * x[3], 0.f);
        }
        *reinterpret_cast<float4*>(o + n) = r;
    }
}
// final_conv: x[(b, t, f)][g] -> spec[b][co][f][t] = W[4][g] . x + bias   (includes the transpose back)
__global__ __launch_bounds__(256) void mdx_final_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ spec, int B, int F, int T, int g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // pixel with t fastest: coalesced writes of spec
    if (i >= (long)B * F * T) return;
    const int t = (int)(i % T), f = (int)((i / T) % F), b = (int)(i / ((long)T * F));
    const float* p = x + (((long)b * T + t) * F + f) * g;
    float acc[4] = {bias[0], bias[1], bias[2], bias[3]};
    for (int n = 0; n < g; n += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + n);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const float* wc = w + co * g + n;
            acc[co] += wc[0] * v.x + wc[1] * v.y + wc[2] * v.z + wc[3] * v.w;
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) spec[(((long)b * 4 + co) * F + f) * T + t] = acc[co];
}
// space-to-depth for the 2x2 stride-2 convolution: xs[(b, t', f')][tap * c + ch] = x[(b, 2t' + kh, 2f' + kw)][ch]
__global__ __launch_bounds__(256) void mdx_s2d_kernel(const float* __restrict__ x, float* __restrict__ xs, long npix_out, int To, int Fo, int c) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // (output pixel, tap, channel quad)
    const int cq = c / 4;
    if (i >= npix_out * 4 * cq) return;
    const int q = (int)(i % cq), tap = (int)((i / cq) % 4);
    const long po = i / (4L * cq);
    const int f = (int)(po % Fo), t = (int)((po / Fo) % To);
    const long b = po / ((long)Fo * To);
    const long pin = ((b * 2 * To + 2 * t + (tap >> 1)) * (2L * Fo)) + 2 * f + (tap & 1);
    *reinterpret_cast<float4*>(xs + po * 4 * c + (long)tap * c + 4 * q) = *reinterpret_cast<const float4*>(x + pin * c + 4 * q);
}

}  // namespace

struct tdx_mdx {
    int device = 0;
    int L = 0, l = 0, g = 0, k = 0, bn = 0, dim_f = 0, dim_t = 0, n = 0;
    float* dev = nullptr;
    size_t first_w = 0, first_b = 0, fin_w = 0, fin_b = 0;
    std::vector<BlockW> enc, dec; BlockW bott;
    std::vector<GW> ds, us;
};

namespace {

struct MdxPlan { size_t xa, xb, skip[8], tmid, s2d, total; long P[9]; int c[9], f[9], t[9]; };
inline bool make_plan(const tdx_mdx* h, int B, MdxPlan& p) {
    if (B < 1) return false;
    size_t maxx = 0, maxt = 0, maxs = 0;
    for (int i = 0; i <= h->n; ++i) {
        p.c[i] = h->g * (i + 1); p.f[i] = h->dim_f >> i; p.t[i] = h->dim_t >> i;
        p.P[i] = (long)B * p.t[i] * p.f[i];
        maxx = std::max(maxx, (size_t)p.P[i] * p.c[i]);
        maxt = std::max(maxt, (size_t)B * p.t[i] * up(std::max(p.f[i] / h->bn, 1), 32) * p.c[i]);
        if (i > 0) maxs = std::max(maxs, (size_t)p.P[i] * 4 * p.c[i - 1]);
    }
    size_t off = 0;
    auto take = [&](size_t nfl) { size_t o = off; off += al(nfl + 256); return o; };      // (+256: K-major B tiles read 128 columns of a c-wide row)
    p.xa = take(maxx); p.xb = take(maxx);
    for (int i = 0; i < h->n; ++i) p.skip[i] = take((size_t)p.P[i] * p.c[i]);
    p.tmid = take(maxt); p.s2d = take(maxs);
    p.total = off;
    return true;
}

// x (NHWC, level lv) -> TFC (l convolutions, ping-pong x <-> y) -> + TDF, result in *px
int run_tfc_tdf(const tdx_mdx* h, const BlockW& bw, int B, int T, int F, float** px, float** py, float* tmid, hipStream_t st) {
    const int c = bw.c;
    const long M = (long)B * T * F;
    for (size_t j = 0; j < bw.conv.size(); ++j) {
        const GW& cw = bw.conv[j];
        GemmArgs g = make_args((int)M, cw.Npad, make_seg(*px, c, h->dev + cw.w, cw.K, c));
        g.n_valid = up(c, 32);
        g.cv_Hin = T; g.cv_Win = F; g.cv_Hout = T; g.cv_Wout = F; g.cv_stride = 1; g.cv_ntaps = 9; g.cv_cin = c;
        if (launch_gemm<false, false, false, false, EpiBiasRelu, 0, true>(g, 1, EpiBiasRelu{h->dev + cw.b, *py, c, c}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        std::swap(*px, *py);
    }
    const TdfW& td = bw.tdf;
    const int Z = B * T, Nn = up(c, 128);
    {   // t[z][f'][c] = relu(BN(W1 x_z + b1))
        GemmSeg s = make_seg(h->dev + td.w1, td.f, *px, c, td.f, 0, (long)td.f * c);
        GemmArgs g = make_args(td.fb, Nn, s);
        g.n_valid = up(c, 32);
        if (launch_gemm<false, true, false, false>(g, Z, EpiTdf1{h->dev + td.b1, h->dev + td.s1, h->dev + td.t1, tmid, c, (long)td.fbp * c, c}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    {   // x_z += relu(BN(W2 t_z + b2))
        GemmSeg s = make_seg(h->dev + td.w2, td.fbp, tmid, c, td.fbp, 0, (long)td.fbp * c, td.fb);
        GemmArgs g = make_args(td.f, Nn, s);
        g.n_valid = up(c, 32);
        if (launch_gemm<false, true, false, false>(g, Z, EpiTdf2{h->dev + td.b2, h->dev + td.s2, h->dev + td.t2, *px, c, (long)td.f * c, c}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
    }
    return TDX_OK;
}

}  // namespace

extern "C" {

int tdx_mdx_create(const tdx_mdx_config* cfg, const void* blob, size_t blob_bytes, int device, tdx_mdx** out) {
    if (!cfg || !blob || !out) return tdx::fail(TDX_E_INVALID, "tdx_mdx_create: null argument");
    const int L = cfg->num_blocks, l = cfg->l, g = cfg->g, k = cfg->k, bn = cfg->bn, dim_f = cfg->dim_f, dim_t = cfg->dim_t, n = L / 2;
    if (L < 1 || (L & 1) == 0 || n > 7 || l < 1 || l > 8 || g < 32 || g % 32 || k != 3 || bn < 1 || dim_f < 32 || dim_t < 1 ||
        (dim_f >> n) << n != dim_f || (dim_t >> n) << n != dim_t || (dim_f >> n) % 32 || (dim_f >> n) % bn)
        return tdx::fail(TDX_E_INVALID, "tdx_mdx_create: unsupported config (need odd num_blocks <= 15, k = 3, g % 32 == 0, dim_f / 2^n a multiple of 32 and of bn, dim_t / 2^n integral)");
    tdx::Blob bl;
    if (!bl.parse(blob, blob_bytes)) return tdx::fail(TDX_E_BLOB, "tdx_mdx_create: malformed TDXW blob");
    std::vector<float> host;
    bool ok = true; std::string missing;
    auto get = [&](const std::string& name, size_t nn) -> const float* {
        const tdx::BlobTensor* t = bl.find(name);
        if (!t || t->numel != nn) { ok = false; if (missing.empty()) missing = name; return nullptr; }
        return t->data;
    };
    auto opt = [&](const std::string& name, size_t nn) -> const float* {
        const tdx::BlobTensor* t = bl.find(name);
        if (t && t->numel != nn) { ok = false; if (missing.empty()) missing = name; return nullptr; }
        return t ? t->data : nullptr;
    };
    struct BN { std::vector<double> s, t; };
    auto bnfold = [&](const std::string& p, int c) -> BN {       // y = x * s + t
        BN r; r.s.assign(c, 1.0); r.t.assign(c, 0.0);
        const float *w = get(p + "weight", c), *b = get(p + "bias", c), *mu = get(p + "running_mean", c), *var = get(p + "running_var", c);
        if (ok) for (int i = 0; i < c; ++i) { r.s[i] = (double)w[i] / std::sqrt((double)var[i] + 1e-5); r.t[i] = (double)b[i] - (double)mu[i] * r.s[i]; }
        return r;
    };
    // Conv2d [N][cin][kh][kw] + bias + BN -> [Npad][taps][cin] (tap = kh * kw_count + kw), bias[Npad]
    auto fold_conv = [&](const std::string& p, const std::string& bnp, int N, int cin, int kh, int kw) -> GW {
        GW w; w.N = N; w.Npad = up(N, 128); w.K = kh * kw * cin;
        const float* W = get(p + "weight", (size_t)N * cin * kh * kw);
        const float* cb = get(p + "bias", N);
        const BN b = bnfold(bnp, N);
        w.w = host.size(); host.resize(host.size() + al((size_t)w.Npad * w.K), 0.f);
        w.b = host.size(); host.resize(host.size() + al(w.Npad), 0.f);
        if (!ok) return w;
        for (int nn = 0; nn < N; ++nn) {
            host[w.b + nn] = (float)((double)cb[nn] * b.s[nn] + b.t[nn]);
            for (int c = 0; c < cin; ++c)
                for (int t = 0; t < kh * kw; ++t)
                    host[w.w + ((size_t)nn * kh * kw + t) * cin + c] = (float)((double)W[((size_t)nn * cin + c) * kh * kw + t] * b.s[nn]);
        }
        return w;
    };
    auto fold_block = [&](const std::string& p, int c, int f) -> BlockW {
        BlockW bw; bw.c = c; bw.f = f;
        for (int j = 0; j < l && ok; ++j)
            bw.conv.push_back(fold_conv(p + "tfc.H." + std::to_string(j) + ".0.", p + "tfc.H." + std::to_string(j) + ".1.", c, c, 3, 3));
        TdfW& td = bw.tdf; td.f = f; td.fb = f / bn; td.fbp = up(td.fb, 32);
        const float* W1 = get(p + "tdf.0.weight", (size_t)td.fb * f);
        const float* B1 = opt(p + "tdf.0.bias", td.fb);
        const float* W2 = get(p + "tdf.3.weight", (size_t)f * td.fb);
        const float* B2 = opt(p + "tdf.3.bias", f);
        const BN n1 = bnfold(p + "tdf.1.", c), n2 = bnfold(p + "tdf.4.", c);
        td.w1 = host.size(); host.resize(host.size() + al((size_t)up(td.fb, 128) * f), 0.f);          // rows read in 128-row tiles
        td.w2 = host.size(); host.resize(host.size() + al((size_t)up(f, 128) * td.fbp), 0.f);
        td.b1 = host.size(); host.resize(host.size() + al(up(td.fb, 128)), 0.f);
        td.b2 = host.size(); host.resize(host.size() + al(up(f, 128)), 0.f);
        const int cp = up(c, 128);
        td.s1 = host.size(); host.resize(host.size() + al(cp), 0.f); td.t1 = host.size(); host.resize(host.size() + al(cp), 0.f);
        td.s2 = host.size(); host.resize(host.size() + al(cp), 0.f); td.t2 = host.size(); host.resize(host.size() + al(cp), 0.f);
        if (!ok) return bw;
        memcpy(host.data() + td.w1, W1, (size_t)td.fb * f * sizeof(float));
        for (int m = 0; m < f; ++m) memcpy(host.data() + td.w2 + (size_t)m * td.fbp, W2 + (size_t)m * td.fb, td.fb * sizeof(float));
        if (B1) memcpy(host.data() + td.b1, B1, td.fb * sizeof(float));
        if (B2) memcpy(host.data() + td.b2, B2, f * sizeof(float));
        for (int i = 0; i < c; ++i) {
            host[td.s1 + i] = (float)n1.s[i]; host[td.t1 + i] = (float)n1.t[i];
            host[td.s2 + i] = (float)n2.s[i]; host[td.t2 + i] = (float)n2.t[i];
        }
        return bw;
    };
    tdx_mdx* h = new tdx_mdx();
    h->L = L; h->l = l; h->g = g; h->k = k; h->bn = bn; h->dim_f = dim_f; h->dim_t = dim_t; h->n = n;
    {   // first_conv [g][4][1][1] + BN
        const float* W = get("first_conv.0.weight", (size_t)g * 4);
        const float* cb = get("first_conv.0.bias", g);
        const BN b = bnfold("first_conv.1.", g);
        h->first_w = host.size(); host.resize(host.size() + al((size_t)g * 4), 0.f);
        h->first_b = host.size(); host.resize(host.size() + al(g), 0.f);
        if (ok) for (int nn = 0; nn < g; ++nn) {
            host[h->first_b + nn] = (float)((double)cb[nn] * b.s[nn] + b.t[nn]);
            for (int c = 0; c < 4; ++c) host[h->first_w + nn * 4 + c] = (float)((double)W[nn * 4 + c] * b.s[nn]);
        }
    }
    int f = dim_f, c = g;
    for (int i = 0; i < n && ok; ++i) {
        h->enc.push_back(fold_block("encoding_blocks." + std::to_string(i) + ".", c, f));
        h->ds.push_back(fold_conv("ds." + std::to_string(i) + ".0.", "ds." + std::to_string(i) + ".1.", c + g, c, 2, 2));
        f /= 2; c += g;
    }
    if (ok) h->bott = fold_block("bottleneck_block.", c, f);
    for (int i = 0; i < n && ok; ++i) {
        // ConvTranspose2d weight [cin = c][cout = c - g][2][2] + bias + BN -> rows r = tap * cg + n of [4 cg (pad 128)][c], bias repeated per tap
        const std::string p = "us." + std::to_string(i) + ".";
        const int cg = c - g;
        GW w; w.N = 4 * cg; w.Npad = up(4 * cg, 128); w.K = c;
        const float* W = get(p + "0.weight", (size_t)c * cg * 4);
        const float* cb = get(p + "0.bias", cg);
        const BN b = bnfold(p + "1.", cg);
        w.w = host.size(); host.resize(host.size() + al((size_t)w.Npad * c), 0.f);
        w.b = host.size(); host.resize(host.size() + al(w.Npad), 0.f);
        if (ok) for (int tap = 0; tap < 4; ++tap)
            for (int nn = 0; nn < cg; ++nn) {
                host[w.b + tap * cg + nn] = (float)((double)cb[nn] * b.s[nn] + b.t[nn]);
                for (int ci = 0; ci < c; ++ci) host[w.w + ((size_t)tap * cg + nn) * c + ci] = (float)((double)W[((size_t)ci * cg + nn) * 4 + tap] * b.s[nn]);
            }
        h->us.push_back(w);
        f *= 2; c -= g;
        if (ok) h->dec.push_back(fold_block("decoding_blocks." + std::to_string(i) + ".", c, f));
    }
    {
        const float* W = get("final_conv.0.weight", (size_t)4 * g);
        const float* cb = get("final_conv.0.bias", 4);
        h->fin_w = host.size(); host.resize(host.size() + al((size_t)4 * g), 0.f);
        h->fin_b = host.size(); host.resize(host.size() + al(4), 0.f);
        if (ok) { memcpy(host.data() + h->fin_w, W, (size_t)4 * g * sizeof(float)); memcpy(host.data() + h->fin_b, cb, 4 * sizeof(float)); }
    }
    if (!ok) { delete h; return tdx::fail(TDX_E_BLOB, "tdx_mdx_create: tensor missing or wrong size: " + missing); }
    {
        const std::string extra = bl.first_unused();
        if (!extra.empty()) { delete h; return tdx::fail(TDX_E_BLOB, "tdx_mdx_create: unexpected tensor in the blob: " + extra); }
    }
    host.resize(host.size() + 4096, 0.f);       // (weight tiles are read in 128-row pieces)
    tdx::DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    h->device = device;
    e = hipMalloc(&h->dev, host.size() * sizeof(float));
    if (e != hipSuccess) { delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(h->dev); delete h; return tdx::fail_hip(e, __FILE__, __LINE__); }
    *out = h;
    return TDX_OK;
}

int tdx_mdx_destroy(tdx_mdx* h) {
    if (h) { if (h->dev) hipFree(h->dev); delete h; }
    return TDX_OK;
}

size_t tdx_mdx_workspace_bytes(const tdx_mdx* h, int B) {
    MdxPlan p;
    if (!h || !make_plan(h, B, p)) return 0;
    return p.total * sizeof(float);
}

double tdx_mdx_flops(const tdx_mdx* h, int B) {
    if (!h || B < 1) return 0.0;
    double fl = 0.0;
    auto blk = [&](int c, int f, int t) { const double P = (double)t * f; return 2.0 * P * (h->l * 9.0 * c * c + 2.0 * (double)(f / h->bn) * c); };
    int f = h->dim_f, t = h->dim_t, c = h->g;
    fl += 2.0 * t * f * 4.0 * c;
    for (int i = 0; i < h->n; ++i) { fl += blk(c, f, t) * 2.0; fl += 2.0 * (t / 2.0) * (f / 2.0) * 4.0 * c * (c + h->g) * 2.0; f /= 2; t /= 2; c += h->g; }
    fl += blk(c, f, t);
    fl += 2.0 * h->dim_t * h->dim_f * 4.0 * h->g;
    return fl * B;
}

int tdx_mdx_forward(tdx_mdx* h, const float* spec, int B, float* out, void* ws_, size_t ws_bytes, void* stream) {
    if (!h || !spec || !out || !ws_) return tdx::fail(TDX_E_INVALID, "tdx_mdx_forward: null argument");
    MdxPlan p;
    if (!make_plan(h, B, p)) return tdx::fail(TDX_E_INVALID, "tdx_mdx_forward: need B >= 1");
    if (ws_bytes < p.total * sizeof(float)) return tdx::fail(TDX_E_WORKSPACE, "tdx_mdx_forward: workspace too small");
    if (p.P[0] >= (1L << 31)) return tdx::fail(TDX_E_INVALID, "tdx_mdx_forward: too many pixels per call (row indices are 32-bit: split the batch)");
    tdx::DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return tdx::fail_hip(guard.err, __FILE__, __LINE__);
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)ws_;
    float *x = ws + p.xa, *y = ws + p.xb, *tmid = ws + p.tmid, *s2d = ws + p.s2d;
    const int F0 = h->dim_f, T0 = h->dim_t, g = h->g;
    hipLaunchKernelGGL(mdx_first_kernel, dim3((unsigned)((p.P[0] + 255) / 256)), dim3(256), 0, st, spec, h->dev + h->first_w, h->dev + h->first_b, x, B, F0, T0, g);
    LAUNCH_CHECK();
    for (int i = 0; i < h->n; ++i) {
        TRY(run_tfc_tdf(h, h->enc[i], B, p.t[i], p.f[i], &x, &y, tmid, st));
        float* sk = ws + p.skip[i];
        if (hipMemcpyAsync(sk, x, (size_t)p.P[i] * p.c[i] * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        // downsample: space-to-depth, then the 2x2 taps as one GEMM  [P_{i+1}, 4c] x [c + g, 4c]^T
        const long Po = p.P[i + 1];
        const int c = p.c[i];
        hipLaunchKernelGGL(mdx_s2d_kernel, dim3((unsigned)((Po * c + 255) / 256)), dim3(256), 0, st, x, s2d, Po, p.t[i + 1], p.f[i + 1], c);
        LAUNCH_CHECK();
        const GW& dw = h->ds[i];
        GemmArgs ga = make_args((int)Po, dw.Npad, make_seg(s2d, 4L * c, h->dev + dw.w, dw.K, dw.K));
        ga.n_valid = up(dw.N, 32);
        if (launch_gemm<false, false, false, false>(ga, 1, EpiBiasRelu{h->dev + dw.b, y, dw.N, dw.N}, st) != hipSuccess) return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        std::swap(x, y);
    }
    TRY(run_tfc_tdf(h, h->bott, B, p.t[h->n], p.f[h->n], &x, &y, tmid, st));
    for (int i = 0; i < h->n; ++i) {
        const int lv = h->n - i;                     // input level
        const GW& uw = h->us[i];
        const int c = p.c[lv], cg = p.c[lv - 1];
        GemmArgs ga = make_args((int)p.P[lv], uw.Npad, make_seg(x, c, h->dev + uw.w, c, c));
        ga.n_valid = up(uw.N, 32);
        if (launch_gemm<false, false, false, false>(ga, 1, EpiUp{h->dev + uw.b, ws + p.skip[lv - 1], y, p.t[lv], p.f[lv], cg}, st) != hipSuccess)
            return tdx::fail_hip(hipGetLastError(), __FILE__, __LINE__);
        std::swap(x, y);
        TRY(run_tfc_tdf(h, h->dec[i], B, p.t[lv - 1], p.f[lv - 1], &x, &y, tmid, st));
    }
    hipLaunchKernelGGL(mdx_final_kernel, dim3((unsigned)((p.P[0] + 255) / 256)), dim3(256), 0, st, x, h->dev + h->fin_w, h->dev + h->fin_b, out, B, F0, T0, g);
    LAUNCH_CHECK();
    return TDX_OK;
}

}  // extern "C"
