// HBM-bound kernels of the UNet forward path: GroupNorm finalize, affine+SiLU+residual, channel
// LayerNorm, token LayerNorm / Linear, time embedding, score map, channel gather.  gfx950, fp32.
#include "common.h"

typedef float floatx2 __attribute__((ext_vector_type(2)));

thread_local char g_idiff_err[512] = "";

extern "C" const char* idiff_last_error(void) { return g_idiff_err; }
std::atomic<unsigned long long> g_idiff_launches{0};
extern "C" int idiff_version(void) { return 3; }
extern "C" int64_t idiff_launch_count(void) { return (int64_t)g_idiff_launches.load(std::memory_order_relaxed); }
extern "C" int idiff_device_info(int* num_cu, int* wave_size, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return IDIFF_OK;
}

namespace {

inline int grid_for(long long n, int per_block, int cap = 256 * 16) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm finalize: one 256-thread workgroup per (b, group); fp64 reduction of the conv partials in a fixed order (thread-strided
// partial sums, butterfly per wave, the four waves combined in order) -> deterministic and independent of the batch around b
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ stats, int ntiles, int C, int groups, int HW,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ film, long long film_ld, float eps,
                                                          float* __restrict__ out_a, float* __restrict__ out_b,
                                                          float* __restrict__ mean_rstd) {
    __shared__ double wsum[4][2];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sp = stats + (long long)b * ntiles * C * 2 + (long long)g * cpg * 2;
    double s = 0.0, q = 0.0;
    // thread = (channel of the group, tile phase): no division in the loop, one 8-byte load per partial
    const int cl = tid % cpg, tph = tid / cpg, tstep = 256 / cpg;
    if (tph < tstep)
        for (int t = tph; t < ntiles; t += tstep) {
            const floatx2 v = *reinterpret_cast<const floatx2*>(sp + ((long long)t * C + cl) * 2);
            s += (double)v.x;
            q += (double)v.y;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) wsum[wave][0] = s, wsum[wave][1] = q;
    __syncthreads();
    s = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
    q = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
    const double cnt = (double)cpg * (double)HW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    if (mean_rstd && tid == 0) {
        mean_rstd[((long long)b * groups + g) * 2 + 0] = (float)mean;
        mean_rstd[((long long)b * groups + g) * 2 + 1] = (float)rstd;
    }
    for (int i = tid; i < cpg; i += 256) {
        const int c = g * cpg + i;
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        float a = (float)rstd * ga;
        float bb = be - (float)mean * a;
        if (film) {
            const float sc = 1.f + film[(long long)b * film_ld + c];
            const float sh = film[(long long)b * film_ld + C + c];
            a *= sc;
            bb = bb * sc + sh;
        }
        out_a[(long long)b * C + c] = a;
        out_b[(long long)b * C + c] = bb;
    }
}

// out = silu(a*h+b) + res + vec ; planes of HW elements, float4 when HW % 4 == 0
template <bool VEC>
__global__ __launch_bounds__(256) void affine_silu_add_kernel(const float* __restrict__ h, long long hbs, const float* __restrict__ a,
                                                              const float* __restrict__ bcoef, const float* __restrict__ res,
                                                              long long rbs, const float* __restrict__ vec, float* __restrict__ out,
                                                              long long obs, int B, int C, int HW) {
    constexpr int V = VEC ? 4 : 1;
    const int pv = HW / V;  // vectors per plane
    const long long total = (long long)B * C * pv;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long plane = i / pv;
        const int p = (int)(i - plane * pv) * V;
        const int b = (int)(plane / C), c = (int)(plane - (long long)b * C);
        const float aa = a ? a[plane] : 1.f, bb = a ? bcoef[plane] : 0.f;
        const float vv = vec ? vec[plane] : 0.f;
        const long long ho = (long long)b * hbs + (long long)c * HW + p;
        const long long oo = (long long)b * obs + (long long)c * HW + p;
        const long long ro = (long long)b * rbs + (long long)c * HW + p;
        if (VEC) {
            floatx4 x = *reinterpret_cast<const floatx4*>(h + ho);
            floatx4 r = {0.f, 0.f, 0.f, 0.f};
            if (res) r = *reinterpret_cast<const floatx4*>(res + ro);
            floatx4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = a ? silu_f(aa * x[k] + bb) : x[k];
                y[k] = t + r[k] + vv;
            }
            *reinterpret_cast<floatx4*>(out + oo) = y;
        } else {
            const float x = h[ho];
            const float t = a ? silu_f(aa * x + bb) : x;
            out[oo] = t + (res ? res[ro] : 0.f) + vv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Linear on few rows: one wave per (output feature n, chunk of 8 rows); lanes stride over K
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == IDIFF_ACT_SILU) return silu_f(v);
    if (act == IDIFF_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

constexpr int LIN_ROWS = 8;
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w, long long ldw,
                                                     const float* __restrict__ bias, const float* __restrict__ res, long long ldr,
                                                     const float* __restrict__ gscale, float* __restrict__ out, long long ldo, int R, int K,
                                                     int N, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = blockIdx.y * LIN_ROWS;
    if (n >= N) return;
    float acc[LIN_ROWS];
#pragma unroll
    for (int r = 0; r < LIN_ROWS; ++r) acc[r] = 0.f;
    const float* wr = w + (long long)n * ldw;
    for (int k = lane; k < K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int r = 0; r < LIN_ROWS; ++r) {
            if (r0 + r < R) acc[r] += act_apply(x[(long long)(r0 + r) * ldx + k], act_in) * wv;
        }
    }
#pragma unroll
    for (int r = 0; r < LIN_ROWS; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
        const float bv = bias ? bias[n] : 0.f;
        const float gs = gscale ? gscale[n] : 1.f;
#pragma unroll
        for (int r = 0; r < LIN_ROWS; ++r) {
            if (r0 + r < R) {
                float v = gs * (acc[r] + bv);
                if (res) v += res[(long long)(r0 + r) * ldr + n];
                out[(long long)(r0 + r) * ldo + n] = act_apply(v, act_out);
            }
        }
    }
}

// LayerNorm over rows of C elements: one wave per row (two-pass, like ATen's fp32 path)
// rpg > 0: the rows form groups of rpg rows with their own gamma / beta rows ([groups][C]: the same LayerNorm of several stacked decoders)
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out, long long ldo, int R,
                                                             int C, float eps, float* __restrict__ mean_rstd, int rpg = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    if (rpg > 0) {
        const int grp = r / rpg;
        if (gamma) gamma += (long long)grp * C;
        if (beta) beta += (long long)grp * C;
    }
    const float* xr = x + (long long)r * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (mean_rstd && lane == 0) {
        mean_rstd[2 * r] = mean;
        mean_rstd[2 * r + 1] = rstd;
    }
    for (int c = lane; c < C; c += 64) {
        const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
        out[(long long)r * ldo + c] = (xr[c] - mean) * rstd * g + bb;
    }
}

__global__ void time_embed_kernel(const float* __restrict__ t, const float* __restrict__ freqs, int B, int dim, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i % dim, half = dim / 2;
    const int f = j < half ? j : j - half;
    const float freq = freqs ? freqs[f] : expf((float)f * (-logf(10000.0f) / (float)(half - 1)));
    const float arg = __fmul_rn(t[b], freq);
    out[i] = j < half ? sinf(arg) : cosf(arg);
}

// LayerNorm across channels of an NCHW map, two-pass.  Workgroup = 64 pixels; lane = pixel (coalesced along p), the four
// waves split the channels and combine their per-pixel partial sums through LDS (a thread-per-pixel layout leaves a 32x32
// map with 64 workgroups for the whole chip).
// JMAX > 0: the wave's channels of a pixel are held in registers (C <= 4 * JMAX): x is read once; the operations and their order are
// those of the JMAX = 0 form (three walks over x), so the results are the same bits.
template <int JMAX>
__global__ __launch_bounds__(256) void chan_layernorm_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out, long long obs, int C,
                                                             int HW, float eps, float* __restrict__ mean_rstd) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < HW;
    const float* xb = x + (long long)b * xbs + (ok ? p : 0);
    float xr[JMAX > 0 ? JMAX : 1];
    if (JMAX > 0) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) xr[j] = wave + 4 * j < C ? xb[(long long)(wave + 4 * j) * HW] : 0.f;
    }
    float s = 0.f;
    if (JMAX > 0) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (wave + 4 * j < C) s += xr[j];
    } else {
        for (int c = wave; c < C; c += 4) s += xb[(long long)c * HW];
    }
    part[wave][lane] = s;
    __syncthreads();
    const float mean = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)C;
    __syncthreads();
    float q = 0.f;
    if (JMAX > 0) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (wave + 4 * j < C) {
                const float d = xr[j] - mean;
                q += d * d;
            }
    } else {
        for (int c = wave; c < C; c += 4) {
            const float d = xb[(long long)c * HW] - mean;
            q += d * d;
        }
    }
    part[wave][lane] = q;
    __syncthreads();
    const float rstd = rsqrtf(((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)C + eps);
    if (!ok) return;
    if (mean_rstd && wave == 0) {
        mean_rstd[((long long)b * HW + p) * 2] = mean;
        mean_rstd[((long long)b * HW + p) * 2 + 1] = rstd;
    }
    float* ob = out + (long long)b * obs + p;
    if (JMAX > 0) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c = wave + 4 * j;
            if (c < C) ob[(long long)c * HW] = (xr[j] - mean) * rstd * gamma[c] + beta[c];
        }
    } else {
        for (int c = wave; c < C; c += 4) ob[(long long)c * HW] = (xb[(long long)c * HW] - mean) * rstd * gamma[c] + beta[c];
    }
}

// score map: normalized feature (per pixel over C) . normalized text vectors
constexpr int SM_KMAX = 8;
__global__ __launch_bounds__(256) void scoremap_kernel(const float* __restrict__ feat, long long fbs, const float* __restrict__ tv,
                                                       float* __restrict__ out, const int* __restrict__ idx, float* __restrict__ sel, int C,
                                                       int HW, int K) {
    extern __shared__ float tvn[];  // [K][C]
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < K; k += 4) {
        const float* tr = tv + ((long long)b * K + k) * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += tr[c] * tr[c];
        const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
        for (int c = lane; c < C; c += 64) tvn[k * C + c] = tr[c] / nrm;
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* fb = feat + (long long)b * fbs + p;
    float dot[SM_KMAX];
#pragma unroll
    for (int k = 0; k < SM_KMAX; ++k) dot[k] = 0.f;
    float n2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float f = fb[(long long)c * HW];
        n2 += f * f;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k)
            if (k < K) dot[k] += f * tvn[k * C + c];
    }
    const float nrm = fmaxf(sqrtf(n2), 1e-12f);
    const int ksel = idx ? idx[b] : -1;
#pragma unroll
    for (int k = 0; k < SM_KMAX; ++k) {
        if (k < K) {
            const float v = dot[k] / nrm;
            out[((long long)b * K + k) * HW + p] = v;
            if (k == ksel) sel[(long long)b * HW + p] = v;
        }
    }
}

// Streaming form: 4 consecutive pixels per thread (16-byte loads, a 1-KB row segment per wave and channel), four channels of loads
// in flight per thread.  The one-pixel-per-thread form above walked the channels with one dependent 4-byte load each and ran at
// 1.25 TB/s (profiles/r03/pmc_kernels); it stays for shapes whose rows are not 16-byte aligned.
__device__ __forceinline__ void scoremap4_body(const float* __restrict__ feat, long long fbs, const float* __restrict__ tv,
                                               float* __restrict__ out, const int* __restrict__ idx, float* __restrict__ sel, int C, int HW, int K,
                                               const int b, const int bx) {
    extern __shared__ float tvn[];  // [K][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < K; k += 4) {
        const float* tr = tv + ((long long)b * K + k) * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += tr[c] * tr[c];
        const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
        for (int c = lane; c < C; c += 64) tvn[k * C + c] = tr[c] / nrm;
    }
    __syncthreads();
    const int p = (bx * blockDim.x + threadIdx.x) * 4;
    if (p >= HW) return;  // HW % 4 == 0
    const float* fb = feat + (long long)b * fbs + p;
    floatx4 dot[SM_KMAX];
#pragma unroll
    for (int k = 0; k < SM_KMAX; ++k) dot[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 n2 = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](const floatx4 f, int c) {  // same per-pixel operation order as the scalar kernel: c ascending
        n2 += f * f;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k)
            if (k < K) dot[k] += f * tvn[k * C + c];
    };
    int c = 0;
    for (; c + 8 <= C; c += 8) {  // eight 16-byte loads in flight per thread (16 waves per CU: 128 KB; four covered an HBM miss only just)
        floatx4 f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const floatx4*>(fb + (long long)(c + i) * HW);
#pragma unroll
        for (int i = 0; i < 8; ++i) step(f[i], c + i);
    }
    for (; c + 4 <= C; c += 4) {
        const floatx4 f0 = *reinterpret_cast<const floatx4*>(fb + (long long)c * HW);
        const floatx4 f1 = *reinterpret_cast<const floatx4*>(fb + (long long)(c + 1) * HW);
        const floatx4 f2 = *reinterpret_cast<const floatx4*>(fb + (long long)(c + 2) * HW);
        const floatx4 f3 = *reinterpret_cast<const floatx4*>(fb + (long long)(c + 3) * HW);
        step(f0, c), step(f1, c + 1), step(f2, c + 2), step(f3, c + 3);
    }
    for (; c < C; ++c) step(*reinterpret_cast<const floatx4*>(fb + (long long)c * HW), c);
    floatx4 rn;
#pragma unroll
    for (int e = 0; e < 4; ++e) rn[e] = fmaxf(sqrtf(n2[e]), 1e-12f);
    const int ksel = idx ? idx[b] : -1;
#pragma unroll
    for (int k = 0; k < SM_KMAX; ++k) {
        if (k < K) {
            const floatx4 v = floatx4{dot[k].x / rn.x, dot[k].y / rn.y, dot[k].z / rn.z, dot[k].w / rn.w};
            *reinterpret_cast<floatx4*>(out + ((long long)b * K + k) * HW + p) = v;
            if (k == ksel) *reinterpret_cast<floatx4*>(sel + (long long)b * HW + p) = v;
        }
    }
}
__global__ __launch_bounds__(256) void scoremap4_kernel(const float* __restrict__ feat, long long fbs, const float* __restrict__ tv,
                                                        float* __restrict__ out, const int* __restrict__ idx, float* __restrict__ sel, int C,
                                                        int HW, int K) {
    scoremap4_body(feat, fbs, tv, out, idx, sel, C, HW, K, blockIdx.y, blockIdx.x);
}
// Grouped launch (idiff_scoremap_grouped_fwd): the score maps of a net's four ScoreMapModules in ONE launch; blockIdx.z picks the level
// (own operands and shape, descriptors in the kernel arguments), blocks beyond a smaller level's pixels exit.  Same body, same bits.
struct ScoremapGroups {
    idiff_scoremap_group g[IDIFF_SCOREMAP_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void scoremap4_grouped_kernel(const ScoremapGroups args, const int* __restrict__ idx, int K) {
    const idiff_scoremap_group& d = args.g[blockIdx.z];
    if ((long long)blockIdx.x * 1024 >= d.HW) return;  // uniform
    scoremap4_body(d.feat, d.feat_bstride, d.tv, d.out, idx, d.sel, d.C, d.HW, K, blockIdx.y, blockIdx.x);
}

// The time-embedding MLP in ONE launch (idiff_time_mlp_fwd):  temb = W2 . GELU(W0 . [sin(t f) ; cos(t f)] + b0) + b2.
// grid (8, B): every workgroup evaluates the sinusoidal embedding and the whole first layer of its sample (thread = hidden unit: its
// weight row of `dim` floats in 16-byte loads, all in flight at once), then an eighth of the second layer's outputs (eight threads per
// output, 32 consecutive weights each, summed over the eight lanes by shuffles).  Every load of a layer is issued before the first
// use: the r05 first version (a wave per output, eight outputs in flight) took 71 us -- twice the three launches it replaced.
__global__ __launch_bounds__(256) void time_mlp_kernel(const float* __restrict__ t, const float* __restrict__ freqs, const float* __restrict__ w0,
                                                       const float* __restrict__ b0, const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ out, int nout) {
    constexpr int DIM = 64, HID = 256;
    __shared__ __attribute__((aligned(16))) float e0[DIM];
    __shared__ __attribute__((aligned(16))) float h1[HID];
    const int b = blockIdx.y, tid = threadIdx.x, half = DIM / 2;
    // this thread's first-layer row: requested before anything else
    floatx4 wr[DIM / 4];
#pragma unroll
    for (int i = 0; i < DIM / 4; ++i) wr[i] = reinterpret_cast<const floatx4*>(w0 + (long long)tid * DIM)[i];
    if (tid < DIM) {
        const int f = tid < half ? tid : tid - half;
        const float freq = freqs ? freqs[f] : expf((float)f * (-logf(10000.0f) / (float)(half - 1)));
        const float arg = __fmul_rn(t[b], freq);
        e0[tid] = tid < half ? sinf(arg) : cosf(arg);
    }
    // second layer: output n = 32 * blockIdx.x + tid / 8, K range 32 * (tid & 7) .. + 31
    const int n = 32 * blockIdx.x + (tid >> 3), kq = tid & 7;
    floatx4 w2r[8];
    const bool live = n < nout;
#pragma unroll
    for (int i = 0; i < 8; ++i) w2r[i] = live ? reinterpret_cast<const floatx4*>(w2 + (long long)n * HID + kq * 32)[i] : floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < DIM / 4; ++i) {
        const floatx4 e = reinterpret_cast<const floatx4*>(e0)[i];
        acc += wr[i].x * e.x + wr[i].y * e.y + wr[i].z * e.z + wr[i].w * e.w;
    }
    h1[tid] = act_apply(acc + (b0 ? b0[tid] : 0.f), IDIFF_ACT_GELU);
    __syncthreads();
    float a2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const floatx4 h = reinterpret_cast<const floatx4*>(h1 + kq * 32)[i];
        a2 += w2r[i].x * h.x + w2r[i].y * h.y + w2r[i].z * h.z + w2r[i].w * h.w;
    }
    a2 += __shfl_xor(a2, 1, 64);
    a2 += __shfl_xor(a2, 2, 64);
    a2 += __shfl_xor(a2, 4, 64);
    if (live && kq == 0) out[(long long)b * nout + n] = a2 + (b2 ? b2[n] : 0.f);
}

__global__ void gather_channel_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ out, int B, int C, int HW) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
        out[i] = x[((long long)b * C + idx[b]) * HW + p];
    }
}

}  // namespace

extern "C" int idiff_gn_finalize(const float* stats, int ntiles, int B, int C, int groups, int HW, const float* gamma, const float* beta,
                                 const float* film, int64_t film_ld, float eps, float* out_a, float* out_b, float* mean_rstd,
                                 idiff_stream_t stream) {
    IDIFF_CHECK_ARG(stats && out_a && out_b, "gn_finalize: null pointer");
    IDIFF_CHECK_ARG(B > 0 && C > 0 && groups > 0 && C % groups == 0 && ntiles > 0 && HW > 0, "gn_finalize: bad dims");
    IDIFF_CHECK_ARG(C / groups <= 256, "gn_finalize: at most 256 channels per group (got %d)", C / groups);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, stats, ntiles, C, groups, HW, gamma, beta,
                       film, (long long)film_ld, eps, out_a, out_b, mean_rstd);
    IDIFF_CHECK_LAUNCH("gn_finalize");
    return IDIFF_OK;
}

extern "C" int idiff_affine_silu_add(const float* h, int64_t h_bstride, const float* a, const float* b, const float* res,
                                     int64_t res_bstride, const float* vec, float* out, int64_t out_bstride, int B, int C, int HW,
                                     idiff_stream_t stream) {
    IDIFF_CHECK_ARG(h && out && B > 0 && C > 0 && HW > 0, "affine_silu_add: bad args");
    IDIFF_CHECK_ARG((a == nullptr) == (b == nullptr), "affine_silu_add: a/b must both be set");
    const bool vec4 = (HW % 4 == 0) && (h_bstride % 4 == 0) && (out_bstride % 4 == 0) && (!res || res_bstride % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
    const long long total = (long long)B * C * (vec4 ? HW / 4 : HW);
    const int grid = grid_for(total, 256, 256 * 32);
    if (vec4)
        hipLaunchKernelGGL(affine_silu_add_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (long long)h_bstride, a, b, res,
                           (long long)res_bstride, vec, out, (long long)out_bstride, B, C, HW);
    else
        hipLaunchKernelGGL(affine_silu_add_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (long long)h_bstride, a, b, res,
                           (long long)res_bstride, vec, out, (long long)out_bstride, B, C, HW);
    IDIFF_CHECK_LAUNCH("affine_silu_add");
    return IDIFF_OK;
}

extern "C" int idiff_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                                const float* gscale, float* out, int64_t ldo, int R, int K, int N, int act_in, int act_out,
                                idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && w && out && R > 0 && K > 0 && N > 0, "linear_fwd: bad args");
    IDIFF_CHECK_ARG(ldx >= K && ldw >= K && ldo >= N, "linear_fwd: bad leading dims");
    dim3 grid((N + 3) / 4, (R + LIN_ROWS - 1) / LIN_ROWS);
    hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, w, (long long)ldw, bias, res,
                       (long long)ldr, gscale, out, (long long)ldo, R, K, N, act_in, act_out);
    IDIFF_CHECK_LAUNCH("linear_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_layernorm_rows_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo, int R,
                                        int C, float eps, float* mean_rstd, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && R > 0 && C > 0, "layernorm_rows: bad args");
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, gamma, beta, out,
                       (long long)ldo, R, C, eps, mean_rstd, 0);
    IDIFF_CHECK_LAUNCH("layernorm_rows");
    return IDIFF_OK;
}
extern "C" int idiff_layernorm_rows_g_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int64_t ldo, int R,
                                          int C, float eps, float* mean_rstd, int groups, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && gamma && beta && R > 0 && C > 0 && groups > 0 && R % groups == 0, "layernorm_rows_g: R must be a multiple of groups");
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, gamma, beta, out,
                       (long long)ldo, R, C, eps, mean_rstd, R / groups);
    IDIFF_CHECK_LAUNCH("layernorm_rows_g");
    return IDIFF_OK;
}

extern "C" int idiff_time_embed_fwd(const float* t, const float* freqs, int B, int dim, float* out, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(t && out && B > 0 && dim >= 4 && dim % 2 == 0, "time_embed: bad args");
    hipLaunchKernelGGL(time_embed_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, freqs, B, dim, out);
    IDIFF_CHECK_LAUNCH("time_embed");
    return IDIFF_OK;
}

extern "C" int idiff_chan_layernorm_fwd(const float* x, int64_t x_bstride, const float* gamma, const float* beta, float* out,
                                        int64_t out_bstride, int B, int C, int HW, float eps, float* mean_rstd, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && gamma && beta && B > 0 && C > 0 && HW > 0, "chan_layernorm: bad args");
    {
        const dim3 grid((HW + 63) / 64, B);
        if (C <= 128)
            hipLaunchKernelGGL(chan_layernorm_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)x_bstride, gamma,
                       beta, out, (long long)out_bstride, C, HW, eps, mean_rstd);
        else if (C <= 256)
            hipLaunchKernelGGL(chan_layernorm_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)x_bstride, gamma,
                       beta, out, (long long)out_bstride, C, HW, eps, mean_rstd);
        else
            hipLaunchKernelGGL(chan_layernorm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)x_bstride, gamma,
                       beta, out, (long long)out_bstride, C, HW, eps, mean_rstd);
    }
    IDIFF_CHECK_LAUNCH("chan_layernorm");
    return IDIFF_OK;
}

extern "C" int idiff_scoremap_fwd(const float* feat, int64_t feat_bstride, const float* tv, float* out, const int32_t* idx, float* sel,
                                  int B, int C, int HW, int K, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(feat && tv && out && B > 0 && C > 0 && HW > 0, "scoremap: bad args");
    IDIFF_CHECK_ARG(K > 0 && K <= SM_KMAX, "scoremap: K must be in 1..%d", SM_KMAX);
    IDIFF_CHECK_ARG((idx == nullptr) == (sel == nullptr), "scoremap: idx/sel must both be set");
    const bool vec4 = HW % 4 == 0 && feat_bstride % 4 == 0 && ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(out) |
                                                                reinterpret_cast<uintptr_t>(sel)) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL(scoremap4_kernel, dim3((HW / 4 + 255) / 256, B), dim3(256), (size_t)K * C * sizeof(float), (hipStream_t)stream, feat,
                           (long long)feat_bstride, tv, out, idx, sel, C, HW, K);
    else
        hipLaunchKernelGGL(scoremap_kernel, dim3((HW + 255) / 256, B), dim3(256), (size_t)K * C * sizeof(float), (hipStream_t)stream, feat,
                           (long long)feat_bstride, tv, out, idx, sel, C, HW, K);
    IDIFF_CHECK_LAUNCH("scoremap");
    return IDIFF_OK;
}

extern "C" int idiff_scoremap_grouped_fwd(const idiff_scoremap_group* groups, int ngroups, const int32_t* idx, int B, int K, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(groups && ngroups >= 1 && ngroups <= IDIFF_SCOREMAP_MAX_GROUPS, "scoremap_grouped: 1..%d groups", IDIFF_SCOREMAP_MAX_GROUPS);
    IDIFF_CHECK_ARG(B > 0 && K > 0 && K <= SM_KMAX, "scoremap_grouped: K must be in 1..%d", SM_KMAX);
    ScoremapGroups args;
    memset(&args, 0, sizeof(args));
    int gx = 0, cmax = 0;
    for (int i = 0; i < ngroups; ++i) {
        const idiff_scoremap_group& d = groups[i];
        IDIFF_CHECK_ARG(d.feat && d.tv && d.out && d.C > 0 && d.HW > 0, "scoremap_grouped: group %d: bad args", i);
        IDIFF_CHECK_ARG((idx == nullptr) == (d.sel == nullptr), "scoremap_grouped: group %d: idx/sel must both be set", i);
        IDIFF_CHECK_ARG(d.HW % 4 == 0 && d.feat_bstride % 4 == 0 &&
                            ((reinterpret_cast<uintptr_t>(d.feat) | reinterpret_cast<uintptr_t>(d.out) | reinterpret_cast<uintptr_t>(d.sel)) & 15) == 0,
                        "scoremap_grouped: group %d: the grouped launch is the 16-byte form (HW %% 4 == 0, aligned operands)", i);
        args.g[i] = d;
        gx = max(gx, (d.HW / 4 + 255) / 256);
        cmax = max(cmax, d.C);
    }
    hipLaunchKernelGGL(scoremap4_grouped_kernel, dim3(gx, B, ngroups), dim3(256), (size_t)K * cmax * sizeof(float), (hipStream_t)stream, args, idx, K);
    IDIFF_CHECK_LAUNCH("scoremap_grouped");
    return IDIFF_OK;
}

extern "C" int idiff_time_mlp_fwd(const float* t, const float* freqs, const float* w0, const float* b0, const float* w2, const float* b2, float* out,
                                  int B, int dim, int hid, int nout, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(t && w0 && w2 && out && B > 0 && nout > 0, "time_mlp: bad args");
    IDIFF_CHECK_ARG(dim == 64 && hid == 256 && nout <= 256, "time_mlp: the fused form is built for the UNet's 64 -> 256 -> <= 256 MLP (got %d -> %d -> %d)", dim, hid, nout);
    IDIFF_CHECK_ARG(((reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(w2)) & 15) == 0, "time_mlp: 16-byte aligned weights required");
    hipLaunchKernelGGL(time_mlp_kernel, dim3(8, B), dim3(256), 0, (hipStream_t)stream, t, freqs, w0, b0, w2, b2, out, nout);
    IDIFF_CHECK_LAUNCH("time_mlp");
    return IDIFF_OK;
}

extern "C" int idiff_gather_channel(const float* x, const int32_t* idx, float* out, int B, int C, int HW, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && idx && out && B > 0 && C > 0 && HW > 0, "gather_channel: bad args");
    hipLaunchKernelGGL(gather_channel_kernel, dim3(grid_for((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, out, B, C,
                       HW);
    IDIFF_CHECK_LAUNCH("gather_channel");
    return IDIFF_OK;
}

// ---- bf16 wire format of the gradient exchange (fp32 master gradients; BASELINE config c3) ------------------------------
namespace {
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t u = __builtin_bit_cast(uint32_t, x[i]);
        uint32_t r;
        if ((u & 0x7fffffffu) > 0x7f800000u) r = (u >> 16) | 0x40u;      // NaN stays NaN (quiet)
        else r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;                 // round to nearest, ties to even
        out[i] = (uint16_t)r;
    }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const uint16_t* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = __builtin_bit_cast(float, (uint32_t)x[i] << 16);
}
}  // namespace

extern "C" int idiff_f32_to_bf16(const float* x, uint16_t* out, int64_t n, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && n > 0, "f32_to_bf16: bad args");
    const long long g = (n + 255) / 256;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n);
    IDIFF_CHECK_LAUNCH("f32_to_bf16");
    return IDIFF_OK;
}
extern "C" int idiff_bf16_to_f32(const uint16_t* x, float* out, int64_t n, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && n > 0, "bf16_to_f32: bad args");
    const long long g = (n + 255) / 256;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n);
    IDIFF_CHECK_LAUNCH("bf16_to_f32");
    return IDIFF_OK;
}

// ---- many tensors -> one flat buffer in ONE launch (the optimizer's flat gradient buffer from the per-parameter gradients autograd
// produced: replaces one accumulate / copy launch per parameter tensor).  Segment table on the device: {src (NULL = zeros), dst
// offset, n, first block}; a block finds its segment by binary search over the first-block column and moves 4096 elements.
struct idiff_gather_seg {
    const float* src;
    long long dst;
    long long n;
    long long blk0;
};
__global__ __launch_bounds__(256) void gather_segments_kernel(const idiff_gather_seg* __restrict__ segs, int nseg, float* __restrict__ dst) {
    int lo = 0, hi = nseg - 1;
    const long long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= blk) lo = mid;
        else hi = mid - 1;
    }
    const idiff_gather_seg sg = segs[lo];
    const long long i0 = (blk - sg.blk0) * 4096;
    const long long i1 = i0 + 4096 < sg.n ? i0 + 4096 : sg.n;
    float* d = dst + sg.dst;
    if (sg.src) {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) d[i] = sg.src[i];
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) d[i] = 0.f;
    }
}
extern "C" int idiff_gather_segments(const void* segs_dev, int nseg, int64_t nblocks, float* dst, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(segs_dev && dst && nseg > 0 && nblocks > 0 && nblocks < (1ll << 31), "gather_segments: bad args");
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, static_cast<const idiff_gather_seg*>(segs_dev), nseg, dst);
    IDIFF_CHECK_LAUNCH("gather_segments");
    return IDIFF_OK;
}
