// HBM-bound kernels of the TRAINING path (models/drift_noise_model.py:242-312): data-gradient reshapes,
// GroupNorm/FiLM/SiLU backward, LayerNorm backward, activation gradients, reductions, score-map backward,
// losses (MSE + bilinear-resize pyramid), fused Adam.  gfx950, fp32, deterministic (no atomics).
#include <math.h>

#include "common.h"

namespace {

inline int bgrid(long long n, int per = 256, int cap = 4096) {
    long long g = (n + per - 1) / per;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// out[b,c,y,x] = sum of the 2x2 block of x (data gradient of nearest x2 upsampling)
__global__ void sumpool2x2_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int h, int w) {
    const long long n = planes * h * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xx = i % w, yy = (i / w) % h;
        const long long pl = i / ((long long)w * h);
        const float* p = x + pl * 4 * h * w + (long long)(2 * yy) * (2 * w) + 2 * xx;
        out[i] = (p[0] + p[1]) + (p[2 * w] + p[2 * w + 1]);
    }
}

// out[b,c,2y+p1,2x+p2] = x[b, c*4+p1*2+p2, y, x]   (data gradient of pixel_unshuffle(2))
__global__ void pixel_shuffle2_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int h, int w) {
    const long long n = (long long)B * C * 4 * h * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ox = i % (2 * w), oy = (i / (2 * w)) % (2 * h);
        const long long bc = i / ((long long)4 * h * w);
        const int c = bc % C;
        const long long b = bc / C;
        out[i] = x[((b * C + c) * 4 + (oy & 1) * 2 + (ox & 1)) * (long long)h * w + (long long)(oy >> 1) * w + (ox >> 1)];
    }
}

// per-plane sums: out[b*C+c] = sum_p x[b,c,p]  (one workgroup per plane, fixed order)
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ x, long long xbs, float* __restrict__ out, int C, int HW) {
    __shared__ float red[4];
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const float* p = x + (long long)b * xbs + (long long)c * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[c] (+)= sum_b in[b*C+c]
__global__ void batch_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += in[(long long)b * C + c];
    out[c] = accumulate ? out[c] + s : s;
}

__device__ __forceinline__ float silu_grad(float z) {
    const float sg = 1.0f / (1.0f + __expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

// GroupNorm(+FiLM)+SiLU backward, pass 1: per (b,c)  S1 = sum dz,  S2 = sum dz*xhat,  dz = dy*silu'(a*h+b); and, from the same reads,
// S3 = sum (h - mean) and S0 = sum dy: with them the plane sums of dh (the bias gradient of the conv that produced h) and of dy (the
// gradient of a per-(sample, channel) vector added behind the activation) need no pass of their own
__global__ __launch_bounds__(256) void gn_silu_bwd_reduce_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ h,
                                                                 long long hbs, const float* __restrict__ a, const float* __restrict__ bc,
                                                                 const float* __restrict__ mean_rstd, float* __restrict__ s4, int C, int groups,
                                                                 int HW) {
    __shared__ float red[4][4];
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const int g = c / (C / groups);
    const float mean = mean_rstd[((long long)b * groups + g) * 2], rstd = mean_rstd[((long long)b * groups + g) * 2 + 1];
    const float aa = a[blockIdx.x], bb = bc[blockIdx.x];
    const float* dp = dy + (long long)b * dybs + (long long)c * HW;
    const float* hp = h + (long long)b * hbs + (long long)c * HW;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s0 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float hv = hp[i], dv = dp[i];
        const float dz = dv * silu_grad(aa * hv + bb);
        s1 += dz;
        s2 += dz * (hv - mean) * rstd;
        s3 += hv - mean;
        s0 += dv;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    s0 = wave_sum(s0);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
        red[2][threadIdx.x >> 6] = s3;
        red[3][threadIdx.x >> 6] = s0;
    }
    __syncthreads();
    if (threadIdx.x < 4) s4[blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// pass 2 (tiny): group means A, Bq per (b,g); dgamma/dbeta (sum over b, fixed order); dfilm [B,2C]; optionally the plane sums
// dh_terms[b,c] = sum_pixels dh = rstd (g S1 - HW A - rstd Bq S3) (summed over b by batch_sum_kernel)  and  dy_sum[b,c] = S0
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ s4, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ film, long long film_ld, const float* __restrict__ mean_rstd,
                                       float* __restrict__ ab, float* __restrict__ gbc, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ dfilm, long long dfilm_ld, float* __restrict__ dh_terms, float* __restrict__ dy_sum, int B,
                                       int C, int groups, int HW, int accumulate) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int cpg = C / groups;
    auto group_means = [&](int b, int g, float& A, float& Bq) {
        float sa = 0.f, sb = 0.f;
        for (int i = 0; i < cpg; ++i) {
            const int c = g * cpg + i;
            const float sc = film ? 1.f + film[(long long)b * film_ld + c] : 1.f;
            const float gg = sc * (gamma ? gamma[c] : 1.f);
            sa += gg * s4[((long long)b * C + c) * 4];
            sb += gg * s4[((long long)b * C + c) * 4 + 1];
        }
        const float inv = 1.0f / ((float)cpg * (float)HW);
        A = sa * inv, Bq = sb * inv;
    };
    if (tid < B * groups) {  // group means
        float A, Bq;
        group_means(tid / groups, tid % groups, A, Bq);
        ab[tid * 2] = A;
        ab[tid * 2 + 1] = Bq;
    }
    if (tid < C) {  // parameter gradients
        float dg = 0.f, db = 0.f;
        for (int b = 0; b < B; ++b) {
            const float sc = film ? 1.f + film[(long long)b * film_ld + tid] : 1.f;
            dg += sc * s4[((long long)b * C + tid) * 4 + 1];
            db += sc * s4[((long long)b * C + tid) * 4];
        }
        if (dgamma) dgamma[tid] = accumulate ? dgamma[tid] + dg : dg;
        if (dbeta) dbeta[tid] = accumulate ? dbeta[tid] + db : db;
    }
    if (tid < B * C) {  // per-(b,c) scale used by the apply pass, and FiLM gradients
        const int b = tid / C, c = tid % C;
        const float sc = film ? 1.f + film[(long long)b * film_ld + c] : 1.f;
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        gbc[tid] = sc * ga;
        if (dfilm) {
            dfilm[(long long)b * dfilm_ld + c] = ga * s4[tid * 4 + 1] + be * s4[tid * 4];  // d scale
            dfilm[(long long)b * dfilm_ld + C + c] = s4[tid * 4];                          // d shift
        }
        if (dy_sum) dy_sum[tid] = s4[tid * 4 + 3];
        if (dh_terms) {  // this plane's share of sum dh: rstd (g S1 - HW A - rstd Bq S3); summed over b by idiff_batch_sum's kernel
            const int g = c / cpg;
            float A, Bq;
            group_means(b, g, A, Bq);
            const float rstd = mean_rstd[((long long)b * groups + g) * 2 + 1];
            dh_terms[tid] = rstd * (sc * ga * s4[tid * 4] - (float)HW * A - rstd * Bq * s4[tid * 4 + 2]);
        }
    }
}

// pass 3: dh = rstd * (dz*g - A - xhat*Bq)
__global__ __launch_bounds__(256) void gn_silu_bwd_apply_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ h,
                                                                long long hbs, const float* __restrict__ a, const float* __restrict__ bc,
                                                                const float* __restrict__ mean_rstd, const float* __restrict__ ab,
                                                                const float* __restrict__ gbc, float* __restrict__ dh, long long dhbs, int C,
                                                                int groups, int HW) {
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const int g = c / (C / groups);
    const float mean = mean_rstd[((long long)b * groups + g) * 2], rstd = mean_rstd[((long long)b * groups + g) * 2 + 1];
    const float A = ab[((long long)b * groups + g) * 2], Bq = ab[((long long)b * groups + g) * 2 + 1];
    const float aa = a[blockIdx.y], bb = bc[blockIdx.y], gg = gbc[blockIdx.y];
    const float* dp = dy + (long long)b * dybs + (long long)c * HW;
    const float* hp = h + (long long)b * hbs + (long long)c * HW;
    float* op = dh + (long long)b * dhbs + (long long)c * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float hv = hp[i];
        const float dz = dp[i] * silu_grad(aa * hv + bb);
        op[i] = rstd * (dz * gg - A - (hv - mean) * rstd * Bq);
    }
}

// dx = dy * act'(x)   (act: SILU or GELU)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long long n, int act) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        float g;
        if (act == IDIFF_ACT_SILU) {
            g = silu_grad(v);
        } else {
            const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
            g = cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
        }
        dx[i] = dy[i] * g;
    }
}
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int act) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = act == IDIFF_ACT_SILU ? silu_f(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}

// Column reductions over token rows (R <= a few hundred): workgroup = 16 columns x 16 row lanes; lane l adds rows l, l + 16, .. (all of
// a lane's loads independent), the 16 lane sums are added in lane order through LDS -> fixed order.  (r04: one thread per column had
// walked all R rows: 16-40 us per launch of pure load latency, ~330 launches per training iteration.)
constexpr int CS_COLS = 16, CS_LANES = 16;
__device__ __forceinline__ float colsum_lanes(float v, float (&red)[CS_LANES][CS_COLS + 1]) {
    const int cl = threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    red[rl][cl] = v;
    __syncthreads();
    float t = 0.f;
    if (rl == 0) {
#pragma unroll
        for (int l = 0; l < CS_LANES; ++l) t += red[l][cl];
    }
    __syncthreads();
    return t;  // valid on row lane 0
}
// out[n] (+)= sum_r x[r, n]   (blockIdx.y = group of R rows with its own output row: the stacked decoders' bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ out, int R, int N, int accumulate) {
    __shared__ float red[CS_LANES][CS_COLS + 1];
    x += (long long)blockIdx.y * R * ldx;
    out += (long long)blockIdx.y * N;
    const int n = blockIdx.x * CS_COLS + threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    float s = 0.f;
    if (n < N)
        for (int r = rl; r < R; r += CS_LANES) s += x[(long long)r * ldx + n];
    s = colsum_lanes(s, red);
    if (rl == 0 && n < N) out[n] = accumulate ? out[n] + s : s;
}

// out[r,n] = x[r,n] * g[n]
__global__ void scale_cols_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, int R, int N) {
    const long long n = (long long)R * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = x[i] * g[i % N];
}
// out[n] = sum_r x[r,n] * y[r,n]
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int R, int N) {
    __shared__ float red[CS_LANES][CS_COLS + 1];
    const int n = blockIdx.x * CS_COLS + threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    float s = 0.f;
    if (n < N)
        for (int r = rl; r < R; r += CS_LANES) s += x[(long long)r * N + n] * y[(long long)r * N + n];
    s = colsum_lanes(s, red);
    if (rl == 0 && n < N) out[n] = s;
}

// LayerNorm rows backward: dx (wave per row) ; dgamma/dbeta by a column pass
__global__ __launch_bounds__(256) void ln_rows_bwd_dx_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                             long long ldx, const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                             float* __restrict__ dx, long long lddx, int R, int C, int rpg = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    if (rpg > 0) gamma += (long long)(r / rpg) * C;  // grouped form: a gamma row per group of rpg rows
    const float mean = mean_rstd[2 * r], rstd = mean_rstd[2 * r + 1];
    const float* dyr = dy + (long long)r * lddy;
    const float* xr = x + (long long)r * ldx;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = dyr[c] * gamma[c];
        s1 += g;
        s2 += g * (xr[c] - mean) * rstd;
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        dx[(long long)r * lddx + c] = rstd * (dyr[c] * gamma[c] - s1 - xh * s2);
    }
}
__global__ __launch_bounds__(256) void ln_rows_bwd_param_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x, long long ldx,
                                                                const float* __restrict__ mean_rstd, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int R, int C, int accumulate) {
    __shared__ float red[CS_LANES][CS_COLS + 1];
    // blockIdx.y = group of R rows with its own dgamma / dbeta row
    dy += (long long)blockIdx.y * R * lddy;
    x += (long long)blockIdx.y * R * ldx;
    mean_rstd += (long long)blockIdx.y * R * 2;
    dgamma += (long long)blockIdx.y * C;
    dbeta += (long long)blockIdx.y * C;
    const int c = blockIdx.x * CS_COLS + threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    float dg = 0.f, db = 0.f;
    if (c < C)
        for (int r = rl; r < R; r += CS_LANES) {
            const float d = dy[(long long)r * lddy + c];
            dg += d * (x[(long long)r * ldx + c] - mean_rstd[2 * r]) * mean_rstd[2 * r + 1];
            db += d;
        }
    dg = colsum_lanes(dg, red);
    db = colsum_lanes(db, red);
    if (rl == 0 && c < C) {
        dgamma[c] = accumulate ? dgamma[c] + dg : dg;
        dbeta[c] = accumulate ? dbeta[c] + db : db;
    }
}

// channel LayerNorm backward (thread = pixel) ; writes dx and per-(b,c) partial sums are done by plane kernels
__global__ __launch_bounds__(256) void chan_ln_bwd_dx_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ x, long long xbs,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                             float* __restrict__ dx, long long dxbs, int C, int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float mean = mean_rstd[((long long)b * HW + p) * 2], rstd = mean_rstd[((long long)b * HW + p) * 2 + 1];
    const float* dp = dy + (long long)b * dybs + p;
    const float* xp = x + (long long)b * xbs + p;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float g = dp[(long long)c * HW] * gamma[c];
        s1 += g;
        s2 += g * (xp[(long long)c * HW] - mean) * rstd;
    }
    s1 /= (float)C;
    s2 /= (float)C;
    float* op = dx + (long long)b * dxbs + p;
    for (int c = 0; c < C; ++c) {
        const float xh = (xp[(long long)c * HW] - mean) * rstd;
        op[(long long)c * HW] = rstd * (dp[(long long)c * HW] * gamma[c] - s1 - xh * s2);
    }
}
// The same with the pixel's channels held in registers: workgroup = 64 pixels x 4 waves, wave w owns channels w, w + 4, ..; dy and x are
// read ONCE (the thread-per-pixel form above walks them twice with a 2-KB-per-thread footprint no cache holds: 5 passes over
// [B, 256, N] tensors instead of 3, 8.5 ms of a training iteration).  C <= 4 * JMAX.
template <int JMAX>
__global__ __launch_bounds__(256) void chan_ln_bwd_dx_reg_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ x, long long xbs,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                                 float* __restrict__ dx, long long dxbs, int C, int HW) {
    __shared__ float part[2][4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < HW;
    const int pc = ok ? p : 0;
    const float mean = mean_rstd[((long long)b * HW + pc) * 2], rstd = mean_rstd[((long long)b * HW + pc) * 2 + 1];
    const float* dp = dy + (long long)b * dybs + pc;
    const float* xp = x + (long long)b * xbs + pc;
    float g[JMAX], xh[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int c = wave + 4 * j;
        g[j] = c < C ? dp[(long long)c * HW] : 0.f;
        xh[j] = c < C ? xp[(long long)c * HW] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int c = wave + 4 * j;
        if (c < C) {
            g[j] *= gamma[c];
            xh[j] = (xh[j] - mean) * rstd;
            s1 += g[j];
            s2 += g[j] * xh[j];
        }
    }
    part[0][wave][lane] = s1;
    part[1][wave][lane] = s2;
    __syncthreads();
    const float S1 = ((part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane])) / (float)C;
    const float S2 = ((part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane])) / (float)C;
    if (!ok) return;
    float* op = dx + (long long)b * dxbs + p;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int c = wave + 4 * j;
        if (c < C) op[(long long)c * HW] = rstd * (g[j] - S1 - xh[j] * S2);
    }
}
// dx AND the parameter-gradient partials in one pass (dgamma[c] = sum dy*xhat, dbeta[c] = sum dy over batch and pixels): the workgroup
// walks TPW consecutive 64-pixel tiles of its sample; per tile every wave turns its [64 channels][64 pixels] products through LDS (row
// stride 65: conflict-free both ways) so that lane L sums the 64 pixels of channel wave + 4 L, and keeps the running sums in two
// registers; one partial per (sample, workgroup, channel) leaves for ws[((b * G + g) * C + c) * 2 + {0, 1}], reduced in a fixed order
// by pair_rows_sum_kernel.  Replaces the separate parameter pass (two more reads of dy and x).
constexpr int CLN_TPW = 8;
template <int JMAX>
__global__ __launch_bounds__(256) void chan_ln_bwd_fused_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ x, long long xbs,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                                float* __restrict__ dx, long long dxbs, float* __restrict__ ws, int C, int HW,
                                                                int tpw) {
    extern __shared__ float cln_smem[];
    float* part = cln_smem;                         // [2][4][64]
    float* gam = cln_smem + 512;                    // [4 * JMAX]
    float* tr = gam + 4 * JMAX + (threadIdx.x >> 6) * (JMAX * 65);  // this wave's [JMAX][65]
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 4 * JMAX; c += 256) gam[c] = c < C ? gamma[c] : 0.f;
    __syncthreads();
    float accg = 0.f, accb = 0.f;  // running sums of channel wave + 4 * lane (lanes < JMAX)
    const int tile0 = blockIdx.x * tpw;
    for (int t = 0; t < tpw; ++t) {
        const int p = (tile0 + t) * 64 + lane;
        if ((tile0 + t) * 64 >= HW) break;  // uniform
        const bool ok = p < HW;
        const int pc = ok ? p : 0;
        const float mean = mean_rstd[((long long)b * HW + pc) * 2], rstd = mean_rstd[((long long)b * HW + pc) * 2 + 1];
        const float* dp = dy + (long long)b * dybs + pc;
        const float* xp = x + (long long)b * xbs + pc;
        float d[JMAX], xh[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c = wave + 4 * j;
            d[j] = (c < C && ok) ? dp[(long long)c * HW] : 0.f;
            xh[j] = c < C ? xp[(long long)c * HW] : 0.f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int c = wave + 4 * j;
            xh[j] = c < C ? (xh[j] - mean) * rstd : 0.f;
            const float g = d[j] * gam[c];
            s1 += g;
            s2 += g * xh[j];
        }
        __syncthreads();  // the previous tile's readers of `part` are done
        part[wave * 64 + lane] = s1;
        part[256 + wave * 64 + lane] = s2;
        __syncthreads();
        const float S1 = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) / (float)C;
        const float S2 = ((part[256 + lane] + part[320 + lane]) + (part[384 + lane] + part[448 + lane])) / (float)C;
        if (ok) {
            float* op = dx + (long long)b * dxbs + p;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int c = wave + 4 * j;
                if (c < C) op[(long long)c * HW] = rstd * (d[j] * gam[c] - S1 - xh[j] * S2);
            }
        }
        // parameter partials: pixels onto the sum, channels onto the lanes (wave-private area; LDS is in order within a wave)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) tr[j * 65 + lane] = d[j] * xh[j];
        if (lane < JMAX) {
            float a = 0.f;
#pragma unroll 16
            for (int i = 0; i < 64; ++i) a += tr[lane * 65 + i];
            accg += a;
        }
#pragma unroll
        for (int j = 0; j < JMAX; ++j) tr[j * 65 + lane] = d[j];
        if (lane < JMAX) {
            float a = 0.f;
#pragma unroll 16
            for (int i = 0; i < 64; ++i) a += tr[lane * 65 + i];
            accb += a;
        }
    }
    const int c = wave + 4 * lane;
    if (lane < JMAX && c < C) {
        float* o = ws + (((long long)b * gridDim.x + blockIdx.x) * C + c) * 2;
        o[0] = accg;
        o[1] = accb;
    }
}
// dgamma[c] (+)= sum_r in[(r*C+c)*2], dbeta[c] (+)= sum_r in[(r*C+c)*2+1] over R rows: 16 channels x 16 row lanes per workgroup, lane sums
// added in lane order
__global__ __launch_bounds__(256) void pair_rows_sum_kernel(const float* __restrict__ in, float* __restrict__ o0, float* __restrict__ o1, int R, int C,
                                                            int accumulate) {
    __shared__ float red[CS_LANES][CS_COLS + 1];
    const int c = blockIdx.x * CS_COLS + threadIdx.x % CS_COLS, rl = threadIdx.x / CS_COLS;
    float a = 0.f, b2 = 0.f;
    if (c < C)
        for (int r = rl; r < R; r += CS_LANES) {
            a += in[((long long)r * C + c) * 2];
            b2 += in[((long long)r * C + c) * 2 + 1];
        }
    a = colsum_lanes(a, red);
    b2 = colsum_lanes(b2, red);
    if (rl == 0 && c < C) {
        o0[c] = accumulate ? o0[c] + a : a;
        o1[c] = accumulate ? o1[c] + b2 : b2;
    }
}
// per plane: out[(b*C+c)*2] = sum_p dy*xhat, [..+1] = sum_p dy   (xhat from per-pixel mean/rstd)
__global__ __launch_bounds__(256) void chan_ln_bwd_param_kernel(const float* __restrict__ dy, long long dybs, const float* __restrict__ x,
                                                                long long xbs, const float* __restrict__ mean_rstd, float* __restrict__ out, int C,
                                                                int HW) {
    __shared__ float red[2][4];
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const float* dp = dy + (long long)b * dybs + (long long)c * HW;
    const float* xp = x + (long long)b * xbs + (long long)c * HW;
    const float* mr = mean_rstd + (long long)b * HW * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float d = dp[i];
        s1 += d * (xp[i] - mr[2 * i]) * mr[2 * i + 1];
        s2 += d;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// dgamma[c] (+)= sum_b in[(b*C+c)*2], dbeta[c] (+)= sum_b in[(b*C+c)*2+1]
__global__ void pair_batch_sum_kernel(const float* __restrict__ in, float* __restrict__ o0, float* __restrict__ o1, int B, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b2 = 0.f;
    for (int b = 0; b < B; ++b) {
        a += in[((long long)b * C + c) * 2];
        b2 += in[((long long)b * C + c) * 2 + 1];
    }
    o0[c] = accumulate ? o0[c] + a : a;
    o1[c] = accumulate ? o1[c] + b2 : b2;
}

// L2 normalisation along channels of a map (thread = pixel): y = x / max(|x|, eps); also the norm
__global__ __launch_bounds__(256) void chan_normalize_fwd_kernel(const float* __restrict__ x, long long xbs, float* __restrict__ y,
                                                                 float* __restrict__ nrm, int C, int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* xp = x + (long long)b * xbs + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xp[(long long)c * HW] * xp[(long long)c * HW];
    const float n = fmaxf(sqrtf(s), 1e-12f);
    nrm[(long long)b * HW + p] = n;
    float* yp = y + (long long)b * C * HW + p;
    for (int c = 0; c < C; ++c) yp[(long long)c * HW] = xp[(long long)c * HW] / n;
}
// dx = (dy - y * <y, dy>) / n
__global__ __launch_bounds__(256) void chan_normalize_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ nrm, float* __restrict__ dx, long long dxbs, int C,
                                                                 int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* dp = dy + (long long)b * C * HW + p;
    const float* yp = y + (long long)b * C * HW + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dp[(long long)c * HW] * yp[(long long)c * HW];
    const float inv = 1.0f / nrm[(long long)b * HW + p];
    float* op = dx + (long long)b * dxbs + p;
    for (int c = 0; c < C; ++c) op[(long long)c * HW] = (dp[(long long)c * HW] - yp[(long long)c * HW] * s) * inv;
}

// The two above with a pixel's channels in registers (workgroup = 64 pixels x 4 waves, wave w owns channels w, w + 4, ..; C <= 4 * JMAX):
// every operand is read once.
template <int JMAX>
__global__ __launch_bounds__(256) void chan_normalize_fwd_reg_kernel(const float* __restrict__ x, long long xbs, float* __restrict__ y,
                                                                     float* __restrict__ nrm, int C, int HW) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < HW;
    const float* xp = x + (long long)b * xbs + (ok ? p : 0);
    float xr[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) xr[j] = wave + 4 * j < C ? xp[(long long)(wave + 4 * j) * HW] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) s += xr[j] * xr[j];
    part[wave][lane] = s;
    __syncthreads();
    const float n = fmaxf(sqrtf((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])), 1e-12f);
    if (!ok) return;
    if (wave == 0) nrm[(long long)b * HW + p] = n;
    float* yp = y + (long long)b * C * HW + p;
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (wave + 4 * j < C) yp[(long long)(wave + 4 * j) * HW] = xr[j] / n;
}
template <int JMAX>
__global__ __launch_bounds__(256) void chan_normalize_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                     const float* __restrict__ nrm, float* __restrict__ dx, long long dxbs, int C,
                                                                     int HW) {
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < HW;
    const int pc = ok ? p : 0;
    const float* dp = dy + (long long)b * C * HW + pc;
    const float* yp = y + (long long)b * C * HW + pc;
    float dr[JMAX], yr[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        dr[j] = wave + 4 * j < C ? dp[(long long)(wave + 4 * j) * HW] : 0.f;
        yr[j] = wave + 4 * j < C ? yp[(long long)(wave + 4 * j) * HW] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) s += dr[j] * yr[j];
    part[wave][lane] = s;
    __syncthreads();
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (!ok) return;
    const float inv = 1.0f / nrm[(long long)b * HW + p];
    float* op = dx + (long long)b * dxbs + p;
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (wave + 4 * j < C) op[(long long)(wave + 4 * j) * HW] = (dr[j] - yr[j] * s) * inv;
}

// out[b, idx[b], p] = x[b, 0, p], other channels zero
__global__ void scatter_channel_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ out, int B, int C, int HW) {
    const long long n = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, c = (i / HW) % C;
        const long long b = i / ((long long)C * HW);
        out[i] = c == idx[b] ? x[b * HW + p] : 0.f;
    }
}

// ---- losses ---------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False, antialias=False (torchvision Resize on tensors with antialias off)
__device__ __forceinline__ void bil_coef(int o, int in_size, float scale, int* i0, int* i1, float* w1) {
    float s = ((float)o + 0.5f) * scale - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in_size - 1) a = in_size - 1;
    *i0 = a;
    *i1 = a < in_size - 1 ? a + 1 : a;
    *w1 = s - (float)a;
}
__global__ void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, long long planes, int H, int W, int oh, int ow) {
    const long long n = planes * oh * ow;
    const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ox = i % ow, oy = (i / ow) % oh;
        const long long pl = i / ((long long)oh * ow);
        int y0, y1, x0, x1;
        float wy, wx;
        bil_coef(oy, H, sy, &y0, &y1, &wy);
        bil_coef(ox, W, sx, &x0, &x1, &wx);
        const float* p = x + pl * H * W;
        const float top = p[(long long)y0 * W + x0] * (1.f - wx) + p[(long long)y0 * W + x1] * wx;
        const float bot = p[(long long)y1 * W + x0] * (1.f - wx) + p[(long long)y1 * W + x1] * wx;
        out[i] = top * (1.f - wy) + bot * wy;
    }
}
// sum of squared differences, one partial per workgroup (fixed grid -> deterministic); d = 2*scale*(a-b) optional
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                             float* __restrict__ grad, long long n, float gscale) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s += d * d;
        if (grad) grad[i] = gscale * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sum_small_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += part[i];
        *out = s * scale;
    }
}

// Adam with L2-in-gradient weight decay (torch.optim.Adam, not AdamW), grad pre-scale (1/world for the flat
// all-reduce), bias corrections folded into step_size / bc2_sqrt on the host
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float lr, float beta1, float beta2, float eps, float wd, float gscale, float bc1, float bc2_sqrt) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gr = g[i] * gscale;
        const float pv = p[i];
        gr += wd * pv;
        const float mm = beta1 * m[i] + (1.f - beta1) * gr;
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mm / denom);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int idiff_sumpool2x2(const float* x, float* out, int64_t planes, int h, int w, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && planes > 0 && h > 0 && w > 0, "sumpool2x2: bad args");
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(bgrid(planes * h * w)), dim3(256), 0, ST, x, out, (long long)planes, h, w);
    IDIFF_CHECK_LAUNCH("sumpool2x2");
    return IDIFF_OK;
}
extern "C" int idiff_pixel_shuffle2(const float* x, float* out, int B, int C, int h, int w, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && B > 0 && C > 0 && h > 0 && w > 0, "pixel_shuffle2: bad args");
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(bgrid((long long)B * C * 4 * h * w)), dim3(256), 0, ST, x, out, B, C, h, w);
    IDIFF_CHECK_LAUNCH("pixel_shuffle2");
    return IDIFF_OK;
}
extern "C" int idiff_plane_sum(const float* x, int64_t x_bstride, float* out_bc, int B, int C, int HW, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out_bc && B > 0 && C > 0 && HW > 0, "plane_sum: bad args");
    hipLaunchKernelGGL(plane_sum_kernel, dim3(B * C), dim3(256), 0, ST, x, (long long)x_bstride, out_bc, C, HW);
    IDIFF_CHECK_LAUNCH("plane_sum");
    return IDIFF_OK;
}
extern "C" int idiff_batch_sum(const float* in_bc, float* out_c, int B, int C, int accumulate, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(in_bc && out_c && B > 0 && C > 0, "batch_sum: bad args");
    hipLaunchKernelGGL(batch_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, in_bc, out_c, B, C, accumulate);
    IDIFF_CHECK_LAUNCH("batch_sum");
    return IDIFF_OK;
}
extern "C" int idiff_gn_silu_bwd(const float* dy, int64_t dy_bstride, const float* h, int64_t h_bstride, const float* a, const float* b,
                                 const float* mean_rstd, const float* gamma, const float* beta, const float* film, int64_t film_ld, float* dh,
                                 int64_t dh_bstride, float* dgamma, float* dbeta, float* dfilm, int64_t dfilm_ld, float* ws, int B, int C,
                                 int groups, int HW, int accumulate, float* dh_sum, float* dy_sum, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && h && a && b && mean_rstd && dh && ws, "gn_silu_bwd: null pointer");
    IDIFF_CHECK_ARG(B > 0 && C > 0 && groups > 0 && C % groups == 0 && HW > 0, "gn_silu_bwd: bad dims");
    float* s4 = ws;                        // [B*C*4]
    float* ab = ws + (size_t)B * C * 4;    // [B*groups*2]
    float* gbc = ab + (size_t)B * groups * 2;  // [B*C]
    float* dh_terms = dh_sum ? gbc + (size_t)B * C : nullptr;  // [B*C]
    hipLaunchKernelGGL(gn_silu_bwd_reduce_kernel, dim3(B * C), dim3(256), 0, ST, dy, (long long)dy_bstride, h, (long long)h_bstride, a, b,
                       mean_rstd, s4, C, groups, HW);
    IDIFF_CHECK_LAUNCH("gn_silu_bwd_reduce");
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, ST, s4, gamma, beta, film, (long long)film_ld, mean_rstd,
                       ab, gbc, dgamma, dbeta, dfilm, (long long)dfilm_ld, dh_terms, dy_sum, B, C, groups, HW, accumulate);
    IDIFF_CHECK_LAUNCH("gn_bwd_finalize");
    if (dh_sum) {
        hipLaunchKernelGGL(batch_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, dh_terms, dh_sum, B, C, 0);
        IDIFF_CHECK_LAUNCH("gn_bwd_dh_sum");
    }
    int gx = (HW + 1023) / 1024;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gn_silu_bwd_apply_kernel, dim3(gx, B * C), dim3(256), 0, ST, dy, (long long)dy_bstride, h, (long long)h_bstride, a, b,
                       mean_rstd, ab, gbc, dh, (long long)dh_bstride, C, groups, HW);
    IDIFF_CHECK_LAUNCH("gn_silu_bwd_apply");
    return IDIFF_OK;
}
extern "C" int64_t idiff_gn_silu_bwd_ws_floats(int B, int C, int groups) { return (int64_t)B * C * 6 + (int64_t)B * groups * 2; }

extern "C" int idiff_act_fwd(const float* x, float* y, int64_t n, int act, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && y && n > 0 && (act == IDIFF_ACT_SILU || act == IDIFF_ACT_GELU), "act_fwd: bad args");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(bgrid(n)), dim3(256), 0, ST, x, y, (long long)n, act);
    IDIFF_CHECK_LAUNCH("act_fwd");
    return IDIFF_OK;
}
extern "C" int idiff_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && x && dx && n > 0 && (act == IDIFF_ACT_SILU || act == IDIFF_ACT_GELU), "act_bwd: bad args");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(bgrid(n)), dim3(256), 0, ST, dy, x, dx, (long long)n, act);
    IDIFF_CHECK_LAUNCH("act_bwd");
    return IDIFF_OK;
}
extern "C" int idiff_colsum(const float* x, int64_t ldx, float* out, int R, int N, int accumulate, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && R > 0 && N > 0 && ldx >= N, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + CS_COLS - 1) / CS_COLS), dim3(256), 0, ST, x, (long long)ldx, out, R, N, accumulate);
    IDIFF_CHECK_LAUNCH("colsum");
    return IDIFF_OK;
}
extern "C" int idiff_colsum_g(const float* x, int64_t ldx, float* out, int R, int N, int groups, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && R > 0 && N > 0 && ldx >= N && groups > 0 && R % groups == 0, "colsum_g: R must be a multiple of groups");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + CS_COLS - 1) / CS_COLS, groups), dim3(256), 0, ST, x, (long long)ldx, out, R / groups, N, 0);
    IDIFF_CHECK_LAUNCH("colsum_g");
    return IDIFF_OK;
}
extern "C" int idiff_layernorm_rows_g_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean_rstd,
                                          float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int C, int groups, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && x && gamma && mean_rstd && dx && dgamma && dbeta && R > 0 && C > 0 && groups > 0 && R % groups == 0, "layernorm_rows_g_bwd: bad args");
    hipLaunchKernelGGL(ln_rows_bwd_dx_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, dy, (long long)lddy, x, (long long)ldx, gamma, mean_rstd, dx,
                       (long long)lddx, R, C, R / groups);
    IDIFF_CHECK_LAUNCH("layernorm_rows_g_bwd_dx");
    hipLaunchKernelGGL(ln_rows_bwd_param_kernel, dim3((C + CS_COLS - 1) / CS_COLS, groups), dim3(256), 0, ST, dy, (long long)lddy, x, (long long)ldx, mean_rstd,
                       dgamma, dbeta, R / groups, C, 0);
    IDIFF_CHECK_LAUNCH("layernorm_rows_g_bwd_param");
    return IDIFF_OK;
}
extern "C" int idiff_scale_cols(const float* x, const float* g, float* out, int R, int N, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && g && out && R > 0 && N > 0, "scale_cols: bad args");
    hipLaunchKernelGGL(scale_cols_kernel, dim3(bgrid((long long)R * N)), dim3(256), 0, ST, x, g, out, R, N);
    IDIFF_CHECK_LAUNCH("scale_cols");
    return IDIFF_OK;
}
extern "C" int idiff_colsum_prod(const float* x, const float* y, float* out, int R, int N, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && y && out && R > 0 && N > 0, "colsum_prod: bad args");
    hipLaunchKernelGGL(colsum_prod_kernel, dim3((N + CS_COLS - 1) / CS_COLS), dim3(256), 0, ST, x, y, out, R, N);
    IDIFF_CHECK_LAUNCH("colsum_prod");
    return IDIFF_OK;
}
extern "C" int idiff_layernorm_rows_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean_rstd,
                                        float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int C, int accumulate,
                                        idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && x && gamma && mean_rstd && dx && R > 0 && C > 0, "layernorm_rows_bwd: bad args");
    hipLaunchKernelGGL(ln_rows_bwd_dx_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, dy, (long long)lddy, x, (long long)ldx, gamma, mean_rstd, dx,
                       (long long)lddx, R, C);
    IDIFF_CHECK_LAUNCH("layernorm_rows_bwd_dx");
    if (dgamma && dbeta) {
        hipLaunchKernelGGL(ln_rows_bwd_param_kernel, dim3((C + CS_COLS - 1) / CS_COLS), dim3(256), 0, ST, dy, (long long)lddy, x, (long long)ldx, mean_rstd,
                           dgamma, dbeta, R, C, accumulate);
        IDIFF_CHECK_LAUNCH("layernorm_rows_bwd_param");
    }
    return IDIFF_OK;
}
static inline int cln_groups(int HW, int* tpw) {  // workgroups per sample of the fused form and tiles per workgroup
    const int ntiles = (HW + 63) / 64;
    *tpw = ntiles >= 64 ? CLN_TPW : 1;
    return (ntiles + *tpw - 1) / *tpw;
}
extern "C" int64_t idiff_chan_layernorm_bwd_ws_floats(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return -1;
    int tpw;
    const int G = cln_groups(HW, &tpw);
    return 2ll * B * C * (G > 1 ? G : 1);
}
extern "C" int idiff_chan_layernorm_bwd(const float* dy, int64_t dy_bstride, const float* x, int64_t x_bstride, const float* gamma,
                                        const float* mean_rstd, float* dx, int64_t dx_bstride, float* dgamma, float* dbeta, float* ws, int B, int C,
                                        int HW, int accumulate, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && x && gamma && mean_rstd && dx && ws && B > 0 && C > 0 && HW > 0, "chan_layernorm_bwd: bad args");
    if (dgamma && dbeta && C <= 256) {  // dx and the parameter partials in one pass
        int tpw;
        const int G = cln_groups(HW, &tpw);
        const int jmax = C <= 128 ? 32 : 64;
        const size_t lds = (size_t)(512 + 4 * jmax + 4 * jmax * 65) * sizeof(float);
        static idiff_dyn_lds_cache lc[2];
        auto kern = C <= 128 ? chan_ln_bwd_fused_kernel<32> : chan_ln_bwd_fused_kernel<64>;
        hipError_t e = idiff_ensure_dyn_lds(lc[C <= 128 ? 0 : 1], reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "chan_layernorm_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, dim3(G, B), dim3(256), lds, ST, dy, (long long)dy_bstride, x, (long long)x_bstride, gamma, mean_rstd, dx,
                           (long long)dx_bstride, ws, C, HW, tpw);
        IDIFF_CHECK_LAUNCH("chan_layernorm_bwd_fused");
        hipLaunchKernelGGL(pair_rows_sum_kernel, dim3((C + CS_COLS - 1) / CS_COLS), dim3(256), 0, ST, ws, dgamma, dbeta, B * G, C, accumulate);
        IDIFF_CHECK_LAUNCH("chan_layernorm_bwd_param_sum");
        return IDIFF_OK;
    }
    if (C <= 128)
        hipLaunchKernelGGL(chan_ln_bwd_dx_reg_kernel<32>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, dy, (long long)dy_bstride, x, (long long)x_bstride,
                           gamma, mean_rstd, dx, (long long)dx_bstride, C, HW);
    else if (C <= 256)
        hipLaunchKernelGGL(chan_ln_bwd_dx_reg_kernel<64>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, dy, (long long)dy_bstride, x, (long long)x_bstride,
                           gamma, mean_rstd, dx, (long long)dx_bstride, C, HW);
    else
        hipLaunchKernelGGL(chan_ln_bwd_dx_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, ST, dy, (long long)dy_bstride, x, (long long)x_bstride,
                           gamma, mean_rstd, dx, (long long)dx_bstride, C, HW);
    IDIFF_CHECK_LAUNCH("chan_layernorm_bwd_dx");
    if (dgamma && dbeta) {
        hipLaunchKernelGGL(chan_ln_bwd_param_kernel, dim3(B * C), dim3(256), 0, ST, dy, (long long)dy_bstride, x, (long long)x_bstride, mean_rstd,
                           ws, C, HW);
        IDIFF_CHECK_LAUNCH("chan_layernorm_bwd_param");
        hipLaunchKernelGGL(pair_batch_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, ws, dgamma, dbeta, B, C, accumulate);
        IDIFF_CHECK_LAUNCH("chan_layernorm_bwd_param_sum");
    }
    return IDIFF_OK;
}
extern "C" int idiff_chan_normalize_fwd(const float* x, int64_t x_bstride, float* y, float* nrm, int B, int C, int HW, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && y && nrm && B > 0 && C > 0 && HW > 0, "chan_normalize_fwd: bad args");
    if (C <= 128)
        hipLaunchKernelGGL(chan_normalize_fwd_reg_kernel<32>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, x, (long long)x_bstride, y, nrm, C, HW);
    else if (C <= 256)
        hipLaunchKernelGGL(chan_normalize_fwd_reg_kernel<64>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, x, (long long)x_bstride, y, nrm, C, HW);
    else
        hipLaunchKernelGGL(chan_normalize_fwd_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, ST, x, (long long)x_bstride, y, nrm, C, HW);
    IDIFF_CHECK_LAUNCH("chan_normalize_fwd");
    return IDIFF_OK;
}
extern "C" int idiff_chan_normalize_bwd(const float* dy, const float* y, const float* nrm, float* dx, int64_t dx_bstride, int B, int C, int HW,
                                        idiff_stream_t stream) {
    IDIFF_CHECK_ARG(dy && y && nrm && dx && B > 0 && C > 0 && HW > 0, "chan_normalize_bwd: bad args");
    if (C <= 128)
        hipLaunchKernelGGL(chan_normalize_bwd_reg_kernel<32>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, dy, y, nrm, dx, (long long)dx_bstride, C, HW);
    else if (C <= 256)
        hipLaunchKernelGGL(chan_normalize_bwd_reg_kernel<64>, dim3((HW + 63) / 64, B), dim3(256), 0, ST, dy, y, nrm, dx, (long long)dx_bstride, C, HW);
    else
        hipLaunchKernelGGL(chan_normalize_bwd_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, ST, dy, y, nrm, dx, (long long)dx_bstride, C, HW);
    IDIFF_CHECK_LAUNCH("chan_normalize_bwd");
    return IDIFF_OK;
}
extern "C" int idiff_scatter_channel(const float* x, const int32_t* idx, float* out, int B, int C, int HW, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && idx && out && B > 0 && C > 0 && HW > 0, "scatter_channel: bad args");
    hipLaunchKernelGGL(scatter_channel_kernel, dim3(bgrid((long long)B * C * HW)), dim3(256), 0, ST, x, idx, out, B, C, HW);
    IDIFF_CHECK_LAUNCH("scatter_channel");
    return IDIFF_OK;
}
extern "C" int idiff_resize_bilinear(const float* x, float* out, int64_t planes, int H, int W, int oh, int ow, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && planes > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "resize_bilinear: bad args");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(bgrid(planes * oh * ow)), dim3(256), 0, ST, x, out, (long long)planes, H, W, oh, ow);
    IDIFF_CHECK_LAUNCH("resize_bilinear");
    return IDIFF_OK;
}
#define IDIFF_MSE_PARTS 256
extern "C" int idiff_mse_loss(const float* a, const float* b, float* loss, float* grad, float* ws, int64_t n, float grad_scale,
                              idiff_stream_t stream) {
    IDIFF_CHECK_ARG(a && b && loss && ws && n > 0, "mse_loss: bad args");
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(IDIFF_MSE_PARTS), dim3(256), 0, ST, a, b, ws, grad, (long long)n, 2.0f * grad_scale / (float)n);
    IDIFF_CHECK_LAUNCH("mse_loss_partial");
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, ST, ws, IDIFF_MSE_PARTS, 1.0f / (float)n, loss);
    IDIFF_CHECK_LAUNCH("mse_loss_sum");
    return IDIFF_OK;
}
extern "C" int idiff_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale, int step, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad args");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(bgrid(n, 256, 8192)), dim3(256), 0, ST, p, g, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, bc1, bc2_sqrt);
    IDIFF_CHECK_LAUNCH("adam_step");
    return IDIFF_OK;
}

// ---- fan-in of gradients: out[b, :] = sum_i src_i[b, :] for 2..4 sources, each with its own batch stride (a gradient may be a channel
// slice of a bigger tensor: one source of a virtual concat).  ONE pass (n reads + 1 write) where autograd's own accumulation makes
// n - 1 passes of two reads + one write each, plus a copy for every non-contiguous operand; the order of the sum is the argument order.
namespace {
struct SumSrc {
    const float* p[4];
    long long bs[4];
};
template <int NS>
__global__ __launch_bounds__(256) void sum_n_kernel(const SumSrc s, float* __restrict__ out, long long obs, long long per4) {
    const int b = blockIdx.y;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < per4; v += (long long)gridDim.x * blockDim.x) {
        floatx4 a = reinterpret_cast<const floatx4*>(s.p[0] + (long long)b * s.bs[0])[v];
#pragma unroll
        for (int i = 1; i < NS; ++i) {
            const floatx4 x = reinterpret_cast<const floatx4*>(s.p[i] + (long long)b * s.bs[i])[v];
            a.x += x.x, a.y += x.y, a.z += x.z, a.w += x.w;
        }
        reinterpret_cast<floatx4*>(out + (long long)b * obs)[v] = a;
    }
}
}  // namespace
extern "C" int idiff_sum_n(const float* const* srcs, const int64_t* src_bstrides, int nsrc, float* out, int64_t out_bstride, int B,
                           int64_t per_sample, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(srcs && src_bstrides && out && nsrc >= 2 && nsrc <= 4 && B > 0 && per_sample > 0, "sum_n: 2..4 sources");
    IDIFF_CHECK_ARG(per_sample % 4 == 0 && out_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sum_n: 16-byte rows required");
    SumSrc s;
    memset(&s, 0, sizeof(s));
    for (int i = 0; i < nsrc; ++i) {
        IDIFF_CHECK_ARG(srcs[i] && src_bstrides[i] % 4 == 0 && (reinterpret_cast<uintptr_t>(srcs[i]) & 15) == 0, "sum_n: source %d: 16-byte rows required", i);
        s.p[i] = srcs[i], s.bs[i] = src_bstrides[i];
    }
    const long long per4 = per_sample / 4;
    long long gx = (per4 + 255) / 256;
    if (gx > 2048) gx = 2048;
    dim3 grid((unsigned)gx, B);
    if (nsrc == 2) hipLaunchKernelGGL(sum_n_kernel<2>, grid, dim3(256), 0, ST, s, out, (long long)out_bstride, per4);
    else if (nsrc == 3) hipLaunchKernelGGL(sum_n_kernel<3>, grid, dim3(256), 0, ST, s, out, (long long)out_bstride, per4);
    else hipLaunchKernelGGL(sum_n_kernel<4>, grid, dim3(256), 0, ST, s, out, (long long)out_bstride, per4);
    IDIFF_CHECK_LAUNCH("sum_n");
    return IDIFF_OK;
}
