// SDE update kernels: fused reverse steps (IRSDE and driftSDE), Philox4x32-10 normals, per-sample mixes.
// HBM-bound streaming kernels (float4 per lane).  The IRSDE step reproduces the reference's fp32 operation
// order exactly (utils/sde_utils.py:45-46,178-188): every product / sum below is rounded once, no FMA
// contraction, IEEE division, so with injected noise the result is bit-identical to the CPU reference.
#include "common.h"

// hipcc defaults to -ffp-contract=fast-honor-pragmas: without this the separate mul/sub below fuse to FMAs
// and the result differs from the reference's op-by-op fp32 arithmetic in the last bit.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }  // (0,1)

// 4 standard normals for counter ctr (Box-Muller on word pairs)
__device__ __forceinline__ floatx4 philox_normal4(uint64_t ctr, uint64_t seed) {
    uint32_t w[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const float r0 = sqrtf(-2.0f * logf(u01(w[0])));
    const float r1 = sqrtf(-2.0f * logf(u01(w[2])));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(w[1]), &s0, &c0);
    sincosf(6.283185307179586f * u01(w[3]), &s1, &c1);
    floatx4 z = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
    return z;
}

__device__ __forceinline__ floatx4 ld4(const float* p, long long i, long long n) {
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (i + 3 < n) return *reinterpret_cast<const floatx4*>(p + i);
    for (int k = 0; k < 4; ++k)
        if (i + k < n) v[k] = p[i + k];
    return v;
}
__device__ __forceinline__ void st4(float* p, long long i, long long n, floatx4 v) {
    if (i + 3 < n) {
        *reinterpret_cast<floatx4*>(p + i) = v;
        return;
    }
    for (int k = 0; k < 4; ++k)
        if (i + k < n) p[i + k] = v[k];
}

template <int MODE>
__global__ __launch_bounds__(256) void irsde_step_kernel(const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ np_,
                                                         const float* __restrict__ z, float* __restrict__ xo, long long n, float theta,
                                                         float sigma, float sigma_bar, float dt, float sqrt_dt, uint64_t seed,
                                                         uint64_t offset) {
    const float s2 = __fmul_rn(sigma, sigma);
    const float coef = MODE == IDIFF_SDE_ODE ? __fmul_rn(0.5f, s2) : s2;
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        const long long i = v * 4;
        const floatx4 xv = ld4(x, i, n), mv = ld4(mu, i, n), nv4 = ld4(np_, i, n);
        floatx4 zv = {0.f, 0.f, 0.f, 0.f};
        if (MODE == IDIFF_SDE_STEP) zv = z ? ld4(z, i, n) : philox_normal4(offset + (uint64_t)v, seed);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // correctly rounded fp32 quotient: an fp64 divide rounded to fp32 is exact for p=24 (53 >= 2p+2),
            // independent of how the compiler lowers fp32 division by a loop-invariant scalar.
            const float score = (float)((double)(-nv4[k]) / (double)sigma_bar);
            const float t1 = __fsub_rn(mv[k], xv[k]);
            const float t2 = __fmul_rn(theta, t1);
            const float t3 = __fmul_rn(coef, score);
            const float t4 = __fsub_rn(t2, t3);
            const float t5 = __fmul_rn(t4, dt);
            float r = __fsub_rn(xv[k], t5);
            if (MODE == IDIFF_SDE_STEP) {
                const float nz = __fmul_rn(zv[k], sqrt_dt);
                r = __fsub_rn(r, __fmul_rn(sigma, nz));
            }
            o[k] = r;
        }
        st4(xo, i, n, o);
    }
}

__global__ __launch_bounds__(256) void drift_step_kernel(const float* __restrict__ x, const float* __restrict__ rh, const float* __restrict__ eh,
                                                         const float* __restrict__ z, const float* __restrict__ cond, float* __restrict__ xo,
                                                         float* __restrict__ xao, long long n, float a, float b, float c, uint64_t seed,
                                                         uint64_t offset) {
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        const long long i = v * 4;
        const floatx4 xv = ld4(x, i, n), rv = ld4(rh, i, n), ev = ld4(eh, i, n);
        floatx4 zv = {0.f, 0.f, 0.f, 0.f};
        if (c != 0.f) zv = z ? ld4(z, i, n) : philox_normal4(offset + (uint64_t)v, seed);
        floatx4 o, oa;
        floatx4 cv = {0.f, 0.f, 0.f, 0.f};
        if (xao) cv = ld4(cond, i, n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = __fsub_rn(xv[k], __fmul_rn(a, rv[k]));
            r = __fsub_rn(r, __fmul_rn(b, ev[k]));
            r = __fadd_rn(r, __fmul_rn(c, zv[k]));
            o[k] = r;
            oa[k] = __fsub_rn(r, cv[k]);
        }
        st4(xo, i, n, o);
        if (xao) st4(xao, i, n, oa);
    }
}

// Graph-replayable form of the drift step: every per-step scalar comes from device memory, so ONE captured HIP graph of a
// denoising step replays for every t.  state = {t, Philox call count, step index of this run}; coef = [3][Tp1] tables of
// (a_t, b_t, c_t); injected noise (parity runs) is indexed by the step index.  In place: x <- update, xa <- x - cond
// (element-wise, each thread reads before it writes its own element).  Same fp32 operation order as drift_step_kernel.
__global__ __launch_bounds__(256) void drift_step_dev_kernel(float* x, const float* __restrict__ rh, const float* __restrict__ eh,
                                                             const float* __restrict__ zbase, const float* __restrict__ cond, float* xa, long long n,
                                                             const float* __restrict__ coef, int Tp1, const int* __restrict__ state, uint64_t seed,
                                                             uint64_t nper, uint64_t offset_base) {
    const int t = state[0];
    const float a = coef[t], b = coef[Tp1 + t], c = coef[2 * Tp1 + t];
    const uint64_t offset = offset_base + (uint64_t)(unsigned)state[1] * nper;
    const float* z = zbase ? zbase + (long long)state[2] * n : nullptr;
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        const long long i = v * 4;
        const floatx4 xv = ld4(x, i, n), rv = ld4(rh, i, n), ev = ld4(eh, i, n);
        floatx4 zv = {0.f, 0.f, 0.f, 0.f};
        if (c != 0.f) zv = z ? ld4(z, i, n) : philox_normal4(offset + (uint64_t)v, seed);
        const floatx4 cv = ld4(cond, i, n);
        floatx4 o, oa;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = __fsub_rn(xv[k], __fmul_rn(a, rv[k]));
            r = __fsub_rn(r, __fmul_rn(b, ev[k]));
            r = __fadd_rn(r, __fmul_rn(c, zv[k]));
            o[k] = r;
            oa[k] = __fsub_rn(r, cv[k]);
        }
        st4(x, i, n, o);
        st4(xa, i, n, oa);
    }
}

// t <- t-1 (wrapping to T below t_stop+1, for benchmark loops), counters += 1, tdev[:] = t  -- the host never touches a
// per-step scalar between graph replays
__global__ void step_state_advance_kernel(int* state, float* tdev, int B, int T, int t_stop) {
    int t = state[0] - 1;
    if (t <= t_stop) t = T;
    for (int i = threadIdx.x; i < B; i += blockDim.x) tdev[i] = (float)t;
    __syncthreads();
    if (threadIdx.x == 0) {
        state[0] = t;
        state[1] += 1;
        state[2] += 1;
    }
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset) {
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x)
        st4(out, v * 4, n, philox_normal4(offset + (uint64_t)v, seed));
}

// Inverted dropout keyed by the Philox stream: element i is kept iff the 24 high bits of word i % 4 of counter offset + i / 4,
// as u in [0, 1), are >= p; kept elements are scaled by 1 / (1 - p).  The mask is a pure function of (seed, offset, i): the backward
// pass applies the same launch to the gradient instead of storing a mask.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float p, float scale,
                                                      uint64_t seed, uint64_t offset) {
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        uint32_t w[4];
        const uint64_t ctr = offset + (uint64_t)v;
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = v * 4 + k;
            if (i < n) {
                const float u = (float)(w[k] >> 8) * (1.0f / 16777216.0f);
                out[i] = u >= p ? __fmul_rn(x[i], scale) : 0.f;
            }
        }
    }
}

__global__ void philox_raw_kernel(uint32_t* __restrict__ out, long long nc, uint64_t seed, uint64_t offset) {
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nc; v += (long long)gridDim.x * blockDim.x) {
        uint32_t w[4];
        const uint64_t ctr = offset + (uint64_t)v;
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        for (int k = 0; k < 4; ++k) out[v * 4 + k] = w[k];
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long long n,
                                                    float alpha, float beta) {
    const long long nv = (n + 3) / 4;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
        const long long i = v * 4;
        const floatx4 xv = ld4(x, i, n), yv = ld4(y, i, n);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = __fadd_rn(__fmul_rn(alpha, xv[k]), __fmul_rn(beta, yv[k]));
        st4(out, i, n, o);
    }
}

__global__ __launch_bounds__(256) void mix3_kernel(const float* __restrict__ x0, const float* __restrict__ cond, const float* __restrict__ eps,
                                                   const float* __restrict__ c0, const float* __restrict__ c1, const float* __restrict__ c2,
                                                   float* __restrict__ out, long long per) {
    const int b = blockIdx.y;
    const float a0 = c0[b], a1 = c1[b], a2 = c2[b];
    const long long base = (long long)b * per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        float r = __fmul_rn(a0, x0[base + i]);
        r = __fadd_rn(r, __fmul_rn(a1, cond[base + i]));
        r = __fadd_rn(r, __fmul_rn(a2, eps[base + i]));
        out[base + i] = r;
    }
}


// ---- IRSDE pieces (idiff_irsde_map): the reference's own decomposition of its steps and closed forms, one launch each ----
struct IrsdeCoef {
    float k[6];
};

template <int OP>
__device__ __forceinline__ float irsde_piece(float a, float b, float z, float m, const float* k) {
    if (OP == IDIFF_IRSDE_SCORE_FROM_NOISE) return (float)((double)(-a) / (double)k[0]);  // correctly rounded fp32 quotient
    if (OP == IDIFF_IRSDE_MU_BAR) return __fadd_rn(m, __fmul_rn(__fsub_rn(a, m), k[0]));
    if (OP == IDIFF_IRSDE_DRIFT) return __fmul_rn(__fmul_rn(k[0], __fsub_rn(m, a)), k[1]);
    if (OP == IDIFF_IRSDE_DISPERSION) return __fmul_rn(k[0], __fmul_rn(z, k[1]));
    if (OP == IDIFF_IRSDE_REV_DRIFT || OP == IDIFF_IRSDE_STEP_MEAN || OP == IDIFF_IRSDE_STEP_SDE) {
        const float rd = __fmul_rn(__fsub_rn(__fmul_rn(k[0], __fsub_rn(m, a)), __fmul_rn(k[1], b)), k[2]);
        if (OP == IDIFF_IRSDE_REV_DRIFT) return rd;
        const float r = __fsub_rn(a, rd);
        if (OP == IDIFF_IRSDE_STEP_MEAN) return r;
        return __fsub_rn(r, __fmul_rn(k[3], __fmul_rn(z, k[4])));
    }
    if (OP == IDIFF_IRSDE_FORWARD_STEP)
        return __fadd_rn(__fadd_rn(a, __fmul_rn(__fmul_rn(k[0], __fsub_rn(m, a)), k[1])), __fmul_rn(k[3], __fmul_rn(z, k[4])));
    if (OP == IDIFF_IRSDE_OPT_STEP)
        return __fadd_rn(__fadd_rn(__fmul_rn(k[0], __fsub_rn(a, m)), __fmul_rn(k[1], __fsub_rn(b, m))), m);
    if (OP == IDIFF_IRSDE_REAL_NOISE || OP == IDIFF_IRSDE_REAL_SCORE) {
        const float d = __fsub_rn(a, __fadd_rn(m, __fmul_rn(__fsub_rn(b, m), k[0])));
        return (float)((double)(OP == IDIFF_IRSDE_REAL_SCORE ? -d : d) / (double)k[1]);
    }
    if (OP == IDIFF_IRSDE_INIT_FROM_NOISE) return __fadd_rn(__fmul_rn(__fsub_rn(__fsub_rn(a, m), __fmul_rn(k[0], b)), k[1]), m);
    if (OP == IDIFF_IRSDE_RANDOM_STATES) return __fadd_rn(__fmul_rn(z, k[1]), __fadd_rn(m, __fmul_rn(__fsub_rn(a, m), k[0])));
    return 0.f;
}

constexpr bool irsde_op_draws(int op) {
    return op == IDIFF_IRSDE_DISPERSION || op == IDIFF_IRSDE_STEP_SDE || op == IDIFF_IRSDE_FORWARD_STEP || op == IDIFF_IRSDE_RANDOM_STATES;
}

// grid = (chunks, B); per % 4 == 0 -> float4 lanes, Philox counter = offset + (flat element index)/4 as in idiff_randn
template <int OP>
__global__ __launch_bounds__(256) void irsde_map_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ z,
                                                        const float* __restrict__ mu, float mu_scalar, float* __restrict__ out, long long per,
                                                        const float* __restrict__ coef_dev, IrsdeCoef kc, uint64_t seed, uint64_t offset, int vec4) {
    const int s = blockIdx.y;
    float k[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) k[i] = coef_dev ? coef_dev[s * 6 + i] : kc.k[i];
    const long long base = (long long)s * per;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const floatx4 mconst = {mu_scalar, mu_scalar, mu_scalar, mu_scalar};
    if (vec4) {  // per % 4 == 0 and every operand 16-byte aligned (checked on the host)
        const long long nv = per / 4;
        for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nv; v += (long long)gridDim.x * blockDim.x) {
            const long long i = base + v * 4;
            const floatx4 av = a ? *reinterpret_cast<const floatx4*>(a + i) : zero;
            const floatx4 bv = b ? *reinterpret_cast<const floatx4*>(b + i) : zero;
            const floatx4 mv = mu ? *reinterpret_cast<const floatx4*>(mu + i) : mconst;
            floatx4 zv = zero;
            if (irsde_op_draws(OP)) zv = z ? *reinterpret_cast<const floatx4*>(z + i) : philox_normal4(offset + (uint64_t)(i >> 2), seed);
            floatx4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = irsde_piece<OP>(av[e], bv[e], zv[e], mv[e], k);
            *reinterpret_cast<floatx4*>(out + i) = o;
        }
    } else {
        for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < per; j += (long long)gridDim.x * blockDim.x) {
            const long long i = base + j;
            float zz = 0.f;
            if (irsde_op_draws(OP)) zz = z ? z[i] : philox_normal4(offset + (uint64_t)(i >> 2), seed)[i & 3];
            out[i] = irsde_piece<OP>(a ? a[i] : 0.f, b ? b[i] : 0.f, zz, mu ? mu[i] : mu_scalar, k);
        }
    }
}

template <int OP>
void launch_irsde_map(const float* a, const float* b, const float* z, const float* mu, float mu_scalar, float* out, int B, long long per,
                      const float* coef_dev, const IrsdeCoef& kc, uint64_t seed, uint64_t offset, hipStream_t st) {
    long long gx = ((per + 3) / 4 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)z | (uintptr_t)mu | (uintptr_t)out;
    const int vec4 = (per % 4 == 0) && (bits % 16 == 0);
    hipLaunchKernelGGL(irsde_map_kernel<OP>, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, st, a, b, z, mu, mu_scalar, out, per, coef_dev, kc, seed,
                       offset, vec4);
}

inline int stream_grid(long long nvec) {
    long long g = (nvec + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > 2048 ? 2048 : g);
}

}  // namespace

extern "C" int idiff_irsde_reverse_step(const float* x, const float* mu, const float* noise_pred, const float* z, float* x_out, int64_t n,
                                        float theta, float sigma, float sigma_bar, float dt, float sqrt_dt, int mode, uint64_t seed,
                                        uint64_t offset, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && mu && noise_pred && x_out && n > 0, "irsde_reverse_step: bad args");
    IDIFF_CHECK_ARG(mode >= 0 && mode <= 2, "irsde_reverse_step: bad mode %d", mode);
    IDIFF_CHECK_ARG(sigma_bar != 0.f, "irsde_reverse_step: sigma_bar == 0");
    const int grid = stream_grid((n + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (mode == IDIFF_SDE_STEP)
        hipLaunchKernelGGL(irsde_step_kernel<IDIFF_SDE_STEP>, dim3(grid), dim3(256), 0, st, x, mu, noise_pred, z, x_out, (long long)n, theta,
                           sigma, sigma_bar, dt, sqrt_dt, seed, offset);
    else if (mode == IDIFF_SDE_MEAN)
        hipLaunchKernelGGL(irsde_step_kernel<IDIFF_SDE_MEAN>, dim3(grid), dim3(256), 0, st, x, mu, noise_pred, z, x_out, (long long)n, theta,
                           sigma, sigma_bar, dt, sqrt_dt, seed, offset);
    else
        hipLaunchKernelGGL(irsde_step_kernel<IDIFF_SDE_ODE>, dim3(grid), dim3(256), 0, st, x, mu, noise_pred, z, x_out, (long long)n, theta,
                           sigma, sigma_bar, dt, sqrt_dt, seed, offset);
    IDIFF_CHECK_LAUNCH("irsde_reverse_step");
    return IDIFF_OK;
}


extern "C" int idiff_irsde_map(int op, const float* a, const float* b, const float* z, const float* mu, float mu_scalar, float* out, int B,
                               int64_t per_sample, const float* coef_dev, const float* k, uint64_t seed, uint64_t offset, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(op >= 0 && op < IDIFF_IRSDE_NUM_OPS, "irsde_map: bad op %d", op);
    IDIFF_CHECK_ARG(out && B > 0 && B <= 65535 && per_sample > 0, "irsde_map: bad args");
    IDIFF_CHECK_ARG(coef_dev || k, "irsde_map: needs coefficients (coef_dev or k)");
    IDIFF_CHECK_ARG(a || op == IDIFF_IRSDE_DISPERSION, "irsde_map: op %d reads operand a", op);
    const bool needs_b = op == IDIFF_IRSDE_REV_DRIFT || op == IDIFF_IRSDE_STEP_MEAN || op == IDIFF_IRSDE_STEP_SDE || op == IDIFF_IRSDE_OPT_STEP ||
                         op == IDIFF_IRSDE_REAL_NOISE || op == IDIFF_IRSDE_REAL_SCORE || op == IDIFF_IRSDE_INIT_FROM_NOISE;
    IDIFF_CHECK_ARG(b || !needs_b, "irsde_map: op %d reads operand b", op);
    IrsdeCoef kc;
    for (int i = 0; i < 6; ++i) kc.k[i] = k ? k[i] : 0.f;
    if (!coef_dev && (op == IDIFF_IRSDE_SCORE_FROM_NOISE || op == IDIFF_IRSDE_REAL_NOISE || op == IDIFF_IRSDE_REAL_SCORE))
        IDIFF_CHECK_ARG(kc.k[op == IDIFF_IRSDE_SCORE_FROM_NOISE ? 0 : 1] != 0.f, "irsde_map: division by sigma_bar == 0");
    hipStream_t st = (hipStream_t)stream;
    const long long per = (long long)per_sample;
#define IDIFF_IRSDE_CASE(OP)                                                                           \
    case OP:                                                                                           \
        launch_irsde_map<OP>(a, b, z, mu, mu_scalar, out, B, per, coef_dev, kc, seed, offset, st); \
        break;
    switch (op) {
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_SCORE_FROM_NOISE)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_MU_BAR)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_DRIFT)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_REV_DRIFT)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_DISPERSION)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_STEP_MEAN)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_STEP_SDE)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_FORWARD_STEP)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_OPT_STEP)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_REAL_NOISE)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_REAL_SCORE)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_INIT_FROM_NOISE)
        IDIFF_IRSDE_CASE(IDIFF_IRSDE_RANDOM_STATES)
    }
#undef IDIFF_IRSDE_CASE
    IDIFF_CHECK_LAUNCH("irsde_map");
    return IDIFF_OK;
}

extern "C" int idiff_drift_reverse_step(const float* x, const float* r_hat, const float* e_hat, const float* z, const float* cond, float* x_out,
                                        float* xa_out, int64_t n, float a, float b, float c, uint64_t seed, uint64_t offset,
                                        idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && r_hat && e_hat && x_out && n > 0, "drift_reverse_step: bad args");
    IDIFF_CHECK_ARG(!xa_out || cond, "drift_reverse_step: xa_out needs cond");
    hipLaunchKernelGGL(drift_step_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, r_hat, e_hat, z, cond, x_out,
                       xa_out, (long long)n, a, b, c, seed, offset);
    IDIFF_CHECK_LAUNCH("drift_reverse_step");
    return IDIFF_OK;
}

extern "C" int idiff_drift_reverse_step_dev(float* x, const float* r_hat, const float* e_hat, const float* z_base, const float* cond, float* xa,
                                            int64_t n, const float* coef, int Tp1, const int32_t* state, uint64_t seed, uint64_t nper,
                                            uint64_t offset_base, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && r_hat && e_hat && cond && xa && coef && state && n > 0 && Tp1 > 1, "drift_reverse_step_dev: bad args");
    hipLaunchKernelGGL(drift_step_dev_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, r_hat, e_hat, z_base, cond, xa,
                       (long long)n, coef, Tp1, state, seed, nper, offset_base);
    IDIFF_CHECK_LAUNCH("drift_reverse_step_dev");
    return IDIFF_OK;
}

extern "C" int idiff_step_state_advance(int32_t* state, float* tdev, int B, int T, int t_stop, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(state && tdev && B > 0 && T > 0 && t_stop >= 0 && t_stop < T, "step_state_advance: bad args");
    hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, tdev, B, T, t_stop);
    IDIFF_CHECK_LAUNCH("step_state_advance");
    return IDIFF_OK;
}

extern "C" int idiff_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(out && n > 0, "randn: bad args");
    hipLaunchKernelGGL(randn_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, (long long)n, seed, offset);
    IDIFF_CHECK_LAUNCH("randn");
    return IDIFF_OK;
}

extern "C" int idiff_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, uint64_t offset, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args (p must be in [0, 1))");
    hipLaunchKernelGGL(dropout_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n, p, 1.0f / (1.0f - p),
                       seed, offset);
    IDIFF_CHECK_LAUNCH("dropout");
    return IDIFF_OK;
}

extern "C" int idiff_philox_raw(uint32_t* out, int64_t ncounters, uint64_t seed, uint64_t offset, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(out && ncounters > 0, "philox_raw: bad args");
    hipLaunchKernelGGL(philox_raw_kernel, dim3(stream_grid(ncounters)), dim3(256), 0, (hipStream_t)stream, out, (long long)ncounters, seed,
                       offset);
    IDIFF_CHECK_LAUNCH("philox_raw");
    return IDIFF_OK;
}

extern "C" int idiff_axpby(const float* x, const float* y, float* out, int64_t n, float alpha, float beta, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && y && out && n > 0, "axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, out, (long long)n, alpha, beta);
    IDIFF_CHECK_LAUNCH("axpby");
    return IDIFF_OK;
}

extern "C" int idiff_mix3_per_sample(const float* x0, const float* cond, const float* eps, const float* c0, const float* c1, const float* c2,
                                     float* out, int B, int64_t per_sample, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x0 && cond && eps && c0 && c1 && c2 && out && B > 0 && per_sample > 0, "mix3_per_sample: bad args");
    dim3 grid(stream_grid(per_sample), B);
    hipLaunchKernelGGL(mix3_kernel, grid, dim3(256), 0, (hipStream_t)stream, x0, cond, eps, c0, c1, c2, out, (long long)per_sample);
    IDIFF_CHECK_LAUNCH("mix3_per_sample");
    return IDIFF_OK;
}
