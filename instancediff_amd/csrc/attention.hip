// Attention kernels (gfx950, fp32).  All compute softmax(q k^T * scale) v per head, i.e. the formula of
// models/_modified_BiomedCLIP.py:464-478, on channel-major ([B, C, N]) feature maps so that the pixel/key
// index is the unit-stride one and maps onto the MFMA N (lane) dimension without any transposes.
//
//  * attn_self   : flash-style self-attention over N = H*W tokens on v_mfma_f32_32x32x2_f32.
//                  S^T = K Q^T (keys on the accumulator rows, queries on the lanes), online softmax over the
//                  16 registers x 2 half-waves, and O^T += V^T P with P taken straight from the S accumulator
//                  (accumulator-as-B-operand, k order permuted to the accumulator's row order).
//  * attn_ctx    : every pixel attends to M <= 32 context tokens (image embedding); VALU, HBM-bound.
//  * attn_tokens : tiny token-major attention for the ScoreMapModule decoder self-attention.
//  * smm_xattn   : ScoreMapModule cross-attention (few text queries vs. N pixel keys) with the K/V
//                  projections folded to the query side; 4 waves split the channel (K) dimension of S and the
//                  channel (M) dimension of P.V, partial S tiles are exchanged through LDS; flash-decoding
//                  style split over keys + combine kernel.
#include <math.h>
#include <type_traits>
#include <stdlib.h>

#include "common.h"

namespace {

#define KAPPA(r) (((r) & 3) + 8 * ((r) >> 2))

// ---------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_self_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse, int C,
                                                        int N, int heads, float scale) {
    constexpr int MB = DH / 32;
    constexpr int KT = DH * 32;       // K tile floats  [DH][32]
    constexpr int VT = DH * 33;       // V tile floats  [DH][33]
    constexpr int BUF = KT + VT + 3;  // keep 16-B multiples irrelevant: scalar LDS access only
    constexpr int NV4 = DH * 32 / 4 / 256;  // float4 per thread per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / heads, h = blockIdx.y % heads;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const float* qb = qkv + (long long)b * 3 * C * N + (long long)(h * DH) * N;
    const float* kb = qb + (long long)C * N;
    const float* vb = qb + (long long)2 * C * N;

    float qreg[DH / 2];
    const int qi = q0 + l31;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) qreg[s] = qi < N ? qb[(long long)(2 * s + half) * N + qi] : 0.f;

    floatx16 O[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const int nkb = (N + 31) / 32;
    floatx4 rk[NV4], rv[NV4];
    auto load_tile = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int f = tid + i * 256;
            const int d = f >> 3, j4 = (f & 7) * 4;
            floatx4 kz = {0.f, 0.f, 0.f, 0.f}, vz = {0.f, 0.f, 0.f, 0.f};
            if (key0 + j4 + 3 < N) {
                kz = *reinterpret_cast<const floatx4*>(kb + (long long)d * N + key0 + j4);
                vz = *reinterpret_cast<const floatx4*>(vb + (long long)d * N + key0 + j4);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) {
                        kz[e] = kb[(long long)d * N + key0 + j4 + e];
                        vz[e] = vb[(long long)d * N + key0 + j4 + e];
                    }
            }
            rk[i] = kz;
            rv[i] = vz;
        }
    };
    auto write_tile = [&](int buf) {
        float* kt = smem + buf * BUF;
        float* vt = kt + KT;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int f = tid + i * 256;
            const int d = f >> 3, j4 = (f & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kt[d * 32 + j4 + e] = rk[i][e];
                vt[d * 33 + j4 + e] = rv[i][e];
            }
        }
    };

    load_tile(0);
    write_tile(0);
    __syncthreads();
    for (int kbi = 0; kbi < nkb; ++kbi) {
        const int buf = kbi & 1;
        if (kbi + 1 < nkb) load_tile(kbi + 1);
        const float* kt = smem + buf * BUF;
        const float* vt = kt + KT;
        floatx16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[(2 * s + half) * 32 + l31], qreg[s], S, 0, 0, 0);
        const int key0 = kbi * 32 + 4 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = (key0 + KAPPA(r) < N) ? S[r] * scale : -INFINITY;
            S[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(S[r] - mnew);
            S[r] = p;
            ps += p;
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
        // (the running maximum of a row rarely moves after the first key blocks: alpha == 1 in every lane -> the products are skipped,
        // the result is the same bits)
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[(m * 32 + l31) * 33 + KAPPA(r) + 4 * half], S[r], O[m], 0, 0, 0);
        if (kbi + 1 < nkb) write_tile(buf ^ 1);
        __syncthreads();
    }
    const float l = lrun + __shfl_xor(lrun, 32, 64);
    const float inv = 1.0f / l;
    if (qi < N) {
        float* ob = out + (long long)b * C * N + (long long)(h * DH) * N + qi;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(long long)(m * 32 + KAPPA(r) + 4 * half) * N] = O[m][r] * inv;
        if (lse && half == 0) lse[((long long)b * heads + h) * N + qi] = mrun + logf(l);
    }
}


// ---------------------------------------------------------------------------------------------------
// attn_self with the two contractions on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 softmax and fp32
// accumulation: a REDUCED-PRECISION VARIANT (BASELINE config c5 names "fp16 MFMA attention"; the reference's only
// half-precision code is its optional flash-attention branch, models/_modified_BiomedCLIP.py:396-400,509-513).  Off by
// default -- the headline path is fp32 end to end; bench.py reports this variant as its own line with its PSNR delta.
// Same tiling as attn_self_kernel<64>: a wave owns 32 queries, 32-key tiles pass through LDS (K as [key][d] rows, V as
// [d][key] rows, both bf16), S^T = K Q^T takes 4 MFMAs per tile instead of 32, O^T += V^T P two per 32 channels instead of 16:
// P is taken from the S accumulator (register r of half-wave g holds key KAPPA(r) + 4g), so the k slots of the second product
// are enumerated in that key order on the V side as well.
// ---------------------------------------------------------------------------------------------------
// HT = __bf16: the variant above.  HT = _Float16: the reference's own half-precision form (v_mfma_f32_32x32x16_f16, q / k / v clamped to
// +-255 before the cast, as models/_modified_BiomedCLIP.py:509-513 does for flash-attn) -- BASELINE config c5's "fp16 MFMA attention".
template <typename HT>
struct HalfVec {
    typedef HT x8 __attribute__((ext_vector_type(8)));
    typedef HT x4 __attribute__((ext_vector_type(4)));
};
template <typename HT>
__device__ __forceinline__ HT to_half(float v) {
    if constexpr (std::is_same<HT, _Float16>::value) v = fminf(fmaxf(v, -255.f), 255.f);
    return (HT)v;
}
template <typename HT>
__device__ __forceinline__ floatx16 mfma_half(typename HalfVec<HT>::x8 a, typename HalfVec<HT>::x8 b, floatx16 c) {
    if constexpr (std::is_same<HT, _Float16>::value) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename HT>
__global__ __launch_bounds__(256) void attn_self_half_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, int N, int heads,
                                                             float scale) {
    constexpr int DH = 64, MB = 2;
    constexpr int KROW = DH + 8;   // bf16 per K row (key-major), padded: rows land 4 banks apart
    constexpr int VROW = 32 + 4;   // bf16 per V row (channel-major), padded
    constexpr int KT = 32 * KROW, VT = DH * VROW, BUF = KT + VT;  // bf16 elements per buffer
    typedef typename HalfVec<HT>::x8 bf16x8;
    typedef typename HalfVec<HT>::x4 bf16x4;
    extern __shared__ __attribute__((aligned(16))) unsigned short bsm_raw[];
    HT* const bsm = reinterpret_cast<HT*>(bsm_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / heads, h = blockIdx.y % heads;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const float* qb = qkv + (long long)b * 3 * C * N + (long long)(h * DH) * N;
    const float* kb = qb + (long long)C * N;
    const float* vb = qb + (long long)2 * C * N;
    const int qi = q0 + l31;
    bf16x8 qreg[4];  // B operand of S^T = K Q^T: query l31, channels 16 s + 8 half + e
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qreg[s][e] = to_half<HT>(qi < N ? qb[(long long)(16 * s + 8 * half + e) * N + qi] : 0.f);
    floatx16 O[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int nkb = (N + 31) / 32;
    floatx4 rk[2], rv[2];  // 64 channels x 32 keys = 512 float4 per tile, 2 per thread
    auto load_tile = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * 256;
            const int d = f >> 3, j4 = (f & 7) * 4;
            floatx4 kz = {0.f, 0.f, 0.f, 0.f}, vz = {0.f, 0.f, 0.f, 0.f};
            if (key0 + j4 + 3 < N) {
                kz = *reinterpret_cast<const floatx4*>(kb + (long long)d * N + key0 + j4);
                vz = *reinterpret_cast<const floatx4*>(vb + (long long)d * N + key0 + j4);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) kz[e] = kb[(long long)d * N + key0 + j4 + e], vz[e] = vb[(long long)d * N + key0 + j4 + e];
            }
            rk[i] = kz, rv[i] = vz;
        }
    };
    auto write_tile = [&](int buf) {
        HT* kt = bsm + buf * BUF;
        HT* vt = kt + KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * 256;
            const int d = f >> 3, j4 = (f & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) kt[(j4 + e) * KROW + d] = to_half<HT>(rk[i][e]);  // transposed: key-major rows
            *reinterpret_cast<bf16x4*>(vt + d * VROW + j4) = bf16x4{to_half<HT>(rv[i][0]), to_half<HT>(rv[i][1]), to_half<HT>(rv[i][2]), to_half<HT>(rv[i][3])};
        }
    };
    load_tile(0);
    write_tile(0);
    __syncthreads();
    for (int kbi = 0; kbi < nkb; ++kbi) {
        const int buf = kbi & 1;
        if (kbi + 1 < nkb) load_tile(kbi + 1);
        const HT* kt = bsm + buf * BUF;
        const HT* vt = kt + KT;
        floatx16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // A: key l31, channels 16 s + 8 half + e
            const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kt + l31 * KROW + 16 * s + 8 * half);
            S = mfma_half<HT>(ka, qreg[s], S);
        }
        const int key0 = kbi * 32 + 4 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = (key0 + KAPPA(r) < N) ? S[r] * scale : -INFINITY;
            S[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float ps = 0.f;
        bf16x8 pb[2];  // B operand of O^T += V^T P: keys KAPPA(e) + 4 half (+16 for the second product)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(S[r] - mnew);
            ps += p;
            pb[r >> 3][r & 7] = (HT)p;
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
        // (the running maximum of a row rarely moves after the first key blocks: alpha == 1 in every lane -> the products are skipped,
        // the result is the same bits)
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m) {  // A: channel 32 m + l31, the same key order as pb
                const HT* vr = vt + (m * 32 + l31) * VROW + 16 * t + 4 * half;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr), hi = *reinterpret_cast<const bf16x4*>(vr + 8);
                const bf16x8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                O[m] = mfma_half<HT>(va, pb[t], O[m]);
            }
        if (kbi + 1 < nkb) write_tile(buf ^ 1);
        __syncthreads();
    }
    const float l = lrun + __shfl_xor(lrun, 32, 64);
    const float inv = 1.0f / l;
    if (qi < N) {
        float* ob = out + (long long)b * C * N + (long long)(h * DH) * N + qi;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(long long)(m * 32 + KAPPA(r) + 4 * half) * N] = O[m][r] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_ctx_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                       float* __restrict__ out, int C, int N, int M, float scale) {
    extern __shared__ float kv[];  // k [M][DH], v [M][DH]
    const int b = blockIdx.z, h = blockIdx.y;
    for (int i = threadIdx.x; i < M * DH; i += blockDim.x) {
        const int m = i / DH, d = i % DH;
        kv[i] = k[((long long)b * M + m) * C + h * DH + d];
        kv[M * DH + i] = v[((long long)b * M + m) * C + h * DH + d];
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const float* qp = q + ((long long)b * C + h * DH) * N + p;
    float qr[DH], o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        qr[d] = qp[(long long)d * N];
        o[d] = 0.f;
    }
    float mrun = -INFINITY, l = 0.f;
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s += qr[d] * kv[m * DH + d];
        s *= scale;
        const float mnew = fmaxf(mrun, s);
        const float alpha = __expf(mrun - mnew), pw = __expf(s - mnew);
        l = l * alpha + pw;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = o[d] * alpha + pw * kv[M * DH + m * DH + d];
        mrun = mnew;
    }
    const float inv = 1.0f / l;
    float* op = out + ((long long)b * C + h * DH) * N + p;
#pragma unroll
    for (int d = 0; d < DH; ++d) op[(long long)d * N] = o[d] * inv;
}

// ---------------------------------------------------------------------------------------------------
// one wave per (b, head, query); lanes = keys
__global__ __launch_bounds__(64) void attn_tokens_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         float* __restrict__ out, int Nq, int M, int C, int heads, float scale, long long ldq,
                                                         long long ldkv) {
    const int dh = C / heads;
    const int n = blockIdx.x % Nq, h = (blockIdx.x / Nq) % heads, b = blockIdx.x / (Nq * heads);
    const int lane = threadIdx.x;
    const float* qp = q + ((long long)b * Nq + n) * ldq + h * dh;
    float s = -INFINITY;
    if (lane < M) {
        const float* kp = k + ((long long)b * M + lane) * ldkv + h * dh;
        float acc = 0.f;
        for (int d = 0; d < dh; ++d) acc += qp[d] * kp[d];
        s = acc * scale;
    }
    const float mx = wave_max(s);
    const float p = lane < M ? __expf(s - mx) : 0.f;
    const float l = wave_sum(p);
    const float pn = p / l;
    float* op = out + ((long long)b * Nq + n) * C + h * dh;
    for (int d = 0; d < dh; ++d) {
        const float t = wave_sum(lane < M ? pn * v[((long long)b * M + lane) * ldkv + h * dh + d] : 0.f);
        if (lane == 0) op[d] = t;
    }
}

// Backward of attn_tokens for the few-token self-attention of the ScoreMapModule decoder (Nq, M <= 8: K = 5 class tokens), training path.
// One wave per (b, head); lane = channel d of the head (dh <= 64).  With P = softmax_j(scale q_i.k_j) recomputed in registers:
//   dP_ij = do_i . v_j ;  D_i = sum_j P_ij dP_ij ;  dS_ij = scale P_ij (dP_ij - D_i)
//   dq_i = sum_j dS_ij k_j ;  dk_j = sum_i dS_ij q_i ;  dv_j = sum_i P_ij do_i
// replaces two batched GEMMs + softmax forward and four + softmax backward (and the head permute copies around them) per layer.
constexpr int ATB = 8;
__global__ __launch_bounds__(64) void attn_tokens_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ d_o, float* __restrict__ dq, float* __restrict__ dk,
                                                             float* __restrict__ dv, int Nq, int M, int C, int heads, float scale, long long ldq,
                                                             long long ldkv, long long ldo, long long lddq, long long lddkv) {
    const int dh = C / heads;
    const int h = blockIdx.x % heads, b = blockIdx.x / heads;
    const int d = threadIdx.x;
    const bool on = d < dh;
    float qr[ATB], kr[ATB], vr[ATB], gr[ATB];
#pragma unroll
    for (int i = 0; i < ATB; ++i) {
        qr[i] = (on && i < Nq) ? q[((long long)b * Nq + i) * ldq + h * dh + d] : 0.f;
        gr[i] = (on && i < Nq) ? d_o[((long long)b * Nq + i) * ldo + h * dh + d] : 0.f;
        kr[i] = (on && i < M) ? k[((long long)b * M + i) * ldkv + h * dh + d] : 0.f;
        vr[i] = (on && i < M) ? v[((long long)b * M + i) * ldkv + h * dh + d] : 0.f;
    }
    float dqa[ATB], dka[ATB], dva[ATB];
#pragma unroll
    for (int i = 0; i < ATB; ++i) dqa[i] = 0.f, dka[i] = 0.f, dva[i] = 0.f;
#pragma unroll
    for (int i = 0; i < ATB; ++i) {
        if (i < Nq) {  // uniform
            float sc[ATB], dp[ATB];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < ATB; ++j) {
                sc[j] = j < M ? wave_sum(qr[i] * kr[j]) * scale : -INFINITY;
                dp[j] = j < M ? wave_sum(gr[i] * vr[j]) : 0.f;
                mx = fmaxf(mx, sc[j]);
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < ATB; ++j) {
                sc[j] = j < M ? __expf(sc[j] - mx) : 0.f;
                l += sc[j];
            }
            float D = 0.f;
#pragma unroll
            for (int j = 0; j < ATB; ++j) {
                sc[j] = sc[j] / l;
                D += sc[j] * dp[j];
            }
#pragma unroll
            for (int j = 0; j < ATB; ++j) {
                const float ds = scale * sc[j] * (dp[j] - D);
                dqa[i] += ds * kr[j];
                dka[j] += ds * qr[i];
                dva[j] += sc[j] * gr[i];
            }
        }
    }
    if (on) {
#pragma unroll
        for (int i = 0; i < ATB; ++i) {
            if (i < Nq) dq[((long long)b * Nq + i) * lddq + h * dh + d] = dqa[i];
            if (i < M) {
                dk[((long long)b * M + i) * lddkv + h * dh + d] = dka[i];
                dv[((long long)b * M + i) * lddkv + h * dh + d] = dva[i];
            }
        }
    }
}

// grouped form (idiff_attn_tokens_grouped_fwd): blockIdx.y picks one of up to IDIFF_LINEAR_MAX_GROUPS operand sets of the same shape
struct AttnTokGroups {
    const float* q[IDIFF_LINEAR_MAX_GROUPS];
    const float* k[IDIFF_LINEAR_MAX_GROUPS];
    const float* v[IDIFF_LINEAR_MAX_GROUPS];
    float* out[IDIFF_LINEAR_MAX_GROUPS];
};
__global__ __launch_bounds__(64) void attn_tokens_grouped_kernel(const AttnTokGroups g, int Nq, int M, int C, int heads, float scale, long long ldq,
                                                                 long long ldkv) {
    const float* __restrict__ q = g.q[blockIdx.y];
    const float* __restrict__ k = g.k[blockIdx.y];
    const float* __restrict__ v = g.v[blockIdx.y];
    float* __restrict__ out = g.out[blockIdx.y];
    const int dh = C / heads;
    const int n = blockIdx.x % Nq, h = (blockIdx.x / Nq) % heads, b = blockIdx.x / (Nq * heads);
    const int lane = threadIdx.x;
    const float* qp = q + ((long long)b * Nq + n) * ldq + h * dh;
    float s = -INFINITY;
    if (lane < M) {
        const float* kp = k + ((long long)b * M + lane) * ldkv + h * dh;
        float acc = 0.f;
        for (int d = 0; d < dh; ++d) acc += qp[d] * kp[d];
        s = acc * scale;
    }
    const float mx = wave_max(s);
    const float p = lane < M ? __expf(s - mx) : 0.f;
    const float l = wave_sum(p);
    const float pn = p / l;
    float* op = out + ((long long)b * Nq + n) * C + h * dh;
    for (int d = 0; d < dh; ++d) {
        const float t = wave_sum(lane < M ? pn * v[((long long)b * M + lane) * ldkv + h * dh + d] : 0.f);
        if (lane == 0) op[d] = t;
    }
}

// ---------------------------------------------------------------------------------------------------
// ScoreMapModule cross attention, Cm = 4*XCW in {72, 136, 256}, rows = Nq*heads <= 32.  XCW = channels per wave
// (even); the P.V product runs on whole 32-channel blocks, so a wave's LDS slice is padded to XCB*32 rows
// (the padding rows produce accumulator rows that are never stored).
template <int XCW>
__device__ __forceinline__ void smm_xattn_body(const float* __restrict__ qf, const float* __restrict__ mem, float* __restrict__ ws, int rows, int N,
                                               int nsplit, int kps, float scale, const int b, const int sp) {
    constexpr int XCM = 4 * XCW;
    constexpr int XCB = (XCW + 31) / 32;      // 32-channel blocks per wave in the P.V product
    constexpr int XTILE = XCB * 32 * 33;      // per-wave mem slice [XCB*32 c][33]
    constexpr int NF4 = (XCW * 8 + 63) / 64;  // float4 loads per lane per 32-key block (the last one partial unless XCW % 8 == 0)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem + (threadIdx.x >> 6) * XTILE;  // private per wave
    float* xch = smem + 4 * XTILE;                    // [4][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int c0 = wave * XCW;
    const float* memb = mem + (long long)b * XCM * N + (long long)c0 * N;

    float qreg[XCW / 2];
#pragma unroll
    for (int t = 0; t < XCW / 2; ++t) qreg[t] = l31 < rows ? qf[((long long)b * rows + l31) * XCM + c0 + 2 * t + half] : 0.f;

    floatx16 O[XCB];
#pragma unroll
    for (int m = 0; m < XCB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const int nkb = (N + 31) / 32;
    const int kb_begin = sp * kps;
    const int kb_end = min(nkb, kb_begin + kps);

    floatx4 rt[NF4];  // XCW c x 32 keys / 64 lanes
    auto load_tile = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;  // float4 index in [XCW][8]
            const int c = f >> 3, j4 = (f & 7) * 4;
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            if (XCW % 8 != 0 && f >= XCW * 8) {
            } else if (key0 + j4 + 3 < N)
                z = *reinterpret_cast<const floatx4*>(memb + (long long)c * N + key0 + j4);
            else
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) z[e] = memb[(long long)c * N + key0 + j4 + e];
            rt[i] = z;
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;
            const int c = f >> 3, j4 = (f & 7) * 4;
            if (XCW % 8 != 0 && f >= XCW * 8) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[c * 33 + j4 + e] = rt[i][e];
        }
    };

    if (kb_begin < kb_end) {
        load_tile(kb_begin);
        write_tile();
    }
    for (int kbi = kb_begin; kbi < kb_end; ++kbi) {
        if (kbi + 1 < kb_end) load_tile(kbi + 1);
        floatx16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int t = 0; t < XCW / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(2 * t + half) * 33 + l31], qreg[t], S, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = S[r];
        __syncthreads();
        const int key0 = kbi * 32 + 4 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sv = xch[(0 * 16 + r) * 64 + lane] + xch[(1 * 16 + r) * 64 + lane] + xch[(2 * 16 + r) * 64 + lane] + xch[(3 * 16 + r) * 64 + lane];
            sv = (key0 + KAPPA(r) < N) ? sv * scale : -INFINITY;
            S[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(S[r] - mnew);
            S[r] = p;
            ps += p;
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
        // (the running maximum of a row rarely moves after the first key blocks: alpha == 1 in every lane -> the products are skipped,
        // the result is the same bits)
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int m = 0; m < XCB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < XCB; ++m)
                O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(m * 32 + l31) * 33 + KAPPA(r) + 4 * half], S[r], O[m], 0, 0, 0);
        if (kbi + 1 < kb_end) write_tile();
        __syncthreads();  // xch reads of this block done before the next block's writes
    }
    // partial out: ws[b][sp][c][row], then m, l rows
    float* wp = ws + ((long long)b * nsplit + sp) * (XCM + 2) * 32;
    const float l = lrun + __shfl_xor(lrun, 32, 64);
#pragma unroll
    for (int m = 0; m < XCB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = m * 32 + KAPPA(r) + 4 * half;
            if (cl < XCW) wp[(c0 + cl) * 32 + l31] = O[m][r];
        }
    if (wave == 0 && half == 0) {
        wp[XCM * 32 + l31] = mrun;
        wp[(XCM + 1) * 32 + l31] = l;
    }
}
template <int XCW>
__global__ __launch_bounds__(256) void smm_xattn_kernel(const float* __restrict__ qf, const float* __restrict__ mem, float* __restrict__ ws,
                                                        int rows, int N, int nsplit, int kps, float scale) {
    smm_xattn_body<XCW>(qf, mem, ws, rows, N, nsplit, kps, scale, blockIdx.y, blockIdx.x);
}

// Wave-per-key-block form for the narrow (compact) memories, CM = 72 / 136: every wave owns whole 32-key blocks (wave w of a workgroup takes
// blocks w, w + 4, ...) with ALL CM channels, so S is complete inside the wave -- no partial tiles through LDS, no barrier per block --
// and P.V runs on ceil(CM / 32) channel blocks instead of 4 x 32 (84 instead of 100 MFMAs per 32 keys at CM = 72).  The four waves keep
// a running max / sum / partial output each and merge them in LDS at the end (wave order): one partial per workgroup, as before.  profiles/r03: the channel-split
// form streamed the 72-row memory at 1.85 TB/s (12 launches, 1.3 ms per step).
template <int CM>
__device__ __forceinline__ void smm_xattn_w_body(const float* __restrict__ qf, const float* __restrict__ mem, float* __restrict__ ws, int rows, int N,
                                                 int nsplit, int kps, float scale, const int b, const int sp) {
    constexpr int CB = (CM + 31) / 32;           // 32-channel blocks of the P.V product
    constexpr int TILE = CB * 32 * 33;           // per-wave mem slice [CB*32 c][33]; rows >= CM stay zero
    constexpr int NF4 = (CM * 8 + 63) / 64;      // float4 loads per lane per 32-key block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    float* tile = smem + wave * TILE;
    const float* memb = mem + (long long)b * CM * N;
    for (int i = lane; i < TILE; i += 64) tile[i] = 0.f;  // (wave-private; LDS is in order within a wave)

    float qreg[CM / 2];
#pragma unroll
    for (int t = 0; t < CM / 2; ++t) qreg[t] = l31 < rows ? qf[((long long)b * rows + l31) * CM + 2 * t + half] : 0.f;
    floatx16 O[CB];
#pragma unroll
    for (int m = 0; m < CB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const int nkb = (N + 31) / 32;
    const int kb_begin = sp * kps + wave;
    const int kb_end = min(nkb, sp * kps + kps);
    floatx4 rt[NF4];
    auto load_tile = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;  // float4 index in [CM][8]
            const int c = f >> 3, j4 = (f & 7) * 4;
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            if ((CM * 8) % 64 != 0 && f >= CM * 8) {
            } else if (key0 + j4 + 3 < N)
                z = *reinterpret_cast<const floatx4*>(memb + (long long)c * N + key0 + j4);
            else
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) z[e] = memb[(long long)c * N + key0 + j4 + e];
            rt[i] = z;
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;
            const int c = f >> 3, j4 = (f & 7) * 4;
            if ((CM * 8) % 64 != 0 && f >= CM * 8) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[c * 33 + j4 + e] = rt[i][e];
        }
    };
    if (kb_begin < kb_end) load_tile(kb_begin);
    for (int kbi = kb_begin; kbi < kb_end; kbi += 4) {
        write_tile();
        if (kbi + 4 < kb_end) load_tile(kbi + 4);
        floatx16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int t = 0; t < CM / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(2 * t + half) * 33 + l31], qreg[t], S, 0, 0, 0);
        const int key0 = kbi * 32 + 4 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = (key0 + KAPPA(r) < N) ? S[r] * scale : -INFINITY;
            S[r] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(S[r] - mnew);
            S[r] = p;
            ps += p;
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
        // (the running maximum of a row rarely moves after the first key blocks: alpha == 1 in every lane -> the products are skipped,
        // the result is the same bits)
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int m = 0; m < CB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < CB; ++m)
                O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(m * 32 + l31) * 33 + KAPPA(r) + 4 * half], S[r], O[m], 0, 0, 0);
    }
    // the four waves' partials meet in LDS (each in its own, now idle, tile area: [c][32 rows], then the m and l rows) and leave as
    // ONE partial per workgroup, combined in wave order
    static_assert((CM + 2) * 32 <= TILE, "partial does not fit the tile area");
    const float l = lrun + __shfl_xor(lrun, 32, 64);
#pragma unroll
    for (int m = 0; m < CB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = m * 32 + KAPPA(r) + 4 * half;
            if (cl < CM) tile[cl * 32 + l31] = O[m][r];
        }
    if (half == 0) {
        tile[CM * 32 + l31] = mrun;
        tile[(CM + 1) * 32 + l31] = l;
    }
    __syncthreads();
    float* wp = ws + ((long long)b * nsplit + sp) * (CM + 2) * 32;
    for (int i = tid; i < CM * 32; i += 256) {
        const int row = i & 31;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, smem[w * TILE + CM * 32 + row]);
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float ms = smem[w * TILE + CM * 32 + row];
            acc += smem[w * TILE + i] * (ms == -INFINITY ? 0.f : __expf(ms - M));
        }
        wp[i] = acc;
    }
    if (tid < 32) {
        float M = -INFINITY, L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, smem[w * TILE + CM * 32 + tid]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float ms = smem[w * TILE + CM * 32 + tid];
            L += smem[w * TILE + (CM + 1) * 32 + tid] * (ms == -INFINITY ? 0.f : __expf(ms - M));
        }
        wp[CM * 32 + tid] = M;
        wp[(CM + 1) * 32 + tid] = L;
    }
}
template <int CM>
__global__ __launch_bounds__(256) void smm_xattn_w_kernel(const float* __restrict__ qf, const float* __restrict__ mem, float* __restrict__ ws, int rows,
                                                          int N, int nsplit, int kps, float scale) {
    smm_xattn_w_body<CM>(qf, mem, ws, rows, N, nsplit, kps, scale, blockIdx.y, blockIdx.x);
}

// Merge of the key splits' partial (max, sum, P.V) triples.  Workgroup = 2 channels x 32 rows x CSL split lanes: lane sl walks the
// splits sl, sl + CSL, .. (max pass, then the weighted sums: every load of a pass independent of the others), the CSL lane results
// are merged in lane order through LDS -- a fixed order that is a function of the split count (of N) alone.  (r04: one thread per
// (channel, row) had walked all 64 splits twice: 15.6 us per launch of load latency, 24 launches per step.)
constexpr int CSL = 4;
__device__ __forceinline__ void smm_xattn_combine_body(const float* __restrict__ ws, float* __restrict__ o, int rows, int nsplit, int XCM,
                                                       float* __restrict__ lse, const int b, const int bx) {
    __shared__ float red[3][CSL][64];
    const int io = threadIdx.x & 63, sl = threadIdx.x >> 6;  // output of the workgroup, split lane (= wave)
    const int i = bx * 64 + io;                              // over XCM * 32: the partials are laid out [c][32 rows]
    const int c = i >> 5, row = i & 31;
    const bool live = i < XCM * 32 && row < rows;
    const long long ss = (long long)(XCM + 2) * 32;  // floats per split
    const float* wb = ws + (long long)b * nsplit * ss;
    float M = -INFINITY, L = 0.f, acc = 0.f;
    if (live) {
#pragma unroll 8
        for (int s = sl; s < nsplit; s += CSL) M = fmaxf(M, wb[s * ss + XCM * 32 + row]);
#pragma unroll 8
        for (int s = sl; s < nsplit; s += CSL) {
            const float* w = wb + s * ss;
            const float ms = w[XCM * 32 + row];
            const float f = ms == -INFINITY ? 0.f : __expf(ms - M);
            L += w[(XCM + 1) * 32 + row] * f;
            acc += w[c * 32 + row] * f;
        }
    }
    red[0][sl][io] = M, red[1][sl][io] = L, red[2][sl][io] = acc;
    __syncthreads();
    if (sl == 0 && live) {
        float Mt = red[0][0][io];
#pragma unroll
        for (int l = 1; l < CSL; ++l) Mt = fmaxf(Mt, red[0][l][io]);
        float Lt = 0.f, At = 0.f;
#pragma unroll
        for (int l = 0; l < CSL; ++l) {
            const float ml = red[0][l][io];
            const float f = ml == -INFINITY ? 0.f : __expf(ml - Mt);
            Lt += red[1][l][io] * f;
            At += red[2][l][io] * f;
        }
        o[((long long)b * rows + row) * XCM + c] = At / Lt;
        if (lse && c == 0) lse[(long long)b * rows + row] = Mt + __logf(Lt);  // log-sum-exp of the scaled scores (training: saved for the backward)
    }
}
__global__ __launch_bounds__(256) void smm_xattn_combine_kernel(const float* __restrict__ ws, float* __restrict__ o, int rows, int nsplit, int XCM,
                                                                float* __restrict__ lse) {
    smm_xattn_combine_body(ws, o, rows, nsplit, XCM, lse, blockIdx.y, blockIdx.x);
}

// Grouped launches (idiff_smm_xattn_grouped_fwd): the cross-attentions of SEVERAL ScoreMapModules (the four UNet levels of a net, which
// advance their decoder chains in lock step) in ONE attention launch + ONE merge launch.  blockIdx.z picks the problem -- its own
// operands, memory width and key count, descriptors in the kernel arguments as for the grouped token linears -- and the problem's
// kernel body (the very code of the single launches: same per-sample arithmetic and order, so the results are the same bits); the grid
// covers the largest split count, blocks beyond a smaller problem's exit.  The small levels (1 024 / 4 096 keys: a few dozen
// workgroups) then run beside the 65 536-key level instead of in launches of their own.
struct XattnGroups {
    idiff_xattn_group g[IDIFF_XATTN_MAX_GROUPS];
    int nsplit[IDIFF_XATTN_MAX_GROUPS], kps[IDIFF_XATTN_MAX_GROUPS], kind[IDIFF_XATTN_MAX_GROUPS];
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void smm_xattn_grouped_kernel(const XattnGroups args, int rows, float scale) {
    const int z = blockIdx.z;
    const idiff_xattn_group& d = args.g[z];
    const int ns = args.nsplit[z], kps = args.kps[z];
    if ((int)blockIdx.x >= ns) return;  // uniform
    switch (args.kind[z]) {             // uniform
        case 0: smm_xattn_body<64>(d.qf, d.mem, d.ws, rows, d.N, ns, kps, scale, blockIdx.y, blockIdx.x); break;
        case 1: smm_xattn_body<34>(d.qf, d.mem, d.ws, rows, d.N, ns, kps, scale, blockIdx.y, blockIdx.x); break;
        case 2: smm_xattn_body<18>(d.qf, d.mem, d.ws, rows, d.N, ns, kps, scale, blockIdx.y, blockIdx.x); break;
        default: smm_xattn_w_body<72>(d.qf, d.mem, d.ws, rows, d.N, ns, kps, scale, blockIdx.y, blockIdx.x); break;
    }
}
__global__ __launch_bounds__(256) void smm_xattn_combine_grouped_kernel(const XattnGroups args, int rows) {
    const int z = blockIdx.z;
    const idiff_xattn_group& d = args.g[z];
    if ((int)blockIdx.x * 64 >= d.Cm * 32) return;  // uniform
    smm_xattn_combine_body(d.ws, d.o, rows, args.nsplit[z], d.Cm, nullptr, blockIdx.y, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// Backward of the ScoreMapModule cross-attention over the full 256-row memory (training path), one pass over the keys:
//   S = scale * qf mem,  P = exp(S - lse),  o = P mem^T          (forward; lse saved)
//   dP = do mem,  D = rowsum(do * o),  G = scale * P * (dP - D)
//   dqf = G mem^T          (per key split; summed over the splits in a fixed order by smm_xattn_bwd_combine_kernel)
//   dmem = do^T P + qf^T G
// replacing, per decoder layer, seven batched-GEMM / softmax launches that each streamed the [B, 256, N] memory or the [B, rows, N]
// score matrices through HBM (91 ms of a 400-ms training step at 256x256, batch 32).  Same tiling as the forward kernel: 4 waves
// split the 256 channels (partial S / dP tiles meet in LDS), 32-key blocks, f32 MFMA 32x32x2 throughout; P and G are taken straight
// from the accumulators as B operands for dqf, and through a 32x32 LDS transpose (keys onto the lanes) for dmem.
//   LDS: mem tile 4 x [64][33]; qf and do as [32 rows][257]; exchange / transpose area 4 x 2304; row constants.
// XCW = channels per wave: 64 (the 256-row memory) or 18 (the compact 72-row memory of the narrow levels, r05: rows >= C + 1 of it are
// padding -- their dmem rows are written and never read).  The 18-channel form pads a wave's P.V-shaped products to one 32-channel
// block (as the forward kernel does) and needs 72 KB of LDS instead of 137: two workgroups per CU.
template <int XCW>
__global__ __launch_bounds__(256) void smm_xattn_bwd_kernel(const float* __restrict__ qf, const float* __restrict__ mem, const float* __restrict__ o,
                                                            const float* __restrict__ lse, const float* __restrict__ d_o, float* __restrict__ ws,
                                                            float* __restrict__ dmem, int rows, int N, int nsplit, int kps, float scale,
                                                            int accumulate) {
    constexpr int XCM = 4 * XCW, XCB = (XCW + 31) / 32, XTILE = XCB * 32 * 33, XP = XCM + 1, XREG = 2304;
    constexpr int NF4 = (XCW * 8 + 63) / 64;  // float4 loads per lane per 32-key block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem + (threadIdx.x >> 6) * XTILE;  // private per wave: [64 c][33]
    float* qL = smem + 4 * XTILE;                     // [32][257]
    float* dL = qL + 32 * XP;                         // [32][257]
    float* xch = dL + 32 * XP;                        // [4][2304]: partial S | dP tiles, then each wave's transposed P | G
    float* rowc = xch + 4 * XREG;                     // [2][32]: lse, D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int c0 = wave * XCW;
    const float* memb = mem + (long long)b * XCM * N + (long long)c0 * N;
    float* dmemb = dmem + (long long)b * XCM * N + (long long)c0 * N;

    // qf, do -> LDS (rows beyond `rows` are zero); D = rowsum(do * o): 8 threads per row, XCM / 8 channels each
    if (XCW % 32 != 0)
        for (int i = lane; i < XTILE; i += 64) tile[i] = 0.f;  // the padding rows of the wave's slice feed MFMA rows that are never stored
    for (int i = tid; i < 32 * XCM; i += 256) {
        const int r = i / XCM, c = i - r * XCM;
        const bool v = r < rows;
        qL[r * XP + c] = v ? qf[((long long)b * rows + r) * XCM + c] : 0.f;
        dL[r * XP + c] = v ? d_o[((long long)b * rows + r) * XCM + c] : 0.f;
    }
    {
        const int r = tid >> 3, part = tid & 7;
        float acc = 0.f;
        if (r < rows)
            for (int c = part * (XCM / 8); c < (part + 1) * (XCM / 8); ++c) acc += d_o[((long long)b * rows + r) * XCM + c] * o[((long long)b * rows + r) * XCM + c];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) {
            rowc[32 + r] = acc;
            rowc[r] = r < rows ? lse[(long long)b * rows + r] : INFINITY;  // padding rows: P = exp(-inf) = 0
        }
    }
    __syncthreads();
    const float my_lse = rowc[l31], my_D = rowc[32 + l31];

    floatx16 Oq[XCB];
#pragma unroll
    for (int m = 0; m < XCB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) Oq[m][r] = 0.f;

    const int nkb = (N + 31) / 32;
    const int kb_begin = sp * kps;
    const int kb_end = min(nkb, kb_begin + kps);
    floatx4 rt[NF4];  // XCW c x 32 keys / 64 lanes
    auto load_tile = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;  // float4 index in [XCW][8]
            const int c = f >> 3, j4 = (f & 7) * 4;
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            if (XCW % 8 != 0 && f >= XCW * 8) {
            } else if (key0 + j4 + 3 < N)
                z = *reinterpret_cast<const floatx4*>(memb + (long long)c * N + key0 + j4);
            else
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) z[e] = memb[(long long)c * N + key0 + j4 + e];
            rt[i] = z;
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;
            const int c = f >> 3, j4 = (f & 7) * 4;
            if (XCW % 8 != 0 && f >= XCW * 8) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[c * 33 + j4 + e] = rt[i][e];
        }
    };
    if (kb_begin < kb_end) {
        load_tile(kb_begin);
        write_tile();
    }
    const float* qB = qL + l31 * XP + c0 + half;  // B operand of the S product: X[row = l31][c0 + 2t + half]
    const float* dB = dL + l31 * XP + c0 + half;
    for (int kbi = kb_begin; kbi < kb_end; ++kbi) {
        if (kbi + 1 < kb_end) load_tile(kbi + 1);
        // ---- partial S and dP over this wave's 64 channels: [32 keys][32 rows] ----
        floatx16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f, dP[r] = 0.f;
#pragma unroll
        for (int t = 0; t < XCW / 2; ++t) {
            const float a = tile[(2 * t + half) * 33 + l31];
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qB[2 * t], S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dB[2 * t], dP, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xch[wave * XREG + r * 64 + lane] = S[r];
            xch[wave * XREG + 1024 + r * 64 + lane] = dP[r];
        }
        __syncthreads();
        const int key0 = kbi * 32 + 4 * half;
        floatx16 G;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = ((xch[r * 64 + lane] + xch[XREG + r * 64 + lane]) + xch[2 * XREG + r * 64 + lane]) + xch[3 * XREG + r * 64 + lane];
            const float dv = ((xch[1024 + r * 64 + lane] + xch[XREG + 1024 + r * 64 + lane]) + xch[2 * XREG + 1024 + r * 64 + lane]) +
                             xch[3 * XREG + 1024 + r * 64 + lane];
            const float p = (key0 + KAPPA(r) < N) ? __expf(sv * scale - my_lse) : 0.f;
            S[r] = p;
            G[r] = scale * p * (dv - my_D);
        }
        __syncthreads();  // every wave has read the partial tiles: the area now takes the transposed P | G of each wave
        // ---- dqf^T[c][row] += sum_key mem[c][key] G[key][row]  (G straight from the accumulator layout) ----
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < XCB; ++m)
                Oq[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(m * 32 + l31) * 33 + KAPPA(r) + 4 * half], G[r], Oq[m], 0, 0, 0);
        // ---- dmem[c][key] = sum_row do[row][c] P[key][row] + qf[row][c] G[key][row]: P, G with the keys on the lanes ----
        float* pt = xch + wave * XREG;   // [32 rows][33]
        float* gt = pt + 32 * 33;        // [32 rows][33]   (2 * 1056 = 2112 <= 2304)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pt[l31 * 33 + KAPPA(r) + 4 * half] = S[r];
            gt[l31 * 33 + KAPPA(r) + 4 * half] = G[r];
        }
        // (wave-private area: the reads below follow the writes in program order; LDS is in order within a wave)
        floatx16 Dm[XCB];
#pragma unroll
        for (int m = 0; m < XCB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Dm[m][r] = 0.f;
        // (a wave's last channel block may reach past its XCW channels: lanes beyond read a neighbour's / the next row's values -- finite
        // numbers that only enter accumulator rows no store ever reads; the clamp keeps the address inside the row arrays)
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int row = 2 * sidx + half;
            const float pb = pt[row * 33 + l31], gb = gt[row * 33 + l31];
#pragma unroll
            for (int m = 0; m < XCB; ++m) {
                const int cc = XCW % 32 == 0 ? c0 + m * 32 + l31 : min(c0 + m * 32 + l31, XCM - 1);
                Dm[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(dL[row * XP + cc], pb, Dm[m], 0, 0, 0);
                Dm[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(qL[row * XP + cc], gb, Dm[m], 0, 0, 0);
            }
        }
        const int key = kbi * 32 + l31;
        if (key < N) {
            if (accumulate) {  // uniform: dmem += (the gradients of the decoder layers that share this memory meet here, in a fixed order)
#pragma unroll
                for (int m = 0; m < XCB; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cl = m * 32 + KAPPA(r) + 4 * half;
                        if (cl < XCW) Dm[m][r] += dmemb[(long long)cl * N + key];
                    }
            }
#pragma unroll
            for (int m = 0; m < XCB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = m * 32 + KAPPA(r) + 4 * half;
                    if (cl < XCW) dmemb[(long long)cl * N + key] = Dm[m][r];
                }
        }
        if (kbi + 1 < kb_end) write_tile();
        __syncthreads();  // the tile and the exchange area are rewritten by the next block
    }
    // partial dqf: ws[b][sp][c][row]
    float* wp = ws + ((long long)b * nsplit + sp) * XCM * 32;
#pragma unroll
    for (int m = 0; m < XCB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = m * 32 + KAPPA(r) + 4 * half;
            if (cl < XCW) wp[(c0 + cl) * 32 + l31] = Oq[m][r];
        }
}

// dqf[b][row][c] = sum over the key splits, in order
__global__ __launch_bounds__(256) void smm_xattn_bwd_combine_kernel(const float* __restrict__ ws, float* __restrict__ dqf, int rows, int nsplit, int XCM) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over XCM * 32
    const int c = i >> 5, row = i & 31;
    if (c >= XCM || row >= rows) return;
    const float* wb = ws + (long long)b * nsplit * XCM * 32;
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < nsplit; ++s) acc += wb[(long long)s * XCM * 32 + c * 32 + row];
    dqf[((long long)b * rows + row) * XCM + c] = acc;
}

inline void smm_split(int B, int N, int* nsplit, int* kps) {
    // The split is a function of N alone, so a sample's reduction order (and its bits) does not depend on the batch it sits
    // in: 64..2048 keys per split, at most 32 splits up to N = 65536 (128 at the 512x512 level).  r05: half the splits of r04 --
    // at batch 16 the 65 536-key level is 512 workgroups, one round of the two that fit a CU, each wave walks 16 key blocks instead
    // of 8 behind the same fixed cost (tile reset, query rows, first tile's latency, the waves' merge), and the merge launch reads
    // half the partials.  IDIFF_XATTN_SPLITS=64 (A/B runs): the r04 rule.
    (void)B;
    static const int per = [] {
        const char* e = getenv("IDIFF_XATTN_SPLITS");
        return e && atoi(e) == 64 ? 64 : 32;
    }();
    const int nkb = (N + 31) / 32;
    int k = nkb / per;
    if (k < 2) k = 2;
    if (k > 2048 / per) k = 2048 / per;
    if (k > nkb) k = nkb;
    *kps = k;
    *nsplit = (nkb + k - 1) / k;
}

}  // namespace

extern "C" int idiff_attn_self_fwd(const float* qkv, float* out, float* lse, int B, int C, int N, int heads, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(qkv && out && B > 0 && C > 0 && N > 0 && heads > 0 && C % heads == 0, "attn_self: bad args");
    const int dh = C / heads;
    IDIFF_CHECK_ARG(dh == 64 || dh == 32, "attn_self: head dim must be 32 or 64 (got %d)", dh);
    IDIFF_CHECK_ARG(N % 4 == 0, "attn_self: N must be a multiple of 4");
    dim3 grid((N + 127) / 128, B * heads);
    hipStream_t st = (hipStream_t)stream;
    if (dh == 64) {
        const size_t lds = 2 * (64 * 32 + 64 * 33 + 3) * sizeof(float);
        hipLaunchKernelGGL(attn_self_kernel<64>, grid, dim3(256), lds, st, qkv, out, lse, C, N, heads, scale);
    } else {
        const size_t lds = 2 * (32 * 32 + 32 * 33 + 3) * sizeof(float);
        hipLaunchKernelGGL(attn_self_kernel<32>, grid, dim3(256), lds, st, qkv, out, lse, C, N, heads, scale);
    }
    IDIFF_CHECK_LAUNCH("attn_self_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_self_bf16_fwd(const float* qkv, float* out, int B, int C, int N, int heads, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(qkv && out && B > 0 && C > 0 && N > 0 && heads > 0 && C % heads == 0, "attn_self_bf16: bad args");
    IDIFF_CHECK_ARG(C / heads == 64, "attn_self_bf16: head dim must be 64 (got %d)", C / heads);
    IDIFF_CHECK_ARG(N % 4 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "attn_self_bf16: N %% 4 and 16-byte alignment required");
    dim3 grid((N + 127) / 128, B * heads);
    const size_t lds = (size_t)2 * (32 * (64 + 8) + 64 * (32 + 4)) * sizeof(uint16_t);
    hipLaunchKernelGGL(attn_self_half_kernel<__bf16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, out, C, N, heads, scale);
    IDIFF_CHECK_LAUNCH("attn_self_bf16_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_self_f16_fwd(const float* qkv, float* out, int B, int C, int N, int heads, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(qkv && out && B > 0 && C > 0 && N > 0 && heads > 0 && C % heads == 0, "attn_self_f16: bad args");
    IDIFF_CHECK_ARG(C / heads == 64, "attn_self_f16: head dim must be 64 (got %d)", C / heads);
    IDIFF_CHECK_ARG(N % 4 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "attn_self_f16: N %% 4 and 16-byte alignment required");
    dim3 grid((N + 127) / 128, B * heads);
    const size_t lds = (size_t)2 * (32 * (64 + 8) + 64 * (32 + 4)) * sizeof(uint16_t);
    hipLaunchKernelGGL(attn_self_half_kernel<_Float16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, out, C, N, heads, scale);
    IDIFF_CHECK_LAUNCH("attn_self_f16_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_ctx_fwd(const float* q, const float* k, const float* v, float* out, int B, int C, int N, int M, int heads, float scale,
                                  idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && out && B > 0 && C > 0 && N > 0 && heads > 0 && C % heads == 0, "attn_ctx: bad args");
    IDIFF_CHECK_ARG(M >= 1 && M <= 32, "attn_ctx: M must be in 1..32 (got %d)", M);
    const int dh = C / heads;
    dim3 grid((N + 255) / 256, heads, B);
    const size_t lds = (size_t)2 * M * dh * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dh == 16)
        hipLaunchKernelGGL(attn_ctx_kernel<16>, grid, dim3(256), lds, st, q, k, v, out, C, N, M, scale);
    else if (dh == 32)
        hipLaunchKernelGGL(attn_ctx_kernel<32>, grid, dim3(256), lds, st, q, k, v, out, C, N, M, scale);
    else if (dh == 64)
        hipLaunchKernelGGL(attn_ctx_kernel<64>, grid, dim3(256), lds, st, q, k, v, out, C, N, M, scale);
    else
        IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "attn_ctx: head dim must be 16, 32 or 64 (got %d)", dh);
    IDIFF_CHECK_LAUNCH("attn_ctx_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_tokens_fwd(const float* q, const float* k, const float* v, float* out, int B, int Nq, int M, int C, int heads,
                                     float scale, int64_t ldq, int64_t ldkv, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && out && B > 0 && Nq > 0 && C > 0 && heads > 0 && C % heads == 0, "attn_tokens: bad args");
    IDIFF_CHECK_ARG(M >= 1 && M <= 64, "attn_tokens: M must be in 1..64 (got %d)", M);
    IDIFF_CHECK_ARG(ldq >= C && ldkv >= C, "attn_tokens: row strides must be >= C");
    hipLaunchKernelGGL(attn_tokens_kernel, dim3(B * heads * Nq), dim3(64), 0, (hipStream_t)stream, q, k, v, out, Nq, M, C, heads, scale,
                       (long long)ldq, (long long)ldkv);
    IDIFF_CHECK_LAUNCH("attn_tokens_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_tokens_bwd(const float* q, const float* k, const float* v, const float* d_o, float* dq, float* dk, float* dv, int B, int Nq,
                                     int M, int C, int heads, float scale, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t lddq, int64_t lddkv,
                                     idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && d_o && dq && dk && dv && B > 0 && C > 0 && heads > 0 && C % heads == 0, "attn_tokens_bwd: bad args");
    IDIFF_CHECK_ARG(Nq >= 1 && Nq <= ATB && M >= 1 && M <= ATB, "attn_tokens_bwd: Nq and M must be in 1..%d (got %d, %d)", ATB, Nq, M);
    IDIFF_CHECK_ARG(C / heads <= 64, "attn_tokens_bwd: head dim must be <= 64 (got %d)", C / heads);
    IDIFF_CHECK_ARG(ldq >= C && ldkv >= C && ldo >= C && lddq >= C && lddkv >= C, "attn_tokens_bwd: row strides must be >= C");
    hipLaunchKernelGGL(attn_tokens_bwd_kernel, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, q, k, v, d_o, dq, dk, dv, Nq, M, C, heads, scale,
                       (long long)ldq, (long long)ldkv, (long long)ldo, (long long)lddq, (long long)lddkv);
    IDIFF_CHECK_LAUNCH("attn_tokens_bwd");
    return IDIFF_OK;
}

extern "C" int idiff_attn_tokens_grouped_fwd(const float* const* q, const float* const* k, const float* const* v, float* const* out, int ngroups,
                                             int B, int Nq, int M, int C, int heads, float scale, int64_t ldq, int64_t ldkv, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && out && ngroups >= 1 && ngroups <= IDIFF_LINEAR_MAX_GROUPS, "attn_tokens_grouped: 1..%d groups", IDIFF_LINEAR_MAX_GROUPS);
    IDIFF_CHECK_ARG(B > 0 && Nq > 0 && C > 0 && heads > 0 && C % heads == 0, "attn_tokens_grouped: bad args");
    IDIFF_CHECK_ARG(M >= 1 && M <= 64, "attn_tokens_grouped: M must be in 1..64 (got %d)", M);
    IDIFF_CHECK_ARG(ldq >= C && ldkv >= C, "attn_tokens_grouped: row strides must be >= C");
    AttnTokGroups g;
    memset(&g, 0, sizeof(g));
    for (int i = 0; i < ngroups; ++i) {
        IDIFF_CHECK_ARG(q[i] && k[i] && v[i] && out[i], "attn_tokens_grouped: group %d: null pointer", i);
        g.q[i] = q[i], g.k[i] = k[i], g.v[i] = v[i], g.out[i] = out[i];
    }
    hipLaunchKernelGGL(attn_tokens_grouped_kernel, dim3(B * heads * Nq, ngroups), dim3(64), 0, (hipStream_t)stream, g, Nq, M, C, heads, scale,
                       (long long)ldq, (long long)ldkv);
    IDIFF_CHECK_LAUNCH("attn_tokens_grouped_fwd");
    return IDIFF_OK;
}

extern "C" int64_t idiff_smm_xattn_ws_floats(int B, int Nq, int heads, int Cm, int N) {
    (void)Nq;
    (void)heads;
    int ns, kps;
    smm_split(B, N, &ns, &kps);
    return (int64_t)B * ns * (Cm + 2) * 32;
}

static int smm_xattn_fwd_impl(const float* qf, const float* mem, float* o, float* lse, float* ws, int B, int Nq, int heads, int Cm, int N, float scale,
                              idiff_stream_t stream) {
    IDIFF_CHECK_ARG(qf && mem && o && ws && B > 0 && Nq > 0 && heads > 0 && N > 0, "smm_xattn: bad args");
    IDIFF_CHECK_ARG(Cm == 256 || Cm == 136 || Cm == 72, "smm_xattn: Cm must be 72, 136 or 256 (got %d)", Cm);
    IDIFF_CHECK_ARG(Nq * heads <= 32, "smm_xattn: Nq*heads must be <= 32 (got %d)", Nq * heads);
    IDIFF_CHECK_ARG(N % 4 == 0, "smm_xattn: N must be a multiple of 4");
    int ns, kps;
    smm_split(B, N, &ns, &kps);
    const int rows = Nq * heads;
    hipStream_t st = (hipStream_t)stream;
    const int xcb = (Cm / 4 + 31) / 32;
    const size_t lds = (size_t)(4 * xcb * 32 * 33 + 4 * 16 * 64) * sizeof(float);
    int ns_eff = ns;  // partials the combine kernel walks
    if (Cm == 256)
        hipLaunchKernelGGL(smm_xattn_kernel<64>, dim3(ns, B), dim3(256), lds, st, qf, mem, ws, rows, N, ns, kps, scale);
    else if (Cm == 136)
        hipLaunchKernelGGL(smm_xattn_kernel<34>, dim3(ns, B), dim3(256), lds, st, qf, mem, ws, rows, N, ns, kps, scale);
    else if (kps >= 4) {  // compact 72-row memory: wave-per-key-block form (every wave has at least one block)
        static const bool wform_off = [] {
            const char* e = getenv("IDIFF_XATTN_WFORM");
            return e && e[0] == '0';
        }();
        if (!wform_off) {
            const size_t ldsw = (size_t)4 * 3 * 32 * 33 * sizeof(float);
            hipLaunchKernelGGL(smm_xattn_w_kernel<72>, dim3(ns, B), dim3(256), ldsw, st, qf, mem, ws, rows, N, ns, kps, scale);
        } else {
            hipLaunchKernelGGL(smm_xattn_kernel<18>, dim3(ns, B), dim3(256), lds, st, qf, mem, ws, rows, N, ns, kps, scale);
        }
    } else
        hipLaunchKernelGGL(smm_xattn_kernel<18>, dim3(ns, B), dim3(256), lds, st, qf, mem, ws, rows, N, ns, kps, scale);
    IDIFF_CHECK_LAUNCH("smm_xattn_fwd");
    hipLaunchKernelGGL(smm_xattn_combine_kernel, dim3((32 * Cm + 63) / 64, B), dim3(256), 0, st, ws, o, rows, ns_eff, Cm, lse);
    IDIFF_CHECK_LAUNCH("smm_xattn_combine");
    return IDIFF_OK;
}

extern "C" int idiff_smm_xattn_grouped_fwd(const idiff_xattn_group* groups, int ngroups, int B, int Nq, int heads, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(groups && ngroups >= 1 && ngroups <= IDIFF_XATTN_MAX_GROUPS, "smm_xattn_grouped: 1..%d groups", IDIFF_XATTN_MAX_GROUPS);
    IDIFF_CHECK_ARG(B > 0 && Nq > 0 && heads > 0 && Nq * heads <= 32, "smm_xattn_grouped: Nq*heads must be in 1..32 (got %d)", Nq * heads);
    XattnGroups args;
    memset(&args, 0, sizeof(args));
    int gx = 0, gc = 0;
    size_t lds = 0;
    for (int i = 0; i < ngroups; ++i) {
        const idiff_xattn_group& d = groups[i];
        IDIFF_CHECK_ARG(d.qf && d.mem && d.o && d.ws && d.N > 0 && d.N % 4 == 0, "smm_xattn_grouped: group %d: bad args", i);
        IDIFF_CHECK_ARG(d.Cm == 256 || d.Cm == 136 || d.Cm == 72, "smm_xattn_grouped: group %d: Cm must be 72, 136 or 256 (got %d)", i, d.Cm);
        args.g[i] = d;
        smm_split(B, d.N, &args.nsplit[i], &args.kps[i]);
        // the kernel each problem would get from idiff_smm_xattn_fwd (same choice, same bits)
        const int xcb = (d.Cm / 4 + 31) / 32;
        size_t l = (size_t)(4 * xcb * 32 * 33 + 4 * 16 * 64) * sizeof(float);
        if (d.Cm == 256) args.kind[i] = 0;
        else if (d.Cm == 136) args.kind[i] = 1;
        else if (args.kps[i] >= 4) args.kind[i] = 3, l = (size_t)4 * 3 * 32 * 33 * sizeof(float);
        else args.kind[i] = 2;
        lds = max(lds, l);
        gx = max(gx, args.nsplit[i]);
        gc = max(gc, (32 * d.Cm + 63) / 64);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smm_xattn_grouped_kernel, dim3(gx, B, ngroups), dim3(256), lds, st, args, Nq * heads, scale);
    IDIFF_CHECK_LAUNCH("smm_xattn_grouped_fwd");
    hipLaunchKernelGGL(smm_xattn_combine_grouped_kernel, dim3(gc, B, ngroups), dim3(256), 0, st, args, Nq * heads);
    IDIFF_CHECK_LAUNCH("smm_xattn_grouped_combine");
    return IDIFF_OK;
}

extern "C" int idiff_smm_xattn_fwd(const float* qf, const float* mem, float* o, float* ws, int B, int Nq, int heads, int Cm, int N, float scale,
                                   idiff_stream_t stream) {
    return smm_xattn_fwd_impl(qf, mem, o, nullptr, ws, B, Nq, heads, Cm, N, scale, stream);
}

extern "C" int idiff_smm_xattn_lse_fwd(const float* qf, const float* mem, float* o, float* lse, float* ws, int B, int rows, int N, float scale,
                                       idiff_stream_t stream) {
    IDIFF_CHECK_ARG(lse, "smm_xattn_lse_fwd: null lse");
    return smm_xattn_fwd_impl(qf, mem, o, lse, ws, B, rows, 1, 256, N, scale, stream);
}

extern "C" int idiff_smm_xattn_cm_lse_fwd(const float* qf, const float* mem, float* o, float* lse, float* ws, int B, int rows, int Cm, int N,
                                          float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(lse, "smm_xattn_cm_lse_fwd: null lse");
    return smm_xattn_fwd_impl(qf, mem, o, lse, ws, B, rows, 1, Cm, N, scale, stream);
}

static int smm_xattn_bwd_impl(const float* qf, const float* mem, const float* o, const float* lse, const float* d_o, float* dqf, float* dmem,
                              int accumulate, float* ws, int B, int rows, int Cm, int N, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(qf && mem && o && lse && d_o && dqf && dmem && ws && B > 0 && N > 0, "smm_xattn_bwd: bad args");
    IDIFF_CHECK_ARG(rows >= 1 && rows <= 32, "smm_xattn_bwd: rows must be in 1..32 (got %d)", rows);
    IDIFF_CHECK_ARG(Cm == 256 || Cm == 72, "smm_xattn_bwd: Cm must be 72 or 256 (got %d)", Cm);
    IDIFF_CHECK_ARG(N % 4 == 0, "smm_xattn_bwd: N must be a multiple of 4");
    int ns, kps;
    smm_split(B, N, &ns, &kps);
    hipStream_t st = (hipStream_t)stream;
    const int xcb = (Cm / 4 + 31) / 32;
    const size_t lds = (size_t)(4 * xcb * 32 * 33 + 2 * 32 * (Cm + 1) + 4 * 2304 + 64) * sizeof(float);
    static bool attr[2] = {false, false};
    const int which = Cm == 256 ? 0 : 1;
    if (!attr[which]) {
        hipError_t e = Cm == 256 ? hipFuncSetAttribute(reinterpret_cast<const void*>(smm_xattn_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                 : hipFuncSetAttribute(reinterpret_cast<const void*>(smm_xattn_bwd_kernel<18>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "smm_xattn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr[which] = true;
    }
    if (Cm == 256)
        hipLaunchKernelGGL(smm_xattn_bwd_kernel<64>, dim3(ns, B), dim3(256), lds, st, qf, mem, o, lse, d_o, ws, dmem, rows, N, ns, kps, scale, accumulate);
    else
        hipLaunchKernelGGL(smm_xattn_bwd_kernel<18>, dim3(ns, B), dim3(256), lds, st, qf, mem, o, lse, d_o, ws, dmem, rows, N, ns, kps, scale, accumulate);
    IDIFF_CHECK_LAUNCH("smm_xattn_bwd");
    hipLaunchKernelGGL(smm_xattn_bwd_combine_kernel, dim3((Cm * 32 + 255) / 256, B), dim3(256), 0, st, ws, dqf, rows, ns, Cm);
    IDIFF_CHECK_LAUNCH("smm_xattn_bwd_combine");
    return IDIFF_OK;
}

extern "C" int idiff_smm_xattn_bwd(const float* qf, const float* mem, const float* o, const float* lse, const float* d_o, float* dqf, float* dmem,
                                   int accumulate, float* ws, int B, int rows, int N, float scale, idiff_stream_t stream) {
    return smm_xattn_bwd_impl(qf, mem, o, lse, d_o, dqf, dmem, accumulate, ws, B, rows, 256, N, scale, stream);
}

extern "C" int idiff_smm_xattn_cm_bwd(const float* qf, const float* mem, const float* o, const float* lse, const float* d_o, float* dqf, float* dmem,
                                      int accumulate, float* ws, int B, int rows, int Cm, int N, float scale, idiff_stream_t stream) {
    return smm_xattn_bwd_impl(qf, mem, o, lse, d_o, dqf, dmem, accumulate, ws, B, rows, Cm, N, scale, stream);
}
