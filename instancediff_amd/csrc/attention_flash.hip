// The reference's half-precision attention form for the ScoreMapModule decoder (gfx950): `Attention_flash`,
// models/_modified_BiomedCLIP.py:481-517, selected by TransformerDecoderLayer_scaled(if_flash=True), :552-590 --
//     x = flash_attn_func(clamp(q, -255, 255).half(), clamp(k, ...).half(), clamp(v, ...).half(), softmax_scale=scale).float()
// i.e. the projected q / k / v are clamped and rounded to fp16, the scores and the softmax statistics are fp32, the (unnormalised)
// probabilities are rounded to fp16 for the second product, the products accumulate in fp32 and the normalised result is rounded to
// fp16.  A REDUCED-PRECISION VARIANT of this library (model option `score_map_if_flash`, inference only; the default path computes
// what `if_flash=False` computes, fp32 end to end): every rounding of that recipe is applied where the recipe applies it; the
// products of fp16 values are exact in fp32, so running them on the f32 matrix cores (v_mfma_f32_32x32x2_f32) accumulates the same
// terms flash-attn's fp16 MFMAs accumulate, in another order.
//
//  * idiff_attn_tokens_f16_fwd : the decoder's token self-attention (a handful of tokens): one wave per (b, head, query), lanes = keys.
//  * idiff_smm_xattn_kv_f16_fwd: the cross-attention of the few text queries over the N pixel keys with UNFOLDED keys and values
//    (with the k / v projections folded onto the queries, as the fp32 path does, there is no k / v tensor to clamp and round):
//    k, v [B, heads*64, N] channel-major.  Workgroup = 4 waves = one key split of one sample; wave w = head w (dh = 64 = the 64
//    channels a wave of the 256-row fp32 kernel owns, so a head's scores are complete inside its wave: no exchange, no barrier);
//    32-key blocks, S^T = K Q^T (keys on the accumulator rows), online softmax, O^T += V^T P with P taken from the S accumulator;
//    flash-decoding style split over the keys + a merge kernel that normalises and applies the final fp16 rounding.
#include <stdlib.h>

#include "common.h"

namespace {

#define KAPPA(r) (((r) & 3) + 8 * ((r) >> 2))

__device__ __forceinline__ float r16(float x) { return (float)(_Float16)x; }                       // round to nearest even fp16
__device__ __forceinline__ float c16(float x) { return r16(fminf(fmaxf(x, -255.f), 255.f)); }     // clamp to +-255, then round

__global__ __launch_bounds__(64) void attn_tokens_f16_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
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
        for (int d = 0; d < dh; ++d) acc = __builtin_fmaf(c16(qp[d]), c16(kp[d]), acc);
        s = acc * scale;
    }
    const float mx = wave_max(s);
    const float p = lane < M ? __expf(s - mx) : 0.f;
    const float l = wave_sum(p);  // the row sum is taken from the fp32 probabilities, the second product from their fp16 roundings
    const float ph = r16(p);
    float* op = out + ((long long)b * Nq + n) * C + h * dh;
    for (int d = 0; d < dh; ++d) {
        const float t = wave_sum(lane < M ? ph * c16(v[((long long)b * M + lane) * ldkv + h * dh + d]) : 0.f);
        if (lane == 0) op[d] = r16(t / l);
    }
}

constexpr int FX_DH = 64;               // head dim = channels per wave
constexpr int FX_TILE = FX_DH * 33;     // per-wave tile [64 c][33] (keys along the row, odd stride: conflict-free column reads)
constexpr int FX_ROWS = 8;              // query rows kept per head in the partials (Nq <= 8)

// ws per (b, split): [heads][FX_ROWS][FX_DH + 2]  (partial P.V, then the running maximum and the row sum)
__global__ __launch_bounds__(256) void smm_xattn_kv_f16_kernel(const float* __restrict__ q, const float* __restrict__ kmem, const float* __restrict__ vmem,
                                                               float* __restrict__ ws, int Nq, int N, int nsplit, int kps, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int heads = 4, C = heads * FX_DH;
    float* tk = smem + wave * 2 * FX_TILE;  // private per wave: the head's 64 key channels x 32 keys
    float* tv = tk + FX_TILE;               //                   the head's 64 value channels x 32 keys
    const float* kb = kmem + ((long long)b * C + wave * FX_DH) * N;
    const float* vb = vmem + ((long long)b * C + wave * FX_DH) * N;

    // B operand of S^T = K Q^T: lane (row = l31, k parity = half) holds q[row][2t + half] of its head, clamped and rounded
    float qreg[FX_DH / 2];
#pragma unroll
    for (int t = 0; t < FX_DH / 2; ++t) qreg[t] = l31 < Nq ? c16(q[((long long)b * Nq + l31) * C + wave * FX_DH + 2 * t + half]) : 0.f;

    floatx16 O[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    const int nkb = (N + 31) / 32;
    const int kb_begin = sp * kps, kb_end = min(nkb, kb_begin + kps);
    constexpr int NF4 = FX_DH * 8 / 64;  // float4 loads per lane, tile and 32-key block
    floatx4 rk[NF4], rv[NF4];
    auto load_tiles = [&](int kbi) {
        const int key0 = kbi * 32;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;  // float4 index in [64][8]
            const int c = f >> 3, j4 = (f & 7) * 4;
            floatx4 zk = {0.f, 0.f, 0.f, 0.f}, zv = zk;
            if (key0 + j4 + 3 < N) {
                zk = *reinterpret_cast<const floatx4*>(kb + (long long)c * N + key0 + j4);
                zv = *reinterpret_cast<const floatx4*>(vb + (long long)c * N + key0 + j4);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (key0 + j4 + e < N) zk[e] = kb[(long long)c * N + key0 + j4 + e], zv[e] = vb[(long long)c * N + key0 + j4 + e];
            }
            rk[i] = zk, rv[i] = zv;
        }
    };
    auto write_tiles = [&]() {  // the clamp and the fp16 rounding of k and v happen here, once per element
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = lane + i * 64;
            const int c = f >> 3, j4 = (f & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tk[c * 33 + j4 + e] = c16(rk[i][e]), tv[c * 33 + j4 + e] = c16(rv[i][e]);
        }
    };
    if (kb_begin < kb_end) {
        load_tiles(kb_begin);
        write_tiles();
    }
    for (int kbi = kb_begin; kbi < kb_end; ++kbi) {
        if (kbi + 1 < kb_end) load_tiles(kbi + 1);
        floatx16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int t = 0; t < FX_DH / 2; ++t) S = __builtin_amdgcn_mfma_f32_32x32x2f32(tk[(2 * t + half) * 33 + l31], qreg[t], S, 0, 0, 0);
        // C layout of 32x32x2: lane (column = query row l31, half) holds accumulator rows (= keys) KAPPA(r) + 4 * half
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
            ps += p;          // fp32 row sum
            S[r] = r16(p);    // fp16 operand of the second product
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[m][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                O[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(tv[(m * 32 + l31) * 33 + KAPPA(r) + 4 * half], S[r], O[m], 0, 0, 0);
        if (kbi + 1 < kb_end) write_tiles();  // (wave-private tiles: LDS is in order within a wave)
    }
    // partial out: O[m][r] = sum over this split's keys for channel m*32 + KAPPA(r) + 4*half of the head, query row l31
    float* wp = ws + (((long long)b * nsplit + sp) * heads + wave) * FX_ROWS * (FX_DH + 2);
    const float l = lrun + __shfl_xor(lrun, 32, 64);
    if (l31 < FX_ROWS) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) wp[l31 * (FX_DH + 2) + m * 32 + KAPPA(r) + 4 * half] = O[m][r];
        if (half == 0) {
            wp[l31 * (FX_DH + 2) + FX_DH] = mrun;
            wp[l31 * (FX_DH + 2) + FX_DH + 1] = l;
        }
    }
}

// out[b][n][h*64 + d] = fp16( sum_s w_s O_s / sum_s w_s l_s ), w_s = exp(m_s - max_s m_s); splits in order
__global__ __launch_bounds__(256) void smm_xattn_kv_f16_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int Nq, int nsplit) {
    const int b = blockIdx.x, n = blockIdx.y;
    const int h = threadIdx.x >> 6, d = threadIdx.x & 63;
    const int heads = 4;
    const long long ss = (long long)heads * FX_ROWS * (FX_DH + 2);
    const float* w0 = ws + (long long)b * nsplit * ss + ((long long)h * FX_ROWS + n) * (FX_DH + 2);
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, w0[s * ss + FX_DH]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = w0[s * ss + FX_DH];
        const float f = ms == -INFINITY ? 0.f : __expf(ms - M);
        L += w0[s * ss + FX_DH + 1] * f;
        acc += w0[s * ss + d] * f;
    }
    out[((long long)b * Nq + n) * (heads * FX_DH) + h * FX_DH + d] = r16(acc / L);
}

inline void fx_split(int N, int* nsplit, int* kps) {  // a function of N alone (a sample's bits do not depend on its batch): <= 32 splits
    const int nkb = (N + 31) / 32;
    int k = nkb / 32;
    if (k < 2) k = 2;
    if (k > nkb) k = nkb;
    *kps = k;
    *nsplit = (nkb + k - 1) / k;
}

}  // namespace

extern "C" int idiff_attn_tokens_f16_fwd(const float* q, const float* k, const float* v, float* out, int B, int Nq, int M, int C, int heads, float scale,
                                         int64_t ldq, int64_t ldkv, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && out && B > 0 && Nq > 0 && M > 0 && heads > 0 && C % heads == 0, "attn_tokens_f16: bad args");
    IDIFF_CHECK_ARG(M <= 64, "attn_tokens_f16: at most 64 keys (got %d)", M);
    hipLaunchKernelGGL(attn_tokens_f16_kernel, dim3(B * heads * Nq), dim3(64), 0, (hipStream_t)stream, q, k, v, out, Nq, M, C, heads, scale,
                       (long long)ldq, (long long)ldkv);
    IDIFF_CHECK_LAUNCH("attn_tokens_f16_fwd");
    return IDIFF_OK;
}

extern "C" int64_t idiff_smm_xattn_kv_f16_ws_floats(int B, int N) {
    int ns, kps;
    fx_split(N, &ns, &kps);
    return (int64_t)B * ns * 4 * FX_ROWS * (FX_DH + 2);
}

extern "C" int idiff_smm_xattn_kv_f16_fwd(const float* q, const float* k, const float* v, float* out, float* ws, int B, int Nq, int heads, int C, int N,
                                          float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(q && k && v && out && ws && B > 0 && N > 0, "smm_xattn_kv_f16: bad args");
    IDIFF_CHECK_ARG(heads == 4 && C == 4 * FX_DH, "smm_xattn_kv_f16: 4 heads x 64 channels (got %d heads, C = %d)", heads, C);
    IDIFF_CHECK_ARG(Nq >= 1 && Nq <= FX_ROWS, "smm_xattn_kv_f16: 1..%d query rows (got %d)", FX_ROWS, Nq);
    IDIFF_CHECK_ARG(N % 4 == 0, "smm_xattn_kv_f16: N must be a multiple of 4");
    int ns, kps;
    fx_split(N, &ns, &kps);
    const size_t lds = (size_t)4 * 2 * FX_TILE * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smm_xattn_kv_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "smm_xattn_kv_f16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smm_xattn_kv_f16_kernel, dim3(ns, B), dim3(256), lds, st, q, k, v, ws, Nq, N, ns, kps, scale);
    IDIFF_CHECK_LAUNCH("smm_xattn_kv_f16_fwd");
    hipLaunchKernelGGL(smm_xattn_kv_f16_merge_kernel, dim3(B, Nq), dim3(256), 0, st, ws, out, Nq, ns);
    IDIFF_CHECK_LAUNCH("smm_xattn_kv_f16_merge");
    return IDIFF_OK;
}
