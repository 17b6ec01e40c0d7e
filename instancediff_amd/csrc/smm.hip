// ScoreMapModule memory projection, fused:  mem = LayerNorm_256( Linear_{C->256}( LayerNorm_C(feature tokens) ) )
// (ContextDecoder.memory_proj, models/_modified_BiomedCLIP.py:1205-1209) on a channel-major feature map.
//
// One workgroup = 64 pixels x all 256 output channels.  The [C][64] input tile is loaded once into LDS
// (coalesced 256-byte rows), normalised in place (per-pixel two-pass statistics, lanes = pixels), multiplied by
// the packed weight [C][256] on v_mfma_f32_32x32x2_f32 (wave w owns output channels 64w..64w+63, A operand
// streamed from L2 with unit stride along co, B operand from LDS with unit stride along pixels), then the
// second LayerNorm runs on the accumulators (per-pixel sums across the 4 waves through LDS).
// HBM traffic = read C*4 B + write 1 KB per pixel (the unfused chain made 3 extra passes over the 1 KB/pixel map).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MP_W = 256;   // decoder width
constexpr int MP_PX = 64;   // pixels per workgroup

__global__ __launch_bounds__(256) void smm_memproj_kernel(const float* __restrict__ feat, long long fbs, const float* __restrict__ g1,
                                                          const float* __restrict__ b1, const float* __restrict__ wpk,
                                                          const float* __restrict__ bias, const float* __restrict__ g2,
                                                          const float* __restrict__ b2, float* __restrict__ out, int C, int N, float eps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                 // [C][64]
    float* part = smem + C * MP_PX;   // [4][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * MP_PX;
    const float* fb = feat + (long long)b * fbs;

    // ---- load the [C][64] tile ------------------------------------------------------------------
    const int nf4 = C * (MP_PX / 4);
    for (int f = tid; f < nf4; f += 256) {
        const int c = f >> 4, j4 = (f & 15) * 4;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (p0 + j4 + 3 < N)
            v = *reinterpret_cast<const floatx4*>(fb + (long long)c * N + p0 + j4);
        else
            for (int e = 0; e < 4; ++e)
                if (p0 + j4 + e < N) v[e] = fb[(long long)c * N + p0 + j4 + e];
        *reinterpret_cast<floatx4*>(xt + c * MP_PX + j4) = v;
    }
    __syncthreads();
    // ---- LayerNorm over C per pixel: wave q covers channels q, q+4, ... ; lane = pixel -----------
    float s = 0.f;
    for (int c = wave; c < C; c += 4) s += xt[c * MP_PX + lane];
    part[wave * MP_PX + lane] = s;
    __syncthreads();
    const float mean1 = (part[lane] + part[MP_PX + lane] + part[2 * MP_PX + lane] + part[3 * MP_PX + lane]) / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = wave; c < C; c += 4) {
        const float d = xt[c * MP_PX + lane] - mean1;
        q += d * d;
    }
    part[wave * MP_PX + lane] = q;
    __syncthreads();
    const float rstd1 = rsqrtf((part[lane] + part[MP_PX + lane] + part[2 * MP_PX + lane] + part[3 * MP_PX + lane]) / (float)C + eps);
    for (int c = wave; c < C; c += 4) xt[c * MP_PX + lane] = (xt[c * MP_PX + lane] - mean1) * rstd1 * g1[c] + b1[c];
    __syncthreads();

    // ---- GEMM: out[co][p] = sum_c W[c][co] * xhat[c][p] -----------------------------------------------
    floatx16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const float* wl = wpk + (long long)half * MP_W + wave * 64 + l31;
    const float* xl = xt + half * MP_PX + l31;
    const int nsteps = C / 2;
    // The weight operand comes straight from L2, one 4-byte element per lane and MFMA row block: a block of MP_PF k-steps is requested
    // while the previous block's 4 * MP_PF MFMAs run (r05: with one step in flight the loop was one L2 round trip per step -- 81 us for
    // 256 -> 256 at 32 x 32, batch 16, against 14 us of matrix time).  Same k order, same sums.
    constexpr int MP_PF = 8;
    auto wload = [&](float (&w)[MP_PF][2], int st0) {
#pragma unroll
        for (int u = 0; u < MP_PF; ++u) {
            w[u][0] = wl[(long long)(2 * (st0 + u)) * MP_W];
            w[u][1] = wl[(long long)(2 * (st0 + u)) * MP_W + 32];
        }
    };
    auto step = [&](int st, float a0, float a1) {
        const float x0 = xl[2 * st * MP_PX];
        const float x1 = xl[2 * st * MP_PX + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x1, acc[1][1], 0, 0, 0);
    };
    const int nfull = nsteps / MP_PF * MP_PF;  // whole blocks; the remainder (C not a multiple of 16) one step at a time
    if (nfull > 0) {
        float wa[MP_PF][2];
        wload(wa, 0);
        for (int st0 = 0; st0 < nfull; st0 += MP_PF) {
            float wn[MP_PF][2];
            // (the last block requests the matrix's first rows again: in range, unused)
            wload(wn, st0 + MP_PF < nfull ? st0 + MP_PF : 0);
            __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the requests behind the block's MFMAs)
#pragma unroll
            for (int u = 0; u < MP_PF; ++u) step(st0 + u, wa[u][0], wa[u][1]);
#pragma unroll
            for (int u = 0; u < MP_PF; ++u) wa[u][0] = wn[u][0], wa[u][1] = wn[u][1];
        }
    }
    for (int st = nfull; st < nsteps; ++st) step(st, wl[(long long)(2 * st) * MP_W], wl[(long long)(2 * st) * MP_W + 32]);
    // ---- + bias, LayerNorm over the 256 output channels per pixel ------------------------------------
    float bv[2][16], gv[2][16], ov[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bv[m][r] = bias[co];
            gv[m][r] = g2[co];
            ov[m][r] = b2[co];
        }
    float ps[2] = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] += bv[m][r];
                ps[n] += acc[m][n][r];
            }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        ps[n] += __shfl_xor(ps[n], 32, 64);
        if (half == 0) part[wave * MP_PX + n * 32 + l31] = ps[n];
    }
    __syncthreads();
    float mean2[2], rstd2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = n * 32 + l31;
        mean2[n] = (part[p] + part[MP_PX + p] + part[2 * MP_PX + p] + part[3 * MP_PX + p]) / (float)MP_W;
    }
    __syncthreads();
    float pq[2] = {0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[m][n][r] - mean2[n];
                pq[n] += d * d;
            }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        pq[n] += __shfl_xor(pq[n], 32, 64);
        if (half == 0) part[wave * MP_PX + n * 32 + l31] = pq[n];
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = n * 32 + l31;
        rstd2[n] = rsqrtf((part[p] + part[MP_PX + p] + part[2 * MP_PX + p] + part[3 * MP_PX + p]) / (float)MP_W + eps);
    }
    float* ob = out + (long long)b * MP_W * N;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = p0 + n * 32 + l31;
        if (p < N) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ob[(long long)co * N + p] = (acc[m][n][r] - mean2[n]) * rstd2[n] * gv[m][r] + ov[m][r];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Compact memory for the folded cross-attention (idiff_smm_memproj_compact_fwd):  out = [ xhat * rstd ; rstd ; 0.. ]
// with xhat = LayerNorm_C(feature token) and rstd = 1/sqrt(var_256(W xhat + b) + eps).  The 256-wide projection is only
// needed for that variance, and   var = |Wc xhat + bc|^2 / 256 = xhat^T G xhat + 2 h.xhat + e   with the centred
// Wc, bc and G = Wc^T Wc / 256 (C x C), h = Wc^T bc / 256, e = |bc|^2 / 256 prepared on the host (fp64): a C -> C product
// on the matrix cores instead of C -> 256 (4x fewer flops at C = 64), which leaves the kernel bandwidth-bound.
// One workgroup = 64 pixels; 32 x 32 output tiles of y = G xhat are dealt to the 4 waves; sum_c' xhat[c'] (y[c'] + 2 h[c'])
// is reduced over a lane's 16 rows, the two half-waves and the tiles through LDS.
// r05 form: the pixel's channels of a wave live in REGISTERS (thread = pixel x channel quarter: channels wave, wave + 4, ..): the tile
// is requested with all its loads in flight at the top of the kernel and never staged raw -- LDS only carries the 4-way partial sums
// of the two LayerNorm passes, the normalised tile (the B operand of the quadratic form) and the per-pixel rstd; the output rows
// leave from the same registers.  The Gram operand of a wave's first 32 x 32 tile is requested before the tile itself (it depends on
// nothing).  Same operations in the same order per pixel as the r01-r04 form (LDS-staged tile, three walks over it): the same bits.
template <int CJ>  // channels per thread = C / 4 (16: C = 64, 32: C = 128)
__device__ __forceinline__ void smm_memproj_gram_body(const float* __restrict__ feat, long long fbs, const float* __restrict__ g1,
                                                      const float* __restrict__ b1, const float* __restrict__ gram,
                                                      const float* __restrict__ hvec, float evar, float* __restrict__ out, int N, int Cm,
                                                      float eps1, float eps2, const int b, const int bx, const float* __restrict__ evar_dev = nullptr) {
    constexpr int C = 4 * CJ;
    if (evar_dev) evar = *evar_dev;  // training: the constant of the quadratic form is a device-side function of the weights
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                 // [C][64] normalised tile
    float* part = smem + C * MP_PX;   // [5][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int p0 = bx * MP_PX;
    const int p = p0 + lane;
    const bool ok = p < N;
    const float* fb = feat + (long long)b * fbs + (ok ? p : 0);
    // the wave's first tile of y = G xhat: (m, n) = (wave % mt, wave / mt); its A operand (Gram rows) travels in registers
    constexpr int mt = C / 32, ntiles = mt * 2;
    constexpr bool PRE = CJ == 16;  // (C = 128: 64 more registers would halve the occupancy of a bandwidth-bound kernel)
    float ga[PRE ? C / 2 : 1];
    if (PRE) {
        const int m = wave % mt;
        const float* wl = gram + (long long)half * C + m * 32 + l31;
#pragma unroll
        for (int st = 0; st < C / 2; ++st) ga[st] = wl[(long long)(2 * st) * C];
    }
    float x[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) x[j] = ok ? fb[(long long)(wave + 4 * j) * N] : 0.f;
    // ---- LayerNorm over C per pixel (two-pass; wave q covers channels q, q+4, ...; lane = pixel) ------------------
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CJ; ++j) s += x[j];
    part[wave * MP_PX + lane] = s;
    __syncthreads();
    const float mean1 = (part[lane] + part[MP_PX + lane] + part[2 * MP_PX + lane] + part[3 * MP_PX + lane]) / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const float d = x[j] - mean1;
        q += d * d;
    }
    part[wave * MP_PX + lane] = q;
    __syncthreads();
    const float rstd1 = rsqrtf((part[lane] + part[MP_PX + lane] + part[2 * MP_PX + lane] + part[3 * MP_PX + lane]) / (float)C + eps1);
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const int c = wave + 4 * j;
        x[j] = (x[j] - mean1) * rstd1 * g1[c] + b1[c];
        xt[c * MP_PX + lane] = x[j];
    }
    __syncthreads();  // (also: every wave has read the second-pass partials before they are overwritten below)
    // ---- quadratic form: tiles (m: 32 rows c', n: 32 pixels) of y = G xhat, folded with xhat on the spot -----------
    float pv[2] = {0.f, 0.f};  // per pixel block n
    for (int t = wave; t < ntiles; t += 4) {
        const int m = t % mt, n = t / mt;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* xl = xt + half * MP_PX + n * 32 + l31;
        if (PRE && t == wave) {
#pragma unroll
            for (int st = 0; st < C / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[PRE ? st : 0], xl[2 * st * MP_PX], acc, 0, 0, 0);
        } else {
            const float* wl = gram + (long long)half * C + m * 32 + l31;
#pragma unroll 8
            for (int st = 0; st < C / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(long long)(2 * st) * C], xl[2 * st * MP_PX], acc, 0, 0, 0);
        }
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cp = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            sum += xt[cp * MP_PX + n * 32 + l31] * (acc[r] + 2.f * hvec[cp]);
        }
        pv[n & 1] += sum;  // n is 0 or 1
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        pv[n] += __shfl_xor(pv[n], 32, 64);
        if (half == 0) part[wave * MP_PX + n * 32 + l31] = pv[n];
    }
    __syncthreads();
    const float r2 = rsqrtf(part[lane] + part[MP_PX + lane] + part[2 * MP_PX + lane] + part[3 * MP_PX + lane] + evar + eps2);
    if (ok) {
        float* ob = out + (long long)b * Cm * N + p;
#pragma unroll
        for (int j = 0; j < CJ; ++j) ob[(long long)(wave + 4 * j) * N] = x[j] * r2;
        for (int c = C + wave; c < Cm; c += 4) ob[(long long)c * N] = c == C ? r2 : 0.f;
    }
}
template <int CJ>
__global__ __launch_bounds__(256) void smm_memproj_gram_kernel(const float* __restrict__ feat, long long fbs, const float* __restrict__ g1,
                                                               const float* __restrict__ b1, const float* __restrict__ gram,
                                                               const float* __restrict__ hvec, float evar, float* __restrict__ out,
                                                               int N, int Cm, float eps1, float eps2, const float* __restrict__ evar_dev) {
    smm_memproj_gram_body<CJ>(feat, fbs, g1, b1, gram, hvec, evar, out, N, Cm, eps1, eps2, blockIdx.y, blockIdx.x, evar_dev);
}

// ---------------------------------------------------------------------------------------------------
// Backward of the compact memory projection (training step, 64-channel levels; r05).  Forward, per pixel:
//   xn = (x - mean_c x) rstd1 ;  xh = g1 xn + b1 ;  u = G xh ;  v = xh.u + 2 h.xh + e ;  r = (v + eps2)^-1/2 ;  m = [xh r ; r]
// Given dm [B, Cm, N] (rows 0..C: the padding rows carry no gradient):
//   dr = sum_c dm_c xh_c + dm_C ;  dv = -r^3 dr / 2 ;  dxh = dm_{0..C-1} r + 2 dv (u + h)          (G symmetric)
//   dG += dv xh xh^T ;  dh += 2 dv xh ;  de += dv ;  dg1 += dxh xn ;  db1 += dxh                       (sums over all pixels)
//   dx = rstd1 (g1 dxh - mean_c(g1 dxh) - xn mean_c(g1 dxh xn))
// Thread = pixel x channel quarter as in the forward (channels wave, wave + 4, ..: x, dm, xn, xh in registers); u = G xh and
// dG = (dv xh) xh^T (pixels as K) on v_mfma_f32_32x32x2_f32, one 32 x 32 tile per wave each; the per-pixel channel sums meet
// through LDS.  A workgroup walks tiles blockIdx.x, + gridDim.x, .. with its parameter-gradient accumulators in registers and leaves
// ONE partial row [C*C | dg1 | db1 | dh | de]; idiff_colsum reduces the rows in a fixed order (no atomics: bitwise reproducible).
constexpr int MB_LD = 65;  // row stride of the [c][pixel] LDS tiles: odd, so that lane = channel reads are conflict-free too
__global__ __launch_bounds__(256) void smm_memproj_gram_bwd_kernel(const float* __restrict__ feat, long long fbs, const float* __restrict__ g1,
                                                                   const float* __restrict__ b1, const float* __restrict__ gram,
                                                                   const float* __restrict__ hvec, const float* __restrict__ evar_dev,
                                                                   const float* __restrict__ dm, float* __restrict__ dfeat, long long dbs,
                                                                   float* __restrict__ ws, int N, int Cm, int ntx, int ntiles, float eps1,
                                                                   float eps2) {
    constexpr int CJ = 16, C = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                    // [C][65] xh
    float* ut = xt + C * MB_LD;          // [C][65] u + h
    float* wt = ut + C * MB_LD;          // [C][65] dv xh
    float* part = wt + C * MB_LD;        // [4 sums][4 waves][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const float evar = *evar_dev;
    float gj[CJ], bj[CJ], pg1[CJ], pb1[CJ], ph[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) gj[j] = g1[wave + 4 * j], bj[j] = b1[wave + 4 * j], pg1[j] = 0.f, pb1[j] = 0.f, ph[j] = 0.f;
    float pe = 0.f;
    floatx16 accG;
#pragma unroll
    for (int r = 0; r < 16; ++r) accG[r] = 0.f;
    const int mt = wave & 1, nt = wave >> 1;  // this wave's 32 x 32 tile of u (rows c', pixel block) and of dG (rows c, columns c')
    float ga[C / 2];                          // Gram rows of the u tile (constant over the tiles)
    {
        const float* wl = gram + (long long)half * C + mt * 32 + l31;
#pragma unroll
        for (int st = 0; st < C / 2; ++st) ga[st] = wl[(long long)(2 * st) * C];
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / ntx, p0 = (t - b * ntx) * MP_PX;
        const int p = p0 + lane;
        const bool ok = p < N;
        const float* fb = feat + (long long)b * fbs + (ok ? p : 0);
        const float* db = dm + (long long)b * Cm * N + (ok ? p : 0);
        float x[CJ], d[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) x[j] = ok ? fb[(long long)(wave + 4 * j) * N] : 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) d[j] = ok ? db[(long long)(wave + 4 * j) * N] : 0.f;
        const float dC = ok ? db[(long long)C * N] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) s += x[j];
        part[wave * 64 + lane] = s;
        __syncthreads();
        const float mean1 = (part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane]) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const float e = x[j] - mean1;
            q += e * e;
        }
        part[256 + wave * 64 + lane] = q;
        __syncthreads();
        const float rstd1 = rsqrtf((part[256 + lane] + part[256 + 64 + lane] + part[256 + 128 + lane] + part[256 + 192 + lane]) / (float)C + eps1);
        float xh[CJ];  // x[] becomes xn
        float drp = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            x[j] = (x[j] - mean1) * rstd1;
            xh[j] = x[j] * gj[j] + bj[j];
            xt[(wave + 4 * j) * MB_LD + lane] = xh[j];
            drp += d[j] * xh[j];
        }
        part[512 + wave * 64 + lane] = drp;
        __syncthreads();
        // ---- u = G xh for (rows mt*32.., pixels nt*32..); v partial folded as in the forward; u + h parked for the per-pixel pass ----
        {
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* xl = xt + half * MB_LD + nt * 32 + l31;
#pragma unroll
            for (int st = 0; st < C / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[st], xl[2 * st * MB_LD], acc, 0, 0, 0);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cp = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float hv = hvec[cp];
                sum += xt[cp * MB_LD + nt * 32 + l31] * (acc[r] + 2.f * hv);
                ut[cp * MB_LD + nt * 32 + l31] = acc[r] + hv;
            }
            sum += __shfl_xor(sum, 32, 64);
            // pixel nt*32 + l31 gets this wave's share (its 32 rows c'); the other pixel block of this wave's slot stays zero
            if (half == 0) {
                part[wave * 64 + nt * 32 + l31] = sum;
                part[wave * 64 + (nt ^ 1) * 32 + l31] = 0.f;
            }
        }
        __syncthreads();
        const float r2 = rsqrtf(part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane] + evar + eps2);
        const float dr = (part[512 + lane] + part[512 + 64 + lane] + part[512 + 128 + lane] + part[512 + 192 + lane]) + dC;
        const float dv = -0.5f * r2 * r2 * r2 * dr;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = wave + 4 * j;
            const float dxh = d[j] * r2 + 2.f * dv * ut[c * MB_LD + lane];
            wt[c * MB_LD + lane] = dv * xh[j];
            pg1[j] += dxh * x[j];
            pb1[j] += dxh;
            ph[j] += 2.f * dv * xh[j];
            d[j] = dxh * gj[j];  // d[] becomes d xn
            a1 += d[j];
            a2 += d[j] * x[j];
        }
        if (wave == 0) pe += dv;
        part[256 + wave * 64 + lane] = a1;
        part[768 + wave * 64 + lane] = a2;
        __syncthreads();
        const float m1 = (part[256 + lane] + part[256 + 64 + lane] + part[256 + 128 + lane] + part[256 + 192 + lane]) / (float)C;
        const float m2 = (part[768 + lane] + part[768 + 64 + lane] + part[768 + 128 + lane] + part[768 + 192 + lane]) / (float)C;
        if (ok) {
            float* ob = dfeat + (long long)b * dbs + p;
#pragma unroll
            for (int j = 0; j < CJ; ++j) ob[(long long)(wave + 4 * j) * N] = rstd1 * (d[j] - m1 - x[j] * m2);
        }
        // ---- dG[c][c'] += sum_px (dv xh)[c][px] xh[c'][px]: rows mt*32.., columns nt*32.. ----
        {
            const float* al = wt + (mt * 32 + l31) * MB_LD + half;
            const float* bl = xt + (nt * 32 + l31) * MB_LD + half;
#pragma unroll
            for (int st = 0; st < MP_PX / 2; ++st) accG = __builtin_amdgcn_mfma_f32_32x32x2f32(al[2 * st], bl[2 * st], accG, 0, 0, 0);
        }
        __syncthreads();  // the tiles and the partial sums are rewritten by the next tile
    }
    // ---- this workgroup's partial row ----
    float* wr = ws + (long long)blockIdx.x * (C * C + 3 * C + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * C + nt * 32 + l31] = accG[r];
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const float a = wave_sum(pg1[j]), bb = wave_sum(pb1[j]), h2 = wave_sum(ph[j]);
        if (lane == 0) {
            wr[C * C + wave + 4 * j] = a;
            wr[C * C + C + wave + 4 * j] = bb;
            wr[C * C + 2 * C + wave + 4 * j] = h2;
        }
    }
    if (wave == 0) {
        const float e = wave_sum(pe);
        if (lane == 0) wr[C * C + 3 * C] = e;
    }
}
// Grouped launch (idiff_smm_memproj_compact_grouped_fwd): the compact memories of several ScoreMapModules in ONE launch; blockIdx.z
// picks the level, blocks beyond a smaller level's pixels exit.  Same body, same bits.  The LDS of the launch is that of the widest
// level, so callers group levels of equal channel count (the two 64-channel levels of the UNet).
struct MemprojGroups {
    idiff_memproj_group g[IDIFF_MEMPROJ_MAX_GROUPS];
};
template <int CJ>
__global__ __launch_bounds__(256) void smm_memproj_gram_grouped_kernel(const MemprojGroups args, float eps1, float eps2) {
    const idiff_memproj_group& d = args.g[blockIdx.z];
    if ((int)blockIdx.x * MP_PX >= d.N) return;  // uniform
    smm_memproj_gram_body<CJ>(d.feat, d.feat_bstride, d.ln1_g, d.ln1_b, d.gram, d.hvec, d.evar, d.out, d.N, d.Cm, eps1, eps2, blockIdx.y, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// Token-side Linear with a pre-transposed weight wT [K][N]: lane = output feature (unit-stride weight reads, no
// cross-lane reduction), 8 rows per workgroup, the 4 waves split K and combine through LDS.
// out[r,n] = res[r,n] + gscale[n] * (sum_k act_in(x[r,k]) * wT[k,n] + bias[n]), then act_out.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply_t(float v, int act) {
    if (act == IDIFF_ACT_SILU) return silu_f(v);
    if (act == IDIFF_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

constexpr int LT_ROWS = 8;
__global__ __launch_bounds__(256) void linear_t_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ wT, long long ldw,
                                                       const float* __restrict__ bias, const float* __restrict__ res, long long ldr,
                                                       const float* __restrict__ gscale, float* __restrict__ out, long long ldo, int R, int K,
                                                       int N, int act_in, int act_out, long long x_hs, long long w_hs, long long b_hs,
                                                       long long o_hs, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                       float ln_eps) {
    // blockIdx.z = head of a per-head batch (idiff_linear_t_heads_fwd): operands advance by their head strides
    x += blockIdx.z * x_hs;
    wT += blockIdx.z * w_hs;
    if (bias) bias += blockIdx.z * b_hs;
    out += blockIdx.z * o_hs;
    extern __shared__ __attribute__((aligned(16))) float lt_smem[];
    float* xs = lt_smem;                                             // [LT_ROWS][K] activated input rows
    float(*red)[LT_ROWS][64] = reinterpret_cast<float(*)[LT_ROWS][64]>(lt_smem + LT_ROWS * K);  // [4][8][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * LT_ROWS;
    for (int i = threadIdx.x; i < LT_ROWS * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const int rr = r0 + r < R ? r0 + r : R - 1;
        xs[i] = act_apply_t(x[(long long)rr * ldx + k], act_in);
    }
    __syncthreads();
    if (ln_g) {
        // fused LayerNorm of the staged rows (idiff_linear_t_ln_fwd): one 32-lane half-wave per row, two-pass statistics
        const int row = threadIdx.x >> 5, l = threadIdx.x & 31;
        float* xr = xs + row * K;
        float sum = 0.f;
        for (int k = l; k < K; k += 32) sum += xr[k];
        const float mean = half_sum(sum) / (float)K;
        float sq = 0.f;
        for (int k = l; k < K; k += 32) {
            const float d = xr[k] - mean;
            sq += d * d;
        }
        const float rstd = rsqrtf(half_sum(sq) / (float)K + ln_eps);
        for (int k = l; k < K; k += 32) xr[k] = (xr[k] - mean) * rstd * ln_g[k] + ln_b[k];
        __syncthreads();
    }
    const int kq = (K + 3) / 4;
    const int k0 = wave * kq, k1 = min(K, k0 + kq);
    float acc[LT_ROWS];
#pragma unroll
    for (int r = 0; r < LT_ROWS; ++r) acc[r] = 0.f;
    const int nn = n < N ? n : N - 1;
    const float* wp = wT + nn;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) {
        const float wv = wp[(long long)k * ldw];
#pragma unroll
        for (int r = 0; r < LT_ROWS; ++r) acc[r] += xs[r * K + k] * wv;  // LDS broadcast read
    }
#pragma unroll
    for (int r = 0; r < LT_ROWS; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // 8 rows x 64 features = 512 outputs, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = threadIdx.x + i * 256;
        const int r = o >> 6, l = o & 63;
        const int nn2 = blockIdx.x * 64 + l;
        if (r0 + r < R && nn2 < N) {
            float v = red[0][r][l] + red[1][r][l] + red[2][r][l] + red[3][r][l];
            v = (gscale ? gscale[nn2] : 1.f) * (v + (bias ? bias[nn2] : 0.f));
            if (res) v += res[(long long)(r0 + r) * ldr + nn2];
            out[(long long)(r0 + r) * ldo + nn2] = act_apply_t(v, act_out);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// The same contract on the f32 matrix cores, for the aligned shapes the token chains actually use (wT 16-byte aligned rows,
// N % 4 == 0): the launch above is latency-bound -- each wave walks its K range with one 256-byte weight row in flight per
// step -- so this one keeps many 16-byte weight loads in flight per lane and lets v_mfma_f32_16x16x4_f32 do the k-group sums.
//   workgroup = 16 rows x 64 output features, 8 waves split K; per step of 4 k: lane (n = lane & 15, kk = lane >> 4) loads
//   the float4 wT[k0+kk][n0 + 4n .. +3] (a wave reads 4 whole 256-byte row segments) and the scalar xs[n][k0+kk] from LDS;
//   MFMA i of 4 multiplies the activations by element i of the float4 -> its column n is output feature n0 + 4n + i.
//   Partials of the 8 waves meet in LDS; bias / gscale / residual / activation as above.  Rows are independent of each other
//   (each output element is one fixed-order chain), so a token row's result does not depend on the batch around it.
// ---------------------------------------------------------------------------------------------------
constexpr int LM_ROWS = 16, LM_WAVES = 8, LM_KPAD = 4;
// body shared by the single launch (blockIdx.z = head of a per-head batch) and the grouped launch (blockIdx.z = group): bx / by are
// the 64-feature block and the 16-row block of this workgroup
__device__ __forceinline__ void linear_t_mfma_body(const float* __restrict__ x, long long ldx, const float* __restrict__ wT, long long ldw,
                                                   const float* __restrict__ bias, const float* __restrict__ res, long long ldr,
                                                   const float* __restrict__ gscale, float* __restrict__ out, long long ldo, int R, int K, int N,
                                                   int act_in, int act_out, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                   float ln_eps, int bx, int by) {
    extern __shared__ __attribute__((aligned(16))) float lm_smem[];
    const int KS = K + LM_KPAD;                         // padded row stride: rows land 4 banks apart
    float* xs = lm_smem;                                // [16][KS] activated (and normalised) input rows
    float* red = lm_smem + LM_ROWS * KS;                // [8 waves][4 mfma][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = by * LM_ROWS, n0 = bx * 64;
    // this wave's K range, in whole steps of 4
    const int nsteps = (K + 3) / 4;
    const int sper = (nsteps + LM_WAVES - 1) / LM_WAVES;
    const int s0 = wave * sper, s1 = min(nsteps, s0 + sper);
    const int n = lane & 15, kk = lane >> 4;
    const int col = n0 + 4 * n;
    const bool col_ok = col < N;  // N % 4 == 0: a lane's four features are in or out together
    const float* wp = wT + (col_ok ? col : 0);
    constexpr int UN = 8;  // steps in flight: 8 x 16 B per lane
    // The weights do not depend on the rows: the wave's first block (all of its K range at K <= 256) is requested BEFORE the rows are
    // staged and normalised, so the two round trips of the launch overlap (r05: 42 launches of 11.7 us per sampling step, all latency).
    floatx4 w[UN];
    auto wload = [&](int sb) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (sb + u) * 4 + kk;
            const bool ok = sb + u < s1 && k < K;
            w[u] = (ok && col_ok) ? *reinterpret_cast<const floatx4*>(wp + (long long)k * ldw) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    wload(s0);
    for (int i = tid; i < LM_ROWS * K; i += 512) {
        const int r = i / K, k = i - r * K;
        const int rr = r0 + r < R ? r0 + r : R - 1;
        xs[r * KS + k] = act_apply_t(x[(long long)rr * ldx + k], act_in);
    }
    __syncthreads();
    if (ln_g) {  // fused LayerNorm: one 32-lane half-wave per row, two-pass statistics (as linear_t_kernel)
        const int row = tid >> 5, l = tid & 31;
        float* xr = xs + row * KS;
        float sum = 0.f;
        for (int k = l; k < K; k += 32) sum += xr[k];
        const float mean = half_sum(sum) / (float)K;
        float sq = 0.f;
        for (int k = l; k < K; k += 32) {
            const float d = xr[k] - mean;
            sq += d * d;
        }
        const float rstd = rsqrtf(half_sum(sq) / (float)K + ln_eps);
        for (int k = l; k < K; k += 32) xr[k] = (xr[k] - mean) * rstd * ln_g[k] + ln_b[k];
        __syncthreads();
    }
    const float* xp = xs + n * KS + kk;
    floatx4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int sb = s0; sb < s1; sb += UN) {
        float a[UN];
        if (sb != s0) wload(sb);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (sb + u) * 4 + kk;
            const bool ok = sb + u < s1 && k < K;
            a[u] = ok ? xp[(sb + u) * 4] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w[u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w[u].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w[u].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w[u].w, acc[3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<floatx4*>(red)[(wave * 4 + i) * 64 + lane] = acc[i];
    __syncthreads();
    // C layout of 16x16x4: lane (n = lane & 15, g = lane >> 4) holds rows 4g .. 4g+3 of column n.  Thread t < 256 finishes
    // mfma i = t >> 6 of lane t & 63: 4 rows x 1 feature.
    if (tid < 256) {
        const int i = tid >> 6, l = tid & 63;
        floatx4 v = reinterpret_cast<const floatx4*>(red)[i * 64 + l];
#pragma unroll
        for (int w8 = 1; w8 < LM_WAVES; ++w8) {
            const floatx4 p = reinterpret_cast<const floatx4*>(red)[(w8 * 4 + i) * 64 + l];
            v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
        }
        const int feat = n0 + 4 * (l & 15) + i;
        if (feat < N) {
            const float gs = gscale ? gscale[feat] : 1.f, bs = bias ? bias[feat] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r0 + 4 * (l >> 4) + e;
                if (row < R) {
                    float y = gs * (v[e] + bs);
                    if (res) y += res[(long long)row * ldr + feat];
                    out[(long long)row * ldo + feat] = act_apply_t(y, act_out);
                }
            }
        }
    }
}

__global__ __launch_bounds__(512) void linear_t_mfma_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ wT, long long ldw,
                                                            const float* __restrict__ bias, const float* __restrict__ res, long long ldr,
                                                            const float* __restrict__ gscale, float* __restrict__ out, long long ldo, int R, int K,
                                                            int N, int act_in, int act_out, long long x_hs, long long w_hs, long long b_hs,
                                                            long long o_hs, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                            float ln_eps) {
    x += blockIdx.z * x_hs;
    wT += blockIdx.z * w_hs;
    if (bias) bias += blockIdx.z * b_hs;
    out += blockIdx.z * o_hs;
    linear_t_mfma_body(x, ldx, wT, ldw, bias, res, ldr, gscale, out, ldo, R, K, N, act_in, act_out, ln_g, ln_b, ln_eps, blockIdx.x, blockIdx.y);
}

// Grouped launch (idiff_linear_t_grouped_fwd): blockIdx.z picks one of up to IDIFF_LINEAR_MAX_GROUPS independent problems -- own
// operands, own shape -- from a descriptor array that travels in the kernel arguments (uniform scalar loads; no device-side table
// to keep alive or to update under HIP-graph capture).  The grid covers the largest problem; blocks outside a smaller one exit.
struct LinGroups {
    idiff_linear_group g[IDIFF_LINEAR_MAX_GROUPS];
};
__global__ __launch_bounds__(512) void linear_t_mfma_grouped_kernel(const LinGroups args) {
    const idiff_linear_group& d = args.g[blockIdx.z];
    if ((int)blockIdx.x * 64 >= d.N || (int)blockIdx.y * LM_ROWS >= d.R) return;  // uniform
    linear_t_mfma_body(d.x, d.ldx, d.wT, d.ldw, d.bias, d.res, d.ldr, d.gscale, d.out, d.ldo, d.R, d.K, d.N, d.act_in, d.act_out, d.ln_g, d.ln_b,
                       d.ln_eps, blockIdx.x, blockIdx.y);
}

}  // namespace

extern "C" int idiff_linear_t_grouped_fwd(const idiff_linear_group* groups, int ngroups, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(groups && ngroups >= 1 && ngroups <= IDIFF_LINEAR_MAX_GROUPS, "linear_t_grouped: 1..%d groups", IDIFF_LINEAR_MAX_GROUPS);
    LinGroups args;
    memset(&args, 0, sizeof(args));
    int gx = 0, gy = 0, kmax = 0;
    for (int i = 0; i < ngroups; ++i) {
        const idiff_linear_group& d = groups[i];
        IDIFF_CHECK_ARG(d.x && d.wT && d.out && d.R > 0 && d.K > 0 && d.N > 0, "linear_t_grouped: group %d: bad args", i);
        IDIFF_CHECK_ARG(d.ldx >= d.K && d.ldw >= d.N && d.ldo >= d.N && (!d.res || d.ldr >= d.N), "linear_t_grouped: group %d: bad leading dims", i);
        IDIFF_CHECK_ARG((d.ln_g == nullptr) == (d.ln_b == nullptr) && !(d.ln_g && d.act_in != IDIFF_ACT_NONE), "linear_t_grouped: group %d: LayerNorm", i);
        IDIFF_CHECK_ARG(d.N % 4 == 0 && d.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(d.wT) & 15) == 0,
                        "linear_t_grouped: group %d: the grouped launch is the matrix-core form (N %% 4 == 0, 16-byte aligned weight rows)", i);
        args.g[i] = d;
        gx = max(gx, (d.N + 63) / 64);
        gy = max(gy, (d.R + LM_ROWS - 1) / LM_ROWS);
        kmax = max(kmax, d.K);
    }
    const size_t lds = ((size_t)LM_ROWS * (kmax + LM_KPAD) + (size_t)LM_WAVES * 4 * 64 * 4) * sizeof(float);
    IDIFF_CHECK_ARG(lds <= 160 * 1024, "linear_t_grouped: K too large (%d)", kmax);
    static size_t attr = 0;
    if (lds > 64 * 1024 && lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_t_mfma_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "linear_t_grouped: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = lds;
    }
    hipLaunchKernelGGL(linear_t_mfma_grouped_kernel, dim3(gx, gy, ngroups), dim3(512), lds, (hipStream_t)stream, args);
    IDIFF_CHECK_LAUNCH("linear_t_grouped_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_smm_memproj_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* wpk,
                                     const float* bias, const float* ln2_g, const float* ln2_b, float* out, int B, int C, int N, float eps,
                                     idiff_stream_t stream) {
    IDIFF_CHECK_ARG(feat && ln1_g && ln1_b && wpk && bias && ln2_g && ln2_b && out, "smm_memproj: null pointer");
    IDIFF_CHECK_ARG(B > 0 && N > 0 && C >= 2 && C % 2 == 0 && C <= 512, "smm_memproj: C must be even and <= 512 (got %d)", C);
    IDIFF_CHECK_ARG(N % 4 == 0 && feat_bstride % 4 == 0, "smm_memproj: N and feat_bstride must be multiples of 4");
    const size_t lds = (size_t)(C * MP_PX + 4 * MP_PX + 2 * MP_PX) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smm_memproj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "smm_memproj: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = lds;
    }
    hipLaunchKernelGGL(smm_memproj_kernel, dim3((N + MP_PX - 1) / MP_PX, B), dim3(256), lds, (hipStream_t)stream, feat, (long long)feat_bstride,
                       ln1_g, ln1_b, wpk, bias, ln2_g, ln2_b, out, C, N, eps);
    IDIFF_CHECK_LAUNCH("smm_memproj_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_smm_memproj_compact_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* gram,
                                             const float* hvec, float evar, float* out, int B, int C, int N, int Cm, float eps1, float eps2,
                                             idiff_stream_t stream) {
    IDIFF_CHECK_ARG(feat && ln1_g && ln1_b && gram && hvec && out, "smm_memproj_compact: null pointer");
    IDIFF_CHECK_ARG(B > 0 && N > 0 && (C == 64 || C == 128), "smm_memproj_compact: C must be 64 or 128 (got %d)", C);
    IDIFF_CHECK_ARG(N % 4 == 0 && feat_bstride % 4 == 0, "smm_memproj_compact: N and feat_bstride must be multiples of 4");
    IDIFF_CHECK_ARG(Cm > C, "smm_memproj_compact: Cm must exceed C (got %d, C = %d)", Cm, C);
    const size_t lds = (size_t)(C * MP_PX + 5 * MP_PX) * sizeof(float);  // <= 34 KB: no attribute needed
    const dim3 grid((N + MP_PX - 1) / MP_PX, B);
    const float* nodev = nullptr;
    if (C == 64)
        hipLaunchKernelGGL(smm_memproj_gram_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, feat, (long long)feat_bstride, ln1_g, ln1_b, gram,
                           hvec, evar, out, N, Cm, eps1, eps2, nodev);
    else
        hipLaunchKernelGGL(smm_memproj_gram_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, feat, (long long)feat_bstride, ln1_g, ln1_b, gram,
                           hvec, evar, out, N, Cm, eps1, eps2, nodev);
    IDIFF_CHECK_LAUNCH("smm_memproj_compact_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_smm_memproj_compact_train_fwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* gram,
                                                   const float* hvec, const float* evar_dev, float* out, int B, int C, int N, int Cm, float eps1,
                                                   float eps2, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(feat && ln1_g && ln1_b && gram && hvec && evar_dev && out, "smm_memproj_compact_train_fwd: null pointer");
    IDIFF_CHECK_ARG(B > 0 && N > 0 && C == 64 && Cm > C, "smm_memproj_compact_train_fwd: C must be 64 (got %d), Cm > C", C);
    IDIFF_CHECK_ARG(N % 4 == 0 && feat_bstride % 4 == 0, "smm_memproj_compact_train_fwd: N and feat_bstride must be multiples of 4");
    const size_t lds = (size_t)(C * MP_PX + 5 * MP_PX) * sizeof(float);
    hipLaunchKernelGGL(smm_memproj_gram_kernel<16>, dim3((N + MP_PX - 1) / MP_PX, B), dim3(256), lds, (hipStream_t)stream, feat, (long long)feat_bstride,
                       ln1_g, ln1_b, gram, hvec, 0.f, out, N, Cm, eps1, eps2, evar_dev);
    IDIFF_CHECK_LAUNCH("smm_memproj_compact_train_fwd");
    return IDIFF_OK;
}

static int memproj_bwd_grid(int B, int N) {
    const long long tiles = (long long)B * ((N + MP_PX - 1) / MP_PX);
    return (int)(tiles < 1024 ? tiles : 1024);
}
extern "C" int64_t idiff_smm_memproj_compact_bwd_ws_floats(int B, int C, int N) {
    return (int64_t)memproj_bwd_grid(B, N) * (C * C + 3 * C + 1);
}
extern "C" int idiff_smm_memproj_compact_bwd(const float* feat, int64_t feat_bstride, const float* ln1_g, const float* ln1_b, const float* gram,
                                             const float* hvec, const float* evar_dev, const float* dm, float* dfeat, int64_t dfeat_bstride,
                                             float* dparams, float* ws, int B, int C, int N, int Cm, float eps1, float eps2,
                                             idiff_stream_t stream) {
    IDIFF_CHECK_ARG(feat && ln1_g && ln1_b && gram && hvec && evar_dev && dm && dfeat && dparams && ws, "smm_memproj_compact_bwd: null pointer");
    IDIFF_CHECK_ARG(B > 0 && N > 0 && C == 64 && Cm > C, "smm_memproj_compact_bwd: C must be 64 (got %d), Cm > C", C);
    const int ntx = (N + MP_PX - 1) / MP_PX, ntiles = B * ntx, grid = memproj_bwd_grid(B, N);
    const int PW = C * C + 3 * C + 1;
    const size_t lds = (size_t)(3 * C * MB_LD + 16 * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smm_memproj_gram_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "smm_memproj_compact_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL(smm_memproj_gram_bwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, feat, (long long)feat_bstride, ln1_g, ln1_b, gram,
                       hvec, evar_dev, dm, dfeat, (long long)dfeat_bstride, ws, N, Cm, ntx, ntiles, eps1, eps2);
    IDIFF_CHECK_LAUNCH("smm_memproj_compact_bwd");
    // dparams [C*C | C | C | C | 1] = column sums of the workgroups' partial rows, fixed order
    return idiff_colsum(ws, PW, dparams, grid, PW, 0, stream);
}

extern "C" int idiff_smm_memproj_compact_grouped_fwd(const idiff_memproj_group* groups, int ngroups, int B, float eps1, float eps2,
                                                     idiff_stream_t stream) {
    IDIFF_CHECK_ARG(groups && ngroups >= 1 && ngroups <= IDIFF_MEMPROJ_MAX_GROUPS, "smm_memproj_compact_grouped: 1..%d groups", IDIFF_MEMPROJ_MAX_GROUPS);
    MemprojGroups args;
    memset(&args, 0, sizeof(args));
    int gx = 0, cmax = 0;
    for (int i = 0; i < ngroups; ++i) {
        const idiff_memproj_group& d = groups[i];
        IDIFF_CHECK_ARG(d.feat && d.ln1_g && d.ln1_b && d.gram && d.hvec && d.out, "smm_memproj_compact_grouped: group %d: null pointer", i);
        IDIFF_CHECK_ARG(d.N > 0 && (d.C == 64 || d.C == 128) && d.N % 4 == 0 && d.feat_bstride % 4 == 0 && d.Cm > d.C,
                        "smm_memproj_compact_grouped: group %d: bad shape (C %d, N %d, Cm %d)", i, d.C, d.N, d.Cm);
        args.g[i] = d;
        gx = max(gx, (d.N + MP_PX - 1) / MP_PX);
        cmax = max(cmax, d.C);
    }
    for (int i = 0; i < ngroups; ++i) IDIFF_CHECK_ARG(groups[i].C == cmax, "smm_memproj_compact_grouped: the groups of a launch share their channel count (%d vs %d)", groups[i].C, cmax);
    const size_t lds = (size_t)(cmax * MP_PX + 5 * MP_PX) * sizeof(float);  // <= 34 KB: no attribute needed
    if (cmax == 64) hipLaunchKernelGGL(smm_memproj_gram_grouped_kernel<16>, dim3(gx, B, ngroups), dim3(256), lds, (hipStream_t)stream, args, eps1, eps2);
    else hipLaunchKernelGGL(smm_memproj_gram_grouped_kernel<32>, dim3(gx, B, ngroups), dim3(256), lds, (hipStream_t)stream, args, eps1, eps2);
    IDIFF_CHECK_LAUNCH("smm_memproj_compact_grouped_fwd");
    return IDIFF_OK;
}

static int linear_t_launch(const float* x, int64_t ldx, const float* wT, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                           const float* gscale, float* out, int64_t ldo, int R, int K, int N, int act_in, int act_out, int heads, int64_t x_hs,
                           int64_t w_hs, int64_t b_hs, int64_t o_hs, const float* ln_g, const float* ln_b, float ln_eps, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && wT && out && R > 0 && K > 0 && N > 0 && heads > 0, "linear_t_fwd: bad args");
    IDIFF_CHECK_ARG(ldx >= K && ldw >= N && ldo >= N, "linear_t_fwd: bad leading dims");
    IDIFF_CHECK_ARG(K <= 8192, "linear_t_fwd: K must be <= 8192 (got %d)", K);
    // matrix-core form for aligned weights (every token-chain shape of the ScoreMapModule); the vector-ALU form takes the rest
    static const bool mfma_off = [] {
        const char* e = getenv("IDIFF_LINEAR_MFMA");
        return e && e[0] == '0';
    }();
    const size_t lds_m = ((size_t)LM_ROWS * (K + LM_KPAD) + (size_t)LM_WAVES * 4 * 64 * 4) * sizeof(float);
    if (!mfma_off && N % 4 == 0 && ldw % 4 == 0 && w_hs % 4 == 0 && (reinterpret_cast<uintptr_t>(wT) & 15) == 0 && lds_m <= 160 * 1024) {
        dim3 gridm((N + 63) / 64, (R + LM_ROWS - 1) / LM_ROWS, heads);
        static size_t attr_m = 0;
        if (lds_m > 64 * 1024 && lds_m > attr_m) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_t_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "linear_t_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_m = lds_m;
        }
        hipLaunchKernelGGL(linear_t_mfma_kernel, gridm, dim3(512), lds_m, (hipStream_t)stream, x, (long long)ldx, wT, (long long)ldw, bias, res,
                           (long long)ldr, gscale, out, (long long)ldo, R, K, N, act_in, act_out, (long long)x_hs, (long long)w_hs,
                           (long long)b_hs, (long long)o_hs, ln_g, ln_b, ln_eps);
        IDIFF_CHECK_LAUNCH("linear_t_fwd(mfma)");
        return IDIFF_OK;
    }
    dim3 grid((N + 63) / 64, (R + LT_ROWS - 1) / LT_ROWS, heads);
    const size_t lds = ((size_t)LT_ROWS * K + 4 * LT_ROWS * 64) * sizeof(float);
    static size_t attr = 0;
    if (lds > 64 * 1024 && lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "linear_t_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = lds;
    }
    hipLaunchKernelGGL(linear_t_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, (long long)ldx, wT, (long long)ldw, bias, res,
                       (long long)ldr, gscale, out, (long long)ldo, R, K, N, act_in, act_out, (long long)x_hs, (long long)w_hs, (long long)b_hs,
                       (long long)o_hs, ln_g, ln_b, ln_eps);
    IDIFF_CHECK_LAUNCH("linear_t_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_linear_t_fwd(const float* x, int64_t ldx, const float* wT, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                                  const float* gscale, float* out, int64_t ldo, int R, int K, int N, int act_in, int act_out,
                                  idiff_stream_t stream) {
    return linear_t_launch(x, ldx, wT, ldw, bias, res, ldr, gscale, out, ldo, R, K, N, act_in, act_out, 1, 0, 0, 0, 0, nullptr, nullptr, 0.f, stream);
}

extern "C" int idiff_linear_t_ln_fwd(const float* x, int64_t ldx, const float* ln_g, const float* ln_b, float ln_eps, const float* wT, int64_t ldw,
                                     const float* bias, const float* res, int64_t ldr, const float* gscale, float* out, int64_t ldo, int R, int K,
                                     int N, int act_out, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(ln_g && ln_b, "linear_t_ln_fwd: null LayerNorm parameters");
    return linear_t_launch(x, ldx, wT, ldw, bias, res, ldr, gscale, out, ldo, R, K, N, IDIFF_ACT_NONE, act_out, 1, 0, 0, 0, 0, ln_g, ln_b, ln_eps, stream);
}

extern "C" int idiff_linear_t_heads_fwd(const float* x, int64_t ldx, int64_t x_hs, const float* wT, int64_t ldw, int64_t w_hs, const float* bias,
                                        int64_t b_hs, float* out, int64_t ldo, int64_t o_hs, int R, int K, int N, int heads,
                                        idiff_stream_t stream) {
    return linear_t_launch(x, ldx, wT, ldw, bias, nullptr, 0, nullptr, out, ldo, R, K, N, IDIFF_ACT_NONE, IDIFF_ACT_NONE, heads, x_hs, w_hs, b_hs,
                           o_hs, nullptr, nullptr, 0.f, stream);
}
