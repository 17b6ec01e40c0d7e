// Weight gradient of the implicit-GEMM convolution on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   dW[co][ci][tap] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+ky-pad,x+kx-pad]
//   GEMM view: M = co (64 per workgroup = 4 waves x 16), N = (ci, tap) pairs of one CK-channel chunk
//   (CK*taps = 144 for 3x3 with CK=16: nine 16-wide N blocks, no padding waste), K = pixels.
//   A workgroup owns one (channel chunk, co block, split) and walks its share of the (sample, 128-pixel tile)
//   list with the accumulators resident in registers; X is gathered exactly like the forward kernel gathers it
//   (virtual concat, nearest x2 upsample, pixel-unshuffle, producer's GroupNorm/FiLM affine + SiLU, zero
//   padding), so the same fused forward needs no materialised im2col / activated tensor for its backward.
//   Results go to a per-split partial buffer (plain coalesced stores, co fastest); a second kernel reduces the
//   splits in a fixed order and writes torch layout [co][ci][ky][kx] -> bitwise reproducible (no atomics).
#include "conv_wgrad_args.h"

namespace {

struct WgArgs {
    const float* src0;
    const float* src1;
    long long bs0, bs1;
    int C0v, C1v, C0r, Cin;
    int B, Hin, Win, Hout, Wout;
    int Cout;
    const float* pro_a;
    const float* pro_b;
    const float* dy;
    long long dybs;
    float* ws;  // [nsplit][taps][Cin][Cout]
    int tiles_x, ntiles, ncob, nchunks, nsplit;
};

// THIN (Cout <= 16: the score-map embedding convs 5 -> 16 and the output conv 64 -> 5 of the training step): one 16-channel block is
// all there is, so the four waves split the K loop (the 128 pixels of a tile) instead of the output channels -- their partial sums
// meet in the result staging -- and only 16 rows of dY are staged.  With CK = 8 for Cin <= 8 (five 16-wide N blocks instead of nine)
// a 5 -> 16 layer issues 7x fewer MFMAs and 3.5x fewer loads than the general form, which spent them on zero rows.
template <int KS, int CK, int TWL, int MODE, bool THIN = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
    constexpr int TW = 1 << TWL;
    constexpr int TH = 128 / TW;
    constexpr int PAD = KS / 2;
    constexpr int TRH = TH + KS - 1;
    constexpr int RS = TW + KS - 1;
    constexpr int PS = TRH * RS;
    constexpr int TAPS = KS * KS;
    constexpr int NN = CK * TAPS;             // (ci, tap) pairs per chunk
    constexpr int NB = (NN + 15) / 16;        // 16-wide N blocks
    constexpr int XT = CK * PS + 4;           // + a zero slot
    constexpr int DYLD = 129;
    constexpr int NL = (CK * PS + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                 // [CK][TRH][RS] (+ zero slot at CK*PS)
    float* dyt = smem + XT;           // [64][129]
    float* protab = dyt + 64 * DYLD;  // [2][C0r] when pro

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int cc = blockIdx.x % a.nchunks;
    const int cob = (blockIdx.x / a.nchunks) % a.ncob;
    const int sp = blockIdx.x / (a.nchunks * a.ncob);
    const int cb = cc * CK, co0 = cob * 64;
    const int HWin = a.Hin * a.Win, HWo = a.Hout * a.Wout;
    const bool has_pro = a.pro_a != nullptr;

    // per-lane B-operand bases: n = nb*16 + l15 -> (ci, tap)
    int bbase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 16 + l15;
        const int ci = n / TAPS, tap = n - ci * TAPS;
        bbase[nb] = n < NN ? ci * PS + (tap / KS) * RS + (tap % KS) : -1;
    }
    floatx4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (tid < 4) xt[CK * PS + tid] = 0.f;

    const int total_tiles = a.B * a.ntiles;
    int cur_b = -1;
    for (int t = sp; t < total_tiles; t += a.nsplit) {
        const int b = t / a.ntiles, tile = t - b * a.ntiles;
        const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
        __syncthreads();  // previous tile's LDS reads are done
        if (has_pro && b != cur_b) {
            for (int i = tid; i < a.C0r; i += 256) {
                protab[i] = a.pro_a[(long long)b * a.C0r + i];
                protab[a.C0r + i] = a.pro_b[(long long)b * a.C0r + i];
            }
            cur_b = b;
            __syncthreads();
        }
        // ---- gather the X tile (same semantics as the forward kernel) --------------------------------
        const float* sample0 = a.src0 + (long long)b * a.bs0;
        const float* base0 = sample0 + (long long)(MODE == IDIFF_CONV_UNSHUFFLE2 ? (cb >> 2) : cb) * HWin;
        const float* base1 = a.src1 ? a.src1 + (long long)b * a.bs1 + (long long)(cb - a.C0v) * HWin : sample0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256;
            if (e < CK * PS) {
                const int ci = e / PS;
                const int rem = e - ci * PS;
                const int r = rem / RS, c = rem - r * RS;
                const int oy = y0 - PAD + r, ox = x0 - PAD + c;
                const int ch = cb + ci;
                float x = 0.f;
                if (oy >= 0 && oy < a.Hout && ox >= 0 && ox < a.Wout && ch < a.Cin) {
                    int off;
                    if (MODE == IDIFF_CONV_UPSAMPLE2)
                        off = ci * HWin + (oy >> 1) * a.Win + (ox >> 1);
                    else if (MODE == IDIFF_CONV_UNSHUFFLE2)
                        off = (ci >> 2) * HWin + (2 * oy + ((ci >> 1) & 1)) * a.Win + 2 * ox + (ci & 1);
                    else
                        off = ci * HWin + oy * a.Win + ox;
                    x = (ch < a.C0v ? base0 : base1)[off];
                    if (has_pro) {
                        const int chr = MODE == IDIFF_CONV_UNSHUFFLE2 ? (ch >> 2) : ch;
                        x = silu_fast(protab[chr] * x + protab[a.C0r + chr]);
                    }
                }
                xt[e] = x;
            }
        }
        // ---- dY tile [64 co][128 px] --------------------------------------------------------------------
        const float* dyb = a.dy + (long long)b * a.dybs;
        for (int e = tid; e < (THIN ? 16 : 64) * 128; e += 256) {
            const int co = e >> 7, p = e & 127;
            const int oy = y0 + (p >> TWL), ox = x0 + (p & (TW - 1));
            float v = 0.f;
            if (co0 + co < a.Cout && oy < a.Hout && ox < a.Wout) v = dyb[(long long)(co0 + co) * HWo + oy * a.Wout + ox];
            dyt[co * DYLD + p] = v;
        }
        __syncthreads();
        // ---- K loop over the 128 pixels, 4 per MFMA ------------------------------------------------------
        const float* ar = dyt + ((THIN ? 0 : wave * 16) + l15) * DYLD + kq;
        constexpr int NS = THIN ? 8 : 32;
        const int sbase = THIN ? wave * 8 : 0;
#pragma unroll 4
        for (int si = 0; si < NS; ++si) {
            const int s = sbase + si;
            const int p = 4 * s + kq;
            const int pixoff = (p >> TWL) * RS + (p & (TW - 1));
            const float av = ar[4 * s];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float bv = xt[bbase[nb] >= 0 ? bbase[nb] + pixoff : CK * PS];
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nb], 0, 0, 0);
            }
        }
    }
    // ---- stage the [NN][64] result through LDS and store it co-fastest into this split's partial --------
    __syncthreads();
    float* ot = smem;  // [NB*16][65]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(nb * 16 + l15) * 65 + wave * 16 + kq * 4 + r] = acc[nb][r];
    __syncthreads();
    float* wsp = a.ws + (long long)sp * TAPS * a.Cin * a.Cout;
    for (int e = tid; e < NN * 64; e += 256) {
        const int n = e >> 6, co = e & 63;
        const int ci = n / TAPS, tap = n - ci * TAPS;
        if (cb + ci < a.Cin && co0 + co < a.Cout)
            wsp[((long long)tap * a.Cin + cb + ci) * a.Cout + co0 + co] =
                THIN ? (ot[n * 65 + co] + ot[n * 65 + 16 + co]) + (ot[n * 65 + 32 + co] + ot[n * 65 + 48 + co]) : ot[n * 65 + co];
    }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 weight gradient as a streaming NT product:  dW[co][ci] = sum_{b,p} dY[b,co,p] * X[b,ci,p]   (plain gather:
// NORMAL mode, no prologue).  HBM-bound (two reads per multiply-add pair of rows), so the kernel is organised around
// wide loads: a workgroup owns a 64 co x 64 ci block and a contiguous share of the (sample, 128-pixel run) list; each run
// is fetched with 16-byte loads one unit ahead (register-staged), written to LDS as [row][128 + 4] (the +4 keeps rows
// 16-byte aligned and makes both MFMA operand reads bank-conflict free), and multiplied on v_mfma_f32_16x16x4_f32 with
// the pixels as the K dimension.  Two workgroups per CU keep ~128 KB of loads in flight per CU.
constexpr int W1_PX = 128, W1_LD = W1_PX + 4;
// UNSH: the input is the pixel-unshuffle(2) of src0 (virtual channel 4c + 2sy + sx = src0[c][2y+sy][2x+sx]): a thread's staging unit is
// a PAIR of virtual rows (sx = 0 / 1) x 4 output pixels = 8 consecutive floats of one input row, de-interleaved in registers.
template <bool UNSH>
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xt = smem;                 // [64 ci][W1_LD]
    float* const dyt = smem + 64 * W1_LD;   // [64 co][W1_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int cob = blockIdx.x % a.ncob;
    const int cib = (blockIdx.x / a.ncob) % a.nchunks;   // nchunks = 64-channel input blocks here
    const int sp = blockIdx.x / (a.ncob * a.nchunks);
    const int co0 = cob * 64, ci0 = cib * 64;
    const int HW = a.Hout * a.Wout;
    const int runs = HW / W1_PX;                          // per sample
    const int total = a.B * runs;
    const int per = (total + a.nsplit - 1) / a.nsplit;
    const int u_begin = sp * per, u_end = u_begin + per < total ? u_begin + per : total;
    // the block's 64 input channels come from one source (C0 % 64 == 0 when there are two)
    const bool from1 = a.src1 != nullptr && ci0 >= a.C0v;
    const float* const srcb = from1 ? a.src1 : a.src0;
    const long long sbs = from1 ? a.bs1 : a.bs0;
    const int chan0 = from1 ? ci0 - a.C0v : ci0;
    const int nci = a.Cin - ci0 < 64 ? a.Cin - ci0 : 64;  // valid rows of this block

    // staging map: 64 rows x 32 float4 = 2048 float4 per operand, 8 per thread; row = f >> 5, float4 column = f & 31
    floatx4 rx[8], ry[8];
    const int HWx = UNSH ? a.Hin * a.Win : HW;
    auto load_unit = [&](int u) {
        const int b = u / runs, p0 = (u - b * runs) * W1_PX;
        const float* yb = a.dy + (long long)b * a.dybs + (long long)co0 * HW + p0;
        if (UNSH) {
            const float* xb = srcb + (long long)b * sbs + (long long)(chan0 >> 2) * HWx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + i * 256;
                const int prow = f >> 5, p = p0 + (f & 31) * 4;  // rows 2 prow, 2 prow + 1: channel prow >> 1 of the block, sy = prow & 1
                const int oy = p / a.Wout, ox = p - oy * a.Wout;
                if (2 * prow < nci) {
                    const float* q = xb + (long long)(prow >> 1) * HWx + (2 * oy + (prow & 1)) * a.Win + 2 * ox;
                    const floatx4 l0 = *reinterpret_cast<const floatx4*>(q), l1 = *reinterpret_cast<const floatx4*>(q + 4);
                    rx[2 * i] = floatx4{l0[0], l0[2], l1[0], l1[2]};
                    rx[2 * i + 1] = floatx4{l0[1], l0[3], l1[1], l1[3]};
                } else {
                    rx[2 * i] = rx[2 * i + 1] = floatx4{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
            const float* xb = srcb + (long long)b * sbs + (long long)chan0 * HW + p0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = tid + i * 256;
                const int row = f >> 5, c4 = (f & 31) * 4;
                rx[i] = row < nci ? *reinterpret_cast<const floatx4*>(xb + (long long)row * HW + c4) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + i * 256;
            const int row = f >> 5, c4 = (f & 31) * 4;
            ry[i] = *reinterpret_cast<const floatx4*>(yb + (long long)row * HW + c4);
        }
    };
    auto store_unit = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + i * 256;
            const int row = f >> 5, c4 = (f & 31) * 4;
            if (UNSH) {
                const int fu = tid + (i >> 1) * 256;
                *reinterpret_cast<floatx4*>(xt + (2 * (fu >> 5) + (i & 1)) * W1_LD + (fu & 31) * 4) = rx[i];
            } else {
                *reinterpret_cast<floatx4*>(xt + row * W1_LD + c4) = rx[i];
            }
            *reinterpret_cast<floatx4*>(dyt + row * W1_LD + c4) = ry[i];
        }
    };
    // wave (wm, wn) owns 32 co x 32 ci = 2 x 2 blocks of 16 x 16
    const int wm = wave & 1, wn = wave >> 1;
    floatx4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* ar = dyt + (wm * 32 + l15) * W1_LD + kq;
    const float* br = xt + (wn * 32 + l15) * W1_LD + kq;
    if (u_begin < u_end) load_unit(u_begin);
    for (int u = u_begin; u < u_end; ++u) {
        __syncthreads();  // the previous unit's operand reads are done
        store_unit();
        __syncthreads();
        if (u + 1 < u_end) load_unit(u + 1);  // in flight during the MFMAs below
#pragma unroll 8
        for (int s = 0; s < W1_PX / 4; ++s) {
            const float a0 = ar[4 * s], a1 = ar[16 * W1_LD + 4 * s];
            const float b0 = br[4 * s], b1 = br[16 * W1_LD + 4 * s];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // C layout: lane holds column l15 (ci) and rows 4*kq + r (co): ws[sp][0][ci][co], co fastest -> float4 stores
    float* const wsp = a.ws + (long long)sp * a.Cin * a.Cout;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ci = ci0 + wn * 32 + n * 16 + l15;
        if (ci < a.Cin) {
#pragma unroll
            for (int m = 0; m < 2; ++m) *reinterpret_cast<floatx4*>(wsp + (long long)ci * a.Cout + co0 + wm * 32 + m * 16 + 4 * kq) = acc[m][n];
        }
    }
}

inline bool wgrad1x1_eligible(const WgArgs& a, int ks, int mode) {
    if (ks != 1 || mode == IDIFF_CONV_UPSAMPLE2 || a.pro_a) return false;
    if (mode == IDIFF_CONV_UNSHUFFLE2 && (a.Wout % 4 || a.src1)) return false;
    if (a.Cout % 64 || ((long long)a.Hout * a.Wout) % W1_PX) return false;
    if (a.src1 && a.C0v % 64) return false;
    if ((reinterpret_cast<uintptr_t>(a.src0) & 15) || (reinterpret_cast<uintptr_t>(a.src1) & 15) || (reinterpret_cast<uintptr_t>(a.dy) & 15)) return false;
    if (a.bs0 % 4 || (a.src1 && a.bs1 % 4) || a.dybs % 4) return false;
    return true;
}
inline void wgrad1x1_geometry(int Cin, int Cout, int B, int HW, int* ncob, int* ncib, int* nsplit) {
    *ncob = Cout / 64;
    *ncib = (Cin + 63) / 64;
    const int total = B * (HW / W1_PX);
    int s = 1024 / (*ncob * *ncib);  // ~2 resident workgroups per CU, two rounds
    if (s < 1) s = 1;
    if (s > total) s = total;
    *nsplit = s;
}

// dW[co][ci][tap] (+)= sum_s ws[s][tap][ci][co]   (fixed order -> deterministic)
// One workgroup owns 1024/KL consecutive elements of the [tap][ci][co] image: thread = (4 consecutive elements, split lane kl); it adds
// the splits kl, kl+KL, .. with 16-byte loads (four in flight), the KL lane sums are then added in lane order through LDS.  The
// layers with few weights and many splits (64->64: 36 864 elements x 256 splits) were 144 workgroups whose threads each walked all
// splits with one 4-byte load in flight: 71 us per launch on average, 9.5 ms of a training iteration.
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int taps, int Cin, int Cout,
                                                           int accumulate) {
    constexpr int NI4 = 256 / KL;  // element quads per workgroup
    __shared__ float4 part[KL][NI4];
    const long long n = (long long)taps * Cin * Cout;
    const int q = threadIdx.x % NI4, kl = threadIdx.x / NI4;
    const long long i0 = ((long long)blockIdx.x * NI4 + q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
    if (vec && i0 + 3 < n) {
        const float* p = ws + i0;
        int k = kl;
        for (; k + 3 * KL < nsplit; k += 4 * KL) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + (long long)k * n);
            const float4 v1 = *reinterpret_cast<const float4*>(p + (long long)(k + KL) * n);
            const float4 v2 = *reinterpret_cast<const float4*>(p + (long long)(k + 2 * KL) * n);
            const float4 v3 = *reinterpret_cast<const float4*>(p + (long long)(k + 3 * KL) * n);
            s.x += v0.x, s.y += v0.y, s.z += v0.z, s.w += v0.w;
            s.x += v1.x, s.y += v1.y, s.z += v1.z, s.w += v1.w;
            s.x += v2.x, s.y += v2.y, s.z += v2.z, s.w += v2.w;
            s.x += v3.x, s.y += v3.y, s.z += v3.z, s.w += v3.w;
        }
        for (; k < nsplit; k += KL) {
            const float4 v = *reinterpret_cast<const float4*>(p + (long long)k * n);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    } else if (i0 < n) {  // image size not a multiple of 4 / unaligned workspace (no layer of the path; kept for the C ABI's generality)
        float* sp = reinterpret_cast<float*>(&s);
        for (int e = 0; e < 4 && i0 + e < n; ++e)
            for (int k = kl; k < nsplit; k += KL) sp[e] += ws[(long long)k * n + i0 + e];
    }
    part[kl][q] = s;
    __syncthreads();
    for (int u = threadIdx.x; u < NI4 * 4; u += 256) {
        const int qq = u / 4, e = u % 4;
        const long long i = ((long long)blockIdx.x * NI4 + qq) * 4 + e;
        if (i < n) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < KL; ++l) t += reinterpret_cast<const float*>(&part[l][qq])[e];
            const int co = i % Cout;
            const int ci = (i / Cout) % Cin;
            const int tap = i / ((long long)Cout * Cin);
            const long long o = ((long long)co * Cin + ci) * taps + tap;
            dw[o] = accumulate ? dw[o] + t : t;
        }
    }
}

inline void launch_wgrad_reduce(const float* ws, float* dw, int nsplit, int taps, int Cin, int Cout, int accumulate, hipStream_t st) {
    const long long n = (long long)taps * Cin * Cout;
    // many splits of a small image: 64 split lanes x 16 elements per workgroup; otherwise 16 lanes x 64 elements
    if (nsplit >= 64 && (n + 63) / 64 < 1024)
        hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, ws, dw, nsplit, taps, Cin, Cout, accumulate);
    else if (nsplit >= 4)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, ws, dw, nsplit, taps, Cin, Cout, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, ws, dw, nsplit, taps, Cin, Cout, accumulate);
}

inline int wg_pick_twl(int Wout) { return Wout >= 32 ? 5 : (Wout >= 16 ? 4 : 3); }

inline void wg_geometry(int ks, int Cin, int Cout, int B, int Hout, int Wout, int* ck, int* nchunks, int* ncob, int* ntiles, int* tiles_x, int* nsplit) {
    *ck = ks == 3 ? (Cin <= 8 ? 8 : 16) : (ks == 1 ? 64 : 2);
    *nchunks = (Cin + *ck - 1) / *ck;
    *ncob = (Cout + 63) / 64;
    const int twl = wg_pick_twl(Wout);
    const int TW = 1 << twl, TH = 128 / TW;
    *tiles_x = (Wout + TW - 1) / TW;
    *ntiles = *tiles_x * ((Hout + TH - 1) / TH);
    const int total_tiles = B * *ntiles;
    int s = 1536 / (*nchunks * *ncob);
    if (s < 1) s = 1;
    if (s > total_tiles) s = total_tiles;
    *nsplit = s;
}

template <int KS, int CK, int MODE, bool THIN = false>
int launch_wg(const WgArgs& a, int twl, hipStream_t st) {
    const int TW = 1 << twl, TH = 128 / TW;
    const int PS = (TH + KS - 1) * (TW + KS - 1);
    const int NB = (CK * KS * KS + 15) / 16;
    size_t fl = (size_t)CK * PS + 4 + 64 * 129 + (a.pro_a ? 2 * (size_t)a.C0r : 0);
    const size_t ot = (size_t)NB * 16 * 65;
    if (ot > fl) fl = ot;
    const size_t lds = fl * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d_wgrad: LDS budget exceeded");
    dim3 grid(a.nchunks * a.ncob * a.nsplit);
#define IDIFF_WG_LAUNCH(TWL)                                                                                             \
    {                                                                                                                    \
        static size_t attr = 0;                                                                                          \
        auto kern = conv_wgrad_kernel<KS, CK, TWL, MODE, THIN>;                                                           \
        if (lds > attr) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
            attr = lds;                                                                                                  \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);                                                           \
    }
    if (twl == 5) IDIFF_WG_LAUNCH(5) else if (twl == 4) IDIFF_WG_LAUNCH(4) else IDIFF_WG_LAUNCH(3)
#undef IDIFF_WG_LAUNCH
    IDIFF_CHECK_LAUNCH("conv2d_wgrad");
    return IDIFF_OK;
}

}  // namespace

static thread_local int g_last_wgrad_algo = IDIFF_CONV_ALGO_DIRECT;
extern "C" int idiff_conv2d_wgrad_last_algo(void) { return g_last_wgrad_algo; }

extern "C" int64_t idiff_conv2d_wgrad_ws_floats(const idiff_conv_desc* d) {
    if (!d) return -1;
    const int Cin = (d->mode == IDIFF_CONV_UNSHUFFLE2 ? d->C0 * 4 : d->C0) + d->C1;
    const int Hout = d->mode == IDIFF_CONV_UPSAMPLE2 ? d->Hin * 2 : (d->mode == IDIFF_CONV_UNSHUFFLE2 ? d->Hin / 2 : d->Hin);
    const int Wout = d->mode == IDIFF_CONV_UPSAMPLE2 ? d->Win * 2 : (d->mode == IDIFF_CONV_UNSHUFFLE2 ? d->Win / 2 : d->Win);
    int ck, nch, ncob, nt, tx, ns;
    wg_geometry(d->ks, Cin, d->Cout, d->B, Hout, Wout, &ck, &nch, &ncob, &nt, &tx, &ns);
    if (d->ks == 1 && d->Cout % 64 == 0 && ((long long)Hout * Wout) % W1_PX == 0) {  // the streaming 1x1 kernel's split count
        int wcob, wcib, wns;
        wgrad1x1_geometry(Cin, d->Cout, d->B, Hout * Wout, &wcob, &wcib, &wns);
        if (wns > ns) ns = wns;
    }
    if (d->ks == 3 && d->Cout % 64 == 0) {  // the Winograd kernels may take this shape with their own split counts
        int wcob, wcib, wns;
        idiff_detail::wino_wgrad_geometry(Cin, d->Cout, d->B, Hout, Wout, &wcob, &wcib, &wns);
        if (wns > ns) ns = wns;
        if (Hout % 4 == 0 && Wout % 16 == 0) {
            idiff_detail::wino4_wgrad_geometry(Cin, d->Cout, d->B, Hout, Wout, &wcob, &wcib, &wns);
            if (wns > ns) ns = wns;
        }
    }
    return (int64_t)ns * d->ks * d->ks * Cin * d->Cout;
}

extern "C" int idiff_conv2d_wgrad(const idiff_conv_desc* d, const float* dy, int64_t dy_bstride, float* dw, int accumulate, float* ws,
                                  idiff_stream_t stream) {
    IDIFF_CHECK_ARG(d && d->src0 && dy && dw && ws, "conv2d_wgrad: null pointer");
    IDIFF_CHECK_ARG(d->B > 0 && d->C0 > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "conv2d_wgrad: bad dims");
    IDIFF_CHECK_ARG(d->ks == 1 || d->ks == 3 || d->ks == 7, "conv2d_wgrad: ks must be 1, 3 or 7");
    IDIFF_CHECK_ARG((d->C1 > 0) == (d->src1 != nullptr), "conv2d_wgrad: src1/C1 mismatch");
    IDIFF_CHECK_ARG(!(d->pro_a && d->C1 > 0), "conv2d_wgrad: prologue needs a single source");
    WgArgs a;
    a.src0 = d->src0;
    a.src1 = d->src1;
    a.bs0 = d->src0_bstride;
    a.bs1 = d->src1_bstride;
    a.B = d->B;
    a.C0r = d->C0;
    a.Hin = d->Hin;
    a.Win = d->Win;
    a.Cout = d->Cout;
    a.pro_a = d->pro_a;
    a.pro_b = d->pro_b;
    a.dy = dy;
    a.dybs = dy_bstride;
    a.ws = ws;
    if (d->mode == IDIFF_CONV_UNSHUFFLE2) {
        IDIFF_CHECK_ARG(d->ks == 1 && d->C1 == 0, "conv2d_wgrad: unshuffle mode needs ks=1 and a single source");
        a.C0v = d->C0 * 4;
        a.C1v = 0;
        a.Hout = d->Hin / 2;
        a.Wout = d->Win / 2;
    } else if (d->mode == IDIFF_CONV_UPSAMPLE2) {
        a.C0v = d->C0;
        a.C1v = d->C1;
        a.Hout = d->Hin * 2;
        a.Wout = d->Win * 2;
    } else {
        a.C0v = d->C0;
        a.C1v = d->C1;
        a.Hout = d->Hin;
        a.Wout = d->Win;
    }
    a.Cin = a.C0v + a.C1v;
    IDIFF_CHECK_ARG(dy_bstride >= (long long)d->Cout * a.Hout * a.Wout, "conv2d_wgrad: dy_bstride too small");
    int ck;
    wg_geometry(d->ks, a.Cin, a.Cout, a.B, a.Hout, a.Wout, &ck, &a.nchunks, &a.ncob, &a.ntiles, &a.tiles_x, &a.nsplit);
    const int twl = wg_pick_twl(a.Wout);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    idiff_detail::WwArgs w;
    w.src0 = a.src0, w.src1 = a.src1, w.bs0 = a.bs0, w.bs1 = a.bs1;
    w.C0v = a.C0v, w.C1v = a.C1v, w.C0r = a.C0r, w.Cin = a.Cin;
    w.B = a.B, w.Hin = a.Hin, w.Win = a.Win, w.Hout = a.Hout, w.Wout = a.Wout, w.Cout = a.Cout;
    w.pro_a = a.pro_a, w.pro_b = a.pro_b, w.dy = a.dy, w.dybs = a.dybs, w.ws = a.ws;
    w.ups = d->mode == IDIFF_CONV_UPSAMPLE2 ? 1 : 0;
    g_last_wgrad_algo = IDIFF_CONV_ALGO_DIRECT;
    if (idiff_detail::wino4_wgrad_eligible(w, d->ks, d->mode)) {
        g_last_wgrad_algo = IDIFF_CONV_ALGO_WINOGRAD4;
        idiff_detail::wino4_wgrad_geometry(w.Cin, w.Cout, w.B, w.Hout, w.Wout, &w.ncob, &w.ncib, &w.nsplit);
        a.nsplit = w.nsplit;  // for the reduction below
        rc = idiff_detail::launch_wino4_wgrad(w, st);
    } else if (idiff_detail::wino_wgrad_eligible(w, d->ks, d->mode)) {
        g_last_wgrad_algo = IDIFF_CONV_ALGO_WINOGRAD;
        idiff_detail::wino_wgrad_geometry(w.Cin, w.Cout, w.B, w.Hout, w.Wout, &w.ncob, &w.ncib, &w.nsplit);
        a.nsplit = w.nsplit;  // for the reduction below
        rc = idiff_detail::launch_wino_wgrad(w, d->mode, st);
    } else if (wgrad1x1_eligible(a, d->ks, d->mode)) {
        g_last_wgrad_algo = IDIFF_CONV_ALGO_STREAM1X1;
        wgrad1x1_geometry(a.Cin, a.Cout, a.B, a.Hout * a.Wout, &a.ncob, &a.nchunks, &a.nsplit);
        const size_t lds = (size_t)2 * 64 * W1_LD * sizeof(float);
        const bool unsh = d->mode == IDIFF_CONV_UNSHUFFLE2;
        auto kern = unsh ? wgrad1x1_kernel<true> : wgrad1x1_kernel<false>;
        static idiff_dyn_lds_cache attr[2];
        {
            hipError_t e = idiff_ensure_dyn_lds(attr[unsh], reinterpret_cast<const void*>(kern), lds);
            if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(a.ncob * a.nchunks * a.nsplit), dim3(256), lds, st, a);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d_wgrad(1x1): %s", hipGetErrorString(le));
        rc = IDIFF_OK;
    } else if (d->ks == 3) {
        IDIFF_CHECK_ARG(d->mode != IDIFF_CONV_UNSHUFFLE2, "conv2d_wgrad: unshuffle needs ks=1");
        if (d->mode == IDIFF_CONV_NORMAL && a.Cin <= 8)
            rc = a.Cout <= 16 ? launch_wg<3, 8, IDIFF_CONV_NORMAL, true>(a, twl, st) : launch_wg<3, 8, IDIFF_CONV_NORMAL>(a, twl, st);
        else if (d->mode == IDIFF_CONV_NORMAL && a.Cout <= 16)
            rc = launch_wg<3, 16, IDIFF_CONV_NORMAL, true>(a, twl, st);
        else if (a.Cin <= 8)  // (wg_geometry picked CK = 8)
            rc = launch_wg<3, 8, IDIFF_CONV_UPSAMPLE2>(a, twl, st);
        else
            rc = d->mode == IDIFF_CONV_NORMAL ? launch_wg<3, 16, IDIFF_CONV_NORMAL>(a, twl, st) : launch_wg<3, 16, IDIFF_CONV_UPSAMPLE2>(a, twl, st);
    } else if (d->ks == 1) {
        IDIFF_CHECK_ARG(d->mode != IDIFF_CONV_UPSAMPLE2, "conv2d_wgrad: upsample needs ks=3");
        rc = d->mode == IDIFF_CONV_NORMAL ? launch_wg<1, 64, IDIFF_CONV_NORMAL>(a, twl, st) : launch_wg<1, 64, IDIFF_CONV_UNSHUFFLE2>(a, twl, st);
    } else {
        IDIFF_CHECK_ARG(d->mode == IDIFF_CONV_NORMAL, "conv2d_wgrad: ks=7 needs normal mode");
        rc = launch_wg<7, 2, IDIFF_CONV_NORMAL>(a, twl, st);
    }
    if (rc != IDIFF_OK) return rc;
    launch_wgrad_reduce(ws, dw, a.nsplit, d->ks * d->ks, a.Cin, a.Cout, accumulate, st);
    IDIFF_CHECK_LAUNCH("conv2d_wgrad_reduce");
    return IDIFF_OK;
}
