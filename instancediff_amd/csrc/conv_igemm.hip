// Implicit-GEMM 2-D convolution for gfx950 on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   GEMM view:  M = Cout (BM = 64 or 32 per workgroup), N = output pixels (256 per workgroup, a TH x TW patch
//   of one sample, TW in {32,16,8}), K = taps * Cin, walked in chunks of CK input channels.
//   A (weights) and B (input patch with halo) are staged through LDS, double buffered, register-staged
//   (issue global loads for chunk c+1, run the MFMAs of chunk c, then activate + write LDS; one barrier per
//   chunk).  Each of the 4 waves owns all BM output channels x 64 pixels = MB x 2 accumulators of 32x32.
//   NCHW keeps the N (pixel) index contiguous along W, so the B-operand LDS reads, the global gathers and
//   the epilogue stores (32 consecutive pixels per half-wave) are all unit-stride.
//
//   Fused in the gather   : virtual channel concat of two sources, nearest x2 upsample, pixel-unshuffle(2),
//                           GroupNorm/FiLM affine + SiLU of the producer (zero padding applied after it).
//   Fused in the epilogue : bias, residual, per-(b,c) vector, "+ silu(a*aux+b)" term, and deterministic
//                           per-(b,c,tile) sum / sum-of-squares partials for the next GroupNorm
//                           (butterfly reduce-scatter over the 32 pixel lanes: 62 cross-lane moves per wave).
//
// f32 MFMA is an exact k-ordered fp32 fma chain (MI355X_MICROARCH.md), i.e. same numerics class as the
// reference's fp32 cuDNN/ATen convs; roofline for this kernel = 157.3 TFLOP/s (f32 matrix peak).
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

using idiff_detail::ConvArgs;

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

// SPEC: 0 = generic (prologue / second source decided at run time); 1 = single source, no prologue;
//       2 = single source + GN/SiLU prologue; 3 = two sources, no prologue.  The specialised forms drop the per-element
//       branches and 64-bit address selects from the staging code of the hot 3x3 layers.
template <int KS, int CK, int TWL, int MODE, bool VECW, int MB, int SPEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BM = 32 * MB;
    constexpr int TW = 1 << TWL;
    constexpr int TH = 256 / TW;
    constexpr int PAD = KS / 2;
    constexpr int TRH = TH + KS - 1;
    constexpr int RS = TW + KS - 1;
    constexpr int PS = TRH * RS;
    constexpr int IN_TILE = ((CK * PS + 3) / 4) * 4;
    constexpr int TAPS = KS * KS;
    constexpr int W_TILE = TAPS * CK * BM;
    constexpr int BUF = IN_TILE + W_TILE;
    constexpr int NL = (CK * PS + 255) / 256;
    constexpr int NW = VECW ? (W_TILE / 4 + 255) / 256 : (W_TILE + 255) / 256;
    constexpr int NS = TAPS * (CK / 2);  // k-steps (of 2) per chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // flattened 1x1 tiles are 256 contiguous pixels per channel: 16-byte staging, and an epilogue that transposes the
    // [BM][256+4] output tile through the (then idle) staging buffers -- the tables below sit past that tile
    constexpr bool VIN = KS == 1 && TWL == 8 && MODE == IDIFF_CONV_NORMAL;
    constexpr int TABOFF = (VIN && 32 * 260 > 2 * BUF) ? 32 * 260 : 2 * BUF;
    float* protab = smem + TABOFF;  // [2][C0r] GroupNorm/FiLM affine of this sample (only when pro_a)
    float* econst = protab + (a.pro_a ? 2 * a.C0r : 0);  // [4][BM] bias, vec, aux_a, aux_b of this block's channels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const unsigned logical = xcd_remap(blockIdx.x, a.total_wg);
    const int cob = logical % a.ncob;
    const int tile = (logical / a.ncob) % a.ntiles;
    const int b = logical / (a.ncob * a.ntiles);
    const int co0 = cob * BM;
    const int y0 = (tile / a.tiles_x) * TH;
    const int x0 = (tile % a.tiles_x) * TW;

    const int HWin = a.Hin * a.Win;
    const bool has_pro = SPEC == 0 ? (a.pro_a != nullptr) : (SPEC == 2);
    const bool two_src = SPEC == 0 ? (a.src1 != nullptr) : (SPEC == 3);

    // ---- per-thread gather descriptors (constant across chunks) ---------------------------------
    int goff[NL];
    bool gval[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + i * 256;
        const int ci = e / PS;
        const int rem = e - ci * PS;
        const int r = rem / RS;
        const int c = rem - r * RS;
        const int oy = y0 - PAD + r;
        const int ox = x0 - PAD + c;
        const bool v = (e < CK * PS) && oy >= 0 && oy < a.Hout && ox >= 0 && ox < a.Wout;
        int off;
        if (MODE == IDIFF_CONV_UPSAMPLE2) {
            off = ci * HWin + (oy >> 1) * a.Win + (ox >> 1);
        } else if (MODE == IDIFF_CONV_UNSHUFFLE2) {
            off = (ci >> 2) * HWin + (2 * oy + ((ci >> 1) & 1)) * a.Win + 2 * ox + (ci & 1);
        } else {
            off = ci * HWin + oy * a.Win + ox;
        }
        goff[i] = v ? off : 0;
        gval[i] = v;
    }

    if (has_pro) {
        for (int i = tid; i < a.C0r; i += 256) {
            protab[i] = a.pro_a[(long long)b * a.C0r + i];
            protab[a.C0r + i] = a.pro_b[(long long)b * a.C0r + i];
        }
    }

    // per-channel epilogue constants -> LDS: the epilogue must not wait on global loads between its stores (vmcnt is
    // in order over loads and stores, so each such wait would drain every store issued before it)
    if (tid < 4 * BM) {
        const int which = tid / BM, co = co0 + tid % BM;
        float v = 0.f;
        if (co < a.Cout) {
            if (which == 0 && a.bias) v = a.bias[co];
            if (which == 1 && a.vec) v = a.vec[(long long)b * a.Cout + co];
            if (which == 2 && a.aux) v = a.aux_a[(long long)b * a.Cout + co];
            if (which == 3 && a.aux) v = a.aux_b[(long long)b * a.Cout + co];
        }
        econst[tid] = v;
    }

    constexpr int NL4 = VIN ? NL / 4 : 1;
    float rin[VIN ? 1 : NL];
    floatx4 rin4[NL4];
    floatx4 rwv[VECW ? NW : 1];
    float rws[VECW ? 1 : NW];

    const int nchunks = (a.Cin + CK - 1) / CK;
    const float* const sample0 = a.src0 + (long long)b * a.bs0;  // always-valid address for masked lanes
    // 8-byte gathers of the pixel-unshuffle path need even strides and an 8-byte-aligned base (uniform)
    const bool un2 = MODE == IDIFF_CONV_UNSHUFFLE2 && ((reinterpret_cast<uintptr_t>(a.src0) & 7) == 0) && (a.bs0 & 1) == 0 && (a.Win & 1) == 0 &&
                     (NL & 1) == 0;

    // issue the global loads of chunk cc (no dependent arithmetic here: the MFMAs of the current chunk run
    // while these are in flight).  Masked elements load a valid dummy address and are zeroed at write time.
    auto load_regs = [&](int cc) {
        const int cb = cc * CK;
        const float* base0 = sample0 + (long long)(MODE == IDIFF_CONV_UNSHUFFLE2 ? (cb >> 2) : cb) * HWin;
        if (VIN) {
            const float* base1 = two_src ? a.src1 + (long long)b * a.bs1 + (long long)(cb - a.C0v) * HWin : base0;
#pragma unroll
            for (int i = 0; i < NL4; ++i) {
                const int e4 = tid + i * 256;  // float4 index in [CK][64]
                const int ci = e4 >> 6, j4 = (e4 & 63) * 4;
                const int ch = cb + ci;
                const float* p = ch < a.Cin ? ((two_src && ch >= a.C0v) ? base1 : base0) + (long long)ci * HWin + x0 + j4 : sample0;
                rin4[i] = *reinterpret_cast<const floatx4*>(p);
            }
        } else if (two_src) {
            const float* base1 = a.src1 + (long long)b * a.bs1 + (long long)(cb - a.C0v) * HWin;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int e = tid + i * 256;
                const int ch = cb + e / PS;
                const bool v = gval[i] && ch < a.Cin;
                const float* p = v ? (ch < a.C0v ? base0 : base1) + goff[i] : sample0;
                rin[i] = *p;
            }
        } else if (MODE == IDIFF_CONV_UNSHUFFLE2 && KS == 1 && un2) {
            // pixel-unshuffle: virtual channels 2j and 2j+1 of a chunk are the two x-neighbours of one input row, element i of a
            // thread is channel i of its pixel (PS = 256): one 8-byte load serves both (lanes then read contiguous 8-byte pieces;
            // the 4-byte form read every line twice at a stride of two floats: 1.6 TB/s, profiles/r03/pmc_kernels)
#pragma unroll
            for (int i = 0; i + 1 < NL; i += 2) {
                const int off = (gval[i] && cb + i < a.Cin) ? goff[i] : 0;
                const floatx2 v = *reinterpret_cast<const floatx2*>(base0 + off);
                rin[i] = v.x;
                rin[i + 1] = v.y;
            }
        } else {
            // single source: uniform (scalar) base + 32-bit per-lane offset; masked lanes read base0[0] (in range: cb < Cin)
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int e = tid + i * 256;
                const int ch = cb + e / PS;
                const int off = (gval[i] && ch < a.Cin) ? goff[i] : 0;
                rin[i] = base0[off];
            }
        }
        const float* wchunk = a.wpk + (long long)cb * a.Cout;  // uniform base of this chunk's weight rows
        if (VECW) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int f = tid + i * 256;
                const int row = f / (BM / 4);  // (tap, ci)
                const int c4 = f - row * (BM / 4);
                const int tap = row / CK;
                const int ci = row - tap * CK;
                const bool v = f < W_TILE / 4 && cb + ci < a.Cin && co0 + c4 * 4 < a.Cout;  // Cout % 4 == 0
                const int off = v ? (tap * a.Cin + ci) * a.Cout + co0 + c4 * 4 : 0;           // < 2^31: weights are small
                rwv[i] = *reinterpret_cast<const floatx4*>(wchunk + off);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int f = tid + i * 256;
                const int row = f / BM;
                const int col = f - row * BM;
                const int tap = row / CK;
                const int ci = row - tap * CK;
                const bool v = f < W_TILE && cb + ci < a.Cin && co0 + col < a.Cout;
                const float* p = v ? a.wpk + ((long long)tap * a.Cin + cb + ci) * a.Cout + co0 + col : a.wpk;
                rws[i] = *p;
            }
        }
    };

    // activation (GroupNorm/FiLM affine + SiLU of the producer) + zero padding + LDS write of chunk cc
    auto write_lds = [&](int cc, int buf) {
        const int cb = cc * CK;
        float* ib = smem + buf * BUF;
        float* wb = ib + IN_TILE;
        if (VIN) {
#pragma unroll
            for (int i = 0; i < NL4; ++i) {
                const int e4 = tid + i * 256;
                const int ch = cb + (e4 >> 6);
                floatx4 x = rin4[i];
                if (has_pro) {
                    const int chc = ch < a.C0r ? ch : 0;
                    const float pa = protab[chc], pb = protab[a.C0r + chc];
                    x = floatx4{silu_fast(pa * x.x + pb), silu_fast(pa * x.y + pb), silu_fast(pa * x.z + pb), silu_fast(pa * x.w + pb)};
                }
                if (!(ch < a.Cin)) x = floatx4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<floatx4*>(ib + e4 * 4) = x;
            }
        }
#pragma unroll
        for (int i = 0; i < (VIN ? 0 : NL); ++i) {
            const int e = tid + i * 256;
            const int ch = cb + e / PS;
            const bool v = gval[i] && ch < a.Cin;
            float x = rin[i];
            if (has_pro) {
                const int chr = MODE == IDIFF_CONV_UNSHUFFLE2 ? (ch >> 2) : ch;
                const int chc = chr < a.C0r ? chr : 0;
                x = silu_fast(protab[chc] * x + protab[a.C0r + chc]);
            }
            if (e < CK * PS) ib[e] = v ? x : 0.f;
        }
        if (VECW) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int f = tid + i * 256;
                const int row = f / (BM / 4);
                const int tap = row / CK;
                const int ci = row - tap * CK;
                const int c4 = f - row * (BM / 4);
                floatx4 w = rwv[i];
                if (!(cb + ci < a.Cin && co0 + c4 * 4 < a.Cout)) w = floatx4{0.f, 0.f, 0.f, 0.f};
                if (f < W_TILE / 4) *reinterpret_cast<floatx4*>(wb + f * 4) = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int f = tid + i * 256;
                const int row = f / BM;
                const int col = f - row * BM;
                const int tap = row / CK;
                const int ci = row - tap * CK;
                const bool v = cb + ci < a.Cin && co0 + col < a.Cout;
                if (f < W_TILE) wb[f] = v ? rws[i] : 0.f;
            }
        }
    };

    floatx16 acc[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // B-operand pixel of this lane in each of the wave's two 32-pixel blocks
    int pixoff[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int p = wave * 64 + nb * 32 + l31;
        pixoff[nb] = (p >> TWL) * RS + (p & (TW - 1));
    }

    load_regs(0);
    if (has_pro) __syncthreads();  // protab visible
    write_lds(0, 0);
    __syncthreads();

    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        if (cc + 1 < nchunks) load_regs(cc + 1);

        const float* ib = smem + buf * BUF;
        const float* wl = ib + IN_TILE + half * BM + l31;
        const float* il0 = ib + half * PS + pixoff[0];
        const float* il1 = ib + half * PS + pixoff[1];
        // k-step s covers channels (2cp, 2cp+1) of tap: A rows 2s, 2s+1 of the [tap][ci] weight image
        float an[MB], bn0, bn1;
#pragma unroll
        for (int m = 0; m < MB; ++m) an[m] = wl[m * 32];
        bn0 = il0[0];
        bn1 = il1[0];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float ac[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) ac[m] = an[m];
            const float b0 = bn0, b1 = bn1;
            if (s + 1 < NS) {  // operands of the next k-step are requested before this step's MFMAs issue
                const int s1 = s + 1;
                const int tap = s1 / (CK / 2), cp = s1 % (CK / 2);
                const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int m = 0; m < MB; ++m) an[m] = wl[2 * s1 * BM + m * 32];
                bn0 = il0[2 * cp * PS + ky * RS + kx];
                bn1 = il1[2 * cp * PS + ky * RS + kx];
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m], b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m], b1, acc[m][1], 0, 0, 0);
            }
            // pin the software pipeline: the LDS reads of k-step s+1 issue ahead of the MFMAs of k-step s, so
            // their latency hides under 2*MB*64 cycles of matrix work (hipcc otherwise sinks each read to its use).
            __builtin_amdgcn_sched_group_barrier(0x100, MB + 2, 0);   // DS reads (next step)
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MB, 0);   // MFMAs (this step)
        }
        if (cc + 1 < nchunks) write_lds(cc + 1, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const int HWo = a.Hout * a.Wout;
    int opix[2];
    bool pval[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int p = wave * 64 + nb * 32 + l31;
        const int oy = y0 + (p >> TWL), ox = x0 + (p & (TW - 1));
        pval[nb] = oy < a.Hout && ox < a.Wout;
        opix[nb] = oy * a.Wout + ox;
    }
    float* outb = a.out + (long long)b * a.obs;
    const float* resb = a.res ? a.res + (long long)b * a.rbs : nullptr;
    const float* auxb = a.aux ? a.aux + (long long)b * a.abs_ : nullptr;
    float* red = smem;  // [4 waves][BM co][2]
    const bool want_stats = a.stats != nullptr;

    if constexpr (VIN) {
        // Flattened 1x1 tiles (HBM-bound layers): the accumulators go through LDS once, as [64 co][256 px], so that every
        // global access of the epilogue is a 16-byte-per-lane, 1-KB-per-wave row segment (the C layout itself would give
        // 128-byte segments).  No GroupNorm partials on this path (idiff_conv2d_fwd keeps those layers on 8x32 patches).
        constexpr int OLD = 256 + 4;  // row pitch: 16-byte aligned rows, conflict-free column writes
        // One 32-channel half at a time ([32][260] floats = 33 KB, inside the staging buffers' 40 KB): the whole [64][260] tile
        // would make the epilogue, not the main loop, set the workgroup's LDS footprint (66.5 KB: two workgroups per CU).  At
        // 41.5 KB three are resident -- the registers' limit -- i.e. 48 KB of loads in flight per CU instead of 32 KB, which is what
        // an HBM-bound kernel is paced by (profiles/r03: 3.0 -> TB/s of the residual 1x1 convs).
        float* const ot = smem;
        const bool has_res = resb != nullptr, has_aux = auxb != nullptr;
        const long long pix0 = (long long)y0 * a.Wout + x0;  // tile start (Hout == 1: x0)
        auto rows = [&](int mb, auto res_tag, auto aux_tag) {
            constexpr bool RES = decltype(res_tag)::value, AUX = decltype(aux_tag)::value;
            constexpr int STEPS = 32 * 64 / 256;  // float4 per thread and half
            floatx4 nres = {0.f, 0.f, 0.f, 0.f}, naux = {0.f, 0.f, 0.f, 0.f};
            auto fetch = [&](int i) {
                const int f = tid + i * 256;
                const int row = mb * 32 + (f >> 6), c4 = (f & 63) * 4;
                const long long o = (co0 + row < a.Cout) ? (long long)(co0 + row) * HWo + pix0 + c4 : pix0 + c4;
                if (RES) nres = *reinterpret_cast<const floatx4*>(resb + o);
                if (AUX) naux = *reinterpret_cast<const floatx4*>(auxb + o);
            };
            fetch(0);  // requested before the barrier below: in flight while the half tile is written to LDS
            __syncthreads();  // every wave is done reading the buffers this half overwrites (main loop / previous half)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) ot[col * OLD + wave * 64 + nb * 32 + l31] = mb ? acc[MB - 1][nb][r] : acc[0][nb][r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int f = tid + i * 256;
                const int lrow = f >> 6, row = mb * 32 + lrow, c4 = (f & 63) * 4;
                const floatx4 cres = nres, caux = naux;
                if (i + 1 < STEPS) fetch(i + 1);
                floatx4 v = *reinterpret_cast<const floatx4*>(ot + lrow * OLD + c4);
                const float add = econst[row] + econst[BM + row];  // bias + per-(b,c) vector
                v = floatx4{v.x + add, v.y + add, v.z + add, v.w + add};
                if (RES) v = floatx4{v.x + cres.x, v.y + cres.y, v.z + cres.z, v.w + cres.w};
                if (AUX) {
                    const float aa = econst[2 * BM + row], ab = econst[3 * BM + row];
                    v = floatx4{v.x + silu_fast(aa * caux.x + ab), v.y + silu_fast(aa * caux.y + ab), v.z + silu_fast(aa * caux.z + ab),
                                v.w + silu_fast(aa * caux.w + ab)};
                }
                if (co0 + row < a.Cout) *reinterpret_cast<floatx4*>(outb + (long long)(co0 + row) * HWo + pix0 + c4) = v;
            }
        };
        auto halves = [&](auto res_tag, auto aux_tag) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) rows(mb, res_tag, aux_tag);
        };
        if (has_res) {
            if (has_aux) halves(std::true_type{}, std::true_type{});
            else halves(std::true_type{}, std::false_type{});
        } else {
            if (has_aux) halves(std::false_type{}, std::true_type{});
            else halves(std::false_type{}, std::false_type{});
        }
        return;
    }

    // Residual / aux operands of step i+1 are requested before the stores of step i (masked lanes read element 0 of
    // the sample and discard it), so no wait ever covers a store younger than one step; the per-channel constants
    // come from LDS.  One instantiation per (residual, aux) combination keeps the steps free of branches on them.
    float sv[MB * 32];  // per (mb, r): {sum, sumsq} over this lane's two pixels
    auto out_steps = [&](auto res_tag, auto aux_tag) {
        constexpr bool RES = decltype(res_tag)::value, AUX = decltype(aux_tag)::value;
        float nres[2], naux[2];
        auto fetch = [&](int i) {
            const int co = co0 + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) + 4 * half;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const long long o = (pval[nb] && co < a.Cout) ? (long long)co * HWo + opix[nb] : 0;
                if (RES) nres[nb] = resb[o];
                if (AUX) naux[nb] = auxb[o];
            }
        };
        fetch(0);
#pragma unroll
        for (int i = 0; i < MB * 16; ++i) {
            const int mb = i >> 4, r = i & 15;
            const int col = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = co0 + col;
            const bool cval = co < a.Cout;
            const float cres[2] = {nres[0], nres[1]}, caux[2] = {naux[0], naux[1]};
            if (i + 1 < MB * 16) fetch(i + 1);
            const float bv = econst[col], add = econst[BM + col];
            float aa = 0.f, ab = 0.f;
            if (AUX) aa = econst[2 * BM + col], ab = econst[3 * BM + col];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float v = acc[mb][nb][r] + bv;
                if (pval[nb] && cval) {
                    s += v;
                    q += v * v;
                    v += add;
                    if (RES) v += cres[nb];
                    if (AUX) v += silu_fast(aa * caux[nb] + ab);
                    outb[(long long)co * HWo + opix[nb]] = v;
                }
            }
            sv[i * 2 + 0] = s;
            sv[i * 2 + 1] = q;
        }
    };
    if (resb) {
        if (auxb) out_steps(std::true_type{}, std::true_type{});
        else out_steps(std::true_type{}, std::false_type{});
    } else {
        if (auxb) out_steps(std::false_type{}, std::true_type{});
        else out_steps(std::false_type{}, std::false_type{});
    }
    if (want_stats) {
        // butterfly reduce-scatter over the 32 lanes of each half-wave: after the step with mask m a lane keeps
        // the half of its values selected by its bit m; lane l31 ends with the totals of (mb, r) = (l31>>4, l31&15).
        constexpr int NV = MB * 32;
#pragma unroll
        for (int step = 0; step < 5; ++step) {
            const int m = 16 >> step;
            const int n = NV >> (step + 1);  // values kept after this step
            if (n >= 1) {
                const bool up = (l31 & m) != 0;
#pragma unroll
                for (int j = 0; j < n; ++j) {
                    const float lo = sv[j], hi = sv[j + n];
                    const float send = up ? lo : hi;
                    const float keep = up ? hi : lo;
                    sv[j] = keep + __shfl_xor(send, m, 64);
                }
            }
        }
        // NV = 64: 2 values/lane (sum, sumsq) for (mb, r) = (l31>>4, l31&15).  NV = 32: after 4 steps 2 values for
        // r = l31>>1 ... handled by the generic index below (the last step with n = 1 leaves 1 value).
        if (MB == 2) {
            const int mbq = l31 >> 4, r = l31 & 15;
            const int col = mbq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            red[(wave * BM + col) * 2 + 0] = sv[0];
            red[(wave * BM + col) * 2 + 1] = sv[1];
        } else {
            // NV = 32 -> after 5 steps one value per lane: index j = l31  (= r*2 + w)
            const int r = l31 >> 1, w = l31 & 1;
            const int col = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[(wave * BM + col) * 2 + w] = sv[0];
        }
        __syncthreads();
        if (tid < BM * 2) {
            const int col = tid >> 1, w = tid & 1;
            const int co = co0 + col;
            if (co < a.Cout) {
                const float t = red[(0 * BM + col) * 2 + w] + red[(1 * BM + col) * 2 + w] + red[(2 * BM + col) * 2 + w] +
                                red[(3 * BM + col) * 2 + w];
                a.stats[(((long long)b * a.ntiles + tile) * a.Cout + co) * 2 + w] = t;
            }
        }
    }
}

template <int KS, int CK, int TWL, int MODE, bool VECW, int MB, int SPEC>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = 32 * MB;
    constexpr int TW = 1 << TWL;
    constexpr int TH = 256 / TW;
    constexpr int TRH = TH + KS - 1;
    constexpr int RS = TW + KS - 1;
    constexpr int IN_TILE = ((CK * TRH * RS + 3) / 4) * 4;
    constexpr int W_TILE = KS * KS * CK * BM;
    size_t taboff = (size_t)2 * (IN_TILE + W_TILE);
    if (KS == 1 && TWL == 8 && MODE == IDIFF_CONV_NORMAL && (size_t)32 * 260 > taboff) taboff = (size_t)32 * 260;  // = TABOFF of the kernel
    const size_t lds = (taboff + (a.pro_a ? 2 * (size_t)a.C0r : 0) + 4 * BM) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d: LDS budget exceeded (%zu bytes)", lds);
    static idiff_dyn_lds_cache lds_cache;
    auto kern = conv_igemm_kernel<KS, CK, TWL, MODE, VECW, MB, SPEC>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(a.total_wg), dim3(256), lds, st, a);
    IDIFF_CHECK_LAUNCH("conv2d_fwd");
    return IDIFF_OK;
}

template <int KS, int CK, int MODE, int MB>
int dispatch_tw(const ConvArgs& a, int twl, bool vecw, hipStream_t st) {
    if (twl == 8) {  // flattened 1x1: 256 consecutive pixels per tile (set up by idiff_conv2d_fwd, KS == 1 only)
        if constexpr (KS == 1 && MODE == IDIFF_CONV_NORMAL && MB == 2) {
            if (!vecw) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d: internal: flattened tile needs vector weights");
            if (a.pro_a) return launch_conv<1, CK, 8, IDIFF_CONV_NORMAL, true, 2, 2>(a, st);
            if (a.src1) return launch_conv<1, CK, 8, IDIFF_CONV_NORMAL, true, 2, 3>(a, st);
            return launch_conv<1, CK, 8, IDIFF_CONV_NORMAL, true, 2, 1>(a, st);
        }
        IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d: internal: flattened tile requested for an unsupported variant");
    }
    if (twl == 5) {
        if (KS == 3 && MODE == IDIFF_CONV_NORMAL && MB == 2 && vecw) {  // the hot layers: exact specialisations
            if (a.pro_a) return launch_conv<KS, CK, 5, MODE, true, MB, 2>(a, st);
            if (a.src1) return launch_conv<KS, CK, 5, MODE, true, MB, 3>(a, st);
            return launch_conv<KS, CK, 5, MODE, true, MB, 1>(a, st);
        }
        return vecw ? launch_conv<KS, CK, 5, MODE, true, MB, 0>(a, st) : launch_conv<KS, CK, 5, MODE, false, MB, 0>(a, st);
    }
    if (twl == 4) return vecw ? launch_conv<KS, CK, 4, MODE, true, MB, 0>(a, st) : launch_conv<KS, CK, 4, MODE, false, MB, 0>(a, st);
    return vecw ? launch_conv<KS, CK, 3, MODE, true, MB, 0>(a, st) : launch_conv<KS, CK, 3, MODE, false, MB, 0>(a, st);
}

template <int KS, int CK, int MODE>
int dispatch_mb(const ConvArgs& a, int twl, int mb, bool vecw, hipStream_t st) {
    return mb == 2 ? dispatch_tw<KS, CK, MODE, 2>(a, twl, vecw, st) : dispatch_tw<KS, CK, MODE, 1>(a, twl, vecw, st);
}

// pixel tile of the direct kernel = 2^twl columns x 256/2^twl rows.  From 24 columns up it is the 8x32 patch of the Winograd kernel
// too (so both kernels write the same GroupNorm-partial layout and a 28-wide level -- 224/8 -- can take either).
inline int pick_twl(int Wout) { return Wout >= 24 ? 5 : (Wout >= 16 ? 4 : 3); }

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int taps, int transpose) {
    // transpose == 0: out[tap][ci][co] = w[co][ci][tap]
    // transpose == 1: out[tap][co][ci] = w[co][ci][taps-1-tap]   (flipped kernel, in/out swapped)
    const long long n = (long long)Cout * Cin * taps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (!transpose) {
            const int co = i % Cout;
            const int ci = (i / Cout) % Cin;
            const int tap = i / ((long long)Cout * Cin);
            out[i] = w[((long long)co * Cin + ci) * taps + tap];
        } else {
            const int ci = i % Cin;
            const int co = (i / Cin) % Cout;
            const int tap = i / ((long long)Cout * Cin);
            out[i] = w[((long long)co * Cin + ci) * taps + (taps - 1 - tap)];
        }
    }
}

}  // namespace

static thread_local int g_last_algo = IDIFF_CONV_ALGO_DIRECT;
extern "C" int idiff_conv2d_last_algo(void) { return g_last_algo; }

extern "C" int idiff_conv2d_num_tiles(int Hout, int Wout) {
    const int twl = pick_twl(Wout);
    const int TW = 1 << twl, TH = 256 / TW;
    return ((Wout + TW - 1) / TW) * ((Hout + TH - 1) / TH);
}

// plan = true: everything up to the choice of the kernel (argument checks, the per-sample-shape rules, g_last_algo), no launch
static int conv2d_run(const idiff_conv_desc* d, idiff_stream_t stream, const bool plan) {
    IDIFF_CHECK_ARG(d && d->src0 && d->wpk && d->out, "conv2d: null pointer");
    IDIFF_CHECK_ARG(d->B > 0 && d->C0 > 0 && d->Cout > 0 && d->Hin > 0 && d->Win > 0, "conv2d: bad dims");
    IDIFF_CHECK_ARG(d->ks == 1 || d->ks == 3 || d->ks == 7, "conv2d: ks must be 1, 3 or 7 (got %d)", d->ks);
    IDIFF_CHECK_ARG(d->mode >= 0 && d->mode <= 2, "conv2d: bad mode %d", d->mode);
    IDIFF_CHECK_ARG((d->C1 > 0) == (d->src1 != nullptr), "conv2d: src1/C1 mismatch");
    IDIFF_CHECK_ARG(!(d->pro_a && d->C1 > 0), "conv2d: prologue needs a single source");
    IDIFF_CHECK_ARG((d->pro_a == nullptr) == (d->pro_b == nullptr), "conv2d: pro_a/pro_b must both be set");
    IDIFF_CHECK_ARG(!d->aux || (d->aux_a && d->aux_b), "conv2d: aux needs aux_a/aux_b");
    ConvArgs a;
    a.src0 = d->src0;
    a.src1 = d->src1;
    a.bs0 = d->src0_bstride;
    a.bs1 = d->src1_bstride;
    a.B = d->B;
    a.C0r = d->C0;
    a.Hin = d->Hin;
    a.Win = d->Win;
    a.Cout = d->Cout;
    a.wpk = d->wpk;
    a.wwino = d->wwino;
    a.wwino4 = d->wwino4;
    a.bias = d->bias;
    a.pro_a = d->pro_a;
    a.pro_b = d->pro_b;
    a.out = d->out;
    a.obs = d->out_bstride;
    a.res = d->res;
    a.rbs = d->res_bstride;
    a.vec = d->vec;
    a.aux = d->aux;
    a.abs_ = d->aux_bstride;
    a.aux_a = d->aux_a;
    a.aux_b = d->aux_b;
    a.stats = d->stats;
    memset(&a.gn, 0, sizeof(a.gn));
    const bool want_gn = d->gn_out_a != nullptr;
    if (want_gn) {
        IDIFF_CHECK_ARG(d->stats && d->gn_out_b && d->gn_groups > 0 && d->Cout % d->gn_groups == 0 && d->Cout / d->gn_groups <= 256,
                        "conv2d: GroupNorm finalize needs stats, gn_out_b and groups dividing Cout (<= 256 channels per group)");
        a.gn.gamma = d->gn_gamma, a.gn.beta = d->gn_beta, a.gn.film = d->gn_film, a.gn.film_ld = d->gn_film_ld, a.gn.eps = d->gn_eps;
        a.gn.groups = d->gn_groups, a.gn.out_a = d->gn_out_a, a.gn.out_b = d->gn_out_b, a.gn.mean_rstd = d->gn_mean_rstd;
    }
    // the separate finalize launch behind kernels without the fused tail
    auto finalize_after = [&](int rc) -> int {
        if (rc != IDIFF_OK || !want_gn) return rc;
        return idiff_gn_finalize(d->stats, a.ntiles, a.B, a.Cout, d->gn_groups, a.Hout * a.Wout, d->gn_gamma, d->gn_beta, d->gn_film, d->gn_film_ld,
                                 d->gn_eps, d->gn_out_a, d->gn_out_b, d->gn_mean_rstd, stream);
    };
    if (d->mode == IDIFF_CONV_UNSHUFFLE2) {
        IDIFF_CHECK_ARG(d->ks == 1 && d->C1 == 0, "conv2d: unshuffle mode needs ks=1 and a single source");
        IDIFF_CHECK_ARG(d->Hin % 2 == 0 && d->Win % 2 == 0, "conv2d: unshuffle needs even H, W");
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
    IDIFF_CHECK_ARG(a.bs0 >= (long long)d->C0 * d->Hin * d->Win, "conv2d: src0_bstride too small");
    IDIFF_CHECK_ARG(d->C1 == 0 || a.bs1 >= (long long)d->C1 * d->Hin * d->Win, "conv2d: src1_bstride too small");
    IDIFF_CHECK_ARG(a.obs >= (long long)d->Cout * a.Hout * a.Wout, "conv2d: out_bstride too small");
    int twl = pick_twl(a.Wout);
    const int mb = a.Cout <= 32 ? 1 : 2;
    const bool vecw = (a.Cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.wpk) & 15) == 0);
    // A 1x1 conv has no halo, so its pixel tile may be any 256 pixels: flatten the image to one row and take 256
    // consecutive pixels per tile -- 1 KB contiguous per input channel instead of 8 rows of 128 B (HBM-bound layers).
    // Not when GroupNorm partials are requested: their layout is per 8x32 patch (idiff_conv2d_num_tiles).
    const bool in16 = (reinterpret_cast<uintptr_t>(a.src0) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.src1) & 15) == 0 && a.bs0 % 4 == 0 &&
                      (!a.src1 || a.bs1 % 4 == 0);
    if (d->ks == 1 && d->mode == IDIFF_CONV_NORMAL && !a.stats && mb == 2 && vecw && in16 && ((long long)a.Hout * a.Wout) % 256 == 0 &&
        a.Wout >= 32) {
        a.Win = a.Wout = a.Hout * a.Wout;
        a.Hin = a.Hout = 1;
        twl = 8;
    }
    const int TW = 1 << twl, TH = 256 / TW;
    const int bm = 32 * mb;
    a.tiles_x = (a.Wout + TW - 1) / TW;
    a.ntiles = a.tiles_x * ((a.Hout + TH - 1) / TH);
    a.ncob = (a.Cout + bm - 1) / bm;
    const long long total = (long long)a.B * a.ntiles * a.ncob;
    IDIFF_CHECK_ARG(total < (1ll << 31), "conv2d: grid too large");
    a.total_wg = (unsigned)total;
    hipStream_t st = (hipStream_t)stream;
    const bool hard = d->algo_request > 0;
    const int req = (hard ? d->algo_request : -d->algo_request) - 1;  // -1: the library picks
    IDIFF_CHECK_ARG(req == -1 || req == IDIFF_CONV_ALGO_DIRECT || req == IDIFF_CONV_ALGO_WINOGRAD || req == IDIFF_CONV_ALGO_WINOGRAD4 ||
                        req == IDIFF_CONV_ALGO_WINOGRAD4H || req == IDIFF_CONV_ALGO_X3,
                    "conv2d: bad algo_request %d", d->algo_request);
    // Flattened 1x1 layers with a split weight image: the bf16x3 kernel (conv1x1_x3.hip) -- fp32-class result at 2.67x the matrix
    // throughput; decided on the layer's shape only.  IDIFF_X3=0 (A/B runs) keeps them on the f32 matrix cores.
    {
        static const bool x3_on = [] {
            const char* e = getenv("IDIFF_X3");
            return !e || atoi(e) != 0;
        }();
        const bool reqx3 = req == IDIFF_CONV_ALGO_X3;
        bool can = false;
        ConvArgs ax = a;
        if (d->ks == 1 && d->wx3 != nullptr && (reinterpret_cast<uintptr_t>(d->wx3) & 15) == 0) {
            if (d->mode == IDIFF_CONV_NORMAL) {
                can = twl == 8 && idiff_detail::conv1x1_x3_eligible(ax, d->mode);
            } else if (d->mode == IDIFF_CONV_UNSHUFFLE2) {  // its own tiling: 256 consecutive output pixels x 64 channels
                ax.ncob = a.Cout / 64;
                ax.ntiles = (int)(((long long)a.Hout * a.Wout) / 256);
                ax.tiles_x = ax.ntiles;
                const long long tot = (long long)a.B * ax.ntiles * ax.ncob;
                ax.total_wg = (unsigned)tot;
                can = tot > 0 && tot < (1ll << 31) && idiff_detail::conv1x1_x3_eligible(ax, d->mode);
            }
        }
        IDIFF_CHECK_ARG(!(hard && reqx3) || can, "conv2d: algo_request bf16x3 but the layer is not a 1x1 that tiles by 256 pixels with a split weight image");
        if (can && (reqx3 || (req == -1 && x3_on))) {
            g_last_algo = IDIFF_CONV_ALGO_X3;
            if (plan) return IDIFF_OK;
            return idiff_detail::launch_conv1x1_x3(ax, d->mode, d->wx3, st);
        }
    }
    // Which F(4x4,3x3) kernel (both read the same weight image): decided on the layer's PER-SAMPLE shape only.
    //   >= 16 items of 16x32 pixels x 64 channels per sample: the 16x32 kernel (r04: with its loads by LDS-DMA and the pipeline
    //      continuous across items it also wins the two-source layers the half-patch kernel held in r03: 144->64 at 256^2 484 vs 547 us);
    //   fewer, but >= 16 half-patch (8x32) items: the half-patch kernel (the 32x32 level at c2: 576->256 136 vs 210 us);
    //   fewer still: F(2x2,3x3) / direct below.
    static const int w4h_mode = [] {  // IDIFF_W4H (A/B runs): 0 = never by itself, 1 (default) = the rule above, 2 = wherever it tiles, 3 = the r03 rule (two-source layers too)
        const char* e = getenv("IDIFF_W4H");
        return e ? atoi(e) : 1;
    }();
    const bool req4 = req == IDIFF_CONV_ALGO_WINOGRAD4, req4h = req == IDIFF_CONV_ALGO_WINOGRAD4H;
    const long long items16 = (req == -1 || req4 || req4h) ? idiff_detail::conv_wino4_items(a, d->ks, d->mode, req4 || req4h) : 0;
    if (items16 > 0) {
        const long long items8 = (long long)a.ntiles * a.ncob;
        // the 16x32 kernel streams its requests four chunks ahead, into the next item: an item has at least four chunks there
        const bool w4_ok = a.Cin >= 16;
        IDIFF_CHECK_ARG(!(hard && req4) || w4_ok, "conv2d: algo_request F(4x4,3x3) 16x32-item kernel needs Cin >= 16 (got %d)", a.Cin);
        bool half;
        if (req4h) half = true;
        else if (req4 && w4_ok) half = false;
        else if (items16 >= 16 && w4_ok) half = w4h_mode == 2 || (w4h_mode == 3 && a.src1 != nullptr);
        else half = w4h_mode >= 1 && items8 >= 16;
        if (half) {
            g_last_algo = IDIFF_CONV_ALGO_WINOGRAD4H;
            if (plan) return IDIFF_OK;
            return finalize_after(idiff_detail::launch_conv_wino4h(a, d->mode, st));
        }
        if (w4_ok && (req4 || items16 >= 16)) {
            g_last_algo = IDIFF_CONV_ALGO_WINOGRAD4;
            if (plan) return IDIFF_OK;
            return finalize_after(idiff_detail::launch_conv_wino4(a, d->mode, st));
        }
    }
    IDIFF_CHECK_ARG(!hard || !(req4 || req4h), "conv2d: algo_request F(4x4,3x3) but the shape does not tile for it");
    if ((req == -1 || req == IDIFF_CONV_ALGO_WINOGRAD || !hard) && req != IDIFF_CONV_ALGO_DIRECT && idiff_detail::conv_wino_eligible(a, d->ks, d->mode)) {
        g_last_algo = IDIFF_CONV_ALGO_WINOGRAD;
        if (plan) return IDIFF_OK;
        return finalize_after(idiff_detail::launch_conv_wino(a, d->mode, st));
    }
    IDIFF_CHECK_ARG(!hard || req != IDIFF_CONV_ALGO_WINOGRAD, "conv2d: algo_request F(2x2,3x3) but the shape does not tile for it");
    g_last_algo = IDIFF_CONV_ALGO_DIRECT;
    if (plan) {
        IDIFF_CHECK_ARG(!(d->ks == 1 && d->mode == IDIFF_CONV_UPSAMPLE2), "conv2d: upsample mode needs ks=3");
        IDIFF_CHECK_ARG(d->ks != 7 || d->mode == IDIFF_CONV_NORMAL, "conv2d: ks=7 needs normal mode");
        return IDIFF_OK;
    }
    if (d->ks == 3) {
        if (d->mode == IDIFF_CONV_NORMAL) return finalize_after(dispatch_mb<3, 8, IDIFF_CONV_NORMAL>(a, twl, mb, vecw, st));
        return finalize_after(dispatch_mb<3, 8, IDIFF_CONV_UPSAMPLE2>(a, twl, mb, vecw, st));
    }
    if (d->ks == 1) {
        IDIFF_CHECK_ARG(d->mode != IDIFF_CONV_UPSAMPLE2, "conv2d: upsample mode needs ks=3");
        if (d->mode == IDIFF_CONV_NORMAL) return finalize_after(dispatch_mb<1, 16, IDIFF_CONV_NORMAL>(a, twl, mb, vecw, st));
        return finalize_after(dispatch_mb<1, 16, IDIFF_CONV_UNSHUFFLE2>(a, twl, mb, vecw, st));
    }
    IDIFF_CHECK_ARG(d->mode == IDIFF_CONV_NORMAL, "conv2d: ks=7 needs normal mode");
    return finalize_after(dispatch_mb<7, 2, IDIFF_CONV_NORMAL>(a, twl, mb, vecw, st));
}

extern "C" int idiff_conv2d_fwd(const idiff_conv_desc* d, idiff_stream_t stream) { return conv2d_run(d, stream, false); }

extern "C" int idiff_conv2d_plan(const idiff_conv_desc* d) {
    const int keep = g_last_algo;
    const int rc = conv2d_run(d, nullptr, true);
    const int algo = g_last_algo;
    g_last_algo = keep;  // idiff_conv2d_last_algo() keeps speaking of launches
    return rc != IDIFF_OK ? rc : algo;
}

static int pack_common(const float* w, float* wpk, int Cout, int Cin, int ks, int tr, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(w && wpk && Cout > 0 && Cin > 0 && ks > 0, "pack_conv_weight: bad args");
    const long long n = (long long)Cout * Cin * ks * ks;
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wpk, Cout, Cin, ks * ks, tr);
    IDIFF_CHECK_LAUNCH("pack_conv_weight");
    return IDIFF_OK;
}
extern "C" int idiff_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int ks, idiff_stream_t stream) {
    return pack_common(w, wpk, Cout, Cin, ks, 0, stream);
}
extern "C" int idiff_pack_conv_weight_T(const float* w, float* wpk, int Cout, int Cin, int ks, idiff_stream_t stream) {
    return pack_common(w, wpk, Cout, Cin, ks, 1, stream);
}
