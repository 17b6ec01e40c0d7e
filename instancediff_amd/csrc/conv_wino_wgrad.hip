// Weight gradient of the 3x3 convolution in Winograd F(2x2,3x3) form on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   forward:  Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 2x2 output tile, summed over input channels
//   hence     dg[co][ci] = G^T [ sum_tiles (A dY A^T)[co] .* (B^T d B)[ci] ] G
//   i.e. 16 independent [Cout x T] x [T x Cin] products over T = all 2x2 tiles of all samples -- 16 multiply-adds per
//   (co, ci, tile) instead of the 36 of the direct form (conv_wgrad.hip), same saving as the forward kernel.
//
//   Workgroup = 512 threads (8 waves, 2 per SIMD) owning a 64 co x 64 ci block of dW and one contiguous share of the
//   tile list (split-K; partials are reduced by wgrad_reduce_kernel in a fixed order -> bitwise reproducible).
//   K chunk = 8 tiles in a row (2 x 16 output pixels).  Per chunk:
//     R [64 ci][4 x 18]   activated, zero-padded input patch with halo               (LDS, single buffer)
//     D [16 xi][4 ci-blocks][4 k][16 ci][2]   B^T d B      (MFMA B operand)           (LDS, double buffer)
//     E [16 xi][4 co-blocks][4 k][16 co][2]   A dY A^T     (MFMA A operand), computed from registers (dY tiles are
//                                                          loaded straight from global)  (LDS, double buffer)
//   wave (ch, cib) owns 32 co x 16 ci x all 16 xi = 128 accumulator registers; the final G^T M G is in-lane.
//   Two barriers per chunk: [MFMA xi 0..7 | stage R of chunk c+1]  barrier  [MFMA xi 8..15 | transforms of chunk c+1,
//   global loads of chunk c+2]  barrier.  The input gather has the forward kernel's semantics (virtual concat, nearest
//   x2 upsample, GroupNorm/FiLM affine + SiLU prologue, zero padding after the activation), so the fused forward needs no
//   materialised activated tensor for its backward.
//   Winograd rows are kept in the forward kernel's order (u0, u1, -u3, u2) on BOTH operands and column 3 is negated on
//   both, which makes the two wave roles of each transform arithmetically identical and removes every negation; the
//   products, and therefore M, are unchanged.
#include <stdlib.h>

#include <type_traits>

#include "conv_wgrad_args.h"

using idiff_detail::WwArgs;

namespace {

constexpr int NT = 512;
constexpr int PR = 4, PC = 18, PS = PR * PC;   // patch rows, cols, elements per channel (72: 8 channels tile the 64 banks)
constexpr int R_FLOATS = 64 * PS;              // 4608 = 9 * 512
constexpr int NL = R_FLOATS / NT;              // 9
constexpr int OP_FLOATS = 16 * 4 * 4 * 16 * 2; // 8192 per operand buffer

typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int MODE, bool PRO>
__global__ __launch_bounds__(NT) void wino_wgrad_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const R = smem;
    float* const Db = smem + R_FLOATS;         // [2][OP_FLOATS]
    float* const Eb = Db + 2 * OP_FLOATS;      // [2][OP_FLOATS]
    float* const protab = Eb + 2 * OP_FLOATS;  // [2][C0r] (PRO)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // MFMA roles
    const int j = lane & 15, k4 = lane >> 4;
    const int ch = wave & 1, cib = wave >> 1;
    // transform roles: (row pair th, channel tc, tile-in-group tk4); th is wave-uniform
    const int tk4 = tid & 3, tc = (tid >> 2) & 63, th = wave >> 2;

    const int cob = blockIdx.x % a.ncob;
    const int cibw = (blockIdx.x / a.ncob) % a.ncib;
    const int sp = blockIdx.x / (a.ncob * a.ncib);
    const int co0 = cob * 64, ci0 = cibw * 64;
    const int HWin = a.Hin * a.Win, HWo = a.Hout * a.Wout;
    const int nxc = a.Wout / 16, nty = a.Hout / 2;
    const int total = a.B * nty * nxc;
    const int per = (total + a.nsplit - 1) / a.nsplit;
    const int c_begin = sp * per;
    const int c_end = c_begin + per < total ? c_begin + per : total;

    // this block's 64 input channels live in ONE source (C0 % 64 == 0 is an eligibility condition)
    const bool from1 = a.src1 != nullptr && ci0 >= a.C0v;
    const float* const srcb = from1 ? a.src1 : a.src0;
    const long long sbs = from1 ? a.bs1 : a.bs0;
    const int chan0 = from1 ? ci0 - a.C0v : ci0;

    // ---- per-thread gather descriptors (constant) ---------------------------------------------------------------
    int gconst[NL];
    unsigned long long eflags = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + i * NT;
        const int ci = e / PS;
        const int rem = e - ci * PS;
        const int r = rem / PC, c = rem - r * PC;
        const int spo = MODE == IDIFF_CONV_UPSAMPLE2 ? (((r - 1) >> 1) + 1) * a.Win + ((c - 1) >> 1) + 1 : r * a.Win + c;
        gconst[i] = (ci0 + ci < a.Cin) ? (ci * HWin + spo) * 4 : -1;
        eflags |= (unsigned long long)((r == 0 ? 1u : 0u) | (r == PR - 1 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == PC - 1 ? 8u : 0u)) << (4 * i);
    }
    const int dyvoff = (tc * HWo + 2 * tk4) * 4;  // this thread's dY tile pair: channel tc, tiles tk4 and tk4 + 4

    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rsx, rsy;
    int goff[NL];
    int cur_b = -1;
    auto setup_chunk = [&](int idx) {  // scalar work + (border chunks only) 9 selects
        const int b = idx / (nty * nxc);
        const int rem = idx - b * (nty * nxc);
        const int ty = rem / nxc, xc = rem - ty * nxc;
        const long long org = MODE == IDIFF_CONV_UPSAMPLE2 ? (long long)(ty - 1) * a.Win + (8 * xc - 1) : (long long)(2 * ty - 1) * a.Win + (16 * xc - 1);
        rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(srcb + (long long)b * sbs + (long long)chan0 * HWin + org), 0, 0x7fffffff, RSRC_FLAGS);
        rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long long)b * a.dybs + (long long)co0 * HWo + (long long)(2 * ty) * a.Wout + 16 * xc), 0,
                                                0x7fffffff, RSRC_FLAGS);
        const unsigned edges = (ty == 0 ? 1u : 0u) | (2 * ty + 2 == a.Hout ? 2u : 0u) | (xc == 0 ? 4u : 0u) | (16 * xc + 16 == a.Wout ? 8u : 0u);
        if (edges == 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i) goff[i] = gconst[i];
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) goff[i] = ((unsigned)(eflags >> (4 * i)) & edges) ? -1 : gconst[i];
        }
        return b;
    };

    float rin[NL];
    floatx2 dyr[4];  // [row p, row q] x [tile group g]
    auto load_raw = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) rin[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, goff[i], 0, 0));
    };
    // row pair th = 0 reads dY rows (0,1) as (p,q); th = 1 reads them swapped
    const int dy_p = th ? a.Wout * 4 : 0, dy_q = th ? 0 : a.Wout * 4;
    auto load_dy = [&]() {
        dyr[0] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rsy, dyvoff, dy_p, 0));
        dyr[1] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rsy, dyvoff + 32, dy_p, 0));
        dyr[2] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rsy, dyvoff, dy_q, 0));
        dyr[3] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(rsy, dyvoff + 32, dy_q, 0));
    };
    auto stage_raw = [&](int i) {
        float x = rin[i];
        if (PRO) {
            const int chn = chan0 + (tid + i * NT) / PS;
            const int chc = chn < a.C0r ? chn : 0;
            x = silu_fast(protab[chc] * x + protab[a.C0r + chc]);
            if (goff[i] < 0) x = 0.f;  // padding is zero AFTER the activation
        }
        R[tid + i * NT] = x;
    };

    // ---- D = B^T d B of (channel tc, tiles tk4 / tk4+4), rows of pair th: same uniform-role form as the forward kernel
    float td[3][4], to[2][4][2];
    const float tsign = th ? -1.f : 1.f;
    const float* const trP = R + tc * PS + (3 * th) * PC + 2 * tk4;
    const float* const trQ = R + tc * PS + (1 + th) * PC + 2 * tk4;
    const float* const trR = R + tc * PS + (2 - th) * PC + 2 * tk4;
    auto tr_read = [&](int g) {
        const float* rows[3] = {trP + 8 * g, trQ + 8 * g, trR + 8 * g};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const floatx2 lo = *reinterpret_cast<const floatx2*>(rows[r]);
            const floatx2 hi = *reinterpret_cast<const floatx2*>(rows[r] + 2);
            td[r][0] = lo.x, td[r][1] = lo.y, td[r][2] = hi.x, td[r][3] = hi.y;
        }
    };
    auto tr_compute = [&](int g) {
        float t[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0][c] = td[0][c] - td[2][c];
            t[1][c] = __builtin_fmaf(tsign, td[2][c], td[1][c]);
        }
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            to[uu][0][g] = t[uu][0] - t[uu][2];
            to[uu][1][g] = t[uu][1] + t[uu][2];
            to[uu][2][g] = t[uu][2] - t[uu][1];
            to[uu][3][g] = t[uu][3] - t[uu][1];  // column 3 negated (and so is E's)
        }
    };
    // operand images: index ((xi*4 + block)*4 + k)*32 + lane16*2 + g, xi = 8*th + 4*uu + v
    const int opw = th * 4096 + ((tc >> 4) * 4 + tk4) * 32 + (tc & 15) * 2;
    auto tr_write = [&](int buf) {
        float* const D = Db + buf * OP_FLOATS + opw;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<floatx2*>(D + (uu * 4 + v) * 512) = floatx2{to[uu][v][0], to[uu][v][1]};
    };
    // ---- E = A dY A^T of (channel tc, tiles tk4 / tk4+4), rows of pair th, from the dY registers -------------------
    //   row vectors (stored order): pair 0: dy0, dy0+dy1;  pair 1: dy1 (= -u3 row), dy0-dy1;  with (p,q) as loaded: p, q +- p
    //   columns of a row vector (a,b): a, a+b, a-b, b (= -column 3)
    auto e_transform = [&](int buf) {
        float* const E = Eb + buf * OP_FLOATS + opw;
        float ev[2][4][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const floatx2 p = dyr[g], q = dyr[2 + g];
            const float ra[2] = {p.x, p.y};
            const float rb[2] = {__builtin_fmaf(tsign, p.x, q.x), __builtin_fmaf(tsign, p.y, q.y)};
            ev[0][0][g] = ra[0], ev[0][1][g] = ra[0] + ra[1], ev[0][2][g] = ra[0] - ra[1], ev[0][3][g] = ra[1];
            ev[1][0][g] = rb[0], ev[1][1][g] = rb[0] + rb[1], ev[1][2][g] = rb[0] - rb[1], ev[1][3][g] = rb[1];
        }
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<floatx2*>(E + (uu * 4 + v) * 512) = floatx2{ev[uu][v][0], ev[uu][v][1]};
    };
    auto load_protab = [&](int b) {
        for (int i = tid; i < a.C0r; i += NT) {
            protab[i] = a.pro_a[(long long)b * a.C0r + i];
            protab[a.C0r + i] = a.pro_b[(long long)b * a.C0r + i];
        }
    };

    floatx4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[xi][mb] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) {
        // ---- pipeline fill: operands of the first chunk in buffer 0, raw data of the second in registers -----------
        int b = setup_chunk(c_begin);
        if (PRO) {
            load_protab(b);
            cur_b = b;
        }
        load_raw();
        load_dy();
        if (PRO) __syncthreads();
#pragma unroll
        for (int i = 0; i < NL; ++i) stage_raw(i);
        __syncthreads();
        tr_read(0), tr_compute(0);
        tr_read(1), tr_compute(1);
        tr_write(0);
        e_transform(0);
        int nb = b;  // sample of the chunk whose raw data sits in the registers
        if (c_begin + 1 < c_end) {
            nb = setup_chunk(c_begin + 1);
            load_raw();
            load_dy();
        }
        __syncthreads();

        const int opoff = k4 * 32 + j * 2;
        auto chunk = [&](int c, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;  // false: last chunk of this workgroup, nothing left to stage
            const int buf = (c - c_begin) & 1;
            if (PRO && MORE && nb != cur_b) {  // the next chunk starts a new sample: its prologue table (rare: contiguous ranges)
                __syncthreads();
                load_protab(nb);
                cur_b = nb;
                __syncthreads();
            }
            const float* D = Db + buf * OP_FLOATS + cib * 128 + opoff;
            const float* E = Eb + buf * OP_FLOATS + ch * 256 + opoff;
            floatx2 ob[2], oa0[2], oa1[2];
            ob[0] = *reinterpret_cast<const floatx2*>(D);
            oa0[0] = *reinterpret_cast<const floatx2*>(E);
            oa1[0] = *reinterpret_cast<const floatx2*>(E + 128);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q + 1 < 16) {
                    ob[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(D + (q + 1) * 512);
                    oa0[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(E + (q + 1) * 512);
                    oa1[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(E + (q + 1) * 512 + 128);
                }
                const floatx2 bv = ob[q & 1], av0 = oa0[q & 1], av1 = oa1[q & 1];
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv.x, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv.x, acc[q][1], 0, 0, 0);
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv.y, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv.y, acc[q][1], 0, 0, 0);
                if (MORE) {  // the slices of the next chunk's staging, one per position
                    if (q < 8) stage_raw(q);
                    if (q == 7) stage_raw(8);
                    if (q == 8) tr_read(0);
                    if (q == 9) tr_compute(0);
                    if (q == 10) tr_read(1);
                    if (q == 11) tr_compute(1);
                    if (q == 12) tr_write(buf ^ 1);
                    if (q == 13) e_transform(buf ^ 1);
                    if (q == 14 && c + 2 < c_end) {
                        nb = setup_chunk(c + 2);
                        load_raw();
                        load_dy();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (q == 7 || q == 15) __syncthreads();
                }
            }
        };
        for (int c = c_begin; c + 1 < c_end; ++c) chunk(c, std::true_type{});
        chunk(c_end - 1, std::false_type{});
    }

    // ---- epilogue: dg = G^T M G in-lane, partial dW of this split to ws[sp][tap][ci][co] (co fastest) -------------------
    // C layout of 16x16x4: lane holds column j (ci) and rows 4*k4 + r (co) of each 16-row block; acc rows = (u0,u1,u3,u2)
    float* const wsp = a.ws + (long long)sp * 9 * a.Cin * a.Cout;
    const int ci = ci0 + cib * 16 + j;
    if (ci < a.Cin) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int co = co0 + ch * 32 + mb * 16 + 4 * k4;
            floatx4 dg[9];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t[3][4];  // t[p][v] = sum_u G[u][p] M[u][v]
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float m0 = acc[0 * 4 + v][mb][r], m1 = acc[1 * 4 + v][mb][r], m3 = acc[2 * 4 + v][mb][r], m2 = acc[3 * 4 + v][mb][r];
                    const float s12 = 0.5f * (m1 + m2), d12 = 0.5f * (m1 - m2);
                    t[0][v] = m0 + s12;
                    t[1][v] = d12;
                    t[2][v] = s12 + m3;
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const float s12 = 0.5f * (t[p][1] + t[p][2]), d12 = 0.5f * (t[p][1] - t[p][2]);
                    dg[p * 3 + 0][r] = t[p][0] + s12;
                    dg[p * 3 + 1][r] = d12;
                    dg[p * 3 + 2][r] = s12 + t[p][3];
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<floatx4*>(wsp + ((long long)tap * a.Cin + ci) * a.Cout + co) = dg[tap];
        }
    }
}

template <int MODE, bool PRO>
int launch(const WwArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)R_FLOATS + 4 * OP_FLOATS + (PRO ? 2 * (size_t)a.C0r : 0)) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d_wgrad(winograd): LDS budget exceeded (%zu bytes)", lds);
    static idiff_dyn_lds_cache lds_cache;
    auto kern = wino_wgrad_kernel<MODE, PRO>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d_wgrad(winograd): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(a.ncob * a.ncib * a.nsplit), dim3(NT), lds, st, a);
    IDIFF_CHECK_LAUNCH("conv2d_wgrad(winograd)");
    return IDIFF_OK;
}

bool wino_wgrad_disabled() {
    static const bool off = [] {
        const char* e = getenv("IDIFF_WINOGRAD");
        return e && e[0] == '0';
    }();
    return off;
}

}  // namespace

namespace idiff_detail {

bool wino_wgrad_eligible(const WwArgs& a, int ks, int mode) {
    if (ks != 3 || wino_wgrad_disabled()) return false;
    if (mode != IDIFF_CONV_NORMAL && mode != IDIFF_CONV_UPSAMPLE2) return false;
    if (a.Cout % 64 || a.Cin % 16 || a.Hout % 2 || a.Wout % 16) return false;
    if (a.src1 && a.C0v % 64) return false;
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.pro_a || a.src1)) return false;
    if (a.pro_a && a.src1) return false;
    if ((reinterpret_cast<uintptr_t>(a.dy) & 7) || (a.dybs & 1)) return false;  // float2 dY loads
    if ((long long)a.Cin * a.Hin * a.Win * 4 >= (1ll << 31) || (long long)a.Cout * a.Hout * a.Wout * 4 >= (1ll << 31)) return false;  // 32-bit lane offsets
    return true;
}

void wino_wgrad_geometry(int Cin, int Cout, int B, int Hout, int Wout, int* ncob, int* ncib, int* nsplit) {
    *ncob = Cout / 64;
    *ncib = (Cin + 63) / 64;
    const int total = B * (Hout / 2) * (Wout / 16);
    int s = 512 / (*ncob * *ncib);  // two rounds of workgroups on 256 CUs: the tail of one round overlaps the next
    if (s < 1) s = 1;
    if (s > total) s = total;
    *nsplit = s;
}

int launch_wino_wgrad(const WwArgs& a, int mode, hipStream_t st) {
    if (mode == IDIFF_CONV_UPSAMPLE2) return launch<IDIFF_CONV_UPSAMPLE2, false>(a, st);
    if (a.pro_a) return launch<IDIFF_CONV_NORMAL, true>(a, st);
    return launch<IDIFF_CONV_NORMAL, false>(a, st);
}

}  // namespace idiff_detail
