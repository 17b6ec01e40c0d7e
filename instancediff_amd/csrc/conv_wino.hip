// Winograd F(2x2,3x3) convolution for gfx950 on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   (Lavin & Gray's minimal filtering form): the 3x3 taps become 16
//   independent [Cout x Cin] x [Cin x tiles] products, one per Winograd-domain position xi -- 2.25x fewer matrix-core
//   flops than the direct implicit GEMM in conv_igemm.hip.  Numerics: fp32 throughout; the transforms only add /
//   subtract (the 0.5 factors live in the pre-transformed weights), so the result differs from the direct kernel by
//   a few ulps of reassociation, the same class of difference as cuDNN's fp32 Winograd algorithms.
//
//   Workgroup = 512 threads (8 waves, 2 per SIMD), one 8x32-pixel output patch (4x16 tiles of 2x2) x 64 output
//   channels of one sample.  Per chunk of 8 input channels:
//     R  [8][10x34 (+pad)]            activated, zero-padded input patch with halo           (LDS, single buffer)
//     V  [16 xi][4 tile-rows][4 k][16 tiles][2]   B^T d B of that patch                      (LDS, double buffer)
//     U  [16 xi][4 co-blocks][4 k][16 co][2]      pre-transformed weights, copied verbatim   (LDS, double buffer)
//   wave (ch, tb) owns 32 output channels x the 16 tiles of tile-row tb x all 16 xi = 128 accumulator registers, so
//   the output transform A^T m A is purely in-lane.  Two barriers per chunk: [MFMA xi 0..7 | stage R,U of chunk c+1]
//   barrier [MFMA xi 8..15 | transform R -> V of chunk c+1, issue global loads of chunk c+2] barrier.
//   Every LDS access of the MFMA phase is a unit-stride ds_read_b64 (512 contiguous bytes per wave).
//
//   Same fused gather (virtual concat, nearest x2 upsample, GroupNorm/FiLM affine + SiLU prologue) and the same
//   epilogue contract (bias, residual, per-(b,c) vector, "+silu(a*aux+b)", GroupNorm partials per 8x32 patch) as
//   conv_igemm.hip; the two kernels are interchangeable behind idiff_conv2d_fwd.
#include <stdlib.h>

#include "conv_args.h"

using idiff_detail::ConvArgs;

namespace {

constexpr int CK = 8;
constexpr int TW = 32, TH = 8;
constexpr int RS = TW + 2;         // 34
constexpr int TRH = TH + 2;        // 10
constexpr int PS = TRH * RS;       // 340
constexpr int PSP = 352;           // padded channel stride of R: ci and ci+1 land 32 banks apart
constexpr int R_FLOATS = CK * PSP; // 2816
constexpr int V_FLOATS = 16 * 4 * 4 * 16 * 2;  // 8192
constexpr int U_FLOATS = V_FLOATS;
constexpr int NT = 512;
constexpr int NL = (CK * PS + NT - 1) / NT;    // 6 gathered elements per thread per chunk
constexpr int NU = U_FLOATS / 4 / NT;          // 4 float4 of weights per thread per chunk

typedef float floatx2 __attribute__((ext_vector_type(2)));

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
template <int MODE, int SPEC>
__global__ __launch_bounds__(NT) void conv_wino_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const R = smem;
    float* const Vb = smem + R_FLOATS;
    float* const Ub = Vb + 2 * V_FLOATS;
    float* const protab = Ub + 2 * U_FLOATS;  // [2][C0r] (SPEC 2)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 15;   // tile column (B operand / C column) and co row within a 16-block (A operand)
    const int k4 = lane >> 4;  // k index within a group of 4 (operands) / row group of the C layout
    const int ch = wave & 1;   // co half of the MFMA role; u-pair of the transform role
    const int tb = wave >> 1;  // tile row (both roles)

    const unsigned logical = xcd_remap(blockIdx.x, a.total_wg);
    const int cob = logical % a.ncob;
    const int tile = (logical / a.ncob) % a.ntiles;
    const int b = logical / (a.ncob * a.ntiles);
    const int co0 = cob * 64;
    const int y0 = (tile / a.tiles_x) * TH;
    const int x0 = (tile % a.tiles_x) * TW;
    const int HWin = a.Hin * a.Win;

    // ---- per-thread gather descriptors (constant across chunks) ---------------------------------
    int goff[NL], rdst[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + i * NT;
        const int ci = e / PS;
        const int rem = e - ci * PS;
        const int r = rem / RS;
        const int c = rem - r * RS;
        const int oy = y0 - 1 + r;
        const int ox = x0 - 1 + c;
        const bool inb = oy >= 0 && oy < a.Hout && ox >= 0 && ox < a.Wout;
        const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
        goff[i] = (e < CK * PS && inb) ? ci * HWin + sp : -1;   // -1: zero padding (or no element)
        rdst[i] = e < CK * PS ? ci * PSP + r * RS + c : -1;
    }
    if (SPEC == 2) {
        for (int i = tid; i < a.C0r; i += NT) {
            protab[i] = a.pro_a[(long long)b * a.C0r + i];
            protab[a.C0r + i] = a.pro_b[(long long)b * a.C0r + i];
        }
    }

    const int nchunks = a.Cin / CK;
    const float* const sample0 = a.src0 + (long long)b * a.bs0;
    const float* const sample1 = SPEC == 3 ? a.src1 + (long long)b * a.bs1 : nullptr;
    const float* const ubase = a.wwino + (long long)cob * U_FLOATS;
    const long long ustride = (long long)a.ncob * U_FLOATS;

    float rin[NL];
    floatx4 ru[NU];

    auto load_regs = [&](int cc) {
        const int cb = cc * CK;
        const float* base = sample0 + (long long)cb * HWin;
        if (SPEC == 3 && cb >= a.C0v) base = sample1 + (long long)(cb - a.C0v) * HWin;  // chunk-uniform: C0v % 8 == 0
#pragma unroll
        for (int i = 0; i < NL; ++i) rin[i] = base[goff[i] < 0 ? 0 : goff[i]];
        const floatx4* up = reinterpret_cast<const floatx4*>(ubase + cc * ustride);
#pragma unroll
        for (int i = 0; i < NU; ++i) ru[i] = up[tid + i * NT];
    };

    // activation + zero padding + LDS write of the staged chunk cc (R, single buffer) and its weights (U[buf])
    auto stage = [&](int cc, int buf) {
        const int cb = cc * CK;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float x = rin[i];
            if (SPEC == 2) {
                const int chn = cb + (tid + i * NT) / PS;
                const int chc = chn < a.C0r ? chn : 0;
                x = silu_fast(protab[chc] * x + protab[a.C0r + chc]);
            }
            if (rdst[i] >= 0) R[rdst[i]] = goff[i] >= 0 ? x : 0.f;
        }
        floatx4* ud = reinterpret_cast<floatx4*>(Ub + buf * U_FLOATS);
#pragma unroll
        for (int i = 0; i < NU; ++i) ud[tid + i * NT] = ru[i];
    };

    // input transform B^T d B of the patch in R -> V[buf].  Thread = (u-pair ch, tile row tb, k4, tile j), both
    // channels ci = k4, k4+4 of the chunk; u-pair 0 needs patch rows 0..2 of the tile, u-pair 1 rows 1..3.
    auto transform = [&](int buf) {
        float* const V = Vb + buf * V_FLOATS;
        float o[2][4][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float* p = R + (k4 + 4 * g) * PSP + (2 * tb + ch) * RS + 2 * j;
            float d[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const floatx2 lo = *reinterpret_cast<const floatx2*>(p + r * RS);
                const floatx2 hi = *reinterpret_cast<const floatx2*>(p + r * RS + 2);
                d[r][0] = lo.x, d[r][1] = lo.y, d[r][2] = hi.x, d[r][3] = hi.y;
            }
            float t[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (ch == 0) {  // rows (0,1,2): u=0: d0-d2, u=1: d1+d2
                    t[0][c] = d[0][c] - d[2][c];
                    t[1][c] = d[1][c] + d[2][c];
                } else {        // rows (1,2,3): u=2: d2-d1, u=3: d1-d3
                    t[0][c] = d[1][c] - d[0][c];
                    t[1][c] = d[0][c] - d[2][c];
                }
            }
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                o[uu][0][g] = t[uu][0] - t[uu][2];
                o[uu][1][g] = t[uu][1] + t[uu][2];
                o[uu][2][g] = t[uu][2] - t[uu][1];
                o[uu][3][g] = t[uu][1] - t[uu][3];
            }
        }
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int xi = (2 * ch + uu) * 4 + v;
                *reinterpret_cast<floatx2*>(V + ((xi * 4 + tb) * 4 + k4) * 32 + j * 2) = floatx2{o[uu][v][0], o[uu][v][1]};
            }
    };

    floatx4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[xi][mb] = floatx4{0.f, 0.f, 0.f, 0.f};

    // 8 Winograd positions of chunk buffer `buf`: per xi one B read + two A reads (ds_read_b64) -> 4 MFMAs
    auto mfma8 = [&](int buf, int lo) {
        const float* V = Vb + buf * V_FLOATS + (tb * 4 + k4) * 32 + j * 2;
        const float* U = Ub + buf * U_FLOATS + (ch * 2 * 4 + k4) * 32 + j * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int xi = lo + q;
            const floatx2 bv = *reinterpret_cast<const floatx2*>(V + xi * 512);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const floatx2 av = *reinterpret_cast<const floatx2*>(U + xi * 512 + mb * 128);
                acc[xi][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[xi][mb], 0, 0, 0);
                acc[xi][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[xi][mb], 0, 0, 0);
            }
        }
    };

    load_regs(0);
    if (SPEC == 2) __syncthreads();  // protab visible
    stage(0, 0);
    __syncthreads();
    transform(0);
    load_regs(nchunks > 1 ? 1 : 0);
    __syncthreads();

    for (int cc = 0; cc + 1 < nchunks; ++cc) {
        const int buf = cc & 1;
        mfma8(buf, 0);
        stage(cc + 1, buf ^ 1);
        __syncthreads();
        mfma8(buf, 8);
        transform(buf ^ 1);
        load_regs(cc + 2 < nchunks ? cc + 2 : nchunks - 1);
        __syncthreads();
    }
    {
        const int buf = (nchunks - 1) & 1;
        mfma8(buf, 0);
        mfma8(buf, 8);
    }

    // ---- epilogue: in-lane output transform A^T m A, then the conv_igemm epilogue contract -----------------------
    // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r of each 16-row block
    const int HWo = a.Hout * a.Wout;
    const int oy = y0 + 2 * tb, ox = x0 + 2 * j;
    float* outb = a.out + (long long)b * a.obs + (long long)oy * a.Wout + ox;
    const float* resb = a.res ? a.res + (long long)b * a.rbs + (long long)oy * a.Wout + ox : nullptr;
    const float* auxb = a.aux ? a.aux + (long long)b * a.abs_ + (long long)oy * a.Wout + ox : nullptr;
    const bool want_stats = a.stats != nullptr;
    float sv[16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + ch * 32 + mb * 16 + 4 * k4 + r;
            float z[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float m0 = acc[u * 4 + 0][mb][r], m1 = acc[u * 4 + 1][mb][r], m2 = acc[u * 4 + 2][mb][r], m3 = acc[u * 4 + 3][mb][r];
                z[u][0] = m0 + m1 + m2;
                z[u][1] = m1 - m2 - m3;
            }
            const float bv = a.bias ? a.bias[co] : 0.f;
            float y[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                y[0][x] = z[0][x] + z[1][x] + z[2][x] + bv;
                y[1][x] = z[1][x] - z[2][x] - z[3][x] + bv;
            }
            sv[(mb * 4 + r) * 2 + 0] = (y[0][0] + y[0][1]) + (y[1][0] + y[1][1]);
            sv[(mb * 4 + r) * 2 + 1] = (y[0][0] * y[0][0] + y[0][1] * y[0][1]) + (y[1][0] * y[1][0] + y[1][1] * y[1][1]);
            float add = 0.f, aa = 0.f, ab = 0.f;
            if (a.vec) add = a.vec[(long long)b * a.Cout + co];
            if (auxb) {
                aa = a.aux_a[(long long)b * a.Cout + co];
                ab = a.aux_b[(long long)b * a.Cout + co];
            }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const long long o = (long long)co * HWo + dy * a.Wout;
                floatx2 v = floatx2{y[dy][0] + add, y[dy][1] + add};
                if (resb) {
                    const floatx2 rr = *reinterpret_cast<const floatx2*>(resb + o);
                    v.x += rr.x, v.y += rr.y;
                }
                if (auxb) {
                    const floatx2 ax = *reinterpret_cast<const floatx2*>(auxb + o);
                    v.x += silu_fast(aa * ax.x + ab), v.y += silu_fast(aa * ax.y + ab);
                }
                *reinterpret_cast<floatx2*>(outb + o) = v;
            }
        }
    }
    if (want_stats) {
        // butterfly reduce-scatter over the 16 tile lanes: lane j ends with the total of value index j = (mb*4+r)*2+w
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int m = 8 >> step;
            const int n = 8 >> step;
            const bool up = (j & m) != 0;
#pragma unroll
            for (int q = 0; q < n; ++q) {
                const float lo = sv[q], hi = sv[q + n];
                const float send = up ? lo : hi;
                const float keep = up ? hi : lo;
                sv[q] = keep + __shfl_xor(send, m, 64);
            }
        }
        __syncthreads();  // all MFMA-phase LDS reads are done: reuse R as the cross-wave scratch [4 tb][64 co][2]
        {
            const int mb = j >> 3, r = (j >> 1) & 3, w = j & 1;
            const int col = ch * 32 + mb * 16 + 4 * k4 + r;
            R[(tb * 64 + col) * 2 + w] = sv[0];
        }
        __syncthreads();
        if (tid < 128) {
            const float t = (R[tid] + R[128 + tid]) + (R[256 + tid] + R[384 + tid]);
            const int col = tid >> 1, w = tid & 1;
            a.stats[(((long long)b * a.ntiles + tile) * a.Cout + co0 + col) * 2 + w] = t;
        }
    }
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + (SPEC == 2 ? 2 * (size_t)a.C0r : 0)) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d(winograd): LDS budget exceeded (%zu bytes)", lds);
    static size_t attr_set = 0;
    auto kern = conv_wino_kernel<MODE, SPEC>;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd): hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(a.total_wg), dim3(NT), lds, st, a);
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd)");
    return IDIFF_OK;
}

// U = G g G^T for one (co, ci); G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int transpose) {
    // conv seen by the kernel: Co x Ci (swapped when transpose)
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    const int ncob = Co / 64;
    const long long n = (long long)Co * Ci;  // one thread per (co, ci): writes its 16 xi values
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int co = i % Co, ci = i / Co;
        float g[3][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[p][q] = transpose ? w[((long long)ci * Cin + co) * 9 + (2 - p) * 3 + (2 - q)] : w[((long long)co * Cin + ci) * 9 + p * 3 + q];
        float t[4][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
        const int cc = ci >> 3, cil = ci & 7, kk = cil & 3, gg = cil >> 2;
        const int cb = co >> 6, col = co & 63, coblk = col >> 4, i16 = col & 15;
        float* dst = out + ((long long)cc * ncob + cb) * U_FLOATS + ((coblk * 4 + kk) * 16 + i16) * 2 + gg;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v0 = t[u][0], v1 = 0.5f * (t[u][0] + t[u][1] + t[u][2]), v2 = 0.5f * (t[u][0] - t[u][1] + t[u][2]), v3 = t[u][2];
            dst[(u * 4 + 0) * 512] = v0;
            dst[(u * 4 + 1) * 512] = v1;
            dst[(u * 4 + 2) * 512] = v2;
            dst[(u * 4 + 3) * 512] = v3;
        }
    }
}

bool wino_disabled() {
    static const bool off = [] {
        const char* e = getenv("IDIFF_WINOGRAD");
        return e && e[0] == '0';
    }();
    return off;
}

}  // namespace

namespace idiff_detail {

bool conv_wino_eligible(const ConvArgs& a, int ks, int mode) {
    if (ks != 3 || !a.wwino || wino_disabled()) return false;
    if (mode != IDIFF_CONV_NORMAL && mode != IDIFF_CONV_UPSAMPLE2) return false;
    if (a.Cout % 64 || a.Cin % CK || a.C0v % CK || a.Hout % TH || a.Wout % TW) return false;
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.pro_a || a.src1)) return false;
    if (a.pro_a && a.src1) return false;
    if ((reinterpret_cast<uintptr_t>(a.wwino) & 15) != 0) return false;
    // float2 epilogue accesses: even row pitch is implied by Wout % 32; batch strides must keep 8-byte alignment
    if ((a.obs & 1) || (a.res && (a.rbs & 1)) || (a.aux && (a.abs_ & 1))) return false;
    if ((reinterpret_cast<uintptr_t>(a.out) & 7) || (reinterpret_cast<uintptr_t>(a.res) & 7) || (reinterpret_cast<uintptr_t>(a.aux) & 7)) return false;
    return true;
}

int launch_conv_wino(const ConvArgs& a, int mode, hipStream_t st) {
    if (mode == IDIFF_CONV_UPSAMPLE2) return launch<IDIFF_CONV_UPSAMPLE2, 1>(a, st);
    if (a.pro_a) return launch<IDIFF_CONV_NORMAL, 2>(a, st);
    if (a.src1) return launch<IDIFF_CONV_NORMAL, 3>(a, st);
    return launch<IDIFF_CONV_NORMAL, 1>(a, st);
}

}  // namespace idiff_detail

extern "C" int idiff_pack_conv_weight_wino(const float* w, float* wwino, int Cout, int Cin, int transpose, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(w && wwino && Cout > 0 && Cin > 0, "pack_conv_weight_wino: bad args");
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    IDIFF_CHECK_ARG(Co % 64 == 0 && Ci % 8 == 0, "pack_conv_weight_wino: needs conv Cout %% 64 == 0 and Cin %% 8 == 0 (got %d, %d)", Co, Ci);
    const long long n = (long long)Cout * Cin;
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wwino, Cout, Cin, transpose);
    IDIFF_CHECK_LAUNCH("pack_conv_weight_wino");
    return IDIFF_OK;
}
