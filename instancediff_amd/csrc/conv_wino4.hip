// Winograd F(4x4,3x3) convolution for gfx950 on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A with 6x6 transforms (interpolation points 0, +-1, +-2, inf): 36 products per
//   4x4 output tile = 2.25 multiply-adds per output and channel pair -- 4x fewer matrix-core flops than the direct form
//   and 1.78x fewer than the F(2x2,3x3) kernel of conv_wino.hip, whose structure this kernel follows.  Numerics: fp32
//   throughout; the 6x6 transforms carry constants up to 8, which costs about one decimal digit against F(2x2,3x3)
//   (max error ~1e-5 of the output range at 64..576 input channels).
//
//   Persistent workgroups of 512 threads (8 waves, 2 per SIMD), one per CU; an item = one 16x32-pixel output patch
//   (4x8 tiles of 4x4) x 64 output channels of one sample.  Per chunk of 4 input channels:
//     R  [4][18 x 40 (34 used)]                 activated, zero-padded input patch with halo            (LDS, double buffer)
//     V  [9 position quads][2 tile blocks][4 k][16 tiles][4]    B^T d B of that patch                  (LDS, double buffer)
//     U  [9 position quads][4 co blocks][4 k][16 co][4]         pre-transformed weights, verbatim     (LDS, double buffer)
//   MFMA role: wave (cb, tblk) owns 16 output channels x the 16 tiles of one 8x32 half-patch x all 36 positions = 144
//   accumulator registers, so the output transform A^T m A is in-lane and a wave's GroupNorm partials cover one whole
//   8x32 patch of the partials grid (no cross-wave combine).  Per position quad one A and one B ds_read_b128 feed four MFMAs.
//   Transform role: thread = (tile, ci) x row set: waves 0-3 produce Winograd rows (1,2) or (3,4), waves 4-7 row 0 or 5
//   of their half-patch -- one heavy and one light wave per SIMD.  ONE barrier per chunk; the staging of chunk c+2, the
//   weight copy and the transform of chunk c+1 and the global loads of chunks c+4 / c+2 are dealt out between the MFMAs
//   of chunk c.
//
//   Same fused gather (virtual concat, nearest x2 upsample, GroupNorm/FiLM affine + SiLU prologue) and the same epilogue
//   contract (bias, residual, per-(b,c) vector, "+silu(a*aux+b)", GroupNorm partials per 8x32 patch) as conv_igemm.hip
//   and conv_wino.hip; the three kernels are interchangeable behind idiff_conv2d_fwd.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"
#include "gn_tail.h"

using idiff_detail::ConvArgs;

namespace {

constexpr int CK = 4;
constexpr int TW = 32, TH = 16;
constexpr int RCOLS = TW + 2;   // 34 columns used
constexpr int RS = 40;          // row stride of R: 4*RS = 32 (mod 64) banks, rows 16-byte aligned
constexpr int TRH = TH + 2;     // 18
constexpr int PS = TRH * RS;    // 720
constexpr int PSP = 768;        // channel stride of R: 0 (mod 64) banks -> the transform's ds_read_b128 are conflict-free
constexpr int NT = 512;
constexpr int NL = 6;           // gathered elements per thread per chunk (element index = R index): 4 * 768 = 6 * 512
constexpr int R_FLOATS = NL * NT;
constexpr int V_FLOATS = 9 * 2 * 64 * 4;   // 4608
constexpr int U_FLOATS = 9 * 4 * 64 * 4;   // 9216
constexpr int NU = 5;                      // float4 of weights per thread per chunk (2304 in all: the fifth only for tid < 256)

typedef float floatx2 __attribute__((ext_vector_type(2)));

// Numbering of the 36 Winograd positions (u, v) into 9 quads of 4: the first four columns of row u fill quad PF(u), its last
// two a half of quad PH(u) shared with the neighbouring row -- every row is one 16-byte and one 8-byte piece at fixed places,
// so the transform writes them without looking at the row's parity.
__host__ __device__ constexpr int PF(int u) { return (3 * u + 1) / 2; }       // 0, 2, 3, 5, 6, 8
__host__ __device__ constexpr int PH(int u) { return 1 + 3 * (u / 2); }       // 1, 1, 4, 4, 7, 7
__host__ __device__ constexpr int pos(int u, int v) { return v < 4 ? 4 * PF(u) + v : 4 * PH(u) + 2 * (u & 1) + (v - 4); }

struct Geo4 {
    int np;        // 16x32 patches per sample
    int total;     // items = B * np * ncob
    int tiles_y8;  // rows of the 8x32 GroupNorm-partials grid
};

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over each 16-lane row by DPP prefix adds: lane 15 of the row ends with the total
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_row_shr<0x111>(v);
    v += dpp_row_shr<0x112>(v);
    v += dpp_row_shr<0x114>(v);
    v += dpp_row_shr<0x118>(v);
    return v;
}

// -DIDIFF_WINO_TRACE: per-phase cycle counts (s_memtime) summed over all items, printed by the launcher (debug builds)
#ifdef IDIFF_WINO_TRACE
#define TRACE_PARAM , long long* trace
#define TRACE_INIT long long tr_t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TRACE_MARK(k)                       \
    tr_t[k] = __builtin_readcyclecounter(); \
    if (k > 0) tr_acc[k - 1] += tr_t[k] - tr_t[k - 1];
#define TRACE_FINI                                                                                                      \
    if (tid == 0) {                                                                                                     \
        for (int q_ = 0; q_ < 8; ++q_) atomicAdd((unsigned long long*)trace + q_, (unsigned long long)tr_acc[q_]);      \
    }
#else
#define TRACE_PARAM
#define TRACE_INIT
#define TRACE_MARK(k)
#define TRACE_FINI
#endif
#ifdef IDIFF_WINO_SLOTS  // with IDIFF_WINO_TRACE: per wave, cycles per chunk of the main loop spent working / waiting at the barrier
#define SLOT_INIT long long sl_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sl_t = 0;
#define SLOT_START sl_t = __builtin_readcyclecounter();
#define SLOT_MARK(k)                                        \
    {                                                       \
        const long long t_ = __builtin_readcyclecounter();  \
        sl_acc[k] += t_ - sl_t;                             \
        sl_t = t_;                                          \
    }
#define SLOT_FINI                                                                                                                   \
    if (lane == 0) {                                                                                                                \
        long long w_ = 0;                                                                                                           \
        for (int q_ = 0; q_ < 9; ++q_) w_ += sl_acc[q_];                                                                            \
        atomicAdd((unsigned long long*)trace + 8 + 2 * wave, (unsigned long long)w_);                                               \
        atomicAdd((unsigned long long*)trace + 9 + 2 * wave, (unsigned long long)sl_acc[9]);                                        \
    }
#else
#define SLOT_INIT
#define SLOT_START
#define SLOT_MARK(k)
#define SLOT_FINI
#endif

// A wave-uniform pointer pinned to scalar registers.  Under register pressure the compiler may keep a uniform 64-bit address in
// vector registers; a buffer resource built from it then costs a waterfall loop (readfirstlane + compare + exec masking) around
// EVERY buffer load that uses it.
__device__ __forceinline__ const float* scalar_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// one 6-point input transform B^T x
__device__ __forceinline__ void bt6(const float (&x)[6], float (&o)[6]) {
    o[0] = __builtin_fmaf(4.f, x[0], __builtin_fmaf(-5.f, x[2], x[4]));
    const float p = __builtin_fmaf(-4.f, x[2], x[4]), q = __builtin_fmaf(-4.f, x[1], x[3]);
    o[1] = p + q;
    o[2] = p - q;
    const float c = x[4] - x[2], e = x[3] - x[1];
    o[3] = __builtin_fmaf(2.f, e, c);
    o[4] = __builtin_fmaf(-2.f, e, c);
    o[5] = __builtin_fmaf(4.f, x[1], __builtin_fmaf(-5.f, x[3], x[5]));
}
// one 6 -> 4 output transform A^T x
__device__ __forceinline__ void at6(const float x0, const float x1, const float x2, const float x3, const float x4, const float x5, float (&o)[4]) {
    const float s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;
    o[0] = (x0 + s1) + s2;
    o[1] = __builtin_fmaf(2.f, d2, d1);
    o[2] = __builtin_fmaf(4.f, s2, s1);
    o[3] = __builtin_fmaf(8.f, d2, d1) + x5;
}

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
// RAG: the image is not a multiple of the 16x32 patch (partial patches at the right / bottom border are masked)
template <int MODE, int SPEC, bool RAG>
__global__ __launch_bounds__(NT) void conv_wino4_kernel(const ConvArgs a, const Geo4 g TRACE_PARAM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Rb = smem;                   // [2][R_FLOATS]
    float* const Vb = smem + 2 * R_FLOATS;    // [2][V_FLOATS]
    float* const Ub = Vb + 2 * V_FLOATS;      // [2][U_FLOATS]
    float* const econst = Ub + 2 * U_FLOATS;  // [4][64] bias, vec, aux_a, aux_b of the item's 64 output channels
    int* const gtab = reinterpret_cast<int*>(econst + 256);  // [NL][NT] gather byte offsets of the current item (thread-private)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k4 = lane >> 4;  // k index of the MFMA operands (lane & 15: tile of the B operand / channel of the A operand)
    const int cb = wave & 3;   // MFMA role: 16-channel block
    const int tblk = wave >> 2;  // MFMA role: upper / lower 8x32 half-patch
    const int HWin = a.Hin * a.Win;
    const int nchunks = a.Cin / CK;  // even (Cin % 8 == 0)

    // ---- per-thread gather descriptors ------------------------------------------------------------------------------------
    // Thread stages R[tid + i*512]: the R index itself enumerates (ci, row, col), so the LDS writes are linear and unmasked.
    // An element's load offset (bytes inside the sample) is -1 for pad slots and for elements outside the image, which the raw
    // buffer load answers with 0.0.  The six offsets are decoded once per item -- a few dozen integer ops -- and parked in LDS
    // (each thread reads back only its own entries): 144 accumulators leave no registers to hold them across the item.
    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rs0, rs1, rsu;
    unsigned omask = 0;  // bit i: element i is padding / outside the image
    // The item index advances by the grid size G: (channel block, patch column, patch row, sample) are carried as a mixed-radix
    // counter with a constant increment -- scalar adds and compares per item instead of five integer divisions.
    int it_b = 0, it_cob = 0, it_px = 0, it_py = 0, it_co0 = 0, it_y0 = 0, it_x0 = 0;
    const int tiles_y = g.np / a.tiles_x;
    int d_cob, d_px, d_py, d_b;
    {
        const int G0 = gridDim.x;
        d_cob = G0 % a.ncob;
        const int r1 = G0 / a.ncob;
        d_px = r1 % a.tiles_x;
        const int r2 = r1 / a.tiles_x;
        d_py = r2 % tiles_y;
        d_b = r2 / tiles_y;
    }
    auto decode_first = [&](int item) {
        it_cob = item % a.ncob;
        const int r1 = item / a.ncob;
        it_px = r1 % a.tiles_x;
        const int r2 = r1 / a.tiles_x;
        it_py = r2 % tiles_y;
        it_b = r2 / tiles_y;
    };
    auto advance_item = [&]() {
        it_cob += d_cob;
        int carry = it_cob >= a.ncob;
        it_cob -= carry ? a.ncob : 0;
        it_px += d_px + carry;
        carry = it_px >= a.tiles_x;
        it_px -= carry ? a.tiles_x : 0;
        it_py += d_py + carry;
        carry = it_py >= tiles_y;
        it_py -= carry ? tiles_y : 0;
        it_b += d_b + carry;
    };
    auto setup_item = [&]() {
        const int cob = it_cob;
        it_co0 = cob * 64;
        it_y0 = it_py * TH;
        it_x0 = it_px * TW;
        rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scalar_ptr(a.wwino4 + (long long)cob * U_FLOATS)), 0, 0x7fffffff, RSRC_FLAGS);
        rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scalar_ptr(a.src0 + (long long)it_b * a.bs0)), 0, 0x7fffffff, RSRC_FLAGS);
        rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scalar_ptr(SPEC == 3 ? a.src1 + (long long)it_b * a.bs1 : a.src0)), 0, 0x7fffffff, RSRC_FLAGS);
        int t = tid;
        asm volatile("" : "+v"(t));  // opaque: keeps the decode here, once per item, instead of hoisted and held in registers
        omask = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = t + i * NT;
            const int ci = e / PSP;
            const int rem = e - ci * PSP;
            const int r = rem / RS;
            const int c = rem - r * RS;
            const int oy = it_y0 - 1 + r, ox = it_x0 - 1 + c;  // output-grid coordinates of the element
            const bool in = rem < PS && c < RCOLS && (unsigned)oy < (unsigned)a.Hout && (unsigned)ox < (unsigned)a.Wout;
            const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
            gtab[i * NT + tid] = in ? (ci * HWin + sp) * 4 : -1;
            omask |= (in ? 0u : 1u) << i;
        }
    };
    const int ustride_b = a.ncob * U_FLOATS * 4;  // bytes between chunks of one channel block

    float rinA[NL], rinB[NL];  // raw patches in flight: even / odd chunks
    // SPEC 2: the GroupNorm/FiLM affine of a chunk's four input channels comes through the scalar cache (uniform addresses;
    // constant address space makes them s_load_dwordx4) -- a wave's 64 staged elements never straddle a channel (768 = 12 * 64),
    // so an element's channel, and with it the affine, is wave-uniform.
    typedef const __attribute__((address_space(4))) floatx4* cfloatx4p;
    struct Pro {
        floatx4 a, b;
    };
    auto load_pro = [&](int bb, int cc) {
        Pro p;
        if (SPEC == 2) {
            const long long o = (long long)bb * a.C0r + cc * CK;
            p.a = *(cfloatx4p)(a.pro_a + o);
            p.b = *(cfloatx4p)(a.pro_b + o);
        }
        return p;
    };
    floatx4 ru[NU];

    auto load_raw = [&](float (&dst)[NL], int cc) {
        int goff[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) goff[i] = gtab[i * NT + tid];
        const int cbase = cc * CK;
        if (SPEC == 3 && cbase >= a.C0v) {  // chunk-uniform: C0v % 4 == 0
            const int so = (cbase - a.C0v) * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, goff[i], so, 0));
        } else {
            const int so = cbase * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, goff[i], so, 0));
        }
    };
    // the 2304 float4 of a weight chunk: four per thread, the last 256 by the light waves (tid >= 256)
    auto load_u = [&](int cc, auto hv_tag) {
        constexpr bool HV = decltype(hv_tag)::value;
#pragma unroll
        for (int i = 0; i < NU - 1; ++i)
            ru[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, tid * 16, cc * ustride_b + i * NT * 16, 0));
        if (!HV) ru[NU - 1] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, tid * 16, cc * ustride_b + (NU - 1) * NT * 16 - 256 * 16, 0));
    };
    auto stage_raw = [&](const float (&src)[NL], int i, const Pro& pro, int rbuf, auto hv_tag) {
        constexpr bool hiw = !decltype(hv_tag)::value;
        float x = src[i];
        if (SPEC == 2) {  // channel of element tid + i*512: (i*512 + wave*64) / 768 = {0, 0|1, 1, 2, 2|3, 3}[i]
            const float pa = i == 0 ? pro.a.x : i == 1 ? (hiw ? pro.a.y : pro.a.x) : i == 2 ? pro.a.y : i == 3 ? pro.a.z : i == 4 ? (hiw ? pro.a.w : pro.a.z) : pro.a.w;
            const float pb = i == 0 ? pro.b.x : i == 1 ? (hiw ? pro.b.y : pro.b.x) : i == 2 ? pro.b.y : i == 3 ? pro.b.z : i == 4 ? (hiw ? pro.b.w : pro.b.z) : pro.b.w;
            x = silu_fast(pa * x + pb);
        }
        Rb[rbuf * R_FLOATS + tid + i * NT] = (SPEC == 2 && ((omask >> i) & 1u)) ? 0.f : x;  // padding is zero AFTER the activation
    };
    auto stage_u = [&](int i, int buf, auto hv_tag) {
        constexpr bool HV = decltype(hv_tag)::value;
        if (i < NU - 1) reinterpret_cast<floatx4*>(Ub + buf * U_FLOATS)[tid + i * NT] = ru[i];
        else if (!HV) reinterpret_cast<floatx4*>(Ub + buf * U_FLOATS)[tid + i * NT - 256] = ru[i];
    };

    // ---- input transform B^T d B of R[rbuf] -> V[buf].  Thread = (ci = k4, tile (tyl, tx) of half-patch thalf) x row set:
    //   heavy waves 0-3: Winograd rows (1,2) (trole 0) or (3,4) (trole 1):  X = d4 + al*d2, Y = d3 + al*d1, rows X +- be*Y
    //   light waves 4-7: row 0 (from d0, d2, d4) or row 5 (from d1, d3, d5): 4*dA - 5*dB + dC
    const bool heavy = wave < 4;  // wave class: uniform; the main loop is instantiated once per class, branch-free
    const int trole = wave & 1;
    const int thalf = (wave >> 1) & 1;
    const int tx = lane & 7, tyl = (lane >> 3) & 1;
    const float al = trole ? -1.f : -4.f, be = trole ? 2.f : 1.f;
    const float* const trbase = Rb + k4 * PSP + (4 * (2 * thalf + tyl) + (heavy ? 1 : trole)) * RS + 4 * tx;
    const int ufirst = heavy ? 1 + 2 * trole : 5 * trole;  // the role's (first) Winograd row
    float* const vwbase = Vb + thalf * 256 + lane * 4;
    // five pieces, each holding at most two patch rows: T0 reads rows (d2, d4 | dA, dB); T1 folds them (X = d4 + al*d2 |
    // P = 4*dA - 5*dB) and reads (d1, d3 | dC); T2 finishes the role's Winograd rows of B^T d; T3 / T4 apply B^T along the
    // columns of one row each and write its three position pairs
    float ta[6], tb[6], tlo[6], thi[6];
    auto rd_row = [&](const float* p, float (&d)[6]) {
        const floatx4 lo = *reinterpret_cast<const floatx4*>(p);
        const floatx2 hi = *reinterpret_cast<const floatx2*>(p + 4);
        d[0] = lo.x, d[1] = lo.y, d[2] = lo.z, d[3] = lo.w, d[4] = hi.x, d[5] = hi.y;
    };
    auto tr_piece = [&](int piece, int rbuf, int buf, auto hv_tag) {
        constexpr bool heavy = decltype(hv_tag)::value;
        const float* p = trbase + rbuf * R_FLOATS;
        if (piece == 0) {
            if (heavy) rd_row(p + 1 * RS, ta), rd_row(p + 3 * RS, tb);  // d2, d4
            else rd_row(p, ta), rd_row(p + 2 * RS, tb);                 // dA, dB
        } else if (piece == 1) {
            if (heavy) {
#pragma unroll
                for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(al, ta[c], tb[c]);  // X
                rd_row(p, ta), rd_row(p + 2 * RS, tb);                                    // d1, d3
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(4.f, ta[c], -5.f * tb[c]);  // P
                rd_row(p + 4 * RS, ta);                                                        // dC
            }
        } else if (piece == 2) {
            if (heavy) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float X = tlo[c], Y = __builtin_fmaf(al, ta[c], tb[c]);
                    tlo[c] = __builtin_fmaf(be, Y, X);
                    thi[c] = __builtin_fmaf(-be, Y, X);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) tlo[c] += ta[c];
            }
        } else {
            const int which = piece - 3;
            if (which == 1 && !heavy) return;
            float o[6];
            bt6(which ? thi : tlo, o);
            const int u = ufirst + which;
            float* const V = vwbase + buf * V_FLOATS;
            *reinterpret_cast<floatx4*>(V + PF(u) * 512) = floatx4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<floatx2*>(V + PH(u) * 512 + 2 * (u & 1)) = floatx2{o[4], o[5]};
        }
    };
    auto clampc = [&](int c) { return c < nchunks ? c : nchunks - 1; };

    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = g.total;
    if (first >= last) {  // no item for this workgroup (the launchers size the grid so that it cannot happen): it still arrives
        if (a.gn.ticket) idiff_detail::gn_arrive_and_finalize(a, Rb);
        return;
    }
    float pre_e = 0.f;
    auto fetch_consts = [&]() {
        if (tid < 256) {
            const int which = tid >> 6, co = it_co0 + (tid & 63);
            pre_e = 0.f;
            if (co < a.Cout) {
                if (which == 0 && a.bias) pre_e = a.bias[co];
                if (which == 1 && a.vec) pre_e = a.vec[(long long)it_b * a.Cout + co];
                if (which == 2 && a.aux) pre_e = a.aux_a[(long long)it_b * a.Cout + co];
                if (which == 3 && a.aux) pre_e = a.aux_b[(long long)it_b * a.Cout + co];
            }
        }
    };
    decode_first(first);
    setup_item();
    load_raw(rinA, 0);
    load_raw(rinB, 1);
    if (heavy) load_u(0, std::true_type{});
    else load_u(0, std::false_type{});
    fetch_consts();
    TRACE_INIT
    SLOT_INIT

    for (int item = first; item < last; item += G) {
        const int b = it_b, co0 = it_co0, y0 = it_y0, x0 = it_x0;  // the epilogue's view of this item
        TRACE_MARK(0)

        // ---- pipeline fill: V[0], U[0] hold chunk 0, R[1] chunk 1; raw(2), raw(3) and U(1) are in registers ---------------
        __syncthreads();  // every wave is done with the previous item's LDS
        TRACE_MARK(1)
        const Pro pro0 = load_pro(b, 0), pro1 = load_pro(b, 1);
        if (tid < 256) econst[tid] = pre_e;
        floatx4 acc[36];
        // fill + main loop, instantiated per wave class (heavy: waves 0-3, light: waves 4-7) so that the role-dependent pieces
        // are straight-line code; every wave passes the same barriers
        auto run_item = [&](auto hv) {
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rinA, i, pro0, 0, hv);
#pragma unroll
            for (int i = 0; i < NU; ++i) stage_u(i, 0, hv);
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rinB, i, pro1, 1, hv);
            load_raw(rinA, clampc(2));
            load_raw(rinB, clampc(3));
            load_u(1, hv);
            TRACE_MARK(2)
            __syncthreads();
            TRACE_MARK(3)
#pragma unroll
            for (int piece = 0; piece < 5; ++piece) tr_piece(piece, 0, 0, hv);

#pragma unroll
            for (int p = 0; p < 36; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            TRACE_MARK(4)

            // ---- main loop, ONE barrier per chunk.  Iteration c runs the 18 position pairs of chunk c and, one slice per pair:
            //   stage raw(c+2) registers -> R[c&1], then load raw(c+4) into them;  stage U(c+1) -> U[(c+1)&1], then load U(c+2);
            //   transform R[(c+1)&1] (staged one iteration ago) -> V[(c+1)&1].
            const int opoff = lane * 4;
            floatx4 ob[2], oa[2];
            ob[0] = *reinterpret_cast<const floatx4*>(Vb + tblk * 256 + opoff);  // quad 0 of chunk 0
            oa[0] = *reinterpret_cast<const floatx4*>(Ub + cb * 256 + opoff);
            const bool have_next = item + G < last;
            auto chunk = [&](int cc, auto par_tag, auto more_tag) {
                constexpr int PAR = decltype(par_tag)::value;      // cc & 1: LDS buffers and the raw register set
                constexpr bool MORE = decltype(more_tag)::value;   // false: last chunk, nothing left to stage
                const float* V = Vb + PAR * V_FLOATS + tblk * 256 + opoff;
                const float* U = Ub + PAR * U_FLOATS + cb * 256 + opoff;
                float(&rin)[NL] = PAR ? rinB : rinA;
                Pro pro;
                if (MORE) pro = load_pro(b, clampc(cc + 2));  // the affine of the chunk staged below
                // Operand quads alternate between two register sets; the parity flips from chunk to chunk (9 quads), so quad 8 of
                // this chunk and quad 0 of the next never share a set: the next chunk's first operands are requested right after
                // the barrier and arrive while the four MFMAs of this chunk's last quad run.
                const float* Vn = Vb + (PAR ^ 1) * V_FLOATS + tblk * 256 + opoff;
                const float* Un = Ub + (PAR ^ 1) * U_FLOATS + cb * 256 + opoff;
                if (MORE) { SLOT_START }
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (q + 1 < 9) {
                        ob[(q + 1 + PAR) & 1] = *reinterpret_cast<const floatx4*>(V + (q + 1) * 512);
                        oa[(q + 1 + PAR) & 1] = *reinterpret_cast<const floatx4*>(U + (q + 1) * 1024);
                    } else if (MORE) {
#ifndef W4_NO_RAW
#ifndef W4_NO_RAWLOAD
                        load_raw(rin, clampc(cc + 4));
#endif
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        SLOT_MARK(8)
                        __syncthreads();
                        SLOT_MARK(9)
                        ob[PAR ^ 1] = *reinterpret_cast<const floatx4*>(Vn);
                        oa[PAR ^ 1] = *reinterpret_cast<const floatx4*>(Un);
                    }
                    const floatx4 bv = ob[(q + PAR) & 1], av = oa[(q + PAR) & 1];
                    acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[4 * q + 0], 0, 0, 0);
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[4 * q + 1], 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[4 * q + 2], 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[4 * q + 3], 0, 0, 0);
                    if (MORE) {
#ifndef W4_NO_TR
                        if (q < 5) tr_piece(q, PAR ^ 1, PAR ^ 1, hv);
#endif
#ifndef W4_NO_U
                        if (q >= 1 && q < 5) stage_u(q - 1, PAR ^ 1, hv);
                        if (q == 5) stage_u(4, PAR ^ 1, hv);
#ifndef W4_NO_ULOAD
                        if (q == 6) load_u(clampc(cc + 2), hv);
#endif
#endif
#ifndef W4_NO_RAW
                        if (q == 5) stage_raw(rin, 0, pro, PAR, hv), stage_raw(rin, 1, pro, PAR, hv);
                        if (q == 6) stage_raw(rin, 2, pro, PAR, hv), stage_raw(rin, 3, pro, PAR, hv);
                        if (q == 7) stage_raw(rin, 4, pro, PAR, hv), stage_raw(rin, 5, pro, PAR, hv);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        if (q < 8) { SLOT_MARK(q) }
                    } else if (have_next) {
                        // Nothing is staged in the last chunk, so the item state is free: switch it to the next item between the
                        // MFMAs and let its first patches and weights travel during the rest of the chunk and the epilogue.
                        if (q == 0) advance_item(), setup_item();
                        if (q == 2) load_raw(rinA, 0);
                        if (q == 3) load_raw(rinB, 1);
                        if (q == 4) load_u(0, hv);
                        if (q == 5) fetch_consts();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            for (int cc = 0; cc + 2 < nchunks; cc += 2) {
                chunk(cc, std::integral_constant<int, 0>{}, std::true_type{});
                chunk(cc + 1, std::integral_constant<int, 1>{}, std::true_type{});
            }
            chunk(nchunks - 2, std::integral_constant<int, 0>{}, std::true_type{});
            chunk(nchunks - 1, std::integral_constant<int, 1>{}, std::false_type{});
        };
        if (heavy) run_item(std::true_type{});
        else run_item(std::false_type{});
        TRACE_MARK(5)

        // ---- epilogue: in-lane output transform A^T m A, then the conv_igemm epilogue contract ---------------------------
        // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r (channels) of the wave's 16-channel block
        if (co0 + cb * 16 < a.Cout) {  // uniform: a 16-channel block beyond a partial Cout has nothing to store
            // lane-derived constants are recomputed here from an opaque copy of the lane id: hoisted out of the item loop they
            // would be spilled (the main loop has no register to spare) and reloaded through the same in-order vmcnt queue as
            // the output stores
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int j = lane_e & 15, k4 = lane_e >> 4;
            const int HWo = a.Hout * a.Wout;
            const int ty0 = y0 + 8 * tblk;  // first row of the wave's half-patch
            const long long wave_org = (long long)(co0 + cb * 16) * HWo + (long long)ty0 * a.Wout + x0;
            float* const outb = a.out + (long long)b * a.obs + wave_org;
            const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
            const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
            const unsigned lane_off = (unsigned)(4 * k4) * (unsigned)HWo + (unsigned)(4 * (j >> 3)) * (unsigned)a.Wout + 4u * (j & 7);
            const float* const ebase = econst + cb * 16 + 4 * k4;
            const bool want_stats = a.stats != nullptr && ty0 < a.Hout;
            const bool has_res = a.res != nullptr, has_aux = a.aux != nullptr;
            // partial patches: H and W are multiples of 4, so a 4x4 tile lies inside the image or outside it
            const bool inside = !RAG || ((ty0 + 4 * (j >> 3) < a.Hout) && (x0 + 4 * (j & 7) < a.Wout));
            float* const stp = want_stats ? a.stats + (((long long)b * a.ntiles + (ty0 >> 3) * a.tiles_x + (x0 >> 5)) * a.Cout + co0 + cb * 16 + 4 * k4) * 2 : nullptr;
            // Phase 1: A^T along the Winograd columns v of every row u and channel r -- 144 accumulators shrink to 96 values
            // (the accumulators of a row die as soon as it is done, so the registers hold either form, never both).
            float zz[6][4][4];  // [u][r][dx]
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    at6(acc[pos(u, 0)][r], acc[pos(u, 1)][r], acc[pos(u, 2)][r], acc[pos(u, 3)][r], acc[pos(u, 4)][r], acc[pos(u, 5)][r], zz[u][r]);
                    // pinned here: left alone, the optimiser sinks these sums to their uses in phase 2 and keeps the accumulators --
                    // spilled -- until then, reloading them through the same in-order vmcnt queue as the output stores
                    asm volatile("" : "+v"(zz[u][r][0]), "+v"(zz[u][r][1]), "+v"(zz[u][r][2]), "+v"(zz[u][r][3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TRACE_MARK(6)
            // Phase 2, per channel r: A^T along u, bias, GroupNorm partials, then four row steps.  The residual / aux row of
            // step s+1 is requested BEFORE the store of step s (vmcnt counts loads and stores in order: a load behind a store
            // would wait for it).  has_res / has_aux are uniform branches.
            floatx4 nres = floatx4{0.f, 0.f, 0.f, 0.f}, naux = nres;
            auto fetch = [&](int s) {
                if (!inside) return;
                const long long so = (long long)(s >> 2) * HWo + (s & 3) * a.Wout;  // uniform
                if (has_res) nres = *reinterpret_cast<const floatx4*>(resb + so + lane_off);
                if (has_aux) naux = *reinterpret_cast<const floatx4*>(auxb + so + lane_off);
            };
            fetch(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = ebase[r];
                float y[4][4];  // [dy][dx]
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float col[4];
                    at6(zz[0][r][x], zz[1][r][x], zz[2][r][x], zz[3][r][x], zz[4][r][x], zz[5][r][x], col);
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy) y[dy][x] = col[dy] + bv;
                }
                if (want_stats) {
                    float ssum = 0.f, ssq = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy) {
                        ssum += (y[dy][0] + y[dy][1]) + (y[dy][2] + y[dy][3]);
                        ssq += (y[dy][0] * y[dy][0] + y[dy][1] * y[dy][1]) + (y[dy][2] * y[dy][2] + y[dy][3] * y[dy][3]);
                    }
                    if (!inside) ssum = 0.f, ssq = 0.f;
                    ssum = row_sum16(ssum);
                    ssq = row_sum16(ssq);
                    if (j == 15) idiff_detail::gn_store_partial(stp + 2 * r, ssum, ssq);  // write-through: read by another workgroup (gn_tail.h)
                }
                const float add = ebase[64 + r];
                float aa = 0.f, ab = 0.f;
                if (has_aux) aa = ebase[128 + r], ab = ebase[192 + r];
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const floatx4 cres = nres, caux = naux;
                    if (4 * r + dy + 1 < 16) fetch(4 * r + dy + 1);
                    floatx4 v = floatx4{y[dy][0] + add, y[dy][1] + add, y[dy][2] + add, y[dy][3] + add};
                    if (has_res) v += cres;
                    if (has_aux) {
                        v.x += silu_fast(aa * caux.x + ab), v.y += silu_fast(aa * caux.y + ab);
                        v.z += silu_fast(aa * caux.z + ab), v.w += silu_fast(aa * caux.w + ab);
                    }
#ifdef W4_NO_STORE  // diagnostic: keeps the arithmetic alive, stores (almost) nothing
                    if (inside && v.x == 12345.678f) *reinterpret_cast<floatx4*>(outb + ((long long)r * HWo + dy * a.Wout) + lane_off) = v;
#else
                    if (inside) *reinterpret_cast<floatx4*>(outb + ((long long)r * HWo + dy * a.Wout) + lane_off) = v;
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TRACE_MARK(7)
        TRACE_MARK(8)
    }
    if (a.gn.ticket) idiff_detail::gn_arrive_and_finalize(a, Rb);  // GroupNorm finalize as the tail of this launch (gn_tail.h)
    TRACE_FINI
    SLOT_FINI
}

template <int MODE, int SPEC, bool RAG>
int launch_rag(const ConvArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)2 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 256 + NL * NT) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d(winograd4): LDS budget exceeded (%zu bytes)", lds);
    static idiff_dyn_lds_cache lds_cache;
    auto kern = conv_wino4_kernel<MODE, SPEC, RAG>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4): cannot query the CU count");
        num_cu = n;
    }
    Geo4 g;
    const int tiles_y16 = (a.Hout + TH - 1) / TH;
    g.np = a.tiles_x * tiles_y16;
    g.tiles_y8 = (a.Hout + 7) / 8;
    const long long total = (long long)a.B * g.np * a.ncob;
    if (total >= (1ll << 31)) IDIFF_FAIL(IDIFF_E_BADARG, "conv2d(winograd4): grid too large");
    g.total = (int)total;
    const int per = (g.total + num_cu - 1) / num_cu;
    const int grid = (g.total + per - 1) / per;
#ifdef IDIFF_WINO_TRACE
    static long long* tr = nullptr;
    if (!tr) (void)hipMalloc(&tr, 32 * sizeof(long long));
    (void)hipMemsetAsync(tr, 0, 32 * sizeof(long long), st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, g, tr);
    long long h[32];
    (void)hipMemcpyAsync(h, tr, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[wino4 trace] Cin=%d Cout=%d H=%d items=%d per=%d | topbar %lld stage %lld bar2 %lld tr+bar3 %lld loop %lld epi1 %lld epi2 %lld (cycles/item, wave 0)\n",
            a.Cin, a.Cout, a.Hout, g.total, per, h[0] / g.total, h[1] / g.total, h[2] / g.total, h[3] / g.total, h[4] / g.total, h[5] / g.total, h[6] / g.total);
#ifdef IDIFF_WINO_SLOTS
    {
        const long long nch = (long long)g.total * (a.Cin / CK - 1);  // staged chunks
        fprintf(stderr, "[wino4 slots] Cin=%d H=%d work/barrier-wait cycles per chunk, waves 0..7:", a.Cin, a.Hout);
        for (int w = 0; w < 8; ++w) fprintf(stderr, " %lld/%lld", h[8 + 2 * w] / nch, h[9 + 2 * w] / nch);
        fprintf(stderr, "\n");
    }
#endif
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, g);
#endif
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd4)");
    return IDIFF_OK;
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, hipStream_t st) {
    if (a.Hout % TH || a.Wout % TW) return launch_rag<MODE, SPEC, true>(a, st);
    return launch_rag<MODE, SPEC, false>(a, st);
}

// U = G g G^T for one (co, ci); G rows: [1/4,0,0], [-1/6,-1/6,-1/6], [-1/6,1/6,-1/6], [1/24,1/12,1/6], [1/24,-1/12,1/6], [0,0,1]
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int transpose) {
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;  // the conv seen by the kernel
    const int ncob = (Co + 63) / 64;
    const long long n = (long long)Co * Ci;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int co = i % Co, ci = i / Co;
        float gk[3][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                gk[p][q] = transpose ? w[((long long)ci * Cin + co) * 9 + (2 - p) * 3 + (2 - q)] : w[((long long)co * Cin + ci) * 9 + p * 3 + q];
        auto g6 = [](float x0, float x1, float x2, float(&o)[6]) {
            o[0] = 0.25f * x0;
            const float s = x0 + x2;
            o[1] = (-1.f / 6.f) * (s + x1);
            o[2] = (-1.f / 6.f) * (s - x1);
            const float t = __builtin_fmaf(4.f, x2, x0);  // x0 + 4 x2
            o[3] = (1.f / 24.f) * __builtin_fmaf(2.f, x1, t);
            o[4] = (1.f / 24.f) * __builtin_fmaf(-2.f, x1, t);
            o[5] = x2;
        };
        float t[6][3];  // G g
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float o[6];
            g6(gk[0][q], gk[1][q], gk[2][q], o);
#pragma unroll
            for (int u = 0; u < 6; ++u) t[u][q] = o[u];
        }
        const int cc = ci >> 2, kk = ci & 3;
        const int cbk = co >> 6, col = co & 63, cb = col >> 4, i16 = col & 15;
        float* dst = out + ((long long)cc * ncob + cbk) * U_FLOATS + cb * 256 + (kk * 16 + i16) * 4;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            float o[6];
            g6(t[u][0], t[u][1], t[u][2], o);
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                const int p = pos(u, v);
                dst[(p >> 2) * 1024 + (p & 3)] = o[v];
            }
        }
    }
}

int wino4_mode() {  // IDIFF_WINOGRAD4: 0 = never, 1 (default) = where eligible
    static const int m = [] {
        const char* e = getenv("IDIFF_WINOGRAD4");
        return e ? atoi(e) : 1;
    }();
    return m;
}
}  // namespace

namespace idiff_detail {

// 0 when the F(4x4,3x3) kernels do not cover the problem (shape, alignment, no weight image, IDIFF_WINOGRAD4=0 unless asked for by
// name); otherwise the number of 16x32-pixel x 64-channel items PER SAMPLE -- what the choice between the kernels is made on, never
// the batch: which kernel serves a layer, and so every bit of its result, must not depend on the batch a sample sits in.
long long conv_wino4_items(const ConvArgs& a, int ks, int mode, bool requested) {
    if (ks != 3 || !a.wwino4 || (wino4_mode() == 0 && !requested)) return 0;
    if (mode != IDIFF_CONV_NORMAL && mode != IDIFF_CONV_UPSAMPLE2) return 0;
    if (a.Cout % 16 || a.Cin % 8 || a.C0v % CK || (a.Hout & 3) || (a.Wout & 3) || a.Wout < 24) return 0;
    if ((long long)a.Cin * a.Hin * a.Win * 4 >= (1ll << 31)) return 0;  // 32-bit byte offsets inside a sample
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.pro_a || a.src1)) return 0;
    if (a.pro_a && a.src1) return 0;
    if (a.pro_a && ((a.C0r & 3) || (reinterpret_cast<uintptr_t>(a.pro_a) & 15) || (reinterpret_cast<uintptr_t>(a.pro_b) & 15))) return 0;  // s_load_dwordx4
    if ((reinterpret_cast<uintptr_t>(a.wwino4) & 15) != 0) return 0;
    // float4 epilogue accesses
    if ((a.obs & 3) || (a.res && (a.rbs & 3)) || (a.aux && (a.abs_ & 3))) return 0;
    if ((reinterpret_cast<uintptr_t>(a.out) & 15) || (reinterpret_cast<uintptr_t>(a.res) & 15) || (reinterpret_cast<uintptr_t>(a.aux) & 15)) return 0;
    return (long long)a.tiles_x * ((a.Hout + TH - 1) / TH) * a.ncob;
}

int launch_conv_wino4(const ConvArgs& a, int mode, hipStream_t st) {
#ifdef W4_ONLY  // register-pressure experiments: one instantiation
    return launch_rag<IDIFF_CONV_NORMAL, 1, false>(a, st);
#else
    if (mode == IDIFF_CONV_UPSAMPLE2) return launch<IDIFF_CONV_UPSAMPLE2, 1>(a, st);
    if (a.pro_a) return launch<IDIFF_CONV_NORMAL, 2>(a, st);
    if (a.src1) return launch<IDIFF_CONV_NORMAL, 3>(a, st);
    return launch<IDIFF_CONV_NORMAL, 1>(a, st);
#endif
}

}  // namespace idiff_detail

extern "C" int idiff_pack_conv_weight_wino4(const float* w, float* wwino4, int Cout, int Cin, int transpose, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(w && wwino4 && Cout > 0 && Cin > 0, "pack_conv_weight_wino4: bad args");
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    IDIFF_CHECK_ARG(Co % 16 == 0 && Ci % 8 == 0, "pack_conv_weight_wino4: needs conv Cout %% 16 == 0 and Cin %% 8 == 0 (got %d, %d)", Co, Ci);
    if (Co % 64) {  // partial last 64-channel block: its unused rows must read as zero
        hipError_t e = hipMemsetAsync(wwino4, 0, (size_t)36 * Ci * ((Co + 63) / 64) * 64 * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "pack_conv_weight_wino4: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    const long long n = (long long)Cout * Cin;
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wwino4, Cout, Cin, transpose);
    IDIFF_CHECK_LAUNCH("pack_conv_weight_wino4");
    return IDIFF_OK;
}
