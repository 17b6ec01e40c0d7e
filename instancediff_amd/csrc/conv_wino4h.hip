// Winograd F(4x4,3x3) convolution, half-patch items with the weights streamed straight into the A operand (gfx950, f32 matrix
// cores, v_mfma_f32_16x16x4_f32).  Same algorithm, weight image (idiff_pack_conv_weight_wino4), fused gather and epilogue contract
// as conv_wino4.hip; different shape of the work:
//
//   conv_wino4.hip   one 512-thread workgroup per CU, item = 16x32 pixels x 64 channels, weights copied to LDS per 4-channel chunk.
//                    Fill (first patches of an item: HBM latency), epilogue (output transform, statistics, 128 KB of stores) and the
//                    drain of those stores run with the matrix pipe idle: 31 % of an item at Cin = 64 (profiles/r03).
//   this kernel      TWO independent 256-thread workgroups per CU (one wave of each per SIMD), item = 8x32 pixels x 64 channels.
//                    While one workgroup fills, stores or waits for memory, the other one's MFMAs keep the SIMD's matrix pipe
//                    busy -- the two are never in phase for long.  There is no LDS for a second copy of the weights (2 x 72 KB), and
//                    none is needed: every wave reads its A operands (16 channels x 4 k x 4 positions = one 16-byte load per lane
//                    and position quad) from the L2-resident weight image through a rolling window of six quads, refilled right
//                    behind the MFMAs that consumed them.  That also takes the weight copy (five ds_write_b128 per thread and
//                    chunk, 13 LDS cycles each) and half of the operand reads off the LDS pipe.
//
//   Per workgroup and chunk of 4 input channels:
//     R  [4][10 x 40 (34 used), stride 512]  activated, zero-padded input patch with halo (LDS, double buffer, 8 elements / thread)
//     V  [9 position quads][4 k][16 tiles][4]   B^T d B of that patch                      (LDS, double buffer)
//   MFMA role: wave cb owns 16 output channels x the 16 tiles of the 8x32 patch x all 36 positions = 144 accumulator registers
//   (in-lane output transform; a wave's GroupNorm partials are one cell of the 8x32 partials grid).  Transform role: lane = (ci,
//   tile), wave = Winograd rows (1,2) | (3,4) | 0 | 5 (two heavy and two light waves).  ONE barrier per chunk; staging of chunk c+2, transform of chunk c+1 and the
//   loads of chunk c+4 are dealt out between the MFMAs of chunk c, as in conv_wino4.hip.
//
//   In the in-order vmcnt queue the six raw patch loads of a chunk (HBM) are issued between two A loads, so an A quad is never
//   waited for behind a raw load younger than six quad slots; the first window of an item is requested after its predecessor's
//   epilogue (it would otherwise sit in 24 registers across it) and the workgroup's partner covers the drain of those stores.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

using idiff_detail::ConvArgs;

namespace {

constexpr int CK = 4;
constexpr int TW = 32, TH = 8;
constexpr int RCOLS = TW + 2;   // 34 columns used
constexpr int RS = 40;          // row stride of R: rows 16-byte aligned, 4*RS = 32 (mod 64) banks
constexpr int TRH = TH + 2;     // 10
constexpr int PS = TRH * RS;    // 400
constexpr int PSP = 512;        // channel stride of R: 0 (mod 64) banks -> the transform's ds_read_b128 are conflict-free; = 2 NT, so
                                // element i of EVERY thread belongs to channel i / 2 of the chunk (a compile-time constant)
constexpr int NT = 256;
constexpr int NL = 8;           // gathered elements per thread per chunk (element index = R index): 4 * 512 = 8 * 256
constexpr int R_FLOATS = NL * NT;
constexpr int V_FLOATS = 9 * 64 * 4;       // 2304
constexpr int U_FLOATS = 9 * 4 * 64 * 4;   // one (chunk, 64-channel block) of the global weight image
constexpr int AW = 6;                      // A-operand window, in position quads

typedef float floatx2 __attribute__((ext_vector_type(2)));

// position numbering of idiff_pack_conv_weight_wino4 (conv_wino4.hip): row u = one 16-byte piece in quad PF(u) + one 8-byte piece
// in half of quad PH(u)
__host__ __device__ constexpr int PF(int u) { return (3 * u + 1) / 2; }
__host__ __device__ constexpr int PH(int u) { return 1 + 3 * (u / 2); }
__host__ __device__ constexpr int pos(int u, int v) { return v < 4 ? 4 * PF(u) + v : 4 * PH(u) + 2 * (u & 1) + (v - 4); }

struct Geo4h {
    int np;         // 8x32 patches per sample
    int total;      // items = B * np * ncob
};

template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v) {  // lane 15 of each 16-lane row ends with the row's total
    v += dpp_row_shr<0x111>(v);
    v += dpp_row_shr<0x112>(v);
    v += dpp_row_shr<0x114>(v);
    v += dpp_row_shr<0x118>(v);
    return v;
}

// a wave-uniform pointer pinned to scalar registers (a resource base left in VGPRs costs a waterfall loop per buffer load)
__device__ __forceinline__ const float* scalar_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void bt6(const float (&x)[6], float (&o)[6]) {  // one 6-point input transform B^T x
    o[0] = __builtin_fmaf(4.f, x[0], __builtin_fmaf(-5.f, x[2], x[4]));
    const float p = __builtin_fmaf(-4.f, x[2], x[4]), q = __builtin_fmaf(-4.f, x[1], x[3]);
    o[1] = p + q;
    o[2] = p - q;
    const float c = x[4] - x[2], e = x[3] - x[1];
    o[3] = __builtin_fmaf(2.f, e, c);
    o[4] = __builtin_fmaf(-2.f, e, c);
    o[5] = __builtin_fmaf(4.f, x[1], __builtin_fmaf(-5.f, x[3], x[5]));
}
__device__ __forceinline__ void at6(const float x0, const float x1, const float x2, const float x3, const float x4, const float x5, float (&o)[4]) {
    const float s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;  // one 6 -> 4 output transform A^T x
    o[0] = (x0 + s1) + s2;
    o[1] = __builtin_fmaf(2.f, d2, d1);
    o[2] = __builtin_fmaf(4.f, s2, s1);
    o[3] = __builtin_fmaf(8.f, d2, d1) + x5;
}

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
// RAG: the image is not a multiple of the 8x32 patch (tiles outside it are masked)
template <int MODE, int SPEC, bool RAG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4h_kernel(const ConvArgs a, const Geo4h g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Rb = smem;                   // [2][R_FLOATS]
    float* const Vb = smem + 2 * R_FLOATS;    // [2][V_FLOATS]
    float* const econst = Vb + 2 * V_FLOATS;  // [4][64] bias, vec, aux_a, aux_b of the item's 64 output channels
    int* const gtab = reinterpret_cast<int*>(econst + 256);  // [NL][NT] gather byte offsets of the current item (thread-private)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k4 = lane >> 4;  // k index of the MFMA operands (lane & 15: tile of the B operand / channel of the A operand)
    const int cb = wave;       // MFMA role: 16-channel block
    const int HWin = a.Hin * a.Win;
    const int nchunks = a.Cin / CK;  // even (Cin % 8 == 0)

    // ---- per-thread gather descriptors (as conv_wino4.hip: element index = R index, offset -1 = padding / outside -> 0.0) -------
    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rs0, rs1, rsu;
    unsigned omask = 0;  // bit i: element i is padding / outside the image
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
    auto advance_item = [&]() {  // mixed-radix counter with the constant increment gridDim.x
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
        asm volatile("" : "+v"(t));  // opaque: the decode stays here, once per item, instead of hoisted and held in registers
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
    // SPEC 2: the GroupNorm/FiLM affine of a chunk's four input channels through the scalar cache (element i <-> channel i / 2)
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
    auto stage_raw = [&](const float (&src)[NL], int i, const Pro& pro, int rbuf) {
        float x = src[i];
        if (SPEC == 2) {
            const int ch = i >> 1;
            const float pa = ch == 0 ? pro.a.x : ch == 1 ? pro.a.y : ch == 2 ? pro.a.z : pro.a.w;
            const float pb = ch == 0 ? pro.b.x : ch == 1 ? pro.b.y : ch == 2 ? pro.b.z : pro.b.w;
            x = silu_fast(pa * x + pb);
        }
        Rb[rbuf * R_FLOATS + tid + i * NT] = (SPEC == 2 && ((omask >> i) & 1u)) ? 0.f : x;  // padding is zero AFTER the activation
    };
    // A operands: quad q of a chunk of parity PAR lives in oa[(q + 3 * PAR) % AW] (9 quads per chunk, window 6: period two chunks)
    floatx4 oa[AW];
    const int a_voff = (cb * 256 + lane * 4) * 4;
    auto load_a = [&](int reg, int cc, int q) {
        oa[reg] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, a_voff, cc * ustride_b + q * 4096, 0));
    };

    // ---- input transform B^T d B of R[rbuf] -> V[buf].  lane = (ci = k4, tile (tyl, tx)); wave role trw:
    //   0 / 1 (heavy): Winograd rows (1,2) / (3,4):  X = d4 + al*d2, Y = d3 + al*d1, rows X +- be*Y
    //   2 / 3 (light): row 0 (from d0, d2, d4) / row 5 (from d1, d3, d5): 4*dA - 5*dB + dC
    const int trw = wave;  // (rotating the roles between the two workgroups of a CU, by any bit of blockIdx.x, measured +-0)
    const bool heavy = trw < 2;  // wave class: uniform; the main loop is instantiated once per class, branch-free
    const int trole = trw & 1;
    const int tx = lane & 7, tyl = (lane >> 3) & 1;
    const float al = trole ? -1.f : -4.f, be = trole ? 2.f : 1.f;
    const float* const trbase = Rb + k4 * PSP + (4 * tyl + (heavy ? 1 : trole)) * RS + 4 * tx;
    const int ufirst = heavy ? 1 + 2 * trole : 5 * trole;  // the role's (first) Winograd row
    float* const vwbase = Vb + lane * 4;
    float ta[6], tb[6], tlo[6], thi[6];
    auto rd_row = [&](const float* p, float (&d)[6]) {
        const floatx4 lo = *reinterpret_cast<const floatx4*>(p);
        const floatx2 hi = *reinterpret_cast<const floatx2*>(p + 4);
        d[0] = lo.x, d[1] = lo.y, d[2] = lo.z, d[3] = lo.w, d[4] = hi.x, d[5] = hi.y;
    };
    auto tr_piece = [&](int piece, int rbuf, int buf, auto hv_tag) {
        constexpr bool hv = decltype(hv_tag)::value;
        const float* p = trbase + rbuf * R_FLOATS;
        if (piece == 0) {
            if (hv) rd_row(p + 1 * RS, ta), rd_row(p + 3 * RS, tb);  // d2, d4
            else rd_row(p, ta), rd_row(p + 2 * RS, tb);              // dA, dB
        } else if (piece == 1) {
            if (hv) {
#pragma unroll
                for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(al, ta[c], tb[c]);  // X
                rd_row(p, ta), rd_row(p + 2 * RS, tb);                                    // d1, d3
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(4.f, ta[c], -5.f * tb[c]);  // P
                rd_row(p + 4 * RS, ta);                                                        // dC
            }
        } else if (piece == 2) {
            if (hv) {
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
            if (which == 1 && !hv) return;
            float o[6];
            bt6(which ? thi : tlo, o);
            const int u = ufirst + which;
            float* const V = vwbase + buf * V_FLOATS;
            *reinterpret_cast<floatx4*>(V + PF(u) * 256) = floatx4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<floatx2*>(V + PH(u) * 256 + 2 * (u & 1)) = floatx2{o[4], o[5]};
        }
    };
    auto clampc = [&](int c) { return c < nchunks ? c : nchunks - 1; };

    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = g.total;
    if (first >= last) return;  // (the launchers size the grid so that it cannot happen)
    float pre_e = 0.f;
    auto fetch_consts = [&]() {
        const int which = tid >> 6, co = it_co0 + (tid & 63);
        pre_e = 0.f;
        if (co < a.Cout) {
            if (which == 0 && a.bias) pre_e = a.bias[co];
            if (which == 1 && a.vec) pre_e = a.vec[(long long)it_b * a.Cout + co];
            if (which == 2 && a.aux) pre_e = a.aux_a[(long long)it_b * a.Cout + co];
            if (which == 3 && a.aux) pre_e = a.aux_b[(long long)it_b * a.Cout + co];
        }
    };
    decode_first(first);
    setup_item();
    load_raw(rinA, 0);
    load_raw(rinB, 1);
    fetch_consts();

    for (int item = first; item < last; item += G) {
        const int b = it_b, co0 = it_co0, y0 = it_y0, x0 = it_x0;  // the epilogue's view of this item

        // ---- pipeline fill: V[0] holds chunk 0, R[1] chunk 1; raw(2), raw(3) are in registers ------------------------------------
        __syncthreads();  // every wave is done with the previous item's LDS
        const Pro pro0 = load_pro(b, 0), pro1 = load_pro(b, 1);
        econst[tid] = pre_e;
        floatx4 acc[36];
        auto run_item = [&](auto hv) {
#pragma unroll
            for (int q = 0; q < AW; ++q) load_a(q, 0, q);  // the item's first window; back long before the fill reaches an MFMA
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rinA, i, pro0, 0);
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rinB, i, pro1, 1);
            load_raw(rinA, clampc(2));
            load_raw(rinB, clampc(3));
            __syncthreads();
#pragma unroll
            for (int piece = 0; piece < 5; ++piece) tr_piece(piece, 0, 0, hv);
#pragma unroll
            for (int p = 0; p < 36; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();

            // ---- main loop, ONE barrier per chunk.  Iteration c runs the 9 position quads of chunk c and, one slice per quad:
            //   the A quad six places ahead;  stage raw(c+2) registers -> R[c&1], then load raw(c+4) into them;
            //   transform R[(c+1)&1] (staged one iteration ago) -> V[(c+1)&1].
            const int opoff = lane * 4;
            floatx4 ob[2];
            ob[0] = *reinterpret_cast<const floatx4*>(Vb + opoff);  // quad 0 of chunk 0
            const bool have_next = item + G < last;
            auto chunk = [&](int cc, auto par_tag, auto more_tag) {
                constexpr int PAR = decltype(par_tag)::value;      // cc & 1: LDS buffers and the raw register set
                constexpr bool MORE = decltype(more_tag)::value;   // false: last chunk, nothing left to stage
                const float* V = Vb + PAR * V_FLOATS + opoff;
                const float* Vn = Vb + (PAR ^ 1) * V_FLOATS + opoff;
                float(&rin)[NL] = PAR ? rinB : rinA;
                Pro pro;
                if (MORE) pro = load_pro(b, clampc(cc + 2));  // the affine of the chunk staged below
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (q + 1 < 9) {
                        ob[(q + 1 + PAR) & 1] = *reinterpret_cast<const floatx4*>(V + (q + 1) * 256);
                    } else if (MORE) {
                        load_raw(rin, clampc(cc + 4));
                        __builtin_amdgcn_sched_barrier(0);
                        __syncthreads();
                        ob[PAR ^ 1] = *reinterpret_cast<const floatx4*>(Vn);
                    }
                    const floatx4 bv = ob[(q + PAR) & 1], av = oa[(q + 3 * PAR) % AW];
                    acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[4 * q + 0], 0, 0, 0);
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[4 * q + 1], 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[4 * q + 2], 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[4 * q + 3], 0, 0, 0);
                    if (MORE) {
                        if (q + AW < 9) load_a((q + 3 * PAR) % AW, cc, q + AW);
                        else load_a((q + 3 * PAR) % AW, cc + 1, q + AW - 9);
                        if (q < 5) tr_piece(q, PAR ^ 1, PAR ^ 1, hv);
                        if (q == 5) stage_raw(rin, 0, pro, PAR), stage_raw(rin, 1, pro, PAR), stage_raw(rin, 2, pro, PAR);
                        if (q == 6) stage_raw(rin, 3, pro, PAR), stage_raw(rin, 4, pro, PAR), stage_raw(rin, 5, pro, PAR);
                        if (q == 7) stage_raw(rin, 6, pro, PAR), stage_raw(rin, 7, pro, PAR);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        // last chunk: quads 6..8 still come through this item's resource; then the item state switches to the next
                        // item and its first patches travel during the rest of the chunk and the epilogue
                        if (q + AW < 9) load_a((q + 3 * PAR) % AW, cc, q + AW);
                        if (have_next) {
                            if (q == 2) advance_item(), setup_item();
                            if (q == 3) load_raw(rinA, 0);
                            if (q == 4) load_raw(rinB, 1);
                            if (q == 5) fetch_consts();
                        }
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

        // ---- epilogue: in-lane output transform A^T m A, then the conv_igemm epilogue contract -----------------------------------
        // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r (channels) of the wave's 16-channel block
        if (co0 + cb * 16 < a.Cout) {  // uniform: a 16-channel block beyond a partial Cout has nothing to store
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));  // lane-derived constants recomputed here: hoisted out of the item loop they would be spilled
            const int j = lane_e & 15, k4e = lane_e >> 4;
            const int HWo = a.Hout * a.Wout;
            const long long wave_org = (long long)(co0 + cb * 16) * HWo + (long long)y0 * a.Wout + x0;
            float* const outb = a.out + (long long)b * a.obs + wave_org;
            const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
            const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
            const unsigned lane_off = (unsigned)(4 * k4e) * (unsigned)HWo + (unsigned)(4 * (j >> 3)) * (unsigned)a.Wout + 4u * (j & 7);
            const float* const ebase = econst + cb * 16 + 4 * k4e;
            const bool want_stats = a.stats != nullptr;
            const bool has_res = a.res != nullptr, has_aux = a.aux != nullptr;
            // partial patches: H and W are multiples of 4, so a 4x4 tile lies inside the image or outside it
            const bool inside = !RAG || ((y0 + 4 * (j >> 3) < a.Hout) && (x0 + 4 * (j & 7) < a.Wout));
            float* const stp = want_stats ? a.stats + (((long long)b * a.ntiles + (y0 >> 3) * a.tiles_x + (x0 >> 5)) * a.Cout + co0 + cb * 16 + 4 * k4e) * 2 : nullptr;
            // Phase 1: A^T along the Winograd columns v of every row u and channel r -- 144 accumulators shrink to 96 values
            float zz[6][4][4];  // [u][r][dx]
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    at6(acc[pos(u, 0)][r], acc[pos(u, 1)][r], acc[pos(u, 2)][r], acc[pos(u, 3)][r], acc[pos(u, 4)][r], acc[pos(u, 5)][r], zz[u][r]);
                    // pinned: left alone the optimiser sinks these sums into phase 2 and keeps the accumulators -- spilled -- until then
                    asm volatile("" : "+v"(zz[u][r][0]), "+v"(zz[u][r][1]), "+v"(zz[u][r][2]), "+v"(zz[u][r][3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // Phase 2, per channel r: A^T along u, bias, GroupNorm partials, then four row steps.  The residual / aux row of step s+1
            // is requested BEFORE the store of step s (vmcnt counts loads and stores in order).
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
                    if (j == 15) stp[2 * r] = ssum, stp[2 * r + 1] = ssq;
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
                    if (inside) *reinterpret_cast<floatx4*>(outb + ((long long)r * HWo + dy * a.Wout) + lane_off) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int MODE, int SPEC, bool RAG>
int launch_rag(const ConvArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)2 * R_FLOATS + 2 * V_FLOATS + 256 + NL * NT) * sizeof(float);  // 44 KB: two workgroups per CU
    static idiff_dyn_lds_cache lds_cache;
    auto kern = conv_wino4h_kernel<MODE, SPEC, RAG>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4h): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4h): cannot query the CU count");
        num_cu = n;
    }
    Geo4h g;
    g.np = a.tiles_x * ((a.Hout + TH - 1) / TH);
    const long long total = (long long)a.B * g.np * a.ncob;
    if (total >= (1ll << 31)) IDIFF_FAIL(IDIFF_E_BADARG, "conv2d(winograd4h): grid too large");
    g.total = (int)total;
    const int slots = 2 * num_cu;  // two co-resident workgroups per CU
    const int per = (g.total + slots - 1) / slots;
    const int grid = (g.total + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, g);
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd4h)");
    return IDIFF_OK;
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, hipStream_t st) {
    if (a.Hout % TH || a.Wout % TW) return launch_rag<MODE, SPEC, true>(a, st);
    return launch_rag<MODE, SPEC, false>(a, st);
}

}  // namespace

namespace idiff_detail {

// Same shape rules as conv_wino4_eligible (the weight image is the same); the caller has checked those.
int launch_conv_wino4h(const ConvArgs& a, int mode, hipStream_t st) {
    if (mode == IDIFF_CONV_UPSAMPLE2) return launch<IDIFF_CONV_UPSAMPLE2, 1>(a, st);
    if (a.pro_a) return launch<IDIFF_CONV_NORMAL, 2>(a, st);
    if (a.src1) return launch<IDIFF_CONV_NORMAL, 3>(a, st);
    return launch<IDIFF_CONV_NORMAL, 1>(a, st);
}

}  // namespace idiff_detail
