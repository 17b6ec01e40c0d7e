// Winograd F(2x2,3x3) convolution for gfx950 on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   (Lavin & Gray's minimal filtering form): the 3x3 taps become 16
//   independent [Cout x Cin] x [Cin x tiles] products, one per Winograd-domain position xi -- 2.25x fewer matrix-core
//   flops than the direct implicit GEMM in conv_igemm.hip.  Numerics: fp32 throughout; the transforms only add /
//   subtract (the 0.5 factors live in the pre-transformed weights), so the result differs from the direct kernel by
//   a few ulps of reassociation, the same class of difference as cuDNN's fp32 Winograd algorithms.
//
//   Persistent workgroups of 512 threads (8 waves, 2 per SIMD), one per CU; an item = one 8x32-pixel output patch
//   (4x16 tiles of 2x2) x 64 output channels of one sample (a partial last channel block when Cout % 64 = 16..48; a partial
//   patch at the right / bottom border when the image is not a multiple of 8x32 -- the reference's native 224 and its lower
//   levels 112 / 56: loads outside the image read zero, stores and GroupNorm partials of outside tiles are masked).
//   Per chunk of 8 input channels:
//     R  [8][10x34 (+pad)]            activated, zero-padded input patch with halo           (LDS, double buffer)
//     V  [16 xi][4 tile-rows][4 k][16 tiles][2]   B^T d B of that patch                      (LDS, double buffer)
//     U  [16 xi][4 co-blocks][4 k][16 co][2]      pre-transformed weights, copied verbatim   (LDS, double buffer)
//   wave (ch, tb) owns 32 output channels x the 16 tiles of tile-row tb x all 16 xi = 128 accumulator registers, so
//   the output transform A^T m A is purely in-lane.  ONE barrier per chunk: while the 16 positions of chunk c run on
//   the matrix cores, the raw patch of chunk c+2 is staged, the weights of chunk c+1 copied, the patch of chunk c+1
//   transformed and the loads of chunks c+3 / c+2 issued -- one slice per position, fenced with sched_barrier.
//   Every LDS access of the MFMA phase is a unit-stride ds_read_b64 (512 contiguous bytes per wave).  The f32 MFMA
//   shares the vector ALU, so the loop is kept nearly VALU-free (raw buffer loads, immediate LDS offsets, uniform
//   transform roles); see DESIGN.md 5a.
//
//   Same fused gather (virtual concat, nearest x2 upsample, GroupNorm/FiLM affine + SiLU prologue) and the same
//   epilogue contract (bias, residual, per-(b,c) vector, "+silu(a*aux+b)", GroupNorm partials per 8x32 patch) as
//   conv_igemm.hip; the two kernels are interchangeable behind idiff_conv2d_fwd.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

using idiff_detail::ConvArgs;

namespace {

constexpr int CK = 8;
constexpr int TW = 32, TH = 8;
constexpr int RS = TW + 2;         // 34
constexpr int TRH = TH + 2;        // 10
constexpr int PS = TRH * RS;       // 340
constexpr int PSP = 352;           // padded channel stride of R: ci and ci+1 land 32 banks apart
constexpr int NT = 512;
constexpr int R_FLOATS = 6 * NT;   // 8*352 = 2816 used; rounded up so every thread stages exactly 6 elements, unmasked
constexpr int V_FLOATS = 16 * 4 * 4 * 16 * 2;  // 8192
constexpr int U_FLOATS = V_FLOATS;
constexpr int NL = R_FLOATS / NT;              // 6 gathered elements per thread per chunk (element index = R index)
constexpr int NU = U_FLOATS / 4 / NT;          // 4 float4 of weights per thread per chunk

typedef float floatx2 __attribute__((ext_vector_type(2)));

// sum over each 16-lane row by DPP prefix adds (row_shr 1, 2, 4, 8; zeros shift in): lane 15 of the row ends with the total
template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
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
#define TRACE_INIT long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define TRACE_MARK(k)                                      \
    tr_t[k] = __builtin_readcyclecounter();                \
    if (k > 0) tr_acc[k - 1] += tr_t[k] - tr_t[k - 1];
#define TRACE_FINI                                                                  \
    if (tid == 0) {                                                                 \
        for (int q_ = 0; q_ < 7; ++q_) atomicAdd((unsigned long long*)trace + q_, (unsigned long long)tr_acc[q_]); \
    }
#else
#define TRACE_PARAM
#define TRACE_INIT
#define TRACE_MARK(k)
#define TRACE_FINI
#endif

#ifndef IDIFF_WINO_PD
#define IDIFF_WINO_PD 1  // operand prefetch distance of the MFMA loop, in Winograd positions
#endif

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
//
// Persistent: the grid is one workgroup per CU (the 156 KB of LDS allow one anyway); each walks a strided sequence
// of (sample, patch, channel-block) items, channel block fastest.  The first global loads of a workgroup's next item
// are issued before the last chunk of the current one.
// RAG: the image is not a multiple of the 8x32 patch (partial patches at the right / bottom border are masked)
template <int MODE, int SPEC, bool RAG>
__global__ __launch_bounds__(NT) void conv_wino_kernel(const ConvArgs a TRACE_PARAM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Rb = smem;                   // [2][R_FLOATS]
    float* const Vb = smem + 2 * R_FLOATS;
    float* const Ub = Vb + 2 * V_FLOATS;
    float* const econst = Ub + 2 * U_FLOATS;  // [4][64] bias, vec, aux_a, aux_b of the item's 64 output channels
    float* const protab = econst + 256;       // [2 item parities][2][C0r] (SPEC 2)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: role-dependent addresses stay on the SALU
    const int j = lane & 15;   // tile column (B operand / C column) and co row within a 16-block (A operand)
    const int k4 = lane >> 4;  // k index within a group of 4 (operands) / row group of the C layout
    const int ch = wave & 1;   // co half of the MFMA role; u-pair of the transform role
    const int tb = wave >> 1;  // tile row (both roles)
    const int HWin = a.Hin * a.Win;
    const int nchunks = a.Cin / CK;

    // ---- per-thread gather descriptors ---------------------------------------------------------------------------------
    // Thread stages R[tid + i*512]: the R index itself enumerates (ci, row, col) with the padded channel stride, so the
    // LDS writes are linear and unmasked.  An element's load offset is -1 for pad slots and for elements outside the image,
    // which the raw buffer load answers with 0.0.
    //   whole patches (RAG = false): gconst = byte offset relative to the patch origin (halo corner), constant for the whole
    //     kernel; eflags = 4 bits per element: on the top / bottom / left / right halo edge.  Per item the edges that fall
    //     outside the image turn their elements' offsets into -1.
    //   partial patches (RAG = true): the offset inside the SAMPLE is decoded afresh for every item from the element's
    //     (ci, r, c) -- a few dozen integer ops per item -- and bounds-checked against the image.
    auto slot_geometry = [&](int t, int i, int& ci, int& r, int& c) {
        const int e = t + i * NT;
        ci = e / PSP;
        const int rem = e - ci * PSP;
        r = rem / RS;
        c = rem - r * RS;
        return ci < CK && rem < PS;
    };
    int gconst[NL];
    unsigned eflags = 0;
    if (!RAG) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int ci, r, c;
            const bool slot = slot_geometry(tid, i, ci, r, c);
            const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (((r - 1) >> 1) + 1) * a.Win + ((c - 1) >> 1) + 1 : r * a.Win + c;
            gconst[i] = slot ? (ci * HWin + sp) * 4 : -1;
            eflags |= ((r == 0 ? 1u : 0u) | (r == TRH - 1 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == RS - 1 ? 8u : 0u)) << (4 * i);
        }
    }
    // ---- per-item state ---------------------------------------------------------------------------------------------
    // Raw buffer loads: uniform base in the resource, chunk offset in an SGPR, per-lane byte offset in one VGPR -> no
    // per-load address arithmetic on the vector ALU (which the f32 MFMAs share); offset -1 fails the range check.
    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rs0, rs1, rsu;
    int goff[NL];
    int it_b = 0, it_tile = 0, it_co0 = 0, it_y0 = 0, it_x0 = 0;
    auto setup_item = [&](int item) {
        const int cob = item % a.ncob;
        it_tile = (item / a.ncob) % a.ntiles;
        it_b = item / (a.ncob * a.ntiles);
        it_co0 = cob * 64;
        it_y0 = (it_tile / a.tiles_x) * TH;
        it_x0 = (it_tile % a.tiles_x) * TW;
        rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino + (long long)cob * U_FLOATS), 0, 0x7fffffff, RSRC_FLAGS);
        if (!RAG) {
            // element (row 0, col 0) of the halo patch; may lie before the tensor for border patches (never dereferenced)
            const long long org = MODE == IDIFF_CONV_UPSAMPLE2 ? (long long)(it_y0 / 2 - 1) * a.Win + (it_x0 / 2 - 1)
                                                               : (long long)(it_y0 - 1) * a.Win + (it_x0 - 1);
            rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0 + (long long)it_b * a.bs0 + org), 0, 0x7fffffff, RSRC_FLAGS);
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SPEC == 3 ? a.src1 + (long long)it_b * a.bs1 + org : a.src0), 0, 0x7fffffff,
                                                    RSRC_FLAGS);
            const unsigned out_edges = (it_y0 == 0 ? 1u : 0u) | (it_y0 + TH == a.Hout ? 2u : 0u) | (it_x0 == 0 ? 4u : 0u) | (it_x0 + TW == a.Wout ? 8u : 0u);
#pragma unroll
            for (int i = 0; i < NL; ++i) goff[i] = ((eflags >> (4 * i)) & out_edges) ? -1 : gconst[i];
        } else {
            rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0 + (long long)it_b * a.bs0), 0, 0x7fffffff, RSRC_FLAGS);
            rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SPEC == 3 ? a.src1 + (long long)it_b * a.bs1 : a.src0), 0, 0x7fffffff, RSRC_FLAGS);
            int t = tid;
            asm volatile("" : "+v"(t));  // opaque: keeps the decode here, once per item, instead of hoisted out of the item loop and held
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                int ci, r, c;
                const bool slot = slot_geometry(t, i, ci, r, c);
                const int oy = it_y0 - 1 + r, ox = it_x0 - 1 + c;  // output-grid coordinates of the element
                const bool in = slot && (unsigned)oy < (unsigned)a.Hout && (unsigned)ox < (unsigned)a.Wout;
                const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
                goff[i] = in ? (ci * HWin + sp) * 4 : -1;
            }
        }
    };
    const int ustride_b = a.ncob * U_FLOATS * 4;  // bytes between chunks of one channel block (whole U < 2^31 bytes)

    float rin[NL], rin1[NL];
    const float* ptab = protab;  // this item's prologue table (parity buffer)
    floatx4 ru[NU];

    // ---- the pieces of one chunk's staging work; the main loop deals them out between the MFMA groups ------------
    auto load_raw = [&](float (&dst)[NL], int cc) {  // global -> registers
        const int cb = cc * CK;
        if (SPEC == 3 && cb >= a.C0v) {  // chunk-uniform: C0v % 8 == 0
            const int so = (cb - a.C0v) * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, goff[i], so, 0));
        } else {
            const int so = cb * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, goff[i], so, 0));
        }
    };
    auto load_u = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NU; ++i)
            ru[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, tid * 16, cc * ustride_b + i * NT * 16, 0));
    };
    // activation (GroupNorm/FiLM affine + SiLU of the producer) + zero padding + LDS write of staged element i
    auto stage_raw = [&](const float (&src)[NL], int i, int cc, int rbuf) {
        float x = src[i];
        if (SPEC == 2) {
            const int cil = (tid + i * NT) / PSP;
            const int chc = cc * CK + (cil < CK ? cil : 0);
            x = silu_fast(ptab[chc] * x + ptab[a.C0r + chc]);
        }
        Rb[rbuf * R_FLOATS + tid + i * NT] = (SPEC == 2 && goff[i] < 0) ? 0.f : x;  // padding is zero AFTER the activation
    };
    auto stage_u = [&](int i, int buf) { reinterpret_cast<floatx4*>(Ub + buf * U_FLOATS)[tid + i * NT] = ru[i]; };

    // input transform B^T d B of the patch in R -> V[buf].  Thread = (u-pair ch, tile row tb, k4, tile j), both
    // channels ci = k4, k4+4 of the chunk.
    // Roles are made arithmetically uniform (no per-role selects on the vector ALU): u-pair 0 reads patch rows
    // (p,q,r) = (0,1,2) of its tile, u-pair 1 reads them reversed, (3,2,1); then for both
    //     tA = p - r          -> u = 0          | -(d1 - d3) = -t[u=3]
    //     tB = q + s*r        -> u = 1 (s = +1) |   d2 - d1  =  t[u=2]   (s = -1)
    // so Winograd rows are kept in the order rho = (u0, u1, -u3, u2); idiff_pack_conv_weight_wino stores U in the
    // same order with row u3 negated (the product U.V is unchanged), and the output transform reads acc rows (0,1,3,2).
    float td[3][4];     // patch rows (p, q, r) in flight between tr_read and tr_compute
    float to[2][4][2];  // transformed values [row within pair][v][g]
    const float* const trP = Rb + k4 * PSP + (2 * tb + 3 * ch) * RS + 2 * j;
    const float* const trQ = Rb + k4 * PSP + (2 * tb + 1 + ch) * RS + 2 * j;
    const float* const trR = Rb + k4 * PSP + (2 * tb + 2 - ch) * RS + 2 * j;
    const float tsign = ch ? -1.f : 1.f;
    auto tr_read = [&](int g, int rbuf) {
        const int o = rbuf * R_FLOATS + 4 * g * PSP;
        const float* rows[3] = {trP + o, trQ + o, trR + o};
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
            to[uu][3][g] = t[uu][1] - t[uu][3];
        }
    };
    float* const vwbase = Vb + ch * 4096 + (tb * 4 + k4) * 32 + j * 2;  // Winograd position = 8*ch + 4*uu + v
    auto tr_write = [&](int uu, int buf) {
        float* const V = vwbase + buf * V_FLOATS;
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<floatx2*>(V + (uu * 4 + v) * 512) = floatx2{to[uu][v][0], to[uu][v][1]};
    };
    auto clampc = [&](int c) { return c < nchunks ? c : nchunks - 1; };

    // Item order: workgroup v (XCD-contiguous numbering) takes items v, v+G, v+2G, ...; neighbours on one XCD thus work
    // on neighbouring items at the same time -- the channel blocks of one patch, then the next patch -- and share the
    // patch (and its halo) through that XCD's L2 instead of each fetching it from HBM at a different time.
    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = (int)a.total_wg;
    if (first >= last) return;
    // Per-item constants (epilogue table: bias, vec, aux affine of the 64 output channels; prologue table: the sample's
    // GroupNorm/FiLM affine) are fetched one item ahead into one register each and land in LDS without anybody waiting
    // on a global load: the prologue table is double-buffered by item parity and written during the previous epilogue.
    float pre_e = 0.f, pre_p = 0.f;
    auto fetch_consts = [&]() {  // for the item setup_item() just selected
        if (tid < 256) {
            const int which = tid >> 6, co = it_co0 + (tid & 63);
            pre_e = 0.f;
            if (co < a.Cout) {  // Cout % 16 == 0: the last 64-channel block may be partial
                if (which == 0 && a.bias) pre_e = a.bias[co];
                if (which == 1 && a.vec) pre_e = a.vec[(long long)it_b * a.Cout + co];
                if (which == 2 && a.aux) pre_e = a.aux_a[(long long)it_b * a.Cout + co];
                if (which == 3 && a.aux) pre_e = a.aux_b[(long long)it_b * a.Cout + co];
            }
        }
        if (SPEC == 2 && tid < 2 * a.C0r) pre_p = (tid < a.C0r ? a.pro_a : a.pro_b - a.C0r)[(long long)it_b * a.C0r + tid];
    };
    setup_item(first);
    load_raw(rin, 0);
    load_raw(rin1, clampc(1));
    load_u(0);
    fetch_consts();
    if (SPEC == 2 && tid < 2 * a.C0r) protab[tid] = pre_p;  // parity 0; visible after the first item's top barrier
    int parity = 0;
    TRACE_INIT

    for (int item = first; item < last; item += G) {
        const int b = it_b, tile = it_tile, co0 = it_co0, y0 = it_y0, x0 = it_x0;  // the epilogue's view of this item
        TRACE_MARK(0)

        // ---- pipeline fill: V[0], U[0] hold chunk 0, R[1] chunk 1; raw(2) and U(1) are in registers ------------------
        __syncthreads();  // every wave is done with the previous item's LDS (last chunk's operands, stats scratch)
        ptab = protab + parity * 2 * a.C0r;
        {
            if (tid < 256) econst[tid] = pre_e;  // read in the epilogue only, many barriers from here
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rin, i, 0, 0);
#pragma unroll
            for (int i = 0; i < NU; ++i) stage_u(i, 0);
#pragma unroll
            for (int i = 0; i < NL; ++i) stage_raw(rin1, i, clampc(1), 1);
        }
        load_raw(rin, clampc(2));
        load_u(clampc(1));
        __syncthreads();
        tr_read(0, 0), tr_compute(0);
        tr_read(1, 0), tr_compute(1);
        tr_write(0, 0), tr_write(1, 0);

        floatx4 acc[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[xi][mb] = floatx4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        TRACE_MARK(1)

        // ---- main loop, ONE barrier per chunk.  Iteration c runs the 16 Winograd positions of chunk c (per position
        // one B read + two A reads (ds_read_b64) -> 4 MFMAs, operands requested PD positions ahead) and, dealt out one
        // slice per position and fenced with sched_barrier so each slice issues between MFMAs:
        //   stage raw(c+2) registers -> R[c&1],  stage U(c+1) registers -> U[(c+1)&1],
        //   global loads raw(c+3), U(c+2) -> registers,  transform R[(c+1)&1] (staged one iteration ago) -> V[(c+1)&1].
        // The slices of one iteration are mutually independent; every buffer written was last read one barrier ago.
        // (Left alone, hipcc hoists the operand reads and sinks the MFMAs across barriers, idling the matrix pipe.)
        constexpr int PD = IDIFF_WINO_PD;
        const int opoff = k4 * 32 + j * 2;
        auto chunk = [&](int cc, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;  // false: last chunk, nothing left to stage
            const int buf = cc & 1;
            const float* V = Vb + buf * V_FLOATS + tb * 128 + opoff;
            const float* U = Ub + buf * U_FLOATS + ch * 256 + opoff;
            floatx2 ob[PD + 1], oa0[PD + 1], oa1[PD + 1];
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                ob[q] = *reinterpret_cast<const floatx2*>(V + q * 512);
                oa0[q] = *reinterpret_cast<const floatx2*>(U + q * 512);
                oa1[q] = *reinterpret_cast<const floatx2*>(U + q * 512 + 128);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q + PD < 16) {
                    ob[(q + PD) % (PD + 1)] = *reinterpret_cast<const floatx2*>(V + (q + PD) * 512);
                    oa0[(q + PD) % (PD + 1)] = *reinterpret_cast<const floatx2*>(U + (q + PD) * 512);
                    oa1[(q + PD) % (PD + 1)] = *reinterpret_cast<const floatx2*>(U + (q + PD) * 512 + 128);
                }
                const floatx2 bv = ob[q % (PD + 1)], av0 = oa0[q % (PD + 1)], av1 = oa1[q % (PD + 1)];
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv.x, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv.x, acc[q][1], 0, 0, 0);
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv.y, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv.y, acc[q][1], 0, 0, 0);
                if (MORE) {
                    if (q < 6) stage_raw(rin, q, clampc(cc + 2), buf);
                    if (q >= 4 && q < 8) stage_u(q - 4, buf ^ 1);
                    if (q == 8) load_raw(rin, clampc(cc + 3));
                    if (q == 9) load_u(clampc(cc + 2));
                    if (q == 10) tr_read(0, buf ^ 1);
                    if (q == 11) tr_compute(0);
                    if (q == 12) tr_read(1, buf ^ 1);
                    if (q == 13) tr_compute(1);
                    if (q == 14) tr_write(0, buf ^ 1);
                    if (q == 15) tr_write(1, buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q == 15) __syncthreads();
                }
            }
        };
        for (int cc = 0; cc + 1 < nchunks; ++cc) chunk(cc, std::true_type{});
        // Nothing is staged in the last chunk, so the item state is free: switch it to the next item now (scalar
        // work, hidden under the MFMAs) and let its first two patches and weights travel during the last chunk and the epilogue.
        if (item + G < last) {
            setup_item(item + G);
            load_raw(rin, 0);
            load_raw(rin1, clampc(1));
            load_u(0);
            fetch_consts();
        }
        chunk(nchunks - 1, std::false_type{});
        parity ^= 1;
        if (SPEC == 2 && item + G < last && tid < 2 * a.C0r) protab[parity * 2 * a.C0r + tid] = pre_p;  // the other parity is idle
        TRACE_MARK(2)

        TRACE_MARK(3)
        // ---- epilogue: in-lane output transform A^T m A, then the conv_igemm epilogue contract -------------------
        // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r of each 16-row block
        const int HWo = a.Hout * a.Wout;
        // addresses = uniform 64-bit base (SGPRs: sample, wave's channel half and tile row) + per-step uniform offset +
        // ONE per-lane 32-bit offset (row group 4*k4 channels down, tile column 2*j across): no 64-bit vector address
        // arithmetic per step, and nothing loop-invariant for the compiler to hoist out of the item loop and spill
        const long long wave_org = (long long)(co0 + ch * 32) * HWo + (long long)(y0 + 2 * tb) * a.Wout + x0;
        float* const outb = a.out + (long long)b * a.obs + wave_org;
        const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
        const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
        const unsigned lane_off = (unsigned)(4 * k4) * (unsigned)HWo + 2u * j;
        // LDS side likewise: one base per table, the step index is an immediate offset
        const float* const ebase = econst + ch * 32 + 4 * k4;             // + mb*16 + r (+ 64 per table)
        float* const sbase = Rb + (tb * 64 + ch * 32 + 4 * k4) * 2;      // + (mb*16 + r)*2
        const bool want_stats = a.stats != nullptr;
        const bool has_res = a.res != nullptr, has_aux = a.aux != nullptr;
        // partial patches (image not a multiple of 8x32): H and W are even, so a 2x2 tile lies inside the image or outside it
        const bool inside = !RAG || ((y0 + 2 * tb < a.Hout) && (x0 + 2 * j < a.Wout));
        // No global load may sit between the stores of two steps: vmcnt counts loads and stores in order, so waiting
        // for such a load would wait for every store before it.  The per-channel constants therefore come from LDS,
        // and the residual / aux operands of step i+1 are requested BEFORE the stores of step i (s_waitcnt then only
        // covers the stores of step i-1).  One instantiation per (residual, aux) combination keeps the steps branch-free.
        // GroupNorm partials: 16-lane row sums by DPP prefix adds (lane 15 of each row holds the total), written to
        // the cross-wave scratch [4 tb][64 co][2] in R[0] -- R was last read (by the transform) a barrier ago, the
        // last chunk reads only U and V, and the barrier at the top of the next item protects its reuse.
        auto out_steps = [&](auto res_tag, auto aux_tag) {
            constexpr bool RES = decltype(res_tag)::value, AUX = decltype(aux_tag)::value;
            floatx2 nres[2], naux[2];
            auto fetch = [&](int i) {
                if (co0 + ch * 32 + (i >> 2) * 16 >= a.Cout) return;  // uniform
                if (!inside) return;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const long long so = (long long)((i >> 2) * 16 + (i & 3)) * HWo + dy * a.Wout;  // uniform
                    if (RES) nres[dy] = *reinterpret_cast<const floatx2*>(resb + so + lane_off);
                    if (AUX) naux[dy] = *reinterpret_cast<const floatx2*>(auxb + so + lane_off);
                }
            };
            fetch(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mb = i >> 2, r = i & 3;
                if (co0 + ch * 32 + mb * 16 >= a.Cout) continue;  // uniform: a 16-channel block beyond a partial Cout
                const floatx2 cres[2] = {nres[0], nres[1]}, caux[2] = {naux[0], naux[1]};
                if (i + 1 < 8) fetch(i + 1);
                float z[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rho = u < 2 ? u : 5 - u;  // accumulator rows are stored in the order (u0, u1, u3, u2)
                    const float m0 = acc[rho * 4 + 0][mb][r], m1 = acc[rho * 4 + 1][mb][r], m2 = acc[rho * 4 + 2][mb][r], m3 = acc[rho * 4 + 3][mb][r];
                    z[u][0] = m0 + m1 + m2;
                    z[u][1] = m1 - m2 - m3;
                }
                const float bv = ebase[mb * 16 + r];
                float y[2][2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    y[0][x] = z[0][x] + z[1][x] + z[2][x] + bv;
                    y[1][x] = z[1][x] - z[2][x] - z[3][x] + bv;
                }
                if (want_stats) {
                    float ssum = (y[0][0] + y[0][1]) + (y[1][0] + y[1][1]);
                    float ssq = (y[0][0] * y[0][0] + y[0][1] * y[0][1]) + (y[1][0] * y[1][0] + y[1][1] * y[1][1]);
                    if (!inside) ssum = 0.f, ssq = 0.f;
                    ssum = row_sum16(ssum);
                    ssq = row_sum16(ssq);
                    if (j == 15) *reinterpret_cast<floatx2*>(sbase + (mb * 16 + r) * 2) = floatx2{ssum, ssq};
                }
                const float add = ebase[64 + mb * 16 + r];
                float aa = 0.f, ab = 0.f;
                if (AUX) aa = ebase[128 + mb * 16 + r], ab = ebase[192 + mb * 16 + r];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    floatx2 v = floatx2{y[dy][0] + add, y[dy][1] + add};
                    if (RES) v.x += cres[dy].x, v.y += cres[dy].y;
                    if (AUX) v.x += silu_fast(aa * caux[dy].x + ab), v.y += silu_fast(aa * caux[dy].y + ab);
                    if (inside) *reinterpret_cast<floatx2*>(outb + ((long long)(mb * 16 + r) * HWo + dy * a.Wout) + lane_off) = v;
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: interleaving them only buys register pressure
            }
        };
        if (has_res) {
            if (has_aux) out_steps(std::true_type{}, std::true_type{});
            else out_steps(std::true_type{}, std::false_type{});
        } else {
            if (has_aux) out_steps(std::false_type{}, std::true_type{});
            else out_steps(std::false_type{}, std::false_type{});
        }
        TRACE_MARK(4)
        TRACE_MARK(5)
        if (want_stats) {
            __syncthreads();
            TRACE_MARK(6)
            if (tid < 128) {
                const float t = (Rb[tid] + Rb[128 + tid]) + (Rb[256 + tid] + Rb[384 + tid]);
                const int col = tid >> 1, w = tid & 1;
                if (co0 + col < a.Cout) a.stats[(((long long)b * a.ntiles + tile) * a.Cout + co0 + col) * 2 + w] = t;
            }
        }
        TRACE_MARK(7)
    }
    TRACE_FINI
}

template <int MODE, int SPEC, bool RAG>
int launch_rag(const ConvArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)2 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 256 + (SPEC == 2 ? 4 * (size_t)a.C0r : 0)) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d(winograd): LDS budget exceeded (%zu bytes)", lds);
    static idiff_dyn_lds_cache lds_cache;
    auto kern = conv_wino_kernel<MODE, SPEC, RAG>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd): cannot query the CU count");
        num_cu = n;
    }
    const int total = (int)a.total_wg;
    const int per = (total + num_cu - 1) / num_cu;          // items per workgroup
    const int grid = (total + per - 1) / per;               // <= one workgroup per CU, none empty, strided item order
#ifdef IDIFF_WINO_TRACE
    static long long* tr = nullptr;
    if (!tr) (void)hipMalloc(&tr, 8 * sizeof(long long));
    (void)hipMemsetAsync(tr, 0, 8 * sizeof(long long), st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, tr);
    long long h[8];
    (void)hipMemcpyAsync(h, tr, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[wino trace] Cin=%d Cout=%d H=%d items=%d per=%d | fill %lld  loop %lld  prefetch %lld  outsteps %lld  butterfly %lld  barrier %lld  statstore %lld (cycles/item, wave 0)\n", a.Cin,
            a.Cout, a.Hout, total, per, h[0] / total, h[1] / total, h[2] / total, h[3] / total, h[4] / total, h[5] / total, h[6] / total);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
#endif
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd)");
    return IDIFF_OK;
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, hipStream_t st) {
    if (a.Hout % TH || a.Wout % TW) return launch_rag<MODE, SPEC, true>(a, st);
    return launch_rag<MODE, SPEC, false>(a, st);
}

// U = G g G^T for one (co, ci); G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int transpose) {
    // conv seen by the kernel: Co x Ci (swapped when transpose)
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    const int ncob = (Co + 63) / 64;         // the last block may be partial: its missing rows stay zero (caller clears)
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
        for (int rho = 0; rho < 4; ++rho) {  // stored row order (u0, u1, -u3, u2): see the kernel's input transform
            const int u = rho < 2 ? rho : 5 - rho;
            const float sg = rho == 2 ? -1.f : 1.f;
            const float v0 = t[u][0], v1 = 0.5f * (t[u][0] + t[u][1] + t[u][2]), v2 = 0.5f * (t[u][0] - t[u][1] + t[u][2]), v3 = t[u][2];
            dst[(rho * 4 + 0) * 512] = sg * v0;
            dst[(rho * 4 + 1) * 512] = sg * v1;
            dst[(rho * 4 + 2) * 512] = sg * v2;
            dst[(rho * 4 + 3) * 512] = sg * v3;
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
    // any even image size from 24 columns up (there the GroupNorm-partial tiling of idiff_conv2d_num_tiles is the 8x32 patch for
    // both kernels); partial patches at the right / bottom border are masked
    if (a.Cout % 16 || a.Cin % CK || a.C0v % CK || (a.Hout & 1) || (a.Wout & 1) || a.Wout < 24) return false;
    if ((long long)a.Cin * a.Hin * a.Win * 4 >= (1ll << 31)) return false;  // 32-bit byte offsets inside a sample
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.pro_a || a.src1)) return false;
    if (a.pro_a && a.src1) return false;
    if (a.pro_a && ((size_t)2 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 256 + 4 * (size_t)a.C0r) * sizeof(float) > 160 * 1024) return false;  // LDS
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
    IDIFF_CHECK_ARG(Co % 16 == 0 && Ci % 8 == 0, "pack_conv_weight_wino: needs conv Cout %% 16 == 0 and Cin %% 8 == 0 (got %d, %d)", Co, Ci);
    if (Co % 64) {  // partial last 64-channel block: its unused rows must read as zero
        hipError_t e = hipMemsetAsync(wwino, 0, (size_t)16 * Ci * ((Co + 63) / 64) * 64 * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "pack_conv_weight_wino: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    const long long n = (long long)Cout * Cin;
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wwino, Cout, Cin, transpose);
    IDIFF_CHECK_LAUNCH("pack_conv_weight_wino");
    return IDIFF_OK;
}
