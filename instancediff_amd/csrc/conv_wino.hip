// Winograd F(2x2,3x3) convolution for gfx950 on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   (Lavin & Gray's minimal filtering form): the 3x3 taps become 16
//   independent [Cout x Cin] x [Cin x tiles] products, one per Winograd-domain position xi -- 2.25x fewer matrix-core
//   flops than the direct implicit GEMM in conv_igemm.hip.  Numerics: fp32 throughout; the transforms only add /
//   subtract (the 0.5 factors live in the pre-transformed weights), so the result differs from the direct kernel by
//   a few ulps of reassociation, the same class of difference as cuDNN's fp32 Winograd algorithms.
//
//   Persistent workgroups of 512 threads (8 waves, 2 per SIMD), one per CU; an item = one 8x32-pixel output patch
//   (4x16 tiles of 2x2) x 64 output channels of one sample (a partial last channel block when Cout % 64 = 16..48, a
//   partial patch at the right / bottom border when the image is not a multiple of 8x32).  Per chunk of 8 input channels:
//     R  [4 k][10 rows][34 cols][2 g]             activated, zero-padded input patch with halo, channels k and k+4
//                                                 interleaved                                     (LDS, double buffer)
//     V  [16 xi][4 tile-rows][4 k][16 tiles][2 g] B^T d B of that patch                           (LDS, double buffer)
//     U  [16 xi][4 co-blocks][4 k][16 co][2 g]    pre-transformed weights, copied verbatim        (LDS, double buffer)
//   wave (ch, tb) owns 32 output channels x the 16 tiles of tile-row tb x all 16 xi = 128 accumulator registers, so
//   the output transform A^T m A is purely in-lane.
//
//   ONE software pipeline runs through the whole item sequence of a workgroup, one barrier per chunk: while the 16
//   positions of chunk s run on the matrix cores, the patch of chunk s+1 is transformed and its weights travel global -> LDS
//   (buffer_load ... lds, no registers), the raw patch of chunk s+2 is staged into LDS and the global loads of chunk s+3
//   are issued -- whichever items those chunks belong to.  The epilogue of an item (output transform, GroupNorm partials, stores) sits between
//   its last chunk and the next item's first, whose operands are already in LDS: no pipeline refill per item.
//   The f32 MFMA shares the vector ALU, so everything around it is kept VALU-lean: raw buffer loads (uniform base in the
//   resource, one per-lane offset, -1 = zero padding), immediate LDS offsets, both transforms in packed fp32 math
//   (v_pk_add_f32 on channel pairs: the g interleave of R / V, row pairs of the accumulators); see DESIGN.md 5a.
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
constexpr int RS = TW + 2;              // 34 columns with halo
constexpr int TRH = TH + 2;             // 10 rows with halo
constexpr int RROW = RS * 2;            // 68 floats per patch row (2 interleaved channels)
constexpr int RPL = TRH * RROW;         // 680 floats per channel-pair plane
constexpr int R_USED = 4 * RPL;         // 2720 floats per chunk
constexpr int R_FLOATS = 2752;          // buffer stride (16-byte multiple)
constexpr int NT = 512;
constexpr int NL = 6;                   // gathered elements per thread per chunk (slot e = tid + i*512; slots >= 2720 unused)
constexpr int V_FLOATS = 16 * 4 * 4 * 16 * 2;  // 8192
constexpr int U_FLOATS = V_FLOATS;
constexpr int NU = U_FLOATS / 4 / NT;   // 4 float4 of weights per thread per chunk
constexpr int ECONST = 256;             // per-item epilogue table: bias, vec, aux_a, aux_b x 64 channels
constexpr int SCRATCH = 4 * 64 * 2;     // per-item GroupNorm partial scratch [4 tile rows][64 co][2]

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
__device__ __forceinline__ floatx2 lo2(floatx4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ floatx2 hi2(floatx4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// number of table parities: a table written when an item's first chunk is 3 steps from the matrix cores must outlive the
// epilogues of the items still ahead of it; with fewer than 3 chunks per item up to three items are in flight
__host__ __device__ inline int table_parities(int nchunks) { return nchunks >= 3 ? 2 : 4; }

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
template <int MODE, int SPEC>
__global__ __launch_bounds__(NT) void conv_wino_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nchunks = a.Cin / CK;
    const int NP = table_parities(nchunks);
    float* const Rb = smem;                        // [2][R_FLOATS]
    float* const Vb = Rb + 2 * R_FLOATS;           // [2][V_FLOATS]
    float* const Ub = Vb + 2 * V_FLOATS;           // [2][U_FLOATS]
    float* const scratch = Ub + 2 * U_FLOATS;      // [2][SCRATCH]
    float* const econst = scratch + 2 * SCRATCH;   // [NP][ECONST]
    float* const protab = econst + NP * ECONST;    // [NP][2][C0r]  (SPEC 2)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: role-dependent addresses stay on the SALU
    const int j = lane & 15;   // tile column (B operand / C column) and co row within a 16-block (A operand)
    const int k4 = lane >> 4;  // k index within a group of 4 (operands) / row group of the C layout
    const int ch = wave & 1;   // co half of the MFMA role; u-pair of the transform role
    const int tb = wave >> 1;  // tile row (both roles)
    const int HWin = a.Hin * a.Win;
    const int HWo = a.Hout * a.Wout;

    // ---- item sequence of this workgroup: items first, first+G, ... (XCD-contiguous numbering: neighbours on one XCD work on
    // neighbouring items at the same time -- the channel blocks of one patch, then the next patch -- and share the patch
    // through that XCD's L2) ------------------------------------------------------------------------------------------------
    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = (int)a.total_wg;
    if (first >= last) return;
    const int nitems = (last - first + G - 1) / G;
    const int total = nitems * nchunks;  // chunks ("steps") this workgroup runs through the matrix cores

    // ---- per-thread gather geometry: slot e = tid + i*512 holds element (channel ci = k + 4g, patch row r, patch column c).
    // Decoded afresh at every item entry (a few dozen integer ops per item) rather than kept in 18 registers for the whole kernel.
    auto slot_geometry = [&](int t, int i, int& ci, int& r, int& c) {
        const int e = t + i * NT;
        const int pl = e / RPL, rem = e - pl * RPL;
        r = rem / RROW;
        const int c2 = rem - r * RROW;
        ci = pl + 4 * (c2 & 1);
        c = c2 >> 1;
        return e < R_USED;
    };
    unsigned cisel = 0;  // 3 bits per slot: the element's channel inside the chunk (prologue table index, SPEC 2)
    if (SPEC == 2) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int ci, r, c;
            slot_geometry(tid, i, ci, r, c);
            cisel |= (unsigned)(ci & 7) << (3 * i);
        }
    }
    const bool slot5 = tid + 5 * NT < R_USED;  // the sixth slot exists for the first 160 threads only

    // ---- stage state -------------------------------------------------------------------------------------------------------
    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, 0x7fffffff, RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rs1 = rs0, rsu = rs0;
    int goff[NL];            // byte offsets of this thread's elements inside the sample (load stage's item); -1 = zero
    unsigned lmask = 0;      // bit i: element i lies inside the image (load stage's item)
    unsigned smask = 0;      // the same for the chunk being staged (one step behind)
    int cr = 0, ir = 0;      // load stage: local chunk, item ordinal (chunk s+3)
    int cu = 0, iu = 0;      // weight stage (chunk s+1): global -> LDS directly
    int cs = 0, is_ = 0;     // raw-staging stage (chunk s+2): channel index / table parity of the prologue
    int cm = 0, im = 0;      // matrix-core stage (chunk s)
    float pre_e = 0.f, pre_p = 0.f;
    float rin[NL];
    const int ustride_b = a.ncob * U_FLOATS * 4;  // bytes between chunks of one channel block (whole U < 2^31 bytes)

    auto decode = [&](int ord, int& b, int& tile, int& cob, int& y0, int& x0) {
        const int item = first + ord * G;
        cob = item % a.ncob;
        tile = (item / a.ncob) % a.ntiles;
        b = item / (a.ncob * a.ntiles);
        y0 = (tile / a.tiles_x) * TH;
        x0 = (tile % a.tiles_x) * TW;
    };
    // load stage enters item `ord`: resources, element offsets, and the item's constant tables on their way to registers
    auto enter_item = [&](int ord) {
        int b, tile, cob, y0, x0;
        decode(ord, b, tile, cob, y0, x0);
        rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0 + (long long)b * a.bs0), 0, 0x7fffffff, RSRC_FLAGS);
        if (SPEC == 3) rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 + (long long)b * a.bs1), 0, 0x7fffffff, RSRC_FLAGS);
        lmask = 0;
        int t = tid;
        asm volatile("" : "+v"(t));  // opaque: keeps the geometry decode here, once per item, instead of hoisted and held
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int ci, r, c;
            const bool slot = slot_geometry(t, i, ci, r, c);
            const int oy = y0 - 1 + r, ox = x0 - 1 + c;  // output-grid coordinates of the element
            const bool in = slot && (unsigned)oy < (unsigned)a.Hout && (unsigned)ox < (unsigned)a.Wout;
            const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
            goff[i] = in ? (ci * HWin + sp) * 4 : -1;
            lmask |= (in ? 1u : 0u) << i;
        }
        const int co0 = cob * 64;
        if (tid < 256) {
            const int which = tid >> 6, co = co0 + (tid & 63);
            pre_e = 0.f;
            if (co < a.Cout) {  // Cout % 16 == 0: the last 64-channel block may be partial
                if (which == 0 && a.bias) pre_e = a.bias[co];
                if (which == 1 && a.vec) pre_e = a.vec[(long long)b * a.Cout + co];
                if (which == 2 && a.aux) pre_e = a.aux_a[(long long)b * a.Cout + co];
                if (which == 3 && a.aux) pre_e = a.aux_b[(long long)b * a.Cout + co];
            }
        }
        if (SPEC == 2 && tid < 2 * a.C0r) pre_p = (tid < a.C0r ? a.pro_a : a.pro_b - a.C0r)[(long long)b * a.C0r + tid];
    };
    auto publish_tables = [&](int ord) {  // registers -> LDS tables of item `ord` (read from the next step on)
        const int par = ord % NP;
        if (tid < 256) econst[par * ECONST + tid] = pre_e;
        if (SPEC == 2 && tid < 2 * a.C0r) protab[par * 2 * a.C0r + tid] = pre_p;
    };
    auto load_raw = [&]() {  // chunk (ir, cr): global -> registers
        const int cb = cr * CK;
        if (SPEC == 3 && cb >= a.C0v) {  // chunk-uniform: C0v % 8 == 0
            const int so = (cb - a.C0v) * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) rin[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, goff[i], so, 0));
        } else {
            const int so = cb * HWin * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) rin[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, goff[i], so, 0));
        }
    };
    // weights of chunk (iu, cu): 32 KB copied verbatim, global -> LDS without passing through registers (buffer_load ... lds:
    // each lane's 16 bytes land at the wave's LDS base + 16*lane, so the destination is the linear image of the source)
    auto copy_u = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            float* dst = Ub + buf * U_FLOATS + (i * NT + wave * 64) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16,
                                                     tid * 16, cu * ustride_b + i * NT * 16, 0, 0);
        }
    };
    // activation (GroupNorm/FiLM affine + SiLU of the producer) + zero padding + LDS write of staged element i
    auto stage_raw = [&](int i, int rbuf) {
        float x = rin[i];
        if (SPEC == 2) {
            const float* ptab = protab + (is_ % NP) * 2 * a.C0r;
            const int chc = cs * CK + ((cisel >> (3 * i)) & 7);
            x = silu_fast(ptab[chc] * x + ptab[a.C0r + chc]);
            x = ((smask >> i) & 1u) ? x : 0.f;  // padding is zero AFTER the activation
        }
        if (i < NL - 1 || slot5) Rb[rbuf * R_FLOATS + tid + i * NT] = x;
    };

    // input transform B^T d B of the patch in R -> V.  Thread = (u-pair ch, tile row tb, k4, tile j), both channels k4, k4+4 of
    // the chunk at once: R interleaves them, so every value below is a float2 over g and every operation a packed one.
    // Roles are made arithmetically uniform (no per-role selects): u-pair 0 reads patch rows (p,q,r) = (0,1,2) of its tile,
    // u-pair 1 reads them reversed, (3,2,1); then for both
    //     tA = p - r          -> u = 0          | -(d1 - d3) = -t[u=3]
    //     tB = q + s*r        -> u = 1 (s = +1) |   d2 - d1  =  t[u=2]   (s = -1)
    // so Winograd rows are kept in the order rho = (u0, u1, -u3, u2); idiff_pack_conv_weight_wino stores U in the
    // same order with row u3 negated (the product U.V is unchanged), and the output transform reads acc rows (0,1,3,2).
    floatx2 td[3][4];   // patch rows (p, q, r) x 4 columns
    floatx2 to[4];      // one transformed row of 4 positions
    const int trbase = k4 * RPL + 4 * j;
    const int trP = trbase + (2 * tb + 3 * ch) * RROW, trQ = trbase + (2 * tb + 1 + ch) * RROW, trR = trbase + (2 * tb + 2 - ch) * RROW;
    const floatx2 tsign = ch ? floatx2{-1.f, -1.f} : floatx2{1.f, 1.f};
    auto tr_read = [&](int r, int rbuf) {  // patch row r of (p, q, r): 4 columns x 2 channels
        const float* R = Rb + rbuf * R_FLOATS + (r == 0 ? trP : r == 1 ? trQ : trR);
        const floatx4 lo = *reinterpret_cast<const floatx4*>(R);
        const floatx4 hi = *reinterpret_cast<const floatx4*>(R + 4);
        td[r][0] = lo2(lo), td[r][1] = hi2(lo), td[r][2] = lo2(hi), td[r][3] = hi2(hi);
    };
    float* const vwbase = Vb + ch * 4096 + (tb * 4 + k4) * 32 + j * 2;  // Winograd position = 8*ch + 4*uu + v
    auto tr_row = [&](int uu, int buf) {  // row uu of the pair: combine the patch rows, transform along the columns, write
        floatx2 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = uu == 0 ? td[0][c] - td[2][c] : tsign * td[2][c] + td[1][c];
        to[0] = t[0] - t[2];
        to[1] = t[1] + t[2];
        to[2] = t[2] - t[1];
        to[3] = t[1] - t[3];
        float* const V = vwbase + buf * V_FLOATS;
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<floatx2*>(V + (uu * 4 + v) * 512) = to[v];
    };

    floatx4 acc[16][2];
    const int opoff = k4 * 32 + j * 2;

    // ---- epilogue of item `ord`: in-lane output transform A^T m A, then the conv_igemm epilogue contract ----------------------
    // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r of each 16-row block.  The transform runs on row PAIRS
    // (r, r+1) of the accumulators in packed math; the results are re-paired along x for the float2 stores.
    auto epilogue = [&](int ord) {
        int b, tile, cob, y0, x0;
        decode(ord, b, tile, cob, y0, x0);
        const int co0 = cob * 64;
        const int par2 = ord & 1;
        const float* const ebase = econst + (ord % NP) * ECONST + ch * 32 + 4 * k4;  // + mb*16 + r (+ 64 per table)
        float* const sbase = scratch + par2 * SCRATCH + (tb * 64 + ch * 32 + 4 * k4) * 2;
        // addresses = uniform 64-bit base (sample, wave's channel half and tile row) + per-step uniform offset + ONE per-lane
        // 32-bit offset (row group 4*k4 channels down, tile column 2*j across)
        const long long wave_org = (long long)(co0 + ch * 32) * HWo + (long long)(y0 + 2 * tb) * a.Wout + x0;
        float* const outb = a.out + (long long)b * a.obs + wave_org;
        const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
        const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
        const unsigned lane_off = (unsigned)(4 * k4) * (unsigned)HWo + 2u * j;
        const bool want_stats = a.stats != nullptr;
        // partial patches (image not a multiple of 8x32): H, W are even, so a 2x2 tile is inside the image or outside it
        const bool inside = (y0 + 2 * tb < a.Hout) && (x0 + 2 * j < a.Wout);
        // No global load may sit between the stores of two steps: vmcnt counts loads and stores in order, so waiting for such a
        // load would wait for every store before it.  Per-channel constants come from LDS; the residual / aux operands of step
        // i+1 are requested BEFORE the stores of step i.  One instantiation per (residual, aux) keeps the steps branch-free.
        auto out_steps = [&](auto res_tag, auto aux_tag) {
            constexpr bool RES = decltype(res_tag)::value, AUX = decltype(aux_tag)::value;
            floatx2 nres[2][2], naux[2][2];  // [channel of the pair][dy]
            auto fetch = [&](int st) {       // st = mb*2 + p: channels mb*16 + 2p, +1 of this lane's row group
                const int mb = st >> 1, p = st & 1;
                if (co0 + ch * 32 + mb * 16 >= a.Cout) return;  // uniform
                if (!inside) return;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const long long so = (long long)(mb * 16 + 2 * p + e) * HWo + dy * a.Wout;  // uniform
                        if (RES) nres[e][dy] = *reinterpret_cast<const floatx2*>(resb + so + lane_off);
                        if (AUX) naux[e][dy] = *reinterpret_cast<const floatx2*>(auxb + so + lane_off);
                    }
            };
            fetch(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int mb = st >> 1, p = st & 1;
                if (co0 + ch * 32 + mb * 16 >= a.Cout) continue;  // uniform: a 16-channel block beyond a partial Cout
                floatx2 cres[2][2], caux[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) cres[e][dy] = nres[e][dy], caux[e][dy] = naux[e][dy];
                if (st + 1 < 4) fetch(st + 1);
                floatx2 z[4][2];  // packed over the channel pair
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rho = u < 2 ? u : 5 - u;  // accumulator rows are stored in the order (u0, u1, u3, u2)
                    const floatx4 a0 = acc[rho * 4 + 0][mb], a1 = acc[rho * 4 + 1][mb], a2 = acc[rho * 4 + 2][mb], a3 = acc[rho * 4 + 3][mb];
                    const floatx2 m0 = p ? hi2(a0) : lo2(a0), m1 = p ? hi2(a1) : lo2(a1), m2 = p ? hi2(a2) : lo2(a2), m3 = p ? hi2(a3) : lo2(a3);
                    z[u][0] = m0 + m1 + m2;
                    z[u][1] = m1 - m2 - m3;
                }
                const floatx2 bv = *reinterpret_cast<const floatx2*>(ebase + mb * 16 + 2 * p);
                floatx2 y[2][2];  // [dy][x], packed over the channel pair
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    y[0][x] = z[0][x] + z[1][x] + z[2][x] + bv;
                    y[1][x] = z[1][x] - z[2][x] - z[3][x] + bv;
                }
                if (want_stats) {
                    floatx2 ssum = (y[0][0] + y[0][1]) + (y[1][0] + y[1][1]);
                    floatx2 ssq = (y[0][0] * y[0][0] + y[0][1] * y[0][1]) + (y[1][0] * y[1][0] + y[1][1] * y[1][1]);
                    if (!inside) ssum = floatx2{0.f, 0.f}, ssq = floatx2{0.f, 0.f};
                    const float s0 = row_sum16(ssum.x), q0 = row_sum16(ssq.x), s1 = row_sum16(ssum.y), q1 = row_sum16(ssq.y);
                    if (j == 15) *reinterpret_cast<floatx4*>(sbase + (mb * 16 + 2 * p) * 2) = floatx4{s0, q0, s1, q1};
                }
                const floatx2 add = *reinterpret_cast<const floatx2*>(ebase + 64 + mb * 16 + 2 * p);
                floatx2 aa = {0.f, 0.f}, ab = {0.f, 0.f};
                if (AUX) aa = *reinterpret_cast<const floatx2*>(ebase + 128 + mb * 16 + 2 * p), ab = *reinterpret_cast<const floatx2*>(ebase + 192 + mb * 16 + 2 * p);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const floatx2 y0p = y[dy][0] + add, y1p = y[dy][1] + add;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        floatx2 v = floatx2{y0p[e], y1p[e]};  // channel e of the pair, pixels x = 0, 1
                        if (RES) v += cres[e][dy];
                        if (AUX) {
                            const floatx2 t = aa[e] * caux[e][dy] + ab[e];
                            v += floatx2{silu_fast(t.x), silu_fast(t.y)};
                        }
                        if (inside)
                            *reinterpret_cast<floatx2*>(outb + ((long long)(mb * 16 + 2 * p + e) * HWo + dy * a.Wout) + lane_off) = v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: interleaving them only buys register pressure
            }
        };
        const bool has_res = a.res != nullptr, has_aux = a.aux != nullptr;
        if (has_res) {
            if (has_aux) out_steps(std::true_type{}, std::true_type{});
            else out_steps(std::true_type{}, std::false_type{});
        } else {
            if (has_aux) out_steps(std::false_type{}, std::true_type{});
            else out_steps(std::false_type{}, std::false_type{});
        }
    };
    // GroupNorm partials of item `ord` (written to LDS scratch by its epilogue, one barrier ago) -> global
    auto flush_stats = [&](int ord) {
        if (tid < 128) {
            int b, tile, cob, y0, x0;
            decode(ord, b, tile, cob, y0, x0);
            const float* S = scratch + (ord & 1) * SCRATCH;
            const float t = (S[tid] + S[128 + tid]) + (S[256 + tid] + S[384 + tid]);
            const int col = tid >> 1, w = tid & 1, co0 = cob * 64;
            if (co0 + col < a.Cout) a.stats[(((long long)b * a.ntiles + tile) * a.Cout + co0 + col) * 2 + w] = t;
        }
    };

    // ---- one pipeline step.  Step s: matrix cores on chunk s | transform chunk s+1, copy its weights global -> LDS |
    // stage raw patch s+2 | global loads raw s+3.  Every stage is guarded by a uniform flag (all true in the steady state; the three
    // warm-up and the three drain steps switch stages off), and the first chunk of an item starts its accumulators from the
    // literal 0 instead of clearing them.  One Winograd position per slice, fenced with sched_barrier so each slice issues
    // between MFMA groups (left alone, hipcc hoists the operand reads and sinks the MFMAs across barriers, idling the matrix pipe).
    auto step = [&](int s) {
        const bool do_mma = s >= 0;                          // s < total always
        const bool do_tr = s + 1 >= 0 && s + 1 < total;      // transform + weight copy of chunk s+1
        const bool do_st = s + 2 >= 0 && s + 2 < total;      // raw staging of chunk s+2
        const bool do_ld = s + 3 < total;                    // raw load of chunk s+3 (s >= -3 always)
        const bool firstc = cm == 0;
        const int buf = s & 1;
        const float* V = Vb + buf * V_FLOATS + tb * 128 + opoff;
        const float* U = Ub + buf * U_FLOATS + ch * 256 + opoff;
        const bool enter = do_ld && cr == 0;  // the load stage crosses into a new item on this step
        floatx2 ob[2], oa0[2], oa1[2];
        if (do_mma) {
            ob[0] = *reinterpret_cast<const floatx2*>(V);
            oa0[0] = *reinterpret_cast<const floatx2*>(U);
            oa1[0] = *reinterpret_cast<const floatx2*>(U + 128);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (do_mma) {
                if (q + 1 < 16) {
                    ob[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(V + (q + 1) * 512);
                    oa0[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(U + (q + 1) * 512);
                    oa1[(q + 1) & 1] = *reinterpret_cast<const floatx2*>(U + (q + 1) * 512 + 128);
                }
                const floatx2 bv = ob[q & 1], av0 = oa0[q & 1], av1 = oa1[q & 1];
                if (firstc) {
                    acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv.x, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv.x, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                } else {
                    acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.x, bv.x, acc[q][0], 0, 0, 0);
                    acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.x, bv.x, acc[q][1], 0, 0, 0);
                }
                acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0.y, bv.y, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1.y, bv.y, acc[q][1], 0, 0, 0);
            }
            if (q == 0 && do_tr) {  // weights first: they must have landed by this step's barrier
                if (cu == 0) {
                    const int cob = (first + iu * G) % a.ncob;
                    rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino + (long long)cob * U_FLOATS), 0, 0x7fffffff, RSRC_FLAGS);
                }
                copy_u(buf ^ 1);
            }
            if (q < 6 && do_st) stage_raw(q, buf);
            if (q == 8 && do_ld) {
                if (enter) enter_item(ir);
                load_raw();
            }
            if (q == 10 && do_tr) tr_read(0, buf ^ 1), tr_read(2, buf ^ 1);
            if (q == 11 && do_tr) tr_row(0, buf ^ 1);       // from rows p, r
            if (q == 12 && do_tr) tr_read(1, buf ^ 1);
            if (q == 13 && do_tr) tr_row(1, buf ^ 1);       // from rows q, r
            if (q == 15 && enter) publish_tables(ir);  // requested at q == 8; nobody reads these tables before the next step
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- stage bookkeeping (scalar) ----
        if (do_tr && ++cu == nchunks) cu = 0, ++iu;
        smask = lmask;  // the staging stage's view on the NEXT step = what the load stage used on this one
        cs = cr, is_ = ir;
        if (do_ld && ++cr == nchunks) cr = 0, ++ir;
        __syncthreads();
    };

    // Three warm-up steps without matrix work fill the pipeline for chunk 0, the last three steps drain it.  An item's epilogue follows the barrier of its last chunk; its GroupNorm partials
    // leave LDS one barrier later (fl_wait -> fl_go), while the matrix cores already work on the next item.
    int fl_wait = -1, fl_go = -1;
    for (int s = -3; s < total; ++s) {
        if (fl_go >= 0) flush_stats(fl_go);
        step(s);
        fl_go = fl_wait, fl_wait = -1;
        if (s >= 0) {
            if (cm == nchunks - 1) {
                epilogue(im);
                if (a.stats) fl_wait = im;
            }
            if (++cm == nchunks) cm = 0, ++im;
        }
    }
    if (a.stats) {
        if (fl_go >= 0) flush_stats(fl_go);
        __syncthreads();
        if (fl_wait >= 0) flush_stats(fl_wait);
    }
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, hipStream_t st) {
    const int nchunks = a.Cin / CK;
    const size_t lds = ((size_t)2 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 2 * SCRATCH + (size_t)table_parities(nchunks) * ECONST +
                        (SPEC == 2 ? (size_t)table_parities(nchunks) * 2 * a.C0r : 0)) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d(winograd): LDS budget exceeded (%zu bytes)", lds);
    static size_t attr_set = 0;
    auto kern = conv_wino_kernel<MODE, SPEC>;
    if (lds > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd): hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = lds;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd)");
    return IDIFF_OK;
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
    // any even image size of at least one patch row's width class (>= 32 wide: the GroupNorm-partial tiling is the 8x32 one of
    // idiff_conv2d_num_tiles); partial patches at the right / bottom border are masked
    if (a.Cout % 16 || a.Cin % CK || a.C0v % CK || (a.Hout & 1) || (a.Wout & 1) || a.Wout < TW) return false;
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.pro_a || a.src1)) return false;
    if (a.pro_a && a.src1) return false;
    const int np = table_parities(a.Cin / CK);
    if (((size_t)2 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 2 * SCRATCH + (size_t)np * ECONST + (a.pro_a ? (size_t)np * 2 * a.C0r : 0)) * sizeof(float) >
        160 * 1024)
        return false;  // LDS
    if ((long long)a.Cin * a.Hin * a.Win * 4 >= (1ll << 31)) return false;  // 32-bit byte offsets inside a sample
    if ((reinterpret_cast<uintptr_t>(a.wwino) & 15) != 0) return false;
    // float2 epilogue accesses: even row pitch is implied by the even width; batch strides must keep 8-byte alignment
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
