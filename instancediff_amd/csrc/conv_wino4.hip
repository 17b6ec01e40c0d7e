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
#define TRACE_MARK(k)                                                  \
    tr_t[k] = __builtin_readcyclecounter();                            \
    if (k > 0) tr_acc[k - 1] += tr_t[k] - tr_t[k - 1];                 \
    else if (tr_t[3] != 0) tr_acc[3] += tr_t[0] - tr_t[3];
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
// HEAVY: the wave's transform class (waves 0-3 / 4-7).  The whole body -- thread constants, fill, item loop, epilogue -- exists once
// per class: constants of the other class are never live.
//
// Round 4 form.  (1) EVERY global load of the main loop is an LDS-DMA (`buffer_load_dword[x4] ... offen lds`): the weights of a
// chunk land verbatim in U (the weight image is the lane-linear LDS image), the gathered input patch lands in R (destination of a
// wave-instruction = M0 + lane * size and thread t stages R[t + i*512]: lane-linear by construction; a lane whose offset fails the
// range check -- padding, outside the image -- writes 0.0: scripts/proto/lds_dma_probe.hip).  No load result ever sits in a VGPR:
// 32 registers fewer than the register-staged form, no scratch traffic at all (a spill reload behind the epilogue's stores would
// drain them: vmcnt retires in order), no ds_write of the operands.  The copies are inline asm, invisible to hipcc's wait counts:
// every wait for them is a counted s_waitcnt before the chunk's barrier.  R has three buffers: position p is requested behind the
// barrier of chunk p-4, has landed by the barrier of chunk p-2 (SPEC 2: p-3, then activated in place -- GroupNorm/FiLM affine +
// SiLU, padding re-zeroed -- during chunk p-2 by the thread that requested it: thread-private, no barrier) and is transformed
// during chunk p-1.  (2) The chunk pipeline is CONTINUOUS ACROSS THE ITEMS of a workgroup: positions >= n are the next item's chunks
// (other resource bases and affine rows by scalar selects; the ONE gather table is rewritten for the next item at the chunk whose
// request crosses over), so an item's main loop starts right behind its predecessor's epilogue -- no fill (9 k cycles of an 83 k
// item at Cin = 64), no barrier at the item's top -- and its first chunk waits for nothing younger than the epilogue's 128 KB of
// stores (their drain overlaps the first chunks).  A workgroup without a next item streams through zero-record resources.
template <int MODE, int SPEC, bool RAG, bool HEAVY>
__device__ __forceinline__ void conv_wino4_body(const ConvArgs& a, const Geo4& g, float* smem TRACE_PARAM) {
    float* const Rb = smem;                   // [3][R_FLOATS]
    float* const Vb = smem + 3 * R_FLOATS;    // [2][V_FLOATS]
    float* const Ub = Vb + 2 * V_FLOATS;      // [2][U_FLOATS]
    float* const econst = Ub + 2 * U_FLOATS;  // [2 item parities][4][64] bias, vec, aux_a, aux_b of an item's 64 output channels
    int* const gtab = reinterpret_cast<int*>(econst + 512);  // [NL][NT] gather byte offsets (thread-private entries)
    constexpr unsigned R_BYTES = R_FLOATS * 4, U_BYTES = U_FLOATS * 4;
    // LDS byte address of the dynamic segment (the DMA destinations are absolute LDS addresses in M0): the low half of its flat address
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<unsigned long long>(smem));
    const unsigned U_LDS0 = lds_base + (3 * R_FLOATS + 2 * V_FLOATS) * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k4 = lane >> 4;  // k index of the MFMA operands (lane & 15: tile of the B operand / channel of the A operand)
    const int cb = wave & 3;   // MFMA role: 16-channel block
    const int tblk = wave >> 2;  // MFMA role: upper / lower 8x32 half-patch
    const int HWin = a.Hin * a.Win;
    const int nchunks = a.Cin / CK;  // even, >= 4 (Cin % 8 == 0, Cin >= 16: conv_wino4_items)

    // ---- per-item state.  Set A = the item being accumulated, set B = the next one (or the null item) ----------------------
    constexpr int RSRC_FLAGS = 0x00020000;
    struct View {  // the epilogue's view of an item (scalars)
        int b, co0, y0, x0;
    };
    const float *p0A = a.src0, *p1A = a.src0, *puA = a.wwino4, *p0B = a.src0, *p1B = a.src0, *puB = a.wwino4;
    View vA = {0, 0, 0, 0}, vB = {0, 0, 0, 0};
    int nrB = 0;                      // num_records of set B's resources: 0 = the null item
    unsigned omaskA = 0, omaskB = 0;  // SPEC 2: bit i = element i is padding / outside the image (zero AFTER the activation)
    // The item index advances by the grid size G: (channel block, patch column, patch row, sample) are carried as a mixed-radix
    // counter with a constant increment -- scalar adds and compares per item instead of five integer divisions.
    int it_b = 0, it_cob = 0, it_px = 0, it_py = 0;
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
    auto decode_B = [&]() {  // the item (it_b, it_cob, it_px, it_py) becomes set B: scalars only
        vB.b = it_b, vB.co0 = it_cob * 64, vB.y0 = it_py * TH, vB.x0 = it_px * TW;
        puB = scalar_ptr(a.wwino4 + (long long)it_cob * U_FLOATS);
        p0B = scalar_ptr(a.src0 + (long long)it_b * a.bs0);
        p1B = scalar_ptr(SPEC == 3 ? a.src1 + (long long)it_b * a.bs1 : a.src0);
        nrB = 0x7fffffff;
    };
    // Thread requests R[tid + i*512]: the R index itself enumerates (ci, row, col).  An element's load offset (bytes inside the
    // sample) is -1 for pad slots and for elements outside the image: the range check fails and the DMA writes 0.0.  The six
    // offsets of an item are decoded once and parked in LDS (thread-private entries); returns the item's padding mask.
    auto write_table = [&](const View& v) {
        int t = tid;
        asm volatile("" : "+v"(t));  // opaque: keeps the decode here, once per item, instead of hoisted and held in registers
        unsigned om = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = t + i * NT;
            const int ci = e / PSP;
            const int rem = e - ci * PSP;
            const int r = rem / RS;
            const int c = rem - r * RS;
            const int oy = v.y0 - 1 + r, ox = v.x0 - 1 + c;  // output-grid coordinates of the element
            const bool in = rem < PS && c < RCOLS && (unsigned)oy < (unsigned)a.Hout && (unsigned)ox < (unsigned)a.Wout;
            const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
            gtab[i * NT + tid] = in ? (ci * HWin + sp) * 4 : -1;
            om |= (in ? 0u : 1u) << i;
        }
        return om;
    };
    auto shift_B_to_A = [&]() { p0A = p0B, p1A = p1B, puA = puB, vA = vB, omaskA = omaskB; };
    const int ustride_b = a.ncob * U_FLOATS * 4;  // bytes between chunks of one channel block

    // SPEC 2: the GroupNorm/FiLM affine of a chunk's four input channels comes through the scalar cache (uniform addresses;
    // constant address space makes them s_load_dwordx4) -- a wave's 64 elements never straddle a channel (768 = 12 * 64), so an
    // element's channel, and with it the affine, is wave-uniform.
    typedef const __attribute__((address_space(4))) floatx4* cfloatx4p;
    struct Pro {
        floatx4 a, b;
    };
    // Stream positions: p < n is chunk p of set A, p >= n chunk p - n of set B (n >= 4 and p <= n + 3: never beyond B).
    auto load_pro = [&](int p) {
        Pro r;
        if (SPEC == 2) {
            const bool inA = p < nchunks;
            const long long o = (long long)(inA ? vA.b : vB.b) * a.C0r + (inA ? p : p - nchunks) * CK;
            r.a = *(cfloatx4p)(a.pro_a + o);
            r.b = *(cfloatx4p)(a.pro_b + o);
        }
        return r;
    };

    // ---- LDS-DMA.  M0 (the destination base of a wave-instruction) is written inside the statement that uses it: hipcc does not
    // preserve it around asm.  None of these loads is in hipcc's vmcnt bookkeeping.  Every statement opens with `s_nop 4`: an "s"
    // operand may be fresh from a VALU write (readfirstlane, the v_readlane of an SGPR spill reload) and a buffer instruction that
    // reads it as descriptor or soffset needs five wait states hipcc's hazard recognizer does not insert for inline asm.
    // The 2304 float4 of weight chunk p -> U[buf]: four 1-KB pieces per wave, the last 256 float4 by the light waves.
    const unsigned u_voff = tid * 16;
    auto dma_u = [&](int p, int buf) {
        const bool inA = p < nchunks;
        const int so = (inA ? p : p - nchunks) * ustride_b;
        const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inA ? puA : puB), 0, inA ? 0x7fffffff : nrB, RSRC_FLAGS);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(U_LDS0 + (unsigned)buf * U_BYTES + (unsigned)wave * 1024u);
        asm volatile(
            "s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %4 offen lds\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %5 offen lds\n\t"
            "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %6 offen lds"
            ::"v"(u_voff), "s"(rsu), "s"(lds0), "s"(so), "s"(so + NT * 16), "s"(so + 2 * NT * 16), "s"(so + 3 * NT * 16)
            : "memory", "scc");
        if (!HEAVY) {  // waves 4-7: float4 2048 + (tid - 256) of the chunk
            const unsigned lds4 = __builtin_amdgcn_readfirstlane(lds0 + 4 * NT * 16 - 256 * 16);
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(u_voff), "s"(rsu), "s"(lds4),
                         "s"(so + 4 * NT * 16 - 256 * 16)
                         : "memory");
        }
    };
    // one 1-KB piece of the same copy: i = 0..3, and 4 = the light waves' fifth.  Between two barriers a chunk's pieces are dealt
    // out one per quad: eight waves issuing 36 of them at once right behind the barrier held the last wave in line ~1 k cycles
    auto dma_u_piece = [&](int p, int buf, int i) {
        const bool inA = p < nchunks;
        const int so = (inA ? p : p - nchunks) * ustride_b + (i < 4 ? i * NT * 16 : 4 * NT * 16 - 256 * 16);
        const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inA ? puA : puB), 0, inA ? 0x7fffffff : nrB, RSRC_FLAGS);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(U_LDS0 + (unsigned)buf * U_BYTES + (unsigned)wave * 1024u + (unsigned)(i < 4 ? i * NT * 16 : 4 * NT * 16 - 256 * 16));
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(u_voff), "s"(rsu), "s"(lds0), "s"(so) : "memory");
    };
    // The gathered patch of position p -> R slot at byte offset rslot: six dwords per thread, element tid + i*512
    auto dma_raw = [&](int p, unsigned rslot) {
        const bool inA = p < nchunks;
        const int cbase = (inA ? p : p - nchunks) * CK;
        int goff[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) goff[i] = gtab[i * NT + tid];
        const int C0v = a.C0v;
        const bool second = SPEC == 3 && cbase >= C0v;  // chunk-uniform: C0v % 4 == 0
        // (selected as integers: a select of selects of pointers next to a field of `a` kept the whole argument block in scratch)
        const unsigned long long b0 = reinterpret_cast<unsigned long long>(inA ? p0A : p0B), b1 = reinterpret_cast<unsigned long long>(inA ? p1A : p1B);
        const float* const base = reinterpret_cast<const float*>(second ? b1 : b0);
        const int so = (second ? cbase - C0v : cbase) * HWin * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, inA ? 0x7fffffff : nrB, RSRC_FLAGS);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + rslot + (unsigned)wave * 256u);
        asm volatile(
            "s_nop 4\n\ts_mov_b32 m0, %8\n\ts_nop 0\n\tbuffer_load_dword %0, %6, %7 offen lds\n\t"
            "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dword %1, %6, %7 offen lds\n\t"
            "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dword %2, %6, %7 offen lds\n\t"
            "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dword %3, %6, %7 offen lds\n\t"
            "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dword %4, %6, %7 offen lds\n\t"
            "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dword %5, %6, %7 offen lds"
            ::"v"(goff[0]), "v"(goff[1]), "v"(goff[2]), "v"(goff[3]), "v"(goff[4]), "v"(goff[5]), "s"(rs), "s"(so), "s"(lds0)
            : "memory", "scc");
    };
    // the same, one element per call (SPEC 1 / 3: dealt out over the quads of a chunk -- eight waves issuing all their copies right
    // behind the barrier queue ~85 wave-instructions at the CU's one address unit: 1.3 k cycles of a 4.1 k chunk for the last in line)
    auto dma_raw_piece = [&](int p, unsigned rslot, int i, int goff_i) {
        const bool inA = p < nchunks;
        const int cbase = (inA ? p : p - nchunks) * CK;
        const int C0v = a.C0v;
        const bool second = SPEC == 3 && cbase >= C0v;
        const unsigned long long b0 = reinterpret_cast<unsigned long long>(inA ? p0A : p0B), b1 = reinterpret_cast<unsigned long long>(inA ? p1A : p1B);
        const float* const base = reinterpret_cast<const float*>(second ? b1 : b0);
        const int so = (second ? cbase - C0v : cbase) * HWin * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, inA ? 0x7fffffff : nrB, RSRC_FLAGS);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + rslot + (unsigned)wave * 256u + (unsigned)i * 2048u);
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(goff_i), "s"(rs), "s"(so), "s"(lds0) : "memory");
    };
    // SPEC 2 (GroupNorm/FiLM affine + SiLU applied while the patch is staged): the VALU has to touch every element anyway, so this
    // form keeps the register path -- plain buffer loads into rinA / rinB (even / odd positions), activation in registers, one
    // ds_write per element -- with two chunks of latency budget (an LDS-DMA'd patch would have to land within ONE chunk to be
    // activated in place during the next: measured +76 us on 64 -> 64 at 256^2).  The loads are inline asm like the copies, i.e.
    // NOT in hipcc's vmcnt bookkeeping: their destinations are only touched behind the counted wait that names them.
    float rinA[NL], rinB[NL];
    auto ld_raw = [&](float (&dst)[NL], int p, int i, int goff_i) {
        const bool inA = p < nchunks;
        const int so = (inA ? p : p - nchunks) * CK * HWin * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inA ? p0A : p0B), 0, inA ? 0x7fffffff : nrB, RSRC_FLAGS);
        asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst[i]) : "v"(goff_i), "s"(rs), "s"(so) : "memory");
    };
    // x -> silu(a_c * x + b_c), padding back to zero, into the R slot: elements i0 .. i1-1 of position p
    auto stage_act = [&](const float (&src)[NL], int p, const Pro& pro, unsigned rslot, int i0, int i1) {
        const unsigned om = p < nchunks ? omaskA : omaskB;
        float* const R = reinterpret_cast<float*>(reinterpret_cast<char*>(Rb) + rslot) + tid;
        constexpr bool hiw = !HEAVY;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i < i0 || i >= i1) continue;
            // channel of element tid + i*512: (i*512 + wave*64) / 768 = {0, 0|1, 1, 2, 2|3, 3}[i]
            const float pa = i == 0 ? pro.a.x : i == 1 ? (hiw ? pro.a.y : pro.a.x) : i == 2 ? pro.a.y : i == 3 ? pro.a.z : i == 4 ? (hiw ? pro.a.w : pro.a.z) : pro.a.w;
            const float pb = i == 0 ? pro.b.x : i == 1 ? (hiw ? pro.b.y : pro.b.x) : i == 2 ? pro.b.y : i == 3 ? pro.b.z : i == 4 ? (hiw ? pro.b.w : pro.b.z) : pro.b.w;
            const float x = silu_fast(pa * src[i] + pb);
            R[i * NT] = ((om >> i) & 1u) ? 0.f : x;
        }
    };
    constexpr int N_U = HEAVY ? 4 : 5;  // 1-KB weight copies per wave and chunk

    // ---- input transform B^T d B of an R slot -> V[buf].  Thread = (ci = k4, tile (tyl, tx) of half-patch thalf) x row set:
    //   heavy waves 0-3: Winograd rows (1,2) (trole 0) or (3,4) (trole 1):  X = d4 + al*d2, Y = d3 + al*d1, rows X +- be*Y
    //   light waves 4-7: row 0 (from d0, d2, d4) or row 5 (from d1, d3, d5): 4*dA - 5*dB + dC
    constexpr bool heavy = HEAVY;
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
    auto tr_piece = [&](int piece, unsigned rslot, int buf) {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(trbase) + rslot);
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

    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = g.total;
    if (first >= last) return;  // (the launchers size the grid so that it cannot happen)
    auto fetch_consts = [&](const View& v) {  // an item's epilogue constants: threads 0..255 (= the heavy waves), one each
        float e = 0.f;
        if (HEAVY) {
            const int which = tid >> 6, co = v.co0 + (tid & 63);
            if (co < a.Cout) {
                if (which == 0 && a.bias) e = a.bias[co];
                if (which == 1 && a.vec) e = a.vec[(long long)v.b * a.Cout + co];
                if (which == 2 && a.aux) e = a.aux_a[(long long)v.b * a.Cout + co];
                if (which == 3 && a.aux) e = a.aux_b[(long long)v.b * a.Cout + co];
            }
        }
        return e;
    };
    TRACE_INIT
    SLOT_INIT

    // ---- the workgroup's first item: the only pipeline fill of the launch --------------------------------------------------------
    // leaves what every item's last chunk leaves: V[0] <- chunk 0, R slot 1 <- chunk 1 (activated), R slot 2 <- chunk 2 (landed),
    // U[0], U[1]; chunk 3 requested into slot 0 behind the last barrier
    decode_first(first);
    decode_B();
    shift_B_to_A();
    omaskA = write_table(vA);
    if (first + G < last) {
        advance_item();
        decode_B();
    } else {
        nrB = 0, vB = vA;
    }
    {
        const float e0 = fetch_consts(vA);  // (hipcc-visible global loads, waited for by hipcc -- ahead of every hidden one)
        if (HEAVY) econst[tid] = e0;
    }
    dma_u(0, 0);
    if (SPEC != 2) {
        dma_raw(0, 0 * R_BYTES);
        dma_raw(1, 1 * R_BYTES);
        dma_raw(2, 2 * R_BYTES);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");  // U(0), positions 0 and 1 have landed
        __syncthreads();  // an LDS-DMA is ordered for a ds_read only by the issuer's vmcnt FOLLOWED by a barrier the reader has passed
    } else {
        int goff[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) goff[i] = gtab[i * NT + tid];
#pragma unroll
        for (int i = 0; i < NL; ++i) ld_raw(rinA, 0, i, goff[i]);
#pragma unroll
        for (int i = 0; i < NL; ++i) ld_raw(rinB, 1, i, goff[i]);
        const Pro pro0 = load_pro(0), pro1 = load_pro(1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rinA[0]), "+v"(rinA[1]), "+v"(rinA[2]), "+v"(rinA[3]), "+v"(rinA[4]), "+v"(rinA[5]), "+v"(rinB[0]),
                     "+v"(rinB[1]), "+v"(rinB[2]), "+v"(rinB[3]), "+v"(rinB[4]), "+v"(rinB[5])::"memory");
        stage_act(rinA, 0, pro0, 0 * R_BYTES, 0, NL);
        stage_act(rinB, 1, pro1, 1 * R_BYTES, 0, NL);
#pragma unroll
        for (int i = 0; i < NL; ++i) ld_raw(rinA, 2, i, goff[i]);  // staged by chunk 0; position 3 is requested during chunk 0
        __syncthreads();
    }
#pragma unroll
    for (int piece = 0; piece < 5; ++piece) tr_piece(piece, 0 * R_BYTES, 0);
    // everything requested so far has landed -- position 2's registers included: the compiler copies them into the loop's own
    // registers at its entry (phi copies: the ONLY place scripts/lint_asm_loads.py finds a hidden load's destination touched without
    // a wait of ours in between, hence this one)
    if (SPEC == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rinA[0]), "+v"(rinA[1]), "+v"(rinA[2]), "+v"(rinA[3]), "+v"(rinA[4]), "+v"(rinA[5])::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dma_u(1, 1);

    unsigned r_t = 1 * R_BYTES, r_a = 2 * R_BYTES, r_x = 0 * R_BYTES;  // R slots: transformed this chunk / activated this chunk / landing
    int epar = 0;        // parity of the item being accumulated: its set of epilogue constants
    bool young = false;  // the requests in flight were issued BEFORE an epilogue: its >= 16 stores are younger than they are
    for (int item = first; item < last; item += G) {
        TRACE_MARK(0)
        floatx4 acc[36];  // started from C = 0 by the first chunk's MFMAs (no 144 v_mov per item)
        {
            // ONE barrier per chunk.  Chunk c runs its 9 position quads and, one slice per quad:
            //   transform position c+1 (slot r_t) -> V[(c+1)&1];  SPEC 2: activate position c+2 from its registers into slot r_a;
            //   the copies of U(c+1) (first one behind the previous barrier) and the six requests of position c+3.
            // At the barrier U(c+1) and position c+2 must have landed / been staged; the six requests of position c+3 may fly on.
            const int opoff = lane * 4;
            floatx4 ob[2], oa[2];
            ob[0] = *reinterpret_cast<const floatx4*>(Vb + tblk * 256 + opoff);  // quad 0 of chunk 0
            oa[0] = *reinterpret_cast<const floatx4*>(Ub + cb * 256 + opoff);
            auto chunk = [&](int cc, auto par_tag, auto zero_tag) {
                constexpr int PAR = decltype(par_tag)::value;      // cc & 1: V / U buffers
                constexpr bool ZERO = decltype(zero_tag)::value;   // the item's first chunk
                const float* V = Vb + PAR * V_FLOATS + tblk * 256 + opoff;
                const float* U = Ub + PAR * U_FLOATS + cb * 256 + opoff;
                // Operand quads alternate between two register sets; the parity flips from chunk to chunk (9 quads), so quad 8 of
                // this chunk and quad 0 of the next never share a set: the next chunk's first operands are requested right after
                // the barrier and arrive while the four MFMAs of this chunk's last quad run.
                const float* Vn = Vb + (PAR ^ 1) * V_FLOATS + tblk * 256 + opoff;
                const float* Un = Ub + (PAR ^ 1) * U_FLOATS + cb * 256 + opoff;
                // Position c+3 is requested during this chunk, one element per quad: SPEC 1 / 3 by LDS-DMA into the slot freed by the
                // previous chunk's transform (r_x), SPEC 2 into the register set staged by the previous chunk; when that request
                // crosses into the next item the table becomes that item's first
                int goff[NL];
                if (cc + 3 == nchunks) omaskB = write_table(vB);
#pragma unroll
                for (int i = 0; i < NL; ++i) goff[i] = gtab[i * NT + tid];
                float(&rin_s)[NL] = PAR ? rinB : rinA;      // SPEC 2: holds position c+2, staged by this chunk
                float(&rin_l)[NL] = PAR ? rinA : rinB;      // SPEC 2: receives position c+3
                Pro pro;
                if (SPEC == 2) pro = load_pro(cc + 2);
                SLOT_START
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if (q + 1 < 9) {
                        ob[(q + 1 + PAR) & 1] = *reinterpret_cast<const floatx4*>(V + (q + 1) * 512);
                        oa[(q + 1 + PAR) & 1] = *reinterpret_cast<const floatx4*>(U + (q + 1) * 1024);
                    } else {
                        // (partial patches: a wave wholly outside the image issues no store -- nothing to count on)
                        if (young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + (RAG ? 0 : 16)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                        young = false;
                        __builtin_amdgcn_sched_barrier(0);
                        SLOT_MARK(8)
                        __syncthreads();
                        SLOT_MARK(9)
                        // U(c+2): before an epilogue all of it at once (its stores must stay younger than these copies), else the
                        // first piece here and the others over the next chunk's first quads
                        if (cc + 1 == nchunks) dma_u(cc + 2, PAR);
                        else dma_u_piece(cc + 2, PAR, 0);
                        ob[PAR ^ 1] = *reinterpret_cast<const floatx4*>(Vn);
                        oa[PAR ^ 1] = *reinterpret_cast<const floatx4*>(Un);
                    }
                    const floatx4 bv = ob[(q + PAR) & 1], av = oa[(q + PAR) & 1];
                    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
                    acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, ZERO ? z4 : acc[4 * q + 0], 0, 0, 0);
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, ZERO ? z4 : acc[4 * q + 1], 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, ZERO ? z4 : acc[4 * q + 2], 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, ZERO ? z4 : acc[4 * q + 3], 0, 0, 0);
                    if (!ZERO) {  // (chunk 0: U(1) was copied whole before the epilogue / by the fill)
                        if (q < 3) dma_u_piece(cc + 1, PAR ^ 1, q + 1);
                        if (q == 3 && !HEAVY) dma_u_piece(cc + 1, PAR ^ 1, 4);
                    }
                    if (SPEC == 2 && q == 5) {
                        // position c+2 (requested over the previous chunk) has arrived: younger than it are the N_U copies of
                        // U(c+1), the two elements of position c+3 requested so far and, behind an epilogue, its stores
                        if (young) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rin_s[0]), "+v"(rin_s[1]), "+v"(rin_s[2]), "+v"(rin_s[3]), "+v"(rin_s[4]), "+v"(rin_s[5]) : "n"(N_U + 2 + (RAG ? 0 : 16)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rin_s[0]), "+v"(rin_s[1]), "+v"(rin_s[2]), "+v"(rin_s[3]), "+v"(rin_s[4]), "+v"(rin_s[5]) : "n"(N_U + 2) : "memory");
                        stage_act(rin_s, cc + 2, pro, r_a, 0, 2);
                    }
                    if (SPEC == 2 && q == 6) stage_act(rin_s, cc + 2, pro, r_a, 2, 4);
                    if (SPEC == 2 && q == 7) stage_act(rin_s, cc + 2, pro, r_a, 4, 6);
                    // position c+3, behind the last piece of U(c+1): the counted wait at the barrier leaves exactly these six in flight
                    auto req = [&](int i) {
                        if (SPEC == 2) ld_raw(rin_l, cc + 3, i, goff[i]);
                        else dma_raw_piece(cc + 3, r_x, i, goff[i]);
                    };
                    if (HEAVY && q == 3) req(0);
                    if (!HEAVY && q == 4) req(0);
                    if (q == 4) req(1);
                    if (q == 5) req(2);
                    if (q == 6) req(3);
                    if (q == 7) req(4), req(5);
                    if (q < 5) tr_piece(q, r_t, PAR ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < 8) { SLOT_MARK(q) }
                }
                const unsigned t = r_t;  // next chunk: transform what was activated, activate what has landed, land into the freed slot
                r_t = r_a, r_a = r_x, r_x = t;
            };
            chunk(0, std::integral_constant<int, 0>{}, std::true_type{});
            chunk(1, std::integral_constant<int, 1>{}, std::false_type{});
            for (int cc = 2; cc < nchunks; cc += 2) {
                chunk(cc, std::integral_constant<int, 0>{}, std::false_type{});
                chunk(cc + 1, std::integral_constant<int, 1>{}, std::false_type{});
            }
        }
        TRACE_MARK(1)

        // ---- item switch: the accumulators belong to `v`; set B (streamed in by the last chunks) becomes set A, the item after it
        // is decoded (scalars; its table is written when the requests reach it), and the new A's epilogue constants are requested
        // BEFORE this epilogue's stores
        const View v = vA;
        const float* const ebase_item = econst + epar * 256;
        const bool more = item + G < last;
        float pre_e = 0.f;
        if (more) {
            shift_B_to_A();
            if (item + 2 * G < last) {
                advance_item();
                decode_B();
            } else {
                nrB = 0, vB = vA;
            }
            pre_e = fetch_consts(vA);
        }
        young = true;

        // ---- epilogue: in-lane output transform A^T m A, then the conv_igemm epilogue contract ---------------------------
        // C layout of 16x16x4: lane holds column j (tile) and rows 4*k4 + r (channels) of the wave's 16-channel block
        const int b = v.b, co0 = v.co0, y0 = v.y0, x0 = v.x0;
        {
            // lane-derived constants are recomputed here from an opaque copy of the lane id (not held through the main loop)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int j = lane_e & 15, k4 = lane_e >> 4;
            const int HWo = a.Hout * a.Wout;
            const int ty0 = y0 + 8 * tblk;  // first row of the wave's half-patch
            const bool wave_live = co0 + cb * 16 < a.Cout;  // uniform: a 16-channel block beyond a partial Cout stores nothing
            // Stores go through buffer resources: a dead wave's resource has zero records -- the range check drops its stores, but
            // they are issued and counted.  Without partial patches (!RAG) no branch surrounds a store: every wave issues the same
            // 16 (+4) stores per item, which is what lets the next item's first chunk COUNT them (vmcnt) instead of draining them.
            const long long wave_org = (long long)(co0 + cb * 16) * HWo + (long long)ty0 * a.Wout + x0;
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scalar_ptr(a.out + (long long)b * a.obs + wave_org)), 0, wave_live ? 0x7fffffff : 0, RSRC_FLAGS);
            const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
            const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
            const bool inside = !RAG || ((ty0 + 4 * (j >> 3) < a.Hout) && (x0 + 4 * (j & 7) < a.Wout));
            const unsigned lane_off = (unsigned)(4 * k4) * (unsigned)HWo + (unsigned)(4 * (j >> 3)) * (unsigned)a.Wout + 4u * (j & 7);
            const float* const ebase = ebase_item + cb * 16 + 4 * k4;
            const bool want_stats = a.stats != nullptr && ty0 < a.Hout && wave_live;
            const bool has_res = a.res != nullptr && wave_live, has_aux = a.aux != nullptr && wave_live;
            const float* const stbase = a.stats ? a.stats + (((long long)b * a.ntiles + (ty0 >> 3) * a.tiles_x + (x0 >> 5)) * a.Cout + co0 + cb * 16) * 2 : a.out;
            const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scalar_ptr(stbase)), 0, want_stats ? 0x7fffffff : 0, RSRC_FLAGS);
            constexpr int AUX_SC1 = 1 << 4;  // write-through
            // Phase 1: A^T along the Winograd columns v of every row u and channel r -- 144 accumulators shrink to 96 values
            // (the accumulators of a row die as soon as it is done, so the registers hold either form, never both).
            float zz[6][4][4];  // [u][r][dx]
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    at6(acc[pos(u, 0)][r], acc[pos(u, 1)][r], acc[pos(u, 2)][r], acc[pos(u, 3)][r], acc[pos(u, 4)][r], acc[pos(u, 5)][r], zz[u][r]);
                    // pinned here: left alone, the optimiser sinks these sums to their uses in phase 2 and keeps the accumulators
                    // alive until then
                    asm volatile("" : "+v"(zz[u][r][0]), "+v"(zz[u][r][1]), "+v"(zz[u][r][2]), "+v"(zz[u][r][3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TRACE_MARK(2)
            // Phase 2, per channel r: A^T along u, bias, GroupNorm partials, then four row steps.  The residual / aux row of
            // step s+1 is requested BEFORE the store of step s (vmcnt counts loads and stores in order: a load behind a store
            // would wait for it).  has_res / has_aux are uniform branches (around loads only).
            floatx4 nres = floatx4{0.f, 0.f, 0.f, 0.f}, naux = nres;
#ifdef W4_STATS_LAST
            typedef unsigned uintx2_ __attribute__((ext_vector_type(2)));
            uintx2_ spr[4];
#endif
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
                {
                    float ssum = 0.f, ssq = 0.f;
                    if (a.stats != nullptr) {  // uniform; around arithmetic only
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy) {
                            ssum += (y[dy][0] + y[dy][1]) + (y[dy][2] + y[dy][3]);
                            ssq += (y[dy][0] * y[dy][0] + y[dy][1] * y[dy][1]) + (y[dy][2] * y[dy][2] + y[dy][3] * y[dy][3]);
                        }
                        if (!inside) ssum = 0.f, ssq = 0.f;
                        ssum = row_sum16(ssum);
                        ssq = row_sum16(ssq);
                    }
                    // lane 15 of each 16-lane row holds the sums of the wave's cell for channel 4*k4 + r
                    typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
                    const uintx2 pr = {__builtin_bit_cast(unsigned, ssum), __builtin_bit_cast(unsigned, ssq)};
#ifndef W4_STATS_LAST
                    __builtin_amdgcn_raw_buffer_store_b64(pr, rst, j == 15 ? (4 * k4 + r) * 8 : -1, 0, AUX_SC1);
#else
                    spr[r] = pr;
#endif
                }
                const float add = ebase[64 + r];
                float aa = 0.f, ab = 0.f;
                if (has_aux) aa = ebase[128 + r], ab = ebase[192 + r];
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const floatx4 cres = nres, caux = naux;
                    if (4 * r + dy + 1 < 16) fetch(4 * r + dy + 1);
                    floatx4 v4 = floatx4{y[dy][0] + add, y[dy][1] + add, y[dy][2] + add, y[dy][3] + add};
                    if (has_res) v4 += cres;
                    if (has_aux) {
                        v4.x += silu_fast(aa * caux.x + ab), v4.y += silu_fast(aa * caux.y + ab);
                        v4.z += silu_fast(aa * caux.z + ab), v4.w += silu_fast(aa * caux.w + ab);
                    }
                    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                    // RAG: lanes outside the image are masked by EXEC, not by the range check -- a 16-byte buffer store with SOME
                    // lanes out of range loses 8 of the 16 bytes of in-range neighbours (measured: profiles/r04/x_wino4_dma_notes.txt);
                    // all lanes out of range (the zero-record resource of a dead wave) is fine
                    if (!RAG || inside) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v4), rso, (int)(lane_off * 4u), (r * HWo + dy * a.Wout) * 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef W4_STATS_LAST
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b64(spr[r], rst, j == 15 ? (4 * k4 + r) * 8 : -1, 0, AUX_SC1);
#endif
        }
        TRACE_MARK(3)
        // the next item's epilogue constants into the other set: its last readers (the epilogue two items back) are behind at least
        // one chunk barrier of the item just finished; its next readers are ordered by the chunk barriers of the item now starting
        if (HEAVY && more) econst[(epar ^ 1) * 256 + tid] = pre_e;
        epar ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the null item's requests (answered with zeros) still target this workgroup's LDS
    TRACE_FINI
    SLOT_FINI
}

template <int MODE, int SPEC, bool RAG>
__global__ __launch_bounds__(NT) void conv_wino4_kernel(const ConvArgs a, const Geo4 g TRACE_PARAM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef IDIFF_WINO_TRACE
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) conv_wino4_body<MODE, SPEC, RAG, true>(a, g, smem, trace);
    else conv_wino4_body<MODE, SPEC, RAG, false>(a, g, smem, trace);
#else
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) conv_wino4_body<MODE, SPEC, RAG, true>(a, g, smem);
    else conv_wino4_body<MODE, SPEC, RAG, false>(a, g, smem);
#endif
}

template <int MODE, int SPEC, bool RAG>
int launch_rag(const ConvArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)3 * R_FLOATS + 2 * V_FLOATS + 2 * U_FLOATS + 2 * 256 + NL * NT) * sizeof(float);  // 158 KB
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
    fprintf(stderr, "[wino4 trace] Cin=%d Cout=%d H=%d items=%d per=%d | main loop %lld | switch + output transform %lld | statistics + stores %lld | to next item %lld (cycles/item, wave 0)\n",
            a.Cin, a.Cout, a.Hout, g.total, per, h[0] / g.total, h[1] / g.total, h[2] / g.total, h[3] / g.total);
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
    // ... and inside the weight image: both F(4x4,3x3) kernels address a chunk of it as chunk * ncob * U_FLOATS * 4 (+ up to 36 KB) in
    // 32-bit scalar offsets (dma_u / dma_u_piece here, the rolling A-operand window of conv_wino4h.hip): 144 bytes per (ci, co) pair
    if (144ll * a.Cin * a.ncob * 64 + U_FLOATS * 4 >= (1ll << 31)) return 0;
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
