// ARCHIVED PROTOTYPE (moved out of instancediff_amd/csrc in round 4: not compiled into libidiff_hip.so, not declared in include/idiff.h,
// not tested).  It was wired behind idiff_conv_desc.algo_request = 1 + 6 with a `wwino4x` weight-image field and
// idiff_pack_conv_weight_wino4x / idiff_conv_wino4x_image_bytes entry points (git history: commit 95dbc1d .. f02b7f7); it lost to the
// f32 kernels on every layer measured and DESIGN.md section 8.1 bounds it at ~1.0-1.3x.  Kept for its measured notes only.
//
// Winograd F(4x4,3x3) convolution on the bf16 matrix cores with fp32 operands split three ways (gfx950, v_mfma_f32_16x16x32_bf16).
// EXPERIMENTAL (round 3): complete (every gather / prologue / epilogue form of conv_wino4.hip, partial patches) and parity-green behind
// idiff_conv_desc.algo_request = 1 + IDIFF_CONV_ALGO_WINOGRAD4X; never the library's own choice -- it does not beat conv_wino4.hip
// yet (64->64 at 256^2: 376 us against 295; 416->256 at 64^2: 420 against 347; profiles/r03/x_wino4x_experiments.txt) and its
// prologue instantiation still spills.  What it fixes for the next round: the arithmetic, the data layouts, the pipeline shape, and
// the measurement that says where the time goes (the weight stream, below).
//
//   Arithmetic.  U = G g G^T and V = B^T d B are formed in fp32 exactly as in conv_wino4.hip; each is then written as the exact sum
//   of three bf16 values (conv1x1_x3.hip) and a Winograd-domain product is the six bf16 products of order >= 2^-16, accumulated in
//   fp32: the error class of the f32 kernel (same test cases, same tolerances).  The 32-deep contraction of the matrix instruction
//   holds 16 input channels x TWO planes: [u1|u2].[v1|v1] + [u1|u2].[v2|v2] + [u1|u3].[v3|v1] = u1v1 + u2v1 + u1v2 + u2v2 + u1v3 + u3v1
//   -- three instructions per 16 channels, so a chunk is 16 channels and R, the activated patch, stays at 48 KB.
//
//   Work.  Persistent 512-thread workgroups, item = 16x32 pixels (32 tiles) x 64 output channels, as conv_wino4.hip (same gather,
//   same fused prologue / epilogue contract, same GroupNorm-partials grid).  V lives in LDS one PAIR of Winograd rows at a time
//   ([row][v][plane][octet][tile][8 bf16], 37 KB, two pair buffers): all 36 positions of a chunk would be 110 KB.  Per chunk:
//   stage R | transform pair 0 | multiply pair 0 with the transform of pair 1 dealt out between its twelve units (one unit = one
//   position x one tile block = three MFMAs; B one unit ahead) | multiply pair 1 with the transform of pair 2 | multiply pair 2 with
//   the next chunk's pixels in flight.  Transform task = (tile, one channel), both rows of the pair from the five patch rows they
//   share, outputs split and stored as single bf16.  MFMA role: wave (cq, vh) owns 16 output channels x all 32 tiles x the 18
//   positions (u, 3 vh + 0..2) = 144 accumulators; A (weights, pre-split image of idiff_pack_conv_weight_wino4x) comes straight from
//   L2 through a rolling window of three positions.
//   Epilogue: A^T along u in-lane (a wave holds all six u of its columns), the two waves of a channel block swap the halves they
//   do not finish (16 tiles each, through LDS), A^T along v over all six columns, then conv_wino4.hip's epilogue unchanged.
//
//   Where the time goes (W4X_TRACE build, wave 0, 64 -> 64 at 256^2, four chunks, cycles per item): setup + first loads 21.8 k |
//   staging + barriers 13.6 k | transform of pair 0 (alone) 8.2 k + 3.8 k | the three multiply phases 47.6 k (4 k each, for 576 cycles
//   of MFMA and ~520 of transform) | epilogue 10.8 k.  A multiply phase moves 96 KB of weights into the CU: 24 B/clk, near what the
//   vector L1 delivered in the prototype (35 B/clk; 64 is its peak) -- every weight byte is used by 32 tiles only and is 1.5x as
//   large as in fp32, at 2.7x the matrix rate.  The next form needs 64 tiles per weight fetch or weights streamed through LDS.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"
#include "gn_tail.h"

using idiff_detail::ConvArgs;

namespace {

constexpr int CK = 4;             // gather granule (channels per staging pass), as conv_wino4.hip
constexpr int SUBS = 4;           // staging passes per chunk
constexpr int CKB = CK * SUBS;    // 16 channels per chunk
constexpr int TW = 32, TH = 16;
constexpr int RCOLS = TW + 2;
constexpr int RS = 40;
constexpr int TRH = TH + 2;
constexpr int PS = TRH * RS;      // 720
constexpr int PSP = 768;
constexpr int NT = 512;
constexpr int NL = 6;
constexpr int R_SUB = NL * NT;            // 3072 floats per staging pass
constexpr int R_FLOATS = SUBS * R_SUB;    // 12288 floats = 48 KB
constexpr int VROW = 6 * 3 * 2 * 512;     // bytes of one Winograd row of V: [v][plane][octet][tile 32] x 16 B
constexpr int UPOS = 3 * 2 * 64 * 16;     // bytes of one position of a (chunk, 64-channel block) of the weight image: 6144
constexpr int UBLK = 36 * UPOS;           // 221184

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Geo4x {
    int np;
    int total;
};

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
template <typename T>
__device__ __forceinline__ T* scalar_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
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
__device__ __forceinline__ void at6(const float x0, const float x1, const float x2, const float x3, const float x4, const float x5, float (&o)[4]) {
    const float s1 = x1 + x2, d1 = x1 - x2, s2 = x3 + x4, d2 = x3 - x4;
    o[0] = (x0 + s1) + s2;
    o[1] = __builtin_fmaf(2.f, d2, d1);
    o[2] = __builtin_fmaf(4.f, s2, s1);
    o[3] = __builtin_fmaf(8.f, d2, d1) + x5;
}
__host__ __device__ constexpr unsigned hi16(unsigned u) { return u & 0xffff0000u; }
__device__ __forceinline__ unsigned pack_hi(unsigned odd, unsigned even) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }
// x -> the bit patterns whose upper halves are the three bf16 slices of x (x = s1 + s2 + s3 exactly)
__device__ __forceinline__ void split3(float x, unsigned& s1, unsigned& s2, unsigned& s3) {
    s1 = __builtin_bit_cast(unsigned, x);
    const float r1 = x - __builtin_bit_cast(float, hi16(s1));
    s2 = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, hi16(s2));
    s3 = __builtin_bit_cast(unsigned, r2);
}
__device__ __forceinline__ floatx4 mma(const uintx4& a, const uintx4& b, const floatx4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef W4X_TRACE  // per-phase cycle counts of wave 0 (s_memtime), summed over the items, printed by the launcher (debug builds)
#define TR_PARAM , long long* trace
#define TR_INIT long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = __builtin_readcyclecounter();
#define TR_MARK(k)                                         \
    {                                                      \
        const long long t_ = __builtin_readcyclecounter(); \
        tr_acc[k] += t_ - tr_t;                            \
        tr_t = t_;                                         \
    }
#define TR_FINI \
    if (tid == 0) for (int q_ = 0; q_ < 8; ++q_) atomicAdd((unsigned long long*)trace + q_, (unsigned long long)tr_acc[q_]);
#else
#define TR_PARAM
#define TR_INIT
#define TR_MARK(k)
#define TR_FINI
#endif

// SPEC: 1 = single source, no prologue; 2 = single source + GN/FiLM/SiLU prologue; 3 = two sources (virtual concat)
template <int MODE, int SPEC, bool RAG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4x_kernel(const ConvArgs a, const Geo4x g, const unsigned char* __restrict__ wimg TR_PARAM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const econst = smem;                                        // [4][64]
    float* const Rb = smem + 256;                                      // [SUBS][R_SUB]; the epilogue's exchange area afterwards
    unsigned char* const Vb = reinterpret_cast<unsigned char*>(Rb + R_FLOATS);  // [2 pair buffers][2 rows][VROW]
    int* const gtab = reinterpret_cast<int*>(Vb + 4 * VROW);           // [NL][NT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kgl = lane >> 4;
    const int cq = wave & 3;    // MFMA role: 16-channel block
    const int vh = wave >> 2;   // MFMA role: Winograd columns 3 vh .. 3 vh + 2; transform role: row parity
    const int HWin = a.Hin * a.Win;
    const int nsub = a.Cin / CK;                 // staging passes in all (Cin % 8 == 0)
    const int nchunks = (a.Cin + CKB - 1) / CKB;

    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rs0, rs1;
    unsigned omask = 0;
    int it_b = 0, it_cob = 0, it_px = 0, it_py = 0, it_co0 = 0, it_y0 = 0, it_x0 = 0;
    const int tiles_y = g.np / a.tiles_x;
    auto decode = [&](int item) {
        it_cob = item % a.ncob;
        const int r1 = item / a.ncob;
        it_px = r1 % a.tiles_x;
        const int r2 = r1 / a.tiles_x;
        it_py = r2 % tiles_y;
        it_b = r2 / tiles_y;
    };
    auto setup_item = [&]() {
        it_co0 = it_cob * 64;
        it_y0 = it_py * TH;
        it_x0 = it_px * TW;
        rs0 = __builtin_amdgcn_make_buffer_rsrc(scalar_ptr(a.src0 + (long long)it_b * a.bs0), 0, 0x7fffffff, RSRC_FLAGS);
        rs1 = __builtin_amdgcn_make_buffer_rsrc(scalar_ptr(SPEC == 3 ? a.src1 + (long long)it_b * a.bs1 : a.src0), 0, 0x7fffffff, RSRC_FLAGS);
        omask = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * NT;
            const int ci = e / PSP;
            const int rem = e - ci * PSP;
            const int r = rem / RS;
            const int c = rem - r * RS;
            const int oy = it_y0 - 1 + r, ox = it_x0 - 1 + c;
            const bool in = rem < PS && c < RCOLS && (unsigned)oy < (unsigned)a.Hout && (unsigned)ox < (unsigned)a.Wout;
            const int sp = MODE == IDIFF_CONV_UPSAMPLE2 ? (oy >> 1) * a.Win + (ox >> 1) : oy * a.Win + ox;
            gtab[i * NT + tid] = in ? (ci * HWin + sp) * 4 : -1;
            omask |= (in ? 0u : 1u) << i;
        }
    };
    typedef const __attribute__((address_space(4))) floatx4* cfloatx4p;
    // staging in passes of 4 channels (sub -> R[slot]); beyond Cin: zeros (their weights are zero too, but LDS must hold finite
    // values).  The next chunk's 24 raw elements travel in registers during the last multiply phase of the current one.
    float rawr[SUBS][NL];
    auto load_chunk = [&](int cc) {
#pragma unroll
        for (int s = 0; s < SUBS; ++s) {
            const int sub = cc * SUBS + s;
            const bool live = sub < nsub;
            const int cbase = sub * CK;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int goff = gtab[i * NT + tid] | (live ? 0 : -1);  // branch-free: a dead pass reads through offset -1 (0.0, no access)
                if (SPEC == 3 && cbase >= a.C0v) rawr[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, goff, (cbase - a.C0v) * HWin * 4, 0));
                else rawr[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, goff, cbase * HWin * 4, 0));
            }
        }
    };
    auto write_chunk = [&](int cc, int b) {
#pragma unroll
        for (int s = 0; s < SUBS; ++s) {
            const int sub = cc * SUBS + s;
            const bool live = sub < nsub;
            const int cbase = sub * CK;
            floatx4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
            if (SPEC == 2 && live) {
                const long long o = (long long)b * a.C0r + cbase;
                pa = *(cfloatx4p)(a.pro_a + o);
                pb = *(cfloatx4p)(a.pro_b + o);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                float v = rawr[s][i];
                if (SPEC == 2) {  // channel of element tid + i*512: ((i*512 + wave*64) / 768), wave-uniform
                    const int ch = (i * NT + wave * 64) / PSP;
                    const float fa = ch == 0 ? pa.x : ch == 1 ? pa.y : ch == 2 ? pa.z : pa.w;
                    const float fb = ch == 0 ? pb.x : ch == 1 ? pb.y : ch == 2 ? pb.z : pb.w;
                    v = silu_fast(fa * v + fb);
                    if (((omask >> i) & 1u) || !live) v = 0.f;  // padding is zero AFTER the activation
                }
                Rb[s * R_SUB + tid + i * NT] = v;
            }
        }
    };

    // ---- transform role: thread = (tile, ONE channel of the chunk), BOTH Winograd rows of a pair: the two rows share their patch rows
    // (five at most), each read once and folded into the two B^T-along-the-rows sums with compile-time coefficients.  Eleven slices,
    // dealt out between the MFMA units of the previous row pair (every slice's LDS read is consumed one slice later); the outputs
    // are split and stored as single bf16 (the two channels of a dword come from lanes 32 apart).
    const int tile = tid & 31, tch = tid >> 5;  // channel tch (0..15) of the chunk
    const int tty = tile >> 3, ttx = tile & 7;
    const float* const trb = Rb + (tch >> 2) * R_SUB + (tch & 3) * PSP + (4 * tty) * RS + 4 * ttx;
    unsigned char* const vwr0 = Vb + (tch >> 3) * 512 + tile * 16 + (tch & 7) * 2;  // + (2 * pair buffer + row) * VROW + ((v * 3 + plane) * 2) * 512
    float t_rw[6], t_wa[6], t_wb[6], t_o[6];
    auto t_slice = [&](int k, auto up_tag, int pbw) {
        constexpr int UP = decltype(up_tag)::value;
        // rows of B^T (F(4x4,3x3)): 0: 4 d0 - 5 d2 + d4   1: -4 d1 - 4 d2 + d3 + d4   2: 4 d1 - 4 d2 - d3 + d4   3: -2 d1 - d2 + 2 d3 + d4
        //                           4: 2 d1 - d2 - 2 d3 + d4   5: 4 d1 - 5 d3 + d5
        constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
        constexpr int R0 = UP == 0 ? 0 : 1;  // first of the five patch rows the pair touches (pair 1 touches four: the fifth has zero weights)
        auto rd = [&](int r) {
            const float* p = trb + r * RS;
            const floatx4 lo = *reinterpret_cast<const floatx4*>(p);
            const floatx2 hi = *reinterpret_cast<const floatx2*>(p + 4);
            t_rw[0] = lo.x, t_rw[1] = lo.y, t_rw[2] = lo.z, t_rw[3] = lo.w, t_rw[4] = hi.x, t_rw[5] = hi.y;
        };
        auto fold = [&](int r, bool first) {
            const float ca = BT[2 * UP][r], cb = BT[2 * UP + 1][r];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                t_wa[c] = first ? ca * t_rw[c] : (ca != 0.f ? __builtin_fmaf(ca, t_rw[c], t_wa[c]) : t_wa[c]);
                t_wb[c] = first ? cb * t_rw[c] : (cb != 0.f ? __builtin_fmaf(cb, t_rw[c], t_wb[c]) : t_wb[c]);
            }
        };
        auto emit = [&](int rr, int v) {
            unsigned s1, s2, s3;
            split3(t_o[v], s1, s2, s3);
            unsigned char* const q = vwr0 + (2 * pbw + rr) * VROW + ((v * 3) * 2) * 512;
            *reinterpret_cast<unsigned short*>(q) = (unsigned short)(s1 >> 16);
            *reinterpret_cast<unsigned short*>(q + 2 * 512) = (unsigned short)(s2 >> 16);
            *reinterpret_cast<unsigned short*>(q + 4 * 512) = (unsigned short)(s3 >> 16);
        };
        if (k == 0) rd(R0);
        if (k == 1) fold(R0, true), rd(R0 + 1);
        if (k == 2) fold(R0 + 1, false), rd(R0 + 2);
        if (k == 3) fold(R0 + 2, false), rd(R0 + 3);
        if (k == 4) fold(R0 + 3, false), rd(R0 + 4);
        if (k == 5) fold(R0 + 4, false), bt6(t_wa, t_o);
        if (k == 6) emit(0, 0), emit(0, 1), emit(0, 2);
        if (k == 7) emit(0, 3), emit(0, 4), emit(0, 5);
        if (k == 8) bt6(t_wb, t_o);
        if (k == 9) emit(1, 0), emit(1, 1), emit(1, 2);
        if (k == 10) emit(1, 3), emit(1, 4), emit(1, 5);
    };

    const int G = gridDim.x;
    const int first = (int)xcd_remap(blockIdx.x, G);
    const int last = g.total;
    if (first >= last) return;

    // operand addresses of the MFMA role (bytes): lane (n16, kgl) reads octet kgl & 1 of plane X (kgl < 2) or Y (kgl >= 2)
    const int oct = kgl & 1, second = kgl >> 1;
    const int a_off0 = (((second ? 1 : 0) * 2 + oct) * 64 + 16 * cq + n16) * 16;  // [u1 | u2]
    const int a_off1 = (((second ? 2 : 0) * 2 + oct) * 64 + 16 * cq + n16) * 16;  // [u1 | u3]
    const int b_off1 = ((0 * 2 + oct) * 32 + n16) * 16;                           // [v1 | v1]   (+ tile block * 256)
    const int b_off2 = ((1 * 2 + oct) * 32 + n16) * 16;                           // [v2 | v2]
    const int b_off3 = (((second ? 0 : 2) * 2 + oct) * 32 + n16) * 16;            // [v3 | v1]

    TR_INIT
    for (int item = first; item < last; item += G) {
        TR_MARK(7)
        decode(item);
        __syncthreads();  // every wave is done with the previous item's LDS (exchange area, gather table)
        setup_item();
        const int b = it_b, co0 = it_co0, y0 = it_y0, x0 = it_x0;
        if (tid < 256) {
            const int which = tid >> 6, co = co0 + (tid & 63);
            float pre = 0.f;
            if (co < a.Cout) {
                if (which == 0 && a.bias) pre = a.bias[co];
                if (which == 1 && a.vec) pre = a.vec[(long long)b * a.Cout + co];
                if (which == 2 && a.aux) pre = a.aux_a[(long long)b * a.Cout + co];
                if (which == 3 && a.aux) pre = a.aux_b[(long long)b * a.Cout + co];
            }
            econst[tid] = pre;
        }
        floatx4 acc[18][2];  // [u * 3 + vl][tile block]
#pragma unroll
        for (int p = 0; p < 18; ++p) acc[p][0] = acc[p][1] = floatx4{0.f, 0.f, 0.f, 0.f};
        // ---- main loop.  Per chunk of 16 channels (R staged):  T(pair 0) alone | M(pair 0) with T(pair 1) dealt out between its twelve
        // units | M(pair 1) with T(pair 2) | M(pair 2) with the next chunk's pixels in flight | stage R.  A unit = one position x one
        // tile block = three MFMAs; its B operands are requested one unit ahead, the A operands three positions ahead (rolling
        // window of three positions: L2 latency), V pair buffers alternate P, P^1, P.
        auto a_pofs = [&](int cc, int up, int j) {  // byte offset of position j (0..5) of row pair up of chunk cc in the weight image
            const int ccl = cc < nchunks ? cc : nchunks - 1;  // past the end: a harmless re-read
            return __builtin_amdgcn_readfirstlane((ccl * a.ncob + it_cob) * UBLK + ((2 * up + j / 3) * 6 + 3 * vh + j % 3) * UPOS);
        };
        uintx4 aw[3][2];
        auto load_aw = [&](int slot, int pofs) {
            const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(scalar_ptr(wimg)), 0, 0x7fffffff, RSRC_FLAGS);
            aw[slot][0] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, a_off0, pofs, 0));
            aw[slot][1] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsu, a_off1, pofs, 0));
        };
        uintx4 bq[2][3];
        auto load_b = [&](int set, int pbr, int n) {  // unit n = 2 j + tb of the pair buffer pbr
            const int j = n >> 1, tb = n & 1;
            const unsigned char* const vb = Vb + (2 * pbr + j / 3) * VROW + (3 * vh + j % 3) * (3 * 2 * 512) + tb * 256;
            bq[set][0] = *reinterpret_cast<const uintx4*>(vb + b_off1);
            bq[set][1] = *reinterpret_cast<const uintx4*>(vb + b_off2);
            bq[set][2] = *reinterpret_cast<const uintx4*>(vb + b_off3);
        };
        // TK: 0 = transform the next pair (up + 1) into the other pair buffer; 1 = nothing; 2 = request the next chunk's pixels first
        auto mphase = [&](int cc, auto up_tag, auto pbr_tag, auto tk_tag) {
            constexpr int up = decltype(up_tag)::value, PBR = decltype(pbr_tag)::value, TK = decltype(tk_tag)::value;
            load_b(0, PBR, 0);
            if (TK == 2) load_chunk(cc + 1);  // (offset -1 past the last chunk: no access) -- behind the window's A loads, in order
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                const int j = n >> 1, tb = n & 1, u = 2 * up + j / 3, vl = j % 3;
                if (n + 1 < 12) load_b((n + 1) & 1, PBR, n + 1);
                floatx4 c = acc[u * 3 + vl][tb];
                c = mma(aw[j % 3][1], bq[n & 1][2], c);
                c = mma(aw[j % 3][0], bq[n & 1][1], c);
                c = mma(aw[j % 3][0], bq[n & 1][0], c);
                acc[u * 3 + vl][tb] = c;
                if (tb == 1) {  // the position is done: its window slot takes the position three ahead
                    if (j + 3 < 6) load_aw(j % 3, a_pofs(cc, up, j + 3));
                    else if (up < 2) load_aw(j % 3, a_pofs(cc, up + 1, j - 3));
                    else load_aw(j % 3, a_pofs(cc + 1, 0, j - 3));
                }
                if (TK == 0 && n < 11) t_slice(n, std::integral_constant<int, (up + 1) % 3>{}, PBR ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        load_chunk(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) load_aw(j, a_pofs(0, 0, j));
        TR_MARK(0)
        for (int cc = 0; cc < nchunks; ++cc) {
            __syncthreads();  // the previous chunk's transforms have read R, its last multiply phase the pair buffer P
            write_chunk(cc, b);
            __syncthreads();
            TR_MARK(1)
#pragma unroll
            for (int k = 0; k < 11; ++k) t_slice(k, I0{}, 0);
            TR_MARK(2)
            __syncthreads();
            TR_MARK(3)
            mphase(cc, I0{}, I0{}, I0{});
            __syncthreads();
            mphase(cc, I1{}, I1{}, I0{});
            __syncthreads();
            mphase(cc, I2{}, I0{}, I2{});
            TR_MARK(4)
        }
        TR_MARK(5)
        // ---- epilogue.  C layout: lane holds tile n16 of tile block tb and channels 4 kgl + r of the wave's 16-channel block.
        // Phase 1: A^T along u for the wave's three columns, both tile blocks: t[tb][vl][r][dy]
        // Phase X: the wave finishes tile block tb = vh; the other block's t goes to the partner wave (cq, 1 - vh) through LDS, in two
        //          rounds of two channels (48 KB each, in the idle R area)
        // Phase 2: A^T along v over all six columns, then the conv_wino4.hip epilogue on the wave's 8x32 half-patch
        float tk[3][4][4];   // [vl][r][dy] of the block the wave keeps
        float tp[3][4][4];   // ... received from the partner: its columns 3 (1 - vh) + vl of the same block
        {
            float ts[3][4][4];  // the block sent away
#pragma unroll
            for (int vl = 0; vl < 3; ++vl)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o0[4], o1[4];
                    at6(acc[0 * 3 + vl][0][r], acc[1 * 3 + vl][0][r], acc[2 * 3 + vl][0][r], acc[3 * 3 + vl][0][r], acc[4 * 3 + vl][0][r], acc[5 * 3 + vl][0][r], o0);
                    at6(acc[0 * 3 + vl][1][r], acc[1 * 3 + vl][1][r], acc[2 * 3 + vl][1][r], acc[3 * 3 + vl][1][r], acc[4 * 3 + vl][1][r], acc[5 * 3 + vl][1][r], o1);
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy) {
                        tk[vl][r][dy] = vh ? o1[dy] : o0[dy];
                        ts[vl][r][dy] = vh ? o0[dy] : o1[dy];
                    }
                }
            float* const xw = Rb + ((1 - vh) * 4 + cq) * (24 * 64) + lane;  // addressed to the partner's wave index
            const float* const xr = Rb + (vh * 4 + cq) * (24 * 64) + lane;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __syncthreads();  // R (main loop) / the previous round has been read
#pragma unroll
                for (int vl = 0; vl < 3; ++vl)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy) xw[((vl * 2 + r2) * 4 + dy) * 64] = ts[vl][2 * half + r2][dy];
                __syncthreads();
#pragma unroll
                for (int vl = 0; vl < 3; ++vl)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy) tp[vl][2 * half + r2][dy] = xr[((vl * 2 + r2) * 4 + dy) * 64];
            }
        }
        const int tblk = vh;  // the wave's 8x32 half-patch
        if (co0 + cq * 16 < a.Cout) {
            const int j = n16, k4 = kgl;
            const int HWo = a.Hout * a.Wout;
            const int ty0 = y0 + 8 * tblk;
            const long long wave_org = (long long)(co0 + cq * 16) * HWo + (long long)ty0 * a.Wout + x0;
            float* const outb = a.out + (long long)b * a.obs + wave_org;
            const float* const resb = a.res ? a.res + (long long)b * a.rbs + wave_org : nullptr;
            const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ + wave_org : nullptr;
            const unsigned lane_off = (unsigned)(4 * k4) * (unsigned)HWo + (unsigned)(4 * (j >> 3)) * (unsigned)a.Wout + 4u * (j & 7);
            const float* const ebase = econst + cq * 16 + 4 * k4;
            const bool want_stats = a.stats != nullptr && ty0 < a.Hout;
            const bool has_res = a.res != nullptr, has_aux = a.aux != nullptr;
            const bool inside = !RAG || ((ty0 + 4 * (j >> 3) < a.Hout) && (x0 + 4 * (j & 7) < a.Wout));
            float* const stp = want_stats ? a.stats + (((long long)b * a.ntiles + (ty0 >> 3) * a.tiles_x + (x0 >> 5)) * a.Cout + co0 + cq * 16 + 4 * k4) * 2 : nullptr;
            floatx4 nres = floatx4{0.f, 0.f, 0.f, 0.f}, naux = nres;
            auto fetch = [&](int s) {
                if (!inside) return;
                const long long so = (long long)(s >> 2) * HWo + (s & 3) * a.Wout;
                if (has_res) nres = *reinterpret_cast<const floatx4*>(resb + so + lane_off);
                if (has_aux) naux = *reinterpret_cast<const floatx4*>(auxb + so + lane_off);
            };
            fetch(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = ebase[r];
                float y[4][4];  // [dy][dx]
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    // columns in natural order: v = 0..2 from the wave with vh = 0, 3..5 from the wave with vh = 1
                    const float c0 = vh ? tp[0][r][dy] : tk[0][r][dy], c1 = vh ? tp[1][r][dy] : tk[1][r][dy], c2 = vh ? tp[2][r][dy] : tk[2][r][dy];
                    const float c3 = vh ? tk[0][r][dy] : tp[0][r][dy], c4 = vh ? tk[1][r][dy] : tp[1][r][dy], c5 = vh ? tk[2][r][dy] : tp[2][r][dy];
                    float row[4];
                    at6(c0, c1, c2, c3, c4, c5, row);
#pragma unroll
                    for (int x = 0; x < 4; ++x) y[dy][x] = row[x] + bv;
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
                    if (j == 15) idiff_detail::gn_store_partial(stp + 2 * r, ssum, ssq);
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
            }
        }
        TR_MARK(6)
    }
    TR_FINI
}

template <int MODE, int SPEC, bool RAG>
int launch_rag(const ConvArgs& a, const void* wimg, hipStream_t st) {
    const size_t lds = 256 * sizeof(float) + (size_t)R_FLOATS * sizeof(float) + 4 * VROW + (size_t)NL * NT * sizeof(int);
    static bool attr_set = false;
    auto kern = conv_wino4x_kernel<MODE, SPEC, RAG>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4x): hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            IDIFF_FAIL(IDIFF_E_HIP, "conv2d(winograd4x): cannot query the CU count");
        num_cu = n;
    }
    Geo4x g;
    g.np = a.tiles_x * ((a.Hout + TH - 1) / TH);
    const long long total = (long long)a.B * g.np * a.ncob;
    if (total >= (1ll << 31)) IDIFF_FAIL(IDIFF_E_BADARG, "conv2d(winograd4x): grid too large");
    g.total = (int)total;
    const int per = (g.total + num_cu - 1) / num_cu;
    const int grid = (g.total + per - 1) / per;
#ifdef W4X_TRACE
    static long long* tr = nullptr;
    if (!tr) (void)hipMalloc(&tr, 8 * sizeof(long long));
    (void)hipMemsetAsync(tr, 0, 8 * sizeof(long long), st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, g, static_cast<const unsigned char*>(wimg), tr);
    long long h[8];
    (void)hipMemcpyAsync(h, tr, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[wino4x trace] Cin=%d Cout=%d H=%d items=%d | setup+load0 %lld  stage+bars %lld  transform %lld  bar(T) %lld  multiply %lld  bar(M) %lld  epilogue %lld  item-top %lld (cycles/item, wave 0)\n",
            a.Cin, a.Cout, a.Hout, g.total, h[0] / g.total, h[1] / g.total, h[2] / g.total, h[3] / g.total, h[4] / g.total, h[5] / g.total, h[6] / g.total, h[7] / g.total);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a, g, static_cast<const unsigned char*>(wimg));
#endif
    IDIFF_CHECK_LAUNCH("conv2d_fwd(winograd4x)");
    return IDIFF_OK;
}

template <int MODE, int SPEC>
int launch(const ConvArgs& a, const void* wimg, hipStream_t st) {
    if (a.Hout % TH || a.Wout % TW) return launch_rag<MODE, SPEC, true>(a, wimg, st);
    return launch_rag<MODE, SPEC, false>(a, wimg, st);
}

// image: [chunk of 16 ci][block of 64 co][position 6u+v][plane 3][octet 2][co 64][8 bf16] of U = G g G^T (fp32, as conv_wino4.hip)
__global__ void pack_wino4x_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin, int transpose) {
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    const int ncob = (Co + 63) / 64;
    const int Cip = ((Ci + CKB - 1) / CKB) * CKB;
    const long long n = (long long)ncob * 64 * Cip;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % (ncob * 64)), ci = (int)(i / (ncob * 64));
        float U[36];
        if (co < Co && ci < Ci) {
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
                const float t = __builtin_fmaf(4.f, x2, x0);
                o[3] = (1.f / 24.f) * __builtin_fmaf(2.f, x1, t);
                o[4] = (1.f / 24.f) * __builtin_fmaf(-2.f, x1, t);
                o[5] = x2;
            };
            float t[6][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float o[6];
                g6(gk[0][q], gk[1][q], gk[2][q], o);
#pragma unroll
                for (int u = 0; u < 6; ++u) t[u][q] = o[u];
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                float o[6];
                g6(t[u][0], t[u][1], t[u][2], o);
#pragma unroll
                for (int v = 0; v < 6; ++v) U[u * 6 + v] = o[v];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 36; ++p) U[p] = 0.f;
        }
        const int cc = ci / CKB, o8 = (ci % CKB) >> 3, e = ci & 7;
        const int cob = co >> 6, col = co & 63;
        unsigned short* const dst = out + ((long long)(cc * ncob + cob) * UBLK) / 2 + (o8 * 64 + col) * 8 + e;
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            unsigned s1, s2, s3;
            split3(U[p], s1, s2, s3);
            dst[(p * UPOS) / 2 + 0 * (2 * 64 * 8)] = (unsigned short)(s1 >> 16);
            dst[(p * UPOS) / 2 + 1 * (2 * 64 * 8)] = (unsigned short)(s2 >> 16);
            dst[(p * UPOS) / 2 + 2 * (2 * 64 * 8)] = (unsigned short)(s3 >> 16);
        }
    }
}

}  // namespace

namespace idiff_detail {

int launch_conv_wino4x(const ConvArgs& a, int mode, const void* wimg, hipStream_t st) {
    if ((long long)((a.Cin + CKB - 1) / CKB) * a.ncob * UBLK >= (1ll << 31)) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d(winograd4x): weight image beyond 2 GB");
    if (mode == IDIFF_CONV_UPSAMPLE2) return launch<IDIFF_CONV_UPSAMPLE2, 1>(a, wimg, st);
    if (a.pro_a) return launch<IDIFF_CONV_NORMAL, 2>(a, wimg, st);
    if (a.src1) return launch<IDIFF_CONV_NORMAL, 3>(a, wimg, st);
    return launch<IDIFF_CONV_NORMAL, 1>(a, wimg, st);
}

}  // namespace idiff_detail

extern "C" long long idiff_conv_wino4x_image_bytes(int Cout, int Cin) {
    return (long long)((Cin + CKB - 1) / CKB) * ((Cout + 63) / 64) * UBLK;
}

extern "C" int idiff_pack_conv_weight_wino4x(const float* w, void* image, int Cout, int Cin, int transpose, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(w && image && Cout > 0 && Cin > 0, "pack_conv_weight_wino4x: bad args");
    const int Co = transpose ? Cin : Cout, Ci = transpose ? Cout : Cin;
    IDIFF_CHECK_ARG(Co % 16 == 0 && Ci % 8 == 0, "pack_conv_weight_wino4x: needs conv Cout %% 16 == 0 and Cin %% 8 == 0 (got %d, %d)", Co, Ci);
    IDIFF_CHECK_ARG((reinterpret_cast<uintptr_t>(image) & 15) == 0, "pack_conv_weight_wino4x: image must be 16-byte aligned");
    const long long n = (long long)((Co + 63) / 64) * 64 * (((Ci + CKB - 1) / CKB) * CKB);
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_wino4x_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, static_cast<unsigned short*>(image), Cout, Cin, transpose);
    IDIFF_CHECK_LAUNCH("pack_conv_weight_wino4x");
    return IDIFF_OK;
}
