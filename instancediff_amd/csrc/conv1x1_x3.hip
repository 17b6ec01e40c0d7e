// 1x1 convolution (channel-mixing GEMM) on the bf16 matrix cores with fp32 operands split three ways (gfx950,
// v_mfma_f32_16x16x32_bf16).
//
//   An fp32 value is EXACTLY the sum of three bf16 values: its top 8 significand bits, the next 8 and the last 8
//   (x1 = trunc16(x), x2 = trunc16(x - x1), x3 = x - x1 - x2: both subtractions are exact).  A product a*b is then the sum of nine
//   bf16 x bf16 products, each of which the matrix core forms exactly and accumulates in fp32; the six of them of order >= 2^-16
//   (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) carry the product to 2^-24 relative -- the class of a single fp32 rounding (measured on
//   random operands against fp64: 2.4e-7 of the output range, fp32 fma chain: 3.8e-7).  Six bf16 MFMAs of K = 32 replace eight f32
//   MFMAs of K = 4 at a sixteenth of the cycles per flop: 2.67x the matrix throughput of conv_igemm.hip's f32 path, which on these
//   layers (128->64 at 256^2 ... 576->256 at 32^2) was matrix-bound at 0.4-0.58 of the f32 peak, i.e. at 2-4x their HBM time.
//
//   GEMM view as conv_igemm.hip's flattened 1x1 tiles: M = 64 output channels, N = 256 consecutive pixels of one sample, K = Cin in
//   chunks of 32.  256 threads; wave w owns pixels [64w, 64w+64) x all 64 channels = 4x4 accumulator blocks of 16x16.
//   Staging (thread = pixel quad q x channel octet kg): eight 16-byte loads (one per channel, 1 KB contiguous per wave), the 3-way
//   split in registers (4 bit-ops/subtractions + 1.5 byte-permutes per value) and, per pixel, one ds_write_b128 per plane: the eight
//   channels of a pixel are exactly the K-octet a lane of the B operand holds.  Weights come pre-split (idiff_pack_conv1x1_x3) and
//   are copied verbatim.  LDS: B planes 51 KB + A planes 12 KB, single buffer (the next two chunks travel in 76 registers) -> two
//   workgroups per CU, one splitting while the other multiplies.  Epilogue = conv_igemm.hip's flattened form (accumulators through
//   LDS as [32 co][256 px] halves, 16-byte row segments; bias, per-(b,c) vector, residual, "+ silu(a*aux+b)"); no GroupNorm partials.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

using idiff_detail::ConvArgs;

namespace {

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 64, CKB = 32, NT = 256;
constexpr int B_ISTR = 1088;             // bytes between the four pixel-in-quad images of an octet (64 quads x 16 B + 64: the 16 lanes
                                         // of an operand read -- 4 pixels-in-quad x 4 quads -- then fall on 16 distinct 16-byte bank groups)
constexpr int B_KGSTR = 4 * B_ISTR;      // 4352: channel octet
constexpr int B_PLSTR = 4 * B_KGSTR;     // 17408: plane
constexpr int B_BYTES = 3 * B_PLSTR;     // 52224
constexpr int A_PLSTR = 4 * 64 * 16;     // 4096: [kg][co 64] x 16 B
constexpr int A_BYTES = 3 * A_PLSTR;     // 12288 = one (chunk, 64-channel block) of the weight image
constexpr int OLD = 256 + 4;             // epilogue tile pitch (floats)
static_assert(32 * OLD * 4 <= B_BYTES, "epilogue tile must fit in the B planes");

__host__ __device__ constexpr unsigned hi16(unsigned u) { return u & 0xffff0000u; }

// upper halves of two dwords -> one dword of two bf16 (element 2d in the low half)
__device__ __forceinline__ unsigned pack_hi(unsigned odd, unsigned even) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }

// x[0..7] (fp32) -> three planes of eight bf16: x = p1 + p2 + p3 exactly
__device__ __forceinline__ void split3(const float (&x)[8], uintx4& p1, uintx4& p2, uintx4& p3) {
    unsigned u1[8], u2[8], u3[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        u1[c] = __builtin_bit_cast(unsigned, x[c]);
        const float r1 = x[c] - __builtin_bit_cast(float, hi16(u1[c]));
        u2[c] = __builtin_bit_cast(unsigned, r1);
        const float r2 = r1 - __builtin_bit_cast(float, hi16(u2[c]));
        u3[c] = __builtin_bit_cast(unsigned, r2);
    }
    p1 = uintx4{pack_hi(u1[1], u1[0]), pack_hi(u1[3], u1[2]), pack_hi(u1[5], u1[4]), pack_hi(u1[7], u1[6])};
    p2 = uintx4{pack_hi(u2[1], u2[0]), pack_hi(u2[3], u2[2]), pack_hi(u2[5], u2[4]), pack_hi(u2[7], u2[6])};
    p3 = uintx4{pack_hi(u3[1], u3[0]), pack_hi(u3[3], u3[2]), pack_hi(u3[5], u3[4]), pack_hi(u3[7], u3[6])};
}

// a wave-uniform pointer pinned to scalar registers (a resource base left in VGPRs costs a waterfall loop per buffer load)
template <typename T>
__device__ __forceinline__ T* scalar_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ floatx4 mma(const uintx4& a, const uintx4& b, const floatx4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// UNSH: pixel-unshuffle(2) gather (IDIFF_CONV_UNSHUFFLE2, single source): virtual channel 4c + 2dy + dx of output pixel (oy, ox) is input
// pixel (2oy + dy, 2ox + dx) of real channel c.  An octet = two real channels; a thread's four output pixels are eight consecutive
// input pixels of two rows: 2 channels x 2 rows x 2 halves = the same eight 16-byte loads, re-indexed in registers.
template <bool UNSH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1x1_x3_kernel(const ConvArgs a, const unsigned short* __restrict__ wx3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const Bs = smem_raw;
    unsigned char* const As = smem_raw + B_BYTES;
    float* const econst = reinterpret_cast<float*>(smem_raw + B_BYTES + A_BYTES);  // [4][64] bias, vec, aux_a, aux_b

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kgl = lane >> 4;

    const unsigned logical = xcd_remap(blockIdx.x, a.total_wg);
    const int cob = logical % a.ncob;
    const int tile = (logical / a.ncob) % a.ntiles;
    const int b = logical / (a.ncob * a.ntiles);
    const int co0 = cob * BM;
    const long long HW = (long long)a.Hout * a.Wout;  // output plane (normal mode: flattened by the caller, Hout == 1)
    const long long pix0 = (long long)tile * 256;

    {
        const int which = tid >> 6, co = co0 + (tid & 63);
        float v = 0.f;
        if (which == 0 && a.bias) v = a.bias[co];
        if (which == 1 && a.vec) v = a.vec[(long long)b * a.Cout + co];
        if (which == 2 && a.aux) v = a.aux_a[(long long)b * a.Cout + co];
        if (which == 3 && a.aux) v = a.aux_b[(long long)b * a.Cout + co];
        econst[tid] = v;
    }

    // ---- staging role: pixel quad q, channel octet kg (= wave: the source of an octet is wave-uniform) ------------------------------
    const int q = lane, kg = wave;
    const int nchunks = (a.Cin + CKB - 1) / CKB;
    floatx4 xrA[8], xrB[8];  // raw channels of the next two chunks (even / odd): loads stay in flight across a whole chunk of MFMAs
    uintx4 ar[3];
    // Branch-free: a uniform branch around loads costs hipcc's wait counts their precision (every wait becomes vmcnt(0) and the far
    // prefetch is drained at each stage).  A chunk that does not exist, or an octet beyond Cin, is loaded through a buffer resource
    // with zero records: no memory traffic, the registers read 0.0.  An octet lies in one source (C0v % 8 == 0) and is whole or
    // beyond Cin (Cin % 8 == 0).
    constexpr int RSRC_FLAGS = 0x00020000;
    const float* const s0 = a.src0 + (long long)b * a.bs0 + (UNSH ? 0 : pix0);
    const float* const s1 = a.src1 ? a.src1 + (long long)b * a.bs1 + pix0 : s0;
    const long long HWin = (long long)a.Hin * a.Win;
    const int chstride_b = (int)((UNSH ? HWin : HW) * 4);  // bytes between (real) channels (8 of them < 2^31: checked by the launcher)
    int xoff = q * 16;                                      // the lane's byte offset inside a channel plane
    if (UNSH) {
        const int p = (int)pix0 + 4 * q;
        const int oy = p / a.Wout, ox = p - oy * a.Wout;
        xoff = (2 * oy * a.Win + 2 * ox) * 4;
    }
    auto load_x = [&](floatx4 (&xr)[8], int cc) {
        const int ch0 = cc * CKB + kg * 8;  // uniform
        const bool valid = ch0 < a.Cin;     // also false for a chunk past the last one
        if (UNSH) {
            const float* const p = s0 + (long long)(valid ? (ch0 >> 2) : 0) * HWin;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(scalar_ptr(p), 0, __builtin_amdgcn_readfirstlane(valid ? 0x7fffffff : 0), RSRC_FLAGS);
#pragma unroll
            for (int c = 0; c < 8; ++c)  // c = (channel cl, row dy, half): the row's eight pixels as two 16-byte pieces
                xr[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff, (c >> 2) * chstride_b + ((c >> 1) & 1) * a.Win * 4 + (c & 1) * 16, 0));
        } else {
            const bool second = ch0 >= a.C0v;
            const float* const p = (second ? s1 : s0) + (long long)(valid ? (second ? ch0 - a.C0v : ch0) : 0) * HW;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(scalar_ptr(p), 0, __builtin_amdgcn_readfirstlane(valid ? 0x7fffffff : 0), RSRC_FLAGS);
#pragma unroll
            for (int c = 0; c < 8; ++c) xr[c] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff, c * chstride_b, 0));
        }
    };
    auto load_w = [&](int cc) {
        const bool valid = cc < nchunks;
        const unsigned short* const wsrc = wx3 + ((long long)(valid ? cc : 0) * a.ncob + cob) * (A_BYTES / 2);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(scalar_ptr(wsrc), 0, __builtin_amdgcn_readfirstlane(valid ? A_BYTES : 0), RSRC_FLAGS);
#pragma unroll
        for (int i = 0; i < 3; ++i) ar[i] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, i * NT * 16, 0));
    };
    unsigned char* const bw = Bs + kg * B_KGSTR + q * 16;
    auto stage = [&](const floatx4 (&xr)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (UNSH) {  // octet element c = 4 cl + 2 dy + dx  <-  row piece (cl, dy), pixel 2 i + dx of its eight
                    const int px = 2 * i + (c & 1);
                    v[c] = xr[(c >> 1) * 2 + (px >> 2)][px & 3];
                } else {
                    v[c] = xr[c][i];
                }
            }
            uintx4 p1, p2, p3;
            split3(v, p1, p2, p3);
            *reinterpret_cast<uintx4*>(bw + i * B_ISTR) = p1;
            *reinterpret_cast<uintx4*>(bw + i * B_ISTR + B_PLSTR) = p2;
            *reinterpret_cast<uintx4*>(bw + i * B_ISTR + 2 * B_PLSTR) = p3;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<uintx4*>(As)[tid + i * NT] = ar[i];
    };

    // ---- MFMA role ------------------------------------------------------------------------------------------------------------------
    floatx4 acc[4][4];  // [co block mi][pixel block j]: lane holds pixel n16 of block j, channels 16 mi + 4 kgl + r
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mi][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const ard = As + kgl * 1024 + n16 * 16;
    const unsigned char* const brd = Bs + kgl * B_KGSTR + (n16 & 3) * B_ISTR + (16 * wave + (n16 >> 2)) * 16;

    auto chunk = [&](int cc, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        floatx4(&xr)[8] = PAR ? xrB : xrA;
        __syncthreads();  // every wave has read the previous chunk's operands (and econst is visible)
        // the split below stays below: hoisted into the previous chunk's MFMAs (hipcc does, it is register-only arithmetic) it would
        // wait for this chunk's pixels a whole chunk early -- half the prefetch distance
        __builtin_amdgcn_sched_barrier(0);
        stage(xr);
        __syncthreads();
        load_w(cc + 1);      // L2 hits, one chunk ahead; requested BEFORE the pixels of chunk cc + 2: vmcnt is in order, so the next
        load_x(xr, cc + 2);  // stage then waits for these weights without draining those loads
        __builtin_amdgcn_sched_barrier(0);
        // two passes over the pixel blocks, two channel blocks each: 24 + 24 operand registers instead of 48 + 24 (the B operands are
        // read twice: 36 ds_read_b128 per wave and chunk for 96 MFMAs)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            uintx4 av[2][3];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int p = 0; p < 3; ++p) av[m2][p] = *reinterpret_cast<const uintx4*>(ard + p * A_PLSTR + (2 * mh + m2) * 256);
            uintx4 bv[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bv[0][p] = *reinterpret_cast<const uintx4*>(brd + p * B_PLSTR);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j + 1 < 4) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) bv[(j + 1) & 1][p] = *reinterpret_cast<const uintx4*>(brd + p * B_PLSTR + (j + 1) * 64);
                }
                const uintx4(&bj)[3] = bv[j & 1];
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    floatx4 c = acc[2 * mh + m2][j];
                    c = mma(av[m2][2], bj[0], c);  // smallest terms first
                    c = mma(av[m2][1], bj[1], c);
                    c = mma(av[m2][0], bj[2], c);
                    c = mma(av[m2][1], bj[0], c);
                    c = mma(av[m2][0], bj[1], c);
                    c = mma(av[m2][0], bj[0], c);
                    acc[2 * mh + m2][j] = c;
                }
            }
        }
    };
    load_x(xrA, 0);
    load_w(0);
    load_x(xrB, 1);
    int cc = 0;
    for (; cc + 1 < nchunks; cc += 2) {
        chunk(cc, std::integral_constant<int, 0>{});
        chunk(cc + 1, std::integral_constant<int, 1>{});
    }
    if (cc < nchunks) chunk(cc, std::integral_constant<int, 0>{});

    // ---- epilogue: [64 co][256 px] through LDS in two 32-channel halves, 16-byte row segments to memory (conv_igemm.hip, flattened form)
    float* const ot = reinterpret_cast<float*>(Bs);
    float* const outb = a.out + (long long)b * a.obs;
    const float* const resb = a.res ? a.res + (long long)b * a.rbs : nullptr;
    const float* const auxb = a.aux ? a.aux + (long long)b * a.abs_ : nullptr;
    auto rows = [&](int mb, auto res_tag, auto aux_tag) {
        constexpr bool RES = decltype(res_tag)::value, AUX = decltype(aux_tag)::value;
        constexpr int STEPS = 32 * 64 / NT;  // float4 per thread and half
        floatx4 nres = {0.f, 0.f, 0.f, 0.f}, naux = {0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int i) {
            const int f = tid + i * NT;
            const int row = mb * 32 + (f >> 6), c4 = (f & 63) * 4;
            const long long o = (long long)(co0 + row) * HW + pix0 + c4;
            if (RES) nres = *reinterpret_cast<const floatx4*>(resb + o);
            if (AUX) naux = *reinterpret_cast<const floatx4*>(auxb + o);
        };
        fetch(0);
        __syncthreads();  // the operands (main loop) / the previous half have been read
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ot[(mh * 16 + 4 * kgl + r) * OLD + 64 * wave + 16 * j + n16] = mb ? acc[2 + mh][j][r] : acc[mh][j][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int f = tid + i * NT;
            const int lrow = f >> 6, row = mb * 32 + lrow, c4 = (f & 63) * 4;
            const floatx4 cres = nres, caux = naux;
            if (i + 1 < STEPS) fetch(i + 1);
            floatx4 v = *reinterpret_cast<const floatx4*>(ot + lrow * OLD + c4);
            const float add = econst[row] + econst[BM + row];
            v = floatx4{v.x + add, v.y + add, v.z + add, v.w + add};
            if (RES) v = floatx4{v.x + cres.x, v.y + cres.y, v.z + cres.z, v.w + cres.w};
            if (AUX) {
                const float aa = econst[2 * BM + row], ab = econst[3 * BM + row];
                v = floatx4{v.x + silu_fast(aa * caux.x + ab), v.y + silu_fast(aa * caux.y + ab), v.z + silu_fast(aa * caux.z + ab),
                            v.w + silu_fast(aa * caux.w + ab)};
            }
            *reinterpret_cast<floatx4*>(outb + (long long)(co0 + row) * HW + pix0 + c4) = v;
        }
    };
    auto halves = [&](auto res_tag, auto aux_tag) {
        rows(0, res_tag, aux_tag);
        rows(1, res_tag, aux_tag);
    };
    if (resb) {
        if (auxb) halves(std::true_type{}, std::true_type{});
        else halves(std::true_type{}, std::false_type{});
    } else {
        if (auxb) halves(std::false_type{}, std::true_type{});
        else halves(std::false_type{}, std::false_type{});
    }
}

// weight image: [chunk of 32 ci][block of 64 co][plane 3][octet kg 4][co 64][8 bf16], zero beyond Cin
__global__ void pack_x3_kernel(const float* __restrict__ w, uintx4* __restrict__ out, int Cout, int Cin, int ncob, long long npieces) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npieces; i += (long long)gridDim.x * blockDim.x) {
        const int co_l = i & 63;
        const int kg = (i >> 6) & 3;
        const int plane = (int)((i >> 8) % 3);
        const long long blk = i / 768;
        const int cob = (int)(blk % ncob), cc = (int)(blk / ncob);
        const int co = cob * 64 + co_l;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = cc * CKB + kg * 8 + e;
            x[e] = (co < Cout && ci < Cin) ? w[(long long)co * Cin + ci] : 0.f;
        }
        uintx4 p1, p2, p3;
        split3(x, p1, p2, p3);
        out[i] = plane == 0 ? p1 : plane == 1 ? p2 : p3;
    }
}

}  // namespace

namespace idiff_detail {

// normal mode: `a` holds the flattened 1x1 geometry (Hout == 1, Wout = pixels per sample); unshuffle mode: the real geometry.
// Either way ntiles = output pixels per sample / 256 and ncob = Cout / 64 (set by the caller for this kernel).
bool conv1x1_x3_eligible(const ConvArgs& a, int mode) {
    const long long hw = (long long)a.Hout * a.Wout;
    if (hw % 256 || hw > (1 << 25) || a.Cout % BM || a.Cin % 8 || a.Cin < CKB || a.pro_a || a.stats) return false;
    if (mode == IDIFF_CONV_UNSHUFFLE2)
        return !a.src1 && a.Wout % 4 == 0 && a.Win == 2 * a.Wout && (long long)a.Hin * a.Win <= (1 << 25) && a.bs0 % 4 == 0 &&
               (reinterpret_cast<uintptr_t>(a.src0) & 15) == 0;
    return mode == IDIFF_CONV_NORMAL && a.Hout == 1 && a.C0v % 8 == 0;
}

int launch_conv1x1_x3(const ConvArgs& a, int mode, const void* wx3, hipStream_t st) {
    const size_t lds = B_BYTES + A_BYTES + 4 * BM * sizeof(float);
    static idiff_dyn_lds_cache lds_cache[2];
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache[0], reinterpret_cast<const void*>(conv1x1_x3_kernel<false>), lds);
        if (e == hipSuccess) e = idiff_ensure_dyn_lds(lds_cache[1], reinterpret_cast<const void*>(conv1x1_x3_kernel<true>), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d(1x1 x3): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if (mode == IDIFF_CONV_UNSHUFFLE2) hipLaunchKernelGGL(conv1x1_x3_kernel<true>, dim3(a.total_wg), dim3(NT), lds, st, a, static_cast<const unsigned short*>(wx3));
    else hipLaunchKernelGGL(conv1x1_x3_kernel<false>, dim3(a.total_wg), dim3(NT), lds, st, a, static_cast<const unsigned short*>(wx3));
    IDIFF_CHECK_LAUNCH("conv2d_fwd(1x1 x3)");
    return IDIFF_OK;
}

}  // namespace idiff_detail

extern "C" long long idiff_conv1x1_x3_image_bytes(int Cout, int Cin) {
    return (long long)((Cin + CKB - 1) / CKB) * ((Cout + BM - 1) / BM) * A_BYTES;
}

extern "C" int idiff_pack_conv1x1_x3(const float* w, void* image, int Cout, int Cin, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(w && image && Cout > 0 && Cin > 0, "pack_conv1x1_x3: bad args");
    IDIFF_CHECK_ARG((reinterpret_cast<uintptr_t>(image) & 15) == 0, "pack_conv1x1_x3: image must be 16-byte aligned");
    const int ncob = (Cout + BM - 1) / BM;
    const long long npieces = idiff_conv1x1_x3_image_bytes(Cout, Cin) / 16;
    const int grid = (int)((npieces + 255) / 256 > 4096 ? 4096 : (npieces + 255) / 256);
    hipLaunchKernelGGL(pack_x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, static_cast<uintx4*>(image), Cout, Cin, ncob, npieces);
    IDIFF_CHECK_LAUNCH("pack_conv1x1_x3");
    return IDIFF_OK;
}
