// Weight gradient of the 3x3 convolution in Winograd F(4x4,3x3) form on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   forward:  Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 4x4 output tile, summed over input channels (conv_wino4.hip)
//   hence     dg[co][ci] = G^T [ sum_tiles (A dY A^T)[co] .* (B^T d B)[ci] ] G
//   i.e. 36 independent [Cout x T] x [T x Cin] products over T = all 4x4 tiles of all samples: 36 multiply-adds per (co, ci) and
//   16 outputs = 2.25 per output against the 4 of the F(2x2,3x3) form (conv_wino_wgrad.hip) and the 9 of the direct form.
//
//   The matrix part IS the forward kernel's: with the tile index as the contraction index, E = A dY A^T [36][64 co][4 tiles] takes the
//   place of the weight image U and D = B^T d B [36][32 ci][4 tiles] the place of V -- the same LDS operand images
//   ([9 position quads][blocks][4 k][16][4], one ds_read_b128 of each per four MFMAs), the same 144 accumulators per wave
//   (wave (cb, ib) owns 16 co x 16 ci x all 36 positions), the final G^T M G in-lane.
//
//   Workgroup = 512 threads owning a 64 co x 32 ci block of dW and one contiguous share of the tile list (split-K; the partials are
//   reduced by wgrad_reduce_kernel in a fixed order -> bitwise reproducible).  K chunk = 4 tiles in a row (4 x 16 output pixels):
//     R [32 ci][6 x 20 (18 used)]   activated, zero-padded input patch with halo                     (LDS, single buffer)
//     D [9 quads][2 ci-blocks][4 tiles][16 ci][4]   B^T d B: thread = (tile, ci) x row set, as the forward transform   (double buffer)
//     E [9 quads][4 co-blocks][4 tiles][16 co][4]   A dY A^T: thread = (tile, co) x row half, from dY registers       (double buffer)
//   Two barriers per chunk: [quads 0-4 | stage R of chunk c+1]  barrier  [quads 5-8 | transforms of chunk c+1, loads of chunk c+2]
//   barrier.  The input gather has the forward kernel's semantics (virtual concat, GroupNorm/FiLM affine + SiLU prologue, zero
//   padding after the activation), so the fused forward needs no materialised activated tensor for its backward.
#include <stdlib.h>

#include <type_traits>

#include "conv_wgrad_args.h"

using idiff_detail::WwArgs;

namespace {

constexpr int NT = 512;
constexpr int PR = 6, PC = 20, PS = PR * PC;  // patch rows, row stride (18 columns used: 16-byte aligned rows), floats per channel
constexpr int CIB = 32;                       // input channels per workgroup
constexpr int R_FLOATS = CIB * PS;            // 3840
constexpr int NL = 8;                         // gathered elements per thread per chunk (8 * 512 = 4096 >= 3840)
constexpr int D_FLOATS = 9 * 2 * 64 * 4;      // 4608
constexpr int E_FLOATS = 9 * 4 * 64 * 4;      // 9216

typedef float floatx2 __attribute__((ext_vector_type(2)));

// position numbering of conv_wino4.hip: row u = one 16-byte piece in quad PF(u) + one 8-byte piece in half of quad PH(u)
__host__ __device__ constexpr int PF(int u) { return (3 * u + 1) / 2; }
__host__ __device__ constexpr int PH(int u) { return 1 + 3 * (u / 2); }
__host__ __device__ constexpr int pos(int u, int v) { return v < 4 ? 4 * PF(u) + v : 4 * PH(u) + 2 * (u & 1) + (v - 4); }

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
// A x for a 4-point x: the six rows of A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
__device__ __forceinline__ void a6(float x0, float x1, float x2, float x3, float (&o)[6]) {
    const float s = x0 + x2, t = x1 + x3;
    o[0] = x0;
    o[1] = s + t;
    o[2] = s - t;
    const float p = __builtin_fmaf(4.f, x2, x0), q = __builtin_fmaf(8.f, x3, 2.f * x1);
    o[3] = p + q;
    o[4] = p - q;
    o[5] = x3;
}
// sum_u G[u][p] m[u] for the six rows of G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
__device__ __forceinline__ void gt3(float m0, float m1, float m2, float m3, float m4, float m5, float (&o)[3]) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o[0] = __builtin_fmaf(0.25f, m0, __builtin_fmaf(-1.f / 6.f, s12, (1.f / 24.f) * s34));
    o[1] = __builtin_fmaf(-1.f / 6.f, d12, (1.f / 12.f) * d34);
    o[2] = __builtin_fmaf(-1.f / 6.f, s12, __builtin_fmaf(1.f / 6.f, s34, m5));
}

template <bool PRO>
__global__ __launch_bounds__(NT) void wino4_wgrad_kernel(const WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const R = smem;                        // [CIB][PS]
    float* const Db = smem + R_FLOATS;            // [2][D_FLOATS]
    float* const Eb = Db + 2 * D_FLOATS;          // [2][E_FLOATS]
    int* const gtab = reinterpret_cast<int*>(Eb + 2 * E_FLOATS);  // [NL][NT] constant gather offsets (thread-private entries)
    float* const protab = reinterpret_cast<float*>(gtab + NL * NT);  // [2][C0r] (PRO)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3;   // MFMA role: 16-co block
    const int ib = wave >> 2;  // MFMA role: 16-ci block

    const int cob = blockIdx.x % a.ncob;
    const int cibw = (blockIdx.x / a.ncob) % a.ncib;
    const int sp = blockIdx.x / (a.ncob * a.ncib);
    const int co0 = cob * 64, ci0 = cibw * CIB;
    const int HWin = a.Hin * a.Win, HWo = a.Hout * a.Wout;
    const int nxc = a.Wout / 16, nty = a.Hout / 4;
    const int total = a.B * nty * nxc;
    const int per = (total + a.nsplit - 1) / a.nsplit;
    const int c_begin = sp * per;
    const int c_end = c_begin + per < total ? c_begin + per : total;

    // this block's 32 input channels live in ONE source (C0 % 32 == 0 is an eligibility condition for two sources)
    const bool from1 = a.src1 != nullptr && ci0 >= a.C0v;
    const float* const srcb = from1 ? a.src1 : a.src0;
    const long long sbs = from1 ? a.bs1 : a.bs0;
    const int chan0 = from1 ? ci0 - a.C0v : ci0;

    // ---- per-thread gather descriptors: element e = tid + i*512 of R = (ci, r, c); byte offset relative to the chunk's origin
    // (patch corner of channel chan0), -1 for slots beyond the patch / beyond Cin; edge flags: bit0 r == 0, bit1 r == 5, bit2 c == 0,
    // bit3 c == 17 (elements outside the image on border chunks)
    unsigned eflags = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + i * NT;
        const int ci = e / PS;
        const int rem = e - ci * PS;
        const int r = rem / PC, c = rem - r * PC;
        const bool used = e < R_FLOATS && c < 18 && ci0 + ci < a.Cin;
        // nearest x2 upsample: output row 4 ty - 1 + r reads input row 2 ty - 1 + ((r + 1) >> 1), likewise the columns -- the same for
        // every chunk, so only this table and the chunk origin know about the mode
        gtab[i * NT + tid] = !used ? -1 : (a.ups ? (ci * HWin + ((r + 1) >> 1) * a.Win + ((c + 1) >> 1)) * 4 : (ci * HWin + r * a.Win + c) * 4);
        eflags |= ((r == 0 ? 1u : 0u) | (r == PR - 1 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == 17 ? 8u : 0u)) << (4 * i);
    }

    constexpr int RSRC_FLAGS = 0x00020000;
    __amdgpu_buffer_rsrc_t rsx, rsy;
    unsigned cur_edges = 0;
    auto setup_chunk = [&](int idx) {  // scalar work only
        const int b = idx / (nty * nxc);
        const int rem = idx - b * (nty * nxc);
        const int ty = rem / nxc, xc = rem - ty * nxc;
        const long long org = a.ups ? (long long)(2 * ty - 1) * a.Win + (8 * xc - 1) : (long long)(4 * ty - 1) * a.Win + (16 * xc - 1);
        rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(srcb + (long long)b * sbs + (long long)chan0 * HWin + org), 0, 0x7fffffff, RSRC_FLAGS);
        rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long long)b * a.dybs + (long long)co0 * HWo + (long long)(4 * ty) * a.Wout + 16 * xc), 0,
                                                0x7fffffff, RSRC_FLAGS);
        cur_edges = (ty == 0 ? 1u : 0u) | (4 * ty + 4 == a.Hout ? 2u : 0u) | (xc == 0 ? 4u : 0u) | (16 * xc + 16 == a.Wout ? 8u : 0u);
        return b;
    };

    float rin[NL];
    unsigned rin_pad = 0;  // bit i: element i of the registers is padding (outside the image / unused slot)
    auto load_raw = [&]() {
        int off[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) off[i] = gtab[i * NT + tid];
        if (cur_edges != 0) {  // border chunks only (uniform): elements outside the image
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if (((eflags >> (4 * i)) & cur_edges) != 0) off[i] = -1;
        }
        if (PRO) {
            rin_pad = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) rin_pad |= (off[i] < 0 ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) rin[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, off[i], 0, 0));
    };
    // dY of (channel co0 + cb*16 + (lane & 15), tile lane >> 4): its four rows of four pixels
    floatx4 dyr[4];
    const int dyvoff = ((cb * 16 + (lane & 15)) * HWo + 4 * (lane >> 4)) * 4;
    auto load_dy = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) dyr[r] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsy, dyvoff, r * a.Wout * 4, 0));
    };
    auto stage_raw = [&](int i) {
        float x = rin[i];
        if (PRO) {
            const int chn = chan0 + (tid + i * NT) / PS;
            const int chc = chn < a.C0r ? chn : 0;
            x = silu_fast(protab[chc] * x + protab[a.C0r + chc]);
            if ((rin_pad >> i) & 1u) x = 0.f;  // padding is zero AFTER the activation
        }
        if (tid + i * NT < R_FLOATS) R[tid + i * NT] = x;
    };

    // ---- D = B^T d B of (tile lane >> 4, channel dib*16 + (lane & 15)) for the wave's row set (as the forward kernel's transform):
    //   waves 0-3 (heavy): Winograd rows (1,2) (trole 0) or (3,4) (trole 1);  waves 4-7 (light): row 0 or row 5;  dib = (wave >> 1) & 1
    const bool heavy = wave < 4;
    const int trole = wave & 1;
    const int dib = (wave >> 1) & 1;
    const float al = trole ? -1.f : -4.f, be = trole ? 2.f : 1.f;
    const float* const trbase = R + (dib * 16 + (lane & 15)) * PS + (heavy ? 1 : trole) * PC + 4 * (lane >> 4);
    const int ufirst = heavy ? 1 + 2 * trole : 5 * trole;
    float* const dwbase = Db + dib * 256 + lane * 4;  // [quad][ib][k = tile][ci16][4]: (tile, ci16) = (lane >> 4, lane & 15)
    auto rd_row = [&](const float* p, float (&d)[6]) {
        const floatx4 lo = *reinterpret_cast<const floatx4*>(p);
        const floatx2 hi = *reinterpret_cast<const floatx2*>(p + 4);
        d[0] = lo.x, d[1] = lo.y, d[2] = lo.z, d[3] = lo.w, d[4] = hi.x, d[5] = hi.y;
    };
    auto d_transform = [&](int buf) {
        float ta[6], tb[6], tlo[6], thi[6];
        const float* p = trbase;
        if (heavy) {
            rd_row(p + 1 * PC, ta), rd_row(p + 3 * PC, tb);  // d2, d4
#pragma unroll
            for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(al, ta[c], tb[c]);  // X = d4 + al*d2
            rd_row(p, ta), rd_row(p + 2 * PC, tb);                                    // d1, d3
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float X = tlo[c], Y = __builtin_fmaf(al, ta[c], tb[c]);
                tlo[c] = __builtin_fmaf(be, Y, X);
                thi[c] = __builtin_fmaf(-be, Y, X);
            }
        } else {
            rd_row(p, ta), rd_row(p + 2 * PC, tb);  // dA, dB
#pragma unroll
            for (int c = 0; c < 6; ++c) tlo[c] = __builtin_fmaf(4.f, ta[c], -5.f * tb[c]);
            rd_row(p + 4 * PC, ta);  // dC
#pragma unroll
            for (int c = 0; c < 6; ++c) tlo[c] += ta[c];
        }
        float* const D = dwbase + buf * D_FLOATS;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            if (which == 1 && !heavy) break;
            float o[6];
            bt6(which ? thi : tlo, o);
            const int u = ufirst + which;
            *reinterpret_cast<floatx4*>(D + PF(u) * 512) = floatx4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<floatx2*>(D + PH(u) * 512 + 2 * (u & 1)) = floatx2{o[4], o[5]};
        }
    };
    // ---- E = A dY A^T of (tile lane >> 4, channel cb*16 + (lane & 15)), rows 3*ib .. 3*ib+2 of the 6x6 result, from the dY registers
    float* const ewbase = Eb + cb * 256 + lane * 4;  // [quad][cb][k = tile][co16][4]
    // the wave's three rows of A x: rows 0-2 (x0, s + t, s - t) or rows 3-5 (p + q, p - q, x3)
    auto a3 = [&](float x0, float x1, float x2, float x3, float (&o)[3]) {
        if (ib == 0) {  // (uniform)
            const float sm = x0 + x2, t = x1 + x3;
            o[0] = x0, o[1] = sm + t, o[2] = sm - t;
        } else {
            const float p = __builtin_fmaf(4.f, x2, x0), q = __builtin_fmaf(8.f, x3, 2.f * x1);
            o[0] = p + q, o[1] = p - q, o[2] = x3;
        }
    };
    auto e_transform = [&](int buf) {
        float w[4][3];  // [column x][row uu] = (A dY)[3*ib + uu][x]
        a3(dyr[0].x, dyr[1].x, dyr[2].x, dyr[3].x, w[0]);
        a3(dyr[0].y, dyr[1].y, dyr[2].y, dyr[3].y, w[1]);
        a3(dyr[0].z, dyr[1].z, dyr[2].z, dyr[3].z, w[2]);
        a3(dyr[0].w, dyr[1].w, dyr[2].w, dyr[3].w, w[3]);
        float* const E = ewbase + buf * E_FLOATS;
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) {
            float z[6];
            a6(w[0][uu], w[1][uu], w[2][uu], w[3][uu], z);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h != ib) continue;  // (uniform: the wave's half; the LDS offsets are immediates per half)
                const int u = 3 * h + uu;
                *reinterpret_cast<floatx4*>(E + PF(u) * 1024) = floatx4{z[0], z[1], z[2], z[3]};
                *reinterpret_cast<floatx2*>(E + PH(u) * 1024 + 2 * (u & 1)) = floatx2{z[4], z[5]};
            }
        }
    };
    auto load_protab = [&](int b) {
        for (int i = tid; i < a.C0r; i += NT) {
            protab[i] = a.pro_a[(long long)b * a.C0r + i];
            protab[a.C0r + i] = a.pro_b[(long long)b * a.C0r + i];
        }
    };

    floatx4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) {
        // ---- pipeline fill: operands of the first chunk in buffer 0, raw data of the second in registers --------------------------
        int cur_b = -1;
        int b = setup_chunk(c_begin);
        if (PRO) {
            load_protab(b);
            cur_b = b;
        }
        load_raw();
        load_dy();
        __syncthreads();  // (protab; gtab is thread-private)
#pragma unroll
        for (int i = 0; i < NL; ++i) stage_raw(i);
        __syncthreads();
        d_transform(0);
        e_transform(0);
        int nb = b;  // sample of the chunk whose raw data sits in the registers
        if (c_begin + 1 < c_end) {
            nb = setup_chunk(c_begin + 1);
            load_raw();
            load_dy();
        }
        __syncthreads();

        const int opoff = lane * 4;
        auto chunk = [&](int c, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;  // false: last chunk of this workgroup, nothing left to stage
            const int buf = (c - c_begin) & 1;
            if (PRO && MORE && nb != cur_b) {  // the next chunk starts a new sample: its prologue table (rare: contiguous ranges)
                __syncthreads();
                load_protab(nb);
                cur_b = nb;
                __syncthreads();
            }
            const float* D = Db + buf * D_FLOATS + ib * 256 + opoff;
            const float* E = Eb + buf * E_FLOATS + cb * 256 + opoff;
            floatx4 ob[2], oa[2];
            ob[0] = *reinterpret_cast<const floatx4*>(D);
            oa[0] = *reinterpret_cast<const floatx4*>(E);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                if (q + 1 < 9) {
                    ob[(q + 1) & 1] = *reinterpret_cast<const floatx4*>(D + (q + 1) * 512);
                    oa[(q + 1) & 1] = *reinterpret_cast<const floatx4*>(E + (q + 1) * 1024);
                }
                const floatx4 bv = ob[q & 1], av = oa[q & 1];
                acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[4 * q + 0], 0, 0, 0);
                acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[4 * q + 1], 0, 0, 0);
                acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[4 * q + 2], 0, 0, 0);
                acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[4 * q + 3], 0, 0, 0);
                if (MORE) {  // the slices of the next chunk's staging, dealt out over the quads
                    if (q < 4) stage_raw(2 * q), stage_raw(2 * q + 1);
                    // the patch of chunk c+2 is requested as soon as the registers are free (staged): most of a chunk of latency
                    // budget (requested behind quad 7 it had two quads and the kernel ran at the memory latency: 14 k cycles per
                    // 36-MFMA chunk); dY of chunk c+2 behind the transform that consumes the registers
                    if (q == 4 && c + 2 < c_end) {
                        nb = setup_chunk(c + 2);
                        load_raw();
                    }
                    if (q == 5) d_transform(buf ^ 1);
                    if (q == 6) e_transform(buf ^ 1);
                    if (q == 7 && c + 2 < c_end) load_dy();
                    __builtin_amdgcn_sched_barrier(0);
                    if (q == 4 || q == 8) __syncthreads();
                }
            }
        };
        for (int c = c_begin; c + 1 < c_end; ++c) chunk(c, std::true_type{});
        chunk(c_end - 1, std::false_type{});
    }

    // ---- epilogue: dg = G^T M G in-lane, partial dW of this split to ws[sp][tap][ci][co] (co fastest) -------------------------------
    // C layout of 16x16x4: lane holds column j (ci) and rows 4*k4 + r (co) of the wave's 16x16 block
    float* const wsp = a.ws + (long long)sp * 9 * a.Cin * a.Cout;
    const int j = lane & 15, k4 = lane >> 4;
    const int ci = ci0 + ib * 16 + j;
    if (ci < a.Cin) {
        const int co = co0 + cb * 16 + 4 * k4;
        floatx4 dg[9];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t[6][3];  // t[v][p] = sum_u G[u][p] M[u][v]
#pragma unroll
            for (int v = 0; v < 6; ++v)
                gt3(acc[pos(0, v)][r], acc[pos(1, v)][r], acc[pos(2, v)][r], acc[pos(3, v)][r], acc[pos(4, v)][r], acc[pos(5, v)][r], t[v]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float o[3];
                gt3(t[0][p], t[1][p], t[2][p], t[3][p], t[4][p], t[5][p], o);
                dg[p * 3 + 0][r] = o[0], dg[p * 3 + 1][r] = o[1], dg[p * 3 + 2][r] = o[2];
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<floatx4*>(wsp + ((long long)tap * a.Cin + ci) * a.Cout + co) = dg[tap];
    }
}

template <bool PRO>
int launch(const WwArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)R_FLOATS + 2 * D_FLOATS + 2 * E_FLOATS + NL * NT + (PRO ? 2 * (size_t)a.C0r : 0)) * sizeof(float);
    if (lds > 160 * 1024) IDIFF_FAIL(IDIFF_E_UNSUPPORTED, "conv2d_wgrad(winograd4): LDS budget exceeded (%zu bytes)", lds);
    static idiff_dyn_lds_cache lds_cache;
    auto kern = wino4_wgrad_kernel<PRO>;
    {
        hipError_t e = idiff_ensure_dyn_lds(lds_cache, reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "conv2d_wgrad(winograd4): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(a.ncob * a.ncib * a.nsplit), dim3(NT), lds, st, a);
    IDIFF_CHECK_LAUNCH("conv2d_wgrad(winograd4)");
    return IDIFF_OK;
}

bool wino4_wgrad_disabled() {  // IDIFF_WGRAD4=0: weight gradients stay on the F(2x2,3x3) kernel (A/B runs)
    static const bool off = [] {
        const char* e = getenv("IDIFF_WGRAD4");
        const char* w = getenv("IDIFF_WINOGRAD");
        return (e && e[0] == '0') || (w && w[0] == '0');
    }();
    return off;
}

}  // namespace

namespace idiff_detail {

bool wino4_wgrad_eligible(const WwArgs& a, int ks, int mode) {
    if (ks != 3 || wino4_wgrad_disabled()) return false;
    if (mode != IDIFF_CONV_NORMAL && mode != IDIFF_CONV_UPSAMPLE2) return false;
    if (mode == IDIFF_CONV_UPSAMPLE2 && (a.src1 || a.pro_a)) return false;
    if (a.Cout % 64 || a.Cin % 16 || a.Hout % 4 || a.Wout % 16) return false;
    if (a.src1 && a.C0v % CIB) return false;
    if (a.pro_a && a.src1) return false;
    if ((reinterpret_cast<uintptr_t>(a.dy) & 15) || (a.dybs & 3) || (a.Wout & 3)) return false;  // float4 dY loads
    if ((long long)a.Cin * a.Hin * a.Win * 4 >= (1ll << 31) || (long long)a.Cout * a.Hout * a.Wout * 4 >= (1ll << 31)) return false;  // 32-bit lane offsets
    if (a.pro_a && ((size_t)R_FLOATS + 2 * D_FLOATS + 2 * E_FLOATS + NL * NT + 2 * (size_t)a.C0r) * sizeof(float) > 160 * 1024) return false;
    return true;
}

void wino4_wgrad_geometry(int Cin, int Cout, int B, int Hout, int Wout, int* ncob, int* ncib, int* nsplit) {
    *ncob = Cout / 64;
    *ncib = (Cin + CIB - 1) / CIB;
    const int total = B * (Hout / 4) * (Wout / 16);
    int s = 512 / (*ncob * *ncib);  // two rounds of workgroups on 256 CUs: the tail of one round overlaps the next
    if (s < 1) s = 1;
    if (s > total) s = total;
    *nsplit = s;
}

int launch_wino4_wgrad(const WwArgs& a, hipStream_t st) {
    if (a.pro_a) return launch<true>(a, st);
    return launch<false>(a, st);
}

}  // namespace idiff_detail
