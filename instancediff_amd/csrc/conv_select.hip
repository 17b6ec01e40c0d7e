// Output layer of the UNet fused with the class gather:  out[b,0,y,x] = bias[k_b] + sum_{ci,tap} w[k_b,ci,tap] * x[b,ci,y+dy,x+dx]
// with k_b = idx[b] -- the reference computes all K = 5 output channels of the final 3x3 conv and then keeps one per sample
// (out_nc = 5 gathered by modality).  One channel per sample is 2*9*C flops per pixel: a vector-ALU job bound by reading x
// once, so no matrix cores here (a 5 -> 32 padded MFMA tile would spend 6x the time on zeros).
// Workgroup = 8 x 32 output pixels of one sample, one pixel per thread; 8-channel chunks of the input patch (with halo)
// are staged through LDS, the 9*C weights of the sample's class once.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int SCK = 8, STH = 8, STW = 32, SRS = STW + 2, SPS = (STH + 2) * SRS;  // 340
constexpr int SNL = (SCK * SPS + 255) / 256;                                      // 11

__global__ __launch_bounds__(256) void conv3x3_select_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const int* __restrict__ idx, float* __restrict__ out,
                                                             int C, int H, int W, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;               // [SCK][10][34]
    float* wk = smem + SCK * SPS;     // [C][9]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int y0 = (blockIdx.x / tiles_x) * STH, x0 = (blockIdx.x % tiles_x) * STW;
    const int k = idx[b];
    for (int i = tid; i < C * 9; i += 256) wk[i] = w[(long long)k * C * 9 + i];
    const float* xb = x + (long long)b * xbs;
    const int HW = H * W;
    int goff[SNL];
#pragma unroll
    for (int i = 0; i < SNL; ++i) {
        const int e = tid + i * 256;
        const int ci = e / SPS, rem = e - ci * SPS;
        const int r = rem / SRS, c = rem - r * SRS;
        const int oy = y0 - 1 + r, ox = x0 - 1 + c;
        goff[i] = (e < SCK * SPS && oy >= 0 && oy < H && ox >= 0 && ox < W) ? ci * HW + oy * W + ox : -1;
    }
    const int ty = tid >> 5, tx = tid & 31;
    const float* tp = tile + ty * SRS + tx;
    float acc = 0.f;
    for (int cb = 0; cb < C; cb += SCK) {
        __syncthreads();  // previous chunk's reads are done (and wk is visible on the first pass)
#pragma unroll
        for (int i = 0; i < SNL; ++i) {
            const int e = tid + i * 256;
            if (e < SCK * SPS) {
                const int ci = e / SPS;
                tile[e] = (goff[i] >= 0 && cb + ci < C) ? xb[(long long)cb * HW + goff[i]] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < SCK; ++ci) {
            if (cb + ci < C) {
                const float* wp = wk + (cb + ci) * 9;
                const float* p = tp + ci * SPS;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc = __builtin_fmaf(p[dy * SRS + dx], wp[dy * 3 + dx], acc);
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < H && ox < W) out[(long long)b * HW + oy * W + ox] = acc + (bias ? bias[k] : 0.f);
}

// r05 streaming form (W % 4 == 0, 16-byte rows): thread = a strip of 4 consecutive output pixels; per input channel its 3 x 6 window comes
// straight from global memory (one 16-byte load + two halo dwords per row: the halo of a strip is its neighbours' data, an L1 hit) and
// feeds 36 FMAs with the class kernel broadcast from LDS -- no staged patch, no barrier per channel chunk.  Per pixel the multiply-adds
// run in the order of the tiled kernel above (channels ascending, taps row-major): the same bits.  The tiled form was bound by its LDS
// reads (two per multiply-add: 130 us at 256 x 256, batch 16, for 268 MB of input).
// This translation unit is compiled with -fno-slp-vectorize (Makefile).  hipcc's SLP pass turns the four accumulator chains into
// v_pk_fma_f32 pairs (72 of them + 149 moves that assemble register pairs around the exec-masked halo loads), and that form produced
// intermittently WRONG low halves of the pairs (pixels 0 and 2 of a strip, lanes 48..63 of a wave) -- only while the other net's
// kernels shared the CUs (two streams), never alone, never with AMD_SERIALIZE_KERNEL=3: the bitwise batch-invariance test of the 256^2
// chain caught it (profiles/r05/x_select_strips_packed_fma.txt).  The scalar form is bit-stable in every configuration tried.
__global__ __launch_bounds__(256) void conv3x3_select4_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const int* __restrict__ idx, float* __restrict__ out,
                                                              int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wk[];  // [C][9]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int k = idx[b];
    for (int i = tid; i < C * 9; i += 256) wk[i] = w[(long long)k * C * 9 + i];
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long p = ((long long)blockIdx.x * 256 + tid) * 4;
    if (p >= HW) return;
    const int y = (int)(p / W), x0 = (int)(p - (long long)y * W);
    const float* xb = x + (long long)b * xbs + p;
    const bool up = y > 0, dn = y + 1 < H, lf = x0 > 0, rt = x0 + 4 < W;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int UN = 4;  // channels of loads in flight
    for (int c0 = 0; c0 < C; c0 += UN) {
        float d[UN][3][6];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float* xc = xb + (long long)(c0 + u < C ? c0 + u : C - 1) * HW;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool rowok = r == 0 ? up : (r == 2 ? dn : true);
                const float* xr = xc + (long long)(r - 1) * W;
                const floatx4 m = rowok ? *reinterpret_cast<const floatx4*>(xr) : floatx4{0.f, 0.f, 0.f, 0.f};
                d[u][r][0] = (rowok && lf) ? xr[-1] : 0.f;
                d[u][r][1] = m.x, d[u][r][2] = m.y, d[u][r][3] = m.z, d[u][r][4] = m.w;
                d[u][r][5] = (rowok && rt) ? xr[4] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (c0 + u < C) {  // uniform
                const float* wc = wk + (c0 + u) * 9;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wv = wc[ky * 3 + kx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(d[u][ky][j + kx], wv, acc[j]);
                    }
            }
        }
    }
    const float bv = bias ? bias[k] : 0.f;
    *reinterpret_cast<floatx4*>(out + (long long)b * HW + p) = floatx4{acc.x + bv, acc.y + bv, acc.z + bv, acc.w + bv};
}

// ---- backward of the fused output layer (training step, r05).  With k_b = idx[b] and g = d pred [B, 1, H, W]:
//   dx[b,ci,y,x]   = sum_{ky,kx} w[k_b,ci,ky,kx] * g[b, y+1-ky, x+1-kx]
//   dw[k,ci,ky,kx] = sum_{b: k_b = k} sum_{y,x} g[b,y,x] * X[b,ci,y+ky-1,x+kx-1] ;   db[k] = sum_{b: k_b = k} sum g[b]
// The unfused form (5-channel conv -> gather; backward: scatter into a zero [B,5,H,W] tensor, a 5 -> 64 data-gradient conv and a
// 64 -> 5 weight-gradient conv on matrix-core tiles padded from 5 to 16 / 64 channels) spent 4/5 of its work on zeros.
// Thread = a strip of 4 consecutive pixels; its 3 x 6 window of g serves both products (the same window: an input pixel's matching
// gradients for tap (ky,kx) sit at (y+1-ky, x+1-kx)).  Both kernels are bound by streaming the 64-channel tensor once.
__device__ __forceinline__ void select_window(const float* __restrict__ gb, int y, int x, int H, int W, float (&d)[3][6]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y - 1 + r;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int xx = x - 1 + c;
            d[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? gb[(long long)yy * W + xx] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void conv3x3_select_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w, const int* __restrict__ idx,
                                                                   float* __restrict__ dx, long long dbs, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float wk[];  // [C][9]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int k = idx[b];
    for (int i = tid; i < C * 9; i += 256) wk[i] = w[(long long)k * C * 9 + i];
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long p = ((long long)blockIdx.x * 256 + tid) * 4;
    if (p >= HW) return;  // W % 4 == 0: a strip never straddles a row
    const int y = (int)(p / W), x = (int)(p - (long long)y * W);
    float d[3][6];
    select_window(g + (long long)b * HW, y, x, H, W, d);
    float* ob = dx + (long long)b * dbs + p;
    for (int ci = 0; ci < C; ++ci) {
        const float* wc = wk + ci * 9;
        floatx4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float wv = wc[ky * 3 + kx];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = __builtin_fmaf(wv, d[2 - ky][j + 2 - kx], a[j]);
            }
        *reinterpret_cast<floatx4*>(ob + (long long)ci * HW) = a;
    }
}

constexpr int SWG_STRIPS = 4;  // strips per thread: a workgroup covers 4096 pixels of one sample
__global__ __launch_bounds__(256) void conv3x3_select_wgrad_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ g,
                                                                   float* __restrict__ ws, int C, int H, int W, int ntile) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [C][4 waves][9] + [4] bias partials
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long HW = (long long)H * W;
    float d[SWG_STRIPS][3][6];
    long long pp[SWG_STRIPS];
    float gsum = 0.f;
#pragma unroll
    for (int s = 0; s < SWG_STRIPS; ++s) {
        const long long p = (((long long)blockIdx.x * SWG_STRIPS + s) * 256 + tid) * 4;
        pp[s] = p < HW ? p : -1;
        if (p < HW) {
            const int y = (int)(p / W), xx = (int)(p - (long long)y * W);
            select_window(g + (long long)b * HW, y, xx, H, W, d[s]);
            gsum += (d[s][1][1] + d[s][1][2]) + (d[s][1][3] + d[s][1][4]);
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) d[s][r][c] = 0.f;
        }
    }
    const float* xb = x + (long long)b * xbs;
    for (int ci = 0; ci < C; ++ci) {
        floatx4 xv[SWG_STRIPS];
#pragma unroll
        for (int s = 0; s < SWG_STRIPS; ++s)
            xv[s] = pp[s] >= 0 ? *reinterpret_cast<const floatx4*>(xb + (long long)ci * HW + pp[s]) : floatx4{0.f, 0.f, 0.f, 0.f};
        float acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = 0.f;
#pragma unroll
        for (int s = 0; s < SWG_STRIPS; ++s)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = __builtin_fmaf(xv[s][j], d[s][2 - ky][j + 2 - kx], acc[ky * 3 + kx]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum(acc[t]);
            if (lane == 0) red[(ci * 4 + wave) * 9 + t] = v;
        }
    }
    {
        const float v = wave_sum(gsum);
        if (lane == 0) red[C * 36 + wave] = v;
    }
    __syncthreads();
    float* wp = ws + ((long long)b * ntile + blockIdx.x) * (C * 9 + 1);
    for (int i = tid; i < C * 9; i += 256) {
        const int ci = i / 9, t = i - ci * 9;
        wp[i] = (red[(ci * 4 + 0) * 9 + t] + red[(ci * 4 + 1) * 9 + t]) + (red[(ci * 4 + 2) * 9 + t] + red[(ci * 4 + 3) * 9 + t]);
    }
    if (tid == 0) wp[C * 9] = (red[C * 36] + red[C * 36 + 1]) + (red[C * 36 + 2] + red[C * 36 + 3]);
}

// dw[k][ci*9 + t] = sum over the samples of class k (ascending b) and their tiles (ascending): a fixed order; classes without a sample get 0
__global__ __launch_bounds__(256) void conv3x3_select_wgrad_reduce_kernel(const float* __restrict__ ws, const int* __restrict__ idx, float* __restrict__ dw,
                                                                          float* __restrict__ db, int B, int ntile, int C, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int PW = C * 9 + 1;
    if (i >= PW) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        if (idx[b] != k) continue;  // uniform
        const float* wb = ws + (long long)b * ntile * PW + i;
        for (int t = 0; t < ntile; ++t) acc += wb[(long long)t * PW];
    }
    if (i < C * 9) dw[(long long)k * C * 9 + i] = acc;
    else if (db) db[k] = acc;
}

}  // namespace

static int select_bwd_tiles(int H, int W) { return (int)(((long long)H * W + 4 * 256 * SWG_STRIPS - 1) / (4 * 256 * SWG_STRIPS)); }
extern "C" int64_t idiff_conv3x3_select_bwd_ws_floats(int B, int C, int H, int W) { return (int64_t)B * select_bwd_tiles(H, W) * (C * 9 + 1); }
extern "C" int idiff_conv3x3_select_bwd(const float* x, int64_t x_bstride, const float* w, const int32_t* idx, const float* dpred, float* dx,
                                        int64_t dx_bstride, float* dw, float* db, float* ws, int B, int C, int K, int H, int W, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && w && idx && dpred && (dx || dw), "conv3x3_select_bwd: null pointer");
    IDIFF_CHECK_ARG(B > 0 && C > 0 && C <= 256 && K > 0 && H > 0 && W > 0 && W % 4 == 0, "conv3x3_select_bwd: bad dims (W %% 4 == 0, C <= 256)");
    IDIFF_CHECK_ARG(x_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "conv3x3_select_bwd: x must have 16-byte rows");
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)H * W;
    if (dx) {
        IDIFF_CHECK_ARG(dx_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0, "conv3x3_select_bwd: dx must have 16-byte rows");
        hipLaunchKernelGGL(conv3x3_select_dgrad_kernel, dim3((unsigned)((HW / 4 + 255) / 256), B), dim3(256), (size_t)C * 9 * sizeof(float), st, dpred, w, idx, dx,
                           (long long)dx_bstride, C, H, W);
        IDIFF_CHECK_LAUNCH("conv3x3_select_dgrad");
    }
    if (dw) {
        IDIFF_CHECK_ARG(ws, "conv3x3_select_bwd: the weight gradient needs a workspace of idiff_conv3x3_select_bwd_ws_floats() floats");
        const int ntile = select_bwd_tiles(H, W);
        hipLaunchKernelGGL(conv3x3_select_wgrad_kernel, dim3(ntile, B), dim3(256), (size_t)(C * 36 + 4) * sizeof(float), st, x, (long long)x_bstride, dpred, ws, C,
                           H, W, ntile);
        IDIFF_CHECK_LAUNCH("conv3x3_select_wgrad");
        hipLaunchKernelGGL(conv3x3_select_wgrad_reduce_kernel, dim3((C * 9 + 1 + 255) / 256, K), dim3(256), 0, st, ws, idx, dw, db, B, ntile, C, K);
        IDIFF_CHECK_LAUNCH("conv3x3_select_wgrad_reduce");
    }
    return IDIFF_OK;
}

extern "C" int idiff_conv3x3_select_fwd(const float* x, int64_t x_bstride, const float* w, const float* bias, const int32_t* idx, float* out, int B,
                                        int C, int K, int H, int W, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && w && idx && out, "conv3x3_select: null pointer");
    IDIFF_CHECK_ARG(B > 0 && C > 0 && K > 0 && H > 0 && W > 0 && C <= 1024, "conv3x3_select: bad dims");
    IDIFF_CHECK_ARG(x_bstride >= (long long)C * H * W, "conv3x3_select: x_bstride too small");
    // The strip form is opt-in (IDIFF_SELECT_STRIPS=1): built without SLP packing it is bit-stable but no faster than the tiled form
    // (126 vs 130 us at 256 x 256, batch 16: both end up bound by their ~2 VALU / LDS issue slots per multiply-add), and its packed
    // build was the one kernel of the library that was ever wrong only in company (note above).
    static const bool strip_on = [] {
        const char* e = getenv("IDIFF_SELECT_STRIPS");
        return e && e[0] == '1';
    }();
    if (strip_on && W % 4 == 0 && x_bstride % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const long long HW = (long long)H * W;
        hipLaunchKernelGGL(conv3x3_select4_kernel, dim3((unsigned)((HW / 4 + 255) / 256), B), dim3(256), (size_t)C * 9 * sizeof(float), (hipStream_t)stream, x,
                           (long long)x_bstride, w, bias, idx, out, C, H, W);
        IDIFF_CHECK_LAUNCH("conv3x3_select_fwd(strips)");
        return IDIFF_OK;
    }
    const int tiles_x = (W + STW - 1) / STW, tiles_y = (H + STH - 1) / STH;
    const size_t lds = ((size_t)SCK * SPS + (size_t)C * 9) * sizeof(float);
    hipLaunchKernelGGL(conv3x3_select_kernel, dim3(tiles_x * tiles_y, B), dim3(256), lds, (hipStream_t)stream, x, (long long)x_bstride, w, bias, idx,
                       out, C, H, W, tiles_x);
    IDIFF_CHECK_LAUNCH("conv3x3_select_fwd");
    return IDIFF_OK;
}
