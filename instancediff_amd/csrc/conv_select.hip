// Output layer of the UNet fused with the class gather:  out[b,0,y,x] = bias[k_b] + sum_{ci,tap} w[k_b,ci,tap] * x[b,ci,y+dy,x+dx]
// with k_b = idx[b] -- the reference computes all K = 5 output channels of the final 3x3 conv and then keeps one per sample
// (out_nc = 5 gathered by modality).  One channel per sample is 2*9*C flops per pixel: a vector-ALU job bound by reading x
// once, so no matrix cores here (a 5 -> 32 padded MFMA tile would spend 6x the time on zeros).
// Workgroup = 8 x 32 output pixels of one sample, one pixel per thread; 8-channel chunks of the input patch (with halo)
// are staged through LDS, the 9*C weights of the sample's class once.
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

}  // namespace

extern "C" int idiff_conv3x3_select_fwd(const float* x, int64_t x_bstride, const float* w, const float* bias, const int32_t* idx, float* out, int B,
                                        int C, int K, int H, int W, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && w && idx && out, "conv3x3_select: null pointer");
    IDIFF_CHECK_ARG(B > 0 && C > 0 && K > 0 && H > 0 && W > 0 && C <= 1024, "conv3x3_select: bad dims");
    IDIFF_CHECK_ARG(x_bstride >= (long long)C * H * W, "conv3x3_select: x_bstride too small");
    const int tiles_x = (W + STW - 1) / STW, tiles_y = (H + STH - 1) / STH;
    const size_t lds = ((size_t)SCK * SPS + (size_t)C * 9) * sizeof(float);
    hipLaunchKernelGGL(conv3x3_select_kernel, dim3(tiles_x * tiles_y, B), dim3(256), lds, (hipStream_t)stream, x, (long long)x_bstride, w, bias, idx,
                       out, C, H, W, tiles_x);
    IDIFF_CHECK_LAUNCH("conv3x3_select_fwd");
    return IDIFF_OK;
}
