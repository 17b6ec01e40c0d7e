// On-device validation metrics of the drivers (trainUM.py:314-329, testUM.py:151-164): RMSE / PSNR (data_range 1)
// and SSIM with skimage's settings (gaussian_weights=True, sigma=1.5 -> 11x11 window, use_sample_covariance=False,
// K1=.01, K2=.03, data_range=1, mean over the interior cropped by (win-1)/2) on images mapped x/2+0.5.
// Replaces a per-image device->host copy + skimage call by one launch per batch and one [B,3] read-back.
#include <math.h>

#include "common.h"

namespace {

constexpr int SS_R = 5;  // radius: int(3.5 * 1.5 + 0.5)
constexpr int SS_PARTS = 64;

__global__ __launch_bounds__(256) void metrics_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, float* __restrict__ part,
                                                              int H, int W) {
    __shared__ float w1[2 * SS_R + 1];
    __shared__ float red[2][4];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = -SS_R; i <= SS_R; ++i) {
            w1[i + SS_R] = expf(-0.5f * (float)(i * i) / (1.5f * 1.5f));
            s += w1[i + SS_R];
        }
        for (int i = 0; i <= 2 * SS_R; ++i) w1[i] /= s;
    }
    __syncthreads();
    const float* p = pred + (long long)b * H * W;
    const float* t = tgt + (long long)b * H * W;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float sse = 0.f, ssim = 0.f;
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const float d = 0.5f * (p[i] - t[i]);  // (p/2+0.5) - (t/2+0.5)
        sse += d * d;
        if (y >= SS_R && y < H - SS_R && x >= SS_R && x < W - SS_R) {
            float ux = 0.f, uy = 0.f, uxx = 0.f, uyy = 0.f, uxy = 0.f;
            for (int dy = -SS_R; dy <= SS_R; ++dy) {
                const float wy = w1[dy + SS_R];
                for (int dx = -SS_R; dx <= SS_R; ++dx) {
                    const float w = wy * w1[dx + SS_R];
                    const float a = 0.5f * p[(y + dy) * W + x + dx] + 0.5f;
                    const float c = 0.5f * t[(y + dy) * W + x + dx] + 0.5f;
                    ux += w * a;
                    uy += w * c;
                    uxx += w * a * a;
                    uyy += w * c * c;
                    uxy += w * a * c;
                }
            }
            const float vx = uxx - ux * ux, vy = uyy - uy * uy, vxy = uxy - ux * uy;
            ssim += ((2.f * ux * uy + C1) * (2.f * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
    }
    sse = wave_sum(sse);
    ssim = wave_sum(ssim);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sse;
        red[1][threadIdx.x >> 6] = ssim;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long long)b * SS_PARTS + blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[((long long)b * SS_PARTS + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void metrics_final_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int H, int W) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double sse = 0.0, ss = 0.0;
    for (int i = 0; i < SS_PARTS; ++i) {
        sse += (double)part[((long long)b * SS_PARTS + i) * 2];
        ss += (double)part[((long long)b * SS_PARTS + i) * 2 + 1];
    }
    const double mse = sse / ((double)H * W);
    const double inner = (double)(H - 2 * SS_R) * (double)(W - 2 * SS_R);
    out[b * 3 + 0] = (float)sqrt(mse);
    out[b * 3 + 1] = (float)(10.0 * log10(1.0 / mse));
    out[b * 3 + 2] = inner > 0 ? (float)(ss / inner) : 0.f;
}

}  // namespace

extern "C" int idiff_image_metrics(const float* pred, const float* target, float* out_b3, float* ws, int B, int H, int W, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(pred && target && out_b3 && ws && B > 0 && H > 0 && W > 0, "image_metrics: bad args");
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(SS_PARTS, B), dim3(256), 0, (hipStream_t)stream, pred, target, ws, H, W);
    IDIFF_CHECK_LAUNCH("image_metrics_partial");
    hipLaunchKernelGGL(metrics_final_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, out_b3, B, H, W);
    IDIFF_CHECK_LAUNCH("image_metrics_final");
    return IDIFF_OK;
}
