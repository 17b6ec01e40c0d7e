// Generic batched GEMM on the f32 matrix cores + row softmax forward/backward.
// Building blocks of the attention / token-side BACKWARD passes (training path), where shapes are small
// (<= a few GFLOP) and generality matters more than the last 20 % of MFMA utilisation.
//   C[b] = alpha * op(A[b]) (M x K) . op(B[b]) (K x N) + beta * C[b],   row-major, leading dims lda/ldb/ldc,
//   op(X) = X or X^T.  64x64 tile per workgroup, 4 waves x one 32x32 accumulator, K chunks of 16 through LDS
//   (both operands are stored k-major in LDS so the MFMA operand reads are unit-stride along m / n).
#include <math.h>

#include "common.h"

namespace {

constexpr int GT = 64;   // tile M = N
constexpr int GK = 16;   // K chunk
constexpr int GLD = GT + 1;

__global__ __launch_bounds__(256) void bgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                                    int K, long long lda, long long ldb, long long ldc, int transA, int transB, long long sA,
                                                    long long sB, long long sC, float alpha, float beta) {
    __shared__ float As[GK][GLD];
    __shared__ float Bs[GK][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const float* Ab = A + (long long)blockIdx.z * sA;
    const float* Bb = B + (long long)blockIdx.z * sB;
    float* Cb = C + (long long)blockIdx.z * sC;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += GK) {
        // stage A tile (64 m x 16 k) and B tile (16 k x 64 n): 4 elements per thread each, unit stride in memory
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (transA) {  // memory [k][m]
                k = tid >> 4;
                m = (tid & 15) * 4 + i;
            } else {  // memory [m][k]
                m = tid >> 2;
                k = (tid & 3) * 4 + i;
            }
            float v = 0.f;
            if (m0 + m < M && k0 + k < K) v = transA ? Ab[(long long)(k0 + k) * lda + m0 + m] : Ab[(long long)(m0 + m) * lda + k0 + k];
            As[k][m] = v;
            int n, kb;
            if (transB) {  // memory [n][k]
                n = tid >> 2;
                kb = (tid & 3) * 4 + i;
            } else {  // memory [k][n]
                kb = tid >> 4;
                n = (tid & 15) * 4 + i;
            }
            float w = 0.f;
            if (n0 + n < N && k0 + kb < K) w = transB ? Bb[(long long)(n0 + n) * ldb + k0 + kb] : Bb[(long long)(k0 + kb) * ldb + n0 + n];
            Bs[kb][n] = w;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * s + half][wm * 32 + l31], Bs[2 * s + half][wn * 32 + l31], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) {
                float v = alpha * acc[r];
                if (beta != 0.f) v += beta * Cb[(long long)m * ldc + n];
                Cb[(long long)m * ldc + n] = v;
            }
        }
    }
}

// one workgroup per row; numerically the usual max-subtracted softmax of (scale * x)
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ out, long long ldo,
                                                               int N, float scale) {
    __shared__ float red[4];
    const float* xr = x + (long long)blockIdx.x * ldx;
    float* orow = out + (long long)blockIdx.x * ldo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 256) mx = fmaxf(mx, xr[i] * scale);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < N; i += 256) s += __expf(xr[i] * scale - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < N; i += 256) orow[i] = __expf(xr[i] * scale - mx) * inv;
}

// ds = scale * p * (dp - sum_j p_j dp_j)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, long long ldp, const float* __restrict__ dp,
                                                               long long lddp, float* __restrict__ ds, long long ldds, int N, float scale) {
    __shared__ float red[4];
    const float* pr = p + (long long)blockIdx.x * ldp;
    const float* dr = dp + (long long)blockIdx.x * lddp;
    float* o = ds + (long long)blockIdx.x * ldds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int i = tid; i < N; i += 256) s += pr[i] * dr[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    for (int i = tid; i < N; i += 256) o[i] = scale * pr[i] * (dr[i] - dot);
}

}  // namespace

extern "C" int idiff_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                           int transB, int64_t sA, int64_t sB, int64_t sC, int batch, float alpha, float beta, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "bgemm: bad args");
    IDIFF_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "bgemm: bad leading dims");
    IDIFF_CHECK_ARG(batch <= 65535, "bgemm: batch too large");
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, batch);
    hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, (long long)lda, (long long)ldb, (long long)ldc,
                       transA, transB, (long long)sA, (long long)sB, (long long)sC, alpha, beta);
    IDIFF_CHECK_LAUNCH("bgemm");
    return IDIFF_OK;
}

extern "C" int idiff_softmax_rows_fwd(const float* x, int64_t ldx, float* out, int64_t ldo, int R, int N, float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && out && R > 0 && N > 0 && ldx >= N && ldo >= N, "softmax_rows_fwd: bad args");
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, out, (long long)ldo, N, scale);
    IDIFF_CHECK_LAUNCH("softmax_rows_fwd");
    return IDIFF_OK;
}

extern "C" int idiff_softmax_rows_bwd(const float* p, int64_t ldp, const float* dp, int64_t lddp, float* ds, int64_t ldds, int R, int N,
                                      float scale, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(p && dp && ds && R > 0 && N > 0 && ldp >= N && lddp >= N && ldds >= N, "softmax_rows_bwd: bad args");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, p, (long long)ldp, dp, (long long)lddp, ds,
                       (long long)ldds, N, scale);
    IDIFF_CHECK_LAUNCH("softmax_rows_bwd");
    return IDIFF_OK;
}
