// Generic batched GEMM on the f32 matrix cores + row softmax forward/backward.
// Building blocks of the attention / token-side passes of the TRAINING path (forward of the unfused chains and
// every backward), where shapes are small and generality matters more than the last 20 % of MFMA utilisation.
//   C[b] = alpha * op(A[b]) (M x K) . op(B[b]) (K x N) + beta * C[b],   row-major, leading dims lda/ldb/ldc,
//   op(X) = X or X^T.  Workgroup tile TM x TN = 64x64 (waves 2x2) or 32x128 (waves 1x4, for the ScoreMapModule's
//   20-row query blocks), K chunks of 16 through LDS (both operands stored k-major so the MFMA operand reads are
//   unit-stride along m / n).  Long-K / tiny-output products (dQ = dS . mem^T with K = H*W) are split over K into
//   per-split partial tiles that a second kernel sums in a fixed order (deterministic, no atomics).
#include <math.h>

#include "common.h"

namespace {

constexpr int GK = 32;  // K chunk

// The next chunk's operands travel in registers while the current one is multiplied (r04: the loop had loaded, stored, synchronised
// and multiplied one 16-deep chunk at a time -- every chunk paid a full global-load latency, and the token-side products of the
// training step, K = 160 ... 256 on 16 workgroups, took 31-41 us each: 650 launches, 22 ms of an iteration).
template <int TM, int TN>
__global__ __launch_bounds__(256) void bgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                                    int K, long long lda, long long ldb, long long ldc, int transA, int transB, long long sA,
                                                    long long sB, long long sC, float alpha, float beta, int nsplit, int kper,
                                                    const float* __restrict__ colbias, long long sbias = 0) {
    constexpr int WN = TN / 32;  // waves along n; waves along m = 4 / WN = TM / 32
    constexpr int NA = TM * GK / 256, NB = TN * GK / 256;
    __shared__ float As[GK][TM + 1];
    __shared__ float Bs[GK][TN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int bz = blockIdx.z / nsplit, sp = blockIdx.z % nsplit;
    const float* Ab = A + (long long)bz * sA;
    const float* Bb = B + (long long)bz * sB;
    float* Cb = C + (long long)blockIdx.z * sC;  // with nsplit > 1, C is the partial buffer [batch*nsplit][M][N]
    const int kbeg = sp * kper, kend = min(K, kbeg + kper);
    // staging map (fixed per thread): element e = tid + 256 i of a chunk -> (k, m) / (k, n), unit stride in memory along e
    int ak[NA], am[NA], bk[NB], bn[NB];
    long long ao[NA], bo[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + i * 256;
        ak[i] = transA ? e / TM : e % GK;
        am[i] = transA ? e % TM : e / GK;
        ao[i] = m0 + am[i] < M ? (transA ? (long long)ak[i] * lda + m0 + am[i] : (long long)(m0 + am[i]) * lda + ak[i]) : -1;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + i * 256;
        bk[i] = transB ? e % GK : e / TN;
        bn[i] = transB ? e / GK : e % TN;
        bo[i] = n0 + bn[i] < N ? (transB ? (long long)(n0 + bn[i]) * ldb + bk[i] : (long long)bk[i] * ldb + n0 + bn[i]) : -1;
    }
    const long long astep = transA ? lda : 1, bstep = transB ? 1 : ldb;  // per unit of k
    float ra[NA], rb[NB];
    auto load_chunk = [&](int k0) {
        const float* Ak = Ab + (long long)k0 * astep;
        const float* Bk = Bb + (long long)k0 * bstep;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = (ao[i] >= 0 && k0 + ak[i] < kend) ? Ak[ao[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = (bo[i] >= 0 && k0 + bk[i] < kend) ? Bk[bo[i]] : 0.f;
    };
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (kbeg < kend) load_chunk(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
#pragma unroll
        for (int i = 0; i < NA; ++i) As[ak[i]][am[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[bk[i]][bn[i]] = rb[i];
        __syncthreads();
        if (k0 + GK < kend) load_chunk(k0 + GK);  // in flight during the MFMAs below
#pragma unroll
        for (int s = 0; s < GK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * s + half][wm * 32 + l31], Bs[2 * s + half][wn * 32 + l31], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
        const float cb = colbias ? colbias[(long long)bz * sbias + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) {
                float v = alpha * acc[r] + cb;
                if (beta != 0.f) v += beta * Cb[(long long)m * ldc + n];
                Cb[(long long)m * ldc + n] = v;
            }
        }
    }
}

// C[b][m][n] = (beta * C) + sum_s part[b*nsplit+s][m][n]
__global__ void bgemm_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int M, int N, long long ldc, long long sC, int nsplit,
                                    float beta) {
    const int b = blockIdx.y;
    const long long mn = (long long)M * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < mn; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[((long long)b * nsplit + k) * mn + i];
        float* c = C + (long long)b * sC + (i / N) * ldc + (i % N);
        *c = beta != 0.f ? beta * *c + s : s;
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

// K split: only when the output is tiny and K long (few workgroups would each walk a very long K)
inline int pick_nsplit(int M, int N, int K, int batch) {
    const int tm = M <= 32 ? 32 : 64, tn = M <= 32 ? 128 : 64;
    const long long tiles = (long long)((M + tm - 1) / tm) * ((N + tn - 1) / tn) * batch;
    if (K < 2048 || tiles >= 512) return 1;
    long long want = 1024 / tiles;
    long long maxs = K / 512;
    if (want > maxs) want = maxs;
    if (want > 128) want = 128;
    return want < 2 ? 1 : (int)want;
}

}  // namespace

extern "C" int64_t idiff_bgemm_ws_floats(int M, int N, int K, int batch) {
    const int ns = pick_nsplit(M, N, K, batch);
    return ns > 1 ? (int64_t)ns * batch * M * N : 0;
}

extern "C" int idiff_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                           int transB, int64_t sA, int64_t sB, int64_t sC, int batch, float alpha, float beta, float* ws, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "bgemm: bad args");
    IDIFF_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "bgemm: bad leading dims");
    const int ns = pick_nsplit(M, N, K, batch);
    IDIFF_CHECK_ARG(ns == 1 || ws, "bgemm: this shape needs a workspace of idiff_bgemm_ws_floats() floats");
    IDIFF_CHECK_ARG((long long)batch * ns <= 65535, "bgemm: batch too large");
    int kper = (K + ns - 1) / ns;
    kper = ((kper + GK - 1) / GK) * GK;
    hipStream_t st = (hipStream_t)stream;
    float* dst = ns > 1 ? ws : C;
    const long long dld = ns > 1 ? N : ldc, dsC = ns > 1 ? (long long)M * N : sC;
    const float a2 = alpha, b2 = ns > 1 ? 0.f : beta;
    if (M <= 32) {
        dim3 grid((N + 127) / 128, (M + 31) / 32, batch * ns);
        hipLaunchKernelGGL((bgemm_kernel<32, 128>), grid, dim3(256), 0, st, A, B, dst, M, N, K, (long long)lda, (long long)ldb, dld, transA, transB,
                           (long long)sA, (long long)sB, dsC, a2, b2, ns, kper, nullptr);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64, batch * ns);
        hipLaunchKernelGGL((bgemm_kernel<64, 64>), grid, dim3(256), 0, st, A, B, dst, M, N, K, (long long)lda, (long long)ldb, dld, transA, transB,
                           (long long)sA, (long long)sB, dsC, a2, b2, ns, kper, nullptr);
    }
    IDIFF_CHECK_LAUNCH("bgemm");
    if (ns > 1) {
        const long long mn = (long long)M * N;
        dim3 grid((unsigned)((mn + 255) / 256 > 1024 ? 1024 : (mn + 255) / 256), batch);
        hipLaunchKernelGGL(bgemm_reduce_kernel, grid, dim3(256), 0, st, ws, C, M, N, (long long)ldc, (long long)sC, ns, beta);
        IDIFF_CHECK_LAUNCH("bgemm_reduce");
    }
    return IDIFF_OK;
}

// idiff_bgemm with a per-column bias per batch entry (colbias [batch][N], batch stride sbias): the stacked token-side linears of the
// training step (r05: the same layer of a net's four ScoreMapModule decoders as ONE batch-4 launch).  No K split (K <= 1024 there).
extern "C" int idiff_bgemm_bias(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                                int transB, int64_t sA, int64_t sB, int64_t sC, int batch, float alpha, const float* colbias, int64_t sbias,
                                idiff_stream_t stream) {
    IDIFF_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "bgemm_bias: bad args");
    IDIFF_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "bgemm_bias: bad leading dims");
    hipStream_t st = (hipStream_t)stream;
    const int kper = ((K + GK - 1) / GK) * GK;
    if (M <= 32) {
        dim3 grid((N + 127) / 128, (M + 31) / 32, batch);
        hipLaunchKernelGGL((bgemm_kernel<32, 128>), grid, dim3(256), 0, st, A, B, C, M, N, K, (long long)lda, (long long)ldb, (long long)ldc, transA, transB,
                           (long long)sA, (long long)sB, (long long)sC, alpha, 0.f, 1, kper, colbias, (long long)sbias);
    } else {
        dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
        hipLaunchKernelGGL((bgemm_kernel<64, 64>), grid, dim3(256), 0, st, A, B, C, M, N, K, (long long)lda, (long long)ldb, (long long)ldc, transA, transB,
                           (long long)sA, (long long)sB, (long long)sC, alpha, 0.f, 1, kper, colbias, (long long)sbias);
    }
    IDIFF_CHECK_LAUNCH("bgemm_bias");
    return IDIFF_OK;
}

// y [R,N] = x [R,K] . w [N,K]^T (+ bias) on the f32 matrix cores: the batched-GEMM kernel with a per-column bias.  The training path's
// token-side linear layers (160 rows): idiff_linear_fwd's wave-per-column form walks K with one 256-byte row in flight, 20 us per launch.
extern "C" int idiff_linear_mfma_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* out, int64_t ldo, int R,
                                     int K, int N, idiff_stream_t stream) {
    IDIFF_CHECK_ARG(x && w && out && R > 0 && K > 0 && N > 0, "linear_mfma_fwd: bad args");
    IDIFF_CHECK_ARG(ldx >= K && ldw >= K && ldo >= N, "linear_mfma_fwd: bad leading dims");
    hipStream_t st = (hipStream_t)stream;
    if (R <= 32) {
        dim3 grid((N + 127) / 128, 1, 1);
        hipLaunchKernelGGL((bgemm_kernel<32, 128>), grid, dim3(256), 0, st, x, w, out, R, N, K, (long long)ldx, (long long)ldw, (long long)ldo, 0, 1,
                           0ll, 0ll, 0ll, 1.f, 0.f, 1, K, bias);
    } else {
        dim3 grid((N + 63) / 64, (R + 63) / 64, 1);
        hipLaunchKernelGGL((bgemm_kernel<64, 64>), grid, dim3(256), 0, st, x, w, out, R, N, K, (long long)ldx, (long long)ldw, (long long)ldo, 0, 1,
                           0ll, 0ll, 0ll, 1.f, 0.f, 1, K, bias);
    }
    IDIFF_CHECK_LAUNCH("linear_mfma_fwd");
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
