// Prototype for the split-operand F(4x4,3x3) kernel planned in DESIGN.md section 8.1 (NOT part of the library): the MFMA phase alone.
// 512 threads = 8 waves = 2 channel halves (32 co) x 4 position quarters (9 of the 36 Winograd positions each); every wave holds a
// 32 co x 32 tile block of 9 positions (144 accumulator registers) and, per position and 32-channel chunk, reads A (2 co blocks x 3
// planes) and B (2 tile blocks x 3 planes) as ds_read_b128 and issues 24 v_mfma_f32_16x16x32_bf16 (six products per block pair).
// Question: what MFMA rate survives when BOTH operands stream from LDS (12 KB per wave and position = 128 B/clk per CU)?
//   hipcc -O3 --offload-arch=gfx950 scripts/proto/w4x_mfma_proto.hip -o scripts/proto/w4x_mfma_proto && scripts/proto/w4x_mfma_proto
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NPOS = 36, PLANES = 3;
// one 32-channel chunk in LDS:  V [pos][plane][kg 4][tile 32] x 16 B = 36 x 6 KB = 216 KB would not fit: the prototype keeps NP_RES
// positions resident (a row group) and re-reads them, which is what the real kernel's row streaming does per group
constexpr int NP_RES = 6;                               // positions resident at a time (one Winograd row)
constexpr int V_POS = PLANES * 4 * 32 * 16;             // 6144 B
constexpr int U_POS = PLANES * 4 * 64 * 16;             // 12288 B
constexpr int LDS_BYTES = NP_RES * (V_POS + U_POS);     // 110592 B

__device__ __forceinline__ floatx4 mma(const uintx4& a, const uintx4& b, const floatx4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool A_FROM_LDS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void proto(const uintx4* __restrict__ ug, float* __restrict__ out, int iters,
                                                                                      long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Vs = smem;
    unsigned char* const Us = smem + NP_RES * V_POS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kgl = lane >> 4;
    const int cohalf = wave >> 2, pq = wave & 3;
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uintx4*>(smem)[i] = uintx4{0x3f803f80u + i, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
    __syncthreads();
    floatx4 acc[9][2][2];
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[p][a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const vrd = Vs + kgl * 512 + n16 * 16;                 // + (plane * 4) * 512 + tb * 256
    const unsigned char* const urd = Us + kgl * 1024 + (32 * cohalf + n16) * 16;  // + (plane * 4) * 1024 + cb * 256
    const uintx4* const ugl = ug + kgl * 64 + 32 * cohalf + n16;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int pr = (p + pq) % NP_RES;  // the resident slot this wave's position maps to
            uintx4 av[2][3], bv[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (A_FROM_LDS) av[h][pl] = *reinterpret_cast<const uintx4*>(urd + pr * U_POS + pl * 4096 + h * 256);
                    else av[h][pl] = ugl[(pr * 3 + pl) * 256 + h * 16 + (it & 1) * 36 * 768];
                    bv[h][pl] = *reinterpret_cast<const uintx4*>(vrd + pr * V_POS + pl * 2048 + h * 256);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    floatx4 c = acc[p][a][b];
                    c = mma(av[a][2], bv[b][0], c);
                    c = mma(av[a][1], bv[b][1], c);
                    c = mma(av[a][0], bv[b][2], c);
                    c = mma(av[a][1], bv[b][0], c);
                    c = mma(av[a][0], bv[b][1], c);
                    c = mma(av[a][0], bv[b][0], c);
                    acc[p][a][b] = c;
                }
            __builtin_amdgcn_sched_barrier(0);  // one position's operands live at a time (hipcc otherwise hoists all nine positions' loads)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) s += acc[p][a][b].x + acc[p][a][b].y + acc[p][a][b].z + acc[p][a][b].w;
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool A_FROM_LDS>
int run(const char* name, int ncu) {
    const int iters = 200;
    float* out;
    long long* cyc;
    uintx4* ug;
    CK(hipMalloc(&out, (size_t)ncu * 512 * 4));
    CK(hipMalloc(&cyc, 8));
    CK(hipMalloc(&ug, (size_t)2 * 36 * 768 * 16));
    CK(hipMemset(ug, 0x3f, (size_t)2 * 36 * 768 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(proto<A_FROM_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(proto<A_FROM_LDS>, dim3(ncu), dim3(512), LDS_BYTES, 0, ug, out, 10, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(proto<A_FROM_LDS>, dim3(ncu), dim3(512), LDS_BYTES, 0, ug, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long h = 0;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    // per workgroup and iteration: 8 waves x 9 positions x 24 MFMAs of 16x16x32 (16 cycles each at the matrix peak: 2 waves per SIMD)
    const double mfma_per_simd = 2.0 * 9 * 24 * iters;
    const double flops = (double)ncu * 8 * 9 * 24 * iters * 2.0 * 16 * 16 * 32;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s bf16 (= %6.1f fp32-equivalent at 6 products)  wave-0 cycles per MFMA slot %.2f (16 = peak)\n", name, ms, flops / ms / 1e9,
           flops / 6 / ms / 1e9, (double)h / mfma_per_simd);
    return 0;
}

int main() {
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    printf("CUs %d\n", ncu);
    if (run<true>("A and B from LDS", ncu)) return 1;
    if (run<false>("A from L2, B from LDS", ncu)) return 1;
    return 0;
}
