// Probe (round 4): what `buffer_load_dword[x4] ... offen lds` (LDS-DMA) does on gfx950 --
//   1. where a wave-instruction's lanes land (M0 base + lane * size?), for 4- and 16-byte loads;
//   2. what a lane whose offset fails the range check writes: 0.0, or nothing (the LDS word keeps its old value)?
//   3. the same for a resource with zero records.
// hipcc -O3 --offload-arch=gfx950 scripts/proto/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(const float* src, float* out, int records) {
    extern __shared__ float sm[];  // 2048 floats
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = -7.0f;  // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, records, 0x00020000);
    const int lane = threadIdx.x & 63;
    // (a) dwordx4: lane l reads src bytes [l*16, l*16+16) -> LDS base 0
    unsigned v16 = lane * 16;
    if (lane == 5) v16 = 0xffffffffu;  // out of range lane
    unsigned base0 = 0, base1 = 4096, zero = 0;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(v16), "s"(rs), "s"(base0), "s"(zero) : "memory");
    // (b) dword: lane l reads src float 256 + (63 - l) -> LDS base 4096 bytes (float 1024): per-lane SOURCE, lane-linear destination
    unsigned v4 = (256 + (63 - lane)) * 4;
    if (lane == 9) v4 = 0xffffffffu;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds" ::"v"(v4), "s"(rs), "s"(base1), "s"(zero) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = sm[i];
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int records : {0x7fffffff, 0}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 2048 * 4, 0, d, o, records);
        std::vector<float> r(2048);
        hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
        printf("records=%d\n  x4 : lane0 %g %g %g %g | lane1 %g %g | lane4 %g | lane5(OOB) %g %g %g %g | lane6 %g | lane63 %g %g %g %g | after %g\n", records, r[0], r[1], r[2], r[3], r[4], r[5], r[16], r[20], r[21], r[22], r[23], r[24], r[252], r[253], r[254], r[255], r[256]);
        printf("  x1 : lane0 %g lane1 %g lane8 %g lane9(OOB) %g lane10 %g lane63 %g after %g\n", r[1024], r[1025], r[1032], r[1033], r[1034], r[1087], r[1088]);
    }
    return 0;
}
