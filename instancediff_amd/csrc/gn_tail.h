// GroupNorm finalize as the tail of the conv launch that produced the partials (conv_wino4.hip, conv_wino4h.hip).
//
// The separate finalize launch (idiff_gn_finalize: 5.6 us of work) sits on the conv -> conv critical path of every ResBlock and, with
// the two nets on two streams, queues behind the other net's persistent grid (25 us measured, 76 launches per step).  Here the
// producing launch finishes the job itself, deterministically -- no atomics on data:
//   producer   every workgroup stores its partials write-through (sc1); when its last item is done each wave waits for its stores
//              (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and ONE lane adds 1 to the launch's arrival counter (agent scope);
//   finalizers the LAST min(grid, B * groups) arrivers -- told by the value their add returned -- poll the counter (sc1 loads, bounded)
//              until every workgroup has arrived, take an agent-scope acquire, and reduce the (sample, group) pairs dealt to them with
//              EXACTLY the arithmetic of gn_finalize_kernel (fp64, the same thread -> partial mapping and combination order on the first
//              256 threads): the (a, b) affine is bit-identical to the separate launch and independent of who came last;
//   clean-up   the last finalizer to finish zeroes the two counters: the buffer serves the layer's next launch (stream-ordered).
// Form and conditions: MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility" (sc1 payload, drained
// stores, one lane per workgroup signalling behind a barrier, counter poll, acquire before the loads; every load of the partials sc1).
// Every workgroup that polls is already resident and the ones it waits for need no resource it holds (finished workgroups free their
// CUs), so the wait cannot deadlock; it is bounded all the same and a timeout leaves the outputs untouched and raises a flag word.
#pragma once
#include "conv_args.h"

namespace idiff_detail {

__device__ __forceinline__ void gn_store_partial(float* p, float sum, float sumsq) {  // 8-byte write-through store
    const unsigned long long bits = (unsigned long long)__builtin_bit_cast(unsigned, sum) | ((unsigned long long)__builtin_bit_cast(unsigned, sumsq) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one (sample, group) pair; all threads of the workgroup call it (threads >= 256 only pass the barriers); wsum: 8 doubles of LDS
__device__ __forceinline__ void gn_finalize_pair(const ConvArgs& a, int b, int g, double* wsum) {
    const GnTail& t = a.gn;
    const int C = a.Cout, cpg = C / t.groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the partials of (b, g): sc1 (L1-bypassing) 8-byte buffer loads -- ordinary loads to the compiler, so the eight or so a thread
    // makes are all in flight at once (agent-scope atomic loads would be issued one round trip at a time)
    const float* sp = a.stats + (long long)b * a.ntiles * C * 2 + (long long)g * cpg * 2;
    const unsigned long long spv = reinterpret_cast<unsigned long long>(sp);
    // __builtin_amdgcn_readfirstlane returns a (signed) int: both halves go through `unsigned` before they are widened.  The r03 form
    // OR-ed the int-typed low half into the 64-bit value directly, which SIGN-extends it -- every partials row whose address has bit
    // 31 set got 0xffff as the upper 16 bits of its base (s_bfe_i64 / s_or_b64 / s_and_b32 0xffff in the ISA): the intermittent
    // "Memory access fault by GPU" of the fused tail.
    const unsigned sp_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)spv);
    const unsigned sp_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(spv >> 32));
    const float* sps = reinterpret_cast<const float*>(((unsigned long long)sp_hi << 32) | (unsigned long long)sp_lo);
    // true extent of what this pair reads: rows 0 .. ntiles-1 of C channels, cpg channels from the group's first (an offset beyond it
    // fails the range check and reads zero instead of faulting)
    const int extent = (((a.ntiles - 1) * C) + cpg) * 8;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sps), 0, extent, 0x00020000);
    constexpr int AUX_SC1 = 1 << 4;
    double s = 0.0, q = 0.0;
    const int cl = tid % cpg, tph = tid / cpg, tstep = 256 / cpg;
    if (tid < 256 && tph < tstep)
        for (int tt = tph; tt < a.ntiles; tt += tstep) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (tt * C + cl) * 8, 0, AUX_SC1));
            s += (double)v.x;
            q += (double)v.y;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0 && wave < 4) wsum[wave * 2] = s, wsum[wave * 2 + 1] = q;
    __syncthreads();
    s = ((wsum[0] + wsum[2]) + wsum[4]) + wsum[6];
    q = ((wsum[1] + wsum[3]) + wsum[5]) + wsum[7];
    const double cnt = (double)cpg * (double)(a.Hout * a.Wout);
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)t.eps);
    if (t.mean_rstd && tid == 0) {
        t.mean_rstd[((long long)b * t.groups + g) * 2 + 0] = (float)mean;
        t.mean_rstd[((long long)b * t.groups + g) * 2 + 1] = (float)rstd;
    }
    for (int i = tid; i < cpg; i += blockDim.x) {
        const int c = g * cpg + i;
        const float ga = t.gamma ? t.gamma[c] : 1.f, be = t.beta ? t.beta[c] : 0.f;
        float av = (float)rstd * ga;
        float bb = be - (float)mean * av;
        if (t.film) {
            const float sc = 1.f + t.film[(long long)b * t.film_ld + c];
            const float sh = t.film[(long long)b * t.film_ld + C + c];
            av *= sc;
            bb = bb * sc + sh;
        }
        t.out_a[(long long)b * C + c] = av;
        t.out_b[(long long)b * C + c] = bb;
    }
    __syncthreads();  // wsum is reused by the next pair
}

// called by every thread of every workgroup once, after the workgroup's last item (also by a workgroup that had no item);
// scratch: 80 bytes of LDS no longer in use (two control words, padding, eight doubles), 8-byte aligned
__device__ __forceinline__ void gn_arrive_and_finalize(const ConvArgs& a, float* scratch) {
    const GnTail& t = a.gn;
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left the CU
    __syncthreads();
    unsigned* sh = reinterpret_cast<unsigned*>(scratch);  // plain LDS accesses, ordered by the barriers (volatile made them flat_*)
    if (tid == 0) sh[0] = __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned mine = sh[0], G = gridDim.x;
    const unsigned npairs = (unsigned)a.B * (unsigned)t.groups;
    unsigned nfin = G < npairs ? G : npairs;
    if (nfin > t.max_finalizers) nfin = t.max_finalizers;
    if (mine < G - nfin) return;
    if (tid == 0) {
        unsigned spins = 0, seen;
        while ((seen = __hip_atomic_load(t.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < G && ++spins < (1u << 21)) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh[1] = seen >= G ? 1u : 0u;
    }
    __syncthreads();
    if (sh[1]) {  // uniform
        double* wsum = reinterpret_cast<double*>(scratch + 4);
        for (unsigned p = mine - (G - nfin); p < npairs; p += nfin) gn_finalize_pair(a, (int)(p / (unsigned)t.groups), (int)(p % (unsigned)t.groups), wsum);
    } else if (tid == 0) {
        __hip_atomic_store(t.ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // timeout flag (checked by tests)
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(t.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == nfin - 1) {  // every finalizer is past its poll: the counters can go back to zero for the layer's next launch
            __hip_atomic_store(t.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace idiff_detail
