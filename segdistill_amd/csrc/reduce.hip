// reduce.hip -- ONE launch that combines the per-workgroup partial slabs of MANY kernels.
//
// The backward of the student leaves ~65 small "sum the partial slabs" jobs behind: the LayerNorm parameter gradients (30 layers:
// [nblk][2][C] partials, csrc/layernorm.hip), the split-K Linear weight gradients (27 layers: [nslabs][M*N (+M)] slabs,
// csrc/align1x1.hip).  Each used to be its own ~10 us launch behind its producer; inside a replayed hipGraph a dependent launch
// costs about that much whatever it does, so they added up to ~0.6 ms of a 14 ms step.  Nothing reads these parameter gradients
// before the optimizer, so the binding (segdistill_amd/deferred.py) collects the jobs during the backward and issues them here in
// one launch per 24 jobs: out[i] = sum_s partials[s*n + i].  Deterministic (fixed summation order), no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

constexpr int kMaxJobs = 24;

struct JobTable {   // passed BY VALUE as the kernel argument: no device-side table, nothing to copy, safe under graph capture
    const float *part[kMaxJobs];
    float *out[kMaxJobs];
    long n[kMaxJobs];
    int nslabs[kMaxJobs];
    int blk_begin[kMaxJobs + 1];
    int njobs;
};

// block = 4 slab groups x 64 consecutive outputs (256-byte rows per wave: coalesced)
__global__ __launch_bounds__(256) void multi_slab_reduce(const JobTable t) {
    __shared__ float red[4][64];
    int j = 0;
    while (j + 1 < t.njobs && (int)blockIdx.x >= t.blk_begin[j + 1]) ++j;   // wave-uniform
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - t.blk_begin[j]) * 64 + o;
    const long n = t.n[j];
    const int ns = t.nslabs[j];
    const float *__restrict__ p = t.part[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int s = grp;
        for (; s + 12 < ns; s += 16) {
            s0 += p[(long)s * n + i];
            s1 += p[(long)(s + 4) * n + i];
            s2 += p[(long)(s + 8) * n + i];
            s3 += p[(long)(s + 12) * n + i];
        }
        for (; s < ns; s += 4) s0 += p[(long)s * n + i];
    }
    red[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) t.out[j][i] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_multi_slab_reduce(const sd_reduce_job *jobs, int njobs, void *stream) {
    if (njobs < 0) return SD_E_SHAPE;
    if (njobs == 0) return SD_OK;
    if (!jobs) return SD_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < njobs; base += sd::kMaxJobs) {
        sd::JobTable t{};
        const int cnt = njobs - base < sd::kMaxJobs ? njobs - base : sd::kMaxJobs;
        long blocks = 0;
        for (int k = 0; k < cnt; ++k) {
            const sd_reduce_job &q = jobs[base + k];
            if (!q.partials || !q.out) return SD_E_NULL;
            if (q.n <= 0 || q.nslabs <= 0) return SD_E_SHAPE;
            t.part[k] = q.partials;
            t.out[k] = q.out;
            t.n[k] = q.n;
            t.nslabs[k] = q.nslabs;
            t.blk_begin[k] = (int)blocks;
            blocks += (q.n + 63) / 64;
            if (blocks > 0x7fffffffL) return SD_E_SHAPE;
        }
        t.blk_begin[cnt] = (int)blocks;
        t.njobs = cnt;
        hipLaunchKernelGGL(sd::multi_slab_reduce, dim3((unsigned)blocks), dim3(256), 0, st, t);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
