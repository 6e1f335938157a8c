// reduce.hip -- ONE launch that combines the per-workgroup partial slabs of MANY kernels.
//
// The backward of the student leaves ~65 small "sum the partial slabs" jobs behind: the LayerNorm parameter gradients (30 layers:
// [nblk][2][C] partials, csrc/layernorm.hip), the split-K Linear weight gradients (27 layers: [nslabs][M*N (+M)] slabs,
// csrc/align1x1.hip).  Each used to be its own ~10 us launch behind its producer; inside a replayed hipGraph a dependent launch
// costs about that much whatever it does, so they added up to ~0.6 ms of a 14 ms step.  Nothing reads these parameter gradients
// before the optimizer, so the binding (segdistill_amd/deferred.py) collects the jobs during the backward and issues them here in
// one launch per 80 jobs: out[i] = sum_s partials[s*n + i].  Deterministic (fixed summation order), no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

constexpr int kMaxJobs = 80;   // 80 x 36 bytes + 81 block offsets = 3.2 KB of the 4 KB kernel-argument segment (24 until round 3: four launches per step)

struct JobTable {   // passed BY VALUE as the kernel argument: no device-side table, nothing to copy, safe under graph capture
    const float *part[kMaxJobs];
    float *out[kMaxJobs];
    long n[kMaxJobs];
    int nslabs[kMaxJobs];
    int blk_begin[kMaxJobs + 1];
    int njobs;
};

// wave-uniform binary search of the job a workgroup belongs to (the table sits in the kernel-argument segment: scalar loads)
template <typename Table>
__device__ __forceinline__ int find_job(const Table &t, int blk) {
    int lo = 0, hi = t.njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk >= t.blk_begin[mid]) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// block = 4 slab groups x 64 consecutive outputs (256-byte rows per wave: coalesced).  Slab group `grp` walks slabs grp, grp + 4, ... with
// four accumulators (slab s goes to accumulator (s / 4) % 4).  Round 3: SIXTEEN slab rows are requested before any is added (four were: one
// dword per lane per request, the walk ran at the load latency -- 0.12 ms per step for ~100 MB); each accumulator still receives its slabs in
// walk order, so the sums are bit-identical to the four-in-flight form (tests/test_deferred_gpu.py emulates that order in torch).
__global__ __launch_bounds__(256) void multi_slab_reduce(const JobTable t) {
    __shared__ float red[4][64];
    const int j = find_job(t, (int)blockIdx.x);
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = (long)((int)blockIdx.x - t.blk_begin[j]) * 64 + o;
    const long n = t.n[j];
    const int ns = t.nslabs[j];
    const float *__restrict__ p = t.part[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int s = grp;
        for (; s + 60 < ns; s += 64) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p[(long)(s + 4 * q) * n + i];
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                s0 += v[q];
                s1 += v[q + 1];
                s2 += v[q + 2];
                s3 += v[q + 3];
            }
        }
        for (; s + 12 < ns; s += 16) {
            const float a = p[(long)s * n + i], b = p[(long)(s + 4) * n + i], c = p[(long)(s + 8) * n + i], d = p[(long)(s + 12) * n + i];
            s0 += a;
            s1 += b;
            s2 += c;
            s3 += d;
        }
        for (; s < ns; s += 4) s0 += p[(long)s * n + i];
    }
    red[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) t.out[j][i] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
}

// ---- column sums of a token-major matrix [rows][C] (the bias gradient of a Linear / 1x1 conv: db = sum over tokens of dY) --------
// ATen's sum(0) over a tall matrix is a multi-block reduction that first zeroes a semaphore buffer: a memset node plus the reduce
// kernel, ~15 us per bias inside a replayed graph, 34 of them per step.  Here: ONE launch that leaves per-workgroup partials
// [nblk][C], combined by the batched pass above (deferred to the end of the backward, or at once).
template <typename T> struct CS;
template <> struct CS<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
};
template <> struct CS<bf16_t> {
    static __device__ __forceinline__ float4 load(const bf16_t *p) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
};

template <typename T>
__global__ __launch_bounds__(256) void colsum_partials(const T *__restrict__ x, float *__restrict__ part, long rows, int C, int cg) {
    extern __shared__ float red[];   // [rpb][C]
    const int j0 = threadIdx.x & (cg - 1), rg = threadIdx.x / cg, rpb = 256 / cg, cv = C / 4;
    const long stride = (long)gridDim.x * rpb;
    for (int jj = j0; jj < cv; jj += cg) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), u = make_float4(0.f, 0.f, 0.f, 0.f);
        long row = (long)blockIdx.x * rpb + rg;
        for (; row + stride < rows; row += 2 * stride) {
            const float4 a = CS<T>::load(x + row * C + 4 * jj), b = CS<T>::load(x + (row + stride) * C + 4 * jj);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            u.x += b.x; u.y += b.y; u.z += b.z; u.w += b.w;
        }
        if (row < rows) {
            const float4 a = CS<T>::load(x + row * C + 4 * jj);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        float *r = red + (size_t)rg * C + 4 * jj;
        r[0] = s.x + u.x; r[1] = s.y + u.y; r[2] = s.z + u.z; r[3] = s.w + u.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
        float t = 0.f;
        for (int r = 0; r < rpb; ++r) t += red[(size_t)r * C + e];
        part[(size_t)blockIdx.x * C + e] = t;
    }
}

// The same for MANY matrices in one launch (the deferred form: the binding keeps the dY tensors of a backward alive and sums their
// columns at its end, so that the per-layer launches leave the critical chain).  Table by value, like JobTable.
struct ColsumTable {
    const void *x[kMaxJobs];
    float *part[kMaxJobs];
    long rows[kMaxJobs];
    int C[kMaxJobs];
    int cg[kMaxJobs];
    int blk_begin[kMaxJobs + 1];
    int njobs;
};

template <typename T>
__global__ __launch_bounds__(256) void multi_colsum_partials(const ColsumTable t) {
    extern __shared__ float red[];   // [rpb][C] of the widest job
    const int j = find_job(t, (int)blockIdx.x);
    const T *__restrict__ x = static_cast<const T *>(t.x[j]);
    float *__restrict__ part = t.part[j];
    const long rows = t.rows[j];
    const int C = t.C[j], cg = t.cg[j];
    const int blk = (int)blockIdx.x - t.blk_begin[j], nblk = t.blk_begin[j + 1] - t.blk_begin[j];
    const int j0 = threadIdx.x & (cg - 1), rg = threadIdx.x / cg, rpb = 256 / cg, cv = C / 4;
    const long stride = (long)nblk * rpb;
    for (int jj = j0; jj < cv; jj += cg) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), u = make_float4(0.f, 0.f, 0.f, 0.f);
        long row = (long)blk * rpb + rg;
        // eight rows requested before any is added (two were: the walk ran at the load latency -- 156 us for the ~280 MB of config 5's bf16
        // gradients, 1.8 TB/s); even rows still go to `s` and odd rows to `u` in walk order, so the sums are bit-identical
        for (; row + 7 * stride < rows; row += 8 * stride) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = CS<T>::load(x + (row + q * stride) * C + 4 * jj);
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w;
                u.x += v[q + 1].x; u.y += v[q + 1].y; u.z += v[q + 1].z; u.w += v[q + 1].w;
            }
        }
        for (; row + stride < rows; row += 2 * stride) {
            const float4 a = CS<T>::load(x + row * C + 4 * jj), b = CS<T>::load(x + (row + stride) * C + 4 * jj);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            u.x += b.x; u.y += b.y; u.z += b.z; u.w += b.w;
        }
        if (row < rows) {
            const float4 a = CS<T>::load(x + row * C + 4 * jj);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        float *r = red + (size_t)rg * C + 4 * jj;
        r[0] = s.x + u.x; r[1] = s.y + u.y; r[2] = s.z + u.z; r[3] = s.w + u.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
        float v = 0.f;
        for (int r = 0; r < rpb; ++r) v += red[(size_t)r * C + e];
        part[(size_t)blk * C + e] = v;
    }
}

int colsum_cg(int C) {
    int p = 1;
    while (p < C / 4 && p < 256) p <<= 1;
    return p;
}
int colsum_blocks(long rows, int C) {
    const long rpb = 256 / colsum_cg(C);
    long n = (rows + rpb * 16 - 1) / (rpb * 16);
    if (n > 256) n = 256;
    return (int)(n < 1 ? 1 : n);
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_colsum_blocks(long rows, int C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    return sd::colsum_blocks(rows, C);
}

int sd_multi_colsum_partials(const sd_colsum_job *jobs, int njobs, int dtype, void *stream) {
    if (njobs < 0) return SD_E_SHAPE;
    if (njobs == 0) return SD_OK;
    if (!jobs) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < njobs; base += sd::kMaxJobs) {
        sd::ColsumTable t{};
        const int cnt = njobs - base < sd::kMaxJobs ? njobs - base : sd::kMaxJobs;
        long blocks = 0;
        size_t lds = 0;
        for (int k = 0; k < cnt; ++k) {
            const sd_colsum_job &q = jobs[base + k];
            if (!q.x || !q.partials) return SD_E_NULL;
            if (q.rows <= 0 || q.C <= 0) return SD_E_SHAPE;
            if (q.C % 4 || q.C > 8192) return SD_E_UNSUPPORTED;
            if (reinterpret_cast<uintptr_t>(q.x) & (dtype == SD_F32 ? 15 : 7)) return SD_E_ALIGN;
            const int cg = sd::colsum_cg(q.C);
            const size_t need = (size_t)(256 / cg) * q.C * sizeof(float);
            if (need > 64 * 1024) return SD_E_UNSUPPORTED;
            lds = need > lds ? need : lds;
            t.x[k] = q.x;
            t.part[k] = q.partials;
            t.rows[k] = q.rows;
            t.C[k] = q.C;
            t.cg[k] = cg;
            t.blk_begin[k] = (int)blocks;
            blocks += sd::colsum_blocks(q.rows, q.C);      // partials: [sd_colsum_blocks(rows, C)][C]
        }
        t.blk_begin[cnt] = (int)blocks;
        t.njobs = cnt;
        if (dtype == SD_F32) hipLaunchKernelGGL(sd::multi_colsum_partials<float>, dim3((unsigned)blocks), dim3(256), lds, st, t);
        else hipLaunchKernelGGL(sd::multi_colsum_partials<sd::bf16_t>, dim3((unsigned)blocks), dim3(256), lds, st, t);
    }
    return (int)hipGetLastError();
}

int sd_colsum_partials(const void *x, int dtype, long rows, int C, float *partials, size_t partials_bytes, void *stream) {
    if (!x || !partials) return SD_E_NULL;
    if (dtype != SD_F32 && dtype != SD_BF16) return SD_E_DTYPE;
    if (rows <= 0 || C <= 0) return SD_E_SHAPE;
    if (C % 4 || C > 8192) return SD_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) & (dtype == SD_F32 ? 15 : 7)) || (reinterpret_cast<uintptr_t>(partials) & 3)) return SD_E_ALIGN;
    const int cg = sd::colsum_cg(C), nblk = sd::colsum_blocks(rows, C);
    if (partials_bytes < (size_t)nblk * C * sizeof(float)) return SD_E_WORKSPACE;
    const size_t lds = (size_t)(256 / cg) * C * sizeof(float);
    if (lds > 64 * 1024) return SD_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SD_F32)
        hipLaunchKernelGGL(sd::colsum_partials<float>, dim3(nblk), dim3(256), lds, st, (const float *)x, partials, rows, C, cg);
    else
        hipLaunchKernelGGL(sd::colsum_partials<sd::bf16_t>, dim3(nblk), dim3(256), lds, st, (const sd::bf16_t *)x, partials, rows, C, cg);
    return (int)hipGetLastError();
}

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
