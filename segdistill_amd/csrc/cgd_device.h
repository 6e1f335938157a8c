// cgd_device.h -- device helpers shared by the CGD kernels (R1 streaming, R2 fused-upsample)
// and the pixel-wise kernel: online-softmax row partials, wave64/LDS combination, 16-byte IO.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;          // independent 16-byte loads per operand in flight per lane
constexpr float kNegBig = -1.0e30f; // finite stand-in for -inf in running maxima

struct RowPart {  // one partial of a row: raw-unit maxima, base-2-scaled sums
    float ms, zs, mt, zt, a;
};

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- 16-byte vector access for the two storage types ------------------------------------
template <typename T> struct VecIO;
template <> struct VecIO<float> {
    static constexpr int N = 4;
    typedef float raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const float *p, float (&o)[4]) {
        raw_t v = __builtin_nontemporal_load(reinterpret_cast<const raw_t *>(p));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    template <bool NT>
    static __device__ __forceinline__ void store(float *p, const float (&o)[4]) {
        raw_t v = {o[0], o[1], o[2], o[3]};
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<raw_t *>(p));
        else *reinterpret_cast<raw_t *>(p) = v;
    }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};
template <> struct VecIO<bf16_t> {
    static constexpr int N = 8;
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const bf16_t *p, float (&o)[8]) {
        raw_t v = __builtin_nontemporal_load(reinterpret_cast<const raw_t *>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(v[i] << 16);
            o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    template <bool NT>
    static __device__ __forceinline__ void store(bf16_t *p, const float (&o)[8]) {
        raw_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (unsigned)f32_to_bf16(o[2 * i]) | ((unsigned)f32_to_bf16(o[2 * i + 1]) << 16);
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<raw_t *>(p));
        else *reinterpret_cast<raw_t *>(p) = v;
    }
    static __device__ __forceinline__ float load1(const bf16_t *p) { return __uint_as_float((unsigned)p->bits << 16); }
    static __device__ __forceinline__ void store1(bf16_t *p, float v) { p->bits = f32_to_bf16(v); }
};

// Fold n elements (raw s, raw t) into the lane's running state.
template <int N>
__device__ __forceinline__ void fold(RowPart &st, const float (&s)[N], const float (&t)[N], float c2) {
    float mxs = s[0], mxt = t[0];
#pragma unroll
    for (int i = 1; i < N; ++i) { mxs = fmaxf(mxs, s[i]); mxt = fmaxf(mxt, t[i]); }
    const float nms = fmaxf(st.ms, mxs), nmt = fmaxf(st.mt, mxt);
    const float rs = ex2((st.ms - nms) * c2), rt = ex2((st.mt - nmt) * c2);
    const float os = -nms * c2, ot = -nmt * c2;
    float zs = st.zs * rs, zt = st.zt * rt, a = st.a * rt;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        zs += ex2(fmaf(s[i], c2, os));
        const float e = ex2(fmaf(t[i], c2, ot));
        zt += e;
        a = fmaf(e, t[i] - s[i], a);
    }
    st.ms = nms; st.zs = zs; st.mt = nmt; st.zt = zt; st.a = a;
}

__device__ __forceinline__ void merge(RowPart &p, const RowPart &q, float c2) {
    const float ms = fmaxf(p.ms, q.ms), mt = fmaxf(p.mt, q.mt);
    const float ps = ex2((p.ms - ms) * c2), qs = ex2((q.ms - ms) * c2);
    const float pt = ex2((p.mt - mt) * c2), qt = ex2((q.mt - mt) * c2);
    p.zs = p.zs * ps + q.zs * qs;
    p.zt = p.zt * pt + q.zt * qt;
    p.a = p.a * pt + q.a * qt;
    p.ms = ms; p.mt = mt;
}

// Combine the lane states of a workgroup of NW waves; result valid in thread 0.
template <int NW = kThreads / 64>
__device__ __forceinline__ RowPart block_combine(RowPart st, float c2) {
    __shared__ RowPart wave_part[NW];
    const float ms = wave_max(st.ms), mt = wave_max(st.mt);
    const float rs = ex2((st.ms - ms) * c2), rt = ex2((st.mt - mt) * c2);
    RowPart w;
    w.ms = ms; w.mt = mt;
    w.zs = wave_sum(st.zs * rs);
    w.zt = wave_sum(st.zt * rt);
    w.a = wave_sum(st.a * rt);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wave_part[wid] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NW; ++i) merge(w, wave_part[i], c2);
    }
    return w;
}


// Host-side launchers of the shared finalisation kernels (defined in cgd_kl.hip).
// part: [B*C*nchunk] partials laid out [(b*C + slot)*nchunk + k].
void launch_row_finalize(const RowPart *part, float *row_lse2, float *row_kl, float *loss, int B, int C, int g, int nchunk,
                         float c2, float inv_tau, float loss_scale, hipStream_t st);
// general layout: partial (image b, slot s, chunk k) at b*batch_stride + chan(s)*slot_stride + k*chunk_stride,
// chan(s) = chan_of_slot ? chan_of_slot[s] : s
void launch_row_finalize_strided(const RowPart *part, float *row_lse2, float *row_kl, float *loss, int B, int C, int g, int nchunk,
                                 long batch_stride, long slot_stride, long chunk_stride, const int32_t *chan_of_slot, float c2, float inv_tau,
                                 float loss_scale, hipStream_t st);

}  // namespace sd
