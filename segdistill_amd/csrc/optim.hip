// optim.hip -- one-launch AdamW over every trainable tensor of the student, gfx950.
//
// The reference trains with torch.optim.AdamW and per-parameter-group learning rates / weight decay (mmseg/apis/train.py:89;
// local_configs/exp_tab5/segformer_CGD+WS.py:60-64: lr 6e-5, weight_decay 0.01, paramwise_cfg: head lr x10, norm / pos_block no decay).  ATen's fused path is one multi_tensor_apply launch per (group, chunk
// list): 11 launches and 0.25 ms per step for Segformer-B0's ~200 small tensors (15 MB of parameters: ~17 us of HBM time).  Here ONE launch
// walks a block table: entry = (tensor, first element); a workgroup updates up to 4096 consecutive elements of one tensor; the decay and the
// parameter-group index come from the tensor's descriptor, the groups' current learning rates and the bias corrections from the arguments.  Elementwise over the STORAGE order:
// parameter, gradient and both moments must share one dense layout (contiguous or channels-last alike), which the caller checks.
// Optional per tensor: a bf16 SHADOW of the parameter, rewritten with the updated value -- under bf16 autocast the forward would cast every
// trainable fp32 weight again each step (one tiny kernel per weight and bias); with the shadow it reads what the optimizer already wrote.
// Arithmetic (fp32, the order of torch/optim/adamw.py::_single_tensor_adamw):
//   p *= 1 - lr*wd;  m += (g - m)(1 - b1);  v = v*b2 + (1 - b2) g*g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),  bc = 1 - b^k (fp64)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sd_common.h"

namespace sd {
namespace {

struct AdamTensor {        // 56 bytes; mirrored by segdistill_amd/engine/optim.py (numpy record '<u8 x5, <f4, <i4, <i8')
    float *p;
    const float *g;
    float *m;
    float *v;
    uint16_t *shadow;      // optional bf16 copy of the parameter, rewritten with the new value (what a bf16-autocast forward would cast anyway)
    float wd;
    int group_missed;      // bits 0-7: index into the learning-rate argument (the schedule changes it every step; the table stays as it is);
                           // bits 8-31: optimizer steps this tensor took no part in (torch counts steps per tensor: its bias corrections lag)
    long n;
};
constexpr int kAdamGroups = 8;
struct AdamLr {
    float lr[kAdamGroups];
};
struct AdamBlock {
    int tensor;
    int first;             // first element / kAdamChunk
};
constexpr int kAdamChunk = 4096;

__global__ __launch_bounds__(256) void adamw_multi(const AdamTensor *__restrict__ tensors, const AdamBlock *__restrict__ blocks, AdamLr lrs, double beta1d,
                                                    double beta2d, float eps, int step) {
    const AdamBlock blk = blocks[blockIdx.x];
    const AdamTensor t = tensors[blk.tensor];
    const long lo = (long)blk.first * kAdamChunk;
    const long hi = lo + kAdamChunk < t.n ? lo + kAdamChunk : t.n;
    const float lr = lrs.lr[t.group_missed & (kAdamGroups - 1)];
    // bias corrections of THIS tensor's step count, in fp64 like the Python scalars of torch/optim/adamw.py (1 - 0.999^k cancels badly in fp32)
    const double k = (double)(step - (t.group_missed >> 8));
    const float inv_bc1 = (float)(1.0 / (1.0 - pow(beta1d, k))), inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(beta2d, k)));
    const float beta2 = (float)beta2d, ob1 = (float)(1.0 - beta1d), ob2 = (float)(1.0 - beta2d);
    const float decay = 1.f - lr * t.wd, step_size = lr * inv_bc1;
    auto update = [&](float &p, float g, float &m, float &v) {
        p *= decay;
        m += (g - m) * ob1;
        v = v * beta2 + ob2 * g * g;
        p -= step_size * m / (sqrtf(v) * inv_bc2_sqrt + eps);
    };
    uint16_t *const sh = t.shadow;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    if (vec) {
        const long hi4 = lo + ((hi - lo) & ~3L);
        for (long i = lo + 4 * threadIdx.x; i < hi4; i += 4 * 256) {
            float4 p = *reinterpret_cast<float4 *>(t.p + i), m = *reinterpret_cast<float4 *>(t.m + i), v = *reinterpret_cast<float4 *>(t.v + i);
            const float4 g = *reinterpret_cast<const float4 *>(t.g + i);
            update(p.x, g.x, m.x, v.x); update(p.y, g.y, m.y, v.y); update(p.z, g.z, m.z, v.z); update(p.w, g.w, m.w, v.w);
            *reinterpret_cast<float4 *>(t.p + i) = p; *reinterpret_cast<float4 *>(t.m + i) = m; *reinterpret_cast<float4 *>(t.v + i) = v;
            if (sh) { sh[i] = f32_to_bf16(p.x); sh[i + 1] = f32_to_bf16(p.y); sh[i + 2] = f32_to_bf16(p.z); sh[i + 3] = f32_to_bf16(p.w); }
        }
        for (long i = hi4 + threadIdx.x; i < hi; i += 256) {
            update(t.p[i], t.g[i], t.m[i], t.v[i]);
            if (sh) sh[i] = f32_to_bf16(t.p[i]);
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            update(t.p[i], t.g[i], t.m[i], t.v[i]);
            if (sh) sh[i] = f32_to_bf16(t.p[i]);
        }
    }
}

}  // namespace
}  // namespace sd

extern "C" {

int sd_adamw_chunk(void) { return sd::kAdamChunk; }

int sd_adamw_max_groups(void) { return sd::kAdamGroups; }

int sd_adamw_multi(const void *tensors, const void *blocks, int nblocks, const float *group_lr, int ngroups, double beta1, double beta2, float eps,
                   int step, void *stream) {
    if (!tensors || !blocks || !group_lr) return SD_E_NULL;
    if (nblocks <= 0 || ngroups <= 0 || step <= 0 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return SD_E_SHAPE;
    if (ngroups > sd::kAdamGroups) return SD_E_UNSUPPORTED;
    sd::AdamLr lrs{};
    for (int i = 0; i < ngroups; ++i) lrs.lr[i] = group_lr[i];
    if ((reinterpret_cast<uintptr_t>(tensors) | reinterpret_cast<uintptr_t>(blocks)) & 7) return SD_E_ALIGN;
    hipLaunchKernelGGL(sd::adamw_multi, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const sd::AdamTensor *>(tensors), static_cast<const sd::AdamBlock *>(blocks), lrs, beta1, beta2, eps, step);
    return (int)hipGetLastError();
}

}  // extern "C"
