// up_device.h -- on-the-fly bilinear up-sampling helpers (align_corners=False, integer factor F)
// shared by the fused-upsample kernels (cgd_up.hip, ce_up.hip).
//
// Geometry: the F output rows Y in [F*j - F/2, F*j + F/2) lie between tap rows j-1 and j ("gap j",
// j = 0..h) with weight lambda_q = (q + 0.5)/F on row j, q = Y - (F*j - F/2); rows outside [0,h)
// clamp, so gap 0 and gap h are half gaps whose outputs equal the edge row.  Same along x: a thread
// that owns tap column kx produces output columns F*kx .. F*kx+F-1 (second half of x-gap kx, first
// half of x-gap kx+1).
#pragma once
#include "cgd_device.h"

namespace sd {

// Horizontal interpolation of one tap row at this thread's F output columns.
template <typename T, int F>
__device__ __forceinline__ void hrow(const T *__restrict__ row, int kx, int w, float (&o)[F]) {
    const float a = VecIO<T>::load1(row + max(kx - 1, 0));
    const float b = VecIO<T>::load1(row + kx);
    const float c = VecIO<T>::load1(row + min(kx + 1, w - 1));
    const float dl = b - a, dr = c - b;
#pragma unroll
    for (int rx = 0; rx < F; ++rx) {
        if (rx < F / 2) o[rx] = fmaf((rx + F / 2 + 0.5f) / F, dl, a);
        else o[rx] = fmaf((rx - F / 2 + 0.5f) / F, dr, b);
    }
}

}  // namespace sd
