// cgd_tok_device.h -- what the token-major criterion (cgd_tok.hip) shares with the fused align + criterion kernels (align_tok.hip): the job-table
// limits, the per-(image, chunk block, channel) partial records' geometry and the ONE finish launch that folds the records of any number of stages
// into row statistics and losses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgd_device.h"

namespace sd {

constexpr int kTokMaxJobs = 8;

template <typename Table>
__device__ __forceinline__ int tok_find_job(const Table &t, int blk) {       // wave-uniform; at most kTokMaxJobs entries
    int j = 0;
    while (j + 1 < t.njobs && blk >= t.blk_begin[j + 1]) ++j;
    return j;
}

// finish: rows + loss of every stage in ONE launch (cgd_tok.hip: cgd_tok_finish).  part[j] holds one RowPart per (image b, chunk block k < nkb,
// channel c): part[(b * nkb + k) * C + c]; blk_begin counts workgroups of four rows.
struct TokFinTable {
    const RowPart *part[kTokMaxJobs];
    const int32_t *perm[kTokMaxJobs];
    float *row_lse2[kTokMaxJobs];
    float *row_kl[kTokMaxJobs];
    float *loss[kTokMaxJobs];
    int B[kTokMaxJobs], C[kTokMaxJobs], g[kTokMaxJobs], G[kTokMaxJobs], nkb[kTokMaxJobs];
    float c2[kTokMaxJobs], inv_tau[kTokMaxJobs], loss_scale[kTokMaxJobs];
    int blk_begin[kTokMaxJobs + 1];
    int njobs;
};

// the arrival tickets of the finish launch live behind job 0's partials (16-byte aligned; the *_workspace_bytes queries reserve the room); they must
// be zeroed by a launch that precedes the finish launch on the stream (the scan kernels do it)
inline size_t tok_part_bytes(int B, int C, int nkb) { return ((size_t)B * C * nkb * sizeof(RowPart) + 15) & ~(size_t)15; }
constexpr size_t kTokTicketBytes = 64;

// enqueue the finish launch for a filled table (fin.blk_begin[fin.njobs] workgroups)
int tok_finish_launch(const TokFinTable &fin, unsigned *tickets, hipStream_t st);

}  // namespace sd
