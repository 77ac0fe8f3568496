// Launchers of pool_loss.hip, shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nrm {
hipError_t bmm_rows_launch(const float* W, long wsb, long wsi, long wsj, const float* X, long xsb, int ldx,
                           float* out, long osb, int ldo, int B, int I, int J, int D, int accumulate, hipStream_t st);
hipError_t rowdot_launch(const float* g, long gsb, int ldg, const float* h, long hsb, int ldh, float* ds,
                         int B, int T, int H, int D, float* zero_out, int zero_n, hipStream_t st);
hipError_t loss_launch(const float* out, int out_stride, const void* label, int label_is_f64, const long* uid, const float* delta,
                       long n_delta, float alpha, int B, int T, float* loss_sum, float* dout, int dout_stride, float* ddelta, int* err,
                       hipStream_t st);
hipError_t adam_launch(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                       float wd, int step, int zero_grad, hipStream_t st);
hipError_t adam_dev_launch(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                           float wd, float* state, int zero_grad, hipStream_t st);
constexpr int GATHER_MAX = 128;
struct GatherTable { const float* src[GATHER_MAX]; long off[GATHER_MAX]; long cnt[GATHER_MAX]; };      // 3 KB of kernel arguments
hipError_t gather_flat_launch(const GatherTable& tab, int n, long max_count, float* flat, hipStream_t st);
hipError_t row_auc_launch(const float* score, const float* label, const int* len, int B, int T, float* auc, int* top1,
                          hipStream_t st);
}  // namespace nrm
