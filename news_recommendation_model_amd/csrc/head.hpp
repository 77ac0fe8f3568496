// Launchers of the BatchNorm column kernels (head.hip), shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nrm {
hipError_t colred_launch(int mode, const float* x, const float* dy, const float* mean, const float* rstd,
                         float* s0, float* s1, int R, int N, int ld, hipStream_t st);
hipError_t bn_apply_launch(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float* y, long R, int N, int ld, hipStream_t st);
hipError_t bn_bwd_launch(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                         const float* s0, const float* s1, const float* add, float* dx, long R, int N, int ld, int training,
                         hipStream_t st);
constexpr int CONCAT_MAX = 8;
struct ConcatTable { const float* src[CONCAT_MAX]; long ld[CONCAT_MAX]; int start[CONCAT_MAX]; int n; };
hipError_t concat_cols_launch(const ConcatTable& tab, float* out, long R, int ldo, int total, hipStream_t st);
hipError_t bn_finalize_launch(int stage, const float* s, float* out, float* running, int R, int N, float momentum, float eps,
                              hipStream_t st);
hipError_t mul_bwd_launch(const float* dy, const float* g, const float* e, float* dg, float* de, long R, int N,
                          int lddy, int ldg, int lde, int ldo, hipStream_t st);
hipError_t small_linear_relu_fwd_launch(const void* x, int x_is_f64, const float* w, const float* b, float* y, long R, int K, int N,
                                        int ldy, hipStream_t st);
hipError_t small_linear_relu_bwd_launch(const void* x, int x_is_f64, const float* w, const float* b, const float* dy, int lddy,
                                        long R, int K, int N, float* dwb, hipStream_t st);
}  // namespace nrm
