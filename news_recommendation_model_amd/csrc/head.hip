// BatchNorm1d over the [B*T, N] head rows (reference models/user_model.py:18,32: nn.BatchNorm1d, eps 1e-5,
// biased variance for normalisation) as HBM-bound column kernels.  gfx950 only.
//   colred_kernel<0>   s0[n] += sum_r x                              (mean)
//   colred_kernel<1>   s0[n] += sum_r (x - mean)^2                   (two-pass variance: no cancellation)
//   colred_kernel<2>   s0[n] += sum_r dy ; s1[n] += sum_r dy * (x - mean) * rstd       (d beta, d gamma)
//   bn_apply_kernel    y = (x - mean) * rstd * gamma + beta
//   bn_bwd_kernel      dx = gamma * rstd * (dy - s0/R - xhat * s1/R)    (train)   |  gamma * rstd * dy  (eval)
// Layout: thread = one float4 of columns; blockDim (32, 8); a workgroup sweeps `rpb` rows of a 128-column slab
// and adds its partial sums with float atomics (R/rpb adds per column).
#include "common.hpp"
#include "head.hpp"

namespace nrm {

template <int MODE>
__global__ __launch_bounds__(256) void colred_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ s0, float* __restrict__ s1,
                                                     int R, int N, int ld, int rpb) {
    __shared__ f32x4 red[2][8][32];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = blockIdx.x * 128 + 4 * tx;
    const bool cok = col < N;                       // N % 4 == 0
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, mu = a0, rs = a0;
    if (cok && MODE >= 1) mu = *reinterpret_cast<const f32x4*>(mean + col);
    if (cok && MODE == 2) rs = *reinterpret_cast<const f32x4*>(rstd + col);
    const int r_lo = blockIdx.y * rpb, r_hi = min(R, r_lo + rpb);
    if (cok) {
        for (int r = r_lo + ty; r < r_hi; r += 8) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)r * ld + col);
            if (MODE == 0) a0 += xv;
            else if (MODE == 1) { const f32x4 d = xv - mu; a0 += d * d; }
            else {
                const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)r * ld + col);
                a0 += g;
                a1 += g * (xv - mu) * rs;
            }
        }
    }
    red[0][ty][tx] = a0;
    if (MODE == 2) red[1][ty][tx] = a1;
    __syncthreads();
    if (ty == 0 && cok) {
        f32x4 t0 = red[0][0][tx], t1 = red[1][0][tx];
#pragma unroll
        for (int y = 1; y < 8; ++y) { t0 += red[0][y][tx]; if (MODE == 2) t1 += red[1][y][tx]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(s0 + col + e, t0[e]);
            if (MODE == 2) atomicAdd(s1 + col + e, t1[e]);
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       long R, int N, int ld) {
    const int n4 = N >> 2;
    const long total = R * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / n4;
        const int col = (int)(i - r * n4) * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ld + col);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col), rs = *reinterpret_cast<const f32x4*>(rstd + col);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col), b = *reinterpret_cast<const f32x4*>(beta + col);
        *reinterpret_cast<f32x4*>(y + r * ld + col) = (xv - mu) * rs * g + b;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ s0,
                                                     const float* __restrict__ s1, const float* __restrict__ add,
                                                     float* __restrict__ dx, long R, int N, int ld, int training) {
    const int n4 = N >> 2;
    const long total = R * n4;
    const float invR = 1.0f / (float)R;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / n4;
        const int col = (int)(i - r * n4) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * ld + col);
        const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + col), ga = *reinterpret_cast<const f32x4*>(gamma + col);
        f32x4 out;
        if (training) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ld + col);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + col);
            const f32x4 a = *reinterpret_cast<const f32x4*>(s0 + col), b = *reinterpret_cast<const f32x4*>(s1 + col);
            const f32x4 xh = (xv - mu) * rs;
            out = ga * rs * (g - a * invR - xh * b * invR);
        } else {
            out = ga * rs * g;
        }
        // the rows feed a second consumer besides BatchNorm (the gate product, user_model.py:33): its gradient joins here
        if (add) out += *reinterpret_cast<const f32x4*>(add + r * ld + col);
        *reinterpret_cast<f32x4*>(dx + r * ld + col) = out;
    }
}

// train-mode statistics between the column reductions (was ~8 tiny ATen launches per BatchNorm call):
//   stage 0: mean = s0 / R;  running_mean = (1 - momentum) running_mean + momentum mean
//   stage 1: var = s1 / R (biased, normalises);  rstd = 1/sqrt(var + eps);
//            running_var = (1 - momentum) running_var + momentum var R/(R-1)      (unbiased, nn.BatchNorm1d)
__global__ __launch_bounds__(256) void bn_finalize_kernel(int stage, const float* __restrict__ s, float* __restrict__ out,
                                                          float* __restrict__ running, int R, int N, float momentum, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float v = s[i] / (float)R;
    if (stage == 0) {
        out[i] = v;
        if (running) running[i] = (1.0f - momentum) * running[i] + momentum * v;
    } else {
        out[i] = 1.0f / sqrtf(v + eps);
        if (running) running[i] = (1.0f - momentum) * running[i] + momentum * v * ((float)R / (float)(R > 1 ? R - 1 : 1));
    }
}

hipError_t bn_finalize_launch(int stage, const float* s, float* out, float* running, int R, int N, float momentum, float eps,
                              hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, stage, s, out, running, R, N, momentum, eps);
    return hipGetLastError();
}

// backward of the gate product y = g * e (reference models/user_model.py:33): dg = dy * e, de = dy * g in one pass
__global__ __launch_bounds__(256) void mul_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ g,
                                                      const float* __restrict__ e, float* __restrict__ dg,
                                                      float* __restrict__ de, long R, int N, int lddy, int ldg, int lde,
                                                      int ldo) {
    const int n4 = N >> 2;
    const long total = R * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / n4;
        const int col = (int)(i - r * n4) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * lddy + col);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + r * ldg + col);
        const f32x4 ev = *reinterpret_cast<const f32x4*>(e + r * lde + col);
        *reinterpret_cast<f32x4*>(dg + r * ldo + col) = d * ev;
        *reinterpret_cast<f32x4*>(de + r * ldo + col) = d * gv;
    }
}

hipError_t colred_launch(int mode, const float* x, const float* dy, const float* mean, const float* rstd,
                         float* s0, float* s1, int R, int N, int ld, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    const int rpb = 256;
    const dim3 grid((N + 127) / 128, (R + rpb - 1) / rpb), block(32, 8);
    if (mode == 0) hipLaunchKernelGGL(colred_kernel<0>, grid, block, 0, st, x, dy, mean, rstd, s0, s1, R, N, ld, rpb);
    else if (mode == 1) hipLaunchKernelGGL(colred_kernel<1>, grid, block, 0, st, x, dy, mean, rstd, s0, s1, R, N, ld, rpb);
    else if (mode == 2) hipLaunchKernelGGL(colred_kernel<2>, grid, block, 0, st, x, dy, mean, rstd, s0, s1, R, N, ld, rpb);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

static int ew_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

hipError_t bn_apply_launch(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float* y, long R, int N, int ld, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(R * (N >> 2))), dim3(256), 0, st, x, mean, rstd, gamma, beta, y, R, N, ld);
    return hipGetLastError();
}

hipError_t mul_bwd_launch(const float* dy, const float* g, const float* e, float* dg, float* de, long R, int N,
                          int lddy, int ldg, int lde, int ldo, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(mul_bwd_kernel, dim3(ew_blocks(R * (N >> 2))), dim3(256), 0, st, dy, g, e, dg, de, R, N, lddy, ldg, lde, ldo);
    return hipGetLastError();
}

hipError_t bn_bwd_launch(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                         const float* s0, const float* s1, const float* add, float* dx, long R, int N, int ld, int training,
                         hipStream_t st) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(ew_blocks(R * (N >> 2))), dim3(256), 0, st, x, dy, mean, rstd, gamma, s0, s1, add, dx, R, N, ld, training);
    return hipGetLastError();
}

// e = cat[parts...] along the columns (reference models/user_model.py:31, user_invariant_interest_model.py:81,88) in one launch:
// up to 8 row-major [R, w_i] sources (any leading dimension) -> out[R, ldo]; columns are copied dword by dword (the part
// widths of the head -- D_l, P, 8 -- need not be multiples of 4 in general)
template <int VEC>      // VEC = 4: every width, leading dimension and base address is a multiple of 4 floats (16 bytes)
__global__ __launch_bounds__(256) void concat_cols_kernel(const ConcatTable tab, float* __restrict__ out, unsigned R, int ldo, int total) {
    const unsigned tv = (unsigned)(total / VEC);
    const unsigned n = R * tv;                                          // host guarantees R * total < 2^31
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned r = i / tv;
        const int c = (int)(i - r * tv) * VEC;
        int k = 0;
#pragma unroll
        for (int j = 1; j < CONCAT_MAX; ++j) k += (j < tab.n && c >= tab.start[j]) ? 1 : 0;
        const float* src = tab.src[k] + (size_t)r * tab.ld[k] + (c - tab.start[k]);
        float* dst = out + (size_t)r * ldo + c;
        if (VEC == 4) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
        else *dst = *src;
    }
}

hipError_t concat_cols_launch(const ConcatTable& tab, float* out, long R, int ldo, int total, hipStream_t st) {
    if (R <= 0 || total <= 0) return hipSuccess;
    if (R * total >= (1L << 31)) return hipErrorInvalidValue;
    bool vec = (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int i = 0; i < tab.n; ++i)
        vec = vec && (tab.start[i] & 3) == 0 && (tab.ld[i] & 3) == 0 && (reinterpret_cast<uintptr_t>(tab.src[i]) & 15) == 0;
    vec = vec && (total & 3) == 0;
    if (vec) hipLaunchKernelGGL(concat_cols_kernel<4>, dim3(ew_blocks(R * (total / 4))), dim3(256), 0, st, tab, out, (unsigned)R, ldo, total);
    else     hipLaunchKernelGGL(concat_cols_kernel<1>, dim3(ew_blocks(R * total)), dim3(256), 0, st, tab, out, (unsigned)R, ldo, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// y = ReLU(x W^T + b) for a handful of input and output columns (reference models/user_instant_interest_model.py: the 3
// popularity scalars of a candidate -> 8 features).  As a GEMM this was a zero-padded copy of x, a K = 4 launch, a clamp --
// and the same again backwards; here: one thread per row forward, a register reduction over rows backward.
constexpr int SL_KMAX = 4, SL_NMAX = 8;

template <typename XT>
__global__ __launch_bounds__(256) void small_linear_relu_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ b, float* __restrict__ y,
                                                                    long R, int K, int N, int ldy) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float xk[SL_KMAX];
#pragma unroll
    for (int k = 0; k < SL_KMAX; ++k) xk[k] = k < K ? (float)x[r * K + k] : 0.f;
    float* yr = y + r * ldy;
    for (int n = 0; n < ldy; ++n) {
        float v = 0.f;
        if (n < N) {
            v = b ? b[n] : 0.f;
#pragma unroll
            for (int k = 0; k < SL_KMAX; ++k) if (k < K) v = fmaf(w[n * K + k], xk[k], v);
            v = fmaxf(v, 0.f);
        }
        yr[n] = v;                                                       // padding columns are written as 0
    }
}

// dwb[n*K + k] += sum_r g[r,n] x[r,k],  dwb[N*K + n] += sum_r g[r,n],  g = dy where the pre-activation is positive
template <typename XT>
__global__ __launch_bounds__(256) void small_linear_relu_bwd_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ b, const float* __restrict__ dy,
                                                                    int lddy, long R, int K, int N, float* __restrict__ dwb) {
    float acc[SL_NMAX][SL_KMAX + 1];
#pragma unroll
    for (int n = 0; n < SL_NMAX; ++n)
#pragma unroll
        for (int k = 0; k <= SL_KMAX; ++k) acc[n][k] = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long)gridDim.x * 256) {
        float xk[SL_KMAX];
#pragma unroll
        for (int k = 0; k < SL_KMAX; ++k) xk[k] = k < K ? (float)x[r * K + k] : 0.f;
#pragma unroll
        for (int n = 0; n < SL_NMAX; ++n) {
            if (n < N) {
                float pre = b ? b[n] : 0.f;
#pragma unroll
                for (int k = 0; k < SL_KMAX; ++k) if (k < K) pre = fmaf(w[n * K + k], xk[k], pre);
                const float g = pre > 0.f ? dy[r * lddy + n] : 0.f;
#pragma unroll
                for (int k = 0; k < SL_KMAX; ++k) acc[n][k] = fmaf(g, xk[k], acc[n][k]);
                acc[n][SL_KMAX] += g;
            }
        }
    }
    // wave sums -> LDS -> one float atomic per output and WORKGROUP (all workgroups hit the same N * (K + 1) addresses: with one
    // atomic per wave this tail was most of the kernel)
    __shared__ float part[4][SL_NMAX * (SL_KMAX + 1)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < SL_NMAX; ++n)
#pragma unroll
        for (int k = 0; k <= SL_KMAX; ++k) {
            const float v = wave_sum64(acc[n][k]);
            if (lane == 0) part[wave][n * (SL_KMAX + 1) + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < SL_NMAX * (SL_KMAX + 1)) {
        const int n = threadIdx.x / (SL_KMAX + 1), k = threadIdx.x - n * (SL_KMAX + 1);
        const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (n < N && (k < K || k == SL_KMAX) && v != 0.f) atomicAdd(dwb + (k == SL_KMAX ? N * K + n : n * K + k), v);
    }
}

hipError_t small_linear_relu_fwd_launch(const void* x, int x_is_f64, const float* w, const float* b, float* y, long R, int K, int N,
                                        int ldy, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    if (K < 1 || K > SL_KMAX || N < 1 || N > SL_NMAX || ldy < N) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((R + 255) / 256));
    if (x_is_f64) hipLaunchKernelGGL(small_linear_relu_fwd_kernel<double>, grid, dim3(256), 0, st, (const double*)x, w, b, y, R, K, N, ldy);
    else          hipLaunchKernelGGL(small_linear_relu_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, w, b, y, R, K, N, ldy);
    return hipGetLastError();
}

hipError_t small_linear_relu_bwd_launch(const void* x, int x_is_f64, const float* w, const float* b, const float* dy, int lddy,
                                        long R, int K, int N, float* dwb, hipStream_t st) {
    if (R <= 0) return hipSuccess;
    if (K < 1 || K > SL_KMAX || N < 1 || N > SL_NMAX || lddy < N) return hipErrorInvalidValue;
    long blocks = (R + 255) / 256;                                       // one row per thread up to 256 workgroups (one atomic per output each)
    if (blocks > 256) blocks = 256;
    if (x_is_f64) hipLaunchKernelGGL(small_linear_relu_bwd_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double*)x, w, b, dy, lddy, R, K, N, dwb);
    else          hipLaunchKernelGGL(small_linear_relu_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, w, b, dy, lddy, R, K, N, dwb);
    return hipGetLastError();
}

}  // namespace nrm
