// Small kernels around the attention: the un-normalised weighted pool and the training loss.  gfx950 only.
//
// pool (reference models/user_invariant_interest_model.py:86-87):  pooled[b,t,:] = sum_h s[b,t,h] * h[b,h,:]
//   bmm_rows_kernel   out[b,i,:] = sum_j W[b,i,j] * X[b,j,:]      (forward: W = s; backward dh: W = s^T, X = g; batched GEMM on the matrix cores)
//   rowdot_kernel     ds[b,t,h]  = sum_d g[b,t,d] * h[b,h,d]      (batched NT GEMM on the matrix cores)
// loss (reference models/user_model.py:37-43):
//   L = (1-alpha) * BCEmean(softmax_T(out), y) + alpha * BCEmean(softmax_T(out + delta[id]), y)
//   loss_kernel: one wave per impression; softmax max/sum and the row reductions are wave shuffles; writes the
//   loss partial sum and, analytically, dL/dout and dL/ddelta (BCELoss clamps log at -100; its gradient uses
//   (p - y) / max(p (1 - p), 1e-12) like torch.nn.BCELoss).
#include <cstdlib>
#include "common.hpp"
#include "pool_loss.hpp"

namespace nrm {

// out[b,i,:] = sum_j W[b,i,j] X[b,j,:]: a small batched GEMM on the matrix cores.  One wave per (impression, 64-column
// slab, group of up to four 16-row tiles of i): MFMA rows = i (A operand W[i][j], any strides: one dword per lane),
// MFMA columns = the slab's columns in the interleaved order of the backward contractions (lane r16 holds columns
// 4*r16 + tile of the four column tiles), so ONE 16-byte load of X[j][4*r16 .. +3] feeds four MFMAs and a lane's four
// results of one output row are consecutive columns (float4 store).  Rows / columns past the edge read as 0.
// JSPLIT (round 5): with few (impression, slab, row group) tasks -- the reference's default sizes have 256: one wave each, walking
// J = 200 history rows in 50 dependent load -> MFMA rounds, 93 us for 0.1 GFLOP -- the four waves of a workgroup share ONE task,
// each takes a quarter of the reduction range and waves 1-3 hand their partial tiles to wave 0 through the LDS.  Either way the
// operands of round j0 + 4 are requested before the MFMAs of round j0.
template <bool JSPLIT>
__global__ __launch_bounds__(256) void bmm_rows_kernel(const float* __restrict__ W, long wsb, long wsi, long wsj,
                                                       const float* __restrict__ X, long xsb, int ldx,
                                                       float* __restrict__ out, long osb, int ldo,
                                                       int B, int I, int J, int D, int accumulate) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) f32x4 part[JSPLIT ? 3 * 16 * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int nslab = (D + 63) >> 6, nig = (I + 63) >> 6;
    const long task = JSPLIT ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    if (task >= (long)B * nslab * nig) return;                         // (JSPLIT: the whole workgroup leaves together)
    const int slab = (int)(task % nslab);
    const int ig = (int)((task / nslab) % nig);
    const int b = (int)(task / ((long)nslab * nig));
    const int d0 = slab * 64, i0 = ig * 64;
    constexpr unsigned OOB = 0x80000000u;
    const float* Wb = W + b * wsb;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) + b * xsb, 0, ((J - 1) * ldx + D) * 4, 0x00020000);
    const unsigned vx = d0 + 4 * r16 < D ? (unsigned)(q * ldx + d0 + 4 * r16) * 4u : OOB;      // D % 4 == 0
    f32x4 acc[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nit = min(4, (I - i0 + 15) >> 4);                         // uniform: live row tiles of this group
    // reduction range of this wave: everything, or its quarter (in whole rounds of 4 rows)
    const int jq = JSPLIT ? ((J + 15) >> 4) * 4 : J;
    const int j_lo = JSPLIT ? wave * jq : 0, j_hi = min(J, j_lo + jq);
    auto load = [&](int j0, f32x4& x4, float (&a)[4]) {
        const int j = j0 + q;
        x4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vx, j0 * ldx * 4, 0));  // rows >= J: 0
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = i0 + 16 * it + r16;
            a[it] = (it < nit && i < I && j < j_hi) ? Wb[(long)i * wsi + (long)j * wsj] : 0.f;
        }
    };
    f32x4 x_cur, x_nxt;
    float a_cur[4], a_nxt[4];
    if (j_lo < j_hi) load(j_lo, x_cur, a_cur);
    for (int j0 = j_lo; j0 < j_hi; j0 += 4) {
        if (j0 + 4 < j_hi) load(j0 + 4, x_nxt, a_nxt);
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < nit)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[it][jt] = mfma16(a_cur[it], x_cur[jt], acc[it][jt]);
        x_cur = x_nxt;
#pragma unroll
        for (int it = 0; it < 4; ++it) a_cur[it] = a_nxt[it];
    }
    if (JSPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (it < nit)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) part[((wave - 1) * 16 + it * 4 + jt) * 64 + lane] = acc[it][jt];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (it < nit)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) acc[it][jt] += part[(w * 16 + it * 4 + jt) * 64 + lane];
    }
    // lane holds acc[it][jt][e] = out[i0 + 16it + 4q + e][d0 + 4*r16 + jt]
    if (d0 + 4 * r16 < D) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + 16 * it + 4 * q + e;
                if (it < nit && i < I) {
                    float* o = out + b * osb + (long)i * ldo + d0 + 4 * r16;
                    f32x4 v = f32x4{acc[it][0][e], acc[it][1][e], acc[it][2][e], acc[it][3][e]};
                    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
                    *reinterpret_cast<f32x4*>(o) = v;
                }
            }
    }
#endif
}

// ds[b,t,h] = sum_d g[b,t,d] h[b,h,d]: a small batched "NT" GEMM (per impression [T x D] x [H x D]^T) on the matrix cores.
// One wave per (impression, tile of 16 candidates): it walks the feature dimension in chunks of 16 with ONE 16-byte load
// per lane and operand row (lane (r16, q) holds columns 16c + 4q .. +3 of row r16, so MFMA e of a chunk contracts the
// columns {16c + 4q + e}: any split of the reduction index over the MFMAs is fine as long as both operands use the same)
// and keeps up to four 16-row history tiles of accumulators; rows / columns past the edge read as 0 through the buffer
// descriptors.  The chunk after the current one is requested before the current MFMAs.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ g, long gsb, int ldg,
                                                     const float* __restrict__ h, long hsb, int ldh,
                                                     float* __restrict__ ds, int B, int T, int H, int D,
                                                     float* __restrict__ zero_out, int zero_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    // optional: clear a small accumulator of the kernels that follow (dw2 | db2 of the attention backward) -- saves their fill
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_out[i] = 0.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int ntt = (T + 15) >> 4;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= B * ntt) return;
    const int b = tile / ntt, t0 = (tile - b * ntt) * 16;
    const int trows = min(16, T - t0);
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g) + b * gsb + (long)t0 * ldg, 0, ((trows - 1) * ldg + D) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(h) + b * hsb, 0, ((H - 1) * ldh + D) * 4, 0x00020000);
    const unsigned va = r16 < trows ? (unsigned)(r16 * ldg + 4 * q) * 4u : OOB;
    const int nchunk = (D + 15) >> 4;
    for (int h0 = 0; h0 < H; h0 += 64) {
        unsigned vb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = h0 + 16 * j + r16 < H ? (unsigned)((h0 + 16 * j + r16) * ldh + 4 * q) * 4u : OOB;
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // columns >= D of the last chunk (D % 16 != 0): D % 4 == 0, so a lane's four columns are all in or all out
        auto load_chunk = [&](int c, f32x4& a4, f32x4 (&b4)[4]) {
            const bool ok = 16 * c + 4 * q < D;
            a4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? va : OOB, c * 64, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, ok ? vb[j] : OOB, c * 64, 0));
        };
        auto mfma_chunk = [&](const f32x4& a4, const f32x4 (&b4)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma16(a4[e], b4[j][e], acc[j]);
        };
        f32x4 a0, a1, b0[4], b1[4];
        load_chunk(0, a0, b0);
        for (int c = 0; c < nchunk; c += 2) {
            if (c + 1 < nchunk) load_chunk(c + 1, a1, b1);
            mfma_chunk(a0, b0);
            if (c + 1 < nchunk) {
                if (c + 2 < nchunk) load_chunk(c + 2, a0, b0);
                mfma_chunk(a1, b1);
            }
        }
        // lane holds acc[j][e] = ds[t0 + 4q + e][h0 + 16j + r16]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hc = h0 + 16 * j + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + 4 * q + e;
                if (t < T && hc < H) ds[((long)b * T + t) * H + hc] = acc[j][e];
            }
        }
    }
#endif
}

hipError_t bmm_rows_launch(const float* W, long wsb, long wsi, long wsj, const float* X, long xsb, int ldx,
                           float* out, long osb, int ldo, int B, int I, int J, int D, int accumulate, hipStream_t st) {
    if (B <= 0 || I <= 0) return hipSuccess;
    const long tasks = (long)B * ((D + 63) / 64) * ((I + 63) / 64);
    if ((tasks + 3) / 4 > 0x7fffffffL) return hipErrorInvalidValue;
    // few tasks (one wave each would leave most SIMDs empty) and a reduction long enough to cut in four: one workgroup per task.
    // Measured (graph replay, same box): reference default sizes, 256 tasks, J = 200: 27.8 -> 14.8 us per launch; C2, 2048 tasks,
    // J = 32: 18.6 -> 22.4 us -- so only up to 1024 tasks.  NRM_POOL_JSPLIT=0|1 forces either form.
    const char* env = getenv("NRM_POOL_JSPLIT");
    const bool jsplit = env ? env[0] == '1' : (tasks <= 1024 && J >= 32);
    if (jsplit)
        hipLaunchKernelGGL(bmm_rows_kernel<true>, dim3((unsigned)tasks), dim3(256), 0, st,
                           W, wsb, wsi, wsj, X, xsb, ldx, out, osb, ldo, B, I, J, D, accumulate);
    else
        hipLaunchKernelGGL(bmm_rows_kernel<false>, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, st,
                           W, wsb, wsi, wsj, X, xsb, ldx, out, osb, ldo, B, I, J, D, accumulate);
    return hipGetLastError();
}

hipError_t rowdot_launch(const float* g, long gsb, int ldg, const float* h, long hsb, int ldh, float* ds,
                         int B, int T, int H, int D, float* zero_out, int zero_n, hipStream_t st) {
    if (B <= 0) return zero_n > 0 ? hipMemsetAsync(zero_out, 0, (size_t)zero_n * sizeof(float), st) : hipSuccess;
    const long tiles = (long)B * ((T + 15) / 16);
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, g, gsb, ldg, h, hsb, ldh, ds, B, T, H, D, zero_out, zero_n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// loss: blockDim = 256 (4 waves = 4 impressions); lane l handles candidates l, l+64, ... (T <= 256)
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// out / dout: element (b, t) at [(b*T + t) * stride] -- stride 4 is the single column of a zero-padded [B*T, 4] matrix (what the
// GEMM that produces the logits writes and the GEMM that consumes their gradient reads): dout then gets whole float4s (g, 0, 0, 0).
template <typename LT>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, int so, const LT* __restrict__ label,
                                                   const long* __restrict__ uid, const float* __restrict__ delta,
                                                   long n_delta, float alpha, int B, int T, float* __restrict__ loss_sum,
                                                   float* __restrict__ dout, int sd, float* __restrict__ ddelta, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    // delta[id] as torch indexes it (models/user_model.py:40): a negative id counts from the end; anything still
    // outside [0, n_delta) raises IndexError in the reference -- here it is clamped and flagged (never an OOB access)
    long id = uid[b];
    if (id < 0) id += n_delta;
    if (id < 0 || id >= n_delta) {
        if (lane == 0) *err = 1;
        id = id < 0 ? 0 : n_delta - 1;
    }
    const float dl = delta[id];
    const float inv = 1.0f / ((float)B * (float)T);
    float o[4], y[4];
    float mx = -3.0e38f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int t = lane + 64 * v;
        o[v] = t < T ? out[((long)b * T + t) * so] : -3.0e38f;
        y[v] = t < T ? (float)label[(long)b * T + t] : 0.f;
        mx = fmaxf(mx, o[v]);
    }
    mx = wave_max64(mx);
    float total = 0.f, gsum_shift = 0.f;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    // the two terms differ by the per-row shift only; softmax(out + c) is computed as written (shift inside exp)
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        const float shift = term ? dl : 0.f;
        const float wt = term ? alpha : 1.0f - alpha;
        float e[4], se = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) { e[v] = (lane + 64 * v < T) ? __expf((o[v] + shift) - (mx + shift)) : 0.f; se += e[v]; }
        se = wave_sum64(se);
        const float rse = 1.0f / se;
        float dp[4], pdp = 0.f, lsum = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool ok = lane + 64 * v < T;
            const float p = e[v] * rse;
            const float lp = fmaxf(__logf(p), -100.f), l1p = fmaxf(__logf(1.0f - p), -100.f);
            lsum += ok ? -(y[v] * lp + (1.0f - y[v]) * l1p) : 0.f;
            dp[v] = ok ? (p - y[v]) / fmaxf(p * (1.0f - p), 1e-12f) : 0.f;          // d BCE / dp  (torch BCELoss backward)
            pdp += p * dp[v];
            e[v] = p;
        }
        pdp = wave_sum64(pdp);
        lsum = wave_sum64(lsum);
        total += wt * lsum;
        float gs = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float gg = wt * inv * e[v] * (dp[v] - pdp);                       // softmax backward
            g[v] += gg;
            gs += gg;
        }
        if (term) gsum_shift = wave_sum64(gs);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int t = lane + 64 * v;
        if (t < T) {
            if (sd == 4) *reinterpret_cast<f32x4*>(dout + ((long)b * T + t) * 4) = f32x4{g[v], 0.f, 0.f, 0.f};
            else dout[((long)b * T + t) * sd] = g[v];
        }
    }
    if (lane == 0) {
        atomicAdd(loss_sum, total * inv);
        atomicAdd(ddelta + id, gsum_shift);
    }
}

// More than 256 candidates per impression (the reference takes any T: models/user_model.py:37-43 is shape-agnostic): the same
// arithmetic with the row re-read from memory in every pass instead of held in four registers per lane -- a lane still owns
// t = lane, lane + 64, ... and sums them in that order, so for T <= 256 both kernels produce bit-identical results.
template <typename LT>
__global__ __launch_bounds__(256) void loss_long_kernel(const float* __restrict__ out, int so, const LT* __restrict__ label,
                                                        const long* __restrict__ uid, const float* __restrict__ delta,
                                                        long n_delta, float alpha, int B, int T, float* __restrict__ loss_sum,
                                                        float* __restrict__ dout, int sd, float* __restrict__ ddelta, int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    long id = uid[b];
    if (id < 0) id += n_delta;
    if (id < 0 || id >= n_delta) {
        if (lane == 0) *err = 1;
        id = id < 0 ? 0 : n_delta - 1;
    }
    const float dl = delta[id];
    const float inv = 1.0f / ((float)B * (float)T);
    const float* ob = out + (long)b * T * so;
    const LT* yb = label + (long)b * T;
    float* gb = dout + (long)b * T * sd;
    float mx = -3.0e38f;
    for (int t = lane; t < T; t += 64) mx = fmaxf(mx, ob[(long)t * so]);
    mx = wave_max64(mx);
    float total = 0.f, gsum_shift = 0.f;
    for (int term = 0; term < 2; ++term) {
        const float shift = term ? dl : 0.f;
        const float wt = term ? alpha : 1.0f - alpha;
        float se = 0.f;
        for (int t = lane; t < T; t += 64) se += __expf((ob[(long)t * so] + shift) - (mx + shift));
        se = wave_sum64(se);
        const float rse = 1.0f / se;
        float pdp = 0.f, lsum = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float p = __expf((ob[(long)t * so] + shift) - (mx + shift)) * rse;
            const float y = (float)yb[t];
            const float lp = fmaxf(__logf(p), -100.f), l1p = fmaxf(__logf(1.0f - p), -100.f);
            lsum += -(y * lp + (1.0f - y) * l1p);
            pdp += p * ((p - y) / fmaxf(p * (1.0f - p), 1e-12f));
        }
        pdp = wave_sum64(pdp);
        lsum = wave_sum64(lsum);
        total += wt * lsum;
        float gs = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float p = __expf((ob[(long)t * so] + shift) - (mx + shift)) * rse;
            const float y = (float)yb[t];
            const float dp = (p - y) / fmaxf(p * (1.0f - p), 1e-12f);
            const float gg = wt * inv * p * (dp - pdp);
            gs += gg;
            // this lane wrote element t in term 0 and is the only one to touch it: plain read-modify-write in term 1
            if (sd == 4) *reinterpret_cast<f32x4*>(gb + (long)t * 4) = f32x4{term ? gb[(long)t * 4] + gg : gg, 0.f, 0.f, 0.f};
            else gb[(long)t * sd] = term ? gb[(long)t * sd] + gg : gg;
        }
        if (term) gsum_shift = wave_sum64(gs);
    }
    if (lane == 0) {
        atomicAdd(loss_sum, total * inv);
        atomicAdd(ddelta + id, gsum_shift);
    }
}

hipError_t loss_launch(const float* out, int out_stride, const void* label, int label_is_f64, const long* uid, const float* delta,
                       long n_delta, float alpha, int B, int T, float* loss_sum, float* dout, int dout_stride, float* ddelta, int* err,
                       hipStream_t st) {
    if (B <= 0) return hipSuccess;
    const char* force_long = getenv("NRM_LOSS_LONG");                 // tests: =1 runs the re-reading kernel for any T
    if (T > 256 || (force_long && force_long[0] == '1')) {
        if (label_is_f64)
            hipLaunchKernelGGL(loss_long_kernel<double>, dim3((B + 3) / 4), dim3(256), 0, st, out, out_stride, (const double*)label, uid,
                               delta, n_delta, alpha, B, T, loss_sum, dout, dout_stride, ddelta, err);
        else
            hipLaunchKernelGGL(loss_long_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, st, out, out_stride, (const float*)label, uid,
                               delta, n_delta, alpha, B, T, loss_sum, dout, dout_stride, ddelta, err);
        return hipGetLastError();
    }
    if (label_is_f64)
        hipLaunchKernelGGL(loss_kernel<double>, dim3((B + 3) / 4), dim3(256), 0, st, out, out_stride, (const double*)label, uid, delta,
                           n_delta, alpha, B, T, loss_sum, dout, dout_stride, ddelta, err);
    else
        hipLaunchKernelGGL(loss_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, st, out, out_stride, (const float*)label, uid, delta,
                           n_delta, alpha, B, T, loss_sum, dout, dout_stride, ddelta, err);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adam over one flat fp32 buffer (reference train.py:48: Adam(lr, weight_decay=1e-5), L2 folded into the
// gradient, bias-corrected, eps added after the sqrt).  Also zeroes the gradient (optimizer.zero_grad()).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2_sqrt, int zero_grad) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = fmaf(wd, pv[e], gv[e]);
            mv[e] = fmaf(b1, mv[e], (1.0f - b1) * gg);
            vv[e] = fmaf(b2, vv[e], (1.0f - b2) * gg * gg);
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] -= (lr / bc1) * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// state = {step, 1 - b1^step, sqrt(1 - b2^step)} kept on the device so that a captured hipGraph replays
// with the right bias correction (no host-side step counter inside the graph)
__global__ void adam_prep_kernel(float* __restrict__ state, float b1, float b2) {
    const float step = state[0] + 1.0f;
    state[0] = step;
    state[1] = 1.0f - powf(b1, step);
    state[2] = sqrtf(1.0f - powf(b2, step));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr, float b1, float b2,
                                                       float eps, float wd, const float* __restrict__ state, int zero_grad) {
    const float bc1 = state[1], bc2_sqrt = state[2];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = fmaf(wd, pv[e], gv[e]);
            mv[e] = fmaf(b1, mv[e], (1.0f - b1) * gg);
            vv[e] = fmaf(b2, vv[e], (1.0f - b2) * gg * gg);
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] -= (lr / bc1) * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Gradients of up to 128 parameter tensors -> their slots of ONE flat buffer in a single launch (the table travels in the
// kernel arguments: no device-side table to keep coherent, and a hipGraph capture bakes the pointers of its own pool).
// blockIdx.y = tensor; a null source zero-fills its slot (a parameter that received no gradient).  Replaces one
// AccumulateGrad add_ launch per parameter (38 per step) with one launch.
__global__ __launch_bounds__(256) void gather_flat_kernel(const GatherTable tab, float* __restrict__ flat) {
    const int i = blockIdx.y;
    const float* __restrict__ src = tab.src[i];
    float* __restrict__ dst = flat + tab.off[i];
    const long n = tab.cnt[i];
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long stride = (long)gridDim.x * blockDim.x;
    if (vec) {
        const long n4 = n >> 2;
        for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += stride)
            reinterpret_cast<f32x4*>(dst)[j] = src ? reinterpret_cast<const f32x4*>(src)[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = src ? src[j] : 0.f;
    }
}

hipError_t gather_flat_launch(const GatherTable& tab, int n, long max_count, float* flat, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long bx = (max_count / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(gather_flat_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, tab, flat);
    return hipGetLastError();
}

static unsigned adam_blocks(long n) {
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

hipError_t adam_launch(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                       float wd, int step, int zero_grad, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(adam_blocks(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, zero_grad);
    return hipGetLastError();
}

hipError_t adam_dev_launch(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                           float wd, float* state, int zero_grad, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, st, state, b1, b2);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(adam_blocks(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, state, zero_grad);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// per-impression ROC-AUC and top-1 hit (reference train.py:77-80, verify.py:25-36, tool/evaluation.py:3-5):
// AUC = (#{pos i, neg j: s_i > s_j} + 0.5 #{s_i == s_j}) / (n_pos n_neg)  -- the Mann-Whitney form sklearn's
// roc_auc_score evaluates for binary labels.  One wave per impression, lane = candidate (T <= 256); only the
// first len[b] candidates count (trailing padding).  auc = -1 where only one class is present (sklearn raises).
__global__ __launch_bounds__(256) void row_auc_kernel(const float* __restrict__ score, const float* __restrict__ label,
                                                      const int* __restrict__ len, int B, int T,
                                                      float* __restrict__ auc, int* __restrict__ top1) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int n = len ? min(len[b], T) : T;
    const float* s = score + (long)b * T;
    const float* y = label + (long)b * T;
    float gt = 0.f, eq = 0.f, npos = 0.f;
    float best_s = -3.0e38f, best_y = -3.0e38f;
    int arg_s = 0x7fffffff, arg_y = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float si = s[i], yi = y[i];
        if (si > best_s) { best_s = si; arg_s = i; }
        if (yi > best_y) { best_y = yi; arg_y = i; }
        if (yi > 0.5f) {
            npos += 1.f;
            for (int j = 0; j < n; ++j) {
                if (y[j] <= 0.5f) { gt += si > s[j] ? 1.f : 0.f; eq += si == s[j] ? 1.f : 0.f; }
            }
        }
    }
    gt = wave_sum64(gt); eq = wave_sum64(eq); npos = wave_sum64(npos);
    // argmax with first-index tie-break (numpy.argmax)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float os = __shfl_xor(best_s, o), oy = __shfl_xor(best_y, o);
        const int oas = __shfl_xor(arg_s, o), oay = __shfl_xor(arg_y, o);
        if (os > best_s || (os == best_s && oas < arg_s)) { best_s = os; arg_s = oas; }
        if (oy > best_y || (oy == best_y && oay < arg_y)) { best_y = oy; arg_y = oay; }
    }
    if (lane == 0) {
        const float nneg = (float)n - npos;
        auc[b] = (npos > 0.f && nneg > 0.f) ? (gt + 0.5f * eq) / (npos * nneg) : -1.0f;
        top1[b] = (n > 0 && arg_s == arg_y) ? 1 : 0;
    }
}

hipError_t row_auc_launch(const float* score, const float* label, const int* len, int B, int T, float* auc, int* top1,
                          hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(row_auc_kernel, dim3((B + 3) / 4), dim3(256), 0, st, score, label, len, B, T, auc, top1);
    return hipGetLastError();
}

}  // namespace nrm
