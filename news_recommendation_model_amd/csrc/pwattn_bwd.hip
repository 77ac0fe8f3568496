// Pointwise history attention, backward.  gfx950 / MI355X only.
//
// Forward (pwattn_fwd.hip):  z[m,k] = u[b,h,k] + v[b,t,k] + sum_d W_p[k,d] t[b,t,d] h[b,h,d],
//                            s[m]   = b2 + sum_k w2[k] gelu(z[m,k]),      m = (b,t,h)
// Given ds[m] = dL/ds[m] and the saved z:
//   (1) bwd_dz_kernel      dz[m,k] = ds[m] * w2[k] * gelu'(z[m,k])   (in place over z)
//                          dw2[k] += sum_m ds[m] * gelu(z[m,k]);  du = sum_t dz;  dv = sum_h dz
//   (2) bwd_e_kernel       for every group g (one (b,t) in pass 1, one (b,h) in pass 2):
//                              E_g[k,d] = sum_r X_g[r,k] * Y_g[r,d]
//                          pass 1: X_g = dz[b,t,:,:] (r = h), Y_g = h[b]    -> dt[b,t,d] = sum_k W_p[k,d] E_g[k,d]
//                                                                             dW_p[k,d] += E_g[k,d] * t[b,t,d]
//                          pass 2: X_g = dz[b,:,h,:] (r = t), Y_g = t[b]    -> dh[b,h,d] = sum_k W_p[k,d] E_g[k,d]
//      Both passes together are exactly the 2x-forward FLOPs of a GEMM backward; dW_p is accumulated in
//      registers across the groups a wave walks (one partial slab per split, summed after), and an output
//      row receives one contiguous float-atomic add per k-range (5 at D=400).
//   The side-projection gradients (from du, dv) are plain GEMMs done by the caller.
#include <cstdlib>
#include <type_traits>
#ifndef NRM_PRIO
#define NRM_PRIO 0        // tuning: s_setprio level of the MFMA phases (0: none)
#endif
#ifndef NRM_PIPE_SGB
#define NRM_PIPE_SGB 1
#endif
// timing diagnostics only (scripts/_diag/build_diag.sh): results are WRONG with any of them set
#ifndef NRM_EPI_AHEAD
#define NRM_EPI_AHEAD 1       // dt/dW pass: W_p^T LDS reads two tiles ahead in the epilogue (0: one read, one wait, per tile)
#endif
#ifndef NRM_DIAG_GELU_AT_LOAD
#define NRM_DIAG_GELU_AT_LOAD 0   // timing only: the serial contraction kernels evaluate gelu'(x) on every X operand they load (what
#endif                            // consuming z directly -- no dz pass -- would cost them)
#ifndef NRM_DIAG_NOEPI
#define NRM_DIAG_NOEPI 0      // contraction kernels without the per-group epilogue
#endif
#ifndef NRM_DIAG_NOATOM
#define NRM_DIAG_NOATOM 0     // contraction kernels without the float atomics of the out-row flush
#endif
#ifndef NRM_DIAG_NOLOAD
#define NRM_DIAG_NOLOAD 0     // contraction kernels that never reload MFMA operands (1: both; serial kernel also 2: X only, 3: Y only, 4: reload step 0 of the group every time)
#endif
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

// ---------------------------------------------------------------------------------------------
// (1) one pass over z: dz in place, plus every reduction of dz that does not need a contraction
//       du[b,h,k] = sum_t dz[b,t,h,k]     dv[b,t,k] = sum_h dz[b,t,h,k]     dw2[k] += sum ds*gelu(z)
//     HBM-bound (read z, write dz: 4.9 GB at C3; a plain in-place torch mul_ over the same bytes takes 0.84 ms).
//     workgroup = (one impression b) x (one slab of <= 128 columns, all slabs equally wide: 4 x 100 at D = 400); its
//     256 threads are nx float4 columns x ny rows and sweep the impression's rows in memory order (ny consecutive
//     history rows at a time, candidate after candidate; the sibling slabs sweep the same rows at the same time, so
//     DRAM sees whole rows -- a variant in which every thread streamed its own candidate ran at 1.6 TB/s).
//     All loads of one candidate are requested before any arithmetic.  dv is a register sum over the inner h loop
//     plus one LDS exchange per candidate (double-buffered: one barrier); du lives in LDS [H][nx] (every cell is owned
//     by exactly one thread: h % ny == ty), so z/dz are touched exactly once.
constexpr int DZ_MAXIT = 5;      // history rows per thread and candidate held in registers (H <= DZ_MAXIT * ny per sweep)

__global__ __launch_bounds__(256) void bwd_dz_kernel(float* __restrict__ z, const float* __restrict__ ds,
                                                     const float* __restrict__ w2, float* __restrict__ dw2,
                                                     float* __restrict__ db2, float* __restrict__ du, float* __restrict__ dv,
                                                     int T, int Hall, int D, int slab_cols, int hchunk, int fmt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [H][nx] du slab | 2 x [256] float4 exchange
    const int nx = slab_cols >> 2, ny = 256 / nx;
    // blockIdx.z = chunk of history rows [h0, h0 + H): one chunk unless the du slab of all rows would not fit the LDS
    // (H > 256); with several chunks dv is accumulated across them with float atomics (the launcher zeroes it)
    const int h0 = blockIdx.z * hchunk;
    const int H = min(hchunk, Hall - h0);
    const bool dv_atomic = gridDim.z > 1;
    f32x4* du_l = reinterpret_cast<f32x4*>(sm);
    f32x4* red = du_l + H * nx;
    const int tid = threadIdx.x;
    const int ty = tid / nx, tx = tid - ty * nx;
    // XCD-aware order: the sibling slabs of one impression share the 128-B lines their 400-B row segments straddle (D = 400:
    // boundaries at 400-B multiples), and hardware deals consecutive block ids round-robin over the 8 XCDs.  Consecutive
    // LOGICAL ids (the slabs of one b) go to one XCD, so a shared line is fetched / written back once, not once per L2.
    const int nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int b = logical / gridDim.x, slab = logical - b * gridDim.x;
    const int col = slab * slab_cols + 4 * tx;
    const bool cok = ty < ny && col < D;                             // D % 4 == 0
    const f32x4 w = cok ? *reinterpret_cast<const f32x4*>(w2 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (ty < ny)
        for (int h = ty; h < H; h += ny) du_l[h * nx + tx] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aw = f32x4{0.f, 0.f, 0.f, 0.f};
    float ab = 0.f;                                                  // db2 = sum ds: counted once, by column 0 of slab 0
    const bool sum_ds = db2 != nullptr && slab == 0 && tx == 0;
    // rows hb + u*ny (u < DZ_MAXIT) of candidate t: request everything, then compute -- and the requests of candidate
    // t+1 go out before the arithmetic of candidate t (two register buffers, the t loop is unrolled by two)
    auto load_rows = [&](int t, int hb, f32x4 (&zz)[DZ_MAXIT], float (&g)[DZ_MAXIT]) {
        const long row0 = ((long)b * T + t) * Hall + h0;
#pragma unroll
        for (int u = 0; u < DZ_MAXIT; ++u) {
            const int h = hb + u * ny;
            const bool ok = cok && t < T && h < H;
            g[u] = ok ? ds[row0 + h] : 0.f;
            zz[u] = ok ? *reinterpret_cast<const f32x4*>(z + (row0 + h) * D + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto compute_rows = [&](int t, int hb, const f32x4 (&zz)[DZ_MAXIT], const float (&g)[DZ_MAXIT], f32x4& av) {
        const long row0 = ((long)b * T + t) * Hall + h0;
#pragma unroll
        for (int u = 0; u < DZ_MAXIT; ++u) {
            const int h = hb + u * ny;
            if (cok && h < H) {
                f32x4 dz;
                if (sum_ds) ab += g[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const GeluParts gp = gelu_parts(zz[u][e]);
                    aw[e] = fmaf(g[u] * zz[u][e], gp.cdf, aw[e]);                          // ds * gelu(z)
                    dz[e] = g[u] * w[e] * fmaf(zz[u][e] * 0.39894228040143267794f, gp.e, gp.cdf);
                }
                if (fmt) {
                    // NRM_DZ_HL4: the four values as 4 bf16 hi + 4 bf16 lo (lo = rounding remainder) in the same 16 bytes -- the
                    // MFMA-ready operand of the bf16 contraction kernels (pwattn_bwd_rw.hip reads it without any conversion)
                    unsigned short hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 hb = (__bf16)dz[e];
                        const __bf16 lb = (__bf16)(dz[e] - (float)hb);
                        hi[e] = __builtin_bit_cast(unsigned short, hb);
                        lo[e] = __builtin_bit_cast(unsigned short, lb);
                    }
                    *reinterpret_cast<u32x4*>(z + (row0 + h) * D + col) =
                        u32x4{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16),
                              (unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
                } else {
                    *reinterpret_cast<f32x4*>(z + (row0 + h) * D + col) = dz;
                }
                av += dz;
                du_l[h * nx + tx] += dz;
            }
        }
    };
    auto finish = [&](int t, const f32x4& av) {
        f32x4* ex = red + (t & 1) * 256;                             // double-buffered: one barrier per candidate
        ex[tid] = av;
        __syncthreads();
        if (ty == 0 && cok) {
            f32x4 sum = ex[tx];
            for (int y = 1; y < ny; ++y) sum += ex[y * nx + tx];
            float* dvp = dv + ((long)b * T + t) * D + col;
            if (!dv_atomic) *reinterpret_cast<f32x4*>(dvp) = sum;
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dvp + e, sum[e]);
        }
    };
    const bool one_sweep = H <= DZ_MAXIT * ny;                       // the usual case: a candidate's rows fit one sweep
    f32x4 zA[DZ_MAXIT], zB[DZ_MAXIT];
    float gA[DZ_MAXIT], gB[DZ_MAXIT];
    if (one_sweep) {
        load_rows(0, ty, zA, gA);
        for (int t = 0; t < T; t += 2) {
            load_rows(t + 1, ty, zB, gB);
            f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f};
            compute_rows(t, ty, zA, gA, av);
            finish(t, av);
            if (t + 1 < T) {
                load_rows(t + 2, ty, zA, gA);
                f32x4 av1 = f32x4{0.f, 0.f, 0.f, 0.f};
                compute_rows(t + 1, ty, zB, gB, av1);
                finish(t + 1, av1);
            }
        }
    } else {
        for (int t = 0; t < T; ++t) {
            f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int hb = ty; hb < H; hb += DZ_MAXIT * ny) {
                load_rows(t, hb, zA, gA);
                compute_rows(t, hb, zA, gA, av);
            }
            finish(t, av);
        }
    }
    if (cok)
        for (int h = ty; h < H; h += ny) *reinterpret_cast<f32x4*>(du + ((long)b * Hall + h0 + h) * D + col) = du_l[h * nx + tx];
    __syncthreads();
    red[tid] = aw;
    __syncthreads();
    if (ty == 0 && cok) {
        f32x4 sum = red[tx];
        for (int y = 1; y < ny; ++y) sum += red[y * nx + tx];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dw2 + col + e, sum[e]);
    }
    if (db2 != nullptr && slab == 0) {                               // uniform per workgroup
        __syncthreads();
        float* redf = reinterpret_cast<float*>(red);
        if (tx == 0) redf[ty] = ab;
        __syncthreads();
        if (tid == 0) {
            float sum = 0.f;
            for (int y = 0; y < ny; ++y) sum += redf[y];
            atomicAdd(db2, sum);
        }
    }
}

// Full-row form of the same pass (round 4).  The slab form above cuts every 1600-byte row (D = 400) into four 400-byte
// segments that straddle 128-byte lines and belong to four workgroups: 4.25 TB/s of algorithmic bytes at C3, against 5.1-5.2
// for D = 384 / 512 with the same kernel.  Here ONE workgroup of 512 threads owns all D columns of an impression: a (b,t)
// block of H rows is one contiguous, 128-byte-aligned range (H * D * 4 = 80 000 bytes = 625 whole lines at C3), every wave
// access is 1 KiB contiguous, and no line is shared between workgroups.  nx = D/4 float4 columns x ny = 512/nx row slots; a
// thread owns rows h = ty + ny * k (k < 2 RC) of every candidate, so du (sum over t) is a REGISTER accumulator per owned
// row -- no [H][nx] LDS slab (80 KB at C3, which had kept this form at one workgroup per CU when round 3 sized it) -- and
// dv (sum over h) is the register sum over the thread's rows plus one LDS exchange per candidate, as above.  A candidate is
// processed in two chunks of RC rows per thread: while chunk c is computed the loads of the next chunk are in flight
// (buffers A / B), which keeps the kernel at <= 128 VGPRs = two workgroups (16 waves) per CU.
template <int THREADS, int RC, int NCH>
__global__ __launch_bounds__(THREADS, 4) void bwd_dz_rows_kernel(float* __restrict__ z, const float* __restrict__ ds,
                                                             const float* __restrict__ w2, float* __restrict__ dw2,
                                                             float* __restrict__ db2, float* __restrict__ du, float* __restrict__ dv,
                                                             int T, int H, int D, int fmt) {
    static_assert(NCH % 3 == 0, "three operand buffers rotate once or twice per candidate");
    __shared__ __attribute__((aligned(16))) f32x4 red[2 * THREADS];
    const int nx = D >> 2, ny = THREADS / nx;
    const int tid = threadIdx.x;
    const int ty = tid / nx, tx = tid - ty * nx;
    const int b = blockIdx.x;
    const int col = 4 * tx;
    const bool act = ty < ny;
    const f32x4 w = act ? *reinterpret_cast<const f32x4*>(w2 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dur[NCH * RC];
#pragma unroll
    for (int k = 0; k < NCH * RC; ++k) dur[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aw = f32x4{0.f, 0.f, 0.f, 0.f};
    float ab = 0.f;
    const bool sum_ds = db2 != nullptr && tx == 0;
    // one buffer descriptor per workgroup (base = the impression's first row): 32-bit offsets, rows past T*H read 0
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(z + (size_t)b * T * H * D, 0, (unsigned)((size_t)T * H * D * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ds) + (size_t)b * T * H, 0, (unsigned)(T * H * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    f32x4 zb[3][RC];
    float gb[3][RC];
    auto load_rows = [&](int t, int c, f32x4 (&zz)[RC], float (&g)[RC]) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int h = ty + ny * (c * RC + u);
            const bool ok = act && t < T && h < H;
            const unsigned row = (unsigned)(t * H + h);
            g[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, ok ? row * 4u : OOB, 0, 0));
            zz[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, ok ? (row * (unsigned)D + (unsigned)col) * 4u : OOB, 0, 0));
        }
    };
    auto compute_rows = [&](int t, int c, const f32x4 (&zz)[RC], const float (&g)[RC], f32x4 (&acc)[RC], f32x4& av) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int h = ty + ny * (c * RC + u);
            if (act && h < H) {
                f32x4 dz;
                if (sum_ds) ab += g[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const GeluParts gp = gelu_parts(zz[u][e]);
                    aw[e] = fmaf(g[u] * zz[u][e], gp.cdf, aw[e]);                          // ds * gelu(z)
                    dz[e] = g[u] * w[e] * fmaf(zz[u][e] * 0.39894228040143267794f, gp.e, gp.cdf);
                }
                const unsigned off = ((unsigned)(t * H + h) * (unsigned)D + (unsigned)col) * 4u;
                if (fmt) {                                                                  // NRM_DZ_HL4 (see the slab form)
                    unsigned short hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 hb = (__bf16)dz[e];
                        const __bf16 lb = (__bf16)(dz[e] - (float)hb);
                        hi[e] = __builtin_bit_cast(unsigned short, hb);
                        lo[e] = __builtin_bit_cast(unsigned short, lb);
                    }
                    // (both stores through the guard of common.hpp: their soffset is a literal 0 today, which hipcc pads by itself --
                    // ADVICE r4 -- but the library's rule is that no 16-byte buffer store relies on that)
                    store_b128_guarded(
                        u32x4{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16),
                              (unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)}, rz, off, 0);
                } else {
                    store_b128_guarded(__builtin_bit_cast(u32x4, dz), rz, off, 0);
                }
                av += dz;
                acc[u] += dz;
            }
        }
    };
    auto finish = [&](int t, const f32x4& av) {
        f32x4* ex = red + (t & 1) * THREADS;                         // double-buffered: one barrier per candidate
        ex[tid] = av;
        __syncthreads();
        if (ty == 0) {
            f32x4 sum = ex[tx];
            for (int y = 1; y < ny; ++y) sum += ex[y * nx + tx];
            *reinterpret_cast<f32x4*>(dv + ((long)b * T + t) * D + col) = sum;
        }
    };
    // loads run two chunks ahead of the arithmetic: chunk (t, c) lives in buffer c % 3 (NCH % 3 == 0: the same buffer for the
    // same c of every candidate, so every index below is a compile-time constant after unrolling)
    load_rows(0, 0, zb[0], gb[0]);
    load_rows(0, 1, zb[1], gb[1]);
    f32x4 (*durc)[RC] = reinterpret_cast<f32x4(*)[RC]>(dur);
    for (int t = 0; t < T; ++t) {
        f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int cn = (c + 2) % NCH, tn = t + (c + 2) / NCH;
            load_rows(tn, cn, zb[(c + 2) % 3], gb[(c + 2) % 3]);     // (masked past the last candidate and past the last row)
            compute_rows(t, c, zb[c % 3], gb[c % 3], durc[c], av);
        }
        finish(t, av);
    }
    if (act)
#pragma unroll
        for (int k = 0; k < NCH * RC; ++k) {
            const int h = ty + ny * k;
            if (h < H) *reinterpret_cast<f32x4*>(du + ((long)b * H + h) * D + col) = dur[k];
        }
    __syncthreads();
    red[tid] = aw;
    __syncthreads();
    if (ty == 0) {
        f32x4 sum = red[tx];
        for (int y = 1; y < ny; ++y) sum += red[y * nx + tx];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dw2 + col + e, sum[e]);
    }
    if (db2 != nullptr) {
        __syncthreads();
        float* redf = reinterpret_cast<float*>(red);
        if (tx == 0 && act) redf[ty] = ab;
        __syncthreads();
        if (tid == 0) {
            float sum = 0.f;
            for (int y = 0; y < ny; ++y) sum += redf[y];
            atomicAdd(db2, sum);
        }
    }
}

// The full-row form applies when the workgroup's threads are well used (nx * ny of them own columns), a thread's rows fit 6
// register accumulators (512 threads, two workgroups per CU; else 1024 threads, one per CU: the 12-row form of 512 threads
// spills at 128 VGPRs), one impression's z block fits 32-bit byte offsets, and the batch is not tiny;
// NRM_DZ_ROWS=0 / 1 forces the slab / the full-row form.  Returns the workgroup size (0: slab form) and the rows per thread.
static int dz_rows_threads(int B, int T, int H, int D, int& rows) {
    const char* env = getenv("NRM_DZ_ROWS");                         // (read per launch: tests switch forms inside one process)
    const int forced = env ? (env[0] == '0' ? 0 : 1) : -1;
    if (forced == 0 || D % 4 || D > 2048 || (size_t)T * H * D * 4 >= (1ull << 31)) return 0;
    const int nx = D >> 2;
    for (int threads = 512; threads <= 1024; threads *= 2) {
        const int ny = threads / nx;
        if (ny < 1) continue;
        rows = (H + ny - 1) / ny;
        if (rows > 6) continue;
        if (forced != 1 && (nx * ny * 20 < threads * 17 || B < 32)) continue;    // >= 85 % of the lanes own a column (measured faster than the slab form from B = 80 up: 0.061 vs 0.064 ms at the reference's test batch, 0.083 vs 0.094 at its training batch)
        return threads;
    }
    return 0;
}

hipError_t bwd_dz_launch(float* z, const float* ds, const float* w2, float* dw2, float* db2, float* du, float* dv,
                         int B, int T, int H, int D, int dz_format, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    int rows = 0;
    if (const int threads = dz_rows_threads(B, T, H, D, rows)) {
        auto kern = threads == 512 ? (rows <= 3 ? bwd_dz_rows_kernel<512, 1, 3> : bwd_dz_rows_kernel<512, 2, 3>)
                                   : (rows <= 3 ? bwd_dz_rows_kernel<1024, 1, 3> : bwd_dz_rows_kernel<1024, 2, 3>);
        hipLaunchKernelGGL(kern, dim3(B), dim3(threads), 0, st, z, ds, w2, dw2, db2, du, dv, T, H, D, dz_format);
        return hipGetLastError();
    }
    int nslab = (D + 127) / 128;
    int slab_cols = ((D + nslab - 1) / nslab + 3) / 4 * 4;           // equal slabs (D = 400: 4 x 100 columns)
    // Long histories: a candidate's rows fit ONE sweep of the workgroup (the path that requests the next candidate's rows before
    // the arithmetic of the current one) only if H <= DZ_MAXIT * (256 / (slab_cols / 4)); narrower slabs give more row slots
    // (reference default sizes, H = 200, D = 64: one 64-column slab has 80, four 16-column slabs 320 -- and 1024 workgroups
    // instead of 256).  The sibling slabs share cache lines; the XCD-aware block order keeps them on one L2.
    static const bool narrow = [] { const char* e = getenv("NRM_DZ_NARROW"); return !(e && e[0] == '0'); }();
    while (narrow && slab_cols > 16 && H > DZ_MAXIT * (256 / (slab_cols / 4)) && H <= 256) {
        nslab *= 2;
        slab_cols = ((D + nslab - 1) / nslab + 3) / 4 * 4;
        if (slab_cols < 16) slab_cols = 16;
    }
    const int hchunk = H <= 256 ? H : 256;                           // history rows whose du slab shares the LDS
    const int nhc = (H + hchunk - 1) / hchunk;
    const size_t shm = ((size_t)hchunk * (slab_cols / 4) + 2 * 256) * sizeof(f32x4);
    if (shm > 160 * 1024 || nhc > 65535) return hipErrorInvalidValue;
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)bwd_dz_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    if (nhc > 1) {                                                    // dv is summed over the history chunks
        hipError_t e = hipMemsetAsync(dv, 0, (size_t)B * T * D * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bwd_dz_kernel, dim3((D + slab_cols - 1) / slab_cols, B, nhc), dim3(256), shm, st, z, ds, w2, dw2, db2, du, dv,
                       T, H, D, slab_cols, hchunk, dz_format);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// (2) grouped contraction E_g = X_g^T Y_g with fused epilogue.
// MFMA orientation: row index i = k (A operand = X = dz), column index j = d (B operand = Y), so a lane
// holds 4 consecutive k of one d: the k-reduction of out[g,d] = sum_k W_p[k,d] E_g[k,d] is in-lane plus two
// cross-lane steps, W_p^T comes as float4 along k from an LDS tile, and dW_p is kept transposed ([d][k]).
//
// Work decomposition: the (KT*16 k) x (DT*16 d) wave tiles of E / dW_p form an nkw x ndcol grid; the G
// groups are cut into `nsplit` splits.  A workgroup = 4 waves that share ONE wave tile (so its W_p^T tile
// lives once in LDS) and take 4 consecutive splits; each wave walks the groups of its split on its own
// (no workgroup barrier after the prologue).  A wave adds its k-partial of out[g, d-range] with float
// atomics shaped as contiguous 256-B segments (nkw adds per output element; `out` is initialised by the
// caller) and stores its dW_p^T partial as one slab per split.
//
// KS is kept in the signature for the launch table only (KS == DT: one pass over the tile per group).
// EXACT: D is a multiple of both tile widths -> no column masks / clamped offsets (fewer VGPRs and VALU).
// Column layout of the wide-load contraction kernel: local column c of an 80-wide (or 64-wide) wave tile lives in MFMA
// tile c&3 at lane-row c>>2 for c < 64 and in tile 4 at lane-row c-64 above; tile_pos is where that column sits in the
// tile-major order (16*tile + row) the LDS W_p^T image and the accumulators are indexed by.
__device__ __forceinline__ int tile_col(int tile, int row) { return tile < 4 ? 4 * row + tile : 64 + row; }
__device__ __forceinline__ int tile_pos(int c) { return c < 64 ? 16 * (c & 3) + (c >> 2) : c; }

// BF16 (NRM_MMA_BF16): the same grouped contraction on v_mfma_f32_16x16x32_bf16.  Operands are read exactly as in the fp32
// form (fp32 rows, the same lane -> column map); eight 4-row reduction steps are rounded to bf16 and packed into ONE
// 32-deep MFMA operand per tile -- reduction position (lane quarter q, element j) is row 32*ss + 4*j + q for BOTH operands, which
// is all a dot product needs.  Accumulators (E and dW_p), the epilogue and every reduction stay fp32.
// MMA == 2 (NRM_MMA_BF16X3): both operands are split hi + lo (lo = the bf16 rounding remainder) and every product takes three
// MFMAs (lo*hi + hi*lo + hi*hi): fp32-class accuracy at 3/16 of the fp32 MFMA time.
// WITH_DT = false (bf16 forms, WITH_DW): the pass only accumulates dW_p -- no W_p^T tile in LDS, no per-group out-row flush; dt
// then comes from pwattn_bwd_rw.hip.  XHL4: X is stored in the NRM_DZ_HL4 format (bf16 hi/lo pairs written by the dz pass):
// its MFMA operand is assembled with one v_perm_b32 per dword instead of a convert / subtract / convert per element.
template <int KT, int DT, int KS, bool WITH_DW, bool EXACT, int MMA = 0, bool WITH_DT = true, bool XHL4 = false>
__global__ __launch_bounds__(256, (MMA && KT > 4) || (MMA == 2 && WITH_DW && WITH_DT) ? 1 : 2) void bwd_e_kernel(const BwdEParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // buffer-descriptor builtins: device pass only (keeps the host stub)
    static_assert(KS == DT, "one pass over the whole d range per group (the 3+2 sub-pass split is gone)");
    constexpr bool BF16 = MMA != 0;
    static_assert(WITH_DT || WITH_DW, "a pass without dt and without dW_p has nothing to do");
    static_assert(!XHL4 || (BF16 && KT == 4), "hl4 operands: bf16 forms on 4x4 tiles");
    // epilogue with the W_p^T reads two tiles ahead (and step 1 of the next group requested after it): pays on 5x5 tiles
    // (C3: 4.57 -> 4.48 ms); on 4x4 tiles the serial one-read-one-FMA form is faster (C5, D = 768: 20.6 vs 23.2 ms)
    constexpr bool AHEAD = WITH_DT && WITH_DW && NRM_EPI_AHEAD && !BF16 && KT == 5;
    constexpr int LDK = KT * 16 + 8;      // padded row of the W_p^T tile: +8 is conflict-free under gfx950 b128 lane groups (+4 is 2-way)
    __shared__ __attribute__((aligned(16))) float smem[WITH_DT ? DT * 16 * LDK + 4 * DT * 16 : 4];
    float* wpt = smem;                                                  // [DT*16 d][LDK]  = W_p[k0+k][d0+d]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably uniform -> SGPR math below
    float* bounce = smem + (WITH_DT ? DT * 16 * LDK + wave * (DT * 16) : 0);   // wave-private
    const int r16 = lane & 15, q = lane >> 4;
    const int D = p.D, R = p.R;

    // XCD-aware block order.  The ndcol tiles that share one (split group, k-range) read the SAME dz bytes:
    // give them consecutive logical ids and map consecutive logical ids to one XCD (hardware deals linear
    // block ids round-robin over the 8 XCDs, each with its own L2), so that four of the five reads of a dz
    // line are L2 hits instead of HBM/fabric fetches.  Pure speed: any mapping is correct.
    const int nblk = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;         // bijective for any nblk
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    // order 1 (big groups, C5): k-range-major -- an XCD then holds few k-ranges (nkw / 8 of them), so the dz columns it streams
    // are ITS columns: every dz byte is fetched by one XCD only, and its co-resident workgroups (a k-range's d-columns of a few
    // split groups) share each dz slice while it is hot.  With order 0 the ndcol * nkw tile kinds of one split group exceed the
    // XCD's resident workgroups (144 kinds against 64 at D = 768), so the kinds run in batches and every batch streams the
    // groups' history rows again.
    const int dcol = logical % p.ndcol;
    int kw, sgrp;
    if (p.order == 1) {
        const int nsg = nblk / (p.ndcol * p.nkw);
        sgrp = (logical / p.ndcol) % nsg;
        kw = logical / (p.ndcol * nsg);
    } else {
        kw = (logical / p.ndcol) % p.nkw;
        sgrp = logical / (p.ndcol * p.nkw);
    }
    const int d0 = dcol * (DT * 16);
    const int k0 = kw * (KT * 16);

    // ---- prologue: W_p^T tile -> LDS (zero padded), one barrier
    if (WITH_DT) {
    for (int idx = tid; idx < KT * 16 * DT * 4; idx += 256) {
        const int kl = idx / (DT * 4), d4 = idx - kl * (DT * 4);
        const int k = k0 + kl, d = d0 + 4 * d4;
        f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < D && d < D) w = *reinterpret_cast<const f32x4*>(p.wp + (long)k * p.ldwp + d);     // D % 4 == 0
#pragma unroll
        for (int e = 0; e < 4; ++e) wpt[tile_pos(4 * d4 + e) * LDK + tile_pos(kl)] = w[e];
    }
    __syncthreads();
    }

    const int split = sgrp * 4 + wave;
    if (split >= p.nsplit) return;
    // Group walk of a wave.  Blocked: the gps consecutive groups of its split.  Interleaved (p.interleave): the 4 waves of the
    // workgroup walk ONE range of 4*gps groups round-robin, so at any time they sit on neighbouring groups -- mostly the
    // same g1, i.e. the same Y rows, and a quarter of the distinct rows in flight per XCD (what the L2 has to hold).
    int g_lo = split * p.gps, g_step = 1;
    int g_hi = min(p.G, g_lo + p.gps);
    if (p.interleave) {
        const int nact = min(4, p.nsplit - sgrp * 4);
        g_lo = sgrp * 4 * p.gps + wave; g_step = nact;
        g_hi = min(p.G, (sgrp * 4 + nact) * p.gps);
    }
    const int nsteps = (R + 3) >> 2;

    // Operands stream from global memory straight into MFMA operand registers through buffer descriptors
    // rebuilt per group (uniform base = the group's first row, extent = R rows): the per-lane part of the
    // address is a loop-invariant 32-bit offset (row q of the step, columns of the lane), the step advances
    // the scalar offset, and rows >= R read as 0 by the hardware bounds check -- no masks, no address VALU.
    // Column layout (tile_col): ONE 16-byte load per lane feeds tiles 0..3 and one dword load tile 4, i.e. 4
    // vector-memory instructions per reduction step, each a set of whole 256-B / 64-B row segments.
    // Columns >= D (general shapes) get the out-of-range offset and read as 0.
    constexpr unsigned OOB = 0x80000000u;
    const unsigned vx4 = (EXACT || k0 + 4 * r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 4 * r16) * 4u : OOB;
    const unsigned vx1 = (EXACT || k0 + 64 + r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 64 + r16) * 4u : OOB;
    const unsigned vy4 = (EXACT || d0 + 4 * r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 4 * r16) * 4u : OOB;
    const unsigned vy1 = (EXACT || d0 + 64 + r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 64 + r16) * 4u : OOB;
    const unsigned xbytes = (unsigned)(((long)(R - 1) * p.xrs + D) * 4);
    const unsigned ybytes = (unsigned)(((long)(R - 1) * p.yrs + D) * 4);
    const int xstep = (int)(p.xrs * 16), ystep = (int)(p.yrs * 16);     // 4 rows, bytes

    f32x4 dW[KT][DT];
    if (WITH_DW) {
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = 0; jt < DT; ++jt) dW[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // two operand register sets (ping-pong, no copies)
    float a0[KT], b0[DT], a1[KT], b1[DT];
    __amdgpu_buffer_rsrc_t rx, ry;
    auto open_group = [&](int gg) {
        const int g1 = gg / p.G2, g2 = gg - g1 * p.G2;
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)g1 * p.xs1 + (long)g2 * p.xs2), 0, xbytes, 0x00020000);
        ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Y + (long)g1 * p.ys1), 0, ybytes, 0x00020000);
    };
    // operands of reduction step s (rows 4s..4s+3) of the open group
    bool diag_first = true;
    auto load_step = [&](float (&a)[KT], float (&b)[DT], int s) {
        if (NRM_DIAG_NOLOAD == 1 && !diag_first) return;
        if (NRM_DIAG_NOLOAD == 4) s = 0;                              // same instruction stream, always the same (L1-resident) rows
        if (!(NRM_DIAG_NOLOAD == 2 && !diag_first)) {
            const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rx, vx4, s * xstep, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = __uint_as_float(va[e]);
        }
        if (!(NRM_DIAG_NOLOAD == 3 && !diag_first)) {
            const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(ry, vy4, s * ystep, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = __uint_as_float(vb[e]);
        }
        if (KT > 4 && !(NRM_DIAG_NOLOAD == 2 && !diag_first))
            a[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx1, s * xstep, 0));
        if (DT > 4 && !(NRM_DIAG_NOLOAD == 3 && !diag_first))
            b[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vy1, s * ystep, 0));
        if (NRM_DIAG_GELU_AT_LOAD) {
#pragma unroll
            for (int e = 0; e < KT; ++e) a[e] = b[0] * gelu_grad_f(a[e]);       // ds * w2 * gelu'(z): one multiply stands in for the scale
        }
    };

    // One group: on entry sets 0 and 1 hold steps 0 and 1.  After its last MFMA batch it prefetches steps 0 and 1 of
    // group gn: their latency hides under the epilogue, and -- vmcnt retires in issue order -- the first load issued
    // AFTER this group's float atomics is then only needed two MFMA batches later.
    auto group_pass = [&](int g, int gn) {
        // scale row of this group for the dW_p update: loaded here, consumed in the epilogue
        float sr[DT];
        if (WITH_DW) {
            const float* srow = p.srow + (long)g * p.lds_;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(srow + ((EXACT || d0 + 4 * r16 < D) ? d0 + 4 * r16 : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) sr[e] = s4[e];
            if (DT > 4) sr[4] = srow[(EXACT || d0 + 64 + r16 < D) ? d0 + 64 + r16 : 0];
        }
        f32x4 E[KT][DT];
        if (BF16) {
            const int nss = (R + 31) >> 5;
            for (int ss = 0; ss < nss; ++ss) {
                bf16x8 af[KT], bf[DT], al[KT], bl[DT];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    load_step(a0, b0, 8 * ss + 2 * jp);
                    load_step(a1, b1, 8 * ss + 2 * jp + 1);
                    if (XHL4) {
                        // a0 / a1 hold the raw hl4 units of rows 2jp / 2jp+1: dwords {hi01, hi23, lo01, lo23} of columns 4*r16 + {0..3};
                        // tile it takes 16-bit half (it & 1) of dword (it >> 1): one v_perm_b32 joins the two rows' halves
#pragma unroll
                        for (int it = 0; it < KT; ++it) {
                            const unsigned sel = (it & 1) ? 0x07060302u : 0x05040100u;
                            const unsigned hw = __builtin_amdgcn_perm(__float_as_uint(a1[it >> 1]), __float_as_uint(a0[it >> 1]), sel);
                            af[it][2 * jp] = __builtin_bit_cast(__bf16, (unsigned short)(hw & 0xffffu));
                            af[it][2 * jp + 1] = __builtin_bit_cast(__bf16, (unsigned short)(hw >> 16));
                            if (MMA == 2) {
                                const unsigned lw = __builtin_amdgcn_perm(__float_as_uint(a1[2 + (it >> 1)]), __float_as_uint(a0[2 + (it >> 1)]), sel);
                                al[it][2 * jp] = __builtin_bit_cast(__bf16, (unsigned short)(lw & 0xffffu));
                                al[it][2 * jp + 1] = __builtin_bit_cast(__bf16, (unsigned short)(lw >> 16));
                            }
                        }
                    } else {
#pragma unroll
                    for (int it = 0; it < KT; ++it) {
                        af[it][2 * jp] = (__bf16)a0[it]; af[it][2 * jp + 1] = (__bf16)a1[it];
                        if (MMA == 2) { al[it][2 * jp] = (__bf16)(a0[it] - (float)af[it][2 * jp]); al[it][2 * jp + 1] = (__bf16)(a1[it] - (float)af[it][2 * jp + 1]); }
                    }
                    }
#pragma unroll
                    for (int jt = 0; jt < DT; ++jt) {
                        bf[jt][2 * jp] = (__bf16)b0[jt]; bf[jt][2 * jp + 1] = (__bf16)b1[jt];
                        if (MMA == 2) { bl[jt][2 * jp] = (__bf16)(b0[jt] - (float)bf[jt][2 * jp]); bl[jt][2 * jp + 1] = (__bf16)(b1[jt] - (float)bf[jt][2 * jp + 1]); }
                    }
                }
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int jt = 0; jt < DT; ++jt) {
                        f32x4 c = E[it][jt];
                        if (ss == 0) c = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (MMA == 2) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[it], bf[jt], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], bl[jt], c, 0, 0, 0);
                        }
                        E[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], bf[jt], c, 0, 0, 0);
                    }
            }
            if (gn < g_hi) open_group(gn);
        }
        auto mfma_batch = [&](const float (&a)[KT], const float (&b)[DT]) {
#pragma unroll
            for (int it = 0; it < 4; ++it)                              // fed by the two 16-byte loads
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) E[it][jt] = mfma16(a[it], b[jt], E[it][jt]);
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int jt = (it < 4 ? 4 : 0); jt < DT; ++jt) E[it][jt] = mfma16(a[it], b[jt], E[it][jt]);
        };
        if (!BF16) {
        if (NRM_PRIO) __builtin_amdgcn_s_setprio(NRM_PRIO);           // reduction steps above the co-resident wave's epilogue
        // step 0 accumulates onto an inline-constant 0 (no accumulator clearing)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) E[it][jt] = mfma16(a0[it], b0[jt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = (it < 4 ? 4 : 0); jt < DT; ++jt) E[it][jt] = mfma16(a0[it], b0[jt], f32x4{0.f, 0.f, 0.f, 0.f});
        if (2 < nsteps) load_step(a0, b0, 2);
        if (1 < nsteps) {
            mfma_batch(a1, b1);
            if (3 < nsteps) load_step(a1, b1, 3);
        }
        for (int s = 2; s < nsteps; s += 2) {
            mfma_batch(a0, b0);
            if (s + 2 < nsteps) load_step(a0, b0, s + 2);
            if (s + 1 < nsteps) {
                mfma_batch(a1, b1);
                if (s + 3 < nsteps) load_step(a1, b1, s + 3);
            }
        }
        if (gn < g_hi) {
            open_group(gn);
            load_step(a0, b0, 0);
            // WITH_DW: step 1 of the next group is requested AFTER the epilogue arithmetic (still before this group's float
            // atomics, which vmcnt would otherwise put in front of it): operand set 1 is dead during the epilogue, and
            // those registers let the W_p^T reads run two tiles ahead of their FMAs instead of being waited one by one
            if (!AHEAD && nsteps > 1) load_step(a1, b1, 1);
        }
        if (NRM_PRIO) __builtin_amdgcn_s_setprio(0);
        }

        // epilogue: lane holds E[k = k0 + tile_col(it, 4q+e)][d = d0 + tile_col(jt, r16)]; the LDS image of W_p^T is
        // in tile-major order (tile_pos), so its rows/columns are addressed by 16*tile + lane-row as the accumulators are
        if (NRM_DIAG_NOEPI && gn < g_hi) {                           // timing diagnostic: one cheap use keeps E alive
            float keep = 0.f;
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int jt = 0; jt < DT; ++jt) keep += E[it][jt][0];
            if (keep == 123.456f) bounce[r16] = keep;
            return;
        }
        if (!WITH_DT) {                                              // dW_p only: no k-reduction, no out row
#pragma unroll
            for (int jt = 0; jt < DT; ++jt)
#pragma unroll
                for (int it = 0; it < KT; ++it) dW[it][jt] += E[it][jt] * sr[jt];
            return;
        }
        if (AHEAD) {
            // flattened (jt, it) order, LDS reads two tiles ahead of their FMAs
            constexpr int NTILE = KT * DT;
            auto rd = [&](int n) { return *reinterpret_cast<const f32x4*>(&wpt[(16 * (n / KT) + r16) * LDK + 16 * (n % KT) + 4 * q]); };
            f32x4 wa = rd(0), wb = rd(1), wc;
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < NTILE; ++n) {
                const int jt = n / KT, it = n % KT;
                if (n + 2 < NTILE) wc = rd(n + 2);
                const f32x4 e4 = E[it][jt];
                acc = fmaf(wa[0], e4[0], fmaf(wa[1], e4[1], fmaf(wa[2], e4[2], fmaf(wa[3], e4[3], acc))));
                dW[it][jt] += e4 * sr[jt];
                if (it == KT - 1) {
                    acc = sum_rows4(acc);
                    if (q == 0) bounce[tile_col(jt, r16)] = acc;
                    acc = 0.f;
                }
                wa = wb; wb = wc;
            }
            if (gn < g_hi && nsteps > 1) load_step(a1, b1, 1);
            return;
        }
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) {
            const int dl = 16 * jt + r16;
            // without dW_p (3 waves/SIMD, registers to spare) the column's LDS reads go out together: one latency per
            // column instead of one per read (serial dh pass 4.36 -> 4.27 ms); with dW_p live that costs registers and
            // measured 10 % slower on 4x4-tile shapes (D = 768: 22.8 vs 20.5 ms), so there each read feeds its FMAs at once
            constexpr bool BATCH = !WITH_DW;
            f32x4 w4[KT];
            if (BATCH) {
#pragma unroll
                for (int it = 0; it < KT; ++it)
                    w4[it] = *reinterpret_cast<const f32x4*>(&wpt[dl * LDK + 16 * it + 4 * q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < KT; ++it) {
                if (!BATCH) w4[it] = *reinterpret_cast<const f32x4*>(&wpt[dl * LDK + 16 * it + 4 * q]);
                const f32x4 e4 = E[it][jt];
                acc = fmaf(w4[it][0], e4[0], fmaf(w4[it][1], e4[1], fmaf(w4[it][2], e4[2], fmaf(w4[it][3], e4[3], acc))));
                if (WITH_DW) dW[it][jt] += e4 * sr[jt];
            }
            acc = sum_rows4(acc);
            if (q == 0) bounce[tile_col(jt, r16)] = acc;
        }
    };

    if (g_lo < g_hi) {
        open_group(g_lo);
        if (!BF16) {
            load_step(a0, b0, 0);
            if (nsteps > 1) load_step(a1, b1, 1);
        }
    }
    diag_first = false;
    for (int g = g_lo; g < g_hi; g += g_step) {
        group_pass(g, g + g_step);
        // k-partial of out[g, d0 .. d0+DT*16): one dword per lane, contiguous segments.
        // bounce[] is wave-private: LDS ops of one wave complete in order, the fences only pin hipcc.
        if (NRM_DIAG_NOEPI && g + g_step < g_hi) continue;
        if (!WITH_DT) continue;
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        float* orow = p.out + (long)g * p.ldo + d0;
#pragma unroll
        for (int c = 0; c < DT * 16; c += 64) {
            const int dl = c + lane;
            if (!NRM_DIAG_NOATOM && dl < DT * 16 && (EXACT || d0 + dl < D)) atomicAdd(orow + dl, bounce[dl]);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
    }

    if (WITH_DW) {      // slab layout is TRANSPOSED: ws[split][d][k]
        // lane holds dW[it][jt][e] = dW_p[k0 + tile_col(it, 4q+e)][d0 + tile_col(jt, r16)]: for one d, tiles 0..3 cover
        // the 16 consecutive k  k0 + 16q + 4e + it  (a float4 per e, gathered across the four accumulators) and tile 4
        // the 4 consecutive k  k0 + 64 + 4q + e.
        float* wsp = p.ws + (long)split * D * D;
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) {
            const int d = d0 + tile_col(jt, r16);
            if (!(EXACT || d < D)) continue;
            float* row = wsp + (long)d * D + k0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 16 * q + 4 * e;
                if (EXACT || k0 + kk < D)
                    *reinterpret_cast<f32x4*>(row + kk) = f32x4{dW[0][jt][e], dW[1][jt][e], dW[2][jt][e], dW[3][jt][e]};
            }
            if (KT > 4 && (EXACT || k0 + 64 + 4 * q < D)) *reinterpret_cast<f32x4*>(row + 64 + 4 * q) = dW[KT - 1][jt];
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// (2a) the dW_p-only pass in fp32 with ONE accumulator set (round 5).  bwd_e_kernel<..., WITH_DT = false> forms E_g = X_g^T Y_g per
// group in one accumulator set and then adds E_g (.) srow_g to dW_p in a second one: 200 accumulator registers (two waves per
// SIMD) and 100 four-wide FMAs between the MFMA streams of two groups.  But dW_p[k,d] = sum_g sum_r X_g[r,k] (Y_g[r,d] srow_g[d])
// is ONE long reduction over (g, r) once the scale row is folded into the B operand (5 multiplies per reduction step, the
// forward's P = t (.) h): a single accumulator set (three waves per SIMD), no per-group epilogue, and the operand prefetch
// (two steps ahead) runs straight across group boundaries.  Same decomposition, grid and slab layout as bwd_e_kernel.
// Needs >= NSET reduction steps per group (the scale row of the group whose operands are being requested is held beside the one
// being consumed: the request pointer may be one group ahead, not two); the launcher keeps bwd_e_kernel for fewer.
#ifndef DW_SETS
#define DW_SETS 4          // tuning: operand register sets = reduction steps requested ahead
#endif
#ifndef DW_WPE
#define DW_WPE 3           // tuning: launch bound, waves per SIMD
#endif
template <int KT, int DT, bool EXACT, int NSET = DW_SETS>
__global__ __launch_bounds__(256, DW_WPE) void bwd_dw_direct_kernel(const BwdEParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int D = p.D, R = p.R;
    // XCD-aware block order, as bwd_e_kernel
    const int nblk = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int dcol = logical % p.ndcol;
    int kw, sgrp;
    if (p.order == 1) {
        const int nsg = nblk / (p.ndcol * p.nkw);
        sgrp = (logical / p.ndcol) % nsg;
        kw = logical / (p.ndcol * nsg);
    } else {
        kw = (logical / p.ndcol) % p.nkw;
        sgrp = logical / (p.ndcol * p.nkw);
    }
    const int d0 = dcol * (DT * 16);
    const int k0 = kw * (KT * 16);
    const int split = sgrp * 4 + wave;
    if (split >= p.nsplit) return;
    int g_lo = split * p.gps, g_step = 1;
    int g_hi = min(p.G, g_lo + p.gps);
    if (p.interleave) {
        const int nact = min(4, p.nsplit - sgrp * 4);
        g_lo = sgrp * 4 * p.gps + wave; g_step = nact;
        g_hi = min(p.G, (sgrp * 4 + nact) * p.gps);
    }
    const int nsteps = (R + 3) >> 2;                                     // >= 2 (launcher)

    constexpr unsigned OOB = 0x80000000u;
    const unsigned vx4 = (EXACT || k0 + 4 * r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 4 * r16) * 4u : OOB;
    const unsigned vx1 = (EXACT || k0 + 64 + r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 64 + r16) * 4u : OOB;
    const unsigned vy4 = (EXACT || d0 + 4 * r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 4 * r16) * 4u : OOB;
    const unsigned vy1 = (EXACT || d0 + 64 + r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 64 + r16) * 4u : OOB;
    const unsigned xbytes = (unsigned)(((long)(R - 1) * p.xrs + D) * 4);
    const unsigned ybytes = (unsigned)(((long)(R - 1) * p.yrs + D) * 4);
    const int xstep = (int)(p.xrs * 16), ystep = (int)(p.yrs * 16);     // 4 rows, bytes
    const int s4off = (EXACT || d0 + 4 * r16 < D) ? d0 + 4 * r16 : 0;   // this lane's columns of a scale row (clamped: the product is
    const int s1off = (EXACT || d0 + 64 + r16 < D) ? d0 + 64 + r16 : 0; // with an operand that reads 0 there)

    f32x4 dW[KT][DT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) dW[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g_lo < g_hi) {
        float a[NSET][KT], b[NSET][DT];                                 // operand register sets (round-robin, no copies)
        float sr_cur[DT], sr_nxt[DT];                                   // scale rows: group being consumed / group being requested
        __amdgpu_buffer_rsrc_t rx, ry;
        int lg = g_lo, ls = 0;                                          // request pointer: (group, step); lg >= g_hi: nothing left
        auto open_group = [&](int gg) {
            const int g1 = gg / p.G2, g2 = gg - g1 * p.G2;
            rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)g1 * p.xs1 + (long)g2 * p.xs2), 0, xbytes, 0x00020000);
            ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Y + (long)g1 * p.ys1), 0, ybytes, 0x00020000);
            const float* srow = p.srow + (long)gg * p.lds_;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(srow + s4off);
#pragma unroll
            for (int e = 0; e < 4; ++e) sr_nxt[e] = s4[e];
            if (DT > 4) sr_nxt[4] = srow[s1off];
        };
        // request the operands of the step under the request pointer into (a, b), advance the pointer
        auto request = [&](float (&a)[KT], float (&b)[DT]) {
            if (lg >= g_hi) return;
            if (ls == 0) open_group(lg);
            const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rx, vx4, ls * xstep, 0);
            const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(ry, vy4, ls * ystep, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = __uint_as_float(va[e]); b[e] = __uint_as_float(vb[e]); }
            if (KT > 4) a[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx1, ls * xstep, 0));
            if (DT > 4) b[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vy1, ls * ystep, 0));
            if (++ls == nsteps) { ls = 0; lg += g_step; }
        };
        int cs = 0;                                                      // step of the group being consumed
        auto consume = [&](const float (&a)[KT], const float (&b)[DT]) {
            if (cs == 0) {
#pragma unroll
                for (int jt = 0; jt < DT; ++jt) sr_cur[jt] = sr_nxt[jt];
            }
            float bs[DT];
#pragma unroll
            for (int jt = 0; jt < DT; ++jt) bs[jt] = b[jt] * sr_cur[jt];
#pragma unroll
            for (int it = 0; it < 4; ++it)                              // fed by the two 16-byte loads
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) dW[it][jt] = mfma16(a[it], bs[jt], dW[it][jt]);
#pragma unroll
            for (int it = 0; it < KT; ++it)
#pragma unroll
                for (int jt = (it < 4 ? 4 : 0); jt < DT; ++jt) dW[it][jt] = mfma16(a[it], bs[jt], dW[it][jt]);
            if (++cs == nsteps) cs = 0;
        };
        const int ngroups = (g_hi - g_lo + g_step - 1) / g_step;
        const int total = ngroups * nsteps;
#pragma unroll
        for (int u = 0; u < NSET; ++u) request(a[u], b[u]);
        for (int n = 0; n < total; n += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u)
                if (u == 0 || n + u < total) {
                    consume(a[u], b[u]);
                    request(a[u], b[u]);
                }
        }
    }

    // slab layout is TRANSPOSED: ws[split][d][k] (as bwd_e_kernel)
    float* wsp = p.ws + (long)split * D * D;
#pragma unroll
    for (int jt = 0; jt < DT; ++jt) {
        const int d = d0 + tile_col(jt, r16);
        if (!(EXACT || d < D)) continue;
        float* row = wsp + (long)d * D + k0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 16 * q + 4 * e;
            if (EXACT || k0 + kk < D)
                *reinterpret_cast<f32x4*>(row + kk) = f32x4{dW[0][jt][e], dW[1][jt][e], dW[2][jt][e], dW[3][jt][e]};
        }
        if (KT > 4 && (EXACT || k0 + 64 + 4 * q < D)) *reinterpret_cast<f32x4*>(row + 64 + 4 * q) = dW[KT - 1][jt];
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// (2b) software-pipelined variant of the contraction WITHOUT dW_p (the dh pass).  The epilogue of a group
// (LDS reads of W_p^T, FMAs, row reduction, bounce, atomics) costs several times its instruction time when it
// runs as one serial block between two MFMA streams (DESIGN.md "stamp findings").  Here a wave keeps TWO
// accumulator sets: while the MFMAs of group g fill one, the epilogue of group g-1 drains the other, one
// d tile per reduction step, inside the same straight-line code -- the matrix pipe never waits for it.
// Costs 2x accumulators (2 waves/SIMD).  Needs nsteps >= PE + 2 (the peeled steps load only from the current
// group); the launcher falls back to bwd_e_kernel otherwise.
template <int KT, int DT, bool EXACT>
__global__ __launch_bounds__(256, 2) void bwd_e_pipe_kernel(const BwdEParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PE = (DT + 1) & ~1;                                  // peeled steps (even, >= DT)
    constexpr int LDK = KT * 16 + 8;
    __shared__ __attribute__((aligned(16))) float smem[DT * 16 * LDK + 4 * DT * 16];
    float* wpt = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* bounce = smem + DT * 16 * LDK + wave * (DT * 16);
    const int r16 = lane & 15, q = lane >> 4;
    const int D = p.D, R = p.R;

    const int nblk = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int dcol = logical % p.ndcol;
    const int kw = (logical / p.ndcol) % p.nkw;
    const int sgrp = logical / (p.ndcol * p.nkw);
    const int d0 = dcol * (DT * 16);
    const int k0 = kw * (KT * 16);

    for (int idx = tid; idx < KT * 16 * DT * 4; idx += 256) {
        const int kl = idx / (DT * 4), d4 = idx - kl * (DT * 4);
        const int k = k0 + kl, d = d0 + 4 * d4;
        f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < D && d < D) w = *reinterpret_cast<const f32x4*>(p.wp + (long)k * p.ldwp + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) wpt[tile_pos(4 * d4 + e) * LDK + tile_pos(kl)] = w[e];
    }
    __syncthreads();

    const int split = sgrp * 4 + wave;
    if (split >= p.nsplit) return;
    const int g_lo = split * p.gps;
    const int g_hi = min(p.G, g_lo + p.gps);
    if (g_lo >= g_hi) return;
    const int nsteps = (R + 3) >> 2;

    // Lane (r16, q) of MFMA tile `it` holds local column tile_col(it, r16): tiles 0..3 interleave over the first 64
    // columns (4*r16 + it) and tile 4 is the tail (64 + r16), so that ONE 16-byte load per lane feeds tiles 0..3 and
    // one dword load tile 4 -- 4 vector-memory instructions per reduction step instead of 10, all of them whole
    // 256-B / 64-B row segments.  Columns >= D (general shapes) get the out-of-range offset and read as 0.
    constexpr unsigned OOB = 0x80000000u;
    const unsigned vx4 = (EXACT || k0 + 4 * r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 4 * r16) * 4u : OOB;
    const unsigned vx1 = (EXACT || k0 + 64 + r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 64 + r16) * 4u : OOB;
    const unsigned vy4 = (EXACT || d0 + 4 * r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 4 * r16) * 4u : OOB;
    const unsigned vy1 = (EXACT || d0 + 64 + r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 64 + r16) * 4u : OOB;
    const unsigned xbytes = (unsigned)(((long)(R - 1) * p.xrs + D) * 4);
    const unsigned ybytes = (unsigned)(((long)(R - 1) * p.yrs + D) * 4);
    const int xstep = (int)(p.xrs * 16), ystep = (int)(p.yrs * 16);

    float a0[KT], b0[DT], a1[KT], b1[DT];
    __amdgpu_buffer_rsrc_t rx, ry, rxn, ryn;                           // current / next group
    auto open_group = [&](int gg, __amdgpu_buffer_rsrc_t& ox, __amdgpu_buffer_rsrc_t& oy) {
        const int g1 = gg / p.G2, g2 = gg - g1 * p.G2;
        ox = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)g1 * p.xs1 + (long)g2 * p.xs2), 0, xbytes, 0x00020000);
        oy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Y + (long)g1 * p.ys1), 0, ybytes, 0x00020000);
    };
    bool diag_first = true;
    auto load_step = [&](__amdgpu_buffer_rsrc_t dx, __amdgpu_buffer_rsrc_t dy, float (&a)[KT], float (&b)[DT], int s) {
        if (NRM_DIAG_NOLOAD == 1 && !(diag_first && s < 2)) return;
        const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(dx, vx4, s * xstep, 0);
        const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(dy, vy4, s * ystep, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = __uint_as_float(va[e]); b[e] = __uint_as_float(vb[e]); }
        if (KT > 4) a[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dx, vx1, s * xstep, 0));
        if (DT > 4) b[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dy, vy1, s * ystep, 0));
    };
    auto mfma_batch = [&](f32x4 (&E)[KT][DT], const float (&a)[KT], const float (&b)[DT]) {
#pragma unroll
        for (int it = 0; it < 4; ++it)                                  // fed by the two 16-byte loads
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) E[it][jt] = mfma16(a[it], b[jt], E[it][jt]);
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = (it < 4 ? 4 : 0); jt < DT; ++jt) E[it][jt] = mfma16(a[it], b[jt], E[it][jt]);
    };
    // epilogue slice: d tile jt of a finished accumulator set -> bounce[16*jt .. 16*jt+15]
    auto slice = [&](const f32x4 (&E)[KT][DT], int jt) {
        const int dl = 16 * jt + r16;
        f32x4 w4[KT];                                                    // the five LDS reads of the column go out together
#pragma unroll
        for (int it = 0; it < KT; ++it) w4[it] = *reinterpret_cast<const f32x4*>(&wpt[dl * LDK + 16 * it + 4 * q]);
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const f32x4 e4 = E[it][jt];
            acc = fmaf(w4[it][0], e4[0], fmaf(w4[it][1], e4[1], fmaf(w4[it][2], e4[2], fmaf(w4[it][3], e4[3], acc))));
        }
        acc = sum_rows4(acc);
        if (q == 0) bounce[tile_col(jt, r16)] = acc;
    };
    auto flush = [&](int grow) {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        float* orow = p.out + (long)grow * p.ldo + d0;
#pragma unroll
        for (int c = 0; c < DT * 16; c += 64) {
            const int dl = c + lane;
            if (!NRM_DIAG_NOATOM && dl < DT * 16 && (EXACT || d0 + dl < D)) atomicAdd(orow + dl, bounce[dl]);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
    };
    // Group g into Ec while Ep (group `prow`) drains.  On entry sets 0/1 hold steps 0/1 of g and (rx, ry) is
    // open on g; on exit they hold steps 0/1 of g+1 (if any) and (rx, ry) is open on g+1.
    auto body = [&](f32x4 (&Ec)[KT][DT], const f32x4 (&Ep)[KT][DT], int g, int prow) {
        const bool has_next = g + 1 < g_hi;
        if (has_next) open_group(g + 1, rxn, ryn);
#pragma unroll
        for (int s = 0; s < PE; ++s) {                                  // nsteps >= PE + 2: loads stay in group g
            if (s == 0 && !NRM_DIAG_NOEPI) {                            // C = inline 0: no accumulator clearing
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) Ec[it][jt] = mfma16(a0[it], b0[jt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int it = 0; it < KT; ++it)
#pragma unroll
                    for (int jt = (it < 4 ? 4 : 0); jt < DT; ++jt) Ec[it][jt] = mfma16(a0[it], b0[jt], f32x4{0.f, 0.f, 0.f, 0.f});
                load_step(rx, ry, a0, b0, 2);
            }
            else if (s & 1) { mfma_batch(Ec, a1, b1); load_step(rx, ry, a1, b1, s + 2); }
            else            { mfma_batch(Ec, a0, b0); load_step(rx, ry, a0, b0, s + 2); }
            if (s < DT && !NRM_DIAG_NOEPI) {
                slice(Ep, s);
#if NRM_PIPE_SGB
                // place the slice in the shadow of this step's MFMAs: LDS reads + next loads in the first gaps,
                // the FMAs / row reduction two per gap after that
#pragma unroll
                for (int i = 0; i < KT * DT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < KT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i < KT + DT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i >= KT) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
#endif
            }
            if (s == DT - 1 && !NRM_DIAG_NOEPI) flush(prow);
        }
        for (int s = PE; s < nsteps; s += 2) {
            mfma_batch(Ec, a0, b0);
            if (s + 2 < nsteps) load_step(rx, ry, a0, b0, s + 2);
            else if (has_next)  load_step(rxn, ryn, a0, b0, 0);
            if (s + 1 < nsteps) {
                mfma_batch(Ec, a1, b1);
                if (s + 3 < nsteps) load_step(rx, ry, a1, b1, s + 3);
                else if (has_next)  load_step(rxn, ryn, a1, b1, 1);
            } else if (has_next) {
                load_step(rxn, ryn, a1, b1, 1);                         // odd nsteps: set 1 is free already
            }
        }
        rx = rxn; ry = ryn;
    };
    auto drain = [&](const f32x4 (&E)[KT][DT], int grow) {
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) slice(E, jt);
        flush(grow);
    };

    f32x4 E0[KT][DT], E1[KT][DT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) E1[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};   // "group before the first": adds 0
    open_group(g_lo, rx, ry);
    rxn = rx; ryn = ry;
    load_step(rx, ry, a0, b0, 0);
    load_step(rx, ry, a1, b1, 1);
    diag_first = false;
    int g = g_lo;
    int prow = g_lo;                                                    // zeros go to the split's own first row
    for (; g + 1 < g_hi; g += 2) {
        body(E0, E1, g, prow);
        body(E1, E0, g + 1, g);
        prow = g + 1;
    }
    if (g < g_hi) { body(E0, E1, g, prow); drain(E0, g); }
    else          { drain(E1, g - 1); }
#endif
}

// ---------------------------------------------------------------------------------------------
// (2c) dW_p-only pass of the resident-W backward (pwattn_bwd_rw.hip) for groups of at most 32 rows (one 32-deep MFMA step per
// group; C2: H = 32): the generic bf16 form above requests a group's operands, waits, converts and only then issues its 48 MFMAs
// -- two thirds of its time is operand traffic (timing-only builds: 0.29 ms with, 0.10 ms without the loads at C2).  Here
//   * Y_g (the history rows h[b]) depends on b only: its bf16 hi / lo fragments are built ONCE per impression and kept in
//     registers across that impression's T groups (no reload, no conversion);
//   * X_g (dz in the NRM_DZ_HL4 format) of group g + 1 is requested before the MFMAs of group g, into the registers whose raw
//     units have just been assembled into fragments (one v_perm_b32 per dword);
//   * no E-epilogue reduction, no LDS: dW_p += E (*) t[b,t,:] per group, slabs as in bwd_e_kernel.
template <bool EXACT, int MMA>
__global__ __launch_bounds__(256, 2) void bwd_dw_r32_kernel(const BwdEParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KT = 4, DT = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int D = p.D, R = p.R;
    const int nblk = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr_ = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + (lin >> 3);
    const int dcol = logical % p.ndcol;
    const int kw = (logical / p.ndcol) % p.nkw;
    const int sgrp = logical / (p.ndcol * p.nkw);
    const int d0 = dcol * 64, k0 = kw * 64;
    const int split = sgrp * 4 + wave;
    if (split >= p.nsplit) return;
    const int g_lo = split * p.gps, g_hi = min(p.G, g_lo + p.gps);

    constexpr unsigned OOB = 0x80000000u;
    const unsigned vx4 = (EXACT || k0 + 4 * r16 < D) ? (unsigned)((long)q * p.xrs + k0 + 4 * r16) * 4u : OOB;
    const unsigned vy4 = (EXACT || d0 + 4 * r16 < D) ? (unsigned)((long)q * p.yrs + d0 + 4 * r16) * 4u : OOB;
    const unsigned xbytes = (unsigned)(((long)(R - 1) * p.xrs + D) * 4);
    const unsigned ybytes = (unsigned)(((long)(R - 1) * p.yrs + D) * 4);
    const int xstep = (int)(p.xrs * 16), ystep = (int)(p.yrs * 16);     // 4 rows, bytes

    f32x4 dW[KT][DT];
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) dW[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // reduction position (lane quarter q, element j) is row 4 j + q of the group, for both operands
    u32x4 xraw[8];
    auto load_x = [&](int g) {
        const int g1 = g / p.G2, g2 = g - g1 * p.G2;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)g1 * p.xs1 + (long)g2 * p.xs2), 0, xbytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j) xraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, vx4, j * xstep, 0);
    };
    bf16x8 ybf[DT], ybl[DT];
    int y_g1 = -1;
    if (g_lo < g_hi) load_x(g_lo);
    for (int g = g_lo; g < g_hi; ++g) {
        const int g1 = g / p.G2;
        if (g1 != y_g1) {                                                // a new impression: its history rows, split once
            y_g1 = g1;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Y + (long)g1 * p.ys1), 0, ybytes, 0x00020000);
            f32x4 yraw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) yraw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, vy4, j * ystep, 0));
#pragma unroll
            for (int jt = 0; jt < DT; ++jt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 hb = (__bf16)yraw[j][jt];
                    ybf[jt][j] = hb;
                    if (MMA == 2) ybl[jt][j] = (__bf16)(yraw[j][jt] - (float)hb);
                }
        }
        // scale row of the group (the candidate's t row), consumed after the MFMAs
        const float* srow = p.srow + (long)g * p.lds_;
        const f32x4 sr = *reinterpret_cast<const f32x4*>(srow + ((EXACT || d0 + 4 * r16 < D) ? d0 + 4 * r16 : 0));
        // X fragments: tile it takes 16-bit half (it & 1) of dword (it >> 1) [hi] / 2 + (it >> 1) [lo] of every row's hl4 unit
        u32x4 afu[KT], alu[KT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const unsigned sel = (it & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                afu[it][jp] = __builtin_amdgcn_perm(xraw[2 * jp + 1][it >> 1], xraw[2 * jp][it >> 1], sel);
                if (MMA == 2) alu[it][jp] = __builtin_amdgcn_perm(xraw[2 * jp + 1][2 + (it >> 1)], xraw[2 * jp][2 + (it >> 1)], sel);
            }
        }
        if (g + 1 < g_hi) load_x(g + 1);                                 // next group's dz rows: under this group's MFMAs
        f32x4 E[KT][DT];
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, afu[it]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, alu[it]);
#pragma unroll
            for (int jt = 0; jt < DT; ++jt) {
                f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
                if (MMA == 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ybf[jt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ybl[jt], c, 0, 0, 0);
                }
                E[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ybf[jt], c, 0, 0, 0);
            }
        }
#pragma unroll
        for (int jt = 0; jt < DT; ++jt)
#pragma unroll
            for (int it = 0; it < KT; ++it) dW[it][jt] += E[it][jt] * sr[jt];
    }

    // slab layout is TRANSPOSED: ws[split][d][k]; lane holds dW[it][jt][e] = dW_p[k0 + 16 q + 4 e + it][d0 + 4 r16 + jt]
    float* wsp = p.ws + (long)split * D * D;
#pragma unroll
    for (int jt = 0; jt < DT; ++jt) {
        const int d = d0 + 4 * r16 + jt;
        if (!(EXACT || d < D)) continue;
        float* row = wsp + (long)d * D + k0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = 16 * q + 4 * e;
            if (EXACT || k0 + kk < D)
                *reinterpret_cast<f32x4*>(row + kk) = f32x4{dW[0][jt][e], dW[1][jt][e], dW[2][jt][e], dW[3][jt][e]};
        }
    }
#endif
}

// min_gps: lower bound on the groups a split walks.  The dt/dW pass passes 96: every split of it writes a [D, D] partial
// slab of dW_p (C2-small at the old 40 groups per split: 384 slabs = 100 MB for a 256 KB gradient; 5.73 -> 5.68 ms per step
// with 120); C3's 126 groups per split are unaffected.
BwdEPlan bwd_e_plan(int D, int G, int target_waves, int min_gps, int mma) {
    const int n16 = (D + 15) / 16;
    const int c5 = (n16 + 4) / 5 * 5, c4 = (n16 + 3) / 4 * 4;
    BwdEPlan pl;
    if (c5 < c4) { pl.DT = 5; pl.KT = 5; } else { pl.DT = 4; pl.KT = 4; }   // the shape that pads less
    // bf16 forms: always 4x4 -- their operand fragments (and the hi/lo split) do not fit beside 5x5 accumulators at two
    // waves per SIMD, and padded MFMAs are cheap there (C3, bf16x3, dt+dW pass: 4.49 ms on 5x5 tiles at one wave per SIMD)
    if (mma != 0) { pl.DT = 4; pl.KT = 4; }
    pl.ndcol = (n16 + pl.DT - 1) / pl.DT;
    pl.nkw = (n16 + pl.KT - 1) / pl.KT;
    const int tiles = pl.ndcol * pl.nkw;
    // tiles * nsplit <= target_waves, nsplit a multiple of 4 (4 splits per workgroup).  target_waves is either
    // at most the number of resident wave slots (ONE round: one task more would add a whole second round) or
    // several times it (many small tasks, dynamically balanced) -- see capi.hip.
    int ns = target_waves / tiles / 4 * 4;
    // too few groups per split: fall back to exactly ONE round of wave tasks (2048 resident wave slots at 2 waves/SIMD) -- never
    // fewer tasks than the chip holds, never a second, mostly empty round
    if (min_gps > 1 && ns > G / min_gps) {
        const int one_round = 2048 / tiles / 4 * 4;
        if (one_round >= 4 && one_round < ns) ns = one_round;
    }
    if (ns < 4) ns = 4;
    if (ns > G) ns = G > 0 ? G : 1;
    pl.gps = (G + ns - 1) / ns;
    pl.nsplit = pl.gps > 0 ? (G + pl.gps - 1) / pl.gps : 1;
    return pl;
}

static bool pipe_enabled() {
    static const bool on = [] { const char* e = getenv("NRM_BH_PIPE"); return !(e && e[0] == '0'); }();
    return on;
}

template <int KT, int DT>
static hipError_t launch_e_t(BwdEParams p, const BwdEPlan& pl, bool with_dw, int mma, hipStream_t st) {
    p.nkw = pl.nkw; p.ndcol = pl.ndcol; p.gps = pl.gps; p.nsplit = pl.nsplit;
    const dim3 grid(pl.nkw * pl.ndcol, (pl.nsplit + 3) / 4), block(256);
    const bool exact = p.D % (16 * KT) == 0 && p.D % (16 * DT) == 0;
    constexpr int KS_DW = DT;     // single pass (40 B of scratch at 5x5); the 3+2 sub-pass split is KS_DW = (DT + 1) / 2
    if constexpr (KT != 4) {
        if (mma != 0) return hipErrorInvalidValue;      // bwd_e_plan gives the bf16 forms 4x4 tiles only
    } else if (mma == 1 || mma == 2) {
        if (!p.with_dt && !p.x_hl4) {                   // dW_p only from fp32 dz (no row gradient wanted; D > 256 keeps fp32 dz)
            if (!with_dw) return hipErrorInvalidValue;
            if (mma == 1) {
                if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, 1, false, false>), grid, block, 0, st, p);
                else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, 1, false, false>), grid, block, 0, st, p);
            } else {
                if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, 2, false, false>), grid, block, 0, st, p);
                else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, 2, false, false>), grid, block, 0, st, p);
            }
            return hipGetLastError();
        }
        if (!p.with_dt || p.x_hl4) {                    // the dW_p-only pass of the resident-W backward (hl4 dz operand)
            if (!with_dw || p.with_dt || !p.x_hl4) return hipErrorInvalidValue;
            static const bool r32 = [] { const char* e = getenv("NRM_DW_R32"); return !(e && e[0] == '0'); }();
            if (r32 && p.R <= 32) {                     // one 32-row MFMA step per group: Y kept in registers, X prefetched
                if (mma == 1) {
                    if (exact) hipLaunchKernelGGL((bwd_dw_r32_kernel<true, 1>), grid, block, 0, st, p);
                    else       hipLaunchKernelGGL((bwd_dw_r32_kernel<false, 1>), grid, block, 0, st, p);
                } else {
                    if (exact) hipLaunchKernelGGL((bwd_dw_r32_kernel<true, 2>), grid, block, 0, st, p);
                    else       hipLaunchKernelGGL((bwd_dw_r32_kernel<false, 2>), grid, block, 0, st, p);
                }
                return hipGetLastError();
            }
            if (mma == 1) {
                if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, 1, false, true>), grid, block, 0, st, p);
                else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, 1, false, true>), grid, block, 0, st, p);
            } else {
                if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, 2, false, true>), grid, block, 0, st, p);
                else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, 2, false, true>), grid, block, 0, st, p);
            }
            return hipGetLastError();
        }
#define NRM_LAUNCH_E(M)                                                                                              \
        if (with_dw) {                                                                                               \
            if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, M>), grid, block, 0, st, p);       \
            else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, M>), grid, block, 0, st, p);      \
        } else {                                                                                                     \
            if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, DT, false, true, M>), grid, block, 0, st, p);         \
            else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, DT, false, false, M>), grid, block, 0, st, p);        \
        }
        if (mma == 1) { NRM_LAUNCH_E(1) } else { NRM_LAUNCH_E(2) }
#undef NRM_LAUNCH_E
        return hipGetLastError();
    }
    if (with_dw && !p.with_dt) {                        // fp32, dW_p only (no row gradient wanted, or beside the dP walk)
        if (p.x_hl4) return hipErrorInvalidValue;
        const char* e = getenv("NRM_DW_DIRECT");        // (read per launch: tests switch forms inside one process)
        if (!(e && e[0] == '0') && (p.R + 3) / 4 >= DW_SETS) { // one accumulator set, the scale row folded into the operand
            if (exact) hipLaunchKernelGGL((bwd_dw_direct_kernel<KT, DT, true>), grid, block, 0, st, p);
            else       hipLaunchKernelGGL((bwd_dw_direct_kernel<KT, DT, false>), grid, block, 0, st, p);
        } else
        if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true, 0, false, false>), grid, block, 0, st, p);
        else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false, 0, false, false>), grid, block, 0, st, p);
    } else if (with_dw) {
        if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, true>), grid, block, 0, st, p);
        else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, KS_DW, true, false>), grid, block, 0, st, p);
    } else if (pipe_enabled() && (p.R + 3) / 4 >= ((DT + 1) & ~1) + 2) {
        if (exact) hipLaunchKernelGGL((bwd_e_pipe_kernel<KT, DT, true>), grid, block, 0, st, p);
        else       hipLaunchKernelGGL((bwd_e_pipe_kernel<KT, DT, false>), grid, block, 0, st, p);
    } else {
        if (exact) hipLaunchKernelGGL((bwd_e_kernel<KT, DT, DT, false, true>), grid, block, 0, st, p);
        else       hipLaunchKernelGGL((bwd_e_kernel<KT, DT, DT, false, false>), grid, block, 0, st, p);
    }
    return hipGetLastError();
}

int pwattn_bwd_diag_flags() {
    return (NRM_EPI_AHEAD != 1 ? 8 : 0) | (NRM_DIAG_GELU_AT_LOAD ? 16 : 0) | (NRM_DIAG_NOEPI ? 32 : 0) | (NRM_DIAG_NOATOM ? 64 : 0) |
           (NRM_DIAG_NOLOAD ? 128 : 0) | (NRM_PIPE_SGB != 1 ? 256 : 0);
}

hipError_t bwd_e_launch(const BwdEParams& p, const BwdEPlan& pl, bool with_dw, int mma, hipStream_t st) {
    if (p.G <= 0) return hipSuccess;
    if (pl.DT == 5) return launch_e_t<5, 5>(p, pl, with_dw, mma, st);
    return launch_e_t<4, 4>(p, pl, with_dw, mma, st);
}

}  // namespace nrm
