// Pointwise history attention, backward.  gfx950 / MI355X only.
//
// Forward (pwattn_fwd.hip):  z[m,k] = u[b,h,k] + v[b,t,k] + sum_d W_p[k,d] t[b,t,d] h[b,h,d],
//                            s[m]   = b2 + sum_k w2[k] gelu(z[m,k]),      m = (b,t,h)
// Given ds[m] = dL/ds[m] and the saved z:
//   (1) bwd_dz_kernel      dz[m,k] = ds[m] * w2[k] * gelu'(z[m,k])   (in place over z)
//                          dw2[k] += sum_m ds[m] * gelu(z[m,k])
//   (2) bwd_e_kernel       for every group g (one (b,t) in pass 1, one (b,h) in pass 2):
//                              E_g[k,d] = sum_r X_g[r,k] * Y_g[r,d]
//                          pass 1: X_g = dz[b,t,:,:] (r = h), Y_g = h[b]    -> dt[b,t,d] = sum_k W_p[k,d] E_g[k,d]
//                                                                             dW_p[k,d] += E_g[k,d] * t[b,t,d]
//                          pass 2: X_g = dz[b,:,h,:] (r = t), Y_g = t[b]    -> dh[b,h,d] = sum_k W_p[k,d] E_g[k,d]
//      Both passes together are exactly the 2x-forward FLOPs of a GEMM backward, but every output row is
//      produced by one workgroup (plain stores, no float atomics on [B,T,D]/[B,H,D]) and dW_p is
//      accumulated in registers across the groups a wave walks (one partial slab per split, summed after).
//   du = sum_t dz, dv = sum_h dz and the side-projection gradients are plain reductions/GEMMs done by the
//   caller.
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

// ---------------------------------------------------------------------------------------------
// (1) elementwise: dz in place + dw2 partial sums.  blockDim = (64, 4): x walks float4 columns, y rows.
constexpr int DZ_MAXC = 4;      // float4 columns per thread  -> D <= 1024

__global__ __launch_bounds__(256) void bwd_dz_kernel(float* __restrict__ z, const float* __restrict__ ds,
                                                     const float* __restrict__ w2, float* __restrict__ dw2,
                                                     long M, int D) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int C4 = D >> 2;
    f32x4 w[DZ_MAXC], acc[DZ_MAXC];
#pragma unroll
    for (int j = 0; j < DZ_MAXC; ++j) {
        const int c = tx + 64 * j;
        w[j] = c < C4 ? *reinterpret_cast<const f32x4*>(w2 + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 4 + ty; row < M; row += (long)gridDim.x * 4) {
        const float g = ds[row];
        float* zr = z + row * D;
#pragma unroll
        for (int j = 0; j < DZ_MAXC; ++j) {
            const int c = tx + 64 * j;
            if (c < C4) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 4 * c);
                f32x4 dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const GeluParts gp = gelu_parts(zz[e]);
                    acc[j][e] = fmaf(g * zz[e], gp.cdf, acc[j][e]);                       // ds * gelu(z)
                    dz[e] = g * w[j][e] * fmaf(zz[e] * 0.39894228040143267794f, gp.e, gp.cdf);
                }
                *reinterpret_cast<f32x4*>(zr + 4 * c) = dz;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DZ_MAXC; ++j) {
        const int c = tx + 64 * j;
        red[ty][tx] = acc[j];
        __syncthreads();
        if (ty == 0 && c < C4) {
            const f32x4 sum = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dw2 + 4 * c + e, sum[e]);
        }
        __syncthreads();
    }
}

hipError_t bwd_dz_launch(float* z, const float* ds, const float* w2, float* dw2, long M, int D, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    long nb = (M + 3) / 4;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bwd_dz_kernel, dim3((unsigned)nb), dim3(64, 4), 0, st, z, ds, w2, dw2, M, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// (2) grouped contraction E_g = X_g^T Y_g with fused epilogue.
// MFMA orientation: row index i = d (A operand = Y), column index j = k (B operand = X), so that a lane
// holds 4 consecutive d of one k: float4 loads of W_p[k, d..d+3], float4 stores of dW_p.
// Every WAVE is an independent task (d-column, k-range, split): 4 tasks per 256-thread workgroup, no
// workgroup barrier.  A wave owns a (DT*16 d) x (KT*16 k) tile of E/dW_p, walks the groups of its split,
// and adds its k-partial of out[g, d-range] with float atomics shaped as contiguous segments
// (ntasks_k adds per output element; `out` must be initialised by the caller).
//
// KS < KT splits the wave's k-range into two sub-passes per group ([0,KS) then [KS,KT)) so that only
// DT*KS accumulator tiles of E are live beside the DT*KT tiles of dW_p (register budget: 2 waves/SIMD).
template <int N> struct IC { static constexpr int value = N; };

template <int DT, int KT, int KS, bool WITH_DW>
__global__ __launch_bounds__(256, 2) void bwd_e_kernel(const BwdEParams p) {
    static_assert(KS <= KT && KT - KS <= KS, "first sub-pass must be the larger one");
    __shared__ __attribute__((aligned(16))) float bounce[4][DT * 16];    // wave-private, no barrier needed
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int D = p.D, R = p.R;

    const long task = (long)blockIdx.x * 4 + wave;
    if (task >= p.ntasks) return;
    const int kw = (int)(task % p.nkw);
    const int dcol = (int)((task / p.nkw) % p.ndcol);
    const int split = (int)(task / ((long)p.nkw * p.ndcol));
    const int d0 = dcol * (DT * 16);
    const int k0 = kw * (KT * 16);

    const int g_lo = split * p.gps;
    const int g_hi = min(p.G, g_lo + p.gps);

    // per-lane operand columns; invalid columns read column 0 and are zeroed
    // (loads are unconditional + AND-masked: a select would let hipcc sink each load into an exec-masked
    //  branch followed by vmcnt(0), which serialises the operand prefetch)
    int offA[DT], offB[KT];
    unsigned mA[DT], mB[KT];
#pragma unroll
    for (int it = 0; it < DT; ++it) { const int d = d0 + 16 * it + r16; mA[it] = d < D ? 0xffffffffu : 0u; offA[it] = d < D ? d : 0; }
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) { const int k = k0 + 16 * jt + r16; mB[jt] = k < D ? 0xffffffffu : 0u; offB[jt] = k < D ? k : 0; }

    f32x4 dW[DT][KT];
    if (WITH_DW) {
#pragma unroll
        for (int it = 0; it < DT; ++it)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) dW[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    float an[DT], bn[KS];       // raw (unmasked) operands of the pending step
    unsigned rmn = 0u;
    // operands of reduction rows r0..r0+3 of group gg, k tiles [J0, J0+NJ)
    auto load_step = [&](auto j0c, auto njc, int gg, int r0) {
        constexpr int J0 = decltype(j0c)::value, NJ = decltype(njc)::value;
        const int g1 = gg / p.G2, g2 = gg - g1 * p.G2;
        const int r = r0 + q;
        rmn = r < R ? 0xffffffffu : 0u;            // row mask of the pending step, applied by its consumer
        const long rr = r < R ? r : 0;
        const float* xp = p.X + (long)g1 * p.xs1 + (long)g2 * p.xs2 + rr * p.xrs;
        const float* yp = p.Y + (long)g1 * p.ys1 + rr * p.yrs;
#pragma unroll
        for (int it = 0; it < DT; ++it) an[it] = yp[offA[it]];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) bn[jt] = xp[offB[J0 + jt]];
    };

    // one sub-pass of group g over k tiles [J0, J0+NJ); prefetches the first step of the following
    // sub-pass (tiles [NJ0, NJ0+NNJ) of group gn) under its last MFMAs.
    auto sub_pass = [&](auto j0c, auto njc, auto nj0c, auto nnjc, int g, int gn, bool first_of_group) {
        constexpr int J0 = decltype(j0c)::value, NJ = decltype(njc)::value;
        f32x4 E[DT][NJ];
#pragma unroll
        for (int it = 0; it < DT; ++it)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) E[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int r0 = 0; r0 < R; r0 += 4) {
            float a[DT], b[NJ];
#pragma unroll
            for (int it = 0; it < DT; ++it) a[it] = __uint_as_float(__float_as_uint(an[it]) & (mA[it] & rmn));
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) b[jt] = __uint_as_float(__float_as_uint(bn[jt]) & (mB[J0 + jt] & rmn));
            if (r0 + 4 < R) load_step(j0c, njc, g, r0 + 4);
            else if (gn < g_hi) load_step(nj0c, nnjc, gn, 0);
#pragma unroll
            for (int it = 0; it < DT; ++it)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) E[it][jt] = mfma16(a[it], b[jt], E[it][jt]);
        }

#pragma unroll
        for (int it = 0; it < DT; ++it) {
            const int dbase = d0 + 16 * it + 4 * q;
            int dd = dbase < D ? dbase : 0;              // D % 4 == 0
            // opaque to the optimiser: otherwise the DT*KT 64-bit W_p tile addresses are hoisted out of the
            // group loop and spilled (they are cheap to recompute: one mad + one 64-bit add)
            asm volatile("" : "+v"(dd));
            f32x4 sr = f32x4{0.f, 0.f, 0.f, 0.f};
            if (WITH_DW) sr = *reinterpret_cast<const f32x4*>(p.srow + (long)g * p.lds_ + dd);
            f32x4 dto = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                // E is exactly 0 wherever k or d is out of range (zero operands), so clamped loads are safe
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.wp + (offB[J0 + jt] * p.ldwp + dd));
                dto += w4 * E[it][jt];
                if (WITH_DW) dW[it][J0 + jt] += E[it][jt] * sr;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dto[e] = wave_sum16(dto[e]);
            if (r16 == 0) {
                f32x4* bp = reinterpret_cast<f32x4*>(&bounce[wave][16 * it + 4 * q]);
                *bp = first_of_group ? dto : (*bp + dto);
            }
            // keep hipcc from hoisting all DT*NJ W_p loads to the top of the epilogue (4 registers each)
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (g_lo < g_hi) load_step(IC<0>{}, IC<KS>{}, g_lo, 0);
    for (int g = g_lo; g < g_hi; ++g) {
        if constexpr (KS == KT) {
            sub_pass(IC<0>{}, IC<KT>{}, IC<0>{}, IC<KT>{}, g, g + 1, true);
        } else {
            sub_pass(IC<0>{}, IC<KS>{}, IC<KS>{}, IC<KT - KS>{}, g, g, true);
            sub_pass(IC<KS>{}, IC<KT - KS>{}, IC<0>{}, IC<KS>{}, g, g + 1, false);
        }
        // k-partial of out[g, d0 .. d0+DT*16): one dword per lane, contiguous segments.
        // bounce[] is wave-private: LDS ops of one wave complete in order, the fences only pin hipcc.
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        float* orow = p.out + (long)g * p.ldo + d0;
#pragma unroll
        for (int c = 0; c < DT * 16; c += 64) {
            const int dl = c + lane;
            if (dl < DT * 16 && d0 + dl < D) atomicAdd(orow + dl, bounce[wave][dl]);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
    }

    if (WITH_DW) {
        float* wsp = p.ws + (long)split * D * D;
#pragma unroll
        for (int it = 0; it < DT; ++it) {
            const int dbase = d0 + 16 * it + 4 * q;
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) {
                const int k = k0 + 16 * jt + r16;
                if (k < D && dbase < D) *reinterpret_cast<f32x4*>(wsp + (long)k * D + dbase) = dW[it][jt];
            }
        }
    }
}


BwdEPlan bwd_e_plan(int D, int G, int target_waves) {
    const int n16 = (D + 15) / 16;
    const int c5 = (n16 + 4) / 5 * 5, c4 = (n16 + 3) / 4 * 4;
    BwdEPlan pl;
    if (c5 < c4) { pl.DT = 5; pl.KT = 5; } else { pl.DT = 4; pl.KT = 4; }   // the shape that pads less
    pl.ndcol = (n16 + pl.DT - 1) / pl.DT;
    pl.nkw = (n16 + pl.KT - 1) / pl.KT;
    const int tiles = pl.ndcol * pl.nkw;
    int ns = (target_waves + tiles - 1) / tiles;
    if (ns < 1) ns = 1;
    if (ns > G) ns = G > 0 ? G : 1;
    pl.gps = (G + ns - 1) / ns;
    pl.nsplit = pl.gps > 0 ? (G + pl.gps - 1) / pl.gps : 1;
    return pl;
}

template <int DT, int KT>
static hipError_t launch_e_t(BwdEParams p, const BwdEPlan& pl, bool with_dw, hipStream_t st) {
    p.nkw = pl.nkw; p.ndcol = pl.ndcol; p.gps = pl.gps;
    p.ntasks = (long)pl.nkw * pl.ndcol * pl.nsplit;
    const long nblk = (p.ntasks + 3) / 4;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    constexpr int KS_DW = (DT * KT > 16) ? (KT + 1) / 2 : KT;     // 5x5 with dW: sub-passes of 3 + 2 tiles
    if (with_dw) hipLaunchKernelGGL((bwd_e_kernel<DT, KT, KS_DW, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else         hipLaunchKernelGGL((bwd_e_kernel<DT, KT, KT, false>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t bwd_e_launch(const BwdEParams& p, const BwdEPlan& pl, bool with_dw, hipStream_t st) {
    if (p.G <= 0) return hipSuccess;
    if (pl.DT == 5) return launch_e_t<5, 5>(p, pl, with_dw, st);
    return launch_e_t<4, 4>(p, pl, with_dw, st);
}

}  // namespace nrm
