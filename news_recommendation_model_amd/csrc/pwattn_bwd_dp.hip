// Pointwise history attention, backward of the bilinear term in fp32, "dP walk" form (round 5).  gfx950 / MI355X only.
//
//     dP[m,d]   = sum_k dz[m,k] W_p[k,d]                          m = (b,t,h)
//     dt[b,t,d] += sum_h dP[m,d] h[b,h,d]         dh[b,h,d] += sum_t dP[m,d] t[b,t,d]
//
// The E-form kernels of pwattn_bwd.hip get dt and dh from two grouped contractions, (b,t) groups for dt + dW_p and (b,h) groups
// for dh: 3 x 2MD^2 FLOPs for dt, dh and dW_p, the (b,t) pass with its serial dt epilogue at 250 VGPRs (0.71 of the fp32 MFMA
// peak at C3).  Here dt and dh come from ONE contraction dP = dz W_p with the forward kernel's skeleton (pwattn_fwd.hip: the
// W operand streamed through a two-deep LDS ring by LDS-DMA in K-chunks of 16, one barrier per chunk), and dW_p from the
// (b,t) pass without its epilogue (bwd_e_kernel<..., WITH_DT = false>, 0.78-0.82 of peak): the same FLOPs, the laggard gone.
//
//   * rows: the history rows r = (b,h) of the whole batch, flattened.  A workgroup of 4 waves owns 64 consecutive rows, a wave 16
//     of them, and WALKS the candidates t: step (t) is the 16 x (16 NT) tile of dP for dz rows m = (b, t, h) of its 16 (b,h) rows;
//   * MFMA orientation: dz rows are MFMA rows, d is the MFMA column: lane (c16, q) holds dP[row 4q + r][d(tile, c16)], r = 0..3.
//     dh += dP * t[b,t,d] stays lane-local in registers for the whole walk (flushed with float atomics when the wave moves to
//     other rows / columns), dt = sum over the 16 rows is in-lane over r plus a reduce-scatter over q (3 v_permlane*_swap):
//     one 256-byte float-atomic row segment per 64 columns and step -- the epilogue of pwattn_bwd_rw.hip (bf16 forms);
//   * 16 flattened rows belong to at most two impressions (H >= 16): every step forms the sums of both (segment A: the
//     impression of the wave's first row, segment B: the next one) with masked multipliers;
//   * d-column layout inside an N-chunk of 16 NT columns: tile it < 4 (NT / 4) of 64-column group g = it >> 2 holds
//     d = 64 g + 4 c16 + (it & 3) (one 16-byte load of a t / h row feeds four tiles, and one dword per lane of the four q
//     rows forms a contiguous 256-byte segment); the NT % 4 tiles behind the last whole group hold d = 64 (NT / 4) + 16 i + c16;
//   * work = (row block, N-chunk, candidate) steps in that order, cut into EQUAL contiguous ranges over a grid of four rounds of
//     workgroups (3 per CU): the tail is a sixteenth of a range, and the LDS ring runs on across steps (the first chunk of
//     step s + 1 is requested under the last chunk of step s).
#include <cstdlib>
// timing diagnostics only (results are WRONG with any bit set; reported by nrm_build_flags): bit 0 no dz DMA after a step's first
// chunk, bit 1 no epilogue, bit 2 no W DMA after a step's first chunk
#ifndef NRM_DIAG_DP
#define NRM_DIAG_DP 0
#endif
#ifndef DP_WPE
#define DP_WPE 3          // tuning: launch bound, waves per SIMD
#endif
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

// column of output tile `it`, MFMA column i, inside an N-chunk of NT tiles (see above)
__host__ __device__ inline int dp_tile_col(int NT, int it, int i) {
    const int ng4 = NT >> 2;
    return it < 4 * ng4 ? 64 * (it >> 2) + 4 * i + (it & 3) : 64 * ng4 + 16 * (it - 4 * ng4) + i;
}

// packed[c][row][16]: row = nc * 16 NT + 16 it + i  <->  d = nc * 16 NT + dp_tile_col(it, i);  the 16 floats are the reduction
// positions k = 16 c + 4 (slot ^ swz4(row)) + (j & 3) of W_p[k][d] (zero behind D): the forward's LDS-DMA image of W_p^T.
__global__ void pack_wpt_f32_kernel(const float* __restrict__ w, int ldw, int D, int NT, int rows, int kchunks,
                                    float* __restrict__ packed) {
    const long total = (long)kchunks * rows * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const long rc = i >> 4;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int slot = (j >> 2) ^ swz4(row);
        const int k = c * 16 + 4 * slot + (j & 3);
        const int nc = row / (16 * NT), rr = row - nc * 16 * NT;
        const int d = nc * 16 * NT + dp_tile_col(NT, rr >> 4, rr & 15);
        packed[i] = (k < D && d < D) ? w[(long)k * ldw + d] : 0.0f;
    }
}

BwdDpPlan pwattn_bwd_dp_plan(int D, int H) {
    BwdDpPlan pl = {};
    if (D <= 0 || D % 4 || H < 16) return pl;                         // NT == 0: not supported (E-form)
    const int n16 = (D + 15) / 16;
    // accumulators of a step + the dh accumulators of the walk: 8 NT registers; NT <= 13 keeps three waves per SIMD
    int nch = (n16 + 12) / 13;
    int nt = (n16 + nch - 1) / nch;
    if (nt < 4) nt = 4;
    static const int kNT[4] = {4, 8, 12, 13};
    int sel = 3;
    for (int i = 0; i < 4; ++i) if (kNT[i] >= nt) { sel = i; break; }
    pl.NT = kNT[sel];
    pl.nchunks = (n16 + pl.NT - 1) / pl.NT;
    pl.rows = pl.nchunks * pl.NT * 16;
    pl.kchunks = n16;
    return pl;
}

long pwattn_bwd_dp_packed_floats(int D, int H) {
    const BwdDpPlan pl = pwattn_bwd_dp_plan(D, H);
    return pl.NT ? (long)pl.kchunks * pl.rows * 16 : 0;
}

hipError_t pwattn_bwd_dp_pack_launch(const float* wp, int ldw, int D, int H, float* packed, hipStream_t st) {
    const BwdDpPlan pl = pwattn_bwd_dp_plan(D, H);
    if (!pl.NT) return hipErrorInvalidValue;
    const long total = (long)pl.kchunks * pl.rows * 16;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wpt_f32_kernel, dim3(blocks), dim3(256), 0, st, wp, ldw, D, pl.NT, pl.rows, pl.kchunks, packed);
    return hipGetLastError();
}

template <int NT>
__global__ __launch_bounds__(256, DP_WPE) void bwd_dp_walk_kernel(const BwdDpParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 4, BM = 64;
    constexpr int WROWS = NT * 16;
    constexpr int NG4 = NT / 4, NREST = NT % 4;                         // whole 64-column groups / plain 16-column tiles behind them
    constexpr int BUF = (WROWS + BM) * 16;                                // one ring stage: [W chunk | dz rows], rows of 16 floats
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF + NW * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;
    const int T = p.T, H = p.H, D = p.D, R = p.B * p.H;
    const int KCH = p.kchunks, NCH = p.nchunks;
    float* bounce = smem + 2 * BUF + wave * 256;                          // wave-private [4 rows][64 columns]

    // this workgroup's range of steps s = ((rb * NCH + nc) * T + t)
    const long S = p.steps;
    const int g = blockIdx.x, G = gridDim.x;
    const int s_lo = (int)(S * g / G), s_hi = (int)(S * (g + 1) / G);
    if (s_lo >= s_hi) return;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wimg), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t), 0, p.t_bytes, 0x00020000);
    const int rslot = 4 * (q ^ swz4(c16));
    const int zstep = H * D * 4;                                          // bytes between candidates of one impression in dz

    // ---- state of the current (row block, N-chunk): set by enter()
    int rb = -1, nc = -1, colbase = 0;
    unsigned live = 0;                                                    // bit it: tile it holds columns < D
    __amdgpu_buffer_rsrc_t rs_z = rs_w;                                   // dz, based at the first impression of the row block
    unsigned voff_z = OOB;                                                // DMA: this lane's row (lane >> 2) and 16-byte slot
    unsigned voff_h4[4], voff_h1[4];                                      // multiplier rows 4q + r: 16-byte units / single columns
    unsigned voff_tA4 = OOB, voff_tB4 = OOB, voff_tA1 = OOB, voff_tB1 = OOB;
    bool isB[4];
    bool has_b = false;                                                   // (uniform) the wave's 16 rows reach into a second impression
    int b_lo = 0;                                                         // (uniform) impression of the wave's first row
    long rowA = 0;                                                        // (uniform) b_lo * T: dt row of segment A at t = 0

    f32x4 dhacc[NT];

    auto enter = [&](int s) {
        const int item = s / T;
        rb = item / NCH; nc = item - rb * NCH;
        colbase = nc * WROWS;
        live = 0;
#pragma unroll
        for (int it = 0; it < NT; ++it) if (colbase + dp_tile_col(NT, it, 0) < D) live |= 1u << it;
        const int r0 = rb * BM;
        const int b_first = r0 / H;
        const size_t imp = (size_t)T * H * D;                             // floats of one impression's dz block
        const int span = min(p.B - b_first, (BM - 1) / H + 2);
        rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz) + (size_t)b_first * imp, 0, (unsigned)(span * imp * 4), 0x00020000);
        {   // DMA lane: row rl of the wave's 16 rows
            const int rl = lane >> 2;
            const int r = r0 + wave * 16 + rl;
            const int b = r / H, hh = r - b * H;
            const unsigned slot = (unsigned)((lane & 3) ^ swz4(rl));
            voff_z = r < R ? (unsigned)(((size_t)(b - b_first) * T * H + hh) * D + 4 * slot) * 4u : OOB;
        }
        const int rw0 = r0 + wave * 16;                                   // uniform
        b_lo = rw0 / H;
        rowA = (long)b_lo * T;
        has_b = rw0 < R && min(rw0 + 15, R - 1) / H != b_lo;             // (a valid row of the next impression)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = rw0 + 4 * q + r;
            isB[r] = rr >= (b_lo + 1) * H;
            voff_h4[r] = rr < R ? (unsigned)(rr * D + 4 * c16) * 4u : OOB;
            voff_h1[r] = rr < R ? (unsigned)(rr * D + c16) * 4u : OOB;
        }
        voff_tA4 = rw0 < R ? (unsigned)(b_lo * T * D + 4 * c16) * 4u : OOB;
        voff_tA1 = rw0 < R ? (unsigned)(b_lo * T * D + c16) * 4u : OOB;
        voff_tB4 = has_b ? voff_tA4 + (unsigned)(T * D * 4) : OOB;
        voff_tB1 = has_b ? voff_tA1 + (unsigned)(T * D * 4) : OOB;
#pragma unroll
        for (int it = 0; it < NT; ++it) dhacc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // dh[row 4q + r, colbase + col(it, c16)] += dhacc[it][r]: whole groups as 256-byte row segments through the wave's bounce
    auto flush = [&]() {
        const int rw0 = rb * BM + wave * 16;
#pragma unroll
        for (int gg = 0; gg < NG4; ++gg) {
            if (!((live >> (4 * gg)) & 1)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 add = f32x4{dhacc[4 * gg][r], dhacc[4 * gg + 1][r], dhacc[4 * gg + 2][r], dhacc[4 * gg + 3][r]};
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<f32x4*>(&bounce[q * 64 + 4 * c16]) = add;          // bounce[row block q][column 4 c16 .. +3]
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                __builtin_amdgcn_wave_barrier();
                const int dcol = colbase + 64 * gg + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {                                          // row block i: row rw0 + 4 i + r
                    const float v = bounce[i * 64 + lane];
                    const int rw = rw0 + 4 * i + r;
                    if (rw < R && dcol < D) atomicAdd(p.dh + (size_t)rw * D + dcol, v);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NREST; ++i) {
            const int it = 4 * NG4 + i;
            if (!((live >> it) & 1)) continue;
            const int dcol = colbase + 64 * NG4 + 16 * i + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = rw0 + 4 * q + r;
                if (rw < R && dcol < D) atomicAdd(p.dh + (size_t)rw * D + dcol, dhacc[it][r]);
            }
        }
    };

    // K-chunk c of step (t) -> ring stage `buf`: every wave its own 16 dz rows (one 1-KiB piece) and every 4th piece of W
    auto dma_chunk = [&](int c, int t, float* buf) {
        const int wbase = (c * p.rows + nc * WROWS) * 64;                 // bytes, uniform
        if (!((NRM_DIAG_DP & 4) && c > 0))
        for (int pc = wave; pc < NT; pc += NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + pc * 256),
                                                     16, lane * 16, wbase + pc * 1024, 0, 0);
        if (!((NRM_DIAG_DP & 1) && c > 0))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, (__attribute__((address_space(3))) void*)(buf + (WROWS + wave * 16) * 16),
                                                 16, voff_z, t * zstep + c * 64, 0, 0);
    };

    enter(s_lo);
    dma_chunk(0, s_lo % T, smem);
    __syncthreads();
    int par = 0;
    for (int s = s_lo; s < s_hi; ++s) {
        const int t = s % T;
        f32x4 acc[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool more = s + 1 < s_hi;
        const bool same_item = (s + 1) % T != 0;                          // the next step stays in this (row block, N-chunk)
        for (int c = 0; c < KCH; ++c) {
            float* cur = smem + par * BUF;
            float* nxt = smem + (par ^ 1) * BUF;
            if (c + 1 < KCH) dma_chunk(c + 1, t, nxt);
            else if (more && same_item) dma_chunk(0, t + 1, nxt);         // the ring runs on across steps of one item
            {
                const f32x4 pf = *reinterpret_cast<const f32x4*>(&cur[(WROWS + wave * 16 + c16) * 16 + rslot]);   // dz[row c16][k = 16c + 4q + j]
                f32x4 af = *reinterpret_cast<const f32x4*>(&cur[c16 * 16 + rslot]);
#pragma unroll
                for (int it = 0; it < NT; ++it) {
                    f32x4 afn = af;
                    if (it + 1 < NT) afn = *reinterpret_cast<const f32x4*>(&cur[((it + 1) * 16 + c16) * 16 + rslot]);
                    if ((live >> it) & 1)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[it] = mfma16(pf[j], af[j], acc[it]);
                    __builtin_amdgcn_sched_barrier(0);
                    af = afn;
                }
            }
            __syncthreads();
            par ^= 1;
        }

        // ---- epilogue of the step: dh (lane-local) and dt (both segments; reduce over the 16 rows, one atomic row segment per group)
        if (NRM_DIAG_DP & 2) {                                            // keep the accumulators alive, nothing else
#pragma unroll
            for (int it = 0; it < NT; ++it) { asm volatile("" :: "v"(acc[it])); dhacc[it] += acc[it]; }
            continue;
        }
        const int soff_h = colbase * 4;
        const int soff_t = (t * D + colbase) * 4;
        float* dtA = p.dt + (rowA + t) * D + colbase;
        float* dtB = dtA + (size_t)T * D;
#pragma unroll
        for (int gg = 0; gg < NG4; ++gg) {
            if (!((live >> (4 * gg)) & 1)) continue;
            f32x4 h4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h4[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, voff_h4[r], soff_h + 256 * gg, 0));
            const f32x4 tA = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, voff_tA4, soff_t + 256 * gg, 0));
            f32x4 tB = tA;
            if (has_b) tB = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, voff_tB4, soff_t + 256 * gg, 0));
            float xA[4], xB[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = acc[4 * gg + j];
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = a[r] * h4[r][j];
                    if (has_b) { sa += isB[r] ? 0.f : pr; sb += isB[r] ? pr : 0.f; }
                    else sa += pr;
                    dhacc[4 * gg + j][r] = fmaf(a[r], (has_b && isB[r]) ? tB[j] : tA[j], dhacc[4 * gg + j][r]);
                }
                xA[j] = sa; xB[j] = sb;
            }
            // reduce-scatter over the four 16-lane rows: row q ends up with the total of x[q]
            auto rs4 = [](const float* x) {
                const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0]), __float_as_uint(x[2]), false, false);
                const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[1]), __float_as_uint(x[3]), false, false);
                const float y02 = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
                const float y13 = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y02), __float_as_uint(y13), false, false);
                return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            };
            const int dl = 64 * gg + 4 * c16 + q;
            const bool dok = colbase + dl < D;
            const float vA = rs4(xA);
            if (dok && voff_tA4 != OOB) atomicAdd(dtA + dl, vA);
            if (has_b) {
                const float vB = rs4(xB);
                if (dok) atomicAdd(dtB + dl, vB);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NREST; ++i) {
            const int it = 4 * NG4 + i;
            if (!((live >> it) & 1)) continue;
            const int so = (64 * NG4 + 16 * i) * 4;
            float h1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_h, voff_h1[r], soff_h + so, 0));
            const float tA = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, voff_tA1, soff_t + so, 0));
            float tB = tA;
            if (has_b) tB = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, voff_tB1, soff_t + so, 0));
            const f32x4 a = acc[it];
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = a[r] * h1[r];
                if (has_b) { sa += isB[r] ? 0.f : pr; sb += isB[r] ? pr : 0.f; }
                else sa += pr;
                dhacc[it][r] = fmaf(a[r], (has_b && isB[r]) ? tB : tA, dhacc[it][r]);
            }
            const int dl = 64 * NG4 + 16 * i + c16;
            const bool dok = colbase + dl < D && q == 0;
            const float vA = sum_rows4(sa);
            if (dok && voff_tA4 != OOB) atomicAdd(dtA + dl, vA);
            if (has_b) {
                const float vB = sum_rows4(sb);
                if (dok) atomicAdd(dtB + dl, vB);
            }
        }

        if (more && !same_item) {                                         // next (row block, N-chunk): hand the dh sums over, restart the ring
            flush();
            enter(s + 1);
            dma_chunk(0, 0, smem + par * BUF);
            __syncthreads();
        }
    }
    flush();
#endif
}

int pwattn_bwd_dp_diag_flags() { return NRM_DIAG_DP ? 1024 : 0; }

static int dp_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
}

template <int NT>
static hipError_t launch_dp(BwdDpParams p, hipStream_t st) {
    // four rounds of workgroups at three per CU (measured at C3, 768 / 1536 / 3072 workgroups: 3.97 / 3.95 / 3.92 ms -- the CUs do not
    // finish equal ranges at the same time); never fewer than 4 steps per workgroup (a range pays one ring start and one dh flush)
    long grid = 12L * dp_cus();
    if (const char* e = getenv("NRM_DP_GRID")) { const long v = atol(e); if (v > 0) grid = v; }
    if (grid > p.steps / 4) grid = p.steps / 4;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((bwd_dp_walk_kernel<NT>), dim3((unsigned)grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t pwattn_bwd_dp_launch(BwdDpParams p, hipStream_t st) {
    const BwdDpPlan pl = pwattn_bwd_dp_plan(p.D, p.H);
    if (!pl.NT) return hipErrorInvalidValue;
    const long R = (long)p.B * p.H;
    const long nrb = (R + 63) / 64;
    p.rows = pl.rows; p.kchunks = pl.kchunks; p.nchunks = pl.nchunks;
    p.steps = nrb * pl.nchunks * p.T;
    if (p.steps <= 0) return hipSuccess;
    if (p.steps > 0x7fffffffL) return hipErrorInvalidValue;
    switch (pl.NT) {
        case 4:  return launch_dp<4>(p, st);
        case 8:  return launch_dp<8>(p, st);
        case 12: return launch_dp<12>(p, st);
        case 13: return launch_dp<13>(p, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace nrm
