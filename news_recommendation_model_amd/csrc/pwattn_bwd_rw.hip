// Pointwise history attention, backward of the bilinear term for the bf16 matrix-core arithmetics (NRM_MMA_BF16 /
// NRM_MMA_BF16X3), "resident W_p" form.  gfx950 / MI355X only.
//
//     dP[m,d]   = sum_k dz[m,k] W_p[k,d]                          m = (b,t,h)
//     dt[b,t,d] += sum_h dP[m,d] h[b,h,d]         dh[b,h,d] += sum_t dP[m,d] t[b,t,d]
//
// The E-form kernels of pwattn_bwd.hip get dt and dh from two grouped contractions (2 x 2MD^2 FLOPs incl. dW_p) whose
// operands every wave tile re-loads and re-converts; with bf16 MFMAs (16x cheaper than fp32) those loads and conversions
// are what they spend their time on.  Here dP is ONE contraction with the forward kernel's structure (pwattn_fwd_rw.hip):
//   * the reduction index k is cut into slices whose W_p[k-slice, all d] image (bf16 hi [+ lo]) stays resident in LDS;
//     a PERSISTENT workgroup of 8 waves loads its slice once, then its waves work independently (no barrier, no DMA);
//   * every dz element is read exactly ONCE, by one lane, already split into bf16 hi/lo by the dz pass (NRM_DZ_HL4: each
//     aligned group of 4 fp32 values is stored as 4 bf16 hi + 4 bf16 lo, same 16 bytes), so the MFMA A operand is two
//     16-byte loads and no conversion;
//   * MFMA orientation: rows of dz are MFMA rows, d is the MFMA column: lane (r16, q) holds dP[row 4q+r][d(tile, r16)].
//     A wave owns (impression b, 16 history rows) and walks the candidates t: dh += dP * t[b,t,d] stays lane-local in
//     registers for the whole walk, dt = sum over the 16 rows is in-lane over r plus a reduce-scatter over q
//     (3 v_permlane*_swap): one 256-byte float-atomic row segment per 64 columns and step;
//   * d-column layout tile_col (as the E-form): tile it of 64-column group g holds d = 64g + 4*r16 + (it & 3), so one
//     16-byte load of a t / h row feeds four tiles and one dword per lane of the four q rows forms a contiguous segment.
// Supports D <= 256 (dh accumulators and the h multipliers of 16 column tiles live in registers); wider attentions keep the
// E-form.  dW_p comes from bwd_e_kernel<..., WITH_DT = false> (the (b,t)-grouped pass without its dt epilogue).
#include <cstdlib>
// timing diagnostics only (results are WRONG with any bit set; reported by nrm_build_flags): bit 0 no dt atomics, bit 1 no
// epilogue at all, bit 2 the dz operand is loaded once per task, bit 3 no dh flush at the end of a task
#ifndef NRM_DIAG_BRW
#define NRM_DIAG_BRW 0
#endif
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

#ifndef BRW_WAVES_N
#define BRW_WAVES_N 8
#endif
#ifndef BRW_RING
#define BRW_RING 4
#endif
#ifndef BRW_LA
#define BRW_LA 3
#endif
constexpr int BRW_WAVES = BRW_WAVES_N;   // waves per persistent workgroup (one workgroup per CU: the image takes up to 128 KB of LDS)
constexpr int BRW_LDS_BUDGET = 128 * 1024 + 16 * 1024;
constexpr int BRW_KC = 4;             // 32-wide reduction chunks per resident slice (bf16x3 at D = 256: 4 x 2 images x 256 rows x 64 B = 128 KB)

// image[c][img][row][32 bf16]: row = 16*it + r  <->  d = 64*(it >> 2) + 4*r + (it & 3);  16-byte slot s of the row (stored at
// slot position s ^ swz4(row), common.hpp) holds the reduction positions of lane quarter s: k = 32c + 8s + {0..7}.
__global__ void pack_wpt_bf16_kernel(const float* __restrict__ w, int ldw, int D, int rows, int k32, int nimg,
                                     __bf16* __restrict__ packed) {
    const long per_chunk = (long)rows * 32;
    const long total = (long)k32 * per_chunk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 31);
        const long rc = i >> 5;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int s = (j >> 3) ^ swz4(row);
        const int k = 32 * c + 8 * s + (j & 7);
        const int it = row >> 4, r = row & 15;
        const int d = 64 * (it >> 2) + 4 * r + (it & 3);
        const float v = (k < D && d < D) ? w[(long)k * ldw + d] : 0.0f;
        const __bf16 hi = (__bf16)v;
        const long o = (long)c * nimg * per_chunk + (long)row * 32 + j;
        packed[o] = hi;
        if (nimg > 1) packed[o + per_chunk] = (__bf16)(v - (float)hi);
    }
}

BwdRwPlan pwattn_bwd_rw_plan(int D, int mma) {
    BwdRwPlan pl = {};
    static const bool on = [] { const char* e = getenv("NRM_BWD_RW"); return !(e && e[0] == '0'); }();
    if (!on || mma == 0 || D <= 0 || D > 256 || D % 4) return pl;       // ng == 0: not supported
    pl.ng = (D + 63) / 64;
    pl.rows = pl.ng == 1 ? 64 : (pl.ng + 1) / 2 * 128;                   // whole d ranges of a wave task (two groups): zero rows behind D
    pl.k32 = (D + 31) / 32;
    pl.wimg = mma == 2 ? 2 : 1;
    pl.kc = BRW_KC;
    pl.nks = (pl.k32 + pl.kc - 1) / pl.kc;
    return pl;
}

long pwattn_bwd_rw_packed_floats(int D, int mma) {
    const BwdRwPlan pl = pwattn_bwd_rw_plan(D, mma);
    if (!pl.ng) return 0;
    return (long)pl.nks * pl.kc * pl.wimg * pl.rows * 16 + 256;          // whole slices (the tail slice's missing chunks are never read)
}

hipError_t pwattn_bwd_rw_pack_launch(const float* wp, int ldw, int D, int mma, float* packed, hipStream_t st) {
    const BwdRwPlan pl = pwattn_bwd_rw_plan(D, mma);
    if (!pl.ng) return hipErrorInvalidValue;
    const long total = (long)pl.k32 * pl.rows * 32;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wpt_bf16_kernel, dim3(blocks), dim3(256), 0, st, wp, ldw, D, pl.rows, pl.k32, pl.wimg,
                       reinterpret_cast<__bf16*>(packed));
    return hipGetLastError();
}

// NG = 64-column groups of d one wave task covers (the image holds all pl.ng groups: a task picks its range);
// EXACT: D % (32 * KC) == 0 -- every slice is full and no chunk is ragged: one lane offset with immediate chunk offsets, no
// per-chunk branches (the whole step is one basic block: W fragments are prefetched across chunk boundaries)
template <int NG, int MMA, bool EXACT>
__global__ __launch_bounds__(BRW_WAVES * 64, BRW_WAVES / 4) void bwd_dp_rw_kernel(const BwdRwParams p, const BwdRwPlan pl, int wgs_per_slice) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WIMG = MMA == 2 ? 2 : 1;
    constexpr int KC = BRW_KC;
    constexpr int NT = NG * 4;
    extern __shared__ __attribute__((aligned(16))) float wres[];         // [KC][WIMG][ROWS][16 floats = 32 bf16] | per-wave bounce [4][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwave = blockDim.x >> 6;                                   // <= BRW_WAVES (the launcher may start fewer)
    const int r16 = lane & 15, q = lane >> 4;
    const int T = p.T, H = p.H, D = p.D;
    const int ROWS = pl.rows;                                            // image rows = 64 * pl.ng
    const int ks = blockIdx.x % pl.nks, wg = blockIdx.x / pl.nks;
    const int c0 = ks * KC;                                              // first reduction chunk of the slice
    const int kc_here = EXACT ? KC : min(KC, pl.k32 - c0);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wimg), 0, p.w_bytes, 0x00020000);

    // ---- once per workgroup: the slice's image -> LDS in 1-KiB pieces of 16 rows, then ONE barrier
    const int npiece = kc_here * WIMG * (ROWS / 16);
    for (int pc = wave; pc < npiece; pc += nwave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wres + pc * 256), 16, lane * 16,
                                                 (c0 * WIMG * ROWS + pc * 16) * 64, 0, 0);
    __syncthreads();
    float* bounce = wres + KC * WIMG * ROWS * 16 + wave * 256;           // wave-private [4 rows][64 columns]

    const int rslot = 4 * (q ^ swz4(r16));
    const int nht = (H + 15) >> 4;
    const int ngs = ROWS / (64 * NG);                                    // d ranges (the image is padded to whole ranges: zero rows)
    const int tsplit = p.tsplit, tlen = (T + tsplit - 1) / tsplit;
    const int ntask = p.B * nht * tsplit * ngs;
    const bool plain_dh = pl.nks == 1 && tsplit == 1;                    // one task owns its dh rows outright: no atomics

    // task = (((b * nht + ht) * tsplit + tp) * ngs + gsel): the d ranges of one (b, rows, candidates) are neighbouring tasks, i.e.
    // neighbouring waves of one workgroup at the same time -- the second reader of a dz row finds it in L1 / L2
    for (int task = wg * nwave + wave; task < ntask; task += wgs_per_slice * nwave) {
        const int gsel = task % ngs;
        int rest = task / ngs;
        const int tp = rest % tsplit; rest /= tsplit;
        const int b = rest / nht, h0 = (rest - b * nht) * 16;
        const int g_base = gsel * NG;
        const int t_lo = tp * tlen, t_hi = min(T, t_lo + tlen);
        if (t_lo >= t_hi) continue;
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dz) + (size_t)b * T * H * D, 0, (unsigned)((size_t)T * H * D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.t) + (size_t)b * T * D, 0, (unsigned)(T * D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h) + (size_t)b * H * D, 0, (unsigned)(H * D * 4), 0x00020000);

        // this lane's dz row (h0 + r16) and its 8 reduction positions k = 32c + 8q + {0..7}: two hl4 units per chunk.
        // Units at k >= D (ragged last chunk) are masked: the bytes behind a row's end belong to the next row.
        const bool rok = h0 + r16 < H;
        const int zrow = ((h0 + r16) * D + 32 * c0 + 8 * q) * 4;        // byte offset inside candidate 0; + t * H*D*4 per step
        unsigned vz0[KC], vz1[KC];
        const unsigned vzx = rok ? (unsigned)zrow : OOB;
        if (!EXACT) {
#pragma unroll
            for (int cl = 0; cl < KC; ++cl) {
                const int k = 32 * (c0 + cl) + 8 * q;
                vz0[cl] = (rok && cl < kc_here && k < D) ? (unsigned)(zrow + cl * 128) : OOB;
                vz1[cl] = (rok && cl < kc_here && k + 4 < D) ? (unsigned)(zrow + cl * 128 + 16) : OOB;
            }
        }
        const int zstep = H * D * 4;
        const int dbase = 64 * g_base + 4 * r16;                         // first d of this lane's 16-byte column unit (group 0 of the task)
        // multipliers of the dt reduction, h[b, h0 + 4q + r, d units of the lane]: fixed for the whole walk, kept in registers
        // (16 per group).  Loading them per step put ten L2 round trips behind every step's float atomics: vmcnt retires in issue
        // order, so a load issued after the atomics waits for them, and a wait for it also waits for the dz prefetch in between.
        f32x4 hreg[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = h0 + 4 * q + r < H && dbase + 64 * g < D;
                hreg[g][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_h, ok ? (unsigned)(((h0 + 4 * q + r) * D + dbase + 64 * g) * 4) : OOB, 0, 0));
            }

        f32x4 dhacc[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) dhacc[it] = f32x4{0.f, 0.f, 0.f, 0.f};

        // dz operand ring: chunk cl of a step lives in slot cl % RING and is requested RING chunks ahead of its use, right after the
        // MFMAs that read the slot (KC % RING == 0 keeps every slot index a compile-time constant)
        constexpr int RING = BRW_RING;
        static_assert(KC % RING == 0, "slot = chunk % RING must not depend on the step");
        u32x4 cur0[RING], cur1[RING];
        auto load_chunk = [&](int slot, int t, int cl) {
            if (EXACT) {
                cur0[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, vzx, t * zstep + cl * 128, 0);
                cur1[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, vzx, t * zstep + cl * 128 + 16, 0);
            } else {
                cur0[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, vz0[cl], t * zstep, 0);
                cur1[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, vz1[cl], t * zstep, 0);
            }
        };
#pragma unroll
        for (int j = 0; j < RING; ++j) load_chunk(j, t_lo, j);

        const float* wbase = wres + (64 * g_base + r16) * 16 + rslot;    // this lane's fragment slot of tile 0 of the task's d range
        // W fragment n = cl * NT + it of the step, image img
        auto rdw = [&](int n, int img) {
            return *reinterpret_cast<const bf16x8*>(&wbase[((n / NT) * WIMG + img) * ROWS * 16 + (n % NT) * 256]);
        };
        for (int t = t_lo; t < t_hi; ++t) {
            // the candidate's t row (multipliers of the dh accumulation): requested FIRST, i.e. in front of the dz prefetch of this
            // step in vmcnt order, consumed in the epilogue
            f32x4 t4s[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                t4s[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_t, dbase + 64 * g < D ? (unsigned)((dbase + 64 * g) * 4) : OOB, t * D * 4, 0));
            f32x4 acc[NT];
#pragma unroll
            for (int it = 0; it < NT; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            // W fragments are read LA tiles ahead of their MFMAs (a tile's 1 | 3 MFMAs issue in 16 | 48 cycles, an LDS read under
            // load takes 100+); the scheduling barriers keep hipcc from hoisting every read in front of the first MFMA.  EXACT: the
            // look-ahead runs across chunk boundaries (one basic block per step); otherwise per chunk, inside the chunk's branch.
            constexpr int NF = EXACT ? KC * NT : NT;
            constexpr int LA = BRW_LA < NF ? BRW_LA : NF;
            bf16x8 whi[NF], wlo[NF];
            if (EXACT) {
#pragma unroll
                for (int n = 0; n < LA; ++n) { whi[n] = rdw(n, 0); if (MMA == 2) wlo[n] = rdw(n, 1); }
            }
#pragma unroll
            for (int cl = 0; cl < KC; ++cl) {
                const int slot = cl % RING;
                if (EXACT || cl < kc_here) {                             // (not EXACT: chunks behind the tail slice's end hold no image)
                    const bf16x8 ahi = __builtin_bit_cast(bf16x8, u32x4{cur0[slot][0], cur0[slot][1], cur1[slot][0], cur1[slot][1]});
                    const bf16x8 alo = __builtin_bit_cast(bf16x8, u32x4{cur0[slot][2], cur0[slot][3], cur1[slot][2], cur1[slot][3]});
                    const int n0 = EXACT ? cl * NT : 0;                  // index of the chunk's first fragment in whi / wlo
                    if (!EXACT) {
#pragma unroll
                        for (int n = 0; n < LA; ++n) { whi[n] = rdw(cl * NT + n, 0); if (MMA == 2) wlo[n] = rdw(cl * NT + n, 1); }
                    }
#pragma unroll
                    for (int it = 0; it < NT; ++it) {
                        const int n = n0 + it;
                        if (n + LA < NF) { whi[n + LA] = rdw(cl * NT + it + LA, 0); if (MMA == 2) wlo[n + LA] = rdw(cl * NT + it + LA, 1); }
                        if (MMA == 2) {
                            acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, whi[n], acc[it], 0, 0, 0);
                            acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, wlo[n], acc[it], 0, 0, 0);
                        }
                        acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, whi[n], acc[it], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // the slot is free: request chunk cl + RING (same slot), of this step or the next
                const int jn = cl + RING;
                const int tn = t + jn / KC;
                if (tn < t_hi && !(NRM_DIAG_BRW & 4)) load_chunk(slot, tn, jn % KC);
            }
            // ---- epilogue: dh (lane-local) and dt (reduce over the 16 rows, one atomic 256-byte segment per group)
            if (NRM_DIAG_BRW & 2) {                                      // keep the accumulators alive, nothing else
#pragma unroll
                for (int it = 0; it < NT; ++it) asm volatile("" :: "v"(acc[it]));
                continue;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 t4 = t4s[g];
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = acc[4 * g + j];
                    x[j] = fmaf(a[0], hreg[g][0][j], fmaf(a[1], hreg[g][1][j], fmaf(a[2], hreg[g][2][j], a[3] * hreg[g][3][j])));
                    dhacc[4 * g + j] += a * t4[j];
                }
                // reduce-scatter over the four 16-lane rows: row q ends up with the total of x[q]
                const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0]), __float_as_uint(x[2]), false, false);
                const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[1]), __float_as_uint(x[3]), false, false);
                const float y02 = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
                const float y13 = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y02), __float_as_uint(y13), false, false);
                const float val = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                const int d = dbase + 64 * g + q;
                if (d < D && !(NRM_DIAG_BRW & 1)) atomicAdd(p.dt + ((size_t)b * T + t) * D + d, val);
                if (NRM_DIAG_BRW & 1) asm volatile("" :: "v"(val));
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- end of the walk: dh[b, h0 + 4q + r, dbase + 64g + j] += dhacc[4g + j][r].  One owner (single slice, unsplit walk): a
        // 16-byte read-modify-write per lane.  Otherwise float atomics, shaped as whole 256-byte row segments: the 16-byte units
        // of a 4-row block go through the wave's LDS bounce and come back one dword per lane in column order.
        if (NRM_DIAG_BRW & 8) {
#pragma unroll
            for (int it = 0; it < NT; ++it) asm volatile("" :: "v"(dhacc[it]));
            continue;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = h0 + 4 * q + r;
                const f32x4 add = f32x4{dhacc[4 * g][r], dhacc[4 * g + 1][r], dhacc[4 * g + 2][r], dhacc[4 * g + 3][r]};
                if (plain_dh) {
                    if (row < H && dbase + 64 * g < D) {
                        float* dst = p.dh + ((size_t)b * H + row) * D + dbase + 64 * g;
                        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + add;
                    }
                } else {
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    __builtin_amdgcn_wave_barrier();
                    *reinterpret_cast<f32x4*>(&bounce[q * 64 + 4 * r16]) = add;          // bounce[row block q][column 4 r16 .. +3]
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    __builtin_amdgcn_wave_barrier();
                    const int dcol = 64 * (g_base + g) + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                                          // row block i: history row h0 + 4 i + r
                        const float v = bounce[i * 64 + lane];
                        const int rw = h0 + 4 * i + r;
                        if (rw < H && dcol < D) atomicAdd(p.dh + ((size_t)b * H + rw) * D + dcol, v);
                    }
                }
            }
        }
    }
#endif
}

int pwattn_bwd_rw_diag_flags() { return NRM_DIAG_BRW ? 512 : 0; }

static int brw_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
}

template <int NG>
static hipError_t launch_brw(BwdRwParams p, const BwdRwPlan& pl, int mma, hipStream_t st) {
    const int nht = (p.H + 15) / 16, ngs = pl.rows / (64 * NG);
    const long base = (long)p.B * nht * ngs;
    if (base <= 0) return hipSuccess;
    int wgs = brw_cus() / pl.nks;
    if (wgs < 1) wgs = 1;
    // the candidate walk is cut into tsplit parts when the tasks would not fill the chip's wave slots twice over (every part adds
    // its dh with float atomics: C2, 2 rounds of tasks: 0.191 ms unsplit, 0.201 ms in two parts), as long as a part keeps >= 8 steps
    int nw = BRW_WAVES;
    if (const char* e = getenv("NRM_BRW_WAVES")) { nw = atoi(e); if (nw < 1 || nw > BRW_WAVES) nw = BRW_WAVES; }
    const long slots = (long)wgs * nw;
    int tsplit = 1;
    if (const char* e = getenv("NRM_BRW_TSPLIT")) tsplit = atoi(e);
    else while (base * tsplit < 2 * slots && p.T / (tsplit + 1) >= 8) ++tsplit;
    if (tsplit < 1) tsplit = 1;
    if (tsplit > p.T) tsplit = p.T;
    p.tsplit = tsplit;
    const long ntask = base * tsplit;
    if ((long)wgs * nw > ntask) wgs = (int)((ntask + nw - 1) / nw);
    const size_t shm = (size_t)pl.kc * pl.wimg * pl.rows * 64 + (size_t)BRW_WAVES * 1024;     // image + per-wave bounce
    const dim3 grid((unsigned)(wgs * pl.nks)), block(nw * 64);
#define NRM_BRW(M_)                                                                                                          \
    {                                                                                                                        \
        auto k = (p.D % (32 * BRW_KC) == 0) ? bwd_dp_rw_kernel<NG, M_, true> : bwd_dp_rw_kernel<NG, M_, false>;                         \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BRW_LDS_BUDGET); \
        if (e != hipSuccess) return e;                                                                                       \
        hipLaunchKernelGGL(k, grid, block, shm, st, p, pl, wgs);                                                             \
    }
    if (mma == 2) NRM_BRW(2) else NRM_BRW(1)
#undef NRM_BRW
    return hipGetLastError();
}

hipError_t pwattn_bwd_rw_launch(const BwdRwParams& p, int mma, hipStream_t st) {
    const BwdRwPlan pl = pwattn_bwd_rw_plan(p.D, mma);
    if (!pl.ng) return hipErrorInvalidValue;
    if (pl.ng == 1) return launch_brw<1>(p, pl, mma, st);
    return launch_brw<2>(p, pl, mma, st);                               // two 64-column groups of d per wave task
}

}  // namespace nrm
