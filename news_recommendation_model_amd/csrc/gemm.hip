// fp32 MFMA GEMMs for the dense layers of the hot path (reference models/attention_model.py:29-32 MLP,
// models/user_invariant_interest_model.py:33,78 w1, the side projections of the attention).  gfx950 only.
//
//   gemm_nt   Y[M,N] = epi( X[M,K] * W^T )      W given as "packed rows" (pack_rows_kernel), X row-major.
//             Used for Linear.forward (rows of W = output features) and for dX = dY * W (rows = input
//             features, packed from W with swapped strides).  Same skeleton as pwattn_fwd_kernel: LDS-DMA
//             staging of a [rows x 16] weight chunk + the workgroup's X rows, double buffered, one barrier per
//             K-chunk; MFMA rows = output columns so a lane owns 4 consecutive outputs of one row (float4 I/O).
//             Epilogues: bias | bias+GELU (stores pre-activation and activation) | *gelu'(saved pre-activation).
//   gemm_tn   C[N,K] = sum_m A[m,N]^T B[m,K]     (dW = dY^T X, plus db = column sums of dY).
//             Reduction runs over the ROW index of both operands, so operands stream straight from global
//             memory into MFMA operand registers (one dword per lane per tile, ping-pong prefetch), the row
//             range is split over waves and every wave stores one partial slab (transposed: [K][N]).
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "gemm.hpp"
// timing diagnostics only (scripts/_diag): results are WRONG with any bit set -- 1: no barrier in the K loop of gemm_nt,
// 2: gemm_nt without its epilogue stores, 4: gemm_nt DMAs only its first K-chunk, 8: gemm_tn never reloads its operands
#ifndef NRM_PRIO
#define NRM_PRIO 0        // tuning: s_setprio level of the K loops (0: none)
#endif
#ifndef NRM_DIAG_GEMM
#define NRM_DIAG_GEMM 0
#endif

namespace nrm {

// packed[c][row][16]: element (row, 16c + 4*(s ^ swz4(row)) + e) of the logical [nrows x ncols] matrix
// src[row*rs + col*cs], zero padded to `rows` rows / 16*kchunks columns (slot swizzle: see pwattn_fwd.hip).
__global__ void pack_rows_kernel(const float* __restrict__ src, long rs, long cs, int nrows, int ncols,
                                 int rows, int kchunks, float* __restrict__ packed) {
    const long total = (long)kchunks * rows * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const long rc = i >> 4;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int slot = (j >> 2) ^ swz4(row);
        const int col = c * 16 + 4 * slot + (j & 3);
        packed[i] = (row < nrows && col < ncols) ? src[row * rs + col * cs] : 0.0f;
    }
}

// Several weight matrices in ONE launch (blockIdx.y = matrix; the table travels in the kernel arguments): every Linear of
// the model in both GEMM orientations, re-packed once per optimizer step instead of once per GEMM call (24 launches per
// step).  An entry may be the sum / difference of two sources: the attention's side projections use W_h - W_d and
// W_t + W_d (pwattn_fwd.hip), formed here instead of by separate elementwise launches.
__global__ __launch_bounds__(256) void pack_rows_multi_kernel(const PackTable tab) {
    const PackEntry e = tab.e[blockIdx.y];
    if (e.fmt) {
        // weight fragments of gemm_nt_rx: dst[it][c][img][r16][32 bf16], element (n = 16 it + r16, k = 32 c + j)
        const int nimg = e.fmt;
        const long total = (long)(e.rows / 16) * e.kchunks * 512;
        __bf16* dst = reinterpret_cast<__bf16*>(e.dst);
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int j = (int)(i & 31), r16 = (int)((i >> 5) & 15);
            const long tc = i >> 9;                                      // it * kchunks + c
            const int c = (int)(tc % e.kchunks), it = (int)(tc / e.kchunks);
            const int n = 16 * it + r16, k = 32 * c + j;
            float v = 0.0f;
            if (n < e.nrows && k < e.ncols) {
                const long o = n * e.rs + k * e.cs;
                v = e.src[o];
                if (e.src2) v = fmaf(e.sign2, e.src2[o], v);
            }
            const __bf16 hi = (__bf16)v;
            const long o = (tc * nimg) * 512 + r16 * 32 + j;
            dst[o] = hi;
            if (nimg > 1) dst[o + 512] = (__bf16)(v - (float)hi);
        }
        return;
    }
    const long total = (long)e.kchunks * e.rows * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const long rc = i >> 4;
        const int row = (int)(rc % e.rows);
        const int c = (int)(rc / e.rows);
        const int slot = (j >> 2) ^ swz4(row);
        const int col = c * 16 + 4 * slot + (j & 3);
        float v = 0.0f;
        if (row < e.nrows && col < e.ncols) {
            const long o = row * e.rs + col * e.cs;
            v = e.src[o];
            if (e.src2) v = fmaf(e.sign2, e.src2[o], v);
        }
        e.dst[i] = v;
    }
}

hipError_t pack_rows_multi_launch(const PackTable& tab, int n, long max_total, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    long bx = (max_total + 255) / 256;
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(pack_rows_multi_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, tab);
    return hipGetLastError();
}

hipError_t pack_rows_launch(const float* src, long rs, long cs, int nrows, int ncols, int rows, int kchunks,
                            float* packed, hipStream_t st) {
    const long total = (long)kchunks * rows * 16;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, st, src, rs, cs, nrows, ncols, rows, kchunks, packed);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// KC (round 5): 16-wide K-chunks per LDS stage.  A launch of few workgroups (the head of C1: 5120 rows = 80 row blocks) has one
// or two workgroups per CU, nothing co-resident to hide a stage's DMA latency behind, and with one chunk per stage a chain of
// K/16 rounds of (DMA ~1.5 us  ||  52 MFMAs = 0.7 us) + barrier: latency-bound at a third of the MFMA rate.  KC = 2 / 4 puts two /
// four consecutive chunks into one stage (the packed operand and the X rows are already laid out chunk after chunk): a quarter
// of the rounds and barriers, 70 / 139 KB of LDS -- which such a launch has to spare.  Big grids keep KC = 1 (four workgroups per CU).
template <int NT, int MT, int EPI, int WPE = 2, int KC = 1>
__global__ __launch_bounds__(256, WPE) void gemm_nt_kernel(const GemmNtParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 4 * MT * 16;
    constexpr int WROWS = NT * 16;
    constexpr int SUB = (WROWS + BM) * 16;                           // one 16-wide chunk: [W chunk | X rows], 16 floats per row
    constexpr int BUF = KC * SUB;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int M = p.M;
    const int m0 = blockIdx.x * BM;
    const int nc = blockIdx.y;                                       // N-chunk of this workgroup
    const int n0 = nc * WROWS;

    constexpr unsigned OOB = 0x80000000u;
    const int rows_here = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.wp_bytes, 0x00020000);
    // per-workgroup descriptors: base = first row of the block, extent = its valid rows -> rows >= M read 0 /
    // are not stored, and every offset fits 32 bits whatever M is
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)m0 * p.ldx, 0, ((rows_here - 1) * p.ldx + p.xcols) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)m0 * p.ldy, 0, rows_here * p.ldy * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        p.z ? p.z + (size_t)m0 * p.ldz : nullptr, 0, p.z ? rows_here * p.ldz * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (p.N + 3) / 4 * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_MUL ? p.m + (size_t)m0 * p.ldm : nullptr), 0, EPI == EPI_MUL ? rows_here * p.ldm * 4 : 0, 0x00020000);

    unsigned voff_x[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int rl = (wave * MT + j) * 16 + (lane >> 2);
        const unsigned slot = (unsigned)((lane & 3) ^ swz4(lane >> 2));
        voff_x[j] = rl < rows_here ? (unsigned)(rl * p.ldx + 4 * slot) * 4u : OOB;
    }
    const int rslot = 4 * (q ^ swz4(r16));

    f32x4 acc[NT][MT];
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Columns >= K of the last chunk are multiplied by the zero padding of the packed weights; what X holds
    // there only has to be finite (caller guarantee: row padding up to ldx is zero-initialised).
    auto dma_chunk = [&](int c, float* buf) {
        const int wbase = (c * p.rows + n0) * 64;
        for (int pc = wave; pc < NT; pc += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + pc * 256),
                                                     16, lane * 16, wbase + pc * 1024, 0, 0);
#pragma unroll
        for (int j = 0; j < MT; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(buf + (WROWS + (wave * MT + j) * 16) * 16),
                                                     16, voff_x[j], c * 64, 0, 0);
    };
    // stage s = chunks KC*s .. KC*s + KC - 1 (those that exist)
    auto dma_stage = [&](int s, float* buf) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (KC == 1 || s * KC + k < p.kchunks) dma_chunk(s * KC + k, buf + k * SUB);
    };
    // FULL = false: column tiles >= ntv -- the zero padding of the last N-chunk (3 of 13 tiles at N = 1608, 1 of 13 at N = 400) --
    // issue no MFMA: 3-4 % of the launch at those widths.  (The zero padding of the last K-chunk cannot be skipped the same way:
    // K-step j of a chunk covers k = 4 q + j over the four lane groups q -- the 16-byte fragment layout -- so at K = 402 every
    // step of the last chunk still holds k = 400 or 401.)
    auto compute = [&](const float* buf, auto full, int ntv) {
        constexpr bool FULL = decltype(full)::value;
        const float* Xl = buf + WROWS * 16;
        f32x4 pf[MT];
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) pf[jt] = *reinterpret_cast<const f32x4*>(&Xl[((wave * MT + jt) * 16 + r16) * 16 + rslot]);
        f32x4 af = *reinterpret_cast<const f32x4*>(&buf[r16 * 16 + rslot]);
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            f32x4 afn = af;
            if (it + 1 < NT) afn = *reinterpret_cast<const f32x4*>(&buf[((it + 1) * 16 + r16) * 16 + rslot]);
            if (FULL || it < ntv) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int jt = 0; jt < MT; ++jt) acc[it][jt] = mfma16(af[j], pf[jt][j], acc[it][jt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            af = afn;
        }
    };

    const int nt_valid = min(NT, (p.N - n0 + 15) >> 4);              // (workgroup-uniform)
    const int nstages = (p.kchunks + KC - 1) / KC;
    dma_stage(0, smem);
    __syncthreads();
    if (NRM_PRIO) __builtin_amdgcn_s_setprio(NRM_PRIO);
    for (int c = 0; c < nstages; ++c) {
        float* cur = smem + (c & 1) * BUF;
        float* nxt = smem + ((c & 1) ^ 1) * BUF;
        if (c + 1 < nstages && !(NRM_DIAG_GEMM & 4)) dma_stage(c + 1, nxt);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (KC > 1 && c * KC + k >= p.kchunks) break;
            if (nt_valid == NT) compute(cur + k * SUB, std::true_type{}, NT);
            else compute(cur + k * SUB, std::false_type{}, nt_valid);
        }
        if (!(NRM_DIAG_GEMM & 1)) __syncthreads();
    }
    if (NRM_PRIO) __builtin_amdgcn_s_setprio(0);
    if (NRM_DIAG_GEMM & 2) return;

    // epilogue: lane holds out[m = m0 + (wave*MT+jt)*16 + r16][n = n0 + 16 it + 4q .. +3]
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        if (it < nt_valid) {
        const int n = n0 + it * 16 + 4 * q;
        const int nb = (n0 + it * 16) * 4;
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_DGELU && p.bias) {
            // the descriptor extent is rounded up to 16 B (a float4 must not straddle it); lanes past N are zeroed
            bb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, 16 * q, nb, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = n + e < p.N ? bb[e] : 0.f;
        }
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            const int rl = (wave * MT + jt) * 16 + r16;
            f32x4 v = acc[it][jt] + bb;
            if (EPI == EPI_BIAS) {
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else if (EPI == EPI_GELU) {
                if (n < p.ldz) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb);
                const f32x4 g = gelu4(v);
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, g), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else if (EPI == EPI_MUL) {   // z = v (kept for backward), y = v * m
                if (n < p.ldz) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb);
                f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < p.ldm) mm = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_m, (unsigned)(rl * p.ldm + 4 * q) * 4u, nb, 0));
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, v * mm), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else {   // EPI_DGELU: y = acc * gelu'(z_saved)
                f32x4 zz = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < p.ldz) zz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb, 0));
                const f32x4 g = gelu_grad4_times(zz, acc[it][jt]);
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, g), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

GemmNtPlan gemm_nt_plan(int N) {
    const int n16 = (N + 15) / 16;
    GemmNtPlan pl;
    if (n16 <= 4) { pl.NT = 4; pl.MT = 4; }
    else {
        // 13x1 (exact for N = 402/416, 3 % padding for 1608) or 8x3, whichever pads less ...
        const int p13 = (n16 + 12) / 13 * 13, p8 = (n16 + 7) / 8 * 8;
        int best = p13 <= p8 ? p13 : p8;
        if (p13 <= p8) { pl.NT = 13; pl.MT = 1; } else { pl.NT = 8; pl.MT = 3; }
        // ... unless 9x2 or 5x4 saves a tenth of the column tiles or more: the 258 / 264-wide hidden layers of D = 256 and
        // of the reference's default sizes are 17 tiles (8-wide chunks compute 24), the 66-wide one is 5 (8-wide: 8)
        static const bool narrow = [] { const char* e = getenv("NRM_NT_NARROW"); return !(e && e[0] == '0'); }();
        const int p9 = (n16 + 8) / 9 * 9, p5 = (n16 + 4) / 5 * 5;
        if (narrow && p9 * 10 <= best * 9) { pl.NT = 9; pl.MT = 2; best = p9; }
        if (narrow && p5 * 10 <= best * 9) { pl.NT = 5; pl.MT = 4; best = p5; }
    }
    pl.nchunks = (n16 + pl.NT - 1) / pl.NT;
    pl.rows = pl.nchunks * pl.NT * 16;
    return pl;
}

template <int NT, int MT, int WPE = 2, int KC = 1>
static hipError_t launch_nt(const GemmNtParams& p, const GemmNtPlan& pl, int epi, hipStream_t st) {
    constexpr int BM = 4 * MT * 16;
    const dim3 grid((p.M + BM - 1) / BM, pl.nchunks), block(256);
    if (KC > 1) {                                                    // more than 64 KB of static LDS: opt in once per kernel and device
        constexpr int lds = 2 * KC * (NT * 16 + BM) * 16 * 4;
        static_assert(lds <= 160 * 1024, "two stages must fit the CU's LDS");
    }
    switch (epi) {
        case EPI_BIAS:  hipLaunchKernelGGL((gemm_nt_kernel<NT, MT, EPI_BIAS, WPE, KC>), grid, block, 0, st, p); break;
        case EPI_GELU:  hipLaunchKernelGGL((gemm_nt_kernel<NT, MT, EPI_GELU, WPE, KC>), grid, block, 0, st, p); break;
        case EPI_DGELU: hipLaunchKernelGGL((gemm_nt_kernel<NT, MT, EPI_DGELU, WPE, KC>), grid, block, 0, st, p); break;
        case EPI_MUL:   hipLaunchKernelGGL((gemm_nt_kernel<NT, MT, EPI_MUL, WPE, KC>), grid, block, 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t gemm_nt_launch(const GemmNtParams& p, const GemmNtPlan& pl, int epi, hipStream_t st) {
    if (p.M <= 0) return hipSuccess;
    // The row-tile count MT of a workgroup (64*MT rows) is free at launch time (the packed operand depends on NT only).  A grid
    // of fewer than two workgroups per CU leaves every workgroup alone with its chain of K-chunk rounds (DMA, barrier, a few
    // hundred MFMA cycles): C2's 258-wide layers were 240 workgroups of 128 rows at 57 % of the MFMA peak.  Such launches
    // take one row tile per wave instead.  NRM_NT_SMALLM=0 keeps the planned MT.
    static const bool small_m = [] { const char* e = getenv("NRM_NT_SMALLM"); return !(e && e[0] == '0'); }();
    const long wgs = (long)((p.M + 64 * pl.MT - 1) / (64 * pl.MT)) * pl.nchunks;
    // Deeper LDS stages (template parameter KC = 2 | 4 chunks per stage) for launches of few workgroups: built on the theory that
    // such a launch is a chain of DMA-latency-bound rounds, measured (graph replay, same box, whole step): C1 1.795 -> 1.84 ms
    // (gemm_nt 33.2 -> 35.3 us per launch), reference default 12.9 -> 12.2 us per launch -- the latency chain is not what limits
    // these launches.  Not a default: NRM_NT_KC=2|4 selects it (tests hold every depth bit-identical to KC = 1).
    const char* kc_e = getenv("NRM_NT_KC");                           // (read per launch: tests switch forms inside one process)
    int kc = kc_e ? atoi(kc_e) : 1;
    if (kc != 2 && kc != 4) kc = 1;
    if (p.kchunks < 2 * kc) kc = p.kchunks >= 4 ? 2 : 1;             // (p.kchunks = ceil(K / 16))
    if (kc > 1 && (pl.MT == 1 || (small_m && wgs < 512))) {
        if (kc == 4) {
            if (pl.NT == 4) return launch_nt<4, 1, 1, 4>(p, pl, epi, st);
            if (pl.NT == 5) return launch_nt<5, 1, 1, 4>(p, pl, epi, st);
            if (pl.NT == 8) return launch_nt<8, 1, 1, 4>(p, pl, epi, st);
            if (pl.NT == 9) return launch_nt<9, 1, 1, 4>(p, pl, epi, st);
            if (pl.NT == 13) return launch_nt<13, 1, 1, 4>(p, pl, epi, st);
        } else {
            if (pl.NT == 4) return launch_nt<4, 1, 2, 2>(p, pl, epi, st);
            if (pl.NT == 5) return launch_nt<5, 1, 2, 2>(p, pl, epi, st);
            if (pl.NT == 8) return launch_nt<8, 1, 2, 2>(p, pl, epi, st);
            if (pl.NT == 9) return launch_nt<9, 1, 2, 2>(p, pl, epi, st);
            if (pl.NT == 13) return launch_nt<13, 1, 2, 2>(p, pl, epi, st);
        }
    }
    if (small_m && pl.MT > 1 && wgs < 512) {
        if (pl.NT == 4) return launch_nt<4, 1, 4>(p, pl, epi, st);
        if (pl.NT == 5) return launch_nt<5, 1, 4>(p, pl, epi, st);
        if (pl.NT == 8) return launch_nt<8, 1, 4>(p, pl, epi, st);
        if (pl.NT == 9) return launch_nt<9, 1, 4>(p, pl, epi, st);
    }
    if (pl.NT == 4) return launch_nt<4, 4>(p, pl, epi, st);
    if (pl.NT == 5) return launch_nt<5, 4>(p, pl, epi, st);
    if (pl.NT == 8) return launch_nt<8, 3>(p, pl, epi, st);
    if (pl.NT == 9) return launch_nt<9, 2, 3>(p, pl, epi, st);
    // 13 column tiles x ONE row tile per wave: 97 VGPRs and 35 KB of LDS, four workgroups per CU.  Measured on the head's
    // layers (M = 30 720, 1608 <-> 402): 0.199 ms per launch against 0.211 ms for 13x2 (129 VGPRs, 43 KB, three per CU),
    // although every W chunk then serves 64 rows instead of 128.  NRM_NT_13X2=1 restores 13x2.
    static const int use13x2 = [] { const char* e = getenv("NRM_NT_13X2"); return e && e[0] == '1' ? 1 : 0; }();
    if (use13x2) return launch_nt<13, 2>(p, pl, epi, st);
    return launch_nt<13, 1, 4>(p, pl, epi, st);
}

// ---------------------------------------------------------------------------------------------
// gemm_tn: every wave owns a (KT*16 i) x (DT*16 j) tile of C[i,j] = sum_r A[r,i] B[r,j] and a row range.
// slab layout: ws[split][j][ldws] (transposed, float4 along i);  colsum[split][i] = sum_r A[r,i] (tiles j0 == 0).
//
// Column layout of an operand block of W MFMA tiles (W in {2, 4, 5, 8}; as in the backward contractions, pwattn_bwd.hip
// tile_col): fragment index r (0..15) of tile t is block column
//     W = 2:  2 r + t            (one  8-byte load per lane and reduction step)
//     W = 4:  4 r + t            (one 16-byte load)
//     W = 5:  4 r + t | 64 + r   (16 + 4 bytes)
//     W = 8:  4 r + t | 64 + 4 r + (t - 4)      (two 16-byte loads)
// so a lane's loads feed W MFMA operands without any shuffle.  Round 4 added W = 2 and 8: the head's weight gradients are
// 26 x 101 tiles of 16 (402 x 1608), which 4 x 4 wave tiles cover with 10.4 % padding and 2 x 8 with 3 % (13 x 13 exactly
// in i); w1's 25 x 26 tiles are exact in 5 x 2.
template <int W> __device__ __forceinline__ int tn_col(int t, int r) {
    if (W == 2) return 2 * r + t;
    if (W == 5 && t == 4) return 64 + r;
    if (W == 8 && t >= 4) return 64 + 4 * r + (t - 4);
    return 4 * r + t;
}
#if defined(__HIP_DEVICE_COMPILE__)
template <int W> __device__ __forceinline__ void tn_load(float (&v)[W], const __amdgpu_buffer_rsrc_t rs, unsigned vlo, unsigned vhi, int soff) {
    if constexpr (W == 2) {
        const auto x = __builtin_amdgcn_raw_buffer_load_b64(rs, vlo, soff, 0);
        v[0] = __uint_as_float(x[0]); v[1] = __uint_as_float(x[1]);
    } else {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, vlo, soff, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(x[e]);
        if constexpr (W == 5) v[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vhi, soff, 0));
        if constexpr (W == 8) {
            const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rs, vhi, soff, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] = __uint_as_float(y[e]);
        }
    }
}
#endif
constexpr int tn_waves_per_simd(int KT, int DT) { return KT * DT <= 16 ? 4 : 3; }      // 64 accumulator VGPRs: 4 waves; 100: 3

template <int KT, int DT>
__global__ __launch_bounds__(256, tn_waves_per_simd(KT, DT)) void gemm_tn_kernel(const GemmTnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    // XCD-aware order (as in the backward contractions): the nti*ntj output tiles of one row split read the SAME rows of
    // A and B; consecutive logical ids go to one XCD, so those rows come from HBM once instead of once per XCD.
    const int nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int bx = logical % gridDim.x, by = logical / gridDim.x;
    const int ti = bx % p.nti, tj = bx / p.nti;
    const int i0 = ti * (KT * 16), j0 = tj * (DT * 16);
    // the gradient this GEMM's slabs will be ADDED to by the slab reduction is zeroed here, on the side (its own fill
    // launch otherwise: 12 per step)
    if (p.zero_out) {
        const long n4 = p.zero_n >> 2;
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long i = (long)lin * 256 + tid; i < n4; i += (long)nblk * 256) reinterpret_cast<f32x4*>(p.zero_out)[i] = z4;
        for (long i = (n4 << 2) + (long)lin * 256 + tid; i < p.zero_n; i += (long)nblk * 256) p.zero_out[i] = 0.f;
    }
    const int split = by * 4 + wave;
    if (split >= p.nsplit) return;
    const int r_lo = split * p.rps;
    const int r_hi = min(p.R, r_lo + p.rps);
    const int nrows = r_hi - r_lo;
    const int nsteps = nrows > 0 ? (nrows + 3) >> 2 : 0;

    // descriptors start at this wave's first row; rows >= r_hi read 0.  Columns past the matrix edge read
    // whatever follows in memory (finite) and only feed outputs that are never stored.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.A) + (size_t)r_lo * p.lda, 0, nrows > 0 ? ((nrows - 1) * p.lda + p.acols) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.B) + (size_t)r_lo * p.ldb, 0, nrows > 0 ? ((nrows - 1) * p.ldb + p.bcols) * 4 : 0, 0x00020000);
    // lane (r16, q) reads reduction row q of a step; low / high part of its columns (tn_col)
    const unsigned va_lo = (unsigned)(q * p.lda + i0 + tn_col<KT>(0, r16)) * 4u, va_hi = (unsigned)(q * p.lda + i0 + tn_col<KT>(KT - 1, r16) - (KT == 8 ? 3 : 0)) * 4u;
    const unsigned vb_lo = (unsigned)(q * p.ldb + j0 + tn_col<DT>(0, r16)) * 4u, vb_hi = (unsigned)(q * p.ldb + j0 + tn_col<DT>(DT - 1, r16) - (DT == 8 ? 3 : 0)) * 4u;
    const int astep = p.lda * 16, bstep = p.ldb * 16;

    f32x4 C[KT][DT];
    float cs[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it) cs[it] = 0.f;

    // four operand register sets: operands are requested three reduction steps ahead of their MFMAs
    float a0[KT], b0[DT], a1[KT], b1[DT], a2[KT], b2[DT], a3[KT], b3[DT];
    auto load_step = [&](float (&a)[KT], float (&b)[DT], int s) {
        tn_load<KT>(a, ra, va_lo, va_hi, s * astep);
        tn_load<DT>(b, rb, vb_lo, vb_hi, s * bstep);
    };
    auto mfma_batch = [&](const float (&a)[KT], const float (&b)[DT]) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            cs[it] += a[it];
#pragma unroll
            for (int jt = 0; jt < DT; ++jt) C[it][jt] = mfma16(a[it], b[jt], C[it][jt]);
        }
    };
#pragma unroll
    for (int it = 0; it < KT; ++it)
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) C[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (0 < nsteps) load_step(a0, b0, 0);
    if (1 < nsteps) load_step(a1, b1, 1);
    if (2 < nsteps) load_step(a2, b2, 2);
    for (int s = 0; s < nsteps; s += 4) {
        if (NRM_DIAG_GEMM & 8) { mfma_batch(a0, b0); mfma_batch(a1, b1); mfma_batch(a2, b2); mfma_batch(a0, b0); continue; }
        if (s + 3 < nsteps) load_step(a3, b3, s + 3);
        mfma_batch(a0, b0);
        if (s + 1 < nsteps) {
            if (s + 4 < nsteps) load_step(a0, b0, s + 4);
            mfma_batch(a1, b1);
        }
        if (s + 2 < nsteps) {
            if (s + 5 < nsteps) load_step(a1, b1, s + 5);
            mfma_batch(a2, b2);
        }
        if (s + 3 < nsteps) {
            if (s + 6 < nsteps) load_step(a2, b2, s + 6);
            mfma_batch(a3, b3);
        }
    }

    // lane holds C[it][jt][e] = c[i0 + tn_col<KT>(it, 4q+e)][j0 + tn_col<DT>(jt, r16)]; the slab is transposed (ws[split][j][i]),
    // and every store below is a float4 of four consecutive i
    float* wsp = p.ws + (size_t)split * p.ncols_j * p.ldws;
#pragma unroll
    for (int jt = 0; jt < DT; ++jt) {
        const int j = j0 + tn_col<DT>(jt, r16);
        if (j >= p.ncols_j) continue;
        float* row = wsp + (size_t)j * p.ldws + i0;
        if constexpr (KT == 2) {
            // i = 8 q + 2 e + it
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ii = 8 * q + 4 * h;
                if (i0 + ii < p.ldws) *reinterpret_cast<f32x4*>(row + ii) = f32x4{C[0][jt][2 * h], C[1][jt][2 * h], C[0][jt][2 * h + 1], C[1][jt][2 * h + 1]};
            }
        } else {
#pragma unroll
            for (int g = 0; g < (KT == 8 ? 2 : 1); ++g)                 // blocks of four tiles: i = 64 g + 16 q + 4 e + (it - 4 g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ii = 64 * g + 16 * q + 4 * e;
                    if (i0 + ii < p.ldws)
                        *reinterpret_cast<f32x4*>(row + ii) = f32x4{C[4 * g][jt][e], C[4 * g + 1][jt][e], C[4 * g + 2][jt][e], C[4 * g + 3][jt][e]};
                }
            if constexpr (KT == 5)
                if (i0 + 64 + 4 * q < p.ldws) *reinterpret_cast<f32x4*>(row + 64 + 4 * q) = C[KT - 1][jt];     // i = 64 + 4 q + e
        }
    }
    if (p.colsum && tj == 0) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const float v = sum_rows4(cs[it]);
            const int i = i0 + tn_col<KT>(it, r16);
            if (q == 0 && i < p.ldws) p.colsum[(size_t)split * p.ldws + i] = v;
        }
    }
#endif
}

// wave-tile shapes the fp32 kernel is built for, in order of preference (fewer loads per MFMA first)
static const int kTnShapes[][2] = {{4, 4}, {5, 5}, {2, 8}, {8, 2}, {5, 2}, {2, 5}};

GemmTnPlan gemm_tn_plan(int ncols_i, int ncols_j, int R, int target_waves, int mma) {
    GemmTnPlan pl;
    const int ni16 = (ncols_i + 15) / 16, nj16 = (ncols_j + 15) / 16;
    // the shape with the fewest padded MFMA tiles (ties: the earlier entry)
    pl.KT = pl.DT = 4;
    long best = -1;
    static const bool wide = [] { const char* e = getenv("NRM_TN_SHAPES"); return !(e && e[0] == '0'); }();      // 0: 4x4 / 5x5 only (round 3)
    for (int k = 0; k < (mma ? 1 : (wide ? 6 : 2)); ++k) {          // bf16 forms (gemm_bf16.hip): 4x4 tiles, 32-row super-steps
        const int kt = kTnShapes[k][0], dt = kTnShapes[k][1];
        const long w = (long)((ni16 + kt - 1) / kt) * kt * ((nj16 + dt - 1) / dt) * dt;
        if (best < 0 || w < best) { best = w; pl.KT = kt; pl.DT = dt; }
    }
    pl.nti = (ni16 + pl.KT - 1) / pl.KT;
    pl.ntj = (nj16 + pl.DT - 1) / pl.DT;
    const int tiles = pl.nti * pl.ntj;
    // one round of wave tasks at the kernel's occupancy (4 waves per SIMD with <= 64 accumulator registers, else 3): with
    // 4096 tasks on the 3072 slots of the 5x5 kernel a third of the chip ran a second round alone
    if (target_waves <= 0) target_waves = 1024 * (mma ? 1 : tn_waves_per_simd(pl.KT, pl.DT));
    int ns = target_waves / tiles / 4 * 4;
    if (ns < 4) ns = 4;
    // at least 128 rows per split: every split costs a [ncols_j x ldws] partial slab that is written and read back by the
    // slab reduction (C2-small, 16 384 x 256 x 256: 256 splits of 64 rows = 67 MB of slabs for a 256 KB gradient; halving
    // them took the step from 5.99 to 5.73 ms); the large shapes (C3: 960 / 320 rows per split) are not affected
    const int max_ns = (R + 127) / 128;
    if (ns > max_ns) ns = max_ns < 1 ? 1 : max_ns;
    pl.rps = ((R + ns - 1) / ns + 3) / 4 * 4;        // multiple of 4 rows
    if (mma) pl.rps = (pl.rps + 31) / 32 * 32;       // whole 32-row MFMA steps
    if (pl.rps < 4) pl.rps = 4;
    pl.nsplit = (R + pl.rps - 1) / pl.rps;
    if (pl.nsplit < 1) pl.nsplit = 1;
    return pl;
}

hipError_t gemm_tn_launch(GemmTnParams p, const GemmTnPlan& pl, hipStream_t st) {
    p.nti = pl.nti; p.nsplit = pl.nsplit; p.rps = pl.rps;
    const dim3 grid(pl.nti * pl.ntj, (pl.nsplit + 3) / 4), block(256);
    const int shape = pl.KT * 16 + pl.DT;
    switch (shape) {
        case 4 * 16 + 4: hipLaunchKernelGGL((gemm_tn_kernel<4, 4>), grid, block, 0, st, p); break;
        case 5 * 16 + 5: hipLaunchKernelGGL((gemm_tn_kernel<5, 5>), grid, block, 0, st, p); break;
        case 2 * 16 + 8: hipLaunchKernelGGL((gemm_tn_kernel<2, 8>), grid, block, 0, st, p); break;
        case 8 * 16 + 2: hipLaunchKernelGGL((gemm_tn_kernel<8, 2>), grid, block, 0, st, p); break;
        case 5 * 16 + 2: hipLaunchKernelGGL((gemm_tn_kernel<5, 2>), grid, block, 0, st, p); break;
        case 2 * 16 + 5: hipLaunchKernelGGL((gemm_tn_kernel<2, 5>), grid, block, 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// slab reduction: the partial slabs ws[s][j][i] of gemm_tn / the attention dW_p pass -> the gradient tensor itself
// (sum over s + transpose + strided placement in ONE launch; was ATen sum(dim=0).t() + cat: 24 reduce launches and a
// concat per step).  blockDim (32, 8); a block owns a 128(i) x 32(j) tile AND a chunk of the splits (blockIdx.y), so a
// 200 MB slab set (324 splits of 400 x 400 at C3) is streamed by thousands of blocks: reads run along i (the slab's fast
// index), the partial tile is transposed through LDS and ADDED to the output along j with float atomics (the caller
// zero-initialises `out`; one add per element and split chunk).  blockIdx.y == gridDim.y - 1 sums the bias vectors.
__device__ __forceinline__ void slab_reduce_block(const SlabReduceParams& p, int spc, int bx, int by, int nby,
                                                  float (*tile)[129]) {
    // tile = 128 (i, the slab's fast index: one float4 per thread and row) x 32 (j); LDS row stride 129: conflict-free both ways
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int nbi = (p.ni + 127) / 128;
    if (by == nby - 1) {                                               // vector leg (db = sum of per-split column sums)
        if (!p.vec) return;
        const int i = bx * 256 + ty * 32 + tx;
        if (i < p.ni) {
            float a = 0.f;
#pragma unroll 8
            for (int s = 0; s < p.nsplit; ++s) a += p.vec[(size_t)s * p.ldws + i];
            p.vec_out[i] = a;
        }
        return;
    }
    if (bx >= nbi * ((p.nj + 31) / 32)) return;                        // grid.x is sized for the longer of the two legs
    const int i0 = (bx % nbi) * 128, j0 = (bx / nbi) * 32;
    const int s_lo = by * spc, s_hi = min(p.nsplit, s_lo + spc);
    const size_t slab = (size_t)p.nj * p.ldws;
    f32x4 a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int i = i0 + 4 * tx;
    if (i < p.ldws) {                                                   // ldws % 4 == 0: the float4 stays inside the padded row
#pragma unroll 4
        for (int s = s_lo; s < s_hi; ++s) {
            const float* src = p.ws + s * slab + (size_t)(j0 + ty) * p.ldws + i;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + ty + 8 * r < p.nj) a[r] += *reinterpret_cast<const f32x4*>(src + (size_t)8 * r * p.ldws);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ty + 8 * r][4 * tx + e] = a[r][e];      // tile[j][i]
    __syncthreads();
    const int t = ty * 32 + tx, jj = t & 31;
    const int j = j0 + jj;
    if (j < p.nj) {
        for (int il = t >> 5; il < 128; il += 8) {
            const int ii = i0 + il;
            if (ii >= p.ni) break;
            const float v = tile[jj][il];
            atomicAdd(p.out + ii * p.ors + j * p.ocs, v);
            if (p.out2) atomicAdd(p.out2 + ii * p.ors2 + j * p.ocs2, p.sign2 * v);
        }
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabReduceParams p, int spc) {
    __shared__ float tile[32][129];
    slab_reduce_block(p, spc, blockIdx.x, blockIdx.y, gridDim.y, tile);
}

// Several slab sets in ONE launch (blockIdx.z = set; the table travels in the kernel arguments): a training step's 14
// reductions are small, latency-bound launches (20-75 us each whatever their size: 0.6 ms of a 2.1 ms step at the reference's
// default sizes) that have nothing to wait for but their own slabs, so the host side defers them and flushes them together
// before the optimizer reads the gradients (ops.flush_slab_reductions).
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabTable tab) {
    __shared__ float tile[32][129];
    // flat grid: block -> (set, tile, split chunk) through the prefix table (sets differ a lot in their block counts: a 3-D grid
    // sized for the largest in both directions would be mostly empty blocks)
    int b = blockIdx.x, e = 0;
    while (e + 1 < tab.n && b >= tab.first[e + 1]) ++e;
    b -= tab.first[e];
    slab_reduce_block(tab.e[e], tab.spc[e], b % tab.gx[e], b / tab.gx[e], tab.nchunk[e] + 1, tile);
}

static void slab_plan(const SlabReduceParams& p, int& gx, int& nchunk, int& spc) {
    const int nbi = (p.ni + 127) / 128, nbj = (p.nj + 31) / 32;
    const int tiles = nbi * nbj;
    gx = tiles;
    if (p.vec && (p.ni + 255) / 256 > gx) gx = (p.ni + 255) / 256;
    // Blocks = tiles x split chunks.  Every block ends with one float atomic per element of its tile, and float atomics run at
    // ~65 G/s chip-wide (measured: they, not the slab reads, were most of this kernel -- 32 chunks x 52 tiles at C3 = 6.8 M
    // atomics), so: about two blocks per CU, each walking its chunk of slabs with several loads in flight (the s loop below is
    // unrolled by 4).  A 64 x 64 gradient that arrives in 1920 slabs (reference default sizes) gets 128 chunks of 15.
    nchunk = 512 / tiles;
    if (nchunk > 128) nchunk = 128;
    if (nchunk > (p.nsplit + 3) / 4) nchunk = (p.nsplit + 3) / 4;       // >= 4 slabs per block
    if (nchunk < 1) nchunk = 1;
    spc = (p.nsplit + nchunk - 1) / nchunk;
    nchunk = (p.nsplit + spc - 1) / spc;
}

hipError_t slab_reduce_launch(const SlabReduceParams& p, hipStream_t st) {
    if (p.ni <= 0 || p.nj <= 0) return hipSuccess;
    int gx, nchunk, spc;
    slab_plan(p, gx, nchunk, spc);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, nchunk + 1), dim3(32, 8), 0, st, p, spc);
    return hipGetLastError();
}

hipError_t slab_reduce_multi_launch(const SlabReduceParams* ps, int n, hipStream_t st) {
    for (int base = 0; base < n; base += SLAB_MAX) {
        SlabTable tab;
        const int m = n - base < SLAB_MAX ? n - base : SLAB_MAX;
        int blocks = 0;
        for (int e = 0; e < m; ++e) {
            tab.e[e] = ps[base + e];
            slab_plan(tab.e[e], tab.gx[e], tab.nchunk[e], tab.spc[e]);
            tab.first[e] = blocks;
            blocks += tab.gx[e] * (tab.nchunk[e] + 1);
        }
        tab.n = m;
        hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(blocks), dim3(32, 8), 0, st, tab);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

int gemm_diag_flags() { return NRM_DIAG_GEMM ? 0x100 : 0; }

}  // namespace nrm
