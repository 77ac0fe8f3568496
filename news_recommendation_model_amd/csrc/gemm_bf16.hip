// Dense-layer GEMMs on the bf16 matrix cores (NRM_MMA_BF16 / NRM_MMA_BF16X3) for BASELINE config 2: the same contracts as the
// fp32 kernels of gemm.hip (reference MLP / Linear layers, models/attention_model.py:29-32, user_model.py:31-35), operands
// rounded to bf16 (MMA 1) or split into bf16 hi + lo with three MFMAs per product (MMA 2: fp32-class accuracy, as the
// attention's bf16x3 arithmetic), fp32 accumulation, fp32 bias / GELU / column sums.  gfx950 only.
//
// With the contraction 16/3x cheaper than on the fp32 pipe, the fp32 kernels' structure (an LDS-DMA'd 16-wide K-chunk and a
// workgroup barrier per chunk) would spend its time in barriers; both kernels here are barrier-free after their prologue:
//
//   gemm_tn_bf16   C[i,j] = sum_r A[r,i] B[r,j] (dW = dY^T X) as gemm_tn: a wave owns a 64 x 64 tile and a row range, operands
//                  stream straight from global memory -- eight 4-row steps make one 32-deep MFMA operand per tile (reduction
//                  position (lane quarter q, element j) = row 4 j + q), the raw rows of super-step s + 1 are requested before
//                  the MFMAs of super-step s.
//   gemm_nt_rx     Y = epi(X W^T): "resident X".  A workgroup converts its BM rows of X (all K columns) ONCE into a bf16 hi/lo
//                  LDS image; its 8 waves then take output-column tiles in turn and stream the weight fragments -- packed per
//                  optimizer step in exactly the MFMA operand order, so a fragment is one coalesced 1-KiB load per image --
//                  from L2, several fragments ahead.  No barrier, no LDS-DMA, accumulators = BM / 16 tiles.
#include <cstdlib>
// timing diagnostics only (results are WRONG with any bit set; nrm_build_flags reports it): bit 0 the X block is not loaded /
// converted, bit 1 weight fragments are loaded for the first group only, bit 2 no epilogue
#ifndef NRM_DIAG_RX
#define NRM_DIAG_RX 0
#endif
#ifndef RX_WAVES_N
#define RX_WAVES_N 8
#endif
#include "common.hpp"
#include "gemm.hpp"

namespace nrm {

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// ---------------------------------------------------------------------------------------------
template <int MMA>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const GemmTnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KT = 4, DT = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int bx = logical % gridDim.x, by = logical / gridDim.x;
    const int ti = bx % p.nti, tj = bx / p.nti;
    const int i0 = ti * 64, j0 = tj * 64;
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
    const int nss = nrows > 0 ? (nrows + 31) >> 5 : 0;

    // descriptors start at this wave's first row; rows >= r_hi read 0.  Columns past the matrix edge read whatever follows in
    // memory (finite: padding is zero-initialised) and only feed outputs that are never stored.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.A) + (size_t)r_lo * p.lda, 0, nrows > 0 ? ((nrows - 1) * p.lda + p.acols) * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.B) + (size_t)r_lo * p.ldb, 0, nrows > 0 ? ((nrows - 1) * p.ldb + p.bcols) * 4 : 0, 0x00020000);
    // lane (r16, q) holds columns 4 r16 + tile of tiles 0..3 (one 16-byte load feeds four MFMA operands); load j of a super-step
    // is row 4 j + q
    constexpr unsigned OOB = 0x80000000u;
    const unsigned va4 = i0 + 4 * r16 < p.acols ? (unsigned)(q * p.lda + i0 + 4 * r16) * 4u : OOB;
    const unsigned vb4 = j0 + 4 * r16 < p.bcols ? (unsigned)(q * p.ldb + j0 + 4 * r16) * 4u : OOB;
    const int astep = p.lda * 16, bstep = p.ldb * 16;                  // 4 rows, bytes

    f32x4 C[KT][DT];
    float cs[KT];
#pragma unroll
    for (int it = 0; it < KT; ++it) {
        cs[it] = 0.f;
#pragma unroll
        for (int jt = 0; jt < DT; ++jt) C[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 araw[8], braw[8];
    auto load_ss = [&](int ss) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            araw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, va4, (8 * ss + j) * astep, 0));
            braw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb4, (8 * ss + j) * bstep, 0));
        }
    };
    if (nss > 0) load_ss(0);
    for (int ss = 0; ss < nss; ++ss) {
        bf16x8 af[KT], al[KT], bf[DT], bl[DT];
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cs[it] += araw[j][it];
                __bf16 h, l;
                split_bf16(araw[j][it], h, l);
                af[it][j] = h;
                if (MMA == 2) al[it][j] = l;
            }
#pragma unroll
        for (int jt = 0; jt < DT; ++jt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 h, l;
                split_bf16(braw[j][jt], h, l);
                bf[jt][j] = h;
                if (MMA == 2) bl[jt][j] = l;
            }
        if (ss + 1 < nss) load_ss(ss + 1);                               // the next 32 rows: under this super-step's MFMAs
#pragma unroll
        for (int it = 0; it < KT; ++it)
#pragma unroll
            for (int jt = 0; jt < DT; ++jt) {
                f32x4 c = C[it][jt];
                if (MMA == 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[it], bf[jt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], bl[jt], c, 0, 0, 0);
                }
                C[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], bf[jt], c, 0, 0, 0);
            }
    }

    // lane holds C[it][jt][e] = c[i0 + 16 q + 4 e + it][j0 + 4 r16 + jt]; the slab is transposed (ws[split][j][i])
    float* wsp = p.ws + (size_t)split * p.ncols_j * p.ldws;
#pragma unroll
    for (int jt = 0; jt < DT; ++jt) {
        const int j = j0 + 4 * r16 + jt;
        if (j >= p.ncols_j) continue;
        float* row = wsp + (size_t)j * p.ldws + i0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ii = 16 * q + 4 * e;
            if (i0 + ii < p.ldws) *reinterpret_cast<f32x4*>(row + ii) = f32x4{C[0][jt][e], C[1][jt][e], C[2][jt][e], C[3][jt][e]};
        }
    }
    if (p.colsum && tj == 0) {
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const float v = sum_rows4(cs[it]);
            const int i = i0 + 4 * r16 + it;
            if (q == 0 && i < p.ldws) p.colsum[(size_t)split * p.ldws + i] = v;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_rx: Y[M, N] = epi(X[M, K] W^T) with the workgroup's X rows resident in LDS as bf16 hi [+ lo].
//   LDS image  xres[c][img][row][32 bf16]: 64-byte rows, 16-byte slots XOR-swizzled (swz4, common.hpp); slot s of chunk c holds
//              the reduction positions k = 32c + 8s + {0..7} of lane quarter s.
//   weights    wq[it][c][img][r16][32 bf16] (pack_wfrag, gemm.hip): output-column tile it, row r16 = column 16 it + r16 -- one
//              1-KiB wave load per image is the MFMA A operand of (tile it, chunk c) for all 64 lanes.
//   MFMA       A = weights (MFMA row = output column), B = X rows: a lane ends up with 4 consecutive output columns of one row
//              (float4 epilogue I/O, as gemm_nt).
constexpr int RX_WAVES = RX_WAVES_N;
constexpr int RX_LDS_BUDGET = 148 * 1024;

// G = reduction chunks per weight-fragment group (3 or 4, whichever pads ceil(K / 32) less): a group's 2 G loads are issued
// together, one group ahead of its MFMAs, into two register sets used alternately
template <int RT, int MMA, int EPI, int G>
__global__ __launch_bounds__(RX_WAVES * 64, RT <= 2 ? RX_WAVES / 2 : RX_WAVES / 4) void gemm_nt_rx_kernel(const GemmRxParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WIMG = MMA == 2 ? 2 : 1;
    constexpr int BM = RT * 16;
    extern __shared__ __attribute__((aligned(16))) float xres[];         // [k32][WIMG][BM][16 floats]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int K32 = p.k32;
    const int m0 = blockIdx.x * BM;
    const int rows_here = min(BM, p.M - m0);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)m0 * p.ldx, 0, ((rows_here - 1) * p.ldx + p.xcols) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wq), 0, p.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)m0 * p.ldy, 0, rows_here * p.ldy * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        p.z ? p.z + (size_t)m0 * p.ldz : nullptr, 0, p.z ? rows_here * p.ldz * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (p.N + 3) / 4 * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == EPI_MUL ? p.m + (size_t)m0 * p.ldm : nullptr), 0, EPI == EPI_MUL ? rows_here * p.ldm * 4 : 0, 0x00020000);

    // ---- prologue: the block's rows, 4 columns per thread and turn (four turns in flight), split once and written as 8-byte
    // halves of a slot
    const int g4_per_row = K32 * 8;
    const int nunit = BM * g4_per_row;
    for (int base = tid; base < ((NRM_DIAG_RX & 1) ? 0 : nunit); base += 4 * RX_WAVES * 64) {
        f32x4 v[4];
        int rowv[4], g4v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * RX_WAVES * 64;
            rowv[u] = idx / g4_per_row; g4v[u] = idx - rowv[u] * g4_per_row;
            const bool ok = idx < nunit && rowv[u] < rows_here && 4 * g4v[u] + 3 < p.xcols;
            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)((rowv[u] * p.ldx + 4 * g4v[u]) * 4) : OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (base + u * RX_WAVES * 64 < nunit) {
                const int row = rowv[u], g4 = g4v[u], k = 4 * g4;
                unsigned short hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ve = k + e < p.K ? v[u][e] : 0.f;        // (columns in [K, ldx) are finite padding: drop them)
                    __bf16 h, l;
                    split_bf16(ve, h, l);
                    hi[e] = __builtin_bit_cast(unsigned short, h);
                    lo[e] = __builtin_bit_cast(unsigned short, l);
                }
                const int c = g4 >> 3, sl = (g4 & 7) >> 1, half = g4 & 1;
                float* dst = xres + ((size_t)(c * WIMG) * BM + row) * 16 + 4 * (sl ^ swz4(row)) + 2 * half;
                *reinterpret_cast<uint2*>(dst) = uint2{(unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16)};
                if (MMA == 2)
                    *reinterpret_cast<uint2*>(dst + BM * 16) = uint2{(unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16)};
            }
        }
    }
    __syncthreads();

    // ---- this wave's output-column tiles it = wave, wave + 8, ...: a flat walk over (tile, group of G chunks); the weight fragments
    // of the NEXT group (possibly the next tile's first) are requested before the MFMAs of the current one.  Loads are never
    // branched around: positions behind the end read with an out-of-range offset (0), chunks >= K32 of a tile's last group too
    // (their X image is not touched: the MFMAs of such a chunk are skipped by a uniform predicate on the chunk index).
    const int ntile_w = p.nt16 > wave ? (p.nt16 - wave + RX_WAVES - 1) / RX_WAVES : 0;
    const int ngrp = (K32 + G - 1) / G;
    const int total = ntile_w * ngrp;
    const int rslot = 4 * (q ^ swz4(r16));
    const unsigned wlane = (unsigned)(r16 * 64 + q * 16);               // row r16 of the fragment, its 16-byte slot q (the 64 lanes cover the 1 KiB)
    u32x4 whA[G], wlA[G], whB[G], wlB[G], whC[G], wlC[G];              // three sets: requested TWO groups ahead of their MFMAs
    auto wload = [&](u32x4 (&wh)[G], u32x4 (&wl)[G], int f) {           // group f of the flat walk
        const int itl = f / ngrp, g = f - itl * ngrp;
        const int tile = wave + RX_WAVES * itl;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int cc = g * G + j;
            const unsigned vo = (f < total && cc < K32) ? wlane : OOB;
            const int so = ((tile * K32 + cc) * WIMG) * 1024;
            wh[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo, so, 0);
            if (MMA == 2) wl[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo, so + 1024, 0);
        }
    };

    f32x4 acc[RT];
    int it = wave;
    auto epilogue = [&]() {
        const int n = it * 16 + 4 * q;
        const int nb = it * 64;                                           // byte offset of the tile's first column
        f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_DGELU && p.bias) {
            bb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, 16 * q, nb, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = n + e < p.N ? bb[e] : 0.f;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int rl = rt * 16 + r16;
            f32x4 v = acc[rt] + bb;
            if (EPI == EPI_BIAS) {
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else if (EPI == EPI_GELU) {
                if (n < p.ldz) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb);
                const f32x4 g = gelu4(v);
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, g), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else if (EPI == EPI_MUL) {
                if (n < p.ldz) store_b128_guarded(__builtin_bit_cast(u32x4, v), rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb);
                f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < p.ldm) mm = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_m, (unsigned)(rl * p.ldm + 4 * q) * 4u, nb, 0));
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, v * mm), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            } else {
                f32x4 zz = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < p.ldz) zz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, (unsigned)(rl * p.ldz + 4 * q) * 4u, nb, 0));
                const f32x4 g = gelu_grad4_times(zz, acc[rt]);
                if (n < p.ldy) store_b128_guarded(__builtin_bit_cast(u32x4, g), rs_y, (unsigned)(rl * p.ldy + 4 * q) * 4u, nb);
            }
        }
    };
    auto compute = [&](const u32x4 (&wh)[G], const u32x4 (&wl)[G], int f) {
        const int itl = f / ngrp, g = f - itl * ngrp;
        it = wave + RX_WAVES * itl;
        if (g == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int cc = g * G + j;
            if (cc < K32) {                                               // uniform; only the tile's last group can be short
                const bf16x8 ahi = __builtin_bit_cast(bf16x8, wh[j]);
                const bf16x8 alo = __builtin_bit_cast(bf16x8, wl[j]);
                const float* xb = xres + (size_t)(cc * WIMG) * BM * 16 + r16 * 16 + rslot;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xb + rt * 256);
                    if (MMA == 2) {
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xb + BM * 16 + rt * 256);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, xh, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, xl, acc[rt], 0, 0, 0);
                    }
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, xh, acc[rt], 0, 0, 0);
                }
            }
        }
        if (g == ngrp - 1 && !(NRM_DIAG_RX & 4)) epilogue();
        if (g == ngrp - 1 && (NRM_DIAG_RX & 4)) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) asm volatile("" :: "v"(acc[rt]));
        }
    };
    if (total > 0) { wload(whA, wlA, 0); wload(whB, wlB, 1); }
    for (int f = 0; f < total; f += 3) {
        if (!(NRM_DIAG_RX & 2)) wload(whC, wlC, f + 2);
        compute(whA, wlA, f);
        if (f + 1 < total) {
            if (!(NRM_DIAG_RX & 2)) wload(whA, wlA, f + 3);
            compute(whB, wlB, f + 1);
        }
        if (f + 2 < total) {
            if (!(NRM_DIAG_RX & 2)) wload(whB, wlB, f + 4);
            compute(whC, wlC, f + 2);
        }
    }
#endif
}

int gemm_nt_rx_bm(int M, int K, int mma) {
    if (mma != 1 && mma != 2) return 0;
    const int k32 = (K + 31) / 32, wimg = mma == 2 ? 2 : 1;
    // (the RT = 4 form once returned wrong rows 12-15 / 28-31 of its block: the store-data hazard of common.hpp's
    // store_b128_guarded, fixed there; tests/test_gpu_dense.py runs 64-row blocks at M = 15360 and 51200)
    int bm = 128;
    if (const char* e = getenv("NRM_RX_BM")) bm = atoi(e);
    while (bm >= 16 && (long)k32 * wimg * bm * 64 > RX_LDS_BUDGET) bm /= 2;
    if (bm < 16) return 0;                                               // K too wide for a resident row block
    // two workgroups per CU (the conversion prologue of one under the MFMAs of the other) when that keeps >= 32 rows per block
    while (bm > 32 && (long)k32 * wimg * bm * 64 > RX_LDS_BUDGET / 2) bm /= 2;
    // ... and enough workgroups for (nearly) every CU: every workgroup streams ALL weight fragments from L2, so rows per
    // workgroup are what amortises that traffic (M = 15360, K = 258, N = 1032: 32 -> 64 rows, 59 -> 49 us)
    static const int min_wgs = [] { const char* e = getenv("NRM_RX_MIN_WGS"); return e ? atoi(e) : 200; }();
    while (bm > 32 && (M + bm - 1) / bm < min_wgs) bm /= 2;
    while (bm > 16 && M <= bm / 2) bm /= 2;
    return bm;
}

template <int RT, int MMA>
static hipError_t launch_rx_e(const GemmRxParams& p, int epi, size_t shm, hipStream_t st) {
    const dim3 grid((p.M + RT * 16 - 1) / (RT * 16)), block(RX_WAVES * 64);
    const bool g3 = (p.k32 + 2) / 3 * 3 < (p.k32 + 3) / 4 * 4;         // groups of 3 chunks pad less than groups of 4
#define NRM_RX(E)                                                                                                        \
    {                                                                                                                    \
        auto k = g3 ? gemm_nt_rx_kernel<RT, MMA, E, 3> : gemm_nt_rx_kernel<RT, MMA, E, 4>;                               \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RX_LDS_BUDGET);   \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL(k, grid, block, shm, st, p);                                                                  \
    }
    switch (epi) {
        case EPI_BIAS:  NRM_RX(EPI_BIAS) break;
        case EPI_GELU:  NRM_RX(EPI_GELU) break;
        case EPI_DGELU: NRM_RX(EPI_DGELU) break;
        case EPI_MUL:   NRM_RX(EPI_MUL) break;
        default: return hipErrorInvalidValue;
    }
#undef NRM_RX
    return hipGetLastError();
}

hipError_t gemm_nt_rx_launch(const GemmRxParams& p, int epi, int mma, hipStream_t st) {
    if (p.M <= 0) return hipSuccess;
    const int bm = gemm_nt_rx_bm(p.M, p.K, mma);
    if (!bm) return hipErrorInvalidValue;
    const size_t shm = (size_t)p.k32 * (mma == 2 ? 2 : 1) * bm * 64;
#define NRM_RX_RT(RT_)  return mma == 2 ? launch_rx_e<RT_, 2>(p, epi, shm, st) : launch_rx_e<RT_, 1>(p, epi, shm, st);
    switch (bm) {
        case 16:  NRM_RX_RT(1)
        case 32:  NRM_RX_RT(2)
        case 64:  NRM_RX_RT(4)
        case 128: NRM_RX_RT(8)
    }
#undef NRM_RX_RT
    return hipErrorInvalidValue;
}

int gemm_bf16_diag_flags() { return NRM_DIAG_RX ? 1024 : 0; }

hipError_t gemm_tn_bf16_launch(GemmTnParams p, const GemmTnPlan& pl, int mma, hipStream_t st) {
    p.nti = pl.nti; p.nsplit = pl.nsplit; p.rps = pl.rps;
    const dim3 grid(pl.nti * pl.ntj, (pl.nsplit + 3) / 4), block(256);
    if (mma == 2) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2>), grid, block, 0, st, p);
    else          hipLaunchKernelGGL((gemm_tn_bf16_kernel<1>), grid, block, 0, st, p);
    return hipGetLastError();
}

}  // namespace nrm
