// Pointwise history attention, forward, "resident W_p" form: fp32 MFMA (NRM_MMA_F32) and bf16 matrix cores (NRM_MMA_BF16 /
// NRM_MMA_BF16X3).  gfx950 / MI355X only.
//
// Same mathematics and the same MFMA orientation as the chunk-streaming kernel of pwattn_fwd.hip (rows m = (b,t,h) are MFMA columns, a lane ends up with
// 4 consecutive k of one row; accumulators start at the fp32 u + v; epilogue = z store, exact GELU, fc2 dot), but built
// around what changes when the contraction runs 16x (bf16) or 16/3x (bf16x3) faster than on the fp32 matrix pipe: one
// K-chunk of MFMA work (a few hundred cycles) is far shorter than the latency of the LDS-DMA that would bring the next W_p
// chunk, and a workgroup barrier per chunk then parks every wave of the CU on it (measured on the chunk-streaming forms:
// 41 % of wave cycles in s_waitcnt / s_barrier, 31 % in issue stalls, matrix pipe 16 % busy).  So here W_p does not
// stream at all:
//   * RESIDENT W_p: the output columns k are cut into `nsplit` slices whose bf16 image (hi [, lo]; all reduction chunks)
//     fits the 160 KB of LDS; a PERSISTENT workgroup loads its slice once and then walks row tiles, so the steady state
//     has no LDS-DMA, no barrier, no shared state: its 8 waves run independently, one 16-row tile at a time each;
//   * t and h do not pass through LDS either: lane (r16, q) reads the 8 + 8 floats of ITS row that its MFMA operand needs
//     straight into registers (two 16-byte loads each, one chunk ahead), forms t*h in fp32 and rounds once (bf16) or
//     splits it into hi + lo (bf16x3);
//   * with more than one slice every slice adds its partial fc2 dot to s[m] (float atomic; two addends commute exactly).
// The fp32 form (MMA == 0: v_mfma_f32_16x16x4_f32, 16-column chunks, the fp32 image of pack_wp_kernel) gains from the same
// structure for a different reason: its K loop is MFMA-bound, and without a barrier per chunk the epilogue of one wave (z
// store, GELU, fc2 dot) runs under the MFMAs of the other waves of its SIMD instead of beside their epilogues.
// timing diagnostics only (scripts/_diag/build_diag.sh): results are WRONG with any bit set
#ifndef NRM_DIAG_RW
#define NRM_DIAG_RW 0         // bit 0: no z store, bit 1: no GELU (plain sum), bit 2: t/h loaded once per tile (chunk 0 re-used),
#endif                        // bit 3: no u+v accumulator-init loads
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

constexpr int FB_WAVES = 16;      // waves per persistent workgroup: 4 per SIMD, <= 128 VGPRs each

// W_p prepack for the bf16 MFMA: packed[c][img][row][32 bf16] = W_p[row][32c .. 32c+31] rounded to bf16 (img 0) and, for
// bf16x3, the rounding remainders lo = bf16(w - float(hi)) (img 1): hi*hi + lo*hi + hi*lo reproduces the fp32 product to
// ~2^-16.  64-B rows with XOR-swizzled 16-B slots as in the fp32 image (pwattn_fwd.hip); slot s holds the 8 reduction
// positions lane quarter s feeds to v_mfma_f32_16x16x32_bf16: columns 32c + 4s + {0..3} and 32c + 16 + 4s + {0..3} -- the order
// in which the kernel holds its two 16-byte t / h loads (any order works for a dot product as long as both operands agree).
__global__ void pack_wp_bf16_kernel(const float* __restrict__ w, int ldw, int D, int rows, int kchunks32, int nimg,
                                    __bf16* __restrict__ packed) {
    const long per_chunk = (long)rows * 32;
    const long total = (long)kchunks32 * per_chunk;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 31);
        const long rc = i >> 5;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int slot = (j >> 3) ^ swz4(row);
        const int e = j & 7;
        const int d = c * 32 + 16 * (e >> 2) + 4 * slot + (e & 3);
        const float v = (row < D && d < D) ? w[(long)row * ldw + d] : 0.0f;
        const __bf16 hi = (__bf16)v;
        const long o = (long)c * nimg * per_chunk + (long)row * 32 + j;
        packed[o] = hi;
        if (nimg > 1) packed[o + per_chunk] = (__bf16)(v - (float)hi);
    }
}

hipError_t pack_wp_bf16_launch(const float* w, int ldw, int D, int mma, float* packed, hipStream_t st) {
    const RwPlan pl = pwattn_rw_plan(D, mma);
    if (pl.nts == 0) return hipErrorInvalidValue;
    const long total = (long)pl.k32 * pl.rows * 32;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wp_bf16_kernel, dim3(blocks), dim3(256), 0, st, w, ldw, D, pl.rows, pl.k32, pl.wimg,
                       reinterpret_cast<__bf16*>(packed));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// resident-W plan: NTS 16-column tiles of output columns per slice, nsplit slices
static const int kRwNts[7] = {1, 2, 3, 4, 5, 6, 8};      // <= 8 tiles: 32 accumulator VGPRs, the kernel stays under 128 (4 waves/SIMD)
constexpr int RW_LDS_BUDGET = 156 * 1024;

RwPlan pwattn_rw_plan(int D, int mma) {
    RwPlan pl;
    const int n16 = (D + 15) / 16;
    pl.k32 = mma ? (D + 31) / 32 : n16;                                 // reduction chunks: 32 columns (bf16) or 16 (fp32) per 64-B row
    pl.wimg = mma == 2 ? 2 : 1;
    const int tile_bytes = pl.k32 * pl.wimg * 1024;                    // one 16-column tile, all chunks and images
    int fit = RW_LDS_BUDGET / tile_bytes;
    if (fit < 1) { pl.nts = 0; pl.nsplit = 0; pl.rows = 0; return pl; }     // D > 2496 (bf16) / 1248 (bf16x3): not supported
    if (fit > 8) fit = 8;
    int nsplit = (n16 + fit - 1) / fit;
    for (;; ++nsplit) {
        const int need = (n16 + nsplit - 1) / nsplit;
        int nts = 0;
        for (int i = 0; i < 7; ++i) if (kRwNts[i] >= need) { nts = kRwNts[i]; break; }
        if (nts && nts <= fit) { pl.nts = nts; break; }
    }
    pl.nsplit = nsplit;
    pl.rows = pl.nsplit * pl.nts * 16;
    return pl;
}

// packed[c][img][row][32 bf16] (pack_wp_bf16_kernel) with rows = plan.rows
template <int NTS, bool SAVE_Z, int MMA>
__global__ __launch_bounds__(FB_WAVES * 64, FB_WAVES / 4) void pwattn_fwd_rw_kernel(const FwdParams p, const RwPlan pl, int wgs_per_split) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WIMG = MMA == 2 ? 2 : 1;
    constexpr int SROWS = NTS * 16;                                    // W rows (= output columns) of a slice
    constexpr int CB = MMA ? 128 : 64;                                 // bytes of a t / h row one reduction chunk covers
    extern __shared__ __attribute__((aligned(16))) float wres[];       // [chunk][WIMG][SROWS][16 floats (fp32) = 32 bf16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int M = (int)p.M, T = p.T, H = p.H, D = p.D;
    const int split = blockIdx.x % pl.nsplit, g = blockIdx.x / pl.nsplit;
    const int k0 = split * SROWS;                                       // first output column of the slice
    const int K = pl.k32;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, D * 4, 0x00020000);

    // ---- once per workgroup: the slice's W_p image -> LDS (1-KiB pieces of 16 rows), then ONE barrier
    const int npiece = K * WIMG * NTS;
    for (int pc = wave; pc < npiece; pc += FB_WAVES) {
        const int ci = pc / NTS, rt = pc - ci * NTS;                   // ci = c * WIMG + img
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wres + pc * 256), 16, lane * 16,
                                                 (ci * pl.rows + k0 + rt * 16) * 64, 0, 0);
    }
    __syncthreads();                                                    // waits for this wave's DMA (vmcnt(0)), then the barrier

    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rslot = 4 * (q ^ swz4(r16));
    const float b2 = split == 0 ? p.b2[0] : 0.f;
    const int ntile = (M + 15) >> 4;
    const int stride = wgs_per_split * FB_WAVES;
    const bool ragged = D % (CB / 4) != 0;

    for (int tile = g * FB_WAVES + wave; tile < ntile; tile += stride) {
        const int m = tile * 16 + r16;
        const unsigned mm = m < M ? (unsigned)m : 0u;
        const unsigned bt = mm / (unsigned)H;
        const unsigned hr = (bt / (unsigned)T) * H + (mm - bt * H);
        const unsigned voff_t = m < M ? (bt * p.ldt + 4 * q) * 4u : OOB;
        const unsigned voff_h = m < M ? (hr * p.ldh + 4 * q) * 4u : OOB;
        const unsigned voff_u = m < M ? (hr * p.ldu + 4 * q) * 4u : OOB;
        const unsigned voff_v = m < M ? (bt * p.ldv + 4 * q) * 4u : OOB;

        f32x4 ta[3], tb[3], ha[3], hb[3];                              // three chunk sets of this lane's operand columns (requested two chunks ahead)
        auto load_th = [&](int c, int set) {
            if ((NRM_DIAG_RW & 4) && c > 1) return;
            // ragged last chunk (D not a multiple of the chunk width): columns >= D belong to the NEXT row -- they would be
            // multiplied by the zero padding of W_p, but a NaN / Inf there must not reach this row's scores: read 0 instead
            unsigned vt = voff_t, vh = voff_h, vt2 = voff_t, vh2 = voff_h;
            if (ragged) {
                const int col = (CB / 4) * c + 4 * q;
                if (col >= D) vt = vh = OOB;
                if (col + 16 >= D) vt2 = vh2 = OOB;
            }
            ta[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, vt, c * CB, 0));
            ha[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vh, c * CB, 0));
            if (MMA) {
                tb[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, vt2, c * CB + 64, 0));
                hb[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vh2, c * CB + 64, 0));
            }
        };
        load_th(0, 0);
        // accumulators start at u[b,h,k] + v[b,t,k]
        f32x4 acc[NTS];
#pragma unroll
        for (int it = 0; it < NTS; ++it) {
            const int kb = (k0 + it * 16) * 4;
            const f32x4 uu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u, kb, 0));
            const f32x4 vv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, voff_v, kb, 0));
            acc[it] = (k0 + it * 16 + 4 * q < D) ? uu + vv : zero4;
            if (NRM_DIAG_RW & 8) acc[it] = zero4;
        }

        auto compute_f32 = [&](int c, int set) {
            const f32x4 pf = ta[set] * ha[set];                         // this lane's 4 reduction columns of P[m,:] = t * h
            const float* buf = wres + c * (SROWS * 16);
            auto rd = [&](int it) { return *reinterpret_cast<const f32x4*>(&buf[(it * 16 + r16) * 16 + rslot]); };
            f32x4 af = rd(0);
#pragma unroll
            for (int it = 0; it < NTS; ++it) {
                f32x4 afn = af;
                if (it + 1 < NTS) afn = rd(it + 1);                     // one tile (4 MFMAs = 128 cycles) ahead
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[it] = mfma16(af[j], pf[j], acc[it]);
                af = afn;
            }
        };
        auto compute = [&](int c, int set) {
            if (MMA == 0) { compute_f32(c, set); return; }
            const f32x4 lo = ta[set] * ha[set], hi = tb[set] * hb[set];
            bf16x8 pf, pl2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = (__bf16)lo[e]; pf[4 + e] = (__bf16)hi[e];
                if (MMA == 2) { pl2[e] = (__bf16)(lo[e] - (float)pf[e]); pl2[4 + e] = (__bf16)(hi[e] - (float)pf[4 + e]); }
            }
            const float* buf = wres + c * (WIMG * SROWS * 16);
            // W fragments are read LA tiles ahead of their MFMAs (a 16x16x32 MFMA issues every 16 cycles, an LDS read takes 64+)
            constexpr int LA = MMA == 2 ? 3 : 6;
            auto rd = [&](int it, int im) { return *reinterpret_cast<const bf16x8*>(&buf[(im * SROWS + it * 16 + r16) * 16 + rslot]); };
            bf16x8 af[NTS], al[NTS];
#pragma unroll
            for (int it = 0; it < LA && it < NTS; ++it) { af[it] = rd(it, 0); if (MMA == 2) al[it] = rd(it, 1); }
#pragma unroll
            for (int it = 0; it < NTS; ++it) {
                if (it + LA < NTS) { af[it + LA] = rd(it + LA, 0); if (MMA == 2) al[it + LA] = rd(it + LA, 1); }
                if (MMA == 2) {                                        // small terms first
                    acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[it], pf, acc[it], 0, 0, 0);
                    acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], pl2, acc[it], 0, 0, 0);
                }
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], pf, acc[it], 0, 0, 0);
            }
        };
        if (1 < K) load_th(1, 1);
        for (int c = 0; c < K; c += 3) {                                // unrolled by the ring length: set indices are constants
            if (c + 2 < K) load_th(c + 2, 2);
            compute(c, 0);
            if (c + 1 >= K) break;
            if (c + 3 < K) load_th(c + 3, 0);
            compute(c + 1, 1);
            if (c + 2 >= K) break;
            if (c + 4 < K) load_th(c + 4, 1);
            compute(c + 2, 2);
        }

        // ---- epilogue: optional z store ; GELU ; partial fc2 dot over the slice's columns
        const int rows_here = min(16, M - tile * 16);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
            SAVE_Z ? p.z + (size_t)tile * 16 * D : nullptr, 0, SAVE_Z ? rows_here * D * 4 : 0, 0x00020000);
        // fc2 weights of the slice's columns (0 beyond D; L1-resident): requested together, one round trip
        f32x4 ww[NTS];
#pragma unroll
        for (int it = 0; it < NTS; ++it)
            ww[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, 16 * q, (k0 + it * 16) * 4, 0));
        float s_part = 0.f;
#pragma unroll
        for (int it = 0; it < NTS; ++it) {
            const f32x4 zz = acc[it];
            if (SAVE_Z && !(NRM_DIAG_RW & 1) && k0 + it * 16 + 4 * q < D)
                store_b128_guarded(__builtin_bit_cast(u32x4, zz), rs_z, (unsigned)(r16 * D + 4 * q) * 4u, (k0 + it * 16) * 4);
            if (NRM_DIAG_RW & 2) s_part += ww[it][0] * zz[0] + ww[it][1] * zz[1] + ww[it][2] * zz[2] + ww[it][3] * zz[3];
            else s_part += gelu_dot4(ww[it], zz);
        }
        const float v = sum_rows4(s_part) + b2;
        if (q == 0 && m < M) {
            if (pl.nsplit == 1) p.s[m] = v; else atomicAdd(p.s + m, v);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// "walk" form of the resident-W forward for the bf16 arithmetics at D = 64 / 128 / 256 (whole 32-wide chunks).  The kernel
// above takes 16-row tiles of the flattened rows m = (b,t,h) one after the other and re-loads, for every tile, its h and u rows
// (each shared by all T candidates of the impression) next to t and v: 56 vector-memory instructions per 192 MFMAs, two
// thirds of them for operands that do not change along t (timing-only builds: 0.077 ms of MFMA work in a 0.32 ms kernel).
// Here a wave owns (impression b, 16 history rows) and WALKS the candidates t, as the resident-W backward does:
//   * its h rows (the t*h operand's factor) and u rows (the accumulators' start) are loaded ONCE per task and stay in
//     registers for the whole walk;
//   * per step only the candidate's t row (two 16-byte loads per chunk, the same for all 16 rows: L1 broadcasts) and, in the
//     epilogue, its v and fc2-weight segments are read; t chunks are requested two chunks ahead, across step boundaries;
//   * z = (u + P W_p^T) + v: the store, GELU and fc2 dot are the kernel above's.
template <int NTS, bool SAVE_Z, int MMA, int KCH>
__global__ __launch_bounds__(512, 2) void pwattn_fwd_walk_kernel(const FwdParams p, const RwPlan pl, int wgs_per_split, int tsplit) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WIMG = MMA == 2 ? 2 : 1;
    constexpr int SROWS = NTS * 16;
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) float wres[];       // [KCH][WIMG][SROWS][16 floats = 32 bf16]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = p.T, H = p.H, D = p.D;
    const int split = blockIdx.x % pl.nsplit, g = blockIdx.x / pl.nsplit;
    const int k0 = split * SROWS;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, D * 4, 0x00020000);
    const int npiece = KCH * WIMG * NTS;
    for (int pc = wave; pc < npiece; pc += NW) {
        const int ci = pc / NTS, rt = pc - ci * NTS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wres + pc * 256), 16, lane * 16,
                                                 (ci * pl.rows + k0 + rt * 16) * 64, 0, 0);
    }
    __syncthreads();

    const int rslot = 4 * (q ^ swz4(r16));
    const float b2 = split == 0 ? p.b2[0] : 0.f;
    const int nht = (H + 15) >> 4;
    const int tlen = (T + tsplit - 1) / tsplit;
    const int ntask = (int)(p.M / ((long)T * H)) * nht * tsplit;         // B * nht * tsplit

    for (int task = g * NW + wave; task < ntask; task += wgs_per_split * NW) {
        const int tp = task % tsplit;
        const int rest = task / tsplit;
        const int b = rest / nht, h0 = (rest - b * nht) * 16;
        const int t_lo = tp * tlen, t_hi = min(T, t_lo + tlen);
        if (t_lo >= t_hi) continue;
        const bool rok = h0 + r16 < H;
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t) + (size_t)b * T * p.ldt, 0, (unsigned)(T * p.ldt * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v) + (size_t)b * T * p.ldv, 0, (unsigned)(T * p.ldv * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h) + (size_t)b * H * p.ldh, 0, (unsigned)(H * p.ldh * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u) + (size_t)b * H * p.ldu, 0, (unsigned)(H * p.ldu * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
            SAVE_Z ? p.z + (size_t)b * T * H * D : nullptr, 0, SAVE_Z ? (unsigned)((size_t)T * H * D * 4) : 0, 0x00020000);

        // ---- once per task: this lane's h row (its 8 reduction columns of every chunk) and u row (its 4 columns of every tile)
        f32x4 ha[KCH], hb[KCH], ureg[NTS];
        const unsigned vh = rok ? (unsigned)(((h0 + r16) * p.ldh + 4 * q) * 4) : OOB;
        const unsigned vu = rok ? (unsigned)(((h0 + r16) * p.ldu + k0 + 4 * q) * 4) : OOB;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            ha[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vh, c * 128, 0));
            hb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vh, c * 128 + 64, 0));
        }
#pragma unroll
        for (int it = 0; it < NTS; ++it)
            ureg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, vu, it * 64, 0));

        // t operand ring: chunk c of a step lives in slot c & 1 and is requested two chunks ahead (KCH is even)
        f32x4 ta[2], tb[2];
        const unsigned vt = (unsigned)(4 * q * 4);
        auto load_t = [&](int slot, int t, int c) {
            ta[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, vt, t * p.ldt * 4 + c * 128, 0));
            tb[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, vt, t * p.ldt * 4 + c * 128 + 64, 0));
        };
        load_t(0, t_lo, 0);
        load_t(1, t_lo, 1);
        const unsigned vv_off = (unsigned)((k0 + 4 * q) * 4);
        const unsigned vz = rok ? (unsigned)(((h0 + r16) * D + k0 + 4 * q) * 4) : OOB;

        for (int t = t_lo; t < t_hi; ++t) {
            f32x4 acc[NTS];
#pragma unroll
            for (int it = 0; it < NTS; ++it) acc[it] = ureg[it];
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int slot = c & 1;
                const f32x4 lo = ta[slot] * ha[c], hi = tb[slot] * hb[c];
                // the slot is free: chunk c + 2 of this step, or chunk c + 2 - KCH of the next
                {
                    const int cn = c + 2;
                    const int tn = t + (cn >= KCH ? 1 : 0);
                    if (tn < t_hi) load_t(slot, tn, cn >= KCH ? cn - KCH : cn);
                }
                bf16x8 pf, pl2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[e] = (__bf16)lo[e]; pf[4 + e] = (__bf16)hi[e];
                    if (MMA == 2) { pl2[e] = (__bf16)(lo[e] - (float)pf[e]); pl2[4 + e] = (__bf16)(hi[e] - (float)pf[4 + e]); }
                }
                const float* buf = wres + c * (WIMG * SROWS * 16);
                constexpr int LA = NTS < 3 ? NTS : 3;
                auto rd = [&](int it, int im) { return *reinterpret_cast<const bf16x8*>(&buf[(im * SROWS + it * 16 + r16) * 16 + rslot]); };
                bf16x8 af[NTS], al[NTS];
#pragma unroll
                for (int it = 0; it < LA; ++it) { af[it] = rd(it, 0); if (MMA == 2) al[it] = rd(it, 1); }
#pragma unroll
                for (int it = 0; it < NTS; ++it) {
                    if (it + LA < NTS) { af[it + LA] = rd(it + LA, 0); if (MMA == 2) al[it + LA] = rd(it + LA, 1); }
                    if (MMA == 2) {
                        acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[it], pf, acc[it], 0, 0, 0);
                        acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], pl2, acc[it], 0, 0, 0);
                    }
                    acc[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[it], pf, acc[it], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- epilogue: z = acc + v ; optional z store ; GELU ; partial fc2 dot.  v and fc2-weight segments of tile it + 2
            // are requested while tile it is processed.
            const int m = (b * T + t) * H + h0 + r16;
            f32x4 vvs[2], wws[2];
            auto load_vw = [&](int it, int set) {
                vvs[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vv_off, t * p.ldv * 4 + it * 64, 0));
                wws[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, 16 * q, (k0 + it * 16) * 4, 0));
            };
            load_vw(0, 0);
            if (NTS > 1) load_vw(1, 1);
            float s_part = 0.f;
#pragma unroll
            for (int it = 0; it < NTS; ++it) {
                const f32x4 zz = acc[it] + vvs[it & 1];
                const f32x4 ww = wws[it & 1];
                if (it + 2 < NTS) load_vw(it + 2, it & 1);
                if (SAVE_Z) store_b128_guarded(__builtin_bit_cast(u32x4, zz), rs_z, vz, t * H * D * 4 + it * 64);
                s_part += gelu_dot4(ww, zz);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float sv = sum_rows4(s_part) + b2;
            if (q == 0 && rok) {
                if (pl.nsplit == 1) p.s[m] = sv; else atomicAdd(p.s + m, sv);
            }
        }
    }
#endif
}

// The same walk on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, the fp32 image of pack_wp_kernel: 16-column chunks), round 5, for
// the widths whose whole W_p is one resident slice and whose h / u rows fit the registers beside the accumulators (D = 64: the
// reference's default sizes, configs/model_config.py:29; D = 128).  The tile-by-tile kernel above re-loads h, u and the fc2
// weights for every candidate: 24 vector-memory instructions per 64 MFMAs at D = 64, 0.113 ms per launch at the reference's default
// sizes (0.35 of the fp32 MFMA peak) for 0.04 ms of MFMA work and 0.033 ms of z store.  Here a wave keeps its 16 history rows (D / 4
// floats per lane), its u tile and the fc2 weights in registers and walks the candidates: 4 t loads (two chunks ahead, across step
// boundaries), NTS v loads and NTS z stores per step.  Same arithmetic per element as the kernel above (accumulators start at u, v
// is added in the epilogue: (u + P W_p^T) + v instead of (u + v) + P W_p^T -- one fp32 rounding apart).
template <int NTS, bool SAVE_Z, int KCH>
__global__ __launch_bounds__(512, 2) void pwattn_fwd_walk_f32_kernel(const FwdParams p, const RwPlan pl, int wgs, int tsplit) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(KCH % 2 == 0 && NTS == KCH, "whole width in one slice: D = 16 KCH = 16 NTS");
    constexpr int SROWS = NTS * 16;
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) float wres[];       // [KCH][SROWS][16 floats]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = p.T, H = p.H, D = p.D;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, D * 4, 0x00020000);
    for (int pc = wave; pc < KCH * NTS; pc += NW) {
        const int c = pc / NTS, rt = pc - c * NTS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wres + pc * 256), 16, lane * 16,
                                                 (c * pl.rows + rt * 16) * 64, 0, 0);
    }
    __syncthreads();

    const int rslot = 4 * (q ^ swz4(r16));
    const float b2 = p.b2[0];
    const int nht = (H + 15) >> 4;
    const int tlen = (T + tsplit - 1) / tsplit;
    const int ntask = (int)(p.M / ((long)T * H)) * nht * tsplit;         // B * nht * tsplit
    f32x4 wreg[NTS];                                                     // fc2 weights of this lane's four columns of every tile
#pragma unroll
    for (int it = 0; it < NTS; ++it)
        wreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, 16 * q, it * 64, 0));

    for (int task = blockIdx.x * NW + wave; task < ntask; task += wgs * NW) {
        const int tp = task % tsplit;
        const int rest = task / tsplit;
        const int b = rest / nht, h0 = (rest - b * nht) * 16;
        const int t_lo = tp * tlen, t_hi = min(T, t_lo + tlen);
        if (t_lo >= t_hi) continue;
        const bool rok = h0 + r16 < H;
        const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t) + (size_t)b * T * p.ldt, 0, (unsigned)(T * p.ldt * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v) + (size_t)b * T * p.ldv, 0, (unsigned)(T * p.ldv * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h) + (size_t)b * H * p.ldh, 0, (unsigned)(H * p.ldh * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u) + (size_t)b * H * p.ldu, 0, (unsigned)(H * p.ldu * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
            SAVE_Z ? p.z + (size_t)b * T * H * D : nullptr, 0, SAVE_Z ? (unsigned)((size_t)T * H * D * 4) : 0, 0x00020000);

        // once per task: this lane's h row (its 4 reduction columns of every chunk) and u row (its 4 columns of every tile)
        f32x4 hreg[KCH], ureg[NTS];
        const unsigned vh = rok ? (unsigned)(((h0 + r16) * p.ldh + 4 * q) * 4) : OOB;
        const unsigned vu = rok ? (unsigned)(((h0 + r16) * p.ldu + 4 * q) * 4) : OOB;
#pragma unroll
        for (int c = 0; c < KCH; ++c) hreg[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_h, vh, c * 64, 0));
#pragma unroll
        for (int it = 0; it < NTS; ++it) ureg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, vu, it * 64, 0));

        // t operand ring: chunk c of a step lives in slot c & 1 and is requested two chunks ahead (KCH is even)
        f32x4 tr[2];
        const unsigned vt = (unsigned)(4 * q * 4);
        auto load_t = [&](int slot, int t, int c) {
            tr[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, vt, t * p.ldt * 4 + c * 64, 0));
        };
        load_t(0, t_lo, 0);
        load_t(1, t_lo, 1);
        const unsigned vv_off = (unsigned)(4 * q * 4);
        const unsigned vz = rok ? (unsigned)(((h0 + r16) * D + 4 * q) * 4) : OOB;

        for (int t = t_lo; t < t_hi; ++t) {
            f32x4 acc[NTS];
#pragma unroll
            for (int it = 0; it < NTS; ++it) acc[it] = ureg[it];
            // this candidate's v segments: requested before the MFMAs, consumed in the epilogue
            f32x4 vreg[NTS];
#pragma unroll
            for (int it = 0; it < NTS; ++it)
                vreg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vv_off, t * p.ldv * 4 + it * 64, 0));
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const int slot = c & 1;
                const f32x4 pf = tr[slot] * hreg[c];                    // this lane's 4 reduction columns of P[m,:] = t * h
                {
                    const int cn = c + 2;
                    const int tn = t + (cn >= KCH ? 1 : 0);
                    if (tn < t_hi) load_t(slot, tn, cn >= KCH ? cn - KCH : cn);
                }
                const float* buf = wres + c * (SROWS * 16);
                auto rd = [&](int it) { return *reinterpret_cast<const f32x4*>(&buf[(it * 16 + r16) * 16 + rslot]); };
                f32x4 af = rd(0);
#pragma unroll
                for (int it = 0; it < NTS; ++it) {
                    f32x4 afn = af;
                    if (it + 1 < NTS) afn = rd(it + 1);                 // one tile (4 MFMAs = 128 cycles) ahead
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[it] = mfma16(af[j], pf[j], acc[it]);
                    __builtin_amdgcn_sched_barrier(0);
                    af = afn;
                }
            }
            // epilogue: z = acc + v ; optional z store ; GELU ; fc2 dot
            const int m = (b * T + t) * H + h0 + r16;
            float s_part = 0.f;
#pragma unroll
            for (int it = 0; it < NTS; ++it) {
                const f32x4 zz = acc[it] + vreg[it];
                if (SAVE_Z) store_b128_guarded(__builtin_bit_cast(u32x4, zz), rs_z, vz, t * H * D * 4 + it * 64);
                const f32x4 ww = wreg[it];
                s_part += gelu_dot4(ww, zz);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float sv = sum_rows4(s_part) + b2;
            if (q == 0 && rok) p.s[m] = sv;
        }
    }
#endif
}

static int rw_cus() {       // of the CURRENT device (a process may drive several): queried per launch, not cached
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
}

template <int NTS>
static hipError_t launch_rw(const FwdParams& p, const RwPlan& pl, int mma, hipStream_t st) {
    const int ntile = (int)((p.M + 15) / 16);
    if (ntile <= 0) return hipSuccess;
    int wgs = rw_cus() / pl.nsplit;                                     // one persistent workgroup per CU, CUs shared evenly by the slices
    if (wgs < 1) wgs = 1;
    if ((long)wgs * FB_WAVES > ntile) wgs = (ntile + FB_WAVES - 1) / FB_WAVES;
    const size_t shm = (size_t)pl.k32 * pl.wimg * NTS * 1024;
    if (pl.nsplit > 1) {                                                // the slices ADD their partial scores
        hipError_t e = hipMemsetAsync(p.s, 0, (size_t)p.M * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)(wgs * pl.nsplit)), block(FB_WAVES * 64);
#define NRM_RW(SZ, M_)                                                                                                   \
    {                                                                                                                    \
        auto k = pwattn_fwd_rw_kernel<NTS, SZ, M_>;                                                                      \
        /* per launch: the attribute is per device, and a process may launch on more than one */                        \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS_BUDGET);  \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL(k, grid, block, shm, st, p, pl, wgs);                                                         \
    }
    if (mma == 2)      { if (p.z) NRM_RW(true, 2) else NRM_RW(false, 2) }
    else if (mma == 1) { if (p.z) NRM_RW(true, 1) else NRM_RW(false, 1) }
    else               { if (p.z) NRM_RW(true, 0) else NRM_RW(false, 0) }
#undef NRM_RW
    return hipGetLastError();
}

int pwattn_fwd_rw_diag_flags() { return NRM_DIAG_RW ? 4 : 0; }

template <int NTS, int KCH>
static hipError_t launch_walk(const FwdParams& p, const RwPlan& pl, int mma, hipStream_t st) {
    const int B = (int)(p.M / ((long)p.T * p.H)), nht = (p.H + 15) / 16;
    const long base = (long)B * nht;
    if (base <= 0) return hipSuccess;
    int wgs = rw_cus() / pl.nsplit;
    if (wgs < 1) wgs = 1;
    int tsplit = 1;
    if (const char* e = getenv("NRM_FWD_TSPLIT")) tsplit = atoi(e);
    else while (base * tsplit < 2L * wgs * 8 && p.T / (tsplit + 1) >= 8) ++tsplit;
    if (tsplit < 1) tsplit = 1;
    if (tsplit > p.T) tsplit = p.T;
    const long ntask = base * tsplit;
    if ((long)wgs * 8 > ntask) wgs = (int)((ntask + 7) / 8);
    const size_t shm = (size_t)pl.k32 * pl.wimg * NTS * 1024;
    if (pl.nsplit > 1) {
        hipError_t e = hipMemsetAsync(p.s, 0, (size_t)p.M * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)(wgs * pl.nsplit)), block(512);
#define NRM_WALK(SZ, M_)                                                                                                 \
    {                                                                                                                    \
        auto k = pwattn_fwd_walk_kernel<NTS, SZ, M_, KCH>;                                                               \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RW_LDS_BUDGET);  \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL(k, grid, block, shm, st, p, pl, wgs, tsplit);                                                 \
    }
    if (mma == 2) { if (p.z) NRM_WALK(true, 2) else NRM_WALK(false, 2) }
    else          { if (p.z) NRM_WALK(true, 1) else NRM_WALK(false, 1) }
#undef NRM_WALK
    return hipGetLastError();
}

template <int NTS>
static hipError_t launch_walk_f32(const FwdParams& p, const RwPlan& pl, hipStream_t st) {
    const int B = (int)(p.M / ((long)p.T * p.H)), nht = (p.H + 15) / 16;
    const long base = (long)B * nht;
    if (base <= 0) return hipSuccess;
    int wgs = 2 * rw_cus();                                             // two workgroups of 8 waves per CU
    int tsplit = 1;
    if (const char* e = getenv("NRM_FWD_TSPLIT")) tsplit = atoi(e);
    else while (base * tsplit < 2L * wgs * 8 && p.T / (tsplit + 1) >= 4) ++tsplit;
    if (tsplit < 1) tsplit = 1;
    if (tsplit > p.T) tsplit = p.T;
    const long ntask = base * tsplit;
    if ((long)wgs * 8 > ntask) wgs = (int)((ntask + 7) / 8);
    const size_t shm = (size_t)pl.k32 * NTS * 1024;
    const dim3 grid((unsigned)wgs), block(512);
    if (p.z) hipLaunchKernelGGL((pwattn_fwd_walk_f32_kernel<NTS, true, NTS>), grid, block, shm, st, p, pl, wgs, tsplit);
    else     hipLaunchKernelGGL((pwattn_fwd_walk_f32_kernel<NTS, false, NTS>), grid, block, shm, st, p, pl, wgs, tsplit);
    return hipGetLastError();
}

hipError_t pwattn_fwd_rw_launch(const FwdParams& p, int mma, hipStream_t st) {
    const RwPlan pl = pwattn_rw_plan(p.D, mma);
    static const bool walk = [] { const char* e = getenv("NRM_FWD_WALK"); return !(e && e[0] == '0'); }();
    // fp32: the walk for the widths that are one resident slice of whole 16-column tiles (NRM_FWD_WALK_F32=0|1 forces per launch)
    if (mma == 0 && pl.nsplit == 1 && p.D == pl.nts * 16 && p.M % ((long)p.T * p.H) == 0 && (long)p.T * p.H * p.D * 4 < (1L << 31)) {
        const char* e = getenv("NRM_FWD_WALK_F32");
        if ((e ? e[0] == '1' : true) && p.D == 64) return launch_walk_f32<4>(p, pl, st);
        // D = 128: h + u + accumulators + v take 196 VGPRs (two waves per SIMD): not the default, NRM_FWD_WALK_F32=1 selects it
        if (e && e[0] == '1' && p.D == 128) return launch_walk_f32<8>(p, pl, st);
    }
    if (walk && mma != 0 && p.M % ((long)p.T * p.H) == 0 && (long)p.T * p.H * p.D * 4 < (1L << 31)) {
        if (p.D == 256 && pl.nts == 8) return launch_walk<8, 8>(p, pl, mma, st);
        if (p.D == 128 && pl.nts == 8) return launch_walk<8, 4>(p, pl, mma, st);
        if (p.D == 64 && pl.nts == 4) return launch_walk<4, 2>(p, pl, mma, st);
    }
    switch (pl.nts) {
        case 1:  return launch_rw<1>(p, pl, mma, st);
        case 2:  return launch_rw<2>(p, pl, mma, st);
        case 3:  return launch_rw<3>(p, pl, mma, st);
        case 4:  return launch_rw<4>(p, pl, mma, st);
        case 5:  return launch_rw<5>(p, pl, mma, st);
        case 6:  return launch_rw<6>(p, pl, mma, st);
        case 8:  return launch_rw<8>(p, pl, mma, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace nrm
