// Pointwise (DIN-style) history attention, forward.  gfx950 / MI355X only.
//
// Reference semantics (models/attention_model.py:52-97): for every impression b, candidate t and
// history item h
//     score[b,t,h] = fc2( GELU( fc1( cat[h, t, t-h, t*h] ) ) )
// with fc1 = [D x 4D] (+bias) and fc2 = [1 x D] (+bias).  The [B,T,H,4D] concat is never built
// here: splitting fc1 = [W_h | W_t | W_d | W_p] gives
//     z[b,t,h,k] = u[b,h,k] + v[b,t,k] + sum_d W_p[k,d] * t[b,t,d] * h[b,h,d]
//     u = h (W_h - W_d)^T + bias1,   v = t (W_t + W_d)^T            (two small GEMMs, done by the caller)
// so the heavy part is ONE GEMM over the flattened rows m = (b,t,h):  Z[M x D] = P[M x D] * W_p^T
// whose A operand P[m,:] = t[b,t,:] * h[b,h,:] is formed on the fly while staging into LDS.
//
// Kernel shape: a workgroup (4 waves) owns BM = 64*MT consecutive rows m and ALL output columns k
// (in chunks of 16*NT), so the GELU -> fc2 dot -> score reduction over k finishes on chip.
// MFMA orientation is "transposed": the MFMA row index is k, the MFMA column index is the data row m,
// so every lane ends up holding 4 consecutive k of one row m -> float4 loads of u/v and float4 stores
// of z.  All MFMA work is v_mfma_f32_16x16x4_f32 (exact f32).
#include <cstdlib>
// timing diagnostics only (scripts/_diag/build_diag.sh): results are WRONG with any of them set
#ifndef XCD_REMAP
#define XCD_REMAP 1
#endif
#ifndef NRM_DIAG_FWD
#define NRM_DIAG_FWD 0        // bit 0: no accumulator-init loads, bit 1: no K-chunk DMA after the first, bit 2: no z store,
#endif                        // bit 3: no GELU / fc2 dot (plain sum instead)
#ifndef NRM_FWD_NW8
#define NRM_FWD_NW8 0      // tuning: 13x1 compact-image forward as 8-wave workgroups (128 rows share a W chunk)
#endif
#ifndef NRM_PRIO
#define NRM_PRIO 0        // tuning: s_setprio level of the K loops (0: none)
#endif
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

// ---------------------------------------------------------------------------------------------
// W_p prepack: packed[c][row][16] = W_p[row][16c .. 16c+15], zero padded to `rows` rows and to a
// multiple of 16 columns, so that one K-chunk of one N-chunk is a single contiguous block that LDS-DMA
// copies verbatim.  Inside a 64-B row the four 16-B slots are XOR-swizzled (slot s holds columns
// 4*(s ^ swz4(row)) ..+3): the unpadded LDS image is then read conflict-free with ds_read_b128.
// W_p = fc1.weight[:, 3D:4D] (row stride ldw = 4D).
__global__ void pack_wp_kernel(const float* __restrict__ w, int ldw, int D, int rows, int kchunks,
                               float* __restrict__ packed) {
    const long total = (long)kchunks * rows * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const long rc = i >> 4;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int slot = (j >> 2) ^ swz4(row);
        const int d = c * 16 + 4 * slot + (j & 3);
        packed[i] = (row < D && d < D) ? w[(long)row * ldw + d] : 0.0f;
    }
}

// CT ("compact t", round 4): the 64 flattened rows (b,t,h) of a workgroup -- h fastest -- belong to at most floor(63 / H) + 2
// candidates, yet every wave used to stage the candidate row of EACH of its 16 rows (a 1-KiB LDS-DMA piece of mostly identical
// 64-byte segments per wave and K-chunk, 4 KB of LDS per stage).  With CT one wave stages the workgroup's DISTINCT candidate rows
// (one piece: row r of the image = candidate bt0 + r, rows past the last one read 0) and every lane reads its row's fragment
// from image row (m / H) - bt0 (a broadcast: <= 2 distinct addresses per 16-lane read group for H >= 16).  Three LDS-DMA
// pieces per workgroup and chunk fewer, 35 KB instead of 43 KB of LDS at 13x1: FOUR workgroups per CU instead of three
// (launch bounds 4 waves per SIMD: 121 VGPRs).  Used for H >= 16 (pwattn_fwd_launch); shorter histories keep the per-row image.
template <int NT, int MT, bool SAVE_Z, int WPE = 2, bool CT = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, WPE) void pwattn_fwd_kernel(const FwdParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // buffer-descriptor types/builtins exist in the device pass only;
                                         // without the guard the host pass silently drops the kernel stub
    constexpr int BM = NW * MT * 16;      // data rows per workgroup (NW waves x MT row tiles of 16)
    constexpr int WROWS = NT * 16;
    // one LDS buffer = [W chunk | t rows | h rows], every row 16 floats (64 B), unpadded (LDS-DMA image); CT: [W chunk | h rows |
    // 16 rows of distinct candidate rows]
    constexpr int BUF = (WROWS + (CT ? BM + 16 : 2 * BM)) * 16;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];       // double buffered

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int M = (int)p.M;                       // host guarantees M < 2^31
    // XCD-aware tile order: the hardware deals consecutive block ids round-robin over the 8 XCDs (each with its own L2);
    // consecutive ROW TILES share their impression's h / u rows and their candidate's t / v row, so give each XCD a
    // contiguous range of tiles (bijective for any grid size) instead of every eighth one.
    const int nblk = gridDim.x, lin = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
    const int tile_id = XCD_REMAP ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3) : lin;
    const int m0 = tile_id * BM;
    const int T = p.T, H = p.H, D = p.D;
    const int rows_here = min(BM, M - m0);

    // All global traffic goes through buffer descriptors: 32-bit lane offsets (no 64-bit pointers to keep
    // live or spill) and hardware bounds checking -- an offset >= num_records reads 0 / drops the store,
    // which is how out-of-range rows (m >= M) are masked.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, p.wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, D * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        SAVE_Z ? p.z + (size_t)m0 * D : nullptr, 0, SAVE_Z ? rows_here * D * 4 : 0, 0x00020000);

    // --- staging (LDS-DMA): a wave copies its OWN MT*16 t rows and h rows (one 1-KiB piece = 16 rows x 64 B
    // per wave-instruction, lane -> row lane>>2, 16-B slot lane&3) and every 4th 1-KiB piece of the W chunk.
    // The source slot is XOR-swizzled with swz4(row) (common.hpp); readers apply the same XOR.
    unsigned voff_t[MT], voff_h[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int rl = lane >> 2;
        const int m = m0 + (wave * MT + j) * 16 + rl;
        const unsigned mm = m < M ? (unsigned)m : 0u;
        const unsigned bt = mm / (unsigned)H;
        const unsigned b = bt / (unsigned)T;
        const unsigned hr = b * H + (mm - bt * H);
        const unsigned slot = (unsigned)((lane & 3) ^ swz4(rl));
        voff_t[j] = (!CT && m < M) ? (bt * p.ldt + 4 * slot) * 4u : OOB;
        voff_h[j] = m < M ? (hr * p.ldh + 4 * slot) * 4u : OOB;
    }
    // CT: image row r (lane >> 2) holds candidate bt0 + r while that candidate has rows in this block (slots swizzled as in every
    // image); a lane's fragment comes from image row tix = (its row's candidate) - bt0
    const unsigned bt0 = (unsigned)m0 / (unsigned)H;
    const unsigned bt_last = (unsigned)(m0 + rows_here - 1) / (unsigned)H;
    const unsigned voff_tc = (CT && bt0 + (unsigned)(lane >> 2) <= bt_last) ? ((bt0 + (unsigned)(lane >> 2)) * p.ldt + 4u * (unsigned)((lane & 3) ^ swz4(lane >> 2))) * 4u : OOB;
    int tfrag[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        const int m = min(m0 + (wave * MT + jt) * 16 + r16, M - 1);
        const int tix = m / H - (int)bt0;
        tfrag[jt] = tix * 16 + 4 * (q ^ swz4(tix));
    }
    // --- epilogue rows of this lane
    unsigned voff_u[MT], voff_v[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        const int m = m0 + (wave * MT + jt) * 16 + r16;
        const unsigned mm = m < M ? (unsigned)m : 0u;
        const unsigned bt = mm / (unsigned)H;
        const unsigned b = bt / (unsigned)T;
        const unsigned hr = b * H + (mm - bt * H);
        voff_u[jt] = m < M ? (hr * p.ldu + 4 * q) * 4u : OOB;
        voff_v[jt] = m < M ? (bt * p.ldv + 4 * q) * 4u : OOB;
    }
    const int rslot = 4 * (q ^ swz4(r16));          // swizzled float offset of this lane's fragment slot

    float s_part[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) s_part[jt] = 0.f;

    for (int nc = 0; nc < p.nchunks; ++nc) {
        // accumulators start at u[b,h,k] + v[b,t,k]: the loads land straight in the accumulator
        // registers, so the epilogue needs no u/v traffic.
        const int kc0 = nc * WROWS;
        const int nt_live = min(NT, (D - kc0 + 15) >> 4);             // uniform: column tiles of this chunk that hold columns < D
        f32x4 acc[NT][MT];
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int kb = (kc0 + it * 16) * 4;                       // uniform byte offset of the tile
                f32x4 uu = f32x4{0.f, 0.f, 0.f, 0.f}, vv = uu;
                if (!(NRM_DIAG_FWD & 1)) {
                    uu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u[jt], kb, 0));
                    vv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, voff_v[jt], kb, 0));
                }
                const bool kok = kc0 + it * 16 + 4 * q < D;               // only matters when D % 16 != 0
                acc[it][jt] = kok ? uu + vv : f32x4{0.f, 0.f, 0.f, 0.f};
                // bound the burst (8 loads in flight): all 2*NT*MT at once would be the register peak of
                // the kernel and push values used in the K loop into scratch
                if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // K-chunk c of this N-chunk -> LDS buffer `buf` (asynchronous; completion is tracked by vmcnt).
        // Columns >= D of the last chunk need no mask: the packed W_p is zero there.
        auto dma_chunk = [&](int c, float* buf) {
            const int wbase = (c * p.rows + nc * WROWS) * 64;             // bytes, uniform
            for (int pc = wave; pc < NT; pc += NW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + pc * 256),
                                                         16, lane * 16, wbase + pc * 1024, 0, 0);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                float* tdst = buf + (WROWS + (wave * MT + j) * 16) * 16;
                if (CT) {                                                // [W | h rows | distinct t rows]
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void*)tdst, 16, voff_h[j], c * 64, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (__attribute__((address_space(3))) void*)tdst, 16, voff_t[j], c * 64, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (__attribute__((address_space(3))) void*)(tdst + BM * 16), 16, voff_h[j], c * 64, 0, 0);
                }
            }
            // the workgroup's distinct candidate rows: one piece, staged by the wave with the fewest W pieces
            if (CT && wave == (NT % NW == 0 ? NW - 1 : NT % NW))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (__attribute__((address_space(3))) void*)(buf + (WROWS + BM) * 16), 16, voff_tc, c * 64, 0, 0);
        };
        auto compute = [&](const float* buf) {
            const float* Tl = buf + WROWS * 16;
            f32x4 pf[MT];                                    // B operand rows: P[m,:] = t[b,t,:] * h[b,h,:]
#pragma unroll
            for (int jt = 0; jt < MT; ++jt) {
                const int ro = ((wave * MT + jt) * 16 + r16) * 16 + rslot;
                if (CT) pf[jt] = *reinterpret_cast<const f32x4*>(&Tl[ro]) * *reinterpret_cast<const f32x4*>(&Tl[BM * 16 + tfrag[jt]]);
                else pf[jt] = *reinterpret_cast<const f32x4*>(&Tl[ro]) * *reinterpret_cast<const f32x4*>(&Tl[BM * 16 + ro]);
            }
            // W fragments are read one tile ahead; the scheduling barrier keeps hipcc from hoisting
            // all NT reads in front of the MFMAs (that costs 4*NT registers).
            f32x4 af = *reinterpret_cast<const f32x4*>(&buf[r16 * 16 + rslot]);
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                f32x4 afn = af;
                if (it + 1 < NT) afn = *reinterpret_cast<const f32x4*>(&buf[((it + 1) * 16 + r16) * 16 + rslot]);
                if (WPE < 3 || it < nt_live)                          // (13x1 plan) all-padding column tiles of the last N-chunk: no MFMAs
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int jt = 0; jt < MT; ++jt) acc[it][jt] = mfma16(af[j], pf[jt][j], acc[it][jt]);
                __builtin_amdgcn_sched_barrier(0);
                af = afn;
            }
        };

        // pipeline: DMA of chunk c+1 flies under the MFMAs of chunk c; one barrier per chunk
        // (__syncthreads() waits for this wave's DMA -- vmcnt(0) -- before the barrier).
        dma_chunk(0, smem);
        __syncthreads();
        if (NRM_PRIO) __builtin_amdgcn_s_setprio(NRM_PRIO);           // MFMA phase above the co-resident waves' epilogues / accumulator loads
        for (int c = 0; c < p.kchunks; ++c) {
            float* cur = smem + (c & 1) * BUF;
            float* nxt = smem + ((c & 1) ^ 1) * BUF;
            if (c + 1 < p.kchunks && !((NRM_DIAG_FWD & 2) && c > 0)) dma_chunk(c + 1, nxt);
            compute(cur);
            __syncthreads();
        }
        if (NRM_PRIO) __builtin_amdgcn_s_setprio(0);

        // --- epilogue of this N-chunk: optional z store ; GELU ; partial fc2 dot
        // fc2 weights of a tile are requested two tiles ahead of their use, and the tiles are kept in order (scheduling barrier): left
        // alone hipcc requests all NT of them in front of the first GELU -- 4 NT registers on top of the live accumulators
        f32x4 wq[3];
#pragma unroll
        for (int it = 0; it < 2 && it < NT; ++it)
            wq[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, 16 * q, (kc0 + it * 16) * 4, 0));   // 0 beyond D
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int kb = (kc0 + it * 16) * 4;
            const bool kok = kc0 + it * 16 + 4 * q < D;
            if (it + 2 < NT) wq[(it + 2) % 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, 16 * q, kb + 128, 0));
            const f32x4 ww = wq[it % 3];
#pragma unroll
            for (int jt = 0; jt < MT; ++jt) {
                const f32x4 zz = acc[it][jt];
                if (SAVE_Z && kok && !(NRM_DIAG_FWD & 4)) {
                    const unsigned vz = (unsigned)(((wave * MT + jt) * 16 + r16) * D + 4 * q) * 4u;   // rows >= M: out of range
                    store_b128_guarded(__builtin_bit_cast(u32x4, zz), rs_z, vz, kb);
                }
                if (NRM_DIAG_FWD & 8) s_part[jt] += ww[0] * zz[0] + ww[1] * zz[1] + ww[2] * zz[2] + ww[3] * zz[3];
                else s_part[jt] += gelu_dot4(ww, zz);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float b2 = p.b2[0];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        const float v = sum_rows4(s_part[jt]);
        const int m = m0 + (wave * MT + jt) * 16 + r16;
        if (q == 0 && m < M) p.s[m] = v + b2;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch (called from capi.hip)

// Tile table: NT 16-column tiles per N-chunk x MT 16-row tiles per wave.  4*NT*MT accumulator registers must
// stay within 2 waves/SIMD (<= 256 VGPRs), and 2 workgroups x 2 LDS buffers within 160 KB.
// Up to 25 column tiles (D <= 400) the whole width is ONE N-chunk: no column padding and a single
// accumulator-init / epilogue per row tile (measured at D=400: 13x3 in two chunks 4.73 ms, 25x1 4.28 ms) -- except
// for 21..26 tiles, see pwattn_fwd_plan.
static const int kNT[10] = {4, 6, 8, 10, 12, 13, 14, 16, 20, 25};
static const int kMT[10] = {4, 4, 3, 1, 1, 1, 2, 1, 1, 1};     // 10..13 tiles: one row tile per wave, 3 waves/SIMD (3 row tiles spilled)

FwdPlan pwattn_fwd_plan(int D) {
    const int n16 = (D + 15) / 16;
    int nch = n16 <= 25 ? 1 : (n16 + 15) / 16;          // wider than 25 tiles: chunks of at most 16 tiles
    // 21..26 column tiles (320 < D <= 416, incl. D = 400): two N-chunks of 13 tiles with ONE row tile per wave.  That
    // kernel needs 129 VGPRs and 43 KB of LDS, so three workgroups fit a CU (3 waves/SIMD instead of 2) -- measured at
    // D = 400: 4.14 ms against 4.29 ms for the single 25-tile chunk, although the rows are streamed twice and the last
    // chunk carries one all-padding tile (its MFMAs are skipped); three chunks of 10 tiles at four workgroups per CU:
    // 4.27 ms.  NRM_FWD_13X1=0 restores the one-chunk plan.
    static const int use13 = [] { const char* e = getenv("NRM_FWD_13X1"); return e && e[0] == '0' ? 0 : 1; }();
    if (use13 && n16 >= 21 && n16 <= 26) nch = 2;
    const int need = (n16 + nch - 1) / nch;
    int sel = 9;
    for (int i = 0; i < 10; ++i) if (kNT[i] >= need) { sel = i; break; }
    FwdPlan pl;
    pl.NT = kNT[sel]; pl.MT = kMT[sel]; pl.nchunks = nch;
    pl.rows = nch * pl.NT * 16; pl.kchunks = n16;
    return pl;
}

template <int NT, int MT, int WPE = 2, bool CT = false, int NW = 4>
static hipError_t launch_fwd_t(const FwdParams& p, hipStream_t st) {
    constexpr int BM = NW * MT * 16;
    const long nblk = (p.M + BM - 1) / BM;
    if (nblk <= 0) return hipSuccess;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    if (p.z) hipLaunchKernelGGL((pwattn_fwd_kernel<NT, MT, true, WPE, CT, NW>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p);
    else     hipLaunchKernelGGL((pwattn_fwd_kernel<NT, MT, false, WPE, CT, NW>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, p);
    return hipGetLastError();
}

// compact candidate image (template parameter CT): histories of at least 16 rows (a workgroup's 64 rows then span <= 5 candidates and a
// 16-lane read group <= 2); NRM_FWD_CT=0 keeps the per-row image (round 3), =1 forces the compact one for any H >= 5
static bool fwd_compact_t(int H, int block_rows = 64) {
    const char* e = getenv("NRM_FWD_CT");                         // (read per launch: tests switch forms inside one process)
    if (e && e[0] == '0') return false;
    if ((block_rows - 1) / H + 2 > 16) return false;             // the image holds 16 candidate rows
    if (e && e[0] == '1') return true;
    return H >= 16;
}

hipError_t pwattn_fwd_launch(const FwdParams& p, const FwdPlan& pl, int mma, hipStream_t st) {
    if (pwattn_fwd_uses_rw(p.D, mma)) return pwattn_fwd_rw_launch(p, mma, st);
    switch (pl.NT) {
        case 4:  return launch_fwd_t<4, 4>(p, st);
        case 6:  return launch_fwd_t<6, 4>(p, st);
        case 8:  return launch_fwd_t<8, 3>(p, st);
        case 10: return fwd_compact_t(p.H) ? launch_fwd_t<10, 1, 4, true>(p, st) : launch_fwd_t<10, 1, 3>(p, st);
        case 12: return fwd_compact_t(p.H) ? launch_fwd_t<12, 1, 4, true>(p, st) : launch_fwd_t<12, 1, 3>(p, st);
        case 13:
            if (!fwd_compact_t(p.H, NRM_FWD_NW8 ? 128 : 64)) return launch_fwd_t<13, 1, 3>(p, st);
#if NRM_FWD_NW8
            return launch_fwd_t<13, 1, 4, true, 8>(p, st);            // tuning: 128-row workgroups of 8 waves, two per CU
#else
            return launch_fwd_t<13, 1, 4, true>(p, st);
#endif
        case 14: return launch_fwd_t<14, 2>(p, st);
        case 16: return launch_fwd_t<16, 1>(p, st);        // (compact image + three workgroups per CU measured SLOWER at C5: 19.43 vs 18.76 ms)
        case 20: return launch_fwd_t<20, 1>(p, st);
        case 25: return launch_fwd_t<25, 1>(p, st);
    }
    return hipErrorInvalidValue;
}

// bf16 forms: always the resident-W forward (pwattn_fwd_rw.hip).  fp32: only where the WHOLE W_p fits one LDS slice
// (D <= 128; reference default D = 64: 0.128 -> 0.104 ms); with several slices the chunk-streaming kernel below is faster
// (measured at C3, D = 400, 5 slices: 4.32 vs 4.13 ms; C5, D = 768: 20.5 vs 19.4 ms) and keeps the scores free of float
// atomics.  NRM_FWD_RW=0 / =2 force the streaming / the resident form for fp32.
bool pwattn_fwd_uses_rw(int D, int mma) {
    static const int mode = [] { const char* e = getenv("NRM_FWD_RW"); return e ? atoi(e) : 1; }();
    const RwPlan pl = pwattn_rw_plan(D, mma);
    if (pl.nts == 0) return false;
    if (mma != 0) return true;
    return mode == 2 || (mode == 1 && pl.nsplit == 1);
}

int pwattn_fwd_diag_flags() { return (NRM_DIAG_FWD ? 1 : 0) | (XCD_REMAP != 1 ? 2 : 0); }

hipError_t pack_wp_launch(const float* w, int ldw, int D, const FwdPlan& pl, int mma, float* packed, hipStream_t st) {
    if (mma) return pack_wp_bf16_launch(w, ldw, D, mma, packed, st);
    const int rows = pwattn_fwd_uses_rw(D, 0) ? pwattn_rw_plan(D, 0).rows : pl.rows;       // same image, the kernel's row padding
    const long total = (long)pl.kchunks * rows * 16;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wp_kernel, dim3(blocks), dim3(256), 0, st, w, ldw, D, rows, pl.kchunks, packed);
    return hipGetLastError();
}

}  // namespace nrm
