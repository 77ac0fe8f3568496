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
#include "common.hpp"
#include "pwattn.hpp"

namespace nrm {

// ---------------------------------------------------------------------------------------------
// W_p prepack: packed[c][row][16] = W_p[row][16c .. 16c+15], zero padded to `rows` rows and to a
// multiple of 16 columns, so that one K-chunk of one N-chunk is a single contiguous block.
// W_p = fc1.weight[:, 3D:4D] (row stride ldw = 4D).
__global__ void pack_wp_kernel(const float* __restrict__ w, int ldw, int D, int rows, int kchunks,
                               float* __restrict__ packed) {
    const long total = (long)kchunks * rows * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        const long rc = i >> 4;
        const int row = (int)(rc % rows);
        const int c = (int)(rc / rows);
        const int d = c * 16 + j;
        packed[i] = (row < D && d < D) ? w[(long)row * ldw + d] : 0.0f;
    }
}

template <int NT, int MT, bool SAVE_Z>
__global__ __launch_bounds__(256, 2) void pwattn_fwd_kernel(const FwdParams p) {
    constexpr int BM = 4 * MT * 16;       // data rows per workgroup
    constexpr int LDW = 20;               // LDS row stride in floats (16 + 4 pad, keeps 16-B alignment)
    constexpr int WROWS = NT * 16;
    constexpr int WF4 = WROWS * 4;        // float4 items of one W chunk
    constexpr int WPT = (WF4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float smem[(WROWS + BM) * LDW];
    float* Ws = smem;
    float* Ps = smem + WROWS * LDW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int M = (int)p.M;                       // host guarantees M < 2^31
    const int m0 = blockIdx.x * BM;
    const int T = p.T, H = p.H, D = p.D;

    // --- P staging assignment: MT float4 items per thread, item = tid + 256*j -> (row, c4 = tid&3).
    // Out-of-range rows read row 0 (always valid) and are zeroed when the product is written.
    const int c4 = tid & 3;
    int t_off[MT], h_off[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + ((tid + 256 * j) >> 2);
        const unsigned mm = m < M ? (unsigned)m : 0u;
        const unsigned bt = mm / (unsigned)H;
        const unsigned b = bt / (unsigned)T;
        const unsigned hr = b * H + (mm - bt * H);
        t_off[j] = (int)(bt * p.ldt);
        h_off[j] = (int)(hr * p.ldh);
    }

    float s_part[MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) s_part[jt] = 0.f;

    for (int nc = 0; nc < p.nchunks; ++nc) {
        // accumulators start at u[b,h,k] + v[b,t,k]: the loads land straight in the accumulator
        // registers, so the epilogue needs no u/v traffic and no extra live registers.
        const int kc0 = nc * WROWS;
        f32x4 acc[NT][MT];
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            const int m = m0 + (wave * MT + jt) * 16 + r16;
            const unsigned mm = m < M ? (unsigned)m : 0u;
            const unsigned bt = mm / (unsigned)H;
            const unsigned b = bt / (unsigned)T;
            const unsigned hr = b * H + (mm - bt * H);
            const float* up = p.u + (size_t)hr * p.ldu;
            const float* vp = p.v + (size_t)bt * p.ldv;
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int k = kc0 + it * 16 + 4 * q;
                const int kk = k < D ? k : 0;
                const f32x4 uv = *reinterpret_cast<const f32x4*>(up + kk) + *reinterpret_cast<const f32x4*>(vp + kk);
                acc[it][jt] = k < D ? uv : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }

        f32x4 wreg[WPT], treg[MT], hreg[MT];
        bool dok = true;
        auto load_chunk = [&](int c) {
            // the packed buffer is over-allocated by 256 float4, so the tail items may over-read
            const f32x4* src = reinterpret_cast<const f32x4*>(p.wp + ((long)c * p.rows + (long)nc * WROWS) * 16);
#pragma unroll
            for (int j = 0; j < WPT; ++j) wreg[j] = src[tid + 256 * j];
            int dcol = c * 16 + 4 * c4;
            dok = dcol < D;                 // D % 4 == 0 -> a float4 is entirely valid or entirely out
            dcol = dok ? dcol : 0;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                treg[j] = *reinterpret_cast<const f32x4*>(p.t + t_off[j] + dcol);
                hreg[j] = *reinterpret_cast<const f32x4*>(p.h + h_off[j] + dcol);
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int idx = tid + 256 * j;
                if (idx < WF4) *reinterpret_cast<f32x4*>(&Ws[(idx >> 2) * LDW + 4 * (idx & 3)]) = wreg[j];
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = (tid + 256 * j) >> 2;
                const bool ok = dok && (m0 + row < M);
                const f32x4 pr = treg[j] * hreg[j];
                *reinterpret_cast<f32x4*>(&Ps[row * LDW + 4 * c4]) = ok ? pr : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };

        load_chunk(0);
        for (int c = 0; c < p.kchunks; ++c) {
            __syncthreads();                 // every wave is done reading the previous chunk
            store_chunk();
            __syncthreads();
            if (c + 1 < p.kchunks) load_chunk(c + 1);   // global loads fly under the MFMAs below

            f32x4 pf[MT];
#pragma unroll
            for (int jt = 0; jt < MT; ++jt)
                pf[jt] = *reinterpret_cast<const f32x4*>(&Ps[((wave * MT + jt) * 16 + r16) * LDW + 4 * q]);
            // W fragments are read one tile ahead; the scheduling barrier keeps hipcc from hoisting
            // all NT reads in front of the MFMAs (that costs 4*NT registers and spills).
            f32x4 af = *reinterpret_cast<const f32x4*>(&Ws[r16 * LDW + 4 * q]);
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                f32x4 afn = af;
                if (it + 1 < NT) afn = *reinterpret_cast<const f32x4*>(&Ws[((it + 1) * 16 + r16) * LDW + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int jt = 0; jt < MT; ++jt) acc[it][jt] = mfma16(af[j], pf[jt][j], acc[it][jt]);
                __builtin_amdgcn_sched_barrier(0);
                af = afn;
            }
        }

        // --- epilogue of this N-chunk: optional z store ; GELU ; partial fc2 dot
        float* zp[MT];
        bool mok[MT];
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            const int m = m0 + (wave * MT + jt) * 16 + r16;
            mok[jt] = m < M;
            zp[jt] = SAVE_Z ? p.z + (size_t)(mok[jt] ? m : 0) * D : nullptr;
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int k = kc0 + it * 16 + 4 * q;
            if (k < D) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(p.w2 + k);
#pragma unroll
                for (int jt = 0; jt < MT; ++jt) {
                    const f32x4 zz = acc[it][jt];
                    if (SAVE_Z && mok[jt]) *reinterpret_cast<f32x4*>(zp[jt] + k) = zz;
                    s_part[jt] += ww[0] * gelu_f(zz[0]) + ww[1] * gelu_f(zz[1]) + ww[2] * gelu_f(zz[2]) + ww[3] * gelu_f(zz[3]);
                }
            }
        }
    }

    const float b2 = p.b2[0];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        float v = s_part[jt];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int m = m0 + (wave * MT + jt) * 16 + r16;
        if (q == 0 && m < M) p.s[m] = v + b2;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch (called from capi.hip)

// Tile table: NT 16-column tiles per N-chunk x MT 16-row tiles per wave.  4*NT*MT accumulator
// registers must leave room for the staging prefetch at 2 waves/SIMD (<= 256 VGPRs, no scratch).
static const int kNT[8] = {4, 6, 8, 10, 12, 13, 14, 16};
static const int kMT[8] = {6, 5, 4, 3, 3, 3, 2, 2};

FwdPlan pwattn_fwd_plan(int D) {
    const int n16 = (D + 15) / 16;
    const int nch = (n16 + 15) / 16;               // at most 16 tiles of 16 columns per N-chunk
    const int need = (n16 + nch - 1) / nch;
    int sel = 7;
    for (int i = 0; i < 8; ++i) if (kNT[i] >= need) { sel = i; break; }
    FwdPlan pl;
    pl.NT = kNT[sel]; pl.MT = kMT[sel]; pl.nchunks = nch;
    pl.rows = nch * pl.NT * 16; pl.kchunks = n16;
    return pl;
}

template <int NT, int MT>
static hipError_t launch_fwd_t(const FwdParams& p, hipStream_t st) {
    constexpr int BM = 4 * MT * 16;
    const long nblk = (p.M + BM - 1) / BM;
    if (nblk <= 0) return hipSuccess;
    if (nblk > 0x7fffffffL) return hipErrorInvalidValue;
    if (p.z) hipLaunchKernelGGL((pwattn_fwd_kernel<NT, MT, true>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    else     hipLaunchKernelGGL((pwattn_fwd_kernel<NT, MT, false>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t pwattn_fwd_launch(const FwdParams& p, const FwdPlan& pl, hipStream_t st) {
    switch (pl.NT) {
        case 4:  return launch_fwd_t<4, 6>(p, st);
        case 6:  return launch_fwd_t<6, 5>(p, st);
        case 8:  return launch_fwd_t<8, 4>(p, st);
        case 10: return launch_fwd_t<10, 3>(p, st);
        case 12: return launch_fwd_t<12, 3>(p, st);
        case 13: return launch_fwd_t<13, 3>(p, st);
        case 14: return launch_fwd_t<14, 2>(p, st);
        case 16: return launch_fwd_t<16, 2>(p, st);
    }
    return hipErrorInvalidValue;
}

hipError_t pack_wp_launch(const float* w, int ldw, int D, const FwdPlan& pl, float* packed, hipStream_t st) {
    const long total = (long)pl.kchunks * pl.rows * 16;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_wp_kernel, dim3(blocks), dim3(256), 0, st, w, ldw, D, pl.rows, pl.kchunks, packed);
    return hipGetLastError();
}

}  // namespace nrm
