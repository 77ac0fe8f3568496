// extern "C" entry points declared in include/nrm_hotpath.h: host-side validation + launches only.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../include/nrm_hotpath.h"
#include "pwattn.hpp"
#include "gemm.hpp"
#include "head.hpp"
#include "pool_loss.hpp"
#include "frontend.hpp"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return NRM_OK;
    return fail(NRM_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// wave slots of the chip at 2 waves/SIMD: 256 CUs x 4 SIMDs x 2
static const int kTargetWaves = 2048;
// Wave tasks of the two backward contraction launches.  Measured on MI355X at C3 (scripts/sweep_waves.sh): several
// rounds of smaller tasks balance better than exactly one round (dh pass 4.81 -> 4.58 ms at 4 rounds of 3 waves/SIMD,
// dt/dW pass 5.81 -> 5.56 ms at 3 rounds; every split of the dW pass costs one [D,D] partial slab).
// NRM_BT_WAVES / NRM_BH_WAVES override them for tuning.
static const int kBtWaves = 6144;
static const int kBtMinGroups = 96;     // groups per split of the dt/dW pass (each split = one [D, D] slab of dW_p)
static const int kBhWaves = 12288;
static const int kTnWaves = 0;         // gemm_tn (dW = dY^T X): 0 = one round of wave tasks at the chosen kernel's occupancy (gemm_tn_plan); NRM_TN_WAVES overrides

static int check_dims(const char* fn, int B, int T, int H, int D) {
    if (B < 0 || T <= 0 || H <= 0 || D <= 0) return fail(NRM_EINVAL, "%s: B=%d T=%d H=%d D=%d must be positive", fn, B, T, H, D);
    if (D % 4) return fail(NRM_EINVAL, "%s: D=%d must be a multiple of 4", fn, D);
    if (D > 1024) return fail(NRM_EINVAL, "%s: D=%d > 1024 not supported", fn, D);
    const long M = (long)B * T * H;
    // buffer descriptors address with 32-bit byte offsets; 2^31 is reserved as the "masked row" marker
    if (M >= (1L << 31) || (long)B * T * D >= (1L << 29) || (long)B * H * D >= (1L << 29))
        return fail(NRM_EINVAL, "%s: B*T*H=%ld exceeds 2^31 rows or a [B,T,D]/[B,H,D] operand exceeds 2^31 bytes", fn, M);
    return NRM_OK;
}

extern "C" {

int nrm_abi_version(void) { return NRM_ABI_VERSION; }
int nrm_build_flags(void) { return nrm::pwattn_fwd_diag_flags() | nrm::pwattn_fwd_rw_diag_flags() | nrm::pwattn_bwd_diag_flags() |
                                    nrm::pwattn_bwd_rw_diag_flags() | nrm::pwattn_bwd_dp_diag_flags() | nrm::gemm_bf16_diag_flags() | nrm::gemm_diag_flags()
#if defined(NRM_STORE_GUARD) && NRM_STORE_GUARD != 2
                                    | 0x200                            // 16-byte stores without (or with a weaker) hazard guard: common.hpp
#endif
                                    ; }
const char* nrm_last_error(void) { return g_err; }

long nrm_pwattn_packed_floats(int D) {
    if (D <= 0) return 0;
    const nrm::FwdPlan pl = nrm::pwattn_fwd_plan(D);
    // fp32: kchunks 16-column chunks of rows x 64 B.  bf16 / bf16x3 (pwattn_fwd_bf16.hip): k32 chunks x (1 | 2) images x plan
    // rows x 64 B, where the resident-W plan may pad the rows up to one more slice -- size for the largest of the three
    const nrm::RwPlan r0 = nrm::pwattn_rw_plan(D, NRM_MMA_F32);
    const long f32 = (long)pl.kchunks * (pl.rows > r0.rows ? pl.rows : r0.rows) * 16;
    const nrm::RwPlan r3 = nrm::pwattn_rw_plan(D, NRM_MMA_BF16X3), r1 = nrm::pwattn_rw_plan(D, NRM_MMA_BF16);
    const long b3 = (long)r3.k32 * 2 * r3.rows * 16, b1 = (long)r1.k32 * r1.rows * 16;
    const long need = f32 > b3 ? (f32 > b1 ? f32 : b1) : (b3 > b1 ? b3 : b1);
    return need + 256 * 4 * 2;   // + over-read pad of the staging loop
}

static int check_mma(const char* fn, int mma) {
    if (mma != NRM_MMA_F32 && mma != NRM_MMA_BF16 && mma != NRM_MMA_BF16X3)
        return fail(NRM_EINVAL, "%s: mma=%d (NRM_MMA_F32, NRM_MMA_BF16 or NRM_MMA_BF16X3)", fn, mma);
    return NRM_OK;
}

int nrm_pwattn_pack_wp(const float* fc1_weight, int ld, int D, int mma, float* packed, nrm_stream_t stream) {
    if (!fc1_weight || !packed) return fail(NRM_EINVAL, "nrm_pwattn_pack_wp: null pointer");
    if (D <= 0 || D % 4 || ld < 4 * D) return fail(NRM_EINVAL, "nrm_pwattn_pack_wp: D=%d ld=%d (need D%%4==0, ld>=4D)", D, ld);
    if (int rc = check_mma("nrm_pwattn_pack_wp", mma)) return rc;
    const nrm::FwdPlan pl = nrm::pwattn_fwd_plan(D);
    return check_hip(nrm::pack_wp_launch(fc1_weight + 3 * (long)D, ld, D, pl, mma, packed, (hipStream_t)stream), "pack_wp");
}

int nrm_pwattn_fwd(const float* t, const float* h, const float* u, const float* v, const float* packed_wp,
                   const float* w2, const float* b2, float* z, float* s,
                   int B, int T, int H, int D, int mma, nrm_stream_t stream) {
    if (int rc = check_dims("nrm_pwattn_fwd", B, T, H, D)) return rc;
    if (int rc = check_mma("nrm_pwattn_fwd", mma)) return rc;
    if (!t || !h || !u || !v || !packed_wp || !w2 || !b2 || !s) return fail(NRM_EINVAL, "nrm_pwattn_fwd: null pointer");
    if (B == 0) return NRM_OK;
    const nrm::FwdPlan pl = nrm::pwattn_fwd_plan(D);
    nrm::FwdParams p;
    p.t = t; p.h = h; p.u = u; p.v = v; p.wp = packed_wp; p.w2 = w2; p.b2 = b2; p.z = z; p.s = s;
    p.M = (long)B * T * H; p.T = T; p.H = H; p.D = D;
    p.ldt = D; p.ldh = D; p.ldu = D; p.ldv = D;
    p.wp_bytes = (unsigned)(nrm_pwattn_packed_floats(D) * 4);
    p.t_bytes = (unsigned)((long)B * T * D * 4);
    p.h_bytes = (unsigned)((long)B * H * D * 4);
    if (mma != NRM_MMA_F32 && !nrm::pwattn_fwd_uses_rw(D, mma))
        return fail(NRM_EINVAL, "nrm_pwattn_fwd: D=%d is too wide for the bf16 forward (one 16-column slice of W_p must fit the LDS)", D);
    p.rows = pl.rows; p.kchunks = pl.kchunks; p.nchunks = pl.nchunks;
    return check_hip(nrm::pwattn_fwd_launch(p, pl, mma, (hipStream_t)stream), "pwattn_fwd");
}

int nrm_pwattn_bwd_dz(float* z_inout, const float* ds, const float* w2, float* dw2, float* db2, float* du, float* dv,
                      int B, int T, int H, int D, int dz_format, nrm_stream_t stream) {
    if (int rc = check_dims("nrm_pwattn_bwd_dz", B, T, H, D)) return rc;
    if (!z_inout || !ds || !w2 || !dw2 || !du || !dv) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dz: null pointer");
    if (B > 65535) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dz: B=%d > 65535 (one grid row per impression)", B);
    if (dz_format != NRM_DZ_F32 && dz_format != NRM_DZ_HL4) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dz: dz_format=%d (NRM_DZ_F32 or NRM_DZ_HL4)", dz_format);
    return check_hip(nrm::bwd_dz_launch(z_inout, ds, w2, dw2, db2, du, dv, B, T, H, D, dz_format, (hipStream_t)stream), "bwd_dz");
}

int nrm_pwattn_bwd_rw_supported(int D, int mma) { return nrm::pwattn_bwd_rw_plan(D, mma).ng ? 1 : 0; }
long nrm_pwattn_bwd_rw_packed_floats(int D, int mma) { return nrm::pwattn_bwd_rw_packed_floats(D, mma); }

int nrm_pwattn_bwd_rw_pack(const float* fc1_weight, int ld, int D, int mma, float* packed, nrm_stream_t stream) {
    if (!fc1_weight || !packed) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_pack: null pointer");
    if (D <= 0 || D % 4 || ld < 4 * D) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_pack: D=%d ld=%d (need D%%4==0, ld>=4D)", D, ld);
    if (!nrm_pwattn_bwd_rw_supported(D, mma)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_pack: D=%d mma=%d has no resident-W backward", D, mma);
    return check_hip(nrm::pwattn_bwd_rw_pack_launch(fc1_weight + 3 * (long)D, ld, D, mma, packed, (hipStream_t)stream), "bwd_rw_pack");
}

int nrm_pwattn_bwd_rw_dtdh(const float* dz_hl4, const float* t, const float* h, const float* packed, float* dt, float* dh,
                           int B, int T, int H, int D, int mma, nrm_stream_t stream) {
    if (int rc = check_dims("nrm_pwattn_bwd_rw_dtdh", B, T, H, D)) return rc;
    if (!dz_hl4 || !t || !h || !packed || !dt || !dh) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_dtdh: null pointer");
    if (!nrm_pwattn_bwd_rw_supported(D, mma)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_dtdh: D=%d mma=%d has no resident-W backward", D, mma);
    if ((long)T * H * D * 4 >= (1L << 31)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_rw_dtdh: one impression's dz block exceeds 2^31 bytes");
    if (B == 0) return NRM_OK;
    nrm::BwdRwParams p = {};
    p.dz = dz_hl4; p.t = t; p.h = h; p.wimg = packed; p.dt = dt; p.dh = dh; p.B = B; p.T = T; p.H = H; p.D = D;
    p.w_bytes = (unsigned)(nrm_pwattn_bwd_rw_packed_floats(D, mma) * 4);
    return check_hip(nrm::pwattn_bwd_rw_launch(p, mma, (hipStream_t)stream), "bwd_rw_dtdh");
}

int nrm_pwattn_bwd_dp_supported(int D, int H) { return nrm::pwattn_bwd_dp_plan(D, H).NT ? 1 : 0; }
long nrm_pwattn_bwd_dp_packed_floats(int D, int H) { return nrm::pwattn_bwd_dp_packed_floats(D, H); }

int nrm_pwattn_bwd_dp_pack(const float* fc1_weight, int ld, int D, int H, float* packed, nrm_stream_t stream) {
    if (!fc1_weight || !packed) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_pack: null pointer");
    if (D <= 0 || D % 4 || ld < 4 * D) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_pack: D=%d ld=%d (need D%%4==0, ld>=4D)", D, ld);
    if (!nrm_pwattn_bwd_dp_supported(D, H)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_pack: D=%d H=%d has no dP-walk backward", D, H);
    return check_hip(nrm::pwattn_bwd_dp_pack_launch(fc1_weight + 3 * (long)D, ld, D, H, packed, (hipStream_t)stream), "bwd_dp_pack");
}

int nrm_pwattn_bwd_dp_dtdh(const float* dz, const float* t, const float* h, const float* packed, float* dt, float* dh,
                           int B, int T, int H, int D, nrm_stream_t stream) {
    if (int rc = check_dims("nrm_pwattn_bwd_dp_dtdh", B, T, H, D)) return rc;
    if (!dz || !t || !h || !packed || !dt || !dh) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_dtdh: null pointer");
    if (!nrm_pwattn_bwd_dp_supported(D, H)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_dtdh: D=%d H=%d has no dP-walk backward (D %% 4 == 0, H >= 16)", D, H);
    // a row block of 64 history rows spans at most 63 / H + 2 impressions, whose dz blocks one buffer descriptor must cover
    if ((long)(63 / H + 2) * T * H * D * 4 >= (1L << 31)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_dtdh: the dz blocks of one row block exceed 2^31 bytes");
    if ((long)B * H * D * 4 >= (1L << 31) || (long)B * T * D * 4 >= (1L << 31)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_dp_dtdh: t or h exceeds 2^31 bytes");
    if (B == 0) return NRM_OK;
    nrm::BwdDpParams p = {};
    p.dz = dz; p.t = t; p.h = h; p.wimg = packed; p.dt = dt; p.dh = dh; p.B = B; p.T = T; p.H = H; p.D = D;
    p.w_bytes = (unsigned)(nrm_pwattn_bwd_dp_packed_floats(D, H) * 4);
    p.t_bytes = (unsigned)((long)B * T * D * 4); p.h_bytes = (unsigned)((long)B * H * D * 4);
    return check_hip(nrm::pwattn_bwd_dp_launch(p, (hipStream_t)stream), "bwd_dp_dtdh");
}

int nrm_pwattn_bwd_nsplit(int B, int T, int H, int D, int mma) {
    if (B <= 0 || T <= 0 || H <= 0 || D <= 0) return 0;
    int tw1 = kBtWaves;
    if (const char* e = getenv("NRM_BT_WAVES")) tw1 = atoi(e);
    return nrm::bwd_e_plan(D, B * T, tw1, kBtMinGroups, mma).nsplit;
}

int nrm_pwattn_bwd_contract(const float* dz, const float* t, const float* h, const float* wp, int ldwp,
                            float* dt, float* dh, float* ws, int B, int T, int H, int D, int passes, int mma,
                            int dz_format, nrm_stream_t stream) {
    if (int rc = check_dims("nrm_pwattn_bwd_contract", B, T, H, D)) return rc;
    if (int rc = check_mma("nrm_pwattn_bwd_contract", mma)) return rc;

    if (passes != 1 && passes != 2 && passes != 3 && passes != 4) return fail(NRM_EINVAL, "nrm_pwattn_bwd_contract: passes=%d", passes);
    if ((dz_format != NRM_DZ_F32 && dz_format != NRM_DZ_HL4) || (dz_format == NRM_DZ_HL4 && (passes != 4 || mma == NRM_MMA_F32)))
        return fail(NRM_EINVAL, "nrm_pwattn_bwd_contract: passes=%d mma=%d with dz_format=%d (only the dW_p-only pass, 4, of the bf16 arithmetics reads NRM_DZ_HL4)", passes, mma, dz_format);
    if (!dz || !t || !h || !wp) return fail(NRM_EINVAL, "nrm_pwattn_bwd_contract: null pointer");
    if (((passes & 1) && (!dt || !ws)) || ((passes & 2) && !dh) || (passes == 4 && !ws)) return fail(NRM_EINVAL, "nrm_pwattn_bwd_contract: null output");
    if (ldwp < D || ldwp % 4) return fail(NRM_EINVAL, "nrm_pwattn_bwd_contract: ldwp=%d", ldwp);
    if (B == 0) return NRM_OK;
    const long HD = (long)H * D, TD = (long)T * D;
    // pass 1: groups (b,t); rows r = h.  X_g = dz[b,t,:,:], Y_g = h[b];  out = dt, scale rows = t
    if ((passes & 1) || passes == 4) {
        nrm::BwdEParams p = {};
        p.with_dt = passes == 4 ? 0 : 1; p.x_hl4 = dz_format == NRM_DZ_HL4 ? 1 : 0;
        p.X = dz; p.xs1 = (long)T * HD; p.xs2 = HD; p.xrs = D;
        p.Y = h; p.ys1 = HD; p.yrs = D;
        p.wp = wp; p.ldwp = ldwp; p.srow = t; p.lds_ = D; p.out = dt; p.ldo = D; p.ws = ws;
        p.G = B * T; p.G2 = T; p.R = H; p.D = D;
        int tw1 = kBtWaves;
        if (const char* e = getenv("NRM_BT_WAVES")) tw1 = atoi(e);
        const nrm::BwdEPlan pl = nrm::bwd_e_plan(D, p.G, tw1, kBtMinGroups, mma);
        // big groups (C5: 128 rows x 768 columns = 393 KB of dz each): the waves of a workgroup walk neighbouring groups, which
        // cuts what an XCD's L2 must hold -- FETCH_SIZE of this pass at C5 26.3 -> 8.0 GB, same duration (the MALL had been
        // absorbing the re-reads); small groups keep the blocked walk.  NRM_BT_INTERLEAVE=0|1 forces either (tests).
        p.interleave = (long)p.R * D * (long)sizeof(float) > (256L << 10);
        if (const char* e = getenv("NRM_BT_INTERLEAVE")) p.interleave = atoi(e);
        p.order = 0;
        if (const char* e = getenv("NRM_BT_ORDER")) p.order = atoi(e);
        if (int rc = check_hip(nrm::bwd_e_launch(p, pl, true, mma, (hipStream_t)stream), "bwd_e pass 1")) return rc;
    }
    // pass 2: groups (b,h); rows r = t.  X_g = dz[b,:,h,:], Y_g = t[b];  out = dh
    if (passes & 2) {
        nrm::BwdEParams p = {};
        p.X = dz; p.xs1 = (long)T * HD; p.xs2 = D; p.xrs = HD;
        p.Y = t; p.ys1 = TD; p.yrs = D;
        p.wp = wp; p.ldwp = ldwp; p.srow = nullptr; p.lds_ = 0; p.out = dh; p.ldo = D; p.ws = nullptr;
        p.G = B * H; p.G2 = H; p.R = T; p.D = D; p.with_dt = 1;
        // this variant keeps no dW_p accumulators (<= 168 VGPRs): three waves per SIMD
        int tw = kBhWaves;
        if (const char* e = getenv("NRM_BH_WAVES")) tw = atoi(e);
        const nrm::BwdEPlan pl = nrm::bwd_e_plan(D, p.G, tw, 1, mma);
        p.interleave = (long)p.R * D * (long)sizeof(float) > (256L << 10);       // read by the serial kernel only (bf16 forms, short T)
        if (const char* e = getenv("NRM_BT_INTERLEAVE")) p.interleave = atoi(e);
        if (int rc = check_hip(nrm::bwd_e_launch(p, pl, false, mma, (hipStream_t)stream), "bwd_e pass 2")) return rc;
    }
    return NRM_OK;
}


// ------------------------------------------------------------------------------------------- dense layers
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

long nrm_gemm_packed_floats(int nrows, int ncols, int mma) {
    if (nrows <= 0 || ncols <= 0) return 0;
    if (mma == NRM_MMA_BF16 || mma == NRM_MMA_BF16X3)                    // weight fragments of gemm_nt_rx: 1 KiB per (tile, chunk, image)
        return (long)((nrows + 15) / 16) * ((ncols + 31) / 32) * (mma == NRM_MMA_BF16X3 ? 2 : 1) * 256 + 1024;
    const nrm::GemmNtPlan pl = nrm::gemm_nt_plan(nrows);
    return (long)((ncols + 15) / 16) * pl.rows * 16 + 1024;
}

int nrm_gemm_nt_bf16_supported(int M, int K, int mma) { return nrm::gemm_nt_rx_bm(M, K, mma) ? 1 : 0; }

int nrm_gemm_pack(const float* src, long row_stride, long col_stride, int nrows, int ncols, float* packed,
                  nrm_stream_t stream) {
    if (!src || !packed || nrows <= 0 || ncols <= 0) return fail(NRM_EINVAL, "nrm_gemm_pack: bad argument");
    const nrm::GemmNtPlan pl = nrm::gemm_nt_plan(nrows);
    return check_hip(nrm::pack_rows_launch(src, row_stride, col_stride, nrows, ncols, pl.rows, (ncols + 15) / 16, packed,
                                           (hipStream_t)stream), "pack_rows");
}

int nrm_gemm_pack_multi(const nrm_pack_desc* descs, int n, nrm_stream_t stream) {
    if (n < 0 || (n > 0 && !descs)) return fail(NRM_EINVAL, "nrm_gemm_pack_multi: bad argument");
    for (int lo = 0; lo < n; lo += nrm::PACK_MAX) {
        const int m = n - lo < nrm::PACK_MAX ? n - lo : nrm::PACK_MAX;
        nrm::PackTable tab = {};
        long mx = 0;
        for (int i = 0; i < m; ++i) {
            const nrm_pack_desc& d = descs[lo + i];
            if (!d.src || !d.packed || d.nrows <= 0 || d.ncols <= 0) return fail(NRM_EINVAL, "nrm_gemm_pack_multi: entry %d is malformed", lo + i);
            const nrm::GemmNtPlan pl = nrm::gemm_nt_plan(d.nrows);
            nrm::PackEntry& e = tab.e[i];
            e.src = d.src; e.src2 = d.src2; e.dst = d.packed; e.rs = d.row_stride; e.cs = d.col_stride;
            e.nrows = d.nrows; e.ncols = d.ncols; e.rows = pl.rows; e.kchunks = (d.ncols + 15) / 16; e.sign2 = d.sign2; e.fmt = 0;
            if (d.mma == NRM_MMA_BF16 || d.mma == NRM_MMA_BF16X3) {
                e.fmt = d.mma == NRM_MMA_BF16X3 ? 2 : 1; e.rows = (d.nrows + 15) / 16 * 16; e.kchunks = (d.ncols + 31) / 32;
            } else if (d.mma != NRM_MMA_F32) return fail(NRM_EINVAL, "nrm_gemm_pack_multi: entry %d: mma=%d", lo + i, d.mma);
            const long total = e.fmt ? (long)(e.rows / 16) * e.kchunks * 512 : (long)e.kchunks * e.rows * 16;
            if (total > mx) mx = total;
        }
        if (int rc = check_hip(nrm::pack_rows_multi_launch(tab, m, mx, (hipStream_t)stream), "pack_rows_multi")) return rc;
    }
    return NRM_OK;
}

int nrm_gemm_nt(const float* x, int ldx, int M, const float* packed, int N, int K, const float* bias,
                float* y, int ldy, float* z, int ldz, const float* m, int ldm, int epilogue, int mma, nrm_stream_t stream) {
    if (!x || !packed || !y) return fail(NRM_EINVAL, "nrm_gemm_nt: null pointer");
    if (M < 0 || N <= 0 || K <= 0) return fail(NRM_EINVAL, "nrm_gemm_nt: M=%d N=%d K=%d", M, N, K);
    if (ldx % 4 || ldy % 4 || ldx < K || ldy < N || !al16(x) || !al16(y))
        return fail(NRM_EINVAL, "nrm_gemm_nt: ldx=%d ldy=%d must be multiples of 4 covering K=%d / N=%d, rows 16-B aligned", ldx, ldy, K, N);
    if (epilogue != NRM_EPI_BIAS && (!z || ldz % 4 || ldz < N || !al16(z)))
        return fail(NRM_EINVAL, "nrm_gemm_nt: epilogue %d needs z with ldz %% 4 == 0, ldz >= N", epilogue);
    if (epilogue < 0 || epilogue > 3) return fail(NRM_EINVAL, "nrm_gemm_nt: epilogue=%d", epilogue);
    if (epilogue == NRM_EPI_MUL && (!m || ldm % 4 || ldm < N || !al16(m)))
        return fail(NRM_EINVAL, "nrm_gemm_nt: NRM_EPI_MUL needs m with ldm %% 4 == 0, ldm >= N");
    if ((long)256 * (ldx > ldy ? ldx : ldy) * 4 >= (1L << 31)) return fail(NRM_EINVAL, "nrm_gemm_nt: leading dimension too large");
    if (int rc = check_mma("nrm_gemm_nt", mma)) return rc;
    if (mma != NRM_MMA_F32) {                                          // bf16 matrix cores: resident-X form (gemm_bf16.hip)
        if (!nrm::gemm_nt_rx_bm(M > 0 ? M : 1, K, mma)) return fail(NRM_EINVAL, "nrm_gemm_nt: K=%d is too wide for the bf16 form (nrm_gemm_nt_bf16_supported)", K);
        nrm::GemmRxParams r = {};
        r.x = x; r.ldx = ldx; r.xcols = ldx; r.wq = packed; r.wq_bytes = (unsigned)(nrm_gemm_packed_floats(N, K, mma) * 4);
        r.bias = bias; r.N = N; r.y = y; r.ldy = ldy;
        r.z = epilogue == NRM_EPI_BIAS ? nullptr : z; r.ldz = epilogue == NRM_EPI_BIAS ? 4 : ldz;
        r.m = epilogue == NRM_EPI_MUL ? m : nullptr; r.ldm = epilogue == NRM_EPI_MUL ? ldm : 4;
        r.M = M; r.K = K; r.k32 = (K + 31) / 32; r.nt16 = (N + 15) / 16;
        return check_hip(nrm::gemm_nt_rx_launch(r, epilogue, mma, (hipStream_t)stream), "gemm_nt_rx");
    }
    const nrm::GemmNtPlan pl = nrm::gemm_nt_plan(N);
    nrm::GemmNtParams p;
    p.x = x; p.ldx = ldx; p.xcols = ldx; p.wp = packed; p.wp_bytes = (unsigned)(nrm_gemm_packed_floats(N, K, NRM_MMA_F32) * 4);
    p.rows = pl.rows; p.bias = bias; p.N = N; p.y = y; p.ldy = ldy;
    p.z = epilogue == NRM_EPI_BIAS ? nullptr : z; p.ldz = epilogue == NRM_EPI_BIAS ? 4 : ldz;
    p.m = epilogue == NRM_EPI_MUL ? m : nullptr; p.ldm = epilogue == NRM_EPI_MUL ? ldm : 4;
    p.M = M; p.kchunks = (K + 15) / 16;
    return check_hip(nrm::gemm_nt_launch(p, pl, epilogue, (hipStream_t)stream), "gemm_nt");
}

// bf16 forms: a wave's row range is consumed 16/3x faster, so fewer, longer ranges (and fewer slabs for the reduction that
// follows): C2 gemm_tn 0.045 -> 0.036 ms per launch, slab reduction 0.20 -> 0.08 ms per step
static const int kTnWavesBf16 = 1024;
static int tn_waves(int mma = 0) {
    if (const char* e = getenv("NRM_TN_WAVES")) return atoi(e);
    return mma ? kTnWavesBf16 : kTnWaves;
}

int nrm_gemm_tn_nsplit(int ncols_i, int ncols_j, int R, int mma) {
    if (ncols_i <= 0 || ncols_j <= 0 || R <= 0) return 0;
    return nrm::gemm_tn_plan(ncols_i, ncols_j, R, tn_waves(mma), mma).nsplit;
}

int nrm_gemm_tn(const float* A, int lda, int ncols_i, const float* B, int ldb, int ncols_j, int R,
                float* ws, int ldws, float* colsum, float* zero_out, long zero_n, int mma, nrm_stream_t stream) {
    if (!A || !B || !ws) return fail(NRM_EINVAL, "nrm_gemm_tn: null pointer");
    if (ncols_i <= 0 || ncols_j <= 0 || R <= 0 || lda < ncols_i || ldb < ncols_j)
        return fail(NRM_EINVAL, "nrm_gemm_tn: ncols_i=%d ncols_j=%d R=%d lda=%d ldb=%d", ncols_i, ncols_j, R, lda, ldb);
    if (ldws % 4 || ldws < ncols_i || !al16(ws)) return fail(NRM_EINVAL, "nrm_gemm_tn: ldws=%d", ldws);
    if (int rc = check_mma("nrm_gemm_tn", mma)) return rc;
    const nrm::GemmTnPlan pl = nrm::gemm_tn_plan(ncols_i, ncols_j, R, tn_waves(mma), mma);
    if ((long)pl.rps * (lda > ldb ? lda : ldb) * 4 >= (1L << 31)) return fail(NRM_EINVAL, "nrm_gemm_tn: split too large");
    if (mma != NRM_MMA_F32 && (lda % 4 || ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)))
        return fail(NRM_EINVAL, "nrm_gemm_tn: the bf16 forms read 16-byte row segments (lda, ldb multiples of 4, 16-byte aligned rows)");
    nrm::GemmTnParams p = {};
    p.A = A; p.lda = lda; p.acols = lda; p.B = B; p.ldb = ldb; p.bcols = ldb;
    if (zero_out && (zero_n < 0 || !al16(zero_out))) return fail(NRM_EINVAL, "nrm_gemm_tn: zero_out must be 16-byte aligned, zero_n >= 0");
    p.ws = ws; p.ldws = ldws; p.colsum = colsum; p.R = R; p.ncols_j = ncols_j;
    p.zero_out = zero_out; p.zero_n = zero_out ? zero_n : 0;
    if (mma != NRM_MMA_F32) return check_hip(nrm::gemm_tn_bf16_launch(p, pl, mma, (hipStream_t)stream), "gemm_tn_bf16");
    return check_hip(nrm::gemm_tn_launch(p, pl, (hipStream_t)stream), "gemm_tn");
}

int nrm_slab_reduce(const float* ws, int nsplit, int nj, int ldws, int ni, float* out, long out_istride, long out_jstride,
                    float* out2, long out2_istride, long out2_jstride, float sign2,
                    const float* vec, float* vec_out, nrm_stream_t stream) {
    if (!ws || !out) return fail(NRM_EINVAL, "nrm_slab_reduce: null pointer");
    if (nsplit <= 0 || nj <= 0 || ni <= 0 || ldws < ni) return fail(NRM_EINVAL, "nrm_slab_reduce: nsplit=%d nj=%d ni=%d ldws=%d", nsplit, nj, ni, ldws);
    if ((vec == nullptr) != (vec_out == nullptr)) return fail(NRM_EINVAL, "nrm_slab_reduce: vec and vec_out go together");
    nrm::SlabReduceParams p;
    p.ws = ws; p.nsplit = nsplit; p.nj = nj; p.ldws = ldws; p.ni = ni; p.out = out; p.ors = out_istride; p.ocs = out_jstride;
    p.out2 = out2; p.ors2 = out2_istride; p.ocs2 = out2_jstride; p.sign2 = sign2; p.vec = vec; p.vec_out = vec_out;
    return check_hip(nrm::slab_reduce_launch(p, (hipStream_t)stream), "slab_reduce");
}

int nrm_slab_reduce_multi(const nrm_slab_desc* descs, int n, nrm_stream_t stream) {
    if (n < 0 || (n > 0 && !descs)) return fail(NRM_EINVAL, "nrm_slab_reduce_multi: bad argument");
    if (n == 0) return NRM_OK;
    std::vector<nrm::SlabReduceParams> ps((size_t)n);
    for (int i = 0; i < n; ++i) {
        const nrm_slab_desc& d = descs[i];
        if (!d.ws || !d.out) return fail(NRM_EINVAL, "nrm_slab_reduce_multi: entry %d: null pointer", i);
        if (d.nsplit <= 0 || d.nj <= 0 || d.ni <= 0 || d.ldws < d.ni)
            return fail(NRM_EINVAL, "nrm_slab_reduce_multi: entry %d: nsplit=%d nj=%d ni=%d ldws=%d", i, d.nsplit, d.nj, d.ni, d.ldws);
        if ((d.vec == nullptr) != (d.vec_out == nullptr)) return fail(NRM_EINVAL, "nrm_slab_reduce_multi: entry %d: vec and vec_out go together", i);
        nrm::SlabReduceParams& p = ps[(size_t)i];
        p.ws = d.ws; p.nsplit = d.nsplit; p.nj = d.nj; p.ldws = d.ldws; p.ni = d.ni; p.out = d.out; p.ors = d.out_istride; p.ocs = d.out_jstride;
        p.out2 = d.out2; p.ors2 = d.out2_istride; p.ocs2 = d.out2_jstride; p.sign2 = d.sign2; p.vec = d.vec; p.vec_out = d.vec_out;
    }
    return check_hip(nrm::slab_reduce_multi_launch(ps.data(), n, (hipStream_t)stream), "slab_reduce_multi");
}

// ------------------------------------------------------------------------------------------- BatchNorm
static int check_bn(const char* fn, int R, int N, int ld) {
    if (R < 0 || N <= 0 || N % 4 || ld % 4 || ld < N) return fail(NRM_EINVAL, "%s: R=%d N=%d ld=%d (N, ld multiples of 4)", fn, R, N, ld);
    return NRM_OK;
}

int nrm_colreduce(int mode, const float* x, const float* dy, const float* mean, const float* rstd,
                  float* s0, float* s1, int R, int N, int ld, nrm_stream_t stream) {
    if (int rc = check_bn("nrm_colreduce", R, N, ld)) return rc;
    if (!x || !s0 || (mode >= 1 && !mean) || (mode == 2 && (!dy || !rstd || !s1)) || mode < 0 || mode > 2)
        return fail(NRM_EINVAL, "nrm_colreduce: bad argument for mode %d", mode);
    return check_hip(nrm::colred_launch(mode, x, dy, mean, rstd, s0, s1, R, N, ld, (hipStream_t)stream), "colreduce");
}

int nrm_bn_finalize(int stage, const float* s, float* out, float* running, int R, int N, float momentum, float eps,
                    nrm_stream_t stream) {
    if (!s || !out) return fail(NRM_EINVAL, "nrm_bn_finalize: null pointer");
    if ((stage != 0 && stage != 1) || R <= 0 || N <= 0) return fail(NRM_EINVAL, "nrm_bn_finalize: stage=%d R=%d N=%d", stage, R, N);
    return check_hip(nrm::bn_finalize_launch(stage, s, out, running, R, N, momentum, eps, (hipStream_t)stream), "bn_finalize");
}

int nrm_mul_bwd(const float* dy, int lddy, const float* g, int ldg, const float* e, int lde, float* dg, float* de, int ldo,
                int R, int N, nrm_stream_t stream) {
    if (!dy || !g || !e || !dg || !de) return fail(NRM_EINVAL, "nrm_mul_bwd: null pointer");
    if (R < 0 || N <= 0 || N % 4 || lddy % 4 || ldg % 4 || lde % 4 || ldo % 4 || lddy < N || ldg < N || lde < N || ldo < N ||
        !al16(dy) || !al16(g) || !al16(e) || !al16(dg) || !al16(de))
        return fail(NRM_EINVAL, "nrm_mul_bwd: R=%d N=%d (N and leading dimensions multiples of 4, 16-byte aligned rows)", R, N);
    return check_hip(nrm::mul_bwd_launch(dy, g, e, dg, de, R, N, lddy, ldg, lde, ldo, (hipStream_t)stream), "mul_bwd");
}

int nrm_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 float* y, int R, int N, int ld, nrm_stream_t stream) {
    if (int rc = check_bn("nrm_bn_apply", R, N, ld)) return rc;
    if (!x || !mean || !rstd || !gamma || !beta || !y) return fail(NRM_EINVAL, "nrm_bn_apply: null pointer");
    return check_hip(nrm::bn_apply_launch(x, mean, rstd, gamma, beta, y, R, N, ld, (hipStream_t)stream), "bn_apply");
}

int nrm_bn_backward(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                    const float* s0, const float* s1, const float* add, float* dx, int R, int N, int ld, int training,
                    nrm_stream_t stream) {
    if (int rc = check_bn("nrm_bn_backward", R, N, ld)) return rc;
    if (!dy || !rstd || !gamma || !dx || (training && (!x || !mean || !s0 || !s1)))
        return fail(NRM_EINVAL, "nrm_bn_backward: null pointer");
    return check_hip(nrm::bn_bwd_launch(x, dy, mean, rstd, gamma, s0, s1, add, dx, R, N, ld, training, (hipStream_t)stream), "bn_backward");
}


int nrm_concat_cols(const float* const* srcs, const long* lds, const int* widths, int n, float* out, int ldo, long R,
                    nrm_stream_t stream) {
    if (!srcs || !lds || !widths || !out) return fail(NRM_EINVAL, "nrm_concat_cols: null pointer");
    if (n < 1 || n > nrm::CONCAT_MAX || R < 0) return fail(NRM_EINVAL, "nrm_concat_cols: n=%d (1..%d) R=%ld", n, nrm::CONCAT_MAX, R);
    nrm::ConcatTable tab = {};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!srcs[i] || widths[i] <= 0 || lds[i] < widths[i]) return fail(NRM_EINVAL, "nrm_concat_cols: part %d (width %d, ld %ld)", i, widths[i], lds[i]);
        tab.src[i] = srcs[i]; tab.ld[i] = lds[i]; tab.start[i] = total;
        total += widths[i];
    }
    tab.n = n;
    if (ldo < total) return fail(NRM_EINVAL, "nrm_concat_cols: ldo=%d < %d columns", ldo, total);
    return check_hip(nrm::concat_cols_launch(tab, out, R, ldo, total, (hipStream_t)stream), "concat_cols");
}

// ------------------------------------------------------------------------------------------- pool / loss / Adam
int nrm_pool_bmm(const float* W, long wsb, long wsi, long wsj, const float* X, int ldx, float* out,
                 int B, int I, int J, int D, int accumulate, nrm_stream_t stream) {
    if (!W || !X || !out) return fail(NRM_EINVAL, "nrm_pool_bmm: null pointer");
    if (B < 0 || I <= 0 || J <= 0 || D <= 0 || D % 4 || B > 65535) return fail(NRM_EINVAL, "nrm_pool_bmm: B=%d I=%d J=%d D=%d", B, I, J, D);
    if (ldx < D || ldx % 4 || !al16(X) || (long)J * ldx * 4 >= (1L << 31))
        return fail(NRM_EINVAL, "nrm_pool_bmm: ldx=%d (>= D, a multiple of 4, X 16-byte aligned, J*ldx*4 < 2^31)", ldx);
    return check_hip(nrm::bmm_rows_launch(W, wsb, wsi, wsj, X, (long)J * ldx, ldx, out, (long)I * D, D, B, I, J, D, accumulate,
                                          (hipStream_t)stream), "pool_bmm");
}

int nrm_pool_rowdot(const float* g, int ldg, const float* h, float* ds, int B, int T, int H, int D, float* zero_out, int zero_n,
                    nrm_stream_t stream) {
    if (!g || !h || !ds) return fail(NRM_EINVAL, "nrm_pool_rowdot: null pointer");
    if (B < 0 || T <= 0 || H <= 0 || D <= 0 || D % 4 || D > 1024 || B > 65535) return fail(NRM_EINVAL, "nrm_pool_rowdot: B=%d T=%d H=%d D=%d", B, T, H, D);
    if (ldg < D || ldg % 4 || !al16(g) || (long)T * ldg * 4 >= (1L << 31))
        return fail(NRM_EINVAL, "nrm_pool_rowdot: ldg=%d (>= D, a multiple of 4, g 16-byte aligned, T*ldg*4 < 2^31)", ldg);
    if (zero_n < 0 || (zero_n > 0 && !zero_out)) return fail(NRM_EINVAL, "nrm_pool_rowdot: zero_n=%d without zero_out", zero_n);
    return check_hip(nrm::rowdot_launch(g, (long)T * ldg, ldg, h, (long)H * D, D, ds, B, T, H, D, zero_out, zero_n, (hipStream_t)stream), "pool_rowdot");
}

int nrm_small_linear_relu_fwd(const void* x, int x_is_f64, const float* weight, const float* bias, float* y, long R, int K, int N, int ldy,
                              nrm_stream_t stream) {
    if (!x || !weight || !y) return fail(NRM_EINVAL, "nrm_small_linear_relu_fwd: null pointer");
    if (R < 0 || K < 1 || K > 4 || N < 1 || N > 8 || ldy < N) return fail(NRM_EINVAL, "nrm_small_linear_relu_fwd: R=%ld K=%d N=%d ldy=%d (K <= 4, N <= 8)", R, K, N, ldy);
    return check_hip(nrm::small_linear_relu_fwd_launch(x, x_is_f64, weight, bias, y, R, K, N, ldy, (hipStream_t)stream), "small_linear_relu_fwd");
}

int nrm_small_linear_relu_bwd(const void* x, int x_is_f64, const float* weight, const float* bias, const float* dy, int lddy, long R,
                              int K, int N, float* dwb, nrm_stream_t stream) {
    if (!x || !weight || !dy || !dwb) return fail(NRM_EINVAL, "nrm_small_linear_relu_bwd: null pointer");
    if (R < 0 || K < 1 || K > 4 || N < 1 || N > 8 || lddy < N) return fail(NRM_EINVAL, "nrm_small_linear_relu_bwd: R=%ld K=%d N=%d lddy=%d (K <= 4, N <= 8)", R, K, N, lddy);
    return check_hip(nrm::small_linear_relu_bwd_launch(x, x_is_f64, weight, bias, dy, lddy, R, K, N, dwb, (hipStream_t)stream), "small_linear_relu_bwd");
}

int nrm_loss_fwd_bwd(const float* out, int out_stride, const void* label, int label_is_f64, const long* user_id, const float* delta,
                     long n_delta, float alpha, int B, int T, float* loss_sum, float* dout, int dout_stride, float* ddelta, int* err,
                     nrm_stream_t stream) {
    if (!out || !label || !user_id || !delta || !loss_sum || !dout || !ddelta || !err) return fail(NRM_EINVAL, "nrm_loss_fwd_bwd: null pointer");
    if (B < 0 || T <= 0 || n_delta < 1 || (long)B * T >= (1L << 29))
        return fail(NRM_EINVAL, "nrm_loss_fwd_bwd: B=%d T=%d n_delta=%ld (T >= 1, n_delta >= 1, B*T < 2^29)", B, T, n_delta);
    if (out_stride < 1 || dout_stride < 1 || (dout_stride == 4 && !al16(dout)))
        return fail(NRM_EINVAL, "nrm_loss_fwd_bwd: out_stride=%d dout_stride=%d (>= 1; dout 16-byte aligned for stride 4)", out_stride, dout_stride);
    return check_hip(nrm::loss_launch(out, out_stride, label, label_is_f64, user_id, delta, n_delta, alpha, B, T, loss_sum, dout, dout_stride,
                                      ddelta, err, (hipStream_t)stream), "loss");
}

int nrm_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, int zero_grad, nrm_stream_t stream) {
    if (!p || !g || !m || !v) return fail(NRM_EINVAL, "nrm_adam_step: null pointer");
    if (n < 0 || n % 4 || step < 1 || !al16(p) || !al16(g) || !al16(m) || !al16(v))
        return fail(NRM_EINVAL, "nrm_adam_step: n=%ld step=%d (n %% 4 == 0, step >= 1, 16-byte aligned buffers)", n, step);
    return check_hip(nrm::adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, zero_grad, (hipStream_t)stream), "adam");
}


int nrm_gather_flat(const float* const* srcs, const long* offsets, const long* counts, int n, float* flat, long flat_n,
                    nrm_stream_t stream) {
    if ((n > 0 && (!srcs || !offsets || !counts)) || !flat) return fail(NRM_EINVAL, "nrm_gather_flat: null pointer");
    if (n < 0) return fail(NRM_EINVAL, "nrm_gather_flat: n=%d", n);
    for (int lo = 0; lo < n; lo += nrm::GATHER_MAX) {
        const int m = n - lo < nrm::GATHER_MAX ? n - lo : nrm::GATHER_MAX;
        nrm::GatherTable tab;
        long mx = 0;
        for (int i = 0; i < m; ++i) {
            if (offsets[lo + i] < 0 || counts[lo + i] < 0 || offsets[lo + i] + counts[lo + i] > flat_n)
                return fail(NRM_EINVAL, "nrm_gather_flat: entry %d (offset %ld, count %ld) leaves the flat buffer of %ld floats",
                            lo + i, offsets[lo + i], counts[lo + i], flat_n);
            tab.src[i] = srcs[lo + i]; tab.off[i] = offsets[lo + i]; tab.cnt[i] = counts[lo + i];
            if (counts[lo + i] > mx) mx = counts[lo + i];
        }
        if (int rc = check_hip(nrm::gather_flat_launch(tab, m, mx, flat, (hipStream_t)stream), "gather_flat")) return rc;
    }
    return NRM_OK;
}

int nrm_adam_step_dev(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float* state, int zero_grad, nrm_stream_t stream) {
    if (!p || !g || !m || !v || !state) return fail(NRM_EINVAL, "nrm_adam_step_dev: null pointer");
    if (n < 0 || n % 4 || !al16(p) || !al16(g) || !al16(m) || !al16(v))
        return fail(NRM_EINVAL, "nrm_adam_step_dev: n=%ld (n %% 4 == 0, 16-byte aligned buffers)", n);
    return check_hip(nrm::adam_dev_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, state, zero_grad, (hipStream_t)stream), "adam_dev");
}

int nrm_row_auc(const float* score, const float* label, const int* len, int B, int T, float* auc, int* top1,
                nrm_stream_t stream) {
    if (!score || !label || !auc || !top1) return fail(NRM_EINVAL, "nrm_row_auc: null pointer");
    if (B < 0 || T <= 0) return fail(NRM_EINVAL, "nrm_row_auc: B=%d T=%d", B, T);
    return check_hip(nrm::row_auc_launch(score, label, len, B, T, auc, top1, (hipStream_t)stream), "row_auc");
}

// ------------------------------------------------------------------------------------------- embedding front end
static int check_fe(const char* fn, int nrows, int xcols, int P, int n_sub, int behaviour, int e0, int e1, int e2, int e3) {
    if (nrows < 0 || P <= 0 || n_sub < 0 || e0 <= 0 || e1 <= 0 || e2 <= 0 || e3 <= 0)
        return fail(NRM_EINVAL, "%s: bad dimension", fn);
    if (n_sub > 16) return fail(NRM_EINVAL, "%s: n_sub=%d (<= 16 sub-category slots)", fn, n_sub);
    const int need = 4 + P + 1 + n_sub + 3 + 1 + (behaviour ? 2 : 0);
    if (xcols < need) return fail(NRM_EINVAL, "%s: packed rows have %d columns, layout needs %d", fn, xcols, need);
    return NRM_OK;
}

int nrm_frontend_fwd(const void* x, int x_is_f64, int nrows, int xcols, int P, int n_sub, int behaviour,
                     const float* cat_tab, int n_cat, int e0, const float* sen_w, const float* sen_b, int e1,
                     const float* type_tab, int n_type, int e2,
                     const float* year_tab, const float* month_tab, const float* day_tab, const float* hour_tab,
                     int n_year, int n_month, int n_day, int n_hour, int e3,
                     float* lab, int ldlab, float* ti, int ldti, int* err, nrm_stream_t stream) {
    if (int rc = check_fe("nrm_frontend_fwd", nrows, xcols, P, n_sub, behaviour, e0, e1, e2, e3)) return rc;
    if (!x || !cat_tab || !sen_w || !sen_b || !type_tab || !year_tab || !month_tab || !day_tab || !hour_tab || !lab || !ti || !err)
        return fail(NRM_EINVAL, "nrm_frontend_fwd: null pointer");
    if (ldlab < e0 + e1 + e2 + e3 + (behaviour ? 2 : 0) || ldti < P) return fail(NRM_EINVAL, "nrm_frontend_fwd: ldlab=%d ldti=%d too small", ldlab, ldti);
    nrm::FrontendParams p = {};
    p.cat_tab = cat_tab; p.sen_w = sen_w; p.sen_b = sen_b; p.type_tab = type_tab;
    p.year_tab = year_tab; p.month_tab = month_tab; p.day_tab = day_tab; p.hour_tab = hour_tab;
    p.n_cat = n_cat; p.n_type = n_type; p.n_year = n_year; p.n_month = n_month; p.n_day = n_day; p.n_hour = n_hour;
    p.e0 = e0; p.e1 = e1; p.e2 = e2; p.e3 = e3; p.P = P; p.n_sub = n_sub; p.xcols = xcols; p.behaviour = behaviour;
    p.lab = lab; p.ldlab = ldlab; p.ti = ti; p.ldti = ldti; p.err = err;
    return check_hip(nrm::frontend_fwd_launch(p, x, x_is_f64, nrows, (hipStream_t)stream), "frontend_fwd");
}

int nrm_frontend_bwd(const void* x, int x_is_f64, int nrows, int xcols, int P, int n_sub, int behaviour,
                     const float* dlab, int lddl, const float* sen_w, const float* sen_b,
                     int n_cat, int e0, int e1, int n_type, int e2, int n_year, int n_month, int n_day, int n_hour, int e3,
                     float* d_cat_tab, float* d_sen_w, float* d_sen_b, float* d_type_tab,
                     float* d_year_tab, float* d_month_tab, float* d_day_tab, float* d_hour_tab, nrm_stream_t stream) {
    if (int rc = check_fe("nrm_frontend_bwd", nrows, xcols, P, n_sub, behaviour, e0, e1, e2, e3)) return rc;
    if (!x || !dlab || !sen_w || !sen_b || !d_sen_w || !d_sen_b || !d_type_tab || !d_year_tab || !d_month_tab || !d_day_tab || !d_hour_tab)
        return fail(NRM_EINVAL, "nrm_frontend_bwd: null pointer");
    if (lddl < e0 + e1 + e2 + e3) return fail(NRM_EINVAL, "nrm_frontend_bwd: lddl=%d too small", lddl);
    nrm::FrontendParams p = {};
    p.sen_w = sen_w; p.sen_b = sen_b;
    p.d_cat_tab = d_cat_tab; p.d_sen_w = d_sen_w; p.d_sen_b = d_sen_b; p.d_type_tab = d_type_tab;
    p.d_year_tab = d_year_tab; p.d_month_tab = d_month_tab; p.d_day_tab = d_day_tab; p.d_hour_tab = d_hour_tab;
    p.n_cat = n_cat; p.n_type = n_type; p.n_year = n_year; p.n_month = n_month; p.n_day = n_day; p.n_hour = n_hour;
    p.e0 = e0; p.e1 = e1; p.e2 = e2; p.e3 = e3; p.P = P; p.n_sub = n_sub; p.xcols = xcols; p.behaviour = behaviour;
    return check_hip(nrm::frontend_bwd_launch(p, x, x_is_f64, dlab, lddl, nrows, (hipStream_t)stream), "frontend_bwd");
}

long nrm_frontend_cat_ws_ints(int n_cat, long nrows_total, int n_sub) {
    if (n_cat <= 0 || nrows_total < 0 || n_sub < 0) return 0;
    return nrm::cat_grad_ws_ints(n_cat, nrows_total, n_sub);
}

int nrm_frontend_cat_grad(const void* x0, int nrows0, int xcols0, const float* dlab0, int lddl0,
                          const void* x1, int nrows1, int xcols1, const float* dlab1, int lddl1, int x_is_f64,
                          int P, int n_sub, int n_cat, int e0, float* d_cat_tab, int* ws, nrm_stream_t stream) {
    if (nrows0 < 0 || nrows1 < 0 || P <= 0 || n_sub < 1 || n_sub > 16 || n_cat <= 0 || e0 <= 0 || e0 > 512)
        return fail(NRM_EINVAL, "nrm_frontend_cat_grad: nrows=%d,%d P=%d n_sub=%d n_cat=%d e0=%d (1 <= n_sub <= 16, e0 <= 512)", nrows0, nrows1, P, n_sub, n_cat, e0);
    if ((nrows0 > 0 && (!x0 || !dlab0 || lddl0 < e0 || xcols0 < 4 + P + 1 + n_sub)) || (nrows1 > 0 && (!x1 || !dlab1 || lddl1 < e0 || xcols1 < 4 + P + 1 + n_sub)) || !d_cat_tab || !ws)
        return fail(NRM_EINVAL, "nrm_frontend_cat_grad: null pointer, lddl < e0 or packed rows too short");
    if (((long)nrows0 + nrows1) * (n_sub + 1) >= (1L << 31)) return fail(NRM_EINVAL, "nrm_frontend_cat_grad: more than 2^31 table references");
    return check_hip(nrm::cat_grad_launch(x0, nrows0, xcols0, dlab0, lddl0, x1, nrows1, xcols1, dlab1, lddl1, x_is_f64, P, n_sub, n_cat, e0,
                                          d_cat_tab, ws, (hipStream_t)stream), "frontend_cat_grad");
}

}  // extern "C"
