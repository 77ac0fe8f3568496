// Parameter blocks and host-side launchers shared by the pointwise-attention kernels and capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nrm {

// ---- forward (pwattn_fwd.hip)
struct FwdParams {
    const float* t;      // [B*T, ldt]
    const float* h;      // [B*H, ldh]
    const float* u;      // [B*H, ldu]   (includes fc1 bias)
    const float* v;      // [B*T, ldv]
    const float* wp;     // packed, see pack_wp_kernel
    const float* w2;     // [D]
    const float* b2;     // [1]
    float* z;            // [M, D] or nullptr
    float* s;            // [M]
    long M;              // B*T*H
    int T, H, D;
    int ldt, ldh, ldu, ldv;
    unsigned wp_bytes, t_bytes, h_bytes;   // buffer-descriptor extents (t/v share t_bytes, h/u share h_bytes)
    int rows;            // padded row count of packed W_p  (= nchunks * NT * 16)
    int kchunks;         // K-chunks of the packed W_p: ceil(D/16) (fp32 operands) or ceil(D/32) (bf16 operands)
    int nchunks;         // number of N-chunks (each NT*16 output columns)
};
struct FwdPlan { int NT, MT, nchunks, rows, kchunks; };
FwdPlan pwattn_fwd_plan(int D);
hipError_t pwattn_fwd_launch(const FwdParams& p, const FwdPlan& pl, int mma, hipStream_t st);
hipError_t pack_wp_launch(const float* w, int ldw, int D, const FwdPlan& pl, int mma, float* packed, hipStream_t st);
// resident-W forward (pwattn_fwd_rw.hip): mma = 0 (fp32 MFMA), 1 (bf16 operands) or 2 (bf16x3: hi + lo split).  The output
// columns are cut into nsplit slices of nts 16-column tiles whose whole image stays resident in LDS (persistent workgroups).
struct RwPlan { int nts, nsplit, rows, k32, wimg; };      // rows = padded rows of the packed image = nsplit * nts * 16; k32 = chunks
RwPlan pwattn_rw_plan(int D, int mma);
bool pwattn_fwd_uses_rw(int D, int mma);                  // which forward (and which packed layout) a call takes
hipError_t pwattn_fwd_rw_launch(const FwdParams& p, int mma, hipStream_t st);
hipError_t pack_wp_bf16_launch(const float* w, int ldw, int D, int mma, float* packed, hipStream_t st);

// ---- backward (pwattn_bwd.hip)
struct BwdEParams {
    const float* X; long xs1, xs2, xrs;   // X_g[r,k] at X[g1*xs1 + g2*xs2 + r*xrs + k],  g = g1*G2 + g2
    const float* Y; long ys1, yrs;        // Y_g[r,d] at Y[g1*ys1 + r*yrs + d]
    const float* wp; int ldwp;            // W_p[k,d] at wp[k*ldwp + d]
    const float* srow; int lds_;          // scale row of group g at srow[g*lds_ + d]     (WITH_DW)
    float* out; int ldo;                  // out[g*ldo + d] += sum_k W_p[k,d] E_g[k,d]
    float* ws;                            // [nsplit][D][D] partial dW_p^T  (ws[s][d][k])      (WITH_DW)
    int G, G2, R, D;
    int gps;                              // groups per split
    int nkw, ndcol;                       // k-ranges and d-columns of the wave-tile grid
    int nsplit;
    int interleave;                       // serial kernel: the 4 waves of a workgroup walk one group range round-robin
    int x_hl4;                            // X is in the NRM_DZ_HL4 format (bf16 forms only): no conversion of the dz operand
    int with_dt;                          // 0: the (b,t)-grouped pass only accumulates dW_p (dt comes from pwattn_bwd_rw.hip)
    int order;                            // serial kernel, block order inside an XCD: 0 = (split group, k-range, d-column), 1 = (k-range, split group, d-column)
};
struct BwdEPlan { int DT, KT, ndcol, nkw, nsplit, gps; };
BwdEPlan bwd_e_plan(int D, int G, int target_waves, int min_gps = 1, int mma = 0);
hipError_t bwd_e_launch(const BwdEParams& p, const BwdEPlan& pl, bool with_dw, int mma, hipStream_t st);
// dz_format: 0 = fp32 dz in place, 1 = NRM_DZ_HL4 (every aligned group of 4 values as 4 bf16 hi + 4 bf16 lo, in place)
hipError_t bwd_dz_launch(float* z, const float* ds, const float* w2, float* dw2, float* db2, float* du, float* dv,
                         int B, int T, int H, int D, int dz_format, hipStream_t st);

// ---- backward of the bilinear term, resident-W form for the bf16 arithmetics (pwattn_bwd_rw.hip): dt and dh from ONE read of dz
struct BwdRwParams {
    const float* dz;     // [B*T*H, D] in the NRM_DZ_HL4 format
    const float* t;      // [B*T, D]
    const float* h;      // [B*H, D]
    const float* wimg;   // packed by pwattn_bwd_rw_pack_launch
    float* dt;           // [B*T, D]  +=
    float* dh;           // [B*H, D]  +=
    int B, T, H, D;
    unsigned w_bytes;
    int tsplit;          // parts of the candidate walk (set by the launcher)
};
struct BwdRwPlan { int ng, kc, nks, wimg, k32, rows; };   // ng == 0: this (D, mma) keeps the E-form
BwdRwPlan pwattn_bwd_rw_plan(int D, int mma);
long pwattn_bwd_rw_packed_floats(int D, int mma);
hipError_t pwattn_bwd_rw_pack_launch(const float* wp, int ldw, int D, int mma, float* packed, hipStream_t st);
hipError_t pwattn_bwd_rw_launch(const BwdRwParams& p, int mma, hipStream_t st);

// ---- backward of the bilinear term in fp32, dP walk form (pwattn_bwd_dp.hip): dt and dh from ONE contraction dP = dz W_p with the
// forward's K-chunk-streaming skeleton; dW_p then comes from the (b,t) pass without its dt epilogue
struct BwdDpParams {
    const float* dz;     // [B*T*H, D] fp32
    const float* t;      // [B*T, D]
    const float* h;      // [B*H, D]
    const float* wimg;   // packed by pwattn_bwd_dp_pack_launch
    float* dt;           // [B*T, D]  +=
    float* dh;           // [B*H, D]  +=
    int B, T, H, D;
    unsigned w_bytes, t_bytes, h_bytes;
    int rows, kchunks, nchunks;   // of the plan (set by the launcher)
    long steps;                   // (row blocks of 64) x N-chunks x T (set by the launcher)
};
struct BwdDpPlan { int NT, nchunks, rows, kchunks; };     // NT == 0: this (D, H) keeps the E-form
BwdDpPlan pwattn_bwd_dp_plan(int D, int H);
long pwattn_bwd_dp_packed_floats(int D, int H);
hipError_t pwattn_bwd_dp_pack_launch(const float* wp, int ldw, int D, int H, float* packed, hipStream_t st);
hipError_t pwattn_bwd_dp_launch(BwdDpParams p, hipStream_t st);

// Non-zero when the translation unit was compiled with a timing-diagnostic override (scripts/_diag): such a library computes
// WRONG results by construction; capi.hip ORs these into nrm_build_flags() and native.load refuses a non-zero value.
int pwattn_fwd_diag_flags();        // pwattn_fwd.hip:     bit 0 NRM_DIAG_FWD, bit 1 XCD_REMAP off
int pwattn_fwd_rw_diag_flags();     // pwattn_fwd_rw.hip:  bit 2 NRM_DIAG_RW
int pwattn_bwd_rw_diag_flags();     // pwattn_bwd_rw.hip:  bit 9 NRM_DIAG_BRW
int pwattn_bwd_dp_diag_flags();     // pwattn_bwd_dp.hip:  bit 10 NRM_DIAG_DP
int pwattn_bwd_diag_flags();        // pwattn_bwd.hip:     bits 3.. NRM_EPI_AHEAD off, GELU_AT_LOAD, NOEPI, NOATOM, NOLOAD, NRM_PIPE_SGB off

}  // namespace nrm
