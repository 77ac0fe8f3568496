// Parameter blocks of the dense-layer GEMMs (gemm.hip), shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nrm {

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_MUL = 3 };

struct GemmNtParams {
    const float* x; int ldx, xcols;       // [M, ldx]; xcols = readable floats per row (<= ldx)
    const float* wp; unsigned wp_bytes;   // packed weight rows (pack_rows_kernel)
    int rows;                             // padded packed rows = nchunks * NT * 16
    const float* bias; int N;             // [N] or nullptr
    float* y; int ldy;                    // [M, ldy]
    float* z; int ldz;                    // EPI_GELU: pre-activation out, EPI_DGELU: pre-activation in, EPI_MUL: un-multiplied out
    const float* m; int ldm;              // EPI_MUL: y = (x W^T + bias) * m   (the gate of user_model.py:33)
    int M, kchunks;
};
struct GemmNtPlan { int NT, MT, nchunks, rows; };
GemmNtPlan gemm_nt_plan(int N);
hipError_t gemm_nt_launch(const GemmNtParams& p, const GemmNtPlan& pl, int epi, hipStream_t st);
hipError_t pack_rows_launch(const float* src, long rs, long cs, int nrows, int ncols, int rows, int kchunks,
                            float* packed, hipStream_t st);

// gemm_nt_rx (gemm_bf16.hip): the bf16 / bf16x3 forward GEMM with the workgroup's X rows resident in LDS
struct GemmRxParams {
    const float* x; int ldx, xcols;       // [M, ldx]
    const float* wq; unsigned wq_bytes;   // weight fragments wq[it][c][img][16][32 bf16] (PackEntry fmt 1 | 2)
    const float* bias; int N;
    float* y; int ldy;
    float* z; int ldz;
    const float* m; int ldm;
    int M, K, k32, nt16;                  // k32 = ceil(K / 32) reduction chunks, nt16 = ceil(N / 16) output-column tiles
};
int gemm_bf16_diag_flags();                                // bit 10: NRM_DIAG_RX (timing-only build)
int gemm_nt_rx_bm(int M, int K, int mma);                  // rows per workgroup; 0: K too wide (the caller keeps the fp32 GEMM)
hipError_t gemm_nt_rx_launch(const GemmRxParams& p, int epi, int mma, hipStream_t st);

constexpr int PACK_MAX = 36;
// fmt 0: the fp32 image of gemm_nt (rows / kchunks as planned); fmt 1 | 2: bf16 hi [+ lo] weight fragments of gemm_nt_rx
// (rows = 16 * ceil(nrows / 16), kchunks = ceil(ncols / 32))
struct PackEntry { const float* src; const float* src2; float* dst; long rs, cs; int nrows, ncols, rows, kchunks; float sign2; int fmt; };
struct PackTable { PackEntry e[PACK_MAX]; };          // 40 x 64 B of kernel arguments
hipError_t pack_rows_multi_launch(const PackTable& tab, int n, long max_total, hipStream_t st);

struct GemmTnParams {
    const float* A; int lda, acols;       // [R, lda]   i-columns
    const float* B; int ldb, bcols;       // [R, ldb]   j-columns
    float* ws; int ldws;                  // [nsplit][ncols_j][ldws]   C^T partial slabs
    float* colsum;                        // [nsplit][ldws] column sums of A, or nullptr
    float* zero_out; long zero_n;         // optional: zero_n floats at zero_out (16-byte aligned) are set to 0 by this launch
    int R, ncols_j;
    int nti, nsplit, rps;                 // filled by gemm_tn_launch from the plan
};
struct GemmTnPlan { int KT, DT, nti, ntj, nsplit, rps; };      // wave tile = (KT*16 i) x (DT*16 j); target_waves <= 0: one round at the kernel's occupancy
GemmTnPlan gemm_tn_plan(int ncols_i, int ncols_j, int R, int target_waves, int mma = 0);
hipError_t gemm_tn_launch(GemmTnParams p, const GemmTnPlan& pl, hipStream_t st);
hipError_t gemm_tn_bf16_launch(GemmTnParams p, const GemmTnPlan& pl, int mma, hipStream_t st);      // gemm_bf16.hip

// out[i*ors + j*ocs] += sum_s ws[s][j][i] (i < ni, j < nj): the split slabs of gemm_tn / the dW_p pass summed and
// ADDED where the gradient lives (any strides: straight or transposed, or a column block of a wider matrix; float
// atomics, the caller zero-initialises); out2 (optional) += sign2 * the same value; vec_out[i] = sum_s vec[s][i].
struct SlabReduceParams {
    const float* ws; int nsplit, nj, ldws, ni;
    float* out; long ors, ocs;
    float* out2; long ors2, ocs2; float sign2;
    const float* vec; float* vec_out;     // [nsplit][ldws] -> [ni], or nullptr
};
hipError_t slab_reduce_launch(const SlabReduceParams& p, hipStream_t st);
constexpr int SLAB_MAX = 24;              // slab sets per multi launch (the table is a kernel argument: 24 x 96 B + 4 x 96 B + 4)
struct SlabTable { SlabReduceParams e[SLAB_MAX]; int spc[SLAB_MAX]; int nchunk[SLAB_MAX]; int gx[SLAB_MAX]; int first[SLAB_MAX]; int n; };
hipError_t slab_reduce_multi_launch(const SlabReduceParams* ps, int n, hipStream_t st);
int gemm_diag_flags();

}  // namespace nrm
