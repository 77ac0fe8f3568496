/* nrm_hotpath.h -- C ABI of the MI355X (gfx950) hot path of News_Recommendation_Model.
 *
 * The reference has no FFI: its boundary is Python nn.Module.  This library is
 * what the Modules in news_recommendation_model_amd/modules.py bind through ctypes; every entry point
 * names the reference code it replaces.  Conventions:
 *   - all pointers are DEVICE pointers to fp32 (or as stated), row-major, caller-owned ("borrowed");
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it and never
 *     synchronises or allocates;
 *   - return value 0 = ok, otherwise a negative NRM_E* code; nrm_last_error() gives the text
 *     (thread-local).  Shapes are validated on the host before any launch;
 *   - D (feature width of one attention) must be a multiple of 4; B*T*H < 2^31.
 */
#ifndef NRM_HOTPATH_H
#define NRM_HOTPATH_H

#ifdef __cplusplus
extern "C" {
#endif

#define NRM_ABI_VERSION 6
#define NRM_OK 0
#define NRM_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define NRM_ELAUNCH (-2)  /* HIP launch error */

typedef void* nrm_stream_t;

int nrm_abi_version(void);
const char* nrm_last_error(void);
/* 0 for a product build.  Non-zero: the library was compiled with one of the timing-diagnostic overrides of scripts/_diag
 * (kernels with parts of their work removed: results are WRONG by construction); the Python binding refuses to load such a
 * library unless NRM_ALLOW_DIAG_LIB=1. */
int nrm_build_flags(void);
/* Provenance (round 5): the sha256 over the kernel sources (csrc/ and this header, news_recommendation_model_amd/build.py
 * sources_digest()) this library was compiled from, and a one-line build record (compiler, flags, UTC time).  The Python
 * binding refuses a library whose digest differs from the sources lying next to it (NRM_ALLOW_STALE_LIB=1 overrides), bench.py
 * prints both, so a measured binary is tied to the sources it is shown with. */
const char* nrm_source_digest(void);
const char* nrm_build_info(void);

/* ---- pointwise history attention: reference models/attention_model.py:52-97
 *      (PointwiseAttentionExpanded.forward), score[b,t,h] = fc2(GELU(fc1(cat[h,t,t-h,t*h]))).
 * The caller supplies the two side projections of the split fc1 = [W_h | W_t | W_d | W_p]:
 *      u[b,h,:] = h[b,h,:] (W_h - W_d)^T + fc1.bias        v[b,t,:] = t[b,t,:] (W_t + W_d)^T          */

/* Arithmetic of the bilinear contraction, chosen per call (BASELINE config 2 names bf16, config 3 fp32):
 *   NRM_MMA_F32   v_mfma_f32_16x16x4_f32: exact fp32 products and sums
 *   NRM_MMA_BF16  v_mfma_f32_16x16x32_bf16: operands rounded to bf16 (the t*h product is formed in fp32 and rounded
 *                 once, W_p / dz / h / t are rounded as they are read), fp32 accumulation starting from the fp32 u + v;
 *                 GELU, the fc2 dot, scores, pool and every reduction stay fp32
 *   NRM_MMA_BF16X3  the same instruction with both operands split into hi + lo bf16 parts (lo = rounding remainder) and
 *                 three MFMAs per product (lo*hi + hi*lo + hi*hi): fp32-class accuracy (~2^-16 per product) at 3/16 of the
 *                 fp32 MFMA time -- the arithmetic that keeps the <= 1e-3 parity gate of BASELINE with bf16 matrix cores */
#define NRM_MMA_F32 0
#define NRM_MMA_BF16 1
#define NRM_MMA_BF16X3 2
/* floats to allocate for the packed copy of W_p used by nrm_pwattn_fwd (either arithmetic) */
long nrm_pwattn_packed_floats(int D);
/* fc1_weight: [D, 4D] with row stride ld; packs W_p = fc1_weight[:, 3D:4D] for the given arithmetic */
int nrm_pwattn_pack_wp(const float* fc1_weight, int ld, int D, int mma, float* packed, nrm_stream_t stream);
/* t [B,T,D], h [B,H,D], u [B,H,D], v [B,T,D] contiguous; w2 = fc2.weight [D]; b2 = fc2.bias [1];
 * z [B,T,H,D] pre-activation saved for backward (NULL in inference); s [B,T,H] scores */
int nrm_pwattn_fwd(const float* t, const float* h, const float* u, const float* v, const float* packed_wp,
                   const float* w2, const float* b2, float* z, float* s,
                   int B, int T, int H, int D, int mma, nrm_stream_t stream);

/* backward, step 1 (autograd of attention_model.py:29-32 through GELU and fc2), one pass over z:
 *   z <- dz = ds * w2 * gelu'(z) in place;  dw2[k] += sum ds*gelu(z)   (dw2 must be initialised)
 *   du[b,h,:] = sum_t dz[b,t,h,:]  (gradient of u)      dv[b,t,:] = sum_h dz[b,t,h,:]  (gradient of v)
 * ds [B,T,H]; du [B,H,D] and dv [B,T,D] are overwritten (histories longer than 256 rows are processed in chunks).
 *   db2 (optional, may be NULL): *db2 += sum ds, the fc2 bias gradient (must be initialised like dw2) */
/* dz_format: how dz is left in z_inout.  NRM_DZ_F32: fp32.  NRM_DZ_HL4 (for the bf16 arithmetics): every aligned group of 4
 * values as 4 bf16 hi + 4 bf16 lo (lo = bf16 of the rounding remainder) in the same 16 bytes -- the MFMA-ready operand that
 * nrm_pwattn_bwd_rw_dtdh and the dW_p-only pass of nrm_pwattn_bwd_contract read without conversion (du, dv, dw2 are computed
 * from the fp32 values either way). */
#define NRM_DZ_F32 0
#define NRM_DZ_HL4 1
int nrm_pwattn_bwd_dz(float* z_inout, const float* ds, const float* w2, float* dw2, float* db2, float* du, float* dv,
                      int B, int T, int H, int D, int dz_format, nrm_stream_t stream);
/* number of [D,D] partial slabs nrm_pwattn_bwd_contract writes into `ws` (depends on the arithmetic: tile shapes differ) */
int nrm_pwattn_bwd_nsplit(int B, int T, int H, int D, int mma);
/* backward, step 2 (the bilinear term): given dz [B,T,H,D], t, h and W_p (row stride ldwp)
 *   dt[b,t,d] += sum_{h,k} dz W_p[k,d] h[b,h,d]      dh[b,h,d] += sum_{t,k} dz W_p[k,d] t[b,t,d]
 *   ws[i][d][k] (i < nsplit) = partial of dW_p[k,d] = sum_{b,t,h} dz[b,t,h,k] t[b,t,d] h[b,h,d]
 *                              (TRANSPOSED slabs: sum them over i, then transpose)
 * dt/dh are accumulated into (float atomics), ws is overwritten.
 * passes: bit 0 = the (b,t)-grouped launch (dt, ws), bit 1 = the (b,h)-grouped launch (dh); 3 = both.
 * passes = 4: the (b,t)-grouped launch WITHOUT its dt epilogue -- only ws (dt, dh may be NULL).  For a caller that wants no
 * row gradients at all (the model's text+image attention reads raw input columns: neither t nor h has a gradient), or -- bf16
 * arithmetics with dz in NRM_DZ_HL4 -- beside nrm_pwattn_bwd_rw_dtdh, which then delivers dt and dh.
 * dz_format names the layout of dz (NRM_DZ_HL4 only with passes = 4 and a bf16 arithmetic). */
int nrm_pwattn_bwd_contract(const float* dz, const float* t, const float* h, const float* wp, int ldwp,
                            float* dt, float* dh, float* ws,
                            int B, int T, int H, int D, int passes, int mma, int dz_format, nrm_stream_t stream);
/* backward, step 2 in the "resident W_p" form (bf16 arithmetics, D <= 256): the same dt / dh as above from ONE contraction
 * dP = dz W_p whose W_p image stays in LDS and whose dz operand (NRM_DZ_HL4) is read exactly once; autograd of
 * models/attention_model.py:81-92 w.r.t. target and history.  nrm_pwattn_bwd_rw_supported: 1 if (D, mma) has this form;
 * packed: nrm_pwattn_bwd_rw_packed_floats floats filled by nrm_pwattn_bwd_rw_pack from fc1_weight [D, 4D] (row stride ld). */
int nrm_pwattn_bwd_rw_supported(int D, int mma);
long nrm_pwattn_bwd_rw_packed_floats(int D, int mma);
int nrm_pwattn_bwd_rw_pack(const float* fc1_weight, int ld, int D, int mma, float* packed, nrm_stream_t stream);
int nrm_pwattn_bwd_rw_dtdh(const float* dz_hl4, const float* t, const float* h, const float* packed, float* dt, float* dh,
                           int B, int T, int H, int D, int mma, nrm_stream_t stream);

/* backward, step 2 in the "dP walk" form (fp32, D % 4 == 0, H >= 16): the same dt / dh as nrm_pwattn_bwd_contract passes 1 + 2
 * from ONE contraction dP = dz W_p with the forward's streaming skeleton (dz fp32, read once per 208-column chunk of d); the
 * caller then takes dW_p from nrm_pwattn_bwd_contract with passes = 4.  Autograd of models/attention_model.py:81-92 w.r.t.
 * target and history.  nrm_pwattn_bwd_dp_supported: 1 if (D, H) has this form;
 * packed: nrm_pwattn_bwd_dp_packed_floats floats filled by nrm_pwattn_bwd_dp_pack from fc1_weight [D, 4D] (row stride ld).
 * dt / dh are accumulated into (float atomics). */
int nrm_pwattn_bwd_dp_supported(int D, int H);
long nrm_pwattn_bwd_dp_packed_floats(int D, int H);
int nrm_pwattn_bwd_dp_pack(const float* fc1_weight, int ld, int D, int H, float* packed, nrm_stream_t stream);
int nrm_pwattn_bwd_dp_dtdh(const float* dz, const float* t, const float* h, const float* packed, float* dt, float* dh,
                           int B, int T, int H, int D, nrm_stream_t stream);

/* ---- dense layers: reference MLP.forward (models/attention_model.py:29-32: fc1 -> GELU -> fc2), the history
 *      projection w1 (models/user_invariant_interest_model.py:78) and the attention's side projections.
 * Weights are re-packed per call into the kernel's streaming layout (they change every optimizer step).
 * All activation matrices are row-major with a leading dimension that is a multiple of 4 floats and 16-byte
 * aligned rows; padding columns [ncols, ld) must hold finite values (the library writes zeros there).       */
#define NRM_EPI_BIAS 0   /* y = x W^T + bias                                                              */
#define NRM_EPI_GELU 1   /* z = x W^T + bias (saved for backward), y = gelu(z)                            */
#define NRM_EPI_DGELU 2  /* y = (x W^T) * gelu'(z)      (z = pre-activation saved by NRM_EPI_GELU)        */
#define NRM_EPI_MUL 3    /* z = x W^T + bias (kept for backward), y = z * m: the gate of user_model.py:33 */
/* mma: the arithmetic the packed image is for (NRM_MMA_F32: the fp32 streaming layout of gemm_nt; NRM_MMA_BF16 / _BF16X3: bf16
 * hi [+ lo] MFMA fragments).  Dense layers on the bf16 matrix cores (BASELINE config 2) keep fp32 accumulation, bias, GELU and
 * column sums; nrm_gemm_nt_bf16_supported says whether a reduction width K fits the resident-row form (else use NRM_MMA_F32). */
long nrm_gemm_packed_floats(int nrows, int ncols, int mma);
int nrm_gemm_nt_bf16_supported(int M, int K, int mma);
/* packs the logical [nrows x ncols] matrix src[r*row_stride + c*col_stride]; rows become output columns of
 * nrm_gemm_nt, columns its reduction index.  Linear.forward: (W[N,K], K, 1, N, K); dX = dY W: (W, 1, K, K, N) */
int nrm_gemm_pack(const float* src, long row_stride, long col_stride, int nrows, int ncols, float* packed,
                  nrm_stream_t stream);
/* the same for n matrices in ONE launch (descs is a HOST array, copied into the kernel arguments): what a training loop
 * calls once per optimizer step for every Linear in both GEMM orientations instead of once per GEMM.  With src2 != NULL the
 * packed matrix is src + sign2 * src2 (same strides): the attention's side projections W_h - W_d and W_t + W_d
 * (reference models/attention_model.py:81-86, re-associated) are formed here. */
typedef struct {
    const float* src; const float* src2; float sign2;
    long row_stride, col_stride; int nrows, ncols;
    float* packed;           /* nrm_gemm_packed_floats(nrows, ncols, mma) floats */
    int mma;                 /* NRM_MMA_*: layout of the packed image */
} nrm_pack_desc;
int nrm_gemm_pack_multi(const nrm_pack_desc* descs, int n, nrm_stream_t stream);
/* y[M, N] (ld ldy) = epilogue( x[M, K] (ld ldx) * packed^T ), bias [N] or NULL; m [M, N] (ld ldm) for NRM_EPI_MUL */
int nrm_gemm_nt(const float* x, int ldx, int M, const float* packed, int N, int K, const float* bias,
                float* y, int ldy, float* z, int ldz, const float* m, int ldm, int epilogue, int mma, nrm_stream_t stream);
/* C[i,j] = sum_r A[r,i] B[r,j]  (dW = dY^T X): writes nsplit TRANSPOSED partial slabs ws[s][j][ldws] and, if
 * colsum != NULL, colsum[s][i] = sum_r A[r,i] (the bias gradient); sum over s.  ldws % 4 == 0, ldws >= ncols_i */
int nrm_gemm_tn_nsplit(int ncols_i, int ncols_j, int R, int mma);
/* zero_out (optional, 16-byte aligned): zero_n floats there are set to 0 by the same launch -- the gradient buffer that
 * nrm_slab_reduce will then add the slabs to */
int nrm_gemm_tn(const float* A, int lda, int ncols_i, const float* B, int ldb, int ncols_j, int R,
                float* ws, int ldws, float* colsum, float* zero_out, long zero_n, int mma, nrm_stream_t stream);

/* the split slabs ws[s][j][ldws] of nrm_gemm_tn / nrm_pwattn_bwd_contract summed over s and ADDED (float atomics; the
 * caller zero-initialises, or accumulates on purpose) where the gradient lives:
 *     out[i*out_istride + j*out_jstride] += sum_s ws[s][j][i]      (i < ni, j < nj)
 * (nj, 1) strides give C[i,j] row-major, i.e. the transpose of a slab; (ld, 1) with an offset pointer a column block of a
 * wider matrix.  out2 (optional) += sign2 * the same value -- the (t - h) block of fc1's gradient
 * (attention_model.py:81-86) is da_t - da_h.  vec [nsplit][ldws] (optional, the colsum slabs of nrm_gemm_tn):
 * vec_out[i] = sum_s vec[s][i]  (the bias gradient; stored, not added). */
int nrm_slab_reduce(const float* ws, int nsplit, int nj, int ldws, int ni, float* out, long out_istride, long out_jstride,
                    float* out2, long out2_istride, long out2_jstride, float sign2,
                    const float* vec, float* vec_out, nrm_stream_t stream);
/* n slab sets in ONE launch (entries as the arguments of nrm_slab_reduce): the 14 reductions of a training step are small,
 * latency-bound launches that depend on nothing but their own slabs, so a caller may defer them and flush them together
 * before the optimizer reads the gradients (reference: autograd's per-parameter accumulation behind train.py:73). */
typedef struct nrm_slab_desc {
    const float* ws; int nsplit, nj, ldws, ni;
    float* out; long out_istride, out_jstride;
    float* out2; long out2_istride, out2_jstride; float sign2;
    const float* vec; float* vec_out;
} nrm_slab_desc;
int nrm_slab_reduce_multi(const nrm_slab_desc* descs, int n, nrm_stream_t stream);

/* ---- BatchNorm1d over rows (reference models/user_model.py:18,32), N % 4 == 0, ld % 4 == 0.
 * nrm_colreduce mode 0: s0 += sum_r x;  1: s0 += sum_r (x-mean)^2;  2: s0 += sum_r dy, s1 += sum_r dy*(x-mean)*rstd */
int nrm_colreduce(int mode, const float* x, const float* dy, const float* mean, const float* rstd,
                  float* s0, float* s1, int R, int N, int ld, nrm_stream_t stream);
/* between the reductions (train mode): stage 0: out = mean = s/R, running = (1-momentum) running + momentum mean;
 * stage 1: var = s/R, out = rstd = 1/sqrt(var + eps), running = (1-momentum) running + momentum var R/(R-1).
 * running may be NULL. */
int nrm_bn_finalize(int stage, const float* s, float* out, float* running, int R, int N, float momentum, float eps,
                    nrm_stream_t stream);
/* backward of the gate product y = g * e (models/user_model.py:33): dg = dy * e, de = dy * g; [R, N], N % 4 == 0 */
int nrm_mul_bwd(const float* dy, int lddy, const float* g, int ldg, const float* e, int lde, float* dg, float* de, int ldo,
                int R, int N, nrm_stream_t stream);
int nrm_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 float* y, int R, int N, int ld, nrm_stream_t stream);
/* training != 0: dx = gamma*rstd*(dy - s0/R - xhat*s1/R) with s0,s1 from nrm_colreduce mode 2; else gamma*rstd*dy.
 * add (optional, same [R, ld] layout) is added to dx: the gradient the rows receive from their second consumer, the gate
 * product of user_model.py:33, joins here instead of in a separate elementwise pass. */
int nrm_bn_backward(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                    const float* s0, const float* s1, const float* add, float* dx, int R, int N, int ld, int training,
                    nrm_stream_t stream);
/* out[R, ldo] = cat(parts, dim=1) in one launch (reference models/user_model.py:31 cat[eu_H, eu_L, ec],
 * user_invariant_interest_model.py:81,88): n <= 8 row-major sources srcs[i] [R, widths[i]] with leading dimension lds[i]
 * (HOST arrays, copied into the kernel arguments) */
int nrm_concat_cols(const float* const* srcs, const long* lds, const int* widths, int n, float* out, int ldo, long R,
                    nrm_stream_t stream);

/* ---- weighted pool (reference models/user_invariant_interest_model.py:86-87, no softmax, no mask)
 * out[b,i,:] (+)= sum_j W[b,i,j] * X[b,j,:]   W element (b,i,j) at W[b*wsb + i*wsi + j*wsj]; X [B,J,D] with row stride ldx
 * (>= D: a column block of a wider matrix, e.g. the head gradient's pooled slice); out [B,I,D] contiguous.
 * forward: W = scores [B,T,H], X = history;  d history: W = scores^T (wsi=1, wsj=H), X = d pooled, accumulate = 1 adds onto
 * the history gradient the attention backward has already written */
int nrm_pool_bmm(const float* W, long wsb, long wsi, long wsj, const float* X, int ldx, float* out,
                 int B, int I, int J, int D, int accumulate, nrm_stream_t stream);
/* ds[b,t,h] = sum_d g[b,t,d] * h[b,h,d]   (gradient of the pool w.r.t. the scores); g [B,T,D] with row stride ldg.
 * zero_out / zero_n (optional): zero_n floats at zero_out are cleared by this launch -- the dw2 | db2 accumulators of the
 * nrm_pwattn_bwd_dz call that follows on the same stream (one launch less per attention) */
int nrm_pool_rowdot(const float* g, int ldg, const float* h, float* ds, int B, int T, int H, int D, float* zero_out, int zero_n,
                    nrm_stream_t stream);

/* ---- instant-interest layer (reference models/user_instant_interest_model.py: ReLU(Linear(3 -> 8)) on the popularity scalars
 * of a candidate).  x [R, K] dense, fp32 or fp64; weight [N, K]; K <= 4, N <= 8.
 * forward: y [R, ldy] = ReLU(x W^T + b), columns N..ldy-1 written as 0.
 * backward: dwb[n*K + k] += sum_r g[r,n] x[r,k] and dwb[N*K + n] += sum_r g[r,n] with g = dy (row stride lddy) where the
 * pre-activation is positive (recomputed from x); dwb (N*K + N floats) must be initialised; x has no gradient. */
int nrm_small_linear_relu_fwd(const void* x, int x_is_f64, const float* weight, const float* bias, float* y, long R, int K, int N, int ldy,
                              nrm_stream_t stream);
int nrm_small_linear_relu_bwd(const void* x, int x_is_f64, const float* weight, const float* bias, const float* dy, int lddy, long R,
                              int K, int N, float* dwb, nrm_stream_t stream);

/* ---- loss (reference models/user_model.py:37-43): (1-alpha)*BCE(softmax_T(out), y) + alpha*BCE(softmax_T(out +
 * delta[id]), y), mean over B*T, log clamped at -100.  Writes loss_sum[0] += loss, dout [B,T] = dL/dout and
 * ddelta[id[b]] += dL/ddelta (loss_sum and ddelta must be zero-initialised).  Any T >= 1 (a lane
 * keeps its candidates in registers up to T = 256 and re-reads the row beyond).  delta has n_delta entries;
 * a negative id counts from the end as in torch indexing, an id still outside [0, n_delta) is clamped and sets
 * err[0] = 1 (the reference raises IndexError at user_model.py:40; never an out-of-bounds access here).
 * out_stride / dout_stride: floats between consecutive (b, t) entries -- 1 for dense [B,T]; 4 for the single column of a
 * zero-padded [B*T, 4] matrix (the layout the logit GEMM writes and the GEMM consuming dout reads; dout then gets (g,0,0,0)).
 * label: [B,T] dense, fp32 or (label_is_f64) fp64 as the reference's DataLoader yields it. */
int nrm_loss_fwd_bwd(const float* out, int out_stride, const void* label, int label_is_f64, const long* user_id, const float* delta,
                     long n_delta, float alpha, int B, int T, float* loss_sum, float* dout, int dout_stride, float* ddelta, int* err,
                     nrm_stream_t stream);

/* ---- Adam over one flat fp32 buffer (reference train.py:48,73-75): g += wd*p; m,v update; bias correction for
 * `step` (1-based); p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps); optionally zeroes g (optimizer.zero_grad()).
 * n % 4 == 0, 16-byte aligned buffers. */
int nrm_adam_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, int zero_grad, nrm_stream_t stream);
/* n gradient tensors -> their slots of the flat gradient buffer in ONE launch (reference train.py:73-74: what sits between
 * loss.backward() and optimizer.step(); with data parallelism, what the single all-reduce then runs on).  srcs / offsets /
 * counts are HOST arrays (copied into the kernel arguments at call time): flat[offsets[i] .. +counts[i]) = srcs[i][0 ..
 * counts[i]); a NULL source zero-fills its slot.  Contiguous fp32 sources. */
int nrm_gather_flat(const float* const* srcs, const long* offsets, const long* counts, int n, float* flat, long flat_n,
                    nrm_stream_t stream);
/* the same with the step counter on the device (state = float[4]: {step, 1-beta1^step, sqrt(1-beta2^step), -},
 * zero-initialised by the caller, advanced by the call): safe to capture in a hipGraph and replay */
int nrm_adam_step_dev(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float* state, int zero_grad, nrm_stream_t stream);

/* ---- embedding front end: reference models/user_invariant_interest_model.py:50-71,74-79 (slice_x,
 * feature_embedding, time_embedding).  x: [nrows, xcols] packed rows (tool/process_data.py:198-240), fp64
 * (x_is_f64 != 0) or fp32: [year,month,day,hour | text_img P | category | sub-category x n_sub | sentiment x3 |
 * type | (behaviour != 0: read_time, scroll)].  Writes the label rows lab[nrows, ldlab] = [Emb_cat(cat)+mean
 * Emb_cat(sub) e0 | ReLU(W_s s + b_s) e1 | Emb_type e2 | sum of 4 time embeddings e3 | (read_time, scroll)] zero
 * padded to ldlab, and the text/image block ti[nrows, ldti] as fp32.  err[0] is set to 1 on an out-of-range index
 * (the index is clamped; the reference raises IndexError). */
int nrm_frontend_fwd(const void* x, int x_is_f64, int nrows, int xcols, int P, int n_sub, int behaviour,
                     const float* cat_tab, int n_cat, int e0, const float* sen_w, const float* sen_b, int e1,
                     const float* type_tab, int n_type, int e2,
                     const float* year_tab, const float* month_tab, const float* day_tab, const float* hour_tab,
                     int n_year, int n_month, int n_day, int n_hour, int e3,
                     float* lab, int ldlab, float* ti, int ldti, int* err, nrm_stream_t stream);
/* backward of the label rows: accumulates (+=, float atomics) the table / sentiment-layer gradients.  d_cat_tab may be NULL:
 * the category-table gradient is then left to nrm_frontend_cat_grad */
int nrm_frontend_bwd(const void* x, int x_is_f64, int nrows, int xcols, int P, int n_sub, int behaviour,
                     const float* dlab, int lddl, const float* sen_w, const float* sen_b,
                     int n_cat, int e0, int e1, int n_type, int e2, int n_year, int n_month, int n_day, int n_hour, int e3,
                     float* d_cat_tab, float* d_sen_w, float* d_sen_b, float* d_type_tab,
                     float* d_year_tab, float* d_month_tab, float* d_day_tab, float* d_hour_tab, nrm_stream_t stream);

/* category-table gradient of up to two row sets (history rows, candidate rows; nrows = 0 skips one) by a counting sort of the
 * (row, slot) references by category id and a gather-sum over runs of equal ids -- instead of (1 + n_sub) * e0 float atomics per
 * row.  d_cat_tab [n_cat, e0] is accumulated into (+=).  ws: nrm_frontend_cat_ws_ints(n_cat, nrows0 + nrows1, n_sub) int32 of
 * scratch.  Both row sets have the same element type (x_is_f64).  Summation order inside an id varies from run to run. */
long nrm_frontend_cat_ws_ints(int n_cat, long nrows_total, int n_sub);
int nrm_frontend_cat_grad(const void* x0, int nrows0, int xcols0, const float* dlab0, int lddl0,
                          const void* x1, int nrows1, int xcols1, const float* dlab1, int lddl1, int x_is_f64,
                          int P, int n_sub, int n_cat, int e0, float* d_cat_tab, int* ws, nrm_stream_t stream);

/* ---- per-impression ROC-AUC and top-1 hit (reference train.py:77-80, verify.py:25-36, tool/evaluation.py:3-5;
 * sklearn roc_auc_score for binary labels = Mann-Whitney U with ties counted 1/2).  score, label [B,T]; len [B]
 * (int32, candidates that count per row, NULL = T).  auc[b] = -1 where a row holds a single class. */
int nrm_row_auc(const float* score, const float* label, const int* len, int B, int T, float* auc, int* top1,
                nrm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NRM_HOTPATH_H */
