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

#define NRM_ABI_VERSION 1
#define NRM_OK 0
#define NRM_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define NRM_ELAUNCH (-2)  /* HIP launch error */

typedef void* nrm_stream_t;

int nrm_abi_version(void);
const char* nrm_last_error(void);

/* ---- pointwise history attention: reference models/attention_model.py:52-97
 *      (PointwiseAttentionExpanded.forward), score[b,t,h] = fc2(GELU(fc1(cat[h,t,t-h,t*h]))).
 * The caller supplies the two side projections of the split fc1 = [W_h | W_t | W_d | W_p]:
 *      u[b,h,:] = h[b,h,:] (W_h - W_d)^T + fc1.bias        v[b,t,:] = t[b,t,:] (W_t + W_d)^T          */

/* floats to allocate for the packed copy of W_p used by nrm_pwattn_fwd */
long nrm_pwattn_packed_floats(int D);
/* fc1_weight: [D, 4D] with row stride ld; packs W_p = fc1_weight[:, 3D:4D] */
int nrm_pwattn_pack_wp(const float* fc1_weight, int ld, int D, float* packed, nrm_stream_t stream);
/* t [B,T,D], h [B,H,D], u [B,H,D], v [B,T,D] contiguous; w2 = fc2.weight [D]; b2 = fc2.bias [1];
 * z [B,T,H,D] pre-activation saved for backward (NULL in inference); s [B,T,H] scores */
int nrm_pwattn_fwd(const float* t, const float* h, const float* u, const float* v, const float* packed_wp,
                   const float* w2, const float* b2, float* z, float* s,
                   int B, int T, int H, int D, nrm_stream_t stream);

/* backward, step 1 (autograd of attention_model.py:29-32 through GELU and fc2):
 *   z <- dz = ds * w2 * gelu'(z) in place;  dw2[k] += sum ds*gelu(z)   (dw2 must be initialised) */
int nrm_pwattn_bwd_dz(float* z_inout, const float* ds, const float* w2, float* dw2,
                      long M, int D, nrm_stream_t stream);
/* number of [D,D] partial slabs nrm_pwattn_bwd_contract writes into `ws` */
int nrm_pwattn_bwd_nsplit(int B, int T, int H, int D);
/* backward, step 2 (the bilinear term): given dz [B,T,H,D], t, h and W_p (row stride ldwp)
 *   dt[b,t,d] += sum_{h,k} dz W_p[k,d] h[b,h,d]      dh[b,h,d] += sum_{t,k} dz W_p[k,d] t[b,t,d]
 *   ws[i][d][k] (i < nsplit) = partial of dW_p[k,d] = sum_{b,t,h} dz[b,t,h,k] t[b,t,d] h[b,h,d]
 *                              (TRANSPOSED slabs: sum them over i, then transpose)
 * dt/dh are accumulated into (float atomics), ws is overwritten.
 * passes: bit 0 = the (b,t)-grouped launch (dt, ws), bit 1 = the (b,h)-grouped launch (dh); 3 = both. */
int nrm_pwattn_bwd_contract(const float* dz, const float* t, const float* h, const float* wp, int ldwp,
                            float* dt, float* dh, float* ws,
                            int B, int T, int H, int D, int passes, nrm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NRM_HOTPATH_H */
