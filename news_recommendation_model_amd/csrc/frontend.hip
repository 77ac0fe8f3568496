// Front end of UserInvariantInterestModel: packed feature rows -> embedded label rows + fp32 text/image rows.
// Reference: models/user_invariant_interest_model.py:50-71,74-79 (slice_x, feature_embedding, time_embedding).
//
// Packed row (tool/process_data.py:198-240):  [year, month, day, hour | text_img P | category | sub-category x NS |
//   sentiment x 3 | type | (history only: read_time, scroll)], fp64 out of the reference DataLoader or fp32.
// Label row written here:  [ Emb_cat(category) + mean_NS Emb_cat(sub) : e0 | ReLU(W_s sentiment + b_s) : e1 |
//   Emb_type(type) : e2 | Emb_year + Emb_month + Emb_day + Emb_hour : e3 | (history: read_time, scroll) ], zero padded
//   to a leading dimension that is a multiple of 4; the text/image block is copied to a dense fp32 [rows, P] matrix.
// The category table serves category AND sub-categories; the mean includes padding id 0 (as the reference does).
//
// Backward scatters d(label row) into the tables.  The category table (3000 rows) takes global float atomics (one
// contiguous run of e0 columns per row id); the tiny hot tables (type 16 rows, year/month/day/hour, sentiment W/b)
// would serialise on a handful of addresses, so each workgroup first sums its rows in LDS and flushes once.
#include "common.hpp"
#include "frontend.hpp"

namespace nrm {

template <typename XT>
__device__ __forceinline__ int row_index(const XT* xr, int col, int limit, int* err) {
    const int i = (int)xr[col];
    if (i < 0 || i >= limit) { *err = 1; return i < 0 ? 0 : limit - 1; }   // reference: IndexError; here: flag + clamp
    return i;
}

// one workgroup (128 threads) per row
template <typename XT>
__global__ __launch_bounds__(128) void frontend_fwd_kernel(const FrontendParams p, const XT* __restrict__ x) {
    const int row = blockIdx.x;
    const XT* xr = x + (size_t)row * p.xcols;
    const int P = p.P, NS = p.n_sub;
    const int c_cat = 4 + P, c_sub = c_cat + 1, c_sen = c_sub + NS, c_typ = c_sen + 3, c_beh = c_typ + 1;
    const int e0 = p.e0, e1 = p.e1, e2 = p.e2, e3 = p.e3;
    const int width = e0 + e1 + e2 + e3 + (p.behaviour ? 2 : 0);
    float* lab = p.lab + (size_t)row * p.ldlab;
    float* ti = p.ti + (size_t)row * p.ldti;

    // wave-uniform row header (every thread reads the same few scalars: L1 broadcast)
    const int iy = row_index(xr, 0, p.n_year, p.err), im = row_index(xr, 1, p.n_month, p.err);
    const int id = row_index(xr, 2, p.n_day, p.err), ih = row_index(xr, 3, p.n_hour, p.err);
    const int icat = row_index(xr, c_cat, p.n_cat, p.err), ityp = row_index(xr, c_typ, p.n_type, p.err);
    const float s0 = (float)xr[c_sen], s1 = (float)xr[c_sen + 1], s2 = (float)xr[c_sen + 2];
    const float inv_ns = 1.0f / (float)NS;

    for (int c = threadIdx.x; c < p.ldlab; c += 128) {
        float v = 0.f;
        if (c < e0) {
            float sub = 0.f;
            for (int k = 0; k < NS; ++k) sub += p.cat_tab[(size_t)row_index(xr, c_sub + k, p.n_cat, p.err) * e0 + c];
            v = p.cat_tab[(size_t)icat * e0 + c] + sub * inv_ns;
        } else if (c < e0 + e1) {
            const int j = c - e0;
            const float pre = p.sen_b[j] + p.sen_w[j * 3] * s0 + p.sen_w[j * 3 + 1] * s1 + p.sen_w[j * 3 + 2] * s2;
            v = fmaxf(pre, 0.f);
        } else if (c < e0 + e1 + e2) {
            v = p.type_tab[(size_t)ityp * e2 + (c - e0 - e1)];
        } else if (c < e0 + e1 + e2 + e3) {
            const int k = c - e0 - e1 - e2;
            v = p.year_tab[(size_t)iy * e3 + k] + p.month_tab[(size_t)im * e3 + k] + p.day_tab[(size_t)id * e3 + k] + p.hour_tab[(size_t)ih * e3 + k];
        } else if (c < width) {
            v = (float)xr[c_beh + (c - (e0 + e1 + e2 + e3))];
        }
        lab[c] = v;
    }
    for (int c = threadIdx.x; c < p.ldti; c += 128) ti[c] = c < P ? (float)xr[4 + c] : 0.f;
}

// one workgroup (128 threads) per FE_ROWS consecutive rows; LDS accumulators for the small tables
constexpr int FE_ROWS = 32;

template <typename XT>
__global__ __launch_bounds__(128) void frontend_bwd_kernel(const FrontendParams p, const XT* __restrict__ x,
                                                          const float* __restrict__ dlab, int lddl, int nrows) {
    extern __shared__ float sm[];
    const int e0 = p.e0, e1 = p.e1, e2 = p.e2, e3 = p.e3;
    float* a_type = sm;                                  // [n_type][e2]
    float* a_year = a_type + p.n_type * e2;              // [n_year][e3]
    float* a_month = a_year + p.n_year * e3;
    float* a_day = a_month + p.n_month * e3;
    float* a_hour = a_day + p.n_day * e3;
    float* a_sen = a_hour + p.n_hour * e3;               // [e1][4] = dW (3) | db
    const int total = (int)(a_sen + e1 * 4 - sm);
    for (int i = threadIdx.x; i < total; i += 128) sm[i] = 0.f;
    __syncthreads();

    const int P = p.P, NS = p.n_sub;
    const int c_cat = 4 + P, c_sub = c_cat + 1, c_sen = c_sub + NS, c_typ = c_sen + 3;
    const float inv_ns = 1.0f / (float)NS;
    const int r_lo = blockIdx.x * FE_ROWS, r_hi = min(nrows, r_lo + FE_ROWS);
    int dummy = 0;
    for (int row = r_lo; row < r_hi; ++row) {
        const XT* xr = x + (size_t)row * p.xcols;
        const float* g = dlab + (size_t)row * lddl;
        const int iy = row_index(xr, 0, p.n_year, &dummy), im = row_index(xr, 1, p.n_month, &dummy);
        const int id = row_index(xr, 2, p.n_day, &dummy), ih = row_index(xr, 3, p.n_hour, &dummy);
        const int icat = row_index(xr, c_cat, p.n_cat, &dummy), ityp = row_index(xr, c_typ, p.n_type, &dummy);
        const float s0 = (float)xr[c_sen], s1 = (float)xr[c_sen + 1], s2 = (float)xr[c_sen + 2];
        for (int c = threadIdx.x; c < e0 + e1 + e2 + e3; c += 128) {
            const float gv = g[c];
            if (c < e0) {
                atomicAdd(p.d_cat_tab + (size_t)icat * e0 + c, gv);
                const float gs = gv * inv_ns;
                for (int k = 0; k < NS; ++k)
                    atomicAdd(p.d_cat_tab + (size_t)row_index(xr, c_sub + k, p.n_cat, &dummy) * e0 + c, gs);
            } else if (c < e0 + e1) {
                const int j = c - e0;
                const float pre = p.sen_b[j] + p.sen_w[j * 3] * s0 + p.sen_w[j * 3 + 1] * s1 + p.sen_w[j * 3 + 2] * s2;
                const float gz = pre > 0.f ? gv : 0.f;           // ReLU'
                a_sen[j * 4 + 0] += gz * s0;                     // column j is owned by this thread: plain LDS RMW
                a_sen[j * 4 + 1] += gz * s1;
                a_sen[j * 4 + 2] += gz * s2;
                a_sen[j * 4 + 3] += gz;
            } else if (c < e0 + e1 + e2) {
                a_type[ityp * e2 + (c - e0 - e1)] += gv;         // one thread per column, rows sequential
            } else {
                const int k = c - e0 - e1 - e2;
                a_year[iy * e3 + k] += gv;
                a_month[im * e3 + k] += gv;
                a_day[id * e3 + k] += gv;
                a_hour[ih * e3 + k] += gv;
            }
        }
    }
    __syncthreads();
    // flush: one float atomic per touched LDS cell (zeros are skipped)
    for (int i = threadIdx.x; i < p.n_type * e2; i += 128) if (a_type[i] != 0.f) atomicAdd(p.d_type_tab + i, a_type[i]);
    for (int i = threadIdx.x; i < p.n_year * e3; i += 128) if (a_year[i] != 0.f) atomicAdd(p.d_year_tab + i, a_year[i]);
    for (int i = threadIdx.x; i < p.n_month * e3; i += 128) if (a_month[i] != 0.f) atomicAdd(p.d_month_tab + i, a_month[i]);
    for (int i = threadIdx.x; i < p.n_day * e3; i += 128) if (a_day[i] != 0.f) atomicAdd(p.d_day_tab + i, a_day[i]);
    for (int i = threadIdx.x; i < p.n_hour * e3; i += 128) if (a_hour[i] != 0.f) atomicAdd(p.d_hour_tab + i, a_hour[i]);
    for (int i = threadIdx.x; i < e1 * 4; i += 128) {
        const float v = a_sen[i];
        if (v != 0.f) {
            const int j = i >> 2, w = i & 3;
            if (w < 3) atomicAdd(p.d_sen_w + j * 3 + w, v); else atomicAdd(p.d_sen_b + j, v);
        }
    }
}

static size_t bwd_lds_bytes(const FrontendParams& p) {
    return sizeof(float) * ((size_t)p.n_type * p.e2 + (size_t)(p.n_year + p.n_month + p.n_day + p.n_hour) * p.e3 + (size_t)p.e1 * 4);
}

hipError_t frontend_fwd_launch(const FrontendParams& p, const void* x, int x_is_f64, int nrows, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    if (x_is_f64) hipLaunchKernelGGL(frontend_fwd_kernel<double>, dim3(nrows), dim3(128), 0, st, p, (const double*)x);
    else          hipLaunchKernelGGL(frontend_fwd_kernel<float>, dim3(nrows), dim3(128), 0, st, p, (const float*)x);
    return hipGetLastError();
}

hipError_t frontend_bwd_launch(const FrontendParams& p, const void* x, int x_is_f64, const float* dlab, int lddl,
                               int nrows, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    const size_t shm = bwd_lds_bytes(p);
    if (shm > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((nrows + FE_ROWS - 1) / FE_ROWS);
    if (x_is_f64) {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<double>, grid, dim3(128), shm, st, p, (const double*)x, dlab, lddl, nrows);
    } else {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<float>, grid, dim3(128), shm, st, p, (const float*)x, dlab, lddl, nrows);
    }
    return hipGetLastError();
}

}  // namespace nrm
