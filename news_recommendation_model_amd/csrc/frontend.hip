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

// Row headers of a block of consecutive rows, decoded once into LDS: the table indices and the sentiment scalars.  Every
// later loop is then a flat walk over (row, column) pairs with no dependent global load in front of its table access -- the
// first version decoded a row's header with every thread and walked rows one after the other (one workgroup per row in the
// forward, 32 serial rows in the backward): latency-bound, 62 + 190 us at the reference's default sizes for 43 MB.
constexpr int FE_MAXSUB = 16;                      // sub-category slots per row (reference: 5)
constexpr int FE_HDR = 6 + FE_MAXSUB;              // year, month, day, hour, category, type, sub-categories
struct RowHeaders {
    int* idx;                                      // [rows][FE_HDR]
    float* sen;                                    // [rows][4]
};

template <typename XT>
__device__ __forceinline__ void stage_headers(const FrontendParams& p, const XT* __restrict__ x, int r_lo, int nr, const RowHeaders& hd,
                                              int nthreads) {
    const int P = p.P, NS = p.n_sub;
    const int c_cat = 4 + P, c_sub = c_cat + 1, c_sen = c_sub + NS, c_typ = c_sen + 3;
    const int items = 9 + NS;                      // 6 + NS indices, 3 scalars
    for (int i = threadIdx.x; i < nr * items; i += nthreads) {
        const int r = i / items, k = i - r * items;
        const XT* xr = x + (size_t)(r_lo + r) * p.xcols;
        if (k < 4) {
            const int lim = k == 0 ? p.n_year : k == 1 ? p.n_month : k == 2 ? p.n_day : p.n_hour;
            hd.idx[r * FE_HDR + k] = row_index(xr, k, lim, p.err);
        } else if (k == 4) {
            hd.idx[r * FE_HDR + 4] = row_index(xr, c_cat, p.n_cat, p.err);
        } else if (k == 5) {
            hd.idx[r * FE_HDR + 5] = row_index(xr, c_typ, p.n_type, p.err);
        } else if (k < 6 + NS) {
            hd.idx[r * FE_HDR + k] = row_index(xr, c_sub + (k - 6), p.n_cat, p.err);
        } else {
            hd.sen[r * 4 + (k - 6 - NS)] = (float)xr[c_sen + (k - 6 - NS)];
        }
    }
}

// FWD_ROWS consecutive rows per workgroup of 256 threads; the outputs of consecutive rows are contiguous, so the flat
// (row, column) walk stores whole cache lines
constexpr int FWD_ROWS = 8;
constexpr int FE_THREADS = 256;

template <typename XT>
__global__ __launch_bounds__(FE_THREADS) void frontend_fwd_kernel(const FrontendParams p, const XT* __restrict__ x, int nrows) {
    __shared__ int h_idx[FWD_ROWS * FE_HDR];
    __shared__ float h_sen[FWD_ROWS * 4];
    const int r_lo = blockIdx.x * FWD_ROWS, nr = min(FWD_ROWS, nrows - r_lo);
    const RowHeaders hd = {h_idx, h_sen};
    stage_headers(p, x, r_lo, nr, hd, FE_THREADS);
    __syncthreads();
    const int P = p.P, NS = p.n_sub;
    const int c_beh = 4 + P + 1 + NS + 3 + 1;
    const int e0 = p.e0, e1 = p.e1, e2 = p.e2, e3 = p.e3;
    const int width = e0 + e1 + e2 + e3 + (p.behaviour ? 2 : 0);
    const float inv_ns = 1.0f / (float)NS;
    float* lab = p.lab + (size_t)r_lo * p.ldlab;
    for (int i = threadIdx.x; i < nr * p.ldlab; i += FE_THREADS) {
        const int r = i / p.ldlab, c = i - r * p.ldlab;
        const int* hi = h_idx + r * FE_HDR;
        float v = 0.f;
        if (c < e0) {
            float sub = 0.f;
            for (int k = 0; k < NS; ++k) sub += p.cat_tab[(size_t)hi[6 + k] * e0 + c];
            v = p.cat_tab[(size_t)hi[4] * e0 + c] + sub * inv_ns;
        } else if (c < e0 + e1) {
            const int j = c - e0;
            const float pre = p.sen_b[j] + p.sen_w[j * 3] * h_sen[r * 4] + p.sen_w[j * 3 + 1] * h_sen[r * 4 + 1] + p.sen_w[j * 3 + 2] * h_sen[r * 4 + 2];
            v = fmaxf(pre, 0.f);
        } else if (c < e0 + e1 + e2) {
            v = p.type_tab[(size_t)hi[5] * e2 + (c - e0 - e1)];
        } else if (c < e0 + e1 + e2 + e3) {
            const int k = c - e0 - e1 - e2;
            v = p.year_tab[(size_t)hi[0] * e3 + k] + p.month_tab[(size_t)hi[1] * e3 + k] + p.day_tab[(size_t)hi[2] * e3 + k] + p.hour_tab[(size_t)hi[3] * e3 + k];
        } else if (c < width) {
            v = (float)x[(size_t)(r_lo + r) * p.xcols + c_beh + (c - (e0 + e1 + e2 + e3))];
        }
        lab[i] = v;
    }
    float* ti = p.ti + (size_t)r_lo * p.ldti;
    for (int i = threadIdx.x; i < nr * p.ldti; i += FE_THREADS) {
        const int r = i / p.ldti, c = i - r * p.ldti;
        ti[i] = c < P ? (float)x[(size_t)(r_lo + r) * p.xcols + 4 + c] : 0.f;
    }
}

// one workgroup (256 threads) per FE_ROWS consecutive rows; LDS accumulators for the small tables
constexpr int FE_ROWS = 32;

template <typename XT>
__global__ __launch_bounds__(FE_THREADS) void frontend_bwd_kernel(const FrontendParams p, const XT* __restrict__ x,
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
    const RowHeaders hd = {reinterpret_cast<int*>(sm + total), sm + total + FE_ROWS * FE_HDR};
    for (int i = threadIdx.x; i < total; i += FE_THREADS) sm[i] = 0.f;
    const int r_lo = blockIdx.x * FE_ROWS, nr = min(FE_ROWS, nrows - r_lo);
    FrontendParams pq = p;
    int dummy = 0;
    pq.err = &dummy;                                     // (the forward has already flagged out-of-range ids of these rows)
    stage_headers(pq, x, r_lo, nr, hd, FE_THREADS);
    __syncthreads();

    const int NS = p.n_sub;
    const float inv_ns = 1.0f / (float)NS;
    const float* g0 = dlab + (size_t)r_lo * lddl;
    // (1) category table: a flat walk over (row, column < e0): 1 + NS float atomics each, nothing serial between rows
    for (int i = threadIdx.x; i < nr * e0; i += FE_THREADS) {
        const int r = i / e0, c = i - r * e0;
        const int* hi = hd.idx + r * FE_HDR;
        const float gv = g0[(size_t)r * lddl + c];
        atomicAdd(p.d_cat_tab + (size_t)hi[4] * e0 + c, gv);
        const float gs = gv * inv_ns;
        for (int k = 0; k < NS; ++k) atomicAdd(p.d_cat_tab + (size_t)hi[6 + k] * e0 + c, gs);
    }
    // (2) the small hot tables: one thread per column, rows in sequence (plain LDS read-modify-write: the column has one owner)
    for (int c = e0 + threadIdx.x; c < e0 + e1 + e2 + e3; c += FE_THREADS) {
    if (c < e0 + e1) {
        const int j = c - e0;
        const float wb = p.sen_b[j], w0 = p.sen_w[j * 3], w1 = p.sen_w[j * 3 + 1], w2 = p.sen_w[j * 3 + 2];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, db = 0.f;
#pragma unroll 8
        for (int r = 0; r < nr; ++r) {
            const float gv = g0[(size_t)r * lddl + c];
            const float s0 = hd.sen[r * 4], s1 = hd.sen[r * 4 + 1], s2 = hd.sen[r * 4 + 2];
            const float gz = wb + w0 * s0 + w1 * s1 + w2 * s2 > 0.f ? gv : 0.f;      // ReLU'
            d0 += gz * s0; d1 += gz * s1; d2 += gz * s2; db += gz;
        }
        a_sen[j * 4 + 0] = d0; a_sen[j * 4 + 1] = d1; a_sen[j * 4 + 2] = d2; a_sen[j * 4 + 3] = db;
    } else if (c < e0 + e1 + e2) {
#pragma unroll 8
        for (int r = 0; r < nr; ++r) a_type[hd.idx[r * FE_HDR + 5] * e2 + (c - e0 - e1)] += g0[(size_t)r * lddl + c];
    } else if (c < e0 + e1 + e2 + e3) {
        const int k = c - e0 - e1 - e2;
#pragma unroll 8
        for (int r = 0; r < nr; ++r) {
            const float gv = g0[(size_t)r * lddl + c];
            const int* hi = hd.idx + r * FE_HDR;
            a_year[hi[0] * e3 + k] += gv;
            a_month[hi[1] * e3 + k] += gv;
            a_day[hi[2] * e3 + k] += gv;
            a_hour[hi[3] * e3 + k] += gv;
        }
    }
    }
    __syncthreads();
    // flush: one float atomic per touched LDS cell (zeros are skipped)
    for (int i = threadIdx.x; i < p.n_type * e2; i += FE_THREADS) if (a_type[i] != 0.f) atomicAdd(p.d_type_tab + i, a_type[i]);
    for (int i = threadIdx.x; i < p.n_year * e3; i += FE_THREADS) if (a_year[i] != 0.f) atomicAdd(p.d_year_tab + i, a_year[i]);
    for (int i = threadIdx.x; i < p.n_month * e3; i += FE_THREADS) if (a_month[i] != 0.f) atomicAdd(p.d_month_tab + i, a_month[i]);
    for (int i = threadIdx.x; i < p.n_day * e3; i += FE_THREADS) if (a_day[i] != 0.f) atomicAdd(p.d_day_tab + i, a_day[i]);
    for (int i = threadIdx.x; i < p.n_hour * e3; i += FE_THREADS) if (a_hour[i] != 0.f) atomicAdd(p.d_hour_tab + i, a_hour[i]);
    for (int i = threadIdx.x; i < e1 * 4; i += FE_THREADS) {
        const float v = a_sen[i];
        if (v != 0.f) {
            const int j = i >> 2, w = i & 3;
            if (w < 3) atomicAdd(p.d_sen_w + j * 3 + w, v); else atomicAdd(p.d_sen_b + j, v);
        }
    }
}

static size_t bwd_lds_bytes(const FrontendParams& p) {
    return sizeof(float) * ((size_t)p.n_type * p.e2 + (size_t)(p.n_year + p.n_month + p.n_day + p.n_hour) * p.e3 + (size_t)p.e1 * 4 +
                            (size_t)FE_ROWS * (FE_HDR + 4));
}

hipError_t frontend_fwd_launch(const FrontendParams& p, const void* x, int x_is_f64, int nrows, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    if (p.n_sub > FE_MAXSUB) return hipErrorInvalidValue;
    const dim3 grid((nrows + FWD_ROWS - 1) / FWD_ROWS);
    if (x_is_f64) hipLaunchKernelGGL(frontend_fwd_kernel<double>, grid, dim3(FE_THREADS), 0, st, p, (const double*)x, nrows);
    else          hipLaunchKernelGGL(frontend_fwd_kernel<float>, grid, dim3(FE_THREADS), 0, st, p, (const float*)x, nrows);
    return hipGetLastError();
}

hipError_t frontend_bwd_launch(const FrontendParams& p, const void* x, int x_is_f64, const float* dlab, int lddl,
                               int nrows, hipStream_t st) {
    if (nrows <= 0) return hipSuccess;
    const size_t shm = bwd_lds_bytes(p);
    if (shm > 160 * 1024 || p.n_sub > FE_MAXSUB) return hipErrorInvalidValue;
    const dim3 grid((nrows + FE_ROWS - 1) / FE_ROWS);
    if (x_is_f64) {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<double>, grid, dim3(FE_THREADS), shm, st, p, (const double*)x, dlab, lddl, nrows);
    } else {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<float>, grid, dim3(FE_THREADS), shm, st, p, (const float*)x, dlab, lddl, nrows);
    }
    return hipGetLastError();
}

}  // namespace nrm
