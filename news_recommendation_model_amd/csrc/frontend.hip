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
#include <algorithm>
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

// A workgroup (256 threads) walks groups of FE_ROWS consecutive rows (grid-stride) and keeps ONE set of LDS accumulators for
// the small tables over all of them: the flush at the end is one float atomic per touched cell and WORKGROUP, all workgroups
// hitting the same few thousand addresses.  At most FE_MAX_BLOCKS workgroups.
constexpr int FE_ROWS = 32;
constexpr int FE_MAX_BLOCKS = 512;

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
    FrontendParams pq = p;
    int dummy = 0;
    pq.err = &dummy;                                     // (the forward has already flagged out-of-range ids of these rows)
    const int NS = p.n_sub;
    const float inv_ns = 1.0f / (float)NS;
    const int small = e1 + e2 + e3;
    const int ngroups = (nrows + FE_ROWS - 1) / FE_ROWS;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int r_lo = grp * FE_ROWS, nr = min(FE_ROWS, nrows - r_lo);
        __syncthreads();                                 // the previous group's readers of the headers are done (and the zeroing)
        stage_headers(pq, x, r_lo, nr, hd, FE_THREADS);
        __syncthreads();
        const float* g0 = dlab + (size_t)r_lo * lddl;
        // (1) category table: a flat walk over (row, column < e0): 1 + NS float atomics each, nothing serial between rows
        //     (d_cat_tab == nullptr: the caller forms that gradient with cat_grad_launch below instead)
        for (int i = threadIdx.x; i < (p.d_cat_tab ? nr * e0 : 0); i += FE_THREADS) {
            const int r = i / e0, c = i - r * e0;
            const int* hi = hd.idx + r * FE_HDR;
            const float gv = g0[(size_t)r * lddl + c];
            atomicAdd(p.d_cat_tab + (size_t)hi[4] * e0 + c, gv);
            const float gs = gv * inv_ns;
            for (int k = 0; k < NS; ++k) atomicAdd(p.d_cat_tab + (size_t)hi[6 + k] * e0 + c, gs);
        }
        // (2) the small hot tables: a flat walk over (row, small column) with LDS float atomics (a few dozen per row)
        for (int i = threadIdx.x; i < nr * small; i += FE_THREADS) {
            const int r = i / small, cs = i - r * small;
            const float gv = g0[(size_t)r * lddl + e0 + cs];
            const int* hi = hd.idx + r * FE_HDR;
            if (cs < e1) {
                const float s0 = hd.sen[r * 4], s1 = hd.sen[r * 4 + 1], s2 = hd.sen[r * 4 + 2];
                const float pre = p.sen_b[cs] + p.sen_w[cs * 3] * s0 + p.sen_w[cs * 3 + 1] * s1 + p.sen_w[cs * 3 + 2] * s2;
                if (pre > 0.f && gv != 0.f) {            // ReLU'
                    atomicAdd(a_sen + cs * 4 + 0, gv * s0);
                    atomicAdd(a_sen + cs * 4 + 1, gv * s1);
                    atomicAdd(a_sen + cs * 4 + 2, gv * s2);
                    atomicAdd(a_sen + cs * 4 + 3, gv);
                }
            } else if (cs < e1 + e2) {
                atomicAdd(a_type + hi[5] * e2 + (cs - e1), gv);
            } else {
                const int k = cs - e1 - e2;
                atomicAdd(a_year + hi[0] * e3 + k, gv);
                atomicAdd(a_month + hi[1] * e3 + k, gv);
                atomicAdd(a_day + hi[2] * e3 + k, gv);
                atomicAdd(a_hour + hi[3] * e3 + k, gv);
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
    const dim3 grid(std::min((nrows + FE_ROWS - 1) / FE_ROWS, FE_MAX_BLOCKS));
    if (x_is_f64) {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<double>, grid, dim3(FE_THREADS), shm, st, p, (const double*)x, dlab, lddl, nrows);
    } else {
        if (shm > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)frontend_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(frontend_bwd_kernel<float>, grid, dim3(FE_THREADS), shm, st, p, (const float*)x, dlab, lddl, nrows);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Category-table gradient without one float atomic per reference.  A row refers to the table 1 + NS times (its category, weight
// 1, and its NS sub-category slots, weight 1/NS each); the scatter above costs (1 + NS) * e0 float atomics per row, and float
// atomics run at 65-165 G/s chip-wide (L2 atomic units; one private copy of the table per XCD changed nothing), which made this
// the slowest small kernel of a step (0.59 ms at C3, 0.21 of a 1.5 ms step at the reference's default sizes).  Instead:
//   count   histogram of the references' category ids                      (integer atomics on n_cat counters)
//   scan    exclusive prefix -> first slot of every id
//   fill    counting sort: reference i goes to slot start[id] + cursor[id]++
//   gather  one wave per chunk of 64 SORTED references: it walks them in order, sums weight * dlab[row, :e0] in registers
//           while the id stays the same and adds a finished run to its table row (float atomics: one run per id and chunk,
//           ~N/64 + n_cat runs instead of N references)
// Both row sets of a step (history and candidate rows) go through one sort.  Order inside a run depends on the fill's atomics,
// so sums differ in the last bits from run to run -- as the atomic scatter's did.
struct CatRefs {
    const void* x[2]; const float* dlab[2];
    int nrows[2], xcols[2], lddl[2];
    int c_cat, NS, n_cat, e0;
    long nref;                                     // (nrows[0] + nrows[1]) * (NS + 1)
};

template <typename XT>
__global__ __launch_bounds__(256) void cat_count_kernel(const CatRefs p, int* __restrict__ count, int* __restrict__ cat_of) {
    const int S = p.NS + 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.nref; i += (long)gridDim.x * 256) {
        long r = i / S;
        const int j = (int)(i - r * S);
        const int set = r >= p.nrows[0];
        if (set) r -= p.nrows[0];
        const XT* xr = reinterpret_cast<const XT*>(p.x[set]) + (size_t)r * p.xcols[set];
        int c = (int)xr[p.c_cat + j];
        c = c < 0 ? 0 : (c >= p.n_cat ? p.n_cat - 1 : c);                 // (out-of-range ids were flagged by the forward)
        cat_of[i] = c;
        atomicAdd(count + c, 1);
    }
}

__global__ __launch_bounds__(1024) void cat_scan_kernel(const int* __restrict__ count, int* __restrict__ start, int n) {
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                           // inclusive scan of the 1024 partial sums
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) { start[i] = run; run += count[i]; }
    if (threadIdx.x == 1023) start[n] = part[1023];
}

__global__ __launch_bounds__(256) void cat_fill_kernel(long nref, const int* __restrict__ cat_of, const int* __restrict__ start,
                                                       int* __restrict__ cursor, int* __restrict__ refs, int* __restrict__ refcat) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nref; i += (long)gridDim.x * 256) {
        const int c = cat_of[i];
        const int pos = start[c] + atomicAdd(cursor + c, 1);
        refs[pos] = (int)i;
        refcat[pos] = c;
    }
}

constexpr int CAT_COLS = 8;                        // column chunks of 64 per lane: e0 <= 512
__global__ __launch_bounds__(256) void cat_gather_kernel(const CatRefs p, const int* __restrict__ refs, const int* __restrict__ refcat,
                                                         float* __restrict__ d_cat) {
    const int lane = threadIdx.x & 63;
    const long chunk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = chunk * 64;
    if (base >= p.nref) return;
    const int S = p.NS + 1;
    const float inv_ns = 1.0f / (float)p.NS;
    const int my_ref = base + lane < p.nref ? refs[base + lane] : -1;
    const int my_cat = base + lane < p.nref ? refcat[base + lane] : -1;
    const int n_here = (int)min((long)64, p.nref - base);
    float acc[CAT_COLS];
#pragma unroll
    for (int k = 0; k < CAT_COLS; ++k) acc[k] = 0.f;
    int cur = __builtin_amdgcn_readfirstlane(my_cat);
    auto flush = [&](int c) {
        float* row = d_cat + (size_t)c * p.e0;
#pragma unroll
        for (int k = 0; k < CAT_COLS; ++k) {
            if (64 * k + lane < p.e0) atomicAdd(row + 64 * k + lane, acc[k]);
            acc[k] = 0.f;
        }
    };
    for (int t0 = 0; t0 < n_here; t0 += 4) {
        // the rows of four references are requested together (nothing between them depends on the run bookkeeping)
        float v[4][CAT_COLS];
        int cs[4];
        float ws[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            const int i = __builtin_amdgcn_readlane(my_ref, t & 63);
            cs[u] = t < n_here ? __builtin_amdgcn_readlane(my_cat, t & 63) : -1;
            long r = i / S;
            const int j = i - (int)r * S;
            ws[u] = j ? inv_ns : 1.0f;
            const int set = r >= p.nrows[0];
            if (set) r -= p.nrows[0];
            const float* g = p.dlab[set] + (size_t)r * p.lddl[set];
#pragma unroll
            for (int k = 0; k < CAT_COLS; ++k) v[u][k] = (t < n_here && 64 * k + lane < p.e0) ? g[64 * k + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cs[u] < 0) break;
            if (cs[u] != cur) { flush(cur); cur = cs[u]; }
#pragma unroll
            for (int k = 0; k < CAT_COLS; ++k) acc[k] = fmaf(ws[u], v[u][k], acc[k]);
        }
    }
    flush(cur);
}

long cat_grad_ws_ints(int n_cat, long nrows_total, int n_sub) { return 3L * n_cat + 4 + 3L * nrows_total * (n_sub + 1); }

hipError_t cat_grad_launch(const void* x0, int nrows0, int xcols0, const float* dlab0, int lddl0,
                           const void* x1, int nrows1, int xcols1, const float* dlab1, int lddl1, int x_is_f64,
                           int P, int n_sub, int n_cat, int e0, float* d_cat, int* ws, hipStream_t st) {
    CatRefs p = {};
    p.x[0] = x0; p.x[1] = x1; p.dlab[0] = dlab0; p.dlab[1] = dlab1;
    p.nrows[0] = nrows0; p.nrows[1] = nrows1; p.xcols[0] = xcols0; p.xcols[1] = xcols1; p.lddl[0] = lddl0; p.lddl[1] = lddl1;
    p.c_cat = 4 + P; p.NS = n_sub; p.n_cat = n_cat; p.e0 = e0;
    p.nref = ((long)nrows0 + nrows1) * (n_sub + 1);
    if (p.nref <= 0) return hipSuccess;
    if (e0 > 64 * CAT_COLS || n_sub < 1 || p.nref >= (1L << 31)) return hipErrorInvalidValue;
    int* count = ws;
    int* cursor = ws + n_cat;
    int* start = ws + 2 * n_cat;                   // n_cat + 1 entries
    int* cat_of = ws + 3 * n_cat + 4;
    int* refs = cat_of + p.nref;
    int* refcat = refs + p.nref;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int) * 2 * (size_t)n_cat, st);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)std::min<long>((p.nref + 255) / 256, 2048);
    if (x_is_f64) hipLaunchKernelGGL(cat_count_kernel<double>, dim3(blocks), dim3(256), 0, st, p, count, cat_of);
    else          hipLaunchKernelGGL(cat_count_kernel<float>, dim3(blocks), dim3(256), 0, st, p, count, cat_of);
    hipLaunchKernelGGL(cat_scan_kernel, dim3(1), dim3(1024), 0, st, count, start, n_cat);
    hipLaunchKernelGGL(cat_fill_kernel, dim3(blocks), dim3(256), 0, st, p.nref, cat_of, start, cursor, refs, refcat);
    const long chunks = (p.nref + 63) / 64;
    hipLaunchKernelGGL(cat_gather_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, st, p, refs, refcat, d_cat);
    return hipGetLastError();
}

}  // namespace nrm
