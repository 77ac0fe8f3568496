// Parameter block of the embedding front end (frontend.hip), shared with capi.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace nrm {
struct FrontendParams {
    // tables / small weights (fp32)
    const float *cat_tab, *sen_w, *sen_b, *type_tab, *year_tab, *month_tab, *day_tab, *hour_tab;
    // their gradients (backward only; accumulated into)
    float *d_cat_tab, *d_sen_w, *d_sen_b, *d_type_tab, *d_year_tab, *d_month_tab, *d_day_tab, *d_hour_tab;
    int n_cat, n_type, n_year, n_month, n_day, n_hour;
    int e0, e1, e2, e3;
    int P, n_sub, xcols, behaviour;        // packed-row geometry; behaviour = 1 for history rows (read_time, scroll)
    float* lab; int ldlab;                 // [rows, ldlab] label rows out
    float* ti; int ldti;                   // [rows, ldti]  text/image rows out
    int* err;                              // set to 1 on an out-of-range index
};
hipError_t frontend_fwd_launch(const FrontendParams& p, const void* x, int x_is_f64, int nrows, hipStream_t st);
hipError_t frontend_bwd_launch(const FrontendParams& p, const void* x, int x_is_f64, const float* dlab, int lddl,
                               int nrows, hipStream_t st);
long cat_grad_ws_ints(int n_cat, long nrows_total, int n_sub);
hipError_t cat_grad_launch(const void* x0, int nrows0, int xcols0, const float* dlab0, int lddl0,
                           const void* x1, int nrows1, int xcols1, const float* dlab1, int lddl1, int x_is_f64,
                           int P, int n_sub, int n_cat, int e0, float* d_cat, int* ws, hipStream_t st);
}  // namespace nrm
