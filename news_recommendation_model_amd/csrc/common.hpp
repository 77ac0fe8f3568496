// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the UserModel hot path.
// fp32 everywhere: the matrix work runs on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrm {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;       // one v_mfma_f32_16x16x32_bf16 operand (4 VGPRs)

constexpr int WAVE = 64;

// D(16x16) += A(16x4) * B(4x16), f32 in / f32 accumulate.
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// it receives D[row = 4*(l>>4) + r][col = l&15] in element r of the accumulator.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Exact-erf GELU (nn.GELU() default in the reference MLP, models/attention_model.py:21,27),
// branch free.  erfc(a), a >= 0, by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7):
//   erfc(a) = (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-a^2),  t = 1/(1 + p a)
// Phi(x) = 0.5 erfc(-x/sqrt2);  gelu(x) = x Phi(x);  gelu'(x) = Phi(x) + x phi(x).
struct GeluParts { float cdf; float e; };      // e = exp(-x^2/2)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float a = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * a * a);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;            // 0.5 * erfc(|x|/sqrt2)
    GeluParts r;
    r.cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
    r.e = e;
    return r;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const GeluParts g = gelu_parts(x);
    return fmaf(x * 0.39894228040143267794f, g.e, g.cdf);
}

// Four GELUs at once on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of work per issue slot), round 5.  The
// same evaluation as gelu_parts -- A&S 7.1.26 with the same coefficients -- rearranged so that nothing but the reciprocal, the exponential
// and |x| is scalar:  he = 0.5 erfc(|x| / sqrt2) = (0.5 poly(t)) t exp2(-x^2 log2(e) / 2),   gelu(x) = x Phi(x) = 0.5 x + |x| (0.5 - he).
// 9.5 instructions per element instead of 19 (the forward's epilogue evaluates D of them per score; at D = 64 as many issue slots as
// the contraction itself).  Differs from gelu_f by roundings only (the constants are folded, |x| c |x| c became x^2 c^2).
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 gelu_half_erfc2(f32x2 x, f32x2 ax) {
    const f32x2 t = f32x2{__builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax[0], 1.0f)),
                          __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax[1], 1.0f))};
    const f32x2 ea = (x * x) * f32x2{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
    f32x2 p = __builtin_elementwise_fma(f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, t, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
    p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
    return (p * t) * e;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
    const f32x2 he = gelu_half_erfc2(x, ax);
    return __builtin_elementwise_fma(ax, f32x2{0.5f, 0.5f} - he, x * f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ f32x4 gelu4(f32x4 x) {
    const f32x2 a = gelu2(f32x2{x[0], x[1]}), b = gelu2(f32x2{x[2], x[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}
// sum_e w[e] * gelu(z[e])
// (the two pairs one after the other: interleaved, as hipcc schedules them by itself, the pair-aligned temporaries of both are live at
// once -- 9 spilled registers in the forward kernel, whose launch bound leaves it 128)
__device__ __forceinline__ float gelu_dot4(f32x4 w, f32x4 z) {
    const f32x2 a = f32x2{w[0], w[1]} * gelu2(f32x2{z[0], z[1]});
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 s = __builtin_elementwise_fma(f32x2{w[2], w[3]}, gelu2(f32x2{z[2], z[3]}), a);
    return s[0] + s[1];
}
// gelu'(x) = Phi(x) + x phi(x),  Phi(x) = 0.5 + copysign(0.5 - he, x),  phi(x) = exp(-x^2 / 2) / sqrt(2 pi): four at once, times g
__device__ __forceinline__ f32x4 gelu_grad4_times(f32x4 x, f32x4 g) {
    f32x4 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 xx = f32x2{x[2 * h], x[2 * h + 1]};
        const f32x2 ax = f32x2{fabsf(xx[0]), fabsf(xx[1])};
        const f32x2 t = f32x2{__builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax[0], 1.0f)),
                              __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax[1], 1.0f))};
        const f32x2 ea = (xx * xx) * f32x2{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};
        const f32x2 e = f32x2{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
        f32x2 p = __builtin_elementwise_fma(f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, t, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
        p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
        p = __builtin_elementwise_fma(p, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
        p = __builtin_elementwise_fma(p, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
        const f32x2 q = f32x2{0.5f, 0.5f} - (p * t) * e;                 // 0.5 - he >= 0
        const f32x2 cdf = f32x2{0.5f, 0.5f} + f32x2{copysignf(q[0], xx[0]), copysignf(q[1], xx[1])};
        const f32x2 d = __builtin_elementwise_fma(xx * f32x2{0.39894228040143267794f, 0.39894228040143267794f}, e, cdf);
        const f32x2 o = d * f32x2{g[2 * h], g[2 * h + 1]};
        r[2 * h] = o[0]; r[2 * h + 1] = o[1];
    }
    return r;
}

__device__ __forceinline__ float wave_sum16(float v) {   // sum over the 16 lanes sharing l>>4
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48) with the gfx950 row-swap instructions: two VALU ops
// per step instead of an LDS-crossbar ds_bpermute.  v_permlane16_swap exchanges rows 1<->0' and 3<->2' of its
// two operands, v_permlane32_swap the upper and lower halves.
__device__ __forceinline__ float sum_rows4(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ float wave_sum64(float v) {
    v = wave_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// 16-B slot swizzle of a 64-B (16-float) LDS row: slot s of row r holds columns 4*(s ^ swz4(r)) ..+3.
// ds_read_b128 is served in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32
// (MI355X_MICROARCH.md, LDS) -- so for a fragment read (row = lane&15, slot = lane>>4) a group holds rows
// {0-3,12-15} of one slot and rows {4-11} of its neighbour: XOR-ing 3 into rows 8-15 makes the 16 reads of every
// group land on 16 different 4-bank columns of the unpadded image.
// 16-byte buffer store whose data registers may be rewritten right afterwards.  hipcc (ROCm 7.2) pads the "VMEM store of more
// than 64 bits followed by a VALU write of its data registers" hazard only when the store's soffset is an immediate; with an
// SGPR soffset it inserts nothing, yet on gfx950 the store was seen delivering values the next VALU instructions had already
// written into its data registers (pwattn_fwd_walk_kernel: gelu(z) instead of z in 0.1 % of the stored pre-activations).  The
// wait states are pinned right behind the store.
__device__ __forceinline__ void store_b128_guarded(u32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only parses the kernels that call this)
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voffset, soffset, 0);
    // NRM_STORE_GUARD (analysis builds only; scripts/_diag/store_hazard_isa.py): 0 = no guard, 1 = the wait
    // states alone, 2 (default) = wait states pinned behind the store by scheduling barriers
#if !defined(NRM_STORE_GUARD) || NRM_STORE_GUARD == 2
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3");
    __builtin_amdgcn_sched_barrier(0);
#elif NRM_STORE_GUARD == 1
    asm volatile("s_nop 3");
#endif
#endif
}

__host__ __device__ inline int swz4(int row) { return ((row >> 3) & 1) * 3; }

}  // namespace nrm
