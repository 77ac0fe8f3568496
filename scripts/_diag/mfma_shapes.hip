// Diagnostic: the K loop of the attention forward / gemm_nt with v_mfma_f32_16x16x4_f32 (what ships) against the
// same loop on v_mfma_f32_32x32x2_f32 tiles -- equal peak rate, half the A/B fragment reads and half the MFMA issue slots per FLOP.
// Bare loops: operands come from a pre-filled LDS image by ds_read_b128 exactly as in pwattn_fwd_kernel (13 A fragments + the
// t * h product per 16-wide chunk for a 16 x 208 wave tile; 6 A fragments + 2 products for a 32 x 96 wave tile), no DMA, no barrier,
// no epilogue, 4 workgroups of 4 waves per CU (LDS-limited, as the product kernel).  Prints TFLOP/s and the fraction of 157.3.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/_diag/mfma_shapes.hip -o scripts/_diag/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CHUNKS = 25;              // K = 400 = 25 chunks of 16, as at C3
constexpr int LDS_FLOATS = 9216;        // 36 KB: four workgroups per CU

template <int SHAPE>                    // 0: 16x16x4, wave tile 13 x 1;  1: 32x32x2, wave tile 3 x 1 (32-row tiles)
__global__ __launch_bounds__(256, 4) void kloop(float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_FLOATS; i += 256) smem[i] = 1.0f + 1e-3f * (float)((i * 7 + blockIdx.x) % 13);
    __syncthreads();
    float sum = 0.f;
    if (SHAPE == 0) {
        const int r16 = lane & 15, q = lane >> 4;
        const int rslot = 4 * (q ^ (((r16 >> 3) & 1) * 3));
        f32x4 acc[13];
#pragma unroll
        for (int it = 0; it < 13; ++it) acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < iters; ++n) {
            for (int c = 0; c < CHUNKS; ++c) {
                const float* buf = smem + (c & 1) * 4352;                                   // [208 W rows | 64 h rows] x 16 floats
                const float* Tl = buf + 208 * 16;
                const int ro = ((wave * 16 + r16) * 16 + rslot) & 1023;
                const f32x4 pf = *reinterpret_cast<const f32x4*>(&Tl[ro]) * *reinterpret_cast<const f32x4*>(&Tl[(ro + 512) & 1023]);
                f32x4 af = *reinterpret_cast<const f32x4*>(&buf[r16 * 16 + rslot]);
#pragma unroll
                for (int it = 0; it < 13; ++it) {
                    f32x4 afn = af;
                    if (it + 1 < 13) afn = *reinterpret_cast<const f32x4*>(&buf[((it + 1) * 16 + r16) * 16 + rslot]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], pf[j], acc[it], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    af = afn;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 13; ++it) sum += acc[it][0] + acc[it][1] + acc[it][2] + acc[it][3];
    } else {
        // lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]: per 16-wide chunk a lane needs k = {kh, kh + 2, ...}:
        // with two 16-byte fragments per row (columns 8 kh .. 8 kh + 7) the eight MFMAs of a chunk take element e of them
        const int r32 = lane & 31, kh = lane >> 5;
        f32x16 acc[3];
#pragma unroll
        for (int it = 0; it < 3; ++it)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[it][e] = 0.f;
        for (int n = 0; n < iters; ++n) {
            for (int c = 0; c < CHUNKS; ++c) {
                const float* buf = smem + (c & 1) * 4352;
                const float* Tl = buf + 96 * 16;                                             // [96 W rows | 128 h rows | ...]
                const int ro = (((wave * 32 + r32) * 16 + 8 * kh)) & 2047;
                f32x4 p0 = *reinterpret_cast<const f32x4*>(&Tl[ro]) * *reinterpret_cast<const f32x4*>(&Tl[(ro + 1024) & 2047]);
                f32x4 p1 = *reinterpret_cast<const f32x4*>(&Tl[ro + 4]) * *reinterpret_cast<const f32x4*>(&Tl[((ro + 1024) & 2047) + 4]);
                f32x4 a0 = *reinterpret_cast<const f32x4*>(&buf[r32 * 16 + 8 * kh]);
                f32x4 a1 = *reinterpret_cast<const f32x4*>(&buf[r32 * 16 + 8 * kh + 4]);
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    f32x4 n0 = a0, n1 = a1;
                    if (it + 1 < 3) {
                        n0 = *reinterpret_cast<const f32x4*>(&buf[((it + 1) * 32 + r32) * 16 + 8 * kh]);
                        n1 = *reinterpret_cast<const f32x4*>(&buf[((it + 1) * 32 + r32) * 16 + 8 * kh + 4]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], p0[j], acc[it], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], p1[j], acc[it], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = n0; a1 = n1;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 3; ++it)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[it][e];
    }
    out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE>
static double run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kloop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kloop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // FLOPs per wave and chunk: 52 x (2 * 16 * 16 * 4)  |  24 x (2 * 32 * 32 * 2)
    const double per_chunk = SHAPE == 0 ? 52.0 * 2048.0 : 24.0 * 4096.0;
    return (double)blocks * 4 * iters * CHUNKS * per_chunk / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const int blocks = 256 * 4 * 4;                                  // four rounds of four workgroups per CU
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        const double a = run<0>(blocks, iters, out), b = run<1>(blocks, iters, out);
        printf("bare K loop, 4 waves per SIMD, operands by ds_read_b128:  16x16x4 (13x1 tiles) %.1f TFLOP/s = %.3f of 157.3   |   "
               "32x32x2 (3x1 tiles of 32) %.1f TFLOP/s = %.3f\n", a, a / 157.3, b, b / 157.3);
    }
    hipFree(out);
    return 0;
}
