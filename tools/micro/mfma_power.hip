// What does the board power limit allow?  512-thread workgroups (two waves per SIMD), one per CU, 1024 workgroups, each wave runs a loop of
// independent fp16 MFMAs on RANDOM operand bits (the clock under MFMA load depends on the data), optionally with LDS fragment reads at the
// rate of the row weight-gradient kernel (2 x ds_read_b64 per 3 MFMAs of 16x16x32).  Reports launch time, clock (shader cycles / 100 MHz
// wall ticks inside the kernel) and the MFMA rate reached.
//   shape 0: v_mfma_f32_16x16x32_f16, 18 accumulator tiles (the kernels' wave tile 48 x 96)
//   shape 1: v_mfma_f32_32x32x16_f16, 6 accumulator tiles (64 x 96)
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 rnd(unsigned s) {
    f16x8 v;
    for (int k = 0; k < 8; ++k) {
        s = s * 1664525u + 1013904223u;
        unsigned short b = (unsigned short)((s >> 16) & 0x83ffu) | (unsigned short)((12 + ((s >> 8) & 3)) << 10);   // |x| in [2^-3, 2^0)
        v[k] = __builtin_bit_cast(_Float16, b);
    }
    return v;
}

template <int SHAPE, int READS>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* st) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        reinterpret_cast<unsigned*>(lds)[i] = (h & 0x83ff83ffu) | 0x30003000u;
    }
    __syncthreads();
    f16x8 fa[2][3], fb[2][6];
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < 3; ++i) fa[p][i] = rnd(tid * 31 + p * 7 + i);
        for (int j = 0; j < 6; ++j) fb[p][j] = rnd(tid * 17 + p * 5 + j + 100);
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[3][6];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[PA[q]][i], fb[PB[q]][j], acc[i][j], 0, 0, 0);
                if (READS) {                                             // 18 x 16 B per wave and 54 MFMAs = the kernel's 36 ds_read_b64
                    const int o = ((it * 3 + q) & 7) * 8192 + lane * 16;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(lds + o + r * 1024);
                        if (q == 0) fa[r & 1][r >> 1] = v; else fb[r & 1][(q - 1) * 3 + (r >> 1)] = v;
                    }
                }
            }
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][3];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)                           // 64 x 96 x K32 per product = 2 K16 steps x 6 tiles
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[PA[q]][i + kk], fb[PB[q]][j + 3 * kk], acc[i][j], 0, 0, 0);
                if (READS) {                                             // (2 + 3) fragments x 2 K16 steps x 2 planes = 20 x 16 B per 36 MFMAs
                    const int o = ((it * 3 + q) & 7) * 8192 + lane * 16;
#pragma unroll
                    for (int r = 0; r < 7; ++r) {
                        if (q == 2 && r == 6) break;
                        const f16x8 v = *reinterpret_cast<const f16x8*>(lds + o + r * 1024);
                        if (q == 0 && r < 6) fa[r & 1][r >> 1] = v; else fb[r & 1][(r + q) % 6] = v;
                    }
                }
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sum += acc[i][j][0] + acc[i][j][15];
    }
    if (tid == 0) { st[blockIdx.x * 2] = clock64() - c0; st[blockIdx.x * 2 + 1] = wall_clock64() - w0; }
    out[blockIdx.x * 512 + tid] = sum;
}

template <int SHAPE, int READS>
void run(float* out, unsigned long long* st) {
    const int grid = 1024, iters = 600;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 30; ++w) hipLaunchKernelGGL((k<SHAPE, READS>), dim3(grid), dim3(512), 0, 0, out, iters, st);
    hipEventRecord(e0);
    const int reps = 60;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, READS>), dim3(grid), dim3(512), 0, 0, out, iters, st);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2048]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0, w = 0; for (int i = 0; i < grid; ++i) { c += (double)h[2 * i]; w += (double)h[2 * i + 1]; }
    const double flops = (double)grid * 8 * iters * (SHAPE == 0 ? 54.0 * 16384 : 36.0 * 32768);
    printf("shape %s reads %d: %.1f us per launch, clock %.2f GHz, %.0f TFLOP/s fp16 MFMA (%.1f %% of 2.5 PF)\n", SHAPE ? "32x32x16" : "16x16x32", READS, ms / reps * 1e3,
           c / w / 10.0, flops / (ms / reps * 1e-3) / 1e12, flops / (ms / reps * 1e-3) / 2.5e15 * 100);
}

int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&st, 2048 * 8);
    run<0, 0>(out, st); run<1, 0>(out, st); run<0, 1>(out, st); run<1, 1>(out, st);
    run<0, 0>(out, st); run<1, 0>(out, st);
    return 0;
}
