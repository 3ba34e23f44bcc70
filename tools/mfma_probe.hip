// Micro-probe of what a bf16x6-shaped inner loop can reach on gfx950: 36 v_mfma_f32_32x32x16_bf16 per iteration on a 2 x 3 grid of
// accumulators (the wave tile of conv_fwd_x6), 256-thread workgroups, two per CU, with the loop's other ingredients added one by one.
//   mode 0: MFMAs only                      mode 1: + 15 ds_read_b128 fragment reads per iteration
//   mode 2: mode 1 + one barrier per iteration
//   mode 3: mode 2 + a 44-instruction vector block and three 8-byte LDS writes per iteration (one activation quad split)
//   mode 4: mode 3 + five 16-byte global loads per iteration (L2-resident)
//   mode 5: mode 4 + five 1 KiB LDS-DMA pieces per wave per iteration and s_waitcnt vmcnt(0) in front of the barrier
// each with smooth (few toggling bits) and random operand bits: the chip's clock under MFMA load depends on the data
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe ; run: ./mfma_probe [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// The wgrad-shaped variant: R extra quad splits per iteration (conv_wgrad_x6 spends ~4.6 vector instructions per MFMA on its
// split + transpose loader), with (IL) or without the 1-MFMA : 5-VALU sched_group_barrier interleave, at one (OCC = 1: the LDS
// array is padded past half a CU) or two workgroups per CU.
template <int OCC, int R, bool IL>
__global__ __launch_bounds__(256, OCC) void probe_wg(const float* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[2][OCC == 1 ? 45000 : 3 * (144 + 192) * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (int)sizeof(lds) / 4; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        reinterpret_cast<unsigned*>(lds)[i] = (h & 0x807f807fu) | (((h >> 7) & 7u) + 120u) << 7 | (((h >> 23) & 7u) + 120u) << 23;
    }
    __syncthreads();
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[3][2], fb[3][3];
    const int aoff = ((wave >> 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16;
    const int boff = 3 * 144 * 32 + ((wave & 1) * 96 + (lane & 31)) * 32 + (lane >> 5) * 16;
    for (int p = 0; p < 3; ++p) {
        for (int i = 0; i < 2; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(&lds[0][p * 144 * 32 + aoff + i * 1024]);
        for (int j = 0; j < 3; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(&lds[0][p * 192 * 32 + boff + j * 1024]);
    }
    f32x4 g[R > 0 ? R : 1];
    for (int r = 0; r < (R > 0 ? R : 1); ++r) g[r] = f32x4{1.f + r, 2.f, 3.f, 4.f};
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 0, 1, 1, 0};
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[sl]][i], fb[PB[sl]][j], acc[i][j], 0, 0, 0);
            // fragments re-read in place after their last use
            if (sl == 0) for (int i = 0; i < 2; ++i) fa[2][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][2 * 144 * 32 + aoff + i * 1024]);
            if (sl == 1) for (int j = 0; j < 3; ++j) fb[2][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][2 * 192 * 32 + boff + j * 1024]);
            if (sl == 3) for (int i = 0; i < 2; ++i) fa[1][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][1 * 144 * 32 + aoff + i * 1024]);
            if (sl == 4) for (int j = 0; j < 3; ++j) fb[1][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][1 * 192 * 32 + boff + j * 1024]);
#pragma unroll
            for (int r = sl; r < R; r += 6) {          // this slot's share of the split work
                f32x4 v = g[r];
                const unsigned h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
                const float r0 = v[0] - lo_f(h01), r1 = v[1] - hi_f(h01), r2 = v[2] - lo_f(h23), r3 = v[3] - hi_f(h23);
                const unsigned m01 = cvt_pk(r0, r1), m23 = cvt_pk(r2, r3);
                const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m23), s3 = r3 - hi_f(m23);
                const unsigned l01 = cvt_pk(s0, s1), l23 = cvt_pk(s2, s3);
                char* st = &lds[buf ^ 1][((tid >> 2) + 64 * (r & 1)) * 32 + (tid & 3) * 8];
                *reinterpret_cast<u32x2*>(st) = u32x2{h01, h23};
                *reinterpret_cast<u32x2*>(st + 144 * 32) = u32x2{m01, m23};
                *reinterpret_cast<u32x2*>(st + 2 * 144 * 32) = u32x2{l01, l23};
                g[r] = g[r] * 1.0001f + f32x4{s0, s1, s2, s3};
            }
            if (IL) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][aoff + i * 1024]);
        for (int j = 0; j < 3; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][boff + j * 1024]);
        __syncthreads();
    }
    float s = g[0][0];
    for (int r = 1; r < R; ++r) s += g[r][0];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int OCC, int R, bool IL>
static void run_wg(const float* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * OCC;
    const int its = iters * 2 / OCC;                    // the same total work either way
    hipLaunchKernelGGL((probe_wg<OCC, R, IL>), dim3(grid), dim3(256), 0, 0, src, out, its);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_wg<OCC, R, IL>), dim3(grid), dim3(256), 0, 0, src, out, its);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = (double)grid * 4 * its * 36 * 32768.0;
    printf("wgrad-shaped: %d workgroup(s)/CU, %d quad splits per iteration (%s): %8.1f us  %7.1f TFLOP/s bf16  (%.3f of 2500)\n", OCC, R,
           IL ? "interleaved" : "block in front", ms * 1e3, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
}

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256, 2) void probe(const float* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[2][3 * (144 + 192) * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (int)sizeof(lds) / 4; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        // RANDOM: two random bf16 values in [-2, 2) per word (random mantissas and signs, exponents 120..127)
        const unsigned rnd = (h & 0x807f807fu) | (((h >> 7) & 7u) + 120u) << 7 | (((h >> 23) & 7u) + 120u) << 23;
        reinterpret_cast<unsigned*>(lds)[i] = RANDOM ? rnd : __builtin_bit_cast(unsigned, 1e-3f * (float)(i & 255));
    }
    __syncthreads();
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[3][2], fb[3][3];
    const int aoff = ((wave >> 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16;
    const int boff = 3 * 144 * 32 + ((wave & 1) * 96 + (lane & 31)) * 32 + (lane >> 5) * 16;
    for (int p = 0; p < 3; ++p) {
        for (int i = 0; i < 2; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(&lds[0][p * 144 * 32 + aoff + i * 1024]);
        for (int j = 0; j < 3; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(&lds[0][p * 192 * 32 + boff + j * 1024]);
    }
    f32x4 g = {1.f, 2.f, 3.f, 4.f};
    f32x4 gl[5];
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        auto mma = [&](int pa, int pb) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
        };
        if (MODE >= 4) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
                gl[k] = *reinterpret_cast<const f32x4*>(src + ((size_t)((it * 5 + k) & 63) * 65536 + (size_t)blockIdx.x * 256 + tid) * 4 % (1 << 22));
        }
        mma(2, 0);
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[2][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][2 * 144 * 32 + aoff + i * 1024]);
        }
        if (MODE >= 3) {       // one quad: split into three bf16 planes and store
            f32x4 v = g;
            const unsigned h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
            const float r0 = v[0] - lo_f(h01), r1 = v[1] - hi_f(h01), r2 = v[2] - lo_f(h23), r3 = v[3] - hi_f(h23);
            const unsigned m01 = cvt_pk(r0, r1), m23 = cvt_pk(r2, r3);
            const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m23), s3 = r3 - hi_f(m23);
            const unsigned l01 = cvt_pk(s0, s1), l23 = cvt_pk(s2, s3);
            char* st = &lds[buf ^ 1][(tid >> 2) * 32 + (tid & 3) * 8];
            *reinterpret_cast<u32x2*>(st) = u32x2{h01, h23};
            *reinterpret_cast<u32x2*>(st + 144 * 32) = u32x2{m01, m23};
            *reinterpret_cast<u32x2*>(st + 2 * 144 * 32) = u32x2{l01, l23};
            g = g * 1.0001f + (MODE >= 4 ? gl[0] + gl[1] + gl[2] + gl[3] + gl[4] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 2);
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[2][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][3 * 144 * 32 + 2 * 192 * 32 + boff - 3 * 144 * 32 + j * 1024]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[1][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][1 * 144 * 32 + aoff + i * 1024]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[1][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][3 * 144 * 32 + 1 * 192 * 32 + boff - 3 * 144 * 32 + j * 1024]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(&lds[buf][aoff + i * 1024]);
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(&lds[buf][3 * 144 * 32 + boff - 3 * 144 * 32 + j * 1024]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 5) {      // five 1 KiB LDS-DMA pieces per wave into the other buffer's weight image, drained before the barrier
            typedef __attribute__((address_space(3))) void lds_void;
            typedef const __attribute__((address_space(1))) void glb_void;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int piece = (wave + 4 * k) % 18;
                const float* gsrc = src + ((size_t)((it * 18 + piece) & 1023) * 256 + lane * 4);
                __builtin_amdgcn_global_load_lds((glb_void*)gsrc, (lds_void*)&lds[buf ^ 1][3 * 144 * 32 + piece * 1024], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = g[0];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE, bool RANDOM>
static void run(const float* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 512;
    hipLaunchKernelGGL((probe<MODE, RANDOM>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, RANDOM>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = (double)grid * 4 * iters * 36 * 32768.0;
    printf("%s mode %d: %8.1f us per launch  %7.1f TFLOP/s bf16  (%.3f of 2500)\n", RANDOM ? "random" : "smooth", MODE, ms * 1e3, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 324;     // 324 stages = the 4 x 128^2, 192 -> 192 conv with one 128 x 192 tile per workgroup
    float *src, *out;
    hipMalloc(reinterpret_cast<void**>(&src), (size_t)(1 << 22) * 4 + 65536);
    hipMemset(src, 0, (size_t)(1 << 22) * 4 + 65536);
    hipMalloc(reinterpret_cast<void**>(&out), 512 * 256 * 4);
    run<0, false>(src, out, iters);
    run<0, true>(src, out, iters);
    run<2, false>(src, out, iters);
    run<2, true>(src, out, iters);
    run<4, false>(src, out, iters);
    run<4, true>(src, out, iters);
    run<5, false>(src, out, iters);
    run<5, true>(src, out, iters);
    run_wg<1, 0, false>(src, out, iters);
    run_wg<2, 0, false>(src, out, iters);
    run_wg<1, 6, false>(src, out, iters);
    run_wg<1, 6, true>(src, out, iters);
    run_wg<2, 6, false>(src, out, iters);
    run_wg<2, 6, true>(src, out, iters);
    return 0;
}
