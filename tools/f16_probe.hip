// Probe for an fp16 two-way split of fp32 GEMM operands on gfx950 (x = h1 + h2 exactly to 2^-24 when scaled into fp16's normal range):
//   1. does v_mfma_f32_32x32x16_f16 honour fp16 DENORMAL inputs?          (decides how much head-room the per-tensor scale needs)
//   2. MFMA issue rate f16 vs bf16 (same instruction shape)
//   3. accuracy of  h1*g1 + h1*g2 + h2*g1  (3 products) against the bf16 three-way split with 6 products and an fp32 fma chain,
//      all against fp64, on a 256 x 192 x 1728 GEMM with activation-like / weight-like / gradient-like operands
// build: hipcc -O3 --offload-arch=gfx950 tools/f16_probe.hip -o gpurun_out/f16_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void denorm_kernel(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)9.5367431640625e-07f; b[k] = (_Float16)1024.f; }   // 2^-20 (subnormal), 2^10
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];           // 16 * 2^-10 = 0.015625 if denormals are honoured, 0 if flushed
}

template <bool F16>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1.f + threadIdx.x * 0.001f + k); b[k] = (_Float16)(0.5f + k); ab[k] = (short)(0x3f80 + threadIdx.x + k); bb[k] = (short)(0x3f00 + k); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2)); }
__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }

// One wave per 32 x 32 output tile.  mode 0: fp16 2-way split, 3 products; 1: fp16 split, 4 products; 2: bf16 3-way split, 6 products; 3: fp32 fma chain
__global__ __launch_bounds__(64) void gemm_kernel(const float* A, const float* B, float* C, int M, int N, int K, float sa, float sb, int mode) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 3) {
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = n0 + li;
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[(long)row * K + k], B[(long)col * K + k], s);
            C[(long)row * N + col] = s;
        }
        return;
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
        for (int k = 0; k < 8; ++k) { av[k] = A[(long)(m0 + li) * K + k0 + lh * 8 + k]; bv[k] = B[(long)(n0 + li) * K + k0 + lh * 8 + k]; }
        if (mode <= 1) {
            f16x8 a1, a2, b1, b2;
            for (int k = 0; k < 8; ++k) {
                const float x = av[k] * sa, w = bv[k] * sb;
                a1[k] = (_Float16)x; a2[k] = (_Float16)(x - (float)a1[k]);
                b1[k] = (_Float16)w; b2[k] = (_Float16)(w - (float)b1[k]);
            }
            if (mode == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
        } else {
            bf16x8 a[3], b[3];
            for (int k = 0; k < 8; ++k) {
                float x = av[k], w = bv[k];
                for (int p = 0; p < 3; ++p) {
                    unsigned hx = cvt_pk(x, 0.f), hw = cvt_pk(w, 0.f);
                    a[p][k] = (short)(hx & 0xffff); b[p][k] = (short)(hw & 0xffff);
                    x -= bf_lo(hx); w -= bf_lo(hw);
                }
            }
            const int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 0, 1, 1, 0};
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[PB[q]], acc, 0, 0, 0);
        }
    }
    const float inv = mode <= 1 ? 1.f / (sa * sb) : 1.f;
    for (int r = 0; r < 16; ++r) C[(long)(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n0 + li] = acc[r] * inv;
}

static float pow2_scale(const std::vector<float>& v, float target) {
    float mx = 0.f;
    for (float x : v) mx = fmaxf(mx, fabsf(x));
    return exp2f(floorf(log2f(target / mx)));
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d);
    float h = -1.f;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("denormal fp16 input through MFMA: acc = %g (0.015625 = honoured, 0 = flushed)\n", h);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f16 = 0; f16 < 2; ++f16) {
        const int iters = 20000, blocks = 256 * 2;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) hipLaunchKernelGGL(rate_kernel<true>, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(rate_kernel<false>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 16 * 6.0 * iters * blocks * 4;
        printf("%s 32x32x16 MFMA rate: %.1f TFLOP/s (%.2f ms)\n", f16 ? "f16 " : "bf16", flop / ms / 1e9, ms);
    }
    const int M = 256, N = 192, K = 1728;
    struct Case { const char* name; float amp_a; float amp_b; int heavy; } cases[] = {
        {"activations x weights", 1.f, 0.05f, 0}, {"heavy-tailed activations x weights", 1.f, 0.05f, 1}, {"gradients (1e-7) x activations", 1e-7f, 1.f, 1}};
    for (auto& c : cases) {
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        srand(7);
        auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
        for (auto& x : A) { float u = rnd(); x = c.amp_a * (c.heavy ? u * u * u * 30.f * (rnd() > 0.98f ? 8.f : 1.f) : u); }
        for (auto& x : B) x = c.amp_b * rnd() * (c.heavy ? (1.f + 3.f * fabsf(rnd())) : 1.f);
        std::vector<double> ref((size_t)M * N);
        double scale = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
                ref[(size_t)m * N + n] = s;
                scale = fmax(scale, fabs(s));
            }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, ref.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s:\n", c.name);
        for (float target : {8192.f, 1024.f, 16.f, 0.25f}) {
            const float sa = pow2_scale(A, target), sb = pow2_scale(B, target);
            for (int mode = 0; mode < 4; ++mode) {
                if (mode >= 2 && target != 8192.f) continue;
                hipLaunchKernelGGL(gemm_kernel, dim3(M / 32, N / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K, sa, sb, mode);
                std::vector<float> C(ref.size());
                hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                double emax = 0, erms = 0;
                for (size_t i = 0; i < C.size(); ++i) { double e = fabs(C[i] - ref[i]); emax = fmax(emax, e); erms += e * e; }
                const char* nm[] = {"fp16 split, 3 products", "fp16 split, 4 products", "bf16 split, 6 products", "fp32 fma chain"};
                printf("   %-24s max|x|*s = %-7g  max err / max|C| = %.3e   rms err / max|C| = %.3e\n", nm[mode], mode <= 1 ? target : 0.f, emax / scale,
                       sqrt(erms / C.size()) / scale);
            }
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
