// Microbenchmark: how fast does a CU take LDS-DMA pieces (global_load_lds dwordx4, 1 KiB per wave instruction) as a function of the
// lane -> address pattern?  Source layout is the plane layout of the H2 tensors, [C/16][M][16] u16: a pixel's 16-channel slice is 32 B.
//   pattern 0: lane = 8 row + cpos  -> pixel row, slice cpos >> 1, half cpos & 1   (the kernels' round-4 mapping: a lane quad touches 2 lines)
//   pattern 1: lane = 16 slice + q  -> slice, 16 B piece q of 8 consecutive pixels  (256 B runs: a lane quad stays in one line)
//   pattern 2: lane = 32 slice + q  -> two slices, 16 pixels each                   (512 B runs)
//   pattern 3: 1 KiB contiguous
// build: hipcc -O3 --offload-arch=gfx950 -o dma_rate dma_rate.hip ; run: ./dma_rate [waves issuing per workgroup = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int PAT>
__global__ __launch_bounds__(512) void k(const char* src, long M, int iters, int issuing, int win, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long off;
    const long slice_stride = M * 32;
    if (PAT == 0) off = (long)((lane & 7) >> 1) * slice_stride + (lane >> 3) * 32 + (lane & 1) * 16;
    else if (PAT == 1) off = (long)(lane >> 4) * slice_stride + (lane & 15) * 16;
    else if (PAT == 2) off = (long)(lane >> 5) * slice_stride + (lane & 31) * 16;
    else off = lane * 16;
    const char* p = src + off + (long)blockIdx.x * 65536 + wave * 256;     // each workgroup its own pixel range
    char* dst = smem + wave * 5 * 1024;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < issuing) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                __builtin_amdgcn_global_load_lds((glb_void*)(p + (long)j * 4 * slice_stride), (lds_void*)(dst + j * 1024), 16, 0, 0);
            p += 8 * 32 * 8;                                                    // next 8 pixels x 8 waves
            if ((it & (win - 1)) == win - 1) p -= (long)win * 8 * 32 * 8;        // stay in a window of win x 2 KiB per slice group
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

template <int PAT>
void run(const char* src, long M, int issuing, int win, int grid, unsigned long long* dcyc) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(512), 64 * 1024, 0, src, M, 100, issuing, win, dcyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(512), 64 * 1024, 0, src, M, iters, issuing, win, dcyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < grid; ++i) avg += (double)h[i]; avg /= grid;
    const double pieces = (double)iters * 5 * issuing;
    printf("pattern %d, %d waves issuing, window %d, grid %d: %.1f us, %.0f cycles -> %.1f cycles per 1 KiB piece per CU (%.1f B/clk/CU), %.2f TB/s chip\n", PAT, issuing, win, grid, ms * 1e3, avg,
           avg / pieces, 1024.0 * pieces / avg, grid * pieces * 1024.0 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int issuing = argc > 1 ? atoi(argv[1]) : 8;
    const long M = 262144;                      // pixels
    char* src; unsigned long long* dcyc;
    hipMalloc(&src, (size_t)M * 32 * 24 + (1 << 20));
    hipMemset(src, 1, (size_t)M * 32 * 24 + (1 << 20));
    hipMalloc(&dcyc, 256 * 8);
    const int win = argc > 2 ? atoi(argv[2]) : 32, grid = argc > 3 ? atoi(argv[3]) : 256;
    run<0>(src, M, issuing, win, grid, dcyc);
    run<1>(src, M, issuing, win, grid, dcyc);
    run<3>(src, M, issuing, win, grid, dcyc);
    return 0;
}
