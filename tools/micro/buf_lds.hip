// Does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset fails the range check, and
// what does a piece cost to issue?  (The conv kernels select a zero-page pointer per lane for padding today: ~6 VALU per piece.)
// build: hipcc -O3 --offload-arch=gfx950 -o buf_lds buf_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

__global__ __launch_bounds__(512) void check(const char* src, unsigned nbytes, int soff, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 256; i += 512) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = wave * 1024 + lane * 16;
    if ((lane % 5) == 0) voff = 0xFFFFFFFFu;                       // "padding" lanes
    if (lane == 63) voff = nbytes - 8;                             // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + wave * 1024), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 256; i += 512) out[i] = ((unsigned*)smem)[i];
}

template <int BUF>
__global__ __launch_bounds__(512) void rate(const char* src, unsigned nbytes, long M, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long slice_stride = M * 32;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = (unsigned)((lane >> 4) * slice_stride + (lane & 15) * 16 + blockIdx.x * 65536 + wave * 256);
    const char* p = src + voff;
    char* dst = smem + wave * 5 * 1024;
    unsigned soff = 0;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + j * 1024), 16, (lane & 7) == it ? 0xFFFFFFFFu : voff, soff + j * 4 * (unsigned)slice_stride, 0, 0);
                else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + soff + (long)j * 4 * slice_stride), (lds_void*)(dst + j * 1024), 16, 0, 0);
            }
            soff = (it & 31) == 31 ? 0 : soff + 2048;
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
    const long M = 262144;
    const size_t total = (size_t)M * 32 * 24;
    char* src; unsigned* out; unsigned long long* dcyc;
    hipMalloc(&src, total + (1 << 20)); hipMalloc(&out, 8 * 1024); hipMalloc(&dcyc, 256 * 8);
    std::vector<unsigned> h(total / 4 > (1 << 20) ? (1 << 20) : total / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const unsigned nbytes = 8 * 1024 + 4096;                        // the descriptor covers 12 KiB
    const int soff = 4096;
    hipLaunchKernelGGL(check, dim3(1), dim3(512), 8 * 1024, 0, src, nbytes, soff, out);
    std::vector<unsigned> o(2048);
    hipMemcpy(o.data(), out, 8192, hipMemcpyDeviceToHost);
    int bad = 0, zeros = 0, kept = 0;
    for (int w = 0; w < 8; ++w)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                const unsigned got = o[w * 256 + l * 4 + d];
                unsigned want = (unsigned)((soff + w * 1024 + l * 16) / 4 + d);
                const bool oob = (l % 5) == 0;
                if (l == 63) { if (w == 0 && d == 0) printf("straddling lane: %08x %08x %08x %08x (in-range dwords would be %08x %08x)\n", o[l*4], o[l*4+1], o[l*4+2], o[l*4+3], (nbytes - 8 + soff) / 4, (nbytes - 8 + soff) / 4 + 1); continue; }
                if (oob) { if (got == 0) ++zeros; else if (got == 0xdeadbeefu) ++kept; else ++bad; }
                else if (got != want) ++bad;
            }
    printf("check: out-of-range lanes wrote zero %d dwords, left LDS untouched %d, wrong %d\n", zeros, kept, bad);
    for (int b = 0; b < 2; ++b) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto go = [&](int iters) {
            if (b) hipLaunchKernelGGL(rate<1>, dim3(32), dim3(512), 64 * 1024, 0, src, 0xFFFFFF00u, M, iters, dcyc);
            else hipLaunchKernelGGL(rate<0>, dim3(32), dim3(512), 64 * 1024, 0, src, 0xFFFFFF00u, M, iters, dcyc);
        };
        go(100); hipEventRecord(e0); go(2000); hipEventRecord(e1); hipDeviceSynchronize();
        unsigned long long hc[32]; hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 32; ++i) avg += (double)hc[i]; avg /= 32;
        printf("%s: %.1f cycles per 1 KiB piece per CU (4 waves issuing, 32 workgroups)\n", b ? "buffer_load lds (select per piece)" : "global_load_lds", avg / (2000.0 * 5 * 4));
    }
    return 0;
}
