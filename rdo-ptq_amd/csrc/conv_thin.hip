// Convolutions with very few input channels (the RGB stems: 3 -> N with 1x1 / 3x3 / 5x5 kernels, patch = KH*KW*Cin <= 96 values).
// The implicit-GEMM kernels waste their 32-deep K steps and 64-wide Cin tiles on them (conv_wgrad_64x64 took 351 us for the
// Cheng2020 stem where the HBM floor is ~10 us); here one thread owns one output channel, keeps its `patch` weights (forward)
// or weight-gradient accumulators (wgrad) in registers, and the im2col rows of a batch of pixels are shared through LDS.
// Both are HBM-bound streams over the [pixels][Cout] tensor.
#include "rdo_common.h"

namespace {

constexpr int PMAX = 96;     // max KH*KW*Cin
constexpr int PB = 32;       // pixels per LDS batch

struct ThinArgs {
    const float* x;
    const float* w;        // fwd: [Cout][patch]
    const float* bias;
    const float* dy;       // wgrad
    float* out;            // fwd: [M][Cout]; wgrad: slabs [nsplit][Cout][patch]
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, patch, epilogue, mchunk;
};

// im2col rows of pixels [m0, m0 + PB) -> rows[p][t], zero outside the image / past mend
template <int STRIDE>
__device__ __forceinline__ void load_rows(const ThinArgs& a, int m0, int mend, float (*rows)[STRIDE]) {
    for (int e = threadIdx.x; e < PB * a.patch; e += blockDim.x) {
        const int p = e / a.patch, t = e - p * a.patch;
        const int m = m0 + p;
        float v = 0.f;
        if (m < mend) {
            const int tap = t / a.Cin, c = t - tap * a.Cin;
            const int kh = tap / a.KW, kw = tap - kh * a.KW;
            const int b = m / (a.Ho * a.Wo);
            const int r = m - b * (a.Ho * a.Wo);
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi = ho * a.stride - a.pad + kh, wi = wo * a.stride - a.pad + kw;
            if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) v = a.x[((long)(b * a.H + hi) * a.W + wi) * a.Cin + c];
        }
        rows[p][t] = v;
    }
}

// grid: (pixel chunks, ceil(Cout / blockDim.x)); thread = output channel
template <int P>
__global__ __launch_bounds__(256) void thin_fwd_kernel(ThinArgs a) {
    __shared__ float rows[PB][P + 1];
    const int co = blockIdx.y * blockDim.x + threadIdx.x;
    const bool live = co < a.Cout;
    float wr[P];
#pragma unroll
    for (int t = 0; t < P; ++t) wr[t] = (live && t < a.patch) ? a.w[(long)co * a.patch + t] : 0.f;
    const float bv = (live && a.bias) ? a.bias[co] : 0.f;
    const int mbeg = blockIdx.x * a.mchunk, mend = min(a.M, mbeg + a.mchunk);
    for (int e = threadIdx.x; e < PB * (P + 1); e += blockDim.x) (&rows[0][0])[e] = 0.f;     // columns >= patch stay zero
    for (int m0 = mbeg; m0 < mend; m0 += PB) {
        __syncthreads();
        load_rows(a, m0, mend, rows);
        __syncthreads();
        if (!live) continue;
        for (int p = 0; p < PB && m0 + p < mend; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < P; ++t) acc = fmaf(rows[p][t], wr[t], acc);
            acc += bv;
            if (a.epilogue == RDO_EPI_LRELU) acc = acc > 0.f ? acc : 0.01f * acc;
            else if (a.epilogue == RDO_EPI_RELU) acc = acc > 0.f ? acc : 0.f;
            a.out[(long)(m0 + p) * a.Cout + co] = acc;
        }
    }
}

// grid: (nsplit, ceil(Cout / 64)); block = G groups of 64 channels: group g takes every G-th pixel batch, partial sums are folded
// through LDS at the end (no float atomics: the slab is bit-reproducible)
template <int P, int G>
__global__ __launch_bounds__(64 * G) void thin_wgrad_kernel(ThinArgs a) {
    __shared__ float rows[G][PB][P + 1];
    __shared__ float fold[64][P + 1];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int co = blockIdx.y * 64 + lane;
    const bool live = co < a.Cout;
    float acc[P];
#pragma unroll
    for (int t = 0; t < P; ++t) acc[t] = 0.f;
    const int mbeg = blockIdx.x * a.mchunk, mend = min(a.M, mbeg + a.mchunk);
    for (int m0 = mbeg + grp * PB; m0 < mend; m0 += G * PB) {
        // each group stages its own batch (its 64 threads only): no block-wide barrier inside the loop
        for (int e = lane; e < PB * a.patch; e += 64) {
            const int p = e / a.patch, t = e - p * a.patch;
            const int m = m0 + p;
            float v = 0.f;
            if (m < mend) {
                const int tap = t / a.Cin, c = t - tap * a.Cin;
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                const int b = m / (a.Ho * a.Wo);
                const int r = m - b * (a.Ho * a.Wo);
                const int ho = r / a.Wo, wo = r - ho * a.Wo;
                const int hi = ho * a.stride - a.pad + kh, wi = wo * a.stride - a.pad + kw;
                if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) v = a.x[((long)(b * a.H + hi) * a.W + wi) * a.Cin + c];
            }
            rows[grp][p][t] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the wave's LDS stores have landed (a group is one wave)
        if (live) {
            float g[PB];                                     // the batch's 32 gradient loads are issued together (latency-bound otherwise)
#pragma unroll
            for (int p = 0; p < PB; ++p) g[p] = m0 + p < mend ? a.dy[(long)(m0 + p) * a.Cout + co] : 0.f;
#pragma unroll 4
            for (int p = 0; p < PB; ++p) {
#pragma unroll
                for (int t = 0; t < P; ++t) acc[t] = fmaf(g[p], rows[grp][p][t], acc[t]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // fold the G groups in a fixed order
    for (int g = 0; g < G; ++g) {
        __syncthreads();
        if (grp == g) {
#pragma unroll
            for (int t = 0; t < P; ++t) fold[lane][t] = (g == 0 ? 0.f : fold[lane][t]) + acc[t];
        }
    }
    __syncthreads();
    if (grp == 0 && live) {
        float* dst = a.out + ((long)blockIdx.x * a.Cout + co) * a.patch;
        for (int t = 0; t < a.patch; ++t) dst[t] = fold[lane][t];
    }
}

ThinArgs make(const rdo_conv_desc* d) {
    ThinArgs a{};
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.patch = d->KH * d->KW * d->Cin;
    a.epilogue = d->epilogue;
    return a;
}

}  // namespace

// shapes the thin kernels take over: few input channels, no squared input; forward additionally: plain / (leaky-)ReLU epilogue
bool rdo_conv_is_thin(const rdo_conv_desc* d, bool forward) {
    if (!(d->Cin <= 4 && d->KH * d->KW * d->Cin <= PMAX && !d->square_input)) return false;
    // wgrad keeps one accumulator per patch value in registers: 3x3x3 and 1x1x3 only (the 75-value 5x5 stem spilled: 3.4 ms
    // against 0.35 ms on the MFMA kernel)
    if (!forward) return d->KH * d->KW * d->Cin <= 32;
    return !d->add_residual && (d->epilogue == RDO_EPI_NONE || d->epilogue == RDO_EPI_LRELU || d->epilogue == RDO_EPI_RELU);
}

int rdo_launch_thin_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, float* out, hipStream_t s) {
    ThinArgs a = make(d);
    a.x = x; a.w = w; a.bias = bias; a.out = out;
    const int threads = a.Cout >= 256 ? 256 : (a.Cout > 128 ? 192 : (a.Cout > 64 ? 128 : 64));
    const int cblocks = (int)rdo::ceil_div(a.Cout, threads);
    long chunks = 2048 / cblocks;
    const long maxc = rdo::ceil_div(a.M, PB);
    if (chunks > maxc) chunks = maxc;
    a.mchunk = (int)(rdo::ceil_div(rdo::ceil_div(a.M, chunks), PB) * PB);
    dim3 grid((unsigned)rdo::ceil_div(a.M, a.mchunk), (unsigned)cblocks);
    if (a.patch <= 4) hipLaunchKernelGGL(thin_fwd_kernel<4>, grid, dim3(threads), 0, s, a);
    else if (a.patch <= 32) hipLaunchKernelGGL(thin_fwd_kernel<32>, grid, dim3(threads), 0, s, a);
    else hipLaunchKernelGGL(thin_fwd_kernel<PMAX>, grid, dim3(threads), 0, s, a);
    return rdo::check_launch("conv_thin_fwd");
}

int rdo_launch_thin_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, hipStream_t s) {
    ThinArgs a = make(d);
    a.x = x; a.dy = dy; a.out = slabs;
    a.mchunk = (int)(rdo::ceil_div(rdo::ceil_div(a.M, nsplit), PB) * PB);
    dim3 grid((unsigned)nsplit, (unsigned)rdo::ceil_div(a.Cout, 64));
    if (a.patch <= 4) hipLaunchKernelGGL((thin_wgrad_kernel<4, 8>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((thin_wgrad_kernel<32, 8>), grid, dim3(512), 0, s, a);
    return rdo::check_launch("conv_thin_wgrad");
}
