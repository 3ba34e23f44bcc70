// Convolutions with very few input channels (the RGB stems: 3 -> N with 1x1 / 3x3 / 5x5 kernels, patch = KH*KW*Cin <= 96 values).
// The implicit-GEMM kernels waste their 32-deep K steps and 64-wide Cin tiles on them (conv_wgrad_64x64 took 351 us for the
// Cheng2020 stem where the HBM floor is ~10 us); here one thread owns one output channel, keeps its `patch` weights (forward)
// or weight-gradient accumulators (wgrad) in registers, and the im2col rows of a batch of pixels are shared through LDS.
// Both are HBM-bound streams over the [pixels][Cout] tensor.
#include "rdo_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
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
    // every lane of a wave reads the same patch value (its own weight is in a register): the reads are LDS broadcasts, and one
    // ds_read_b32 per fma made the kernel LDS-issue-bound (198 us for the 5x5 stem at 768 x 512); rows are 16-byte aligned and read
    // four values at a time
    static_assert(P % 4 == 0, "thin_fwd_kernel: patch slots in fours");
    __shared__ __attribute__((aligned(16))) float rows[PB][P];
    const int co = blockIdx.y * blockDim.x + threadIdx.x;
    const bool live = co < a.Cout;
    float wr[P];
#pragma unroll
    for (int t = 0; t < P; ++t) wr[t] = (live && t < a.patch) ? a.w[(long)co * a.patch + t] : 0.f;
    const float bv = (live && a.bias) ? a.bias[co] : 0.f;
    const int mbeg = blockIdx.x * a.mchunk, mend = min(a.M, mbeg + a.mchunk);
    for (int e = threadIdx.x; e < PB * P; e += blockDim.x) (&rows[0][0])[e] = 0.f;           // columns >= patch stay zero
    for (int m0 = mbeg; m0 < mend; m0 += PB) {
        __syncthreads();
        load_rows(a, m0, mend, rows);
        __syncthreads();
        if (!live) continue;
        for (int p = 0; p < PB && m0 + p < mend; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < P; t += 4) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(&rows[p][t]);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = fmaf(r4[k], wr[t + k], acc);
            }
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

// ---- MFMA weight gradient for 4 < patch <= 32 values (3x3x3): no LDS staging, no im2col ---------------------------------------------------
// The im2col matrix [pixels][patch] is never materialised: a lane GATHERS its element straight from the (L2-resident, 3 MB) input
// image as the MFMA operand.  v_mfma_f32_32x32x2_f32 (true fp32): D[co][j] with A = dY^T (lane = co, k = pixel: one coalesced 128-byte
// row segment per k) and B = patch values (lane = patch index j -- its tap offset is a per-lane constant --, k = pixel).
// 26 us for the RGB stem at 4 x 128^2 x 192 with 256 pixel splits (stream kernel: 52 us with 128, 115 with the old 37).  The same
// construction for the FORWARD (A = gathered patch values, B = weights in registers) was measured at 38 - 49 us against 40 for the
// stream kernel and dropped; for 1x1 kernels (patch <= 4) the stream kernels win both ways (12 / 15 us against 21 / 25).
typedef float f32x16t __attribute__((ext_vector_type(16)));

// element offset and validity of patch value j of the output pixel whose input origin is (hi0, wi0) in image b
struct PatchIdx { int dh, dw, c; };
__device__ __forceinline__ PatchIdx patch_idx(const ThinArgs& a, int j) {
    const int tap = j / a.Cin;
    PatchIdx p;
    p.c = j - tap * a.Cin;
    p.dh = tap / a.KW;
    p.dw = tap - p.dh * a.KW;
    return p;
}

// grid (nsplit); the four waves of a workgroup take quarters of the chunk's pixels, fold through LDS in wave order
template <int CT>
__global__ __launch_bounds__(256) void thin_wgrad_mfma_kernel(ThinArgs a) {
    __shared__ float fold[CT * 32 * 32];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    const int mbeg = blockIdx.x * a.mchunk, mend = min(a.M, mbeg + a.mchunk);
    const int per_wave = ((mend - mbeg + 3) / 4 + 1) & ~1;   // even: a k-step is two pixels
    const int wbeg = mbeg + wave * per_wave, wend = min(mend, wbeg + per_wave);
    const PatchIdx pj = patch_idx(a, li < a.patch ? li : 0);
    const bool jok = li < a.patch;
    const int HoWo = a.Ho * a.Wo;
    f32x16t acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    constexpr int U = 4;                                     // k-steps per trip: their loads are issued together
    for (int m0 = wbeg; m0 < wend; m0 += 2 * U) {
        float bv[U], av[U][CT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 2 * u + lh;
            const bool okm = m < wend;
            const int mm = okm ? m : 0;
            const int b = mm / HoWo, r = mm - b * HoWo;
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi = ho * a.stride - a.pad + pj.dh, wi = wo * a.stride - a.pad + pj.dw;
            const bool ok = okm && jok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            bv[u] = ok ? a.x[((long)(b * a.H + hi) * a.W + wi) * a.Cin + pj.c] : 0.f;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int co = 32 * t + li;
                av[u][t] = (okm && co < a.Cout) ? a.dy[(long)mm * a.Cout + co] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u], acc[t], 0, 0, 0);
    }
    // D[co = 32 t + row][j = li], row = (q & 3) + 8 (q >> 2) + 4 lh
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float* f = &fold[(t * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh) * 32 + li];
                    *f = (w == 0 ? 0.f : *f) + acc[t][q];
                }
        }
    }
    __syncthreads();
    float* dst = a.out + (long)blockIdx.x * a.Cout * a.patch;
    for (int e = threadIdx.x; e < CT * 32 * 32; e += 256) {
        const int co = e >> 5, j = e & 31;
        if (co < a.Cout && j < a.patch) dst[(long)co * a.patch + j] = fold[e];
    }
}

// instances: Cout <= 128 / 192 / 256
inline int thin_ct(int Cout) { return Cout <= 128 ? 4 : (Cout <= 192 ? 6 : (Cout <= 256 ? 8 : 0)); }

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
    if (thin_ct(a.Cout) && a.patch > 4 && a.patch <= 32 && rdo::tuning(rdo::T_THIN_MFMA)) {
        a.mchunk = (int)(rdo::ceil_div(rdo::ceil_div(a.M, nsplit), 8) * 8);
        const int ct = thin_ct(a.Cout);
        if (ct == 4) hipLaunchKernelGGL(thin_wgrad_mfma_kernel<4>, dim3((unsigned)nsplit), dim3(256), 0, s, a);
        else if (ct == 6) hipLaunchKernelGGL(thin_wgrad_mfma_kernel<6>, dim3((unsigned)nsplit), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(thin_wgrad_mfma_kernel<8>, dim3((unsigned)nsplit), dim3(256), 0, s, a);
        return rdo::check_launch("conv_thin_wgrad_mfma");
    }
    dim3 grid((unsigned)nsplit, (unsigned)rdo::ceil_div(a.Cout, 64));
    if (a.patch <= 4) hipLaunchKernelGGL((thin_wgrad_kernel<4, 8>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((thin_wgrad_kernel<32, 8>), grid, dim3(512), 0, s, a);
    return rdo::check_launch("conv_thin_wgrad");
}
