// 3 x 3 / stride 1 / pad 1 convolutions with very few OUTPUT channels (Cout <= 16: the 192 -> 12 sub-pixel last layer of the synthesis
// transform, reference compressai/layers subpel_conv3x3 as wrapped by quant_block.py).  The implicit-GEMM kernels pad Cout to their 32- /
// 64-wide N tiles: the forward ran 98 us and the weight gradient 180 us at 4 x 128^2 x 192 for 2.7 GFLOP and 53 MB of traffic each.
// Here N is ONE 16-wide MFMA column (v_mfma_f32_16x16x4_f32: true fp32 products and sums, no operand split needed), a workgroup owns a
// 16 x 16 pixel patch, and the 18 x 18 halo of the patch is staged once per channel slice and serves all nine taps.
//
//   forward   D[pixel][n] += x[pixel + tap][c] * w[n][tap][c]      A = activations (16 pixels x 4 channels), B = weights
//   wgrad     D[n][c]     += dy[pixel][n] * x[pixel + tap][c]      A = dy^T (16 n x 4 pixels), B = activations (4 pixels x 16 channels)
//
// K order inside one b128 fragment read: lane group g = lane / 16 reads channels 4g .. 4g+3 of its pixel with ONE 16-byte LDS read;
// MFMA j of the four that consume it takes element j, i.e. its k index stands for channel 4k + j -- the weight fragment is gathered
// with the same permutation, so the product is unchanged.
#include <utility>

#include "rdo_common.h"

int rdo_thincout_patches(const rdo_conv_desc* d);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PT = 16, HWD = PT + 2, HPIX = HWD * HWD;      // patch edge, halo edge, halo pixels

struct TcArgs {
    const float* x;        // [B][H][W][Cin]
    const float* w;        // forward: [Cout][3][3][Cin]
    const float* bias;
    const float* dy;       // wgrad: [B][H][W][Cout]
    float* out;            // forward: [B][H][W][Cout]; wgrad: slabs [patches][Cout][3][3][Cin]
    int B, H, W, Cin, Cout, epilogue;
};

struct Patch { int b, h0, w0; };
__device__ __forceinline__ Patch patch_of(const TcArgs& a, int id) {
    const int pw = a.W / PT, ph = a.H / PT;
    Patch p;
    p.b = id / (ph * pw);
    const int r = id - p.b * (ph * pw);
    p.h0 = (r / pw) * PT;
    p.w0 = (r % pw) * PT;
    return p;
}

// ---- forward ------------------------------------------------------------------------------------------------------------------------
// PH = patch height: 16, or 8 when 16-row patches would leave the chip with one wave per SIMD (a lone wave cannot hide its own LDS
// and global latency: 44 us against 2x the MFMA time at 4 x 128^2 x 192)
constexpr int FS = 20;                                      // halo row stride in floats (16 channels + 4): conflict-free b128 reads
template <int PH>
__global__ __launch_bounds__(256) void thincout_fwd_kernel(TcArgs a) {
    constexpr int HP = (PH + 2) * HWD;                       // halo pixels
    constexpr int NQ = (HP * 4 + 255) / 256;                 // halo quads per thread and slice
    constexpr int RW = PH / 4;                               // patch rows (MFMA row tiles) per wave
    __shared__ __attribute__((aligned(16))) float halo[2][HP * FS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int pw = a.W / PT, ph = a.H / PH;
    const int pb = blockIdx.x / (ph * pw);
    const int prem = blockIdx.x - pb * (ph * pw);
    const int h0 = (prem / pw) * PH, w0 = (prem % pw) * PT;
    const int csteps = a.Cin / 16;

    int src[NQ];                                            // element offset of the quad at channel slice 0, -1 outside the image
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = tid + 256 * i;
        const int q = e >> 2, c4 = e & 3;
        const int hr = q / HWD, hc = q - hr * HWD;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
        const bool ok = e < HP * 4 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        src[i] = ok ? ((pb * a.H + hi) * a.W + wi) * a.Cin + c4 * 4 : -1;
    }
    f32x4 stage[NQ];
    auto load_halo = [&](int cs) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            stage[i] = src[i] >= 0 ? *reinterpret_cast<const f32x4*>(a.x + src[i] + cs * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int e = tid + 256 * i;
            if (e < HP * 4) *reinterpret_cast<f32x4*>(&halo[buf][(e >> 2) * FS + (e & 3) * 4]) = stage[i];
        }
    };
    // weight fragments of a slice: lane (n = li, g) holds w[n][tap][16 cs + 4g .. + 3]
    f32x4 bw[9], bw_next[9];
    auto load_w = [&](int cs, f32x4 (&dst)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            dst[t] = li < a.Cout ? *reinterpret_cast<const f32x4*>(a.w + ((long)li * 9 + t) * a.Cin + cs * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    f32x4 acc[RW];
#pragma unroll
    for (int t = 0; t < RW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_halo(0);
    load_w(0, bw);
    store_halo(0);
    __syncthreads();
    for (int cs = 0; cs < csteps; ++cs) {
        const int buf = cs & 1;
        const bool more = cs + 1 < csteps;
        if (more) {
            load_halo(cs + 1);
            load_w(cs + 1, bw_next);
        }
        const float* hb = halo[buf];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int t = 0; t < RW; ++t) {                   // patch row RW wave + t = 16 pixels = one MFMA row tile
                const int q = (RW * wave + t + tap / 3) * HWD + li + tap % 3;
                const f32x4 av = *reinterpret_cast<const f32x4*>(hb + q * FS + g * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bw[tap][j], acc[t], 0, 0, 0);
            }
        }
        if (more) {
            store_halo(buf ^ 1);                            // the other buffer: last read in slice cs - 1, before the barrier below
#pragma unroll
            for (int t = 0; t < 9; ++t) bw[t] = bw_next[t];
        }
        __syncthreads();
    }
    // D[pixel = 4 g + r][n = li]
    if (li < a.Cout) {
        const float bv = a.bias ? a.bias[li] : 0.f;
#pragma unroll
        for (int t = 0; t < RW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = ((long)pb * a.H + h0 + RW * wave + t) * a.W + w0 + 4 * g + r;
                float v = acc[t][r] + bv;
                if (a.epilogue == RDO_EPI_LRELU) v = v > 0.f ? v : 0.01f * v;
                else if (a.epilogue == RDO_EPI_RELU) v = v > 0.f ? v : 0.f;
                a.out[m * a.Cout + li] = v;
            }
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------------
// grid (patches, Cin / (16 NT)); the four waves split the patch's pixels (four patch rows each), every wave accumulates all 9 NT
// output tiles [16 n x 16 c]; the four partial sums are folded through LDS in wave order (no atomics: the slab is bit-reproducible)
template <int NT>
__global__ __launch_bounds__(256) void thincout_wgrad_kernel(TcArgs a) {
    constexpr int XS = 16 * NT;                              // halo row stride in floats: XS mod 64 in {16, 32, 48} keeps the four pixel
                                                             // groups of a b32 fragment read on different banks (NT = 1, 2 or 3)
    extern __shared__ __attribute__((aligned(16))) float tc_smem[];
    float* const xs = tc_smem;                               // halo tile [324][XS] (>= 9 NT 256 floats); afterwards the fold buffer
    float* const dys = tc_smem + HPIX * XS;                  // dy of the patch, [pixel][16], zero for n >= Cout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const Patch p = patch_of(a, blockIdx.x);
    const int c0 = blockIdx.y * XS;

    // every global load of the tile is issued before the first LDS store (a load-store loop would pay the memory latency per trip)
    constexpr int QPR = XS / 4;                              // quads per halo pixel
    constexpr int NQ = (HPIX * QPR + 255) / 256;
    f32x4 hq[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = tid + 256 * i;
        const int q = e / QPR, c4 = e - q * QPR;
        const int hr = q / HWD, hc = q - hr * HWD;
        const int hi = p.h0 - 1 + hr, wi = p.w0 - 1 + hc;
        hq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < HPIX * QPR && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
            hq[i] = *reinterpret_cast<const f32x4*>(a.x + ((long)(p.b * a.H + hi) * a.W + wi) * a.Cin + c0 + c4 * 4);
    }
    // dy: a patch row is 16 Cout contiguous floats; thread -> (row = tid / 16, pixel = tid % 16)
    float dv[16];
    {
        const long m = ((long)p.b * a.H + p.h0 + (tid >> 4)) * a.W + p.w0 + (tid & 15);
#pragma unroll
        for (int n = 0; n < 16; ++n) dv[n] = n < a.Cout ? a.dy[m * a.Cout + n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = tid + 256 * i;
        if (e < HPIX * QPR) *reinterpret_cast<f32x4*>(&xs[(e / QPR) * XS + (e % QPR) * 4]) = hq[i];
    }
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4)
        *reinterpret_cast<f32x4*>(&dys[tid * 16 + n4 * 4]) = f32x4{dv[4 * n4], dv[4 * n4 + 1], dv[4 * n4 + 2], dv[4 * n4 + 3]};
    __syncthreads();

    f32x4 acc[9][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int k4 = 0; k4 < 16; ++k4) {                        // four consecutive pixels of patch row 4 wave + k4 / 4
        const int row = 4 * wave + (k4 >> 2), col = (k4 & 3) * 4 + g;
        const float av = dys[(row * 16 + col) * 16 + li];   // A[n = li][k = g]
        const int q0 = row * HWD + col;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* xr = xs + (q0 + (t / 3) * HWD + (t % 3)) * XS + li;
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xr[c * 16], acc[t][c], 0, 0, 0);   // B[k = g][c = li]
        }
    }
    // fold: wave 0 stores, waves 1..3 add in order; fold[(tile)][n = 4 g + r][c = li]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* f = &xs[((t * NT + c) * 16 + 4 * g + r) * 16 + li];
                        *f = (w == 0 ? 0.f : *f) + acc[t][c][r];
                    }
        }
    }
    __syncthreads();
    float* slab = a.out + (long)blockIdx.x * a.Cout * 9 * a.Cin;
    for (int e = tid; e < 9 * NT * 256; e += 256) {
        const int tile = e >> 8, n = (e >> 4) & 15, c = e & 15;
        const int t = tile / NT, ct = tile - t * NT;
        if (n < a.Cout) slab[((long)n * 9 + t) * a.Cin + c0 + ct * 16 + c] = xs[e];
    }
}

TcArgs make(const rdo_conv_desc* d) {
    TcArgs a{};
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.epilogue = d->epilogue;
    return a;
}

}  // namespace

// shapes these kernels take: 3x3 / stride 1 / pad 1, Cout <= 16, Cin a multiple of 16, H and W multiples of 16
bool rdo_conv_is_thincout(const rdo_conv_desc* d, bool forward) {
    if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cout >= 1 && d->Cout <= 16 && d->Cin % 16 == 0 && d->Cin >= 16 &&
          d->H % PT == 0 && d->W % PT == 0 && d->Ho == d->H && d->Wo == d->W && !d->square_input))
        return false;
    if ((double)d->B * d->H * d->W * d->Cin >= 2147483648.0) return false;
    if (!forward) return rdo_thincout_patches(d) <= 1024;
    return !d->add_residual && (d->epilogue == RDO_EPI_NONE || d->epilogue == RDO_EPI_LRELU || d->epilogue == RDO_EPI_RELU);
}

int rdo_thincout_patches(const rdo_conv_desc* d) { return d->B * (d->H / PT) * (d->W / PT); }

int rdo_launch_thincout_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, float* out, hipStream_t s) {
    TcArgs a = make(d);
    a.x = x; a.w = w; a.bias = bias; a.out = out;
    const int patches = rdo_thincout_patches(d);
    if (patches < 1024) hipLaunchKernelGGL(thincout_fwd_kernel<8>, dim3((unsigned)(2 * patches)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(thincout_fwd_kernel<16>, dim3((unsigned)patches), dim3(256), 0, s, a);
    return rdo::check_launch("conv_thincout_fwd");
}

// slabs: [patches][Cout][3][3][Cin] -- one slab per 16 x 16 patch (rdo_conv2d_wgrad_nsplit reports the count)
int rdo_launch_thincout_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    TcArgs a = make(d);
    a.x = x; a.dy = dy; a.out = slabs;
    const int slices = a.Cin / 16;
    const unsigned patches = (unsigned)rdo_thincout_patches(d);
    auto go = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;
        constexpr size_t lds = (size_t)(HPIX * 16 * NT + 256 * 16) * sizeof(float);
        static rdo::PerDevice attr;
        if (!attr.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(thincout_wgrad_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(thincout_wgrad) failed");
            attr.mark();
        }
        hipLaunchKernelGGL(thincout_wgrad_kernel<NT>, dim3(patches, (unsigned)(slices / NT)), dim3(256), lds, s, a);
        return rdo::check_launch("conv_thincout_wgrad");
    };
    if (slices % 3 == 0) return go(std::integral_constant<int, 3>{});
    if (slices % 2 == 0) return go(std::integral_constant<int, 2>{});
    return go(std::integral_constant<int, 1>{});
}
