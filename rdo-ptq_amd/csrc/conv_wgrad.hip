// K11 (weight gradient) -- dw[co][kh][kw][ci] = sum_m dy[m][co] * x[pix(m,kh,kw)][ci] on the fp32 MFMA.
//
// This is the autograd backward of F.conv2d w.r.t. its weight (reference: `err.backward()` at
// /root/reference/task-oriented-PTQ/quantization/layer_opt.py:306 through quant_layer.py:123).
//
// GEMM view per tap: rows = co, cols = ci, reduction = output pixels m.  A workgroup owns a TCO x TCI tile of one tap
// over one chunk of pixels and writes its partial tile into slab[chunk]; slabs are summed deterministically by the fused
// AdaRound step (adaround.hip) or rdo_reduce_slabs.  Both operands are staged [32 pixels][channels] exactly as they lie in
// NHWC memory (channels contiguous), so the MFMA fragments are plain conflict-free ds_read_b32 (lane = channel).
#include "rdo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PK = 32;  // pixels per reduction step

struct WgArgs {
    const float* x;
    const float* dy;
    float* slabs;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, mchunk, nsplit;
    int tiles_co, tiles_ci;
    int square_input;
};

template <int TCO, int TCI, bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
    constexpr int WCO = TCO / 2, WCI = TCI / 2;  // 2x2 waves
    constexpr int TM = WCO / 32, TN = WCI / 32;
    constexpr int YQ = PK * TCO / 4 / 256;  // dy quads per thread per step
    constexpr int XQ = PK * TCI / 4 / 256;
    static_assert(YQ >= 1 && XQ >= 1, "tile too small for 256 threads");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                   // [2][PK][TCO]
    float* Xs = smem + 2 * PK * TCO;    // [2][PK][TCI]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wco0 = (wave >> 1) * WCO, wci0 = (wave & 1) * WCI;

    const int chunk = blockIdx.x;
    int t = blockIdx.y;
    const int tci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tco = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = tco * TCO, ci0 = tci * TCI;

    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = (mend - mbeg + PK - 1) / PK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ry[YQ], rx[XQ];
    const int HoWo = a.Ho * a.Wo;

    auto load_step = [&](int s) {
        const int mb = mbeg + s * PK;
#pragma unroll
        for (int j = 0; j < YQ; ++j) {
            int e = tid + 256 * j;
            int row = e / (TCO / 4), qd = e - row * (TCO / 4);
            int m = mb + row, c = co0 + 4 * qd;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) {
                const float* p = a.dy + (long)m * a.Cout + c;
                if (VEC) {
                    if (c < a.Cout) v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k < a.Cout) v[k] = p[k];
                }
            }
            ry[j] = v;
        }
#pragma unroll
        for (int j = 0; j < XQ; ++j) {
            int e = tid + 256 * j;
            int row = e / (TCI / 4), qd = e - row * (TCI / 4);
            int m = mb + row, c = ci0 + 4 * qd;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) {
                int b = m / HoWo;
                int rem = m - b * HoWo;
                int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                int hi = ho * a.stride - a.pad + kh, wi = wo * a.stride - a.pad + kw;
                if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
                    const float* p = a.x + (((long)b * a.H + hi) * a.W + wi) * a.Cin + c;
                    if (VEC) {
                        if (c < a.Cin) v = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (c + k < a.Cin) v[k] = p[k];
                    }
                }
            }
            if (a.square_input) v = v * v;
            rx[j] = v;
        }
    };
    auto store_step = [&](int buf) {
        float* Y = Ys + buf * PK * TCO;
        float* X = Xs + buf * PK * TCI;
#pragma unroll
        for (int j = 0; j < YQ; ++j) *reinterpret_cast<f32x4*>(Y + (tid + 256 * j) * 4) = ry[j];
#pragma unroll
        for (int j = 0; j < XQ; ++j) *reinterpret_cast<f32x4*>(X + (tid + 256 * j) * 4) = rx[j];
    };

    if (nsteps > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);
        const float* Y = Ys + buf * PK * TCO;
        const float* X = Xs + buf * PK * TCI;
#pragma unroll 4
        for (int kk = 0; kk < PK / 2; ++kk) {
            const int prow = 2 * kk + lh;
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = Y[prow * TCO + wco0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = X[prow * TCI + wci0 + j * 32 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    const long wsize = (long)a.Cout * a.KH * a.KW * a.Cin;
    float* slab = a.slabs + (long)chunk * wsize;
    const int taps = a.KH * a.KW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wci0 + j * 32 + li;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.Cout) slab[((long)co * taps + tap) * a.Cin + ci] = acc[i][j][r];
            }
    }
}

template <int TCO, int TCI, bool VEC>
int launch(WgArgs a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * PK * (TCO + TCI) * sizeof(float);
    auto kern = conv_wgrad_kernel<TCO, TCI, VEC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad, %zu B LDS) failed", lds);
        attr_set = true;
    }
    a.tiles_co = (int)rdo::ceil_div(a.Cout, TCO);
    a.tiles_ci = (int)rdo::ceil_div(a.Cin, TCI);
    dim3 grid((unsigned)a.nsplit, (unsigned)(a.KH * a.KW * a.tiles_co * a.tiles_ci));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return rdo::check_launch("conv_wgrad");
}

// tile choice shared by the split heuristic and the launcher
inline bool big_tiles(const rdo_conv_desc* d) { return d->Cout >= 160 && d->Cin >= 160; }

inline int tiles_total(const rdo_conv_desc* d) {
    const int t = big_tiles(d) ? 192 : 64;
    return d->KH * d->KW * (int)rdo::ceil_div(d->Cout, t) * (int)rdo::ceil_div(d->Cin, t);
}

}  // namespace

extern "C" int rdo_conv2d_wgrad_nsplit(const rdo_conv_desc* d) {
    if (!d) return 1;
    const long M = (long)d->B * d->Ho * d->Wo;
    // aim at ~1 workgroup per CU (the big tile uses 96 KiB of LDS), chunks of at least 256 pixels
    const long target = big_tiles(d) ? 256 : 512;
    long ns = target / tiles_total(d);
    const long max_by_m = rdo::ceil_div(M, 256);
    if (ns > max_by_m) ns = max_by_m;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return (int)ns;
}

extern "C" int rdo_conv2d_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit,
                                void* stream) {
    RDO_REQUIRE(d && x && dy && slabs, "rdo_conv2d_wgrad: null argument");
    RDO_REQUIRE(nsplit >= 1, "rdo_conv2d_wgrad: nsplit < 1");
    WgArgs a;
    a.x = x; a.dy = dy; a.slabs = slabs;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.nsplit = nsplit;
    a.mchunk = (int)(rdo::ceil_div(rdo::ceil_div(a.M, nsplit), PK) * PK);
    a.square_input = d->square_input;
    a.tiles_co = a.tiles_ci = 0;
    const bool vec = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) % 16 == 0);
    const bool big = big_tiles(d);
    return rdo::dispatch(
        [a, vec, big](hipStream_t s) {
            if (big) return vec ? launch<192, 192, true>(a, s) : launch<192, 192, false>(a, s);
            return vec ? launch<64, 64, true>(a, s) : launch<64, 64, false>(a, s);
        },
        stream, big ? "conv_wgrad_192x192" : "conv_wgrad_64x64", 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW,
        4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * a.KH * a.KW * a.Cin));
}
