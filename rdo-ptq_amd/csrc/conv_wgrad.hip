// K11 (weight gradient) -- dw[co][kh][kw][ci] = sum_m dy[m][co] * x[pix(m,kh,kw)][ci] on the fp32 MFMA.
//
// This is the autograd backward of F.conv2d w.r.t. its weight (reference: `err.backward()` at
// /root/reference/task-oriented-PTQ/quantization/layer_opt.py:306 through quant_layer.py:123).
//
// GEMM view per tap: rows = co, cols = ci, reduction = output pixels m.  A workgroup owns a TCO x TCI tile of one tap
// over one chunk of pixels and writes its partial tile into slab[chunk]; slabs are summed deterministically by the fused
// AdaRound step (adaround.hip) or rdo_reduce_slabs.  Both operands are staged [32 pixels][channels] exactly as they lie in
// NHWC memory (channels contiguous), so the MFMA fragments are plain conflict-free ds_read_b32 (lane = channel).
//
// Pipeline (same idea as conv_fwd.hip): a reduction step is 16 "slots" (pixel pairs) of TM*TN MFMAs; slot j also writes
// staging quad j (data of step s+1, loaded a full step earlier) into the idle LDS buffer and re-issues its global load
// for step s+2, and prefetches the next slot's fragments, so the matrix pipe stays fed from a single wave per SIMD.
#include <utility>

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
    constexpr int YQ = TCO / 32, XQ = TCI / 32;  // staging quads per thread per step (one pixel row per thread)
    constexpr int NQ = YQ + XQ;
    static_assert(NQ <= 16, "one staging quad per slot");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                   // [2][PK][TCO]
    float* Xs = smem + 2 * PK * TCO;    // [2][PK][TCI]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wco0 = (wave >> 1) * WCO, wci0 = (wave & 1) * WCI;
    const int srow = tid >> 3, sq = tid & 7;   // staging: pixel row of the step, first quad column

    const int chunk = blockIdx.x;
    int t = blockIdx.y;
    const int tci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tco = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = tco * TCO, ci0 = tci * TCI;

    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + PK - 1) / PK : 0;
    const int HoWo = a.Ho * a.Wo;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rq[NQ];
    unsigned okmask = 0;
    // element offsets of this thread's pixel row for the step being loaded (-1: row masked / in the padding halo)
    int yoff = -1, xoff = -1;
    auto decode = [&](int s) {
        const int m = mbeg + s * PK + srow;
        if (m < mend) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int hi = ho * a.stride - a.pad + kh, wi = wo * a.stride - a.pad + kw;
            yoff = m * a.Cout;
            xoff = ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) ? ((b * a.H + hi) * a.W + wi) * a.Cin : -1;
        } else {
            yoff = xoff = -1;
        }
    };
    auto load_quad = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bool ok;
        if constexpr (j < YQ) {
            const int c = co0 + 4 * (sq + 8 * j);
            ok = yoff >= 0 && c < a.Cout;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(a.dy + (ok ? yoff + c : 0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok && c + k < a.Cout) v[k] = a.dy[yoff + c + k];
            }
        } else {
            const int c = ci0 + 4 * (sq + 8 * (j - YQ));
            ok = xoff >= 0 && c < a.Cin;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(a.x + (ok ? xoff + c : 0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok && c + k < a.Cin) v[k] = a.x[xoff + c + k];
            }
        }
        rq[j] = v;
        okmask = ok ? (okmask | (1u << j)) : (okmask & ~(1u << j));
    };
    auto store_quad = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        f32x4 v = ((okmask >> j) & 1u) ? rq[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (j < YQ) {
            *reinterpret_cast<f32x4*>(Ys + buf * PK * TCO + srow * TCO + 4 * (sq + 8 * j)) = v;
        } else {
            if (a.square_input) v = v * v;
            *reinterpret_cast<f32x4*>(Xs + buf * PK * TCI + srow * TCI + 4 * (sq + 8 * (j - YQ))) = v;
        }
    };
    auto for_each_quad = [&](auto&& f) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
        (std::make_integer_sequence<int, NQ>{});
    };

    if (nsteps > 0) {
        decode(0);
        for_each_quad([&](auto jc) { load_quad(jc); });
        for_each_quad([&](auto jc) { store_quad(jc, 0); });
        decode(nsteps > 1 ? 1 : 0);
        for_each_quad([&](auto jc) { load_quad(jc); });
    }
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const float* Y = Ys + buf * PK * TCO;
        const float* X = Xs + buf * PK * TCI;
        float fa[2][TM], fb[2][TN];
        auto read_frags = [&](int kk, int set) {
            const int prow = 2 * kk + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[set][i] = Y[prow * TCO + wco0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[set][j] = X[prow * TCI + wci0 + j * 32 + li];
        };
        read_frags(0, 0);
        decode(s + 2 < nsteps ? s + 2 : nsteps - 1);   // row offsets of the step whose loads are re-issued below
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
                 if constexpr (SL + 1 < PK / 2) read_frags(SL + 1, (SL + 1) & 1);
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SL & 1][i], fb[SL & 1][j], acc[i][j], 0, 0, 0);
                 if constexpr (SL < NQ) {
                     store_quad(std::integral_constant<int, SL>{}, buf ^ 1);
                     load_quad(std::integral_constant<int, SL>{});
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, PK / 2>{});
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
    static rdo::PerDevice attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad, %zu B LDS) failed", lds);
        attr_set.mark();
    }
    a.tiles_co = (int)rdo::ceil_div(a.Cout, TCO);
    a.tiles_ci = (int)rdo::ceil_div(a.Cin, TCI);
    dim3 grid((unsigned)a.nsplit, (unsigned)(a.KH * a.KW * a.tiles_co * a.tiles_ci));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return rdo::check_launch("conv_wgrad");
}

// ---- tile / split selection (tools/sweep_wgrad.py) -----------------------------------------------------------------------
int g_force_big = -1, g_force_ns = -1;   // tuning overrides (rdo_debug_force_wgrad_choice)

// 192x192 tiles (3x3 MFMA tiles per wave, 96 KiB LDS, one workgroup per CU) need a long pixel reduction per workgroup
// to pay off; with few output pixels the 64x64 tile (one MFMA tile per wave, 5 workgroups per CU) spreads the work wider.
inline bool one_tile_many_pixels(const rdo_conv_desc* d) {
    return d->KH == 1 && d->KW == 1 && d->Cout >= 160 && d->Cout <= 192 && d->Cin >= 160 && d->Cin <= 192 &&
           (long)d->B * d->Ho * d->Wo >= 16384;
}

inline bool big_tiles(const rdo_conv_desc* d) {
    if (g_force_big >= 0) return g_force_big != 0;
    const long M = (long)d->B * d->Ho * d->Wo;
    // a 1x1 conv has a single 192x192 tile: filling the chip with it would need hundreds of pixel splits, and every split
    // is one more slab for the AdaRound step to read back -- use the small tile (9x more tiles, 9x fewer splits) instead
    const long big_tiles_total = (long)d->KH * d->KW * rdo::ceil_div(d->Cout, 192) * rdo::ceil_div(d->Cin, 192);
    // many tiles (the 768 / 1152-channel sub-pixel convs): few pixel splits fill the chip, so short reductions still pay off
    const long min_m = big_tiles_total >= 32 ? 1024 : 4096;
    // (three tiles: the 192 -> 576 qkv Linear of the Swin blocks over 16 K tokens -- 138 us on the fp32 small tile)
    if (d->Cout >= 160 && d->Cin >= 160 && M >= min_m && big_tiles_total >= 3) return true;
    // (two tiles over >= 16 K pixels: the 192 <-> 384 MLP Linears of the Swin blocks, 117 us on the fp32 small tile)
    if (d->Cout >= 160 && d->Cin >= 160 && M >= 16384 && big_tiles_total == 2) return true;
    // ... except over >= 16 K pixels (the GDN gamma gradient of the 64^2 and 128^2 units): there the split-bf16 tile with 256 splits
    // beats the fp32 small tile even after the AdaRound step has read 256 slabs of 144 KiB (tools/bench_wgrad_1x1.py: 37 + 12 us
    // against 87 + 9 at 4 x 128^2, 19 + 12 against 27 + 9 at 4 x 64^2)
    return one_tile_many_pixels(d);
}

inline int tiles_total(const rdo_conv_desc* d) {
    const int t = big_tiles(d) ? 192 : 64;
    return d->KH * d->KW * (int)rdo::ceil_div(d->Cout, t) * (int)rdo::ceil_div(d->Cin, t);
}

}  // namespace

int rdo_launch_wgrad_x6(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, int mchunk, hipStream_t s);
bool rdo_linear_wgrad_h2_ok(const rdo_conv_desc* d, const float* x, const float* dy);                                   // conv_wgrad_h2.hip
int rdo_launch_linear_wgrad_h2(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, int mchunk, hipStream_t s);
bool rdo_conv_is_thin(const rdo_conv_desc* d, bool forward);                                                            // conv_thin.hip
int rdo_launch_thin_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, hipStream_t s);
bool rdo_conv_is_thincout(const rdo_conv_desc* d, bool forward);                                                        // conv_thincout.hip
int rdo_thincout_patches(const rdo_conv_desc* d);
int rdo_launch_thincout_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);

// 1 when rdo_conv2d_wgrad runs this shape on the split-bf16 MFMA path (conv_wgrad_x6.hip): big-tile problems whose output
// rows are a multiple of 4 pixels wide and whose channel counts allow 16-byte quads.  RDO_CONV_X6=0 disables it.
extern "C" int rdo_conv2d_wgrad_uses_bf16x6(const rdo_conv_desc* d) {
    const bool enabled = rdo::tuning(rdo::T_CONV_X6) != 0;
    if (!d || !enabled || g_force_big == 0) return 0;
    // the x6 loader addresses both tensors with 32-bit byte offsets (buffer loads)
    const bool fits = (double)d->B * d->H * d->W * d->Cin * 4.0 < 4.0e9 && (double)d->B * d->Ho * d->Wo * d->Cout * 4.0 < 4.0e9;
    return big_tiles(d) && fits && d->Wo % 4 == 0 && d->Wo >= 16 && d->Cin % 4 == 0 && d->Cout % 4 == 0;
}

extern "C" void rdo_debug_force_wgrad_choice(int big, int nsplit) {
    g_force_big = big;
    g_force_ns = nsplit;
}

extern "C" int rdo_conv2d_wgrad_nsplit(const rdo_conv_desc* d) {
    if (!d) return 1;
    if (g_force_big < 0 && rdo_conv_is_thincout(d, false)) return rdo_thincout_patches(d);      // one slab per 16 x 16 patch
    if (g_force_ns >= 1) return g_force_ns;
    const long M = (long)d->B * d->Ho * d->Wo;
    // few input channels (conv_thin.hip): one workgroup column per 64 output channels, so the pixel splits alone have to fill the chip
    if (rdo_conv_is_thin(d, false)) {
        const long per = d->KH * d->KW * d->Cin > 4 ? 256 : 512;      // pixels per split: the MFMA form (patch > 4) wants 256 workgroups
        const long cap = d->KH * d->KW * d->Cin > 4 ? 256 : 128;
        const long ns = M / per;
        return (int)(ns < 1 ? 1 : (ns > cap ? cap : ns));
    }
    // big tile: ~1 workgroup per CU; small tile: ~4 per CU.  Chunks of at least 128 pixels (4 reduction steps).
    const long target = big_tiles(d) ? 256 : 1024;
    long ns = target / tiles_total(d);
    const long max_by_m = rdo::ceil_div(M, 128);
    if (ns > max_by_m) ns = max_by_m;
    if (ns < 1) ns = 1;
    // (token-matrix gradients on linear_wgrad_h2_kernel stream 48 KB per 32-token stage and workgroup: they want every CU, 128 slabs at two tiles)
    const long cap = (g_force_big < 0 && one_tile_many_pixels(d)) ? 256 : ((g_force_big != 0 && rdo_linear_wgrad_h2_ok(d, nullptr, nullptr)) ? 128 : 64);
    if (ns > cap) ns = cap;
    return (int)ns;
}

extern "C" int rdo_conv2d_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit,
                                void* stream) {
    RDO_REQUIRE(d && x && dy && slabs, "rdo_conv2d_wgrad: null argument");
    RDO_REQUIRE(nsplit >= 1, "rdo_conv2d_wgrad: nsplit < 1");
    RDO_REQUIRE((long)d->B * d->H * d->W * d->Cin < (1L << 31) && (long)d->B * d->Ho * d->Wo * d->Cout < (1L << 31),
                "rdo_conv2d_wgrad: tensors of 2^31 or more elements are not supported (32-bit staging offsets)");
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
    if (g_force_big < 0 && rdo_conv_is_thincout(d, false) && nsplit == rdo_thincout_patches(d)) {
        const rdo_conv_desc dd = *d;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_thincout_wgrad(&dd, x, dy, slabs, s); }, stream, "conv_thincout_wgrad",
                             2.0 * a.M * (double)a.Cout * a.Cin * 9.0,
                             4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * 9.0 * a.Cin));
    }
    if (rdo_conv_is_thin(d, false)) {
        const rdo_conv_desc dd = *d;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_thin_wgrad(&dd, x, dy, slabs, nsplit, s); }, stream,
                             "conv_thin_wgrad", 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW,
                             4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * a.KH * a.KW * a.Cin));
    }
    // token-matrix gradients (1 x 1, channel counts in blocks of 192) straight from fp32 on split-fp16 MFMA with a per-stage scale
    if (g_force_big != 0 && rdo_linear_wgrad_h2_ok(d, x, dy) && (long)nsplit * a.mchunk >= a.M) {
        const rdo_conv_desc dd = *d;
        const int mchunk = a.mchunk;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_linear_wgrad_h2(&dd, x, dy, slabs, nsplit, mchunk, s); }, stream,
                             "linear_wgrad_h2", 2.0 * a.M * (double)a.Cout * a.Cin,
                             4.0 * ((double)a.M * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * a.Cin));
    }
    if (vec && rdo_conv2d_wgrad_uses_bf16x6(d)) {
        const rdo_conv_desc dd = *d;
        const int mchunk = a.mchunk;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_wgrad_x6(&dd, x, dy, slabs, nsplit, mchunk, s); }, stream,
                             "conv_wgrad_x6_192x192", 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW,
                             4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * a.KH * a.KW * a.Cin));
    }
    return rdo::dispatch(
        [a, vec, big](hipStream_t s) {
            if (big) return vec ? launch<192, 192, true>(a, s) : launch<192, 192, false>(a, s);
            return vec ? launch<64, 64, true>(a, s) : launch<64, 64, false>(a, s);
        },
        stream, big ? "conv_wgrad_192x192" : "conv_wgrad_64x64", 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW,
        4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout + (double)nsplit * a.Cout * a.KH * a.KW * a.Cin));
}
