// K1/K2/K3 -- convolution forward (and stride-1 dgrad) as an implicit GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   out[m][n] = epilogue( sum_{kh,kw,ci} x[b, ho*s-p+kh, wo*s-p+kw, ci] * w[n][kh][kw][ci] + bias[n] )
//   m = (b*Ho + ho)*Wo + wo  (NHWC pixels),  n = output channel.
//
// Replaces F.conv2d at /root/reference/task-oriented-PTQ/quantization/quant_layer.py:123 and the 1x1
// `F.conv2d(x**2, gamma, beta)` of f_gdn (quant_layer.py:147).  fp32 in / fp32 accumulate: the MFMA result is a
// k-ordered fmaf chain, i.e. plain fp32 conv numerics (no TF32/bf16 shortcuts).
//
// Tiling (gfx950, wave64): a 256-thread workgroup owns a BM x BN output tile; the K loop walks (tap, 32-channel slice).
// Per K step the workgroup stages an A tile [BM pixels][32 ch] (gathered, zero-filled at the padding halo) and a B tile
// [BN couts][32 ch] into LDS as 16-byte quads, XOR-swizzled by (row>>1)&7 so that both the ds_write_b128 staging stores
// and the ds_read_b128 fragment loads are bank-conflict free.  Global loads for step s+1 are issued before the MFMAs of
// step s and written to the other LDS buffer afterwards (one barrier per step).  Each lane feeds 4 consecutive k of its
// row to 4 MFMAs (the k pairing between the two lane halves is arbitrary as long as A and B agree).
#include <utility>

#include "rdo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;  // channels per K step

struct FwdArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* aux;
    const float* residual;
    float* out;
    float* pre;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M;          // B*Ho*Wo
    int csteps;     // ceil(Cin/32)
    int epilogue, square_input, add_residual;
    float* partial;  // split-K: raw accumulators [ksplit][M][Cout]; nullptr when the kernel writes the final output
    int ksplit;
    int partial_only;  // split-K: leave the slabs in the workspace, no epilogue launch (rdo_conv2d_fwd_partials)
};

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__device__ __forceinline__ int swz(int row, int q) { return (q ^ ((row >> 1) & 7)) << 2; }

// bias has been added: pre-activation copy, activation, residual, store.  HEAVY: the instantiations that also carry the GELU epilogues
// (erff / expf polynomials: compiled into every kernel they cost the plain launches registers and 10-50 % of their time, measured on
// the Lu2022 linears) -- the two tiles the Swin linears select and the split-K second pass
template <bool HEAVY>
__device__ __forceinline__ void finish(const FwdArgs& a, long o, float v) {
    if (a.pre) a.pre[o] = v;
    switch (a.epilogue) {
        case RDO_EPI_LRELU: v = v > 0.f ? v : 0.01f * v; break;
        case RDO_EPI_LRELU_BWD: v = a.aux[o] > 0.f ? v : 0.01f * v; break;
        case RDO_EPI_RELU: v = v > 0.f ? v : 0.f; break;
        case RDO_EPI_RELU_BWD: v = a.aux[o] > 0.f ? v : 0.f; break;
        case RDO_EPI_GDN: v = a.aux[o] * __frsqrt_rn(v); break;
        case RDO_EPI_IGDN: v = a.aux[o] * __fsqrt_rn(v); break;
        case RDO_EPI_GELU: if constexpr (HEAVY) v = rdo::gelu(v); break;
        case RDO_EPI_GELU_BWD: if constexpr (HEAVY) v *= rdo::gelu_grad(a.aux[o]); break;
        default: break;
    }
    if (a.add_residual) v += a.residual[o];
    a.out[o] = v;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC, bool HEAVY = false>
__global__ __launch_bounds__(256) void conv_fwd_kernel(FwdArgs a) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AQ = BM / 32;  // A quads per thread per step
    constexpr int BQ = BN / 32;  // B quads per thread per step
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][32]
    float* Bs = smem + 2 * BM * BK;      // [2][BN][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int q = tid & 7, r0 = tid >> 3;  // staging: quad column, first row

    // per-thread staging rows (32-bit element offsets: tensors on this path are < 2^31 elements, checked by the host)
    int hi0[AQ], wi0[AQ], abase[AQ];
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
        int m = m0 + r0 + 32 * j;
        const bool okr = m < a.M;
        int mm = okr ? m : 0;
        int b = mm / (a.Ho * a.Wo);
        int rem = mm - b * (a.Ho * a.Wo);
        int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[j] = okr ? ho * a.stride - a.pad : -(1 << 28);   // rows past M fail the bounds test below
        wi0[j] = wo * a.stride - a.pad;
        abase[j] = b * a.H * a.W * a.Cin;
    }
    int bbase[BQ];
#pragma unroll
    for (int j = 0; j < BQ; ++j) {
        int n = n0 + r0 + 32 * j;
        bbase[j] = n < a.Cout ? n * a.KH * a.KW * a.Cin : -1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging registers: quad j < AQ belongs to the A tile, the rest to the B tile.  `okmask` bit j says whether the quad
    // held in reg j is real data (else it is a dummy read that must be zeroed when it is written to LDS).
    constexpr int NQ = AQ + BQ;
    static_assert(NQ <= 16, "one staging quad per (u,t) slot");
    f32x4 rq[NQ];
    unsigned okmask = 0;
    // split-K: this workgroup reduces K steps [sbeg, sbeg + nsteps) of the KH*KW*csteps total
    const int steps_total = a.KH * a.KW * a.csteps;
    const int sbeg = (int)((long)steps_total * blockIdx.z / a.ksplit);
    const int nsteps = (int)((long)steps_total * (blockIdx.z + 1) / a.ksplit) - sbeg;

    // issue the global load of staging quad j for K step `s` (wave-uniform tap / channel slice)
    auto load_quad = [&](auto jc, int srel) {
        constexpr int j = decltype(jc)::value;
        // K order: channel slice OUTER, taps INNER (as the bf16x6 kernels): the taps of one 32-channel slice read overlapping input
        // rows back to back, so they are served from L1 / L2 -- with taps outer every tap re-fetched its rows through the fabric
        // (g_s.7.0, 192 -> 12 at 4 x 128^2: rocprofv3 FETCH_SIZE 376 MiB for a 50 MB input)
        const int s = sbeg + srel;
        const int ktaps = a.KH * a.KW;
        const int cs = s / ktaps;
        const int tap = s - cs * ktaps;
        const int c0 = cs * BK + 4 * q;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bool ok;
        if constexpr (j < AQ) {
            const int hi = hi0[j] + kh, wi = wi0[j] + kw;
            ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W && c0 < a.Cin;
            const int off = abase[j] + (hi * a.W + wi) * a.Cin + c0;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(a.x + (ok ? off : 0));     // branch-free: dummy address when masked
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok && c0 + e < a.Cin) v[e] = a.x[off + e];
            }
        } else {
            constexpr int jb = j - AQ;
            ok = bbase[jb] >= 0 && c0 < a.Cin;
            const int off = bbase[jb] + (kh * a.KW + kw) * a.Cin + c0;
            if (VEC) {
                v = *reinterpret_cast<const f32x4*>(a.w + (ok ? off : 0));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok && c0 + e < a.Cin) v[e] = a.w[off + e];
            }
        }
        rq[j] = v;
        okmask = ok ? (okmask | (1u << j)) : (okmask & ~(1u << j));
    };
    // write staging quad j into LDS buffer `buf`
    auto store_quad = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        f32x4 v = ((okmask >> j) & 1u) ? rq[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (j < AQ) {
            if (a.square_input) v = v * v;
            const int row = r0 + 32 * j;
            *reinterpret_cast<f32x4*>(As + buf * BM * BK + row * BK + swz(row, q)) = v;
        } else {
            const int row = r0 + 32 * (j - AQ);
            *reinterpret_cast<f32x4*>(Bs + buf * BN * BK + row * BK + swz(row, q)) = v;
        }
    };
    auto for_each_quad = [&](auto&& f) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
        (std::make_integer_sequence<int, NQ>{});
    };

    // prologue: step 0 -> LDS buffer 0, step 1 -> staging registers
    for_each_quad([&](auto jc) { load_quad(jc, 0); });
    for_each_quad([&](auto jc) { store_quad(jc, 0); });
    for_each_quad([&](auto jc) { load_quad(jc, nsteps > 1 ? 1 : 0); });
    __syncthreads();

    // Main loop.  Each K step is 16 slots (u,t) of TM*TN MFMAs; slot j also (1) writes staging quad j (data of step s+1,
    // loaded one full step earlier) into the idle LDS buffer and (2) re-issues its global load for step s+2.  The non-MFMA
    // work is thus spread evenly under the MFMAs instead of forming a separate phase: with phases, the two co-resident
    // workgroups of a CU fall into lock-step (fair MFMA arbitration makes them finish their MFMA blocks together) and the
    // matrix pipe idled ~27 % of the time (rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE = 0.73 before this).
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;     // clamp: tail iterations re-load the last step (unused)
        const float* A = As + buf * BM * BK;
        const float* Bt = Bs + buf * BN * BK;
        f32x4 fa[2][TM], fb[2][TN];
        auto read_frags = [&](int u, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + li;
                fa[set][i] = *reinterpret_cast<const f32x4*>(A + row * BK + swz(row, 2 * u + lh));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + li;
                fb[set][j] = *reinterpret_cast<const f32x4*>(Bt + row * BK + swz(row, 2 * u + lh));
            }
        };
        read_frags(0, 0);
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
                 constexpr int u = SL / 4, t = SL % 4;
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][i][t], fb[u & 1][j][t], acc[i][j], 0, 0, 0);
                 if constexpr (SL < NQ) {
                     store_quad(std::integral_constant<int, SL>{}, buf ^ 1);
                     load_quad(std::integral_constant<int, SL>{}, s2);
                 }
                 if constexpr (t == 1 && u < 3) read_frags(u + 1, (u + 1) & 1);
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 16>{});
        __syncthreads();
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*lh][col = li]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + li;
        if (n >= a.Cout) continue;
        const float bv = (a.bias && !a.partial) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                const long o = (long)m * a.Cout + n;
                if (a.partial) {
                    a.partial[(long)blockIdx.z * a.M * a.Cout + o] = acc[i][j][r];
                } else {
                    finish<HEAVY>(a, o, acc[i][j][r] + bv);
                }
            }
        }
    }
}

// split-K second pass: sum the partial accumulators, then the same bias / activation / residual epilogue
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(FwdArgs a) {
    const long total = (long)a.M * a.Cout;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < a.ksplit; ++z) v += a.partial[(long)z * total + o];
        if (a.bias) v += a.bias[o % a.Cout];
        finish<true>(a, o, v);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC, bool HEAVY = false>
int launch(const FwdArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    auto kern = conv_fwd_kernel<BM, BN, WAVES_M, WAVES_N, VEC, HEAVY>;
    static rdo::PerDevice attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd, %zu B LDS) failed", lds);
        attr_set.mark();
    }
    dim3 grid((unsigned)rdo::ceil_div(a.M, BM), (unsigned)rdo::ceil_div(a.Cout, BN), (unsigned)a.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    if (int rc = rdo::check_launch("conv_fwd")) return rc;
    if (a.ksplit > 1 && !a.partial_only) {
        long g = rdo::ceil_div((long)a.M * a.Cout, 256);
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
        return rdo::check_launch("conv_splitk_epilogue");
    }
    return RDO_OK;
}

// ---- tile / split-K selection ---------------------------------------------------------------------------------------
struct Tile { int bm, bn, occ; const char* tag; };
constexpr Tile kTiles[] = {
    {128, 192, 2, "conv_fwd_128x192"}, {128, 64, 3, "conv_fwd_128x64"}, {128, 32, 4, "conv_fwd_128x32"},
    {64, 192, 2, "conv_fwd_64x192"},   {64, 64, 4, "conv_fwd_64x64"},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
struct Choice { int tile, ksplit; };
int g_force_tile = -1, g_force_ks = -1;   // tuning overrides (rdo_debug_force_fwd_choice)

// Tile / split-K rule distilled from tools/sweep_fwd.py on MI355X (all Cheng2020 shapes, B=4):
//   * 64-row tiles win everywhere: 64x192 when Cout is a multiple of 192 and that still yields >= 256 tiles, else 64x64
//     (128x32 for the 12-channel output conv); twice as many workgroups as 128-row tiles means two rounds per CU, so one
//     workgroup's tile store / cold first loads hide under another's MFMAs (128x192 single-round: 441 us vs 407 us at 128^2);
//   * split K until ~512 workgroups are in flight (2 per CU), keeping >= 6 K steps per split; partial tiles are reduced by a
//     second, L2-resident pass that also applies the epilogue (deterministic, no float atomics).
Choice choose(const FwdArgs& a, long ws_floats) {
    const int steps = a.KH * a.KW * a.csteps;
    int t;
    if (a.Cout <= 32) {
        t = 2;
    } else if (a.Cout % 192 == 0 && rdo::ceil_div(a.M, 64) * (a.Cout / 192) >= 256) {
        t = 3;
    } else {
        t = 4;
    }
    if (g_force_tile >= 0 && g_force_tile < kNumTiles) t = g_force_tile;
    const long tiles = rdo::ceil_div(a.M, kTiles[t].bm) * rdo::ceil_div(a.Cout, kTiles[t].bn);
    long ks = rdo::ceil_div(512, tiles);
    const long ks_max = steps / 6 > 1 ? steps / 6 : 1;
    if (ks > ks_max) ks = ks_max;
    if (g_force_ks >= 1) ks = g_force_ks;
    while (ks > 1 && (ks * a.M * a.Cout > ws_floats || steps / ks < 1)) --ks;
    return Choice{t, (int)ks};
}

template <bool VEC>
int launch_choice(const FwdArgs& a, int tile, hipStream_t s) {
    if (a.epilogue >= RDO_EPI_GELU && !(a.ksplit > 1 && !a.partial_only)) {      // (a split-K launch applies its epilogue in the second pass)
        if constexpr (VEC) {
            if (tile == 3) return launch<64, 192, 2, 2, true, true>(a, s);
            if (tile == 4) return launch<64, 64, 2, 2, true, true>(a, s);
        }
        return rdo::set_error(RDO_EINVAL, "rdo_conv2d_fwd: the GELU epilogues are built for the 64-row tiles with Cin %% 4 == 0 (Linear layers)");
    }
    switch (tile) {
        case 0: return launch<128, 192, 2, 2, VEC>(a, s);
        case 1: return launch<128, 64, 2, 2, VEC>(a, s);
        case 2: return launch<128, 32, 4, 1, VEC>(a, s);
        case 3: return launch<64, 192, 2, 2, VEC>(a, s);
        default: return launch<64, 64, 2, 2, VEC>(a, s);
    }
}

FwdArgs make_args(const rdo_conv_desc* d) {
    FwdArgs a{};
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.csteps = (d->Cin + BK - 1) / BK;
    a.epilogue = d->epilogue; a.square_input = d->square_input; a.add_residual = d->add_residual;
    a.ksplit = 1;
    return a;
}

}  // namespace

// tuning/debug: force the tile instance (index into kTiles, -1 = model) and split factor (-1 = model); returns the tile tag
extern "C" const char* rdo_debug_force_fwd_choice(int tile, int ksplit) {
    g_force_tile = tile;
    g_force_ks = ksplit;
    return (tile >= 0 && tile < kNumTiles) ? kTiles[tile].tag : "model";
}

extern "C" int rdo_conv2d_fwd_bf16x6(const rdo_conv_desc* d, const float* x, const void* wplanes, const float* bias, const float* aux,
                                     const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats,
                                     void* stream);
extern "C" int rdo_conv2d_fwd_bf16x6_ksplit(const rdo_conv_desc* d);

// 1 when rdo_conv2d_fwd would take the split-bf16 path for this shape if weight planes are supplied: problems that fill the
// chip with 128 x 192 tiles, split over K if necessary (rdo_conv2d_fwd_bf16x6_ksplit).  RDO_CONV_X6=0 disables it.
extern "C" int rdo_conv2d_fwd_uses_bf16x6(const rdo_conv_desc* d) {
    const bool enabled = rdo::tuning(rdo::T_CONV_X6) != 0;
    if (!d || !enabled) return 0;
    if (3.0 * d->Cout * d->KH * d->KW * d->Cin >= 2147483648.0) return 0;      // 32-bit plane offsets in the LDS-DMA loaders
    return rdo_conv2d_fwd_bf16x6_ksplit(d) >= 1;
}

// floats of scratch the caller should provide so that small problems can be split over K (0: never needed)
extern "C" int64_t rdo_conv2d_fwd_workspace(const rdo_conv_desc* d) {
    if (!d) return 0;
    const FwdArgs a = make_args(d);
    const Choice c = choose(a, 1L << 40);
    int64_t need = c.ksplit > 1 ? (int64_t)c.ksplit * a.M * a.Cout : 0;
    const int ks6 = rdo_conv2d_fwd_uses_bf16x6(d) ? rdo_conv2d_fwd_bf16x6_ksplit(d) : 0;
    if (ks6 > 1 && (int64_t)ks6 * a.M * a.Cout > need) need = (int64_t)ks6 * a.M * a.Cout;
    return need;
}

bool rdo_conv_is_thin(const rdo_conv_desc* d, bool forward);                                                            // conv_thin.hip
int rdo_launch_thin_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, float* out, hipStream_t s);
bool rdo_conv_is_thincout(const rdo_conv_desc* d, bool forward);                                                        // conv_thincout.hip
int rdo_launch_thincout_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, float* out, hipStream_t s);

int rdo_conv2d_fwd_bf16x6_partials(const rdo_conv_desc* d, const float* x, const void* wplanes, float* workspace,
                                   int64_t workspace_floats, void* stream);                                               // conv_fwd_x6.hip

// Split factor rdo_conv2d_fwd would use for this shape (1: no split-K pass, e.g. the thin kernels or enough tiles without it)
extern "C" int rdo_conv2d_fwd_ksplit(const rdo_conv_desc* d, int has_planes, int64_t workspace_floats) {
    if (!d) return 1;
    if (rdo_conv_is_thincout(d, true) || rdo_conv_is_thin(d, true)) return 1;
    const long M = (long)d->B * d->Ho * d->Wo;
    if (has_planes && rdo_conv2d_fwd_uses_bf16x6(d)) {
        const int ks = rdo_conv2d_fwd_bf16x6_ksplit(d);
        return (ks > 1 && (long)ks * M * d->Cout <= workspace_floats) ? ks : 1;
    }
    FwdArgs a = make_args(d);
    return choose(a, workspace_floats).ksplit;
}

// The K-split accumulation of rdo_conv2d_fwd WITHOUT its second pass: the raw partial sums stay in `workspace` as
// [ksplit][B*Ho*Wo][Cout] (ksplit = rdo_conv2d_fwd_ksplit(...) >= 2, else an error) for a consumer that folds "sum the slabs, add
// the bias" into its own first load -- rdo_loss_act_bwd_splitk, which is then conv epilogue, loss and activation backward in one pass.
extern "C" int rdo_conv2d_fwd_partials(const rdo_conv_desc* d, const float* x, const float* w, const void* wplanes, float* workspace,
                                       int64_t workspace_floats, void* stream) {
    RDO_REQUIRE(d && x && w && workspace, "rdo_conv2d_fwd_partials: null argument");
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd_partials: Ho/Wo do not match geometry");
    RDO_REQUIRE((long)d->B * d->H * d->W * d->Cin < (1L << 31) && (long)d->Cout * d->KH * d->KW * d->Cin < (1L << 31) &&
                    (long)d->B * d->Ho * d->Wo < (1L << 31), "rdo_conv2d_fwd_partials: tensor too large");
    const int ks = rdo_conv2d_fwd_ksplit(d, wplanes != nullptr, workspace_floats);
    RDO_REQUIRE(ks >= 2, "rdo_conv2d_fwd_partials: this shape is not split over K (rdo_conv2d_fwd_ksplit = %d)", ks);
    if (wplanes && rdo_conv2d_fwd_uses_bf16x6(d)) return rdo_conv2d_fwd_bf16x6_partials(d, x, wplanes, workspace, workspace_floats, stream);
    FwdArgs a = make_args(d);
    a.x = x; a.w = w;
    const Choice c = choose(a, workspace_floats);
    a.ksplit = c.ksplit;
    a.partial = workspace;
    a.partial_only = 1;
    const int tile = c.tile;
    const bool vec = (d->Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout * c.ksplit + (double)a.Cout * a.KH * a.KW * a.Cin);
    return rdo::dispatch([a, vec, tile](hipStream_t s) { return vec ? launch_choice<true>(a, tile, s) : launch_choice<false>(a, tile, s); },
                         stream, kTiles[tile].tag, flops, bytes);
}

extern "C" int rdo_conv2d_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, const float* aux,
                              const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats,
                              const void* wplanes, void* stream) {
    RDO_REQUIRE(d && x && w && out, "rdo_conv2d_fwd: null argument");
    RDO_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 &&
                    d->pad >= 0,
                "rdo_conv2d_fwd: bad geometry");
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    RDO_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "rdo_conv2d_fwd: too many output pixels");
    RDO_REQUIRE((long)d->B * d->H * d->W * d->Cin < (1L << 31) && (long)d->Cout * d->KH * d->KW * d->Cin < (1L << 31),
                "rdo_conv2d_fwd: tensors of 2^31 or more elements are not supported (32-bit staging offsets)");
    const int epi = d->epilogue;
    RDO_REQUIRE(epi >= RDO_EPI_NONE && epi <= RDO_EPI_GELU_BWD, "rdo_conv2d_fwd: unknown epilogue %d", epi);
    RDO_REQUIRE(epi == RDO_EPI_NONE || epi == RDO_EPI_LRELU || epi == RDO_EPI_RELU || epi == RDO_EPI_GELU || aux != nullptr,
                "rdo_conv2d_fwd: epilogue %d needs aux", epi);
    RDO_REQUIRE(!d->add_residual || residual != nullptr, "rdo_conv2d_fwd: add_residual without residual");
    if (rdo_conv_is_thincout(d, true) && pre == nullptr) {
        const rdo_conv_desc dd = *d;
        const double M = (double)d->B * d->Ho * d->Wo;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_thincout_fwd(&dd, x, w, bias, out, s); }, stream, "conv_thincout_fwd",
                             2.0 * M * d->Cout * d->Cin * 9.0, 4.0 * ((double)d->B * d->H * d->W * d->Cin + M * d->Cout));
    }
    if (rdo_conv_is_thin(d, true) && pre == nullptr) {
        const rdo_conv_desc dd = *d;
        const double M = (double)d->B * d->Ho * d->Wo;
        return rdo::dispatch([=](hipStream_t s) { return rdo_launch_thin_fwd(&dd, x, w, bias, out, s); }, stream, "conv_thin_fwd",
                             2.0 * M * d->Cout * d->Cin * d->KH * d->KW, 4.0 * ((double)d->B * d->H * d->W * d->Cin + M * d->Cout));
    }
    if (wplanes && rdo_conv2d_fwd_uses_bf16x6(d))
        return rdo_conv2d_fwd_bf16x6(d, x, wplanes, bias, aux, residual, out, pre, workspace, workspace_floats, stream);
    FwdArgs a = make_args(d);
    a.x = x; a.w = w; a.bias = bias; a.aux = aux; a.residual = residual; a.out = out; a.pre = pre;
    const Choice c = choose(a, workspace ? workspace_floats : 0);
    a.ksplit = c.ksplit;
    a.partial = c.ksplit > 1 ? workspace : nullptr;
    const int tile = c.tile;
    const bool vec = (d->Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout * (1 + (pre != nullptr) + (aux != nullptr) +
                                (residual != nullptr)) + (double)a.Cout * a.KH * a.KW * a.Cin);
    return rdo::dispatch([a, vec, tile](hipStream_t s) { return vec ? launch_choice<true>(a, tile, s) : launch_choice<false>(a, tile, s); },
                         stream, kTiles[tile].tag, flops, bytes);
}
