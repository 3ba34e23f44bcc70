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
};

__device__ __forceinline__ int swz(int row, int q) { return (q ^ ((row >> 1) & 7)) << 2; }

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC>
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

    // per-thread A rows
    int hi0[AQ], wi0[AQ];
    long abase[AQ];
    bool arow_ok[AQ];
#pragma unroll
    for (int j = 0; j < AQ; ++j) {
        int m = m0 + r0 + 32 * j;
        arow_ok[j] = m < a.M;
        int mm = arow_ok[j] ? m : 0;
        int b = mm / (a.Ho * a.Wo);
        int rem = mm - b * (a.Ho * a.Wo);
        int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[j] = ho * a.stride - a.pad;
        wi0[j] = wo * a.stride - a.pad;
        abase[j] = (long)b * a.H * a.W * a.Cin;
    }
    long bbase[BQ];
    bool brow_ok[BQ];
#pragma unroll
    for (int j = 0; j < BQ; ++j) {
        int n = n0 + r0 + 32 * j;
        brow_ok[j] = n < a.Cout;
        bbase[j] = (long)(brow_ok[j] ? n : 0) * a.KH * a.KW * a.Cin;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[AQ], rb[BQ];
    const int nsteps = a.KH * a.KW * a.csteps;

    auto load_step = [&](int s) {
        int tap = s / a.csteps;
        int c0 = (s - tap * a.csteps) * BK + 4 * q;
        int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
        for (int j = 0; j < AQ; ++j) {
            int hi = hi0[j] + kh, wi = wi0[j] + kw;
            bool ok = arow_ok[j] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const float* p = a.x + abase[j] + ((long)hi * a.W + wi) * a.Cin + c0;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                if (ok && c0 < a.Cin) v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ok && c0 + e < a.Cin) v[e] = p[e];
            }
            if (a.square_input) v = v * v;
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BQ; ++j) {
            const float* p = a.w + bbase[j] + (long)(kh * a.KW + kw) * a.Cin + c0;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                if (brow_ok[j] && c0 < a.Cin) v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (brow_ok[j] && c0 + e < a.Cin) v[e] = p[e];
            }
            rb[j] = v;
        }
    };
    auto store_step = [&](int buf) {
        float* A = As + buf * BM * BK;
        float* Bt = Bs + buf * BN * BK;
#pragma unroll
        for (int j = 0; j < AQ; ++j) {
            int row = r0 + 32 * j;
            *reinterpret_cast<f32x4*>(A + row * BK + swz(row, q)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BQ; ++j) {
            int row = r0 + 32 * j;
            *reinterpret_cast<f32x4*>(Bt + row * BK + swz(row, q)) = rb[j];
        }
    };

    load_step(0);
    store_step(0);
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_step(s + 1);
        const float* A = As + buf * BM * BK;
        const float* Bt = Bs + buf * BN * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int row = wm0 + i * 32 + li;
                fa[i] = *reinterpret_cast<const f32x4*>(A + row * BK + swz(row, 2 * u + lh));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int row = wn0 + j * 32 + li;
                fb[j] = *reinterpret_cast<const f32x4*>(Bt + row * BK + swz(row, 2 * u + lh));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*lh][col = li]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + li;
        if (n >= a.Cout) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                const long o = (long)m * a.Cout + n;
                float v = acc[i][j][r] + bv;
                if (a.pre) a.pre[o] = v;
                switch (a.epilogue) {
                    case RDO_EPI_LRELU: v = v > 0.f ? v : 0.01f * v; break;
                    case RDO_EPI_LRELU_BWD: v = a.aux[o] > 0.f ? v : 0.01f * v; break;
                    case RDO_EPI_GDN: v = a.aux[o] * __frsqrt_rn(v); break;
                    case RDO_EPI_IGDN: v = a.aux[o] * __fsqrt_rn(v); break;
                    default: break;
                }
                if (a.add_residual) v += a.residual[o];
                a.out[o] = v;
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC>
int launch(const FwdArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    auto kern = conv_fwd_kernel<BM, BN, WAVES_M, WAVES_N, VEC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd, %zu B LDS) failed", lds);
        attr_set = true;
    }
    dim3 grid((unsigned)rdo::ceil_div(a.M, BM), (unsigned)rdo::ceil_div(a.Cout, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return rdo::check_launch("conv_fwd");
}

// tile instance for a shape: 0 -> 128x32, 1 -> 128x64, 2 -> 128x192
int pick_tile(const FwdArgs& a) {
    if (a.Cout <= 32) return 0;
    if (a.Cout <= 64) return 1;
    // tiles of 192 output channels fit the N=192 family (192, 384, 768, 1152); otherwise 64-wide tiles waste less.
    const long waste192 = rdo::ceil_div(a.Cout, 192) * 192 - a.Cout;
    const long waste64 = rdo::ceil_div(a.Cout, 64) * 64 - a.Cout;
    const long tiles192 = rdo::ceil_div(a.M, 128) * rdo::ceil_div(a.Cout, 192);   // small problems: prefer more workgroups
    return (waste192 <= waste64 && tiles192 >= 192) ? 2 : 1;
}

template <bool VEC>
int launch_by_shape(const FwdArgs& a, hipStream_t s) {
    switch (pick_tile(a)) {
        case 0: return launch<128, 32, 4, 1, VEC>(a, s);
        case 1: return launch<128, 64, 2, 2, VEC>(a, s);
        default: return launch<128, 192, 2, 2, VEC>(a, s);
    }
}

}  // namespace

extern "C" int rdo_conv2d_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, const float* aux,
                              const float* residual, float* out, float* pre, void* stream) {
    RDO_REQUIRE(d && x && w && out, "rdo_conv2d_fwd: null argument");
    RDO_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 &&
                    d->pad >= 0,
                "rdo_conv2d_fwd: bad geometry");
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    RDO_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "rdo_conv2d_fwd: too many output pixels");
    const int epi = d->epilogue;
    RDO_REQUIRE(epi >= RDO_EPI_NONE && epi <= RDO_EPI_IGDN, "rdo_conv2d_fwd: unknown epilogue %d", epi);
    RDO_REQUIRE(epi < RDO_EPI_LRELU_BWD || aux != nullptr, "rdo_conv2d_fwd: epilogue %d needs aux", epi);
    RDO_REQUIRE(!d->add_residual || residual != nullptr, "rdo_conv2d_fwd: add_residual without residual");
    FwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.aux = aux; a.residual = residual; a.out = out; a.pre = pre;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.csteps = (d->Cin + BK - 1) / BK;
    a.epilogue = epi; a.square_input = d->square_input; a.add_residual = d->add_residual;
    const bool vec = (d->Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
    static const char* kTags[3] = {"conv_fwd_128x32", "conv_fwd_128x64", "conv_fwd_128x192"};
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout * (1 + (pre != nullptr) + (aux != nullptr) +
                                (residual != nullptr)) + (double)a.Cout * a.KH * a.KW * a.Cin);
    return rdo::dispatch([a, vec](hipStream_t s) { return vec ? launch_by_shape<true>(a, s) : launch_by_shape<false>(a, s); },
                         stream, kTags[pick_tile(a)], flops, bytes);
}
