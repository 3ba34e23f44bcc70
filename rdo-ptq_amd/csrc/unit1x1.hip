// One launch for the data path of a 1 x 1 LAYER unit's calibration iteration (round 6):
//
//     pre = x W~^T + b          (QuantModule.forward of a 1x1 Conv2d with soft-rounded weights, quant_layer.py:113-123)
//     out = act(pre)            (the activation fused into the unit, quant_layer.py:100,126-128)
//     L  += coef * lp_loss(out, target[idx])        -> loss log        (layer_opt.py:133,150; quantizer.py:71-79)
//     dpre = act'(pre) * dL/dout
//     dW~ partial = dpre^T x    over this workgroup's tokens -> slab   (autograd's weight gradient of quant_layer.py:123)
//
// i.e. what rdo_conv2d_fwd(_partials) + rdo_loss_act_bwd(_splitk) + rdo_conv2d_wgrad do in three launches for the 1x1 192 <-> 96 convs of
// Cheng2020-attn's attention blocks (76 of BASELINE config 3's 105 units are such layer units; at 16^2 = 1 K pixels each of the three
// launches is a ramp, ONE K step per workgroup of the split-K conv, and a round trip of partial sums).  Exact fp32 arithmetic
// (v_mfma_f32_32x32x2_f32: an fmaf chain per accumulator), so no plane scales and no overflow words.
//
// A workgroup (4 waves) owns 32 tokens x 96 output channels (grid.y walks the 96-blocks of Cout) and walks T token tiles: per tile the
// 32 x K input rows and -- once -- the 96 x K weight block go to LDS; waves 0-2 multiply one 32 x 32 output tile each (K / 2 MFMAs), apply
// bias / activation / loss / activation backward on the accumulators (the arithmetic of loss_act_quad, fused_tail.hip, element by
// element) and leave dpre in LDS; then all four waves multiply the 3 x (K / 32) tiles of dpre^T x over the tile's 32 tokens into
// accumulators that live across the T tiles, and store them as this chunk's slab.  Needs K <= 192, K % 32 == 0, Cout % 32 == 0, tokens % 32 == 0.
#include "rdo_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 32;      // tokens per tile
constexpr int NB = 96;      // output channels per workgroup
constexpr int KMAX = 192;

struct U1Args {
    const float* x;         // [M][K] mini-batch (NHWC with C = K)
    const float* w;         // [N][K] soft-rounded weights (kernel layout of a 1x1 conv)
    const float* bias;      // [N] or nullptr
    const float* tgt;       // target cache [n_img][ppi][N]
    const int32_t* idx;     // [iters][B]
    const int32_t* iter;
    int32_t* pub;           // rdo_iter_bind_publish
    float* slabs;           // [nchunk][N][K]
    float* loss;            // [iters][RDO_LOG_SLOTS]
    int M, K, N, B, ppi, act, T;
    float gs, loss_scale;
};

__global__ __launch_bounds__(256) void unit1x1_kernel(U1Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = a.K, S = K + 4;                            // row stride: 16-byte fragment reads of 32 consecutive rows are conflict free
    float* Xs = smem;                                        // [2][BM][S]   the current tile and the next one (loaded by wave 3 under the forward)
    float* Ws = Xs + 2 * BM * S;                             // [NB][S]
    float* Ds = Ws + NB * S;                                 // [BM][NB + 4]   dL/dpre of the tile
    constexpr int SD = NB + 4;
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int chunk = blockIdx.x, nb0 = blockIdx.y * NB;
    const int nrows = min(NB, a.N - nb0);                    // 96, or 32 / 64 in the last block
    const int it = *a.iter;
    if (a.pub && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.pub = it;

    const int kq = K >> 2;
    const int mfirst = chunk * a.T * BM;
    // the weight block (once) and the first token tile
    // Eight lanes per row, each taking the quads (lane & 7) + 8 j of its row: no index division, and every request of the block is issued
    // before the first LDS store (the plain "load, store" loop compiled to 18 serial round trips: load, s_waitcnt vmcnt(0), ds_write).
    const int kq8 = kq >> 3;                                 // quads per lane and row: 1 .. 6
    {
        const int r0 = tid >> 3, c0 = (tid & 7) * 4;
        f32x4 wv[3][6], xv[6];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                wv[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < kq8 && 32 * p + r0 < nrows) wv[p][j] = *reinterpret_cast<const f32x4*>(a.w + (long)(nb0 + 32 * p + r0) * K + c0 + 32 * j);
            }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < kq8) xv[j] = *reinterpret_cast<const f32x4*>(a.x + (long)(mfirst + r0) * K + c0 + 32 * j);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < kq8) *reinterpret_cast<f32x4*>(Ws + (32 * p + r0) * S + c0 + 32 * j) = wv[p][j];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < kq8) *reinterpret_cast<f32x4*>(Xs + r0 * S + c0 + 32 * j) = xv[j];
    }

    // weight-gradient accumulators: tiles t = wave, wave + 4, ... of the 3 x (K / 32) grid (row block t % 3, column block t / 3)
    const int ktiles = K >> 5, ntile = 3 * ktiles;
    f32x16 gacc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[u][r] = 0.f;

    const float slope = a.act == 1 ? 0.01f : 0.f;
    float lsum = 0.f;
    __syncthreads();
    for (int tt = 0; tt < a.T; ++tt) {
        const int m0 = mfirst + tt * BM;
        if (m0 >= a.M) break;
        const float* Xc = Xs + (tt & 1) * BM * S;
        if (wave == 3) {
            // the next tile goes into the other buffer while waves 0-2 multiply this one (its last readers finished behind the barrier
            // that closed iteration tt - 1)
            const int m1 = m0 + BM;
            if (tt + 1 < a.T && m1 < a.M) {
                float* Xn = Xs + ((tt + 1) & 1) * BM * S;
                const int r1 = lane >> 3, c1 = (lane & 7) * 4;          // eight rows per pass, four passes, in two halves of <= 12 requests
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4 pv[2][6];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int j = 0; j < 6; ++j)
                            if (j < kq8) pv[p][j] = *reinterpret_cast<const f32x4*>(a.x + (long)(m1 + 8 * (2 * half + p) + r1) * K + c1 + 32 * j);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int j = 0; j < 6; ++j)
                            if (j < kq8) *reinterpret_cast<f32x4*>(Xn + (8 * (2 * half + p) + r1) * S + c1 + 32 * j) = pv[p][j];
                }
            }
        } else if (wave * 32 < nrows) {
            // ---- forward: waves 0-2, one 32 x 32 tile of the 32 x 96 block each.  Lane half lh takes k = 8 j + 4 lh + {0..3} of its row for
            // four consecutive MFMAs (the assignment of k to the two lane halves is free as long as both operands agree): one 16-byte read
            // per operand.  The 16 target values of the lane's accumulator elements are requested first: their latency hides under the MFMAs.
            const int n = nb0 + wave * 32 + li;
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int b = m / a.ppi;
                y[r] = a.tgt[((long)a.idx[(long)it * a.B + b] * a.ppi + (m - b * a.ppi)) * a.N + n];
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* xa = Xc + li * S + 4 * lh;
            const float* wb = Ws + (wave * 32 + li) * S + 4 * lh;
            for (int j = 0; j < (K >> 3); ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(xa + 8 * j), bv = *reinterpret_cast<const f32x4*>(wb + 8 * j);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
            // acc[r]: token (r & 3) + 8 (r >> 2) + 4 lh of the tile, channel n
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tk = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float p = acc[r] + bv;
                float o = p;
                if (a.act) o = p > 0.f ? p : slope * p;
                const float dd = o - y[r];
                const float g = dd * a.gs;
                Ds[tk * SD + wave * 32 + li] = a.act ? (p > 0.f ? g : slope * g) : g;
                lsum += dd * dd;
            }
        } else {                                              // rows past Cout in the last block: no gradient
#pragma unroll
            for (int r = 0; r < 16; ++r) Ds[((r & 3) + 8 * (r >> 2) + 4 * lh) * SD + wave * 32 + li] = 0.f;
        }
        __syncthreads();                                     // dpre of the tile is complete (and the next tile has landed)
        // ---- weight gradient of the tile: dW[n][k] += sum over the 32 tokens dpre[token][n] * x[token][k]; lane half lh takes tokens 2 q + lh
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = wave + 4 * u;
            if (t >= ntile) break;
            const int rb = t % 3, cb = t / 3;
            const float* da = Ds + lh * SD + rb * 32 + li;
            const float* xb = Xc + lh * S + cb * 32 + li;
#pragma unroll
            for (int q = 0; q < BM / 2; ++q)
                gacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[2 * q * SD], xb[2 * q * S], gacc[u], 0, 0, 0);
        }
        __syncthreads();                                     // Ds and this X buffer are free for iteration tt + 1 / tt + 2
    }
    // ---- this chunk's slab
    float* slab = a.slabs + (long)chunk * a.N * K;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int t = wave + 4 * u;
        if (t >= ntile) break;
        const int rb = t % 3, cb = t / 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (nl < nrows) slab[(long)(nb0 + nl) * K + cb * 32 + li] = gacc[u][r];
        }
    }
    // ---- loss: one atomic per workgroup (32 log slots per iteration)
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o, 64);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0 && a.loss)
        atomicAdd(a.loss + (long)it * RDO_LOG_SLOTS + ((blockIdx.x + gridDim.x * blockIdx.y) & (RDO_LOG_SLOTS - 1)),
                  (red[0] + red[1] + red[2] + red[3]) * a.loss_scale);
}

}  // namespace

extern "C" {

int rdo_unit1x1_supported(int64_t M, int32_t K, int32_t N) {
    return (M > 0 && M % BM == 0 && K >= 32 && K <= KMAX && K % 32 == 0 && N >= 32 && N % 32 == 0 && M * (int64_t)K < (1LL << 31) &&
            M * (int64_t)N < (1LL << 31))
               ? 1
               : 0;
}

// token tiles per workgroup: as few as keep the launch at one workgroup per CU (256 CUs; the kernel's LDS allows one per CU)
static int tiles_per_wg(int64_t M, int32_t N) {
    const int64_t tiles = M / BM, nblk = (N + NB - 1) / NB;
    const int64_t T = (tiles * nblk + 255) / 256;
    return (int)(T < 1 ? 1 : T);
}

// slabs a launch writes: one per chunk of T token tiles (<= 256; the AdaRound step sums them)
int rdo_unit1x1_nslab(int64_t M, int32_t N) {
    const int64_t tiles = M / BM, T = tiles_per_wg(M, N);
    return (int)((tiles + T - 1) / T);
}

int rdo_unit1x1(const float* x, int64_t M, int32_t K, int32_t N, const float* w, const float* bias, const float* tgt_cache,
                const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, float coef, int32_t act, float* slabs, int32_t nslab,
                float* loss_out, void* stream) {
    RDO_REQUIRE(x && w && tgt_cache && idx_table && iter_ptr && slabs, "rdo_unit1x1: null pointer");
    RDO_REQUIRE(rdo_unit1x1_supported(M, K, N), "rdo_unit1x1: shape %ld x %d -> %d is not supported (tokens %% 32, 32 <= K <= 192, K and N in blocks of 32)",
                (long)M, K, N);
    RDO_REQUIRE(B > 0 && M % B == 0, "rdo_unit1x1: %ld tokens do not split over a mini-batch of %d", (long)M, B);
    RDO_REQUIRE(act >= 0 && act <= 2, "rdo_unit1x1: act %d (0 none, 1 LeakyReLU(0.01), 2 ReLU)", act);
    RDO_REQUIRE(nslab == rdo_unit1x1_nslab(M, N), "rdo_unit1x1: %d slabs given, rdo_unit1x1_nslab says %d", nslab, rdo_unit1x1_nslab(M, N));
    U1Args a;
    a.x = x; a.w = w; a.bias = bias; a.tgt = tgt_cache; a.idx = idx_table; a.iter = iter_ptr; a.pub = rdo::take_iter_publish();
    a.slabs = slabs; a.loss = loss_out;
    a.M = (int)M; a.K = K; a.N = N; a.B = B; a.ppi = (int)(M / B); a.act = act;
    a.T = tiles_per_wg(M, N);
    const float inv_npix = (float)(1.0 / (double)M);
    a.gs = coef * 2.f * inv_npix;
    a.loss_scale = inv_npix * coef;
    const size_t lds = sizeof(float) * ((size_t)(2 * BM + NB) * (K + 4) + (size_t)BM * (NB + 4));
    const double flops = 4.0 * (double)M * K * N;
    const double bytes = 4.0 * ((double)M * K + (double)M * N + (double)N * K * (1.0 + nslab));
    return rdo::dispatch(
        [a, lds, nslab](hipStream_t s) {
            static rdo::PerDevice attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(unit1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(float) * ((size_t)(2 * BM + NB) * (KMAX + 4) + (size_t)BM * (NB + 4)))) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(unit1x1) failed");
                attr.mark();
            }
            dim3 grid((unsigned)nslab, (unsigned)rdo::ceil_div(a.N, NB));
            hipLaunchKernelGGL(unit1x1_kernel, grid, dim3(256), lds, s, a);
            return rdo::check_launch("unit1x1");
        },
        stream, "unit1x1", flops, bytes);
}

}  // extern "C"
