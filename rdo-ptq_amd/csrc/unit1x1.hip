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
#include <algorithm>
#include <type_traits>

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

// ---- split-fp16 form (round 6) -------------------------------------------------------------------------------------------------------
// The fp32 MFMAs above are the kernel's time at 64^2 (6.5-7.4 us per token tile measured by varying T: 96 + 80 v_mfma_f32_32x32x2_f32 of
// 64 cycles per wave).  This form runs both GEMMs on v_mfma_f32_16x16x32_f16 in the H2 arithmetic of linear_h2.hip (x s = h1 + h2 in fp16,
// three products h1 g1 + h1 g2 + h2 g1, fp32 accumulate) with scales computed on the fly -- nothing probed, nothing that can overflow:
//   * weights: one power-of-two scale per OUTPUT CHANNEL (row of W); the wave's 3 x (K / 32) fragments of both planes stay in registers for
//     all T token tiles (A operand = weights: a lane's four accumulator values are four consecutive channels of one token);
//   * inputs: one scale per TOKEN (as rdo_linear_h2): split once per tile into LDS, row-major for the forward and transposed for the weight
//     gradient;
//   * dL/dpre: the token's input scale is divided out of its row first (exact: a power of two), then ONE scale per tile (the block's
//     largest magnitude): dW~ += (dpre / sx)^T (x sx) needs no per-token factor behind the sum over tokens.
// Forward: wave w owns tokens [16 (w & 1), +16) x channels [48 (w >> 1), +48): 3 tiles, K / 32 steps, 9 MFMAs per step.  Weight gradient:
// wave w owns the 16-column blocks w, w + 4, w + 8 of K x all six 16-row blocks of the 96 channels (18 tiles of ONE 32-token step).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int ST = BM + 8;   // halfs per row of the transposed tiles (32 tokens + 16 bytes: fragment reads of 16 consecutive rows are conflict free)

__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }
// exponent of the power-of-two scale that puts a largest magnitude `amax` into [2^7, 2^8); 0 where there is nothing to keep
__device__ __forceinline__ int scale_exp(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu) - 127;
    return e < -100 ? 0 : 7 - e;
}
__device__ __forceinline__ float amax4(const f32x4& v, float m) {
    return fmaxf(fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))), m);
}
// workgroup barrier for LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for the tile prefetched for the NEXT iteration
// (measured: 3.4 us per token tile with everything but the loads and the barriers switched off)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned pk2(_Float16 a, _Float16 b) { return __builtin_bit_cast(unsigned, f16x2{a, b}); }
// Transposed stores two tokens at a time.  Lane `odd` and its partner (DPP control `CTRL`: the lane whose token is the neighbour) each hold
// the values v0 v1 | v2 v3 of ONE token for four consecutive rows; after one exchange the even lane owns rows 0, 1 and the odd lane rows
// 2, 3 for BOTH tokens: w0 / w1 = (even token, odd token) of the lane's two rows -- two ds_write_b32 instead of four ds_write_b16.
template <int CTRL>
__device__ __forceinline__ void pair_rows(unsigned p01, unsigned p23, bool odd, unsigned& w0, unsigned& w1) {
    const unsigned send = odd ? p01 : p23;
    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, CTRL, 0xf, 0xf, false);
    const unsigned keep = odd ? p23 : p01;
    const unsigned ev = odd ? recv : keep, od = odd ? keep : recv;
    w0 = (ev & 0xffffu) | (od << 16);
    w1 = (ev >> 16) | (od & 0xffff0000u);
}
__device__ __forceinline__ void split1(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

template <int NKS>        // K / 32: 3 or 6 (compile-time loop bounds: guarded loops became branches, serial loads and 800 AGPR moves)
__global__ __launch_bounds__(256) void unit1x1_h2_kernel(U1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_h2[];
    constexpr int K = 32 * NKS, SX = K + 8;                           // halfs per row of the row-major tile
    constexpr int xh_halfs = 2 * BM * SX, buf_halfs = xh_halfs + 2 * K * ST;
    _Float16* const base = reinterpret_cast<_Float16*>(smem_h2);   // two buffers of { Xh [2][BM][SX], XT [2][K][ST] }
    _Float16* const DT = base + 2 * buf_halfs;               // [2][NB][ST]: (dpre / sx) sd, transposed
    float* const scl = reinterpret_cast<float*>(DT + 2 * NB * ST);   // [2][BM]: 1 / sx of the buffer's tokens
    float* const red = scl + 2 * BM;                         // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, kg = lane >> 4;
    const int tt = wave & 1, cg = wave >> 1;
    const int chunk = blockIdx.x, nb0 = blockIdx.y * NB;
    const int it = *a.iter;
    if (a.pub && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.pub = it;
    const int mfirst = chunk * a.T * BM;
    const int r0 = tid >> 3, c0 = (tid & 7) * 4;             // loader role: row r0, quads at c0 + 32 j

    f32x4 xv[NKS];
    auto request = [&](int m0) {
#pragma unroll
        for (int j = 0; j < NKS; ++j) xv[j] = *reinterpret_cast<const f32x4*>(a.x + (long)(m0 + r0) * K + c0 + 32 * j);
    };
    // the loader's row: largest magnitude over its eight lanes, split by the token's scale into both layouts of buffer `b`
    auto deposit = [&](int b) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < NKS; ++j) m = amax4(xv[j], m);
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        const int e = scale_exp(m);
        const float s = pow2f(e);
        _Float16* xh = base + b * buf_halfs;
        _Float16* xt = xh + xh_halfs;
        if ((tid & 7) == 0) scl[b * BM + r0] = pow2f(-e);
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            f16x4 h, l;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                _Float16 hh, ll;
                split1(xv[j][e4] * s, hh, ll);
                h[e4] = hh; l[e4] = ll;
            }
            *reinterpret_cast<f16x4*>(xh + r0 * SX + c0 + 32 * j) = h;
            *reinterpret_cast<f16x4*>(xh + (BM + r0) * SX + c0 + 32 * j) = l;
            // transposed: rows k, tokens (r0 & ~1, r0 | 1) as one dword at position r0 >> 1, its 16-byte group XORed with (k >> 4) & 3
            // (lanes q and q + 4 of a row write rows 16 apart = the same bank otherwise); row_ror:8 = the lane of the neighbouring token
            const bool odd = (r0 & 1) != 0;
            const int k = c0 + 32 * j + (odd ? 2 : 0), tp = r0 >> 1;
            const int pos = ((((tp >> 2) ^ (k >> 4)) & 3) << 2) | (tp & 3);
            unsigned w0, w1;
            pair_rows<0x128>(pk2(h[0], h[1]), pk2(h[2], h[3]), odd, w0, w1);
            unsigned* xt32 = reinterpret_cast<unsigned*>(xt);
            xt32[k * (ST / 2) + pos] = w0;
            xt32[(k + 1) * (ST / 2) + pos] = w1;
            pair_rows<0x128>(pk2(l[0], l[1]), pk2(l[2], l[3]), odd, w0, w1);
            xt32[(K + k) * (ST / 2) + pos] = w0;
            xt32[(K + k + 1) * (ST / 2) + pos] = w1;
        }
    };

    // Every global load inside the tile loop is unconditional and independent of other loads (the index row sits in LDS, the bias in
    // registers, the tiles behind the last one are re-requests of the last): straight-line code whose waits the compiler can count --
    // with a dependent index load, guarded requests and bias loads in the loop every tile began with s_waitcnt vmcnt(0), i.e. paid the
    // full latency of the tile prefetched for the next iteration (2 us per tile with everything else switched off).
    __shared__ int sidx[256];
    if (tid < a.B) sidx[tid] = a.idx[(long)it * a.B + tid];
    const int ntw = min(a.T, (a.M - mfirst) / BM);           // token tiles of this workgroup (>= 1)
    auto tile_m = [&](int t) { return mfirst + min(t, ntw - 1) * BM; };
    f32x4 yn[3];
    auto request_y = [&](int m0, bool first = false) {      // (first: before the index row is in LDS)
        const int m = m0 + 16 * tt + l16, bimg = m / a.ppi;
        const int img = first ? a.idx[(long)it * a.B + bimg] : sidx[bimg];
        const float* yrow = a.tgt + ((long)img * a.ppi + (m - bimg * a.ppi)) * a.N + nb0 + 48 * cg + 4 * kg;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) yn[ct] = *reinterpret_cast<const f32x4*>(yrow + 16 * ct);
    };
    request(mfirst);
    request_y(mfirst, true);
    f32x4 bvr[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        bvr[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) bvr[ct] = *reinterpret_cast<const f32x4*>(a.bias + nb0 + 48 * cg + 16 * ct + 4 * kg);
    }
    // ---- the wave's weight fragments: rows nb0 + 48 cg + 16 ct + l16, k = 32 ks + 8 kg + {0..7}; one scale per row
    f16x8 wf[3][NKS][2];
    float isw[3][4];
    {
        f32x4 wr[3][NKS][2];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                    const float* src = a.w + (long)(nb0 + 48 * cg + 16 * ct + l16) * K + 32 * ks + 8 * kg;
                    wr[ct][ks][0] = *reinterpret_cast<const f32x4*>(src);
                    wr[ct][ks][1] = *reinterpret_cast<const f32x4*>(src + 4);
                }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            float m = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) m = amax4(wr[ct][ks][1], amax4(wr[ct][ks][0], m));
            m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
            const int e = scale_exp(m);
            const float s = pow2f(e), inv = pow2f(-e);
#pragma unroll
            for (int i = 0; i < 4; ++i) isw[ct][i] = __shfl(inv, 4 * kg + i, 64);      // the accumulator rows of this lane: channels 4 kg + i
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    _Float16 hh, ll;
                    split1(wr[ct][ks][e8 >> 2][e8 & 3] * s, hh, ll);
                    wf[ct][ks][0][e8] = hh; wf[ct][ks][1][e8] = ll;
                }
            }
        }
    }
    deposit(0);
    if (ntw > 1) request(tile_m(1));
    lds_barrier();                                           // the index row and tile 0

    constexpr int ktiles = K >> 4, KU = (ktiles + 3) / 4;      // 16-column blocks of K; per wave at most KU of them
    f32x4 gacc[KU][6];
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) gacc[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float slope = a.act == 1 ? 0.01f : 0.f;
    float lsum = 0.f;
    int de = 0;                                              // exponent of the accumulators' running scale
    // one token tile; `last_c`: the workgroup's last (nothing behind it to request or deposit -- at 16^2 the only one)
    auto tile = [&](int ti, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int cur = ti & 1;
        const _Float16* xh = base + cur * buf_halfs;
        const _Float16* xt = xh + xh_halfs;
        // ---- forward.  The targets of this tile were requested behind the previous tile's tail (one workgroup per CU and one wave per
        // SIMD: nothing else would hide their latency -- the products of a tile are 0.4 us).
        const int tok = 16 * tt + l16;
        f32x4 acc[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const _Float16* xrow = xh + tok * SX + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(xrow + 32 * ks), b2 = *reinterpret_cast<const f16x8*>(xrow + BM * SX + 32 * ks);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][ks][0], b1, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][ks][0], b2, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][ks][1], b1, acc[ct], 0, 0, 0);
            }
        }
        // ---- tail on the accumulators (the arithmetic of the fp32 form, element by element); dpre leaves with the token's input scale divided out
        const float isx = scl[cur * BM + tok];
        float dmax = 0.f;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = acc[ct][i] * (isw[ct][i] * isx) + bvr[ct][i];
                float o = p;
                if (a.act) o = p > 0.f ? p : slope * p;
                const float dd = o - yn[ct][i];
                const float g = dd * a.gs;
                const float d = (a.act ? (p > 0.f ? g : slope * g) : g) * isx;
                acc[ct][i] = d;
                dmax = fmaxf(dmax, __builtin_fabsf(d));
                lsum += dd * dd;
            }
        }
        if constexpr (!LAST) request_y(tile_m(ti + 1));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
        if (lane == 0) red[wave] = dmax;
        lds_barrier();                                       // (B) the tile's largest |dpre / sx|
        // The gradient accumulators carry ONE running scale 2^de: a tile keeps it while its largest magnitude lands in [2^0, 2^15) under it
        // (fp16 split: 22 bits down to 2^-3 of that, absolute 2^-25 below), else the accumulators move to the tile's own scale (exact).
        {
            const float tmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const int et = scale_exp(tmax), pos = 7 - et + de;            // floor(log2(tmax 2^de))
            if (ti == 0) de = et;
            else if (tmax > 0.f && (pos < 0 || pos > 14)) {
                const int d = et - de;
                const float r1 = pow2f(d / 2), r2 = pow2f(d - d / 2);
#pragma unroll
                for (int u = 0; u < KU; ++u)
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) gacc[u][nt][i] = gacc[u][nt][i] * r1 * r2;
                de = et;
            }
        }
        const float sd = pow2f(de);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            _Float16 hh[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) split1(acc[ct][i] * sd, hh[i], ll[i]);
            // rows 4 kg + {0..3}, tokens (tok & ~1, tok | 1) as one dword (quad_perm [1,0,3,2] = the lane of the neighbouring token)
            const bool odd = (l16 & 1) != 0;
            const int nl = 48 * cg + 16 * ct + 4 * kg + (odd ? 2 : 0), tp = tok >> 1;
            unsigned w0, w1;
            unsigned* dt32 = reinterpret_cast<unsigned*>(DT);
            pair_rows<0xB1>(pk2(hh[0], hh[1]), pk2(hh[2], hh[3]), odd, w0, w1);
            dt32[nl * (ST / 2) + tp] = w0;
            dt32[(nl + 1) * (ST / 2) + tp] = w1;
            pair_rows<0xB1>(pk2(ll[0], ll[1]), pk2(ll[2], ll[3]), odd, w0, w1);
            dt32[(NB + nl) * (ST / 2) + tp] = w0;
            dt32[(NB + nl + 1) * (ST / 2) + tp] = w1;
        }
        // the next tile (requested one iteration ago) goes into the other buffer; the one behind it is requested
        if constexpr (!LAST) {
            deposit(cur ^ 1);
            request(tile_m(ti + 2));
        }
        lds_barrier();                                       // (C) DT complete, the next tile in place
        // ---- weight gradient of the tile: D[n][k] = sum over the 32 tokens DT[n][t] XT[k][t]
        f16x8 bx[KU][2];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kt = (wave + 4 * u < ktiles) ? wave + 4 * u : wave;          // (a wave without a u-th block re-reads its first)
            bx[u][0] = *reinterpret_cast<const f16x8*>(xt + (16 * kt + l16) * ST + 8 * ((kg ^ kt) & 3));
            bx[u][1] = *reinterpret_cast<const f16x8*>(xt + (K + 16 * kt + l16) * ST + 8 * ((kg ^ kt) & 3));
        }
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const f16x8 d1 = *reinterpret_cast<const f16x8*>(DT + (16 * nt + l16) * ST + 8 * kg);
            const f16x8 d2 = *reinterpret_cast<const f16x8*>(DT + (NB + 16 * nt + l16) * ST + 8 * kg);
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                if (ktiles % 4 != 0 && wave + 4 * u >= ktiles) break;
                gacc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, bx[u][0], gacc[u][nt], 0, 0, 0);
                gacc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, bx[u][1], gacc[u][nt], 0, 0, 0);
                gacc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2, bx[u][0], gacc[u][nt], 0, 0, 0);
            }
        }
        // (no barrier here: the next writes to `red`, DT and this X buffer all come behind barrier (B) of the next iteration)
    };
    for (int ti = 0; ti + 1 < ntw; ++ti) tile(ti, std::false_type{});
    tile(ntw - 1, std::true_type{});
    // ---- this chunk's slab: lane holds rows 16 nt + 4 kg + i, column 16 kt + l16
    float* slab = a.slabs + (long)chunk * a.N * K;
    const float inv_run = pow2f(-de);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kt = wave + 4 * u;
        if (ktiles % 4 != 0 && kt >= ktiles) break;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[(long)(nb0 + 16 * nt + 4 * kg + i) * K + 16 * kt + l16] = gacc[u][nt][i] * inv_run;
    }
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o, 64);
    __syncthreads();                                         // `red` was the amax exchange
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0 && a.loss)
        atomicAdd(a.loss + (long)it * RDO_LOG_SLOTS + ((blockIdx.x + gridDim.x * blockIdx.y) & (RDO_LOG_SLOTS - 1)),
                  (red[0] + red[1] + red[2] + red[3]) * a.loss_scale);
}

size_t h2_lds_bytes(int K) {
    return sizeof(_Float16) * (2 * (size_t)(2 * BM * (K + 8) + 2 * K * ST) + 2 * (size_t)NB * ST) + sizeof(float) * (2 * BM + 4);
}

int g_form = 1;          // 1: split-fp16 where the shape allows it (Cout in blocks of 96); 0: always the exact fp32 form

}  // namespace

extern "C" {

int rdo_unit1x1_supported(int64_t M, int32_t K, int32_t N) {
    return (M > 0 && M % BM == 0 && K >= 32 && K <= KMAX && K % 32 == 0 && N >= 32 && N % 32 == 0 && M * (int64_t)K < (1LL << 31) &&
            M * (int64_t)N < (1LL << 31))
               ? 1
               : 0;
}

static bool h2_form(int32_t N, int32_t K, int32_t B) { return g_form != 0 && N % NB == 0 && (K == 96 || K == 192) && B <= 256; }

// token tiles per workgroup: as few as keep the launch at one workgroup per CU (256 CUs; the kernels' LDS allows one per CU).  The
// split-fp16 form's tiles are cheap enough to cap the slabs at 128 as well (Cout = 96 at 4 x 64^2: four tiles on 128 workgroups instead
// of two on 256 -- the step has half the slabs to sum: 42 -> 40 us per unit-iteration; eight tiles lose again).
static int tiles_per_wg(int64_t M, int32_t N) {
    const int64_t tiles = M / BM, nblk = (N + NB - 1) / NB;
    int64_t T = (tiles * nblk + 255) / 256;
    if (g_form != 0 && N % NB == 0) T = std::max<int64_t>(T, (tiles + 127) / 128);
    return (int)(T < 1 ? 1 : T);
}

// slabs a launch writes: one per chunk of T token tiles (<= 256; the AdaRound step sums them)
int rdo_unit1x1_nslab(int64_t M, int32_t N) {
    const int64_t tiles = M / BM, T = tiles_per_wg(M, N);
    return (int)((tiles + T - 1) / T);
}

// 1 (default): Cout in blocks of 96 runs the split-fp16 form; 0: always the exact fp32 form.  Returns the previous setting.  Process-wide;
// rdo_unit1x1_nslab follows it, so switch before the slabs of a unit are sized, not between a unit's launches.
int rdo_unit1x1_form(int32_t form) {
    const int was = g_form;
    if (form == 0 || form == 1) g_form = form;
    return was;
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
    const bool h2 = h2_form(N, K, B);
    const size_t lds = h2 ? h2_lds_bytes(K) : sizeof(float) * ((size_t)(2 * BM + NB) * (K + 4) + (size_t)BM * (NB + 4));
    const double flops = 4.0 * (double)M * K * N;
    const double bytes = 4.0 * ((double)M * K + (double)M * N + (double)N * K * (1.0 + nslab));
    return rdo::dispatch(
        [a, lds, nslab, h2](hipStream_t s) {
            static rdo::PerDevice attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(unit1x1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(sizeof(float) * ((size_t)(2 * BM + NB) * (KMAX + 4) + (size_t)BM * (NB + 4)))) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(unit1x1_h2_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)h2_lds_bytes(192)) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(unit1x1_h2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)h2_lds_bytes(96)) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(unit1x1) failed");
                attr.mark();
            }
            dim3 grid((unsigned)nslab, (unsigned)rdo::ceil_div(a.N, NB));
            if (h2 && a.K == 192) hipLaunchKernelGGL(unit1x1_h2_kernel<6>, grid, dim3(256), lds, s, a);
            else if (h2) hipLaunchKernelGGL(unit1x1_h2_kernel<3>, grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL(unit1x1_kernel, grid, dim3(256), lds, s, a);
            return rdo::check_launch("unit1x1");
        },
        stream, h2 ? "unit1x1_h2" : "unit1x1", flops, bytes);
}

}  // extern "C"
