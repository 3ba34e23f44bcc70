// Token-matrix Linear on fp16 two-way-split MFMA with a per-token dynamic scale (round 5; the Lu2022 path).
//
//   Y[M][N] = X[M][K] W^T (+ bias)        X fp32 in HBM (tokens x channels, the NHWC image of a [B,H,W,C] map), W [N][K]
//
// replaces F.linear of the Swin blocks (models/layers.py:147,163 qkv / proj, :44-47 fc1 / fc2; quant_layer.py:119 for the wrapped ones) and
// its input gradient (dX = dY W: the same kernel on W^T) for the LARGE token matrices -- the 64^2 and 128^2 maps, 16 K / 64 K tokens of
// 192 channels -- which the calibration engine ran on the split-bf16 conv kernel as 1x1 convolutions: six MFMA products per fp32
// product, activations split in that kernel's loader, 150 us for the 14.5-GFLOP qkv Linear of a 128^2 map (95 TFLOP/s).
//
// Arithmetic: the H2 form of conv_fwd_h2k.hip (x s = h1 + h2 in fp16, THREE products h1 g1 + h1 g2 + h2 g1 on v_mfma_f32_16x16x32_f16,
// fp32 accumulate, rdo_common.h) -- but the tape engine's tensors have no probed static scales, and gradients span many octaves
// from tensor to tensor.  A Linear reduces over a token's OWN channels only, so the scale can be per token: a workgroup loads the
// whole [64 tokens][192 channels] panel of a K block before any product, takes each token's largest magnitude, and splits
// x * 2^(7 - floor(log2 amax)): every token's values land in [2^-?, 2^8), nothing can overflow, no probe iteration, no flag, and a
// token of tiny gradients keeps full relative precision.  The accumulator of a token is rescaled by the (exact, power-of-two) ratio
// of two K blocks' scales when K spans several blocks.
//
// Structure: one workgroup = 64 tokens x one 192-channel chunk of the outputs, 4 waves; wave w owns output channels [48 w, 48 w + 48) of
// the chunk and all 64 tokens (3 x 4 accumulator tiles of 16 x 16).  A = weights (rows = output channels), B = activations
// (columns = tokens): a lane's four accumulator values are four consecutive channels of one token -> 16-byte stores straight from the
// accumulators, one scale per lane and tile.  The activation panel is STATIONARY in LDS ([plane][16-channel slice][token][32 B]:
// every ds_read_b128 fragment read is 2 x 512 contiguous bytes, conflict-free); weight fragments never touch LDS: the planes are
// stored in fragment order (rdo_split_h2_linear: [plane][K/32][N/16][lane][8 halfs], 1 KiB per fragment) and each wave loads its own
// three fragments per plane and K step with one coalesced global_load_dwordx4 each, one K step ahead -- no barrier inside the K loop.
// LDS 48.5 KiB: three workgroups per CU, so one workgroup's panel load (HBM) hides under the others' products.
#include <cstdlib>

#include "rdo_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;            // tokens per workgroup
constexpr int KB = 192;           // channels per K block (one LDS panel)
constexpr int NC = 192;           // output channels per chunk (4 waves x 48)
constexpr int SLICES = KB / 16;   // 12
constexpr int PLANE = SLICES * BM * 32;   // bytes per plane of the panel: 24 KiB
constexpr int LDS_BYTES = 2 * PLANE + 2 * BM * 4;   // + two scale rows (current / previous K block)

struct LinArgs {
    const float* x;        // [M][K]
    const unsigned short* wp;   // [2][K/32][N/16][64][8] fp16 bit patterns of W * wscale
    const float* bias;     // [N] or null
    float* out;            // [M][N]
    int M, K, N;
    float inv_wscale;      // 1 / wscale
    int square;            // the input enters squared (GDN norm pool: beta' + gamma' . x^2, quant_layer.py:147)
    int epi;               // 0: none; 1: out = gelu(y), pre = y (Mlp.fc1 + nn.GELU, layers.py:44-47); 2: out = y * gelu'(aux) (input gradient through that GELU)
    float* pre;            // epi 1: pre-activation [M][N]
    const float* aux;      // epi 2: the GELU's input [M][N]
};

__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

template <int EPI>
__global__ __launch_bounds__(256, 3) void linear_h2_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* scl = reinterpret_cast<float*>(smem + 2 * PLANE);          // [2][BM]: 1 / scale of the token for K block (kb & 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int nkb = a.K / KB, nchunks = a.N / NC, nblk = a.N / 16, ksteps = a.K / 32;
    int wg_tile, wg_chunk;
    {
        const int ntiles = a.M / BM;
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        const int full = ntiles >> 3, rest = ntiles & 7;
        const int mine = full + (xcd < rest ? 1 : 0), first = xcd * full + (xcd < rest ? xcd : rest);
        const int lt = slot / nchunks;
        wg_chunk = slot - lt * nchunks;
        if (lt >= mine) return;                               // (grid padded to a multiple of 8 tiles)
        wg_tile = first + lt;
    }
    const long m0 = (long)wg_tile * BM;

    // ---- panel loader: 16 lanes per token, three float4 per lane (channels 4 (l16 + 16 k) ...), 16 tokens per pass, 4 passes
    // All twelve loads of a thread are issued before the first is used: ONE memory round trip per panel (pass by pass -- load, reduce,
    // split, store -- a workgroup spent four dependent HBM latencies before its first product: 25 us per workgroup for 2 us of MFMAs).
    auto load_panel = [&](int kb) {
        const int row_in_pass = tid >> 4;
        f32x4 v[BM / 16][3];
#pragma unroll
        for (int pass = 0; pass < BM / 16; ++pass) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.x + (m0 + pass * 16 + row_in_pass) * a.K + (long)kb * KB);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[pass][k] = src[l16 + 16 * k];
        }
        if (a.square) {
#pragma unroll
            for (int pass = 0; pass < BM / 16; ++pass)
#pragma unroll
                for (int k = 0; k < 3; ++k) v[pass][k] *= v[pass][k];
        }
#pragma unroll
        for (int pass = 0; pass < BM / 16; ++pass) {
            const int r = pass * 16 + row_in_pass;
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[pass][k][0]), fabsf(v[pass][k][1])), fmaxf(fabsf(v[pass][k][2]), fabsf(v[pass][k][3]))));
            amax = fmaxf(amax, __shfl_xor(amax, 8, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 16));
            // scale 2^(7 - floor(log2 amax)): the token's largest value lands in [2^7, 2^8).  Zero / denormal / non-finite rows: scale 1
            // (zeros stay zeros; inf / NaN propagate through fp16 as they would through fp32)
            const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFF) - 127;
            const bool plain = e < -100 || e > 100;
            const float s = plain ? 1.f : pow2f(7 - e);
            if (l16 == 0) scl[(kb & 1) * BM + r] = plain ? 1.f : pow2f(e - 7);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c4 = l16 + 16 * k;                              // float4 index inside the K block: slice c4 / 4, 8-byte piece c4 % 4
                const f32x4 xs = v[pass][k] * s;
                const f16x4 hi = __builtin_convertvector(xs, f16x4);
                const f16x4 lo = __builtin_convertvector(xs - __builtin_convertvector(hi, f32x4), f16x4);
                char* dst = smem + (c4 >> 2) * (BM * 32) + r * 32 + (c4 & 3) * 8;
                *reinterpret_cast<f16x4*>(dst) = hi;
                *reinterpret_cast<f16x4*>(dst + PLANE) = lo;
            }
        }
    };

    // ---- weight fragments: plane p, K step ks, 16-channel block b -> 1 KiB at ((p * ksteps + ks) * nblk + b) * 512 halfs; lane -> 16 bytes
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp) + lane;
    auto load_w = [&](f16x8 (&fw)[2][3], int ks, int chunk) {
        const long b0 = (long)chunk * (NC / 16) + wave * 3;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                fw[p][i] = __builtin_bit_cast(f16x8, wbase[(((long)p * ksteps + ks) * nblk + b0 + i) * 64]);
    };
    // activation fragments of K step ks (inside the panel): token tile j, plane p
    const int fx_lane = (kg >> 1) * (BM * 32) + l16 * 32 + (kg & 1) * 16;
    auto read_x = [&](f16x8 (&fx)[2][4], int ksl) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fx[p][j] = *reinterpret_cast<const f16x8*>(smem + p * PLANE + (2 * ksl) * (BM * 32) + j * (16 * 32) + fx_lane);
    };

    // ---- (token tile, chunk) of this workgroup.  Launch ids go round-robin over the 8 XCDs; the chunks of one token tile read the same
    // panel, so they are neighbours INSIDE an XCD (slot = id / 8 walks (tile of this XCD, chunk), chunk fastest): the panel comes from
    // HBM once and from that XCD's L2 for the other chunks.  One chunk per workgroup: 3 x as many, shorter workgroups for the qkv
    // Linear (1024 workgroups on 768 resident slots left a third of the chip idle for the last round).
    f32x4 acc[3][4];
    {
        const int chunk = wg_chunk;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nkb; ++kb) {
            if (kb > 0) __syncthreads();                       // every wave is done with the previous panel
            load_panel(kb);
            f16x8 fw[2][2][3];
            load_w(fw[0], kb * (KB / 32), chunk);              // (in flight across the barrier)
            __syncthreads();
            if (kb > 0) {                                      // accumulators carry the previous block's token scale: exact power-of-two ratio
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ratio = scl[((kb - 1) & 1) * BM + 16 * j + l16] / scl[(kb & 1) * BM + 16 * j + l16];
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i][j] *= ratio;
                }
            }
#pragma unroll
            for (int ksl = 0; ksl < KB / 32; ++ksl) {
                // the next K step's weight fragments are requested HERE, a whole step (36 MFMAs) ahead of their first use: left to itself
                // hipcc sinks each of the six loads to a few instructions in front of the MFMA that needs it (`global_load; s_waitcnt
                // vmcnt(1); v_mfma` -- the L2 latency exposed six times per step)
                if (ksl + 1 < KB / 32) load_w(fw[(ksl + 1) & 1], kb * (KB / 32) + ksl + 1, chunk);
                __builtin_amdgcn_sched_barrier(0);
                f16x8 fx[2][4];
                read_x(fx, ksl);
                const int c = ksl & 1;
                // product-major: twelve independent accumulators between two MFMAs on the same one (small products first)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][1][i], fx[0][j], acc[i][j], 0, 0, 0);    // w_lo x_hi
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][0][i], fx[1][j], acc[i][j], 0, 0, 0);    // w_hi x_lo
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][0][i], fx[0][j], acc[i][j], 0, 0, 0);    // w_hi x_hi
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: y = acc / (token scale * weight scale) + bias, four consecutive channels of one token per lane
        const float* inv_s = scl + ((nkb - 1) & 1) * BM;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = 16 * j + l16;
            const float f = inv_s[tok] * a.inv_wscale;
            const long o = (m0 + tok) * a.N + chunk * NC + wave * 48 + 4 * kg;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 y = acc[i][j] * f;
                if (a.bias) y += *reinterpret_cast<const f32x4*>(a.bias + chunk * NC + wave * 48 + 16 * i + 4 * kg);
                if constexpr (EPI == 1) {                       // Mlp.fc1 + GELU: the pre-activation is kept for the backward
                    *reinterpret_cast<f32x4*>(a.pre + o + 16 * i) = y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = rdo::gelu(y[e]);
                } else if constexpr (EPI == 2) {                // the input gradient taken through that GELU
                    const f32x4 u = *reinterpret_cast<const f32x4*>(a.aux + o + 16 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] *= rdo::gelu_grad(u[e]);
                }
                *reinterpret_cast<f32x4*>(a.out + o + 16 * i) = y;
            }
        }
    }
}

// ---- persistent form (default): a workgroup walks token tiles t, t + grid, ... and handles ALL output chunks of a tile from one panel.
// The panel of the NEXT step (next K block of the tile, or the next tile) is fetched into registers right behind the barrier that
// publishes the current one, so HBM latency and the loads run under the products and the output stores: two workgroups per CU
// (registers), 512 resident, 1024 tiles of a 128^2 map = two each -- no partial last round, one panel load per tile instead of one per
// chunk.  K spanning several blocks (fc2, the qkv input gradient) is supported with one output chunk (N = 192); one K block with any
// number of chunks: every Linear shape of the Lu2022 blocks.
// EPI: 0 plain, 1 GELU (+ pre-activation output), 2 GELU_BWD -- compile-time, so the erf / exp code (and its registers) stays out of the
// plain instantiation; MULTIK: K spans several panels (one output chunk).
template <int EPI, bool MULTIK>
__global__ __launch_bounds__(256, 2) void linear_h2p_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* scl = reinterpret_cast<float*>(smem + 2 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    const int nkb = a.K / KB, nchunks = a.N / NC, nblk = a.N / 16, ksteps = a.K / 32;
    const int ntiles = a.M / BM;
    const int row_in_pass = tid >> 4;

    f32x4 v[BM / 16][3];
    auto fetch = [&](int tile, int kb) {
#pragma unroll
        for (int pass = 0; pass < BM / 16; ++pass) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.x + ((long)tile * BM + pass * 16 + row_in_pass) * a.K + (long)kb * KB);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[pass][k] = src[l16 + 16 * k];
        }
    };
    auto stash = [&](int kb) {
#pragma unroll
        for (int pass = 0; pass < BM / 16; ++pass) {
            const int r = pass * 16 + row_in_pass;
            if (a.square) {
#pragma unroll
                for (int k = 0; k < 3; ++k) v[pass][k] *= v[pass][k];
            }
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[pass][k][0]), fabsf(v[pass][k][1])), fmaxf(fabsf(v[pass][k][2]), fabsf(v[pass][k][3]))));
            amax = fmaxf(amax, __shfl_xor(amax, 8, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 16));
            const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFF) - 127;
            const bool plain = e < -100 || e > 100;
            const float s = plain ? 1.f : pow2f(7 - e);
            if (l16 == 0) scl[(kb & 1) * BM + r] = plain ? 1.f : pow2f(e - 7);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c4 = l16 + 16 * k;
                const f32x4 xs = v[pass][k] * s;
                const f16x4 hi = __builtin_convertvector(xs, f16x4);
                const f16x4 lo = __builtin_convertvector(xs - __builtin_convertvector(hi, f32x4), f16x4);
                char* dst = smem + (c4 >> 2) * (BM * 32) + r * 32 + (c4 & 3) * 8;
                *reinterpret_cast<f16x4*>(dst) = hi;
                *reinterpret_cast<f16x4*>(dst + PLANE) = lo;
            }
        }
    };
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp) + lane;
    auto load_w = [&](f16x8 (&fw)[2][3], int ks, int chunk) {
        const long b0 = (long)chunk * (NC / 16) + wave * 3;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                fw[p][i] = __builtin_bit_cast(f16x8, wbase[(((long)p * ksteps + ks) * nblk + b0 + i) * 64]);
    };
    const int fx_lane = (kg >> 1) * (BM * 32) + l16 * 32 + (kg & 1) * 16;
    auto read_x = [&](f16x8 (&fx)[2][4], int ksl) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fx[p][j] = *reinterpret_cast<const f16x8*>(smem + p * PLANE + (2 * ksl) * (BM * 32) + j * (16 * 32) + fx_lane);
    };
    f32x4 acc[3][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto kloop = [&](int kb, int chunk) {
        f16x8 fw[2][2][3];
        load_w(fw[0], kb * (KB / 32), chunk);
#pragma unroll
        for (int ksl = 0; ksl < KB / 32; ++ksl) {
            if (ksl + 1 < KB / 32) load_w(fw[(ksl + 1) & 1], kb * (KB / 32) + ksl + 1, chunk);
            f16x8 fx[2][4];
            read_x(fx, ksl);
            const int c = ksl & 1;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][1][i], fx[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][0][i], fx[1][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[c][0][i], fx[0][j], acc[i][j], 0, 0, 0);
        }
    };
    auto epilogue = [&](int tile, int chunk) {
        const float* inv_s = scl + ((nkb - 1) & 1) * BM;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = 16 * j + l16;
            const float f = inv_s[tok] * a.inv_wscale;
            const long o = ((long)tile * BM + tok) * a.N + chunk * NC + wave * 48 + 4 * kg;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 y = acc[i][j] * f;
                if (a.bias) y += *reinterpret_cast<const f32x4*>(a.bias + chunk * NC + wave * 48 + 16 * i + 4 * kg);
                if constexpr (EPI == 1) {
                    *reinterpret_cast<f32x4*>(a.pre + o + 16 * i) = y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = rdo::gelu(y[e]);
                } else if constexpr (EPI == 2) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(a.aux + o + 16 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] *= rdo::gelu_grad(u[e]);
                }
                *reinterpret_cast<f32x4*>(a.out + o + 16 * i) = y;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    fetch(tile, 0);
    bool first = true;
    for (; tile < ntiles; tile += gridDim.x) {
        for (int kb = 0; kb < nkb; ++kb) {
            if (!first) __syncthreads();                       // every wave is done with the previous panel (and has read its scales)
            first = false;
            stash(kb);
            // the next step's panel: next K block of this tile, else the first block of the next tile
            const int ntile = kb + 1 < nkb ? tile : tile + (int)gridDim.x;
            const int nkbi = kb + 1 < nkb ? kb + 1 : 0;
            if (ntile < ntiles) fetch(ntile, nkbi);
            __syncthreads();
            if constexpr (!MULTIK) {
#pragma clang loop unroll(disable)
                for (int chunk = 0; chunk < nchunks; ++chunk) {
                    int c = chunk;
                    asm volatile("" : "+s"(c));              // (opaque: the optimiser otherwise keeps one induction pointer per weight fragment
                                                             //  and K step of the chunk loop alive -- 113 spilled registers)
                    zero_acc();
                    kloop(0, c);
                    epilogue(tile, c);
                }
            } else {
                if (kb == 0) {
                    zero_acc();
                } else {                                       // accumulators carry the previous block's token scale: exact power-of-two ratio
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ratio = scl[((kb - 1) & 1) * BM + 16 * j + l16] / scl[(kb & 1) * BM + 16 * j + l16];
#pragma unroll
                        for (int i = 0; i < 3; ++i) acc[i][j] *= ratio;
                    }
                }
                kloop(kb, 0);
                if (kb == nkb - 1) epilogue(tile, 0);
            }
        }
    }
}

// W [N][K] fp32 -> fragment-ordered fp16 planes of W * scale.  One thread per (plane-independent) 8-half record.
__global__ __launch_bounds__(256) void split_h2_linear_kernel(const float* w, int N, int K, float scale, unsigned short* planes) {
    const int ksteps = K / 32, nblk = N / 16;
    const long recs = (long)ksteps * nblk * 64;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < recs; t += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long fb = t >> 6;                         // ks * nblk + b
        const int b = (int)(fb % nblk), ks = (int)(fb / nblk);
        const int row = 16 * b + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
        const float* src = w + (long)row * K + k0;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = src[e] * scale;
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)(v - (float)h);
        }
        *reinterpret_cast<f16x8*>(planes + t * 8) = hi;
        *reinterpret_cast<f16x8*>(planes + (recs + t) * 8) = lo;
    }
}

bool pow2(float s) {
    if (!(s > 0.f) || s != s || s > 3.0e38f) return false;
    int e;
    return frexpf(s, &e) == 0.5f;
}

}  // namespace

extern "C" int rdo_linear_h2_supported(int64_t M, int32_t K, int32_t N) {
    return M > 0 && M % BM == 0 && K > 0 && K % KB == 0 && N > 0 && N % NC == 0 && (double)M * (K > N ? K : N) * 4.0 < 4.0e9;
}

// tuning: 0 (default) = one (token tile, chunk) per workgroup, three workgroups per CU; 1 = persistent workgroups with register prefetch
// (linear_h2p_kernel: two per CU, all chunks of a tile from one panel) -- measured SLOWER on every Lu2022 shape (tools/bench_linear_h2.py:
// 80 against 68 us for the 192 -> 576 Linear over 64 K tokens, 33 against 20.5 over 16 K): a third resident workgroup and three times
// as many, shorter workgroups hide more latency than a software prefetch at lower occupancy.  Kept for A/B runs.
static int lin_persistent() {
    static const int v = [] { const char* e = getenv("RDO_LIN_H2_PERSISTENT"); return e ? atoi(e) : 0; }();
    return v;
}

extern "C" int rdo_split_h2_linear(const float* w, int32_t N, int32_t K, float scale, void* planes, void* stream) {
    RDO_REQUIRE(w && planes && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "rdo_split_h2_linear: bad argument (N %% 16, K %% 32)");
    RDO_REQUIRE(pow2(scale), "rdo_split_h2_linear: scale %g is not a power of two", (double)scale);
    const long recs = (long)(K / 32) * (N / 16) * 64;
    unsigned short* p = reinterpret_cast<unsigned short*>(planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(recs, 256);
            hipLaunchKernelGGL(split_h2_linear_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, s, w, N, K, scale, p);
            return rdo::check_launch("split_h2_linear");
        },
        stream, "split_h2_linear", 0.0, 8.0 * (double)N * K);
}

extern "C" int rdo_linear_h2_epi(const float* x, int64_t M, int32_t K, int32_t N, const void* wplanes, float wscale, const float* bias,
                                 int32_t square_input, int32_t epilogue, float* pre, const float* aux, float* out, void* stream);

extern "C" int rdo_linear_h2(const float* x, int64_t M, int32_t K, int32_t N, const void* wplanes, float wscale, const float* bias,
                             int32_t square_input, float* out, void* stream) {
    return rdo_linear_h2_epi(x, M, K, N, wplanes, wscale, bias, square_input, RDO_EPI_NONE, nullptr, nullptr, out, stream);
}

extern "C" int rdo_linear_h2_epi(const float* x, int64_t M, int32_t K, int32_t N, const void* wplanes, float wscale, const float* bias,
                                 int32_t square_input, int32_t epilogue, float* pre, const float* aux, float* out, void* stream) {
    RDO_REQUIRE(x && wplanes && out, "rdo_linear_h2: null argument");
    RDO_REQUIRE(epilogue == RDO_EPI_NONE || epilogue == RDO_EPI_GELU || epilogue == RDO_EPI_GELU_BWD, "rdo_linear_h2: epilogue %d (NONE / GELU / GELU_BWD)", epilogue);
    RDO_REQUIRE(epilogue != RDO_EPI_GELU || pre, "rdo_linear_h2: RDO_EPI_GELU needs the pre-activation output");
    RDO_REQUIRE(epilogue != RDO_EPI_GELU_BWD || aux, "rdo_linear_h2: RDO_EPI_GELU_BWD needs aux");
    RDO_REQUIRE(((reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(aux)) & 15) == 0, "rdo_linear_h2: pointers must be 16-byte aligned");
    RDO_REQUIRE(rdo_linear_h2_supported(M, K, N), "rdo_linear_h2: shape %ld x %d -> %d is not supported (M %% 64, K %% 192, N %% 192)", (long)M, K, N);
    RDO_REQUIRE(pow2(wscale), "rdo_linear_h2: weight scale %g is not a power of two", (double)wscale);
    RDO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wplanes) | reinterpret_cast<uintptr_t>(bias) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, "rdo_linear_h2: pointers must be 16-byte aligned");
    LinArgs a;
    a.x = x; a.wp = reinterpret_cast<const unsigned short*>(wplanes); a.bias = bias; a.out = out;
    a.M = (int)M; a.K = K; a.N = N; a.inv_wscale = 1.0f / wscale; a.square = square_input ? 1 : 0;
    a.epi = epilogue == RDO_EPI_GELU ? 1 : (epilogue == RDO_EPI_GELU_BWD ? 2 : 0);
    a.pre = pre; a.aux = aux;
    const double flops = 2.0 * (double)M * K * N;
    // the persistent kernel takes one K block with any number of chunks, or several K blocks with one chunk; epilogues only there
    const bool pers = lin_persistent() && (K == KB || N == NC);
    return rdo::dispatch(
        [a, pers](hipStream_t s) {
            if (pers) {
                const int ntiles = a.M / BM;
                const int grid = ntiles < 512 ? ntiles : 512;        // two workgroups per CU (register budget)
                const bool mk = a.K != KB;
                auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a); };
                if (a.epi == 0) { if (mk) go(linear_h2p_kernel<0, true>); else go(linear_h2p_kernel<0, false>); }
                else if (a.epi == 1) { if (mk) go(linear_h2p_kernel<1, true>); else go(linear_h2p_kernel<1, false>); }
                else { if (mk) go(linear_h2p_kernel<2, true>); else go(linear_h2p_kernel<2, false>); }
                return rdo::check_launch("linear_h2p");
            }
            const unsigned grid = (unsigned)(rdo::ceil_div(a.M / BM, 8) * 8 * (a.N / NC));
            if (a.epi == 0) hipLaunchKernelGGL(linear_h2_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, s, a);
            else if (a.epi == 1) hipLaunchKernelGGL(linear_h2_kernel<1>, dim3(grid), dim3(256), LDS_BYTES, s, a);
            else hipLaunchKernelGGL(linear_h2_kernel<2>, dim3(grid), dim3(256), LDS_BYTES, s, a);
            return rdo::check_launch("linear_h2");
        },
        stream, "linear_h2", flops, 4.0 * ((double)M * K + (double)M * N + (double)K * N));
}
