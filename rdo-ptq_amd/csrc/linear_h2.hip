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
#ifdef RDO_DIAG
    int diag;              // ablation bits (diagnostic build only): 1 no output stores, 2 no MFMAs, 4 no panel loads
#endif
    int epi;               // 0: none; 1: out = gelu(y), pre = y (Mlp.fc1 + nn.GELU, layers.py:44-47); 2: out = y * gelu'(aux) (input gradient through that GELU)
    float* pre;            // epi 1: pre-activation [M][N]
    const float* aux;      // epi 2: the GELU's input [M][N]
};

__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// PART: K = 96 or N = 96 (half a panel / half a chunk) -- an instantiation of its own, the full shapes keep their register budget
template <int EPI, bool PART = false>
__global__ __launch_bounds__(256, 3) void linear_h2_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* scl = reinterpret_cast<float*>(smem + 2 * PLANE);          // [2][BM]: 1 / scale of the token for K block (kb & 1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kg = lane >> 4;
    // K a multiple of 192, or 96 (half a panel: lanes past the row re-read it and are zeroed, three K steps); N a multiple of 192, or 96
    // (half a chunk: waves 2 and 3 only keep the barriers)
    const int nkb = PART ? (a.K + KB - 1) / KB : a.K / KB, nchunks = PART ? (a.N + NC - 1) / NC : a.N / NC, nblk = a.N / 16, ksteps = a.K / 32;
    const int kvalid4 = PART ? (a.K < KB ? a.K : KB) >> 2 : KB / 4;           // float4 per row of a K block
    const int nks = PART ? (ksteps < KB / 32 ? ksteps : KB / 32) : KB / 32;   // K steps per block
    const bool wave_live = !PART || a.N >= NC || wave * 48 < a.N;             // (N = 96: channels 0-95 = waves 0, 1)
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
            for (int k = 0; k < 3; ++k) {
#ifdef RDO_DIAG
                if (a.diag & 4) { v[pass][k] = f32x4{1.f, 2.f, 3.f, 4.f}; continue; }
#endif
                const int c4 = l16 + 16 * k;
                v[pass][k] = src[(!PART || c4 < kvalid4) ? c4 : 0];     // (unconditional: a load behind a per-lane branch costs a vmcnt(0) at the join)
            }
        }
        if (PART && kvalid4 < KB / 4) {                      // (uniform) K = 96: zero what lies past the row
#pragma unroll
            for (int pass = 0; pass < BM / 16; ++pass)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (l16 + 16 * k >= kvalid4) v[pass][k] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        const long b0 = wave_live ? (long)chunk * (NC / 16) + wave * 3 : 0;        // (idle waves of a half chunk read block 0, unused)
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
                if (PART && ksl >= nks) break;                 // (uniform; K = 96: three steps)
                // the next K step's weight fragments are requested HERE, a whole step (36 MFMAs) ahead of their first use: left to itself
                // hipcc sinks each of the six loads to a few instructions in front of the MFMA that needs it (`global_load; s_waitcnt
                // vmcnt(1); v_mfma` -- the L2 latency exposed six times per step)
                if (ksl + 1 < KB / 32 && (!PART || ksl + 1 < nks)) load_w(fw[(ksl + 1) & 1], kb * (KB / 32) + ksl + 1, chunk);
                __builtin_amdgcn_sched_barrier(0);
#ifdef RDO_DIAG
                if (a.diag & 2) continue;
#endif
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
        if (!wave_live) return;                                // (no barrier follows)
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
#ifdef RDO_DIAG
                if ((a.diag & 1) && y[0] != 12345.678f) continue;
#endif
                *reinterpret_cast<f32x4*>(a.out + o + 16 * i) = y;
            }
        }
    }
}

// ---- weight-stationary form (default where it applies): the WEIGHTS live in registers, tokens stream through LDS.
// tools/linear_h2_ablate.py on the kernel above (192 -> 576 over 64 K tokens, 65 us): with panel loads, MFMAs and stores all switched off it
// still takes 24 us -- every workgroup pulls its chunk's 147 KB of weight fragments from L2 for 64 tokens (452 MB through the vector
// memory path for 200 MB of tensor traffic) -- and the marginal cost of loads (+8), MFMAs (+19) and stores (+14) adds up exactly to the
// full time: a launch is only four rounds of workgroups that start together, so their phases never interleave.
// A (64-token, 192-channel chunk) product needs 192 x 192 weights = 144 KiB as planes: that FITS the register file of a CU.  One
// workgroup per CU, 8 waves = 4 channel groups x 2 token halves; a wave keeps the fragments of ITS channels for the whole K in registers
// (NT x 16 channels x K: 48 x 192 for K = 192, 16 x K for K = 384 / 576 -- 144 VGPRs either way), loaded once per launch, and walks
// token tiles: the raw fp32 panel of the next step is in registers (24 VGPRs, fetched a whole step ahead), split into the OTHER of two
// LDS panels while the current one is multiplied, one barrier per panel.  Panel loads, products and output stores of consecutive tiles
// overlap inside one workgroup; the weight traffic of a launch is 256 x 288 KB instead of 147 KB per 64 tokens.
// The workgroups of one token-tile group (one per output chunk) sit in one XCD and run in step: the panel comes from HBM once.
// EPI as above (compile-time); NKB = K / 192 (the K blocks of a tile are walked with the accumulators rescaled in between).
template <int NT, int NKB, int EPI>
__global__ __launch_bounds__(512, 1) void linear_h2w_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = 2 * PLANE;                           // one panel: two planes
    constexpr int KS = (KB / 32) * NKB;                      // K steps of the whole reduction
    constexpr int NCW = 64 * NT;                             // output channels per workgroup (4 channel groups x NT x 16)
    float* scl = reinterpret_cast<float*>(smem + 2 * BUF);   // [4][BM]: 1 / token scale of panel (p & 3)
    float* bsm = scl + 4 * BM;                               // [NCW]: the chunk's bias (zeros without one)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave & 3, th = wave >> 2;
    const int l16 = lane & 15, kg = lane >> 4;
    const int nchunks = a.N / NCW, nblk = a.N / 16;
    const int ntiles = a.M / BM;
    // (XCD, slot) -> (tile group, chunk): the chunks of a group are neighbours inside an XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int groups = slots / nchunks;
    const int grp = slot / nchunks, chunk = slot - grp * nchunks;
    const int full = ntiles >> 3, rest = ntiles & 7;
    const int mine = full + (xcd < rest ? 1 : 0), first = xcd * full + (xcd < rest ? xcd : rest);
    if (grp >= groups || grp >= mine) return;

    const int row_in_pass = tid >> 4;                        // 32 tokens per pass, two passes
    f32x4 v[2][3];
    // The prefetch loads are inline asm, invisible to the compiler's wait-count bookkeeping, and waited for by hand: the counter is shared
    // with the output stores, a loop header is reached from the prologue (nothing behind the loads) and from the back edge (the epilogue's
    // stores behind them), and the compiler has to emit ONE immediate that is right for both -- `vmcnt(5) ... vmcnt(0)`, i.e. every step
    // waited for the previous tile's stores to be acknowledged.
    auto fetch = [&](int tile, int kb) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float* src = a.x + ((long)tile * BM + pass * 32 + row_in_pass) * a.K + (long)kb * KB + 4 * l16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[pass][0]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(v[pass][1]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:512" : "=v"(v[pass][2]) : "v"(src) : "memory");
        }
    };
    // behind = vector-memory instructions issued after the loads that have to be back (the counter retires in order): -1 = all of them
    auto landed = [&](int behind) {
#ifdef RDO_DIAG
        behind = -1;            // a diagnostic build's ablations change how many stores sit behind the loads: never count, wait for all
#endif
        if (behind < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (behind == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (behind == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(v[pass][k]));
    };
    auto stash = [&](int p) {                                // registers -> planes of panel p (LDS buffer p & 1, scale row p & 3)
        char* buf = smem + (p & 1) * BUF;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = pass * 32 + row_in_pass;
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
            const float sc = plain ? 1.f : pow2f(7 - e);
            if (l16 == 0) scl[(p & 3) * BM + r] = plain ? 1.f : pow2f(e - 7);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c4 = l16 + 16 * k;
                const f32x4 xs = v[pass][k] * sc;
                const f16x4 hi = __builtin_convertvector(xs, f16x4);
                const f16x4 lo = __builtin_convertvector(xs - __builtin_convertvector(hi, f32x4), f16x4);
                char* dst = buf + (c4 >> 2) * (BM * 32) + r * 32 + (c4 & 3) * 8;
                *reinterpret_cast<f16x4*>(dst) = hi;
                *reinterpret_cast<f16x4*>(dst + PLANE) = lo;
            }
        }
    };

    int tl = grp;                                            // tile index inside this XCD's share
    fetch(first + tl, 0);
    // (the bias goes through LDS: a global load in the epilogue costs an `s_waitcnt vmcnt(0)` there, which also drains the panel prefetch)
    if (tid < NCW) bsm[tid] = a.bias ? a.bias[chunk * NCW + tid] : 0.f;
    // the wave's weight fragments, whole K: [plane][K step][16-channel tile]
    f16x8 fw[2][KS][NT];
    {
        const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp) + lane;
        const long b0 = (long)chunk * (NCW / 16) + cg * NT;
        u32x4 raw[2][KS][NT];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < NT; ++i) raw[p][ks][i] = wbase[(((long)p * KS + ks) * nblk + b0 + i) * 64];
        // (opaque: left to itself the compiler RE-LOADS half of the fragments inside every step -- a load from a constant address is
        //  cheaper to rematerialise than to keep -- and the in-order vmcnt waits of those 18 loads also drain the panel prefetch)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    asm volatile("" : "+v"(raw[p][ks][i]));
                    fw[p][ks][i] = __builtin_bit_cast(f16x8, raw[p][ks][i]);
                }
    }
    const int fx_lane = (kg >> 1) * (BM * 32) + (32 * th + l16) * 32 + (kg & 1) * 16;
    f32x4 acc[NT][2];
    int p = 0;                                               // panel counter of this workgroup
    landed(-1);
    stash(0);
    {
        const int ntl = NKB > 1 ? tl : tl + groups;
        fetch(first + (ntl < mine ? ntl : tl), NKB > 1 ? 1 : 0);
    }
    __syncthreads();
    constexpr int NST = 2 * NT * (EPI ? 2 : 1);              // vector-memory instructions of an epilogue
    static_assert(NST == 2 || NST == 6 || NST == 12, "landed()");
    bool first_step = true;
    for (; tl < mine; tl += groups) {
        const int tile = first + tl;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb, ++p) {
            // the panel after this one (already in registers) goes into the other LDS buffer -- every wave left it at the last barrier --
            // and the one after that is requested: a whole step of products between a load and its use
            // UNCONDITIONALLY (past the end: a valid tile again, never used): behind a branch the compiler has to assume the loads of the
            // skipped path are still in flight and puts `s_waitcnt vmcnt(0)` -- a wait for the previous tile's output stores -- in front of
            // the new requests
            {
                // the loads of panel p + 1 were issued in the previous step; behind them only that step's epilogue, if it had one
                if (kb == 0 && !first_step) landed(NST);
                else landed(0);
                first_step = false;
                stash(p + 1);
                int t2 = tl, k2 = kb + 2;
                if (k2 >= NKB) { t2 += groups; k2 -= NKB; }
                if (k2 >= NKB) { t2 += groups; k2 -= NKB; }      // (NKB = 1: two tiles on)
                fetch(first + (t2 < mine ? t2 : tl), k2);
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* buf = smem + (p & 1) * BUF;
            if (kb == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {                                             // accumulators carry the previous block's token scale: exact power-of-two ratio
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tok = 32 * th + 16 * j + l16;
                    const float ratio = scl[((p - 1) & 3) * BM + tok] / scl[(p & 3) * BM + tok];
#pragma unroll
                    for (int i = 0; i < NT; ++i) acc[i][j] *= ratio;
                }
            }
#pragma unroll
            for (int ksl = 0; ksl < KB / 32; ++ksl) {
                f16x8 fx[2][2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        fx[pl][j] = *reinterpret_cast<const f16x8*>(buf + pl * PLANE + (2 * ksl) * (BM * 32) + j * (16 * 32) + fx_lane);
                const int ks = kb * (KB / 32) + ksl;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[1][ks][i], fx[0][j], acc[i][j], 0, 0, 0);    // w_lo x_hi
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][ks][i], fx[1][j], acc[i][j], 0, 0, 0);    // w_hi x_lo
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][ks][i], fx[0][j], acc[i][j], 0, 0, 0);    // w_hi x_hi
            }
            if (kb == NKB - 1) {                                 // epilogue: four consecutive channels of one token per lane
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tok = 32 * th + 16 * j + l16;
                    const float f = scl[(p & 3) * BM + tok] * a.inv_wscale;
                    const int ch = chunk * NCW + cg * (16 * NT) + 4 * kg;
                    const long o = ((long)tile * BM + tok) * a.N + ch;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        f32x4 y = acc[i][j] * f;
                        y += *reinterpret_cast<const f32x4*>(bsm + cg * (16 * NT) + 4 * kg + 16 * i);
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
            }
            __syncthreads();                                     // panel p + 1 is published, panel p is free
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
    return M > 0 && M % BM == 0 && K > 0 && (K % KB == 0 || K == 96) && N > 0 && (N % NC == 0 || N == 96) && (double)M * (K > N ? K : N) * 4.0 < 4.0e9;
}

// tuning: 1 (default) = the weight-stationary kernel for one K block (K = 192) when every workgroup gets at least four token tiles (the
// 128^2 maps: 192 -> 576 over 64 K tokens 67 -> 63 us, -> 384 48 -> 38.5, -> 192 29 -> 26; over 16 K tokens the one-time weight load and the
// tile quantisation lose: 23.6 against 20.0), 2 = also for K = 384 / 576 (16 channels x K per wave: LDS-read-bound, measured 60 / 102 us
// against 44 / 61 -- kept for A/B runs), 0 = always one (token tile, chunk) per workgroup with the weights streamed from L2
static int lin_stationary() {      // read per call (calls are recorded once per plan): a test compares the two kernels in one process
    const char* e = getenv("RDO_LIN_H2_STATIONARY");
    return e ? atoi(e) : 1;
}
static int cu_count() {
    static const int v = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n & ~7;
    }();
    return v;
}

template <int NT, int NKB, int EPI>
static int launch_w(const LinArgs& a, int slots, hipStream_t s) {
    constexpr int lds = 4 * PLANE + 4 * BM * 4 + 64 * NT * 4;   // two panels, four scale rows, the chunk's bias
    static rdo::PerDevice attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_h2w_kernel<NT, NKB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(linear_h2w) failed");
        attr.mark();
    }
    hipLaunchKernelGGL((linear_h2w_kernel<NT, NKB, EPI>), dim3((unsigned)(8 * slots)), dim3(512), lds, s, a);
    return rdo::check_launch("linear_h2w");
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
    RDO_REQUIRE(rdo_linear_h2_supported(M, K, N), "rdo_linear_h2: shape %ld x %d -> %d is not supported (M %% 64; K and N multiples of 192, or 96)", (long)M, K, N);
    RDO_REQUIRE(pow2(wscale), "rdo_linear_h2: weight scale %g is not a power of two", (double)wscale);
    RDO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wplanes) | reinterpret_cast<uintptr_t>(bias) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, "rdo_linear_h2: pointers must be 16-byte aligned");
    LinArgs a;
    a.x = x; a.wp = reinterpret_cast<const unsigned short*>(wplanes); a.bias = bias; a.out = out;
    a.M = (int)M; a.K = K; a.N = N; a.inv_wscale = 1.0f / wscale; a.square = square_input ? 1 : 0;
    a.epi = epilogue == RDO_EPI_GELU ? 1 : (epilogue == RDO_EPI_GELU_BWD ? 2 : 0);
    a.pre = pre; a.aux = aux;
#ifdef RDO_DIAG
    { const char* e = getenv("RDO_LIN_DIAG"); a.diag = e ? atoi(e) : 0; }
#endif
    const double flops = 2.0 * (double)M * K * N;
    const int nkb = K % KB == 0 ? K / KB : 0;                // (0: the half panel K = 96 -- streaming kernel only)
    const int slots = cu_count() / 8;
    bool stat = false;
    // (not with RDO_EPI_GELU_BWD: its aux reads sit behind the products and their wait drains the panel prefetch -- 83 against 79 us)
    if (lin_stationary() && nkb == 1 && N % NC == 0 && N / NC <= slots && (a.epi != 2 || lin_stationary() == 2)) stat = (M / BM) / (8 * (slots / (N / NC))) >= 4;
    else if (lin_stationary() == 2 && (nkb == 2 || nkb == 3) && a.epi == 0 && N / 64 <= slots) stat = true;
    return rdo::dispatch(
        [a, stat, nkb, slots](hipStream_t s) {
            if (stat) {
                if (nkb == 1) {
                    if (a.epi == 0) return launch_w<3, 1, 0>(a, slots, s);
                    if (a.epi == 1) return launch_w<3, 1, 1>(a, slots, s);
                    return launch_w<3, 1, 2>(a, slots, s);
                }
                if (nkb == 2) return launch_w<1, 2, 0>(a, slots, s);
                return launch_w<1, 3, 0>(a, slots, s);
            }
            const unsigned grid = (unsigned)(rdo::ceil_div(a.M / BM, 8) * 8 * rdo::ceil_div(a.N, NC));
            const bool part = a.K % KB != 0 || a.N % NC != 0;
            if (part) {
                if (a.epi == 0) hipLaunchKernelGGL((linear_h2_kernel<0, true>), dim3(grid), dim3(256), LDS_BYTES, s, a);
                else if (a.epi == 1) hipLaunchKernelGGL((linear_h2_kernel<1, true>), dim3(grid), dim3(256), LDS_BYTES, s, a);
                else hipLaunchKernelGGL((linear_h2_kernel<2, true>), dim3(grid), dim3(256), LDS_BYTES, s, a);
            } else if (a.epi == 0) hipLaunchKernelGGL((linear_h2_kernel<0, false>), dim3(grid), dim3(256), LDS_BYTES, s, a);
            else if (a.epi == 1) hipLaunchKernelGGL((linear_h2_kernel<1, false>), dim3(grid), dim3(256), LDS_BYTES, s, a);
            else hipLaunchKernelGGL((linear_h2_kernel<2, false>), dim3(grid), dim3(256), LDS_BYTES, s, a);
            return rdo::check_launch("linear_h2");
        },
        stream, "linear_h2", flops, 4.0 * ((double)M * K + (double)M * N + (double)K * N));
}
