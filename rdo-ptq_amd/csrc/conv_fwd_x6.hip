// Forward / dgrad conv of the LARGE problems (rdo_conv2d_fwd_uses_bf16x6) with each fp32 operand split exactly into three
// bf16 planes and the six significant cross products issued on v_mfma_f32_32x32x16_bf16 ("bf16x6").
// rdo_conv2d_fwd dispatches to v6 (stride-1 3-wide kernels: the three kw taps share one activation image in LDS) or v5 (everything
// else): 128 x 192 tile, K stage 16, weight tile staged by LDS-DMA, fragment reads pipelined one stage ahead, channel-slice-outer /
// taps-inner stage order, XCD-aware tile numbering.  v3 (register-staged) remains as the fallback for weight tensors beyond 2^31
// plane elements.
//
//   x = x1 + x2 + x3 exactly (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)), same for w;
//   x*w ~= x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1          dropped terms <= 3 * 2^-24 |x w|  (fp32-rounding level)
//
// Each bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result has fp32-level accuracy at 16/6 the
// fp32-MFMA rate.  Weights arrive pre-split in fragment order ([3][Cin/16][KH][KW][Cout][16] bf16, rdo_split_bf16x3_conv /
// the AdaRound step; rdo::frag_index); activations are split in the loader.
#include <utility>

#include "rdo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int BK = 32;

struct X6Args {
    const float* x;
    const u16* wp;        // [3][Cin/16][KH][KW][Cout][16] bf16 planes (fragment order)
    const float* bias;
    const float* aux;
    const float* residual;
    float* out;
    float* pre;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, csteps;
    int epilogue, square_input, add_residual;
    long wplane;          // elements per weight plane
    int xcd_mode;         // 1: XCD-aware tile numbering
    float* partial;       // split-K (v3): raw accumulators [ksplit][M][Cout]; nullptr -> final output
    int ksplit;
    int partial_only;     // split-K: no epilogue launch, the slabs stay in the workspace (rdo_conv2d_fwd_partials)
};

// exact three-way split through the hardware RNE conversion (a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u16 bf16_bits(__bf16 h) { return __builtin_bit_cast(u16, h); }
__device__ __forceinline__ void split3(float v, u16& a, u16& b, u16& c) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    a = bf16_bits(h); b = bf16_bits(m); c = bf16_bits(l);
}

typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// two RNE conversions in one v_cvt_pk_bf16_f32; the packed pair is already the LDS image of two consecutive k
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
// exact three-way split of a quad of consecutive channels -> three 8-byte runs (22 vector instructions for 4 values)
__device__ __forceinline__ void split3_x4(const f32x4& v, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    const unsigned h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
    const float r0 = v[0] - lo_f(h01), r1 = v[1] - hi_f(h01), r2 = v[2] - lo_f(h23), r3 = v[3] - hi_f(h23);
    const unsigned m01 = cvt_pk(r0, r1), m23 = cvt_pk(r2, r3);
    const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m23), s3 = r3 - hi_f(m23);
    const unsigned l01 = cvt_pk(s0, s1), l23 = cvt_pk(s2, s3);
    p0 = __builtin_bit_cast(bf16x4, u32x2{h01, h23});
    p1 = __builtin_bit_cast(bf16x4, u32x2{m01, m23});
    p2 = __builtin_bit_cast(bf16x4, u32x2{l01, l23});
}

// XCD-aware tile id: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2).  Re-number
// so that every XCD works on a CONTIGUOUS range of logical tile ids: neighbouring M tiles read overlapping input rows (3x3 halo)
// and the same weight tile, so they should share an L2.  Bijective for any grid size (q, r split of the tail).
struct TileId { int m, n, z; };
__device__ __forceinline__ TileId xcd_tile_id(int mode) {
    if (mode == 0) return TileId{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    // N fastest: the N tiles of one M tile read the same activations; then M; split-K slices last
    TileId t;
    t.n = id % gridDim.y;
    const int rest = id / gridDim.y;
    t.m = rest % gridDim.x;
    t.z = rest / gridDim.x;
    return t;
}

// HEAVY: with the GELU epilogues (see conv_fwd.hip: kept out of the plain instantiations)
template <bool HEAVY>
__device__ __forceinline__ void finish(const X6Args& a, long o, float v) {
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

// LDS rows are 32 B per plane (2 chunks of 8 bf16); chunk swizzle c ^= (row>>3)&1 keeps ds_read_b128 conflict-free.
__device__ __forceinline__ int chunk_off16(int row, int c) { return row * 32 + ((c ^ ((row >> 3) & 1)) << 4); }

#ifdef RDO_DIAG   // superseded variants: only in a `make DIAG=1` build (A/B measurements, tuning key "fwd_x6_ver" = 3 / 4)
// ---- v3: as v2 with a 128 x 192 workgroup tile (wave tile 64 x 96): half the weight-tile traffic and fragment reads per MFMA
// LDS rows are 32 B per plane (2 chunks); chunk swizzle c ^= (row>>3)&1.

// DMA = true stages the weight tile with global_load_lds (no VGPR round trip, no select / ds_write instructions): wave w issues
// the 1 KiB pieces w, w+4, ... of the 18 KiB [plane][row][chunk] image, the chunk swizzle applied on the per-lane SOURCE address.
template <bool DMA>
__global__ __launch_bounds__(256, 2) void conv_fwd_x6v3_kernel(X6Args a) {
    constexpr int BM = 128, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;                          // wave tile 64 x 96 (2x2 waves)
    constexpr int APLANE = BM * 32, BPLANE = BN * 32;      // bytes per plane per stage
    constexpr int STAGE = 3 * (APLANE + BPLANE);           // 24 KiB
    constexpr int NB = 5;                                  // B chunks per thread (the 5th only for tid < 128)

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE] + 4 KiB dummy target for masked chunk stores
    char* dummy = smem + 2 * STAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;

    // A: two fp32 quads per thread per stage: rows tid>>2 and 64 + tid>>2, k-quad = tid&3
    const int aq = tid & 3;
    int hi0[2], wi0[2], abase[2], a_lds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int arow = (tid >> 2) + 64 * t;
        const int m = m0 + arow;
        const bool okr = m < a.M;
        const int mm = okr ? m : 0;
        const int b = mm / (a.Ho * a.Wo);
        const int rem = mm - b * (a.Ho * a.Wo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[t] = okr ? ho * a.stride - a.pad : -(1 << 28);
        wi0[t] = wo * a.stride - a.pad;
        abase[t] = b * a.H * a.W * a.Cin + 4 * aq;
        a_lds[t] = chunk_off16(arow, aq >> 1) + (aq & 1) * 8;
    }
    // B: chunk e = tid + 256 j over [plane][row][chunk(2)]
    long wbase[NB];
    int b_lds[NB];
    bool b_ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int e = tid + 256 * j;
        const bool live = e < 3 * BN * 2;
        const int ee = live ? e : 0;
        const int pl = ee / (BN * 2);
        const int r = ee - pl * (BN * 2);
        const int row = r >> 1, ch = r & 1;
        const int n = n0 + row;
        b_ok[j] = live && n < a.Cout;
        wbase[j] = (long)pl * a.wplane + (long)(n < a.Cout ? n : 0) * 16 + 8 * ch;      // fragment order: + stage block
        b_lds[j] = live ? 3 * APLANE + pl * BPLANE + chunk_off16(row, ch) : -1;
    }

    // DMA pieces of this wave: piece k = wave + 4 j (k < 18); lane -> linear chunk 64 k + lane of the B image
    int dma_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;   // waves 2,3 repeat pieces 16,17 (same bytes): no branch
        const int e = k * 64 + lane;
        const int pl = e / (BN * 2);
        const int r = e - pl * (BN * 2);
        const int row = r >> 1, ch = (r & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row < a.Cout ? n0 + row : 0;      // rows past Cout feed only columns that are never stored
        dma_src[j] = (int)(pl * a.wplane) + n * 16 + 8 * ch;      // fragment order (rdo::frag_index): + stage block
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = (a.Cin + KS - 1) / KS;
    // split-K: this workgroup reduces stages [sbeg, sbeg + nsteps) of the KH*KW*csteps total
    const int steps_total = a.KH * a.KW * csteps;
    const int sbeg = (int)((long)steps_total * tile.z / a.ksplit);
    const int nsteps = (int)((long)steps_total * (tile.z + 1) / a.ksplit) - sbeg;

    // load cursor (wave-uniform): tap and channel offset of the stage being loaded; per-thread pixel offset of that tap
    int ltap = sbeg / csteps;
    int lc = (sbeg - ltap * csteps) * KS;
    int lkh = ltap / a.KW, lkw = ltap - lkh * a.KW;
    int apix[2];
    bool apix_ok[2];
    auto retap = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int hi = hi0[t] + lkh, wi = wi0[t] + lkw;
            apix_ok[t] = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            apix[t] = abase[t] + (hi * a.W + wi) * a.Cin;
        }
    };
    auto advance = [&]() {      // move the cursor one stage forward (clamped at the last stage by the caller)
        lc += KS;
        if (lc >= a.Cin) {
            lc = 0;
            ++ltap;
            ++lkw;
            if (lkw == a.KW) { lkw = 0; ++lkh; }
            retap();
        }
    };
    retap();
    // stage block of (tap, channel slice) in the fragment-ordered planes: ((slice * taps + tap) * Cout) * 16 elements
    const int ktaps = a.KH * a.KW;
    auto wblock = [&](int tap, int c) { return (long)(((c / KS) * ktaps + tap) * a.Cout) * 16; };
    int btap = ltap, bc = lc;          // DMA cursor (one stage ahead of the load cursor)
    long bcur = wblock(btap, bc);
    auto badvance = [&]() {
        bc += KS;
        if (bc >= a.Cin) { bc = 0; ++btap; }
        bcur = wblock(btap, bc);
    };
    auto dma_b = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        char* dst = smem + buf * STAGE + 3 * APLANE + k * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(a.wp + dma_src[j] + bcur), (lds_void*)dst, 16, 0, 0);
    };

    f32x4 ra[2];
    bf16x8 rb[NB];
    bool ra_ok[2];
    auto load_a = [&](int t) {
        ra_ok[t] = apix_ok[t] && lc + 4 * aq < a.Cin;
        ra[t] = *reinterpret_cast<const f32x4*>(a.x + (ra_ok[t] ? apix[t] + lc : 0));
    };
    auto load_b = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const long off = wbase[j] + wblock(ltap, lc);
        rb[j] = *reinterpret_cast<const bf16x8*>(a.wp + (b_ok[j] ? off : 0));
    };
    auto store_a = [&](int t, int buf) {
        f32x4 v = ra_ok[t] ? ra[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.square_input) v = v * v;
        bf16x4 p0, p1, p2;
        split3_x4(v, p0, p1, p2);
        char* st = smem + buf * STAGE + a_lds[t];
        *reinterpret_cast<bf16x4*>(st) = p0;
        *reinterpret_cast<bf16x4*>(st + APLANE) = p1;
        *reinterpret_cast<bf16x4*>(st + 2 * APLANE) = p2;
    };
    auto store_b = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        const bf16x8 v = b_ok[j] ? rb[j] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        char* dst = b_lds[j] >= 0 ? smem + buf * STAGE + b_lds[j] : dummy + tid * 16;
        *reinterpret_cast<bf16x8*>(dst) = v;
    };
    auto all_b = [&](auto&& f) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
        (std::make_integer_sequence<int, NB>{});
    };

    // prologue: stage 0 -> LDS buffer 0; stage 1 -> registers
    load_a(0); load_a(1);
    if constexpr (DMA) all_b([&](auto jc) { dma_b(jc, 0); });
    else all_b([&](auto jc) { load_b(jc); });
    store_a(0, 0); store_a(1, 0);
    if constexpr (!DMA) all_b([&](auto jc) { store_b(jc, 0); });
    if (nsteps > 1) { advance(); badvance(); }
    load_a(0); load_a(1);
    if constexpr (!DMA) all_b([&](auto jc) { load_b(jc); });
    __syncthreads();

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    // fragment addresses are loop-invariant up to the buffer offset
    int fa_off[3][TM], fb_off[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_off[p][i] = p * APLANE + chunk_off16(wm0 + i * 32 + li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = 3 * APLANE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    }

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const char* st = smem + buf * STAGE;
        bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(st + fa_off[p][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(st + fb_off[p][j]);
        }
        if (s + 2 < nsteps) advance();     // cursor -> stage s+2 (uniform branch; tail stages re-load the last stage)
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[SL]][i], fb[PB[SL]][j], acc[i][j], 0, 0, 0);
                 if constexpr (DMA) {
                     // stage s+1's weight image goes straight to LDS (it must land before the barrier that ends this stage)
                     // the register-staged A quads are written before any DMA is issued: hipcc waits vmcnt(0) at a use of
                     // an ordinary load while a DMA is outstanding, and everything was drained at the last barrier anyway
                     if constexpr (SL == 0) {
                         store_a(0, buf ^ 1);
                         store_a(1, buf ^ 1);
                         load_a(0);
                         load_a(1);
                     }
                     if constexpr (SL < NB) dma_b(std::integral_constant<int, SL>{}, buf ^ 1);
                 } else if constexpr (SL == 0) {
                     store_a(0, buf ^ 1);
                     load_a(0);
                     store_a(1, buf ^ 1);
                     load_a(1);
                 } else {
                     store_b(std::integral_constant<int, SL - 1>{}, buf ^ 1);
                     load_b(std::integral_constant<int, SL - 1>{});
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 6>{});
        if (DMA && s + 2 < nsteps) badvance();
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + li;
        if (n >= a.Cout) continue;
        const float bv = (a.bias && !a.partial) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                const long o = (long)m * a.Cout + n;
                if (a.partial) a.partial[(long)tile.z * a.M * a.Cout + o] = acc[i][j][r];
                else finish<false>(a, o, acc[i][j][r] + bv);
            }
    }
}

#endif  // RDO_DIAG

// ---- v5: the v3/DMA tile with the fragment reads software-pipelined one stage ahead ---------------------------------------------
// Measured on v3: with loads and stores ablated the kernel still took 194 us against ~125 us of MFMA time -- every stage began with
// 15 exposed ds_read_b128 behind the barrier.  Here stage t runs its MFMAs from registers while it (1) reads stage t+1's fragments
// from LDS buffer (t+1)&1 as soon as a plane's registers die (plane 0 is double-buffered in registers, planes 1/2 are re-read in
// place), (2) writes stage t+2's image into buffer t&1 (A: split + ds_write, B: LDS-DMA) and (3) loads stage t+3's A quads.
// Slot order (A plane, B plane): (2,0) (0,2) (1,0) (1,1) (0,1) (0,0); fa2 is re-read after slot 0, fb2 after slot 1, fa1 after
// slot 3, fb1 after slot 4, and next stage's fa0/fb0 go to the alternate register set at the top of the stage.
template <bool HEAVY>
__global__ __launch_bounds__(256, 2) void conv_fwd_x6v5_kernel(X6Args a) {
    constexpr int BM = 128, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;
    constexpr int APLANE = BM * 32, BPLANE = BN * 32;
    constexpr int STAGE = 3 * (APLANE + BPLANE);           // 24 KiB
    constexpr int NB = 5;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;

    const int aq = tid & 3;
    int hi0[2], wi0[2], abase[2], a_lds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int arow = (tid >> 2) + 64 * t;
        const int m = m0 + arow;
        const bool okr = m < a.M;
        const int mm = okr ? m : 0;
        const int b = mm / (a.Ho * a.Wo);
        const int rem = mm - b * (a.Ho * a.Wo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[t] = okr ? ho * a.stride - a.pad : -(1 << 28);
        wi0[t] = wo * a.stride - a.pad;
        abase[t] = b * a.H * a.W * a.Cin + 4 * aq;
        a_lds[t] = chunk_off16(arow, aq >> 1) + (aq & 1) * 8;
    }
    int dma_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        const int e = k * 64 + lane;
        const int pl = e / (BN * 2);
        const int r = e - pl * (BN * 2);
        const int row = r >> 1, ch = (r & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row < a.Cout ? n0 + row : 0;
        dma_src[j] = (int)(pl * a.wplane) + n * 16 + 8 * ch;      // fragment order (rdo::frag_index): + stage block
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cin / KS;
    const int steps_total = a.KH * a.KW * csteps;
    const int sbeg = (int)((long)steps_total * tile.z / a.ksplit);
    const int nsteps = (int)((long)steps_total * (tile.z + 1) / a.ksplit) - sbeg;

    // Stage order: channel slice OUTER, the KH*KW taps INNER.  The nine taps of one 16-channel slice read three input rows that
    // overlap almost completely, so their activations stay in L1/L2 between uses; with taps outer every tap re-fetched its rows
    // through the fabric (rocprofv3 FETCH_SIZE: 444 MiB per launch for a 50 MB input = no reuse at all across the 9 taps).
    const int taps = a.KH * a.KW;
    int ls = sbeg;                       // absolute stage index of the load cursor
    int lc = 0;
    int apix[2];
    bool apix_ok[2];
    auto retap = [&]() {                 // cursor -> (tap, channel slice) of stage ls
        const int cs = ls / taps, tap = ls - cs * taps;
        const int lkh = tap / a.KW, lkw = tap - lkh * a.KW;
        lc = cs * KS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int hi = hi0[t] + lkh, wi = wi0[t] + lkw;
            apix_ok[t] = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            apix[t] = abase[t] + (hi * a.W + wi) * a.Cin;
        }
    };
    auto advance = [&]() {
        ++ls;
        retap();
    };
    retap();
    // stage s (relative to sbeg) in the fragment-ordered planes: stages are stored in exactly this kernel's order
    // (channel slice, tap), each a contiguous [Cout][16] block
    auto boff = [&](int s) { return (sbeg + s) * a.Cout * 16; };

    f32x4 ra[2];
    bool ra_ok[2];
    auto load_a = [&](int t) {
        ra_ok[t] = apix_ok[t];
        ra[t] = *reinterpret_cast<const f32x4*>(a.x + (ra_ok[t] ? apix[t] + lc : 0));
    };
    auto store_a = [&](int t, int buf) {
        f32x4 v = ra_ok[t] ? ra[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.square_input) v = v * v;
        bf16x4 p0, p1, p2;
        split3_x4(v, p0, p1, p2);
        char* st = smem + buf * STAGE + a_lds[t];
        *reinterpret_cast<bf16x4*>(st) = p0;
        *reinterpret_cast<bf16x4*>(st + APLANE) = p1;
        *reinterpret_cast<bf16x4*>(st + 2 * APLANE) = p2;
    };
    auto dma_b = [&](int j, int buf, int stage) {       // stage: index relative to sbeg, clamped by the caller
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        char* dst = smem + buf * STAGE + 3 * APLANE + k * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(a.wp + dma_src[j] + boff(stage)), (lds_void*)dst, 16, 0, 0);
    };

    int fa_off[3][TM], fb_off[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_off[p][i] = p * APLANE + chunk_off16(wm0 + i * 32 + li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = 3 * APLANE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    }
    // fragment registers: plane 0 double-buffered by stage parity, planes 1 and 2 single
    bf16x8 fa0[2][TM], fb0[2][TN], fa12[2][TM], fb12[2][TN];
    auto read_a = [&](bf16x8 (&dst)[TM], int p, int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(smem + buf * STAGE + fa_off[p][i]);
    };
    auto read_b = [&](bf16x8 (&dst)[TN], int p, int buf) {
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(smem + buf * STAGE + fb_off[p][j]);
    };
    auto mma = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    // prologue: images of stages 0 and 1 in LDS, A quads of stage 2 in flight, fragments of stage 0 in registers
    load_a(0); load_a(1);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_b(j, 0, 0);
    store_a(0, 0); store_a(1, 0);
    if (nsteps > 1) advance();
    load_a(0); load_a(1);
    {
        const int st1 = nsteps > 1 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) dma_b(j, 1, st1);
    }
    store_a(0, 1); store_a(1, 1);
    if (nsteps > 2) advance();
    load_a(0); load_a(1);
    __syncthreads();
    read_a(fa0[0], 0, 0); read_b(fb0[0], 0, 0);
    read_a(fa12[0], 1, 0); read_b(fb12[0], 1, 0);
    read_a(fa12[1], 2, 0); read_b(fb12[1], 2, 0);
    __syncthreads();      // every wave holds stage 0's fragments before buffer 0 is overwritten with stage 2

    auto stage = [&](auto parc, int t) {
        constexpr int PAR = decltype(parc)::value;          // t & 1: this stage's plane-0 register set; buffer t&1 is rewritten
        constexpr int NXT = PAR ^ 1;                        // LDS buffer holding stage t+1
        const int wst = t + 2 < nsteps ? t + 2 : nsteps - 1;
        // top: next stage's plane-0 fragments into the alternate set
        read_a(fa0[NXT], 0, NXT); read_b(fb0[NXT], 0, NXT);
        if (t + 3 < nsteps) advance();
        // slot 0: (A2, B0)
        mma(fa12[1], fb0[PAR]);
        read_a(fa12[1], 2, NXT);
        store_a(0, PAR); store_a(1, PAR);
        load_a(0); load_a(1);
        dma_b(0, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        // slot 1: (A0, B2)
        mma(fa0[PAR], fb12[1]);
        read_b(fb12[1], 2, NXT);
        dma_b(1, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        // slot 2: (A1, B0)
        mma(fa12[0], fb0[PAR]);
        dma_b(2, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        // slot 3: (A1, B1)
        mma(fa12[0], fb12[0]);
        read_a(fa12[0], 1, NXT);
        dma_b(3, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        // slot 4: (A0, B1)
        mma(fa0[PAR], fb12[0]);
        read_b(fb12[0], 1, NXT);
        dma_b(4, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        // slot 5: (A0, B0)
        mma(fa0[PAR], fb0[PAR]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    int t = 0;
    for (; t + 1 < nsteps; t += 2) {
        stage(std::integral_constant<int, 0>{}, t);
        stage(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nsteps) stage(std::integral_constant<int, 0>{}, t);

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + li;
        if (n >= a.Cout) continue;
        const float bv = (a.bias && !a.partial) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                const long o = (long)m * a.Cout + n;
                if (a.partial) a.partial[(long)tile.z * a.M * a.Cout + o] = acc[i][j][r];
                else finish<HEAVY>(a, o, acc[i][j][r] + bv);
            }
    }
}

// ---- v6: v5 for stride-1 convs with KW == 3: the three kw taps of a (channel slice, kh) pair share ONE activation image -------------
// The A tile of tap kw is the tile of tap kw-1 shifted by one pixel, so instead of loading and splitting 128 rows per stage the
// workgroup keeps the halo rows of its output pixels -- R image rows of Wt + 2 pixels, R * Wt = 128 -- in LDS per GROUP of three
// stages and the fragment reads of tap kw start kw rows further down: a third of the activation loads, splits and LDS writes.
// Stage order (channel slice, kh, kw); K is split at group boundaries.  Requirements (host-checked): stride 1, KW == 3, pad == 1,
// Cin % 16 == 0, M % 128 == 0 and W either a multiple of 128 or a divisor of it with H * W % 128 == 0.
__global__ __launch_bounds__(256, 2) void conv_fwd_x6v6_kernel(X6Args a) {
    constexpr int BM = 128, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;
    constexpr int AROWS = 144;                               // halo rows per image: 8 x 18 (W = 16) is the largest case
    constexpr int APLANE = AROWS * 32, BPLANE = BN * 32;
    constexpr int ABUF = 3 * APLANE, BBUF = 3 * BPLANE;
    constexpr int NB = 5, NA = 3;                            // DMA pieces / activation quads per thread

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][ABUF] [2][BBUF]
    char* const sA = smem;
    char* const sB = smem + 2 * ABUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;

    // tile geometry: 128 consecutive output pixels = R rows of Wt pixels of one image
    const int Wt = a.W < 128 ? a.W : 128, Wh = Wt + 2, R = 128 / Wt, NQ = R * Wh;
    const int b0 = m0 / (a.H * a.W);
    const int rem0 = m0 - b0 * (a.H * a.W);
    const int ho0 = rem0 / a.W, wo0 = rem0 - ho0 * a.W;

    // activation quads of this thread: e = tid + 256 j over [halo row q][quad of 4 channels]
    int a_off[NA], a_hi[NA], a_lds[NA];
    bool a_live[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + 256 * j;
        const int q = e >> 2, aq = e & 3;
        const int r = q / Wh, wq = q - r * Wh;
        const int wi = wo0 + wq - 1;
        a_live[j] = q < NQ && (unsigned)wi < (unsigned)a.W;
        a_hi[j] = ho0 + r - a.pad;                           // + kh
        a_off[j] = ((b0 * a.H + a_hi[j]) * a.W + wi) * a.Cin + 4 * aq;      // + kh * W * Cin + channel slice
        a_lds[j] = q < AROWS ? chunk_off16(q, aq >> 1) + (aq & 1) * 8 : -1;
    }
    int dma_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        const int e = k * 64 + lane;
        const int pl = e / (BN * 2);
        const int r = e - pl * (BN * 2);
        const int row = r >> 1, ch = (r & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row < a.Cout ? n0 + row : 0;
        dma_src[j] = (int)(pl * a.wplane) + n * 16 + 8 * ch;      // fragment order (rdo::frag_index): + stage block
    }
    // halo row of this lane's output pixels (tap kw reads row hb + kw)
    int hb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm0 + i * 32 + li;
        const int r = ml / Wt;
        hb[i] = r * Wh + (ml - r * Wt);
    }
    int fb_off[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cin / KS;
    const int groups_total = csteps * a.KH;                  // group = (channel slice, kh); three stages (kw) each
    const int gbeg = (int)((long)groups_total * tile.z / a.ksplit);
    const int ngroups = (int)((long)groups_total * (tile.z + 1) / a.ksplit) - gbeg;
    const int nsteps = 3 * ngroups;

    f32x4 ra[NA];
    bool ra_ok[NA];
    auto load_a = [&](int j, int g) {                         // quad j of group gbeg + g (clamped by the caller)
        const int G = gbeg + g;
        const int cs = G / a.KH, kh = G - cs * a.KH;
        ra_ok[j] = a_live[j] && (unsigned)(a_hi[j] + kh) < (unsigned)a.H;
        ra[j] = *reinterpret_cast<const f32x4*>(a.x + (ra_ok[j] ? a_off[j] + kh * a.W * a.Cin + cs * KS : 0));
    };
    auto store_a = [&](int j, int abuf) {
        if (a_lds[j] < 0) return;
        f32x4 v = ra_ok[j] ? ra[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.square_input) v = v * v;
        bf16x4 p0, p1, p2;
        split3_x4(v, p0, p1, p2);
        char* st = sA + abuf * ABUF + a_lds[j];
        *reinterpret_cast<bf16x4*>(st) = p0;
        *reinterpret_cast<bf16x4*>(st + APLANE) = p1;
        *reinterpret_cast<bf16x4*>(st + 2 * APLANE) = p2;
    };
    // stage t (relative to the first stage of this workgroup) in the fragment-ordered planes: group (channel slice, kh) and kw
    // are stored in exactly this order, each stage a contiguous [Cout][16] block
    auto boff = [&](int t) { return (gbeg * 3 + t) * a.Cout * 16; };
    auto dma_b = [&](int j, int buf, int t) {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        char* dst = sB + buf * BBUF + k * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(a.wp + dma_src[j] + boff(t)), (lds_void*)dst, 16, 0, 0);
    };

    bf16x8 fa0[2][TM], fb0[2][TN], fa12[2][TM], fb12[2][TN];
    auto read_a = [&](bf16x8 (&dst)[TM], int p, int abuf, int kw) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            dst[i] = *reinterpret_cast<const bf16x8*>(sA + abuf * ABUF + p * APLANE + chunk_off16(hb[i] + kw, lh));
    };
    auto read_b = [&](bf16x8 (&dst)[TN], int p, int buf) {
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(sB + buf * BBUF + fb_off[p][j]);
    };
    auto mma = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    // prologue: activation image of group 0 in A buffer 0, quads of group 1 in registers, weight images of stages 0 and 1
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_b(j, 0, 0);
#pragma unroll
    for (int j = 0; j < NA; ++j) store_a(j, 0);
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j, ngroups > 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_b(j, 1, 1);
    __syncthreads();
    read_a(fa0[0], 0, 0, 0); read_b(fb0[0], 0, 0);
    read_a(fa12[0], 1, 0, 0); read_b(fb12[0], 1, 0);
    read_a(fa12[1], 2, 0, 0); read_b(fb12[1], 2, 0);
    __syncthreads();

    // one stage: MFMAs of stage t from registers | fragments of stage t+1 | weight image of stage t+2 | the group's share of the
    // activation work (kw 0: two quads of the next group's image, kw 1: the third; kw 2: loads of the group after next)
    auto stage = [&](auto parc, auto kwc, int t, int g) {
        constexpr int PAR = decltype(parc)::value, NXT = PAR ^ 1, KW = decltype(kwc)::value;
        const int ab = g & 1;                                        // A buffer of this stage's group
        const int an = KW == 2 ? ab ^ 1 : ab, kn = KW == 2 ? 0 : KW + 1;   // buffer / tap of stage t+1
        const int wst = t + 2 < nsteps ? t + 2 : nsteps - 1;
        read_a(fa0[NXT], 0, an, kn); read_b(fb0[NXT], 0, NXT);
        // slot 0: (A2, B0)
        mma(fa12[1], fb0[PAR]);
        read_a(fa12[1], 2, an, kn);
        if constexpr (KW == 0) { store_a(0, ab ^ 1); store_a(1, ab ^ 1); }
        if constexpr (KW == 1) store_a(2, ab ^ 1);
        if constexpr (KW == 2) {
            const int g2 = g + 2 < ngroups ? g + 2 : ngroups - 1;
#pragma unroll
            for (int j = 0; j < NA; ++j) load_a(j, g2);
        }
        dma_b(0, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0[PAR], fb12[1]);
        read_b(fb12[1], 2, NXT);
        dma_b(1, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa12[0], fb0[PAR]);
        dma_b(2, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa12[0], fb12[0]);
        read_a(fa12[0], 1, an, kn);
        dma_b(3, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0[PAR], fb12[0]);
        read_b(fb12[0], 1, NXT);
        dma_b(4, PAR, wst);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0[PAR], fb0[PAR]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // two groups (six stages) per trip: the stage parity -- which plane-0 register set and weight buffer a stage uses -- is back
    // where it started, so no control-flow join mixes the two register assignments
    int g = 0;
    for (; g + 1 < ngroups; g += 2) {
        const int t = 3 * g;
        stage(I0{}, I0{}, t, g); stage(I1{}, I1{}, t + 1, g); stage(I0{}, I2{}, t + 2, g);
        stage(I1{}, I0{}, t + 3, g + 1); stage(I0{}, I1{}, t + 4, g + 1); stage(I1{}, I2{}, t + 5, g + 1);
    }
    if (g < ngroups) {
        const int t = 3 * g;
        stage(I0{}, I0{}, t, g); stage(I1{}, I1{}, t + 1, g); stage(I0{}, I2{}, t + 2, g);
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + li;
        if (n >= a.Cout) continue;
        const float bv = (a.bias && !a.partial) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                const long o = (long)m * a.Cout + n;
                if (a.partial) a.partial[(long)tile.z * a.M * a.Cout + o] = acc[i][j][r];
                else finish<false>(a, o, acc[i][j][r] + bv);
            }
    }
}

// split-K second pass of the v3 kernel: sum the partial accumulators, bias, epilogue
__global__ __launch_bounds__(256) void x6_splitk_epilogue_kernel(X6Args a) {
    const long total = (long)a.M * a.Cout;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < a.ksplit; ++z) v += a.partial[(long)z * total + o];
        if (a.bias) v += a.bias[o % a.Cout];
        finish<true>(a, o, v);
    }
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* w, long n, u16* planes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        u16 h, m, l;
        split3(w[i], h, m, l);
        planes[i] = h; planes[n + i] = m; planes[2 * n + i] = l;
    }
}

__global__ __launch_bounds__(256) void split_bf16x3_conv_kernel(const float* w, long n, int Cout, int KH, int KW, int Cin, u16* planes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        u16 h, m, l;
        split3(w[i], h, m, l);
        const long f = rdo::frag_index(i, Cout, KH, KW, Cin);
        planes[f] = h; planes[n + f] = m; planes[2 * n + f] = l;
    }
}

}  // namespace

extern "C" int rdo_split_bf16x3_conv(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, void* planes, void* stream) {
    RDO_REQUIRE(w && planes && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "rdo_split_bf16x3_conv: bad argument");
    u16* p = reinterpret_cast<u16*>(planes);
    const long n = (long)Cout * KH * KW * Cin;
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(n, 256);
            hipLaunchKernelGGL(split_bf16x3_conv_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, w, n, Cout, KH, KW, Cin, p);
            return rdo::check_launch("split_bf16x3_conv");
        },
        stream, "split_bf16x3", 0.0, 10.0 * n);
}

extern "C" int rdo_split_bf16x3(const float* w, int64_t n, void* planes, void* stream) {
    RDO_REQUIRE(w && planes && n > 0, "rdo_split_bf16x3: bad argument");
    u16* p = reinterpret_cast<u16*>(planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(n, 256);
            hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, w, (long)n, p);
            return rdo::check_launch("split_bf16x3");
        },
        stream, "split_bf16x3", 0.0, 10.0 * n);
}

// split factor for a shape: enough 128 x 192 x (K/ks) workgroups to fill the chip (>= 256, preferably >= 512), at least
// 12 K stages (of 16 channels) per split, at most 16 splits; 0 = the shape is too small for this path
extern "C" int rdo_conv2d_fwd_bf16x6_ksplit(const rdo_conv_desc* d) {
    if (!d || d->Cin % 16 != 0 || d->Cout < 160) return 0;
    const long M = (long)d->B * d->Ho * d->Wo;
    const long tiles = rdo::ceil_div(M, 128) * rdo::ceil_div(d->Cout, 192);
    const long stages = (long)d->KH * d->KW * (d->Cin / 16);
    // two co-resident workgroups per CU: 512 fill the chip evenly (measured on 128 tiles: ks=4 -> 80 us, ks=3 -> 87 us)
    int best = 0;
    for (int ks = 1; ks <= 16; ++ks)
        if (tiles * ks >= 256 && stages / ks >= 12) {
            best = ks;
            if (tiles * ks >= 512) break;
        }
    return best;
}

static int fwd_bf16x6_impl(const rdo_conv_desc* d, const float* x, const void* wplanes, const float* bias, const float* aux,
                           const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats, void* stream,
                           bool partial_only) {
    RDO_REQUIRE(d && x && wplanes && (out || partial_only), "rdo_conv2d_fwd_bf16x6: null argument");
    RDO_REQUIRE(d->Cin % 16 == 0, "rdo_conv2d_fwd_bf16x6: Cin must be a multiple of 16");
    X6Args a{};
    a.x = x; a.wp = reinterpret_cast<const u16*>(wplanes); a.bias = bias; a.aux = aux; a.residual = residual; a.out = out; a.pre = pre;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.csteps = (d->Cin + BK - 1) / BK;
    a.epilogue = d->epilogue; a.square_input = d->square_input; a.add_residual = d->add_residual;
    a.wplane = (long)d->Cout * d->KH * d->KW * d->Cin;
    a.xcd_mode = rdo::tuning(rdo::T_XCD);
    int ks = rdo_conv2d_fwd_bf16x6_ksplit(d);
    if (ks < 1) ks = 1;
    if (ks > 1 && (!workspace || (long)ks * a.M * a.Cout > workspace_floats)) ks = 1;
    a.ksplit = ks;
    a.partial = ks > 1 ? workspace : nullptr;
    a.partial_only = partial_only ? 1 : 0;
    RDO_REQUIRE(!partial_only || ks > 1, "rdo_conv2d_fwd_bf16x6: partial sums requested for a shape that is not split over K");
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    return rdo::dispatch(
        [a](hipStream_t s) {
            // default: v6 where its geometry applies, else v5 ("fwd_x6_ver" = 5 forces v5; 3 / 4, the earlier register-staged / LDS-DMA
            // variants of the same tile, exist only in a DIAG build).  The LDS-DMA loaders address the weight planes with 32-bit
            // element offsets: rdo_conv2d_fwd_uses_bf16x6 refuses larger weights.
            const int ver_env = rdo::tuning(rdo::T_FWD_X6_VER);
#ifdef RDO_DIAG
            const int ver = ver_env >= 5 ? 5 : (ver_env < 3 ? 3 : ver_env);
#else
            constexpr int ver = 5;
#endif
            // v6 (the three kw taps share one activation image) takes the stride-1 3-wide kernels whose tiles are whole image rows
            const bool v6_ok = a.epilogue < RDO_EPI_GELU && a.stride == 1 && a.KW == 3 && a.pad == 1 && a.Cin % 16 == 0 && a.M % 128 == 0 && a.W >= 16 &&
                               (a.W % 128 == 0 || (128 % a.W == 0 && (a.H * a.W) % 128 == 0)) &&
                               a.ksplit <= (a.Cin / 16) * a.KH;
            if (ver_env >= 6 && v6_ok) {
                constexpr size_t lds6 = (size_t)2 * 3 * 144 * 32 + (size_t)2 * 3 * 192 * 32;
                static rdo::PerDevice attr6;
                if (!attr6.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_x6v6_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6 v6) failed");
                    attr6.mark();
                }
                dim3 grid6((unsigned)rdo::ceil_div(a.M, 128), (unsigned)rdo::ceil_div(a.Cout, 192), (unsigned)a.ksplit);
                hipLaunchKernelGGL(conv_fwd_x6v6_kernel, grid6, dim3(256), lds6, s, a);
                if (int rc = rdo::check_launch("conv_fwd_x6v6")) return rc;
                if (a.ksplit > 1 && !a.partial_only) {
                    long g = rdo::ceil_div((long)a.M * a.Cout, 256);
                    hipLaunchKernelGGL(x6_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
                    return rdo::check_launch("x6_splitk_epilogue");
                }
                return RDO_OK;
            }
            const size_t lds = (size_t)2 * 3 * (128 + 192) * 32 + (ver == 5 ? 0 : 4096);
            dim3 grid((unsigned)rdo::ceil_div(a.M, 128), (unsigned)rdo::ceil_div(a.Cout, 192), (unsigned)a.ksplit);
#ifdef RDO_DIAG
            // the diagnostic build has only the plain instantiations: an unsplit launch with a GELU epilogue would return the
            // pre-activation without a word (conv_fwd.hip rejects the equivalent case the same way)
            if (a.epilogue >= RDO_EPI_GELU && a.ksplit == 1)
                return rdo::set_error(RDO_EINVAL, "rdo_conv2d_fwd_bf16x6: the GELU epilogues of an unsplit launch are not compiled into a diagnostic build");
            const void* kern = ver == 3 ? reinterpret_cast<const void*>(conv_fwd_x6v3_kernel<false>)
                             : ver == 4 ? reinterpret_cast<const void*>(conv_fwd_x6v3_kernel<true>)
                                        : reinterpret_cast<const void*>(conv_fwd_x6v5_kernel<false>);
            static rdo::PerDevice attr[3];
            if (!attr[ver - 3].done()) {
                if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6 v%d) failed", ver);
                attr[ver - 3].mark();
            }
            if (ver == 3) hipLaunchKernelGGL(conv_fwd_x6v3_kernel<false>, grid, dim3(256), lds, s, a);
            else if (ver == 4) hipLaunchKernelGGL(conv_fwd_x6v3_kernel<true>, grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL(conv_fwd_x6v5_kernel<false>, grid, dim3(256), lds, s, a);
#else
            // the GELU epilogues of an unsplit launch: their own instantiation (a split launch applies them in its second pass)
            const bool heavy = a.epilogue >= RDO_EPI_GELU && a.ksplit == 1;
            static rdo::PerDevice attr5[2];
            if (!attr5[heavy].done()) {
                const void* k5 = heavy ? reinterpret_cast<const void*>(conv_fwd_x6v5_kernel<true>) : reinterpret_cast<const void*>(conv_fwd_x6v5_kernel<false>);
                if (hipFuncSetAttribute(k5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6 v5) failed");
                attr5[heavy].mark();
            }
            if (heavy) hipLaunchKernelGGL(conv_fwd_x6v5_kernel<true>, grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL(conv_fwd_x6v5_kernel<false>, grid, dim3(256), lds, s, a);
#endif
            if (int rc = rdo::check_launch("conv_fwd_x6")) return rc;
            if (a.ksplit > 1 && !a.partial_only) {
                long g = rdo::ceil_div((long)a.M * a.Cout, 256);
                hipLaunchKernelGGL(x6_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
                return rdo::check_launch("x6_splitk_epilogue");
            }
            return RDO_OK;
        },
        stream, "conv_fwd_x6_128x192", flops,
        4.0 * ((double)a.B * a.H * a.W * a.Cin + (double)a.M * a.Cout) + 6.0 * a.wplane);
}

extern "C" int rdo_conv2d_fwd_bf16x6(const rdo_conv_desc* d, const float* x, const void* wplanes, const float* bias, const float* aux,
                                     const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats,
                                     void* stream) {
    return fwd_bf16x6_impl(d, x, wplanes, bias, aux, residual, out, pre, workspace, workspace_floats, stream, false);
}

int rdo_conv2d_fwd_bf16x6_partials(const rdo_conv_desc* d, const float* x, const void* wplanes, float* workspace, int64_t workspace_floats,
                                   void* stream) {
    return fwd_bf16x6_impl(d, x, wplanes, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_floats, stream, true);
}
