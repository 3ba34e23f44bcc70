// Halo forward / dgrad conv on H2 tensors, second form ("K32"): 3 x 3, stride 1, pad 1 convs whose H and W are multiples of 16 and whose
// Cin is a multiple of 32.  Same tile as conv_fwd_h2h_kernel (conv_fwd_h2.hip) -- the 256 output pixels of a workgroup are a 16 x 16
// patch of one image, the 18 x 18 halo of the patch is fetched once per channel group and serves all nine taps, eight waves, one
// workgroup per CU -- but
//   * a K stage is 32 input channels (two slices of the H2 layout) of one tap: half as many stages, barriers and loop skeletons
//     (measured on the 16-channel kernel: 229 of the 1 704 cycles of a stage were barrier + skeleton with nothing else in it);
//   * the products run on v_mfma_f32_16x16x32_f16: under the board's power limit the chip holds a higher clock on that shape than on
//     32x32x16 at the same cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7), and these kernels run AT that limit
//     (1.90 GHz in the K loop of the 16-channel kernel);
//   * A = weights (rows = output channels), B = activations (columns = pixels): a lane's four accumulator values are four consecutive
//     CHANNELS of one pixel, so the accumulators go to the epilogue's staging tile as 16-byte stores (the 32 x 32 form: 4-byte stores).
// LDS images, per plane [slice of 16 channels][row][32 bytes] with row = halo pixel q (activations: 324 rows per slice) or output channel
// (weights: blocks of 16 channels, each [slice][16 channels][32 B] = one 1 KiB DMA piece); the k-groups of the MFMA (8 input channels each)
// are the two halves of a row's 32 bytes in the two slices.  A ds_read_b128 of a fragment -- 16 consecutive rows, any start row, x 4
// k-groups -- touches all 64 banks once in each of its four lane groups without a swizzle, and sixteen DMA lanes read 256+ contiguous
// bytes of one slice (a 1 KiB piece in 16 cycles of the texture addresser; the [row][4 k-groups] image of round 3 took 32).
// LDS: two halo buffers of 42 KiB (slice pair cs and cs + 1), 1 KiB that swallows the DMA pieces beyond the image, the three-deep weight
// ring of 24 KiB (BN = 192; 8 KiB for BN = 64): 158 KiB / 110 KiB; the epilogue stages 128 x (BN + 4) floats in the same memory.
// DMA per wave: per stage its weight pieces of stage s + 2 (three for BN = 192, one for BN = 64) and, in taps 0-5, one of its six pieces of
// the next slice pair's halo, issued before the weight pieces -- the counted wait at the top of a stage is vmcnt(BJ + 1) in taps 1-6 and
// vmcnt(BJ) otherwise.
// Products of a stage in the order (X1, W0) (X0, W1) (X0, W0), small terms first, accumulated into the same fp32 registers as the
// 16-channel kernel accumulates them: per output element the same three-product sums in the same stage order, K = 32 per instruction
// instead of 16 -- results agree with the 16-channel kernel to the rounding of the MFMA's internal sums (tests/test_gpu_h2.py).
#include "conv_h2_common.h"

namespace {

typedef float f32x4k __attribute__((ext_vector_type(4)));
#ifdef RDO_DIAG
__device__ unsigned long long g_h2k_stamps[256 * 4];
#endif

// Unit tail on the four finished channels [n, n + 4) of a pixel (v = conv + bias, the pre-activation): v becomes dL/dpre, the return value
// is sum d^2.  The arithmetic of tail16 (conv_h2_common.h) / loss_act_quad (fused_tail.hip), operation for operation per element.  Its
// inputs -- the target quad y4 and the residual's two plane quads p0 / p1 -- are fetched by the caller, a whole row of tiles ahead.
typedef unsigned u32x2k __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tail4_math(const H2Args& a, f32x4& v, const f32x4& y4, const u32x2k& p0, const u32x2k& p1) {
    const float slope = a.tail_act == 1 ? 0.01f : 0.f;
    const float gs = a.tail_coef * 2.f * a.tail_inv_npix;
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.tail_resp) {                                       // the residual from its planes: (h1 + h2) / s, the sum is exact in fp32
        r[0] = (h2_lo(p0[0]) + h2_lo(p1[0])) * a.tail_res_inv;
        r[1] = (h2_hi(p0[0]) + h2_hi(p1[0])) * a.tail_res_inv;
        r[2] = (h2_lo(p0[1]) + h2_lo(p1[1])) * a.tail_res_inv;
        r[3] = (h2_hi(p0[1]) + h2_hi(p1[1])) * a.tail_res_inv;
    }
    float dd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float p = v[k];
        const bool act = a.tail_act != 0, res = a.tail_resp != nullptr;      // wave-uniform: selects, not branches per element
        float o = act ? (p > 0.f ? p : slope * p) : p;
        o = res ? o + r[k] : o;
        dd[k] = o - y4[k];
        const float g = dd[k] * gs;
        v[k] = act ? (p > 0.f ? g : slope * g) : g;
    }
    return (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
}

// WM x WN waves (8), wave tile TP pixel tiles x TC channel tiles of 16 x 16: WM * TP = 16 (256 patch pixels), BN = WN * TC * 16
// ABL (diagnostic builds only, `make DIAG=1`; the shipped library instantiates ABL = 0): compile-time ablation mask -- 1 no halo DMA,
// 2 no weight DMA, 4 no MFMA, 8 no fragment reads, 16 no epilogue.  Results are wrong when non-zero.
template <int WM, int WN, int TP, int TC, int ABL = 0, int ORD = 0>
__global__ __launch_bounds__(512, 2) void conv_fwd_h2k_kernel(H2Args a) {
    static_assert(WM * WN == 8 && WM * TP == 16, "eight waves, 256 patch pixels");
    constexpr int BN = WN * TC * 16;
    constexpr int PT = 16, HWD = PT + 2, HPIX = HWD * HWD;   // patch edge, halo edge, halo pixels (324)
    constexpr int APL = HPIX * 64;                           // one plane of the halo tile: 20 736 B
    constexpr int ACHUNKS = NP * HPIX * 4;                   // 16-byte chunks of the two planes: 2 592 (40.5 wave instructions)
    constexpr int AJ = 6;                                    // halo pieces per wave (8 * 6 = 48 >= 41)
    constexpr int APIECES = (ACHUNKS + 63) / 64;             // 41
    constexpr int ABUF = 42 * 1024;
    constexpr int ADUMMY = 2 * ABUF;                         // 1 KiB: destination of the pieces beyond the image
    constexpr int BBASE = ADUMMY + 1024;
    constexpr int BPLANE = BN * 64, BSTAGE = NP * BPLANE;    // 24 KiB (BN = 192) / 8 KiB
    constexpr int BPIECES = BSTAGE / 1024;                   // 24 / 8
    constexpr int BJ = (BPIECES + 7) / 8;                    // weight pieces per wave and stage: 3 / 1 (BN = 48: six pieces, waves 6 and 7 fetch a dummy)
    constexpr int RING = 3;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave / WN, wcol = wave - wrow * WN;
    const int wn0 = wcol * (TC * 16);
    const int l16 = lane & 15, kg = lane >> 4;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int n0 = tile.n * BN;
    const int pw_n = a.W / PT, ph_n = a.H / PT;
    const int pb = tile.m / (ph_n * pw_n);
    const int prem = tile.m - pb * (ph_n * pw_n);
    const int h0 = (prem / pw_n) * PT, w0 = (prem % pw_n) * PT;
    const int pbase = (pb * a.H + h0) * a.W + w0;            // first pixel of the patch
    const long Min = (long)a.B * a.H * a.W;

    // ---- LDS-DMA through buffer descriptors (`buffer_load_dwordx4 ... lds`): a lane whose offset fails the range check writes ZEROS,
    // which is the padding around the image (tools/micro/buf_lds.hip); the per-stage part of an address is a scalar offset.
    constexpr unsigned OOB = 0xFFFFFFFFu;
    constexpr int RSRC3 = 0x00020000;                        // raw buffer, 32-bit data format
    __amdgpu_buffer_rsrc_t rsrc[2];                          // [0] activations, [1] weights
    rsrc[0] = __builtin_amdgcn_make_buffer_rsrc((void*)a.xp, 0, (unsigned)(a.xplane * 4), RSRC3);   // both planes
    rsrc[1] = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (unsigned)(a.wplane * 4), RSRC3);
    // ---- halo loader: pieces wave, wave + 8, ... of the [plane][slice][halo pixel][32 B] image: lane e of the image is 16 bytes (half a
    // slice record) of halo pixel (e % 648) / 2 of block e / 648.  Consecutive lanes walk a halo row (18 pixels = 576 contiguous bytes
    // of the source): the texture addresser takes such a piece in 16 cycles against 32 for the [pixel][4 k-groups] image of round 3,
    // whose lane quads straddled the two slices (tools/micro/dma_rate.hip).  The lane's offset of a piece is recomputed when the piece
    // is issued (once per stage at most: ~20 vector instructions) -- six such offsets kept in registers for the whole K loop were what
    // spilled the 256-register budget of the 256 x 192 tile.
    auto halo_off = [&](int piece) -> unsigned {             // byte offset into a.xp for slice pair 0, or OOB: padding
        const int e = piece * 64 + lane;
        const int blk = (e * 6473) >> 22;                    // e / 648 for e < 3 072
        const int r = e - blk * (HPIX * 2);
        const int q = r >> 1;
        const int hr = (q * 3641) >> 16;                     // q / 18 for q < 1 296
        const int hc = q - hr * HWD;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
        const bool ok = e < ACHUNKS && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const unsigned off = (unsigned)((blk >> 1) * a.xplane + ((blk & 1) * Min + ((pb * a.H + hi) * a.W + wi)) * 16) * 2u + (r & 1) * 16u;
        return ok ? off : OOB;
    };
    // ---- weight loader: pieces wave, wave + 8, ... of the [plane][16 channels][slice][channel][32 B] image of a stage: piece k is plane
    // k / (BN / 16), channels 16 (k % (BN / 16)) ...: a wave-uniform part per piece and ONE per-lane part (slice = lane >> 5: 9 Cout
    // records further in the fragment-ordered planes; channel (lane >> 1) & 15; half record lane & 1) -- 512 contiguous bytes per slice
    const unsigned dma_lane = (unsigned)((lane >> 5) * 9 * a.Cout * 16 + ((lane >> 1) & 15) * 16 + (lane & 1) * 8) * 2u;
    auto dma_uni = [&](int j) -> unsigned {                  // wave-uniform part of piece wave + 8 j (beyond the image: piece 0 again), bytes
        const int k = wave + 8 * j < BPIECES ? wave + 8 * j : 0;
        const int pl = k / (BN / 16), kk = k - pl * (BN / 16);
        return (unsigned)(pl * a.wplane + (n0 + kk * 16) * 16) * 2u;
    };

    f32x4k acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4k{0.f, 0.f, 0.f, 0.f};

    // K split (a.ksplit > 1: few tiles, e.g. the 32^2 convs): workgroup z owns the slice pairs [cp0, cp1) and writes raw partial sums
    const int cpairs_all = a.Cin / 32;
    const int cp0 = cpairs_all * tile.z / a.ksplit, cp1 = cpairs_all * (tile.z + 1) / a.ksplit;
    const int cpairs = cp1 - cp0;
    const int nstages = 9 * cpairs;

    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_a = [&](int j, int cp_next) {                   // piece j of the wave, halo of slice pair cp_next into buffer cp_next & 1
        const int cn = cp0 + (cp_next < cpairs ? cp_next : cpairs - 1);
        const int piece = wave + 8 * j;
        const unsigned off = halo_off(piece);
        char* dst = piece < APIECES ? smem + (cp_next & 1) * ABUF + piece * 1024 : smem + ADUMMY;
        if constexpr (!(ABL & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[0], (lds_void*)dst, 16, off, (unsigned)cn * (unsigned)(Min * 64), 0, 0);
    };
    unsigned wsoff = 0;
    auto dma_b = [&](int j, int nb) {
        char* dst = wave + 8 * j < BPIECES ? smem + BBASE + nb * BSTAGE + (wave + 8 * j) * 1024 : smem + ADUMMY;
        const unsigned so = wsoff + dma_uni(j);
        if constexpr (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc[1], (lds_void*)dst, 16, dma_lane, so, 0, 0);
    };
    // stage s = slice pair s / 9, tap s % 9: its weights start at slice 2 (s / 9), tap s % 9 of the fragment-ordered planes
    auto set_wsrc = [&](int stage) {
        const int st = stage < nstages ? stage : nstages - 1;
        const int cp = st / 9, tap = st - cp * 9;
        wsoff = (unsigned)(((2 * (cp0 + cp)) * 9 + tap) * a.Cout * 16) * 2u;
    };

    // fragment addresses.  Weights: row = wn0 + 16 i + l16 with wn0 and 16 i multiples of 8, so the swizzle depends on the lane only:
    // ONE base register, (plane, tile) are immediate offsets.  Halo: q = (patch row + kh) * 18 + l16 + kw, swizzled per (tile, tap):
    // recomputed every stage from `qlane` (kept opaque to the optimiser there: hoisted out of the slice-pair loop the 9 x TP addresses
    // would cost 36 registers and spill the accumulators).
    const int fw_base = BBASE + (wn0 >> 4) * 1024 + (kg >> 1) * 512 + l16 * 32 + (kg & 1) * 16;
    const int fx_lane = (kg >> 1) * (HPIX * 32) + (kg & 1) * 16;
    int qlane = (wrow * TP) * HWD + l16;

    // prologue: halo of slice pair 0, weights of stages 0 and 1
#pragma unroll
    for (int j = 0; j < AJ; ++j) dma_a(j, 0);
    set_wsrc(0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 0);
    set_wsrc(1);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 1);

    // All eight waves run the K loop in lock-step.  The staggered SIMD partners of the 16-channel kernel (waves 4-7 two thirds of a
    // stage behind, conv_fwd_h2.hip) measured 2.5-8 % SLOWER here (same-process A/B, 4 x 128^2: 105.3 vs 102.6 us; 4 x 64^2: 37.8 vs
    // 34.9): with 72 MFMAs per wave and stage the two partners of a SIMD drift apart by themselves.
#ifdef RDO_DIAG
    const unsigned long long st_c0 = clock64(), st_r0 = wall_clock64();
#endif
    {
        f16x8 fx[NP][TP], fw[NP][TC];
        auto mma = [&](int px, int pw) {
            if constexpr (ABL & 4) {                         // keep the fragments alive without multiplying
#pragma unroll
                for (int i = 0; i < TC; ++i) asm volatile("" ::"v"(fw[pw][i]));
#pragma unroll
                for (int j = 0; j < TP; ++j) asm volatile("" ::"v"(fx[px][j]));
                return;
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[pw][i], fx[px][j], acc[i][j], 0, 0, 0);
        };
        int buf = 0, s = 0;
        auto stage = [&](auto tapc, int cp) {
            constexpr int TAP = decltype(tapc)::value;
            constexpr int KH = TAP / 3, KW = TAP % 3;
            // stage s (weights) and, in tap 0, this slice pair's halo have landed once at most the DMAs issued after them are outstanding
            if constexpr (ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (TAP >= 1 && TAP <= AJ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nb = buf >= 1 ? buf - 1 : RING - 1;    // (s + 2) % 3
            const char* stw = smem + buf * BSTAGE;
            const char* stx = smem + (cp & 1) * ABUF;
            asm volatile("" : "+v"(qlane));                  // (see above: keeps the address arithmetic inside the stage)
            int fx_off[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int q = qlane + (j + KH) * HWD + KW;
                fx_off[j] = q * 32 + fx_lane;
            }
            auto rd_x = [&](int p) {
                if constexpr (ABL & 8) return;
#pragma unroll
                for (int j = 0; j < TP; ++j) fx[p][j] = *reinterpret_cast<const f16x8*>(stx + p * APL + fx_off[j]);
            };
            auto rd_w = [&](int p) {
                if constexpr (ABL & 8) return;
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[p][i] = *reinterpret_cast<const f16x8*>(stw + fw_base + p * BPLANE + i * 1024);
            };
            set_wsrc(s + 2);
            // DMA slots of this wave in a stage: slot 0 = the halo piece (taps 0-5) + the first weight piece, slot 1 = the other weight pieces
            auto slot = [&](int k) {
                if (k == 0) {
                    if constexpr (TAP < AJ) dma_a(TAP, cp + 1);
                    dma_b(0, nb);
                } else {
#pragma unroll
                    for (int j = 1; j < BJ; ++j) dma_b(j, nb);
                }
            };
            // Program order of a stage; NO scheduling fences: hipcc's own interleaving of the 20 fragment reads, the 3-4 DMA issues and the
            // 72 MFMAs measured 1.8 % faster than the same order with a sched_barrier around every MFMA group, 2.8 % faster than all reads up
            // front or the DMA issue first (ORD 1-3: diagnostic builds, tools/ab_ablate.py; all four give the same bits).
            if constexpr (ORD == 1) {                        // every fragment read of the stage up front, fenced
                rd_x(1); rd_w(0);
                rd_x(0); rd_w(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                slot(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                slot(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (ORD == 2) {                 // DMA first, then reads, fenced
                slot(0); slot(1);
                rd_x(1); rd_w(0);
                rd_x(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                rd_w(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (ORD == 3) {                 // the shipped order with a fence around every MFMA group
                rd_x(1); rd_w(0);
                rd_x(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                rd_w(1);
                slot(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                slot(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                rd_x(1); rd_w(0);
                rd_x(0);
                mma(1, 0);
                rd_w(1);
                slot(0);
                mma(0, 1);
                slot(1);
                mma(0, 0);
            }
            buf = buf + 1 == RING ? 0 : buf + 1;
            ++s;
        };
        for (int cp = 0; cp < cpairs; ++cp) {
            [&]<int... T>(std::integer_sequence<int, T...>) { (stage(std::integral_constant<int, T>{}, cp), ...); }
            (std::make_integer_sequence<int, 9>{});
        }
    }
#ifdef RDO_DIAG
    if (tid == 0 && blockIdx.x < 256 && blockIdx.y == 0) {   // shader-clock cycles and 100 MHz wall ticks of the K loop (rdo_diag_h2k_stamps)
        g_h2k_stamps[blockIdx.x * 4 + 0] = st_c0; g_h2k_stamps[blockIdx.x * 4 + 1] = st_r0;
        g_h2k_stamps[blockIdx.x * 4 + 2] = clock64(); g_h2k_stamps[blockIdx.x * 4 + 3] = wall_clock64();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 16) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }

    // ---- epilogue straight from the accumulators: lane (kg, l16) of tile (i, j) holds channels n .. n + 3 (n = n0 + wn0 + 16 i + 4 kg)
    // of pixel m = patch row wrow * TP + j, column l16 -- a quad of the fp32 NHWC tensors (16-byte accesses; the four lanes of a pixel
    // cover one 64-byte slice, a tile row of 16 pixels is 16 such segments) and a quarter of an H2 record (8 bytes per plane; the 64 lanes
    // of a tile cover 16 consecutive records = 512 contiguous bytes per plane).  No staging tile, no barrier.
    if (a.partial) {                                         // K split: raw sums (already divided by s_x s_w) for the second pass
        float* const part = a.partial + (long)tile.z * a.M * a.Cout;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int m = pbase + (wrow * TP + j) * a.W + l16;
                *reinterpret_cast<f32x4*>(part + (long)m * a.Cout + n0 + wn0 + 16 * i + 4 * kg) = acc[i][j] * a.acc_scale;
            }
        return;
    }
    // Loads and stores share ONE counter on this chip (vmcnt, in order): a load issued behind a store is only known to have landed once
    // the store has been acknowledged, and hipcc has to wait vmcnt(0) for any load whose issue sits behind a run-time branch.  A tile's
    // "load inputs, compute, store" therefore ran as 24 serial memory round trips per wave (planes + LeakyReLU: 111.9 us against 96.4 with
    // no activation -- the activation only added BRANCHES around loads that never ran; conv + tail 120).  So: the inputs of a whole row
    // of tiles (same 16 channels, TP pixel rows) are fetched first, then the row is computed and stored with no load in between.
    const bool need_planes = a.outp != nullptr;
    const bool has_act = a.epilogue != RDO_EPI_NONE;
    // the (leaky) rectifiers and their backward masks as selects on wave-uniform flags: `activate`'s switch per ELEMENT compiled to a tree
    // of branches per element -- ~3 000 basic blocks of epilogue, 19 us on the planes + LeakyReLU launch
    const bool rect = a.epilogue == RDO_EPI_LRELU || a.epilogue == RDO_EPI_LRELU_BWD || a.epilogue == RDO_EPI_RELU || a.epilogue == RDO_EPI_RELU_BWD;
    const bool rect_relu = a.epilogue == RDO_EPI_RELU || a.epilogue == RDO_EPI_RELU_BWD;
    const bool rect_aux = a.epilogue == RDO_EPI_LRELU_BWD || a.epilogue == RDO_EPI_RELU_BWD;      // the mask is the sign of the aux operand
    const float* yimg = nullptr;                             // unit tail: the target image of this patch, so that the quad of (m, n) is yimg[m Cout + n]
    if (a.tail_tgt) yimg = a.tail_tgt + ((long)a.tail_idx[(long)(*a.tail_iter) * a.tail_B + pb] - pb) * a.tail_per_image;
    // The fetches are UNCONDITIONAL: four loads per tile (fp32 aux quad | fp32 residual or target quad | aux or tail-residual plane-0 quad |
    // tail-residual plane-1 quad), and an input the launch does not have reads the first 16 bytes of the activation planes instead (one
    // cache line for the whole wave).  A load behind a run-time branch costs hipcc a vmcnt(0) at the join -- which, with stores and loads
    // on one counter, is a wait for the previous row's stores even when the load never ran.
    const bool f_a = has_act && a.aux != nullptr;
    const bool f_ap = has_act && a.aux == nullptr && a.auxp != nullptr;
    const bool f_y = yimg != nullptr, f_rp = f_y && a.tail_resp != nullptr;
    const bool f_b = a.add_residual || f_y, f_c = f_ap || f_rp;
    const float* const dummyf = reinterpret_cast<const float*>(a.xp);
    const float* const pA = f_a ? a.aux : dummyf;
    const float* const pB = a.add_residual ? a.residual : (f_y ? yimg : dummyf);
    const u16* const pC = f_ap ? a.auxp : (f_rp ? a.tail_resp : a.xp);
    const u16* const pD = f_rp ? a.tail_resp + a.oplane : a.xp;
    float tail_loss = 0.f;
    int bad = 0;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int n = n0 + wn0 + 16 * i + 4 * kg;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
        f32x4 q0[TP], q1[TP];                                // q0: fp32 activation operand (mask / GDN input); q1: fp32 residual or tail target
        u32x2k rp0[TP], rp1[TP];                             // plane-0 quad of the aux tensor or of the tail residual; its plane-1 quad
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = pbase + (wrow * TP + j) * a.W + l16;
            const long o = (long)m * a.Cout + n;
            const long po = ((long)(n >> 4) * a.M + m) * 16 + (n & 15);
            q0[j] = *reinterpret_cast<const f32x4*>(pA + (f_a ? o : 0L));
            q1[j] = *reinterpret_cast<const f32x4*>(pB + (f_b ? o : 0L));
            rp0[j] = *reinterpret_cast<const u32x2k*>(pC + (f_c ? po : 0L));
            rp1[j] = *reinterpret_cast<const u32x2k*>(pD + (f_rp ? po : 0L));
        }
        // every fetched register passes through an empty asm HERE: hipcc waits for the row's loads once, in front of it, and no longer
        // connects the values with a (possibly) pending load -- otherwise each tile's first use of q0 / q1 behind the previous tile's
        // conditional stores is another vmcnt(0), i.e. a wait for those stores
#pragma unroll
        for (int j = 0; j < TP; ++j) asm volatile("" : "+v"(q0[j]), "+v"(q1[j]), "+v"(rp0[j]), "+v"(rp1[j]));
        asm volatile("" : "+v"(b4));
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int m = pbase + (wrow * TP + j) * a.W + l16;
            const long o = (long)m * a.Cout + n;
            f32x4 v = acc[i][j] * a.acc_scale + b4;
            if (a.pre) *reinterpret_cast<f32x4*>(a.pre + o) = v;
            if (f_ap)                                        // plane 0 of the aux tensor: same sign as the value, enough for the masks
                q0[j] = f32x4{h2_lo(rp0[j][0]), h2_hi(rp0[j][0]), h2_lo(rp0[j][1]), h2_hi(rp0[j][1])};
            if (rect) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = rect_aux ? q0[j][e] : v[e];
                    const float lo = rect_relu ? 0.f : 0.01f * v[e];
                    v[e] = t > 0.f ? v[e] : lo;
                }
            } else if (has_act) {                            // GDN / IGDN: aux * v^-1/2 / aux * v^1/2 (the arithmetic of `activate`)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = q0[j][e] * (a.epilogue == RDO_EPI_IGDN ? __fsqrt_rn(v[e]) : __frsqrt_rn(v[e]));
            }
            if (a.add_residual) v += q1[j];
            if (a.out) *reinterpret_cast<f32x4*>(a.out + o) = v;
            if (need_planes) {
                if (f_y) tail_loss += tail4_math(a, v, q1[j], rp0[j], rp1[j]);
                u16* dst = a.outp + ((long)(n >> 4) * a.M + m) * 16 + (n & 15);
                u32x2 hi, lo;
                unsigned h, l;
                rdo::h2_split_pk(v[0], v[1], a.out_scale, h, l, bad); hi[0] = h; lo[0] = l;
                rdo::h2_split_pk(v[2], v[3], a.out_scale, h, l, bad); hi[1] = h; lo[1] = l;
                *reinterpret_cast<u32x2*>(dst) = hi;
                *reinterpret_cast<u32x2*>(dst + a.oplane) = lo;
            }
        }
    }
    if (a.tail_tgt) {                                        // one atomic per workgroup
        float* red = reinterpret_cast<float*>(smem);         // (every wave is past the K loop's last LDS read: the barrier above)
        for (int o = 32; o > 0; o >>= 1) tail_loss += __shfl_down(tail_loss, o, 64);
        if (lane == 0) red[wave] = tail_loss;
        __syncthreads();
        if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && a.tail_pub) *a.tail_pub = *a.tail_iter;
        if (tid == 0 && a.tail_loss)
            atomicAdd(a.tail_loss + (long)(*a.tail_iter) * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)),
                      (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) * (a.tail_inv_npix * a.tail_coef));
    }
    rdo::h2_report(bad, a.ovf);
}

template <int WM, int WN, int TP, int TC, int ABL = 0, int ORD = 0>
int launch(const H2Args& a, const char* what, hipStream_t s) {
    constexpr int BN = WN * TC * 16;
    constexpr size_t lds = (size_t)2 * 42 * 1024 + 1024 + (size_t)3 * 2 * BN * 64;
    static_assert(lds <= 160 * 1024, "LDS of the K32 halo kernel");
    static rdo::PerDevice attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_h2k_kernel<WM, WN, TP, TC, ABL, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(%s) failed", what);
        attr.mark();
    }
    dim3 grid((unsigned)(a.M / 256), (unsigned)(a.Cout / BN), (unsigned)a.ksplit);
    hipLaunchKernelGGL((conv_fwd_h2k_kernel<WM, WN, TP, TC, ABL, ORD>), grid, dim3(512), lds, s, a);
    return rdo::check_launch(what);
}

}  // namespace

namespace rdo {

bool h2k_supported(const H2Args& a, int shape) {
    return a.Cout % (shape == 1 ? 192 : (shape == 3 ? 48 : 64)) == 0 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H % 16 == 0 && a.W % 16 == 0 &&
           a.Cin % 32 == 0 && a.ksplit >= 1 && a.ksplit <= a.Cin / 32 && (a.ksplit == 1) == (a.partial == nullptr) && (shape >= 2 || a.ksplit == 1) &&
           (long)a.Cin * 9 * a.Cout * 16 < (1L << 30) && a.xplane < (1L << 30);      // (one buffer descriptor spans both planes: < 4 GiB)
}

int h2k_launch(const H2Args& a, int shape, hipStream_t s) {
    if (shape == 1) {
#ifdef RDO_DIAG
        switch (a.ablate) {                                  // tuning key "x6p_ablate" (diagnostic builds)
            case 3: return launch<4, 2, 4, 6, 3>(a, "conv_fwd_h2k abl3", s);
            case 4: return launch<4, 2, 4, 6, 4>(a, "conv_fwd_h2k abl4", s);
            case 8: return launch<4, 2, 4, 6, 8>(a, "conv_fwd_h2k abl8", s);
            case 11: return launch<4, 2, 4, 6, 11>(a, "conv_fwd_h2k abl11", s);
            case 15: return launch<4, 2, 4, 6, 15>(a, "conv_fwd_h2k abl15", s);
            case 16: return launch<4, 2, 4, 6, 16>(a, "conv_fwd_h2k abl16", s);
            case 31: return launch<4, 2, 4, 6, 31>(a, "conv_fwd_h2k abl31", s);
            case 101: return launch<4, 2, 4, 6, 0, 1>(a, "conv_fwd_h2k ord1", s);
            case 102: return launch<4, 2, 4, 6, 0, 2>(a, "conv_fwd_h2k ord2", s);
            case 103: return launch<4, 2, 4, 6, 0, 3>(a, "conv_fwd_h2k ord3", s);
            default: break;
        }
#endif
        return launch<4, 2, 4, 6>(a, "conv_fwd_h2k 256x192", s);
    }
    if (shape == 3) return launch<8, 1, 2, 3>(a, "conv_fwd_h2k 256x48", s);
    return launch<8, 1, 2, 4>(a, "conv_fwd_h2k 256x64", s);
}

}  // namespace rdo

#ifdef RDO_DIAG
extern "C" int rdo_diag_h2k_stamps(unsigned long long* out, int n) {
    if (!out || n <= 0 || n > 1024) return RDO_EINVAL;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2k_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? RDO_OK : RDO_EHIP;
}
#endif
