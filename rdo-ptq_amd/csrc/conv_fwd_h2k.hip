// Halo forward / dgrad conv on H2 tensors, second form ("K32"): 3 x 3, stride 1, pad 1 convs whose H and W are multiples of 16 and whose
// Cin is a multiple of 32.  Same tile as conv_fwd_h2h_kernel (conv_fwd_h2.hip) -- the 256 output pixels of a workgroup are a 16 x 16
// patch of one image, the 18 x 18 halo of the patch is fetched once per channel group and serves all nine taps, eight waves, one
// workgroup per CU, staggered SIMD partners -- but
//   * a K stage is 32 input channels (two slices of the H2 layout) of one tap: half as many stages, barriers and loop skeletons
//     (measured on the 16-channel kernel: 229 of the 1 704 cycles of a stage were barrier + skeleton with nothing else in it);
//   * the products run on v_mfma_f32_16x16x32_f16: under the board's power limit the chip holds a higher clock on that shape than on
//     32x32x16 at the same cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7), and these kernels run AT that limit
//     (1.90 GHz in the K loop of the 16-channel kernel);
//   * A = weights (rows = output channels), B = activations (columns = pixels): a lane's four accumulator values are four consecutive
//     CHANNELS of one pixel, so the accumulators go to the epilogue's staging tile as 16-byte stores (the 32 x 32 form: 4-byte stores).
// LDS images, both [row][64 bytes] per plane with row = halo pixel q (activations) or output channel (weights) and the four 16-byte
// chunks of a row = the four k-groups of the MFMA (8 input channels each; k-groups 0, 1 = first slice, 2, 3 = second slice), stored at
// chunk position g ^ (2 * ((row >> 2) & 1)): every ds_read_b128 of a fragment -- 16 consecutive rows x 4 k-groups, any start row --
// touches all 64 banks once.  The swizzle sits on the per-lane SOURCE address of the LDS-DMA pieces (1 KiB = 16 rows x 64 bytes).
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
__device__ __attribute__((aligned(64))) unsigned g_zero_page_k[16];    // zero-initialised: source of masked DMA lanes

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

// WM x WN waves (8), wave tile TP pixel tiles x TC channel tiles of 16 x 16: WM * TP = 16 (256 patch pixels), BN = WN * TC * 16
template <int WM, int WN, int TP, int TC>
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
    constexpr int BJ = BPIECES / 8;                          // weight pieces per wave and stage: 3 / 1
    constexpr int RING = 3;
    constexpr int SROW = BN + 4;
    constexpr int PROWS = 128;                               // tile rows (pixels) finished per epilogue pass
    static_assert(BPIECES % 8 == 0, "uniform weight-piece count per wave");
    static_assert(TP % 2 == 0, "two epilogue passes of TP / 2 pixel tiles per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave / WN, wcol = wave - wrow * WN;
    const int wm0 = wrow * (TP * 16), wn0 = wcol * (TC * 16);
    const int l16 = lane & 15, kg = lane >> 4;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int n0 = tile.n * BN;
    const int pw_n = a.W / PT, ph_n = a.H / PT;
    const int pb = tile.m / (ph_n * pw_n);
    const int prem = tile.m - pb * (ph_n * pw_n);
    const int h0 = (prem / pw_n) * PT, w0 = (prem % pw_n) * PT;
    const int pbase = (pb * a.H + h0) * a.W + w0;            // first pixel of the patch
    const long Min = (long)a.B * a.H * a.W;
    const u16* const zero = reinterpret_cast<const u16*>(g_zero_page_k);

    // ---- halo loader: pieces wave, wave + 8, ... of the [plane][halo pixel][chunk position] image.  The lane's source offset of a
    // piece is recomputed when the piece is issued (once per stage at most: ~25 vector instructions) -- six such offsets kept in
    // registers for the whole K loop were what spilled the 256-register budget of the 256 x 192 tile.
    auto halo_off = [&](int piece) -> int {                  // element offset into a.xp for slice pair 0, or -1: zero page
        const int e = piece * 64 + lane;
        const int pl = e >= HPIX * 4 ? 1 : 0;
        const int r = e - pl * (HPIX * 4);
        const int q = r >> 2;
        const int g = (r & 3) ^ swz(q);                      // k-group that belongs at chunk position r & 3 of row q
        const int hr = (q * 3641) >> 16;                     // q / 18 for q < 1 296
        const int hc = q - hr * HWD;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
        const bool ok = e < ACHUNKS && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        return ok ? (int)(pl * a.xplane) + (int)(((g >> 1) * Min + ((pb * a.H + hi) * a.W + wi)) * 16) + (g & 1) * 8 : -1;
    };
    // ---- weight loader: pieces wave, wave + 8, ... of the [plane][channel][chunk position] image of a stage.  BN * 4 chunks per plane
    // are BN / 16 whole pieces, so piece k is plane k / (BN / 16), channels 16 (k % (BN / 16)) ...: a wave-uniform part per piece and ONE
    // per-lane part (channel within the piece, k-group through the swizzle, which depends on the lane only)
    int dma_lane;
    {
        const int row = lane >> 2;                           // channel within the piece; (16 k' + row) >> 2 has the parity of row >> 2
        const int g = (lane & 3) ^ swz(row);
        dma_lane = (g >> 1) * 9 * a.Cout * 16 + row * 16 + (g & 1) * 8;
    }
    auto dma_uni = [&](int j) -> int {                       // wave-uniform part of piece wave + 8 j
        const int k = wave + 8 * j;
        const int pl = k / (BN / 16), kk = k - pl * (BN / 16);
        return (int)(pl * a.wplane) + (n0 + kk * 16) * 16;
    };

    f32x4k acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4k{0.f, 0.f, 0.f, 0.f};

    const int cpairs = a.Cin / 32;
    const int nstages = 9 * cpairs;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma_a = [&](int j, int cp_next) {                   // piece j of the wave, halo of slice pair cp_next into buffer cp_next & 1
        const int cn = cp_next < cpairs ? cp_next : cpairs - 1;
        const int piece = wave + 8 * j;
        const int off = halo_off(piece);
        const u16* src = off >= 0 ? a.xp + off + (long)cn * Min * 32 : zero;
        char* dst = piece < APIECES ? smem + (cp_next & 1) * ABUF + piece * 1024 : smem + ADUMMY;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
    };
    const u16* wsrc = a.wp;
    auto dma_b = [&](int j, int nb) {
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + dma_uni(j) + dma_lane), (lds_void*)(smem + BBASE + nb * BSTAGE + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    // stage s = slice pair s / 9, tap s % 9: its weights start at slice 2 (s / 9), tap s % 9 of the fragment-ordered planes
    auto set_wsrc = [&](int stage) {
        const int st = stage < nstages ? stage : nstages - 1;
        const int cp = st / 9, tap = st - cp * 9;
        wsrc = a.wp + ((long)(2 * cp) * 9 + tap) * a.Cout * 16;
    };

    // fragment addresses.  Weights: row = wn0 + 16 i + l16 with wn0 and 16 i multiples of 8, so the swizzle depends on the lane only:
    // ONE base register, (plane, tile) are immediate offsets.  Halo: q = (patch row + kh) * 18 + l16 + kw, swizzled per (tile, tap):
    // recomputed every stage from `qlane` (kept opaque to the optimiser there: hoisted out of the slice-pair loop the 9 x TP addresses
    // would cost 36 registers and spill the accumulators).
    const int fw_base = BBASE + (wn0 + l16) * 64 + ((kg ^ swz(l16)) << 4);
    int qlane = (wrow * TP) * HWD + l16;
    const bool late = wave >= 4;

    // prologue: halo of slice pair 0, weights of stages 0 and 1
#pragma unroll
    for (int j = 0; j < AJ; ++j) dma_a(j, 0);
    set_wsrc(0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 0);
    set_wsrc(1);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 1);

    auto k_loop = [&](auto late_c) {
        constexpr bool LATE = decltype(late_c)::value;
        f16x8 fx[NP][TP], fw[NP][TC];
        if constexpr (LATE) {                                // the "previous stage" of stage 0: zero fragments
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int j = 0; j < TP; ++j) fx[p][j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < TC; ++i) fw[p][i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        auto mma = [&](int px, int pw) {
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
            if constexpr (TAP >= 1 && TAP <= AJ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ + 1) : "memory");
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
                fx_off[j] = q * 64 + ((kg ^ swz(q)) << 4);
            }
            auto rd_x = [&](int p) {
#pragma unroll
                for (int j = 0; j < TP; ++j) fx[p][j] = *reinterpret_cast<const f16x8*>(stx + p * APL + fx_off[j]);
            };
            auto rd_w = [&](int p) {
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
            if constexpr (!LATE) {
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
                rd_x(1);                                     // fx[1] is free: (X1, W0) of the previous stage is done
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);                                   // previous stage
                __builtin_amdgcn_sched_barrier(0);
                rd_w(1);
                slot(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);                                   // previous stage
                __builtin_amdgcn_sched_barrier(0);
                rd_w(0); rd_x(0);
                slot(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = buf + 1 == RING ? 0 : buf + 1;
            ++s;
        };
        for (int cp = 0; cp < cpairs; ++cp) {
            [&]<int... T>(std::integer_sequence<int, T...>) { (stage(std::integral_constant<int, T>{}, cp), ...); }
            (std::make_integer_sequence<int, 9>{});
        }
        if constexpr (LATE) {                                // the two products of the last stage these waves still owe
            mma(0, 1);
            mma(0, 0);
        }
    };
    if (late && a.stagger) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();

    // ---- epilogue through LDS, two passes of 128 patch pixels (pass p: pixel tiles p * TP / 2 ... of every wave).  Per pass: the
    // accumulators to the staging tile [pixel][channel] (one 16-byte store per MFMA tile and lane), then (1) consecutive lanes along
    // the CHANNELS finish quads -- bias, activation, residual, coalesced 16-byte loads / stores of the fp32 tensors -- and put the
    // finished values back; (2) consecutive lanes along the PIXELS split the 16 channels of one slice and write the two H2 records.
    float* const stg = reinterpret_cast<float*>(smem);
    const bool need_planes = a.outp != nullptr;
    float tail_loss = 0.f;
    int bad = 0;
    constexpr int HP = TP / 2;                               // pixel tiles of a wave per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                const f32x4k v = acc[i][pass * HP + j] * a.acc_scale;
                *reinterpret_cast<f32x4k*>(stg + (wrow * (HP * 16) + j * 16 + l16) * SROW + wn0 + i * 16 + 4 * kg) = v;
            }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < PROWS * (BN / 4) / 512; ++k) {
            const int q = tid + 512 * k;                     // PROWS rows x BN / 4 quads of channels
            const int row = q / (BN / 4), c4 = q - row * (BN / 4);
            const int tl = (row / (HP * 16)) * (TP * 16) + pass * (HP * 16) + (row % (HP * 16));
            const int m = pbase + (tl >> 4) * a.W + (tl & 15);
            const int n = n0 + c4 * 4;
            if (n >= a.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * SROW + c4 * 4);
            const long o = (long)m * a.Cout + n;
            if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.pre) *reinterpret_cast<f32x4*>(a.pre + o) = v;
            if (a.epilogue != RDO_EPI_NONE) {
                const f32x4 x4 = aux_quad(a, m, n, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = activate(a, v[e], x4[e]);
            }
            if (a.add_residual) v += *reinterpret_cast<const f32x4*>(a.residual + o);
            if (a.out) *reinterpret_cast<f32x4*>(a.out + o) = v;
            if (need_planes) *reinterpret_cast<f32x4*>(stg + row * SROW + c4 * 4) = v;
        }
        if (need_planes) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PROWS * (BN / 16) / 512; ++k) {
                const int q = tid + 512 * k;                 // PROWS rows x BN / 16 slices of 16 channels
                const int row = q % PROWS, sl = q / PROWS;
                const int tl = (row / (HP * 16)) * (TP * 16) + pass * (HP * 16) + (row % (HP * 16));
                const int m = pbase + (tl >> 4) * a.W + (tl & 15);
                const int n = n0 + sl * 16;
                if (n >= a.Cout) continue;
                float v[16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(stg + row * SROW + sl * 16 + 4 * c);
                    v[4 * c] = t4[0]; v[4 * c + 1] = t4[1]; v[4 * c + 2] = t4[2]; v[4 * c + 3] = t4[3];
                }
                if (a.tail_tgt) tail_loss += tail16(a, m, n, v);
                store_slice(a, m, n, v, bad);
            }
        }
    }
    if (a.tail_tgt) {                                        // one atomic per workgroup
        __shared__ float red[8];
        for (int o = 32; o > 0; o >>= 1) tail_loss += __shfl_down(tail_loss, o, 64);
        if (lane == 0) red[wave] = tail_loss;
        __syncthreads();
        if (tid == 0 && a.tail_loss)
            atomicAdd(a.tail_loss + (long)(*a.tail_iter) * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)),
                      (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) * (a.tail_inv_npix * a.tail_coef));
    }
    rdo::h2_report(bad, a.ovf);
}

template <int WM, int WN, int TP, int TC>
int launch(const H2Args& a, const char* what) {
    constexpr int BN = WN * TC * 16;
    constexpr size_t k_lds = (size_t)2 * 42 * 1024 + 1024 + (size_t)3 * 2 * BN * 64;
    constexpr size_t e_lds = (size_t)128 * (BN + 4) * 4;
    constexpr size_t lds = k_lds > e_lds ? k_lds : e_lds;
    static_assert(lds <= 160 * 1024, "LDS of the K32 halo kernel");
    static rdo::PerDevice attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_h2k_kernel<WM, WN, TP, TC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(%s) failed", what);
        attr.mark();
    }
    return 0;
}

}  // namespace

namespace rdo {

bool h2k_supported(const H2Args& a, int shape) {
    return a.Cout % (shape == 1 ? 192 : 64) == 0 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.H % 16 == 0 && a.W % 16 == 0 && a.Cin % 32 == 0 && a.Cout % 64 == 0 && a.ksplit == 1 &&
           a.partial == nullptr && (long)a.Cin * 9 * a.Cout * 16 < (1L << 31);
}

int h2k_launch(const H2Args& a, int shape, hipStream_t s) {
    if (shape == 1) {
        constexpr size_t lds = (size_t)2 * 42 * 1024 + 1024 + (size_t)3 * 2 * 192 * 64;
        if (int rc = launch<4, 2, 4, 6>(a, "conv_fwd_h2k 256x192")) return rc;
        dim3 grid((unsigned)(a.M / 256), (unsigned)rdo::ceil_div(a.Cout, 192), 1);
        hipLaunchKernelGGL((conv_fwd_h2k_kernel<4, 2, 4, 6>), grid, dim3(512), lds, s, a);
        return rdo::check_launch("conv_fwd_h2k 256x192");
    }
    constexpr size_t k64 = (size_t)2 * 42 * 1024 + 1024 + (size_t)3 * 2 * 64 * 64;
    if (int rc = launch<8, 1, 2, 4>(a, "conv_fwd_h2k 256x64")) return rc;
    dim3 grid((unsigned)(a.M / 256), (unsigned)(a.Cout / 64), 1);
    hipLaunchKernelGGL((conv_fwd_h2k_kernel<8, 1, 2, 4>), grid, dim3(512), k64, s, a);
    return rdo::check_launch("conv_fwd_h2k 256x64");
}

}  // namespace rdo
