// Forward / dgrad conv of the LARGE problems on "H2" activation tensors: the fp32 activation is already stored as the exact two-way
// fp16 split of its power-of-two-scaled self (x s = h1 + h2, rdo_common.h), written by the producing kernel, so BOTH operands of the
// split-precision GEMM arrive in LDS by LDS-DMA and the K loop contains no conversion arithmetic at all.  Three fp16 MFMA products per
// fp32 product (h1 g1 + h1 g2 + h2 g1, small terms first) with fp32 accumulation: half the matrix work and two thirds of the operand
// bytes of the three-way bf16 split (six products) these kernels used before, at the same accuracy (tools/f16_probe.hip).
//
// H2 layout ("slice-major planes", rdo_ptq_hip.h): element (pixel m, channel c) of a tensor with M pixels and C channels lives at
//     plane * M * C + ((c >> 4) * M + m) * 16 + (c & 15)            i.e. planes [2][C/16][pixel][16]
// One K stage of the kernel reads the 16 channels of ONE slice for a run of pixels, and consecutive pixels of a slice are consecutive
// 32-byte records: every 1-KiB LDS-DMA piece is a contiguous KiB of memory.
//
// Kernel: 256 x 192 tile on eight waves (4 x 2, wave tile 64 x 96 = 2 x 3 tiles of v_mfma_f32_32x32x16_f16), ONE workgroup per CU,
// K stage = 16 channels of one tap (channel slice outer, taps inner), THREE LDS stage buffers: the DMA of stage t+2 is issued during
// stage t and has two full stages to land (counted vmcnt, raw s_barrier).
//
// Epilogue through LDS: the accumulators of 128 tile rows are staged as fp32 (already multiplied by 1 / (s_x s_w)), then every thread
// finishes the 16 channels of one slice of one pixel (consecutive lanes on consecutive pixels = the H2 record order): bias / aux /
// residual, fp32 out / pre, and the two fp16 planes of the result (scaled by the output tensor's own s), two 16-byte stores per plane.
#include "conv_h2_common.h"

namespace {

#ifdef RDO_DIAG
__device__ unsigned long long g_h2_stamps[256 * 4];
#endif
__device__ __attribute__((aligned(64))) unsigned g_zero_page[16];      // zero-initialised: source of masked DMA lanes

__global__ __launch_bounds__(512, 2) void conv_fwd_h2_kernel(H2Args a) {
    constexpr int BM = 256, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;
    constexpr int APLANE = BM * 32, BPLANE = BN * 32;
    constexpr int BOFF = NP * APLANE;                      // weight image of a stage: [plane][row][chunk] behind the activation planes
    constexpr int STAGE = NP * (APLANE + BPLANE);          // 28 KiB
    constexpr int RING = 3;
    constexpr int BPIECES = NP * BN * 2 * 16 / 1024;       // 1-KiB DMA pieces of the weight image: 12
    constexpr int SROW = 196;                              // staging row stride in floats: 16-byte aligned, conflict-free b128 reads down a column

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [RING][STAGE]; reused by the epilogue as fp32 [128][SROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;
    const long Min = (long)a.B * a.H * a.W;

    // ---- A loader: this lane fetches 16-byte chunk (lane & 1) of tile row 32 * wave + (lane >> 1), for each of the two planes
    const int arow = wave * 32 + (lane >> 1);
    const int achunk = (lane & 1) ^ ((arow >> 3) & 1);      // source chunk that belongs at LDS position (arow, lane & 1)
    int hi0, wi0, apix0;
    {
        const int m = m0 + arow;
        const bool okr = m < a.M;
        const int mm = okr ? m : 0;
        const int b = mm / (a.Ho * a.Wo);
        const int rem = mm - b * (a.Ho * a.Wo);
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0 = okr ? ho * a.stride - a.pad : -(1 << 28);
        wi0 = wo * a.stride - a.pad;
        apix0 = b * a.H * a.W;
    }
    const u16* const zero = reinterpret_cast<const u16*>(g_zero_page);

    // ---- B loader: pieces wave and wave + 8 of the 12 KiB [plane][row][chunk] image (waves 4..7 repeat their first piece)
    int dma_src[2], dma_k[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = wave + 8 * j < BPIECES ? wave + 8 * j : wave;
        dma_k[j] = k;
        const int e = k * 64 + lane;
        const int pl = e / (BN * 2);
        const int r = e - pl * (BN * 2);
        const int row = r >> 1, ch = (r & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row < a.Cout ? n0 + row : 0;
        dma_src[j] = (int)(pl * a.wplane) + n * 16 + 8 * ch;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cin / KS;
    const int taps = a.KH * a.KW;
    const int steps_total = taps * csteps;
    const int sbeg = (int)((long)steps_total * tile.z / a.ksplit);
    const int nsteps = (int)((long)steps_total * (tile.z + 1) / a.ksplit) - sbeg;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    // DMA cursor (wave-uniform): (channel slice, kh, kw) of the next stage to fetch; stages run channel slice outer, taps inner.  The
    // workgroups of an XCD walk the weight stream together (a per-tile rotation of the K order measured slower): one L2 fill serves all.
    int cabs = sbeg;                                         // absolute stage (channel slice * taps + tap) under the cursor
    int ccs = cabs / taps;
    int ctap = cabs - ccs * taps;
    int ckh = ctap / a.KW, ckw = ctap - ckh * a.KW;
    int cstage = 0;
    // four DMA instructions per wave and stage (two activation planes, two weight pieces): the vmcnt bookkeeping below counts on it.
    // dma_prepare computes the stage's source addresses, dma_issue(j) issues instruction j and the cursor moves on.
    const u16* asrc = zero;
    long apstep = 0;
    const u16* wsrc = a.wp;
    auto dma_prepare = [&]() {
        const int hi = hi0 + ckh, wi = wi0 + ckw;
        const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const long off = ((long)ccs * Min + (apix0 + hi * a.W + wi)) * 16 + achunk * 8;
        asrc = ok ? a.xp + off : zero;
        apstep = ok ? a.xplane : 0;
        wsrc = a.wp + (long)cabs * a.Cout * 16;
        if (cstage + 1 < nsteps) {                           // advance, clamped at the last stage (tail stages re-fetch it: harmless)
            ++cstage;
            ++cabs;
            ++ckw;
            if (ckw == a.KW) { ckw = 0; ++ckh; }
            if (ckh == a.KH) { ckh = 0; ++ccs; }
        }
    };
    auto dma_issue = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        char* const base = smem + buf * STAGE;
        if constexpr (j < NP) {
            if (!(a.ablate & 1))
                __builtin_amdgcn_global_load_lds((glb_void*)(asrc + j * apstep), (lds_void*)(base + wave * 1024 + j * APLANE), 16, 0, 0);
        } else {
            if (!(a.ablate & 2))
                __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + dma_src[j - NP]), (lds_void*)(base + BOFF + dma_k[j - NP] * 1024), 16, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    auto dma_all = [&](int buf) {
        dma_prepare();
        dma_issue(I0{}, buf); dma_issue(I1{}, buf); dma_issue(I2{}, buf); dma_issue(I3{}, buf);
    };

    int fa_off[NP][TM], fb_off[NP][TN];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_off[p][i] = p * APLANE + chunk_off16(wm0 + i * 32 + li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = BOFF + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    }
    // the second-dispatched half of the workgroup (waves 4-7, the SIMD partners of waves 0-3) issues its DMAs one MFMA slot later than
    // the first half: while one wave of a SIMD is held up issuing DMAs its partner multiplies
    const bool late = wave >= 4;

    dma_all(0);
    dma_all(1);
    int buf = 0;
    for (int t = 0; t < nsteps; ++t) {
        // stage t has landed for this wave once at most the four youngest DMAs (stage t+1) are outstanding; the barrier extends that to
        // every wave and also tells that nobody still reads buffer (t+2) % 3 = (t-1) % 3
        if (a.ablate & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nb = buf >= 1 ? buf - 1 : RING - 1;        // (t + 2) % 3
        const char* st = smem + buf * STAGE;
        f16x8 fa[NP][TM], fb[NP][TN];
        auto rd_a = [&](int p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const f16x8*>(st + fa_off[p][i]);
        };
        auto rd_b = [&](int p) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const f16x8*>(st + fb_off[p][j]);
        };
        auto mma = [&](int pa, int pb) {
            if (a.ablate & 4) return;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
        };
        dma_prepare();
        // fragments in the order the products need them: (A1,B0) (A0,B1) (A0,B0), small terms first
        // (every MFMA group is fenced on both sides: left to itself the scheduler hoists the DMA issue and its address arithmetic in
        // front of the group's MFMAs, which cost 20 % on the 4 x 128^2 conv)
        rd_a(1); rd_b(0);
        rd_a(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd_b(1);
        if (!late) { dma_issue(I0{}, nb); dma_issue(I1{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!late) { dma_issue(I2{}, nb); dma_issue(I3{}, nb); }
        else { dma_issue(I0{}, nb); dma_issue(I1{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (late) { dma_issue(I2{}, nb); dma_issue(I3{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();

    // ---- epilogue through LDS: two passes of 128 tile rows (pass i: MFMA row tile i of every wave).  Per pass: (1) consecutive lanes
    // along the CHANNELS finish quads -- bias, activation, residual, coalesced 16-byte loads / stores of the fp32 tensors -- and put
    // the finished values back into the staging tile; (2) consecutive lanes along the PIXELS split the 16 channels of one slice and
    // write the two H2 records (32 contiguous bytes per plane and pixel, consecutive pixels adjacent).
    float* const stg = reinterpret_cast<float*>(smem);       // [128][SROW] fp32 = 98 KiB
    const long part_base = (long)tile.z * a.M * a.Cout;
    const bool need_planes = a.outp != nullptr && a.partial == nullptr;
    int bad = 0;
#pragma unroll
    for (int pass = 0; pass < TM; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stg[((wave >> 1) * 32 + rr) * SROW + wn0 + j * 32 + li] = acc[pass][j][r] * a.acc_scale;
            }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 12; ++k) {
            const int q = tid + 512 * k;                     // 128 rows x 48 quads of channels
            const int row = q / 48, c4 = q - row * 48;
            const int m = m0 + (row >> 5) * 64 + pass * 32 + (row & 31);
            const int n = n0 + c4 * 4;
            if (m >= a.M || n >= a.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * SROW + c4 * 4);
            const long o = (long)m * a.Cout + n;
            if (a.partial) {
                *reinterpret_cast<f32x4*>(a.partial + part_base + o) = v;
                continue;
            }
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
            for (int k = 0; k < 3; ++k) {
                const int q = tid + 512 * k;                 // 128 rows x 12 slices of 16 channels
                const int row = q & 127, sl = q >> 7;
                const int m = m0 + (row >> 5) * 64 + pass * 32 + (row & 31);
                const int n = n0 + sl * 16;
                if (m >= a.M || n >= a.Cout) continue;
                float v[16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(stg + row * SROW + sl * 16 + 4 * c);
                    v[4 * c] = t4[0]; v[4 * c + 1] = t4[1]; v[4 * c + 2] = t4[2]; v[4 * c + 3] = t4[3];
                }
                store_slice(a, m, n, v, bad);
            }
        }
    }
    rdo::h2_report(bad, a.ovf);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Halo variant for 3 x 3, stride 1, pad 1 convs whose H and W are multiples of 16: the 256 output pixels of a workgroup are a
// 16 x 16 PATCH of one image, and the 18 x 18 halo of that patch (16 channels of one slice, two planes = 20.3 KiB) is fetched ONCE
// per channel slice and serves all nine taps -- a tap is an offset into the halo tile when the A fragments are read.  The kernel above
// re-fetches a shifted 256-pixel tile for every tap: 16 KiB x 9 per slice against 20.3 KiB here (weights unchanged).  That matters
// because these kernels run at the board's power limit (tools/clock_probe.py): the energy that can be saved is data movement.
// LDS: two halo buffers (slice cs and cs + 1) + the three-deep weight ring = 84 KiB (the epilogue staging needs 98 KiB).  DMA per wave
// and stage: two weight pieces for stage s + 2 and, in taps 0-2, one of the wave's three pieces of the next slice's halo; the counted
// vmcnt at the top of a stage is therefore 3 in taps 1-3 and 2 otherwise (the halo piece is issued before the weight pieces of its
// stage).
// Wave layout WM x WN (8 waves), wave tile (TM x TN) MFMA tiles of 32 x 32: the 256 patch pixels are WM * TM * 32, the N tile is
// WN * TN * 32 channels.  <4, 2, 2, 3>: 256 x 192, the shape for problems with >= ~190 such tiles (the 128^2 convs).  <8, 1, 1, 2>:
// 256 x 64 -- three N tiles per 192 output channels -- for the 64^2 convs (64 patches: 192 workgroups) and the 32^2 convs with 768
// output channels, which would otherwise run the per-tap kernel split over K with a second pass over 4-12 partial tensors.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(512, 2) void conv_fwd_h2h_kernel(H2Args a) {
    static_assert(WM * WN == 8 && WM * TM == 8, "eight waves, 256 patch pixels");
    constexpr int BN = WN * TN * 32;
    constexpr int PT = 16, HWD = PT + 2, HPIX = HWD * HWD;   // patch edge, halo edge, halo pixels (324)
    constexpr int APL = HPIX * 32;                           // one plane of the halo tile: 10368 B
    constexpr int APIECES = NP * HPIX * 2;                   // 16-byte pieces of the two planes: 1296 (21 wave instructions)
    constexpr int AJ = 3;                                    // pieces per wave (8 * 3 = 24 >= 21)
    constexpr int ABUF = 8 * AJ * 1024;                      // 24 KiB: every wave piece lands inside the buffer, masked ones as zeros
    constexpr int BPLANE = BN * 32, BSTAGE = NP * BPLANE;    // 12 KiB
    constexpr int BPIECES = BSTAGE / 1024;                   // 12 (BN = 192) or 4 (BN = 64)
    constexpr int BJ = (BPIECES + 7) / 8;                    // weight pieces per wave and stage
    constexpr int RING = 3;
    constexpr int BBASE = 2 * ABUF;
    constexpr int SROW = BN + 4;
    constexpr int PROWS = WM * 32;                           // tile rows finished per epilogue pass

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][ABUF] halo tiles, [RING][BSTAGE] weights; epilogue: fp32 [128][SROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave / WN, wcol = wave - wrow * WN;
    const int wm0 = wrow * (TM * 32), wn0 = wcol * (TN * 32);
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int n0 = tile.n * BN;
    const int pw_n = a.W / PT, ph_n = a.H / PT;
    const int pb = tile.m / (ph_n * pw_n);
    const int prem = tile.m - pb * (ph_n * pw_n);
    const int h0 = (prem / pw_n) * PT, w0 = (prem % pw_n) * PT;
    const int pbase = (pb * a.H + h0) * a.W + w0;            // first pixel of the patch
    const long Min = (long)a.B * a.H * a.W;
    const u16* const zero = reinterpret_cast<const u16*>(g_zero_page);

    // ---- A loader: pieces wave, wave + 8, wave + 16 of the [plane][halo pixel][chunk] image
    int a_off[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int e = (wave + 8 * j) * 64 + lane;
        const int pl = e / (HPIX * 2);
        const int r = e - pl * (HPIX * 2);
        const int q = r >> 1;
        const int chunk = (r & 1) ^ ((q >> 3) & 1);          // source chunk that belongs at LDS position (q, r & 1)
        const int hr = q / HWD, hc = q - hr * HWD;
        const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
        const bool ok = e < APIECES && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        a_off[j] = ok ? (int)(pl * a.xplane) + ((pb * a.H + hi) * a.W + wi) * 16 + chunk * 8 : -1;
    }
    // ---- B loader: as in conv_fwd_h2_kernel
    int dma_src[BJ], dma_k[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int k = (wave + 8 * j) % BPIECES;              // (waves beyond the image repeat a piece: uniform DMA counts per wave)
        dma_k[j] = k;
        const int e = k * 64 + lane;
        const int pl = e / (BN * 2);
        const int r = e - pl * (BN * 2);
        const int row = r >> 1, ch = (r & 1) ^ ((row >> 3) & 1);
        const int n = n0 + row < a.Cout ? n0 + row : 0;
        dma_src[j] = (int)(pl * a.wplane) + n * 16 + 8 * ch;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int csteps = a.Cin / 16;
    const int nstages = 9 * csteps;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma_a = [&](int j, int cs_next) {                   // piece j of the wave, halo of slice cs_next into buffer cs_next & 1
        if (a.ablate & 1) return;
        const int cn = cs_next < csteps ? cs_next : csteps - 1;
        const u16* src = a_off[j] >= 0 ? a.xp + a_off[j] + (long)cn * Min * 16 : zero;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(smem + (cs_next & 1) * ABUF + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    const u16* wsrc = a.wp;
    auto dma_b = [&](int j, int nb) {
        if (a.ablate & 2) return;
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + dma_src[j]), (lds_void*)(smem + BBASE + nb * BSTAGE + dma_k[j] * 1024), 16, 0, 0);
    };
    auto set_wsrc = [&](int stage) { wsrc = a.wp + (long)(stage < nstages ? stage : nstages - 1) * a.Cout * 16; };

    int q00[TM], fb_off[NP][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tl = wm0 + i * 32 + li;
        q00[i] = (tl >> 4) * HWD + (tl & 15);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = BBASE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    const bool late = wave >= 4;

    // prologue: halo of slice 0, weights of stages 0 and 1
#pragma unroll
    for (int j = 0; j < AJ; ++j) dma_a(j, 0);
    set_wsrc(0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 0);
    set_wsrc(1);
#pragma unroll
    for (int j = 0; j < BJ; ++j) dma_b(j, 1);

    // products of a stage in the order (A1,B0) (A0,B1) (A0,B0), small terms first -- for every wave.  STAGGER: the two waves of a SIMD
    // (w and w + 4) leave the barrier together; if both then read their fragments and both then multiply, the matrix pipe idles
    // through every read / DMA / barrier phase (ablation: 58 us of skeleton + 63 us of MFMAs = the 121 us of the kernel, no overlap at
    // all).  Waves 4-7 therefore run two thirds of a stage behind: after the barrier they first issue products two and three of the
    // PREVIOUS stage from the fragments still in their registers -- while waves 0-3 read -- and read this stage's fragments while
    // waves 0-3 multiply.  Every read of a stage still happens inside that stage's barrier interval; the DMA bookkeeping is unchanged.
    // The two roles are two separate instantiations of the K loop (no per-stage branch, no register merge at a join).
    auto k_loop = [&](auto late_c) {
        constexpr bool LATE = decltype(late_c)::value;
        f16x8 fa[NP][TM], fb[NP][TN];
        if constexpr (LATE) {                                // the "previous stage" of stage 0: zero fragments
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[p][i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[p][j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        auto mma = [&](int pa, int pb_) {
            if (a.ablate & 4) return;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[pa][i], fb[pb_][j], acc[i][j], 0, 0, 0);
        };
        int buf = 0, s = 0;
        auto stage = [&](auto tapc, int cs) {
            constexpr int TAP = decltype(tapc)::value;
            constexpr int KH = TAP / 3, KW = TAP % 3;
            // stage s (weights) and, in tap 0, this slice's halo have landed once at most the DMAs issued after them are outstanding
            if (a.ablate & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (TAP >= 1 && TAP <= AJ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BJ) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nb = buf >= 1 ? buf - 1 : RING - 1;    // (s + 2) % 3
            const char* stb = smem + buf * BSTAGE;
            const char* sta = smem + (cs & 1) * ABUF;
            int fa_off[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int q = q00[i] + KH * HWD + KW;
                fa_off[i] = q * 32 + (((lh ^ (q >> 3)) & 1) << 4);
            }
            auto rd_a = [&](int p) {
                if (a.ablate & 8) return;                    // diagnostic: no fragment reads (stale registers multiplied)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const f16x8*>(sta + p * APL + fa_off[i]);
            };
            auto rd_b = [&](int p) {
                if (a.ablate & 8) return;
                if ((a.ablate & 32) && p == 1) return;       // diagnostic: 30 % fewer fragment bytes (what a 128 x 96 wave tile would read)
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const f16x8*>(stb + fb_off[p][j]);
            };
            set_wsrc(s + 2);
            auto slot = [&](int k) {                         // DMA slot k of this wave's two: halo piece first (taps 0-2), then weight piece k
                if (k == 0) {
                    if constexpr (TAP < AJ) dma_a(TAP, cs + 1);
                }
                if (k < BJ) dma_b(k, nb);
            };
            if constexpr (!LATE) {
                rd_a(1); rd_b(0);
                rd_a(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
                rd_b(1);
                slot(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                slot(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                rd_a(1);                                     // fa[1] is free: (A1,B0) of the previous stage is done
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);                                   // previous stage
                __builtin_amdgcn_sched_barrier(0);
                rd_b(1);
                slot(0);
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 0);                                   // previous stage
                __builtin_amdgcn_sched_barrier(0);
                rd_b(0); rd_a(0);
                slot(1);
                __builtin_amdgcn_sched_barrier(0);
                mma(1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = buf + 1 == RING ? 0 : buf + 1;
            ++s;
        };
        for (int cs = 0; cs < csteps; ++cs) {
            [&]<int... T>(std::integer_sequence<int, T...>) { (stage(std::integral_constant<int, T>{}, cs), ...); }
            (std::make_integer_sequence<int, 9>{});
        }
        if constexpr (LATE) {                                // the two products of the last stage these waves still owe
            mma(0, 1);
            mma(0, 0);
        }
    };
#ifdef RDO_DIAG
    const unsigned long long st_c0 = clock64(), st_r0 = wall_clock64();
#endif
    if (late && a.stagger) k_loop(std::true_type{});
    else k_loop(std::false_type{});
#ifdef RDO_DIAG
    if (tid == 0 && blockIdx.x < 256 && blockIdx.y == 0) {   // shader-clock cycles and 100 MHz wall ticks of the K loop (rdo_diag_h2_stamps)
        g_h2_stamps[blockIdx.x * 4 + 0] = st_c0; g_h2_stamps[blockIdx.x * 4 + 1] = st_r0;
        g_h2_stamps[blockIdx.x * 4 + 2] = clock64(); g_h2_stamps[blockIdx.x * 4 + 3] = wall_clock64();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: as in conv_fwd_h2_kernel, tile row tl = pixel (tl / 16, tl % 16) of the patch
    if (a.ablate & 16) return;                               // diagnostic: no epilogue at all
    float* const stg = reinterpret_cast<float*>(smem);
    const bool need_planes = a.outp != nullptr;
    float tail_loss = 0.f;
    int bad = 0;
#pragma unroll
    for (int pass = 0; pass < TM; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stg[(wrow * 32 + rr) * SROW + wn0 + j * 32 + li] = acc[pass][j][r] * a.acc_scale;
            }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < PROWS * (BN / 4) / 512; ++k) {
            const int q = tid + 512 * k;                     // PROWS rows x BN / 4 quads of channels
            const int row = q / (BN / 4), c4 = q - row * (BN / 4);
            const int tl = (row >> 5) * (TM * 32) + pass * 32 + (row & 31);
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
                const int tl = (row >> 5) * (TM * 32) + pass * 32 + (row & 31);
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
    if (a.tail_tgt) {                                        // one atomic per workgroup (256 of them, 8 per log slot)
        __shared__ float red[8];
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

// split-K second pass: sum the partial accumulators, then the same per-slice finish (planes included); lanes along pixels
__global__ __launch_bounds__(256) void h2_splitk_epilogue_kernel(H2Args a) {
    const int spr = a.Cout / 16;                             // slices per row
    const long total = (long)a.M * spr;
    const long slab = (long)a.M * a.Cout;
    int bad = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int sl = (int)(q / a.M), m = (int)(q - (long)sl * a.M), n = sl * 16;
        const float* src = a.partial + (long)m * a.Cout + n;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0.f;
        for (int z = 0; z < a.ksplit; ++z) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + z * slab + 4 * c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[4 * c + k] += t4[k];
            }
        }
        finish16(a, m, n, v, bad);
    }
    rdo::h2_report(bad, a.ovf);
}

// fp32 NHWC tensor [M][C] -> H2 planes (producer of last resort for tensors that no fused kernel writes as planes)
__global__ __launch_bounds__(256) void split_h2_kernel(const float* x, long M, int C, float scale, u16* planes, int* ovf) {
    const int spr = C / 16;
    const long total = M * spr * 2;                          // half slices of 8 channels
    const long pstride = M * C;
    int bad = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int h = (int)(q & 1);
        const long r = q >> 1;
        const int sl = (int)(r % spr);
        const long m = r / spr;
        const float* src = x + m * C + sl * 16 + 8 * h;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
        const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        u32x4 p0, p1;
        split2_x8(v, scale, p0, p1, bad);
        u16* dst = planes + ((long)sl * M + m) * 16 + 8 * h;
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + pstride) = p1;
    }
    rdo::h2_report(bad, ovf);
}

// conv weight [Cout][KH][KW][Cin] (fp32) -> two fp16 planes of weight * scale in fragment order (rdo::frag_index): the operand
// layout of the split-precision kernels for weights that no AdaRound step rewrites (frozen weights, tests)
__global__ __launch_bounds__(256) void split_h2_conv_kernel(const float* w, long numel, int Cout, int KH, int KW, int Cin, float scale, u16* planes,
                                                            int* ovf) {
    int bad = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (long)gridDim.x * blockDim.x)
        rdo::h2_split_store(w[e], scale, planes, numel, rdo::frag_index(e, Cout, KH, KW, Cin), bad);
    rdo::h2_report(bad, ovf);
}

}  // namespace

extern "C" int rdo_conv2d_fwd_uses_bf16x6(const rdo_conv_desc* d);

// K split of the plane kernel: enough 256 x 192 x (K / ks) workgroups for one per CU, at least 9 K stages per split, at most 12 splits
static int h2_ksplit(const rdo_conv_desc* d) {
    const long M = (long)d->B * d->Ho * d->Wo;
    const long tiles = rdo::ceil_div(M, 256) * rdo::ceil_div(d->Cout, 192);
    const long stages = (long)d->KH * d->KW * (d->Cin / 16);
    int best = 0;
    for (int ks = 1; ks <= 12; ++ks)
        if (tiles * ks >= 192 && stages / ks >= 9) {
            best = ks;
            if (tiles * ks >= 256) break;
        }
    return best;
}

// 3 x 3 "same" convs on 16 x 16 patches with too few 256 x 64 tiles for the chip (the 32^2 convs with 192 output channels: 48): the K32
// halo kernel with its slice pairs split over workgroups -- the smallest split that reaches 192 workgroups -- and the second pass of the
// per-tap kernel.  Against the per-tap kernel's 12-way split the partial sums shrink (6 x 3.1 MB instead of 12 x 3.1 MB for those convs)
// and the K loop is the halo kernel's.  0: not applicable.
static int h2k_ksplit(const rdo_conv_desc* d, int* shape_out = nullptr) {
    const int mode = rdo::tuning(rdo::T_H2_K32);
    if (!(rdo::tuning(rdo::T_X6P_HALO) && mode >= 2 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
          d->H % 16 == 0 && d->W % 16 == 0 && d->Cin % 32 == 0 && d->Cout % 64 == 0))
        return 0;
    const long patches = (long)d->B * d->H * d->W / 256;
    const int cpairs = d->Cin / 32;
    // mode 3: 256 x 48 tiles and the LARGEST even split that keeps every workgroup resident at once (one per CU: no second round)
    if (mode >= 3 && d->Cout % 48 == 0) {
        const long tiles = patches * (d->Cout / 48);
        if (tiles < 200 && tiles >= 24) {
            int best = 0;
            for (int ks = 2; ks <= cpairs; ++ks)
                if (cpairs % ks == 0 && tiles * ks <= 256) best = ks;
            if (best) {
                if (shape_out) *shape_out = 3;
                return best;
            }
        }
    }
    const long tiles = patches * (d->Cout / 64);
    if (tiles >= 160 || tiles < 24) return 0;
    if (shape_out) *shape_out = 2;
    for (int ks = 2; ks <= cpairs; ++ks)
        if (cpairs % ks == 0 && tiles * ks >= 192) return ks;
    return 0;
}

static bool pow2_scale(float s) {
    int e;
    return s > 0.f && std::isfinite(s) && std::frexp(s, &e) == 0.5f;
}

extern "C" int rdo_split_h2(const float* x, int64_t npix, int32_t C, float scale, void* planes, void* stream) {
    RDO_REQUIRE(x && planes && npix > 0 && C > 0 && C % 16 == 0, "rdo_split_h2: C (%d) must be a positive multiple of 16", C);
    RDO_REQUIRE(pow2_scale(scale), "rdo_split_h2: scale %g is not a power of two", (double)scale);
    u16* p = reinterpret_cast<u16*>(planes);
    int* ovf = rdo::h2_overflow_flag();
    const double n = (double)npix * C;
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(npix * (C / 8), 256);
            hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, x, (long)npix, C, scale, p, ovf);
            return rdo::check_launch("split_h2");
        },
        stream, "split_h2", 0.0, 8.0 * n);
}

extern "C" int rdo_split_h2_conv(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, float scale, void* planes, void* stream) {
    RDO_REQUIRE(w && planes && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "rdo_split_h2_conv: bad argument");
    RDO_REQUIRE(pow2_scale(scale), "rdo_split_h2_conv: scale %g is not a power of two", (double)scale);
    const long numel = (long)Cout * KH * KW * Cin;
    u16* p = reinterpret_cast<u16*>(planes);
    int* ovf = rdo::h2_overflow_flag();
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(numel, 256);
            hipLaunchKernelGGL(split_h2_conv_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, w, numel, Cout, KH, KW, Cin, scale, p, ovf);
            return rdo::check_launch("split_h2_conv");
        },
        stream, "split_h2_conv", 0.0, 8.0 * (double)numel);
}

extern "C" int rdo_conv2d_fwd_h2_supported(const rdo_conv_desc* d) {
    if (!d || d->square_input || d->Cin % 16 != 0 || d->Cout % 16 != 0) return 0;
    if ((double)d->B * d->H * d->W * d->Cin * 2.0 >= 2147483648.0 || (double)d->Cout * d->KH * d->KW * d->Cin * 2.0 >= 2147483648.0) return 0;
    if (!rdo::tuning(rdo::T_CONV_X6)) return 0;
    return h2_ksplit(d) >= 1;
}

extern "C" int64_t rdo_conv2d_fwd_h2_workspace(const rdo_conv_desc* d) {
    if (!d || !rdo_conv2d_fwd_h2_supported(d)) return 0;
    int ks = h2_ksplit(d);
    if (h2k_ksplit(d) > ks) ks = h2k_ksplit(d);
    return ks > 1 ? (int64_t)ks * d->B * d->Ho * d->Wo * d->Cout : 0;
}

// 3 x 3 "same" convs on 16 x 16 patches run the halo kernel: 1 = with the 256 x 192 tile (enough such tiles to fill the chip without a K
// split), 2 = with the 256 x 64 tile (problems whose 256 x 192 tiles would need a K split but whose 256 x 64 tiles fill >= 5/8 of the
// CUs: the 64^2 convs and the 32^2 convs with 768 output channels), 0 = no
static int h2_halo_shape(const rdo_conv_desc* d) {
    if (!(rdo::tuning(rdo::T_X6P_HALO) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->H % 16 == 0 && d->W % 16 == 0)) return 0;
    if (h2_ksplit(d) == 1) return 1;
    const long patches = (long)d->B * d->H * d->W / 256;
    if (rdo::tuning(rdo::T_X6P_HALO) != 2 && d->Cout % 64 == 0 && patches * (d->Cout / 64) >= 160) return 2;    // "x6p_halo" = 2: wide tile only
    // 3 = the 256 x 48 tile of the K32 kernel for output channel counts in blocks of 48 but not of 64 (the 3 x 3 96 -> 96 convs of
    // Cheng2020-attn's attention blocks at 64^2: 128 tiles, nothing wasted -- the per-tap kernel's 256 x 192 tile ran them half empty)
    if (rdo::tuning(rdo::T_X6P_HALO) != 2 && rdo::tuning(rdo::T_H2_K32) && d->Cout % 48 == 0 && d->Cout % 64 != 0 && d->Cin % 32 == 0 &&
        patches * (d->Cout / 48) >= 96)
        return 3;
    return 0;
}

static int conv2d_fwd_h2_impl(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* wplanes, float w_scale, const float* bias,
                              const float* aux, const void* aux_planes, const float* residual, float* out, float* pre, void* out_planes,
                              float out_scale, float* workspace, int64_t workspace_floats, void* stream, const H2Args* tail) {
    RDO_REQUIRE(d && x_planes && wplanes && (out || out_planes || pre), "rdo_conv2d_fwd_h2: null argument");
    RDO_REQUIRE(rdo_conv2d_fwd_h2_supported(d), "rdo_conv2d_fwd_h2: shape not on the split-precision plane path (rdo_conv2d_fwd_h2_supported)");
    RDO_REQUIRE(pow2_scale(x_scale) && pow2_scale(w_scale) && (!out_planes || pow2_scale(out_scale)),
                "rdo_conv2d_fwd_h2: the scales (%g, %g, %g) must be powers of two", (double)x_scale, (double)w_scale, (double)out_scale);
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd_h2: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    const int epi = d->epilogue;
    RDO_REQUIRE(epi >= RDO_EPI_NONE && epi <= RDO_EPI_RELU_BWD, "rdo_conv2d_fwd_h2: unknown epilogue %d", epi);
    RDO_REQUIRE(epi == RDO_EPI_NONE || epi == RDO_EPI_LRELU || epi == RDO_EPI_RELU || aux != nullptr ||
                    (aux_planes != nullptr && (epi == RDO_EPI_LRELU_BWD || epi == RDO_EPI_RELU_BWD)),
                "rdo_conv2d_fwd_h2: epilogue %d needs aux (aux_planes serve the activation-backward epilogues only)", epi);
    RDO_REQUIRE(!d->add_residual || residual != nullptr, "rdo_conv2d_fwd_h2: add_residual without residual");
    H2Args a{};
    a.xp = reinterpret_cast<const u16*>(x_planes);
    a.xplane = (long)d->B * d->H * d->W * d->Cin;
    a.wp = reinterpret_cast<const u16*>(wplanes);
    a.wplane = (long)d->Cout * d->KH * d->KW * d->Cin;
    a.bias = bias; a.aux = aux; a.auxp = aux ? nullptr : reinterpret_cast<const u16*>(aux_planes); a.residual = residual; a.out = out; a.pre = pre;
    a.outp = reinterpret_cast<u16*>(out_planes);
    a.acc_scale = 1.f / (x_scale * w_scale);
    a.out_scale = out_planes ? out_scale : 1.f;
    a.ovf = rdo::h2_overflow_flag();
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.oplane = (long)a.M * a.Cout;
    a.epilogue = d->epilogue; a.add_residual = d->add_residual;
    a.xcd_mode = rdo::tuning(rdo::T_XCD);
    a.stagger = rdo::tuning(rdo::T_H2_STAGGER);
#ifdef RDO_DIAG
    a.ablate = rdo::tuning(rdo::T_X6P_ABLATE);
#endif
    if (tail) {
        a.tail_tgt = tail->tail_tgt; a.tail_idx = tail->tail_idx; a.tail_iter = tail->tail_iter; a.tail_pub = tail->tail_pub; a.tail_resp = tail->tail_resp;
        a.tail_res_inv = tail->tail_res_inv;
        a.tail_loss = tail->tail_loss; a.tail_per_image = tail->tail_per_image; a.tail_B = tail->tail_B; a.tail_act = tail->tail_act;
        a.tail_coef = tail->tail_coef; a.tail_inv_npix = tail->tail_inv_npix;
    }
    int ks = h2_ksplit(d);
    if (ks < 1) ks = 1;
    if (ks > 1 && (!workspace || (long)ks * a.M * a.Cout > workspace_floats)) ks = 1;
    a.ksplit = ks;
    a.partial = ks > 1 ? workspace : nullptr;
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 4.0 * a.xplane + 4.0 * a.wplane +
                         (double)a.M * a.Cout * (4.0 * ((out != nullptr) + (pre != nullptr) + (aux != nullptr) + (residual != nullptr)) +
                                                 4.0 * (out_planes != nullptr));
    constexpr size_t lds = (size_t)128 * 196 * 4;            // the epilogue's fp32 staging tile is the largest user (98 KiB)
    // 3 x 3 "same" convs on 16 x 16 patches: the halo kernel (no K split: the shapes that qualify fill the chip with tiles)
    const int halo = h2_halo_shape(d);
    RDO_REQUIRE(!tail || halo, "rdo_conv2d_fwd_h2_tail: shape not on the halo kernel (rdo_conv2d_fwd_h2_tail_supported)");
    // few tiles: the K32 halo kernel split over slice pairs + the second pass (h2k_ksplit)
    if (!halo && !tail) {
        int kshape = 2;
        const int ks2 = h2k_ksplit(d, &kshape);
        if (ks2 >= 2 && workspace && (long)ks2 * a.M * a.Cout <= workspace_floats) {
            H2Args k = a;
            k.ksplit = ks2;
            k.partial = workspace;
            if (rdo::h2k_supported(k, kshape))
                return rdo::dispatch(
                    [k, kshape](hipStream_t s) {
                        if (int rc = rdo::h2k_launch(k, kshape, s)) return rc;
                        long g = rdo::ceil_div((long)k.M * k.Cout / 16, 256);
                        hipLaunchKernelGGL(h2_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, k);
                        return rdo::check_launch("h2_splitk_epilogue");
                    },
                    stream, "conv_fwd_h2_halo64k", flops, bytes);
        }
    }
    // ... in its 32-channel-stage / 16x16x32 form where the shape allows (conv_fwd_h2k.hip; tuning key "h2_k32")
    if (halo && rdo::tuning(rdo::T_H2_K32) && (halo >= 2 || ks == 1)) {
        H2Args k = a;
        k.ksplit = 1;
        k.partial = nullptr;
        // 256 x 64 tiles that leave a quarter of the CUs idle (the 64^2 convs with 192 output channels: 192 workgroups) run 256 x 48
        // tiles instead (256 workgroups; tuning key "h2_n48")
        int shape = halo;
        const long patches = (long)d->B * d->H * d->W / 256;
        if (halo == 2 && rdo::tuning(rdo::T_H2_N48) && d->Cout % 48 == 0 && patches * (d->Cout / 64) < 256 && patches * (d->Cout / 48) <= 256 &&
            rdo::h2k_supported(k, 3))
            shape = 3;
        // (measured and dropped: the 256 x 192 tiles as two rounds of 256 x 96 tiles, so that the first round's epilogue stores drain under
        // the second round's K loop -- 96.9 vs 95.9 us back to back, 6.30 vs 6.29 ms per step: the epilogue is not what the loop waits on)
        if (rdo::h2k_supported(k, shape))
            return rdo::dispatch([k, shape](hipStream_t s) { return rdo::h2k_launch(k, shape, s); }, stream,
                                 shape == 1 ? "conv_fwd_h2_halo" : "conv_fwd_h2_halo64", flops, bytes);
    }
    if (halo == 1 && ks == 1)
        return rdo::dispatch(
            [a](hipStream_t s) {
                static_assert(lds >= 2 * 24576 + 3 * 2 * 192 * 32, "halo kernel LDS");
                static rdo::PerDevice attr;
                if (!attr.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_h2h_kernel<4, 2, 2, 3>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_h2h) failed");
                    attr.mark();
                }
                dim3 grid((unsigned)(a.M / 256), (unsigned)rdo::ceil_div(a.Cout, 192), 1);
                hipLaunchKernelGGL((conv_fwd_h2h_kernel<4, 2, 2, 3>), grid, dim3(512), lds, s, a);
                return rdo::check_launch("conv_fwd_h2h");
            },
            stream, "conv_fwd_h2_halo", flops, bytes);
    if (halo == 2) {
        H2Args b = a;
        b.ksplit = 1;
        b.partial = nullptr;
        return rdo::dispatch(
            [b](hipStream_t s) {
                constexpr size_t lds64 = (size_t)2 * 24576 + 3 * 2 * 64 * 32;      // K loop 60 KiB; the epilogue stages 256 x 68 floats = 68 KiB
                constexpr size_t ldsn = lds64 > (size_t)256 * 68 * 4 ? lds64 : (size_t)256 * 68 * 4;
                static rdo::PerDevice attr;
                if (!attr.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_h2h_kernel<8, 1, 1, 2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_h2h 256x64) failed");
                    attr.mark();
                }
                dim3 grid((unsigned)(b.M / 256), (unsigned)(b.Cout / 64), 1);
                hipLaunchKernelGGL((conv_fwd_h2h_kernel<8, 1, 1, 2>), grid, dim3(512), ldsn, s, b);
                return rdo::check_launch("conv_fwd_h2h 256x64");
            },
            stream, "conv_fwd_h2_halo64", flops, bytes);
    }
    return rdo::dispatch(
        [a](hipStream_t s) {
            static_assert(lds >= 3 * 2 * (256 + 192) * 32, "per-tap kernel LDS");
            static rdo::PerDevice attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_h2) failed");
                attr.mark();
            }
            dim3 grid((unsigned)rdo::ceil_div(a.M, 256), (unsigned)rdo::ceil_div(a.Cout, 192), (unsigned)a.ksplit);
            hipLaunchKernelGGL(conv_fwd_h2_kernel, grid, dim3(512), lds, s, a);
            if (int rc = rdo::check_launch("conv_fwd_h2")) return rc;
            if (a.ksplit > 1) {
                long g = rdo::ceil_div((long)a.M * a.Cout / 16, 256);
                hipLaunchKernelGGL(h2_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
                return rdo::check_launch("h2_splitk_epilogue");
            }
            return RDO_OK;
        },
        stream, "conv_fwd_h2_256x192", flops, bytes);
}

extern "C" int rdo_conv2d_fwd_h2(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* wplanes, float w_scale,
                                 const float* bias, const float* aux, const void* aux_planes, const float* residual, float* out, float* pre,
                                 void* out_planes, float out_scale, float* workspace, int64_t workspace_floats, void* stream) {
    return conv2d_fwd_h2_impl(d, x_planes, x_scale, wplanes, w_scale, bias, aux, aux_planes, residual, out, pre, out_planes, out_scale, workspace,
                              workspace_floats, stream, nullptr);
}

extern "C" int rdo_conv2d_fwd_h2_tail_supported(const rdo_conv_desc* d) {
    const int hs = d ? h2_halo_shape(d) : 0;
    return d && rdo_conv2d_fwd_h2_supported(d) && (hs == 1 || hs == 2) && d->epilogue == RDO_EPI_NONE && !d->add_residual;
}

// Last conv of a unit + its tail in ONE launch (halo kernel): the epilogue forms out = act(conv + bias) + residual, the loss against
// the cached target rows and dL/dpre, and writes only the planes of dL/dpre -- rdo_conv2d_fwd_h2 + rdo_loss_act_bwd without the
// pre-activation tensor ever reaching memory.
extern "C" int rdo_conv2d_fwd_h2_tail(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* wplanes, float w_scale,
                                      const float* bias, const void* residual_planes, float residual_scale, const float* tgt_cache,
                                      const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, float coef, int32_t act, void* dpre_planes,
                                      float dpre_scale, float* loss_out, void* stream) {
    RDO_REQUIRE(d && tgt_cache && idx_table && iter_ptr && dpre_planes, "rdo_conv2d_fwd_h2_tail: null argument");
    RDO_REQUIRE(rdo_conv2d_fwd_h2_tail_supported(d), "rdo_conv2d_fwd_h2_tail: shape / epilogue not supported (rdo_conv2d_fwd_h2_tail_supported)");
    RDO_REQUIRE(B == d->B && act >= 0 && act <= 2, "rdo_conv2d_fwd_h2_tail: B must be the conv's batch, act in 0..2");
    RDO_REQUIRE(!residual_planes || pow2_scale(residual_scale), "rdo_conv2d_fwd_h2_tail: residual_scale %g is not a power of two", (double)residual_scale);
    H2Args t{};
    t.tail_tgt = tgt_cache; t.tail_idx = idx_table; t.tail_iter = iter_ptr; t.tail_pub = rdo::take_iter_publish(); t.tail_resp = reinterpret_cast<const u16*>(residual_planes);
    t.tail_res_inv = residual_planes ? 1.f / residual_scale : 1.f;
    t.tail_loss = loss_out; t.tail_per_image = (long)d->Ho * d->Wo * d->Cout; t.tail_B = B; t.tail_act = act; t.tail_coef = coef;
    t.tail_inv_npix = (float)(1.0 / ((double)B * d->Ho * d->Wo));
    return conv2d_fwd_h2_impl(d, x_planes, x_scale, wplanes, w_scale, bias, nullptr, nullptr, nullptr, nullptr, nullptr, dpre_planes, dpre_scale,
                              nullptr, 0, stream, &t);
}

#ifdef RDO_DIAG
// diagnostic build only: per workgroup {shader clock at K-loop start, 100 MHz wall ticks at start, clock at end, ticks at end} of the
// last halo-kernel launch (tools/bench_h2.py H2_STAMPS=1)
extern "C" int rdo_diag_h2_stamps(unsigned long long* out, int n) {
    if (!out || n <= 0 || n > 1024) return RDO_EINVAL;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? RDO_OK : RDO_EHIP;
}
#endif
