// Weight gradient of the LARGE problems on "H2" tensors (exact two-way fp16 splits of the scaled values in slice-major planes,
// rdo_common.h / rdo_ptq_hip.h):
//   dw[co][tap][ci] = sum_m dy[m][co] * x[pix(m, tap)][ci]      on v_mfma_f32_16x16x32_f16, three products, fp32 accumulate.
//
// The fp32-input kernel (conv_wgrad_x6.hip) spends 4.6 vector instructions per MFMA in its loader: it splits both operands for
// every tap workgroup again and transposes them in registers, because the MFMA wants 8 consecutive PIXELS per lane while NHWC memory
// has channels contiguous.  Here both operands arrive by LDS-DMA exactly as they lie in memory -- [pixel][channel] records of the
// planes the producers wrote -- and the transposition is done by the LDS itself: ds_read_b64_tr_b16 hands every lane the four pixels
// of its channel.  The K loop contains no conversion and no permutation arithmetic.
//
// Tile 192 (co) x 192 (ci) per tap, 512 threads: wave (w >> 1, w & 1) owns 48 x 96 as 3 x 6 MFMA tiles, 32 pixels per stage.
// LDS image of a stage, per operand and plane: [3 channel sub-tiles of 64][32 pixels][8 chunks of 16 B] = 12 KiB, i.e. 48 KiB per
// stage, double buffered.  A DMA piece is 8 pixels x 128 bytes of one sub-tile (1 KiB, lane l -> pixel l >> 3, chunk l & 7); waves
// 0-3 fetch dY, waves 4-7 fetch X, six pieces each per stage, all for ONE pixel row per lane.  The DMA of stage s+1 is issued at
// the top of stage s: a full stage to land.
// Bank conflicts: chunk position c' = c ^ 2 ((pixel >> 1) & 3) (applied on the per-lane SOURCE address); a transposed read of a
// 32-lane half covers 8 consecutive pixels x 32 bytes and then touches all 64 banks once.
// K slots: the MFMA sums over 32 k values and both operands may use ANY common assignment of pixels to k slots.  Lane group g
// (lanes 16g..16g+15) holds pixels 4g..4g+3 (first read) and 16+4g..16+4g+3 (second read) of the stage.
#include <utility>

#include "rdo_common.h"

namespace {

typedef float f32x4acc __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int NP = 2;                          // planes of an H2 tensor
typedef unsigned short u16;

__device__ __attribute__((aligned(64))) unsigned g_zero_page_w[16];

constexpr int PK = 32;
constexpr int SUBB = 32 * 128;                 // bytes of one [32 pixels][64 channels] sub-tile image
constexpr int PLANEB = 3 * SUBB;               // 192 channels of one plane
constexpr int OPB = NP * PLANEB;               // one operand (two planes)
constexpr int STAGEB = 2 * OPB;                // 48 KiB

struct WgPArgs {
    const u16* xp;
    const u16* yp;
    long xplane, yplane;       // elements per plane
    float* slabs;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, Min, mchunk, nsplit;
    int tiles_co, tiles_ci;
    float inv_scale;           // 1 / (s_x * s_dy): the accumulators hold s_x s_dy times the gradient sums
    int ablate;                // diagnostic build only (tuning key "x6p_ablate"): 1 no DMA, 4 no MFMA, 8 no fragment reads (row kernel)
};
#ifdef RDO_DIAG
#define WG_ABL(bit) (a.ablate & (bit))
__device__ unsigned long long g_wg_stamps[256 * 4];
__device__ unsigned long long g_wg_phase[256 * 8 * 8];      // [workgroup][wave][phase]: cycles summed over the stages (ablate bit 64)
#else
#define WG_ABL(bit) false
#endif

// Two transposed reads = the eight pixels of one lane's MFMA operand.  Issued as INLINE ASSEMBLY on purpose: for the builtin
// (__builtin_amdgcn_ds_read_tr16_b64_v4i16) hipcc's wait-count pass assumes that the read may alias every LDS-DMA in flight and puts
// `s_waitcnt vmcnt(0)` in front of the first such read behind a global_load_lds -- every wave then waited for its just-issued DMA
// pieces to LAND at the top of every stage (the "739 cycles of DMA issue" of tools/wgrad_phases.py), whatever the ring depth; plain
// LDS loads do not get that wait.  The asm is invisible to that pass, so the caller orders reads and uses itself: `tr_wait()` (lgkmcnt(0)
// + a scheduling fence, cdna_hip_programming.md rule 18) between the last read of a set and its first use; which buffer a read may
// touch is the K loop's counted vmcnt + barrier, as for every other DMA consumer.
__device__ __forceinline__ f16x8 tr_pair(const char* p) {
    typedef __attribute__((address_space(3))) const char lds_char;
    const unsigned addr = (unsigned)(unsigned long long)(lds_char*)p;
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(addr));      // pixels + 16 (16 rows of 128 bytes)
    return __builtin_bit_cast(f16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512, 2) void conv_wgrad_h2_kernel(WgPArgs a) {
    constexpr int T = 192;
    constexpr int TM = 3, TN = 6;

    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][STAGEB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lc = lane >> 4;
    const int wco0 = (wave >> 1) * 48, wci0 = (wave & 1) * 96;

    // XCD-aware numbering, tap fastest (as conv_wgrad_x6.hip)
    const int nwg = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int chunk = lid / gridDim.y;
    int t = lid - chunk * gridDim.y;
    const int tci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tco = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = tco * T, ci0 = tci * T;

    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + PK - 1) / PK : 0;
    const int HoWo = a.Ho * a.Wo;

    // ---- loader: waves 0-3 fetch dY, waves 4-7 fetch X; this lane always serves pixel row 8 (wave & 3) + (lane >> 3) of the stage
    const bool ldx = wave >= 4;
    const int prow = (wave & 3) * 8 + (lane >> 3);
    const int cpos = lane & 7;
    const int csrc = cpos ^ (2 * ((prow >> 1) & 3));                     // source chunk that belongs at LDS position cpos
    const u16* const zero = reinterpret_cast<const u16*>(g_zero_page_w);
    // element offset of this lane's chunk inside a plane, without the pixel: sub-tile `sub` adds 4 slices
    const int ch0 = (ldx ? ci0 : co0) + 8 * csrc;                        // first channel of the chunk in sub-tile 0
    const int Cop = ldx ? a.Cin : a.Cout;
    const long Mop = ldx ? a.Min : a.M;
    const long pstride = ldx ? a.xplane : a.yplane;
    const u16* const opbase = ldx ? a.xp : a.yp;
    // running output-pixel position of this lane (advances by 32 per stage)
    int st_m = mbeg + prow;
    int st_wo = 0, st_ho = 0, st_hi = 0, st_wi = 0, st_pix = 0;
    if (ldx) {
        const int mm = st_m < a.M ? st_m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        st_ho = rem / a.Wo;
        st_wo = rem - st_ho * a.Wo;
        st_hi = st_ho * a.stride - a.pad + kh;
        st_wi = st_wo * a.stride - a.pad + kw;
        st_pix = (b * a.H + st_hi) * a.W + st_wi;
    }
    auto advance = [&]() {
        st_m += PK;
        if (!ldx) return;
        st_wo += PK;
        st_wi += PK * a.stride;
        st_pix += PK * a.stride;
        while (st_wo >= a.Wo) {
            st_wo -= a.Wo;
            st_wi -= a.Wo * a.stride;
            st_pix += (a.stride * a.W - a.Wo * a.stride);
            st_hi += a.stride;
            if (++st_ho >= a.Ho) {
                st_ho = 0;
                st_hi -= a.Ho * a.stride;
                st_pix += (a.H * a.W - a.Ho * a.stride * a.W);
            }
        }
    };
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma_stage = [&](int buf) {
        bool ok = st_m < mend;
        long pix = st_m;
        if (ldx) {
            ok = ok && (unsigned)st_hi < (unsigned)a.H && (unsigned)st_wi < (unsigned)a.W;
            pix = st_pix;
        }
        char* dst = smem + buf * STAGEB + (ldx ? OPB : 0) + (wave & 3) * 1024;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            const int ch = ch0 + 64 * sub;
            const bool okc = ok && ch < Cop;
            const u16* src = okc ? opbase + ((long)(ch >> 4) * Mop + pix) * 16 + (ch & 15) : zero;
            const long ps = okc ? pstride : 0;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(src + p * ps), (lds_void*)(dst + p * PLANEB + sub * SUBB), 16, 0, 0);
        }
        advance();
    };

    f32x4acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses: lane (g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3) supplies pixel 4g + q, channels 4 p4 .. 4 p4 + 3
    const int fr = 4 * lc + ((lane >> 2) & 3);                           // pixel row of the first read
    const int p4 = lane & 3;
    const int fsw = 2 * ((fr >> 1) & 3);
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int c = wco0 + 16 * i;                                     // first channel of the tile inside the 192-wide image
        const int chunkpos = (2 * ((c & 63) >> 4) + (p4 >> 1)) ^ fsw;
        fa_off[i] = (c >> 6) * SUBB + fr * 128 + chunkpos * 16 + (p4 & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = wci0 + 16 * j;
        const int chunkpos = (2 * ((c & 63) >> 4) + (p4 >> 1)) ^ fsw;
        fb_off[j] = OPB + (c >> 6) * SUBB + fr * 128 + chunkpos * 16 + (p4 & 1) * 8;
    }
    (void)l16;

    constexpr int PA[3] = {1, 0, 0};             // (dy2, x1) (dy1, x2) (dy1, x1): small terms first
    constexpr int PB[3] = {0, 1, 0};

    if (nsteps > 0) dma_stage(0);
    f16x8 fa[NP][TM], fb[2][NP][2];
    auto read_b = [&](auto setc, const char* st, int third) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[S][p][j] = tr_pair(st + p * PLANEB + fb_off[2 * third + j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's pieces of stage s have landed
        __builtin_amdgcn_s_barrier();                                    // ... everybody's have, and nobody reads buffer buf ^ 1 any more
        // The two waves of a SIMD (w and w + 4) run this loop in lock-step: if both issued their nine DMAs here, the matrix pipe would
        // idle for the whole issue phase.  Waves 0-3 (dY) issue now, waves 4-7 (X) after their first three MFMA slots -- while one wave
        // of a SIMD is held up issuing, its partner multiplies.  X still has two thirds of a stage to land.
        const bool more = s + 1 < nsteps;
        if (more && !ldx) dma_stage(buf ^ 1);
        const char* st = smem + buf * STAGEB;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = tr_pair(st + p * PLANEB + fa_off[i]);
        read_b(S0{}, st, 0);
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
                 constexpr int T3 = SL / 3, Q = SL % 3, SET = T3 & 1;
                 if constexpr (Q == 0) {
                     tr_wait();                                          // this third's fragments (and, in slot 0, the A fragments) are there
                     if constexpr (T3 < 2) {
                         if constexpr (SET == 0) read_b(S1{}, st, T3 + 1);
                         else read_b(S0{}, st, T3 + 1);
                     }
                 }
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < 2; ++j)
                         acc[i][2 * T3 + j] =
                             __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T3 + j], 0, 0, 0);
                 __builtin_amdgcn_sched_barrier(0);
                 if constexpr (SL == 2) {
                     if (more && ldx) dma_stage(buf ^ 1);
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 9>{});
    }

    const long wsize = (long)a.Cout * a.KH * a.KW * a.Cin;
    float* slab = a.slabs + (long)chunk * wsize;
    const int taps = a.KH * a.KW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wci0 + j * 16 + l16;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wco0 + i * 16 + 4 * lc + r;
                if (co < a.Cout) slab[((long)co * taps + tap) * a.Cin + ci] = acc[i][j][r] * a.inv_scale;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Token-matrix weight gradient from FP32 operands (round 5):  dW[n][k] = sum_t dY[t][n] X[t][k]  -- the weight gradients of the
// trainable Linears of the Lu2022 blocks (quant_layer.py:119 wrapped nn.Linear; 1 x 1 "convs" over [tokens][channels]) and the gamma
// gradient of a GDN (X = the squared input, quant_layer.py:147).  They ran on the split-bf16 kernel (conv_wgrad_x6.hip: six products,
// both operands split and transposed in registers by its loader -- 112 us for the 576 x 192 gradient over 64 K tokens, 1.8 TB/s).
// Here: the MFMA side of conv_wgrad_h2_kernel above unchanged (LDS images [pixel][channel] of two fp16 planes, transposed reads, three
// products) behind a loader that takes fp32 rows, scales a 32-token stage by a power of two and splits it in registers.
// Scale: the reduction runs over tokens, so a scale has to be common to all 32 tokens of a stage (one MFMA sums them); it is taken from
// the largest magnitude of the stage (-> [2^13, 2^14)), never larger than the scale of the stages before it, and the accumulators are
// multiplied by the (exact, <= 1) ratio when a new largest stage arrives: products of small stages carry the running scale -- their
// low bits fall below the fp32 accumulator's own resolution for the sum, like the small terms of any fp32 accumulation.
// Pipeline per 32-token stage (one barrier): split stage s+1 (registers -> the other LDS image) | request stage s+3 | largest magnitude
// of stage s+2 (landed) -> LDS slots | products of stage s.  Requests are inline asm with counted waits (see linear_h2.hip).
struct LwArgs {
    const float* x;      // [M][Cin]
    const float* dy;     // [M][Cout]
    float* slabs;        // [nsplit][Cout][Cin]
    int M, Cin, Cout, mchunk, nsplit, tiles_co, tiles_ci;
};

typedef float lw4 __attribute__((ext_vector_type(4)));
typedef _Float16 lwh8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float lw_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

template <bool SQ>
__global__ __launch_bounds__(512, 2) void linear_wgrad_h2_kernel(LwArgs a) {
    constexpr int T = 192;
    constexpr int TM = 3, TN = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][STAGEB], then the small tables
    float* amax_slot = reinterpret_cast<float*>(smem + 2 * STAGEB);      // [2 (stage parity)][2 (dY, X)][4 waves]
    int* stage_exp = reinterpret_cast<int*>(amax_slot + 16);             // [4 (stage & 3)][2]: log2 of the scale of (dY, X)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lc = lane >> 4;
    const int wco0 = (wave >> 1) * 48, wci0 = (wave & 1) * 96;

    const int nwg = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int chunk = lid / gridDim.y;
    int t = lid - chunk * gridDim.y;
    const int tci = t % a.tiles_ci;
    const int tco = t / a.tiles_ci;
    const int co0 = tco * T, ci0 = tci * T;
    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg) / PK : 0;             // (M and mchunk are multiples of 32)

    // ---- loader roles: waves 0-3 take dY, waves 4-7 take X; thread -> token row (t >> 3) of the stage, 8-channel chunk (t & 7) of each of
    // the three 64-channel sub-tiles: two float4 per chunk
    const bool ldx = wave >= 4;
    const int lt = tid & 255, prow = lt >> 3, cpos = lt & 7;
    const int Cop = ldx ? a.Cin : a.Cout;
    // channel counts that do not fill the 192-wide tile (96: the 1 x 1 convs of Cheng2020-attn's attention blocks): chunks past the tensor
    // re-read the row's first chunk (the request count per wave stays six) and are zeroed when they land
    const int cbase = (ldx ? ci0 : co0) + 8 * cpos;
    bool cok[3];
#pragma unroll
    for (int sub = 0; sub < 3; ++sub) cok[sub] = cbase + 64 * sub < Cop;
    const float* const rowbase = (ldx ? a.x : a.dy) + (long)(mbeg + prow) * Cop;
    char* const ldst = smem + (ldx ? OPB : 0) + prow * 128 + ((cpos ^ (2 * ((prow >> 1) & 3))) * 16);
    lw4 R[2][6];
    auto request = [&](lw4 (&r)[6], int stage) {
        const int st = stage < nsteps ? stage : (nsteps > 0 ? nsteps - 1 : 0);      // past the end: a valid stage again (never multiplied)
        const float* row = rowbase + (long)st * PK * Cop;
        const float* s0 = row + (cok[0] ? cbase : 0);
        const float* s1 = row + (cok[1] ? cbase + 64 : 0);
        const float* s2 = row + (cok[2] ? cbase + 128 : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[0]) : "v"(s0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[1]) : "v"(s0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[2]) : "v"(s1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[3]) : "v"(s1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[4]) : "v"(s2) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[5]) : "v"(s2) : "memory");
    };
    // the six requests issued BEFORE the youngest six have landed -> largest magnitude of the stage into this wave's slot
    auto landed_amax = [&](lw4 (&r)[6], int parity) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(r[k]));
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (!cok[k >> 1]) r[k] = lw4{0.f, 0.f, 0.f, 0.f};
            if (SQ && ldx) r[k] *= r[k];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(r[k][0]), fabsf(r[k][1])), fmaxf(fabsf(r[k][2]), fabsf(r[k][3]))));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) amax_slot[parity * 8 + (ldx ? 4 : 0) + (wave & 3)] = m;
    };
    int erun = 100;                                                      // log2 of this operand's running scale (never grows)
    auto split = [&](lw4 (&r)[6], int stage) {
        const float* sl = amax_slot + (stage & 1) * 8 + (ldx ? 4 : 0);
        const float m = fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]));
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFF) - 127;
        int own = e < -100 ? 100 : 13 - e;                               // zeros / denormals: any scale; inf / NaN propagate through fp16
        own = own < -100 ? -100 : own;
        erun = own < erun ? own : erun;
        if (lt == 0) stage_exp[(stage & 3) * 2 + (ldx ? 1 : 0)] = erun;
        const float sc = lw_pow2(erun);
        char* dst = ldst + (stage & 1) * STAGEB;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            const lw4 v0 = r[2 * sub] * sc, v1 = r[2 * sub + 1] * sc;
            lwh8 hi, lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = (_Float16)v0[k];
                hi[4 + k] = (_Float16)v1[k];
                lo[k] = (_Float16)(v0[k] - (float)hi[k]);
                lo[4 + k] = (_Float16)(v1[k] - (float)hi[4 + k]);
            }
            *reinterpret_cast<lwh8*>(dst + sub * SUBB) = hi;
            *reinterpret_cast<lwh8*>(dst + PLANEB + sub * SUBB) = lo;
        }
    };

    f32x4acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses (as conv_wgrad_h2_kernel)
    const int fr = 4 * lc + ((lane >> 2) & 3);
    const int p4 = lane & 3;
    const int fsw = 2 * ((fr >> 1) & 3);
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int c = wco0 + 16 * i;
        const int chunkpos = (2 * ((c & 63) >> 4) + (p4 >> 1)) ^ fsw;
        fa_off[i] = (c >> 6) * SUBB + fr * 128 + chunkpos * 16 + (p4 & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = wci0 + 16 * j;
        const int chunkpos = (2 * ((c & 63) >> 4) + (p4 >> 1)) ^ fsw;
        fb_off[j] = OPB + (c >> 6) * SUBB + fr * 128 + chunkpos * 16 + (p4 & 1) * 8;
    }
    constexpr int PA[3] = {1, 0, 0};             // (dy2, x1) (dy1, x2) (dy1, x1): small terms first
    constexpr int PB[3] = {0, 1, 0};

    int Ey = 200, Ex = 200;                      // log2 of the scales the accumulators carry
    if (nsteps > 0) {                            // (uniform; a chunk past the end of the tokens writes a slab of zeros)
    // ---- prologue: stage 0 split, stage 1 measured, stage 2 requested
    request(R[0], 0);
    request(R[1], 1);
    landed_amax(R[0], 0);
    __syncthreads();
    split(R[0], 0);
    request(R[0], 2);
    landed_amax(R[1], 1);

    f16x8 fa[NP][TM], fb[2][NP][2];
    auto read_b = [&](auto setc, const char* st, int third) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[S][p][j] = tr_pair(st + p * PLANEB + fb_off[2 * third + j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto stage = [&](auto parc, int s) {
        constexpr int PAR = decltype(parc)::value;                       // s & 1: stage s + 1 sits in R[PAR ^ 1], stage s + 2 in R[PAR]
        __syncthreads();                                                 // image s, its scales and the slots of stage s + 1 are complete
        split(R[PAR ^ 1], s + 1);
        request(R[PAR ^ 1], s + 3);
        landed_amax(R[PAR], s & 1);                                      // stage s + 2 (parity (s + 2) & 1)
        // accumulators follow the running scale (a new largest stage: exact ratio <= 1)
        const int ey = stage_exp[(s & 3) * 2], ex = stage_exp[(s & 3) * 2 + 1];
        if (ey + ex != Ey + Ex) {                                        // (uniform)
            int d = (ey + ex) - (Ey + Ex);
            d = d < -126 ? -126 : d;
            const float ratio = s == 0 ? 1.f : lw_pow2(d);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] *= ratio;
            Ey = ey; Ex = ex;
        }
        const char* st = smem + (s & 1) * STAGEB;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = tr_pair(st + p * PLANEB + fa_off[i]);
        read_b(S0{}, st, 0);
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
                 constexpr int T3 = SL / 3, Q = SL % 3, SET = T3 & 1;
                 if constexpr (Q == 0) {
                     tr_wait();
                     if constexpr (T3 < 2) {
                         if constexpr (SET == 0) read_b(S1{}, st, T3 + 1);
                         else read_b(S0{}, st, T3 + 1);
                     }
                 }
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < 2; ++j)
                         acc[i][2 * T3 + j] =
                             __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T3 + j], 0, 0, 0);
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 9>{});
    };
    for (int s = 0; s < nsteps; s += 2) {
        stage(S0{}, s);
        if (s + 1 < nsteps) stage(S1{}, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the clamped tail requests)
    } else {
        Ey = Ex = 0;
    }

    const float fy = lw_pow2(-Ey), fx = lw_pow2(-Ex);                   // |E| <= 100 each: two exact factors
    float* slab = a.slabs + (long)chunk * a.Cout * a.Cin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wci0 + j * 16 + l16;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wco0 + i * 16 + 4 * lc + r;
                if (co < a.Cout) slab[(long)co * a.Cin + ci] = (acc[i][j][r] * fy) * fx;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Row variant for 3 x 3 / stride 1 / pad 1 with Wo a multiple of 32 and Cin a multiple of 64: a workgroup owns the THREE kw taps of one
// kh for a block of 64 input channels -- still a 192 x 192 output tile (192 co x [3 taps x 64 ci]) -- and a stage is 32 consecutive
// pixels of ONE output row.  The three taps read the same input row shifted by one pixel, so the X image of a stage is 34 pixels x 64
// channels (8.5 KiB for the two planes) instead of 32 x 192 (24 KiB): 33 KiB of L2 -> LDS traffic per stage instead of 48, for the
// same 54 MFMAs per wave.  (Why it matters: these kernels sit at the board power limit, tools/clock_probe.py; the forward kernel
// gained 11 % from the same kind of cut.)  A tap is a row offset when the B fragments are read; everything else -- LDS images, chunk
// swizzle, transposed reads, k-slot assignment, MFMA order -- is the kernel above.
constexpr int XROWS = 40;                      // 34 used; five 8-row DMA pieces per plane
constexpr int XPLANEB = XROWS * 128;
constexpr int STAGE3B = OPB + NP * XPLANEB;    // 24 + 10 = 34 KiB

// Template parameters = the stage shapes behind the tuning key "wgrad_sub" (all give the same bits):
//   SUB    32-pixel segments per stage (per barrier / DMA wait); 2 = two segments per ring slot, half as many barriers (136 KiB of LDS)
//   RING3  LDS ring slots (RING3 - 1 stages of DMA in flight)
//   LSLOT  the MFMA slot (0..8) behind which waves 4-7 issue their DMAs (waves 0-3 issue at the top of the stage)
//   PF     1 / 2: the fragments of the NEXT stage's first third are read one stage ahead (ring of three); 2 = those reads interleaved
//          one pair per MFMA into the last third.  <1, 3, 2, 2> is the shipped one ("wgrad_sub" = 13).
// What the in-kernel phase stamps (tools/wgrad_phases.py, all eight waves) and the ablations (tools/wgrad_ablate.py) showed, in the order
// it was found: (1) issuing an LDS-DMA piece cost a wave ~130-150 cycles when four waves issue together -- the pixel-major lane roles
// made the texture addresser take 32 cycles per piece (lane quads straddling two slices) and every piece carried ~20 scalar / vector
// instructions of pointer selection; slice-major pieces through buffer descriptors (below) halve both.  (2) With that gone, waves 0-3 wait
// ~900 cycles per stage at the barrier for waves 4-7, whose MFMAs share the SIMD's pipe with their partner exactly when both multiply:
// the look-ahead reads (PF) let both start right behind the barrier.  (3) In a sustained stream the loop runs at 2.1-2.2 GHz: at the
// board's power limit, like the forward kernel.
template <int SUB, int RING3, int LSLOT, int PF = 0>
__global__ __launch_bounds__(512, 2) void conv_wgrad_h2r_kernel(WgPArgs a) {
    constexpr int T = 192;
    constexpr int TM = 3, TN = 6;
    constexpr int SLOTB = SUB * STAGE3B;                                 // one ring slot

    extern __shared__ __attribute__((aligned(16))) char smem[];          // [RING3][SUB][STAGE3B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lc = lane >> 4;
    const int wco0 = (wave >> 1) * 48;

    // XCD-aware numbering, (kh, ci block, co tile) fastest
    const int nwg = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int chunk = lid / gridDim.y;
    int t = lid - chunk * gridDim.y;
    const int cib = t % a.tiles_ci; t /= a.tiles_ci;                     // tiles_ci = Cin / 64 here
    const int tco = t % a.tiles_co; t /= a.tiles_co;
    const int kh = __builtin_amdgcn_readfirstlane(t);                   // (the divisions above run on the vector unit: tell the compiler it is uniform)
    const int co0 = tco * T, ci0 = cib * 64;

    // stages = 32-pixel segments of output rows; this chunk owns segments [sbeg, send)
    const int segs_row = a.Wo / PK;
    const int segs_total = a.B * a.Ho * segs_row;
    const int spc = (segs_total + a.nsplit - 1) / a.nsplit;
    const int sbeg = chunk * spc;
    const int send = min(segs_total, sbeg + spc);
    const int nsegs = send > sbeg ? send - sbeg : 0;
    const int nsteps = (nsegs + SUB - 1) / SUB;                          // a short last stage multiplies zeros for its missing segment

    // DMA lane roles: 16-channel slice (lane >> 4) of the piece's 64 channels, pixel ((lane >> 1) & 7) of its eight, half slice (lane & 1).
    // Sixteen consecutive lanes read 256 contiguous bytes of one slice: the texture addresser takes such a piece in 16 cycles, against 32
    // for the pixel-major roles of round 3 whose lane quads straddled two slices (tools/micro/dma_rate.hip).  The LDS image of a piece is
    // therefore [slice][pixel][32 B]; a transposed fragment read covers 2 x 256 contiguous bytes, so no swizzle is needed.
    const int prow8 = (lane >> 1) & 7, dsl = lane >> 4, dhalf = lane & 1;
    // The pieces go through buffer descriptors (`buffer_load_dwordx4 ... lds`): a lane whose offset fails the descriptor's range check
    // writes ZEROS to LDS (tools/micro/buf_lds.hip), which is the padding -- no zero-page pointer select per lane -- and the per-stage
    // part of an address is one scalar offset.  A piece costs a v_cndmask (where lanes can be padding), an M0 write and the load.
    // dY pieces of this wave: k = wave + 8 j of the 24 [plane][sub-tile][8-row group] pieces
    constexpr int XP = NP * 5;                                          // X pieces (dY: 24 = three per wave)
    constexpr unsigned OOB = 0xFFFFFFFFu;
    constexpr int RSRC3 = 0x00020000;                                   // raw buffer, 32-bit data format
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned ybytes = (unsigned)(a.yplane * 2), xbytes = (unsigned)(a.xplane * 2);
    __amdgpu_buffer_rsrc_t ry[3], rx[2];
    unsigned yvoff[3], xvoff[2];
    int ydst[3], xdst[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = wave + 8 * j;
        const int plane = k / 12, sub = (k % 12) >> 2, rg = k & 3;
        const int row = 8 * rg + prow8;
        const int ch = co0 + 64 * sub + 16 * dsl + 8 * dhalf;
        ry[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.yp + plane * a.yplane), 0, ybytes, RSRC3);
        yvoff[j] = ch < a.Cout ? (unsigned)(((ch >> 4) * a.M + row) * 16 + (ch & 15)) * 2u : OOB;
        ydst[j] = plane * PLANEB + sub * SUBB + rg * 1024;
    }
    // X pieces: k = wave + 8 j of the 10 [plane][8-row group] pieces; row r of the image is input pixel wo0 - 1 + r.  The descriptor
    // starts one pixel BEFORE the plane so that the stage offset (pixel of row 1) is never negative; the lanes of a row left of the image
    // are padding and never touch that pixel.
    int xedge[2];                                                       // bit 0: left-most image row, bit 1: right-most, bit 2: always
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = wave + 8 * j < XP ? wave + 8 * j : wave;           // (waves 2-7 repeat their one piece: five DMAs per wave and stage)
        const int plane = k / 5, rg = k - plane * 5;
        const int row = 8 * rg + prow8;
        const int ch = ci0 + 16 * dsl + 8 * dhalf;
        rx[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xp + plane * a.xplane - 16), 0, xbytes + 32u, RSRC3);
        xvoff[j] = row < PK + 2 ? (unsigned)(((ch >> 4) * a.Min + row) * 16 + (ch & 15)) * 2u : OOB;
        xedge[j] = (row == 0 ? 1 : 0) | (row == PK + 1 ? 2 : 0) | 4;
        xdst[j] = OPB + plane * XPLANEB + rg * 1024;
    }
    // Segment cursor.  Rows are whole segments (Wo % 32 == 0, Wo == W, Ho == H), so the first output pixel of segment `seg` is 32 seg and
    // both stage offsets are LINEAR in it: dY at pixel 32 seg, X (image row 1) at pixel 32 seg + (kh - 1) W.  Only the padding needs the
    // position inside the image: a segment counter per row (left / right edge) and a row counter per image (top / bottom edge for kh 0 / 2).
    // (Phase stamps, tools/wgrad_phases.py: ~60 scalar instructions of cursor and address arithmetic per stage cost every wave ~260 cycles
    // at the top of the stage, when nobody multiplies.)
    int seg = sbeg;
    int cho = (sbeg / segs_row) % a.Ho;
    int cws = sbeg % segs_row;
    unsigned ysoff = (unsigned)sbeg * (PK * 32u);
    unsigned xsoff = (unsigned)(sbeg * PK + (kh - 1) * a.W) * 32u;         // (wraps below zero only where every lane is padding)
    const int row_bad = kh == 0 ? 0 : kh == 2 ? a.Ho - 1 : -1;            // the output row whose input row hi = ho + kh - 1 is outside
    auto dma_seg = [&](char* const dst) {
        const bool live = seg < send;                                    // beyond the chunk: zeros (the stage's MFMAs add nothing)
        const int emask = (cws == 0 ? 1 : 0) | (cws == segs_row - 1 ? 2 : 0) | ((!live || cho == row_bad) ? 4 : 0);   // padding of this stage
        const unsigned ys = ysoff, xs = xsoff;
        ++seg; ysoff += PK * 32u; xsoff += PK * 32u;
        if (++cws == segs_row) { cws = 0; if (++cho == a.Ho) cho = 0; }
        if (WG_ABL(1)) return;
#pragma unroll
        for (int j = 0; j < 3; ++j)                                      // 24 pieces: three per wave
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ry[j], (lds_void*)(dst + ydst[j]), 16, live ? yvoff[j] : OOB, ys, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (RING3 == 2 && wave + 8 * j >= XP) break;                 // (no counted wait: the repeated piece is not needed)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx[j], (lds_void*)(dst + xdst[j]), 16, (xedge[j] & emask) ? OOB : xvoff[j], xs, 0, 0);
        }
    };
    auto dma_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < SUB; ++u) dma_seg(smem + buf * SLOTB + u * STAGE3B);
    };

    f32x4acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses (lane roles as above); column tile jt = 6 (wave & 1) + j of the 12: tap kw = jt / 4, channels 16 (jt % 4)
    const int fr = 4 * lc + ((lane >> 2) & 3);
    const int p4 = lane & 3;
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int c = wco0 + 16 * i;
        fa_off[i] = (c >> 6) * SUBB + (fr >> 3) * 1024 + ((c & 63) >> 4) * 256 + (fr & 7) * 32 + p4 * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jt = 6 * (wave & 1) + j;
        const int row = fr + (jt >> 2);                                  // tap kw: input pixel = output pixel + kw - 1 = image row + kw
        fb_off[j] = OPB + (row >> 3) * 1024 + (jt & 3) * 256 + (row & 7) * 32 + p4 * 8;
    }

    constexpr int PA[3] = {1, 0, 0};             // (dy2, x1) (dy1, x2) (dy1, x1): small terms first
    constexpr int PB[3] = {0, 1, 0};

    // RING3 stages, RING3 - 1 in flight.  In-kernel clocks (tools/h2_stamps.py, 4 x 128^2, cycles per stage at 2.36-2.39 GHz -- this
    // kernel is NOT at the power limit): complete 3436, DMA ablated 2521, fragment reads ablated 2532, MFMAs only 2094 (1728 = back to
    // back), no MFMA 2394, skeleton 609.  The DMA share is issue cost and LDS write bandwidth against the transposed reads, not
    // latency: a ring of three (two stages in flight, counted vmcnt(5), five DMAs per wave and stage) ran 3660 cycles per stage.
    const bool late = wave >= 4;
    if (nsteps > 0) {
#pragma unroll
        for (int r = 0; r < RING3 - 1; ++r) dma_stage(r);
    }
    if constexpr (PF != 0) {
        // PF: the fragments of a stage's FIRST third (A: 6 register sets, B: 4) are read one stage ahead, during the previous stage's last
        // third, so that MFMAs start right behind the barrier instead of ~500 cycles of read issue and latency later (phase stamps: the
        // MFMA pipe idled from the barrier until the late waves' first fragments arrived, ~750 of ~3000 cycles).  Needs a ring of three:
        // buffer s + 1 is read in stage s while DMA(s + 2) lands in the third.  The B register set of third T3 alternates with the stage
        // parity, (T3 + P) & 1, so the look-ahead always targets the set the running third does not use.
        static_assert(SUB == 1 && RING3 == 3, "look-ahead variant: one segment per stage, ring of three");
        f16x8 fa[2][NP][TM], fb[2][NP][2];
        auto rd_a = [&](auto pc, const char* st) {
            constexpr int P = decltype(pc)::value;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[P][p][i] = tr_pair(st + p * PLANEB + fa_off[i]);
        };
        auto rd_b = [&](auto setc, const char* st, int third) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[S][p][j] = tr_pair(st + p * XPLANEB + fb_off[2 * third + j]);
        };
#ifdef RDO_DIAG
        const unsigned long long st_c0 = clock64(), st_r0 = wall_clock64();
#endif
        if (nsteps > 0) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                // stage 0 has landed (five DMAs per wave and stage)
            __builtin_amdgcn_s_barrier();
            rd_a(std::integral_constant<int, 0>{}, smem);
            rd_b(std::integral_constant<int, 0>{}, smem, 0);
        }
        auto stage = [&](auto pc, const int s, const int buf) {
            constexpr int P = decltype(pc)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // my DMAs of stage s + 1 have landed ...
            __builtin_amdgcn_s_barrier();                                    // ... everybody's have, and nobody reads buffer (s - 1) % 3 any more
            const int nb = buf >= 1 ? buf - 1 : 2;                           // (s + 2) % 3
            const int nx = buf == 2 ? 0 : buf + 1;                           // (s + 1) % 3
            const bool more = s + 2 < nsteps, ahead = s + 1 < nsteps;
            if (more && !late) dma_stage(nb);
            const char* st = smem + buf * SLOTB;
            const char* sn = smem + nx * SLOTB;
            [&]<int... SL>(std::integer_sequence<int, SL...>) {
                (([&] {
                     constexpr int T3 = SL / 3, Q = SL % 3, SET = (T3 + P) & 1;
                     using NS = std::integral_constant<int, SET ^ 1>;
                     if constexpr (Q == 0) {
                         tr_wait();                                          // this third's fragments are there
                         if constexpr (PF == 1) {
                             if constexpr (T3 < 2) rd_b(NS{}, st, T3 + 1);
                             else if (ahead) { rd_a(std::integral_constant<int, P ^ 1>{}, sn); rd_b(NS{}, sn, 0); }
                         }
                     }
                     if constexpr (PF == 1) {
#pragma unroll
                         for (int i = 0; i < TM; ++i)
#pragma unroll
                             for (int j = 0; j < 2; ++j)
                                 acc[i][2 * T3 + j] =
                                     __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[P][PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T3 + j], 0, 0, 0);
                     } else {
                         // PF == 2: the reads of the next third (4 pairs) / of the next stage's first third (10 pairs) go out one pair
                         // behind each of this third's first MFMAs, in the shadow of the 16-cycle MFMA, instead of in a block in front.
                         // (In the last stage the look-ahead reads whatever the next ring slot holds; nobody uses it.)
                         [&]<int... U>(std::integer_sequence<int, U...>) {
                             (([&] {
                                  constexpr int i = U / 2, j = U % 2, K = Q * 6 + U;
                                  acc[i][2 * T3 + j] =
                                      __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[P][PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T3 + j], 0, 0, 0);
                                  if constexpr (T3 < 2) {
                                      if constexpr (K < 4) fb[SET ^ 1][K / 2][K % 2] = tr_pair(st + (K / 2) * XPLANEB + fb_off[2 * (T3 + 1) + K % 2]);
                                  } else {
                                      if constexpr (K < 6) fa[P ^ 1][K / 3][K % 3] = tr_pair(sn + (K / 3) * PLANEB + fa_off[K % 3]);
                                      else if constexpr (K < 10) fb[SET ^ 1][(K - 6) / 2][(K - 6) % 2] = tr_pair(sn + ((K - 6) / 2) * XPLANEB + fb_off[(K - 6) % 2]);
                                  }
                                  __builtin_amdgcn_sched_barrier(0);
                              }()),
                              ...);
                         }
                         (std::make_integer_sequence<int, 6>{});
                     }
                     __builtin_amdgcn_sched_barrier(0);
                     if constexpr (SL == LSLOT) {
                         if (more && late) dma_stage(nb);
                     }
                     __builtin_amdgcn_sched_barrier(0);
                 }()),
                 ...);
            }
            (std::make_integer_sequence<int, 9>{});
        };
        int buf = 0;
        for (int s = 0; s < nsteps; s += 2) {
            stage(std::integral_constant<int, 0>{}, s, buf);
            buf = buf == 2 ? 0 : buf + 1;
            if (s + 1 < nsteps) {
                stage(std::integral_constant<int, 1>{}, s + 1, buf);
                buf = buf == 2 ? 0 : buf + 1;
            }
        }
        tr_wait();
#ifdef RDO_DIAG
        if (tid == 0 && lin < 256) {
            g_wg_stamps[lin * 4 + 0] = clock64() - st_c0; g_wg_stamps[lin * 4 + 1] = wall_clock64() - st_r0;
            g_wg_stamps[lin * 4 + 2] = (unsigned long long)nsteps; g_wg_stamps[lin * 4 + 3] = 0;
        }
#endif
    } else {
    f16x8 fa[NP][TM], fb[2][NP][2];
    auto read_b = [&](auto setc, const char* st, int third) {
        constexpr int S = decltype(setc)::value;
        if (WG_ABL(8)) return;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[S][p][j] = tr_pair(st + p * XPLANEB + fb_off[2 * third + j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
#ifdef RDO_DIAG
    const unsigned long long st_c0 = clock64(), st_r0 = wall_clock64();
    if (WG_ABL(8)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][p][j] = fb[1][p][j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#endif
    int buf = 0;
#ifdef RDO_DIAG
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const bool phase_on = WG_ABL(64);
#define WG_T(v) unsigned long long v = phase_on ? clock64() : 0ull
#else
#define WG_T(v)
#endif
    for (int s = 0; s < nsteps; ++s) {
        WG_T(t0);
        if (WG_ABL(1) || RING3 == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (RING3 - 2)) : "memory");
        WG_T(t1);
        __builtin_amdgcn_s_barrier();                                    // ... everybody's have, and nobody reads buffer (s - 1) % RING3 any more
        WG_T(t2);
        const int nb = buf >= 1 ? buf - 1 : RING3 - 1;                   // (s + RING3 - 1) % RING3
        const bool more = RING3 > 2 || s + 1 < nsteps;                   // (a counted wait needs the same DMA count in every stage)
        if (more && !late) dma_stage(nb);
        WG_T(t3);
#ifdef RDO_DIAG
        if (phase_on) { ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; }
#endif
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const char* st = smem + buf * SLOTB + u * STAGE3B;
            if (!WG_ABL(8)) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[p][i] = tr_pair(st + p * PLANEB + fa_off[i]);
            }
            read_b(S0{}, st, 0);
#ifdef RDO_DIAG
            if (phase_on) {                                              // how long until the first fragments are there
                const unsigned long long ta = clock64();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ph[3] += clock64() - ta;
            }
            const unsigned long long tb = phase_on ? clock64() : 0ull;
#endif
            [&]<int... SL>(std::integer_sequence<int, SL...>) {
                (([&] {
                     constexpr int T3 = SL / 3, Q = SL % 3, SET = T3 & 1;
                     if constexpr (Q == 0) {
                         tr_wait();                                      // this third's fragments (and, in slot 0, the A fragments) are there
                         if constexpr (T3 < 2) {
                             if constexpr (SET == 0) read_b(S1{}, st, T3 + 1);
                             else read_b(S0{}, st, T3 + 1);
                         }
                     }
                     if (!WG_ABL(4)) {
#pragma unroll
                         for (int i = 0; i < TM; ++i)
#pragma unroll
                             for (int j = 0; j < 2; ++j)
                                 acc[i][2 * T3 + j] =
                                     __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T3 + j], 0, 0, 0);
                     }
                     __builtin_amdgcn_sched_barrier(0);
                     if constexpr (SL == LSLOT) {
                         if (u == SUB - 1 && more && late) dma_stage(nb);
                     }
                     __builtin_amdgcn_sched_barrier(0);
                 }()),
                 ...);
            }
            (std::make_integer_sequence<int, 9>{});
#ifdef RDO_DIAG
            if (phase_on) ph[4] += clock64() - tb;
#endif
        }
        buf = buf + 1 == RING3 ? 0 : buf + 1;
    }
#ifdef RDO_DIAG
    if (tid == 0 && lin < 256) {       // shader-clock cycles and 100 MHz wall ticks of the K loop, stage count in the low bits of slot 1
        g_wg_stamps[lin * 4 + 0] = clock64() - st_c0; g_wg_stamps[lin * 4 + 1] = wall_clock64() - st_r0;
        g_wg_stamps[lin * 4 + 2] = (unsigned long long)(nsteps * SUB); g_wg_stamps[lin * 4 + 3] = 0;
    }
    if (phase_on && lane == 0 && lin < 256)
        for (int k = 0; k < 6; ++k) g_wg_phase[(lin * 8 + wave) * 8 + k] = ph[k];
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the clamped tail DMAs still target this workgroup's LDS

    const long wsize = (long)a.Cout * 9 * a.Cin;
    float* slab = a.slabs + (long)chunk * wsize;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int jt = 6 * (wave & 1) + j;
        const int tap = kh * 3 + (jt >> 2);
        const int ci = ci0 + (jt & 3) * 16 + l16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wco0 + i * 16 + 4 * lc + r;
                if (co < a.Cout) slab[((long)co * 9 + tap) * a.Cin + ci] = acc[i][j][r] * a.inv_scale;
            }
    }
}

}  // namespace

extern "C" int rdo_conv2d_wgrad_uses_bf16x6(const rdo_conv_desc* d);
extern "C" int rdo_conv2d_wgrad_nsplit(const rdo_conv_desc* d);

extern "C" int rdo_conv2d_wgrad_h2_supported(const rdo_conv_desc* d) {
    if (!d || d->square_input || d->Cin % 16 != 0 || d->Cout % 16 != 0) return 0;
    // (the row kernel's buffer descriptors span one plane plus a 32-byte lead-in: 2 x elements + 32 bytes must fit 32 bits)
    if ((double)d->B * d->H * d->W * d->Cin >= 2147483648.0 - 64.0 || (double)d->B * d->Ho * d->Wo * d->Cout >= 2147483648.0 - 64.0) return 0;
    return rdo_conv2d_wgrad_uses_bf16x6(d);
}

extern "C" int rdo_conv2d_wgrad_h2_layer_supported(const rdo_conv_desc* d) {
    if (!d || d->square_input || d->Cin % 16 != 0 || d->Cout % 16 != 0) return 0;
    if ((double)d->B * d->H * d->W * d->Cin >= 2147483648.0 - 64.0 || (double)d->B * d->Ho * d->Wo * d->Cout >= 2147483648.0 - 64.0) return 0;
    if (rdo_conv2d_wgrad_h2_supported(d)) return 1;
    return d->KH * d->KW > 1 && d->Cin >= 64 && d->Cout >= 64 && (long)d->B * d->Ho * d->Wo >= 8192;
}

extern "C" int rdo_conv2d_wgrad_h2(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* dy_planes, float dy_scale,
                                   float* slabs, int nsplit, void* stream) {
    RDO_REQUIRE(d && x_planes && dy_planes && slabs && nsplit >= 1, "rdo_conv2d_wgrad_h2: bad argument");
    RDO_REQUIRE(rdo_conv2d_wgrad_h2_layer_supported(d), "rdo_conv2d_wgrad_h2: shape not on the split-precision plane path (rdo_conv2d_wgrad_h2_supported / _layer_supported)");
    constexpr int T = 192;
    WgPArgs a{};
    a.xp = reinterpret_cast<const u16*>(x_planes);
    a.yp = reinterpret_cast<const u16*>(dy_planes);
    a.slabs = slabs;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.Min = d->B * d->H * d->W;
    a.xplane = (long)a.Min * a.Cin;
    a.yplane = (long)a.M * a.Cout;
    a.nsplit = nsplit;
    RDO_REQUIRE(x_scale > 0.f && dy_scale > 0.f, "rdo_conv2d_wgrad_h2: scales must be positive powers of two");
    a.inv_scale = 1.f / (x_scale * dy_scale);
#ifdef RDO_DIAG
    a.ablate = rdo::tuning(rdo::T_X6P_ABLATE);
#endif
    // same chunking as rdo_conv2d_wgrad: whole 32-pixel steps per chunk
    long mchunk = rdo::ceil_div(rdo::ceil_div((long)a.M, nsplit), 32) * 32;
    a.mchunk = (int)mchunk;
    RDO_REQUIRE((long)nsplit * mchunk >= a.M, "rdo_conv2d_wgrad_h2: nsplit too small");
    a.tiles_co = (int)rdo::ceil_div(a.Cout, T);
    a.tiles_ci = (int)rdo::ceil_div(a.Cin, T);
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 4.0 * (a.xplane + a.yplane) + 4.0 * nsplit * (double)a.Cout * a.KH * a.KW * a.Cin;
    // 3 x 3 / stride 1 / pad 1 over whole 32-pixel row segments, Cin in blocks of 64: the three kw taps share one input image
    // (shapes only the layer predicate accepts -- narrow channel counts -- stay on the general kernel, whose loader and epilogue mask the tile)
    if (rdo_conv2d_wgrad_h2_supported(d) && rdo::tuning(rdo::T_WGRAD_P3_ROW) && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Wo % 32 == 0 && a.Cin % 64 == 0 &&
        a.Wo == a.W && a.Ho == a.H && (long)(a.M / 32) >= nsplit) {
        WgPArgs b = a;
        b.tiles_ci = a.Cin / 64;
        // tuning key "wgrad_sub" -> <SUB, RING3, LSLOT, PF> (template comment above).  Release builds carry the shipped stage shape and the
        // four it is tested bit-identical against (tests/test_gpu_h2.py):  13 = <1, 3, 2, 2> shipped (default of the key);  1 = <1, 2, 2, 0>
        // the round-3 loop (also what any unknown value selects);  2 = <2, 2, 2, 0> two segments per barrier;  3 = <1, 3, 8, 0> ring of three,
        // waves 4-7 issue at the end of the stage;  9 = <1, 3, 2, 1> look-ahead reads in one block.  `make DIAG=1` adds the sweep points of
        // round 4 (4-8, 10-12, 14, 15: other issue slots / ring depths; DESIGN 3 has their timings).
        const int variant = rdo::tuning(rdo::T_WGRAD_SUB);
        return rdo::dispatch(
            [b, variant](hipStream_t s) {
                auto go = [&](auto subc, auto ringc, auto slotc, auto pfc) -> int {
                    constexpr int SUB = decltype(subc)::value, RING = decltype(ringc)::value, LS = decltype(slotc)::value, PF = decltype(pfc)::value;
                    constexpr size_t lds = (size_t)RING * SUB * STAGE3B;
                    static_assert(lds <= 160 * 1024, "row weight-gradient LDS");
                    static rdo::PerDevice attr;
                    if (!attr.done()) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2r_kernel<SUB, RING, LS, PF>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad_h2r) failed");
                        attr.mark();
                    }
                    dim3 grid((unsigned)b.nsplit, (unsigned)(3 * b.tiles_co * b.tiles_ci));
                    hipLaunchKernelGGL((conv_wgrad_h2r_kernel<SUB, RING, LS, PF>), grid, dim3(512), lds, s, b);
                    return rdo::check_launch("conv_wgrad_h2r");
                };
                using rdo_i1 = std::integral_constant<int, 1>; using rdo_i2 = std::integral_constant<int, 2>; using rdo_i3 = std::integral_constant<int, 3>;
                using rdo_i5 [[maybe_unused]] = std::integral_constant<int, 5>; using rdo_i8 = std::integral_constant<int, 8>;
                using rdo_i0 = std::integral_constant<int, 0>;
                switch (variant) {
                    case 2: return go(rdo_i2{}, rdo_i2{}, rdo_i2{}, rdo_i0{});
                    case 3: return go(rdo_i1{}, rdo_i3{}, rdo_i8{}, rdo_i0{});
                    case 9: return go(rdo_i1{}, rdo_i3{}, rdo_i2{}, rdo_i1{});
                    case 13: return go(rdo_i1{}, rdo_i3{}, rdo_i2{}, rdo_i2{});
#ifdef RDO_DIAG
                    case 4: return go(rdo_i1{}, rdo_i3{}, rdo_i5{}, rdo_i0{});
                    case 5: return go(rdo_i1{}, rdo_i2{}, rdo_i5{}, rdo_i0{});
                    case 6: return go(rdo_i1{}, rdo_i2{}, rdo_i8{}, rdo_i0{});
                    case 7: return go(rdo_i1{}, rdo_i2{}, rdo_i0{}, rdo_i0{});
                    case 8: return go(rdo_i1{}, rdo_i3{}, rdo_i5{}, rdo_i1{});
                    case 10: return go(rdo_i1{}, rdo_i3{}, rdo_i8{}, rdo_i1{});
                    case 11: return go(rdo_i1{}, rdo_i3{}, rdo_i0{}, rdo_i1{});
                    case 12: return go(rdo_i1{}, rdo_i3{}, rdo_i3{}, rdo_i1{});
                    case 14: return go(rdo_i1{}, rdo_i3{}, rdo_i5{}, rdo_i2{});
                    case 15: return go(rdo_i1{}, rdo_i3{}, rdo_i0{}, rdo_i2{});
#endif
                    default: return go(rdo_i1{}, rdo_i2{}, rdo_i2{}, rdo_i0{});
                }
            },
            stream, "conv_wgrad_h2_rows", flops, bytes);
    }
    return rdo::dispatch(
        [a](hipStream_t s) {
            constexpr size_t lds = (size_t)2 * STAGEB;
            static rdo::PerDevice attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad_h2) failed");
                attr.mark();
            }
            dim3 grid((unsigned)a.nsplit, (unsigned)(a.KH * a.KW * a.tiles_co * a.tiles_ci));
            hipLaunchKernelGGL(conv_wgrad_h2_kernel, grid, dim3(512), lds, s, a);
            return rdo::check_launch("conv_wgrad_h2");
        },
        stream, "conv_wgrad_h2_192x192", flops, bytes);
}

// ---- fp32 token-matrix weight gradient (linear_wgrad_h2_kernel): called by rdo_conv2d_wgrad (conv_wgrad.hip) in front of the split-bf16 kernel
static long lw_min_tokens() {
    static const long v = [] { const char* e = getenv("RDO_LIN_WGRAD_H2_MIN"); return e ? atol(e) : 4096L; }();
    return v;
}
bool rdo_linear_wgrad_h2_ok(const rdo_conv_desc* d, const float* x, const float* dy) {
    static const int on = [] { const char* e = getenv("RDO_LIN_WGRAD_H2"); return e ? atoi(e) : 1; }();
    if (!on || !d) return false;
    const long M = (long)d->B * d->Ho * d->Wo;
    // channel counts in blocks of 192, or 96 (a half tile, masked)
    const bool cin_ok = d->Cin % 192 == 0 || d->Cin == 96, cout_ok = d->Cout % 192 == 0 || d->Cout == 96;
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && cin_ok && cout_ok && M % 32 == 0 && M >= lw_min_tokens() &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
}

int rdo_launch_linear_wgrad_h2(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, int mchunk, hipStream_t s) {
    LwArgs a{};
    a.x = x; a.dy = dy; a.slabs = slabs;
    a.M = d->B * d->Ho * d->Wo; a.Cin = d->Cin; a.Cout = d->Cout;
    a.mchunk = mchunk; a.nsplit = nsplit;
    a.tiles_co = (a.Cout + 191) / 192; a.tiles_ci = (a.Cin + 191) / 192;
    if ((long)nsplit * mchunk < a.M || mchunk % 32 != 0) return rdo::set_error(RDO_EINVAL, "linear_wgrad_h2: bad chunking (%d x %d over %d tokens)", nsplit, mchunk, a.M);
    constexpr size_t lds = (size_t)2 * STAGEB + 16 * sizeof(float) + 8 * sizeof(int);
    static rdo::PerDevice attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_h2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(linear_wgrad_h2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(linear_wgrad_h2) failed");
        attr.mark();
    }
    dim3 grid((unsigned)nsplit, (unsigned)(a.tiles_co * a.tiles_ci));
    if (d->square_input) hipLaunchKernelGGL((linear_wgrad_h2_kernel<true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((linear_wgrad_h2_kernel<false>), grid, dim3(512), lds, s, a);
    return rdo::check_launch("linear_wgrad_h2");
}

#ifdef RDO_DIAG
// diagnostic build only: per workgroup {K-loop cycles, K-loop 100 MHz ticks, stages, 0} of the last row-kernel launch (tools/h2_stamps.py)
extern "C" int rdo_diag_wgrad_phases(unsigned long long* out, int n) {
    if (!out || n <= 0 || n > 256 * 8 * 8) return RDO_EINVAL;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_phase), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? RDO_OK : RDO_EHIP;
}
extern "C" int rdo_diag_wgrad_stamps(unsigned long long* out, int n) {
    if (!out || n <= 0 || n > 1024) return RDO_EINVAL;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? RDO_OK : RDO_EHIP;
}
#endif
