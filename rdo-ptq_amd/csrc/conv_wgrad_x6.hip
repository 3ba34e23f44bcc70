// Weight gradient of the LARGE problems on the bf16 MFMA with exact three-way bf16 splits of both operands ("bf16x6",
// see conv_fwd_x6.hip): dw[co][tap][ci] = sum_m dy[m][co] * x[pix(m,tap)][ci], fp32-level accuracy.
//
// The MFMA wants 8 consecutive reduction indices (pixels) per lane, while NHWC memory has channels contiguous.  The
// loader therefore transposes in registers: a thread owns a 4-pixel x 4-channel block (four 16-byte loads, one per
// pixel), splits the 16 values and writes, per channel and plane, one 8-byte run of 4 consecutive pixels into an LDS tile
// laid out [channel][32 pixels] -- the k-contiguous row layout of the forward kernel, read back with the same
// conflict-free ds_read_b128.  Tile 192 x 192 per tap (3x3 MFMA tiles of 32x32 per wave), 32 pixels per step
// (two 16-deep MFMA steps x 6 products), 144 KiB LDS double buffer, one workgroup per CU.
#include <utility>

#include "rdo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int PK = 32;

struct WgX6Args {
    const float* x;
    const float* dy;
    float* slabs;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, mchunk, nsplit;
    int tiles_co, tiles_ci;
    int square_input;
};

__device__ __forceinline__ u16 bf16_bits(__bf16 h) { return __builtin_bit_cast(u16, h); }
__device__ __forceinline__ void split3(float v, short& a, short& b, short& c) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    a = (short)bf16_bits(h); b = (short)bf16_bits(m); c = (short)bf16_bits(l);
}

// byte offset of 16-byte chunk c (0..3) of `row` inside one plane of a [rows][32 bf16] tile.  Every group of four rows is
// followed by a 16-byte pad: rows r and r+4 then start 4 banks apart, which keeps the fragment ds_read_b128 conflict-free (8
// consecutive rows hit 8 different 16-byte bank slots) AND spreads the loader's 8-byte transpose stores -- lanes of one store
// differ in the channel quad, i.e. in row/4 -- over 16 slots instead of the 4 an XOR swizzle of the chunk index leaves them
// (rocprofv3 SQ_LDS_BANK_CONFLICT was 21 % of the kernel's cycles with the XOR layout).
constexpr int ROWB = 64, GROUPB = 4 * ROWB + 16;     // bytes per row / per padded group of four rows
__device__ __forceinline__ int chunk_off(int row, int c) { return (row >> 2) * GROUPB + (row & 3) * ROWB + (c << 4); }

template <int TCO, int TCI>
__global__ __launch_bounds__(256) void conv_wgrad_x6_kernel(WgX6Args a) {
    constexpr int WCO = TCO / 2, WCI = TCI / 2;
    constexpr int TM = WCO / 32, TN = WCI / 32;
    constexpr int YPLANE = TCO / 4 * GROUPB, XPLANE = TCI / 4 * GROUPB;       // bytes per plane per stage
    constexpr int STAGE = 3 * (YPLANE + XPLANE);
    constexpr int QY = TCO / 4, QX = TCI / 4;                 // channel quads per pixel
    constexpr int NBLK = (QY + QX) * 8 / 256;                 // 4-pixel x 4-channel blocks per thread per step
    static_assert((QY + QX) * 8 % 256 == 0, "blocks must divide evenly over 256 threads");
    static_assert(NBLK * 4 <= 12, "one block part per MFMA group");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wco0 = (wave >> 1) * WCO, wci0 = (wave & 1) * WCI;

    const int chunk = blockIdx.x;
    int t = blockIdx.y;
    const int tci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tco = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = tco * TCO, ci0 = tci * TCI;

    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + PK - 1) / PK : 0;
    const int HoWo = a.Ho * a.Wo;

    // per-thread blocks: e = tid + 256 j over [dY quads | X quads] x 8 pixel groups of 4
    bool blk_x[NBLK];
    int blk_pg[NBLK], blk_c[NBLK], blk_lds[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int e = tid + 256 * j;
        blk_x[j] = e >= QY * 8;
        const int r = blk_x[j] ? e - QY * 8 : e;
        const int nq = blk_x[j] ? QX : QY;
        blk_pg[j] = r / nq;
        const int cq = r - blk_pg[j] * nq;
        blk_c[j] = 4 * cq;                                             // first channel of the quad inside the tile
        blk_lds[j] = (blk_x[j] ? 3 * YPLANE : 0) + chunk_off(4 * cq, blk_pg[j] >> 1) + (blk_pg[j] & 1) * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rq[NBLK][4];
    unsigned okmask = 0;   // bit 4*j + p: pixel p of block j holds real data

    // issue the four pixel loads of block j for reduction step s (4 consecutive output pixels never straddle an image row:
    // the host guarantees Wo % 4 == 0 and chunk starts that are multiples of 32)
    auto load_block = [&](auto jc, int s) {
        constexpr int j = decltype(jc)::value;
        const int m = mbeg + s * PK + 4 * blk_pg[j];
        const bool inrange = m < mend;
        if (!blk_x[j]) {
            const int c = co0 + blk_c[j];
            const bool okc = inrange && c < a.Cout;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool ok = okc && m + p < mend;
                rq[j][p] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? (m + p) * a.Cout + c : 0));
                okmask = ok ? (okmask | (1u << (4 * j + p))) : (okmask & ~(1u << (4 * j + p)));
            }
        } else {
            const int c = ci0 + blk_c[j];
            const int mm = inrange ? m : 0;
            const int b = mm / HoWo;
            const int rem = mm - b * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int hi = ho * a.stride - a.pad + kh, wi0 = wo * a.stride - a.pad + kw;
            const bool okr = inrange && c < a.Cin && (unsigned)hi < (unsigned)a.H;
            const int base = ((b * a.H + hi) * a.W + wi0) * a.Cin + c;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int wi = wi0 + p * a.stride;
                const bool ok = okr && m + p < mend && (unsigned)wi < (unsigned)a.W;
                rq[j][p] = *reinterpret_cast<const f32x4*>(a.x + (ok ? base + p * a.stride * a.Cin : 0));
                okmask = ok ? (okmask | (1u << (4 * j + p))) : (okmask & ~(1u << (4 * j + p)));
            }
        }
    };
    // split + transpose-store channel cc of block j: three 8-byte runs (4 consecutive pixels) into row (4cq + cc)
    auto store_part = [&](auto jc, auto ccc, int buf) {
        constexpr int j = decltype(jc)::value;
        constexpr int cc = decltype(ccc)::value;
        bf16x4 p0, p1, p2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v = ((okmask >> (4 * j + p)) & 1u) ? rq[j][p][cc] : 0.f;
            if (blk_x[j] && a.square_input) v = v * v;
            short h, m, l;
            split3(v, h, m, l);
            p0[p] = h; p1[p] = m; p2[p] = l;
        }
        // row + cc: rows of one aligned group of four are 64 bytes apart
        char* st = smem + buf * STAGE + blk_lds[j] + cc * 64;
        const int plane = blk_x[j] ? XPLANE : YPLANE;
        *reinterpret_cast<bf16x4*>(st) = p0;
        *reinterpret_cast<bf16x4*>(st + plane) = p1;
        *reinterpret_cast<bf16x4*>(st + 2 * plane) = p2;
    };
    auto for_blocks = [&](auto&& f) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
        (std::make_integer_sequence<int, NBLK>{});
    };
    auto store_block = [&](auto jc, int buf) {
        store_part(jc, std::integral_constant<int, 0>{}, buf);
        store_part(jc, std::integral_constant<int, 1>{}, buf);
        store_part(jc, std::integral_constant<int, 2>{}, buf);
        store_part(jc, std::integral_constant<int, 3>{}, buf);
    };

    if (nsteps > 0) {
        for_blocks([&](auto jc) { load_block(jc, 0); });
        for_blocks([&](auto jc) { store_block(jc, 0); });
        for_blocks([&](auto jc) { load_block(jc, nsteps > 1 ? 1 : 0); });
    }
    __syncthreads();

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
        const char* st = smem + buf * STAGE;
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            bf16x8 fa[3][TM], fb[3][TN];
            (([&] {
                 constexpr int kb = SL / 6, pr = SL % 6;
                 if constexpr (pr == 0) {
#pragma unroll
                     for (int p = 0; p < 3; ++p) {
#pragma unroll
                         for (int i = 0; i < TM; ++i)
                             fa[p][i] = *reinterpret_cast<const bf16x8*>(st + p * YPLANE + chunk_off(wco0 + i * 32 + li, 2 * kb + lh));
#pragma unroll
                         for (int j = 0; j < TN; ++j)
                             fb[p][j] = *reinterpret_cast<const bf16x8*>(st + 3 * YPLANE + p * XPLANE +
                                                                          chunk_off(wci0 + j * 32 + li, 2 * kb + lh));
                     }
                 }
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pr]][i], fb[PB[pr]][j], acc[i][j], 0, 0, 0);
                 if constexpr (SL < 4 * NBLK) {
                     constexpr int jb = SL / 4, cc = SL % 4;
                     store_part(std::integral_constant<int, jb>{}, std::integral_constant<int, cc>{}, buf ^ 1);
                     if constexpr (cc == 3) load_block(std::integral_constant<int, jb>{}, s2);
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 12>{});
        __syncthreads();
    }

    const long wsize = (long)a.Cout * a.KH * a.KW * a.Cin;
    float* slab = a.slabs + (long)chunk * wsize;
    const int taps = a.KH * a.KW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wci0 + j * 32 + li;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.Cout) slab[((long)co * taps + tap) * a.Cin + ci] = acc[i][j][r];
            }
    }
}

}  // namespace

// called by rdo_conv2d_wgrad (conv_wgrad.hip) for the shapes rdo_conv2d_wgrad_uses_bf16x6 accepts
int rdo_launch_wgrad_x6(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, int mchunk,
                        hipStream_t s) {
    constexpr int T = 192;
    WgX6Args a;
    a.x = x; a.dy = dy; a.slabs = slabs;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.nsplit = nsplit; a.mchunk = mchunk; a.square_input = d->square_input;
    a.tiles_co = (int)rdo::ceil_div(a.Cout, T);
    a.tiles_ci = (int)rdo::ceil_div(a.Cin, T);
    constexpr size_t lds = (size_t)2 * 3 * (T + T) / 4 * (4 * 64 + 16);
    auto kern = conv_wgrad_x6_kernel<T, T>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad_x6, %zu B LDS) failed", lds);
        attr_set = true;
    }
    dim3 grid((unsigned)nsplit, (unsigned)(a.KH * a.KW * a.tiles_co * a.tiles_ci));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return rdo::check_launch("conv_wgrad_x6");
}
