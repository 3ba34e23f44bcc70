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
    unsigned x_bytes, dy_bytes;
};

typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two RNE conversions in one v_cvt_pk_bf16_f32; the packed pair is already the LDS image of two consecutive k
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
// exact three-way split of four values (four consecutive pixels of one channel) -> three 8-byte runs
__device__ __forceinline__ void split3_x4(float v0, float v1, float v2, float v3, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    const unsigned h01 = cvt_pk(v0, v1), h23 = cvt_pk(v2, v3);
    const float r0 = v0 - lo_f(h01), r1 = v1 - hi_f(h01), r2 = v2 - lo_f(h23), r3 = v3 - hi_f(h23);
    const unsigned m01 = cvt_pk(r0, r1), m23 = cvt_pk(r2, r3);
    const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m23), s3 = r3 - hi_f(m23);
    const unsigned l01 = cvt_pk(s0, s1), l23 = cvt_pk(s2, s3);
    p0 = __builtin_bit_cast(bf16x4, u32x2{h01, h23});
    p1 = __builtin_bit_cast(bf16x4, u32x2{m01, m23});
    p2 = __builtin_bit_cast(bf16x4, u32x2{l01, l23});
}

// byte offset of 16-byte chunk c (0..3) of `row` inside one plane of a [rows][32 bf16] tile.  Every group of four rows is
// followed by a 16-byte pad: rows r and r+4 then start 4 banks apart, which keeps the fragment ds_read_b128 conflict-free (8
// consecutive rows hit 8 different 16-byte bank slots) AND spreads the loader's 8-byte transpose stores -- lanes of one store
// differ in the channel quad, i.e. in row/4 -- over 16 slots instead of the 4 an XOR swizzle of the chunk index leaves them
// (rocprofv3 SQ_LDS_BANK_CONFLICT was 21 % of the kernel's cycles with the XOR layout).
constexpr int ROWB = 64, GROUPB = 4 * ROWB + 16;     // bytes per row / per padded group of four rows
__device__ __forceinline__ int chunk_off(int row, int c) { return (row >> 2) * GROUPB + (row & 3) * ROWB + (c << 4); }

#ifdef RDO_DIAG   // the four-wave predecessor of conv_wgrad_x6w8_kernel: only in a `make DIAG=1` build (tuning key "wgrad_x6_w8" = 0)
template <int TCO, int TCI, bool SQ>
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

    // XCD-aware numbering: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs.  Give every XCD a
    // contiguous range of logical ids with the TAP fastest: the nine tap-workgroups of one pixel chunk read the same dY and
    // (shifted) X pixels at the same time and should share an L2 instead of fetching them eight times through the fabric.
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
    // Loads go through buffer descriptors: an invalid pixel (halo, ragged tail, channel past the tensor) is requested at an
    // out-of-range offset and the hardware returns zeros -- no per-value select, no validity mask to carry to the store phase
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // Running position of every block: the loads of consecutive reduction steps are 32 output pixels apart, so the pixel
    // decomposition (image, row, column) and the element offsets advance with a few adds / compares instead of the divisions and
    // 64-bit multiplies a from-scratch decode costs every step (they were a third of the kernel's vector instructions).
    int st_step[NBLK], st_m[NBLK], st_off[NBLK], st_wo[NBLK], st_ho[NBLK], st_hi[NBLK], st_wi0[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        st_step[j] = 0;
        const int m = mbeg + 4 * blk_pg[j];
        st_m[j] = m;
        if (!blk_x[j]) {
            st_off[j] = m * a.Cout + co0 + blk_c[j];
            st_wo[j] = st_ho[j] = st_hi[j] = st_wi0[j] = 0;
        } else {
            const int mm = m < a.M ? m : 0;
            const int b = mm / HoWo;
            const int rem = mm - b * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            st_wo[j] = wo; st_ho[j] = ho;
            st_hi[j] = ho * a.stride - a.pad + kh;
            st_wi0[j] = wo * a.stride - a.pad + kw;
            st_off[j] = ((b * a.H + st_hi[j]) * a.W + st_wi0[j]) * a.Cin + ci0 + blk_c[j];
        }
    }
    const int x_step = PK * a.stride * a.Cin;                                  // 32 output pixels further along the row
    const int x_row = (a.stride * a.W - a.Wo * a.stride) * a.Cin;              // wrap to the next output row
    const int x_img = (a.H * a.W - a.Ho * a.stride * a.W) * a.Cin;             // wrap to the next image
    auto advance = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        st_m[j] += PK;
        if (!blk_x[j]) {
            st_off[j] += PK * a.Cout;
            return;
        }
        st_wo[j] += PK;
        st_off[j] += x_step;
        st_wi0[j] += PK * a.stride;
        while (st_wo[j] >= a.Wo) {
            st_wo[j] -= a.Wo;
            st_wi0[j] -= a.Wo * a.stride;
            st_off[j] += x_row;
            st_hi[j] += a.stride;
            if (++st_ho[j] >= a.Ho) {
                st_ho[j] = 0;
                st_hi[j] -= a.Ho * a.stride;
                st_off[j] += x_img;
            }
        }
    };
    // issue the four pixel loads of block j for reduction step s (4 consecutive output pixels never straddle an image row:
    // the host guarantees Wo % 4 == 0 and chunk starts that are multiples of 32); steps are requested in non-decreasing order
    auto load_block = [&](auto jc, int s) {
        constexpr int j = decltype(jc)::value;
        if (s > st_step[j]) {
            advance(jc);
            st_step[j] = s;
        }
        const int m = st_m[j];
        const bool inrange = m < mend;
        if (!blk_x[j]) {
            const bool okc = inrange && co0 + blk_c[j] < a.Cout;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool ok = okc && m + p < mend;
                rq[j][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsrc_y, ok ? (unsigned)(st_off[j] + p * a.Cout) * 4u : OOB, 0, 0));
            }
        } else {
            const bool okr = inrange && ci0 + blk_c[j] < a.Cin && (unsigned)st_hi[j] < (unsigned)a.H;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int wi = st_wi0[j] + p * a.stride;
                const bool ok = okr && m + p < mend && (unsigned)wi < (unsigned)a.W;
                rq[j][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsrc_x, ok ? (unsigned)(st_off[j] + p * a.stride * a.Cin) * 4u : OOB, 0, 0));
            }
        }
    };
    // split + transpose-store channel cc of block j: three 8-byte runs (4 consecutive pixels) into row (4cq + cc)
    auto store_part = [&](auto jc, auto ccc, int buf) {
        constexpr int j = decltype(jc)::value;
        constexpr int cc = decltype(ccc)::value;
        bf16x4 p0, p1, p2;
        float v0 = rq[j][0][cc], v1 = rq[j][1][cc], v2 = rq[j][2][cc], v3 = rq[j][3][cc];
        if constexpr (SQ) {
            const bool q = blk_x[j];
            v0 = q ? v0 * v0 : v0; v1 = q ? v1 * v1 : v1; v2 = q ? v2 * v2 : v2; v3 = q ? v3 * v3 : v3;
        }
        split3_x4(v0, v1, v2, v3, p0, p1, p2);
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

    // One workgroup (one wave per SIMD) per CU: nothing else hides this wave's LDS latency, so the fragment reads are pipelined
    // half a step ahead into a second register set, and a step needs a single barrier, in its middle:
    //   first half : MFMAs of k-group 0 from set A | all 12 transpose-stores of step s+1 into the other buffer | set B <- (s, k-group 1)
    //   barrier    : step s+1's image is complete, and nobody reads this step's buffer any more (its fragments are in registers)
    //   second half: MFMAs of k-group 1 from set B | global loads of step s+2 | set A <- (s+1, k-group 0) from the other buffer
    bf16x8 fa[2][3][TM], fb[2][3][TN];
    auto read_frag = [&](auto setc, const char* st, int kb, int idx) {       // idx 0..17: plane-major A then B fragments
        constexpr int S = decltype(setc)::value;
        const int p = idx / (TM + TN), r = idx - p * (TM + TN);
        if (r < TM)
            fa[S][p][r] = *reinterpret_cast<const bf16x8*>(st + p * YPLANE + chunk_off(wco0 + r * 32 + li, 2 * kb + lh));
        else
            fb[S][p][r - TM] = *reinterpret_cast<const bf16x8*>(st + 3 * YPLANE + p * XPLANE +
                                                                 chunk_off(wci0 + (r - TM) * 32 + li, 2 * kb + lh));
    };
    static_assert(3 * (TM + TN) == 18 && NBLK == 3, "slot schedule below is written for the 192 x 192 tile");
    if (nsteps > 0) {
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (read_frag(std::integral_constant<int, 0>{}, smem, 0, I), ...);
        }(std::make_integer_sequence<int, 18>{});
    }

    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
        const char* st = smem + buf * STAGE;
        const char* stn = smem + (buf ^ 1) * STAGE;
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[SL]][i], fb[0][PB[SL]][j], acc[i][j], 0, 0, 0);
                 store_part(std::integral_constant<int, (2 * SL) / 4>{}, std::integral_constant<int, (2 * SL) % 4>{}, buf ^ 1);
                 store_part(std::integral_constant<int, (2 * SL + 1) / 4>{}, std::integral_constant<int, (2 * SL + 1) % 4>{}, buf ^ 1);
                 read_frag(std::integral_constant<int, 1>{}, st, 1, 3 * SL);
                 read_frag(std::integral_constant<int, 1>{}, st, 1, 3 * SL + 1);
                 read_frag(std::integral_constant<int, 1>{}, st, 1, 3 * SL + 2);
                 // one wave per SIMD: the split arithmetic of the two store parts must sit BETWEEN the nine MFMAs (each leaves
                 // ~24 issue cycles), not in front of them; the LDS writes / reads follow
#pragma unroll
                 for (int g = 0; g < TM * TN; ++g) {
                     __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                     __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // 6 VALU
                 }
                 __builtin_amdgcn_sched_group_barrier(0x300, 9, 0);         // 6 ds_write_b64 + 3 ds_read_b128
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 6>{});
        __syncthreads();
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < TN; ++j)
                         acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[SL]][i], fb[1][PB[SL]][j], acc[i][j], 0, 0, 0);
                 if constexpr (SL < NBLK) load_block(std::integral_constant<int, SL>{}, s2);
                 read_frag(std::integral_constant<int, 0>{}, stn, 0, 3 * SL);
                 read_frag(std::integral_constant<int, 0>{}, stn, 0, 3 * SL + 1);
                 read_frag(std::integral_constant<int, 0>{}, stn, 0, 3 * SL + 2);
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 6>{});
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

#endif  // RDO_DIAG

// ---- eight-wave variant: the same 192 x 192 tile and LDS images, but 512 threads ---------------------------------------------------
// One wave per SIMD cannot keep the matrix pipe busy (tools/mfma_probe.hip: a wgrad-shaped loop reaches 0.35 of the nominal rate
// with one workgroup of four waves per CU and 0.56-0.59 with two waves per SIMD), and the 144 accumulators + 144 KiB of LDS of the
// four-wave kernel allow nothing else.  Here each of 8 waves owns 48 x 96 of the tile as 3 x 6 tiles of v_mfma_f32_16x16x32_bf16
// (72 accumulators; one MFMA spans the whole 32-pixel step), so the CU runs two waves per SIMD from ONE workgroup with the same
// 144 KiB double buffer.  The 768 loader blocks (4 pixels x 4 channels) of a step go 2 : 1 to waves 0-3 : 4-7 -- a SIMD hosts one
// wave of each kind (waves are dealt to SIMDs cyclically), so the loader work is balanced per SIMD.
typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <bool SQ>
__global__ __launch_bounds__(512, 2) void conv_wgrad_x6w8_kernel(WgX6Args a) {
    constexpr int TCO = 192, TCI = 192, NT = 512;
    constexpr int TM = 3, TN = 6;                                             // 16 x 16 tiles per wave: 48 (co) x 96 (ci)
    constexpr int YPLANE = TCO / 4 * GROUPB, XPLANE = TCI / 4 * GROUPB;
    constexpr int STAGE = 3 * (YPLANE + XPLANE);
    constexpr int QY = TCO / 4, QX = TCI / 4;
    constexpr int NBLK = 2;                                                   // block 1 only on waves 0-3
    static_assert((QY + QX) * 8 == NT + NT / 2, "768 blocks: one per thread plus one more for the first 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lc = lane >> 4;
    const int wco0 = (wave >> 1) * 48, wci0 = (wave & 1) * 96;
    const bool has2 = wave < 4;

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
    const int co0 = tco * TCO, ci0 = tci * TCI;

    const int mbeg = chunk * a.mchunk;
    const int mend = min(a.M, mbeg + a.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + PK - 1) / PK : 0;
    const int HoWo = a.Ho * a.Wo;

    // blocks: e = tid + 512 j over [dY quads | X quads] x 8 pixel groups of 4 (j = 1: blocks 512 .. 767, waves 0-3 only)
    bool blk_x[NBLK];
    int blk_pg[NBLK], blk_c[NBLK], blk_lds[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int e = tid + NT * j;
        const int ee = e < (QY + QX) * 8 ? e : 0;
        blk_x[j] = ee >= QY * 8;
        const int r = blk_x[j] ? ee - QY * 8 : ee;
        const int nq = blk_x[j] ? QX : QY;
        blk_pg[j] = r / nq;
        const int cq = r - blk_pg[j] * nq;
        blk_c[j] = 4 * cq;
        blk_lds[j] = (blk_x[j] ? 3 * YPLANE : 0) + chunk_off(4 * cq, blk_pg[j] >> 1) + (blk_pg[j] & 1) * 8;
    }

    f32x4acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    f32x4 rq[NBLK][4];
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const auto rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    int st_step[NBLK], st_m[NBLK], st_off[NBLK], st_wo[NBLK], st_ho[NBLK], st_hi[NBLK], st_wi0[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        st_step[j] = 0;
        const int m = mbeg + 4 * blk_pg[j];
        st_m[j] = m;
        if (!blk_x[j]) {
            st_off[j] = m * a.Cout + co0 + blk_c[j];
            st_wo[j] = st_ho[j] = st_hi[j] = st_wi0[j] = 0;
        } else {
            const int mm = m < a.M ? m : 0;
            const int b = mm / HoWo;
            const int rem = mm - b * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            st_wo[j] = wo; st_ho[j] = ho;
            st_hi[j] = ho * a.stride - a.pad + kh;
            st_wi0[j] = wo * a.stride - a.pad + kw;
            st_off[j] = ((b * a.H + st_hi[j]) * a.W + st_wi0[j]) * a.Cin + ci0 + blk_c[j];
        }
    }
    const int x_step = PK * a.stride * a.Cin;
    const int x_row = (a.stride * a.W - a.Wo * a.stride) * a.Cin;
    const int x_img = (a.H * a.W - a.Ho * a.stride * a.W) * a.Cin;
    auto advance = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        st_m[j] += PK;
        if (!blk_x[j]) {
            st_off[j] += PK * a.Cout;
            return;
        }
        st_wo[j] += PK;
        st_off[j] += x_step;
        st_wi0[j] += PK * a.stride;
        while (st_wo[j] >= a.Wo) {
            st_wo[j] -= a.Wo;
            st_wi0[j] -= a.Wo * a.stride;
            st_off[j] += x_row;
            st_hi[j] += a.stride;
            if (++st_ho[j] >= a.Ho) {
                st_ho[j] = 0;
                st_hi[j] -= a.Ho * a.stride;
                st_off[j] += x_img;
            }
        }
    };
    auto load_block = [&](auto jc, int s) {
        constexpr int j = decltype(jc)::value;
        if (s > st_step[j]) {
            advance(jc);
            st_step[j] = s;
        }
        const int m = st_m[j];
        const bool inrange = m < mend;
        if (!blk_x[j]) {
            const bool okc = inrange && co0 + blk_c[j] < a.Cout;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool ok = okc && m + p < mend;
                rq[j][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsrc_y, ok ? (unsigned)(st_off[j] + p * a.Cout) * 4u : OOB, 0, 0));
            }
        } else {
            const bool okr = inrange && ci0 + blk_c[j] < a.Cin && (unsigned)st_hi[j] < (unsigned)a.H;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int wi = st_wi0[j] + p * a.stride;
                const bool ok = okr && m + p < mend && (unsigned)wi < (unsigned)a.W;
                rq[j][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rsrc_x, ok ? (unsigned)(st_off[j] + p * a.stride * a.Cin) * 4u : OOB, 0, 0));
            }
        }
    };
    // split + transpose-store channel cc of block j: three 8-byte runs (4 consecutive pixels) into row (4cq + cc)
    auto store_part = [&](auto jc, auto ccc, int buf) {
        constexpr int j = decltype(jc)::value;
        constexpr int cc = decltype(ccc)::value;
        bf16x4 p0, p1, p2;
        float v0 = rq[j][0][cc], v1 = rq[j][1][cc], v2 = rq[j][2][cc], v3 = rq[j][3][cc];
        if constexpr (SQ) {
            const bool q = blk_x[j];
            v0 = q ? v0 * v0 : v0; v1 = q ? v1 * v1 : v1; v2 = q ? v2 * v2 : v2; v3 = q ? v3 * v3 : v3;
        }
        split3_x4(v0, v1, v2, v3, p0, p1, p2);
        char* st = smem + buf * STAGE + blk_lds[j] + cc * 64;
        const int plane = blk_x[j] ? XPLANE : YPLANE;
        *reinterpret_cast<bf16x4*>(st) = p0;
        *reinterpret_cast<bf16x4*>(st + plane) = p1;
        *reinterpret_cast<bf16x4*>(st + 2 * plane) = p2;
    };
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    auto store_block = [&](auto jc, int buf) {
        store_part(jc, std::integral_constant<int, 0>{}, buf);
        store_part(jc, std::integral_constant<int, 1>{}, buf);
        store_part(jc, std::integral_constant<int, 2>{}, buf);
        store_part(jc, std::integral_constant<int, 3>{}, buf);
    };

    if (nsteps > 0) {
        load_block(J0{}, 0);
        if (has2) load_block(J1{}, 0);
        store_block(J0{}, 0);
        if (has2) store_block(J1{}, 0);
        load_block(J0{}, nsteps > 1 ? 1 : 0);
        if (has2) load_block(J1{}, nsteps > 1 ? 1 : 0);
    }
    __syncthreads();

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // fragments: lane (l16, lc) holds row l16 of a 16-row tile, pixels 8 lc .. 8 lc + 7 of the step (16-byte chunk lc of the row)
    int fa_off[TM], fb_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa_off[i] = chunk_off(wco0 + i * 16 + l16, lc);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb_off[j] = 3 * YPLANE + chunk_off(wci0 + j * 16 + l16, lc);

    // Per step: the A fragments (all three planes) once, the B fragments in three thirds of the N range (2 tiles each) through two
    // register sets -- the next third is read while the current one is multiplied; 18 slots of 6 MFMAs.  The loader's work rides in
    // the slots: transpose-stores of step s+1 into the other buffer, then that block's global loads for step s+2.
    bf16x8 fa[3][TM], fb[2][3][2];
    auto read_b = [&](auto setc, const char* st, int third) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[S][p][j] = *reinterpret_cast<const bf16x8*>(st + p * XPLANE + fb_off[2 * third + j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(st + p * YPLANE + fa_off[i]);
        read_b(S0{}, st, 0);
        [&]<int... SL>(std::integer_sequence<int, SL...>) {
            (([&] {
                 constexpr int T = SL / 6, Q = SL % 6, SET = T & 1;
                 if constexpr (Q == 0 && T < 2) {
                     if constexpr (SET == 0) read_b(S1{}, st, T + 1);
                     else read_b(S0{}, st, T + 1);
                 }
#pragma unroll
                 for (int i = 0; i < TM; ++i)
#pragma unroll
                     for (int j = 0; j < 2; ++j)
                         acc[i][2 * T + j] =
                             __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[Q]][i], fb[SET][PB[Q]][j], acc[i][2 * T + j], 0, 0, 0);
                 if constexpr (SL >= 1 && SL < 5) store_part(J0{}, std::integral_constant<int, SL - 1>{}, buf ^ 1);
                 if constexpr (SL == 5) load_block(J0{}, s2);
                 if constexpr (SL >= 7 && SL < 11) { if (has2) store_part(J1{}, std::integral_constant<int, SL - 7>{}, buf ^ 1); }
                 if constexpr (SL == 11) { if (has2) load_block(J1{}, s2); }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }
        (std::make_integer_sequence<int, 18>{});
        __syncthreads();
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
    a.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->Cin * sizeof(float));     // < 2^32: checked by uses_bf16x6
    a.dy_bytes = (unsigned)((size_t)a.M * d->Cout * sizeof(float));
    constexpr size_t lds = (size_t)2 * 3 * (T + T) / 4 * (4 * 64 + 16);
    dim3 grid((unsigned)nsplit, (unsigned)(a.KH * a.KW * a.tiles_co * a.tiles_ci));
    // the eight-wave kernel (two waves per SIMD); its four-wave predecessor ("wgrad_x6_w8" = 0) exists only in a DIAG build
#ifdef RDO_DIAG
    if (!rdo::tuning(rdo::T_WGRAD_X6_W8)) {
        static rdo::PerDevice attr_set;
        if (!attr_set.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_x6_kernel<T, T, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_x6_kernel<T, T, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad_x6, %zu B LDS) failed", lds);
            attr_set.mark();
        }
        if (a.square_input) hipLaunchKernelGGL((conv_wgrad_x6_kernel<T, T, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((conv_wgrad_x6_kernel<T, T, false>), grid, dim3(256), lds, s, a);
        return rdo::check_launch("conv_wgrad_x6");
    }
#endif
    static rdo::PerDevice attr8;
    if (!attr8.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_x6w8_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_x6w8_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_wgrad_x6w8, %zu B LDS) failed", lds);
        attr8.mark();
    }
    if (a.square_input) hipLaunchKernelGGL((conv_wgrad_x6w8_kernel<true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((conv_wgrad_x6w8_kernel<false>), grid, dim3(512), lds, s, a);
    return rdo::check_launch("conv_wgrad_x6w8");
}
