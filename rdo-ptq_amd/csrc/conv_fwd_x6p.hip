// Forward / dgrad conv of the LARGE problems on "P3" activation tensors: the fp32 activation is already stored as its exact
// three-way bf16 split (x = p0 + p1 + p2, planes [3][pixels][C], written by the producing kernel), so BOTH operands of the bf16x6
// GEMM (see conv_fwd_x6.hip) arrive in LDS by LDS-DMA and the K loop contains no conversion arithmetic at all.  In the fp32-input
// kernels every activation quad was split in the loader -- 44 vector instructions, once per (tap x N tile) use, in front of the
// MFMAs of an in-order wave; here the split is done once per element by the producer's epilogue.
//
// Same tile as conv_fwd_x6v5: 128 x 192 per 256-thread workgroup (wave tile 64 x 96 = 2 x 3 tiles of v_mfma_f32_32x32x16_bf16),
// K stage 16 channels of one tap, channel-slice-outer / taps-inner stage order, fragment reads pipelined one stage ahead, two
// workgroups per CU, XCD-aware tile numbering, optional split over K.  New:
//   * A tile by LDS-DMA: wave w fetches rows 32w..32w+31 of each plane (one 1-KiB piece per plane and stage), chunk swizzle on the
//     per-lane SOURCE address, padding / ragged rows read a 16-byte zero page.
//   * Epilogue through LDS: the accumulators of 64 tile rows are staged as fp32 [64][192], then every thread finishes 8
//     consecutive channels of one pixel: 16-byte loads of bias / aux / residual, 16-byte stores of out / pre, and the three bf16
//     planes of the result (what the next conv, or the weight gradient, consumes) -- one 16-byte store per plane.
#include <utility>

#include "rdo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(64))) unsigned g_zero_page[16];      // zero-initialised: source of masked DMA lanes

struct X6PArgs {
    const u16* xp;        // activation planes [3][B*H*W][Cin] bf16
    long xplane;          // elements per activation plane
    const u16* wp;        // weight planes, fragment order [3][Cin/16][KH][KW][Cout][16]
    long wplane;
    const float* bias;
    const float* aux;
    const float* residual;
    float* out;           // nullable
    float* pre;           // nullable
    u16* outp;            // nullable: result planes [3][M][Cout]
    long oplane;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M;
    int epilogue, add_residual;
    int xcd_mode;
    float* partial;       // split-K: raw accumulators [ksplit][M][Cout]
    int ksplit;
};

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// exact three-way split of 8 consecutive channels -> one 16-byte run per plane
__device__ __forceinline__ void split3_x8(const float (&v)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = v[2 * k], b = v[2 * k + 1];
        const unsigned h = cvt_pk(a, b);
        const float ra = a - lo_f(h), rb = b - hi_f(h);
        const unsigned m = cvt_pk(ra, rb);
        const float sa = ra - lo_f(m), sb = rb - hi_f(m);
        p0[k] = h; p1[k] = m; p2[k] = cvt_pk(sa, sb);
    }
}

struct TileId { int m, n, z; };
__device__ __forceinline__ TileId xcd_tile_id(int mode) {
    if (mode == 0) return TileId{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    TileId t;
    t.n = id % gridDim.y;
    const int rest = id / gridDim.y;
    t.m = rest % gridDim.x;
    t.z = rest / gridDim.x;
    return t;
}

__device__ __forceinline__ int chunk_off16(int row, int c) { return row * 32 + ((c ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ float activate(const X6PArgs& a, float v, float aux) {
    switch (a.epilogue) {
        case RDO_EPI_LRELU: return v > 0.f ? v : 0.01f * v;
        case RDO_EPI_LRELU_BWD: return aux > 0.f ? v : 0.01f * v;
        case RDO_EPI_RELU: return v > 0.f ? v : 0.f;
        case RDO_EPI_RELU_BWD: return aux > 0.f ? v : 0.f;
        case RDO_EPI_GDN: return aux * __frsqrt_rn(v);
        case RDO_EPI_IGDN: return aux * __fsqrt_rn(v);
        default: return v;
    }
}

// finish 8 consecutive channels [n, n+8) of output pixel m from their raw sums `v` (bias not yet added)
__device__ __forceinline__ void finish8(const X6PArgs& a, int m, int n, float (&v)[8]) {
    const long o = (long)m * a.Cout + n;
    if (a.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
    }
    if (a.pre) {
        *reinterpret_cast<f32x4*>(a.pre + o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(a.pre + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (a.epilogue != RDO_EPI_NONE) {
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
        if (a.aux) { x0 = *reinterpret_cast<const f32x4*>(a.aux + o); x1 = *reinterpret_cast<const f32x4*>(a.aux + o + 4); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = activate(a, v[k], x0[k]); v[4 + k] = activate(a, v[4 + k], x1[k]); }
    }
    if (a.add_residual) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.residual + o), r1 = *reinterpret_cast<const f32x4*>(a.residual + o + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += r0[k]; v[4 + k] += r1[k]; }
    }
    if (a.out) {
        *reinterpret_cast<f32x4*>(a.out + o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(a.out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (a.outp) {
        u32x4 p0, p1, p2;
        split3_x8(v, p0, p1, p2);
        *reinterpret_cast<u32x4*>(a.outp + o) = p0;
        *reinterpret_cast<u32x4*>(a.outp + a.oplane + o) = p1;
        *reinterpret_cast<u32x4*>(a.outp + 2 * a.oplane + o) = p2;
    }
}

__global__ __launch_bounds__(256, 2) void conv_fwd_x6p_kernel(X6PArgs a) {
    constexpr int BM = 128, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;
    constexpr int APLANE = BM * 32, BPLANE = BN * 32;
    constexpr int STAGE = 3 * (APLANE + BPLANE);           // 30 KiB
    constexpr int NB = 5;                                  // weight DMA pieces per wave (18 over 4 waves, two repeated)

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE]; reused by the epilogue as fp32 [64][192]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;

    // ---- A loader: this lane fetches 16-byte chunk (lane & 1) of tile row 32 * wave + (lane >> 1), for each of the three planes
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

    // ---- B loader (as conv_fwd_x6v5): piece k = wave + 4 j of the 18 KiB [plane][row][chunk] image
    int dma_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
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
    // DMA cursor (wave-uniform): (channel slice, kh, kw) of the next stage to fetch; stages run channel slice outer, taps inner
    int ccs = sbeg / taps;
    int ctap = sbeg - ccs * taps;
    int ckh = ctap / a.KW, ckw = ctap - ckh * a.KW;
    int cstage = 0;                                          // stage index (relative to sbeg) the cursor points at
    auto dma_stage_a = [&](int buf) {                        // A image of the cursor's stage -> LDS buffer `buf`
        const int hi = hi0 + ckh, wi = wi0 + ckw;
        const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const long off = ((long)(apix0 + hi * a.W + wi)) * a.Cin + ccs * KS + achunk * 8;
        const u16* src = ok ? a.xp + off : zero;
        const long pstep = ok ? a.xplane : 0;
        char* dst = smem + buf * STAGE + wave * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(src + pstep), (lds_void*)(dst + APLANE), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(src + 2 * pstep), (lds_void*)(dst + 2 * APLANE), 16, 0, 0);
    };
    auto dma_b = [&](int j, int buf) {                       // weight piece j of the cursor's stage
        const int k = wave + 4 * j < 18 ? wave + 4 * j : wave + 4 * j - 2;
        char* dst = smem + buf * STAGE + 3 * APLANE + k * 1024;
        __builtin_amdgcn_global_load_lds((glb_void*)(a.wp + dma_src[j] + (long)(sbeg + cstage) * a.Cout * 16), (lds_void*)dst, 16, 0, 0);
    };
    auto cursor_next = [&]() {                               // advance to the next stage, clamped at the last one
        if (cstage + 1 >= nsteps) return;
        ++cstage;
        ++ckw;
        if (ckw == a.KW) { ckw = 0; ++ckh; }
        if (ckh == a.KH) { ckh = 0; ++ccs; }
    };

    int fa_off[3][TM], fb_off[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_off[p][i] = p * APLANE + chunk_off16(wm0 + i * 32 + li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = 3 * APLANE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    }
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

    // prologue: images of stages 0 and 1 in LDS, fragments of stage 0 in registers; the cursor then points at stage 2
    dma_stage_a(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_b(j, 0);
    cursor_next();
    dma_stage_a(1);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma_b(j, 1);
    cursor_next();
    __syncthreads();
    read_a(fa0[0], 0, 0); read_b(fb0[0], 0, 0);
    read_a(fa12[0], 1, 0); read_b(fb12[0], 1, 0);
    read_a(fa12[1], 2, 0); read_b(fb12[1], 2, 0);
    __syncthreads();      // every wave holds stage 0's fragments before buffer 0 is overwritten with stage 2

    // stage t: MFMAs of stage t from registers | fragments of stage t+1 from buffer (t+1)&1 | DMA of stage t+2 into buffer t&1
    auto stage = [&](auto parc) {
        constexpr int PAR = decltype(parc)::value, NXT = PAR ^ 1;
        read_a(fa0[NXT], 0, NXT); read_b(fb0[NXT], 0, NXT);
        // slot 0: (A2, B0)
        mma(fa12[1], fb0[PAR]);
        read_a(fa12[1], 2, NXT);
        dma_stage_a(PAR);
        dma_b(0, PAR);
        __builtin_amdgcn_sched_barrier(0);
        // slot 1: (A0, B2)
        mma(fa0[PAR], fb12[1]);
        read_b(fb12[1], 2, NXT);
        dma_b(1, PAR);
        __builtin_amdgcn_sched_barrier(0);
        // slot 2: (A1, B0)
        mma(fa12[0], fb0[PAR]);
        dma_b(2, PAR);
        __builtin_amdgcn_sched_barrier(0);
        // slot 3: (A1, B1)
        mma(fa12[0], fb12[0]);
        read_a(fa12[0], 1, NXT);
        dma_b(3, PAR);
        __builtin_amdgcn_sched_barrier(0);
        // slot 4: (A0, B1)
        mma(fa0[PAR], fb12[0]);
        read_b(fb12[0], 1, NXT);
        dma_b(4, PAR);
        __builtin_amdgcn_sched_barrier(0);
        // slot 5: (A0, B0)
        mma(fa0[PAR], fb0[PAR]);
        cursor_next();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    int t = 0;
    for (; t + 1 < nsteps; t += 2) {
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
    }
    if (t < nsteps) stage(std::integral_constant<int, 0>{});

    // ---- epilogue through LDS: two passes of 64 tile rows (pass p: MFMA row tile p of every wave)
    float* const stg = reinterpret_cast<float*>(smem);       // [64][192] fp32 = 48 KiB
    const long part_base = (long)tile.z * a.M * a.Cout;
#pragma unroll
    for (int pass = 0; pass < TM; ++pass) {
        if (pass) __syncthreads();                           // the previous pass has been read out
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stg[((wave >> 1) * 32 + rr) * BN + wn0 + j * 32 + li] = acc[pass][j][r];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int q = tid + 256 * k;                     // 64 rows x 24 chunks of 8 channels
            const int row = q / 24, ch = q - row * 24;
            const int m = m0 + (row >> 5) * 64 + pass * 32 + (row & 31);
            const int n = n0 + ch * 8;
            if (m >= a.M || n >= a.Cout) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * BN + ch * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * BN + ch * 8 + 4);
            if (a.partial) {
                float* dst = a.partial + part_base + (long)m * a.Cout + n;
                *reinterpret_cast<f32x4*>(dst) = v0;
                *reinterpret_cast<f32x4*>(dst + 4) = v1;
            } else {
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                finish8(a, m, n, v);
            }
        }
    }
}

// split-K second pass: sum the partial accumulators, then the same per-8-channel finish (planes included)
__global__ __launch_bounds__(256) void x6p_splitk_epilogue_kernel(X6PArgs a) {
    const int cpr = a.Cout / 8;                              // chunks per row
    const long total = (long)a.M * cpr;
    const long slab = (long)a.M * a.Cout;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int m = (int)(q / cpr), n = (int)(q - (long)m * cpr) * 8;
        const float* src = a.partial + (long)m * a.Cout + n;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        for (int z = 0; z < a.ksplit; ++z) {
            s0 += *reinterpret_cast<const f32x4*>(src + z * slab);
            s1 += *reinterpret_cast<const f32x4*>(src + z * slab + 4);
        }
        float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        finish8(a, m, n, v);
    }
}

// fp32 NHWC tensor -> its three bf16 planes (producer of last resort for tensors that no fused kernel writes as planes)
__global__ __launch_bounds__(256) void split_p3_kernel(const float* x, long n8, u16* planes, long pstride) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n8; q += (long)gridDim.x * blockDim.x) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + q * 8), a1 = *reinterpret_cast<const f32x4*>(x + q * 8 + 4);
        const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        u32x4 p0, p1, p2;
        split3_x8(v, p0, p1, p2);
        *reinterpret_cast<u32x4*>(planes + q * 8) = p0;
        *reinterpret_cast<u32x4*>(planes + pstride + q * 8) = p1;
        *reinterpret_cast<u32x4*>(planes + 2 * pstride + q * 8) = p2;
    }
}

}  // namespace

extern "C" int rdo_conv2d_fwd_bf16x6_ksplit(const rdo_conv_desc* d);
extern "C" int rdo_conv2d_fwd_uses_bf16x6(const rdo_conv_desc* d);

extern "C" int rdo_split_p3(const float* x, int64_t n, void* planes, void* stream) {
    RDO_REQUIRE(x && planes && n > 0 && n % 8 == 0, "rdo_split_p3: n (%ld) must be a positive multiple of 8", (long)n);
    u16* p = reinterpret_cast<u16*>(planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(n / 8, 256);
            hipLaunchKernelGGL(split_p3_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, x, (long)(n / 8), p, (long)n);
            return rdo::check_launch("split_p3");
        },
        stream, "split_p3", 0.0, 10.0 * n);
}

extern "C" int rdo_conv2d_fwd_p3_supported(const rdo_conv_desc* d) {
    if (!d || d->square_input || d->Cin % 16 != 0 || d->Cout % 8 != 0) return 0;
    if ((double)d->B * d->H * d->W * d->Cin * 3.0 >= 2147483648.0 || (double)d->Cout * d->KH * d->KW * d->Cin * 3.0 >= 2147483648.0) return 0;
    return rdo_conv2d_fwd_uses_bf16x6(d);
}

extern "C" int rdo_conv2d_fwd_p3(const rdo_conv_desc* d, const void* x_planes, const void* wplanes, const float* bias, const float* aux,
                                 const float* residual, float* out, float* pre, void* out_planes, float* workspace,
                                 int64_t workspace_floats, void* stream) {
    RDO_REQUIRE(d && x_planes && wplanes && (out || out_planes || pre), "rdo_conv2d_fwd_p3: null argument");
    RDO_REQUIRE(rdo_conv2d_fwd_p3_supported(d), "rdo_conv2d_fwd_p3: shape not on the split-bf16 plane path (rdo_conv2d_fwd_p3_supported)");
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd_p3: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    const int epi = d->epilogue;
    RDO_REQUIRE(epi >= RDO_EPI_NONE && epi <= RDO_EPI_RELU_BWD, "rdo_conv2d_fwd_p3: unknown epilogue %d", epi);
    RDO_REQUIRE(epi == RDO_EPI_NONE || epi == RDO_EPI_LRELU || epi == RDO_EPI_RELU || aux != nullptr, "rdo_conv2d_fwd_p3: epilogue %d needs aux", epi);
    RDO_REQUIRE(!d->add_residual || residual != nullptr, "rdo_conv2d_fwd_p3: add_residual without residual");
    X6PArgs a{};
    a.xp = reinterpret_cast<const u16*>(x_planes);
    a.xplane = (long)d->B * d->H * d->W * d->Cin;
    a.wp = reinterpret_cast<const u16*>(wplanes);
    a.wplane = (long)d->Cout * d->KH * d->KW * d->Cin;
    a.bias = bias; a.aux = aux; a.residual = residual; a.out = out; a.pre = pre;
    a.outp = reinterpret_cast<u16*>(out_planes);
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.oplane = (long)a.M * a.Cout;
    a.epilogue = d->epilogue; a.add_residual = d->add_residual;
    a.xcd_mode = rdo::tuning(rdo::T_XCD);
    int ks = rdo_conv2d_fwd_bf16x6_ksplit(d);
    if (ks < 1) ks = 1;
    if (ks > 1 && (!workspace || (long)ks * a.M * a.Cout > workspace_floats)) ks = 1;
    a.ksplit = ks;
    a.partial = ks > 1 ? workspace : nullptr;
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 6.0 * a.xplane + 6.0 * a.wplane +
                         (double)a.M * a.Cout * (4.0 * ((out != nullptr) + (pre != nullptr) + (aux != nullptr) + (residual != nullptr)) +
                                                 6.0 * (out_planes != nullptr));
    return rdo::dispatch(
        [a](hipStream_t s) {
            constexpr size_t lds = (size_t)2 * 3 * (128 + 192) * 32;
            static bool attr = false;
            if (!attr) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_x6p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6p) failed");
                attr = true;
            }
            dim3 grid((unsigned)rdo::ceil_div(a.M, 128), (unsigned)rdo::ceil_div(a.Cout, 192), (unsigned)a.ksplit);
            hipLaunchKernelGGL(conv_fwd_x6p_kernel, grid, dim3(256), lds, s, a);
            if (int rc = rdo::check_launch("conv_fwd_x6p")) return rc;
            if (a.ksplit > 1) {
                long g = rdo::ceil_div((long)a.M * a.Cout / 8, 256);
                hipLaunchKernelGGL(x6p_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
                return rdo::check_launch("x6p_splitk_epilogue");
            }
            return RDO_OK;
        },
        stream, "conv_fwd_x6_p3_128x192", flops, bytes);
}
