// Forward / dgrad conv of the LARGE problems on "P3" activation tensors: the fp32 activation is already stored as its exact
// three-way bf16 split (x = p0 + p1 + p2), written by the producing kernel, so BOTH operands of the bf16x6 GEMM (see
// conv_fwd_x6.hip) arrive in LDS by LDS-DMA and the K loop contains no conversion arithmetic at all.  In the fp32-input kernels
// every activation quad was split in the loader -- 44 vector instructions, once per (tap x N tile) use, in front of the MFMAs of an
// in-order wave; here the split is done once per element by the producer's epilogue.
//
// P3 layout ("slice-major planes", rdo_ptq_hip.h): element (pixel m, channel c) of a tensor with M pixels and C channels lives at
//     plane * M * C + ((c >> 4) * M + m) * 16 + (c & 15)            i.e. planes [3][C/16][pixel][16]
// One K stage of the kernel reads the 16 channels of ONE slice for a run of pixels, and consecutive pixels of a slice are consecutive
// 32-byte records: every 1-KiB LDS-DMA piece is a contiguous KiB of memory.  (With pixel-major planes [pixel][C] a stage touched 32
// bytes of every 128-byte line, 4x the L2 -> CU traffic.)
//
// Kernel: 256 x 192 tile on eight waves (4 x 2, wave tile 64 x 96 = 2 x 3 tiles of v_mfma_f32_32x32x16_bf16), ONE workgroup per CU,
// K stage = 16 channels of one tap (channel slice outer, taps inner), THREE LDS stage buffers: the DMA of stage t+2 is issued at the
// top of stage t and has two full stages to land (counted vmcnt, raw s_barrier).  Why: ablating a two-buffer 128 x 192 version of
// this kernel (tuning key "x6p_ablate") on the 4 x 128^2, 192 -> 192, 3x3 conv gave 230 us complete, 132 us with MFMAs + fragment
// reads only and 150 us with the DMA stream only -- one L2 round trip (~1.4 us) per stage, fully exposed because every stage ended
// in vmcnt(0) + barrier.  The bigger tile also halves the weight-tile traffic per MFMA.
//
// Epilogue through LDS: the accumulators of 128 tile rows are staged as fp32, then every thread finishes the 16 channels of one
// slice of one pixel (consecutive lanes on consecutive pixels = the P3 record order): bias / aux / residual, fp32 out / pre, and the
// three bf16 planes of the result, two 16-byte stores per plane.
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
    const u16* xp;        // activation planes (P3 layout of [B*H*W][Cin])
    long xplane;          // elements per activation plane
    const u16* wp;        // weight planes, fragment order [3][Cin/16][KH][KW][Cout][16]
    long wplane;
    const float* bias;
    const float* aux;
    const u16* auxp;      // alternative to aux for the sign-only epilogues (LRELU_BWD / RELU_BWD): P3 planes of the aux tensor (plane 0 is read)
    const float* residual;
    float* out;           // nullable
    float* pre;           // nullable
    u16* outp;            // nullable: result planes (P3 layout of [M][Cout])
    long oplane;
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M;
    int epilogue, add_residual;
    int xcd_mode;
    float* partial;       // split-K: raw accumulators [ksplit][M][Cout]
    int ksplit;
    // unit tail folded into the halo kernel's epilogue (rdo_conv2d_fwd_p3_tail): out = act(conv + bias) + residual, loss against the
    // cached target rows, dL/dpre written as the result planes -- rdo_loss_act_bwd without the pre-activation round trip
    const float* tail_tgt;    // nullptr: no tail
    const int32_t* tail_idx;
    const int32_t* tail_iter;
    const u16* tail_resp;     // residual as P3 planes of [M][Cout] (nullable)
    float* tail_loss;
    long tail_per_image;
    int tail_B, tail_act;
    float tail_coef, tail_inv_npix;
    // diagnostic bit mask (tuning key "x6p_ablate"): 1 no A DMA, 2 no B DMA, 4 no MFMA, 8 no fragment reads -- results are WRONG when
    // non-zero, so the masks exist only in a `make DIAG=1` build (-DRDO_DIAG); the shipped library folds every test away
#ifdef RDO_DIAG
    int ablate;
#else
    static constexpr int ablate = 0;
#endif
};

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// exact three-way split of 8 consecutive channels -> one 16-byte run per plane
__device__ __forceinline__ void split3_x8(const float* v, u32x4& p0, u32x4& p1, u32x4& p2) {
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

// four aux values of channels [n, n+4) of pixel m: the fp32 tensor, or plane 0 of its P3 form (same sign as the value: enough for the
// activation-backward masks)
__device__ __forceinline__ f32x4 aux_quad(const X6PArgs& a, int m, int n, long o) {
    if (a.aux) return *reinterpret_cast<const f32x4*>(a.aux + o);
    if (a.auxp) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 b = *reinterpret_cast<const u32x2*>(a.auxp + ((long)(n >> 4) * a.M + m) * 16 + (n & 15));
        return f32x4{lo_f(b[0]), hi_f(b[0]), lo_f(b[1]), hi_f(b[1])};
    }
    return f32x4{0.f, 0.f, 0.f, 0.f};
}

// the three P3 records of (slice n / 16, pixel m) from 16 finished channel values: 32 bytes per plane
__device__ __forceinline__ void store_slice(const X6PArgs& a, int m, int n, const float (&v)[16]) {
    u16* dst = a.outp + ((long)(n >> 4) * a.M + m) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 p0, p1, p2;
        split3_x8(v + 8 * h, p0, p1, p2);
        *reinterpret_cast<u32x4*>(dst + 8 * h) = p0;
        *reinterpret_cast<u32x4*>(dst + a.oplane + 8 * h) = p1;
        *reinterpret_cast<u32x4*>(dst + 2 * a.oplane + 8 * h) = p2;
    }
}

// finish the 16 channels [n, n+16) (one slice, n % 16 == 0) of output pixel m from their raw sums `v` (bias not yet added)
__device__ __forceinline__ void finish16(const X6PArgs& a, int m, int n, float (&v)[16]) {
    const long o = (long)m * a.Cout + n;
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + n + 4 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * c + k] += b4[k];
        }
    }
    if (a.pre) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(a.pre + o + 4 * c) = f32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
    }
    if (a.epilogue != RDO_EPI_NONE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 x4 = aux_quad(a, m, n + 4 * c, o + 4 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * c + k] = activate(a, v[4 * c + k], x4[k]);
        }
    }
    if (a.add_residual) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(a.residual + o + 4 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * c + k] += r4[k];
        }
    }
    if (a.out) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(a.out + o + 4 * c) = f32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
    }
    if (a.outp) store_slice(a, m, n, v);
}

// Unit tail on the 16 finished channels [n, n+16) of pixel m (v = conv + bias, the pre-activation): v becomes dL/dpre, the return value
// is sum d^2.  The arithmetic of loss_act_quad (fused_tail.hip), operation for operation.
__device__ __forceinline__ float tail16(const X6PArgs& a, int m, int n, float (&v)[16]) {
    const int b = (int)(((long)m * a.Cout) / a.tail_per_image);
    const int it = *a.tail_iter;
    const float* y = a.tail_tgt + (long)a.tail_idx[(long)it * a.tail_B + b] * a.tail_per_image + ((long)m * a.Cout - (long)b * a.tail_per_image) + n;
    const float slope = a.tail_act == 1 ? 0.01f : 0.f;
    const float gs = a.tail_coef * 2.f * a.tail_inv_npix;
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = 0.f;
    if (a.tail_resp) {                                       // exact fp32 residual from its planes: (p0 + p1) + p2
        const u16* rp = a.tail_resp + ((long)(n >> 4) * a.M + m) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32x4 p0 = *reinterpret_cast<const u32x4*>(rp + 8 * h), p1 = *reinterpret_cast<const u32x4*>(rp + a.oplane + 8 * h),
                        p2 = *reinterpret_cast<const u32x4*>(rp + 2 * a.oplane + 8 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[8 * h + 2 * k] = (lo_f(p0[k]) + lo_f(p1[k])) + lo_f(p2[k]);
                r[8 * h + 2 * k + 1] = (hi_f(p0[k]) + hi_f(p1[k])) + hi_f(p2[k]);
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 y4 = *reinterpret_cast<const f32x4*>(y + 4 * c);
        float dd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float p = v[4 * c + k];
            float o = p;
            if (a.tail_act) o = p > 0.f ? p : slope * p;
            if (a.tail_resp) o += r[4 * c + k];
            dd[k] = o - y4[k];
            const float g = dd[k] * gs;
            v[4 * c + k] = a.tail_act ? (p > 0.f ? g : slope * g) : g;
        }
        acc += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
    }
    return acc;
}

__global__ __launch_bounds__(512, 2) void conv_fwd_x6p_kernel(X6PArgs a) {
    constexpr int BM = 256, BN = 192, KS = 16;
    constexpr int TM = 2, TN = 3;
    constexpr int APLANE = BM * 32, BPLANE = BN * 32;
    constexpr int STAGE = 3 * (APLANE + BPLANE);           // 42 KiB
    constexpr int RING = 3;
    constexpr int SROW = 196;                              // staging row stride in floats: 16-byte aligned, conflict-free b128 reads down a column

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [RING][STAGE]; reused by the epilogue as fp32 [128][SROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    const int li = lane & 31, lh = lane >> 5;
    const TileId tile = xcd_tile_id(a.xcd_mode);
    const int m0 = tile.m * BM, n0 = tile.n * BN;
    const long Min = (long)a.B * a.H * a.W;

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

    // ---- B loader: pieces wave, wave + 8 and wave + 16 of the 18 KiB [plane][row][chunk] image (waves 2..7 repeat their second piece)
    int dma_src[3], dma_k[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = wave + 8 * j < 18 ? wave + 8 * j : wave + 8;
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
    // DMA cursor (wave-uniform): (channel slice, kh, kw) of the next stage to fetch; stages run channel slice outer, taps inner
    // Optional rotation of the K order per tile (ablate bit 16): stage s of this workgroup is stage (s + rot) mod nsteps of its K
    // range.  Measured SLOWER (4 x 128^2 conv: 217 vs 204 us): the workgroups of an XCD walking the weight stream together is what lets
    // one L2 fill serve all of them -- kept only as a diagnostic.
    const int rot = (a.ablate & 16) ? (int)(((unsigned)tile.m * 37u + (unsigned)tile.n * 11u) % (unsigned)nsteps) : 0;
    int cabs = sbeg + rot;                                   // absolute stage (channel slice * taps + tap) under the cursor
    int ccs = cabs / taps;
    int ctap = cabs - ccs * taps;
    int ckh = ctap / a.KW, ckw = ctap - ckh * a.KW;
    int cstage = 0;
    // six DMA instructions per wave and stage (three activation planes, three weight pieces): the vmcnt bookkeeping below counts on
    // it.  dma_prepare computes the stage's source addresses, dma_issue(j) issues instruction j, dma_advance moves the cursor.
    const u16* asrc = zero;
    long apstep = 0;
    const u16* wsrc = a.wp;
    auto dma_prepare = [&]() {
        const int hi = hi0 + ckh, wi = wi0 + ckw;
        const bool ok = (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const long off = ((long)ccs * Min + (apix0 + hi * a.W + wi)) * 16 + achunk * 8;
        asrc = ok ? a.xp + off : zero;
        apstep = ok ? a.xplane : 0;
        if ((a.ablate & 64) && (ckh | ckw)) {                // diagnostic: activation tile fetched for tap (0,0) only (wrong results)
            asrc = zero;
            apstep = 0;
        }
        wsrc = a.wp + (long)cabs * a.Cout * 16;
        if (cstage + 1 < nsteps) {                           // advance, clamped at the last stage (tail stages re-fetch it: harmless)
            ++cstage;
            ++cabs;
            ++ckw;
            if (ckw == a.KW) { ckw = 0; ++ckh; }
            if (ckh == a.KH) { ckh = 0; ++ccs; }
            if (cabs == sbeg + nsteps) {                     // wrap to the start of this workgroup's K range
                cabs = sbeg;
                ccs = sbeg / taps;
                const int tp = sbeg - ccs * taps;
                ckh = tp / a.KW;
                ckw = tp - ckh * a.KW;
            }
        }
    };
    auto dma_issue = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        char* const base = smem + buf * STAGE;
        if constexpr (j < 3) {
            if (!(a.ablate & 1))
                __builtin_amdgcn_global_load_lds((glb_void*)(asrc + j * apstep), (lds_void*)(base + wave * 1024 + j * APLANE), 16, 0, 0);
        } else {
            if (!(a.ablate & 2))
                __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + dma_src[j - 3]), (lds_void*)(base + 3 * APLANE + dma_k[j - 3] * 1024), 16, 0, 0);
        }
    };
    auto dma_all = [&](int buf) {
        dma_prepare();
        [&]<int... J>(std::integer_sequence<int, J...>) { (dma_issue(std::integral_constant<int, J>{}, buf), ...); }
        (std::make_integer_sequence<int, 6>{});
    };

    int fa_off[3][TM], fb_off[3][TN];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa_off[p][i] = p * APLANE + chunk_off16(wm0 + i * 32 + li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = 3 * APLANE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    }
    // the second-dispatched half of the workgroup (waves 4-7, the SIMD partners of waves 0-3) issues its DMAs in the LAST three MFMA
    // slots of a stage, the first half in the FIRST three: while one wave of a SIMD is held up issuing DMAs its partner multiplies
    const bool late = wave >= 4;

    dma_all(0);
    dma_all(1);
    int buf = 0;
    for (int t = 0; t < nsteps; ++t) {
        // stage t has landed for this wave once at most the six youngest DMAs (stage t+1) are outstanding; the barrier extends that to
        // every wave and also tells that nobody still reads buffer (t+2) % 3 = (t-1) % 3
        if (a.ablate & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nb = buf >= 1 ? buf - 1 : RING - 1;        // (t + 2) % 3
        const char* st = smem + buf * STAGE;
        bf16x8 fa[3][TM], fb[3][TN];
        auto rd_a = [&](int p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(st + fa_off[p][i]);
        };
        auto rd_b = [&](int p) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(st + fb_off[p][j]);
        };
        auto mma = [&](int pa, int pb) {
            if (a.ablate & 4) return;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
        auto dma6 = [&]() {
            dma_prepare();
            dma_issue(I0{}, nb); dma_issue(I1{}, nb); dma_issue(I2{}, nb); dma_issue(I3{}, nb); dma_issue(I4{}, nb); dma_issue(I5{}, nb);
        };
        // The SIMD partners (wave w and w + 4) leave the barrier together.  Shipped schedule: two DMAs per MFMA slot, waves 0-3 in slots
        // 0-2, waves 4-7 in slots 3-5 (203 us on the 4 x 128^2 conv, MFMAs alone 170 us on that box).  ablate bit 32 selects the
        // weight-gradient kernel's schedule instead -- waves 0-3 issue all six before their reads, waves 4-7 after three slots -- which
        // is SLOWER here (220 us).
        const bool old_sched = (a.ablate & 32) == 0;
        if (old_sched) dma_prepare();
        else if (!late) dma6();
        // fragments in the order the products need them: (A2,B0) (A0,B2) (A1,B0) (A1,B1) (A0,B1) (A0,B0), small terms first
        rd_a(2); rd_b(0);
        rd_a(0); rd_b(2);
        __builtin_amdgcn_sched_barrier(0);
        mma(2, 0);
        rd_a(1);
        if (old_sched && !late) { dma_issue(I0{}, nb); dma_issue(I1{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 2);
        rd_b(1);
        if (old_sched && !late) { dma_issue(I2{}, nb); dma_issue(I3{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        if (old_sched && !late) { dma_issue(I4{}, nb); dma_issue(I5{}, nb); }
        if (!old_sched && late) dma6();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        if (old_sched && late) { dma_issue(I0{}, nb); dma_issue(I1{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        if (old_sched && late) { dma_issue(I2{}, nb); dma_issue(I3{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        if (old_sched && late) { dma_issue(I4{}, nb); dma_issue(I5{}, nb); }
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();

    // ---- epilogue through LDS: two passes of 128 tile rows (pass i: MFMA row tile i of every wave).  Per pass: (1) consecutive lanes
    // along the CHANNELS finish quads -- bias, activation, residual, coalesced 16-byte loads / stores of the fp32 tensors -- and put
    // the finished values back into the staging tile; (2) consecutive lanes along the PIXELS split the 16 channels of one slice and
    // write the three P3 records (32 contiguous bytes per plane and pixel, consecutive pixels adjacent).
    float* const stg = reinterpret_cast<float*>(smem);       // [128][SROW] fp32 = 98 KiB
    const long part_base = (long)tile.z * a.M * a.Cout;
    const bool need_planes = a.outp != nullptr && a.partial == nullptr;
#pragma unroll
    for (int pass = 0; pass < TM; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stg[((wave >> 1) * 32 + rr) * SROW + wn0 + j * 32 + li] = acc[pass][j][r];
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
                store_slice(a, m, n, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Halo variant for 3 x 3, stride 1, pad 1 convs whose H and W are multiples of 16: the 256 output pixels of a workgroup are a
// 16 x 16 PATCH of one image, and the 18 x 18 halo of that patch (16 channels of one slice, three planes = 30.4 KiB) is fetched ONCE
// per channel slice and serves all nine taps -- a tap is an offset into the halo tile when the A fragments are read.  The kernel above
// re-fetches a shifted 256-pixel tile for every tap: 24 KiB x 9 per slice against 30.4 KiB here, i.e. 57 % of its L2 -> LDS bytes
// overall (weights unchanged).  That matters because this kernel runs at the board's power limit (tools/clock_probe.py: 1.37 kW,
// clock pulled down to 2.2-2.3 GHz; MFMAs alone 0.94 kW at 2.4 GHz): with the activation DMA of eight taps in nine pointed at the zero
// page (x6p_ablate bit 64) the same launch took 181 us instead of 204.  LDS: two halo buffers (slice cs and cs + 1) + the three-deep
// weight ring = 118 KiB.  DMA per wave and stage: three weight pieces for stage s + 2 and, in taps 0-3, one of the wave's four pieces
// of the next slice's halo; the counted vmcnt at the top of a stage is therefore 4 in taps 1-4 and 3 otherwise (the halo piece is
// issued before the weight pieces of its stage).
__global__ __launch_bounds__(512, 2) void conv_fwd_x6h_kernel(X6PArgs a) {
    constexpr int BN = 192;
    constexpr int TM = 2, TN = 3;
    constexpr int PT = 16, HWD = PT + 2, HPIX = HWD * HWD;   // patch edge, halo edge, halo pixels (324)
    constexpr int APL = HPIX * 32;                           // one plane of the halo tile: 10368 B
    constexpr int APIECES = 3 * HPIX * 2;                    // 16-byte pieces of the three planes: 1944 (31 wave instructions)
    constexpr int ABUF = 32768;
    constexpr int BPLANE = BN * 32, BSTAGE = 3 * BPLANE;     // 18 KiB
    constexpr int RING = 3;
    constexpr int BBASE = 2 * ABUF;
    constexpr int SROW = 196;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][ABUF] halo tiles, [RING][BSTAGE] weights; epilogue: fp32 [128][SROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
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

    // ---- A loader: pieces wave, wave + 8, wave + 16, wave + 24 of the [plane][halo pixel][chunk] image
    int a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
    // ---- B loader: as in conv_fwd_x6p_kernel
    int dma_src[3], dma_k[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = wave + 8 * j < 18 ? wave + 8 * j : wave + 8;
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

    int q00[TM], fb_off[3][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tl = wm0 + i * 32 + li;
        q00[i] = (tl >> 4) * HWD + (tl & 15);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb_off[p][j] = BBASE + p * BPLANE + chunk_off16(wn0 + j * 32 + li, lh);
    const bool late = wave >= 4;

    // prologue: halo of slice 0, weights of stages 0 and 1
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(j, 0);
    set_wsrc(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_b(j, 0);
    set_wsrc(1);
#pragma unroll
    for (int j = 0; j < 3; ++j) dma_b(j, 1);

    int buf = 0, s = 0;
    auto stage = [&](auto tapc, int cs) {
        constexpr int TAP = decltype(tapc)::value;
        constexpr int KH = TAP / 3, KW = TAP % 3;
        // stage s (weights) and, in tap 0, this slice's halo have landed once at most the DMAs issued after them are outstanding
        if (a.ablate & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (TAP >= 1 && TAP <= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nb = buf >= 1 ? buf - 1 : RING - 1;        // (s + 2) % 3
        const char* stb = smem + buf * BSTAGE;
        const char* sta = smem + (cs & 1) * ABUF;
        int fa_off[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = q00[i] + KH * HWD + KW;
            fa_off[i] = q * 32 + (((lh ^ (q >> 3)) & 1) << 4);
        }
        bf16x8 fa[3][TM], fb[3][TN];
        auto rd_a = [&](int p) {
            if (a.ablate & 8) return;                        // diagnostic: no fragment reads (stale registers multiplied)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const bf16x8*>(sta + p * APL + fa_off[i]);
        };
        auto rd_b = [&](int p) {
            if (a.ablate & 8) return;
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(stb + fb_off[p][j]);
        };
        auto mma = [&](int pa, int pb_) {
            if (a.ablate & 4) return;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb_][j], acc[i][j], 0, 0, 0);
        };
        set_wsrc(s + 2);
        auto slot = [&](int k) {                             // DMA slot k of this wave's three: halo piece first (taps 0-3), then weight piece k
            if (k == 0) {
                if constexpr (TAP < 4) dma_a(TAP, cs + 1);
            }
            dma_b(k, nb);
        };
        // fragments in the order the products need them: (A2,B0) (A0,B2) (A1,B0) (A1,B1) (A0,B1) (A0,B0), small terms first
        rd_a(2); rd_b(0);
        rd_a(0); rd_b(2);
        __builtin_amdgcn_sched_barrier(0);
        mma(2, 0);
        rd_a(1);
        if (!late) slot(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 2);
        rd_b(1);
        if (!late) slot(1);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        if (!late) slot(2);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        if (late) slot(0);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        if (late) slot(1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        if (late) slot(2);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == RING ? 0 : buf + 1;
        ++s;
    };
    for (int cs = 0; cs < csteps; ++cs) {
        [&]<int... T>(std::integer_sequence<int, T...>) { (stage(std::integral_constant<int, T>{}, cs), ...); }
        (std::make_integer_sequence<int, 9>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail DMAs still target LDS: drain before the epilogue reuses it
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: as in conv_fwd_x6p_kernel, tile row tl = pixel (tl / 16, tl % 16) of the patch
    float* const stg = reinterpret_cast<float*>(smem);
    const bool need_planes = a.outp != nullptr;
    float tail_loss = 0.f;
#pragma unroll
    for (int pass = 0; pass < TM; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stg[((wave >> 1) * 32 + rr) * SROW + wn0 + j * 32 + li] = acc[pass][j][r];
            }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 12; ++k) {
            const int q = tid + 512 * k;                     // 128 rows x 48 quads of channels
            const int row = q / 48, c4 = q - row * 48;
            const int tl = (row >> 5) * 64 + pass * 32 + (row & 31);
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
            for (int k = 0; k < 3; ++k) {
                const int q = tid + 512 * k;                 // 128 rows x 12 slices of 16 channels
                const int row = q & 127, sl = q >> 7;
                const int tl = (row >> 5) * 64 + pass * 32 + (row & 31);
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
                store_slice(a, m, n, v);
            }
        }
    }
    if (a.tail_tgt) {                                        // one atomic per workgroup (256 of them, 8 per log slot)
        __shared__ float red[8];
        for (int o = 32; o > 0; o >>= 1) tail_loss += __shfl_down(tail_loss, o, 64);
        if (lane == 0) red[wave] = tail_loss;
        __syncthreads();
        if (tid == 0 && a.tail_loss)
            atomicAdd(a.tail_loss + (long)(*a.tail_iter) * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)),
                      (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) * (a.tail_inv_npix * a.tail_coef));
    }
}

// split-K second pass: sum the partial accumulators, then the same per-slice finish (planes included); lanes along pixels
__global__ __launch_bounds__(256) void x6p_splitk_epilogue_kernel(X6PArgs a) {
    const int spr = a.Cout / 16;                             // slices per row
    const long total = (long)a.M * spr;
    const long slab = (long)a.M * a.Cout;
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
        finish16(a, m, n, v);
    }
}

// fp32 NHWC tensor [M][C] -> P3 planes (producer of last resort for tensors that no fused kernel writes as planes)
__global__ __launch_bounds__(256) void split_p3_kernel(const float* x, long M, int C, u16* planes) {
    const int spr = C / 16;
    const long total = M * spr * 2;                          // half slices of 8 channels
    const long pstride = M * C;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int h = (int)(q & 1);
        const long r = q >> 1;
        const int sl = (int)(r % spr);
        const long m = r / spr;
        const float* src = x + m * C + sl * 16 + 8 * h;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
        const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        u32x4 p0, p1, p2;
        split3_x8(v, p0, p1, p2);
        u16* dst = planes + ((long)sl * M + m) * 16 + 8 * h;
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + pstride) = p1;
        *reinterpret_cast<u32x4*>(dst + 2 * pstride) = p2;
    }
}

}  // namespace

extern "C" int rdo_conv2d_fwd_uses_bf16x6(const rdo_conv_desc* d);

// K split of the plane kernel: enough 256 x 192 x (K / ks) workgroups for one per CU, at least 12 K stages per split, at most 9 splits
static int p3_ksplit(const rdo_conv_desc* d) {
    const long M = (long)d->B * d->Ho * d->Wo;
    const long tiles = rdo::ceil_div(M, 256) * rdo::ceil_div(d->Cout, 192);
    const long stages = (long)d->KH * d->KW * (d->Cin / 16);
    int best = 0;
    for (int ks = 1; ks <= 9; ++ks)
        if (tiles * ks >= 192 && stages / ks >= 12) {
            best = ks;
            if (tiles * ks >= 256) break;
        }
    return best;
}

extern "C" int rdo_split_p3(const float* x, int64_t npix, int32_t C, void* planes, void* stream) {
    RDO_REQUIRE(x && planes && npix > 0 && C > 0 && C % 16 == 0, "rdo_split_p3: C (%d) must be a positive multiple of 16", C);
    u16* p = reinterpret_cast<u16*>(planes);
    const double n = (double)npix * C;
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div(npix * (C / 8), 256);
            hipLaunchKernelGGL(split_p3_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, x, (long)npix, C, p);
            return rdo::check_launch("split_p3");
        },
        stream, "split_p3", 0.0, 10.0 * n);
}

extern "C" int rdo_conv2d_fwd_p3_supported(const rdo_conv_desc* d) {
    if (!d || d->square_input || d->Cin % 16 != 0 || d->Cout % 16 != 0) return 0;
    if ((double)d->B * d->H * d->W * d->Cin * 3.0 >= 2147483648.0 || (double)d->Cout * d->KH * d->KW * d->Cin * 3.0 >= 2147483648.0) return 0;
    if (!rdo::tuning(rdo::T_CONV_X6)) return 0;
    return p3_ksplit(d) >= 1;
}

extern "C" int64_t rdo_conv2d_fwd_p3_workspace(const rdo_conv_desc* d) {
    if (!d || !rdo_conv2d_fwd_p3_supported(d)) return 0;
    const int ks = p3_ksplit(d);
    return ks > 1 ? (int64_t)ks * d->B * d->Ho * d->Wo * d->Cout : 0;
}

static bool p3_halo_shape(const rdo_conv_desc* d) {
    return rdo::tuning(rdo::T_X6P_HALO) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->H % 16 == 0 && d->W % 16 == 0 &&
           p3_ksplit(d) == 1;
}

static int conv2d_fwd_p3_impl(const rdo_conv_desc* d, const void* x_planes, const void* wplanes, const float* bias, const float* aux,
                              const void* aux_planes, const float* residual, float* out, float* pre, void* out_planes,
                              float* workspace, int64_t workspace_floats, void* stream, const X6PArgs* tail) {
    RDO_REQUIRE(d && x_planes && wplanes && (out || out_planes || pre), "rdo_conv2d_fwd_p3: null argument");
    RDO_REQUIRE(rdo_conv2d_fwd_p3_supported(d), "rdo_conv2d_fwd_p3: shape not on the split-bf16 plane path (rdo_conv2d_fwd_p3_supported)");
    const int ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    RDO_REQUIRE(ho == d->Ho && wo == d->Wo, "rdo_conv2d_fwd_p3: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
    const int epi = d->epilogue;
    RDO_REQUIRE(epi >= RDO_EPI_NONE && epi <= RDO_EPI_RELU_BWD, "rdo_conv2d_fwd_p3: unknown epilogue %d", epi);
    RDO_REQUIRE(epi == RDO_EPI_NONE || epi == RDO_EPI_LRELU || epi == RDO_EPI_RELU || aux != nullptr ||
                    (aux_planes != nullptr && (epi == RDO_EPI_LRELU_BWD || epi == RDO_EPI_RELU_BWD)),
                "rdo_conv2d_fwd_p3: epilogue %d needs aux (aux_planes serve the activation-backward epilogues only)", epi);
    RDO_REQUIRE(!d->add_residual || residual != nullptr, "rdo_conv2d_fwd_p3: add_residual without residual");
    X6PArgs a{};
    a.xp = reinterpret_cast<const u16*>(x_planes);
    a.xplane = (long)d->B * d->H * d->W * d->Cin;
    a.wp = reinterpret_cast<const u16*>(wplanes);
    a.wplane = (long)d->Cout * d->KH * d->KW * d->Cin;
    a.bias = bias; a.aux = aux; a.auxp = aux ? nullptr : reinterpret_cast<const u16*>(aux_planes); a.residual = residual; a.out = out; a.pre = pre;
    a.outp = reinterpret_cast<u16*>(out_planes);
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.oplane = (long)a.M * a.Cout;
    a.epilogue = d->epilogue; a.add_residual = d->add_residual;
    a.xcd_mode = rdo::tuning(rdo::T_XCD);
#ifdef RDO_DIAG
    a.ablate = rdo::tuning(rdo::T_X6P_ABLATE);
#endif
    if (tail) {
        a.tail_tgt = tail->tail_tgt; a.tail_idx = tail->tail_idx; a.tail_iter = tail->tail_iter; a.tail_resp = tail->tail_resp;
        a.tail_loss = tail->tail_loss; a.tail_per_image = tail->tail_per_image; a.tail_B = tail->tail_B; a.tail_act = tail->tail_act;
        a.tail_coef = tail->tail_coef; a.tail_inv_npix = tail->tail_inv_npix;
    }
    int ks = p3_ksplit(d);
    if (ks < 1) ks = 1;
    if (ks > 1 && (!workspace || (long)ks * a.M * a.Cout > workspace_floats)) ks = 1;
    a.ksplit = ks;
    a.partial = ks > 1 ? workspace : nullptr;
    const double flops = 2.0 * a.M * (double)a.Cout * a.Cin * a.KH * a.KW;
    const double bytes = 6.0 * a.xplane + 6.0 * a.wplane +
                         (double)a.M * a.Cout * (4.0 * ((out != nullptr) + (pre != nullptr) + (aux != nullptr) + (residual != nullptr)) +
                                                 6.0 * (out_planes != nullptr));
    // 3 x 3 "same" convs on 16 x 16 patches: the halo kernel (no K split: the shapes that qualify for P3 fill the chip with tiles)
    const bool halo = p3_halo_shape(d) && ks == 1;
    RDO_REQUIRE(!tail || halo, "rdo_conv2d_fwd_p3_tail: shape not on the halo kernel (rdo_conv2d_fwd_p3_tail_supported)");
    if (halo)
        return rdo::dispatch(
            [a](hipStream_t s) {
                constexpr size_t lds = (size_t)2 * 32768 + 3 * 3 * 192 * 32;
                static rdo::PerDevice attr;
                if (!attr.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_x6h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6h) failed");
                    attr.mark();
                }
                dim3 grid((unsigned)(a.M / 256), (unsigned)rdo::ceil_div(a.Cout, 192), 1);
                hipLaunchKernelGGL(conv_fwd_x6h_kernel, grid, dim3(512), lds, s, a);
                return rdo::check_launch("conv_fwd_x6h");
            },
            stream, "conv_fwd_x6_p3_halo", flops, bytes);
    return rdo::dispatch(
        [a](hipStream_t s) {
            constexpr size_t lds = (size_t)3 * 3 * (256 + 192) * 32;
            static rdo::PerDevice attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_x6p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(conv_fwd_x6p) failed");
                attr.mark();
            }
            dim3 grid((unsigned)rdo::ceil_div(a.M, 256), (unsigned)rdo::ceil_div(a.Cout, 192), (unsigned)a.ksplit);
            hipLaunchKernelGGL(conv_fwd_x6p_kernel, grid, dim3(512), lds, s, a);
            if (int rc = rdo::check_launch("conv_fwd_x6p")) return rc;
            if (a.ksplit > 1) {
                long g = rdo::ceil_div((long)a.M * a.Cout / 16, 256);
                hipLaunchKernelGGL(x6p_splitk_epilogue_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s, a);
                return rdo::check_launch("x6p_splitk_epilogue");
            }
            return RDO_OK;
        },
        stream, "conv_fwd_x6_p3_256x192", flops, bytes);
}

extern "C" int rdo_conv2d_fwd_p3(const rdo_conv_desc* d, const void* x_planes, const void* wplanes, const float* bias, const float* aux,
                                 const void* aux_planes, const float* residual, float* out, float* pre, void* out_planes,
                                 float* workspace, int64_t workspace_floats, void* stream) {
    return conv2d_fwd_p3_impl(d, x_planes, wplanes, bias, aux, aux_planes, residual, out, pre, out_planes, workspace, workspace_floats, stream,
                              nullptr);
}

extern "C" int rdo_conv2d_fwd_p3_tail_supported(const rdo_conv_desc* d) {
    return d && rdo_conv2d_fwd_p3_supported(d) && p3_halo_shape(d) && d->epilogue == RDO_EPI_NONE && !d->add_residual;
}

// Last conv of a unit + its tail in ONE launch (halo kernel): the epilogue forms out = act(conv + bias) + residual, the loss against
// the cached target rows and dL/dpre, and writes only the planes of dL/dpre -- rdo_conv2d_fwd_p3 + rdo_loss_act_bwd without the
// pre-activation tensor ever reaching memory.
extern "C" int rdo_conv2d_fwd_p3_tail(const rdo_conv_desc* d, const void* x_planes, const void* wplanes, const float* bias,
                                      const void* residual_planes, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr,
                                      int32_t B, float coef, int32_t act, void* dpre_planes, float* loss_out, void* stream) {
    RDO_REQUIRE(d && tgt_cache && idx_table && iter_ptr && dpre_planes, "rdo_conv2d_fwd_p3_tail: null argument");
    RDO_REQUIRE(rdo_conv2d_fwd_p3_tail_supported(d), "rdo_conv2d_fwd_p3_tail: shape / epilogue not supported (rdo_conv2d_fwd_p3_tail_supported)");
    RDO_REQUIRE(B == d->B && act >= 0 && act <= 2, "rdo_conv2d_fwd_p3_tail: B must be the conv's batch, act in 0..2");
    X6PArgs t{};
    t.tail_tgt = tgt_cache; t.tail_idx = idx_table; t.tail_iter = iter_ptr; t.tail_resp = reinterpret_cast<const u16*>(residual_planes);
    t.tail_loss = loss_out; t.tail_per_image = (long)d->Ho * d->Wo * d->Cout; t.tail_B = B; t.tail_act = act; t.tail_coef = coef;
    t.tail_inv_npix = (float)(1.0 / ((double)B * d->Ho * d->Wo));
    return conv2d_fwd_p3_impl(d, x_planes, wplanes, bias, nullptr, nullptr, nullptr, nullptr, nullptr, dpre_planes, nullptr, 0, stream, &t);
}
