// Mini-batch assembly (K7: gather + QDrop, layer_opt.py:289-292) as device functions shared by the stand-alone gather kernels
// (elementwise.hip, fused_tail.hip) and by the AdaRound step launch (adaround.hip), which from round 6 on also assembles the NEXT
// iteration's mini-batch: one implementation of the index arithmetic, the counter RNG and the plane split, three callers.
//
//   keep = u32(seed, iter, i) < floor(p * 2^32) takes the quantised-prefix input; i = the element's index in the GLOBAL mini-batch
//   ((batch_offset + b) * per_image + offset): a data-parallel rank draws exactly the mask slice a single process would.
#pragma once
#include "rdo_common.h"

namespace rdo {
namespace gq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

struct H2Out {          // planes of an H2 tensor being written: [2][M*C] fp16, values times `s`
    unsigned short* p;
    float s;
    int* ovf;           // sticky overflow flag (rdo_h2_overflow)
};
struct H2In {           // planes being read back as fp32: (h1 + h2) * inv
    const unsigned short* p;
    float inv;
};

// ---- pixel-major work split of the kernels that write (or read) planes ----------------------------------------------------------
// A virtual block covers 64 pixels x one group of 32 channels (one 128-byte line of the fp32 tensors per pixel); a thread owns 8
// consecutive channels of one pixel (quads at c0 and c0 + 4): two 16-byte fp32 accesses per tensor and ONE 16-byte access per plane.
// Lanes 4p..4p+3 hold pixel p, so the plane stores of a wave are two runs of 16 consecutive 32-byte records (slices 2g and 2g+1)
// instead of 8-byte pieces scattered over twelve slices, which is what a channel-major thread order produces.  (Measured at
// 4 x 128^2 x 192, tools/bench_tails.py: 16-byte plane stores in this order 35 us for gather + planes; 8-byte stores -- channel-major,
// or pixel-major with the two quads 16 channels apart so that the fp32 side is sector-complete -- 42-56 us.)
constexpr int Q2 = 4;       // channel distance between the two quads of a thread

struct Oct {
    long m;     // pixel
    int c0;     // first of the 8 channels
};
__device__ __forceinline__ bool oct_of(long vb, int ngroups, long M, int C, Oct& o) {
    const int grp = (int)(vb % ngroups);
    o.m = (vb / ngroups) * 64 + (threadIdx.x >> 2);
    o.c0 = grp * 32 + (threadIdx.x & 3) * 8;
    return o.m < M && o.c0 < C;          // C % 16 == 0: the 8 channels exist together
}
inline long oct_blocks(long M, int C) { return rdo::ceil_div(M, 64L) * rdo::ceil_div(C, 32); }
inline unsigned oct_grid(long M, int C) {
    const long g = oct_blocks(M, C);
    return (unsigned)(g > 8192 ? 8192 : g);
}

__device__ __forceinline__ void store_h2_oct(const H2Out& pl, long M, int C, const Oct& o, const f32x4& a, const f32x4& b, int& bad) {
    const long pstride = M * C;
    const long e = ((long)(o.c0 >> 4) * M + o.m) * 16 + (o.c0 & 15);
    u32x4 hi, lo;
    unsigned h, l;
    rdo::h2_split_pk(a[0], a[1], pl.s, h, l, bad); hi[0] = h; lo[0] = l;
    rdo::h2_split_pk(a[2], a[3], pl.s, h, l, bad); hi[1] = h; lo[1] = l;
    rdo::h2_split_pk(b[0], b[1], pl.s, h, l, bad); hi[2] = h; lo[2] = l;
    rdo::h2_split_pk(b[2], b[3], pl.s, h, l, bad); hi[3] = h; lo[3] = l;
    *reinterpret_cast<u32x4*>(pl.p + e) = hi;
    *reinterpret_cast<u32x4*>(pl.p + pstride + e) = lo;
}

__device__ __forceinline__ const f32x4& ldq(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stq(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 qdrop_quad(const float* cq, const float* cfp, long src, uint32_t i0, uint32_t key, unsigned long long thr) {
    const f32x4 q = ldq(cq + src), f = ldq(cfp + src);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((unsigned long long)lowbias32((i0 + e) ^ key) < thr) ? q[e] : f[e];
    return o;
}

// What a gather needs besides the iteration number: the caches, the mini-batch index table, the QDrop key and the destination(s).
struct Gather {
    const float* cq;
    const float* cfp;
    const int32_t* idx_table;      // [n_iters][B]
    int n_iters;                   // rows of idx_table (the step launch skips the gather behind the last iteration)
    int B, batch_offset;
    long per_image;
    int C;                         // channels (plane form only)
    unsigned long long thr;        // floor(p * 2^32)
    uint32_t seed;
    float* out;                    // fp32 mini-batch (nullable in the plane form)
    H2Out planes;                  // p == nullptr: fp32 form
};

// fp32 form: block `vb` of `nvb` (256 threads each), grid-stride over the 16-byte quads
__device__ __forceinline__ void gather_fp32_body(const Gather& g, int it, long vb, long nvb) {
    const uint32_t key = lowbias32((uint32_t)it + g.seed * 0x9E3779B9u);
    const long quads = g.per_image / 4;
    const long total = (long)g.B * quads;
    for (long t = vb * 256 + threadIdx.x; t < total; t += nvb * 256) {
        const int b = (int)(t / quads);
        const long off = (t - (long)b * quads) * 4;
        const long src = (long)g.idx_table[(long)it * g.B + b] * g.per_image + off;
        const uint32_t i0 = (uint32_t)((long)(g.batch_offset + b) * g.per_image + off);     // element index in the GLOBAL mini-batch
        stq(g.out + (long)b * g.per_image + off, qdrop_quad(g.cq, g.cfp, src, i0, key, g.thr));
    }
}

// plane form (+ optional fp32 copy): virtual blocks of 64 pixels x 32 channels, block `vb0` of `nvb` walks them with stride nvb
__device__ __forceinline__ void gather_h2_body(const Gather& g, int it, long vb0, long nvb) {
    int bad = 0;
    const uint32_t key = lowbias32((uint32_t)it + g.seed * 0x9E3779B9u);
    const int C = g.C;
    const long ppi = g.per_image / C, M = (long)g.B * ppi;
    const int ngroups = (C + 31) / 32;
    const long total = ((M + 63) / 64) * ngroups;
    for (long vb = vb0; vb < total; vb += nvb) {
        Oct o;
        if (!oct_of(vb, ngroups, M, C, o)) continue;
        const int b = (int)(o.m / ppi);
        const long row = (long)g.idx_table[(long)it * g.B + b] * g.per_image;
        f32x4 v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = o.c0 + k * Q2;
            const long off = (o.m - (long)b * ppi) * C + c;                           // element offset inside the image
            const uint32_t i0 = (uint32_t)((long)(g.batch_offset + b) * g.per_image + off);
            v[k] = qdrop_quad(g.cq, g.cfp, row + off, i0, key, g.thr);
        }
        // all loads before any store (the stores may alias the inputs as far as the compiler knows)
        if (g.out) {
            stq(g.out + o.m * C + o.c0, v[0]);
            stq(g.out + o.m * C + o.c0 + Q2, v[1]);
        }
        store_h2_oct(g.planes, M, C, o, v[0], v[1], bad);
    }
    rdo::h2_report(bad, g.planes.ovf);
}

inline long gather_blocks(const Gather& g) {      // workgroups of 256 threads that cover the gather once
    if (g.planes.p) return oct_blocks((long)g.B * (g.per_image / g.C), g.C);
    return rdo::ceil_div((long)g.B * g.per_image / 4, 256L);
}

}  // namespace gq
}  // namespace rdo
