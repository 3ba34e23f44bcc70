// Fused "unit tail" kernels: everything between the last forward conv of a reconstruction unit and the first backward GEMM, in one
// HBM pass, plus the producers of "H2" tensors (exact two-way fp16 splits of the power-of-two-scaled values, planes [2][n],
// rdo_common.h) that the split-precision GEMM kernels read by LDS-DMA (conv_fwd_h2.hip).
//
//   rdo_gather_qdrop_h2   cached rows -> QDrop mix -> mini-batch as fp32 AND as planes                   (layer_opt.py:289-292)
//   rdo_loss_act_bwd      out = act(pre) + res ; d = out - tgt[idx] ; loss ; dL/dout ; dL/dpre (fp32 / planes)
//                         = conv epilogue activation + lp_loss forward/backward + activation backward     (layer_opt.py:133,150,303-306)
//   rdo_loss_gdn_bwd      out = x * norm^(-+1/2) + res ; loss ; dL/dout ; t = dL/dnorm (fp32 / planes)        (quant_layer.py:142-154)
//   rdo_gdn_bwd_dx_h2     dx = g * norm^(-+1/2) + 2 x acc as fp32 and planes
//   rdo_pixel_shuffle_h2  F.pixel_shuffle(r = 2) writing fp32 and planes
//
// The arithmetic of each stage is the one of the separate kernels it replaces (elementwise.hip, the conv epilogues), op for op,
// and the file is built with -ffp-contract=off like them: fused and unfused paths produce the same bits except for the order in
// which the loss partial sums are added.
#include "rdo_common.h"
#include "gather_body.h"

namespace {

using namespace rdo::gq;          // f32x4, u32x4, H2Out / H2In, the pixel-major work split (Oct), lowbias32, the gather bodies
typedef unsigned short u16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// loss kernels: every workgroup ends with one float atomic into one of the 32 log slots; same-address atomics serialise in L2, so the
// grid is capped (grid-stride loops cover the rest)
inline unsigned loss_grid(long blocks) {
    const long cap = rdo::tuning(rdo::T_TAIL_GRID);
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

inline unsigned grid_for(long n) {
    long g = rdo::ceil_div(n, 256L);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// (pub: rdo_iter_bind_publish -- the launch's first thread leaves the iteration number there for the AdaRound step of the same iteration)
__device__ __forceinline__ void block_loss_add(float acc, float scale, float* loss_out, int it, int32_t* pub) {
    if (pub && blockIdx.x == 0 && threadIdx.x == 0) *pub = it;
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out)
        atomicAdd(loss_out + (long)it * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)), (red[0] + red[1] + red[2] + red[3]) * scale);
}

// fp32 values of the thread's 8 channels from planes: (h1 + h2) / s -- the sum is exact in fp32, the value is the original to 2^-24
__device__ __forceinline__ void load_h2_oct(const H2In& pl, long M, int C, const Oct& o, f32x4& a, f32x4& b) {
    const long pstride = M * C;
    const long e = ((long)(o.c0 >> 4) * M + o.m) * 16 + (o.c0 & 15);
    const u32x4 h = *reinterpret_cast<const u32x4*>(pl.p + e), l = *reinterpret_cast<const u32x4*>(pl.p + pstride + e);
    using rdo::h2_hi;
    using rdo::h2_lo;
    a = f32x4{(h2_lo(h[0]) + h2_lo(l[0])) * pl.inv, (h2_hi(h[0]) + h2_hi(l[0])) * pl.inv, (h2_lo(h[1]) + h2_lo(l[1])) * pl.inv,
              (h2_hi(h[1]) + h2_hi(l[1])) * pl.inv};
    b = f32x4{(h2_lo(h[2]) + h2_lo(l[2])) * pl.inv, (h2_hi(h[2]) + h2_hi(l[2])) * pl.inv, (h2_lo(h[3]) + h2_lo(l[3])) * pl.inv,
              (h2_hi(h[3]) + h2_hi(l[3])) * pl.inv};
}

// ---- gather + QDrop (gather_body.h) ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_qdrop_h2_kernel(Gather g, const int32_t* iter_ptr, int32_t* iter_publish) {
    const int it = *iter_ptr;
    if (iter_publish && blockIdx.x == 0 && threadIdx.x == 0) *iter_publish = it;
    gather_h2_body(g, it, blockIdx.x, gridDim.x);
}

// ---- conv activation + loss + gradient + activation backward ------------------------------------------------------------------------
// act: 0 none, 1 LeakyReLU(0.01), 2 ReLU.  One quad: o = act(p) (+ r); d = o - y; returns sum d^2; g = dL/dout; dp = dL/dpre
__device__ __forceinline__ float loss_act_quad(const f32x4& p, const f32x4& y, const f32x4* r, int act, float slope, float gs, f32x4& o,
                                               f32x4& g, f32x4& dp) {
    o = p;
    if (act) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = p[k] > 0.f ? p[k] : slope * p[k];
    }
    if (r) o += *r;
    const f32x4 dd = o - y;
    g = dd * gs;
    dp = g;
    if (act) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dp[k] = p[k] > 0.f ? g[k] : slope * g[k];
    }
    return (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
}

// fp32 outputs only: channel-major thread order, every access a fully coalesced 16 bytes per lane
// `partial` != nullptr: the pre-activation is still the K-split partial sums of the conv (rdo_conv2d_fwd_partials): pre = (0 + slab 0
// + slab 1 + ...) + bias, the sum order of the conv's own second pass
__global__ __launch_bounds__(256) void loss_act_bwd_kernel(const float* pre, const float* partial, int ks, long slab, const float* bias,
                                                           int C, const float* res, const float* tgt, const int32_t* idx_table,
                                                           const int32_t* iter_ptr, int B, long per_image, float inv_npix, float coef,
                                                           int act, float* out, float* gout, float* dpre, float* loss_out, int32_t* iter_pub) {
    const int it = *iter_ptr;
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    const float gs = coef * 2.f * inv_npix;
    const float slope = act == 1 ? 0.01f : 0.f;
    float acc = 0.f;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / quads);
        const long off = (t - (long)b * quads) * 4;
        const long e = (long)b * per_image + off;
        const f32x4 y = ldq(tgt + (long)idx_table[(long)it * B + b] * per_image + off);
        f32x4 p;
        if (partial) {
            p = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < ks; ++z) p += ldq(partial + z * slab + e);
            if (bias) p += ldq(bias + (int)(e % C));
        } else {
            p = ldq(pre + e);
        }
        f32x4 r, o, g, dp;
        if (res) r = ldq(res + e);
        acc += loss_act_quad(p, y, res ? &r : nullptr, act, slope, gs, o, g, dp);
        if (out) stq(out + e, o);
        if (gout) stq(gout + e, g);
        if (dpre) stq(dpre + e, dp);
    }
    block_loss_add(acc, inv_npix * coef, loss_out, it, iter_pub);
}

// with planes on either side (dL/dpre out, residual in): pixel-major
__global__ __launch_bounds__(256) void loss_act_bwd_pix_kernel(const float* pre, const float* res, H2In res_planes, const float* tgt,
                                                               const int32_t* idx_table, const int32_t* iter_ptr, int B, long per_image,
                                                               int C, float inv_npix, float coef, int act, float* out, float* gout,
                                                               float* dpre, H2Out dpre_planes, float* loss_out, int32_t* iter_pub) {
    int bad = 0;
    const int it = *iter_ptr;
    const long ppi = per_image / C, M = (long)B * ppi;
    const int ngroups = (C + 31) / 32;
    const long nvb = ((M + 63) / 64) * ngroups;
    const float gs = coef * 2.f * inv_npix;
    const float slope = act == 1 ? 0.01f : 0.f;
    float acc = 0.f;
    for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        Oct oc;
        if (!oct_of(vb, ngroups, M, C, oc)) continue;
        const int b = (int)(oc.m / ppi);
        const float* yrow = tgt + ((long)idx_table[(long)it * B + b] - b) * per_image;       // + e = the target of element e
        f32x4 r[2], d[2], o[2], g[2];
        const bool has_r = res || res_planes.p;
        const long e0 = oc.m * C + oc.c0;
        if (res_planes.p && !res) load_h2_oct(res_planes, M, C, oc, r[0], r[1]);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long e = e0 + k * Q2;
            if (res) r[k] = ldq(res + e);
            acc += loss_act_quad(ldq(pre + e), ldq(yrow + e), has_r ? &r[k] : nullptr, act, slope, gs, o[k], g[k], d[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (out) stq(out + e0 + k * Q2, o[k]);
            if (gout) stq(gout + e0 + k * Q2, g[k]);
            if (dpre) stq(dpre + e0 + k * Q2, d[k]);
        }
        if (dpre_planes.p) store_h2_oct(dpre_planes, M, C, oc, d[0], d[1], bad);
    }
    if (dpre_planes.p) rdo::h2_report(bad, dpre_planes.ovf);
    block_loss_add(acc, inv_npix * coef, loss_out, it, iter_pub);
}

// ---- GDN / IGDN epilogue + loss + gradient + dL/dnorm ----------------------------------------------------------------------------------
__device__ __forceinline__ float loss_gdn_quad(const f32x4& xv, const f32x4& nv, const f32x4& y, const f32x4* r, int inverse, float gs,
                                               f32x4& o, f32x4& g, f32x4& tv) {
    f32x4 rs;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rs[k] = __frsqrt_rn(nv[k]);
        o[k] = xv[k] * (inverse ? __fsqrt_rn(nv[k]) : rs[k]);       // the GDN / IGDN epilogue of the norm-pool conv
    }
    if (r) o += *r;
    const f32x4 dd = o - y;
    g = dd * gs;
#pragma unroll
    for (int k = 0; k < 4; ++k)       // GDN: y = x n^-1/2 -> dy/dn = -1/2 x n^-3/2 ; IGDN: y = x n^1/2 -> dy/dn = 1/2 x n^-1/2
        tv[k] = inverse ? (0.5f * g[k] * xv[k]) * rs[k] : (-0.5f * g[k] * xv[k]) * (rs[k] * rs[k] * rs[k]);
    return (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
}

__global__ __launch_bounds__(256) void loss_gdn_bwd_kernel(const float* x, const float* nrm, const float* res, const float* tgt,
                                                           const int32_t* idx_table, const int32_t* iter_ptr, int B, long per_image,
                                                           float inv_npix, float coef, int inverse, float* out, float* gout, float* tbuf,
                                                           float* loss_out, int32_t* iter_pub) {
    const int it = *iter_ptr;
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    const float gs = coef * 2.f * inv_npix;
    float acc = 0.f;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int b = (int)(q / quads);
        const long off = (q - (long)b * quads) * 4;
        const long e = (long)b * per_image + off;
        f32x4 r, o, g, tv;
        if (res) r = ldq(res + e);
        acc += loss_gdn_quad(ldq(x + e), ldq(nrm + e), ldq(tgt + (long)idx_table[(long)it * B + b] * per_image + off), res ? &r : nullptr,
                             inverse, gs, o, g, tv);
        if (out) stq(out + e, o);
        stq(gout + e, g);
        stq(tbuf + e, tv);
    }
    block_loss_add(acc, inv_npix * coef, loss_out, it, iter_pub);
}

__global__ __launch_bounds__(256) void loss_gdn_bwd_pix_kernel(const float* x, const float* nrm, const float* res, const float* tgt,
                                                               const int32_t* idx_table, const int32_t* iter_ptr, int B, long per_image,
                                                               int C, float inv_npix, float coef, int inverse, float* out, float* gout,
                                                               float* tbuf, H2Out t_planes, float* loss_out, int32_t* iter_pub) {
    int bad = 0;
    const int it = *iter_ptr;
    const long ppi = per_image / C, M = (long)B * ppi;
    const int ngroups = (C + 31) / 32;
    const long nvb = ((M + 63) / 64) * ngroups;
    const float gs = coef * 2.f * inv_npix;
    float acc = 0.f;
    for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        Oct oc;
        if (!oct_of(vb, ngroups, M, C, oc)) continue;
        const int b = (int)(oc.m / ppi);
        const float* yrow = tgt + ((long)idx_table[(long)it * B + b] - b) * per_image;
        f32x4 tv[2], o[2], g[2];
        const long e0 = oc.m * C + oc.c0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long e = e0 + k * Q2;
            f32x4 r;
            if (res) r = ldq(res + e);
            acc += loss_gdn_quad(ldq(x + e), ldq(nrm + e), ldq(yrow + e), res ? &r : nullptr, inverse, gs, o[k], g[k], tv[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (out) stq(out + e0 + k * Q2, o[k]);
            stq(gout + e0 + k * Q2, g[k]);
            if (tbuf) stq(tbuf + e0 + k * Q2, tv[k]);
        }
        store_h2_oct(t_planes, M, C, oc, tv[0], tv[1], bad);
    }
    rdo::h2_report(bad, t_planes.ovf);
    block_loss_add(acc, inv_npix * coef, loss_out, it, iter_pub);
}

// ---- GDN backward: dx -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 gdn_dx_quad(const f32x4& gv, const f32x4& xv, const f32x4& nv, const f32x4& av, int inverse) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float f = inverse ? __fsqrt_rn(nv[k]) : __frsqrt_rn(nv[k]);
        o[k] = gv[k] * f + 2.f * xv[k] * av[k];
    }
    return o;
}

__global__ __launch_bounds__(256) void gdn_bwd_dx_kernel(const float* g, const float* x, const float* nrm, const float* acc, long n4,
                                                         int inverse, float* dx) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const long e = q * 4;
        stq(dx + e, gdn_dx_quad(ldq(g + e), ldq(x + e), ldq(nrm + e), ldq(acc + e), inverse));
    }
}

__global__ __launch_bounds__(256) void gdn_bwd_dx_pix_kernel(const float* g, const float* x, const float* nrm, const float* acc, long M,
                                                             int C, int inverse, float* dx, H2Out dx_planes) {
    int bad = 0;
    const int ngroups = (C + 31) / 32;
    const long nvb = ((M + 63) / 64) * ngroups;
    for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        Oct oc;
        if (!oct_of(vb, ngroups, M, C, oc)) continue;
        f32x4 o[2];
        const long e0 = oc.m * C + oc.c0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            o[k] = gdn_dx_quad(ldq(g + e0 + k * Q2), ldq(x + e0 + k * Q2), ldq(nrm + e0 + k * Q2), ldq(acc + e0 + k * Q2), inverse);
        if (dx) {
            stq(dx + e0, o[0]);
            stq(dx + e0 + Q2, o[1]);
        }
        store_h2_oct(dx_planes, M, C, oc, o[0], o[1], bad);
    }
    rdo::h2_report(bad, dx_planes.ovf);
}

// ---- r = 2 pixel shuffle and its gradient --------------------------------------------------------------------------------------------------
// Channel c of large pixel (2h + dy, 2w + dx) is channel 4c + 2dy + dx of small pixel (h, w).  fp32-only: one thread per (small
// pixel, quad of LARGE channels): 16 consecutive channels of the small tensor (four 16-byte loads) hold four large channels x four
// sub-pixels; a 4 x 4 transpose in registers yields one 16-byte quad for each of the four large pixels.
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* x, long npix_small, int H, int W, int C, float* out) {
    const int cq = C / 4;
    const long total = npix_small * cq;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t % cq);
        const long p = t / cq;                             // small pixel (b, h, w)
        const int w = (int)(p % W);
        const long bh = p / W;                             // b * H + h
        const float* src = x + p * (4L * C) + 16 * q;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ldq(src + 4 * i);
        const long big0 = ((bh * 2) * (2L * W) + 2 * w) * C + 4 * q;      // large pixel (b, 2h, 2w)
        const long row = 2L * W * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) stq(out + big0 + (j >> 1) * row + (j & 1) * C, f32x4{v[0][j], v[1][j], v[2][j], v[3][j]});
    }
}

// with planes: pixel-major over the SMALL pixels; a thread owns 8 large channels = 32 consecutive small channels (128 bytes) and
// writes 8 channels of each of the four large pixels
__global__ __launch_bounds__(256) void pixel_shuffle2_pix_kernel(const float* x, long npix_small, int H, int W, int C, float* out,
                                                                 H2Out planes) {
    int bad = 0;
    const int ngroups = (C + 31) / 32;
    const long nvb = ((npix_small + 63) / 64) * ngroups;
    const long ML = npix_small * 4;
    for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        Oct oc;                                            // (small pixel, first of 8 large channels)
        if (!oct_of(vb, ngroups, npix_small, C, oc)) continue;
        const int w = (int)(oc.m % W);
        const long bh = oc.m / W;
        const float* src = x + oc.m * (4L * C) + 4 * oc.c0;
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ldq(src + 4 * k);
        const long big_pix = (bh * 2) * (2L * W) + 2 * w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = {v[0][j], v[1][j], v[2][j], v[3][j]}, b = {v[4][j], v[5][j], v[6][j], v[7][j]};
            const Oct big{big_pix + (j >> 1) * (2L * W) + (j & 1), oc.c0};
            if (out) {
                stq(out + big.m * C + oc.c0, a);
                stq(out + big.m * C + oc.c0 + 4, b);
            }
            store_h2_oct(planes, ML, C, big, a, b, bad);
        }
    }
    rdo::h2_report(bad, planes.ovf);
}

// gradient of the shuffle: [B,2H,2W,C] -> [B,H,W,4C]; fp32-only: one thread per (small pixel, quad of large channels): four 16-byte
// loads from the four large pixels, 4 x 4 transpose, four 16-byte stores (16 consecutive channels of the small pixel)
__global__ __launch_bounds__(256) void pixel_unshuffle2_kernel(const float* x, long npix_small, int H, int W, int C, float* out) {
    const int cq = C / 4;
    const long total = npix_small * cq;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t % cq);
        const long p = t / cq;
        const int w = (int)(p % W);
        const long bh = p / W;
        const long big0 = ((bh * 2) * (2L * W) + 2 * w) * C + 4 * q;
        const long row = 2L * W * C;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ldq(x + big0 + (j >> 1) * row + (j & 1) * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) stq(out + p * (4L * C) + 16 * q + 4 * i, f32x4{v[0][i], v[1][i], v[2][i], v[3][i]});
    }
}

// with planes: a virtual block covers 32 small pixels x 128 small channels (= 32 large channels, one 128-byte line of each of the
// four large pixels); a thread owns one 16-channel slice of one small pixel (= 4 large channels): four 16-byte loads, a complete
// 32-byte record per plane
__global__ __launch_bounds__(256) void pixel_unshuffle2_pix_kernel(const float* x, long npix_small, int H, int W, int C, float* out,
                                                                   H2Out planes) {
    int bad = 0;
    const int ngroups = (C + 31) / 32;
    const long nvb = ((npix_small + 31) / 32) * ngroups;
    const int C4 = 4 * C;
    for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        const int grp = (int)(vb % ngroups);
        const long p = (vb / ngroups) * 32 + (threadIdx.x >> 3);
        const int cl = grp * 32 + (threadIdx.x & 7) * 4;   // first of the 4 large channels
        if (p >= npix_small || cl >= C) continue;
        const int w = (int)(p % W);
        const long bh = p / W;
        const long big0 = ((bh * 2) * (2L * W) + 2 * w) * C + cl;
        const long row = 2L * W * C;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ldq(x + big0 + (j >> 1) * row + (j & 1) * C);
        const f32x4 o0 = {v[0][0], v[1][0], v[2][0], v[3][0]}, o1 = {v[0][1], v[1][1], v[2][1], v[3][1]};
        const f32x4 o2 = {v[0][2], v[1][2], v[2][2], v[3][2]}, o3 = {v[0][3], v[1][3], v[2][3], v[3][3]};
        if (out) {
            float* dst = out + p * (long)C4 + 4 * cl;
            stq(dst, o0); stq(dst + 4, o1); stq(dst + 8, o2); stq(dst + 12, o3);
        }
        store_h2_oct(planes, npix_small, C4, Oct{p, 4 * cl}, o0, o1, bad);
        store_h2_oct(planes, npix_small, C4, Oct{p, 4 * cl + 8}, o2, o3, bad);
    }
    rdo::h2_report(bad, planes.ovf);
}

}  // namespace

extern "C" {

int rdo_gather_qdrop_h2(const float* cache_q, const float* cache_fp, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                        int32_t batch_offset, int64_t per_image, int32_t C, float prob, uint32_t seed, float* out, void* out_planes,
                        float out_scale, int32_t* iter_publish, void* stream) {
    RDO_REQUIRE(cache_q && cache_fp && idx_table && iter_ptr && out_planes, "rdo_gather_qdrop_h2: null pointer");
    RDO_REQUIRE(B > 0 && batch_offset >= 0 && per_image > 0 && C > 0 && C % 16 == 0 && per_image % C == 0, "rdo_gather_qdrop_h2: bad shape");
    RDO_REQUIRE((long)(batch_offset + B) * per_image < (1L << 32), "rdo_gather_qdrop_h2: batch tensor exceeds the 32-bit RNG counter");
    RDO_REQUIRE(prob >= 0.f && prob <= 1.f, "rdo_gather_qdrop_h2: prob out of [0,1]");
    double t = floor((double)prob * 4294967296.0);
    const unsigned long long thr = (unsigned long long)(t > 4294967296.0 ? 4294967296.0 : t);
    RDO_REQUIRE(out_scale > 0.f, "rdo_gather_qdrop_h2: out_scale must be a positive power of two");
    const H2Out pl{reinterpret_cast<u16*>(out_planes), out_scale, rdo::h2_overflow_flag()};
    const Gather g{cache_q, cache_fp, idx_table, 0, B, batch_offset, (long)per_image, C, thr, seed, out, pl};
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gather_qdrop_h2_kernel, dim3(oct_grid((long)B * (per_image / C), C)), dim3(256), 0, s, g, iter_ptr, iter_publish);
            return rdo::check_launch("gather_qdrop_h2");
        },
        stream, "gather_qdrop_h2", 0.0, (8.0 + (out ? 4.0 : 0.0) + 4.0) * B * per_image);
}

int rdo_loss_act_bwd(const float* pre, const float* residual, const void* residual_planes, float residual_scale, const float* tgt_cache,
                     const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, int64_t per_image, int32_t C, float coef, int32_t act,
                     float* out, float* grad_out, float* dpre, void* dpre_planes, float dpre_scale, float* loss_out, void* stream) {
    RDO_REQUIRE(pre && tgt_cache && idx_table && iter_ptr && (dpre || dpre_planes || grad_out), "rdo_loss_act_bwd: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_loss_act_bwd: bad shape");
    RDO_REQUIRE((!dpre_planes && !residual_planes) || C % 16 == 0, "rdo_loss_act_bwd: H2 tensors need C % 16 == 0");
    RDO_REQUIRE((!dpre_planes || dpre_scale > 0.f) && (!residual_planes || residual_scale > 0.f), "rdo_loss_act_bwd: plane scales must be positive powers of two");
    const H2In rpl{residual ? nullptr : reinterpret_cast<const u16*>(residual_planes), residual_planes ? 1.f / residual_scale : 1.f};
    RDO_REQUIRE(act >= 0 && act <= 2, "rdo_loss_act_bwd: act %d (0 none, 1 LeakyReLU(0.01), 2 ReLU)", act);
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    const H2Out pl{reinterpret_cast<u16*>(dpre_planes), dpre_scale, rdo::h2_overflow_flag()};
    int32_t* const pub = rdo::take_iter_publish();
    const double n = (double)B * per_image;
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (pl.p || rpl.p)
                hipLaunchKernelGGL(loss_act_bwd_pix_kernel, dim3(loss_grid(oct_blocks((long)B * (per_image / C), C))), dim3(256), 0, s, pre, residual, rpl,
                                   tgt_cache, idx_table, iter_ptr, B, (long)per_image, C, inv_npix, coef, act, out, grad_out, dpre, pl,
                                   loss_out, pub);
            else
                hipLaunchKernelGGL(loss_act_bwd_kernel, dim3(loss_grid(rdo::ceil_div((long)B * per_image / 4, 256L))), dim3(256), 0, s, pre,
                                   (const float*)nullptr, 0, 0L, (const float*)nullptr, C, residual, tgt_cache, idx_table, iter_ptr, B,
                                   (long)per_image, inv_npix, coef, act, out, grad_out, dpre, loss_out, pub);
            return rdo::check_launch("loss_act_bwd");
        },
        stream, "loss_act_bwd", 0.0,
        n * (8.0 + 4.0 * ((residual != nullptr) + (out != nullptr) + (grad_out != nullptr) + (dpre != nullptr)) + (pl.p ? 4.0 : 0.0)));
}

int rdo_loss_act_bwd_splitk(const float* partial, int32_t ksplit, const float* bias, const float* residual, const float* tgt_cache,
                            const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, int64_t per_image, int32_t C, float coef,
                            int32_t act, float* out, float* grad_out, float* dpre, float* loss_out, void* stream) {
    RDO_REQUIRE(partial && ksplit >= 2 && tgt_cache && idx_table && iter_ptr && (dpre || grad_out), "rdo_loss_act_bwd_splitk: bad argument");
    RDO_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && per_image > 0 && per_image % C == 0, "rdo_loss_act_bwd_splitk: bad shape");
    RDO_REQUIRE(act >= 0 && act <= 2, "rdo_loss_act_bwd_splitk: act %d (0 none, 1 LeakyReLU(0.01), 2 ReLU)", act);
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    const long slab = (long)B * per_image;
    int32_t* const pub = rdo::take_iter_publish();
    const double n = (double)B * per_image;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(loss_act_bwd_kernel, dim3(loss_grid(rdo::ceil_div((long)B * per_image / 4, 256L))), dim3(256), 0, s,
                               (const float*)nullptr, partial, ksplit, slab, bias, C, residual, tgt_cache, idx_table, iter_ptr, B,
                               (long)per_image, inv_npix, coef, act, out, grad_out, dpre, loss_out, pub);
            return rdo::check_launch("loss_act_bwd_splitk");
        },
        stream, "loss_act_bwd", 0.0,
        n * (4.0 + 4.0 * ksplit + 4.0 * ((residual != nullptr) + (out != nullptr) + (grad_out != nullptr) + (dpre != nullptr))));
}

int rdo_loss_gdn_bwd(const float* x, const float* norm, const float* residual, const float* tgt_cache, const int32_t* idx_table,
                     const int32_t* iter_ptr, int32_t B, int64_t per_image, int32_t C, float coef, int32_t inverse, float* out,
                     float* grad_out, float* t, void* t_planes, float t_scale, float* loss_out, void* stream) {
    RDO_REQUIRE(x && norm && tgt_cache && idx_table && iter_ptr && grad_out && (t || t_planes), "rdo_loss_gdn_bwd: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_loss_gdn_bwd: bad shape");
    RDO_REQUIRE(!t_planes || C % 16 == 0, "rdo_loss_gdn_bwd: H2 output needs C % 16 == 0");
    RDO_REQUIRE(!t_planes || t_scale > 0.f, "rdo_loss_gdn_bwd: t_scale must be a positive power of two");
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    const H2Out pl{reinterpret_cast<u16*>(t_planes), t_scale, rdo::h2_overflow_flag()};
    int32_t* const pub = rdo::take_iter_publish();
    const double n = (double)B * per_image;
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (pl.p)
                hipLaunchKernelGGL(loss_gdn_bwd_pix_kernel, dim3(loss_grid(oct_blocks((long)B * (per_image / C), C))), dim3(256), 0, s, x, norm, residual,
                                   tgt_cache, idx_table, iter_ptr, B, (long)per_image, C, inv_npix, coef, inverse, out, grad_out, t, pl,
                                   loss_out, pub);
            else
                hipLaunchKernelGGL(loss_gdn_bwd_kernel, dim3(loss_grid(rdo::ceil_div((long)B * per_image / 4, 256L))), dim3(256), 0, s, x, norm, residual, tgt_cache,
                                   idx_table, iter_ptr, B, (long)per_image, inv_npix, coef, inverse, out, grad_out, t, loss_out, pub);
            return rdo::check_launch("loss_gdn_bwd");
        },
        stream, "loss_gdn_bwd", 0.0,
        n * (16.0 + 4.0 * ((residual != nullptr) + (out != nullptr) + (t != nullptr)) + (pl.p ? 4.0 : 0.0)));
}

int rdo_gdn_bwd_dx_h2(const float* g, const float* x, const float* norm, const float* acc, int64_t n, int32_t C, int32_t inverse,
                      float* dx, void* dx_planes, float dx_scale, void* stream) {
    RDO_REQUIRE(g && x && norm && acc && (dx || dx_planes) && n > 0 && n % 4 == 0 && C > 0 && n % C == 0, "rdo_gdn_bwd_dx_h2: bad argument");
    RDO_REQUIRE(!dx_planes || C % 16 == 0, "rdo_gdn_bwd_dx_h2: H2 output needs C % 16 == 0");
    RDO_REQUIRE(!dx_planes || dx_scale > 0.f, "rdo_gdn_bwd_dx_h2: dx_scale must be a positive power of two");
    const H2Out pl{reinterpret_cast<u16*>(dx_planes), dx_scale, rdo::h2_overflow_flag()};
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (pl.p)
                hipLaunchKernelGGL(gdn_bwd_dx_pix_kernel, dim3(oct_grid(n / C, C)), dim3(256), 0, s, g, x, norm, acc, (long)(n / C), C, inverse,
                                   dx, pl);
            else
                hipLaunchKernelGGL(gdn_bwd_dx_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, g, x, norm, acc, (long)(n / 4), inverse, dx);
            return rdo::check_launch("gdn_bwd_dx_h2");
        },
        stream, "gdn_bwd_dx", 0.0, (double)n * (16.0 + (dx ? 4.0 : 0.0) + (pl.p ? 4.0 : 0.0)));
}

int rdo_pixel_shuffle_h2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, float out_scale,
                         void* stream) {
    RDO_REQUIRE(x && (out || out_planes) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "rdo_pixel_shuffle_h2: bad argument");
    RDO_REQUIRE(!out_planes || C % 16 == 0, "rdo_pixel_shuffle_h2: H2 output needs C % 16 == 0");
    RDO_REQUIRE(!out_planes || out_scale > 0.f, "rdo_pixel_shuffle_h2: out_scale must be a positive power of two");
    const H2Out pl{reinterpret_cast<u16*>(out_planes), out_scale, rdo::h2_overflow_flag()};
    const long nps = (long)B * H * W, npl = nps * 4;
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (pl.p)
                hipLaunchKernelGGL(pixel_shuffle2_pix_kernel, dim3(oct_grid(nps, C)), dim3(256), 0, s, x, nps, H, W, C, out, pl);
            else
                hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for(nps * C / 4)), dim3(256), 0, s, x, nps, H, W, C, out);
            return rdo::check_launch("pixel_shuffle_h2");
        },
        stream, "pixel_shuffle", 0.0, (double)npl * C * (4.0 + (out ? 4.0 : 0.0) + (pl.p ? 4.0 : 0.0)));
}

int rdo_pixel_unshuffle2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, float out_scale,
                         void* stream) {
    RDO_REQUIRE(x && (out || out_planes) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "rdo_pixel_unshuffle2: bad argument");
    RDO_REQUIRE(!out_planes || out_scale > 0.f, "rdo_pixel_unshuffle2: out_scale must be a positive power of two");
    const long nps = (long)B * H * W;
    const H2Out pl{reinterpret_cast<u16*>(out_planes), out_scale, rdo::h2_overflow_flag()};
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (pl.p) {
                const long g = rdo::ceil_div(nps, 32L) * rdo::ceil_div(C, 32);
                hipLaunchKernelGGL(pixel_unshuffle2_pix_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, s, x, nps, H, W, C, out, pl);
            } else {
                hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid_for(nps * C / 4)), dim3(256), 0, s, x, nps, H, W, C, out);
            }
            return rdo::check_launch("pixel_unshuffle2");
        },
        stream, "pixel_shuffle", 0.0, (double)nps * C * 32.0);
}

}  // extern "C"
