// Fused "unit tail" kernels: everything between the last forward conv of a reconstruction unit and the first backward GEMM, in one
// HBM pass, plus the producers of "P3" tensors (exact three-way bf16 splits, planes [3][n]) that the split-bf16 GEMM kernels
// read by LDS-DMA (conv_fwd_x6p.hip).
//
//   rdo_gather_qdrop_p3   cached rows -> QDrop mix -> mini-batch as fp32 AND as planes                   (layer_opt.py:289-292)
//   rdo_loss_act_bwd      out = act(pre) + res ; d = out - tgt[idx] ; loss ; dL/dout ; dL/dpre (fp32 / planes)
//                         = conv epilogue activation + lp_loss forward/backward + activation backward     (layer_opt.py:133,150,303-306)
//   rdo_loss_gdn_bwd      out = x * norm^(-+1/2) + res ; loss ; dL/dout ; t = dL/dnorm (fp32 / planes)        (quant_layer.py:142-154)
//   rdo_gdn_bwd_dx_p3     dx = g * norm^(-+1/2) + 2 x acc as fp32 and planes
//   rdo_pixel_shuffle_p3  F.pixel_shuffle(r = 2) writing fp32 and planes
//
// The arithmetic of each stage is the one of the separate kernels it replaces (elementwise.hip, the conv epilogues), op for op,
// and the file is built with -ffp-contract=off like them: fused and unfused paths produce the same bits except for the order in
// which the loss partial sums are added.
#include "rdo_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

inline unsigned grid_for(long n) {
    long g = rdo::ceil_div(n, 256L);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16v2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// exact three-way split of a quad of channels [c0, c0 + 4) of pixel m -> one 8-byte store per plane, P3 slice-major layout
// (rdo_ptq_hip.h): element (m, c) of a tensor with M pixels and C channels at plane * M * C + ((c >> 4) * M + m) * 16 + (c & 15)
__device__ __forceinline__ void store_p3(u16* planes, long M, int C, long m, int c0, const f32x4& v) {
    const long pstride = M * C;
    const long e = ((long)(c0 >> 4) * M + m) * 16 + (c0 & 15);
    const unsigned h01 = cvt_pk(v[0], v[1]), h23 = cvt_pk(v[2], v[3]);
    const float r0 = v[0] - lo_f(h01), r1 = v[1] - hi_f(h01), r2 = v[2] - lo_f(h23), r3 = v[3] - hi_f(h23);
    const unsigned m01 = cvt_pk(r0, r1), m23 = cvt_pk(r2, r3);
    const float s0 = r0 - lo_f(m01), s1 = r1 - hi_f(m01), s2 = r2 - lo_f(m23), s3 = r3 - hi_f(m23);
    *reinterpret_cast<u32x2*>(planes + e) = u32x2{h01, h23};
    *reinterpret_cast<u32x2*>(planes + pstride + e) = u32x2{m01, m23};
    *reinterpret_cast<u32x2*>(planes + 2 * pstride + e) = u32x2{cvt_pk(s0, s1), cvt_pk(s2, s3)};
}
// element offset e of an NHWC tensor with C channels -> (pixel, first channel of the quad)
__device__ __forceinline__ void store_p3_at(u16* planes, long M, int C, long e, const f32x4& v) {
    const long m = e / C;
    store_p3(planes, M, C, m, (int)(e - m * C), v);
}

__device__ __forceinline__ void block_loss_add(float acc, float scale, float* loss_out, int it) {
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out)
        atomicAdd(loss_out + (long)it * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)), (red[0] + red[1] + red[2] + red[3]) * scale);
}

__global__ __launch_bounds__(256) void gather_qdrop_p3_kernel(const float* cq, const float* cfp, const int32_t* idx_table,
                                                              const int32_t* iter_ptr, int B, int batch_offset, long per_image, int C,
                                                              unsigned long long thr, uint32_t seed, float* out, u16* planes) {
    const int it = *iter_ptr;
    const uint32_t key = lowbias32((uint32_t)it + seed * 0x9E3779B9u);
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / quads);
        const long off = (t - (long)b * quads) * 4;
        const long src = (long)idx_table[(long)it * B + b] * per_image + off;
        const f32x4 q = *reinterpret_cast<const f32x4*>(cq + src);
        const f32x4 f = *reinterpret_cast<const f32x4*>(cfp + src);
        const uint32_t i0 = (uint32_t)((long)(batch_offset + b) * per_image + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((unsigned long long)lowbias32((i0 + e) ^ key) < thr) ? q[e] : f[e];
        const long dst = (long)b * per_image + off;
        if (out) *reinterpret_cast<f32x4*>(out + dst) = o;
        store_p3_at(planes, total * 4 / C, C, dst, o);
    }
}

// act: 0 none, 1 LeakyReLU(0.01), 2 ReLU
__global__ __launch_bounds__(256) void loss_act_bwd_kernel(const float* pre, const float* res, const float* tgt, const int32_t* idx_table,
                                                           const int32_t* iter_ptr, int B, long per_image, int C, float inv_npix, float coef,
                                                           int act, float* out, float* gout, float* dpre, u16* dpre_planes,
                                                           float* loss_out) {
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
        const f32x4 p = *reinterpret_cast<const f32x4*>(pre + e);
        const f32x4 y = *reinterpret_cast<const f32x4*>(tgt + (long)idx_table[(long)it * B + b] * per_image + off);
        f32x4 o = p;
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = p[k] > 0.f ? p[k] : slope * p[k];
        }
        if (res) o += *reinterpret_cast<const f32x4*>(res + e);
        if (out) *reinterpret_cast<f32x4*>(out + e) = o;
        const f32x4 dd = o - y;
        acc += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
        f32x4 g = dd * gs;
        if (gout) *reinterpret_cast<f32x4*>(gout + e) = g;
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = p[k] > 0.f ? g[k] : slope * g[k];
        }
        if (dpre) *reinterpret_cast<f32x4*>(dpre + e) = g;
        if (dpre_planes) store_p3_at(dpre_planes, total * 4 / C, C, e, g);
    }
    block_loss_add(acc, inv_npix * coef, loss_out, it);
}

__global__ __launch_bounds__(256) void loss_gdn_bwd_kernel(const float* x, const float* nrm, const float* res, const float* tgt,
                                                           const int32_t* idx_table, const int32_t* iter_ptr, int B, long per_image, int C,
                                                           float inv_npix, float coef, int inverse, float* out, float* gout, float* tbuf,
                                                           u16* t_planes, float* loss_out) {
    const int it = *iter_ptr;
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    const float gs = coef * 2.f * inv_npix;
    float acc = 0.f;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int b = (int)(q / quads);
        const long off = (q - (long)b * quads) * 4;
        const long e = (long)b * per_image + off;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + e);
        const f32x4 nv = *reinterpret_cast<const f32x4*>(nrm + e);
        const f32x4 y = *reinterpret_cast<const f32x4*>(tgt + (long)idx_table[(long)it * B + b] * per_image + off);
        f32x4 r, o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = __frsqrt_rn(nv[k]);
            o[k] = xv[k] * (inverse ? __fsqrt_rn(nv[k]) : r[k]);       // the GDN / IGDN epilogue of the norm-pool conv
        }
        if (res) o += *reinterpret_cast<const f32x4*>(res + e);
        if (out) *reinterpret_cast<f32x4*>(out + e) = o;
        const f32x4 dd = o - y;
        acc += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
        const f32x4 g = dd * gs;
        *reinterpret_cast<f32x4*>(gout + e) = g;
        f32x4 tv;
#pragma unroll
        for (int k = 0; k < 4; ++k)       // GDN: y = x n^-1/2 -> dy/dn = -1/2 x n^-3/2 ; IGDN: y = x n^1/2 -> dy/dn = 1/2 x n^-1/2
            tv[k] = inverse ? (0.5f * g[k] * xv[k]) * r[k] : (-0.5f * g[k] * xv[k]) * (r[k] * r[k] * r[k]);
        if (tbuf) *reinterpret_cast<f32x4*>(tbuf + e) = tv;
        if (t_planes) store_p3_at(t_planes, total * 4 / C, C, e, tv);
    }
    block_loss_add(acc, inv_npix * coef, loss_out, it);
}

__global__ __launch_bounds__(256) void gdn_bwd_dx_p3_kernel(const float* g, const float* x, const float* nrm, const float* acc, long n4,
                                                            int C, int inverse, float* dx, u16* dx_planes) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const long e = q * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e), xv = *reinterpret_cast<const f32x4*>(x + e);
        const f32x4 nv = *reinterpret_cast<const f32x4*>(nrm + e), av = *reinterpret_cast<const f32x4*>(acc + e);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = inverse ? __fsqrt_rn(nv[k]) : __frsqrt_rn(nv[k]);
            o[k] = gv[k] * f + 2.f * xv[k] * av[k];
        }
        if (dx) *reinterpret_cast<f32x4*>(dx + e) = o;
        if (dx_planes) store_p3_at(dx_planes, n4 * 4 / C, C, e, o);
    }
}

// r = 2 pixel shuffle, one thread per (small pixel, quad of LARGE channels): 16 consecutive channels of the small tensor (four 16-byte
// loads) hold four large channels x four sub-pixels; a 4 x 4 transpose in registers yields one 16-byte quad for each of the four
// large pixels (2h+dy, 2w+dx) -- channel c of that pixel is channel 4c + 2dy + dx of the small one -- written as fp32 and as planes
__global__ __launch_bounds__(256) void pixel_shuffle2_p3_kernel(const float* x, long npix_small, int H, int W, int C, float* out,
                                                                u16* planes) {
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
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + 4 * i);
        const long big0 = ((bh * 2) * (2L * W) + 2 * w) * C + 4 * q;      // large pixel (b, 2h, 2w)
        const long row = 2L * W * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 o = {v[0][j], v[1][j], v[2][j], v[3][j]};
            const long e = big0 + (j >> 1) * row + (j & 1) * C;
            if (out) *reinterpret_cast<f32x4*>(out + e) = o;
            if (planes) store_p3(planes, npix_small * 4, C, e / C, 4 * q, o);
        }
    }
}

// inverse (the gradient of the r = 2 pixel shuffle): [B,2H,2W,C] -> [B,H,W,4C]; one thread per (small pixel, quad of large channels):
// four 16-byte loads from the four large pixels, 4 x 4 transpose, four 16-byte stores (16 consecutive channels of the small pixel)
__global__ __launch_bounds__(256) void pixel_unshuffle2_kernel(const float* x, long npix_small, int H, int W, int C, float* out,
                                                               u16* planes) {
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
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(x + big0 + (j >> 1) * row + (j & 1) * C);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 o = {v[0][i], v[1][i], v[2][i], v[3][i]};
            if (out) *reinterpret_cast<f32x4*>(out + p * (4L * C) + 16 * q + 4 * i) = o;
            if (planes) store_p3(planes, npix_small, 4 * C, p, 16 * q + 4 * i, o);
        }
    }
}

}  // namespace

extern "C" {

int rdo_gather_qdrop_p3(const float* cache_q, const float* cache_fp, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                        int32_t batch_offset, int64_t per_image, int32_t C, float prob, uint32_t seed, float* out, void* out_planes,
                        void* stream) {
    RDO_REQUIRE(cache_q && cache_fp && idx_table && iter_ptr && out_planes, "rdo_gather_qdrop_p3: null pointer");
    RDO_REQUIRE(B > 0 && batch_offset >= 0 && per_image > 0 && C > 0 && C % 16 == 0 && per_image % C == 0, "rdo_gather_qdrop_p3: bad shape");
    RDO_REQUIRE((long)(batch_offset + B) * per_image < (1L << 32), "rdo_gather_qdrop_p3: batch tensor exceeds the 32-bit RNG counter");
    RDO_REQUIRE(prob >= 0.f && prob <= 1.f, "rdo_gather_qdrop_p3: prob out of [0,1]");
    double t = floor((double)prob * 4294967296.0);
    const unsigned long long thr = (unsigned long long)(t > 4294967296.0 ? 4294967296.0 : t);
    u16* pl = reinterpret_cast<u16*>(out_planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gather_qdrop_p3_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, cache_q, cache_fp,
                               idx_table, iter_ptr, B, batch_offset, (long)per_image, C, thr, seed, out, pl);
            return rdo::check_launch("gather_qdrop_p3");
        },
        stream, "gather_qdrop_p3", 0.0, (8.0 + (out ? 4.0 : 0.0) + 6.0) * B * per_image);
}

int rdo_loss_act_bwd(const float* pre, const float* residual, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr,
                     int32_t B, int64_t per_image, int32_t C, float coef, int32_t act, float* out, float* grad_out, float* dpre,
                     void* dpre_planes, float* loss_out, void* stream) {
    RDO_REQUIRE(pre && tgt_cache && idx_table && iter_ptr && (dpre || dpre_planes || grad_out), "rdo_loss_act_bwd: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_loss_act_bwd: bad shape");
    RDO_REQUIRE(!dpre_planes || C % 16 == 0, "rdo_loss_act_bwd: P3 output needs C % 16 == 0");
    RDO_REQUIRE(act >= 0 && act <= 2, "rdo_loss_act_bwd: act %d (0 none, 1 LeakyReLU(0.01), 2 ReLU)", act);
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    u16* pl = reinterpret_cast<u16*>(dpre_planes);
    const double n = (double)B * per_image;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(loss_act_bwd_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, pre, residual, tgt_cache,
                               idx_table, iter_ptr, B, (long)per_image, C, inv_npix, coef, act, out, grad_out, dpre, pl, loss_out);
            return rdo::check_launch("loss_act_bwd");
        },
        stream, "loss_act_bwd", 0.0,
        n * (8.0 + 4.0 * ((residual != nullptr) + (out != nullptr) + (grad_out != nullptr) + (dpre != nullptr)) + (pl ? 6.0 : 0.0)));
}

int rdo_loss_gdn_bwd(const float* x, const float* norm, const float* residual, const float* tgt_cache, const int32_t* idx_table,
                     const int32_t* iter_ptr, int32_t B, int64_t per_image, int32_t C, float coef, int32_t inverse, float* out,
                     float* grad_out, float* t, void* t_planes, float* loss_out, void* stream) {
    RDO_REQUIRE(x && norm && tgt_cache && idx_table && iter_ptr && grad_out && (t || t_planes), "rdo_loss_gdn_bwd: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_loss_gdn_bwd: bad shape");
    RDO_REQUIRE(!t_planes || C % 16 == 0, "rdo_loss_gdn_bwd: P3 output needs C % 16 == 0");
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    u16* pl = reinterpret_cast<u16*>(t_planes);
    const double n = (double)B * per_image;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(loss_gdn_bwd_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, x, norm, residual, tgt_cache,
                               idx_table, iter_ptr, B, (long)per_image, C, inv_npix, coef, inverse, out, grad_out, t, pl, loss_out);
            return rdo::check_launch("loss_gdn_bwd");
        },
        stream, "loss_gdn_bwd", 0.0,
        n * (16.0 + 4.0 * ((residual != nullptr) + (out != nullptr) + (t != nullptr)) + (pl ? 6.0 : 0.0)));
}

int rdo_gdn_bwd_dx_p3(const float* g, const float* x, const float* norm, const float* acc, int64_t n, int32_t C, int32_t inverse,
                      float* dx, void* dx_planes, void* stream) {
    RDO_REQUIRE(g && x && norm && acc && (dx || dx_planes) && n > 0 && n % 4 == 0 && C > 0 && n % C == 0, "rdo_gdn_bwd_dx_p3: bad argument");
    RDO_REQUIRE(!dx_planes || C % 16 == 0, "rdo_gdn_bwd_dx_p3: P3 output needs C % 16 == 0");
    u16* pl = reinterpret_cast<u16*>(dx_planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gdn_bwd_dx_p3_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, g, x, norm, acc, (long)(n / 4), C, inverse, dx, pl);
            return rdo::check_launch("gdn_bwd_dx_p3");
        },
        stream, "gdn_bwd_dx", 0.0, (double)n * (16.0 + (dx ? 4.0 : 0.0) + (pl ? 6.0 : 0.0)));
}

int rdo_pixel_shuffle_p3(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, void* stream) {
    RDO_REQUIRE(x && (out || out_planes) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "rdo_pixel_shuffle_p3: bad argument");
    RDO_REQUIRE(!out_planes || C % 16 == 0, "rdo_pixel_shuffle_p3: P3 output needs C % 16 == 0");
    u16* pl = reinterpret_cast<u16*>(out_planes);
    const long nps = (long)B * H * W, npl = nps * 4;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(pixel_shuffle2_p3_kernel, dim3(grid_for(nps * C / 4)), dim3(256), 0, s, x, nps, H, W, C, out, pl);
            return rdo::check_launch("pixel_shuffle_p3");
        },
        stream, "pixel_shuffle", 0.0, (double)npl * C * (4.0 + (out ? 4.0 : 0.0) + (pl ? 6.0 : 0.0)));
}

int rdo_pixel_unshuffle2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, void* stream) {
    RDO_REQUIRE(x && (out || out_planes) && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "rdo_pixel_unshuffle2: bad argument");
    const long nps = (long)B * H * W;
    u16* pl = reinterpret_cast<u16*>(out_planes);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid_for(nps * C / 4)), dim3(256), 0, s, x, nps, H, W, C, out, pl);
            return rdo::check_launch("pixel_unshuffle2");
        },
        stream, "pixel_shuffle", 0.0, (double)nps * C * 32.0);
}

}  // extern "C"
