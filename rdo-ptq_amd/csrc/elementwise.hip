// K6 / K7 / K8(lp) / K9 -- HBM-bound element-wise stages of the calibration loop on NHWC fp32 tensors.
// 16-byte (float4) accesses, grid-stride loops capped at ~2048 workgroups (MI355X: 256 CUs x 8).
// Built with -ffp-contract=off (products and sums round separately, like the reference's op chains).
#include "rdo_common.h"
#include "gather_body.h"

namespace {

using rdo::gq::f32x4;

inline unsigned grid_for(long n, int per_thread = 1) {
    long g = rdo::ceil_div(n, 256L * per_thread);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

using rdo::gq::lowbias32;

// ---- K7: gather + QDrop (layer_opt.py:289-292; gather_body.h).  keep = u32(seed, iter, i) < floor(p * 2^32) takes the quantised-prefix input.
__global__ __launch_bounds__(256) void gather_qdrop_kernel(rdo::gq::Gather g, const int32_t* iter_ptr, int32_t* iter_publish) {
    const int it = *iter_ptr;
    if (iter_publish && blockIdx.x == 0 && threadIdx.x == 0) *iter_publish = it;      // rdo_ptq_hip.h: iteration-counter hand-over
    rdo::gq::gather_fp32_body(g, it, blockIdx.x, gridDim.x);
}

// ---- K8: p=2 lp_loss forward + gradient against the cached FP target rows
__global__ __launch_bounds__(256) void lp2_kernel(const float* pred, const float* tgt, const int32_t* idx_table,
                                                  const int32_t* iter_ptr, int B, long per_image, float inv_npix, float coef,
                                                  float* grad, float* loss_out, int32_t* iter_pub) {
    const int it = *iter_ptr;
    if (iter_pub && blockIdx.x == 0 && threadIdx.x == 0) *iter_pub = it;       // rdo_iter_bind_publish
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    float acc = 0.f;
    const float gs = coef * 2.f * inv_npix;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / quads);
        const long off = (t - (long)b * quads) * 4;
        const f32x4 p = *reinterpret_cast<const f32x4*>(pred + (long)b * per_image + off);
        const f32x4 y = *reinterpret_cast<const f32x4*>(tgt + (long)idx_table[(long)it * B + b] * per_image + off);
        const f32x4 dd = p - y;
        acc += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
        *reinterpret_cast<f32x4*>(grad + (long)b * per_image + off) = dd * gs;
    }
    __shared__ float red[4];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    // 32 accumulation slots per iteration: same-address float atomics serialise at the memory side (~20 ns each), so a
    // 2048-block launch would spend tens of microseconds on them; the reader sums the slots
    if (threadIdx.x == 0 && loss_out)
        atomicAdd(loss_out + (long)it * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1)), (red[0] + red[1] + red[2] + red[3]) * inv_npix * coef);
}

// general exponent: loss = coef2 * sum d^2 + coefp * sum |d|^p, both over the same (pred, tgt) pair -- rec_loss (p = 2) and the
// task term with main2.py's --task_loss exponent (layer_opt.py:133,150) in one pass
__global__ __launch_bounds__(256) void lp_kernel(const float* pred, const float* tgt, const int32_t* idx_table,
                                                 const int32_t* iter_ptr, int B, long per_image, float inv_npix, float coef2,
                                                 float coefp, float pw, float* grad, float* loss_out, float* loss_out_p, int32_t* iter_pub) {
    const int it = *iter_ptr;
    if (iter_pub && blockIdx.x == 0 && threadIdx.x == 0) *iter_pub = it;
    const long quads = per_image / 4;
    const long total = (long)B * quads;
    float acc = 0.f, accp = 0.f;                                 // the p = 2 term and the |d|^p term, logged apart on request
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / quads);
        const long off = (t - (long)b * quads) * 4;
        const f32x4 p = *reinterpret_cast<const f32x4*>(pred + (long)b * per_image + off);
        const f32x4 y = *reinterpret_cast<const f32x4*>(tgt + (long)idx_table[(long)it * B + b] * per_image + off);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = p[e] - y[e], a = fabsf(d);
            const float am1 = powf(a, pw - 1.f);                 // |d|^(p-1); pow(0, 0) = 1 and the sign factor below is 0
            acc += coef2 * d * d;
            accp += coefp * am1 * a;
            const float sg = (float)((d > 0.f) - (d < 0.f));
            g[e] = (coef2 * 2.f * d + coefp * pw * am1 * sg) * inv_npix;
        }
        *reinterpret_cast<f32x4*>(grad + (long)b * per_image + off) = g;
    }
    __shared__ float red[8];
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        accp += __shfl_down(accp, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; red[4 + (threadIdx.x >> 6)] = accp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s2 = (red[0] + red[1] + red[2] + red[3]) * inv_npix, sp = (red[4] + red[5] + red[6] + red[7]) * inv_npix;
        const long slot = (long)it * RDO_LOG_SLOTS + (blockIdx.x & (RDO_LOG_SLOTS - 1));
        if (loss_out_p) {
            atomicAdd(loss_out_p + slot, sp);
            if (loss_out) atomicAdd(loss_out + slot, s2);
        } else if (loss_out) {
            atomicAdd(loss_out + slot, s2 + sp);
        }
    }
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* g, const float* y, long n4, float slope, float* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x) {
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[t], yv = reinterpret_cast<const f32x4*>(y)[t];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? gv[e] : slope * gv[e];
        reinterpret_cast<f32x4*>(out)[t] = o;
    }
}

__global__ __launch_bounds__(256) void lrelu_fwd_kernel(const float* x, long n4, float slope, float* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[t];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : slope * v[e];
        reinterpret_cast<f32x4*>(out)[t] = o;
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, long n4, float* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[t] = reinterpret_cast<const f32x4*>(a)[t] + reinterpret_cast<const f32x4*>(b)[t];
}

// pixel shuffle on NHWC: out[b][h*r+dy][w*r+dx][c] = in[b][h][w][c*r*r + dy*r + dx]   (inverse: the other way round)
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float* x, int B, int H, int W, int C, int r, int inverse,
                                                            float* out) {
    // C = channels of the SHUFFLED (large) tensor; small tensor has C*r*r channels at HxW
    const long total = (long)B * H * r * W * r * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long u = t;
        const int c = (int)(u % C); u /= C;
        const int wo = (int)(u % (W * r)); u /= (W * r);
        const int ho = (int)(u % (H * r));
        const int b = (int)(u / (H * r));
        const int h = ho / r, dy = ho - h * r, w = wo / r, dx = wo - w * r;
        const long small = (((long)b * H + h) * W + w) * ((long)C * r * r) + (long)c * r * r + dy * r + dx;
        if (inverse) out[small] = x[t]; else out[t] = x[small];
    }
}

// r == 2: one thread per (small pixel, channel c): the four sub-pixels of c are one aligned float4 of the small tensor, and each
// of them lands in its own output pixel with lanes running over c -- 16-byte accesses on the small side, 256-byte runs on the large
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* x, long npix_small, int H, int W, int C, int inverse,
                                                             float* out) {
    const long total = npix_small * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long p = t / C;                              // small pixel (b, h, w)
        const int w = (int)(p % W);
        const long bh = p / W;                             // b * H + h
        const long big0 = ((bh * 2) * (2L * W) + 2 * w) * C + c;      // large pixel (b, 2h, 2w)
        const long row = 2L * W * C;
        if (inverse) {
            f32x4 v;
            v[0] = x[big0]; v[1] = x[big0 + C]; v[2] = x[big0 + row]; v[3] = x[big0 + row + C];
            reinterpret_cast<f32x4*>(out)[t] = v;
        } else {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[t];
            out[big0] = v[0]; out[big0 + C] = v[1]; out[big0 + row] = v[2]; out[big0 + row + C] = v[3];
        }
    }
}

__global__ __launch_bounds__(256) void gdn_bwd_t_kernel(const float* g, const float* x, const float* nrm, long n, int inverse,
                                                        float* t) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float nv = nrm[i];
        // GDN: y = x n^-1/2 -> dy/dn = -1/2 x n^-3/2 ; IGDN: y = x n^1/2 -> dy/dn = 1/2 x n^-1/2
        const float r = __frsqrt_rn(nv);
        t[i] = inverse ? (0.5f * g[i] * x[i]) * r : (-0.5f * g[i] * x[i]) * (r * r * r);
    }
}

__global__ __launch_bounds__(256) void gdn_bwd_dx_kernel(const float* g, const float* x, const float* nrm, const float* acc,
                                                         long n, int inverse, float* dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float nv = nrm[i];
        const float f = inverse ? __fsqrt_rn(nv) : __frsqrt_rn(nv);
        dx[i] = g[i] * f + 2.f * x[i] * acc[i];
    }
}

__global__ __launch_bounds__(256) void nchw_nhwc_kernel(const float* x, int B, int C, int H, int W, int inverse, float* out) {
    const long total = (long)B * C * H * W;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long u = t;  // t indexes NHWC
        const int c = (int)(u % C); u /= C;
        const int w = (int)(u % W); u /= W;
        const int h = (int)(u % H);
        const int b = (int)(u / H);
        const long nchw = (((long)b * C + c) * H + h) * W + w;
        if (inverse) out[nchw] = x[t]; else out[t] = x[nchw];
    }
}

__global__ void iter_advance_kernel(int32_t* it) { *it += 1; }

// zero-insertion + padding for transposed convolution: out[b][pt + h*s][pl + w*s][c] = x[b][h][w][c], zeros elsewhere
__global__ __launch_bounds__(256) void zero_insert_kernel(const float* x, int B, int H, int W, int C, int s, int pt, int pl,
                                                          int Ho, int Wo, float* out) {
    const long total = (long)B * Ho * Wo * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long u = t;
        const int c = (int)(u % C); u /= C;
        const int wo = (int)(u % Wo); u /= Wo;
        const int ho = (int)(u % Ho);
        const int b = (int)(u / Ho);
        const int hh = ho - pt, ww = wo - pl;
        float v = 0.f;
        if (hh >= 0 && ww >= 0 && hh % s == 0 && ww % s == 0) {
            const int h = hh / s, w = ww / s;
            if (h < H && w < W) v = x[(((long)b * H + h) * W + w) * C + c];
        }
        out[t] = v;
    }
}

// LayerNorm over the last dimension (one wave per row): y = (x - mean) * rsqrt(var + eps) * w + b      F.layer_norm, quant_layer.py:121
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* x, const float* w, const float* b, long rows, int C,
                                                         float eps, float* out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s1 = 0.f;
    for (int c = lane; c < C; c += 64) s1 += xr[c];
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / (float)C;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; s2 += d * d; }
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = rsqrtf(s2 / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        float v = (xr[c] - mean) * rstd;
        if (w) v *= w[c];
        if (b) v += b[c];
        out[row * C + c] = v;
    }
}

// ---- K6: per-channel dynamic activation quantisation (NHWC: channel = fastest dim)
// Two-level reduction, no atomics: up to kAqBlocks workgroups each leave the min / max of their share of the pixels per channel
// ([block][2][C] behind the 2 C result floats of the workspace), a small second kernel folds the blocks, the third applies.  (The first
// version let 1024 workgroups atomicMin / atomicMax into the same 2 C words: 0.4 M contended atomics per call were most of its time.)
constexpr int kAqBlocks = 256;

// thread = (pixel lane, group of W channels): W-wide loads down the pixels, four in flight, then an LDS fold over the pixel lanes
template <int W>
__global__ __launch_bounds__(256) void aq_partial_kernel(const float* x, long npix, int C, float* part) {
    typedef float vec_t __attribute__((ext_vector_type(W)));
    const int QN = C / W;
    const int qpb = QN < 256 ? QN : 256;
    const int PL = 256 / qpb;
    const int pl = threadIdx.x / qpb, ql = threadIdx.x - pl * qpb;
    __shared__ float smn[256 * W], smx[256 * W];
    const long step = (long)gridDim.x * PL;
    for (int qb = 0; qb < QN; qb += qpb) {
        const int q = qb + ql;
        const bool live = pl < PL && q < QN;
        vec_t mn, mx;
#pragma unroll
        for (int k = 0; k < W; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
        if (live) {
            const float* src = x + (long)q * W;
            long p = (long)blockIdx.x * PL + pl;
            for (; p + 3 * step < npix; p += 4 * step) {
                vec_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vec_t*>(src + (p + u * step) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < W; ++k) { mn[k] = fminf(mn[k], v[u][k]); mx[k] = fmaxf(mx[k], v[u][k]); }
            }
            for (; p < npix; p += step) {
                const vec_t v = *reinterpret_cast<const vec_t*>(src + p * C);
#pragma unroll
                for (int k = 0; k < W; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
            }
#pragma unroll
            for (int k = 0; k < W; ++k) { smn[(pl * qpb + ql) * W + k] = mn[k]; smx[(pl * qpb + ql) * W + k] = mx[k]; }
        }
        __syncthreads();
        if (live && pl == 0) {
            for (int r = 1; r < PL; ++r)
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    mn[k] = fminf(mn[k], smn[(r * qpb + ql) * W + k]);
                    mx[k] = fmaxf(mx[k], smx[(r * qpb + ql) * W + k]);
                }
            float* dst = part + (long)blockIdx.x * 2 * C + (long)q * W;
#pragma unroll
            for (int k = 0; k < W; ++k) { dst[k] = mn[k]; dst[C + k] = mx[k]; }
        }
        __syncthreads();
    }
}

// ws[0 .. C) = min, ws[C .. 2C) = max over the `nblk` partial rows: sixteen lanes per entry walk the rows (independent loads, no
// serial chain of nblk round trips), then fold across the lanes
__global__ __launch_bounds__(256) void aq_fold_kernel(const float* part, int nblk, int C, float* ws) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    const bool live = i < 2 * C;
    const bool is_max = i >= C;
    float r = is_max ? -INFINITY : INFINITY;
    if (live) {
        const float* src = part + i;
        int b = j;
        for (; b + 48 < nblk; b += 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(long)(b + 16 * u) * 2 * C];
#pragma unroll
            for (int u = 0; u < 4; ++u) r = is_max ? fmaxf(r, v[u]) : fminf(r, v[u]);
        }
        for (; b < nblk; b += 16) {
            const float v = src[(long)b * 2 * C];
            r = is_max ? fmaxf(r, v) : fminf(r, v);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float v = __shfl_xor(r, o, 16);
        r = is_max ? fmaxf(r, v) : fminf(r, v);
    }
    if (live && j == 0) ws[i] = r;
}

template <int W>
__global__ __launch_bounds__(256) void aq_apply_kernel(const float* x, long nvec, int C, const float* ws, float bit_range, float* out) {
    typedef float vec_t __attribute__((ext_vector_type(W)));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * W) % C);
        const vec_t xv = *reinterpret_cast<const vec_t*>(x + i * W);
        vec_t o;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float zp = ws[c + k];
            const float rng = fmaxf(ws[C + c + k] - zp, 1e-6f);
            const float xn = xv[k] - zp;
            const float q = rintf(fminf(fmaxf(xn / rng, -1.f), 1.f) * bit_range);
            o[k] = (q / bit_range) * rng + zp;
        }
        *reinterpret_cast<vec_t*>(out + i * W) = o;
    }
}

// ---- transposed conv as a sub-pixel conv: phase weights ----------------------------------------------------------------------------
// ConvTranspose2d(k, stride s, padding p) with output size s * H is a stride-1 conv with s^2 * Cout output channels followed by a
// pixel shuffle: output pixel (s i + a, s j + b) only ever meets the taps kh = (a + p) mod s + s t, and those taps read input rows
// i + q - t -- a K' x K' correlation window around pixel (i, j) whose entries are single taps of the kernel or zero (k = 5, s = 2: 36
// window entries for 25 taps, against the 100 multiply-adds per input pixel of the zero-insertion form).  `map` [s^2][K'][K'] holds the
// tap index (in the engine's kernel-layout weight [Cout][K][K][Cin]) of every window entry, -1 for a structural zero.
__global__ __launch_bounds__(256) void tconv_expand_kernel(const float* w, const int32_t* map, long total, int S2, int KK2, int taps, int Cin,
                                                          float* wp) {
    // wp[(co * S2 + ph)][u][v][ci]
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % Cin);
        long t = e / Cin;
        const int uv = (int)(t % KK2);
        t /= KK2;
        const int ph = (int)(t % S2);
        const long co = t / S2;
        const int tap = map[ph * KK2 + uv];
        wp[e] = tap >= 0 ? w[(co * taps + tap) * Cin + ci] : 0.f;
    }
}
// gradient slabs of the phase weight -> slabs of the kernel weight: every tap sits at exactly one window entry (`inv` [taps] = its
// offset (ph * K'^2 + u * K' + v))
__global__ __launch_bounds__(256) void tconv_fold_kernel(const float* sp, const int32_t* inv, long total, long per_slab, int S2, int KK2, int taps,
                                                        int Cin, float* out) {
    const long per_slab_p = per_slab / taps * ((long)S2 * KK2);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long slab = e / per_slab, r = e - slab * per_slab;
        const int ci = (int)(r % Cin);
        long t = r / Cin;
        const int tap = (int)(t % taps);
        const long co = t / taps;
        const int o = inv[tap];
        const int ph = o / KK2, uv = o - ph * KK2;
        out[e] = sp[slab * per_slab_p + (((co * S2 + ph) * KK2 + uv) * Cin + ci)];
    }
}

}  // namespace

extern "C" {

int rdo_gather_qdrop(const float* cache_q, const float* cache_fp, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                     int32_t batch_offset, int64_t per_image, float prob, uint32_t seed, float* out, int32_t* iter_publish, void* stream) {
    RDO_REQUIRE(cache_q && cache_fp && idx_table && iter_ptr && out, "rdo_gather_qdrop: null pointer");
    RDO_REQUIRE(batch_offset >= 0, "rdo_gather_qdrop: negative batch_offset");
    RDO_REQUIRE(B > 0 && per_image > 0 && per_image % 4 == 0, "rdo_gather_qdrop: per_image (%ld) must be a multiple of 4",
                (long)per_image);
    RDO_REQUIRE((long)(batch_offset + B) * per_image < (1L << 32), "rdo_gather_qdrop: batch tensor exceeds the 32-bit RNG counter");
    RDO_REQUIRE(prob >= 0.f && prob <= 1.f, "rdo_gather_qdrop: prob out of [0,1]");
    double t = floor((double)prob * 4294967296.0);
    const unsigned long long thr = (unsigned long long)(t > 4294967296.0 ? 4294967296.0 : t);
    const rdo::gq::Gather g{cache_q, cache_fp, idx_table, 0, B, batch_offset, (long)per_image, 0, thr, seed, out, rdo::gq::H2Out{nullptr, 0.f, nullptr}};
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gather_qdrop_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, g, iter_ptr, iter_publish);
            return rdo::check_launch("gather_qdrop");
        },
        stream, "gather_qdrop", 0.0, 12.0 * B * per_image);
}

int rdo_lp2_loss_grad(const float* pred, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                      int64_t per_image, int32_t C, float coef, float* grad, float* loss_out, void* stream) {
    RDO_REQUIRE(pred && tgt_cache && idx_table && iter_ptr && grad, "rdo_lp2_loss_grad: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_lp2_loss_grad: bad shape");
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    int32_t* const pub = rdo::take_iter_publish();
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(lp2_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, pred, tgt_cache, idx_table,
                               iter_ptr, B, (long)per_image, inv_npix, coef, grad, loss_out, pub);
            return rdo::check_launch("lp2_loss_grad");
        },
        stream, "lp2_loss_grad", 0.0, 12.0 * B * per_image);
}

int rdo_lp_loss_grad(const float* pred, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                     int64_t per_image, int32_t C, float coef2, float coefp, float p, float* grad, float* loss_out,
                     float* loss_out_p, void* stream) {
    RDO_REQUIRE(pred && tgt_cache && idx_table && iter_ptr && grad, "rdo_lp_loss_grad: null pointer");
    RDO_REQUIRE(B > 0 && C > 0 && per_image > 0 && per_image % 4 == 0 && per_image % C == 0, "rdo_lp_loss_grad: bad shape");
    RDO_REQUIRE(p >= 1.f, "rdo_lp_loss_grad: exponent %g < 1 has no finite gradient at zero", (double)p);
    const float inv_npix = (float)(1.0 / ((double)B * (double)(per_image / C)));
    int32_t* const pub = rdo::take_iter_publish();
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(lp_kernel, dim3(grid_for((long)B * per_image / 4)), dim3(256), 0, s, pred, tgt_cache, idx_table,
                               iter_ptr, B, (long)per_image, inv_npix, coef2, coefp, p, grad, loss_out, loss_out_p, pub);
            return rdo::check_launch("lp_loss_grad");
        },
        stream, "lp_loss_grad", 0.0, 12.0 * B * per_image);
}

static int act_bwd(const float* g, const float* y, int64_t n, float slope, float* out, void* stream, const char* tag) {
    RDO_REQUIRE(g && y && out && n > 0 && n % 4 == 0, "%s: bad argument", tag);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, g, y, (long)(n / 4), slope, out);
            return rdo::check_launch(tag);
        },
        stream);
}

static int act_fwd(const float* x, int64_t n, float slope, float* out, void* stream, const char* tag) {
    RDO_REQUIRE(x && out && n > 0 && n % 4 == 0, "%s: bad argument", tag);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, (long)(n / 4), slope, out);
            return rdo::check_launch(tag);
        },
        stream);
}

int rdo_lrelu_bwd(const float* g, const float* y, int64_t n, float* out, void* stream) { return act_bwd(g, y, n, 0.01f, out, stream, "rdo_lrelu_bwd"); }
int rdo_lrelu_fwd(const float* x, int64_t n, float* out, void* stream) { return act_fwd(x, n, 0.01f, out, stream, "rdo_lrelu_fwd"); }
int rdo_relu_bwd(const float* g, const float* y, int64_t n, float* out, void* stream) { return act_bwd(g, y, n, 0.f, out, stream, "rdo_relu_bwd"); }
int rdo_relu_fwd(const float* x, int64_t n, float* out, void* stream) { return act_fwd(x, n, 0.f, out, stream, "rdo_relu_fwd"); }

int rdo_add(const float* a, const float* b, int64_t n, float* out, void* stream) {
    RDO_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "rdo_add: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, a, b, (long)(n / 4), out);
            return rdo::check_launch("add");
        },
        stream);
}

int rdo_pixel_shuffle(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t r, int32_t inverse, float* out,
                      void* stream) {
    RDO_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && r > 0, "rdo_pixel_shuffle: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (r == 2 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 == 0)
                hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, s, x, (long)B * H * W, H, W,
                                   C, inverse, out);
            else
                hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for((long)B * H * W * C * r * r)), dim3(256), 0, s, x, B, H, W, C, r,
                                   inverse, out);
            return rdo::check_launch("pixel_shuffle");
        },
        stream);
}

int rdo_gdn_bwd_t(const float* g, const float* x, const float* norm, int64_t n, int32_t inverse, float* t, void* stream) {
    RDO_REQUIRE(g && x && norm && t && n > 0, "rdo_gdn_bwd_t: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gdn_bwd_t_kernel, dim3(grid_for(n)), dim3(256), 0, s, g, x, norm, (long)n, inverse, t);
            return rdo::check_launch("gdn_bwd_t");
        },
        stream);
}

int rdo_gdn_bwd_dx(const float* g, const float* x, const float* norm, const float* acc, int64_t n, int32_t inverse, float* dx,
                   void* stream) {
    RDO_REQUIRE(g && x && norm && acc && dx && n > 0, "rdo_gdn_bwd_dx: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gdn_bwd_dx_kernel, dim3(grid_for(n)), dim3(256), 0, s, g, x, norm, acc, (long)n, inverse, dx);
            return rdo::check_launch("gdn_bwd_dx");
        },
        stream);
}

int rdo_nchw_to_nhwc(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t inverse, float* out, void* stream) {
    RDO_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "rdo_nchw_to_nhwc: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(nchw_nhwc_kernel, dim3(grid_for((long)B * C * H * W)), dim3(256), 0, s, x, B, C, H, W, inverse, out);
            return rdo::check_launch("nchw_to_nhwc");
        },
        stream);
}

int rdo_zero_insert(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, int32_t pad_top, int32_t pad_left,
                    int32_t Ho, int32_t Wo, float* out, void* stream) {
    RDO_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && stride > 0 && pad_top >= 0 && pad_left >= 0 &&
                    Ho >= pad_top + (H - 1) * stride + 1 && Wo >= pad_left + (W - 1) * stride + 1,
                "rdo_zero_insert: bad geometry");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(zero_insert_kernel, dim3(grid_for((long)B * Ho * Wo * C)), dim3(256), 0, s, x, B, H, W, C, stride,
                               pad_top, pad_left, Ho, Wo, out);
            return rdo::check_launch("zero_insert");
        },
        stream);
}

int rdo_tconv_expand(const float* w, const int32_t* map, int32_t Cout, int32_t taps, int32_t Cin, int32_t S2, int32_t KK2, float* wp,
                     void* stream) {
    RDO_REQUIRE(w && map && wp && Cout > 0 && taps > 0 && Cin > 0 && S2 > 0 && KK2 > 0, "rdo_tconv_expand: bad argument");
    const long total = (long)Cout * S2 * KK2 * Cin;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(tconv_expand_kernel, dim3(grid_for(total)), dim3(256), 0, s, w, map, total, S2, KK2, taps, Cin, wp);
            return rdo::check_launch("tconv_expand");
        },
        stream, "tconv_expand", 0.0, 8.0 * total);
}

int rdo_tconv_fold(const float* slabs_phase, const int32_t* inv, int32_t nsplit, int32_t Cout, int32_t taps, int32_t Cin, int32_t S2,
                   int32_t KK2, float* slabs, void* stream) {
    RDO_REQUIRE(slabs_phase && inv && slabs && nsplit > 0 && Cout > 0 && taps > 0 && Cin > 0 && S2 > 0 && KK2 > 0, "rdo_tconv_fold: bad argument");
    const long per_slab = (long)Cout * taps * Cin, total = per_slab * nsplit;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(tconv_fold_kernel, dim3(grid_for(total)), dim3(256), 0, s, slabs_phase, inv, total, per_slab, S2, KK2, taps, Cin,
                               slabs);
            return rdo::check_launch("tconv_fold");
        },
        stream, "tconv_fold", 0.0, 8.0 * total);
}

int rdo_layer_norm(const float* x, const float* weight, const float* bias, int64_t rows, int32_t C, float eps, float* out,
                   void* stream) {
    RDO_REQUIRE(x && out && rows > 0 && C > 0, "rdo_layer_norm: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)rdo::ceil_div(rows, 4)), dim3(256), 0, s, x, weight, bias, (long)rows,
                               C, eps, out);
            return rdo::check_launch("layer_norm");
        },
        stream);
}

int rdo_iter_advance(int32_t* iter_ptr, void* stream) {
    RDO_REQUIRE(iter_ptr != nullptr, "rdo_iter_advance: null pointer");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(iter_advance_kernel, dim3(1), dim3(1), 0, s, iter_ptr);
            return rdo::check_launch("iter_advance");
        },
        stream);
}

int rdo_actquant_perchannel(const float* x, int64_t npix, int32_t C, int32_t n_bits, float* out, float* ws_minmax, void* stream) {
    RDO_REQUIRE(x && out && ws_minmax && npix > 0 && C > 0, "rdo_actquant_perchannel: bad argument");
    RDO_REQUIRE(n_bits >= 2 && n_bits <= 16, "rdo_actquant_perchannel: n_bits %d outside [2, 16]", n_bits);
    const float bit_range = (float)((1 << n_bits) - 1);
    float* ws = ws_minmax;
    float* part = ws_minmax + 2 * (long)C;
    const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    return rdo::dispatch(
        [=](hipStream_t s) {
            const int W = vec ? 4 : 1;
            const int QN = C / W, qpb = QN < 256 ? QN : 256, PL = 256 / qpb;
            // enough workgroups to keep HBM busy, few enough that each has >= 8 pixels per lane to amortise its fold
            long g = rdo::ceil_div(npix, (long)PL * 8);
            const int nblk = (int)(g < 1 ? 1 : (g > kAqBlocks ? kAqBlocks : g));
            if (vec) hipLaunchKernelGGL(aq_partial_kernel<4>, dim3(nblk), dim3(256), 0, s, x, (long)npix, C, part);
            else hipLaunchKernelGGL(aq_partial_kernel<1>, dim3(nblk), dim3(256), 0, s, x, (long)npix, C, part);
            hipLaunchKernelGGL(aq_fold_kernel, dim3((unsigned)rdo::ceil_div(2 * C, 16)), dim3(256), 0, s, part, nblk, C, ws);
            const long nvec = (long)npix * C / W;
            if (vec) hipLaunchKernelGGL(aq_apply_kernel<4>, dim3(grid_for(nvec)), dim3(256), 0, s, x, nvec, C, ws, bit_range, out);
            else hipLaunchKernelGGL(aq_apply_kernel<1>, dim3(grid_for(nvec)), dim3(256), 0, s, x, nvec, C, ws, bit_range, out);
            return rdo::check_launch("actquant_perchannel");
        },
        stream);
}

int64_t rdo_actquant_workspace(int32_t C) { return C > 0 ? 2 * (int64_t)C * (kAqBlocks + 1) : 0; }

}  // extern "C"
