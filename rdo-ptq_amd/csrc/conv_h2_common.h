// Shared by the plane-input ("H2") forward kernels (conv_fwd_h2.hip, conv_fwd_h2k.hip): launch arguments, the per-element epilogue
// helpers and the XCD-aware tile numbering.  Everything here has internal linkage (anonymous namespace): each translation unit gets
// its own copy.
#pragma once
#include <cmath>
#include <utility>

#include "rdo_common.h"

namespace rdo {
struct H2Args {
    const unsigned short* xp;        // activation planes (H2 layout of [B*H*W][Cin])
    long xplane;          // elements per activation plane
    const unsigned short* wp;        // weight planes, fragment order [2][Cin/16][KH][KW][Cout][16]
    long wplane;
    const float* bias;
    const float* aux;
    const unsigned short* auxp;      // alternative to aux for the sign-only epilogues (LRELU_BWD / RELU_BWD): H2 planes of the aux tensor (plane 0 is read)
    const float* residual;
    float* out;           // nullable
    float* pre;           // nullable
    unsigned short* outp;            // nullable: result planes (H2 layout of [M][Cout])
    long oplane;
    float acc_scale;      // 1 / (s_x * s_w): the accumulators hold s_x s_w times the convolution sums
    float out_scale;      // s of the tensor written to outp
    int* ovf;             // sticky overflow flag (rdo_h2_overflow)
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M;
    int epilogue, add_residual;
    int xcd_mode;
    int stagger;          // halo kernel: waves 4-7 run two thirds of a stage behind waves 0-3 (tuning key "h2_stagger")
    float* partial;       // split-K: raw accumulators [ksplit][M][Cout]
    int ksplit;
    // unit tail folded into the halo kernel's epilogue (rdo_conv2d_fwd_h2_tail): out = act(conv + bias) + residual, loss against the
    // cached target rows, dL/dpre written as the result planes -- rdo_loss_act_bwd without the pre-activation round trip
    const float* tail_tgt;    // nullptr: no tail
    const int32_t* tail_idx;
    const int32_t* tail_iter;
    int32_t* tail_pub;        // rdo_iter_bind_publish: the launch's first thread leaves *tail_iter there (nullable)
    const unsigned short* tail_resp;     // residual as H2 planes of [M][Cout] (nullable)
    float tail_res_inv;       // 1 / s of that tensor
    float* tail_loss;
    long tail_per_image;
    int tail_B, tail_act;
    float tail_coef, tail_inv_npix;
    // diagnostic bit mask: 1 no A DMA, 2 no B DMA, 4 no MFMA, 8 no fragment reads -- results are WRONG when non-zero.  The host sets it
    // only in a `make DIAG=1` build (tuning key "x6p_ablate"); in the shipped library it is always 0.  It stays a RUN-TIME value on
    // purpose: the basic-block boundaries its tests create pin the order of fragment reads, MFMA groups and DMA issue in the K loop;
    // with the tests folded away at compile time the scheduler's own order was 13-20 % slower on the 4 x 128^2 conv (137-145 us
    // against 121 us, same box).
    int ablate;
};
}  // namespace rdo

namespace {

using rdo::H2Args;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rdo::h2_hi;
using rdo::h2_lo;
constexpr int NP = 2;                                          // planes of an H2 tensor



// two-way split of 8 consecutive channels (times s) -> one 16-byte run per plane
__device__ __forceinline__ void split2_x8(const float* v, float s, u32x4& p0, u32x4& p1, int& bad) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned hi, lo;
        rdo::h2_split_pk(v[2 * k], v[2 * k + 1], s, hi, lo, bad);
        p0[k] = hi;
        p1[k] = lo;
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

__device__ __forceinline__ float activate(const H2Args& a, float v, float aux) {
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

// four aux values of channels [n, n+4) of pixel m: the fp32 tensor, or plane 0 of its H2 form (same sign as the value: enough for the
// activation-backward masks)
__device__ __forceinline__ f32x4 aux_quad(const H2Args& a, int m, int n, long o) {
    if (a.aux) return *reinterpret_cast<const f32x4*>(a.aux + o);
    if (a.auxp) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 b = *reinterpret_cast<const u32x2*>(a.auxp + ((long)(n >> 4) * a.M + m) * 16 + (n & 15));
        return f32x4{h2_lo(b[0]), h2_hi(b[0]), h2_lo(b[1]), h2_hi(b[1])};
    }
    return f32x4{0.f, 0.f, 0.f, 0.f};
}

// the two H2 records of (slice n / 16, pixel m) from 16 finished channel values: 32 bytes per plane
__device__ __forceinline__ void store_slice(const H2Args& a, int m, int n, const float (&v)[16], int& bad) {
    u16* dst = a.outp + ((long)(n >> 4) * a.M + m) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 p0, p1;
        split2_x8(v + 8 * h, a.out_scale, p0, p1, bad);
        *reinterpret_cast<u32x4*>(dst + 8 * h) = p0;
        *reinterpret_cast<u32x4*>(dst + a.oplane + 8 * h) = p1;
    }
}

// finish the 16 channels [n, n+16) (one slice, n % 16 == 0) of output pixel m from their raw sums `v` (bias not yet added)
__device__ __forceinline__ void finish16(const H2Args& a, int m, int n, float (&v)[16], int& bad) {
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
    if (a.outp) store_slice(a, m, n, v, bad);
}

// Unit tail on the 16 finished channels [n, n+16) of pixel m (v = conv + bias, the pre-activation): v becomes dL/dpre, the return value
// is sum d^2.  The arithmetic of loss_act_quad (fused_tail.hip), operation for operation.
__device__ __forceinline__ float tail16(const H2Args& a, int m, int n, float (&v)[16]) {
    const int b = (int)(((long)m * a.Cout) / a.tail_per_image);
    const int it = *a.tail_iter;
    const float* y = a.tail_tgt + (long)a.tail_idx[(long)it * a.tail_B + b] * a.tail_per_image + ((long)m * a.Cout - (long)b * a.tail_per_image) + n;
    const float slope = a.tail_act == 1 ? 0.01f : 0.f;
    const float gs = a.tail_coef * 2.f * a.tail_inv_npix;
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = 0.f;
    if (a.tail_resp) {                                       // the residual from its planes: (h1 + h2) / s, the sum is exact in fp32
        const u16* rp = a.tail_resp + ((long)(n >> 4) * a.M + m) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32x4 p0 = *reinterpret_cast<const u32x4*>(rp + 8 * h), p1 = *reinterpret_cast<const u32x4*>(rp + a.oplane + 8 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[8 * h + 2 * k] = (h2_lo(p0[k]) + h2_lo(p1[k])) * a.tail_res_inv;
                r[8 * h + 2 * k + 1] = (h2_hi(p0[k]) + h2_hi(p1[k])) * a.tail_res_inv;
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


}  // namespace

namespace rdo {
// conv_fwd_h2k.hip: the halo kernel with 32-channel K stages on v_mfma_f32_16x16x32_f16.  shape 1: 256 x 192 tile, 2: 256 x 64, 3: 256 x 48.
bool h2k_supported(const H2Args& a, int shape);
int h2k_launch(const H2Args& a, int shape, hipStream_t s);
}  // namespace rdo
