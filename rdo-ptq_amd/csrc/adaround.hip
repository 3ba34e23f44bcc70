// K4 / K5 / K8(round) / K10 -- AdaRound soft quantiser forward + backward, rounding regulariser and Adam, fused into one
// pass over the weight that also reduces the split-K wgrad slabs and emits next iteration's kernel-layout weights.
//
// Reference arithmetic being restated (task-oriented-PTQ/quantization):
//   quantizer.py:437-449  w~ = (clamp(floor(w/d) + h(alpha) + zp, 0, L-1) - zp) * d        (soft)   | + (alpha>=0)  (hard)
//   quantizer.py:451-452  h(alpha) = clamp(sigmoid(alpha) * 1.2 - 0.1, 0, 1)
//   quantizer.py:454-462  alpha0 = -log(1.2 / (frac(w/d) + 0.1) - 1)
//   layer_opt.py:159-165  round_loss = weight * sum(1 - (2|h - .5|)^b)
//   layer_opt.py:254,307  torch.optim.Adam(lr=1e-3, betas=(.9,.999), eps=1e-8), one step per iteration
//   quant_layer.py:142-146 (GDN gamma): gamma' = max(w~, bound)^2 - pedestal, LowerBound gradient rule (CompressAI)
// This file is built with -ffp-contract=off so products and sums round separately, as the reference's op chain does.
#include "rdo_common.h"
#include "gather_body.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kGamma = -0.1f, kZeta = 1.1f;
constexpr float kBeta1 = 0.9f, kBeta2 = 0.999f, kAdamEps = 1e-8f;

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + __expf(-a)); }

// u^p for u in [0, 1], p in [1, 19] (the rounding regulariser's (2|h - 1/2|)^(b-1)) on the transcendental unit: exp2(p * log2 u).
// HIP's __powf is the correctly rounded library pow -- ~190 instructions, more than the rest of an element's step together and
// ~150 us of every calibration step of the Cheng2020 schedule (tools/ada_step_ablate.py, profiles/r06a_ada_step_ablate.md).  v_log_f32 /
// v_exp_f32 are good to 1 ulp of their results: the power's relative error is <= 4e-8 * |p log2 u|, i.e. <= 3 ulp wherever the
// result is above 1e-3 and an absolute error below 1e-9 everywhere else (the term is a regulariser's value and slope next to a data
// gradient that carries the fp32 summation-order noise of 65 536-pixel reductions).
__device__ __forceinline__ float pow_u(float u, float p) {
    return (u > 0.f) ? __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u) * p) : 0.f;
}

struct AdaArgs {
    rdo_ada_desc d;
    const float* w;
    const float* delta;
    const float* zp;
    const float* slabs;     // [nsplit][numel] data gradient w.r.t. the kernel-layout weight (w~ or gamma')
    int nsplit;
    const float* dalpha_in; // apply mode: data gradient w.r.t. alpha (already chained)
    float* dalpha_out;      // grad mode
    float grad_scale, round_weight;
    const rdo_sched_row* sched;
    const int32_t* iter_ptr;
    float* alpha;
    float* m;
    float* v;
    float* wq;
    float* wd;
    float* round_loss_out;
    int mode;               // 0 fused step, 1 grad only, 2 apply
    unsigned short* wq_planes;   // optional: planes of the new wq in fragment order -- exact bf16 three-way split [3][numel] (conv_fwd_x6.hip
                                 // operand) when wq_pscale == 0, else the fp16 two-way split of wq * wq_pscale [2][numel] (conv_fwd_h2.hip)
    unsigned short* wd_planes;   // optional: the same for the dgrad layout wd
    float wq_pscale, wd_pscale;
    int* ovf;                    // sticky fp16 overflow flag (rdo_h2_overflow)
    unsigned short* lin_fwd;     // optional: planes of wq * lin_pscale in rdo_linear_h2's fragment order ([plane][inner/32][rows/16][lane][8])
    unsigned short* lin_bwd;     // optional: the same of the transpose (rows and inner swapped)
    float lin_pscale;
};
#ifdef RDO_DIAG
// ablation mask of a diagnostic build (tools/ada_step_ablate.py; set by rdo_diag_ada_ablate, read by the kernels at run time so that
// captured graphs follow it): 1 slab reads off, 2 plane writes off, 4 rounding term off, 8 Adam state off, 16 dgrad-layout work off,
// 32 fp32 wq write off
__device__ int g_ada_abl = 0;
#define ADA_ABL(bit) ((g_ada_abl & (bit)) != 0)
#else
#define ADA_ABL(bit) false
#endif

// element (n, k) of W [rows][inner] in the fragment order of rdo_linear_h2 (linear_h2.hip, rdo_split_h2_linear): half index inside a plane
__device__ __forceinline__ long lin_frag_index(int n, int k, int nblk) {
    return ((((long)(k >> 5) * nblk + (n >> 4)) * 64 + (n & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7);
}
__device__ __forceinline__ void lin_split(float v, float scale, unsigned short& hi, unsigned short& lo, int& bad) {
    const float x = v * scale;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
    if (!(fabsf(x) <= 65504.f)) bad = 1;
}

// exact three-way bf16 split (hardware RNE conversions), as rdo_split_bf16x3
__device__ __forceinline__ void split3_store(float v, unsigned short* planes, long n, long i) {
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    planes[i] = __builtin_bit_cast(unsigned short, h);
    planes[n + i] = __builtin_bit_cast(unsigned short, m);
    planes[2 * n + i] = __builtin_bit_cast(unsigned short, l);
}

__device__ __forceinline__ long wd_index(const rdo_ada_desc& d, long e) {
    // e = ((co*KH + kh)*KW + kw)*Cin + ci  ->  ((ci*KH + KH-1-kh)*KW + KW-1-kw)*Cout + co
    const int Cin = d.Cin, KW = d.KW, KH = d.KH;
    long t = e / Cin;
    const int ci = (int)(e - t * Cin);
    const int kw = (int)(t % KW); t /= KW;
    const int kh = (int)(t % KH);
    const long co = t / KH;
    return (((long)ci * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * d.rows + co;
}

// emit kernel-layout weight(s) from the quantised value
__device__ __forceinline__ void emit(const rdo_ada_desc& d, long e, float q, float* wq, float* wd) {
    float o = q;
    if (d.reparam) {
        const float lb = fmaxf(q, d.reparam_bound);
        o = lb * lb - d.reparam_pedestal;
    }
    wq[e] = o;
    if (wd && d.Cin > 0) wd[wd_index(d, e)] = o;
}

// One thread per 4 consecutive weight elements (16-byte accesses on every stream: w, alpha, m, v, nsplit slabs, wq).
// The dgrad layout wd is produced afterwards by wd_transpose_kernel (LDS-tiled, coalesced on both sides).
template <int W>   // W = 4: float4 streams (numel % 4 == 0); W = 1: scalar fallback for odd-sized tensors
__device__ __forceinline__ void ada_step_body(const AdaArgs& a, const long bid, const long nblk) {
    typedef float vec_t __attribute__((ext_vector_type(W)));
    const rdo_ada_desc d = a.d;
    const long inner = d.numel / d.rows;
    const float Lm1 = (float)(d.n_levels - 1);
    float b = 0.f, round_on = 0.f, step_size = 0.f, bc2 = 1.f;
    int it = 0;
    if (a.mode != 1) {
        it = *a.iter_ptr;
        const rdo_sched_row s = a.sched[it];
        b = s.b; round_on = s.round_on; step_size = s.step_size; bc2 = s.bc2_sqrt;
    }
    float rl_local = 0.f;
    const long nq = d.numel / W;
    for (long qi = bid * blockDim.x + threadIdx.x; qi < nq; qi += nblk * blockDim.x) {
        const long e0 = qi * W;
        const vec_t wv4 = *reinterpret_cast<const vec_t*>(a.w + e0);
        vec_t al4 = *reinterpret_cast<const vec_t*>(a.alpha + e0);
        vec_t g4 = wv4 * 0.f;
        if (a.mode == 2) {
            g4 = *reinterpret_cast<const vec_t*>(a.dalpha_in + e0);
        } else if (!ADA_ABL(1)) {
            // eight slab loads in flight per thread: with ~5 waves per CU at these sizes a serial chain of nsplit dependent
            // loads (one L2 / fabric round trip each) was the whole kernel time
            const float* sp = a.slabs + e0;
            int s = 0;
            for (; s + 16 <= a.nsplit; s += 16) {            // long chains (the 28-slab 3x3 convs, the 256-slab 1x1): sixteen in flight
                vec_t t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                      (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
            }
            for (; s + 8 <= a.nsplit; s += 8) {
                vec_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            if (s + 4 <= a.nsplit) {
                vec_t t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += (t[0] + t[1]) + (t[2] + t[3]);
                s += 4;
            }
            for (; s < a.nsplit; ++s) g4 += *reinterpret_cast<const vec_t*>(sp + (long)s * d.numel);
        }
        vec_t m4 = wv4 * 0.f, v4 = m4, o4 = m4;
        if (a.mode != 1 && !ADA_ABL(8)) {
            m4 = *reinterpret_cast<const vec_t*>(a.m + e0);
            v4 = *reinterpret_cast<const vec_t*>(a.v + e0);
        }
        // row (output channel) of the W elements: ONE division per thread -- the 64-bit division per ELEMENT was ~50 of the ~170 vector
        // instructions an element costs, and the step of the 1.3 M-element tensors is bound by exactly those (19 -> 25 us with the
        // rounding term's powf on top: 5.3 M elements x 170 instructions / 64 lanes against 614 G wave-instructions/s)
        int rows[W];
        if (inner >= W && d.numel < (1L << 31)) {            // (wave-uniform)
            const unsigned row0 = (unsigned)e0 / (unsigned)inner;
            const unsigned rem0 = (unsigned)e0 - row0 * (unsigned)inner;
#pragma unroll
            for (int k = 0; k < W; ++k) rows[k] = (int)(row0 + (rem0 + k >= (unsigned)inner ? 1u : 0u));
        } else {
#pragma unroll
            for (int k = 0; k < W; ++k) rows[k] = (int)((e0 + k) / inner);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int row = rows[k];
            const float dl = a.delta[row], z = a.zp[row], wv = wv4[k];
            float al = al4[k];
            const float xf = floorf(wv / dl);
            const float sg = sigmoidf_(al);
            const float hraw = sg * (kZeta - kGamma) + kGamma;
            const float h = fminf(fmaxf(hraw, 0.f), 1.f);
            const float xint = xf + h + z;       // (x_floor + h) + zp
            const float pass_h = (hraw >= 0.f && hraw <= 1.f) ? 1.f : 0.f;
            const float dh_da = pass_h * ((kZeta - kGamma) * (sg * (1.f - sg)));
            float g_alpha = g4[k];  // mode 2: already the data gradient w.r.t. alpha
            if (a.mode != 2) {
                float g = g4[k];
                if (d.reparam) {
                    const float q = (fminf(fmaxf(xint, 0.f), Lm1) - z) * dl;
                    const float lb = fmaxf(q, d.reparam_bound);
                    const float go = g * (2.f * lb);                      // d(lb^2)/dlb
                    g = (q >= d.reparam_bound || go < 0.f) ? go : 0.f;    // LowerBound backward
                }
                const float pass_q = (xint >= 0.f && xint <= Lm1) ? 1.f : 0.f;
                g_alpha = (g * dl) * pass_q * dh_da;
            }
            if (a.mode == 1) {
                o4[k] = g_alpha;
                continue;
            }
            // rounding regulariser (value of the current alpha, gradient through h)
            float g_total = g_alpha * a.grad_scale;
            if (round_on != 0.f && !ADA_ABL(4)) {
                const float u = fabsf(h - 0.5f) * 2.f;
                const float ub1 = pow_u(u, b - 1.f);                       // u^(b-1); u^b = u * u^(b-1)
                rl_local += a.round_weight * (1.f - u * ub1);
                const float sgn = (h > 0.5f) ? 1.f : ((h < 0.5f) ? -1.f : 0.f);
                g_total += (-a.round_weight * (b * ub1) * 2.f * sgn) * dh_da;
            }
            // Adam (torch.optim.Adam defaults; alpha has no weight decay)
            float mm = m4[k], vv = v4[k];
            mm = mm + (g_total - mm) * (1.f - kBeta1);
            vv = vv * kBeta2 + (1.f - kBeta2) * g_total * g_total;
            const float denom = sqrtf(vv) / bc2 + kAdamEps;
            al = al - step_size * (mm / denom);
            m4[k] = mm; v4[k] = vv; al4[k] = al;
            // next iteration's soft weight
            const float sg2 = sigmoidf_(al);
            const float h2 = fminf(fmaxf(sg2 * (kZeta - kGamma) + kGamma, 0.f), 1.f);
            float o = (fminf(fmaxf(xf + h2 + z, 0.f), Lm1) - z) * dl;
            if (d.reparam) {
                const float lb = fmaxf(o, d.reparam_bound);
                o = lb * lb - d.reparam_pedestal;
            }
            o4[k] = o;
        }
        if (a.mode == 1) {
            *reinterpret_cast<vec_t*>(a.dalpha_out + e0) = o4;
        } else {
            if (!ADA_ABL(8)) {
                *reinterpret_cast<vec_t*>(a.m + e0) = m4;
                *reinterpret_cast<vec_t*>(a.v + e0) = v4;
            }
            *reinterpret_cast<vec_t*>(a.alpha + e0) = al4;
            if (!ADA_ABL(32)) *reinterpret_cast<vec_t*>(a.wq + e0) = o4;
            if (ADA_ABL(2)) continue;
            if (a.lin_fwd || a.lin_bwd) {                      // planes for rdo_linear_h2 (a Linear / GDN gamma on large token matrices)
                const int in_ = (int)inner, nr = d.rows;
                const long plane = (long)nr * in_;
                int bad = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int n = rows[k], kk = (int)(e0 + k - (long)n * in_);
                    unsigned short hi, lo;
                    lin_split(o4[k], a.lin_pscale, hi, lo, bad);
                    if (a.lin_fwd) {
                        const long f = lin_frag_index(n, kk, nr >> 4);
                        a.lin_fwd[f] = hi; a.lin_fwd[plane + f] = lo;
                    }
                    if (a.lin_bwd) {
                        const long f = lin_frag_index(kk, n, in_ >> 4);
                        a.lin_bwd[f] = hi; a.lin_bwd[plane + f] = lo;
                    }
                }
                rdo::h2_report(bad, a.ovf);
            }
            if (a.wq_planes && a.wq_pscale > 0.f) {            // fp16 two-way split of wq * scale, fragment order
                int bad = 0;
                if constexpr (W == 4) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 ph, pl;
                    unsigned h, l;
                    rdo::h2_split_pk(o4[0], o4[1], a.wq_pscale, h, l, bad); ph[0] = h; pl[0] = l;
                    rdo::h2_split_pk(o4[2], o4[3], a.wq_pscale, h, l, bad); ph[1] = h; pl[1] = l;
                    const long f0 = d.Cin > 0 ? rdo::frag_index(e0, d.rows, d.KH, d.KW, d.Cin) : e0;
                    *reinterpret_cast<u32x2*>(a.wq_planes + f0) = ph;
                    *reinterpret_cast<u32x2*>(a.wq_planes + d.numel + f0) = pl;
                } else {
#pragma unroll
                    for (int k = 0; k < W; ++k)
                        rdo::h2_split_store(o4[k], a.wq_pscale, a.wq_planes, d.numel,
                                            d.Cin > 0 ? rdo::frag_index(e0 + k, d.rows, d.KH, d.KW, d.Cin) : e0 + k, bad);
                }
                rdo::h2_report(bad, a.ovf);
            } else if (a.wq_planes) {
                if constexpr (W == 4) {                        // one 8-byte store per plane
                    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                    u16x4 ph, pm, pl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const __bf16 h = (__bf16)o4[k];
                        const float r1 = o4[k] - (float)h;
                        const __bf16 m = (__bf16)r1;
                        const __bf16 l = (__bf16)(r1 - (float)m);
                        ph[k] = __builtin_bit_cast(unsigned short, h);
                        pm[k] = __builtin_bit_cast(unsigned short, m);
                        pl[k] = __builtin_bit_cast(unsigned short, l);
                    }
                    // fragment order (rdo::frag_index): four consecutive channels stay consecutive (Cin % 16 == 0, e0 % 4 == 0)
                    const long f0 = d.Cin > 0 ? rdo::frag_index(e0, d.rows, d.KH, d.KW, d.Cin) : e0;
                    *reinterpret_cast<u16x4*>(a.wq_planes + f0) = ph;
                    *reinterpret_cast<u16x4*>(a.wq_planes + d.numel + f0) = pm;
                    *reinterpret_cast<u16x4*>(a.wq_planes + 2 * d.numel + f0) = pl;
                } else {
#pragma unroll
                    for (int k = 0; k < W; ++k)
                        split3_store(o4[k], a.wq_planes, d.numel, d.Cin > 0 ? rdo::frag_index(e0 + k, d.rows, d.KH, d.KW, d.Cin) : e0 + k);
                }
            }
        }
    }
    if (a.mode != 1 && a.round_loss_out && round_on != 0.f) {
        __shared__ float red[4];
        float vsum = rl_local;
        for (int o = 32; o > 0; o >>= 1) vsum += __shfl_down(vsum, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = red[0] + red[1] + red[2] + red[3];
            if (t != 0.f) atomicAdd(a.round_loss_out + (long)it * RDO_LOG_SLOTS + (bid & (RDO_LOG_SLOTS - 1)), t);
        }
    }
}


// ---- tile form of the fused step (modes 0 and 2) -------------------------------------------------------------------------------------
// One workgroup owns 32 rows (output channels) x 32 input channels of ONE tap of a conv weight [rows][KH][KW][Cin] (a GDN gamma or a
// Linear is the one-tap case): thread (r = tid / 8, q = tid % 8) steps the four channels 4q .. 4q + 3 of row r -- every stream is read and
// written in 128-byte row segments (eight lanes per segment), a thread's quantisation row is fixed (no index division per element) --
// and the new soft weights go through a 32 x 32 LDS tile so that the SAME launch writes the dgrad layout wd[ci][KH-1-kh][KW-1-kw][co],
// its planes and the transposed rdo_linear_h2 planes with 16-byte / 8-byte stores along co.  Until round 6 that was a second launch
// (wd_transpose_batch_kernel) that re-read wq: 13 launches and ~0.1-0.2 ms of every calibration step of the Cheng2020 schedule
// (profiles/r06a_ada_step_ablate.md).  Needs Cin % 4 == 0 and rows % 4 == 0; everything else stays on the flat form above.
__device__ __forceinline__ void ada_step_tile_body(const AdaArgs& a, int bid) {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const rdo_ada_desc d = a.d;
    const int Cin = d.Cin, taps = d.KH * d.KW, nrows = d.rows;
    const int ctiles = (Cin + 31) >> 5;
    const int bid0 = bid;
    const int ct = bid % ctiles; bid /= ctiles;
    const int tap = bid % taps;
    const int rt = bid / taps;
    const int q = threadIdx.x & 7, r = threadIdx.x >> 3;
    const int row = rt * 32 + r, ci = ct * 32 + 4 * q;
    const bool live = row < nrows && ci < Cin;
    const float Lm1 = (float)(d.n_levels - 1);
    const int it = *a.iter_ptr;
    const rdo_sched_row sc = a.sched[it];
    const float b = sc.b, round_on = sc.round_on, step_size = sc.step_size, bc2 = sc.bc2_sqrt;
    float rl_local = 0.f;
    vec_t o4 = {0.f, 0.f, 0.f, 0.f};
    const long e0 = ((long)row * taps + tap) * Cin + ci;
    if (live) {
        const vec_t wv4 = *reinterpret_cast<const vec_t*>(a.w + e0);
        vec_t al4 = *reinterpret_cast<const vec_t*>(a.alpha + e0);
        vec_t g4 = {0.f, 0.f, 0.f, 0.f};
        if (a.mode == 2) {
            g4 = *reinterpret_cast<const vec_t*>(a.dalpha_in + e0);
        } else if (!ADA_ABL(1)) {
            const float* sp = a.slabs + e0;                     // same summation order as the flat form
            int s = 0;
            for (; s + 16 <= a.nsplit; s += 16) {
                vec_t t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                      (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
            }
            for (; s + 8 <= a.nsplit; s += 8) {
                vec_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            if (s + 4 <= a.nsplit) {
                vec_t t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const vec_t*>(sp + (long)(s + u) * d.numel);
                g4 += (t[0] + t[1]) + (t[2] + t[3]);
                s += 4;
            }
            for (; s < a.nsplit; ++s) g4 += *reinterpret_cast<const vec_t*>(sp + (long)s * d.numel);
        }
        vec_t m4 = {0.f, 0.f, 0.f, 0.f}, v4 = m4;
        if (!ADA_ABL(8)) {
            m4 = *reinterpret_cast<const vec_t*>(a.m + e0);
            v4 = *reinterpret_cast<const vec_t*>(a.v + e0);
        }
        const float dl = a.delta[row], z = a.zp[row];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float al = al4[k];
            const float xf = floorf(wv4[k] / dl);
            const float sg = sigmoidf_(al);
            const float hraw = sg * (kZeta - kGamma) + kGamma;
            const float h = fminf(fmaxf(hraw, 0.f), 1.f);
            const float xint = xf + h + z;
            const float pass_h = (hraw >= 0.f && hraw <= 1.f) ? 1.f : 0.f;
            const float dh_da = pass_h * ((kZeta - kGamma) * (sg * (1.f - sg)));
            float g_alpha = g4[k];
            if (a.mode != 2) {
                float g = g4[k];
                if (d.reparam) {
                    const float qv = (fminf(fmaxf(xint, 0.f), Lm1) - z) * dl;
                    const float lb = fmaxf(qv, d.reparam_bound);
                    const float go = g * (2.f * lb);
                    g = (qv >= d.reparam_bound || go < 0.f) ? go : 0.f;
                }
                const float pass_q = (xint >= 0.f && xint <= Lm1) ? 1.f : 0.f;
                g_alpha = (g * dl) * pass_q * dh_da;
            }
            float g_total = g_alpha * a.grad_scale;
            if (round_on != 0.f && !ADA_ABL(4)) {
                const float u = fabsf(h - 0.5f) * 2.f;
                const float ub1 = pow_u(u, b - 1.f);
                rl_local += a.round_weight * (1.f - u * ub1);
                const float sgn = (h > 0.5f) ? 1.f : ((h < 0.5f) ? -1.f : 0.f);
                g_total += (-a.round_weight * (b * ub1) * 2.f * sgn) * dh_da;
            }
            float mm = m4[k], vv = v4[k];
            mm = mm + (g_total - mm) * (1.f - kBeta1);
            vv = vv * kBeta2 + (1.f - kBeta2) * g_total * g_total;
            const float denom = sqrtf(vv) / bc2 + kAdamEps;
            al = al - step_size * (mm / denom);
            m4[k] = mm; v4[k] = vv; al4[k] = al;
            const float sg2 = sigmoidf_(al);
            const float h2 = fminf(fmaxf(sg2 * (kZeta - kGamma) + kGamma, 0.f), 1.f);
            float o = (fminf(fmaxf(xf + h2 + z, 0.f), Lm1) - z) * dl;
            if (d.reparam) {
                const float lb = fmaxf(o, d.reparam_bound);
                o = lb * lb - d.reparam_pedestal;
            }
            o4[k] = o;
        }
        if (!ADA_ABL(8)) {
            *reinterpret_cast<vec_t*>(a.m + e0) = m4;
            *reinterpret_cast<vec_t*>(a.v + e0) = v4;
        }
        *reinterpret_cast<vec_t*>(a.alpha + e0) = al4;
        if (!ADA_ABL(32)) *reinterpret_cast<vec_t*>(a.wq + e0) = o4;
        if (!ADA_ABL(2)) {
            int bad = 0;
            if (a.lin_fwd) {                                    // rdo_linear_h2 planes of W [rows][Cin]: four consecutive k = 8 bytes
                unsigned hh, ll;
                u32x2 ph, pl;
                rdo::h2_split_pk(o4[0], o4[1], a.lin_pscale, hh, ll, bad); ph[0] = hh; pl[0] = ll;
                rdo::h2_split_pk(o4[2], o4[3], a.lin_pscale, hh, ll, bad); ph[1] = hh; pl[1] = ll;
                const long f = lin_frag_index(row, ci, nrows >> 4);
                *reinterpret_cast<u32x2*>(a.lin_fwd + f) = ph;
                *reinterpret_cast<u32x2*>(a.lin_fwd + (long)nrows * Cin + f) = pl;
            }
            if (a.wq_planes) {
                const long f0 = (Cin & 15) ? e0 : (((((long)(ci >> 4) * taps + tap) * nrows + row) << 4) + (ci & 15));   // rdo::frag_index
                if (a.wq_pscale > 0.f) {
                    unsigned hh, ll;
                    u32x2 ph, pl;
                    rdo::h2_split_pk(o4[0], o4[1], a.wq_pscale, hh, ll, bad); ph[0] = hh; pl[0] = ll;
                    rdo::h2_split_pk(o4[2], o4[3], a.wq_pscale, hh, ll, bad); ph[1] = hh; pl[1] = ll;
                    *reinterpret_cast<u32x2*>(a.wq_planes + f0) = ph;
                    *reinterpret_cast<u32x2*>(a.wq_planes + d.numel + f0) = pl;
                } else {
                    u16x4 ph, pm, pl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const __bf16 hb = (__bf16)o4[k];
                        const float r1 = o4[k] - (float)hb;
                        const __bf16 mb = (__bf16)r1;
                        const __bf16 lb = (__bf16)(r1 - (float)mb);
                        ph[k] = __builtin_bit_cast(unsigned short, hb);
                        pm[k] = __builtin_bit_cast(unsigned short, mb);
                        pl[k] = __builtin_bit_cast(unsigned short, lb);
                    }
                    *reinterpret_cast<u16x4*>(a.wq_planes + f0) = ph;
                    *reinterpret_cast<u16x4*>(a.wq_planes + d.numel + f0) = pm;
                    *reinterpret_cast<u16x4*>(a.wq_planes + 2 * d.numel + f0) = pl;
                }
            }
            rdo::h2_report(bad, a.ovf);
        }
    }
    // ---- dgrad layout: transpose the tile's new soft weights through LDS (wave-uniform condition: the barrier is reached by all or none)
    __shared__ float tile[32][33];
    __shared__ float red[4];
    if ((a.wd || a.lin_bwd) && !ADA_ABL(16)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[r][4 * q + k] = o4[k];
        __syncthreads();
        const int ci2 = ct * 32 + r, co2 = rt * 32 + 4 * q;      // this thread now owns four consecutive output channels of one input channel
        if (ci2 < Cin && co2 < nrows) {
            vec_t t4;
#pragma unroll
            for (int k = 0; k < 4; ++k) t4[k] = tile[4 * q + k][r];
            const int kh = tap / d.KW, kw = tap - kh * d.KW;
            const int tapf = (d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw);
            const long o = ((long)ci2 * taps + tapf) * nrows + co2;
            if (a.wd) *reinterpret_cast<vec_t*>(a.wd + o) = t4;
            int bad = 0;
            if (a.lin_bwd && !ADA_ABL(2)) {
                unsigned hh, ll;
                u32x2 ph, pl;
                rdo::h2_split_pk(t4[0], t4[1], a.lin_pscale, hh, ll, bad); ph[0] = hh; pl[0] = ll;
                rdo::h2_split_pk(t4[2], t4[3], a.lin_pscale, hh, ll, bad); ph[1] = hh; pl[1] = ll;
                const long f = lin_frag_index(ci2, co2, Cin >> 4);
                *reinterpret_cast<u32x2*>(a.lin_bwd + f) = ph;
                *reinterpret_cast<u32x2*>(a.lin_bwd + (long)nrows * Cin + f) = pl;
            }
            if (a.wd && a.wd_planes && !ADA_ABL(2)) {
                // wd is the weight [Cin][KH][KW][Cout] of the dgrad conv: its planes go in THAT conv's fragment order (slices of 16 co)
                const long f0 = (nrows & 15) ? o : (((((long)(co2 >> 4) * taps + tapf) * Cin + ci2) << 4) + (co2 & 15));
                if (a.wd_pscale > 0.f) {
                    unsigned hh, ll;
                    u32x2 ph, pl;
                    rdo::h2_split_pk(t4[0], t4[1], a.wd_pscale, hh, ll, bad); ph[0] = hh; pl[0] = ll;
                    rdo::h2_split_pk(t4[2], t4[3], a.wd_pscale, hh, ll, bad); ph[1] = hh; pl[1] = ll;
                    *reinterpret_cast<u32x2*>(a.wd_planes + f0) = ph;
                    *reinterpret_cast<u32x2*>(a.wd_planes + d.numel + f0) = pl;
                } else {
                    u16x4 ph, pm, pl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const __bf16 hb = (__bf16)t4[k];
                        const float r1 = t4[k] - (float)hb;
                        const __bf16 mb = (__bf16)r1;
                        const __bf16 lb = (__bf16)(r1 - (float)mb);
                        ph[k] = __builtin_bit_cast(unsigned short, hb);
                        pm[k] = __builtin_bit_cast(unsigned short, mb);
                        pl[k] = __builtin_bit_cast(unsigned short, lb);
                    }
                    *reinterpret_cast<u16x4*>(a.wd_planes + f0) = ph;
                    *reinterpret_cast<u16x4*>(a.wd_planes + d.numel + f0) = pm;
                    *reinterpret_cast<u16x4*>(a.wd_planes + 2 * d.numel + f0) = pl;
                }
            }
            rdo::h2_report(bad, a.ovf);
        }
    }
    if (a.round_loss_out && round_on != 0.f) {
        float vsum = rl_local;
        for (int o = 32; o > 0; o >>= 1) vsum += __shfl_down(vsum, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = red[0] + red[1] + red[2] + red[3];
            if (t != 0.f) atomicAdd(a.round_loss_out + (long)it * RDO_LOG_SLOTS + (bid0 & (RDO_LOG_SLOTS - 1)), t);
        }
    }
}

// Does a tensor's fused step / apply run in the tile form?  (host side; the kernel gets the answer as `tile` in the batch descriptor)
inline bool tile_ok(const rdo_ada_desc& d, int mode) {
    return mode != 1 && d.Cin > 0 && d.Cin % 4 == 0 && d.rows % 4 == 0 && (long)d.rows * d.KH * d.KW * d.Cin == d.numel && d.numel < (1L << 31);
}
inline long tile_blocks(const rdo_ada_desc& d) { return rdo::ceil_div(d.rows, 32) * d.KH * d.KW * rdo::ceil_div(d.Cin, 32); }

template <int W>
__global__ __launch_bounds__(256) void ada_step_kernel(AdaArgs a) {
    ada_step_body<W>(a, blockIdx.x, gridDim.x);
}

// every weight tensor of a unit in ONE launch (the AdaRound step of a block unit was 3-5 launches of 6-36 us, each a few hundred
// workgroups that left most of the chip idle): block ranges [blk_end[t-1], blk_end[t]) belong to tensor t
constexpr int kMaxBatch = 8;
// (the slab count from which a tensor is walked one element per thread is the tuning key "ada_w1_min", default 128: lower values measured slower)
struct AdaBatch {
    int w1_min;             // tensors with at least this many gradient slabs run one element per thread (tuning key "ada_w1_min")
    int32_t* iter_shadow;   // nullable: block 0 leaves *iter_ptr + 1 here (iteration-counter hand-over, rdo_ptq_hip.h)
    AdaArgs a[kMaxBatch];
    int blk_end[kMaxBatch];
    unsigned char tile[kMaxBatch];   // this tensor runs the tile form (ada_step_tile_body): the dgrad layout is written by the same launch
    int n;
    // Round 6: the launch also assembles the NEXT iteration's mini-batch (rdo_adaround_step_batch_gather): blocks from `gather_beg` on run
    // the gather for iteration *iter_ptr + 1 -- two memory streams that share no data in one launch, one launch less per iteration.
    int gather_beg;                  // first gather block (= the number of step blocks; no gather: never reached)
    rdo::gq::Gather g;
};
__global__ __launch_bounds__(256) void ada_step_batch_kernel(AdaBatch b) {
    if (b.iter_shadow && blockIdx.x == 0 && threadIdx.x == 0) *b.iter_shadow = *b.a[0].iter_ptr + 1;
    if ((int)blockIdx.x >= b.gather_beg) {                   // the next iteration's mini-batch (nothing behind the last iteration)
        const int it1 = *b.a[0].iter_ptr + 1;
        if (it1 < b.g.n_iters) {
            if (b.g.planes.p) rdo::gq::gather_h2_body(b.g, it1, (long)blockIdx.x - b.gather_beg, (long)gridDim.x - b.gather_beg);
            else rdo::gq::gather_fp32_body(b.g, it1, (long)blockIdx.x - b.gather_beg, (long)gridDim.x - b.gather_beg);
        }
        return;
    }
    int t = 0;
    while (t + 1 < b.n && (int)blockIdx.x >= b.blk_end[t]) ++t;
    const int beg = t ? b.blk_end[t - 1] : 0;
    if (b.tile[t]) { ada_step_tile_body(b.a[t], (int)blockIdx.x - beg); return; }
    // many slabs over a small tensor (the 1x1 GDN gamma gradient, 256 slabs of 36 K elements): one element per thread -- four times
    // the threads walking the slab chain; same per-element summation order
    if (b.a[t].nsplit >= b.w1_min && b.a[t].mode != 2) ada_step_body<1>(b.a[t], (long)blockIdx.x - beg, (long)b.blk_end[t] - beg);
    else ada_step_body<4>(b.a[t], (long)blockIdx.x - beg, (long)b.blk_end[t] - beg);
}

// wd[ci][KH-1-kh][KW-1-kw][co] = wq[co][kh][kw][ci]: 32x32 tiles through LDS, 128-byte segments on both sides
__device__ __forceinline__ void wd_transpose_body(const rdo_ada_desc& d, const float* wq, float* wd, unsigned short* wd_planes, float pscale,
                                                  int* ovf, int bid) {
    const int taps = d.KH * d.KW, cdim = d.Cin;
    const int ctiles = (cdim + 31) / 32;
    const int ct = bid % ctiles; bid /= ctiles;
    const int tap = bid % taps;
    const int rt = bid / taps;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    __shared__ float tile[32][33];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = rt * 32 + ty + 8 * k, c = ct * 32 + tx;
        if (row < d.rows && c < cdim) tile[ty + 8 * k][tx] = wq[((long)row * taps + tap) * cdim + c];
    }
    __syncthreads();
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int tapf = (d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = ct * 32 + ty + 8 * k, co = rt * 32 + tx;
        if (ci < cdim && co < d.rows) {
            const long o = ((long)ci * taps + tapf) * d.rows + co;
            wd[o] = tile[tx][ty + 8 * k];
            // wd is the weight [Cin][KH][KW][Cout] of the dgrad conv: its planes go in that conv's fragment order
            if (wd_planes && pscale > 0.f) {
                int bad = 0;
                rdo::h2_split_store(tile[tx][ty + 8 * k], pscale, wd_planes, d.numel, rdo::frag_index(o, d.Cin, d.KH, d.KW, d.rows), bad);
                rdo::h2_report(bad, ovf);
            } else if (wd_planes) {
                split3_store(tile[tx][ty + 8 * k], wd_planes, d.numel, rdo::frag_index(o, d.Cin, d.KH, d.KW, d.rows));
            }
        }
    }
}
__global__ __launch_bounds__(256) void wd_transpose_kernel(rdo_ada_desc d, const float* wq, float* wd, unsigned short* wd_planes, float pscale,
                                                           int* ovf) {
    wd_transpose_body(d, wq, wd, wd_planes, pscale, ovf, blockIdx.x);
}
// batched form; also advances the device iteration counter when asked to (the launch does not read it, and every kernel of the
// iteration that does has completed before this one starts: stream order)
__global__ __launch_bounds__(256) void wd_transpose_batch_kernel(AdaBatch b, int32_t* advance) {
    int t = 0;
    while (t + 1 < b.n && (int)blockIdx.x >= b.blk_end[t]) ++t;
    const int beg = t ? b.blk_end[t - 1] : 0;
    if (!ADA_ABL(16)) wd_transpose_body(b.a[t].d, b.a[t].wq, b.a[t].wd, b.a[t].wd_planes, b.a[t].wd_pscale, b.a[t].ovf, (int)blockIdx.x - beg);
    if (advance && blockIdx.x == 0 && threadIdx.x == 0) *advance += 1;
}

__global__ void iter_advance1_kernel(int32_t* it) { *it += 1; }

__global__ __launch_bounds__(256) void ada_fwd_kernel(rdo_ada_desc d, const float* w, const float* alpha, const float* delta,
                                                      const float* zp, int mode /*0 hard,1 soft,2 nearest*/, float* wq,
                                                      float* wd) {
    const long inner = d.numel / d.rows;
    const float Lm1 = (float)(d.n_levels - 1);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < d.numel; e += (long)gridDim.x * blockDim.x) {
        const int row = (int)(e / inner);
        const float dl = delta[row], z = zp[row], wv = w[e];
        float xint;
        if (mode == 2) {
            xint = rintf(wv / dl) + z;
        } else {
            const float xf = floorf(wv / dl);
            float h;
            if (mode == 1) {
                h = fminf(fmaxf(sigmoidf_(alpha[e]) * (kZeta - kGamma) + kGamma, 0.f), 1.f);
            } else {
                h = alpha[e] >= 0.f ? 1.f : 0.f;
            }
            xint = xf + h + z;
        }
        emit(d, e, (fminf(fmaxf(xint, 0.f), Lm1) - z) * dl, wq, wd);
    }
}

__global__ __launch_bounds__(256) void ada_init_kernel(rdo_ada_desc d, const float* w, const float* delta, float* alpha) {
    const long inner = d.numel / d.rows;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < d.numel; e += (long)gridDim.x * blockDim.x) {
        const float dl = delta[e / inner];
        const float r = w[e] / dl;
        const float rest = r - floorf(r);
        alpha[e] = -logf((kZeta - kGamma) / (rest - kGamma) - 1.f);
    }
}

// one block per row: min / max -> delta, zp  ('max' init, quantizer.py:281-298: min/max clamped through 0, python-double
// arithmetic for the range, fp32 for the zero point)
__global__ __launch_bounds__(256) void uaq_minmax_kernel(const float* w, long inner, int n_levels, float* delta, float* zp) {
    const float* p = w + (long)blockIdx.x * inner;
    float mn = 0.f, mx = 0.f;
    for (long i = threadIdx.x; i < inner; i += blockDim.x) {
        const float v = p[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o, 64));
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        float dl = (float)(((double)mx - (double)mn) / (double)(n_levels - 1));
        dl = fmaxf(dl, 1e-8f);
        delta[blockIdx.x] = dl;
        // the reference writes (-x_min / delta) with a Python float on the left: torch evaluates that as delta.reciprocal() * (-x_min)
        // in fp32 (Tensor.__rdiv__), one rounding more than a division -- it decides the ties at x.5 (symmetric weight ranges)
        zp[blockIdx.x] = rintf(__fmul_rn(__frcp_rn(dl), -mn));
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* slabs, int nsplit, long numel, float* out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < numel; e += (long)gridDim.x * blockDim.x) {
        float g = 0.f;
        for (int s = 0; s < nsplit; ++s) g += slabs[(long)s * numel + e];
        out[e] = g;
    }
}

inline unsigned grid_for(long n) {
    long g = rdo::ceil_div(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int check_desc(const rdo_ada_desc* d, const char* who) {
    RDO_REQUIRE(d != nullptr, "%s: null descriptor", who);
    RDO_REQUIRE(d->numel > 0 && d->rows > 0 && d->numel % d->rows == 0, "%s: numel %ld not divisible by rows %d", who,
                (long)d->numel, d->rows);
    RDO_REQUIRE(d->n_levels >= 4 && d->n_levels <= 65536, "%s: n_levels %d unsupported", who, d->n_levels);
    if (d->Cin > 0)
        RDO_REQUIRE(d->KH > 0 && d->KW > 0 && (long)d->rows * d->KH * d->KW * d->Cin == d->numel,
                    "%s: conv layout [%d][%d][%d][%d] does not match numel %ld", who, d->rows, d->KH, d->KW, d->Cin,
                    (long)d->numel);
    return RDO_OK;
}

int run_step(AdaArgs a, void* stream) {

    return rdo::dispatch(
        [a](hipStream_t s) {
            if (a.d.numel % 4 == 0)
                hipLaunchKernelGGL(ada_step_kernel<4>, dim3(grid_for(a.d.numel / 4)), dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL(ada_step_kernel<1>, dim3(grid_for(a.d.numel)), dim3(256), 0, s, a);
            if (a.mode != 1 && a.wd && a.d.Cin > 0) {
                const long blocks = rdo::ceil_div(a.d.rows, 32) * a.d.KH * a.d.KW * rdo::ceil_div(a.d.Cin, 32);
                hipLaunchKernelGGL(wd_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.d, (const float*)a.wq, a.wd, a.wd_planes, a.wd_pscale,
                                   a.ovf);
            }
            return rdo::check_launch("ada_step");
        },
        stream, "ada_step", 0.0, 4.0 * a.d.numel * ((a.mode == 2 ? 1 : a.nsplit) + 9.0));
}

}  // namespace

extern "C" {

#ifdef RDO_DIAG
int rdo_diag_ada_ablate(int mask) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ada_abl), &mask, sizeof mask) == hipSuccess ? RDO_OK : rdo::set_error(RDO_EHIP, "rdo_diag_ada_ablate");
}
#endif

int rdo_adaround_init_alpha(const rdo_ada_desc* d, const float* w, const float* delta, float* alpha, void* stream) {
    if (int rc = check_desc(d, "rdo_adaround_init_alpha")) return rc;
    RDO_REQUIRE(w && delta && alpha, "rdo_adaround_init_alpha: null pointer");
    const rdo_ada_desc dd = *d;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(ada_init_kernel, dim3(grid_for(dd.numel)), dim3(256), 0, s, dd, w, delta, alpha);
            return rdo::check_launch("ada_init");
        },
        stream);
}

int rdo_adaround_fwd(const rdo_ada_desc* d, const float* w, const float* alpha, const float* delta, const float* zp, int soft,
                     float* wq, float* wd, void* stream) {
    if (int rc = check_desc(d, "rdo_adaround_fwd")) return rc;
    RDO_REQUIRE(w && alpha && delta && zp && wq, "rdo_adaround_fwd: null pointer");
    const rdo_ada_desc dd = *d;
    const int mode = soft ? 1 : 0;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(ada_fwd_kernel, dim3(grid_for(dd.numel)), dim3(256), 0, s, dd, w, alpha, delta, zp, mode, wq, wd);
            return rdo::check_launch("ada_fwd");
        },
        stream);
}

int rdo_uaq_fakequant(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, float* wq, float* wd,
                      void* stream) {
    if (int rc = check_desc(d, "rdo_uaq_fakequant")) return rc;
    RDO_REQUIRE(w && delta && zp && wq, "rdo_uaq_fakequant: null pointer");
    const rdo_ada_desc dd = *d;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(ada_fwd_kernel, dim3(grid_for(dd.numel)), dim3(256), 0, s, dd, w, (const float*)nullptr, delta, zp,
                               2, wq, wd);
            return rdo::check_launch("uaq_fakequant");
        },
        stream);
}

int rdo_uaq_init_minmax(const float* w, int32_t rows, int64_t inner, int32_t n_levels, float* delta, float* zp, void* stream) {
    RDO_REQUIRE(w && delta && zp && rows > 0 && inner > 0 && n_levels >= 4, "rdo_uaq_init_minmax: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(uaq_minmax_kernel, dim3(rows), dim3(256), 0, s, w, (long)inner, n_levels, delta, zp);
            return rdo::check_launch("uaq_minmax");
        },
        stream);
}

int rdo_reduce_slabs(const float* slabs, int nsplit, int64_t numel, float* out, void* stream) {
    RDO_REQUIRE(slabs && out && nsplit >= 1 && numel > 0, "rdo_reduce_slabs: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(numel)), dim3(256), 0, s, slabs, nsplit, (long)numel, out);
            return rdo::check_launch("reduce_slabs");
        },
        stream);
}

int rdo_adaround_step(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, const float* slabs, int nsplit,
                      float grad_scale, float round_weight, const rdo_sched_row* sched, const int32_t* iter_ptr, float* alpha,
                      float* adam_m, float* adam_v, float* wq, float* wd, float* round_loss_out, void* wq_planes, void* wd_planes,
                      float wq_plane_scale, float wd_plane_scale, void* stream) {
    if (int rc = check_desc(d, "rdo_adaround_step")) return rc;
    RDO_REQUIRE(w && delta && zp && slabs && nsplit >= 1 && sched && iter_ptr && alpha && adam_m && adam_v && wq,
                "rdo_adaround_step: null pointer");
    AdaArgs a{};
    a.d = *d; a.w = w; a.delta = delta; a.zp = zp; a.slabs = slabs; a.nsplit = nsplit; a.grad_scale = grad_scale;
    a.round_weight = round_weight; a.sched = sched; a.iter_ptr = iter_ptr; a.alpha = alpha; a.m = adam_m; a.v = adam_v;
    a.wq = wq; a.wd = wd; a.round_loss_out = round_loss_out; a.mode = 0;
    a.wq_planes = static_cast<unsigned short*>(wq_planes);
    a.wd_planes = wd ? static_cast<unsigned short*>(wd_planes) : nullptr;
    a.wq_pscale = wq_plane_scale; a.wd_pscale = wd_plane_scale; a.ovf = rdo::h2_overflow_flag();
    return run_step(a, stream);
}

static int step_batch_impl(const rdo_ada_step_item* items, int32_t n, int32_t mode, float grad_scale, float round_weight,
                           const rdo_sched_row* sched, const int32_t* iter_ptr, float* round_loss_out, int32_t* advance_iter,
                           int32_t* iter_shadow, const rdo_gather_desc* next, void* stream) {
    RDO_REQUIRE(items && n >= 1 && n <= kMaxBatch, "rdo_adaround_step_batch: bad argument (1 <= n <= %d)", kMaxBatch);
    RDO_REQUIRE(mode >= 0 && mode <= 2, "rdo_adaround_step_batch: mode %d (0 fused step, 1 gradient only, 2 apply)", mode);
    RDO_REQUIRE(mode == 1 || (sched && iter_ptr), "rdo_adaround_step_batch: schedule / iteration counter missing");
    RDO_REQUIRE(!(advance_iter && iter_shadow), "rdo_adaround_step_batch: advance_iter and iter_shadow are alternatives");
    RDO_REQUIRE(!iter_shadow || iter_ptr, "rdo_adaround_step_batch: iter_shadow needs iter_ptr");
    AdaBatch b{}, bw{};
    const int w1_min = rdo::tuning(rdo::T_ADA_W1_MIN);
    b.w1_min = bw.w1_min = w1_min;
    int blocks = 0, wblocks = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const rdo_ada_step_item& it = items[i];
        if (int rc = check_desc(&it.d, "rdo_adaround_step_batch")) return rc;
        RDO_REQUIRE(it.d.numel % 4 == 0, "rdo_adaround_step_batch: numel %ld of item %d is not a multiple of 4", (long)it.d.numel, i);
        RDO_REQUIRE(it.w && it.delta && it.zp && it.alpha, "rdo_adaround_step_batch: null pointer in item %d", i);
        RDO_REQUIRE(mode == 2 || (it.slabs && it.nsplit >= 1), "rdo_adaround_step_batch: item %d has no gradient slabs", i);
        RDO_REQUIRE(mode == 0 || it.dalpha, "rdo_adaround_step_batch: item %d has no dalpha buffer", i);
        RDO_REQUIRE(mode == 1 || (it.adam_m && it.adam_v && it.wq), "rdo_adaround_step_batch: null optimiser state in item %d", i);
        AdaArgs& a = b.a[i];
        a.d = it.d; a.w = it.w; a.delta = it.delta; a.zp = it.zp; a.slabs = it.slabs; a.nsplit = it.nsplit;
        a.dalpha_in = it.dalpha; a.dalpha_out = it.dalpha;
        a.grad_scale = grad_scale; a.round_weight = round_weight; a.sched = sched; a.iter_ptr = iter_ptr;
        a.alpha = it.alpha; a.m = it.adam_m; a.v = it.adam_v; a.wq = it.wq; a.wd = it.wd; a.round_loss_out = round_loss_out; a.mode = mode;
        a.wq_planes = static_cast<unsigned short*>(it.wq_planes);
        a.wd_planes = it.wd ? static_cast<unsigned short*>(it.wd_planes) : nullptr;
        a.wq_pscale = it.wq_plane_scale; a.wd_pscale = it.wd_plane_scale; a.ovf = rdo::h2_overflow_flag();
        if (mode != 1 && (it.lin_fwd_planes || it.lin_bwd_planes)) {
            RDO_REQUIRE(it.d.rows % 32 == 0 && (it.d.numel / it.d.rows) % 32 == 0 && it.lin_plane_scale > 0.f,
                        "rdo_adaround_step_batch: item %d: Linear planes need rows and inner in blocks of 32 and a positive scale", i);
            a.lin_fwd = static_cast<unsigned short*>(it.lin_fwd_planes);
            a.lin_bwd = static_cast<unsigned short*>(it.lin_bwd_planes);
            a.lin_pscale = it.lin_plane_scale;
        }
        b.tile[i] = tile_ok(it.d, mode) ? 1 : 0;
        blocks += b.tile[i] ? (int)tile_blocks(it.d) : (int)grid_for((it.nsplit >= w1_min && mode != 2) ? it.d.numel : it.d.numel / 4);
        b.blk_end[i] = blocks;
        bytes += 4.0 * it.d.numel * ((mode == 2 ? 1 : it.nsplit) + (mode == 1 ? 3.0 : 9.0));
        if (mode != 1 && it.wd && it.d.Cin > 0 && !b.tile[i]) {
            bw.a[bw.n] = a;
            wblocks += (int)(rdo::ceil_div(it.d.rows, 32) * it.d.KH * it.d.KW * rdo::ceil_div(it.d.Cin, 32));
            bw.blk_end[bw.n++] = wblocks;
        }
    }
    b.n = n;
    b.iter_shadow = iter_shadow;
    b.gather_beg = blocks;
    if (next) {
        RDO_REQUIRE(mode != 1 && iter_ptr, "rdo_adaround_step_batch_gather: the gather rides on a fused step or an apply launch");
        RDO_REQUIRE(next->cache_q && next->cache_fp && next->idx_table && (next->out || next->out_planes) && next->n_iters > 0,
                    "rdo_adaround_step_batch_gather: null pointer in the gather descriptor");
        RDO_REQUIRE(next->B > 0 && next->batch_offset >= 0 && next->per_image > 0 && next->per_image % 4 == 0 &&
                    (long)(next->batch_offset + next->B) * next->per_image < (1L << 32), "rdo_adaround_step_batch_gather: bad gather shape");
        RDO_REQUIRE(!next->out_planes || (next->C > 0 && next->C % 16 == 0 && next->per_image % next->C == 0 && next->out_scale > 0.f),
                    "rdo_adaround_step_batch_gather: the plane form needs C %% 16 == 0 and a positive power-of-two scale");
        RDO_REQUIRE(next->out_planes || next->out, "rdo_adaround_step_batch_gather: no destination");
        RDO_REQUIRE(next->prob >= 0.f && next->prob <= 1.f, "rdo_adaround_step_batch_gather: prob out of [0,1]");
        const double t = floor((double)next->prob * 4294967296.0);
        const unsigned long long thr = (unsigned long long)(t > 4294967296.0 ? 4294967296.0 : t);
        b.g = rdo::gq::Gather{next->cache_q, next->cache_fp, next->idx_table, next->n_iters, next->B, next->batch_offset, (long)next->per_image,
                              next->C, thr, next->seed, next->out,
                              rdo::gq::H2Out{static_cast<unsigned short*>(next->out_planes), next->out_scale,
                                             next->overflow_flag ? reinterpret_cast<int*>(next->overflow_flag) : rdo::h2_overflow_flag()}};
        long gb = rdo::gq::gather_blocks(b.g);
        const long cap = b.g.planes.p ? 8192 : 2048;            // the grids of the stand-alone gather kernels
        blocks += (int)(gb > cap ? cap : gb);
        bytes += (8.0 + (next->out ? 4.0 : 0.0) + (next->out_planes ? 4.0 : 0.0)) * next->B * (double)next->per_image;
    }
    const char* tag = next ? "ada_step_gather" : "ada_step";
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(ada_step_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
            if (bw.n > 0)
                hipLaunchKernelGGL(wd_transpose_batch_kernel, dim3((unsigned)wblocks), dim3(256), 0, s, bw, advance_iter);
            else if (advance_iter)
                hipLaunchKernelGGL(iter_advance1_kernel, dim3(1), dim3(1), 0, s, advance_iter);
            return rdo::check_launch("ada_step_batch");
        },
        stream, tag, 0.0, bytes);
}

int rdo_adaround_step_batch(const rdo_ada_step_item* items, int32_t n, int32_t mode, float grad_scale, float round_weight,
                            const rdo_sched_row* sched, const int32_t* iter_ptr, float* round_loss_out, int32_t* advance_iter,
                            int32_t* iter_shadow, void* stream) {
    return step_batch_impl(items, n, mode, grad_scale, round_weight, sched, iter_ptr, round_loss_out, advance_iter, iter_shadow, nullptr, stream);
}

int rdo_adaround_step_batch_gather(const rdo_ada_step_item* items, int32_t n, int32_t mode, float grad_scale, float round_weight,
                                   const rdo_sched_row* sched, const int32_t* iter_ptr, float* round_loss_out, int32_t* iter_shadow,
                                   const rdo_gather_desc* next, void* stream) {
    RDO_REQUIRE(next != nullptr, "rdo_adaround_step_batch_gather: null gather descriptor");
    return step_batch_impl(items, n, mode, grad_scale, round_weight, sched, iter_ptr, round_loss_out, nullptr, iter_shadow, next, stream);
}

int rdo_adaround_grad(const rdo_ada_desc* d, const float* w, const float* alpha, const float* delta, const float* zp,
                      const float* slabs, int nsplit, float* dalpha, void* stream) {
    if (int rc = check_desc(d, "rdo_adaround_grad")) return rc;
    RDO_REQUIRE(w && alpha && delta && zp && slabs && nsplit >= 1 && dalpha, "rdo_adaround_grad: null pointer");
    AdaArgs a{};
    a.d = *d; a.w = w; a.delta = delta; a.zp = zp; a.slabs = slabs; a.nsplit = nsplit; a.alpha = const_cast<float*>(alpha);
    a.dalpha_out = dalpha; a.mode = 1;
    return run_step(a, stream);
}

int rdo_adaround_apply(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, const float* dalpha,
                       float grad_scale, float round_weight, const rdo_sched_row* sched, const int32_t* iter_ptr, float* alpha,
                       float* adam_m, float* adam_v, float* wq, float* wd, float* round_loss_out, void* wq_planes, void* wd_planes,
                       float wq_plane_scale, float wd_plane_scale, void* stream) {
    if (int rc = check_desc(d, "rdo_adaround_apply")) return rc;
    RDO_REQUIRE(w && delta && zp && dalpha && sched && iter_ptr && alpha && adam_m && adam_v && wq,
                "rdo_adaround_apply: null pointer");
    AdaArgs a{};
    a.d = *d; a.w = w; a.delta = delta; a.zp = zp; a.dalpha_in = dalpha; a.grad_scale = grad_scale;
    a.round_weight = round_weight; a.sched = sched; a.iter_ptr = iter_ptr; a.alpha = alpha; a.m = adam_m; a.v = adam_v;
    a.wq = wq; a.wd = wd; a.round_loss_out = round_loss_out; a.mode = 2;
    a.wq_planes = static_cast<unsigned short*>(wq_planes);
    a.wd_planes = wd ? static_cast<unsigned short*>(wd_planes) : nullptr;
    a.wq_pscale = wq_plane_scale; a.wd_pscale = wd_plane_scale; a.ovf = rdo::h2_overflow_flag();
    return run_step(a, stream);
}

}  // extern "C"
