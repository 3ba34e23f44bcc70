// K12 -- entropy-model likelihoods and the rate/distortion reductions, element-wise on NHWC tensors.
//
// Restates (CompressAI 1.2.4 definitions, pinned only against the oracle's restatement -- SURVEY 8c "parity unpinned"):
//   EntropyBottleneck.forward (eval): z^ = round(z - median) + median ; p(z^) = |sigmoid(s*F(z^+.5)) - sigmoid(s*F(z^-.5))|,
//       F = per-channel 1-3-3-3-3-1 net with softplus(matrix) x + bias (+ tanh(factor) * tanh(.) between layers), floor 1e-9
//   GaussianConditional.forward (eval): y^ = round(y - mu) + mu ; p = Phi((.5-|y^-mu|)/s) - Phi((-.5-|y^-mu|)/s),
//       s = max(sigma, 0.11), Phi(x) = erfc(-x/sqrt2)/2, floor 1e-9
// Call sites in the reference: model forward inside save_inp_oup_data (quantization/utils.py:227,241) and Test_kodak
// (test_datasets.py:95); bpp / MSE sums of losses/losses.py:15-35 and test_datasets.py:21-33.
#include "rdo_common.h"

namespace {

constexpr int kParamsPerChannel = 58;   // 33 softplus(matrix) + 13 bias + 12 tanh(factor)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// p: [33 matrices | 13 biases | 12 factors] of one channel (matrices already softplus'ed, factors already tanh'ed)
__device__ __forceinline__ float eb_logits(const float* p, float x) {
    const float* M = p;
    const float* Bv = p + 33;
    const float* Fv = p + 46;
    float h[3], g[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {              // layer 0: 1 -> 3
        float v = M[o] * x + Bv[o];
        h[o] = v + Fv[o] * tanhf(v);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {              // layers 1..3: 3 -> 3
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float v = Bv[3 + 3 * l + o];
#pragma unroll
            for (int i = 0; i < 3; ++i) v += M[3 + 9 * l + 3 * o + i] * h[i];
            g[o] = v + Fv[3 + 3 * l + o] * tanhf(v);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) h[o] = g[o];
    }
    float v = Bv[12];                          // layer 4: 3 -> 1
#pragma unroll
    for (int i = 0; i < 3; ++i) v += M[30 + i] * h[i];
    return v;
}

__global__ __launch_bounds__(256) void eb_fwd_kernel(const float* z, const float* params, const float* medians, long n, int C,
                                                     float* zhat, float* lik) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float med = medians[c];
        const float q = rintf(z[i] - med) + med;
        const float* p = params + (long)c * kParamsPerChannel;
        const float lo = eb_logits(p, q - 0.5f), hi = eb_logits(p, q + 0.5f);
        const float t = lo + hi;
        const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
        zhat[i] = q;
        lik[i] = fmaxf(fabsf(sigm(sgn * hi) - sigm(sgn * lo)), 1e-9f);
    }
}

// value and derivative dF/dx of the cumulative-logits net (forward-mode through the five layers)
__device__ __forceinline__ float eb_logits_d(const float* p, float x, float& dFdx) {
    const float* M = p;
    const float* Bv = p + 33;
    const float* Fv = p + 46;
    float h[3], dh[3], g[3], dg[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float v = M[o] * x + Bv[o];
        const float th = tanhf(v);
        h[o] = v + Fv[o] * th;
        dh[o] = M[o] * (1.f + Fv[o] * (1.f - th * th));
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float v = Bv[3 + 3 * l + o], dv = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v += M[3 + 9 * l + 3 * o + i] * h[i];
                dv += M[3 + 9 * l + 3 * o + i] * dh[i];
            }
            const float th = tanhf(v);
            g[o] = v + Fv[3 + 3 * l + o] * th;
            dg[o] = dv * (1.f + Fv[3 + 3 * l + o] * (1.f - th * th));
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) { h[o] = g[o]; dh[o] = dg[o]; }
    }
    float v = Bv[12], dv = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v += M[30 + i] * h[i];
        dv += M[30 + i] * dh[i];
    }
    dFdx = dv;
    return v;
}

// d(grad_scale * -log2 p)/dz^ of the factorised prior (z^ = the rounded value: the straight-through estimator hands it to z):
// p = |sigmoid(s hi) - sigmoid(s lo)|, hi = F(z^ + .5), lo = F(z^ - .5);  dp/dz^ = sign(.) * s * (sig'(s hi) F'(z^ + .5) - sig'(s lo) F'(z^ - .5));
// zero where the 1e-9 floor is active.
__global__ __launch_bounds__(256) void eb_bwd_kernel(const float* zhat, const float* params, long n, int C, float gscale, float* dz) {
    const float inv_ln2 = 1.4426950408889634f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float q = zhat[i];
        const float* p = params + (long)c * kParamsPerChannel;
        float dlo, dhi;
        const float lo = eb_logits_d(p, q - 0.5f, dlo), hi = eb_logits_d(p, q + 0.5f, dhi);
        const float t = lo + hi;
        const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
        const float sh = sigm(sgn * hi), sl = sigm(sgn * lo);
        const float diff = sh - sl;
        const float pr = fabsf(diff);
        float g = 0.f;
        if (pr > 1e-9f) {
            const float dp = (diff > 0.f ? 1.f : -1.f) * sgn * (sh * (1.f - sh) * dhi - sl * (1.f - sl) * dlo);
            g = -gscale * inv_ln2 / pr * dp;
        }
        dz[i] = g;
    }
}

__device__ __forceinline__ float std_cum(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

__global__ __launch_bounds__(256) void gc_fwd_kernel(const float* y, const float* scales, const float* means, long n,
                                                     float scale_bound, float* yhat, float* lik) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float mu = means ? means[i] : 0.f;
        const float q = rintf(y[i] - mu) + mu;
        const float v = fabsf(q - mu);
        const float s = fmaxf(scales[i], scale_bound);
        if (yhat) yhat[i] = q;
        lik[i] = fmaxf(std_cum((0.5f - v) / s) - std_cum((-0.5f - v) / s), 1e-9f);
    }
}

// d(-log2 p)/d(sigma), d(-log2 p)/d(mu) of the Gaussian conditional (y^ treated as constant, i.e. the STE of the rounding):
// p = Phi(a) - Phi(b), a = (.5 - v)/s, b = (-.5 - v)/s, v = |y^ - mu|; dp/ds = -(phi(a) a - phi(b) b)/s,
// dp/dv = -(phi(a) - phi(b))/s, dv/dmu = -sign(y^ - mu); gradients vanish where the floors (scale bound, 1e-9) are active.
__global__ __launch_bounds__(256) void gc_bwd_kernel(const float* yhat, const float* scales, const float* means, long n,
                                                     float scale_bound, float gscale, float* dscales, float* dmeans) {
    const float inv_sqrt2pi = 0.3989422804014327f, inv_ln2 = 1.4426950408889634f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float mu = means ? means[i] : 0.f;
        const float d = yhat[i] - mu;
        const float v = fabsf(d);
        const float sr = scales[i];
        const float s = fmaxf(sr, scale_bound);
        const float a = (0.5f - v) / s, b = (-0.5f - v) / s;
        const float p = std_cum(a) - std_cum(b);
        const float pa = inv_sqrt2pi * expf(-0.5f * a * a), pb = inv_sqrt2pi * expf(-0.5f * b * b);
        float ds = 0.f, dm = 0.f;
        if (p > 1e-9f) {
            const float coef = -gscale * inv_ln2 / p;             // d(-log2 p) = -dp / (p ln 2)
            if (sr >= scale_bound) ds = coef * (-(pa * a - pb * b) / s);
            const float dpdv = -(pa - pb) / s;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            dm = coef * dpdv * (-sgn);
        }
        if (dscales) dscales[i] = ds;
        if (dmeans) dmeans[i] = dm;
    }
}

// out[0] += sum(-log2(lik)) * scale   (bpp numerator);  out[0] += sum((a-b)^2) * scale  (MSE numerator)
__global__ __launch_bounds__(256) void neg_log2_sum_kernel(const float* lik, long n, float scale, float* out) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc -= log2f(lik[i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

__global__ __launch_bounds__(256) void sq_diff_sum_kernel(const float* a, const float* b, long n, float scale, int clamp01,
                                                          float* out) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x = a[i];
        if (clamp01) x = fminf(fmaxf(x, 0.f), 1.f);
        const float d = x - b[i];
        acc += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

inline unsigned grid_for(long n) {
    long g = rdo::ceil_div(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" {

int rdo_factorized_likelihood_fwd(const float* z, const float* params, const float* medians, int64_t n, int32_t C, float* zhat,
                                  float* lik, void* stream) {
    RDO_REQUIRE(z && params && medians && zhat && lik && n > 0 && C > 0, "rdo_factorized_likelihood_fwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(eb_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, z, params, medians, (long)n, C, zhat, lik);
            return rdo::check_launch("factorized_likelihood_fwd");
        },
        stream, "entropy", 0.0, 12.0 * n);
}

int rdo_factorized_likelihood_bwd(const float* zhat, const float* params, int64_t n, int32_t C, float grad_scale, float* dz, void* stream) {
    RDO_REQUIRE(zhat && params && dz && n > 0 && C > 0, "rdo_factorized_likelihood_bwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(eb_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, zhat, params, (long)n, C, grad_scale, dz);
            return rdo::check_launch("factorized_likelihood_bwd");
        },
        stream, "entropy", 0.0, 8.0 * n);
}

int rdo_gaussian_likelihood_fwd(const float* y, const float* scales, const float* means, int64_t n, float scale_bound, float* yhat,
                                float* lik, void* stream) {
    RDO_REQUIRE(y && scales && lik && n > 0, "rdo_gaussian_likelihood_fwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gc_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, scales, means, (long)n, scale_bound, yhat, lik);
            return rdo::check_launch("gaussian_likelihood_fwd");
        },
        stream, "entropy", 0.0, 20.0 * n);
}

int rdo_gaussian_likelihood_bwd(const float* yhat, const float* scales, const float* means, int64_t n, float scale_bound,
                                float grad_scale, float* dscales, float* dmeans, void* stream) {
    RDO_REQUIRE(yhat && scales && (dscales || dmeans) && n > 0, "rdo_gaussian_likelihood_bwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gc_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, yhat, scales, means, (long)n, scale_bound,
                               grad_scale, dscales, dmeans);
            return rdo::check_launch("gaussian_likelihood_bwd");
        },
        stream, "entropy", 0.0, 20.0 * n);
}

int rdo_neg_log2_sum(const float* lik, int64_t n, float scale, float* out, void* stream) {
    RDO_REQUIRE(lik && out && n > 0, "rdo_neg_log2_sum: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(neg_log2_sum_kernel, dim3(grid_for(n)), dim3(256), 0, s, lik, (long)n, scale, out);
            return rdo::check_launch("neg_log2_sum");
        },
        stream, "entropy", 0.0, 4.0 * n);
}

int rdo_sq_diff_sum(const float* a, const float* b, int64_t n, float scale, int32_t clamp01, float* out, void* stream) {
    RDO_REQUIRE(a && b && out && n > 0, "rdo_sq_diff_sum: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(sq_diff_sum_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, (long)n, scale, clamp01, out);
            return rdo::check_launch("sq_diff_sum");
        },
        stream, "entropy", 0.0, 8.0 * n);
}

}  // extern "C"
