// MS-SSIM building blocks for the evaluation path (reference: `ms_ssim` of pytorch_msssim, test_datasets.py:25-27,
// losses/losses.py:27,54): one scale of SSIM / contrast-structure statistics with an 11-tap separable Gaussian window (valid
// padding) and the 2x2 average pooling between scales.  Planes are [planes][H][W] fp32 (NCHW images: planes = B*C).
// A 16 x 16 output tile per workgroup: the 26 x 26 input halo of both images goes to LDS, rows are filtered into five 26 x 16
// LDS planes (mu1, mu2, E[xx], E[yy], E[xy]), then columns; per-plane sums accumulate with one atomic per workgroup.
#include "rdo_common.h"

namespace {

constexpr int WIN = 11, TILE = 16, HALO = TILE + WIN - 1;

struct SsimArgs {
    const float* x;
    const float* y;
    int planes, H, W;
    float c1, c2;
    float win[WIN];
    float* ssim_sum;   // [planes]
    float* cs_sum;     // [planes]
};

__global__ __launch_bounds__(256) void ssim_level_kernel(SsimArgs a) {
    __shared__ float sx[HALO][HALO + 1], sy[HALO][HALO + 1];
    __shared__ float r[5][HALO][TILE + 1];
    __shared__ float red[2][4];
    const int Ho = a.H - WIN + 1, Wo = a.W - WIN + 1;
    const int plane = blockIdx.z, h0 = blockIdx.y * TILE, w0 = blockIdx.x * TILE;
    const float* px = a.x + (long)plane * a.H * a.W;
    const float* py = a.y + (long)plane * a.H * a.W;
    for (int e = threadIdx.x; e < HALO * HALO; e += 256) {
        const int i = e / HALO, j = e - i * HALO;
        const int h = h0 + i, w = w0 + j;
        const bool ok = h < a.H && w < a.W;
        sx[i][j] = ok ? px[(long)h * a.W + w] : 0.f;
        sy[i][j] = ok ? py[(long)h * a.W + w] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HALO * TILE; e += 256) {
        const int i = e / TILE, j = e - i * TILE;
        float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float g = a.win[k], u = sx[i][j + k], v = sy[i][j + k];
            m1 += g * u; m2 += g * v; xx += g * u * u; yy += g * v * v; xy += g * u * v;
        }
        r[0][i][j] = m1; r[1][i][j] = m2; r[2][i][j] = xx; r[3][i][j] = yy; r[4][i][j] = xy;
    }
    __syncthreads();
    const int i = threadIdx.x / TILE, j = threadIdx.x % TILE;
    float ssim = 0.f, cs = 0.f;
    if (h0 + i < Ho && w0 + j < Wo) {
        float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float g = a.win[k];
            m1 += g * r[0][i + k][j]; m2 += g * r[1][i + k][j]; xx += g * r[2][i + k][j]; yy += g * r[3][i + k][j];
            xy += g * r[4][i + k][j];
        }
        const float s11 = xx - m1 * m1, s22 = yy - m2 * m2, s12 = xy - m1 * m2;
        cs = (2.f * s12 + a.c2) / (s11 + s22 + a.c2);
        ssim = ((2.f * m1 * m2 + a.c1) / (m1 * m1 + m2 * m2 + a.c1)) * cs;
    }
    for (int o = 32; o > 0; o >>= 1) { ssim += __shfl_down(ssim, o, 64); cs += __shfl_down(cs, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ssim; red[1][threadIdx.x >> 6] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv = 1.0f / ((float)Ho * (float)Wo);
        atomicAdd(a.ssim_sum + plane, (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv);
        atomicAdd(a.cs_sum + plane, (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv);
    }
}

// F.avg_pool2d(x, 2, padding=(H%2, W%2)) with count_include_pad=True: out[h][w] = sum of the 2x2 window (zeros outside) / 4
__global__ __launch_bounds__(256) void avg_pool2_kernel(const float* x, int planes, int H, int W, int ph, int pw, int Ho, int Wo, float* out) {
    const long total = (long)planes * Ho * Wo;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int w = (int)(t % Wo);
        const int h = (int)((t / Wo) % Ho);
        const long p = t / ((long)Wo * Ho);
        float acc = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int hi = 2 * h - ph + dh, wi = 2 * w - pw + dw;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) acc += x[(p * H + hi) * W + wi];
            }
        out[t] = 0.25f * acc;
    }
}

}  // namespace

extern "C" {

int rdo_ssim_level(const float* x, const float* y, int32_t planes, int32_t H, int32_t W, const float* window11, float c1, float c2,
                   float* ssim_sum, float* cs_sum, void* stream) {
    RDO_REQUIRE(x && y && window11 && ssim_sum && cs_sum && planes > 0, "rdo_ssim_level: bad argument");
    RDO_REQUIRE(H >= WIN && W >= WIN, "rdo_ssim_level: the %dx%d plane is smaller than the 11-tap window", H, W);
    SsimArgs a;
    a.x = x; a.y = y; a.planes = planes; a.H = H; a.W = W; a.c1 = c1; a.c2 = c2; a.ssim_sum = ssim_sum; a.cs_sum = cs_sum;
    for (int k = 0; k < WIN; ++k) a.win[k] = window11[k];       // host pointer: 11 floats copied into the launch arguments
    const int Ho = H - WIN + 1, Wo = W - WIN + 1;
    return rdo::dispatch(
        [=](hipStream_t s) {
            if (hipMemsetAsync(a.ssim_sum, 0, sizeof(float) * a.planes, s) != hipSuccess ||
                hipMemsetAsync(a.cs_sum, 0, sizeof(float) * a.planes, s) != hipSuccess)
                return rdo::set_error(RDO_EHIP, "rdo_ssim_level: memset failed");
            dim3 grid((unsigned)rdo::ceil_div(Wo, TILE), (unsigned)rdo::ceil_div(Ho, TILE), (unsigned)a.planes);
            hipLaunchKernelGGL(ssim_level_kernel, grid, dim3(256), 0, s, a);
            return rdo::check_launch("ssim_level");
        },
        stream, "ssim_level", 0.0, 8.0 * planes * H * W);
}

int rdo_avg_pool2(const float* x, int32_t planes, int32_t H, int32_t W, float* out, void* stream) {
    RDO_REQUIRE(x && out && planes > 0 && H > 0 && W > 0, "rdo_avg_pool2: bad argument");
    const int ph = H % 2, pw = W % 2;
    const int Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
    return rdo::dispatch(
        [=](hipStream_t s) {
            long g = rdo::ceil_div((long)planes * Ho * Wo, 256);
            hipLaunchKernelGGL(avg_pool2_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, x, planes, H, W, ph, pw, Ho, Wo, out);
            return rdo::check_launch("avg_pool2");
        },
        stream, "avg_pool2", 0.0, 5.0 * planes * H * W);
}

}  // extern "C"
