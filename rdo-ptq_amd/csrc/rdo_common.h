// Shared host-side plumbing for librdoptq_hip.so: error reporting, op recording for the unit executor.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rdo_ptq_hip.h"

namespace rdo {

struct Op {
    std::function<int(hipStream_t)> fn;
    const char* tag;   // kernel family / tile instance, static string
    double flops;      // algorithmic FLOPs of this launch (0 for byte-moving ops)
    double bytes;      // algorithmic HBM bytes of this launch
};

struct Recorder {
    bool active = false;
    std::vector<Op>* sink = nullptr;
};
Recorder& recorder();                      // thread-local
int set_error(int code, const char* fmt, ...);

// Run `op` now on `stream`, or append it to the plan being recorded on this thread.
inline int dispatch(std::function<int(hipStream_t)> fn, void* stream, const char* tag = "op", double flops = 0.0,
                    double bytes = 0.0) {
    Recorder& r = recorder();
    if (r.active) {
        r.sink->push_back(Op{std::move(fn), tag, flops, bytes});
        return RDO_OK;
    }
    return fn(reinterpret_cast<hipStream_t>(stream));
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(RDO_EHIP, "%s: %s", what, hipGetErrorString(e));
    return RDO_OK;
}

#define RDO_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return rdo::set_error(RDO_EINVAL, __VA_ARGS__); \
    } while (0)

constexpr int kWave = 64;

// "Has this once-per-device set-up been done on the CURRENT device?" -- hipFuncSetAttribute (the > 64 KiB dynamic-LDS limit of the big
// tiles) applies to the device that is current when it is called, so a process that drives a second GPU must repeat it there.
struct PerDevice {
    std::atomic<uint64_t> bits{0};
    static uint64_t bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool done() const { return (bits.load(std::memory_order_acquire) & bit()) != 0; }
    void mark() { bits.fetch_or(bit(), std::memory_order_release); }
};

// Process-wide kernel-variant switches (rdo_set_tuning / rdo_get_tuning; initialised once from the RDO_* environment variables
// of the same meaning so that command-line A/B runs keep working).
enum Tune { T_WGRAD_X6_W8 = 0, T_CONV_X6, T_FWD_X6_VER, T_XCD, T_X6P_ABLATE, T_GRAPH_UNROLL, T_TAIL_GRID, T_X6P_HALO, T_WGRAD_P3_ROW, T_THIN_MFMA, T_H2_STAGGER, T_ADA_W1_MIN, T_H2_K32, T_WGRAD_SUB, T_H2_N48, T_COUNT };
int tuning(Tune t);

// bf16x6 weight planes ("fragment order"): element (co, kh, kw, ci) of a conv weight [Cout][KH][KW][Cin] lives at
//   plane + ((((ci / 16) * KH + kh) * KW + kw) * Cout + co) * 16 + ci % 16
// i.e. [channel slice of 16][kh][kw][Cout][16]: the 192 x 16 weight tile of one K stage of conv_fwd_x6 is one contiguous run, so
// the LDS-DMA of a stage reads whole cache lines (with the [Cout][KH][KW][Cin] order every 32-byte piece came from its own line:
// 2-4x the L2 -> CU traffic, measured as 10 % of the kernel time).  Cin % 16 != 0 (never consumed by the x6 kernels): linear order.
__host__ __device__ inline long frag_index(long e, int Cout, int KH, int KW, int Cin) {
    if (Cin % 16 != 0) return e;
    long t = e / Cin;
    const int ci = (int)(e - t * Cin);
    const int kw = (int)(t % KW);
    t /= KW;
    const int kh = (int)(t % KH);
    const long co = t / KH;
    return (((((long)(ci >> 4) * KH + kh) * KW + kw) * Cout + co) << 4) + (ci & 15);
}
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- H2 tensors: an fp32 value as an exact two-way fp16 split of its power-of-two-scaled self ---------------------------------------
//   x * s = h1 + h2 (+ at most 2^-24 |x s|),  h1 = fp16_rne(x s),  h2 = fp16_rne(x s - h1)         (x s - h1 is exact in fp32)
// A product of two such values needs THREE fp16 MFMA products (h1 g1 + h1 g2 + h2 g1; h2 g2 <= 2^-22 of the product) where the
// three-way bf16 split needed six, and the operands take 4 bytes per element instead of 6.  fp16 has 5 exponent bits, hence the
// per-tensor scale s (a power of two: exact): the engine picks it from a probe iteration so that max |x s| ~ 2^7; results stay at
// fp32-chain accuracy for max |x s| anywhere in [2^-2, 2^15] (tools/f16_probe.hip: rms error 1.1e-7 of max |C| against 1.5e-7 for the
// bf16 six-product form and 1.8e-7 for an fp32 fma chain; MFMA honours fp16 denormals).  |x s| > 65504 cannot be represented: the
// producers raise a sticky device flag (rdo_h2_overflow) instead of failing silently.
int* h2_overflow_flag();                   // device pointer of the sticky flag on the current device (allocated on first use)
int32_t* take_iter_publish();              // the word bound by rdo_iter_bind_publish for the next loss / tail launch (and clears the binding)

#if defined(__HIPCC__)
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
// two values -> packed h1 pair, packed h2 pair.  `bad` accumulates the LARGEST |x s| met, as the fp32 bit pattern with the sign
// cleared (an integer max: inf and NaN patterns are larger than every finite one); h2_report raises the word when that exceeds fp16's
// 65504 -- the word then tells the engine by how much the scale was outgrown (quantization/engine.py, _recover).
constexpr int kH2MaxBits = 0x477FE000;      // bit pattern of 65504.0f
__device__ __forceinline__ int h2_mag(float a, float b, int bad) {
    const int ia = __builtin_bit_cast(int, a) & 0x7FFFFFFF, ib = __builtin_bit_cast(int, b) & 0x7FFFFFFF;
    return max(max(ia, ib), bad);
}
__device__ __forceinline__ void h2_split_pk(float a, float b, float s, unsigned& hi, unsigned& lo, int& bad) {
    const float xa = a * s, xb = b * s;
    const h2_f16x2 h = __builtin_convertvector(h2_f32x2{xa, xb}, h2_f16x2);
    const h2_f32x2 hf = __builtin_convertvector(h, h2_f32x2);
    const h2_f16x2 l = __builtin_convertvector(h2_f32x2{xa - hf[0], xb - hf[1]}, h2_f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    bad = h2_mag(xa, xb, bad);
}
__device__ __forceinline__ float h2_lo(unsigned pk) { return (float)__builtin_bit_cast(h2_f16x2, pk)[0]; }
__device__ __forceinline__ float h2_hi(unsigned pk) { return (float)__builtin_bit_cast(h2_f16x2, pk)[1]; }
// scalar form: element i of planes [2][n]
__device__ __forceinline__ void h2_split_store(float v, float s, unsigned short* planes, long n, long i, int& bad) {
    const float x = v * s;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    planes[i] = __builtin_bit_cast(unsigned short, h);
    planes[n + i] = __builtin_bit_cast(unsigned short, l);
    bad = h2_mag(x, x, bad);
}
// flag[0]: atomic max of the FINITE magnitudes beyond fp16's range (what the scale must grow by); flag[1]: set when a value was inf / NaN
// (after a first overflow everything downstream is: those must not bury the finite magnitude that started it)
__device__ __forceinline__ void h2_report(int bad, int* flag) {
    // flag == nullptr: the default word could not be allocated (rdo_h2_overflow then returns -1)
    if (bad > kH2MaxBits && flag) {
        if (bad < 0x7F800000) atomicMax(flag, bad);
        else atomicOr(flag + 1, 1);
    }
}
// exact (erf) GELU of nn.GELU() (models/layers.py:37,40) and its derivative; shared by the element-wise kernels (swin.hip) and the
// RDO_EPI_GELU / RDO_EPI_GELU_BWD epilogues of the conv kernels
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float v) {
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
    return cdf + v * pdf;
}
#endif

}  // namespace rdo
