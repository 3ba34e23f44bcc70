// Lu2022 transformer path (reference: models/layers.py:87-305, quantization/quant_block.py:330-547): window attention core with
// the cyclic shift / window partition folded into the addressing, LayerNorm backward, GELU.  Windows have <= 64 tokens and head
// dims 12..48: one workgroup owns one (window, head) pair with its Q/K/V tiles in LDS.  The linears around them run on the conv
// kernels as 1x1 convolutions.
//
// Token order: qkv / out are [B, H, W, *] in NATURAL pixel order.  Window (b, wh, ww), token (ih, iw) reads the pixel
// ((wh*ws + ih + shift) % H, (ww*ws + iw + shift) % W): torch.roll(x, -shift) followed by window_partition (layers.py:271-279),
// and writes its output to the same pixel (window_reverse + roll(+shift), :288-294).
// qkv channel layout is that of nn.Linear(dim, 3*dim) reshaped (3, heads, hd) (layers.py:147): c = which*C + head*hd + d.
#include "rdo_common.h"

namespace {

constexpr int NMAX = 64;

struct AttnGeom {
    int B, H, W, C, heads, ws, shift, N, hd, hs;   // hs: LDS row stride (odd)
    float scale;
#ifdef RDO_DIAG
    int diag;                                      // ablation bits (diagnostic build, RDO_ATTN_DIAG): 1 no stores, 2 no products, 4 no loads, 8 no softmax
#endif
};
#ifdef RDO_DIAG
#define ATTN_ABL(bit) (g.diag & (bit))
#else
#define ATTN_ABL(bit) false
#endif

__device__ __forceinline__ int token_pixel(const AttnGeom& g, int win, int tok) {
    const int nww = g.W / g.ws, nwh = g.H / g.ws;
    const int b = win / (nwh * nww);
    const int r = win - b * (nwh * nww);
    const int wh = r / nww, ww = r - wh * nww;
    const int ih = tok / g.ws, iw = tok - ih * g.ws;
    int h = wh * g.ws + ih + g.shift, w = ww * g.ws + iw + g.shift;
    if (h >= g.H) h -= g.H;
    if (w >= g.W) w -= g.W;
    return (b * g.H + h) * g.W + w;
}

// region id of a token in the SHIFTED image (layers.py:239-250): 3 bands per axis
__device__ __forceinline__ int token_region(const AttnGeom& g, int win, int tok) {
    const int nww = g.W / g.ws, nwh = g.H / g.ws;
    const int r = win % (nwh * nww);
    const int wh = r / nww, ww = r - wh * nww;
    const int ih = tok / g.ws, iw = tok - ih * g.ws;
    const int sh = wh * g.ws + ih, sw = ww * g.ws + iw;
    const int rh = sh < g.H - g.ws ? 0 : (sh < g.H - g.shift ? 1 : 2);
    const int rw = sw < g.W - g.ws ? 0 : (sw < g.W - g.shift ? 1 : 2);
    return rh * 3 + rw;
}

// cooperative load of one of q/k/v (or a gradient w.r.t. the output) of this (window, head) into LDS rows of stride hs
__device__ __forceinline__ void load_tile(const AttnGeom& g, const float* src, int row_stride, int ch0, int win, float* dst, float mul) {
    for (int e = threadIdx.x; e < g.N * g.hd; e += 64) {
        const int tok = e / g.hd, d = e - tok * g.hd;
        dst[tok * g.hs + d] = mul * src[(long)token_pixel(g, win, tok) * row_stride + ch0 + d];
    }
}

// out = probs @ v with probs given (after activation quantisation): the second half of the split evaluation path
__global__ __launch_bounds__(64) void win_attn_pv_kernel(const float* qkv, const float* probs, AttnGeom g, float* out) {
    extern __shared__ float lds[];
    float* V = lds;
    const int win = blockIdx.x, head = blockIdx.y, i = threadIdx.x;
    load_tile(g, qkv, 3 * g.C, 2 * g.C + head * g.hd, win, V, 1.f);
    __syncthreads();
    if (i >= g.N) return;
    const float* pr = probs + (((long)win * g.N + i) * g.N) * g.heads + head;
    float S[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) S[j] = j < g.N ? pr[(long)j * g.heads] : 0.f;
    float* o = out + (long)token_pixel(g, win, i) * g.C + head * g.hd;
    for (int d = 0; d < g.hd; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
            if (j < g.N) acc += S[j] * V[j * g.hs + d];
        o[d] = acc;
    }
}

// ---- fused core: one 4-wave workgroup per (window, head); the small GEMMs (Q K^T, P V and, backward, dO V^T, dS K, dS^T Q,
// P^T dO) run on v_mfma_f32_32x32x2_f32 (exact fp32 products) straight from fp32 LDS images, wave w owning the 32 x 32 tile
// (w>>1, w&1) of every 64 x 64 / 64 x hd product; softmax rows are handled by one thread each between the GEMM phases.
// Matrices are zero-padded to 64 rows; [64][hs] operands use an odd row stride hs >= hd + 2, score matrices [64][65].
//   backward: dP = dO V^T;  D_i = sum_j P_ij dP_ij;  dS = P o (dP - D);  dQ = scale * dS K;  dK = dS^T (scale Q);  dV = P^T dO
// (a first VALU version -- one wave per pair, score row in registers -- took 0.4 / 1.9 ms per launch on the 128^2 maps of the
// full-size model, 71 % of a Lu2022 iteration; see profiles/r01_lu2022.md)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float af4 __attribute__((ext_vector_type(4)));
constexpr int SS = 65;

// C[i][j] += sum_k A(i,k) * B(k,j) for the wave's tile; TA: A stored [k][i]; TB: B stored [j][k] (else [k][j]).
// bcols: valid columns of B when B is [k][j] (reads past it are replaced by 0 so the tile may overhang the matrix)
template <bool TA, bool TB>
__device__ __forceinline__ f32x16 tile_gemm(const float* A, int lda, const float* B, int ldb, int ti, int tj, int K, int bcols) {
    const int l = threadIdx.x & 63, lr = l & 31, lk = l >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ai = 32 * ti + lr, bj = 32 * tj + lr;
    for (int s = 0; s < K; s += 2) {
        const int k = s + lk;
        const float a = TA ? A[k * lda + ai] : A[ai * lda + k];
        float b;
        if (TB) b = B[bj * ldb + k];
        else b = bj < bcols ? B[k * ldb + bj] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// NG independent products over the same K, stepped together: the operand loads of four k pairs of EVERY product are issued before
// the first MFMA of the group (the one-product loop above compiles to  2 x ds_read_b32 -> s_waitcnt lgkmcnt(0) -> v_mfma  per k pair:
// the LDS latency fully exposed behind every 64-cycle MFMA), and consecutive MFMAs go to different accumulators.
// `ld(g, k, a, b)`: operands of product g for reduction index k (the lane's own k = s + lk is passed in).
// KEEP0: accumulator 0 arrives pre-loaded (the score product starts from bias + shift mask, `score_init`).
template <int NG, bool KEEP0 = false, typename LD>
__device__ __forceinline__ void tile_gemm_multi(f32x16 (&acc)[NG], int K, LD ld) {
    const int lk = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int g = KEEP0 ? 1 : 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    int s = 0;
    for (; s + 8 <= K; s += 8) {
        float a[NG][4], b[NG][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < NG; ++g) ld(g, s + 2 * u + lk, a[g][u], b[g][u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u], b[g][u], acc[g], 0, 0, 0);
    }
    for (; s < K; s += 2) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float a, b;
            ld(g, s + lk, a, b);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void tile_to_lds(const f32x16& acc, float* D, int ldd, int ti, int tj) {
    const int l = threadIdx.x & 63, lr = l & 31, lk = l >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) D[(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lk) * ldd + 32 * tj + lr] = acc[r];
}

// Second-stage products, computed TRANSPOSED (O^T = V^T P^T, dQ^T = K^T dS^T, ...: rows = head channels, columns = tokens, reduction over
// the window's 64 tokens) on v_mfma_f32_16x16x4_f32: wave w owns token block w (16 tokens) and ALL channel blocks of every product.  A
// lane's four accumulator values are then four CONSECUTIVE channels of one token -- one 16-byte store (the backward kernel spent a
// quarter of its time issuing 48 dword stores per thread and item: tools/attn_ablate.py).  With 32 x 32 tiles a head of 24 channels kept two of the four
// waves idle for the whole stage and the other two multiplied 32-wide tiles a quarter full; 48 channels wasted a quarter of the stage.
// K slots: lane group kq = lane >> 4 takes k = 16 kq + s (s = 0..15) -- any assignment common to A and B is valid, and this one makes
// every operand read conflict-free ([k][16 consecutive] and [16 consecutive][k] both map the 64 lanes to 64 distinct banks for the odd
// row strides used here).  Always 64 deep: rows past N are zeros in every LDS image.
// lda(g, k, d) = A_g[d][k] (channel d, zero for d >= hd), ldb(g, k, t) = B_g[k][t] (token t)
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <int NG, int NDB, typename LDA, typename LDB>
__device__ __forceinline__ void stage2_gemm(f32x4s (&acc)[NG][NDB], LDA lda, LDB ldb) {
    const int l = threadIdx.x & 63, l16 = l & 15, kq = l >> 4, tok = 16 * (threadIdx.x >> 6) + l16;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int db = 0; db < NDB; ++db) acc[g][db] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < 16; s += 2) {
        float a[2][NG][NDB], b[2][NG];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                b[u][g] = ldb(g, 16 * kq + s + u, tok);
#pragma unroll
                for (int db = 0; db < NDB; ++db) a[u][g][db] = lda(g, 16 * kq + s + u, 16 * db + l16);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int db = 0; db < NDB; ++db) acc[g][db] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][g][db], b[u][g], acc[g][db], 0, 0, 0);
    }
}
// a lane's four accumulator values of channel block db = channels 16 db + 4 kq .. + 3 of token 16 w + l16: one 16-byte store
template <int NDB>
__device__ __forceinline__ void stage2_store(const AttnGeom& g, const f32x4s (&acc)[NDB], float* dst, int row_stride, int ch0, const int* pix, float mul) {
    const int l = threadIdx.x & 63, l16 = l & 15, kq = l >> 4, tok = 16 * (threadIdx.x >> 6) + l16;
    if (tok >= g.N) return;
    if (ATTN_ABL(1) && acc[0][0] != 12345.678f) return;
    float* row = dst + (long)pix[tok] * row_stride + ch0;
    const bool vec = ((g.hd | row_stride | ch0) & 3) == 0;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const int d0 = 16 * db + 4 * kq;
        if (vec) {
            if (d0 < g.hd) *reinterpret_cast<af4*>(row + d0) = af4{mul * acc[db][0], mul * acc[db][1], mul * acc[db][2], mul * acc[db][3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d0 + e < g.hd) row[d0 + e] = mul * acc[db][e];
        }
    }
}

// accumulator image of  bias[head][i][j] (+ -100 where the shift mask separates i and j)  for the wave's score tile: the score product
// starts from it, so the sixteen (L2) loads of a lane are in flight under the tile stash and the products instead of inside the row phase
__device__ __forceinline__ void score_init(const AttnGeom& g, const float* bias, const int* reg, int head, int ti, int tj, f32x16& acc) {
    const int l = threadIdx.x & 63, lr = l & 31, lk = l >> 5;
    const int j = 32 * tj + lr, jc = j < g.N ? j : g.N - 1;
    const int i0 = 32 * ti + 4 * lk;                                   // the lane's rows: i0 + (r & 3) + 8 (r >> 2)
    if (g.N == NMAX) {                                                 // every row and column a token: one base pointer, constant row offsets
        const float* b0 = bias + ((long)head * NMAX + i0) * NMAX + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b0[((r & 3) + 8 * (r >> 2)) * NMAX];
        if (g.shift > 0) {
            const int rj = reg[j];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (reg[i0 + (r & 3) + 8 * (r >> 2)] != rj) acc[r] += -100.0f;
        }
        return;
    }
    const int rj = g.shift > 0 ? reg[jc] : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2), ic = i < g.N ? i : g.N - 1;
        float v = bias[((long)head * g.N + ic) * g.N + jc];
        if (g.shift > 0 && reg[ic] != rj) v += -100.0f;
        acc[r] = v;
    }
}

__device__ __forceinline__ void zero_lds(float* p, int n) {
    for (int e = threadIdx.x; e < n; e += 256) p[e] = 0.f;
}

__device__ __forceinline__ void load_tile256(const AttnGeom& g, const float* src, int row_stride, int ch0, const int* pix, float* dst, float mul) {
    // head dims that are multiples of 4 (the Lu2022 models: 32): 16-byte global loads, one token-pixel computation per quad -- with
    // 4-byte loads a thread issued 8 dependent-address loads per tile (three integer divisions each), and the four tiles of the
    // backward kernel cost more than its five GEMMs
    if (((g.hd | row_stride | ch0) & 3) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int qpt = g.hd >> 2, nq = g.N * qpt;
        for (int e = threadIdx.x; e < nq; e += 256) {
            const int tok = e / qpt, dq = e - tok * qpt;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)pix[tok] * row_stride + ch0 + 4 * dq);
            float* d = dst + tok * g.hs + 4 * dq;            // (hs is odd: scalar LDS stores)
            d[0] = mul * v[0]; d[1] = mul * v[1]; d[2] = mul * v[2]; d[3] = mul * v[3];
        }
        return;
    }
    for (int e = threadIdx.x; e < g.N * g.hd; e += 256) {
        const int tok = e / g.hd, d = e - tok * g.hd;
        dst[tok * g.hs + d] = mul * src[(long)pix[tok] * row_stride + ch0 + d];
    }
}

// softmax of the rows of S (+ bias, + shift mask, reg = the window's token -> mask-region table in LDS) in place, by ALL 256 threads of
// the workgroup: row i = thread / 4, columns j = (thread & 3) + 4 k, k < 16 -- the four threads of a
// row are neighbours in one wave (maximum and sum by two xor-shuffles), a thread's sixteen values stay in registers between the passes,
// and the loops have compile-time bounds.  (`softmax_row` on 64 threads walked a row of LDS three times in run-time loops while three
// quarters of the workgroup waited: the row phase, not the five 64 x 64 x hd products, was most of the attention kernels' time.)
// `dP` != nullptr (backward): also D = sum_j P dP and dS = P (dP - D), written over dP.  Rows / columns beyond N are zeroed in both.
// (bias and shift mask are inside S already: the score product started from them, `score_init`)
__device__ __forceinline__ void softmax_rows256(const AttnGeom& g, float* S, float* dP, int head, float* probs, int win) {
    const int i = threadIdx.x >> 2, part = threadIdx.x & 3;
    const bool rowok = i < g.N;
    float* row = S + i * SS;
    float v[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = part + 4 * k;
        float sv = -3.0e38f;
        if (rowok && j < g.N) sv = row[j];
        v[k] = sv;
        mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = part + 4 * k;
        const float e = (rowok && j < g.N) ? __expf(v[k] - mx) : 0.f;      // v_exp_f32(x log2 e): |error| <= ~|x| 2^-23 of e^x, x in [-100, 0]
        v[k] = e;
        sum += e;
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = rowok ? 1.0f / sum : 0.f;
    float D = 0.f;
    float dp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = part + 4 * k;
        v[k] *= inv;
        row[j] = v[k];
        if (dP) {
            dp[k] = (rowok && j < g.N) ? dP[i * SS + j] : 0.f;
            D += v[k] * dp[k];
        }
        if (probs && rowok && j < g.N) probs[(((long)win * g.N + i) * g.N + j) * g.heads + head] = v[k];
    }
    if (dP) {
        D += __shfl_xor(D, 1, 64);
        D += __shfl_xor(D, 2, 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) dP[i * SS + part + 4 * k] = v[k] * (dp[k] - D);
    }
}

// (window, head) of work item `id`.  Launch ids go round-robin over the 8 XCDs (id % 8), each with its own L2: the heads of one window
// read interleaved pieces of the same qkv rows (head h of a token is hd floats inside a 3C-float row: neighbouring heads share cache
// lines), so they are numbered next to each other INSIDE one XCD -- slot = id / 8 walks (window of this XCD, head) with the head
// fastest.  Window counts that are not a multiple of 8 (toy shapes): plain order.
__device__ __forceinline__ void attn_item(int id, int windows, int heads, int& win, int& head) {
    if (windows & 7) {
        win = id / heads;
        head = id - win * heads;
        return;
    }
    const int xcd = id & 7, slot = id >> 3;
    const int lw = slot / heads;
    head = slot - lw * heads;
    win = xcd * (windows >> 3) + lw;
}

// One tile (Q, K, V or dO of a (window, head) pair) travels global -> registers -> LDS in two steps, so that the loads of the NEXT work
// item are in flight while the current one is multiplied (`fetch` right behind the barrier that publishes the current tiles, `stash`
// at the top of the next item).  Vector form only: head dim and channel offsets multiples of 4 (every Lu2022 shape).
// TILE_IT = ceil(tokens x hd / 4 / 256): quads per thread and tile (hd <= 16: 1, <= 32: 2, <= 48: 3, <= 64: 4) -- a template parameter, the
// prefetch registers of a kernel are 3 or 4 tiles x TILE_IT x 4
// The thread's quads of a tile: token, channel offset, LDS offset -- the same for every tile and every item, computed ONCE per kernel
// (per tile and item they cost two integer divisions by a run-time value each, in `fetch` and again in `stash`: a fifth of the
// backward kernel went into issuing its twelve loads).
template <int TILE_IT>
struct TileMap {
    int tok[TILE_IT], ch[TILE_IT], lds[TILE_IT];
    bool ok[TILE_IT];
    __device__ __forceinline__ void init(const AttnGeom& g) {
        const int qpt = g.hd >> 2, nq = g.N * qpt;
#pragma unroll
        for (int it = 0; it < TILE_IT; ++it) {
            const int e0 = threadIdx.x + 256 * it;
            ok[it] = e0 < nq;
            const int e = ok[it] ? e0 : nq - 1;              // (a lane past the tile re-reads the last quad: loads stay unconditional)
            tok[it] = e / qpt;
            ch[it] = 4 * (e - tok[it] * qpt);
            lds[it] = tok[it] * g.hs + ch[it];
        }
    }
};
// UNCONDITIONAL loads: a load behind a per-lane branch is a basic block of its own, and hipcc waits vmcnt(0) at every join -- twelve serial
// memory round trips per prefetch instead of one.  `rowoff[it]` = pixel of the quad's token x row stride (one table lookup per quad and item)
template <int TILE_IT>
__device__ __forceinline__ void tile_fetch(const AttnGeom& g, const TileMap<TILE_IT>& m, const float* src, const long (&rowoff)[TILE_IT], int ch0,
                                           af4 (&r)[TILE_IT]) {
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        if (ATTN_ABL(4)) { r[it] = af4{0.1f, 0.2f, -0.1f, 0.3f}; continue; }
        r[it] = *reinterpret_cast<const af4*>(src + rowoff[it] + ch0 + m.ch[it]);
    }
}
template <int TILE_IT>
__device__ __forceinline__ void tile_stash(const TileMap<TILE_IT>& m, const af4 (&r)[TILE_IT], float* dst, float mul) {
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        if (m.ok[it]) {
            float* d = dst + m.lds[it];                      // (hs is odd: scalar LDS stores)
            d[0] = mul * r[it][0]; d[1] = mul * r[it][1]; d[2] = mul * r[it][2]; d[3] = mul * r[it][3];
        }
    }
}

// Round 5.  (1) LDS budget: the score matrices are held in the MFMA accumulators until every wave is done with the operands they were
// computed from, then written OVER those operands -- forward: S over Q / K (3 x 64 x hs floats in all: four workgroups per CU at
// hd = 48, two before), backward: P over V (which no later product reads) + dS behind it (71 KiB instead of 85 at hd = 48: two
// workgroups per CU instead of one).  (2) PERSISTENT workgroups: the kernels are latency chains (load tiles -> products -> row softmax ->
// products -> store); with one (window, head) per workgroup all resident workgroups load, then multiply, then store at the same time
// (29 us per pair for 4.4 us of MFMAs on the 128^2 maps).  A workgroup now walks items id, id + grid, ... (same XCD) and fetches the
// next item's tiles into registers right after publishing the current ones: HBM latency and the loads themselves run under the
// products.  PF = false (head dims that are not multiples of 4): tiles loaded in place, no prefetch.
template <bool PF, int TILE_IT>
__global__ __launch_bounds__(256, 3) void win_attn_fwd_mfma_kernel(const float* qkv, const float* bias, AttnGeom g, float* out, float* probs,
                                                                int no_pv, int windows, int nitems) {
    constexpr int NDB = PF ? TILE_IT : 4;                     // 16-channel blocks of a head (TILE_IT = ceil(hd / 16) on the vector path)
    extern __shared__ float lds[];
    float* V = lds;
    float* Q = V + 64 * g.hs;
    float* K = Q + 64 * g.hs;
    float* S = Q;                                             // [64][65] over Q and K (2 x 64 x hs >= 64 x 65 for hs >= 33; else behind V)
    __shared__ int pix[2][NMAX], reg[2][NMAX];               // token -> pixel (roll + window partition) and -> mask region, per item
    const int w = threadIdx.x >> 6, ti = w >> 1, tj = w & 1;
    const int lr = threadIdx.x & 31, ai = 32 * ti + lr, bj = 32 * tj + lr, hs = g.hs;
    const bool full = g.N == NMAX && (g.hd & 1) == 0;         // every row a token, no odd-K zero column: nothing reads the padding
    const int hk = (g.hd + 1) & ~1;
    auto tables = [&](int b, int win) {
        if (threadIdx.x < g.N) {
            pix[b][threadIdx.x] = token_pixel(g, win, threadIdx.x);
            reg[b][threadIdx.x] = g.shift > 0 ? token_region(g, win, threadIdx.x) : 0;
        }
    };
    af4 rq[TILE_IT], rk[TILE_IT], rv[TILE_IT];
    TileMap<TILE_IT> tm;
    tm.init(g);
    auto fetch = [&](int b, int head) {
        long ro3[TILE_IT];
#pragma unroll
        for (int it = 0; it < TILE_IT; ++it) ro3[it] = (long)pix[b][tm.tok[it]] * (3 * g.C);
        tile_fetch(g, tm, qkv, ro3, head * g.hd, rq);
        tile_fetch(g, tm, qkv, ro3, g.C + head * g.hd, rk);
        tile_fetch(g, tm, qkv, ro3, 2 * g.C + head * g.hd, rv);
    };
    int id = blockIdx.x, buf = 0, win, head;
    attn_item(id, windows, g.heads, win, head);
    tables(0, win);
    __syncthreads();
    if (PF) fetch(0, head);
    for (;;) {
        if (!full) {
            zero_lds(lds, 3 * 64 * g.hs);
            __syncthreads();
        }
        if (PF) {
            tile_stash(tm, rq, Q, g.scale);
            tile_stash(tm, rk, K, 1.f);
            tile_stash(tm, rv, V, 1.f);
        } else {
            load_tile256(g, qkv, 3 * g.C, head * g.hd, pix[buf], Q, g.scale);
            load_tile256(g, qkv, 3 * g.C, g.C + head * g.hd, pix[buf], K, 1.f);
            load_tile256(g, qkv, 3 * g.C, 2 * g.C + head * g.hd, pix[buf], V, 1.f);
        }
        f32x16 sc[1];
        score_init(g, bias, reg[buf], head, ti, tj, sc[0]);  // (behind the stash: the prefetch registers are free; its loads land under the barrier)
        const int nid = id + gridDim.x;
        const bool more = nid < nitems;
        int nwin = 0, nhead = 0;
        if (more) {
            attn_item(nid, windows, g.heads, nwin, nhead);
            tables(buf ^ 1, nwin);
        }
        __syncthreads();                                      // tiles of this item and tables of the next are visible
        if (PF && more) fetch(buf ^ 1, nhead);               // in flight while this item is multiplied
        tile_gemm_multi<1, true>(sc, ATTN_ABL(2) ? 0 : hk, [&](int, int k, float& a, float& b) { a = Q[ai * hs + k]; b = K[bj * hs + k]; });
        __syncthreads();                                      // every wave has read Q and K
        tile_to_lds(sc[0], S, SS, ti, tj);
        __syncthreads();
        if (!ATTN_ABL(8)) softmax_rows256(g, S, nullptr, head, probs, win);
        if (!no_pv) {
            __syncthreads();
            f32x4s o[1][NDB];                                 // O^T = V^T P^T: rows = channels, columns = tokens
            if (!ATTN_ABL(2))
                stage2_gemm<1, NDB>(o, [&](int, int k, int d) { return d < g.hd ? V[k * hs + d] : 0.f; }, [&](int, int k, int t) { return S[t * SS + k]; });
            stage2_store<NDB>(g, o[0], out, g.C, head * g.hd, pix[buf], 1.f);
        }
        if (!more) break;
        __syncthreads();                                      // every wave is done with this item's LDS images
        id = nid; win = nwin; head = nhead; buf ^= 1;
    }
}

template <bool PF, int TILE_IT>
__global__ __launch_bounds__(256, 2) void win_attn_bwd_mfma_kernel(const float* qkv, const float* bias, const float* dout, AttnGeom g,
                                                                float* dqkv, int windows, int nitems) {
    constexpr int NDB = PF ? TILE_IT : 4;
    extern __shared__ float lds[];
    float* Q = lds;                       // pre-scaled
    float* K = Q + 64 * g.hs;
    float* dO = K + 64 * g.hs;
    float* V = dO + 64 * g.hs;
    float* P = V;                         // [64][65]: scores, then probabilities -- over V once the dP product has read it
    float* dS = P + 64 * SS;              // [64][65]: dP, then dS
    __shared__ int pix[2][NMAX], reg[2][NMAX];
    const int w = threadIdx.x >> 6, ti = w >> 1, tj = w & 1;
    const int lr = threadIdx.x & 31, ai = 32 * ti + lr, bj = 32 * tj + lr, hs = g.hs;
    const bool full = g.N == NMAX && (g.hd & 1) == 0;
    const int hk = (g.hd + 1) & ~1;
    auto tables = [&](int b, int win) {
        if (threadIdx.x < g.N) {
            pix[b][threadIdx.x] = token_pixel(g, win, threadIdx.x);
            reg[b][threadIdx.x] = g.shift > 0 ? token_region(g, win, threadIdx.x) : 0;
        }
    };
    af4 rq[TILE_IT], rk[TILE_IT], rv[TILE_IT], ro[TILE_IT];
    TileMap<TILE_IT> tm;
    tm.init(g);
    auto fetch = [&](int b, int head) {
        long ro3[TILE_IT], ro1[TILE_IT];
#pragma unroll
        for (int it = 0; it < TILE_IT; ++it) {
            const long px = pix[b][tm.tok[it]];
            ro3[it] = px * (3 * g.C);
            ro1[it] = px * g.C;
        }
        tile_fetch(g, tm, qkv, ro3, head * g.hd, rq);
        tile_fetch(g, tm, qkv, ro3, g.C + head * g.hd, rk);
        tile_fetch(g, tm, qkv, ro3, 2 * g.C + head * g.hd, rv);
        tile_fetch(g, tm, dout, ro1, head * g.hd, ro);
    };
    int id = blockIdx.x, buf = 0, win, head;
    attn_item(id, windows, g.heads, win, head);
    tables(0, win);
    __syncthreads();
    if (PF) fetch(0, head);
    for (;;) {
        f32x16 sd[2];                                                                              // scores Q K^T (from bias + mask) and dP = dO V^T, stepped together
        score_init(g, bias, reg[buf], head, ti, tj, sd[0]);
        if (!full) {
            zero_lds(lds, 4 * 64 * g.hs);
            __syncthreads();
        }
        if (PF) {
            tile_stash(tm, rq, Q, g.scale);
            tile_stash(tm, rk, K, 1.f);
            tile_stash(tm, rv, V, 1.f);
            tile_stash(tm, ro, dO, 1.f);
        } else {
            load_tile256(g, qkv, 3 * g.C, head * g.hd, pix[buf], Q, g.scale);
            load_tile256(g, qkv, 3 * g.C, g.C + head * g.hd, pix[buf], K, 1.f);
            load_tile256(g, qkv, 3 * g.C, 2 * g.C + head * g.hd, pix[buf], V, 1.f);
            load_tile256(g, dout, g.C, head * g.hd, pix[buf], dO, 1.f);
        }
        const int nid = id + gridDim.x;
        const bool more = nid < nitems;
        int nwin = 0, nhead = 0;
        if (more) {
            attn_item(nid, windows, g.heads, nwin, nhead);
            tables(buf ^ 1, nwin);
        }
        __syncthreads();
        if (PF && more) fetch(buf ^ 1, nhead);
        tile_gemm_multi<2, true>(sd, ATTN_ABL(2) ? 0 : hk, [&](int gi, int k, float& a, float& b) {
            if (gi == 0) { a = Q[ai * hs + k]; b = K[bj * hs + k]; }
            else { a = dO[ai * hs + k]; b = V[bj * hs + k]; }
        });
        __syncthreads();                                                                           // every wave has read V
        tile_to_lds(sd[0], P, SS, ti, tj);
        tile_to_lds(sd[1], dS, SS, ti, tj);
        __syncthreads();
        if (!ATTN_ABL(8)) softmax_rows256(g, P, dS, head, nullptr, win);
        __syncthreads();
        {
            // transposed (rows = channels, columns = tokens): dQ^T = scale * K^T dS^T ; dK^T = Qs^T dS (Qs carries the scale) ; dV^T = dO^T P --
            // three independent products, stepped together
            f32x4s d3[3][NDB];
            if (!ATTN_ABL(2))
                stage2_gemm<3, NDB>(d3,
                    [&](int gi, int k, int d) { return d < g.hd ? (gi == 0 ? K : (gi == 1 ? Q : dO))[k * hs + d] : 0.f; },
                    [&](int gi, int k, int t) { return gi == 0 ? dS[t * SS + k] : (gi == 1 ? dS : P)[k * SS + t]; });
            stage2_store<NDB>(g, d3[0], dqkv, 3 * g.C, head * g.hd, pix[buf], g.scale);
            stage2_store<NDB>(g, d3[1], dqkv, 3 * g.C, g.C + head * g.hd, pix[buf], 1.f);
            stage2_store<NDB>(g, d3[2], dqkv, 3 * g.C, 2 * g.C + head * g.hd, pix[buf], 1.f);
        }
        if (!more) break;
        __syncthreads();
        id = nid; win = nwin; head = nhead; buf ^= 1;
    }
}

// ---- LayerNorm backward: one wave per row, a block's 4 waves walk rows with stride; dgamma partials per block (slab layout
// [blocks][C], summed by the AdaRound step like the wgrad slabs).  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma
constexpr int LN_CPL = 8;   // channels per lane: C <= 512
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float* x, const float* gamma, const float* dy, long rows, int C, float eps,
                                                             float* dx, float* dgamma_slabs) {
    __shared__ float part[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[LN_CPL];
#pragma unroll
    for (int k = 0; k < LN_CPL; ++k) dg[k] = 0.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * C;
        const float* gr = dy + row * C;
        float s1 = 0.f;
        for (int c = lane; c < C; c += 64) s1 += xr[c];
        for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)C;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; s2 += d * d; }
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 / (float)C + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < LN_CPL; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const float xh = (xr[c] - mean) * rstd;
                const float g = gr[c] * (gamma ? gamma[c] : 1.f);
                a += g;
                b += g * xh;
                dg[k] += gr[c] * xh;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        a /= (float)C;
        b /= (float)C;
        if (dx) {
#pragma unroll
            for (int k = 0; k < LN_CPL; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const float xh = (xr[c] - mean) * rstd;
                    const float g = gr[c] * (gamma ? gamma[c] : 1.f);
                    dx[row * C + c] = rstd * (g - a - xh * b);
                }
            }
        }
    }
    if (!dgamma_slabs) return;
#pragma unroll
    for (int k = 0; k < LN_CPL; ++k) {
        const int c = lane + 64 * k;
        if (c < C) part[wave][c] = dg[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        dgamma_slabs[(long)blockIdx.x * C + c] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
}

// ---- residual add + LayerNorm in one pass, and LayerNorm backward with the residual gradient folded in (round 5) ----------------
// A Swin block is  t1 = t + proj(attn(LN1(t)));  t2 = t1 + fc2(gelu(fc1(LN2(t1))))  (models/layers.py:260-300): every LayerNorm but the
// first of an RSTB reads a sum that a separate add kernel has just written, and the gradient of that sum is the LayerNorm's input
// gradient PLUS what flows past it along the residual path.  `add_ln_kernel`: s = a (+ b) -> s_out (if wanted), y = LN(s).
// `ln_bwd_add_kernel`: dx = (add1) (+ add2) + LN-backward(dy).  63 add launches of a g_a1 iteration become 4.
// Layout: 16 lanes per row, VPL float4 per lane (C / 4 <= 16 * VPL), four rows per wave: every load instruction of a 16-lane group is
// 256 contiguous bytes; the two row statistics are four-step butterflies inside the group.  Same arithmetic as the scalar kernels
// (two-pass mean / variance).
typedef float ln4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ln_sum4(ln4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float ln_red16(float v) {
    v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
    return v;
}

template <int VPL>
__global__ __launch_bounds__(256) void add_ln_kernel(const ln4* __restrict__ a, const ln4* __restrict__ b, const ln4* __restrict__ w,
                                                     const ln4* __restrict__ bias, long rows, int C4, float eps, ln4* __restrict__ s_out,
                                                     ln4* __restrict__ y_out) {
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const float invC = 1.f / (float)(4 * C4);
    ln4 wv[VPL], bv[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int c = l16 + 16 * k;
        wv[k] = (w && c < C4) ? w[c] : ln4{1.f, 1.f, 1.f, 1.f};
        bv[k] = (bias && c < C4) ? bias[c] : ln4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 16 + grp; row < rows; row += (long)gridDim.x * 16) {
        const long base = row * C4;
        ln4 v[VPL];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = l16 + 16 * k;
            if (c < C4) {
                v[k] = a[base + c];
                if (b) v[k] += b[base + c];
                if (s_out) s_out[base + c] = v[k];
                s1 += ln_sum4(v[k]);
            } else {
                v[k] = ln4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mean = ln_red16(s1) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            if (l16 + 16 * k < C4) {
                v[k] -= mean;
                s2 += ln_sum4(v[k] * v[k]);
            }
        }
        const float rstd = rsqrtf(ln_red16(s2) * invC + eps);
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = l16 + 16 * k;
            if (c < C4) y_out[base + c] = v[k] * rstd * wv[k] + bv[k];
        }
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_add_kernel(const ln4* __restrict__ x, const ln4* __restrict__ gamma, const ln4* __restrict__ dy,
                                                         const ln4* __restrict__ add1, const ln4* __restrict__ add2, long rows, int C4,
                                                         float eps, ln4* __restrict__ dx, float* __restrict__ dgamma_slabs) {
    __shared__ float part[4][512];
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const float invC = 1.f / (float)(4 * C4);
    ln4 gv[VPL], dg[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int c = l16 + 16 * k;
        gv[k] = (gamma && c < C4) ? gamma[c] : ln4{1.f, 1.f, 1.f, 1.f};
        dg[k] = ln4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 16 + grp; row < rows; row += (long)gridDim.x * 16) {
        const long base = row * C4;
        ln4 xv[VPL], g[VPL];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = l16 + 16 * k;
            if (c < C4) {
                xv[k] = x[base + c];
                g[k] = dy[base + c];
                s1 += ln_sum4(xv[k]);
            } else {
                xv[k] = ln4{0.f, 0.f, 0.f, 0.f};
                g[k] = ln4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mean = ln_red16(s1) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            if (l16 + 16 * k < C4) {
                xv[k] -= mean;
                s2 += ln_sum4(xv[k] * xv[k]);
            }
        }
        const float rstd = rsqrtf(ln_red16(s2) * invC + eps);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            xv[k] *= rstd;                               // xhat (zero beyond C)
            dg[k] += g[k] * xv[k];
            g[k] *= gv[k];                               // dy * gamma
            sa += ln_sum4(g[k]);
            sb += ln_sum4(g[k] * xv[k]);
        }
        sa = ln_red16(sa) * invC;
        sb = ln_red16(sb) * invC;
        if (dx) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const int c = l16 + 16 * k;
                if (c < C4) {
                    ln4 r = rstd * (g[k] - sa - xv[k] * sb);
                    if (add1) r = add1[base + c] + r;
                    if (add2) r = add2[base + c] + r;
                    dx[base + c] = r;
                }
            }
        }
    }
    if (!dgamma_slabs) return;
    // the four row groups of a wave hold the same channels: fold them (lanes l, l+16, l+32, l+48), then the four waves through LDS
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = dg[k][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int c = l16 + 16 * k;
            if ((threadIdx.x & 63) < 16 && c < C4) part[wave][4 * c + e] = v;
        }
    }
    __syncthreads();
    const int C = 4 * C4;
    for (int c = threadIdx.x; c < C; c += 256)
        dgamma_slabs[(long)blockIdx.x * C + c] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
}

// out = (a + b) + c: the sum of an RSTB's last block and the RSTB's own residual in the reference's order (layers.py:300, 433)
__global__ __launch_bounds__(256) void add3_kernel(const ln4* a, const ln4* b, const ln4* c, long n4, ln4* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x) out[t] = (a[t] + b[t]) + c[t];
}

// exact (erf) GELU, nn.GELU() default (layers.py:37,40)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* x, long n, float* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const float v = x[t];
        out[t] = rdo::gelu(v);
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dy, const float* x, long n, float* dx) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        dx[t] = dy[t] * rdo::gelu_grad(x[t]);
    }
}
__global__ __launch_bounds__(256) void round_kernel(const float* x, long n, float* out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) out[t] = rintf(x[t]);
}

unsigned grid1d(long n) {
    long g = rdo::ceil_div(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int make_geom(const rdo_attn_desc* d, AttnGeom* g, const char* who) {
    RDO_REQUIRE(d != nullptr, "%s: null descriptor", who);
    RDO_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->heads > 0 && d->window > 0, "%s: bad geometry", who);
    RDO_REQUIRE(d->C % d->heads == 0, "%s: C=%d not divisible by heads=%d", who, d->C, d->heads);
    RDO_REQUIRE(d->H % d->window == 0 && d->W % d->window == 0, "%s: %dx%d not divisible by window %d", who, d->H, d->W, d->window);
    RDO_REQUIRE(d->window * d->window <= NMAX, "%s: windows of more than %d tokens are not supported", who, NMAX);
    RDO_REQUIRE(d->shift >= 0 && d->shift < d->window, "%s: shift must be in [0, window)", who);
    g->B = d->B; g->H = d->H; g->W = d->W; g->C = d->C; g->heads = d->heads; g->ws = d->window; g->shift = d->shift;
    g->N = d->window * d->window;
    g->hd = d->C / d->heads;
    g->hs = (g->hd + 1) | 1;                 // odd row stride; an odd head dim gets the zero column its last k pair reads (hd + 2)
    RDO_REQUIRE(g->hd <= 64, "%s: head dims above 64 are not supported", who);
    g->scale = d->scale;
#ifdef RDO_DIAG
    { const char* e = getenv("RDO_ATTN_DIAG"); g->diag = e ? atoi(e) : 0; }
#endif
    return RDO_OK;
}

}  // namespace

// persistent grid of an attention kernel: as many workgroups as are RESIDENT at once -- registers count as well as LDS (the forward
// kernels fit four times into a CU's LDS at hd = 48 but three times into its register file: a grid of 1024 ran as 768 + a second round
// of 256 on a third of the chip) -- a multiple of 8 so that a workgroup's items stay on one XCD
template <typename K>
static long attn_grid(K kern, size_t lds, int nitems, int slot, bool bwd) {
    struct Memo { size_t lds; int dev; long resident; };
    static thread_local Memo memo[2][5] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    Memo& m = memo[bwd ? 1 : 0][slot];
    if (m.resident == 0 || m.lds != lds || m.dev != dev) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
#ifdef RDO_DIAG
        if (const char* e = getenv("RDO_ATTN_MAXOCC")) per_cu = atoi(e) < per_cu ? atoi(e) : per_cu;
#endif
        m = Memo{lds, dev, (long)per_cu * cus / 8 * 8};
    }
    return nitems < m.resident ? nitems : m.resident;
}

extern "C" {

int rdo_window_attention_fwd(const rdo_attn_desc* d, const float* qkv, const float* bias, float* out, float* probs, void* stream) {
    AttnGeom g;
    if (int rc = make_geom(d, &g, "rdo_window_attention_fwd")) return rc;
    RDO_REQUIRE(qkv && bias && (out || probs), "rdo_window_attention_fwd: null argument");
    const int windows = g.B * (g.H / g.ws) * (g.W / g.ws);
    // V, then Q and K with the score matrix over them (or behind V where two narrow tiles are smaller than it)
    const size_t qk = (size_t)2 * 64 * g.hs, sm = (size_t)64 * SS;
    const size_t lds = ((size_t)64 * g.hs + (qk > sm ? qk : sm)) * sizeof(float);
    const int no_pv = out == nullptr;
    const int nitems = windows * g.heads;
    const bool pf = ((g.hd | g.C) & 3) == 0 && ((reinterpret_cast<uintptr_t>(qkv)) & 15) == 0;
    return rdo::dispatch(
        [=](hipStream_t s) {
            auto go = [&](auto kern, int slot) -> int {
                static rdo::PerDevice attr[5];
                if (!attr[slot].done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(win_attn_fwd_mfma) failed");
                    attr[slot].mark();
                }
                const long grid = attn_grid(kern, lds, nitems, slot, false);
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, qkv, bias, g, out, probs, no_pv, windows, nitems);
                return rdo::check_launch("window_attention_fwd");
            };
            if (!pf) return go(win_attn_fwd_mfma_kernel<false, 1>, 0);
            const int it = (int)rdo::ceil_div((long)g.N * (g.hd >> 2), 256);
            if (it <= 1) return go(win_attn_fwd_mfma_kernel<true, 1>, 1);
            if (it == 2) return go(win_attn_fwd_mfma_kernel<true, 2>, 2);
            if (it == 3) return go(win_attn_fwd_mfma_kernel<true, 3>, 3);
            return go(win_attn_fwd_mfma_kernel<true, 4>, 4);
        },
        stream, "window_attention_fwd", 4.0 * windows * g.heads * (double)g.N * g.N * g.hd,
        4.0 * ((double)g.B * g.H * g.W * g.C * 4));
}

int rdo_window_attention_pv(const rdo_attn_desc* d, const float* qkv, const float* probs, float* out, void* stream) {
    AttnGeom g;
    if (int rc = make_geom(d, &g, "rdo_window_attention_pv")) return rc;
    RDO_REQUIRE(qkv && probs && out, "rdo_window_attention_pv: null argument");
    const int windows = g.B * (g.H / g.ws) * (g.W / g.ws);
    const size_t lds = (size_t)g.N * g.hs * sizeof(float);
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(win_attn_pv_kernel, dim3(windows, g.heads), dim3(64), lds, s, qkv, probs, g, out);
            return rdo::check_launch("window_attention_pv");
        },
        stream, "window_attention_pv");
}

int rdo_window_attention_bwd(const rdo_attn_desc* d, const float* qkv, const float* bias, const float* dout, float* dqkv, void* stream) {
    AttnGeom g;
    if (int rc = make_geom(d, &g, "rdo_window_attention_bwd")) return rc;
    RDO_REQUIRE(qkv && bias && dout && dqkv, "rdo_window_attention_bwd: null argument");
    const int windows = g.B * (g.H / g.ws) * (g.W / g.ws);
    // Q, K, dO, then V with P over it and dS behind
    const size_t lds = ((size_t)3 * 64 * g.hs + (size_t)2 * 64 * SS) * sizeof(float);
    static_assert(SS >= 64 + 1, "P must cover a 64-float row");
    const int nitems = windows * g.heads;
    const bool pf = ((g.hd | g.C) & 3) == 0 && ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0;
    return rdo::dispatch(
        [=](hipStream_t s) {
            auto go = [&](auto kern, int slot) -> int {
                static rdo::PerDevice attr[5];
                if (!attr[slot].done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess)
                        return rdo::set_error(RDO_EHIP, "hipFuncSetAttribute(win_attn_bwd_mfma) failed");
                    attr[slot].mark();
                }
                const long grid = attn_grid(kern, lds, nitems, slot, true);
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, qkv, bias, dout, g, dqkv, windows, nitems);
                return rdo::check_launch("window_attention_bwd");
            };
            if (!pf) return go(win_attn_bwd_mfma_kernel<false, 1>, 0);
            const int it = (int)rdo::ceil_div((long)g.N * (g.hd >> 2), 256);
            if (it <= 1) return go(win_attn_bwd_mfma_kernel<true, 1>, 1);
            if (it == 2) return go(win_attn_bwd_mfma_kernel<true, 2>, 2);
            if (it == 3) return go(win_attn_bwd_mfma_kernel<true, 3>, 3);
            return go(win_attn_bwd_mfma_kernel<true, 4>, 4);
        },
        stream, "window_attention_bwd", 10.0 * windows * g.heads * (double)g.N * g.N * g.hd,
        4.0 * ((double)g.B * g.H * g.W * g.C * 8));
}

int rdo_layer_norm_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int32_t C, float eps, float* dx,
                       float* dgamma_slabs, int32_t nslabs, void* stream) {
    RDO_REQUIRE(x && dy && rows > 0 && C > 0 && C <= 512, "rdo_layer_norm_bwd: bad argument (C <= 512)");
    RDO_REQUIRE(dx || dgamma_slabs, "rdo_layer_norm_bwd: nothing to compute");
    RDO_REQUIRE(!dgamma_slabs || nslabs > 0, "rdo_layer_norm_bwd: nslabs must be positive");
    long blocks = rdo::ceil_div(rows, 4);
    if (dgamma_slabs) blocks = nslabs;       // every slab is written (rows are strided over the blocks)
    else if (blocks > 2048) blocks = 2048;
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, dy, (long)rows, C, eps, dx,
                               dgamma_slabs);
            return rdo::check_launch("layer_norm_bwd");
        },
        stream, "layer_norm_bwd", 0.0, 12.0 * rows * C);
}

int rdo_add_layer_norm(const float* a, const float* b, const float* weight, const float* bias, int64_t rows, int32_t C, float eps,
                       float* sum_out, float* out, void* stream) {
    RDO_REQUIRE(a && out && rows > 0 && C > 0 && C % 4 == 0 && C <= 512, "rdo_add_layer_norm: bad argument (C a multiple of 4, <= 512)");
    RDO_REQUIRE(!sum_out || b, "rdo_add_layer_norm: sum_out without a second addend");
    RDO_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(weight) |
                  reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(sum_out) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                "rdo_add_layer_norm: pointers must be 16-byte aligned");
    const int C4 = C / 4;
    long blocks = rdo::ceil_div(rows, 16);
    if (blocks > 4096) blocks = 4096;
    return rdo::dispatch(
        [=](hipStream_t s) {
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, s, (const ln4*)a, (const ln4*)b, (const ln4*)weight,
                                   (const ln4*)bias, (long)rows, C4, eps, (ln4*)sum_out, (ln4*)out);
            };
            if (C4 <= 48) go(add_ln_kernel<3>);
            else if (C4 <= 80) go(add_ln_kernel<5>);
            else go(add_ln_kernel<8>);
            return rdo::check_launch("add_layer_norm");
        },
        stream, "add_layer_norm", 0.0, 4.0 * rows * C * (2.0 + (b ? 1.0 : 0.0) + (sum_out ? 1.0 : 0.0)));
}

int rdo_layer_norm_bwd_add(const float* x, const float* gamma, const float* dy, const float* add1, const float* add2, int64_t rows,
                           int32_t C, float eps, float* dx, float* dgamma_slabs, int32_t nslabs, void* stream) {
    RDO_REQUIRE(x && dy && rows > 0 && C > 0 && C % 4 == 0 && C <= 512, "rdo_layer_norm_bwd_add: bad argument (C a multiple of 4, <= 512)");
    RDO_REQUIRE(dx || dgamma_slabs, "rdo_layer_norm_bwd_add: nothing to compute");
    RDO_REQUIRE(!dgamma_slabs || nslabs > 0, "rdo_layer_norm_bwd_add: nslabs must be positive");
    RDO_REQUIRE(!add2 || add1, "rdo_layer_norm_bwd_add: add2 without add1");
    RDO_REQUIRE(dx || !add1, "rdo_layer_norm_bwd_add: an addend without dx");
    RDO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dy) |
                  reinterpret_cast<uintptr_t>(add1) | reinterpret_cast<uintptr_t>(add2) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
                "rdo_layer_norm_bwd_add: pointers must be 16-byte aligned");
    const int C4 = C / 4;
    long blocks = rdo::ceil_div(rows, 16);
    if (dgamma_slabs) blocks = nslabs;       // every slab is written (rows are strided over the blocks)
    else if (blocks > 4096) blocks = 4096;
    return rdo::dispatch(
        [=](hipStream_t s) {
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, s, (const ln4*)x, (const ln4*)gamma, (const ln4*)dy,
                                   (const ln4*)add1, (const ln4*)add2, (long)rows, C4, eps, (ln4*)dx, dgamma_slabs);
            };
            if (C4 <= 48) go(ln_bwd_add_kernel<3>);
            else if (C4 <= 80) go(ln_bwd_add_kernel<5>);
            else go(ln_bwd_add_kernel<8>);
            return rdo::check_launch("layer_norm_bwd_add");
        },
        stream, "layer_norm_bwd_add", 0.0, 4.0 * rows * C * (3.0 + (add1 ? 1.0 : 0.0) + (add2 ? 1.0 : 0.0)));
}

int rdo_add3(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream) {
    RDO_REQUIRE(a && b && c && out && n > 0 && n % 4 == 0, "rdo_add3: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(add3_kernel, dim3(grid1d(n / 4)), dim3(256), 0, s, (const ln4*)a, (const ln4*)b, (const ln4*)c, (long)(n / 4),
                               (ln4*)out);
            return rdo::check_launch("add3");
        },
        stream, "add3", 0.0, 16.0 * n);
}

int rdo_gelu_fwd(const float* x, int64_t n, float* out, void* stream) {
    RDO_REQUIRE(x && out && n > 0, "rdo_gelu_fwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, s, x, (long)n, out);
            return rdo::check_launch("gelu_fwd");
        },
        stream, "gelu_fwd", 0.0, 8.0 * n);
}

int rdo_gelu_bwd(const float* dy, const float* x, int64_t n, float* dx, void* stream) {
    RDO_REQUIRE(dy && x && dx && n > 0, "rdo_gelu_bwd: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, s, dy, x, (long)n, dx);
            return rdo::check_launch("gelu_bwd");
        },
        stream, "gelu_bwd", 0.0, 12.0 * n);
}

int rdo_round(const float* x, int64_t n, float* out, void* stream) {
    RDO_REQUIRE(x && out && n > 0, "rdo_round: bad argument");
    return rdo::dispatch(
        [=](hipStream_t s) {
            hipLaunchKernelGGL(round_kernel, dim3(grid1d(n)), dim3(256), 0, s, x, (long)n, out);
            return rdo::check_launch("round");
        },
        stream, "round", 0.0, 8.0 * n);
}

}  // extern "C"
