// Dense layers of the denoiser / discriminator on the fp32 MFMA (v_mfma_f32_32x32x2_f32),
// with the element-wise work of the neighbouring layers fused into the operand load and the
// BatchNorm column statistics fused into the epilogue.  gfx950 only.
//
// Reference ops restated (relative to the reference tree):
//   Generator_x0.forward           model/Diffusion.py:179-187  (Linear/BatchNorm1d/ReLU/Dropout)
//   Discriminator_noise_x.forward  model/Discriminator_model.py:105-113
//
// Tiling: block = 4 waves (2x2), block tile 128 rows x 128 out-cols, K staged in chunks of 32
// through double-buffered LDS (rows padded to 36 floats: conflict-free ds_read_b128).  A wave
// owns a 64x64 sub-tile = 2x2 MFMA tiles (64 accumulator registers).  The K order inside a
// chunk is permuted (lane half h consumes k = 8u+4h..8u+4h+3) so that every operand fragment
// is one 16-byte LDS read; A and B use the same permutation, so the dot products are unchanged
// up to fp32 summation order.
#include "common.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 32
#define LDS_STRIDE 36

struct LinArgs {
    const float *A, *A2;
    int in1;
    const float *W, *bias;
    const _Float16 *Ws;        // W pre-split in two f16 halves, [out][in / 4][hi x 4 | lo x 4] (k_split_weights), or null
    const float *emb;
    const int64_t *t;
    int t_uniform;
    int emb_rows;              // rows of emb: a timestep outside [0, emb_rows) is clamped (torch would raise)
    const float *scale, *shift;
    const uint8_t *drop;
    float drop_p;
    uint64_t seed, stream_id;
    float *Y;
    double *stats;
    int64_t rows;
    int in, out;
};

__device__ __forceinline__ float lin_fetch_a(const LinArgs &p, int64_t row, int k) {
    // raw element of the (possibly concatenated) A operand
    if (k < p.in1) return p.A[row * p.in1 + k];
    return p.A2[row * (int64_t)(p.in - p.in1) + (k - p.in1)];
}

template <int PRO>
__device__ __forceinline__ float lin_transform(const LinArgs &p, float a, int64_t row, int k, int trow,
                                               float keep_scale) {
    if (PRO == DDG_PRO_ADD_EMB) {
        return a + p.emb[(int64_t)trow * p.in + k];
    } else if (PRO == DDG_PRO_EMB_RELU) {
        return fmaxf(a + p.emb[(int64_t)trow * p.in + k], 0.f);
    } else if (PRO == DDG_PRO_BN_RELU) {
        float v = fmaxf(a * p.scale[k] + p.shift[k], 0.f);
        if (p.drop) {
            v = p.drop[row * p.in + k] ? v * keep_scale : 0.f;
        } else if (p.drop_p > 0.f) {
            const uint64_t e = (uint64_t)row * (uint64_t)p.in + (uint64_t)k;
            const float u = philox_uniform4(p.seed, p.stream_id, e >> 2)[e & 3];
            v = (u > p.drop_p) ? v * keep_scale : 0.f;
        }
        return v;
    }
    return a;
}

// one float4 of the transformed A operand: rows `row`, columns k0..k0+3 (zeros outside the matrix)
template <int PRO>
__device__ __forceinline__ f32x4 lin_stage_a(const LinArgs &p, int64_t row, int k0, int trow, float keep_scale, bool fast) {
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (row >= p.rows) return av;
    if (fast) {
        // no concatenation and in % 4 == 0 -> one 16-byte load, the dropout mask one 4-byte load, one Philox call
        if (k0 >= p.in) return av;
        av = *reinterpret_cast<const f32x4 *>(p.A + row * p.in + k0);
        if (PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) {
            const f32x4 ev = *reinterpret_cast<const f32x4 *>(p.emb + (int64_t)trow * p.in + k0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] += ev[e];
                if (PRO == DDG_PRO_EMB_RELU) av[e] = fmaxf(av[e], 0.f);
            }
        } else if (PRO == DDG_PRO_BN_RELU) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + k0);
            const f32x4 sh = *reinterpret_cast<const f32x4 *>(p.shift + k0);
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = fmaxf(av[e] * sc[e] + sh[e], 0.f);
            if (p.drop) {
                const uint32_t mk = *reinterpret_cast<const uint32_t *>(p.drop + row * p.in + k0);
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = ((mk >> (8 * e)) & 0xffu) ? av[e] * keep_scale : 0.f;
            } else if (p.drop_p > 0.f) {
                const uint64_t e0 = (uint64_t)row * (uint64_t)p.in + (uint64_t)k0;
                const f32x4 u = philox_uniform4(p.seed, p.stream_id, e0 >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = (u[e] > p.drop_p) ? av[e] * keep_scale : 0.f;
            }
        }
        return av;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        if (k < p.in) av[e] = lin_transform<PRO>(p, lin_fetch_a(p, row, k), row, k, trow, keep_scale);
    }
    return av;
}

// one float4 of W: output column n, inputs k0..k0+3
__device__ __forceinline__ f32x4 lin_stage_b(const LinArgs &p, int n, int k0, bool fast) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (n >= p.out) return bv;
    if (fast) {
        if (k0 < p.in) bv = *reinterpret_cast<const f32x4 *>(p.W + (int64_t)n * p.in + k0);
        return bv;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        if (k < p.in) bv[e] = p.W[(int64_t)n * p.in + k];
    }
    return bv;
}

// k_linear_wide splits lin_stage_a's fast path in two, so that the global loads of a chunk are ISSUED a whole chunk before
// anything depends on them: lin_raw_a = the loads (operand, embedding row / BatchNorm scale and shift of the same four
// columns), lin_finish_a = the prologue arithmetic on the loaded values (same expressions as lin_stage_a: bit-identical).
struct LinRaw {
    f32x4 a, e0, e1;      // operand; embedding row (ADD_EMB / EMB_RELU) or BatchNorm scale (BN_RELU); BatchNorm shift
    uint32_t mk;          // explicit dropout mask bytes (BN_RELU with p.drop)
};
template <int PRO>
__device__ __forceinline__ void lin_raw_a(const LinArgs &p, int64_t row, int k0, int trow, LinRaw &r) {
    r.a = f32x4{0.f, 0.f, 0.f, 0.f};
    r.e0 = r.a; r.e1 = r.a; r.mk = 0;
    if (row >= p.rows || k0 >= p.in) return;
    r.a = *reinterpret_cast<const f32x4 *>(p.A + row * p.in + k0);
    if (PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) {
        r.e0 = *reinterpret_cast<const f32x4 *>(p.emb + (int64_t)trow * p.in + k0);
    } else if (PRO == DDG_PRO_BN_RELU) {
        r.e0 = *reinterpret_cast<const f32x4 *>(p.scale + k0);
        r.e1 = *reinterpret_cast<const f32x4 *>(p.shift + k0);
        if (p.drop) r.mk = *reinterpret_cast<const uint32_t *>(p.drop + row * p.in + k0);
    }
}
template <int PRO>
__device__ __forceinline__ f32x4 lin_finish_a(const LinArgs &p, const LinRaw &r, int64_t row, int k0, float keep_scale) {
    f32x4 av = r.a;
    if (row >= p.rows || k0 >= p.in) return f32x4{0.f, 0.f, 0.f, 0.f};
    if (PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            av[e] += r.e0[e];
            if (PRO == DDG_PRO_EMB_RELU) av[e] = fmaxf(av[e], 0.f);
        }
    } else if (PRO == DDG_PRO_BN_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = fmaxf(av[e] * r.e0[e] + r.e1[e], 0.f);
        if (p.drop) {
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = ((r.mk >> (8 * e)) & 0xffu) ? av[e] * keep_scale : 0.f;
        } else if (p.drop_p > 0.f) {
            const uint64_t e0 = (uint64_t)row * (uint64_t)p.in + (uint64_t)k0;
            const f32x4 u = philox_uniform4(p.seed, p.stream_id, e0 >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = (u[e] > p.drop_p) ? av[e] * keep_scale : 0.f;
        }
    }
    return av;
}

// epilogue shared by both kernels: + bias, store, fp64 column statistics
#define LIN_EPILOGUE(VAL)                                                                                       \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                          \
        const int col = col0 + wc * 64 + ct * 32 + r32;                                                         \
        const bool cok = col < p.out;                                                                           \
        const float bj = cok ? p.bias[col] : 0.f;                                                               \
        _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                      \
            _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                    \
                const int64_t row = row0 + wr * 64 + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;                  \
                if (cok && row < p.rows) {                                                                      \
                    const float v = (VAL) + bj;                                                                 \
                    p.Y[row * p.out + col] = v;                                                                 \
                    s1[ct] += (double)v;                                                                        \
                    s2[ct] += (double)v * (double)v;                                                            \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }

#define LIN_STATS()                                                                                             \
    if (p.stats) {                                                                                              \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                      \
            double a = s1[ct], b = s2[ct];                                                                      \
            a += __shfl_xor(a, 32);                                                                             \
            b += __shfl_xor(b, 32);                                                                             \
            if (h == 0) { red[wid][ct][0][r32] = a; red[wid][ct][1][r32] = b; }                                 \
        }                                                                                                       \
        __syncthreads();                                                                                        \
        if (wr == 0 && h == 0) {                                                                                \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                  \
                const int col = col0 + wc * 64 + ct * 32 + r32;                                                 \
                if (col < p.out) {                                                                              \
                    const double a = red[wid][ct][0][r32] + red[wid + 2][ct][0][r32];                           \
                    const double b = red[wid][ct][1][r32] + red[wid + 2][ct][1][r32];                           \
                    atomicAdd(&p.stats[col], a);                                                                \
                    atomicAdd(&p.stats[p.out + col], b);                                                        \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }

template <int PRO>
__global__ __launch_bounds__(256, 2) void k_linear(LinArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_STRIDE];
    __shared__ double red[4][2][2][32];   // [wave][ct][sum|sumsq][col]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.y * BN;
    const int nchunks = (p.in + BK - 1) / BK;
    const int64_t ntiles = ceil_div64(p.rows, BM);
    const float keep_scale = 1.f / (1.f - p.drop_p);
    const bool fast = (p.A2 == nullptr) && ((p.in & 3) == 0);

    double s1[2] = {0, 0}, s2[2] = {0, 0};

    // staging assignment: 128 rows x 8 float4 per operand chunk -> 4 float4 per thread
    const int st_c4 = tid & 7;          // which float4 of the 32-wide chunk
    const int st_r = tid >> 3;          // 0..31, rows st_r + 32*i

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.f;

        int trows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + st_r + 32 * i;
            trows[i] = 0;
            if ((PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) && row < p.rows)
                trows[i] = min(max(p.t ? (int)p.t[row] : p.t_uniform, 0), p.emb_rows - 1);
        }

        auto stage = [&](int buf, int chunk) {
            const int k0 = chunk * BK + st_c4 * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = st_r + 32 * i;
                *reinterpret_cast<f32x4 *>(&As[buf][rl * LDS_STRIDE + st_c4 * 4]) =
                    lin_stage_a<PRO>(p, row0 + rl, k0, trows[i], keep_scale, fast);
                *reinterpret_cast<f32x4 *>(&Bs[buf][rl * LDS_STRIDE + st_c4 * 4]) = lin_stage_b(p, col0 + rl, k0, fast);
            }
        };

        stage(0, 0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int cur = c & 1;
            if (c + 1 < nchunks) stage(cur ^ 1, c + 1);
            const float *Ab = &As[cur][(wr * 64 + r32) * LDS_STRIDE + 4 * h];
            const float *Bb = &Bs[cur][(wc * 64 + r32) * LDS_STRIDE + 4 * h];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ab + 8 * u);
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(Ab + 32 * LDS_STRIDE + 8 * u);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + 8 * u);
                const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + 32 * LDS_STRIDE + 8 * u);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        LIN_EPILOGUE(acc[rt][ct][g])
    }
    LIN_STATS()
}

// ---------------------------------------------------------------------------------------------
// The same layer on the f16 matrix pipe with both operands split in two halves (large row counts: the sampler).
//   x = hi + 2^-11 lo,  hi = f16(x),  lo = f16((x - hi) * 2^11)        |x - hi - 2^-11 lo| <= 2^-22 |x|
//   a*w ~= hi_a hi_w + 2^-11 (hi_a lo_w + lo_a hi_w)                   (lo*lo dropped: <= 2^-22 |a||w|)
// Three v_mfma_f32_32x32x16_f16 per 16-deep step instead of eight v_mfma_f32_32x32x2_f32: f16 products are exact in
// the fp32 accumulator, so the result differs from the fp32 kernel's by <= ~3 * 2^-22 * sum|a||w| (fp32-class, stated
// in DESIGN.md); the lo halves are scaled by 2^11 so that they stay normal f16 numbers and go to their own
// accumulator, folded in once in the epilogue.  LDS rows: 32 hi halves | 32 lo halves (128 B) + 16 B pad.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define SPLIT_STRIDE 144            // bytes per LDS row
#define SPLIT_LO_SCALE 2048.f

__device__ __forceinline__ void split4(const f32x4 v, f16x4 &hi, f16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)((v[e] - (float)hh) * SPLIT_LO_SCALE);
    }
}

// RT = 32-row tiles per wave along M: block tile (64 RT) x 128.  RT = 1 halves the accumulators (64 registers for the
// hi and the cross sums together), which lets four workgroups share a CU: the layer is bound by the latency of the
// operand loads (K is only 176 / 352 deep, a chunk's MFMAs last ~0.35 us), so more waves in flight is what pays.
// One LDS buffer; the next chunk's global loads are in flight in registers while the current chunk is multiplied.
// Measured and NOT kept (sampler pass at 1M x 176, ms): four workgroups per CU 49.5 (spills), 64-deep chunks 65.9
// (spills), the A operand fetched two chunks ahead 39.6 (no change), output tiles transposed through LDS for 16-byte
// stores 42.9; kept: 39.6.
template <int PRO, int RT, int KCH, bool FAST>
__global__ __launch_bounds__(256, (RT == 1 ? 3 : 2)) void k_linear_split(LinArgs p) {
    constexpr int BMS = 64 * RT;
    constexpr int SSTR = 4 * KCH + 16;          // bytes per LDS row: KCH hi halves | KCH lo halves | pad
    constexpr int C4 = KCH / 4;                 // float4 per row and chunk
    constexpr int RPP = 256 / C4;               // rows staged per pass of the workgroup
    constexpr int NPA = BMS / RPP, NPB = BN / RPP;
    // one LDS block: A tile | B tile; the fp64 column partials of the last phase reuse the A tile
    __shared__ __attribute__((aligned(16))) char lds_[(BMS + BN) * SSTR];
    char *As = lds_, *Bs = lds_ + BMS * SSTR;
    static_assert(BMS * SSTR >= 4 * 2 * 2 * 32 * 8, "A tile holds the statistics partials");
    double(*red)[2][2][32] = reinterpret_cast<double(*)[2][2][32]>(lds_);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.y * BN;
    const int nchunks = (p.in + KCH - 1) / KCH;
    const int64_t ntiles = ceil_div64(p.rows, BMS);
    const float keep_scale = 1.f / (1.f - p.drop_p);
    constexpr bool fast = FAST;          // FAST: no concatenated operand and in % 4 == 0 (16-byte operand loads); a compile-time
                                         // flag: with both paths alive the kernel spilled 26-49 registers (sampler 41.9 -> 39.6 ms)

    double s1[2] = {0, 0}, s2[2] = {0, 0};
    const int st_c4 = tid % C4, st_r = tid / C4;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * BMS;
        f32x16 accH[RT][2], accX[RT][2];
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 16; ++g) { accH[a][b][g] = 0.f; accX[a][b][g] = 0.f; }

        int trows[NPA];
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int64_t row = row0 + st_r + RPP * i;
            trows[i] = 0;
            if ((PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) && row < p.rows)
                trows[i] = min(max(p.t ? (int)p.t[row] : p.t_uniform, 0), p.emb_rows - 1);
        }

        f32x4 pa[NPA], pb[NPB];
        auto fetch = [&](int chunk) {
            const int k0 = chunk * KCH + st_c4 * 4;
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                pa[i] = lin_stage_a<PRO>(p, row0 + st_r + RPP * i, k0, trows[i], keep_scale, fast);
#pragma unroll
            for (int i = 0; i < NPB; ++i) pb[i] = lin_stage_b(p, col0 + st_r + RPP * i, k0, fast);
        };
        auto commit = [&]() {
            f16x4 hi, lo;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int rl = st_r + RPP * i;
                split4(pa[i], hi, lo);
                *reinterpret_cast<f16x4 *>(&As[rl * SSTR + st_c4 * 8]) = hi;
                *reinterpret_cast<f16x4 *>(&As[rl * SSTR + 2 * KCH + st_c4 * 8]) = lo;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int rl = st_r + RPP * i;
                split4(pb[i], hi, lo);
                *reinterpret_cast<f16x4 *>(&Bs[rl * SSTR + st_c4 * 8]) = hi;
                *reinterpret_cast<f16x4 *>(&Bs[rl * SSTR + 2 * KCH + st_c4 * 8]) = lo;
            }
        };

        fetch(0);
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();                       // the previous chunk's fragment reads are done
            commit();
            if (c + 1 < nchunks) fetch(c + 1);     // in flight while this chunk is multiplied
            __syncthreads();
            // fragment of k-step u (16 deep): lane half h holds k = 16u + 8h .. + 7 of its row: one 16-byte read
            const char *Ab = &As[(wr * 32 * RT + r32) * SSTR + 16 * h];
            const char *Bb = &Bs[(wc * 64 + r32) * SSTR + 16 * h];
#pragma unroll
            for (int u = 0; u < KCH / 16; ++u) {
                f16x8 ah[RT], al[RT], bh[2], bl[2];
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    ah[q] = *reinterpret_cast<const f16x8 *>(Ab + q * 32 * SSTR + 32 * u);
                    al[q] = *reinterpret_cast<const f16x8 *>(Ab + q * 32 * SSTR + 2 * KCH + 32 * u);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bh[q] = *reinterpret_cast<const f16x8 *>(Bb + q * 32 * SSTR + 32 * u);
                    bl[q] = *reinterpret_cast<const f16x8 *>(Bb + q * 32 * SSTR + 2 * KCH + 32 * u);
                }
#pragma unroll
                for (int ra = 0; ra < RT; ++ra)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        accH[ra][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ra], bh[cb], accH[ra][cb], 0, 0, 0);
                        accX[ra][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ra], bl[cb], accX[ra][cb], 0, 0, 0);
                        accX[ra][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ra], bh[cb], accX[ra][cb], 0, 0, 0);
                    }
            }
        }
        // epilogue: + bias, store, fp64 column statistics
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = col0 + wc * 64 + ct * 32 + r32;
            const bool cok = col < p.out;
            const float bj = cok ? p.bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int64_t row = row0 + wr * 32 * RT + rt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (cok && row < p.rows) {
                        const float v = accH[rt][ct][g] + accX[rt][ct][g] * (1.f / SPLIT_LO_SCALE) + bj;
                        p.Y[row * p.out + col] = v;
                        s1[ct] += (double)v;
                        s2[ct] += (double)v * (double)v;
                    }
                }
            }
        }
    }
    __syncthreads();                               // the last tile's fragment reads are done: the A tile becomes `red`
    LIN_STATS()
}

// ---------------------------------------------------------------------------------------------
// The split layer with WIDE column tiles (the sampler's full-table calls, 16-byte operand loads).  k_linear_split is
// bound by the VALU work of staging: the A operand's prologue (BatchNorm + ReLU + a Philox dropout mask per float4) and
// the split of both operands are redone for every 128-column tile of a row tile, and a 176-wide output wastes 31 % of
// its two tiles.  Here eight waves own 32 rows x 96 columns each: WR x WC = 4 x 2 (128 x 192, outputs up to 192 wide) or
// 2 x 4 (64 x 384), so that a 176- or 352-wide layer is ONE column tile and the prologue runs once per element.  Two LDS
// buffers and one barrier per 32-deep chunk: the split + commit of chunk c+1 and the loads of chunk c+2 are in the same
// stretch of code as the MFMAs of chunk c, and the two waves of a SIMD overlap one's VALU with the other's MFMAs.
// Same products in the same order per output element as k_linear_split: bit-identical results.
// ---------------------------------------------------------------------------------------------
// PS: the weights come pre-split (p.Ws): their staging is two 8-byte copies per float4 instead of a split.
template <int PRO, int WR, int WC, bool PS>
__global__ __launch_bounds__(512, 2) void k_linear_wide(LinArgs p) {
    static_assert(WR * WC == 8, "eight waves");
    constexpr int KCH = 32;
    constexpr int SSTR = 4 * KCH + 16;          // bytes per LDS row: KCH hi halves | KCH lo halves | pad
    constexpr int BMW = 32 * WR, BNW = 96 * WC;
    constexpr int NPA = BMW / 64, NPB = BNW / 64;      // float4 per thread and chunk (64 rows per pass of the workgroup)
    static_assert(NPA * 64 == BMW && NPB * 64 == BNW, "tile rows are a multiple of the staging pass");
    constexpr int BUFB = (BMW + BNW) * SSTR;    // bytes per buffer: A tile | B tile
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    static_assert(2 * BUFB >= 8 * 3 * 2 * 32 * 8, "the buffers hold the statistics partials");
    double(*red)[3][2][32] = reinterpret_cast<double(*)[3][2][32]>(wlds);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC;
    const int r32 = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.y * BNW;
    const int nchunks = (p.in + KCH - 1) / KCH;
    const int64_t ntiles = ceil_div64(p.rows, BMW);
    const float keep_scale = 1.f / (1.f - p.drop_p);

    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    const int st_c4 = tid & 7, st_r = tid >> 3;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * BMW;
        f32x16 accH[3], accX[3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) { accH[b][g] = 0.f; accX[b][g] = 0.f; }

        int trows[NPA];
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int64_t row = row0 + st_r + 64 * i;
            trows[i] = 0;
            if ((PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) && row < p.rows)
                trows[i] = min(max(p.t ? (int)p.t[row] : p.t_uniform, 0), p.emb_rows - 1);
        }

        LinRaw pa[NPA];
        f32x4 pb[PS ? 1 : NPB];
        f16x4 pbh[PS ? NPB : 1], pbl[PS ? NPB : 1];
        int k0_held = 0;                            // first column of the chunk held in pa / pb
        auto fetch = [&](int chunk) {
            const int k0 = chunk * KCH + st_c4 * 4;
            k0_held = k0;
#pragma unroll
            for (int i = 0; i < NPA; ++i) lin_raw_a<PRO>(p, row0 + st_r + 64 * i, k0, trows[i], pa[i]);
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                if constexpr (PS) {
                    const int n = col0 + st_r + 64 * i;
                    pbh[i] = f16x4{0, 0, 0, 0};
                    pbl[i] = f16x4{0, 0, 0, 0};
                    if (n < p.out && k0 < p.in) {
                        const f16x8 hl = *reinterpret_cast<const f16x8 *>(p.Ws + ((int64_t)n * p.in + k0) * 2);      // one 16-byte load
                        pbh[i] = f16x4{hl[0], hl[1], hl[2], hl[3]};
                        pbl[i] = f16x4{hl[4], hl[5], hl[6], hl[7]};
                    }
                } else {
                    pb[i] = lin_stage_b(p, col0 + st_r + 64 * i, k0, true);
                }
            }
        };
        auto commit = [&](char *buf) {
            char *As = buf, *Bs = buf + BMW * SSTR;
            f16x4 hi, lo;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int rl = st_r + 64 * i;
                split4(lin_finish_a<PRO>(p, pa[i], row0 + rl, k0_held, keep_scale), hi, lo);
                *reinterpret_cast<f16x4 *>(&As[rl * SSTR + st_c4 * 8]) = hi;
                *reinterpret_cast<f16x4 *>(&As[rl * SSTR + 2 * KCH + st_c4 * 8]) = lo;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int rl = st_r + 64 * i;
                if constexpr (PS) { hi = pbh[i]; lo = pbl[i]; } else split4(pb[i], hi, lo);
                *reinterpret_cast<f16x4 *>(&Bs[rl * SSTR + st_c4 * 8]) = hi;
                *reinterpret_cast<f16x4 *>(&Bs[rl * SSTR + 2 * KCH + st_c4 * 8]) = lo;
            }
        };

        fetch(0);
        commit(wlds);
        if (nchunks > 1) fetch(1);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const char *cur = wlds + (c & 1) * BUFB;
            const char *Ab = cur + (wr * 32 + r32) * SSTR + 16 * h;
            const char *Bb = cur + BMW * SSTR + (wc * 96 + r32) * SSTR + 16 * h;
#pragma unroll
            for (int u = 0; u < KCH / 16; ++u) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(Ab + 32 * u);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(Ab + 2 * KCH + 32 * u);
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const f16x8 bh = *reinterpret_cast<const f16x8 *>(Bb + cb * 32 * SSTR + 32 * u);
                    const f16x8 bl = *reinterpret_cast<const f16x8 *>(Bb + cb * 32 * SSTR + 2 * KCH + 32 * u);
                    accH[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accH[cb], 0, 0, 0);
                    accX[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accX[cb], 0, 0, 0);
                    accX[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accX[cb], 0, 0, 0);
                }
            }
            // chunk c+1 into the other buffer (last read by the MFMAs of chunk c-1, a barrier ago), chunk c+2 into registers
            if (c + 1 < nchunks) commit(wlds + ((c + 1) & 1) * BUFB);
            if (c + 2 < nchunks) fetch(c + 2);
            __syncthreads();
        }
        // epilogue: + bias, store, fp64 column statistics
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int col = col0 + wc * 96 + ct * 32 + r32;
            const bool cok = col < p.out;
            const float bj = cok ? p.bias[col] : 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int64_t row = row0 + wr * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                if (cok && row < p.rows) {
                    const float v = accH[ct][g] + accX[ct][g] * (1.f / SPLIT_LO_SCALE) + bj;
                    p.Y[row * p.out + col] = v;
                    s1[ct] += (double)v;
                    s2[ct] += (double)v * (double)v;
                }
            }
        }
    }
    if (p.stats) {                                 // (every wave is past the last tile's barrier: the buffers are free)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            double a = s1[ct], b = s2[ct];
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            if (h == 0) { red[wid][ct][0][r32] = a; red[wid][ct][1][r32] = b; }
        }
        __syncthreads();
        if (wr == 0 && h == 0) {
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int col = col0 + wc * 96 + ct * 32 + r32;
                if (col < p.out) {
                    double a = 0, b = 0;
#pragma unroll
                    for (int q = 0; q < WR; ++q) { a += red[q * WC + wc][ct][0][r32]; b += red[q * WC + wc][ct][1][r32]; }
                    atomicAdd(&p.stats[col], a);
                    atomicAdd(&p.stats[p.out + col], b);
                }
            }
        }
    }
}

// W [out][in] fp32 -> f16 halves, interleaved per four elements: [out][in / 4][hi x 4 | lo x 4] (same split as split4; in % 4 == 0)
__global__ void k_split_weights(const float *__restrict__ W, _Float16 *__restrict__ Ws, int out, int in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)out * in) return;
    const int n = (int)(i / in), k = (int)(i - (int64_t)n * in);
    const float v = W[i];
    const _Float16 hh = (_Float16)v;
    const int64_t o = ((int64_t)n * in + (k & ~3)) * 2 + (k & 3);
    Ws[o] = hh;
    Ws[o + 4] = (_Float16)((v - (float)hh) * SPLIT_LO_SCALE);
}

template <int PRO, int WR, int WC, bool PS>
static int launch_wide(const LinArgs &p, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (32 * WR + 96 * WC) * (4 * 32 + 16);
    static std::atomic<int> granted{0};
    if (!granted.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_linear_wide<PRO, WR, WC, PS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            fprintf(stderr, "ddgani: hipFuncSetAttribute(%zu): %s\n", lds, hipGetErrorString(e));
            return DDG_E_LAUNCH;
        }
        granted.store(1, std::memory_order_release);
    }
    const int64_t ntiles = ceil_div64(p.rows, 32 * WR);
    const int ncol = (p.out + 96 * WC - 1) / (96 * WC);
    int64_t g = ntiles;
    const int64_t cap = (256 + ncol - 1) / ncol;            // one resident workgroup per CU
    if (g > cap) g = cap;
    if (g >= 8) g -= g % 8;
    k_linear_wide<PRO, WR, WC, PS><<<dim3((unsigned)g, (unsigned)ncol), 512, lds, s>>>(p);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// rows from which the split kernel serves a layer (the sampler's full-table calls); below it -- every B = 128 training
// step, the small fixtures -- the fp32 kernel, whose arithmetic the training-trajectory pins were taken with.
// DDG_LINEAR_F32=1 in the environment (read once) keeps the fp32 kernel everywhere (A/B timing).
#define SPLIT_MIN_ROWS 8192
static bool linear_force_f32() {
    static const bool v = [] { const char *e = getenv("DDG_LINEAR_F32"); return e && e[0] == '1'; }();
    return v;
}

// ---------------------------------------------------------------------------------------------
// The plain layer Y = [A | A2] W^T + b for a few rows (the B = 128 training steps): a 128-row tile would occupy two or
// three CUs and walk the whole depth as one dependent chain of global loads (20-67 us per layer).  Here a workgroup owns
// a 32 x 32 tile of Y and its four waves a quarter of the depth each (fp32 MFMA, 16-byte operand loads when the widths
// allow); the partial tiles meet in LDS and are added in a fixed order, so the result does not depend on timing.
// fp64 column sums of y and y^2 (BatchNorm statistics) as in k_linear: one atomicAdd per column and row tile.
// ---------------------------------------------------------------------------------------------
#define SMALL_MAX_ROWS 256
template <int PRO>
__global__ __launch_bounds__(256) void k_linear_small(LinArgs p) {
    __shared__ float red[4][32 * 33];
    __shared__ double sred[2][8][32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int m = m0 + r32, n = n0 + r32;
    const bool mok = m < p.rows;
    const int K = p.in;
    // depth slice of this wave, a multiple of 8; lane half h takes positions k + 4h .. k + 4h + 3 of an 8-deep step
    // (the same permutation on both operands); the operands come through the prologue helpers of k_linear
    const int kper = ((K + 31) / 32) * 8;
    const int k0 = wid * kper, k1 = min(((K + 7) / 8) * 8, k0 + kper);
    const bool fast = (p.A2 == nullptr) && ((p.in & 3) == 0);
    const float keep_scale = 1.f / (1.f - p.drop_p);
    int trow = 0;
    if ((PRO == DDG_PRO_ADD_EMB || PRO == DDG_PRO_EMB_RELU) && mok)
        trow = min(max(p.t ? (int)p.t[m] : p.t_uniform, 0), p.emb_rows - 1);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    if (fast) {
        for (int k = k0; k < k1; k += 8) {
            const int kk = k + 4 * h;
            const f32x4 a4 = lin_stage_a<PRO>(p, mok ? m : p.rows, kk, trow, keep_scale, true);     // zeros past the last row
            const f32x4 w4 = lin_stage_b(p, n, kk, true);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], w4[j], acc, 0, 0, 0);
        }
    } else {
        // concatenated or odd-width operands: element by element, depth positions k + h in ascending order
        const bool nok = n < p.out;
        for (int k = k0; k < min(K, k1); k += 2) {
            const int kk = k + h;
            float a = 0.f, w = 0.f;
            if (kk < K) {
                if (mok) a = lin_transform<PRO>(p, lin_fetch_a(p, m, kk), m, kk, trow, keep_scale);
                if (nok) w = p.W[(int64_t)n * K + kk];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) red[wid][((g & 3) + 8 * (g >> 2) + 4 * h) * 33 + r32] = acc[g];
    __syncthreads();
    // thread -> column tid & 31, rows 4 (tid >> 5) .. + 3 of the tile
    const int c = tid & 31, rg = tid >> 5;
    const int col = n0 + c;
    const float bias = col < p.out ? p.bias[col] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = rg * 4 + i;
        const int row = m0 + rl;
        const float y = ((red[0][rl * 33 + c] + red[1][rl * 33 + c]) + (red[2][rl * 33 + c] + red[3][rl * 33 + c])) + bias;
        if (row < p.rows && col < p.out) {
            p.Y[(int64_t)row * p.out + col] = y;
            s1 += (double)y;
            s2 += (double)y * (double)y;
        }
    }
    if (p.stats) {
        sred[0][rg][c] = s1;
        sred[1][rg][c] = s2;
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5;
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += sred[which][j][c];
            if (col < p.out) atomicAdd(&p.stats[which * p.out + col], t);
        }
    }
}

static int launch_linear(const LinArgs &p, int prologue, hipStream_t s) {
    if (p.rows <= SMALL_MAX_ROWS) {
        static const bool big = [] { const char *e = getenv("DDG_LINEAR_TILE128"); return e && e[0] == '1'; }();
        if (!big) {
            dim3 g((unsigned)((p.out + 31) / 32), (unsigned)((p.rows + 31) / 32));
            switch (prologue) {
                case DDG_PRO_NONE: k_linear_small<DDG_PRO_NONE><<<g, 256, 0, s>>>(p); break;
                case DDG_PRO_ADD_EMB: k_linear_small<DDG_PRO_ADD_EMB><<<g, 256, 0, s>>>(p); break;
                case DDG_PRO_BN_RELU: k_linear_small<DDG_PRO_BN_RELU><<<g, 256, 0, s>>>(p); break;
                case DDG_PRO_EMB_RELU: k_linear_small<DDG_PRO_EMB_RELU><<<g, 256, 0, s>>>(p); break;
                default: return DDG_E_ARG;
            }
            DDG_CHECK_LAUNCH();
            return DDG_OK;
        }
    }
    const int64_t ntiles = ceil_div64(p.rows, BM);
    const int ncol = (p.out + BN - 1) / BN;
    int64_t gx = ntiles;
    const int64_t cap = (256 * 2 + ncol - 1) / ncol;   // ~2 resident blocks per CU in total
    if (gx > cap) gx = cap;
    // workgroup (x, y) has linear id x + y * gx and lands on XCD id % 8: with gx a multiple of 8 the column tiles of a
    // row tile share an XCD, so the A rows come from HBM once and from that XCD's L2 for the other column tiles
    if (gx >= 8) gx -= gx % 8;
    dim3 grid((unsigned)gx, (unsigned)ncol);
    if (p.rows >= SPLIT_MIN_ROWS && !linear_force_f32()) {
        static const int rt = [] { const char *e = getenv("DDG_SPLIT_RT"); return (e && e[0] == '2') ? 2 : 1; }();
        // chunk depth: 32 (default); DDG_SPLIT_KCH=64 selects 64-deep chunks, measured SLOWER (sampler 65.9 vs 41.9 ms
        // at 1M x 176: the doubled prefetch registers spill ~100 VGPRs at three workgroups per CU)
        static const int kch = [] { const char *e = getenv("DDG_SPLIT_KCH"); return (e && e[0] == '6') ? 64 : 32; }();
        const int64_t nt = ceil_div64(p.rows, 64 * rt);
        int64_t g = nt;
        const int64_t cap2 = (256 * (rt == 1 ? 3 : 2) + ncol - 1) / ncol;
        if (g > cap2) g = cap2;
        if (g >= 8) g -= g % 8;
        dim3 grid2((unsigned)g, (unsigned)ncol);
        const bool fast = (p.A2 == nullptr) && ((p.in & 3) == 0);
        // wide column tiles (DDG_LINEAR_WIDE=0: the 128-column kernel, A/B)
        static const bool wide = [] { const char *e = getenv("DDG_LINEAR_WIDE"); return !(e && e[0] == '0'); }();
        if (fast && wide) {
#define LAUNCH_WIDE(PRO_)                                                                                          \
            return p.Ws ? (p.out <= 192 ? launch_wide<PRO_, 4, 2, true>(p, s) : launch_wide<PRO_, 2, 4, true>(p, s))  \
                        : (p.out <= 192 ? launch_wide<PRO_, 4, 2, false>(p, s) : launch_wide<PRO_, 2, 4, false>(p, s));
            switch (prologue) {
                case DDG_PRO_NONE: LAUNCH_WIDE(DDG_PRO_NONE)
                case DDG_PRO_ADD_EMB: LAUNCH_WIDE(DDG_PRO_ADD_EMB)
                case DDG_PRO_BN_RELU: LAUNCH_WIDE(DDG_PRO_BN_RELU)
                case DDG_PRO_EMB_RELU: LAUNCH_WIDE(DDG_PRO_EMB_RELU)
                default: return DDG_E_ARG;
            }
#undef LAUNCH_WIDE
        }
#define LAUNCH_SPLIT(PRO_)                                                          \
        if (!fast) k_linear_split<PRO_, 1, 32, false><<<grid2, 256, 0, s>>>(p);     \
        else if (rt == 1 && kch == 64) k_linear_split<PRO_, 1, 64, true><<<grid2, 256, 0, s>>>(p);   \
        else if (rt == 1) k_linear_split<PRO_, 1, 32, true><<<grid2, 256, 0, s>>>(p);     \
        else k_linear_split<PRO_, 2, 32, true><<<grid2, 256, 0, s>>>(p);
        switch (prologue) {
            case DDG_PRO_NONE: LAUNCH_SPLIT(DDG_PRO_NONE) break;
            case DDG_PRO_ADD_EMB: LAUNCH_SPLIT(DDG_PRO_ADD_EMB) break;
            case DDG_PRO_BN_RELU: LAUNCH_SPLIT(DDG_PRO_BN_RELU) break;
            case DDG_PRO_EMB_RELU: LAUNCH_SPLIT(DDG_PRO_EMB_RELU) break;
            default: return DDG_E_ARG;
        }
#undef LAUNCH_SPLIT
        DDG_CHECK_LAUNCH();
        return DDG_OK;
    }
    switch (prologue) {
        case DDG_PRO_NONE: k_linear<DDG_PRO_NONE><<<grid, 256, 0, s>>>(p); break;
        case DDG_PRO_ADD_EMB: k_linear<DDG_PRO_ADD_EMB><<<grid, 256, 0, s>>>(p); break;
        case DDG_PRO_BN_RELU: k_linear<DDG_PRO_BN_RELU><<<grid, 256, 0, s>>>(p); break;
        case DDG_PRO_EMB_RELU: k_linear<DDG_PRO_EMB_RELU><<<grid, 256, 0, s>>>(p); break;
        default: return DDG_E_ARG;
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

static int linear_impl(const float *A, const float *A2, int in1, const float *W, const float *bias, int prologue,
                       const float *emb, int emb_rows, const int64_t *t, int t_uniform, const float *scale,
                       const float *shift, const uint8_t *drop, float drop_p, uint64_t seed, uint64_t stream_id, float *Y,
                       double *stats, int64_t rows, int in, int out, void *stream, _Float16 *wsplit = nullptr) {
    // wsplit: scratch of 2 * out * in f16 for the pre-split weights of the wide kernel, or null (split per tile)
    if (!W || !bias || rows < 0 || in < 1 || out < 1) return DDG_E_ARG;
    if (rows > 0 && (!A || !Y)) return DDG_E_ARG;       // (an empty batch: nothing is launched, A / Y may be NULL)
    if (A2 ? (in1 < 1 || in1 >= in) : (in1 != in)) return DDG_E_ARG;
    if ((prologue == DDG_PRO_ADD_EMB || prologue == DDG_PRO_EMB_RELU) && !emb) return DDG_E_ARG;
    if (prologue == DDG_PRO_BN_RELU && (!scale || !shift)) return DDG_E_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return DDG_E_ARG;
    if (rows == 0) return DDG_OK;
    LinArgs p;
    p.A = A; p.A2 = A2; p.in1 = in1; p.W = W; p.bias = bias; p.emb = emb; p.t = t; p.t_uniform = t_uniform;
    p.emb_rows = emb_rows < 1 ? 1 : emb_rows;
    p.scale = scale; p.shift = shift; p.drop = drop; p.drop_p = drop_p; p.seed = seed; p.stream_id = stream_id;
    p.Y = Y; p.stats = stats; p.rows = rows; p.in = in; p.out = out;
    p.Ws = nullptr;
    if (wsplit && rows >= SPLIT_MIN_ROWS && !A2 && (in & 3) == 0 && !linear_force_f32()) {
        const int64_t n = (int64_t)out * in;
        k_split_weights<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, wsplit, out, in);
        DDG_CHECK_LAUNCH();
        p.Ws = wsplit;
    }
    return launch_linear(p, prologue, (hipStream_t)stream);
}

// the generic entry point does not know how many rows emb has: the caller guarantees 0 <= t < n_steps
extern "C" int ddg_linear(const float *A, const float *A2, int in1, const float *W, const float *bias, int prologue,
                          const float *emb, const int64_t *t, int t_uniform, const float *scale, const float *shift,
                          const uint8_t *drop, float drop_p, uint64_t seed, uint64_t stream_id, float *Y,
                          double *stats, int64_t rows, int in, int out, void *stream) {
    return linear_impl(A, A2, in1, W, bias, prologue, emb, 0x7fffffff, t, t_uniform, scale, shift, drop, drop_p, seed,
                       stream_id, Y, stats, rows, in, out, stream);
}

// ---------------------------------------------------------------------------------------------
// a6 Generator_x0.forward, single-GPU composition
// workspace: z1 [rows,enc] | z2 [rows,lat] | stats double[2*max] x3 | scale/shift float[max] x 2 x 3
//            | from SPLIT_MIN_ROWS rows: W1..W4 pre-split in f16 halves
// (z3 reuses z1; the pre-head logits go straight to `out`)
// ---------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t ddg_denoiser_workspace_bytes(int64_t rows, int in_dim, int enc_dim, int lat_dim) {
    const size_t mx = (size_t)(enc_dim > lat_dim ? enc_dim : lat_dim);
    size_t b = align256((size_t)rows * enc_dim * 4) + align256((size_t)rows * lat_dim * 4) + 3 * align256(2 * mx * 8) +
               6 * align256(mx * 4);
    // full-table calls: the four weight matrices pre-split in f16 halves (4 bytes per element, as the fp32 originals)
    if (rows >= SPLIT_MIN_ROWS)
        b += 2 * align256((size_t)in_dim * enc_dim * 4) + 2 * align256((size_t)enc_dim * lat_dim * 4);
    return b;
}

extern "C" int ddg_denoiser_fwd(const ddg_gen_weights *w, const float *x_t, const int64_t *t, int t_uniform,
                                const uint8_t *drop0, const uint8_t *drop1, const uint8_t *drop2, float drop_p,
                                uint64_t seed, uint64_t stream_id, float bn_eps, const int32_t *offsets,
                                const uint8_t *is_cat, int C, int apply_head, float *out, int64_t rows,
                                void *workspace, size_t workspace_bytes, void *stream) {
    if (!w || !x_t || !out || rows < 1) return DDG_E_ARG;
    const int D = w->in_dim, E = w->enc_dim, L = w->lat_dim;
    if (D < 1 || E < 1 || L < 1) return DDG_E_ARG;
    if (workspace_bytes < ddg_denoiser_workspace_bytes(rows, D, E, L) || !workspace) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t mx = (size_t)(E > L ? E : L);
    char *p = (char *)workspace;
    float *z1 = (float *)p; p += align256((size_t)rows * E * 4);
    float *z2 = (float *)p; p += align256((size_t)rows * L * 4);
    double *st[3];
    for (int i = 0; i < 3; ++i) { st[i] = (double *)p; p += align256(2 * mx * 8); }
    float *sc[3], *sh[3];
    for (int i = 0; i < 3; ++i) { sc[i] = (float *)p; p += align256(mx * 4); sh[i] = (float *)p; p += align256(mx * 4); }
    _Float16 *ws4[4] = {nullptr, nullptr, nullptr, nullptr};
    if (rows >= SPLIT_MIN_ROWS) {
        ws4[0] = (_Float16 *)p; p += align256((size_t)D * E * 4);
        ws4[1] = (_Float16 *)p; p += align256((size_t)E * L * 4);
        ws4[2] = (_Float16 *)p; p += align256((size_t)E * L * 4);
        ws4[3] = (_Float16 *)p; p += align256((size_t)D * E * 4);
    }
    int rc = zero_async(st[0], 3 * align256(2 * mx * 8), s);      // not hipMemsetAsync: see common.h
    if (rc) return rc;
    // layer 1: (x + temb[t]) W1^T + b1
    rc = linear_impl(x_t, nullptr, D, w->W1, w->b1, DDG_PRO_ADD_EMB, w->temb, w->n_steps, t, t_uniform, nullptr, nullptr, nullptr,
                    0.f, 0, 0, z1, st[0], rows, D, E, stream, ws4[0]);
    if (rc) return rc;
    rc = ddg_bn_finalize(st[0], w->g1, w->be1, (double)rows, bn_eps, sc[0], sh[0], E, stream);
    if (rc) return rc;
    // layer 2: drop(relu(bn(z1))) W2^T + b2
    rc = linear_impl(z1, nullptr, E, w->W2, w->b2, DDG_PRO_BN_RELU, nullptr, 1, nullptr, 0, sc[0], sh[0], drop0, drop_p,
                     seed, stream_id * 4 + 0, z2, st[1], rows, E, L, stream, ws4[1]);
    if (rc) return rc;
    rc = ddg_bn_finalize(st[1], w->g2, w->be2, (double)rows, bn_eps, sc[1], sh[1], L, stream);
    if (rc) return rc;
    // layer 3 (z3 overwrites z1: z1 is dead once layer 2 has been enqueued, same stream)
    rc = linear_impl(z2, nullptr, L, w->W3, w->b3, DDG_PRO_BN_RELU, nullptr, 1, nullptr, 0, sc[1], sh[1], drop1, drop_p,
                     seed, stream_id * 4 + 1, z1, st[2], rows, L, E, stream, ws4[2]);
    if (rc) return rc;
    rc = ddg_bn_finalize(st[2], w->g3, w->be3, (double)rows, bn_eps, sc[2], sh[2], E, stream);
    if (rc) return rc;
    // layer 4 -> logits
    rc = linear_impl(z1, nullptr, E, w->W4, w->b4, DDG_PRO_BN_RELU, nullptr, 1, nullptr, 0, sc[2], sh[2], drop2, drop_p,
                     seed, stream_id * 4 + 2, out, nullptr, rows, E, D, stream, ws4[3]);
    if (rc) return rc;
    if (apply_head) {
        if (!offsets || !is_cat || C < 1) return DDG_E_ARG;
        rc = ddg_head(out, offsets, is_cat, C, out, rows, D, stream);
        if (rc) return rc;
    }
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// ddg_denoiser_fwd_static: ddg_denoiser_fwd for callers that cannot allocate or bring a workspace
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
// in the reference.  The layer activations z1 / z2, the fp64 BatchNorm sums and the scale / shift
// vectors live in a file-scope device array of DDG_STATIC_SCRATCH_BYTES; every GPU holds its own copy
// at its own address until the process ends.  The array's name is only a host-side handle:
// its device address is looked up once per device with hipGetSymbolAddress and that pointer is what
// the kernels receive.  The array belongs to the CURRENT device, so the caller's tensors and stream must live on it: the
// entry point checks that x_t and out do (hipPointerGetAttributes) and refuses otherwise -- a symbol address of one GPU
// is not valid on another.  Calls that overlap on one GPU share the array and race.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[DDG_STATIC_SCRATCH_BYTES];
#define DDG_MAX_DEVICES 64
static void *g_scratch_ptr[DDG_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int static_scratch(void **out, const void *p0, const void *p1) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DDG_MAX_DEVICES) return DDG_E_LAUNCH;
    for (const void *p : {p0, p1}) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); continue; }      // not a tracked pointer: nothing to compare
        if (at.type == hipMemoryTypeDevice && at.device != dev) {
            fprintf(stderr, "ddgani: ddg_denoiser_fwd_static: a tensor lives on device %d, the current device is %d "
                            "(the static scratch belongs to the current device)\n", at.device, dev);
            return DDG_E_ARG;
        }
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));     // address on the CURRENT device
        if (e != hipSuccess || !ptr) {
            fprintf(stderr, "ddgani: hipGetSymbolAddress(scratch) on device %d: %s\n", dev, hipGetErrorString(e));
            return DDG_E_LAUNCH;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return DDG_OK;
}

extern "C" int ddg_denoiser_fwd_static(const ddg_gen_weights *w, const float *x_t, const int64_t *t, int t_uniform,
                                       const uint8_t *drop0, const uint8_t *drop1, const uint8_t *drop2, float drop_p,
                                       uint64_t seed, uint64_t stream_id, float bn_eps, const int32_t *offsets,
                                       const uint8_t *is_cat, int C, int apply_head, float *out, int64_t rows,
                                       void *stream) {
    if (!w || !x_t || !out || rows < 1) return DDG_E_ARG;
    if (w->in_dim < 1 || w->enc_dim < 1 || w->lat_dim < 1) return DDG_E_ARG;
    const size_t need = ddg_denoiser_workspace_bytes(rows, w->in_dim, w->enc_dim, w->lat_dim);
    if (need > (size_t)DDG_STATIC_SCRATCH_BYTES) {
        fprintf(stderr, "ddgani: ddg_denoiser_fwd_static: %lld rows x (%d + %d) columns need %zu bytes of scratch, the "
                        "static array holds %zu; use ddg_denoiser_fwd with a workspace\n",
                (long long)rows, w->enc_dim, w->lat_dim, need, (size_t)DDG_STATIC_SCRATCH_BYTES);
        return DDG_E_SCRATCH_LIMIT;
    }
    void *ptr = nullptr;
    int rc = static_scratch(&ptr, x_t, out);
    if (rc) return rc;
    // the BatchNorm sums must start at zero: ddg_denoiser_fwd clears them (a kernel on `ptr`, zero_async) on every call;
    // everything else in the array is written before it is read
    return ddg_denoiser_fwd(w, x_t, t, t_uniform, drop0, drop1, drop2, drop_p, seed, stream_id, bn_eps, offsets, is_cat,
                            C, apply_head, out, rows, ptr, (size_t)DDG_STATIC_SCRATCH_BYTES, stream);
}

// ---------------------------------------------------------------------------------------------
// a10 Discriminator_noise_x.forward.  workspace: h1 [rows,lat] | h2 [rows,lat]
// The final sigmoid is a single-"numeric-field"-per-column head; done by a tiny kernel here.
// ---------------------------------------------------------------------------------------------
__global__ void k_sigmoid(float *__restrict__ y, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        y[e] = 1.f / (1.f + expf(-y[e]));
}

extern "C" size_t ddg_disc_workspace_bytes(int64_t rows, int lat_dim) {
    return 2 * align256((size_t)rows * lat_dim * 4);
}

extern "C" int ddg_disc_fwd(const ddg_disc_weights *w, const float *xtj1, const float *xt, const int64_t *t,
                            int t_uniform, float *out, int64_t rows, void *workspace, size_t workspace_bytes,
                            void *stream) {
    if (!w || !xtj1 || !xt || !out || rows < 1) return DDG_E_ARG;
    const int I = w->in_dim, L = w->lat_dim, O = w->out_dim;
    if (I < 2 || (I & 1) || L < 1 || O < 1) return DDG_E_ARG;
    if (!workspace || workspace_bytes < ddg_disc_workspace_bytes(rows, L)) return DDG_E_WORKSPACE;
    float *h1 = (float *)workspace;
    float *h2 = (float *)((char *)workspace + align256((size_t)rows * L * 4));
    int rc;
    rc = ddg_linear(xtj1, xt, I / 2, w->W1, w->b1, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0,
                    0, h1, nullptr, rows, I, L, stream);
    if (rc) return rc;
    rc = linear_impl(h1, nullptr, L, w->W2, w->b2, DDG_PRO_EMB_RELU, w->e1, w->n_steps, t, t_uniform, nullptr, nullptr, nullptr, 0.f,
                    0, 0, h2, nullptr, rows, L, L, stream);
    if (rc) return rc;
    rc = linear_impl(h2, nullptr, L, w->W3, w->b3, DDG_PRO_EMB_RELU, w->e2, w->n_steps, t, t_uniform, nullptr, nullptr, nullptr, 0.f,
                    0, 0, h1, nullptr, rows, L, L, stream);
    if (rc) return rc;
    rc = linear_impl(h1, nullptr, L, w->W4, w->b4, DDG_PRO_EMB_RELU, w->e3, w->n_steps, t, t_uniform, nullptr, nullptr, nullptr, 0.f,
                    0, 0, out, nullptr, rows, L, O, stream);
    if (rc) return rc;
    const int64_t n = rows * O;
    int64_t g = ceil_div64(n, 256);
    if (g > 4096) g = 4096;
    k_sigmoid<<<(int)g, 256, 0, (hipStream_t)stream>>>(out, n);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// this translation unit's copy of the stream-offset pointer (ddg_set_stream_offset, elementwise.hip)
int ddg_tu_set_stream_off_gemm(const uint64_t *dev_ptr) { return set_stream_off_tu(dev_ptr); }
