// Training side of the hot path at the reference's minibatch size (B = 128 rows): forward with
// saved activations, the three losses, hand-written backward of the denoiser and of the
// time-conditioned discriminator, and torch.optim.Adam.  gfx950 only.
//
// Reference functions restated (relative to the reference tree):
//   train_one_epoch_Generator                 model/Diffusion.py:245-318   (one step = :287-311)
//   train_one_epoch_discriminator_noise_x     model/Diffusion.py:321-379   (one step = :356-372)
//   res_loss_func / G_d_loss_func / D_loss_func   model/Diffusion.py:452-526
//   optim.Adam(G, lr 1e-4, wd 1e-4) / optim.Adam(Dn, lr 1e-4)   model/Diffusion.py:571-573
//
// What autograd does in the reference is written out here layer by layer.  All GEMMs (forward
// Y = A W^T, input gradients dA = dY W, weight gradients dW = dY^T A) run on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32): the step is 128 rows deep, so every GEMM is a handful of 32x32 tiles
// and the step is about 45 kernels of a few microseconds each: ddgani_amd replays it as a HIP graph in throughput mode
// (model/Diffusion.py::GraphedSteps; the random-stream numbers then move in device memory, ddg_set_stream_offset).
#include <math.h>
#include "common.h"

// ---------------------------------------------------------------------------------------------
// generic small GEMM, one workgroup (four waves splitting the depth) per 32x32 tile of C:  C[m,n] = sum_k A(m,k) * B(k,n)
// with element strides (sam, sak) / (sbk, sbn), so the same kernel serves NN, NT and TN.
// ---------------------------------------------------------------------------------------------
struct SmallGemm {
    const float *A; int64_t sam, sak;
    const float *B; int64_t sbk, sbn;
    float *C; int64_t ldc;
    int M, N, K;
};
__device__ __forceinline__ void gemm_small_tile(const SmallGemm &g, float (*red)[32 * 33]) {
    // the four waves of a workgroup take a quarter of the depth each; the partial tiles are added in LDS in a fixed order
    const int M = g.M, N = g.N, K = g.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= M || n0 >= N) return;                  // (a pair launch covers the larger of the two tile grids)
    const int m = m0 + r32, n = n0 + r32;
    const bool mok = m < M, nok = n < N;
    const float *Ap = g.A + (int64_t)(mok ? m : 0) * g.sam;
    const float *Bp = g.B + (int64_t)(nok ? n : 0) * g.sbn;
    const int64_t sak = g.sak, sbk = g.sbk;
    const int kper = ((K + 7) / 8) * 2;              // a multiple of 2
    const int kbeg = wid * kper, kend = min(K, kbeg + kper);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    int k = kbeg;
    for (; k + 16 <= kend; k += 16) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k + 2 * u + h;
            a[u] = mok ? Ap[(int64_t)kk * sak] : 0.f;
            b[u] = nok ? Bp[(int64_t)kk * sbk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < kend; k += 2) {
        const int kk = k + h;
        const float a = (mok && kk < kend) ? Ap[(int64_t)kk * sak] : 0.f;
        const float b = (nok && kk < kend) ? Bp[(int64_t)kk * sbk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wid][((q & 3) + 8 * (q >> 2) + 4 * h) * 33 + r32] = acc[q];
    __syncthreads();
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    if (n0 + c < N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = rg * 4 + i;
            if (m0 + rl < M)
                g.C[(int64_t)(m0 + rl) * g.ldc + n0 + c] =
                    (red[0][rl * 33 + c] + red[1][rl * 33 + c]) + (red[2][rl * 33 + c] + red[3][rl * 33 + c]);
        }
    }
}
__global__ __launch_bounds__(256) void k_gemm_small(SmallGemm g) {
    __shared__ float red[4][32 * 33];
    gemm_small_tile(g, red);
}
// two independent products in ONE launch (blockIdx.z picks): the weight gradient and the input gradient of a layer's
// backward both start from dY; at B = 128 each is a 5 us kernel on a few CUs and a launch saved is a sixth of its cost.
// Same arithmetic per output element as two k_gemm_small launches.
// A third plane (blockIdx.z == 2, when cs.out is set) sums the columns of dY: the layer's bias gradient, k_colsum's code.
struct ColSum { const float *X; int rows, N; float *out; };
__global__ __launch_bounds__(256) void k_gemm_small_pair(SmallGemm g0, SmallGemm g1, ColSum cs) {
    __shared__ float red[4][32 * 33];
    if (blockIdx.z == 0) { gemm_small_tile(g0, red); return; }
    if (blockIdx.z == 1) { gemm_small_tile(g1, red); return; }
    if (blockIdx.y != 0 || (int)blockIdx.x * 32 >= cs.N) return;
    float(*r8)[32] = reinterpret_cast<float(*)[32]>(&red[0][0]);
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float sum = 0.f;
    if (col < cs.N)
        for (int r = slice; r < cs.rows; r += 8) sum += cs.X[(int64_t)r * cs.N + col];
    r8[slice][c] = sum;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += r8[j][c];
    if (slice == 0 && col < cs.N) cs.out[col] = t;
}

static int gemm_small(const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn, float *Cm,
                      int64_t ldc, int M, int N, int K, hipStream_t s) {
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
    k_gemm_small<<<grid, 256, 0, s>>>(SmallGemm{A, sam, sak, B, sbk, sbn, Cm, ldc, M, N, K});
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
static int gemm_small_pair(const SmallGemm &g0, const SmallGemm &g1, hipStream_t s, ColSum cs = ColSum{nullptr, 0, 0, nullptr}) {
    int nmax = g0.N > g1.N ? g0.N : g1.N;
    if (cs.out && cs.N > nmax) nmax = cs.N;
    const int nx = (nmax + 31) / 32, ny = ((g0.M > g1.M ? g0.M : g1.M) + 31) / 32;
    k_gemm_small_pair<<<dim3((unsigned)nx, (unsigned)ny, cs.out ? 3 : 2), 256, 0, s>>>(g0, g1, cs);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
// dA[B,in] = dY[B,out] * W[out,in]
static SmallGemm op_dA(const float *dY, const float *W, float *dA, int rows, int in, int out) {
    return SmallGemm{dY, out, 1, W, in, 1, dA, in, rows, in, out};
}
// dW[out,in] = dY[B,out]^T * H[B,in]
static SmallGemm op_dW(const float *dY, const float *H, float *dW, int rows, int in, int out) {
    return SmallGemm{dY, 1, out, H, in, 1, dW, in, out, in, rows};
}
static int gemm_dA(const float *dY, const float *W, float *dA, int rows, int in, int out, hipStream_t s) {
    return gemm_small(dY, out, 1, W, in, 1, dA, in, rows, in, out, s);
}
static int gemm_dW(const float *dY, const float *H, float *dW, int rows, int in, int out, hipStream_t s) {
    return gemm_small(dY, 1, out, H, in, 1, dW, in, out, in, rows, s);
}
// both gradients of one layer's backward in one launch
// (db: the bias gradient = column sums of dY, or nullptr)
static int gemm_dW_dA(const float *dY, const float *H, float *dW, const float *W, float *dA, int rows, int in, int out,
                      hipStream_t s, float *db = nullptr) {
    return gemm_small_pair(op_dW(dY, H, dW, rows, in, out), op_dA(dY, W, dA, rows, in, out), s, ColSum{dY, rows, out, db});
}

// Column reductions over the B rows of a minibatch: a workgroup owns 32 columns, its 8 row slices (rows slice, slice + 8,
// ...) run in parallel and meet in LDS in a fixed order -- one thread per column walking all rows was a dependent chain
// of B loads (12-55 us per kernel at B = 128).
#define CR_SLICES 8
__device__ __forceinline__ float colred8(float v, float (*red)[32], int slice, int c) {
    red[slice][c] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < CR_SLICES; ++j) t += red[j][c];
    __syncthreads();
    return t;
}
static inline unsigned gcol32(int n) { return (unsigned)((n + 31) / 32); }

// out[col] = sum_r X[r,col]                                  (bias gradients)
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ X, int rows, int N, float *__restrict__ out) {
    __shared__ float red[CR_SLICES][32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    float s = 0.f;
    if (col < N)
        for (int r = slice; r < rows; r += CR_SLICES) s += X[(int64_t)r * N + col];
    s = colred8(s, red, slice, c);
    if (slice == 0 && col < N) out[col] = s;
}

// out[s,col] = sum over rows r with t[r] == s of X[r, col]   (embedding gradients, dense like torch's); grid (cols/32, n_steps)
__global__ __launch_bounds__(256) void k_emb_grad(const float *__restrict__ X, int64_t ldx, const int64_t *__restrict__ t,
                                                  int rows, int N, int n_steps, float *__restrict__ out) {
    __shared__ float red[CR_SLICES][32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c, st = blockIdx.y;
    float a = 0.f;
    if (col < N)
        for (int r = slice; r < rows; r += CR_SLICES)
            if ((int)t[r] == st) a += X[(int64_t)r * ldx + col];
    a = colred8(a, red, slice, c);
    if (slice == 0 && col < N) out[(int64_t)st * N + col] = a;
}

// (timesteps outside the embedding table are clamped in every kernel below: torch would raise)
__global__ void k_add_emb(const float *__restrict__ x, const float *__restrict__ emb, const int64_t *__restrict__ t,
                          int n_steps, int rows, int D, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * D) return;
    const int r = e / D, k = e - r * D;
    out[e] = x[e] + emb[(int64_t)min(max((int)t[r], 0), n_steps - 1) * D + k];
}

// BatchNorm(train) + ReLU + Dropout, forward, with the gate (0 or 1/(1-p) where the unit is on) saved.  Scale and shift
// come straight from the fp64 column sums (k_bn_finalize's arithmetic, one launch less per layer).
__global__ void k_bn_act_fwd(const float *__restrict__ z, const double *__restrict__ stats, double n,
                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                             const uint8_t *__restrict__ drop, float drop_p,
                             uint64_t seed, uint64_t stream_id, int rows, int width, float *__restrict__ hout,
                             float *__restrict__ gate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width) return;
    const int k = e % width;
    const double mean = stats[k] / n;
    double var = stats[width + k] / n - mean * mean;
    if (var < 0) var = 0;
    const double sc = (double)gamma[k] * (1.0 / sqrt(var + (double)eps));
    const float bn = z[e] * (float)sc + (float)((double)beta[k] - mean * sc);
    float g = bn > 0.f ? 1.f : 0.f;
    const float keep_scale = 1.f / (1.f - drop_p);
    if (drop) {
        g = drop[e] ? g * keep_scale : 0.f;
    } else if (drop_p > 0.f) {
        const float u = philox_uniform4(seed, stream_id, (uint64_t)e >> 2)[e & 3];
        g = (u > drop_p) ? g * keep_scale : 0.f;
    }
    hout[e] = fmaxf(bn, 0.f) * g;
    gate[e] = g;
}

// backward of the above for one layer; 32 columns x 8 row slices per workgroup (see colred8).
//   da = dh * gate ; dbeta = sum da ; dgamma = sum da * xhat ; dz = gamma*invstd*(da - dbeta/B - xhat*dgamma/B)
__global__ __launch_bounds__(256) void k_bn_act_bwd(const float *__restrict__ dh, const float *__restrict__ gate,
                                                    const float *__restrict__ z, const double *__restrict__ stats,
                                                    const float *__restrict__ gamma, float eps, int rows, int width,
                                                    float *__restrict__ dz, float *__restrict__ dgamma,
                                                    float *__restrict__ dbeta, float *__restrict__ dbias) {
    __shared__ float red[CR_SLICES][32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + c;
    const bool ok = k < width;
    const int kk = ok ? k : 0;
    const double mean_d = stats[kk] / rows;
    double var = stats[width + kk] / rows - mean_d * mean_d;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mean_d;
    float sda = 0.f, sdx = 0.f;
    if (ok)
        for (int r = slice; r < rows; r += CR_SLICES) {
            const int64_t e = (int64_t)r * width + k;
            const float da = dh[e] * gate[e];
            sda += da;
            sdx += da * ((z[e] - mean) * invstd);
        }
    sda = colred8(sda, red, slice, c);
    sdx = colred8(sdx, red, slice, c);
    if (ok && slice == 0) {
        dgamma[k] = sdx;
        dbeta[k] = sda;
    }
    const float gi = gamma[kk] * invstd, inv_n = 1.f / (float)rows;
    float sdz = 0.f;                         // gradient of the layer's bias = column sum of dz (one launch less per layer)
    if (ok)
        for (int r = slice; r < rows; r += CR_SLICES) {
            const int64_t e = (int64_t)r * width + k;
            const float da = dh[e] * gate[e];
            const float xh = (z[e] - mean) * invstd;
            const float v = gi * (da - sda * inv_n - xh * sdx * inv_n);
            dz[e] = v;
            sdz += v;
        }
    sdz = colred8(sdz, red, slice, c);
    if (ok && slice == 0) dbias[k] = sdz;
}

// The same backward split at its batch reduction (data-parallel step: the two column sums are all-reduced between
// the kernels).  `rows` = local rows, `rows_g` = rows of the global batch (BatchNorm statistics are global).
__global__ __launch_bounds__(256) void k_bn_bwd_sums(const float *__restrict__ dh, const float *__restrict__ gate,
                                                     const float *__restrict__ z, const double *__restrict__ stats, float eps,
                                                     int rows, double rows_g, int width, float *__restrict__ sums,
                                                     float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[CR_SLICES][32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + c;
    const bool ok = k < width;
    const int kk = ok ? k : 0;
    const double mean_d = stats[kk] / rows_g;
    double var = stats[width + kk] / rows_g - mean_d * mean_d;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mean_d;
    float sda = 0.f, sdx = 0.f;
    if (ok)
        for (int r = slice; r < rows; r += CR_SLICES) {
            const int64_t e = (int64_t)r * width + k;
            const float da = dh[e] * gate[e];
            sda += da;
            sdx += da * ((z[e] - mean) * invstd);
        }
    sda = colred8(sda, red, slice, c);
    sdx = colred8(sdx, red, slice, c);
    if (ok && slice == 0) {
        dgamma[k] = sdx;          // this rank's share: the gradient all-reduce completes it
        dbeta[k] = sda;
        sums[k] = sda;
        sums[width + k] = sdx;
    }
}
__global__ void k_bn_bwd_apply(const float *__restrict__ dh, const float *__restrict__ gate, const float *__restrict__ z,
                               const double *__restrict__ stats, const float *__restrict__ gamma, float eps, int rows,
                               double rows_g, int width, const float *__restrict__ sums, float *__restrict__ dz) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width) return;
    const int k = e % width;
    const double mean_d = stats[k] / rows_g;
    double var = stats[width + k] / rows_g - mean_d * mean_d;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mean_d;
    const float gi = gamma[k] * invstd, inv_n = 1.f / (float)rows_g;
    const float da = dh[e] * gate[e];
    const float xh = (z[e] - mean) * invstd;
    dz[e] = gi * (da - sums[k] * inv_n - xh * sums[width + k] * inv_n);
}

// rows of the local batch whose mask block of field f is all ones (res_loss_func's per-field row count)
__global__ __launch_bounds__(256) void k_res_ngood(const float *__restrict__ M, const int32_t *__restrict__ offsets, int C,
                                                   int rows, int D, float *__restrict__ ngood) {
    __shared__ int s_cnt[4];
    const int f = blockIdx.x;                 // one workgroup per field, its threads take the rows
    const int a = offsets[f], b = offsets[f + 1];
    int cnt = 0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        bool good = true;
        for (int j = a; j < b; ++j) good = good && (M[(int64_t)r * D + j] == 1.f);
        cnt += good ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) ngood[f] = (float)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

// discriminator hidden activation r = relu(a + e[t]) and its backward
__global__ void k_emb_relu_fwd(const float *__restrict__ a, const float *__restrict__ emb,
                               const int64_t *__restrict__ t, int n_steps, int rows, int width,
                               float *__restrict__ r_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width) return;
    const int r = e / width, k = e - r * width;
    r_out[e] = fmaxf(a[e] + emb[(int64_t)min(max((int)t[r], 0), n_steps - 1) * width + k], 0.f);
}
__global__ void k_relu_bwd(const float *__restrict__ dr, const float *__restrict__ r_act, int n,
                           float *__restrict__ da) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    da[e] = r_act[e] > 0.f ? dr[e] : 0.f;
}

// x_{t-1} of G_d_loss_func / D_loss_func (model/Diffusion.py:487-502), reference op order
#pragma clang fp contract(off)
__global__ void k_xtj1_train(const float *__restrict__ pred, const float *__restrict__ xt,
                             const float *__restrict__ x0, const float *__restrict__ eps,
                             const float *__restrict__ M, const int64_t *__restrict__ t,
                             const float *__restrict__ sched, int T, int rows, int D, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * D) return;
    const int r = e / D;
    const int ti = min(max((int)t[r], 2), T);
    const float *betas = sched, *sa = sched + 2 * T, *ac = sched + 5 * T, *sac = sched + 6 * T, *s1m = sched + 7 * T;
    const float o = sac[ti - 2] * x0[e] + s1m[ti - 2] * eps[e];
    const float one_m = 1.f - ac[ti - 2];
    const float den = 1.f - ac[ti - 1];
    const float gm = (sac[ti - 2] * betas[ti - 1] * pred[e] + sa[ti - 1] * one_m * xt[e]) / den;
    const float gs = one_m * betas[ti - 1] * eps[e] / den;
    const float g = gm + gs;
    out[e] = M[e] * o + (1.f - M[e]) * g;
}
#pragma clang fp contract(on)

// res_loss_func (model/Diffusion.py:452-484): value and gradient w.r.t. the prediction.  One workgroup per table field:
// its threads take the rows, count the rows whose mask block is complete (unless the caller hands over the counts of the
// global batch), then write the gradient and reduce the field's loss sum into lpart[f] (fixed order).  k_head_bwd, the
// next kernel of the step, adds the fields up in field order and writes the loss value.
__global__ __launch_bounds__(256) void k_res_loss_grad(const float *__restrict__ pred, const float *__restrict__ x0,
                                                       const float *__restrict__ M, const int32_t *__restrict__ offsets,
                                                       const uint8_t *__restrict__ is_cat, int C, int rows, int D,
                                                       const float *__restrict__ ngood_global, float *__restrict__ dpred,
                                                       double *__restrict__ lpart) {
    __shared__ int s_cnt[4];
    __shared__ double s_l[4];
    const int f = blockIdx.x;
    const int a = offsets[f], b = offsets[f + 1];
    const bool cat = is_cat[f] != 0;
    int nn = 0, nc = 0;
    for (int g = 0; g < C; ++g) { if (is_cat[g]) ++nc; else ++nn; }
    const float nfield = (float)(cat ? (nc > 0 ? nc : 1) : (nn > 0 ? nn : 1));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ng;
    if (ngood_global) {          // data-parallel step: the counts of the GLOBAL batch (all-reduced by the caller)
        ng = ngood_global[f];
    } else {
        int cnt = 0;
        for (int r = threadIdx.x; r < rows; r += blockDim.x) {
            bool good = true;
            for (int j = a; j < b; ++j) good = good && (M[(int64_t)r * D + j] == 1.f);
            cnt += good ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) s_cnt[wid] = cnt;
        __syncthreads();
        ng = (float)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    }
    double l = 0.0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        bool good = true;
        for (int j = a; j < b; ++j) good = good && (M[(int64_t)r * D + j] == 1.f);
        const float *p = pred + (int64_t)r * D, *x = x0 + (int64_t)r * D;
        float *dp = dpred + (int64_t)r * D;
        if (!good) {
            for (int j = a; j < b; ++j) dp[j] = 0.f;
            continue;
        }
        if (!cat) {
            const float nel = ng * (float)(b - a);
            double lr = 0.0;
            for (int j = a; j < b; ++j) {
                const float d = p[j] - x[j];
                dp[j] = 2.f * d / (nel * nfield);
                lr += (double)d * d;
            }
            l += lr / (double)nel;
        } else {
            float sdot = 0.f;
            for (int j = a; j < b; ++j) sdot += (p[j] + 1e-6f) * x[j];
            sdot += 1e-6f;
            for (int j = a; j < b; ++j) dp[j] = -x[j] / sdot / (ng * nfield);
            l += -(double)logf(sdot) / (double)ng;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    if (lane == 0) s_l[wid] = l;
    __syncthreads();
    if (threadIdx.x == 0) lpart[f] = ((s_l[0] + s_l[1]) + (s_l[2] + s_l[3])) / (double)nfield;
}

// discriminator output, loss and d(loss)/d(logit).  mode 0: G_d_loss_func (:504); mode 1: D_loss_func (:523)
__global__ void k_disc_loss_grad(const float *__restrict__ a4, const float *__restrict__ m, int n, int64_t n_norm,
                                 int mode, float scale, float *__restrict__ da4, float *__restrict__ loss_out) {
    __shared__ double part[1024];
    double l = 0.0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const float d = 1.f / (1.f + expf(-a4[e]));
        const float mm = m[e];
        float dl;   // d loss_sum / d d
        if (mode == 0) {
            l += -(double)((1.f - mm) * logf(d + 1e-8f));
            dl = -(1.f - mm) / (d + 1e-8f);
        } else {
            l += -(double)((1.f - mm) * logf(1.f - d + 1e-4f) + mm * logf(d + 1e-4f));
            dl = (1.f - mm) / (1.f - d + 1e-4f) - mm / (d + 1e-4f);
        }
        da4[e] = scale * dl / (float)n_norm * d * (1.f - d);
    }
    part[threadIdx.x] = l;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)(part[0] / (double)n_norm);
}

// total d(loss)/d(pred) = w_res * dres + dxtj1 * (1-M) * sqrt(abar_{t-1}) beta_t / (1-abar_t), then the
// backward of the per-field head (sigmoid / softmax): one thread per (row, field)
__global__ void k_head_bwd(const float *__restrict__ dres, float w_res, const float *__restrict__ dcat, int64_t ldcat,
                           const float *__restrict__ M, const int64_t *__restrict__ t,
                           const float *__restrict__ sched, int T, const float *__restrict__ pred,
                           const int32_t *__restrict__ offsets, const uint8_t *__restrict__ is_cat, int C, int rows,
                           int D, float *__restrict__ dy, const double *__restrict__ lpart,
                           float *__restrict__ res_loss_out) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0 && lpart) {       // res_loss value: the fields' sums of k_res_loss_grad, numeric ones first, in field order
        double ln = 0.0, lc = 0.0;
        for (int g = 0; g < C; ++g) { if (is_cat[g]) lc += lpart[g]; else ln += lpart[g]; }
        res_loss_out[0] = (float)(ln + lc);
    }
    if (w >= rows * C) return;
    const int r = w / C, f = w - r * C;
    const int a = offsets[f], b = offsets[f + 1];
    const int ti = min(max((int)t[r], 2), T);
    const float coef = sched[6 * T + ti - 2] * sched[ti - 1] / (1.f - sched[5 * T + ti - 1]);
    const float *p = pred + (int64_t)r * D;
    auto dp = [&](int j) {
        const int64_t e = (int64_t)r * D + j;
        float v = dres ? w_res * dres[e] : 0.f;
        if (dcat) v += dcat[(int64_t)r * ldcat + j] * (1.f - M[e]) * coef;
        return v;
    };
    if (is_cat[f]) {
        float dot = 0.f;
        for (int j = a; j < b; ++j) dot += dp(j) * p[j];
        for (int j = a; j < b; ++j) dy[(int64_t)r * D + j] = p[j] * (dp(j) - dot);
    } else {
        for (int j = a; j < b; ++j) dy[(int64_t)r * D + j] = dp(j) * p[j] * (1.f - p[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// torch.optim.Adam (non-amsgrad, coupled L2), all tensors of a net in one launch
// ---------------------------------------------------------------------------------------------
#define ADAM_MAX_TENSORS 16
struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
    int count;
};

#pragma clang fp contract(off)
__global__ void k_adam(AdamTable tb, float step_size, float one_m_b1, float b2, float one_m_b2, float eps, float wd,
                       float sqrt_bc2) {
    const int ti = blockIdx.y;
    if (ti >= tb.count) return;
    float *p = tb.p[ti];
    const float *g = tb.g[ti];
    float *m = tb.m[ti], *v = tb.v[ti];
    const int64_t n = tb.n[ti];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[e];
        if (wd != 0.f) gg = gg + wd * p[e];
        // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2)
        const float mn = m[e] + one_m_b1 * (gg - m[e]);
        const float vn = v[e] * b2 + one_m_b2 * gg * gg;
        const float denom = sqrtf(vn) / sqrt_bc2 + eps;
        m[e] = mn;
        v[e] = vn;
        p[e] = p[e] - step_size * (mn / denom);
    }
}
#pragma clang fp contract(on)

extern "C" int ddg_adam_step(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                             float *const *exp_avg_sq, const int64_t *numel, int step, double lr, double beta1,
                             double beta2, double eps, double weight_decay, void *stream) {
    if (n_tensors < 1 || n_tensors > ADAM_MAX_TENSORS || !params || !grads || !exp_avg || !exp_avg_sq || !numel ||
        step < 1)
        return DDG_E_ARG;
    AdamTable tb;
    int64_t nmax = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return DDG_E_ARG;
        tb.p[i] = params[i]; tb.g[i] = grads[i]; tb.m[i] = exp_avg[i]; tb.v[i] = exp_avg_sq[i]; tb.n[i] = numel[i];
        if (numel[i] > nmax) nmax = numel[i];
    }
    tb.count = n_tensors;
    // scalar prefactors in double like torch's Python-side arithmetic, tensors in fp32
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    int64_t gx = ceil_div64(nmax, 256);
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)n_tensors);
    k_adam<<<grid, 256, 0, (hipStream_t)stream>>>(tb, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                                                  (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                                                  (float)sqrt(bc2));
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// workspace of one training step
// ---------------------------------------------------------------------------------------------
static inline size_t tal(size_t x) { return (x + 255) & ~(size_t)255; }

struct TrainWs {
    float *xt, *h0, *z1, *h1, *g1, *z2, *h2, *g2, *z3, *h3, *g3, *y, *pred;
    double *st[3];
    float *sc[3], *sf[3];
    float *xtj1, *a1, *r1, *a2, *r2, *a3, *r3, *a4, *da4, *dA, *dB, *dcat;
    float *dres, *dy, *dh, *dz;
    double *lpart;             // res_loss_func: per-field loss sums [C]
    float *bsum, *ngood;       // data-parallel step: BatchNorm backward column sums [2*max width], per-field row counts [C]
    float *fwd_ws;             // workspace of ddg_denoiser_fwd (discriminator step)
    size_t fwd_ws_bytes;
    size_t total;
};

static void train_carve(TrainWs &w, char *base, int64_t B, int D, int E, int L, int H, int C) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += tal(bytes); return p; };
    const size_t mx = (size_t)(E > L ? E : L);
    auto f = [&](size_t n) { return (float *)take(n * 4); };
    w.xt = f(B * D); w.h0 = f(B * D);
    w.z1 = f(B * E); w.h1 = f(B * E); w.g1 = f(B * E);
    w.z2 = f(B * L); w.h2 = f(B * L); w.g2 = f(B * L);
    w.z3 = f(B * E); w.h3 = f(B * E); w.g3 = f(B * E);
    w.y = f(B * D); w.pred = f(B * D);
    for (int i = 0; i < 3; ++i) w.st[i] = (double *)take(2 * mx * 8);
    for (int i = 0; i < 3; ++i) { w.sc[i] = f(mx); w.sf[i] = f(mx); }
    w.xtj1 = f(B * D);
    w.a1 = f(B * H); w.r1 = f(B * H); w.a2 = f(B * H); w.r2 = f(B * H); w.a3 = f(B * H); w.r3 = f(B * H);
    w.a4 = f(B * C); w.da4 = f(B * C); w.dA = f(B * H); w.dB = f(B * H); w.dcat = f(B * 2 * D);
    w.dres = f(B * D); w.dy = f(B * D); w.dh = f(B * mx); w.dz = f(B * mx);
    w.bsum = f(2 * mx); w.ngood = f(C);
    w.lpart = (double *)take((size_t)C * 8);
    w.fwd_ws_bytes = ddg_denoiser_workspace_bytes(B, D, E, L);
    w.fwd_ws = (float *)take(w.fwd_ws_bytes);
    w.total = o;
}

extern "C" size_t ddg_train_workspace_bytes(int64_t batch_rows, int in_dim, int enc_dim, int lat_dim, int disc_lat_dim,
                                            int C) {
    if (batch_rows < 1 || in_dim < 1 || enc_dim < 1 || lat_dim < 1 || disc_lat_dim < 1 || C < 1) return 0;
    TrainWs w;
    train_carve(w, nullptr, batch_rows, in_dim, enc_dim, lat_dim, disc_lat_dim, C);
    return w.total;
}

static inline unsigned g256(int64_t n) { return (unsigned)((n + 255) / 256); }

static int check_batch(const ddg_train_batch *b) {
    if (!b || !b->x0 || !b->M || !b->m || !b->t || b->rows < 1 || b->rows > (1 << 20)) return DDG_E_ARG;
    if (b->drop_p < 0.f || b->drop_p >= 1.f) return DDG_E_ARG;
    return DDG_OK;
}

// forward of the denoiser with every activation kept (model/Diffusion.py:179-201, train mode)
// data-parallel context of one step: rows of the global batch and the caller's all-reduce (NULL: single process)
struct Dp {
    int64_t Bg;
    ddg_allreduce_fn ar;
    void *ctx;
};
static Dp dp_of(const ddg_dp *dp, int64_t local_rows) {
    Dp d;
    d.Bg = (dp && dp->allreduce) ? dp->global_rows : local_rows;
    d.ar = dp ? dp->allreduce : nullptr;
    d.ctx = dp ? dp->ctx : nullptr;
    return d;
}
static int dp_allreduce(const Dp &dp, void *buf, int64_t count, int is_f64, void *stream) {
    if (!dp.ar) return DDG_OK;
    return dp.ar(dp.ctx, buf, count, is_f64, stream) == 0 ? DDG_OK : DDG_E_COLLECTIVE;
}

static int gen_forward_saved(const ddg_gen_weights *w, const ddg_train_batch *b, const float *sched, int T,
                             const int32_t *offsets, const uint8_t *is_cat, int C, float bn_eps, TrainWs &ws,
                             void *stream, const Dp &dp) {
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)b->rows, D = w->in_dim, E = w->enc_dim, L = w->lat_dim;
    int rc = ddg_forward_noise(b->x0, b->eps, b->t, 0, sched, T, b->seed, b->stream_id * 8 + 7, ws.xt,
                               b->eps ? nullptr : ws.dres /* on-chip eps kept for the x_{t-1} build */, B, D, stream);
    if (rc) return rc;
    k_add_emb<<<g256((int64_t)B * D), 256, 0, s>>>(ws.xt, w->temb, b->t, w->n_steps, B, D, ws.h0);
    const size_t mx = (size_t)(E > L ? E : L);
    if ((rc = zero_async(ws.st[0], 3 * tal(2 * mx * 8), s))) return rc;      // not hipMemsetAsync: see common.h
    const float *Wl[3] = {w->W1, w->W2, w->W3}, *bl[3] = {w->b1, w->b2, w->b3};
    const float *gl[3] = {w->g1, w->g2, w->g3}, *bel[3] = {w->be1, w->be2, w->be3};
    const uint8_t *dl[3] = {b->drop0, b->drop1, b->drop2};
    float *zl[3] = {ws.z1, ws.z2, ws.z3}, *hl[3] = {ws.h1, ws.h2, ws.h3}, *gt[3] = {ws.g1, ws.g2, ws.g3};
    const int win[3] = {D, E, L}, wout[3] = {E, L, E};
    const float *hin = ws.h0;
    for (int l = 0; l < 3; ++l) {
        rc = ddg_linear(hin, nullptr, win[l], Wl[l], bl[l], DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr,
                        nullptr, 0.f, 0, 0, zl[l], ws.st[l], B, win[l], wout[l], stream);
        if (rc) return rc;
        if ((rc = dp_allreduce(dp, ws.st[l], 2 * wout[l], 1, stream))) return rc;      // sync-BatchNorm: global sums
        k_bn_act_fwd<<<g256((int64_t)B * wout[l]), 256, 0, s>>>(zl[l], ws.st[l], (double)dp.Bg, gl[l], bel[l], bn_eps, dl[l], b->drop_p, b->seed,
                                                               b->stream_id * 4 + l, B, wout[l], hl[l], gt[l]);
        hin = hl[l];
    }
    rc = ddg_linear(ws.h3, nullptr, E, w->W4, w->b4, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                    0, 0, ws.y, nullptr, B, E, D, stream);
    if (rc) return rc;
    rc = ddg_head(ws.y, offsets, is_cat, C, ws.pred, B, D, stream);
    if (rc) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// discriminator forward with saved activations on xtj1 built from `pred`
static int disc_forward_saved(const ddg_disc_weights *dw, const ddg_train_batch *b, const float *pred,
                              const float *eps, const float *sched, int T, int D, TrainWs &ws, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)b->rows, H = dw->lat_dim, C = dw->out_dim;
    k_xtj1_train<<<g256((int64_t)B * D), 256, 0, s>>>(pred, ws.xt, b->x0, eps, b->M, b->t, sched, T, B, D, ws.xtj1);
    int rc = ddg_linear(ws.xtj1, ws.xt, D, dw->W1, dw->b1, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                        0.f, 0, 0, ws.a1, nullptr, B, 2 * D, H, stream);
    if (rc) return rc;
    k_emb_relu_fwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.a1, dw->e1, b->t, dw->n_steps, B, H, ws.r1);
    rc = ddg_linear(ws.r1, nullptr, H, dw->W2, dw->b2, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                    0, 0, ws.a2, nullptr, B, H, H, stream);
    if (rc) return rc;
    k_emb_relu_fwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.a2, dw->e2, b->t, dw->n_steps, B, H, ws.r2);
    rc = ddg_linear(ws.r2, nullptr, H, dw->W3, dw->b3, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                    0, 0, ws.a3, nullptr, B, H, H, stream);
    if (rc) return rc;
    k_emb_relu_fwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.a3, dw->e3, b->t, dw->n_steps, B, H, ws.r3);
    rc = ddg_linear(ws.r3, nullptr, H, dw->W4, dw->b4, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                    0, 0, ws.a4, nullptr, B, H, C, stream);
    if (rc) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// generator step: loss = w_res * Res + w_gd * G_D, gradients of every generator parameter
// ---------------------------------------------------------------------------------------------
extern "C" int ddg_train_gen_grads_dp(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                                      const float *sched_dev, int T, const int32_t *offsets_dev,
                                      const uint8_t *is_cat_dev, int C, float w_res, float w_gd, float bn_eps,
                                      const ddg_gen_grads *grads, float *losses_dev, float *pred_out, void *workspace,
                                      size_t workspace_bytes, void *stream, const ddg_dp *dpp) {
    int rc = check_batch(b);
    if (rc) return rc;
    const Dp dp = dp_of(dpp, b->rows);
    if (dp.Bg < b->rows) return DDG_E_ARG;
    if (!gw || !dw || !sched_dev || !offsets_dev || !is_cat_dev || !grads || !losses_dev || C < 1 || T < 2)
        return DDG_E_ARG;
    const int B = (int)b->rows, D = gw->in_dim, E = gw->enc_dim, L = gw->lat_dim, H = dw->lat_dim;
    if (dw->in_dim != 2 * D || dw->out_dim != C) return DDG_E_ARG;
    TrainWs ws;
    train_carve(ws, (char *)workspace, B, D, E, L, H, C);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;

    rc = gen_forward_saved(gw, b, sched_dev, T, offsets_dev, is_cat_dev, C, bn_eps, ws, stream, dp);
    if (rc) return rc;
    const float *eps = b->eps ? b->eps : ws.dres;
    // G_D branch first (it reads the on-chip eps parked in ws.dres), then the reconstruction loss
    rc = disc_forward_saved(dw, b, ws.pred, eps, sched_dev, T, D, ws, stream);
    if (rc) return rc;
    k_disc_loss_grad<<<1, 1024, 0, s>>>(ws.a4, b->m, B * C, dp.Bg * C, 0, w_gd, ws.da4, losses_dev + 1);
    // input gradient of the discriminator: da4 -> r3 -> a3 -> r2 -> a2 -> r1 -> a1 -> cat(xtj1, x_t)
    if ((rc = gemm_dA(ws.da4, dw->W4, ws.dA, B, H, C, s))) return rc;
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r3, B * H, ws.dB);
    if ((rc = gemm_dA(ws.dB, dw->W3, ws.dA, B, H, H, s))) return rc;
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r2, B * H, ws.dB);
    if ((rc = gemm_dA(ws.dB, dw->W2, ws.dA, B, H, H, s))) return rc;
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r1, B * H, ws.dB);
    if ((rc = gemm_dA(ws.dB, dw->W1, ws.dcat, B, 2 * D, H, s))) return rc;

    const float *ngood = nullptr;
    if (dp.ar) {                 // res_loss_func normalises per field by the row count of the GLOBAL batch
        k_res_ngood<<<C, 256, 0, s>>>(b->M, offsets_dev, C, B, D, ws.ngood);
        if ((rc = dp_allreduce(dp, ws.ngood, C, 0, stream))) return rc;
        ngood = ws.ngood;
    }
    k_res_loss_grad<<<C, 256, 0, s>>>(ws.pred, b->x0, b->M, offsets_dev, is_cat_dev, C, B, D, ngood, ws.dres, ws.lpart);
    k_head_bwd<<<g256((int64_t)B * C), 256, 0, s>>>(ws.dres, w_res, ws.dcat, 2 * D, b->M, b->t, sched_dev, T, ws.pred,
                                                   offsets_dev, is_cat_dev, C, B, D, ws.dy, ws.lpart, losses_dev);
    // layer 4
    if ((rc = gemm_dW_dA(ws.dy, ws.h3, grads->W4, gw->W4, ws.dh, B, E, D, s, grads->b4))) return rc;
    // layers 3, 2, 1
    const float *Wl[3] = {gw->W1, gw->W2, gw->W3}, *gl[3] = {gw->g1, gw->g2, gw->g3};
    float *dWl[3] = {grads->W1, grads->W2, grads->W3}, *dbl[3] = {grads->b1, grads->b2, grads->b3};
    float *dgl[3] = {grads->g1, grads->g2, grads->g3}, *dbel[3] = {grads->be1, grads->be2, grads->be3};
    float *zl[3] = {ws.z1, ws.z2, ws.z3}, *gt[3] = {ws.g1, ws.g2, ws.g3};
    const float *hin[3] = {ws.h0, ws.h1, ws.h2};
    const int win[3] = {D, E, L}, wout[3] = {E, L, E};
    for (int l = 2; l >= 0; --l) {
        if (dp.ar) {
            k_bn_bwd_sums<<<gcol32(wout[l]), 256, 0, s>>>(ws.dh, gt[l], zl[l], ws.st[l], bn_eps, B, (double)dp.Bg,
                                                             wout[l], ws.bsum, dgl[l], dbel[l]);
            if ((rc = dp_allreduce(dp, ws.bsum, 2 * wout[l], 0, stream))) return rc;
            k_bn_bwd_apply<<<g256((int64_t)B * wout[l]), 256, 0, s>>>(ws.dh, gt[l], zl[l], ws.st[l], gl[l], bn_eps, B,
                                                                     (double)dp.Bg, wout[l], ws.bsum, ws.dz);
        } else {
            k_bn_act_bwd<<<gcol32(wout[l]), 256, 0, s>>>(ws.dh, gt[l], zl[l], ws.st[l], gl[l], bn_eps, B, wout[l],
                                                            ws.dz, dgl[l], dbel[l], dbl[l]);
        }
        if ((rc = gemm_dW_dA(ws.dz, hin[l], dWl[l], Wl[l], ws.dh, B, win[l], wout[l], s, dp.ar ? dbl[l] : nullptr))) return rc;
    }
    k_emb_grad<<<dim3(gcol32(D), (unsigned)gw->n_steps), 256, 0, s>>>(ws.dh, D, b->t, B, D, gw->n_steps, grads->temb);
    if (pred_out && (rc = copy_f32_async(pred_out, ws.pred, (size_t)B * D, s))) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_train_gen_grads(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                                   const float *sched_dev, int T, const int32_t *offsets_dev,
                                   const uint8_t *is_cat_dev, int C, float w_res, float w_gd, float bn_eps,
                                   const ddg_gen_grads *grads, float *losses_dev, float *pred_out, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    return ddg_train_gen_grads_dp(gw, dw, b, sched_dev, T, offsets_dev, is_cat_dev, C, w_res, w_gd, bn_eps, grads,
                                  losses_dev, pred_out, workspace, workspace_bytes, stream, nullptr);
}

// ---------------------------------------------------------------------------------------------
// discriminator step: loss = loss_scale * D_loss, gradients of every discriminator parameter
// ---------------------------------------------------------------------------------------------
extern "C" int ddg_train_disc_grads_dp(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                                       const float *sched_dev, int T, const int32_t *offsets_dev,
                                       const uint8_t *is_cat_dev, int C, float loss_scale, float bn_eps,
                                       const ddg_disc_grads *grads, float *loss_dev, float *pred_out, void *workspace,
                                       size_t workspace_bytes, void *stream, const ddg_dp *dpp) {
    int rc = check_batch(b);
    if (rc) return rc;
    const Dp dp = dp_of(dpp, b->rows);
    if (dp.Bg < b->rows) return DDG_E_ARG;
    if (!gw || !dw || !sched_dev || !offsets_dev || !is_cat_dev || !grads || !loss_dev || C < 1 || T < 2)
        return DDG_E_ARG;
    const int B = (int)b->rows, D = gw->in_dim, E = gw->enc_dim, L = gw->lat_dim, H = dw->lat_dim;
    if (dw->in_dim != 2 * D || dw->out_dim != C) return DDG_E_ARG;
    TrainWs ws;
    train_carve(ws, (char *)workspace, B, D, E, L, H, C);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;

    if (dp.ar) {
        // data-parallel: the layer-by-layer forward, which has the sync-BatchNorm seams
        rc = gen_forward_saved(gw, b, sched_dev, T, offsets_dev, is_cat_dev, C, bn_eps, ws, stream, dp);
        if (rc) return rc;
    } else {
        // generator forward only (fused inference path, train-mode BN + dropout)
        rc = ddg_forward_noise(b->x0, b->eps, b->t, 0, sched_dev, T, b->seed, b->stream_id * 8 + 7, ws.xt,
                               b->eps ? nullptr : ws.dres, B, D, stream);
        if (rc) return rc;
        rc = ddg_denoiser_fwd(gw, ws.xt, b->t, 0, b->drop0, b->drop1, b->drop2, b->drop_p, b->seed, b->stream_id,
                              bn_eps, offsets_dev, is_cat_dev, C, 1, ws.pred, B, ws.fwd_ws, ws.fwd_ws_bytes, stream);
        if (rc) return rc;
    }
    const float *eps = b->eps ? b->eps : ws.dres;
    rc = disc_forward_saved(dw, b, ws.pred, eps, sched_dev, T, D, ws, stream);
    if (rc) return rc;
    k_disc_loss_grad<<<1, 1024, 0, s>>>(ws.a4, b->m, B * C, dp.Bg * C, 1, loss_scale, ws.da4, loss_dev);
    // layer 4
    if ((rc = gemm_dW_dA(ws.da4, ws.r3, grads->W4, dw->W4, ws.dA, B, H, C, s, grads->b4))) return rc;
    // layer 3
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r3, B * H, ws.dB);
    k_emb_grad<<<dim3(gcol32(H), (unsigned)dw->n_steps), 256, 0, s>>>(ws.dB, H, b->t, B, H, dw->n_steps, grads->e3);
    if ((rc = gemm_dW_dA(ws.dB, ws.r2, grads->W3, dw->W3, ws.dA, B, H, H, s, grads->b3))) return rc;
    // layer 2
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r2, B * H, ws.dB);
    k_emb_grad<<<dim3(gcol32(H), (unsigned)dw->n_steps), 256, 0, s>>>(ws.dB, H, b->t, B, H, dw->n_steps, grads->e2);
    if ((rc = gemm_dW_dA(ws.dB, ws.r1, grads->W2, dw->W2, ws.dA, B, H, H, s, grads->b2))) return rc;
    // layer 1: input = cat(xtj1, x_t)
    k_relu_bwd<<<g256((int64_t)B * H), 256, 0, s>>>(ws.dA, ws.r1, B * H, ws.dB);
    k_emb_grad<<<dim3(gcol32(H), (unsigned)dw->n_steps), 256, 0, s>>>(ws.dB, H, b->t, B, H, dw->n_steps, grads->e1);
    if ((rc = gemm_small_pair(SmallGemm{ws.dB, 1, H, ws.xtj1, D, 1, grads->W1, 2 * D, H, D, B},
                              SmallGemm{ws.dB, 1, H, ws.xt, D, 1, grads->W1 + D, 2 * D, H, D, B}, s,
                              ColSum{ws.dB, B, H, grads->b1}))) return rc;
    if (pred_out && (rc = copy_f32_async(pred_out, ws.pred, (size_t)B * D, s))) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_train_disc_grads(const ddg_gen_weights *gw, const ddg_disc_weights *dw, const ddg_train_batch *b,
                                    const float *sched_dev, int T, const int32_t *offsets_dev,
                                    const uint8_t *is_cat_dev, int C, float loss_scale, float bn_eps,
                                    const ddg_disc_grads *grads, float *loss_dev, float *pred_out, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return ddg_train_disc_grads_dp(gw, dw, b, sched_dev, T, offsets_dev, is_cat_dev, C, loss_scale, bn_eps, grads,
                                   loss_dev, pred_out, workspace, workspace_bytes, stream, nullptr);
}

// ---------------------------------------------------------------------------------------------
// (f4) the classifiers of the "data-dependency" plug-in            model/FD_model.py:12-65
//
// FDModel.forward is fc1 -> ReLU -> fc3 (fc2 is constructed and never used, :22-26); train_Model runs
// minibatch SGD (batch 256, lr 0.02, CrossEntropyLoss) over shuffled rows (:33-52); test_model counts arg-max
// hits (:55-65); get_my_FD_loss evaluates CrossEntropyLoss of a trained model under no_grad (:76-111).
// One call of ddg_fd_train_epoch = one pass of the reference's inner loop over the DataLoader: the caller passes
// the epoch's row order (its shuffle), the library walks it in batches: gather, forward, loss gradient,
// backward, SGD update -- about a dozen short launches per batch on the fp32 MFMA.
// ---------------------------------------------------------------------------------------------
__global__ void k_fd_gather(const float *__restrict__ X, const int32_t *__restrict__ Y, const int64_t *__restrict__ perm,
                            int rows, int in, float *__restrict__ Xb, int32_t *__restrict__ Yb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * in) return;
    const int r = e / in, k = e - r * in;
    const int64_t src = perm[r];
    Xb[e] = X[src * in + k];
    if (k == 0) Yb[r] = Y[src];
}
__global__ void k_fd_relu(const float *__restrict__ a, int n, float *__restrict__ h) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) h[e] = fmaxf(a[e], 0.f);
}
// softmax / cross-entropy of one row per thread.  d_out (optional): (softmax - onehot) / rows_norm.
// acc[0] += sum of -log softmax[y]; hits[0] += rows whose first arg-max equals y (torch.max tie rule).
__global__ void k_fd_ce(const float *__restrict__ o, const int32_t *__restrict__ y, int rows, int out, float rows_norm,
                        float *__restrict__ d_out, float *__restrict__ acc, int32_t *__restrict__ hits) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    int hit = 0;
    if (r < rows) {
        const float *p = o + (int64_t)r * out;
        float mx = p[0];
        int arg = 0;
        for (int j = 1; j < out; ++j)
            if (p[j] > mx) { mx = p[j]; arg = j; }
        float s = 0.f;
        for (int j = 0; j < out; ++j) s += expf(p[j] - mx);
        const int yy = y[r];
        l = -(p[yy] - mx - logf(s));
        hit = arg == yy;
        if (d_out)
            for (int j = 0; j < out; ++j)
                d_out[(int64_t)r * out + j] = (expf(p[j] - mx) / s - (j == yy ? 1.f : 0.f)) / rows_norm;
    }
    // wave reduction, one atomic per wave
    for (int of = 32; of > 0; of >>= 1) {
        l += __shfl_xor(l, of);
        hit += __shfl_xor(hit, of);
    }
    if ((threadIdx.x & 63) == 0) {
        if (acc) atomicAdd(acc, l);
        if (hits) atomicAdd(hits, hit);
    }
}
__global__ void k_fd_sgd(float *__restrict__ p, const float *__restrict__ g, int n, float lr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = p[e] - lr * g[e];
}

__global__ void k_fd_mean(float *__restrict__ acc, float n) { acc[0] = acc[0] / n; }

struct FdWs {
    float *Xb, *a1, *h, *o, *dO, *dh, *da, *dW1, *db1, *dW3, *db3, *acc;
    int32_t *Yb, *hits;
    size_t total;
};
static void fd_carve(FdWs &w, char *base, int64_t B, int in, int out) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += tal(bytes); return p; };
    auto f = [&](size_t n) { return (float *)take(n * 4); };
    w.Xb = f(B * in); w.a1 = f(B * in); w.h = f(B * in); w.o = f(B * out); w.dO = f(B * out); w.dh = f(B * in);
    w.da = f(B * in); w.dW1 = f((size_t)in * in); w.db1 = f(in); w.dW3 = f((size_t)out * in); w.db3 = f(out);
    w.acc = f(1);
    w.Yb = (int32_t *)take(B * 4); w.hits = (int32_t *)take(4);
    w.total = o;
}
extern "C" size_t ddg_fd_workspace_bytes(int64_t batch_rows, int in_dim, int out_dim) {
    if (batch_rows < 1 || in_dim < 1 || out_dim < 1) return 0;
    FdWs w;
    fd_carve(w, nullptr, batch_rows, in_dim, out_dim);
    return w.total;
}
static int fd_check(const ddg_fd_weights *w) {
    if (!w || !w->W1 || !w->b1 || !w->W3 || !w->b3 || w->in_dim < 1 || w->out_dim < 1) return DDG_E_ARG;
    return DDG_OK;
}
// forward of `rows` rows already in X (row-major [rows,in]): a1, h, o in the workspace
static int fd_forward(const ddg_fd_weights *w, const float *X, int rows, FdWs &ws, void *stream) {
    const int in = w->in_dim, out = w->out_dim;
    int rc = ddg_linear(X, nullptr, in, w->W1, w->b1, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                        0, 0, ws.a1, nullptr, rows, in, in, stream);
    if (rc) return rc;
    k_fd_relu<<<g256((int64_t)rows * in), 256, 0, (hipStream_t)stream>>>(ws.a1, rows * in, ws.h);
    return ddg_linear(ws.h, nullptr, in, w->W3, w->b3, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f,
                      0, 0, ws.o, nullptr, rows, in, out, stream);
}

extern "C" int ddg_fd_train_epoch(const ddg_fd_weights *w, const float *X, const int32_t *Y, const int64_t *perm,
                                  int64_t n, int batch, float lr, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    int rc = fd_check(w);
    if (rc) return rc;
    if (!X || !Y || !perm || n < 0 || batch < 1 || batch > 65536) return DDG_E_ARG;
    const int in = w->in_dim, out = w->out_dim;
    FdWs ws;
    fd_carve(ws, (char *)workspace, batch, in, out);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    for (int64_t r0 = 0; r0 < n; r0 += batch) {
        const int B = (int)(n - r0 < batch ? n - r0 : batch);           // the DataLoader's last batch is ragged
        k_fd_gather<<<g256((int64_t)B * in), 256, 0, s>>>(X, Y, perm + r0, B, in, ws.Xb, ws.Yb);
        if ((rc = fd_forward(w, ws.Xb, B, ws, stream))) return rc;
        k_fd_ce<<<g256(B), 256, 0, s>>>(ws.o, ws.Yb, B, out, (float)B, ws.dO, nullptr, nullptr);
        if ((rc = gemm_dW(ws.dO, ws.h, ws.dW3, B, in, out, s))) return rc;
        k_colsum<<<gcol32(out), 256, 0, s>>>(ws.dO, B, out, ws.db3);
        if ((rc = gemm_dA(ws.dO, w->W3, ws.dh, B, in, out, s))) return rc;
        k_relu_bwd<<<g256((int64_t)B * in), 256, 0, s>>>(ws.dh, ws.h, B * in, ws.da);
        if ((rc = gemm_dW(ws.da, ws.Xb, ws.dW1, B, in, in, s))) return rc;
        k_colsum<<<gcol32(in), 256, 0, s>>>(ws.da, B, in, ws.db1);
        k_fd_sgd<<<g256((int64_t)in * in), 256, 0, s>>>(w->W1, ws.dW1, in * in, lr);
        k_fd_sgd<<<g256(in), 256, 0, s>>>(w->b1, ws.db1, in, lr);
        k_fd_sgd<<<g256((int64_t)out * in), 256, 0, s>>>(w->W3, ws.dW3, out * in, lr);
        k_fd_sgd<<<g256(out), 256, 0, s>>>(w->b3, ws.db3, out, lr);
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_fd_eval(const ddg_fd_weights *w, const float *X, const int32_t *Y, int64_t n, int32_t *out_correct,
                           float *out_ce_mean, void *workspace, size_t workspace_bytes, int64_t workspace_rows,
                           void *stream) {
    int rc = fd_check(w);
    if (rc) return rc;
    if (!X || !Y || n < 1 || workspace_rows < 1 || (!out_correct && !out_ce_mean)) return DDG_E_ARG;
    const int in = w->in_dim, out = w->out_dim;
    FdWs ws;
    fd_carve(ws, (char *)workspace, workspace_rows, in, out);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws.acc, 0, 4, s) != hipSuccess || hipMemsetAsync(ws.hits, 0, 4, s) != hipSuccess) return DDG_E_LAUNCH;
    for (int64_t r0 = 0; r0 < n; r0 += workspace_rows) {
        const int B = (int)(n - r0 < workspace_rows ? n - r0 : workspace_rows);
        if ((rc = fd_forward(w, X + r0 * in, B, ws, stream))) return rc;
        k_fd_ce<<<g256(B), 256, 0, s>>>(ws.o, Y + r0, B, out, 1.f, nullptr, ws.acc, ws.hits);
    }
    if (out_correct && hipMemcpyAsync(out_correct, ws.hits, 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return DDG_E_LAUNCH;
    if (out_ce_mean) {
        k_fd_mean<<<1, 1, 0, s>>>(ws.acc, (float)n);
        if (hipMemcpyAsync(out_ce_mean, ws.acc, 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return DDG_E_LAUNCH;
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// this translation unit's copy of the stream-offset pointer (ddg_set_stream_offset, elementwise.hip)
int ddg_tu_set_stream_off_train(const uint64_t *dev_ptr) { return set_stream_off_tu(dev_ptr); }

// ---------------------------------------------------------------------------------------------
// (f1, rest) the plain discriminator D that get_valid_data_index trains when fewer than 20 % of the rows are complete
//                                                                    utils/util.py:623-646, model/Discriminator_model.py:5-39
//   p = sigmoid(fc3(drop(relu(fc2(drop(relu(fc1(x))))))))     train mode: Dropout(0.2) live, also in the final scoring pass
//   loss = -mean(m log(p + 1e-8) + (1 - m) log(1 - p + 1e-8))  over all N x C cells, 1 000 FULL-BATCH Adam steps (lr 0.002)
// Full batch means N rows per step (10M at BASELINE config 4, where no row is complete): the three forward layers go
// through ddg_linear (the wide f16-split kernel from 8 192 rows on), the input gradients likewise on transposed weights,
// and the weight gradients dW = dZ^T H -- a reduction over ALL rows -- through k_gemm_tn below: row chunks x 32 x 32
// output tiles on the fp32 MFMA, partial tiles reduced in a fixed order (deterministic).  The bias gradient is the extra
// column of dZ^T [H | 1].  Dropout keeps come from explicit masks (parity: the caller draws them from torch's CPU generator
// in the reference's order) or from Philox on chip.
// ---------------------------------------------------------------------------------------------
#define TN_ROWS 4096        // rows per workgroup of k_gemm_tn (1 024 per wave)
// part[chunk][o][i] = sum_{r in chunk} dZ[r][o] * Hx[r][i],  Hx = [H | 1]  (i == in: the ones column)
__global__ __launch_bounds__(256) void k_gemm_tn(const float *__restrict__ dZ, int out, const float *__restrict__ H, int in,
                                                 int64_t rows, float *__restrict__ part) {
    __shared__ float red[4][32 * 33];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const int ti = blockIdx.x, to = blockIdx.y;
    const int64_t chunk = blockIdx.z;
    const int o = to * 32 + r32, i = ti * 32 + r32;
    const bool ook = o < out, iok = i <= in;
    const int64_t r0 = chunk * TN_ROWS + (int64_t)wid * (TN_ROWS / 4);
    int64_t r1 = r0 + TN_ROWS / 4;
    if (r1 > rows) r1 = rows;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    for (int64_t r = r0; r < r1; r += 16) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t rr = r + 2 * u + h;
            const bool rok = rr < r1;
            a[u] = (rok && ook) ? dZ[rr * out + o] : 0.f;
            b[u] = (rok && iok) ? (i == in ? 1.f : H[rr * in + i]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) red[wid][((g & 3) + 8 * (g >> 2) + 4 * h) * 33 + r32] = acc[g];
    __syncthreads();
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float *dst = part + chunk * (int64_t)out * (in + 1);
    if (ti * 32 + c <= in) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rl = rg * 4 + k;
            if (to * 32 + rl < out)
                dst[(int64_t)(to * 32 + rl) * (in + 1) + ti * 32 + c] =
                    (red[0][rl * 33 + c] + red[1][rl * 33 + c]) + (red[2][rl * 33 + c] + red[3][rl * 33 + c]);
        }
    }
}
// dW[o][i] (i < in) and db[o] (i == in) = sum over the chunks, ascending
__global__ void k_tn_reduce(const float *__restrict__ part, int64_t nchunks, int out, int in, float *__restrict__ dW,
                            float *__restrict__ db) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= out * (in + 1)) return;
    float s = 0.f;
    for (int64_t c = 0; c < nchunks; ++c) s += part[c * (int64_t)out * (in + 1) + e];
    const int o = e / (in + 1), i = e - o * (in + 1);
    if (i == in) db[o] = s; else dW[(int64_t)o * in + i] = s;
}
__global__ void k_transpose(const float *__restrict__ W, int rows, int cols, float *__restrict__ Wt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * cols) return;
    const int r = e / cols, c = e - r * cols;
    Wt[(int64_t)c * rows + r] = W[e];
}
// h = relu(z) * keep / (1 - p), in place
__global__ void k_relu_drop_fwd(float *__restrict__ z, const uint8_t *__restrict__ drop, float drop_p, uint64_t seed,
                                uint64_t stream_id, int64_t n) {
    const float ks = 1.f / (1.f - drop_p);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float v = fmaxf(z[e], 0.f);
        if (drop) v = drop[e] ? v * ks : 0.f;
        else if (drop_p > 0.f) v = (philox_uniform4(seed, stream_id, (uint64_t)e >> 2)[e & 3] > drop_p) ? v * ks : 0.f;
        z[e] = v;
    }
}
// dz = dh * (h > 0 ? 1 / (1 - p) : 0): a kept unit with z > 0 has h = z / (1 - p) > 0, every other unit has h == 0
__global__ void k_relu_drop_bwd(const float *__restrict__ dh, const float *__restrict__ hact, float drop_p,
                                float *__restrict__ dz, int64_t n) {
    const float ks = 1.f / (1.f - drop_p);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        dz[e] = hact[e] > 0.f ? dh[e] * ks : 0.f;
}
// z3 -> d(loss)/d(z3) in place (p = sigmoid(z3); torch's backward of the expression at utils/util.py:636), or p itself
__global__ void k_dplain_out(float *__restrict__ z3, const float *__restrict__ m, int64_t n, int want_grad) {
    const float inv = 1.f / (float)n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float p = 1.f / (1.f + expf(-z3[e]));
        if (!want_grad) { z3[e] = p; continue; }
        const float mm = m[e];
        const float dp = -(mm / (p + 1e-8f) - (1.f - mm) / (1.f - p + 1e-8f)) * inv;
        z3[e] = dp * p * (1.f - p);
    }
}

struct DplainWs {
    float *h1, *h2, *z3, *dh, *dz, *W2t, *W3t, *zero, *part;
    int64_t nchunks;
    size_t total;
};
static void dplain_carve(DplainWs &w, char *base, int64_t N, int D, int C) {
    size_t o = 0;
    auto f = [&](size_t n) { char *p = base ? base + o : nullptr; o += tal(n * 4); return (float *)p; };
    w.h1 = f((size_t)N * D); w.h2 = f((size_t)N * D); w.z3 = f((size_t)N * C);
    w.dh = f((size_t)N * D); w.dz = f((size_t)N * D);
    w.W2t = f((size_t)D * D); w.W3t = f((size_t)D * C); w.zero = f((size_t)D);
    w.nchunks = ceil_div64(N, TN_ROWS);
    w.part = f((size_t)w.nchunks * D * (D + 1));
    w.total = o;
}
extern "C" size_t ddg_dplain_workspace_bytes(int64_t rows, int in_dim, int out_dim) {
    if (rows < 1 || in_dim < 1 || out_dim < 1 || out_dim > in_dim) return 0;
    DplainWs w;
    dplain_carve(w, nullptr, rows, in_dim, out_dim);
    return w.total;
}
static inline unsigned gstride(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g > 65536 ? 65536 : (g < 1 ? 1 : g)); }

static int dplain_forward(const ddg_dplain_weights *w, const float *x, const uint8_t *drop1, const uint8_t *drop2,
                          float drop_p, uint64_t seed, uint64_t stream_id, int64_t N, DplainWs &ws, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const int D = w->in_dim, C = w->out_dim;
    int rc = ddg_linear(x, nullptr, D, w->W1, w->b1, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0, 0,
                        ws.h1, nullptr, N, D, D, stream);
    if (rc) return rc;
    k_relu_drop_fwd<<<gstride(N * D), 256, 0, s>>>(ws.h1, drop1, drop_p, seed, stream_id * 2, N * D);
    rc = ddg_linear(ws.h1, nullptr, D, w->W2, w->b2, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0, 0,
                    ws.h2, nullptr, N, D, D, stream);
    if (rc) return rc;
    k_relu_drop_fwd<<<gstride(N * D), 256, 0, s>>>(ws.h2, drop2, drop_p, seed, stream_id * 2 + 1, N * D);
    rc = ddg_linear(ws.h2, nullptr, D, w->W3, w->b3, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0, 0,
                    ws.z3, nullptr, N, D, C, stream);
    if (rc) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

static int gemm_tn(const float *dZ, int out, const float *H, int in, int64_t N, DplainWs &ws, float *dW, float *db, hipStream_t s) {
    dim3 grid((unsigned)((in + 1 + 31) / 32), (unsigned)((out + 31) / 32), (unsigned)ws.nchunks);
    k_gemm_tn<<<grid, 256, 0, s>>>(dZ, out, H, in, N, ws.part);
    k_tn_reduce<<<(unsigned)((out * (in + 1) + 255) / 256), 256, 0, s>>>(ws.part, ws.nchunks, out, in, dW, db);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// one step up to the gradients (zero_grad + forward + loss.backward of utils/util.py:633-637)
extern "C" int ddg_dplain_grads(const ddg_dplain_weights *w, const float *x, const float *m, const uint8_t *drop1,
                                const uint8_t *drop2, float drop_p, uint64_t seed, uint64_t stream_id, int64_t rows,
                                const ddg_dplain_weights *grads, void *workspace, size_t workspace_bytes, void *stream) {
    if (!w || !x || !m || !grads || rows < 1 || w->in_dim < 1 || w->out_dim < 1 || w->out_dim > w->in_dim) return DDG_E_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (ceil_div64(rows, TN_ROWS) > 65535)) return DDG_E_ARG;
    const int D = w->in_dim, C = w->out_dim;
    DplainWs ws;
    dplain_carve(ws, (char *)workspace, rows, D, C);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int rc = dplain_forward(w, x, drop1, drop2, drop_p, seed, stream_id, rows, ws, stream);
    if (rc) return rc;
    k_dplain_out<<<gstride(rows * C), 256, 0, s>>>(ws.z3, m, rows * C, 1);                       // z3 <- dz3
    if ((rc = gemm_tn(ws.z3, C, ws.h2, D, rows, ws, grads->W3, grads->b3, s))) return rc;
    k_transpose<<<g256((int64_t)C * D), 256, 0, s>>>(w->W3, C, D, ws.W3t);                       // W3 [C][D] -> [D][C]
    // dh2 = dz3 W3: a layer with weight W3^T and a zero bias
    if ((rc = zero_async(ws.zero, tal((size_t)D * 4), s))) return rc;
    rc = ddg_linear(ws.z3, nullptr, C, ws.W3t, ws.zero, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0, 0,
                    ws.dh, nullptr, rows, C, D, stream);
    if (rc) return rc;
    k_relu_drop_bwd<<<gstride(rows * D), 256, 0, s>>>(ws.dh, ws.h2, drop_p, ws.dz, rows * D);
    k_transpose<<<g256((int64_t)D * D), 256, 0, s>>>(w->W2, D, D, ws.W2t);
    rc = ddg_linear(ws.dz, nullptr, D, ws.W2t, ws.zero, DDG_PRO_NONE, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.f, 0, 0,
                    ws.dh, nullptr, rows, D, D, stream);                                          // dh1 = dz2 W2
    if (rc) return rc;
    if ((rc = gemm_tn(ws.dz, D, ws.h1, D, rows, ws, grads->W2, grads->b2, s))) return rc;       // dW2 = dz2^T h1
    k_relu_drop_bwd<<<gstride(rows * D), 256, 0, s>>>(ws.dh, ws.h1, drop_p, ws.dz, rows * D);    // dz1
    if ((rc = gemm_tn(ws.dz, D, x, D, rows, ws, grads->W1, grads->b1, s))) return rc;           // dW1 = dz1^T x
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// the scoring pass (utils/util.py:638: D_pro = curr_dis(impute_data_code), still in train mode): p_out [rows][out_dim]
extern "C" int ddg_dplain_fwd(const ddg_dplain_weights *w, const float *x, const uint8_t *drop1, const uint8_t *drop2,
                              float drop_p, uint64_t seed, uint64_t stream_id, int64_t rows, float *p_out, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!w || !x || !p_out || rows < 1 || w->in_dim < 1 || w->out_dim < 1 || w->out_dim > w->in_dim) return DDG_E_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return DDG_E_ARG;
    DplainWs ws;
    dplain_carve(ws, (char *)workspace, rows, w->in_dim, w->out_dim);
    if (!workspace || workspace_bytes < ws.total) return DDG_E_WORKSPACE;
    int rc = dplain_forward(w, x, drop1, drop2, drop_p, seed, stream_id, rows, ws, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    k_dplain_out<<<gstride(rows * w->out_dim), 256, 0, s>>>(ws.z3, nullptr, rows * w->out_dim, 0);
    if ((rc = copy_f32_async(p_out, ws.z3, (size_t)rows * w->out_dim, s))) return rc;
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}


// ---------------------------------------------------------------------------------------------
// The minibatch gather of a training step (sample_x's x[rows, :], utils/util.py:582-585, applied to the encoded table, its
// expanded mask and the [N, C] mask) in ONE launch, and the book-keeping of a step that is replayed as a captured graph:
// counter -> offset before the step's first draw, counter += 2^32 and offset = 0 after its last (ddg_set_stream_offset).
// Row indices are clamped into the table (torch would raise).
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_batch(const float *__restrict__ x, const float *__restrict__ M, const float *__restrict__ m,
                               const int64_t *__restrict__ idx, int64_t n_rows, int B, int D, int C,
                               float *__restrict__ x0, float *__restrict__ Mb, float *__restrict__ mb,
                               const int64_t *__restrict__ counter, int64_t *__restrict__ offset) {
    if (offset && blockIdx.x == 0 && threadIdx.x == 0) offset[0] = counter[0];
    const int per = 2 * D + C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)B * per; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / per), j = (int)(e - (int64_t)r * per);
        int64_t row = idx[r];
        row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
        if (j < D) x0[(int64_t)r * D + j] = x[row * D + j];
        else if (j < 2 * D) Mb[(int64_t)r * D + j - D] = M[row * D + j - D];
        else mb[(int64_t)r * C + j - 2 * D] = m[row * C + j - 2 * D];
    }
}
__global__ void k_step_end(int64_t *__restrict__ counter, int64_t *__restrict__ offset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { counter[0] += (int64_t)1 << 32; offset[0] = 0; }
}

extern "C" int ddg_gather_batch(const float *x, const float *M, const float *m, const int64_t *idx, int64_t n_rows, int B,
                                int D, int C, float *x0, float *Mb, float *mb, const int64_t *counter, int64_t *offset,
                                void *stream) {
    if (!x || !M || !m || !idx || !x0 || !Mb || !mb || n_rows < 1 || B < 1 || D < 1 || C < 1) return DDG_E_ARG;
    if ((counter == nullptr) != (offset == nullptr)) return DDG_E_ARG;
    const int64_t n = (int64_t)B * (2 * D + C);
    k_gather_batch<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, M, m, idx, n_rows, B, D, C, x0, Mb, mb,
                                                                               counter, offset);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_step_end(int64_t *counter, int64_t *offset, void *stream) {
    if (!counter || !offset) return DDG_E_ARG;
    k_step_end<<<1, 64, 0, (hipStream_t)stream>>>(counter, offset);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
