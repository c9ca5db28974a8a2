// the "data-utility" Learner (model/Learner.py:10-67) as one fused pass per SGD step.
//
// The reference trains p = softmax(x W^T + b) for `epoch` full-batch SGD steps (lr 0.01) under
// CrossEntropyLoss(p, y), which applies log_softmax to p once more (:36-39, :47), then evaluates loss and
// accuracy on the validation rows (:61-66).  Per step and row:
//     q   = softmax(p)                      g_p = (q - onehot(y)) / n
//     g_z = p * (g_p - sum_j g_p[j] p[j])   dW += g_z x^T,  db += g_z
// HBM-bound: the only large operand is x [n, D]; each step reads it ONCE (torch's op sequence reads it for the
// forward GEMM and again for the weight-gradient GEMM, and round-trips p, q and the gradients through HBM).
// k_learner_step: persistent workgroups stride over tiles of R rows staged in LDS; logits, both softmaxes and g_z
//   stay in LDS; every thread owns weight-gradient columns and keeps their L partial sums in registers over
//   all its tiles; one partial [L][D+1] per workgroup goes to the workspace (no atomics).
// k_learner_apply: sums the partials in a fixed order (fp64) and applies the SGD update: bitwise reproducible.
// k_learner_eval / k_learner_eval_finish: validation loss and number of correct arg-max predictions.
#include <atomic>
#include "common.h"

#define LRN_NT 256
#define LRN_BLOCKS 1536               // persistent workgroups of the step kernel (6 per CU)
#define LRN_LDS_TARGET (26 * 1024)    // LDS per workgroup aimed at, so that 6 workgroups share a CU (latency hiding)

struct LrnArgs {
    const float *x;
    const int32_t *y;
    int64_t n;
    int D, L, R;                      // R = rows per LDS tile
    const float *W, *b;               // [L][D], [L]
    float *partial;                   // [gridDim.x][L][D + 1]   (column D = bias)
    float inv_n;
};

// LDS carve shared by the step and the eval kernel
struct LrnLds {
    float *w;     // [L][D + 1]  weights, column D = bias; row stride D + 1 (+1 more when even: odd stride)
    float *x;     // [R][D]
    float *z;     // [R][L]  logits -> g_z (step) / p (eval)
    int ws;       // weight row stride
};
__device__ __forceinline__ LrnLds lrn_carve(float *base, int D, int L, int R) {
    LrnLds s;
    s.ws = (D + 1) | 1;
    s.w = base;
    s.x = s.w + (((size_t)L * s.ws + 3) & ~(size_t)3);
    s.z = s.x + (size_t)R * D;
    return s;
}
static inline size_t lrn_lds_bytes(int D, int L, int R) {
    return ((((size_t)L * ((D + 1) | 1) + 3) & ~(size_t)3) + (size_t)R * D + (size_t)R * L) * 4;
}

__device__ __forceinline__ void lrn_load_weights(const LrnLds &s, const float *W, const float *b, int D, int L) {
    for (int i = threadIdx.x; i < L * (D + 1); i += LRN_NT) {
        const int j = i / (D + 1), d = i - j * (D + 1);
        s.w[j * s.ws + d] = d < D ? W[(int64_t)j * D + d] : b[j];
    }
}
// rows [row0, row0 + nr) of x -> LDS tile (coalesced; rows are contiguous in memory)
__device__ __forceinline__ void lrn_load_tile(const LrnLds &s, const float *x, int64_t row0, int nr, int D) {
    const float *src = x + row0 * D;
    const int tot = nr * D;
    if (((row0 * D) & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)s.x & 15) == 0) {    // 16-byte path
        const int tot4 = tot >> 2;
        for (int i = threadIdx.x; i < tot4; i += LRN_NT)
            reinterpret_cast<f32x4 *>(s.x)[i] = reinterpret_cast<const f32x4 *>(src)[i];
        for (int i = (tot4 << 2) + threadIdx.x; i < tot; i += LRN_NT) s.x[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < tot; i += LRN_NT) s.x[i] = src[i];
    }
}
// logits of the tile: 8 threads per row, each over an eighth of the columns for ALL classes (x is read once, the
// weight reads are broadcasts), then a 3-step butterfly over the 8 partners; rows in passes of LRN_NT / 8
template <int LCAP>
__device__ __forceinline__ void lrn_logits(const LrnLds &s, int nr, int D, int L) {
    const int q = threadIdx.x & 7;
    const int dq = (D + 7) >> 3;
    const int d0 = q * dq, d1 = (d0 + dq < D) ? d0 + dq : D;
    for (int r0 = 0; r0 < nr; r0 += LRN_NT / 8) {
        const int r = r0 + (threadIdx.x >> 3);
        float a[LCAP];
#pragma unroll
        for (int j = 0; j < LCAP; ++j) a[j] = 0.f;
        if (r < nr) {
            const float *xr = s.x + r * D;
            for (int d = d0; d < d1; ++d) {
                const float xv = xr[d];
#pragma unroll
                for (int j = 0; j < LCAP; ++j)
                    if (j < L) a[j] = fmaf(xv, s.w[j * s.ws + d], a[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < LCAP; ++j)
            if (j < L) {
                float v = a[j];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                if (q == 0 && r < nr) s.z[r * L + j] = v + s.w[j * s.ws + D];
            }
    }
}

template <int LCAP, int DM>
__global__ __launch_bounds__(LRN_NT) void k_learner_step(LrnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const LrnLds s = lrn_carve(lsm, p.D, p.L, p.R);
    const int D = p.D, L = p.L, R = p.R;
    lrn_load_weights(s, p.W, p.b, D, L);
    float acc[DM][LCAP];
#pragma unroll
    for (int m = 0; m < DM; ++m)
#pragma unroll
        for (int j = 0; j < LCAP; ++j) acc[m][j] = 0.f;
    const int64_t ntiles = (p.n + R - 1) / R;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * R;
        const int nr = (int)(p.n - row0 < R ? p.n - row0 : R);
        __syncthreads();                       // previous tile fully consumed (and the weights are in place)
        lrn_load_tile(s, p.x, row0, nr, D);
        __syncthreads();
        lrn_logits<LCAP>(s, nr, D, L);
        __syncthreads();
        if ((int)threadIdx.x < nr) {           // one thread per row: both softmaxes and the logit gradient
            float *z = s.z + threadIdx.x * L;
            const int yy = p.y[row0 + threadIdx.x];
            float mz = z[0];
            for (int j = 1; j < L; ++j) mz = fmaxf(mz, z[j]);
            float sz = 0.f;
            for (int j = 0; j < L; ++j) { z[j] = expf(z[j] - mz); sz += z[j]; }
            float sq = 0.f;
            for (int j = 0; j < L; ++j) { z[j] = z[j] / sz; sq += expf(z[j]); }   // p in [0,1]: exp cannot overflow
            float dot = 0.f;
            for (int j = 0; j < L; ++j) {
                const float gp = (expf(z[j]) / sq - (j == yy ? 1.f : 0.f)) * p.inv_n;
                dot = fmaf(gp, z[j], dot);
            }
            for (int j = 0; j < L; ++j) {
                const float pj = z[j];
                const float gp = (expf(pj) / sq - (j == yy ? 1.f : 0.f)) * p.inv_n;
                z[j] = pj * (gp - dot);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < DM; ++m) {         // weight-gradient columns d = tid + 256 m (d == D: the bias)
            const int d = threadIdx.x + LRN_NT * m;
            if (d <= D) {
                for (int r = 0; r < nr; ++r) {
                    const float xv = d < D ? s.x[r * D + d] : 1.f;
                    const float *g = s.z + r * L;
#pragma unroll
                    for (int j = 0; j < LCAP; ++j)
                        if (j < L) acc[m][j] = fmaf(g[j], xv, acc[m][j]);
                }
            }
        }
    }
    float *out = p.partial + (int64_t)blockIdx.x * L * (D + 1);
#pragma unroll
    for (int m = 0; m < DM; ++m) {
        const int d = threadIdx.x + LRN_NT * m;
        if (d <= D) {
#pragma unroll
            for (int j = 0; j < LCAP; ++j)
                if (j < L) out[j * (D + 1) + d] = acc[m][j];
        }
    }
}

// W[j][d] -= lr * sum_blocks partial[block][j][d].  Workgroup = 16 outputs x 16 segments of the block range; every
// segment is summed in order (fp64), the 16 segment sums are added in order: bitwise reproducible.
__global__ __launch_bounds__(256) void k_learner_apply(const float *__restrict__ partial, int nblocks, int D, int L,
                                                       float lr, float *__restrict__ W, float *__restrict__ b) {
    __shared__ double seg[16][17];
    const int o = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const int tot = L * (D + 1);
    const int per = (nblocks + 15) / 16;
    const int g0 = sg * per, g1 = g0 + per < nblocks ? g0 + per : nblocks;
    double a = 0.0;
    if (i < tot)
        for (int g = g0; g < g1; ++g) a += (double)partial[(int64_t)g * tot + i];
    seg[sg][o] = a;
    __syncthreads();
    if (sg == 0 && i < tot) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += seg[k][o];
        const int j = i / (D + 1), d = i - j * (D + 1);
        const float upd = lr * (float)t;
        if (d < D) W[(int64_t)j * D + d] -= upd; else b[j] -= upd;
    }
}

// validation: per workgroup sum of (logsumexp(p) - p[y]) and count of argmax(p) == y
template <int LCAP>
__global__ __launch_bounds__(LRN_NT) void k_learner_eval(LrnArgs p, double *__restrict__ loss_part,
                                                         int32_t *__restrict__ hit_part) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    __shared__ double red_l[LRN_NT];
    __shared__ int red_h[LRN_NT];
    const LrnLds s = lrn_carve(lsm, p.D, p.L, p.R);
    const int D = p.D, L = p.L, R = p.R;
    lrn_load_weights(s, p.W, p.b, D, L);
    double lsum = 0.0;
    int hits = 0;
    const int64_t ntiles = (p.n + R - 1) / R;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * R;
        const int nr = (int)(p.n - row0 < R ? p.n - row0 : R);
        __syncthreads();
        lrn_load_tile(s, p.x, row0, nr, D);
        __syncthreads();
        lrn_logits<LCAP>(s, nr, D, L);
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            const float *z = s.z + threadIdx.x * L;
            const int yy = p.y[row0 + threadIdx.x];
            float mz = z[0];
            int am = 0;
            for (int j = 1; j < L; ++j)
                if (z[j] > mz) { mz = z[j]; am = j; }        // first maximum, as torch.max
            float sz = 0.f;
            for (int j = 0; j < L; ++j) sz += expf(z[j] - mz);
            float sq = 0.f, py = 0.f;
            for (int j = 0; j < L; ++j) {
                const float pj = expf(z[j] - mz) / sz;
                sq += expf(pj);
                if (j == yy) py = pj;
            }
            lsum += (double)(logf(sq) - py);
            hits += am == yy;
        }
    }
    red_l[threadIdx.x] = lsum;
    red_h[threadIdx.x] = hits;
    __syncthreads();
    for (int o = LRN_NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red_l[threadIdx.x] += red_l[threadIdx.x + o]; red_h[threadIdx.x] += red_h[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss_part[blockIdx.x] = red_l[0]; hit_part[blockIdx.x] = red_h[0]; }
}

__global__ void k_learner_eval_finish(const double *__restrict__ loss_part, const int32_t *__restrict__ hit_part,
                                      int nblocks, int64_t n, float *__restrict__ out_loss,
                                      int32_t *__restrict__ out_correct) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0;
    int h = 0;
    for (int g = 0; g < nblocks; ++g) { a += loss_part[g]; h += hit_part[g]; }
    out_loss[0] = (float)(a / (double)n);
    out_correct[0] = h;
}

// ---------------------------------------------------------------------------------------------
static inline size_t lrn_al(size_t x) { return (x + 255) & ~(size_t)255; }
static int lrn_rows_per_tile(int D, int L) {
    // the largest power of two that keeps the workgroup near LRN_LDS_TARGET; wide tables fall back to what fits at all
    for (int R = LRN_NT; R >= 8; R >>= 1)
        if (lrn_lds_bytes(D, L, R) <= LRN_LDS_TARGET) return R;
    for (int R = 64; R >= 8; R >>= 1)
        if (lrn_lds_bytes(D, L, R) <= 150 * 1024) return R;
    return 0;
}
static int lrn_check(int64_t n_train, int64_t n_valid, int D, int L) {
    if (n_train < 1 || n_valid < 1 || D < 1 || L < 1) return DDG_E_ARG;
    if (!((L <= 16 && D <= 4 * LRN_NT - 1) || (L <= 64 && D <= LRN_NT - 1))) return DDG_E_UNSUPPORTED;
    if (lrn_rows_per_tile(D, L) == 0) return DDG_E_UNSUPPORTED;
    return DDG_OK;
}

extern "C" size_t ddg_learner_workspace_bytes(int D, int L) {
    return lrn_al((size_t)LRN_BLOCKS * L * (D + 1) * 4) + lrn_al((size_t)LRN_BLOCKS * 8) + lrn_al((size_t)LRN_BLOCKS * 4);
}

template <int LCAP, int DM>
static int lrn_launch_step(const LrnArgs &a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<size_t> granted{48 * 1024};      // per instantiation; atomic: callers may use several host threads
    if (lds > granted.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)k_learner_step<LCAP, DM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return DDG_E_LAUNCH;
        granted.store(lds, std::memory_order_release);
    }
    k_learner_step<LCAP, DM><<<grid, LRN_NT, lds, s>>>(a);
    return DDG_OK;
}

extern "C" int ddg_learner_train(const float *x_train, const int32_t *y_train, int64_t n_train, const float *x_valid,
                                 const int32_t *y_valid, int64_t n_valid, int D, int L, int epochs, float lr,
                                 float *W, float *b, float *out_val_loss, int32_t *out_val_correct,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    if (!x_train || !y_train || !x_valid || !y_valid || !W || !b || !out_val_loss || !out_val_correct || epochs < 0)
        return DDG_E_ARG;
    int rc = lrn_check(n_train, n_valid, D, L);
    if (rc) return rc;
    if (!workspace || workspace_bytes < ddg_learner_workspace_bytes(D, L)) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *wp = (char *)workspace;
    float *partial = (float *)wp;
    wp += lrn_al((size_t)LRN_BLOCKS * L * (D + 1) * 4);
    double *loss_part = (double *)wp;
    wp += lrn_al((size_t)LRN_BLOCKS * 8);
    int32_t *hit_part = (int32_t *)wp;

    const int R = lrn_rows_per_tile(D, L);
    const size_t lds = lrn_lds_bytes(D, L, R);
    LrnArgs a;
    a.x = x_train; a.y = y_train; a.n = n_train; a.D = D; a.L = L; a.R = R; a.W = W; a.b = b; a.partial = partial;
    a.inv_n = 1.0f / (float)n_train;
    const int64_t ntiles = (n_train + R - 1) / R;
    const int grid = (int)(ntiles < LRN_BLOCKS ? ntiles : LRN_BLOCKS);
    const int DMn = (D + 1 + LRN_NT - 1) / LRN_NT;
    const int napply = (L * (D + 1) + 15) / 16;
    for (int e = 0; e < epochs; ++e) {
#define LRN_CASE(LC, DMC) rc = lrn_launch_step<LC, DMC>(a, grid, lds, s)
        const int dm = DMn <= 1 ? 1 : (DMn == 2 ? 2 : 4);
        if (L > 16) LRN_CASE(64, 1);
        else if (L <= 4) { if (dm == 1) LRN_CASE(4, 1); else if (dm == 2) LRN_CASE(4, 2); else LRN_CASE(4, 4); }
        else if (L <= 8) { if (dm == 1) LRN_CASE(8, 1); else if (dm == 2) LRN_CASE(8, 2); else LRN_CASE(8, 4); }
        else { if (dm == 1) LRN_CASE(16, 1); else if (dm == 2) LRN_CASE(16, 2); else LRN_CASE(16, 4); }
#undef LRN_CASE
        if (rc) return rc;
        k_learner_apply<<<napply, 256, 0, s>>>(partial, grid, D, L, lr, W, b);
    }
    DDG_CHECK_LAUNCH();

    LrnArgs v = a;
    v.x = x_valid; v.y = y_valid; v.n = n_valid; v.partial = nullptr;
    const int64_t vt = (n_valid + R - 1) / R;
    const int vgrid = (int)(vt < LRN_BLOCKS ? vt : LRN_BLOCKS);
    static std::atomic<size_t> granted_eval[3] = {{48 * 1024}, {48 * 1024}, {48 * 1024}};
    const int ev = L <= 8 ? 0 : (L <= 16 ? 1 : 2);
    const void *evf = ev == 0 ? (const void *)k_learner_eval<8> : ev == 1 ? (const void *)k_learner_eval<16> : (const void *)k_learner_eval<64>;
    if (lds > granted_eval[ev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(evf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DDG_E_LAUNCH;
        granted_eval[ev].store(lds, std::memory_order_release);
    }
    if (ev == 0) k_learner_eval<8><<<vgrid, LRN_NT, lds, s>>>(v, loss_part, hit_part);
    else if (ev == 1) k_learner_eval<16><<<vgrid, LRN_NT, lds, s>>>(v, loss_part, hit_part);
    else k_learner_eval<64><<<vgrid, LRN_NT, lds, s>>>(v, loss_part, hit_part);
    k_learner_eval_finish<<<1, 64, 0, s>>>(loss_part, hit_part, vgrid, n_valid, out_val_loss, out_val_correct);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
