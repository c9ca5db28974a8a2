// Row-attention top-k fill (hot kernel #1) for gfx950: masked similarity on the fp32 MFMA with
// LDS-staged key tiles, exact k-th-logit selection, softmax-weighted accumulation, mask-merge.
//
// Reference restated: init_attn_2's weighted branch, utils/util.py:439-496 (closed form in
// SURVEY.md 3.3).  What the reference does per table column c with dense N x N matrices
//     S = (X0 * W[c]) X0^T ; diag, rows of keys missing c -> -inf ; P = softmax(S, dim=0)
//     thr = k-th largest of P[:, j] ; P[P < thr] = 0 ; P /= colsum ; fill = P^T X0[:, blk(c)]
// is done here per (query row j, column c) pair with m[j,c]==0 only (all others are discarded by
// the reference's own mask-merge, utils/util.py:496) and only against keys i with m[i,c]==1
// (all others are -inf in the reference); the query's own row is never among those keys.
//
// Data layout in HBM (workspace):
//   Xkp [Nk][DP]   key rows, zero padded to DP = 8*KS4 columns (16-byte aligned rows)
//   Xqp [Nq][DP]   query rows (aliases Xkp when queries and keys are the same array)
//   Wp  [C][DP]    column weights
//   klist [C][Nk]  ascending indices of the keys observed in column c (nk[c] of them)
//   qlist [QB*128] per column, ascending local indices of the rows MISSING column c, each
//                  column padded with -1 to a multiple of 128 (one block = 128 queries of ONE column)
//   tau / smax / prefix / rank  [QB*128]  per-query selection state
//
// Kernel geometry: block = 4 waves; wave w owns 32 queries whose operand (W[c] * x_j, DP/2
// registers per lane) stays in registers for the whole key stream; key tiles of 32 rows are
// gathered through klist into double-buffered LDS (row stride DP+4 floats: conflict-free
// ds_read_b128) and shared by the 4 waves.  One v_mfma_f32_32x32x2_f32 chain of DP/2 steps
// gives a 32 keys x 32 queries logit tile: lane l holds query (l&31), key rows
// (g&3)+8(g>>2)+4(l>>5), g = 0..15.  The k order is permuted identically for both operands.
#include <math.h>
#include "common.h"

#define QPB 128          // queries per block
#define TK 32            // keys per tile
#define HBINS 128
#define HSTRIDE 129

enum { MODE_TOPK = 0, MODE_HIST = 1, MODE_FILL = 2 };

struct AttnArgs {
    const float *Xkp, *Xqp, *Wp;
    const int32_t *klist;      // [C][Nk]
    const int32_t *nk;         // [C]
    const int32_t *qlist;      // [QB*128]
    const int32_t *qb_start;   // [C+1] in blocks
    const int32_t *offsets;    // [C+1] device
    float *tau, *smax;
    uint32_t *prefix;
    int32_t *rank;
    int32_t *flag;             // [QB*128] MODE_TOPK: 1 = ties at the k-th logit, pair left to the MODE_FILL pass
    float *out_code;           // [Nq][D]
    float *out_tau;            // [Nq][C] or null
    int32_t *out_nvalid;       // [Nq][C] or null
    int64_t Nk;
    int C, D, DP;
    int k;
    int level;                 // radix level (MODE_HIST)
    int dcs;                   // LDS accumulator stride (MODE_FILL)
};

__device__ __forceinline__ uint32_t f2key(float v) {
    uint32_t u = __float_as_uint(v + 0.f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// radix levels: bits [31:25], [24:18], [17:11], [10:4], [3:0]
__device__ __forceinline__ int level_shift(int L) { return L < 4 ? 25 - 7 * L : 0; }
__device__ __forceinline__ int level_bits(int L) { return L < 4 ? 7 : 4; }

// sorted (descending) insertion that carries the key position along with the logit
template <int KCAP>
__device__ __forceinline__ void list_insert_kv(float (&lst)[KCAP], int (&pos)[KCAP], float v, int q) {
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        const bool gt = v > lst[i];
        const float lv = lst[i];
        const int lq = pos[i];
        lst[i] = gt ? v : lv;
        pos[i] = gt ? q : lq;
        v = gt ? lv : v;
        q = gt ? lq : q;
    }
}

template <int KS4, int MODE, int KCAP>
__global__ __launch_bounds__(256, (KS4 >= 32 ? 1 : 2)) void k_attn(AttnArgs p) {
    constexpr int DP = KS4 * 8;
    constexpr int LSTR = DP + 4;
    constexpr int NPF = KS4 / 4;           // float4 prefetch registers per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *kbuf0 = smem;
    float *kbuf1 = smem + TK * LSTR;
    float *extra = smem + 2 * TK * LSTR;   // MODE_HIST: histograms; MODE_FILL: accumulators
    __shared__ int s_col;

    const int nblocks = p.qb_start[p.C];
    if ((int)blockIdx.x >= nblocks) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    if (tid == 0) {
        int lo = 0, hi = p.C;      // largest c with qb_start[c] <= blockIdx.x
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.qb_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
        }
        s_col = lo;
    }
    __syncthreads();
    const int c = s_col;
    const int nk = p.nk[c];
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    const int off_c = p.offsets[c];
    const int d_c = p.offsets[c + 1] - off_c;

    const int64_t slot = ((int64_t)blockIdx.x * 4 + wid) * 32 + r32;
    const int qrow = p.qlist[slot];

    // ---- query operand in registers: breg[4u+s] = W[c][k] * x_q[k], k = 8u + 4h + s
    float breg[KS4 * 4];
    {
        const float *wq = p.Wp + (int64_t)c * DP + 4 * h;
        const float *xq = p.Xqp + (int64_t)(qrow < 0 ? 0 : qrow) * DP + 4 * h;
#pragma unroll
        for (int u = 0; u < KS4; ++u) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + 8 * u);
            f32x4 xv = *reinterpret_cast<const f32x4 *>(xq + 8 * u);
            if (qrow < 0) xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) breg[4 * u + s] = wv[s] * xv[s];
        }
    }

    // ---- per-mode state
    // MODE_TOPK tracks the k+1 largest logits (and their positions in the key list): slots
    // [ndummy, KCAP) are real, so the k-th largest ends in lst[KCAP-2] and the (k+1)-th in lst[KCAP-1]
    float lst[KCAP];
    int lpos[KCAP];
    float smax = -INFINITY;
    float tau = 0.f, Z = 0.f, acc1 = 0.f;
    uint32_t prefix = 0;
    bool active = true;
    if constexpr (MODE == MODE_TOPK) {
        const int ndummy = KCAP - (p.k + 1);
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            lst[i] = (i < ndummy) ? INFINITY : -INFINITY;
            lpos[i] = -1;
        }
    }
    if (MODE == MODE_HIST) {
        uint32_t *hist = reinterpret_cast<uint32_t *>(extra);
        for (int i = tid; i < QPB * HSTRIDE; i += 256) hist[i] = 0;
        if (p.level > 0) {
            prefix = p.prefix[slot];
            active = p.rank[slot] > 0;
        }
    }
    if (MODE == MODE_FILL) {
        tau = p.tau[slot];
        smax = p.smax[slot];
        float *accb = extra + (int64_t)tid * p.dcs;
        for (int e = 0; e < d_c; ++e) accb[e] = 0.f;
    }

    // ---- key streaming
    const int st_row = tid >> 3, st_c = tid & 7;
    f32x4 pf[NPF];
    auto prefetch = [&](int t) {
        const int pos = t * TK + st_row;
        const int kidx = pos < nk ? kl[pos] : -1;
        const float *src = p.Xkp + (int64_t)(kidx < 0 ? 0 : kidx) * DP + st_c * 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            pf[i] = *reinterpret_cast<const f32x4 *>(src + 32 * i);
            if (kidx < 0) pf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&](float *buf) {
        float *dst = buf + st_row * LSTR + st_c * 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) *reinterpret_cast<f32x4 *>(dst + 32 * i) = pf[i];
    };

    const int ntk = (nk + TK - 1) / TK;
    if (ntk > 0) { prefetch(0); commit(kbuf0); }
    __syncthreads();

    for (int t = 0; t < ntk; ++t) {
        float *cur = (t & 1) ? kbuf1 : kbuf0;
        float *nxt = (t & 1) ? kbuf0 : kbuf1;
        if (t + 1 < ntk) prefetch(t + 1);

        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        const float *Ab = cur + r32 * LSTR + 4 * h;
#pragma unroll
        for (int u = 0; u < KS4; ++u) {
            const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Ab + 8 * u);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], breg[4 * u + s], acc, 0, 0, 0);
        }

        const int rem = nk - t * TK;          // valid key rows in this tile (>= 1)
        const bool full = rem >= TK;

        if constexpr (MODE == MODE_TOPK) {
            float thr = lst[KCAP - 1];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                float v = acc[g];
                if (!full && kr >= rem) v = -INFINITY;
                if (__any(v > thr)) {
                    list_insert_kv<KCAP>(lst, lpos, v, t * TK + kr);
                    thr = lst[KCAP - 1];
                }
            }
        } else if (MODE == MODE_HIST) {
            uint32_t *hist = reinterpret_cast<uint32_t *>(extra) + (wid * 32 + r32) * HSTRIDE;
            const int sh = level_shift(p.level);
            const int nb = level_bits(p.level);
            const uint32_t bmask = (1u << nb) - 1u;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                const float v = acc[g];
                const bool valid = (full || kr < rem) && active && qrow >= 0;
                if (valid) {
                    if (p.level == 0) smax = fmaxf(smax, v);
                    const uint32_t key = f2key(v);
                    const bool match = (p.level == 0) || ((key >> (sh + nb)) == (prefix >> (sh + nb)));
                    if (match) atomicAdd(&hist[(key >> sh) & bmask], 1u);
                }
            }
        } else {  // MODE_FILL
            float *accb = extra + (int64_t)tid * p.dcs;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                const float v = acc[g];
                const bool keep = (full || kr < rem) && (v >= tau);
                if (__any(keep)) {
                    const float w = keep ? expf(v - smax) : 0.f;
                    Z += w;
                    const float *xr = cur + kr * LSTR + off_c;
                    if (d_c == 1) {
                        acc1 += w * xr[0];
                    } else {
                        for (int e = 0; e < d_c; ++e) accb[e] += w * xr[e];
                    }
                }
            }
        }

        if (t + 1 < ntk) commit(nxt);
        __syncthreads();
    }

    // ---- per-mode finish
    if constexpr (MODE == MODE_TOPK) {
        // merge the two half-wave lists of each query, then (lanes h == 0) softmax over the kept k
        // entries and gather their column-c values: the fill without a second similarity pass
        const int ndummy = KCAP - (p.k + 1);
        float ov[KCAP];
        int op[KCAP];
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            ov[i] = __shfl_xor(lst[i], 32);
            op[i] = __shfl_xor(lpos[i], 32);
        }
#pragma unroll
        for (int i = 0; i < KCAP; ++i) list_insert_kv<KCAP>(lst, lpos, (i >= ndummy) ? ov[i] : -INFINITY, op[i]);
        if (h == 0 && qrow >= 0) {
            const bool tie = nk > p.k && lst[KCAP - 1] == lst[KCAP - 2];
            p.tau[slot] = (nk >= p.k) ? lst[KCAP - 2] : -INFINITY;
            float vmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < KCAP - 1; ++i)
                if (i >= ndummy) vmax = fmaxf(vmax, lst[i]);
            p.smax[slot] = vmax;
            p.flag[slot] = tie ? 1 : 0;
            if (!tie) {
                float wgt[KCAP];
                float Zs = 0.f;
#pragma unroll
                for (int i = 0; i < KCAP - 1; ++i) {
                    const bool kept = i >= ndummy && lpos[i] >= 0;
                    wgt[i] = kept ? expf(lst[i] - vmax) : 0.f;
                    Zs += wgt[i];
                    lpos[i] = kept ? kl[lpos[i]] : -1;      // key position -> key row
                }
                float *dst = p.out_code + (int64_t)qrow * p.D + off_c;
                for (int e = 0; e < d_c; ++e) {
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < KCAP - 1; ++i)
                        if (lpos[i] >= 0) a += wgt[i] * p.Xkp[(int64_t)lpos[i] * DP + off_c + e];
                    dst[e] = (Zs > 0.f) ? a / Zs : 0.f;
                }
            }
        }
    } else if (MODE == MODE_HIST) {
        smax = fmaxf(smax, __shfl_xor(smax, 32));
        if (p.level == 0 && h == 0) p.smax[slot] = smax;
        __syncthreads();   // all histogram atomics of the block are done
        if (tid < QPB) {
            const int64_t sl = (int64_t)blockIdx.x * QPB + tid;
            const uint32_t *hq = reinterpret_cast<uint32_t *>(extra) + tid * HSTRIDE;
            int r = (p.level == 0) ? p.k : p.rank[sl];
            uint32_t pre = (p.level == 0) ? 0u : p.prefix[sl];
            const int sh = level_shift(p.level);
            const int nbins = 1 << level_bits(p.level);
            if (p.level == 0 && nk < p.k) r = 0;     // fewer valid keys than k: keep them all
            if (r > 0) {
                int cum = 0, b = nbins - 1;
                for (; b > 0; --b) {
                    const int cnt = (int)hq[b];
                    if (cum + cnt >= r) break;
                    cum += cnt;
                }
                r -= cum;
                pre |= ((uint32_t)b) << sh;
            }
            p.rank[sl] = r;
            p.prefix[sl] = pre;
            if (p.level == 4) p.tau[sl] = (r > 0) ? key2f(pre) : -INFINITY;
        }
    } else {
        const float Zt = Z + __shfl_xor(Z, 32);
        if (d_c == 1) {
            const float a = acc1 + __shfl_xor(acc1, 32);
            if (h == 0 && qrow >= 0) p.out_code[(int64_t)qrow * p.D + off_c] = (Zt > 0.f) ? a / Zt : 0.f;
        } else {
            __syncthreads();
            if (h == 0 && qrow >= 0) {
                const float *a0 = extra + (int64_t)tid * p.dcs;
                const float *a1 = extra + (int64_t)(tid + 32) * p.dcs;
                for (int e = 0; e < d_c; ++e)
                    p.out_code[(int64_t)qrow * p.D + off_c + e] = (Zt > 0.f) ? (a0[e] + a1[e]) / Zt : 0.f;
            }
        }
        if (h == 0 && qrow >= 0) {
            if (p.out_tau) p.out_tau[(int64_t)qrow * p.C + c] = tau;
            if (p.out_nvalid) p.out_nvalid[(int64_t)qrow * p.C + c] = nk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// preparation kernels: padding and per-column index lists
// ---------------------------------------------------------------------------------------------
__global__ void k_pad_rows(const float *__restrict__ X, float *__restrict__ Xp, int64_t rows, int D, int DP) {
    const int64_t n = rows * DP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / DP;
        const int k = (int)(e - r * DP);
        Xp[e] = k < D ? X[r * D + k] : 0.f;
    }
}

// counts of observed cells per (column, chunk of 256 rows):  cnt[c][chunk]
__global__ void k_count_obs(const uint8_t *__restrict__ m, int64_t rows, int C, int nchunks, int32_t *__restrict__ cnt) {
    __shared__ int wsum[4];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        const bool obs = row < rows && m[row * C + c] != 0;
        const unsigned long long b = __ballot(obs);
        if (lane == 0) wsum[wid] = __popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) cnt[(int64_t)c * nchunks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// exclusive scan over chunks, one block per column; total -> tot[c]
__global__ void k_scan_chunks(int32_t *__restrict__ cnt, int nchunks, int32_t *__restrict__ tot) {
    __shared__ int part[256];
    int32_t *row = cnt + (int64_t)blockIdx.x * nchunks;
    const int per = (nchunks + 255) / 256;
    const int b = threadIdx.x * per, e = (b + per < nchunks) ? b + per : nchunks;
    int s = 0;
    for (int i = b; i < e; ++i) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
        tot[blockIdx.x] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = b; i < e; ++i) { const int v = row[i]; row[i] = run; run += v; }
}

// qb_start[c] = sum_{c'<c} ceil((rows - obs[c'])/128)
__global__ void k_qb_start(const int32_t *__restrict__ obs_q, int64_t rows, int C, int32_t *__restrict__ qb_start) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) {
            qb_start[c] = run;
            run += (int)(((int64_t)rows - obs_q[c] + QPB - 1) / QPB);
        }
        qb_start[C] = run;
    }
}

// write the lists.  WRITE_K: klist[c][base + pos] = row for observed rows.
// WRITE_Q: qlist[qb_start[c]*128 + (chunk_row0 - obs_before) + pos_missing] = row for missing rows.
template <bool WRITE_K, bool WRITE_Q>
__global__ void k_fill_lists(const uint8_t *__restrict__ m, int64_t rows, int C, int nchunks,
                             const int32_t *__restrict__ cnt_excl, int32_t *__restrict__ klist, int64_t kstride,
                             const int32_t *__restrict__ qb_start, int32_t *__restrict__ qlist) {
    __shared__ int wsum[4];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int64_t row = row0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        const bool inb = row < rows;
        const bool obs = inb && m[row * C + c] != 0;
        const unsigned long long b = __ballot(obs);
        const int before_in_wave = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(b);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wsum[w];
        const int pos_obs = wbase + before_in_wave;                 // observed rows before me in the chunk
        const int obs_before = cnt_excl[(int64_t)c * nchunks + blockIdx.x];
        if (WRITE_K && obs) klist[(int64_t)c * kstride + obs_before + pos_obs] = (int32_t)row;
        if (WRITE_Q && inb && !obs) {
            const int64_t miss_before = row0 - obs_before;          // missing rows in earlier chunks
            const int pos_miss = threadIdx.x - pos_obs;             // missing rows before me in the chunk
            qlist[(int64_t)qb_start[c] * QPB + miss_before + pos_miss] = (int32_t)row;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// tie fallback of the register path: pairs whose k-th and (k+1)-th logits are equal (flag == 1)
// are compacted into a second query list, one block = 128 pairs of ONE column as before, and go
// through the MODE_FILL pass.  cnt2[c] = flagged pairs of column c, cur2[c] = append cursor.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_column(const int32_t *qb_start, int C, int b) {
    int lo = 0, hi = C;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (qb_start[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool WRITE>
__global__ void k_compact_flagged(const int32_t *__restrict__ flag, const int32_t *__restrict__ qlist,
                                  const float *__restrict__ tau, const float *__restrict__ smax,
                                  const int32_t *__restrict__ qb_start, int C, int32_t *__restrict__ cnt2,
                                  int32_t *__restrict__ cur2, const int32_t *__restrict__ qb_start2,
                                  int32_t *__restrict__ qlist2, float *__restrict__ tau2, float *__restrict__ smax2) {
    __shared__ int s_base, s_n;
    if ((int)blockIdx.x >= qb_start[C]) return;
    const int c = block_column(qb_start, C, blockIdx.x);
    const int64_t slot = (int64_t)blockIdx.x * QPB + threadIdx.x;       // blockDim.x == QPB
    const int q = qlist[slot];
    const bool f = q >= 0 && flag[slot] != 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long b = __ballot(f);
    int wbase = 0;
    if (lane == 0 && b) wbase = atomicAdd(&s_n, __popcll(b));
    wbase = __shfl(wbase, 0);
    __syncthreads();
    if (s_n == 0) return;
    if (!WRITE) {
        if (threadIdx.x == 0) atomicAdd(&cnt2[c], s_n);
        return;
    }
    if (threadIdx.x == 0) s_base = atomicAdd(&cur2[c], s_n);
    __syncthreads();
    if (f) {
        const int64_t d = (int64_t)qb_start2[c] * QPB + s_base + wbase + __popcll(b & ((1ull << lane) - 1ull));
        qlist2[d] = q;
        tau2[d] = tau[slot];
        smax2[d] = smax[slot];
    }
}

// qb_start2[c] = sum_{c'<c} ceil(cnt2[c']/128)
__global__ void k_qb_start2(const int32_t *__restrict__ cnt2, int C, int32_t *__restrict__ qb_start2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) {
            qb_start2[c] = run;
            run += (cnt2[c] + QPB - 1) / QPB;
        }
        qb_start2[C] = run;
    }
}

// debug outputs of the register path: tau and the number of valid keys per (row, column) pair
__global__ void k_write_debug(const int32_t *__restrict__ qlist, const float *__restrict__ tau,
                              const int32_t *__restrict__ qb_start, const int32_t *__restrict__ nk, int C,
                              float *__restrict__ out_tau, int32_t *__restrict__ out_nvalid) {
    if ((int)blockIdx.x >= qb_start[C]) return;
    const int c = block_column(qb_start, C, blockIdx.x);
    const int64_t slot = (int64_t)blockIdx.x * QPB + threadIdx.x;
    const int q = qlist[slot];
    if (q < 0) return;
    if (out_tau) out_tau[(int64_t)q * C + c] = tau[slot];
    if (out_nvalid) out_nvalid[(int64_t)q * C + c] = nk[c];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int ks4_for(int D) { return D <= 64 ? 8 : D <= 128 ? 16 : D <= 192 ? 24 : D <= 256 ? 32 : 0; }

struct AttnWs {
    float *Xkp, *Xqp, *Wp;
    int32_t *klist, *nk, *obs_q, *cnt_k, *cnt_q, *qlist, *qb_start, *offsets;
    float *tau, *smax;
    uint32_t *prefix;
    int32_t *rank;
    int32_t *flags;            // [0] = 1 when queries alias keys (set by prepare)
    // tie fallback of the register path
    int32_t *flag, *qlist2, *qb_start2, *cnt2;   // cnt2: [C] counts then [C] cursors
    float *tau2, *smax2;
    size_t total;
    int64_t qb_max;
    int nchunk_k, nchunk_q;
};

static void attn_carve(AttnWs &w, char *base, int64_t Nq, int64_t Nk, int C, int D, bool same) {
    const int DP = ks4_for(D) * 8;
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += al256(bytes); return p; };
    w.nchunk_k = (int)ceil_div64(Nk, 256);
    w.nchunk_q = (int)ceil_div64(Nq, 256);
    w.qb_max = (Nq * C) / QPB + C;
    w.Xkp = (float *)take((size_t)Nk * DP * 4);
    w.Xqp = (float *)take((size_t)Nq * DP * 4);
    if (same) w.Xqp = w.Xkp;
    w.Wp = (float *)take((size_t)C * DP * 4);
    w.klist = (int32_t *)take((size_t)C * Nk * 4);
    w.nk = (int32_t *)take((size_t)C * 4);
    w.obs_q = (int32_t *)take((size_t)C * 4);
    w.cnt_k = (int32_t *)take((size_t)C * w.nchunk_k * 4);
    w.cnt_q = (int32_t *)take((size_t)C * w.nchunk_q * 4);
    if (same) w.cnt_q = w.cnt_k;
    w.qlist = (int32_t *)take((size_t)w.qb_max * QPB * 4);
    w.qb_start = (int32_t *)take((size_t)(C + 1) * 4);
    w.offsets = (int32_t *)take((size_t)(C + 1) * 4);
    w.tau = (float *)take((size_t)w.qb_max * QPB * 4);
    w.smax = (float *)take((size_t)w.qb_max * QPB * 4);
    w.prefix = (uint32_t *)take((size_t)w.qb_max * QPB * 4);
    w.rank = (int32_t *)take((size_t)w.qb_max * QPB * 4);
    w.flags = (int32_t *)take(256);
    w.flag = (int32_t *)take((size_t)w.qb_max * QPB * 4);
    w.qlist2 = (int32_t *)take((size_t)w.qb_max * QPB * 4);
    w.tau2 = (float *)take((size_t)w.qb_max * QPB * 4);
    w.smax2 = (float *)take((size_t)w.qb_max * QPB * 4);
    w.qb_start2 = (int32_t *)take((size_t)(C + 1) * 4);
    w.cnt2 = (int32_t *)take((size_t)2 * C * 4);
    w.total = o;
}

extern "C" size_t ddg_attn_workspace_bytes(int64_t Nq, int64_t Nk, int C, int D, int64_t k) {
    (void)k;
    if (ks4_for(D) == 0 || Nq < 0 || Nk < 0 || C < 1) return 0;
    AttnWs w;
    attn_carve(w, nullptr, Nq, Nk, C, D, false);   // the layout does not depend on aliasing
    return w.total;
}

template <int KS4, int MODE, int KCAP>
static int launch_attn(const AttnArgs &a, int64_t qb_max, size_t extra_bytes, hipStream_t s) {
    const size_t lds = (size_t)2 * TK * (KS4 * 8 + 4) * 4 + extra_bytes;
    if (lds > 160 * 1024 - 64) return DDG_E_UNSUPPORTED;
    static size_t attr_lds = 48 * 1024;   // dynamic LDS above the default limit must be opted into
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void *)k_attn<KS4, MODE, KCAP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            fprintf(stderr, "ddgani: hipFuncSetAttribute(%zu): %s\n", lds, hipGetErrorString(e));
            return DDG_E_LAUNCH;
        }
        attr_lds = lds;
    }
    k_attn<KS4, MODE, KCAP><<<(unsigned)qb_max, 256, lds, s>>>(a);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

template <int KS4>
static int run_select(AttnArgs &a, int64_t qb_max, hipStream_t s) {
    int rc;
    if (a.k < 16) return launch_attn<KS4, MODE_TOPK, 16>(a, qb_max, 0, s);
    if (a.k <= DDG_MAX_TOPK_REG) return launch_attn<KS4, MODE_TOPK, 32>(a, qb_max, 0, s);
    for (int L = 0; L < 5; ++L) {
        a.level = L;
        rc = launch_attn<KS4, MODE_HIST, 1>(a, qb_max, (size_t)QPB * HSTRIDE * 4, s);
        if (rc) return rc;
    }
    return DDG_OK;
}

template <int KS4>
static int run_fill(AttnArgs &a, int64_t qb_max, int dcmax, hipStream_t s) {
    a.dcs = dcmax | 1;
    return launch_attn<KS4, MODE_FILL, 1>(a, qb_max, (size_t)256 * a.dcs * 4, s);
}

static int attn_check_dims(int64_t Nq, int64_t Nk, int C, int D) {
    if (Nq < 0 || Nk < 1 || C < 1 || D < 1) return DDG_E_ARG;
    if (ks4_for(D) == 0 || Nk >= (int64_t)1 << 31 || (Nq * C) / QPB + C >= (int64_t)1 << 31) return DDG_E_UNSUPPORTED;
    return DDG_OK;
}

static void attn_args(AttnArgs &a, const AttnWs &w, int64_t Nk, int C, int D, int k) {
    a.Xkp = w.Xkp; a.Xqp = w.Xqp; a.Wp = w.Wp; a.klist = w.klist; a.nk = w.nk; a.qlist = w.qlist;
    a.qb_start = w.qb_start; a.offsets = w.offsets; a.tau = w.tau; a.smax = w.smax; a.prefix = w.prefix;
    a.rank = w.rank; a.flag = w.flag; a.out_code = nullptr; a.out_tau = nullptr; a.out_nvalid = nullptr; a.Nk = Nk; a.C = C;
    a.D = D; a.DP = ks4_for(D) * 8; a.k = k; a.level = 0; a.dcs = 1;
}

// phase 1: padded operands + per-column key / query index lists
extern "C" int ddg_attn_prepare(const float *Xq0, const uint8_t *mq, int64_t q_row_offset, const float *Xk0,
                                const uint8_t *mk, const float *W, const int32_t *offsets_host, int64_t Nq,
                                int64_t Nk, int C, int D, void *workspace, size_t workspace_bytes, void *stream) {
    if (!Xq0 || !mq || !Xk0 || !mk || !W || !offsets_host) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D);
    if (rc) return rc;
    if (q_row_offset < 0 || q_row_offset + Nq > Nk) return DDG_E_ARG;
    if (offsets_host[0] != 0 || offsets_host[C] != D) return DDG_E_ARG;
    for (int c = 0; c < C; ++c)
        if (offsets_host[c + 1] - offsets_host[c] < 1) return DDG_E_ARG;
    if (Nq == 0) return DDG_OK;
    const bool same = (Xq0 == Xk0 && mq == mk && Nq == Nk);
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, same);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int DP = ks4_for(D) * 8;

    if (hipMemcpyAsync(w.offsets, offsets_host, (size_t)(C + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        return DDG_E_LAUNCH;
    if (hipMemsetAsync(w.qlist, 0xFF, (size_t)w.qb_max * QPB * 4, s) != hipSuccess) return DDG_E_LAUNCH;
    if (hipMemsetAsync(w.flags, same ? 1 : 0, 4, s) != hipSuccess) return DDG_E_LAUNCH;

    auto egrid = [](int64_t n) { int64_t g = ceil_div64(n, 256); return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
    k_pad_rows<<<egrid(Nk * DP), 256, 0, s>>>(Xk0, w.Xkp, Nk, D, DP);
    if (!same) k_pad_rows<<<egrid(Nq * DP), 256, 0, s>>>(Xq0, w.Xqp, Nq, D, DP);
    k_pad_rows<<<egrid((int64_t)C * DP), 256, 0, s>>>(W, w.Wp, C, D, DP);

    k_count_obs<<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k);
    k_scan_chunks<<<C, 256, 0, s>>>(w.cnt_k, w.nchunk_k, w.nk);
    if (same) {
        k_qb_start<<<1, 64, 0, s>>>(w.nk, Nq, C, w.qb_start);
        k_fill_lists<true, true><<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, w.klist, Nk, w.qb_start, w.qlist);
    } else {
        k_count_obs<<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q);
        k_scan_chunks<<<C, 256, 0, s>>>(w.cnt_q, w.nchunk_q, w.obs_q);
        k_qb_start<<<1, 64, 0, s>>>(w.obs_q, Nq, C, w.qb_start);
        k_fill_lists<true, false><<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, w.klist, Nk, w.qb_start, w.qlist);
        k_fill_lists<false, true><<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q, w.klist, Nk, w.qb_start, w.qlist);
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// phase 2: one similarity pass.  out_code <- Xq0, then
//   k <= DDG_MAX_TOPK_REG: the k+1 largest logits and their key rows per (query row, column) pair
//     in registers; pairs without a tie at the k-th logit are filled right here (softmax over
//     the kept k, gather of the key values), the others are flagged for phase 3;
//   larger k: 5-level radix select of tau (5 similarity passes), every pair is filled in phase 3.
extern "C" int ddg_attn_select(const float *Xq0, int64_t Nq, int64_t Nk, int C, int D, int64_t k,
                               int queries_alias_keys, float *out_code, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (!Xq0 || !out_code) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D);
    if (rc) return rc;
    if (k < 1) return DDG_E_ARG;
    if (Nq == 0) return DDG_OK;
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, queries_alias_keys != 0);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (out_code != Xq0 &&
        hipMemcpyAsync(out_code, Xq0, (size_t)Nq * D * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return DDG_E_LAUNCH;
    AttnArgs a;
    attn_args(a, w, Nk, C, D, (int)(k > Nk ? Nk : k));
    a.out_code = out_code;
    switch (ks4_for(D)) {
        case 8: return run_select<8>(a, w.qb_max, s);
        case 16: return run_select<16>(a, w.qb_max, s);
        case 24: return run_select<24>(a, w.qb_max, s);
        case 32: return run_select<32>(a, w.qb_max, s);
    }
    return DDG_E_UNSUPPORTED;
}

// phase 3: softmax-weighted accumulation over {s >= tau} (second similarity pass) for the pairs
// phase 2 left open: all of them on the radix path, only the tie-flagged ones on the register path
extern "C" int ddg_attn_fill(const int32_t *offsets_host, int64_t Nq, int64_t Nk, int C, int D, int64_t k,
                             int queries_alias_keys, float *out_code, float *out_tau, int32_t *out_nvalid,
                             void *workspace, size_t workspace_bytes, void *stream) {
    if (!offsets_host || !out_code) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D);
    if (rc) return rc;
    if (k < 1) return DDG_E_ARG;
    if (Nq == 0) return DDG_OK;
    int dcmax = 1;
    for (int c = 0; c < C; ++c) {
        const int d = offsets_host[c + 1] - offsets_host[c];
        if (d < 1) return DDG_E_ARG;
        if (d > dcmax) dcmax = d;
    }
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, queries_alias_keys != 0);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    AttnArgs a;
    const int kk = (int)(k > Nk ? Nk : k);
    attn_args(a, w, Nk, C, D, kk);
    a.out_code = out_code; a.out_tau = out_tau; a.out_nvalid = out_nvalid;
    if (kk <= DDG_MAX_TOPK_REG) {
        const unsigned g = (unsigned)w.qb_max;
        if (hipMemsetAsync(w.cnt2, 0, (size_t)2 * C * 4, s) != hipSuccess) return DDG_E_LAUNCH;
        if (hipMemsetAsync(w.qlist2, 0xFF, (size_t)w.qb_max * QPB * 4, s) != hipSuccess) return DDG_E_LAUNCH;
        k_compact_flagged<false><<<g, QPB, 0, s>>>(w.flag, w.qlist, w.tau, w.smax, w.qb_start, C, w.cnt2, w.cnt2 + C,
                                                   w.qb_start2, w.qlist2, w.tau2, w.smax2);
        k_qb_start2<<<1, 64, 0, s>>>(w.cnt2, C, w.qb_start2);
        k_compact_flagged<true><<<g, QPB, 0, s>>>(w.flag, w.qlist, w.tau, w.smax, w.qb_start, C, w.cnt2, w.cnt2 + C,
                                                  w.qb_start2, w.qlist2, w.tau2, w.smax2);
        if (out_tau || out_nvalid)
            k_write_debug<<<g, QPB, 0, s>>>(w.qlist, w.tau, w.qb_start, w.nk, C, out_tau, out_nvalid);
        DDG_CHECK_LAUNCH();
        a.qlist = w.qlist2; a.qb_start = w.qb_start2; a.tau = w.tau2; a.smax = w.smax2;
        a.out_tau = nullptr; a.out_nvalid = nullptr;
    }
    switch (ks4_for(D)) {
        case 8: return run_fill<8>(a, w.qb_max, dcmax, s);
        case 16: return run_fill<16>(a, w.qb_max, dcmax, s);
        case 24: return run_fill<24>(a, w.qb_max, dcmax, s);
        case 32: return run_fill<32>(a, w.qb_max, dcmax, s);
    }
    return DDG_E_UNSUPPORTED;
}

extern "C" int ddg_attn_topk_fill(const float *Xq0, const uint8_t *mq, int64_t q_row_offset, const float *Xk0,
                                  const uint8_t *mk, const float *W, const int32_t *offsets_host, int64_t Nq,
                                  int64_t Nk, int C, int D, int64_t k, float *out_code, float *out_tau,
                                  int32_t *out_nvalid, void *workspace, size_t workspace_bytes, void *stream) {
    if (!out_code || k < 1) return DDG_E_ARG;
    int rc = ddg_attn_prepare(Xq0, mq, q_row_offset, Xk0, mk, W, offsets_host, Nq, Nk, C, D, workspace,
                              workspace_bytes, stream);
    if (rc) return rc;
    const int same = (Xq0 == Xk0 && mq == mk && Nq == Nk) ? 1 : 0;
    rc = ddg_attn_select(Xq0, Nq, Nk, C, D, k, same, out_code, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return ddg_attn_fill(offsets_host, Nq, Nk, C, D, k, same, out_code, out_tau, out_nvalid, workspace,
                         workspace_bytes, stream);
}
