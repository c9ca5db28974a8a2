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
//   Xkp [Nk][DP]   key rows, zero padded to DP = 8*KS4 columns (16-byte aligned rows); in table order, except on the
//                  screening path (k <= DDG_MAX_TOPK_REG, filter mode): there in descending-norm order (k_bucket_count)
//   Xqp [Nq][DP]   query rows, table order (aliases Xkp when queries and keys are the same array and Xkp is in table order)
//   Wp  [C][DP]    column weights
//   klist [C][Nk]  ascending positions in Xkp of the keys observed in column c (nk[c] of them)
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
#include <stdlib.h>
#include "common.h"
#include <atomic>
#include <type_traits>

#define QPB 128          // queries per block
#define TK 32            // keys per tile
#define HBINS 128
#define HSTRIDE 129
// window passes: counter row of a query = [0] below the bracket, [1 .. WBINS] its bins, [WBINS + 1] above.  59 bins: the
// rows of a workgroup (30.5 KB) and the two key tiles (49 KB at DP = 192) leave room for two workgroups per CU
#define WBINS 59
#define WHSTRIDE 61

enum { MODE_TOPK = 0, MODE_HIST = 1, MODE_FILL = 2, MODE_WIN = 3, MODE_WFILL = 4 };
// Window selection for k > DDG_MAX_TOPK_REG (run_window_select / run_window_fill below): the k-th largest logit of
// a pair is bracketed on a strided SAMPLE of its keys (three cheap passes), the bracket is resolved to one of 59
// bins by ONE full pass that also counts what lies above it, and the second full pass accumulates everything
// above that bin while listing the bin's own keys (<= WIN_CAP) for an exact in-list selection (k_attn_wlist).
// Two full similarity passes instead of the radix path's six; pairs whose bracket misses go through the radix path.
enum { WIN_MINMAX = 0, WIN_SHIST = 1, WIN_FULL = 2 };
#define WIN_SAMPLE 8192      // sampled keys per column at least (stride = nk / sample size, win_kstride below)
#define WIN_BIN_KEYS 2212.f  // sample size = (nk sqrt(q (1 - q)) / WIN_BIN_KEYS)^2: about 450 keys in a bin of the full pass
#define WIN_CAP 1024         // listed keys per pair
#define WIN_GROUP 4096       // workgroups per launch group of the second full pass (bounds the lists)
#define WIN_SIGMAS 6.f       // half width of the sample bracket in standard deviations of the sample rank
#define ATTN_WINDOW 100      // internal selector of run_select (not a public mode)
#define WST_OPENTOP 1        // the bracket has no upper edge: logits above it count into bin 127
#define WST_OPENBOT 2        // ... no lower edge: logits below it count into bin 0
#define WST_ALL 4            // nk < k: every valid key is kept
// The register path tracks the k+2 largest logits of a pair: the k kept ones, the (k+1)-th that decides
// whether the k-th is tied, and the (k+2)-th that tells a two-way tie (both kept, resolved on the spot)
// from a longer one (left to the threshold pass).
#define NLIST(k) ((k) + 2)
static_assert(NLIST(DDG_MAX_TOPK_REG) <= 32, "register lists hold at most 32 entries");

// buckets of the norm-ordered key lists of the screening path (see k_bucket_count): bucket b holds the keys with
// floor(norm / largest norm * KEY_BUCKETS) == KEY_BUCKETS - 1 - b, so every key of bucket b or later has
// norm < largest norm * (KEY_BUCKETS - b) / KEY_BUCKETS
#define KEY_BUCKETS 256
__device__ __forceinline__ int key_bucket(float nrm, float mx) {
    const int b = mx > 0.f ? (int)(nrm / mx * (float)KEY_BUCKETS) : 0;
    return KEY_BUCKETS - 1 - (b < 0 ? 0 : (b > KEY_BUCKETS - 1 ? KEY_BUCKETS - 1 : b));       // bucket 0: the largest norms
}

struct AttnArgs {
    const float *Xkp, *Xqp, *Wp;
    const __bf16 *Xkh;         // [Nk][DP] bf16(Xkp) (filter kernel)
    const _Float16 *Xks;       // [Nk][2 DP] two-half f16 split of kscale * Xkp: DP leading halves, DP scaled remainders
    const float *kscale;       // [1] power of two that brings the largest |key element| into [0.5, 1)
    const float *kmax;         // [1] largest key 2-norm (filter kernel)
    const float *knorm;        // [Nk] key 2-norms when the key lists are in descending-norm order, else nullptr
    float *st_val;             // filter kernel: top-(k+1) lists parked between key-chunk launches,
    int32_t *st_pos;           //   [block - blk_lo][k+1][FILT_NT threads]
    int key_lo, key_hi;        // key-list positions [key_lo, key_hi) streamed by this launch
    int32_t *cand_pos;         // screening -> exact pass: [block - blk_lo][FILT_CAPH][FILT_NT] key-list positions
    int32_t *cand_cnt;         //   [block - blk_lo][FILT_NT] candidates of the chunk (> FILT_CAPH = overflow)
    int32_t *st_ovf;           //   [block - blk_lo][FILT_NT] sticky overflow marker of the pair
    int blk_lo;                // first workgroup of this launch's group
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
    // window selection (k > DDG_MAX_TOPK_REG)
    float *wlo, *ww;           // [slots] lower edge and bin width of the pair's current bracket (WBINS bins)
    int32_t *wst;              // [slots] WST_* bits | selected bin << 8
    int32_t *nab;              // [slots] sample passes: sampled logits above the bracket; after MODE_WFILL: listed keys
    float *zab;                // [slots] MODE_WFILL: sum of the weights above the selected bin
    int32_t *lpos;             // [(block - blk_lo) * 128 + pair][wcap] key-list positions of the selected bin's keys
    float *lval;               //   ... and their logits
    int sample;                // 1: stream every (nk / WIN_SAMPLE)-th key only
    int wcap;                  // list capacity in use (<= WIN_CAP; tests shrink it to force the radix fallback)
};

// Stride of the sample passes over a column's key list.  The bracket the sample yields spans the sample ranks
// q S -+ (WIN_SIGMAS sqrt(S q (1 - q)) + 3), q = k / nk, i.e. a fraction ~ 12 sqrt(q (1 - q) / S) of the nk keys, which
// the full pass resolves into WBINS bins; the bin that holds the k-th largest logit is listed (<= WIN_CAP entries).  A fixed
// S = 8192 keeps ~ 5.5e-4 nk keys per bin: fine at 1M rows (440), past the list capacity from ~ 2M valid keys on (c4 has
// 5M).  So S grows with nk^2: keys per bin = 12 nk sqrt(q (1 - q)) / (WBINS sqrt(S)) stays near 450, and the three sample
// passes cost 3 S / nk of a full pass (c4 at top_k = 0.05: S = 460k of 5M keys, 28 %).
__device__ __forceinline__ int win_kstride(int nk_all, int k) {
    if (nk_all < 2 * WIN_SAMPLE) return 1;
    const float q = fminf((float)k / (float)nk_all, 1.f);
    const float r = (float)nk_all * sqrtf(q * (1.f - q)) * (1.f / WIN_BIN_KEYS);
    const float S = fmaxf(r * r, (float)WIN_SAMPLE);
    const int st = (int)((float)nk_all / S);
    return st < 1 ? 1 : st;
}

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

// Instrumentation builds (FILT_TIMING: s_memtime sums, FILT_STATS: event counters; neither changes a result) must be
// asked for with -DDDG_EXPERIMENTS as well, so that no such switch slips into a product build; the ablations that
// produced wrong results on purpose (timing of a loop without its loads / epilogue / survivor path) are gone.
#if (defined(FILT_TIMING) || defined(FILT_STATS)) && !defined(DDG_EXPERIMENTS)
#error "FILT_TIMING / FILT_STATS are instrumentation builds: add -DDDG_EXPERIMENTS"
#endif
#ifdef FILT_TIMING
__device__ unsigned long long g_filt_time[8];      // experiments only: s_memtime sums per wave of k_attn_screen
extern "C" int ddg_debug_filt_time(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_filt_time), 64) != hipSuccess) return DDG_E_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_filt_time), z, 64) != hipSuccess) return DDG_E_LAUNCH; }
    return DDG_OK;
}
#define TSTAMP() __builtin_amdgcn_s_memtime()
#endif
#ifdef FILT_STATS
__device__ unsigned long long g_filt_stats[4];     // experiments only
#endif
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

__device__ __forceinline__ int block_column(const int32_t *qb_start, int C, int b) {
    int lo = 0, hi = C;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (qb_start[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

// Epilogue shared by the selection kernels: merge the two half-wave lists of each query, then
// (lanes h == 0) flag a tie at the k-th logit, or softmax over the kept k entries and gather their
// column-c values: the fill without a second similarity pass (utils/util.py:463-496).
template <int KCAP>
__device__ __forceinline__ void merge_halves(const AttnArgs &p, float (&lst)[KCAP], int (&lpos)[KCAP]) {
    const int ndummy = KCAP - NLIST(p.k);
    float ov[KCAP];
    int op[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        ov[i] = __shfl_xor(lst[i], 32);
        op[i] = __shfl_xor(lpos[i], 32);
    }
#pragma unroll
    for (int i = 0; i < KCAP; ++i) list_insert_kv<KCAP>(lst, lpos, (i >= ndummy) ? ov[i] : -INFINITY, op[i]);
}

// lanes h == 0 hold the merged top-(k+2) list of their query: lst[KCAP-3] is the k-th largest logit (tau),
// lst[KCAP-2] the (k+1)-th, lst[KCAP-1] the (k+2)-th.  The reference keeps every key whose probability
// reaches the k-th largest (utils/util.py:470-476), so a key tied with the k-th is kept as well.
template <int KCAP>
__device__ __forceinline__ void topk_output(const AttnArgs &p, float (&lst)[KCAP], int (&lpos)[KCAP], int64_t slot,
                                            int qrow, int nk, const int32_t *kl, int off_c, int d_c, int DP, int h,
                                            bool force_flag = false) {
    const int ndummy = KCAP - NLIST(p.k);
    if (h == 0 && qrow >= 0) {
        const bool tie = nk > p.k && lst[KCAP - 3] == lst[KCAP - 2];
        // a longer tie may go on past the end of the list
        const bool open = force_flag || (tie && nk > p.k + 1 && lst[KCAP - 2] == lst[KCAP - 1]);
        const int last = tie ? KCAP - 2 : KCAP - 3;             // last kept entry
        p.tau[slot] = (nk >= p.k) ? lst[KCAP - 3] : -INFINITY;
        float vmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < KCAP - 1; ++i)
            if (i >= ndummy && i <= last) vmax = fmaxf(vmax, lst[i]);
        p.smax[slot] = vmax;
        p.flag[slot] = open ? 1 : 0;
#ifdef FILT_STATS
        if (open && atomicAdd(&g_filt_stats[3], 1ull) < 6) printf("tie: qrow %d nk %d k %d vals %g %g %g\n", qrow, nk, p.k, lst[KCAP - 3], lst[KCAP - 2], lst[KCAP - 1]);
#endif
        if (!open) {
            float wgt[KCAP];
            float Zs = 0.f;
#pragma unroll
            for (int i = 0; i < KCAP - 1; ++i) {
                const bool kept = i >= ndummy && i <= last && lpos[i] >= 0;
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
}

template <int KCAP>
__device__ __forceinline__ void topk_finish(const AttnArgs &p, float (&lst)[KCAP], int (&lpos)[KCAP], int64_t slot,
                                            int qrow, int nk, const int32_t *kl, int off_c, int d_c, int DP, int h) {
    merge_halves<KCAP>(p, lst, lpos);
    topk_output<KCAP>(p, lst, lpos, slot, qrow, nk, kl, off_c, d_c, DP, h);
}

// SPLIT (window passes only): the logits come from the f16 MFMA on operands split in two halves,
//   x = hi + 2^-11 lo,  hi = f16(x), lo = f16(2^11 (x - hi))      (both operands pre-scaled by powers of two into f16 range)
//   x y ~ hi_x hi_y + 2^-11 (lo_x hi_y + hi_x lo_y)               (the 2^-22 lo_x lo_y term is dropped; two accumulators)
// three v_mfma_f32_32x32x16_f16 per 16-deep step instead of eight v_mfma_f32_32x32x2_f32 (a sixteenth of the matrix
// time each), fp32 accumulation.  |error| <= 3 * 2^-22 sum_d |q_d||k_d| + fp32 accumulation, the class of the fp32
// chain's own rounding; the key tiles keep their byte size (4 bytes per element), so the staging is shared.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// VALU instructions scheduled behind each MFMA of the pipelined loop (per-logit work of a tile / MFMAs of a tile)
#ifndef PIPE_VALU_WIN
#define PIPE_VALU_WIN 5
#define PIPE_VALU_ONE 7
#define PIPE_VALU_BLK 10
#endif
#define PIPE_VALU(MODE, ONE) ((MODE) == MODE_WIN ? PIPE_VALU_WIN : (ONE) ? PIPE_VALU_ONE : PIPE_VALU_BLK)
// PIPE (SPLIT only; 0 = off): the key stream is software-pipelined one tile deep, the MFMAs of tile t+1 interleaved with the
// per-logit work of tile t in one basic block; PIPE - 1 = trailing 16-deep steps of the row stride that hold only zero
// padding (not multiplied).  Same products in the same order as PIPE = 0: bit-identical logits.
template <int KS4, int MODE, int KCAP, bool SPLIT = false, int PIPE = 0>
__global__ __launch_bounds__(256, (KS4 >= 32 ? 1 : 2)) void k_attn(AttnArgs p) {
    static_assert(!SPLIT || MODE == MODE_WIN || MODE == MODE_WFILL, "split arithmetic: window passes only");
    static_assert(PIPE == 0 || (SPLIT && PIPE - 1 < KS4 / 2), "pipelined loop: split window passes only");
    constexpr int DP = KS4 * 8;
    constexpr int KS = KS4 / 2;            // 16-deep steps (SPLIT)
    constexpr int XOFF = (SPLIT && MODE == MODE_WFILL) ? 768 : 0;      // floats of the three fp32 side tiles (MODE_WFILL)
    constexpr int LSTR = DP + 4;
    constexpr int NPF = KS4 / 4;           // float4 prefetch registers per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *kbuf0 = smem;
    float *kbuf1 = smem + TK * LSTR;
    float *extra = smem + 2 * TK * LSTR;   // MODE_HIST: histograms; MODE_FILL: accumulators
    __shared__ int s_col;

    const int nblocks = p.qb_start[p.C];
    const int blk = p.blk_lo + (int)blockIdx.x;
    if (blk >= nblocks) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    if (tid == 0) s_col = block_column(p.qb_start, p.C, blk);
    __syncthreads();
    const int c = s_col;
    const int nk_all = p.nk[c];
    // sample passes stream every kstride-th position of the key list
    const int kstride = (MODE == MODE_WIN && p.sample) ? win_kstride(nk_all, p.k) : 1;
    const int nk = (nk_all + kstride - 1) / kstride;
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    const int off_c = p.offsets[c];
    const int d_c = p.offsets[c + 1] - off_c;

    const int64_t slot = ((int64_t)blk * 4 + wid) * 32 + r32;
    const int qrow = p.qlist[slot];

    // ---- query operand in registers: breg[4u+s] = W[c][k] * x_q[k], k = 8u + 4h + s
    f16x8 bh[SPLIT ? KS : 1], bl[SPLIT ? KS : 1];
    float qinv = 1.f;                      // SPLIT: logit = accumulator * qinv
    const int ksc = (p.D + 15) >> 4;       // SPLIT: 16-deep steps that hold data
    if constexpr (SPLIT) {
        // q_d = W[c][d] x_q[d] for d = 16 u + 8 h + j, scaled by a power of two so that max |q_d| lies in [2^13, 2^14)
        const float *wq = p.Wp + (int64_t)c * DP + 8 * h;
        const float *xq = p.Xqp + (int64_t)(qrow < 0 ? 0 : qrow) * DP + 8 * h;
        float qmax = 0.f;
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + 16 * u + 4 * j4);
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xq + 16 * u + 4 * j4);
#pragma unroll
                for (int j = 0; j < 4; ++j) qmax = fmaxf(qmax, fabsf(wv[j] * xv[j]));
            }
        qmax = fmaxf(qmax, __shfl_xor(qmax, 32));
        int eq = 0;
        if (qmax > 0.f && qrow >= 0) frexpf(qmax, &eq); else eq = 14;
        const float qsc = ldexpf(1.f, 14 - eq);
        qinv = ldexpf(1.f, eq - 14) / p.kscale[0];
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + 16 * u + 4 * j4);
                f32x4 xv = *reinterpret_cast<const f32x4 *>(xq + 16 * u + 4 * j4);
                if (qrow < 0) xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q = wv[j] * xv[j] * qsc;
                    const _Float16 qh = (_Float16)q;
                    bh[u][4 * j4 + j] = qh;
                    bl[u][4 * j4 + j] = (_Float16)((q - (float)qh) * 2048.f);
                }
            }
    }
    float breg[SPLIT ? 1 : KS4 * 4];
    if constexpr (!SPLIT) {
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
        const int ndummy = KCAP - NLIST(p.k);
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
    // window passes: tt = fma(v, winv, nwl) is the bin coordinate of a logit inside the pair's bracket (the same
    // expression in both full passes, so a logit falls into the same bin in both)
    float winv = 0.f, nwl = 0.f, smin = INFINITY, fb = INFINITY, fb1 = INFINITY, nsmax2 = 0.f, kinv = 1.f;
    float accr[8];
    int wst = 0;
    if (MODE == MODE_WIN) {
        uint32_t *hist = reinterpret_cast<uint32_t *>(extra);
        for (int i = tid; i < QPB * WHSTRIDE; i += 256) hist[i] = 0;
        if (p.level != WIN_MINMAX) {
            winv = 1.f / p.ww[slot];
            nwl = -p.wlo[slot] * winv;
        }
    }
    if (MODE == MODE_WFILL) {
        int *lcur = reinterpret_cast<int *>(extra);                 // [QPB] list cursors, then the accumulators
        for (int i = tid; i < QPB; i += 256) lcur[i] = 0;
        winv = 1.f / p.ww[slot];
        nwl = -p.wlo[slot] * winv;
        wst = p.wst[slot];
        smax = p.smax[slot];
        nsmax2 = -smax * 1.44269504088896340736f;
        if (SPLIT) kinv = 1.f / p.kscale[0];
        active = qrow >= 0 && p.flag[slot] == 0;
        if (active) {
            // selected bin b: logits with tt >= b + 1 lie above it, b <= tt < b + 1 inside.  A bracket without an upper
            // (lower) edge counted everything above (below) it into bin 127 (0).  WST_ALL: nwl = +inf, all kept.
            const int b = wst >> 8;
            fb = ((wst & WST_OPENBOT) && b == 0) ? -INFINITY : (float)b;
            fb1 = ((wst & WST_OPENTOP) && b == WBINS - 1) ? INFINITY : (float)(b + 1);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) accr[e] = 0.f;
        if (d_c > 8) {
            float *accb = extra + QPB + XOFF + (int64_t)tid * p.dcs;
            for (int e = 0; e < d_c; ++e) accb[e] = 0.f;
        }
    }

    // ---- two-level accumulation of the fill sums (MODE_WFILL, MODE_FILL).  A lane adds up to nk / 2 weights of 0.95 .. 1
    // one after the other; once the running sum passes ~ 1e5 its ulp is no longer small against the SPREAD of the addends
    // and round-to-nearest stops being unbiased: measured +3e-4 (k = 320 000) to +1e-3 (k = 640 000) relative error of the
    // weight sum Z, i.e. of every filled value, against 1e-7 for a pairwise fp32 sum (the reference's GEMM).  So the registers
    // (and the LDS rows of wide column blocks) only hold the sums of FLUSH_TILES tiles (1 024 addends per lane: their ulp stays 400x below the spread); they are then added to a second level
    // -- out_code itself and zab / a register for Z -- whose addends vary by far more than its ulp.
    // Key lists below 65 536 entries keep ONE level (their sums stay far below the biased regime, and a changed summation
    // order would move the exactly tied categorical votes of small tables, whose arg-max the reference's rounding decides:
    // tests/test_gpu_e2e.py pins those).
    constexpr int FLUSH_TILES = 64;
    const bool two_level = nk >= 65536;
    float Z2 = 0.f;                                           // MODE_FILL: second level of Z
    if constexpr (MODE == MODE_WFILL || MODE == MODE_FILL) {
        const bool on = (MODE == MODE_WFILL) ? active : qrow >= 0;
        if (h == 0 && on) {
            for (int e = 0; e < d_c; ++e) p.out_code[(int64_t)qrow * p.D + off_c + e] = 0.f;
            if (MODE == MODE_WFILL) p.zab[slot] = 0.f;
        }
    }
    auto flush = [&](auto onec) {
        constexpr bool ONE = decltype(onec)::value;            // only accr[0] / acc1 is live
        if constexpr (MODE == MODE_WFILL || MODE == MODE_FILL) {
            // opaque copies: nothing derived from them (addresses, predicates) can be hoisted out of the key loop, whose
            // register budget has no room for it
            int q_ = qrow, tid_ = tid;
            uint32_t slo_ = (uint32_t)slot, shi_ = (uint32_t)((uint64_t)slot >> 32);
            asm volatile("" : "+v"(q_), "+v"(tid_), "+v"(slo_), "+v"(shi_));
            const int64_t slot_ = (int64_t)(((uint64_t)shi_ << 32) | slo_);
            const bool on = (tid_ & 32) == 0 && ((MODE == MODE_WFILL) ? active : q_ >= 0);
            float *dst = p.out_code + (int64_t)(q_ < 0 ? 0 : q_) * p.D + off_c;
            if (MODE == MODE_FILL ? d_c == 1 : d_c <= 8) {
#pragma unroll
                for (int e = 0; e < (ONE || MODE == MODE_FILL ? 1 : 8); ++e) {
                    float &r = (MODE == MODE_FILL) ? acc1 : accr[e];
                    const float a = r + __shfl_xor(r, 32);
                    if (on && e < d_c) dst[e] += a;
                    r = 0.f;
                }
            } else {
                float *accb = extra + (MODE == MODE_WFILL ? QPB + XOFF : 0) + (int64_t)tid_ * p.dcs;
                __syncthreads();
                if (on) {
                    const float *a1 = accb + (int64_t)32 * p.dcs;
                    for (int e = 0; e < d_c; ++e) dst[e] += accb[e] + a1[e];
                }
                __syncthreads();
                for (int e = 0; e < d_c; ++e) accb[e] = 0.f;
            }
            if constexpr (MODE == MODE_WFILL) {
                const float Zt = Z + __shfl_xor(Z, 32);
                if (on) p.zab[slot_] += Zt;
            } else {
                Z2 += Z;
            }
            Z = 0.f;
        }
    };

    // ---- key streaming
    const int st_row = tid >> 3, st_c = tid & 7;
    f32x4 pf[NPF];
    bool pfx_ok = false;
    float pfx = 0.f;                       // SPLIT + MODE_WFILL: fp32 value st_c of the key row's column block (side tile)
    float *xs0 = extra + QPB, *xs1 = extra + QPB + 256;       // (a third one follows: pipelined loop)
    const float *ksrc = SPLIT ? reinterpret_cast<const float *>(p.Xks) : p.Xkp;     // rows of DP * 4 bytes either way
    auto prefetch = [&](int t) {
        const int pos = t * TK + st_row;
        const int kidx = pos < nk ? kl[(int64_t)pos * kstride] : -1;
        const float *src = ksrc + (int64_t)(kidx < 0 ? 0 : kidx) * DP + st_c * 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            pf[i] = *reinterpret_cast<const f32x4 *>(src + 32 * i);      // rows past the list: zeroed at commit time, so that
        }                                                                //   nothing waits for the loads here
        pfx_ok = kidx >= 0;
        // unconditional load (clamped address): a predicated one makes the compiler wait for it right here
        if (SPLIT && MODE == MODE_WFILL) pfx = p.Xkp[(int64_t)(kidx < 0 ? 0 : kidx) * DP + off_c + (st_c < d_c ? st_c : 0)];
    };
    auto commit = [&](float *buf, float *xsd) {
        float *dst = buf + st_row * LSTR + st_c * 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) *reinterpret_cast<f32x4 *>(dst + 32 * i) = pfx_ok ? pf[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (SPLIT && MODE == MODE_WFILL) xsd[tid] = (pfx_ok && st_c < d_c) ? pfx : 0.f;
    };

    // the similarity of one 32-key tile against the wave's 32 queries
    auto mma = [&](const float *cur) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        const float *Ab = cur + r32 * LSTR + 4 * h;
        if constexpr (SPLIT) {
            // two independent chains: leading products, and the cross products that carry a factor 2^-11
            f32x16 accx;
#pragma unroll
            for (int g = 0; g < 16; ++g) accx[g] = 0.f;
            const float qinvx = qinv * 0x1p-11f;
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                if (u >= KS - 3 && u >= ksc) continue;     // steps that hold only the zero padding of the row (uniform)
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(Ab + 8 * u);            // halves 16 u + 8 h ..
                const f16x8 al = *reinterpret_cast<const f16x8 *>(Ab + DP / 2 + 8 * u);   // their remainders
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], accx, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = fmaf(accx[g], qinvx, acc[g] * qinv);
        } else {
#pragma unroll
            for (int u = 0; u < KS4; ++u) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Ab + 8 * u);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], breg[4 * u + s], acc, 0, 0, 0);
            }
        }
        return acc;
    };
    // per-logit work of a mode on the tile's 32 x 32 logits (xst: fp32 side tile of the key rows' column block)
    auto epi = [&](f32x16 &acc, int t, const float *cur, const float *xst) {
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
        } else if (MODE == MODE_WIN) {
            // branch-free: every logit adds one to a counter of its query's row -- [0] below the bracket, [1 + bin],
            // [1 + WBINS] above it (rows of the last tile past the key list count as "below": taken out at the end)
            uint32_t *hist = reinterpret_cast<uint32_t *>(extra) + (wid * 32 + r32) * WHSTRIDE;
            if (!full) {
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    if ((g & 3) + 8 * (g >> 2) + 4 * h >= rem) acc[g] = (p.level == WIN_MINMAX) ? NAN : -INFINITY;
            }
            if (p.level == WIN_MINMAX) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    smax = fmaxf(smax, acc[g]);       // fmaxf / fminf drop a NaN operand
                    smin = fminf(smin, acc[g]);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float v = acc[g];
                    if (p.level == WIN_FULL) smax = fmaxf(smax, v);
                    int idx = (int)floorf(fmaf(v, winv, nwl));
                    idx = min(max(idx, -1), WBINS) + 1;
                    atomicAdd(&hist[idx], 1u);
                }
            }
        } else if (MODE == MODE_WFILL) {
            int *lcur = reinterpret_cast<int *>(extra);
            if (!full) {
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    if ((g & 3) + 8 * (g >> 2) + 4 * h >= rem) acc[g] = NAN;           // tt = NaN: neither kept nor listed
            }
            // The common path is straight-line code per column class (the class is workgroup-uniform): weights of all 16
            // logits, accumulation, and ONE wave-wide test per tile whether any logit fell into the selected bin.
            const float *xsb = SPLIT ? xst : cur + off_c;    // fp32 values of the key rows' column block
            constexpr int XSTR = SPLIT ? 8 : LSTR;
            unsigned inbm = 0;
            auto weight = [&](int g) {
                const float v = acc[g];
                const float tt = fmaf(v, winv, nwl);
                const bool keep = tt >= fb1;
                const float w = keep ? __builtin_amdgcn_exp2f(fmaf(v, 1.44269504088896340736f, nsmax2)) : 0.f;
                Z += w;
                inbm |= (tt >= fb && !keep) ? (1u << g) : 0u;
                return w;
            };
            if (d_c == 1) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                    accr[0] += weight(g) * xsb[kr * XSTR];
                }
            } else {
                float *accb = extra + QPB + XOFF + (int64_t)tid * p.dcs;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                    const float w = weight(g);
                    if (d_c <= 8) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) accr[e] += w * xsb[kr * XSTR + e];   // e >= d_c: zeros / finite neighbours, never read back
                    } else if (__any(w != 0.f)) {
                        if constexpr (SPLIT) {       // blocks wider than the side tile: rebuilt from the two halves
                            const _Float16 *hr = reinterpret_cast<const _Float16 *>(cur + kr * LSTR) + off_c;
                            for (int e = 0; e < d_c; ++e)
                                accb[e] += w * (((float)hr[e] + (float)hr[DP + e] * 0x1p-11f) * kinv);
                        } else {
                            for (int e = 0; e < d_c; ++e) accb[e] += w * xsb[kr * XSTR + e];
                        }
                    }
                }
            }
            if (__any(inbm != 0)) {                  // rare: keys inside the selected bin are listed for k_attn_wlist
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (inbm & (1u << g)) {
                        const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                        const int i = atomicAdd(&lcur[wid * 32 + r32], 1);
                        if (i < p.wcap) {
                            const int64_t o = (((int64_t)blockIdx.x * QPB) + wid * 32 + r32) * p.wcap + i;
                            p.lpos[o] = t * TK + kr;
                            p.lval[o] = acc[g];
                        }
                    }
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
    };

    const int ntk = (nk + TK - 1) / TK;
    if constexpr (PIPE == 0) {
        if (ntk > 0) { prefetch(0); commit(kbuf0, xs0); }
        __syncthreads();
    }

    if constexpr (PIPE > 0) {
        // Software pipeline one tile deep.  The key rows of tile t+2 are written over tile t's buffer, which only the MFMAs
        // of the previous iteration read; the side tiles rotate over three.  The accumulator pair of tile t+1 (leading and
        // cross products) is combined into logits at the end of the iteration.  The last tile (the only one that can be
        // partial) takes the plain path.  Host: not for column blocks wider than 8 (their accumulate step reads the key tile)
        // and not for the WIN_MINMAX pass.
        constexpr int KSC = KS - (PIPE - 1);
        constexpr int NM = 3 * KSC;                 // MFMAs per tile
        // Key rows go straight from global memory into LDS (LDS-DMA: no staging registers, no ds_write).  A DMA piece is 64
        // lanes x 16 bytes, lane-linear in LDS, so rows are dense (DP * 4 bytes: a multiple of 64 banks) and bank conflicts
        // of the fragment reads are removed by an XOR swizzle of the 16-byte chunk position with key = row & 15, applied
        // to the SOURCE address of the DMA and to the ds_read address.  Wave w fills the 1 KiB pieces w*NPC .. of a tile =
        // its rows 8 w .. 8 w + 7: one key-list load per wave and tile (lane l: row l & 7), issued a whole tile before the
        // rows that depend on it, handed to the lanes by ds_bpermute -- the same index serves the side tile's gather.
        // Rows past the end of the list re-read its last key (the last tile's per-logit work masks them).
        constexpr int RB = DP * 4;                  // bytes per key row
        constexpr int CPR = DP / 4;                 // 16-byte chunks per row
        constexpr int NPC = RB / 128;               // pieces per wave and tile
        static_assert(NPC * 1024 == 8 * RB && CPR % 16 == 0, "a wave's pieces cover its eight rows");
        int kix = 0;
        uint32_t usrc[NPC];
        int urow4[NPC];
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int n = i * 64 + lane;
            const int rrow = n / CPR, pos = n - rrow * CPR;
            const int key = (wid * 8 + rrow) & 15;
            urow4[i] = rrow * 4;
            usrc[i] = (uint32_t)(((pos & ~15) | ((pos ^ key) & 15)) * 16);
        }
        auto load_kix = [&](int t) {
            int pos = t * TK + wid * 8 + (lane & 7);
            pos = pos < nk ? pos : nk - 1;
            kix = kl[(int64_t)pos * kstride];
        };
        auto dma_tile = [&](float *buf) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const uint32_t kid = (uint32_t)__builtin_amdgcn_ds_bpermute(urow4[i], kix);
                const char *src = reinterpret_cast<const char *>(ksrc) + (uint64_t)kid * RB + usrc[i];
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(src)),
                    reinterpret_cast<__attribute__((address_space(3))) void *>(
                        (uint32_t)(uintptr_t)(reinterpret_cast<char *>(buf) + (wid * NPC + i) * 1024)), 16, 0, 0);
            }
        };
        // side tile (MODE_WFILL): thread (st_row, st_c) gathers fp32 value st_c of key row st_row's column block
        auto side_prefetch = [&](int t) {
            if constexpr (MODE == MODE_WFILL) {
                const uint32_t kid = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 3) * 4, kix);
                pfx_ok = t * TK + st_row < nk;
                pfx = p.Xkp[(int64_t)kid * DP + off_c + (st_c < d_c ? st_c : 0)];
            }
        };
        auto side_commit = [&](float *xsd) {
            if constexpr (MODE == MODE_WFILL) xsd[tid] = (pfx_ok && st_c < d_c) ? pfx : 0.f;
        };
        // fragment u' (chunk 2 u' + h) of tile row r32: position (2 u' + h) ^ key; eight lane offsets cover the low bits
        int foff[8];
        {
            const int hk = h ^ (r32 & 15);
#pragma unroll
            for (int m = 0; m < 8; ++m) foff[m] = r32 * RB + ((2 * m) ^ hk) * 16;
        }
        auto frag = [&](const float *buf, int up) {
            return *reinterpret_cast<const f16x8 *>(reinterpret_cast<const char *>(buf) + foff[up & 7] + 32 * (up & ~7));
        };
        if (ntk > 0) {
            load_kix(0);
            dma_tile(kbuf0);
            side_prefetch(0);
            load_kix(1);
            dma_tile(kbuf1);
            side_commit(xs0);
            side_prefetch(1);
            load_kix(2);
            side_commit(xs1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f32x16 pa, px;
            {
#pragma unroll
                for (int g = 0; g < 16; ++g) { pa[g] = 0.f; px[g] = 0.f; }
#pragma unroll
                for (int u = 0; u < KSC; ++u) {
                    const f16x8 ah = frag(kbuf0, u);
                    const f16x8 al = frag(kbuf0, DP / 16 + u);
                    pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], pa, 0, 0, 0);
                    px = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], px, 0, 0, 0);
                    px = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], px, 0, 0, 0);
                }
            }
            const float qinvx = qinv * 0x1p-11f;
#pragma unroll
            for (int g = 0; g < 16; ++g) pa[g] = fmaf(px[g], qinvx, pa[g] * qinv);      // the logits of tile 0
            __syncthreads();                        // every wave has read tile 0: its buffer takes tile 2
            int x_cur = 0, x_new = 512;             // side tile of tile t, of tile t+2 (floats past xs0)
            auto piped = [&](auto onec) {
                constexpr bool ONE = decltype(onec)::value;      // MODE_WFILL: column blocks of one element
                // (segments of FLUSH_TILES tiles, the second accumulation level between them: outside the inner loop, whose
                // register budget it must not touch)
                for (int tb = 0; tb + 1 < ntk; tb += FLUSH_TILES) {
                const int te = min(tb + FLUSH_TILES, ntk - 1);
                for (int t = tb; t < te; ++t) {
                    float *cur = (t & 1) ? kbuf1 : kbuf0;
                    float *nxt = (t & 1) ? kbuf0 : kbuf1;
                    // tile t+2 (rows and side values; past the end: copies of the last key), key-list indices of tile t+3
                    dma_tile(cur);
                    side_prefetch(t + 2);
                    load_kix(t + 3);
                    f32x16 na, nx;
#pragma unroll
                    for (int g = 0; g < 16; ++g) { na[g] = 0.f; nx[g] = 0.f; }
                    const float *xsb = xs0 + x_cur;
                    unsigned inbm = 0;
                    // ---- one basic block, written out slot by slot: each MFMA of tile t+1 is followed by the fragment reads
                    // of the next step or by the whole work of one logit of tile t (its side-tile values are read one logit
                    // ahead); sched_barrier keeps the slots as written
                    f16x8 ah = frag(nxt, 0), al = frag(nxt, DP / 16);
                    f16x8 ahn = ah, aln = al;
                    [[maybe_unused]] f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = xa;          // side-tile values of the logit in work
                    [[maybe_unused]] uint32_t *hist = reinterpret_cast<uint32_t *>(extra) + (wid * 32 + r32) * WHSTRIDE;
                    auto side = [&](int g) {
                        const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                        if constexpr (MODE == MODE_WFILL) {
                            if constexpr (ONE) {
                                xa[0] = xsb[kr * 8];
                            } else {
                                xa = *reinterpret_cast<const f32x4 *>(xsb + kr * 8);
                                xb = *reinterpret_cast<const f32x4 *>(xsb + kr * 8 + 4);
                            }
                        }
                    };
                    side(0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int u = m / 3, r = m % 3;
                        if (r == 0) na = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], na, 0, 0, 0);
                        else if (r == 1) nx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], nx, 0, 0, 0);
                        else nx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], nx, 0, 0, 0);
                        if (r == 0 && u + 1 < KSC) {
                            ahn = frag(nxt, u + 1);
                            aln = frag(nxt, DP / 16 + u + 1);
                        }
                        if (r == 2) { ah = ahn; al = aln; }
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            if ((g * NM) / 16 != m) continue;
                            float v = pa[g];
                            asm volatile("" : "+v"(v));       // the logit's work starts in this slot, not earlier
                            if constexpr (MODE == MODE_WIN) {
                                smax = fmaxf(smax, v);
                                int idx = (int)floorf(fmaf(v, winv, nwl));
                                idx = min(max(idx, -1), WBINS) + 1;
                                atomicAdd(&hist[idx], 1u);
                            } else {
                                const f32x4 ya = xa, yb = xb;
                                if (g + 1 < 16) side(g + 1);
                                const float tt = fmaf(v, winv, nwl);
                                const bool keep = tt >= fb1;
                                const float w = keep ? __builtin_amdgcn_exp2f(fmaf(v, 1.44269504088896340736f, nsmax2)) : 0.f;
                                Z += w;
                                inbm |= (tt >= fb && !keep) ? (1u << g) : 0u;
                                if constexpr (ONE) {
                                    accr[0] += w * ya[0];
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { accr[e] += w * ya[e]; accr[4 + e] += w * yb[e]; }
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // (the accumulations stay in this block: without the pins they are sunk past the commit below)
                    if constexpr (MODE == MODE_WIN) asm volatile("" : "+v"(smax));
                    else asm volatile("" : "+v"(Z), "+v"(inbm));
                    if constexpr (MODE == MODE_WFILL) {
#pragma unroll
                        for (int e = 0; e < (ONE ? 1 : 8); ++e) asm volatile("" : "+v"(accr[e]));
                    }
                    if constexpr (MODE == MODE_WFILL) {
                        if (__any(inbm != 0)) {              // rare: keys inside the selected bin are listed for k_attn_wlist
                            int *lcur = reinterpret_cast<int *>(extra);
#pragma unroll
                            for (int g = 0; g < 16; ++g) {
                                if (inbm & (1u << g)) {
                                    const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
                                    const int i = atomicAdd(&lcur[wid * 32 + r32], 1);
                                    if (i < p.wcap) {
                                        const int64_t o = (((int64_t)blockIdx.x * QPB) + wid * 32 + r32) * p.wcap + i;
                                        p.lpos[o] = t * TK + kr;
                                        p.lval[o] = pa[g];
                                    }
                                }
                            }
                        }
                    }
                    side_commit(xs0 + x_new);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t+2 are in LDS
                    __syncthreads();
#pragma unroll
                    for (int g = 0; g < 16; ++g) pa[g] = fmaf(nx[g], qinvx, na[g] * qinv);      // the logits of tile t+1
                    x_cur = x_cur == 512 ? 0 : x_cur + 256;
                    x_new = x_new == 512 ? 0 : x_new + 256;
                }
                if constexpr (MODE == MODE_WFILL) {
                    if (two_level) flush(onec);
                }
                }
            };
            if (MODE == MODE_WFILL && d_c == 1) {
                piped(std::true_type{});
#pragma unroll
                for (int e = 1; e < 8; ++e) accr[e] = 0.f;      // (dead inside the loop: seven registers less)
            } else {
                piped(std::false_type{});
            }
            epi(pa, ntk - 1, ((ntk - 1) & 1) ? kbuf1 : kbuf0, xs0 + x_cur);
        }
    } else {
        for (int tb = 0; tb < ntk; tb += FLUSH_TILES) {
            const int te = min(tb + FLUSH_TILES, ntk);
            for (int t = tb; t < te; ++t) {
                float *cur = (t & 1) ? kbuf1 : kbuf0;
                float *nxt = (t & 1) ? kbuf0 : kbuf1;
                if (t + 1 < ntk) prefetch(t + 1);
                f32x16 acc = mma(cur);
                epi(acc, t, cur, (t & 1) ? xs1 : xs0);
                if (t + 1 < ntk) commit(nxt, (t & 1) ? xs0 : xs1);
                __syncthreads();
            }
            if (two_level) flush(std::false_type{});
        }
    }

    // ---- per-mode finish
    if constexpr (MODE == MODE_TOPK) {
        topk_finish<KCAP>(p, lst, lpos, slot, qrow, nk, kl, off_c, d_c, DP, h);
    } else if (MODE == MODE_WIN) {
        uint32_t *hist = reinterpret_cast<uint32_t *>(extra);
        smax = fmaxf(smax, __shfl_xor(smax, 32));
        smin = fminf(smin, __shfl_xor(smin, 32));
        if (p.level == WIN_MINMAX) {
            if (h == 0) {
                // the whole sampled range as the first bracket; nk_all < k keeps every key (wlo = -inf puts them all above)
                const bool all = nk_all < p.k;
                p.wlo[slot] = all ? -INFINITY : smin;
                p.ww[slot] = all ? 1.f : fmaxf((smax - smin) * (1.f / WBINS), 1e-30f);
                p.wst[slot] = all ? WST_ALL : 0;
                p.flag[slot] = 0;
                p.rank[slot] = 0;
            }
        } else {
            if (p.level == WIN_FULL && h == 0) p.smax[slot] = smax;
            __syncthreads();   // all histogram atomics of the block are done
            if (tid < QPB) {
                const int64_t sl = (int64_t)blk * QPB + tid;
                const uint32_t *hrow = hist + tid * WHSTRIDE;
                const uint32_t *hq = hrow + 1;
                const int st = p.wst[sl];
                if (!(st & WST_ALL) && p.qlist[sl] >= 0) {
                    const float lo = p.wlo[sl], w = p.ww[sl];
                    const int below = (int)hrow[0] - (ntk * TK - nk), above = (int)hrow[1 + WBINS];
                    if (p.level == WIN_SHIST) {
                        // sample ranks that bracket the population's k-th largest with WIN_SIGMAS of slack
                        const float q = (float)p.k / (float)nk_all;
                        const float rs = q * (float)nk;
                        const float gs = WIN_SIGMAS * sqrtf((float)nk * q * (1.f - q)) + 3.f;
                        const int tgt_hi = (int)floorf(rs - gs);     // at most this many sampled logits above the bracket
                        const int tgt_lo = (int)ceilf(rs + gs);      // at least this many at or above its lower edge
                        int nst = st, ab = above, b_hi = WBINS - 1;
                        if ((st & WST_OPENTOP) || tgt_hi < ab) {
                            nst |= WST_OPENTOP;
                        } else {
                            while (b_hi > 0 && ab + (int)hq[b_hi] <= tgt_hi) ab += (int)hq[b_hi--];
                        }
                        int cum = (nst & WST_OPENTOP) ? above : ab, b_lo = b_hi;
                        for (;; --b_lo) {
                            cum += (int)hq[b_lo];
                            if (cum >= tgt_lo || b_lo == 0) break;
                        }
                        if ((st & WST_OPENBOT) || cum < tgt_lo) nst |= WST_OPENBOT;
                        p.wlo[sl] = lo + (float)b_lo * w;
                        p.ww[sl] = fmaxf((float)(b_hi - b_lo + 1) * w * (1.f / WBINS), 1e-30f);
                        p.wst[sl] = nst;
                    } else {
                        // the bin of the k-th largest logit and its rank inside the bin; a bracket without an upper
                        // (lower) edge takes everything above (below) into its last (first) bin
                        const int top_extra = (st & WST_OPENTOP) ? above : 0;
                        const int bot_extra = (st & WST_OPENBOT) ? below : 0;
                        const int r = p.k - ((st & WST_OPENTOP) ? 0 : above);
                        int cum = 0, b = WBINS - 1, cnt = 0;
                        bool found = false;
                        if (r > 0) {
                            for (; b >= 0; --b) {
                                cnt = (int)hq[b] + (b == WBINS - 1 ? top_extra : 0) + (b == 0 ? bot_extra : 0);
                                if (cum + cnt >= r) { found = true; break; }
                                cum += cnt;
                            }
                        }
                        if (found && cnt <= p.wcap) {
                            p.wst[sl] = (st & 0xff) | (b << 8);
                            p.rank[sl] = r - cum;
                            p.flag[sl] = 0;
                        } else {
                            p.flag[sl] = 1;      // bracket missed, or too many keys in the bin: radix path
                            p.rank[sl] = 0;
                        }
                    }
                }
            }
        }
    } else if (MODE == MODE_WFILL) {
        int *lcur = reinterpret_cast<int *>(extra);
        // un-normalised sums of everything above the selected bin, in out_code / zab: the last partial level joins them
        // here; k_attn_wlist adds the bin's kept keys and divides
        flush(std::false_type{});
        __syncthreads();
        if (h == 0 && active) p.nab[slot] = lcur[wid * 32 + r32];
    } else if (MODE == MODE_HIST) {
        smax = fmaxf(smax, __shfl_xor(smax, 32));
        if (p.level == 0 && h == 0) p.smax[slot] = smax;
        __syncthreads();   // all histogram atomics of the block are done
        if (tid < QPB) {
            const int64_t sl = (int64_t)blk * QPB + tid;
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
        flush(std::false_type{});
        const float Zt = Z2 + __shfl_xor(Z2, 32);
        if (h == 0 && qrow >= 0)
            for (int e = 0; e < d_c; ++e) {
                float *dst = p.out_code + (int64_t)qrow * p.D + off_c + e;
                *dst = (Zt > 0.f) ? *dst / Zt : 0.f;
            }
        if (h == 0 && qrow >= 0) {
            if (p.out_tau) p.out_tau[(int64_t)qrow * p.C + c] = tau;
            if (p.out_nvalid) p.out_nvalid[(int64_t)qrow * p.C + c] = nk;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- window selection, last step: the keys MODE_WFILL listed for a pair (those of the bin that holds its k-th largest
// logit, with their exact logits) are ranked against each other; tau is the rank-th largest of them (every key tied
// with it is kept, utils/util.py:470-476), the kept ones join the sums MODE_WFILL left in out_code / zab, and the fill
// is normalised.  One wave per pair at a time, 32 pairs per wave.
__global__ __launch_bounds__(256) void k_attn_wlist(AttnArgs p) {
    __shared__ float s_val[4][WIN_CAP];
    __shared__ int s_pos[4][WIN_CAP];
    const int nblocks = p.qb_start[p.C];
    const int blk = p.blk_lo + (int)blockIdx.x;
    if (blk >= nblocks) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = block_column(p.qb_start, p.C, blk);
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    const int off_c = p.offsets[c];
    const int d_c = p.offsets[c + 1] - off_c;
    float *val = s_val[wid];
    int *pos = s_pos[wid];
    for (int pr = 0; pr < 32; ++pr) {
        const int64_t slot = (int64_t)blk * QPB + wid * 32 + pr;
        const int qrow = p.qlist[slot];
        const bool on = qrow >= 0 && p.flag[slot] == 0;
        const int n = on ? min(p.nab[slot], p.wcap) : 0;
        const int r = on ? p.rank[slot] : 0;
        const float smax = on ? p.smax[slot] : 0.f;
        const int64_t lo = ((int64_t)blockIdx.x * QPB + wid * 32 + pr) * p.wcap;
        __syncthreads();        // the previous pair's list is no longer read
        for (int i = lane; i < n; i += 64) {
            val[i] = p.lval[lo + i];
            pos[i] = p.lpos[lo + i];
        }
        __syncthreads();
        float tau = -INFINITY;
        if (r > 0) {
            for (int i = lane; i < n; i += 64) {
                const float vi = val[i];
                int gt = 0, ge = 0;
                for (int j = 0; j < n; ++j) {
                    const float vj = val[j];
                    gt += vj > vi;
                    ge += vj >= vi;
                }
                if (gt < r && r <= ge) tau = vi;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tau = fmaxf(tau, __shfl_xor(tau, o));
        }
        __syncthreads();        // every lane has ranked against the raw logits
        float Z = 0.f;
        for (int i = lane; i < n; i += 64) {
            const bool keep = r > 0 && val[i] >= tau;
            const float w = keep ? expf(val[i] - smax) : 0.f;
            val[i] = w;
            pos[i] = kl[pos[i]];                  // key position -> key row
            Z += w;
        }
        Z = wave_sum(Z);
        const float Zt = on ? p.zab[slot] + Z : 0.f;
        for (int e = 0; e < d_c && on; ++e) {
            float a = 0.f;
            for (int i = lane; i < n; i += 64)
                if (val[i] > 0.f) a += val[i] * p.Xkp[(int64_t)pos[i] * p.DP + off_c + e];
            a = wave_sum(a);
            if (lane == 0) {
                float *dst = p.out_code + (int64_t)qrow * p.D + off_c + e;
                *dst = (Zt > 0.f) ? (*dst + a) / Zt : 0.f;
            }
        }
        if (lane == 0 && on) {
            if (p.out_tau) p.out_tau[(int64_t)qrow * p.C + c] = tau;
            if (p.out_nvalid) p.out_nvalid[(int64_t)qrow * p.C + c] = p.nk[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 screening kernel (DDG_ATTN_BF16_FILTER, k <= DDG_MAX_TOPK_REG).
//
// The logits of one (query, column) pair are first computed from the bf16 roundings of both operands
// on the bf16 MFMA (v_mfma_f32_32x32x16_bf16: 16x the fp32-matrix rate, half the key bytes).  For
// q = W[c]*x_j and key k:  | sum_d bf16(q_d) bf16(k_d) (fp32 accumulate) - sum_d q_d k_d |
//      <= (2^-7 (1 + 2^-9) + DP 2^-23) * sum_d |q_d||k_d|  <=  FILT_EPS * ||q||_2 * ||k||_2 .
// A key can only belong to the k+2 largest exact logits (NLIST) if its screened logit exceeds the current
// exact (k+2)-th largest minus that margin (with max_i ||k_i|| for ||k||).  Every such candidate is
// re-scored EXACTLY in fp32 (16 lanes compute sum_d (w_d x_d) k_d cooperatively from the fp32 tables) and
// only exact values enter the top-(k+2) list, so the kept set is the fp32 one.
// Three kernels per group of workgroups (launch_filt): k_attn_warm (first WARMK key positions, fp32 MFMA,
// gives every pair a threshold), then per key chunk k_attn_screen (bf16 MFMA stream, thresholds frozen
// at the chunk start, survivors recorded) and k_attn_rescore (exact pass over the survivors, list update,
// fill at the last chunk).  Workgroup = 8 waves x 32 queries of one column sharing one key stream.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// Screening operands.  SCREEN_F16 = 1 (default): f16 with power-of-two scaling -- the same MFMA rate as bf16 and a unit
// roundoff of 2^-11 instead of 2^-8, i.e. a rigorous margin 8x smaller and about a quarter fewer re-scored candidates.
// Keys are stored as f16(kscale * x) (kscale: power of two, largest |element| in [0.5, 1)); a query's operand is
// f16(qsc * q) with qsc the power of two that puts its largest |element| into [2^13, 2^14); thresholds are compared in
// the scaled domain.  With a = qsc q, b = kscale k, |f16(a_d) - a_d| <= u |a_d| + 2^-25 (the second term covers
// subnormal results), the same for b, |b_d| <= 1, exact f16 x f16 products and fp32 accumulation:
//   |sum f16(a) f16(b) - sum a b| <= (2u + u^2 + DP 2^-23) ||a|| ||b|| + 2^-25 (1 + u) (sqrt(DP) ||a|| + DP) + DP 2^-50
// SCREEN_F16 = 0 builds the bf16 screen of round 1 (FILT_EPS).
#ifndef SCREEN_F16
#define SCREEN_F16 1
#endif
#if SCREEN_F16
typedef f16x8 scrx8;
#define SCREEN_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define FILT_EPS16 1.01e-3f          // 2u + u^2 + DP 2^-23 for DP <= 256, rounded up
#else
typedef bf16x8 scrx8;
#define SCREEN_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
#ifdef FILT_STATS   // experiments only: [0] re-scored, [2] batches, [3] tie-flagged pairs
extern "C" int ddg_debug_filt_stats(unsigned long long *out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_filt_stats), 32) != hipSuccess) return DDG_E_LAUNCH;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_filt_stats), z, 32) != hipSuccess) return DDG_E_LAUNCH; }
    return DDG_OK;
}
#endif
#ifndef FILT_NW
#define FILT_NW 8                     // waves per workgroup of the screening kernel (4: two workgroups per CU, measured slower)
#endif
#define FILT_NT (64 * FILT_NW)        // threads per workgroup
#define QPBF (32 * FILT_NW)           // queries per workgroup
#define FILT_EPS 8.0e-3f

// sum over the 16 lanes of a DPP row, result in every lane: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

__device__ __forceinline__ void set_wave_prio(int pr) {      // s_setprio takes an immediate
    switch (pr) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#ifndef FILT_RENDEZVOUS
#define FILT_SPLIT_BARRIER              // -DFILT_RENDEZVOUS: one plain s_barrier per stage instead (A/B builds)
#endif
#ifndef FILT_SIG_J
#define FILT_SIG_J 1                 // half tile after which a wave arrives at the split barrier (0..3 measured: 1 is best)
#endif
#ifndef FILT_PRIO
#define FILT_PRIO(s) (3 - ((s) & 3))
#endif
#define FILT_CAPH 64                  // candidate slots per (query, lane half) and key chunk

// ---- screening pass over one chunk of the key lists: bf16 MFMA logits against the threshold each query
// had when the chunk started; the key-list positions of everything that may reach the top-(k+1) are
// appended to the lane's candidate slots.  No exact arithmetic, no lists: a pure MFMA stream.
// x * M for the small constants the row stride needs, without the quarter-rate 32-bit multiply
template <int M>
__device__ __forceinline__ uint32_t mul_small(uint32_t x) {
    if constexpr (M % 3 == 0) {
        uint32_t t;
        asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(t) : "v"(x));
        return t * (M / 3);
    } else {
        return x * M;
    }
}
// keys per LDS stage of the screening kernel: three stage buffers must fit in 160 KB
#ifndef SCREEN_LA2
#define SCREEN_LA2 4                // fragments in flight per wave for QW = 2 with rows wider than 256 bytes
#endif
#ifndef SCREEN_HALF_KS
#define SCREEN_HALF_KS 8            // QW = 2: half-size stages and two workgroups per CU up to this many steps per row
#endif
#ifndef SCREEN_HALF
#define SCREEN_HALF 1               // 0: full-size stages also for QW = 2 (A/B builds)
#endif
__host__ __device__ constexpr int screen_stage_keys(int KS, int QW = 1) {
    // QW = 2 and rows of <= 256 bytes: half-size stages, so that TWO four-wave workgroups (two waves per SIMD) fit a CU
    return FILT_NW != 8 ? 64 : (KS <= 4 ? 256 : KS <= 8 ? 192 : KS <= 12 ? 128 : 96) / ((SCREEN_HALF && QW == 2 && KS <= SCREEN_HALF_KS) ? 2 : 1);     // 96 ... 147 KB for three buffers
}
// QW = query sets (32 queries each) per wave.  QW = 2: a workgroup is FILT_NW / 2 waves, every key fragment read from LDS
// feeds two MFMAs (one per set): one ds_read_b128 per v_mfma_f32_32x32x16_bf16 is exactly the LDS peak of a CU, so with
// QW = 1 the matrix pipe cannot pass ~85 %.  The per-pair arrays are indexed by the VIRTUAL wave QW * wid + set, so
// k_attn_warm / k_attn_rescore see no difference.
#ifndef SCREEN_LB1
#define SCREEN_LB1 0                // 1: one wave per SIMD for every QW = 2 instantiation (A/B builds)
#endif
// KSC = 16-deep steps that hold data (ceil(D / 16) <= KS): the zero padding up to the row stride is copied to LDS with
// the row but neither read from there nor multiplied (c3: D = 176 in 192-element rows, 11 of 12 steps).
template <int KS, int QW = 1, int KSC = KS>
__global__ __launch_bounds__(FILT_NT / QW, ((QW == 1 || (SCREEN_HALF && !SCREEN_LB1 && KS <= SCREEN_HALF_KS)) ? 2 : 1)) void k_attn_screen(AttnArgs p) {
    static_assert(KSC >= 1 && KSC <= KS, "compute steps");
    constexpr int NWP = FILT_NW / QW;        // waves of the workgroup
    constexpr int NTP = 64 * NWP;            // its threads
    constexpr int DP = KS * 16;
    constexpr int RSTR = DP * 2;             // LDS row stride in bytes: rows are dense (the LDS-DMA image is lane-linear)
    constexpr int CPR = DP / 8;              // 16-byte chunks per key row
    // Bank conflicts are avoided by an XOR swizzle of the chunk position inside the row, applied on the
    // SOURCE address of the DMA and on the ds_read address.  A ds_read_b128 is served in groups of 16 lanes
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), bank = (byte / 4) mod 64.  Row strides of 32 / 96 dwords
    // (KS 4, 12) alternate between two bank halves, so 8 positions x 2 row parities cover the 64 banks when
    // rows of equal parity get distinct XOR keys: (row >> 1) & 7.  Strides of 64 / 128 dwords (KS 8, 16)
    // need 16 distinct positions: key = row & 15.  Both keys are distinct inside every 16-lane group.
    constexpr int SMASK = (KS % 8 == 4) ? 7 : 15;
    constexpr int TKS = screen_stage_keys(KS, QW);   // keys per stage
    constexpr int NCH = TKS * CPR;
    constexpr int NPF = NCH / NTP;
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    auto kbuf = [&](int b) -> char * { return fsm + b * (TKS * RSTR); };      // three stage buffers
    __shared__ int s_col;
#ifdef FILT_SPLIT_BARRIER
    __shared__ unsigned s_sync;               // waves that have their DMA pieces of stages 1 .. t+1 in LDS, summed
#endif

    const int nblocks = p.qb_start[p.C];
    const int blk = p.blk_lo + (int)blockIdx.x;
    if (blk >= nblocks) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    if (tid == 0) s_col = block_column(p.qb_start, p.C, blk);
#ifdef FILT_SPLIT_BARRIER
    if (tid == 0) s_sync = 0;
#endif
    __syncthreads();
    const int c = s_col;
    const int nk = p.nk[c];
    const int key_lo = p.key_lo;
    const int key_hi = p.key_hi < nk ? p.key_hi : nk;
    if (key_lo >= nk) return;                 // column exhausted by an earlier chunk (or by the warm-up kernel)
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    // the largest key norm this launch can meet: with norm-ordered lists every key from position key_lo on lies in the
    // bucket of the key AT key_lo or a later one -- the margin of a late chunk is that much smaller than the table's
    float kmax_here = p.kmax[0];
    if (p.knorm) {
        const int b = key_bucket(p.knorm[kl[key_lo]], kmax_here);
        kmax_here = fminf(kmax_here, kmax_here * ((float)(KEY_BUCKETS - b) / (float)KEY_BUCKETS) * 1.00001f);
    }
    // query operands: bf16(W[c][k] * x_q[k]) for k = 16 st + 8 h + j, and the margin from ||q||, per query set
    scrx8 breg[QW][KS];
    float thr_adj[QW];
    int32_t *cand[QW];
    int ncand[QW];
    int vtid[QW];                             // thread number of the (virtual wave, lane) in the per-pair arrays
#pragma unroll
    for (int qs = 0; qs < QW; ++qs) {
        const int vw = QW * wid + qs;
        vtid[qs] = vw * 64 + lane;
        const int64_t slot = (int64_t)blk * QPBF + vw * 32 + r32;
        const int qrow = p.qlist[slot];
        float qn2 = 0.f;
        const float *wq = p.Wp + (int64_t)c * DP + 8 * h;
        const float *xq = p.Xqp + (int64_t)(qrow < 0 ? 0 : qrow) * DP + 8 * h;
#if SCREEN_F16
        float qmax = 0.f;
#pragma unroll
        for (int st = 0; st < KS; ++st)
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + 16 * st + 4 * j4);
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xq + 16 * st + 4 * j4);
#pragma unroll
                for (int j = 0; j < 4; ++j) qmax = fmaxf(qmax, fabsf(wv[j] * xv[j]));
            }
        qmax = fmaxf(qmax, __shfl_xor(qmax, 32));
        int eq = 14;
        if (qmax > 0.f && qmax < INFINITY && qrow >= 0) frexpf(qmax, &eq);
        const float qsc = ldexpf(1.f, 14 - eq);
#else
        const float qsc = 1.f;
#endif
#pragma unroll
        for (int st = 0; st < KS; ++st) {
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wq + 16 * st + 4 * j4);
                f32x4 xv = *reinterpret_cast<const f32x4 *>(xq + 16 * st + 4 * j4);
                if (qrow < 0) xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q = wv[j] * xv[j];
                    qn2 += q * q;
#if SCREEN_F16
                    breg[qs][st][4 * j4 + j] = (_Float16)(q * qsc);
#else
                    breg[qs][st][4 * j4 + j] = (__bf16)q;
#endif
                }
            }
        }
        qn2 += __shfl_xor(qn2, 32);
        // threshold = the query's exact (k+1)-th largest logit so far (parked by its h == 0 lane)
        const float thr = p.st_val[((int64_t)blockIdx.x * NLIST(p.k) + NLIST(p.k) - 1) * FILT_NT + (vtid[qs] & ~32)];
#if SCREEN_F16
        {   // margin and threshold in the scaled domain (see the bound above)
            const float ksc = p.kscale[0];
            const float an = qsc * sqrtf(qn2) * 1.000001f;                         // ||a||, rounded up
            const float marg = FILT_EPS16 * an * (ksc * kmax_here) + 0x1p-25f * 1.001f * (sqrtf((float)DP) * an + (float)DP) + 1e-30f;
            thr_adj[qs] = (qrow < 0) ? INFINITY : thr * (qsc * ksc) - marg * 1.000001f;
        }
#else
        const float qe = FILT_EPS * sqrtf(qn2) * 1.000001f * kmax_here + 1e-30f;
        thr_adj[qs] = (qrow < 0) ? INFINITY : thr - qe;
#endif
        cand[qs] = p.cand_pos + (int64_t)blockIdx.x * FILT_CAPH * FILT_NT + vtid[qs];
        ncand[qs] = 0;
    }

    // key staging, three LDS stage buffers: while stage t is computed the rows of stage t+1 are written to
    // their buffer, the rows of stage t+2 are in flight to registers and the key-list indices of stage
    // t+3 are in flight too, so neither the index -> row dependence nor the Infinity-Cache / HBM latency
    // is waited for.  Rows past the end of the chunk re-read its last key (no branch, no zero fill); the
    // candidate collection ignores them through `rem`.
    static_assert(NCH % NTP == 0, "key stage must divide evenly over the workgroup");
    // LDS unit n (16 bytes) of a stage buffer: wave `wid` fills pieces wid*NPF .. wid*NPF+NPF-1 of 1 KiB, lane l of
    // piece q is unit 64 q + l = (row, position); it receives chunk position ^ key(row) of that key row.
    // The pieces of wave `wid` cover exactly the rows [wid*RPW, wid*RPW + RPW) of the stage: one index load per
    // wave and stage (lane l fetches the index of row l % RPW), handed to the lanes of a piece by ds_bpermute.
    constexpr int RPW = TKS / NWP;            // key rows per wave and stage
    static_assert(RPW * CPR == NPF * 64 && RPW <= 64, "a wave's pieces cover whole rows");
    int kix = 0;
    int urow4[NPF];                           // 4 * (row of the stage this lane fetches with piece i, relative to the wave)
    uint32_t usrc[NPF];                       // byte offset of the source chunk inside the key row
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int n = i * 64 + lane;
        const int rrow = n / CPR, pos = n - rrow * CPR;
        const int row = wid * RPW + rrow;
        const int key = (SMASK == 7) ? ((row >> 1) & 7) : (row & 15);
        urow4[i] = rrow * 4;
        usrc[i] = (uint32_t)(((pos & ~SMASK) | ((pos ^ key) & SMASK)) * 16);
    }
    auto load_idx = [&](int t) {
        const int pos = key_lo + t * TKS + wid * RPW + (lane % RPW);
        kix = kl[pos < key_hi ? pos : key_hi - 1];
    };
    auto row_offsets = [&](uint32_t (&kr)[NPF]) {
#pragma unroll
        for (int i = 0; i < NPF; ++i)
            kr[i] = mul_small<KS>((uint32_t)__builtin_amdgcn_ds_bpermute(urow4[i], kix));
    };
    // row bytes = 32 * KS * index: a 32-bit shift-add and one 64-bit add instead of a quarter-rate multiply
    // (indices < 2^27: attn_check_dims)
    uint32_t krow[NPF];                       // 3-bit-shifted row offset; written a half tile before the DMA: hipcc
                                              // drains vmcnt(0) at a use of a plain load's result while a DMA is in flight
    auto dma_piece = [&](int b, int i) {
        const uint64_t roff = ((uint64_t)krow[i] << 5) + usrc[i];
        __builtin_amdgcn_global_load_lds(
            reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(p.Xkh) + roff),
            reinterpret_cast<__attribute__((address_space(3))) void *>(
                (uint32_t)(uintptr_t)(kbuf(b) + (wid * NPF + i) * 1024)), 16, 0, 0);
    };

    // the not-so-rare path of one 32-key tile (about one tile in five at 1M rows): which of the lane's 16
    // logits reach the threshold.  It is on the wave's critical path and a stage waits for its slowest
    // wave, so it is kept short and branch-free: a 16-bit mask of the hits, stored with the tile's number
    // as ONE word per lane and tile.  k_attn_rescore unpacks the words (and drops the padding rows past the
    // end of the chunk, which hold copies of its last key).
    auto collect = [&](const f32x16 &acc, int tile, int qs) {
        // bit g = sign of (threshold - logit g): one subtract and one v_alignbit per logit, no compare / select
        // chain through vcc (logit 15 first, so that logit g ends at bit g)
        unsigned cm = 0;
#pragma unroll
        for (int g = 15; g >= 0; --g)
            cm = __builtin_amdgcn_alignbit(cm, __float_as_uint(thr_adj[qs] - acc[g]), 31);
        cm &= 0xffffu;
        if (cm) {
            if (ncand[qs] < FILT_CAPH) cand[qs][ncand[qs] * FILT_NT] = (int32_t)(((unsigned)tile << 16) | cm);
            ++ncand[qs];
        }
    };
    // the 8 operations of a tile's row maximum, one at a time (slot fillers)
    auto max_op = [&](const f32x16 &pa, float &m, int o) {
        m = (o == 0) ? fmaxf(pa[0], pa[1]) : fmaxf(fmaxf(m, pa[2 * o]), pa[2 * o + 1]);
    };

    // The key stream is one unbroken software pipeline of half tiles (H MFMAs each), written out by hand.
    // Every MFMA slot carries the ds_read of the fragment that the slot H positions later consumes -- at
    // the end of a stage that is already the first fragment of the next stage's buffer -- plus one filler:
    // a key-list index load of stage t+2 (first half tile), a max3 of the previous tile's logits (second
    // halves), or an LDS-DMA piece of stage t+2's key rows (last half tile).  sched_barrier keeps the
    // slots as written.  Two accumulator sets alternate, so a tile's threshold test runs one tile late,
    // beside the next tile's MFMAs.
    // Synchronisation, three stage buffers: the DMA of stage t+1 was issued in the last half tile of stage
    // t-1; each wave retires its own pieces with a counted vmcnt (the younger index loads stay in flight)
    // before the one barrier of stage t, which precedes the last half tile -- the first reader of buffer
    // t+1.  The DMA of stage t+2 is issued after that barrier into the buffer of stage t-1, whose last
    // fragment reads were issued before the barrier of stage t-1 and consumed a whole stage ago.
    constexpr int H = KS / 2;
    constexpr int NT_STAGE = TKS / 32;
    static_assert(NT_STAGE >= 2 && FILT_SIG_J < 2 * NT_STAGE - 1, "slot plan");
    const int nst = (key_hi - key_lo + TKS - 1) / TKS;
    load_idx(0);
    row_offsets(krow);
#pragma unroll
    for (int i = 0; i < NPF; ++i) dma_piece(0, i);
    load_idx(1);
    row_offsets(krow);
#pragma unroll
    for (int i = 0; i < NPF; ++i) dma_piece(1, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // fragment st of tile row r32: chunk 2 st + h, stored at position chunk ^ key(r32); the low bits of the
    // chunk take SMASK/2+1 values, one precomputed lane offset each
    constexpr int NOFF = (SMASK + 1) / 2;
    int foff[NOFF];
    {
        const int key = (SMASK == 7) ? ((r32 >> 1) & 7) : (r32 & 15);
#pragma unroll
        for (int m = 0; m < NOFF; ++m) foff[m] = r32 * RSTR + (((2 * m + h) ^ key) & SMASK) * 16;
    }
    auto frag = [&](const char *buf, int tile, int st) {
        return *reinterpret_cast<const scrx8 *>(buf + foff[st % NOFF] + tile * 32 * RSTR + ((2 * st) & ~SMASK) * 16);
    };
    // ring of the LA fragments in flight: slot n of a stage (n = 12-slot tile number * KS + step) consumes entry n % LA
    // and refills it with the fragment of slot n + LA
    constexpr int LA = (QW == 2 && KS > 8) ? SCREEN_LA2 : H;
    static_assert(LA <= H && (NT_STAGE * KS) % LA == 0, "fragment ring");
    scrx8 fr[LA];
    f32x16 cacc[QW][2];
    float mx[QW][2];
#pragma unroll
    for (int qs = 0; qs < QW; ++qs) {
        mx[qs][0] = mx[qs][1] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) { cacc[qs][0][g] = 0.f; cacc[qs][1][g] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < LA; ++i)
        if (i < KSC) fr[i] = frag(kbuf(0), 0, i);
    __builtin_amdgcn_sched_barrier(0);
#ifdef FILT_TIMING
    unsigned long long tm_bar = 0, tm_coll = 0, tm_vm = 0, tm_ct = 0, ts_begin = TSTAMP();
#endif
    int b_cur = 0, b_nxt = 1;
    // one stage; PAR = parity of the stage's first tile (the two accumulator sets alternate per tile, and a stage of
    // an odd number of tiles flips the parity: the loop below then runs stages in pairs)
    auto stage_body = [&](auto parc, int t) {
        constexpr int PAR = decltype(parc)::value;
        const char *base = kbuf(b_cur);
        const char *base_n = kbuf(b_nxt);
        const int b_nn = 3 - b_cur - b_nxt;
#pragma unroll
        for (int s = 0; s < NT_STAGE; ++s) {
            // The SIMD's arbiter favours its oldest wave, and a dependent MFMA chain is always ready: left
            // alone, one wave of the pair runs ahead and then idles at the barrier while its partner runs
            // alone.  Priority falls with the tile index, so the wave that is behind owns the matrix pipe.
            // (two workgroups per CU: their waves must not out-prioritise each other)
#ifdef SCREEN_Q2_PRIO
            set_wave_prio(FILT_PRIO(s));
#else
            set_wave_prio((QW == 2 && KS <= 8) ? 0 : FILT_PRIO(s));
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * s + u;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    if (u == 0 && i == 0) {
#pragma unroll
                        for (int qs = 0; qs < QW; ++qs)
#pragma unroll
                            for (int g = 0; g < 16; ++g) cacc[qs][(s + PAR) & 1][g] = 0.f;
                    }
#pragma unroll
                    for (int qs = 0; qs < QW; ++qs)
                        if (u * H + i < KSC)
                            cacc[qs][(s + PAR) & 1] = SCREEN_MFMA(fr[(j * H + i) % LA], breg[qs][u * H + i], cacc[qs][(s + PAR) & 1], 0, 0, 0);
                    {
                        constexpr int SLOTS = NT_STAGE * KS;
                        const int m = j * H + i + LA;             // the slot this entry serves next
                        if (m < SLOTS) {
                            if (m % KS < KSC) fr[(j * H + i) % LA] = frag(base, m / KS, m % KS);
                        } else if (m - SLOTS < KSC)
                            fr[(j * H + i) % LA] = frag(base_n, 0, m - SLOTS);
                    }
                    if (j == 0 && i == 0) load_idx(t + 2);        // key-list indices of stage t+2
                    if (j == 2 * NT_STAGE - 2) {
#pragma unroll
                        for (int e = i; e < NPF; e += H)
                            krow[e] = mul_small<KS>((uint32_t)__builtin_amdgcn_ds_bpermute(urow4[e], kix));
                    }
                    if (j == 2 * NT_STAGE - 1) {                  // key rows of stage t+2
#pragma unroll
                        for (int e = i; e < NPF; e += H) dma_piece(b_nn, e);
                    }
                    if (u == 1) {                        // previous tile's row maximum, 8 ops over H slots
#pragma unroll
                        for (int o = 0; o < 8; ++o)
                            if (o * H / 8 == i) {
#pragma unroll
                                for (int qs = 0; qs < QW; ++qs) max_op(cacc[qs][(s + 1 + PAR) & 1], mx[qs][(s + 1 + PAR) & 1], o);
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef FILT_SPLIT_BARRIER
                // Split barrier.  Arrive (mid-stage): this wave's DMA pieces of stage t+1 have landed -> one LDS
                // atomic by one lane (exec set by hand: no control flow in the tile stream).  Wait (before the
                // last half tile, the first reader of buffer t+1): all FILT_NW waves have arrived for stage t.
                // A wave can run about half a stage ahead of the slowest one instead of stopping at a
                // rendezvous.  The DMA of stage t+2, issued after the wait, overwrites the buffer of stage t-1:
                // every wave has arrived for stage t, so it is past the middle of stage t and has consumed its
                // fragments of stage t-1.
                if (j == FILT_SIG_J) {
                    unsigned long long sv;
                    asm volatile("s_waitcnt vmcnt(1)\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                                 : "=&s"(sv) : "v"((uint32_t)(uintptr_t)&s_sync), "v"(1u) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j == 2 * NT_STAGE - 2) {
                    const unsigned target = (unsigned)NWP * (unsigned)(t + 1);
#ifdef FILT_TIMING
                    const unsigned long long tb0 = TSTAMP();
#endif
                    while (__builtin_amdgcn_readfirstlane(*(volatile unsigned *)&s_sync) < target) __builtin_amdgcn_s_sleep(1);
#ifdef FILT_TIMING
                    tm_bar += TSTAMP() - tb0;
#endif
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                if (j == 2 * NT_STAGE - 2) {
                    // own DMA pieces of stage t+1 landed (the index load issued since may still fly),
                    // then every wave's: buffer t+1 is complete before its first read
#ifdef FILT_TIMING
                    const unsigned long long tb0 = TSTAMP();
                    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    const unsigned long long tb1 = TSTAMP();
                    tm_vm += tb1 - tb0;
                    asm volatile("s_barrier" ::: "memory");
                    tm_bar += TSTAMP() - tb0;
#else
                    asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            if (s > 0 || t > 0) {                        // the tile before this one (s == 0: the previous stage's last)
#pragma unroll
                for (int qs = 0; qs < QW; ++qs) {
                    if (__builtin_amdgcn_ballot_w64(mx[qs][(s + 1 + PAR) & 1] > thr_adj[qs]) != 0)
                        collect(cacc[qs][(s + 1 + PAR) & 1], t * NT_STAGE + s - 1, qs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        b_cur = b_nxt;
        b_nxt = b_nn;
    };
    if constexpr (NT_STAGE % 2 == 0) {
#pragma unroll 1
        for (int t = 0; t < nst; ++t) stage_body(std::integral_constant<int, 0>{}, t);
    } else {
#pragma unroll 1
        for (int t = 0; t < nst; t += 2) {
            stage_body(std::integral_constant<int, 0>{}, t);
            if (t + 1 < nst) stage_body(std::integral_constant<int, 1>{}, t + 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA of the stages past the end must not outlive the wave
    set_wave_prio(0);
    auto last_tile = [&](const f32x16 &pa, int qs) {     // last tile of the chunk
        float m = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) max_op(pa, m, o);
        if (__builtin_amdgcn_ballot_w64(m > thr_adj[qs]) != 0) collect(pa, nst * NT_STAGE - 1, qs);
    };
    // (a branch per parity, not a reference picked at run time: that would put the accumulators into scratch memory)
#pragma unroll
    for (int qs = 0; qs < QW; ++qs) {
        if ((nst * NT_STAGE - 1) & 1) last_tile(cacc[qs][1], qs); else last_tile(cacc[qs][0], qs);
    }
#ifdef FILT_TIMING
    if (lane == 0) {
        atomicAdd(&g_filt_time[2], tm_bar); atomicAdd(&g_filt_time[3], (unsigned long long)nst);
        atomicAdd(&g_filt_time[4], TSTAMP() - ts_begin); atomicAdd(&g_filt_time[5], tm_coll); atomicAdd(&g_filt_time[6], tm_vm); atomicAdd(&g_filt_time[7], tm_ct);
    }
#endif
#pragma unroll
    for (int qs = 0; qs < QW; ++qs)
        p.cand_cnt[(int64_t)blockIdx.x * FILT_NT + vtid[qs]] = ncand[qs];      // > FILT_CAPH: overflow, pair goes to the fp32 path
}

// ---- exact pass over the candidates of one chunk: every candidate is re-scored in fp32 (16 lanes per
// candidate, four candidates per wave instruction stream) and offered to the query's top-(k+1) list.
// No key stream and no workgroup barrier: waves only wait for their own gathers.
template <int KS, int KCAP>
__global__ __launch_bounds__(FILT_NT, 2) void k_attn_rescore(AttnArgs p) {
    constexpr int DP = KS * 16;
    __shared__ int s_col;
    const int nblocks = p.qb_start[p.C];
    const int blk = p.blk_lo + (int)blockIdx.x;
    if (blk >= nblocks) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    if (tid == 0) s_col = block_column(p.qb_start, p.C, blk);
    __syncthreads();
    const int c = s_col;
    const int nk = p.nk[c];
    if (p.key_lo >= nk) return;
    const bool last_chunk = p.key_hi >= nk;
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    const int off_c = p.offsets[c];
    const int d_c = p.offsets[c + 1] - off_c;
    const int64_t slot = (int64_t)blk * QPBF + wid * 32 + r32;
    const int qrow = p.qlist[slot];

    float lst[KCAP];
    int lpos[KCAP];
    const int ndummy = KCAP - NLIST(p.k);
    float *sval = p.st_val + ((int64_t)blockIdx.x * NLIST(p.k)) * FILT_NT + tid;
    int32_t *spos = p.st_pos + ((int64_t)blockIdx.x * NLIST(p.k)) * FILT_NT + tid;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        lst[i] = INFINITY;
        lpos[i] = -1;
        if (i >= ndummy && h == 0) {
            lst[i] = sval[(int64_t)(i - ndummy) * FILT_NT];
            lpos[i] = spos[(int64_t)(i - ndummy) * FILT_NT];
        }
    }
    int32_t *ovfp = p.st_ovf + (int64_t)blockIdx.x * FILT_NT + tid;
    const int32_t *cand = p.cand_pos + (int64_t)blockIdx.x * FILT_CAPH * FILT_NT + tid;
    int ncand = qrow < 0 ? 0 : p.cand_cnt[(int64_t)blockIdx.x * FILT_NT + tid];
    int ovf = *ovfp;
    if (ncand > FILT_CAPH) { ovf = 1; ncand = FILT_CAPH; }
    ovf |= __shfl_xor(ovf, 32);

    const float *wq = p.Wp + (int64_t)c * DP + KS * (lane & 15);
    f32x4 wreg[KS / 4];
#pragma unroll
    for (int i = 0; i < KS / 4; ++i) wreg[i] = *reinterpret_cast<const f32x4 *>(wq + 4 * i);

    // candidate words of k_attn_screen: (tile of the chunk << 16) | mask of the lane's 16 logits of that tile
    const int key_end = p.key_hi < nk ? p.key_hi : nk;
    int ent = 0, tpos = 0;
    unsigned msk = 0;
    while (__any(msk != 0 || ent < ncand)) {
        if (msk == 0 && ent < ncand) {
            const unsigned w = (unsigned)cand[(int64_t)ent * FILT_NT];
            ++ent;
            msk = w & 0xffffu;
            tpos = p.key_lo + (int)(w >> 16) * 32 + 4 * h;
        }
        bool has = msk != 0;
        int mypos = 0;
        if (has) {
            const int g = __ffs((int)msk) - 1;
            msk &= msk - 1;
            mypos = tpos + (g & 3) + 8 * (g >> 2);
            has = mypos < key_end;                       // padding rows of the chunk's last stage
        }
        const int mykidx = has ? kl[mypos] : 0;
        float mysv = -INFINITY;
        // Four queries per batch, one per 16-lane group; a group scores the candidates of BOTH lanes of its query
        // (lane r and lane r + 32) against one load of the query row.
        const unsigned long long bal = __ballot(has);
        unsigned qm = (unsigned)bal | (unsigned)(bal >> 32);
        while (qm) {
            int src[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                src[j] = -1;
                if (qm) { src[j] = __ffs((int)qm) - 1; qm &= qm - 1; }
            }
            const int grp = lane >> 4;
            const int sm = grp == 0 ? src[0] : grp == 1 ? src[1] : grp == 2 ? src[2] : src[3];
            const int smc = sm < 0 ? 0 : sm;
            int ka = __shfl(mykidx, smc);
            int kb = __shfl(mykidx, smc + 32);
            int qr = __shfl(qrow, smc);
            if (sm < 0) { qr = 0; ka = 0; kb = 0; }       // idle group: loads row 0, results ignored (lanes without a
                                                          // candidate hold key index 0 as well)
            const float *xq = p.Xqp + (int64_t)qr * DP + KS * (lane & 15);
            const float *xa = p.Xkp + (int64_t)ka * DP + KS * (lane & 15);
            const float *xb = p.Xkp + (int64_t)kb * DP + KS * (lane & 15);
            f32x4 qw[KS / 4], xav[KS / 4], xbv[KS / 4];
#pragma unroll
            for (int i = 0; i < KS / 4; ++i) {
                qw[i] = *reinterpret_cast<const f32x4 *>(xq + 4 * i);
                xav[i] = *reinterpret_cast<const f32x4 *>(xa + 4 * i);
                xbv[i] = *reinterpret_cast<const f32x4 *>(xb + 4 * i);
            }
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int i = 0; i < KS / 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = wreg[i][e] * qw[i][e];
                    pa = fmaf(q, xav[i][e], pa);
                    pb = fmaf(q, xbv[i][e], pb);
                }
            const float sva = row16_sum(pa), svb = row16_sum(pb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float aj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(sva), 16 * j));
                const float bj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__float_as_int(svb), 16 * j));
                if (has && lane == src[j]) mysv = aj;
                if (has && src[j] >= 0 && lane == src[j] + 32) mysv = bj;
            }
        }
#ifdef FILT_STATS
        if (has) atomicAdd(&g_filt_stats[0], 1ull);
        if (lane == 0) atomicAdd(&g_filt_stats[2], 1ull);
#endif
        const float osv = __shfl_xor(mysv, 32);
        const int opos = __shfl_xor(mypos, 32);
        if (__any(h == 0 && mysv > lst[KCAP - 1])) {
            if (h == 0 && mysv > lst[KCAP - 1]) list_insert_kv<KCAP>(lst, lpos, mysv, mypos);
        }
        if (__any(h == 0 && osv > lst[KCAP - 1])) {
            if (h == 0 && osv > lst[KCAP - 1]) list_insert_kv<KCAP>(lst, lpos, osv, opos);
        }
    }
    if (last_chunk) {
        topk_output<KCAP>(p, lst, lpos, slot, qrow, nk, kl, off_c, d_c, DP, h, ovf != 0);
    } else {
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < KCAP; ++i)
                if (i >= ndummy) {
                    sval[(int64_t)(i - ndummy) * FILT_NT] = lst[i];
                    spos[(int64_t)(i - ndummy) * FILT_NT] = lpos[i];
                }
        }
        *ovfp = ovf;
    }
}

// Warm-up of the screening kernel: the first WARMK key-list positions in exact fp32 (fp32 MFMA, as
// k_attn<MODE_TOPK>, same workgroup geometry as k_attn_filt).  It gives every query a threshold before
// the bf16 stream starts, so that only about (k+1) ln(N / WARMK) + margin keys per query need the
// exact re-scoring instead of the first few hundred.  Lists are parked for k_attn_filt (h == 0 lanes),
// or finished right here when the column has no more than WARMK keys.
#define WARMK 512
template <int KS4, int KCAP>
__global__ __launch_bounds__(FILT_NT, 2) void k_attn_warm(AttnArgs p) {
    constexpr int DP = KS4 * 8;
    constexpr int LSTR = DP + 4;
    constexpr int NPF = (TK * DP / 4 + FILT_NT - 1) / FILT_NT;   // float4 per thread and 32-key tile
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    __shared__ int s_col;
    const int nblocks = p.qb_start[p.C];
    const int blk = p.blk_lo + (int)blockIdx.x;
    if (blk >= nblocks) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    if (tid == 0) s_col = block_column(p.qb_start, p.C, blk);
    __syncthreads();
    const int c = s_col;
    const int nk = p.nk[c];
    const int nkw = nk < WARMK ? nk : WARMK;
    const int32_t *kl = p.klist + (int64_t)c * p.Nk;
    const int off_c = p.offsets[c];
    const int d_c = p.offsets[c + 1] - off_c;
    const int64_t slot = (int64_t)blk * QPBF + wid * 32 + r32;
    const int qrow = p.qlist[slot];

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
    float lst[KCAP];
    int lpos[KCAP];
    const int ndummy = KCAP - NLIST(p.k);
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        lst[i] = (i < ndummy || qrow < 0) ? INFINITY : -INFINITY;
        lpos[i] = -1;
    }

    f32x4 pf[NPF];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + FILT_NT * i;
            const int row = idx / (DP / 4), c4 = idx - row * (DP / 4);
            const int pos = t * TK + row;
            const int kidx = (row < TK && pos < nkw) ? kl[pos] : -1;
            pf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kidx >= 0) pf[i] = *reinterpret_cast<const f32x4 *>(p.Xkp + (int64_t)kidx * DP + c4 * 4);
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + FILT_NT * i;
            const int row = idx / (DP / 4), c4 = idx - row * (DP / 4);
            if (row < TK) *reinterpret_cast<f32x4 *>(buf + row * LSTR + c4 * 4) = pf[i];
        }
    };
    float *kbuf0 = wsm, *kbuf1 = wsm + TK * LSTR;
    const int ntk = (nkw + TK - 1) / TK;
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
        const int rem = nkw - t * TK;
        float thr = lst[KCAP - 1];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int kr = (g & 3) + 8 * (g >> 2) + 4 * h;
            float v = acc[g];
            if (kr >= rem) v = -INFINITY;
            if (__any(v > thr)) {
                list_insert_kv<KCAP>(lst, lpos, v, t * TK + kr);
                thr = lst[KCAP - 1];
            }
        }
        if (t + 1 < ntk) commit(nxt);
        __syncthreads();
    }
    merge_halves<KCAP>(p, lst, lpos);
    if (nk <= WARMK) {
        topk_output<KCAP>(p, lst, lpos, slot, qrow, nk, kl, off_c, d_c, DP, h);
    } else {
        p.st_ovf[(int64_t)blockIdx.x * FILT_NT + tid] = 0;
    }
    if (nk > WARMK && h == 0) {
        float *sval = p.st_val + ((int64_t)blockIdx.x * NLIST(p.k)) * FILT_NT + tid;
        int32_t *spos = p.st_pos + ((int64_t)blockIdx.x * NLIST(p.k)) * FILT_NT + tid;
#pragma unroll
        for (int i = 0; i < KCAP; ++i)
            if (i >= ndummy) {
                sval[(int64_t)(i - ndummy) * FILT_NT] = lst[i];
                spos[(int64_t)(i - ndummy) * FILT_NT] = lpos[i];
            }
    }
}

// keys -> screening operand (f16 of kscale * x, or bf16 of x), and the largest key 2-norm (of x, rounded up) for the
// screening margin; one wave per row.  max_bits: largest |element| (k_maxabs), kscale_out: the power of two derived from it
__global__ void k_keys_screen(const float *__restrict__ Xkp, int64_t rows, int DP, __bf16 *__restrict__ Xkh,
                              unsigned *__restrict__ kmax_bits, const unsigned *__restrict__ max_bits,
                              float *__restrict__ kscale_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
#if SCREEN_F16
    const float mxv = __uint_as_float(max_bits[0]);
    int e2 = 0;
    if (mxv > 0.f && mxv < INFINITY) frexpf(mxv, &e2);
    const float ksc = ldexpf(1.f, -e2);
    if (blockIdx.x == 0 && threadIdx.x == 0) kscale_out[0] = ksc;
    _Float16 *out = reinterpret_cast<_Float16 *>(Xkh);
#endif
    float best = 0.f;
    for (int64_t r = row0; r < rows; r += nw) {
        float s2 = 0.f;
        for (int e = lane; e < DP; e += 64) {
            const float v = Xkp[r * DP + e];
#if SCREEN_F16
            out[r * DP + e] = (_Float16)(v * ksc);
#else
            Xkh[r * DP + e] = (__bf16)v;
#endif
            s2 += v * v;
        }
        s2 = wave_sum(s2);
        best = fmaxf(best, sqrtf(s2) * 1.000001f);
    }
    if (lane == 0 && kmax_bits) atomicMax(kmax_bits, __float_as_uint(best));     // non-negative floats order like their bits
}

// largest |element| of the padded keys (non-negative floats order like their bits)
__global__ void k_maxabs(const float *__restrict__ X, int64_t n, unsigned *__restrict__ out_bits) {
    float best = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        best = fmaxf(best, fabsf(X[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(best));
}

// keys -> two f16 halves of kscale * x per element (row = DP leading halves, then DP remainders scaled by 2^11);
// kscale = the power of two that brings the largest |element| into [0.5, 1), written to kscale_out by every thread
__global__ void k_keys_split(const float *__restrict__ Xkp, int64_t rows, int DP, _Float16 *__restrict__ Xks,
                             const unsigned *__restrict__ max_bits, float *__restrict__ kscale_out) {
    const float mx = __uint_as_float(max_bits[0]);
    int e2 = 0;
    if (mx > 0.f && mx < INFINITY) frexpf(mx, &e2);
    const float ksc = ldexpf(1.f, -e2);
    if (blockIdx.x == 0 && threadIdx.x == 0) kscale_out[0] = ksc;
    const int64_t n = rows * DP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / DP;
        const int k = (int)(e - r * DP);
        const float x = Xkp[e] * ksc;
        const _Float16 hi = (_Float16)x;
        Xks[r * 2 * DP + k] = hi;
        Xks[r * 2 * DP + DP + k] = (_Float16)((x - (float)hi) * 2048.f);
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

// ---------------------------------------------------------------------------------------------
// Key order of the screening path: by DESCENDING 2-norm (a stable counting sort into KEY_BUCKETS linear buckets of
// norm / largest norm).  The top-k logits of a query come mostly from the keys of large norm, so with those first the
// queries' thresholds are close to final after the first chunks and the later chunks pass fewer candidates to the exact
// re-scoring (measured at c3, 1M rows: select 2.556 -> 2.447 s with the rows sorted by norm, 3.507 s with the opposite
// order).  Which keys are kept does not depend on the order (only the resolution of ties below the rounding of a logit
// can: the first WARMK positions are scored by another kernel).  perm[pos] = key row at position pos of that order.  The
// padded key rows (Xkp, Xkh) are STORED in that order and the per-column key lists hold positions, ascending: the
// streams stay sequential in memory (lists in norm order over rows in table order measured only -1.5 %: random 384-byte
// gathers).  Queries keep their row order (their padded copy is then separate from the keys' even when the caller's
// queries alias its keys).  Deterministic: equal buckets keep their row order.
// ---------------------------------------------------------------------------------------------
// 2-norm of every key row of the caller's table (one wave per row) and the largest one, rounded up
__global__ void k_row_norms(const float *__restrict__ X, int64_t rows, int D, float *__restrict__ knorm,
                            unsigned *__restrict__ kmax_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    float s2 = 0.f;
    for (int e = lane; e < D; e += 64) { const float v = X[r * D + e]; s2 += v * v; }
    s2 = wave_sum(s2);
    if (lane == 0) {
        knorm[r] = sqrtf(s2);
        const unsigned bits = __float_as_uint(sqrtf(s2) * 1.000001f);
        // (a million atomics on one word serialise: only a row that beats the maximum seen so far issues one)
        if (bits > __hip_atomic_load(kmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(kmax_bits, bits);
    }
}
// Xp[pos] = zero-padded X[perm[pos]]
__global__ void k_pad_rows_perm(const float *__restrict__ X, float *__restrict__ Xp, int64_t rows, int D, int DP,
                                const int32_t *__restrict__ perm) {
    const int64_t n = rows * DP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / DP;
        const int k = (int)(e - r * DP);
        Xp[e] = k < D ? X[(int64_t)perm[r] * D + k] : 0.f;
    }
}
// bcnt[b][chunk] = rows of the chunk (256 rows) in bucket b
__global__ void k_bucket_count(const float *__restrict__ knorm, int64_t rows, const unsigned *__restrict__ kmax_bits,
                               int nchunks, int32_t *__restrict__ bcnt) {
    __shared__ int h[KEY_BUCKETS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < rows) atomicAdd(&h[key_bucket(knorm[row], __uint_as_float(kmax_bits[0]))], 1);
    __syncthreads();
    bcnt[(int64_t)threadIdx.x * nchunks + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of the bucket totals
__global__ void k_bucket_starts(const int32_t *__restrict__ btot, int32_t *__restrict__ bstart) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < KEY_BUCKETS; ++b) { bstart[b] = run; run += btot[b]; }
    }
}
// perm[bstart[b] + rows of bucket b in earlier chunks + rows of bucket b before me in my chunk] = my row
__global__ void k_bucket_fill(const float *__restrict__ knorm, int64_t rows, const unsigned *__restrict__ kmax_bits,
                              int nchunks, const int32_t *__restrict__ bcnt_excl, const int32_t *__restrict__ bstart,
                              int32_t *__restrict__ perm, float *__restrict__ knorm_pos) {
    __shared__ int sb[256];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = row < rows ? key_bucket(knorm[row], __uint_as_float(kmax_bits[0])) : -1;
    sb[threadIdx.x] = b;
    __syncthreads();
    if (b < 0) return;
    int before = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) before += sb[t] == b;
    const int dst = bstart[b] + bcnt_excl[(int64_t)b * nchunks + blockIdx.x] + before;
    perm[dst] = (int32_t)row;
    knorm_pos[dst] = knorm[row];                      // the norms by position (the screening kernel's per-chunk bound)
}

// counts of observed cells per (column, chunk of 256 positions):  cnt[c][chunk]; perm: position -> row (nullptr: identity)
__global__ void k_count_obs(const uint8_t *__restrict__ m, int64_t rows, int C, int nchunks, int32_t *__restrict__ cnt,
                            const int32_t *__restrict__ perm = nullptr) {
    __shared__ int wsum[4];
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = (perm && pos < rows) ? perm[pos] : pos;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        const bool obs = pos < rows && m[row * C + c] != 0;
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

// qb_start[c] = sum_{c'<c} ceil((rows - obs[c'])/qpb)   (qpb = queries per workgroup of the selection kernel)
__global__ void k_qb_start(const int32_t *__restrict__ obs_q, int64_t rows, int C, int qpb,
                           int32_t *__restrict__ qb_start) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) {
            qb_start[c] = run;
            run += (int)(((int64_t)rows - obs_q[c] + qpb - 1) / qpb);
        }
        qb_start[C] = run;
    }
}

// write the lists.  WRITE_K: klist[c][base + pos] = row for observed rows.
// WRITE_Q: qlist[qb_start[c]*128 + (chunk_row0 - obs_before) + pos_missing] = row for missing rows.
template <bool WRITE_K, bool WRITE_Q>
__global__ void k_fill_lists(const uint8_t *__restrict__ m, int64_t rows, int C, int nchunks,
                             const int32_t *__restrict__ cnt_excl, int32_t *__restrict__ klist, int64_t kstride,
                             const int32_t *__restrict__ qb_start, int qpb, int32_t *__restrict__ qlist,
                             const int32_t *__restrict__ perm = nullptr) {         // perm: keys only, stored in position order
    __shared__ int wsum[4];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int64_t pos = row0 + threadIdx.x;
    const int64_t row = (perm && pos < rows) ? perm[pos] : pos;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        const bool inb = pos < rows;
        const bool obs = inb && m[row * C + c] != 0;
        const unsigned long long b = __ballot(obs);
        const int before_in_wave = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(b);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wsum[w];
        const int pos_obs = wbase + before_in_wave;                 // observed rows before me in the chunk
        const int obs_before = cnt_excl[(int64_t)c * nchunks + blockIdx.x];
        if (WRITE_K && obs) klist[(int64_t)c * kstride + obs_before + pos_obs] = (int32_t)pos;   // (= row without perm)
        if (WRITE_Q && inb && !obs) {
            const int64_t miss_before = row0 - obs_before;          // missing rows in earlier chunks
            const int pos_miss = threadIdx.x - pos_obs;             // missing rows before me in the chunk
            qlist[(int64_t)qb_start[c] * qpb + miss_before + pos_miss] = (int32_t)row;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// tie fallback of the register path: pairs whose k-th and (k+1)-th logits are equal (flag == 1)
// are compacted into a second query list, one block = 128 pairs of ONE column as before, and go
// through the MODE_FILL pass.  cnt2[c] = flagged pairs of column c, cur2[c] = append cursor.
// ---------------------------------------------------------------------------------------------
template <bool WRITE>
__global__ void k_compact_flagged(const int32_t *__restrict__ flag, const int32_t *__restrict__ qlist,
                                  const float *__restrict__ tau, const float *__restrict__ smax,
                                  const int32_t *__restrict__ qb_start, int C, int32_t *__restrict__ cnt2,
                                  int32_t *__restrict__ cur2, const int32_t *__restrict__ qb_start2,
                                  int32_t *__restrict__ qlist2, float *__restrict__ tau2, float *__restrict__ smax2) {
    __shared__ int s_base, s_n;
    if ((int)blockIdx.x >= qb_start[C]) return;
    const int c = block_column(qb_start, C, blockIdx.x);
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // blockDim.x == queries per source block
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
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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

// one compacted list of open pairs: 128 pairs of one column per workgroup
struct PairList {
    int32_t *qlist, *qb_start, *cnt, *flag;   // cnt: [C] counts then [C] cursors
    float *tau, *smax;
};

struct AttnWs {
    float *Xkp, *Xqp, *Wp;
    __bf16 *Xkh;               // bf16 keys (k <= DDG_MAX_TOPK_REG) or their two-half f16 split (larger k): one region
    int32_t *klist, *nk, *obs_q, *cnt_k, *cnt_q, *offsets;
    int32_t *cnt_q_own;        // the query-side chunk counts' own storage (cnt_q aliases cnt_k when queries alias keys)
    float *knorm, *knorm_row;  // key order of the screening path: 2-norms (by position / row), position -> row, bucket counts / totals / starts
    int32_t *perm, *bcnt, *btot;
    uint32_t *prefix;
    int32_t *rank;
    int32_t *flags;            // [0] = 1 when queries alias keys (set by prepare); [1] = bits of max key norm
    PairList L1;               // primary list (128 or 256 pairs per workgroup, by mode)
    PairList L2, L3;           // fallback lists (128 per workgroup)
    float *st_val;             // parked lists of the screening path, FILT_GROUP workgroups
    int32_t *st_pos, *cand_pos, *cand_cnt, *st_ovf;
    int32_t *lpos;             // window path (k > DDG_MAX_TOPK_REG): listed keys of WIN_GROUP workgroups
    float *lval;
    size_t total;
    int64_t slots;             // capacity of every per-pair array
    int nchunk_k, nchunk_q;
};

#define FILT_GROUP (16384 * 8 / FILT_NW)       // workgroups per launch group of the filter kernel (bounds the parked state)
// Key chunks of the screening path: [WARMK, 2 WARMK), then doubling up to FILT_CHUNK_MAX positions.  A
// chunk is screened against the thresholds the queries had when it started, so a chunk as long as the
// prefix before it yields about k+1 candidates per query; the cap keeps a chunk's bf16 rows inside the
// Infinity Cache.  DDG_FILT_CHUNK (multiple of 128) forces a fixed chunk length: tests use a small one.
#define FILT_CHUNK_MAX 262144       // <= 65536 tiles of 32 keys: the tile number of a candidate word has 16 bits
// DDG_FILT_CHUNK in the environment is read ONCE per process (production calls never touch getenv again);
// ddg_attn_set_key_chunk overrides it from code (the parity tests switch schedules inside one process).
static int64_t filt_chunk_round(int64_t c) { return c <= 0 ? 0 : (c < 128 ? 128 : (c / 128) * 128); }
static std::atomic<int64_t> g_chunk_override{-1};
static int64_t filt_chunk_fixed() {
    static const int64_t once = [] { const char *e = getenv("DDG_FILT_CHUNK"); return filt_chunk_round(e ? atoll(e) : 0); }();
    const int64_t o = g_chunk_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : once;
}
extern "C" int ddg_attn_set_key_chunk(int64_t positions) {
    g_chunk_override.store(positions < 0 ? -1 : filt_chunk_round(positions), std::memory_order_relaxed);
    return DDG_OK;
}

// DDG_ATTN_RADIX=1: k > DDG_MAX_TOPK_REG goes through the 5-level radix select (six full passes) as in round 1;
// DDG_WIN_CAP=n (tests): smaller lists, so that pairs overflow into the radix fallback.  Both read per call: the
// path they steer runs for seconds.
// Tables of fewer than WIN_MIN_KEYS rows keep the radix path as well (DDG_ATTN_WINDOW=1 overrides, tests): six passes
// cost milliseconds there, and its accumulation order is the one whose rounding reproduces the reference's votes on
// the exactly tied categorical cells of tests/golden/e2e_hospital (a different summation order flips such an arg-max).
#define WIN_MIN_KEYS (2 * WIN_SAMPLE)
static bool window_path(int64_t Nk) {
    const char *e = getenv("DDG_ATTN_RADIX");
    if (e && e[0] == '1') return false;
    const char *f = getenv("DDG_ATTN_WINDOW");
    return Nk >= WIN_MIN_KEYS || (f && f[0] == '1');
}
// DDG_WIN_F32=1: the window passes on the fp32 MFMA chain (the radix path's arithmetic) instead of the f16 split
static bool window_f32() { const char *e = getenv("DDG_WIN_F32"); return e && e[0] == '1'; }
static int win_cap() {
    const char *e = getenv("DDG_WIN_CAP");
    const int c = e ? atoi(e) : WIN_CAP;
    return c < 1 ? 1 : (c > WIN_CAP ? WIN_CAP : c);
}

// DDG_KEY_ORDER=row (read once): the screening path's key lists in row order like the other paths' (A/B)
static bool key_order_by_norm() {
    static const bool row_order = [] { const char *e = getenv("DDG_KEY_ORDER"); return e && e[0] == 'r'; }();
    return !row_order;
}

static void attn_carve(AttnWs &w, char *base, int64_t Nq, int64_t Nk, int C, int D, int64_t k, bool same) {
    const int DP = ks4_for(D) * 8;
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += al256(bytes); return p; };
    w.nchunk_k = (int)ceil_div64(Nk, 256);
    w.nchunk_q = (int)ceil_div64(Nq, 256);
    w.slots = Nq * C + (int64_t)QPBF * C;       // pairs + per-column padding at either block size
    w.Xkp = (float *)take((size_t)Nk * DP * 4);
    w.Xqp = (float *)take((size_t)Nq * DP * 4);
    // (with norm-ordered keys the queries keep a padded copy of their own, in row order)
    if (same && !(k <= DDG_MAX_TOPK_REG && key_order_by_norm())) w.Xqp = w.Xkp;
    w.Xkh = (__bf16 *)take((size_t)Nk * DP * (k > DDG_MAX_TOPK_REG ? 4 : 2));
    w.Wp = (float *)take((size_t)C * DP * 4);
    w.klist = (int32_t *)take((size_t)C * Nk * 4);
    w.nk = (int32_t *)take((size_t)C * 4);
    w.obs_q = (int32_t *)take((size_t)C * 4);
    w.cnt_k = (int32_t *)take((size_t)C * w.nchunk_k * 4);
    w.cnt_q = (int32_t *)take((size_t)C * w.nchunk_q * 4);
    w.cnt_q_own = w.cnt_q;
    if (same) w.cnt_q = w.cnt_k;
    {
        const bool reg = k <= DDG_MAX_TOPK_REG;          // the screening path exists for the register-list sizes only
        w.knorm = (float *)take(reg ? (size_t)Nk * 4 : 0);               // by position
        w.knorm_row = (float *)take(reg ? (size_t)Nk * 4 : 0);           // by row
        w.perm = (int32_t *)take(reg ? (size_t)Nk * 4 : 0);
        w.bcnt = (int32_t *)take(reg ? (size_t)KEY_BUCKETS * w.nchunk_k * 4 : 0);
        w.btot = (int32_t *)take(reg ? (size_t)2 * KEY_BUCKETS * 4 : 0);
    }
    w.offsets = (int32_t *)take((size_t)(C + 1) * 4);
    w.prefix = (uint32_t *)take((size_t)w.slots * 4);
    w.rank = (int32_t *)take((size_t)w.slots * 4);
    w.flags = (int32_t *)take(256);
    PairList *ls[3] = {&w.L1, &w.L2, &w.L3};
    for (PairList *l : ls) {
        l->qlist = (int32_t *)take((size_t)w.slots * 4);
        l->tau = (float *)take((size_t)w.slots * 4);
        l->smax = (float *)take((size_t)w.slots * 4);
        l->flag = (int32_t *)take((size_t)w.slots * 4);
        l->qb_start = (int32_t *)take((size_t)(C + 1) * 4);
        l->cnt = (int32_t *)take((size_t)2 * C * 4);
    }
    {
        const int64_t kk = k > DDG_MAX_TOPK_REG ? 0 : NLIST(k < 1 ? 1 : k);
        const int64_t grp = w.slots / QPBF + C < FILT_GROUP ? w.slots / QPBF + C : FILT_GROUP;
        const size_t st = (size_t)grp * FILT_NT * kk * 4;
        w.st_val = (float *)take(st);
        w.st_pos = (int32_t *)take(st);
        w.cand_pos = (int32_t *)take(kk ? (size_t)grp * FILT_CAPH * FILT_NT * 4 : 0);
        w.cand_cnt = (int32_t *)take(kk ? (size_t)grp * FILT_NT * 4 : 0);
        w.st_ovf = (int32_t *)take(kk ? (size_t)grp * FILT_NT * 4 : 0);
    }
    {
        const int64_t nb = w.slots / QPB + C;
        const size_t ls = k > DDG_MAX_TOPK_REG ? (size_t)(nb < WIN_GROUP ? nb : WIN_GROUP) * QPB * WIN_CAP * 4 : 0;
        w.lpos = (int32_t *)take(ls);
        w.lval = (float *)take(ls);
    }
    w.total = o;
}

extern "C" size_t ddg_attn_workspace_bytes(int64_t Nq, int64_t Nk, int C, int D, int64_t k) {
    if (ks4_for(D) == 0 || Nq < 0 || Nk < 0 || C < 1) return 0;
    AttnWs w;
    attn_carve(w, nullptr, Nq, Nk, C, D, k > Nk ? Nk : k, false);   // the layout does not depend on aliasing or mode
    return w.total;
}

// `granted`: the largest dynamic-LDS size already opted into for this kernel (per kernel instantiation; atomic: entry
// points may be called from several host threads -- setting the attribute twice is harmless, a torn read is not)
static int set_lds_attr(const void *fn, size_t lds, std::atomic<size_t> &granted) {
    if (lds > 160 * 1024 - 64) return DDG_E_UNSUPPORTED;
    if (lds > granted.load(std::memory_order_acquire)) {   // dynamic LDS above the default limit must be opted into
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            fprintf(stderr, "ddgani: hipFuncSetAttribute(%zu): %s\n", lds, hipGetErrorString(e));
            return DDG_E_LAUNCH;
        }
        granted.store(lds, std::memory_order_release);
    }
    return DDG_OK;
}

template <int KS4, int MODE, int KCAP, bool SPLIT = false, int PIPE = 0>
static int launch_attn(const AttnArgs &a, int64_t nblocks_max, size_t extra_bytes, hipStream_t s) {
    const size_t lds = (size_t)2 * TK * (KS4 * 8 + 4) * 4 + extra_bytes;
    static std::atomic<size_t> granted{48 * 1024};
    int rc = set_lds_attr((const void *)k_attn<KS4, MODE, KCAP, SPLIT, PIPE>, lds, granted);
    if (rc) return rc;
    k_attn<KS4, MODE, KCAP, SPLIT, PIPE><<<(unsigned)nblocks_max, 256, lds, s>>>(a);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
// split window pass, pipelined (pipe = 1 + padding steps to skip, at most 1) or plain (pipe = 0)
template <int KS4, int MODE>
static int launch_split(int pipe, const AttnArgs &a, int64_t nblocks_max, size_t extra_bytes, hipStream_t s) {
    if (pipe == 1) return launch_attn<KS4, MODE, 1, true, 1>(a, nblocks_max, extra_bytes, s);
    if constexpr (KS4 >= 4) {
        if (pipe == 2) return launch_attn<KS4, MODE, 1, true, 2>(a, nblocks_max, extra_bytes, s);
    }
    return launch_attn<KS4, MODE, 1, true, 0>(a, nblocks_max, extra_bytes, s);
}
// DDG_WIN_PIPE=0: plain loops (A/B; the logits are bit-identical either way)
static int window_pipe(int KS4, int D) {
    static const bool off = [] { const char *e = getenv("DDG_WIN_PIPE"); return e && e[0] == '0'; }();
    if (off) return 0;
    return (KS4 >= 4 && KS4 / 2 - (D + 15) / 16 >= 1) ? 2 : 1;
}

// one instantiation of the screening kernel: opt into its LDS size once, then launch
template <int KS, int QW, int KSC>
static int screen_one(const AttnArgs &b, int64_t gn, bool prepare, hipStream_t s) {
    const size_t lds = (size_t)3 * screen_stage_keys(KS, QW) * (KS * 32);
    static std::atomic<size_t> granted{48 * 1024};
    if (prepare) return set_lds_attr((const void *)k_attn_screen<KS, QW, KSC>, lds, granted);
    k_attn_screen<KS, QW, KSC><<<(unsigned)gn, FILT_NT / QW, lds, s>>>(b);
    return DDG_OK;
}
// pad = 16-deep steps of the row stride that hold only zero padding (0 .. 3; rows are padded to 64 elements)
template <int KS>
static int screen_launch(int qw, int pad, const AttnArgs &b, int64_t gn, bool prepare, hipStream_t s) {
    if (qw == 2) {
        if constexpr (KS == 8) {
            switch (pad) {
            case 1: return screen_one<KS, 2, KS - 1>(b, gn, prepare, s);
            case 2: return screen_one<KS, 2, KS - 2>(b, gn, prepare, s);
            case 3: return screen_one<KS, 2, KS - 3>(b, gn, prepare, s);
            }
        }
        return screen_one<KS, 2, KS>(b, gn, prepare, s);
    }
    switch (pad) {
    case 1: return screen_one<KS, 1, KS - 1>(b, gn, prepare, s);
    case 2: return screen_one<KS, 1, KS - 2>(b, gn, prepare, s);
    case 3: return screen_one<KS, 1, KS - 3>(b, gn, prepare, s);
    }
    return screen_one<KS, 1, KS>(b, gn, prepare, s);
}

template <int KS, int KCAP>
static int launch_filt(const AttnArgs &a, int64_t nblocks_max, hipStream_t s) {
    // 64 queries per wave (four-wave workgroups, every key fragment feeds two MFMAs): the default for 256-byte key rows
    // (D in (64, 128]: measured 5.6 % faster at the c4 shape -- half-size stages keep two waves per SIMD there); for wider
    // rows only one workgroup fits a CU and it is slower.  DDG_SCREEN_QW=1 / 2 overrides.
    static const int qw = [] {
        const char *e = getenv("DDG_SCREEN_QW");
        return (e && e[0] == '2') ? 2 : (e && e[0] == '1') ? 1 : (KS == 8 ? 2 : 1);
    }();
    const size_t lds_w = (size_t)2 * TK * (KS * 16 + 4) * 4;
    static std::atomic<size_t> granted_w{48 * 1024};
    static const bool nopad = [] { const char *e = getenv("DDG_SCREEN_PAD"); return e && e[0] == '1'; }();   // A/B: multiply the padding too
    int pad = nopad ? 0 : KS - (a.D + 15) / 16;
    if (pad < 0 || pad > 3) pad = 0;
    int rc = screen_launch<KS>(qw, pad, a, 0, true, s);
    if (rc) return rc;
    if ((rc = set_lds_attr((const void *)k_attn_warm<KS * 2, KCAP>, lds_w, granted_w))) return rc;
    // groups of workgroups (outer, bounds the parked state) x key chunks (inner).  Per chunk: the bf16
    // screening pass, then the exact fp32 pass over its candidates.
    AttnArgs b = a;
    const int64_t fixed = filt_chunk_fixed();
    // growth of the chunk length in percent (200: every chunk as long as everything before it)
    static const int64_t growth = [] { const char *e = getenv("DDG_FILT_GROWTH"); const int64_t g = e ? atoll(e) : 200; return g < 110 ? 110 : g; }();
    for (int64_t g0 = 0; g0 < nblocks_max; g0 += FILT_GROUP) {
        const int64_t gn = nblocks_max - g0 < FILT_GROUP ? nblocks_max - g0 : FILT_GROUP;
        b.blk_lo = (int)g0;
        b.key_lo = 0; b.key_hi = WARMK;
        k_attn_warm<KS * 2, KCAP><<<(unsigned)gn, FILT_NT, lds_w, s>>>(b);          // positions [0, WARMK), exact fp32
        int64_t len = fixed ? fixed : WARMK;
        auto next_len = [&](int64_t l) {
            if (fixed) return fixed;
            int64_t n = (l * growth / 100 / 128) * 128;
            if (n <= l) n = l + 128;
            return n < FILT_CHUNK_MAX ? n : (int64_t)FILT_CHUNK_MAX;
        };
        for (int64_t k0 = WARMK; k0 < a.Nk; k0 += len, len = next_len(len)) {
            b.key_lo = (int)k0;
            b.key_hi = (int)(k0 + len < a.Nk ? k0 + len : ((int64_t)1 << 31) - 1);
            screen_launch<KS>(qw, pad, b, gn, false, s);
            k_attn_rescore<KS, KCAP><<<(unsigned)gn, FILT_NT, 0, s>>>(b);
        }
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

template <int KS4>
static int run_select(AttnArgs &a, int64_t nb, int mode, hipStream_t s) {
    int rc;
    if (a.k <= DDG_MAX_TOPK_REG && mode == DDG_ATTN_BF16_FILTER) {
        if (NLIST(a.k) <= 16) return launch_filt<KS4 / 2, 16>(a, nb, s);
        return launch_filt<KS4 / 2, 32>(a, nb, s);
    }
    if (NLIST(a.k) <= 16) return launch_attn<KS4, MODE_TOPK, 16>(a, nb, 0, s);
    if (a.k <= DDG_MAX_TOPK_REG) return launch_attn<KS4, MODE_TOPK, 32>(a, nb, 0, s);
    if (mode == ATTN_WINDOW) {
        // sample: range, then two bracketing histograms (the second 128x finer); full: the bracket's histogram
        const size_t ex = (size_t)QPB * WHSTRIDE * 4;
        const bool f32 = window_f32();
        auto pass = [&](int level, int sample) {
            a.level = level;
            a.sample = sample;
            if (f32) return launch_attn<KS4, MODE_WIN, 1, false>(a, nb, ex, s);
            return launch_split<KS4, MODE_WIN>(level == WIN_MINMAX ? 0 : window_pipe(KS4, a.D), a, nb, ex, s);
        };
        if ((rc = pass(WIN_MINMAX, 1))) return rc;
        if ((rc = pass(WIN_SHIST, 1))) return rc;
        if ((rc = pass(WIN_SHIST, 1))) return rc;
        return pass(WIN_FULL, 0);
    }
    for (int L = 0; L < 5; ++L) {
        a.level = L;
        rc = launch_attn<KS4, MODE_HIST, 1>(a, nb, (size_t)QPB * HSTRIDE * 4, s);
        if (rc) return rc;
    }
    return DDG_OK;
}

template <int KS4>
static int run_fill(AttnArgs &a, int64_t nb, int dcmax, bool window, hipStream_t s) {
    a.dcs = dcmax | 1;
    if (window) {      // second full pass of the window path + the in-list selection, group by group
        AttnArgs b = a;
        for (int64_t g0 = 0; g0 < nb; g0 += WIN_GROUP) {
            const int64_t gn = nb - g0 < WIN_GROUP ? nb - g0 : WIN_GROUP;
            b.blk_lo = (int)g0;
            int rc = window_f32() ? launch_attn<KS4, MODE_WFILL, 1, false>(b, gn, (size_t)(QPB + 256 * a.dcs) * 4, s)
                                  : launch_split<KS4, MODE_WFILL>(dcmax <= 8 ? window_pipe(KS4, a.D) : 0, b, gn,
                                                                  (size_t)(QPB + 768 + 256 * a.dcs) * 4, s);
            if (rc) return rc;
            k_attn_wlist<<<(unsigned)gn, 256, 0, s>>>(b);
        }
        DDG_CHECK_LAUNCH();
        return DDG_OK;
    }
    return launch_attn<KS4, MODE_FILL, 1>(a, nb, (size_t)256 * a.dcs * 4, s);
}

static int dispatch_select(AttnArgs &a, int64_t nb, int mode, int D, hipStream_t s) {
    switch (ks4_for(D)) {
        case 8: return run_select<8>(a, nb, mode, s);
        case 16: return run_select<16>(a, nb, mode, s);
        case 24: return run_select<24>(a, nb, mode, s);
        case 32: return run_select<32>(a, nb, mode, s);
    }
    return DDG_E_UNSUPPORTED;
}

static int dispatch_fill(AttnArgs &a, int64_t nb, int dcmax, int D, hipStream_t s, bool window = false) {
    switch (ks4_for(D)) {
        case 8: return run_fill<8>(a, nb, dcmax, window, s);
        case 16: return run_fill<16>(a, nb, dcmax, window, s);
        case 24: return run_fill<24>(a, nb, dcmax, window, s);
        case 32: return run_fill<32>(a, nb, dcmax, window, s);
    }
    return DDG_E_UNSUPPORTED;
}

static int attn_check_dims(int64_t Nq, int64_t Nk, int C, int D, int mode) {
    if (Nq < 0 || Nk < 1 || C < 1 || D < 1) return DDG_E_ARG;
    if (mode != DDG_ATTN_F32 && mode != DDG_ATTN_BF16_FILTER) return DDG_E_ARG;
    if (ks4_for(D) == 0 || Nk >= (int64_t)1 << 31 || (Nq * C) / QPB + C >= (int64_t)1 << 31) return DDG_E_UNSUPPORTED;
    if (mode == DDG_ATTN_BF16_FILTER && Nk >= (int64_t)1 << 27) return DDG_E_UNSUPPORTED;   // 32-bit row offsets / 16 in k_attn_screen
    return DDG_OK;
}

// pairs per workgroup of the primary list
static inline int primary_qpb(int mode, int64_t k) {
    return (mode == DDG_ATTN_BF16_FILTER && k <= DDG_MAX_TOPK_REG) ? QPBF : QPB;
}
static inline int64_t blocks_max(const AttnWs &w, int C, int qpb) { return w.slots / qpb + C; }

static void attn_args(AttnArgs &a, const AttnWs &w, const PairList &l, int64_t Nk, int C, int D, int k) {
    a.Xks = reinterpret_cast<const _Float16 *>(w.Xkh); a.kscale = reinterpret_cast<const float *>(w.flags + 3);
    a.Xkp = w.Xkp; a.Xqp = w.Xqp; a.Wp = w.Wp; a.Xkh = w.Xkh; a.kmax = reinterpret_cast<const float *>(w.flags + 1);
    a.knorm = (w.knorm && key_order_by_norm()) ? w.knorm : nullptr;          // (read by the screening kernel only)
    a.st_val = w.st_val; a.st_pos = w.st_pos; a.cand_pos = w.cand_pos; a.cand_cnt = w.cand_cnt; a.st_ovf = w.st_ovf; a.key_lo = 0; a.key_hi = 0x7fffffff; a.blk_lo = 0;
    a.klist = w.klist; a.nk = w.nk; a.qlist = l.qlist; a.qb_start = l.qb_start; a.offsets = w.offsets;
    a.tau = l.tau; a.smax = l.smax; a.flag = l.flag; a.prefix = w.prefix; a.rank = w.rank;
    a.out_code = nullptr; a.out_tau = nullptr; a.out_nvalid = nullptr; a.Nk = Nk; a.C = C;
    a.D = D; a.DP = ks4_for(D) * 8; a.k = k; a.level = 0; a.dcs = 1;
    // window-path state lives in arrays the k > DDG_MAX_TOPK_REG paths do not use otherwise (L3, prefix)
    a.wlo = w.L3.tau; a.ww = w.L3.smax; a.wst = w.L3.flag; a.nab = w.L3.qlist;
    a.zab = reinterpret_cast<float *>(w.prefix); a.lpos = w.lpos; a.lval = w.lval; a.sample = 0; a.wcap = win_cap();
}

// flagged pairs of `src` (workgroups of qpb_src pairs) -> `dst` (workgroups of 128)
static int compact_pairs(const AttnWs &w, const PairList &src, int qpb_src, const PairList &dst, int C, hipStream_t s) {
    const unsigned g = (unsigned)blocks_max(w, C, qpb_src);
    if (hipMemsetAsync(dst.cnt, 0, (size_t)2 * C * 4, s) != hipSuccess) return DDG_E_LAUNCH;
    if (hipMemsetAsync(dst.qlist, 0xFF, (size_t)w.slots * 4, s) != hipSuccess) return DDG_E_LAUNCH;
    k_compact_flagged<false><<<g, qpb_src, 0, s>>>(src.flag, src.qlist, src.tau, src.smax, src.qb_start, C, dst.cnt,
                                                   dst.cnt + C, dst.qb_start, dst.qlist, dst.tau, dst.smax);
    k_qb_start2<<<1, 64, 0, s>>>(dst.cnt, C, dst.qb_start);
    k_compact_flagged<true><<<g, qpb_src, 0, s>>>(src.flag, src.qlist, src.tau, src.smax, src.qb_start, C, dst.cnt,
                                                  dst.cnt + C, dst.qb_start, dst.qlist, dst.tau, dst.smax);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// phase 1: padded operands (+ bf16 keys and their largest norm in filter mode) and the per-column
// key / query index lists
extern "C" int ddg_attn_prepare(const float *Xq0, const uint8_t *mq, int64_t q_row_offset, const float *Xk0,
                                const uint8_t *mk, const float *W, const int32_t *offsets_host, int64_t Nq,
                                int64_t Nk, int C, int D, int64_t k, int mode, void *workspace,
                                size_t workspace_bytes, void *stream) {
    if (Nq == 0 && Nk >= 0 && C >= 1 && k >= 1) return DDG_OK;       // an empty query shard (its pointers may be NULL)
    if (!Xq0 || !mq || !Xk0 || !mk || !W || !offsets_host || k < 1) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D, mode);
    if (rc) return rc;
    if (q_row_offset < 0 || q_row_offset + Nq > Nk) return DDG_E_ARG;
    if (offsets_host[0] != 0 || offsets_host[C] != D) return DDG_E_ARG;
    for (int c = 0; c < C; ++c)
        if (offsets_host[c + 1] - offsets_host[c] < 1) return DDG_E_ARG;
    if (Nq == 0) return DDG_OK;
    const bool same = (Xq0 == Xk0 && mq == mk && Nq == Nk);
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, k > Nk ? Nk : k, same);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int DP = ks4_for(D) * 8;
    const int qpb = primary_qpb(mode, k > Nk ? Nk : k);

    if (hipMemcpyAsync(w.offsets, offsets_host, (size_t)(C + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        return DDG_E_LAUNCH;
    if (hipMemsetAsync(w.L1.qlist, 0xFF, (size_t)w.slots * 4, s) != hipSuccess) return DDG_E_LAUNCH;
    if (hipMemsetAsync(w.flags, 0, 16, s) != hipSuccess) return DDG_E_LAUNCH;
    if (same && hipMemsetAsync(w.flags, 1, 1, s) != hipSuccess) return DDG_E_LAUNCH;

    auto egrid = [](int64_t n) { int64_t g = ceil_div64(n, 256); return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); };
    const int32_t *perm = nullptr;
    if (qpb == QPBF && key_order_by_norm()) {
        // screening path: the padded key rows in descending-norm order (see k_bucket_count)
        unsigned *kmax = reinterpret_cast<unsigned *>(w.flags + 1);
        k_row_norms<<<(unsigned)ceil_div64(Nk, 4), 256, 0, s>>>(Xk0, Nk, D, w.knorm_row, kmax);
        k_bucket_count<<<w.nchunk_k, 256, 0, s>>>(w.knorm_row, Nk, kmax, w.nchunk_k, w.bcnt);
        k_scan_chunks<<<KEY_BUCKETS, 256, 0, s>>>(w.bcnt, w.nchunk_k, w.btot);
        k_bucket_starts<<<1, 64, 0, s>>>(w.btot, w.btot + KEY_BUCKETS);
        k_bucket_fill<<<w.nchunk_k, 256, 0, s>>>(w.knorm_row, Nk, kmax, w.nchunk_k, w.bcnt, w.btot + KEY_BUCKETS, w.perm,
                                                 w.knorm);
        perm = w.perm;
        k_pad_rows_perm<<<egrid(Nk * DP), 256, 0, s>>>(Xk0, w.Xkp, Nk, D, DP, perm);
    } else {
        k_pad_rows<<<egrid(Nk * DP), 256, 0, s>>>(Xk0, w.Xkp, Nk, D, DP);
    }
    if (w.Xqp != w.Xkp) k_pad_rows<<<egrid(Nq * DP), 256, 0, s>>>(Xq0, w.Xqp, Nq, D, DP);     // (aliased: see attn_carve)
    k_pad_rows<<<egrid((int64_t)C * DP), 256, 0, s>>>(W, w.Wp, C, D, DP);
    if (qpb == QPBF)
    {
        k_maxabs<<<egrid(Nk * DP), 256, 0, s>>>(w.Xkp, Nk * DP, reinterpret_cast<unsigned *>(w.flags + 2));
        // (norm-ordered keys: the largest norm is k_row_norms' and must stay EXACTLY the value the buckets were cut with --
        // the screening kernel recomputes a key's bucket from it; a last-bit difference from another summation order
        // could move a key at a bucket's edge and make the per-chunk bound one bucket too small)
        k_keys_screen<<<egrid(Nk * 64), 256, 0, s>>>(w.Xkp, Nk, DP, w.Xkh, perm ? nullptr : reinterpret_cast<unsigned *>(w.flags + 1),
                                                     reinterpret_cast<const unsigned *>(w.flags + 2),
                                                     reinterpret_cast<float *>(w.flags + 3));
    }
    if ((k > Nk ? Nk : k) > DDG_MAX_TOPK_REG && window_path(Nk) && !window_f32()) {
        k_maxabs<<<egrid(Nk * DP), 256, 0, s>>>(w.Xkp, Nk * DP, reinterpret_cast<unsigned *>(w.flags + 2));
        k_keys_split<<<egrid(Nk * DP), 256, 0, s>>>(w.Xkp, Nk, DP, reinterpret_cast<_Float16 *>(w.Xkh),
                                                    reinterpret_cast<const unsigned *>(w.flags + 2),
                                                    reinterpret_cast<float *>(w.flags + 3));
    }

    k_count_obs<<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, perm);
    k_scan_chunks<<<C, 256, 0, s>>>(w.cnt_k, w.nchunk_k, w.nk);
    if (same && perm) {
        // keys in norm order, queries in row order: two passes over the one mask, the query side with counts of its own
        k_count_obs<<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q_own);
        k_scan_chunks<<<C, 256, 0, s>>>(w.cnt_q_own, w.nchunk_q, w.obs_q);
        k_qb_start<<<1, 64, 0, s>>>(w.obs_q, Nq, C, qpb, w.L1.qb_start);
        k_fill_lists<true, false><<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, w.klist, Nk,
                                                             w.L1.qb_start, qpb, w.L1.qlist, perm);
        k_fill_lists<false, true><<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q_own, w.klist, Nk,
                                                             w.L1.qb_start, qpb, w.L1.qlist);
    } else if (same) {
        k_qb_start<<<1, 64, 0, s>>>(w.nk, Nq, C, qpb, w.L1.qb_start);
        k_fill_lists<true, true><<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, w.klist, Nk, w.L1.qb_start,
                                                            qpb, w.L1.qlist);
    } else {
        k_count_obs<<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q);
        k_scan_chunks<<<C, 256, 0, s>>>(w.cnt_q, w.nchunk_q, w.obs_q);
        k_qb_start<<<1, 64, 0, s>>>(w.obs_q, Nq, C, qpb, w.L1.qb_start);
        k_fill_lists<true, false><<<w.nchunk_k, 256, 0, s>>>(mk, Nk, C, w.nchunk_k, w.cnt_k, w.klist, Nk,
                                                             w.L1.qb_start, qpb, w.L1.qlist, perm);
        k_fill_lists<false, true><<<w.nchunk_q, 256, 0, s>>>(mq, Nq, C, w.nchunk_q, w.cnt_q, w.klist, Nk,
                                                             w.L1.qb_start, qpb, w.L1.qlist);
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// phase 2: one similarity pass over every open pair.  out_code <- Xq0, then
//   k <= DDG_MAX_TOPK_REG: the k+1 largest logits and their key rows per pair are kept in registers
//     (fp32 MFMA, or bf16 MFMA screening + exact fp32 re-scoring in DDG_ATTN_BF16_FILTER mode); pairs
//     without a tie at the k-th logit are filled right here, the others are flagged for phase 3;
//   larger k: 5-level radix select of tau (5 fp32 similarity passes), every pair is filled in phase 3.
extern "C" int ddg_attn_select(const float *Xq0, int64_t Nq, int64_t Nk, int C, int D, int64_t k, int mode,
                               int queries_alias_keys, float *out_code, void *workspace, size_t workspace_bytes,
                               void *stream) {
    if (Nq == 0 && Nk >= 0 && C >= 1 && k >= 1) return DDG_OK;       // an empty query shard (its pointers may be NULL)
    if (!Xq0 || !out_code) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D, mode);
    if (rc) return rc;
    if (k < 1) return DDG_E_ARG;
    if (Nq == 0) return DDG_OK;
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, k > Nk ? Nk : k, queries_alias_keys != 0);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (out_code != Xq0 &&
        hipMemcpyAsync(out_code, Xq0, (size_t)Nq * D * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return DDG_E_LAUNCH;
    AttnArgs a;
    const int kk = (int)(k > Nk ? Nk : k);
    attn_args(a, w, w.L1, Nk, C, D, kk);
    a.out_code = out_code;
    if (kk > DDG_MAX_TOPK_REG && window_path(Nk)) mode = ATTN_WINDOW;
    return dispatch_select(a, blocks_max(w, C, primary_qpb(mode, kk)), mode, D, s);
}

// phase 3: the pairs phase 2 left open.
//   radix path: every pair, softmax-weighted accumulation over {s >= tau} (one more similarity pass);
//   register path, fp32: tie-flagged pairs -> compacted list -> threshold pass;
//   register path, bf16 filter: flagged pairs -> compacted list -> fp32 register kernel (its tau and
//     logits come from one arithmetic) -> its flagged pairs -> second compacted list -> threshold pass.
extern "C" int ddg_attn_fill(const int32_t *offsets_host, int64_t Nq, int64_t Nk, int C, int D, int64_t k, int mode,
                             int queries_alias_keys, float *out_code, float *out_tau, int32_t *out_nvalid,
                             void *workspace, size_t workspace_bytes, void *stream) {
    if (Nq == 0 && Nk >= 0 && C >= 1 && k >= 1) return DDG_OK;       // an empty query shard (its pointers may be NULL)
    if (!offsets_host || !out_code) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D, mode);
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
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, k > Nk ? Nk : k, queries_alias_keys != 0);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    AttnArgs a;
    const int kk = (int)(k > Nk ? Nk : k);
    const int64_t nb128 = blocks_max(w, C, QPB);
    if (kk > DDG_MAX_TOPK_REG) {
        attn_args(a, w, w.L1, Nk, C, D, kk);
        a.out_code = out_code; a.out_tau = out_tau; a.out_nvalid = out_nvalid;
        if (!window_path(Nk)) return dispatch_fill(a, nb128, dcmax, D, s);
        if ((rc = dispatch_fill(a, nb128, dcmax, D, s, true))) return rc;
        // pairs whose bracket missed the k-th logit (or whose bin overflowed the list): radix select + threshold pass
        if ((rc = compact_pairs(w, w.L1, QPB, w.L2, C, s))) return rc;
        attn_args(a, w, w.L2, Nk, C, D, kk);
        a.out_code = out_code; a.out_tau = out_tau; a.out_nvalid = out_nvalid;
        if ((rc = dispatch_select(a, nb128, DDG_ATTN_F32, D, s))) return rc;
        return dispatch_fill(a, nb128, dcmax, D, s);
    }
    const int qpb = primary_qpb(mode, kk);
    if (out_tau || out_nvalid) {
        k_write_debug<<<(unsigned)blocks_max(w, C, qpb), qpb, 0, s>>>(w.L1.qlist, w.L1.tau, w.L1.qb_start, w.nk, C,
                                                                     out_tau, out_nvalid);
        DDG_CHECK_LAUNCH();
    }
    if ((rc = compact_pairs(w, w.L1, qpb, w.L2, C, s))) return rc;
    const PairList *open = &w.L2;
    if (qpb == QPBF) {
        attn_args(a, w, w.L2, Nk, C, D, kk);
        a.out_code = out_code;
        if ((rc = dispatch_select(a, nb128, DDG_ATTN_F32, D, s))) return rc;
        if (out_tau) {      // pairs that overflowed their candidate slots parked an incomplete list: tau from the fp32 pass
            k_write_debug<<<(unsigned)nb128, QPB, 0, s>>>(w.L2.qlist, w.L2.tau, w.L2.qb_start, w.nk, C, out_tau, nullptr);
            DDG_CHECK_LAUNCH();
        }
        if ((rc = compact_pairs(w, w.L2, QPB, w.L3, C, s))) return rc;
        open = &w.L3;
    }
    attn_args(a, w, *open, Nk, C, D, kk);
    a.out_code = out_code;
    return dispatch_fill(a, nb128, dcmax, D, s);
}

// How many pairs the last ddg_attn_fill on this workspace sent to its exact fp32 fallback (window path: bracket missed or
// bin above the list capacity; register paths: ties longer than the list / candidate-slot overflow): the sum of the
// compacted list's per-column counts, written to out_count_dev[0] on the stream.
__global__ void k_sum_counts(const int32_t *__restrict__ cnt, int C, int32_t *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int c = 0; c < C; ++c) s += cnt[c];
        out[0] = s;
    }
}
extern "C" int ddg_attn_fallback_pairs(int64_t Nq, int64_t Nk, int C, int D, int64_t k, int queries_alias_keys,
                                       int32_t *out_count_dev, void *workspace, size_t workspace_bytes, void *stream) {
    if (!out_count_dev || k < 1) return DDG_E_ARG;
    int rc = attn_check_dims(Nq, Nk, C, D, DDG_ATTN_F32);
    if (rc) return rc;
    AttnWs w;
    attn_carve(w, (char *)workspace, Nq, Nk, C, D, k > Nk ? Nk : k, queries_alias_keys != 0);
    if (!workspace || workspace_bytes < w.total) return DDG_E_WORKSPACE;
    k_sum_counts<<<1, 64, 0, (hipStream_t)stream>>>(w.L2.cnt, C, out_count_dev);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_attn_topk_fill(const float *Xq0, const uint8_t *mq, int64_t q_row_offset, const float *Xk0,
                                  const uint8_t *mk, const float *W, const int32_t *offsets_host, int64_t Nq,
                                  int64_t Nk, int C, int D, int64_t k, int mode, float *out_code, float *out_tau,
                                  int32_t *out_nvalid, void *workspace, size_t workspace_bytes, void *stream) {
    if (Nq == 0 && Nk >= 0 && C >= 1 && k >= 1) return DDG_OK;       // an empty query shard (its pointers may be NULL)
    if (!out_code || k < 1) return DDG_E_ARG;
    int rc = ddg_attn_prepare(Xq0, mq, q_row_offset, Xk0, mk, W, offsets_host, Nq, Nk, C, D, k, mode, workspace,
                              workspace_bytes, stream);
    if (rc) return rc;
    const int same = (Xq0 == Xk0 && mq == mk && Nq == Nk) ? 1 : 0;
    rc = ddg_attn_select(Xq0, Nq, Nk, C, D, k, mode, same, out_code, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return ddg_attn_fill(offsets_host, Nq, Nk, C, D, k, mode, same, out_code, out_tau, out_nvalid, workspace,
                         workspace_bytes, stream);
}
