// Table-side passes of the drop-in path that the reference runs in pandas / numpy on the host (SURVEY.md 8(f) rows
// 2 and 3), as HBM-bound integer / fp64 kernels.  gfx950 only.
//
// Reference functions restated (relative to the reference tree):
//   get_attr_map_Null_kendall     utils/get_attr_attn.py:35-80   (the C^2 crosstabs :61-77 -> ddg_contingency)
//   reconvert_data                utils/util.py:228-254          (decode of the missing cells  -> ddg_score_table)
//   errorLoss / normalization     utils/util.py:17-48, :83-139   (per-column error sums        -> ddg_score_table)
#include <math.h>
#include "common.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------
// (f2) contingency tables of every column pair i < j over the rows observed in both.
// codes_t: uint16 [C][N] column-major discretised values (20-bin numerics, ordinal categoricals), DDG_CODE_MISSING
// where the cell is missing.  One workgroup = one pair x one chunk of rows; the pair's table (card_i x card_j
// counters) is built in LDS when it fits and flushed with one global atomic per non-zero cell, otherwise counted
// straight into global memory.
// ---------------------------------------------------------------------------------------------
#define CT_ROWS_PER_WG 65536
#define CT_LDS_CELLS 8192

__global__ __launch_bounds__(256) void k_contingency(const uint16_t *__restrict__ codes_t, int64_t N, int C,
                                                      const int32_t *__restrict__ card,
                                                      const int32_t *__restrict__ pair_i,
                                                      const int32_t *__restrict__ pair_j,
                                                      const int64_t *__restrict__ pair_off, int nchunks,
                                                      unsigned *__restrict__ tables) {
    __shared__ unsigned tab[CT_LDS_CELLS];
    const int pair = blockIdx.x / nchunks, chunk = blockIdx.x - pair * nchunks;
    const int i = pair_i[pair], j = pair_j[pair];
    const int ci = card[i], cj = card[j];
    const int cells = ci * cj;
    unsigned *gt = tables + pair_off[pair];
    const bool in_lds = cells <= CT_LDS_CELLS;
    if (in_lds) {
        for (int e = threadIdx.x; e < cells; e += blockDim.x) tab[e] = 0;
        __syncthreads();
    }
    const uint16_t *a = codes_t + (int64_t)i * N, *b = codes_t + (int64_t)j * N;
    const int64_t r0 = (int64_t)chunk * CT_ROWS_PER_WG;
    const int64_t r1 = r0 + CT_ROWS_PER_WG < N ? r0 + CT_ROWS_PER_WG : N;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const unsigned va = a[r], vb = b[r];
        if (va == DDG_CODE_MISSING || vb == DDG_CODE_MISSING) continue;
        if (va >= (unsigned)ci || vb >= (unsigned)cj) continue;          // codes outside the declared cardinality
        if (in_lds) atomicAdd(&tab[va * cj + vb], 1u);
        else atomicAdd(&gt[va * cj + vb], 1u);
    }
    if (in_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < cells; e += blockDim.x)
            if (tab[e]) atomicAdd(&gt[e], tab[e]);
    }
}

extern "C" int ddg_contingency(const uint16_t *codes_t, int64_t N, int C, const int32_t *card_dev,
                               const int32_t *pair_i_dev, const int32_t *pair_j_dev, const int64_t *pair_off_dev,
                               int64_t n_pairs, int64_t total_cells, uint32_t *tables, void *stream) {
    if (!codes_t || !card_dev || !pair_i_dev || !pair_j_dev || !pair_off_dev || !tables || N < 0 || C < 1 ||
        n_pairs < 0 || total_cells < 0)
        return DDG_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(tables, 0, (size_t)total_cells * 4, s) != hipSuccess) return DDG_E_LAUNCH;
    if (n_pairs == 0 || N == 0) return DDG_OK;
    const int64_t nchunks = ceil_div64(N, CT_ROWS_PER_WG);
    if (n_pairs * nchunks >= ((int64_t)1 << 31)) return DDG_E_UNSUPPORTED;
    k_contingency<<<(unsigned)(n_pairs * nchunks), 256, 0, s>>>(codes_t, N, C, card_dev, pair_i_dev, pair_j_dev,
                                                                pair_off_dev, (int)nchunks, tables);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// (f3) decode + score of an encoded table in one pass over its missing cells.
// For every cell (r, c) with m[r,c] == 0:
//   numeric      v = code[r, off_c] / K_c + Min_c                          (NumericalField.reverse, utils/field.py:75-82),
//                in the arithmetic numpy uses there: fp64 when K / Min are numpy float64 scalars, fp32 (both
//                operations) when they are Python floats, which is what an object-typed column yields (dec_f32)
//   categorical  v = value of the arg-max position of the block (first maximum, np.argmax)
//   vn = (v - min_c) / den_c     (errorLoss's normalisation by the ground truth column: utils/util.py:17-48)
//   d  = ori_n[r,c] - vn
// and per column: sum d^2, sum |d| (numeric) or the number of cells with ori_n != vn (categorical).
// sums: double [C][3], ADDED to (zero it first).  ARMSE / AMAE follow on the host from these and the missing counts.
// ---------------------------------------------------------------------------------------------
struct ScoreArgs {
    const float *code;
    const double *ori_n;
    const uint8_t *m;
    const int32_t *offsets;
    const uint8_t *is_cat;
    const double *dec_K, *dec_min, *pos_value, *nmin, *nden;
    const uint8_t *dec_f32;
    double *sums;
    int64_t N;
    int C, D;
};

__global__ __launch_bounds__(256) void k_score_table(ScoreArgs p) {
    extern __shared__ double acc[];          // [C][3]
    for (int e = threadIdx.x; e < 3 * p.C; e += blockDim.x) acc[e] = 0.0;
    __syncthreads();
    const int64_t total = p.N * p.C;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        if (p.m[w]) continue;
        const int64_t r = w / p.C;
        const int c = (int)(w - r * p.C);
        const int a = p.offsets[c], b = p.offsets[c + 1];
        const float *x = p.code + r * p.D;
        double v;
        if (p.is_cat[c]) {
            int best = a;
            float bv = x[a];
            for (int jx = a + 1; jx < b; ++jx)
                if (x[jx] > bv) { bv = x[jx]; best = jx; }
            v = p.pos_value[best];
        } else {
            // numpy's result type: K / Min learnt from an object-typed column are Python floats ("weak" scalars): the
            // fp32 code stays fp32 through both operations; numpy float64 scalars promote it to fp64
            if (p.dec_f32[c]) v = (double)__fadd_rn(__fdiv_rn(x[a], (float)p.dec_K[c]), (float)p.dec_min[c]);
            else v = (double)x[a] / p.dec_K[c] + p.dec_min[c];
        }
        const double vn = (v - p.nmin[c]) / p.nden[c];
        const double on = p.ori_n[w];
        if (p.is_cat[c]) {
            if (on != vn) atomicAdd(&acc[3 * c + 2], 1.0);
        } else {
            const double d = on - vn;
            atomicAdd(&acc[3 * c], d * d);
            atomicAdd(&acc[3 * c + 1], fabs(d));
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * p.C; e += blockDim.x)
        if (acc[e] != 0.0) atomicAdd(&p.sums[e], acc[e]);
}

extern "C" int ddg_score_table(const float *code, const double *ori_norm, const uint8_t *m, const int32_t *offsets_dev,
                               const uint8_t *is_cat_dev, const double *dec_K, const double *dec_min,
                               const uint8_t *dec_f32, const double *pos_value, const double *norm_min,
                               const double *norm_den, double *sums, int64_t N, int C, int D, void *stream) {
    if (!code || !ori_norm || !m || !offsets_dev || !is_cat_dev || !dec_K || !dec_min || !dec_f32 || !pos_value || !norm_min ||
        !norm_den || !sums || N < 0 || C < 1 || D < C)
        return DDG_E_ARG;
    if ((size_t)C * 3 * 8 > 60 * 1024) return DDG_E_UNSUPPORTED;
    if (N == 0) return DDG_OK;
    ScoreArgs p{code, ori_norm, m, offsets_dev, is_cat_dev, dec_K, dec_min, pos_value, norm_min, norm_den, dec_f32, sums, N, C, D};
    int64_t g = ceil_div64(N * C, 256 * 8);
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    k_score_table<<<(unsigned)g, 256, (size_t)C * 3 * 8, (hipStream_t)stream>>>(p);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// (f3) encode of a table on the device: Data_convert                          utils/util.py:52-78
//      NumericalField.convert  utils/field.py:65-67   K * (x - Min)   ("minmax")   /   (x - mu) / sigma   ("mean_std")
//      CategoricalField.convert utils/field.py:138-150 one-hot over the sorted distinct values
// vals [N][C] fp64: the numeric value, or for a categorical column the position of the cell's value among the column's
// sorted distinct values (what CategoricalField.dict maps it to).  The arithmetic is the reference's: fp64 (pandas /
// numpy on float64 columns), rounded ONCE to fp32 where the reference builds its tensor (torch.tensor(..., dtype=float),
// BaseLine/DiffGANI.py:31): with the same K / Min / mu / sigma the result is bit-identical.  One thread per encoded
// element (coalesced stores); HBM-bound: 4 N D bytes out, 8 N C in.  A categorical position outside its block leaves the
// block all zero (the host mirror raises KeyError for such a value before it gets here).
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_encode_table(const double *__restrict__ vals, const int32_t *__restrict__ offsets,
                                                      const uint8_t *__restrict__ is_cat, const uint8_t *__restrict__ mode,
                                                      const double *__restrict__ pa, const double *__restrict__ pb,
                                                      float *__restrict__ code, int64_t N, int C, int D) {
    const int64_t total = N * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D;
        const int d = (int)(e - r * D);
        int lo = 0, hi = C;                      // column of encoded position d: offsets[lo] <= d < offsets[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offsets[mid] <= d) lo = mid; else hi = mid;
        }
        const double v = vals[r * C + lo];
        float out;
        if (is_cat[lo]) {
            out = ((double)(d - offsets[lo]) == v) ? 1.f : 0.f;
        } else if (mode[lo] == 0) {
            out = (float)(pa[lo] * (v - pb[lo]));
        } else {
            out = (float)((v - pb[lo]) / pa[lo]);
        }
        code[e] = out;
    }
}
#pragma clang fp contract(on)

extern "C" int ddg_encode_table(const double *vals, const int32_t *offsets_dev, const uint8_t *is_cat_dev,
                                const uint8_t *mode_dev, const double *scale_dev, const double *shift_dev, float *code,
                                int64_t N, int C, int D, void *stream) {
    if (!vals || !offsets_dev || !is_cat_dev || !mode_dev || !scale_dev || !shift_dev || !code || N < 0 || C < 1 || D < C)
        return DDG_E_ARG;
    if (N == 0) return DDG_OK;
    int64_t g = ceil_div64(N * D, 256 * 4);
    if (g > 256 * 16) g = 256 * 16;
    k_encode_table<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(vals, offsets_dev, is_cat_dev, mode_dev, scale_dev, shift_dev,
                                                                 code, N, C, D);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// (f3) decode of a whole encoded table: reconvert_data                        utils/util.py:228-254
// out[r,c] = obs[r,c] * m[r,c] + cur[r,c] * (1 - m[r,c])   (fp64, the reference's expression, :248), with cur the decoded
// cell exactly as in k_score_table (numeric x / K + Min in numpy's arithmetic for the field, categorical: value of the
// first arg-max position) and obs [N][C] fp64 the table the observed cells come from (categorical columns as their
// ordinal codes).  The host maps the categorical codes back to labels (enc.inverse_transform).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode_table(const float *__restrict__ code, const uint8_t *__restrict__ m,
                                                      const double *__restrict__ obs, const int32_t *__restrict__ offsets,
                                                      const uint8_t *__restrict__ is_cat, const double *__restrict__ dec_K,
                                                      const double *__restrict__ dec_min, const uint8_t *__restrict__ dec_f32,
                                                      const double *__restrict__ pos_value, double *__restrict__ out, int64_t N,
                                                      int C, int D) {
    const int64_t total = N * C;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = w / C;
        const int c = (int)(w - r * C);
        const int a = offsets[c], b = offsets[c + 1];
        const float *x = code + r * D;
        double v;
        if (is_cat[c]) {
            int best = a;
            float bv = x[a];
            for (int jx = a + 1; jx < b; ++jx)
                if (x[jx] > bv) { bv = x[jx]; best = jx; }
            v = pos_value[best];
        } else {
            if (dec_f32[c]) v = (double)__fadd_rn(__fdiv_rn(x[a], (float)dec_K[c]), (float)dec_min[c]);
            else v = (double)x[a] / dec_K[c] + dec_min[c];
        }
        const double mm = m[w] ? 1.0 : 0.0;
        out[w] = obs[w] * mm + v * (1.0 - mm);       // (this file is built with -ffp-contract=off: two products, one sum)
    }
}

extern "C" int ddg_decode_table(const float *code, const uint8_t *m, const double *obs, const int32_t *offsets_dev,
                                const uint8_t *is_cat_dev, const double *dec_K, const double *dec_min, const uint8_t *dec_f32,
                                const double *pos_value, double *out, int64_t N, int C, int D, void *stream) {
    if (!code || !m || !obs || !offsets_dev || !is_cat_dev || !dec_K || !dec_min || !dec_f32 || !pos_value || !out || N < 0 ||
        C < 1 || D < C)
        return DDG_E_ARG;
    if (N == 0) return DDG_OK;
    int64_t g = ceil_div64(N * C, 256 * 4);
    if (g > 256 * 16) g = 256 * 16;
    k_decode_table<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(code, m, obs, offsets_dev, is_cat_dev, dec_K, dec_min, dec_f32,
                                                                 pos_value, out, N, C, D);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}
