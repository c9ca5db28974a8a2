// HBM-bound passes of the sampler: noise schedule, forward noising, fused reverse step,
// per-field head, mask-merge, BatchNorm finalize.  gfx950 only.
//
// Reference functions restated here (relative to the reference tree):
//   Diffusion_setting.initialize   model/Diffusion.py:93-108
//   forward_diffusion / get_xt     model/Diffusion.py:391-403
//   G_D_onestep (after G call)     model/Diffusion.py:415-424
//   Generator_x0 head              model/Diffusion.py:188-201
//   mask-merge + re-softmax        model/Diffusion.py:598, :606-615
#include <math.h>
#include <type_traits>
#include "common.h"

extern "C" int ddg_version(void) { return 100; }

extern "C" const char *ddg_error_string(int code) {
    switch (code) {
        case DDG_OK: return "ok";
        case DDG_E_ARG: return "bad argument";
        case DDG_E_UNSUPPORTED: return "unsupported shape";
        case DDG_E_WORKSPACE: return "workspace too small";
        case DDG_E_LAUNCH: return "kernel launch failed";
        case DDG_E_COLLECTIVE: return "the caller's all-reduce callback failed";
        case DDG_E_SCRATCH_LIMIT: return "input exceeds the static scratch of this entry point (use the workspace variant)";
        default: return "unknown error";
    }
}

// ---------------------------------------------------------------------------------------------
// a3 schedule (host).  Same fp32 operation order as torch: linspace -> beta[0]=0 -> 1-beta ->
// sqrt -> cumprod -> sqrt.  torch.linspace(fp32) computes start + step*i for the first half and
// end - step*(steps-1-i) for the second half, step = (end-start)/(steps-1) in fp32.
// ---------------------------------------------------------------------------------------------
extern "C" int ddg_schedule(int T, float *o) {
    if (T < 1 || !o) return DDG_E_ARG;
    float *betas = o, *alphas = o + T, *sa = o + 2 * T, *obsa = o + 3 * T, *sb = o + 4 * T, *ac = o + 5 * T,
          *sac = o + 6 * T, *s1mac = o + 7 * T;
    const float start = 1e-4f, end = 2e-2f;
    if (T == 1) {
        betas[0] = start;
    } else {
        const float step = (end - start) / (float)(T - 1);
        const int half = T / 2;
        for (int i = 0; i < T; ++i) {
            volatile float v = (i < half) ? start + step * (float)i : end - step * (float)(T - 1 - i);
            betas[i] = v;
        }
    }
    betas[0] = 0.f;
    double cp = 1.0;   // torch.cumprod accumulates fp32 inputs in double on the CPU
    for (int i = 0; i < T; ++i) {
        alphas[i] = 1.f - betas[i];
        sa[i] = sqrtf(alphas[i]);
        obsa[i] = 1.f / sa[i];
        sb[i] = sqrtf(betas[i]);
        cp *= (double)alphas[i];
        ac[i] = (float)cp;
        sac[i] = sqrtf(ac[i]);
        s1mac[i] = sqrtf(1.f - ac[i]);
    }
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// a5 forward noising
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

__global__ void k_forward_noise_uniform(const float *__restrict__ x0, const float *__restrict__ eps,
                                        float ca, float cb, uint64_t seed, uint64_t stream_id,
                                        float *__restrict__ xt, float *__restrict__ eps_out, int64_t n) {
    const int64_t ngroups = (n + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = g * 4;
        if (e0 + 3 < n) {
            f32x4 xv = *reinterpret_cast<const f32x4 *>(x0 + e0);
            f32x4 ev = eps ? *reinterpret_cast<const f32x4 *>(eps + e0) : philox_normal4(seed, stream_id, (uint64_t)g);
            f32x4 r;
            r.x = ca * xv.x + cb * ev.x; r.y = ca * xv.y + cb * ev.y;
            r.z = ca * xv.z + cb * ev.z; r.w = ca * xv.w + cb * ev.w;
            *reinterpret_cast<f32x4 *>(xt + e0) = r;
            if (eps_out) *reinterpret_cast<f32x4 *>(eps_out + e0) = ev;
        } else {
            f32x4 ev4 = eps ? f32x4{0, 0, 0, 0} : philox_normal4(seed, stream_id, (uint64_t)g);
            for (int c = 0; c < 4 && e0 + c < n; ++c) {
                float ev = eps ? eps[e0 + c] : ev4[c];
                xt[e0 + c] = ca * x0[e0 + c] + cb * ev;
                if (eps_out) eps_out[e0 + c] = ev;
            }
        }
    }
}

__global__ void k_forward_noise_rows(const float *__restrict__ x0, const float *__restrict__ eps,
                                     const int64_t *__restrict__ t, const float *__restrict__ sched, int T,
                                     uint64_t seed, uint64_t stream_id, float *__restrict__ xt,
                                     float *__restrict__ eps_out, int64_t rows, int D) {
    const int64_t n = rows * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D;
        const int ti = min(max((int)t[r], 1), T) - 1;      // a timestep outside [1, T] is clamped (torch would raise)
        const float ca = sched[6 * T + ti], cb = sched[7 * T + ti];
        float ev;
        if (eps) ev = eps[e];
        else ev = philox_normal4(seed, stream_id, (uint64_t)(e >> 2))[e & 3];
        xt[e] = __fadd_rn(__fmul_rn(ca, x0[e]), __fmul_rn(cb, ev));
        if (eps_out) eps_out[e] = ev;
    }
}

#define EW_GRID_MAX ((int64_t)1 << 30)
static inline int ew_grid(int64_t work_items, int block) {
    // one work item per thread, no persistent loop: measured on this chip (scratch stream_bench, 4 streams in / 1 out,
    // 704 MB each) a full grid streams at 5.6 TB/s, 4 096 grid-striding workgroups at 4.3 (a plain copy: 6.2 vs 4.7)
    int64_t g = ceil_div64(work_items, block);
    if (g > EW_GRID_MAX) g = EW_GRID_MAX;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int ddg_forward_noise(const float *x0, const float *eps, const int64_t *t, int t_uniform,
                                 const float *sched, int T, uint64_t seed, uint64_t stream_id, float *x_t,
                                 float *eps_out, int64_t rows, int D, void *stream) {
    if (rows < 0 || D < 1 || T < 1) return DDG_E_ARG;
    if (rows == 0) return DDG_OK;                       // an empty table: nothing to launch, data pointers may be NULL
    if (!x0 || !x_t) return DDG_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = rows * D;
    if (t) {
        if (!sched) return DDG_E_ARG;
        k_forward_noise_rows<<<ew_grid(n, 256), 256, 0, s>>>(x0, eps, t, sched, T, seed, stream_id, x_t, eps_out, rows, D);
    } else {
        if (t_uniform < 1 || t_uniform > T) return DDG_E_ARG;
        float h[8 * 64];
        float *hs = h;
        float *dyn = nullptr;
        if (T > 64) { dyn = (float *)malloc(sizeof(float) * 8 * (size_t)T); hs = dyn; }
        ddg_schedule(T, hs);
        const float ca = hs[6 * T + t_uniform - 1], cb = hs[7 * T + t_uniform - 1];
        if (dyn) free(dyn);
        k_forward_noise_uniform<<<ew_grid((n + 3) / 4, 256), 256, 0, s>>>(x0, eps, ca, cb, seed, stream_id, x_t, eps_out, n);
    }
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// a7 fused reverse step: posterior mean + noise + observed-cell re-noising + mask-merge.
// One HBM pass: reads x_t, x, M, pred (+z when explicit), writes x_{t-1}.
// ---------------------------------------------------------------------------------------------
struct RevCoef { float c1, c2, c3, den, sacm1, s1mm1; };

__device__ __forceinline__ float rev_one(float xt, float x, float M, float p, float z, const RevCoef &k) {
    const float gm = __fdiv_rn(__fadd_rn(__fmul_rn(k.c1, p), __fmul_rn(k.c2, xt)), k.den);
    const float gs = __fdiv_rn(__fmul_rn(k.c3, z), k.den);
    const float g = __fadd_rn(gm, gs);
    const float o = __fadd_rn(__fmul_rn(k.sacm1, x), __fmul_rn(k.s1mm1, z));
    return __fadd_rn(__fmul_rn(M, o), __fmul_rn(__fsub_rn(1.f, M), g));
}

__global__ void k_reverse_step(const float *__restrict__ xt, const float *__restrict__ x, const float *__restrict__ M,
                               const float *__restrict__ pred, const float *__restrict__ z, RevCoef k, uint64_t seed,
                               uint64_t stream_id, float *__restrict__ out, int64_t n) {
    const int64_t ngroups = (n + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = g * 4;
        if (e0 + 3 < n) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(xt + e0);
            f32x4 b = *reinterpret_cast<const f32x4 *>(x + e0);
            f32x4 m = *reinterpret_cast<const f32x4 *>(M + e0);
            f32x4 p = *reinterpret_cast<const f32x4 *>(pred + e0);
            f32x4 zz = z ? *reinterpret_cast<const f32x4 *>(z + e0) : philox_normal4(seed, stream_id, (uint64_t)g);
            f32x4 r;
            r.x = rev_one(a.x, b.x, m.x, p.x, zz.x, k); r.y = rev_one(a.y, b.y, m.y, p.y, zz.y, k);
            r.z = rev_one(a.z, b.z, m.z, p.z, zz.z, k); r.w = rev_one(a.w, b.w, m.w, p.w, zz.w, k);
            *reinterpret_cast<f32x4 *>(out + e0) = r;
        } else {
            f32x4 z4 = z ? f32x4{0, 0, 0, 0} : philox_normal4(seed, stream_id, (uint64_t)g);
            for (int c = 0; c < 4 && e0 + c < n; ++c)
                out[e0 + c] = rev_one(xt[e0 + c], x[e0 + c], M[e0 + c], pred[e0 + c], z ? z[e0 + c] : z4[c], k);
        }
    }
}

static int rev_coef(const float *sh, int T, int t, RevCoef &k);
static inline int rev_coef_of(const float *sh, int T, int t, RevCoef &k) { return rev_coef(sh, T, t, k); }

extern "C" int ddg_reverse_step(const float *x_t, const float *x, const float *M, const float *pred, const float *z,
                                const float *sh, int T, int t, uint64_t seed, uint64_t stream_id, float *out,
                                int64_t rows, int D, void *stream) {
    if (!sh || rows < 0 || D < 1) return DDG_E_ARG;
    if (rows > 0 && (!x_t || !x || !M || !pred || !out)) return DDG_E_ARG;
    RevCoef k;
    int rc = rev_coef_of(sh, T, t, k);
    if (rc) return rc;
    if (rows == 0) return DDG_OK;
    const int64_t n = rows * D;
    k_reverse_step<<<ew_grid((n + 3) / 4, 256), 256, 0, (hipStream_t)stream>>>(x_t, x, M, pred, z, k, seed, stream_id, out, n);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// a6 head / a9 mask-merge: a block stages ROWS_PB rows in LDS (coalesced), one thread per
// (row, field) applies sigmoid / softmax over the field's block, then the rows go back coalesced.
// ---------------------------------------------------------------------------------------------
// Field lists of a head kernel in LDS, built once per workgroup: the categorical fields (their softmax needs the whole
// block: one thread per (row, field)) and the numeric COLUMNS (one thread per (row, column) applies the sigmoid in the
// tile).  A sigmoid evaluated per element and selected away on categorical elements cost an expf and an IEEE division
// on every element of a table that is mostly one-hot blocks (c3: 128 of 176 columns).
struct HeadLists { uint16_t *numc; int *catf; int *cnt; };
__device__ __forceinline__ HeadLists head_lists_at(float *after_tile, int D, int C) {
    HeadLists h;
    h.numc = reinterpret_cast<uint16_t *>(after_tile);                                     // [D]
    h.catf = reinterpret_cast<int *>(after_tile + ((D + 1) >> 1));                          // [C]
    h.cnt = h.catf + C;                                                                     // [2]: ncat, nnum
    return h;
}
// Built by the first wave alone, positions from ballots and a wave scan: no LDS atomics and no barrier of its own -- the
// caller's barrier after the tile load publishes the lists, and the (L2-resident) reads of offsets / is_cat overlap with
// the tile's loads.  Lists come out in field order.
__device__ __forceinline__ void head_lists_build(const HeadLists &h, const int32_t *offsets, const uint8_t *is_cat, int C) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    int ncat = 0, nnum = 0;
    for (int f0 = 0; f0 < C; f0 += 64) {
        const int f = f0 + lane;
        const bool valid = f < C;
        const bool ic = valid && is_cat[f];
        const int a = valid ? offsets[f] : 0, wdt = (valid && !ic) ? offsets[f + 1] - a : 0;
        const uint64_t m = __ballot(ic);
        if (ic) h.catf[ncat + __popcll(m & ((1ull << lane) - 1))] = f;
        ncat += __popcll(m);
        int incl = wdt;                                   // inclusive scan of the numeric widths over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
        for (int j = 0; j < wdt; ++j) h.numc[nnum + incl - wdt + j] = (uint16_t)(a + j);
        nnum += __shfl(incl, 63);
    }
    if (lane == 0) { h.cnt[0] = ncat; h.cnt[1] = nnum; }
}
// w / n for 0 <= w < 2^16 without an integer division: (w + 0.5) / n is at least 0.5 / n away from an integer, far more
// than the product's rounding error
__device__ __forceinline__ int div_small(int w, float inv_n) { return (int)(((float)w + 0.5f) * inv_n); }
__device__ __forceinline__ float sigmoid_ref(float v) { return 1.f / (1.f + expf(-v)); }
// the two in-tile phases of a head: softmax over every categorical block, sigmoid of every numeric column (the numeric
// phase only when the table has categorical fields: an all-numeric table applies the sigmoid where the element is used)
__device__ __forceinline__ void head_in_tile(float *tile, const HeadLists &h, const int32_t *offsets, int nr, int D,
                                             bool sigmoid_phase) {
    const int ncat = h.cnt[0], nnum = h.cnt[1];
    const float inv_ncat = 1.f / (float)(ncat > 0 ? ncat : 1), inv_nnum = 1.f / (float)(nnum > 0 ? nnum : 1);
    for (int w = threadIdx.x; w < nr * ncat; w += blockDim.x) {
        const int r = div_small(w, inv_ncat), f = h.catf[w - r * ncat];
        const int a = offsets[f], b = offsets[f + 1];
        float *p = tile + r * D;
        float mx = p[a];
        for (int j = a + 1; j < b; ++j) mx = fmaxf(mx, p[j]);
        float s = 0.f;
        for (int j = a; j < b; ++j) { const float e = expf(p[j] - mx); p[j] = e; s += e; }
        for (int j = a; j < b; ++j) p[j] = p[j] / s;
    }
    if (sigmoid_phase)
        for (int w = threadIdx.x; w < nr * nnum; w += blockDim.x) {
            const int r = div_small(w, inv_nnum);
            float *p = tile + r * D + h.numc[w - r * nnum];
            *p = sigmoid_ref(*p);
        }
}

template <bool MERGE>
__global__ void k_head_merge(const float *__restrict__ y, const float *__restrict__ x, const float *__restrict__ M,
                             const int32_t *__restrict__ offsets, const uint8_t *__restrict__ is_cat, int C,
                             float *__restrict__ out, int64_t rows, int D, int rows_pb) {
    // as in k_head_reverse: the (row, field) phase only for the categorical fields; a numeric element's sigmoid is
    // applied where the element is written back
    extern __shared__ float tile[];
    const HeadLists h = head_lists_at(tile + (size_t)rows_pb * D, D, C);
    head_lists_build(h, offsets, is_cat, C);               // published by the first tile's barrier
    const int64_t nblk = ceil_div64(rows, rows_pb);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = blk * rows_pb;
        const int nr = (int)((rows - r0 < rows_pb) ? rows - r0 : rows_pb);
        const int64_t base = r0 * D;
        const int nel = nr * D;
        const int n4 = (base & 3) == 0 ? nel >> 2 : 0;          // 16-byte accesses where the tile starts aligned
        for (int g = threadIdx.x; g < n4; g += blockDim.x) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(y + base + 4 * g);
            if (MERGE) {
                const f32x4 m = *reinterpret_cast<const f32x4 *>(M + base + 4 * g);
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + base + 4 * g);
                v.x = m.x * xv.x + (1.f - m.x) * v.x; v.y = m.y * xv.y + (1.f - m.y) * v.y;
                v.z = m.z * xv.z + (1.f - m.z) * v.z; v.w = m.w * xv.w + (1.f - m.w) * v.w;
            }
            *reinterpret_cast<f32x4 *>(tile + 4 * g) = v;
        }
        for (int i = 4 * n4 + threadIdx.x; i < nel; i += blockDim.x) {
            float v = y[base + i];
            if (MERGE) {
                const float m = M[base + i];
                v = m * x[base + i] + (1.f - m) * v;
            }
            tile[i] = v;
        }
        __syncthreads();
        const bool inline_sigmoid = !MERGE && h.cnt[0] == 0;   // all-numeric table: sigmoid where the element is written back
        head_in_tile(tile, h, offsets, nr, D, !MERGE && !inline_sigmoid);
        __syncthreads();
        for (int g = threadIdx.x; g < n4; g += blockDim.x) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(tile + 4 * g);
            if (inline_sigmoid) { v.x = sigmoid_ref(v.x); v.y = sigmoid_ref(v.y); v.z = sigmoid_ref(v.z); v.w = sigmoid_ref(v.w); }
            *reinterpret_cast<f32x4 *>(out + base + 4 * g) = v;
        }
        for (int i = 4 * n4 + threadIdx.x; i < nel; i += blockDim.x) out[base + i] = inline_sigmoid ? sigmoid_ref(tile[i]) : tile[i];
        __syncthreads();
    }
}

// a6 head + a7 reverse step in ONE pass: the denoiser's pre-head logits y go through the per-field head inside the LDS
// tile and straight into the reverse-step formula; the prediction never travels to HBM and back (saves a write and a
// read of [rows, D] per reverse step).  Same arithmetic as k_head_merge<false> followed by k_reverse_step, so the
// results are bit-identical to the two-kernel sequence.  Noise: explicit z, or Philox indexed by the GLOBAL element
// (group e >> 2, lane e & 3) exactly as k_reverse_step does.
__global__ void k_head_reverse(const float *__restrict__ y, const float *__restrict__ xt, const float *__restrict__ x,
                               const float *__restrict__ M, const float *__restrict__ z, RevCoef k, uint64_t seed,
                               uint64_t stream_id, const int32_t *__restrict__ offsets,
                               const uint8_t *__restrict__ is_cat, int C, float *__restrict__ out, int64_t rows, int D,
                               int rows_pb) {
    // Work split without divergence: the categorical fields go through a (row, field) phase -- their softmax needs the
    // whole block -- and the numeric columns through a (row, column) phase (head_in_tile); one thread per (row, field)
    // over ALL fields made every wave run the 8-element softmax path and the sigmoid path one after the other (0.35 of
    // the HBM rate), a sigmoid per element selected away on categorical ones wasted an expf and a division on most of
    // c3's columns.  Same expressions per element: bit-identical results.
    extern __shared__ float tile[];
    const HeadLists h = head_lists_at(tile + (size_t)rows_pb * D, D, C);
    head_lists_build(h, offsets, is_cat, C);               // published by the first tile's barrier
    // the elementwise phase, compiled twice: with the sigmoid applied where the element is consumed (all-numeric tables)
    // and without (the tile already holds the head's output)
    auto reverse_phase = [&](auto SIG, int64_t base, int nel) {
        constexpr bool sig = decltype(SIG)::value;
        auto pred_of = [&](float v) { return sig ? sigmoid_ref(v) : v; };
        if ((base & 3) == 0) {
            const int ng = (nel + 3) / 4;
            for (int g = threadIdx.x; g < ng; g += blockDim.x) {
                const int i0 = g * 4;
                const int64_t e0 = base + i0;
                if (i0 + 3 < nel) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(xt + e0);
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(x + e0);
                    const f32x4 m = *reinterpret_cast<const f32x4 *>(M + e0);
                    const f32x4 zz = z ? *reinterpret_cast<const f32x4 *>(z + e0) : philox_normal4(seed, stream_id, (uint64_t)(e0 >> 2));
                    const f32x4 pv = *reinterpret_cast<const f32x4 *>(tile + i0);
                    f32x4 r;
                    r.x = rev_one(a.x, b.x, m.x, pred_of(pv.x), zz.x, k); r.y = rev_one(a.y, b.y, m.y, pred_of(pv.y), zz.y, k);
                    r.z = rev_one(a.z, b.z, m.z, pred_of(pv.z), zz.z, k); r.w = rev_one(a.w, b.w, m.w, pred_of(pv.w), zz.w, k);
                    *reinterpret_cast<f32x4 *>(out + e0) = r;
                } else {
                    const f32x4 z4 = z ? f32x4{0, 0, 0, 0} : philox_normal4(seed, stream_id, (uint64_t)(e0 >> 2));
                    for (int c = 0; c < 4 && i0 + c < nel; ++c)
                        out[e0 + c] = rev_one(xt[e0 + c], x[e0 + c], M[e0 + c], pred_of(tile[i0 + c]), z ? z[e0 + c] : z4[c], k);
                }
            }
        } else {
            for (int i = threadIdx.x; i < nel; i += blockDim.x) {
                const int64_t e = base + i;
                const float zz = z ? z[e] : philox_normal4(seed, stream_id, (uint64_t)(e >> 2))[e & 3];
                out[e] = rev_one(xt[e], x[e], M[e], pred_of(tile[i]), zz, k);
            }
        }
    };
    const int64_t nblk = ceil_div64(rows, rows_pb);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = blk * rows_pb;
        const int nr = (int)((rows - r0 < rows_pb) ? rows - r0 : rows_pb);
        const int64_t base = r0 * D;
        const int nel = nr * D;
        if ((base & 3) == 0) {
            const int n4 = nel >> 2;
            for (int i = threadIdx.x; i < n4; i += blockDim.x)
                *reinterpret_cast<f32x4 *>(tile + 4 * i) = *reinterpret_cast<const f32x4 *>(y + base + 4 * i);
            for (int i = 4 * n4 + threadIdx.x; i < nel; i += blockDim.x) tile[i] = y[base + i];
        } else {
            for (int i = threadIdx.x; i < nel; i += blockDim.x) tile[i] = y[base + i];
        }
        __syncthreads();
        const bool inline_sigmoid = h.cnt[0] == 0;
        head_in_tile(tile, h, offsets, nr, D, !inline_sigmoid);
        __syncthreads();
        if (inline_sigmoid) reverse_phase(std::true_type{}, base, nel);
        else reverse_phase(std::false_type{}, base, nel);
        __syncthreads();
    }
}

static int rev_coef(const float *sh, int T, int t, RevCoef &k) {
    if (t < 2 || t > T) return DDG_E_ARG;
    const float *betas = sh, *sa = sh + 2 * T, *ac = sh + 5 * T, *sac = sh + 6 * T, *s1mac = sh + 7 * T;
    // coefficient products in the reference's fp32 association (model/Diffusion.py:415-421)
    volatile float one_m_acm1 = 1.f - ac[t - 2];
    volatile float c1 = sac[t - 2] * betas[t - 1];
    volatile float c2 = sa[t - 1] * one_m_acm1;
    volatile float c3 = one_m_acm1 * betas[t - 1];
    volatile float den = 1.f - ac[t - 1];
    k.c1 = c1; k.c2 = c2; k.c3 = c3; k.den = den; k.sacm1 = sac[t - 2]; k.s1mm1 = s1mac[t - 2];
    return DDG_OK;
}

extern "C" int ddg_head_reverse_step(const float *y, const float *x_t, const float *x, const float *M, const float *z,
                                     const float *sh, int T, int t, uint64_t seed, uint64_t stream_id,
                                     const int32_t *offsets, const uint8_t *is_cat, int C, float *out, int64_t rows,
                                     int D, void *stream) {
    if (!sh || !offsets || !is_cat || rows < 0 || D < 1 || C < 1) return DDG_E_ARG;
    if (rows > 0 && (!y || !x_t || !x || !M || !out)) return DDG_E_ARG;
    RevCoef k;
    int rc = rev_coef(sh, T, t, k);
    if (rc) return rc;
    if (rows == 0) return DDG_OK;
    // rows per workgroup: ~16 KiB tiles, eight workgroups per CU (the phases of a tile are separated by barriers: other
    // workgroups fill the gaps); DDG_HEAD_TILE=<floats> overrides (A/B)
    static const int tile_floats = [] { const char *e = getenv("DDG_HEAD_TILE"); const int v = e ? atoi(e) : 4096; return v < 256 ? 256 : (v > 8192 ? 8192 : v); }();
    int rows_pb = tile_floats / D;
    if (rows_pb < 1) rows_pb = 8192 / D;
    if (rows_pb < 1) return DDG_E_UNSUPPORTED;
    if (rows_pb > 64) rows_pb = 64;
    const size_t lds = (size_t)rows_pb * D * sizeof(float) + (size_t)((D + 1) >> 1) * 4 + (size_t)(C + 2) * sizeof(int);
    const int64_t nblk = ceil_div64(rows, rows_pb);
    const int grid = (int)(nblk < EW_GRID_MAX ? nblk : EW_GRID_MAX);      // one tile per workgroup (see ew_grid)
    k_head_reverse<<<grid, 256, lds, (hipStream_t)stream>>>(y, x_t, x, M, z, k, seed, stream_id, offsets, is_cat, C, out,
                                                            rows, D, rows_pb);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

static int launch_head_merge(bool merge, const float *y, const float *x, const float *M, const int32_t *offsets,
                             const uint8_t *is_cat, int C, float *out, int64_t rows, int D, void *stream) {
    if (!offsets || !is_cat || rows < 0 || D < 1 || C < 1) return DDG_E_ARG;
    if (rows == 0) return DDG_OK;                       // an empty table: data pointers may be NULL
    if (!y || !out || (merge && (!x || !M))) return DDG_E_ARG;
    int rows_pb = 8192 / D;            // <= 32 KiB of LDS
    if (rows_pb < 1) return DDG_E_UNSUPPORTED;
    if (rows_pb > 64) rows_pb = 64;
    // a minibatch (B = 128) would fill three workgroups: fewer rows per workgroup until about 512 of them exist
    if (rows < (int64_t)512 * rows_pb) rows_pb = (int)(rows / 512 > 1 ? rows / 512 : 1);
    const size_t lds = (size_t)rows_pb * D * sizeof(float) + (size_t)((D + 1) >> 1) * 4 + (size_t)(C + 2) * sizeof(int);
    int64_t nblk = ceil_div64(rows, rows_pb);
    int grid = (int)(nblk < EW_GRID_MAX ? nblk : EW_GRID_MAX);
    if (merge)
        k_head_merge<true><<<grid, 256, lds, (hipStream_t)stream>>>(y, x, M, offsets, is_cat, C, out, rows, D, rows_pb);
    else
        k_head_merge<false><<<grid, 256, lds, (hipStream_t)stream>>>(y, x, M, offsets, is_cat, C, out, rows, D, rows_pb);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

extern "C" int ddg_head(const float *y, const int32_t *offsets, const uint8_t *is_cat, int C, float *out,
                        int64_t rows, int D, void *stream) {
    return launch_head_merge(false, y, nullptr, nullptr, offsets, is_cat, C, out, rows, D, stream);
}

extern "C" int ddg_mask_merge(const float *x, const float *M, const float *xhat, const int32_t *offsets,
                              const uint8_t *is_cat, int C, float *code, int64_t rows, int D, void *stream) {
    return launch_head_merge(true, xhat, x, M, offsets, is_cat, C, code, rows, D, stream);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d train-mode statistics -> per-column scale/shift
// ---------------------------------------------------------------------------------------------
__global__ void k_bn_finalize(const double *__restrict__ stats, const float *__restrict__ gamma,
                              const float *__restrict__ beta, double n, float eps, float *__restrict__ scale,
                              float *__restrict__ shift, int width) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= width) return;
    const double mean = stats[k] / n;
    double var = stats[width + k] / n - mean * mean;
    if (var < 0) var = 0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[k] * inv;
    scale[k] = (float)sc;
    shift[k] = (float)((double)beta[k] - mean * sc);
}

extern "C" int ddg_bn_finalize(const double *stats, const float *gamma, const float *beta, double total_rows,
                               float eps, float *scale, float *shift, int width, void *stream) {
    if (!stats || !gamma || !beta || !scale || !shift || width < 1 || total_rows <= 0) return DDG_E_ARG;
    k_bn_finalize<<<(width + 255) / 256, 256, 0, (hipStream_t)stream>>>(stats, gamma, beta, total_rows, eps, scale, shift, width);
    DDG_CHECK_LAUNCH();
    return DDG_OK;
}

// ---------------------------------------------------------------------------------------------
// stream offset in device memory (see common.h): set in every translation unit that draws random numbers
// ---------------------------------------------------------------------------------------------
int ddg_tu_set_stream_off_gemm(const uint64_t *dev_ptr);
int ddg_tu_set_stream_off_train(const uint64_t *dev_ptr);
extern "C" int ddg_set_stream_offset(const uint64_t *dev_ptr) {
    int rc = set_stream_off_tu(dev_ptr);
    if (!rc) rc = ddg_tu_set_stream_off_gemm(dev_ptr);
    if (!rc) rc = ddg_tu_set_stream_off_train(dev_ptr);
    return rc;
}
