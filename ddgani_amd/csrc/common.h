// Shared device/host helpers for libddgani_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ddgani.h"

#include <stdio.h>
#define DDG_CHECK_LAUNCH()                                                                   \
    do {                                                                                     \
        hipError_t e__ = hipGetLastError();                                                  \
        if (e__ != hipSuccess) {                                                             \
            fprintf(stderr, "ddgani: %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return DDG_E_LAUNCH;                                                             \
        }                                                                                    \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline __host__ __device__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zeroing by a kernel of our own instead of hipMemsetAsync, in every entry point that callers capture into a HIP graph
// (the training steps, the denoiser forward): a memset NODE of a replayed graph was observed to fill parts of the
// BatchNorm sums with a non-zero garbage pattern about once in five runs of the hospital pipeline,
// a kernel node of ours is ordered like every other kernel of the step.  bytes: a multiple of 16, p 16-byte aligned.
static __global__ void k_zero16(uint4 *__restrict__ p, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static inline int zero_async(void *p, size_t bytes, hipStream_t s) {
    if (((uintptr_t)p & 15) || (bytes & 15)) return DDG_E_ARG;
    if (!bytes) return DDG_OK;
    const size_t n16 = bytes / 16;
    k_zero16<<<(unsigned)((n16 + 255) / 256), 256, 0, s>>>((uint4 *)p, n16);
    return hipGetLastError() == hipSuccess ? DDG_OK : DDG_E_LAUNCH;
}

static __global__ void k_copy_f32(const float *__restrict__ src, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
static inline int copy_f32_async(float *dst, const float *src, size_t n, hipStream_t s) {      // (a kernel, not a memcpy node)
    if (!n) return DDG_OK;
    k_copy_f32<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, dst, n);
    return hipGetLastError() == hipSuccess ? DDG_OK : DDG_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011): one call -> 4 x uint32.
// counter = (idx_lo, idx_hi, stream_lo, stream_hi), key = seed.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (one 64-bit product each: v_mad_u64_u32 instead of a v_mul_hi_u32 + v_mul_lo_u32 pair, all quarter rate)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Stream offset in device memory (ddg_set_stream_offset): every on-chip random draw uses stream_id + *g_stream_off when the
// pointer is set.  A caller that replays a captured HIP graph -- whose kernel arguments, stream numbers included, are
// frozen -- advances that one device word between replays and gets fresh noise.  One copy per translation unit.
static __device__ const uint64_t *g_stream_off = nullptr;
__device__ __forceinline__ uint64_t stream_with_offset(uint64_t stream_id) {
    const uint64_t *p = g_stream_off;
    return p ? stream_id + *p : stream_id;
}
static inline int set_stream_off_tu(const uint64_t *dev_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stream_off), &dev_ptr, sizeof(dev_ptr)) == hipSuccess ? DDG_OK : DDG_E_LAUNCH;
}

// uniform in (0,1]: never 0, so log() is finite
__device__ __forceinline__ float u32_to_unit(uint32_t u) { return ((u >> 8) + 1) * (1.0f / 16777216.0f); }

// 4 standard normals for element group `idx` of random stream `stream_id`
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t stream_id, uint64_t idx) {
    uint32_t r[4];
    stream_id = stream_with_offset(stream_id);
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    float u0 = u32_to_unit(r[0]), u1 = u32_to_unit(r[1]), u2 = u32_to_unit(r[2]), u3 = u32_to_unit(r[3]);
    float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    f32x4 o;
    o.x = ra * c0; o.y = ra * s0; o.z = rb * c1; o.w = rb * s1;
    return o;
}

// 4 uniforms in (0,1] for element group idx
__device__ __forceinline__ f32x4 philox_uniform4(uint64_t seed, uint64_t stream_id, uint64_t idx) {
    uint32_t r[4];
    stream_id = stream_with_offset(stream_id);
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    f32x4 o;
    o.x = u32_to_unit(r[0]); o.y = u32_to_unit(r[1]); o.z = u32_to_unit(r[2]); o.w = u32_to_unit(r[3]);
    return o;
}
