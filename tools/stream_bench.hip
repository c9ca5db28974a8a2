// Microbenchmark: how fast can a 4-in / 1-out fp32 streaming kernel run on this box, by access shape?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int UNROLL, bool NT>
__global__ void k_stream5(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, const f32x4 *__restrict__ c,
                          const f32x4 *__restrict__ d, f32x4 *__restrict__ o, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride * UNROLL) {
        f32x4 va[UNROLL], vb[UNROLL], vc[UNROLL], vd[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = g + u * stride;
            if (i < n4) {
                if (NT) { va[u] = __builtin_nontemporal_load(a + i); vb[u] = __builtin_nontemporal_load(b + i); vc[u] = __builtin_nontemporal_load(c + i); vd[u] = __builtin_nontemporal_load(d + i); }
                else { va[u] = a[i]; vb[u] = b[i]; vc[u] = c[i]; vd[u] = d[i]; }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = g + u * stride;
            if (i < n4) {
                f32x4 r = va[u] * vb[u] + vc[u] * vd[u];
                if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
            }
        }
    }
}
// block-contiguous: each block owns a contiguous span
template <int UNROLL>
__global__ void k_stream5_span(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, const f32x4 *__restrict__ c,
                               const f32x4 *__restrict__ d, f32x4 *__restrict__ o, int64_t n4) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    for (int64_t g = lo + threadIdx.x; g < hi; g += (int64_t)blockDim.x * UNROLL) {
        f32x4 va[UNROLL], vb[UNROLL], vc[UNROLL], vd[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const int64_t i = g + u * blockDim.x; if (i < hi) { va[u] = a[i]; vb[u] = b[i]; vc[u] = c[i]; vd[u] = d[i]; } }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const int64_t i = g + u * blockDim.x; if (i < hi) o[i] = va[u] * vb[u] + vc[u] * vd[u]; }
    }
}
__global__ void k_copy(const f32x4 *__restrict__ a, f32x4 *__restrict__ o, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) o[g] = a[g];
}
__global__ void k_read(const f32x4 *__restrict__ a, float *__restrict__ o, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f32x4 s = {0, 0, 0, 0};
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) s += a[g];
    if (s.x + s.y + s.z + s.w == 123.456f) o[0] = 1;
}

int main() {
    const int64_t n = (int64_t)1000000 * 176, n4 = n / 4;
    float *buf[5];
    for (auto &p : buf) { CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *nm, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-44s %.3f ms  %.0f GB/s\n", nm, ms, bytes / (ms * 1e-3) / 1e9);
    };
    const f32x4 *A = (const f32x4 *)buf[0], *B = (const f32x4 *)buf[1], *C = (const f32x4 *)buf[2], *D = (const f32x4 *)buf[3];
    f32x4 *O = (f32x4 *)buf[4];
    run("copy grid 4096x256", 8.0 * n, [&] { k_copy<<<4096, 256>>>(A, O, n4); });
    run("copy grid 2048x256", 8.0 * n, [&] { k_copy<<<2048, 256>>>(A, O, n4); });
    run("copy full grid", 8.0 * n, [&] { k_copy<<<(unsigned)((n4 + 255) / 256), 256>>>(A, O, n4); });
    run("read grid 4096x256", 4.0 * n, [&] { k_read<<<4096, 256>>>(A, buf[4], n4); });
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, 96, "stream5 u1 grid %d x256", grid); run(nm, 20.0 * n, [&] { k_stream5<1, false><<<grid, 256>>>(A, B, C, D, O, n4); });
        snprintf(nm, 96, "stream5 u2 grid %d x256", grid); run(nm, 20.0 * n, [&] { k_stream5<2, false><<<grid, 256>>>(A, B, C, D, O, n4); });
        snprintf(nm, 96, "stream5 u4 grid %d x256", grid); run(nm, 20.0 * n, [&] { k_stream5<4, false><<<grid, 256>>>(A, B, C, D, O, n4); });
    }
    run("stream5 u1 NT grid 4096x256", 20.0 * n, [&] { k_stream5<1, true><<<4096, 256>>>(A, B, C, D, O, n4); });
    run("stream5 u2 NT grid 4096x256", 20.0 * n, [&] { k_stream5<2, true><<<4096, 256>>>(A, B, C, D, O, n4); });
    run("stream5 u1 full grid", 20.0 * n, [&] { k_stream5<1, false><<<(unsigned)((n4 + 255) / 256), 256>>>(A, B, C, D, O, n4); });
    run("stream5 u1 grid 2048x512", 20.0 * n, [&] { k_stream5<1, false><<<2048, 512>>>(A, B, C, D, O, n4); });
    run("stream5 u1 grid 1024x1024", 20.0 * n, [&] { k_stream5<1, false><<<1024, 1024>>>(A, B, C, D, O, n4); });
    for (int grid : {2048, 4096, 16384}) {
        char nm[96];
        snprintf(nm, 96, "span u1 grid %d x256", grid); run(nm, 20.0 * n, [&] { k_stream5_span<1><<<grid, 256>>>(A, B, C, D, O, n4); });
        snprintf(nm, 96, "span u2 grid %d x256", grid); run(nm, 20.0 * n, [&] { k_stream5_span<2><<<grid, 256>>>(A, B, C, D, O, n4); });
    }
    return 0;
}
