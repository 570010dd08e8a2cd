// Microbenchmark (diagnostic): the memory floor of the fused no-blank launch -- one 1024-thread
// workgroup per sample reads its T rows of x[T][B][C] into registers, then (after a barrier: no
// gradient row can be produced before every row has been read) writes T rows of the same shape.
// Variants: 8- or 16-byte accesses, non-temporal / plain / write-through stores, C = 158 (632-byte
// rows: 128-byte lines shared with the neighbouring samples) or C = 160 (whole lines), read-only,
// write-only.  20 launches per hipGraph replay; prints us per launch.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/rw_phase.hip -o tools/micro/rw_phase
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4), aligned(8)));

__device__ __forceinline__ int xcd_sample(int bid, int B)
{
    const int x = bid & 7, k = bid >> 3, q = B >> 3, r = B & 7;
    return x * q + (x < r ? x : r) + k;
}

// MODE bit 0: read, bit 1: write; ST: 0 = nt, 1 = plain, 2 = write-through (sc0 sc1)
template <int VEC, int MODE, int ST, bool XCD>
__global__ __launch_bounds__(1024) void phase_kernel(const float *x, float *g, int T, int B, int C)
{
    constexpr int R = 10;                                    // rows per wave (T <= 160)
    const int b = XCD ? xcd_sample(blockIdx.x, B) : blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (C + VEC - 1) / VEC;                     // lanes per row
    const int rows_per_instr = 64 / per > 0 ? 64 / per : 1;  // VEC = 4: one row per instruction (40 lanes)
    float v[R][2][VEC];
    int col[2];
    for (int h = 0; h < 2; ++h) {
        int c = (lane + 64 * h) * VEC;
        if (c >= C) c = -1;
        else if (c + VEC > C) c = C - VEC;
        col[h] = c;
    }
    (void)rows_per_instr;
    if (MODE & 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = w + 16 * r;
            if (t >= T) break;
            const float *row = x + ((size_t)t * B + b) * C;
#pragma unroll
            for (int h = 0; h < (VEC == 2 ? 2 : 1); ++h)
                if (col[h] >= 0) {
                    if (VEC == 2) { f2 q = *reinterpret_cast<const f2 *>(row + col[h]); v[r][h][0] = q.x; v[r][h][1] = q.y; }
                    else { f4 q = *reinterpret_cast<const f4 *>(row + col[h]); v[r][h][0] = q.x; v[r][h][1] = q.y; v[r][h][2] = q.z; v[r][h][3] = q.w; }
                }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            for (int h = 0; h < 2; ++h)
                for (int k = 0; k < VEC; ++k) v[r][h][k] = (float)(lane + r);
    }
    __syncthreads();
    if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = w + 16 * r;
            if (t >= T) break;
            float *row = g + ((size_t)t * B + b) * C;
#pragma unroll
            for (int h = 0; h < (VEC == 2 ? 2 : 1); ++h)
                if (col[h] >= 0) {
                    if (VEC == 2) {
                        f2 q = {v[r][h][0], v[r][h][1]};
                        f2 *p = reinterpret_cast<f2 *>(row + col[h]);
                        if (ST == 0) __builtin_nontemporal_store(q, p);
                        else if (ST == 1) *p = q;
                        else if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(q) : "memory");
                        else if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
                        else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(q) : "memory");
                    } else {
                        f4 q = {v[r][h][0], v[r][h][1], v[r][h][2], v[r][h][3]};
                        f4 *p = reinterpret_cast<f4 *>(row + col[h]);
                        if (ST == 0) __builtin_nontemporal_store(q, p);
                        else if (ST == 1) *p = q;
                        else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(q) : "memory");
                        else if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(q) : "memory");
                    }
                }
        }
    } else {
        float s = 0.f;
        for (int r = 0; r < R; ++r)
            for (int h = 0; h < 2; ++h)
                for (int k = 0; k < VEC; ++k) s += v[r][h][k];
        if (s == 12345.678f) g[0] = s;
    }
}

// read-only, 4-byte accesses (lane = column, three loads per row: the row layout of the binary kernels)
template <bool XCD>
__global__ __launch_bounds__(1024) void read_dword_kernel(const float *x, float *g, int T, int B, int C)
{
    constexpr int R = 10;
    const int b = XCD ? xcd_sample(blockIdx.x, B) : blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = w + 16 * r;
        if (t >= T) break;
        const float *row = x + ((size_t)t * B + b) * C;
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int c = lane + 64 * j; v[r][j] = row[c < C ? c : C - 1]; }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) { if (w + 16 * r >= T) break; for (int j = 0; j < 3; ++j) s += v[r][j]; }
    if (s == 12345.678f) g[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VEC, int MODE, int ST, bool XCD>
static int run(const char *name, const float *x, float *g, int T, int B, int C, hipStream_t s)
{
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((phase_kernel<VEC, MODE, ST, XCD>), dim3(B), dim3(1024), 0, s, x, g, T, B, C);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double us = best * 1e3 / 200;
    const double mb = ((MODE & 1) + ((MODE & 2) >> 1)) * (double)T * B * C * 4 / 1e6;
    printf("%-44s C=%3d B=%4d: %6.2f us per launch  (%5.1f MB -> %5.2f TB/s)\n", name, C, B, us, mb, mb / us / 1e6 * 1e6 / 1e6 * 1e0);
    return 0;
}

int main()
{
    const int T = 150;
    float *x, *g;
    CK(hipMalloc(&x, (size_t)T * 2048 * 160 * 4));
    CK(hipMalloc(&g, (size_t)T * 2048 * 160 * 4));
    CK(hipMemset(x, 0, (size_t)T * 2048 * 160 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int B : {256, 2048})
        for (int C : {158, 160}) {
            if (run<2, 3, 0, true>("read then write  8 B nt", x, g, T, B, C, s)) return 1;
            if (run<2, 3, 2, true>("read then write  8 B write-through sc0 sc1", x, g, T, B, C, s)) return 1;
            if (run<4, 3, 0, true>("read then write 16 B nt", x, g, T, B, C, s)) return 1;
            if (run<4, 3, 1, true>("read then write 16 B plain", x, g, T, B, C, s)) return 1;
            if (run<4, 3, 2, true>("read then write 16 B write-through sc0 sc1", x, g, T, B, C, s)) return 1;
            if (run<4, 3, 3, true>("read then write 16 B sc1", x, g, T, B, C, s)) return 1;
            if (run<4, 3, 4, true>("read then write 16 B sc0", x, g, T, B, C, s)) return 1;
            if (run<2, 1, 0, true>("read  8 B", x, g, T, B, C, s)) return 1;
            if (run<4, 1, 0, true>("read 16 B", x, g, T, B, C, s)) return 1;
            {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((read_dword_kernel<true>), dim3(B), dim3(1024), 0, s, x, g, T, B, C);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("%-44s C=%3d B=%4d: %6.2f us per launch (eager launches back to back)\n", "read  4 B (lane = column, 3 loads per row)", C, B, best * 1e3 / 200);
            }
            if (run<4, 2, 0, true>("write 16 B nt", x, g, T, B, C, s)) return 1;
            if (run<4, 2, 2, true>("write 16 B write-through", x, g, T, B, C, s)) return 1;
            if (run<4, 2, 3, true>("write 16 B sc1", x, g, T, B, C, s)) return 1;
        }
    return 0;
}
