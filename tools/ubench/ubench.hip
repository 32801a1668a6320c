// Micro-benchmarks that size the MSDeformAttn kernel designs on MI355X (standalone; hipcc ubench.hip -o ubench).
//   lds_add      ds_add_f32 throughput (bank-spread / same-8-banks patterns)
//   lds_read     ds_read_b128 throughput (row gather pattern: 8 lanes x 16 B per 128-B row)
//   gatomic      global_atomic_add_f32 throughput by address pattern
//   l1_rows      buffer_load_dwordx4 of 128-B rows (8 rows per wave instruction) by working-set size
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// mode 0: rotated channel order (32 banks x2), 1: unrotated (8 banks x8), 2: random dword, 3: u32 atomics rotated
__global__ __launch_bounds__(256) void k_lds_add(float* out, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 3, sub = lane & 7;
    for (int i = tid; i < 4096; i += 256) lds[i] = 0.f;
    __syncthreads();
    unsigned idx[8];
    unsigned x = (tid >> 3) * 2654435761u + blockIdx.x * 977u;   // same for the 8 lanes of a row group
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x = x * 1664525u + 1013904223u;
        const unsigned row = (x >> 8) & 127;                      // 128 rows of 32 floats = 16 KB
        if (mode == 0 || mode >= 3) idx[j] = row * 32 + sub * 4 + ((j + grp) & 3);
        else if (mode == 1) idx[j] = row * 32 + sub * 4 + (j & 3);
        else idx[j] = ((x >> 4) * (lane + 1)) & 4095;
    }
    unsigned* ldsu = reinterpret_cast<unsigned*>(lds);
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (mode == 3) atomicAdd(&ldsu[idx[j]], 1u);
            else if (mode == 4) atomicAdd(reinterpret_cast<unsigned long long*>(&ldsu[idx[j] & ~1u]), 1ull);
            else __hip_atomic_fetch_add(&lds[idx[j]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds[5];
}

// ds_read_b128 row gather (8 lanes x 16 B per 128-B row), random rows per 8-lane group
__global__ __launch_bounds__(256) void k_lds_read(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7;
    for (int i = tid; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    const f4* lds4 = reinterpret_cast<const f4*>(lds);
    unsigned idx[8];
    unsigned x = (tid >> 3) * 2654435761u + blockIdx.x * 977u;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; idx[j] = ((x >> 8) & 127) * 8 + sub; }
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += lds4[idx[j]];
            idx[j] = (idx[j] + 8 * 37) & 1023;      // new row next time (keeps the loads in the loop)
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1;
}

// Pair reads: the 16 lanes that own a (query, head) row read the two horizontally adjacent pixels of a bilinear
// sample = 256 contiguous bytes at a random 128-byte-aligned LDS address.
//   MODE 0: a row = 16 consecutive lanes
//   MODE 1: a row = one of the four 16-lane groups the LDS services a ds_read_b128 in
//           ({0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}) -> conflict free
template <int MODE>
__global__ __launch_bounds__(256) void k_lds_pair(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192 + 32; i += 256) lds[i] = (float)i;      // 257 pixels of 128 B
    __syncthreads();
    const f4* lds4 = reinterpret_cast<const f4*>(lds);
    int row, chunk;                    // chunk 0..15 of the 256-byte pair
    if (MODE == 0) { row = lane >> 4; chunk = lane & 15; }
    else {
        const int j = lane & 15, odd = (lane >> 4) & 1, jq = j >> 2;
        const bool inner = jq == 1 || jq == 2;
        row = ((lane >> 5) << 1) + ((inner == (odd != 0)) ? 0 : 1);
        const int half = j >> 3;
        chunk = half * 8 + (half ? 3 - (j & 3) : (j & 3)) + 4 * odd;
    }
    unsigned idx[8];
    unsigned x = (unsigned)((tid >> 6) * 4 + row) * 2654435761u + blockIdx.x * 977u;   // same for the lanes of a row
#pragma unroll
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; idx[j] = ((x >> 8) & 255u) * 8 + chunk; }
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += lds4[idx[j]];
            idx[j] = ((idx[j] + 8 * 37) & 2047) ;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_gatomic(float* buf, unsigned n_rows, int iters, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 3, sub = lane & 7;
    unsigned x = (blockIdx.x * 256 + tid) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        unsigned idx;
        if (mode == 0) {            // 64 lanes contiguous: 2 full rows
            unsigned r = (x >> 8) % n_rows; r = __shfl(r, 0); idx = ((r + (lane >> 5)) % n_rows) * 32 + (lane & 31);
        } else if (mode == 1) {     // two independent rows of 32 lanes
            unsigned r = (x >> 8) % n_rows; r = __shfl(r, lane & 32); idx = r * 32 + (lane & 31);
        } else if (mode == 2) {     // 8 rows x 8 lanes, stride 16 B (the d32_gather pattern)
            unsigned r = (x >> 8) % n_rows; r = __shfl(r, lane & ~7); idx = r * 32 + sub * 4 + (it & 3);
        } else if (mode == 3) {     // 8 rows x 8 lanes contiguous 32 B
            unsigned r = (x >> 8) % n_rows; r = __shfl(r, lane & ~7); idx = r * 32 + sub + 8 * (it & 3);
        } else {                    // fully random dwords
            idx = (x >> 4) % (n_rows * 32);
        }
        unsafeAtomicAdd(&buf[idx], 1.0f);
        (void)grp;
    }
}

__global__ __launch_bounds__(256) void k_l1_rows(const float* buf, float* out, unsigned n_rows, int iters, unsigned bytes, int local) {
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
    unsigned x = (blockIdx.x * 256 + tid) * 2654435761u;
    f4 acc = {0, 0, 0, 0};
    const unsigned base_row = local ? (blockIdx.x * 97u) % n_rows : 0;
    const unsigned span = local ? 512u : n_rows;   // local: each block re-reads a 64 KB neighbourhood
    for (int it = 0; it < iters; it += 4) {
        unsigned rows[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x = x * 1664525u + 1013904223u; unsigned rr = (base_row + (x >> 8) % span) % n_rows; rows[j] = __shfl(rr, lane & ~7); }
        f4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, rows[j] * 128 + sub * 16, 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1;
}

// "Ideal" forward of the encoder call (178,584 rows x 16 points x 4 corner rows of 128 B + 16 FMAs per lane and point):
// the corner rows come from a 32 KB LDS pool (mode 0: no global traffic at all -- the LDS-side ceiling of any tiled
// design) or from a 32 KB global slab that stays in the vector L1 (mode 1: the L1 request-rate ceiling with perfect
// locality).  Row order is random per (row, point); no records, no index arithmetic, no output: only the gather + FMA.
template <int MODE>
__global__ __launch_bounds__(256) void k_ideal_fwd(const float* buf, float* out, long n_tasks, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)i;           // 256 rows of 128 B
    __syncthreads();
    const f4* lds4 = reinterpret_cast<const f4*>(lds);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
    const long wave = ((long)blockIdx.x * 256 + tid) >> 6, n_waves = ((long)gridDim.x * 256) >> 6;
    f4 total = {0, 0, 0, 0};
    for (long task = wave; task < n_tasks; task += n_waves) {
        unsigned x = (unsigned)(task * 8 + (lane >> 3)) * 2654435761u;     // same for the 8 lanes of a row
        f4 acc = {0, 0, 0, 0};
        for (int t = 0; t < 16; t += 4) {
            f4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                x = x * 1664525u + 1013904223u;
                const unsigned row = (x >> 8) & 255;
                v[j] = MODE == 0 ? lds4[row * 8 + sub]
                                 : __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, row * 128 + sub * 16, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += 0.25f * v[j];
        }
        total += acc;
    }
    if (total.x + total.y + total.z + total.w == 12345.f) out[0] = 1;
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    float* out; CK(hipMalloc(&out, 1 << 20));
    const int blocks = 256 * 8;
    printf("== LDS ds_add_f32 (256 thr/WG, %d WGs, 16 KB LDS) ==\n", blocks);
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 2048;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_add, dim3(blocks), dim3(256), 16384, 0, out, iters, mode); });
        double ops = (double)blocks * 256 * iters;
        printf("lds_add mode %d: %.3f ms  %.1f G lane-atomics/s  (%.2f per clk per CU @2.4GHz)\n", mode, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    }
    {
        const int iters = 4096;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_read, dim3(blocks), dim3(256), 16384, 0, out, iters); });
        double bytes = (double)blocks * 256 * iters * 16;
        printf("lds_read_b128 rows: %.3f ms  %.1f TB/s  (%.1f B/clk/CU)\n", ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4);
    }
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 4096;
        float ms = mode == 0 ? time_ms([&] { hipLaunchKernelGGL(k_lds_pair<0>, dim3(blocks), dim3(256), 33024, 0, out, iters); })
                             : time_ms([&] { hipLaunchKernelGGL(k_lds_pair<1>, dim3(blocks), dim3(256), 33024, 0, out, iters); });
        double bytes = (double)blocks * 256 * iters * 16;
        printf("lds_read_b128 pixel pairs, %s: %.3f ms  %.1f TB/s  (%.1f B/clk/CU)\n",
               mode == 0 ? "16 consecutive lanes per row" : "LDS service groups as rows  ", ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4);
    }
    printf("== global atomic add f32 ==\n");
    const unsigned n_rows = 22323 * 8;  // 22.9 MB like value
    float* buf; CK(hipMalloc(&buf, (size_t)n_rows * 128)); CK(hipMemset(buf, 0, (size_t)n_rows * 128));
    const char* names[] = {"64 contiguous", "2x32 contiguous", "8 rows x 8 lanes stride16B", "8 rows x 8 lanes contiguous", "random dwords"};
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 256;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_gatomic, dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters, mode); }, 3);
        double ops = (double)blocks * 256 * iters;
        printf("gatomic %-28s: %.3f ms  %.1f G lane-atomics/s\n", names[mode], ms, ops / ms / 1e6);
    }
    for (unsigned small_rows : {64u, 4096u}) {
        const int iters = 256;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_gatomic, dim3(blocks), dim3(256), 0, 0, buf, small_rows, iters, 1); }, 3);
        double ops = (double)blocks * 256 * iters;
        printf("gatomic 2x32 contiguous over %u rows (contention): %.3f ms  %.1f G lane-atomics/s\n", small_rows, ms, ops / ms / 1e6);
    }
    printf("== 128-B row gathers (buffer_load_dwordx4, 8 rows per wave instruction) ==\n");
    for (int local = 0; local < 2; ++local)
        for (unsigned rows : {256u, 8192u, 22323u * 8u}) {
            const int iters = 1024;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_l1_rows, dim3(blocks), dim3(256), 0, 0, buf, out, rows, iters, (unsigned)((size_t)n_rows * 128), local); }, 3);
            double bytes = (double)blocks * 256 * iters * 16;
            printf("row gather over %8u rows (%7.1f KB) local=%d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU)\n", rows, rows * 128 / 1024.0, local, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4);
        }
    printf("== ideal encoder-call forward: 22323 tasks x 8 rows x 16 points x 4 corner rows + FMAs ==\n");
    for (int mode = 0; mode < 2; ++mode) {
        const long n_tasks = 22323;
        float ms = mode == 0 ? time_ms([&] { hipLaunchKernelGGL(k_ideal_fwd<0>, dim3(256 * 8), dim3(256), 32768, 0, buf, out, n_tasks, (unsigned)((size_t)n_rows * 128)); }, 20)
                             : time_ms([&] { hipLaunchKernelGGL(k_ideal_fwd<1>, dim3(256 * 8), dim3(256), 32768, 0, buf, out, n_tasks, (unsigned)((size_t)n_rows * 128)); }, 20);
        double bytes = (double)n_tasks * 8 * 16 * 4 * 128;
        printf("ideal forward, corner rows from %s: %.1f us  (%.2f TB/s of row reads, %.1f B/clk/CU; 80.0 MB algorithmic -> %.1f %% of 8 TB/s)\n",
               mode == 0 ? "LDS      " : "L1 (32 KB)", ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4, 80.0057 / ms / 8.0 * 100 / 1e3 * 1e3 / 1e3);
    }
    return 0;
}
