// msda_select.h -- per-call kernel selection from the measured share of sampling points that leave their windows.
//
// The windowed kernels (forward: msda_fwd_d32_win, backward: msda_bwd_d32_bins) are fast while the sampling points of a
// region stay near it; a point whose bilinear footprint leaves its window takes a slow path (a global gather /
// per-element float atomics).  How many do depends on the sampling offsets the model has LEARNT, which no launch
// parameter tells: at the initialisation's offsets 0-1 % of the corners leave a 4-pixel margin, with uniformly
// random locations all of them do and the windowed backward is 3x slower than the generic kernel.  The reference
// kernel's cost does not depend on the locations at all (ms_deform_im2col_cuda.cuh:237-403), so the choice has to
// follow the data:
//
//   * the windowed kernels count, per launch, the valid corners / those outside the window / those outside a window
//     shrunk by `shrink` pixels (what a smaller margin would have lost) in a small device record; the next launch on
//     the record stores the totals, the level they were measured at and a sequence number into a 32-byte record in
//     mapped host memory -- no copy, no event, nothing on the stream;
//   * the host reads that record at the next call of the same (call site, geometry) -- whatever has arrived; a few
//     calls of delay are harmless, the offsets drift over thousands of steps -- and moves between LEVELS:
//       backward  0: bins, small margin   1: bins, large margin   2: no windows (generic kernel)
//       forward   0: windows              1: head-major gather
//     with hysteresis; a level without windows produces no statistics, so every `kSelProbeEvery`-th call probes one
//     level down.
// The record is allocated on first use (one hipMalloc + one hipHostMalloc, never while the stream is capturing) and
// lives for the life of the process; `msda_set_option("auto_select", 0)` switches the mechanism off (level 0 always).
#pragma once

#include <mutex>

constexpr int kSelSlots = 256;
constexpr unsigned kSelProbeEvery = 32;

struct SelKey {
    int dev, kind;                  // kind 0: forward, 1: backward
    unsigned long long site;        // caller's tag (msda_set_call_site): one record per module, not per geometry only
    int N, S, M, L, P, Lq, dt;
    bool operator==(const SelKey &o) const {
        return dev == o.dev && kind == o.kind && site == o.site && N == o.N && S == o.S && M == o.M && L == o.L &&
               P == o.P && Lq == o.Lq && dt == o.dt;
    }
};

struct SelSlot {
    SelKey key;
    bool used;
    unsigned *dev;                  // device: kSelDevWords words (below)
    unsigned launches;              // parity of the next launch
    volatile unsigned *host;        // mapped host: valid, off, inner, seq, level the launch ran at
    unsigned *host_dev;             // the device's pointer to `host`
    unsigned seen;
    int level;
    unsigned calls;
    float frac, frac_inner;         // last measured shares (of the valid corners)
};

// thresholds in 1/1000 of the valid corners (options sel_*): measured crossovers, profiles/r04_bwd_selector.txt
struct SelRule {
    int up0, up1, down1, down2;
};

// The transition function, host-only and pure (tests/test_selector_cpu.py drives it through msda_selector_next).
//   backward: level 0 -> 1 when more than up0 of the corners leave the small window; 1 -> 2 when more than up1 leave
//   the large one; 1 -> 0 when fewer than down1 would leave the small one (`inner`); 2 -> 1 when a probe at level 1
//   sees fewer than down2 leave.  forward: two levels, rules up0 / down1 on the same share.
// (off / inner: shares in 1/1000)
inline int sel_next_level(int kind, int level, float f, float fi, const SelRule &r) {
    if (kind == 0) {
        if (level == 0) return f > (float)r.up0 ? 1 : 0;
        return f < (float)r.down1 ? 0 : 1;
    }
    if (level == 0) return f > (float)r.up0 ? 1 : 0;
    if (level == 1) return f > (float)r.up1 ? 2 : (fi < (float)r.down1 ? 0 : 1);
    return f < (float)r.down2 ? 1 : 2;
}

// Device record: two buffers (launch parity) of kSelShards x {valid, off, inner, pad} counters, then {level of the
// launch that filled buffer 0, of buffer 1, sequence number}.
constexpr int kSelShards = 32;
constexpr int kSelDevWords = 2 * kSelShards * 4 + 4;

#ifdef __HIPCC__
// How the counts travel (what NOT to do was measured first: a ticket counter that lets the launch's last workgroup
// publish costs one RETURNING atomic per workgroup on one address -- 8736 of them serialise at ~12 ns each and the
// waiting wavefronts doubled the kernel's time; an agent-scope fence per workgroup is a full L2 write-back, worse):
//   * every workgroup adds its counts, fire-and-forget, to one of kSelShards shards of the buffer of its launch's
//     parity (zeros are not sent);
//   * workgroup 0 of the NEXT launch on the record (stream order: the previous launch has finished) sums the other
//     parity's shards, clears them and stores totals + level + a sequence number into the mapped host record.
// The host therefore sees a launch's statistics two calls later -- the offsets drift over thousands of steps.
__device__ __forceinline__ void sel_add(unsigned *dev, int parity, unsigned shard, unsigned valid, unsigned off,
                                        unsigned inner) {
    unsigned *c = dev + ((unsigned)parity * kSelShards + (shard & (kSelShards - 1))) * 4u;
    if (valid) __hip_atomic_fetch_add(c + 0, valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (off) __hip_atomic_fetch_add(c + 1, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (inner) __hip_atomic_fetch_add(c + 2, inner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One full wavefront of the launch's workgroup 0 (lane = threadIdx.x & 63): publish what the previous launch counted.
__device__ __forceinline__ void sel_publish_previous(unsigned *dev, unsigned *host, int parity, unsigned level, int lane) {
    unsigned *c = dev + ((unsigned)(parity ^ 1) * kSelShards + (unsigned)(lane & (kSelShards - 1))) * 4u;
    unsigned v = 0u, o = 0u, i = 0u;
    if (lane < kSelShards) {
        v = __hip_atomic_exchange(c + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o = __hip_atomic_exchange(c + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i = __hip_atomic_exchange(c + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        v += __shfl_xor(v, s, 64);
        o += __shfl_xor(o, s, 64);
        i += __shfl_xor(i, s, 64);
    }
    if (lane == 0) {
        unsigned *tail = dev + 2 * kSelShards * 4;
        const unsigned prev_level = tail[parity ^ 1];
        tail[parity] = level;
        const unsigned seq = ++tail[2];
        __hip_atomic_store(host + 0, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host + 1, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host + 2, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host + 4, prev_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif
