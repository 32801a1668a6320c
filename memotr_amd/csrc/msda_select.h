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
//     shrunk by `shrink` pixels (what a smaller margin would have lost) into CUMULATIVE 64-bit counters of a small
//     device record, one row of counters per selector level the launch ran at; one wavefront of every launch sums the
//     rows and stores the totals + a sequence number into a 128-byte record in mapped host memory -- no copy, no
//     event, nothing on the stream.  Nothing is ever cleared and no argument changes from launch to launch (round 5:
//     round 4's two parity buffers were exchanged by a kernel argument, which a captured launch replays for ever), so
//     launches inside a replayed hipGraph count exactly like eager ones;
//   * the host reads that record at the next call of the same call site -- whatever has arrived; a few calls of
//     delay are harmless, the offsets drift over thousands of steps -- takes the difference to what it saw last per
//     level, and moves between LEVELS:
//       backward  0: bins, small margin   1: bins, large margin   2: no windows (sort + gather through the caller's
//                 scratch, msda_bwd_sorted.h -- round 6; whole-row float atomics, msda_bwd_rows.h, for callers without)
//       forward   0: windows              1: head-major gather
//     with hysteresis; a level without windows produces no statistics, so every `kSelProbeEvery`-th call probes one
//     level down;
//   * a record belongs to (device, direction, call site, M, L, P, element size): the off-window share is a property of
//     the module's learnt offsets, not of the image size (round 4 keyed on the geometry too: multi-scale training
//     filled the table within ~21 clips and never saw a geometry twice).  All records live in ONE device block and one
//     mapped host block allocated at the first eager call; when the table is full the least recently used record is
//     given to the new key;
//   * callers that replay captured launches (the model's hipGraph caches) call msda_selector_poll() before a replay:
//     it reads every record, moves the levels and returns a signature of the levels in force; a cache keyed on it
//     replays the graph captured at those levels and captures another when a level has moved (a capturing call takes
//     the level the last poll announced).
// `msda_set_option("auto_select", 0)` switches the mechanism off (level 0 always).
#pragma once

#include <mutex>

constexpr int kSelSlots = 256;
constexpr unsigned kSelProbeEvery = 32;
constexpr int kSelLevels = 3;
constexpr int kSelShards = 32;
// device record (64-bit words): [level][shard]{valid, off, inner, pad}, then the publishers' sequence counter
constexpr int kSelCntWords = kSelLevels * kSelShards * 4;
constexpr int kSelDevWords = 1024;                      // 8 KiB per record
// ... and, for the windowed forward, where its windows go (device-only, never read by the host): per (head, level) the
// running sums {dx, dy, n} of the sampling offsets relative to the query's own pixel (floats, added by the counting
// workgroups, consumed by the publishing wavefront), the mean offsets the NEXT launches centre their windows on, and
// one bit per (head, level) "this mean has been measured"
constexpr int kSelHintHeads = 16, kSelHintLevels = 4;
constexpr int kSelHintAccWord = 392;                    // 16 x 4 x {dx, dy, n, -} floats
constexpr int kSelHintWord = kSelHintAccWord + kSelHintHeads * kSelHintLevels * 2;      // 16 x 4 x {dx, dy} floats
constexpr int kSelHintValidWord = kSelHintWord + kSelHintHeads * kSelHintLevels;
constexpr int kSelHostWords = 16;                       // 128 B per record: [level]{valid, off, inner}, seq at [9]
constexpr unsigned long long kSelMinSample = 2048;      // valid corners a level's difference must hold to be judged
static_assert(kSelCntWords + 1 <= kSelHintAccWord && kSelHintValidWord < kSelDevWords, "record too small");

struct SelKey {
    int dev, kind;                  // kind 0: forward, 1: backward
    unsigned long long site;        // caller's tag (msda_set_call_site): one record per module
    int M, L, P, dt;
    bool operator==(const SelKey &o) const {
        return dev == o.dev && kind == o.kind && site == o.site && M == o.M && L == o.L && P == o.P && dt == o.dt;
    }
};

struct SelSlot {
    SelKey key;
    bool used, primed;              // primed: `last` holds a baseline of this key's counters
    unsigned long long *dev;        // device record
    volatile unsigned long long *host;   // mapped host record
    unsigned long long *host_dev;   // the device's pointer to `host`
    unsigned long long seen;        // sequence number of the last record read
    unsigned long long last[kSelLevels][3];
    unsigned long long stamp;       // last use (least-recently-used replacement)
    int level;                      // what the data ask for
    int eff;                        // what a capturing call runs at (= level, or one below while a probe is due)
    bool polled;                    // msda_selector_poll() has announced `eff`: eager calls leave it alone from then on
    bool scratch;                   // the site's caller sizes scratch for its backward calls (msda_backward_workspace_bytes): level 2 = sorted
    unsigned long long pub_seen;    // publishers' sequence number at the last poll (launches arriving = record in use)
    unsigned calls;
    float frac, frac_inner;         // last measured shares (of the valid corners)
};

// thresholds in 1/1000 of the valid corners (options sel_*): measured crossovers, profiles/r04_selector_probe.txt
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

#ifdef __HIPCC__
// How the counts travel (what NOT to do was measured in round 4: a ticket counter that lets the launch's last
// workgroup publish costs one RETURNING atomic per workgroup on one address -- 8736 of them serialise at ~12 ns each
// and the waiting wavefronts doubled the kernel's time; an agent-scope fence per workgroup is a full L2 write-back):
//   * a counting workgroup adds its counts, fire-and-forget, to one of kSelShards shards of the row of the level its
//     launch runs at (zeros are not sent);
//   * one wavefront of every launch sums all rows and stores the totals + a sequence number into the mapped host
//     record.  Workgroups of the same launch may already have added theirs: the host judges differences of at least
//     kSelMinSample corners, a fraction of a launch more or less does not move a share.
__device__ __forceinline__ void sel_add(unsigned long long *dev, int level, unsigned shard, unsigned valid, unsigned off,
                                        unsigned inner) {
    unsigned long long *c = dev + ((unsigned)level * kSelShards + (shard & (kSelShards - 1))) * 4u;
    if (valid) __hip_atomic_fetch_add(c + 0, (unsigned long long)valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (off) __hip_atomic_fetch_add(c + 1, (unsigned long long)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (inner) __hip_atomic_fetch_add(c + 2, (unsigned long long)inner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One full wavefront of the launch (lane = threadIdx.x & 63): lane j < 9 sums counter j % 3 of level j / 3 over the
// shards and stores it; lane 0 then bumps the sequence number.
__device__ __forceinline__ void sel_publish(unsigned long long *dev, unsigned long long *host, int lane) {
    if (lane < kSelLevels * 3) {
        const unsigned long long *c = dev + (unsigned)(lane / 3) * kSelShards * 4u + (unsigned)(lane % 3);
        unsigned long long s = 0ull;
#pragma unroll 8
        for (int i = 0; i < kSelShards; ++i) s += __hip_atomic_load(c + i * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0) {
        const unsigned long long seq =
            __hip_atomic_fetch_add(dev + kSelCntWords, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        __hip_atomic_store(host + 9, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Window placement of the windowed forward, device side only.  A counting workgroup adds the offsets it measured on its
// first rows; the publishing wavefront (lane = (head, level)) turns a batch of at least 32 samples into the mean the next
// launches use and takes the batch out (an exchange: what later workgroups of the same launch add stays for the next
// publisher).  No argument changes from launch to launch, so replayed captures keep adapting.
__device__ __forceinline__ void sel_hint_add(unsigned long long *dev, int head, int level, float sx, float sy, float n) {
    float *a = reinterpret_cast<float *>(dev + kSelHintAccWord) + (head * kSelHintLevels + level) * 4;
    unsafeAtomicAdd(a + 0, sx);
    unsafeAtomicAdd(a + 1, sy);
    unsafeAtomicAdd(a + 2, n);
}

__device__ __forceinline__ void sel_hint_publish(unsigned long long *dev, int lane, int M, int L) {
    const int head = lane / kSelHintLevels, level = lane % kSelHintLevels;
    bool any = false;
    if (head < M && head < kSelHintHeads && level < L) {
        float *a = reinterpret_cast<float *>(dev + kSelHintAccWord) + (head * kSelHintLevels + level) * 4;
        const float n0 = __hip_atomic_load(a + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n0 >= 32.f) {
            const float n = __hip_atomic_exchange(a + 2, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float sx = __hip_atomic_exchange(a + 0, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float sy = __hip_atomic_exchange(a + 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n > 0.f) {
                float *h = reinterpret_cast<float *>(dev + kSelHintWord) + (head * kSelHintLevels + level) * 2;
                __hip_atomic_store(h + 0, sx / n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(h + 1, sy / n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                any = true;
            }
        }
    }
    // one "measured" bit per (head, level): lane = head * 4 + level is the bit's number
    const unsigned long long bits = __builtin_amdgcn_ballot_w64(any);
    if (bits != 0ull && lane == 0)
        __hip_atomic_fetch_or(dev + kSelHintValidWord, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
