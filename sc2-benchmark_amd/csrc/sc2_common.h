// Shared helpers for the gfx950 kernels of libsc2amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#include "../../include/sc2_bottleneck.h"

void sc2_set_error(const char *fmt, ...);
const sc2_policy &sc2_pol();   // the process-wide dispatch policy (abi.cpp; include/sc2_bottleneck.h): the library reads no environment variable

#define SC2_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            sc2_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define SC2_CHECK_LAUNCH()                                              \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) {                                        \
            sc2_set_error("hip launch failed: %s", hipGetErrorString(e__)); \
            return SC2_ERR_LAUNCH;                                      \
        }                                                               \
    } while (0)

// Non-temporal output stores (round 4), per kernel by measurement: they pay where a launch writes a map far larger than the
// caches while re-reading operands of its own (conv0_gdn96: 616 MB out, - 8 %; conv2x2_gdn512: 822 MB out, - 2 %) and they LOSE
// where the next launch finds part of the map in L2 / the memory-side cache (the head's layers: + 2.5 % over the head, layer4's
// 13 - 51 MB maps + 10 - 15 % each; conv2x2_win + 2 %).  Each kernel has its switch (-DSC2_NT_<KERNEL>=0/1).
typedef __attribute__((ext_vector_type(4))) unsigned sc2_u32x4_t;
__device__ __forceinline__ void sc2_store16_nt(uint4 *dst, const uint4 v) {
    __builtin_nontemporal_store(sc2_u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<sc2_u32x4_t *>(dst));
}
constexpr int SC2_BUF_AUX_NT = 2;   // `aux` of __builtin_amdgcn_raw_buffer_store_*: the nt bit

// Per-DEVICE once-flags of the launchers (hipFuncSetAttribute for > 64 KB of dynamic LDS is a per-device property of a function:
// a process that launches on a second device must set it there too).  Usage: static bool f[SC2_MAX_DEVICES] = {}; bool &done = f[sc2_device_slot()];
constexpr int SC2_MAX_DEVICES = 16;
inline int sc2_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d < SC2_MAX_DEVICES ? d : SC2_MAX_DEVICES - 1;
}

// CU count of the CURRENT device, cached per device (a launcher may be called from one host thread per GPU: ADVICE r5).
inline int sc2_device_cus() {
    static std::atomic<int> cus[SC2_MAX_DEVICES];
    std::atomic<int> &c = cus[sc2_device_slot()];
    int n = c.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        c.store(n, std::memory_order_relaxed);
    }
    return n;
}

// Zero-filled device words for a persistent launcher's work counters: ONE allocation per (ring object, device), made once
// under the ring's mutex and visible to the other host threads through the acquire / release pair.
// A launch that is being CAPTURED into a HIP graph (round 6: the bs-1 evaluation forward replays graphs) keeps its counter
// address for the life of the graph, while eager launches rotate through the ring: a replay on one stream and an eager launch
// on another could meet on one counter.  Captured launches therefore take their counters from a second region behind the ring,
// handed out in order (sc2_counter_ring::launch_slot): 2^18 words (1 MB) per device and kernel family, i.e. 2 048 captured launches
// of the hungriest family (128 words: conv1x1_kres) = some 250 captures of the whole evaluation forward before the region wraps
// around and the OLDEST captures' words are handed out again (graphs that old are dropped ones in any workload of this package: a
// model re-captures only when its parameters change).
constexpr size_t SC2_CAPTURE_WORDS = (size_t)1 << 18;
inline bool sc2_stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

struct sc2_counter_ring {
    std::mutex mu;
    std::atomic<unsigned *> base[SC2_MAX_DEVICES];
    std::atomic<unsigned> captured[SC2_MAX_DEVICES];
    // the counters of ONE launch on stream s: `per_launch` consecutive zeroed words.  Eager: slot seq mod (ring_words / per_launch)
    // of the ring; captured: the next unused words of the capture region.  nullptr: allocation failed /
    // first use of this kernel family inside a capture (hipMalloc is not capturable: run the launch once eagerly first).
    unsigned *launch_slot(hipStream_t s, size_t ring_words, unsigned per_launch, std::atomic<unsigned> &seq) {
        const bool cap = sc2_stream_capturing(s);
        const int d = sc2_device_slot();
        if (cap && !base[d].load(std::memory_order_acquire)) {
            sc2_set_error("a persistent kernel's first launch on this device cannot be captured into a graph: warm it up eagerly");
            return nullptr;
        }
        unsigned *ring = get(ring_words);
        if (!ring) return nullptr;
        if (!cap) return ring + (size_t)per_launch * (seq.fetch_add(1) % (unsigned)(ring_words / per_launch));
        // (per_launch divides the region: 1, 8 or 128 words; the modulo wraps to the oldest words)
        const unsigned at = (captured[d].fetch_add(per_launch)) % (unsigned)SC2_CAPTURE_WORDS;
        return ring + ring_words + at;
    }
    unsigned *get(size_t ring_words) {   // nullptr: the allocation failed (sc2_set_error holds the reason)
        const size_t words = ring_words + SC2_CAPTURE_WORDS;
        const int d = sc2_device_slot();
        unsigned *p = base[d].load(std::memory_order_acquire);
        if (p) return p;
        std::lock_guard<std::mutex> lock(mu);
        p = base[d].load(std::memory_order_relaxed);
        if (p) return p;
        void *ptr = nullptr;
        if (hipMalloc(&ptr, words * sizeof(unsigned)) != hipSuccess || hipMemset(ptr, 0, words * sizeof(unsigned)) != hipSuccess ||
            hipStreamSynchronize(nullptr) != hipSuccess) {   // (the fill runs on the null stream; launches go to non-blocking streams)
            sc2_set_error("cannot allocate / clear %zu work-counter words on device %d", words, d);
            return nullptr;
        }
        p = static_cast<unsigned *>(ptr);
        base[d].store(p, std::memory_order_release);
        return p;
    }
};

// "resident workgroups per CU" of one kernel instantiation, cached per device
struct sc2_per_device_int {
    std::atomic<int> v[SC2_MAX_DEVICES];
    std::atomic<int> &here() { return v[sc2_device_slot()]; }
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}
