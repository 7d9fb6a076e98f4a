// rg_gen.h -- what the level generators share: the RNG (xorshift128 + rand-0.7 sampling at the reference's call-site widths), direction tables,
// small bit helpers, the write-through store of the spare hand-off.  Included by rg_kernels.hip (the wave-per-level generator inside k_build / k_step /
// k_regen) and rg_regen_lanes.hip (the level-per-lane generator of the background spare pipeline).
#pragma once
#include "rg_device.h"

typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// ---------------------------------------------------------------------------------------------
// RNG: xorshift128 + rand-0.7 sample_single (core/src/rng.rs:48-98)
// ---------------------------------------------------------------------------------------------
struct Rng { uint32_t x, y, z, w; };

// Wave-uniform values.  Level generation runs with the whole wave working on ONE env (gen_service): every lane holds the same scalars, so the
// compiler keeps the RNG, the loop counters and the decisions on the scalar unit.  A value that comes back from memory is uniform in fact but
// not provably so; uni() tells the compiler.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// set bits of a wave mask below this lane (v_mbcnt: two instructions, no (1 << lane) - 1 mask held in two registers)
__device__ __forceinline__ int lanes_below(uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }

__device__ __forceinline__ void rng_seed(Rng &r, uint64_t lo, uint64_t hi) {
    r.x = (uint32_t)lo; r.y = (uint32_t)(lo >> 32); r.z = (uint32_t)hi; r.w = (uint32_t)(hi >> 32);
    if ((r.x | r.y | r.z | r.w) == 0) r.x = r.y = r.z = r.w = 0x0BAD5EEDu;
}
__device__ __forceinline__ uint32_t rng_u32(Rng &r) {
    uint32_t t = r.x ^ (r.x << 11);
    r.x = r.y; r.y = r.z; r.z = r.w;
    r.w = r.w ^ (r.w >> 19) ^ (t ^ (t >> 8));
    return r.w;
}
// u32 / i32 call sites: one next_u32 per attempt
__device__ __forceinline__ uint32_t range32(Rng &r, uint32_t low, uint32_t high) {
    uint32_t range = high - low;
    uint32_t zone = (range << __clz((int)range)) - 1u;
    uint32_t v = rng_u32(r);
    while (__builtin_expect(v * range > zone, 0)) v = rng_u32(r);  // rejections are rare: straight-line code on the accepted path
    return low + __umulhi(v, range);
}
// usize / i64 call sites: next_u64 = two next_u32 (low word first), 128-bit product.
// Every 64-bit call site of the engine has range < 2^32 (room counts, cell counts, dice), so the 128-bit product
// v * range is two 32x32->64 multiplies: lo64 = l*range + ((h*range) << 32), hi64 = (h*range >> 32) + carry, and
// "lo64 <= zone" with zone = ((range << clz) << 32) - 1 reduces to a compare of the high words.
__device__ __forceinline__ uint64_t range64(Rng &r, uint64_t low, uint64_t high) {
    uint64_t range = high - low;
    if ((range >> 32) == 0) {
        uint32_t rg = (uint32_t)range;
        uint32_t top = rg << __clz((int)rg);  // zone = (top << 32) - 1
        uint32_t l, h, p0_hi, mid;
        do {
            l = rng_u32(r); h = rng_u32(r);
            p0_hi = __umulhi(l, rg);
            mid = p0_hi + h * rg;                  // bits 32..63 of the low half of the product
        } while (__builtin_expect(!(mid < top), 0));
        return low + (uint64_t)(__umulhi(h, rg) + (mid < p0_hi ? 1u : 0u));
    }
    uint64_t zone = (range << __clzll((long long)range)) - 1ull;
    for (;;) {
        uint64_t l = rng_u32(r), h = rng_u32(r);
        uint64_t v = (h << 32) | l;
        uint64_t lo = v * range;
        if (lo <= zone) return low + __umul64hi(v, range);
    }
}
__device__ __forceinline__ bool does_happen(Rng &r, uint32_t p_inv) { return range32(r, 0, p_inv) == 0; }
// Reservoir choice among n <= 4 candidates taken in order: candidate i replaces the pick when does_happen(i + 1) (maze.rs:73, passages.rs:79).
// Unrolled so that every range is a compile-time constant (zone and multiply fold away).  Returns the index of the pick (n >= 1).
__device__ __forceinline__ int reservoir4(Rng &r, int n) {
    int pick = 0;
    (void)does_happen(r, 1);  // i = 0 always wins but still consumes its draws
    if (n > 1 && does_happen(r, 2)) pick = 1;
    if (n > 2 && does_happen(r, 3)) pick = 2;
    if (n > 3 && does_happen(r, 4)) pick = 3;
    return pick;
}
__device__ __forceinline__ bool parcent(Rng &r, uint32_t p) { return range32(r, 1, 101) <= p; }

// Direction -> (dx, dy) as immediates (a __constant__ table indexed per lane is a memory load); same order as kDX / kDY
constexpr uint32_t dir_pack(const int (&t)[9]) {
    uint32_t r = 0;
    for (int d = 0; d < 9; d++) r |= (uint32_t)(t[d] + 1) << (2 * d);
    return r;
}
constexpr int kDXc[9] = {0, 0, -1, 1, -1, 1, -1, 1, 0}, kDYc[9] = {-1, 1, 0, 0, -1, -1, 1, 1, 0};
__device__ __forceinline__ int dir_dx(int d) { return (int)((dir_pack(kDXc) >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int d) { return (int)((dir_pack(kDYc) >> (2 * d)) & 3u) - 1; }


// nth set bit of a small mask
__device__ __forceinline__ int nth_bit(uint32_t m, int nth) {
    for (int i = 0; i < nth; i++) m &= m - 1;
    return __ffs((int)m) - 1;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t lev_add_of(const RgConfig &c, uint32_t level) { return c.amulet_level < level ? level - c.amulet_level : 0; }


// A store another kernel will read while this one is still running (the spare state k_regen hands to k_step): WT = write-through (`sc1`: a relaxed
// agent-scope atomic store of <= 8 bytes), so that the hand-off needs no release fence.  A release at agent scope is buffer_wbl2 -- the write-back
// of the XCD's WHOLE L2, which the k_step running beside the generator keeps full of dirty lines: 450 of them per step made a 28 us generation
// last up to 100 us and cost the step ~10 us (round 4; MI355X_MICROARCH.md: "16-B sc1 stores + drained flag").
template <bool WT, typename T> __device__ __forceinline__ void st_pub(T *p, T v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

