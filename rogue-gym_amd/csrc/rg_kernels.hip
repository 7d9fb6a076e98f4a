// rg_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the batched Rogue-Gym stepper.
//
//   k_build   : GameConfig::build (core/src/lib.rs:193-228) for every env            1 env / lane
//   k_step    : GameStateImpl::react + ThreadConductor auto-reset                    1 env / lane,
//               (python/src/state_impls.rs:51-79, thread_impls.rs:61-81)             wave-cooperative BFS
//   k_render  : RunTime::draw_screen -> PlayerState.map / history mirrors            1 block / env group
//   k_gray / k_symbol : PlayerState::{gray,symbol}_image[_with_hist] for the batch   16-byte stores
//
// No dense contraction anywhere => no MFMA; everything is integer/byte work bounded by HBM traffic
// (render/encode) or by dependent-load latency and divergence (step/generate).
// file:line citations are relative to /root/reference.


#include "rg_device.h"


#include "rg_gen.h"


// optional phase trace (development aid, tools/microbench.py prof): lane 0 of every wave appends (phase, elapsed shader-clock ticks)
// records to the wave's own 64-word row of S.prof with plain stores -- no atomics, so the trace does not perturb what it measures.
// Word 0 = number of records, word 63 = whole-wave ticks, word 62 = start | end on the 100 MHz clock.
#ifdef RG_FINE_PROF
#define PFM(id) pf.mark(id)
#else
#define PFM(id) ((void)0)
#endif
struct Prof {
    unsigned long long *p; unsigned long long t, t0, r0; int k;
    __device__ __forceinline__ void start(unsigned long long *pp) { p = pp ? pp + (size_t)blockIdx.x * 64 : nullptr; k = 0; if (p) { r0 = __builtin_amdgcn_s_memrealtime(); t = t0 = __builtin_amdgcn_s_memtime(); } }
    __device__ __forceinline__ void rec(int phase, unsigned long long v) {
        if (p && threadIdx.x == 0 && k < 61) p[1 + k++] = ((unsigned long long)phase << 48) | (v & 0xffffffffffffull);
    }
    __device__ __forceinline__ void mark(int phase) {
        if (!p) return;
        unsigned long long now = __builtin_amdgcn_s_memtime();
        rec(phase, now - t);
        t = now;
    }
    __device__ __forceinline__ void finish() {
        // word 62: the wave's start and end on the chip-wide 100 MHz clock (low 32 bits each): where a launch's time goes outside its waves
        if (p && threadIdx.x == 0) { p[0] = (unsigned long long)k; p[63] = __builtin_amdgcn_s_memtime() - t0; p[62] = (r0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull); }
    }
};

// ---------------------------------------------------------------------------------------------
// per-lane environment view
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct Env {
    int e;              // env index
    int n;              // env count (SoA stride)
    uint16_t *cell;     // this env's grid: global memory, or the LDS staging copy while a level is generated
    uint16_t *gcell;    // this env's grid in global memory
    Rng rd, ri, re;     // dungeon / item / enemy streams
    int px, py;
    int hp, hpmax, plvl;
    uint32_t exp, food, quiet, gold, dlevel;
    uint32_t mon_alive, mon_active;
    lds_u16 *lc;        // generation: the lane's LDS staging grid through an LDS-typed pointer (ds_read / ds_write instead of flat accesses)
    uint16_t *stk_lds;  // generation: LDS maze stack of the lane's slot (GEN_STACK_LDS entries), else nullptr
    lds_u32 *mc;        // k_step: this lane's column of the wave's LDS monster cache (word s at mc[s * WAVE]); write-through.  (LDS-typed: one register and
                        // ds_read / ds_write, where the generic pointer of rounds 1-4 was two registers and flat accesses)
    uint32_t err;       // RG_FLAG_ERR_INTERNAL if a capacity guard tripped (each guard carries its proof of unreachability)
    uint32_t on_stairs; // set by place_player: the player was put on the staircase of the level just generated
#ifdef RG_FINE_PROF
    Prof *pfp;
#endif
    // generation: the room table of the level being generated, room i's rect / meta in LANE i of one VGPR each (<= 32 rooms <= 64 lanes).  The
    // generator reads them ~60 times in its RNG-ordered chain (doors, corridors, gold, monsters, placement); from the LDS table each read was a
    // round trip (ds_read, wait, readfirstlane: 100+ cycles), from here it is one v_readlane.  Written to the LDS table once, at the end.
    uint32_t g_rect, g_meta;
    uint32_t g_ea, g_eb;  // ... and the corridor records (edge k in lane k; a level has fewer than 2 x rooms, records >= 64 -- only possible with more
                          // than 32 rooms -- go to GenTabs::edge_a / edge_b in LDS), replayed for the deferred gen_attr draws
};
// GM = the generator instance: 0 up to 32 rooms, 1 up to 64 (both: lane-indexed registers), 2 up to RG_MAX_ROOMS (more rooms than lanes: the
// tables stay in the generator's LDS view S = L of gen_service, column 0 of a one-env SoA)
template <int GM> __device__ __forceinline__ uint32_t gen_rect(const RgState &S, const Env &E, int i) { return GM < 2 ? lane_get(E.g_rect, i) : uni(S.room_rect[i]); }
template <int GM> __device__ __forceinline__ uint32_t gen_meta(const RgState &S, const Env &E, int i) { return GM < 2 ? lane_get(E.g_meta, i) : uni((uint32_t)S.room_meta[i]); }
template <int GM> __device__ __forceinline__ void gen_set_rect(const RgState &S, Env &E, int i, uint32_t v) {
    if (GM < 2) E.g_rect = (int)threadIdx.x == i ? v : E.g_rect; else S.room_rect[i] = v;
}
template <int GM> __device__ __forceinline__ void gen_set_meta(const RgState &S, Env &E, int i, uint32_t v) {
    if (GM < 2) E.g_meta = (int)threadIdx.x == i ? v : E.g_meta; else S.room_meta[i] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------
// monsters table helpers
// ---------------------------------------------------------------------------------------------
// Monster word 0 accessors.  MC = true (the turn code of k_step): read from the wave's LDS cache, loaded once per launch --
// the monster phases re-read the table dozens of times (ordering, blocking tests for 9 directions, overwrite checks), and
// against global memory every pass is another round of VMEM instructions and waits.  Stores go to both.
template <bool MC> __device__ __forceinline__ uint32_t mon_rd(const RgState &S, const Env &E, int s) { return MC ? E.mc[s * WAVE] : uni(S.mon_w0[s * E.n + E.e]); }  // MC = false: generation (wave-uniform)
template <bool MC> __device__ __forceinline__ void mon_wr(const RgState &S, const Env &E, int s, uint32_t w) {
    S.mon_w0[s * E.n + E.e] = w;
    if (MC) E.mc[s * WAVE] = w;
}

__device__ __forceinline__ int mon_find(const RgState &S, const Env &E, int nrooms, uint32_t pos) {
    if (E.mon_alive == 0) return -1;
    int found = -1;
    for (int s = 0; s < nrooms; s++) {
        uint32_t w = mon_rd<true>(S, E, s);
        uint32_t fl = w >> 24;
        if ((fl & MF_ALIVE) && (w & 0xffff) == pos) found = s;
    }
    return found;
}

// EnemyHandler::activate_area (enemies.rs:342-362): wake MEAN sleepers inside room `rid`'s assigned area
template <bool MC>
__device__ __forceinline__ void activate_room(const RgState &S, const RgConfig &c, Env &E, int rid) {
    if (E.mon_alive == E.mon_active) return;
    int nrooms = c.room_num_x * c.room_num_y;
    for (int s = 0; s < nrooms; s++) {
        uint32_t w = mon_rd<MC>(S, E, s);
        uint32_t fl = w >> 24;
        if (!(fl & MF_ALIVE) || (fl & MF_ACTIVE)) continue;
        if (!(c.mon[(w >> 16) & 0xff].attr & EA_MEAN)) continue;
        if (room_id_of(c, POS_X(w), POS_Y(w)) != rid) continue;
        mon_wr<MC>(S, E, s, w | (MF_ACTIVE << 24));
        E.mon_active++;
    }
}

// ---------------------------------------------------------------------------------------------
// field-of-view at level entry (floor.rs:201-312).  The per-step form lives in move_player (register window).
// ---------------------------------------------------------------------------------------------
// Floor::player_in(cd, init = true) on the LDS staging grid, wave-uniform (called from place_player)
template <int GM>
__device__ __forceinline__ void player_in_init(const RgState &S, const RgConfig &c, Env &E, int x, int y) {
    lds_u16 *cell = E.lc;
    int W = c.width;
    int rid = room_id_of(c, x, y);
    if (rid >= 0) {
        uint32_t meta = gen_meta<GM>(S, E, rid);
        if (!(meta & RM_VISITED)) {  // Floor::enters_room (floor.rs:231-247)
            gen_set_meta<GM>(S, E, rid, meta | RM_VISITED);
            if ((meta & RM_KIND_MASK) == RK_NORMAL && !(meta & RM_DARK)) {
                int x0, y0, x1, y1;
                unpack_rect(gen_rect<GM>(S, E, rid), x0, y0, x1, y1);
                const int rw = x1 - x0, area = rw * (y1 - y0);
                for (int t = threadIdx.x; t < area; t += WAVE) {  // one cell per lane
                    const int yy = small_div(t, rw), xx = t - yy * rw;
                    cell[(y0 + yy) * W + x0 + xx] |= C_DRAWN | C_VISIBLE;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        activate_room<false>(S, c, E, rid);
    }
    cell[y * W + x] |= C_VISITED;
    for (int d = 0; d < 9; d++) {
        int cx = x + dir_dx(d), cy = y + dir_dy(d);
        if (!in_bounds(c, cx, cy)) continue;
        uint32_t v = uni(cell[cy * W + cx]);
        bool diag = d >= 4 && d < 8;
        if (diag && (v & C_SURF_MASK) == S_PASSAGE) continue;
        if (v & C_HIDDEN) continue;  // Cell::approached (field.rs:20-26)
        cell[cy * W + cx] = (uint16_t)(v | C_DRAWN | C_VISIBLE);
    }
}

// ---------------------------------------------------------------------------------------------
// generator (rogue/rooms.rs, maze.rs, passages.rs, floor.rs, rogue/mod.rs:434-481)
// ---------------------------------------------------------------------------------------------
// Lane-parallel tile tests for the wave-uniform generator: the cells of a rectangle in row-major order, 64 per round.
// Bit t of the result: cell base + t has one of `bits` set and is not `excl`.
__device__ __forceinline__ uint64_t rect_ballot(const RgConfig &c, const Env &E, int x0, int y0, int rw, int area, int base, uint32_t bits, uint32_t excl) {
    const int t = base + (int)threadIdx.x;
    bool in = false;
    if (t < area) {
        const int yy = small_div(t, rw), xx = t - yy * rw;
        in = (E.lc[(y0 + yy) * c.width + x0 + xx] & bits) && POS(x0 + xx, y0 + yy) != excl;
    }
    return __ballot(in);
}
// position of the nth (0-based) set bit of a ballot mask (nth < popcount)
__device__ __forceinline__ int nth_set64(uint64_t m, int nth) {
    const int lane = threadIdx.x;
    const bool mine = ((m >> lane) & 1ull) && lanes_below(m) == nth;
    return __ffsll((long long)__ballot(mine)) - 1;
}
// nth cell (row-major) of the rectangle with one of `bits` set, skipping `excl`; count must have come from the same test
__device__ __forceinline__ uint32_t rect_nth(const RgConfig &c, const Env &E, int x0, int y0, int rw, int area, uint32_t bits, uint32_t excl, int nth) {
    for (int base = 0; base < area; base += WAVE) {
        const uint64_t m = rect_ballot(c, E, x0, y0, rw, area, base, bits, excl);
        const int cnt = __popcll(m);
        if (nth < cnt) {
            const int t = base + nth_set64(m, nth);
            const int yy = small_div(t, rw);
            return POS(x0 + t - yy * rw, y0 + yy);
        }
        nth -= cnt;
    }
    return POS(x0, y0);
}
__device__ __forceinline__ int rect_count(const RgConfig &c, const Env &E, int x0, int y0, int rw, int area, uint32_t bits, uint32_t excl) {
    int count = 0;
    for (int base = 0; base < area; base += WAVE) count += __popcll(rect_ballot(c, E, x0, y0, rw, area, base, bits, excl));
    return count;
}

// free-cell selection.  The reference keeps a FenwickSet per room; only `nth` over "members minus a
// handful of filled cells" is ever observed during level creation, so the set is
// implicit: interior cells (Normal) or C_MAZE cells (Maze) in row-major order, minus `excl`.
template <int GM>
__device__ __forceinline__ bool room_select(const RgState &S, const RgConfig &c, Env &E, int rid, uint32_t excl /* pos or ~0u */, uint32_t &out) {
    uint32_t meta = gen_meta<GM>(S, E, rid);
    int kind = meta & RM_KIND_MASK;
    if (kind == RK_EMPTY) return false;
    int x0, y0, x1, y1;
    unpack_rect(gen_rect<GM>(S, E, rid), x0, y0, x1, y1);
    if (kind == RK_NORMAL) {
        int iw = x1 - x0 - 2, ih = y1 - y0 - 2;
        int count = iw * ih;
        int eo = -1;
        if (excl != ~0u) { eo = (POS_Y(excl) - y0 - 1) * iw + (POS_X(excl) - x0 - 1); count--; }
        if (count <= 0) return false;
        int nth = (int)range64(E.rd, 0, (uint64_t)count);
        if (eo >= 0 && eo <= nth) nth++;
        int qy = small_div(nth, iw);
        out = POS(x0 + 1 + (nth - qy * iw), y0 + 1 + qy);
        return true;
    }
    const int rw = x1 - x0, area = rw * (y1 - y0);  // Maze: the dug cells of the room, counted and picked 64 cells per round
    const int count = rect_count(c, E, x0, y0, rw, area, C_MAZE, excl);
    if (count == 0) return false;
    const int nth = (int)range64(E.rd, 0, (uint64_t)count);
    out = rect_nth(c, E, x0, y0, rw, area, C_MAZE, excl, nth);
    return true;
}
// Room sets are bit masks, wave-uniform in the generator (scalar unit).  The generator comes in three instances (GM): 0 for room grids of up to 32
// rooms (32-bit sets; every corridor record fits the 64 lanes) -- the only one the W <= 32 step kernel contains, whose descent chain bounds the
// headline launch --, 1 for up to 64 rooms (64-bit sets, corridor records beyond 64 in LDS; measured on the mini dungeon, the 64-bit form costs a
// generation 1.3 us of 28), and 2 for up to RG_MAX_ROOMS = 384 (six 64-bit words; the room table in LDS instead of one room per lane): the
// reference has no limit (rooms.rs:165-211), geometry has -- 160 x 48 with min_room_size 3 holds at most 40 x 9 = 360 rooms.
struct RS6 { uint64_t w[6]; };
template <int GM> struct RoomSet { typedef uint32_t type; };
template <> struct RoomSet<1> { typedef uint64_t type; };
template <> struct RoomSet<2> { typedef RS6 type; };
__device__ __forceinline__ int nth_set64_scalar(uint64_t m, int nth) {
    for (int i = 0; i < nth; i++) m &= m - 1;
    return __ffsll((long long)m) - 1;
}
// -- 32 / 64-bit sets
__device__ __forceinline__ void rs_clear(uint32_t &m) { m = 0; }
__device__ __forceinline__ void rs_clear(uint64_t &m) { m = 0; }
__device__ __forceinline__ void rs_fill(uint32_t &m, int n) { m = n >= 32 ? ~0u : ((1u << n) - 1u); }
__device__ __forceinline__ void rs_fill(uint64_t &m, int n) { m = n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
__device__ __forceinline__ void rs_set(uint32_t &m, int i) { m |= 1u << i; }
__device__ __forceinline__ void rs_set(uint64_t &m, int i) { m |= 1ull << i; }
__device__ __forceinline__ void rs_reset(uint32_t &m, int i) { m &= ~(1u << i); }
__device__ __forceinline__ void rs_reset(uint64_t &m, int i) { m &= ~(1ull << i); }
__device__ __forceinline__ bool rs_test(uint32_t m, int i) { return (m >> (i & 31)) & 1u; }
__device__ __forceinline__ bool rs_test(uint64_t m, int i) { return (m >> (i & 63)) & 1ull; }
__device__ __forceinline__ int rs_count(uint32_t m) { return __popc(m); }
__device__ __forceinline__ int rs_count(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int rs_nth(uint32_t m, int nth) { return nth_bit(m, nth); }
__device__ __forceinline__ int rs_nth(uint64_t m, int nth) { return nth_set64_scalar(m, nth); }
__device__ __forceinline__ bool rs_any(uint32_t m) { return m != 0; }
__device__ __forceinline__ bool rs_any(uint64_t m) { return m != 0; }
__device__ __forceinline__ void rs_remove(uint32_t &m, uint32_t o) { m &= ~o; }
__device__ __forceinline__ void rs_remove(uint64_t &m, uint64_t o) { m &= ~o; }
// -- six-word sets (word index by select chains: a run-time array index would put the set into scratch memory)
__device__ __forceinline__ void rs_clear(RS6 &m) {
#pragma unroll
    for (int j = 0; j < 6; j++) m.w[j] = 0;
}
__device__ __forceinline__ void rs_fill(RS6 &m, int n) {
#pragma unroll
    for (int j = 0; j < 6; j++) { const int r = n - 64 * j; m.w[j] = r >= 64 ? ~0ull : (r > 0 ? ((1ull << r) - 1ull) : 0ull); }
}
__device__ __forceinline__ void rs_set(RS6 &m, int i) {
    const uint64_t b = 1ull << (i & 63); const int k = i >> 6;
#pragma unroll
    for (int j = 0; j < 6; j++) m.w[j] |= j == k ? b : 0ull;
}
__device__ __forceinline__ void rs_reset(RS6 &m, int i) {
    const uint64_t b = 1ull << (i & 63); const int k = i >> 6;
#pragma unroll
    for (int j = 0; j < 6; j++) m.w[j] &= j == k ? ~b : ~0ull;
}
__device__ __forceinline__ bool rs_test(const RS6 &m, int i) {
    const int k = i >> 6; uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) w = j == k ? m.w[j] : w;
    return i >= 0 && ((w >> (i & 63)) & 1ull);
}
__device__ __forceinline__ int rs_count(const RS6 &m) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) c += __popcll(m.w[j]);
    return c;
}
__device__ __forceinline__ int rs_nth(const RS6 &m, int nth) {
    int res = -1;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int c = __popcll(m.w[j]);
        if (res < 0 && nth < c) res = 64 * j + nth_set64_scalar(m.w[j], nth);
        nth -= c;
    }
    return res;
}
__device__ __forceinline__ bool rs_any(const RS6 &m) { return (m.w[0] | m.w[1] | m.w[2] | m.w[3] | m.w[4] | m.w[5]) != 0; }
__device__ __forceinline__ void rs_remove(RS6 &m, const RS6 &o) {
#pragma unroll
    for (int j = 0; j < 6; j++) m.w[j] &= ~o.w[j];
}
// Floor::select_cell (floor.rs:333-346)
template <int GM>
__device__ __forceinline__ bool floor_select(const RgState &S, const RgConfig &c, Env &E, const typename RoomSet<GM>::type &non_empty, int mode /*0 stair, 1 player*/,
                                             uint32_t &out) {
    typename RoomSet<GM>::type cand = non_empty;
    while (rs_any(cand)) {
        int idx = rs_nth(cand, (int)range64(E.rd, 0, (uint64_t)rs_count(cand)));
        uint32_t excl = ~0u;
        if (mode == 0) { uint32_t g = uni(S.gold_pos[idx * E.n + E.e]); if (g & 0x10000u) excl = g & 0xffff; }
        else { uint32_t w = uni(S.mon_w0[idx * E.n + E.e]); if ((w >> 24) & MF_ALIVE) excl = w & 0xffff; }
        if (room_select<GM>(S, c, E, idx, excl, out)) return true;
        rs_reset(cand, idx);
    }
    return false;
}

// gen_attr (floor.rs:420-451) for Passage / Door cells; returns C_HIDDEN / C_LOCKED / 0
__device__ __forceinline__ uint32_t gen_attr_corridor(const RgConfig &c, Env &E, int kind, uint32_t level) {
    if (range32(E.rd, 0, c.dark_level) < level) {
        if (kind == S_PASSAGE) { if (does_happen(E.rd, c.hidden_passage_rate_inv)) return C_HIDDEN; }
        else { if (does_happen(E.rd, c.locked_door_rate_inv)) return C_LOCKED; }
    }
    return 0;
}
// select_start_or_end (passages.rs:143-179).  dir: 0 Up 1 Down 2 Left 3 Right
template <int GM>
__device__ __forceinline__ uint32_t select_door(const RgState &S, const RgConfig &c, Env &E, int rid, int dir) {
    uint32_t meta = gen_meta<GM>(S, E, rid);
    int kind = meta & RM_KIND_MASK;
    int x0, y0, x1, y1;
    unpack_rect(gen_rect<GM>(S, E, rid), x0, y0, x1, y1);
    if (kind == RK_EMPTY) return POS(x0, y0);
    if (kind == RK_NORMAL) {  // edges(range, dir, inclusive): the wall without its corners; SliceRandom::choose = 64-bit draw
        if (dir < 2) {
            int k = (int)range64(E.rd, 0, (uint64_t)(x1 - x0 - 2));
            return POS(x0 + 1 + k, dir == 1 ? y1 - 1 : y0);
        }
        int k = (int)range64(E.rd, 0, (uint64_t)(y1 - y0 - 2));
        return POS(dir == 2 ? x0 : x1 - 1, y0 + 1 + k);
    }
    // Maze: shrink the probe rectangle from the facing side until its edge holds maze cells.
    // Termination: Up / Left probe the edge through the lattice origin (x0, y0), which dig_maze always digs => they return in round 0 (the
    // reference's `start -= 1` quirk of those arms, passages.rs:163-171, is never reached).  Down / Right shrink the far side by one per round
    // and reach the origin row / column after at most y1 - y0 / x1 - x0 <= RG_MAX_W rounds.  The guard below cannot trip; if it ever does, the
    // env raises RG_FLAG_ERR_INTERNAL instead of silently using a wrong door.
    int rx0 = x0, ry0 = y0, rx1 = x1, ry1 = y1;
    for (int guard = 0; guard <= RG_MAX_W && rx0 < rx1 && ry0 < ry1; guard++) {
        // the facing edge of the probe rectangle, clipped to the room: a 1-cell-thick rectangle scanned by the lanes
        int lx0, ly0, lrw, larea;
        if (dir < 2) {
            const int yy = dir == 1 ? ry1 - 1 : ry0;
            lx0 = rx0 > x0 ? rx0 : x0; ly0 = yy;
            const int lx1 = rx1 < x1 ? rx1 : x1;
            lrw = lx1 - lx0; larea = (yy >= y0 && yy < y1 && lrw > 0) ? lrw : 0;
        } else {
            const int xx = dir == 2 ? rx0 : rx1 - 1;
            lx0 = xx; ly0 = ry0 > y0 ? ry0 : y0;
            const int ly1 = ry1 < y1 ? ry1 : y1;
            lrw = 1; larea = (xx >= x0 && xx < x1 && ly1 > ly0) ? ly1 - ly0 : 0;
        }
        if (larea > 0) {
            const int cnt = rect_count(c, E, lx0, ly0, lrw, larea, C_MAZE, ~0u);
            if (cnt) return rect_nth(c, E, lx0, ly0, lrw, larea, C_MAZE, ~0u, (int)range64(E.rd, 0, (uint64_t)cnt));
        }
        if (dir == 1) ry1--; else if (dir == 2) rx0--; else if (dir == 3) rx1--; else ry0--;
    }
    E.err |= RG_FLAG_ERR_INTERNAL;
    return POS(x0, y0);
}

// connect_2rooms (passages.rs:84-133): draws the two doors and the bend now, records the corridor for
// the deferred gen_attr pass (the reference collects Positioned<Surface> in a Vec, floor.rs:73-86)
template <int GM>
__device__ __forceinline__ void connect_rooms(const RgState &S, const RgConfig &c, Env &E, int r1, int r2, int dir, int &n_edges) {
#ifdef RG_FINE_PROF
    Prof &pf = *E.pfp;
#endif
    PFM(40);
    if (dir == 0 || dir == 2) { int t = r1; r1 = r2; r2 = t; dir ^= 1; }
    uint32_t s = select_door<GM>(S, c, E, r1, dir);
    PFM(41);
    uint32_t t = select_door<GM>(S, c, E, r2, dir ^ 1);
    PFM(42);
    int k1 = (gen_meta<GM>(S, E, r1) & RM_KIND_MASK) == RK_NORMAL;
    int k2 = (gen_meta<GM>(S, E, r2) & RM_KIND_MASK) == RK_NORMAL;
    int bend;
    if (dir == 1) bend = (int)range32(E.rd, (uint32_t)(POS_Y(s) + 1), (uint32_t)POS_Y(t));
    else bend = (int)range32(E.rd, (uint32_t)(POS_X(s) + 1), (uint32_t)POS_X(t));
    if (n_edges < 2 * c.room_num_x * c.room_num_y) {  // always: a level has fewer corridors than 2 x rooms, the size of the record tables (rg_state.h)
        const uint32_t ea = s | (t << 16), eb = (uint32_t)bend | ((uint32_t)(dir == 1) << 8) | ((uint32_t)k1 << 9) | ((uint32_t)k2 << 10);
        if (GM == 0 || n_edges < WAVE) {  // (<= 32 rooms: fewer than 64 records)
            E.g_ea = (int)threadIdx.x == n_edges ? ea : E.g_ea;
            E.g_eb = (int)threadIdx.x == n_edges ? eb : E.g_eb;
        } else { S.edge_a[n_edges] = ea; S.edge_b[n_edges] = eb; }  // (S = the generator's LDS table view, gen_service)
        n_edges++;
    } else E.err |= RG_FLAG_ERR_INTERNAL;
    PFM(43);
}
// replay one recorded corridor in registration order (passages.rs:98-132 + floor.rs:87-101): start door, end door, then the three legs.
// The cells of one corridor are distinct, so the lanes fetch them all up front (cell i in lane i), the gen_attr draws run in
// registration order on the scalar unit, and the lanes write their cells back together (register_cell's update rule).
__device__ __forceinline__ void paint_corridor(const RgConfig &c, Env &E, uint32_t a, uint32_t b, uint32_t level) {
    const int W = c.width, lane = threadIdx.x;
    const int sx = POS_X(a), sy = POS_Y(a), ex = POS_X(a >> 16), ey = POS_Y(a >> 16);
    const int bend = b & 0xff;
    const bool down = (b >> 8) & 1;
    const int kind_s = ((b >> 9) & 1) ? S_DOOR : S_PASSAGE, kind_e = ((b >> 10) & 1) ? S_DOOR : S_PASSAGE;
    const int dx = down ? 0 : 1, dy = down ? 1 : 0;
    const int tsx = down ? sx : bend, tsy = down ? bend : sy;
    const int tex = down ? ex : bend, tey = down ? bend : ey;
    const int tdx = down ? (sx < ex ? 1 : -1) : 0, tdy = down ? 0 : (sy < ey ? 1 : -1);
    const int n1 = (down ? bend - sy : bend - sx) - 1;                        // start + 1 .. turn - 1
    const int n2 = down ? (ex > sx ? ex - sx : sx - ex) : (ey > sy ? ey - sy : sy - ey);  // along the turn, its far end excluded
    const int n3 = down ? ey - bend : ex - bend;                              // turn end .. end - 1
    const int total = 2 + n1 + n2 + n3;
    for (int base = 0; base < total; base += WAVE) {
        const int i = base + lane;
        int x = sx, y = sy;
        if (i == 1) { x = ex; y = ey; }
        else if (i >= 2) {
            int k = i - 2;
            if (k < n1) { x = sx + dx * (k + 1); y = sy + dy * (k + 1); }
            else if ((k -= n1) < n2) { x = tsx + tdx * k; y = tsy + tdy * k; }
            else { k -= n2; x = tex + dx * k; y = tey + dy * k; }
        }
        const bool mine = i < total;
        uint32_t v = mine ? (uint32_t)E.lc[y * W + x] : 0u;
        uint32_t attr = 0;
        const int cnt = total - base < WAVE ? total - base : WAVE;
        for (int j = 0; j < cnt; j++) {
            const int kind = base + j == 0 ? kind_s : (base + j == 1 ? kind_e : S_PASSAGE);
            const uint32_t at = gen_attr_corridor(c, E, kind, level);
            if (lane == j) attr = at;
        }
        if (mine) {
            const int kind = i == 0 ? kind_s : (i == 1 ? kind_e : S_PASSAGE);
            v = (v & ~C_ATTR_MASK) | attr;
            if (kind == S_DOOR) v |= C_DOOR;
            if (!attr) v = (v & ~C_SURF_MASK) | (uint32_t)kind;
            E.lc[y * W + x] = (uint16_t)v;
        }
    }
}

// select_candidate (passages.rs:69-82): reservoir over the grid-neighbour rooms not in `excl_mask`, in ascending id = Up, Left, Right, Down
// excl_set: rooms excluded by id (the spanning tree's `selected`); excl_dirs: candidate slots excluded directly (bit k = slot k in the order
// Up, Left, Right, Down: the room graph keeps, per room, which of its four neighbours it is already joined to)
template <typename M>
__device__ __forceinline__ int select_candidate(const RgConfig &c, Env &E, int nrooms, int node, const M &excl_set, uint32_t excl_dirs, int &dir_out) {
    const int rnx = c.room_num_x, rny = c.room_num_y;
    const int ny0 = small_div(node, rnx), nx0 = node - ny0 * rnx;
    // candidate slots in ascending room id; direction codes 0 Up 1 Down 2 Left 3 Right
    const int ids[4] = {node - rnx, node - 1, node + 1, node + rnx};
    const bool ok[4] = {ny0 > 0, nx0 > 0, nx0 + 1 < rnx, ny0 + 1 < rny};
    const int dirs[4] = {0, 2, 3, 1};
    uint32_t cand = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (ok[k] && !rs_test(excl_set, ok[k] ? ids[k] : 0) && !((excl_dirs >> k) & 1u)) cand |= 1u << k;
    if (!cand) return -1;
    const int k = nth_bit(cand, reservoir4(E.rd, __popc(cand)));
    int res = ids[0];
    dir_out = dirs[0];
#pragma unroll
    for (int t = 1; t < 4; t++)
        if (k == t) { res = ids[t]; dir_out = dirs[t]; }
    (void)nrooms;
    return res;
}

// Per-slot generator tables in LDS (gen_service): while a level is generated, the room / monster / gold tables, the corridor records and the
// maze stack are read and written hundreds of times in a dependent chain -- against global memory each access is a round trip (a maze room
// alone cost ~30 us through its global-memory stack).  Copied out to the env's SoA columns when the level is done.
#define GEN_STACK_LDS 128
// Sized by the config's room count nr at run time (the mini dungeon's 4 rooms: 420 bytes; a fixed 64-room layout was 2.9 KB of every step wave's
// LDS and cost the launch ~0.5 us): six tables of nr words, two corridor tables of 2 nr words, the maze stack, nr meta bytes.
struct GenTabs {
    uint32_t *room_rect, *mon_w0, *mon_exp, *gold_pos, *gold_amt;
    int32_t *mon_hp;
    uint32_t *edge_a, *edge_b;
    uint16_t *stack;
    uint8_t *room_meta;
};
#define GEN_GRID_BYTES(hw) ((((size_t)(hw)) * 2 + 15) & ~(size_t)15)
#define GEN_TABS_BYTES(nr) ((((size_t)(nr)) * 42 + GEN_STACK_LDS * 2 + 15) & ~(size_t)15)   // (+ a byte per room behind room_meta: the room graph of GM 2)
#define GEN_SLOT_BYTES(hw, nr) (GEN_GRID_BYTES(hw) + GEN_TABS_BYTES(nr))
__device__ __forceinline__ GenTabs gen_tabs(uint8_t *slot, int hw, int nr) {
    uint32_t *w = reinterpret_cast<uint32_t *>(slot + GEN_GRID_BYTES(hw));
    GenTabs T;
    T.room_rect = w; T.mon_w0 = w + nr; T.mon_exp = w + 2 * nr; T.gold_pos = w + 3 * nr; T.gold_amt = w + 4 * nr;
    T.mon_hp = reinterpret_cast<int32_t *>(w + 5 * nr);
    T.edge_a = w + 6 * nr; T.edge_b = w + 8 * nr;
    T.stack = reinterpret_cast<uint16_t *>(w + 10 * nr);
    T.room_meta = reinterpret_cast<uint8_t *>(T.stack + GEN_STACK_LDS);
    return T;
}

// dig_maze (maze.rs:38-89) with an explicit stack (the reference recurses; same visiting and draw order)
__device__ __forceinline__ void dig_maze(const RgState &S, const RgConfig &c, Env &E, int x0, int y0, int x1, int y1) {
    // maze nodes = every other cell of the room in x and y; the DFS holds at most one stack entry per node
    const int mw = (x1 - x0 + 1) >> 1, mh = (y1 - y0 + 1) >> 1;
    lds_u16 *ls = (lds_u16 *)E.stk_lds;
    uint16_t *gs = S.maze_stack + (size_t)E.e * S.maze_cap;
    const bool in_lds = E.stk_lds && mw * mh <= GEN_STACK_LDS;
    const bool bitmap = mw * mh <= 64;  // dug nodes as a 64-bit scalar mask: the neighbour tests never touch memory
    const int W = c.width;
    const uint16_t dug = (uint16_t)(S_NONE | C_MAZE);  // the room's area is untouched (fresh Field) until its maze is painted
    E.lc[y0 * W + x0] = dug;
    uint64_t seen = 1ull;
    int stk_v = 0;
    int cx = x0, cy = y0, sp = 1;  // sp counts the current cell as the reference's recursion depth does
    for (;;) {
        uint32_t cand = 0;
        if (bitmap) {
            const int ix = (cx - x0) >> 1, iy = (cy - y0) >> 1;
            if (iy > 0 && !((seen >> ((iy - 1) * mw + ix)) & 1ull)) cand |= 1u;        // Up
            if (iy + 1 < mh && !((seen >> ((iy + 1) * mw + ix)) & 1ull)) cand |= 2u;  // Down
            if (ix > 0 && !((seen >> (iy * mw + ix - 1)) & 1ull)) cand |= 4u;          // Left
            if (ix + 1 < mw && !((seen >> (iy * mw + ix + 1)) & 1ull)) cand |= 8u;    // Right
        } else {  // the four candidate cells tested by lanes 0..3 at once
            const int d = (int)threadIdx.x & 3;
            const int nx = cx + 2 * dir_dx(d), ny = cy + 2 * dir_dy(d);
            const bool ok = threadIdx.x < 4 && !(nx < x0 || nx >= x1 || ny < y0 || ny >= y1) && !(E.lc[ny * W + nx] & C_MAZE);
            cand = (uint32_t)__ballot(ok) & 0xfu;
        }
        int dig = -1;
        if (cand) dig = nth_bit(cand, reservoir4(E.rd, __popc(cand)));  // candidates in direction order Up, Down, Left, Right
        if (dig < 0) {  // dead end: back to the parent
            if (--sp == 0) break;
            const uint32_t top = bitmap ? (uint32_t)__builtin_amdgcn_readlane(stk_v, sp - 1) : uni(in_lds ? ls[sp - 1] : gs[sp - 1]);
            cx = POS_X(top); cy = POS_Y(top);
            continue;
        }
        const int ddx = dir_dx(dig), ddy = dir_dy(dig);
        E.lc[(cy + ddy) * W + cx + ddx] = dug;
        E.lc[(cy + 2 * ddy) * W + cx + 2 * ddx] = dug;
        if (bitmap) seen |= 1ull << ((((cy + 2 * ddy) - y0) >> 1) * mw + (((cx + 2 * ddx) - x0) >> 1));
        const int stack_cap = bitmap ? 64 : (in_lds ? GEN_STACK_LDS : S.maze_cap);  // every form holds one entry per maze node of the room (maze_cap: rg_api.cpp)
        if (sp >= stack_cap) E.err |= RG_FLAG_ERR_INTERNAL;  // never: the DFS path visits a node at most once
        if (sp < stack_cap) {  // descend: the current cell goes on the stack
            const uint16_t cur = (uint16_t)POS(cx, cy);
            if (bitmap) stk_v = (int)threadIdx.x == sp - 1 ? (int)cur : stk_v;  // <= 64 nodes: the stack is one VGPR, entry i in lane i
            else if (in_lds) ls[sp - 1] = cur;
            else gs[sp - 1] = cur;
            sp++;
            cx += 2 * ddx; cy += 2 * ddy;
        }
    }
}

// Dungeon::new_level_ (rogue/mod.rs:434-481).  Returns the bitmask of non-empty rooms.
// Dungeon::new_level in two halves.  gen_structure: everything up to the stairs -- it draws on the dungeon stream and (gold) on the item stream only,
// and the item stream is touched by nothing but level generation, so its outcome is a pure function of (level, dungeon stream, item stream).
// gen_populate: the monsters (dungeon stream for the cell, enemy stream for everything else) and the reveal.  k_regen runs the first half ahead of a
// descent ("next-level structures", gen_service), the descent then only runs the second.
template <int GM>
__device__ __forceinline__ typename RoomSet<GM>::type gen_structure(const RgState &S, const RgConfig &c, Env &E, Prof &pf) {
    typedef typename RoomSet<GM>::type rmask_t;
    const int W = c.width, HW = W * c.height, n = E.n, e = E.e;
    const int rnx = c.room_num_x, nrooms = rnx * c.room_num_y;
    lds_u16 *cell = E.lc;
    const uint32_t level = ++E.dlevel;

    // fresh Field: Surface::None, no attributes (16-byte stores; the grid is 16-byte aligned)
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t nn = S_NONE | (S_NONE << 16);
        u32x4 v = {nn, nn, nn, nn};
        __attribute__((address_space(3))) u32x4 *p = (__attribute__((address_space(3))) u32x4 *)cell;
        int n16 = HW / 8;
        for (int i = threadIdx.x; i < n16; i += WAVE) p[i] = v;
        for (int i = n16 * 8 + threadIdx.x; i < HW; i += WAVE) cell[i] = S_NONE;
    }
    for (int s = 0; s < nrooms; s++) {  // remove_enemies + fresh Floor::items
        S.mon_w0[s * n + e] = 0;
        S.gold_pos[s * n + e] = 0;
    }
    E.mon_alive = E.mon_active = 0;

    pf.mark(8);
    // ---- gen_rooms (rooms.rs:165-211) ----
    uint32_t empty_num = range32(E.rd, 0, c.max_empty_rooms + 1);
    if (empty_num >= (uint32_t)nrooms) empty_num = nrooms - 1;
    rmask_t empty_mask;
    rs_clear(empty_mask);
    {
        rmask_t sel;
        rs_fill(sel, nrooms);
        for (uint32_t k = 0; k < empty_num; k++) {  // rng.select(0..room_num).take(empty_num): 64-bit nth
            int id = rs_nth(sel, (int)range64(E.rd, 0, (uint64_t)rs_count(sel)));
            rs_reset(sel, id);
            rs_set(empty_mask, id);
        }
    }
    for (int i = 0; i < nrooms; i++) {  // make_room (rooms.rs:214-269)
        int ax0, ay0, ax1, ay1;
        assigned_area(c, i, ax0, ay0, ax1, ay1);
        int rsx = ax1 - ax0, rsy = ay1 - ay0;
        uint32_t rect; uint8_t meta;
        if (rs_test(empty_mask, i)) {
            int x = (int)range32(E.rd, 1, (uint32_t)(rsx - 1)) + ax0;
            int y = (int)range32(E.rd, 1, (uint32_t)(rsy - 1)) + ay0;
            rect = (uint32_t)x | ((uint32_t)y << 8);
            meta = RK_EMPTY | RM_DARK;
        } else {
            bool dark = range32(E.rd, 0, c.dark_level) < level;
            if (dark && does_happen(E.rd, c.maze_rate_inv)) {
                int mx1 = ax0 + rsx - 1, my1 = ay0 + rsy - 1;
                rect = (uint32_t)ax0 | ((uint32_t)ay0 << 8) | ((uint32_t)mx1 << 16) | ((uint32_t)my1 << 24);
                meta = RK_MAZE | RM_DARK;
                dig_maze(S, c, E, ax0, ay0, mx1, my1);
            } else {
                int sx = (int)range32(E.rd, (uint32_t)c.min_room_x, (uint32_t)rsx);
                int sy = (int)range32(E.rd, (uint32_t)c.min_room_y, (uint32_t)rsy);
                int ox = (int)range32(E.rd, 0, (uint32_t)(rsx - sx)) + ax0;
                int oy = (int)range32(E.rd, 0, (uint32_t)(rsy - sy)) + ay0;
                rect = (uint32_t)ox | ((uint32_t)oy << 8) | ((uint32_t)(ox + sx) << 16) | ((uint32_t)(oy + sy) << 24);
                meta = RK_NORMAL | (dark ? RM_DARK : 0);
            }
        }
        gen_set_rect<GM>(S, E, i, rect);
        gen_set_meta<GM>(S, E, i, meta);
    }
    pf.mark(9);
    // ---- paint rooms in id order (floor.rs:61-71; Room::draw rooms.rs:58-82) ----
    for (int i = 0; i < nrooms; i++) {
        uint32_t meta = gen_meta<GM>(S, E, i);
        int kind = meta & RM_KIND_MASK;
        if (kind == RK_EMPTY) continue;
        int x0, y0, x1, y1;
        unpack_rect(gen_rect<GM>(S, E, i), x0, y0, x1, y1);
        if (kind == RK_NORMAL) {
            const uint16_t fl = (uint16_t)(S_FLOOR | ((meta & RM_DARK) ? C_DARK : 0));
            const int rw = x1 - x0, rh = y1 - y0, area = rw * rh;
            for (int t = threadIdx.x; t < area; t += WAVE) {  // one cell per lane: top / bottom walls (corners included), side walls, floor
                const int yy = small_div(t, rw), xx = t - yy * rw;
                cell[(y0 + yy) * W + x0 + xx] = (yy == 0 || yy == rh - 1) ? (uint16_t)S_WALLX : ((xx == 0 || xx == rw - 1) ? (uint16_t)S_WALLY : fl);
            }
        } else {  // maze cells in ascending range index; each draws gen_attr (Passage)
            const int rw = x1 - x0, area = rw * (y1 - y0);
            for (int base = 0; base < area; base += WAVE) {
                uint64_t mm = rect_ballot(c, E, x0, y0, rw, area, base, C_MAZE, ~0u);
                while (mm) {
                    const int t = base + __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int yy = small_div(t, rw), xx = t - yy * rw;
                    const uint32_t a = gen_attr_corridor(c, E, S_PASSAGE, level);
                    cell[(y0 + yy) * W + x0 + xx] = (uint16_t)(C_MAZE | S_PASSAGE | a);
                }
            }
        }
    }
    pf.mark(10);
    // ---- dig_passges (passages.rs:16-67) ----
    int n_edges = 0;
    {
        // the room graph (passages.rs:222-270): per room, which of its four grid neighbours it is already joined to -- 4 bits (slot order Up, Left, Right,
        // Down), room i's in LANE i of one VGPR (<= 64 rooms = 64 lanes), read with v_readlane and updated with a lane-select.  (A `conn[rooms]`
        // array indexed at run time lived in scratch memory: a memory round trip per access inside this RNG-ordered chain.)
        // (GM 2, more rooms than lanes: a byte per room in LDS, right behind the generator's room_meta table -- gen_tabs)
        uint32_t conn_v = 0;
        uint8_t *conn_lds = S.room_meta + nrooms;
        if (GM == 2) {
            for (int i = threadIdx.x; i < nrooms; i += WAVE) conn_lds[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        const int lane_id = (int)threadIdx.x;
        auto conn_get = [&](int i) -> uint32_t { return GM < 2 ? lane_get(conn_v, i) : uni((uint32_t)conn_lds[i]); };
        auto conn_join = [&](int a, int b, int dir_ab) {  // dir_ab: 0 Up 1 Down 2 Left 3 Right, as seen from a
            const uint32_t slot_of_dir = 0x2130u;         // direction code -> candidate slot (nibbles): Up 0, Down 3, Left 1, Right 2
            const uint32_t ka = (slot_of_dir >> (4 * dir_ab)) & 3u, kb = (slot_of_dir >> (4 * (dir_ab ^ 1))) & 3u;
            if (GM < 2) conn_v = lane_id == a ? (conn_v | (1u << ka)) : (lane_id == b ? (conn_v | (1u << kb)) : conn_v);
            else { conn_lds[a] = (uint8_t)(uni((uint32_t)conn_lds[a]) | (1u << ka)); conn_lds[b] = (uint8_t)(uni((uint32_t)conn_lds[b]) | (1u << kb)); }
        };
        rmask_t selected, none;
        rs_clear(selected); rs_clear(none);
        int cur = (int)range64(E.rd, 0, (uint64_t)nrooms), n_sel = 1;
        rs_set(selected, cur);
        while (n_sel < nrooms) {
            int dir = 0;
            int nxt = select_candidate<rmask_t>(c, E, nrooms, cur, selected, 0u, dir);
            if (nxt >= 0) {
                rs_set(selected, nxt); n_sel++;
                conn_join(cur, nxt, dir);
                connect_rooms<GM>(S, c, E, cur, nxt, dir, n_edges);
            } else {
                cur = rs_nth(selected, (int)range64(E.rd, 0, (uint64_t)n_sel));
            }
        }
        uint32_t try_num = range32(E.rd, 0, c.max_extra_edges);
        for (uint32_t t = 0; t < try_num; t++) {
            int room1 = (int)range64(E.rd, 0, (uint64_t)nrooms), dir = 0;
            int room2 = select_candidate<rmask_t>(c, E, nrooms, room1, none, conn_get(room1), dir);
            if (room2 >= 0) {
                conn_join(room1, room2, dir);
                connect_rooms<GM>(S, c, E, room1, room2, dir, n_edges);
            }
        }
    }
    pf.mark(11);
    for (int k = 0; k < n_edges; k++) {
        if (GM == 0 || k < WAVE) paint_corridor(c, E, lane_get(E.g_ea, k), lane_get(E.g_eb, k), level);
        else paint_corridor(c, E, uni(S.edge_a[k]), uni(S.edge_b[k]), level);
    }

    rmask_t non_empty;
    rs_fill(non_empty, nrooms);
    rs_remove(non_empty, empty_mask);
    pf.mark(12);
    // ---- gold (floor.rs:132-153, item/gold.rs:18-24) ----
    for (int i = 0; i < nrooms; i++) {
        uint32_t pos;
        if (!room_select<GM>(S, c, E, i, ~0u, pos)) continue;
        if (!does_happen(E.ri, c.gold_rate_inv)) continue;
        uint32_t num = range32(E.ri, 0, c.gold_base + c.gold_per_level * level) + c.gold_minimum;
        S.gold_pos[i * n + e] = pos | 0x10000u;
        S.gold_amt[i * n + e] = num;
        gen_set_meta<GM>(S, E, i, gen_meta<GM>(S, E, i) | RM_HAS_GOLD);
        cell[POS_Y(pos) * W + POS_X(pos)] |= C_GOLD;
    }
    pf.mark(13);
    // ---- stair (floor.rs:156-167) ----
    {
        uint32_t pos;
        if (floor_select<GM>(S, c, E, non_empty, 0, pos)) {
            uint32_t v = uni(cell[POS_Y(pos) * W + POS_X(pos)]);
            cell[POS_Y(pos) * W + POS_X(pos)] = (uint16_t)((v & ~C_SURF_MASK) | S_STAIR);
        }
    }
    pf.mark(14);
    return non_empty;
}
template <int GM>
__device__ __forceinline__ void gen_populate(const RgState &S, const RgConfig &c, Env &E, Prof &pf) {
    const int W = c.width, H = c.height, n = E.n, e = E.e;
    const int nrooms = c.room_num_x * c.room_num_y;
    lds_u16 *cell = E.lc;
    const uint32_t level = E.dlevel;
    // ---- monsters (floor.rs:106-130, enemies.rs:265-320) ----
    if (c.n_enemies > 0) {
        uint32_t mn = level >= 4 ? level - 4 : 0, mx = level + 6;
        uint32_t lev_add = lev_add_of(c, level);
        for (int i = 0; i < nrooms; i++) {
            uint32_t pos;
            PFM(44);
            if (!room_select<GM>(S, c, E, i, ~0u, pos)) continue;
            PFM(45);
            bool has_gold = gen_meta<GM>(S, E, i) & RM_HAS_GOLD;
            if (!parcent(E.re, has_gold ? c.appear_rate_gold : c.appear_rate_nogold)) continue;
            uint32_t len = (uint32_t)c.n_enemies;
            uint32_t idx = range32(E.re, mn, mx);
            if (idx > len) { uint32_t rg = len < 5 ? len : 5; idx = (uint32_t)range64(E.re, len - rg, len); }
            if (idx >= len) continue;
            uint32_t type = idx;
            PFM(46);
            int64_t mlevel = (int64_t)c.mon[type].level + lev_add, hp = 0;
            uint32_t exp_add;
            if (mlevel >= 1 && mlevel < (1 << 24)) {  // every real table: the eight rolls sum to < 2^27, so 32-bit adds and a 32-bit divide (the i64 `/ 6` alone is ~40 instructions)
                const uint32_t ml = (uint32_t)mlevel;
                uint32_t h32 = 0;
                for (int k = 0; k < 8; k++) h32 += (uint32_t)range64(E.re, 1, (uint64_t)ml + 1);
                const uint32_t base = ml == 1 ? h32 / 8u : h32 / 6u;
                exp_add = ml >= 10 ? base * 20u : base * 4u;
                hp = h32;
            } else {
                for (int k = 0; k < 8; k++) hp += (int64_t)range64(E.re, 1, (uint64_t)mlevel + 1);
                int64_t base = mlevel == 1 ? hp / 8 : hp / 6;
                exp_add = mlevel >= 10 ? (uint32_t)base * 20u : (uint32_t)base * 4u;
            }
            PFM(47);
            S.mon_w0[i * n + e] = pos | (type << 16) | ((uint32_t)MF_ALIVE << 24);
            S.mon_hp[i * n + e] = (int32_t)hp;
            S.mon_exp[i * n + e] = c.mon[type].exp + lev_add * 10u + exp_add;
            E.mon_alive++;
            PFM(48);
        }
    }
    pf.mark(15);
    if (!c.hide_dungeon)
        for (int i = W + (int)threadIdx.x; i < (H - 1) * W; i += WAVE) cell[i] |= C_VISIBLE;  // rows 1..H-2
}

// actions::new_level's tail (actions.rs:130-137): place the player and enter the room
template <int GM>
__device__ __forceinline__ void place_player(const RgState &S, const RgConfig &c, Env &E, const typename RoomSet<GM>::type &non_empty) {
    uint32_t pos = 0;
    floor_select<GM>(S, c, E, non_empty, 1, pos);
    E.px = POS_X(pos); E.py = POS_Y(pos);
    E.on_stairs = (uni(E.lc[E.py * c.width + E.px]) & C_SURF_MASK) == S_STAIR;  // (select_cell only avoids characters: the stairs are a legal spot)
    player_in_init<GM>(S, c, E, E.px, E.py);
}

// GameConfig::build (core/src/lib.rs:193-228), split around the level generator
// GameConfig::to_global's seed choice (core/src/lib.rs:157-165).  A configured seed is used as is.  `seed: None` draws a fresh seed for EVERY
// build (rng::gen_seed / gen_ranged_seed use thread_rng, so the values themselves are not parity-relevant): build k of the env uses
// hash(base, k) with k taken atomically -- the inline generation of k_step and a k_regen running concurrently on the side stream each get
// their own k, and nothing is read-modify-written non-atomically.  With a seed_range the value is uniform in [r0, r1) by mask rejection
// (no 128-bit division on the device).
__device__ __forceinline__ void build_prologue(const RgState &S, Env &E) {
    // (E.e may be an entry of the spare view, slot * n + env -- rg_state.h sp_slots: the seed arrays are the envs' own)
    const int se = E.e >= S.n ? E.e % S.n : E.e;
    uint64_t lo = uni((uint32_t)S.seed_lo[se]) | ((uint64_t)uni((uint32_t)(S.seed_lo[se] >> 32)) << 32);
    uint64_t hi = uni((uint32_t)S.seed_hi[se]) | ((uint64_t)uni((uint32_t)(S.seed_hi[se] >> 32)) << 32);
    const uint32_t mode = uni(S.reseed[se]);
    if (mode) {
        uint32_t k = 0;
        if (threadIdx.x == 0) k = atomicAdd(&S.build_ctr[se], 1u);  // wave-uniform caller: one lane takes the ticket
        k = uni(k);
        uint64_t z = splitmix64(lo ^ splitmix64(hi + k)), y = splitmix64(z ^ hi);
        if (mode == 2 && S.range_lo) {
            const int n = S.n, e = se;
            const uint64_t r_lo = S.range_lo[e], r_hi = S.range_lo[n + e], sp_lo = S.range_span[e], sp_hi = S.range_span[n + e];
            uint64_t m_lo, m_hi;  // smallest 2^b - 1 >= span - 1
            if (sp_hi) { m_lo = ~0ull; m_hi = ~0ull >> __clzll((long long)sp_hi); }
            else { m_hi = 0; m_lo = sp_lo > 1 ? ~0ull >> __clzll((long long)(sp_lo - 1)) : 0ull; }
            for (int t = 0; t < 64; t++) {  // accepts with p >= 1/2 per round
                const uint64_t c_lo = z & m_lo, c_hi = y & m_hi;
                if (c_hi < sp_hi || (c_hi == sp_hi && c_lo < sp_lo) || t == 63) { z = c_lo; y = c_hi; break; }
                z = splitmix64(z); y = splitmix64(y ^ z);
            }
            if (!(y < sp_hi || (y == sp_hi && z < sp_lo))) { z = 0; y = 0; }  // 2^-63: fall back to r0 rather than leave the range
            lo = r_lo + z; hi = r_hi + y + (lo < r_lo ? 1ull : 0ull);
        } else { lo = z; hi = y; }
    }
    rng_seed(E.ri, lo, hi); rng_seed(E.re, lo, hi); rng_seed(E.rd, lo, hi);
    E.dlevel = 0;
}
__device__ __forceinline__ void build_epilogue(const RgState &S, const RgConfig &c, Env &E) {
    // Player::init_items (player.rs:136-153): every InitItem::Weapon of the config draws `rng.range(init_num)` on the item stream, in list order
    // (WeaponStatus::build, weapon.rs:159); the default pack is mace 1..2, bow 1..2, arrow 8..17 (weapon.rs:179-188).  Resolved by rg_items.cpp.
    for (int i = 0; i < c.n_init_draws; i++) (void)range32(E.ri, S.init_draws[2 * i], S.init_draws[2 * i + 1]);
    E.hp = E.hpmax = c.init_hp; E.plvl = 1; E.exp = 0;
    E.food = c.hunger_time; E.quiet = 0; E.gold = c.init_gold;
}

// Level generation service.  Generating a level is one long RNG-ordered chain of decisions and tile reads/writes with no parallelism across
// the steps of one level and very little in common between two levels (lanes generating side by side diverge almost everywhere).  So the wave
// generates ONE level at a time, all 64 lanes working on the same env: the scalars are wave-uniform (scalar unit: RNG, loop control,
// decisions), the grid and the generator's tables live in LDS, and the bulk tile work (clear, room paint, reveals, copy-out) is spread
// over the lanes.  Requests of several lanes are served one after the other.  is_build: GameConfig::build, else
// Dungeon::new_level (rogue/mod.rs:434-481) followed by actions::new_level's player placement.
__device__ __forceinline__ void env_from_lane(Env &U, const Env &E, int src) {
    U.e = (int)lane_get((uint32_t)E.e, src); U.n = (int)lane_get((uint32_t)E.n, src);  // (the SoA stride is the request's: a spare-view entry and a next-level request may share a wave)
    uint32_t *d = reinterpret_cast<uint32_t *>(&U.rd);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(&E.rd);
#pragma unroll
    for (int i = 0; i < 12; i++) d[i] = lane_get(q[i], src);  // rd, ri, re
    U.px = (int)lane_get((uint32_t)E.px, src); U.py = (int)lane_get((uint32_t)E.py, src);
    U.hp = (int)lane_get((uint32_t)E.hp, src); U.hpmax = (int)lane_get((uint32_t)E.hpmax, src); U.plvl = (int)lane_get((uint32_t)E.plvl, src);
    U.exp = lane_get(E.exp, src); U.food = lane_get(E.food, src); U.quiet = lane_get(E.quiet, src); U.gold = lane_get(E.gold, src);
    U.dlevel = lane_get(E.dlevel, src); U.mon_alive = lane_get(E.mon_alive, src); U.mon_active = lane_get(E.mon_active, src);
    U.err = 0; U.on_stairs = 0;
}
__device__ __forceinline__ void env_to_lane(Env &E, const Env &U) {  // the generated env's scalars back into its own lane
    E.rd = U.rd; E.ri = U.ri; E.re = U.re;
    E.px = U.px; E.py = U.py; E.hp = U.hp; E.hpmax = U.hpmax; E.plvl = U.plvl;
    E.exp = U.exp; E.food = U.food; E.quiet = U.quiet; E.gold = U.gold; E.dlevel = U.dlevel; E.mon_alive = U.mon_alive; E.mon_active = U.mon_active;
    E.err |= U.err; E.on_stairs = U.on_stairs;
}
static_assert(sizeof(Rng) == 16, "Rng is 4 words");

// Next-level structures.  A descent generates its level inside the turn, and the wave that does is the longest chain of a launch (mini: 33 of 51 us,
// the default dungeon: 71 of 107 us -- the launch lasts as long as that wave).  Two thirds of that generation (gen_structure) draw only on the dungeon and
// item streams, which play leaves alone except for a successful search next to a hidden passage / locked door and the moves of an erratic monster
// (rogue/mod.rs:376-397, floor.rs:349-370): between a level's generation and the descent out of it the dungeon stream was untouched in 89 of 91
// descents of the random policy (tools/tmp measurements in profiles/r04_experiments.txt).  So when a player comes near the stairs (k_step: a staircase in
// his 5x5 window) the env ASKS for the structure of the next level; k_regen generates it beside the following step from the env's streams as they are
// then, and writes it through with the dungeon stream it started from as the KEY; the descent compares the key with its own dungeon stream and, on a
// match, loads the structure into the generator's LDS slot and runs gen_populate + place_player only.  No match (not asked, not ready, stream moved
// on): the level is generated inline as before -- the result is the same bits either way, the structure only moves work off the critical wave.
//
// load_structure: the generator's state as gen_structure would have left it (grid, room / gold tables, streams, level, non-empty set).
template <int GM>
__device__ __forceinline__ typename RoomSet<GM>::type load_structure(const RgState &S, const RgConfig &c, Env &U, int real_e, int real_n, const GenTabs *T) {
    const int HW = c.width * c.height, nrooms = c.room_num_x * c.room_num_y, lane = threadIdx.x;
    const RgNext *NX = S.nx;  // (each pointer fetched where it is used: the copy of the struct held eight of them in SGPRs through the whole load)
    const uint16_t *src = NX->cell + (size_t)real_e * HW;
    if ((HW & 7) == 0) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        __attribute__((address_space(3))) u32x4 *d4 = (__attribute__((address_space(3))) u32x4 *)U.lc;
        const u32x4 *s4 = reinterpret_cast<const u32x4 *>(src);
#pragma nounroll
        for (int i = lane; i < HW / 8; i += WAVE) d4[i] = s4[i];
    } else {
#pragma nounroll
        for (int i = lane; i < HW; i += WAVE) U.lc[i] = src[i];
    }
    uint32_t meta = RK_EMPTY;
    if (lane < nrooms) {
        const size_t g = (size_t)lane * real_n + real_e;
        U.g_rect = NX->room_rect[g]; meta = NX->room_meta[g]; U.g_meta = meta;
        T->gold_pos[lane] = NX->gold_pos[g]; T->gold_amt[lane] = NX->gold_amt[g]; T->mon_w0[lane] = 0;
    }
    const uint32_t *q = S.nx_rng + real_e;
    U.rd = {uni(q[4 * (size_t)real_n]), uni(q[5 * (size_t)real_n]), uni(q[6 * (size_t)real_n]), uni(q[7 * (size_t)real_n])};
    U.ri = {uni(q[8 * (size_t)real_n]), uni(q[9 * (size_t)real_n]), uni(q[10 * (size_t)real_n]), uni(q[11 * (size_t)real_n])};
    U.dlevel++;
    U.mon_alive = U.mon_active = 0;
    const uint64_t ne = __ballot(lane < nrooms && (meta & RM_KIND_MASK) != RK_EMPTY);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return (typename RoomSet<GM>::type)ne;
}

// from_nx / struct_only: bit i = lane i's level comes from its next-level structure / lane i asks for the structure alone (k_regen)
template <int GM, bool WT = false>
__device__ __forceinline__ void gen_service(const RgState &S, const RgConfig &c, Env &E, int lane, int e, bool need, bool is_build_in,
                                            uint16_t *lds_grid, Prof &pf, const uint64_t from_nx = 0, const uint64_t struct_only = 0) {
    const int HW = c.width * c.height, nrooms = c.room_num_x * c.room_num_y;
    uint8_t *slot = reinterpret_cast<uint8_t *>(lds_grid);
    const GenTabs TT = gen_tabs(slot, HW, nrooms), *T = &TT;
    uint64_t m = __ballot(need);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        unsigned long long tg0 = pf.p ? __builtin_amdgcn_s_memtime() : 0;
        Env U;
        env_from_lane(U, E, src);
        const int real_e = U.e, real_n = U.n;
        U.cell = U.gcell = reinterpret_cast<uint16_t *>(slot);
        U.lc = (lds_u16 *)U.cell;
        U.stk_lds = T->stack; U.mc = nullptr;
        const bool so = GM < 2 && ((struct_only >> src) & 1ull), is_build = is_build_in && !so;
        if (is_build) build_prologue(S, U);
        // table view: column 0 of a 1-env SoA in LDS
        RgState L = S;
        L.room_rect = T->room_rect; L.room_meta = T->room_meta; L.mon_w0 = T->mon_w0; L.mon_hp = T->mon_hp; L.mon_exp = T->mon_exp;
        L.gold_pos = T->gold_pos; L.gold_amt = T->gold_amt; L.edge_a = T->edge_a; L.edge_b = T->edge_b;
        L.maze_stack = S.maze_stack + (size_t)(real_e >= S.n ? real_e % S.n : real_e) * S.maze_cap;  // (per env, also for a spare-view entry)
        U.e = 0; U.n = 1;
        U.g_rect = U.g_meta = U.g_ea = U.g_eb = 0;
#ifdef RG_FINE_PROF
        U.pfp = &pf;
#endif
        // ONE site each for the two halves of the generator, whatever the request (whole level, structure alone, level from a structure): a second
        // inlined copy of either was 25 KB of code in k_regen and 6-8 KB in every step kernel -- instruction-cache space the turn code needs
        typename RoomSet<GM>::type non_empty;
        bool loaded = false;
        if constexpr (GM < 2) {
            if ((from_nx >> src) & 1ull) { non_empty = load_structure<GM>(S, c, U, real_e, real_n, T); loaded = true; }
        }
        if (!loaded) non_empty = gen_structure<GM>(L, c, U, pf);
        if constexpr (GM < 2) {
            if (so) {  // the structure alone, written through to the env's nx_* arrays with the stream it started from (k_regen)
                const RgNext NX = *S.nx;
                if (lane < nrooms) { T->room_rect[lane] = U.g_rect; T->room_meta[lane] = (uint8_t)U.g_meta; }
                __syncthreads();
                if (lane < nrooms) {
                    const size_t g = (size_t)lane * real_n + real_e;
                    st_pub<true>(&NX.room_rect[g], T->room_rect[lane]); st_pub<true>(&NX.room_meta[g], T->room_meta[lane]);
                    st_pub<true>(&NX.gold_pos[g], T->gold_pos[lane]); st_pub<true>(&NX.gold_amt[g], T->gold_amt[lane]);
                }
                {
                    uint16_t *dst = NX.cell + (size_t)real_e * HW;
                    const uint16_t *srcp = reinterpret_cast<const uint16_t *>(slot);
                    if ((HW & 7) == 0) {
                        unsigned long long *d8 = reinterpret_cast<unsigned long long *>(dst);
                        const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(srcp);
                        for (int i = lane; i < HW / 4; i += WAVE) st_pub<true>(&d8[i], s8[i]);
                    } else
                        for (int i = lane; i < HW; i += WAVE) st_pub<true>(&dst[i], srcp[i]);
                }
                if (lane < 8) {  // the streams after it (words 0..3, the dungeon stream it started from, and the level are the request's own: the key)
                    const uint32_t r[8] = {U.rd.x, U.rd.y, U.rd.z, U.rd.w, U.ri.x, U.ri.y, U.ri.z, U.ri.w};
                    uint32_t mine = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) mine = lane == k ? r[k] : mine;
                    st_pub<true>(&S.nx_rng[(size_t)(4 + lane) * real_n + real_e], mine);
                }
                if (lane == src) E.err |= U.err;  // a capacity guard that fired: the caller raises it and does NOT publish the structure
                __syncthreads();
                continue;
            }
        }
        gen_populate<GM>(L, c, U, pf);
        pf.mark(17);
        if (is_build) build_epilogue(L, c, U);
        place_player<GM>(L, c, U, non_empty);
        pf.mark(18);
        U.e = real_e; U.n = real_n;
        if (GM < 2 && lane < nrooms) { T->room_rect[lane] = U.g_rect; T->room_meta[lane] = (uint8_t)U.g_meta; }  // the room table, for the copy-out below (GM 2: already there)
        if (lane == src) {
            env_to_lane(E, U); need = false;
            // k_step's LDS monster cache of the requesting lane: filled straight from the generator's table (the alternative -- reloading the column
            // from the global table that is written just below -- is a memory round trip in the descent chain, the longest chain of a step)
            if (E.mc) for (int s = 0; s < nrooms; s++) E.mc[s * WAVE] = T->mon_w0[s];
        }
        pf.mark(19);
        __syncthreads();
        pf.mark(21);
        // tables: one room slot per lane (and round, with more rooms than lanes); grid: 16 bytes per lane
        for (int r = lane; r < nrooms; r += WAVE) {
            const size_t g = (size_t)r * real_n + real_e;
            st_pub<WT>(&S.room_rect[g], T->room_rect[r]); st_pub<WT>(&S.room_meta[g], T->room_meta[r]);
            st_pub<WT>(&S.mon_w0[g], T->mon_w0[r]); st_pub<WT>(&S.mon_hp[g], T->mon_hp[r]); st_pub<WT>(&S.mon_exp[g], T->mon_exp[r]);
            st_pub<WT>(&S.gold_pos[g], T->gold_pos[r]); st_pub<WT>(&S.gold_amt[g], T->gold_amt[r]);
            if constexpr (!WT) {  // (the live view's tables: the env's observation record follows them -- rg_state.h obs_rec; the player's word is the caller's)
                if (S.obs_rec) {
                    uint32_t *rec = S.obs_rec + (size_t)real_e * RG_OBS_REC_WORDS(nrooms);
                    rec[r] = T->mon_w0[r]; rec[nrooms + 1 + r] = T->room_rect[r]; reinterpret_cast<uint8_t *>(&rec[2 * nrooms + 1])[r] = T->room_meta[r];
                }
                if (S.ovl) S.ovl[(size_t)r * real_n + real_e] = (uint16_t)(((T->mon_w0[r] >> 24) & MF_ALIVE) ? ((T->mon_w0[r] & 0xffffu) | 0x40u) : 0xffffu);  // (| OVL_UNKNOWN)  // (the player's: the caller)
            }
            if (GM < 2) break;
        }
        {
            uint16_t *dst = S.cell + (size_t)real_e * HW;
            const uint16_t *srcp = reinterpret_cast<const uint16_t *>(slot);
            if ((HW & 7) == 0) {
                if constexpr (WT) {  // (8 bytes per store: the widest relaxed atomic store)
                    unsigned long long *d8 = reinterpret_cast<unsigned long long *>(dst);
                    const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(srcp);
                    for (int i = lane; i < HW / 4; i += WAVE) st_pub<true>(&d8[i], s8[i]);
                } else {
                    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
                    const uint4 *s4 = reinterpret_cast<const uint4 *>(srcp);
                    for (int i = lane; i < HW / 8; i += WAVE) d4[i] = s4[i];
                }
            } else
                for (int i = lane; i < HW; i += WAVE) st_pub<WT>(&dst[i], srcp[i]);
        }
        if (pf.p) pf.rec(20, __builtin_amdgcn_s_memtime() - tg0);
        pf.mark(16);
        __syncthreads();
        pf.mark(23);
    }
    (void)e;
}

// RunTime::player_status (core/src/lib.rs:345-356, player.rs:107-118) -> mirror
__device__ __forceinline__ void write_status(const RgState &S, const RgConfig &c, const Env &E) {
    int32_t *st = S.status + (size_t)E.e * 10;
    uint32_t hunger = c.hunger_time / 10;
    st[0] = (int32_t)E.dlevel; st[1] = (int32_t)E.gold; st[2] = E.hp; st[3] = E.hpmax; st[4] = 16; st[5] = 16; st[6] = 0;
    st[7] = E.plvl; st[8] = (int32_t)E.exp;
    st[9] = E.food <= hunger ? 2 : (E.food <= hunger * 2 ? 1 : 0);
}

__device__ __forceinline__ void load_env(const RgState &S, Env &E, int e) {
    int n = S.n;
    E.e = e; E.n = n;
    E.cell = E.gcell = S.cell + (size_t)e * S.hw;
    E.rd = {S.rng[0 * n + e], S.rng[1 * n + e], S.rng[2 * n + e], S.rng[3 * n + e]};
    E.ri = {S.rng[4 * n + e], S.rng[5 * n + e], S.rng[6 * n + e], S.rng[7 * n + e]};
    E.re = {S.rng[8 * n + e], S.rng[9 * n + e], S.rng[10 * n + e], S.rng[11 * n + e]};
    uint32_t p = S.p_pos[e];
    E.px = POS_X(p); E.py = POS_Y(p);
    E.hp = S.p_hp[e]; E.hpmax = S.p_hpmax[e]; E.plvl = S.p_lvl[e];
    E.exp = S.p_exp[e]; E.food = S.food[e]; E.quiet = S.quiet[e]; E.gold = S.pack_gold[e]; E.dlevel = S.dlevel[e];
    uint32_t mc = S.mon_cnt[e];
    E.mon_alive = mc & 0xff; E.mon_active = (mc >> 8) & 0xff;
}
template <bool WT = false>
__device__ __forceinline__ void store_env(const RgState &S, const Env &E) {
    int n = E.n, e = E.e;  // (E.n: the SoA stride of the view E.e indexes -- the spare view's is sp_slots * n)
    const uint32_t r[12] = {E.rd.x, E.rd.y, E.rd.z, E.rd.w, E.ri.x, E.ri.y, E.ri.z, E.ri.w, E.re.x, E.re.y, E.re.z, E.re.w};
#pragma unroll
    for (int k = 0; k < 12; k++) st_pub<WT>(&S.rng[k * n + e], r[k]);
    st_pub<WT>(&S.p_pos[e], (uint16_t)POS(E.px, E.py));
    st_pub<WT>(&S.p_hp[e], (int32_t)E.hp); st_pub<WT>(&S.p_hpmax[e], (int32_t)E.hpmax); st_pub<WT>(&S.p_lvl[e], (int32_t)E.plvl);
    st_pub<WT>(&S.p_exp[e], E.exp); st_pub<WT>(&S.food[e], E.food); st_pub<WT>(&S.quiet[e], E.quiet); st_pub<WT>(&S.pack_gold[e], E.gold); st_pub<WT>(&S.dlevel[e], E.dlevel);
    st_pub<WT>(&S.mon_cnt[e], E.mon_alive | (E.mon_active << 8));
}

// The stair set a producer launch writes for the k_step after it (rg_state.h): every env it owns gets its byte, marked envs are appended to the
// list with one atomic per wave.
__device__ __forceinline__ void stair_publish(const RgState &S, int lane, int e, bool mine, bool on) {
    const int w = (S.stair_gen + 1) & 1;
    if (mine) S.stair_mark[(size_t)w * S.n + e] = on ? 1 : 0;
    const uint64_t m = __ballot(mine && on);
    if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&S.stair_cnt[(S.stair_gen + 1) % 3], (uint32_t)__popcll(m));
        base = uni(base);
        if (mine && on) S.stair_list[(size_t)w * S.n + base + lanes_below(m)] = e;
    }
}
__device__ __forceinline__ void stair_recycle(const RgState &S) {  // one thread of the launch: the counter nobody reads or writes right now
    S.stair_cnt[(S.stair_gen + 2) % 3] = 0;
    S.stair_cnt[4 + (S.stair_gen + 2) % 3] = 0;  // ... and its take counter (k_step's stair blocks)
}

// Action-history log (RunTime::saved_inputs: react_to_input pushes every mapped key before it is processed, core/src/lib.rs:288; a rebuilt
// RunTime starts an empty log).  Two buffers per env: the running episode and the one before it, so the keys of an episode that ended in an
// auto-reset can still be dumped (rg_dump_history).
__device__ __forceinline__ void klog_new_episode(const RgState &S, int e) {
    if (!S.klog) return;
    const uint32_t cur = S.klog_cur[e] ^ 1u;
    S.klog_cur[e] = (uint8_t)cur;
    S.klog_len[cur * S.n + e] = 0;
}
__device__ __forceinline__ void klog_push(const RgState &S, int e, uint32_t key) {
    if (!S.klog) return;
    const uint32_t cur = S.klog_cur[e];
    const uint32_t len = S.klog_len[cur * S.n + e];
    if (len < (uint32_t)S.klog_cap) S.klog[((size_t)e * 2 + cur) * S.klog_cap + len] = (uint8_t)key;
    S.klog_len[cur * S.n + e] = len + 1;
}

// ---------------------------------------------------------------------------------------------
// k_build: (re)build every env from its seed
// ---------------------------------------------------------------------------------------------
extern __shared__ __align__(16) uint8_t g_smem[];

// BUILD_EPB envs per wave: a wave generates its levels one after the other, so fewer envs per wave = more waves per SIMD to overlap the
// generator's latencies (create / rg_reset only; not on the step path)
#define BUILD_EPB 16
template <int GM>  // (one kernel per generator instance: two in one kernel cost the capped k_regen 20 bytes of scratch)
__global__ void __launch_bounds__(WAVE) k_build(RgState S, RgConfig c) {
    const int lane = threadIdx.x;
    const int e = blockIdx.x * BUILD_EPB + lane;
    const bool valid = lane < BUILD_EPB && e < S.n;
    Env E;
    E.e = valid ? e : 0; E.n = S.n; E.cell = E.gcell = S.cell + (size_t)E.e * S.hw; E.err = 0; E.mc = nullptr;
    Prof pf; pf.start(S.prof);
    E.on_stairs = 0;
    gen_service<GM>(S, c, E, lane, e, valid, true, reinterpret_cast<uint16_t *>(g_smem), pf);
    pf.finish();
    if (blockIdx.x == 0 && lane == 0) stair_recycle(S);
    stair_publish(S, lane, e, valid, E.on_stairs != 0);
    if (!valid) return;
    store_env(S, E);
    write_status(S, c, E);
    if (S.obs_rec) { const int nr = c.room_num_x * c.room_num_y; S.obs_rec[(size_t)e * RG_OBS_REC_WORDS(nr) + nr] = POS(E.px, E.py); }  // (the rest of the record: gen_service)
    if (S.ovl) S.ovl[(size_t)(c.room_num_x * c.room_num_y) * S.n + e] = (uint16_t)(POS(E.px, E.py) | 0x40u);  // (| OVL_UNKNOWN)
    S.dc_len[e] = 0; S.dc_head[e] = 0; S.dc_part[e] = 0; S.dc_own[e] = 0;  // a rebuilt RunTime owns a fresh DistCache
    if (S.nx_state) (void)__hip_atomic_fetch_and(&S.nx_state[e], RG_NX_DROP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (as step_wave's new level)
    S.steps[e] = 0;
    S.flags[e] = RG_FLAG_REDRAW | RG_FLAG_HIST_DIRTY | E.err;
    if (E.err) atomicOr(S.err_any, E.err);
    S.reward[e] = 0.f;
    S.done[e] = 0;
    klog_new_episode(S, e);
}

// Parity / property-test hook (rg_debug_descend): every env takes Dungeon::new_level + actions::new_level's player placement as if it had
// pressed '>' on the stairs, without the turn around it -- the descent path of k_step (gen_service, is_build = false) on its own, so tests
// can look at levels 2..30 of thousands of seeds without walking there.
__device__ __forceinline__ void snapshot_walk_service(const RgState &S, const RgConfig &c, int lane, int e, bool want);
template <int GM>
__global__ void __launch_bounds__(WAVE) k_debug_descend(RgState S, RgConfig c) {
    const int lane = threadIdx.x;
    const int e = blockIdx.x * BUILD_EPB + lane;
    const bool valid = lane < BUILD_EPB && e < S.n;
    Env E;
    E.err = 0; E.mc = nullptr;
    load_env(S, E, valid ? e : 0);
    Prof pf; pf.start(nullptr);
    E.on_stairs = 0;
    if (S.dc_walk) snapshot_walk_service(S, c, lane, e, valid);  // (defined below) partial dist maps keep the level they were made on
    gen_service<GM>(S, c, E, lane, e, valid, false, reinterpret_cast<uint16_t *>(g_smem), pf);
    if (blockIdx.x == 0 && lane == 0) stair_recycle(S);
    stair_publish(S, lane, e, valid, E.on_stairs != 0);
    if (!valid) return;
    store_env(S, E);
    write_status(S, c, E);
    if (S.obs_rec) { const int nr = c.room_num_x * c.room_num_y; S.obs_rec[(size_t)e * RG_OBS_REC_WORDS(nr) + nr] = POS(E.px, E.py); }  // (the rest of the record: gen_service)
    if (S.ovl) S.ovl[(size_t)(c.room_num_x * c.room_num_y) * S.n + e] = (uint16_t)(POS(E.px, E.py) | 0x40u);  // (| OVL_UNKNOWN)
    if (S.nx_state) (void)__hip_atomic_fetch_and(&S.nx_state[e], RG_NX_DROP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (as step_wave's new level)
    S.flags[e] = (S.flags[e] & (RG_FLAG_TERMINAL | RG_FLAG_DEAD)) | RG_FLAG_REDRAW | RG_FLAG_HIST_DIRTY | E.err;
    if (E.err) atomicOr(S.err_any, E.err);
}

// ---------------------------------------------------------------------------------------------
// k_regen: background producer of the spare level-1 states (runs on a low-priority side stream)
// ---------------------------------------------------------------------------------------------
// Every auto-reset rebuilds the env from its (config, seed) -- a ~130-300 us single-lane dependency chain
// that would otherwise sit on k_step's critical path ~450 times per 65 536-env step.  The rebuild only
// depends on the seed, so it is produced AHEAD of time into a second ("spare") copy of the env state,
// one generation per consumed spare, overlapped with the following steps; k_step's reset then is a copy.
// SP is an RgState whose core pointers address the spare arrays.  Hand-off per env through sp_ready with
// agent-scope release/acquire (the consumer kernel runs concurrently on another stream).
#ifdef RG_REGEN_NOCAP
#define RG_REGEN_ATTR
#else
#define RG_REGEN_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // <= 128 registers: two generator waves beside a step wave on a SIMD (tests/test_kernel_resources.py)
#endif
template <int GM>
__device__ __forceinline__ void regen_body(const RgState &SP, const RgConfig &c, int epb, int max_claims, int spares) {
    const int lane = threadIdx.x;
    const int e = blockIdx.x * epb + lane;
    const bool valid = lane < epb && e < SP.n;
    // ONE spare per wave and launch: a wave generates its claims one after the other (28 us each, more beside a step wave), so the few waves that found
    // two or three consumed spares decided how long the launch lasted -- 120-160 us, i.e. through k_step AND the observation pass behind it and into the
    // next step (round 4: k_obs 47.5 -> 52 us with a launch beside every step).  The other consumed spares of the wave's eight envs wait for the next
    // launch, one step later; a spare is wanted an episode after it was consumed.
    // (spares == 0: the consumed spares are rebuilt by the level-per-lane producer, rg_regen_lanes.hip; this launch serves the next-level structures only)
    // spares == 1: the wave-per-level producer of the one-slot layout (ROGUE_GYM_HIP_WAVE_REGEN, > 32 rooms).  spares == 2: the consumed spares are rebuilt
    // 64 levels per wave by rg_regen_lanes.hip, a launch every 16 steps (a round takes 300-450 us whatever its size: a launch beside every few steps left one
    // running behind every short window of steps); what this launch adds is the URGENT case -- an env that is down to its last ready spare (fixed-seed envs
    // that die within a dozen steps do so episode after episode) gets one built here, beside the next step.
    int es = e;  // the spare-view entry to build
    bool want = false;
    if (valid && spares == 1) want = __hip_atomic_load(&SP.sp_ready[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    else if (valid && spares >= 2) {  // (spares == 3, development: only an env with NO ready spare left)
        int live_slots = 0, free_slot = -1;
        for (int sl = SP.sp_slots - 1; sl >= 0; sl--) {
            const uint32_t st = __hip_atomic_load(&SP.sp_ready[(size_t)sl * SP.n + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st == 0u) free_slot = sl; else live_slots++;  // (ready, or claimed by a producer)
        }
        want = live_slots <= (spares == 3 ? 0 : 1) && free_slot >= 0;
        if (want) es = free_slot * SP.n + e;
    }
    // ... or ONE next-level structure (gen_service), which goes first: it is wanted within two or three steps, a spare an episode later
    const bool want_nx = GM < 2 && valid && SP.nx_state && __hip_atomic_load(&SP.nx_state[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == RG_NX_ASKED;
    const uint64_t wm = __ballot(want), wx = __ballot(want_nx);
    if (!(wm | wx)) return;
    bool claim = false, claim_nx = false;
    if (max_claims <= 1) {
        if (wx) { if (lane == __ffsll((long long)wx) - 1) claim_nx = atomicCAS(&SP.nx_state[e], RG_NX_ASKED, RG_NX_CLAIMED) == RG_NX_ASKED; }
        else if (lane == __ffsll((long long)wm) - 1) claim = atomicCAS(&SP.sp_ready[es], 0u, 2u) == 0u;
    } else {
        if (want_nx) claim_nx = atomicCAS(&SP.nx_state[e], RG_NX_ASKED, RG_NX_CLAIMED) == RG_NX_ASKED;
        if (want && !claim_nx) claim = atomicCAS(&SP.sp_ready[es], 0u, 2u) == 0u;  // (one kind per lane and launch: the lane's registers carry one request)
    }
    if (!__any(claim || claim_nx)) return;
    Env E;
    // a spare: entry `es` of the spare view, whose SoA stride is sp_slots * n; a next-level request: the env's own index, stride n
    E.e = valid ? (claim ? es : e) : 0; E.n = claim ? SP.n * SP.sp_slots : SP.n; E.cell = E.gcell = SP.cell + (size_t)E.e * SP.hw; E.err = 0; E.mc = nullptr;
    Prof pf; pf.start(nullptr);
    E.on_stairs = 0;
    const uint64_t cx = __ballot(claim_nx);
    E.rd = E.ri = E.re = {0, 0, 0, 0};
    E.px = E.py = E.hp = E.hpmax = E.plvl = 0; E.exp = E.food = E.quiet = E.gold = E.dlevel = E.mon_alive = E.mon_active = 0;
    if (claim_nx) {
        // the request: the streams and the level the asking k_step wrote with it (coherent loads: the words may have been written during this launch)
        const uint32_t *q = SP.nx_rng + e;
        const size_t n = (size_t)SP.n;
        E.rd = {__hip_atomic_load(&q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&q[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&q[2 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&q[3 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        E.ri = {__hip_atomic_load(&q[8 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&q[9 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                __hip_atomic_load(&q[10 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&q[11 * n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        E.dlevel = __hip_atomic_load(&SP.nx->level[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // everything k_step will take over is written THROUGH (st_pub<true>): the hand-off is "sc1 payload, every writing wave drained, then the flag" --
    // no release fence (see st_pub); the consumer's side is an agent-scope acquire (take_spares, step_wave's descent)
    // (ONE call site for both kinds: a second instance of the generator cost the capped kernel scratch memory)
    gen_service<GM, true>(SP, c, E, lane, E.e, claim || claim_nx, true, reinterpret_cast<uint16_t *>(g_smem), pf, 0ull, cx);
    if (claim) { store_env<true>(SP, E); st_pub<true>(&SP.on_stairs[es], (uint8_t)E.on_stairs); }
    if ((claim || claim_nx) && E.err) atomicOr(SP.err_any, E.err);  // (the flag word belongs to the concurrently running k_step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (claim) __hip_atomic_store(&SP.sp_ready[es], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (by CAS: an env that got a new level meanwhile turned the CLAIMED into DROP -- the structure is for a level it has left, and the env may ask again
    // from here on)
    // (a structure whose generation raised a capacity guard is never published: the descent generates inline and raises the guard on the env's own flags)
    if (claim_nx && (E.err || atomicCAS(&SP.nx_state[e], RG_NX_CLAIMED, RG_NX_READY) != RG_NX_CLAIMED))
        __hip_atomic_store(&SP.nx_state[e], RG_NX_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The gate in front of a generator launch, alone on the generator's stream: ONE wave that waits until the k_step launched beside it has started
// (launch_mark reached `target`).  The generator follows in stream order, so it runs beside that k_step -- not in front of it, where its waves would take
// the slots of k_step's blocks (and, the host running hundreds of steps ahead of the GPU, long before the spares it is to refill are consumed), and with
// no event on the handle's stream.  The wait is bounded by the wall clock (s_memrealtime, 100 MHz): one second -- a k_step that never starts must not
// hang rg_sync / rg_destroy, which drain this stream; after a time-out the generator merely runs early.
__global__ void __launch_bounds__(WAVE) k_regen_gate(const uint32_t *__restrict__ mark, uint32_t target, uint32_t *__restrict__ err_any) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;
    }
    (void)err_any;
}
template <int GM> __global__ void __launch_bounds__(WAVE) RG_REGEN_ATTR k_regen(RgState SP, RgConfig c, int epb, int max_claims, int spares) { regen_body<GM>(SP, c, epb, max_claims, spares); }
// (the 384-room instance does not fit the 128-register cap without scratch; its configs run at one step wave per SIMD anyway)
__global__ void __launch_bounds__(WAVE) k_regen_huge(RgState SP, RgConfig c, int epb, int max_claims, int spares) { regen_body<2>(SP, c, epb, max_claims, spares); }

// ---------------------------------------------------------------------------------------------
// wave-cooperative bit-parallel BFS (Floor::make_dist_map, floor.rs:395-416)
// ---------------------------------------------------------------------------------------------
// The reference runs a FIFO BFS over 8 directions; the result is the exact unit-weight shortest
// distance, so any level-synchronous formulation gives identical maps.  Here lane y of the wave owns
// grid row y as bitmasks held in registers (WW 64-bit words per row: walkable, visited, frontier).  One
// BFS level = two cross-lane shuffles (the frontier of rows y-1 / y+1) plus a handful of shifts/ANDs;
// the diagonal rule of can_move_impl (both orthogonal neighbours walkable, floor.rs:177-180) is
// expressed on the masks.  No LDS traffic or barrier inside the level loop; distances are staged in LDS
// and streamed to the env's DistCache slot with 16-byte stores.
// Up to G = 64 / pow2ceil(H) requests are served at once: the wave is split into G groups of rows
// (mini: 4 groups of 16 lanes), and the neighbour rows come from DPP whole-wave shifts (1 VALU op each).
#ifndef RG_BFS_NO_ROW16
#define RG_BFS_NO_ROW16 0  // (A/B aid: 1 = whole-wave shifts also for H <= 16)
#endif
template <int WW> struct RowBits { uint64_t w[WW]; };

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }  // lane i <- lane i-1
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }  // lane i <- lane i+1

// ... the same inside a DPP row of 16 lanes, zeros shifted in at the row's ends (bound_ctrl): with H <= 16 a BFS group IS a DPP row, and the first / last grid
// row's "no neighbour" comes for free -- no select behind the move, and the move folds into the instruction that consumes it
__device__ __forceinline__ uint32_t row_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }  // lane i <- lane i-1 of its row, lane 0 <- 0
__device__ __forceinline__ uint32_t row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }  // lane i <- lane i+1 of its row, lane 15 <- 0

template <int WW> __device__ __forceinline__ RowBits<WW> rb_shl1(const RowBits<WW> &a) {  // cell x-1 -> x
    RowBits<WW> r;
#pragma unroll
    for (int k = 0; k < WW; k++) r.w[k] = (a.w[k] << 1) | (k > 0 ? a.w[k - 1] >> 63 : 0ull);
    return r;
}
template <int WW> __device__ __forceinline__ RowBits<WW> rb_shr1(const RowBits<WW> &a) {  // cell x+1 -> x
    RowBits<WW> r;
#pragma unroll
    for (int k = 0; k < WW; k++) r.w[k] = (a.w[k] >> 1) | (k + 1 < WW ? a.w[k + 1] << 63 : 0ull);
    return r;
}
// the row above (UP = true: lane-1) or below (lane+1) of the same group; `ok` masks group boundaries.
// NARROW: rows fit 32 bits, so only the low halves travel.
template <int WW, bool NARROW, bool UP> __device__ __forceinline__ RowBits<WW> rb_neighbour(const RowBits<WW> &a, bool ok) {
    RowBits<WW> r;
#pragma unroll
    for (int k = 0; k < WW; k++) {
        uint32_t lo = UP ? wave_shr1((uint32_t)a.w[k]) : wave_shl1((uint32_t)a.w[k]);
        uint32_t hi = 0;
        if (!NARROW) hi = UP ? wave_shr1((uint32_t)(a.w[k] >> 32)) : wave_shl1((uint32_t)(a.w[k] >> 32));
        r.w[k] = ok ? (((uint64_t)hi << 32) | lo) : 0ull;
    }
    return r;
}

// one round: group g of the wave computes the dist map of request (env, tx, ty, slot) given per lane (uniform per group).
// Distances are kept as bit-planes per row (plane b = the cells whose distance has bit b set): writing a u16 per newly reached cell inside
// the level loop (a per-lane loop over the new bits) cost several times the level step itself.  Levels are taken 8 at a time: the three low
// planes are updated with compile-time knowledge of the level's low bits, the ten high planes once per block with the OR of the block's
// new cells (distances < 8192 >= every cell of the largest grid).  At the end every lane expands its row and stores it straight into the env's
// DistCache slot.
template <int WW, bool NARROW>
__device__ __forceinline__ void bfs_rows(const RgState &S, const RgConfig &c, uint64_t *lds_hi /* [10][WW][64] high planes */, bool active, int env, int tx, int ty,
                                         int slot, int row) {
    const int W = c.width, H = c.height, HW = W * H, lane = threadIdx.x;
    const uint16_t *cell = S.cell + (size_t)env * HW;
    const bool row_ok = active && row < H;
    RowBits<WW> wk, vis, fr, inject, p0, p1, p2;  // the ten high planes live in LDS (60 registers for the widest grid otherwise: the step kernel
                                                  // must stay small enough for a k_regen wave to share its SIMD)
#pragma unroll
    for (int k = 0; k < WW; k++) wk.w[k] = vis.w[k] = fr.w[k] = inject.w[k] = p0.w[k] = p1.w[k] = p2.w[k] = 0ull;
    if (row_ok) {  // walkable mask of my row (Surface::can_walk, rogue/mod.rs:175-182)
        const uint16_t *rowp = cell + row * W;
        if ((W & 7) == 0) {
            const uint4 *r4 = reinterpret_cast<const uint4 *>(rowp);
            for (int j = 0; j < W / 8; j++) {
                uint4 v = r4[j];
                uint32_t q[4] = {v.x, v.y, v.z, v.w};
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    bits |= (uint32_t)can_walk(q[t] & 0xffff) << (2 * t);
                    bits |= (uint32_t)can_walk(q[t] >> 16) << (2 * t + 1);
                }
                int x = j * 8;
#pragma unroll
                for (int k = 0; k < WW; k++)
                    if ((x >> 6) == k) wk.w[k] |= (uint64_t)bits << (x & 63);
            }
        } else {
            for (int x = 0; x < W; x++) {
                uint64_t bb = (uint64_t)can_walk(rowp[x]) << (x & 63);
#pragma unroll
                for (int k = 0; k < WW; k++)
                    if ((x >> 6) == k) wk.w[k] |= bb;
            }
        }
    }
    const bool up_ok = row > 0, dn_ok = row + 1 < H;
    const RowBits<WW> wu = rb_neighbour<WW, NARROW, true>(wk, up_ok);    // walkable mask of row y-1
    const RowBits<WW> wd = rb_neighbour<WW, NARROW, false>(wk, dn_ok);   // and of row y+1
    if (row_ok && row == ty) {  // level 0: the target cell itself, walkable or not
#pragma unroll
        for (int k = 0; k < WW; k++)
            if ((tx >> 6) == k) inject.w[k] = 1ull << (tx & 63);
    }
    uint32_t blk = 0;
    for (;; blk++) {  // levels 8 * blk .. 8 * blk + 7
        RowBits<WW> acc;
#pragma unroll
        for (int k = 0; k < WW; k++) acc.w[k] = 0ull;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const RowBits<WW> fu = rb_neighbour<WW, NARROW, true>(fr, up_ok);
            const RowBits<WW> fd = rb_neighbour<WW, NARROW, false>(fr, dn_ok);
            RowBits<WW> au, ad;  // neighbour-row frontier restricted to cells whose vertical step lands on a walkable cell of my row
#pragma unroll
            for (int k = 0; k < WW; k++) { au.w[k] = fu.w[k] & wk.w[k]; ad.w[k] = fd.w[k] & wk.w[k]; }
            const RowBits<WW> sl = rb_shl1<WW>(fr), sr = rb_shr1<WW>(fr);
            const RowBits<WW> aul = rb_shl1<WW>(au), aur = rb_shr1<WW>(au), adl = rb_shl1<WW>(ad), adr = rb_shr1<WW>(ad);
#pragma unroll
            for (int k = 0; k < WW; k++) {
                // Left/Right | Down/Up | diagonals: source (x-+1, y-+1) needs walk(x, y-+1) and walk(x-+1, y)
                const uint64_t tgt = sl.w[k] | sr.w[k] | fu.w[k] | fd.w[k] | ((aul.w[k] | aur.w[k]) & wu.w[k]) | ((adl.w[k] | adr.w[k]) & wd.w[k]);
                const uint64_t nw = (tgt & wk.w[k] & ~vis.w[k]) | inject.w[k];
                inject.w[k] = 0ull;
                vis.w[k] |= nw;
                fr.w[k] = nw;
                if (j & 1) p0.w[k] |= nw;
                if (j & 2) p1.w[k] |= nw;
                if (j & 4) p2.w[k] |= nw;
                acc.w[k] |= nw;
            }
        }
#pragma unroll
        for (int b = 0; b < 10; b++)
            if ((blk >> b) & 1u) {  // wave-uniform condition; the first block with bit b set (blk == 1 << b) initialises the plane
#pragma unroll
                for (int k = 0; k < WW; k++) {
                    uint64_t *q = &lds_hi[(b * WW + k) * WAVE + lane];
                    *q = blk == (1u << b) ? acc.w[k] : (*q | acc.w[k]);
                }
            }
        bool any = false;
#pragma unroll
        for (int k = 0; k < WW; k++) any = any || fr.w[k] != 0ull;
        if (!__any(any) || blk == 1023u) break;
    }
    if (row_ok) {  // expand my row: cell x -> u16 distance, 0xFFFF where the cell was never reached
        const int nhi = 32 - __clz((int)blk);  // high planes in use (wave-uniform)
        uint16_t *out = S.dc_map + ((size_t)env * RG_DIST_SLOTS + slot) * HW + row * W;
        const bool vec = (W & 7) == 0;
#pragma unroll
        for (int k = 0; k < WW; k++) {
            const int xw = k * 64;
            if (xw >= W) break;
            uint64_t hp[10];  // this word's high planes back from LDS, once
#pragma unroll
            for (int b = 0; b < 10; b++) hp[b] = b < nhi ? lds_hi[(b * WW + k) * WAVE + lane] : 0ull;
            for (int g8 = 0; g8 < 8 && xw + g8 * 8 < W; g8++) {  // 8 cells = one 16-byte store
                uint32_t d[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int sh = g8 * 8 + 2 * q;
                    const uint32_t a0 = (uint32_t)(p0.w[k] >> sh), a1 = (uint32_t)(p1.w[k] >> sh), a2 = (uint32_t)(p2.w[k] >> sh);
                    uint32_t lo = (a0 & 1u) | ((a1 & 1u) << 1) | ((a2 & 1u) << 2);
                    uint32_t hi = ((a0 >> 1) & 1u) | (((a1 >> 1) & 1u) << 1) | (((a2 >> 1) & 1u) << 2);
#pragma unroll
                    for (int b = 0; b < 10; b++)
                        if (b < nhi) {
                            const uint32_t ab = (uint32_t)(hp[b] >> sh);
                            lo |= (ab & 1u) << (3 + b);
                            hi |= ((ab >> 1) & 1u) << (3 + b);
                        }
                    const uint32_t un = ~(uint32_t)(vis.w[k] >> sh);
                    if (un & 1u) lo = 0xFFFFu;
                    if (un & 2u) hi = 0xFFFFu;
                    d[q] = lo | (hi << 16);
                }
                const int x = xw + g8 * 8;
                if (vec) *reinterpret_cast<uint4 *>(out + x) = make_uint4(d[0], d[1], d[2], d[3]);
                else
                    for (int t = 0; t < 8 && x + t < W; t++) out[x + t] = (uint16_t)(d[t >> 1] >> ((t & 1) * 16));
            }
        }
    }
    __syncthreads();  // the requesting lanes read their maps right after (monsters_move): the stores of the other lanes must have landed
}

// W == 32 (the narrowest screen, the mini dungeon): a row is one 32-bit mask and the level step is ~25 VALU ops + 2 DPP moves.  Distances
// are not written per newly reached cell (a per-lane loop over the new bits cost more than the level step itself): each row keeps the
// distance of its 32 cells as BIT-PLANES (plane b = cells whose distance has bit b set).  Levels are taken 8 at a time, so the three low
// planes are updated with compile-time knowledge of the level's low bits and the eight high planes once per block with the OR of the
// block's new cells (distances < 2048 = every cell of a 32 x 64 grid).  At the end each lane expands its row to 32 u16 and stores its
// 64 bytes straight into the env's DistCache slot -- no LDS staging.
// ROW16: H <= 16, a group of rows is one DPP row (row_shr1 / row_shl1 above); else whole-wave shifts and a select per level.
template <bool ROW16>
__device__ __forceinline__ void bfs_rows_w32(const RgState &S, const RgConfig &c, bool active, int env, int tx, int ty, int slot, int row) {
    const int W = 32, H = c.height, HW = W * H;
    const uint16_t *cell = S.cell + (size_t)env * HW;
    const bool row_ok = active && row < H;
    uint32_t wk = 0;
    if (row_ok) {  // walkable mask of my row (Surface::can_walk, rogue/mod.rs:175-182)
        const uint4 *r4 = reinterpret_cast<const uint4 *>(cell + row * W);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint4 v = r4[j];
            uint32_t q[4] = {v.x, v.y, v.z, v.w};
            uint32_t bits = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                bits |= (uint32_t)can_walk(q[t] & 0xffff) << (2 * t);
                bits |= (uint32_t)can_walk(q[t] >> 16) << (2 * t + 1);
            }
            wk |= bits << (j * 8);
        }
    }
    const bool up_ok = row > 0, dn_ok = row + 1 < H;
    const uint32_t wu = ROW16 ? row_shr1(wk) : (up_ok ? wave_shr1(wk) : 0u), wd = ROW16 ? row_shl1(wk) : (dn_ok ? wave_shl1(wk) : 0u);  // (rows >= H hold no walkable cell)
    uint32_t vis = 0, fr = 0;
    uint32_t inject = (row_ok && row == ty) ? 1u << tx : 0u;  // level 0: the target cell itself, walkable or not
    uint32_t p0 = 0, p1 = 0, p2 = 0, ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t blk = 0;
    for (;; blk++) {  // levels 8 * blk .. 8 * blk + 7
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t fu, fd;
            if (ROW16) { fu = row_shr1(fr); fd = row_shl1(fr); }
            else { fu = wave_shr1(fr); fd = wave_shl1(fr); fu = up_ok ? fu : 0u; fd = dn_ok ? fd : 0u; }
            const uint32_t au = fu & wk, ad = fd & wk;
            // Left/Right | Down/Up | diagonals: source (x-+1, y-+1) needs walk(x, y-+1) and walk(x-+1, y)
            const uint32_t tgt = (fr << 1) | (fr >> 1) | fu | fd | (((au << 1) | (au >> 1)) & wu) | (((ad << 1) | (ad >> 1)) & wd);
            const uint32_t nw = (tgt & wk & ~vis) | inject;
            inject = 0;
            vis |= nw;
            fr = nw;
            if (j & 1) p0 |= nw;
            if (j & 2) p1 |= nw;
            if (j & 4) p2 |= nw;
            acc |= nw;
        }
#pragma unroll
        for (int b = 0; b < 8; b++)
            if ((blk >> b) & 1u) ph[b] |= acc;  // wave-uniform condition
        if (!__any(fr != 0) || blk == 255u) break;
    }
    if (row_ok) {  // expand my row: cell x -> u16 distance, 0xFFFF where the cell was never reached
        const int nhi = 32 - __clz((int)blk);  // high planes in use (wave-uniform)
        const uint32_t unv = ~vis;
        uint32_t out[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t lo = 0, hi = 0;
            const int x = 2 * j;
            lo |= ((p0 >> x) & 1u) | (((p1 >> x) & 1u) << 1) | (((p2 >> x) & 1u) << 2);
            hi |= ((p0 >> (x + 1)) & 1u) | (((p1 >> (x + 1)) & 1u) << 1) | (((p2 >> (x + 1)) & 1u) << 2);
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (b < nhi) {
                    lo |= ((ph[b] >> x) & 1u) << (3 + b);
                    hi |= ((ph[b] >> (x + 1)) & 1u) << (3 + b);
                }
            if ((unv >> x) & 1u) lo = 0xFFFFu;
            if ((unv >> (x + 1)) & 1u) hi = 0xFFFFu;
            out[j] = lo | (hi << 16);
        }
        uint4 *o4 = reinterpret_cast<uint4 *>(S.dc_map + ((size_t)env * RG_DIST_SLOTS + slot) * HW + row * W);
#pragma unroll
        for (int j = 0; j < 4; j++) o4[j] = make_uint4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
    }
    __syncthreads();  // the requesting lanes read their maps right after (monsters_move): the stores of the other lanes must have landed
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {  // sum over the 64 lanes (values are tiny: per-lane event counts)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}
// 32 < W <= 96 with H * W <= 4096 (the reference's default 80 x 24 among them): the row is WN = 2 or 3 32-bit words, everything in registers.
// The 64-bit-word form below spends two VALU ops on every logic op and three on every shift; here a shift across the word boundary is one
// v_alignbit_b32 and a level step is ~26 ops per word.  Distances are < the number of walkable cells < 4096 = 3 low + 9 high bit-planes.
// walkable mask of one grid row as WN 32-bit words (Surface::can_walk, rogue/mod.rs:175-182)
template <int WN>
__device__ __forceinline__ void row_walk_mask(const uint16_t *rowp, int W, uint32_t (&wk)[WN]) {
#pragma unroll
    for (int k = 0; k < WN; k++) wk[k] = 0u;
    if ((W & 7) == 0) {
        const uint4 *r4 = reinterpret_cast<const uint4 *>(rowp);
#pragma unroll
        for (int j = 0; j < WN * 4; j++) {  // (fully unrolled: compile-time word indices; the guard is the only run-time part)
            if (j * 8 < W) {
                const uint4 v = r4[j];
                const uint32_t q[4] = {v.x, v.y, v.z, v.w};
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    bits |= (uint32_t)can_walk(q[t] & 0xffff) << (2 * t);
                    bits |= (uint32_t)can_walk(q[t] >> 16) << (2 * t + 1);
                }
                wk[j >> 2] |= bits << ((j & 3) * 8);
            }
        }
    } else {
        for (int x = 0; x < W; x++) {
            const uint32_t bb = (uint32_t)can_walk(rowp[x]) << (x & 31);
#pragma unroll
            for (int k = 0; k < WN; k++)
                if ((x >> 5) == k) wk[k] |= bb;
        }
    }
}

// 32 < W <= 96 with H * W <= 4096 (the reference's default 80 x 24 among them): the row is WN = 2 or 3 32-bit words, everything in registers.
// The 64-bit-word form below spends two VALU ops on every logic op and three on every shift; here a shift across the word boundary is one
// v_alignbit_b32 and a level step is ~26 ops per word.  Distances are < the number of walkable cells < 4096 = 3 low + 9 high bit-planes.
//
// PARTIAL MAPS.  A full map of the 80 x 24 dungeon is 100-150 levels (~25 us of a wave that has nothing else to run); what this turn reads of it
// is the 3 x 3 around every chasing monster, typically a dozen levels out.  The expansion therefore stops at the end of the first 8-level block
// in which every chaser's own cell (`mcol`: the requesting lane's column of the LDS monster cache, PENDING and not RANDOM) has been reached --
// or when the frontier dies (then the map is COMPLETE, the return value).  A partial map is exact wherever it has a value, and 0xFFFF elsewhere;
// monsters_move's choice is the minimum over the 3 x 3 of a monster whose own cell has a value, so every cell that could win has one too (a
// neighbour without one is farther than the own cell).  The DistCache keeps such a map under the reference's key like any other (stale maps
// are the reference's behaviour, rogue/mod.rs:504-517): when a later turn needs more of it (dist_map_sufficient), the SAME expansion is run
// again, farther -- over the walkable mask the map was made from.  That mask is the level's own until something changes it (a descent,
// a search that opens a hidden passage or a locked door: floor.rs:349-370); right before such a change snapshot_walk_service saves it into the
// slot (`own`), so the map the reference would have computed in full back then is what the extension continues.
template <int WN>
__device__ __forceinline__ bool bfs_rows_n32(const RgState &S, const RgConfig &c, bool active, int env, int tx, int ty, int slot, int row, uint64_t grp_lanes,
                                             const lds_u32 *mcol, bool own) {
    const int W = c.width, H = c.height, HW = W * H;
    const uint16_t *cell = S.cell + (size_t)env * HW;
    const bool row_ok = active && row < H;
    uint32_t wk[WN], wu[WN], wd[WN], vis[WN], fr[WN], inject[WN], p0[WN], p1[WN], p2[WN], ph[9][WN], want[WN];
#pragma unroll
    for (int k = 0; k < WN; k++) {
        wk[k] = vis[k] = fr[k] = inject[k] = p0[k] = p1[k] = p2[k] = want[k] = 0u;
#pragma unroll
        for (int b = 0; b < 9; b++) ph[b][k] = 0u;
    }
    if (row_ok) {
        if (own) {
            const uint32_t *wp = S.dc_walk + (((size_t)env * RG_DIST_SLOTS + slot) * H + row) * WN;
#pragma unroll
            for (int k = 0; k < WN; k++) wk[k] = wp[k];
        } else row_walk_mask<WN>(cell + row * W, W, wk);
        const int nrooms = c.room_num_x * c.room_num_y;
        for (int q = 0; q < nrooms; q++) {  // the chasers standing in my row
            const uint32_t w = mcol[q * WAVE];
            if (((w >> 24) & (MF_PENDING | MF_RANDOM)) == MF_PENDING && POS_Y(w) == row) {
#pragma unroll
                for (int k = 0; k < WN; k++)
                    if ((POS_X(w) >> 5) == k) want[k] |= 1u << (POS_X(w) & 31);
            }
        }
    }
    const bool up_ok = row > 0, dn_ok = row + 1 < H;
#pragma unroll
    for (int k = 0; k < WN; k++) {
        wu[k] = up_ok ? wave_shr1(wk[k]) : 0u;   // walkable mask of row y-1
        wd[k] = dn_ok ? wave_shl1(wk[k]) : 0u;   // and of row y+1
        if (row_ok && row == ty && (tx >> 5) == k) inject[k] = 1u << (tx & 31);  // level 0: the target cell itself, walkable or not
    }
    uint32_t blk = 0;
    uint64_t alive = 0;
    for (;; blk++) {  // levels 8 * blk .. 8 * blk + 7
        uint32_t acc[WN];
#pragma unroll
        for (int k = 0; k < WN; k++) acc[k] = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t fu[WN], fd[WN], au[WN], ad[WN], nw[WN];
#pragma unroll
            for (int k = 0; k < WN; k++) {
                fu[k] = up_ok ? wave_shr1(fr[k]) : 0u;
                fd[k] = dn_ok ? wave_shl1(fr[k]) : 0u;
                au[k] = fu[k] & wk[k]; ad[k] = fd[k] & wk[k];  // neighbour-row frontier whose vertical step lands on a walkable cell of my row
            }
#pragma unroll
            for (int k = 0; k < WN; k++) {
                // cell x-1 -> x (shift left, carry in from the word below) and x+1 -> x (shift right, carry in from the word above)
                const uint32_t lo_fr = k > 0 ? fr[k - 1] : 0u, hi_fr = k + 1 < WN ? fr[k + 1] : 0u;
                const uint32_t lo_au = k > 0 ? au[k - 1] : 0u, hi_au = k + 1 < WN ? au[k + 1] : 0u;
                const uint32_t lo_ad = k > 0 ? ad[k - 1] : 0u, hi_ad = k + 1 < WN ? ad[k + 1] : 0u;
                const uint32_t sl = __builtin_amdgcn_alignbit(fr[k], lo_fr, 31), sr = __builtin_amdgcn_alignbit(hi_fr, fr[k], 1);
                const uint32_t aul = __builtin_amdgcn_alignbit(au[k], lo_au, 31), aur = __builtin_amdgcn_alignbit(hi_au, au[k], 1);
                const uint32_t adl = __builtin_amdgcn_alignbit(ad[k], lo_ad, 31), adr = __builtin_amdgcn_alignbit(hi_ad, ad[k], 1);
                // Left/Right | Down/Up | diagonals: source (x-+1, y-+1) needs walk(x, y-+1) and walk(x-+1, y)
                const uint32_t tgt = sl | sr | fu[k] | fd[k] | ((aul | aur) & wu[k]) | ((adl | adr) & wd[k]);
                nw[k] = (tgt & wk[k] & ~vis[k]) | inject[k];
            }
#pragma unroll
            for (int k = 0; k < WN; k++) {
                inject[k] = 0u;
                vis[k] |= nw[k];
                fr[k] = nw[k];
                if (j & 1) p0[k] |= nw[k];
                if (j & 2) p1[k] |= nw[k];
                if (j & 4) p2[k] |= nw[k];
                acc[k] |= nw[k];
            }
        }
#pragma unroll
        for (int b = 0; b < 9; b++)
            if ((blk >> b) & 1u) {  // wave-uniform condition
#pragma unroll
                for (int k = 0; k < WN; k++) ph[b][k] |= acc[k];
            }
        bool any = false;
#pragma unroll
        for (int k = 0; k < WN; k++) any = any || fr[k] != 0u;
        bool miss = false;
#pragma unroll
        for (int k = 0; k < WN; k++) miss = miss || (want[k] & ~vis[k]) != 0u;
        alive = __ballot(any) & grp_lanes;
        // my request goes on while it still has a frontier and a chaser it has not reached; the wave goes on while any request does
        const bool go_on = alive != 0 && (S.full_bfs || (__ballot(miss) & grp_lanes) != 0);
        if (!__any(go_on) || blk == 511u) break;
    }
    if (row_ok) {  // expand my row: cell x -> u16 distance, 0xFFFF where the cell was never reached
        const int nhi = 32 - __clz((int)blk);  // high planes in use (wave-uniform)
        uint16_t *out = S.dc_map + ((size_t)env * RG_DIST_SLOTS + slot) * HW + row * W;
        const bool vec = (W & 7) == 0;
#pragma unroll
        for (int k = 0; k < WN; k++) {
            const uint32_t unv = ~vis[k];
#pragma unroll
            for (int g8 = 0; g8 < 4; g8++) {  // 8 cells = one 16-byte store
                const int x0 = k * 32 + g8 * 8;
                if (x0 >= W) continue;
                uint32_t d[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int x = g8 * 8 + 2 * q;
                    uint32_t lo = ((p0[k] >> x) & 1u) | (((p1[k] >> x) & 1u) << 1) | (((p2[k] >> x) & 1u) << 2);
                    uint32_t hi = ((p0[k] >> (x + 1)) & 1u) | (((p1[k] >> (x + 1)) & 1u) << 1) | (((p2[k] >> (x + 1)) & 1u) << 2);
#pragma unroll
                    for (int b = 0; b < 9; b++)
                        if (b < nhi) {
                            lo |= ((ph[b][k] >> x) & 1u) << (3 + b);
                            hi |= ((ph[b][k] >> (x + 1)) & 1u) << (3 + b);
                        }
                    if ((unv >> x) & 1u) lo = 0xFFFFu;
                    if ((unv >> (x + 1)) & 1u) hi = 0xFFFFu;
                    d[q] = lo | (hi << 16);
                }
                if (vec) *reinterpret_cast<uint4 *>(out + x0) = make_uint4(d[0], d[1], d[2], d[3]);
                else
                    for (int t = 0; t < 8 && x0 + t < W; t++) out[x0 + t] = (uint16_t)(d[t >> 1] >> ((t & 1) * 16));
            }
        }
    }
    __syncthreads();  // the requesting lanes read their maps right after (monsters_move): the stores of the other lanes must have landed
    return alive == 0;  // the frontier died: every reachable cell has its distance
}

// serve every lane of `need` (ballot mask): G requests per round.  BW = width class of the grid (0: W = 32, else 64-bit words per row): the
// step kernel is instantiated per class, so a narrow grid does not pay the register footprint of the wide-row BFS (above 384 registers a
// k_regen wave no longer fits beside a step wave on the SIMD).
// Returns, to every requesting lane, whether its map is COMPLETE (always, except for the partial maps of bfs_rows_n32).  own_req: the lanes whose
// request continues a map over the walkable mask saved in its slot.
template <int BW>
__device__ __forceinline__ bool bfs_service(const RgState &S, const RgConfig &c, uint64_t *lds, uint64_t need, int e, int px, int py, int map_slot, int lane, const lds_u32 *mcbase,
                                            uint64_t own_req) {
    const int H = c.height;
    const int rows_pow2 = H <= 16 ? 16 : (H <= 32 ? 32 : 64);
    const int G = WAVE / rows_pow2;
    const int grp = lane / rows_pow2, row = lane - grp * rows_pow2;
    const uint64_t grp_lanes = (rows_pow2 == 64 ? ~0ull : ((1ull << rows_pow2) - 1ull)) << (grp * rows_pow2);
    bool my_complete = true;
    while (need) {
        int src = -1, served_by = -1;
        for (int g = 0; g < G; g++) {
            int sg = need ? __ffsll((long long)need) - 1 : -1;
            if (need) need &= need - 1;
            if (g == grp) src = sg;
            if (sg == lane) served_by = g;
        }
        const bool active = src >= 0;
        const int s = active ? src : 0;
        int env_s = __shfl(e, s), tx = __shfl(px, s), ty = __shfl(py, s), sl = __shfl(map_slot, s);
        // width class BW: 0: W = 32 | 1, 2: W <= 64 / 96 and H * W <= 4096 (rows of 2 / 3 32-bit words) | 3, 4: wider or larger (2 / 3 64-bit words)
        if constexpr (BW == 0) { if (rows_pow2 == 16 && !RG_BFS_NO_ROW16) bfs_rows_w32<true>(S, c, active, env_s, tx, ty, sl, row); else bfs_rows_w32<false>(S, c, active, env_s, tx, ty, sl, row); }
        else if constexpr (BW == 1 || BW == 2) {
            const bool comp = bfs_rows_n32<BW + 1>(S, c, active, env_s, tx, ty, sl, row, grp_lanes, mcbase + s, (own_req >> s) & 1ull);
            const uint64_t cb = __ballot(comp);  // (uniform inside a group)
            if (served_by >= 0) my_complete = (cb >> (served_by * rows_pow2)) & 1ull;
        } else bfs_rows<BW - 1, false>(S, c, lds, active, env_s, tx, ty, sl, row);
#ifdef RG_EXP_BFS_ONE_ROUND
        break;  // (experiment build only -- wrong pictures: a wave's surplus requests cost it nothing, as if other waves had built them; profiles/r06_experiments.txt)
#endif
    }
    return my_complete;
}

// Does the (partial) map in `slot` have a value at every chaser's own cell?  (see bfs_rows_n32)
__device__ __forceinline__ bool dist_map_sufficient(const RgState &S, const RgConfig &c, const Env &E, int slot) {
    const uint16_t *dist = S.dc_map + ((size_t)E.e * RG_DIST_SLOTS + slot) * S.hw;
    const int nrooms = c.room_num_x * c.room_num_y;
    bool ok = true;
    for (int q0 = 0; q0 < nrooms; q0 += 2) {  // two monsters per round: their probes are in flight together (one load -> test per monster was a round trip each)
        const bool has1 = q0 + 1 < nrooms;
        const uint32_t wa = E.mc[q0 * WAVE], wb = E.mc[(has1 ? q0 + 1 : q0) * WAVE];
        const uint32_t da = dist[POS_Y(wa) * c.width + POS_X(wa)], db = dist[POS_Y(wb) * c.width + POS_X(wb)];
        if (((wa >> 24) & (MF_PENDING | MF_RANDOM)) == MF_PENDING && da == DIST_INF) ok = false;
        if (has1 && ((wb >> 24) & (MF_PENDING | MF_RANDOM)) == MF_PENDING && db == DIST_INF) ok = false;
    }
    return ok;
}

// The walkable mask of the envs in `want` is about to change (descent, an opening search): save it into every partial map that still
// relies on the level's own cells (bits of dc_part without dc_own), so that the map can be continued later exactly as it began (bfs_rows_n32).
// Reads the cells in global memory: they hold the state before this turn's window write-back.
__device__ __forceinline__ void snapshot_walk_service(const RgState &S, const RgConfig &c, int lane, int e, bool want) {
    uint32_t todo = 0;
    if (want) todo = (uint32_t)S.dc_part[e] & ~(uint32_t)S.dc_own[e];
    uint64_t m = __ballot(todo != 0);
    if (!m) return;
    const int W = c.width, H = c.height, wpr = RG_WALK_WORDS(W);  // = the WN of the step class that reads it back (bfs_rows_n32)
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int env_s = __shfl(e, src);
        const uint32_t t = (uint32_t)__shfl((int)todo, src);
        for (int row = lane; row < H; row += WAVE) {
            uint32_t wk[3];
            row_walk_mask<3>(S.cell + (size_t)env_s * S.hw + row * W, W, wk);
            for (int sl = 0; sl < RG_DIST_SLOTS; sl++) {
                if (!((t >> sl) & 1u)) continue;
                uint32_t *o = S.dc_walk + (((size_t)env_s * RG_DIST_SLOTS + sl) * H + row) * wpr;
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (k < wpr) o[k] = wk[k];
            }
        }
    }
    if (todo) S.dc_own[e] = (uint16_t)(S.dc_own[e] | todo);
}

// ---------------------------------------------------------------------------------------------
// the turn (core/src/actions.rs, character/{player,fight,enemies}.rs)
// ---------------------------------------------------------------------------------------------
#define R_REDRAW 1u
#define R_STATUS 2u
#define R_GRAVE 4u
#define R_HIST_STALE 8u
#define R_HIST_CHANGED 16u  // a cell became VISITED: the history mirror must be rewritten at the next Redraw
#define MSG_HIT_FROM (1u << 8)
#define MSG_HIT_TO (2u << 8)
#define MSG_MISS_TO (4u << 8)
#define MSG_MISS_FROM (8u << 8)
#define MSG_KILLED (16u << 8)
#define MSG_SECRET_DOOR (32u << 8)
#define MSG_NO_DOWNSTAIR (64u << 8)

enum { ACT_INVALID = -1, ACT_MOVE = 0, ACT_MOVE_UNTIL, ACT_DOWNSTAIR, ACT_SEARCH, ACT_NOOP };

// KeyMap::ai (input.rs:73-100)
__device__ __forceinline__ int decode_key(uint32_t key, int &dir) {
    uint32_t lower = key | 0x20u;
    int d = -1;
    switch (lower) {
    case 'k': d = 0; break; case 'j': d = 1; break; case 'h': d = 2; break; case 'l': d = 3; break;
    case 'y': d = 4; break; case 'u': d = 5; break; case 'b': d = 6; break; case 'n': d = 7; break;
    }
    if (d >= 0 && ((key >= 'a' && key <= 'z') || (key >= 'A' && key <= 'Z'))) { dir = d; return (key & 0x20u) ? ACT_MOVE : ACT_MOVE_UNTIL; }
    if (key == '.') return ACT_NOOP;
    if (key == 's') return ACT_SEARCH;
    if (key == '>') return ACT_DOWNSTAIR;
    return ACT_INVALID;
}

// fight::attack_rate (fight.rs:84-87) + Parcent::truncate
__device__ __forceinline__ uint32_t attack_rate(int64_t level, int64_t armor, int64_t revision) {
    int64_t v = (level + armor + revision + 1) * 5;
    return (uint32_t)(v < 0 ? 0 : (v > 100 ? 100 : v));
}

// Player::level_up (player.rs:185-197, 345-352)
__device__ __forceinline__ bool level_up(const RgConfig &c, Env &E, uint32_t exp) {
    E.exp += exp;
    int cur = E.plvl - 1, diff = 0;
    if (cur >= c.n_level_exps) return false;
    while (cur + diff < c.n_level_exps && !(E.exp < c.level_exps[cur + diff])) diff++;
    if (diff == 0) return false;
    E.plvl += diff;
    int add = 0;
    for (int i = 0; i < diff; i++) add += (int)range64(E.re, 1, 11);
    E.hpmax += add; E.hp += add;
    return true;
}

// actions::player_attack + fight::player_attack (actions.rs:140-166, fight.rs:6-39):
// the wielded weapon's dice / hit_plus / dam_plus (or bare hands 1d4; the default pack wields a mace 2d4 +1,+1) come resolved from the config
// (rg_items.cpp); strength 16 => +0 / +0 (fight.rs:89-109); the monster is always `running` by the time of the roll
// (hp / exp_gain: the monster's mon_hp / mon_exp words, loaded by the caller together with everything else the player's move may need)
__device__ __forceinline__ void player_attack(const RgState &S, const RgConfig &c, Env &E, int slot, uint32_t &react, int hp, uint32_t exp_gain) {
    int idx = slot * E.n + E.e;
    uint32_t w = mon_rd<true>(S, E, slot);
    E.quiet = 0;
    if (!((w >> 24) & MF_ACTIVE)) { w |= (uint32_t)MF_ACTIVE << 24; E.mon_active++; mon_wr<true>(S, E, slot, w); }
    uint32_t type = (w >> 16) & 0xff;
    int64_t def = (int64_t)c.mon[type].defense - (int64_t)lev_add_of(c, E.dlevel);
    uint32_t rate = attack_rate(E.plvl, def, c.wpn_hit_plus);
    if (parcent(E.re, rate)) {
        int dmg = c.wpn_dam_plus;  // Dice::random (character/mod.rs:229-234): `times` rolls of 1..=max as i64, + dam_plus (fight.rs:66)
        for (int t = 0; t < c.wpn_times; t++) dmg += (int)range64(E.re, 1, (uint64_t)c.wpn_max + 1);
        react |= MSG_HIT_TO;
        if (hp <= dmg) {  // Enemy::get_damage (enemies.rs:205-213)
            mon_wr<true>(S, E, slot, 0);
            E.mon_alive--; E.mon_active--;
            if (level_up(c, E, exp_gain)) react |= R_STATUS;
            react |= MSG_KILLED | R_REDRAW;
        } else S.mon_hp[idx] = dmg - hp;  // reference quirk: stores damage - cur
    } else react |= MSG_MISS_TO;
}


// ---------------------------------------------------------------------------------------------
// the player's turn on a register window.  Every tile the player's own action reads or writes lies in the 5x5 block around
// the position the action starts from (3x3 of the old cell for can_move / Cell::left / search, 3x3 of the new cell for
// Cell::approached), except the whole-room updates of enters_room / leaves_room.  Against global memory those ~25 accesses
// are ~25 dependent round trips (load -> test -> store, one after the other); here they are ONE round of 25 independent
// loads, register arithmetic with compile-time indices, and a write-back of the cells that changed.  The whole-room updates
// are recorded as rectangles and applied by the wave (fill_service), mirrored into the window in the reference's order.
// ---------------------------------------------------------------------------------------------
struct Win {
    lds_u16 *v;            // cell (ox + i, oy + j) = v[((j + 2) * 5 + (i + 2)) * WAVE] (this lane's column of the wave's [25][64] LDS block); 0 outside the grid
    uint32_t inb, dirty;   // bit k: cell k lies inside the grid / was modified since the load; inb bit 31 (WIN_STAIR): a staircase was next to the centre of a
                           // window loaded in this turn (step_wave: the env asks for its next level's structure)
    int ox, oy;
};
// The window lives in LDS, not in registers: 25 VGPRs held from the first load to the last line of the turn (the stair test of the tail reads it)
// were a tenth of the kernel's register budget, and a run-time index into registers is a 25-deep select chain where LDS takes an address.
#define WIN_K(i, j) (((j) + 2) * 5 + (i) + 2)
#define WIN_SLOTS 25
#define WIN_STAIR 0x80000000u
#define WV(w, k) ((uint32_t)(w).v[(k) * WAVE])
#define WSET(w, k, val) ((w).v[(k) * WAVE] = (uint16_t)(val))

#ifndef RG_NX_NEAR
#define RG_NX_NEAR 1   // how near a staircase must be for the env to ask for its next level's structure (cells, Chebyshev; 2 measured: more structures generated, the same number used)
#endif
__device__ __forceinline__ void win_load(const RgConfig &c, const uint16_t *cell, Win &w, int ox, int oy) {
    w.ox = ox; w.oy = oy; w.inb &= WIN_STAIR; w.dirty = 0;
    uint32_t t[25];
    bool st = false;
#pragma unroll
    for (int j = -2; j <= 2; j++)
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            const int x = ox + i, y = oy + j;
            const bool in = in_bounds(c, x, y);
            const uint32_t val = cell[in ? y * c.width + x : oy * c.width + ox];  // unconditional load: all 25 are in flight together
            t[WIN_K(i, j)] = in ? val : 0u;
            if (in) w.inb |= 1u << WIN_K(i, j);
            if (i >= -RG_NX_NEAR && i <= RG_NX_NEAR && j >= -RG_NX_NEAR && j <= RG_NX_NEAR) st = st || (in && (val & C_SURF_MASK) == S_STAIR);
        }
    if (st) w.inb |= WIN_STAIR;
#pragma unroll
    for (int k = 0; k < 25; k++) WSET(w, k, t[k]);
}
__device__ __forceinline__ void win_flush(const RgConfig &c, uint16_t *cell, Win &w) {
    if (!w.dirty) return;
#pragma unroll
    for (int j = -2; j <= 2; j++)
#pragma unroll
        for (int i = -2; i <= 2; i++)
            if ((w.dirty >> WIN_K(i, j)) & 1u) cell[(w.oy + j) * c.width + w.ox + i] = (uint16_t)WV(w, WIN_K(i, j));
    w.dirty = 0;
}
// LDS of a step wave: every per-lane column FIRST, at compile-time offsets -- window, parked overlay words + glyph bytes, monster cache (its length is the
// config's) -- so that a column's address is lane * 4 (or * 2) + an immediate of the DS instruction: ONE live register for all of them, where run-time offsets
// behind the generator's staging area (rounds 2-5) took a VGPR per column from the top of the wave to its tail.  The staging area (generator grid + tables, BFS
// planes of wide grids) follows at `stage_off`, a kernel argument.
#define STEP_LDS_WIN 0
#define STEP_LDS_OVL (STEP_LDS_WIN + WIN_SLOTS * WAVE * 2)
#define STEP_LDS_LUT (STEP_LDS_OVL + (RG_OVL_MAX + 6) * WAVE * 4 + 64)   // glyph -> gray value, 128 floats (a bound gray observation tensor: mirror_update)
#define STEP_LDS_MC (STEP_LDS_LUT + 128 * 4)
__device__ __forceinline__ uint32_t win_get(const Win &w, int k) { return WV(w, k); }  // run-time index
__device__ __forceinline__ void win_set(Win &w, int k, uint32_t val) {
    WSET(w, k, val);
    w.dirty |= 1u << k;
}
// apply `v = (v & ~clr) | set` to the window cells inside the half-open rectangle, and mark them for write-back (the wave's global
// fill of the same rectangle works on the pre-step values of these cells)
__device__ __forceinline__ void win_rect(Win &w, int x0, int y0, int x1, int y1, uint32_t clr, uint32_t set) {
#pragma unroll
    for (int j = -2; j <= 2; j++)
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            const int x = w.ox + i, y = w.oy + j;
            if (x >= x0 && x < x1 && y >= y0 && y < y1 && ((w.inb >> WIN_K(i, j)) & 1u)) {
                WSET(w, WIN_K(i, j), (WV(w, WIN_K(i, j)) & ~clr) | set);
                w.dirty |= 1u << WIN_K(i, j);
            }
        }
}
__device__ __forceinline__ uint32_t pack_rect(int x0, int y0, int x1, int y1) { return (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24); }

// Floor::can_move_impl (floor.rs:169-182) for the player standing on the window centre
__device__ __forceinline__ bool win_can_move(const Win &w, int dx, int dy) {
    const int k = WIN_K(dx, dy);
    if (!((w.inb >> k) & 1u)) return false;
    const uint32_t nc = win_get(w, k);
    bool res = can_walk(nc) && !(nc & (C_HIDDEN | C_LOCKED));
    if (dx != 0 && dy != 0) res = res && can_walk(win_get(w, WIN_K(dx, 0))) && can_walk(win_get(w, WIN_K(0, dy)));
    return res;
}

// Whole-room updates requested by a lane during its move, served by the wave after the per-lane code
struct FillReq { uint32_t leave, enter; };  // packed half-open rects, 0 = none: leaves_room clears VISIBLE, enters_room sets DRAWN | VISIBLE

// actions::move_player + get_item (actions.rs:168-231).  Returns `done` (true = a MoveUntil run stops here).
// The window is centred on the player's position before the move.
// rooms_l: the lane's column of the wave's parked words (step_wave) when rect and meta of the room of the cell the key starts on ([RG_OVL_MAX + 2 / + 3]) and of
// the cell it points at ([+ 4 / + 5]) were fetched into LDS with the first round of loads -- for the mirror update -- and this is the key's first turn: the two
// door branches then read them there instead of paying a memory round trip each (divergent branches: the wave runs them one after the other).  Else nullptr.
__device__ __forceinline__ bool move_player(const RgState &S, const RgConfig &c, Env &E, Win &w, int d, uint32_t &react, FillReq &fr, const lds_u32 *rooms_l) {
    const int nrooms = c.room_num_x * c.room_num_y, n = E.n, e = E.e;
    const int dx = dir_dx(d), dy = dir_dy(d);
    // The lanes of a wave take different branches here -- an attack (monster hp / exp), a step through a door (room meta + rect of the room left
    // and of the room entered), a step onto gold (the gold table) -- and a wave runs its branches one after the other.  Inside each branch everything
    // it may need is requested TOGETHER: hp with exp, meta with rect, the whole gold table at once -- one round trip per branch where rounds 1-3 had
    // dependent chains (hp -> exp; meta -> rect, twice; gold slot after gold slot -> amount: up to ten round trips for the union of a wave's cases).
    // (Hoisting all of it in front of the branches -- one round trip for the whole move -- spilled the capped W <= 32 kernel: measured, not kept.)
    if (!win_can_move(w, dx, dy)) return true;  // Notify(CantMove): no mirror effect
    const int nx = E.px + dx, ny = E.py + dy;
    const int ms = mon_find(S, E, nrooms, POS(nx, ny));
    if (ms >= 0) {
        const int a_hp = S.mon_hp[ms * n + e];
        const uint32_t a_exp = S.mon_exp[ms * n + e];
        player_attack(S, c, E, ms, react, a_hp, a_exp);
        return true;
    }
    const int nk = WIN_K(dx, dy);
    const int rid_o = (WV(w, WIN_K(0, 0)) & C_DOOR) ? room_id_of(c, E.px, E.py) : -1;   // Floor::leaves_room's room
    // ---- Floor::player_out at the old cell (field-of-view, floor.rs:201-312) ----
    if (rid_o >= 0) {  // Floor::leaves_room (floor.rs:249-261)
        const uint32_t meta_o = rooms_l ? rooms_l[(RG_OVL_MAX + 3) * WAVE] & 0xffu : (uint32_t)S.room_meta[rid_o * n + e];
        const uint32_t rect_o = rooms_l ? rooms_l[(RG_OVL_MAX + 2) * WAVE] : S.room_rect[rid_o * n + e];
        if ((meta_o & RM_VISITED) && (meta_o & RM_DARK)) {
            int x0, y0, x1, y1;
            if ((meta_o & RM_KIND_MASK) == RK_EMPTY) assigned_area(c, rid_o, x0, y0, x1, y1);
            else unpack_rect(rect_o, x0, y0, x1, y1);
            if (x1 - x0 > 2 && y1 - y0 > 2) {
                fr.leave = pack_rect(x0 + 1, y0 + 1, x1 - 1, y1 - 1);
                win_rect(w, x0 + 1, y0 + 1, x1 - 1, y1 - 1, C_VISIBLE, 0);
            }
        }
    }
#pragma unroll
    for (int j = -1; j <= 1; j++)
#pragma unroll
        for (int i = -1; i <= 1; i++) {  // Cell::left on the 3x3 around the old cell
            const uint32_t v = WV(w, WIN_K(i, j));
            if (((w.inb >> WIN_K(i, j)) & 1u) && (v & C_SURF_MASK) == S_FLOOR && (v & C_DARK) && (v & C_VISIBLE)) {
                WSET(w, WIN_K(i, j), v & ~C_VISIBLE);
                w.dirty |= 1u << WIN_K(i, j);
            }
        }
    // ---- Floor::player_in at the new cell ----
    uint32_t here = win_get(w, nk);
    if (here & C_DOOR) {
        const int rid_n = room_id_of(c, nx, ny);
        if (rid_n >= 0) {
            const uint32_t meta_n = rooms_l ? rooms_l[(RG_OVL_MAX + 5) * WAVE] & 0xffu : (uint32_t)S.room_meta[rid_n * n + e];
            const uint32_t rect_n = rooms_l ? rooms_l[(RG_OVL_MAX + 4) * WAVE] : S.room_rect[rid_n * n + e];
            if (!(meta_n & RM_VISITED)) {  // Floor::enters_room (floor.rs:231-247)
                S.room_meta[rid_n * n + e] = (uint8_t)(meta_n | RM_VISITED);
                if ((meta_n & RM_KIND_MASK) == RK_NORMAL && !(meta_n & RM_DARK)) {
                    int x0, y0, x1, y1;
                    unpack_rect(rect_n, x0, y0, x1, y1);
                    fr.enter = pack_rect(x0, y0, x1, y1);
                    win_rect(w, x0, y0, x1, y1, 0, C_DRAWN | C_VISIBLE);
                }
            }
            activate_room<true>(S, c, E, rid_n);
        }
        here = win_get(w, nk);
    }
    if (!(here & C_VISITED)) { win_set(w, nk, here | C_VISITED); react |= R_HIST_CHANGED; }
#pragma nounroll
    for (int ddy = -1; ddy <= 1; ddy++)
#pragma unroll
        for (int ddx = -1; ddx <= 1; ddx++) {  // Cell::approached (field.rs:20-26) on the 3x3 around the new cell: nine window cells at a run-time offset
            const int k = WIN_K(dx + ddx, dy + ddy);  // (inside the 5x5 window: |dx + ddx| <= 2)
            const uint32_t v = win_get(w, k);
            const bool diag = ddx != 0 && ddy != 0;
            if (((w.inb >> k) & 1u) && !(diag && (v & C_SURF_MASK) == S_PASSAGE) && !(v & C_HIDDEN) && (v & (C_DRAWN | C_VISIBLE)) != (C_DRAWN | C_VISIBLE))
                win_set(w, k, v | C_DRAWN | C_VISIBLE);
        }
    E.px = nx; E.py = ny;
    react |= R_REDRAW;
    const uint32_t v = win_get(w, nk);
    if ((v & C_GOLD) && c.can_pickup) {  // ItemBox::entry -> Merge into the pack's gold, or its first free slot (itembox.rs:30-40); a full pack without
                                         // a Gold item makes get_item return None and the gold stays on the floor (actions.rs:206-231)
        const uint32_t want = POS(nx, ny) | 0x10000u;
        for (int s0 = 0; s0 < nrooms; s0 += 4) {  // the gold table, four slots (positions and amounts) per round: the mini dungeon's whole table at once
            uint32_t p4[4], a4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sl = s0 + k < nrooms ? s0 + k : s0;
                p4[k] = S.gold_pos[sl * n + e]; a4[k] = S.gold_amt[sl * n + e];
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (s0 + k < nrooms && p4[k] == want) { E.gold += a4[k]; S.gold_pos[(s0 + k) * n + e] = 0; }
        }
        win_set(w, nk, v & ~C_GOLD);
        react |= R_STATUS;
        return true;
    }
    return false;
}

// the wave applies the whole-room updates of its lanes to the global grids: leaves_room first (clears), then enters_room (sets),
// 64 cells per round instead of one lane walking the room
__device__ __forceinline__ void fill_service(const RgState &S, const RgConfig &c, int lane, int e, const FillReq &fr) {
#pragma unroll
    for (int kind = 0; kind < 2; kind++) {
        const uint32_t mine = kind ? fr.enter : fr.leave;
        uint64_t m = __ballot(mine != 0);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t r = (uint32_t)__shfl((int)mine, src);
            uint16_t *cell = S.cell + (size_t)__shfl(e, src) * S.hw;
            int x0, y0, x1, y1;
            unpack_rect(r, x0, y0, x1, y1);
            const int rw = x1 - x0, area = rw * (y1 - y0);
            for (int t = lane; t < area; t += WAVE) {
                const int yy = small_div(t, rw), xx = t - yy * rw;
                uint16_t *p = cell + (y0 + yy) * c.width + x0 + xx;
                *p = kind ? (uint16_t)(*p | C_DRAWN | C_VISIBLE) : (uint16_t)(*p & ~C_VISIBLE);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the owner lane's window write-back (and a following fill) must land after these stores
        }
    }
}

// Floor::search (floor.rs:349-370) on the window (centred on the player)
__device__ __forceinline__ bool do_search(const RgConfig &c, Env &E, Win &w, uint32_t &react) {
    bool opened = false;  // a cell became walkable
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const int k = WIN_K(kDXc[d], kDYc[d]);
        if (!((w.inb >> k) & 1u)) continue;
        uint32_t v = WV(w, k);
        const uint32_t v_in = v;
        if ((v & C_HIDDEN) && does_happen(E.rd, c.passage_unlock_rate_inv))
            v = (v & ~(C_LOCKED | C_HIDDEN | C_SURF_MASK)) | C_VISIBLE | S_PASSAGE;
        if ((v & C_LOCKED) && does_happen(E.rd, c.door_unlock_rate_inv)) {
            v = (v & ~(C_LOCKED | C_HIDDEN | C_SURF_MASK)) | C_VISIBLE | S_DOOR;
            react |= MSG_SECRET_DOOR;
        }
        if (v != v_in) { WSET(w, k, v); w.dirty |= 1u << k; opened = true; }
    }
    react |= R_REDRAW;
    return opened;
}

// Player::turn_passed + heal (player.rs:163-176,221-240)
__device__ __forceinline__ void turn_passed(const RgConfig &c, Env &E, uint32_t &react) {
    E.food -= 1;  // u32: wraps in release builds
    if (E.food == 0) return;  // [PlayerEvent::Dead], ignored by after_turn (actions.rs:75)
    uint32_t hunger = c.hunger_time / 10;
    if (E.food == hunger || E.food == hunger * 2) react |= R_STATUS;
    E.quiet += 1;
    int64_t quiet = E.quiet, level = E.plvl, heal;
    if (level < 8) { heal = quiet + (level << 1) - 20; heal = heal < 0 ? 0 : (heal > 1 ? 1 : heal); }
    else if (quiet >= 3) heal = (int64_t)range64(E.re, 1, (uint64_t)(level - 6));
    else heal = 0;
    if (heal > 0) {
        E.hp += (int)heal;
        if (E.hp > E.hpmax) E.hp = E.hpmax;
        E.quiet = 0;
        react |= R_STATUS;
    }
}

// smallest PENDING monster key strictly greater than `last` (BTreeMap iteration order); -1 if none
__device__ __forceinline__ int next_pending(const RgState &S, const Env &E, int nrooms, int last, int &slot_out) {
    int best = 0x7fffffff, bs = -1;
    for (int s = 0; s < nrooms; s++) {
        uint32_t w = mon_rd<true>(S, E, s);
        if (!((w >> 24) & MF_PENDING)) continue;
        int key = (int)(w & 0xffff);
        if (key > last && key < best) { best = key; bs = s; }
    }
    slot_out = bs;
    return bs < 0 ? -1 : best;
}

// EnemyHandler::move_actives, RNG part (enemies.rs:399-404, rogue/mod.rs:383): the per-monster draws do not
// depend on positions, so they are taken first (same per-stream order) to learn whether a dist map is needed.
// A monster that moves at random this turn carries MF_RANDOM and its direction (3 bits) in the flag byte of its CACHE word, next to MF_PENDING --
// per-lane bit tables for that (a mask + 4 bits per slot) were five registers held across the BFS, and capped the table at 32 slots.
__device__ __forceinline__ bool monsters_prepass(const RgState &S, const RgConfig &c, Env &E) {
    const int nrooms = c.room_num_x * c.room_num_y;
    for (int s = 0; s < nrooms; s++) {  // the taken map: every active monster is pending
        uint32_t w = mon_rd<true>(S, E, s);
        if (((w >> 24) & (MF_ALIVE | MF_ACTIVE)) == (MF_ALIVE | MF_ACTIVE)) E.mc[s * WAVE] = w | ((uint32_t)MF_PENDING << 24);  // PENDING lives in the cache only
    }
    bool need_map = false;
    int last = -1, slot;
    while ((last = next_pending(S, E, nrooms, last, slot)) >= 0) {
        uint32_t attr = c.mon[(mon_rd<true>(S, E, slot) >> 16) & 0xff].attr;
        bool rnd = false;
        if (does_happen(E.re, 2) && (attr & EA_RANDOM)) rnd = true;
        else if (!does_happen(E.re, 5) && (attr & EA_CONFUSED)) rnd = true;
        if (rnd) {
            const uint32_t d = (uint32_t)range64(E.rd, 0, 8);
            E.mc[slot * WAVE] = mon_rd<true>(S, E, slot) | ((MF_RANDOM | (d << MF_DIR_SHIFT)) << 24);
        } else need_map = true;
    }
    return need_map;
}

// DistCache::make_dist_map lookup (rogue/mod.rs:504-517): FIFO ring of 9 maps keyed by target coord only
__device__ __forceinline__ bool dist_cache_lookup(const RgState &S, const Env &E, uint32_t key, int &slot) {
    const int len = S.dc_len[E.e], head = S.dc_head[E.e];
    uint32_t keys[RG_DIST_SLOTS];
#pragma unroll
    for (int i = 0; i < RG_DIST_SLOTS; i++) keys[i] = S.dc_key[i * E.n + E.e];  // one round of independent loads instead of a dependent scan
    int hit = -1;
#pragma unroll
    for (int i = RG_DIST_SLOTS - 1; i >= 0; i--) {  // FIFO order from `head`; the first match wins (keys are unique anyway)
        int age = i - head; if (age < 0) age += RG_DIST_SLOTS;
        if (age < len && keys[i] == key) hit = i;
    }
    if (hit >= 0) { slot = hit; return true; }
    if (len < RG_DIST_SLOTS) { slot = head + len; if (slot >= RG_DIST_SLOTS) slot -= RG_DIST_SLOTS; S.dc_len[E.e] = (uint8_t)(len + 1); }
    else { slot = head; S.dc_head[E.e] = (uint8_t)(head + 1 >= RG_DIST_SLOTS ? 0 : head + 1); }
    S.dc_key[slot * E.n + E.e] = (uint16_t)key;
    return false;
}

// EnemyHandler::move_actives moves + actions::move_active_enemies attacks
// (enemies.rs:366-424, rogue/mod.rs:339-397, actions.rs:82-119, fight.rs:41-72)
// EnemyHandler::move_actives moves + actions::move_active_enemies attacks
// (enemies.rs:366-424, rogue/mod.rs:339-397, actions.rs:82-119, fight.rs:41-72)
__device__ __forceinline__ bool monsters_move(const RgState &S, const RgConfig &c, Env &E, int map_slot, uint32_t &react) {
    const int nrooms = c.room_num_x * c.room_num_y, W = c.width, e = E.e;
    const uint16_t *dist = S.dc_map + ((size_t)e * RG_DIST_SLOTS + (map_slot < 0 ? 0 : map_slot)) * S.hw;
    const uint32_t ppos = POS(E.px, E.py);
    int n_att = 0;
    int last = -1, slot;
    while ((last = next_pending(S, E, nrooms, last, slot)) >= 0) {
        const uint32_t wt = mon_rd<true>(S, E, slot);
        const uint32_t w = wt & ~((uint32_t)MF_TURN_BITS << 24);
        E.mc[slot * WAVE] = w;  // leaves the taken map; not yet in the new one, so it never blocks itself
        int cx = POS_X(w), cy = POS_Y(w);
        uint32_t fin = w & 0xffff;
        bool reach = false;
        const bool random = (wt >> 24) & MF_RANDOM;
        // one round of independent loads: the 3x3 tiles around the monster (can_move needs the target and, for diagonals, its two
        // orthogonal neighbours) and, for a chaser, the 3x3 of the dist map; the decision below then runs on registers only
        uint32_t walk = 0, dv[9];
#pragma unroll
        for (int d = 0; d < 9; d++) {
            const int nx = cx + kDXc[d], ny = cy + kDYc[d];
            const bool in = in_bounds(c, nx, ny);
            const int idx = in ? ny * W + nx : cy * W + cx;
            const uint32_t cv = E.cell[idx];
            const uint32_t dd = random ? DIST_INF : (uint32_t)dist[idx];
            dv[d] = in ? dd : DIST_INF;
            if (in && can_walk(cv)) walk |= 1u << d;
        }
        // Floor::can_move_impl for an enemy (floor.rs:169-182): bit d of `cm`
        uint32_t cm = walk;
        if (!((walk >> 2) & (walk >> 0) & 1u)) cm &= ~(1u << 4);  // LeftUp needs Left and Up
        if (!((walk >> 3) & (walk >> 0) & 1u)) cm &= ~(1u << 5);  // RightUp: Right, Up
        if (!((walk >> 2) & (walk >> 1) & 1u)) cm &= ~(1u << 6);  // LeftDown: Left, Down
        if (!((walk >> 3) & (walk >> 1) & 1u)) cm &= ~(1u << 7);  // RightDown: Right, Down
        // skip() of EnemyHandler::move_actives (enemies.rs:386-387) for the nine cells around this monster, ONCE: bit d = the cell in direction d holds an asleep monster
        // or an active one that has already moved (the monster itself is in neither map while it moves).  The table does not change while this monster decides, and asking it cell by
        // cell was nine passes over the wave's LDS monster table per monster.
        uint32_t occ = 0;
        for (int s = 0; s < nrooms; s++) {
            const uint32_t o = mon_rd<true>(S, E, s);
            const uint32_t fl = o >> 24;
            const int ddx = POS_X(o) - cx, ddy = POS_Y(o) - cy;
            if (s != slot && (fl & MF_ALIVE) && !(fl & MF_PENDING) && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1)
                occ |= 1u << (uint32_t)((0x716382504ull >> (4 * ((ddy + 1) * 3 + ddx + 1))) & 15ull);  // (dx, dy) -> direction index (kDXc / kDYc order)
        }
        if (random) {  // Dungeon::move_enemy_randomly
            int d = (int)((wt >> (24 + MF_DIR_SHIFT)) & 7u);
            uint32_t np = POS(cx + dir_dx(d), cy + dir_dy(d));
            if (!((occ >> d) & 1u) && ((cm >> d) & 1u)) {
                if (np == ppos) reach = true; else fin = np;
            }
        } else {  // Dungeon::move_enemy: greedy step on the (possibly stale) dist map, 9 directions incl. Stay
            uint32_t best = DIST_INF; bool found = false; uint32_t bp = fin;
#pragma unroll
            for (int d = 0; d < 9; d++) {
                const uint32_t np = POS(cx + kDXc[d], cy + kDYc[d]);
                if (reach || ((occ >> d) & 1u)) continue;
                const uint32_t nd = dv[d];
                if (nd == 0 && ((cm >> d) & 1u)) { reach = true; continue; }
                if (nd != DIST_INF && nd > 0 && (!found || nd < best)) { best = nd; bp = np; found = true; }
            }
            if (!reach && found) fin = bp;
        }
        // A reaching monster stays on its cell and is remembered by a bit of its cache word (a packed slot list in a register capped the table
        // at 64 slots): the attacks below run in BTreeMap order = ascending position, which is the order these monsters were visited in.
        if (reach) { E.mc[slot * WAVE] = w | ((uint32_t)MF_REACH << 24); n_att++; }
        if (fin == (w & 0xffff)) {
            // BTreeMap::insert on its own key replaces a monster that already moved onto this cell
            for (int s = 0; s < nrooms; s++) {
                if (s == slot) continue;
                uint32_t o = mon_rd<true>(S, E, s);
                uint32_t fl = o >> 24;
                if ((fl & MF_ALIVE) && (fl & MF_ACTIVE) && !(fl & MF_PENDING) && (o & 0xffff) == fin) { mon_wr<true>(S, E, s, 0); E.mon_alive--; E.mon_active--; }
            }
        } else mon_wr<true>(S, E, slot, (w & 0xffff0000u) | fin);
    }
    if (n_att > 0) E.quiet = 0;  // player.buttle()
    bool did_hit = false;
    uint32_t lev_add = lev_add_of(c, E.dlevel);
    int last_key = -1;
    for (int i = 0; i < n_att; i++) {
        int s = -1, best = 0x7fffffff;
        for (int q = 0; q < nrooms; q++) {  // the next attacker: smallest position above the last one
            const uint32_t o = mon_rd<true>(S, E, q);
            const int key = (int)(o & 0xffff);
            if (((o >> 24) & MF_REACH) && key > last_key && key < best) { best = key; s = q; }
        }
        if (s < 0) break;
        last_key = best;
        E.mc[s * WAVE] = mon_rd<true>(S, E, s) & ~((uint32_t)MF_REACH << 24);
        uint32_t type = (mon_rd<true>(S, E, s) >> 16) & 0xff;
        uint32_t rate = attack_rate((int64_t)c.mon[type].level + lev_add, c.armor_def /* Player::arm: the default pack wears ring mail 3 + 1 */, 0 /* hit_prob_plus(10) */);
        int sum = 0; bool hit = false;
        for (int k = 0; k < c.mon[type].n_att; k++) {
            if (!parcent(E.re, rate)) continue;
            hit = true;
            int times = c.mon[type].att[k][0], mx = c.mon[type].att[k][1];
            for (int t = 0; t < times; t++) sum += (int)range64(E.re, 1, (uint64_t)mx + 1);
        }
        if (hit) {
            react |= MSG_HIT_FROM;
            did_hit = true;
            E.hp = E.hp - sum > 0 ? E.hp - sum : 0;  // Player::get_damage (player.rs:177-184)
            if (E.hp == 0) {  // the player died: the remaining attackers do not get their turn (and a MoveUntil run goes on: leave no marks behind)
                for (int q = 0; q < nrooms; q++) E.mc[q * WAVE] = mon_rd<true>(S, E, q) & ~((uint32_t)MF_REACH << 24);
                react |= R_GRAVE; return true;
            }
        } else react |= MSG_MISS_FROM;
    }
    if (did_hit) react |= R_STATUS;
    return false;
}

// What a cell shows without overlays (rg_obs.hip's decode, core/src/lib.rs:264-285 + rogue/mod.rs:278-300): the glyph, bit 7 = an object on it is drawn
__device__ __forceinline__ uint32_t base_glyph(const RgConfig &c, uint32_t v, int y) {
    const bool inner = y >= 1 && y < c.height - 1;
    uint32_t gl = ' ';
    if (inner && (v & C_VISIBLE)) gl = glyph_of(v);
    if (inner && (v & (C_VISIBLE | C_DRAWN))) gl = ((v & C_GOLD) ? (uint32_t)'*' : gl) | 0x80u;
    return gl;
}
// The incremental form of draw_screen for an ordinary Redraw (step_wave).  S.ovl remembers, per monster slot and for the player, where the overlay stood at the
// env's last Redraw and whether it SHOWED then (OVL_SHOW; OVL_UNKNOWN after a Redraw that was drawn from the tiles): a sleeping monster that shows as it did,
// on a cell the turn did not touch, costs nothing -- no tile load, no store.  Stores go to the cells that really change: the turn's written-back window cells,
// overlays that moved / appeared / disappeared, the player's old and new cell (a byte store is a partial-line write: twelve of them per lane cost k_step 7 us).
#define OVL_SHOW 0x80u     // (the y field of a position holds 0..63: its two top bits are free)
#define OVL_UNKNOWN 0x40u
#define OVL_NONE 0xffffu
// Returns false -- nothing written -- when a cell it would have to write lies outside the window: its tile is not at hand, and a load here would wait for every
// store of the turn; the Redraw then goes to the observation pass as before.
template <bool BND>  // BND: the handle has a bound observation tensor (rg_obs_bind) -- a step-kernel instance of its own, so that the ordinary one carries none of it
__device__ __forceinline__ bool mirror_update(const RgState &S, const RgConfig &c, const Env &E, const Win &w, uint32_t react, int rid0) {
    const int W = c.width, nrooms = c.room_num_x * c.room_num_y, n = E.n;
    // (the env index through an opaque move: the addresses below are then computed HERE -- left alone, the compiler computes `S.ovl + ... + e`, `S.hist + e * hw`
    // at the top of the kernel, spills them, and reloads them here one by one, each reload waiting for every store the turn has issued: 5 us per wave)
    int e = E.e;
    asm volatile("" : "+v"(e));
    uint8_t *scr = S.screen + (size_t)e * S.hw;
    // a bound GRAY observation tensor (rg_obs_bind): every byte written to the mirror below is also written to the env's image as the f32 the observation pass
    // would encode it to (the wave's LDS table, filled at the top of step_wave: the same expression) -- the env then needs no pass at all
    float *og = (BND && S.bound_gray) ? S.bound_gray + (size_t)e * S.hw : nullptr;
    const __attribute__((address_space(3))) float *lutg = (const __attribute__((address_space(3))) float *)(g_smem + STEP_LDS_LUT);
    auto put = [&](int idx, uint32_t g) {
        scr[idx] = (uint8_t)g;
        if constexpr (BND) { if (og) og[idx] = lutg[g & 0x7fu]; }
    };
    // (the lane's LDS columns from the lane id, here: a pointer carried from the top of the wave is one more spilled register to reload)
    int ln = threadIdx.x;
    asm volatile("" : "+v"(ln));
    const lds_u32 *mc = (const lds_u32 *)(g_smem + STEP_LDS_MC) + ln;
    const lds_u32 *ovl_l = (const lds_u32 *)(g_smem + STEP_LDS_OVL) + ln;
    const uint32_t ds = ovl_l[0] & 0x1ffffffu;
    const int px = E.px, py = E.py, rid = room_id_of(c, px, py);
    auto in_win = [&](int x, int y) { const int i = x - w.ox, j = y - w.oy; return i >= -2 && i <= 2 && j >= -2 && j <= 2; };
    auto dirty_at = [&](int x, int y) { return in_win(x, y) && ((ds >> WIN_K(x - w.ox, y - w.oy)) & 1u); };
    // (the two candidate rooms were fetched with the window: the one of the cell the turn started on, rid0, and the one of the cell the key pointed at)
    const int rsel = rid == rid0 ? 0 : 2;
    const uint32_t r_rect = ovl_l[(RG_OVL_MAX + 2 + rsel) * WAVE], r_meta = ovl_l[(RG_OVL_MAX + 3 + rsel) * WAVE] & 0xffu;
    const __attribute__((address_space(3))) uint8_t *mt = (const __attribute__((address_space(3))) uint8_t *)(ovl_l - ln + (RG_OVL_MAX + 6) * WAVE);
    // pass 1: per slot, what is to be done -- 3 bits each in one register (bit 0 restore the old cell, bit 1 draw the monster, bit 2 it shows now); loops,
    // not unrolled code with register arrays: this runs between the monsters' turn and the tail, where every register it takes is one the allocator spills --
    // and a spill's reload is a LOAD: it waits for every store the turn has issued (measured: five reloads, 5 us per wave)
    uint32_t actp = 0;
    bool far = false;
#pragma nounroll
    for (int s0 = 0; s0 < nrooms; s0++) {
        const uint32_t p = ovl_l[(1 + s0) * WAVE], m = mc[s0 * WAVE];
        const bool alive = (m >> 24) & MF_ALIVE, had = p != OVL_NONE;
        const uint32_t pp = p & 0xff3fu, pn = m & 0xffffu;
        bool show = false;
        if (alive) {
            const int x = POS_X(pn), y = POS_Y(pn), dx = px - x, dy = py - y;
            show = dx * dx + dy * dy <= 2;
            if (!show && rid >= 0 && room_id_of(c, x, y) == rid) {  // Floor::in_same_room (floor.rs:381-393)
                if ((r_meta & RM_KIND_MASK) == RK_EMPTY) show = true;
                else {
                    int x0, y0, x1, y1;
                    unpack_rect(r_rect, x0, y0, x1, y1);
                    const bool ina = px >= x0 && px < x1 && py >= y0 && py < y1, inb = x >= x0 && x < x1 && y >= y0 && y < y1;
                    show = ina == inb;
                }
            }
        }
        const bool unknown = had && (p & OVL_UNKNOWN), showed = had && (p & OVL_SHOW);
        const bool same = had && alive && pp == pn && !unknown && !dirty_at(POS_X(pn), POS_Y(pn));
        uint32_t a = 0;
        if (same) { if (showed && !show) a = 1u; else if (!showed && show) a = 2u; }
        else { if (had && (showed || unknown)) a |= 1u; if (show) a |= 2u; }
        a |= show ? 4u : 0u;
        actp |= a << (3 * s0);
        if (((a & 1u) && !in_win(POS_X(pp), POS_Y(pp))) || ((a & 2u) && !in_win(POS_X(pn), POS_Y(pn)))) far = true;
        // where the overlay stands now and how it shows -- right whoever draws this Redraw (below, or the observation pass from the tiles)
        const uint32_t now = alive ? (pn | (show ? OVL_SHOW : 0u)) : OVL_NONE;
        if (now != p) S.ovl[(size_t)s0 * n + e] = (uint16_t)now;
    }
    const uint32_t pl = ovl_l[(1 + nrooms) * WAVE];
    const uint32_t plp = pl & 0xff3fu;
    const bool pl_moved = pl == OVL_NONE || (pl & OVL_UNKNOWN) || plp != POS(px, py) || dirty_at(px, py);
    if (pl_moved) S.ovl[(size_t)nrooms * n + e] = (uint16_t)POS(px, py);
    if (far || (pl_moved && pl != OVL_NONE && !in_win(POS_X(plp), POS_Y(plp)))) return false;
    // 1. the cells the turn wrote back: what they show by themselves (overlays on them are redrawn below: such a slot is never `same`)
    for (uint32_t d = ds; d;) {
        const int k = __ffs((int)d) - 1;
        d &= d - 1;
        const int j = k / 5, i = k - j * 5, x = w.ox + i - 2, y = w.oy + j - 2;
        put(y * W + x, base_glyph(c, WV(w, k), y) & 0x7fu);
    }
    if (react & R_HIST_CHANGED) S.hist[(size_t)e * S.hw + py * W + px] = 1;  // (the one way a cell becomes VISITED: the player steps on it, move_player)
    // 2. the player's old cell, the cells monsters left or stopped showing on
    if (pl_moved && pl != OVL_NONE) {
        const int x = POS_X(plp), y = POS_Y(plp);
        put(y * W + x, base_glyph(c, WV(w, WIN_K(x - w.ox, y - w.oy)), y) & 0x7fu);  // (one turn: the old cell is the window's centre or next to it)
    }
#pragma nounroll
    for (int s0 = 0; s0 < nrooms; s0++)
        if ((actp >> (3 * s0)) & 1u) {
            const uint32_t pp = ovl_l[(1 + s0) * WAVE] & 0xff3fu;
            const int x = POS_X(pp), y = POS_Y(pp);
            put(y * W + x, base_glyph(c, WV(w, WIN_K(x - w.ox, y - w.oy)), y) & 0x7fu);
        }
    // 3. the monsters that (newly) show: draw priority monster < gold < player (core/src/lib.rs:271-283)
#pragma nounroll
    for (int s0 = 0; s0 < nrooms; s0++)
        if ((actp >> (3 * s0)) & 2u) {
            const uint32_t m = mc[s0 * WAVE];
            const int x = POS_X(m), y = POS_Y(m);
            const uint32_t under = base_glyph(c, WV(w, WIN_K(x - w.ox, y - w.oy)), y);
            if ((under & 0x80u) && under != (0x80u | '*')) put(y * W + x, mt[(m >> 16) & 0xff]);
        }
    // 4. the player
    if (pl_moved && (base_glyph(c, WV(w, WIN_K(px - w.ox, py - w.oy)), py) & 0x80u)) put(py * W + px, '@');
    return true;
}

// ---------------------------------------------------------------------------------------------
// k_step: one key for every env
// ---------------------------------------------------------------------------------------------
// ThreadConductor's auto-reset (thread_impls.rs:69-79) for the lanes whose spare level-1 state is ready: spare -> live, memory to memory, at the
// end of the wave.  Everything is requested before anything is stored, so the whole reset costs the wave about one memory round trip (the
// round-1 form -- scalars into registers, tables, then the grid, each waiting for the one before -- cost ~25 us of dependent round trips in
// every wave with a terminal lane, a third of the waves of a step).  The spare's pointers come from a device-resident RgState: one kernel
// argument instead of a second 60-pointer struct held in SGPRs by a kernel that is already spilling them.
__device__ __forceinline__ void take_spares(const RgState &S, const RgState *__restrict__ SPd, const RgConfig &c, int lane, int e, bool taken, bool &on_stairs) {
    const uint64_t tm = __ballot(taken);
    if (!tm) return;
    const RgState &SP = *SPd;
    const int HW = S.hw, n = S.n, nrooms = c.room_num_x * c.room_num_y;
    // the spare taken: the first of the env's spares that is ready (rg_state.h sp_slots; a slot only ever leaves READY through this env's own take, so
    // one is); es = its index in the spare view, whose SoA stride is ns
    const size_t ns = (size_t)n * S.sp_slots;
    int es = e;
    if (taken)
        for (int sl = S.sp_slots - 1; sl >= 0; sl--)
            if (__hip_atomic_load(&S.sp_ready[(size_t)sl * n + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) es = sl * n + e;
    // The acquire BEHIND the flag loads that pick the slot and in front of every payload load: pairs with the producers' hand-off (sc1 payload, drained, then
    // sp_ready = 1).  In front of the lookup (round 5) a slot that turned READY between the fence and its flag load could be picked and its payload -- whose
    // SoA lines it shares with its neighbours' -- read from a line cached before the hand-off.  Measured free (round 4).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // grids: the wave streams each taken env's 2 * HW bytes with 16-byte accesses (mini: one access per lane and env), FOUR envs per round: all their
    // loads are in flight before the first store, so a wave with several terminal lanes (the episodes of a batch created together end together) pays one
    // memory round trip per four resets.  (Round 3's slowest waves spent 15-20 us here, two envs per round.  Spelled out without arrays: the array form
    // put the capped kernel's copies into scratch memory.)
    if ((HW & 7) == 0) {
        const int q = HW / 8;
        uint64_t mm = tm;
        while (mm) {
            const int s0 = __ffsll((long long)mm) - 1; mm &= mm - 1;
            const int s1 = mm ? __ffsll((long long)mm) - 1 : -1; if (mm) mm &= mm - 1;
            const int s2 = mm ? __ffsll((long long)mm) - 1 : -1; if (mm) mm &= mm - 1;
            const int s3 = mm ? __ffsll((long long)mm) - 1 : -1; if (mm) mm &= mm - 1;
            const int e0 = __shfl(e, s0), e1 = __shfl(e, s1 >= 0 ? s1 : s0), e2 = __shfl(e, s2 >= 0 ? s2 : s0), e3 = __shfl(e, s3 >= 0 ? s3 : s0);
            const int q0 = __shfl(es, s0), q1 = __shfl(es, s1 >= 0 ? s1 : s0), q2 = __shfl(es, s2 >= 0 ? s2 : s0), q3 = __shfl(es, s3 >= 0 ? s3 : s0);
            const uint4 *a0 = reinterpret_cast<const uint4 *>(SP.cell + (size_t)q0 * HW), *a1 = reinterpret_cast<const uint4 *>(SP.cell + (size_t)q1 * HW);
            const uint4 *a2 = reinterpret_cast<const uint4 *>(SP.cell + (size_t)q2 * HW), *a3 = reinterpret_cast<const uint4 *>(SP.cell + (size_t)q3 * HW);
            uint4 *d0 = reinterpret_cast<uint4 *>(S.cell + (size_t)e0 * HW), *d1 = reinterpret_cast<uint4 *>(S.cell + (size_t)e1 * HW);
            uint4 *d2 = reinterpret_cast<uint4 *>(S.cell + (size_t)e2 * HW), *d3 = reinterpret_cast<uint4 *>(S.cell + (size_t)e3 * HW);
            for (int i = lane; i < q; i += WAVE) {
                const uint4 v0 = a0[i];
                uint4 v1 = v0, v2 = v0, v3 = v0;
                if (s1 >= 0) v1 = a1[i];
                if (s2 >= 0) v2 = a2[i];
                if (s3 >= 0) v3 = a3[i];
                d0[i] = v0;
                if (s1 >= 0) d1[i] = v1;
                if (s2 >= 0) d2[i] = v2;
                if (s3 >= 0) d3[i] = v3;
            }
        }
    } else {
        uint64_t mm = tm;
        while (mm) {
            const int src = __ffsll((long long)mm) - 1; mm &= mm - 1;
            const int env_s = __shfl(e, src);
            const uint16_t *sp = SP.cell + (size_t)__shfl(es, src) * HW;
            uint16_t *dp = S.cell + (size_t)env_s * HW;
            for (int i = lane; i < HW; i += WAVE) dp[i] = sp[i];
        }
    }
    if (taken) {
        uint32_t r[12];
#pragma unroll
        for (int k = 0; k < 12; k++) r[k] = SP.rng[k * ns + es];
        const uint16_t pp = SP.p_pos[es];
        const int32_t hp = SP.p_hp[es], hpm = SP.p_hpmax[es], lv = SP.p_lvl[es];
        const uint32_t ex = SP.p_exp[es], fd = SP.food[es], qu = SP.quiet[es], pg = SP.pack_gold[es], dl = SP.dlevel[es], mc = SP.mon_cnt[es];
        on_stairs = SP.on_stairs[es] != 0;
#pragma unroll
        for (int k = 0; k < 12; k++) S.rng[k * n + e] = r[k];
        S.p_pos[e] = pp; S.p_hp[e] = hp; S.p_hpmax[e] = hpm; S.p_lvl[e] = lv;
        if (S.obs_rec) S.obs_rec[(size_t)e * RG_OBS_REC_WORDS(nrooms) + nrooms] = pp;
        if (S.ovl) S.ovl[(size_t)nrooms * n + e] = (uint16_t)(pp | 0x40u);
        S.p_exp[e] = ex; S.food[e] = fd; S.quiet[e] = qu; S.pack_gold[e] = pg; S.dlevel[e] = dl; S.mon_cnt[e] = mc;
        for (int s0 = 0; s0 < nrooms; s0 += 4) {  // tables, four slots per round
            uint32_t rr[4], mw[4], me[4], gp[4], ga[4]; int32_t mh[4]; uint8_t rm[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t g = (size_t)(s0 + k < nrooms ? s0 + k : s0) * ns + es;
                rr[k] = SP.room_rect[g]; rm[k] = SP.room_meta[g]; mw[k] = SP.mon_w0[g]; mh[k] = SP.mon_hp[g]; me[k] = SP.mon_exp[g]; gp[k] = SP.gold_pos[g]; ga[k] = SP.gold_amt[g];
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (s0 + k < nrooms) {
                    const size_t g = (size_t)(s0 + k) * n + e;
                    S.room_rect[g] = rr[k]; S.room_meta[g] = rm[k]; S.mon_w0[g] = mw[k]; S.mon_hp[g] = mh[k]; S.mon_exp[g] = me[k]; S.gold_pos[g] = gp[k]; S.gold_amt[g] = ga[k];
                    if (S.obs_rec) {  // the env's observation record follows its tables (rg_state.h obs_rec)
                        uint32_t *rec = S.obs_rec + (size_t)e * RG_OBS_REC_WORDS(nrooms);
                        rec[s0 + k] = mw[k]; rec[nrooms + 1 + s0 + k] = rr[k]; reinterpret_cast<uint8_t *>(&rec[2 * nrooms + 1])[s0 + k] = rm[k];
                    }
                    if (S.ovl) S.ovl[(size_t)(s0 + k) * n + e] = (uint16_t)(((mw[k] >> 24) & MF_ALIVE) ? ((mw[k] & 0xffffu) | 0x40u) : 0xffffu);  // (| OVL_UNKNOWN)
                }
        }
    }
    __syncthreads();  // every lane's reads of the spares are complete (vmcnt drained) ...
    // ... before k_regen may refill them.  (ROGUE_GYM_HIP_KEEP_SPARES: an env with a fixed seed rebuilds the SAME level-1 state at every reset --
    // GameConfig::build is a pure function of config and seed, core/src/lib.rs:193-228 -- so its spare stays valid and is left in place.)
    // (a RELAXED store: what must precede it is that the spare has been READ, which the barrier above guarantees (it drains vmcnt) -- nothing this wave
    // wrote has to be visible to k_regen.  The release store of rounds 2-3 was a buffer_wbl2, the write-back of the XCD's whole L2, in a third of the
    // waves of every launch.)
    if (taken && !(S.keep_spares && S.reseed[e] == 0)) __hip_atomic_store(&S.sp_ready[es], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave's share of a step: lane i plays the key of env `e` (any env index -- the lanes of a wave need not hold consecutive envs), `valid`
// lanes only; the other lanes still take part in the wave-cooperative services.
template <int BW, int GM, bool BND = false>
__device__ __forceinline__ void step_wave(const RgState &S, const RgState *__restrict__ SPd, const RgConfig &c, const uint8_t *__restrict__ keys, int use_spares, int stage_off,
                                          const int e, const bool valid_in, const int stair_role) {
    // stair_role: 0 = no stair isolation, 1 = this wave serves listed (on-stairs) envs, 2 = index-order wave: listed envs are somebody else's
    uint16_t *lds_grid = reinterpret_cast<uint16_t *>(g_smem + stage_off);  // (behind the per-lane columns: STEP_LDS_*)
    const int lane = threadIdx.x;
    Prof pf; pf.start(S.prof);
    Env E;
    E.err = 0; E.on_stairs = 0;
    uint32_t react = 0, err = 0, old_flags = 0, steps = 0, flags = 0;
    int act = ACT_NOOP, dir = 0;
    int gold0 = 0;       // the status mirror's gold before this key
    bool live = false;   // this lane processes a key this call
    bool ui_dead = false, terminal = false;
    bool inc_done = false;  // the turn applied its Redraw to the screen mirror itself (mirror_update)
    bool taken = false;  // terminal + auto-reset + spare ready: the spare becomes the live state at the end of the wave (take_spares)
    uint32_t n_bfs = 0, n_inline = 0, n_taken = 0, n_cont = 0;  // workload counters (S.stats)
    uint32_t key = 0, nxs = RG_NX_NONE;
    bool listed = false;  // the env is in the stair set this launch reads (its player stands on the stairs)
    // LDS monster cache: column `lane` of [nrooms][64] words behind the generation / BFS staging area
    const int nrooms_k = c.room_num_x * c.room_num_y;
    E.mc = (lds_u32 *)(g_smem + STEP_LDS_MC) + lane;
    const uint32_t glyph_r = S.ovl ? c.mon[lane & 31].tile : 0u;  // (the monster glyphs by type, for the wave's LDS table: requested with the first round of loads)
    if (valid_in) {
        // ONE round of independent loads: the step's inputs, the env's scalars and its monster words together.  (Whether the lane plays at all is
        // only known from the first few -- loading the env behind that decision was a second dependent round trip in every wave; a lane that turns
        // out to be somebody else's (stair_role 2), dead or past max_steps just drops what it loaded.)
        listed = S.stair_mark[(size_t)(S.stair_gen & 1) * S.n + e] != 0;
        uint32_t nxk = 0;  // the last word of the dungeon stream the env's next-level structure starts from
        // (k_regen moves these two words while this kernel runs: relaxed agent-scope loads, as on its side -- the protocol tolerates any value that was
        // current at some point since the previous launch, and an atomic load is what says so to the compiler and the caches)
        if (S.nx_state) {
            nxs = __hip_atomic_load(&S.nx_state[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nxk = __hip_atomic_load(&S.nx_rng[(size_t)3 * S.n + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old_flags = S.flags[e];
        steps = S.steps[e];
        gold0 = S.status[(size_t)e * 10 + 1];
        if (e < S.n_keys) key = keys[e];  // (an env beyond the key prefix has no key: key = 0, never '>')
        load_env(S, E, e);
        if (nxs == RG_NX_READY && nxk != E.rd.w) nxs = RG_NX_STALE;  // the stream has moved on since: no use for a descent; asked again when the stairs are near
        // the monster words, four slots per round into registers first: written as `E.mc[s * WAVE] = S.mon_w0[..]` in a loop the compiler emitted one
        // load -> wait -> LDS store per slot, i.e. nrooms SERIAL round trips in every wave (found in the ISA, round 4)
        for (int s0 = 0; s0 < nrooms_k; s0 += 4) {
            uint32_t m4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) m4[k] = S.mon_w0[(size_t)(s0 + k < nrooms_k ? s0 + k : s0) * S.n + e];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (s0 + k < nrooms_k) E.mc[(s0 + k) * WAVE] = m4[k];
        }
    }
    // An env is played by a stair wave iff its player stands on the stairs AND this key is '>' (the only way into a level generation); both
    // kinds of wave decide from the same two values -- the key and the env's byte of the stair set this launch READS, which nothing writes while
    // the launch runs (the set for the next launch is a different buffer) -- so exactly one of them takes the env, whenever it starts.  A listed
    // env with any other key costs its stair wave one round of loads.
    // Parked in LDS until the end of the turn: the step count, the old flags and the status mirror's gold -- three registers held through every service
    // of the turn by a kernel that has none to spare (the capped instance spilled them to scratch memory instead)
    const bool to_stair_wave = listed && key == '>';
    const bool valid = valid_in && (stair_role == 0 || (stair_role == 1) == to_stair_wave);
    const bool has_key = valid && e < S.n_keys;  // ThreadConductor::step zips keys with envs (thread_impls.rs:62-64)
    if (valid) {
        if (has_key && !(steps > c.max_steps)) {  // state_impls.rs:52-54
            act = decode_key(key, dir);
            if (act == ACT_INVALID) err = RG_FLAG_ERR_KEY;            // ErrorKind::InvalidInput: not mapped, not logged
            else {
                klog_push(S, e, key);                                  // saved_inputs.push precedes the modal check (core/src/lib.rs:288)
                if (old_flags & RG_FLAG_DEAD) err = RG_FLAG_ERR_DEAD; // Grave modal + InputCode::Act => IgnoredInput
                else live = true;
            }
        }
    }
    // Action::DownStair (actions.rs:27-36): the new level is produced by the generation service below
    pf.mark(0);
    bool need_gen = false, descends = false;
    Win w;  // the 5x5 tiles around the player: one round of loads serves the whole player action
    w.v = (lds_u16 *)(g_smem + STEP_LDS_WIN) + lane;
    // [0]: the dirty mask of the turn's window write-back (bit 31: a whole-room fill happened), [1 .. RG_OVL_MAX + 1]: where overlays stood at the env's last
    // Redraw (S.ovl) -- parked here from the first load round to the incremental mirror update at the end of the turn
    lds_u32 *ovl_l = (lds_u32 *)(g_smem + STEP_LDS_OVL) + lane;
    ovl_l[0] = 0;
    if (BND && S.bound_gray) {
        // glyph -> gray value as the observation pass encodes it (rg_obs.hip k_obs `lutf`, python/src/lib.rs:84), for the pixels mirror_update writes into the bound
        // tensor: the host's table straight into LDS (computed per wave -- two tile_to_sym + two IEEE divisions per lane -- it cost every wave 1.1 us)
        typedef const __attribute__((address_space(1))) void *gptr;
        typedef __attribute__((address_space(3))) void *lptr;
        __builtin_amdgcn_global_load_lds((gptr)(S.gray_lut + lane), (lptr)(g_smem + STEP_LDS_LUT), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(S.gray_lut + 64 + lane), (lptr)(g_smem + STEP_LDS_LUT + 256), 4, 0, 0);
    }
    if (S.ovl && lane < RG_MAX_ENEMY_KINDS + 6)
        ((__attribute__((address_space(3))) uint8_t *)(g_smem + STEP_LDS_OVL + (RG_OVL_MAX + 6) * WAVE * 4))[lane] = (uint8_t)glyph_r;
    if (S.ovl) {
        // where the overlays of the env's screen mirror stand, straight into LDS (global_load_lds: no register, no wait -- they are read at the end of the
        // turn).  Unconditional for every lane and slot (an idle lane reads env 0's): LDS-DMA calls under divergent control flow are what the compiler
        // merges into one instruction with a per-lane M0 (profiles/r05_experiments.txt).
        typedef const __attribute__((address_space(1))) void *gptr;
        typedef __attribute__((address_space(3))) void *lptr;
        lds_u32 *base = (lds_u32 *)(g_smem + STEP_LDS_OVL);
        const size_t es = valid_in ? (size_t)e : 0;
#pragma unroll
        for (int s0 = 0; s0 <= RG_OVL_MAX; s0++)
            __builtin_amdgcn_global_load_lds((gptr)(S.ovl + (size_t)(s0 <= nrooms_k ? s0 : 0) * S.n + es), (lptr)(base + (1 + s0) * WAVE), 2, 0, 0);
    }
    // (a stair wave's lanes press '>' on the staircase by construction: nothing of the old level is looked at again, so no window)
    w.inb = 0;
    if (S.ovl) {
        // With the window: kind and rect of the room(s) the player can stand in after this key -- the cell he is on and the one the key points at -- straight
        // into LDS (the end of the turn picks; Floor::in_same_room), and the monster glyphs by type.  Nothing the mirror update needs is loaded at the
        // end of the turn: a load there waits for every store the turn has issued (vmcnt is in order): measured 6 us per wave.
        typedef const __attribute__((address_space(1))) void *gptr;
        typedef __attribute__((address_space(3))) void *lptr;
        lds_u32 *base = (lds_u32 *)(g_smem + STEP_LDS_OVL);
        const size_t es = valid_in ? (size_t)e : 0;
        const bool mv = live && (act == ACT_MOVE || act == ACT_MOVE_UNTIL);
        const int ra = room_id_of(c, E.px, E.py), rb = mv ? room_id_of(c, E.px + dir_dx(dir), E.py + dir_dy(dir)) : ra;
        const size_t aa = (size_t)(ra >= 0 ? ra : 0) * S.n + es, ab = (size_t)(rb >= 0 ? rb : 0) * S.n + es;
        __builtin_amdgcn_global_load_lds((gptr)(S.room_rect + aa), (lptr)(base + (RG_OVL_MAX + 2) * WAVE), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(S.room_meta + aa), (lptr)(base + (RG_OVL_MAX + 3) * WAVE), 1, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(S.room_rect + ab), (lptr)(base + (RG_OVL_MAX + 4) * WAVE), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(S.room_meta + ab), (lptr)(base + (RG_OVL_MAX + 5) * WAVE), 1, 0, 0);
    }
    if (live && stair_role != 1) win_load(c, E.cell, w, E.px, E.py);
    else { w.inb = 0; w.dirty = 0; w.ox = w.oy = 0; }
    pf.mark(26);
    if (live && act == ACT_DOWNSTAIR) {
        if (stair_role == 1 || (WV(w, WIN_K(0, 0)) & C_SURF_MASK) == S_STAIR) {
            need_gen = descends = true;
            react |= R_REDRAW | R_STATUS | R_HIST_STALE;  // Redraw precedes StatusUpdated: history keeps the old level
        } else react |= MSG_NO_DOWNSTAIR;
    }
    // Next-level structure (gen_service): a descending lane whose structure is READY and starts from the dungeon stream the env holds now loads it
    // instead of generating the first two thirds of the level
    if (S.nx_state) {
        if (descends && nxs == RG_NX_READY) {  // (only a real descent pays the round trip: every wave has a lane that presses '>' somewhere off the stairs)
            const uint32_t *q = S.nx_rng + e;
            const size_t n = (size_t)S.n;
            const uint32_t k0 = q[0], k1 = q[n], k2 = q[2 * n], kl = S.nx->level[e];  // (the fourth word was compared when the env was loaded)
            if (k0 == E.rd.x && k1 == E.rd.y && k2 == E.rd.z && kl == E.dlevel) nxs = RG_NX_HIT;
        }
        if (__any(nxs == RG_NX_HIT)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // pairs with k_regen's hand-off (sc1 payload, drained, then nx_state = READY)
    }
    {   // Two descents in a row: the first one's Redraw kept the older level's history (HIST_STALE), so the mirror never showed the level that
        // is about to be discarded -- but the reference shows exactly that level's history after the second descent.  Write it now.
        uint64_t lag = __ballot(descends && ((old_flags & RG_FLAG_HIST_LAG) || ((old_flags & RG_FLAG_REDRAW) && (old_flags & RG_FLAG_HIST_STALE))));
        while (lag) {
            const int src = __ffsll((long long)lag) - 1;
            lag &= lag - 1;
            const int env_s = __shfl(e, src);
            const uint16_t *cp = S.cell + (size_t)env_s * S.hw;
            uint8_t *hp = S.hist + (size_t)env_s * S.hw;
            for (int i = lane; i < S.hw; i += WAVE) hp[i] = (cp[i] & C_VISITED) ? 1 : 0;
        }
    }
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: levels for descending lanes; pass 1: rebuilds for terminal lanes (ThreadConductor auto-reset)
        pf.mark(1);
        if (pass == 1 && use_spares) {
            // The env's pre-generated spare (k_regen) IS its post-reset state: if it is ready, nothing is generated here, and nothing of it is
            // needed in registers either -- the lane only notes `taken`; the spare is moved into place at the very end of the wave in one
            // batched memory-to-memory copy (take_spares).  Without a ready spare the level is generated inline below.
            if (need_gen) {  // (which of the env's spares it will be is looked up again by take_spares: no register carries it through the wave's tail)
                bool rdy = false;
                for (int sl = 0; sl < S.sp_slots; sl++) rdy = rdy || __hip_atomic_load(&S.sp_ready[(size_t)sl * S.n + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
                if (rdy) { taken = true; need_gen = false; n_taken++; }
            }
        }
        const bool regenerated = descends && pass == 0;
        if constexpr (BW == 1 || BW == 2) { if (pass == 0 && S.dc_walk) snapshot_walk_service(S, c, lane, e, descends); }
        const uint64_t nx_now = pass == 0 ? __ballot(need_gen && nxs == RG_NX_HIT) : 0ull;
        if (need_gen) n_inline++;
        gen_service<GM>(S, c, E, lane, e, need_gen, pass == 1, lds_grid, pf, nx_now);
  // k_step_w32 carries the 32-room generator (rgk_step routes by room count too), the wider instances the 64-room one
        (void)regenerated;  // (a descended lane's monster-cache column was refilled by gen_service from the generator's own table)
        pf.mark(2);
        need_gen = false;
        if (pass == 1) break;

        bool running = live && act != ACT_NOOP;
        int iter = 0;
        while (__any(running)) {
            bool do_turn = false, need_bfs = false, opened = false, own_req = false;
            int map_slot = -1;
            FillReq fr; fr.leave = fr.enter = 0;
            if (running) {
                switch (act) {  // actions::process_action (actions.rs:16-65)
                case ACT_DOWNSTAIR:
                    do_turn = true; running = false;
                    break;
                case ACT_MOVE:
                case ACT_MOVE_UNTIL: {
                    bool done = move_player(S, c, E, w, dir, react, fr, (S.ovl && iter == 0) ? (const lds_u32 *)ovl_l : nullptr);
                    if (act == ACT_MOVE) { do_turn = true; running = false; break; }
                    uint32_t v = win_get(w, WIN_K(E.px - w.ox, E.py - w.oy));
                    uint32_t tile = (v & C_VISIBLE) ? glyph_of(v) : ' ';
                    if (done || (tile != '.' && tile != '#')) running = false;  // MoveUntil stops without after_turn
                    else do_turn = true;
                    break;
                }
                case ACT_SEARCH:
                    opened = do_search(c, E, w, react);
                    do_turn = true; running = false;
                    break;
                }
            }
            if constexpr (BW == 1 || BW == 2) { if (__any(opened) && S.dc_walk) snapshot_walk_service(S, c, lane, e, opened); }
            pf.mark(28);
            if (do_turn) turn_passed(c, E, react);  // actions::after_turn (actions.rs:67-80)
            pf.mark(29);
            bool need_map = false;
            if (do_turn && E.mon_active > 0) need_map = monsters_prepass(S, c, E);
            pf.mark(30);
            if (need_map) {
                uint32_t part_bits = 0, own_bits = 0;
                if constexpr (BW == 1 || BW == 2) { part_bits = S.dc_part[e]; own_bits = S.dc_own[e]; }  // (requested with the ring's keys: one round trip, not three)
                need_bfs = !dist_cache_lookup(S, E, POS(E.px, E.py), map_slot);
                if constexpr (BW == 1 || BW == 2) {
                    // a cached PARTIAL map that does not reach one of this turn's chasers is continued (bfs_rows_n32)
                    if (!need_bfs && ((part_bits >> map_slot) & 1u) && !dist_map_sufficient(S, c, E, map_slot)) {
                        need_bfs = true;
                        own_req = (own_bits >> map_slot) & 1u;
                        if (own_req) n_cont++;
                    }
                }
            }
            pf.mark(3);
            // whole-room reveals / hides by the wave, then the lanes' changed window cells (the final word on those cells); both before any
            // monster or BFS read of the grid
            // (for the incremental mirror update.  A lane's own turn is the loop's FIRST iteration unless it runs (MoveUntil); the later iterations -- driven by some
            // other lane's run -- must not overwrite what it parked)
            if (iter == 0) ovl_l[0] = w.dirty | ((fr.leave | fr.enter) ? 0x80000000u : 0u);
            fill_service(S, c, lane, e, fr);
            win_flush(c, E.cell, w);
            pf.mark(27);
#ifdef RG_EXP_NO_BFS
            need_bfs = false;  // (experiment build only -- wrong pictures: what the launch would cost with every dist map built elsewhere; profiles/r06_experiments.txt)
#endif
            uint64_t m = __ballot(need_bfs);
            if (need_bfs) n_bfs++;
            if (m) {  // serve the requesting lanes with the whole wave, several maps per round
                unsigned long long tb0 = pf.p ? __builtin_amdgcn_s_memtime() : 0;
                const bool complete = bfs_service<BW>(S, c, reinterpret_cast<uint64_t *>(lds_grid), m, e, E.px, E.py, map_slot, lane, E.mc - lane, __ballot(own_req));
                if constexpr (BW == 1 || BW == 2) {
                    if (need_bfs) {  // a new map starts from the level's own cells; a complete one needs no saved mask any more
                        const uint32_t bit = 1u << map_slot;
                        const uint32_t part = S.dc_part[e], own = S.dc_own[e];
                        S.dc_part[e] = (uint16_t)(complete ? part & ~bit : part | bit);
                        S.dc_own[e] = (uint16_t)((own_req && !complete) ? own : own & ~bit);
                    }
                } else (void)complete;
                if (pf.p) { pf.rec(24, __builtin_amdgcn_s_memtime() - tb0); pf.rec(25, (unsigned long long)__popcll(m)); }
            }
            pf.mark(4);
            if (do_turn && E.mon_active > 0) ui_dead = monsters_move(S, c, E, map_slot, react);  // `ui` of the LAST after_turn wins
            else if (do_turn) ui_dead = false;
            pf.mark(5);
            // a MoveUntil run moves one cell in a fixed direction per iteration => it ends after at most max(W, H) iterations
            if (++iter > RG_MAX_W + RG_MAX_H && running) { running = false; E.err |= RG_FLAG_ERR_INTERNAL; }
            if (running) win_load(c, E.cell, w, E.px, E.py);  // a MoveUntil run continues from the new cell
        }
        if (live) {
            // GameStateImpl::react's reaction loop (state_impls.rs:56-78)
            flags = (react & 0x7f00u);                       // message flags of this key only
            if (react & R_REDRAW) flags |= RG_FLAG_REDRAW | ((react & R_HIST_STALE) ? RG_FLAG_HIST_STALE : 0);
            else flags |= old_flags & (RG_FLAG_REDRAW | RG_FLAG_HIST_STALE);
            flags |= old_flags & (RG_FLAG_HIST_LAG | RG_FLAG_HIST_DIRTY | (BND ? RG_FLAG_SCR_CHANGED : 0u));
            if ((react & R_HIST_CHANGED) || descends) flags |= RG_FLAG_HIST_DIRTY;
            if (react & R_STATUS) write_status(S, c, E);
            if (ui_dead) flags |= RG_FLAG_DEAD;
            steps += 1;
            terminal = (react & R_GRAVE) || steps >= c.max_steps;
            need_gen = terminal && c.auto_reset;  // ThreadConductor::step (thread_impls.rs:69-79)
            // ---- the screen mirror kept current by the turn itself (VERDICT r4 task 4a) ----
            // An ordinary Redraw -- one turn, no whole-room reveal, no new level, history plane in step -- changes the screen in two places only: the window
            // cells the turn wrote back, and the overlays (monsters, player: where they stood at the last Redraw -- S.ovl -- and where they stand now).  The
            // lane writes those bytes of the mirror (and of the history plane) itself and raises no Redraw: the observation pass then treats the env like
            // the 57 % that did not redraw -- mirror -> f32, no tile decode, no overlay phases, no mirror write-back (its Redraw path is issue-bound, not
            // byte-bound: profiles/r05_experiments.txt).  Everything else keeps the Redraw flag and is drawn from the tiles as before.
            // (one turn per key = every action but a run: a per-LANE property -- one running lane in the wave no longer sends the other 63 to the tile-drawn Redraw)
            if (S.ovl && (react & R_REDRAW) && !descends && !need_gen && act != ACT_MOVE_UNTIL &&
                !(old_flags & (RG_FLAG_REDRAW | RG_FLAG_HIST_STALE | RG_FLAG_HIST_LAG | RG_FLAG_HIST_DIRTY)) && !(ovl_l[0] >> 31)) {
                // (SCR_CHANGED: bytes of the mirror changed without a Redraw flag -- what a bound observation tensor, rg_obs_bind, re-encodes this env for)
                // (... unless the update wrote the image's pixels too: a bound gray tensor)
                if (mirror_update<BND>(S, c, E, w, react, room_id_of(c, w.ox, w.oy))) flags = (flags & ~(RG_FLAG_REDRAW | RG_FLAG_HIST_DIRTY)) | ((!BND || S.bound_gray) ? 0u : RG_FLAG_SCR_CHANGED);
                inc_done = true;  // (the overlays' positions and how they show are recorded either way)
            }
        }
    }
    pf.mark(6);
    // The tail addresses everything from an OPAQUE copy of the env index: the load round at the top of the wave and the stores down here use the same
    // `field + e` addresses, and the compiler otherwise keeps those 64-bit per-lane addresses (two registers each, a dozen of them) alive through the whole
    // turn -- i.e. spills them, and a spill's reload down here is a LOAD that waits for every store of the turn.  Recomputed in place: two VALU ops each.
    int et = e;
    asm volatile("" : "+v"(et));
    E.e = et;
    // A bound observation tensor's work list (rg_state.h obs_list): this wave's slots are taken HERE -- one returning atomic per wave -- and filled at the very
    // end: its return value is first needed behind all the stores of the tail, and vmcnt being in order it is there by then (asked for at the end, the wave
    // waited 2.6 us for it: it returns behind every store issued before it).
    bool pend = false; uint64_t pm = 0; uint32_t lbase = 0;
    if (BND && S.obs_list) {
        const uint32_t fw = (live ? ((terminal && c.auto_reset) ? RG_FLAG_REDRAW : flags) : old_flags);
        pend = valid && (fw & (RG_FLAG_REDRAW | RG_FLAG_SCR_CHANGED));
        pm = __ballot(pend);
        if (pm && lane == 0) lbase = atomicAdd(&S.obs_cnt[S.obs_par], (uint32_t)__popcll(pm));
    }
    if (S.stats) {
        // per-BLOCK rows (one atomicAdd per wave and counter on a SHARED 64-byte line -- 7 000 same-line atomics per launch -- cost the kernel 20 us,
        // measured in round 2).  On the block's own line a no-return atomic is a fire-and-forget add; the plain `+=` of rounds 2-3 was a load the
        // wave had to wait for before it could store and end (round 4).
        const uint32_t cnt[8] = {(uint32_t)__popcll(__ballot(live && terminal && c.auto_reset)), (uint32_t)__popcll(__ballot(descends)),
                                 wave_sum(n_bfs), wave_sum(n_inline), wave_sum(n_taken), (uint32_t)__popcll(__ballot(live && (react & R_REDRAW))),
                                 (uint32_t)__popcll(__ballot(live)), (BW == 1 || BW == 2) ? wave_sum(n_cont) : 0u};
        const uint32_t n_nx = (uint32_t)__popcll(__ballot(descends && nxs == RG_NX_HIT));  // [8]: descents that loaded their next-level structure
        int ln = threadIdx.x;
        asm volatile("" : "+v"(ln));  // (the counter's address from an opaque lane id: computed at the top of the kernel it was spilled, and its reload here waited for every store of the turn)
        if (ln < RG_STAT_COLS) {
            uint32_t mine = ln == 8 ? n_nx : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) mine = ln == k ? cnt[k] : mine;
            if (mine) atomicAdd(&S.stats[(size_t)blockIdx.x * RG_STAT_COLS + ln], (unsigned long long)mine);  // (no return value: nothing waits for it)
        }
    }
    if (valid && err) {
        S.flags[et] = (old_flags & ~RG_FLAG_ERR_MASK) | err;
        S.reward[et] = 0.f;
        S.done[et] = (old_flags & RG_FLAG_TERMINAL) ? 1 : 0;
        atomicOr(S.err_any, err);
    } else if (valid && !live) {  // steps > max_steps / no key for this env: silent no-op
        S.reward[et] = 0.f; S.done[et] = (old_flags & RG_FLAG_TERMINAL) ? 1 : 0;
    } else if (valid) {
        if (terminal && c.auto_reset) {
            if (taken) {  // the status of a freshly built RunTime (GameConfig::build: level 1, Player::new + init_items, core/src/lib.rs:193-228)
                E.dlevel = 1; E.gold = c.init_gold; E.hp = E.hpmax = c.init_hp; E.plvl = 1; E.exp = 0; E.food = c.hunger_time;
            }
            write_status(S, c, E);
            S.dc_len[et] = 0; S.dc_head[et] = 0; S.dc_part[et] = 0; S.dc_own[et] = 0;  // a rebuilt RunTime owns a fresh DistCache
            steps = 0;
            flags = RG_FLAG_REDRAW | RG_FLAG_HIST_DIRTY;
            klog_new_episode(S, et);
        }
        if (E.err) { flags |= E.err; atomicOr(S.err_any, E.err); }
        if (terminal) flags |= RG_FLAG_TERMINAL;
        if (!taken) store_env(S, E);  // (a taken env's live scalars are the spare's: copied below)
        S.steps[et] = steps;
        S.flags[et] = flags;
        S.done[et] = terminal ? 1 : 0;
        // The env's observation record (rg_state.h obs_rec; rg_obs.hip ObsTabs): the fused observation pass overlays a Redraw from these words -- one line
        // per env instead of the env's column of the [slot][env] tables.  Here: the monsters as they stand after their turn (the wave's LDS table) and the
        // player; the room half is written with the level's tables (gen_service, take_spares -- which also writes a taken env's monsters and player).
        if (S.ovl && !taken && !inc_done && ((react & R_REDRAW) || descends || (terminal && c.auto_reset))) {
            // a Redraw the observation / render pass draws from the tiles: where the overlays stand as of it; how they show is not known here (OVL_UNKNOWN)
            for (int s0 = 0; s0 < nrooms_k; s0++) {
                const uint32_t mw = E.mc[s0 * WAVE];
                S.ovl[(size_t)s0 * S.n + et] = (uint16_t)(((mw >> 24) & MF_ALIVE) ? ((mw & 0xffffu) | OVL_UNKNOWN) : OVL_NONE);
            }
            S.ovl[(size_t)nrooms_k * S.n + et] = (uint16_t)(POS(E.px, E.py) | OVL_UNKNOWN);
        }
        if (S.obs_rec && (flags & RG_FLAG_REDRAW) && !taken) {
            uint32_t *rec = S.obs_rec + (size_t)et * RG_OBS_REC_WORDS(nrooms_k);
            for (int s0 = 0; s0 < nrooms_k; s0 += 4) {
                uint32_t m4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) m4[k] = E.mc[(s0 + k < nrooms_k ? s0 + k : s0) * WAVE];
                if (s0 + 4 <= nrooms_k) *reinterpret_cast<uint4 *>(rec + s0) = make_uint4(m4[0], m4[1], m4[2], m4[3]);
                else
                    for (int k = 0; k < 4 && s0 + k < nrooms_k; k++) rec[s0 + k] = m4[k];
            }
            rec[nrooms_k] = POS(E.px, E.py);
        }
        // reward = the gold delta of the status mirror (parallel.py:59-64).  The mirror's gold is E.gold wherever this key rewrote the status (a status
        // reaction, or the post-reset status), else what it was: known in registers -- reading it back from memory was a dependent round trip
        // (store -> load -> store, ~2 us) at the very end of every wave
        const int gold_before = gold0;
        const int gold_after = ((terminal && c.auto_reset) || (react & R_STATUS)) ? (int)E.gold : gold_before;
        // ... plus the stair bonus of StairRewardParallel (wrappers.py:45-64; rg_set_stair_reward): "the reported level is above the one reported a step earlier".
        // The reported level is the status mirror's, which follows E.dlevel (a descent is a status reaction; a reset rewrites it with level 1): it rises in
        // exactly the steps that descend and do not end in an auto-reset -- both known in registers, no level array, no extra pass.
        const float bonus = (descends && !(terminal && c.auto_reset)) ? S.stair_reward : 0.f;
        S.reward[et] = (float)(gold_after - gold_before > 0 ? gold_after - gold_before : 0) + bonus;
    }
    // the stair set for the NEXT k_step: where does this env's player stand now?  A level generated in this turn reported it (place_player), a taken
    // spare carries it, otherwise it is the tile under the player in the window (centred on where the last move started; the player is within one cell)
    bool on_next = false;
    if (valid && !live) on_next = S.stair_mark[(size_t)(S.stair_gen & 1) * S.n + et] != 0;  // (an env that did not play stands where it stood: its byte again, not a register held through the turn)
    if (live) on_next = ((descends || (terminal && c.auto_reset && !taken)) ? E.on_stairs != 0
                                                                              : (win_get(w, WIN_K(E.px - w.ox, E.py - w.oy)) & C_SURF_MASK) == S_STAIR);
    if (S.nx_state) {
        // a new level (descent, reset): nothing asked for it yet.  Else, the first time a staircase shows up next to the player: ASK for the next level's
        // structure (k_regen, beside the next step) -- or again, when the dungeon stream has left the one a READY structure starts from (an erratic
        // monster moved, a search found something).  The request carries the streams and the level it is for, written through and DRAINED before the
        // flag: k_regen works from this copy, never from the env's live state (which the k_step beside it may be replacing, its plain stores reaching
        // memory in any order).  k_regen touches the env's request and structure only between its claim (ASKED -> CLAIMED) and its publish / drop, and
        // an env asks only from NONE or READY -- a new level while the claim is out leaves DROP, not NONE (rg_state.h) -- so the two sides never read or
        // write these words at the same time.
        const bool fresh = live && (descends || (terminal && c.auto_reset));
        const bool ask = live && !fresh && (w.inb & WIN_STAIR) && (nxs == RG_NX_NONE || nxs == RG_NX_STALE);
        if (fresh && nxs != RG_NX_NONE) (void)__hip_atomic_fetch_and(&S.nx_state[et], RG_NX_DROP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // CLAIMED -> DROP, else -> NONE
        if (__any(ask)) {
            if (ask) {
                uint32_t *q = S.nx_rng + et;
                const size_t n = (size_t)S.n;
                st_pub<true>(&q[0], E.rd.x); st_pub<true>(&q[n], E.rd.y); st_pub<true>(&q[2 * n], E.rd.z); st_pub<true>(&q[3 * n], E.rd.w);
                st_pub<true>(&q[8 * n], E.ri.x); st_pub<true>(&q[9 * n], E.ri.y); st_pub<true>(&q[10 * n], E.ri.z); st_pub<true>(&q[11 * n], E.ri.w);
                st_pub<true>(&S.nx->level[et], E.dlevel);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ask) __hip_atomic_store(&S.nx_state[et], RG_NX_ASKED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (BND && pm) {
        const uint32_t lb = uni(lbase);  // (lane 0's: read with every lane active, not inside the `pend` branch -- readfirstlane takes the first ACTIVE lane)
        if (pend) S.obs_list[(size_t)S.obs_par * S.n + lb + lanes_below(pm)] = et;
    }
    take_spares(S, SPd, c, lane, et, taken, on_next);
    stair_publish(S, lane, et, valid, on_next);
    pf.mark(7);
    pf.finish();
}

// ---------------------------------------------------------------------------------------------
// k_step: one key for every env.
//
// Stair isolation.  The one thing in a step that takes far longer than everything else is a DESCENT: the new level is generated inside the turn
// (35-45 us by the whole wave, and the 63 other lanes of the wave wait for it) -- ~20 descents per 65 536-env step, and the launch lasts as
// long as its slowest wave (80-100 us with the descent inside a 64-env wave, against 30-50 us for every other wave).  A descent needs the player
// on the stairs, and whether he is there is known BEFORE the step: whoever moved the players last (the previous k_step, k_build, the debug descent)
// left a byte per env and the list of the marked envs (S.stair_mark / stair_list, ~240 of 65 536; rg_state.h).  The blocks at the FRONT of the grid look the listed envs' keys up: an env that presses
// '>' gets that wave for itself -- its turn + generation chain (55-70 us) starts at t = 0 and nobody waits for it -- and the index-order
// wave holding its lane skips it; the other listed envs stay with their index-order waves.  (Sorting the envs of a step into
// descent / awake-monster / plain waves with a classification kernel was tried first: neutral)
// ---------------------------------------------------------------------------------------------
#define STAIR_BLOCKS 256   // blocks at the front of the grid that serve the stair list (grid-stride beyond that)

// (the body is spelled out twice -- below for the capped W <= 32 instance -- rather than shared through a device function: routing the template through
// one more inlined call changed the allocation of the wider instances for the worse, 141 -> 153 us on the default 80x24 dungeon)
#define RG_STEP_BLOCK_BODY(BWV, GMV, BNDV) \
    __builtin_amdgcn_s_setprio(3); \
    const int lane = threadIdx.x; \
    if (blockIdx.x == 0 && lane == 0) { \
        stair_recycle(S); \
        if (BNDV && S.obs_cnt) S.obs_cnt[S.obs_par ^ 1] = 0;  /* (rg_state.h obs_list: the half the NEXT k_step appends to) */ \
        __hip_atomic_store(S.launch_mark, (uint32_t)S.stair_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  /* this launch has started: k_regen_gate */ \
    } \
    const bool stair = parity >= 0 && (int)blockIdx.x < STAIR_BLOCKS; \
    const int32_t *list = S.stair_list + (size_t)(S.stair_gen & 1) * S.n; \
    const int items = stair ? (int)S.stair_cnt[S.stair_gen % 3] : S.n; \
    const int first = stair ? (int)blockIdx.x : ((int)blockIdx.x - (parity >= 0 ? STAIR_BLOCKS : 0)) * epw; \
    uint32_t *next = S.stair_cnt + 4 + S.stair_gen % 3; \
    for (int i0 = first; i0 < items;) { \
        bool v; int e; \
        if (stair) { e = list[i0]; v = lane == 0 && e < S.n_keys && keys[e] == '>'; } \
        else { v = lane < epw && i0 + lane < items; e = v ? i0 + lane : 0; } \
        if (!stair || __any(v)) step_wave<BWV, GMV, BNDV>(S, SPd, c, keys, use_spares, stage_off, e, v, parity >= 0 ? (stair ? 1 : 2) : 0); \
        if (!stair || items <= STAIR_BLOCKS) break; \
        __syncthreads(); \
        uint32_t t = 0; \
        if (lane == 0) t = atomicAdd(next, 1u); \
        i0 = STAIR_BLOCKS + (int)uni(t); \
    }
template <int BW, bool BND = false>
__global__ void __launch_bounds__(WAVE) k_step(RgState S, const RgState *__restrict__ SPd, RgConfig c, const uint8_t *__restrict__ keys, int use_spares, int stage_off, int epw,
                                               int parity) {
    // this block's work: ONE call site of the turn code, whatever the role.  Stair block b looks at entry b of the stair list; entries beyond the first
    // STAIR_BLOCKS are handed out one at a time through a counter, so that a block busy with a descent (60 us) never has a second one queued behind
    // it while its neighbours sit idle.  An entry whose env does not press '>' stays with its index-order wave (step_wave's rule, applied here before
    // anything else of the env is loaded: such a block is gone in ~2 us).
    RG_STEP_BLOCK_BODY(BW, (BW != 0 ? 1 : 0), BND)
}
// The W <= 32 instance with the register allocation capped for TWO waves per SIMD (256 VGPRs).  With the 5x5 window in LDS it needs ~250: told to,
// the allocator fits it without a spill (left alone it lands on either side of the line from build to build).  Every block of a 65 536-env launch
// (1024 index-order + the stair blocks) is then resident from t = 0.  At one wave per SIMD the ~20 stair waves that really descend kept their SIMDs
// for the whole launch, as many index-order blocks started only when the first waves ended (33-38 us) and finished last (84 us against 57-67 us
// for every other wave).  The wider instances spill under the cap (15-136 VGPRs: a measured loss) and keep their natural allocation.
// (BND: the instance for handles with a bound observation tensor, rg_obs_bind -- the list of redrawn envs, the gray pixels of the incremental mirror update)
template <bool BND>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_step_w32(RgState S, const RgState *__restrict__ SPd, RgConfig c, const uint8_t *__restrict__ keys, int use_spares, int stage_off, int epw, int parity) {
    RG_STEP_BLOCK_BODY(0, 0, BND)
}
// More than 64 rooms (only possible on wide grids: 65 rooms need W >= 65): the generic row-width class with the 384-room generator.  Its LDS
// monster table (a column of `rooms` words per lane) exceeds the 64 KB default, see rgk_step.
__global__ void __launch_bounds__(WAVE) k_step_huge(RgState S, const RgState *__restrict__ SPd, RgConfig c, const uint8_t *__restrict__ keys, int use_spares, int stage_off,
                                                    int epw, int parity) {
    RG_STEP_BLOCK_BODY(4, 2, false)
}
#undef RG_STEP_BLOCK_BODY

// ---------------------------------------------------------------------------------------------
// host-callable launchers (used by rg_api.cpp)
// ---------------------------------------------------------------------------------------------
static int gen_mode_of(const RgConfig *c) { const int nr = c->room_num_x * c->room_num_y; return nr <= 32 ? 0 : (nr <= 64 ? 1 : 2); }
extern "C" {
void rgk_build(const RgState *S, const RgConfig *c, hipStream_t st) {
    int hw = c->width * c->height;
    size_t smem = GEN_SLOT_BYTES(hw, c->room_num_x * c->room_num_y);  // one level at a time per wave: one staging grid + the generator's tables
    const dim3 grid((S->n + BUILD_EPB - 1) / BUILD_EPB);
    switch (gen_mode_of(c)) {
    case 0: hipLaunchKernelGGL(k_build<0>, grid, dim3(WAVE), smem, st, *S, *c); break;
    case 1: hipLaunchKernelGGL(k_build<1>, grid, dim3(WAVE), smem, st, *S, *c); break;
    default: hipLaunchKernelGGL(k_build<2>, grid, dim3(WAVE), smem, st, *S, *c);
    }
}
// envs per index-order wave of k_step.  A batch below 64 x 1024 envs is spread over more, emptier waves (less divergence per wave, no idle SIMDs);
// more waves than wave SLOTS never pays: a wave's cost is the union of its lanes' paths, and a block without a slot starts when the first waves
// END.  The capped W <= 32 kernel has two slots per SIMD (2048): 1024 index-order blocks + the stair blocks all fit.  The wider instances have ONE
// (1024), and the few stair waves that really descend keep theirs for the whole launch -- with exactly 1024 index-order blocks as many of them
// waited ~38 us for a slot and ended last (round 3, default dungeon: block 1278 start 39.5 us, end 104.5 us of a 108 us launch).  So there the
// index-order blocks leave 16 slots free: 32 768 envs run as 993 waves of 33 envs.
int rgk_step_epw(int n, int slots_per_simd) {
    static const int epw_env = getenv("ROGUE_GYM_HIP_EPW") ? atoi(getenv("ROGUE_GYM_HIP_EPW")) : 0;
    int epw = WAVE;
    while (epw > 16 && (n + epw - 1) / epw < 1024) epw >>= 1;
    if (slots_per_simd < 2 && epw < WAVE && (n + epw - 1) / epw > 1008) { epw = (n + 1007) / 1008; if (epw > WAVE) epw = WAVE; }  // (n = 64 513..65 472 gave 65: a lane per env, never more)
    if (epw_env >= 16 && epw_env <= 64) epw = epw_env;  // (>= 16: S.stats has one row per block of the largest grid, STAIR_BLOCKS + ceil(n / 16); rg_api.cpp)
    return epw;
}
// LDS of one step wave for this config, and how many of them a CU holds when every block of a launch is resident (two per SIMD for the capped W <= 32
// instance, else one): what the level-per-lane producer must leave free on a CU (rg_regen_lanes.hip lanes_plan)
static size_t step_lds(const RgConfig *c, int *stage_off_out) {
    const int hw = c->width * c->height;
    size_t stage = GEN_SLOT_BYTES(hw, c->room_num_x * c->room_num_y);  // the generator's staging grid + tables (inline descents, spare misses), shared with ...
    const bool n32 = c->width <= 96 && hw <= 4096;  // BFS rows as 32-bit words in registers (bfs_rows_n32): no LDS planes
    const size_t bfs_hi = n32 ? 0 : (size_t)10 * ((c->width + 63) / 64) * WAVE * 8;  // ... the BFS high distance planes
    if (bfs_hi > stage) stage = bfs_hi;
    stage = (stage + 15) & ~(size_t)15;
    // the per-lane columns first (STEP_LDS_*: window, parked overlay words, monster cache), then the staging area
    const int stage_off = STEP_LDS_MC + c->room_num_x * c->room_num_y * WAVE * 4;
    if (stage_off_out) *stage_off_out = stage_off;
    return (size_t)stage_off + stage;
}
size_t rgk_step_lds_per_cu(const RgConfig *c) { return step_lds(c, nullptr) * ((c->width <= 32 && gen_mode_of(c) == 0) ? 8 : 4); }
// the step-kernel classes that have an instance for a bound observation tensor (rg_obs_bind): the capped W <= 32 one and the 33..96-column one (BASELINE.json's
// mini and default grids); elsewhere a bound tensor is simply re-encoded in full
int rgk_step_bound_capable(const RgConfig *c) {
    const int hw = c->width * c->height;
    if (gen_mode_of(c) == 2) return 0;
    if (c->width <= 32 && gen_mode_of(c) == 0) return 1;
    return (c->width <= 96 && hw <= 4096 && c->width > 64) ? 1 : 0;
}
int rgk_step(const RgState *S, const RgState *SP_dev, const RgConfig *c, const uint8_t *keys, int use_spares, int parity, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const bool bnd = S->obs_list != nullptr && rgk_step_bound_capable(c);
    int hw = c->width * c->height;
    const bool n32 = c->width <= 96 && hw <= 4096;
    int stage_off = 0;
    const size_t smem = step_lds(c, &stage_off);
    const int epw = rgk_step_epw(S->n, (c->width <= 32 && gen_mode_of(c) == 0) ? 2 : 1);
    // parity >= 0: stair isolation with the list the last render / observation pass wrote into set `parity`
    const int nb = (S->n + epw - 1) / epw;
    const dim3 grid(parity >= 0 ? STAIR_BLOCKS + nb : nb), block(WAVE);
    // (ev0 / ev1: optional events stamped with this dispatch's own begin and end -- rg_timing; ev1 alone: the completion event k_regen's stream waits for)
#define RG_LAUNCH_STEP(K) do { if (ev0 || ev1) hipExtLaunchKernelGGL(K, grid, block, (uint32_t)smem, st, ev0, ev1, 0, *S, SP_dev, *c, keys, use_spares, stage_off, epw, parity); \
                               else hipLaunchKernelGGL(K, grid, block, smem, st, *S, SP_dev, *c, keys, use_spares, stage_off, epw, parity); } while (0)
    if (gen_mode_of(c) == 2) {
        // the LDS monster table of a > 64-room dungeon (256 B per room and wave) goes beyond the 64 KB a kernel gets by default: raise the kernel's limit once
        // (the attribute is per DEVICE: a process may hold handles on several; a failure stays in hipGetLastError, which rg_step_prefix checks right after)
        static size_t raised[64] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || smem > raised[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_huge), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess && dev >= 0 && dev < 64) raised[dev] = smem;
        }
        RG_LAUNCH_STEP(k_step_huge);
    } else if (c->width <= 32 && gen_mode_of(c) == 0) { if (bnd) RG_LAUNCH_STEP(k_step_w32<true>); else RG_LAUNCH_STEP(k_step_w32<false>); }   // (its generator instance holds 32-bit room sets)
    else if (n32 && c->width <= 64) RG_LAUNCH_STEP(k_step<1>);                        // ... a 32-column grid with 33..64 rooms (e.g. 32x48 with 8x5) steps here
    else if (n32) { if (bnd) RG_LAUNCH_STEP((k_step<2, true>)); else RG_LAUNCH_STEP(k_step<2>); }
    else if (c->width <= 128) RG_LAUNCH_STEP(k_step<3>);
    else RG_LAUNCH_STEP(k_step<4>);
#undef RG_LAUNCH_STEP
    return (int)grid.x;
}
void rgk_debug_descend(const RgState *S, const RgConfig *c, hipStream_t st) {
    int hw = c->width * c->height;
    const size_t smem = GEN_SLOT_BYTES(hw, c->room_num_x * c->room_num_y);
    const dim3 grid((S->n + BUILD_EPB - 1) / BUILD_EPB);
    switch (gen_mode_of(c)) {
    case 0: hipLaunchKernelGGL(k_debug_descend<0>, grid, dim3(WAVE), smem, st, *S, *c); break;
    case 1: hipLaunchKernelGGL(k_debug_descend<1>, grid, dim3(WAVE), smem, st, *S, *c); break;
    default: hipLaunchKernelGGL(k_debug_descend<2>, grid, dim3(WAVE), smem, st, *S, *c);
    }
}
void rgk_regen_gate(const uint32_t *mark, uint32_t target, uint32_t *err_any, hipStream_t st) { hipLaunchKernelGGL(k_regen_gate, dim3(1), dim3(WAVE), 0, st, mark, target, err_any); }
// spares = 0: next-level structures only; 2: ... and the URGENT spares (regen_body; the bulk comes from rgk_regen_lanes): a few dozen requests per step to find, so 64 envs per wave
void rgk_regen(const RgState *SP, const RgConfig *c, int bulk, int spares, const uint32_t *mark, uint32_t target, uint32_t *err_any, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if (mark) rgk_regen_gate(mark, target, err_any, st);
    int hw = c->width * c->height;
    size_t smem = GEN_SLOT_BYTES(hw, c->room_num_x * c->room_num_y);
    // envs per wave: a wave generates its claimed spares one after the other, so with 64 envs per wave the launch lasts as long as its unluckiest wave
    // (4-6 claims: 270 us, the next launch queued behind it) and its waves sit beside two or three launches of the step kernels.  8 envs per wave: 0.08
    // claims per wave, the launch is over in about one generation time, and the 8192 blocks that find nothing are gone at once.  (A/B knob.)
    // (bulk: every consumed spare the wave finds, not one -- the launches that build ALL spares: creation, after rg_seed)
#ifdef RG_DEV_KNOBS
    static const int epb_env = getenv("ROGUE_GYM_HIP_REGEN_EPB") ? atoi(getenv("ROGUE_GYM_HIP_REGEN_EPB")) : 0;
    const int epb = (epb_env >= 4 && epb_env <= WAVE) ? epb_env : (spares == 1 ? 8 : WAVE);
    static const int claims_env = getenv("ROGUE_GYM_HIP_REGEN_CLAIMS") ? atoi(getenv("ROGUE_GYM_HIP_REGEN_CLAIMS")) : 1;
    const int max_claims = bulk ? WAVE : claims_env;
#else
    const int epb = spares == 1 ? 8 : WAVE, max_claims = bulk ? WAVE : 1;
#endif
    const dim3 grid((SP->n + epb - 1) / epb);
    // (ev0 / ev1: optional events stamped with this dispatch's own begin and end -- rg_timing, kernel 4)
#define RG_LAUNCH_REGEN(K) do { if (ev0 || ev1) hipExtLaunchKernelGGL(K, grid, dim3(WAVE), (uint32_t)smem, st, ev0, ev1, 0, *SP, *c, epb, max_claims, spares); \
                                else hipLaunchKernelGGL(K, grid, dim3(WAVE), smem, st, *SP, *c, epb, max_claims, spares); } while (0)
    switch (gen_mode_of(c)) {
    case 0: RG_LAUNCH_REGEN(k_regen<0>); break;
    case 1: RG_LAUNCH_REGEN(k_regen<1>); break;
    default: RG_LAUNCH_REGEN(k_regen_huge);
    }
#undef RG_LAUNCH_REGEN
}
}
