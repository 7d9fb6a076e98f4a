// rg_regen_lanes.hip -- the background producer of the spare level-1 states, ONE LEVEL PER LANE (gfx950).
//
// k_regen (rg_kernels.hip) generates one level per WAVE: the scalars are wave-uniform, the RNG-ordered chain runs on the scalar unit and the lanes
// share the tile work.  That is the right shape for a level somebody waits for (a descent inside the turn, a next-level structure wanted two steps
// later): ~28 us.  It is the wrong shape for the ~270 auto-reset spares a 65 536-env step consumes, which nobody needs before the env's next
// episode ends: 290 one-level waves per step issued 9.6 M instructions -- as many as k_step itself -- and cost every SIMD 9 us of issue slots per
// step (round 4: 597 M env-steps/s against 684 M without regeneration).  Here a wave generates up to 64 levels AT ONCE, one per lane, as plain
// per-lane code: every lane runs GameConfig::build (core/src/lib.rs:193-228) for its own env -- its own three RNG streams in registers, its own
// grid, room / monster / gold / corridor tables and maze stack in LDS -- and the wave pays the union of its lanes' paths once.  Level-1 builds of
// one config follow nearly the same path (the same room grid, the same loops; mazes need a dark room and are rare on level 1), so the union costs
// about twice one lane's path: ~20 000 vector instructions for 64 levels instead of ~30 000 mostly scalar ones for ONE.
//
// The result is bit-identical to the wave-per-level generator (same draws in the same order on the same streams, same tile words): the existing
// hand-off tests compare envs whose spares came from here with the CPU restatement of the reference, and tests/test_gpu_features.py runs the two producers side by side.
//
// Layout in LDS (dynamic): L grids `u16 [lane][H*W]` with a lane stride of H*W*2 + 4 bytes -- an odd number of 4-byte words, so the lanes of a
// wave touching the same cell index hit 64 different banks, and a lane's grid is contiguous for the copy-out --, the tables `u32 [slot][64]`
// (11 per room: rect, meta, three monster words, two gold words, four corridor-record words), the maze stacks `u16 [entry][64]`, and the claim list.
// L = 64 for the mini dungeon (81 KB); wider grids run with fewer active lanes per wave (80x24: 32).
//
// file:line citations are relative to /root/reference.
#include <cstdio>
#include "rg_gen.h"

#define LG_TABS_PER_ROOM 11

struct LG {
    lds_u16 *g;      // this lane's grid
    lds_u32 *t;      // this lane's column of the tables: word k at t[k * WAVE]
    lds_u16 *stk;    // this lane's column of the maze stacks: entry i at stk[i * WAVE]
    int nr, W, stack_cap;
    Rng rd, ri, re;  // dungeon / item / enemy streams
    uint32_t err;
#ifdef RG_DEV_KNOBS
    unsigned long long *pf, pt;  // development aid (RG_LANES_PROF=1): wave time per phase, summed over the waves of all launches
#endif
};
#ifdef RG_DEV_KNOBS
#define LGM(G, id) do { if ((G).pf) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { atomicAdd(&(G).pf[id], now_ - (G).pt); atomicAdd(&(G).pf[32 + (id)], 1ull); } (G).pt = now_; } } while (0)
#else
#define LGM(G, id) ((void)0)
#endif
#define LT(G, tab, i) ((G).t[((tab) * (G).nr + (i)) * WAVE])
enum { LT_RECT = 0, LT_META, LT_MONW, LT_MONHP, LT_MONEXP, LT_GOLDPOS, LT_GOLDAMT, LT_EA /* 2 per room */, LT_EB = LT_EA + 2 /* 2 per room */ };
#define LM_CONN_SHIFT 8   // bits 8..11 of the meta word: which of the room's four grid neighbours it is joined to (slot order Up, Left, Right, Down; passages.rs:222-270)

// Free-cell selection of one room (rooms.rs:134 via floor.rs:122,144,337-339): the implicit set of rg_kernels.hip's room_select -- interior cells of a
// Normal room / dug cells of a Maze room in row-major order, minus `excl` -- here with the maze cells counted by the lane itself.
__device__ __forceinline__ bool lg_room_select(LG &G, int rid, uint32_t excl, uint32_t &out) {
    const uint32_t meta = LT(G, LT_META, rid);
    const int kind = meta & RM_KIND_MASK;
    if (kind == RK_EMPTY) return false;
    int x0, y0, x1, y1;
    unpack_rect(LT(G, LT_RECT, rid), x0, y0, x1, y1);
    if (kind == RK_NORMAL) {
        const int iw = x1 - x0 - 2, ih = y1 - y0 - 2;
        int count = iw * ih, eo = -1;
        if (excl != ~0u) { eo = (POS_Y(excl) - y0 - 1) * iw + (POS_X(excl) - x0 - 1); count--; }
        if (count <= 0) return false;
        int nth = (int)range64(G.rd, 0, (uint64_t)count);
        if (eo >= 0 && eo <= nth) nth++;
        const int qy = small_div(nth, iw);
        out = POS(x0 + 1 + (nth - qy * iw), y0 + 1 + qy);
        return true;
    }
    int count = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) count += ((G.g[y * G.W + x] & C_MAZE) && POS(x, y) != excl) ? 1 : 0;
    if (count == 0) return false;
    int nth = (int)range64(G.rd, 0, (uint64_t)count);
    out = POS(x0, y0);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            if ((G.g[y * G.W + x] & C_MAZE) && POS(x, y) != excl && nth-- == 0) out = POS(x, y);
    return true;
}
// Floor::select_cell (floor.rs:333-346)
__device__ __forceinline__ bool lg_floor_select(LG &G, uint32_t non_empty, int mode /*0 stair, 1 player*/, uint32_t &out) {
    uint32_t cand = non_empty;
    while (cand) {
        const int idx = nth_bit(cand, (int)range64(G.rd, 0, (uint64_t)__popc(cand)));
        uint32_t excl = ~0u;
        if (mode == 0) { const uint32_t g = LT(G, LT_GOLDPOS, idx); if (g & 0x10000u) excl = g & 0xffff; }
        else { const uint32_t w = LT(G, LT_MONW, idx); if ((w >> 24) & MF_ALIVE) excl = w & 0xffff; }
        if (lg_room_select(G, idx, excl, out)) return true;
        cand &= ~(1u << idx);
    }
    return false;
}
// gen_attr (floor.rs:420-451) for Passage / Door cells
__device__ __forceinline__ uint32_t lg_gen_attr(const RgConfig &c, LG &G, int kind, uint32_t level) {
    if (range32(G.rd, 0, c.dark_level) < level) {
        if (kind == S_PASSAGE) { if (does_happen(G.rd, c.hidden_passage_rate_inv)) return C_HIDDEN; }
        else { if (does_happen(G.rd, c.locked_door_rate_inv)) return C_LOCKED; }
    }
    return 0;
}
// dig_maze (maze.rs:38-89) with an explicit stack, the lane's own (same visiting and draw order as the recursion)
__device__ __forceinline__ void lg_dig_maze(LG &G, int x0, int y0, int x1, int y1) {
    const int W = G.W;
    const int mw = (x1 - x0 + 1) >> 1, mh = (y1 - y0 + 1) >> 1;  // maze nodes = every other cell of the room in x and y
    const bool bitmap = mw * mh <= 64;  // the dug nodes as a 64-bit mask of the lane: the four neighbour tests of a visit never touch LDS
    const uint16_t dug = (uint16_t)(S_NONE | C_MAZE);
    G.g[y0 * W + x0] = dug;
    uint64_t seen = 1ull;
    int cx = x0, cy = y0, sp = 1;
    for (;;) {
        uint32_t cand = 0;
        if (bitmap) {
            const int ix = (cx - x0) >> 1, iy = (cy - y0) >> 1, at = iy * mw + ix;
            if (iy > 0 && !((seen >> (at - mw)) & 1ull)) cand |= 1u;       // Up
            if (iy + 1 < mh && !((seen >> (at + mw)) & 1ull)) cand |= 2u;  // Down
            if (ix > 0 && !((seen >> (at - 1)) & 1ull)) cand |= 4u;        // Left
            if (ix + 1 < mw && !((seen >> (at + 1)) & 1ull)) cand |= 8u;   // Right
        } else {
#pragma unroll
            for (int d = 0; d < 4; d++) {  // Up, Down, Left, Right
                const int nx = cx + 2 * dir_dx(d), ny = cy + 2 * dir_dy(d);
                if (!(nx < x0 || nx >= x1 || ny < y0 || ny >= y1) && !(G.g[ny * W + nx] & C_MAZE)) cand |= 1u << d;
            }
        }
        if (!cand) {  // dead end: back to the parent
            if (--sp == 0) break;
            const uint32_t top = G.stk[(sp - 1) * WAVE];
            cx = POS_X(top); cy = POS_Y(top);
            continue;
        }
        const int dig = nth_bit(cand, reservoir4(G.rd, __popc(cand)));
        const int ddx = dir_dx(dig), ddy = dir_dy(dig);
        G.g[(cy + ddy) * W + cx + ddx] = dug;
        G.g[(cy + 2 * ddy) * W + cx + 2 * ddx] = dug;
        if (sp >= G.stack_cap) { G.err |= RG_FLAG_ERR_INTERNAL; break; }  // never: one entry per maze node of the room at most (stack_cap: rg_api.cpp's maze_cap)
        G.stk[(sp - 1) * WAVE] = (uint16_t)POS(cx, cy);
        sp++;
        cx += 2 * ddx; cy += 2 * ddy;
        seen |= 1ull << ((((cy - y0) >> 1) * mw + ((cx - x0) >> 1)) & 63);
    }
}
// select_start_or_end (passages.rs:143-179).  dir: 0 Up 1 Down 2 Left 3 Right
__device__ __forceinline__ uint32_t lg_select_door(LG &G, int rid, int dir) {
    const uint32_t meta = LT(G, LT_META, rid);
    const int kind = meta & RM_KIND_MASK;
    int x0, y0, x1, y1;
    unpack_rect(LT(G, LT_RECT, rid), x0, y0, x1, y1);
    if (kind == RK_EMPTY) return POS(x0, y0);
    if (kind == RK_NORMAL) {
        if (dir < 2) {
            const int k = (int)range64(G.rd, 0, (uint64_t)(x1 - x0 - 2));
            return POS(x0 + 1 + k, dir == 1 ? y1 - 1 : y0);
        }
        const int k = (int)range64(G.rd, 0, (uint64_t)(y1 - y0 - 2));
        return POS(dir == 2 ? x0 : x1 - 1, y0 + 1 + k);
    }
    // Maze: shrink the probe rectangle from the facing side until its edge holds maze cells (termination: rg_kernels.hip select_door)
    int rx0 = x0, ry0 = y0, rx1 = x1, ry1 = y1;
    for (int guard = 0; guard <= RG_MAX_W && rx0 < rx1 && ry0 < ry1; guard++) {
        int lx0, ly0, ldx, ldy, ln;
        if (dir < 2) {
            const int yy = dir == 1 ? ry1 - 1 : ry0;
            lx0 = rx0 > x0 ? rx0 : x0; ly0 = yy;
            const int lx1 = rx1 < x1 ? rx1 : x1;
            ldx = 1; ldy = 0; ln = (yy >= y0 && yy < y1 && lx1 > lx0) ? lx1 - lx0 : 0;
        } else {
            const int xx = dir == 2 ? rx0 : rx1 - 1;
            lx0 = xx; ly0 = ry0 > y0 ? ry0 : y0;
            const int ly1 = ry1 < y1 ? ry1 : y1;
            ldx = 0; ldy = 1; ln = (xx >= x0 && xx < x1 && ly1 > ly0) ? ly1 - ly0 : 0;
        }
        int cnt = 0;
        for (int i = 0; i < ln; i++) cnt += (G.g[(ly0 + i * ldy) * G.W + lx0 + i * ldx] & C_MAZE) ? 1 : 0;
        if (cnt) {
            int nth = (int)range64(G.rd, 0, (uint64_t)cnt);
            uint32_t res = POS(lx0, ly0);
            for (int i = 0; i < ln; i++)
                if ((G.g[(ly0 + i * ldy) * G.W + lx0 + i * ldx] & C_MAZE) && nth-- == 0) res = POS(lx0 + i * ldx, ly0 + i * ldy);
            return res;
        }
        if (dir == 1) ry1--; else if (dir == 2) rx0--; else if (dir == 3) rx1--; else ry0--;
    }
    G.err |= RG_FLAG_ERR_INTERNAL;
    return POS(x0, y0);
}
// connect_2rooms (passages.rs:84-133): the two doors and the bend now, the cells in the deferred gen_attr pass
__device__ __forceinline__ void lg_connect_rooms(LG &G, int r1, int r2, int dir, int &n_edges) {
    if (dir == 0 || dir == 2) { const int t = r1; r1 = r2; r2 = t; dir ^= 1; }
    const uint32_t s = lg_select_door(G, r1, dir);
    const uint32_t t = lg_select_door(G, r2, dir ^ 1);
    const int k1 = (LT(G, LT_META, r1) & RM_KIND_MASK) == RK_NORMAL, k2 = (LT(G, LT_META, r2) & RM_KIND_MASK) == RK_NORMAL;
    int bend;
    if (dir == 1) bend = (int)range32(G.rd, (uint32_t)(POS_Y(s) + 1), (uint32_t)POS_Y(t));
    else bend = (int)range32(G.rd, (uint32_t)(POS_X(s) + 1), (uint32_t)POS_X(t));
    if (n_edges < 2 * G.nr) {  // always: a level has fewer corridors than 2 x rooms (rg_state.h)
        LT(G, LT_EA, n_edges) = s | (t << 16);
        LT(G, LT_EB, n_edges) = (uint32_t)bend | ((uint32_t)(dir == 1) << 8) | ((uint32_t)k1 << 9) | ((uint32_t)k2 << 10);
        n_edges++;
    } else G.err |= RG_FLAG_ERR_INTERNAL;
}
// one recorded corridor in registration order (passages.rs:98-132 + floor.rs:87-101): start door, end door, then the three legs; register_cell's update rule
__device__ __forceinline__ void lg_paint_corridor(const RgConfig &c, LG &G, uint32_t a, uint32_t b, uint32_t level) {
    const int W = G.W;
    const int sx = POS_X(a), sy = POS_Y(a), ex = POS_X(a >> 16), ey = POS_Y(a >> 16);
    const int bend = b & 0xff;
    const bool down = (b >> 8) & 1;
    const int kind_s = ((b >> 9) & 1) ? S_DOOR : S_PASSAGE, kind_e = ((b >> 10) & 1) ? S_DOOR : S_PASSAGE;
    const int dx = down ? 0 : 1, dy = down ? 1 : 0;
    const int tsx = down ? sx : bend, tsy = down ? bend : sy;
    const int tex = down ? ex : bend, tey = down ? bend : ey;
    const int tdx = down ? (sx < ex ? 1 : -1) : 0, tdy = down ? 0 : (sy < ey ? 1 : -1);
    const int n1 = (down ? bend - sy : bend - sx) - 1;
    const int n2 = down ? (ex > sx ? ex - sx : sx - ex) : (ey > sy ? ey - sy : sy - ey);
    const int n3 = down ? ey - bend : ex - bend;
    const int total = 2 + n1 + n2 + n3;
    for (int i = 0; i < total; i++) {
        int x = sx, y = sy, kind = kind_s;
        if (i == 1) { x = ex; y = ey; kind = kind_e; }
        else if (i >= 2) {
            int k = i - 2;
            kind = S_PASSAGE;
            if (k < n1) { x = sx + dx * (k + 1); y = sy + dy * (k + 1); }
            else if ((k -= n1) < n2) { x = tsx + tdx * k; y = tsy + tdy * k; }
            else { k -= n2; x = tex + dx * k; y = tey + dy * k; }
        }
        const uint32_t attr = lg_gen_attr(c, G, kind, level);
        uint32_t v = G.g[y * W + x];
        v = (v & ~C_ATTR_MASK) | attr;
        if (kind == S_DOOR) v |= C_DOOR;
        if (!attr) v = (v & ~C_SURF_MASK) | (uint32_t)kind;
        G.g[y * W + x] = (uint16_t)v;
    }
}
// select_candidate (passages.rs:69-82): reservoir over the grid-neighbour rooms in ascending id = Up, Left, Right, Down
__device__ __forceinline__ int lg_select_candidate(const RgConfig &c, LG &G, int node, uint32_t excl_set, uint32_t excl_dirs, int &dir_out) {
    const int rnx = c.room_num_x, rny = c.room_num_y;
    const int ny0 = small_div(node, rnx), nx0 = node - ny0 * rnx;
    const int ids[4] = {node - rnx, node - 1, node + 1, node + rnx};
    const bool ok[4] = {ny0 > 0, nx0 > 0, nx0 + 1 < rnx, ny0 + 1 < rny};
    const int dirs[4] = {0, 2, 3, 1};
    uint32_t cand = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (ok[k] && !((excl_set >> (ids[k] & 31)) & 1u) && !((excl_dirs >> k) & 1u)) cand |= 1u << k;
    if (!cand) return -1;
    const int k = nth_bit(cand, reservoir4(G.rd, __popc(cand)));
    int res = ids[0];
    dir_out = dirs[0];
#pragma unroll
    for (int t = 1; t < 4; t++)
        if (k == t) { res = ids[t]; dir_out = dirs[t]; }
    return res;
}

// Dungeon::new_level_ (rogue/mod.rs:434-481) up to the stairs -- rg_kernels.hip's gen_structure, one lane's worth.  The grid was cleared by the wave.
__device__ __forceinline__ uint32_t lg_structure(const RgConfig &c, LG &G, uint32_t level) {
    const int W = G.W, nrooms = G.nr, rnx = c.room_num_x;
    for (int s = 0; s < nrooms; s++) { LT(G, LT_MONW, s) = 0; LT(G, LT_GOLDPOS, s) = 0; LT(G, LT_MONHP, s) = 0; LT(G, LT_MONEXP, s) = 0; LT(G, LT_GOLDAMT, s) = 0; }  // (the unused slots' words too: LDS keeps the last round's)
    // ---- gen_rooms (rooms.rs:165-211) ----
    uint32_t empty_num = range32(G.rd, 0, c.max_empty_rooms + 1);
    if (empty_num >= (uint32_t)nrooms) empty_num = nrooms - 1;
    uint32_t empty_mask = 0;
    {
        uint32_t sel = nrooms >= 32 ? ~0u : ((1u << nrooms) - 1u);
        for (uint32_t k = 0; k < empty_num; k++) {
            const int id = nth_bit(sel, (int)range64(G.rd, 0, (uint64_t)__popc(sel)));
            sel &= ~(1u << id);
            empty_mask |= 1u << id;
        }
    }
    for (int i = 0; i < nrooms; i++) {  // make_room (rooms.rs:214-269)
        int ax0, ay0, ax1, ay1;
        assigned_area(c, i, ax0, ay0, ax1, ay1);
        const int rsx = ax1 - ax0, rsy = ay1 - ay0;
        uint32_t rect, meta;
        if ((empty_mask >> i) & 1u) {
            const int x = (int)range32(G.rd, 1, (uint32_t)(rsx - 1)) + ax0;
            const int y = (int)range32(G.rd, 1, (uint32_t)(rsy - 1)) + ay0;
            rect = (uint32_t)x | ((uint32_t)y << 8);
            meta = RK_EMPTY | RM_DARK;
        } else {
            const bool dark = range32(G.rd, 0, c.dark_level) < level;
            if (dark && does_happen(G.rd, c.maze_rate_inv)) {
                const int mx1 = ax0 + rsx - 1, my1 = ay0 + rsy - 1;
                rect = (uint32_t)ax0 | ((uint32_t)ay0 << 8) | ((uint32_t)mx1 << 16) | ((uint32_t)my1 << 24);
                meta = RK_MAZE | RM_DARK;
                lg_dig_maze(G, ax0, ay0, mx1, my1);
            } else {
                const int sx = (int)range32(G.rd, (uint32_t)c.min_room_x, (uint32_t)rsx);
                const int sy = (int)range32(G.rd, (uint32_t)c.min_room_y, (uint32_t)rsy);
                const int ox = (int)range32(G.rd, 0, (uint32_t)(rsx - sx)) + ax0;
                const int oy = (int)range32(G.rd, 0, (uint32_t)(rsy - sy)) + ay0;
                rect = (uint32_t)ox | ((uint32_t)oy << 8) | ((uint32_t)(ox + sx) << 16) | ((uint32_t)(oy + sy) << 24);
                meta = RK_NORMAL | (dark ? RM_DARK : 0);
            }
        }
        LT(G, LT_RECT, i) = rect;
        LT(G, LT_META, i) = meta;
    }
    LGM(G, 2);
    // ---- paint rooms in id order (floor.rs:61-71; Room::draw rooms.rs:58-82) ----
    for (int i = 0; i < nrooms; i++) {
        const uint32_t meta = LT(G, LT_META, i);
        const int kind = meta & RM_KIND_MASK;
        if (kind == RK_EMPTY) continue;
        int x0, y0, x1, y1;
        unpack_rect(LT(G, LT_RECT, i), x0, y0, x1, y1);
        if (kind == RK_NORMAL) {
            const uint16_t fl = (uint16_t)(S_FLOOR | ((meta & RM_DARK) ? C_DARK : 0));
            for (int y = y0; y < y1; y++) {
                lds_u16 *row = G.g + y * W;
                const bool wall_row = y == y0 || y == y1 - 1;
                row[x0] = wall_row ? (uint16_t)S_WALLX : (uint16_t)S_WALLY;
                for (int x = x0 + 1; x < x1 - 1; x++) row[x] = wall_row ? (uint16_t)S_WALLX : fl;
                row[x1 - 1] = wall_row ? (uint16_t)S_WALLX : (uint16_t)S_WALLY;
            }
        } else {  // maze cells in ascending range index; each draws gen_attr (Passage)
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++)
                    if (G.g[y * W + x] & C_MAZE) G.g[y * W + x] = (uint16_t)(C_MAZE | S_PASSAGE | lg_gen_attr(c, G, S_PASSAGE, level));
        }
    }
    LGM(G, 3);
    // ---- dig_passges (passages.rs:16-67) ----
    int n_edges = 0;
    {
        auto conn_join = [&](int a, int b, int dir_ab) {  // dir_ab: 0 Up 1 Down 2 Left 3 Right, as seen from a
            const uint32_t slot_of_dir = 0x2130u;         // direction code -> candidate slot: Up 0, Down 3, Left 1, Right 2
            const uint32_t ka = (slot_of_dir >> (4 * dir_ab)) & 3u, kb = (slot_of_dir >> (4 * (dir_ab ^ 1))) & 3u;
            LT(G, LT_META, a) = LT(G, LT_META, a) | (1u << (LM_CONN_SHIFT + ka));
            LT(G, LT_META, b) = LT(G, LT_META, b) | (1u << (LM_CONN_SHIFT + kb));
        };
        uint32_t selected = 0;
        int cur = (int)range64(G.rd, 0, (uint64_t)nrooms), n_sel = 1;
        selected |= 1u << cur;
        while (n_sel < nrooms) {
            int dir = 0;
            const int nxt = lg_select_candidate(c, G, cur, selected, 0u, dir);
            if (nxt >= 0) {
                selected |= 1u << nxt; n_sel++;
                conn_join(cur, nxt, dir);
                lg_connect_rooms(G, cur, nxt, dir, n_edges);
            } else cur = nth_bit(selected, (int)range64(G.rd, 0, (uint64_t)n_sel));
        }
        const uint32_t try_num = range32(G.rd, 0, c.max_extra_edges);
        for (uint32_t t = 0; t < try_num; t++) {
            int dir = 0;
            const int room1 = (int)range64(G.rd, 0, (uint64_t)nrooms);
            const int room2 = lg_select_candidate(c, G, room1, 0u, (LT(G, LT_META, room1) >> LM_CONN_SHIFT) & 0xfu, dir);
            if (room2 >= 0) {
                conn_join(room1, room2, dir);
                lg_connect_rooms(G, room1, room2, dir, n_edges);
            }
        }
    }
    LGM(G, 4);
    for (int k = 0; k < n_edges; k++) lg_paint_corridor(c, G, LT(G, LT_EA, k), LT(G, LT_EB, k), level);
    LGM(G, 5);
    const uint32_t non_empty = (nrooms >= 32 ? ~0u : ((1u << nrooms) - 1u)) & ~empty_mask;
    (void)rnx;
    // ---- gold (floor.rs:132-153, item/gold.rs:18-24) ----
    for (int i = 0; i < nrooms; i++) {
        uint32_t pos;
        if (!lg_room_select(G, i, ~0u, pos)) continue;
        if (!does_happen(G.ri, c.gold_rate_inv)) continue;
        const uint32_t num = range32(G.ri, 0, c.gold_base + c.gold_per_level * level) + c.gold_minimum;
        LT(G, LT_GOLDPOS, i) = pos | 0x10000u;
        LT(G, LT_GOLDAMT, i) = num;
        LT(G, LT_META, i) = LT(G, LT_META, i) | RM_HAS_GOLD;
        G.g[POS_Y(pos) * W + POS_X(pos)] |= C_GOLD;
    }
    LGM(G, 6);
    // ---- stair (floor.rs:156-167) ----
    {
        uint32_t pos;
        if (lg_floor_select(G, non_empty, 0, pos)) {
            const uint32_t v = G.g[POS_Y(pos) * W + POS_X(pos)];
            G.g[POS_Y(pos) * W + POS_X(pos)] = (uint16_t)((v & ~C_SURF_MASK) | S_STAIR);
        }
    }
    return non_empty;
}
// the monsters (floor.rs:106-130, enemies.rs:265-320): rg_kernels.hip's gen_populate without the no-hide reveal (applied by the copy-out)
__device__ __forceinline__ uint32_t lg_populate(const RgConfig &c, LG &G, uint32_t level) {
    uint32_t alive = 0;
    if (c.n_enemies <= 0) return 0;
    const uint32_t mn = level >= 4 ? level - 4 : 0, mx = level + 6;
    const uint32_t lev_add = lev_add_of(c, level);
    for (int i = 0; i < G.nr; i++) {
        uint32_t pos;
        if (!lg_room_select(G, i, ~0u, pos)) continue;
        const bool has_gold = LT(G, LT_META, i) & RM_HAS_GOLD;
        if (!parcent(G.re, has_gold ? c.appear_rate_gold : c.appear_rate_nogold)) continue;
        const uint32_t len = (uint32_t)c.n_enemies;
        uint32_t idx = range32(G.re, mn, mx);
        if (idx > len) { const uint32_t rg = len < 5 ? len : 5; idx = (uint32_t)range64(G.re, len - rg, len); }
        if (idx >= len) continue;
        const uint32_t type = idx;
        const int64_t mlevel = (int64_t)c.mon[type].level + lev_add;
        int64_t hp = 0;
        uint32_t exp_add;
        if (mlevel >= 1 && mlevel < (1 << 24)) {
            const uint32_t ml = (uint32_t)mlevel;
            uint32_t h32 = 0;
            for (int k = 0; k < 8; k++) h32 += (uint32_t)range64(G.re, 1, (uint64_t)ml + 1);
            const uint32_t base = ml == 1 ? h32 / 8u : h32 / 6u;
            exp_add = ml >= 10 ? base * 20u : base * 4u;
            hp = h32;
        } else {
            for (int k = 0; k < 8; k++) hp += (int64_t)range64(G.re, 1, (uint64_t)mlevel + 1);
            const int64_t base = mlevel == 1 ? hp / 8 : hp / 6;
            exp_add = mlevel >= 10 ? (uint32_t)base * 20u : (uint32_t)base * 4u;
        }
        LT(G, LT_MONW, i) = pos | (type << 16) | ((uint32_t)MF_ALIVE << 24);
        LT(G, LT_MONHP, i) = (uint32_t)(int32_t)hp;
        LT(G, LT_MONEXP, i) = c.mon[type].exp + lev_add * 10u + exp_add;
        alive++;
    }
    return alive;
}
// Floor::player_in(cd, init = true) (floor.rs:264-295) on the lane's grid.  Returns the number of monsters woken (EnemyHandler::activate_area, enemies.rs:342-362).
__device__ __forceinline__ uint32_t lg_player_in_init(const RgConfig &c, LG &G, int x, int y, uint32_t alive) {
    const int W = G.W;
    uint32_t woken = 0;
    const int rid = room_id_of(c, x, y);
    if (rid >= 0) {
        const uint32_t meta = LT(G, LT_META, rid);
        if (!(meta & RM_VISITED)) {  // Floor::enters_room (floor.rs:231-247)
            LT(G, LT_META, rid) = meta | RM_VISITED;
            if ((meta & RM_KIND_MASK) == RK_NORMAL && !(meta & RM_DARK)) {
                int x0, y0, x1, y1;
                unpack_rect(LT(G, LT_RECT, rid), x0, y0, x1, y1);
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) G.g[yy * W + xx] |= C_DRAWN | C_VISIBLE;
            }
        }
        if (alive)
            for (int s = 0; s < G.nr; s++) {
                const uint32_t w = LT(G, LT_MONW, s);
                const uint32_t fl = w >> 24;
                if (!(fl & MF_ALIVE) || (fl & MF_ACTIVE)) continue;
                if (!(c.mon[(w >> 16) & 0xff].attr & EA_MEAN)) continue;
                if (room_id_of(c, POS_X(w), POS_Y(w)) != rid) continue;
                LT(G, LT_MONW, s) = w | ((uint32_t)MF_ACTIVE << 24);
                woken++;
            }
    }
    G.g[y * W + x] |= C_VISITED;
    for (int d = 0; d < 9; d++) {
        const int cx = x + dir_dx(d), cy = y + dir_dy(d);
        if (!in_bounds(c, cx, cy)) continue;
        const uint32_t v = G.g[cy * W + cx];
        const bool diag = d >= 4 && d < 8;
        if (diag && (v & C_SURF_MASK) == S_PASSAGE) continue;
        if (v & C_HIDDEN) continue;  // Cell::approached (field.rs:20-26)
        G.g[cy * W + cx] = (uint16_t)(v | C_DRAWN | C_VISIBLE);
    }
    return woken;
}

// GameConfig::to_global's seed choice (core/src/lib.rs:157-165) for one lane's env: rg_kernels.hip's build_prologue, per lane (the ticket of a
// `seed: None` env is taken by the lane itself)
__device__ __forceinline__ void lg_seed(const RgState &S, int e, uint64_t &lo, uint64_t &hi) {
    lo = S.seed_lo[e]; hi = S.seed_hi[e];
    const uint32_t mode = S.reseed[e];
    if (!mode) return;
    const uint32_t k = atomicAdd(&S.build_ctr[e], 1u);
    uint64_t z = splitmix64(lo ^ splitmix64(hi + k)), y = splitmix64(z ^ hi);
    if (mode == 2 && S.range_lo) {
        const int n = S.n;
        const uint64_t r_lo = S.range_lo[e], r_hi = S.range_lo[n + e], sp_lo = S.range_span[e], sp_hi = S.range_span[n + e];
        uint64_t m_lo, m_hi;  // smallest 2^b - 1 >= span - 1
        if (sp_hi) { m_lo = ~0ull; m_hi = ~0ull >> __clzll((long long)sp_hi); }
        else { m_hi = 0; m_lo = sp_lo > 1 ? ~0ull >> __clzll((long long)(sp_lo - 1)) : 0ull; }
        for (int t = 0; t < 64; t++) {
            const uint64_t c_lo = z & m_lo, c_hi = y & m_hi;
            if (c_hi < sp_hi || (c_hi == sp_hi && c_lo < sp_lo) || t == 63) { z = c_lo; y = c_hi; break; }
            z = splitmix64(z); y = splitmix64(y ^ z);
        }
        if (!(y < sp_hi || (y == sp_hi && z < sp_lo))) { z = 0; y = 0; }
        lo = r_lo + z; hi = r_hi + y + (lo < r_lo ? 1ull : 0ull);
    } else { lo = z; hi = y; }
}

extern __shared__ __align__(16) uint8_t lg_smem[];

// The consumed spares of a launch, compacted: k_lanes_scan appends the index of every env whose spare is consumed (sp_ready == 0) to `list` (count in
// q[0]); k_regen_lanes then takes them L at a time.  Two kernels in stream order on the generator's stream, so the list is complete when it is read --
// and every wave of the producer is full whatever the reset rate (270 per step under the random policy, 2 200 with 30-step episodes, all of them
// after creation / rg_seed).  1024 envs per scanning wave: a uint4 of sp_ready words per lane and round, four rounds in flight.
#define LG_SCAN_EPW 1024
__global__ void __launch_bounds__(WAVE) k_lanes_scan(const uint32_t *__restrict__ sp_ready, int n, uint32_t *__restrict__ q, int32_t *__restrict__ list) {
    const int lane = threadIdx.x;
    const int base = blockIdx.x * LG_SCAN_EPW;
    uint4 v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int e = base + r * 256 + lane * 4;
        v[r] = make_uint4(1, 1, 1, 1);
        if (e + 3 < n) v[r] = *reinterpret_cast<const uint4 *>(sp_ready + e);
        else {
            if (e < n) v[r].x = sp_ready[e];
            if (e + 1 < n) v[r].y = sp_ready[e + 1];
            if (e + 2 < n) v[r].z = sp_ready[e + 2];
        }
    }
    int mine = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) mine += (v[r].x == 0u) + (v[r].y == 0u) + (v[r].z == 0u) + (v[r].w == 0u);
    int incl = mine;  // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    const int total = __shfl(incl, WAVE - 1);
    if (total == 0) return;
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&q[0], (uint32_t)total);
    int w = (int)uni(at) + incl - mine;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t qv[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (qv[k] == 0u) list[w++] = base + r * 256 + lane * 4 + k;
    }
}

// One wave: take L listed envs at a time, claim them, build them all at once -- one per lane --, write them through.
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4)))  // <= 128 registers: beside a step wave on any SIMD
k_regen_lanes(RgState SP, RgConfig c, const uint32_t *__restrict__ q, const int32_t *__restrict__ list, int L, int lane_stride /* bytes */, int tab_off, int stk_off, int stack_cap, int slots, int prio, unsigned long long *prof) {
    if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    // `slots` spares per env (rg_state.h sp_slots): spare (slot s, env e) is entry s * SP.n + e of the spare arrays; the seed arrays are the envs' own
    const int lane = threadIdx.x, n = SP.n * slots, W = c.width, H = c.height, HW = W * H, nrooms = c.room_num_x * c.room_num_y;
    const int total = (int)q[0];
  for (int first = blockIdx.x * L; first < total; first += gridDim.x * L) {
    const bool listed = lane < L && first + lane < total;
    const int e = listed ? list[first + lane] : 0;
    // (by CAS: nobody else refills spares while this producer is in use, but a claim is the protocol -- rg_kernels.hip regen_body)
    const bool claim = listed && atomicCAS(&SP.sp_ready[e], 0u, 2u) == 0u;
    const uint64_t cm = __ballot(claim);
    if (!cm) continue;
    // ---- fresh Fields: Surface::None, no attributes, for all L grids at once ----
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t nn = S_NONE | (S_NONE << 16);
        const u32x4 v = {nn, nn, nn, nn};
        __attribute__((address_space(3))) u32x4 *p = (__attribute__((address_space(3))) u32x4 *)lg_smem;
        const int n16 = (L * lane_stride) / 16;  // (L is a multiple of 4, the stride of 4)
        for (int i = lane; i < n16; i += WAVE) p[i] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    LG G;
    G.g = (lds_u16 *)(lg_smem + (size_t)lane * lane_stride);
    G.t = (lds_u32 *)(lg_smem + tab_off) + lane;
    G.stk = (lds_u16 *)(lg_smem + stk_off) + lane;
    G.nr = nrooms; G.W = W; G.stack_cap = stack_cap; G.err = 0;
#ifdef RG_DEV_KNOBS
    G.pf = prof; G.pt = __builtin_amdgcn_s_memtime();
#endif
    int px = 0, py = 0;
    uint32_t alive = 0, active = 0, on_stairs = 0;
    if (claim) {
        uint64_t lo, hi;
        lg_seed(SP, e % SP.n, lo, hi);
        rng_seed(G.ri, lo, hi); rng_seed(G.re, lo, hi); rng_seed(G.rd, lo, hi);
        LGM(G, 1);
        const uint32_t non_empty = lg_structure(c, G, 1u);
        LGM(G, 7);
        alive = lg_populate(c, G, 1u);
        LGM(G, 8);
        // Player::init_items (player.rs:136-153): one item-stream draw per InitItem::Weapon, in list order (rg_kernels.hip build_epilogue)
        for (int i = 0; i < c.n_init_draws; i++) (void)range32(G.ri, SP.init_draws[2 * i], SP.init_draws[2 * i + 1]);
        // actions::new_level's tail (actions.rs:130-137): place the player and enter the room
        uint32_t pos = 0;
        lg_floor_select(G, non_empty, 1, pos);
        px = POS_X(pos); py = POS_Y(pos);
        on_stairs = (G.g[py * W + px] & C_SURF_MASK) == S_STAIR;
        active = lg_player_in_init(c, G, px, py, alive);
        LGM(G, 9);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- hand-off: everything k_step will take over is written THROUGH (st_pub<true>), every store drained, then sp_ready = 1 (rg_kernels.hip regen_body) ----
    if (claim) {
        const uint32_t r[12] = {G.rd.x, G.rd.y, G.rd.z, G.rd.w, G.ri.x, G.ri.y, G.ri.z, G.ri.w, G.re.x, G.re.y, G.re.z, G.re.w};
#pragma unroll
        for (int k = 0; k < 12; k++) st_pub<true>(&SP.rng[(size_t)k * n + e], r[k]);
        st_pub<true>(&SP.p_pos[e], (uint16_t)POS(px, py));
        st_pub<true>(&SP.p_hp[e], (int32_t)c.init_hp); st_pub<true>(&SP.p_hpmax[e], (int32_t)c.init_hp); st_pub<true>(&SP.p_lvl[e], (int32_t)1);
        st_pub<true>(&SP.p_exp[e], 0u); st_pub<true>(&SP.food[e], c.hunger_time); st_pub<true>(&SP.quiet[e], 0u); st_pub<true>(&SP.pack_gold[e], c.init_gold);
        st_pub<true>(&SP.dlevel[e], 1u);
        st_pub<true>(&SP.mon_cnt[e], alive | (active << 8));
        st_pub<true>(&SP.on_stairs[e], (uint8_t)on_stairs);
        for (int s = 0; s < nrooms; s++) {
            const size_t g = (size_t)s * n + e;
            st_pub<true>(&SP.room_rect[g], (uint32_t)LT(G, LT_RECT, s)); st_pub<true>(&SP.room_meta[g], (uint8_t)LT(G, LT_META, s));
            st_pub<true>(&SP.mon_w0[g], (uint32_t)LT(G, LT_MONW, s)); st_pub<true>(&SP.mon_hp[g], (int32_t)LT(G, LT_MONHP, s)); st_pub<true>(&SP.mon_exp[g], (uint32_t)LT(G, LT_MONEXP, s));
            st_pub<true>(&SP.gold_pos[g], (uint32_t)LT(G, LT_GOLDPOS, s)); st_pub<true>(&SP.gold_amt[g], (uint32_t)LT(G, LT_GOLDAMT, s));
        }
        if (G.err) atomicOr(SP.err_any, G.err);  // (the env's flag word belongs to the k_step running beside this launch)
        LGM(G, 10);
    }
    // grids: the wave streams each claimed lane's grid, 8 bytes per lane and store; hide_dungeon = false (rogue/mod.rs:465-475) is applied on the way
    // (it only sets VISIBLE on rows 1..H-2, and nothing between it and here clears a bit)
    const uint32_t vis = c.hide_dungeon ? 0u : (uint32_t)C_VISIBLE;
    for (uint64_t mm = cm; mm;) {
        const int src = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int es = __shfl(e, src);
        const lds_u32 *g32 = (const lds_u32 *)(lg_smem + (size_t)src * lane_stride);
        if ((HW & 3) == 0) {
            unsigned long long *d8 = reinterpret_cast<unsigned long long *>(SP.cell + (size_t)es * HW);
            for (int i = lane; i < HW / 4; i += WAVE) {
                uint32_t a = g32[2 * i], b = g32[2 * i + 1];
                if (vis) {
                    const int c0 = 4 * i;
                    if (c0 >= W && c0 < HW - W) a |= vis;
                    if (c0 + 1 >= W && c0 + 1 < HW - W) a |= vis << 16;
                    if (c0 + 2 >= W && c0 + 2 < HW - W) b |= vis;
                    if (c0 + 3 >= W && c0 + 3 < HW - W) b |= vis << 16;
                }
                st_pub<true>(&d8[i], (unsigned long long)a | ((unsigned long long)b << 32));
            }
        } else {
            const lds_u16 *g16 = (const lds_u16 *)g32;
            uint16_t *dst = SP.cell + (size_t)es * HW;
            for (int i = lane; i < HW; i += WAVE) st_pub<true>(&dst[i], (uint16_t)(g16[i] | ((i >= W && i < HW - W) ? vis : 0u)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LGM(G, 11);
    if (claim) __hip_atomic_store(&SP.sp_ready[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // the next round clears the grids the copy-out has just read
  }
  (void)H; (void)prof;
}

// ---------------------------------------------------------------------------------------------
// host-callable launcher (rg_api.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" {
// Geometry of a launch for this config; lanes = 0 when the level-per-lane producer does not apply (more than 32 rooms: the room sets are 32-bit masks;
// a grid so large that fewer than 16 lanes' worth fits the CU's LDS) -- the caller then keeps the wave-per-level producer.
size_t rgk_step_lds_per_cu(const RgConfig *c);  // rg_kernels.hip
struct RgLanesPlan { int lanes, lane_stride, tab_off, stk_off, smem; };
static RgLanesPlan lanes_plan(const RgConfig *c, int maze_cap) {
    RgLanesPlan p = {0, 0, 0, 0, 0};
    const int hw = c->width * c->height, nr = c->room_num_x * c->room_num_y;
    if (nr > 32) return p;
    const int stride = ((hw * 2 + 3) & ~3) + 4;            // an odd number of 4-byte words when hw is even: conflict-free lock-step accesses
    const int fixed = LG_TABS_PER_ROOM * nr * WAVE * 4 + maze_cap * WAVE * 2 + 64;
    // Of the CU's 160 KB, what the step waves of a fully resident launch leave (four waves of k_step<2> on the 80x24 dungeon: 58 KB; eight of the capped
    // W <= 32 instance): with 150 KB for a producer wave whatever the config (round 5) a CU holding one had room for no step wave at all -- 25 CUs out of
    // 256 on the default workload, ~70 of its 993 one-per-SIMD step waves waiting for a second round: k_step<2> p99 153 us / max 380 us against 96 us p50
    // (profiles/r06_experiments.txt).  Fewer lanes per wave is the price (80x24: 36 -> 24).
    int budget = 160 * 1024 - (int)rgk_step_lds_per_cu(c) - 1024;
    if (budget > 150 * 1024) budget = 150 * 1024;
#ifdef RG_DEV_KNOBS
    if (const char *ev = getenv("ROGUE_GYM_HIP_LANE_LDS_KB")) budget = atoi(ev) * 1024;  // (the sweep of profiles/r06_experiments.txt)
#endif
    int lanes = (budget - fixed) / stride;
    if (lanes > WAVE) lanes = WAVE;
    lanes &= ~3;
    if (lanes < 16) return p;
    p.lanes = lanes; p.lane_stride = stride;
    p.tab_off = (lanes * stride + 15) & ~15;
    p.stk_off = p.tab_off + LG_TABS_PER_ROOM * nr * WAVE * 4;
    p.smem = (p.stk_off + maze_cap * WAVE * 2 + 15) & ~15;
    return p;
}
int rgk_regen_lanes_supported(const RgConfig *c, int maze_cap) { return lanes_plan(c, maze_cap).lanes; }
// q: [1] counter + list: [n] env indices (device scratch of the handle, used by one launch at a time: the generator's stream serialises them).
// bulk: as many waves as the whole batch needs for one round each (creation, rg_seed); else `waves` waves that loop over the list.
int rgk_regen_lanes(const RgState *SP, const RgConfig *c, uint32_t *q, int32_t *list, int bulk, int waves, int slots, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const RgLanesPlan p = lanes_plan(c, SP->maze_cap);
    if (!p.lanes) return 0;
    static size_t raised[64] = {0};  // more than the 64 KB a kernel gets by default: raise the kernel's limit once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (p.smem > 64 * 1024 && (dev < 0 || dev >= 64 || (size_t)p.smem > raised[dev])) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_regen_lanes), hipFuncAttributeMaxDynamicSharedMemorySize, p.smem) == hipSuccess && dev >= 0 && dev < 64) raised[dev] = (size_t)p.smem;
    }
    unsigned long long *prof = nullptr;
#ifdef RG_DEV_KNOBS
    static unsigned long long *d_prof = nullptr;
    if (getenv("RG_LANES_PROF")) {
        if (!d_prof) {
            (void)hipMalloc((void **)&d_prof, 64 * 8); (void)hipMemset(d_prof, 0, 64 * 8);
            atexit([] { unsigned long long hbuf[64]; (void)hipDeviceSynchronize(); (void)hipMemcpy(hbuf, d_prof, 64 * 8, hipMemcpyDeviceToHost);
                        for (int i = 1; i < 12; i++) fprintf(stderr, "lanes phase %2d: waves %llu  ticks/wave %.0f\n", i, hbuf[32 + i], hbuf[32 + i] ? (double)hbuf[i] / (double)hbuf[32 + i] : 0.0); });
        }
        prof = d_prof;
    }
#endif
    int prio = 0;
#ifdef RG_DEV_KNOBS
    static const int prio_env = getenv("ROGUE_GYM_HIP_LANE_PRIO") ? atoi(getenv("ROGUE_GYM_HIP_LANE_PRIO")) : 0;
    prio = prio_env;
#endif
    int L = p.lanes;
#ifdef RG_DEV_KNOBS
    static const int l_env = getenv("ROGUE_GYM_HIP_LANE_L") ? atoi(getenv("ROGUE_GYM_HIP_LANE_L")) : 0;  // fewer levels per wave and round (steady-state launches only)
    if (!bulk && l_env >= 4 && l_env < L) L = l_env & ~3;
#endif
    (void)hipMemsetAsync(q, 0, 4, st);
    const int n_sp = SP->n * slots;
    hipLaunchKernelGGL(k_lanes_scan, dim3((n_sp + LG_SCAN_EPW - 1) / LG_SCAN_EPW), dim3(WAVE), 0, st, SP->sp_ready, n_sp, q, list);
    if (bulk) waves = (n_sp + L - 1) / L;
    if (waves < 1) waves = 1;
    const dim3 grid(waves);
    if (ev0 || ev1) hipExtLaunchKernelGGL(k_regen_lanes, grid, dim3(WAVE), (uint32_t)p.smem, st, ev0, ev1, 0, *SP, *c, q, list, L, p.lane_stride, p.tab_off, p.stk_off, SP->maze_cap, slots, prio, prof);
    else hipLaunchKernelGGL(k_regen_lanes, grid, dim3(WAVE), p.smem, st, *SP, *c, q, list, L, p.lane_stride, p.tab_off, p.stk_off, SP->maze_cap, slots, prio, prof);
    return 1;
}
}
