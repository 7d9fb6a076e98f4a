// rg_state.h -- HBM layout of the batched Rogue-Gym state (shared by kernels and the C-ABI host).
//
// One environment per lane; 64 environments share a wavefront.  Scalars and entity tables are
// struct-of-arrays `[field][slot][env]` so a wavefront's loads of one field are one coalesced
// transaction; tile grids are `[env][cell]` (u16 per cell) so an env's 3x3 neighbourhood sits in
// <= 3 cache lines and the render / encode kernels stream rows with 16-byte accesses.
#pragma once
#include <cstdint>

#include "rg_config.h"

// ---- cell word (u16): surface | door | CellAttr | generator marks ----
#define C_SURF_MASK 0x0007u   // Surface enum order, rogue/mod.rs:137-147
#define C_DOOR      0x0008u   // member of Floor::doors (floor.rs:18)
#define C_ATTR_SHIFT 4        // CellAttr bits (field.rs:107-124) live in bits 4..9
#define C_VISITED   0x0010u
#define C_HIDDEN    0x0020u
#define C_VISIBLE   0x0040u
#define C_DRAWN     0x0080u
#define C_LOCKED    0x0100u
#define C_DARK      0x0200u
#define C_ATTR_MASK 0x03f0u
#define C_MAZE      0x0400u   // maze.passages membership (maze.rs:12-15)
#define C_GOLD      0x0800u   // Floor::items has an entry here (floor.rs:25)

enum { S_PASSAGE = 0, S_FLOOR, S_WALLX, S_WALLY, S_STAIR, S_DOOR, S_TRAP, S_NONE };

// room_meta bits
#define RM_KIND_MASK 0x03u    // 0 normal, 1 maze, 2 empty (rooms.rs:11-19)
#define RM_DARK      0x04u
#define RM_VISITED   0x08u
#define RM_HAS_GOLD  0x10u
enum { RK_NORMAL = 0, RK_MAZE = 1, RK_EMPTY = 2 };

// monster word 0: pos (x<<8|y, so the numeric order is the BTreeMap<[level,x,y]> order) | type<<16 | flags<<24
#define MF_ALIVE   0x01u
#define MF_ACTIVE  0x02u      // in active_enemies (and `running`)
#define MF_PENDING 0x04u      // active and not yet moved this turn (still in the taken map, enemies.rs:376-380)
// cache-only bits of the running turn (k_step's LDS monster cache); never stored to global memory
#define MF_RANDOM 0x08u
#define MF_DIR_SHIFT 4        // bits 4..6 of the flag byte
#define MF_REACH 0x80u        // stands next to the player and attacks at the end of this turn (actions::move_active_enemies, actions.rs:82-119)
#define MF_TURN_BITS (MF_PENDING | MF_RANDOM | (7u << MF_DIR_SHIFT) | MF_REACH)   // cache-only bits of the running turn; never stored to global memory

// Every connect_2rooms call joins a pair of grid-adjacent rooms that was not joined before (dig_passges excludes joined pairs, passages.rs:33-66),
// so a level has at most rnx*(rny-1) + rny*(rnx-1) < 2 * rooms corridors whatever max_extra_edges says.
#define RG_MAX_EDGES (2 * RG_MAX_ROOMS)   // (the tables are allocated per config: 2 x its rooms)
// words per grid row of a saved walkable mask (dc_walk) = the row words of the step class that builds partial maps (rg_kernels.hip bfs_rows_n32<2|3>)
#define RG_WALK_WORDS(w) ((w) <= 64 ? 2 : 3)
// ... and which configs step in that class: rows of <= 96 columns with H * W <= 4096, except the 32-column grids of <= 32 rooms (k_step_w32: whole maps)
#define RG_PARTIAL_MAPS(w, h, rooms) ((w) <= 96 && (w) * (h) <= 4096 && (rooms) <= 64 && !((w) <= 32 && (rooms) <= 32))

#define RG_STAT_COLS 16   /* counters per row of RgState::stats: one 128-byte line */
#define RG_NX_ASKED 1u    /* k_step: the player is next to the stairs, generate the structure (the request's streams and level are written, drained, then this) */
#define RG_NX_CLAIMED 6u  /* k_regen is generating it */
#define RG_NX_READY 3u    /* payload complete */
#define RG_NX_NONE 0u     /* nothing asked for this level yet */
#define RG_NX_DROP 4u     /* the env got a new level while k_regen was generating: k_regen finds this instead of its CLAIMED, drops the result and stores NONE.
                             (k_step's "new level" is ONE atomic AND with RG_NX_DROP: CLAIMED -> DROP, every other state -> NONE.  An env asks only from NONE /
                             READY, so never while a claim is outstanding: the request and the structure are never written by both sides at once.) */
#define RG_NX_STALE 8u    /* (k_step's registers only) READY, but made for a dungeon stream the env no longer holds */
#define RG_NX_HIT 9u      /* (k_step's registers only) READY and made for exactly the streams and level the env holds: this descent loads it */
struct RgNext {
    uint16_t *cell;      // [n][hw] the grid after the stair placement
    uint32_t *room_rect, *gold_pos, *gold_amt;  // [rooms][n]
    uint8_t *room_meta;  // [rooms][n]
    uint32_t *level;     // [n] the level the env was on when it asked
};
struct RgState {
    int32_t n;          // environments on this device
    int32_t hw;         // H*W
    // grids
    uint16_t *cell;     // [n][hw]
    uint8_t *screen;    // [n][hw]   PlayerState.map mirror
    uint8_t *hist;      // [n][hw]   PlayerState.history mirror
    // player + runtime scalars [n]
    uint16_t *p_pos;    // x<<8|y
    int32_t *p_hp, *p_hpmax, *p_lvl;
    uint32_t *p_exp, *food, *quiet, *pack_gold, *dlevel;
    uint32_t *steps, *flags;
    float *reward;
    uint8_t *done;      // [n] 1 = the last key ended the episode (is_terminal of the returned state)
    uint32_t *rng;      // [12][n]  dungeon{x,y,z,w}, item{..}, enemy{..}
    uint64_t *seed_lo, *seed_hi;  // [n] seed of the next build (reseed == 0), else the base every build's seed is derived from
    uint8_t *reseed;    // [n] 0 = `seed` given; 1 = no seed: a fresh one per build (rng::gen_seed); 2 = fresh one inside `seed_range`
                        //     (rng::gen_ranged_seed, core/src/lib.rs:157-165)
    uint32_t *build_ctr;          // [n] builds taken so far: build k of a reseed env uses hash(base, k); taken atomically, so k_step's inline
                                  //     generation and a concurrent k_regen never share or tear a seed
    const uint32_t *init_draws;   // [cfg.n_init_draws][2]: (lo, hi) of the item-stream draw of every InitItem::Weapon of player.init_items, in list order
                                  // (WeaponStatus::build, weapon.rs:159; resolved by rg_items.cpp); NULL when there is none
    uint64_t *range_lo, *range_span;  // [2][n] (low word, high word) of seed_range[0] and of seed_range[1] - seed_range[0]; NULL if no env has a range
    // rooms [rooms of the config][n]
    uint32_t *room_rect;  // x0 | y0<<8 | x1<<16 | y1<<24 (half-open; Empty: x0,y0 = up_left)
    uint8_t *room_meta;
    // monsters [rooms][n] (at most one spawn per room per level, floor.rs:106-130)
    uint32_t *mon_w0;
    int32_t *mon_hp;
    uint32_t *mon_exp;
    uint32_t *mon_cnt;  // [n] alive | active<<8
    // gold [rooms][n]: pos | 0x10000 when present; amounts
    uint32_t *gold_pos, *gold_amt;
    // generator scratch
    uint32_t *edge_a, *edge_b;  // [2 x rooms][n] corridor records, replayed for gen_attr (floor.rs:73-102)
    uint16_t *maze_stack;       // [n][maze_cap]: DFS stack of a maze room too large for the LDS stack
    int32_t maze_cap;           // >= the maze nodes of the largest assigned area (one stack entry per node at most)
    // DistCache (rogue/mod.rs:492-518)
    uint16_t *dc_map;   // [n][RG_DIST_SLOTS][hw], 0xFFFF = unreachable
    uint16_t *dc_key;   // [RG_DIST_SLOTS][n] target pos
    uint8_t *dc_head, *dc_len;  // [n] FIFO ring
    // partial maps (grids of 33..96 columns, rg_kernels.hip bfs_rows_n32): bit s of dc_part = slot s holds a map that was not expanded to the end;
    // bit s of dc_own = its walkable mask was saved into dc_walk when the level's cells were about to change
    uint16_t *dc_part, *dc_own;  // [n]
    uint32_t *dc_walk;           // [n][RG_DIST_SLOTS][H][RG_WALK_WORDS(W)] or null
    int keep_spares;             // ROGUE_GYM_HIP_KEEP_SPARES=1: the spare of a FIXED-seed env is not consumed by a reset (its level-1 state is a pure function of config and seed)
    int full_bfs;                // ROGUE_GYM_HIP_FULL_BFS=1: every map is expanded to the end (the A side of tests/test_gpu_features.py::test_partial_dist_maps_*)
    // optional in-kernel phase profile (development aid): [2][32] u64 = {max cycles, sum cycles} per phase, NULL = off
    unsigned long long *prof;
    // spare-level pipeline: 0 = spare must be (re)generated, 1 = ready, 2 = generation in progress
    uint32_t *sp_ready; // [sp_slots][n] (shared by the live and the spare view)
    // Spares per env.  1: the wave-per-level producer (k_regen), whose spare is back one or two steps after it was consumed.  2: the level-per-lane producer
    // (rg_regen_lanes.hip) builds 64 levels per wave in the time the other builds a dozen, but a round of 64 takes it four or five steps -- and under the
    // random policy 1 % of the episodes are over within five steps: with a second spare per env a reset misses only when TWO episodes in a row end inside
    // the producer's latency.  Spare (slot s, env e) is entry s * n + e of every array of the spare view (whose SoA stride is sp_slots * n).
    int32_t sp_slots;
    // status mirror
    int32_t *status;    // [n][10]
    // observation records [n][RG_OBS_REC_WORDS(rooms)] (rg_obs.hip ObsTabs): what the fused observation pass overlays on an ordinary Redraw, one line per env.
    // Follows the env's tables: k_step writes the monster words and the player's position of every env whose key produced a Redraw, the generator's
    // copy-out (gen_service) and the spare hand-off (take_spares) the rooms and the new level's monsters.  NULL: more rooms than the fused pass handles
    uint32_t *obs_rec;
    // [rooms + 1][n]: where the overlays of the env's screen mirror stand -- the monsters' and (last row) the player's position as of the env's last Redraw,
    // 0xFFFF = none: what k_step's incremental mirror update restores before it draws the overlays anew.  NULL: more than RG_OVL_MAX rooms.
    uint16_t *ovl;
    // A bound observation tensor (rg_obs_bind): the envs whose screen this k_step changed -- final flag word with REDRAW or SCR_CHANGED -- appended to
    // obs_list[obs_par][..] (one atomic per wave; count in obs_cnt[obs_par]), for the in-place observation pass right behind it (k_obs<.., BOUND>), which
    // zeroes the other counter.  NULL = no bound tensor.
    float *bound_gray;   // the bound tensor itself when it is a GRAY image [n][1][H][W]: the turn's incremental mirror update writes the pixels it changes straight
                         // into it (k_step mirror_update), so such an env needs no observation pass at all; NULL otherwise
    const float *gray_lut;  // [128] glyph -> gray value as the observation pass encodes it (rg_obs_bind fills it on the host: symbol id / symbols, one IEEE division)
    int32_t *obs_list;   // [2][n]
    uint32_t *obs_cnt;   // [2]
    int32_t obs_par;     // which half this launch writes (set by the host before every k_step)
    // action-history log (RunTime::saved_inputs, core/src/lib.rs:288): the keys of the current and of the previous episode, NULL = off
    uint8_t *klog;      // [n][2][klog_cap]
    uint32_t *klog_len; // [2][n] keys accepted in episode buffer 0 / 1 (may exceed klog_cap: the tail is then not stored)
    uint8_t *klog_cur;  // [n] which buffer holds the running episode
    int32_t klog_cap;
    // workload counters (bench.py: resets/s, descents/s, BFS maps/s): [0] auto-resets [1] descents [2] dist maps built [3] inline level
    // generations [4] spare levels taken [5] Redraw reactions [6] keys processed; one atomicAdd per wave and counter
    unsigned long long *stats;  // [rows][RG_STAT_COLS] (a row per block); [8] descents that loaded their next-level structure (counted in [3] as well)
    // Envs whose player stands on the staircase, for k_step's stair waves.  Whoever changes player positions (k_build, k_step, the debug descent)
    // PRODUCES the set for the k_step after it: a byte per env + the list of the marked envs, double-buffered, with three rotating counters
    // (producer number g reads set g & 1 / counter g % 3, writes set (g + 1) & 1 / counter (g + 1) % 3, zeroes counter (g + 2) % 3) --
    // so nothing a launch reads is ever written during that launch.
    uint8_t *stair_mark;   // [2][n]
    int32_t *stair_list;   // [2][n]
    uint32_t *stair_cnt;   // [8]: [0..2] the rotating entry counts, [4..6] k_step's take counters for entries beyond its first STAIR_BLOCKS
    int32_t stair_gen;     // producers launched so far (set by the host before every producer launch)
    uint32_t *launch_mark; // [1] stair_gen + 1 of the newest k_step that has STARTED (block 0 publishes it): what the generator's gate kernel waits for
    uint8_t *on_stairs;    // [n] spare view only: the pre-generated state's player stands on the stairs
    // Next-level structures (rg_kernels.hip "next-level structures"): the part of Dungeon::new_level that only draws on the dungeon and item streams
    // (rooms, passages, gold, stairs), generated AHEAD of the descent by k_regen from the streams' state when the player came near the stairs, and
    // valid at the descent iff the dungeon stream is still where the structure started from (nx_rng[0..3] == the env's rd, nx_level == its level).
    // NULL = off (no auto-reset spares, more than 64 rooms, ROGUE_GYM_HIP_NO_NEXT_LEVELS).  Shared by the live and the spare view.
    uint32_t *nx_state;     // [n] RG_NX_*
    uint32_t *nx_rng;       // [12][n]: [0..3] the dungeon stream the structure starts from (the key, written with the request), [4..7] the dungeon stream
                            // after it, [8..11] the item stream (the request's, then the one after it)
    const struct RgNext *nx;  // the rest of it, device-resident (read on the rare paths only: a second dozen of pointers in the kernel argument costs the capped
                              // step kernel registers it does not have)
    // handle with per-env configs that differ in more than the seed: this RgState is one config GROUP, and env e of the group is env ext[e] of the
    // handle (observation tensors are written at the handle's index); NULL = the group is the whole handle
    const int32_t *ext;
    uint32_t *err_any;  // [1] OR of every error bit raised since the last rg_sync
    // ThreadConductor::step zips keys with envs (thread_impls.rs:62-64): envs >= n_keys receive no key this call
    int32_t n_keys;
    // StairRewardParallel (python/rogue_gym/envs/wrappers.py:45-64) inside the step: added to reward[e] whenever the level the env reports after the key is
    // above the one it reported one step earlier (rg_set_stair_reward; 0 = off)
    float stair_reward;
};
