/*
 * rogue_oracle.c -- CPU restatement of the kngwyu/rogue-gym engine (TEST INFRASTRUCTURE ONLY).
 * See rogue_oracle.h for the role of this file and how it is pinned.
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Clarity over speed: sets are byte arrays with linear `nth`, BFS is a FIFO queue, monsters are
 * kept in two (asleep/active) arrays sorted on demand -- the same observable semantics as the
 * reference's FenwickSet / VecDeque / BTreeMap.
 */
#include "rogue_oracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* RNG: rand_xorshift 0.2 XorShiftRng + rand 0.7 UniformInt::sample_single (un-vendored crates;
 * semantics per SURVEY.md Appendix A, call sites core/src/rng.rs:48-98).                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t x, y, z, w; uint64_t count; } rng_t;

static void rng_seed(rng_t *r, uint64_t lo, uint64_t hi) { /* rng.rs:48-55 (u128 -> 16 LE bytes) */
    r->x = (uint32_t)lo; r->y = (uint32_t)(lo >> 32);
    r->z = (uint32_t)hi; r->w = (uint32_t)(hi >> 32);
    if ((r->x | r->y | r->z | r->w) == 0) r->x = r->y = r->z = r->w = 0x0BAD5EEDu;
    r->count = 0;
}
static uint32_t rng_u32(rng_t *r) {
    uint32_t t = r->x ^ (r->x << 11);
    r->x = r->y; r->y = r->z; r->z = r->w;
    r->w = r->w ^ (r->w >> 19) ^ (t ^ (t >> 8));
    r->count++;
    return r->w;
}
static uint64_t rng_u64(rng_t *r) { /* next_u64_via_u32: low word first */
    uint64_t lo = rng_u32(r), hi = rng_u32(r);
    return (hi << 32) | lo;
}
/* gen_range for u32 / i32 call sites (32-bit widening multiply + rejection zone) */
static uint32_t range32(rng_t *r, uint32_t low, uint32_t high) {
    uint32_t range = high - low;
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint64_t m = (uint64_t)rng_u32(r) * (uint64_t)range;
        if ((uint32_t)m <= zone) return low + (uint32_t)(m >> 32);
    }
}
static int32_t __attribute__((unused)) range_i32(rng_t *r, int32_t low, int32_t high) {
    return (int32_t)range32(r, (uint32_t)low, (uint32_t)high);
}
/* gen_range for usize / i64 call sites (64-bit) */
static uint64_t range64(rng_t *r, uint64_t low, uint64_t high) {
    uint64_t range = high - low;
    uint64_t zone = (range << __builtin_clzll(range)) - 1ull;
    for (;;) {
        unsigned __int128 m = (unsigned __int128)rng_u64(r) * range;
        if ((uint64_t)m <= zone) return low + (uint64_t)(m >> 64);
    }
}
static int __attribute__((unused)) does_happen(rng_t *r, uint32_t p_inv) { return range32(r, 0, p_inv) == 0; } /* rng.rs:91 (call sites: DH) */
static int __attribute__((unused)) parcent(rng_t *r, uint32_t p) { return range32(r, 1, 101) <= p; } /* rng.rs:95 (the call sites carry their mutant number: PC) */

/* ------------------------------------------------------------------------------------------ */
/* Mutation testing (tests/test_oracle_mutations.py).  -DORC_MUTANT=k builds this file with ONE  */
/* plausible misreading of the reference at RNG call site / quirk k; the   */
/* test runs the reference's goldens against every mutant and records which golden notices --     */
/* the sites nobody notices are the ones where "HIP == oracle" rests on the reading alone          */
/* (profiles/r05_pin_map.txt).  0 = the restatement itself; the product never sees any of this.    */
/* ------------------------------------------------------------------------------------------ */
#ifndef ORC_MUTANT
#define ORC_MUTANT 0
#endif
#define MUT(k) (ORC_MUTANT == (k))
int orc_mutant_id(void) { return ORC_MUTANT; }
/* the reference samples a 32-bit type at site k (one next_u32 per attempt); the mutant samples 64 bits -- and the other way round */
static uint32_t w32(int mut, rng_t *r, uint32_t lo, uint32_t hi) { return mut ? (uint32_t)range64(r, lo, hi) : range32(r, lo, hi); }
static uint64_t w64(int mut, rng_t *r, uint64_t lo, uint64_t hi) { return mut ? (uint64_t)range32(r, (uint32_t)lo, (uint32_t)hi) : range64(r, lo, hi); }
#define W32(k, r, lo, hi) w32(MUT(k), (r), (uint32_t)(lo), (uint32_t)(hi))
#define W64(k, r, lo, hi) w64(MUT(k), (r), (uint64_t)(lo), (uint64_t)(hi))
#define DH(k, r, p) (W32(k, r, 0, p) == 0)       /* does_happen at site k */
#define PC(k, r, p) (W32(k, r, 1, 101) <= (p))   /* parcent at site k */

/* ------------------------------------------------------------------------------------------ */
/* geometry: rect-iter RectRange<i32> (half-open, row-major x fastest),
 * Direction enum order (dungeon/coord.rs:198-242)                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int x0, y0, x1, y1; } rect_t;
static int rect_contains(const rect_t *r, int x, int y) { return x >= r->x0 && x < r->x1 && y >= r->y0 && y < r->y1; }
static int rect_xlen(const rect_t *r) { return r->x1 - r->x0; }
static int rect_len(const rect_t *r) { return (r->x1 - r->x0) * (r->y1 - r->y0); }
static int rect_index(const rect_t *r, int x, int y) { return rect_contains(r, x, y) ? (y - r->y0) * rect_xlen(r) + (x - r->x0) : -1; }
static void rect_nth(const rect_t *r, int n, int *x, int *y) { *x = r->x0 + n % rect_xlen(r); *y = r->y0 + n / rect_xlen(r); }
static int rect_is_horiz_edge(const rect_t *r, int y) { return y == r->y0 || y == r->y1 - 1; }
static int rect_is_vert_edge(const rect_t *r, int x) { return x == r->x0 || x == r->x1 - 1; }
static int rect_valid(const rect_t *r) { return r->x0 < r->x1 && r->y0 < r->y1; }

enum { D_UP, D_DOWN, D_LEFT, D_RIGHT, D_LEFTUP, D_RIGHTUP, D_LEFTDOWN, D_RIGHTDOWN, D_STAY };
static const int DX[9] = {0, 0, -1, 1, -1, 1, -1, 1, 0};
static const int DY[9] = {-1, 1, 0, 0, -1, -1, 1, 1, 0};
static int dir_reverse(int d) { static const int R[9] = {D_DOWN, D_UP, D_RIGHT, D_LEFT, D_RIGHTDOWN, D_LEFTDOWN, D_RIGHTUP, D_LEFTUP, D_STAY}; return R[d]; }
static int dir_is_diag(int d) { return d >= D_LEFTUP && d <= D_RIGHTDOWN; }

/* Surface (rogue/mod.rs:137-183) and CellAttr (field.rs:107-124) */
enum { S_PASSAGE, S_FLOOR, S_WALLX, S_WALLY, S_STAIR, S_DOOR, S_TRAP, S_NONE };
static const uint8_t SURFACE_GLYPH[8] = {'#', '.', '-', '|', '%', '+', '^', ' '};
static int can_walk(int s) { return !(s == S_WALLX || s == S_WALLY || s == S_NONE); }
enum { A_VISITED = 1, A_HIDDEN = 2, A_VISIBLE = 4, A_DRAWN = 8, A_LOCKED = 16, A_DARK = 32 };

/* ------------------------------------------------------------------------------------------ */
/* monsters: BUILTIN_ENEMIES (character/enemies.rs:474-761)                                   */
/* ------------------------------------------------------------------------------------------ */
enum { EA_MEAN = 1, EA_FLYING = 2, EA_REGENERATE = 4, EA_GREEDY = 8, EA_INVISIBLE = 16, EA_RUSTS = 32,
       EA_STEAL_GOLD = 64, EA_REDUCE_STR = 128, EA_FREEZES = 256, EA_RANDOM = 512, EA_CONFUSED = 1024 };
typedef struct { int n_attack; int att_times[4]; int att_max[4]; int attr; int defense; uint32_t exp; int level; int rarity; uint8_t tile; } mstat_t;
static const mstat_t BUILTIN[26] = {
    /* A aquator   */ {1, {0}, {0}, EA_MEAN | EA_RUSTS, 2 | 8, 20, 5, 12, 'A'},
    /* B bat       */ {1, {1}, {2}, EA_FLYING | EA_RANDOM, 3, 1, 1, 2, 'B'},
    /* C centaur   */ {3, {1, 1, 1}, {2, 5, 5}, 0, 4, 17, 4, 10, 'C'},
    /* D dragon    */ {3, {1, 1, 3}, {8, 8, 10}, EA_MEAN, 3, 5000, 10, 25, 'D'},
    /* E emu       */ {1, {1}, {2}, EA_MEAN, 7, 2, 1, 1, 'E'},
    /* F flytrap   */ {0, {0}, {0}, EA_MEAN, 3, 80, 8, 15, 'F'},
    /* G griffin   */ {2, {4, 3}, {3, 5}, EA_FLYING | EA_MEAN | EA_REGENERATE, 2, 2000, 13, 23, 'G'},
    /* H hobgoblin */ {1, {1}, {8}, EA_MEAN, 5, 3, 1, 4, 'H'},
    /* I icemonster*/ {1, {0}, {0}, EA_FREEZES, 9, 5, 1, 5, 'I'},
    /* J jabberwock*/ {2, {2, 2}, {12, 4}, 0, 6, 3000, 15, 24, 'J'},
    /* K kestrel   */ {1, {1}, {4}, EA_MEAN, 7, 1, 1, 0, 'K'},
    /* L leprechaun*/ {1, {1}, {1}, EA_STEAL_GOLD, 8, 10, 3, 9, 'L'},
    /* M medusa    */ {3, {3, 3, 2}, {4, 4, 5}, EA_MEAN, 2, 200, 8, 21, 'M'},
    /* N nymph     */ {1, {0}, {0}, 0, 9, 37, 3, 13, 'N'},
    /* O orc       */ {1, {1}, {8}, EA_GREEDY, 6, 5, 1, 7, 'O'},
    /* P phantom   */ {1, {4}, {4}, EA_INVISIBLE, 3, 120, 8, 18, 'P'},
    /* Q quagga    */ {2, {1, 1}, {5, 5}, EA_MEAN, 3, 15, 3, 11, 'Q'},
    /* R rattlesnk */ {1, {1}, {6}, EA_REDUCE_STR | EA_MEAN, 3, 9, 2, 6, 'R'},
    /* S snake     */ {1, {1}, {3}, EA_MEAN, 5, 2, 1, 3, 'S'},
    /* T troll     */ {3, {1, 1, 2}, {8, 8, 6}, EA_MEAN | EA_REGENERATE, 4, 120, 6, 16, 'T'},
    /* U urvile    */ {3, {1, 1, 2}, {9, 9, 9}, EA_MEAN, -2, 190, 7, 20, 'U'},
    /* V vampire   */ {1, {1}, {19}, EA_MEAN | EA_REGENERATE, 1, 350, 8, 22, 'V'},
    /* W wraith    */ {1, {1}, {6}, 0, 4, 55, 5, 17, 'W'},
    /* X xeroc     */ {1, {4}, {4}, 0, 7, 100, 7, 19, 'X'},
    /* Y yeti      */ {2, {1, 1}, {6, 6}, 0, 6, 50, 4, 14, 'Y'},
    /* Z zombie    */ {1, {1}, {8}, EA_MEAN, 8, 6, 2, 8, 'Z'},
};

typedef struct {
    int x, y, type;
    int running;
    int64_t hp, max_hp, level;
    int defense;
    uint32_t exp;
} mon_t;

/* ------------------------------------------------------------------------------------------ */
/* rooms / floor / dungeon / player / env                                                     */
/* ------------------------------------------------------------------------------------------ */
enum { RK_NORMAL, RK_MAZE, RK_EMPTY };
typedef struct {
    int kind, is_dark, is_visited, has_gold, id;
    rect_t assigned;
    rect_t range;              /* Normal: room rect; Maze: maze.range */
    int upx, upy;              /* Empty: up_left */
    int cap;                   /* capacity of the index sets below (= len(range), or 1) */
    uint8_t *empty_cells;      /* membership by range index  (rooms.rs:36-40) */
    uint8_t *nochar_cells;
    uint8_t *maze_passages;    /* Maze only */
} room_t;

#define MAX_ROOMS 384  /* rooms.rs:165-211 has no limit; geometry does: 160 x 48 with min_room_size 3 holds at most 40 x 9 = 360 rooms (stack scratch only:
                        * the per-level room table and the monster tables are sized by the config) */
#define MAX_MON MAX_ROOMS
#define DIST_INF 0xFFFFFFFFu
#define DIST_CACHE_CAP 10

typedef struct { int x, y, kind; } pcell_t; /* Positioned<Surface> */

typedef struct {
    int n_rooms;
    room_t *rooms;             /* [n_rooms], allocated per level (gen_rooms) */
    uint8_t *surface, *attr, *doors;
    int32_t *gold;      /* amount or -1 (Floor.items, floor.rs:25) */
    uint8_t *non_empty;        /* [n_rooms] */
} floor_t;

typedef struct { uint32_t *map; int kx, ky; } dcache_t;

/* Reaction list (core/src/lib.rs:378-403) and message flags (python/src/flags.rs:6-39) */
enum { RE_REDRAW = 1, RE_STATUS = 2, RE_GRAVE = 4, RE_NOTIFY = 8 };
enum { MSG_HIT_FROM = 1, MSG_HIT_TO = 2, MSG_MISS_TO = 4, MSG_MISS_FROM = 8, MSG_KILLED = 16, MSG_SECRET_DOOR = 32, MSG_NO_DOWNSTAIR = 64 };
typedef struct { uint8_t kind, msg; } reaction_t;
#define RLIST_CAP 2048 /* a run is at most W+H iterations of a few reactions each */
typedef struct { reaction_t v[RLIST_CAP]; int n; } rlist_t;
static __thread rlist_t tls_scratch_a, tls_scratch_b; /* reaction lists of the current react (per thread, no per-step malloc) */
static void rpush(rlist_t *l, int kind, uint32_t msg) { if (l->n < RLIST_CAP) l->v[l->n++] = (reaction_t){(uint8_t)kind, (uint8_t)msg}; }



struct orc_env {
    orc_config cfg;
    uint64_t max_steps;
    int W, H, symbols;
    /* RunTime parts */
    rng_t rng_d, rng_i, rng_e;
    uint32_t level;
    floor_t fl;
    uint8_t **past_visited; int n_past, cap_past; /* past_floors' history maps (rogue/mod.rs:329-338) */
    dcache_t dcache[DIST_CACHE_CAP]; int n_dcache;
    /* enemies */
    mstat_t stats[32]; int n_stats;       /* monster statuses sorted by rarity (enemies.rs:250-261); mon_t.type indexes this */
    mon_t *placed; int n_placed;  /* asleep; [rooms] each: at most one spawn per room per level (floor.rs:106-130) */
    mon_t *active; int n_active;
    /* player (player.rs:280-306) */
    int px, py;
    int64_t hp, hp_max, plevel;
    uint32_t exp, food_left, quiet;
    /* ItemBox (itembox.rs:8-12): slot ch holds pack[ch]; slots fill from 0 upwards and nothing reachable ever removes an item, so the
     * BTreeMap<usize, ItemToken> is the array prefix.  weapon / armor: the equipped tokens (player.rs:95-96), as pack slots. */
    orc_item pack[ORC_MAX_INIT_ITEMS + 1]; int n_pack;
    int weapon, armor;
    int dead; /* ui == Grave */
    /* GameStateImpl + PlayerState mirror (state_impls.rs, python/src/lib.rs:29-38) */
    uint64_t steps;
    uint8_t *screen, *hist;
    uint32_t status[10];
    uint32_t message;
    int is_terminal;
};

static inline int IDX(const orc_env *e, int x, int y) { return y * e->W + x; }
/* Field::try_get_xy bounds (field.rs:163-175). The reference accepts x == width / y == height
 * (off-by-one); those coordinates are unreachable for in-room positions, so the
 * restatement uses exact bounds. */
static inline int INB(const orc_env *e, int x, int y) { return x >= 0 && y >= 0 && x < e->W && y < e->H; }

void orc_config_default(orc_config *c) {
    memset(c, 0, sizeof *c);
    c->width = 80; c->height = 24; c->hide_dungeon = 1;
    c->room_num_x = 3; c->room_num_y = 3; c->min_room_x = 4; c->min_room_y = 4;
    c->max_empty_rooms = 3; c->amulet_level = 25; c->maze_rate_inv = 15; c->dark_level = 10;
    c->hidden_passage_rate_inv = 40; c->locked_door_rate_inv = 5; c->max_extra_edges = 5;
    c->door_unlock_rate_inv = 5; c->passage_unlock_rate_inv = 3;
    c->gold_rate_inv = 2; c->gold_base = 50; c->gold_per_level = 10; c->gold_minimum = 2;
    c->hunger_time = 1300; c->init_hp = 12;
    c->appear_rate_gold = 80; c->appear_rate_nogold = 25;
    c->n_enemies = 26;
    for (int i = 0; i < 26; i++) c->enemy_builtin[i] = i;
    c->choose_width = 64;
    /* weapon::Config::default / armor::Config::default: every builtin preset in order (weapon.rs:66-68, armor.rs:34-36) */
    static const struct { const char *name; int t, m; uint32_t lo, hi, attr; } BW[9] = { /* BUILTIN_WEAPONS (weapon.rs:198-298) */
        {"mace", 2, 4, 1, 2, 0}, {"long-sword", 3, 4, 1, 2, 0}, {"bow", 1, 1, 1, 2, 0}, {"arrow", 1, 1, 8, 17, 6}, {"dagger", 1, 6, 2, 7, 2},
        {"two-handed-sword", 4, 4, 1, 2, 0}, {"dart", 1, 1, 8, 17, 6}, {"shuriken", 1, 2, 8, 17, 6}, {"spear", 2, 3, 8, 17, 4}};
    static const struct { const char *name; int def; } BA[8] = { /* BUILTIN_ARMORS (armor.rs:170-219) */
        {"leather armor", 2}, {"ring mail", 3}, {"studded leather armor", 3}, {"scale mail", 4}, {"chain mail", 5}, {"splint mail", 6},
        {"banded mail", 6}, {"plate mail", 7}};
    c->n_weapons = 9; c->n_armors = 8;
    for (int i = 0; i < 9; i++) {
        orc_weapon_stat *w = &c->weapons[i];
        strncpy(w->name, BW[i].name, ORC_NAME_CAP - 1); w->wield_times = (uint64_t)BW[i].t; w->wield_max = BW[i].m;
        w->init_lo = BW[i].lo; w->init_hi = BW[i].hi; w->attr = BW[i].attr;
    }
    for (int i = 0; i < 8; i++) { strncpy(c->armors[i].name, BA[i].name, ORC_NAME_CAP - 1); c->armors[i].def = BA[i].def; }
    /* default_init_items (player.rs:68-75): 0 gold, a food ration, ring mail +1 (armor.rs:68-73), mace +1,+1, bow +1,+0, arrows +25 (weapon.rs:179-188) */
    c->max_items = 27;
    c->n_init_items = 6;
    c->init_items[0].tag = ORC_INIT_NOINIT; c->init_items[0].item.kind = ORC_KIND_GOLD; c->init_items[0].item.how_many = 0; c->init_items[0].item.attr = ORC_ATTR_IS_MANY;
    c->init_items[1].tag = ORC_INIT_NOINIT; c->init_items[1].item.kind = ORC_KIND_FOOD; c->init_items[1].item.how_many = 1; c->init_items[1].item.attr = ORC_ATTR_IS_MANY;
    c->init_items[2].tag = ORC_INIT_ARMOR; strcpy(c->init_items[2].name, "ring mail"); c->init_items[2].def_plus = 1;
    c->init_items[3].tag = ORC_INIT_WEAPON; strcpy(c->init_items[3].name, "mace"); c->init_items[3].hit_plus = 1; c->init_items[3].dam_plus = 1;
    c->init_items[4].tag = ORC_INIT_WEAPON; strcpy(c->init_items[4].name, "bow"); c->init_items[4].hit_plus = 1;
    c->init_items[5].tag = ORC_INIT_WEAPON; strcpy(c->init_items[5].name, "arrow"); c->init_items[5].num_plus = 25;
}

/* ------------------------------------------------------------------------------------------ */
/* room sets                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static int set_len(const uint8_t *s, int cap) { int n = 0; for (int i = 0; i < cap; i++) n += s[i]; return n; }
static int set_nth(const uint8_t *s, int cap, int n) { /* FenwickSet::nth (fenwick.rs:75-82) */
    for (int i = 0; i < cap; i++) if (s[i]) { if (n == 0) return i; n--; }
    return -1;
}
/* FenwickSet::select (fenwick.rs:88-94): usize sample => 64-bit */
static int set_select_m(const uint8_t *s, int cap, rng_t *r, int mut) {
    int n = set_len(s, cap);
    if (n == 0) return -1;
    return set_nth(s, cap, (int)w64(mut, r, 0, (uint64_t)n));
}


static void room_free(room_t *rm) { free(rm->empty_cells); free(rm->nochar_cells); free(rm->maze_passages); rm->empty_cells = rm->nochar_cells = rm->maze_passages = NULL; }

/* Room::new + gen_empty_cells (rooms.rs:43-56,147-162) */
static void room_init_sets(room_t *rm) {
    if (rm->kind == RK_EMPTY) { rm->cap = 1; rm->empty_cells = calloc(1, 1); rm->nochar_cells = calloc(1, 1); return; }
    rm->cap = rect_len(&rm->range);
    rm->empty_cells = calloc(rm->cap, 1);
    rm->nochar_cells = calloc(rm->cap, 1);
    if (rm->kind == RK_NORMAL) {
        for (int i = 0; i < rm->cap; i++) {
            int x, y; rect_nth(&rm->range, i, &x, &y);
            if (!(rect_is_horiz_edge(&rm->range, y) || rect_is_vert_edge(&rm->range, x))) rm->empty_cells[i] = 1;
        }
    } else {
        memcpy(rm->empty_cells, rm->maze_passages, rm->cap);
    }
    memcpy(rm->nochar_cells, rm->empty_cells, rm->cap);
}
/* Room::select_cell (rooms.rs:126-144) */
static int room_select_cell(const room_t *rm, rng_t *r, int is_character, int *x, int *y) {
    if (rm->kind == RK_EMPTY) return 0;
    int n = set_select_m(is_character ? rm->nochar_cells : rm->empty_cells, rm->cap, r, MUT(24)); /* M24 rooms.rs:134 */
    if (n < 0) return 0;
    rect_nth(&rm->range, n, x, y);
    return 1;
}
/* Room::fill_cell / unfill_cell (rooms.rs:96-116) */
static int room_fill(room_t *rm, int x, int y, int is_character) {
    if (rm->kind == RK_EMPTY) return 0;
    int id = rect_index(&rm->range, x, y);
    if (id < 0) return 0;
    if (is_character) rm->nochar_cells[id] = 0;
    int was = rm->empty_cells[id]; rm->empty_cells[id] = 0; return was;
}
static int room_unfill(room_t *rm, int x, int y, int is_character) {
    if (rm->kind == RK_EMPTY) return 0;
    int id = rect_index(&rm->range, x, y);
    if (id < 0) return 0;
    if (is_character) rm->nochar_cells[id] = 1;
    int was = rm->empty_cells[id]; rm->empty_cells[id] = 1; return !was;
}

/* ------------------------------------------------------------------------------------------ */
/* maze (rogue/maze.rs:38-89): recursive DFS on the 2-step lattice                            */
/* ------------------------------------------------------------------------------------------ */
static void dig_impl(const rect_t *range, rng_t *r, uint8_t *used /* by range index */, int cx, int cy) {
    for (;;) {
        int dig = -1, i = 0;
        for (int d = 0; d < 4; d++) { /* Direction::into_enum_iter().take(4) */
            int nx = cx + 2 * DX[d], ny = cy + 2 * DY[d];
            if (!rect_contains(range, nx, ny) || used[rect_index(range, nx, ny)]) continue;
            if (DH(9, r, (uint32_t)i + 1) && !(MUT(10) && dig >= 0)) dig = d; /* .enumerate().filter(does_happen(i+1)).last()   M9 maze.rs:73 width, M10 first instead of last */
            i++;
        }
        if (dig < 0) break;
        for (int k = 1; k <= 2; k++) { /* direc_iter(..).skip(1).take(2): `used.insert` + register */
            int x = cx + k * DX[dig], y = cy + k * DY[dig];
            used[rect_index(range, x, y)] = 1;
        }
        dig_impl(range, r, used, cx + 2 * DX[dig], cy + 2 * DY[dig]);
    }
}
static void dig_maze(const rect_t *range, rng_t *r, uint8_t *passages) {
    passages[rect_index(range, range->x0, range->y0)] = 1; /* start = lower_left */
    dig_impl(range, r, passages, range->x0, range->y0);     /* registered set == used set */
}

/* ------------------------------------------------------------------------------------------ */
/* rooms (rogue/rooms.rs:165-269)                                                             */
/* ------------------------------------------------------------------------------------------ */
static void make_room(orc_env *e, room_t *rm, int is_empty, int rsx, int rsy, int llx, int lly, int id, uint32_t level) {
    const orc_config *c = &e->cfg; rng_t *r = &e->rng_d;
    memset(rm, 0, sizeof *rm);
    rm->id = id;
    rm->assigned = (rect_t){llx, lly, llx + rsx, lly + rsy};
    if (is_empty) { /* rooms.rs:224-236: x then y (TupleMap2::map order) */
        int x, y; /* M3 rooms.rs:226 width, M4 y drawn before x */
        if (MUT(4)) { y = (int)W32(3, r, 1, rsy - 1) + lly; x = (int)W32(3, r, 1, rsx - 1) + llx; }
        else { x = (int)W32(3, r, 1, rsx - 1) + llx; y = (int)W32(3, r, 1, rsy - 1) + lly; }
        rm->kind = RK_EMPTY; rm->is_dark = 1; rm->upx = x; rm->upy = y;
        room_init_sets(rm);
        return;
    }
    rm->is_dark = W32(5, r, 0, c->dark_level) < level; /* M5 rooms.rs:237 width */
    int maze_roll = MUT(52) ? DH(6, r, c->maze_rate_inv) : 0; /* M52: the maze draw taken whether or not the room is dark */
    if (rm->is_dark && (MUT(52) ? maze_roll : DH(6, r, c->maze_rate_inv))) { /* M6 rooms.rs:238 width */
        rm->kind = RK_MAZE;
        rm->range = (rect_t){llx, lly, llx + rsx - 1, lly + rsy - 1};
        rm->maze_passages = calloc(rect_len(&rm->range), 1);
        dig_maze(&rm->range, r, rm->maze_passages);
    } else {
        int sx, sy, ox, oy; /* M7 rooms.rs:258,262 width; M8 size x, offset x, size y, offset y instead of both sizes first */
        if (MUT(8)) { sx = (int)W32(7, r, c->min_room_x, rsx); ox = (int)W32(7, r, 0, rsx - sx) + llx; sy = (int)W32(7, r, c->min_room_y, rsy); oy = (int)W32(7, r, 0, rsy - sy) + lly; }
        else { sx = (int)W32(7, r, c->min_room_x, rsx); sy = (int)W32(7, r, c->min_room_y, rsy); ox = (int)W32(7, r, 0, rsx - sx) + llx; oy = (int)W32(7, r, 0, rsy - sy) + lly; }
        rm->kind = RK_NORMAL;
        rm->range = (rect_t){ox, oy, ox + sx, oy + sy};
    }
    room_init_sets(rm);
}

static void gen_rooms(orc_env *e, floor_t *fl, uint32_t level) {
    const orc_config *c = &e->cfg; rng_t *r = &e->rng_d;
    int rnx = c->room_num_x, rny = c->room_num_y, room_num = rnx * rny;
    int rsx0 = e->W / rnx, rsy0 = e->H / rny;
    uint32_t empty_num = W32(1, r, 0, c->max_empty_rooms + (MUT(53) ? 0 : 1)); /* M1 rooms.rs:179 width; M53 0..max instead of 0..=max */
    if (empty_num >= (uint32_t)room_num) empty_num = room_num - 1;
    uint8_t is_empty[MAX_ROOMS] = {0};
    { /* rng.select(0..room_num).take(empty_num) (rng.rs:59-73,121-143) */
        uint8_t sel[MAX_ROOMS];
        for (int i = 0; i < room_num; i++) sel[i] = 1;
        for (uint32_t k = 0; k < empty_num; k++) {
            int rest = set_len(sel, room_num);
            int n = (int)W64(2, r, 0, (uint64_t)rest); /* M2 rooms.rs:189 width */
            int id = set_nth(sel, room_num, n);
            sel[id] = 0; is_empty[id] = 1;
        }
    }
    fl->n_rooms = room_num;
    fl->rooms = calloc((size_t)room_num, sizeof(room_t));
    fl->non_empty = calloc((size_t)room_num, 1);
    for (int i = 0; i < room_num; i++) {
        int x = i % rnx, y = i / rnx;
        int rsx = rsx0, rsy = rsy0, llx, lly;
        if (y == 0) { rsy -= 1; llx = rsx * x; lly = 1; }
        else { llx = rsx * x; lly = rsy * y; }
        if (lly + rsy == e->H) rsy -= 1;
        make_room(e, &fl->rooms[i], is_empty[i], rsx, rsy, llx, lly, i, level);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* passages (rogue/passages.rs:16-270)                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { pcell_t *v; int n, cap; } plist_t;
static void plist_push(plist_t *p, int x, int y, int kind) {
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 256; p->v = realloc(p->v, p->cap * sizeof(pcell_t)); }
    p->v[p->n++] = (pcell_t){x, y, kind};
}
static uint64_t choose_index(orc_env *e, int len) { /* SliceRandom::choose (passages.rs:146,156) */
    if (e->cfg.choose_width == 32) return range32(&e->rng_d, 0, (uint32_t)len);
    return W64(20, &e->rng_d, 0, (uint64_t)len); /* M20 passages.rs:146,156 width */
}
/* edges() (passages.rs:181-219) */
static int edges(const rect_t *range, int direction, int inclusive, int *xs, int *ys) {
    int off = inclusive ? 1 : 0, n = 0;
    int bx = range->x1 - off, by = range->y1 - off;
    switch (direction) {
    case D_DOWN:  for (int x = range->x0 + off; x < bx; x++) { xs[n] = x; ys[n++] = range->y1 - 1; } break; /* upper_left = (x0, y1-1) */
    case D_LEFT:  for (int y = range->y0 + off; y < by; y++) { xs[n] = range->x0; ys[n++] = y; } break;
    case D_RIGHT: for (int y = range->y0 + off; y < by; y++) { xs[n] = range->x1 - 1; ys[n++] = y; } break;
    case D_UP:    for (int x = range->x0 + off; x < bx; x++) { xs[n] = x; ys[n++] = range->y0; } break;
    }
    return n;
}
int orc_kat_edges(int x0, int y0, int x1, int y1, int direction, int inclusive, int *xs, int *ys) { /* for passages.rs:272-296 */
    rect_t r = {x0, y0, x1, y1};
    return edges(&r, direction, inclusive, xs, ys);
}
static int maze_has_cd(const room_t *rm, int x, int y) { int id = rect_index(&rm->range, x, y); return id >= 0 && rm->maze_passages[id]; }
/* select_start_or_end (passages.rs:143-179) */
static void select_start_or_end(orc_env *e, const room_t *rm, int direction, int *ox, int *oy) {
    int xs[256], ys[256];
    if (rm->kind == RK_NORMAL) {
        int n = edges(&rm->range, direction, 1, xs, ys);
        int k = (int)choose_index(e, n);
        *ox = xs[k]; *oy = ys[k];
    } else if (rm->kind == RK_MAZE) {
        rect_t range = rm->range;
        while (rect_valid(&range)) {
            int n = edges(&range, direction, 0, xs, ys), m = 0;
            for (int i = 0; i < n; i++) if (maze_has_cd(rm, xs[i], ys[i])) { xs[m] = xs[i]; ys[m++] = ys[i]; }
            if (m > 0) { int k = (int)choose_index(e, m); *ox = xs[k]; *oy = ys[k]; return; }
            switch (direction) {
            case D_DOWN: range.y1 -= 1; break;
            case D_LEFT: range.x0 -= 1; break;
            case D_RIGHT: range.x1 -= 1; break;
            case D_UP: range.y0 -= 1; break;
            }
        }
        fprintf(stderr, "oracle: cannot find maze floor\n"); abort();
    } else { *ox = rm->upx; *oy = rm->upy; }
}
static int door_kind(const room_t *rm) { return rm->kind == RK_NORMAL ? S_DOOR : S_PASSAGE; }
/* connect_2rooms (passages.rs:84-133) */
static void connect_2rooms(orc_env *e, const room_t *r1, const room_t *r2, int direction, plist_t *out) {
    if ((direction == D_UP || direction == D_LEFT)) { const room_t *t = r1; r1 = r2; r2 = t; direction = dir_reverse(direction); } /* (App. C-4) */
    int sx, sy, ex, ey;
    if (MUT(21)) { select_start_or_end(e, r2, dir_reverse(direction), &ex, &ey); select_start_or_end(e, r1, direction, &sx, &sy); } /* M21: end door drawn first */
    else { select_start_or_end(e, r1, direction, &sx, &sy); select_start_or_end(e, r2, dir_reverse(direction), &ex, &ey); }
    plist_push(out, sx, sy, door_kind(r1));
    plist_push(out, ex, ey, door_kind(r2));
    int tsx, tsy, tex, tey, tdir;
    if (direction == D_DOWN) {
        int y = (int)W32(22, &e->rng_d, sy + 1, ey + (MUT(23) ? 1 : 0)); /* M22 passages.rs:106 width, M23 inclusive upper bound */
        tdir = sx < ex ? D_RIGHT : D_LEFT;
        tsx = sx; tsy = y; tex = ex; tey = y;
    } else {
        int x = (int)W32(22, &e->rng_d, sx + 1, ex + (MUT(23) ? 1 : 0)); /* passages.rs:115 */
        tdir = sy < ey ? D_DOWN : D_UP;
        tsx = x; tsy = sy; tex = x; tey = ey;
    }
    int x = sx + DX[direction], y = sy + DY[direction]; /* start.direc_iter(..).skip(1) */
    while (!(x == tsx && y == tsy)) { plist_push(out, x, y, S_PASSAGE); x += DX[direction]; y += DY[direction]; }
    x = tsx; y = tsy;
    while (!(x == tex && y == tey)) { plist_push(out, x, y, S_PASSAGE); x += DX[tdir]; y += DY[tdir]; }
    x = tex; y = tey;
    while (!(x == ex && y == ey)) { plist_push(out, x, y, S_PASSAGE); x += DX[direction]; y += DY[direction]; }
}
/* RoomGraph / Node::candidates (passages.rs:222-270): neighbour room id -> direction or -1 */
static int graph_candidate(int rnx, int rny, int node, int other) {
    int x = node % rnx, y = node / rnx;
    for (int d = 0; d < 4; d++) {
        int nx = x + DX[d], ny = y + DY[d];
        if (nx < 0 || ny < 0 || nx >= rnx || ny >= rny) continue;
        if (nx + ny * rnx == other) return d;
    }
    return -1;
}
/* select_candidate (passages.rs:69-82); mode 0: !selected.contains, mode 1: !connections.contains */
static int select_candidate(orc_env *e, int num_rooms, int node, const uint8_t *excl, int *dir_out) {
    int rnx = e->cfg.room_num_x, rny = e->cfg.room_num_y, res = -1, i = 0;
    for (int id = 0; id < num_rooms; id++) {
        if (excl[id]) continue;
        int d = graph_candidate(rnx, rny, node, id);
        if (d < 0) continue;
        if (DH(16, &e->rng_d, (uint32_t)i + 1)) { res = id; *dir_out = d; } /* M16 passages.rs:79 width */
        i++;
    }
    return res;
}
static void dig_passages(orc_env *e, floor_t *fl, plist_t *out) {
    int n = fl->n_rooms;
    static __thread uint8_t conn[MAX_ROOMS][MAX_ROOMS];
    memset(conn, 0, sizeof conn);
    uint8_t selected[MAX_ROOMS] = {0};
    int cur = (int)W64(14, &e->rng_d, 0, (uint64_t)n), n_sel = 1; /* M14 passages.rs:30 width */
    selected[cur] = 1;
    while (n_sel < n) {
        int dir = 0;
        int nxt = select_candidate(e, n, cur, selected, &dir);
        if (nxt >= 0) {
            selected[nxt] = 1; n_sel++;
            conn[cur][nxt] = conn[nxt][cur] = 1;
            connect_2rooms(e, &fl->rooms[cur], &fl->rooms[nxt], dir, out);
            if (MUT(44)) cur = nxt; /* M44 App. C-4: the walk advances to the room it has just connected */
        } else {
            cur = set_select_m(selected, n, &e->rng_d, MUT(17)); /* M17 passages.rs:49 width */
        }
    }
    uint32_t try_num = W32(18, &e->rng_d, 0, e->cfg.max_extra_edges + (MUT(19) ? 1 : 0)); /* M18 passages.rs:54 width, M19 0..=max */
    for (uint32_t t = 0; t < try_num; t++) {
        int room1 = (int)W64(15, &e->rng_d, 0, (uint64_t)n), dir = 0; /* M15 passages.rs:56 width */
        int room2 = select_candidate(e, n, room1, conn[room1], &dir);
        if (room2 >= 0) {
            conn[room1][room2] = conn[room2][room1] = 1;
            connect_2rooms(e, &fl->rooms[room1], &fl->rooms[room2], dir, out);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* floor (rogue/floor.rs)                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* gen_attr (floor.rs:420-451) */
static uint8_t gen_attr(orc_env *e, int surface, int is_dark, uint32_t level) {
    const orc_config *c = &e->cfg; rng_t *r = &e->rng_d;
    switch (surface) {
    /* M11 floor.rs:430,437 width of the level draw, M12 :431,438 width of the rate draw, M13 the rate draw taken whatever the level draw said */
    case S_PASSAGE:
    case S_DOOR: {
        const uint32_t rate = surface == S_PASSAGE ? c->hidden_passage_rate_inv : c->locked_door_rate_inv;
        const uint8_t flag = surface == S_PASSAGE ? A_HIDDEN : A_LOCKED;
        const int deep = W32(11, r, 0, c->dark_level) < level;
        if (MUT(13)) { const int h = DH(12, r, rate); return deep && h ? flag : 0; }
        return deep && DH(12, r, rate) ? flag : 0;
    }
    case S_FLOOR:   return is_dark ? A_DARK : 0;
    default: return 0;
    }
}
static void floor_free(floor_t *fl) {
    for (int i = 0; i < fl->n_rooms; i++) room_free(&fl->rooms[i]);
    free(fl->surface); free(fl->attr); free(fl->doors); free(fl->gold); free(fl->rooms); free(fl->non_empty);
    memset(fl, 0, sizeof *fl);
}
/* Floor::gen_floor (floor.rs:50-104) */
static void gen_floor(orc_env *e, floor_t *fl, uint32_t level) {
    int n = e->W * e->H;
    memset(fl, 0, sizeof *fl);
    gen_rooms(e, fl, level);
    fl->surface = malloc(n); memset(fl->surface, S_NONE, n);
    fl->attr = calloc(n, 1); fl->doors = calloc(n, 1);
    fl->gold = malloc(n * sizeof(int32_t)); for (int i = 0; i < n; i++) fl->gold[i] = -1;
    for (int i = 0; i < fl->n_rooms; i++) { /* Room::draw (rooms.rs:58-82) */
        room_t *rm = &fl->rooms[i];
        if (rm->kind == RK_NORMAL) {
            for (int k = 0; k < rm->cap; k++) {
                int x, y; rect_nth(&rm->range, k, &x, &y);
                int s = rect_is_horiz_edge(&rm->range, y) ? S_WALLX : rect_is_vert_edge(&rm->range, x) ? S_WALLY : S_FLOOR;
                fl->surface[IDX(e, x, y)] = s;
                fl->attr[IDX(e, x, y)] = gen_attr(e, s, rm->is_dark, level);
            }
        } else if (rm->kind == RK_MAZE) {
            for (int k = 0; k < rm->cap; k++) if (rm->maze_passages[k]) { /* ascending index */
                int x, y; rect_nth(&rm->range, k, &x, &y);
                fl->surface[IDX(e, x, y)] = S_PASSAGE;
                fl->attr[IDX(e, x, y)] = gen_attr(e, S_PASSAGE, rm->is_dark, level);
            }
        }
    }
    plist_t pl = {0};
    dig_passages(e, fl, &pl);
    for (int i = 0; i < pl.n; i++) {
        int id = IDX(e, pl.v[i].x, pl.v[i].y), s = pl.v[i].kind;
        if (s == S_DOOR) fl->doors[id] = 1;
        fl->attr[id] = gen_attr(e, s, 0, level);
        if (MUT(45) || !(fl->attr[id] & (A_HIDDEN | A_LOCKED))) fl->surface[id] = s; /* M45 App. C-3: painted even when hidden / locked */
    }
    free(pl.v);
    for (int i = 0; i < fl->n_rooms; i++) fl->non_empty[i] = fl->rooms[i].kind != RK_EMPTY; /* Floor::new */
}
/* Floor::cd_to_room_id (floor.rs:194-200) */
static int cd_to_room_id(const floor_t *fl, int x, int y) {
    for (int i = 0; i < fl->n_rooms; i++) if (rect_contains(&fl->rooms[i].assigned, x, y)) return i;
    return -1;
}
/* Floor::set_obj / remove_obj (floor.rs:315-330) */
static int set_obj(floor_t *fl, int x, int y, int is_character) { int id = cd_to_room_id(fl, x, y); return id >= 0 && room_fill(&fl->rooms[id], x, y, is_character); }
static int remove_obj(floor_t *fl, int x, int y, int is_character) { int id = cd_to_room_id(fl, x, y); return id >= 0 && room_unfill(&fl->rooms[id], x, y, is_character); }
/* Floor::select_cell (floor.rs:333-346) */
static int floor_select_cell(orc_env *e, floor_t *fl, int is_character, int *x, int *y) {
    uint8_t cand[MAX_ROOMS];
    memcpy(cand, fl->non_empty, fl->n_rooms);
    while (set_len(cand, fl->n_rooms) > 0) {
        int idx = set_select_m(cand, fl->n_rooms, &e->rng_d, MUT(25)); /* M25 floor.rs:337-339 room id width */
        if (room_select_cell(&fl->rooms[idx], &e->rng_d, is_character, x, y)) return 1;
        cand[idx] = 0;
    }
    return 0;
}
/* Floor::setup_items + gold::Config::gen (floor.rs:132-153, item/gold.rs:18-24, item/mod.rs:407-410) */
static void setup_items(orc_env *e, floor_t *fl, uint32_t level) {
    const orc_config *c = &e->cfg;
    for (int i = 0; i < fl->n_rooms; i++) {
        room_t *rm = &fl->rooms[i]; int x, y;
        if (!room_select_cell(rm, &e->rng_d, 0, &x, &y)) continue;
        /* M26 gold.rs:19 width, M27 gold.rs:22 width, M28 the amount drawn before the 1-in-2 roll, M29 the amount drawn on the DUNGEON stream */
        uint32_t num = 0;
        if (MUT(28)) { num = W32(27, &e->rng_i, 0, c->gold_base + c->gold_per_level * level) + c->gold_minimum; if (!DH(26, &e->rng_i, c->gold_rate_inv)) continue; }
        else { if (!DH(26, &e->rng_i, c->gold_rate_inv)) continue; num = W32(27, MUT(29) ? &e->rng_d : &e->rng_i, 0, c->gold_base + c->gold_per_level * level) + c->gold_minimum; }
        room_fill(rm, x, y, 0);
        rm->has_gold = 1;
        fl->gold[IDX(e, x, y)] = (int32_t)num;
    }
}
/* Floor::setup_stair (floor.rs:156-167) */
static void setup_stair(orc_env *e, floor_t *fl) {
    int x, y;
    if (!floor_select_cell(e, fl, 0, &x, &y)) { fprintf(stderr, "oracle: no empty cell for stair\n"); abort(); }
    fl->surface[IDX(e, x, y)] = S_STAIR;
    set_obj(fl, x, y, 0);
}
/* EnemyHandler::select / exp_add / gen_enemy (enemies.rs:265-320) */
static int gen_enemy(orc_env *e, uint32_t min, uint32_t max, int64_t lev_add, int has_gold, mon_t *out) {
    const orc_config *c = &e->cfg; rng_t *r = &e->rng_e;
    if (!PC(30, r, has_gold ? c->appear_rate_gold : c->appear_rate_nogold)) return 0; /* M30 enemies.rs:297 width */
    size_t len = (size_t)e->n_stats;
    size_t idx = W32(31, r, min, max); /* M31 enemies.rs:266 width */
    if (idx > len) { size_t rg = len < 5 ? len : 5; idx = (size_t)range64(r, len - rg, len); }
    if (idx >= len) return 0; /* enemy_stats.get(idx)? */
    const mstat_t *st = &e->stats[idx];
    int64_t level = st->level + lev_add, hp = 0;
    for (int i = 0; i < 8; i++) hp += (int64_t)W64(32, r, 1, (uint64_t)level + (MUT(33) ? 0 : 1)); /* Dice::new(8, level).exec::<i64>   M32 character/mod.rs:218 width, M33 1..level instead of 1..=level */
    int64_t base = level == 1 ? hp / 8 : hp / 6;
    uint32_t exp_add = level >= 10 ? (uint32_t)base * 20u : (uint32_t)base * 4u;
    memset(out, 0, sizeof *out);
    out->type = (int)idx;
    out->level = level; out->hp = out->max_hp = hp;
    out->defense = st->defense - (int)lev_add;
    out->exp = st->exp + (uint32_t)(lev_add * 10) + exp_add;
    out->running = 0;
    return 1;
}
/* Floor::place_enemies (floor.rs:106-130) */
static void place_enemies(orc_env *e, floor_t *fl, uint32_t level, uint32_t lev_add) {
    if (e->n_stats == 0) return;
    uint32_t min = level >= 4 ? level - 4 : 0, max = level + 6;
    for (int i = 0; i < fl->n_rooms; i++) {
        room_t *rm = &fl->rooms[i]; int x, y; mon_t m;
        if (!room_select_cell(rm, &e->rng_d, 1, &x, &y)) continue;
        if (gen_enemy(e, min, max, (int64_t)lev_add, rm->has_gold, &m)) {
            m.x = x; m.y = y;
            int dup = -1; /* BTreeMap::insert replaces an existing key (enemies.rs:321-325) */
            for (int k = 0; k < e->n_placed; k++) if (e->placed[k].x == x && e->placed[k].y == y) dup = k;
            if (dup >= 0) e->placed[dup] = m; else e->placed[e->n_placed++] = m;
            room_fill(rm, x, y, 1);
        }
    }
}
/* Floor::can_move_impl (floor.rs:169-182) */
static int can_move_impl(const orc_env *e, int x, int y, int d, int is_enemy) {
    const floor_t *fl = &e->fl;
    int nx = x + DX[d], ny = y + DY[d];
    if (!INB(e, nx, ny)) return 0;
    int id = IDX(e, nx, ny);
    int res = can_walk(fl->surface[id]);
    if (!is_enemy) { res &= !(fl->attr[id] & A_HIDDEN); res &= !(fl->attr[id] & A_LOCKED); }
    if (dir_is_diag(d)) {
        if (!INB(e, x + DX[d], y) || !INB(e, x, y + DY[d])) return 0;
        res &= can_walk(fl->surface[IDX(e, x + DX[d], y)]);
        res &= can_walk(fl->surface[IDX(e, x, y + DY[d])]);
    }
    return res;
}
/* EnemyHandler::activate_area + activate (enemies.rs:342-362) */
static void mon_insert_active(orc_env *e, const mon_t *m) {
    for (int k = 0; k < e->n_active; k++) if (e->active[k].x == m->x && e->active[k].y == m->y) { e->active[k] = *m; return; }
    e->active[e->n_active++] = *m;
}
static int activate_at(orc_env *e, int x, int y) {
    for (int k = 0; k < e->n_placed; k++) if (e->placed[k].x == x && e->placed[k].y == y) {
        mon_t m = e->placed[k];
        e->placed[k] = e->placed[--e->n_placed];
        m.running = 1;
        mon_insert_active(e, &m);
        return 1;
    }
    return 0;
}
static void activate_area(orc_env *e, const rect_t *area) {
    int xs[MAX_MON], ys[MAX_MON], n = 0;
    for (int k = 0; k < e->n_placed; k++)
        if (rect_contains(area, e->placed[k].x, e->placed[k].y) && (e->stats[e->placed[k].type].attr & EA_MEAN)) { xs[n] = e->placed[k].x; ys[n++] = e->placed[k].y; }
    for (int i = 0; i < n; i++) activate_at(e, xs[i], ys[i]);
}
/* Floor::with_current_room / enters_room / leaves_room (floor.rs:201-261) */
static void enters_room(orc_env *e, int x, int y) {
    floor_t *fl = &e->fl;
    int id = cd_to_room_id(fl, x, y);
    if (id < 0) { fprintf(stderr, "oracle: no room for coord (%d,%d)\n", x, y); abort(); }
    room_t *rm = &fl->rooms[id];
    if (rm->is_visited) return;
    rm->is_visited = 1;
    if (!(rm->kind == RK_NORMAL && (MUT(67) || !rm->is_dark))) return; /* M67 App. C-7: dark rooms are lit up on entry too */
    for (int k = 0; k < rm->cap; k++) { int cx, cy; rect_nth(&rm->range, k, &cx, &cy); fl->attr[IDX(e, cx, cy)] |= A_DRAWN | A_VISIBLE; }
}
static void leaves_room(orc_env *e, int x, int y) {
    floor_t *fl = &e->fl;
    int id = cd_to_room_id(fl, x, y);
    if (id < 0) { fprintf(stderr, "oracle: no room for coord (%d,%d)\n", x, y); abort(); }
    room_t *rm = &fl->rooms[id];
    if (!(rm->is_visited && rm->is_dark)) return;
    rect_t range = rm->kind == RK_EMPTY ? rm->assigned : rm->range;
    for (int cy = range.y0; cy < range.y1; cy++) for (int cx = range.x0; cx < range.x1; cx++)
        if (!(rect_is_horiz_edge(&range, cy) || rect_is_vert_edge(&range, cx))) fl->attr[IDX(e, cx, cy)] &= ~A_VISIBLE;
}
/* Floor::player_in (floor.rs:264-295) */
static void player_in(orc_env *e, int x, int y, int init) {
    floor_t *fl = &e->fl;
    if (init || fl->doors[IDX(e, x, y)]) {
        enters_room(e, x, y);
        int id = cd_to_room_id(fl, x, y);
        if (id >= 0) activate_area(e, &fl->rooms[id].assigned);
    }
    fl->attr[IDX(e, x, y)] |= A_VISITED;
    set_obj(fl, x, y, 1);
    for (int d = 0; d < 9; d++) {
        int cx = x + DX[d], cy = y + DY[d];
        if (!INB(e, cx, cy)) continue;
        int id = IDX(e, cx, cy);
        if (MUT(58) || !dir_is_diag(d) || fl->surface[id] != S_PASSAGE) { /* Cell::approached (field.rs:20-26)   M58 App. C-7: diagonal passages revealed too */
            if (fl->attr[id] & A_HIDDEN) continue;
            fl->attr[id] |= A_DRAWN | A_VISIBLE;
        }
    }
}
/* Floor::player_out (floor.rs:298-312) */
static void player_out(orc_env *e, int x, int y) {
    floor_t *fl = &e->fl;
    if (fl->doors[IDX(e, x, y)]) leaves_room(e, x, y);
    remove_obj(fl, x, y, 1);
    for (int d = 0; d < 9; d++) {
        int cx = x + DX[d], cy = y + DY[d];
        if (!INB(e, cx, cy)) continue;
        int id = IDX(e, cx, cy);
        if (!MUT(66) && fl->surface[id] == S_FLOOR && (fl->attr[id] & A_DARK)) fl->attr[id] &= ~A_VISIBLE; /* Cell::left   M66 App. C-7: dark floor stays visible behind the player */
    }
}
/* Floor::in_same_room (floor.rs:381-393) */
static int in_same_room(const orc_env *e, int ax, int ay, int bx, int by) {
    const floor_t *fl = &e->fl;
    int id = cd_to_room_id(fl, ax, ay);
    if (id < 0) return 0;
    if (cd_to_room_id(fl, bx, by) != id) return 0;
    const room_t *rm = &fl->rooms[id];
    if (rm->kind == RK_EMPTY) return 1;
    return rect_contains(&rm->range, ax, ay) == rect_contains(&rm->range, bx, by);
}
/* Floor::make_dist_map (floor.rs:395-416): FIFO BFS over 8 directions */
static uint32_t *make_dist_map(const orc_env *e, int fx, int fy) {
    int n = e->W * e->H;
    uint32_t *dist = malloc(n * sizeof(uint32_t));
    int *queue = malloc(n * sizeof(int));
    for (int i = 0; i < n; i++) dist[i] = DIST_INF;
    int qh = 0, qt = 0;
    dist[IDX(e, fx, fy)] = 0; queue[qt++] = IDX(e, fx, fy);
    while (qh < qt) {
        int cur = queue[qh++], cx = cur % e->W, cy = cur / e->W;
        for (int d = 0; d < 8; d++) {
            int nx = cx + DX[d], ny = cy + DY[d];
            if (!INB(e, nx, ny)) continue;
            int nid = IDX(e, nx, ny);
            if (dist[nid] != DIST_INF || !can_move_impl(e, cx, cy, d, 1)) continue;
            queue[qt++] = nid;
            dist[nid] = dist[cur] + 1;
        }
    }
    free(queue);
    return dist;
}
/* DistCache::make_dist_map (rogue/mod.rs:492-518): FIFO keyed by target coord only, never
 * invalidated, holds up to 9 maps */
static const uint32_t *dist_cache_get(orc_env *e, int x, int y) {
    for (int i = 0; i < e->n_dcache; i++) if (e->dcache[i].kx == x && e->dcache[i].ky == y) return e->dcache[i].map;
    uint32_t *m = make_dist_map(e, x, y);
    int len = e->n_dcache;
    e->dcache[e->n_dcache++] = (dcache_t){m, x, y};
    if (len > 8) { /* MAX_CACHED_DIST = 8: pop_front */
        free(e->dcache[0].map);
        memmove(&e->dcache[0], &e->dcache[1], (e->n_dcache - 1) * sizeof(dcache_t));
        e->n_dcache--;
    }
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* dungeon (rogue/mod.rs)                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* Dungeon::new_level_ (rogue/mod.rs:434-481) */
static void new_level_(orc_env *e, int is_initial) {
    uint32_t level = ++e->level;
    floor_t nf;
    gen_floor(e, &nf, level);
    setup_items(e, &nf, level); /* set_gold is always true: GameInfo.is_cleared is never set */
    setup_stair(e, &nf);
    if (!is_initial) { e->n_placed = 0; e->n_active = 0; } /* remove_enemies */
    uint32_t lev_add = e->cfg.amulet_level < level ? level - e->cfg.amulet_level : 0;
    place_enemies(e, &nf, level, lev_add);
    if (!e->cfg.hide_dungeon)
        for (int y = 1; y < e->H - 1; y++) for (int x = 0; x < e->W; x++) nf.attr[IDX(e, x, y)] |= A_VISIBLE;
    if (!is_initial) { /* past_floors.push(old floor): only its history map is ever read */
        if (e->n_past == e->cap_past) { e->cap_past = e->cap_past ? e->cap_past * 2 : 8; e->past_visited = realloc(e->past_visited, e->cap_past * sizeof(uint8_t *)); }
        int n = e->W * e->H;
        uint8_t *v = malloc(n);
        for (int i = 0; i < n; i++) v[i] = (e->fl.attr[i] & A_VISITED) != 0;
        e->past_visited[e->n_past++] = v;
        floor_free(&e->fl);
    }
    e->fl = nf;
    if (MUT(47)) { for (int i = 0; i < e->n_dcache; i++) free(e->dcache[i].map); e->n_dcache = 0; } /* M47 App. C-11: the DistCache dropped with the level */
}
/* actions::new_level (actions.rs:121-138) */
static void actions_new_level(orc_env *e, int is_init) {
    if (!is_init) new_level_(e, 0);
    int x, y;
    if (!floor_select_cell(e, &e->fl, 1, &x, &y)) { fprintf(stderr, "oracle: no space for player\n"); abort(); }
    e->px = x; e->py = y;
    player_in(e, x, y, 1);
}
static const mon_t *mon_at(const orc_env *e, int x, int y, int *is_active) {
    for (int k = 0; k < e->n_placed; k++) if (e->placed[k].x == x && e->placed[k].y == y) { if (is_active) *is_active = 0; return &e->placed[k]; }
    for (int k = 0; k < e->n_active; k++) if (e->active[k].x == x && e->active[k].y == y) { if (is_active) *is_active = 1; return &e->active[k]; }
    return NULL;
}
/* Dungeon::move_enemy (rogue/mod.rs:339-375). skip = occupied by an already-moved active or any
 * asleep monster (enemies.rs:386-387) */
typedef int (*skip_fn)(const orc_env *, int, int);
static int skip_occupied(const orc_env *e, int x, int y) { return mon_at(e, x, y, NULL) != NULL; }
static int skip_never(const orc_env *e, int x, int y) { (void)e; (void)x; (void)y; return 0; }
enum { MR_CANTMOVE, MR_CANMOVE, MR_REACH };
static int move_enemy(orc_env *e, int cx, int cy, int tx, int ty, skip_fn skip, int *ox, int *oy) {
    const uint32_t *dist = dist_cache_get(e, tx, ty);
    uint32_t best = DIST_INF; int found = 0;
    for (int d = 0; d < 9; d++) {
        int nx = cx + DX[d], ny = cy + DY[d];
        if (skip(e, nx, ny)) continue;
        /* A neighbour outside the grid (a chaser on column 0 / W-1: the single cell of an Empty room, rooms.rs:226, may lie there): the reference's
         * `*dist_map.get_p(next)` (rect-iter Get2D::get_p = try_get_p(..).expect(..)) PANICS on it -- the worker thread dies -- so there is no result to
         * reproduce.  Here (and in the HIP stepper's monsters_move) such a neighbour is simply not a candidate.  (Found by the ASan build, round 6: the
         * unguarded read landed on the allocator's header word, 0, which the tests below happened to treat the same way.) */
        if (!INB(e, nx, ny)) continue;
        uint32_t nd = dist[IDX(e, nx, ny)];
        if (nd == 0 && can_move_impl(e, cx, cy, d, 1)) return MR_REACH;
        if (MUT(48) && !can_move_impl(e, cx, cy, d, 1)) continue; /* M48 App. C-10: candidates checked for legality (no corner cutting) */
        if (nd != DIST_INF && nd > 0 && (!found || (MUT(49) ? nd <= best : nd < best))) { best = nd; *ox = nx; *oy = ny; found = 1; } /* stable sort, first minimum; M49 last minimum */
    }
    return found ? MR_CANMOVE : MR_CANTMOVE;
}
/* Dungeon::move_enemy_randomly (rogue/mod.rs:376-397) */
static int move_enemy_randomly(orc_env *e, int cx, int cy, int px, int py, skip_fn skip, int *ox, int *oy) {
    int d = (int)W64(37, MUT(38) ? &e->rng_e : &e->rng_d, 0, 8); /* M37 rogue/mod.rs:383 width, M38 drawn on the ENEMY stream */
    int nx = cx + DX[d], ny = cy + DY[d];
    if (skip(e, nx, ny) || !can_move_impl(e, cx, cy, d, 1)) return MR_CANTMOVE;
    if (nx == px && ny == py) return MR_REACH;
    *ox = nx; *oy = ny;
    return MR_CANMOVE;
}

/* ------------------------------------------------------------------------------------------ */
/* player (character/player.rs) and fight (character/fight.rs)                                */
/* ------------------------------------------------------------------------------------------ */
static const uint32_t LEVEL_EXPS[21] = {10, 20, 40, 80, 160, 320, 640, 1300, 2600, 5200, 13000, 26000, 50000, 100000,
                                        200000, 400000, 800000, 2000000, 4000000, 8000000, 0xFFFFFFFFu};
static const int64_t HIT_PLUS[32] = {-7, -6, -5, -4, -3, -2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3};
static const int64_t DAM_PLUS[32] = {-7, -6, -5, -4, -3, -2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 6};
static int64_t hit_prob_plus(int64_t st) { return (st <= 0 || st > 32) ? 0 : HIT_PLUS[st - 1]; }
static int64_t damage_plus(int64_t st) { return (st <= 0 || st > 32) ? 0 : DAM_PLUS[st - 1]; }
static uint32_t attack_rate(int64_t level, int armor, int64_t revision) { /* fight.rs:84-87 + Parcent::truncate */
    int64_t v = (level + armor + revision + 1) * 5;
    return (uint32_t)(v < 0 ? 0 : v > 100 ? 100 : v);
}
#define PLAYER_STR 16    /* StatusInner::from_config: Strength(16) (player.rs:288) */
#define ENEMY_STR 10     /* Enemy::STRENGTH (enemies.rs:161) */
/* Player::arm (player.rs:125-132): def + def_plus of the equipped armor (Armor::def, armor.rs:100-102), Defense(0) without one */
static int player_arm(const orc_env *e) { return e->armor < 0 ? 0 : e->pack[e->armor].def + e->pack[e->armor].def_plus; }

/* Player::heal (player.rs:221-240) */
static int player_heal(orc_env *e) {
    e->quiet += 1;
    int64_t quiet = e->quiet, level = e->plevel, heal;
    if (level < (MUT(42) ? 9 : 8)) { heal = quiet + (level << 1) - (MUT(65) ? 19 : 20); /* M65 player.rs:224: heals one turn earlier */ /* M42 player.rs:225: the random heal from level 9 on */ heal = heal < 0 ? 0 : heal > 1 ? 1 : heal; }
    else if (quiet >= 3) heal = (int64_t)W64(41, &e->rng_e, 1, (uint64_t)(level - 6)); /* M41 player.rs:228 width */
    else heal = 0;
    if (heal > 0) {
        e->hp += heal; if (e->hp > e->hp_max) e->hp = e->hp_max;
        e->quiet = 0;
        return 1;
    }
    return 0;
}
/* Player::level_up (player.rs:185-197) + Leveling::check_level (player.rs:345-352) */
static int player_level_up(orc_env *e, uint32_t exp) {
    e->exp += exp;
    size_t cur = (size_t)(e->plevel - 1), diff = 0;
    if (cur < 21) { while (!(e->exp < LEVEL_EXPS[cur + diff])) diff++; }
    if (diff > 0) {
        e->plevel += (int64_t)diff;
        int64_t add = 0;
        for (size_t i = 0; i < diff; i++) add += (int64_t)W64(40, &e->rng_e, 1, 11); /* M40 player.rs:193 width */
        e->hp_max += add; e->hp += add;
        return 1;
    }
    return 0;
}
/* actions::move_active_enemies + EnemyHandler::move_actives + fight::enemy_attack
 * (actions.rs:82-119, enemies.rs:366-424, fight.rs:41-72) */
static int mon_cmp(const void *a, const void *b) { const mon_t *p = a, *q = b; return p->x != q->x ? p->x - q->x : p->y - q->y; }
static int move_active_enemies(orc_env *e, rlist_t *res) {
    mon_t tmp[MAX_MON], attackers[MAX_MON]; int n = e->n_active, n_att = 0;
    memcpy(tmp, e->active, n * sizeof(mon_t));
    qsort(tmp, n, sizeof(mon_t), mon_cmp); /* BTreeMap key order [level, x, y] */
    e->n_active = 0;
    for (int i = 0; i < n; i++) {
        mon_t m = tmp[i];
        int nx = m.x, ny = m.y, ox = 0, oy = 0, r;
        int attr = e->stats[m.type].attr;
        /* (rng.does_happen(2) && is_random) || (!rng.does_happen(5) && is_confused) */
        int random_move = 0;
        /* M34 enemies.rs:401 width, M35 :402 width, M36 both draws always taken (no short circuit) */
        if (MUT(36)) { int a = DH(34, &e->rng_e, 2), b = DH(35, &e->rng_e, 5); random_move = (a && (attr & EA_RANDOM)) || (!b && (attr & EA_CONFUSED)); }
        else if (DH(34, &e->rng_e, 2) && (attr & EA_RANDOM)) random_move = 1;
        else if (!DH(35, &e->rng_e, 5) && (attr & EA_CONFUSED)) random_move = 1;
        if (random_move) r = move_enemy_randomly(e, m.x, m.y, e->px, e->py, skip_occupied, &ox, &oy);
        else r = move_enemy(e, m.x, m.y, e->px, e->py, skip_occupied, &ox, &oy);
        if (r == MR_REACH) attackers[n_att++] = m;
        else if (r == MR_CANMOVE) { nx = ox; ny = oy; }
        m.x = nx; m.y = ny;
        if (MUT(46) && mon_at(e, m.x, m.y, NULL)) { /* M46 App. C-10: an occupied cell keeps its earlier occupant, the newcomer is dropped */ } else
        mon_insert_active(e, &m); /* BTreeMap::insert: same key overwrites */
    }
    if (n_att > 0) e->quiet = 0; /* player.buttle() */
    int did_hit = 0;
    for (int i = 0; i < n_att; i++) {
        const mon_t *m = &attackers[i]; const mstat_t *st = &e->stats[m->type];
        uint32_t rate = attack_rate(m->level, player_arm(e), hit_prob_plus(ENEMY_STR));
        int64_t dam_plus = damage_plus(ENEMY_STR) + damage_plus(PLAYER_STR), sum = 0; int hit = 0;
        for (int k = 0; k < st->n_attack; k++) {
            if (!PC(39, &e->rng_e, rate)) continue; /* M39 fight.rs:61 width */
            hit = 1;
            int64_t dmg = 0;
            for (int t = 0; t < st->att_times[k]; t++) dmg += (int64_t)W64(43, &e->rng_e, 1, (uint64_t)st->att_max[k] + 1); /* M43 character/mod.rs:232 width */
            sum += dmg + dam_plus;
        }
        if (hit) {
            rpush(res, RE_NOTIFY, MSG_HIT_FROM);
            did_hit = 1;
            e->hp = e->hp - sum > 0 ? e->hp - sum : 0; /* Player::get_damage (player.rs:177-184) */
            if (e->hp == 0) { rpush(res, RE_GRAVE, 0); return 1; }
        } else rpush(res, RE_NOTIFY, MSG_MISS_FROM);
    }
    if (did_hit) rpush(res, RE_STATUS, 0);
    return 0;
}
/* actions::after_turn + Player::turn_passed (actions.rs:67-80, player.rs:163-176) */
static int after_turn(orc_env *e, rlist_t *res) {
    e->food_left -= 1; /* u32, wraps in release builds */
    if (MUT(68) && e->food_left == 0) { e->dead = 1; rpush(res, RE_GRAVE, 0); return 1; } /* M68 App. C-9: starvation kills (the reference ignores PlayerEvent::Dead here) */
    if (e->food_left != 0) {
        uint32_t hunger = e->cfg.hunger_time / 10;
        if (e->food_left == hunger || e->food_left == hunger * 2) rpush(res, RE_STATUS, 0);
        if (player_heal(e)) rpush(res, RE_STATUS, 0);
    } /* else: [PlayerEvent::Dead], ignored (actions.rs:75) */
    return move_active_enemies(e, res);
}
/* actions::player_attack + fight::player_attack (actions.rs:140-166, fight.rs:6-39) */
static void player_attack(orc_env *e, int x, int y, rlist_t *res) {
    e->quiet = 0;
    activate_at(e, x, y);
    int is_active = 0;
    mon_t *m = (mon_t *)mon_at(e, x, y, &is_active);
    /* no thrown weapon: hit_plus / dam_plus / at_weild of Player::weapon, else 0 / 0 / 1d4 (fight.rs:20-33) */
    const orc_item *wp = e->weapon < 0 ? NULL : &e->pack[e->weapon];
    int64_t str_p = hit_prob_plus(PLAYER_STR) + ((m->running && !MUT(51)) ? 0 : 4) + (wp ? wp->hit_plus : 0); /* M51 App. C-13: the +4 against a monster that was asleep when attacked */
    uint32_t rate = attack_rate(e->plevel, m->defense, str_p);
    if (PC(39, &e->rng_e, rate)) { /* roll over the one die (fight.rs:52-72) */
        uint64_t times = wp ? wp->wield_times : 1; int64_t mx = wp ? wp->wield_max : 4;
        int64_t dmg = 0;
        for (uint64_t t = 0; t < times; t++) dmg += (int64_t)W64(43, &e->rng_e, 1, (uint64_t)mx + 1); /* Damage::random (character/mod.rs:229-234) */
        dmg += (wp ? wp->dam_plus : 0) + damage_plus(PLAYER_STR);
        rpush(res, RE_NOTIFY, MSG_HIT_TO);
        if (m->hp <= dmg) { /* Enemy::get_damage (enemies.rs:205-213) */
            uint32_t exp = m->exp;
            if (is_active) { *m = e->active[--e->n_active]; } else { *m = e->placed[--e->n_placed]; }
            if (player_level_up(e, exp)) rpush(res, RE_STATUS, 0);
            rpush(res, RE_NOTIFY, MSG_KILLED);
            rpush(res, RE_REDRAW, 0);
        } else m->hp = MUT(50) ? m->hp - dmg : dmg - m->hp; /* quirk: stores damage - cur   M50 App. C-13: the sane subtraction */
    } else rpush(res, RE_NOTIFY, MSG_MISS_TO);
}
/* actions::move_player + get_item (actions.rs:168-231). returns `done` */
static int move_player(orc_env *e, int d, rlist_t *res) {
    if (!can_move_impl(e, e->px, e->py, d, 0)) { rpush(res, RE_NOTIFY, 0 /* CantMove: no flag */); return 1; }
    int nx = e->px + DX[d], ny = e->py + DY[d];
    if (mon_at(e, nx, ny, NULL)) { player_attack(e, nx, ny, res); return 1; }
    player_out(e, e->px, e->py);
    player_in(e, nx, ny, 0);
    e->px = nx; e->py = ny;
    rpush(res, RE_REDRAW, 0);
    int id = IDX(e, nx, ny);
    if (e->fl.gold[id] >= 0) { /* actions::get_item (actions.rs:206-231) -> ItemBox::entry (itembox.rs:30-40) */
        /* dungeon gold is `many` (item/mod.rs:409): merge into the first pack item of the same kind (check_merge, itembox.rs:52-58) ... */
        int slot = -1;
        for (int k = 0; k < e->n_pack && slot < 0; k++) if (e->pack[k].kind == ORC_KIND_GOLD) slot = k;
        if (slot >= 0) { e->pack[slot].how_many += (uint32_t)e->fl.gold[id]; e->pack[slot].attr |= ORC_ATTR_IS_MANY; } /* MergeEntry::exec (itembox.rs:74-79) */
        else if ((uint64_t)e->n_pack < e->cfg.max_items && e->n_pack < ORC_MAX_INIT_ITEMS + 1) { /* ... else the lowest free slot (InsertEntry) ... */
            orc_item *g = &e->pack[e->n_pack++];
            memset(g, 0, sizeof *g); g->kind = ORC_KIND_GOLD; g->how_many = (uint32_t)e->fl.gold[id]; g->attr = ORC_ATTR_IS_MANY;
        } else return 0; /* ... else `entry` is None: get_item returns Ok(None), nothing is picked up or removed */
        remove_obj(&e->fl, nx, ny, 0);
        e->fl.gold[id] = -1;
        rpush(res, RE_NOTIFY, 0 /* GotItem */);
        rpush(res, RE_STATUS, 0);
        return 1;
    }
    return 0;
}
/* Floor::search + actions::search (floor.rs:349-370, actions.rs:197-204) */
static void do_search(orc_env *e, rlist_t *res) {
    floor_t *fl = &e->fl; const orc_config *c = &e->cfg;
    for (int d = 0; d < 8; d++) {
        int x = e->px + DX[d], y = e->py + DY[d];
        if (!INB(e, x, y)) continue;
        int id = IDX(e, x, y);
        if ((fl->attr[id] & A_HIDDEN) && DH(56, &e->rng_d, c->passage_unlock_rate_inv)) { /* M56 floor.rs:359 width */
            fl->attr[id] &= ~(A_LOCKED | A_HIDDEN); fl->attr[id] |= A_VISIBLE; fl->surface[id] = S_PASSAGE;
        }
        if ((fl->attr[id] & A_LOCKED) && DH(57, &e->rng_d, c->door_unlock_rate_inv)) { /* M57 floor.rs:363 width */
            fl->attr[id] &= ~(A_LOCKED | A_HIDDEN); fl->attr[id] |= A_VISIBLE; fl->surface[id] = S_DOOR;
            rpush(res, RE_NOTIFY, MSG_SECRET_DOOR);
        }
    }
    rpush(res, RE_REDRAW, 0);
}
enum { ACT_MOVE, ACT_MOVE_UNTIL, ACT_DOWNSTAIR, ACT_SEARCH, ACT_NOOP };
/* KeyMap::ai (input.rs:73-100) */
static int key_to_action(uint8_t key, int *dir) {
    switch (key) {
    case 'l': *dir = D_RIGHT; return ACT_MOVE; case 'k': *dir = D_UP; return ACT_MOVE;
    case 'j': *dir = D_DOWN; return ACT_MOVE; case 'h': *dir = D_LEFT; return ACT_MOVE;
    case 'u': *dir = D_RIGHTUP; return ACT_MOVE; case 'y': *dir = D_LEFTUP; return ACT_MOVE;
    case 'n': *dir = D_RIGHTDOWN; return ACT_MOVE; case 'b': *dir = D_LEFTDOWN; return ACT_MOVE;
    case 'L': *dir = D_RIGHT; return ACT_MOVE_UNTIL; case 'K': *dir = D_UP; return ACT_MOVE_UNTIL;
    case 'J': *dir = D_DOWN; return ACT_MOVE_UNTIL; case 'H': *dir = D_LEFT; return ACT_MOVE_UNTIL;
    case 'U': *dir = D_RIGHTUP; return ACT_MOVE_UNTIL; case 'Y': *dir = D_LEFTUP; return ACT_MOVE_UNTIL;
    case 'N': *dir = D_RIGHTDOWN; return ACT_MOVE_UNTIL; case 'B': *dir = D_LEFTDOWN; return ACT_MOVE_UNTIL;
    case '.': return ACT_NOOP; case 's': return ACT_SEARCH; case '>': return ACT_DOWNSTAIR;
    default: return -1;
    }
}
/* actions::process_action (actions.rs:16-65). returns 1 if the player died */
static int process_action(orc_env *e, int act, int dir, rlist_t *out) {
    int ui = 0;
    switch (act) {
    case ACT_DOWNSTAIR:
        if (e->fl.surface[IDX(e, e->px, e->py)] == S_STAIR) {
            actions_new_level(e, 0);
            rpush(out, RE_REDRAW, 0); rpush(out, RE_STATUS, 0);
        } else rpush(out, RE_NOTIFY, MSG_NO_DOWNSTAIR);
        ui = after_turn(e, out);
        break;
    case ACT_MOVE:
        move_player(e, dir, out);
        ui = after_turn(e, out);
        break;
    case ACT_MOVE_UNTIL:
        for (;;) {
            rlist_t *res = &tls_scratch_b; res->n = 0;
            int done = move_player(e, dir, res);
            int id = IDX(e, e->px, e->py);
            uint8_t tile = (e->fl.attr[id] & A_VISIBLE) ? SURFACE_GLYPH[e->fl.surface[id]] : ' '; /* Cell::tile (field.rs:91-98) */
            if (done || (tile != '.' && tile != '#')) { for (int i = 0; i < res->n; i++) rpush(out, res->v[i].kind, res->v[i].msg); if (MUT(62)) ui = after_turn(e, out); /* M62 App. C-8: the stopping iteration costs a turn too */ break; }
            else if (out->n == 0) { for (int i = 0; i < res->n; i++) rpush(out, res->v[i].kind, res->v[i].msg); }
            ui = after_turn(e, out);
        }
        break;
    case ACT_SEARCH:
        do_search(e, out);
        if (!MUT(64)) ui = after_turn(e, out); /* M64 App. C-8: Search costs no turn */
        break;
    case ACT_NOOP: if (MUT(63)) { ui = after_turn(e, out); break; } /* M63 App. C-8: NoOp costs a turn */ return 0;
    }
    return ui;
}

/* ------------------------------------------------------------------------------------------ */
/* RunTime::draw_screen / player_status and the PlayerState mirror                            */
/* ------------------------------------------------------------------------------------------ */
/* core/src/lib.rs:264-285 + rogue/mod.rs:278-300,398-404 */
static void draw_screen(const orc_env *e, uint8_t *map) {
    const floor_t *fl = &e->fl;
    for (int y = 1; y < e->H - 1; y++) for (int x = 0; x < e->W; x++) {
        int id = IDX(e, x, y);
        map[id] = (fl->attr[id] & A_VISIBLE) ? SURFACE_GLYPH[fl->surface[id]] : ' ';
    }
    for (int y = 1; y < e->H - 1; y++) for (int x = 0; x < e->W; x++) {
        int id = IDX(e, x, y);
        if (!(fl->attr[id] & (A_VISIBLE | A_DRAWN))) continue;
        if (x == e->px && y == e->py) { map[id] = '@'; continue; }
        if (fl->gold[id] >= 0) { map[id] = '*'; continue; }
        const mon_t *m = mon_at(e, x, y, NULL);
        if (m) {
            int dx = e->px - x, dy = e->py - y;
            if (dx * dx + dy * dy <= 2 || in_same_room(e, e->px, e->py, x, y)) map[id] = e->stats[m->type].tile;
        }
    }
}
/* RunTime::player_status + Player::fill_status (core/src/lib.rs:345-356, player.rs:107-118) */
static void player_status(const orc_env *e, uint32_t st[10]) {
    uint32_t hunger = e->cfg.hunger_time / 10;
    uint32_t gold = 0; /* the first Gold token of the pack, or 0 (core/src/lib.rs:348-353) */
    for (int k = 0; k < e->n_pack; k++) if (e->pack[k].kind == ORC_KIND_GOLD) { gold = e->pack[k].how_many; break; }
    st[0] = e->level; st[1] = gold; st[2] = (uint32_t)e->hp; st[3] = (uint32_t)e->hp_max;
    st[4] = PLAYER_STR; st[5] = PLAYER_STR; st[6] = 0 /* defense never filled */; st[7] = (uint32_t)e->plevel; st[8] = e->exp;
    st[9] = e->food_left <= hunger ? 2 : e->food_left <= hunger * 2 ? 1 : 0;
}
/* PlayerState::draw_map (python/src/lib.rs:59-68): history of the level in the MIRROR status */
static void mirror_draw_map(orc_env *e) {
    int n = e->W * e->H;
    uint32_t lvl = e->status[0];
    if (lvl == e->level) for (int i = 0; i < n; i++) e->hist[i] = (e->fl.attr[i] & A_VISITED) != 0;
    else if (lvl >= 1 && (int)(lvl - 1) < e->n_past) memcpy(e->hist, e->past_visited[lvl - 1], n);
    else { fprintf(stderr, "oracle: history unwrap on None\n"); abort(); }
    draw_screen(e, e->screen);
}

/* ------------------------------------------------------------------------------------------ */
/* build / reset / react                                                                      */
/* ------------------------------------------------------------------------------------------ */
static void runtime_free(orc_env *e) {
    floor_free(&e->fl);
    for (int i = 0; i < e->n_past; i++) free(e->past_visited[i]);
    e->n_past = 0;
    for (int i = 0; i < e->n_dcache; i++) free(e->dcache[i].map);
    e->n_dcache = 0;
    e->n_placed = e->n_active = 0;
}
/* GameConfig::build (core/src/lib.rs:193-228) */
/* ItemHandler::init_player_items (item/mod.rs:411-422) over InitItem::initialize (item/mod.rs:181-221), then the two equip_from_box calls of
 * Player::init_items (player.rs:140-152).  Returns 1 where the reference returns an InvalidSetting error. */
static int player_init_items(orc_env *e) {
    const orc_config *c = &e->cfg;
    e->n_pack = 0; e->weapon = e->armor = -1;
    for (int i = 0; i < c->n_init_items; i++) {
        const orc_init_item *it = &c->init_items[i];
        orc_item item;
        memset(&item, 0, sizeof item);
        if (it->tag == ORC_INIT_NOINIT) item = it->item;
        else if (it->tag == ORC_INIT_WEAPON) { /* Handler::gen_item_by: the first status of that name (handler.rs:54-62) */
            const orc_weapon_stat *st = NULL;
            for (int k = 0; k < c->n_weapons && !st; k++) if (!strcmp(c->weapons[k].name, it->name)) st = &c->weapons[k];
            if (!st) return 1; /* "Specified item {} is not registerd to WeaponHandler" */
            uint32_t num = W32(59, &e->rng_i, st->init_lo, st->init_hi); /* M59 weapon.rs:159 width   WeaponStatus::build (weapon.rs:148-170): the one draw, on the item stream */
            item.kind = ORC_KIND_WEAPON; strcpy(item.name, st->name); item.wield_times = st->wield_times; item.wield_max = st->wield_max;
            item.hit_plus = 0 + it->hit_plus; item.dam_plus = 0 + it->dam_plus; item.attr = st->attr; item.how_many = num + it->num_plus;
        } else { /* ArmorStatus::build draws nothing (armor.rs:141-169) */
            const orc_armor_stat *st = NULL;
            for (int k = 0; k < c->n_armors && !st; k++) if (!strcmp(c->armors[k].name, it->name)) st = &c->armors[k];
            if (!st) return 1;
            item.kind = ORC_KIND_ARMOR; strcpy(item.name, st->name); item.def = st->def; item.def_plus = 0 + it->def_plus; item.how_many = 1;
        }
        if ((uint64_t)e->n_pack >= c->max_items) return 1; /* ItemBox::add finds no empty char: "[init_player_items] Failed to add item" */
        e->pack[e->n_pack++] = item;
    }
    /* get_initial_weapon / get_initial_armor: the name of the FIRST InitItem of that variant (player.rs:198-213); equip_from_box: the first
     * pack item of that kind and name (player.rs:214-220, ItemBox::find_by iterates in slot order) */
    for (int i = 0; i < c->n_init_items; i++) if (c->init_items[i].tag == ORC_INIT_WEAPON) {
        for (int k = 0; k < e->n_pack && e->weapon < 0; k++)
            if (e->pack[k].kind == ORC_KIND_WEAPON && !strcmp(e->pack[k].name, c->init_items[i].name)) { e->weapon = k; e->pack[k].attr |= ORC_ATTR_EQUIPPED; }
        break;
    }
    for (int i = 0; i < c->n_init_items; i++) if (c->init_items[i].tag == ORC_INIT_ARMOR) {
        for (int k = 0; k < e->n_pack && e->armor < 0; k++)
            if (e->pack[k].kind == ORC_KIND_ARMOR && !strcmp(e->pack[k].name, c->init_items[i].name)) { e->armor = k; e->pack[k].attr |= ORC_ATTR_EQUIPPED; }
        break;
    }
    return 0;
}
static int runtime_build(orc_env *e) {
    const orc_config *c = &e->cfg;
    runtime_free(e);
    rng_seed(&e->rng_i, c->seed_lo, c->seed_hi);  /* ItemHandler::new (item/mod.rs:390) */
    rng_seed(&e->rng_e, c->seed_lo, c->seed_hi);  /* enemies::Config::build (enemies.rs:34) */
    rng_seed(&e->rng_d, c->seed_lo, c->seed_hi);  /* rogue::Dungeon::new (rogue/mod.rs:417) */
    /* EnemyHandler::new: stable sort by rarity (enemies.rs:250-261) */
    e->n_stats = c->n_enemies;
    for (int i = 0; i < e->n_stats; i++) {
        if (c->enemy_builtin[i] >= 0) e->stats[i] = BUILTIN[c->enemy_builtin[i]];
        else {
            const orc_monstat *q = &c->enemy_custom[i]; mstat_t *t = &e->stats[i];
            memset(t, 0, sizeof *t);
            t->n_attack = q->n_attack;
            for (int k = 0; k < 4; k++) { t->att_times[k] = q->att_times[k]; t->att_max[k] = q->att_max[k]; }
            t->attr = q->attr; t->defense = q->defense; t->exp = q->exp; t->level = q->level; t->rarity = q->rarity; t->tile = (uint8_t)q->tile;
        }
    }
    for (int i = 1; i < e->n_stats; i++) { /* insertion sort = stable, like Vec::sort_by_key */
        mstat_t v = e->stats[i]; int j = i;
        while (j > 0 && e->stats[j - 1].rarity > v.rarity) { e->stats[j] = e->stats[j - 1]; j--; }
        e->stats[j] = v;
    }
    e->level = 0;
    new_level_(e, 1);
    /* Player::build (player.rs:78-91) + Player::init_items (player.rs:136-153) */
    e->hp = e->hp_max = c->init_hp; e->plevel = 1; e->exp = 0;
    e->food_left = c->hunger_time; e->quiet = 0; e->dead = 0;
    if (!MUT(60) && player_init_items(e)) return 1;
    actions_new_level(e, 1);
    if (MUT(60) && player_init_items(e)) return 1; /* M60 App. C-1: the pack's item-stream draws after the player's placement (no golden can tell: nothing reads the item stream in between) */
    return 0;
}
static void mirror_reset(orc_env *e) { /* PlayerState::reset (python/src/lib.rs:52-58) */
    player_status(e, e->status);
    mirror_draw_map(e);
    e->message = 0;
    e->is_terminal = 0;
}
static int symbols_of(const orc_config *c) { /* GameConfig::symbol_max (core/src/lib.rs:150-155) + 1 */
    if (c->n_enemies == 0) return 17;
    int mx = 0;
    for (int i = 0; i < c->n_enemies; i++) {
        int t = c->enemy_builtin[i] >= 0 ? BUILTIN[c->enemy_builtin[i]].tile : c->enemy_custom[i].tile;
        if (t > mx) mx = t;
    }
    return mx - 'A' + 17 + 1;
}
orc_env *orc_new(const orc_config *cfg, uint64_t max_steps) {
    if (cfg->width < 32 || cfg->width > 160 || cfg->height < 16 || cfg->height > 48) return NULL;
    if (cfg->room_num_x * cfg->room_num_y > MAX_ROOMS) return NULL;
    orc_env *e = calloc(1, sizeof *e);
    e->cfg = *cfg; e->max_steps = max_steps; e->W = cfg->width; e->H = cfg->height;
    e->symbols = symbols_of(cfg);
    int n = e->W * e->H;
    e->screen = malloc(n); memset(e->screen, ' ', n);
    e->hist = calloc(n, 1);
    e->placed = calloc((size_t)cfg->room_num_x * cfg->room_num_y + 1, sizeof(mon_t));
    e->active = calloc((size_t)cfg->room_num_x * cfg->room_num_y + 1, sizeof(mon_t));
    if (runtime_build(e)) { orc_free(e); return NULL; } /* GameConfig::build failed in Player::init_items (core/src/lib.rs:209) */
    mirror_reset(e);
    e->steps = 0;
    return e;
}
void orc_free(orc_env *e) { if (!e) return; runtime_free(e); free(e->past_visited); free(e->screen); free(e->hist); free(e->placed); free(e->active); free(e); }
void orc_set_seed(orc_env *e, uint64_t lo, uint64_t hi) { e->cfg.seed_lo = lo; e->cfg.seed_hi = hi; }
int orc_reset(orc_env *e) {
    if (runtime_build(e)) return 1; /* cannot happen: the same config built in orc_new */
    mirror_reset(e);
    e->steps = 0;
    return 0;
}
int orc_react(orc_env *e, uint8_t key) {
    if (e->steps > e->max_steps) return 0;
    int dir = 0, act = key_to_action(key, &dir);
    if (act < 0) return 1;
    if (e->dead) return 2; /* UiState::Mordal + InputCode::Act => IgnoredInput */
    rlist_t *res = &tls_scratch_a; res->n = 0;
    if (process_action(e, act, dir, res)) e->dead = 1;
    e->message = 0;
    int dead = 0;
    for (int i = 0; i < res->n; i++) {
        switch (res->v[i].kind) {
        case RE_REDRAW: mirror_draw_map(e); break;
        case RE_STATUS: player_status(e, e->status); break;
        case RE_GRAVE: dead = 1; break;
        case RE_NOTIFY: e->message |= res->v[i].msg; break;
        }
    }
    e->steps += 1;
    e->is_terminal = dead || e->steps >= e->max_steps;
    return 0;
}
int orc_step_autoreset(orc_env *e, uint8_t key) {
    int rc = orc_react(e, key);
    if (rc) return rc;
    if (e->is_terminal) { orc_reset(e); e->is_terminal = 1; }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* getters                                                                                    */
/* ------------------------------------------------------------------------------------------ */
void orc_screen(const orc_env *e, uint8_t *out) { memcpy(out, e->screen, e->W * e->H); }
void orc_hist(const orc_env *e, uint8_t *out) { memcpy(out, e->hist, e->W * e->H); }
void orc_status(const orc_env *e, uint32_t out[10]) { memcpy(out, e->status, sizeof e->status); }
void orc_flags(const orc_env *e, uint32_t out[5]) {
    out[0] = (uint32_t)e->is_terminal; out[1] = e->message; out[2] = (uint32_t)e->steps; out[3] = (uint32_t)e->dead; out[4] = (uint32_t)e->symbols;
}
void orc_grid(const orc_env *e, uint8_t *surface, uint8_t *attr, uint8_t *doors, int32_t *gold) {
    int n = e->W * e->H;
    if (surface) memcpy(surface, e->fl.surface, n);
    if (attr) memcpy(attr, e->fl.attr, n);
    if (doors) memcpy(doors, e->fl.doors, n);
    if (gold) memcpy(gold, e->fl.gold, n * sizeof(int32_t));
}
void orc_scalars(const orc_env *e, int64_t out[16]) {
    memset(out, 0, 16 * sizeof(int64_t));
    out[0] = e->px; out[1] = e->py; out[2] = e->level; out[3] = e->hp; out[4] = e->hp_max; out[5] = e->exp;
    out[6] = e->plevel; out[7] = e->food_left; out[8] = e->quiet; out[10] = e->n_placed + e->n_active;
    uint32_t st[10]; player_status(e, st); out[9] = st[1];
    out[11] = e->n_pack; out[12] = e->weapon; out[13] = e->armor;
}
/* rooms of the current level, 12 ints each: kind (0 normal 1 maze 2 empty), is_dark, is_visited, has_gold, range x0 y0 x1 y1 (Empty: up_left twice),
 * assigned area x0 y0 x1 y1 -- for tests/shadow_turn.py, which re-derives the turn from the source text and starts every level from the oracle's */
int orc_rooms(const orc_env *e, int32_t *out, int cap) {
    for (int i = 0; i < e->fl.n_rooms && i < cap; i++) {
        const room_t *rm = &e->fl.rooms[i]; int32_t *o = out + 12 * i;
        o[0] = rm->kind; o[1] = rm->is_dark; o[2] = rm->is_visited; o[3] = rm->has_gold;
        if (rm->kind == RK_EMPTY) { o[4] = o[6] = rm->upx; o[5] = o[7] = rm->upy; }
        else { o[4] = rm->range.x0; o[5] = rm->range.y0; o[6] = rm->range.x1; o[7] = rm->range.y1; }
        o[8] = rm->assigned.x0; o[9] = rm->assigned.y0; o[10] = rm->assigned.x1; o[11] = rm->assigned.y1;
    }
    return e->fl.n_rooms;
}
int orc_monsters(const orc_env *e, orc_monster *out, int cap) {
    mon_t all[2 * MAX_MON]; int act[2 * MAX_MON]; int n = 0;
    /* tag by sorting pairs: small n, do a simple insertion by (x,y) */
    for (int k = 0; k < e->n_placed; k++) { all[n] = e->placed[k]; act[n++] = 0; }
    for (int k = 0; k < e->n_active; k++) { all[n] = e->active[k]; act[n++] = 1; }
    for (int i = 1; i < n; i++) {
        mon_t v = all[i]; int a = act[i], j = i;
        while (j > 0 && mon_cmp(&all[j - 1], &v) > 0) { all[j] = all[j - 1]; act[j] = act[j - 1]; j--; }
        all[j] = v; act[j] = a;
    }
    for (int i = 0; i < n && i < cap; i++) {
        out[i].x = all[i].x; out[i].y = all[i].y; out[i].type = e->stats[all[i].type].tile - 'A'; out[i].active = act[i]; out[i].running = all[i].running;
        out[i].hp = all[i].hp; out[i].max_hp = all[i].max_hp; out[i].level = all[i].level; out[i].defense = all[i].defense; out[i].exp = all[i].exp;
    }
    return n;
}
void orc_rng(const orc_env *e, uint32_t s[12], uint64_t counts[3]) {
    const rng_t *r[3] = {&e->rng_d, &e->rng_i, &e->rng_e};
    for (int i = 0; i < 3; i++) { s[4 * i] = r[i]->x; s[4 * i + 1] = r[i]->y; s[4 * i + 2] = r[i]->z; s[4 * i + 3] = r[i]->w; counts[i] = r[i]->count; }
}
/* test hook: Dungeon::new_level + actions::new_level's player placement as on a successful '>' (actions.rs:27-33,121-138), without the turn
 * around it; the mirrors are left alone */
void orc_debug_descend(orc_env *e) { actions_new_level(e, 0); }
int orc_move_enemy_kat(orc_env *e, int fx, int fy, int tx, int ty, int *nx, int *ny) {
    return move_enemy(e, fx, fy, tx, ty, skip_never, nx, ny);
}
void orc_kat_u32(uint64_t lo, uint64_t hi, int n, uint32_t *out) { rng_t r; rng_seed(&r, lo, hi); for (int i = 0; i < n; i++) out[i] = rng_u32(&r); }
uint64_t orc_kat_range64(uint64_t slo, uint64_t shi, uint64_t lo, uint64_t hi, int n_skip) {
    rng_t r; rng_seed(&r, slo, shi);
    for (int i = 0; i < n_skip; i++) rng_u32(&r);
    return range64(&r, lo, hi);
}

/* ------------------------------------------------------------------------------------------ */
/* observation encoders                                                                       */
/* ------------------------------------------------------------------------------------------ */
static int tile_to_sym(uint8_t t) { /* Symbol::from_tile (symbol.rs:17-40) */
    switch (t) {
    case ' ': return 0; case '@': return 1; case '#': return 2; case '.': return 3; case '-': case '|': return 4;
    case '%': return 5; case '+': return 6; case '^': return 7; case '!': return 8; case '?': return 9; case ']': return 10;
    case ')': return 11; case '/': return 12; case '*': return 13; case ':': return 14; case '=': return 15; case ',': return 16;
    default: if (t >= 'A' && t <= 'Z') return t - 'A' + 17; return -1;
    }
}
/* StatusFlagInner::to_vector (flags.rs:67-87) */
static const int STATUS_ORDER[9] = {0, 2, 3, 4, 5, 6, 7, 8, 9}; /* dungeon_level, hp_cur, hp_max, str_cur, str_max, defense, player_level, exp, hunger */
int orc_status_vec(const uint32_t st[10], uint32_t flag, int32_t *out) {
    int n = 0;
    for (int b = 0; b < 9; b++) if (flag & (1u << b)) out[n++] = (int32_t)st[STATUS_ORDER[b]];
    return n;
}
/* StatusFlagInner::copy_status (flags.rs:88-115) */
static int copy_status(const uint32_t st[10], uint32_t flag, int start, int hw, float *out) {
    int off = start;
    for (int b = 0; b < 9; b++) if (flag & (1u << b)) {
        float v = (float)(int32_t)st[STATUS_ORDER[b]];
        for (int i = 0; i < hw; i++) out[off * hw + i] = v;
        off++;
    }
    return off;
}
int orc_gray_image(const uint8_t *screen, int h, int w, int symbols, const uint32_t st[10], uint32_t flag, const uint8_t *hist, float *out) {
    int hw = h * w, c = 1 + __builtin_popcount(flag) + (hist ? 1 : 0);
    memset(out, 0, (size_t)c * hw * sizeof(float));
    for (int i = 0; i < hw; i++) { /* python/src/lib.rs:72-87 */
        int s = tile_to_sym(screen[i]);
        if (s < 0) return 1;
        out[i] = (float)(uint8_t)s / (float)(uint8_t)symbols;
    }
    int off = copy_status(st, flag, 1, hw, out);
    if (hist) for (int i = 0; i < hw; i++) out[off * hw + i] = hist[i] ? 1.0f : 0.0f; /* copy_hist (:105-111) */
    return 0;
}
int orc_symbol_image(const uint8_t *screen, int h, int w, int symbols, const uint32_t st[10], uint32_t flag, const uint8_t *hist, float *out) {
    int hw = h * w, c = symbols + __builtin_popcount(flag) + (hist ? 1 : 0);
    memset(out, 0, (size_t)c * hw * sizeof(float));
    int symbol_max = symbols - 1; /* python/src/lib.rs:96-102 -> symbol.rs:51-71 */
    for (int ch = 0; ch < symbol_max; ch++)
        for (int i = 0; i < hw; i++) {
            int s = tile_to_sym(screen[i]);
            if (s < 0 || s >= symbol_max) return 1;
            out[ch * hw + i] = s == ch ? 1.0f : 0.0f;
        }
    int off = copy_status(st, flag, symbols, hw, out);
    if (hist) for (int i = 0; i < hw; i++) out[off * hw + i] = hist[i] ? 1.0f : 0.0f;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* batch driver: static partition of envs over a pthread pool (CPU baseline)  */
/* ------------------------------------------------------------------------------------------ */
struct orc_batch {
    int n, n_threads;
    orc_env **envs;
    pthread_t *threads;
    pthread_barrier_t bar_start, bar_end;
    const uint8_t *keys; float *obs; int stop; int err;
    struct orc_worker { struct orc_batch *b; int tid; } *workers;
};
static void batch_work(orc_batch *b, int tid) {
    int lo = (int)((int64_t)b->n * tid / b->n_threads), hi = (int)((int64_t)b->n * (tid + 1) / b->n_threads);
    for (int i = lo; i < hi; i++) {
        orc_env *e = b->envs[i];
        if (orc_step_autoreset(e, b->keys[i])) b->err = 1;
        if (b->obs) orc_gray_image(e->screen, e->H, e->W, e->symbols, e->status, 0, NULL, b->obs + (size_t)i * e->H * e->W);
    }
}
static void *batch_thread(void *arg) {
    struct orc_worker *w = arg;
    for (;;) {
        pthread_barrier_wait(&w->b->bar_start);
        if (w->b->stop) return NULL;
        batch_work(w->b, w->tid);
        pthread_barrier_wait(&w->b->bar_end);
    }
}
orc_batch *orc_batch_new(const orc_config *cfgs, int n, uint64_t max_steps, int n_threads) {
    orc_batch *b = calloc(1, sizeof *b);
    b->n = n; b->n_threads = n_threads < 1 ? 1 : n_threads;
    b->envs = calloc(n, sizeof(orc_env *));
    for (int i = 0; i < n; i++) { b->envs[i] = orc_new(&cfgs[i], max_steps); if (!b->envs[i]) { orc_batch_free(b); return NULL; } }
    if (b->n_threads > 1) {
        pthread_barrier_init(&b->bar_start, NULL, b->n_threads);
        pthread_barrier_init(&b->bar_end, NULL, b->n_threads);
        b->threads = calloc(b->n_threads, sizeof(pthread_t));
        b->workers = calloc(b->n_threads, sizeof(*b->workers));
        for (int t = 1; t < b->n_threads; t++) { b->workers[t].b = b; b->workers[t].tid = t; pthread_create(&b->threads[t], NULL, batch_thread, &b->workers[t]); }
    }
    return b;
}
void orc_batch_free(orc_batch *b) {
    if (!b) return;
    if (b->threads) {
        b->stop = 1;
        pthread_barrier_wait(&b->bar_start);
        for (int t = 1; t < b->n_threads; t++) pthread_join(b->threads[t], NULL);
        pthread_barrier_destroy(&b->bar_start); pthread_barrier_destroy(&b->bar_end);
        free(b->threads); free(b->workers);
    }
    for (int i = 0; i < b->n; i++) orc_free(b->envs[i]);
    free(b->envs); free(b);
}
orc_env *orc_batch_env(orc_batch *b, int i) { return b->envs[i]; }
int orc_batch_step(orc_batch *b, const uint8_t *keys, float *obs) {
    b->keys = keys; b->obs = obs; b->err = 0;
    if (b->n_threads > 1) { pthread_barrier_wait(&b->bar_start); batch_work(b, 0); pthread_barrier_wait(&b->bar_end); }
    else batch_work(b, 0);
    return b->err;
}
