/*
 * rogue_oracle.h -- CPU restatement of the kngwyu/rogue-gym hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity ORACLE: a plain-C, single-env-at-a-time restatement of the reference
 * Rust engine (dungeon generation + turn step + screen draw + observation encode). It exists
 * only so tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can check / time
 * the HIP product path against it.  The product (rogue-gym_amd/) never links or imports it.
 *
 * Pinning: the reference cannot be compiled or imported here (no Rust toolchain, no vendored
 * crates), so the oracle is pinned by the reference's own golden vectors
 * (python/tests/data.py, test_ff_env.py, test_st_env.py, test_parallel.py,
 * core/src/dungeon/rogue/mod.rs:566-578) -- see tests/test_oracle_golden.py.
 * Behaviour not covered by those goldens (levels >= 4, death, level-up, bats, search, ...)
 * follows the reference source text only: "parity unpinned" for those paths.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef ROGUE_ORACLE_H
#define ROGUE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a custom monster status (Preset::Custom(Status), character/enemies.rs:87-121) */
typedef struct orc_monstat {
    int32_t n_attack, att_times[4], att_max[4];
    int32_t attr, defense;
    uint32_t exp;
    int32_t level, rarity, tile;
} orc_monstat;

/* item tables and the player's initial pack (item/weapon.rs:129-140, item/armor.rs:133-139, item/mod.rs:166-229).  Only the fields something
 * reachable from the RL action set reads are carried (throw dice, worth, launcher, appear rates: not). */
#define ORC_NAME_CAP 32
#define ORC_MAX_STATS 16
#define ORC_MAX_INIT_ITEMS 16
typedef struct orc_weapon_stat { char name[ORC_NAME_CAP]; uint64_t wield_times; int64_t wield_max; uint32_t init_lo, init_hi; uint32_t attr; } orc_weapon_stat;
typedef struct orc_armor_stat { char name[ORC_NAME_CAP]; int32_t def; } orc_armor_stat;
enum { ORC_KIND_ARMOR = 0, ORC_KIND_FOOD, ORC_KIND_GOLD, ORC_KIND_POTION, ORC_KIND_RING, ORC_KIND_SCROLL, ORC_KIND_WAND, ORC_KIND_WEAPON }; /* ItemKind, item/mod.rs:32-41 */
#define ORC_ATTR_IS_MANY 4u      /* ItemAttr::IS_MANY (item/mod.rs:131) */
#define ORC_ATTR_EQUIPPED 8u     /* ItemAttr::IS_EQUIPPED (item/mod.rs:132) */
typedef struct orc_item { /* Item (item/mod.rs:224-229) with the Weapon / Armor payload flattened */
    int32_t kind; uint32_t how_many, attr;
    char name[ORC_NAME_CAP];
    uint64_t wield_times; int64_t wield_max, hit_plus, dam_plus; /* Weapon (weapon.rs:83-92) */
    int32_t def, def_plus;                                        /* Armor (armor.rs:88-94) */
} orc_item;
enum { ORC_INIT_NOINIT = 0, ORC_INIT_ARMOR, ORC_INIT_WEAPON };    /* InitItem (item/mod.rs:166-178) */
typedef struct orc_init_item {
    int32_t tag;
    char name[ORC_NAME_CAP];
    uint32_t num_plus; int32_t hit_plus, dam_plus, def_plus;
    orc_item item; /* Noinit */
} orc_init_item;

/* Flat config (the test harness parses the JSON; the oracle stays JSON-free).
 * Field defaults: core/src/lib.rs:134-140, dungeon/rogue/mod.rs:23-134,
 * character/enemies.rs:56-85, character/player.rs:37-66, item/gold.rs:27-52. */
typedef struct orc_config {
    int32_t width, height;
    uint64_t seed_lo, seed_hi; /* u128 seed */
    int32_t hide_dungeon;
    int32_t room_num_x, room_num_y, min_room_x, min_room_y;
    uint32_t max_empty_rooms, amulet_level, maze_rate_inv, dark_level;
    uint32_t hidden_passage_rate_inv, locked_door_rate_inv, max_extra_edges;
    uint32_t door_unlock_rate_inv, passage_unlock_rate_inv;
    uint32_t gold_rate_inv, gold_base, gold_per_level, gold_minimum;
    uint32_t hunger_time;
    int64_t init_hp;
    uint32_t appear_rate_gold, appear_rate_nogold;
    int32_t n_enemies;          /* number of entries of enemy_builtin in use (0 = no enemies) */
    int32_t enemy_builtin[32];  /* indices into BUILTIN_ENEMIES (enemies.rs:474-761), or -1: use enemy_custom[i] */
    orc_monstat enemy_custom[32];
    int32_t choose_width;       /* 64 (what reproduces the goldens) or 32; SliceRandom::choose */
    /* item::Config tables + player.{init_items,max_items} (weapon.rs:34-47, armor.rs:46-60, player.rs:26-29) */
    int32_t n_weapons, n_armors, n_init_items;
    orc_weapon_stat weapons[ORC_MAX_STATS];
    orc_armor_stat armors[ORC_MAX_STATS];
    orc_init_item init_items[ORC_MAX_INIT_ITEMS];
    uint64_t max_items;
} orc_config;

void orc_config_default(orc_config *c);

typedef struct orc_env orc_env;

/* GameStateImpl::new (python/src/state_impls.rs:20-38). Returns NULL on invalid size, or when Player::init_items fails (an InitItem names an
 * item that is not in the table; the pack is full: item/mod.rs:216-220,415-419). */
orc_env *orc_new(const orc_config *cfg, uint64_t max_steps);
void orc_free(orc_env *e);
/* GameState.set_seed / Instruction::Seed: takes effect at the next reset. */
void orc_set_seed(orc_env *e, uint64_t lo, uint64_t hi);
/* GameStateImpl::reset (state_impls.rs:38-44) */
int orc_reset(orc_env *e);
/* GameStateImpl::react (state_impls.rs:51-79).  0 = ok, 1 = invalid key, 2 = ignored input
 * (action key while in the Grave modal, core/src/lib.rs:301-315). */
int orc_react(orc_env *e, uint8_t key);
/* one env of ThreadConductor::step (thread_impls.rs:61-81): react, then auto-reset on
 * terminal with is_terminal forced true.  Returns the error code of react. */
int orc_step_autoreset(orc_env *e, uint8_t key);

/* ---- mirrors (PlayerState, python/src/lib.rs:29-38) ---- */
void orc_screen(const orc_env *e, uint8_t *out /* H*W */);
void orc_hist(const orc_env *e, uint8_t *out /* H*W, 0/1 */);
void orc_status(const orc_env *e, uint32_t out[10]); /* Status::to_vec order, player.rs:418-430 */
/* out[0]=is_terminal out[1]=message flags out[2]=steps out[3]=dead(grave) out[4]=symbols */
void orc_flags(const orc_env *e, uint32_t out[5]);

/* ---- internal state, for deep GPU==oracle comparison ---- */
typedef struct orc_monster {
    int32_t x, y, type /* glyph - 'A' (= builtin index for builtin monsters) */, active, running;
    int64_t hp, max_hp, level;
    int32_t defense;
    uint32_t exp;
} orc_monster;
/* surface: 0 Passage 1 Floor 2 WallX 3 WallY 4 Stair 5 Door 6 Trap 7 None (rogue/mod.rs:137-147)
 * attr: CellAttr bits (dungeon/field.rs:107-124) */
void orc_grid(const orc_env *e, uint8_t *surface, uint8_t *attr, uint8_t *doors, int32_t *gold);
/* out: px,py,level,hp,hp_max,exp,plevel,food_left,quiet,gold,n_monsters,pack items,equipped weapon slot (-1 none),equipped armor slot (-1 none) */
void orc_scalars(const orc_env *e, int64_t out[16]);
int orc_monsters(const orc_env *e, orc_monster *out, int cap); /* sorted by (x,y) */
int orc_rooms(const orc_env *e, int32_t *out, int cap);          /* 12 ints per room, see rogue_oracle.c */
/* rng state: 3 streams (dungeon,item,enemy) x {x,y,z,w}; counts = u32 outputs consumed */
void orc_rng(const orc_env *e, uint32_t state[12], uint64_t counts[3]);
/* Dungeon::move_enemy with skip = |_| false (rogue/mod.rs:339-375), for the reference KAT
 * rogue/mod.rs:566-578.  Returns 0 CantMove, 1 CanMove (nx,ny set), 2 Reach. */
int orc_move_enemy_kat(orc_env *e, int fx, int fy, int tx, int ty, int *nx, int *ny);
/* passages::edges (passages.rs:181-219) on the half-open rect [x0,x1) x [y0,y1), for the reference KAT passages.rs:272-296.
 * direction: 0 Up 1 Down 2 Left 3 Right (Direction order, coord.rs:198-208).  Returns the number of cells written to xs / ys. */
int orc_kat_edges(int x0, int y0, int x1, int y1, int direction, int inclusive, int *xs, int *ys);
/* test hook: generate the next level and place the player as on a successful '>' (actions.rs:27-33,121-138) without the turn around it */
void orc_debug_descend(orc_env *e);

/* ---- observation encoders (python/src/lib.rs:72-205, flags.rs:67-115, symbol.rs:17-71) ---- */
int orc_status_vec(const uint32_t status[10], uint32_t flag, int32_t *out); /* returns len */
/* out is [C,H,W] f32 with C = 1 + popcount(flag) (+1 with hist). returns 0, or 1 on bad tile */
int orc_gray_image(const uint8_t *screen, int h, int w, int symbols, const uint32_t status[10],
                   uint32_t flag, const uint8_t *hist /* NULL = no hist plane */, float *out);
/* C = symbols + popcount(flag) (+1).  returns 1 if a tile's symbol >= symbols-1 (e.g. 'Z') */
int orc_symbol_image(const uint8_t *screen, int h, int w, int symbols, const uint32_t status[10],
                     uint32_t flag, const uint8_t *hist, float *out);

/* ---- xorshift / rand-0.7 primitives exposed for known-answer tests ---- */
void orc_kat_u32(uint64_t seed_lo, uint64_t seed_hi, int n, uint32_t *out);
uint64_t orc_kat_range64(uint64_t seed_lo, uint64_t seed_hi, uint64_t lo, uint64_t hi, int n_skip);

/* ---- batch driver (CPU baseline timing; one env per task over a pthread pool) ---- */
typedef struct orc_batch orc_batch;
orc_batch *orc_batch_new(const orc_config *cfgs, int n, uint64_t max_steps, int n_threads);
void orc_batch_free(orc_batch *b);
orc_env *orc_batch_env(orc_batch *b, int i);
/* lock-step ThreadConductor::step over all envs + gray obs encode into obs [n,1,H,W] (may be NULL) */
int orc_batch_step(orc_batch *b, const uint8_t *keys, float *obs);

#ifdef __cplusplus
}
#endif
#endif
