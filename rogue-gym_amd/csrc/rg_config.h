// rg_config.h -- host-side GameConfig (core/src/lib.rs:42-86) flattened into the POD the kernels take.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#define RG_MAX_ROOMS 384    // room_num_x * room_num_y.  The reference has no limit (rooms.rs:165-211); geometry has: 160 x 48 with min_room_size 3 holds at most
                            // 40 x 9 = 360 rooms.  Three generator instances (rg_kernels.hip): <= 32 rooms, <= 64, <= 384.
// words of an env's observation record (rg_obs.hip ObsTabs): nr monster words, the player's position, nr room rects, nr room-meta bytes; a multiple of 4 (16-byte stores)
#define RG_OBS_REC_WORDS(nr) ((((nr) * 2 + 1 + ((nr) + 3) / 4) + 3) & ~3)
#define RG_OVL_MAX 9          // rooms (= monsters) up to which k_step keeps the screen mirror current itself (rg_state.h ovl)
#define RG_OBS_MAX_ROOMS 64 // the fused observation kernel stages its overlay tables in LDS for up to 64 rooms; larger grids take the unfused render + encode
#define RG_MAX_ENEMY_KINDS 26
#define RG_MAX_W 160        // core/src/lib.rs:134-140
#define RG_MAX_H 48
#define RG_MIN_W 32
#define RG_MIN_H 16
#define RG_DIST_SLOTS 9     // DistCache holds up to 9 maps (rogue/mod.rs:492-518)

// One monster status (character/enemies.rs:110-121): a builtin preset or a custom one from the config
struct RgMonStat {
    uint32_t exp;
    int32_t defense;
    int16_t level;
    uint16_t attr;          // EnemyAttr bits (enemies.rs:126-139)
    uint8_t tile;           // glyph 'A'..'Z'
    uint8_t rarity;
    uint8_t n_att;          // attack dice in use (<= 4)
    uint8_t att[4][2];      // {times, max} per die
    uint8_t pad;
};

// Everything the device code needs; identical for all envs of a handle (seeds are per-env arrays).
struct RgConfig {
    int32_t width, height;
    int32_t hide_dungeon;
    int32_t room_num_x, room_num_y, min_room_x, min_room_y;
    uint32_t max_empty_rooms, amulet_level, maze_rate_inv, dark_level;
    uint32_t hidden_passage_rate_inv, locked_door_rate_inv, max_extra_edges;
    uint32_t door_unlock_rate_inv, passage_unlock_rate_inv;
    uint32_t gold_rate_inv, gold_base, gold_per_level, gold_minimum;
    uint32_t hunger_time;
    int32_t init_hp;
    // Player::init_items resolved on the host (player.rs:136-153, item/mod.rs:181-221; rg_items.cpp): what the equipped weapon / armor and the
    // pack contribute to the turn.  Defaults = the rogue pack: mace 2d4 +1,+1 (weapon.rs:179-188,200-203), ring mail 3 + 1 (armor.rs:68-73).
    int32_t wpn_times, wpn_max;           // Item::at_weild of Player::weapon, or 1d4 bare hands (fight.rs:27-33)
    int32_t wpn_hit_plus, wpn_dam_plus;   // Weapon::hit_plus / dam_plus (fight.rs:21-24)
    int32_t armor_def;                    // Player::arm = def + def_plus of the equipped armor, 0 without one (player.rs:125-132)
    uint32_t init_gold;                   // how_many of the pack's first Gold item (core/src/lib.rs:348-353)
    int32_t can_pickup;                   // ItemBox::entry finds a Gold item to merge into or a free slot (itembox.rs:30-40); else gold stays on the floor
    int32_t n_init_draws;                 // InitItem::Weapon entries: one `rng.range(init_num)` each on the item stream (weapon.rs:159); RgState::init_draws
    uint32_t appear_rate_gold, appear_rate_nogold;
    int32_t n_enemies;                            // length of the rarity-sorted table
    RgMonStat mon[RG_MAX_ENEMY_KINDS + 6];        // monster statuses, stable-sorted by rarity (enemies.rs:250-261)
    uint32_t level_exps[21];                      // Leveling::exps (player.rs:308-343)
    int32_t n_level_exps;
    int32_t symbols;                              // symbol_max + 1 (core/src/lib.rs:150-155)
    uint32_t max_steps;
    int32_t auto_reset;
    // derived (rg_config_derive; not part of rg_config_equal's comparison, which sees parsed configs): the assigned-area size width / room_num_x, height /
    // room_num_y (rooms.rs:192-209) and the correctly rounded reciprocals room_id_of / assigned_area divide by (rg_device.h small_div_inv)
    int32_t rsx, rsy;
    float inv_rsx, inv_rsy, inv_rnx;
};
static inline void rg_config_derive(RgConfig *c) {
    c->rsx = c->width / c->room_num_x; c->rsy = c->height / c->room_num_y;
    c->inv_rsx = 1.0f / (float)c->rsx; c->inv_rsy = 1.0f / (float)c->rsy; c->inv_rnx = 1.0f / (float)c->room_num_x;
}

struct RgParsed {
    RgConfig cfg;
    bool has_seed;
    uint64_t seed_lo, seed_hi;
    bool has_seed_range;
    unsigned __int128 seed_range[2];
    RgMonStat presets[RG_MAX_ENEMY_KINDS + 6];   // as given (unsorted), for dump_config
    int preset_builtin[RG_MAX_ENEMY_KINDS + 6];  // builtin index, or -1 for a custom status
    std::string preset_name[RG_MAX_ENEMY_KINDS + 6];
    uint32_t preset_gold[RG_MAX_ENEMY_KINDS + 6];
    int n_presets;
    bool enemies_given;
    // item / player-pack configuration (rg_items.cpp)
    std::vector<uint32_t> init_draws;  // (lo, hi) per InitItem::Weapon in list order: half-open init_num of the matched WeaponStatus
    std::string weapon_json, armor_json, init_items_json;  // canonical re-serialisation of item.weapon / item.armor / player.init_items
    bool weapon_default, armor_default, init_items_default;
    uint64_t max_items;                // player.max_items (ItemBox capacity, player.rs:26-27,83)
    int64_t init_str;                  // read by serde, never by the engine (StatusInner::from_config hard-codes 16, player.rs:284): carried for dump_config
    uint32_t heal_threshold;           // likewise unused by the engine (Player::heal hard-codes 20, player.rs:221-240)
    bool enable_trap;                  // dungeon.enable_trap: read by serde, no trap code exists (rogue/mod.rs:35)
    bool exps_given;
    std::string keymap_json;           // `keymap`: GameStateImpl::new overwrites it with KeyMap::ai (python/src/state_impls.rs:27); carried for dump_config
};

// Returns "" on success, else the error text ("Failed to parse config: ...").
std::string rg_parse_config(const char *json, RgParsed *out);
// true if a and b agree on everything the device code reads (the POD and the init-draw list)
bool rg_config_equal(const RgParsed &a, const RgParsed &b);
namespace rgjson { struct JVal; }
// item.weapon / item.armor / player.{init_items,max_items} -> the resolved fields of cfg + init_draws + the canonical texts (rg_items.cpp).
// item / player may be NULL (section absent).  Returns "" or the error text.
std::string rg_resolve_items(const rgjson::JVal *item, const rgjson::JVal *player, RgParsed *out);
// JSON array describing every GameConfig key the reference's serde structs know and what the stepper does with it (tests/test_config_schema.py)
std::string rg_config_schema_json();
std::string rg_dump_config_json(const RgParsed &p, uint64_t seed_lo, uint64_t seed_hi, bool has_seed);
