// rg_config.h -- host-side GameConfig (core/src/lib.rs:42-86) flattened into the POD the kernels take.
#pragma once
#include <cstdint>
#include <string>

#define RG_MAX_ROOMS 32     // room_num_x * room_num_y (reference default 3x3; no limit there; 32 = the width of the room bitmasks)
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
    uint32_t appear_rate_gold, appear_rate_nogold;
    int32_t n_enemies;                            // length of the rarity-sorted table
    RgMonStat mon[RG_MAX_ENEMY_KINDS + 6];        // monster statuses, stable-sorted by rarity (enemies.rs:250-261)
    uint32_t level_exps[21];                      // Leveling::exps (player.rs:308-343)
    int32_t n_level_exps;
    int32_t symbols;                              // symbol_max + 1 (core/src/lib.rs:150-155)
    uint32_t max_steps;
    int32_t auto_reset;
};

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
};

// Returns "" on success, else the error text ("Failed to parse config: ...").
std::string rg_parse_config(const char *json, RgParsed *out);
// true if a and b agree on every field the device code reads
bool rg_config_equal(const RgConfig &a, const RgConfig &b);
std::string rg_dump_config_json(const RgParsed &p, uint64_t seed_lo, uint64_t seed_hi, bool has_seed);
