// rg_config.cpp -- minimal JSON reader for GameConfig (core/src/lib.rs:42-86 and the per-module
// Config structs: rogue/mod.rs:23-134, enemies.rs:18-85, player.rs:17-66, item/gold.rs:6-52).
// Like serde without deny_unknown_fields, keys the reference's structs do not know are ignored; wrong types are errors.  Every key the
// structs DO know is either honoured or provably inert in the engine (rg_config_schema_json lists which and why); none is dropped silently.
#include "rg_config.h"
#include "rg_json.h"

#include <cctype>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

using rgjson::JVal;
using rgjson::Parser;

namespace {

struct Ctx { std::string err; };

bool get_int(Ctx &c, const JVal *o, const char *key, int64_t lo, int64_t hi, int64_t *out) {
    const JVal *v = o->get(key);
    if (!v) return true;
    if (v->kind != JVal::Num || !v->is_int) { c.err = std::string("invalid type for `") + key + "`, expected integer"; return false; }
    if (v->mag > (unsigned __int128)0x7fffffffffffffffLL) { c.err = std::string("`") + key + "` out of range"; return false; }
    int64_t x = v->neg ? -(int64_t)v->mag : (int64_t)v->mag;
    if (x < lo || x > hi) { c.err = std::string("`") + key + "` out of range"; return false; }
    *out = x;
    return true;
}
#define GET_U32(obj, key, field) do { int64_t t_ = (field); if (!get_int(c, obj, key, 0, 0xffffffffLL, &t_)) return c.err; (field) = (uint32_t)t_; } while (0)
#define GET_I32(obj, key, field) do { int64_t t_ = (field); if (!get_int(c, obj, key, -0x80000000LL, 0x7fffffffLL, &t_)) return c.err; (field) = (int32_t)t_; } while (0)

// BUILTIN_ENEMIES (character/enemies.rs:474-761), index = tile - 'A'
struct Builtin { const char *name; uint32_t exp; int defense; int level; int attr; int rarity; uint32_t gold; int n_att; int att[3][2]; };
const Builtin BUILTIN[26] = {
    {"aquator", 20, 2 | 8, 5, 1 | 32, 12, 0, 1, {{0, 0}}},
    {"bat", 1, 3, 1, 2 | 512, 2, 0, 1, {{1, 2}}},
    {"centaur", 17, 4, 4, 0, 10, 15, 3, {{1, 2}, {1, 5}, {1, 5}}},
    {"dragon", 5000, 3, 10, 1, 25, 100, 3, {{1, 8}, {1, 8}, {3, 10}}},
    {"emu", 2, 7, 1, 1, 1, 0, 1, {{1, 2}}},
    {"venus flytrap", 80, 3, 8, 1, 15, 0, 0, {{0, 0}}},
    {"griffin", 2000, 2, 13, 2 | 1 | 4, 23, 20, 2, {{4, 3}, {3, 5}}},
    {"hobgoblin", 3, 5, 1, 1, 4, 0, 1, {{1, 8}}},
    {"icemonster", 5, 9, 1, 256, 5, 0, 1, {{0, 0}}},
    {"jabberwock", 3000, 6, 15, 0, 24, 70, 2, {{2, 12}, {2, 4}}},
    {"kestrel", 1, 7, 1, 1, 0, 0, 1, {{1, 4}}},
    {"leperachaun", 10, 8, 3, 64, 9, 0, 1, {{1, 1}}},
    {"medusa", 200, 2, 8, 1, 21, 40, 3, {{3, 4}, {3, 4}, {2, 5}}},
    {"nymph", 37, 9, 3, 0, 13, 100, 1, {{0, 0}}},
    {"orc", 5, 6, 1, 8, 7, 15, 1, {{1, 8}}},
    {"phantom", 120, 3, 8, 16, 18, 0, 1, {{4, 4}}},
    {"quagga", 15, 3, 3, 1, 11, 0, 2, {{1, 5}, {1, 5}}},
    {"rattlesnake", 9, 3, 2, 128 | 1, 6, 0, 1, {{1, 6}}},
    {"snake", 2, 5, 1, 1, 3, 0, 1, {{1, 3}}},
    {"troll", 120, 4, 6, 1 | 4, 16, 50, 3, {{1, 8}, {1, 8}, {2, 6}}},
    {"urvile", 190, -2, 7, 1, 20, 0, 3, {{1, 9}, {1, 9}, {2, 9}}},
    {"vampire", 350, 1, 8, 1 | 4, 22, 20, 1, {{1, 19}}},
    {"wraith", 55, 4, 5, 0, 17, 0, 1, {{1, 6}}},
    {"xeroc", 100, 7, 7, 0, 19, 30, 1, {{4, 4}}},
    {"yeti", 50, 6, 4, 0, 14, 30, 2, {{1, 6}, {1, 6}}},
    {"zombie", 6, 8, 2, 1, 8, 0, 1, {{1, 8}}},
};
void fill_builtin(RgParsed *p, int slot, int id) {
    RgMonStat &m = p->presets[slot];
    memset(&m, 0, sizeof m);
    const Builtin &b = BUILTIN[id];
    m.exp = b.exp; m.defense = b.defense; m.level = (int16_t)b.level; m.attr = (uint16_t)b.attr; m.tile = (uint8_t)('A' + id);
    m.rarity = (uint8_t)b.rarity; m.n_att = (uint8_t)b.n_att;
    for (int k = 0; k < b.n_att; k++) { m.att[k][0] = (uint8_t)b.att[k][0]; m.att[k][1] = (uint8_t)b.att[k][1]; }
    p->preset_builtin[slot] = id; p->preset_name[slot] = b.name; p->preset_gold[slot] = b.gold;
}
const uint32_t DEFAULT_EXPS[21] = {10, 20, 40, 80, 160, 320, 640, 1300, 2600, 5200, 13000, 26000, 50000, 100000,
                                   200000, 400000, 800000, 2000000, 4000000, 8000000, 0xFFFFFFFFu};

void set_defaults(RgParsed *p) {
    memset(&p->cfg, 0, sizeof p->cfg);
    p->has_seed = p->has_seed_range = p->enemies_given = false;
    p->seed_lo = p->seed_hi = 0; p->seed_range[0] = p->seed_range[1] = 0;
    RgConfig &c = p->cfg;
    c.width = 80; c.height = 24; c.hide_dungeon = 1;
    c.room_num_x = 3; c.room_num_y = 3; c.min_room_x = 4; c.min_room_y = 4;
    c.max_empty_rooms = 3; c.amulet_level = 25; c.maze_rate_inv = 15; c.dark_level = 10;
    c.hidden_passage_rate_inv = 40; c.locked_door_rate_inv = 5; c.max_extra_edges = 5;
    c.door_unlock_rate_inv = 5; c.passage_unlock_rate_inv = 3;
    c.gold_rate_inv = 2; c.gold_base = 50; c.gold_per_level = 10; c.gold_minimum = 2;
    c.hunger_time = 1300; c.init_hp = 12;
    c.appear_rate_gold = 80; c.appear_rate_nogold = 25;
    p->n_presets = 26;
    for (int i = 0; i < 26; i++) fill_builtin(p, i, i);
    memcpy(c.level_exps, DEFAULT_EXPS, sizeof DEFAULT_EXPS);
    c.n_level_exps = 21;
    p->init_str = 16; p->heal_threshold = 20; p->enable_trap = true; p->exps_given = false; p->keymap_json.clear();
    (void)rg_resolve_items(nullptr, nullptr, p);  // the default pack: mace 2d4 +1,+1, ring mail 3 + 1, draws 1..2, 1..2, 8..17, gold 0
}

// KeyMap::default (input.rs:23-66) in its serialised form (input.rs:141-198; data/config-default.json holds the same 28 entries): `keymap`
// is carried only so that dump_config can skip it when default and write it back otherwise
bool keymap_is_default(const JVal &km) {
    static const struct { const char *key, *code; } D[28] = {
        {"l", "{\"Act\": {\"Move\": \"Right\"}}"}, {"k", "{\"Act\": {\"Move\": \"Up\"}}"}, {"j", "{\"Act\": {\"Move\": \"Down\"}}"},
        {"h", "{\"Act\": {\"Move\": \"Left\"}}"}, {"u", "{\"Act\": {\"Move\": \"RightUp\"}}"},
        {"y", "{\"Both\": {\"act\": {\"Move\": \"LeftUp\"}, \"sys\": \"Yes\"}}"}, {"n", "{\"Both\": {\"act\": {\"Move\": \"RightDown\"}, \"sys\": \"No\"}}"},
        {"b", "{\"Act\": {\"Move\": \"LeftDown\"}}"}, {"L", "{\"Act\": {\"MoveUntil\": \"Right\"}}"}, {"K", "{\"Act\": {\"MoveUntil\": \"Up\"}}"},
        {"J", "{\"Act\": {\"MoveUntil\": \"Down\"}}"}, {"H", "{\"Act\": {\"MoveUntil\": \"Left\"}}"}, {"U", "{\"Act\": {\"MoveUntil\": \"RightUp\"}}"},
        {"Y", "{\"Act\": {\"MoveUntil\": \"LeftUp\"}}"}, {"N", "{\"Act\": {\"MoveUntil\": \"RightDown\"}}"}, {"B", "{\"Act\": {\"MoveUntil\": \"LeftDown\"}}"},
        {"s", "{\"Act\": \"Search\"}"}, {".", "{\"Act\": \"NoOp\"}"}, {">", "{\"Act\": \"DownStair\"}"},
        {"Up", "{\"Act\": {\"Move\": \"Up\"}}"}, {"Down", "{\"Act\": {\"Move\": \"Down\"}}"}, {"Left", "{\"Act\": {\"Move\": \"Left\"}}"},
        {"Right", "{\"Act\": {\"Move\": \"Right\"}}"}, {"Esc", "{\"Sys\": \"Cancel\"}"}, {"S", "{\"Sys\": \"Save\"}"}, {"Q", "{\"Sys\": \"Quit\"}"},
        {"i", "{\"Sys\": \"Inventory\"}"}, {" ", "{\"Sys\": \"Continue\"}"},
    };
    if (km.obj.size() != 28) return false;
    for (auto &d : D) {
        const JVal *v = km.get(d.key);
        if (!v || rgjson::to_string(*v) != d.code) return false;
    }
    return true;
}

std::string finish(RgParsed *p) {
    RgConfig &c = p->cfg;
    // GameConfig::to_global size checks (core/src/lib.rs:166-184)
    if (c.width < RG_MIN_W) return "Invalid Setting: screen width is too narrow";
    if (c.width > RG_MAX_W) return "Invalid Setting: screen width is too wide";
    if (c.height < RG_MIN_H) return "Invalid Setting: screen height is too narrow";
    if (c.height > RG_MAX_H) return "Invalid Setting: screen height is too wide";
    if (c.room_num_x < 1 || c.room_num_y < 1 || c.room_num_x * c.room_num_y > RG_MAX_ROOMS)
        return "Invalid Setting: room_num_x * room_num_y must be in 1..=" + std::to_string(RG_MAX_ROOMS) + " for the HIP stepper";
    int rsx = c.width / c.room_num_x, rsy = c.height / c.room_num_y;
    if (c.min_room_x < 3 || c.min_room_y < 3 || c.min_room_x >= rsx || c.min_room_y >= rsy - 1)
        return "Invalid Setting: min_room_size does not fit the room grid";
    if (c.dark_level == 0 || c.maze_rate_inv == 0 || c.hidden_passage_rate_inv == 0 || c.locked_door_rate_inv == 0 ||
        c.max_extra_edges == 0 || c.door_unlock_rate_inv == 0 || c.passage_unlock_rate_inv == 0 || c.gold_rate_inv == 0)
        return "Invalid Setting: zero rate (the reference asserts `invalid range!!`)";
    // EnemyHandler::new: stable sort by rarity
    c.n_enemies = p->n_presets;
    memset(c.mon, 0, sizeof c.mon);
    for (int i = 0; i < c.n_enemies; i++) c.mon[i] = p->presets[i];
    for (int i = 1; i < c.n_enemies; i++) {  // insertion sort = stable, like Vec::sort_by_key
        RgMonStat v = c.mon[i]; int j = i;
        while (j > 0 && c.mon[j - 1].rarity > v.rarity) { c.mon[j] = c.mon[j - 1]; j--; }
        c.mon[j] = v;
    }
    // GameConfig::symbol_max (+1): max enemy tile, or 'A' decremented when there are none
    int mx = -1;
    for (int i = 0; i < c.n_enemies; i++) if (p->presets[i].tile - 'A' > mx) mx = p->presets[i].tile - 'A';
    c.symbols = mx < 0 ? 17 : mx + 17 + 1;
    rg_config_derive(&c);
    return "";
}

} // namespace

std::string rg_parse_config(const char *json, RgParsed *out) {
    set_defaults(out);
    if (!json) return finish(out);
    Parser ps{json, json + strlen(json), ""};
    JVal root;
    if (!ps.parse(root)) return ps.err;
    ps.ws();
    if (ps.p != ps.end) return "trailing characters";
    if (root.kind != JVal::Obj) return "invalid type: expected struct GameConfig";
    Ctx c;
    RgConfig &g = out->cfg;
    GET_I32(&root, "width", g.width);
    GET_I32(&root, "height", g.height);
    if (const JVal *v = root.get("seed")) {
        if (v->kind == JVal::Num && v->is_int && !v->neg) { out->has_seed = true; out->seed_lo = (uint64_t)v->mag; out->seed_hi = (uint64_t)(v->mag >> 64); }
        else if (v->kind != JVal::Null) return "invalid type for `seed`, expected u128";
    }
    if (const JVal *v = root.get("seed_range")) {
        if (v->kind == JVal::Arr && v->arr.size() == 2 && v->arr[0].kind == JVal::Num && v->arr[1].kind == JVal::Num) {
            out->has_seed_range = true; out->seed_range[0] = v->arr[0].mag; out->seed_range[1] = v->arr[1].mag;
        } else if (v->kind != JVal::Null) return "invalid type for `seed_range`, expected [u128; 2]";
    }
    if (const JVal *v = root.get("hide_dungeon")) {
        if (v->kind != JVal::Bool) return "invalid type for `hide_dungeon`, expected a boolean";
        g.hide_dungeon = v->b;
    }
    if (const JVal *d = root.get("dungeon")) {
        if (d->kind != JVal::Obj) return "invalid type for `dungeon`";
        const JVal *st = d->get("style");
        if (!st || st->kind != JVal::Str) return "missing field `style`";
        if (st->s != "rogue") return "unknown variant `" + st->s + "` (only the rogue dungeon style is implemented)";
        GET_I32(d, "room_num_x", g.room_num_x);
        GET_I32(d, "room_num_y", g.room_num_y);
        if (const JVal *m = d->get("min_room_size")) {
            if (m->kind != JVal::Obj) return "invalid type for `min_room_size`";
            GET_I32(m, "x", g.min_room_x);
            GET_I32(m, "y", g.min_room_y);
        }
        if (const JVal *t = d->get("enable_trap")) {  // read by serde, used by nothing: the engine has no trap code (rogue/mod.rs:33-35)
            if (t->kind != JVal::Bool) return "invalid type for `enable_trap`, expected a boolean";
            out->enable_trap = t->b;
        }
        GET_U32(d, "max_empty_rooms", g.max_empty_rooms);
        GET_U32(d, "amulet_level", g.amulet_level);
        GET_U32(d, "maze_rate_inv", g.maze_rate_inv);
        GET_U32(d, "dark_level", g.dark_level);
        GET_U32(d, "hidden_passage_rate_inv", g.hidden_passage_rate_inv);
        GET_U32(d, "locked_door_rate_inv", g.locked_door_rate_inv);
        GET_U32(d, "max_extra_edges", g.max_extra_edges);
        GET_U32(d, "door_unlock_rate_inv", g.door_unlock_rate_inv);
        GET_U32(d, "passage_unlock_rate_inv", g.passage_unlock_rate_inv);
    }
    if (const JVal *it = root.get("item")) {
        if (it->kind != JVal::Obj) return "invalid type for `item`";
        if (const JVal *gd = it->get("gold")) {
            if (gd->kind != JVal::Obj) return "invalid type for `gold`";
            GET_U32(gd, "rate_inv", g.gold_rate_inv);
            GET_U32(gd, "base", g.gold_base);
            GET_U32(gd, "per_level", g.gold_per_level);
            GET_U32(gd, "minimum", g.gold_minimum);
        }
        // item.armor / item.weapon: the tables the player's init_items are looked up in (rg_resolve_items below)
    }
    const JVal *pl = root.get("player");
    if (pl) {
        if (pl->kind != JVal::Obj) return "invalid type for `player`";
        GET_U32(pl, "hunger_time", g.hunger_time);
        GET_I32(pl, "init_hp", g.init_hp);
        {   // init_str / heal_threshold: deserialised, then never read (StatusInner::from_config uses Strength(16), player.rs:284; Player::heal
            // uses the literal 20, player.rs:226): accepted, type-checked, carried for dump_config
            int64_t t = out->init_str;
            if (!get_int(c, pl, "init_str", -0x7fffffffffffffffLL - 1, 0x7fffffffffffffffLL, &t)) return c.err;
            out->init_str = t;
            GET_U32(pl, "heal_threshold", out->heal_threshold);
        }
        if (const JVal *ex = pl->get("exps")) {
            out->exps_given = true;
            if (ex->kind != JVal::Arr || ex->arr.empty() || ex->arr.size() > 21) return "invalid `exps` (1..=21 entries supported)";
            g.n_level_exps = (int)ex->arr.size();
            for (size_t i = 0; i < ex->arr.size(); i++) {
                if (ex->arr[i].kind != JVal::Num || ex->arr[i].neg || ex->arr[i].mag > 0xffffffffu) return "invalid type in `exps`, expected u32";
                g.level_exps[i] = (uint32_t)ex->arr[i].mag;
            }
        }
    }
    if (const JVal *en = root.get("enemies")) {
        if (en->kind != JVal::Obj) return "invalid type for `enemies`";
        if (const JVal *l = en->get("enemies")) {
            if (l->kind != JVal::Arr) return "invalid type for `enemies.enemies`, expected a sequence";
            if (l->arr.size() > RG_MAX_ENEMY_KINDS + 6) return "too many enemy presets (at most 32)";
            out->enemies_given = true;
            out->n_presets = 0;
            for (auto &x : l->arr) {
                int slot = out->n_presets;
                if (x.kind == JVal::Num && x.is_int && !x.neg) {  // Preset::Builtin(usize)
                    if (x.mag > 25) return "enemy preset index out of range (builtin monsters are 0..=25)";
                    fill_builtin(out, slot, (int)x.mag);
                } else if (x.kind == JVal::Obj) {  // Preset::Custom(Status) (enemies.rs:110-121)
                    RgMonStat &m = out->presets[slot];
                    memset(&m, 0, sizeof m);
                    int64_t attr = 0, defense = 0, exp = 0, gold = 0, level = 0, tile = 0, rarity = 0;
                    for (const char *k : {"attack", "attr", "defense", "exp", "gold", "level", "name", "tile", "rarelity"})
                        if (!x.get(k)) return std::string("custom enemy status: missing field `") + k + "`";
                    if (!get_int(c, &x, "attr", 0, 0xffff, &attr) || !get_int(c, &x, "defense", -1000, 1000, &defense) ||
                        !get_int(c, &x, "exp", 0, 0xffffffffLL, &exp) || !get_int(c, &x, "gold", 0, 0xffffffffLL, &gold) ||
                        !get_int(c, &x, "level", 1, 1000, &level) || !get_int(c, &x, "tile", 'A', 'Z', &tile) ||
                        !get_int(c, &x, "rarelity", 0, 255, &rarity))
                        return "custom enemy status: " + c.err + " (tile must be 'A'..'Z' = 65..90)";
                    const JVal *nm = x.get("name");
                    if (nm->kind != JVal::Str) return "custom enemy status: invalid type for `name`";
                    const JVal *at = x.get("attack");
                    if (at->kind != JVal::Arr || at->arr.size() > 4) return "custom enemy status: `attack` must be a sequence of at most 4 dice";
                    for (size_t k = 0; k < at->arr.size(); k++) {
                        int64_t times = 0, mx = 0;
                        if (at->arr[k].kind != JVal::Obj || !at->arr[k].get("times") || !at->arr[k].get("max") ||
                            !get_int(c, &at->arr[k], "times", 0, 255, &times) || !get_int(c, &at->arr[k], "max", 0, 255, &mx))
                            return "custom enemy status: each attack die needs `times` and `max` in 0..=255";
                        if (times > 0 && mx < 1) return "custom enemy status: a die that is rolled needs max >= 1";
                        m.att[k][0] = (uint8_t)times; m.att[k][1] = (uint8_t)mx;
                    }
                    m.n_att = (uint8_t)at->arr.size();
                    m.attr = (uint16_t)attr; m.defense = (int32_t)defense; m.exp = (uint32_t)exp; m.level = (int16_t)level;
                    m.tile = (uint8_t)tile; m.rarity = (uint8_t)rarity;
                    out->preset_builtin[slot] = -1; out->preset_name[slot] = nm->s; out->preset_gold[slot] = (uint32_t)gold;
                } else return "invalid enemy preset: expected a builtin index or a status object";
                out->n_presets++;
            }
        }
        GET_U32(en, "appear_rate_gold", g.appear_rate_gold);
        GET_U32(en, "appear_rate_nogold", g.appear_rate_nogold);
    }
    if (const JVal *km = root.get("keymap")) {  // GameStateImpl::new replaces RunTime::keymap with KeyMap::ai (python/src/state_impls.rs:27,40): inert at this boundary
        if (km->kind != JVal::Obj) return "invalid type for `keymap`, expected a map";
        if (!keymap_is_default(*km)) out->keymap_json = rgjson::to_string(*km);  // to_json skips a default keymap (core/src/lib.rs:71-73)
    }
    {   // item.weapon / item.armor / player.init_items / player.max_items (rg_items.cpp)
        std::string e = rg_resolve_items(root.get("item"), pl, out);
        if (!e.empty()) return e;
    }
    return finish(out);
}

bool rg_config_equal(const RgParsed &a, const RgParsed &b) { return memcmp(&a.cfg, &b.cfg, sizeof(RgConfig)) == 0 && a.init_draws == b.init_draws; }

static std::string u128_str(uint64_t lo, uint64_t hi) { return rgjson::u128_str(((unsigned __int128)hi << 64) | lo); }

// GameConfig::to_json with skip_serializing_if = is_default (core/src/lib.rs:42-86): default-valued
// sections are omitted, so `json.loads(dump) == config_dict` holds for the reference's test configs.
std::string rg_dump_config_json(const RgParsed &p, uint64_t seed_lo, uint64_t seed_hi, bool has_seed) {
    RgParsed d; set_defaults(&d);
    const RgConfig &c = p.cfg, &z = d.cfg;
    std::string s = "{";
    auto sep = [&]() { if (s.size() > 1) s += ", "; };
    if (c.width != 80) { sep(); s += "\"width\": " + std::to_string(c.width); }
    if (c.height != 24) { sep(); s += "\"height\": " + std::to_string(c.height); }
    if (has_seed) { sep(); s += "\"seed\": " + u128_str(seed_lo, seed_hi); }
    if (p.has_seed_range) {  // kept whether or not a seed is set (it only takes effect without one, core/src/lib.rs:57-61)
        sep();
        s += "\"seed_range\": [" + u128_str((uint64_t)p.seed_range[0], (uint64_t)(p.seed_range[0] >> 64)) + ", " +
             u128_str((uint64_t)p.seed_range[1], (uint64_t)(p.seed_range[1] >> 64)) + "]";
    }
    bool dung_default = c.room_num_x == z.room_num_x && c.room_num_y == z.room_num_y && c.min_room_x == z.min_room_x && c.min_room_y == z.min_room_y &&
        c.max_empty_rooms == z.max_empty_rooms && c.amulet_level == z.amulet_level && c.maze_rate_inv == z.maze_rate_inv && c.dark_level == z.dark_level &&
        c.hidden_passage_rate_inv == z.hidden_passage_rate_inv && c.locked_door_rate_inv == z.locked_door_rate_inv && c.max_extra_edges == z.max_extra_edges &&
        c.door_unlock_rate_inv == z.door_unlock_rate_inv && c.passage_unlock_rate_inv == z.passage_unlock_rate_inv && p.enable_trap;
    if (!dung_default) {
        sep();
        s += "\"dungeon\": {\"style\": \"rogue\", \"room_num_x\": " + std::to_string(c.room_num_x) + ", \"room_num_y\": " + std::to_string(c.room_num_y) +
             ", \"min_room_size\": {\"x\": " + std::to_string(c.min_room_x) + ", \"y\": " + std::to_string(c.min_room_y) + "}, \"enable_trap\": " + (p.enable_trap ? "true" : "false") +
             ", \"max_empty_rooms\": " + std::to_string(c.max_empty_rooms) + ", \"amulet_level\": " + std::to_string(c.amulet_level) +
             ", \"maze_rate_inv\": " + std::to_string(c.maze_rate_inv) + ", \"dark_level\": " + std::to_string(c.dark_level) +
             ", \"hidden_passage_rate_inv\": " + std::to_string(c.hidden_passage_rate_inv) + ", \"locked_door_rate_inv\": " + std::to_string(c.locked_door_rate_inv) +
             ", \"max_extra_edges\": " + std::to_string(c.max_extra_edges) + ", \"door_unlock_rate_inv\": " + std::to_string(c.door_unlock_rate_inv) +
             ", \"passage_unlock_rate_inv\": " + std::to_string(c.passage_unlock_rate_inv) + "}";
    }
    // item::Config has three sections without skip rules (item/mod.rs:24-29): a non-default `item` is written whole
    if (c.gold_rate_inv != z.gold_rate_inv || c.gold_base != z.gold_base || c.gold_per_level != z.gold_per_level || c.gold_minimum != z.gold_minimum ||
        !p.weapon_default || !p.armor_default) {
        sep();
        s += "\"item\": {\"armor\": " + p.armor_json + ", \"gold\": {\"rate_inv\": " + std::to_string(c.gold_rate_inv) + ", \"base\": " + std::to_string(c.gold_base) +
             ", \"per_level\": " + std::to_string(c.gold_per_level) + ", \"minimum\": " + std::to_string(c.gold_minimum) + "}, \"weapon\": " + p.weapon_json + "}";
    }
    if (!p.keymap_json.empty()) { sep(); s += "\"keymap\": " + p.keymap_json; }
    // player::Config (player.rs:17-32): `exps` flattened first, then the fields in declaration order
    bool exps_default = c.n_level_exps == 21 && memcmp(c.level_exps, DEFAULT_EXPS, sizeof DEFAULT_EXPS) == 0;
    if (c.hunger_time != z.hunger_time || c.init_hp != z.init_hp || !exps_default || p.init_str != 16 || p.max_items != 27 || !p.init_items_default ||
        p.heal_threshold != 20) {
        sep();
        s += "\"player\": {\"exps\": [";
        for (int i = 0; i < c.n_level_exps; i++) s += (i ? ", " : "") + std::to_string(c.level_exps[i]);
        s += "], \"hunger_time\": " + std::to_string(c.hunger_time) + ", \"init_hp\": " + std::to_string(c.init_hp) + ", \"init_str\": " + std::to_string(p.init_str) +
             ", \"max_items\": " + std::to_string(p.max_items) + ", \"init_items\": " + p.init_items_json + ", \"heal_threshold\": " + std::to_string(p.heal_threshold) + "}";
    }
    bool en_default = p.n_presets == 26;
    for (int i = 0; en_default && i < 26; i++) en_default = p.preset_builtin[i] == i;
    if (!en_default || c.appear_rate_gold != 80 || c.appear_rate_nogold != 25) {
        sep();
        s += "\"enemies\": {\"enemies\": [";
        for (int i = 0; i < p.n_presets; i++) {
            if (i) s += ", ";
            if (p.preset_builtin[i] >= 0) { s += std::to_string(p.preset_builtin[i]); continue; }
            const RgMonStat &m = p.presets[i];
            s += "{\"attack\": [";
            for (int k = 0; k < m.n_att; k++) s += std::string(k ? ", " : "") + "{\"times\": " + std::to_string(m.att[k][0]) + ", \"max\": " + std::to_string(m.att[k][1]) + "}";
            s += "], \"attr\": " + std::to_string(m.attr) + ", \"defense\": " + std::to_string(m.defense) + ", \"exp\": " + std::to_string(m.exp) +
                 ", \"gold\": " + std::to_string(p.preset_gold[i]) + ", \"level\": " + std::to_string(m.level) + ", \"name\": \"" + p.preset_name[i] +
                 "\", \"tile\": " + std::to_string(m.tile) + ", \"rarelity\": " + std::to_string(m.rarity) + "}";
        }
        s += "]";
        if (c.appear_rate_gold != 80) s += ", \"appear_rate_gold\": " + std::to_string(c.appear_rate_gold);
        if (c.appear_rate_nogold != 25) s += ", \"appear_rate_nogold\": " + std::to_string(c.appear_rate_nogold);
        s += "}";
    }
    sep(); s += std::string("\"hide_dungeon\": ") + (c.hide_dungeon ? "true" : "false");
    s += "}";
    return s;
}

// ---------------------------------------------------------------------------------------------
// The config surface, key by key.  "honoured": the stepper's results depend on it exactly as the engine's do.  "inert": serde reads it, the
// engine (as driven through python/src) never does -- accepted, type-checked, written back by dump_config, no effect, like the reference.
// There is no third state: a key the engine reads and the stepper does not implement would have to be a creation error
// (tests/test_config_schema.py walks this table against the key lists of the reference's structs).
// ---------------------------------------------------------------------------------------------
std::string rg_config_schema_json() {
    static const struct { const char *path, *status, *why; } K[] = {
        {"width", "honoured", "core/src/lib.rs:45-48,166-184"}, {"height", "honoured", "core/src/lib.rs:49-52"},
        {"seed", "honoured", "core/src/lib.rs:53-57,157-165"}, {"seed_range", "honoured", "core/src/lib.rs:58-62; only without a seed"},
        {"hide_dungeon", "honoured", "core/src/lib.rs:83-85; rogue/mod.rs:465-475"},
        {"keymap", "inert", "GameStateImpl::new overwrites RunTime::keymap with KeyMap::ai (python/src/state_impls.rs:27,40)"},
        {"dungeon.style", "honoured", "dungeon/mod.rs:16-29: only `rogue` builds; the other variants are unimplemented!() there and an error here"},
        {"dungeon.room_num_x", "honoured", "rogue/mod.rs:25-27"}, {"dungeon.room_num_y", "honoured", "rogue/mod.rs:28-30"},
        {"dungeon.min_room_size", "honoured", "rogue/mod.rs:31-33"},
        {"dungeon.enable_trap", "inert", "rogue/mod.rs:34-36: no trap code exists in the engine"},
        {"dungeon.max_empty_rooms", "honoured", "rooms.rs:179"}, {"dungeon.amulet_level", "honoured", "enemies.rs:275-285"},
        {"dungeon.maze_rate_inv", "honoured", "rooms.rs:238"}, {"dungeon.dark_level", "honoured", "rooms.rs:237; floor.rs:430,437"},
        {"dungeon.hidden_passage_rate_inv", "honoured", "floor.rs:431"}, {"dungeon.locked_door_rate_inv", "honoured", "floor.rs:438"},
        {"dungeon.max_extra_edges", "honoured", "passages.rs:54"}, {"dungeon.door_unlock_rate_inv", "honoured", "floor.rs:363"},
        {"dungeon.passage_unlock_rate_inv", "honoured", "floor.rs:359"},
        {"item.gold.rate_inv", "honoured", "gold.rs:19"}, {"item.gold.base", "honoured", "gold.rs:22"}, {"item.gold.per_level", "honoured", "gold.rs:22"},
        {"item.gold.minimum", "honoured", "gold.rs:22"},
        {"item.weapon.weapons", "honoured", "weapon.rs:34-47: the table InitItem::Weapon names are looked up in"},
        {"item.weapon.weapons[].at_weild", "honoured", "fight.rs:27-33"}, {"item.weapon.weapons[].name", "honoured", "item/mod.rs:189-191"},
        {"item.weapon.weapons[].init_num", "honoured", "weapon.rs:159: one item-stream draw per InitItem::Weapon"},
        {"item.weapon.weapons[].at_throw", "inert", "throwing is unreachable from the 19-key ai keymap (input.rs:73-100)"},
        {"item.weapon.weapons[].attr", "inert", "IS_MANY / CAN_THROW of a weapon: read by inventory code only"},
        {"item.weapon.weapons[].is_initial", "inert", "read by nothing"}, {"item.weapon.weapons[].appear_rate", "inert", "Handler::gen_item has no caller (handler.rs:42-53)"},
        {"item.weapon.weapons[].worth", "inert", "read by nothing"}, {"item.weapon.weapons[].launcher", "inert", "only for thrown weapons (fight.rs:12-18)"},
        {"item.weapon.cursed_rate", "inert", "Handler::gen_item has no caller (handler.rs:42-53)"}, {"item.weapon.powerup_rate", "inert", "Handler::gen_item has no caller"},
        {"item.armor.armors", "honoured", "armor.rs:46-60: the table InitItem::Armor names are looked up in"},
        {"item.armor.armors[].name", "honoured", "item/mod.rs:204-206"}, {"item.armor.armors[].def", "honoured", "armor.rs:100-102; fight.rs:80-82"},
        {"item.armor.armors[].appear_rate", "inert", "Handler::gen_item has no caller"}, {"item.armor.armors[].worth", "inert", "read by nothing"},
        {"item.armor.cursed_rate", "inert", "Handler::gen_item has no caller"}, {"item.armor.powerup_rate", "inert", "Handler::gen_item has no caller"},
        {"player.exps", "honoured", "player.rs:308-353"}, {"player.hunger_time", "honoured", "player.rs:107-118,163-176,286"},
        {"player.init_hp", "honoured", "player.rs:283"},
        {"player.init_str", "inert", "StatusInner::from_config hard-codes Strength(16) (player.rs:284)"},
        {"player.max_items", "honoured", "ItemBox capacity: player.rs:83; itembox.rs:21-40"},
        {"player.init_items", "honoured", "player.rs:136-153; item/mod.rs:181-221"},
        {"player.init_items[].Weapon.name", "honoured", "item/mod.rs:189-191; player.rs:198-205"},
        {"player.init_items[].Weapon.num_plus", "inert", "the weapon count is read by nothing reachable"},
        {"player.init_items[].Weapon.hit_plus", "honoured", "item/mod.rs:195; fight.rs:21-22"}, {"player.init_items[].Weapon.dam_plus", "honoured", "item/mod.rs:196; fight.rs:23"},
        {"player.init_items[].Armor.name", "honoured", "item/mod.rs:204-206; player.rs:206-213"}, {"player.init_items[].Armor.def_plus", "honoured", "item/mod.rs:208"},
        {"player.init_items[].Noinit.kind", "honoured", "Gold: core/src/lib.rs:348-353; Weapon / Armor: candidates of equip_from_box (player.rs:214-220)"},
        {"player.init_items[].Noinit.how_many", "honoured", "the initial gold count (core/src/lib.rs:348-353)"},
        {"player.init_items[].Noinit.attr", "inert", "dungeon gold is always `many` (item/mod.rs:409), so a merge never consults the pack item's attr"},
        {"player.heal_threshold", "inert", "Player::heal uses the literal 20 (player.rs:226)"},
        {"enemies.enemies", "honoured", "enemies.rs:20-21,250-261"}, {"enemies.enemies[].attack", "honoured", "fight.rs:41-50"},
        {"enemies.enemies[].attr", "honoured", "enemies.rs:126-139"}, {"enemies.enemies[].defense", "honoured", "fight.rs:74-78"},
        {"enemies.enemies[].exp", "honoured", "enemies.rs:308"}, {"enemies.enemies[].level", "honoured", "enemies.rs:303"},
        {"enemies.enemies[].tile", "honoured", "core/src/lib.rs:150-155"}, {"enemies.enemies[].rarelity", "honoured", "enemies.rs:250-261"},
        {"enemies.enemies[].gold", "inert", "copied into Enemy (enemies.rs:450), dropped by nothing"}, {"enemies.enemies[].name", "inert", "only inside GameMsg texts"},
        {"enemies.appear_rate_gold", "honoured", "enemies.rs:297"}, {"enemies.appear_rate_nogold", "honoured", "enemies.rs:297"},
    };
    std::string s = "[";
    for (size_t i = 0; i < sizeof K / sizeof K[0]; i++)
        s += std::string(i ? ", " : "") + "{\"path\": " + rgjson::quote(K[i].path) + ", \"status\": \"" + K[i].status + "\", \"why\": " + rgjson::quote(K[i].why) + "}";
    return s + "]";
}
