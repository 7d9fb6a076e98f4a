// rg_config.cpp -- minimal JSON reader for GameConfig (core/src/lib.rs:42-86 and the per-module
// Config structs: rogue/mod.rs:23-134, enemies.rs:18-85, player.rs:17-66, item/gold.rs:6-52).
// Like serde without deny_unknown_fields, unknown keys are ignored; wrong types are errors.
#include "rg_config.h"

#include <cctype>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

namespace {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool neg = false, is_int = true;
    unsigned __int128 mag = 0; // integer magnitude
    double d = 0;
    std::string s;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct Parser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    bool parse(JVal &v) {
        ws();
        if (p >= end) return fail("EOF while parsing a value");
        char c = *p;
        if (c == '{') {
            v.kind = JVal::Obj; p++; ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                JVal k;
                if (p >= end || *p != '"') return fail("key must be a string");
                if (!parse(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected `:`");
                p++;
                JVal x;
                if (!parse(x)) return false;
                v.obj.emplace_back(k.s, std::move(x));
                ws();
                if (p < end && *p == ',') { p++; ws(); if (p < end && *p == '}') return fail("trailing comma"); continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected `,` or `}`");
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr; p++; ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                JVal x;
                if (!parse(x)) return false;
                v.arr.push_back(std::move(x));
                ws();
                if (p < end && *p == ',') { p++; ws(); if (p < end && *p == ']') return fail("trailing comma"); continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected `,` or `]`");
            }
        }
        if (c == '"') {
            v.kind = JVal::Str; p++;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) { p++; char e = *p++; v.s += (e == 'n' ? '\n' : e == 't' ? '\t' : e); }
                else v.s += *p++;
            }
            if (p >= end) return fail("EOF while parsing a string");
            p++;
            return true;
        }
        if (!strncmp(p, "true", 4) && end - p >= 4) { v.kind = JVal::Bool; v.b = true; p += 4; return true; }
        if (!strncmp(p, "false", 5) && end - p >= 5) { v.kind = JVal::Bool; v.b = false; p += 5; return true; }
        if (!strncmp(p, "null", 4) && end - p >= 4) { v.kind = JVal::Null; p += 4; return true; }
        if (c == '-' || isdigit((unsigned char)c)) {
            v.kind = JVal::Num;
            const char *s = p;
            if (c == '-') { v.neg = true; p++; }
            if (p >= end || !isdigit((unsigned char)*p)) return fail("invalid number");
            while (p < end && isdigit((unsigned char)*p)) { v.mag = v.mag * 10 + (unsigned)(*p - '0'); p++; }
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {
                v.is_int = false;
                while (p < end && (isdigit((unsigned char)*p) || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) p++;
                v.d = strtod(std::string(s, p).c_str(), nullptr);
            }
            return true;
        }
        return fail(std::string("expected value, found `") + c + "`");
    }
};

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

// rarity of BUILTIN_ENEMIES[i] (enemies.rs:474-761), tile = 'A' + i
const uint8_t BUILTIN_RARITY[26] = {12, 2, 10, 25, 1, 15, 23, 4, 5, 24, 0, 9, 21, 13, 7, 18, 11, 6, 3, 16, 20, 22, 17, 19, 14, 8};
const uint32_t DEFAULT_EXPS[21] = {10, 20, 40, 80, 160, 320, 640, 1300, 2600, 5200, 13000, 26000, 50000, 100000,
                                   200000, 400000, 800000, 2000000, 4000000, 8000000, 0xFFFFFFFFu};

void set_defaults(RgParsed *p) {
    memset(p, 0, sizeof *p);
    RgConfig &c = p->cfg;
    c.width = 80; c.height = 24; c.hide_dungeon = 1;
    c.room_num_x = 3; c.room_num_y = 3; c.min_room_x = 4; c.min_room_y = 4;
    c.max_empty_rooms = 3; c.amulet_level = 25; c.maze_rate_inv = 15; c.dark_level = 10;
    c.hidden_passage_rate_inv = 40; c.locked_door_rate_inv = 5; c.max_extra_edges = 5;
    c.door_unlock_rate_inv = 5; c.passage_unlock_rate_inv = 3;
    c.gold_rate_inv = 2; c.gold_base = 50; c.gold_per_level = 10; c.gold_minimum = 2;
    c.hunger_time = 1300; c.init_hp = 12;
    c.appear_rate_gold = 80; c.appear_rate_nogold = 25;
    p->n_enemy_ids = 26;
    for (int i = 0; i < 26; i++) p->enemy_ids[i] = i;
    memcpy(c.level_exps, DEFAULT_EXPS, sizeof DEFAULT_EXPS);
    c.n_level_exps = 21;
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
    c.n_enemies = p->n_enemy_ids;
    for (int i = 0; i < c.n_enemies; i++) c.enemy_sorted[i] = (uint8_t)p->enemy_ids[i];
    for (int i = 1; i < c.n_enemies; i++) {
        uint8_t v = c.enemy_sorted[i]; int j = i;
        while (j > 0 && BUILTIN_RARITY[c.enemy_sorted[j - 1]] > BUILTIN_RARITY[v]) { c.enemy_sorted[j] = c.enemy_sorted[j - 1]; j--; }
        c.enemy_sorted[j] = v;
    }
    // GameConfig::symbol_max (+1): max enemy tile, or 'A' decremented when there are none
    int mx = -1;
    for (int i = 0; i < c.n_enemies; i++) if (p->enemy_ids[i] > mx) mx = p->enemy_ids[i];
    c.symbols = mx < 0 ? 17 : mx + 17 + 1;
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
        // armor / weapon tables only matter for item drops, which the engine never generates
    }
    if (const JVal *pl = root.get("player")) {
        if (pl->kind != JVal::Obj) return "invalid type for `player`";
        GET_U32(pl, "hunger_time", g.hunger_time);
        GET_I32(pl, "init_hp", g.init_hp);
        if (const JVal *ex = pl->get("exps")) {
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
            if (l->arr.size() > RG_MAX_ENEMY_KINDS) return "too many enemy presets";
            out->enemies_given = true;
            out->n_enemy_ids = 0;
            for (auto &x : l->arr) {
                if (x.kind != JVal::Num || !x.is_int || x.neg || x.mag > 25)
                    return "enemy presets must be builtin indices 0..=25 (custom monster statuses are not supported by the HIP stepper yet)";
                out->enemy_ids[out->n_enemy_ids++] = (int)x.mag;
            }
        }
        GET_U32(en, "appear_rate_gold", g.appear_rate_gold);
        GET_U32(en, "appear_rate_nogold", g.appear_rate_nogold);
    }
    return finish(out);
}

bool rg_config_equal(const RgConfig &a, const RgConfig &b) { return memcmp(&a, &b, sizeof(RgConfig)) == 0; }

static std::string u128_str(uint64_t lo, uint64_t hi) {
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    if (v == 0) return "0";
    std::string s;
    while (v) { s.insert(s.begin(), (char)('0' + (int)(v % 10))); v /= 10; }
    return s;
}

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
    bool dung_default = c.room_num_x == z.room_num_x && c.room_num_y == z.room_num_y && c.min_room_x == z.min_room_x && c.min_room_y == z.min_room_y &&
        c.max_empty_rooms == z.max_empty_rooms && c.amulet_level == z.amulet_level && c.maze_rate_inv == z.maze_rate_inv && c.dark_level == z.dark_level &&
        c.hidden_passage_rate_inv == z.hidden_passage_rate_inv && c.locked_door_rate_inv == z.locked_door_rate_inv && c.max_extra_edges == z.max_extra_edges &&
        c.door_unlock_rate_inv == z.door_unlock_rate_inv && c.passage_unlock_rate_inv == z.passage_unlock_rate_inv;
    if (!dung_default) {
        sep();
        s += "\"dungeon\": {\"style\": \"rogue\", \"room_num_x\": " + std::to_string(c.room_num_x) + ", \"room_num_y\": " + std::to_string(c.room_num_y) +
             ", \"min_room_size\": {\"x\": " + std::to_string(c.min_room_x) + ", \"y\": " + std::to_string(c.min_room_y) + "}, \"enable_trap\": true" +
             ", \"max_empty_rooms\": " + std::to_string(c.max_empty_rooms) + ", \"amulet_level\": " + std::to_string(c.amulet_level) +
             ", \"maze_rate_inv\": " + std::to_string(c.maze_rate_inv) + ", \"dark_level\": " + std::to_string(c.dark_level) +
             ", \"hidden_passage_rate_inv\": " + std::to_string(c.hidden_passage_rate_inv) + ", \"locked_door_rate_inv\": " + std::to_string(c.locked_door_rate_inv) +
             ", \"max_extra_edges\": " + std::to_string(c.max_extra_edges) + ", \"door_unlock_rate_inv\": " + std::to_string(c.door_unlock_rate_inv) +
             ", \"passage_unlock_rate_inv\": " + std::to_string(c.passage_unlock_rate_inv) + "}";
    }
    if (c.gold_rate_inv != z.gold_rate_inv || c.gold_base != z.gold_base || c.gold_per_level != z.gold_per_level || c.gold_minimum != z.gold_minimum) {
        sep();
        s += "\"item\": {\"gold\": {\"rate_inv\": " + std::to_string(c.gold_rate_inv) + ", \"base\": " + std::to_string(c.gold_base) +
             ", \"per_level\": " + std::to_string(c.gold_per_level) + ", \"minimum\": " + std::to_string(c.gold_minimum) + "}}";
    }
    if (c.hunger_time != z.hunger_time || c.init_hp != z.init_hp) {
        sep();
        s += "\"player\": {\"hunger_time\": " + std::to_string(c.hunger_time) + ", \"init_hp\": " + std::to_string(c.init_hp) + "}";
    }
    bool en_default = p.n_enemy_ids == 26;
    for (int i = 0; en_default && i < 26; i++) en_default = p.enemy_ids[i] == i;
    if (!en_default || c.appear_rate_gold != 80 || c.appear_rate_nogold != 25) {
        sep();
        s += "\"enemies\": {\"enemies\": [";
        for (int i = 0; i < p.n_enemy_ids; i++) { if (i) s += ", "; s += std::to_string(p.enemy_ids[i]); }
        s += "]";
        if (c.appear_rate_gold != 80) s += ", \"appear_rate_gold\": " + std::to_string(c.appear_rate_gold);
        if (c.appear_rate_nogold != 25) s += ", \"appear_rate_nogold\": " + std::to_string(c.appear_rate_nogold);
        s += "}";
    }
    sep(); s += std::string("\"hide_dungeon\": ") + (c.hide_dungeon ? "true" : "false");
    s += "}";
    return s;
}
