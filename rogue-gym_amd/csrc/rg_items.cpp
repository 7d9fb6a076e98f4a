// rg_items.cpp -- item::Config (weapon / armor tables) and the player's initial pack, resolved on the host.
//
// In the reference these are run-time objects: `ItemHandler` holds a `Handler<WeaponStatus>` / `Handler<ArmorStatus>` (item/mod.rs:378-397,
// handler.rs:34-63), `Player::init_items` turns every `InitItem` of `player.init_items` into an `Item` in the `ItemBox`
// (player.rs:136-153 -> item/mod.rs:411-422 -> InitItem::initialize, item/mod.rs:181-221) and equips the first weapon / armor named by an
// `InitItem::Weapon` / `InitItem::Armor` (player.rs:198-221).  Nothing of that changes after `GameConfig::build` except the pack's gold
// count -- only gold is ever placed in the dungeon (item/mod.rs:407-410) and the 19-key `ai` keymap has no wield / wear / drop -- so the
// stepper needs, per config:
//   * the (lo, hi) of every `rng.range(init_num)` the build draws on the ITEM stream, in list order (weapon.rs:148-170; armor draws nothing,
//     armor.rs:157-159; a Noinit item draws nothing, item/mod.rs:184),
//   * the equipped weapon's at_weild dice / hit_plus / dam_plus, or bare hands 1d4 +0 +0 (fight.rs:21-33),
//   * the equipped armor's def + def_plus, or 0 (player.rs:125-132),
//   * the pack's initial gold count and whether gold can be picked up at all (itembox.rs:30-40, core/src/lib.rs:348-353).
// Errors the reference raises at build time are raised at creation here, with its texts.
#include <cstring>

#include "rg_config.h"
#include "rg_json.h"

using rgjson::JVal;
using rgjson::quote;

namespace {

struct Dice { uint64_t times; int64_t max; };
struct WStat {  // WeaponStatus (weapon.rs:129-140)
    std::string name; Dice wield, thrw; uint32_t init_lo, init_hi, attr; bool is_initial; uint32_t appear_rate, worth;
    bool has_launcher; std::string launcher; int builtin;
};
struct AStat { std::string name; uint32_t appear_rate, worth; int32_t def; int builtin; };  // ArmorStatus (armor.rs:133-139)

// BUILTIN_WEAPONS (weapon.rs:198-298): wield dice, throw dice, name, attr, init_num, is_initial, launcher
const struct { int wt, wm, tt, tm; const char *name; int attr; int lo, hi; bool initial; const char *launcher; } BW[9] = {
    {2, 4, 1, 3, "mace", 0, 1, 2, true, nullptr},           {3, 4, 1, 2, "long-sword", 0, 1, 2, false, nullptr},
    {1, 1, 1, 1, "bow", 0, 1, 2, true, nullptr},            {1, 1, 2, 3, "arrow", 4 | 2, 8, 17, true, "bow"},
    {1, 6, 1, 4, "dagger", 2, 2, 7, false, nullptr},        {4, 4, 1, 2, "two-handed-sword", 0, 1, 2, false, nullptr},
    {1, 1, 1, 3, "dart", 4 | 2, 8, 17, false, nullptr},     {1, 2, 2, 4, "shuriken", 4 | 2, 8, 17, false, nullptr},
    {2, 3, 1, 6, "spear", 4, 8, 17, false, nullptr},
};
// BUILTIN_ARMORS (armor.rs:170-219)
const struct { const char *name; int rate, worth, def; } BA[8] = {
    {"leather armor", 20, 20, 2}, {"ring mail", 15, 25, 3},   {"studded leather armor", 15, 20, 3}, {"scale mail", 13, 30, 4},
    {"chain mail", 12, 75, 5},    {"splint mail", 10, 80, 6}, {"banded mail", 10, 90, 6},           {"plate mail", 5, 150, 7},
};
WStat builtin_weapon(int i) {
    WStat w;
    w.name = BW[i].name; w.wield = {(uint64_t)BW[i].wt, BW[i].wm}; w.thrw = {(uint64_t)BW[i].tt, BW[i].tm};
    w.init_lo = BW[i].lo; w.init_hi = BW[i].hi; w.attr = BW[i].attr; w.is_initial = BW[i].initial; w.appear_rate = 11; w.worth = 8;
    w.has_launcher = BW[i].launcher != nullptr; w.launcher = BW[i].launcher ? BW[i].launcher : ""; w.builtin = i;
    return w;
}
AStat builtin_armor(int i) { return AStat{BA[i].name, (uint32_t)BA[i].rate, (uint32_t)BA[i].worth, BA[i].def, i}; }

// ---- typed field readers: serde semantics (a missing field of a struct without #[serde(default)] is an error) ----
struct Rd {
    std::string err;
    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    bool integer(const JVal *o, const char *ctx, const char *key, int64_t lo, int64_t hi, int64_t *out) {
        const JVal *v = o->get(key);
        if (!v) return fail(std::string(ctx) + ": missing field `" + key + "`");
        if (v->kind != JVal::Num || !v->is_int) return fail(std::string(ctx) + ": invalid type for `" + key + "`, expected an integer");
        if (v->mag > (unsigned __int128)0x7fffffffffffffffLL) return fail(std::string(ctx) + ": `" + key + "` out of range");
        int64_t x = v->neg ? -(int64_t)v->mag : (int64_t)v->mag;
        if (x < lo || x > hi) return fail(std::string(ctx) + ": `" + key + "` out of range");
        *out = x;
        return true;
    }
    bool u32(const JVal *o, const char *ctx, const char *key, uint32_t *out) { int64_t t; if (!integer(o, ctx, key, 0, 0xffffffffLL, &t)) return false; *out = (uint32_t)t; return true; }
    bool i32(const JVal *o, const char *ctx, const char *key, int32_t *out) { int64_t t; if (!integer(o, ctx, key, -0x80000000LL, 0x7fffffffLL, &t)) return false; *out = (int32_t)t; return true; }
    bool str(const JVal *o, const char *ctx, const char *key, std::string *out) {
        const JVal *v = o->get(key);
        if (!v) return fail(std::string(ctx) + ": missing field `" + key + "`");
        if (v->kind != JVal::Str) return fail(std::string(ctx) + ": invalid type for `" + key + "`, expected a small string");
        *out = v->s;
        return true;
    }
    bool dice(const JVal *o, const char *ctx, const char *key, Dice *out) {  // Dice<HitPoint> {times: usize, max: HitPoint(i64)} (character/mod.rs:200-204)
        const JVal *v = o->get(key);
        if (!v) return fail(std::string(ctx) + ": missing field `" + key + "`");
        if (v->kind != JVal::Obj) return fail(std::string(ctx) + ": invalid type for `" + key + "`, expected struct Dice");
        int64_t t, m;
        if (!integer(v, key, "times", 0, 0x7fffffffffffffffLL, &t) || !integer(v, key, "max", -0x7fffffffffffffffLL - 1, 0x7fffffffffffffffLL, &m)) return false;
        *out = Dice{(uint64_t)t, m};
        return true;
    }
};
std::string dice_json(const Dice &d) { return "{\"times\": " + std::to_string(d.times) + ", \"max\": " + std::to_string(d.max) + "}"; }

// optional Parcent field of weapon::Config / armor::Config (cursed_rate, powerup_rate): never read by the engine -- Handler::gen_item, the only
// consumer (handler.rs:42-53), has no caller -- but serde reads it and to_json writes it back unless default
bool opt_u32(Rd &r, const JVal *o, const char *ctx, const char *key, uint32_t *out) { return !o->get(key) || r.u32(o, ctx, key, out); }

std::string weapon_stat_json(const WStat &w) {  // field order of the struct (serde writes fields in declaration order)
    return "{\"at_weild\": " + dice_json(w.wield) + ", \"at_throw\": " + dice_json(w.thrw) + ", \"name\": " + quote(w.name) + ", \"init_num\": {\"start\": " +
           std::to_string(w.init_lo) + ", \"end\": " + std::to_string(w.init_hi) + "}, \"attr\": " + std::to_string(w.attr) + ", \"is_initial\": " +
           (w.is_initial ? "true" : "false") + ", \"appear_rate\": " + std::to_string(w.appear_rate) + ", \"worth\": " + std::to_string(w.worth) +
           ", \"launcher\": " + (w.has_launcher ? quote(w.launcher) : std::string("null")) + "}";
}
std::string armor_stat_json(const AStat &a) {
    return "{\"name\": " + quote(a.name) + ", \"appear_rate\": " + std::to_string(a.appear_rate) + ", \"worth\": " + std::to_string(a.worth) + ", \"def\": " +
           std::to_string(a.def) + "}";
}

// one entry of the pack after init_player_items
struct PackItem {
    enum Kind { Armor, Food, Gold, Potion, Ring, Scroll, Wand, Weapon } kind;
    std::string name;          // Armor / Weapon
    uint32_t how_many;
    Dice wield;                // Weapon
    int64_t hit_plus, dam_plus;
    int64_t def;               // Armor: def + def_plus (Armor::def, armor.rs:100-102; Defense is i32, the sum wraps there and is range-checked here)
};

}  // namespace

std::string rg_resolve_items(const JVal *item, const JVal *player, RgParsed *out) {
    Rd r;
    RgConfig &g = out->cfg;
    // ---- item.weapon (weapon::Config, weapon.rs:12-22): default = the nine builtin presets in order ----
    std::vector<WStat> weapons;
    uint32_t w_cursed = 10, w_power = 5, a_cursed = 20, a_power = 8;  // weapon.rs:50-56, armor.rs:38-44
    bool weapons_given = false, armors_given = false;
    const JVal *wc = nullptr, *ac = nullptr;
    if (item) {
        wc = item->get("weapon"); ac = item->get("armor");
        // item::Config has no #[serde(default)] on its fields (item/mod.rs:24-29): a given `item` must name all three sections
        if (!ac) return "missing field `armor`";
        if (!item->get("gold")) return "missing field `gold`";
        if (!wc) return "missing field `weapon`";
        if (wc->kind != JVal::Obj) return "invalid type for `item.weapon`, expected struct Config";
        if (ac->kind != JVal::Obj) return "invalid type for `item.armor`, expected struct Config";
    }
    if (wc) {
        if (const JVal *l = wc->get("weapons")) {
            if (l->kind != JVal::Arr) return "invalid type for `item.weapon.weapons`, expected a sequence";
            weapons_given = true;
            for (const JVal &x : l->arr) {
                if (x.kind == JVal::Num && x.is_int && !x.neg) {  // Preset::Builtin(usize): BUILTIN_WEAPONS[i] (weapon.rs:74-81; out of range panics there)
                    if (x.mag >= 9) return "Invalid Setting: weapon preset index out of range (builtin weapons are 0..=8)";
                    weapons.push_back(builtin_weapon((int)x.mag));
                } else if (x.kind == JVal::Obj) {  // Preset::Custom(WeaponStatus)
                    WStat w; w.builtin = -1;
                    const char *ctx = "custom weapon status";
                    int64_t lo, hi, attr;
                    const JVal *rg = x.get("init_num");
                    if (!rg) return std::string(ctx) + ": missing field `init_num`";
                    if (rg->kind != JVal::Obj) return std::string(ctx) + ": invalid type for `init_num`, expected struct Range";
                    if (!r.dice(&x, ctx, "at_weild", &w.wield) || !r.dice(&x, ctx, "at_throw", &w.thrw) || !r.str(&x, ctx, "name", &w.name) ||
                        !r.integer(rg, "init_num", "start", 0, 0xffffffffLL, &lo) || !r.integer(rg, "init_num", "end", 0, 0xffffffffLL, &hi) ||
                        !r.integer(&x, ctx, "attr", 0, 255, &attr) || !r.u32(&x, ctx, "appear_rate", &w.appear_rate) || !r.u32(&x, ctx, "worth", &w.worth))
                        return r.err;
                    const JVal *ini = x.get("is_initial");
                    if (!ini) return std::string(ctx) + ": missing field `is_initial`";
                    if (ini->kind != JVal::Bool) return std::string(ctx) + ": invalid type for `is_initial`, expected a boolean";
                    const JVal *la = x.get("launcher");  // Option<SmallStr>: missing = None
                    if (la && la->kind != JVal::Null && la->kind != JVal::Str) return std::string(ctx) + ": invalid type for `launcher`, expected a string or null";
                    w.init_lo = (uint32_t)lo; w.init_hi = (uint32_t)hi; w.attr = (uint32_t)attr; w.is_initial = ini->b;
                    w.has_launcher = la && la->kind == JVal::Str; w.launcher = w.has_launcher ? la->s : "";
                    weapons.push_back(w);
                } else return "invalid weapon preset: expected a builtin index or a status object";
            }
        }
        if (!opt_u32(r, wc, "item.weapon", "cursed_rate", &w_cursed) || !opt_u32(r, wc, "item.weapon", "powerup_rate", &w_power)) return r.err;
    }
    if (!weapons_given) for (int i = 0; i < 9; i++) weapons.push_back(builtin_weapon(i));
    // ---- item.armor (armor::Config, armor.rs:10-21) ----
    std::vector<AStat> armors;
    if (ac) {
        if (const JVal *l = ac->get("armors")) {
            if (l->kind != JVal::Arr) return "invalid type for `item.armor.armors`, expected a sequence";
            armors_given = true;
            for (const JVal &x : l->arr) {
                if (x.kind == JVal::Num && x.is_int && !x.neg) {
                    if (x.mag >= 8) return "Invalid Setting: armor preset index out of range (builtin armors are 0..=7)";
                    armors.push_back(builtin_armor((int)x.mag));
                } else if (x.kind == JVal::Obj) {
                    AStat a; a.builtin = -1;
                    const char *ctx = "custom armor status";
                    if (!r.str(&x, ctx, "name", &a.name) || !r.u32(&x, ctx, "appear_rate", &a.appear_rate) || !r.u32(&x, ctx, "worth", &a.worth) ||
                        !r.i32(&x, ctx, "def", &a.def))
                        return r.err;
                    armors.push_back(a);
                } else return "invalid armor preset: expected a builtin index or a status object";
            }
        }
        if (!opt_u32(r, ac, "item.armor", "cursed_rate", &a_cursed) || !opt_u32(r, ac, "item.armor", "powerup_rate", &a_power)) return r.err;
    }
    if (!armors_given) for (int i = 0; i < 8; i++) armors.push_back(builtin_armor(i));
    // canonical texts (Config field order; the two rates are skipped when default, weapon.rs:15-21 / armor.rs:14-20)
    auto presets_json = [](auto &list, auto stat_json) {
        std::string s = "[";
        for (size_t i = 0; i < list.size(); i++) s += (i ? ", " : "") + (list[i].builtin >= 0 ? std::to_string(list[i].builtin) : stat_json(list[i]));
        return s + "]";
    };
    out->weapon_json = "{\"weapons\": " + presets_json(weapons, weapon_stat_json) + (w_cursed != 10 ? ", \"cursed_rate\": " + std::to_string(w_cursed) : "") +
                       (w_power != 5 ? ", \"powerup_rate\": " + std::to_string(w_power) : "") + "}";
    out->armor_json = "{\"armors\": " + presets_json(armors, armor_stat_json) + (a_cursed != 20 ? ", \"cursed_rate\": " + std::to_string(a_cursed) : "") +
                      (a_power != 8 ? ", \"powerup_rate\": " + std::to_string(a_power) : "") + "}";
    out->weapon_default = out->weapon_json == "{\"weapons\": [0, 1, 2, 3, 4, 5, 6, 7, 8]}";
    out->armor_default = out->armor_json == "{\"armors\": [0, 1, 2, 3, 4, 5, 6, 7]}";

    // ---- player.max_items / player.init_items (player.rs:26-29; default_init_items, player.rs:68-75) ----
    out->max_items = 27;
    const JVal *list = nullptr;
    if (player) {
        int64_t mi = 27;
        if (player->get("max_items") && !r.integer(player, "player", "max_items", 0, 0x7fffffffffffffffLL, &mi)) return r.err;
        out->max_items = (uint64_t)mi;
        list = player->get("init_items");
        if (list && list->kind != JVal::Arr) return "invalid type for `player.init_items`, expected a sequence";
    }
    static const char *DEFAULT_INIT_ITEMS =
        "[{\"Noinit\": {\"kind\": \"Gold\", \"how_many\": 0, \"attr\": 4}}, {\"Noinit\": {\"kind\": {\"Food\": \"Ration\"}, \"how_many\": 1, \"attr\": 4}}, "
        "{\"Armor\": {\"name\": \"ring mail\", \"def_plus\": 1}}, {\"Weapon\": {\"name\": \"mace\", \"num_plus\": 0, \"hit_plus\": 1, \"dam_plus\": 1}}, "
        "{\"Weapon\": {\"name\": \"bow\", \"num_plus\": 0, \"hit_plus\": 1, \"dam_plus\": 0}}, {\"Weapon\": {\"name\": \"arrow\", \"num_plus\": 25, \"hit_plus\": 0, \"dam_plus\": 0}}]";
    JVal deflist;
    if (!list) {
        rgjson::Parser ps{DEFAULT_INIT_ITEMS, DEFAULT_INIT_ITEMS + strlen(DEFAULT_INIT_ITEMS), ""};
        if (!ps.parse(deflist)) return "internal: default init_items";
        list = &deflist;
    }
    // ItemHandler::init_player_items (item/mod.rs:411-422): initialize every entry in order, add it to the pack
    std::vector<PackItem> pack;
    out->init_draws.clear();
    std::string first_weapon, first_armor;      // Player::get_initial_weapon / get_initial_armor: the FIRST InitItem of that variant (player.rs:198-213)
    bool have_first_weapon = false, have_first_armor = false;
    std::string canon = "[";
    for (size_t idx = 0; idx < list->arr.size(); idx++) {
        const JVal &e = list->arr[idx];
        // InitItem is an externally tagged enum (item/mod.rs:166-178): {"Noinit": Item} | {"Armor": {...}} | {"Weapon": {...}}
        if (e.kind != JVal::Obj || e.obj.size() != 1 || e.obj[0].second.kind != JVal::Obj)
            return "invalid `init_items` entry: expected {\"Noinit\": {..}}, {\"Armor\": {..}} or {\"Weapon\": {..}}";
        const std::string &tag = e.obj[0].first;
        const JVal &b = e.obj[0].second;
        PackItem it{};
        if (idx) canon += ", ";
        if (tag == "Weapon") {
            std::string name; uint32_t num_plus; int32_t hit_plus, dam_plus;
            if (!r.str(&b, "InitItem::Weapon", "name", &name) || !r.u32(&b, "InitItem::Weapon", "num_plus", &num_plus) ||
                !r.i32(&b, "InitItem::Weapon", "hit_plus", &hit_plus) || !r.i32(&b, "InitItem::Weapon", "dam_plus", &dam_plus))
                return r.err;
            const WStat *st = nullptr;  // Handler::gen_item_by: the first status with that name (handler.rs:54-62)
            for (const WStat &w : weapons) if (w.name == name) { st = &w; break; }
            if (!st) return "Invalid Setting: Specified item " + name + " is not registerd to WeaponHandler";  // item/mod.rs:216-220
            if (!(st->init_lo < st->init_hi)) return "Invalid Setting: weapon `" + name + "` has an empty init_num (the reference asserts `invalid range!!`, rng.rs:84-89)";
            out->init_draws.push_back(st->init_lo); out->init_draws.push_back(st->init_hi);  // WeaponStatus::build: rng.range(init_num) (weapon.rs:159)
            it.kind = PackItem::Weapon; it.name = st->name; it.wield = st->wield; it.hit_plus = hit_plus; it.dam_plus = dam_plus;  // 0 + plus (weapon.rs:164-165, item/mod.rs:195-196)
            it.how_many = 0;  // num + num_plus: the weapon count, read by nothing on the hot path
            if (!have_first_weapon) { have_first_weapon = true; first_weapon = name; }
            canon += "{\"Weapon\": {\"name\": " + quote(name) + ", \"num_plus\": " + std::to_string(num_plus) + ", \"hit_plus\": " + std::to_string(hit_plus) +
                     ", \"dam_plus\": " + std::to_string(dam_plus) + "}}";
        } else if (tag == "Armor") {
            std::string name; int32_t def_plus;
            if (!r.str(&b, "InitItem::Armor", "name", &name) || !r.i32(&b, "InitItem::Armor", "def_plus", &def_plus)) return r.err;
            const AStat *st = nullptr;
            for (const AStat &a : armors) if (a.name == name) { st = &a; break; }
            if (!st) return "Invalid Setting: Specified item " + name + " is not registerd to WeaponHandler";  // (the reference's text names the weapon handler for armor too)
            it.kind = PackItem::Armor; it.name = st->name; it.def = (int64_t)st->def + def_plus; it.how_many = 1;
            if (!have_first_armor) { have_first_armor = true; first_armor = name; }
            canon += "{\"Armor\": {\"name\": " + quote(name) + ", \"def_plus\": " + std::to_string(def_plus) + "}}";
        } else if (tag == "Noinit") {  // a literal Item {kind, how_many, attr} (item/mod.rs:224-229)
            int64_t attr;
            if (!r.u32(&b, "Item", "how_many", &it.how_many) || !r.integer(&b, "Item", "attr", 0, 255, &attr)) return r.err;
            const JVal *k = b.get("kind");
            if (!k) return "Item: missing field `kind`";
            std::string kind_json;
            if (k->kind == JVal::Str) {  // unit variants
                static const struct { const char *n; PackItem::Kind k; } UNIT[] = {{"Gold", PackItem::Gold}, {"Potion", PackItem::Potion}, {"Ring", PackItem::Ring},
                                                                                  {"Scroll", PackItem::Scroll}, {"Wand", PackItem::Wand}};
                bool ok = false;
                for (auto &u : UNIT) if (k->s == u.n) { it.kind = u.k; ok = true; }
                if (!ok) return "Item: unknown variant `" + k->s + "` of ItemKind";
                kind_json = quote(k->s);
            } else if (k->kind == JVal::Obj && k->obj.size() == 1) {
                const std::string &kt = k->obj[0].first;
                const JVal &kb = k->obj[0].second;
                if (kt == "Food") {
                    if (kb.kind != JVal::Str || (kb.s != "Ration" && kb.s != "Slime" && kb.s != "Custom")) return "Item: invalid Food variant";
                    it.kind = PackItem::Food;
                    kind_json = "{\"Food\": " + quote(kb.s) + "}";
                } else if (kt == "Armor" && kb.kind == JVal::Obj) {  // Armor {name, worth, def, def_plus} (armor.rs:88-94)
                    uint32_t worth; int32_t def, def_plus;
                    if (!r.str(&kb, "Armor", "name", &it.name) || !r.u32(&kb, "Armor", "worth", &worth) || !r.i32(&kb, "Armor", "def", &def) || !r.i32(&kb, "Armor", "def_plus", &def_plus))
                        return r.err;
                    it.kind = PackItem::Armor; it.def = (int64_t)def + def_plus;
                    kind_json = "{\"Armor\": {\"name\": " + quote(it.name) + ", \"worth\": " + std::to_string(worth) + ", \"def\": " + std::to_string(def) + ", \"def_plus\": " +
                                std::to_string(def_plus) + "}}";
                } else if (kt == "Weapon" && kb.kind == JVal::Obj) {  // Weapon {at_weild, at_throw, name, hit_plus, dam_plus, worth, launcher} (weapon.rs:83-92)
                    Dice thrw; uint32_t worth;
                    if (!r.dice(&kb, "Weapon", "at_weild", &it.wield) || !r.dice(&kb, "Weapon", "at_throw", &thrw) || !r.str(&kb, "Weapon", "name", &it.name) ||
                        !r.integer(&kb, "Weapon", "hit_plus", -0x7fffffffffffffffLL - 1, 0x7fffffffffffffffLL, &it.hit_plus) ||
                        !r.integer(&kb, "Weapon", "dam_plus", -0x7fffffffffffffffLL - 1, 0x7fffffffffffffffLL, &it.dam_plus) || !r.u32(&kb, "Weapon", "worth", &worth))
                        return r.err;
                    const JVal *la = kb.get("launcher");
                    if (la && la->kind != JVal::Null && la->kind != JVal::Str) return "Weapon: invalid type for `launcher`";
                    it.kind = PackItem::Weapon;
                    kind_json = "{\"Weapon\": {\"at_weild\": " + dice_json(it.wield) + ", \"at_throw\": " + dice_json(thrw) + ", \"name\": " + quote(it.name) + ", \"hit_plus\": " +
                                std::to_string(it.hit_plus) + ", \"dam_plus\": " + std::to_string(it.dam_plus) + ", \"worth\": " + std::to_string(worth) + ", \"launcher\": " +
                                (la && la->kind == JVal::Str ? quote(la->s) : std::string("null")) + "}}";
                } else return "Item: unknown variant `" + kt + "` of ItemKind";
            } else return "Item: invalid type for `kind`";
            canon += "{\"Noinit\": {\"kind\": " + kind_json + ", \"how_many\": " + std::to_string(it.how_many) + ", \"attr\": " + std::to_string(attr) + "}}";
        } else return "unknown variant `" + tag + "`, expected one of `Noinit`, `Armor`, `Weapon`";
        // ItemBox::add: the lowest free slot of `max_items`, else the build fails (itembox.rs:21-29, item/mod.rs:415-419)
        if (pack.size() >= out->max_items) return "Invalid Setting: [init_player_items] Failed to add item";
        pack.push_back(it);
    }
    canon += "]";
    out->init_items_json = canon;
    out->init_items_default = canon == DEFAULT_INIT_ITEMS;
    g.n_init_draws = (int32_t)(out->init_draws.size() / 2);

    // Player::init_items: equip_from_box = the first pack item of the kind whose name equals the first InitItem's (player.rs:140-152,214-220)
    const PackItem *wpn = nullptr, *arm = nullptr, *gold = nullptr;
    for (const PackItem &p : pack) {
        if (!wpn && have_first_weapon && p.kind == PackItem::Weapon && p.name == first_weapon) wpn = &p;
        if (!arm && have_first_armor && p.kind == PackItem::Armor && p.name == first_armor) arm = &p;
        if (!gold && p.kind == PackItem::Gold) gold = &p;  // RunTime::player_status: the first Gold token (core/src/lib.rs:348-353)
    }
    // fight::player_attack (fight.rs:21-33): weapon dice or 1d4; hit_plus / dam_plus or 0
    Dice d = wpn ? wpn->wield : Dice{1, 4};
    int64_t hit_plus = wpn ? wpn->hit_plus : 0, dam_plus = wpn ? wpn->dam_plus : 0;
    if (d.times > 0 && d.max < 1) return "Invalid Setting: the wielded weapon's dice need max >= 1 (the reference asserts `invalid range!!`, rng.rs:84-89)";
    // what the 32-bit device arithmetic holds: every roll < 2^31 and the whole damage sum < 2^31 (a real table is 1..4 dice of 1..6)
    const int64_t LIM = 0x3fffffff;
    if (d.times > 1024 || d.max > LIM || (int64_t)d.times * (d.max > 0 ? d.max : 0) > LIM || dam_plus > LIM || dam_plus < -LIM || hit_plus > LIM || hit_plus < -LIM)
        return "Invalid Setting: weapon dice / plus values beyond 2^30 (or more than 1024 dice) are not supported by the HIP stepper";
    g.wpn_times = (int32_t)d.times; g.wpn_max = d.times ? (int32_t)d.max : 1;
    g.wpn_hit_plus = (int32_t)hit_plus; g.wpn_dam_plus = (int32_t)dam_plus;
    int64_t def = arm ? arm->def : 0;
    if (def > LIM || def < -LIM) return "Invalid Setting: armor def beyond 2^30 is not supported by the HIP stepper";
    g.armor_def = (int32_t)def;
    g.init_gold = gold ? gold->how_many : 0;
    // actions::get_item -> ItemBox::entry (itembox.rs:30-40): dungeon gold is_many, so it merges into the first Gold item; without one it takes the
    // lowest free slot -- and from then on merges.  With a full pack and no Gold item every pickup fails (get_item returns None, the gold stays).
    g.can_pickup = (gold != nullptr || pack.size() < out->max_items) ? 1 : 0;
    return "";
}
