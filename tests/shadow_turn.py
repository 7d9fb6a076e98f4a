"""A SECOND, independently written restatement of the reference's TURN -- test infrastructure, like oracle/.

tools/pin_map.py (profiles/r05_pin_map.txt) shows which readings of the source the reference's own goldens pin: nearly all of the level generator, and
almost nothing of the turn behind the first few keys (combat dice, level-up, healing, erratic monsters, search, the turn structure of runs / NoOp /
search, the monsters' overwrite and corner-cutting rules, the stale DistCache).  There "HIP == C oracle" rested on ONE reading of the Rust text.  This
file is another reading, written from the Rust sources function by function (the citations are the lines it follows) without looking at the C: a plain
Python model of `actions::process_action` and everything below it.  tests/test_oracle_shadow.py runs it in lock step with the C oracle -- levels come from
the oracle (the generator is pinned by the reference's goldens), every key is then played by BOTH, and the whole state is compared after every key.

Not modelled (out of the hot path's scope): items other than gold, traps, thrown weapons, the inventory."""

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF

# Direction (dungeon/coord.rs:198-242) in enum order, and its x() / y() components
DIRS = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, -1), (-1, 1), (1, 1), (0, 0)]
UP, DOWN, LEFT, RIGHT, LEFTUP, RIGHTUP, LEFTDOWN, RIGHTDOWN, STAY = range(9)
# Surface (dungeon/rogue/mod.rs:137-147) and its tiles (:149-163)
PASSAGE, FLOOR, WALLX, WALLY, STAIR, DOOR, TRAP, NONE = range(8)
TILES = "#.-|%+^ "
# CellAttr (dungeon/field.rs:107-124)
VISITED, HIDDEN, VISIBLE, DRAWN, LOCKED, DARK = 1, 2, 4, 8, 16, 32
# EnemyAttr (character/enemies.rs:126-139)
MEAN, RANDOM, CONFUSED = 0b1, 0b001_000_000_000, 0b010_000_000_000
INF = 0xFFFFFFFF

# KeyMap::ai (input.rs:73-100)
KEYMAP = {"l": ("move", RIGHT), "k": ("move", UP), "j": ("move", DOWN), "h": ("move", LEFT), "u": ("move", RIGHTUP), "y": ("move", LEFTUP),
          "n": ("move", RIGHTDOWN), "b": ("move", LEFTDOWN), ".": ("noop", None), "L": ("run", RIGHT), "K": ("run", UP), "J": ("run", DOWN),
          "H": ("run", LEFT), "U": ("run", RIGHTUP), "Y": ("run", LEFTUP), "N": ("run", RIGHTDOWN), "B": ("run", LEFTDOWN), "s": ("search", None),
          ">": ("downstair", None)}

# BUILTIN_ENEMIES (character/enemies.rs:474-761), transcribed by script from the Rust table: glyph -> (attack dice (times, max), attr names, defense, exp, level)
BUILTIN = {
    "A": ([(0, 0)], ("MEAN", "RUSTS_ARMOR"), 10, 20, 5), "B": ([(1, 2)], ("FLYING", "RANDOM"), 3, 1, 1), "C": ([(1, 2), (1, 5), (1, 5)], (), 4, 17, 4),
    "D": ([(1, 8), (1, 8), (3, 10)], ("MEAN",), 3, 5000, 10), "E": ([(1, 2)], ("MEAN",), 7, 2, 1), "F": ([], ("MEAN",), 3, 80, 8),
    "G": ([(4, 3), (3, 5)], ("FLYING", "MEAN", "REGENERATE"), 2, 2000, 13), "H": ([(1, 8)], ("MEAN",), 5, 3, 1), "I": ([(0, 0)], ("FREEZES",), 9, 5, 1),
    "J": ([(2, 12), (2, 4)], (), 6, 3000, 15), "K": ([(1, 4)], ("MEAN",), 7, 1, 1), "L": ([(1, 1)], ("STEAL_GOLD",), 8, 10, 3),
    "M": ([(3, 4), (3, 4), (2, 5)], ("MEAN",), 2, 200, 8), "N": ([(0, 0)], (), 9, 37, 3), "O": ([(1, 8)], ("GREEDY",), 6, 5, 1),
    "P": ([(4, 4)], ("INVISIBLE",), 3, 120, 8), "Q": ([(1, 5), (1, 5)], ("MEAN",), 3, 15, 3), "R": ([(1, 6)], ("REDUCE_STR", "MEAN"), 3, 9, 2),
    "S": ([(1, 3)], ("MEAN",), 5, 2, 1), "T": ([(1, 8), (1, 8), (2, 6)], ("MEAN", "REGENERATE"), 4, 120, 6), "U": ([(1, 9), (1, 9), (2, 9)], ("MEAN",), -2, 190, 7),
    "V": ([(1, 19)], ("MEAN", "REGENERATE"), 1, 350, 8), "W": ([(1, 6)], (), 4, 55, 5), "X": ([(4, 4)], (), 7, 100, 7), "Y": ([(1, 6), (1, 6)], (), 6, 50, 4),
    "Z": ([(1, 8)], ("MEAN",), 8, 6, 2),
}
# Leveling::default (character/player.rs:308-343)
LEVEL_EXPS = [10, 20, 40, 80, 160, 320, 640, 1300, 2600, 5200, 13000, 26000, 50000, 100_000, 200_000, 400_000, 800_000, 2_000_000, 4_000_000, 8_000_000, 0xFFFFFFFF]
# fight.rs:89-109
HIT_PROB_PLUS = [-7, -6, -5, -4, -3, -2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3]
DAMAGE_PLUS = [-7, -6, -5, -4, -3, -2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 6]
PLAYER_STRENGTH = 16  # StatusInner::from_config (player.rs:284-290)
ENEMY_STRENGTH = 10   # Enemy::STRENGTH (enemies.rs:161)


class Rng:
    """rand_xorshift 0.2 XorShiftRng + rand 0.7 UniformInt::sample_single, via RngHandle (core/src/rng.rs:84-98)."""
    __slots__ = ("s",)

    def __init__(self, words):
        self.s = [int(w) for w in words]

    def next_u32(self):
        x, y, z, w = self.s
        t = (x ^ (x << 11)) & M32
        w2 = (w ^ (w >> 19) ^ t ^ (t >> 8)) & M32
        self.s = [y, z, w, w2]
        return w2

    def next_u64(self):
        lo = self.next_u32()
        return (self.next_u32() << 32) | lo

    def gen_range(self, low, high, bits):
        mask = M32 if bits == 32 else M64
        rng = (high - low) & mask
        zone = ((rng << (bits - rng.bit_length())) - 1) & mask
        while True:
            v = self.next_u32() if bits == 32 else self.next_u64()
            m = v * rng
            if (m & mask) <= zone:
                return low + (m >> bits)

    def does_happen(self, p_inv):      # rng.rs:91-93: gen_range(0u32, p_inv) == 0
        return self.gen_range(0, p_inv, 32) == 0

    def parcent(self, p):              # rng.rs:95-98: range(1..=100) of Parcent's u32
        return self.gen_range(1, 101, 32) <= p


def strength_table(table, st):         # fight.rs:89-109
    return 0 if st <= 0 or st > len(table) else table[st - 1]


class Monster:
    __slots__ = ("glyph", "hp", "exp", "level", "defense", "running")

    def __init__(self, glyph, hp, exp, level, defense, running):
        self.glyph, self.hp, self.exp, self.level, self.defense, self.running = glyph, hp, exp, level, defense, running

    def attr(self):
        v = 0
        for a in BUILTIN[self.glyph][1]:
            v |= {"MEAN": MEAN, "RANDOM": RANDOM, "CONFUSED": CONFUSED}.get(a, 0)
        return v


class Shadow:
    """RunTime's mutable state for the turn: Dungeon (floor, DistCache, rng), EnemyHandler (two BTreeMaps, rng), Player."""

    def __init__(self, width, height, hunger_time, passage_unlock_rate_inv, door_unlock_rate_inv, weapon=(2, 4, 1, 1), armor=4):
        self.W, self.H = width, height
        self.hunger_time = hunger_time
        self.passage_unlock_rate_inv, self.door_unlock_rate_inv = passage_unlock_rate_inv, door_unlock_rate_inv
        self.weapon = weapon        # at_weild (times, max), hit_plus, dam_plus of Player::weapon, or None (fight.rs:20-33): the default pack wields a mace 2d4 +1,+1
        self.armor = armor          # Player::arm: def + def_plus of the equipped armor (player.rs:125-132); ring mail 3 + 1
        self.dist_cache = []        # DistCache (rogue/mod.rs:492-518): a VecDeque of (map, coord), lives as long as the Dungeon

    # ---- state exchange with the oracle (levels are the oracle's; the turn is played here) --------------------------------------------
    def load_level(self, o):
        surf, attr, doors, gold = o.grid()
        self.surface = [int(v) for v in surf.reshape(-1)]
        self.attr = [int(v) for v in attr.reshape(-1)]
        self.doors = {i for i, v in enumerate(doors.reshape(-1)) if v}
        self.items = {i: int(v) for i, v in enumerate(gold.reshape(-1)) if v >= 0}
        self.rooms = o.rooms()
        self.placed, self.active = {}, {}   # BTreeMap<DungeonPath, Rc<Enemy>> keyed by [level, x, y]: iteration order = (x, y)
        for m in o.monsters():
            mon = Monster(chr(65 + m["type"]), m["hp"], m["exp"], m["level"], m["defense"], bool(m["running"]))
            (self.active if m["active"] else self.placed)[(m["x"], m["y"])] = mon
        sc = o.scalars()
        self.pos = (sc["px"], sc["py"])
        self.level = sc["level"]
        words, _ = o.rng()
        self.rng_dungeon, self.rng_enemy = Rng(words[0]), Rng(words[2])   # Dungeon.rng / EnemyHandler.rng (the item stream is only drawn on by generation)
        self.rng_item_words = [int(w) for w in words[1]]

    def load_player(self, o):
        sc = o.scalars()
        self.hp, self.hp_max, self.exp, self.plevel = sc["hp"], sc["hp_max"], sc["exp"], sc["plevel"]
        self.food_left, self.quiet, self.gold = sc["food_left"], sc["quiet"], sc["gold"]
        self.dead = False

    def new_game(self, o):
        self.dist_cache = []
        self.load_level(o)
        self.load_player(o)

    def snapshot(self):
        mons = sorted([(x, y, m.glyph, 0, m.hp, m.exp) for (x, y), m in self.placed.items()] + [(x, y, m.glyph, 1, m.hp, m.exp) for (x, y), m in self.active.items()])
        return dict(pos=self.pos, hp=self.hp, hp_max=self.hp_max, exp=self.exp, plevel=self.plevel, food_left=self.food_left & M32, quiet=self.quiet, gold=self.gold,
                    level=self.level, rng_dungeon=list(self.rng_dungeon.s), rng_enemy=list(self.rng_enemy.s), monsters=mons, dead=self.dead)

    # ---- field helpers --------------------------------------------------------------------------------------------------------------
    def idx(self, x, y):
        return y * self.W + x

    def inside(self, x, y):          # Field::try_get_p for the coordinates the turn produces (App. C-6: the off-by-one bounds are not reachable)
        return 0 <= x < self.W and 0 <= y < self.H

    @staticmethod
    def can_walk(surface):           # Surface::can_walk (rogue/mod.rs:177-182)
        return surface not in (WALLX, WALLY, NONE)

    def can_move_impl(self, cd, d, is_enemy):   # Floor::can_move_impl (floor.rs:169-182): None (off the field) counts as false for every caller
        nx, ny = cd[0] + DIRS[d][0], cd[1] + DIRS[d][1]
        if not self.inside(nx, ny):
            return False
        i = self.idx(nx, ny)
        res = self.can_walk(self.surface[i])
        if not is_enemy:
            res = res and not (self.attr[i] & HIDDEN) and not (self.attr[i] & LOCKED)
        if d in (LEFTUP, RIGHTUP, LEFTDOWN, RIGHTDOWN):
            ax, ay = cd[0] + DIRS[d][0], cd[1]        # cd + direction.x()
            bx, by = cd[0], cd[1] + DIRS[d][1]        # cd + direction.y()
            if not self.inside(ax, ay) or not self.inside(bx, by):
                return False
            res = res and self.can_walk(self.surface[self.idx(ax, ay)]) and self.can_walk(self.surface[self.idx(bx, by)])
        return res

    def room_of(self, cd):           # Floor::cd_to_room_id (floor.rs:194-200): the first room whose ASSIGNED AREA contains cd
        for i, r in enumerate(self.rooms):
            x0, y0, x1, y1 = r["assigned"]
            if x0 <= cd[0] < x1 and y0 <= cd[1] < y1:
                return i
        return None

    def with_current_room(self, cd, select, mark):   # floor.rs:201-229
        rid = self.room_of(cd)
        assert rid is not None, "[Floor::with_current_room] no room for given coord"
        room = self.rooms[rid]
        if not select(room):
            return
        x0, y0, x1, y1 = room["range"] if room["kind"] != 2 else room["assigned"]   # room.range().unwrap_or(assigned_area)
        for y in range(y0, y1):
            for x in range(x0, x1):
                is_edge = x in (x0, x1 - 1) or y in (y0, y1 - 1)
                mark(self.idx(x, y), is_edge)

    def enters_room(self, cd):       # floor.rs:231-247
        def select(room):
            if room["visited"]:
                return False
            room["visited"] = True
            return room["kind"] == 0 and not room["dark"]

        def mark(i, _):
            self.attr[i] |= DRAWN | VISIBLE
        self.with_current_room(cd, select, mark)

    def leaves_room(self, cd):       # floor.rs:249-261
        def mark(i, is_edge):
            if not is_edge:
                self.attr[i] &= ~VISIBLE
        self.with_current_room(cd, lambda room: room["visited"] and room["dark"], mark)

    def activate(self, place):       # EnemyHandler::activate (enemies.rs:353-358)
        m = self.placed.pop(place, None)
        if m is None:
            return
        m.running = True
        self.active[place] = m

    def player_in(self, cd, init):   # Floor::player_in (floor.rs:264-295)
        if init or self.idx(*cd) in self.doors:
            self.enters_room(cd)
            rid = self.room_of(cd)
            if rid is not None:
                x0, y0, x1, y1 = self.rooms[rid]["assigned"]
                for p in [p for p, m in sorted(self.placed.items()) if x0 <= p[0] < x1 and y0 <= p[1] < y1 and (m.attr() & MEAN)]:   # activate_area (enemies.rs:342-352)
                    self.activate(p)
        self.attr[self.idx(*cd)] |= VISITED
        for d in range(9):
            x, y = cd[0] + DIRS[d][0], cd[1] + DIRS[d][1]
            if not self.inside(x, y):
                continue
            i = self.idx(x, y)
            diag = d in (LEFTUP, RIGHTUP, LEFTDOWN, RIGHTDOWN)
            if not diag or self.surface[i] != PASSAGE:
                if not (self.attr[i] & HIDDEN):            # Cell::approached (field.rs:20-26)
                    self.attr[i] |= DRAWN | VISIBLE

    def player_out(self, cd):        # Floor::player_out (floor.rs:298-312)
        if self.idx(*cd) in self.doors:
            self.leaves_room(cd)
        for d in range(9):
            x, y = cd[0] + DIRS[d][0], cd[1] + DIRS[d][1]
            if self.inside(x, y) and self.surface[self.idx(x, y)] == FLOOR and (self.attr[self.idx(x, y)] & DARK):   # Cell::left (field.rs:29-33)
                self.attr[self.idx(x, y)] &= ~VISIBLE

    # ---- monsters: Floor::make_dist_map, DistCache, Dungeon::move_enemy[_randomly] ----------------------------------------------------
    def make_dist_map(self, origin):  # floor.rs:395-416
        dist = [INF] * (self.W * self.H)
        dist[self.idx(*origin)] = 0
        queue, head = [origin], 0
        while head < len(queue):
            cur = queue[head]
            head += 1
            cdist = dist[self.idx(*cur)]
            for d in range(8):
                nx, ny = cur[0] + DIRS[d][0], cur[1] + DIRS[d][1]
                if not self.inside(nx, ny):
                    continue
                if dist[self.idx(nx, ny)] != INF or not self.can_move_impl(cur, d, True):
                    continue
                queue.append((nx, ny))
                dist[self.idx(nx, ny)] = cdist + 1
        return dist

    def cached_dist_map(self, cd):    # DistCache::make_dist_map (rogue/mod.rs:504-517): keyed by the coordinate alone, never invalidated
        for m, key in self.dist_cache:
            if key == cd:
                return m
        m = self.make_dist_map(cd)
        before = len(self.dist_cache)
        self.dist_cache.append((m, cd))
        if before > 8:                # MAX_CACHED_DIST
            self.dist_cache.pop(0)
        return m

    def move_enemy(self, cur, target, skip):   # rogue/mod.rs:339-375 -> ("reach" | "cant" | (x, y))
        dist_map = self.cached_dist_map(target)
        cand = []
        for d in range(9):            # Direction::into_enum_iter(): all nine, Stay included
            nxt = (cur[0] + DIRS[d][0], cur[1] + DIRS[d][1])
            if skip(nxt):
                continue
            if not self.inside(*nxt):  # `*dist_map.get_p(next)` panics outside the grid (a chaser on column 0 / W-1): no result to reproduce; not a candidate
                continue
            ndist = dist_map[self.idx(*nxt)]
            if ndist == 0 and self.can_move_impl(cur, d, True):
                return "reach"
            if ndist != INF and ndist > 0:
                cand.append((ndist, nxt))
        if not cand:
            return "cant"
        cand.sort(key=lambda t: t[0])  # sort_by_key is stable: the first minimum in direction order
        return cand[0][1]

    def move_enemy_randomly(self, cur, player, skip):   # rogue/mod.rs:376-397
        d = self.rng_dungeon.gen_range(0, 8, 64)        # rng.range(0..8) over usize
        nxt = (cur[0] + DIRS[d][0], cur[1] + DIRS[d][1])
        if skip(nxt) or not self.can_move_impl(cur, d, True):
            return "cant"
        return "reach" if nxt == player else nxt

    def move_actives(self):           # EnemyHandler::move_actives (enemies.rs:366-424), gold_pos = None (actions.rs:88)
        attacks = []
        taken, self.active = self.active, {}
        for path in sorted(taken):    # BTreeMap order of [level, x, y]
            enemy = taken[path]

            def skip(p):
                return p in self.active or p in self.placed
            attr = enemy.attr()
            if (self.rng_enemy.does_happen(2) and (attr & RANDOM)) or (not self.rng_enemy.does_happen(5) and (attr & CONFUSED)):
                res = self.move_enemy_randomly(path, self.pos, skip)
            else:
                res = self.move_enemy(path, self.pos, skip)
            if res == "reach":
                attacks.append(enemy)
                nxt = path
            elif res == "cant":
                nxt = path
            else:
                nxt = res
            self.active[nxt] = enemy  # BTreeMap::insert: an enemy already at `nxt` is replaced
        return attacks

    # ---- fight.rs --------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def attack_rate(level, armor, revision):            # fight.rs:84-87 + Parcent::truncate
        return max(0, min(100, (level + armor + revision + 1) * 5))

    def roll(self, dices, rate, dam_plus):               # fight.rs:52-72
        did_hit, total = False, 0
        for times, mx in dices:
            if not self.rng_enemy.parcent(rate):
                continue
            did_hit = True
            for _ in range(times):                       # Damage::random for Dice<HitPoint> (character/mod.rs:229-234): range(1..=max) over i64
                total += self.rng_enemy.gen_range(1, mx + 1, 64)
            total += dam_plus
        return total if did_hit else None

    def fight_player_attack(self, enemy):                # fight.rs:6-39 without a thrown weapon
        hit_plus = self.weapon[2] if self.weapon else 0
        str_p = strength_table(HIT_PROB_PLUS, PLAYER_STRENGTH) + (0 if enemy.running else 4) + hit_plus    # attack_rate_player (fight.rs:74-78)
        rate = self.attack_rate(self.plevel, enemy.defense, str_p)
        dam_plus = self.weapon[3] if self.weapon else 0
        dice = (self.weapon[0], self.weapon[1]) if self.weapon else (1, 4)
        return self.roll([dice], rate, dam_plus + strength_table(DAMAGE_PLUS, PLAYER_STRENGTH))

    def fight_enemy_attack(self, enemy):                 # fight.rs:41-50
        rate = self.attack_rate(enemy.level, self.armor, strength_table(HIT_PROB_PLUS, ENEMY_STRENGTH))
        return self.roll(BUILTIN[enemy.glyph][0], rate, strength_table(DAMAGE_PLUS, ENEMY_STRENGTH) + strength_table(DAMAGE_PLUS, PLAYER_STRENGTH))

    # ---- character/player.rs ---------------------------------------------------------------------------------------------------------
    def level_up(self, exp):                             # player.rs:185-197 + Leveling::check_level (:345-352)
        self.exp += exp
        cur = self.plevel - 1
        diff = 0 if cur >= len(LEVEL_EXPS) else next(i for i, e in enumerate(LEVEL_EXPS[cur:]) if self.exp < e)
        if diff > 0:
            self.plevel += diff
            gain = sum(self.rng_enemy.gen_range(1, 11, 64) for _ in range(diff))   # Dice::new(diff, HitPoint(10)).exec::<i64>
            self.hp_max += gain                                                    # Maxed<HitPoint> += : both ends (character/mod.rs)
            self.hp += gain
            return True
        return False

    def heal(self):                                      # player.rs:221-240
        self.quiet += 1
        quiet, level = self.quiet, self.plevel
        if level < 8:
            amount = max(0, min(1, quiet + (level << 1) - 20))
        elif quiet >= 3:
            amount = self.rng_enemy.gen_range(1, level - 6, 64)   # rng.range(1..level - 6) over i64
        else:
            amount = 0
        if amount > 0:
            self.hp = min(self.hp + amount, self.hp_max)
            self.quiet = 0
            return True
        return False

    def turn_passed(self):                               # player.rs:163-176
        self.food_left -= 1                              # u32: wraps below zero in a release build (App. C-9)
        if self.food_left == 0:
            return                                       # [PlayerEvent::Dead], which after_turn ignores (actions.rs:75)
        self.heal()                                      # (notify_hungry only reports)

    # ---- actions.rs ------------------------------------------------------------------------------------------------------------------
    def after_turn(self):                                # actions.rs:67-119 -> True when the player died
        self.turn_passed()
        attacks = self.move_actives()
        if attacks:
            self.quiet = 0                               # player.buttle()
        for enemy in attacks:
            hp = self.fight_enemy_attack(enemy)
            if hp is not None:
                self.hp = max(self.hp - hp, 0)           # Player::get_damage (player.rs:177-184)
                if self.hp == 0:
                    return True
        return False

    def player_attack(self, place):                      # actions.rs:140-166
        enemy = self.placed.get(place) or self.active.get(place)     # get_cloned BEFORE activate: the same Rc afterwards
        self.quiet = 0
        self.activate(place)
        hp = self.fight_player_attack(enemy)
        if hp is not None:
            cur = enemy.hp                               # Enemy::get_damage (enemies.rs:205-213)
            if cur <= hp:
                self.placed.pop(place, None)
                self.active.pop(place, None)
                self.level_up(enemy.exp)
            else:
                enemy.hp = hp - cur                      # (sic) damage - cur

    def move_player(self, d):                            # actions.rs:168-194 -> done
        if not self.can_move_impl(self.pos, d, False):
            return True
        new_pos = (self.pos[0] + DIRS[d][0], self.pos[1] + DIRS[d][1])
        if new_pos in self.placed or new_pos in self.active:
            self.player_attack(new_pos)
            return True
        self.player_out(self.pos)                        # Dungeon::move_player (rogue/mod.rs:237-258)
        self.player_in(new_pos, False)
        self.pos = new_pos
        i = self.idx(*new_pos)
        if i in self.items:                              # get_item (actions.rs:206-231): gold merges into the pack's gold; Floor::remove_obj always finds the
            self.gold += self.items.pop(i)               # cell filled (the player stands on it), so the item is removed
            return True
        return False

    def search(self):                                    # Floor::search (floor.rs:349-370) on the dungeon's own stream
        for d in range(8):
            x, y = self.pos[0] + DIRS[d][0], self.pos[1] + DIRS[d][1]
            if not self.inside(x, y):
                continue
            i = self.idx(x, y)
            if (self.attr[i] & HIDDEN) and self.rng_dungeon.does_happen(self.passage_unlock_rate_inv):
                self.attr[i] = (self.attr[i] & ~(LOCKED | HIDDEN)) | VISIBLE     # Cell::unlock (field.rs:83-87)
                self.surface[i] = PASSAGE
            if (self.attr[i] & LOCKED) and self.rng_dungeon.does_happen(self.door_unlock_rate_inv):
                self.attr[i] = (self.attr[i] & ~(LOCKED | HIDDEN)) | VISIBLE
                self.surface[i] = DOOR

    def tile_under_player(self):                         # Cell::tile (field.rs:91-98)
        i = self.idx(*self.pos)
        return TILES[self.surface[i]] if self.attr[i] & VISIBLE else " "

    def process_action(self, key, new_level):
        """actions::process_action (actions.rs:16-65) for one key of the `ai` keymap.  new_level(): called for a DownStair on the stairs, must put the next
        level (generated by the oracle) into this object and place the player (actions::new_level).  Returns whether the key's reactions hold a Grave transition."""
        assert not self.dead, "IgnoredInput: the Grave modal takes no action (core/src/lib.rs:301-315)"
        act, d = KEYMAP[key]
        ui_dead, grave = False, False     # `ui`: the LAST after_turn's result; `grave`: a Reaction::UiTransition(Grave) was pushed by ANY of them
        if act == "downstair":
            if self.surface[self.idx(*self.pos)] == STAIR:
                new_level()
            ui_dead = grave = self.after_turn()
        elif act == "move":
            self.move_player(d)
            ui_dead = grave = self.after_turn()
        elif act == "run":
            while True:
                done = self.move_player(d)
                tile = self.tile_under_player()
                if done or tile not in ".#":
                    break
                ui_dead = self.after_turn()          # a run goes on after the player's death: a later turn without a hit leaves `ui` = None again,
                grave = grave or ui_dead             # but the Grave reaction stays in the key's reaction list
        elif act == "search":
            self.search()
            ui_dead = grave = self.after_turn()
        if ui_dead:
            self.dead = True                         # RunTime::ui = Mordal(Grave) (core/src/lib.rs:316-318)
        return grave                                 # GameStateImpl::react: is_terminal = a Grave transition among the reactions || steps >= max_steps (state_impls.rs:56-78)
