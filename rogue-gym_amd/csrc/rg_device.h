// rg_device.h -- device helpers shared by the step kernels (rg_kernels.hip) and the render / observation kernels (rg_obs.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "../../include/rogue_gym_hip.h"
#include "rg_state.h"

#define WAVE 64
#define DIST_INF 0xFFFFu

// ---------------------------------------------------------------------------------------------
// static tables
// ---------------------------------------------------------------------------------------------
// Direction enum order (dungeon/coord.rs:198-242): Up Down Left Right LeftUp RightUp LeftDown RightDown Stay
static __device__ __constant__ int8_t kDX[9] = {0, 0, -1, 1, -1, 1, -1, 1, 0};
static __device__ __constant__ int8_t kDY[9] = {-1, 1, 0, 0, -1, -1, 1, 1, 0};
// Surface::tile (rogue/mod.rs:149-163): '#' '.' '-' '|' '%' '+' '^' ' ' packed into one 64-bit immediate (a __constant__ table indexed per lane
// would be a memory load per cell)
// (v_perm_b32: byte `surface & 7` of the 8-byte table, one VALU op instead of a 64-bit shift)
__device__ __forceinline__ uint32_t glyph_of(uint32_t surface) { return (uint32_t)(0x205E2B257C2D2E23ull >> (8 * (surface & 7))) & 0xffu; }

// monster statuses come from the config (RgConfig::mon, rarity-sorted): builtin presets (character/enemies.rs:474-761)
// or custom ones; a monster's `type` is its index in that table
#define EA_MEAN 1
#define EA_RANDOM 512
#define EA_CONFUSED 1024

// Symbol::from_tile (core/src/symbol.rs:17-40); 255 = not a symbol
__device__ __forceinline__ uint32_t tile_to_sym(uint32_t t) {
    switch (t) {
    case ' ': return 0; case '@': return 1; case '#': return 2; case '.': return 3; case '-': case '|': return 4;
    case '%': return 5; case '+': return 6; case '^': return 7; case '!': return 8; case '?': return 9; case ']': return 10;
    case ')': return 11; case '/': return 12; case '*': return 13; case ':': return 14; case '=': return 15; case ',': return 16;
    default: return (t >= 'A' && t <= 'Z') ? t - 'A' + 17 : 255u;
    }
}

// exact n / d for the small non-negative operands of this engine (n < 2^20, 0 < d < 2^12): one v_rcp instead of the
// ~30-instruction integer division sequence
// (v_rcp_f32 instead of the IEEE-rounded reciprocal would also be exact for n < 2^21 and saves ~10 instructions per call: measured neutral,
// k_build 54.1 vs 54.4 us per level, 524 vs 524 M env-steps/s -- round 3)
__device__ __forceinline__ int small_div(int n, int d) { return (int)(((float)n + 0.5f) * __frcp_rn((float)d)); }
// ... with the correctly rounded reciprocal of a CONFIG constant taken from RgConfig (rg_config_derive: 1.0f / d on the host is the same IEEE value): the device has
// no scalar float unit, so a reciprocal of a wave-uniform divisor is VALU work whose result sits in a VECTOR register -- hoisted to the top of the kernel and
// held through the whole turn (six registers in k_step_w32, round 5)
__device__ __forceinline__ int small_div_inv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

#define POS(x, y) ((uint32_t)(((x) << 8) | (y)))
#define POS_X(p) ((int)(((p) >> 8) & 0xff))
#define POS_Y(p) ((int)((p) & 0xff))

__device__ __forceinline__ bool can_walk(uint32_t c) {
    uint32_t s = c & C_SURF_MASK;
    return !(s == S_WALLX || s == S_WALLY || s == S_NONE);
}
__device__ __forceinline__ bool in_bounds(const RgConfig &c, int x, int y) { return x >= 0 && y >= 0 && x < c.width && y < c.height; }

// Room::assigned_area of room id i (rooms.rs:192-209), half-open
__device__ __forceinline__ void assigned_area(const RgConfig &c, int i, int &x0, int &y0, int &x1, int &y1) {
    int rsx = c.rsx, rsy = c.rsy;
    int cy = small_div_inv(i, c.inv_rnx), cx = i - cy * c.room_num_x;
    x0 = cx * rsx; x1 = x0 + rsx;
    y0 = cy == 0 ? 1 : cy * rsy;
    y1 = (cy + 1) * rsy;
    if (y1 == c.height) y1 -= 1;
}
// Floor::cd_to_room_id (floor.rs:194-200): areas are disjoint, so arithmetic replaces the scan
__device__ __forceinline__ int room_id_of(const RgConfig &c, int x, int y) {
    const int rsy = c.rsy;
    if (y < 1 || x < 0) return -1;
    int cx = small_div_inv(x, c.inv_rsx), cy = small_div_inv(y, c.inv_rsy);
    if (cx >= c.room_num_x || cy >= c.room_num_y) return -1;
    if ((cy + 1) * rsy == c.height && y == c.height - 1) return -1;
    return cy * c.room_num_x + cx;
}
__device__ __forceinline__ void unpack_rect(uint32_t r, int &x0, int &y0, int &x1, int &y1) {
    x0 = r & 0xff; y0 = (r >> 8) & 0xff; x1 = (r >> 16) & 0xff; y1 = r >> 24;
}

