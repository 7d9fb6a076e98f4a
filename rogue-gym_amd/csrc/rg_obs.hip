// rg_obs.hip -- render / observation-encode kernels of the batched Rogue-Gym stepper (gfx950).
//
//   k_obs<gray|symbol> : fused RunTime::draw_screen (mirror refresh of Redraw envs) + PlayerState::{gray,symbol}_image
//   k_render, k_gray, k_symbol, k_encode_scalar : unfused fallbacks
//
// Built as its own translation unit with -Os: these kernels are bandwidth/latency-bound and measurably faster with less
// unrolling (k_obs 83 -> 67-70 us at 65 536 mini envs), while the issue-bound step kernel wants -O3.
// file:line citations are relative to /root/reference.
#include <cstdlib>
#include "rg_device.h"

// ---------------------------------------------------------------------------------------------
// k_render: RunTime::draw_screen (core/src/lib.rs:264-285; rogue/mod.rs:278-300,398-404) into the
// PlayerState mirrors, only for envs whose last key produced Reaction::Redraw
// ---------------------------------------------------------------------------------------------
#define RENDER_THREADS 256
__device__ __forceinline__ bool in_same_room(const RgState &S, const RgConfig &c, int e, int ax, int ay, int bx, int by) {
    int id = room_id_of(c, ax, ay);  // Floor::in_same_room (floor.rs:381-393)
    if (id < 0 || room_id_of(c, bx, by) != id) return false;
    uint8_t meta = S.room_meta[id * S.n + e];
    if ((meta & RM_KIND_MASK) == RK_EMPTY) return true;
    int x0, y0, x1, y1;
    unpack_rect(S.room_rect[id * S.n + e], x0, y0, x1, y1);
    bool ina = ax >= x0 && ax < x1 && ay >= y0 && ay < y1, inb = bx >= x0 && bx < x1 && by >= y0 && by < y1;
    return ina == inb;
}

__global__ void __launch_bounds__(RENDER_THREADS) k_render(RgState S, RgConfig c) {
    __shared__ uint8_t s_scr[RG_MAX_W * RG_MAX_H];
    const int tid = threadIdx.x, W = c.width, H = c.height, HW = W * H, n = S.n;
    const int nrooms = c.room_num_x * c.room_num_y;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const uint32_t fl = S.flags[e];
        if (!(fl & RG_FLAG_REDRAW)) continue;
        const uint16_t *cell = S.cell + (size_t)e * HW;
        const bool upd_hist = !(fl & RG_FLAG_HIST_STALE);
        uint8_t *hist = S.hist + (size_t)e * HW;
        for (int i = tid; i < HW; i += RENDER_THREADS) {
            uint32_t v = cell[i];
            int y = i / W;
            uint8_t g = ' ';
            if (y >= 1 && y < H - 1 && (v & C_VISIBLE)) g = glyph_of(v);
            s_scr[i] = g;
            if (upd_hist) hist[i] = (v & C_VISITED) ? 1 : 0;
        }
        __syncthreads();
        const uint32_t ppos = S.p_pos[e];
        const int px = POS_X(ppos), py = POS_Y(ppos);
        // draw priority: player > gold > monster (core/src/lib.rs:271-283): lowest priority first
        for (int r = tid; r < nrooms; r += RENDER_THREADS) {  // (a grid may have more rooms than the block has threads)
            uint32_t w = S.mon_w0[r * n + e];
            if ((w >> 24) & MF_ALIVE) {
                int x = POS_X(w), y = POS_Y(w);
                uint32_t v = cell[y * W + x];
                int dx = px - x, dy = py - y;
                if ((v & (C_VISIBLE | C_DRAWN)) && y >= 1 && y < H - 1 && (dx * dx + dy * dy <= 2 || in_same_room(S, c, e, px, py, x, y)))
                    s_scr[y * W + x] = c.mon[(w >> 16) & 0xff].tile;
            }
        }
        __syncthreads();
        for (int r = tid; r < nrooms; r += RENDER_THREADS) {
            uint32_t g = S.gold_pos[r * n + e];
            if (g & 0x10000u) {
                int x = POS_X(g), y = POS_Y(g);
                if ((cell[y * W + x] & (C_VISIBLE | C_DRAWN)) && y >= 1 && y < H - 1) s_scr[y * W + x] = '*';
            }
        }
        __syncthreads();
        if (tid == 0 && (cell[py * W + px] & (C_VISIBLE | C_DRAWN)) && py >= 1 && py < H - 1) s_scr[py * W + px] = '@';
        __syncthreads();
        uint8_t *scr = S.screen + (size_t)e * HW;
        if ((HW & 3) == 0) {
            uint32_t *d4 = reinterpret_cast<uint32_t *>(scr);
            const uint32_t *s4 = reinterpret_cast<const uint32_t *>(s_scr);
            for (int i = tid; i < HW / 4; i += RENDER_THREADS) d4[i] = s4[i];
        } else
            for (int i = tid; i < HW; i += RENDER_THREADS) scr[i] = s_scr[i];
        if (tid == 0)  // the history plane was written unless this Redraw was stale
            S.flags[e] = (fl & ~(RG_FLAG_REDRAW | RG_FLAG_HIST_STALE | RG_FLAG_HIST_LAG | (upd_hist ? RG_FLAG_HIST_DIRTY : 0u))) | ((fl & RG_FLAG_HIST_STALE) ? RG_FLAG_HIST_LAG : 0u);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k_gray / k_symbol: observation encode (python/src/lib.rs:72-111,162-205; flags.rs:88-115)
// one thread = 4 consecutive cells of one env, all planes; float4 (16-byte) stores
// ---------------------------------------------------------------------------------------------
// StatusFlagInner bit b -> index into Status::to_vec
__device__ __constant__ uint8_t kStatusIdx[9] = {0, 2, 3, 4, 5, 6, 7, 8, 9};

// rs / rst: distance between two envs' records in bytes (screen, hist) and in i32 words (status): hw and 10 for the mirrors, the record size for
// a packed compact batch (rg_pack_compact)
__global__ void __launch_bounds__(256) k_gray(const uint8_t *__restrict__ screen, const uint8_t *__restrict__ hist, const int32_t *__restrict__ status,
                                              int n, int hw, size_t rs, size_t rst, int symbols, uint32_t sflag, int with_hist, float *__restrict__ out,
                                              const int32_t *__restrict__ ext) {
    const int q = hw >> 2;  // quads per env (hw % 4 == 0 checked on the host)
    const size_t total = (size_t)n * q;
    const int nplanes = 1 + __popc(sflag) + (with_hist ? 1 : 0);
    const float fsym = (float)(uint8_t)symbols;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        int e = (int)(g / q), i = (int)(g - (size_t)e * q);
        uint32_t s4 = reinterpret_cast<const uint32_t *>(screen + (size_t)e * rs)[i];
        float4 v;
        v.x = (float)(uint8_t)tile_to_sym(s4 & 0xff) / fsym;
        v.y = (float)(uint8_t)tile_to_sym((s4 >> 8) & 0xff) / fsym;
        v.z = (float)(uint8_t)tile_to_sym((s4 >> 16) & 0xff) / fsym;
        v.w = (float)(uint8_t)tile_to_sym(s4 >> 24) / fsym;
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)(ext ? ext[e] : e) * nplanes * hw) + i;
        o[0] = v;
        int p = 1;
        for (int b = 0; b < 9; b++)
            if (sflag & (1u << b)) {
                float f = (float)status[(size_t)e * rst + kStatusIdx[b]];
                float4 sv; sv.x = sv.y = sv.z = sv.w = f;
                o[(size_t)p * q] = sv;
                p++;
            }
        if (with_hist) {
            uint32_t h4 = reinterpret_cast<const uint32_t *>(hist + (size_t)e * rs)[i];
            float4 hv;
            hv.x = (h4 & 0xff) ? 1.f : 0.f; hv.y = (h4 & 0xff00) ? 1.f : 0.f; hv.z = (h4 & 0xff0000) ? 1.f : 0.f; hv.w = (h4 >> 24) ? 1.f : 0.f;
            o[(size_t)p * q] = hv;
        }
    }
}

__global__ void __launch_bounds__(256) k_symbol(const uint8_t *__restrict__ screen, const uint8_t *__restrict__ hist, const int32_t *__restrict__ status,
                                                uint32_t *__restrict__ flags, uint32_t *__restrict__ err_any,
                                                int n, int hw, size_t rs, size_t rst, int symbols, int planes_sym, uint32_t sflag, int with_hist, float *__restrict__ out,
                                                const int32_t *__restrict__ ext) {
    const int q = hw >> 2;
    const size_t total = (size_t)n * q;
    const int nplanes = planes_sym + __popc(sflag) + (with_hist ? 1 : 0);  // planes_sym >= symbols: the handle's one-hot depth (groups of a handle may differ)
    const uint32_t symbol_max = (uint32_t)symbols - 1;  // construct_symbol_map fills channels 0..symbols-2 (symbol.rs:51-71)
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        int e = (int)(g / q), i = (int)(g - (size_t)e * q);
        uint32_t s4 = reinterpret_cast<const uint32_t *>(screen + (size_t)e * rs)[i];
        uint32_t a = tile_to_sym(s4 & 0xff), b = tile_to_sym((s4 >> 8) & 0xff), cc = tile_to_sym((s4 >> 16) & 0xff), d = tile_to_sym(s4 >> 24);
        if (a >= symbol_max || b >= symbol_max || cc >= symbol_max || d >= symbol_max) {  // InvalidTileError (e.g. 'Z')
            if (flags) atomicOr(&flags[e], RG_FLAG_ERR_TILE);
            atomicOr(err_any, RG_FLAG_ERR_TILE);
        }
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)(ext ? ext[e] : e) * nplanes * hw) + i;
        for (uint32_t ch = 0; ch < (uint32_t)planes_sym; ch++) {
            float4 v;
            v.x = a == ch ? 1.f : 0.f; v.y = b == ch ? 1.f : 0.f; v.z = cc == ch ? 1.f : 0.f; v.w = d == ch ? 1.f : 0.f;
            if (ch >= symbol_max) v.x = v.y = v.z = v.w = 0.f;
            o[(size_t)ch * q] = v;
        }
        int p = planes_sym;
        for (int bb = 0; bb < 9; bb++)
            if (sflag & (1u << bb)) {
                float f = (float)status[(size_t)e * rst + kStatusIdx[bb]];
                float4 sv; sv.x = sv.y = sv.z = sv.w = f;
                o[(size_t)p * q] = sv;
                p++;
            }
        if (with_hist) {
            uint32_t h4 = reinterpret_cast<const uint32_t *>(hist + (size_t)e * rs)[i];
            float4 hv;
            hv.x = (h4 & 0xff) ? 1.f : 0.f; hv.y = (h4 & 0xff00) ? 1.f : 0.f; hv.z = (h4 & 0xff0000) ? 1.f : 0.f; hv.w = (h4 >> 24) ? 1.f : 0.f;
            o[(size_t)p * q] = hv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_obs: fused mirror refresh + observation encode (the steady-state path: one pass per step)
// ---------------------------------------------------------------------------------------------
// For every env: if the last key produced a Redraw, draw the screen from the tile words (+ entity
// overlays) and refresh the screen / history mirrors; otherwise re-read the 1-byte-per-cell mirror.  The
// screen is staged in LDS, then encoded straight into the caller's f32 tensor with float4 stores.
// HBM traffic per env-step (mini gray): 1 KB tiles (Redraw envs) or 0.5 KB mirror read, 0.5 KB mirror
// write (Redraw envs), 2 KB observation write.  A block of 256 threads serves `epb` envs, `tpe` threads each;
// a thread owns 8 consecutive cells (one 16-byte tile load, two float4 stores per plane).
#define OBS_THREADS 256
// The observation tensor is a write-once 134 MB stream per step (mini gray): non-temporal stores keep it from evicting the
// env state (tile grids, mirrors, tables) that the next k_step re-reads from L2 / Infinity Cache.
__device__ __forceinline__ void store_obs(float4 *p, float4 v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v nv = {v.x, v.y, v.z, v.w};
#ifdef RG_EXP_OBS_PLAIN_STORES
    *reinterpret_cast<f4v *>(p) = nv;  // (experiment build: profiles/r06_experiments.txt)
#else
    __builtin_nontemporal_store(nv, reinterpret_cast<f4v *>(p));
#endif
}
// Per-env overlay inputs staged in LDS, in the layout of the env's OBSERVATION RECORD (RgState::obs_rec, RG_OBS_REC_WORDS): words [0, nr) the monster
// words, [nr] the player's position, [nr + 1, 2 nr + 1) the room rects, then the room metas one byte each.  The record is what an ordinary Redraw reads --
// one 64-byte line per env, written by whoever writes the env's tables (k_step: monsters and player after every turn that redraws; the generator and the
// spare hand-off: the rooms) -- instead of the env's column of four
// [slot][env] tables: 16 lines of 64 bytes for 52 useful ones, 0.43 x 65 536 times per step (28 MB of the pass's 228; measured: 44.7 -> 40.3 us).
// The gold overlay needs no table at all: the tile word carries Floor::items' membership bit (C_GOLD).
struct ObsTabs { uint32_t w[RG_OBS_REC_WORDS(RG_OBS_MAX_ROOMS)]; };
#define OBS_ENV_BYTES(hw) ((((size_t)(hw) + sizeof(ObsTabs)) + 15) & ~(size_t)15)

// LDS-only workgroup barrier: unlike __syncthreads() it does not drain vmcnt, so the prefetched global loads of the next env stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// GROUPS: the batch is one config group of a handle with several (RgState::ext maps its envs to the handle's env order, and the one-hot depth is the
// handle's): compiled separately so that the ordinary kernel carries none of it (its 72 registers = 7 waves per SIMD are what its bandwidth rests on)
// BOUND (rg_obs_bind): `out` is the handle's bound observation tensor and its contents are current up to the last k_step: only the envs that k_step listed
// (RgState::obs_list: final flag word with REDRAW or SCR_CHANGED) are encoded -- work item i is env list[i] -- and their SCR_CHANGED bit is cleared; every other
// env's image is already what a full encode would write.  A separate instance: the ordinary kernel carries none of it.
template <int KIND, bool GROUPS, bool BOUND = false>
__global__ void __launch_bounds__(OBS_THREADS) k_obs(RgState S, RgConfig c, uint32_t sflag, int with_hist, float *__restrict__ out,
                                                    uint32_t *__restrict__ err_any, int tpe, int epb, int planes_sym, int hi_prio) {
    // Wide grids: above the background generator's waves (k_regen, priority 0), which otherwise take issue slots from this bandwidth-bound pass for as long
    // as the two overlap (80x24: 79.9 -> 75.2 us, 222 -> 229 M).  Not where the step kernel is the capped two-waves-per-SIMD instance (W <= 32): generator waves
    // held back here are still resident when the next k_step starts, and there a resident generator wave keeps a step block waiting for its registers
    // (mini: k_obs 46.7 -> 45.6 us but k_step 55.7 -> 60.3 us; profiles/r04_experiments.txt).
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __align__(16) uint8_t smem[];
    float *lutf = reinterpret_cast<float *>(smem);        // glyph -> gray value (KIND 0)
    uint8_t *luts = smem + 512;                            // glyph -> symbol id
    uint8_t *mtile = smem + 512 + 128;                     // monster type -> glyph (RgConfig::mon[].tile: indexed per lane, so not from the kernarg segment)
    uint8_t *envs = smem + 512 + 128 + 64;                 // epb x {HW staged screen bytes, ObsTabs}
    const int tid = threadIdx.x, W = c.width, H = c.height, HW = W * H, Q8 = HW >> 3;
    // (BOUND: the work items are the entries of the last k_step's list)
    const int32_t *list = BOUND ? S.obs_list + (size_t)S.obs_par * S.n : nullptr;
    const int n = BOUND ? (int)S.obs_cnt[S.obs_par] : S.n;
    const int nrooms = c.room_num_x * c.room_num_y;
    const int symbols = c.symbols;
    for (int g = tid; g < 128; g += blockDim.x) {
        uint32_t sy = tile_to_sym((uint32_t)g);
        luts[g] = (uint8_t)sy;
        lutf[g] = (float)(uint8_t)sy / (float)(uint8_t)symbols;  // python/src/lib.rs:84 (same single division)
    }
    for (int g = tid; g < RG_MAX_ENEMY_KINDS + 6; g += blockDim.x) mtile[g] = c.mon[g].tile;
    const int le = tid / tpe, lt = tid - le * tpe;
    const int base_planes = KIND ? (GROUPS ? planes_sym : symbols) : 1;  // planes_sym >= symbols: the handle's one-hot depth (config groups of one handle may differ)
    const int nplanes = base_planes + __popc(sflag) + (with_hist ? 1 : 0);
    uint8_t *scr = envs + (size_t)le * OBS_ENV_BYTES(HW);
    ObsTabs *tb = reinterpret_cast<ObsTabs *>(scr + HW);
    // Persistent blocks, software-pipelined two envs deep: the flag word of the env after next and -- now that its flag word is known -- the
    // inputs of the next env (tile quad + entity tables if it redraws, its screen mirror if not) are requested before the current env is
    // encoded, so an env costs no exposed round trip and nothing is fetched that is not used.
    struct Pre { uint4 v0; uint32_t rec; };  // (`rec`: the lane's word of the env's observation record)
    const int rec_words = RG_OBS_REC_WORDS(nrooms);
    const uint32_t *rec_all = S.obs_rec;  // (rgk_obs: rec_words <= tpe, one word per thread)
    const int stride = gridDim.x * epb;
    // An env WITHOUT a pending Redraw -- nine in ten since the turn keeps the mirror itself -- is a plain stream: mirror word -> table -> float4.  With one env per
    // block (always) and at most two words per thread, such an env takes no LDS staging and none of the four barriers of the Redraw path: its words go from
    // the prefetch registers straight to the stores (round 6: the pass was latency-bound per block -- LDS round trips and barriers between the load and the
    // store of every env -- at 5.0 TB/s where a plain fill of the same 134 MB runs at 6.5).
    const int Q4 = HW >> 2;
    // Up to four words per thread: on the 32x16 grid a thread carries 16 cells and ONE wave serves TWO envs per iteration (tpe = 32, epb = 2: rgk_obs) -- twice the
    // bytes in flight per wave of a pass whose blocks are latency-bound (one env per memory round trip).  Two envs per block only where the block is a single wave:
    // its halves then take the two paths under lane masks and the staged path's barriers are the wave's own.
    const bool one_wave = blockDim.x == WAVE;
    const bool fast_ok = (epb == 1 || one_wave) && Q4 <= 4 * tpe && sflag == 0 && !with_hist;  // (status / history planes: the general path)
    // item -> env: the identity, or (BOUND) the list entry -- one more dependent load, fetched one iteration earlier than the flag word
    auto load_env = [&](int base) -> int {
        const int i = base + le;
        if (!BOUND) return i;
        return (le < epb && i < n) ? list[i] : 0;
    };
    auto load_flag = [&](int base, int e) -> uint32_t {
        return (le < epb && base + le < n) ? S.flags[e] : 0u;
    };
    auto prefetch = [&](int base, int e, uint32_t fl) {
        Pre p; p.v0 = make_uint4(0, 0, 0, 0); p.rec = 0;
        if (le < epb && base + le < n) {
            if (fl & RG_FLAG_REDRAW) {
                if (lt < Q8) p.v0 = reinterpret_cast<const uint4 *>(S.cell + (size_t)e * HW)[lt];
                if (lt < rec_words) p.rec = rec_all[(size_t)e * rec_words + lt];
            } else if (fast_ok) {  // the words this thread ENCODES (phase C's layout: words lt, lt + tpe, ...), see the fast path below
                const uint32_t *m4 = reinterpret_cast<const uint32_t *>(S.screen + (size_t)e * HW);
                if (lt < Q4) p.v0.x = m4[lt];
                if (lt + tpe < Q4) p.v0.y = m4[lt + tpe];
                if (lt + 2 * tpe < Q4) p.v0.z = m4[lt + 2 * tpe];
                if (lt + 3 * tpe < Q4) p.v0.w = m4[lt + 3 * tpe];
            } else if (lt < Q8) {
                const uint2 m = reinterpret_cast<const uint2 *>(S.screen + (size_t)e * HW)[lt];
                p.v0.x = m.x; p.v0.y = m.y;
            }
        }
        return p;
    };
    const uint32_t smax = (uint32_t)symbols - 1;  // construct_symbol_map fills channels 0..symbols-2 (symbol.rs:51-71)
    // one word (4 cells) of an env's image(s): glyph planes, status planes, history plane -- whole-line float4 stores (lanes contiguous in q)
    auto emit_glyphs = [&](float4 *o, int q, uint32_t g, bool &bad) {
        const int q4 = Q4;
        const uint32_t b0 = g & 0x7f, b1 = (g >> 8) & 0x7f, b2 = (g >> 16) & 0x7f, b3 = g >> 24;
        if (KIND == 0) {
            float4 v; v.x = lutf[b0]; v.y = lutf[b1]; v.z = lutf[b2]; v.w = lutf[b3];
            store_obs(&o[q], v);
        } else {
            const uint32_t s0 = luts[b0], s1 = luts[b1], s2 = luts[b2], s3 = luts[b3];
            bad = bad || s0 >= smax || s1 >= smax || s2 >= smax || s3 >= smax;
            for (uint32_t ch = 0; ch < smax; ch++) {
                float4 v;
                v.x = s0 == ch ? 1.f : 0.f; v.y = s1 == ch ? 1.f : 0.f; v.z = s2 == ch ? 1.f : 0.f; v.w = s3 == ch ? 1.f : 0.f;
                store_obs(&o[(size_t)ch * q4 + q], v);
            }
            float4 z; z.x = z.y = z.z = z.w = 0.f;
            for (uint32_t ch = smax; ch < (uint32_t)base_planes; ch++) store_obs(&o[(size_t)ch * q4 + q], z);  // the last channel is never set
        }
    };
    auto emit = [&](float4 *o, int q, uint32_t g, const float *stf, int nst, const uint32_t *hist4, bool &bad) {
        const int q4 = Q4;
        emit_glyphs(o, q, g, bad);
        int p = base_planes;
        for (int b = 0; b < nst; b++, p++) {
            float4 sv; sv.x = sv.y = sv.z = sv.w = stf[b];
            store_obs(&o[(size_t)p * q4 + q], sv);
        }
        if (with_hist) {
            const uint32_t h4 = hist4[q];  // (a redrawn env: written in phase A by this block, barrier in between)
            float4 a;
            a.x = (h4 & 0xff) ? 1.f : 0.f; a.y = (h4 & 0xff00) ? 1.f : 0.f; a.z = (h4 & 0xff0000) ? 1.f : 0.f; a.w = (h4 >> 24) ? 1.f : 0.f;
            store_obs(&o[(size_t)p * q4 + q], a);
        }
    };
    lds_barrier();  // the tables are ready (the fast path below has no barrier of its own)
    const int base0 = blockIdx.x * epb;
    int e_cur = load_env(base0), e_nxt = load_env(base0 + stride), e_nn = load_env(base0 + 2 * stride);
    uint32_t fl_cur = load_flag(base0, e_cur), fl_nxt = load_flag(base0 + stride, e_nxt);
    Pre nxt = prefetch(base0, e_cur, fl_cur);
    for (int base = base0; base < n; base += stride) {
        const int e = e_cur;
        const bool valid = le < epb && base + le < n;
        const Pre cur = nxt;
        const uint32_t fl = fl_cur;
        e_cur = e_nxt; e_nxt = e_nn;
        fl_cur = fl_nxt;
        fl_nxt = load_flag(base + 2 * stride, e_nxt);
        e_nn = load_env(base + 3 * stride);
        if (base + stride < n) nxt = prefetch(base + stride, e_cur, fl_cur);
        const uint4 *cell4 = reinterpret_cast<const uint4 *>(S.cell + (size_t)e * HW);
        const uint4 v0 = cur.v0;
        const uint32_t t_rec = cur.rec;
        // the register-to-store path of an env without a pending Redraw (a half-wave of a one-wave block, else the whole block: one env per block there)
        const bool my_fast = fast_ok && valid && !(fl & RG_FLAG_REDRAW);
        if (my_fast) {
            const int xe = GROUPS ? __builtin_amdgcn_readfirstlane(S.ext[e]) : e;
            float4 *o = reinterpret_cast<float4 *>(out + (size_t)xe * nplanes * HW);
            bool bad = false;
            if (lt < Q4) emit_glyphs(o, lt, v0.x & 0x7f7f7f7fu, bad);
            if (lt + tpe < Q4) emit_glyphs(o, lt + tpe, v0.y & 0x7f7f7f7fu, bad);
            if (lt + 2 * tpe < Q4) emit_glyphs(o, lt + 2 * tpe, v0.z & 0x7f7f7f7fu, bad);
            if (lt + 3 * tpe < Q4) emit_glyphs(o, lt + 3 * tpe, v0.w & 0x7f7f7f7fu, bad);
            if (KIND == 1 && bad) { atomicOr(&S.flags[e], RG_FLAG_ERR_TILE); atomicOr(err_any, RG_FLAG_ERR_TILE); }
            if (BOUND && lt == 0) atomicAnd(&S.flags[e], ~RG_FLAG_SCR_CHANGED);
        }
        const bool staged = valid && !my_fast;  // the lanes that take the staged path below
        if (fast_ok && !(one_wave ? __any(staged) : staged)) continue;  // (block-uniform: a one-wave block votes; several waves serve ONE env)
        const bool redraw = staged && (fl & RG_FLAG_REDRAW);
        lds_barrier();  // previous iteration's LDS reads done
        if (staged) {
            if (redraw) {
                if (lt < rec_words) tb->w[lt] = t_rec;
                // the history plane is rewritten only when the visited set changed since it was last written (k_step: HIST_DIRTY), never on a
                // stale Redraw
                const bool upd_hist = !(fl & RG_FLAG_HIST_STALE) && (fl & RG_FLAG_HIST_DIRTY);
                uint2 *hist8 = reinterpret_cast<uint2 *>(S.hist + (size_t)e * HW);
                for (int i = lt; i < Q8; i += tpe) {
                    uint4 v = i == lt ? v0 : cell4[i];
                    uint32_t q[4] = {v.x, v.y, v.z, v.w};
                    uint32_t g[2] = {0, 0}, hb[2] = {0, 0};
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        uint32_t cw = (q[t >> 1] >> ((t & 1) * 16)) & 0xffff;
                        int idx = i * 8 + t;
                        bool inner = idx >= W && idx < HW - W;  // rows 1..H-2 only (rogue/mod.rs:278-290)
                        uint32_t gl = ' ';
                        if (inner && (cw & C_VISIBLE)) gl = glyph_of(cw);
                        if (inner && (cw & (C_VISIBLE | C_DRAWN))) gl = ((cw & C_GOLD) ? (uint32_t)'*' : gl) | 0x80u;  // bit 7: an object on this cell is drawn (draw_ranges);
                                                                                                                  // gold is drawn over a monster, under the player
                        g[t >> 2] |= gl << ((t & 3) * 8);
                        hb[t >> 2] |= ((cw & C_VISITED) ? 1u : 0u) << ((t & 3) * 8);
                    }
                    reinterpret_cast<uint2 *>(scr)[i] = make_uint2(g[0], g[1]);
                    if (upd_hist) hist8[i] = make_uint2(hb[0], hb[1]);
                }
            } else {
                const uint2 *m8 = reinterpret_cast<const uint2 *>(S.screen + (size_t)e * HW);
                for (int i = lt; i < Q8; i += tpe) reinterpret_cast<uint2 *>(scr)[i] = i == lt ? make_uint2(v0.x, v0.y) : m8[i];
            }
        }
        lds_barrier();
        // ---- phase B: entity overlays from LDS only; draw priority monster < gold < player (core/src/lib.rs:271-283): a monster never replaces the
        //      '*' the decode put there (no other glyph is '*'), the player replaces anything ----
        const uint32_t ppos = redraw ? tb->w[nrooms] : 0;
        const int px = POS_X(ppos), py = POS_Y(ppos);
        if (redraw && lt < nrooms) {
            uint32_t w = tb->w[lt];
            if ((w >> 24) & MF_ALIVE) {
                int x = POS_X(w), y = POS_Y(w);
                int dx = px - x, dy = py - y;
                bool show = dx * dx + dy * dy <= 2;
                if (!show) {  // Floor::in_same_room (floor.rs:381-393)
                    int id = room_id_of(c, px, py);
                    if (id >= 0 && room_id_of(c, x, y) == id) {
                        if ((reinterpret_cast<const uint8_t *>(&tb->w[2 * nrooms + 1])[id] & RM_KIND_MASK) == RK_EMPTY) show = true;
                        else {
                            int x0, y0, x1, y1;
                            unpack_rect(tb->w[nrooms + 1 + id], x0, y0, x1, y1);
                            bool ina = px >= x0 && px < x1 && py >= y0 && py < y1, inb = x >= x0 && x < x1 && y >= y0 && y < y1;
                            show = ina == inb;
                        }
                    }
                }
                const uint32_t under = scr[y * W + x];
                if (show && (under & 0x80u) && under != (0x80u | '*')) scr[y * W + x] = (uint8_t)(0x80u | mtile[(w >> 16) & 0xff]);
            }
        }
        lds_barrier();
        if (redraw && lt == 0 && (scr[py * W + px] & 0x80u)) scr[py * W + px] = (uint8_t)(0x80u | '@');
        lds_barrier();
        if (with_hist) __syncthreads();  // the history plane is re-read from global memory below (written in phase A by other lanes)
        // ---- phase C: mirror write-back + encode.  One float4 (4 cells) per lane per plane, lanes contiguous: every wave-level store
        //      covers whole 128-byte lines (1 KB per instruction) ----
        if (staged) {
            uint32_t *m4 = reinterpret_cast<uint32_t *>(S.screen + (size_t)e * HW);
            const uint32_t *scr4 = reinterpret_cast<const uint32_t *>(scr);
            const uint32_t *hist4 = reinterpret_cast<const uint32_t *>(S.hist + (size_t)e * HW);
            const int xe = GROUPS ? __builtin_amdgcn_readfirstlane(S.ext[e]) : e;  // config-group handles write at the handle's env index (one env per block: uniform)
            float4 *o = reinterpret_cast<float4 *>(out + (size_t)xe * nplanes * HW);
            float stf[9];
            int nst = 0;
            for (int b = 0; b < 9; b++)
                if (sflag & (1u << b)) stf[nst++] = (float)S.status[(size_t)e * 10 + kStatusIdx[b]];
            bool bad = false;
            for (int q = lt; q < Q4; q += tpe) {
                const uint32_t g = scr4[q] & 0x7f7f7f7fu;
                if (redraw) m4[q] = g;
                emit(o, q, g, stf, nst, hist4, bad);
            }
            if (KIND == 1 && bad) { atomicOr(&S.flags[e], RG_FLAG_ERR_TILE); atomicOr(err_any, RG_FLAG_ERR_TILE); }
            const uint32_t seen = BOUND ? RG_FLAG_SCR_CHANGED : 0u;  // (the bound tensor now shows this env's screen)
            if (redraw && lt == 0)  // a stale Redraw leaves the history mirror one level behind (k_step refreshes it before the next descent)
                S.flags[e] = (fl & ~(seen | RG_FLAG_REDRAW | RG_FLAG_HIST_STALE | RG_FLAG_HIST_LAG | ((fl & RG_FLAG_HIST_STALE) ? 0u : RG_FLAG_HIST_DIRTY))) |
                             ((fl & RG_FLAG_HIST_STALE) ? RG_FLAG_HIST_LAG : 0u) | (KIND == 1 && bad ? RG_FLAG_ERR_TILE : 0);
            else if (BOUND && !redraw && lt == 0) atomicAnd(&S.flags[e], ~RG_FLAG_SCR_CHANGED);  // (atomic: the one-hot kind ORs its error bit into the same word)
        }
    }
}

// one 4-byte word of the compact record per thread (hw % 4 == 0: every record section is word-aligned).  Record = {screen u8[hw], status i32[10],
// reward f32, flags u32 (the public bits: terminal, dead, message flags, error bits -- not the mirror bookkeeping), hist u8[hw] if with_hist}:
// everything ThreadConductor::step hands back per env in one reply (state AND terminal flag, python/src/thread_impls.rs:61-81; parallel.py:59-64
// derives reward and done from exactly that), so the one collective of the sharded path carries the learner's whole step
#define RG_PUBLIC_FLAGS (RG_FLAG_TERMINAL | RG_FLAG_DEAD | RG_FLAG_MSG_MASK | RG_FLAG_ERR_MASK)
__global__ void __launch_bounds__(256) k_pack(const uint8_t *__restrict__ screen, const uint8_t *__restrict__ hist, const int32_t *__restrict__ status,
                                              const float *__restrict__ reward, const uint32_t *__restrict__ flags, int n, int hw, int with_hist, uint32_t *__restrict__ out) {
    const int qs = hw >> 2, qr = qs + 12 + (with_hist ? qs : 0);
    const size_t total = (size_t)n * qr;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(g / qr), i = (int)(g - (size_t)e * qr);
        uint32_t v;
        if (i < qs) v = reinterpret_cast<const uint32_t *>(screen + (size_t)e * hw)[i];
        else if (i < qs + 10) v = (uint32_t)status[(size_t)e * 10 + (i - qs)];
        else if (i == qs + 10) v = __float_as_uint(reward[e]);
        else if (i == qs + 11) v = flags[e] & RG_PUBLIC_FLAGS;
        else v = reinterpret_cast<const uint32_t *>(hist + (size_t)e * hw)[i - qs - 12];
        out[g] = v;
    }
}

// rg_step_fetch: everything ParallelGameState::step hands back, written by ONE launch to wherever the caller wants it -- pinned host memory (the kernel's
// stores cross PCIe themselves: no copy engine, no staging) or a device snapshot for the screens.  4-byte words, grid-stride: [screen][hist][status][flags],
// and the error word (read and cleared: what rg_sync does with a 4-byte copy and a memset).
__global__ void __launch_bounds__(256) k_export(const uint32_t *__restrict__ screen, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ status,
                                                const uint32_t *__restrict__ flags, uint32_t *__restrict__ err_any, size_t w_scr, size_t w_status, size_t w_flags,
                                                uint32_t *__restrict__ o_screen, uint32_t *__restrict__ o_hist, uint32_t *__restrict__ o_status,
                                                uint32_t *__restrict__ o_flags, uint32_t *__restrict__ o_err) {
    const size_t total = (o_screen ? 2 * w_scr : 0) + w_status + w_flags;
    const size_t base = o_screen ? 2 * w_scr : 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        if (g < base) { if (g < w_scr) o_screen[g] = screen[g]; else o_hist[g - w_scr] = hist[g - w_scr]; }
        else if (g < base + w_status) o_status[g - base] = status[g - base];
        else o_flags[g - base - w_status] = flags[g - base - w_status];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (an exchange: the background producers OR their error bits into this word while this kernel runs, and a bit raised between a read and a clear
        // would never be reported)
        const uint32_t e = atomicExch(err_any, 0u);
        *o_err = e;
    }
}
__global__ void __launch_bounds__(256) k_scatter_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int32_t *__restrict__ ext, int n, int row_bytes) {
    if ((row_bytes & 3) == 0) {
        const int rw = row_bytes >> 2;
        const size_t total = (size_t)n * rw;
        for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
            const int e = (int)(g / rw), i = (int)(g - (size_t)e * rw);
            reinterpret_cast<uint32_t *>(dst)[(size_t)ext[e] * rw + i] = reinterpret_cast<const uint32_t *>(src)[g];
        }
    } else {
        const size_t total = (size_t)n * row_bytes;
        for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
            const int e = (int)(g / row_bytes), i = (int)(g - (size_t)e * row_bytes);
            dst[(size_t)ext[e] * row_bytes + i] = src[g];
        }
    }
}
__global__ void __launch_bounds__(256) k_gather_keys(const uint8_t *__restrict__ keys, const int32_t *__restrict__ ext, uint8_t *__restrict__ dst, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] = keys[ext[e]];
}

__global__ void k_probe_clock(unsigned long long *out, int spin) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = threadIdx.x + 1u;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;  // dependent VALU chain: 2 instructions per iteration
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = x; out[3] = (unsigned long long)spin; }
}

// scalar fallbacks for H*W not divisible by 4 (never the case for the benchmark sizes)
__global__ void __launch_bounds__(256) k_encode_scalar(const uint8_t *__restrict__ screen, const uint8_t *__restrict__ hist, const int32_t *__restrict__ status,
                                                       uint32_t *__restrict__ flags, uint32_t *__restrict__ err_any, int n, int hw, size_t rs, size_t rst, int symbols,
                                                       int planes_sym, uint32_t sflag, int with_hist, int kind, float *__restrict__ out, const int32_t *__restrict__ ext) {
    const size_t total = (size_t)n * hw;
    const int base = kind ? planes_sym : 1;
    const int nplanes = base + __popc(sflag) + (with_hist ? 1 : 0);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        int e = (int)(g / hw), i = (int)(g - (size_t)e * hw);
        uint32_t sym = tile_to_sym(screen[(size_t)e * rs + i]);
        float *o = out + (size_t)(ext ? ext[e] : e) * nplanes * hw + i;
        if (!kind) o[0] = (float)(uint8_t)sym / (float)(uint8_t)symbols;
        else {
            if (sym >= (uint32_t)symbols - 1) { if (flags) atomicOr(&flags[e], RG_FLAG_ERR_TILE); atomicOr(err_any, RG_FLAG_ERR_TILE); }
            for (int ch = 0; ch < planes_sym; ch++) o[(size_t)ch * hw] = (sym == (uint32_t)ch && ch < symbols - 1) ? 1.f : 0.f;
        }
        int p = base;
        for (int b = 0; b < 9; b++)
            if (sflag & (1u << b)) { o[(size_t)p * hw] = (float)status[(size_t)e * rst + kStatusIdx[b]]; p++; }
        if (with_hist) o[(size_t)p * hw] = hist[(size_t)e * rs + i] ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (used by rg_api.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" {
void rgk_render(const RgState *S, const RgConfig *c, hipStream_t st) {
    int blocks = S->n < 8192 ? S->n : 8192;
    hipLaunchKernelGGL(k_render, dim3(blocks), dim3(RENDER_THREADS), 0, st, *S, *c);
}
// fused mirror refresh + encode; returns 0 if the geometry is not supported (caller falls back to k_render + encode)
int rgk_obs(const RgState *S, const RgConfig *c, uint32_t sflag, int with_hist, int kind, float *out, uint32_t *err_any, int planes_sym, int bound, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    int hw = c->width * c->height;
    if (hw & 7) return 0;
    if (c->room_num_x * c->room_num_y > RG_OBS_MAX_ROOMS || !S->obs_rec) return 0;  // the fused kernel's LDS overlay tables hold 64 rooms (one thread per room + the player): unfused path
    int q8 = hw / 8;
    int tpe = q8 >= OBS_THREADS ? OBS_THREADS : ((q8 + 63) / 64) * 64;  // threads per env: a whole number of waves
    if (tpe > OBS_THREADS) tpe = OBS_THREADS;
    if (RG_OBS_REC_WORDS(c->room_num_x * c->room_num_y) > tpe) return 0;  // the env's observation record is fetched one word per thread (a <= 512-cell grid with more than 28 rooms: unfused path)
    const int bthreads = tpe, epb = 1;  // one env per block: no cross-env barrier coupling (4 envs per 256-thread block measured 10-20 % slower; two envs per one-wave
                                        // block on the 32x16 grid -- half a wave and four words per thread each, which k_obs supports -- 30.0 against 28.9 us: round 6)
    const bool groups = S->ext != nullptr;
    size_t smem = 512 + 128 + 64 + (size_t)epb * OBS_ENV_BYTES(hw);
    int blocks = (S->n + epb - 1) / epb;
    // persistent grid: launching one tiny workgroup per env is dispatch-rate bound (65 536 one-wave blocks: 71 us; 16 384 looping blocks: 51 us)
    {
        int cap = bthreads <= 64 ? 16384 : 8192;
#ifdef RG_DEV_KNOBS
        if (const char *ev = getenv("RG_OBS_BLOCKS")) cap = atoi(ev);
#endif
        if (blocks > cap) blocks = cap;
    }
    const int hi_prio = !(c->width <= 32 && c->room_num_x * c->room_num_y <= 32);  // (rg_kernels.hip rgk_step: those configs step with k_step_w32)
#define RG_LAUNCH_OBS(...) do { if (ev0 || ev1) hipExtLaunchKernelGGL((__VA_ARGS__), dim3(blocks), dim3(bthreads), (uint32_t)smem, st, ev0, ev1, 0, *S, *c, sflag, with_hist, out, err_any, tpe, epb, planes_sym, hi_prio); \
                               else hipLaunchKernelGGL((__VA_ARGS__), dim3(blocks), dim3(bthreads), smem, st, *S, *c, sflag, with_hist, out, err_any, tpe, epb, planes_sym, hi_prio); } while (0)
    if (bound && !groups && !kind) RG_LAUNCH_OBS(k_obs<0, false, true>);       // (rg_obs_bind: the in-place pass over the last k_step's list)
    else if (bound && !groups) RG_LAUNCH_OBS(k_obs<1, false, true>);
    else if (!kind && !groups) RG_LAUNCH_OBS(k_obs<0, false>);
    else if (!kind) RG_LAUNCH_OBS(k_obs<0, true>);
    else if (!groups) RG_LAUNCH_OBS(k_obs<1, false>);
    else RG_LAUNCH_OBS(k_obs<1, true>);
#undef RG_LAUNCH_OBS
    return 1;
}
void rgk_encode(const uint8_t *screen, const uint8_t *hist, const int32_t *status, uint32_t *flags, uint32_t *err_any, int n, int hw, size_t rs, size_t rst,
                int symbols, int planes_sym, uint32_t sflag, int with_hist, int kind, float *out, const int32_t *ext, hipStream_t st) {
    if ((hw & 3) == 0 && (rs & 3) == 0) {
        size_t total = (size_t)n * (hw >> 2);
        int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        if (blocks < 1) blocks = 1;
        if (!kind) hipLaunchKernelGGL(k_gray, dim3(blocks), dim3(256), 0, st, screen, hist, status, n, hw, rs, rst, symbols, sflag, with_hist, out, ext);
        else hipLaunchKernelGGL(k_symbol, dim3(blocks), dim3(256), 0, st, screen, hist, status, flags, err_any, n, hw, rs, rst, symbols, planes_sym, sflag, with_hist, out, ext);
    } else {
        size_t total = (size_t)n * hw;
        int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(k_encode_scalar, dim3(blocks), dim3(256), 0, st, screen, hist, status, flags, err_any, n, hw, rs, rst, symbols, planes_sym, sflag, with_hist, kind, out, ext);
    }
}
void rgk_export(const RgState *S, uint32_t *err_any, void *o_screen, void *o_hist, void *o_status, void *o_flags, uint32_t *o_err, hipStream_t st) {
    const size_t n = (size_t)S->n, w_scr = n * (size_t)S->hw / 4, w_status = n * 10, w_flags = n;
    const size_t total = (o_screen ? 2 * w_scr : 0) + w_status + w_flags;
    int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_export, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(S->screen), reinterpret_cast<const uint32_t *>(S->hist),
                       reinterpret_cast<const uint32_t *>(S->status), S->flags, err_any, w_scr, w_status, w_flags, static_cast<uint32_t *>(o_screen), static_cast<uint32_t *>(o_hist),
                       static_cast<uint32_t *>(o_status), static_cast<uint32_t *>(o_flags), o_err);
}
// compact record of every env: {screen u8[hw], status i32[10], reward f32, flags u32, hist u8[hw] (optional)}, back to back -- the payload of the ONE
// all-gather per step of the multi-GPU path; expanded on the consumer by rgk_encode with rs = record size
void rgk_pack(const RgState *S, int with_hist, uint8_t *out, hipStream_t st) {
    const int hw = S->hw;
    const size_t rec = (size_t)hw + RG_COMPACT_FIXED_BYTES + (with_hist ? hw : 0);
    const size_t total = (size_t)S->n * (rec / 4);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, st, S->screen, S->hist, S->status, S->reward, S->flags, S->n, hw, with_hist, reinterpret_cast<uint32_t *>(out));
}
// handle with several config groups: rows of a group's array -> the handle's array at the group's env indices (row_words 4-byte words per env)
void rgk_scatter_rows(const void *src, void *dst, const int32_t *ext, int n, int row_bytes, hipStream_t st) {
    const size_t total = (size_t)n * row_bytes;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_scatter_rows, dim3(blocks), dim3(256), 0, st, (const uint8_t *)src, (uint8_t *)dst, ext, n, row_bytes);
}
void rgk_gather_keys(const uint8_t *keys, const int32_t *ext, uint8_t *dst, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_keys, dim3((n + 255) / 256), dim3(256), 0, st, keys, ext, dst, n);
}
// shader-clock probe: one wave spins for `spin` iterations and reports {s_memtime ticks (shader clock), s_memrealtime ticks (constant 100 MHz)}
void rgk_probe_clock(unsigned long long *out, int spin, hipStream_t st) { hipLaunchKernelGGL(k_probe_clock, dim3(1), dim3(64), 0, st, out, spin); }
}
