// rg_api.cpp -- C-ABI host of librogue_gym_hip.so (see include/rogue_gym_hip.h).
// Owns the HBM state, parses configs, sequences the kernels on one HIP stream.  There is no CPU
// fallback: without a HIP device every entry point that needs one fails loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../../include/rogue_gym_hip.h"
#include "rg_state.h"

// The few RCCL declarations this file needs, spelled out: librccl is bound with dlopen at run time, so building the single-GPU library must not
// need the RCCL development headers either.  (ABI of nccl.h / rccl.h 2.x: ncclUniqueId = 128 opaque bytes passed by value, ncclComm_t an opaque
// pointer, ncclSuccess = 0, ncclUint8 = 1.)
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclUint8 = 1;

extern "C" {
void rgk_build(const RgState *S, const RgConfig *c, hipStream_t st);
int rgk_step(const RgState *S, const RgState *SP_dev, const RgConfig *c, const uint8_t *keys, int use_spares, int parity, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
void rgk_probe_clock(unsigned long long *out, int spin, hipStream_t st);
void rgk_regen(const RgState *SP, const RgConfig *c, int bulk, int spares, const uint32_t *mark, uint32_t target, uint32_t *err_any, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
void rgk_regen_gate(const uint32_t *mark, uint32_t target, uint32_t *err_any, hipStream_t st);
void rgk_export(const RgState *S, uint32_t *err_any, void *o_screen, void *o_hist, void *o_status, void *o_flags, uint32_t *o_err, hipStream_t st);
int rgk_regen_lanes_supported(const RgConfig *c, int maze_cap);
int rgk_step_bound_capable(const RgConfig *c);
int rgk_regen_lanes(const RgState *SP, const RgConfig *c, uint32_t *q, int32_t *list, int bulk, int waves, int slots, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
void rgk_debug_descend(const RgState *S, const RgConfig *c, hipStream_t st);
void rgk_render(const RgState *S, const RgConfig *c, hipStream_t st);
void rgk_encode(const uint8_t *screen, const uint8_t *hist, const int32_t *status, uint32_t *flags, uint32_t *err_any, int n, int hw, size_t rs, size_t rst,
                int symbols, int planes_sym, uint32_t sflag, int with_hist, int kind, float *out, const int32_t *ext, hipStream_t st);
void rgk_pack(const RgState *S, int with_hist, uint8_t *out, hipStream_t st);
void rgk_scatter_rows(const void *src, void *dst, const int32_t *ext, int n, int row_bytes, hipStream_t st);
void rgk_gather_keys(const uint8_t *keys, const int32_t *ext, uint8_t *dst, int n, hipStream_t st);
int rgk_obs(const RgState *S, const RgConfig *c, uint32_t sflag, int with_hist, int kind, float *out, uint32_t *err_any, int planes_sym, int bound, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
}

#define RG_TIMED_KERNELS 5   // k_step, k_render, k_obs (or the unfused encode), k_build, k_regen
struct rg_handle {
    RgParsed parsed;             // config of env 0 (all envs agree except for the seed)
    RgConfig cfg;
    RgState S;
    RgState SP;                  // spare view: core pointers address the pre-generated next level-1 state (k_regen)
    bool spares = false;
    uint32_t *lane_q = nullptr; int32_t *lane_list = nullptr;  // its claim list (one launch at a time: the generator's stream serialises them)
    hipStream_t side3 = nullptr; // ... alternating with side2 (a launch may outlast a step)
    hipStream_t side2 = nullptr; // stream of the level-per-lane spare producer (LOW priority; its launches are long and far apart, the next-level structures' short and with every step)
    int lane_flip = 0;           // which of the two carries the next launch (each with its own claim list)
    uint64_t lane_last = 0;      // step_count of its last launch
    uint8_t *pin_keys = nullptr;   // rg_step_fetch: the keys of the call in pinned, device-visible host memory (k_step reads them across PCIe: no copy call)
    uint32_t *pin_err = nullptr;   // ... and where k_export leaves the error word
    bool lane_regen = false;     // the consumed spares are rebuilt one level per LANE (rg_regen_lanes.hip) instead of one per wave (k_regen); ROGUE_GYM_HIP_WAVE_REGEN=1 keeps the latter
    uint64_t step_count = 0;
    hipStream_t side = nullptr;  // stream of the background generator (LOW priority: the step kernel's blocks are placed first, k_regen takes what is left; rg_step_prefix)
    int regen_idle_after = -1;   // ROGUE_GYM_HIP_KEEP_SPARES with fixed seeds only: > 0 = that many more k_regen launches (after creation / rg_seed), 0 = none needed, -1 = off
    int regen_bulk = 0;          // that many of the next k_regen launches rebuild EVERY consumed spare they find (after rg_seed dropped them all), not one per wave
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    uint32_t *d_err = nullptr;
    uint8_t *d_keys = nullptr;
    bool render_pending = false;
    // rg_obs_bind: the caller's standing observation tensor and whether its contents are the current screens of every env up to the SCR_CHANGED / REDRAW flags
    float *bound_out = nullptr; int bound_kind = 0; bool bound_valid = false;
    int32_t *obs_list_mem = nullptr; uint32_t *obs_cnt_mem = nullptr; float *gray_lut_mem = nullptr;
    int bound_steps = 0;   // k_step launches since the bound tensor was last written: its in-place pass works from the list of exactly ONE
    int stair_gen = 0;           // producers of the stair set launched so far (k_build, k_step, the debug descent; rg_state.h)
    float *obs_scratch = nullptr;  // rg_obs_host: device-side observation buffer, kept between calls
    size_t obs_scratch_cap = 0;
    std::vector<uint64_t> seed_lo, seed_hi;  // host copy of the seeds the next reset will use (reseed envs: the base of the per-build hash)
    std::vector<uint8_t> reseed;             // 0 = configured seed, 1 = fresh seed per build, 2 = fresh seed inside seed_range per build
    std::vector<uint64_t> range_lo, range_span;  // [2][n] low / high words; empty if no env has a seed_range
    unsigned long long *d_probe = nullptr;
    // ---- per-env configs that differ in more than the seed (python/src/lib.rs:270-294 takes one GameConfig per env): the handle is then a
    // PARENT over one sub-handle per distinct config ("group").  Every group is an ordinary homogeneous batch with its own state and
    // kernels; the parent owns the mirrors in the handle's env order (S.screen .. S.done, assembled from the groups') and routes the calls.
    std::vector<rg_handle *> sub;      // empty for an ordinary handle
    std::vector<int> g_of, l_of;       // env -> (group, index inside the group); groups keep the env order
    int32_t *d_ext = nullptr;          // (sub-handle) device copy of its local -> handle env index map
    std::vector<int32_t> ext;          // (sub-handle) the same on the host
    uint8_t *d_keys_sub = nullptr;     // (sub-handle) keys of its envs gathered from the handle's key vector
    int planes_sym = 0;                // one-hot depth of the handle's symbol image (= symbols of env 0's config, like ParallelGameState::symbols)
    bool screens_stale = true;         // (parent) S.screen / S.hist / S.flags need assembling
    bool mixed = false;                // (parent) the groups differ in width / height (python/src/lib.rs:270-294 takes ANY GameConfig per env): there is no
                                       // [n_env][H][W] tensor then -- screens travel as ragged host copies (rg_fetch_states), the tensor entry points refuse
    std::vector<size_t> ragged_off;    // (mixed parent) [n + 1] byte offset of env i's H_i * W_i screen in the ragged env-order layout
    size_t stat_rows = 0;        // workload counters: one row of 8 per k_step block (summed by rg_counters)
    RgState *d_SP = nullptr;     // device-resident copy of SP: k_step reads the spare's pointers from it on the rare take path (one kernel argument
                                 // instead of a second 60-pointer struct in SGPRs)
    ncclComm_t comm = nullptr;   // rg_comm_init: the RCCL communicator of the one collective of the sharded path (rg_allgather_compact)
    int comm_rank = 0, comm_world = 1;
    std::string err;
    // per-kernel HIP-event timing (rg_timing_*)
    bool timing = false;
    std::vector<hipEvent_t> ev[RG_TIMED_KERNELS];   // start/stop pairs
    size_t ev_used[RG_TIMED_KERNELS] = {0, 0, 0, 0, 0};
    uint64_t timing_seq[RG_TIMED_KERNELS] = {0, 0, 0, 0, 0};   // launches seen while timing is on (sampled or not)
    uint64_t timing_stride = 1;
};

#define RG_TIMING_MAX 4096
struct TimedLaunch {  // times one kernel launch with an event pair when timing is on
    // Two forms.  bracket(): hipEventRecord before and after the launch -- the pair then also contains the marker packets' own processing and the
    // gap to the neighbouring kernels (~5-10 us around an 80 us kernel).  ext(): the pair is handed to hipExtLaunchKernelGGL, which stamps it with the
    // dispatch's own begin / end -- the same clock rocprofv3 --kernel-trace reports; used for the two kernels of the step.
    rg_handle *h; int k; bool on, ext_;
    TimedLaunch(rg_handle *h_, int k_, bool ext = false) : h(h_), k(k_), on(false), ext_(ext) {
        // sampled: an event pair costs a few us of stream time, so only every `timing_stride`-th launch of a kernel is timed
        // (the sampled launch of every stride is its MIDDLE one, not its first: the first launch behind a synchronize runs on an idle, cold chip -- ~110 us
        // for a 65 us k_step -- and with 4 samples in the driver's 20-step window that one outlier made the reported average 80 us)
        if (h->timing && (h->timing_seq[k]++ % h->timing_stride) == h->timing_stride / 2 && h->ev_used[k] + 2 <= h->ev[k].size()) {
            on = true;
            if (!ext_) (void)hipEventRecord(h->ev[k][h->ev_used[k]], h->stream);
        }
    }
    hipEvent_t start_ev() const { return on && ext_ ? h->ev[k][h->ev_used[k]] : nullptr; }
    hipEvent_t stop_ev() const { return on && ext_ ? h->ev[k][h->ev_used[k] + 1] : nullptr; }
    void cancel() { on = false; }  // nothing was launched: leave the (unrecorded) pair unused
    void stop() { if (on) { if (!ext_) (void)hipEventRecord(h->ev[k][h->ev_used[k] + 1], h->stream); h->ev_used[k] += 2; on = false; } }
    ~TimedLaunch() { stop(); }
};

static thread_local std::string g_create_err;

// Environment knobs.  The ones with a purpose for users and tests are read with getenv where they apply (ROGUE_GYM_HIP_NO_SPARES, _NO_STAIR_WAVES,
// _KEEP_SPARES, _FULL_BFS, _EPW: each is named in a test).  The placements of the background generator that DESIGN_HISTORY.md records as measured
// and rejected exist only in a development build (-DRG_DEV_KNOBS): no run-time branch of them is left in the default launch sequence.
#ifdef RG_DEV_KNOBS
#define RG_DEV_ENV(name) getenv(name)
#else
#define RG_DEV_ENV(name) ((const char *)nullptr)
#endif

#define HIPCHK(h, call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

template <typename T>
static bool dev_alloc(rg_handle *h, T **p, size_t count) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&q, bytes) != hipSuccess) { h->err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; return false; }
    if (hipMemset(q, 0, bytes) != hipSuccess) { h->err = "hipMemset failed"; return false; }
    h->allocs.push_back(q);
    *p = (T *)q;
    return true;
}

static void free_all(rg_handle *h) {
    for (void *p : h->allocs) (void)hipFree(p);
    h->allocs.clear();
}

static int flush_render(rg_handle *h) {
    if (h->render_pending) {
        { TimedLaunch t(h, 1); rgk_render(&h->S, &h->cfg, h->stream); }
        HIPCHK(h, hipGetLastError());
        h->render_pending = false;
        h->bound_valid = false;  // (Redraw flags were consumed without the bound observation tensor being written: its next call encodes every env)
    }
    return 0;
}

static int upload_seeds(rg_handle *h, size_t n) {  // the first n envs
    HIPCHK(h, hipMemcpyAsync(h->S.seed_lo, h->seed_lo.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->S.seed_hi, h->seed_hi.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->S.reseed, h->reseed.data(), n, hipMemcpyHostToDevice, h->stream));
    return 0;  // pageable sources: the runtime stages them before returning, so the host vectors may change right after
}

extern "C" {

const char *rg_last_error(const rg_t *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

#ifndef RG_BUILD_ID
#define RG_BUILD_ID "unstamped"
#endif
// sha256[:16] of the sources this library was built from (__graft_entry__.source_id); the marker makes it readable from the file without dlopen
const char *rg_build_id(void) { static const char id[] = "RGBUILDID:" RG_BUILD_ID; return id + 10; }

static void destroy_handle(rg_handle *h);  // (synchronises and destroys the background streams before it frees: every failure path of create_homog goes through it)
// what differs between the envs of one config group: the seed, or the range a fresh seed is drawn from
struct EnvSeed { bool has_seed, has_range; uint64_t lo, hi; unsigned __int128 r0, r1; };

static int create_homog(const RgParsed &parsed, const EnvSeed *seeds, int n_env, uint64_t max_steps, int device, int auto_reset, rg_handle **out) {
    *out = nullptr;
    rg_handle *h = new rg_handle();
    std::random_device rd;
    std::mt19937_64 gen(((uint64_t)rd() << 32) ^ rd());
    h->seed_lo.resize(n_env); h->seed_hi.resize(n_env); h->reseed.resize(n_env);
    h->parsed = parsed;
    for (int i = 0; i < n_env; i++) {
        const EnvSeed &p = seeds[i];
        if (p.has_seed) { h->seed_lo[i] = p.lo; h->seed_hi[i] = p.hi; h->reseed[i] = 0; }
        else { h->seed_lo[i] = gen(); h->seed_hi[i] = gen(); h->reseed[i] = 1; }  // `seed: None`: every build draws its own seed on the device from this base (build_prologue)
        if (p.has_range) {  // ... inside seed_range if one is given (kept, for dump_config, also when a seed overrides it: core/src/lib.rs:57-61)
            if (!(p.r1 > p.r0)) {  // rng::gen_ranged_seed -> gen_range(start, end) panics when start >= end (core/src/rng.rs:42-45) ...
                // ... but the range is consulted only without a seed (core/src/lib.rs:157-165): {seed: 7, seed_range: [5, 5]} builds there
                if (!p.has_seed) { g_create_err = "Invalid Setting: seed_range must satisfy start < end"; delete h; return 1; }
                continue;  // kept for dump_config (RgParsed), never used
            }
            if (h->range_lo.empty()) { h->range_lo.assign(2 * (size_t)n_env, 0); h->range_span.assign(2 * (size_t)n_env, 0); }
            const unsigned __int128 span = p.r1 - p.r0;
            h->range_lo[i] = (uint64_t)p.r0; h->range_lo[n_env + i] = (uint64_t)(p.r0 >> 64);
            h->range_span[i] = (uint64_t)span; h->range_span[n_env + i] = (uint64_t)(span >> 64);
            if (!p.has_seed) h->reseed[i] = 2;
        }
    }
    h->cfg = h->parsed.cfg;
    h->planes_sym = h->cfg.symbols;
    h->cfg.max_steps = max_steps > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)max_steps;
    h->cfg.auto_reset = auto_reset ? 1 : 0;
    h->device = device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device available (librogue_gym_hip has no CPU fallback)"; delete h; return 1;
    }
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) { g_create_err = "invalid HIP device " + std::to_string(device); delete h; return 1; }

    RgState &S = h->S;
    memset(&S, 0, sizeof S);
    const size_t n = (size_t)n_env, hw = (size_t)h->cfg.width * h->cfg.height;
    S.n = n_env; S.hw = (int)hw; S.n_keys = n_env;
    // one DFS stack entry per maze node at most: nodes of the largest assigned area (maze rooms span the area minus one row / column, rooms.rs:240-247)
    const int maze_cap = ((h->cfg.width / h->cfg.room_num_x + 1) / 2) * ((h->cfg.height / h->cfg.room_num_y + 1) / 2) + 1;
    S.maze_cap = maze_cap;
    const size_t nr = (size_t)h->cfg.room_num_x * h->cfg.room_num_y, ne = 2 * nr;  // slots in use: rooms (= monster / gold slots), corridor records (rg_state.h)
    h->stat_rows = 2048 + (n + 15) / 16;  // >= the largest k_step grid (descent + monster blocks + one block per 16 envs)
    bool ok = dev_alloc(h, &S.cell, n * hw) && dev_alloc(h, &S.screen, n * hw) && dev_alloc(h, &S.hist, n * hw) &&
              dev_alloc(h, &S.p_pos, n) && dev_alloc(h, &S.p_hp, n) && dev_alloc(h, &S.p_hpmax, n) && dev_alloc(h, &S.p_lvl, n) &&
              dev_alloc(h, &S.p_exp, n) && dev_alloc(h, &S.food, n) && dev_alloc(h, &S.quiet, n) && dev_alloc(h, &S.pack_gold, n) &&
              dev_alloc(h, &S.dlevel, n) && dev_alloc(h, &S.steps, n) && dev_alloc(h, &S.flags, n) && dev_alloc(h, &S.reward, n) && dev_alloc(h, &S.done, n) &&
              dev_alloc(h, &S.rng, 12 * n) && dev_alloc(h, &S.seed_lo, n) && dev_alloc(h, &S.seed_hi, n) && dev_alloc(h, &S.reseed, n) &&
              dev_alloc(h, &S.room_rect, nr * n) && dev_alloc(h, &S.room_meta, nr * n) &&
              dev_alloc(h, &S.mon_w0, nr * n) && dev_alloc(h, &S.mon_hp, nr * n) && dev_alloc(h, &S.mon_exp, nr * n) &&
              dev_alloc(h, &S.mon_cnt, n) && dev_alloc(h, &S.gold_pos, nr * n) && dev_alloc(h, &S.gold_amt, nr * n) &&
              dev_alloc(h, &S.edge_a, ne * n) && dev_alloc(h, &S.edge_b, ne * n) &&
              dev_alloc(h, &S.maze_stack, (size_t)maze_cap * n) && dev_alloc(h, &S.build_ctr, n) && dev_alloc(h, &S.stats, RG_STAT_COLS * h->stat_rows) && dev_alloc(h, &S.stair_list, 2 * n) && dev_alloc(h, &S.stair_cnt, 8) && dev_alloc(h, &S.stair_mark, 2 * n) &&
              dev_alloc(h, &S.dc_map, h->cfg.n_enemies > 0 ? n * RG_DIST_SLOTS * hw : 16) && dev_alloc(h, &S.dc_key, RG_DIST_SLOTS * n) &&
              dev_alloc(h, &S.dc_head, n) && dev_alloc(h, &S.dc_len, n) && dev_alloc(h, &S.dc_part, n) && dev_alloc(h, &S.dc_own, n) && dev_alloc(h, &S.status, n * 10) &&
              dev_alloc(h, &h->d_err, 4) && dev_alloc(h, &h->d_keys, n);
    // the envs' observation records (rg_state.h obs_rec): for the grids the fused observation pass handles
    if (ok && nr <= RG_OBS_MAX_ROOMS) ok = dev_alloc(h, &S.obs_rec, n * (size_t)RG_OBS_REC_WORDS(nr));
    if (ok && nr <= RG_OVL_MAX && getenv("ROGUE_GYM_HIP_NO_MIRROR_UPDATE") == nullptr)  // (the A side: every Redraw drawn from the tiles by the observation pass)
        ok = dev_alloc(h, &S.ovl, (nr + 1) * n) && hipMemset(S.ovl, 0xff, (nr + 1) * n * 2) == hipSuccess;
    h->spares = auto_reset != 0 && getenv("ROGUE_GYM_HIP_NO_SPARES") == nullptr;
    // which producer refills the consumed spares: one level per LANE (rg_regen_lanes.hip; two spares per env, rg_state.h sp_slots) where it applies,
    // else -- or with ROGUE_GYM_HIP_WAVE_REGEN=1 -- one level per wave (k_regen, one spare per env)
    h->lane_regen = h->spares && getenv("ROGUE_GYM_HIP_WAVE_REGEN") == nullptr && rgk_regen_lanes_supported(&h->cfg, maze_cap) > 0;
    // ROGUE_GYM_HIP_SP_SLOTS = 1..8 (default 4): spare level-1 states kept per env.  Each costs a copy of the env's grid and tables (80x24: 3.9 KB -- 1 GB for
    // four at 65 536 envs); fewer slots = more resets that find none and generate inline, never another result.
    const char *slots_env = getenv("ROGUE_GYM_HIP_SP_SLOTS");
    S.sp_slots = h->lane_regen ? (slots_env ? atoi(slots_env) : 4) : 1;
    if (S.sp_slots < 1 || S.sp_slots > 8) S.sp_slots = 4;
    const size_t ns = n * (size_t)S.sp_slots;  // entries of the spare view
    ok = ok && dev_alloc(h, &S.sp_ready, ns) && dev_alloc(h, &h->d_probe, 4) && dev_alloc(h, &S.launch_mark, 4);
    // the grid class whose dist maps may be partial (rg_kernels.hip bfs_rows_n32): one saved walkable mask per map
    if (ok && h->cfg.n_enemies > 0 && RG_PARTIAL_MAPS(h->cfg.width, h->cfg.height, (int)nr))
        ok = dev_alloc(h, &S.dc_walk, n * RG_DIST_SLOTS * h->cfg.height * RG_WALK_WORDS(h->cfg.width));
    S.full_bfs = getenv("ROGUE_GYM_HIP_FULL_BFS") != nullptr;
    S.keep_spares = getenv("ROGUE_GYM_HIP_KEEP_SPARES") != nullptr;
    S.err_any = h->d_err;
    if (ok && !h->range_lo.empty()) {
        ok = dev_alloc(h, &S.range_lo, 2 * n) && dev_alloc(h, &S.range_span, 2 * n) &&
             hipMemcpy(S.range_lo, h->range_lo.data(), 2 * n * 8, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(S.range_span, h->range_span.data(), 2 * n * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok && h->err.empty()) h->err = "hipMemcpy failed";
    }
    if (ok && !h->parsed.init_draws.empty()) {  // Player::init_items' item-stream draws (rg_items.cpp): (lo, hi) pairs, read by build_epilogue
        uint32_t *d = nullptr;
        ok = dev_alloc(h, &d, h->parsed.init_draws.size()) &&
             hipMemcpy(d, h->parsed.init_draws.data(), h->parsed.init_draws.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok && h->err.empty()) h->err = "hipMemcpy failed";
        S.init_draws = d;
    }
    // next-level structures (rg_kernels.hip gen_service): generated by the spare pipeline's kernel, for the generator instances with the room table in
    // registers (<= 64 rooms).  ROGUE_GYM_HIP_NO_NEXT_LEVELS: every descent generates its whole level inline (the A side of the parity / A-B tests)
    if (ok && h->spares && nr <= 64 && getenv("ROGUE_GYM_HIP_NO_NEXT_LEVELS") == nullptr) {
        RgNext nx;
        memset(&nx, 0, sizeof nx);
        RgNext *d_nx = nullptr;
        ok = dev_alloc(h, &nx.cell, n * hw) && dev_alloc(h, &nx.room_rect, nr * n) && dev_alloc(h, &nx.room_meta, nr * n) &&
             dev_alloc(h, &nx.gold_pos, nr * n) && dev_alloc(h, &nx.gold_amt, nr * n) && dev_alloc(h, &nx.level, n) &&
             dev_alloc(h, &S.nx_rng, 12 * n) && dev_alloc(h, &S.nx_state, n) && dev_alloc(h, &d_nx, 1);
        ok = ok && hipMemcpy(d_nx, &nx, sizeof nx, hipMemcpyHostToDevice) == hipSuccess;
        S.nx = d_nx;
    }
    h->SP = S;
    if (ok && h->spares) {
        RgState &P = h->SP;
        ok = dev_alloc(h, &P.cell, ns * hw) && dev_alloc(h, &P.p_pos, ns) && dev_alloc(h, &P.p_hp, ns) && dev_alloc(h, &P.p_hpmax, ns) && dev_alloc(h, &P.p_lvl, ns) &&
             dev_alloc(h, &P.p_exp, ns) && dev_alloc(h, &P.food, ns) && dev_alloc(h, &P.quiet, ns) && dev_alloc(h, &P.pack_gold, ns) && dev_alloc(h, &P.dlevel, ns) &&
             dev_alloc(h, &P.rng, 12 * ns) && dev_alloc(h, &P.room_rect, nr * ns) && dev_alloc(h, &P.room_meta, nr * ns) &&
             dev_alloc(h, &P.mon_w0, nr * ns) && dev_alloc(h, &P.mon_hp, nr * ns) && dev_alloc(h, &P.mon_exp, nr * ns) &&
             dev_alloc(h, &P.mon_cnt, ns) && dev_alloc(h, &P.gold_pos, nr * ns) && dev_alloc(h, &P.gold_amt, nr * ns) &&
             dev_alloc(h, &P.edge_a, ne * n) && dev_alloc(h, &P.edge_b, ne * n) && dev_alloc(h, &P.maze_stack, (size_t)maze_cap * n) &&
             dev_alloc(h, &P.on_stairs, ns);
        P.prof = nullptr;
        int lo = 0, hi = 0;
        if (ok && (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, RG_DEV_ENV("ROGUE_GYM_HIP_SIDE_HIPRIO") ? hi : lo) != hipSuccess)) {
            h->err = "failed to create the background generation stream"; ok = false;
        }
        // (default priority: the packets of a LOW-priority queue are fetched only while the handle's own queue has none waiting -- in a free-running loop,
        // never: measured, the launches ran thousands of steps late -- and it is the waves' own priority, not the queue's, that keeps these few waves out of
        // k_step's way)
        static const bool lanes_low = RG_DEV_ENV("ROGUE_GYM_HIP_LANES_LOWPRIO") != nullptr;
        if (ok && (hipStreamCreateWithPriority(&h->side2, hipStreamNonBlocking, lanes_low ? lo : (lo + hi) / 2) != hipSuccess || hipStreamCreateWithPriority(&h->side3, hipStreamNonBlocking, lanes_low ? lo : (lo + hi) / 2) != hipSuccess)) { h->err = "failed to create the background generation stream"; ok = false; }
    }
    ok = ok && dev_alloc(h, &h->d_SP, 1) && hipMemcpy(h->d_SP, &h->SP, sizeof(RgState), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { g_create_err = h->err.empty() ? "device allocation failed" : h->err; destroy_handle(h); return 1; }
    // screen rows 0 and H-1 are never drawn: PlayerState::new fills the map with b' ' (python/src/lib.rs:41-50)
    if (hipMemset(S.screen, ' ', n * hw) != hipSuccess) { g_create_err = "hipMemset failed"; destroy_handle(h); return 1; }
    if (upload_seeds(h, n)) { g_create_err = h->err; destroy_handle(h); return 1; }
    h->S.stair_gen = h->stair_gen++;  // k_build produces the stair set of the first k_step
    rgk_build(&h->S, &h->cfg, h->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { g_create_err = std::string("k_build: ") + hipGetErrorString(e); destroy_handle(h); return 1; }
    h->render_pending = true;
    if (h->spares) {
        // first spares.  rg_create waits for them: left in the background, this one-off generation of EVERY env's spare (~2 ms at 65 536 envs)
        // competes with the first few hundred steps for issue slots (the driver's 20-step bench ran k_step at 141 us instead of ~100 us).
        if (h->lane_regen && !(dev_alloc(h, &h->lane_q, 8) && dev_alloc(h, &h->lane_list, 2 * ns))) { g_create_err = h->err; destroy_handle(h); return 1; }
        if (h->lane_regen) (void)rgk_regen_lanes(&h->SP, &h->cfg, h->lane_q, h->lane_list, 1, 0, h->S.sp_slots, h->side2, nullptr, nullptr);  // (bulk: every spare; no gate)
        else rgk_regen(&h->SP, &h->cfg, 1, 1, nullptr, 0, h->d_err, h->side, nullptr, nullptr);
        e = hipGetLastError();
        if (e == hipSuccess && !RG_DEV_ENV("ROGUE_GYM_HIP_ASYNC_FIRST_SPARES")) { e = hipStreamSynchronize(h->side); if (e == hipSuccess) e = hipStreamSynchronize(h->side2); if (e == hipSuccess) e = hipStreamSynchronize(h->side3); }  // (dev knob: the round-1 behaviour)
        if (e != hipSuccess) { g_create_err = std::string("k_regen: ") + hipGetErrorString(e); destroy_handle(h); return 1; }
        h->regen_bulk = 1;  // (the first steady-state launch too: whatever the creation launch left, e.g. when it ran in the background)
        bool all_fixed = true;
        for (uint8_t m : h->reseed) all_fixed = all_fixed && m == 0;
        if (h->S.keep_spares && all_fixed && !h->S.nx_state) h->regen_idle_after = 1;  // kept spares are never consumed: one more launch (if the first one ran in the background), then none
    }
    *out = h;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// handles whose envs differ in more than the seed: parent over one homogeneous sub-handle per distinct config
// ---------------------------------------------------------------------------------------------
#define SUBCHK(h, sh, call) do { if (call) { (h)->err = (sh)->err; return 1; } } while (0)

static void destroy_handle(rg_handle *h);
static int comm_release(rg_handle *h, bool teardown);

// status / flags / reward / done of every group -> the parent's arrays in env order (44 + 5 bytes per env); after every step and reset, so that
// device pointers handed out once (rg_reward, rg_done, rg_status ...) stay current
static int assemble_small(rg_handle *h) {
    for (rg_handle *sh : h->sub) {
        const int m = sh->S.n;
        rgk_scatter_rows(sh->S.status, h->S.status, sh->d_ext, m, 40, h->stream);
        rgk_scatter_rows(sh->S.reward, h->S.reward, sh->d_ext, m, 4, h->stream);
        rgk_scatter_rows(sh->S.done, h->S.done, sh->d_ext, m, 1, h->stream);
        rgk_scatter_rows(sh->S.flags, h->S.flags, sh->d_ext, m, 4, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
// screens, history planes (and the flag words, which the render pass updates) on demand
static int assemble_screens(rg_handle *h) {
    if (!h->screens_stale) return 0;
    for (rg_handle *sh : h->sub) {
        SUBCHK(h, sh, flush_render(sh));
        const int m = sh->S.n;
        if (!h->mixed) {  // (a mixed-size batch has no common screen tensor: its screens stay with the groups, rg_fetch_states reads them there)
            rgk_scatter_rows(sh->S.screen, h->S.screen, sh->d_ext, m, h->S.hw, h->stream);
            rgk_scatter_rows(sh->S.hist, h->S.hist, sh->d_ext, m, h->S.hw, h->stream);
        }
        rgk_scatter_rows(sh->S.flags, h->S.flags, sh->d_ext, m, 4, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    h->screens_stale = false;
    return 0;
}

int rg_create(const char *const *cfg_json, int n_env, uint64_t max_steps, int device, int auto_reset, rg_t **out) {
    if (!out || n_env <= 0) { g_create_err = "rg_create: invalid arguments"; return 1; }
    *out = nullptr;
    // parse every config ONCE and group the envs by parsed config (everything the device code reads; seeds and seed ranges stay per env)
    std::vector<RgParsed> reps;
    std::vector<EnvSeed> seeds(n_env);
    std::vector<int> g_of(n_env), l_of(n_env);
    std::vector<std::vector<int>> members;
    {
        const char *prev = nullptr; int prev_g = -1;
        RgParsed p;
        for (int i = 0; i < n_env; i++) {
            const char *js = cfg_json ? cfg_json[i] : nullptr;
            const bool same_text = i > 0 && ((js == nullptr && prev == nullptr) || (js && prev && strcmp(js, prev) == 0));
            int g = prev_g;
            if (!same_text) {
                std::string e = rg_parse_config(js, &p);
                if (!e.empty()) { g_create_err = "Failed to parse config: " + e; return 1; }
                g = -1;
                for (size_t k = 0; k < reps.size(); k++) if (rg_config_equal(reps[k], p)) { g = (int)k; break; }
                if (g < 0) { g = (int)reps.size(); reps.push_back(p); members.emplace_back(); }
                prev = js; prev_g = g;
            }
            seeds[i] = EnvSeed{p.has_seed, p.has_seed_range, p.seed_lo, p.seed_hi, p.seed_range[0], p.seed_range[1]};
            g_of[i] = g; l_of[i] = (int)members[g].size(); members[g].push_back(i);
        }
    }
    if (reps.size() == 1) return create_homog(reps[0], seeds.data(), n_env, max_steps, device, auto_reset, out);
    bool mixed = false;
    for (size_t k = 1; k < reps.size(); k++) mixed = mixed || reps[k].cfg.width != reps[0].cfg.width || reps[k].cfg.height != reps[0].cfg.height;
    rg_handle *h = new rg_handle();
    h->device = device;
    h->mixed = mixed;
    h->g_of = g_of; h->l_of = l_of;
    for (size_t k = 0; k < reps.size(); k++) {
        std::vector<EnvSeed> gs(members[k].size());
        for (size_t j = 0; j < members[k].size(); j++) gs[j] = seeds[members[k][j]];
        rg_handle *sh = nullptr;
        if (create_homog(reps[k], gs.data(), (int)gs.size(), max_steps, device, auto_reset, &sh)) { destroy_handle(h); return 1; }
        h->sub.push_back(sh);
        sh->ext.assign(members[k].begin(), members[k].end());
        bool ok = dev_alloc(sh, &sh->d_ext, gs.size()) && dev_alloc(sh, &sh->d_keys_sub, gs.size()) &&
                  hipMemcpy(sh->d_ext, sh->ext.data(), gs.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) { g_create_err = "device allocation failed"; destroy_handle(h); return 1; }
        sh->S.ext = sh->d_ext;
    }
    if (mixed) {
        h->ragged_off.assign((size_t)n_env + 1, 0);
        for (int i = 0; i < n_env; i++) h->ragged_off[i + 1] = h->ragged_off[i] + (size_t)reps[g_of[i]].cfg.width * reps[g_of[i]].cfg.height;
    }
    rg_handle *first = h->sub[g_of[0]];
    h->parsed = first->parsed; h->cfg = first->cfg;
    h->planes_sym = first->cfg.symbols;  // ParallelGameState::new takes `symbols` from configs[0] (python/src/lib.rs:281-285)
    for (rg_handle *sh : h->sub) sh->planes_sym = h->planes_sym;
    RgState &S = h->S;
    memset(&S, 0, sizeof S);
    const size_t n = (size_t)n_env, hw = (size_t)h->cfg.width * h->cfg.height;
    S.n = n_env; S.hw = (int)hw; S.n_keys = n_env;
    bool ok = (mixed || (dev_alloc(h, &S.screen, n * hw) && dev_alloc(h, &S.hist, n * hw))) && dev_alloc(h, &S.status, n * 10) && dev_alloc(h, &S.flags, n) &&
              dev_alloc(h, &S.reward, n) && dev_alloc(h, &S.done, n) && dev_alloc(h, &h->d_err, 4) && dev_alloc(h, &h->d_probe, 4);
    if (!ok) { g_create_err = h->err; destroy_handle(h); return 1; }
    S.err_any = h->d_err;
    if (assemble_small(h) || hipStreamSynchronize(h->stream) != hipSuccess) { g_create_err = h->err.empty() ? "assemble failed" : h->err; destroy_handle(h); return 1; }
    *out = h;
    return 0;
}

static void destroy_handle(rg_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (rg_handle *sh : h->sub) destroy_handle(sh);
    h->sub.clear();
    (void)hipStreamSynchronize(h->stream);
    if (h->comm) (void)comm_release(h, true);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->side2) { (void)hipStreamSynchronize(h->side2); (void)hipStreamDestroy(h->side2); }
    if (h->side3) { (void)hipStreamSynchronize(h->side3); (void)hipStreamDestroy(h->side3); }
    for (int k = 0; k < RG_TIMED_KERNELS; k++) for (auto &e : h->ev[k]) (void)hipEventDestroy(e);
    if (h->obs_scratch) (void)hipFree(h->obs_scratch);
    if (h->pin_keys) (void)hipHostFree(h->pin_keys);
    if (h->pin_err) (void)hipHostFree(h->pin_err);
    free_all(h);
    delete h;
}

void rg_destroy(rg_t *h) { destroy_handle(h); }

int rg_dims(const rg_t *h, int *height, int *width, int *symbols, int *n_env) {
    if (height) *height = h->cfg.height;
    if (width) *width = h->cfg.width;
    if (symbols) *symbols = h->planes_sym;
    if (n_env) *n_env = h->S.n;
    return 0;
}

int rg_env_dims(const rg_t *h, int32_t *heights, int32_t *widths) {
    for (int i = 0; i < h->S.n; i++) {
        const RgConfig &c = h->sub.empty() ? h->cfg : h->sub[h->g_of[i]]->cfg;
        if (heights) heights[i] = c.height;
        if (widths) widths[i] = c.width;
    }
    return 0;
}
// the entry points that hand out or fill [n_env][..][H][W] tensors cannot serve a batch whose envs differ in width / height
static bool refuse_mixed(rg_handle *h, const char *what) {
    if (!h->mixed) return false;
    h->err = std::string(what) + ": the envs of this batch differ in width / height, so there is no [n_env][H][W] tensor (per-env sizes: rg_env_dims; "
             "ragged host copies: rg_fetch_states; or one handle per size)";
    return true;
}

int rg_env_symbols(const rg_t *h, int32_t *out_host) {
    if (h->sub.empty()) { for (int i = 0; i < h->S.n; i++) out_host[i] = h->cfg.symbols; return 0; }
    for (int i = 0; i < h->S.n; i++) out_host[i] = h->sub[h->g_of[i]]->cfg.symbols;
    return 0;
}

int rg_set_stream(rg_t *h, void *hip_stream) {
    for (rg_handle *sh : h->sub) SUBCHK(h, sh, rg_set_stream(sh, hip_stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }
    h->stream = (hipStream_t)hip_stream;
    return 0;
}

int rg_seed(rg_t *h, const uint64_t *seed_lo, const uint64_t *seed_hi, int n) {
    if (n > h->S.n) n = h->S.n;
    if (n <= 0) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {  // the first n envs of the handle are a prefix of every group (groups keep the env order)
        for (rg_handle *sh : h->sub) {
            std::vector<uint64_t> lo, hi;
            for (int l = 0; l < sh->S.n && sh->ext[l] < n; l++) { lo.push_back(seed_lo[sh->ext[l]]); hi.push_back(seed_hi ? seed_hi[sh->ext[l]] : 0); }
            if (!lo.empty()) SUBCHK(h, sh, rg_seed(sh, lo.data(), hi.data(), (int)lo.size()));
        }
        return 0;
    }
    for (int i = 0; i < n; i++) { h->seed_lo[i] = seed_lo[i]; h->seed_hi[i] = seed_hi ? seed_hi[i] : 0; h->reseed[i] = 0; }
    if (h->spares) {
        // the spares of these envs were generated from the old seeds: drop them (k_step generates inline until k_regen has refilled them).
        // A k_regen in flight may be about to publish one of them, so the side stream is drained first; the main stream is not.
        HIPCHK(h, hipStreamSynchronize(h->side));
        HIPCHK(h, hipStreamSynchronize(h->side2));
        HIPCHK(h, hipStreamSynchronize(h->side3));
        for (int sl = 0; sl < h->S.sp_slots; sl++) HIPCHK(h, hipMemsetAsync(h->S.sp_ready + (size_t)sl * h->S.n, 0, (size_t)n * 4, h->stream));
        if (h->regen_idle_after >= 0) h->regen_idle_after = 2;  // rebuild the dropped spares, then idle again
        h->regen_bulk = 2;
    }
    return upload_seeds(h, (size_t)n);  // only the touched prefix travels; the seeds of `seed: None` envs are derived on the device and never read back
}

int rg_reset(rg_t *h) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {
        for (rg_handle *sh : h->sub) SUBCHK(h, sh, rg_reset(sh));
        h->screens_stale = true;
        return assemble_small(h);
    }
    h->S.stair_gen = h->stair_gen++;
    { TimedLaunch t(h, 3); rgk_build(&h->S, &h->cfg, h->stream); }  // (k_build and a k_regen in flight share only the atomically advanced build counters)
    HIPCHK(h, hipGetLastError());
    h->render_pending = true;
    return 0;
}

static int obs_common(rg_t *h, uint32_t status_flag, int with_hist, int kind, float *out_dev);
int rg_step(rg_t *h, const uint8_t *keys, int keys_on_device) { return rg_step_prefix(h, keys, h->S.n, keys_on_device); }
int rg_step_obs_gray(rg_t *h, const uint8_t *keys, int keys_on_device, uint32_t status_flag, int with_hist, float *out_dev) {
    return rg_step_prefix(h, keys, h->S.n, keys_on_device) ? 1 : obs_common(h, status_flag, with_hist, 0, out_dev);
}

int rg_step_prefix(rg_t *h, const uint8_t *keys, int n_keys, int keys_on_device) {
    HIPCHK(h, hipSetDevice(h->device));
    if (n_keys < 0) { h->err = "rg_step_prefix: negative key count"; return 1; }
    if (n_keys > h->S.n) n_keys = h->S.n;  // zip: surplus keys are dropped
    if (!h->sub.empty()) {
        const uint8_t *dkeys = keys;
        if (!keys_on_device) {
            if (!h->d_keys && !dev_alloc(h, &h->d_keys, (size_t)h->S.n)) return 1;
            HIPCHK(h, hipMemcpyAsync(h->d_keys, keys, (size_t)n_keys, hipMemcpyHostToDevice, h->stream));
            dkeys = h->d_keys;
        }
        for (rg_handle *sh : h->sub) {
            int m = 0;  // the keyed envs of the group: a prefix of it
            while (m < sh->S.n && sh->ext[m] < n_keys) m++;
            if (m) rgk_gather_keys(dkeys, sh->d_ext, sh->d_keys_sub, m, h->stream);
            SUBCHK(h, sh, rg_step_prefix(sh, sh->d_keys_sub, m, 1));
        }
        HIPCHK(h, hipGetLastError());
        h->screens_stale = true;
        return assemble_small(h);
    }
    if (flush_render(h)) return 1;
    const uint8_t *dk = keys;
    if (!keys_on_device) {
        HIPCHK(h, hipMemcpyAsync(h->d_keys, keys, (size_t)n_keys, hipMemcpyHostToDevice, h->stream));
        dk = h->d_keys;
    }
    h->S.n_keys = n_keys;
    // The consumed spares are refilled on the side stream (k_regen).  WHERE that work lands decides what it costs: ~450 level generations per 65 536-env
    // step are ~12 000 wave-us of scalar-unit work.  The current form (round 4; the earlier placements and their numbers are in DESIGN_HISTORY.md): a
    // launch with EVERY step, enqueued right behind the k_step launch and let in by a one-wave GATE kernel once that k_step has started (its block 0
    // publishes the launch number) -- so the generator runs beside k_step, the latency-bound kernel, never in front of it and not beside the bandwidth-
    // bound observation pass; low stream priority, so k_step's blocks are placed first; eight envs per generator wave and at most ONE generation per
    // wave and launch (rgk_regen), so the launch is over in about one generation time.  No event and no packet of this is on the handle's stream: the
    // round-2/3 form hung the launch on the observation pass's completion event, and an event handed to a launch costs that stream a few us each time.
    static const int regen_every = RG_DEV_ENV("ROGUE_GYM_HIP_REGEN_EVERY") ? atoi(RG_DEV_ENV("ROGUE_GYM_HIP_REGEN_EVERY")) : 1;
    h->step_count++;
    bool regen = h->spares && (regen_every <= 1 || h->step_count % (uint64_t)regen_every == 0);
    if (regen && h->regen_idle_after >= 0) { if (h->regen_idle_after == 0) regen = false; else h->regen_idle_after--; }
    static const bool no_stair_waves = getenv("ROGUE_GYM_HIP_NO_STAIR_WAVES") != nullptr;
    {   // stair isolation (rg_kernels.hip, k_step): on unless ROGUE_GYM_HIP_NO_STAIR_WAVES is set (the A/B knob of DESIGN.md's measurement)
        TimedLaunch t(h, 0, true);
        h->S.stair_gen = h->stair_gen++;  // reads the stair set its predecessor produced, produces the next one
        h->S.obs_par = (int32_t)(h->step_count & 1); h->bound_steps++;  // (rg_obs_bind: this launch's half of the work list)
        (void)rgk_step(&h->S, h->d_SP, &h->cfg, dk, h->spares ? 1 : 0, no_stair_waves ? -1 : 0, h->stream, t.start_ev(), t.stop_ev());
    }
    HIPCHK(h, hipGetLastError());
    if (regen) {
        TimedLaunch t(h, 4, true);
        if (h->lane_regen) {
            // Two producers.  Beside EVERY step (gate + k_regen on `side`), one level per wave: the next-level structures -- wanted within two steps -- and
            // the urgent spares (an env down to its last ready one).  Every 16th step, one level per LANE (rg_regen_lanes.hip): every spare consumed since,
            // ~5 000 of the 4 x 65 536.  A round of 64 levels takes a wave 300-450 us whatever the launch's size, i.e. three to five steps: launched beside
            // every fourth step, one of them was still running behind ANY short window of steps (the driver's 20: +24 us per step).  Rarer launches leave
            // more envs to the urgent path, whose 45-us wave-per-level builds beside k_step cost the step more than the bulk launches do (1500 steps, one
            // box: every 8th / 16th 675-676 M, 24th 667 M, 32nd 664 M, 64th 647 M, 128th 622 M; an urgent build only for envs with NO spare left: 654 M at
            // 16 -- a reset that finds no spare generates inline, 45 us in an index-order wave).  The launches alternate between two streams of their own.
            static const int lane_waves = RG_DEV_ENV("ROGUE_GYM_HIP_LANE_WAVES") ? atoi(RG_DEV_ENV("ROGUE_GYM_HIP_LANE_WAVES")) : 256;
            static const int lane_every = RG_DEV_ENV("ROGUE_GYM_HIP_LANE_EVERY") ? atoi(RG_DEV_ENV("ROGUE_GYM_HIP_LANE_EVERY")) : 16;
            static const bool time_lanes = RG_DEV_ENV("ROGUE_GYM_HIP_TIME_LANES") != nullptr;  // (development: the event pair of rg_timing's kernel 4 goes to the level-per-lane launch)
            const bool lanes_now = h->regen_bulk > 0 || h->step_count - h->lane_last >= (uint64_t)(lane_every > 0 ? lane_every : 1);
            // (spares = 2: next-level structures + the urgent spares -- envs down to their last ready one; rg_kernels.hip regen_body)
            static const int urgent_mode = RG_DEV_ENV("ROGUE_GYM_HIP_URGENT_AT0") ? 3 : 2;
            rgk_regen(&h->SP, &h->cfg, 0, urgent_mode, h->S.launch_mark, (uint32_t)h->S.stair_gen + 1u, h->d_err, h->side, time_lanes ? nullptr : t.start_ev(), time_lanes ? nullptr : t.stop_ev());
            if (time_lanes && !lanes_now) t.cancel();
            if (lanes_now) {
                h->lane_last = h->step_count;
                h->lane_flip ^= 1;
                // (gated like k_regen: the host runs hundreds of steps ahead of the GPU, and a launch that is not held back until ITS k_step has started scans
                // for consumed spares long before they are consumed -- measured: a backlog of 12 000 spares with a launch every fourth step)
                rgk_regen_gate(h->S.launch_mark, (uint32_t)h->S.stair_gen + 1u, h->d_err, h->lane_flip ? h->side3 : h->side2);
                (void)rgk_regen_lanes(&h->SP, &h->cfg, h->lane_q + 4 * h->lane_flip, h->lane_list + (size_t)h->S.n * h->S.sp_slots * h->lane_flip, h->regen_bulk > 0 ? 1 : 0, lane_waves, h->S.sp_slots,
                                      h->lane_flip ? h->side3 : h->side2, time_lanes ? t.start_ev() : nullptr, time_lanes ? t.stop_ev() : nullptr);
            }
        } else rgk_regen(&h->SP, &h->cfg, h->regen_bulk > 0 ? 1 : 0, 1, h->S.launch_mark, (uint32_t)h->S.stair_gen + 1u, h->d_err, h->side, t.start_ev(), t.stop_ev());
        if (h->regen_bulk > 0) h->regen_bulk--;
    }
    HIPCHK(h, hipGetLastError());
    h->render_pending = true;
    return 0;
}

int rg_sync(rg_t *h) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {
        int rc = 0;
        for (rg_handle *sh : h->sub)
            if (rg_sync(sh) && !rc) { rc = 1; h->err = sh->err; }  // (every group is drained and its error word cleared)
        if (rc) return rc;
    }
    uint32_t err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->d_err, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }
    if (err) {
        HIPCHK(h, hipMemsetAsync(h->d_err, 0, 4, h->stream));
        if (err & RG_FLAG_ERR_INTERNAL) h->err = "internal capacity guard of the HIP stepper tripped (please report the config)";
        else if (err & RG_FLAG_ERR_KEY) h->err = "Invalid input (key is not in the ai keymap)";
        else if (err & RG_FLAG_ERR_DEAD) h->err = "Ignored input (action while the player is dead)";
        else h->err = "Invalid tile in symbol image (symbol >= symbols - 1)";
        return 1;
    }
    return 0;
}

// mirrors up to date: the pending render of an ordinary handle, the assembly of a parent's
static int flush_mirrors(rg_handle *h) { return h->sub.empty() ? flush_render(h) : assemble_screens(h); }
int rg_screen(rg_t *h, uint8_t **dev) { if (refuse_mixed(h, "rg_screen")) return 1; HIPCHK(h, hipSetDevice(h->device)); if (flush_mirrors(h)) return 1; *dev = h->S.screen; return 0; }
int rg_hist(rg_t *h, uint8_t **dev) { if (refuse_mixed(h, "rg_hist")) return 1; HIPCHK(h, hipSetDevice(h->device)); if (flush_mirrors(h)) return 1; *dev = h->S.hist; return 0; }
int rg_status(rg_t *h, int32_t **dev) { *dev = h->S.status; return 0; }
int rg_flags(rg_t *h, uint32_t **dev) { HIPCHK(h, hipSetDevice(h->device)); if (flush_mirrors(h)) return 1; *dev = h->S.flags; return 0; }
int rg_reward(rg_t *h, float **dev) { *dev = h->S.reward; return 0; }
int rg_done(rg_t *h, uint8_t **dev) { *dev = h->S.done; return 0; }
int rg_set_stair_reward(rg_t *h, float bonus) {
    if (!(bonus == bonus) || bonus < 0.f) { h->err = "rg_set_stair_reward: the bonus must be a number >= 0"; return 1; }
    h->S.stair_reward = bonus;  // (a kernel argument of the next rg_step: nothing to upload)
    for (rg_handle *sh : h->sub) sh->S.stair_reward = bonus;
    return 0;
}

int rg_obs_channels(const rg_t *h, int symbol, uint32_t status_flag, int with_hist) {
    return (symbol ? h->planes_sym : 1) + __builtin_popcount(status_flag & 0x1ffu) + (with_hist ? 1 : 0);
}

static int obs_common(rg_t *h, uint32_t status_flag, int with_hist, int kind, float *out_dev) {
    if (refuse_mixed(h, kind ? "rg_obs_symbol" : "rg_obs_gray")) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {  // every group encodes its envs straight into the handle's tensor (RgState::ext)
        for (rg_handle *sh : h->sub) {
            if (kind && sh->cfg.symbols > h->planes_sym) {
                h->err = "symbol image: a config of the batch has more symbols (" + std::to_string(sh->cfg.symbols) + ") than env 0's (" + std::to_string(h->planes_sym) +
                         "), which sets the channel count (python/src/lib.rs:281-285)";
                return 1;
            }
            SUBCHK(h, sh, obs_common(sh, status_flag, with_hist, kind, out_dev));
        }
        return 0;
    }
    {   // steady state: one fused pass refreshes the mirrors of Redraw envs and encodes every env
        TimedLaunch t(h, 2, true);
        // the bound tensor with its own image setting: in place (only the envs whose screen changed), once its contents are known to be current
        const bool bound = h->bound_out && out_dev == h->bound_out && kind == h->bound_kind && (status_flag & 0x1ffu) == 0 && !with_hist;
        if (rgk_obs(&h->S, &h->cfg, status_flag & 0x1ffu, with_hist ? 1 : 0, kind, out_dev, h->d_err, h->planes_sym, bound && h->S.obs_list && h->bound_valid && h->bound_steps == 1, h->stream, t.start_ev(), t.stop_ev())) {
            HIPCHK(h, hipGetLastError());
            h->render_pending = false;
            h->bound_valid = bound;  // (any other observation call consumed Redraw flags the bound tensor has not seen)
            h->bound_steps = 0;
            return 0;
        }
        t.cancel();  // H*W % 8 != 0: nothing was launched, the fallback below brackets its own pair
    }
    h->bound_valid = false;
    if (flush_render(h)) return 1;
    {
        TimedLaunch t(h, 2);
        rgk_encode(h->S.screen, h->S.hist, h->S.status, h->S.flags, h->d_err, h->S.n, h->S.hw, (size_t)h->S.hw, 10, h->cfg.symbols, h->planes_sym,
                   status_flag & 0x1ffu, with_hist ? 1 : 0, kind, out_dev, h->S.ext, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
// Symbol::from_tile (core/src/symbol.rs:17-40) on the host, as rg_device.h tile_to_sym; 255 = not a symbol
static uint32_t host_tile_to_sym(uint32_t t) {
    switch (t) {
    case ' ': return 0; case '@': return 1; case '#': return 2; case '.': return 3; case '-': case '|': return 4;
    case '%': return 5; case '+': return 6; case '^': return 7; case '!': return 8; case '?': return 9; case ']': return 10;
    case ')': return 11; case '/': return 12; case '*': return 13; case ':': return 14; case '=': return 15; case ',': return 16;
    default: return (t >= 'A' && t <= 'Z') ? t - 'A' + 17 : 255u;
    }
}
int rg_obs_bind(rg_t *h, int kind, uint32_t status_flag, int with_hist, float *out_dev) {
    if (!h->sub.empty()) { h->err = "rg_obs_bind: not for a handle with config groups"; return 1; }
    if (out_dev && ((status_flag & 0x1ffu) || with_hist || (kind != 0 && kind != 1))) { h->err = "rg_obs_bind: gray or symbol image without status planes and history plane"; return 1; }
    HIPCHK(h, hipSetDevice(h->device));
    const bool capable = rgk_step_bound_capable(&h->cfg) != 0;  // (else: bound, but every call re-encodes every env)
    if (!capable) { h->S.obs_list = nullptr; h->S.obs_cnt = nullptr; h->S.bound_gray = nullptr; h->bound_out = out_dev; h->bound_kind = kind; h->bound_valid = false; h->bound_steps = 0; return 0; }
    if (out_dev && !h->obs_list_mem) {  // the work list k_step leaves for the in-place pass (rg_state.h obs_list)
        if (!dev_alloc(h, &h->obs_list_mem, 2 * (size_t)h->S.n) || !dev_alloc(h, &h->obs_cnt_mem, 2)) return 1;
    }
    if (out_dev) HIPCHK(h, hipMemsetAsync(h->obs_cnt_mem, 0, 8, h->stream));
    h->S.obs_list = out_dev ? h->obs_list_mem : nullptr; h->S.obs_cnt = out_dev ? h->obs_cnt_mem : nullptr;
    h->S.bound_gray = (out_dev && kind == 0) ? out_dev : nullptr;  // (k_step's mirror update writes a gray image's changed pixels itself)
    if (h->S.bound_gray && !h->gray_lut_mem) {  // ... through this table: the value k_obs encodes a glyph to (rg_obs.hip `lutf`: the same single IEEE division)
        float lut[128];
        for (uint32_t g = 0; g < 128; g++) lut[g] = (float)(uint8_t)host_tile_to_sym(g) / (float)(uint8_t)h->cfg.symbols;
        if (!dev_alloc(h, &h->gray_lut_mem, 128)) return 1;
        HIPCHK(h, hipMemcpy(h->gray_lut_mem, lut, sizeof lut, hipMemcpyHostToDevice));
    }
    h->S.gray_lut = h->gray_lut_mem;
    h->bound_out = out_dev; h->bound_kind = kind; h->bound_valid = false; h->bound_steps = 0;
    return 0;
}
int rg_obs_gray(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev) { return obs_common(h, status_flag, with_hist, 0, out_dev); }
int rg_obs_symbol(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev) { return obs_common(h, status_flag, with_hist, 1, out_dev); }

int rg_fetch_states(rg_t *h, uint8_t *screen, uint8_t *hist, int32_t *status, uint32_t *flags) {
    HIPCHK(h, hipSetDevice(h->device));
    if (flush_mirrors(h)) return 1;
    size_t n = (size_t)h->S.n, hw = (size_t)h->S.hw;
    if (h->mixed) {  // ragged env-order layout: env i's H_i x W_i bytes at ragged_off[i] (rg_env_dims); group by group, rows scattered on the host
        std::vector<uint8_t> tmp;
        for (rg_handle *sh : h->sub) {
            const size_t m = (size_t)sh->S.n, ghw = (size_t)sh->S.hw;
            for (int which = 0; which < 2; which++) {
                uint8_t *dst = which ? hist : screen;
                if (!dst) continue;
                tmp.resize(m * ghw);
                HIPCHK(h, hipMemcpyAsync(tmp.data(), which ? sh->S.hist : sh->S.screen, m * ghw, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                for (size_t l = 0; l < m; l++) memcpy(dst + h->ragged_off[sh->ext[l]], tmp.data() + l * ghw, ghw);
            }
        }
        screen = hist = nullptr;
    }
    if (screen) HIPCHK(h, hipMemcpyAsync(screen, h->S.screen, n * hw, hipMemcpyDeviceToHost, h->stream));
    if (hist) HIPCHK(h, hipMemcpyAsync(hist, h->S.hist, n * hw, hipMemcpyDeviceToHost, h->stream));
    if (status) HIPCHK(h, hipMemcpyAsync(status, h->S.status, n * 40, hipMemcpyDeviceToHost, h->stream));
    if (flags) HIPCHK(h, hipMemcpyAsync(flags, h->S.flags, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ParallelGameState::step (python/src/lib.rs:315-321) for a SMALL batch, where a step is launch and copy overhead and nothing else: rg_step_prefix +
// rg_sync + rg_fetch_states were a key upload, four device-to-host copies, an error-word copy and five stream waits -- ~145 us per step whatever the batch
// (profiles/r04_value_api.txt: 64 envs 0.44 M env-steps/s, below the CPU port's 0.60 M).  Here: the keys are read by k_step from pinned host memory, the
// mirror refresh follows, ONE kernel (k_export) writes status, flags, the error word -- and the screens, to `screen` / `hist` if given -- straight to their
// destinations, and the call waits for the handle's stream once.  The background generator's streams are not waited for (rg_sync / rg_destroy do).
// status / flags (and screen / hist, each n_env * H * W bytes, or NULL for neither) must be device-visible: pinned host memory from rg_host_alloc, or
// device memory (a snapshot).  Not for handles with config groups.  Errors are reported as by rg_sync.
int rg_step_fetch(rg_t *h, const uint8_t *keys_host, int n_keys, uint8_t *screen, uint8_t *hist, int32_t *status, uint32_t *flags) {
    if (refuse_mixed(h, "rg_step_fetch")) return 1;
    if (!h->sub.empty()) { h->err = "rg_step_fetch: not for a handle with config groups (rg_step_prefix + rg_fetch_states)"; return 1; }
    if (!status || !flags || (screen == nullptr) != (hist == nullptr)) { h->err = "rg_step_fetch: status and flags are required, screen and hist come together"; return 1; }
    if (h->S.hw & 3) { h->err = "rg_step_fetch needs H*W divisible by 4"; return 1; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n_keys < 0) { h->err = "rg_step_fetch: negative key count"; return 1; }
    if (n_keys > h->S.n) n_keys = h->S.n;
    // (one test per buffer: a failed second allocation must not leave the first behind as "both are there")
    if (!h->pin_keys) HIPCHK(h, hipHostMalloc((void **)&h->pin_keys, (size_t)h->S.n + 16, hipHostMallocDefault));
    if (!h->pin_err) HIPCHK(h, hipHostMalloc((void **)&h->pin_err, 16, hipHostMallocDefault));
    memcpy(h->pin_keys, keys_host, (size_t)n_keys);
    if (rg_step_prefix(h, h->pin_keys, n_keys, 1)) return 1;  // ("on device": device-visible)
    if (flush_render(h)) return 1;
    rgk_export(&h->S, h->d_err, screen, hist, status, flags, h->pin_err, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t err = h->pin_err[0];
    if (err) {
        if (err & RG_FLAG_ERR_INTERNAL) h->err = "internal capacity guard of the HIP stepper tripped (please report the config)";
        else if (err & RG_FLAG_ERR_KEY) h->err = "Invalid input (key is not in the ai keymap)";
        else if (err & RG_FLAG_ERR_DEAD) h->err = "Ignored input (action while the player is dead)";
        else h->err = "Invalid tile in symbol image (symbol >= symbols - 1)";
        return 1;
    }
    return 0;
}

// Device-side snapshots for the value-object API.  ParallelGameState::step hands back Vec<PlayerState> -- values that stay valid when the envs move
// on -- but a caller of the RL loop reads only gold / is_terminal of every state (parallel.py:59-64): shipping 1 KB of screen + history per env
// over PCIe on every step was what bounded `ParallelRogueEnv.step` (29.5 M env-steps/s at 65 536 envs).  rg_snapshot_take copies the two mirrors
// device-to-device (67 MB: ~30 us) into a caller-owned device buffer; the host copies are made on first access (rg_dev_read), whole or per env.
int rg_dev_alloc(int device, size_t bytes, void **out) {
    if (hipSetDevice(device) != hipSuccess || hipMalloc(out, bytes ? bytes : 16) != hipSuccess) { g_create_err = "hipMalloc failed (" + std::to_string(bytes) + " bytes)"; return 1; }
    return 0;
}
void rg_dev_free(int device, void *p) { if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p); }
int rg_snapshot_take(rg_t *h, void *dev) {
    if (refuse_mixed(h, "rg_snapshot_take")) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (flush_mirrors(h)) return 1;
    const size_t bytes = (size_t)h->S.n * (size_t)h->S.hw;
    HIPCHK(h, hipMemcpyAsync(dev, h->S.screen, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(static_cast<uint8_t *>(dev) + bytes, h->S.hist, bytes, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}
int rg_dev_read_rows(rg_t *h, const void *dev_src, size_t src_pitch, void *host_dst, size_t row_bytes, int rows) {  // `rows` rows of row_bytes, src_pitch apart -> packed
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(host_dst, row_bytes, dev_src, src_pitch, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}
int rg_dev_read(rg_t *h, const void *dev_src, void *host_dst, size_t bytes) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// Device scratch of the stateless encode (rg_encode_host*), one per device, grown on demand and kept: the value-object API calls this once per
// PlayerState.gray_image() &c., and a hipMalloc / hipFree pair per call cost more than the encode itself.
namespace {
struct EncodeScratch { uint8_t *buf = nullptr; size_t cap = 0; uint32_t *err = nullptr; };
EncodeScratch g_scratch[64];
std::mutex g_scratch_mu;
bool scratch_reserve(int device, size_t bytes) {
    EncodeScratch &sc = g_scratch[device];
    if (!sc.err && hipMalloc((void **)&sc.err, 16) != hipSuccess) return false;
    if (sc.cap >= bytes) return true;
    if (sc.buf) (void)hipFree(sc.buf);
    sc.buf = nullptr; sc.cap = 0;
    size_t want = bytes + bytes / 2;
    if (hipMalloc((void **)&sc.buf, want) != hipSuccess) return false;
    sc.cap = want;
    return true;
}
}  // namespace

int rg_encode_host_batch(int device, int n, const uint8_t *screen, const uint8_t *hist, const int32_t *status, int height, int width, int symbols,
                         uint32_t status_flag, int with_hist, int kind, float *out_host) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device available (librogue_gym_hip has no CPU fallback)"; return 1; }
    if (device < 0 || device >= ndev || device >= 64 || hipSetDevice(device) != hipSuccess) { g_create_err = "invalid HIP device"; return 1; }
    if (n <= 0) return 0;
    const size_t hw = (size_t)height * width, hw4 = (hw + 3) & ~(size_t)3;
    status_flag &= 0x1ffu;
    const int c = (kind ? symbols : 1) + __builtin_popcount(status_flag) + (with_hist ? 1 : 0);
    // scratch layout: screens [n][hw4] | hists [n][hw4] | status [n][10] i32 | out [n][c][hw] f32
    const size_t o_scr = 0, o_hist = (size_t)n * hw4, o_st = 2 * (size_t)n * hw4, o_out = (o_st + (size_t)n * 40 + 255) & ~(size_t)255;
    const size_t out_bytes = (size_t)n * c * hw * 4;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!scratch_reserve(device, o_out + out_bytes)) { g_create_err = "hipMalloc failed"; return 1; }
    EncodeScratch &sc = g_scratch[device];
    uint32_t err = 0;
    bool ok = hipMemcpy2D(sc.buf + o_scr, hw4, screen, hw, hw, (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(sc.buf + o_st, status, (size_t)n * 40, hipMemcpyHostToDevice) == hipSuccess && hipMemset(sc.err, 0, 4) == hipSuccess &&
              (!with_hist || hipMemcpy2D(sc.buf + o_hist, hw4, hist, hw, hw, (size_t)n, hipMemcpyHostToDevice) == hipSuccess);
    if (!ok) { g_create_err = "hipMemcpy failed"; return 1; }
    rgk_encode(sc.buf + o_scr, sc.buf + o_hist, reinterpret_cast<const int32_t *>(sc.buf + o_st), nullptr, sc.err, n, (int)hw, hw4, 10, symbols, symbols, status_flag,
               with_hist ? 1 : 0, kind, reinterpret_cast<float *>(sc.buf + o_out), nullptr, nullptr);
    if (hipMemcpy(out_host, sc.buf + o_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&err, sc.err, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        g_create_err = "encode kernel failed"; return 1;
    }
    if (err) { g_create_err = "Invalid tile in symbol image (symbol >= symbols - 1)"; return 1; }
    return 0;
}

int rg_encode_host(int device, const uint8_t *screen, const uint8_t *hist, const int32_t *status, int height, int width, int symbols,
                   uint32_t status_flag, int with_hist, int kind, float *out_host) {
    return rg_encode_host_batch(device, 1, screen, hist, status, height, width, symbols, status_flag, with_hist, kind, out_host);
}

// PlayerState images of the whole batch into host memory (the value-object API's batched path): the fused k_obs pass into a device scratch
// kept by the handle, then one D2H copy.  out_host should be pinned (rg_host_alloc) for full PCIe rate.
int rg_obs_host(rg_t *h, int kind, uint32_t status_flag, int with_hist, float *out_host) {
    if (refuse_mixed(h, "rg_obs_host")) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)h->S.n * rg_obs_channels(h, kind, status_flag, with_hist) * h->S.hw * 4;
    if (h->obs_scratch_cap < bytes) {
        if (h->obs_scratch) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->obs_scratch); h->obs_scratch = nullptr; h->obs_scratch_cap = 0; }
        HIPCHK(h, hipMalloc((void **)&h->obs_scratch, bytes));
        h->obs_scratch_cap = bytes;
    }
    if (kind ? rg_obs_symbol(h, status_flag, with_hist, h->obs_scratch) : rg_obs_gray(h, status_flag, with_hist, h->obs_scratch)) return 1;
    HIPCHK(h, hipMemcpyAsync(out_host, h->obs_scratch, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int rg_host_alloc(size_t bytes, void **out) {
    if (hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { g_create_err = "hipHostMalloc failed"; return 1; }
    return 0;
}
void rg_host_free(void *p) { if (p) (void)hipHostFree(p); }

int rg_compact_record_bytes(const rg_t *h, int with_hist) { return h->S.hw + RG_COMPACT_FIXED_BYTES + (with_hist ? h->S.hw : 0); }

int rg_pack_compact(rg_t *h, int with_hist, uint8_t *out_dev) {
    if (refuse_mixed(h, "rg_pack_compact")) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->S.hw & 3) { h->err = "rg_pack_compact needs H*W divisible by 4"; return 1; }
    if (flush_mirrors(h)) return 1;
    rgk_pack(&h->S, with_hist ? 1 : 0, out_dev, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The one collective of the sharded path behind the C-ABI: RCCL over xGMI.  librccl is bound at run time (dlopen on first use):
// a single-GPU user of this library needs no RCCL, and inside a PyTorch process the soname resolves to the copy torch already loaded, so
// there is one RCCL per process.
// ---------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional (rg_comm_count)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;  // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("librccl not found: ") + dlerror(); return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) r.err = "librccl lacks an expected symbol";
    });
    return &r;
}
}  // namespace

int rg_comm_unique_id(uint8_t id[128]) {
    Rccl *r = rccl();
    if (!r->err.empty()) { g_create_err = r->err; return 1; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    ncclUniqueId u;
    ncclResult_t e = r->GetUniqueId(&u);
    if (e != ncclSuccess) { g_create_err = std::string("ncclGetUniqueId: ") + r->GetErrorString(e); return 1; }
    memcpy(id, &u, 128);
    return 0;
}

int rg_comm_init(rg_t *h, const uint8_t id[128], int rank, int world) {
    Rccl *r = rccl();
    if (!r->err.empty()) { h->err = r->err; return 1; }
    if (world < 1 || rank < 0 || rank >= world) { h->err = "rg_comm_init: invalid rank / world"; return 1; }
    if (h->comm) { h->err = "rg_comm_init: the handle already has a communicator"; return 1; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t e = r->CommInitRank(&h->comm, world, u, rank);
    if (e != ncclSuccess) { h->comm = nullptr; h->err = std::string("ncclCommInitRank: ") + r->GetErrorString(e); return 1; }
    h->comm_rank = rank; h->comm_world = world;
    return 0;
}

// teardown (rg_destroy): ncclCommAbort where the library has it -- ncclCommDestroy may block when peer ranks have already exited
static int comm_release(rg_handle *h, bool teardown) {
    if (!h->comm) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    Rccl *r = rccl();
    ncclResult_t e = (teardown && r->CommAbort) ? r->CommAbort(h->comm) : r->CommDestroy(h->comm);
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0;
    if (e != ncclSuccess) { h->err = std::string("ncclCommDestroy: ") + r->GetErrorString(e); return 1; }
    return 0;
}
int rg_comm_destroy(rg_t *h) { return comm_release(h, false); }

// What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what rg_comm_init was told: the proof that RCCL saw `world` ranks.
int rg_comm_count(rg_t *h, int *count, int *rank) {
    if (!h->comm) { h->err = "rg_comm_count: no communicator (rg_comm_init first)"; return 1; }
    Rccl *r = rccl();
    if (!r->CommCount || !r->CommUserRank) { h->err = "rg_comm_count: librccl lacks ncclCommCount / ncclCommUserRank"; return 1; }
    int c = 0, u = 0;
    ncclResult_t e = r->CommCount(h->comm, &c);
    if (e == ncclSuccess) e = r->CommUserRank(h->comm, &u);
    if (e != ncclSuccess) { h->err = std::string("ncclCommCount: ") + r->GetErrorString(e); return 1; }
    if (count) *count = c;
    if (rank) *rank = u;
    return 0;
}

int rg_allgather_compact(rg_t *h, int with_hist, uint8_t *out_dev) {
    if (!h->comm) { h->err = "rg_allgather_compact: no communicator (rg_comm_init first)"; return 1; }
    const size_t bytes = (size_t)h->S.n * (size_t)rg_compact_record_bytes(h, with_hist);
    // pack straight into this rank's slice of the gathered batch, then gather IN PLACE (sendbuff == recvbuff + rank * count) on the handle's stream:
    // no staging copy, and the collective is ordered behind the step like any other launch of the handle
    uint8_t *mine = out_dev + (size_t)h->comm_rank * bytes;
    if (rg_pack_compact(h, with_hist, mine)) return 1;
    ncclResult_t e = rccl()->AllGather(mine, out_dev, bytes, ncclUint8, h->comm, h->stream);
    if (e != ncclSuccess) { h->err = std::string("ncclAllGather: ") + rccl()->GetErrorString(e); return 1; }
    return 0;
}

int rg_expand_compact(rg_t *h, const uint8_t *packed_dev, int n, int packed_has_hist, int kind, uint32_t status_flag, int with_hist, float *out_dev) {
    if (refuse_mixed(h, "rg_expand_compact")) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->S.hw & 3) { h->err = "rg_expand_compact needs H*W divisible by 4"; return 1; }
    if (with_hist && !packed_has_hist) { h->err = "rg_expand_compact: the packed batch carries no history plane"; return 1; }
    const size_t hw = (size_t)h->S.hw, rec = hw + RG_COMPACT_FIXED_BYTES + (packed_has_hist ? hw : 0);
    rgk_encode(packed_dev, packed_dev + RG_COMPACT_HIST_OFFSET(hw), reinterpret_cast<const int32_t *>(packed_dev + RG_COMPACT_STATUS_OFFSET(hw)), nullptr, h->d_err, n, (int)hw, rec, rec / 4, h->cfg.symbols,
               h->cfg.symbols, status_flag & 0x1ffu, with_hist ? 1 : 0, kind, out_dev, nullptr, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int rg_status_vec(rg_t *h, uint32_t status_flag, int32_t *out_host) {
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->S.n;
    std::vector<int32_t> st(n * 10);
    HIPCHK(h, hipMemcpyAsync(st.data(), h->S.status, n * 40, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    static const int order[9] = {0, 2, 3, 4, 5, 6, 7, 8, 9};  // StatusFlagInner::to_vector (flags.rs:67-87): gold is not part of it
    size_t k = 0;
    for (size_t e = 0; e < n; e++)
        for (int b = 0; b < 9; b++)
            if (status_flag & (1u << b)) out_host[k++] = st[e * 10 + order[b]];
    return 0;
}

// ---- action-history log (GameState::dump_history, python/src/lib.rs:245-250 -> RunTime::saved_inputs_as_json, core/src/lib.rs:357-375) ----
int rg_history_enable(rg_t *h, int cap_per_env) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) { for (rg_handle *sh : h->sub) SUBCHK(h, sh, rg_history_enable(sh, cap_per_env)); return 0; }
    if (cap_per_env <= 0) { h->err = "rg_history_enable: capacity must be positive"; return 1; }
    if (h->S.klog) { h->err = "rg_history_enable: already enabled"; return 1; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }
    const size_t n = (size_t)h->S.n;
    uint8_t *log = nullptr, *cur = nullptr; uint32_t *len = nullptr;
    if (!dev_alloc(h, &log, n * 2 * (size_t)cap_per_env) || !dev_alloc(h, &len, 2 * n) || !dev_alloc(h, &cur, n)) return 1;
    h->S.klog = log; h->S.klog_len = len; h->S.klog_cur = cur; h->S.klog_cap = cap_per_env;
    h->SP.klog = nullptr;  // the spare view never logs
    return 0;
}

int rg_history_keys(rg_t *h, int env, int which, uint8_t *keys, size_t cap, uint32_t *len) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {
        if (env < 0 || env >= h->S.n) { h->err = "rg_history_keys: env / which out of range"; return 1; }
        rg_handle *sh = h->sub[h->g_of[env]];
        const int rc = rg_history_keys(sh, h->l_of[env], which, keys, cap, len);
        if (rc) h->err = sh->err;
        return rc;
    }
    if (!h->S.klog) { h->err = "the action history is not enabled (rg_history_enable)"; return 1; }
    if (env < 0 || env >= h->S.n || (which != 0 && which != 1)) { h->err = "rg_history_keys: env / which out of range"; return 1; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->S.n;
    uint8_t cur = 0;
    HIPCHK(h, hipMemcpy(&cur, h->S.klog_cur + env, 1, hipMemcpyDeviceToHost));
    const uint32_t buf = which ? (cur ^ 1u) : cur;
    uint32_t l = 0;
    HIPCHK(h, hipMemcpy(&l, h->S.klog_len + buf * n + env, 4, hipMemcpyDeviceToHost));
    if (len) *len = l;
    if (l > (uint32_t)h->S.klog_cap) { h->err = "action history truncated: " + std::to_string(l) + " keys in the episode, capacity " + std::to_string(h->S.klog_cap); return 2; }
    if (keys) {
        if (cap < l) { h->err = "rg_history_keys: buffer too small"; return 3; }
        if (l) HIPCHK(h, hipMemcpy(keys, h->S.klog + ((size_t)env * 2 + buf) * h->S.klog_cap, l, hipMemcpyDeviceToHost));
    }
    return 0;
}

// serde form of InputCode for a key of KeyMap::ai (input.rs:73-100; enum Action / Direction names), pretty-printed like
// serde_json::to_string_pretty (4-space indent), the format of data/learned/*/best-actions.json
static void append_input_code(std::string &s, uint8_t key) {
    static const char *dirs[8] = {"Up", "Down", "Left", "Right", "LeftUp", "RightUp", "LeftDown", "RightDown"};
    const char lower = (char)(key | 0x20);
    int d = -1;
    switch (lower) { case 'k': d = 0; break; case 'j': d = 1; break; case 'h': d = 2; break; case 'l': d = 3; break;
                     case 'y': d = 4; break; case 'u': d = 5; break; case 'b': d = 6; break; case 'n': d = 7; break; }
    s += "    {\n        \"Act\": ";
    if (d >= 0 && key != '.' && key != '>' ) {
        s += std::string("{\n            \"") + ((key & 0x20) ? "Move" : "MoveUntil") + "\": \"" + dirs[d] + "\"\n        }";
    } else s += std::string("\"") + (key == '.' ? "NoOp" : key == 's' ? "Search" : "DownStair") + "\"";
    s += "\n    }";
}

int rg_dump_history(rg_t *h, int env, int which, char *buf, size_t cap, size_t *needed) {
    uint32_t len = 0;
    int rc = rg_history_keys(h, env, which, nullptr, 0, &len);
    if (rc) return rc;
    std::vector<uint8_t> keys(len ? len : 1);
    rc = rg_history_keys(h, env, which, keys.data(), keys.size(), &len);
    if (rc) return rc;
    std::string s = len ? "[\n" : "[]";
    for (uint32_t i = 0; i < len; i++) { append_input_code(s, keys[i]); s += i + 1 < len ? ",\n" : "\n"; }
    if (len) s += "]";
    if (needed) *needed = s.size() + 1;
    if (!buf) return 0;
    if (s.size() + 1 > cap) { h->err = "rg_dump_history: buffer too small"; return 3; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

int rg_counters_ex(rg_t *h, uint64_t *out, int n_out, int reset) {
    HIPCHK(h, hipSetDevice(h->device));
    if (n_out < 0 || n_out > RG_STAT_COLS) { h->err = "rg_counters_ex: at most 16 counters"; return 1; }
    if (!h->sub.empty()) {
        if (out) for (int k = 0; k < n_out; k++) out[k] = 0;
        for (rg_handle *sh : h->sub) {
            uint64_t o[RG_STAT_COLS];
            SUBCHK(h, sh, rg_counters_ex(sh, o, n_out, reset));
            if (out) for (int k = 0; k < n_out; k++) out[k] += o[k];
        }
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->S.stats) { if (out) memset(out, 0, 8 * (size_t)n_out); return 0; }
    if (out) {
        std::vector<unsigned long long> rows(RG_STAT_COLS * h->stat_rows);
        HIPCHK(h, hipMemcpy(rows.data(), h->S.stats, rows.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < n_out; k++) out[k] = 0;
        for (size_t r = 0; r < h->stat_rows; r++)
            for (int k = 0; k < n_out; k++) out[k] += rows[r * RG_STAT_COLS + k];
    }
    if (reset) HIPCHK(h, hipMemset(h->S.stats, 0, 8 * RG_STAT_COLS * h->stat_rows));
    return 0;
}
int rg_counters(rg_t *h, uint64_t out[8], int reset) { return rg_counters_ex(h, out, 8, reset); }

int rg_probe_sclk(rg_t *h, double *mhz) {
    HIPCHK(h, hipSetDevice(h->device));
    rgk_probe_clock(h->d_probe, 20000, h->stream);  // ~40 k dependent VALU instructions: 50-100 us
    HIPCHK(h, hipGetLastError());
    unsigned long long r[4] = {0, 0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(r, h->d_probe, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (mhz) *mhz = r[1] ? 100.0 * (double)r[0] / (double)r[1] : 0.0;  // s_memrealtime ticks at a constant 100 MHz
    return 0;
}

// development aid: in-kernel phase trace; out = [ceil(n_env / 64)][64] words, one row per wave of the LAST k_step / k_build launch
// (word 0 = record count, words 1.. = phase << 48 | ticks, word 63 = whole-wave ticks)
int rg_prof(rg_t *h, int enable, unsigned long long *out) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) { h->err = "rg_prof: not available for a batch with several config groups"; return 1; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t words = (size_t)((h->S.n + 15) / 16) * 64;  // k_step runs 16..64 envs per wave
    if (out && h->S.prof) HIPCHK(h, hipMemcpy(out, h->S.prof, words * 8, hipMemcpyDeviceToHost));
    if (enable && !h->S.prof) { if (!dev_alloc(h, &h->S.prof, words)) return 1; }
    if (h->S.prof) HIPCHK(h, hipMemset(h->S.prof, 0, words * 8));
    if (!enable) h->S.prof = nullptr;
    return 0;
}

int rg_timing_enable(rg_t *h, int on) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) { for (rg_handle *sh : h->sub) SUBCHK(h, sh, rg_timing_enable(sh, on)); return 0; }
    if (on && h->ev[0].empty())
        for (int k = 0; k < RG_TIMED_KERNELS; k++) {
            h->ev[k].resize(2 * RG_TIMING_MAX);
            for (auto &e : h->ev[k]) HIPCHK(h, hipEventCreate(&e));
        }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }
    for (int k = 0; k < RG_TIMED_KERNELS; k++) { h->ev_used[k] = 0; h->timing_seq[k] = 0; }
    h->timing = on != 0;
    h->timing_stride = on > 1 ? (uint64_t)on : 1;  // on = N > 1: bracket every N-th launch only
    return 0;
}

int rg_timing_read_all(rg_t *h, int n, double *ms, uint64_t *sampled, uint64_t *launches) {
    HIPCHK(h, hipSetDevice(h->device));
    if (n > RG_TIMED_KERNELS) n = RG_TIMED_KERNELS;
    if (!h->sub.empty()) {  // sums over the groups: the counts then are group launches
        for (int k = 0; k < n; k++) { ms[k] = 0; sampled[k] = 0; if (launches) launches[k] = 0; }
        for (rg_handle *sh : h->sub) {
            double m[RG_TIMED_KERNELS]; uint64_t l[RG_TIMED_KERNELS], a[RG_TIMED_KERNELS];
            SUBCHK(h, sh, rg_timing_read_all(sh, n, m, l, a));
            for (int k = 0; k < n; k++) { ms[k] += m[k]; sampled[k] += l[k]; if (launches) launches[k] += a[k]; }
        }
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }  // (k_regen's pairs are stamped on the side stream)
    for (int k = 0; k < n; k++) {
        double sum = 0;
        for (size_t i = 0; i + 1 < h->ev_used[k]; i += 2) {
            float t = 0;
            HIPCHK(h, hipEventElapsedTime(&t, h->ev[k][i], h->ev[k][i + 1]));
            sum += t;
        }
        ms[k] = sum; sampled[k] = h->ev_used[k] / 2;
        if (launches) launches[k] = h->timing_seq[k];
        h->ev_used[k] = 0; h->timing_seq[k] = 0;
    }
    return 0;
}
int rg_timing_read(rg_t *h, double ms[4], uint64_t launches[4]) { return rg_timing_read_all(h, 4, ms, launches, nullptr); }
int rg_timing_read_samples(rg_t *h, int kernel, float *ms, int cap, int *n) {
    if (!ms || !n || cap < 0 || kernel < -1 || kernel >= RG_TIMED_KERNELS) { h->err = "rg_timing_read_samples: invalid arguments"; return 1; }
    if (!h->sub.empty()) { h->err = "rg_timing_read_samples: not for a handle with config groups"; return 1; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) { HIPCHK(h, hipStreamSynchronize(h->side)); HIPCHK(h, hipStreamSynchronize(h->side2)); HIPCHK(h, hipStreamSynchronize(h->side3)); }
    const int ka = kernel < 0 ? 0 : kernel, kb = kernel < 0 ? 2 : kernel;  // (a step: k_step's begin .. the observation pass's end)
    size_t pairs = (h->ev_used[ka] < h->ev_used[kb] ? h->ev_used[ka] : h->ev_used[kb]) / 2;
    if (pairs > (size_t)cap) pairs = (size_t)cap;
    for (size_t i = 0; i < pairs; i++) HIPCHK(h, hipEventElapsedTime(&ms[i], h->ev[ka][2 * i], h->ev[kb][2 * i + 1]));
    *n = (int)pairs;
    return 0;
}

int rg_dump_config(const rg_t *h, int env, char *buf, size_t cap) {
    if (env < 0 || env >= h->S.n) return 1;
    if (!h->sub.empty()) return rg_dump_config(h->sub[h->g_of[env]], h->l_of[env], buf, cap);
    RgParsed p = h->parsed;  // the group's config with this env's own seed / seed range
    const size_t n = (size_t)h->S.n;
    p.has_seed_range = !h->range_span.empty() && (h->range_span[env] | h->range_span[n + env]) != 0;
    if (p.has_seed_range) {
        p.seed_range[0] = ((unsigned __int128)h->range_lo[n + env] << 64) | h->range_lo[env];
        p.seed_range[1] = p.seed_range[0] + (((unsigned __int128)h->range_span[n + env] << 64) | h->range_span[env]);
    }
    std::string s = rg_dump_config_json(p, h->seed_lo[env], h->seed_hi[env], !h->reseed[env]);
    if (s.size() + 1 > cap) return 1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

int rg_config_schema(char *buf, size_t cap, size_t *needed) {
    std::string s = rg_config_schema_json();
    if (needed) *needed = s.size() + 1;
    if (!buf || s.size() + 1 > cap) return buf ? 1 : 0;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

int rg_config_resolved(const char *cfg_json, char *buf, size_t cap) {
    RgParsed p;
    std::string e = rg_parse_config(cfg_json, &p);
    if (!e.empty()) { g_create_err = "Failed to parse config: " + e; return 1; }
    const RgConfig &c = p.cfg;
    std::string s = "{\"weapon\": {\"times\": " + std::to_string(c.wpn_times) + ", \"max\": " + std::to_string(c.wpn_max) + ", \"hit_plus\": " +
                    std::to_string(c.wpn_hit_plus) + ", \"dam_plus\": " + std::to_string(c.wpn_dam_plus) + "}, \"armor_def\": " + std::to_string(c.armor_def) +
                    ", \"init_gold\": " + std::to_string(c.init_gold) + ", \"can_pickup\": " + (c.can_pickup ? "true" : "false") + ", \"init_draws\": [";
    for (size_t i = 0; i + 1 < p.init_draws.size(); i += 2)
        s += std::string(i ? ", " : "") + "[" + std::to_string(p.init_draws[i]) + ", " + std::to_string(p.init_draws[i + 1]) + "]";
    s += "], \"symbols\": " + std::to_string(c.symbols) + ", \"n_enemies\": " + std::to_string(c.n_enemies) + "}";
    if (s.size() + 1 > cap) { g_create_err = "rg_config_resolved: buffer too small"; return 1; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

int rg_config_canonical(const char *cfg_json, char *buf, size_t cap) {
    RgParsed p;
    std::string e = rg_parse_config(cfg_json, &p);
    if (!e.empty()) { g_create_err = "Failed to parse config: " + e; return 1; }
    std::string s = rg_dump_config_json(p, p.seed_lo, p.seed_hi, p.has_seed);
    if (s.size() + 1 > cap) { g_create_err = "buffer too small"; return 1; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

#ifdef RG_DEV_KNOBS
// development library only: the spare pipeline's state words [sp_slots][n_env] -> host (tools/tmp/sp_state.py)
int rg_dev_sp_ready(rg_t *h, uint32_t *out_host, int *slots) {
    HIPCHK(h, hipSetDevice(h->device));
    *slots = h->S.sp_slots;
    HIPCHK(h, hipMemcpy(out_host, h->S.sp_ready, (size_t)h->S.n * h->S.sp_slots * 4, hipMemcpyDeviceToHost));
    return 0;
}
#endif
int rg_debug_descend(rg_t *h) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->sub.empty()) {
        for (rg_handle *sh : h->sub) SUBCHK(h, sh, rg_debug_descend(sh));
        h->screens_stale = true;
        return assemble_small(h);
    }
    if (flush_render(h)) return 1;
    h->S.stair_gen = h->stair_gen++;
    rgk_debug_descend(&h->S, &h->cfg, h->stream);
    HIPCHK(h, hipGetLastError());
    h->render_pending = true;
    return 0;
}

int rg_debug_fetch(rg_t *h, int env, rg_debug_state *out, uint16_t *cells) {
    HIPCHK(h, hipSetDevice(h->device));
    if (env < 0 || env >= h->S.n) { h->err = "env out of range"; return 1; }
    if (!h->sub.empty()) {
        rg_handle *sh = h->sub[h->g_of[env]];
        SUBCHK(h, sh, rg_debug_fetch(sh, h->l_of[env], out, cells));
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const RgState &S = h->S;
    const size_t n = (size_t)S.n, e = (size_t)env;
    memset(out, 0, sizeof *out);
#define GET1(dst, src) HIPCHK(h, hipMemcpy(&(dst), (src) + e, sizeof(dst), hipMemcpyDeviceToHost))
    uint16_t pos; int32_t i32; uint32_t u32;
    GET1(pos, S.p_pos); out->px = pos >> 8; out->py = pos & 0xff;
    GET1(i32, S.p_hp); out->hp = i32; GET1(i32, S.p_hpmax); out->hp_max = i32; GET1(i32, S.p_lvl); out->player_level = i32;
    GET1(u32, S.dlevel); out->dungeon_level = (int32_t)u32;
    GET1(out->exp, S.p_exp); GET1(out->food_left, S.food); GET1(out->quiet, S.quiet); GET1(out->pack_gold, S.pack_gold); GET1(out->steps, S.steps);
    for (int k = 0; k < 12; k++) HIPCHK(h, hipMemcpy(&out->rng[k], S.rng + k * n + e, 4, hipMemcpyDeviceToHost));
    int nm = 0, ng = 0;
    // monsters sorted by (x, y)
    struct M { uint32_t w; int32_t hp; uint32_t exp; } ms[RG_MAX_ROOMS];
    int cnt = 0;
    const int n_slots = h->cfg.room_num_x * h->cfg.room_num_y;  // one monster / gold slot per room (floor.rs:106-153)
    for (int s = 0; s < n_slots; s++) {
        uint32_t w; HIPCHK(h, hipMemcpy(&w, S.mon_w0 + s * n + e, 4, hipMemcpyDeviceToHost));
        if (!((w >> 24) & MF_ALIVE)) continue;
        ms[cnt].w = w;
        HIPCHK(h, hipMemcpy(&ms[cnt].hp, S.mon_hp + s * n + e, 4, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(&ms[cnt].exp, S.mon_exp + s * n + e, 4, hipMemcpyDeviceToHost));
        cnt++;
    }
    for (int i = 1; i < cnt; i++) { M v = ms[i]; int j = i; while (j > 0 && (ms[j - 1].w & 0xffff) > (v.w & 0xffff)) { ms[j] = ms[j - 1]; j--; } ms[j] = v; }
    for (int i = 0; i < cnt; i++) {
        out->mon_x[nm] = (ms[i].w >> 8) & 0xff; out->mon_y[nm] = ms[i].w & 0xff; out->mon_type[nm] = h->cfg.mon[(ms[i].w >> 16) & 0xff].tile - 'A';  /* reported as glyph index so builtin monsters keep their builtin id */
        out->mon_active[nm] = ((ms[i].w >> 24) & MF_ACTIVE) ? 1 : 0; out->mon_hp[nm] = ms[i].hp; out->mon_exp[nm] = ms[i].exp; nm++;
    }
    out->n_monsters = nm;
    for (int s = 0; s < n_slots; s++) {
        uint32_t g; HIPCHK(h, hipMemcpy(&g, S.gold_pos + s * n + e, 4, hipMemcpyDeviceToHost));
        if (!(g & 0x10000u)) continue;
        out->gold_x[ng] = (g >> 8) & 0xff; out->gold_y[ng] = g & 0xff;
        uint32_t a; HIPCHK(h, hipMemcpy(&a, S.gold_amt + s * n + e, 4, hipMemcpyDeviceToHost));
        out->gold_amount[ng] = (int32_t)a; ng++;
    }
    out->n_gold = ng;
    out->n_rooms = h->cfg.room_num_x * h->cfg.room_num_y;
    for (int s = 0; s < out->n_rooms; s++) {
        uint8_t m;
        HIPCHK(h, hipMemcpy(&out->room_rect[s], S.room_rect + s * n + e, 4, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(&m, S.room_meta + s * n + e, 1, hipMemcpyDeviceToHost));
        out->room_meta[s] = m;
    }
    if (cells) HIPCHK(h, hipMemcpy(cells, S.cell + e * S.hw, (size_t)S.hw * 2, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
