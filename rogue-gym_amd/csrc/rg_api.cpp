// rg_api.cpp -- C-ABI host of librogue_gym_hip.so (see include/rogue_gym_hip.h).
// Owns the HBM state, parses configs, sequences the kernels on one HIP stream.  There is no CPU
// fallback: without a HIP device every entry point that needs one fails loudly.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/rogue_gym_hip.h"
#include "rg_state.h"

extern "C" {
void rgk_build(const RgState *S, const RgConfig *c, hipStream_t st);
void rgk_step(const RgState *S, const RgState *SP, const RgConfig *c, const uint8_t *keys, uint32_t *err_any, int use_spares, hipStream_t st);
void rgk_regen(const RgState *SP, const RgConfig *c, hipStream_t st);
void rgk_render(const RgState *S, const RgConfig *c, hipStream_t st);
void rgk_encode(const uint8_t *screen, const uint8_t *hist, const int32_t *status, uint32_t *flags, uint32_t *err_any, int n, int hw, int symbols,
                uint32_t sflag, int with_hist, int kind, float *out, hipStream_t st);
int rgk_obs(const RgState *S, const RgConfig *c, uint32_t sflag, int with_hist, int kind, float *out, uint32_t *err_any, hipStream_t st);
}

struct rg_handle {
    RgParsed parsed;             // config of env 0 (all envs agree except for the seed)
    RgConfig cfg;
    RgState S;
    RgState SP;                  // spare view: core pointers address the pre-generated next level-1 state (k_regen)
    bool spares = false;
    uint64_t step_count = 0;
    hipStream_t side = nullptr;  // stream of the background generator (high priority: a low-priority queue starves behind the back-to-back step kernels)
    hipEvent_t ev_step = nullptr, ev_regen = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    uint32_t *d_err = nullptr;
    uint8_t *d_keys = nullptr;
    bool render_pending = false;
    std::vector<uint64_t> seed_lo, seed_hi;  // host copy of the seeds the next reset will use
    std::vector<uint8_t> reseed;
    std::string err;
    // per-kernel HIP-event timing (rg_timing_*)
    bool timing = false;
    std::vector<hipEvent_t> ev[4];   // start/stop pairs
    size_t ev_used[4] = {0, 0, 0, 0};
    uint64_t timing_seq[4] = {0, 0, 0, 0};
    uint64_t timing_stride = 1;
};

#define RG_TIMING_MAX 4096
struct TimedLaunch {  // brackets one kernel launch with an event pair when timing is on
    rg_handle *h; int k; bool on;
    TimedLaunch(rg_handle *h_, int k_) : h(h_), k(k_), on(false) {
        // sampled: an event pair costs a few us of stream time, so only every `timing_stride`-th launch of a kernel is bracketed
        if (h->timing && (h->timing_seq[k]++ % h->timing_stride) == 0 && h->ev_used[k] + 2 <= h->ev[k].size()) {
            on = true; (void)hipEventRecord(h->ev[k][h->ev_used[k]], h->stream);
        }
    }
    ~TimedLaunch() { if (on) { (void)hipEventRecord(h->ev[k][h->ev_used[k] + 1], h->stream); h->ev_used[k] += 2; } }
};

static thread_local std::string g_create_err;

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
    }
    return 0;
}

static int upload_seeds(rg_handle *h) {
    size_t n = (size_t)h->S.n;
    HIPCHK(h, hipMemcpyAsync(h->S.seed_lo, h->seed_lo.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->S.seed_hi, h->seed_hi.data(), n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->S.reseed, h->reseed.data(), n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // host vectors may change right after
    return 0;
}

extern "C" {

const char *rg_last_error(const rg_t *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int rg_create(const char *const *cfg_json, int n_env, uint64_t max_steps, int device, int auto_reset, rg_t **out) {
    if (!out || n_env <= 0) { g_create_err = "rg_create: invalid arguments"; return 1; }
    *out = nullptr;
    rg_handle *h = new rg_handle();
    std::random_device rd;
    std::mt19937_64 gen(((uint64_t)rd() << 32) ^ rd());
    h->seed_lo.resize(n_env); h->seed_hi.resize(n_env); h->reseed.resize(n_env);
    const char *prev = nullptr;
    RgParsed p;
    for (int i = 0; i < n_env; i++) {
        const char *js = cfg_json ? cfg_json[i] : nullptr;
        bool same_text = i > 0 && ((js == nullptr && prev == nullptr) || (js && prev && strcmp(js, prev) == 0));
        if (!same_text) {
            std::string e = rg_parse_config(js, &p);
            if (!e.empty()) { g_create_err = "Failed to parse config: " + e; delete h; return 1; }
            if (i == 0) h->parsed = p;
            else if (!rg_config_equal(p.cfg, h->parsed.cfg)) {
                g_create_err = "configs of env 0 and env " + std::to_string(i) + " differ in more than the seed (unsupported by the batched stepper)";
                delete h; return 1;
            }
            prev = js;
        }
        if (p.has_seed) { h->seed_lo[i] = p.seed_lo; h->seed_hi[i] = p.seed_hi; h->reseed[i] = 0; }
        else {
            unsigned __int128 s = ((unsigned __int128)gen() << 64) | gen();
            if (p.has_seed_range && p.seed_range[1] > p.seed_range[0]) s = p.seed_range[0] + s % (p.seed_range[1] - p.seed_range[0]);
            h->seed_lo[i] = (uint64_t)s; h->seed_hi[i] = (uint64_t)(s >> 64); h->reseed[i] = 1;
        }
    }
    h->cfg = h->parsed.cfg;
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
    S.n = n_env; S.hw = (int)hw;
    bool ok = dev_alloc(h, &S.cell, n * hw) && dev_alloc(h, &S.screen, n * hw) && dev_alloc(h, &S.hist, n * hw) &&
              dev_alloc(h, &S.p_pos, n) && dev_alloc(h, &S.p_hp, n) && dev_alloc(h, &S.p_hpmax, n) && dev_alloc(h, &S.p_lvl, n) &&
              dev_alloc(h, &S.p_exp, n) && dev_alloc(h, &S.food, n) && dev_alloc(h, &S.quiet, n) && dev_alloc(h, &S.pack_gold, n) &&
              dev_alloc(h, &S.dlevel, n) && dev_alloc(h, &S.steps, n) && dev_alloc(h, &S.flags, n) && dev_alloc(h, &S.reward, n) && dev_alloc(h, &S.done, n) &&
              dev_alloc(h, &S.rng, 12 * n) && dev_alloc(h, &S.seed_lo, n) && dev_alloc(h, &S.seed_hi, n) && dev_alloc(h, &S.reseed, n) &&
              dev_alloc(h, &S.room_rect, RG_MAX_ROOMS * n) && dev_alloc(h, &S.room_meta, RG_MAX_ROOMS * n) &&
              dev_alloc(h, &S.mon_w0, RG_MAX_ROOMS * n) && dev_alloc(h, &S.mon_hp, RG_MAX_ROOMS * n) && dev_alloc(h, &S.mon_exp, RG_MAX_ROOMS * n) &&
              dev_alloc(h, &S.mon_cnt, n) && dev_alloc(h, &S.gold_pos, RG_MAX_ROOMS * n) && dev_alloc(h, &S.gold_amt, RG_MAX_ROOMS * n) &&
              dev_alloc(h, &S.edge_a, RG_MAX_EDGES * n) && dev_alloc(h, &S.edge_b, RG_MAX_EDGES * n) &&
              dev_alloc(h, &S.maze_stack, RG_MAZE_STACK * n) &&
              dev_alloc(h, &S.dc_map, h->cfg.n_enemies > 0 ? n * RG_DIST_SLOTS * hw : 16) && dev_alloc(h, &S.dc_key, RG_DIST_SLOTS * n) &&
              dev_alloc(h, &S.dc_head, n) && dev_alloc(h, &S.dc_len, n) && dev_alloc(h, &S.status, n * 10) &&
              dev_alloc(h, &h->d_err, 4) && dev_alloc(h, &h->d_keys, n);
    ok = ok && dev_alloc(h, &S.sp_ready, n);
    h->SP = S;
    h->spares = auto_reset != 0 && getenv("ROGUE_GYM_HIP_NO_SPARES") == nullptr;
    if (ok && h->spares) {
        RgState &P = h->SP;
        ok = dev_alloc(h, &P.cell, n * hw) && dev_alloc(h, &P.p_pos, n) && dev_alloc(h, &P.p_hp, n) && dev_alloc(h, &P.p_hpmax, n) && dev_alloc(h, &P.p_lvl, n) &&
             dev_alloc(h, &P.p_exp, n) && dev_alloc(h, &P.food, n) && dev_alloc(h, &P.quiet, n) && dev_alloc(h, &P.pack_gold, n) && dev_alloc(h, &P.dlevel, n) &&
             dev_alloc(h, &P.rng, 12 * n) && dev_alloc(h, &P.room_rect, RG_MAX_ROOMS * n) && dev_alloc(h, &P.room_meta, RG_MAX_ROOMS * n) &&
             dev_alloc(h, &P.mon_w0, RG_MAX_ROOMS * n) && dev_alloc(h, &P.mon_hp, RG_MAX_ROOMS * n) && dev_alloc(h, &P.mon_exp, RG_MAX_ROOMS * n) &&
             dev_alloc(h, &P.mon_cnt, n) && dev_alloc(h, &P.gold_pos, RG_MAX_ROOMS * n) && dev_alloc(h, &P.gold_amt, RG_MAX_ROOMS * n) &&
             dev_alloc(h, &P.edge_a, RG_MAX_EDGES * n) && dev_alloc(h, &P.edge_b, RG_MAX_EDGES * n) && dev_alloc(h, &P.maze_stack, RG_MAZE_STACK * n);
        P.prof = nullptr;
        int lo = 0, hi = 0;
        if (ok && (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, getenv("ROGUE_GYM_HIP_SIDE_LOWPRIO") ? lo : hi) != hipSuccess ||
                   hipEventCreateWithFlags(&h->ev_step, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_regen, hipEventDisableTiming) != hipSuccess)) {
            h->err = "failed to create the background generation stream"; ok = false;
        }
    }
    if (!ok) { g_create_err = h->err; free_all(h); delete h; return 1; }
    // screen rows 0 and H-1 are never drawn: PlayerState::new fills the map with b' ' (python/src/lib.rs:41-50)
    if (hipMemset(S.screen, ' ', n * hw) != hipSuccess) { g_create_err = "hipMemset failed"; free_all(h); delete h; return 1; }
    if (upload_seeds(h)) { g_create_err = h->err; free_all(h); delete h; return 1; }
    rgk_build(&h->S, &h->cfg, h->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { g_create_err = std::string("k_build: ") + hipGetErrorString(e); free_all(h); delete h; return 1; }
    h->render_pending = true;
    if (h->spares) {  // first spares: generated in the background right away
        rgk_regen(&h->SP, &h->cfg, h->side);
        (void)hipEventRecord(h->ev_regen, h->side);
    }
    *out = h;
    return 0;
}

void rg_destroy(rg_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->ev_step) (void)hipEventDestroy(h->ev_step);
    if (h->ev_regen) (void)hipEventDestroy(h->ev_regen);
    for (int k = 0; k < 4; k++) for (auto &e : h->ev[k]) (void)hipEventDestroy(e);
    free_all(h);
    delete h;
}

int rg_dims(const rg_t *h, int *height, int *width, int *symbols, int *n_env) {
    if (height) *height = h->cfg.height;
    if (width) *width = h->cfg.width;
    if (symbols) *symbols = h->cfg.symbols;
    if (n_env) *n_env = h->S.n;
    return 0;
}

int rg_set_stream(rg_t *h, void *hip_stream) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(h, hipStreamSynchronize(h->side));
    h->stream = (hipStream_t)hip_stream;
    return 0;
}

int rg_seed(rg_t *h, const uint64_t *seed_lo, const uint64_t *seed_hi, int n) {
    if (n > h->S.n) n = h->S.n;
    HIPCHK(h, hipSetDevice(h->device));
    // envs without a configured seed advance their seed on the device at every build: keep those values
    if (h->side) HIPCHK(h, hipStreamSynchronize(h->side));
    std::vector<uint64_t> lo(h->S.n), hi(h->S.n);
    HIPCHK(h, hipMemcpyAsync(lo.data(), h->S.seed_lo, (size_t)h->S.n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(hi.data(), h->S.seed_hi, (size_t)h->S.n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < h->S.n; i++)
        if (h->reseed[i]) { h->seed_lo[i] = lo[i]; h->seed_hi[i] = hi[i]; }
    for (int i = 0; i < n; i++) { h->seed_lo[i] = seed_lo[i]; h->seed_hi[i] = seed_hi ? seed_hi[i] : 0; h->reseed[i] = 0; }
    if (h->spares) {  // spares were generated from the old seeds: drop them (k_step falls back to inline generation until refilled)
        HIPCHK(h, hipStreamSynchronize(h->side));
        HIPCHK(h, hipMemsetAsync(h->S.sp_ready, 0, (size_t)h->S.n * 4, h->stream));
    }
    return upload_seeds(h);
}

int rg_reset(rg_t *h) {
    HIPCHK(h, hipSetDevice(h->device));
    { TimedLaunch t(h, 3); rgk_build(&h->S, &h->cfg, h->stream); }
    HIPCHK(h, hipGetLastError());
    h->render_pending = true;
    return 0;
}

int rg_step(rg_t *h, const uint8_t *keys, int keys_on_device) {
    HIPCHK(h, hipSetDevice(h->device));
    if (flush_render(h)) return 1;
    const uint8_t *dk = keys;
    if (!keys_on_device) {
        HIPCHK(h, hipMemcpyAsync(h->d_keys, keys, (size_t)h->S.n, hipMemcpyHostToDevice, h->stream));
        dk = h->d_keys;
    }
    { TimedLaunch t(h, 0); rgk_step(&h->S, &h->SP, &h->cfg, dk, h->d_err, h->spares ? 1 : 0, h->stream); }
    HIPCHK(h, hipGetLastError());
    static int regen_every = getenv("ROGUE_GYM_HIP_REGEN_EVERY") ? atoi(getenv("ROGUE_GYM_HIP_REGEN_EVERY")) : 1;
    if (h->spares && (++h->step_count % (uint64_t)(regen_every < 1 ? 1 : regen_every)) == 0) {
        // refill the consumed spares behind this step on the side stream.  Purely stream-ordered (the host runs far ahead of
        // the GPU, so polling an event here would be meaningless); a launch that finds nothing to do costs ~10 us, concurrently.
        HIPCHK(h, hipEventRecord(h->ev_step, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_step, 0));
        rgk_regen(&h->SP, &h->cfg, h->side);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev_regen, h->side));
    }
    h->render_pending = true;
    return 0;
}

int rg_sync(rg_t *h) {
    HIPCHK(h, hipSetDevice(h->device));
    uint32_t err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->d_err, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->side) HIPCHK(h, hipStreamSynchronize(h->side));
    if (err) {
        HIPCHK(h, hipMemsetAsync(h->d_err, 0, 4, h->stream));
        if (err & RG_FLAG_ERR_KEY) h->err = "Invalid input (key is not in the ai keymap)";
        else if (err & RG_FLAG_ERR_DEAD) h->err = "Ignored input (action while the player is dead)";
        else h->err = "Invalid tile in symbol image (symbol >= symbols - 1)";
        return 1;
    }
    return 0;
}

int rg_screen(rg_t *h, uint8_t **dev) { HIPCHK(h, hipSetDevice(h->device)); if (flush_render(h)) return 1; *dev = h->S.screen; return 0; }
int rg_hist(rg_t *h, uint8_t **dev) { HIPCHK(h, hipSetDevice(h->device)); if (flush_render(h)) return 1; *dev = h->S.hist; return 0; }
int rg_status(rg_t *h, int32_t **dev) { *dev = h->S.status; return 0; }
int rg_flags(rg_t *h, uint32_t **dev) { HIPCHK(h, hipSetDevice(h->device)); if (flush_render(h)) return 1; *dev = h->S.flags; return 0; }
int rg_reward(rg_t *h, float **dev) { *dev = h->S.reward; return 0; }
int rg_done(rg_t *h, uint8_t **dev) { *dev = h->S.done; return 0; }

int rg_obs_channels(const rg_t *h, int symbol, uint32_t status_flag, int with_hist) {
    return (symbol ? h->cfg.symbols : 1) + __builtin_popcount(status_flag & 0x1ffu) + (with_hist ? 1 : 0);
}

static int obs_common(rg_t *h, uint32_t status_flag, int with_hist, int kind, float *out_dev) {
    HIPCHK(h, hipSetDevice(h->device));
    {   // steady state: one fused pass refreshes the mirrors of Redraw envs and encodes every env
        TimedLaunch t(h, 2);
        if (rgk_obs(&h->S, &h->cfg, status_flag & 0x1ffu, with_hist ? 1 : 0, kind, out_dev, h->d_err, h->stream)) {
            HIPCHK(h, hipGetLastError());
            h->render_pending = false;
            return 0;
        }
        t.on = false;
    }
    if (flush_render(h)) return 1;
    {
        TimedLaunch t(h, 2);
        rgk_encode(h->S.screen, h->S.hist, h->S.status, h->S.flags, h->d_err, h->S.n, h->S.hw, h->cfg.symbols, status_flag & 0x1ffu, with_hist ? 1 : 0, kind,
                   out_dev, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}
int rg_obs_gray(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev) { return obs_common(h, status_flag, with_hist, 0, out_dev); }
int rg_obs_symbol(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev) { return obs_common(h, status_flag, with_hist, 1, out_dev); }

int rg_fetch_states(rg_t *h, uint8_t *screen, uint8_t *hist, int32_t *status, uint32_t *flags) {
    HIPCHK(h, hipSetDevice(h->device));
    if (flush_render(h)) return 1;
    size_t n = (size_t)h->S.n, hw = (size_t)h->S.hw;
    if (screen) HIPCHK(h, hipMemcpyAsync(screen, h->S.screen, n * hw, hipMemcpyDeviceToHost, h->stream));
    if (hist) HIPCHK(h, hipMemcpyAsync(hist, h->S.hist, n * hw, hipMemcpyDeviceToHost, h->stream));
    if (status) HIPCHK(h, hipMemcpyAsync(status, h->S.status, n * 40, hipMemcpyDeviceToHost, h->stream));
    if (flags) HIPCHK(h, hipMemcpyAsync(flags, h->S.flags, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int rg_encode_host(int device, const uint8_t *screen, const uint8_t *hist, const int32_t *status, int height, int width, int symbols,
                   uint32_t status_flag, int with_hist, int kind, float *out_host) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device available (librogue_gym_hip has no CPU fallback)"; return 1; }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "invalid HIP device"; return 1; }
    size_t hw = (size_t)height * width;
    status_flag &= 0x1ffu;
    int c = (kind ? symbols : 1) + __builtin_popcount(status_flag) + (with_hist ? 1 : 0);
    uint8_t *d_scr = nullptr, *d_hist = nullptr; int32_t *d_st = nullptr; uint32_t *d_err = nullptr; float *d_out = nullptr;
    int rc = 1;
    uint32_t err = 0;
    if (hipMalloc((void **)&d_scr, hw) != hipSuccess || hipMalloc((void **)&d_hist, hw) != hipSuccess || hipMalloc((void **)&d_st, 40) != hipSuccess ||
        hipMalloc((void **)&d_err, 4) != hipSuccess || hipMalloc((void **)&d_out, (size_t)c * hw * 4) != hipSuccess) { g_create_err = "hipMalloc failed"; goto done; }
    if (hipMemcpy(d_scr, screen, hw, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_st, status, 40, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_err, 0, 4) != hipSuccess || (with_hist && hipMemcpy(d_hist, hist, hw, hipMemcpyHostToDevice) != hipSuccess)) { g_create_err = "hipMemcpy failed"; goto done; }
    rgk_encode(d_scr, d_hist, d_st, nullptr, d_err, 1, (int)hw, symbols, status_flag, with_hist ? 1 : 0, kind, d_out, nullptr);
    if (hipMemcpy(out_host, d_out, (size_t)c * hw * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        g_create_err = "encode kernel failed"; goto done;
    }
    if (err) { g_create_err = "Invalid tile in symbol image (symbol >= symbols - 1)"; goto done; }
    rc = 0;
done:
    (void)hipFree(d_scr); (void)hipFree(d_hist); (void)hipFree(d_st); (void)hipFree(d_err); (void)hipFree(d_out);
    return rc;
}

// development aid: in-kernel phase trace; out = [ceil(n_env / 64)][64] words, one row per wave of the LAST k_step / k_build launch
// (word 0 = record count, words 1.. = phase << 48 | ticks, word 63 = whole-wave ticks)
int rg_prof(rg_t *h, int enable, unsigned long long *out) {
    HIPCHK(h, hipSetDevice(h->device));
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
    if (on && h->ev[0].empty())
        for (int k = 0; k < 4; k++) {
            h->ev[k].resize(2 * RG_TIMING_MAX);
            for (auto &e : h->ev[k]) HIPCHK(h, hipEventCreate(&e));
        }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 4; k++) h->ev_used[k] = 0;
    h->timing = on != 0;
    h->timing_stride = on > 1 ? (uint64_t)on : 1;  // on = N > 1: bracket every N-th launch only
    return 0;
}

int rg_timing_read(rg_t *h, double ms[4], uint64_t launches[4]) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 4; k++) {
        double sum = 0;
        for (size_t i = 0; i + 1 < h->ev_used[k]; i += 2) {
            float t = 0;
            HIPCHK(h, hipEventElapsedTime(&t, h->ev[k][i], h->ev[k][i + 1]));
            sum += t;
        }
        ms[k] = sum; launches[k] = h->ev_used[k] / 2;
        h->ev_used[k] = 0;
    }
    return 0;
}

int rg_dump_config(const rg_t *h, int env, char *buf, size_t cap) {
    if (env < 0 || env >= h->S.n) return 1;
    std::string s = rg_dump_config_json(h->parsed, h->seed_lo[env], h->seed_hi[env], !h->reseed[env]);
    if (s.size() + 1 > cap) return 1;
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

int rg_debug_fetch(rg_t *h, int env, rg_debug_state *out, uint16_t *cells) {
    HIPCHK(h, hipSetDevice(h->device));
    if (env < 0 || env >= h->S.n) { h->err = "env out of range"; return 1; }
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
    for (int s = 0; s < RG_MAX_ROOMS; s++) {
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
    for (int s = 0; s < RG_MAX_ROOMS; s++) {
        uint32_t g; HIPCHK(h, hipMemcpy(&g, S.gold_pos + s * n + e, 4, hipMemcpyDeviceToHost));
        if (!(g & 0x10000u)) continue;
        out->gold_x[ng] = (g >> 8) & 0xff; out->gold_y[ng] = g & 0xff;
        uint32_t a; HIPCHK(h, hipMemcpy(&a, S.gold_amt + s * n + e, 4, hipMemcpyDeviceToHost));
        out->gold_amount[ng] = (int32_t)a; ng++;
    }
    out->n_gold = ng;
    if (cells) HIPCHK(h, hipMemcpy(cells, S.cell + e * S.hw, (size_t)S.hw * 2, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
