/*
 * rogue_gym_hip.h -- C ABI of the MI355X-native batched Rogue-Gym stepper (librogue_gym_hip.so).
 *
 * This is the drop-in boundary: the entry points are what the reference's FFI crate
 * (`rogue_gym_python._rogue_gym`, /root/reference/python/src/lib.rs) would bind in place of
 * its per-env `GameStateImpl` + one-OS-thread-per-env `ThreadConductor`.  Plain pointers and
 * sizes only -- no torch / PyO3 types.  All buffers named "dev" live in HBM on the handle's
 * device; observations are written into caller-owned device buffers (e.g. torch tensors) and
 * never leave HBM.
 *
 * Conventions: every function returns 0 on success, non-zero on failure; the message is
 * available from rg_last_error() (the Python shim raises
 * RuntimeError("Error in rogue-gym: " + msg), mirroring python/src/lib.rs:20-26).
 * A handle is not thread-safe; distinct handles are independent.  Calls are asynchronous on the
 * handle's HIP stream unless stated otherwise.
 *
 * file:line citations are relative to /root/reference.
 */
#ifndef ROGUE_GYM_HIP_H
#define ROGUE_GYM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_handle rg_t;

/* bits of the per-env flag word returned by rg_flags()/rg_fetch_states() */
#define RG_FLAG_TERMINAL   0x00000001u  /* PlayerState.is_terminal (state_impls.rs:77) */
#define RG_FLAG_DEAD       0x00000002u  /* engine is in the Grave modal (core/src/lib.rs:301-315) */
#define RG_FLAG_REDRAW     0x00000004u  /* internal: screen mirror refresh pending */
#define RG_FLAG_HIST_STALE 0x00000008u  /* internal: this Redraw keeps the old level's history */
#define RG_FLAG_HIST_LAG   0x00000010u  /* internal: the history mirror still shows the level before the current one */
#define RG_FLAG_HIST_DIRTY 0x00000020u  /* internal: the visited set changed since the history mirror was last written */
#define RG_FLAG_SCR_CHANGED 0x00000040u /* internal: the turn itself changed bytes of the screen mirror since the bound observation tensor was last written (rg_obs_bind) */
#define RG_FLAG_MSG_SHIFT  8            /* bits 8..14: MessageFlagInner (python/src/flags.rs:6-39) */
#define RG_FLAG_MSG_MASK   0x00007f00u
#define RG_FLAG_ERR_KEY    0x00010000u  /* ErrorKind::InvalidInput: key not in KeyMap::ai (input.rs:73-100) */
#define RG_FLAG_ERR_DEAD   0x00020000u  /* ErrorKind::IgnoredInput: action key while dead */
#define RG_FLAG_ERR_TILE   0x00040000u  /* symbol image: glyph with symbol >= symbols-1 (python/src/lib.rs:96-102) */
#define RG_FLAG_ERR_INTERNAL 0x00080000u /* a capacity guard of the stepper tripped (each is proven unreachable; see rg_kernels.hip) */
#define RG_FLAG_ERR_MASK   0x00ff0000u

/* Replaces GameState::__new__ / ParallelGameState::new (python/src/lib.rs:217-225,270-294) and
 * ThreadConductor::new (thread_impls.rs:14-34): parses one GameConfig JSON per env
 * (core/src/lib.rs:42-86).  Envs may differ in anything: envs with equal configs (seeds aside) form a group that is
 * stepped as one homogeneous batch, and the handle presents all groups in the caller's env order.  When the groups differ in width / height
 * there is no [n_env][H][W] tensor: rg_screen / rg_hist / rg_obs_* / rg_pack_compact / rg_expand_compact / rg_obs_host then fail with a message
 * that says so, and rg_fetch_states delivers the screens in a ragged layout (rg_env_dims); everything else works unchanged.  Allocates the SoA
 * state for n_env environments on HIP device `device`, generates every level-1 dungeon and
 * draws the first screens.  auto_reset != 0 selects ThreadConductor::step semantics (terminal
 * envs are rebuilt inside rg_step and report the post-reset state with is_terminal forced
 * true, thread_impls.rs:69-79); 0 selects single GameState semantics.
 * cfg_json[i] may be NULL (= GameConfig::default()). */
int rg_create(const char *const *cfg_json, int n_env, uint64_t max_steps, int device, int auto_reset, rg_t **out);
void rg_destroy(rg_t *h);
/* error text of the last failed call on `h` (h == NULL: last failed rg_create on this thread) */
const char *rg_last_error(const rg_t *h);

/* Build id: the first 16 hex digits of the sha256 over the library's sources (the csrc files and this header), so that a test can tell whether the library
 * that is loaded was built from the sources that are checked out. */
const char *rg_build_id(void);

/* GameState::screen_size / symbols (python/src/lib.rs:226-228,255-257,295-300) */
int rg_dims(const rg_t *h, int *height, int *width, int *symbols, int *n_env);
/* Height and width of every env's own config (host arrays of n_env i32; either may be NULL).  Equal for all envs unless the batch mixes sizes. */
int rg_env_dims(const rg_t *h, int32_t *heights, int32_t *widths);
/* `symbols` of every env's own config (GameStateImpl::new computes it per env, state_impls.rs:21-25; PlayerState.symbols): out_host = i32 [n_env].
 * rg_dims reports env 0's, like ParallelGameState::symbols (python/src/lib.rs:281-285,298-300). */
int rg_env_symbols(const rg_t *h, int32_t *out_host);
/* Run all later work of this handle on `hip_stream` (a hipStream_t; NULL = default stream). */
int rg_set_stream(rg_t *h, void *hip_stream);

/* GameState::set_seed / ParallelGameState::seed (python/src/lib.rs:229-232,301-307;
 * thread_impls.rs:45-50,125-128): u128 seeds as (lo, hi) words, used at the next reset. n may be
 * smaller than n_env (zip semantics). Host pointers. */
int rg_seed(rg_t *h, const uint64_t *seed_lo, const uint64_t *seed_hi, int n);
/* GameState::reset / ParallelGameState::reset: rebuild every env from its config (+ seed). */
int rg_reset(rg_t *h);
/* GameState::react / ParallelGameState::step (python/src/lib.rs:241-243,315-321 ->
 * state_impls.rs:51-79, thread_impls.rs:61-81): one key byte per env.  `keys` is a device pointer
 * when keys_on_device != 0, else a host pointer (copied H2D on the stream). */
int rg_step(rg_t *h, const uint8_t *keys, int keys_on_device);
/* ThreadConductor::step zips the key vector with the envs (thread_impls.rs:62-64): with n_keys < n_env only the first n_keys envs receive a
 * key; surplus keys are dropped.  (The reference then blocks forever on the reply of the envs that got no instruction; here they simply keep
 * their state, reward 0.)  rg_step == rg_step_prefix with n_keys = n_env. */
int rg_step_prefix(rg_t *h, const uint8_t *keys, int n_keys, int keys_on_device);
/* rg_step followed by rg_obs_gray(status_flag, with_hist, out_dev) as ONE call -- what a learner's loop does every step (GameState::react then
 * PlayerState::gray_image, python/src/lib.rs:72-87,251-256): one trip through the binding instead of two.  (A step kernel whose waves also drew and
 * encoded their own envs was built and measured in round 5: 196 us against 52 + 44 us for the two launches -- profiles/r05_experiments.txt.) */
int rg_step_obs_gray(rg_t *h, const uint8_t *keys, int keys_on_device, uint32_t status_flag, int with_hist, float *out_dev);
/* A BOUND observation tensor (opt-in): `out_dev` becomes the handle's standing observation batch for exactly this image setting (kind 0 gray / 1 symbol,
 * no status planes, no history plane) -- what the reference's learner rebuilds from the PlayerState values after every step (parallel.py:44-66 ->
 * ImageSetting.expand -> PlayerState::gray_image / symbol_image, python/src/lib.rs:72-205).  While bound, every rg_step_obs_gray / rg_obs_gray / rg_obs_symbol
 * call with the same arguments keeps the tensor CURRENT IN PLACE: it rewrites the images of the envs whose screen changed since the last such call (a Redraw
 * drawn from the tiles, or bytes the turn wrote into the screen mirror itself) and leaves the others -- whose f32 image is already what a full encode would
 * write -- untouched; the contents after the call are bit-identical to the unbound call's.  The caller must not write to the tensor.  The first call after
 * binding, and the first after anything else drew the mirrors (rg_fetch_states, an observation call with other arguments, rg_screen ...), encodes every env.
 * out_dev = NULL unbinds.  Not for handles with config groups.  (65 536 mini envs: 57 % of the envs of a step of the random policy change nothing on screen.) */
int rg_obs_bind(rg_t *h, int kind, uint32_t status_flag, int with_hist, float *out_dev);
/* Wait for the stream; returns non-zero (and sets the error text) if any env raised an error flag
 * since the last call (invalid key / action while dead), like the PyRuntimeError of lib.rs:20-26. */
int rg_sync(rg_t *h);

/* Device-resident mirrors (PlayerState, python/src/lib.rs:29-38), valid until rg_destroy:
 *   screen  u8  [n_env][H][W]   glyph bytes, refreshed only on Reaction::Redraw
 *   hist    u8  [n_env][H][W]   0/1 visited-history plane (copy_hist, lib.rs:105-111)
 *   status  i32 [n_env][10]     Status::to_vec order (player.rs:418-430), refreshed on StatusUpdated
 *   flags   u32 [n_env]         RG_FLAG_* bits
 *   reward  f32 [n_env]         max(0, gold_after - gold_before) of the last rg_step (parallel.py:60-63)
 *   done    u8  [n_env]         is_terminal of the state returned by the last rg_step (parallel.py:64)
 * Reading them (rg_screen/rg_hist/rg_obs_*) flushes the pending screen render first. */
int rg_screen(rg_t *h, uint8_t **dev);
int rg_hist(rg_t *h, uint8_t **dev);
int rg_status(rg_t *h, int32_t **dev);
int rg_flags(rg_t *h, uint32_t **dev);
int rg_reward(rg_t *h, float **dev);
int rg_done(rg_t *h, uint8_t **dev);

/* StairRewardParallel (python/rogue_gym/envs/wrappers.py:45-64) fused into the step kernel: from the next rg_step on, `bonus` is added to
 * reward[e] whenever the dungeon level env e reports after the key is above the level it reported one step earlier (the comparison level
 * follows the reported one, so it is back at 1 after an auto-reset: a descent that ends the episode pays nothing, exactly as the wrapper).
 * The bonus is part of the reward mirror and of the compact record (rg_pack_compact / rg_allgather_compact).  0 (the default) = off. */
int rg_set_stair_reward(rg_t *h, float bonus);

/* PlayerState::gray_image[_with_hist] / symbol_image[_with_hist] for the whole batch
 * (python/src/lib.rs:72-111,162-205; flags.rs:88-115; symbol.rs:17-71), written straight into
 * out_dev = f32 [n_env][C][H][W] with C = 1 (gray) or `symbols` (one-hot) + popcount(status_flag)
 * (+1 with hist).  Bit layout of status_flag: StatusFlagInner (flags.rs:45-55). */
int rg_obs_gray(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev);
int rg_obs_symbol(rg_t *h, uint32_t status_flag, int with_hist, float *out_dev);
int rg_obs_channels(const rg_t *h, int symbol, uint32_t status_flag, int with_hist);

/* PlayerState::status_vec (python/src/lib.rs:158-161, flags.rs:67-87) for the whole batch: out_host = i32 [n_env][popcount(flag)].  Synchronous. */
int rg_status_vec(rg_t *h, uint32_t status_flag, int32_t *out_host);

/* Host copies for the value-object API (ParallelGameState::states/step return Vec<PlayerState>):
 * synchronous D2H of the mirrors; any pointer may be NULL.  screen / hist: u8 [n_env][H][W]; for a batch that mixes sizes the envs follow
 * each other in env order, env i taking H_i * W_i bytes (rg_env_dims). */
int rg_fetch_states(rg_t *h, uint8_t *screen, uint8_t *hist, int32_t *status, uint32_t *flags);
/* ParallelGameState::step (python/src/lib.rs:315-321) in one call and ONE stream wait, for small batches: keys from host memory (zipped with the envs like
 * rg_step_prefix), then the mirror refresh, then one kernel writes status i32 [n][10], flags u32 [n] -- and, if given, screen / hist u8 [n][H*W] -- straight to
 * their destinations, which must be device-visible (pinned host memory from rg_host_alloc, or device memory).  Errors as rg_sync.  Not for config groups. */
int rg_step_fetch(rg_t *h, const uint8_t *keys_host, int n_keys, uint8_t *screen, uint8_t *hist, int32_t *status, uint32_t *flags);

/* Device-side snapshots of the two big mirrors, for value-object callers that read mostly status / flags (parallel.py:59-64 uses gold and
 * is_terminal of every state, nothing else): rg_snapshot_take flushes the pending render and copies screen + hist device-to-device into
 * `dev` = u8 [2][n_env][H][W] (screen first; allocate with rg_dev_alloc), asynchronously on the handle's stream; rg_dev_read is a synchronous
 * D2H copy of any part of it (or of any device buffer of the handle) made when a PlayerState's screen is first looked at.  Not for batches that
 * mix sizes. */
int rg_dev_alloc(int device, size_t bytes, void **out);
void rg_dev_free(int device, void *p);
int rg_snapshot_take(rg_t *h, void *dev);
int rg_dev_read(rg_t *h, const void *dev_src, void *host_dst, size_t bytes);
/* the same for `rows` pieces of row_bytes that lie src_pitch apart on the device (screen and history of ONE env of a snapshot: one copy) */
int rg_dev_read_rows(rg_t *h, const void *dev_src, size_t src_pitch, void *host_dst, size_t row_bytes, int rows);

/* Stateless encode of ONE host-side PlayerState snapshot on the GPU (PlayerState.gray_image &c.
 * called on a cloned value object): uploads, runs the same encode kernel, downloads.
 * kind 0 = gray, 1 = symbol.  Returns non-zero on the symbol-image tile error. */
int rg_encode_host(int device, const uint8_t *screen, const uint8_t *hist, const int32_t *status, int height, int width,
                   int symbols, uint32_t status_flag, int with_hist, int kind, float *out_host);

/* The same for n snapshots at once (screen / hist u8 [n][H][W], status i32 [n][10], out f32 [n][C][H][W]); the device scratch is cached per
 * device, so repeated calls allocate nothing. */
int rg_encode_host_batch(int device, int n, const uint8_t *screen, const uint8_t *hist, const int32_t *status, int height, int width,
                         int symbols, uint32_t status_flag, int with_hist, int kind, float *out_host);
/* PlayerState images of the handle's CURRENT states for the whole batch, into host memory: fused mirror refresh + encode on the device
 * (scratch kept by the handle), one D2H copy.  Synchronous.  out_host = f32 [n_env][C][H][W], ideally pinned (rg_host_alloc). */
int rg_obs_host(rg_t *h, int kind, uint32_t status_flag, int with_hist, float *out_host);
/* Pinned (page-locked) host memory for rg_fetch_states / rg_obs_host destinations (D2H at full PCIe rate). */
int rg_host_alloc(size_t bytes, void **out);
void rg_host_free(void *p);

/* Multi-GPU: the ONE per-step collective gathers a compact record per env instead of the f32 observation.
 * rg_pack_compact writes u8 [n_env][rg_compact_record_bytes] = {screen u8[H*W], status i32[10], reward f32, flags u32, hist u8[H*W] if
 * with_hist} into out_dev (flushes the pending render first) -- everything ThreadConductor::step returns per env in one reply, state AND
 * terminal flag (python/src/thread_impls.rs:61-81; parallel.py:59-64 derives reward and done from it): `reward` is the step's gold delta,
 * `flags` the public RG_FLAG_* bits (TERMINAL = done, DEAD, message bits 8..14, error bits; the mirror bookkeeping bits are masked out).
 * After the all-gather, the consumer expands any number of records with rg_expand_compact into out_dev = f32 [n][C][H][W] --
 * PlayerState::{gray,symbol}_image[_with_hist] (python/src/lib.rs:72-111; symbol.rs:51-71) from the packed bytes -- and reads reward / flags
 * at RG_COMPACT_REWARD_OFFSET(H*W) / RG_COMPACT_FLAGS_OFFSET(H*W) of every record.  H*W must be divisible by 4. */
#define RG_COMPACT_FIXED_BYTES 48                       /* status i32[10] + reward f32 + flags u32 */
#define RG_COMPACT_STATUS_OFFSET(hw) (hw)
#define RG_COMPACT_REWARD_OFFSET(hw) ((hw) + 40)
#define RG_COMPACT_FLAGS_OFFSET(hw) ((hw) + 44)
#define RG_COMPACT_HIST_OFFSET(hw) ((hw) + RG_COMPACT_FIXED_BYTES)
int rg_compact_record_bytes(const rg_t *h, int with_hist);
int rg_pack_compact(rg_t *h, int with_hist, uint8_t *out_dev);
/* The collective itself ("exactly one per step: ncclAllGather (RCCL over xGMI) of the obs slice, in place into rank-ordered
 * slices"), for hosts that do not bring their own (the reference has none: its workers are threads, python/src/thread_impls.rs:14-34).
 * rg_comm_unique_id: ncclGetUniqueId on one rank, handed to the others by the caller (any channel; 128 bytes).  rg_comm_init: ncclCommInitRank
 * for the handle's device; every rank's handle must hold the same n_env.  rg_allgather_compact: packs this rank's records into slice `rank` of
 * out_dev = u8 [world * n_env][record] and all-gathers in place on the handle's stream (asynchronous like every launch; expand with
 * rg_expand_compact).  librccl is bound at run time: only these four calls need it. */
int rg_comm_unique_id(uint8_t id[128]);
int rg_comm_init(rg_t *h, const uint8_t id[128], int rank, int world);
int rg_comm_destroy(rg_t *h);
/* What the communicator reports about itself (ncclCommCount / ncclCommUserRank) -- not what rg_comm_init was told. */
int rg_comm_count(rg_t *h, int *count, int *rank);
int rg_allgather_compact(rg_t *h, int with_hist, uint8_t *out_dev);
int rg_expand_compact(rg_t *h, const uint8_t *packed_dev, int n, int packed_has_hist, int kind, uint32_t status_flag, int with_hist, float *out_dev);

/* Action-history log (RunTime::saved_inputs, core/src/lib.rs:288; GameState::dump_history, python/src/lib.rs:245-250).  Off by default;
 * rg_history_enable allocates a device-side key log of cap_per_env keys for the running and for the previous episode of every env (an
 * episode = one RunTime: it ends at rg_reset or at the auto-reset).  Every key that is in KeyMap::ai is logged when it is received (also
 * the ones rejected with IgnoredInput while dead), as the reference does.
 * rg_history_keys: raw key bytes; which = 0 running episode, 1 previous episode; *len = keys received (returns 2 if that exceeds the
 * capacity, i.e. the log is truncated).  rg_dump_history: the same as serde_json::to_string_pretty(Vec<InputCode>), the format of
 * data/learned/ddqn-minidungeon/best-actions.json; *needed = bytes incl. NUL (buf may be NULL to query). */
int rg_history_enable(rg_t *h, int cap_per_env);
int rg_history_keys(rg_t *h, int env, int which, uint8_t *keys, size_t cap, uint32_t *len);
int rg_dump_history(rg_t *h, int env, int which, char *buf, size_t cap, size_t *needed);

/* Workload counters accumulated by rg_step since the last reset of the counters: out[0] auto-resets, [1] descents, [2] dist maps built (BFS),
 * [3] levels generated inline by the step kernel (descents, and resets that found no spare), [4] spare levels taken, [5] Redraw reactions, [6] keys processed, [7] partial dist maps continued over the walkable mask saved with them (a map
 * begun before a descent or an opening search, grids of 33..96 columns; counted in [2] as well).  Synchronous. */
int rg_counters(rg_t *h, uint64_t out[8], int reset);
/* ... and the counters beyond the first eight (n_out <= 16): [8] descents whose level came from its next-level structure -- the first two thirds of
 * Dungeon::new_level (rogue/mod.rs:434-481: rooms, passages, gold, stairs) generated ahead of the descent by the background generator, the monsters and
 * the player's placement inside the turn; counted in [3] as well.  [9..15] reserved (0). */
int rg_counters_ex(rg_t *h, uint64_t *out, int n_out, int reset);
/* Effective shader clock right now: a one-wave spin kernel on the handle's stream compares s_memtime (shader-clock ticks) with
 * s_memrealtime (constant 100 MHz).  Synchronous; bench.py's evidence for the clock state of a run. */
int rg_probe_sclk(rg_t *h, double *mhz);

/* Per-kernel timing with HIP events recorded on the handle's stream (bench.py's roofline leg).
 * While enabled, every rg_step / render flush / rg_obs_* launch is bracketed by an event pair (up to
 * 4096 launches per kernel between reads).  rg_timing_read synchronises, returns the summed elapsed
 * milliseconds and launch counts per kernel {0: k_step, 1: k_render, 2: k_gray|k_symbol, 3: k_build}
 * and clears the accumulators.  on = N > 1 brackets only every N-th launch of each kernel (an event pair costs a few
 * microseconds of stream time). */
int rg_timing_enable(rg_t *h, int on);
int rg_timing_read(rg_t *h, double ms[4], uint64_t launches[4]);
/* The same for the first n <= 5 kernels {.., 4: k_regen -- the background generator, stamped on its own low-priority stream}: summed milliseconds and
 * number of SAMPLED launches per kernel, and (launches != NULL) how many launches of the kernel there were in all since the last read / enable. */
int rg_timing_read_all(rg_t *h, int n, double *ms, uint64_t *sampled, uint64_t *launches);
/* The individual samples behind those sums, for latency percentiles (bench.py `step_us`): the duration in ms of every sampled launch of `kernel`
 * (rg_timing_read_all's order) since rg_timing_enable, oldest first, at most `cap` of them, *n = how many.  kernel = -1: one value per STEP -- from the
 * begin of its k_step to the end of its observation pass (kernel 2), for the steps in which both were sampled (rg_timing_enable(h, 1): all).  Call it
 * before rg_timing_read / rg_timing_read_all, which reset the samples.  Not for handles with config groups. */
int rg_timing_read_samples(rg_t *h, int kernel, float *ms, int cap, int *n);

/* GameState::dump_config (python/src/lib.rs:252-254): canonical JSON of env i's effective config. */
int rg_dump_config(const rg_t *h, int env, char *buf, size_t cap);

/* Stateless: parse one GameConfig JSON (GameConfig::from_json, core/src/lib.rs:144-149) and write its canonical
 * re-serialisation (GameConfig::to_json with the reference's skip-if-default rules) into buf.  Needs no device.
 * Returns non-zero and sets rg_last_error(NULL) on a parse / validation error. */
int rg_config_canonical(const char *cfg_json, char *buf, size_t cap);

/* Stateless: what the stepper resolved from the item tables and the player's pack of one GameConfig (Player::init_items, player.rs:136-153;
 * InitItem::initialize, item/mod.rs:181-221) -- JSON {"weapon": {"times", "max", "hit_plus", "dam_plus"} (the wielded dice, fight.rs:21-33),
 * "armor_def" (Player::arm), "init_gold", "can_pickup", "init_draws": [[lo, hi] ...] (the item-stream draws of the build, weapon.rs:159),
 * "symbols", "n_enemies"}.  Needs no device; errors as rg_config_canonical. */
int rg_config_resolved(const char *cfg_json, char *buf, size_t cap);

/* The config surface, key by key: a JSON array of {"path", "status": "honoured" | "inert", "why"} for every key the reference's serde structs
 * know (core/src/lib.rs:42-86, rogue/mod.rs:23-66, item/{mod,gold,weapon,armor}.rs, player.rs:17-32, enemies.rs:18-121).  "inert" = serde reads
 * it and the engine never does; such keys are accepted, type-checked and written back by rg_dump_config.  Needs no device.  *needed = bytes incl.
 * NUL (buf may be NULL to query). */
int rg_config_schema(char *buf, size_t cap, size_t *needed);

/* Parity/debug: synchronous copy of env i's internal state to the host. */
typedef struct rg_debug_state {
    int32_t px, py, dungeon_level, hp, hp_max, player_level, n_monsters, n_gold;
    uint32_t exp, food_left, quiet, pack_gold, steps;
    uint32_t rng[12];        /* dungeon, item, enemy streams x {x,y,z,w} */
    int32_t mon_x[384], mon_y[384], mon_type[384], mon_active[384], mon_hp[384];   /* one slot per room; 160 x 48 holds at most 40 x 9 = 360 rooms */
    uint32_t mon_exp[384];
    int32_t gold_x[384], gold_y[384], gold_amount[384];
    int32_t n_rooms;          /* room_num_x * room_num_y, row-major room ids */
    uint32_t room_rect[384];  /* x0 | y0<<8 | x1<<16 | y1<<24, half-open (Empty room: x0, y0 = its anchor cell) */
    int32_t room_meta[384];   /* bits 0-1 kind (0 Normal, 1 Maze, 2 Empty; rooms.rs:11-19), 4 dark, 8 visited, 16 has gold */
} rg_debug_state;
/* cells: u16 [H][W] = surface (bits 0-2, rogue/mod.rs:137-147) | door<<3 | CellAttr<<4 (field.rs:107-124) */
int rg_debug_fetch(rg_t *h, int env, rg_debug_state *out, uint16_t *cells);
/* Parity/debug: every env generates its next dungeon level and places the player there (Dungeon::new_level, rogue/mod.rs:434-481 +
 * actions::new_level, actions.rs:121-138) as on a successful '>' but without the turn around it (no hunger tick, no monster move,
 * no step count).  Lets tests reach levels 2..30 of thousands of seeds directly.  Mirrors are redrawn at the next read. */
int rg_debug_descend(rg_t *h);

#ifdef __cplusplus
}
#endif
#endif
