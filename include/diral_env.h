/*
 * diral_env.h - C-ABI of libdiral_env.so: the MI355X-native, batched V2V
 * resource-allocation environment step.
 *
 * The reference (gundoganalperen/DIRAL) has NO FFI/plugin layer: the env is a
 * duck-typed Python object (`TestEnv`, envs/test_env.py:6) that agents and the
 * driver call directly.  This header is the boundary a maintainer would bind
 * with ctypes to replace that object's hot path; every entry point names the
 * reference method(s) it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types.  Every data pointer is a
 *    DEVICE pointer (HBM) unless its name ends in `_host`.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    entry points that take a stream only ENQUEUE work on it (no host sync, no
 *    allocation) - safe inside hipGraph capture.  Exceptions, which SYNCHRONISE
 *    the stream: diral_env_check, diral_env_set_trace, and diral_env_reset /
 *    diral_env_import_state when they are given y positions (the host keeps an
 *    "every y == 0" flag that selects the kernel instantiation).
 *  - every entry point runs with the handle's device current and restores the
 *    caller's current device before it returns.
 *  - B = parallel envs (independent episodes), N = num_users, A = num_channels,
 *    S = state_space.  Batched arrays are row-major [B][N], [B][N][A], [B][N][S].
 *  - return value: 0 = DIRAL_OK, <0 = DiralStatus error; diral_env_strerror().
 *  - a handle is bound to one device and is NOT thread-safe.
 */
#ifndef DIRAL_ENV_H
#define DIRAL_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIRAL_ABI_VERSION 8

/* ---- status codes --------------------------------------------------------- */
typedef enum DiralStatus {
  DIRAL_OK = 0,
  DIRAL_ERR_BAD_ARG = -1,      /* null pointer, bad size, bad enum            */
  DIRAL_ERR_BAD_CONFIG = -2,   /* config the reference itself cannot run, or a
                                  combination this build rejects */
  DIRAL_ERR_UNSUPPORTED = -3,  /* valid reference config outside this build's
                                  limits (e.g. N > 256)                        */
  DIRAL_ERR_HIP = -4,          /* a HIP runtime call failed; see strerror      */
  DIRAL_ERR_NO_DEVICE = -5,    /* no gfx950 device / device index out of range */
  DIRAL_ERR_ACTION_RANGE = -6, /* an action outside [0, A) was seen (sticky flag
                                  raised by the step kernel; test_env.py:592)   */
  DIRAL_ERR_SEQ_OVERFLOW = -7, /* more than DIRAL_MAX_SLOTS steps since reset   */
  DIRAL_ERR_CAPTURE = -8,      /* the call would launch a ring <-> plane conversion
                                  (first step after a kernel-path switch, export /
                                  observe after ring steps) while `stream` is being
                                  captured into a hipGraph: do it outside the capture */
  DIRAL_ERR_TABLE_CONFLICT = -9, /* imported tables: two entries about one subject carry
                                  the same sequence number but different xpos - no
                                  run of the reference produces that (an entry IS the
                                  subject's stamp at that number, vehicle.py:35-63);
                                  raised by diral_env_check after an import      */
  DIRAL_ERR_PIGGY_NO_TX = -10  /* State.piggybacking: a receiver found no transmitter in range
                                  on a used resource - the reference's `self.prev_obs[tx_id]`
                                  with tx_id None, a KeyError (test_env.py:243; sticky flag
                                  raised by the step, reported by diral_env_check)      */
} DiralStatus;

/* ---- config flags: the booleans of the `EnvironmentTest` YAML block -------- */
enum {
  DIRAL_F_MOBILITY          = 1u << 0,  /* mobility            test_env.py:14  */
  DIRAL_F_MOBILITY_VARY     = 1u << 1,  /* mobility_vary       test_env.py:15  */
  DIRAL_F_TOY_WEIGHTS       = 1u << 2,  /* congestion_test     test_env.py:48, network.py:284-290 */
  DIRAL_F_ADD_ACTION        = 1u << 3,  /* State.add_action    test_env.py:29  */
  DIRAL_F_ACTION_REAL       = 1u << 4,  /* State.action_index == "real" (else "binary") test_env.py:32 */
  DIRAL_F_ADD_CHANNEL_OBS   = 1u << 5,  /* State.add_channel_obs test_env.py:41 */
  DIRAL_F_ADD_REWARD        = 1u << 6,  /* State.add_reward    test_env.py:28  */
  DIRAL_F_ADD_INDEX         = 1u << 7,  /* State.add_index     test_env.py:30  */
  DIRAL_F_ADD_VELOCITY      = 1u << 8,  /* State.add_velocity  test_env.py:31  */
  DIRAL_F_ADD_POSITION      = 1u << 9,  /* State.add_position  test_env.py:34  */
  DIRAL_F_ADD_POSDIST       = 1u << 10, /* State.add_positional_dist test_env.py:35 */
  DIRAL_F_ADD_POSDIST_PIGGY = 1u << 11, /* State.add_positional_dist_piggy test_env.py:36 */
  DIRAL_F_FINGERPRINT       = 1u << 12, /* enable_fingerprint  test_env.py:19  */
  DIRAL_F_PROPORTIONAL_FAIR = 1u << 13, /* proportional_fair   test_env.py:22  */
  DIRAL_F_DESIGN_TOPOLOGY   = 1u << 14, /* enable_design_topology test_env.py:16 */
  DIRAL_F_PIGGYBACKING      = 1u << 15, /* State.piggybacking  test_env.py:33, 71-79, 241-254, 260-264: my_step returns,
                                           per agent, its observation with the previous observation of each resource's
                                           closest transmitter inserted (np.insert at the resource's index): A * A values.
                                           Needs State.type 2 and State.add_channel_obs (DIRAL_ERR_BAD_CONFIG otherwise:
                                           type 1 makes the vector's length depend on the slot's traffic, and without the
                                           channel-observation section obtain_state never reads what get_state_space()
                                           counts); my_step_ch / my_step_design return DIRAL_ERR_BAD_CONFIG on such a handle
                                           (they hand obtain_state the plain A-wide observation, test_env.py:316, 443) */
  /* build extensions (no reference counterpart) */
  DIRAL_F_TRACK_ARRIVAL     = 1u << 16, /* keep last_arrival_time[N][N] (network.py:39-42)
                                           so diral_env_info_age() works        */
  DIRAL_F_TRACK_PRR         = 1u << 17  /* accumulate PRR metrics in my_step too */
};

/* One env configuration = the `EnvironmentTest` block (test_env.py:12-48) plus
 * its nested `State` block (test_env.py:26-41) and the driver's
 * `episode_interval` (main_test.py:226). */
typedef struct DiralCfg {
  uint32_t struct_bytes;        /* = sizeof(DiralCfg); ABI guard               */
  uint32_t flags;               /* DIRAL_F_*                                   */
  int32_t  num_users;           /* N  test_env.py:12                           */
  int32_t  num_channels;        /* A  test_env.py:13 (= action space, :44)     */
  int32_t  num_bins;            /* K  State.num_bins test_env.py:40            */
  int32_t  reward_design;       /* 1..5  test_env.py:20                        */
  int32_t  state_type;          /* State.type 1|2  test_env.py:27              */
  int32_t  posdist_type;        /* State.add_positional_dist_type 1|2  :37     */
  int32_t  episode_interval;    /* main_test.py:226 (25); done = t%EI == EI-1  */
  int32_t  info_age_limit;      /* 20: table entry valid iff last_updated < 20 (network.py:547) */
  int32_t  pf_threshold;        /* 10  test_env.py:89                          */
  int32_t  reserved0;
  double   pf_penalty;          /* -10 test_env.py:90                          */
  double   highway_length;      /* L   test_env.py:18                          */
  double   highway_height;      /* 2   network.py:31                           */
  double   communication_range; /* Rc  test_env.py:21                          */
  double   bin_range;           /* Rb  test_env.py:24                          */
} DiralCfg;

/* which reference step function a diral_env_step() call replaces */
typedef enum DiralStepMode {
  DIRAL_STEP_MY_STEP = 0,   /* TestEnv.my_step         test_env.py:124-266 */
  DIRAL_STEP_MY_STEP_CH = 1,/* TestEnv.my_step_ch      test_env.py:351-443 */
  DIRAL_STEP_DESIGN = 2     /* TestEnv.my_step_design  test_env.py:269-349 */
} DiralStepMode;

/* element type of the floating-point OUTPUT buffers (state, reward, chobs).
 * The arithmetic is always float64, like the reference; F32 is a final cast. */
typedef enum DiralDType { DIRAL_F32 = 0, DIRAL_F64 = 1 } DiralDType;

/* limits of this build (the reference has none: TestEnv takes any N, A and State.num_bins, test_env.py:12-13, 40).
 * The specialised kernels (csrc/step_fast64.hpp, step_wide.hpp: every number bench.py reports) serve num_users <= 256
 * with num_channels <= 64; 64 < num_channels <= 256 at num_users <= 256 is stepped by the general kernel
 * (csrc/step_kernel.hpp: same results, bit for bit, 2.5-4 x the time).  Beyond num_users 256, num_channels 256 or
 * num_bins 64 the step runs as three launches with the tables' columns spread over the chip (csrc/step_large.hpp,
 * DIRAL_KERNEL_LARGE: same results, bit for bit) up to the sizes below - as long as the per-env working set fits a
 * workgroup's 160 KB of LDS (28 bytes per vehicle + 8 per resource: 4096 vehicles with 4096 resources do), else
 * DIRAL_ERR_UNSUPPORTED.  diral_env_last_kernel() says which kernel ran.  Not served: State.piggybacking beyond 256
 * resources (A * A values per agent), the type-1 histogram beyond ~1400 vehicles (DIRAL_ERR_UNSUPPORTED at create). */
#define DIRAL_MAX_USERS    4096
#define DIRAL_MAX_CHANNELS 4096
#define DIRAL_MAX_BINS     1024
#define DIRAL_SMALL_MAX_USERS    256   /* the one-workgroup kernels */
#define DIRAL_SMALL_MAX_CHANNELS 256
#define DIRAL_SMALL_MAX_BINS     64
#define DIRAL_MAX_SLOTS    16777214   /* steps between resets (24-bit sequence numbers) */

/* per-env episode metric columns written by diral_env_metrics() */
enum {
  DIRAL_M_SLOTS = 0,        /* steps taken since reset/clear                    */
  DIRAL_M_SUM_REWARD,       /* sum over slots and agents of reward              */
  DIRAL_M_TX_SOLE,          /* #transmissions alone on their resource           */
  DIRAL_M_TX_COLLIDED,      /* #transmissions sharing their resource            */
  DIRAL_M_PRR_SUM,          /* sum over transmissions of R (test_env.py:402-405; 1 for a sole tx) */
  DIRAL_M_PRR_CNT,          /* #transmissions that R was summed over            */
  DIRAL_M_COLUMNS
};

typedef struct DiralEnv DiralEnv;   /* opaque handle */

/* ---- pure host helpers ----------------------------------------------------- */

/* Fill *cfg with the reference's defaults (test_env.py:12-48 kwargs.setdefault
 * values; State flags all off; episode_interval 25). */
void diral_cfg_defaults(DiralCfg* cfg);

/* Replaces TestEnv.get_state_space() (test_env.py:492; sizing :49-85).
 * Returns S >= 0, or a negative DiralStatus for an invalid config. */
int diral_env_state_space(const DiralCfg* cfg);

/* 0 if `cfg` can be run by this build, else the DiralStatus explaining why. */
int diral_env_validate(const DiralCfg* cfg);

const char* diral_env_strerror(int status);
int diral_env_abi_version(void);

/* ---- lifetime ---------------------------------------------------------------- */

/* Replaces TestEnv.__init__ -> Network.__init__ (test_env.py:7-107,
 * network.py:15-67) for B independent envs on HIP device `device`.
 * Allocates all persistent state in HBM; tables zeroed (vehicle.py:24-33).
 * Test hooks, read from the process environment ONCE here: DIRAL_NO_FAST64 /
 * DIRAL_NO_WIDE (the general kernel also for N <= 64 / N > 64), DIRAL_NO_RING
 * (N <= 64: no xpos ring, every xpos in the per-entry plane). */
int diral_env_create(const DiralCfg* cfg, int batch, int device, DiralEnv** out);
int diral_env_destroy(DiralEnv* env);

/* bytes of HBM the handle owns (for sizing batches against 288 GB) */
int64_t diral_env_hbm_bytes(const DiralEnv* env);

/* ---- handle options ---------------------------------------------------------- */
typedef enum DiralOption {
  /* global index of this handle's env 0.  The reference draws topology, actions and
   * velocity changes from unseeded global RNGs (network.py:103-110, 214;
   * test_env.py:121); here every device draw is a pure function of (seed, global
   * env index, vehicle), so a batch sharded over several handles / GPUs (shard g
   * owning envs [offset_g, offset_g + B_g)) draws exactly what one handle holding
   * the whole batch draws with the same seed. */
  DIRAL_OPT_ENV_OFFSET = 1,
  /* DIRAL_PATH_AUTO (default): configurations the specialised kernels serve run on
   * them; DIRAL_PATH_GENERAL: always the general kernel (tests compare the two);
   * DIRAL_PATH_LARGE: always the three-launch form of csrc/step_large.hpp (tests run the
   * reference's fixtures through it; not with State.piggybacking's A * A section). */
  DIRAL_OPT_KERNEL_PATH = 2
} DiralOption;
enum { DIRAL_PATH_AUTO = 0, DIRAL_PATH_GENERAL = 1, DIRAL_PATH_LARGE = 2 };
int diral_env_set_option(DiralEnv* env, int option, int64_t value);

/* which kernel the last diral_env_step / diral_env_observe call launched:
 * DIRAL_KERNEL_GENERAL, or DIRAL_KERNEL_FAST64 / DIRAL_KERNEL_WIDE or'ed with the
 * instantiation bits; negative before the first call. */
enum {
  DIRAL_KERNEL_GENERAL = 0,   /* csrc/step_kernel.hpp  */
  DIRAL_KERNEL_FAST64  = 1,   /* csrc/step_fast64.hpp, N <= 64       */
  DIRAL_KERNEL_WIDE    = 2,   /* csrc/step_wide.hpp,  64 < N <= 256  */
  DIRAL_KERNEL_OBSERVE = 3,   /* csrc/observe_kernel.hpp: diral_env_observe (stand-alone obtain_state) */
  DIRAL_KERNEL_LARGE   = 4,   /* csrc/step_large.hpp: N > 256, A > 256 or K > 64 (step and observe) */
  DIRAL_KERNEL_RICH    = 16,  /* channel-obs output / cheap State flags (csrc/rich_out.hpp) */
  DIRAL_KERNEL_EXTRA   = 32,  /* my_step_design / arrival stamps / trace replay */
  DIRAL_KERNEL_CH      = 64,  /* my_step_ch */
  DIRAL_KERNEL_RING    = 128, /* the xpos ring (the per-entry xpos plane only for old entries) */
  DIRAL_KERNEL_PACKED  = 256, /* the packed table form: thermometer codes + ages + own sequence numbers (N <= 64 always;
                                 128 < N <= 256 on dense topologies), else the (seq, age) plane */
  DIRAL_KERNEL_POLICY  = 512  /* diral_env_step_policy ran as ONE launch (reward shaping + SPS decision in the step) */
};
int diral_env_last_kernel(const DiralEnv* env);

/* ---- topology / reset ------------------------------------------------------- */

/* Replaces Network.initialize_mobility_topology (network.py:92-119) /
 * reset_positions (network.py:181-187).  x0,y0,v0: [B][N] float64 device
 * arrays; any may be NULL => drawn on device from `seed`
 * (x0 integer-uniform in [0,L), y0 = 0, v0 ~ U(1.1,2.7) or 1.7 if
 * mobility_vary).  Zeroes tables, arrival stamps, pf counters and metrics. */
int diral_env_reset(DiralEnv* env, const double* x0, const double* y0,
                    const double* v0, uint64_t seed, void* stream);

/* ---- the hot path ----------------------------------------------------------- */

/* One time-slot for all B envs in ONE fused launch.  Replaces, per env,
 *   my_step / my_step_ch / my_step_design (per `mode`) followed by
 *   obtain_state(obs, actions, rewards, episode, epsilon) (test_env.py:527-583).
 * actions   [B][N] int32 in [0,A)
 * t         the driver's time_step (used for `done`, arrival stamps)
 * state_out [B][N][S] (dtype `out_dtype`), NULL => observation not built
 * rew_out   [B][N]    (dtype `out_dtype`), NULL allowed
 * done_out  [B] uint8, NULL allowed: t % episode_interval == episode_interval-1
 * chobs_out [B][N][A] (dtype `out_dtype`), NULL allowed: the `obs` dict of the
 *           reference step (test_env.py:143, 206, 228, 240).  With DIRAL_F_PIGGYBACKING
 *           [B][N][A * A]: the `piggy_obs` dict my_step returns instead (test_env.py:263-264),
 *           and the env keeps this slot's plain observation as the next slot's `prev_obs`
 *           (test_env.py:260-261)
 * episode, epsilon: only used with DIRAL_F_FINGERPRINT (test_env.py:577-579) */
int diral_env_step(DiralEnv* env, int mode, const int32_t* actions, int64_t t,
                   void* state_out, void* rew_out, uint8_t* done_out,
                   void* chobs_out, int out_dtype, double episode,
                   double epsilon, void* stream);

/* Observation only, no state change: replaces a stand-alone
 * TestEnv.obtain_state(obs, acts, rewards, episode, eps) (test_env.py:527-583)
 * on the CURRENT tables/positions.  chobs_in [B][N][A] ([B][N][A * A] with
 * DIRAL_F_PIGGYBACKING) and rew_in [B][N] are float64 device arrays (NULL allowed
 * when the State flags do not use them). */
int diral_env_observe(DiralEnv* env, const int32_t* actions,
                      const double* chobs_in, const double* rew_in,
                      void* state_out, int out_dtype, double episode,
                      double epsilon, void* stream);

/* Replaces TestEnv.update_velocity -> Network.update_velocity
 * (test_env.py:498-504, network.py:208-223).  draws [B][N] uint8 in {1,2,3}
 * (random.randrange(1,4)); NULL => drawn on device from `seed`.
 * No-op unless DIRAL_F_MOBILITY_VARY. */
int diral_env_update_velocity(DiralEnv* env, const uint8_t* draws,
                              uint64_t seed, void* stream);

/* Replaces TestEnv.sample (test_env.py:116-122): uniform actions [B][N]. */
int diral_env_sample(DiralEnv* env, int32_t* actions_out, uint64_t seed,
                     void* stream);

/* Replaces Network.get_information_age(t) (network.py:560-574):
 * out [B][100] int32.  Needs DIRAL_F_TRACK_ARRIVAL. */
int diral_env_info_age(DiralEnv* env, int64_t t, int32_t* out, void* stream);

/* Replaces Network.load_x_positions + the replay branch of update_positions
 * (network.py:171-178, 194-199; TestEnv.load_saved_positions, test_env.py:109-114):
 * after each step pos_x[u] = x_positions[t % T][u] instead of the velocity move.
 * x_positions: float64 device array [T][N] (per_env == 0, shared by all envs, the
 * reference's shape) or [B][T][N] (per_env != 0).  The trace is COPIED into
 * handle-owned HBM (synchronises `stream`); T == 0 or NULL removes it. */
int diral_env_set_trace(DiralEnv* env, const double* x_positions, int T, int per_env, void* stream);

/* ---- state export / import (checkpoint, golden replay, debugging) --------- */

/* Reference-shaped copies of the env state, all device pointers, any NULL
 * skipped.  pos_x,pos_y,vel [B][N] f64; tab_seq, tab_age [B][N][N] int32 and
 * tab_x, tab_y [B][N][N] f64 indexed [env][viewer][subject]
 * (Vehicle.pos_of_neighbors, vehicle.py:20-33; age saturates at 255);
 * last_arrival [B][N][N] int32 indexed [env][tx][rx] (network.py:39-42).
 * Imported tables must be states the reference can reach in this one respect:
 * entries about one subject that carry equal sequence numbers carry equal xpos
 * (an entry IS the subject's stamp at that number, vehicle.py:56-63; every
 * exported state qualifies).  The kernels build on it: N <= 64 keeps the xpos of
 * entries at most 7 stamps old in a per-subject ring instead of the per-entry
 * plane, N > 64 routes xpos through a rank-indexed table; export / observe /
 * other consumers see the plane completed first (no caller-visible difference).
 * An import that violates the requirement is detected: the next
 * diral_env_check() returns DIRAL_ERR_TABLE_CONFLICT.  A call that needs the
 * plane <-> ring conversion launch (the first step after a kernel-path switch,
 * export / observe after ring steps) while `stream` is being captured into a
 * hipGraph returns DIRAL_ERR_CAPTURE and launches nothing: a captured
 * conversion would replay against tables it no longer describes. */
int diral_env_export_state(DiralEnv* env, double* pos_x, double* pos_y,
                           double* vel, int32_t* tab_seq, int32_t* tab_age,
                           double* tab_x, double* tab_y, int32_t* last_arrival,
                           void* stream);
int diral_env_import_state(DiralEnv* env, const double* pos_x,
                           const double* pos_y, const double* vel,
                           const int32_t* tab_seq, const int32_t* tab_age,
                           const double* tab_x, const int32_t* last_arrival,
                           void* stream);

/* DIRAL_F_PIGGYBACKING: TestEnv.prev_obs (test_env.py:76-79, 260-261), the plain channel
 * observation of the last my_step, [B][N][A] float64 device arrays; zeros after a reset.
 * Handles without the flag return DIRAL_ERR_BAD_CONFIG. */
int diral_env_export_prev_obs(DiralEnv* env, double* prev_obs, void* stream);
int diral_env_import_prev_obs(DiralEnv* env, const double* prev_obs, void* stream);

/* The same tables as 16-byte records in the field order of the RealNeS bridge's
 * MA_NeighborTableEntry message (envs/ma_messages_pb2.py:195-230: float pos_x,
 * float pos_y, int32 seq_num, int32 last_update - what
 * realness_bridge.py:168-191 unpacks into the pos_of_neighbors dict the
 * observation code reads).  entries [B][N][N] device records indexed
 * [env][viewer][subject].  export narrows the f64 positions to f32 (round to
 * nearest); import widens pos_x exactly, ignores pos_y (an entry's ypos is the
 * subject's lane) and saturates last_update at 255 like
 * diral_env_import_state, whose reachability note applies. */
typedef struct DiralNeighborEntry {
  float pos_x, pos_y;
  int32_t seq_num, last_update;
} DiralNeighborEntry;
int diral_env_export_entries(DiralEnv* env, DiralNeighborEntry* entries, void* stream);
int diral_env_import_entries(DiralEnv* env, const DiralNeighborEntry* entries, void* stream);

/* ---- metrics ------------------------------------------------------------------ */

/* out [B][DIRAL_M_COLUMNS] float64 device array; clear != 0 zeroes the
 * accumulators afterwards. */
int diral_env_metrics(DiralEnv* env, double* out, int clear, void* stream);

/* Sticky device-side error flags (action range, sequence overflow, piggybacking without a transmitter) raised by
 * kernels since the last call.  SYNCHRONISES `stream`.  Returns DIRAL_OK or the
 * first error. */
int diral_env_check(DiralEnv* env, void* stream);

/* ---- the driver's reward post-processing (main_test.py:150-206) ----------------- */

/* What `marl_test` does to the rewards of one slot between `env.my_step*` and
 * `memory.add`, for `envs` envs in ONE launch (stateless: the caller owns the counters):
 *   sum_r      = np.sum(reward) in NumPy's pairwise order (:171), collision = A - sum_r (:178)
 *   ia_sum     = utils/misc.calculate_ia_penalty(ia) = sum (i+1)*ia[i]      (:151; ia [envs][100] or NULL)
 *   ia_penalty = -1 / +1 / 0 as ia_sum rose / fell / stayed vs *sum_ia_prev  (:153-160; flags bit 1)
 *   reward'    = reward + ia_penalty (:190-192); counter penalty: an agent that repeats an
 *                unsuccessful action (reward' < 1) more than `ia_penalty_threshold` times gets
 *                `ia_penalty_value` (:194-203; flags bit 2; pen_counter, prev_actions [envs][N] in/out);
 *                + sum_r / N (global_reward_avg, :205-206; flags bit 0)
 * reward_in / reward_out / sum_r_out / collision_out: `dtype` (arithmetic in that type, like the
 * torch statement diral_amd/driver.py keeps for the CPU-backed tests; float64 = the reference's).
 * Any output pointer may be NULL except reward_out. */
int diral_driver_shape(int envs, int num_users, int num_channels, const void* reward_in, int dtype,
                       const int32_t* actions, const int32_t* ia, int64_t* sum_ia_prev, int32_t* pen_counter,
                       int32_t* prev_actions, int flags, int ia_penalty_threshold, double ia_penalty_value,
                       void* reward_out, void* sum_r_out, void* collision_out, int64_t* ia_sum_out,
                       int32_t* ia_penalty_out, void* stream);

/* ---- SPS baseline policy (agent side, stateless entry points) ----------------- */

/* Replaces SemiPersistentScheduling.step + choose_new_resource
 * (algorithms/v2x_sps.py:76-104, 24-74) for `agents` independent agents in one
 * launch.  selection_window [agents][A] float64: the averaged-RSSI vector each
 * agent senses (lower = better); prev_action, counter [agents] int32: the
 * per-agent state (self.prev_action, self.reselection_counter), updated in place;
 * actions_out [agents] int32.  The reference draws from Python's global RNG; here
 * every draw is injectable (NULL => counter-based device RNG from `seed`):
 *   draw_counter [agents] int32 in [5,16]   random.randint(5, 16)      (v2x_sps.py:91)
 *   draw_keep    [agents] float64 in [0,1)  random.random()           (v2x_sps.py:93)
 *   draw_choice  [agents] int32 >= 0        random.choice(sB) = sB[draw % len(sB)] (v2x_sps.py:57)
 * rssi_threshold, inc_db (3), keep_prob (0.8): v2x_sps.py:12,18,22. */
int diral_sps_step(int agents, int num_channels, const double* selection_window, int32_t* prev_action,
                   int32_t* counter, double rssi_threshold, double inc_db, double keep_prob,
                   const int32_t* draw_counter, const double* draw_keep, const int32_t* draw_choice,
                   uint64_t seed, int32_t* actions_out, void* stream);

/* Build extension - the reference never wires its SPS agent to the toy env, so the map from
 * the env's channel observation to the "averaged RSSI" window SPS consumes is this build's:
 * window[a][i] = -60 on the agent's own resource (actions[a] == i); else from
 * d = chobs[a][i] (test_env.py:206, 240; network.py:385): d >= 100000 (busy, nobody in
 * range) -> -160; d > 0 -> -40 - 30 log10(max(d, 1)) (log-distance path loss, dB);
 * d == 0 (idle) -> -200.  chobs [agents][A] of `chobs_dtype`, window_out [agents][A] f64. */
int diral_sps_window_from_chobs(int agents, int num_channels, const void* chobs, int chobs_dtype,
                                const int32_t* actions, double* window_out, void* stream);

/* diral_sps_window_from_chobs + diral_sps_step in ONE launch (same decisions, bit for bit):
 * only the agents that re-select this slot build their window, in private memory.
 * num_channels <= 64, else DIRAL_ERR_UNSUPPORTED (use the two calls). */
int diral_sps_step_chobs(int agents, int num_channels, const void* chobs, int chobs_dtype,
                         const int32_t* actions, int32_t* prev_action, int32_t* counter,
                         double rssi_threshold, double inc_db, double keep_prob,
                         const int32_t* draw_counter, const double* draw_keep,
                         const int32_t* draw_choice, uint64_t seed, int32_t* actions_out, void* stream);

/* ---- the closed loop of a policy-only rollout as ONE launch per slot -----------------------------
 * The reference's slot is env step -> reward shaping -> policy (main_test.py:127-206 with algorithms/v2x_sps.py as the
 * policy); diral_env_step + diral_driver_shape + diral_sps_step_chobs are three dependent launches and the channel
 * observation travels through HBM between the first and the third.  diral_env_step_policy runs the same slot as one
 * launch where the configuration allows (N <= 64 on the one-lane highway, my_step, none of the run-time extras: the
 * channel observation is staged in LDS by the step and the agents decide from there), and as the three launches
 * otherwise (then `chobs_out` must be given) - same results either way, bit for bit:
 *   shaped_out / sum_r_out / collision_out, pen_*: diral_driver_shape without the information-age terms
 *     (shape_flags: bit 0 global_reward_avg, bit 2 stuck-action penalty); shaped_out NULL = no shaping;
 *   sps_*, draws, seed: diral_sps_step_chobs (seed_clock != NULL: diral_sps_step_chobs_clocked, injected draws ignored
 *     there too); actions_out [B][N]: the actions of the NEXT slot.
 * `chobs_out` may be NULL when the slot runs fused: the observation then never leaves the chip. */
typedef struct DiralSlotPolicy {
  uint32_t struct_bytes;          /* = sizeof(DiralSlotPolicy) */
  int32_t  shape_flags;
  int32_t  pen_threshold;
  int32_t  reserved0;
  double   pen_value;
  void*    shaped_out;            /* [B][N] out dtype or NULL */
  void*    sum_r_out;             /* [B] or NULL */
  void*    collision_out;         /* [B] or NULL */
  int32_t* pen_counter;           /* [B][N] (shape_flags bit 2) */
  int32_t* pen_prev_actions;      /* [B][N] */
  int32_t* sps_prev_action;       /* [B][N] */
  int32_t* sps_counter;           /* [B][N] */
  double   rssi_threshold, inc_db, keep_prob;
  const int32_t* draw_counter;    /* [B][N] or NULL */
  const double*  draw_keep;
  const int32_t* draw_choice;
  uint64_t seed;
  const int64_t* seed_clock;      /* NULL, or a device counter added to the seed */
  int32_t* actions_out;           /* [B][N] */
  /* K slots in ONE launch (ABI 6).  slots <= 1: one slot, as above.  slots = K > 1 (configurations the fused kernel takes -
   * N <= 64, my_step, the flat highway; otherwise DIRAL_ERR_UNSUPPORTED with nothing launched): the workgroup of an env
   * keeps it on the chip for K slots of [env step -> shaping -> SPS decision], `actions` being slot 0's and the policy's
   * decisions the later ones'.  Equal, bit for bit, to K one-slot calls with t, t + 1, ... and seed, seed + 1, ... (plus
   * diral_env_update_velocity(env, NULL, vel_seed + slot / episode_interval) behind every slot that ends an episode, when
   * the config has mobility_vary) - tables, positions, velocities, metrics, policy state, actions_out (the actions of
   * slot t + K).  state_out / rew_out / done_out / chobs_out: of the LAST slot (each may be NULL; without state_out no
   * slot computes the positional histogram); shaped_out / sum_r_out / collision_out: [K][...] arrays, slot-major.
   * Injected draws (draw_*) must be NULL. */
  int32_t  slots;
  int32_t  reserved1;
  uint64_t vel_seed;
} DiralSlotPolicy;
int diral_env_step_policy(DiralEnv* env, int mode, const int32_t* actions, int64_t t, void* state_out, void* rew_out,
                          uint8_t* done_out, void* chobs_out, int out_dtype, const DiralSlotPolicy* policy,
                          void* stream);

/* The driver's random prefill (main_test.py:99-114) as ONE launch of K slots (ABI 8; step_fast64_slots_kernel):
 *   for k = 0 .. K - 1:   a_k = (k == 0) ? actions : TestEnv.sample()          [diral_env_sample(env, ., seed + k, .)]
 *                         obs, _ = my_step_design(a_k, 0)                        [test_env.py:269-349]
 *                         states_out[k] = obtain_state(obs, a_k, rews)           [rews = rew_in: the bootstrap step's, :110]
 * with the env kept on the chip from slot to slot.  Equal, bit for bit, to that loop of diral_env_sample +
 * diral_env_step(DIRAL_STEP_DESIGN) + diral_env_observe calls: states, tables, positions, metrics.
 *   actions [B][N]            slot 0's actions (e.g. diral_env_sample(env, actions, seed, stream))
 *   states_out [K][B][N][S]   out dtype, or NULL (no state is built)
 *   actions_all_out [K][B][N] a_k, or NULL          actions_next_out [B][N]  the draw of seed + K (mandatory: what the
 *                                                    next call passes as `actions`, with seed + K)
 *   rew_in [B][N] float64     the reward column of the state vectors (State.add_reward), or NULL: each slot's own reward
 * Configurations the fused kernel takes - N <= 64 (>= 8), A <= 64, the one-lane highway, piggybacked tables, no arrival /
 * PRR tracking, no trace replay, no secondary observation mode; otherwise DIRAL_ERR_UNSUPPORTED with nothing launched
 * (loop over the three calls instead: diral_amd.driver.DriverLoop.prefill does). */
int diral_env_prefill(DiralEnv* env, const int32_t* actions, int32_t slots, uint64_t seed, void* states_out, int out_dtype,
                      int32_t* actions_all_out, int32_t* actions_next_out, const double* rew_in, double episode,
                      double epsilon, void* stream);

/* ---- slot clock: rollouts captured into a hipGraph ---------------------------------------------
 * A captured sequence of K slots (env step, reward shaping, policy) bakes every by-value argument into its
 * kernel nodes; what has to move on from replay to replay - the slot number behind `done`, arrival stamps and
 * trace replay; the seed of the policy's draws - is read from a DEVICE counter instead.
 *   diral_env_set_clock(env, t_dev): from now on diral_env_step's `t` is an OFFSET: the slot number is
 *     *t_dev + t, read by the kernel (t_dev NULL: back to the by-value slot number).  int64 in HBM, owned by
 *     the caller, alive as long as it is set.
 *   diral_clock_add(clock, inc, stream): *clock += inc as a one-thread launch (the last node of the graph).
 *   diral_sps_step_chobs_clocked: diral_sps_step_chobs with device draws seeded by seed + *clock. */
int diral_env_set_clock(DiralEnv* env, const int64_t* t_dev);
/* Slow-first dispatch inside captured graphs (N <= 64).  A step launch leaves the list of the envs
 * it found slow for the NEXT launch in one of three rotating sets; the host counts launches to know which set a launch
 * reads, builds and clears.  A captured launch is baked with its sets, so by default it only READS the set of the last
 * eager launch and builds none - correct, but the list ages with the replays.
 *   diral_env_set_capture_rotation(env, 1, &phase): from now on captured step launches rotate the sets like eager ones.
 *     The caller promises: every graph captured while this is on holds a MULTIPLE OF 3 step launches of this env, is
 *     replayed whole, and diral_env_align_phase(env, phase, stream) is called (on the replay's stream) before a replay
 *     that follows anything but another replay of the same graph.  *phase (may be NULL) = the env's phase (launch count
 *     mod 3) at the call: the phase of the first launch captured next.  (…, 0, …): back to the default.
 *   diral_env_align_phase(env, phase, stream): make the env's phase `phase`; if it has to move, the three sets are
 *     emptied first (an empty list is always a valid one: the next launch runs every env in dispatch order). */
int diral_env_set_capture_rotation(DiralEnv* env, int on, int* phase);
int diral_env_align_phase(DiralEnv* env, int phase, void* stream);
int diral_clock_add(int64_t* clock, int64_t inc, void* stream);
int diral_sps_step_chobs_clocked(int agents, int num_channels, const void* chobs, int chobs_dtype,
                                 const int32_t* actions, int32_t* prev_action, int32_t* counter,
                                 double rssi_threshold, double inc_db, double keep_prob, uint64_t seed,
                                 const int64_t* clock, int32_t* actions_out, void* stream);

/* Replaces SemiPersistentScheduling.__init__ (v2x_sps.py:8-22): prev_action =
 * randint(0, selection_window) (inclusive, as in the reference), counter =
 * randint(5, 15); device RNG from `seed`. */
int diral_sps_init(int agents, int selection_window, int32_t* prev_action, int32_t* counter, uint64_t seed,
                   void* stream);

/* last HIP error string seen by this handle (host-side, for DIRAL_ERR_HIP) */
const char* diral_env_last_hip_error(const DiralEnv* env);

#ifdef __cplusplus
}
#endif
#endif /* DIRAL_ENV_H */
