// diral_env.hip - C-ABI implementation of include/diral_env.h for gfx950.
// Host side: config validation, HBM allocation, kernel launches.  No torch
// types anywhere; every data pointer in the ABI is a device pointer.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "aux_kernels.hpp"
#include "common.hpp"
#include "launch.hpp"
#include "observe_kernel.hpp"
#include "step_kernel.hpp"
#include "step_fast64.hpp"
#include "step_wide.hpp"
#include "posdist_kernel.hpp"
#include "piggyback_kernel.hpp"

using namespace diral;

struct DiralEnv {
  DiralCfg cfg;
  int B = 0, N = 0, A = 0, K = 0, S = 0, NV = 0, NR = 0, vpl = 1;
  int device = 0;
  StepParams base;       // everything that does not change per call
  RichParams rich;       // section offsets etc. of the RICH output tail (rich_out.hpp)
  uint32_t lds_bytes = 0;
  int64_t env_offset = 0;  // global index of env 0 (DIRAL_OPT_ENV_OFFSET): device RNG draws are indexed globally
  int kernel_path = DIRAL_PATH_AUTO;   // DIRAL_OPT_KERNEL_PATH
  int last_kernel = -1;    // DIRAL_KERNEL_* of the last step / observe launch
  double* pos_x = nullptr;
  double* pos_y = nullptr;
  double* vel = nullptr;
  uint32_t* tkey = nullptr;
  double* tx = nullptr;
  // the xpos ring of the N <= 64 kernel (aux_kernels.hpp): while it is in use the per-entry plane `tx` is only
  // kept for entries older than the ring reaches; `plane_valid` / `ring_valid` say which of the two is complete
  double* ring = nullptr;
  bool plane_valid = true, ring_valid = false;
  // the packed table of step_fast64 (N <= 64; step_fast64.hpp): thermometer codes of the entries' lags and their ages,
  // four per word, the subjects' own sequence numbers, the old-quad flags.  It travels with the ring: `ring_valid`
  // = ring AND packed words are current; `plane_valid` = the planes tkey / tx are complete and current
  uint32_t* tcode = nullptr;
  uint32_t* tage = nullptr;
  uint32_t* tseq = nullptr;
  uint32_t* told = nullptr;
  // slow envs first (step_fast64.hpp FastParams::slow_*): three rotating sets of [count (16 words) | list | flag per env]
  uint32_t* slow = nullptr;
  uint64_t slow_launches = 0;   // launches that rotated the sets
  bool slow_first = true;       // DIRAL_NO_SLOW_FIRST=1 at create: blocks = envs in order (A/B timing, tests)
  bool capture_rotates = false; // diral_env_set_capture_rotation: captured launches rotate the sets too (graphs of 3 k launches)
  bool wide_slow_first = false;  // DIRAL_WIDE_SLOW_FIRST=1 at create: step_wide's packed form at N <= 128 dispatches its slow envs first (round 5's default)
  bool type1_wide_lanes = false; // DIRAL_TYPE1_LANES=wide at create: rounds 3-4's 64 values per lane in posdist_type1_lanes_kernel (A/B)
  int f32_margin = -1;          // DIRAL_F32_MARGIN=<n> at create: 0 = no float32 screening of the bin, n > 0 = a band of at
                                //   least n / 65536 bin widths (tests: a wide band sends many entries to float64); -1 = the bound
  int32_t* la = nullptr;
  int32_t* pf = nullptr;
  double* metrics = nullptr;
  uint32_t* err = nullptr;
  double* edges = nullptr;
  double* edges1 = nullptr;
  double* inv_tab = nullptr;   // [256] 1.0 / n (n = 0: 0): f32 histogram output, csrc/step_fast64.hpp
  double* trace = nullptr;    // handle-owned copy of the replay trace
  int trace_len = 0, trace_per_env = 0;   // np.linspace(-1, 1, K+1) for the type-1 histogram
  int64_t hbm_bytes = 0;
  bool flat_y = true;      // every pos_y == 0 (random topologies, network.py:104): |dx| distance path
  uint32_t* yflag = nullptr;
  // State.piggybacking (piggyback_kernel.hpp): TestEnv.prev_obs, and this slot's plain observation / closest transmitters
  int off_chobs_pb = -1;       // column of the A * A section in a state row (the kernels are handed off_chobs = -1)
  double* prev_obs = nullptr;
  double* obs_new = nullptr;
  int32_t* txid = nullptr;
  // num_users > 256 / num_channels > 256 / num_bins > 64 (step_large.hpp): no one-workgroup kernel holds such an env
  bool large = false;
  LargeScratch lg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned long long* dbg = nullptr;   // DIRAL_TIMING builds: [B][waves][8] timestamps
  bool capture_violation = false;   // a ring <-> plane switch was asked for inside a stream capture
  std::string last_hip_error;
};

namespace {

bool has(const DiralCfg* c, uint32_t f) { return (c->flags & f) != 0; }

int note_hip(DiralEnv* e, hipError_t st, const char* what) {
  if (st == hipSuccess) return DIRAL_OK;
  if (e && e->capture_violation) {
    e->capture_violation = false;
    e->last_hip_error = std::string(what) + ": the call needs a ring <-> plane conversion launch, which cannot be part of a stream capture";
    return DIRAL_ERR_CAPTURE;
  }
  if (e) e->last_hip_error = std::string(what) + ": " + hipGetErrorString(st);
  return DIRAL_ERR_HIP;
}

#define HIP_TRY(env, call)                                   \
  do {                                                       \
    hipError_t st__ = (call);                                \
    if (st__ != hipSuccess) return note_hip(env, st__, #call); \
  } while (0)

// np.linspace(start, stop, num), endpoint=True (numpy/_core/function_base.py):
// y[i] = i*step + start with both roundings, last = stop.
void np_linspace(double start, double stop, int num, std::vector<double>& out) {
  out.assign(num, 0.0);
  const int div = num - 1;
  const double delta = stop - start;
  if (div > 0) {
    const double step = delta / (double)div;
    for (int i = 0; i < num; ++i) {
      volatile double y = (double)i;
      if (step == 0.0) { y = y / (double)div; y = y * delta; }
      else y = y * step;
      volatile double r = y + start;
      out[i] = r;
    }
    out[num - 1] = stop;
  } else if (num == 1) {
    out[0] = start;
  }
}

struct Offsets { int act, chobs, posdist, hist, rew, idx, pos, vel, fp, S; };

// section order of TestEnv.obtain_state (test_env.py:527-583)
Offsets state_offsets(const DiralCfg* c) {
  Offsets o{-1, -1, -1, -1, -1, -1, -1, -1, -1, 0};
  int p = 0;
  if (has(c, DIRAL_F_ADD_ACTION)) { o.act = p; p += has(c, DIRAL_F_ACTION_REAL) ? 1 : c->num_channels; }
  // (State.piggybacking: `obs[user_i]` is piggy_obs, A * A values - test_env.py:71-72, 263-264, 539-541)
  if (has(c, DIRAL_F_ADD_CHANNEL_OBS)) { o.chobs = p; p += has(c, DIRAL_F_PIGGYBACKING) ? c->num_channels * c->num_channels : c->num_channels; }
  if (has(c, DIRAL_F_ADD_POSDIST)) { o.posdist = p; p += c->num_users - 1; }
  if (has(c, DIRAL_F_ADD_POSDIST_PIGGY)) { o.hist = p; p += c->num_bins; }
  if (has(c, DIRAL_F_ADD_REWARD)) { o.rew = p; p += 1; }
  if (has(c, DIRAL_F_ADD_INDEX)) { o.idx = p; p += 1; }
  if (has(c, DIRAL_F_ADD_POSITION)) { o.pos = p; p += 2; }
  if (has(c, DIRAL_F_ADD_VELOCITY)) { o.vel = p; p += 1; }
  if (has(c, DIRAL_F_FINGERPRINT)) { o.fp = p; p += 2; }
  o.S = p;
  return o;
}

int vpl_for(int N) { return N <= 64 ? 1 : (N <= 128 ? 2 : 4); }

// sizes no one-workgroup kernel holds: the three launches of step_large.hpp
bool is_large_cfg(const DiralCfg* c) {
  if (c->num_users > DIRAL_SMALL_MAX_USERS || c->num_channels > DIRAL_SMALL_MAX_CHANNELS ||
      (has(c, DIRAL_F_ADD_POSDIST_PIGGY) && c->num_bins > DIRAL_SMALL_MAX_BINS))
    return true;
  // ... and the combinations below those sizes whose env does not fit the general kernel's workgroup (e.g. 200 vehicles,
  // 200 resources, 64 bins): every handle can fall back on that kernel, so they run here as well
  const int vpl = vpl_for(c->num_users);
  return lds_layout(64 * vpl, c->num_channels, c->num_bins > 0 ? c->num_bins : 1, vpl, 4 * vpl).total > 160u * 1024u;
}
bool runs_large(const DiralEnv* e) { return e->large || e->kernel_path == DIRAL_PATH_LARGE; }

// The table form of a handle (fixed at create): packed thermometer codes + ages + own sequence numbers, or the (seq, age)
// plane `tkey` of round 2.  N <= 64: always packed (step_fast64 has no other form).  64 < N <= 256 (step_wide): packed
// where the vehicles are dense enough for tables to stay fresh - on average at least kPackedMinNeighbours vehicles within
// communication range (N * 2 Rc / L; BASELINE configs[2]: 32, configs[4]: 16) -, else the plane: on a sparse highway most
// entries lag their subject by more than the 7 stamps the codes carry, and every pass of the packed form would detour
// through the planes (N > 128: measured + 60 % at 2.4 neighbours, + 58 % at 10.7).  At configs[4]'s density one env in
// ten has broken into clusters and runs nearly all its passes through the planes at three times the price of a plane-form
// pass, the other nine run coded passes at a third of it: 1.11 -> 1.02-1.04 ms with those envs dispatched first (step_wide.hpp).
// DIRAL_TABLE_FORM = packed | plane in the environment overrides the choice for N > 64 (tests run both forms on both
// kinds of topology).
constexpr double kPackedMinNeighbours = 20.0;
constexpr double kPackedMinNeighbours2 = 15.0;           // 64 < N <= 128
bool use_packed_table(const DiralEnv* e) {
  if (e->vpl == 1) return true;
  if (const char* f = std::getenv("DIRAL_TABLE_FORM")) {
    if (std::strcmp(f, "packed") == 0) return true;
    if (std::strcmp(f, "plane") == 0) return false;
  }
  const double neigh = e->N * 2.0 * e->cfg.communication_range / e->cfg.highway_length;
  return neigh >= (e->vpl == 2 ? kPackedMinNeighbours2 : kPackedMinNeighbours);
}

// Every entry point that touches the device runs with the handle's device current and
// restores the caller's (torch's) current device afterwards.
struct DeviceGuard {
  int prev = -1, dev;
  bool ok = true;
  explicit DeviceGuard(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

// State flags that only add OUTPUT columns to the state vector (test_env.py:527-583): served
// by the RICH instantiations of the specialised kernels (rich_out.hpp)
constexpr uint32_t kRichFlags = DIRAL_F_ACTION_REAL | DIRAL_F_ADD_CHANNEL_OBS | DIRAL_F_ADD_REWARD | DIRAL_F_ADD_INDEX |
                                DIRAL_F_ADD_VELOCITY | DIRAL_F_ADD_POSITION | DIRAL_F_FINGERPRINT | DIRAL_F_PIGGYBACKING;

// Configurations the specialised kernels (step_fast64 / step_wide) serve: every step kind on a mobile
// topology with piggybacked neighbour tables, any combination of outputs and State flags, proportional
// fairness.  The type-2 piggybacked histogram is part of the kernels; the secondary observation modes
// (a15/a16: sorted true distances, type-1 weighted histogram) are columns posdist_kernel fills right
// after the step, so the step itself runs here all the same (RICH instantiation, histogram columns off
// unless type 2).  A State block without piggybacked tables (test_env.py:138, 231-238: no gossip at all), PRR
// tracking in my_step and static topologies are run-time switches of the EXTRA instantiations.  What is left
// for the general kernel: A > 64, and vehicles off the common lane at N > 64 (nothing the reference draws).
bool is_specialised_cfg(const StepParams& p) {
  const uint32_t want = 0u;
  const uint32_t ignore = DIRAL_F_MOBILITY | DIRAL_F_ADD_POSDIST_PIGGY | DIRAL_F_TOY_WEIGHTS | DIRAL_F_MOBILITY_VARY | DIRAL_F_DESIGN_TOPOLOGY | DIRAL_F_TRACK_ARRIVAL |
                          DIRAL_F_TRACK_PRR | DIRAL_F_ADD_ACTION | DIRAL_F_PROPORTIONAL_FAIR | DIRAL_F_ADD_POSDIST | kRichFlags;
  return (p.flags & ~ignore) == want &&
         (p.mode == DIRAL_STEP_MY_STEP || p.mode == DIRAL_STEP_MY_STEP_CH || p.mode == DIRAL_STEP_DESIGN);
}
// the state vector carries the type-2 piggybacked histogram
bool has_hist2(const StepParams& p) { return (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) && p.posdist_type == 2; }
// ... of which the PLAIN instantiations serve the toy YAML's State flags with the state vector
// as the only observation output (the metric's configuration)
bool is_plain_cfg(const StepParams& p) {
  return (p.flags & (kRichFlags | DIRAL_F_PROPORTIONAL_FAIR | DIRAL_F_ADD_POSDIST)) == 0 && has_hist2(p) &&
         (p.flags & DIRAL_F_ADD_ACTION) && p.state_out != nullptr && p.chobs_out == nullptr;
}

int blocks(size_t total, int threads) { return (int)((total + threads - 1) / threads); }

// The handle-less entry points (diral_sps_*, diral_driver_shape) take device pointers only: they run on the
// device that owns `p` (their first mandatory buffer), whatever device is current in the calling thread.
int device_of_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (p && hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
  (void)hipGetLastError();
  return -1;
}
struct PtrDeviceGuard {
  int prev = -1, dev = -1;
  bool ok = true;
  explicit PtrDeviceGuard(const void* p) : dev(device_of_ptr(p)) {
    if (dev < 0) return;                                        // not a device allocation HIP knows: current device
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~PtrDeviceGuard() { if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

// A switch between the xpos ring and the per-entry plane (ring_rebuild / ring_materialize) is a launch that
// depends on HOST-side validity flags: recorded into a hipGraph it would replay against tables it no longer
// describes.  Such a call fails with DIRAL_ERR_CAPTURE instead (do the first step of a new kernel path, or the
// export / observe, outside the capture).
bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

// every consumer of the per-entry xpos plane other than step_fast64 goes through here first
hipError_t ensure_plane(DiralEnv* e, hipStream_t s) {
  if (e->plane_valid || !e->ring) { e->plane_valid = true; return hipSuccess; }
  if (stream_is_capturing(s)) { e->capture_violation = true; return hipErrorStreamCaptureUnsupported; }
  if (e->tcode) {
    const size_t total = (size_t)e->B * (e->NR / 4) * e->NV;
    hipLaunchKernelGGL(unpack_codes_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tcode,
                       e->tage, e->tseq, e->ring, e->tkey, e->tx);
  } else {
    const size_t total = (size_t)e->B * e->N * e->N;
    hipLaunchKernelGGL(ring_materialize_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey,
                       e->ring, e->tx);
  }
  e->plane_valid = true;
  return hipGetLastError();
}
hipError_t ensure_ring(DiralEnv* e, hipStream_t s) {
  if (e->ring_valid) return hipSuccess;
  if (stream_is_capturing(s)) { e->capture_violation = true; return hipErrorStreamCaptureUnsupported; }
  const size_t total = (size_t)e->B * e->N * e->N;
  hipLaunchKernelGGL(ring_rebuild_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey,
                     e->tx, e->ring);
  if (e->tcode) {
    const size_t nq = (size_t)e->B * (e->NR / 4);
    hipError_t st = hipMemsetAsync(e->told, 0, nq * 4, s);
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL(pack_codes_kernel, dim3(blocks(nq * e->NV, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey,
                       e->tcode, e->tage, e->tseq, e->told);
  }
  e->ring_valid = true;
  return hipGetLastError();
}

// The RICH output-tail description of one call: section layout of the state vector, and which of its columns
// belong to the observation launch that follows (posdist_kernel.hpp) and are left alone here.
RichParams rich_for(const DiralEnv* e, const StepParams& p) {
  RichParams r = e->rich;
  r.chobs_out = nullptr; r.episode = p.episode; r.eps = p.eps;
  r.plain_state = ((p.flags & (kRichFlags | DIRAL_F_ADD_POSDIST)) == 0 && has_hist2(p) && (p.flags & DIRAL_F_ADD_ACTION)) ? 1 : 0;
  if (!has_hist2(p)) r.off_hist = -1;                         // (type 1: posdist_kernel writes those columns)
  // the sorted-distance columns and the type-1 histogram are posdist_kernel.hpp's, every one of them (and
  // neighbours in the state vector, state_offsets): not written here at all
  const bool skip_full = (p.flags & DIRAL_F_ADD_POSDIST) && p.state_out && p.off_posdist >= 0 && p.N > 1;
  const bool skip_t1 = (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) && p.posdist_type == 1 && p.state_out && p.off_hist >= 0;
  // State.piggybacking: the A * A channel-observation section is piggy_emit_kernel's (piggyback_kernel.hpp); it sits
  // right in front of the other two (state_offsets), so the skipped columns stay one range
  const bool skip_pb = (e->cfg.flags & DIRAL_F_PIGGYBACKING) && p.state_out && e->off_chobs_pb >= 0;
  r.off_skip = skip_pb ? e->off_chobs_pb : (skip_full ? p.off_posdist : (skip_t1 ? p.off_hist : 0));
  r.len_skip = (skip_pb ? p.A * p.A : 0) + (skip_full ? p.N - 1 : 0) + (skip_t1 ? p.K : 0);
  r.pf = nullptr;
  return r;
}

// TestEnv.obtain_state as its own launch (observe_kernel.hpp): reads the tables as they are - young entries'
// xpos from the ring when it is valid, so no plane materialisation is needed - and changes nothing
hipError_t launch_observe_any(DiralEnv* e, const StepParams& p, hipStream_t s) {
  const bool use_ring = e->ring != nullptr && e->ring_valid;
  if (!use_ring) {
    const hipError_t st = ensure_plane(e, s);
    if (st != hipSuccess) return st;
  }
  ObserveParams o;
  o.N = p.N; o.A = p.A; o.K = p.K; o.NV = p.NV; o.NR = p.NR; o.flags = p.flags; o.age_limit = p.age_limit;
  o.want_hist = (has_hist2(p) && p.off_hist >= 0) ? 1 : 0;
  o.L = p.L; o.Rb = p.Rb; o.inv_w = p.hist_inv_width;
  o.actions = p.actions; o.chobs_in = p.chobs_in; o.rew_in = p.rew_in;
  o.pos_x = p.pos_x; o.pos_y = p.pos_y; o.vel = p.vel; o.tkey = p.tkey; o.tx = p.tx;
  o.ring = use_ring ? e->ring : nullptr;
  const bool packed = use_ring && e->tcode != nullptr;           // codes, ages, own sequence numbers
  o.tcode = packed ? e->tcode : nullptr; o.tage = packed ? e->tage : nullptr; o.tseq = packed ? e->tseq : nullptr;
  o.edges = p.edges; o.inv_tab = e->inv_tab; o.err = p.err; o.state_out = p.state_out;
  const RichParams r = rich_for(e, p);
  e->last_kernel = DIRAL_KERNEL_OBSERVE | (use_ring ? DIRAL_KERNEL_RING : 0);
  return launch_observe(o, r, e->flat_y, p.out_f64 != 0, p.B, s);
}

// after an import: the ring must answer every young entry with that entry's own xpos (aux_kernels.hpp)
hipError_t verify_ring(DiralEnv* e, hipStream_t s) {
  const size_t total = (size_t)e->B * e->N * e->N;
  hipLaunchKernelGGL(ring_verify_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey, e->tx,
                     e->ring, e->err);
  return hipGetLastError();
}

// scratch of the three-launch form (step_large.hpp), and its kernels' LDS attributes: at create for a large handle,
// at diral_env_set_option(DIRAL_OPT_KERNEL_PATH, DIRAL_PATH_LARGE) for any other
hipError_t alloc_large_scratch(DiralEnv* e) {
  if (e->lg.src) return hipSuccess;
  const size_t bn = (size_t)e->B * e->N, ba = (size_t)e->B * e->A;
  struct { void** p; size_t bytes; } bufs[] = {
    {(void**)&e->lg.src, ba * e->N * 2}, {(void**)&e->lg.cnt, ba * 4}, {(void**)&e->lg.alist, ba * 2},
    {(void**)&e->lg.nact, (size_t)e->B * 4}, {(void**)&e->lg.qflag, (size_t)e->B * ((e->N + 1) / 2) + 16}, {(void**)&e->lg.px0, bn * 8},
    {(void**)&e->lg.rew, bn * 8}, {(void**)&e->lg.rtx, bn * 8}};
  for (auto& q : bufs) {
    hipError_t r = hipMalloc(q.p, q.bytes);
    if (r != hipSuccess) return r;
    e->hbm_bytes += (int64_t)q.bytes;
    r = hipMemset(*q.p, 0, q.bytes);
    if (r != hipSuccess) return r;
  }
  return set_attr_large(e->N, e->A, e->K);
}

size_t slow_set_words(const DiralEnv* e) { return 16 + (size_t)fast_slow_max(e->B) + (size_t)e->B; }
hipError_t clear_slow_sets(DiralEnv* e, hipStream_t s) {
  if (!e->slow) return hipSuccess;
  e->slow_launches = 0;
  return hipMemsetAsync(e->slow, 0, 3 * slow_set_words(e) * 4, s);
}

// Which kernel family and instantiation a step call runs on: decided in ONE place, for the launch itself
// (launch_step_any) and for everyone who must know the outcome before anything is launched (diral_env_step_policy).
struct StepDispatch {
  bool spec, plain, ch, use_fast64, use_wide;
  // the run-time switches of the EXTRA instantiations, host-folded (FastParams::design ... nomove)
  bool design, prr, notab, nomove;
  int32_t* la;
  const double* trace;
  bool extra;
  bool pol_ok;          // the POL instantiation of step_fast64 (policy epilogue inside the launch) takes this call
  bool prefill_ok;      // ... and the K-slot form of it runs this configuration's random prefill (diral_env_prefill: my_step_design's
                        // reward is computed in P2 there, so the design switch does not count against it)
};
StepDispatch step_dispatch(const DiralEnv* e, const StepParams& p) {
  StepDispatch d;
  d.spec = is_specialised_cfg(p) && e->kernel_path == DIRAL_PATH_AUTO && !e->large;
  d.plain = d.spec && is_plain_cfg(p);
  d.ch = p.mode == DIRAL_STEP_MY_STEP_CH;
  // the wide kernels read the reward column of a RICH state back from rew_out
  const bool wide_rich_ok = d.plain || !(p.flags & DIRAL_F_ADD_REWARD) || p.state_out == nullptr || p.rew_out != nullptr;
  d.use_fast64 = d.spec && e->vpl == 1 && p.A <= kFastMaxA && p.NV == 64;
  d.use_wide = d.spec && e->vpl > 1 && p.A <= kWideMaxA && e->flat_y && wide_rich_ok;
  d.design = p.mode == DIRAL_STEP_DESIGN;
  d.prr = (p.flags & DIRAL_F_TRACK_PRR) && p.mode == DIRAL_STEP_MY_STEP;
  d.notab = !(p.flags & DIRAL_F_ADD_POSDIST_PIGGY);          // no piggybacked tables: test_env.py:138-139, 231-238
  d.nomove = !(p.flags & DIRAL_F_MOBILITY);                  // static (design) topology: network.py:302-305
  d.la = (p.flags & DIRAL_F_TRACK_ARRIVAL) ? p.la : nullptr;
  d.trace = d.nomove ? nullptr : p.trace;
  d.extra = d.design || d.la != nullptr || d.trace != nullptr || d.prr || d.notab || d.nomove;
  d.pol_ok = d.use_fast64 && e->flat_y && !d.ch && !d.extra && p.N >= 8;
  d.prefill_ok = d.use_fast64 && e->flat_y && d.design && d.la == nullptr && d.trace == nullptr && !d.prr && !d.notab && !d.nomove && p.N >= 8;
  return d;
}

// `pol` / `fused`: diral_env_step_policy - when the configuration runs on the POL instantiation of step_fast64 the policy
// epilogue is part of this launch and *fused is set; otherwise the plain step is launched and the caller adds the two
// policy launches
hipError_t launch_step_any(DiralEnv* e, const StepParams& p, hipStream_t s, const PolParams* pol = nullptr, bool* fused = nullptr) {
  const int vpl = e->vpl;
  const bool flat_y = e->flat_y;
  if (runs_large(e)) {
    // (and diral_env_observe: the search kernel's observe mode + the histogram launch)
    const hipError_t st = ensure_plane(e, s);
    if (st != hipSuccess) return st;
    if (p.mode != kModeObserve) e->ring_valid = false;
    e->last_kernel = DIRAL_KERNEL_LARGE;
    return launch_large(p, e->lg, s);
  }
  const StepDispatch d = step_dispatch(e, p);
  const bool plain = d.plain, ch = d.ch, use_fast64 = d.use_fast64, use_wide = d.use_wide;
  // xpos ring (step_fast64.hpp): the N <= 64 kernel keeps the plane only for entries older than the ring reaches
  const bool use_ring = (use_fast64 || use_wide) && e->ring != nullptr;
  if (use_ring) {
    const hipError_t st = ensure_ring(e, s);
    if (st != hipSuccess) return st;
  } else {
    const hipError_t st = ensure_plane(e, s);
    if (st != hipSuccess) return st;
    if (p.mode != kModeObserve) e->ring_valid = false;         // this step moves the tables without the ring
  }
  if (use_fast64 || use_wide) {
    FastParams f;
    f.N = p.N; f.A = p.A; f.K = p.K; f.NR = p.NR; f.NV = p.NV; f.flags = p.flags;
    f.reward_design = p.reward_design; f.age_limit = p.age_limit; f.episode_interval = p.episode_interval;
    f.design = d.design ? 1 : 0;
    f.done_now = (!p.t_dev && (p.t % p.episode_interval) == p.episode_interval - 1) ? 1 : 0;   // main_test.py:226 (with a slot clock: on the device)
    f.prr = d.prr ? 1 : 0;
    f.notab = d.notab ? 1 : 0;
    f.nomove = d.nomove ? 1 : 0;
    f.trace = d.trace;
    f.chobs_mode = (p.chobs_out ? 1 : 0) | ((p.mode == DIRAL_STEP_MY_STEP && p.state_type == 2) ? 2 : 0);
    f.L = p.L; f.Rc = p.Rc; f.Rb = p.Rb; f.inv_w = p.hist_inv_width; f.t = p.t; f.t_dev = p.t_dev;
    f.actions = p.actions; f.pos_x = p.pos_x; f.pos_y = p.pos_y; f.vel = p.vel; f.tkey = p.tkey; f.tx = p.tx;
    f.metrics = p.metrics; f.err = p.err; f.edges = p.edges; f.inv_tab = e->inv_tab;
    f.ring = use_ring ? e->ring : nullptr;
    f.tcode = e->tcode; f.tage = e->tage; f.tseq = e->tseq; f.told = e->told;
    if (use_ring) e->plane_valid = false;
    f.la = d.la;
    f.trace_len = p.trace_len; f.trace_per_env = p.trace_per_env;
    f.state_out = p.state_out; f.rew_out = p.rew_out;
    f.done_out = p.done_out; f.dbg = p.dbg;
    f.B = p.B;
    {
      // float32 screening of the histogram bin (step_fast64.hpp, fast quads): everything float32 can lose, in bin widths,
      // for positions in [0, L] - two conversions of a position, the subtraction of an in-range difference, the
      // constant and the fma at t <= K - with a factor of two on top; as 1 / 65536ths, rounded up, + 1
      const double w = (p.Rb - (-p.Rb)) / (double)p.K;
      const double xmax = p.L;
      // (the kernel computes t = fma(float(xpos), float(inv_w 2^16), float((Rb - npx) inv_w 2^16)): the conversion of the
      // stamp and of the factor lose 2 |xpos| 2^-24 / w bin widths, the addend's |Rb - npx| 2^-24 / w <= (xmax + Rb) 2^-24 / w,
      // the fma's own rounding K 2^-24)
      const double lost = 2.0 * (((3.0 * xmax + p.Rb) * 0x1p-24) / w + 2.0 * (double)p.K * 0x1p-23);
      const double m = std::ceil(lost * 65536.0) + 1.0;
      const bool on = use_fast64 && m <= 64.0 && xmax < 1e6 && e->f32_margin != 0;
      f.f32_m16 = on ? std::max((int)m, std::min(e->f32_margin, 16384)) : 0;
      f.f32_xmax = (float)xmax;
    }
    f.slow_cnt_r = nullptr; f.slow_list_r = nullptr; f.slow_flag_r = nullptr;
    f.slow_cnt_w = nullptr; f.slow_list_w = nullptr; f.slow_flag_w = nullptr; f.slow_cnt_z = nullptr;
    bool slow_first = false;
    // K slots per launch (diral_env_step_policy, DiralSlotPolicy::slots > 1): blocks = envs in order - over K slots a
    // straggler averages out; the slow-env sets stay as the last one-slot launch left them (complete or empty), unread
    const bool kslots = pol && ((d.pol_ok && pol->K > 1) || (pol->prefill && d.prefill_ok));
    // (step_wide: the packed form at N <= 128 only - its slow envs are 4 x the others; the plane form's are 1.6 x and measured
    // 4 % SLOWER dispatched first, N > 128 packed runs on dense topologies without any: - 0.7 % for the bookkeeping)
    // Round 6: with the far-entry guard (step_wide.hpp `wide_far_guard`) the flagged passes of a highway that broke apart run
    // on the coded path - those envs are no longer 3 x the others, and the slow-first grid (B / 4 more blocks, a flag load per
    // block) now costs more than it orders: C5 0.930 ms with it, 0.875 in batch order (one box, interleaved, profiles/r06).
    // DIRAL_WIDE_SLOW_FIRST=1 at create brings it back (A/B).
    const bool wide_slow = use_wide && vpl == 2 && e->tcode != nullptr && e->wide_slow_first;
    if ((use_fast64 || wide_slow) && e->slow && e->slow_first && !kslots) {
      const size_t w = slow_set_words(e);
      uint32_t* const set_r = e->slow + (e->slow_launches % 3) * w;
      uint32_t* const set_w = e->slow + ((e->slow_launches + 1) % 3) * w;
      uint32_t* const set_z = e->slow + ((e->slow_launches + 2) % 3) * w;
      f.slow_cnt_r = set_r; f.slow_list_r = set_r + 16; f.slow_flag_r = set_r + 16 + fast_slow_max(e->B);
      if (e->capture_rotates || !stream_is_capturing(s)) {
        f.slow_cnt_w = set_w; f.slow_list_w = set_w + 16; f.slow_flag_w = set_w + 16 + fast_slow_max(e->B);
        f.slow_cnt_z = set_z;
        ++e->slow_launches;
      }
      // (a CAPTURED launch reads the set it was baked with and builds none: a replayed graph could not rotate the sets -
      // unless the caller keeps every graph at a multiple of three launches and its phase aligned:
      // diral_env_set_capture_rotation.  Eager launches between two replays keep rotating through that set: they leave it
      // either rebuilt or EMPTY - count and flags, step_fast64.hpp - never half cleared, so the replay runs every env
      // exactly once whatever happened in between; the list it reads ages - slow envs stay slow for hundreds of slots,
      // profiles/launch_timeline.py - or is empty, which costs time, not correctness.)
      slow_first = true;
    }
    RichParams r = rich_for(e, p);
    r.chobs_out = p.chobs_out;
    r.pf = ((p.flags & DIRAL_F_PROPORTIONAL_FAIR) && p.mode == DIRAL_STEP_MY_STEP) ? p.pf : nullptr;
    r.pf_threshold = p.pf_threshold; r.pf_penalty = p.pf_penalty;
    KernelSel k;
    k.flat = flat_y; k.out64 = p.out_f64 != 0; k.full = p.N == 64 * vpl; k.ch = ch;
    k.extra = d.extra;                                          // EXTRA instantiation: the run-time switches compiled in
    k.rich = !plain;
    k.packed = use_wide && e->tcode != nullptr;
    e->last_kernel = (use_wide ? DIRAL_KERNEL_WIDE : DIRAL_KERNEL_FAST64) | (k.rich ? DIRAL_KERNEL_RICH : 0) |
                     ((k.packed || use_fast64) ? DIRAL_KERNEL_PACKED : 0) |
                     (k.extra ? DIRAL_KERNEL_EXTRA : 0) | (k.ch ? DIRAL_KERNEL_CH : 0) | (use_ring ? DIRAL_KERNEL_RING : 0);
    if (use_wide) {
      const int grid = p.B + (slow_first ? fast_slow_max(p.B) : 0);
      return vpl == 2 ? launch_wide2(f, r, k, grid, s) : launch_wide4(f, r, k, grid, s);
    }
    if (kslots) {
      // the env stays on the chip from slot to slot: step_fast64_slots_kernel
      if (!k.rich) { r.plain_state = 1; }
      e->last_kernel |= DIRAL_KERNEL_RICH | DIRAL_KERNEL_POLICY;
      if (fused) *fused = true;
      return launch_fast64_slots(f, r, *pol, k.out64, p.B, s);
    }
    const int grid = p.B + (slow_first ? fast_slow_max(p.B) : 0);
    if (pol && d.pol_ok) {
      // the policy epilogue: RICH instantiation (the channel observation is staged in LDS whether or not it is written out)
      if (!k.rich) { r.plain_state = 1; }
      e->last_kernel |= DIRAL_KERNEL_RICH | DIRAL_KERNEL_POLICY;
      if (fused) *fused = true;
      return launch_fast64_policy(f, r, *pol, k.out64, grid, s);
    }
    return launch_fast64(f, r, k, grid, s);
  }
  // the generic FAST instantiation of the general kernel: the plain configuration on sizes the
  // specialised kernels do not take (A > 64, vehicles off the y = 0 lane at N > 64): my_step,
  // f32 outputs, no arrival stamps, no trace replay
  const bool fast = is_specialised_cfg(p) && is_plain_cfg(p) && !p.out_f64 && p.mode == DIRAL_STEP_MY_STEP &&
                    !(p.flags & DIRAL_F_TRACK_ARRIVAL) && p.trace == nullptr;
  e->last_kernel = DIRAL_KERNEL_GENERAL;
  return launch_general(vpl, fast, p, e->lds_bytes, s);
}

// Secondary observation modes (a15/a16) run as their own launch after the step (posdist_kernel.hpp).
hipError_t launch_posdist_if_needed(DiralEnv* e, const StepParams& p, hipStream_t s) {
  const bool full = (p.flags & DIRAL_F_ADD_POSDIST) != 0;
  const bool type1 = (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) && p.posdist_type == 1;
  if (!p.state_out || !(full || type1)) return hipSuccess;
  PosdistParams q;
  q.N = p.N; q.A = p.A; q.K = p.K; q.S = p.S; q.NV = p.NV; q.NR = p.NR; q.flags = p.flags;
  q.posdist_type = p.posdist_type; q.age_limit = p.age_limit; q.out_f64 = p.out_f64;
  q.off_posdist = p.off_posdist; q.off_hist = p.off_hist;
  q.pos_x = p.pos_x; q.pos_y = p.pos_y; q.tkey = p.tkey; q.tx = p.tx; q.edges1 = e->edges1; q.state_out = p.state_out;
  q.do_full = 0; q.do_type1 = 0; q.ring = nullptr; q.tcode = nullptr; q.tage = nullptr; q.tseq = nullptr;
  q.flat_y = e->flat_y ? 1 : 0;
  const bool full_flat = full && e->flat_y;                     // one ranking of the env's x serves every viewer
  const bool type1_n64 = type1 && p.NV == 64 && p.K <= DIRAL_SMALL_MAX_BINS;   // lane = viewer, sort in registers
  const bool type1_lanes = type1 && !type1_n64 && p.N <= DIRAL_SMALL_MAX_USERS && p.K <= DIRAL_SMALL_MAX_BINS;   // 2 / 4 lanes per viewer (N <= 128 / 256)
  const bool type1_generic = type1 && !type1_n64 && !type1_lanes;   // beyond: posdist_kernel's literal statement
  if (full_flat) {
    q.do_full = 1;
    hipLaunchKernelGGL(posdist_sorted_flat_kernel, dim3(p.B), dim3(256), posdist_flat_lds_bytes(p.N), s, q);
    q.do_full = 0;
  }
  if (type1_n64) {
    q.do_type1 = 1;
    q.ring = (e->ring && !e->plane_valid) ? e->ring : nullptr;  // young entries straight from the ring: no materialise pass
    if (q.ring && e->tcode) { q.tcode = e->tcode; q.tage = e->tage; q.tseq = e->tseq; }   // ... and the words from the packed table
    hipLaunchKernelGGL(posdist_type1_n64_kernel, dim3(p.B), dim3(64), posdist_type1_lds_bytes(p.K), s, q);
    q.do_type1 = 0; q.ring = nullptr; q.tcode = nullptr; q.tage = nullptr; q.tseq = nullptr;
  }
  if (type1_lanes) {
    const bool wide_lanes = e->type1_wide_lanes;               // (read once at create: never an env lookup on the step path)
    const int npad = p.N <= 128 ? 128 : 256;
    const int lpv = npad / (wide_lanes ? 64 : 32), vw = 64 / lpv, nvb = (p.N + vw - 1) / vw;
    q.do_type1 = 1;
    q.ring = (e->ring && !e->plane_valid) ? e->ring : nullptr;
    if (q.ring && e->tcode) { q.tcode = e->tcode; q.tage = e->tage; q.tseq = e->tseq; }
    const uint32_t lds = posdist_type1_lanes_lds_bytes(p.K, npad, lpv);
    const dim3 grid((unsigned)p.B * nvb);
    if (wide_lanes && lpv == 2) hipLaunchKernelGGL((posdist_type1_lanes_kernel<2, 64>), grid, dim3(64), lds, s, q);
    else if (wide_lanes) hipLaunchKernelGGL((posdist_type1_lanes_kernel<4, 64>), grid, dim3(64), lds, s, q);
    else if (lpv == 4) hipLaunchKernelGGL((posdist_type1_lanes_kernel<4, 32>), grid, dim3(64), lds, s, q);
    else hipLaunchKernelGGL((posdist_type1_lanes_kernel<8, 32>), grid, dim3(64), lds, s, q);
    q.do_type1 = 0; q.ring = nullptr; q.tcode = nullptr; q.tage = nullptr; q.tseq = nullptr;
  }
  q.do_full = (full && !full_flat) ? 1 : 0;
  q.do_type1 = type1_generic ? 1 : 0;
  if (q.do_full || q.do_type1) {
    if (q.do_type1) {                                           // (the sorted true distances read no table)
      const hipError_t st = ensure_plane(e, s);
      if (st != hipSuccess) return st;
    }
    if (posdist_lds_bytes(p.N, p.K) > 160u * 1024u) return hipErrorInvalidValue;   // (off-lane vehicles at N >~ 1400: one env's values do not fit)
    hipLaunchKernelGGL(posdist_kernel, dim3(p.B), dim3(64 * kPdWaves), posdist_lds_bytes(p.N, p.K), s, q);
  }
  return hipGetLastError();
}

PiggyParams piggy_params(const DiralEnv* e, const StepParams& p) {
  PiggyParams q;
  q.N = e->N; q.A = e->A; q.S = e->S; q.off_chobs = e->off_chobs_pb; q.out_f64 = p.out_f64; q.Rc = p.Rc;
  q.actions = p.actions; q.pos_x = e->pos_x; q.pos_y = e->pos_y;
  q.obs_new = e->obs_new; q.txid = e->txid; q.prev_obs = e->prev_obs;
  q.chobs_out = p.chobs_out; q.state_out = p.state_out; q.chobs_in = p.chobs_in; q.err = e->err;
  return q;
}

// Recompute DiralEnv::flat_y (all pos_y == 0) after pos_y was written by the
// caller.  Synchronises the stream; only reset/import call it, never step.
int refresh_flat_y(DiralEnv* e, hipStream_t s) {
  const size_t bn = (size_t)e->B * e->N;
  if (hipMemsetAsync(e->yflag, 0, 4, s) != hipSuccess) return DIRAL_ERR_HIP;
  hipLaunchKernelGGL(any_nonzero_kernel, dim3(blocks(bn, 256)), dim3(256), 0, s, (int)bn, e->pos_y, e->yflag);
  uint32_t f = 1;
  if (hipMemcpyAsync(&f, e->yflag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return DIRAL_ERR_HIP;
  if (hipStreamSynchronize(s) != hipSuccess) return DIRAL_ERR_HIP;
  e->flat_y = (f == 0);
  return DIRAL_OK;
}

}  // namespace

extern "C" {

int diral_env_abi_version(void) { return DIRAL_ABI_VERSION; }

const char* diral_env_strerror(int status) {
  switch (status) {
    case DIRAL_OK: return "ok";
    case DIRAL_ERR_BAD_ARG: return "bad argument";
    case DIRAL_ERR_BAD_CONFIG: return "config rejected (the reference cannot run it either, or undefined behaviour there)";
    case DIRAL_ERR_UNSUPPORTED: return "valid reference config outside this build's limits";
    case DIRAL_ERR_HIP: return "HIP runtime error (see diral_env_last_hip_error)";
    case DIRAL_ERR_NO_DEVICE: return "no usable HIP device";
    case DIRAL_ERR_ACTION_RANGE: return "action outside [0, num_channels)";
    case DIRAL_ERR_SEQ_OVERFLOW: return "more than DIRAL_MAX_SLOTS steps since reset";
    case DIRAL_ERR_CAPTURE: return "call needs a ring <-> plane conversion and the stream is being captured into a hipGraph";
    case DIRAL_ERR_TABLE_CONFLICT: return "imported tables hold entries about one subject with equal sequence numbers but different xpos";
    case DIRAL_ERR_PIGGY_NO_TX: return "State.piggybacking: a receiver heard no transmitter on a used resource (the reference's prev_obs[None] KeyError)";
    default: return "unknown status";
  }
}

void diral_cfg_defaults(DiralCfg* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->struct_bytes = sizeof(DiralCfg);
  c->flags = 0;
  c->num_users = 3;            // test_env.py:12
  c->num_channels = 3;         // test_env.py:13
  c->num_bins = 20;
  c->reward_design = 1;        // test_env.py:20
  c->state_type = 2;
  c->posdist_type = 2;
  c->episode_interval = 25;    // main_test.py:22
  c->info_age_limit = 20;      // network.py:547
  c->pf_threshold = 10;        // test_env.py:89
  c->pf_penalty = -10.0;       // test_env.py:90
  c->highway_length = 200;     // test_env.py:18
  c->highway_height = 2;       // network.py:31
  c->communication_range = 1;  // test_env.py:21
  c->bin_range = 500;          // test_env.py:24
}

int diral_env_validate(const DiralCfg* c) {
  if (!c || c->struct_bytes != sizeof(DiralCfg)) return DIRAL_ERR_BAD_ARG;
  if (c->num_users < 1 || c->num_channels < 1) return DIRAL_ERR_BAD_CONFIG;
  if (c->reward_design < 1 || c->reward_design > 5) return DIRAL_ERR_BAD_CONFIG;   // test_env.py:198-199
  if (c->state_type != 1 && c->state_type != 2) return DIRAL_ERR_BAD_CONFIG;
  if (!has(c, DIRAL_F_MOBILITY) && !has(c, DIRAL_F_DESIGN_TOPOLOGY)) return DIRAL_ERR_BAD_CONFIG;  // network.py:54-60
  if (c->episode_interval < 1 || c->info_age_limit < 0 || c->info_age_limit > 255) return DIRAL_ERR_BAD_CONFIG;
  if (!(c->highway_length > 0) || !(c->highway_height > 0)) return DIRAL_ERR_BAD_CONFIG;
  if (has(c, DIRAL_F_ADD_POSDIST_PIGGY)) {
    if (c->posdist_type != 1 && c->posdist_type != 2) return DIRAL_ERR_BAD_CONFIG;  // test_env.py:555-560
    if (c->num_bins < 1 || !(c->bin_range > 0)) return DIRAL_ERR_BAD_CONFIG;
    if (c->num_bins > DIRAL_MAX_BINS) return DIRAL_ERR_UNSUPPORTED;
  }
  if (has(c, DIRAL_F_PIGGYBACKING)) {
    // type 1 inserts on idle resources only (test_env.py:226-232, 250-254): ragged observations; without the
    // channel-observation section obtain_state never reads what get_state_space() counts (test_env.py:71-72, 539-541)
    if (c->state_type != 2 || !has(c, DIRAL_F_ADD_CHANNEL_OBS)) return DIRAL_ERR_BAD_CONFIG;
  }
  if (c->num_users > DIRAL_MAX_USERS || c->num_channels > DIRAL_MAX_CHANNELS) return DIRAL_ERR_UNSUPPORTED;
  const int kbins = c->num_bins > 0 ? c->num_bins : 1;
  if (is_large_cfg(c)) {
    // step_large.hpp: the env's vehicles and resource lists in one workgroup's LDS, a table column in a wave's
    if (large_lds_bytes(c->num_users, c->num_channels, kbins) > 160u * 1024u) return DIRAL_ERR_UNSUPPORTED;
    // State.piggybacking: A * A values per agent, indexed with 32 bits in piggyback_kernel.hpp
    if (has(c, DIRAL_F_PIGGYBACKING) && c->num_channels > DIRAL_SMALL_MAX_CHANNELS) return DIRAL_ERR_UNSUPPORTED;
    // the type-1 histogram at these sizes is posdist_kernel's literal statement: one env's values in LDS (N <~ 1400)
    if (has(c, DIRAL_F_ADD_POSDIST_PIGGY) && c->posdist_type == 1 && posdist_lds_bytes(c->num_users, kbins) > 160u * 1024u)
      return DIRAL_ERR_UNSUPPORTED;
    if (has(c, DIRAL_F_ADD_POSDIST) && posdist_flat_lds_bytes(c->num_users) > 160u * 1024u) return DIRAL_ERR_UNSUPPORTED;
    return DIRAL_OK;
  }
  const int vpl = vpl_for(c->num_users);
  const LdsLayout l = lds_layout(64 * vpl, c->num_channels, kbins, vpl, 4 * vpl);
  if (l.total > 160u * 1024u) return DIRAL_ERR_UNSUPPORTED;
  return DIRAL_OK;
}

int diral_env_state_space(const DiralCfg* c) {
  if (!c || c->struct_bytes != sizeof(DiralCfg)) return DIRAL_ERR_BAD_ARG;
  return state_offsets(c).S;
}

int diral_env_create(const DiralCfg* cfg, int batch, int device, DiralEnv** out) {
  if (!out) return DIRAL_ERR_BAD_ARG;
  *out = nullptr;
  int st = diral_env_validate(cfg);
  if (st != DIRAL_OK) return st;
  if (batch < 1) return DIRAL_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return DIRAL_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return DIRAL_ERR_NO_DEVICE;

  DiralEnv* e = new DiralEnv();
  e->cfg = *cfg;
  e->B = batch; e->N = cfg->num_users; e->A = cfg->num_channels;
  e->K = cfg->num_bins > 0 ? cfg->num_bins : 1;
  e->NV = e->N <= 64 ? 64 : (int)align_up((uint32_t)e->N, 16);   // one wave lane per viewer, no lane predicate
  e->NR = (int)align_up((uint32_t)e->N, 16);   // every wave owns 16 existing subject rows
  e->vpl = vpl_for(e->N);
  e->large = is_large_cfg(cfg);
  e->device = device;
  const Offsets off = state_offsets(cfg);
  e->S = off.S;

  auto fail = [&](int code) { diral_env_destroy(e); return code; };
  DeviceGuard guard(device);
  if (!guard.ok) return fail(DIRAL_ERR_NO_DEVICE);

  const size_t bn = (size_t)e->B * e->N;
  const size_t tab = (size_t)e->B * e->NR * e->NV;
  auto alloc = [&](void** p, size_t bytes) {
    hipError_t r = hipMalloc(p, bytes);
    if (r == hipSuccess) e->hbm_bytes += (int64_t)bytes;
    return r;
  };
#define CREATE_TRY(call) do { hipError_t r__ = (call); if (r__ != hipSuccess) { note_hip(e, r__, #call); \
      std::fprintf(stderr, "diral_env_create: %s\n", e->last_hip_error.c_str()); return fail(DIRAL_ERR_HIP); } } while (0)
  CREATE_TRY(alloc((void**)&e->pos_x, bn * 8));
  CREATE_TRY(alloc((void**)&e->pos_y, bn * 8));
  CREATE_TRY(alloc((void**)&e->vel, bn * 8));
  // + 256 elements of slack: step_wide.hpp loads a padded viewer slot (lane + 64 j) past the
  // end of a row without clamping (the values are masked, never stored)
  // + 64 rows: the type-1 kernels (posdist_kernel.hpp) read the rows up to the next multiple of 64 of their env
  // unmasked as well
  CREATE_TRY(alloc((void**)&e->tkey, (tab + 512 + 64 * 256) * 4));
  CREATE_TRY(alloc((void**)&e->tx, (tab + 512 + 64 * 256) * 8));
  // the xpos ring of the specialised kernels
  if (!e->large && ((e->vpl == 1 && e->NV == 64 && e->A <= kFastMaxA) || (e->vpl > 1 && e->A <= kWideMaxA))) {
    CREATE_TRY(alloc((void**)&e->ring, (size_t)e->B * e->NR * 8 * 8));
    CREATE_TRY(hipMemset(e->ring, 0, (size_t)e->B * e->NR * 8 * 8));
    if (use_packed_table(e)) {                                  // the packed table of step_fast64 and of step_wide at N > 128
      // (+ 512 words of slack: step_wide.hpp loads a padded viewer slot past the end of a row without clamping)
      const size_t nq = (size_t)e->B * (e->NR / 4);
      CREATE_TRY(alloc((void**)&e->tcode, (nq * e->NV + 512) * 4));
      CREATE_TRY(alloc((void**)&e->tage, (nq * e->NV + 512) * 4));
      CREATE_TRY(alloc((void**)&e->tseq, ((size_t)e->B * e->NR + 64) * 4));
      CREATE_TRY(alloc((void**)&e->told, (nq + 16) * 4));
      CREATE_TRY(hipMemset(e->tcode, 0, (nq * e->NV + 512) * 4));
      CREATE_TRY(hipMemset(e->tage, 0, (nq * e->NV + 512) * 4));
      CREATE_TRY(hipMemset(e->tseq, 0, ((size_t)e->B * e->NR + 64) * 4));
      CREATE_TRY(hipMemset(e->told, 0, (nq + 16) * 4));
    }
    {
      CREATE_TRY(alloc((void**)&e->slow, 3 * slow_set_words(e) * 4));
      CREATE_TRY(hipMemset(e->slow, 0, 3 * slow_set_words(e) * 4));
      const char* off = std::getenv("DIRAL_NO_SLOW_FIRST");
      e->slow_first = !(off && off[0] == '1');
    }
    if (e->vpl == 1) {
      if (const char* fm = std::getenv("DIRAL_F32_MARGIN")) e->f32_margin = std::max(0, std::atoi(fm));
    }
    e->ring_valid = true;                                       // all tables zero: never heard, age 0, xpos 0
  }
  CREATE_TRY(alloc((void**)&e->metrics, (size_t)e->B * DIRAL_M_COLUMNS * 8));
  CREATE_TRY(alloc((void**)&e->err, 4));
  CREATE_TRY(alloc((void**)&e->yflag, 4));
  CREATE_TRY(alloc((void**)&e->edges, (size_t)(e->K + 1) * 8));
  CREATE_TRY(alloc((void**)&e->edges1, (size_t)(e->K + 1) * 8));
  CREATE_TRY(alloc((void**)&e->inv_tab, 256 * 8));
  if (has(cfg, DIRAL_F_TRACK_ARRIVAL)) CREATE_TRY(alloc((void**)&e->la, bn * e->N * 4));
  if (has(cfg, DIRAL_F_PROPORTIONAL_FAIR)) CREATE_TRY(alloc((void**)&e->pf, bn * 4));
  if (has(cfg, DIRAL_F_PIGGYBACKING)) {
    if (piggy_search_lds_bytes(e->N) > 48u * 1024u)
      CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(piggy_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)piggy_search_lds_bytes(e->N)));
    CREATE_TRY(alloc((void**)&e->prev_obs, bn * e->A * 8));
    CREATE_TRY(alloc((void**)&e->obs_new, bn * e->A * 8));
    CREATE_TRY(alloc((void**)&e->txid, bn * e->A * 4));
    CREATE_TRY(hipMemset(e->prev_obs, 0, bn * e->A * 8));       // test_env.py:76-79
  }

  std::vector<double> edges;
  np_linspace(-cfg->bin_range, cfg->bin_range, e->K + 1, edges);
  for (int i = 0; i < e->K; ++i)   // np.histogram: "Too many bins for data range"
    if (!(edges[i] < edges[i + 1])) return fail(DIRAL_ERR_BAD_CONFIG);
  CREATE_TRY(hipMemcpy(e->edges, edges.data(), edges.size() * 8, hipMemcpyHostToDevice));
  np_linspace(-1.0, 1.0, e->K + 1, edges);
  CREATE_TRY(hipMemcpy(e->edges1, edges.data(), edges.size() * 8, hipMemcpyHostToDevice));
  {
    std::vector<double> inv(256, 0.0);
    for (int n = 1; n < 256; ++n) { volatile double d = (double)n; inv[n] = 1.0 / d; }
    CREATE_TRY(hipMemcpy(e->inv_tab, inv.data(), 256 * 8, hipMemcpyHostToDevice));
  }
  if (posdist_lds_bytes(e->N, e->K) <= 160u * 1024u)
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(posdist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)posdist_lds_bytes(e->N, e->K)));
  if (posdist_flat_lds_bytes(e->N) > 48u * 1024u && posdist_flat_lds_bytes(e->N) <= 160u * 1024u)
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(posdist_sorted_flat_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)posdist_flat_lds_bytes(e->N)));
  if (!e->large)
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(posdist_type1_n64_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)posdist_type1_lds_bytes(e->K)));
  CREATE_TRY(hipMemset(e->pos_x, 0, bn * 8));
  CREATE_TRY(hipMemset(e->pos_y, 0, bn * 8));
  CREATE_TRY(hipMemset(e->vel, 0, bn * 8));
  CREATE_TRY(hipMemset(e->tkey, 0, (tab + 512 + 64 * 256) * 4));
  CREATE_TRY(hipMemset(e->tx, 0, (tab + 512 + 64 * 256) * 8));
  CREATE_TRY(hipMemset(e->metrics, 0, (size_t)e->B * DIRAL_M_COLUMNS * 8));
  CREATE_TRY(hipMemset(e->err, 0, 4));
  if (e->la) CREATE_TRY(hipMemset(e->la, 0xFF, bn * e->N * 4));
  if (e->pf) CREATE_TRY(hipMemset(e->pf, 0, bn * 4));

  if (e->large) {
    CREATE_TRY(alloc_large_scratch(e));
  } else {
    const LdsLayout l = lds_layout(64 * e->vpl, e->A, e->K, e->vpl, 4 * e->vpl);
    e->lds_bytes = l.total;
    CREATE_TRY(set_attr_general(e->vpl, l.total));
    if (e->vpl == 2 && e->A <= kWideMaxA) CREATE_TRY(set_attr_wide2(e->A, e->K));
    if (e->vpl == 4 && e->A <= kWideMaxA) CREATE_TRY(set_attr_wide4(e->A, e->K));
    CREATE_TRY(set_attr_observe(e->N, e->K));
  }
#undef CREATE_TRY

  StepParams& p = e->base;
  std::memset(&p, 0, sizeof(p));
  p.B = e->B; p.N = e->N; p.A = e->A; p.K = e->K; p.S = e->S; p.NV = e->NV; p.NR = e->NR;
  p.flags = cfg->flags;
  p.reward_design = cfg->reward_design; p.state_type = cfg->state_type; p.posdist_type = cfg->posdist_type;
  p.age_limit = cfg->info_age_limit; p.pf_threshold = cfg->pf_threshold; p.pf_penalty = cfg->pf_penalty;
  p.L = cfg->highway_length; p.H = cfg->highway_height; p.Rc = cfg->communication_range;
  p.Rb = cfg->bin_range;
  p.hist_inv_width = (double)e->K / (cfg->bin_range - (-cfg->bin_range));
  p.episode_interval = cfg->episode_interval;
  p.off_act = off.act; p.off_chobs = off.chobs; p.off_posdist = off.posdist; p.off_hist = off.hist;
  if (has(cfg, DIRAL_F_PIGGYBACKING)) {
    // the step / observe kernels build a state vector WITHOUT the channel-observation section and leave its A * A
    // columns alone (rich_for); piggyback_kernel.hpp fills them
    e->off_chobs_pb = off.chobs;
    p.off_chobs = -1;
    p.flags &= ~(uint32_t)DIRAL_F_ADD_CHANNEL_OBS;
  }
  p.off_rew = off.rew; p.off_idx = off.idx; p.off_pos = off.pos; p.off_vel = off.vel; p.off_fp = off.fp;
  p.pos_x = e->pos_x; p.pos_y = e->pos_y; p.vel = e->vel; p.tkey = e->tkey; p.tx = e->tx;
#if defined(DIRAL_TIMING) || defined(DIRAL_DEBUG_XG)
  if (hipMalloc((void**)&e->dbg, (size_t)e->B * 4096 * 8) == hipSuccess) (void)hipMemset(e->dbg, 0, (size_t)e->B * 4096 * 8);
#endif
  p.dbg = e->dbg;
  p.la = e->la; p.pf = e->pf; p.metrics = e->metrics; p.err = e->err; p.edges = e->edges;
  RichParams& r = e->rich;
  std::memset(&r, 0, sizeof(r));
  r.S = e->S; r.state_type = cfg->state_type;
  r.off_act = off.act; r.off_chobs = has(cfg, DIRAL_F_PIGGYBACKING) ? -1 : off.chobs; r.off_hist = off.hist; r.off_rew = off.rew; r.off_idx = off.idx;
  r.off_skip = 0; r.len_skip = 0;
  r.off_pos = off.pos; r.off_vel = off.vel; r.off_fp = off.fp;
  r.H = cfg->highway_height; r.vel = e->vel; r.pos_y = e->pos_y;
  // test hooks, read ONCE here (never on the step path): force the general kernel
  if (const char* ws = std::getenv("DIRAL_WIDE_SLOW_FIRST")) e->wide_slow_first = ws[0] == '1';
  if (const char* tl = std::getenv("DIRAL_TYPE1_LANES")) e->type1_wide_lanes = std::strcmp(tl, "wide") == 0;
  if (std::getenv("DIRAL_NO_FAST64") && e->vpl == 1) e->kernel_path = DIRAL_PATH_GENERAL;
  if (std::getenv("DIRAL_NO_WIDE") && e->vpl > 1 && !e->large) e->kernel_path = DIRAL_PATH_GENERAL;
  *out = e;
  return DIRAL_OK;
}

int diral_env_destroy(DiralEnv* e) {
  if (!e) return DIRAL_OK;
  DeviceGuard guard(e->device);
  void* ptrs[] = {e->pos_x, e->pos_y, e->vel, e->tkey, e->tx, e->ring, e->tcode, e->tage, e->tseq, e->told, e->slow, e->la, e->pf, e->metrics, e->err, e->edges, e->edges1, e->inv_tab, e->trace, e->yflag,
                  e->prev_obs, e->obs_new, e->txid, e->dbg, e->lg.src, e->lg.cnt, e->lg.alist, e->lg.nact, e->lg.qflag, e->lg.px0, e->lg.rew, e->lg.rtx};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  delete e;
  return DIRAL_OK;
}

int64_t diral_env_hbm_bytes(const DiralEnv* e) { return e ? e->hbm_bytes : 0; }

int diral_env_set_option(DiralEnv* e, int option, int64_t value) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  switch (option) {
    case DIRAL_OPT_ENV_OFFSET:
      if (value < 0) return DIRAL_ERR_BAD_ARG;
      e->env_offset = value;
      return DIRAL_OK;
    case DIRAL_OPT_KERNEL_PATH:
      if (value != DIRAL_PATH_AUTO && value != DIRAL_PATH_GENERAL && value != DIRAL_PATH_LARGE) return DIRAL_ERR_BAD_ARG;
      if (e->large) return value == DIRAL_PATH_GENERAL ? DIRAL_ERR_UNSUPPORTED : DIRAL_OK;   // (there is one path for such a handle)
      if (value == DIRAL_PATH_LARGE) {
        if (large_lds_bytes(e->N, e->A, e->K) > 160u * 1024u) return DIRAL_ERR_UNSUPPORTED;
        DeviceGuard guard(e->device);
        if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
        HIP_TRY(e, alloc_large_scratch(e));
      }
      e->kernel_path = (int)value;
      return DIRAL_OK;
    default:
      return DIRAL_ERR_BAD_ARG;
  }
}

int diral_env_last_kernel(const DiralEnv* e) { return e ? e->last_kernel : DIRAL_ERR_BAD_ARG; }

const char* diral_env_last_hip_error(const DiralEnv* e) { return e ? e->last_hip_error.c_str() : ""; }

int diral_env_reset(DiralEnv* e, const double* x0, const double* y0, const double* v0, uint64_t seed,
                    void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t bn = (size_t)e->B * e->N;
  const size_t tab = (size_t)e->B * e->NR * e->NV;
  HIP_TRY(e, hipMemsetAsync(e->tkey, 0, tab * 4, s));
  HIP_TRY(e, hipMemsetAsync(e->tx, 0, tab * 8, s));
  if (e->ring) HIP_TRY(e, hipMemsetAsync(e->ring, 0, (size_t)e->B * e->NR * 8 * 8, s));
  if (e->tcode) {
    const size_t nq = (size_t)e->B * (e->NR / 4);
    HIP_TRY(e, hipMemsetAsync(e->tcode, 0, nq * e->NV * 4, s));
    HIP_TRY(e, hipMemsetAsync(e->tage, 0, nq * e->NV * 4, s));
    HIP_TRY(e, hipMemsetAsync(e->tseq, 0, (size_t)e->B * e->NR * 4, s));
    HIP_TRY(e, hipMemsetAsync(e->told, 0, nq * 4, s));
  }
  HIP_TRY(e, clear_slow_sets(e, s));
  e->plane_valid = true; e->ring_valid = e->ring != nullptr;
  HIP_TRY(e, hipMemsetAsync(e->metrics, 0, (size_t)e->B * DIRAL_M_COLUMNS * 8, s));
  if (e->la) HIP_TRY(e, hipMemsetAsync(e->la, 0xFF, bn * e->N * 4, s));
  if (e->pf) HIP_TRY(e, hipMemsetAsync(e->pf, 0, bn * 4, s));
  if (e->prev_obs) HIP_TRY(e, hipMemsetAsync(e->prev_obs, 0, bn * e->A * 8, s));
  hipLaunchKernelGGL(reset_kernel, dim3(blocks(bn, 256)), dim3(256), 0, s, (int)bn, e->cfg.highway_length,
                     has(&e->cfg, DIRAL_F_MOBILITY_VARY) ? 1 : 0, seed, (uint64_t)e->env_offset * (uint64_t)e->N, x0, y0, v0,
                     e->pos_x, e->pos_y, e->vel);
  HIP_TRY(e, hipGetLastError());
  if (y0) { if (refresh_flat_y(e, s) != DIRAL_OK) return DIRAL_ERR_HIP; }
  else e->flat_y = true;
  return DIRAL_OK;
}

int diral_env_step(DiralEnv* e, int mode, const int32_t* actions, int64_t t, void* state_out, void* rew_out,
                   uint8_t* done_out, void* chobs_out, int out_dtype, double episode, double epsilon,
                   void* stream) {
  if (!e || !actions) return DIRAL_ERR_BAD_ARG;
  if (mode != DIRAL_STEP_MY_STEP && mode != DIRAL_STEP_MY_STEP_CH && mode != DIRAL_STEP_DESIGN)
    return DIRAL_ERR_BAD_ARG;
  if (out_dtype != DIRAL_F32 && out_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  // my_step_ch defines rewards only for reward_design 2,3,4 (test_env.py:413-429)
  if (mode == DIRAL_STEP_MY_STEP_CH && (e->cfg.reward_design < 2 || e->cfg.reward_design > 4))
    return DIRAL_ERR_BAD_CONFIG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  StepParams p = e->base;
  p.mode = mode; p.t = t; p.episode = episode; p.eps = epsilon; p.out_f64 = (out_dtype == DIRAL_F64);
  p.actions = actions;
  p.state_out = e->S > 0 ? state_out : nullptr;
  p.rew_out = rew_out; p.done_out = done_out; p.chobs_out = chobs_out;
  p.chobs_in = nullptr; p.rew_in = nullptr;
  if (e->prev_obs) {
    // State.piggybacking: my_step_ch / my_step_design hand obtain_state the plain A-wide observation (test_env.py:316,
    // 443) - a state vector shorter than get_state_space()
    if (mode != DIRAL_STEP_MY_STEP) return DIRAL_ERR_BAD_CONFIG;
    PiggyParams q = piggy_params(e, p);
    p.chobs_out = nullptr;                                       // (the A * A observation is piggy_emit_kernel's)
    hipLaunchKernelGGL(piggy_search_kernel, dim3(e->B), dim3(256), piggy_search_lds_bytes(e->N), (hipStream_t)stream, q);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, launch_step_any(e, p, (hipStream_t)stream));
    HIP_TRY(e, launch_posdist_if_needed(e, p, (hipStream_t)stream));
    hipLaunchKernelGGL(piggy_emit_kernel, dim3(e->B), dim3(256), 0, (hipStream_t)stream, q);
    HIP_TRY(e, hipGetLastError());
    return DIRAL_OK;
  }
  HIP_TRY(e, launch_step_any(e, p, (hipStream_t)stream));
  HIP_TRY(e, launch_posdist_if_needed(e, p, (hipStream_t)stream));
  return DIRAL_OK;
}

static int sps_step_chobs_impl(int agents, int num_channels, const void* chobs, int chobs_dtype, const int32_t* actions,
                               int32_t* prev_action, int32_t* counter, double rssi_threshold, double inc_db,
                               double keep_prob, const int32_t* draw_counter, const double* draw_keep,
                               const int32_t* draw_choice, uint64_t seed, const long long* clock, int32_t* actions_out,
                               void* stream);

int diral_env_step_policy(DiralEnv* e, int mode, const int32_t* actions, int64_t t, void* state_out, void* rew_out,
                          uint8_t* done_out, void* chobs_out, int out_dtype, const DiralSlotPolicy* pol, void* stream) {
  if (!e || !actions || !pol || pol->struct_bytes != sizeof(DiralSlotPolicy)) return DIRAL_ERR_BAD_ARG;
  if (!pol->sps_prev_action || !pol->sps_counter || !pol->actions_out) return DIRAL_ERR_BAD_ARG;
  if (mode != DIRAL_STEP_MY_STEP && mode != DIRAL_STEP_MY_STEP_CH) return DIRAL_ERR_BAD_ARG;
  if (out_dtype != DIRAL_F32 && out_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  if (mode == DIRAL_STEP_MY_STEP_CH && (e->cfg.reward_design < 2 || e->cfg.reward_design > 4)) return DIRAL_ERR_BAD_CONFIG;
  if (pol->shaped_out && !rew_out) return DIRAL_ERR_BAD_ARG;                 // (the three-launch form shapes rew_out)
  if ((pol->shape_flags & ~5) != 0) return DIRAL_ERR_BAD_ARG;                // global_reward_avg | stuck-action penalty
  if (pol->shaped_out && (pol->shape_flags & 4) && (!pol->pen_counter || !pol->pen_prev_actions)) return DIRAL_ERR_BAD_ARG;
  if (e->A > kSpsWaveMaxA) return DIRAL_ERR_UNSUPPORTED;
  if (e->prev_obs) return DIRAL_ERR_UNSUPPORTED;                // State.piggybacking: the SPS agents sense A values, not A * A
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  StepParams p = e->base;
  p.mode = mode; p.t = t; p.episode = 0.0; p.eps = 1.0; p.out_f64 = (out_dtype == DIRAL_F64);
  p.actions = actions;
  p.state_out = e->S > 0 ? state_out : nullptr;
  p.rew_out = rew_out; p.done_out = done_out; p.chobs_out = chobs_out;
  p.chobs_in = nullptr; p.rew_in = nullptr;
  PolParams q;
  q.shape_flags = pol->shape_flags; q.pen_threshold = pol->pen_threshold; q.pen_value = pol->pen_value;
  q.shaped_out = pol->shaped_out; q.sum_r_out = pol->sum_r_out; q.coll_out = pol->collision_out;
  q.pen_counter = pol->pen_counter; q.pen_prev = pol->pen_prev_actions;
  q.sps_prev = pol->sps_prev_action; q.sps_counter = pol->sps_counter;
  q.threshold = pol->rssi_threshold; q.inc_db = pol->inc_db; q.keep_prob = pol->keep_prob;
  q.draw_counter = pol->draw_counter; q.draw_keep = pol->draw_keep; q.draw_choice = pol->draw_choice;
  q.seed = pol->seed; q.clock = (const long long*)pol->seed_clock; q.actions_out = pol->actions_out;
  const int slots = pol->slots > 1 ? pol->slots : 1;
  q.K = slots; q.vel_vary = has(&e->cfg, DIRAL_F_MOBILITY_VARY) ? 1 : 0; q.vel_seed = pol->vel_seed;
  q.idx0 = (uint64_t)e->env_offset * (uint64_t)e->N; q.vel_w = e->vel;
  q.prefill = 0; q.actions_all = nullptr; q.rew_in = nullptr;
  if (slots > 1 && (pol->draw_counter || pol->draw_keep || pol->draw_choice)) return DIRAL_ERR_BAD_ARG;
  // (decided before anything is launched: a caller without a channel-observation buffer can retry with one)
  const bool will_fuse = step_dispatch(e, p).pol_ok;
  if (!will_fuse && (!chobs_out || slots > 1)) return DIRAL_ERR_UNSUPPORTED;
  bool fused = false;
  HIP_TRY(e, launch_step_any(e, p, (hipStream_t)stream, will_fuse ? &q : nullptr, &fused));
  HIP_TRY(e, launch_posdist_if_needed(e, p, (hipStream_t)stream));
  if (fused) return DIRAL_OK;
  // the same slot as three launches (configurations the POL instantiation does not take)
  if (!chobs_out) return DIRAL_ERR_HIP;                                      // (cannot happen: one step_dispatch decides both)
  if (pol->shaped_out) {
    const int st = diral_driver_shape(e->B, e->N, e->A, rew_out, out_dtype, actions, nullptr, nullptr, pol->pen_counter,
                                      pol->pen_prev_actions, pol->shape_flags, pol->pen_threshold, pol->pen_value,
                                      pol->shaped_out, pol->sum_r_out, pol->collision_out, nullptr, nullptr, stream);
    if (st != DIRAL_OK) return st;
  }
  return sps_step_chobs_impl(e->B * e->N, e->A, chobs_out, out_dtype, actions, pol->sps_prev_action, pol->sps_counter,
                             pol->rssi_threshold, pol->inc_db, pol->keep_prob, pol->draw_counter, pol->draw_keep,
                             pol->draw_choice, pol->seed, (const long long*)pol->seed_clock, pol->actions_out, stream);
}

int diral_env_prefill(DiralEnv* e, const int32_t* actions, int32_t slots, uint64_t seed, void* states_out, int out_dtype,
                      int32_t* actions_all_out, int32_t* actions_next_out, const double* rew_in, double episode, double epsilon,
                      void* stream) {
  if (!e || !actions || !actions_next_out || slots < 1) return DIRAL_ERR_BAD_ARG;
  if (out_dtype != DIRAL_F32 && out_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  if (e->prev_obs) return DIRAL_ERR_BAD_CONFIG;                  // State.piggybacking: my_step only (diral_env_step)
  // the secondary observation modes are launches of their own behind ONE state vector (posdist_kernel.hpp)
  if (states_out && (has(&e->cfg, DIRAL_F_ADD_POSDIST) || (has(&e->cfg, DIRAL_F_ADD_POSDIST_PIGGY) && e->cfg.posdist_type == 1)))
    return DIRAL_ERR_UNSUPPORTED;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  StepParams p = e->base;
  p.mode = DIRAL_STEP_DESIGN; p.t = 0; p.episode = episode; p.eps = epsilon; p.out_f64 = (out_dtype == DIRAL_F64);
  p.actions = actions;
  p.state_out = e->S > 0 ? states_out : nullptr;
  p.rew_out = nullptr; p.done_out = nullptr; p.chobs_out = nullptr; p.chobs_in = nullptr; p.rew_in = nullptr;
  if (!step_dispatch(e, p).prefill_ok) return DIRAL_ERR_UNSUPPORTED;   // (nothing launched: the caller loops sample + step + observe)
  PolParams q;
  std::memset(&q, 0, sizeof(q));
  q.K = slots; q.prefill = 1; q.seed = seed; q.idx0 = (uint64_t)e->env_offset * (uint64_t)e->N;
  q.actions_out = actions_next_out; q.actions_all = actions_all_out; q.rew_in = rew_in; q.vel_w = e->vel;
  bool fused = false;
  HIP_TRY(e, launch_step_any(e, p, (hipStream_t)stream, &q, &fused));
  return fused ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_env_observe(DiralEnv* e, const int32_t* actions, const double* chobs_in, const double* rew_in,
                      void* state_out, int out_dtype, double episode, double epsilon, void* stream) {
  if (!e || !actions || !state_out) return DIRAL_ERR_BAD_ARG;
  if (out_dtype != DIRAL_F32 && out_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  if (e->S == 0) return DIRAL_OK;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  StepParams p = e->base;
  p.mode = kModeObserve; p.t = 0; p.episode = episode; p.eps = epsilon; p.out_f64 = (out_dtype == DIRAL_F64);
  p.actions = actions; p.state_out = state_out;
  p.rew_out = nullptr; p.done_out = nullptr; p.chobs_out = nullptr;
  p.chobs_in = chobs_in; p.rew_in = rew_in;
  if (e->kernel_path == DIRAL_PATH_GENERAL || runs_large(e)) HIP_TRY(e, launch_step_any(e, p, (hipStream_t)stream));   // (tests: the general kernel's observe mode; step_large.hpp's)
  else HIP_TRY(e, launch_observe_any(e, p, (hipStream_t)stream));
  HIP_TRY(e, launch_posdist_if_needed(e, p, (hipStream_t)stream));
  if (e->prev_obs && e->off_chobs_pb >= 0) {                    // `obs` = piggy_obs, A * A values per agent
    const PiggyParams q = piggy_params(e, p);
    const size_t total = (size_t)e->B * e->N * e->A * e->A;
    hipLaunchKernelGGL(piggy_fill_kernel, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, q, total);
    HIP_TRY(e, hipGetLastError());
  }
  return DIRAL_OK;
}

int diral_env_update_velocity(DiralEnv* e, const uint8_t* draws, uint64_t seed, void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  if (!has(&e->cfg, DIRAL_F_MOBILITY_VARY)) return DIRAL_OK;   // test_env.py:503
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const size_t bn = (size_t)e->B * e->N;
  hipLaunchKernelGGL(velocity_kernel, dim3(blocks(bn, 256)), dim3(256), 0, (hipStream_t)stream, (int)bn, draws,
                     seed, (uint64_t)e->env_offset * (uint64_t)e->N, e->vel);
  HIP_TRY(e, hipGetLastError());
  return DIRAL_OK;
}

int diral_env_sample(DiralEnv* e, int32_t* actions_out, uint64_t seed, void* stream) {
  if (!e || !actions_out) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const size_t bn = (size_t)e->B * e->N;
  hipLaunchKernelGGL(sample_kernel, dim3(blocks(bn, 256)), dim3(256), 0, (hipStream_t)stream, (int)bn, e->A, seed,
                     (uint64_t)e->env_offset * (uint64_t)e->N, actions_out);
  HIP_TRY(e, hipGetLastError());
  return DIRAL_OK;
}

int diral_env_info_age(DiralEnv* e, int64_t t, int32_t* out, void* stream) {
  if (!e || !out) return DIRAL_ERR_BAD_ARG;
  if (!e->la) return DIRAL_ERR_BAD_CONFIG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipLaunchKernelGGL(info_age_kernel, dim3(e->B), dim3(256), 0, (hipStream_t)stream, e->N, (long long)t, e->la, out);
  HIP_TRY(e, hipGetLastError());
  return DIRAL_OK;
}

int diral_env_export_state(DiralEnv* e, double* pos_x, double* pos_y, double* vel, int32_t* tab_seq,
                           int32_t* tab_age, double* tab_x, double* tab_y, int32_t* last_arrival, void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t bn = (size_t)e->B * e->N;
  if (pos_x) HIP_TRY(e, hipMemcpyAsync(pos_x, e->pos_x, bn * 8, hipMemcpyDeviceToDevice, s));
  if (pos_y) HIP_TRY(e, hipMemcpyAsync(pos_y, e->pos_y, bn * 8, hipMemcpyDeviceToDevice, s));
  if (vel) HIP_TRY(e, hipMemcpyAsync(vel, e->vel, bn * 8, hipMemcpyDeviceToDevice, s));
  if (tab_seq || tab_age || tab_x || tab_y) {
    const size_t total = bn * e->N;
    HIP_TRY(e, ensure_plane(e, s));                               // (sequence numbers and ages too: the packed table of N <= 64)
    hipLaunchKernelGGL(export_tables_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey,
                       e->tx, e->pos_y, tab_seq, tab_age, tab_x, tab_y);
    HIP_TRY(e, hipGetLastError());
  }
  if (last_arrival) {
    if (!e->la) return DIRAL_ERR_BAD_CONFIG;
    HIP_TRY(e, hipMemcpyAsync(last_arrival, e->la, bn * e->N * 4, hipMemcpyDeviceToDevice, s));
  }
  return DIRAL_OK;
}

int diral_env_import_state(DiralEnv* e, const double* pos_x, const double* pos_y, const double* vel,
                           const int32_t* tab_seq, const int32_t* tab_age, const double* tab_x,
                           const int32_t* last_arrival, void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t bn = (size_t)e->B * e->N;
  if (pos_x) HIP_TRY(e, hipMemcpyAsync(e->pos_x, pos_x, bn * 8, hipMemcpyDeviceToDevice, s));
  if (pos_y) {
    HIP_TRY(e, hipMemcpyAsync(e->pos_y, pos_y, bn * 8, hipMemcpyDeviceToDevice, s));
    if (refresh_flat_y(e, s) != DIRAL_OK) return DIRAL_ERR_HIP;
  }
  if (vel) HIP_TRY(e, hipMemcpyAsync(e->vel, vel, bn * 8, hipMemcpyDeviceToDevice, s));
  if (tab_seq || tab_age || tab_x) {
    const size_t total = bn * e->N;
    HIP_TRY(e, ensure_plane(e, s));                             // (a partial import keeps the other planes)
    e->ring_valid = false;
    hipLaunchKernelGGL(import_tables_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, tab_seq,
                       tab_age, tab_x, e->tkey, e->tx);
    HIP_TRY(e, hipGetLastError());
    // the ring again, right away (not at the next step: a step sequence captured into a hipGraph must not
    // contain a rebuild from a plane that later replays find stale)
    if (e->ring) { HIP_TRY(e, ensure_ring(e, s)); HIP_TRY(e, verify_ring(e, s)); }
  }
  if (last_arrival) {
    if (!e->la) return DIRAL_ERR_BAD_CONFIG;
    HIP_TRY(e, hipMemcpyAsync(e->la, last_arrival, bn * e->N * 4, hipMemcpyDeviceToDevice, s));
  }
  return DIRAL_OK;
}

int diral_env_export_prev_obs(DiralEnv* e, double* prev_obs, void* stream) {
  if (!e || !prev_obs) return DIRAL_ERR_BAD_ARG;
  if (!e->prev_obs) return DIRAL_ERR_BAD_CONFIG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  HIP_TRY(e, hipMemcpyAsync(prev_obs, e->prev_obs, (size_t)e->B * e->N * e->A * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DIRAL_OK;
}

int diral_env_import_prev_obs(DiralEnv* e, const double* prev_obs, void* stream) {
  if (!e || !prev_obs) return DIRAL_ERR_BAD_ARG;
  if (!e->prev_obs) return DIRAL_ERR_BAD_CONFIG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  HIP_TRY(e, hipMemcpyAsync(e->prev_obs, prev_obs, (size_t)e->B * e->N * e->A * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DIRAL_OK;
}

int diral_env_export_entries(DiralEnv* e, DiralNeighborEntry* entries, void* stream) {
  if (!e || !entries) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)e->B * e->N * e->N;
  HIP_TRY(e, ensure_plane(e, s));
  hipLaunchKernelGGL(export_entries_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, e->tkey,
                     e->tx, e->pos_y, entries);
  HIP_TRY(e, hipGetLastError());
  return DIRAL_OK;
}

int diral_env_import_entries(DiralEnv* e, const DiralNeighborEntry* entries, void* stream) {
  if (!e || !entries) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)e->B * e->N * e->N;
  e->plane_valid = true;                                        // every entry of the plane is rewritten
  e->ring_valid = false;
  hipLaunchKernelGGL(import_entries_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, e->B, e->N, e->NV, e->NR, entries,
                     e->tkey, e->tx);
  HIP_TRY(e, hipGetLastError());
  if (e->ring) { HIP_TRY(e, ensure_ring(e, s)); HIP_TRY(e, verify_ring(e, s)); }   // as in diral_env_import_state
  return DIRAL_OK;
}

int diral_env_set_trace(DiralEnv* e, const double* x_positions, int T, int per_env, void* stream) {
  if (!e || T < 0) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(e, hipStreamSynchronize(s));               // no launch may still read the old copy
  if (e->trace) { (void)hipFree(e->trace); e->trace = nullptr; }
  e->trace_len = 0; e->trace_per_env = 0;
  // the step parameters stop pointing at the freed copy BEFORE anything below can fail
  e->base.trace = nullptr; e->base.trace_len = 0; e->base.trace_per_env = 0;
  if (x_positions && T > 0) {
    const size_t n = (size_t)(per_env ? e->B : 1) * T * e->N;
    double* fresh = nullptr;
    HIP_TRY(e, hipMalloc((void**)&fresh, n * 8));
    hipError_t st = hipMemcpyAsync(fresh, x_positions, n * 8, hipMemcpyDeviceToDevice, s);
    if (st == hipSuccess) st = hipStreamSynchronize(s);
    if (st != hipSuccess) { (void)hipFree(fresh); return note_hip(e, st, "diral_env_set_trace copy"); }
    e->trace = fresh;
    e->trace_len = T; e->trace_per_env = per_env ? 1 : 0;
  }
  e->base.trace = e->trace; e->base.trace_len = e->trace_len; e->base.trace_per_env = e->trace_per_env;
  return DIRAL_OK;
}

int diral_env_metrics(DiralEnv* e, double* out, int clear, void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const int total = e->B * DIRAL_M_COLUMNS;
  hipLaunchKernelGGL(metrics_kernel, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, total, e->metrics,
                     out, clear);
  HIP_TRY(e, hipGetLastError());
  return DIRAL_OK;
}

#if defined(DIRAL_TIMING) || defined(DIRAL_DEBUG_XG)
// -DDIRAL_TIMING tuning builds only (profiles/phase_timing.py binds it by name); compiled out of the
// release library, so the release symbol table is exactly include/diral_env.h: copy the phase
// timestamps [B][waves][8] to a host buffer.
int diral_env_debug_timing(DiralEnv* e, unsigned long long* host_out, int waves) {
  if (!e || !e->dbg || !host_out) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  if (hipDeviceSynchronize() != hipSuccess) return DIRAL_ERR_HIP;
  if (hipMemcpy(host_out, e->dbg, (size_t)e->B * waves * 8 * 8, hipMemcpyDeviceToHost) != hipSuccess) return DIRAL_ERR_HIP;
  return DIRAL_OK;
}
#endif

int diral_sps_step(int agents, int num_channels, const double* selection_window, int32_t* prev_action,
                   int32_t* counter, double rssi_threshold, double inc_db, double keep_prob,
                   const int32_t* draw_counter, const double* draw_keep, const int32_t* draw_choice, uint64_t seed,
                   int32_t* actions_out, void* stream) {
  if (agents < 1 || num_channels < 1 || !selection_window || !prev_action || !counter || !actions_out)
    return DIRAL_ERR_BAD_ARG;
  PtrDeviceGuard guard(prev_action);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(blocks((size_t)agents, 256)), t(256);
#define DIRAL_SPS_WAVE(NC)                                                                                           \
  hipLaunchKernelGGL((sps_step_wave_kernel<NC, double, false>), g, t, 0, st, agents, num_channels, selection_window, \
                     (const int32_t*)nullptr, prev_action, counter, rssi_threshold, inc_db, keep_prob, draw_counter, \
                     draw_keep, draw_choice, seed, (const long long*)nullptr, actions_out)
  if (num_channels <= 64) DIRAL_SPS_WAVE(1);
  else if (num_channels <= 128) DIRAL_SPS_WAVE(2);
  else if (num_channels <= kSpsWaveMaxA) DIRAL_SPS_WAVE(4);
  else                                                            // one thread per agent
    hipLaunchKernelGGL(sps_step_kernel, dim3(blocks((size_t)agents, 128)), dim3(128), 0, st, agents, num_channels,
                       selection_window, prev_action, counter, rssi_threshold, inc_db, keep_prob, draw_counter, draw_keep,
                       draw_choice, seed, actions_out);
#undef DIRAL_SPS_WAVE
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_driver_shape(int envs, int num_users, int num_channels, const void* reward_in, int dtype,
                       const int32_t* actions, const int32_t* ia, int64_t* sum_ia_prev, int32_t* pen_counter,
                       int32_t* prev_actions, int flags, int ia_penalty_threshold, double ia_penalty_value,
                       void* reward_out, void* sum_r_out, void* collision_out, int64_t* ia_sum_out,
                       int32_t* ia_penalty_out, void* stream) {
  if (envs < 1 || num_users < 1 || !reward_in || !reward_out) return DIRAL_ERR_BAD_ARG;
  if (dtype != DIRAL_F32 && dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  if ((flags & 4) && (!actions || !pen_counter || !prev_actions)) return DIRAL_ERR_BAD_ARG;
  if ((flags & 2) && (!ia || !sum_ia_prev)) return DIRAL_ERR_BAD_ARG;
  PtrDeviceGuard guard(reward_in);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  if (!ia && !(flags & 2) && num_users >= 8 && num_users <= 64) {
    // no information-age terms, one wave per env (aux_kernels.hpp)
    const dim3 gw(blocks((size_t)envs, kShapeWaveBlock / 64)), tw(kShapeWaveBlock);
    if (dtype == DIRAL_F64)
      hipLaunchKernelGGL(driver_shape_wave_kernel<double>, gw, tw, 0, (hipStream_t)stream, envs, num_users, num_channels,
                         static_cast<const double*>(reward_in), actions, pen_counter, prev_actions, flags, ia_penalty_threshold,
                         ia_penalty_value, static_cast<double*>(reward_out), static_cast<double*>(sum_r_out),
                         static_cast<double*>(collision_out));
    else
      hipLaunchKernelGGL(driver_shape_wave_kernel<float>, gw, tw, 0, (hipStream_t)stream, envs, num_users, num_channels,
                         static_cast<const float*>(reward_in), actions, pen_counter, prev_actions, flags, ia_penalty_threshold,
                         ia_penalty_value, static_cast<float*>(reward_out), static_cast<float*>(sum_r_out),
                         static_cast<float*>(collision_out));
    return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
  }
  const dim3 g(blocks((size_t)envs, kShapeEnvsPerBlock)), t(256);
  if (dtype == DIRAL_F64)
    hipLaunchKernelGGL(driver_shape_kernel<double>, g, t, 0, (hipStream_t)stream, envs, num_users, num_channels,
                       static_cast<const double*>(reward_in), actions, ia, (long long*)sum_ia_prev, pen_counter, prev_actions,
                       flags, ia_penalty_threshold, ia_penalty_value, static_cast<double*>(reward_out),
                       static_cast<double*>(sum_r_out), static_cast<double*>(collision_out), (long long*)ia_sum_out,
                       ia_penalty_out);
  else
    hipLaunchKernelGGL(driver_shape_kernel<float>, g, t, 0, (hipStream_t)stream, envs, num_users, num_channels,
                       static_cast<const float*>(reward_in), actions, ia, (long long*)sum_ia_prev, pen_counter, prev_actions,
                       flags, ia_penalty_threshold, ia_penalty_value, static_cast<float*>(reward_out),
                       static_cast<float*>(sum_r_out), static_cast<float*>(collision_out), (long long*)ia_sum_out,
                       ia_penalty_out);
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_sps_window_from_chobs(int agents, int num_channels, const void* chobs, int chobs_dtype,
                                const int32_t* actions, double* window_out, void* stream) {
  if (agents < 1 || num_channels < 1 || !chobs || !actions || !window_out) return DIRAL_ERR_BAD_ARG;
  if (chobs_dtype != DIRAL_F32 && chobs_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  PtrDeviceGuard guard(chobs);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const size_t total = (size_t)agents * num_channels;
  if (chobs_dtype == DIRAL_F64)
    hipLaunchKernelGGL(sps_window_kernel<double>, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       num_channels, static_cast<const double*>(chobs), actions, window_out);
  else
    hipLaunchKernelGGL(sps_window_kernel<float>, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, total,
                       num_channels, static_cast<const float*>(chobs), actions, window_out);
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

static int sps_step_chobs_impl(int agents, int num_channels, const void* chobs, int chobs_dtype, const int32_t* actions,
                               int32_t* prev_action, int32_t* counter, double rssi_threshold, double inc_db,
                               double keep_prob, const int32_t* draw_counter, const double* draw_keep,
                               const int32_t* draw_choice, uint64_t seed, const long long* clock, int32_t* actions_out,
                               void* stream) {
  if (agents < 1 || num_channels < 1 || !chobs || !actions || !prev_action || !counter || !actions_out)
    return DIRAL_ERR_BAD_ARG;
  if (chobs_dtype != DIRAL_F32 && chobs_dtype != DIRAL_F64) return DIRAL_ERR_BAD_ARG;
  if (num_channels > kSpsWaveMaxA) return DIRAL_ERR_UNSUPPORTED;   // use window_from_chobs + sps_step
  PtrDeviceGuard guard(prev_action);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(blocks((size_t)agents, 256)), t(256);
#define DIRAL_SPS_WAVE(NC, T)                                                                                          \
  hipLaunchKernelGGL((sps_step_wave_kernel<NC, T, true>), g, t, 0, st, agents, num_channels, static_cast<const T*>(chobs), \
                     actions, prev_action, counter, rssi_threshold, inc_db, keep_prob, draw_counter, draw_keep,        \
                     draw_choice, seed, clock, actions_out)
#define DIRAL_SPS_WAVE_T(T)                          \
  do {                                               \
    if (num_channels <= 64) DIRAL_SPS_WAVE(1, T);    \
    else if (num_channels <= 128) DIRAL_SPS_WAVE(2, T); \
    else DIRAL_SPS_WAVE(4, T);                       \
  } while (0)
  if (chobs_dtype == DIRAL_F64) DIRAL_SPS_WAVE_T(double);
  else DIRAL_SPS_WAVE_T(float);
#undef DIRAL_SPS_WAVE_T
#undef DIRAL_SPS_WAVE
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_sps_step_chobs(int agents, int num_channels, const void* chobs, int chobs_dtype, const int32_t* actions,
                         int32_t* prev_action, int32_t* counter, double rssi_threshold, double inc_db,
                         double keep_prob, const int32_t* draw_counter, const double* draw_keep,
                         const int32_t* draw_choice, uint64_t seed, int32_t* actions_out, void* stream) {
  return sps_step_chobs_impl(agents, num_channels, chobs, chobs_dtype, actions, prev_action, counter, rssi_threshold, inc_db,
                             keep_prob, draw_counter, draw_keep, draw_choice, seed, nullptr, actions_out, stream);
}

int diral_sps_step_chobs_clocked(int agents, int num_channels, const void* chobs, int chobs_dtype, const int32_t* actions,
                                 int32_t* prev_action, int32_t* counter, double rssi_threshold, double inc_db,
                                 double keep_prob, uint64_t seed, const int64_t* clock, int32_t* actions_out, void* stream) {
  if (!clock) return DIRAL_ERR_BAD_ARG;
  return sps_step_chobs_impl(agents, num_channels, chobs, chobs_dtype, actions, prev_action, counter, rssi_threshold, inc_db,
                             keep_prob, nullptr, nullptr, nullptr, seed, (const long long*)clock, actions_out, stream);
}

int diral_clock_add(int64_t* clock, int64_t inc, void* stream) {
  if (!clock) return DIRAL_ERR_BAD_ARG;
  PtrDeviceGuard guard(clock);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipLaunchKernelGGL(clock_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)clock, (long long)inc);
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_env_set_clock(DiralEnv* e, const int64_t* t_dev) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  e->base.t_dev = (const long long*)t_dev;
  return DIRAL_OK;
}

int diral_env_set_capture_rotation(DiralEnv* e, int on, int* phase) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  e->capture_rotates = on != 0;
  if (phase) *phase = (int)(e->slow_launches % 3);
  return DIRAL_OK;
}

int diral_env_align_phase(DiralEnv* e, int phase, void* stream) {
  if (!e || phase < 0 || phase > 2) return DIRAL_ERR_BAD_ARG;
  if (!e->slow || (int)(e->slow_launches % 3) == phase) return DIRAL_OK;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipStream_t s = (hipStream_t)stream;
  if (stream_is_capturing(s)) { e->capture_violation = true; return DIRAL_ERR_CAPTURE; }
  // the sets a launch of the new phase reads, builds and clears must not hold what launches of another phase left half
  // done (a cleared count under flags that still stand): all three empty = every env runs in dispatch order once
  if (hipMemsetAsync(e->slow, 0, 3 * slow_set_words(e) * 4, s) != hipSuccess) return DIRAL_ERR_HIP;
  while ((int)(e->slow_launches % 3) != phase) ++e->slow_launches;
  return DIRAL_OK;
}

int diral_sps_init(int agents, int selection_window, int32_t* prev_action, int32_t* counter, uint64_t seed,
                   void* stream) {
  if (agents < 1 || selection_window < 0 || !prev_action || !counter) return DIRAL_ERR_BAD_ARG;
  PtrDeviceGuard guard(prev_action);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  hipLaunchKernelGGL(sps_init_kernel, dim3(blocks((size_t)agents, 256)), dim3(256), 0, (hipStream_t)stream, agents,
                     selection_window, seed, prev_action, counter);
  return hipGetLastError() == hipSuccess ? DIRAL_OK : DIRAL_ERR_HIP;
}

int diral_env_check(DiralEnv* e, void* stream) {
  if (!e) return DIRAL_ERR_BAD_ARG;
  DeviceGuard guard(e->device);
  if (!guard.ok) return DIRAL_ERR_NO_DEVICE;
  uint32_t flags = 0;
  HIP_TRY(e, hipMemcpyAsync(&flags, e->err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
  if (flags) HIP_TRY(e, hipMemsetAsync(e->err, 0, 4, (hipStream_t)stream));
  if (flags & kErrAction) return DIRAL_ERR_ACTION_RANGE;
  if (flags & kErrSeq) return DIRAL_ERR_SEQ_OVERFLOW;
  if (flags & kErrTable) return DIRAL_ERR_TABLE_CONFLICT;
  if (flags & kErrPiggy) return DIRAL_ERR_PIGGY_NO_TX;
  return DIRAL_OK;
}

}  // extern "C"
