// step_fast64.hpp - the fused env-step kernel specialised for the headline
// configuration: N <= 64 vehicles (one wavefront lane per vehicle), A <= 64
// resources, the toy YAML's State flags (one-hot action + type-2 piggybacked
// positional histogram), my_step / my_step_ch / my_step_design + obtain_state,
// float32 or float64 outputs.
//
// Same semantics as step_kernel.hpp (which stays the general path and is what
// the parity tests compare this kernel against, bit for bit); what changes is
// the schedule.  Profiling on MI355X (profiles/) showed the general kernel
//   - LATENCY-bound on LDS round trips (63 % of wave cycles waiting),
//   - then instruction-ISSUE-bound (~7000 instructions per wave, 16 waves per
//     SIMD per launch), with ~90 KB of inlined rare-path code (sqrt/exp/fmod),
//   - with the gossip merge sitting on the CU-shared LDS pipe (a
//     ds_bpermute_b32 costs ~5.5 LDS cycles; one env-slot needs ~1800).
// So here:
//   * closest-transmitter search without LDS: lane = vehicle, a transmitter's
//     position is broadcast with v_readlane; each wave owns the resources
//     i == wave (mod 4) and walks their transmitters in ascending id with a
//     strict '<' (the reference's lowest-id tie-break, network.py:387-392);
//   * gossip merge key[u] = max(key[u], key[m_i(u)]): one ds_bpermute + max per
//     (resource, column PAIR): two columns travel as 16-bit (rank, source) keys in
//     one register (exactness argument and fallback at the merge loop);
//   * a compact parameter block, 32-bit table offsets, padded
//     viewer stride (NV = 64) so table loads/stores need no lane predicate;
//   * when every vehicle has y == 0 (any random topology, network.py:104) the
//     distance is |dx| exactly and the dy logic is compiled out (FLAT);
//   * branch-free histogram bin search; rare paths out of line.
#pragma once
#include "common.hpp"
#include "policy_device.hpp"
#include "rich_out.hpp"
#include "step_kernel.hpp"

namespace diral {

#ifndef DIRAL_FAST_MINWAVES_KS
#define DIRAL_FAST_MINWAVES_KS 6        // step_fast64_slots_kernel (K slots per launch): <= 84 VGPRs, six workgroups per CU.  C2, 4096 envs,
                                        // K = 25, us per slot: 57.4 / 52.6 / 50.8 / 55.8 for 4 / 5 / 6 / 7 (the one-slot fused launch: 56.8)
#endif
#ifndef DIRAL_FAST_MINWAVES
#define DIRAL_FAST_MINWAVES 7           // <= 72 VGPRs, 7 waves/SIMD.  8 (64 VGPRs) was the optimum while the phases were latency-bound;
                                        // with the xpos ring (VALU-bound, 15 spilled registers at 64) 7 is 3 % faster, 6 no better
#endif

// Thermometer codes of table lags: c(lag) = (0xff << lag) & 0xff for lag 0..7, 0 = never heard.
// The codes form a chain under bit inclusion, so the code of the smaller lag (the fresher entry)
// is the bitwise OR, and the lag comes back as 8 - popcount.  Four lag bytes (0..7 exact, 12 =
// never heard) -> four codes with one v_perm_b32: selectors 0-7 pick bytes of the table
// {0xff, 0xfe, 0xfc, 0xf8, 0xf0, 0xe0, 0xc0, 0x80}, selector 12 yields 0x00.
__device__ inline unsigned int thermo_codes(unsigned int lag_bytes) {
  return __builtin_amdgcn_perm(0x80c0e0f0u, 0xf8fcfeffu, lag_bytes);
}

constexpr int kFastMaxA = 64;           // LDS is sized by the actual A (rounded up to 32): A <= 32 keeps 8 workgroups per CU

struct FastParams {
  int N, A, K, NR;               // NR: padded subject rows (multiple of 16); viewer stride is 64
  int NV;                        // viewer stride (step_wide.hpp; 64 for step_fast64)
  uint32_t flags;
  int reward_design, age_limit, episode_interval;
  int design;                    // 1: my_step_design (test_env.py:269-349) - runtime switch of the non-CH instantiation
  int done_now;                  // t % episode_interval == episode_interval - 1 (main_test.py:226), evaluated on the host
  int notab;                     // 1: State.add_positional_dist_piggy is off - the reference keeps no neighbour tables at all
                                 // (test_env.py:138-139, 231-238: no periodic_update, no received_update): stamp, merge and
                                 // histogram are skipped - EXTRA + RICH instantiations
  int nomove;                    // 1: static topology (`mobility: False` with the design topology, network.py:54-60, 302-305):
                                 // update_mobility does nothing - EXTRA instantiations
  int prr;                       // 1: my_step also accumulates the PRR metric columns (DIRAL_F_TRACK_PRR, a build extension:
                                 // the reception ratio of test_env.py:384-405 per colliding transmitter) - EXTRA instantiations
  int chobs_mode;                // RICH: bit 0 = chobs_out is set; bit 1 = the channel observation is the distance to the
                                 // closest in-range transmitter (my_step with State.type 2) instead of the constant 1
                                 // (my_step_ch, my_step_design, State.type 1).  Host-folded: P1 touches no RichParams field
  double L, Rc, Rb, inv_w;
  long long t;
  const long long* t_dev;        // slot clock (diral_env_set_clock) or null: the slot number is t + *t_dev, read on the device -
                                 // a captured hipGraph of K steps replays with a clock that moves on
  const int32_t* actions;
  double* pos_x;
  const double* pos_y;
  const double* vel;
  uint32_t* tkey;
  double* tx;
  double* ring;                  // [B][NR][8] xpos ring (always set for step_fast64; step_wide: null = every xpos from the plane)
  // the PACKED table of step_fast64: what the merge works on is what is stored.  Row-quad q = k / 4:
  uint32_t* tcode;               // [B][NR/4][64]: byte c = thermometer code of the lag of viewer u's entry about subject 4q + c
                                 // (0xff << lag for lag 0..7; 0 = never heard, or older than 7: then `tkey` holds its sequence number)
  uint32_t* tage;                // [B][NR/4][64]: byte c = last_updated (saturating at 255) of the same entry - EVERY entry
  uint32_t* tseq;                // [B][NR]: the subjects' own sequence numbers
  uint32_t* told;                // [B][NR/4]: != 0: the quad holds an entry older than the codes reach -> keyed path
  int32_t* la;                   // last_arrival_time[tx][rx] (network.py:39-42) or null: not tracked
  const double* trace;           // replayed x positions (network.py:171-178, 194-199) or null
  int trace_len, trace_per_env;
  double* metrics;
  uint32_t* err;
  const double* edges;
  const double* inv_tab;          // [256] 1.0 / n (0 for n = 0): the f32 histogram output multiplies instead of dividing
  void* state_out;                // float* or double* (OUT64)
  void* rew_out;
  uint8_t* done_out;
  unsigned long long* dbg;
  int B;                          // envs of the handle (the grid may be larger: slow-first blocks below)
  // float32 screening of the histogram bin in the fast quads (P3b): the bin of v = xpos - own position computed from
  // float32 copies, exact whenever its fraction is further than `f32_m16` / 65536 bin widths from an integer (the
  // host's bound on everything float32 can lose for positions up to `f32_xmax`); lanes inside the band take the
  // float64 statement.  f32_m16 = 0: off (a highway too long for float32 to be worth it).
  int f32_m16;
  float f32_xmax;
  // Slow envs first (step_fast64 only).  An env whose tables hold entries beyond the codes runs its
  // quads on the keyed path and takes two to three times as long as the others; a launch ends when its last workgroup
  // does, so such a workgroup must not be among the last to START.  Every launch leaves, for the next one, the list of
  // the envs it found slow (`told` flags set for the next slot) and a flag per env; the next launch runs the listed envs
  // in its first fast_slow_max(B) blocks - dispatched first - and the block that would have taken such an env in dispatch
  // order exits at once.  Three rotating sets (the host counts launches): read set r, build set r + 1, EMPTY set r + 2
  // (its count and every env's flag: at every launch boundary each set is either a complete list or empty, so a launch
  // that reads any of them - a captured launch replays against the set it was baked with, whatever the eager launches in
  // between did to it - steps every env exactly once).  A captured launch gets the read set only (slow_*_w / _z null: a
  // replayed graph cannot rotate), unless the graph rotates as a whole (diral_env_set_capture_rotation).
  // slow_cnt_r null: blocks = envs in order (DIRAL_NO_SLOW_FIRST).
  const uint32_t* slow_cnt_r;     // [1] number of listed envs
  const uint32_t* slow_list_r;    // [fast_slow_max(B)]
  const uint32_t* slow_flag_r;    // [B] != 0: listed
  uint32_t* slow_cnt_w;
  uint32_t* slow_list_w;
  uint32_t* slow_flag_w;
  uint32_t* slow_cnt_z;           // the set [count | list | flags] the launch after the next will build: emptied here
};
// listed envs per launch (an env beyond that keeps its place in dispatch order): a quarter of the batch, 16 ... 4096
#ifndef DIRAL_SLOW_SHIFT
#define DIRAL_SLOW_SHIFT 2             // a quarter of the batch (an eighth: sticky policies overflow the list, c2_sticky_0.9 58 -> 52 us; half: no better)
#endif
__host__ __device__ inline int fast_slow_max(int B) { const int m = B >> DIRAL_SLOW_SHIFT; return m < 16 ? 16 : (m > 4096 ? 4096 : m); }

// Late-bound kernel arguments.  The compiler hoists the scalar loads of EVERY by-value kernel
// argument to the kernel entry and then keeps (or spills, through v_writelane / v_readlane - VALU
// instructions inside the hot loops) the SGPRs of values only the last phases use: output
// pointers, section offsets.  Reading such fields through the kernarg segment pointer behind an
// opaque asm pins their s_load to the point of use instead (SGPR spills of every instantiation:
// profiles/r02/resource_usage.txt).
struct RichParams;
typedef const __attribute__((address_space(4))) FastParams* LateFastArgs;
__device__ inline unsigned long long late_kernarg_base() {
  unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return a;
}
// byte offset of the second kernel argument (RichParams) in the kernarg segment
constexpr unsigned long long kRichArgOffset = (sizeof(FastParams) + alignof(RichParams) - 1) / alignof(RichParams) * alignof(RichParams);
typedef const __attribute__((address_space(4))) RichParams* LateRichArgs;
// ... and of the third (PolParams)
constexpr unsigned long long kPolArgOffset = (kRichArgOffset + sizeof(RichParams) + alignof(PolParams) - 1) / alignof(PolParams) * alignof(PolParams);
typedef const __attribute__((address_space(4))) PolParams* LatePolArgs;
__device__ inline RichParams load_rich_args(unsigned long long kernarg_base) {
  const LateRichArgs a = (LateRichArgs)(kernarg_base + kRichArgOffset);
  RichParams r;
  r.chobs_out = a->chobs_out; r.S = a->S; r.state_type = a->state_type; r.plain_state = a->plain_state;
  r.off_act = a->off_act; r.off_chobs = a->off_chobs; r.off_hist = a->off_hist; r.off_rew = a->off_rew;
  r.off_idx = a->off_idx; r.off_pos = a->off_pos; r.off_vel = a->off_vel; r.off_fp = a->off_fp;
  r.off_skip = a->off_skip; r.len_skip = a->len_skip;
  r.H = a->H; r.episode = a->episode; r.eps = a->eps; r.vel = a->vel; r.pos_y = a->pos_y;
  r.pf = a->pf; r.pf_threshold = a->pf_threshold; r.pf_penalty = a->pf_penalty;
  return r;
}

#ifndef DIRAL_FAST_F32_FMA
#define DIRAL_FAST_F32_FMA 1             // float32 screening of the bin: the own position folded into the fma's addend (one VALU instruction less per entry)
#endif
struct FastLds {
  uint32_t rv, edges, mask, act, hist, cnt, slow, inv, mtab, rtx, inr, px, py, npx, rew, stage, nact, kvel, total;
};
// row stride (elements) of the channel-observation staging array [vehicle][resource] of the RICH
// instantiations: a multiple of 4 elements, so that the write-out reads a 16-byte piece of a row
// with ONE ds_read_b128 (f32) / ds_read_b128 of two doubles; the 4 extra elements skew the rows
// over the banks (the column writes of P1, lane = vehicle, then conflict 4-way: 8 cheap writes
// per wave)
__host__ __device__ constexpr int fast_stage_stride(int A) { return (A <= 32 ? 32 : 64) + 4; }
// histogram row stride (words): odd (lane = row: conflict-free increments) and at least K + 1 - slot K of a row is a
// spare that takes the increments of entries that do not count (the column loop needs no exec-mask branch then)
__host__ __device__ constexpr int fast_hist_stride(int K) { return (K + 1) | 1; }
// gather-source table [vehicle][resource], one BYTE per entry (source lane * 4, the ds_bpermute address, <= 252): the
// merge reads the sources of FOUR consecutive resources with one ds_read_b32; the row stride of a32 + 4 bytes = 9 / 17
// words puts the 64 lanes' words - and the byte writes of P1, lane = row - on distinct banks
__host__ __device__ constexpr int fast_mtab_stride(int A) { return (A <= 32 ? 32 : 64) + 4; }
__host__ __device__ inline FastLds fast_lds_layout(int K, int A, bool rich, bool out64, bool flat, bool ratios = true, bool pol = false) {
  FastLds l;
  uint32_t o = 0;
  const uint32_t a32 = A <= 32 ? 32u : 64u;
  l.rv = o;    o += 8u * a32;
  l.edges = o; o += 8u * (K + 2);
  l.mask = o;  o += 8u * a32;
  l.act = o;   o += 4u * 64;
  l.hist = o;  o += 4u * fast_hist_stride(K) * 64;
  l.cnt = o;   o += 4u * 64;
  l.slow = o;  o += 16u;                     // the workgroup holds a quad flagged for the keyed path of the next slot
  l.inv = o;   o += out64 ? 0u : 8u * 64;   // float32 outputs: 1.0 / n for the 64 possible neighbour counts (P4), staged in P0
  l.mtab = o;  o += 64u * fast_mtab_stride(A);      // [vehicle][resource] gather source lane * 4 (bpermute address), bytes
  // (CH and EXTRA instantiations only - `ratios`: without them the RICH workgroup of A <= 32 stays at 20 KB, eight per CU)
  l.rtx = o;   o += ratios ? 8u * 64 : 0u;  // my_step_ch: reception ratio R per transmitter
  l.inr = o;   o += ratios ? 4u * 64 : 0u;  // my_step_ch: receivers in range per transmitter
  l.px = l.py = l.npx = l.rew = l.stage = o;
  if (!rich) { l.px = o; o += 8u * 64; }     // the pre-move positions, for P2 behind the last barrier (RICH: part of the tail below)
  if (rich) {                               // RICH output tail (rich_out.hpp): per-vehicle values by index
    l.px = o;  o += 8u * 64;
    l.py = o;  o += flat ? 0u : 8u * 64;    // every pos_y == 0: not staged (keeps 8 workgroups per CU at A <= 32)
    l.npx = o; o += 8u * 64;
    l.rew = o; o += 8u * 64;
    o = align_up(o, 16);
    l.stage = o; o += (out64 ? 8u : 4u) * 64 * fast_stage_stride(A);   // channel observation [vehicle][resource]
  }
  l.nact = o;  o += pol ? 4u * 64 : 0u;     // POL: the agents' actions of the next slot (K slots per launch)
  o = align_up(o, 8);
  l.kvel = o;  o += pol ? 8u * 64 : 0u;     // K slots per launch: the velocities the last slot's state vector reports (they live in registers)
  l.total = align_up(o, 16);
  return l;
}

// Streaming (non-temporal) stores for the state vectors: they are the last thing a
// workgroup does and nothing on the chip reads them back, so they should neither claim L2
// lines nor hold the wave until a cached write is acknowledged (measured on C2: 98 -> 89 us
// per slot; on the table stores, which the barrier and P4 already overlap, it does not pay).
__device__ inline void stream_store(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ inline void stream_store(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ inline void stream_store4(float* p, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(p));
}
__device__ inline void stream_store2(double* p, double2 v) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 vv = {v.x, v.y};
  __builtin_nontemporal_store(vv, reinterpret_cast<d2*>(p));
}

// A wave-uniform row pointer pinned into an SGPR pair, typed as a GLOBAL
// (address_space(1)) pointer: loads/stores take the scalar-base + 32-bit lane offset
// form.  Without the pin the compiler hoists per-lane 64-bit addresses out of the column
// loops (16 VGPRs); without the address space a pointer rebuilt from integers is generic
// and every access becomes a FLAT instruction (which also counts on lgkmcnt).
template <typename T>
using global_ptr = __attribute__((address_space(1))) T*;
template <typename T>
__device__ inline global_ptr<T> uniform_ptr(T* base, size_t elem_off) {
  return (global_ptr<T>)uniform_u64((unsigned long long)(base + elem_off));
}

// Sum of a double over the 64 lanes of a wave with DPP moves (row_shr 8 / 4 / 2 / 1, then row_bcast 15 and 31): VALU only -
// `__shfl_down` compiles to ds_bpermute, an LDS round trip per level, and this sits on the critical path of the wave that
// does P2.  The total lands in lane 63 and is returned wave-uniform.  (The order differs from the shuffle tree: callers whose
// sums are compared bit for bit across kernels keep the tree.)
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ inline double wave_sum_f64(double v) {
  v = dpp_add_f64<0x118, 0xf>(v);      // row_shr:8 (lanes without a source add 0)
  v = dpp_add_f64<0x114, 0xf>(v);      // row_shr:4
  v = dpp_add_f64<0x112, 0xf>(v);      // row_shr:2
  v = dpp_add_f64<0x111, 0xf>(v);      // row_shr:1: lane 15 of every row holds the row's sum
  v = dpp_add_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_add_f64<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Count of trailing zeros of byte BYTE of a word (-1 for a zero byte): one SDWA instruction.  The lag of a thermometer
// code (0xff << lag) & 0xff.
template <int BYTE>
__device__ inline int ffbl_byte(unsigned int w) {
  int r;
  static_assert(BYTE >= 0 && BYTE < 4, "byte select");
  if constexpr (BYTE == 0) asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
  if constexpr (BYTE == 1) asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
  if constexpr (BYTE == 2) asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
  if constexpr (BYTE == 3) asm("v_ffbl_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
  return r;
}
// double -> int32, truncating, SATURATING, NaN -> 0 (the hardware conversion; a C cast is undefined out of range)
__device__ inline int cvt_i32_f32_sat(float x) {                   // truncating, saturating, NaN -> 0
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ inline int cvt_i32_f64_sat(double x) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ inline double readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
__device__ inline unsigned long long readlane_u64(unsigned long long v, int srclane) {
  const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)v, srclane);
  const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(v >> 32), srclane);
  return ((unsigned long long)hi << 32) | lo;
}

// Network.dist for the fast kernel.  FLAT (all y == 0): sqrt(fl(dx*dx)) == |dx|
// exactly when 2^-500 <= |dx| <= 2^500; the exponent test is two integer ops.
template <bool FLAT>
__device__ inline double fast_dist(double x1, double y1, double x2, double y2) {
  const double dx = x2 - x1;
  const unsigned int hi = (unsigned int)__double2hiint(dx) & 0x7fffffffu;
  // (one-sided: beyond 2^500 the square overflows to inf in the reference while |dx| stays finite, but every
  // use of the result compares it with a finite range first - `d < Rc`, `d < Rb`, `d > Rc` - and agrees)
  // ... or dx == 0 exactly, where sqrt(0) == |dx| == 0 as well: the search of P1 measures every transmitter against
  // ITSELF too, so without this case each of its iterations pays the out-of-line call for that one lane
  const bool in_range = hi >= 0x20b00000u || (hi | (unsigned int)__double2loint(dx)) == 0u;
  if (FLAT) {
    if (in_range) return __hiloint2double((int)hi, __double2loint(dx));
    return dist_general(dx, 0.0);
  } else {
    const double dy = y2 - y1;
    if (in_range && dy == 0.0) return __hiloint2double((int)hi, __double2loint(dx));
    return dist_general(dx, dy);
  }
}

// Reward of a colliding resource (test_env.py:163-199) incl.
// Network.calculate_reward_weights (network.py:273-300); wave-uniform, positions
// broadcast from lanes.  Out of line: runs ~once per colliding resource.
__device__ DIRAL_OUTLINE double fast_collision_reward(int rd, uint32_t flags, double L, double Rc, int N,
                                                                  unsigned long long mk, int c, double mypx,
                                                                  double mypy) {
  int wgt = 0;
  if (rd == 1 || ((rd == 2 || rd == 5) && c == 2)) {
    double s = 0.0;                    // calculate_avg_distance (network.py:307-316)
    int cnt = 0;
    unsigned long long ma = mk;
    while (ma) {
      const int a = __builtin_ctzll(ma);
      ma &= ma - 1;
      const double xa = readlane_f64(mypx, a), ya = readlane_f64(mypy, a);
      unsigned long long mb = ma;
      while (mb) {
        const int b = __builtin_ctzll(mb);
        mb &= mb - 1;
        s = s + dist2d_leaf(xa, ya, readlane_f64(mypx, b), readlane_f64(mypy, b));
        ++cnt;
      }
    }
    const double m = (cnt == 1) ? s : s / (double)cnt;       // s/1 == s exactly
    if (flags & DIRAL_F_TOY_WEIGHTS) {
      double x_min = L + 1, x_max = -L - 1;      // calculate_norm (network.py:225-246)
      int umin = 0, umax = 0;
      for (int u = 0; u < N; ++u) {
        const double x = readlane_f64(mypx, u);
        if (x < x_min) { x_min = x; umin = u; }
        if (x > x_max) { x_max = x; umax = u; }
      }
      wgt = (m == dist2d_leaf(readlane_f64(mypx, umin), readlane_f64(mypy, umin), readlane_f64(mypx, umax),
                         readlane_f64(mypy, umax)));
    } else {
      wgt = (m > Rc);
    }
  }
  if (rd == 1) { const double R = (double)wgt / (double)c; return -1.0 * (1.0 - R); }
  if (rd == 2) return (c == 2) ? 2.0 * (double)wgt - (double)c : 0.0 - (double)c;
  if (rd == 3) { const double R = 1.0 / (double)c; return -1.0 * exp(1.0 - R); }
  if (rd == 4) return 1.0 / (double)c;
  return (c == 2 && wgt == 1) ? 0.0 : -1.0;
}

// my_step_ch reward of one transmitter (test_env.py:411-429) from its reception ratio
// R = received / in_range (1 for a sole transmitter).  Out of line: exp().
__device__ DIRAL_OUTLINE double fast_ch_reward(int rd, bool collided, double R) {
  if (collided) {
    if (rd == 3) return 1.0 - exp(1.0 - R);
    if (rd == 4) return -1.0 * exp(1.0 - R);
    return -1.0 * (1.0 - R);
  }
  if (rd == 4) return exp(1.0);
  return 1.0;
}

// The SPS agents of one env decide from the channel observation staged in LDS (POL instantiations): what
// sps_step_wave_kernel<1, T, true> does with the rows it loads from HBM, for the 64 lanes = vehicles of this wave.
// `stage`: [vehicle][SA] of out dtype, as written to chobs_out; `own`: this slot's action of the lane's vehicle.
// Out of line: log10 and the candidate ranking stay out of the step kernel's register allocation; runs once per env.
template <typename T>
__device__ DIRAL_OUTLINE void fast_sps_decide(const T* stage, int SA, int A, int N, size_t bN, int lane, int own,
                                              int& action, int& cnt, const PolParams* q, int ks = 0, bool last = true) {
  // (`action`, `cnt`: the agent's prev_action and reselection counter, loaded by the caller ahead of P3 and kept in its
  // registers from slot to slot of a K-slot launch; slot `ks` draws with seed + ks: what K single-slot calls are given)
  const uint64_t seed = q->seed + (uint64_t)ks + (q->clock ? (uint64_t)*q->clock : 0ull);
  const int i = (int)bN + lane;
  const bool live = lane < N;
  const bool resel = live && sps_advance(i, cnt, q->keep_prob, q->draw_counter, q->draw_keep, seed);
  unsigned int r = 0;
  if (resel) r = q->draw_choice ? (unsigned int)q->draw_choice[i] : (unsigned int)(rng_u64(seed, 9, (uint64_t)i) >> 33);
  unsigned long long todo = __ballot(resel);
  while (todo) {
    const int j = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int prev_j = __builtin_amdgcn_readlane(action, j);
    const int own_j = __builtin_amdgcn_readlane(own, j);
    const unsigned int r_j = (unsigned int)__builtin_amdgcn_readlane((int)r, j);
    double d[1];
    d[0] = lane < A ? (double)stage[j * SA + lane] : 0.0;
    const int ch = sps_choose_chobs_wave<1>(d, lane, A, prev_j, own_j, q->threshold, q->inc_db, r_j);
    if (lane == j) action = ch;
  }
  // (prev_action only ever changes to the action chosen - v2x_sps.py:98 -, so behind the last slot of the launch the
  // agent's action IS its prev_action, whichever slot chose it)
  if (live && last) {
    q->sps_prev[i] = action;
    q->sps_counter[i] = cnt;
    q->actions_out[i] = action;
  }
}

#ifdef DIRAL_TIMING
#define DIRAL_FSTAMP(i) do { if (lane == 0 && p.dbg && !listed) { p.dbg[((size_t)b * 4 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 7) { __builtin_amdgcn_s_waitcnt(0); atomicMax(&p.dbg[(size_t)p.B * 40 + (((size_t)(p.t & 1) * gridDim.x + b) * 2) + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } } } while (0)
#else
#define DIRAL_FSTAMP(i) do {} while (0)
#endif

// CH: my_step_ch (test_env.py:351-443) instead of my_step: the reward of a transmitter is
// built from its reception ratio (PRR) instead of the collision count; the gossip, the
// move and the observation are the same.
// EXTRA: the rarely used run-time switches (my_step_design, arrival stamps) are compiled in;
// the plain instantiations stay free of them (they cost the headline kernel 4 spilled VGPRs).
// RICH: the output tail of rich_out.hpp (channel observation output, the cheap State flags)
// instead of the fixed [one-hot | histogram] state; `r` is only read by these instantiations.
// POL: the policy epilogue (PolParams: reward shaping + the SPS agents' decisions for the next slot) - RICH instantiations
// of my_step only; `q` is only read by these.

#define DIRAL_FAST_KERNEL step_fast64_kernel
#include "step_fast64_body.inc"
#undef DIRAL_FAST_KERNEL
#define DIRAL_FAST_KERNEL step_fast64_slots_kernel
#define DIRAL_FAST_KSLOTS 1
#include "step_fast64_body.inc"
#undef DIRAL_FAST_KSLOTS
#undef DIRAL_FAST_KERNEL

}  // namespace diral
