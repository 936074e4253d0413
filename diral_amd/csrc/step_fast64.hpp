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

struct FastLds {
  uint32_t rv, edges, mask, act, hist, cnt, slow, inv, mtab, rtx, inr, px, py, npx, rew, stage, total;
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
__host__ __device__ inline FastLds fast_lds_layout(int K, int A, bool rich, bool out64, bool flat, bool ratios = true) {
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
                                              int action, int cnt, const PolParams* q) {
  // (`action`, `cnt`: the agent's prev_action and reselection counter, loaded by the caller ahead of P3)
  const uint64_t seed = q->seed + (q->clock ? (uint64_t)*q->clock : 0ull);
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
  if (resel) q->sps_prev[i] = action;                                 // v2x_sps.py:98
  if (live) {
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
template <bool FLAT, bool OUT64, bool CH, bool EXTRA, bool RICH, bool POL = false>
__global__ __launch_bounds__(256, DIRAL_FAST_MINWAVES) void step_fast64_kernel(const FastParams p, const RichParams r,
                                                                                const PolParams q) {
  static_assert(!POL || (RICH && !CH), "the policy epilogue needs the staged channel observation of a my_step slot");
  extern __shared__ __align__(16) unsigned char smem[];
  const FastLds lay = fast_lds_layout(p.K, p.A, RICH, OUT64, FLAT, CH || EXTRA);
  double* s_rv = reinterpret_cast<double*>(smem + lay.rv);
  double* s_edges = reinterpret_cast<double*>(smem + lay.edges);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem + lay.mask);
  int* s_act = reinterpret_cast<int*>(smem + lay.act);
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + lay.hist);
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(smem + lay.cnt);
  typedef unsigned char mtab_t;
  typedef typename std::conditional<OUT64, double, float>::type out_t;
  mtab_t* s_mtab = reinterpret_cast<mtab_t*>(smem + lay.mtab);
  out_t* s_stage = reinterpret_cast<out_t*>(smem + lay.stage);   // RICH only
  double* s_rtx = reinterpret_cast<double*>(smem + lay.rtx);
  int* s_inr = reinterpret_cast<int*>(smem + lay.inr);
  double* s_px = reinterpret_cast<double*>(smem + lay.px);       // (s_py, s_npx, s_rew: RICH only)
  double* s_py = reinterpret_cast<double*>(smem + lay.py);
  double* s_npx = reinterpret_cast<double*>(smem + lay.npx);
  double* s_rew = reinterpret_cast<double*>(smem + lay.rew);
  const int MS = fast_mtab_stride(p.A);                          // gather-source table row stride (bytes), [vehicle][resource]
  const int SA = fast_stage_stride(p.A);

  // which env: blocks = envs in order, or (slow envs first) the listed envs in the first fast_slow_max(B) blocks
  int b = blockIdx.x;
  unsigned int listed = 0u;                  // (ordinary block) != 0: this env runs in one of the first blocks
  unsigned int* const s_slow = reinterpret_cast<unsigned int*>(smem + lay.slow);
  if (p.slow_cnt_r) {
    const int smax = fast_slow_max(p.B);
    if (blockIdx.x < (unsigned int)smax) {
      if (blockIdx.x >= *p.slow_cnt_r) return;
      b = (int)p.slow_list_r[blockIdx.x];
    } else {
      b = (int)blockIdx.x - smax;
      listed = p.slow_flag_r[b];             // a scalar load in flight next to the loads of P0; tested before any store
    }
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, A = p.A, K = p.K;
  const int KP = fast_hist_stride(K);
  const size_t bN = (size_t)b * N;
  const bool live = lane < N;
  constexpr int NV = 64;                       // padded viewer stride (host guarantees p.NV == 64)
  DIRAL_FSTAMP(0);
#ifdef DIRAL_TIMING
  // chip-wide clock (s_memrealtime, 100 MHz) at the start and the end of every workgroup, the launches alternating between
  // two sets: start skew, drain, and the gap between two launches of one stream (profiles/launch_timeline.py)
  if (tid == 0 && p.dbg && !listed) p.dbg[(size_t)p.B * 40 + (((size_t)(p.t & 1) * gridDim.x + b) * 2)] = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && p.dbg && !listed) {     // where this wave runs: HW_ID (wave / SIMD / CU / SE) and XCC_ID, behind the stamps
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[(size_t)p.B * 32 + (size_t)b * 4 + wave] = ((unsigned long long)xcc << 32) | hw;
  }
#endif

  // ---- P0: per-vehicle state straight into registers (every wave, lane = vehicle)
  // (unconditional, index-clamped loads pinned ahead of the table loads: vmcnt
  // retires in order, so P1 must not sit behind the 16 table words)
  const size_t vi = bN + (live ? lane : 0);
  const double my_edge = p.edges[tid < K ? tid : K];        // oldest load: must not sit behind the table
  const double my_inv = OUT64 ? 0.0 : p.inv_tab[lane];      // (every wave: a branch around a load makes the others wait)
  int myact = p.actions[vi];
  double mypx = p.pos_x[vi];
  double mypy = FLAT ? 0.0 : p.pos_y[vi];
  double myvel = p.vel[vi];
  __builtin_amdgcn_sched_barrier(0);
  if (!live) { myact = -1; mypx = 0.0; mypy = 0.0; myvel = 0.0; }
  // this wave's 16 subject columns (rows are padded to a multiple of 16, viewers
  // to 64: no predicate), issued AFTER the small loads (vmcnt retires in order)
  // wave-uniform base pointers + a per-lane element offset: lets the compiler use the
  // scalar-base addressing form instead of 64-bit per-lane pointer arithmetic
  unsigned int* const tk = p.tkey + ((size_t)b * p.NR + wave * 16) * NV;
  double* const txp = p.tx + ((size_t)b * p.NR + wave * 16) * NV;
  // A State block without piggybacked tables (test_env.py:138-139, 231-238: no periodic_update, no received_update,
  // no histogram) has no P3: every wave then behaves like the idle wave of a small env - no columns, no ring, no
  // active resource in the merge loops (a branch around the whole phase trips the backend: "illegal VGPR to SGPR copy")
  const bool notab = EXTRA && RICH && p.notab != 0;
  const bool has_cols = wave * 16 < p.NR && !notab;      // uniform: idle waves of a small env
  // The packed table words of this wave's 16 subject columns = 4 row-quads: FOUR code words and FOUR age words per
  // lane (a byte per column) where the (seq, age) plane took sixteen.  (Unconditional: a branch around the loads
  // would make the compiler wait for ALL of them at the first use of the per-vehicle values - the waitcnt pass
  // merges both paths conservatively; idle waves of a small env re-read quads 0..3, which always exist: NR >= 16.)
  const int NQ = p.NR >> 2;
  const size_t qbase = (size_t)b * NQ + (has_cols ? wave * 4 : 0);
  unsigned int* const tc = p.tcode + qbase * NV;
  unsigned int* const ta = p.tage + qbase * NV;
  unsigned int cw[4], ag[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { cw[q] = tc[q * NV + lane]; ag[q] = ta[q * NV + lane]; }
  // lane c (and c + 16, ...): the own sequence number of column c's subject
  unsigned int* const tsq = p.tseq + (size_t)b * p.NR + (has_cols ? wave * 16 : 0);
  const unsigned int ts_own = tsq[lane & 15];
  // lanes 0..3: the flags of this wave's four quads (`told`, left by the previous slot) - loaded here, one word a lane,
  // and turned into `badq` in front of the merge, which would otherwise wait there for a round trip to L2
  const unsigned int told_l = ((LateFastArgs)late_kernarg_base())->told[qbase + (lane & 3)];
  __builtin_amdgcn_sched_barrier(0);
  if (listed) return;                        // (uniform; nothing has been stored yet)
  if (live && (myact < 0 || myact >= A)) { atomicOr(p.err, kErrAction); myact = -1; }
  // P1 measures |x_w - x_u| for every (transmitter, vehicle) pair; |dx| IS the reference's sqrt(fl(dx^2)) iff dx == 0
  // or |dx| >= 2^-500 (fast_dist).  If every position of the env is 0 or at least 2^-447 in magnitude, every NONZERO
  // difference of two of them is at least one ulp of the smaller one, 2^-499: the per-pair exponent test is then
  // decided once per env (always, outside imported corner cases) and the search loop runs without it.
  bool p1_fast = false;
  if constexpr (FLAT) {
    const unsigned int xh = (unsigned int)__double2hiint(mypx) & 0x7fffffffu;
    const bool xsafe = xh >= 0x24000000u || (xh | (unsigned int)__double2loint(mypx)) == 0u;   // 2^-447: exponent field 0x240
    p1_fast = __ballot(!xsafe) == 0ull;
  }
  double mynpx = live ? py_mod_pos(mypx + myvel + p.L, p.L) : 0.0;             // network.py:203
  if (EXTRA && p.trace && live) {                                              // replay branch, network.py:194-199
    long long tt = (p.t + (p.t_dev ? *p.t_dev : 0ll)) % p.trace_len;
    if (tt < 0) tt += p.trace_len;
    const size_t base = p.trace_per_env ? (size_t)b * p.trace_len : 0;
    mynpx = p.trace[(base + (size_t)tt) * N + lane];
  }
  if (EXTRA && p.nomove) mynpx = mypx;                                         // network.py:302-305: no mobility, no move
  if (wave == 1) {
    // the transmitter sets of ALL resources (test_env.py:153-157) for P2 and the RICH tail: every vehicle ORs its bit
    // into the mask of its resource - one 64-bit LDS atomic for the wave, behind the zeroing (a wave's LDS operations
    // execute in order); the waves of P1 use their own ballots
    if (lane < (A <= 32 ? 32 : 64)) s_mask[lane] = 0ull;          // (the array holds 32 or 64 masks)
    if (myact >= 0) atomicOr(&s_mask[myact], 1ull << lane);
  }
  if (wave == 0) {
    if (lane == 0) s_slow[0] = 0u;
    s_act[lane] = myact; s_cnt[lane] = 0u;
    s_px[lane] = mypx;
    if constexpr (RICH) {
      s_npx[lane] = mynpx; s_rew[lane] = 0.0;
      if constexpr (!FLAT) s_py[lane] = mypy;
    }
  }
  for (int j = tid; j < KP * 64; j += 256) s_hist[j] = 0u;
  if constexpr (!OUT64) {
    // 1.0 / n, n = 0 .. 63 (a vehicle counts at most the other 63): the table P4 multiplies by, read HERE from memory,
    // where the wave waits for its loads anyway, and from LDS behind the last barrier - there a global load was a round
    // trip to L2 on the tail of every workgroup
    if (wave == 2) reinterpret_cast<double*>(smem + lay.inv)[lane] = my_inv;
  }
  if (tid <= K + 1) s_edges[tid] = my_edge;                 // K <= 64 < 256 threads
  DIRAL_FSTAMP(1);

  // ---- P1: per owned resource i = wave + 4*s: transmitter set, closest in-range
  // transmitter per vehicle, gather sources, collision reward ----------------------
  const bool dist_obs = RICH && (p.chobs_mode & 2) != 0;
  const bool rd2_lanes = FLAT && !CH && p.reward_design == 2 && !(p.flags & DIRAL_F_TOY_WEIGHTS) && !(EXTRA && p.design);
  const bool emit_chobs = RICH && (p.chobs_mode & 1) != 0;
  const bool need_rw = !CH && !(EXTRA && p.design) && !rd2_lanes;      // collision reward per resource, here (uniform)
  mtab_t* const mtab_row = s_mtab + lane * MS;               // this vehicle's rows of the gather table / the staging array
  out_t* const stage_row = s_stage + lane * SA;
#pragma unroll 1
  for (int i = wave; i < A; i += 4) {
    const unsigned long long mk = __ballot(myact == i);     // tx set (test_env.py:153-157)
    const int c = __popcll(mk);
    // Network.find_closest_tx (network.py:378-398): ascending id, strict '<'.  `bid` starts as the own lane: a
    // transmitter of resource i is never its own receiver, so "bid == lane" IS "no transmitter in range"
    double best = 100000.0;
    int bid = lane;
    auto search = [&](auto fast_tag) {
      constexpr bool ABS = decltype(fast_tag)::value;     // |dx| without the per-pair exponent test (p1_fast)
      unsigned long long m = mk;                            // (not empty: the caller's test)
      do {
        const int w = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(w));      // m &= ~(1 << w): one scalar instruction (m & (m - 1): three)
        double d, keep;                                     // `best` holds `keep`: the distance, or (ABS) the SIGNED difference -
        if constexpr (ABS) {                                // its magnitude is taken by the compares' source modifiers and once
          keep = mypx - readlane_f64(mypx, w);              // behind the loop instead of with an extra instruction per transmitter
          d = __builtin_fabs(keep);
        } else {
          d = keep = fast_dist<FLAT>(readlane_f64(mypx, w), FLAT ? 0.0 : readlane_f64(mypy, w), mypx, mypy);
        }
        const bool inr = d < p.Rc;
        const bool bt = inr && (d < __builtin_fabs(best));
        best = bt ? keep : best;
        bid = bt ? w : bid;
        if ((CH || (EXTRA && p.prr)) && c > 1) {                // in_range[tx] (test_env.py:395-397)
          const int n_in = __popcll(__ballot(live && (myact != i) && inr));
          if (lane == 0) s_inr[w] = n_in;
        }
        // find_closest_tx side effect (network.py:394): an out-of-range transmitter's arrival
        // stamp at this receiver becomes -1
        if (EXTRA && p.la && live && (myact != i) && !inr) p.la[(bN + w) * N + lane] = -1;
        if (EXTRA && !CH && p.design && c > 1) {
          // my_step_design: reward by the number of transmitters of this resource within 2 Rc
          // of this one (network.py:122-157): alone 1, else -n (a pair inside 2 Rc gets -2)
          const int n = 1 + __popcll(__ballot((myact == i) && (lane != w) && (d < 2.0 * p.Rc)));
          if (lane == 0) s_rtx[w] = (n == 1) ? 1.0 : -(double)n;
        }
      } while (m);
    };
    // my_step without the EXTRA switches: nothing needs "in range" per transmitter, and the closest IN-RANGE transmitter
    // (strict '<', first of equals) is the closest of all if that one is in range, else none - so the loop keeps a plain
    // running minimum (v_min_f64 on the magnitude: one instruction where the select of a float64 takes two, and no range
    // compare per transmitter) and the range test runs once per resource.  7 -> 4 vector instructions per transmitter.
    auto search_min = [&](auto fast_tag) {
      constexpr bool ABS = decltype(fast_tag)::value;
      unsigned long long m = mk;
      do {
        const int w = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(w));
        double d;
        if constexpr (ABS) d = mypx - readlane_f64(mypx, w);                     // signed: the magnitude through source modifiers
        else d = fast_dist<FLAT>(readlane_f64(mypx, w), FLAT ? 0.0 : readlane_f64(mypy, w), mypx, mypy);
        const bool bt = __builtin_fabs(d) < best;
        bid = bt ? w : bid;
        asm("v_min_f64 %0, %0, |%1|" : "+v"(best) : "v"(d));
      } while (m);
      if (!(best < p.Rc)) { best = 100000.0; bid = lane; }                       // network.py:385-386: none in range
    };
    if (mk != 0ull) {
      if constexpr (!CH && !EXTRA) {
        if (FLAT && p1_fast) search_min(std::true_type{});
        else search_min(std::false_type{});
      } else {
        if (FLAT && p1_fast) search(std::true_type{});
        else search(std::false_type{});
      }
    }
    best = __builtin_fabs(best);
    const bool self = !live || myact == i;                  // padded lanes and the transmitters of i gather from themselves
    const int src_lane = self ? lane : bid;
    const bool got = src_lane != lane;
    mtab_row[i] = (mtab_t)(src_lane << 2);
    if constexpr (RICH) {
      if (emit_chobs || POL) {
        // `obs[user][i]` of the reference step (test_env.py:143, 206, 228, 240, 306, 432): 0 on the own
        // resource or an unused one; my_step with State.type 2: the distance to the closest in-range
        // transmitter, 100000 (network.py:385) when none is in range; otherwise the constant 1.
        // Straight from the registers of the search into the staging array (lane = row); rows leave
        // coalesced after the barrier.
        const double ob = (myact == i || c == 0) ? 0.0 : (dist_obs ? best : 1.0);
        stage_row[i] = (out_t)ob;
      }
    }
    if (EXTRA && CH && p.la && got) p.la[(bN + bid) * N + lane] = (int32_t)(p.t + (p.t_dev ? *p.t_dev : 0ll));   // test_env.py:436
    if (CH || (EXTRA && p.prr)) {
      if (c > 1) {
        // received[tx] = #rx whose nearest in-range tx is tx; R = received / in_range (test_env.py:398-405)
        wave_lds_order();
        unsigned long long m2 = mk;
        while (m2) {
          const int w = __builtin_ctzll(m2);
          m2 &= m2 - 1;
          const int n_rec = __popcll(__ballot(live && (myact != i) && bid == w));
          if (lane == 0) {
            const int n_in = s_inr[w];
            s_rtx[w] = n_in > 0 ? (double)n_rec / (double)n_in : 1.0;
          }
        }
      }
    }
    // reward of a colliding resource (test_env.py:159-199).  The common case - reward_design 2 with the distance weight of
    // network.py:291-295 on the one-lane highway - is computed for ALL resources at once in P2, one lane per resource
    // (`rd2_lanes`): here it cost every wave ~10 instructions per owned resource
    if (need_rw) {
      if (c > 1) {
        const double rw = fast_collision_reward(p.reward_design, p.flags, p.L, p.Rc, N, mk, c, mypx, mypy);
        if (lane == 0) s_rv[i] = rw;
      }
    }
  }
  DIRAL_FSTAMP(2);
  __syncthreads();

  // ---- P2 (wave 0): reward per transmitter, metrics -------------------------------
  // Nothing of it is needed before the state vectors are written unless they carry the rewards (the RICH tail) or the
  // policy epilogue shapes them: otherwise (`late_p2`) wave 0 runs P2 BEHIND the last barrier, next to the other three
  // waves' P4, instead of in front of its P3 - where the other waves then waited for it at that barrier.
  bool late_p2 = !POL;
  if constexpr (RICH) late_p2 = late_p2 && ((LateRichArgs)(late_kernarg_base() + kRichArgOffset))->plain_state != 0;
  auto p2_body = [&](const double px_own, const int act_own) {
    const LateFastArgs lp = (LateFastArgs)late_kernarg_base();     // rew_out, metrics, done_out: used here only
    if (rd2_lanes) {
      // lane i: resource i.  A pair is rewarded 2 * [dist > Rc] - 2 (the mean distance of ONE pair is the distance:
      // (0 + d) / 1 == d exactly), more than two transmitters -c
      const unsigned long long mkl = lane < A ? s_mask[lane] : 0ull;
      const int cl = __popcll(mkl);
      const unsigned long long mk2 = mkl & (mkl - 1ull);
      const int la = cl > 0 ? __builtin_ctzll(mkl) : 0, lb = cl > 1 ? __builtin_ctzll(mk2) : 0;
      const double xa = __hiloint2double(__builtin_amdgcn_ds_bpermute(la << 2, __double2hiint(px_own)),
                                         __builtin_amdgcn_ds_bpermute(la << 2, __double2loint(px_own)));
      const double xb = __hiloint2double(__builtin_amdgcn_ds_bpermute(lb << 2, __double2hiint(px_own)),
                                         __builtin_amdgcn_ds_bpermute(lb << 2, __double2loint(px_own)));
      // (`p1_fast`, decided again from the positions at hand rather than kept across P3)
      const unsigned int xh2 = (unsigned int)__double2hiint(px_own) & 0x7fffffffu;
      const bool absd = __ballot(!(xh2 >= 0x24000000u || (xh2 | (unsigned int)__double2loint(px_own)) == 0u)) == 0ull;
      const double dab = absd ? __builtin_fabs(xb - xa) : fast_dist<true>(xa, 0.0, xb, 0.0);
      const double rwl = cl == 2 ? 2.0 * (double)(dab > p.Rc) - 2.0 : 0.0 - (double)cl;
      if (cl > 1) s_rv[lane] = rwl;
      wave_lds_order();
    }
    double rw = 0.0;
    int sole = 0, coll = 0;
    double prr = 0.0;
    if (live && act_own >= 0) {
      const int c = __popcll(s_mask[act_own]);
      if (CH) {
        const double R = (c > 1) ? s_rtx[lane] : 1.0;         // test_env.py:411-429
        const bool plain = (p.reward_design == 2);
        rw = plain ? ((c > 1) ? -1.0 * (1.0 - R) : 1.0) : fast_ch_reward(p.reward_design, c > 1, R);
        coll = c > 1; sole = !(c > 1); prr = R;
      } else if (c > 1) { rw = (EXTRA && p.design) ? s_rtx[lane] : s_rv[act_own]; coll = 1; } else { rw = 1.0; sole = 1; }   // test_env.py:211-222, 297-301
      if (!CH && EXTRA && p.prr) prr = (c > 1) ? s_rtx[lane] : 1.0;   // the metric only: the reward stays my_step's
      if constexpr (RICH && !CH) {
        // proportional fairness (test_env.py:215-222, my_step only): a transmitter that collided more than
        // pf_threshold slots in a row is paid pf_penalty; a successful transmission resets its counter
        const LateRichArgs lr = (LateRichArgs)((unsigned long long)lp + kRichArgOffset);
        int32_t* const pf = lr->pf;
        if (pf && !(EXTRA && p.design)) {
          if (c > 1) {
            const int pc = pf[bN + lane];
            if (pc > lr->pf_threshold) rw = lr->pf_penalty;
            pf[bN + lane] = pc + 1;
          } else {
            pf[bN + lane] = 0;
          }
        }
      }
      void* const rew_out = lp->rew_out;
      if (rew_out) {
        if constexpr (OUT64) static_cast<double*>(rew_out)[bN + lane] = rw;
        else static_cast<float*>(rew_out)[bN + lane] = (float)rw;
      }
      if constexpr (RICH) s_rew[lane] = rw;
    }
    // metric partials: the two counts are ballots, the reward sum a DPP reduction (no LDS round trips on this wave's
    // critical path); the PRR sum keeps the shuffle tree of the general kernel (compared bit for bit with it)
    const double vr = wave_sum_f64(rw);
    const int vs = __popcll(__ballot(sole != 0)), vc = __popcll(__ballot(coll != 0));
    double vp = prr;
    if (CH || EXTRA) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) vp += __shfl_down(vp, off);
    }
    if (lane == 0) {
      double* mt = lp->metrics + (size_t)b * DIRAL_M_COLUMNS;
      // (hardware f64 atomic adds without a return value: `mt[i] += x` is a global load the wave waits for - a microsecond on
      // the critical path of the workgroup; one workgroup per env and launch adds here, so the sum is the same)
      unsafeAtomicAdd(&mt[DIRAL_M_SLOTS], 1.0);
      unsafeAtomicAdd(&mt[DIRAL_M_SUM_REWARD], vr);
      unsafeAtomicAdd(&mt[DIRAL_M_TX_SOLE], (double)vs);
      unsafeAtomicAdd(&mt[DIRAL_M_TX_COLLIDED], (double)vc);
      if (CH || (EXTRA && p.prr)) { unsafeAtomicAdd(&mt[DIRAL_M_PRR_SUM], vp); unsafeAtomicAdd(&mt[DIRAL_M_PRR_CNT], (double)vs + (double)vc); }
      uint8_t* const done_out = lp->done_out;
      if (done_out) {
        int dn = lp->done_now;                                    // folded on the host ...
        const long long* const td = lp->t_dev;                     // ... unless the slot number lives on the device
        if (td) dn = ((unsigned int)(lp->t + *td) % (unsigned int)lp->episode_interval) == (unsigned int)lp->episode_interval - 1u;
        done_out[b] = (uint8_t)dn;
      }
    }
  };
  if (wave == 0) {
    if (!late_p2) p2_body(mypx, myact);
    if (live) p.pos_x[bN + lane] = mynpx;
  }
  if constexpr (RICH) {
    if (emit_chobs) {
      void* const chobs_out = ((LateRichArgs)(late_kernarg_base() + kRichArgOffset))->chobs_out;
      // channel observation [N][A] of this env, 16 bytes per lane, consecutive lanes on consecutive
      // pieces of a row (streaming: nothing on the chip reads it back); overlaps P3 of the other waves
      constexpr int CV = OUT64 ? 2 : 4;
      out_t* const co = static_cast<out_t*>(chobs_out) + bN * A;
      if ((A % CV) == 0) {
        const int qpr = A / CV, total = N * qpr;
        const int du = 256 / qpr, dq = 256 - du * qpr;
        int u = tid / qpr, qr = tid - u * qpr;
        for (int q = tid; q < total; q += 256) {
          const out_t* src = s_stage + u * SA + qr * CV;        // 16-byte aligned: SA % 4 == 0
          if constexpr (OUT64) stream_store2(co + 2 * q, *reinterpret_cast<const double2*>(src));
          else stream_store4(co + 4 * q, *reinterpret_cast<const float4*>(src));
          u += du; qr += dq;
          if (qr >= qpr) { qr -= qpr; u += 1; }
        }
      } else {
        for (int e = tid; e < N * A; e += 256) {
          const int u = e / A;
          stream_store(co + e, s_stage[u * SA + (e - u * A)]);
        }
      }
    }
  }
  int pol_action = 0, pol_cnt = 1;          // POL, wave 3: the agents' policy state, loaded here and used behind P3
  if constexpr (POL) {
    if (wave == 3 && live) {
      const LatePolArgs lq = (LatePolArgs)(late_kernarg_base() + kPolArgOffset);
      pol_action = lq->sps_prev[bN + lane];
      pol_cnt = lq->sps_counter[bN + lane];
    }
  }
  DIRAL_FSTAMP(3);

  // ---- P3a: stamp + gossip merge over this wave's 16 subject columns -------------
  //
  // The table is stored the way the merge wants it (round 3): per row-quad and viewer ONE word of four
  // thermometer codes c(lag) = (0xff << lag) & 0xff of the entries' lags behind their subjects' own sequence
  // numbers (0 = never heard), and one word of four ages.  In steady state an entry lags its subject by a handful
  // of slots (C2: <= 5, profiles/lag_distribution.py), so:
  //  * Vehicle.periodic_update (vehicle.py:56-70) is a shift: every subject takes a fresh number each slot, every
  //    lag grows by one - c' = (c << 1) & 0xfe per byte, two instructions per FOUR entries; the own entry becomes
  //    0xff / age 0; the ages take one packed saturating increment;
  //  * Vehicle.received_update for every (resource, rx), resources ascending: the codes are a chain under bit
  //    inclusion, the fresher entry's code is the bitwise OR - ONE ds_bpermute + ONE v_or per four columns and
  //    resource, no gather source to track (the alias of transmitted / live table, SURVEY Q1, is what makes the
  //    in-place form exact);
  //  * an entry's xpos is a function of (subject, sequence number) and the xpos ring keeps every subject's 8
  //    latest stamps: the xpos of EVERY coded entry is one lookup in the subject's ring row (two ds_bpermute from
  //    the lanes that hold it); the per-entry planes `tkey` / `tx` are neither read nor written;
  //  * what goes back to HBM is the merged code word and the age word (ages of updated entries cleared with a
  //    byte mask): 4 + 4 stores per lane.
  // An entry that reaches lag 7 hands over: its sequence number goes to `tkey`, its xpos to `tx`, and the quad is
  // flagged in `told`; from the next slot on (code 0, like a never-heard entry, but tkey.seq != 0) the quad takes
  // the KEYED path below - 32-bit keys (seq << 8) | source lane, sequence numbers rebuilt from the codes or read
  // from `tkey` - until the entry is refreshed.  Never-heard entries (seq 0) are code 0 with tkey.seq == 0; their
  // ghost xpos (0, or whatever was imported) lives in `tx`, their ages in the age words like everybody's.
  // (no resource in use - or no tables at all, `notab` - : no merge.  The test also keeps the backend alive: without a
  // use of the mask here the RICH instantiations die with "illegal VGPR to SGPR copy" - hipcc 7.2)
  const unsigned long long actw = notab ? 0ull : __ballot(lane < A && s_mask[lane < A ? lane : 0] != 0ull);
  const LateFastArgs lpr = (LateFastArgs)late_kernarg_base();
  // the xpos rings of this wave's 16 subjects: lane l holds slot l & 7 of subject l >> 3 (ringv0) / 8 + (l >> 3)
  // (ringv1); loaded here, first needed after the merge
  double* const ringp = lpr->ring + ((size_t)b * p.NR + (has_cols ? wave * 16 : 0)) * 8;
  double ringv0 = ringp[lane], ringv1 = ringp[64 + lane];
  // quads with an entry older than the codes reach (a straggler is usually ONE subject; sticky policies produce
  // them: profiles/lag_distribution*.py): bit q of `badq`, from the flags the previous slot left
  const unsigned int badq = has_cols ? (unsigned int)__ballot(told_l != 0u) & 15u : 0u;
  const int own_col = live ? lane - wave * 16 : -1;            // column of this lane's own entry, if in this wave
  const bool own_here = has_cols && (unsigned int)own_col < 16u;
  // fresh sequence numbers of the 16 subjects: lane c holds column c's (lanes >= 16 repeat them)
  const unsigned int tkov = ts_own + 1u;
  if (has_cols && lane < 16) tsq[lane] = tkov;
  // sequence-number overflow (24 bits: the keyed path packs (seq << 8) | lane)
  if (tkov >= (1u << 24) - 1u) atomicOr(lpr->err, kErrSeq);
  double st_px0, st_px1;                                        // this slot's stamps and their sequence numbers, ring layout
  unsigned int st_tk0, st_tk1;
  unsigned int cold[4];                                         // the codes after the stamp, before the merge
  const unsigned int own_byte = own_here ? (0xffu << (8 * (own_col & 3))) : 0u;   // this lane's own entry in its quad's words
  {
    // Vehicle.periodic_update (vehicle.py:56-70)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned int om = (own_col >> 2) == q ? own_byte : 0u;
      cw[q] = ((cw[q] << 1) & 0xfefefefeu) | om;               // every lag + 1; the own entry: lag 0
      // ages + 1, saturating at 255, four per word (no carry crosses a byte: the low 7 bits are added alone)
      const unsigned int a = ag[q];
      const unsigned int hi = a & 0x80808080u;
      const unsigned int lo = (a & 0x7f7f7f7fu) + 0x01010101u;
      const unsigned int sat = lo & hi;                        // 0x80 where the byte was 0xff
      ag[q] = ((lo ^ hi) | sat | (sat - (sat >> 7))) & ~om;    // own entry: age 0
      cold[q] = cw[q];
    }
    // this slot's stamps (vehicle.py:61-63: the subject's pre-move position under its fresh sequence number) into
    // the ring rows, in registers and in memory
    const int cl = lane >> 3;
    const unsigned int tk0 = (unsigned int)__builtin_amdgcn_ds_bpermute(cl << 2, (int)tkov);
    const unsigned int tk1 = (unsigned int)__builtin_amdgcn_ds_bpermute((cl + 8) << 2, (int)tkov);
    const int a0 = ((wave * 16 + cl) & 63) << 2, a1 = ((wave * 16 + cl + 8) & 63) << 2;
    const double px0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(a0, __double2hiint(mypx)),
                                        __builtin_amdgcn_ds_bpermute(a0, __double2loint(mypx)));
    const double px1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(a1, __double2hiint(mypx)),
                                        __builtin_amdgcn_ds_bpermute(a1, __double2loint(mypx)));
    const bool h0 = (unsigned int)(lane & 7) == (tk0 & 7u), h1 = (unsigned int)(lane & 7) == (tk1 & 7u);
    if (has_cols) {
      if (h0) ringp[lane] = px0;
      if (h1) ringp[64 + lane] = px1;
    }
    // (into the ring rows in registers: BEHIND the merge - `ring_stamp` - so that the merge does not wait for the rows)
    st_px0 = px0; st_px1 = px1; st_tk0 = tk0; st_tk1 = tk1;
  }
  auto ring_stamp = [&]() {
    const bool h0 = (unsigned int)(lane & 7) == (st_tk0 & 7u), h1 = (unsigned int)(lane & 7) == (st_tk1 & 7u);
    ringv0 = h0 ? st_px0 : ringv0;
    ringv1 = h1 ? st_px1 : ringv1;
    // ... and re-arranged by LAG for the lookups of P3b: lane l takes the stamp (l & 7) numbers behind the fresh
    // one of its subject, ring[k][(t_k - (l & 7)) & 7].  An entry's xpos is then the value of lane 8 (c & 7) + lag -
    // the lag straight from the code (count of trailing zeros), no sequence number, no per-column v_readlane
    const int r0 = (int)(((unsigned int)lane & 56u) | ((st_tk0 - (unsigned int)lane) & 7u)) << 2;
    const int r1 = (int)(((unsigned int)lane & 56u) | ((st_tk1 - (unsigned int)lane) & 7u)) << 2;
    ringv0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(r0, __double2hiint(ringv0)),
                              __builtin_amdgcn_ds_bpermute(r0, __double2loint(ringv0)));
    ringv1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(r1, __double2hiint(ringv1)),
                              __builtin_amdgcn_ds_bpermute(r1, __double2loint(ringv1)));
  };
  // Vehicle.received_update for the resources in ascending order: w[j] = comb(w[j], w[j] of lane
  // m_i / 4) for the resources i in use and the wave's four words j.  The gather sources of FOUR consecutive resources
  // arrive with one ds_read_b32 of this vehicle's table row (the next word a group ahead); a resource nobody transmits on
  // is skipped (its row is the identity; uniform test of the active mask).  SOFTWARE-PIPELINED by one step: the four
  // words are four independent dependency chains (gather, combine, gather ...), so a step first combines what the
  // previous step's gathers brought and then issues its own - word j's gather leaves as soon as ITS combine is done,
  // while the gathers of the other words are still in flight (LDS operations return in order): a step costs one LDS
  // round trip instead of four issue slots plus a round trip.  (`t` starts as the words themselves: comb(w, w) = w.)
  auto merge_walk = [&](unsigned int (&w)[4], auto&& comb) {
    const unsigned int* const mrow = reinterpret_cast<const unsigned int*>(s_mtab + lane * MS);   // MS % 4 == 0
    const int ng = A >> 2;
    unsigned int mw = mrow[0];
    unsigned long long act = actw;
    int t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = (int)w[j];
    auto step = [&](int m4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[j] = comb(w[j], (unsigned int)t[j]);
        t[j] = __builtin_amdgcn_ds_bpermute(m4, (int)w[j]);
        __builtin_amdgcn_sched_barrier(0);                       // (keep the chains interleaved: the scheduler would regroup)
      }
    };
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
      const unsigned int cur = mw;
      mw = mrow[g + 1];                                          // (word ng: the tail resources, or the row's padding)
      const unsigned int a4 = (unsigned int)act & 15u;
      act >>= 4;
      if (a4 & 1u) step((int)(cur & 255u));
      if (a4 & 2u) step((int)((cur >> 8) & 255u));
      if (a4 & 4u) step((int)((cur >> 16) & 255u));
      if (a4 & 8u) step((int)(cur >> 24));
    }
    for (int i = 0; i < (A & 3); ++i) {
      if ((unsigned int)act & (1u << i)) step((int)((mw >> (8 * i)) & 255u));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = comb(w[j], (unsigned int)t[j]);
  };
  if (badq != 15u) {
    // the coded merge (the words of a bad quad are merged along - meaningless, never read; none at all when every
    // quad of the wave is keyed: a sparse topology)
    if (actw != 0ull) merge_walk(cw, [](unsigned int a, unsigned int b) { return a | b; });
  }
  ring_stamp();
  DIRAL_FSTAMP(4);

  // ---- P3b: the entry's xpos, the table words, the histogram -----------------------
  unsigned int mycnt = 0u;
#ifdef DIRAL_TIMING
  unsigned int dbg_path = badq;          // which path the wave's quads took: bits 0-3 keyed, bits 4-7 the general column loop
#endif
  const double inv_w = p.inv_w;
  // inv_w * 2^20 (exact: the exponent field + 20) and K * 2^20: the fixed-point bin of the fast quads
  const double inv_w20 = __hiloint2double(__double2hiint(inv_w) + (20 << 20), __double2loint(inv_w));
  const unsigned int k20 = (unsigned int)K << 20;
  // float32 screening (FastParams::f32_m16): float32 copies of the lag-ordered ring rows and of the own position; off for
  // this wave when a stamp or a position lies outside the range the host's error bound was made for
  const int m16 = p.f32_m16;
  const float ringf0 = (float)ringv0, ringf1 = (float)ringv1, npxf = (float)mynpx;
  const bool f32_ok = FLAT && m16 > 0 &&
                      __ballot(!(__builtin_fabs(ringv0) <= (double)p.f32_xmax && __builtin_fabs(ringv1) <= (double)p.f32_xmax &&
                                 __builtin_fabs(mynpx) <= (double)p.f32_xmax)) == 0ull;
  const float invw16f = (float)(inv_w * 65536.0), halfk16f = (float)(p.Rb * inv_w * 65536.0);
  const unsigned int k16m = ((unsigned int)K << 16) + 2u * (unsigned int)m16;   // -m16 <= t16 < K * 2^16 + m16
  unsigned int* const hrow = s_hist + lane * KP;
  // Network.dist_piggy + get_positional_dist_2_piggy (network.py:538-558, 473-513) of one entry:
  // d = dist(entry, own post-move position), kept if d < Rb, value d * sign(x1 - x2)
  auto tally = [&](int k, bool heard, unsigned int age, double xg) {
    double d, v;
    if constexpr (FLAT) {
      // all y == 0: v = x1 - x2 IS d * sign exactly (fl(a-b) == -fl(b-a)), d = |v| - unless the square underflows
      // (0 < |v| < 2^-500), and even then the keep test agrees (both far below Rb) and so does the bin estimate
      // (v + Rb absorbs either): only the comparison with a bin edge at exactly 0 tells -tiny from the reference's
      // -0.  The exponent test therefore sits in the rare edge branch below, not in front of every entry.
      v = xg - mynpx;
      d = __builtin_fabs(v);
    } else {
      const double pyk = readlane_f64(mypy, k);
      d = fast_dist<false>(xg, heard ? pyk : 0.0, mynpx, mypy);
      v = (xg - mynpx > 0.0) ? d : -d;
    }
    const bool ok = live && (k < N) && (lane != k) && ((int)age < p.age_limit) && (d < p.Rb);
    if (ok) {
      // |v| < Rb, so the estimate is in [0, K] and the edge correction needs no bounds
      // tests: edges[0] = -Rb <= v and v < Rb = edges[K] hold by construction
      bool unsafe;
      int bin = hist_bin_estimate(v, p.Rb, inv_w, K, unsafe);      // (step_kernel.hpp: the edges are read only near an edge)
      if (unsafe) {
        bin = hist_bin_clamp(bin, K);
        if constexpr (FLAT) {
          if (((unsigned int)__double2hiint(v) & 0x7fffffffu) < 0x20b00000u) {   // |v| below 2^-500 (its square underflows) or 0
            d = dist_general(mynpx - xg, 0.0);
            v = (v > 0.0) ? d : -d;
          }
        }
        const double e0 = s_edges[bin], e1 = s_edges[bin + 1];
        bin += (v >= e1 ? 1 : 0) - (v < e0 ? 1 : 0);
      }
      atomicAdd(&hrow[bin], 1u);
      mycnt += 1u;
    }
  };
  // xpos of column c's entry that lags its subject by `lag` (0..7) stamps, from the lag-ordered ring rows
  auto ring_x = [&](double rv, int c, unsigned int lag) -> double {
    const int src = (int)((((unsigned int)c & 7u) << 3) | (lag & 7u)) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(rv)),
                            __builtin_amdgcn_ds_bpermute(src, __double2loint(rv)));
  };
  if (has_cols) {
    // -- coded quads: ages of updated entries cleared (a byte mask of the code bytes that changed), the two words
    //    stored, then column by column: xpos from the ring, hand-over at lag 7, histogram.
    //    (four rolled loops of four columns, one per packed word: a dynamically indexed word array would live
    //    in scratch memory, a register rotation costs a dozen moves per column)
    unsigned int handed = 0u;                                     // bit q: a lane of quad q handed an entry over
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((badq >> q) & 1u) continue;                              // (uniform)
      const unsigned int cnq = cw[q];
      unsigned int agq;
      {
        const unsigned int x = cnq ^ cold[q];
        const unsigned int nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // 0x80 per changed byte
        agq = ag[q] & ~(nz | (nz - (nz >> 7)));                  // a fresh copy has age 0
      }
      tc[q * NV + lane] = cnq;
      ta[q * NV + lane] = agq;
      const double rv = q >= 2 ? ringv1 : ringv0;
      if constexpr (FLAT) {
        // -- the FAST quad: no viewer holds a never-heard entry (code 0) or one at lag 7 (0x80: hand-over) about these
        //    four subjects - every quad of a dense topology in steady state.  Straight-line, four columns, no
        //    exec-mask branch:
        //    * the lag is the count of trailing zeros of the code byte, the xpos the lag-ordered ring value;
        //    * whether the entry counts - vehicle exists, not the own entry, age < limit - is ONE byte compare against
        //      the age word with the own byte (and every byte of a padded lane) forced to 255 (age_limit <= 255);
        //    * the bin in fixed point: ti = trunc((v + Rb) * inv_w * 2^20); 0 <= ti <= K * 2^20 is the range test
        //      |v| < Rb, ti >> 20 the bin, and when the 20 fraction bits are neither all 0 nor all 1 the value is at least
        //      2^-20 bin widths inside that bin (`hist_bin_estimate`: the estimate is within 2^-45 of exact) - else
        //      (an edge hit, |v| at Rb) the lane takes the exact statement below, behind a uniform branch;
        //    * an entry that does not count increments the spare slot K of the row.
        const unsigned int y7 = cnq & 0x7f7f7f7fu;
        const bool rare = live && ((y7 + 0x7f7f7f7fu) & 0x80808080u) != 0x80808080u;   // a byte with (code & 0x7f) == 0
#ifdef DIRAL_TIMING
        if (__ballot(rare) != 0ull) dbg_path |= 16u << q;
#endif
        if (__ballot(rare) == 0ull) {
          const unsigned int agt = live ? (agq | ((own_col >> 2) == q ? own_byte : 0u)) : 0xffffffffu;
          // (float32 screening, below) which of this lane's entries the float64 body counts: all of them when the
          // screening is off, else those whose float32 fraction fell into the band - recomputed here, the pass is rare
          auto band32 = [&](int cc) -> bool {
            if (!f32_ok) return true;
            const int src = ((int)((__builtin_ctz((cnq >> (8 * cc)) | 0x100u)) & 7) << 2) + (((4 * q + cc) & 7) << 5);
            const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(q >= 2 ? ringf1 : ringf0)));
            const int t16 = cvt_i32_f32_sat(__builtin_fmaf(xf - npxf, invw16f, halfk16f));
            return (unsigned int)(t16 + m16) < k16m && (((unsigned int)(t16 - m16)) & 0xffffu) >= 65536u - 2u * (unsigned int)m16;
          };
          auto column = [&](auto cc_tag) {
            constexpr int cc = decltype(cc_tag)::value;
            const int c = 4 * q + cc;
            const int src = (ffbl_byte<cc>(cnq) << 2) + ((c & 7) << 5);
            const double xg = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(rv)),
                                               __builtin_amdgcn_ds_bpermute(src, __double2loint(rv)));
            const double v = xg - mynpx;
            const int ti = cvt_i32_f64_sat((v + p.Rb) * inv_w20);
            const bool agev = band32(cc) && ((agt >> (8 * cc)) & 255u) < (unsigned int)p.age_limit;
            const bool in = (unsigned int)ti <= k20;
            const bool edge = ((unsigned int)(ti + 1) & 0xfffffu) <= 1u;
            const bool m = agev && in;
            const bool need = m && edge;
            bool cnt = m && !need;
            int bin = ti >> 20;
            if (need) {                                            // (rare: an exact edge hit, |v| at Rb)
              double vv = v;
              cnt = __builtin_fabs(vv) < p.Rb;
              bin = 0;
              if (cnt) {
                bool unsafe;
                bin = hist_bin_clamp(hist_bin_estimate(vv, p.Rb, inv_w, K, unsafe), K);
                if (((unsigned int)__double2hiint(vv) & 0x7fffffffu) < 0x20b00000u) {   // |v| below 2^-500 (its square underflows) or 0
                  const double d = dist_general(mynpx - xg, 0.0);
                  vv = (vv > 0.0) ? d : -d;
                }
                const double e0 = s_edges[bin], e1 = s_edges[bin + 1];
                bin += (vv >= e1 ? 1 : 0) - (vv < e0 ? 1 : 0);
              }
            }
#ifdef DIRAL_DEBUG_XG
            if (p.dbg) p.dbg[((size_t)b * 64 + wave * 16 + c) * 64 + lane] = (unsigned long long)__double_as_longlong(xg);
#endif
            atomicAdd(&hrow[cnt ? bin : K], 1u);
            mycnt += cnt ? 1u : 0u;
          };
          // float32 first (FastParams::f32_m16): ONE gather and three float32 operations per entry; t16 = (v + Rb) *
          // inv_w * 2^16 truncated - its integer part is the bin, and |v| < Rb is 0 <= t16 < K * 2^16, unless the 16 fraction bits
          // are within m16 of an integer (everything float32 lost is below m16 / 65536 bin widths).  Lanes inside that
          // band are left for the float64 body, which then runs once more for the quad (a uniform test: rare)
          bool redo = false;                                        // an entry of this lane is left for float64
          auto column32 = [&](auto cc_tag) {
            constexpr int cc = decltype(cc_tag)::value;
            const int c = 4 * q + cc;
            const int src = (ffbl_byte<cc>(cnq) << 2) + ((c & 7) << 5);
            const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(q >= 2 ? ringf1 : ringf0)));
            const int t16 = cvt_i32_f32_sat(__builtin_fmaf(xf - npxf, invw16f, halfk16f));
            const bool agev = ((agt >> (8 * cc)) & 255u) < (unsigned int)p.age_limit;
            const bool mm = agev && (unsigned int)(t16 + m16) < k16m;
            const bool band = (((unsigned int)(t16 - m16)) & 0xffffu) >= 65536u - 2u * (unsigned int)m16;
            const bool cnt = mm && !band;
            redo = redo || (mm && band);
            atomicAdd(&hrow[cnt ? (t16 >> 16) : K], 1u);
            mycnt += cnt ? 1u : 0u;
          };
          if (f32_ok) {                                             // (uniform)
            column32(std::integral_constant<int, 0>{});
            column32(std::integral_constant<int, 1>{});
            column32(std::integral_constant<int, 2>{});
            column32(std::integral_constant<int, 3>{});
            if (__ballot(redo) == 0ull) continue;
          }
          column(std::integral_constant<int, 0>{});
          column(std::integral_constant<int, 1>{});
          column(std::integral_constant<int, 2>{});
          column(std::integral_constant<int, 3>{});
          continue;
        }
      }
      bool hand = false;
#pragma unroll 1
      for (int cc = 0; cc < 4; ++cc) {
        const int c = 4 * q + cc;
        const int k = wave * 16 + c;
        const unsigned int sh = 8u * (unsigned int)cc;
        const unsigned int rf = (cnq >> sh) & 255u, age = (agq >> sh) & 255u;
        // sequence number back from the code: lag = 8 - popcount (the subject's own number - 8 rides in the add
        // operand of v_bcnt)
        const unsigned int tk8 = (unsigned int)__builtin_amdgcn_readlane((int)tkov, c) - 8u;
        unsigned int seqn;
        asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(seqn) : "v"(rf), "s"(tk8));   // popcount(rf) + tk8
        double xg = ring_x(rv, c, tk8 - seqn);                      // lag = t_k - seq (tk8 = t_k mod 8)
        if ((rf & 0x7fu) == 0u) {                                   // 0: never heard; 0x80: lag 7 - both rare, one test
          // (the plane row through a scalar base + lane offset, formed here: per-lane 64-bit pointers stepping from
          // column to column cost the common path two VALU instructions per column)
          const global_ptr<double> txrow = uniform_ptr(p.tx, ((size_t)b * p.NR + k) * NV);
          if (rf == 0u) {                                           // never heard: the ghost xpos lives in the plane
            xg = txrow[(unsigned int)lane];
            // (the load is consumed INSIDE the branch: left to the compiler, its s_waitcnt vmcnt(0) lands behind the join
            // and every column - taken or not - then waits for the table stores issued just before the loop)
            asm volatile("" : "+v"(xg));
          } else {                                                  // lag 7: from the next slot on beyond the codes
            const global_ptr<unsigned int> tkrow = uniform_ptr(p.tkey, ((size_t)b * p.NR + k) * NV);
            txrow[(unsigned int)lane] = xg;
            tkrow[(unsigned int)lane] = (seqn << 8) | age;
            hand = true;
          }
        }
#ifdef DIRAL_DEBUG_XG
        if (p.dbg) p.dbg[((size_t)b * 64 + k) * 64 + lane] = (unsigned long long)__double_as_longlong(xg);
#endif
        tally(k, rf != 0u, age, xg);
      }
      handed |= (__ballot(hand) != 0ull ? 1u : 0u) << q;
    }
    // -- keyed quads: 32-bit keys (seq << 8) | source lane.  Sequence numbers from the (stamped) codes, or from
    //    `tkey` where the code is 0; one ds_bpermute + max per column and resource; then column by column the
    //    xpos (young entries' from the ring, old ones' from the plane) through ONE gather from the recorded
    //    source, and the entry goes back coded if it came back within reach of the codes, else into `tkey`.
    //    The plane `tx` and `tkey` are left complete for these columns.  (Rolled over the quads with the words
    //    rotating through element 0: rare, and four unrolled copies cost the coded path registers.)
    unsigned int still = 0u;                                      // bit q: quad q keeps an entry beyond the codes
    if (badq) {
      unsigned int rc0 = cold[0], rc1 = cold[1], rc2 = cold[2], rc3 = cold[3];
      unsigned int ra0 = ag[0], ra1 = ag[1], ra2 = ag[2], ra3 = ag[3];
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        if ((badq >> q) & 1u) {
          const double rv = q >= 2 ? ringv1 : ringv0;
          unsigned int kq[4], k0[4];
          double xp4[4];                            // (all eight loads in flight together: one round trip, not four)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            const unsigned int wr = tk[c * NV + lane];
            xp4[i] = txp[c * NV + lane];
            const unsigned int r = (rc0 >> (8 * i)) & 255u;
            const unsigned int tk_own = (unsigned int)__builtin_amdgcn_readlane((int)tkov, c);
            const unsigned int seq = r ? tk_own - 8u + (unsigned int)__popc(r) : (wr >> 8);
            k0[i] = kq[i] = (seq << 8) | (unsigned int)lane;
          }
          merge_walk(kq, [](unsigned int a, unsigned int b) { return a > b ? a : b; });
          unsigned int ncode = 0u, nage = 0u;
          bool keep = false;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            const int k = wave * 16 + c;
            const int off = c * NV + lane;
            const unsigned int r = (rc0 >> (8 * i)) & 255u, a0 = (ra0 >> (8 * i)) & 255u;
            const unsigned int tk_own = (unsigned int)__builtin_amdgcn_readlane((int)tkov, c);
            const unsigned int seq0 = k0[i] >> 8, seqf = kq[i] >> 8;
            const bool upd = seqf != seq0;
            // a coded entry's xpos is in the ring, not (necessarily) in the plane.  (The lookup is unconditional: under a
            // branch the ds_bpermute would run with the code-0 lanes switched off - and read 0 from them.)
            const double xr = ring_x(rv, c, tk_own - seq0);
            const double x_cur = r ? xr : xp4[i];
            const double xs = (lane == k) ? mypx : x_cur;             // own stamp (vehicle.py:63)
            const int src4 = (int)(kq[i] & 255u) << 2;
            const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(xs));
            const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(xs));
            const double xg = upd ? __hiloint2double(hi, lo) : xs;
            const unsigned int age = upd ? 0u : a0;                   // (the own entry's age is 0 since the stamp)
            const unsigned int lagf = tk_own - seqf;
            const bool coded = seqf != 0u && lagf <= 7u;
            ncode |= (coded ? ((0xffu << lagf) & 0xffu) : 0u) << (8 * i);
            nage |= age << (8 * i);
            keep = keep || (seqf != 0u && lagf >= 7u);                // beyond the codes from the next slot on
            tk[off] = (seqf << 8) | age;
            txp[off] = xg;
#ifdef DIRAL_DEBUG_XG
            if (p.dbg) p.dbg[((size_t)b * 64 + k) * 64 + lane] = (unsigned long long)__double_as_longlong(-xg - 1e6);
#endif
            tally(k, seqf != 0u, age, xg);
          }
          tc[q * NV + lane] = ncode;
          ta[q * NV + lane] = nage;
          still |= (__ballot(keep) != 0ull ? 1u : 0u) << q;
        }
        rc0 = rc1; rc1 = rc2; rc2 = rc3;
        ra0 = ra1; ra1 = ra2; ra2 = ra3;
      }
    }
    // the flags of the next slot: quads that handed an entry over, keyed quads that still hold one
    const unsigned int newf = (handed & ~badq) | still;
    if (newf != badq || handed) {
      if (lane < 4) lpr->told[qbase + lane] = (newf >> lane) & 1u;
    }
    if (newf && lane == 0) s_slow[0] = 1u;                        // (zeroed in P0, two barriers ago)
  }
  if (mycnt) atomicAdd(&s_cnt[lane], mycnt);
  if constexpr (POL) {
    // the SPS agents' decisions for the next slot (algorithms/v2x_sps.py:76-104): wave 3, in the time it would
    // otherwise wait at the barrier for wave 0 (which did P2); the staging array is complete since the P1 barrier
    if (wave == 3) {
      const PolParams* const qp = reinterpret_cast<const PolParams*>(late_kernarg_base() + kPolArgOffset);
      fast_sps_decide<out_t>(s_stage, SA, A, N, bN, lane, s_act[lane], pol_action, pol_cnt, qp);
    }
  }
#ifdef DIRAL_TIMING
  if (lane == 0 && p.dbg) p.dbg[(size_t)p.B * 48 + (size_t)b * 4 + wave] = dbg_path;
#endif
  DIRAL_FSTAMP(5);
  __syncthreads();
  DIRAL_FSTAMP(6);
  if (tid == 0) {
    // the next launch's order: a slow env asks for a place among the first blocks
    const LateFastArgs ls = (LateFastArgs)late_kernarg_base();
    uint32_t* const flag_w = ls->slow_flag_w;
    if (flag_w) {
      unsigned int fl = 0u;
      if (s_slow[0]) {
        const unsigned int pos = atomicAdd(ls->slow_cnt_w, 1u);
        if (pos < (unsigned int)fast_slow_max(ls->B)) { ls->slow_list_w[pos] = (unsigned int)b; fl = 1u; }
      }
      flag_w[b] = fl;
      // the set the launch after the next builds: count AND flags (a cleared count under standing flags would make a
      // launch that still reads this set - a graph captured two launches ago - skip the flagged envs)
      uint32_t* const set_z = ls->slow_cnt_z;
      set_z[16 + fast_slow_max(ls->B) + b] = 0u;
      if (b == 0) *set_z = 0u;
    }
  }

  if constexpr (POL) {
    // the driver's reward shaping (main_test.py:171, 178, 194-206; diral_driver_shape without the information-age
    // terms) of this env: wave 1, lane = vehicle, from the rewards P2 left in LDS - np.sum in NumPy's order
    if (wave == 1) {
      const LatePolArgs lq = (LatePolArgs)(late_kernarg_base() + kPolArgOffset);
      void* const shaped_out = lq->shaped_out;
      if (shaped_out) {
        const int sflags = lq->shape_flags;
        const out_t a = live ? (out_t)s_rew[lane] : (out_t)0;
        const out_t sr = np_row_sum_wave(a, N, lane);
        if (lane == 0) {
          void* const so = lq->sum_r_out;
          void* const co = lq->coll_out;
          if (so) static_cast<out_t*>(so)[b] = sr;
          if (co) static_cast<out_t*>(co)[b] = (out_t)A - sr;
        }
        if (live) {
          out_t rr = a;
          if (sflags & 4) {
            int32_t* const pc = lq->pen_counter;
            int32_t* const pp = lq->pen_prev;
            const int ac = s_act[lane];
            const bool stuck = (rr < (out_t)1) && (ac == pp[bN + lane]);
            const int c = stuck ? pc[bN + lane] + 1 : 0;
            pc[bN + lane] = c;
            if (c > lq->pen_threshold) rr = (out_t)lq->pen_value;
            pp[bN + lane] = ac;
          }
          if (sflags & 1) rr = rr + sr / (out_t)N;
          static_cast<out_t*>(shaped_out)[bN + lane] = rr;
        }
      }
    }
  }

  if (late_p2 && wave == 0) {                // (uniform) P2, while the other waves write the state vectors
    p2_body(s_px[lane], s_act[lane]);
    DIRAL_FSTAMP(7);
    return;
  }
  // ---- P4: state = [one-hot(action) (A) | histogram (K)] ---------------------------
  // float64 (the reference's dtype, h/n as one IEEE division - np.histogram counts
  // divided by the neighbour count) or float32 (= the float32 cast of that value).
  // (by the waves P2 leaves free: all four, or waves 1-3 with `late_p2`)
  const int T4 = late_p2 ? 192 : 256, t4 = late_p2 ? tid - 64 : tid;
  const unsigned long long late = late_kernarg_base();             // state_out and the RICH section layout: from here on
  void* const state_out = ((LateFastArgs)late)->state_out;
  if constexpr (RICH) {
    const RichParams rr = load_rich_args(late);
    // the channel-observation SECTION of the state vector (State.add_channel_obs): the same
    // values P1 stored to chobs_out, rebuilt from the gather sources (the same distance
    // expression P1 evaluated)
    auto chv = [&](int u, int i) -> double {
      if (s_act[u] == i || s_mask[i] == 0ull) return 0.0;
      if (!dist_obs) return 1.0;
      const int src = s_mtab[u * MS + i] >> 2;
      if (src == u) return 100000.0;
      return fast_dist<FLAT>(s_px[src], FLAT ? 0.0 : s_py[src], s_px[u], FLAT ? 0.0 : s_py[u]);
    };
    if (state_out && !rr.plain_state) {
      rich_write_state<OUT64>(
          rr, p.flags, N, A, K, p.L, state_out, bN, tid, 256, [&](int u) { return s_act[u]; }, chv,
          [&](int u, int bin) {
            const unsigned int n = s_cnt[u];
            return n ? (double)s_hist[u * KP + bin] / (double)n : 0.0;
          },
          [&](int u) { return s_rew[u]; }, [&](int u) { return s_npx[u]; },
          [&](int u) { return FLAT ? 0.0 : s_py[u]; }, [&](int u) { return rr.vel[bN + u]; });
    }
    // the reference's call pattern on the toy YAML's flags (my_step* with `obs` + obtain_state):
    // channel observation above, the plain state vector by the vectorised writer below
    if (!(state_out && rr.plain_state)) { DIRAL_FSTAMP(7); return; }
  }
  const int S = A + K;
  if constexpr (OUT64) {
    double* out = static_cast<double*>(state_out) + bN * S;
    if (((A | K) & 1) == 0) {
      const int q_per_row = S >> 1, total = N * q_per_row;
      const int du = T4 / q_per_row, dq = T4 - du * q_per_row;
      int u = t4 / q_per_row, qr = t4 - u * q_per_row;
      for (int q = t4; q < total; q += T4) {
        const int s0 = qr << 1;
        double2 v;
        if (s0 < A) {
          const int a = s_act[u] - s0;
          v = make_double2(a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0);
        } else {
          const unsigned int n = s_cnt[u];
          const unsigned int* h = s_hist + u * KP + (s0 - A);
          const double dn = (double)n;
          v = n ? make_double2((double)h[0] / dn, (double)h[1] / dn) : make_double2(0.0, 0.0);
        }
        stream_store2(out + 2 * q, v);
        u += du; qr += dq;
        if (qr >= q_per_row) { qr -= q_per_row; u += 1; }
      }
    } else {
      for (int e = t4; e < N * S; e += T4) {
        const int u = e / S, s = e - u * S;
        double val;
        if (s < A) val = (s_act[u] == s) ? 1.0 : 0.0;
        else {
          const unsigned int n = s_cnt[u];
          val = n ? (double)s_hist[u * KP + (s - A)] / (double)n : 0.0;
        }
        stream_store(out + e, val);
      }
    }
  } else {
  float* out = static_cast<float*>(state_out) + bN * S;
  const double* const inv_tab = reinterpret_cast<const double*>(smem + lay.inv);
  if (((A | K) & 3) == 0) {
    const int q_per_row = S >> 2, total = N * q_per_row;
    // (row, quad) advance incrementally: one integer division per thread instead of one per store
    const int du = T4 / q_per_row, dq = T4 - du * q_per_row;
    int u = t4 / q_per_row, qr = t4 - u * q_per_row;
    for (int q = t4; q < total; q += T4) {
      const int s0 = qr << 2;
      float4 v;
      if (s0 < A) {
        const int a = s_act[u] - s0;
        v = make_float4(a == 0 ? 1.f : 0.f, a == 1 ? 1.f : 0.f, a == 2 ? 1.f : 0.f, a == 3 ? 1.f : 0.f);
      } else {
        const unsigned int n = s_cnt[u];
        const unsigned int* h = s_hist + u * KP + (s0 - A);
        // (float)((double)h * fl(1.0 / n)) == (float)((double)h / (double)n) == the correctly rounded float
        // quotient for every 0 <= h <= n <= 255 (checked exhaustively; the quotient of two small integers
        // is never within 2^-50 of a float rounding boundary): one table load instead of four IEEE divisions
        const double inv = inv_tab[n];
        v = make_float4((float)((double)h[0] * inv), (float)((double)h[1] * inv), (float)((double)h[2] * inv),
                        (float)((double)h[3] * inv));
      }
      stream_store4(out + 4 * q, v);
      u += du; qr += dq;
      if (qr >= q_per_row) { qr -= q_per_row; u += 1; }
    }
  } else {
    for (int e = t4; e < N * S; e += T4) {
      const int u = e / S, s = e - u * S;
      float val;
      if (s < A) val = (s_act[u] == s) ? 1.f : 0.f;
      else {
        const unsigned int n = s_cnt[u];
        val = n ? __fdiv_rn((float)s_hist[u * KP + (s - A)], (float)n) : 0.f;
      }
      stream_store(out + e, val);
    }
  }
  }
  DIRAL_FSTAMP(7);
}

}  // namespace diral
