// policy_device.hpp - device functions shared by the stand-alone policy / shaping kernels (aux_kernels.hpp) and the
// policy epilogue of the fused step (step_fast64.hpp, POL instantiations): the counter-based generator, the SPS agent
// (algorithms/v2x_sps.py:8-104), np.sum's summation order walked by one wave.  ONE statement of each, so that the fused
// slot and the three-launch slot decide and round identically.
#pragma once
#include "common.hpp"

namespace diral {

// Third kernel argument of step_fast64: the policy epilogue of the POL instantiations (diral_env_step_policy,
// include/diral_env.h DiralSlotPolicy) - reward shaping (main_test.py:171-206 without the information-age terms) and
// the SPS agent's decision (algorithms/v2x_sps.py:76-104) from the channel observation the step has staged in LDS.
struct PolParams {
  int shape_flags;               // bit 0: global_reward_avg; bit 2: stuck-action penalty (diral_driver_shape's flags)
  int pen_threshold;
  double pen_value;
  void* shaped_out;              // [B][N] out dtype, or null: no shaping
  void* sum_r_out;               // [B] or null
  void* coll_out;                // [B] or null
  int32_t* pen_counter;          // [B][N] (flag bit 2)
  int32_t* pen_prev;             // [B][N]
  int32_t* sps_prev;             // [B][N] SemiPersistentScheduling.prev_action
  int32_t* sps_counter;          // [B][N] .reselection_counter
  double threshold, inc_db, keep_prob;
  const int32_t* draw_counter;   // injected draws or null (device generator from `seed`)
  const double* draw_keep;
  const int32_t* draw_choice;
  uint64_t seed;
  const long long* clock;        // null, or a device counter added to the seed (captured rollouts)
  int32_t* actions_out;          // [B][N] the next slot's actions
  // K slots in ONE launch (diral_env_step_policy with DiralSlotPolicy::slots > 1): the workgroup keeps its env - code
  // and age words, ring rows, positions, velocities, the agents' policy state - in registers and LDS from slot to slot
  int K;                         // slots of this launch (>= 1)
  int vel_vary;                  // mobility_vary: Network.update_velocity (network.py:208-223) at every episode end inside the launch
  uint64_t vel_seed;             // ... device draws seeded vel_seed + episode index, as diral_env_update_velocity(env, NULL, seed)
  uint64_t idx0;                 // global index of agent 0 of env 0 (DIRAL_OPT_ENV_OFFSET * N): the draws are indexed globally
  double* vel_w;                 // [B][N] velocities (written back behind the last slot when vel_vary)
  // The driver's random prefill (main_test.py:99-114: sample -> my_step_design -> obtain_state, every state kept) as the K
  // slots of one launch (diral_env_prefill): the agents of slot ks + 1 act at random - diral_env_sample(seed + ks + 1) -,
  // the reward is my_step_design's (network.py:122-157; P2), every slot's state vector leaves ([K][B][N][S]), its reward
  // column taken from `rew_in` (the bootstrap step's rewards: main_test.py:110 hands obtain_state the stale `rews`)
  int prefill;
  int32_t* actions_all;          // [K][B][N] the actions slot ks ran with, or null
  const double* rew_in;          // [B][N] or null (null: the slot's own reward)
};

// counter-based generator (splitmix64 finaliser over seed/stream/index); the
// reference uses unseeded global RNGs, so only the distribution matters.
__device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ inline uint64_t rng_u64(uint64_t seed, uint64_t stream, uint64_t idx) {
  return mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + idx);
}
__device__ inline double rng_unit(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }


// SemiPersistentScheduling.step (algorithms/v2x_sps.py:76-104); one thread per agent.
// Returns true when the agent has to choose a new resource (then `cnt` is already redrawn).
__device__ inline bool sps_advance(int i, int& cnt, double keep_prob, const int32_t* draw_counter,
                                   const double* draw_keep, uint64_t seed) {
  if (cnt != 0) { cnt -= 1; return false; }                        // v2x_sps.py:85-89
  cnt = draw_counter ? draw_counter[i] : 5 + (int)(rng_u64(seed, 7, (uint64_t)i) % 12ull);   // randint(5, 16)
  const double u = draw_keep ? draw_keep[i] : rng_unit(rng_u64(seed, 8, (uint64_t)i));
  return !(u < keep_prob);                                         // v2x_sps.py:93-98
}


// Build extension (the reference never wires SPS to the toy env): an RSSI-like selection window
// from the toy env's type-2 channel observation `obs[user][i]` (test_env.py:206, 240,
// network.py:385): distance d to the nearest in-range transmitter -> log-distance path loss
// -40 - 30 log10(max(d, 1)) dB; 100000 (busy, nobody in range) -> -160; 0 (idle) -> -200; the
// agent's own resource reads as busy (-60).  Lower = quieter.
__device__ inline double sps_rssi_from_chobs(double d, bool own) {
  if (own) return -60.0;
  if (d >= 100000.0) return -160.0;
  if (d > 0.0) return -40.0 - 30.0 * log10(d < 1.0 ? 1.0 : d);
  return -200.0;
}


// ---- wave-cooperative SPS step (A <= 256) ------------------------------------------------
// Re-selection is rare (counter expiry x 20 % = 1.7 % of the agents per slot), but with one
// thread per agent almost every wave holds one such lane and then runs at the speed of that
// lane's serial window scan.  Here the 64 lanes of the wave serve each of their re-selecting
// agents together: lane s holds subframe s (+64c) of that agent's window - one coalesced row
// read, one log10 per lane - the threshold loop is a ballot + popcount, and the stable-sort
// position of every candidate is counted against the candidates' values read lane by lane.
constexpr int kSpsWaveMaxA = 256;

__device__ inline double sps_readlane(double v, int j) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}

// choose_new_resource (algorithms/v2x_sps.py:24-74) for ONE agent by the whole wave; every argument
// but `w` / `lane` is wave-uniform, and so is the result.
template <int NC>
__device__ inline int sps_choose_wave(const double (&w)[NC], int lane, int A, int prev, double threshold, double inc_db,
                                      unsigned int r) {
  const double min_sA = (double)A / 5.0;                           // len(selection_window)/5
  double thr_next = threshold, thr = threshold;
  unsigned long long el[NC];                                       // sA: candidates of chunk c
  int n_sa = 0;
  for (int it = 0; it < 100000; ++it) {                            // while len(sA) < min_sA
    thr = thr_next;
    n_sa = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int s = lane + 64 * c;
      el[c] = __ballot(s < A && s != prev && w[c] < thr);
      n_sa += __popcll(el[c]);
    }
    thr_next = thr + inc_db;                                       // tmp_threshold += self.inc_dB
    if (!((double)n_sa < min_sA)) break;
  }
  const double min_len = min_sA < (double)n_sa ? min_sA : (double)n_sa;
  int need = (int)min_len;
  if ((double)need < min_len) need += 1;                           // sB grows until len(sB) >= min_len
  if (need < 1) need = 1;
  const int pick = (int)(r % (unsigned int)need);                  // random.choice(sB)
  // position of every candidate in sorted(sA.items(), key=value): a stable sort, i.e. ordered by
  // (value, subframe)
  int rank[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) rank[c] = 0;
#pragma unroll
  for (int cj = 0; cj < NC; ++cj) {
    unsigned long long m = el[cj];
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      const double wj = sps_readlane(w[cj], j);
      const int sj = j + 64 * cj;
#pragma unroll
      for (int c = 0; c < NC; ++c) rank[c] += (wj < w[c] || (wj == w[c] && sj < lane + 64 * c)) ? 1 : 0;
    }
  }
  int chosen = prev;                                               // (only if sA stayed empty: cannot happen)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const unsigned long long hit = __ballot(((el[c] >> lane) & 1ull) && rank[c] == pick);
    if (hit) chosen = __builtin_ctzll(hit) + 64 * c;
  }
  return chosen;
}


// choose_new_resource for ONE agent by the whole wave, straight from its row of the channel observation (`d[c]`:
// subframe lane + 64 c) - the same decision as sps_choose_wave on the window sps_rssi_from_chobs builds from it, bit for
// bit, usually without a single log10: idle subframes read -200 and busy ones without a transmitter in range -160
// whatever the distances are, and a heard one reads above -160 as long as its transmitter is closer than 5 km (then
// -40 - 30 log10(d) > -151).  If those unheard subframes alone hold ceil(A / 5) candidates below the threshold, the
// 3 dB loop stops at its first pass, `need` is ceil(A / 5), and the first `need` places of the stable sort by
// (value, subframe) are unheard subframes - the idle ones in subframe order, then the out-of-range ones: the pick is
// found on the two ballot masks.  Anything else (few unheard subframes, a far transmitter, a threshold below -160)
// takes the general path.
template <int NC>
__device__ inline int sps_choose_chobs_wave(const double (&d)[NC], int lane, int A, int prev, int own, double threshold,
                                            double inc_db, unsigned int r) {
  unsigned long long m_idle[NC], m_oor[NC];
  bool far = false;
  int n0 = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = lane + 64 * c;
    const bool cand = s < A && s != own && s != prev;
    const bool oor = d[c] >= 100000.0, heard = !oor && d[c] > 0.0;
    m_idle[c] = __ballot(cand && !oor && !heard && -200.0 < threshold);
    m_oor[c] = __ballot(cand && oor && -160.0 < threshold);
    far = far || (s < A && s != own && heard && !(d[c] <= 5000.0));
    n0 += __popcll(m_idle[c]) + __popcll(m_oor[c]);
  }
  const double min_sA = (double)A / 5.0;                           // len(selection_window)/5
  int need = (int)min_sA;
  if ((double)need < min_sA) need += 1;
  if (need < 1) need = 1;
  if (n0 >= need && __ballot(far) == 0ull) {
    int pick = (int)(r % (unsigned int)need);                      // random.choice(sB)
    // the pick-th candidate in (value, subframe) order: idle subframes ascending, then out-of-range ones
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        unsigned long long m = pass == 0 ? m_idle[c] : m_oor[c];
        const int n = __popcll(m);
        if (pick < n) {
          for (int k = 0; k < pick; ++k) m &= m - 1;
          return __builtin_ctzll(m) + 64 * c;
        }
        pick -= n;
      }
    }
  }
  double w[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = lane + 64 * c;
    w[c] = 0.0;
    if (s < A) w[c] = sps_rssi_from_chobs(d[c], s == own);
  }
  return sps_choose_wave<NC>(w, lane, A, prev, threshold, inc_db, r);
}

template <typename T>
__device__ inline T shfl_t(T v, int src) {
  if constexpr (sizeof(T) == 8) {
    const double d = (double)v;
    return (T)__hiloint2double(__shfl(__double2hiint(d), src), __shfl(__double2loint(d), src));
  } else {
    return __shfl(v, src);
  }
}

// np.sum of one row held one element per lane (N <= 64, lanes >= N hold 0) in NumPy's order (numpy/_core/src/umath/
// loops_utils.h pairwise_sum, n <= 128: eight running accumulators over the blocks of eight, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then a sequential tail; fewer than 8 elements: sequentially), walked with lane
// shuffles; wave-uniform result.  Float addition is not associative: main_test.py:171, 205-206 are bit-identical only
// in this order.
template <typename T>
__device__ inline T np_row_sum_wave(T a, int N, int lane) {
  // r[j] = a[j] + a[8 + j] + a[16 + j] + ... (in that order) for j = 0..7, over the whole blocks of eight
  const int nb = N >> 3;
  T r = a;
  for (int i = 1; i < nb; ++i) {
    const T v = shfl_t(a, (lane & 7) + 8 * i);
    r = r + v;
  }
  // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)): a shuffle-down tree over lanes 0..7
  T t = r + shfl_t(r, lane + 1);
  t = t + shfl_t(t, lane + 2);
  t = t + shfl_t(t, lane + 4);
  T sr = shfl_t(t, 0);
  for (int i = nb * 8; i < N; ++i) sr = sr + shfl_t(a, i);          // the sequential tail
  return sr;
}

}  // namespace diral
