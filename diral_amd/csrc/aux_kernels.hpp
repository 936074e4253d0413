// aux_kernels.hpp - the small kernels around the step: topology reset, action
// sampling, velocity update, state export/import, information-age histogram,
// metric read-out.  All are elementwise / tiny; none is on the hot path.
#pragma once
#include "common.hpp"
#include "policy_device.hpp"

namespace diral {

// Network.initialize_mobility_topology (network.py:92-119): x = randint(0, L)
// (integer valued), y = randint(0, H/2) = 0, v = 1.7 if mobility_vary else
// uniform(1.1, 2.7); any of x0/y0/v0 given => copied instead.
// Device draws are indexed by the GLOBAL vehicle index idx0 + i (idx0 = env offset of this
// handle x N, DIRAL_OPT_ENV_OFFSET): a batch sharded over several handles / GPUs draws
// exactly what one handle holding the whole batch draws.
__global__ void reset_kernel(int total, double L, int vary, uint64_t seed, uint64_t idx0, const double* x0,
                             const double* y0, const double* v0, double* pos_x, double* pos_y,
                             double* vel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  double x, y, v;
  if (x0) x = x0[i];
  else {
    const double Lf = floor(L);
    x = floor(rng_unit(rng_u64(seed, 1, idx0 + (uint64_t)i)) * Lf);
    if (x >= Lf) x = Lf - 1.0;
  }
  y = y0 ? y0[i] : 0.0;
  if (v0) v = v0[i];
  else v = vary ? 1.7 : 1.1 + rng_unit(rng_u64(seed, 2, idx0 + (uint64_t)i)) * (2.7 - 1.1);
  pos_x[i] = x; pos_y[i] = y; vel[i] = v;
}

// TestEnv.sample (test_env.py:116-122)
__global__ void sample_kernel(int total, int A, uint64_t seed, uint64_t idx0, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  out[i] = (int32_t)(rng_u64(seed, 3, idx0 + (uint64_t)i) % (uint64_t)A);
}

// Network.update_velocity (network.py:208-223)
__global__ void velocity_kernel(int total, const uint8_t* draws, uint64_t seed, uint64_t idx0, double* vel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r = draws ? (int)draws[i] : 1 + (int)(rng_u64(seed, 4, idx0 + (uint64_t)i) % 3ull);
  double v = vel[i];
  if (r == 1) { v += 0.55; if (v > 2.77) v = 2.77; }
  else if (r == 2) { v -= 0.55; if (v < 1.1) v = 1.1; }
  vel[i] = v;
}

// subject-major packed table -> reference-shaped [env][viewer][subject] planes
__global__ void export_tables_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, const double* tx,
                                     const double* pos_y, int32_t* seq, int32_t* age, double* x,
                                     double* y) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * N;
  if (i >= total) return;
  const int k = (int)(i % N);
  const int u = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t src = ((size_t)b * NR + k) * NV + u;
  const uint32_t w = tkey[src];
  if (seq) seq[i] = (int32_t)(w >> 8);
  if (age) age[i] = (int32_t)(w & 255u);
  if (x) x[i] = tx[src];
  if (y) y[i] = (w >> 8) ? pos_y[(size_t)b * N + k] : 0.0;
}

// The xpos ring of the N <= 64 kernel (step_fast64.hpp): ring[env][subject][seq & 7]
// is the subject's stamp at sequence number `seq`, for its 8 most recent numbers.  An entry that
// lags its subject by at most 7 finds its xpos there; only older entries need the per-entry plane.
// rebuild: plane -> ring (after an import or a step of another kernel family); materialise: ring ->
// plane (before anything else reads the plane).  Entries of one subject with equal sequence numbers
// hold equal xpos in every reachable state, so concurrent writers of a slot agree.
__global__ void ring_rebuild_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, const double* tx, double* ring) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N * N) return;
  const int u = (int)(i % N);
  const int k = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t row = (size_t)b * NR + k;
  const uint32_t seq = tkey[row * NV + u] >> 8, tk = tkey[row * NV + k] >> 8;
  if (tk - seq <= 7u) ring[row * 8 + (seq & 7u)] = tx[row * NV + u];
}
// After an import: every entry the ring answers for must find ITS xpos there.  Tables no run of the reference
// produces (two entries about one subject with the same sequence number and different xpos) make the writers of
// ring_rebuild_kernel race; the loser is found here and reported through the sticky error word
// (diral_env_check: DIRAL_ERR_TABLE_CONFLICT).  Bitwise comparison: -0.0 / NaN payloads count as written.
__global__ void ring_verify_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, const double* tx, const double* ring,
                                   uint32_t* err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N * N) return;
  const int u = (int)(i % N);
  const int k = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t row = (size_t)b * NR + k;
  const uint32_t seq = tkey[row * NV + u] >> 8, tk = tkey[row * NV + k] >> 8;
  if (tk - seq <= 7u &&
      __double_as_longlong(ring[row * 8 + (seq & 7u)]) != __double_as_longlong(tx[row * NV + u]))
    atomicOr(err, kErrTable);
}
__global__ void ring_materialize_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, const double* ring, double* tx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N * N) return;
  const int u = (int)(i % N);
  const int k = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t row = (size_t)b * NR + k;
  const uint32_t seq = tkey[row * NV + u] >> 8, tk = tkey[row * NV + k] >> 8;
  if (tk - seq <= 7u) tx[row * NV + u] = ring[row * 8 + (seq & 7u)];
}

// The PACKED table of step_fast64 (step_fast64.hpp: codes, ages, own sequence numbers, old-quad flags) from the
// planes and back.  One thread per (env, row-quad, viewer): four table entries.
//   pack: after an import or a step of another kernel family (`told` zeroed by the caller beforehand).  An entry
//   is coded if it was heard (seq != 0) and lags its subject by at most 7; one that lags 7 or more flags its
//   quad - from the next slot on it is beyond the codes, and the planes hold it (they are complete right now).
//   unpack: before anything reads the planes.  A coded entry's sequence number is the subject's own minus its lag,
//   its xpos the subject's stamp of that number in the ring; a code-0 entry keeps the plane's sequence number
//   (0: never heard) and xpos; every entry's age comes from the age words.
__global__ void pack_codes_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, uint32_t* tcode, uint32_t* tage,
                                  uint32_t* tseq, uint32_t* told) {
  const int NQ = NR >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * NQ * NV) return;
  const int u = (int)(i % NV);
  const int q = (int)((i / NV) % NQ);
  const int b = (int)(i / ((size_t)NV * NQ));
  uint32_t code = 0u, age = 0u;
  bool flag = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * q + c;
    const size_t row = (size_t)b * NR + k;
    const bool ex = k < N && u < N;
    const uint32_t w = ex ? tkey[row * NV + u] : 0u;
    const uint32_t tk = k < N ? tkey[row * NV + k] >> 8 : 0u;
    const uint32_t seq = w >> 8, lag = tk - seq;
    const bool coded = seq != 0u && lag <= 7u;
    code |= (coded ? ((0xffu << lag) & 0xffu) : 0u) << (8 * c);
    age |= (w & 255u) << (8 * c);
    flag = flag || (seq != 0u && lag >= 7u);
    if (u == k || (k >= N && u == 0)) tseq[row] = tk;
  }
  tcode[i] = code;
  tage[i] = age;
  if (flag) atomicOr(&told[(size_t)b * NQ + q], 1u);
}
__global__ void unpack_codes_kernel(int B, int N, int NV, int NR, const uint32_t* tcode, const uint32_t* tage,
                                    const uint32_t* tseq, const double* ring, uint32_t* tkey, double* tx) {
  const int NQ = NR >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * NQ * NV) return;
  const int u = (int)(i % NV);
  const int q = (int)((i / NV) % NQ);
  const int b = (int)(i / ((size_t)NV * NQ));
  if (u >= N) return;
  const uint32_t code = tcode[i], age = tage[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * q + c;
    if (k >= N) break;
    const size_t row = (size_t)b * NR + k;
    const uint32_t r = (code >> (8 * c)) & 255u, a = (age >> (8 * c)) & 255u;
    if (r) {
      const uint32_t seq = tseq[row] - 8u + (uint32_t)__popc(r);
      tkey[row * NV + u] = (seq << 8) | a;
      tx[row * NV + u] = ring[row * 8 + (seq & 7u)];
    } else {
      tkey[row * NV + u] = (tkey[row * NV + u] & ~255u) | a;
    }
  }
}

__global__ void import_tables_kernel(int B, int N, int NV, int NR, const int32_t* seq, const int32_t* age,
                                     const double* x, uint32_t* tkey, double* tx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * N;
  if (i >= total) return;
  const int k = (int)(i % N);
  const int u = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t dst = ((size_t)b * NR + k) * NV + u;
  if (seq || age) {
    uint32_t w = tkey[dst];
    uint32_t s = seq ? (uint32_t)seq[i] : (w >> 8);
    int a = age ? age[i] : (int)(w & 255u);
    if (a > 255) a = 255;
    if (a < 0) a = 0;
    tkey[dst] = (s << 8) | (uint32_t)a;
  }
  if (x) tx[dst] = x[i];
}

// The tables as RealNeS MA_NeighborTableEntry records (diral_env.h: DiralNeighborEntry), one 16-byte
// record per thread: a coalesced dwordx4 on the record side, a stride-NV gather on the plane side.
__global__ void export_entries_kernel(int B, int N, int NV, int NR, const uint32_t* tkey, const double* tx,
                                      const double* pos_y, DiralNeighborEntry* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N * N) return;
  const int k = (int)(i % N);
  const int u = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t src = ((size_t)b * NR + k) * NV + u;
  const uint32_t w = tkey[src];
  DiralNeighborEntry r;
  r.pos_x = (float)tx[src];
  r.pos_y = (w >> 8) ? (float)pos_y[(size_t)b * N + k] : 0.0f;
  r.seq_num = (int32_t)(w >> 8);
  r.last_update = (int32_t)(w & 255u);
  out[i] = r;
}

__global__ void import_entries_kernel(int B, int N, int NV, int NR, const DiralNeighborEntry* in, uint32_t* tkey,
                                      double* tx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N * N) return;
  const int k = (int)(i % N);
  const int u = (int)((i / N) % N);
  const int b = (int)(i / ((size_t)N * N));
  const size_t dst = ((size_t)b * NR + k) * NV + u;
  const DiralNeighborEntry r = in[i];
  const int a = r.last_update > 255 ? 255 : (r.last_update < 0 ? 0 : r.last_update);
  tkey[dst] = ((uint32_t)r.seq_num << 8) | (uint32_t)a;
  tx[dst] = (double)r.pos_x;
}

// Network.get_information_age (network.py:560-574); one workgroup per env.
// Python's negative list indexing (ia in [-100,-1]) is reproduced.
__global__ void info_age_kernel(int N, long long t, const int32_t* la, int32_t* out) {
  __shared__ int bins[100];
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < 100; j += blockDim.x) bins[j] = 0;
  __syncthreads();
  const int32_t* l = la + (size_t)b * N * N;
  for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
    const int tx = i / N, rx = i - tx * N;
    if (tx == rx) continue;
    const int32_t v = l[i];
    if (v != -1) {
      long long ia = t - (long long)v;
      if (ia < 100) { if (ia < 0) ia += 100; if (ia >= 0) atomicAdd(&bins[(int)ia], 1); }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 100; j += blockDim.x) out[(size_t)b * 100 + j] = bins[j];
}

// the slot clock of a captured rollout (diral_env_set_clock): one thread
__global__ void clock_add_kernel(long long* clock, long long inc) { *clock += inc; }

// SemiPersistentScheduling.__init__ (algorithms/v2x_sps.py:8-22)
__global__ void sps_init_kernel(int agents, int window, uint64_t seed, int32_t* prev_action, int32_t* counter) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= agents) return;
  prev_action[i] = (int32_t)(rng_u64(seed, 5, (uint64_t)i) % (uint64_t)(window + 1));   // randint(0, window)
  counter[i] = 5 + (int32_t)(rng_u64(seed, 6, (uint64_t)i) % 11ull);                     // randint(5, 15)
}

// SemiPersistentScheduling.choose_new_resource (algorithms/v2x_sps.py:24-74) for one agent;
// `w(s)` is its selection window.  O(A^2) stable rank selection: re-selection is rare (counter
// expiry x 20 %), so only a few lanes run it.
template <typename W>
__device__ inline int sps_choose(W w, int A, int prev, double threshold, double inc_db, unsigned int r) {
  const double min_sA = (double)A / 5.0;                           // len(selection_window)/5
  double thr_next = threshold, thr = threshold;
  int n_sa = 0;
  for (int it = 0; it < 100000; ++it) {                            // while len(sA) < min_sA
    thr = thr_next;                                                // the threshold THIS sA is built with
    n_sa = 0;
    for (int s = 0; s < A; ++s) n_sa += (s != prev && w(s) < thr) ? 1 : 0;
    thr_next = thr + inc_db;                                       // tmp_threshold += self.inc_dB
    if (!((double)n_sa < min_sA)) break;
  }
  const double min_len = min_sA < (double)n_sa ? min_sA : (double)n_sa;
  int need = (int)min_len;
  if ((double)need < min_len) need += 1;                           // sB grows until len(sB) >= min_len
  if (need < 1) need = 1;
  const int pick = (int)(r % (unsigned int)need);                  // random.choice(sB)
  // sB[pick]: the (pick + 1)-th entry of sorted(sA.items(), key=value) - a stable sort, i.e. ordered by
  // (value, subframe).  pick < need <= A / 5 + 1, so pick + 1 minimum scans beat ranking every subframe.
  int chosen = prev, last_s = -1;
  double last_w = 0.0;
  for (int k = 0; k <= pick; ++k) {
    int best_s = -1;
    double best_w = 0.0;
    for (int s = 0; s < A; ++s) {
      const double ws = w(s);
      if (s == prev || !(ws < thr)) continue;
      if (k > 0 && !(ws > last_w || (ws == last_w && s > last_s))) continue;      // at or before the previous pick
      if (best_s < 0 || ws < best_w) { best_s = s; best_w = ws; }               // first minimum: lowest subframe wins ties
    }
    if (best_s < 0) break;                                       // cannot happen: need <= len(sA)
    last_s = best_s; last_w = best_w; chosen = best_s;
  }
  return chosen;
}

__global__ void sps_step_kernel(int agents, int A, const double* win, int32_t* prev_action, int32_t* counter,
                                double threshold, double inc_db, double keep_prob, const int32_t* draw_counter,
                                const double* draw_keep, const int32_t* draw_choice, uint64_t seed,
                                int32_t* actions_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= agents) return;
  int action = prev_action[i];
  int cnt = counter[i];
  if (sps_advance(i, cnt, keep_prob, draw_counter, draw_keep, seed)) {
    const double* w = win + (size_t)i * A;
    const unsigned int r = draw_choice ? (unsigned int)draw_choice[i]
                                       : (unsigned int)(rng_u64(seed, 9, (uint64_t)i) >> 33);
    action = sps_choose([&](int s) { return w[s]; }, A, action, threshold, inc_db, r);
    prev_action[i] = action;                                           // v2x_sps.py:98
  }
  counter[i] = cnt;
  actions_out[i] = action;
}

template <typename T>
__global__ void sps_window_kernel(size_t total, int A, const T* chobs, const int32_t* actions, double* win) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t i = e / A;
  win[e] = sps_rssi_from_chobs((double)chobs[e], (int)(e - i * A) == actions[i]);
}

// SemiPersistentScheduling.step for 64 agents per wave.  CHOBS: `src` is the env's channel observation
// [agents][A] (T) and the window is built on the fly; otherwise `src` is the window itself (double).
template <int NC, typename T, bool CHOBS>
__global__ void sps_step_wave_kernel(int agents, int A, const T* src, const int32_t* actions_in, int32_t* prev_action,
                                     int32_t* counter, double threshold, double inc_db, double keep_prob,
                                     const int32_t* draw_counter, const double* draw_keep, const int32_t* draw_choice,
                                     uint64_t seed0, const long long* clock, int32_t* actions_out) {
  // (`clock`: a device counter added to the seed - the draws of a captured graph move on with its replays)
  const uint64_t seed = seed0 + (clock ? (uint64_t)*clock : 0ull);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i < agents;
  int action = live ? prev_action[i] : 0;
  int cnt = live ? counter[i] : 1;
  const bool resel = live && sps_advance(i, cnt, keep_prob, draw_counter, draw_keep, seed);
  unsigned int r = 0;
  int own = -1;
  if (resel) {
    r = draw_choice ? (unsigned int)draw_choice[i] : (unsigned int)(rng_u64(seed, 9, (uint64_t)i) >> 33);
    if constexpr (CHOBS) own = actions_in[i];
  }
  unsigned long long todo = __ballot(resel);
  const int i0 = i - lane;
  // the re-selecting agents of the wave, four at a time: their rows are loaded together (one HBM round trip per batch
  // instead of one per agent - the kernel is a chain of such round trips), then decided one by one
  constexpr int NB = 4;
  while (todo) {
    int js[NB];
    T raw[NB][NC];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      js[q] = todo ? __builtin_ctzll(todo) : -1;               // (wave-uniform)
      todo &= todo - 1;                                        // (0 stays 0)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int s = lane + 64 * c;
        raw[q][c] = (T)0;
        if (js[q] >= 0 && s < A) raw[q][c] = src[(size_t)(i0 + js[q]) * A + s];
      }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int j = js[q];
      if (j < 0) break;
      const int prev_j = __builtin_amdgcn_readlane(action, j);
      const int own_j = __builtin_amdgcn_readlane(own, j);
      const unsigned int r_j = (unsigned int)__builtin_amdgcn_readlane((int)r, j);
      double w[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) w[c] = (lane + 64 * c < A) ? (double)raw[q][c] : 0.0;
      int ch;
      if constexpr (CHOBS) ch = sps_choose_chobs_wave<NC>(w, lane, A, prev_j, own_j, threshold, inc_db, r_j);
      else ch = sps_choose_wave<NC>(w, lane, A, prev_j, threshold, inc_db, r_j);
      if (lane == j) action = ch;
    }
  }
  if (resel) prev_action[i] = action;                                 // v2x_sps.py:98
  if (live) {
    counter[i] = cnt;
    actions_out[i] = action;
  }
}

// np.sum over one row in NumPy's order (numpy/_core/src/umath/loops_utils.h pairwise_sum): fewer
// than 8 elements sequentially; up to 128: eight running accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; above: split at n/2 rounded down to a
// multiple of 8, recursively.  Float addition is not associative: the driver's shaped rewards
// (main_test.py:171, 205-206) are bit-identical only in this order.
template <typename T>
__device__ T np_pairwise_sum(const T* a, int n) {
  if (n < 8) {
    T res = (T)0;
    for (int i = 0; i < n; ++i) res = res + a[i];
    return res;
  }
  if (n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// The driver's per-slot reward post-processing (main_test.py:150-206), for `envs` envs in one
// launch; 64 envs per 256-thread block: one thread per env sums and decides, then all threads
// rewrite the block's rewards coalesced.
//   ia_sum      = sum_i (i + 1) * ia[i] over bins with ia[i] > 0          (utils/misc.py:1-12)
//   ia_penalty  = -1 / +1 / 0 as ia_sum rose / fell / stayed (ia_averaging, :153-160)
//   sum_r       = np.sum(reward) (:171), collision = A - sum_r (:178)
//   reward'     = reward + ia_penalty (:190-192); counter / threshold penalty (:194-203);
//                 + sum_r / N (global_reward_avg, :205-206)
constexpr int kShapeEnvsPerBlock = 64;

// The common case - no information-age terms, 8 <= N <= 64 - one WAVE per env: lane = vehicle, the rewards arrive with
// one coalesced load, np.sum's order (eight running accumulators over blocks of eight, a pairwise tree, a sequential
// tail: numpy/_core/src/umath/loops_utils.h pairwise_sum) is walked with lane shuffles, and every lane rewrites its own
// reward.  (The thread-per-env kernel below reads 64 strided rows per wave and sums them one element at a time:
// 12 us at 4096 envs against 3 us here.)
constexpr int kShapeWaveBlock = 256;
template <typename T>
__global__ __launch_bounds__(kShapeWaveBlock) void driver_shape_wave_kernel(int envs, int N, int A, const T* reward_in,
                                                                           const int32_t* actions, int32_t* pen_counter,
                                                                           int32_t* prev_actions, int flags, int pen_threshold,
                                                                           double pen_value, T* reward_out, T* sum_r_out,
                                                                           T* collision_out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (kShapeWaveBlock / 64) + (threadIdx.x >> 6);
  if (b >= envs) return;                                            // (wave-uniform)
  const bool global_avg = flags & 1, pen_enable = flags & 4;
  const size_t g = (size_t)b * N + lane;
  const bool live = lane < N;
  const T a = live ? reward_in[g] : (T)0;
  const T sr = np_row_sum_wave(a, N, lane);
  if (lane == 0) {
    if (sum_r_out) sum_r_out[b] = sr;
    if (collision_out) collision_out[b] = (T)A - sr;
  }
  if (live) {
    T rr = a;
    if (pen_enable) {
      const int ac = actions[g];
      const bool stuck = (rr < (T)1) && (ac == prev_actions[g]);
      const int c = stuck ? pen_counter[g] + 1 : 0;
      pen_counter[g] = c;
      if (c > pen_threshold) rr = (T)pen_value;
      prev_actions[g] = ac;
    }
    if (global_avg) rr = rr + sr / (T)N;
    reward_out[g] = rr;
  }
}

template <typename T>
__global__ void driver_shape_kernel(int envs, int N, int A, const T* reward_in, const int32_t* actions,
                                    const int32_t* ia, long long* sum_ia_prev, int32_t* pen_counter,
                                    int32_t* prev_actions, int flags, int pen_threshold, double pen_value,
                                    T* reward_out, T* sum_r_out, T* collision_out, long long* ia_sum_out,
                                    int32_t* ia_pen_out) {
  __shared__ T s_sum[kShapeEnvsPerBlock];
  __shared__ T s_pen[kShapeEnvsPerBlock];
  const int e0 = blockIdx.x * kShapeEnvsPerBlock;
  const int ne = min(kShapeEnvsPerBlock, envs - e0);
  const bool global_avg = flags & 1, ia_avg = flags & 2, pen_enable = flags & 4;
  if ((int)threadIdx.x < ne) {
    const int b = e0 + threadIdx.x;
    const T sr = np_pairwise_sum(reward_in + (size_t)b * N, N);
    s_sum[threadIdx.x] = sr;
    if (sum_r_out) sum_r_out[b] = sr;
    if (collision_out) collision_out[b] = (T)A - sr;
    T pen = (T)0;
    if (ia) {
      long long acc = 0;
      for (int i = 0; i < 100; ++i) { const int v = ia[(size_t)b * 100 + i]; acc += v > 0 ? (long long)(i + 1) * v : 0; }
      if (ia_sum_out) ia_sum_out[b] = acc;
      if (ia_avg && sum_ia_prev) {
        const long long prev = sum_ia_prev[b];
        const int p = acc > prev ? -1 : (acc < prev ? 1 : 0);
        sum_ia_prev[b] = acc;
        if (ia_pen_out) ia_pen_out[b] = p;
        pen = (T)p;
      }
    }
    s_pen[threadIdx.x] = pen;
  }
  __syncthreads();
  const int total = ne * N;
  for (int j = threadIdx.x; j < total; j += blockDim.x) {
    const int le = j / N;
    const size_t g = (size_t)e0 * N + j;
    T r = reward_in[g];
    if (ia_avg) r = r + s_pen[le];
    if (pen_enable) {
      const int a = actions[g];
      const bool stuck = (r < (T)1) && (a == prev_actions[g]);
      const int c = stuck ? pen_counter[g] + 1 : 0;
      pen_counter[g] = c;
      if (c > pen_threshold) r = (T)pen_value;
      prev_actions[g] = a;
    }
    if (global_avg) r = r + s_sum[le] / (T)N;
    reward_out[g] = r;
  }
}

__global__ void any_nonzero_kernel(int total, const double* v, uint32_t* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total && v[i] != 0.0) atomicOr(flag, 1u);
}

__global__ void metrics_kernel(int total, double* metrics, double* out, int clear) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (out) out[i] = metrics[i];
  if (clear) metrics[i] = 0.0;
}

}  // namespace diral
