// aux_kernels.hpp - the small kernels around the step: topology reset, action
// sampling, velocity update, state export/import, information-age histogram,
// metric read-out.  All are elementwise / tiny; none is on the hot path.
#pragma once
#include "common.hpp"

namespace diral {

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
  int chosen = prev;
  for (int s = 0; s < A; ++s) {                                    // sorted(sA.items(), key=value): stable
    const double ws = w(s);
    if (s == prev || !(ws < thr)) continue;
    int rank = 0;
    for (int q = 0; q < A; ++q) {
      const double wq = w(q);
      if (q == prev || !(wq < thr)) continue;
      rank += (wq < ws || (wq == ws && q < s)) ? 1 : 0;
    }
    if (rank == pick) chosen = s;
  }
  return chosen;
}

// SemiPersistentScheduling.step (algorithms/v2x_sps.py:76-104); one thread per agent.
// Returns true when the agent has to choose a new resource (then `cnt` is already redrawn).
__device__ inline bool sps_advance(int i, int& cnt, double keep_prob, const int32_t* draw_counter,
                                   const double* draw_keep, uint64_t seed) {
  if (cnt != 0) { cnt -= 1; return false; }                        // v2x_sps.py:85-89
  cnt = draw_counter ? draw_counter[i] : 5 + (int)(rng_u64(seed, 7, (uint64_t)i) % 12ull);   // randint(5, 16)
  const double u = draw_keep ? draw_keep[i] : rng_unit(rng_u64(seed, 8, (uint64_t)i));
  return !(u < keep_prob);                                         // v2x_sps.py:93-98
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

template <typename T>
__global__ void sps_window_kernel(size_t total, int A, const T* chobs, const int32_t* actions, double* win) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t i = e / A;
  win[e] = sps_rssi_from_chobs((double)chobs[e], (int)(e - i * A) == actions[i]);
}

constexpr int kSpsFusedMaxA = 64;

// The two steps above in one launch: the window is only built (in private memory) by the few agents
// that re-select this slot - no [agents][A] float64 array, no log10 for everybody.
template <typename T>
__global__ void sps_step_chobs_kernel(int agents, int A, const T* chobs, const int32_t* actions_in,
                                      int32_t* prev_action, int32_t* counter, double threshold, double inc_db,
                                      double keep_prob, const int32_t* draw_counter, const double* draw_keep,
                                      const int32_t* draw_choice, uint64_t seed, int32_t* actions_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= agents) return;
  int action = prev_action[i];
  int cnt = counter[i];
  if (sps_advance(i, cnt, keep_prob, draw_counter, draw_keep, seed)) {
    double wl[kSpsFusedMaxA];
    const T* row = chobs + (size_t)i * A;
    const int own = actions_in[i];
    for (int s = 0; s < A; ++s) wl[s] = sps_rssi_from_chobs((double)row[s], s == own);
    const unsigned int r = draw_choice ? (unsigned int)draw_choice[i]
                                       : (unsigned int)(rng_u64(seed, 9, (uint64_t)i) >> 33);
    action = sps_choose([&](int s) { return wl[s]; }, A, action, threshold, inc_db, r);
    prev_action[i] = action;
  }
  counter[i] = cnt;
  actions_out[i] = action;
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
