// piggyback_kernel.hpp - State.piggybacking (test_env.py:33, 71-79, 241-254, 260-264).
//
// With the flag, my_step returns per agent u not its channel observation obs[u] (A values) but piggy_obs[u]:
// starting from A zeros, for every resource i in ascending order
//   u transmits on i:             piggy_obs[u][i] = 0                                        (test_env.py:209)
//   somebody else does:           piggy_obs[u][i] = tx_dist; np.insert(piggy_obs[u], i, prev_obs[tx_id])   (:241-247)
//   nobody does:                  np.insert(piggy_obs[u], i, zeros(A))                       (:250-254)
// with (tx_dist, tx_id) = find_closest_tx (network.py:378-398) and prev_obs = the plain `obs` dict of the previous
// my_step (:260-261).  Every agent sees A - 1 inserts of A values: A * A values, the channel-observation section of
// the state vector (test_env.py:71-72, 539-541).  tx_id None (a receiver with no transmitter in range on a used
// resource) is `self.prev_obs[None]`, a KeyError: here the sticky kErrPiggy flag (DIRAL_ERR_PIGGY_NO_TX).
//
// The indices i of the assignments and inserts address the array AS IT HAS GROWN, so earlier inserts are split and
// partly overwritten by later ones.  The kernels do not replay the inserts: for an output position p they walk the
// resources BACKWARDS, undoing one operation at a time, until they meet the operation that produced the value:
//   for i = A - 1 ... 0:
//     i == own action:  p == i -> 0; (no insert)
//     else:             p in [i, i + A) -> the inserted value (prev_obs[tx_i][p - i], or 0 on an idle resource);
//                       p >= i + A -> p -= A;  then, on a used resource, p == i -> tx_dist_i
//   nothing met -> 0 (the initial zeros).
// (oracle/diral_oracle.c replays the inserts on real arrays, like the reference: two independent statements.)
//
// Two launches around the step, because find_closest_tx needs the positions BEFORE the slot's move and the output
// belongs behind the step kernel's own state writer:
//   piggy_search_kernel  (before the step)  obs_new[b][u][i], txid[b][u][i] from actions + positions
//   piggy_emit_kernel    (after the step)   piggy_obs -> chobs_out / the state section; prev_obs = obs_new
// A secondary mode (no reference YAML sets it; defined only while every receiver hears a transmitter): simple
// one-workgroup-per-env kernels, not tuned.
#pragma once
#include "common.hpp"

namespace diral {

constexpr uint32_t kErrPiggy = 8u;

struct PiggyParams {
  int N, A, S;
  int off_chobs;               // column of the A * A section in a state row, or -1
  int out_f64;
  double Rc;
  const int32_t* actions;      // [B][N]
  const double* pos_x;         // [B][N] (search: the positions the slot starts with)
  const double* pos_y;
  double* obs_new;             // [B][N][A] this slot's plain observation
  int32_t* txid;               // [B][N][A] closest in-range transmitter, -1 none in range, -2 not a receiver here
  double* prev_obs;            // [B][N][A]
  void* chobs_out;             // [B][N][A * A] out dtype, or null
  void* state_out;             // [B][N][S] out dtype, or null
  const double* chobs_in;      // observe: [B][N][A * A] f64 copied into the state section (emit not used)
  uint32_t* err;
};

__device__ inline void piggy_store(void* base, size_t idx, double v, int f64) {
  if (f64) static_cast<double*>(base)[idx] = v;
  else static_cast<float*>(base)[idx] = (float)v;
}

__host__ __device__ inline uint32_t piggy_search_lds_bytes(int N) { return 20u * (uint32_t)N + 16u; }
// one workgroup per env, one thread per (receiver, resource) pair
__global__ void piggy_search_kernel(PiggyParams p) {
  const int b = blockIdx.x, N = p.N, A = p.A;
  const size_t bN = (size_t)b * N;
  // the env's actions and positions once into LDS (piggy_search_lds_bytes(N): 20 bytes per vehicle): every (receiver,
  // resource) pair walks all of them
  extern __shared__ __align__(16) unsigned char pg_smem[];
  double* const px = reinterpret_cast<double*>(pg_smem);
  double* const py = px + N;
  int32_t* const act = reinterpret_cast<int32_t*>(py + N);
  for (int u = threadIdx.x; u < N; u += blockDim.x) {
    act[u] = p.actions[bN + u];
    px[u] = p.pos_x[bN + u];
    py[u] = p.pos_y[bN + u];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * A; e += blockDim.x) {
    const int u = e / A, i = e - u * A;
    double best = 100000.0;                                      // network.py:385
    int bid = -1;
    int ntx = 0;
    const bool own = act[u] == i;
    for (int tx = 0; tx < N; ++tx) {                             // ascending id, strict '<' twice: network.py:387-392
      if (act[tx] != i) continue;
      ++ntx;
      if (own) continue;
      const double dx = px[u] - px[tx], dy = py[u] - py[tx];
      const double d = __builtin_sqrt(dx * dx + dy * dy);        // network.py:332
      if (d < p.Rc && d < best) { best = d; bid = tx; }
    }
    const bool rx = !own && ntx > 0;
    p.obs_new[bN * A + e] = rx ? best : 0.0;                     // test_env.py:206, 240
    p.txid[bN * A + e] = rx ? bid : -2;
    if (rx && bid < 0) atomicOr(p.err, kErrPiggy);               // test_env.py:243: prev_obs[None]
  }
}

__global__ void piggy_emit_kernel(PiggyParams p) {
  const int b = blockIdx.x, N = p.N, A = p.A, W = p.A * p.A;
  const size_t bN = (size_t)b * N;
  const int32_t* const act = p.actions + bN;
  const double* const obs = p.obs_new + bN * A;
  const int32_t* const txid = p.txid + bN * A;
  const double* const prev = p.prev_obs + bN * A;
  for (int e = threadIdx.x; e < N * W; e += blockDim.x) {
    const int u = e / W;
    int pos = e - u * W;
    const int a = act[u];
    double val = 0.0;
    for (int i = A - 1; i >= 0; --i) {
      if (i == a) {
        if (pos == i) break;                                     // piggy_obs[user][i] = 0
        continue;
      }
      const int t = txid[u * A + i];                             // >= 0 used, -1 used but nobody in range, -2 idle
      if (pos >= i && pos < i + A) {
        if (t >= 0) val = prev[t * A + (pos - i)];               // the inserted prev_obs[tx_id]
        break;                                                   // (idle: zeros; nobody in range: the error slot)
      }
      if (pos >= i + A) pos -= A;
      if (t != -2 && pos == i) { val = obs[u * A + i]; break; }  // piggy_obs[user][i] = tx_dist
    }
    if (p.chobs_out) piggy_store(p.chobs_out, bN * W + e, val, p.out_f64);
    if (p.state_out && p.off_chobs >= 0)
      piggy_store(p.state_out, (bN + u) * (size_t)p.S + p.off_chobs + (e - u * W), val, p.out_f64);
  }
  __syncthreads();                                               // every read of prev_obs of this env is done
  for (int e = threadIdx.x; e < N * A; e += blockDim.x) p.prev_obs[bN * A + e] = obs[e];   // test_env.py:260-261
}

// diral_env_observe: `obs` is an argument of obtain_state (test_env.py:539-541) - copy it into the section
__global__ void piggy_fill_kernel(PiggyParams p, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int W = p.A * p.A;
  const size_t row = e / W;
  const int c = (int)(e - row * W);
  piggy_store(p.state_out, row * (size_t)p.S + p.off_chobs + c, p.chobs_in ? p.chobs_in[e] : 0.0, p.out_f64);
}

}  // namespace diral
