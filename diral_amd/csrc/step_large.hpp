// step_large.hpp - the env step for sizes the one-workgroup kernels do not hold: num_users > 256,
// num_channels > 256 or num_bins > 64 (TestEnv takes any N, A - test_env.py:12-13 - and any State.num_bins, :40).
//
// Same statement of the reference as step_kernel.hpp (TestEnv.my_step / my_step_ch / my_step_design +
// obtain_state, test_env.py:124-266 / 351-443 / 269-349 / 527-583), split where the data stops fitting a CU:
//   large_search_kernel   one workgroup per env: tx lists per resource (a stable counting sort of the
//                         actions), the closest in-range transmitter of every (viewer, resource) pair
//                         (Network.find_closest_tx, network.py:378-398), rewards, PRR, arrival stamps, the
//                         move (network.py:189-206), and every column of the state vector that needs no
//                         table; leaves the gather sources of the gossip merge in HBM (`src`, u16 [A][N])
//   large_mergen_kernel / large_merge_kernel
//                         one WAVE per table column (subject k; four columns on thermometer codes at N <= 512 where the table is fresh, two on rank keys up to 1024): Vehicle.periodic_update + every
//                         Vehicle.received_update of the slot (vehicle.py:35-70) as key[u] = max(key[u],
//                         key[src_i(u)]) for the resources in ascending order, key = (sequence number, source
//                         viewer), the column in the wave's LDS (the wave's own LDS queue is in order: no
//                         barrier); xpos follows the winning number from the source viewer's entry
//   large_hist_kernel     one workgroup per 64 viewers (fewer for many bins): Network.dist_piggy +
//                         get_positional_dist_2_piggy (network.py:538-558, 473-513), lane = viewer, four waves
//                         sharing the subjects, each lane's histogram row private in LDS
// The tables are the (seq << 8 | age) and xpos planes of common.hpp.  Columns are independent (tx rows
// are read-only within a resource, a receiver writes its own row), so the merge spreads over B x N waves where the
// one-workgroup kernels have B workgroups: at N = 1024 that is what fills the chip.
// Not tuned like step_fast64 / step_wide: no BASELINE.json configuration runs here.
#pragma once
#include "common.hpp"
#include "step_kernel.hpp"

namespace diral {

constexpr int kLargeMaxThreads = 1024;       // large_search_kernel: 4 ... 16 waves by the env's size
__host__ __device__ inline int large_search_threads(int N) { return N <= 256 ? 256 : (N <= 512 ? 512 : 1024); }
constexpr uint32_t kLargeMergeLdsBudget = 64u * 1024u;

struct LargeLds {
  uint32_t px, py, inr, rec, act, list, cnt, off, red, total;
};
__host__ __device__ inline LargeLds large_lds_layout(int N, int A) {
  const uint32_t np = (uint32_t)((N + 63) & ~63);
  LargeLds l;
  uint32_t o = 0;
  l.px = o;   o += 8u * np;
  l.py = o;   o += 8u * np;
  l.red = o;  o += 8u * 4u * (np / 64u);
  l.inr = o;  o += 4u * np;
  l.rec = o;  o += 4u * np;
  l.cnt = o;  o += 4u * (uint32_t)A;
  l.off = o;  o += 4u * ((uint32_t)A + 1u);
  l.act = o;  o += 2u * np;
  l.list = o; o += 2u * np;
  l.total = align_up(o, 16);
  return l;
}
// waves (= columns) per workgroup of the merge, and its LDS: 8 bytes per viewer and column + one flag word per 64 viewers
__host__ __device__ inline int large_merge_waves(int N) {
  const uint32_t np = (uint32_t)((N + 63) & ~63);
  const uint32_t per = 8u * np + 8u * (np / 64u);
  const uint32_t w = kLargeMergeLdsBudget / per;
  return w >= 4u ? 4 : (w >= 2u ? 2 : 1);
}
__host__ __device__ inline uint32_t large_merge_lds(int N) {
  const uint32_t np = (uint32_t)((N + 63) & ~63);
  return (uint32_t)large_merge_waves(N) * (8u * np + 8u * (np / 64u));
}
// large_mergen_kernel<CH, NC>: four waves, each 64 CH key vectors of NC words - and never less than the 64 CH + CH 64-bit
// words large_merge_column needs when it takes the slice over
__host__ __device__ inline uint32_t large_mergen_slice(int ch, int nc) {
  const uint32_t a = 4u * (uint32_t)nc * 64u * (uint32_t)ch, b = 8u * (uint32_t)(64 * ch + ch);
  return align_up(a > b ? a : b, 16);
}
__host__ __device__ inline uint32_t large_mergen_lds(int ch, int nc) { return 4u * large_mergen_slice(ch, nc); }
// large_mergec_kernel<CH>: four waves, each 64 CH code words + the [4][8] xpos table
__host__ __device__ inline uint32_t large_mergec_lds(int ch) { return 4u * 4u * (uint32_t)(64 * ch + 64); }
// viewers per workgroup of the histogram kernel: 64, fewer when the rows of K counters would not fit
__host__ __device__ inline int large_hist_viewers(int K) {
  int vw = 64;
  while (vw > 1 && 4u * (uint32_t)vw * 4u * (uint32_t)(K | 1) > 96u * 1024u) vw >>= 1;   // four waves' rows
  return vw;
}
__host__ __device__ inline uint32_t large_hist_lds(int K) {
  return 8u * (uint32_t)(K + 2) + 4u * 4u * 64u + 4u * (uint32_t)large_hist_viewers(K) * 4u * (uint32_t)(K | 1);
}

// Network.calculate_reward_weights (network.py:273-300) over the ascending transmitter list of one resource;
// wave-uniform (every lane walks the same pairs in the same order: calculate_avg_distance sums serially)
__device__ inline int large_reward_weight(const StepParams& p, const unsigned short* lst, int c, const double* s_px,
                                          const double* s_py) {
  double s = 0.0;
  int cnt = 0;
  for (int qa = 0; qa < c; ++qa) {
    const int a = lst[qa];
    for (int qb = qa + 1; qb < c; ++qb) {
      const int b = lst[qb];
      s = s + dist2d(s_px[a], s_py[a], s_px[b], s_py[b]);
      ++cnt;
    }
  }
  const double m = s / (double)cnt;
  if (p.flags & DIRAL_F_TOY_WEIGHTS) {
    double x_min = p.L + 1, x_max = -p.L - 1;      // calculate_norm (network.py:225-246)
    int umin = 0, umax = 0;
    for (int u = 0; u < p.N; ++u) {
      const double x = s_px[u];
      if (x < x_min) { x_min = x; umin = u; }
      if (x > x_max) { x_max = x; umax = u; }
    }
    return m == dist2d(s_px[umin], s_py[umin], s_px[umax], s_py[umax]);
  }
  return m > p.Rc;
}

__global__ __launch_bounds__(kLargeMaxThreads) void large_search_kernel(const StepParams p, const LargeScratch g) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int kLargeThreads = (int)blockDim.x, kLargeWaves = kLargeThreads >> 6;
  const LargeLds lay = large_lds_layout(p.N, p.A);
  double* s_px = reinterpret_cast<double*>(smem + lay.px);
  double* s_py = reinterpret_cast<double*>(smem + lay.py);
  double* s_red = reinterpret_cast<double*>(smem + lay.red);
  int* s_inr = reinterpret_cast<int*>(smem + lay.inr);
  int* s_rec = reinterpret_cast<int*>(smem + lay.rec);
  int* s_cnt = reinterpret_cast<int*>(smem + lay.cnt);
  int* s_off = reinterpret_cast<int*>(smem + lay.off);
  short* s_act = reinterpret_cast<short*>(smem + lay.act);
  unsigned short* s_list = reinterpret_cast<unsigned short*>(smem + lay.list);

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, A = p.A, K = p.K;
  const int NP = (N + 63) & ~63;
  const int mode = p.mode;
  const bool do_step = mode != kModeObserve;
  const bool piggy = (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) != 0;
  const bool track_la = (p.flags & DIRAL_F_TRACK_ARRIVAL) && p.la != nullptr;
  const bool want_prr = do_step && (mode == DIRAL_STEP_MY_STEP_CH || (mode == DIRAL_STEP_MY_STEP && (p.flags & DIRAL_F_TRACK_PRR)));
  const bool mobile = (p.flags & DIRAL_F_MOBILITY) != 0;
  const bool use_pf = (p.flags & DIRAL_F_PROPORTIONAL_FAIR) != 0;
  const int out_f64 = p.out_f64;
  const size_t bN = (size_t)b * N, bA = (size_t)b * A;
  const long long t_now = p.t + (p.t_dev ? *p.t_dev : 0ll);

  // ---- stage the vehicles; the move (update_positions, network.py:189-206) -----------------------------
  for (int u = tid; u < NP; u += kLargeThreads) {
    int a = -1;
    double x = 0.0, y = 0.0;
    if (u < N) {
      a = p.actions[bN + u];
      if (a < 0 || a >= A) { atomicOr(p.err, kErrAction); a = -1; }
      x = p.pos_x[bN + u];
      y = p.pos_y[bN + u];
      if (do_step) {
        g.px0[bN + u] = x;
        g.rew[bN + u] = 0.0;
        if (mobile) {
          double nx;
          if (p.trace) {                                             // replay branch, network.py:194-199
            long long tt = t_now % p.trace_len;
            if (tt < 0) tt += p.trace_len;
            const size_t base = p.trace_per_env ? (size_t)b * p.trace_len : 0;
            nx = p.trace[(base + (size_t)tt) * N + u];
          } else {
            nx = py_mod_pos(x + p.vel[bN + u] + p.L, p.L);           // network.py:203
          }
          p.pos_x[bN + u] = nx;
        }
      }
    }
    s_act[u] = (short)a; s_px[u] = x; s_py[u] = y; s_inr[u] = 0; s_rec[u] = 0;
  }
  for (int i = tid; i < A; i += kLargeThreads) s_cnt[i] = 0;
  __syncthreads();

  if (do_step) {
    // ---- transmitter lists per resource, ascending id (test_env.py:141-157): a stable counting sort -------
    for (int u = tid; u < N; u += kLargeThreads) {
      const int a = s_act[u];
      if (a >= 0) atomicAdd(&s_cnt[a], 1);
    }
    __syncthreads();
    if (wave == 0) {
      int running = 0;
      for (int base = 0; base < A; base += 64) {
        const int i = base + lane;
        const int c = i < A ? s_cnt[i] : 0;
        int inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_up(inc, off);
          if (lane >= off) inc += o;
        }
        if (i < A) { s_off[i] = running + inc - c; s_cnt[i] = running + inc - c; }   // s_cnt: the fill cursor
        running += __shfl(inc, 63);
      }
      if (lane == 0) s_off[A] = running;
      wave_lds_order();
      for (int base = 0; base < NP; base += 64) {
        const int a = s_act[base + lane];
        unsigned long long todo = __ballot(a >= 0);
        while (todo) {
          const int l = __builtin_ctzll(todo);
          const int a0 = __builtin_amdgcn_readlane(a, l);
          const unsigned long long m = __ballot(a == a0);
          const int cur = s_cnt[a0];
          wave_lds_order();
          if (a == a0) s_list[cur + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(base + lane);
          if (lane == l) s_cnt[a0] = cur + __popcll(m);
          wave_lds_order();
          todo &= ~m;
        }
      }
    }
    __syncthreads();
    if (wave == 1) {                                                  // the used resources, ascending: what the merge walks
      int n = 0;
      for (int base = 0; base < A; base += 64) {
        const int i = base + lane;
        const bool used = i < A && s_off[i + 1] > s_off[i];
        const unsigned long long m = __ballot(used);
        if (used) g.alist[bA + n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
        n += __popcll(m);
      }
      if (lane == 0) g.nact[b] = (uint32_t)n;
    }

    // ---- per resource: closest in-range transmitter per viewer, channel observation, rewards ----------------
    const bool want_obs = p.chobs_out != nullptr || (p.state_out != nullptr && p.off_chobs >= 0);
    for (int i = wave; i < A; i += kLargeWaves) {
      const int o0 = s_off[i], c = s_off[i + 1] - o0;
      const unsigned short* lst = s_list + o0;
      if (lane == 0) g.cnt[bA + i] = (uint32_t)c;
      if (c == 0 && !want_obs) continue;
      for (int vb = 0; vb < NP; vb += 64) {
        const int u = vb + lane;
        const bool valid = u < N;
        const double mx = s_px[u], my = s_py[u];
        const int myact = s_act[u];
        const bool rx = valid && myact != i;
        double best = 100000.0;                                      // network.py:385-386
        int bid = -1;
        for (int q = 0; q < c; ++q) {                                // ascending tx id, strict '<' (network.py:387-392)
          const int w = lst[q];
          const double d = dist2d(s_px[w], s_py[w], mx, my);
          const bool inr = d < p.Rc;
          if (inr && d < best) { best = d; bid = w; }
          if (track_la && rx && !inr) p.la[(bN + w) * N + u] = -1;   // network.py:394
          if (want_prr && c > 1) {                                   // test_env.py:395-397
            const int n = __popcll(__ballot(rx && inr));
            if (lane == 0) s_inr[w] += n;
          }
        }
        const bool is_tx = myact == i;
        const bool got = !is_tx && bid >= 0 && valid && c > 0;
        if (valid && c > 0) g.src[(bA + i) * N + u] = (unsigned short)(got ? bid : u);
        if (valid) {
          double ob = 0.0;                                           // `obs[user][i]`
          if (!is_tx && c > 0) {
            if (mode == DIRAL_STEP_MY_STEP && p.state_type == 2) ob = best;   // test_env.py:240
            else ob = 1.0;                                           // :228, :306, :432
          }
          if (p.chobs_out) store_out(p.chobs_out, (bN + u) * A + i, ob, out_f64);
          if (p.state_out && p.off_chobs >= 0) store_out(p.state_out, (bN + u) * p.S + p.off_chobs + i, ob, out_f64);
          if (track_la && mode == DIRAL_STEP_MY_STEP_CH && got) p.la[(bN + bid) * N + u] = (int32_t)t_now;   // test_env.py:436
        }
        if (want_prr && c > 1 && rx && bid >= 0) atomicAdd(&s_rec[bid], 1);   // test_env.py:398-400
      }
      if (c == 0) continue;
      wave_lds_order();
      // one value per colliding resource (test_env.py:163-199)
      double rw = 0.0;
      if (mode == DIRAL_STEP_MY_STEP && c > 1) {
        const int rd = p.reward_design;
        if (rd == 1) {
          const double R = (double)large_reward_weight(p, lst, c, s_px, s_py) / (double)c;
          rw = -1.0 * (1.0 - R);
        } else if (rd == 2) {
          if (c == 2) rw = 2.0 * (double)large_reward_weight(p, lst, c, s_px, s_py) - (double)c;
          else rw = 0.0 - (double)c;
        } else if (rd == 3) {
          const double R = 1.0 / (double)c;
          rw = -1.0 * exp(1.0 - R);
        } else if (rd == 4) {
          rw = 1.0 / (double)c;
        } else {
          if (c == 2) rw = (large_reward_weight(p, lst, c, s_px, s_py) == 1) ? 0.0 : -1.0;
          else rw = -1.0;
        }
      }
      // the reward of each of its transmitters
      for (int q = lane; q < c; q += 64) {
        const int u = lst[q];
        double r = 0.0;
        if (mode == DIRAL_STEP_MY_STEP) {                              // test_env.py:211-222
          if (c > 1) {
            r = rw;
            if (use_pf) {
              const int pc = p.pf[bN + u];
              if (pc > p.pf_threshold) r = p.pf_penalty;
              p.pf[bN + u] = pc + 1;
            }
          } else {
            r = 1.0;
            if (use_pf) p.pf[bN + u] = 0;
          }
          if (want_prr && c > 1) {
            const int n_in = s_inr[u];
            g.rtx[bN + u] = n_in > 0 ? (double)s_rec[u] / (double)n_in : 1.0;     // test_env.py:402-405
          }
        } else if (mode == DIRAL_STEP_MY_STEP_CH) {                     // test_env.py:411-429
          const int rd = p.reward_design;
          if (c > 1) {
            const int n_in = s_inr[u];
            const double R = n_in > 0 ? (double)s_rec[u] / (double)n_in : 1.0;
            g.rtx[bN + u] = R;
            if (rd == 3) r = 1.0 - exp(1.0 - R);
            else if (rd == 4) r = -1.0 * exp(1.0 - R);
            else if (rd == 2) r = -1.0 * (1.0 - R);
          } else {
            if (rd == 3) r = 1.0;
            else if (rd == 4) r = exp(1.0);
            else if (rd == 2) r = 1.0;
          }
        } else {                                                       // test_env.py:297-301, 319-349
          if (c == 1) r = 1.0;
          else {
            int n = 1;
            double dlast = 0.0;
            for (int qo = 0; qo < c; ++qo) {
              const int o = lst[qo];
              if (o == u) continue;
              const double d = dist2d(s_px[u], s_py[u], s_px[o], s_py[o]);
              if (d < 2.0 * p.Rc) { if (n == 1) dlast = d; ++n; }      // network.py:122-133
            }
            if (n == 1) r = 1.0;
            else if (n == 2) r = ((dlast / 1.0) > p.Rc * 2.0) ? 0.0 : -2.0;     // network.py:135-157
            else r = -(double)n;
          }
        }
        g.rew[bN + u] = r;
        if (p.rew_out) store_out(p.rew_out, bN + u, r, out_f64);
      }
    }
    __syncthreads();

    // ---- metric accumulators: per 64 vehicles a shuffle tree, the partial sums added in order ---------------
    for (int u = tid; u < NP; u += kLargeThreads) {
      double vr = 0.0, vp = 0.0;
      int vs = 0, vc = 0;
      const int a = s_act[u];
      if (u < N) {
        if (a >= 0) {
          const int c = s_off[a + 1] - s_off[a];
          vr = g.rew[bN + u];
          vs = c == 1; vc = c > 1;
          if (want_prr) vp = c > 1 ? g.rtx[bN + u] : 1.0;
        } else if (p.rew_out) {
          store_out(p.rew_out, bN + u, 0.0, out_f64);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        vr += __shfl_down(vr, off);
        vp += __shfl_down(vp, off);
        vs += __shfl_down(vs, off);
        vc += __shfl_down(vc, off);
      }
      if (lane == 0) {
        const int slot = u >> 6;
        s_red[slot * 4 + 0] = vr; s_red[slot * 4 + 1] = vp; s_red[slot * 4 + 2] = (double)vs; s_red[slot * 4 + 3] = (double)vc;
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (p.done_out) p.done_out[b] = (uint8_t)((t_now % p.episode_interval) == p.episode_interval - 1);
      double sr = 0.0, sp = 0.0, ss = 0.0, sc = 0.0;
      for (int w = 0; w < NP / 64; ++w) { sr += s_red[w * 4 + 0]; sp += s_red[w * 4 + 1]; ss += s_red[w * 4 + 2]; sc += s_red[w * 4 + 3]; }
      double* mt = p.metrics + (size_t)b * DIRAL_M_COLUMNS;
      mt[DIRAL_M_SLOTS] += 1.0;
      mt[DIRAL_M_SUM_REWARD] += sr;
      mt[DIRAL_M_TX_SOLE] += ss;
      mt[DIRAL_M_TX_COLLIDED] += sc;
      if (want_prr) { mt[DIRAL_M_PRR_SUM] += sp; mt[DIRAL_M_PRR_CNT] += ss + sc; }
    }
  }

  // ---- the state vector (test_env.py:527-583) except its table columns ----------------------------------------
  if (p.state_out) {
    const int S = p.S;
    const size_t total = (size_t)N * S;
    const int act_w = (p.flags & DIRAL_F_ACTION_REAL) ? 1 : A;
    for (size_t e = tid; e < total; e += kLargeThreads) {
      const int u = (int)(e / S), s = (int)(e - (size_t)u * S);
      double val = 0.0;
      bool write = true;
      if (p.off_act >= 0 && s >= p.off_act && s < p.off_act + act_w) {
        val = (p.flags & DIRAL_F_ACTION_REAL) ? (double)s_act[u] : ((s_act[u] == s - p.off_act) ? 1.0 : 0.0);   // test_env.py:585-595
      } else if (p.off_chobs >= 0 && s >= p.off_chobs && s < p.off_chobs + A) {
        if (do_step) write = false;                                   // written above
        else val = p.chobs_in ? p.chobs_in[(bN + u) * A + (s - p.off_chobs)] : 0.0;
      } else if (p.off_hist >= 0 && s >= p.off_hist && s < p.off_hist + K) {
        write = !(piggy && (p.posdist_type == 2 || p.posdist_type == 1));   // large_hist_kernel / posdist_kernel
      } else if (p.off_posdist >= 0 && s >= p.off_posdist && s < p.off_posdist + N - 1) {
        write = false;                                                // posdist_kernel
      } else if (s == p.off_rew) {
        val = do_step ? g.rew[bN + u] : (p.rew_in ? p.rew_in[bN + u] : 0.0);
      } else if (s == p.off_idx) {
        val = (double)(u + 1);
      } else if (p.off_pos >= 0 && s == p.off_pos) {
        val = p.pos_x[bN + u] / p.L;                                  // network.py:403-407 (behind the move)
      } else if (p.off_pos >= 0 && s == p.off_pos + 1) {
        val = s_py[u] / p.H;
      } else if (s == p.off_vel) {
        val = p.vel[bN + u];
      } else if (p.off_fp >= 0 && s == p.off_fp) {
        val = p.episode;
      } else if (p.off_fp >= 0 && s == p.off_fp + 1) {
        val = p.eps;
      }
      if (write) store_out(p.state_out, bN * S + e, val, out_f64);
    }
  }
}

// One table column (subject k of env b) by one wave, any N: 64-bit keys (sequence number, source viewer) in the wave's LDS
// slice `key` [NP + NP / 64].
__device__ inline void large_merge_column(const StepParams& p, const LargeScratch& g, int b, int k, int lane,
                                          unsigned long long* key) {
  const int N = p.N, A = p.A, NV = p.NV;
  const int NP = (N + 63) & ~63;
  unsigned long long* s_msk = key + NP;
  double* xs = reinterpret_cast<double*>(key);
  const size_t bN = (size_t)b * N, bA = (size_t)b * A;
  const size_t row = ((size_t)b * p.NR + k) * NV;
  bool seq_ovf = false;

  // Vehicle.periodic_update (vehicle.py:56-70) of this column's entries
  auto stamp = [&](unsigned int w, int u) -> unsigned int {
    const bool own = u == k;
    const unsigned int seq = (w >> 8) + (own ? 1u : 0u);
    const unsigned int a0 = w & 255u;
    const unsigned int age = own ? 0u : (a0 + (a0 < 255u ? 1u : 0u));
    seq_ovf = seq_ovf || (own && seq >= (1u << 24) - 1u);
    return (seq << 8) | age;
  };
  for (int vb = 0; vb < NP; vb += 64) {
    const int u = vb + lane;
    const unsigned int w = u < N ? stamp(p.tkey[row + u], u) : 0u;
    key[u] = ((unsigned long long)(w >> 8) << 32) | (unsigned int)u;
  }
  wave_lds_order();
  // Vehicle.received_update (vehicle.py:35-47) for every (resource, receiver) in reference order: the transmitters of
  // resource i do not merge on i (test_env.py:204-209), so the entries read in step i are not written in step i
  const int na = (int)g.nact[b];
  for (int qa = 0; qa < na; ++qa) {
    const int i = g.alist[bA + qa];
    const unsigned short* src = g.src + (bA + i) * N;
    // eight chunks of 64 viewers at a time: their sources, then their gathers, then their writes - one chunk after the
    // other is a chain of round trips (HBM, LDS, LDS) per chunk
    constexpr int G = 8;
    for (int vb = 0; vb < NP; vb += 64 * G) {
      int m[G];
      unsigned long long v[G], mine[G];
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int u = vb + c * 64 + lane;
        m[c] = u < N ? (int)src[u] : (u < NP ? u : NP - 1);
      }
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int u = vb + c * 64 + lane;
        v[c] = key[m[c]];
        mine[c] = key[u < NP ? u : NP - 1];
      }
      wave_lds_order();
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const int u = vb + c * 64 + lane;
        if (u < N && v[c] > mine[c]) key[u] = v[c];
      }
    }
    wave_lds_order();
  }
  // xpos follows the winning sequence number (from the source viewer's entry as the slot found it, or the subject's own
  // stamp), last_updated resets where the entry changed
  const double pxk = g.px0[bN + k];
  for (int vb = 0; vb < NP; vb += 64) {
    const int u = vb + lane;
    const bool valid = u < N;
    bool wr = false;
    double xg = 0.0;
    if (valid) {
      const unsigned int ws = stamp(p.tkey[row + u], u);
      const unsigned long long kf = key[u];
      const unsigned int seqf = (unsigned int)(kf >> 32), sv = (unsigned int)kf;
      const bool upd = seqf != (ws >> 8);
      p.tkey[row + u] = upd ? (seqf << 8) : ws;
      if (u == k) { xg = pxk; wr = true; }                            // vehicle.py:63
      else if (upd) { xg = (int)sv == k ? pxk : p.tx[row + sv]; wr = true; }
    }
    wave_lds_order();
    xs[u] = xg;                                                       // (the key of this viewer is consumed)
    const unsigned long long mk = __ballot(wr);
    if (lane == 0) s_msk[vb >> 6] = mk;
  }
  wave_lds_order();
  for (int vb = 0; vb < NP; vb += 64) {
    const int u = vb + lane;
    const unsigned long long mk = s_msk[vb >> 6];
    if ((mk >> lane) & 1ull) p.tx[row + u] = xs[u];
  }
  wave_lds_order();
  if (seq_ovf) atomicOr(p.err, kErrSeq);
}

// Any N: one wave per table column.  grid = B * ceil(N / W), W = large_merge_waves(N) waves per workgroup.
__global__ __launch_bounds__(256) void large_merge_kernel(const StepParams p, const LargeScratch g) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = p.N;
  const int NP = (N + 63) & ~63;
  const int W = large_merge_waves(N);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform for the compiler too: scalar row bases)
  const int nblk = (N + W - 1) / W;
  const int b = blockIdx.x / nblk, k = (blockIdx.x - b * nblk) * W + wave;
  if (k >= N) return;                                                 // (no workgroup barrier)
  large_merge_column(p, g, b, k, lane, reinterpret_cast<unsigned long long*>(smem) + (size_t)wave * (NP + NP / 64));
}

// N <= 64 CH (CH = 4, 8, 16): NC = 2 or 4 columns per wave, the wave's own keys in registers, LDS only as the gather target.
// A key is 32 bits: (rank << 12) | source viewer, rank = 2^20 - 1 - (the subject's own number - the entry's number), 0 for
// a never-heard entry - the same order as the numbers while every heard entry lags its subject by less than 2^20 - 1
// stamps; a column group holding an older entry (imported tables) takes large_merge_column.  Per (resource, 64 viewers):
// one 16-bit load (shared by the NC columns), one ds_read_b64 / b128, NC v_max_u32, one ds_write_b64 / b128; the next
// resource's sources are in flight meanwhile.  grid = B * ceil(ceil(N / NC) / 4), 256 threads.
constexpr unsigned int kLargeRankMax = (1u << 20) - 1u;
#ifndef DIRAL_LARGE_RANK_AHEAD
#define DIRAL_LARGE_RANK_AHEAD 2          // large_mergen_kernel<16, 2>: resources whose gather sources are fetched ahead - 1024 / 64 / B = 256:
                                          // 4.96 / 4.40 / 7.23 ms for 1 / 2 / 4 (interleaved variant libraries; 4 spills)
#endif
template <int NC> struct LargeKeyVec;
template <> struct LargeKeyVec<2> { typedef uint2 type; };
template <> struct LargeKeyVec<4> { typedef uint4 type; };
__device__ inline unsigned int large_kv(const uint2& v, int j) { return j ? v.y : v.x; }
__device__ inline unsigned int large_kv(const uint4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ inline uint2 large_mk(const unsigned int (&k)[2]) { return make_uint2(k[0], k[1]); }
__device__ inline uint4 large_mk(const unsigned int (&k)[4]) { return make_uint4(k[0], k[1], k[2], k[3]); }
// (waves per SIMD the register allocation is held to: left alone the compiler takes 108 VGPRs for <8, 2> - four waves -
// where 80 do: 512 / 64 / B = 1024 7.0 -> 6.1 ms)
// GATED: the launch behind large_mergec_kernel - only the flagged column pairs, and the hint for the next slot at the end
template <int CH, int NC, bool GATED>
__global__ __launch_bounds__(256, (CH * NC <= 16 ? 6 : 3)) void large_mergen_kernel(const StepParams p, const LargeScratch g) {
  static_assert(!GATED || NC == 2, "the flags are per column pair");
  typedef typename LargeKeyVec<NC>::type kv_t;
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NP = 64 * CH;
  const int N = p.N, A = p.A, NV = p.NV;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform for the compiler too: scalar row bases)
  const int ngrp = (N + NC - 1) / NC, nblk = (ngrp + 3) >> 2;
  const int b = blockIdx.x / nblk, k0 = NC * ((blockIdx.x - b * nblk) * 4 + wave);
  if (k0 >= N) return;                                                // (no workgroup barrier)
  // behind large_mergec_kernel (NC == 2): only the column pairs of the quads it left alone (an entry beyond the codes)
  unsigned char* const my_flag = g.qflag + (size_t)b * ((N + 1) >> 1) + (k0 >> 1);
  if constexpr (GATED) { if (*my_flag == 0) return; }
  // the wave's slice: NP key vectors (as large_merge_column's 64-bit form: at least 8 NP + 8 CH bytes)
  unsigned char* const slice = smem + (size_t)wave * large_mergen_slice(CH, NC);
  kv_t* const kl = reinterpret_cast<kv_t*>(slice);
  const int ncol = N - k0 < NC ? N - k0 : NC;                         // (the rows k0 .. k0 + NC - 1 exist either way: NR is N rounded up to 16)
  const size_t bN = (size_t)b * N, bA = (size_t)b * A;
  const size_t row0 = ((size_t)b * p.NR + k0) * NV;
  unsigned int tko[NC];                                               // the subjects' own numbers behind this slot's stamp
#pragma unroll
  for (int j = 0; j < NC; ++j) tko[j] = j < ncol ? (p.tkey[row0 + (size_t)j * NV + k0 + j] >> 8) + 1u : 0u;
  unsigned int key[CH][NC];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int u = c * 64 + lane;
    const bool in = u < N;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const unsigned int w = (in && j < ncol) ? p.tkey[row0 + (size_t)j * NV + u] : 0u;
      const unsigned int sq = (w >> 8) + ((j < ncol && u == k0 + j) ? 1u : 0u);
      const unsigned int lg = tko[j] - sq;
      bad = bad || (sq != 0u && lg >= kLargeRankMax) || sq >= (1u << 24) - 1u;
      key[c][j] = ((sq != 0u ? kLargeRankMax - lg : 0u) << 12) | (unsigned int)u;
    }
  }
  if (__ballot(bad) != 0ull) {                                        // (uniform) old entries, or a number about to overflow: the 64-bit form
    for (int j = 0; j < ncol; ++j) large_merge_column(p, g, b, k0 + j, lane, reinterpret_cast<unsigned long long*>(slice));
    return;                                                           // (the pair stays flagged for the next slot)
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) kl[c * 64 + lane] = large_mk(key[c]);
  wave_lds_order();
  const int na = (int)g.nact[b];
  const unsigned short* const alist = g.alist + bA;
  // the gather sources of D resources at a time, the next D in flight (large_mergec_kernel: the rows are 1-2 us away)
  constexpr int D = CH >= 16 ? DIRAL_LARGE_RANK_AHEAD : 1;          // (the gated forms sit at their 80-VGPR cap)
  int cur[D][CH], nxt[D][CH];
  auto load_group = [&](int q0, int (&dst)[D][CH]) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int qa = q0 + d < na ? q0 + d : na - 1;
      const unsigned short* src = g.src + (bA + alist[qa]) * N;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int u = c * 64 + lane;
        dst[d][c] = u < N ? (int)src[u] : u;
      }
    }
  };
  if (na > 0) load_group(0, cur);
  for (int q0 = 0; q0 < na; q0 += D) {
    if (q0 + D < na) load_group(q0 + D, nxt);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (q0 + d < na) {                                              // (uniform)
        kv_t v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = kl[cur[d][c]];
        // (the transmitters of a resource do not merge on it, test_env.py:204-209: every gather of the step may precede its writes)
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
          for (int j = 0; j < NC; ++j) key[c][j] = max(key[c][j], large_kv(v[c], j));
          kl[c * 64 + lane] = large_mk(key[c]);
        }
        wave_lds_order();
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int c = 0; c < CH; ++c) cur[d][c] = nxt[d][c];
  }
  // back to numbers; xpos from the source viewer's entry as the slot found it (every gather before the first store)
  bool stale = false;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    if (j >= ncol) break;
    const int k = k0 + j;
    const size_t row = row0 + (size_t)j * NV;
    const double pxk = g.px0[bN + k];
    double xg[CH];
    unsigned int wrm = 0u;                                            // bit c: this lane writes the xpos of chunk c
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int u = c * 64 + lane;
      xg[c] = 0.0;
      if (u < N) {
        const unsigned int w = p.tkey[row + u];
        const bool own = u == k;
        const unsigned int a0 = w & 255u;
        const unsigned int so = (w >> 8) + (own ? 1u : 0u);
        const unsigned int ws = (so << 8) | (own ? 0u : (a0 + (a0 < 255u ? 1u : 0u)));   // vehicle.py:56-70
        const unsigned int kf = key[c][j];
        const unsigned int rank = kf >> 12, sv = kf & 4095u;
        const unsigned int seqf = rank ? tko[j] - (kLargeRankMax - rank) : 0u;
        if constexpr (GATED) stale = stale || (rank != 0u && kLargeRankMax - rank >= 7u);   // the next slot's stamp takes it beyond the codes
        const bool upd = seqf != so;
        p.tkey[row + u] = upd ? (seqf << 8) : ws;
        if (own) { xg[c] = pxk; wrm |= 1u << c; }                    // vehicle.py:63
        else if (upd) { xg[c] = (int)sv == k ? pxk : p.tx[row + sv]; wrm |= 1u << c; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every old xpos is here before the first new one leaves
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if ((wrm >> c) & 1u) p.tx[row + c * 64 + lane] = xg[c];
    asm volatile("" ::: "memory");
  }
  // the hint large_mergec_kernel reads next slot: does this pair still hold an entry its codes would not reach
  if constexpr (GATED) {
    const bool st = __ballot(stale) != 0ull;
    if (lane == 0) *my_flag = st ? 1 : 0;
  }
}

// N <= 512, the common case of a table that stays fresh: FOUR columns per wave as ONE 32-bit word per viewer - byte j the
// thermometer code of the entry's lag behind subject k0 + j, (0xff << lag) & 0xff for lag 0 ... 7, 0 = never heard
// (step_fast64.hpp's codes, derived here from the (seq << 8 | age) plane every slot).  The codes of one subject form a chain
// under bit inclusion, so Vehicle.received_update (vehicle.py:35-47) is a bitwise OR of words: per (resource, 64 viewers) one
// 16-bit load shared by the four columns, one ds_read_b32 gather, one v_or_b32, one ds_write_b32 - a quarter of the LDS
// bytes of the rank keys, whose gathers pace large_mergen_kernel.  xpos is a function of (subject, number): the pre-slot
// entries of a column fill an 8-entry table [lag] in LDS (every viewer writes its own entry's xpos to the slot of its lag:
// equal lags carry equal xpos), a changed entry reads it there.  A quad with a heard entry 8 or more stamps behind is left
// untouched and flagged (`qflag`, per column pair): large_mergen_kernel<CH, 2, true> runs behind this launch for exactly those,
// and leaves a hint for the next slot - a pair that still holds such an entry is not looked at here again until it is fresh.
// grid = B * ceil(ceil(N / 4) / 4), 256 threads.
template <int CH>
__global__ __launch_bounds__(256, (CH <= 8 ? 6 : 4)) void large_mergec_kernel(const StepParams p, const LargeScratch g) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NP = 64 * CH;
  const int N = p.N, A = p.A, NV = p.NV;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform for the compiler too: scalar row bases)
  const int nquad = (N + 3) >> 2, nblk = (nquad + 3) >> 2;
  const int b = blockIdx.x / nblk, q = (blockIdx.x - b * nblk) * 4 + wave, k0 = 4 * q;
  if (k0 >= N) return;                                                // (no workgroup barrier)
  // one flag per column PAIR (what a wave of large_mergen_kernel<CH, 2> owns).  As this launch finds them they are last
  // slot's hints: a pair that still held an entry 7 or more stamps behind after its merge (large_mergen_kernel's last
  // statement) cannot be clean now - the quad goes to the rank keys without a look at its words.  (Hints only: any value
  // is correct, a wrong one costs the detour.)
  unsigned char* const qf = g.qflag + (size_t)b * ((N + 1) >> 1) + 2 * q;
  const bool two_pairs = k0 + 2 < N;
  if ((qf[0] | (two_pairs ? qf[1] : 0)) != 0) {
    if (lane == 0) { qf[0] = 1; if (two_pairs) qf[1] = 1; }
    return;
  }
  unsigned int* const kl = reinterpret_cast<unsigned int*>(smem) + (size_t)wave * (NP + 64);
  double* const xtab = reinterpret_cast<double*>(kl + NP);            // [4][8] xpos by (column, lag)
  const int ncol = N - k0 < 4 ? N - k0 : 4;                           // (the rows k0 .. k0 + 3 exist either way: NR is N rounded up to 16)
  const size_t bN = (size_t)b * N, bA = (size_t)b * A;
  const size_t row0 = ((size_t)b * p.NR + k0) * NV;
  unsigned int tko[4];                                                // the subjects' own numbers behind this slot's stamp
#pragma unroll
  for (int j = 0; j < 4; ++j) tko[j] = j < ncol ? (p.tkey[row0 + (size_t)j * NV + k0 + j] >> 8) + 1u : 0u;
  unsigned int code[CH];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int u = c * 64 + lane;
    const bool in = u < N;
    unsigned int cw = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = in && j < ncol;
      const unsigned int w = live ? p.tkey[row0 + (size_t)j * NV + u] : 0u;
      const unsigned int sq = (w >> 8) + ((live && u == k0 + j) ? 1u : 0u);   // Vehicle.periodic_update, vehicle.py:56-70
      const unsigned int lg = tko[j] - sq;
      const bool heard = sq != 0u;
      bad = bad || (heard && lg > 7u) || sq >= (1u << 24) - 1u;
      cw |= (heard ? ((0xffu << (lg & 7u)) & 0xffu) : 0u) << (8 * j);
      // the xpos this number carries, for whoever receives it in this slot (lag 0 is the subject's own stamp)
      if (live && heard && lg >= 1u && lg <= 7u) xtab[j * 8 + lg] = p.tx[row0 + (size_t)j * NV + u];
    }
    code[c] = cw;
    // (four chunks' loads in flight at a time: all 16 x 8 of a 1024-vehicle quad cost 59 spilled VGPRs and 345 SGPRs)
    if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  const bool flagged = __ballot(bad) != 0ull;
  if (lane == 0) { qf[0] = flagged ? 1 : 0; if (two_pairs) qf[1] = flagged ? 1 : 0; }
  if (flagged) return;                                                // (uniform; nothing of the quad has been written)
#pragma unroll
  for (int c = 0; c < CH; ++c) kl[c * 64 + lane] = code[c];
  wave_lds_order();
  const int na = (int)g.nact[b];
  const unsigned short* const alist = g.alist + bA;
  // The gather sources of D resources at a time, the next D already in flight: one resource ahead (what large_mergen_kernel
  // does) left the wave waiting for the row - in a timing probe with every resource reading ONE cached row the kernel ran in
  // 1.08 ms instead of 1.87 (C3's shape): the rows come from L2 / the Infinity Cache, 1-2 us away.
  constexpr int D = 4;                                               // (C3 shape: 1.87 / 1.61 / 1.75 ms for 1 / 4 / 8 resources ahead; 512 vehicles: 3.27 / 3.00 / 2.85 for 1 / 2 / 4)
  int cur[D][CH], nxt[D][CH];
  auto load_group = [&](int q0, int (&dst)[D][CH]) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int qa = q0 + d < na ? q0 + d : na - 1;                   // (behind the last resource: its row again, not used)
      const unsigned short* src = g.src + (bA + alist[qa]) * N;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int u = c * 64 + lane;
        dst[d][c] = u < N ? (int)src[u] : u;
      }
    }
  };
  if (na > 0) load_group(0, cur);
  for (int q0 = 0; q0 < na; q0 += D) {
    if (q0 + D < na) load_group(q0 + D, nxt);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (q0 + d < na) {                                              // (uniform)
        unsigned int v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = kl[cur[d][c]];
        // (the transmitters of a resource do not merge on it, test_env.py:204-209: every gather of the step may precede its writes)
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          code[c] |= v[c];
          kl[c * 64 + lane] = code[c];
        }
        wave_lds_order();
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int c = 0; c < CH; ++c) cur[d][c] = nxt[d][c];
  }
  // back to numbers and ages; a changed entry takes the xpos of its new number.  (One column at a time, the loop rolled:
  // unrolled four times the 1024-vehicle form spilled 47 VGPRs and 354 SGPRs.)
#pragma unroll 1
  for (int j = 0; j < ncol; ++j) {
    const int k = k0 + j;
    const size_t row = row0 + (size_t)j * NV;
    const double pxk = g.px0[bN + k];
    const unsigned int tko_j = (p.tkey[row + k] >> 8) + 1u;           // (read before this column's own entry is rewritten below)
    const int sh = 8 * j;
    const double* const xt = xtab + j * 8;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int u = c * 64 + lane;
      if (u < N) {
        const unsigned int w = p.tkey[row + u];
        const bool own = u == k;
        const unsigned int a0 = w & 255u;
        const unsigned int so = (w >> 8) + (own ? 1u : 0u);
        const unsigned int ws = (so << 8) | (own ? 0u : (a0 + (a0 < 255u ? 1u : 0u)));
        const unsigned int byte = (code[c] >> sh) & 0xffu;
        const unsigned int lagn = 8u - (unsigned int)__popc(byte);           // (byte != 0 wherever it is used)
        const unsigned int seqf = byte ? tko_j - lagn : 0u;
        const bool upd = seqf != so;
        if (upd) p.tkey[row + u] = seqf << 8;
        else if (ws != w) p.tkey[row + u] = ws;
        if (own) p.tx[row + u] = pxk;                                       // vehicle.py:63
        else if (upd) p.tx[row + u] = lagn == 0u ? pxk : xt[lagn];
      }
    }
  }
}

// The type-2 piggybacked histogram of 64 (or fewer) viewers per workgroup: four waves share the subjects (wave w takes k = w,
// w + 4, ...: B * N / 64 waves alone are four per SIMD at 1024 vehicles - too few to cover the table loads), each with its
// own rows of counters, added up behind one barrier.  grid = B * ceil(N / VW), 256 threads.
constexpr int kLargeHistWaves = 4;
__global__ __launch_bounds__(64 * kLargeHistWaves) void large_hist_kernel(const StepParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = p.N, K = p.K, NV = p.NV, KP = K | 1;
  const int VW = large_hist_viewers(K);
  double* s_edges = reinterpret_cast<double*>(smem);
  unsigned int* s_n = reinterpret_cast<unsigned int*>(smem + 8u * (K + 2));       // [waves][64]
  unsigned int* s_hist = s_n + 64 * kLargeHistWaves;                              // [waves][VW][KP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N + VW - 1) / VW;
  const int b = blockIdx.x / nblk, vb0 = (blockIdx.x - b * nblk) * VW;
  const int u = vb0 + lane;
  const bool mine = lane < VW && u < N;
  const size_t bN = (size_t)b * N, bR = (size_t)b * p.NR;
  for (int j = tid; j <= K; j += 64 * kLargeHistWaves) s_edges[j] = p.edges[j];
  for (int j = tid; j < kLargeHistWaves * VW * KP; j += 64 * kLargeHistWaves) s_hist[j] = 0u;
  __syncthreads();
  const double x2 = mine ? p.pos_x[bN + u] : 0.0, y2 = mine ? p.pos_y[bN + u] : 0.0;
  unsigned int* hrow = s_hist + ((size_t)wave * VW + lane) * KP;
  unsigned int cnt = 0u;
  const size_t col = mine ? (size_t)u : 0;
#pragma unroll 4
  for (int k = wave; k < N; k += kLargeHistWaves) {
    const size_t idx = (bR + k) * NV + col;
    const unsigned int w = p.tkey[idx];
    const double x1 = p.tx[idx];
    const double pyk = p.pos_y[bN + k];
    // Network.dist_piggy + get_positional_dist_2_piggy (network.py:538-558, 473-513)
    if (mine && u != k && (int)(w & 255u) < p.age_limit) {
      const double y1 = (w >> 8) ? pyk : 0.0;
      const double d = dist2d(x1, y1, x2, y2);
      if (d < p.Rb) {
        const double v = (x1 - x2 > 0.0) ? d : -d;
        hrow[hist_bin(v, -p.Rb, p.hist_inv_width, K, s_edges)] += 1u;
        cnt += 1u;
      }
    }
  }
  s_n[wave * 64 + lane] = cnt;
  __syncthreads();
  const int rows = min(VW, N - vb0);
  for (int e = tid; e < rows * K; e += 64 * kLargeHistWaves) {
    const int r = e / K, j = e - r * K;
    unsigned int n = 0u, h = 0u;
#pragma unroll
    for (int w = 0; w < kLargeHistWaves; ++w) { n += s_n[w * 64 + r]; h += s_hist[((size_t)w * VW + r) * KP + j]; }
    const double val = n ? (double)h / (double)n : 0.0;                // network.py:501
    store_out(p.state_out, (bN + vb0 + r) * p.S + p.off_hist + j, val, p.out_f64);
  }
}

}  // namespace diral
