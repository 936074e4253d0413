/*
 * diral_oracle.c - CPU restatement of the reference's env hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The
 * product (diral_amd/, libdiral_env.so) never links, imports or calls it.
 *
 * It restates, function by function, what /root/reference/envs does for ONE
 * env in the reference's own data layout (per-viewer neighbour tables), looped
 * over a batch.  Every function cites the reference lines it follows.  Parity
 * is PINNED: tests/test_oracle_golden.py checks it bit-for-bit against the
 * fixtures under tests/golden/ (npz), which tests/golden/gen_golden.py recorded by
 * importing the reference itself.
 *
 * Arithmetic notes (all float64, like CPython floats):
 *  - `x ** 2` in the reference is libm pow(x, 2.0) (CPython float_pow and the
 *    NumPy float64 scalar both call it), which on glibc 2.35 differs from the
 *    correctly rounded x*x by 1 ulp for ~0.08 % of inputs.  sq_mode 0 restates
 *    that (pow); sq_mode 1 uses x*x, which is what the HIP kernels compute.
 *    tests/ prove the two modes agree on every integer/index output and within
 *    1 ulp on distances.
 *  - `%` is CPython/NumPy float remainder: fmod + sign fix-up.
 *  - np.histogram / np.linspace (third-party: NumPy, README pins 1.19.2,
 *    fixtures recorded with 2.2.6) are restated from their published
 *    algorithm: numpy/lib/_histograms_impl.py (uniform-bin fast path and the
 *    cumulative path) and numpy/_core/function_base.py (linspace).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, optional -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/diral_env.h"

typedef struct {
  double *px, *py, *vel;      /* [N]   Vehicle.pos_x/pos_y/velocity vehicle.py:9-14 */
  int32_t *seq, *age;         /* [N][N] viewer-major: pos_of_neighbors[k] vehicle.py:29-33 */
  double *tx, *ty;            /* [N][N] xpos / ypos                                 */
  int64_t *la;                /* [N][N] last_arrival_time[tx][rx] network.py:39-42  */
  int32_t *pf;                /* [N]   pf_counter test_env.py:87-92                 */
  double *prev_obs;           /* [N][A] TestEnv.prev_obs test_env.py:76-79, 260-261 (State.piggybacking) */
  double metrics[DIRAL_M_COLUMNS];
} OEnv;

typedef struct {
  DiralCfg cfg;
  int B, N, A, K, S, sq_mode, threads;
  int piggy_keyerror;         /* a step hit `self.prev_obs[None]` (test_env.py:243): the reference raises KeyError */
  double *edges;              /* [K+1] np.linspace(-Rb, Rb, K+1) */
  double *edges1;             /* [K+1] np.linspace(-1, 1, K+1)   */
  double *trace;              /* [T][N] Network.x_positions (network.py:171-178), NULL = none */
  int trace_len;
  OEnv *envs;
} Oracle;

static inline int has(const Oracle *o, uint32_t f) { return (o->cfg.flags & f) != 0; }

/* the reference's `v ** 2` (see header note) */
static inline double sq(const Oracle *o, double v) {
  if (o->sq_mode == 0) { volatile double two = 2.0; return pow(v, two); }
  return v * v;
}

/* CPython float_rem / NumPy npy_remainder: Python's float `%` */
static double py_mod(double vx, double wx) {
  double mod = fmod(vx, wx);
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) mod += wx;
  } else {
    mod = copysign(0.0, wx);
  }
  return mod;
}

/* np.linspace(start, stop, num) with endpoint=True: numpy/_core/function_base.py
 * y = arange(num) * step + start ; y[-1] = stop  (two roundings per edge) */
static void np_linspace(double start, double stop, int num, double *out) {
  int div = num - 1;
  double delta = stop - start;
  if (div > 0) {
    double step = delta / (double)div;
    for (int i = 0; i < num; ++i) {
      double y = (double)i;
      if (step == 0.0) { y = y / (double)div; y = y * delta; }
      else y = y * step;
      out[i] = y + start;
    }
    out[num - 1] = stop;
  } else if (num == 1) {
    out[0] = start;
  }
}

/* Network.dist (network.py:318-332): no ring wrap */
static inline double dist_xy(const Oracle *o, double x1, double y1, double x2, double y2) {
  return sqrt(sq(o, x2 - x1) + sq(o, y2 - y1));
}
static inline double dist_uu(const Oracle *o, const OEnv *e, int p1, int p2) {
  return dist_xy(o, e->px[p1], e->py[p1], e->px[p2], e->py[p2]);
}

/* Network.periodic_update -> Vehicle.periodic_update (network.py:587-593,
 * vehicle.py:56-70).  `transmitted` aliases the live table (vehicle.py:61), so
 * nothing is snapshotted. */
static void periodic_update(const Oracle *o, OEnv *e) {
  int N = o->N;
  for (int u = 0; u < N; ++u) {
    int32_t *seq = e->seq + (size_t)u * N, *age = e->age + (size_t)u * N;
    seq[u] += 1;
    e->tx[(size_t)u * N + u] = e->px[u];
    e->ty[(size_t)u * N + u] = e->py[u];
    for (int k = 0; k < N; ++k) {
      if (k == u) age[k] = 0; else age[k] += 1;
    }
  }
}

/* Network.received_update -> Vehicle.received_update (network.py:576-585,
 * vehicle.py:35-47): strict-greater sequence number wins, age reset */
static void received_update(const Oracle *o, OEnv *e, int rx, int tx) {
  int N = o->N;
  size_t r = (size_t)rx * N, t = (size_t)tx * N;
  for (int k = 0; k < N; ++k) {
    if (e->seq[t + k] > e->seq[r + k]) {
      e->tx[r + k] = e->tx[t + k];
      e->ty[r + k] = e->ty[t + k];
      e->seq[r + k] = e->seq[t + k];
      e->age[r + k] = 0;
    }
  }
}

/* Network.find_closest_tx (network.py:378-398): strict '<' twice => the lowest
 * id wins ties; out-of-range tx stamp last_arrival_time = -1 (line 394) */
static double find_closest_tx(const Oracle *o, OEnv *e, const int *txs, int ntx,
                              int rx, int *tx_id) {
  double min_dist = 100000.0;
  int min_id = -1;
  for (int j = 0; j < ntx; ++j) {
    int tx = txs[j];
    double d = dist_uu(o, e, tx, rx);
    if (d < o->cfg.communication_range) {
      if (d < min_dist) { min_dist = d; min_id = tx; }
    } else {
      e->la[(size_t)tx * o->N + rx] = -1;
    }
  }
  *tx_id = min_id;
  return min_dist;
}

/* Network.calculate_avg_distance (network.py:307-316): itertools.combinations
 * order, left-to-right sum starting from int 0 */
static double avg_distance(const Oracle *o, const OEnv *e, const int *users, int n) {
  double s = 0.0;
  int cnt = 0;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b) { s = s + dist_uu(o, e, users[a], users[b]); ++cnt; }
  return s / (double)cnt;
}

/* Network.calculate_norm (network.py:225-246) */
static double calculate_norm(const Oracle *o, const OEnv *e) {
  double L = o->cfg.highway_length;
  double x_min = L + 1, x_max = -L - 1;
  int umin = -1, umax = -1;
  for (int u = 0; u < o->N; ++u) {
    if (e->px[u] < x_min) { x_min = e->px[u]; umin = u; }
    if (e->px[u] > x_max) { x_max = e->px[u]; umax = u; }
  }
  return dist_uu(o, e, umin, umax);
}

/* Network.calculate_reward_weights (network.py:273-300) -> weights[0] */
static int reward_weight(const Oracle *o, const OEnv *e, const int *users, int n) {
  double m = avg_distance(o, e, users, n);
  if (has(o, DIRAL_F_TOY_WEIGHTS)) return m == calculate_norm(o, e);
  return m > o->cfg.communication_range;
}

/* TestEnv.calculate_reward_design (test_env.py:319-349) with
 * Network.is_comm_range / calculate_reward_weights_design (network.py:122-157) */
static double reward_design_tx(const Oracle *o, const OEnv *e, int tx_user,
                               const int *txs, int ntx) {
  int grp[2];
  int n = 1;
  grp[0] = tx_user;
  for (int j = 0; j < ntx; ++j) {
    int other = txs[j];
    if (other == tx_user) continue;
    if (dist_uu(o, e, tx_user, other) < 2 * o->cfg.communication_range) {
      if (n < 2) grp[n] = other;
      ++n;
    }
  }
  if (n == 1) return 1.0;
  if (n == 2) {
    double m = avg_distance(o, e, grp, 2);
    int w = (m > o->cfg.communication_range * 2) ? 1 : 0;
    return w == 1 ? 0.0 : -2.0;
  }
  return -(double)n;
}

/* Network.update_mobility / update_positions (network.py:302-305, 189-206);
 * all vehicles drive "right" (network.py:101,111) */
static void update_mobility(const Oracle *o, OEnv *e, int64_t timestep) {
  if (!has(o, DIRAL_F_MOBILITY)) return;
  if (o->trace) {                                   /* network.py:194-199: replay, t % len */
    int64_t tt = timestep % o->trace_len;
    if (tt < 0) tt += o->trace_len;                 /* Python's % */
    for (int u = 0; u < o->N; ++u) e->px[u] = o->trace[(size_t)tt * o->N + u];
    return;
  }
  double L = o->cfg.highway_length;
  for (int u = 0; u < o->N; ++u) e->px[u] = py_mod(e->px[u] + e->vel[u] + L, L);
}

/* one env, one slot: my_step / my_step_ch / my_step_design
 * (test_env.py:124-266 / 351-443 / 269-349) */
/* np.insert(arr, i, vals) for 1-D arrays (test_env.py:247, 254): `m` values in front of index i */
static void np_insert(double *arr, int *len, int i, const double *vals, int m) {
  memmove(arr + i + m, arr + i, sizeof(double) * (size_t)(*len - i));
  if (vals) memcpy(arr + i, vals, sizeof(double) * (size_t)m);
  else for (int j = 0; j < m; ++j) arr[i + j] = 0.0;
  *len += m;
}

/* `chobs_out`: the dict my_step* returns, [N][A] - or, with State.piggybacking in my_step, piggy_obs [N][A * A]
 * (test_env.py:263-264).  Returns 1 when the reference would have raised KeyError (prev_obs[None], :243). */
static int step_env(const Oracle *o, OEnv *e, int mode, const int32_t *act,
                    int64_t t, double *rews, double *chobs_out) {
  const int N = o->N, A = o->A;
  const int rd = o->cfg.reward_design;
  const int piggy = has(o, DIRAL_F_ADD_POSDIST_PIGGY);
  const int pb = has(o, DIRAL_F_PIGGYBACKING) && mode == DIRAL_STEP_MY_STEP;
  int keyerror = 0;
  int *txs = (int *)malloc(sizeof(int) * (size_t)N);
  char *is_tx = (char *)malloc((size_t)N);
  double *r_tx = (double *)malloc(sizeof(double) * (size_t)N);
  /* piggybacking: `obs` (plain, what becomes prev_obs) and `piggy_obs` (returned) are two dicts (test_env.py:134-135);
   * a piggy_obs[user] array grows by A per insert, A * A at most */
  double *chobs = pb ? (double *)malloc(sizeof(double) * (size_t)N * A) : chobs_out;
  double *pobs = pb ? (double *)calloc((size_t)N * (A * A + A), sizeof(double)) : NULL;
  int *plen = pb ? (int *)malloc(sizeof(int) * (size_t)N) : NULL;
  const size_t pstride = (size_t)A * A + A;
  for (int u = 0; u < N; ++u) rews[u] = 0.0;
  for (size_t j = 0; j < (size_t)N * A; ++j) chobs[j] = 0.0;
  if (pb) for (int u = 0; u < N; ++u) plen[u] = A;    /* np.zeros((action_space,)) test_env.py:145 */

  if (piggy) periodic_update(o, e);                    /* test_env.py:138-139 */

  for (int i = 0; i < A; ++i) {
    int tot = 0;
    for (int u = 0; u < N; ++u) {                       /* test_env.py:153-157 */
      is_tx[u] = (act[u] == i);
      if (is_tx[u]) txs[tot++] = u;
    }
    double rewards = 0.0;
    if (mode == DIRAL_STEP_MY_STEP && tot > 1) {        /* test_env.py:163-199 */
      if (rd == 1) {
        int w = reward_weight(o, e, txs, tot);
        double R = (double)w / (double)tot;
        rewards = -1 * (1 - R);
      } else if (rd == 2) {
        if (tot == 2) rewards = 2 * reward_weight(o, e, txs, tot) - (double)tot;
        else rewards = 0 - (double)tot;
      } else if (rd == 3) {
        double R = 1 / (double)tot;
        rewards = -1 * exp(1 - R);
      } else if (rd == 4) {
        rewards = 1 / (double)tot;
      } else if (rd == 5) {
        if (tot == 2) rewards = (reward_weight(o, e, txs, tot) == 1) ? 0 : -1;
        else rewards = -1;
      }
    }
    /* PRR ratio R per colliding tx.  my_step_ch always; my_step only when the
     * build-extension flag DIRAL_F_TRACK_PRR asks for the metric (find_closest_tx's
     * arrival side effect is idempotent, so this changes no reference state) */
    const int want_prr = (mode == DIRAL_STEP_MY_STEP_CH) ||
                         (mode == DIRAL_STEP_MY_STEP && has(o, DIRAL_F_TRACK_PRR));
    if (want_prr && tot > 1) {                          /* test_env.py:384-405 */
      for (int j = 0; j < tot; ++j) {
        int tx = txs[j], received = 0, in_range = 0;
        for (int rx = 0; rx < N; ++rx) {
          if (is_tx[rx]) continue;
          if (!(dist_uu(o, e, tx, rx) < o->cfg.communication_range)) continue;
          ++in_range;
          int nearest;
          find_closest_tx(o, e, txs, tot, rx, &nearest);
          if (tx == nearest) ++received;
        }
        r_tx[tx] = in_range > 0 ? (double)received / (double)in_range : 1.0;
      }
    }
    for (int u = 0; u < N; ++u) {
      if (is_tx[u]) {
        chobs[(size_t)u * A + i] = 0.0;                 /* half duplex */
        if (pb) pobs[u * pstride + i] = 0.0;            /* test_env.py:209: index i of the GROWN array */
        if (mode == DIRAL_STEP_MY_STEP) {               /* test_env.py:211-222 */
          if (tot > 1) {
            rews[u] = rewards;
            if (has(o, DIRAL_F_PROPORTIONAL_FAIR)) {
              if (e->pf[u] > o->cfg.pf_threshold) rews[u] = o->cfg.pf_penalty;
              e->pf[u] += 1;
            }
            e->metrics[DIRAL_M_TX_COLLIDED] += 1;
            if (want_prr) e->metrics[DIRAL_M_PRR_SUM] += r_tx[u];
          } else {
            rews[u] = 1.0;
            if (has(o, DIRAL_F_PROPORTIONAL_FAIR)) e->pf[u] = 0;
            e->metrics[DIRAL_M_TX_SOLE] += 1;
            if (want_prr) e->metrics[DIRAL_M_PRR_SUM] += 1.0;
          }
          if (want_prr) e->metrics[DIRAL_M_PRR_CNT] += 1;
        } else if (mode == DIRAL_STEP_MY_STEP_CH) {     /* test_env.py:411-429 */
          if (tot > 1) {
            double R = r_tx[u];
            if (rd == 3) rews[u] = 1 - exp(1 - R);
            else if (rd == 4) rews[u] = -1 * exp(1 - R);
            else if (rd == 2) rews[u] = -1 * (1 - R);
            e->metrics[DIRAL_M_TX_COLLIDED] += 1;
            e->metrics[DIRAL_M_PRR_SUM] += R;
          } else {
            if (rd == 3) rews[u] = 1;
            else if (rd == 4) rews[u] = exp(1.0);
            else if (rd == 2) rews[u] = 1;
            e->metrics[DIRAL_M_TX_SOLE] += 1;
            e->metrics[DIRAL_M_PRR_SUM] += 1.0;
          }
          e->metrics[DIRAL_M_PRR_CNT] += 1;
        } else {                                         /* test_env.py:297-301 */
          if (tot == 1) { rews[u] = 1; e->metrics[DIRAL_M_TX_SOLE] += 1; }
          else { rews[u] = reward_design_tx(o, e, u, txs, tot); e->metrics[DIRAL_M_TX_COLLIDED] += 1; }
        }
      } else if (tot > 0) {
        int tx_id;
        double tx_dist = find_closest_tx(o, e, txs, tot, u, &tx_id);
        if (mode == DIRAL_STEP_MY_STEP) {               /* test_env.py:225-240 */
          if (o->cfg.state_type == 1) {
            chobs[(size_t)u * A + i] = 1.0;
            /* reference crashes on tx_id None here; defined as skip */
            if (piggy && tx_id >= 0) received_update(o, e, u, tx_id);
          } else if (o->cfg.state_type == 2) {
            if (piggy && tx_id >= 0) received_update(o, e, u, tx_id);
            chobs[(size_t)u * A + i] = tx_dist;
            if (pb) {                                   /* test_env.py:241-247 */
              if (tx_id < 0) { keyerror = 1; break; }   /* self.prev_obs[None]: KeyError */
              else {
                const double *tmp_a = e->prev_obs + (size_t)tx_id * A;
                pobs[u * pstride + i] = tx_dist;
                np_insert(pobs + u * pstride, &plen[u], i, tmp_a, A);
              }
            }
          }
        } else if (mode == DIRAL_STEP_MY_STEP_CH) {     /* test_env.py:431-439 */
          chobs[(size_t)u * A + i] = 1.0;
          if (tx_id >= 0) {
            e->la[(size_t)tx_id * N + u] = t;
            if (piggy) received_update(o, e, u, tx_id);
          }
        } else {                                         /* test_env.py:305-311 */
          chobs[(size_t)u * A + i] = 1.0;
          if (piggy && tx_id >= 0) received_update(o, e, u, tx_id);
        }
      } else if (pb) {
        /* no transmission on i: A zeros at index i (test_env.py:250-254) */
        np_insert(pobs + u * pstride, &plen[u], i, NULL, A);
      }
    }
    if (keyerror) break;                                /* the exception leaves my_step here */
  }
  if (!keyerror) {
    update_mobility(o, e, t);                           /* test_env.py:259 */
    e->metrics[DIRAL_M_SLOTS] += 1;
    for (int u = 0; u < N; ++u) e->metrics[DIRAL_M_SUM_REWARD] += rews[u];
  }
  if (pb) {
    if (!keyerror) {
      memcpy(e->prev_obs, chobs, sizeof(double) * (size_t)N * A);   /* self.prev_obs = obs, test_env.py:260-261 */
      for (int u = 0; u < N; ++u)                                    /* every array ends A * A long: A - 1 inserts of A */
        memcpy(chobs_out + (size_t)u * A * A, pobs + u * pstride, sizeof(double) * (size_t)A * A);
    }
    free(chobs); free(pobs); free(plen);
  }
  free(txs); free(is_tx); free(r_tx);
  return keyerror;
}

/* Network.dist_piggy (network.py:538-558) */
static int dist_piggy(const Oracle *o, const OEnv *e, int rx_id, int tx_id,
                      double *d, int *sign) {
  size_t idx = (size_t)tx_id * o->N + rx_id;
  if (e->age[idx] < o->cfg.info_age_limit) {
    double x1 = e->tx[idx], y1 = e->ty[idx];
    double x2 = e->px[tx_id], y2 = e->py[tx_id];
    *d = dist_xy(o, x1, y1, x2, y2);
    *sign = (x1 - x2 > 0.0) ? 1 : -1;
    return 1;
  }
  return 0;
}

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* np.histogram(a, K, range=(first,last))[0], uniform fast path
 * (numpy/lib/_histograms_impl.py, "Fast algorithm for equal bins") */
static void np_histogram_uniform(const double *a, int n, int K, double first,
                                 double last, const double *edges, int64_t *out) {
  double norm_denom = last - first;
  for (int j = 0; j < K; ++j) out[j] = 0;
  for (int j = 0; j < n; ++j) {
    double v = a[j];
    if (!(v >= first && v <= last)) continue;
    double f = ((v - first) / norm_denom) * (double)K;
    int64_t idx = (int64_t)f;
    if (idx == K) idx -= 1;
    if (v < edges[idx]) idx -= 1;
    if (v >= edges[idx + 1] && idx != K - 1) idx += 1;
    out[idx] += 1;
  }
}

/* Network.get_positional_dist_2_piggy (network.py:473-513) */
static void posdist_piggy2(const Oracle *o, const OEnv *e, int tx_user, double *out) {
  int N = o->N, K = o->K, n = 0;
  double *vals = (double *)malloc(sizeof(double) * (size_t)N);
  int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
  for (int user = 0; user < N; ++user) {
    if (user == tx_user) continue;
    double d; int sign;
    if (dist_piggy(o, e, user, tx_user, &d, &sign)) {
      if (d < o->cfg.bin_range) vals[n++] = d * sign;
    }
  }
  if (n > 0) {
    qsort(vals, (size_t)n, sizeof(double), cmp_double);
    np_histogram_uniform(vals, n, K, -o->cfg.bin_range, o->cfg.bin_range, o->edges, cnt);
    for (int j = 0; j < K; ++j) out[j] = (double)cnt[j] / (double)n;
  } else {
    for (int j = 0; j < K; ++j) out[j] = 0.0;
  }
  free(vals); free(cnt);
}

/* Network.get_positional_dist_piggy (network.py:432-471): inf-norm scaled,
 * explicit edges + weights => NumPy's cumulative path:
 *   cw = [0, cumsum(sorted w)]; idx = searchsorted(sa, edges[:-1],'left') ++
 *   searchsorted(sa, edges[-1],'right'); n = diff(cw[idx]) */
static void posdist_piggy1(const Oracle *o, const OEnv *e, int tx_user, double *out) {
  int N = o->N, K = o->K, n = 0;
  double *vals = (double *)malloc(sizeof(double) * (size_t)(N + 1));
  double *cw = (double *)malloc(sizeof(double) * (size_t)(N + 2));
  double norm = 0.0;
  for (int user = 0; user < N; ++user) {
    if (user == tx_user) continue;
    double d; int sign;
    if (dist_piggy(o, e, user, tx_user, &d, &sign)) {
      vals[n++] = d * sign;
      if (fabs(d) > norm) norm = fabs(d);
    }
  }
  if (n > 0) {
    qsort(vals, (size_t)n, sizeof(double), cmp_double);
    for (int j = 0; j < n; ++j) vals[j] = vals[j] / norm;
    cw[0] = 0.0;
    for (int j = 0; j < n; ++j) cw[j + 1] = cw[j] + vals[j];
    double prev = 0.0;
    for (int b = 0; b <= K; ++b) {
      int idx = 0;
      if (b < K) { while (idx < n && vals[idx] < o->edges1[b]) ++idx; }   /* left  */
      else       { while (idx < n && vals[idx] <= o->edges1[b]) ++idx; }  /* right */
      double c = cw[idx];
      if (b > 0) out[b - 1] = c - prev;
      prev = c;
    }
  } else {
    for (int j = 0; j < K; ++j) out[j] = 0.0;
  }
  free(vals); free(cw);
}

/* Network.get_positional_dist + dist_sign (network.py:409-430, 334-349) */
static void posdist_full(const Oracle *o, const OEnv *e, int tx_user, double *out) {
  int N = o->N, n = 0;
  double max_dist = 0.0;
  for (int user = 0; user < N; ++user) {
    if (user == tx_user) continue;
    double d = dist_uu(o, e, user, tx_user);
    if (d > max_dist) max_dist = d;
    int sign = (e->px[user] - e->px[tx_user] > 0.0) ? 1 : -1;
    out[n++] = d * sign;
  }
  qsort(out, (size_t)n, sizeof(double), cmp_double);
  for (int j = 0; j < n; ++j) out[j] = out[j] / max_dist;
}

/* TestEnv.obtain_state (test_env.py:527-583), one env */
static void obtain_state_env(const Oracle *o, const OEnv *e, const int32_t *act,
                             const double *chobs, const double *rews,
                             double episode, double eps, double *state) {
  const int N = o->N, A = o->A, S = o->S;
  for (int u = 0; u < N; ++u) {
    double *s = state + (size_t)u * S;
    int p = 0;
    if (has(o, DIRAL_F_ADD_ACTION)) {
      if (has(o, DIRAL_F_ACTION_REAL)) s[p++] = (double)act[u];
      else { for (int i = 0; i < A; ++i) s[p++] = (act[u] == i) ? 1.0 : 0.0; }
    }
    if (has(o, DIRAL_F_ADD_CHANNEL_OBS)) {            /* `obs[user_i]`: A values, A * A with piggybacking */
      const int CW = has(o, DIRAL_F_PIGGYBACKING) ? A * A : A;
      for (int i = 0; i < CW; ++i) s[p++] = chobs[(size_t)u * CW + i];
    }
    if (has(o, DIRAL_F_ADD_POSDIST)) { posdist_full(o, e, u, s + p); p += N - 1; }
    if (has(o, DIRAL_F_ADD_POSDIST_PIGGY)) {
      if (o->cfg.posdist_type == 1) posdist_piggy1(o, e, u, s + p);
      else posdist_piggy2(o, e, u, s + p);
      p += o->K;
    }
    if (has(o, DIRAL_F_ADD_REWARD)) s[p++] = rews[u];
    if (has(o, DIRAL_F_ADD_INDEX)) s[p++] = (double)(u + 1);
    if (has(o, DIRAL_F_ADD_POSITION)) {               /* network.py:403-407 */
      s[p++] = e->px[u] / o->cfg.highway_length;
      s[p++] = e->py[u] / o->cfg.highway_height;
    }
    if (has(o, DIRAL_F_ADD_VELOCITY)) s[p++] = e->vel[u];
    if (has(o, DIRAL_F_FINGERPRINT)) { s[p++] = episode; s[p++] = eps; }
  }
}

/* ------------------------------------------------------------------ API ---- */

/* state-space sizing, test_env.py:49-85 */
int oracle_state_space(const DiralCfg *c) {
  int S = 0;
  if (c->flags & DIRAL_F_ADD_ACTION) S += (c->flags & DIRAL_F_ACTION_REAL) ? 1 : c->num_channels;
  if (c->flags & DIRAL_F_ADD_CHANNEL_OBS) S += c->num_channels;
  if (c->flags & DIRAL_F_ADD_REWARD) S += 1;
  if (c->flags & DIRAL_F_ADD_INDEX) S += 1;
  if (c->flags & DIRAL_F_ADD_VELOCITY) S += 1;
  if (c->flags & DIRAL_F_ADD_POSITION) S += 2;
  if (c->flags & DIRAL_F_ADD_POSDIST) S += c->num_users - 1;
  if (c->flags & DIRAL_F_PIGGYBACKING) S += c->num_channels * (c->num_channels - 1);   /* test_env.py:71-72 */
  if (c->flags & DIRAL_F_FINGERPRINT) S += 2;
  if (c->flags & DIRAL_F_ADD_POSDIST_PIGGY) S += c->num_bins;
  return S;
}

void *oracle_create(const DiralCfg *cfg, int B, int sq_mode, int threads) {
  Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
  o->cfg = *cfg;
  o->B = B; o->N = cfg->num_users; o->A = cfg->num_channels; o->K = cfg->num_bins;
  o->S = oracle_state_space(cfg);
  o->sq_mode = sq_mode;
  o->threads = threads > 0 ? threads : 1;
  int K = o->K > 0 ? o->K : 1;
  o->edges = (double *)calloc((size_t)K + 1, sizeof(double));
  o->edges1 = (double *)calloc((size_t)K + 1, sizeof(double));
  np_linspace(-cfg->bin_range, cfg->bin_range, K + 1, o->edges);
  np_linspace(-1.0, 1.0, K + 1, o->edges1);
  o->envs = (OEnv *)calloc((size_t)B, sizeof(OEnv));
  size_t N = (size_t)o->N;
  for (int b = 0; b < B; ++b) {
    OEnv *e = &o->envs[b];
    e->px = (double *)calloc(N, sizeof(double));
    e->py = (double *)calloc(N, sizeof(double));
    e->vel = (double *)calloc(N, sizeof(double));
    e->seq = (int32_t *)calloc(N * N, sizeof(int32_t));
    e->age = (int32_t *)calloc(N * N, sizeof(int32_t));
    e->tx = (double *)calloc(N * N, sizeof(double));
    e->ty = (double *)calloc(N * N, sizeof(double));
    e->la = (int64_t *)malloc(N * N * sizeof(int64_t));
    e->pf = (int32_t *)calloc(N, sizeof(int32_t));
    e->prev_obs = (double *)calloc(N * (size_t)o->A, sizeof(double));   /* test_env.py:76-79 */
    for (size_t j = 0; j < N * N; ++j) e->la[j] = -1;
  }
  return o;
}

void oracle_destroy(void *h) {
  Oracle *o = (Oracle *)h;
  if (!o) return;
  for (int b = 0; b < o->B; ++b) {
    OEnv *e = &o->envs[b];
    free(e->px); free(e->py); free(e->vel); free(e->seq); free(e->age);
    free(e->tx); free(e->ty); free(e->la); free(e->pf); free(e->prev_obs);
  }
  free(o->envs); free(o->edges); free(o->edges1); free(o->trace); free(o);
}

/* fresh Vehicles: vehicle.py:24-33 (tables zero), network.py:39-42 (la = -1) */
void oracle_reset(void *h, const double *x0, const double *y0, const double *v0) {
  Oracle *o = (Oracle *)h;
  size_t N = (size_t)o->N;
  for (int b = 0; b < o->B; ++b) {
    OEnv *e = &o->envs[b];
    memcpy(e->px, x0 + b * N, N * sizeof(double));
    memcpy(e->py, y0 + b * N, N * sizeof(double));
    memcpy(e->vel, v0 + b * N, N * sizeof(double));
    memset(e->seq, 0, N * N * sizeof(int32_t));
    memset(e->age, 0, N * N * sizeof(int32_t));
    memset(e->tx, 0, N * N * sizeof(double));
    memset(e->ty, 0, N * N * sizeof(double));
    memset(e->pf, 0, N * sizeof(int32_t));
    memset(e->prev_obs, 0, N * (size_t)o->A * sizeof(double));
    memset(e->metrics, 0, sizeof(e->metrics));
    for (size_t j = 0; j < N * N; ++j) e->la[j] = -1;
  }
}

/* chobs: [B][N][A], or [B][N][A * A] for my_step on a State.piggybacking config.  Returns 0, or 1 when some env hit the
 * reference's KeyError (test_env.py:243: the state of that env is then whatever the exception left behind). */
int oracle_step(void *h, int mode, const int32_t *actions, int64_t t,
                double *rews, double *chobs) {
  Oracle *o = (Oracle *)h;
  size_t N = (size_t)o->N, A = (size_t)o->A;
  const size_t CW = (has(o, DIRAL_F_PIGGYBACKING) && mode == DIRAL_STEP_MY_STEP) ? A * A : A;
  int err = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(o->threads) reduction(|:err)
#endif
  for (int b = 0; b < o->B; ++b)
    err |= step_env(o, &o->envs[b], mode, actions + b * N, t, rews + b * N, chobs + b * N * CW);
  if (err) o->piggy_keyerror = 1;
  return err;
}

void oracle_prev_obs(void *h, double *out) {
  Oracle *o = (Oracle *)h;
  size_t n = (size_t)o->N * o->A;
  for (int b = 0; b < o->B; ++b) memcpy(out + b * n, o->envs[b].prev_obs, n * sizeof(double));
}

void oracle_obtain_state(void *h, const int32_t *actions, const double *chobs,
                         const double *rews, double episode, double eps,
                         double *state) {
  Oracle *o = (Oracle *)h;
  size_t N = (size_t)o->N, A = (size_t)o->A, S = (size_t)o->S;
  if (has(o, DIRAL_F_PIGGYBACKING)) A = A * A;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(o->threads)
#endif
  for (int b = 0; b < o->B; ++b)
    obtain_state_env(o, &o->envs[b], actions + b * N, chobs + b * N * A,
                     rews + b * N, episode, eps, state + b * N * S);
}

/* Network.update_velocity (network.py:208-223); draws = random.randrange(1,4) */
void oracle_update_velocity(void *h, const uint8_t *draws) {
  Oracle *o = (Oracle *)h;
  if (!has(o, DIRAL_F_MOBILITY_VARY)) return;          /* test_env.py:503 */
  size_t N = (size_t)o->N;
  for (int b = 0; b < o->B; ++b) {
    OEnv *e = &o->envs[b];
    for (size_t u = 0; u < N; ++u) {
      int r = draws[b * N + u];
      if (r == 1) { e->vel[u] += 0.55; if (e->vel[u] > 2.77) e->vel[u] = 2.77; }
      else if (r == 2) { e->vel[u] -= 0.55; if (e->vel[u] < 1.1) e->vel[u] = 1.1; }
    }
  }
}

/* Network.get_information_age (network.py:560-574); Python negative indexing
 * of the 100-slot list is reproduced */
void oracle_info_age(void *h, int64_t t, int32_t *out) {
  Oracle *o = (Oracle *)h;
  int N = o->N;
  for (int b = 0; b < o->B; ++b) {
    const OEnv *e = &o->envs[b];
    int32_t *ia = out + (size_t)b * 100;
    memset(ia, 0, 100 * sizeof(int32_t));
    for (int tx = 0; tx < N; ++tx)
      for (int rx = 0; rx < N; ++rx) {
        if (tx == rx) continue;
        int64_t la = e->la[(size_t)tx * N + rx];
        if (la != -1) {
          int64_t v = t - la;
          if (v < 100) { if (v < 0) v += 100; if (v >= 0) ia[v] += 1; }
        }
      }
  }
}

void oracle_export(void *h, double *px, double *py, double *vel, int32_t *seq,
                   int32_t *age, double *tx, double *ty, int64_t *la, int32_t *pf) {
  Oracle *o = (Oracle *)h;
  size_t N = (size_t)o->N;
  for (int b = 0; b < o->B; ++b) {
    const OEnv *e = &o->envs[b];
    if (px) memcpy(px + b * N, e->px, N * sizeof(double));
    if (py) memcpy(py + b * N, e->py, N * sizeof(double));
    if (vel) memcpy(vel + b * N, e->vel, N * sizeof(double));
    if (seq) memcpy(seq + b * N * N, e->seq, N * N * sizeof(int32_t));
    if (age) memcpy(age + b * N * N, e->age, N * N * sizeof(int32_t));
    if (tx) memcpy(tx + b * N * N, e->tx, N * N * sizeof(double));
    if (ty) memcpy(ty + b * N * N, e->ty, N * N * sizeof(double));
    if (la) memcpy(la + b * N * N, e->la, N * N * sizeof(int64_t));
    if (pf) memcpy(pf + b * N, e->pf, N * sizeof(int32_t));
  }
}

void oracle_import(void *h, const double *px, const double *py, const double *vel,
                   const int32_t *seq, const int32_t *age, const double *tx,
                   const double *ty, const int64_t *la) {
  Oracle *o = (Oracle *)h;
  size_t N = (size_t)o->N;
  for (int b = 0; b < o->B; ++b) {
    OEnv *e = &o->envs[b];
    if (px) memcpy(e->px, px + b * N, N * sizeof(double));
    if (py) memcpy(e->py, py + b * N, N * sizeof(double));
    if (vel) memcpy(e->vel, vel + b * N, N * sizeof(double));
    if (seq) memcpy(e->seq, seq + b * N * N, N * N * sizeof(int32_t));
    if (age) memcpy(e->age, age + b * N * N, N * N * sizeof(int32_t));
    if (tx) memcpy(e->tx, tx + b * N * N, N * N * sizeof(double));
    if (ty) memcpy(e->ty, ty + b * N * N, N * N * sizeof(double));
    if (la) memcpy(e->la, la + b * N * N, N * N * sizeof(int64_t));
  }
}

/* Network.load_x_positions (network.py:171-178): one [T][N] trace shared by all envs */
void oracle_set_trace(void *h, const double *trace, int T) {
  Oracle *o = (Oracle *)h;
  free(o->trace);
  o->trace = NULL; o->trace_len = 0;
  if (trace && T > 0) {
    o->trace = (double *)malloc(sizeof(double) * (size_t)T * o->N);
    memcpy(o->trace, trace, sizeof(double) * (size_t)T * o->N);
    o->trace_len = T;
  }
}

void oracle_metrics(void *h, double *out) {
  Oracle *o = (Oracle *)h;
  for (int b = 0; b < o->B; ++b)
    memcpy(out + (size_t)b * DIRAL_M_COLUMNS, o->envs[b].metrics, sizeof(o->envs[b].metrics));
}

void oracle_edges(void *h, double *out) {
  Oracle *o = (Oracle *)h;
  memcpy(out, o->edges, ((size_t)o->K + 1) * sizeof(double));
}

int oracle_has_openmp(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}
