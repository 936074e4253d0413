/*
 * san_check.c - drives oracle/diral_oracle.c under AddressSanitizer + UBSan
 * ("compile the CPU restatement with -fsanitize").  TEST
 * INFRASTRUCTURE ONLY.  Exercises every step kind, reward design and observation
 * mode on ragged sizes with pseudo-random actions; any out-of-bounds access or
 * undefined behaviour aborts with a non-zero exit code.
 *   make -C oracle san_check && oracle/san_check
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/diral_env.h"

void *oracle_create(const DiralCfg *cfg, int B, int sq_mode, int threads);
void oracle_destroy(void *h);
void oracle_reset(void *h, const double *x0, const double *y0, const double *v0);
int oracle_step(void *h, int mode, const int32_t *actions, int64_t t, double *rews, double *chobs);
void oracle_obtain_state(void *h, const int32_t *actions, const double *chobs, const double *rews,
                         double episode, double eps, double *state);
void oracle_update_velocity(void *h, const uint8_t *draws);
void oracle_info_age(void *h, int64_t t, int32_t *out);
void oracle_set_trace(void *h, const double *trace, int T);
int oracle_state_space(const DiralCfg *c);

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); }

static int run(int N, int A, int K, int rd, int mode, uint32_t extra_flags, int posdist_type, int B, int T) {
  DiralCfg c;
  memset(&c, 0, sizeof(c));
  c.struct_bytes = sizeof(c);
  c.flags = DIRAL_F_MOBILITY | DIRAL_F_ADD_ACTION | DIRAL_F_ADD_POSDIST_PIGGY | DIRAL_F_MOBILITY_VARY | extra_flags;
  c.num_users = N; c.num_channels = A; c.num_bins = K; c.reward_design = rd; c.state_type = 2;
  c.posdist_type = posdist_type; c.episode_interval = 25; c.info_age_limit = 20; c.pf_threshold = 10;
  c.pf_penalty = -10; c.highway_length = 30.0 * N + 60; c.highway_height = 2; c.communication_range = 120;
  c.bin_range = 500;
  /* State.piggybacking: A * A values per agent; every receiver in range (run A) or the KeyError path (run B: rc 1, the env abandoned) */
  const int pb = (extra_flags & DIRAL_F_PIGGYBACKING) != 0;
  if (pb && N <= 16) c.communication_range = c.highway_length + 1;
  const int CW = pb ? A * A : A;
  const int S = oracle_state_space(&c);
  void *h = oracle_create(&c, B, 0, 1);
  double *x0 = malloc(sizeof(double) * B * N), *y0 = calloc((size_t)B * N, sizeof(double)), *v0 = malloc(sizeof(double) * B * N);
  for (int i = 0; i < B * N; ++i) { x0[i] = rnd() % (int)c.highway_length; v0[i] = 1.1 + (rnd() % 1600) / 1000.0; }
  oracle_reset(h, x0, y0, v0);
  int32_t *act = malloc(sizeof(int32_t) * B * N);
  double *rews = malloc(sizeof(double) * B * N), *chobs = malloc(sizeof(double) * B * N * CW), *state = malloc(sizeof(double) * B * N * (S + 1));
  uint8_t *draws = malloc((size_t)B * N);
  int32_t *ia = malloc(sizeof(int32_t) * B * 100);
  double *trace = malloc(sizeof(double) * 5 * N);
  for (int i = 0; i < 5 * N; ++i) trace[i] = rnd() % (int)c.highway_length;
  double acc = 0;
  for (int t = 0; t < T; ++t) {
    for (int i = 0; i < B * N; ++i) act[i] = (int32_t)(rnd() % (uint32_t)A);
    if (oracle_step(h, mode, act, t, rews, chobs)) break;   /* the reference's KeyError (piggybacking, nobody in range) */
    oracle_obtain_state(h, act, chobs, rews, t / 25, 0.5, state);
    oracle_info_age(h, t, ia);
    if (t % 25 == 24) { for (int i = 0; i < B * N; ++i) draws[i] = 1 + rnd() % 3; oracle_update_velocity(h, draws); }
    if (t == T / 2) oracle_set_trace(h, trace, 5);
    if (t == T / 2 + 7) oracle_set_trace(h, NULL, 0);
    for (int i = 0; i < B * N * S; ++i) acc += state[i];
  }
  oracle_destroy(h);
  free(x0); free(y0); free(v0); free(act); free(rews); free(chobs); free(state); free(draws); free(ia); free(trace);
  return acc == acc ? 0 : 1;   /* NaN would mean a divide by zero slipped through */
}

int main(void) {
  int bad = 0;
  const uint32_t all = DIRAL_F_ADD_CHANNEL_OBS | DIRAL_F_ADD_REWARD | DIRAL_F_ADD_INDEX | DIRAL_F_ADD_VELOCITY |
                       DIRAL_F_ADD_POSITION | DIRAL_F_FINGERPRINT | DIRAL_F_PROPORTIONAL_FAIR | DIRAL_F_TRACK_PRR;
  for (int rd = 1; rd <= 5; ++rd) bad |= run(64, 32, 20, rd, DIRAL_STEP_MY_STEP, 0, 2, 3, 40);
  for (int rd = 2; rd <= 4; ++rd) bad |= run(33, 7, 10, rd, DIRAL_STEP_MY_STEP_CH, all, 2, 2, 40);
  bad |= run(70, 5, 40, 2, DIRAL_STEP_DESIGN, all, 2, 2, 30);
  bad |= run(20, 40, 7, 1, DIRAL_STEP_MY_STEP, all | DIRAL_F_ADD_POSDIST, 1, 2, 60);
  bad |= run(1, 1, 3, 2, DIRAL_STEP_MY_STEP, DIRAL_F_ADD_POSDIST, 1, 2, 10);
  bad |= run(2, 1, 3, 2, DIRAL_STEP_MY_STEP_CH, all, 2, 2, 10);
  bad |= run(130, 64, 20, 2, DIRAL_STEP_MY_STEP, DIRAL_F_TOY_WEIGHTS, 2, 1, 12);
  bad |= run(12, 5, 10, 2, DIRAL_STEP_MY_STEP, all | DIRAL_F_PIGGYBACKING, 2, 3, 50);
  bad |= run(40, 9, 10, 3, DIRAL_STEP_MY_STEP, DIRAL_F_ADD_CHANNEL_OBS | DIRAL_F_PIGGYBACKING | DIRAL_F_ADD_POSDIST, 1, 2, 30);
  printf("san_check %s\n", bad ? "FAILED" : "ok");
  return bad;
}
