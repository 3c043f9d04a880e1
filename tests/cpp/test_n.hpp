// test_n.hpp -- the state count of a handler test: every tests/cpp program runs for the 15-state filter and for the 21-state one
// (gyro / accelerometer biases estimated online, BASELINE config 5) -- "n21" anywhere on the command line selects the latter.
#pragma once
#include <cstring>
#include <string>

#include "../../oracle/pronto_oracle.h"
#include "../../pronto_amd/csrc/mav_state_est_batch.hpp"

// removes "n15" / "n21" from argv and returns 15 / 21 (15 when absent)
static inline int take_n_states(int &argc, char **argv)
{
  int n = 15, w = 1;
  for (int i = 1; i < argc; i++) {
    if (strcmp(argv[i], "n21") == 0) n = 21;
    else if (strcmp(argv[i], "n15") == 0) n = 15;
    else argv[w++] = argv[i];
  }
  argc = w;
  return n;
}

// the INS keys that differ between the two filters: 21 states = bias random walks on, both biases updated online
static inline void set_ins_bias_keys(MavStateEst::BotParam &param, int n)
{
  param.set("state_estimator.ins.q_gyro_bias", n == 21 ? 0.001 : 0.0);
  param.set("state_estimator.ins.q_accel_bias", n == 21 ? 0.0001 : 0.0);
  param.set("state_estimator.ins.accel_bias_update_online", n == 21 ? "true" : "false");
  param.set("state_estimator.ins.gyro_bias_update_online", n == 21 ? "true" : "false");
}

// initial bias standard deviations / values of the 21-state runs (rbis_initializer.cpp:85-91's sigma_gyro_bias / sigma_accel_bias)
static const double TEST_SIG_BIAS[6] = { .008, .008, .008, .1, .1, .1 };

// gives filter b of a 21-state run its bias prior and a non-zero initial bias on both sides (shim containers and oracle)
static inline void init_bias_states(int n, int b, MavStateEst::RBIS &x0, MavStateEst::RBIM &P0, po_rbis *ox, po_rbim *oP, double (*unit_rand)())
{
  if (n != 21) return;
  for (int i = 0; i < 6; i++) {
    const double v = (i < 3 ? 0.004 : 0.05) * (unit_rand() - 0.5);
    x0(15 + i, b) = v;
    ox->vec[15 + i] = v;
    P0(15 + i, 15 + i, b) = TEST_SIG_BIAS[i] * TEST_SIG_BIAS[i];
    oP->m[(15 + i) * 21 + 15 + i] = TEST_SIG_BIAS[i] * TEST_SIG_BIAS[i];
  }
}
