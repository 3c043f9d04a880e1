// ref_torque_wrap.cpp -- a C entry point in front of the REFERENCE's own TorqueAdjustment class, so that the tests can call it
// through ctypes.  This file is ours; the class behind it is compiled from the reference's source where it lies
// (/root/reference/estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp, unmodified) by `make -C oracle ref`
// into oracle/_ref/libref_torque_adjustment.so.  Test infrastructure: nothing under pronto_amd/ loads it.
//
// It is the one piece of this path that the image can build from the reference: every other file needs Eigen3 / boost / LCM /
// libbot2 / KDL.  What it pins: po_torque_adjust (oracle/leg_odometry.c), the numpy witness, the device
// function torque_adjust (rbis_legodo.hpp, via the host harness) and the ta_in / ta_out vectors of tests/golden/leg_fk.npz.
#include <estimate_tools/torque_adjustment.hpp>

#include <string>
#include <vector>

extern "C" {
// TorqueAdjustment(adjust_names, gains).processSample(names, positions, efforts): positions [n] is adjusted in place.
// (The constructor prints its gains on stdout, the reference's behaviour.)
int ref_torque_adjustment(int n_adjust, const char *const *adjust_names, const float *gains, int n, const char *const *names,
                          float *positions, const float *efforts)
{
  std::vector<std::string> adj, nm;
  std::vector<float> g(gains, gains + n_adjust), pos(positions, positions + n), eff(efforts, efforts + n);
  for (int i = 0; i < n_adjust; i++) adj.push_back(adjust_names[i]);
  for (int i = 0; i < n; i++) nm.push_back(names[i]);
  EstimateTools::TorqueAdjustment ta(adj, g);
  ta.processSample(nm, pos, eff);
  std::cout.flush();   // (the constructor's line: out now, so that a caller that redirected stdout around this call is rid of it)
  for (int i = 0; i < n; i++) positions[i] = pos[(size_t) i];
  return 0;
}
}
