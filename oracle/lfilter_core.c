/* TEST / BENCH INFRASTRUCTURE, NOT PRODUCT -- plain-C restatement of the reference's native IIR core loop
 * (/root/reference/src/libtorchaudio/lfilter.cpp:17-48, `host_lfilter_core_loop<float>`):
 *
 *   for every (batch, channel) sequence i, for every sample n:
 *       o = x[i][n] - sum_{c < n_order} y_padded[i][n + c] * a_flipped[channel][c]
 *       y_padded[i][n + n_order - 1] = o
 *
 * with y_padded = n_order - 1 leading zeros + the outputs (filtering.py:995-1003) and a_flipped = the normalised feedback
 * coefficients in reverse order.  The reference parallelises over sequences with at::parallel_for; this restatement takes a
 * range of sequences [seq_lo, seq_hi), so that oracle/cpu_baselines.py can deal sequences to host threads the same way.
 * Pinned against the reference's own compiled loop (oracle/_ref, built from the reference sources by oracle/build_ref.py) by
 * tests/test_oracle_golden.py::test_c_lfilter_core_equals_the_reference_binary -- bit for bit (the same operations in the same order).
 * Built by oracle/Makefile into oracle/_build/liboracle_lfilter.so. */
#include <stdint.h>

void oracle_lfilter_core_f32(const float* x, const float* a_flipped, float* y_padded, int64_t n_channel, int64_t n_samples,
                             int64_t n_order, int64_t seq_lo, int64_t seq_hi) {
  const int64_t n_out = n_samples + n_order - 1;
  for (int64_t i = seq_lo; i < seq_hi; ++i) {
    const float* xi = x + i * n_samples;
    float* yi = y_padded + i * n_out;
    const float* a = a_flipped + (i % n_channel) * n_order;
    for (int64_t n = 0; n < n_samples; ++n) {
      float o = xi[n];
      for (int64_t c = 0; c < n_order; ++c) o -= yi[n + c] * a[c];
      yi[n + n_order - 1] = o;
    }
  }
}
