/* The C-ABI boundary used from plain C (no Python, no torch): the reference's first Hungarian
 * known-answer test (hungarian_tf_tests.py:9-22) through ra_hungarian_f32, plus the error path.
 * Built and run by tests/test_c_binding.py:  gcc hungarian_kat.c -I include -L <pkg> -lrecattend */
#include <stdio.h>
#include <string.h>

#include "recattend.h"

int main(void) {
  const float w[9] = {3, 2, 2, 1, 2, 0, 2, 2, 1};
  float m[9], cx[3], cy[3];
  int rc = ra_hungarian_f32(w, 1, 3, 3, m, cx, cy);
  if (rc != 0) {
    fprintf(stderr, "rc %d: %s\n", rc, ra_last_error_string());
    return 1;
  }
  const float want_m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, want_cx[3] = {2, 1, 1}, want_cy[3] = {1, 1, 0};
  if (memcmp(m, want_m, sizeof m) || memcmp(cx, want_cx, sizeof cx) || memcmp(cy, want_cy, sizeof cy)) {
    fprintf(stderr, "wrong answer\n");
    return 2;
  }
  /* invalid arguments come back as a negative code and a message, never an abort */
  rc = ra_hungarian_f32(NULL, 1, 3, 3, m, cx, cy);
  if (rc >= 0 || strlen(ra_last_error_string()) == 0) return 3;
  /* host-side helpers of the conv path need no device either */
  if (ra_conv_cout_padded(8) != 16 || ra_conv_packed_floats(4, 8) != 9u * 4u * 16u) return 4;
  printf("ok version %d\n", ra_version());
  return 0;
}
