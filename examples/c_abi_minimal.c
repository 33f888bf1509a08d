/* The C-ABI from plain C: build a tiny view graph, solve it on the device, print the summary.
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_abi_minimal.c -Lglobalsfmpy_amd -lgsfm_rot \
 *       -Wl,-rpath,$PWD/globalsfmpy_amd -lm -o c_abi_minimal
 * Four cameras on a ring with one gross outlier edge; SoftLOne(0.1) as in EstimateRotations (estimator.cpp:44-45). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "gsfm_rot.h"

int main(void) {
  /* ground truth: camera k rotated by 0.1 k rad about z; measurement of (i, j) = R_j R_i^T = rotation by 0.1 (j - i) about z */
  const uint32_t edge_i[6] = {0, 1, 2, 0, 0, 1};
  const uint32_t edge_j[6] = {1, 2, 3, 3, 2, 3};
  double rel_aa[18];
  double rot[12];
  gsfm_rot_problem* p = NULL;
  gsfm_rot_options opt;
  gsfm_rot_summary sum;
  gsfm_loss_node loss;
  int e, k;
  for (e = 0; e < 6; ++e) { rel_aa[3 * e] = 0.0; rel_aa[3 * e + 1] = 0.0; rel_aa[3 * e + 2] = 0.1 * (double)(edge_j[e] - edge_i[e]); }
  rel_aa[3 * 5 + 0] = 1.0; /* edge (1, 3): an outlier */
  memset(rot, 0, sizeof(rot));
  if (gsfm_rot_problem_create(4, 6, edge_i, edge_j, rel_aa, GSFM_ROT_ANGLE_AXIS, NULL, NULL, NULL, &p) != GSFM_OK) {
    fprintf(stderr, "create failed: %s\n", gsfm_last_error());
    return 2;
  }
  memset(&loss, 0, sizeof(loss));
  loss.kind = GSFM_LOSS_SOFT_L1;
  loss.p[0] = 0.1;
  if (gsfm_rot_set_loss(p, &loss, 1) != GSFM_OK) { fprintf(stderr, "set_loss failed: %s\n", gsfm_last_error()); return 2; }
  gsfm_rot_options_default(&opt);
  if (gsfm_rot_solve(p, rot, &opt, &sum) != GSFM_OK) { fprintf(stderr, "solve failed: %s\n", gsfm_last_error()); return 2; }
  printf("termination %d after %d iterations, cost %.6e -> %.6e\n", (int)sum.termination, (int)sum.num_iterations, sum.initial_cost, sum.final_cost);
  for (k = 0; k < 4; ++k) printf("camera %d: relative to camera 0: %+.4f %+.4f %+.4f\n", k, rot[3 * k] - rot[0], rot[3 * k + 1] - rot[1], rot[3 * k + 2] - rot[2]);
  gsfm_rot_problem_destroy(p);
  /* the inlier ring fixes the z-angles at 0.1 k up to the common gauge */
  return fabs((rot[3 * 3 + 2] - rot[2]) - 0.3) < 0.02 ? 0 : 1;
}
