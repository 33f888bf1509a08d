/* TEST INFRASTRUCTURE — CPU oracle of the reference's rotation-averaging path.
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liborc.so.  The entry points mirror include/gsfm_rot.h
 * (same argument meaning) so a parity test calls both sides the same way. */
#ifndef GSFM_ORACLE_API_H_
#define GSFM_ORACLE_API_H_
#include "../include/gsfm_rot.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TRACE_COLS 8 /* iteration, cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius, cg_iters */

typedef struct orc_problem orc_problem;

void orc_options_default(gsfm_rot_options* o);
int orc_set_num_threads(int n); /* OpenMP threads for the parallel loops; returns the previous maximum */
int32_t orc_residual_dim(int32_t error_type);
orc_problem* orc_problem_create(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j,
                                const double* rel_aa, int32_t error_type, const double* cov6,
                                const double* inlier_weight);
void orc_problem_destroy(orc_problem* p);
int orc_set_loss(orc_problem* p, const gsfm_loss_node* prog, int32_t n_nodes);
int orc_set_loss_callback(orc_problem* p, gsfm_loss_callback fn, void* user);
int orc_set_linear_solver(orc_problem* p, int32_t kind); /* 0 auto, 1 dense Cholesky, 2 PCG(1e-14) */
int orc_set_edge_weights(orc_problem* p, const double* w);
int orc_residuals(orc_problem* p, const double* rot_aa, double* s_out, double* rho_out, double* residual_out,
                  double* cost);
int orc_linearize(orc_problem* p, const double* rot_aa, double* gradient, double* diag_blocks, double* cost);
int orc_normal_matvec(orc_problem* p, const double* v, double* y);
int orc_solve(orc_problem* p, double* rot_aa_inout, const gsfm_rot_options* opt, gsfm_rot_summary* summary);
int orc_solve_sigma_consensus(orc_problem* p, double* rot_aa_inout, int32_t iters_num, double sigma_max,
                              const gsfm_rot_options* opt, gsfm_rot_summary* summary);
int32_t orc_get_trace(orc_problem* p, double* out, int32_t cap_rows);
/* test hook: the linear systems (J^T J + diag(D)^2) y = rhs of the first LM iterations of the next solve, see ref_solver.cpp */
int orc_capture_steps(orc_problem* p, int32_t max_steps);
int orc_captured_step(orc_problem* p, int32_t k, double* Ji, double* Jj, double* rt, double* D, double* rhs, double* y, double* scale);

void orc_loss_eval(const gsfm_loss_node* prog, int32_t n, double s, double* out3);
int32_t orc_magsac_table(int32_t nu, double* out, int32_t cap);
void orc_magsac_constants(int32_t nu, double* C, double* sigma_quantile, double* gamma_k);
void orc_whitening(int32_t error_type, const double* cov6, double inlier_w, double* W9);

void orc_angle_axis_to_rotation_matrix(const double* aa, double* R9_row_major);
void orc_rotation_matrix_to_angle_axis(const double* R9_row_major, double* aa);
void orc_angle_axis_to_quaternion(const double* aa, double* q_wxyz);
void orc_quaternion_to_angle_axis(const double* q_wxyz, double* aa);
void orc_pairwise_rotation_error(const double* aa1, const double* aa2, const double* rel_aa, double weight,
                                 double* out3);
int orc_edge_jacobians(orc_problem* p, uint64_t e, const double* rot_aa, double* r, double* Ji, double* Jj);

/* ---- per-edge rotation covariance (reference src/uncertainty.cpp:36-198) ---- */
int orc_cov_estimate(uint64_t n_edges, const uint64_t* match_ptr, const double* matches_x1y1x2y2, const double* intrinsics6,
                     const double* rot_in, const double* trans_in, int32_t max_iterations, double* cov9_out,
                     double* rot_out, double* trans_out, int32_t* status_out, int32_t* iters_out);
double orc_sampson_residual(const double* match4, const double* intr6, const double* rot, const double* t, double* jac6);
void orc_homogeneous_plus(const double* x3, const double* d2, double* out3);
void orc_homogeneous_jacobian(const double* x3, double* J32);

#ifdef __cplusplus
}
#endif
#endif
