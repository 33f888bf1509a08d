/*
 * gsfm_rot.h — C ABI of the MI355X-native robust rotation-averaging solver.
 *
 * This is the drop-in boundary for ONE path of zhangganlin/GlobalSfMpy: the
 * robust rotation-averaging solve behind
 *   theia::GSfMNonlinearRotationEstimator::EstimateRotations*()
 * (reference: include/GSfM_nonlinear_rotation_estimator.hpp:22-59,
 *  src/GSfM_nonlinear_rotation_estimator.cpp:24-457).  In the reference that
 * path is a ceres::Problem (one AutoDiff residual block per view-graph edge)
 * solved by Levenberg-Marquardt + SPARSE_NORMAL_CHOLESKY.  Here the whole
 * inner loop (residuals, 3x3 Jacobians, covariance whitening, robust-loss
 * correction, normal-equation assembly, block-Jacobi PCG, LM control) runs as
 * hand-written fp64 HIP kernels for gfx950 behind these entry points.
 *
 * Conventions (all identical to the reference):
 *   - a camera orientation is a world->camera angle-axis 3-vector;
 *   - an edge (i, j) carries the angle-axis of R_ij ~= R_j * R_i^T
 *     (thirdparty/TheiaSfM/src/theia/sfm/global_pose_estimation/pairwise_rotation_error.h:47-48);
 *   - cost = sum_e 1/2 rho(||r_e||^2) (Ceres convention, scripts/loss_functions.py:16-19).
 *
 * Plain pointers and sizes only: no C++ types, no torch types.  All pointers are
 * HOST pointers unless the name ends in _dev.  The caller owns every buffer; the
 * library copies what it needs into HBM at create time.  Every function returns a
 * gsfm_status and never aborts; gsfm_last_error() gives the message of the last
 * failure on the calling thread.  One problem may be used by one thread at a time.
 */
#ifndef GSFM_ROT_H_
#define GSFM_ROT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSFM_ROT_ABI_VERSION 4

typedef enum {
  GSFM_OK = 0,
  GSFM_ERR_INVALID_ARG = 1,
  GSFM_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime failure at init */
  GSFM_ERR_HIP = 3,         /* a HIP call failed */
  GSFM_ERR_EMPTY = 4,       /* no cameras or no usable edges (reference returns false) */
  GSFM_ERR_COMM = 5,        /* a shard communicator callback failed */
  GSFM_ERR_UNSUPPORTED = 6
} gsfm_status;

/* Same numeric values as `enum class RotationErrorType`
 * (reference include/pairwise_rotation_error_quat.hpp:50-61). */
typedef enum {
  GSFM_ROT_QUATERNION_NORM = 0,        /* 4 residuals, PairwiseRotationErrorQuatFNorm  (quat.hpp:108-150) */
  GSFM_ROT_ROTATION_MAT_FNORM = 1,     /* 9 residuals, PairwiseRotationErrorRotFNorm   (quat.hpp:152-196) */
  GSFM_ROT_QUATERNION_COSINE = 2,      /* 3 residuals, PairwiseRotationErrorQuat       (quat.hpp:65-106)  */
  GSFM_ROT_ANGLE_AXIS_COVARIANCE = 3,  /* r = Lt * log(.)  Lt = chol((1e8*Sigma)^-1)^T  (estimator.cpp:251-256) */
  GSFM_ROT_ANGLE_AXIS = 4,             /* r = log(.)                                    (estimator.cpp:257-259) */
  GSFM_ROT_ANGLE_AXIS_INLIERS = 5,     /* r = (#common/100) * log(.)                    (estimator.cpp:260-264) */
  GSFM_ROT_ANGLE_AXIS_COV_INLIERS = 6, /* r = (#common/100) * Lt * log(.)               (estimator.cpp:265-274) */
  GSFM_ROT_ANGLE_AXIS_COVTRACE = 7,    /* r = sqrt(1/trace(1e8*Sigma)) * log(.)         (estimator.cpp:275-282) */
  GSFM_ROT_ANGLE_AXIS_COVNORM = 8      /* r = sqrt(1/||1e8*Sigma||_F) * log(.)          (estimator.cpp:283-287) */
} gsfm_rot_error_type;

/* ------------------------------------------------------------------------- */
/* Robust losses: a native descriptor of scripts/loss_functions.py            */
/* ------------------------------------------------------------------------- */
/* A loss is a short postfix program run by a tiny stack machine, once per edge,
 * on the device.  Leaves push (rho, rho', rho'')(arg); combinators rewrite the
 * stack.  arg is the top of an argument stack that starts as [s].
 *   ScaledLoss(rho, a)    ->  prog(rho), SCALE(a)
 *   ComposedLoss(f, g)    ->  prog(g), PUSH_ARG, prog(f), COMPOSE
 * A NULL / empty program is the Ceres NULL loss (cost = s/2, no correction).   */
typedef enum {
  GSFM_LOSS_TRIVIAL = 0,       /* loss_functions.py:47   */
  GSFM_LOSS_HUBER = 1,         /* :56   p0 = a           */
  GSFM_LOSS_SOFT_L1 = 2,       /* :74   p0 = a           */
  GSFM_LOSS_CAUCHY = 3,        /* :88   p0 = a           */
  GSFM_LOSS_ARCTAN = 4,        /* :101  p0 = a           */
  GSFM_LOSS_TOLERANT = 5,      /* :114  p0 = a, p1 = b   */
  GSFM_LOSS_TUKEY = 6,         /* :167  p0 = a           */
  GSFM_LOSS_LONE_HALF = 7,     /* :187  p0 = a           */
  GSFM_LOSS_LTWO = 8,          /* :216  p0 = a (sigma2 ignored, as in the reference) */
  GSFM_LOSS_GEMAN_MCCLURE = 9, /* :239  p0 = a, p1 = sigma2 */
  GSFM_LOSS_MAGSAC = 10,       /* :285/:344/:402  p0 = sigma, p1 = nu (3|4|9), p2 = inverse (0|1) */
  GSFM_LOSS_OP_SCALE = 32,     /* :267  p0 = a           */
  GSFM_LOSS_OP_PUSH_ARG = 33,  /* push top-of-stack rho as the new argument */
  GSFM_LOSS_OP_COMPOSE = 34    /* :250  pops f, g -> (f.rho, f'.g', f''.g'^2 + f'.g'') */
} gsfm_loss_kind;

typedef struct {
  int32_t kind;     /* gsfm_loss_kind */
  int32_t reserved; /* must be 0 */
  double p[3];
} gsfm_loss_node;

#define GSFM_LOSS_MAX_NODES 16
#define GSFM_LOSS_MAX_STACK 6

/* Arbitrary user loss (a Python subclass of LossFunction without a native
 * descriptor): evaluated on the HOST exactly like the reference's trampoline
 * (bind_src/GlobalSfMpy.cpp:33-65), once per edge per residual evaluation; the
 * residuals, Jacobians and the solve still run on the device. */
typedef void (*gsfm_loss_callback)(void* user, double sq_norm, double out[3]);

/* ------------------------------------------------------------------------- */
/* Options / summary                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t max_num_iterations;          /* 200   (estimator.cpp:73,178,301,442) */
  int32_t num_threads;                 /* accepted for signature parity; unused on the device path */
  double function_tolerance;           /* 1e-6  Ceres 1.14 default */
  double gradient_tolerance;           /* 1e-10 */
  double parameter_tolerance;          /* 1e-8  */
  double initial_trust_region_radius;  /* 1e4   */
  double max_trust_region_radius;      /* 1e16  */
  double min_trust_region_radius;      /* 1e-32 */
  double min_relative_decrease;        /* 1e-3  */
  double min_lm_diagonal;              /* 1e-6  */
  double max_lm_diagonal;              /* 1e32  */
  int32_t jacobi_scaling;              /* 1     */
  int32_t max_cg_iterations;           /* PCG replaces CHOLMOD: iteration cap per LM step (default 20000, the oracle's; 1000 until round 4).  Beyond its
                                          first 1000 iterations a solve goes on only while the relative residual still halves within 512 iterations.
                                          A solve that ends ABOVE cg_relative_tolerance either way is not the reference's exact step
                                          (estimator.cpp:300): graphs of at most dense_cholesky_auto_cams cameras hand that step, and the rest of the
                                          run, to the factorisation; larger ones evaluate it and say so in gsfm_rot_summary::num_pcg_capped_steps */
  double cg_relative_tolerance;        /* stop when sqrt(r.M^-1 r / b.M^-1 b) <= tol. Default 1e-12, the stand-in for the
                                          reference's exact sparse Cholesky: every parity claim is made at this value.
                                          Looser values are a documented trade: on the C5 graph
                                          1e-8 halves the PCG work and moves the solution by 3e-10 rad (mean), but on the
                                          ill-conditioned real Madrid graph 1e-10 already costs 1e-6 rad mid-trajectory.
                                          Absolute floor (round 5): no solve runs below the point where block-Jacobi's estimate of what ANY camera's
                                          update still lacks is under 2e-14 rad (a relative 1e-12 of a one-degree step leaves more) -- 1e-11 rad for a
                                          disconnected problem under a smooth loss, whose converged scenes must not cost iterations while another iterates.
                                          A DISCONNECTED view graph (several scenes batched as one problem) is solved per connected component when
                                          unsharded: components of at most dense_cholesky_max_cams cameras by exact factorisations side by side, the
                                          others by PCG (csrc/solver_components.hpp); where one PCG still covers several components it is never solved looser than
                                          1e-14: the residual norm is global, and a component that has already converged would otherwise be
                                          left with an error that is large against its own right-hand side (measured: 3e-5 rad on the real
                                          Madrid component of the 14-scene batch at 1e-12, 4e-9 rad at 1e-14, for 9 % more iterations).
                                          Preconditioner: block-Jacobi (the 3x3 diagonal blocks), plus -- chosen automatically for
                                          problems of >= 4096 cameras whose numbering keeps neighbours close (as given, or after the locality
                                          relabelling; judged by cameras / mean index distance of the edges) -- a coarse space of 16-64 aggregates of the camera order in the body frame, where the gauge
                                          rotation is the constant vector: 10-15x fewer iterations on spatially coherent graphs (430 -> 40 ms on
                                          100k cameras / 2M edges), the same answer to this tolerance.  Environment GSFM_PCG_COARSE=n forces n
                                          aggregates, =0 switches it off.  Its coarse matrix is summed with integer atomics on a fixed-point
                                          image: results stay bit-identical from run to run. */
  int32_t cg_check_interval;           /* CG iterations enqueued between host checks (default 8) */
  int32_t verbose;                     /* 1: print one line per LM iteration to stderr */
  int32_t pcg_single_reduction;        /* 0: textbook PCG, 4 dependent kernels per iteration; 1: Chronopoulos-Gear single-reduction PCG, 2 kernels
                                          per iteration (mat-vec with the fused w.u partials, then one vector kernel; same iterates to
                                          rounding; 3 on the column-sorted layout); -1 (default): the latter when the problem has at most
                                          2M directed entries -- there the iteration is bounded by dependent-launch latency, not by
                                          bandwidth (C2: 10k cameras / 200k edges) -- and on every sharded problem that is connected and
                                          solved to a tolerance >= 1e-13: its delta partials travel in the tail of the iteration's one
                                          all-gather (3 kernels + 1 collective per iteration instead of 5 + 1).  Large single-GPU problems
                                          keep the textbook recurrence (measured equal there). */
  int32_t cg_stall_iterations;         /* opt-in: stop PCG when the relative residual has not halved for this many
                                          iterations (default 0 = never).  On the real Madrid graph (MAGSAC weights spanning
                                          1e-5..5e4, vanishing damping) PCG needs up to 574 iterations per step; 64 here saves
                                          12 % of them but already perturbs the early trajectory by 5e-8 in cost. */
  int32_t dense_cholesky_max_cams;     /* Exact LM steps for small graphs: a tiled Cholesky of the dense damped normal matrix on the
                                          device (csrc/dense_kernels.hpp) -- literally the reference's 'normal equations + Cholesky'
                                          (estimator.cpp:72-74), so the step equals the reference's to rounding instead of to the PCG tolerance.
                                          > 0: every step of graphs with at most that many cameras (default 512; capped at 1706 = 5120 / 3
                                          unknowns); 0: never (PCG only); < 0: graphs up to |value| cameras, from the moment one PCG solve of
                                          the run has needed more than 150 iterations.  Measured on MI355X (tools/bench_chol.hip, factor +
                                          both substitutions as one graph replay): 0.59 ms at 394 cameras, 1.9 ms at 800, 7.7 ms at 1500;
                                          a PCG iteration of a graph this size costs ~14 us, and real view graphs need 60-570 of them per LM
                                          step once the damping has vanished.  Madrid (394 cameras, 62 LM iterations, MAGSAC): 43 ms exact vs
                                          325 ms PCG; SoftL1 32 vs 58 ms; quaternion-Huber 19 vs 26 ms.  PCG remains the fallback when a pivot
                                          is not positive, and the only solver of sharded problems. */
  int32_t pcg_hip_graph;               /* default 1: the chunk of cg_check_interval PCG iterations between two host checks
                                          (2-4 dependent kernels each) is captured once into a hipGraph and replayed -- the loop is
                                          launch-latency-bound on small graphs.  Same kernels, same order, same iterates.  Falls back to
                                          plain launches when the problem runs on a stream that cannot be captured (the legacy default stream).
                                          On a sharded problem the chunk contains the collective callbacks: it is captured (collectives included)
                                          when the shard descriptor carries GSFM_SHARD_CAPTURABLE (the native RCCL communicator,
                                          csrc/gsfm_rccl.cpp: RCCL collectives are stream-capturable) -- the default since round 3, 2 is the old
                                          explicit opt-in and means the same, GSFM_PCG_GRAPH_COLLECTIVES=0 in the environment switches it off;
                                          host-staged callbacks (gloo) keep plain launches.  0: plain launches everywhere. */
  int32_t pcg_forcing;                 /* default 1: forcing schedule for the PCG solves -- an LM step is solved only as accurately as the answer
                                          needs.  PCG first stops once the estimated relative energy-norm error of the step is below tau, with tau
                                          chosen so that tau x (rms step size, predicted from the previous accepted step) <= pcg_forcing_tolerance
                                          radians.  (Estimate: Hestenes-Stiefel -- the squared energy error is the sum of the LATER iterations' model
                                          decreases alpha_j r_j.z_j, extrapolated geometrically from the last four; csrc/kernels.hpp cg_energy_stop.
                                          Unlike a residual norm it bounds the missing share of the step whatever the conditioning and the
                                          preconditioner.)  The loose step is evaluated, and every decision taken from it must be the exact step's:
                                          a cost change or step norm more than a factor two away from the function / parameter tolerance decides
                                          termination (those quantities agree with the exact step's to O(tau); a terminating step is never applied,
                                          so the answer is the same), a relative decrease above 0.25 decides acceptance, and an accepted loose step
                                          must be within pcg_forcing_tolerance (tau x its MEASURED rms size) of the exact one.  In every other case
                                          -- too close to a threshold, a doubtful or invalid step, a step larger than predicted -- PCG CONTINUES from
                                          where it stopped, to cg_relative_tolerance or to the tau the measured size asks for; the solver state is
                                          resumable and the iterates are bit for bit those of an uninterrupted solve at that tolerance
                                          (estimator.cpp:300's SPARSE_NORMAL_CHOLESKY is what cg_relative_tolerance stands in for).  The last step
                                          an iteration cap allows is always solved tightly.  Not applied to disconnected graphs (their 1e-14 rule
                                          above stands), to exact Cholesky steps, or to QUATERNION_NORM (a discontinuous residual: its sign
                                          canonicalisation flips under differences the schedule allows).
                                          ROUND 5 -- the schedule is kept only where it cannot move the answer: loose solves are taken only while
                                          every accepted step so far was at most 0.3 x its predecessor (the benchmark configurations: 0.06-0.21;
                                          far starts under redescending or cut-off losses: 0.3-0.95 for dozens of iterations, where nothing injected
                                          early is forgotten and -- MAGSAC: rho and rho' are piecewise constant in s -- one edge moved across a
                                          table cell puts the run on another self-consistent weighting).  The first violation switches the schedule
                                          off for the rest of the run, and a run that has already applied an inexact step is REDONE from the initial
                                          rotations with exact steps (gsfm_rot_summary::num_forcing_restarts; bit-identical to pcg_forcing = 0).  Also
                                          redone: a run whose termination / acceptance decision on a cost-lowering candidate hangs by less than the
                                          noise of the loss (MAGSAC: +- 100 / sqrt(edges), at most a factor two; smooth losses: one per mille).  Every
                                          loose step is moreover checked camera by camera (block-Jacobi's estimate of what each camera's update still
                                          lacks: below 1e-6 rad).  Not applied under a loss that switches edges off (Tukey) or one the library cannot
                                          see into (a host callback).  tests/manual/fuzz_forcing.py, 4 840 default-option trials: none beyond 1e-6 rad (largest 4.7e-7).
                                          Under the MAGSAC losses the schedule is also given up -- the solve in hand continued to the tight tolerance,
                                          exact steps from there on -- the moment a loose solve has needed more than 64 iterations: an ill-conditioned
                                          system, where the energy estimate says little about weakly coupled camera clusters and MAGSAC turns what they
                                          are left short of into another set of inlier edges (fuzz seed 2 trial 45).  The benchmark graph from the
                                          spanning-tree start is such a run (308 loose iterations in its first step): 218 ms with exact steps by
                                          default, 106 ms and 1.5e-9 rad from it with pcg_forcing = 3.
                                          EXCEPTION, documented (round-5 review): what the schedule keeps is the ROTATIONS, not in every case the LM
                                          iteration COUNT ("IRLS iterations to 1e-6", gsfm_rot_summary::num_iterations / iters_to_1e6).  A MAGSAC run
                                          that ends on REJECTED candidates hovering at the function tolerance -- cost changes of 0.9 against 1.1 x
                                          1e-6 of the cost, each a sum of table-cell jumps -- ends one to seven rejections earlier or later than the
                                          exact schedule does, the state not moving in between (<= 5e-9 rad): 7 of 4 840 fuzz trials, LM 7 against 14
                                          once.  A rejected candidate changes nothing whichever iteration the run ends at, so such a run is NOT
                                          redone (only decisions on cost-LOWERING candidates inside the noise band are, above); a caller who needs
                                          the reference's iteration count under MAGSAC sets pcg_forcing = 0.  On every BASELINE configuration the two
                                          schedules give the same count (bench.py prints both, tests/test_gpu_bench.py asserts it on C5).
                                          0: every step at cg_relative_tolerance (rounds 1-3).  3: as 1 without the exclusions by loss (Tukey, callback)
                                          and by conditioning -- the contraction gate and the restarts stay -- for callers who know their graphs.
                                          2 (a testing aid): every loose solve is continued
                                          to cg_relative_tolerance whatever its evaluation says -- the solve must then reproduce pcg_forcing = 0
                                          bit for bit, PCG iteration counts included (tests/test_gpu_round4.py). */
  int32_t dense_cholesky_auto_cams;    /* default 5333 (the largest matrix the exact step supports; 0 = off): graphs with more than
                                          dense_cholesky_max_cams and at most this many cameras start on PCG and switch to exact Cholesky steps --
                                          what the reference does for every graph, estimator.cpp:300 -- from the moment one PCG-solved step has
                                          cost more time (elapsed, host clock around the solve) than 1.25 x the factorisation of their size is measured to take on
                                          MI355X (0.40 ms at 394 cameras, 0.89 at 800, 2.25 at 1500, 11.8 at 3000; tools/bench_chol_large.hip).  Easy
                                          graphs (a few dozen PCG iterations per step) never switch; Madrid-like ones (hundreds) do after their
                                          first step.  Ignored for sharded problems and when dense_cholesky_max_cams < 0. */
  double pcg_forcing_tolerance;        /* default 1e-8 rad: largest estimated rms deviation of an inexact step from the exact one (or 5e-6 x the squared
                                          rms step size in radians, whichever is larger: a Gauss-Newton step of several degrees carries a far larger linearisation error) -- two orders below
                                          north_star's parity bar of 1e-6 rad (DESIGN.md section 6 has the measured trade: 1e-7 is 4 % faster on
                                          the benchmark graph and flips a borderline termination on one MAGSAC test graph).  A loose iterate's
                                          component along the gauge (all cameras rotated alike: the null space of J^T J, invisible to the energy
                                          norm) is removed before the step is taken, as the exact step has none (kernels.hpp, k_gauge_part; for the
                                          error types whose cost depends on R_j R_i^T alone, i.e. all but QUATERNION_NORM / ROTATION_MAT_FNORM). */
  int32_t lm_device_control;           /* default 1: for EXACT (Cholesky) steps of unsharded problems with a native loss the trust-region decisions
                                          of an LM iteration -- step validity, function / parameter tolerance, acceptance, the radius law -- are
                                          taken by a one-lane kernel from the scalars the step and cost kernels left on the device; the accept path
                                          (state copy, linearisation) is enqueued predicated on its verdict and the damping is rebuilt from the
                                          radius it wrote, so an iteration costs ONE host read-back instead of two and no host decision sits between
                                          its kernels.  Same formulas in the same order as the host loop: bit-identical trajectories
                                          (tests/test_gpu_round4.py).  0: the host loop everywhere. */
  int32_t component_rest;              /* default 1.  A DEPARTURE from Ceres' single global stopping rule, for disconnected view graphs (several
                                          scenes batched as one problem; csrc/solver_components.hpp) under a smooth loss: a connected component of at
                                          most dense_cholesky_max_cams cameras is PUT TO REST for the remainder of the solve once its exact
                                          (factorised) step has fallen below 1e-10 rad on every one of its cameras -- and has at least halved against
                                          the step measured before it, or the trust radius is at or above its initial value: a step that is small
                                          because the scene converged, not because another scene's rejections collapsed the shared radius --, and the PCG tolerance of such a problem has an absolute floor of 1e-11 rad per
                                          camera instead of 2e-14.  The reference has no such notion: ceres::Solve factorises every block of the
                                          block-diagonal system in every iteration until the GLOBAL function / gradient / parameter test fires
                                          (estimator.cpp:299-305).  The scenes of a batch are independent problems, a scene whose Newton step is
                                          1e-10 rad has at most ~1e-9 rad left to go, and resting it is what keeps 13 converged scenes from being
                                          factorised 30 more times while the 14th iterates on (BASELINE C4: 61 -> 46 ms); each component stays within
                                          1e-6 rad of the reference's answer (tests/test_gpu_round5.py, tests/manual/fuzz_components.py).  Never under
                                          the MAGSAC losses or a host-callback loss (1e-10 rad can be another table cell).
                                          0: off -- every component is solved in every LM iteration and the floor is 2e-14 rad, as on a connected graph.
                                          (This field was `reserved1_` until round 6: a caller that zero-filled it gets the reference's semantics.) */
} gsfm_rot_options;

typedef enum {
  GSFM_TERM_FUNCTION_TOLERANCE = 0,
  GSFM_TERM_GRADIENT_TOLERANCE = 1,
  GSFM_TERM_PARAMETER_TOLERANCE = 2,
  GSFM_TERM_NO_CONVERGENCE = 3, /* max_num_iterations reached */
  GSFM_TERM_FAILURE = 4         /* too many invalid steps / trust region collapsed / non-finite */
} gsfm_rot_termination;

typedef struct {
  int32_t termination;            /* gsfm_rot_termination */
  int32_t num_iterations;         /* LM iterations (iteration 0 = initial evaluation, not counted) */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_residual_sweeps;    /* full passes over all edges (linearisations + trial-cost evaluations) */
  int32_t num_linearizations;
  int32_t num_cg_iterations;      /* total PCG iterations (= normal-equation mat-vecs) */
  int32_t iters_to_1e6;           /* first LM iteration with |dcost|/cost <= 1e-6, -1 if never */
  int32_t nonfinite;              /* 1 if a NaN/Inf cost was seen */
  int32_t outer_iterations;       /* sigma-consensus only: IRLS outer iterations run */
  uint64_t num_edges_used;        /* residual blocks in the problem (this rank's cost-owned edges when sharded) */
  double initial_cost;
  double final_cost;
  double final_gradient_max_norm;
  double final_radius;
  double last_weight_change;      /* sigma-consensus: last mean |w - w_prev| */
  double t_total_ms;              /* host wall time of the call (uploads of the 3N rotations included) */
  double t_linearize_ms;          /* GPU time (HIP events) in the linearise kernel.  The three phase times are taken only for problems of
                                     2 M directed entries or more and never inside the device-controlled exact-step pipeline (an event
                                     record costs ~3 us on the stream: 11 % of a 10k-camera solve); otherwise they stay 0.
                                     GSFM_PHASE_TIMERS=1 / 0 in the environment forces them on / off */
  double t_sweep_ms;              /* GPU time in the residual+reweight cost sweep */
  double t_cg_ms;                 /* GPU time in PCG kernels (and dense solves) */
  int32_t num_dense_solves;       /* LM steps solved by the dense Cholesky path */
  int32_t num_graph_launches;     /* hipGraph replays of PCG chunks / dense solves in this call (0 = every kernel was launched plainly) */
  int32_t num_collectives;        /* sharded problems: collectives issued by this call (all-gathers + all-reduces, replayed ones included) */
  int32_t num_pcg_collectives;    /*   ... of which inside PCG iterations: exactly one per LAUNCHED iteration (the all-gather of the A.p slices) */
  int32_t num_pcg_launched;       /* PCG iterations enqueued (chunks of cg_check_interval: the ones past convergence return at once, but a sharded
                                     problem's collective in them still runs); num_cg_iterations counts the effective ones */
  int32_t num_forcing_refinements; /* forcing schedule: LM steps whose loose solve was continued to the tight tolerance after its trial evaluation */
  int32_t num_inexact_steps;       /* forcing schedule: LM steps taken from a loose solve */
  int32_t num_pcg_capped_steps;    /* (ABI v4) LM steps whose PCG solve ended at max_cg_iterations (or on cg_stall_iterations) ABOVE cg_relative_tolerance
                                      and were evaluated all the same: such a step is NOT the reference's exact Cholesky step (estimator.cpp:300).
                                      0 on every run whose parity claim holds; non-zero = the answer may differ from the reference's beyond the
                                      PCG tolerance (raise max_cg_iterations, or use the exact step: dense_cholesky_max_cams) */
  double worst_accepted_cg_residual; /* (ABI v4) largest relative residual sqrt(r.M^-1 r / b.M^-1 b) of a TIGHT PCG solve whose step was evaluated
                                      (<= cg_relative_tolerance unless num_pcg_capped_steps > 0); loose solves of the forcing schedule are
                                      counted by num_inexact_steps instead */
  int32_t num_forcing_restarts;    /* (ABI v4) 1 if the forcing schedule was abandoned mid-run and the solve was REDONE from the initial rotations
                                      with every step exact (the trajectory turned out not to contract fast enough for inexact steps to be
                                      forgotten, see pcg_forcing); the counters and times above include the abandoned attempt */
  int32_t reserved_;
} gsfm_rot_summary;

/* ------------------------------------------------------------------------- */
/* Multi-GPU sharding (one process per GPU)                                    */
/* ------------------------------------------------------------------------- */
/* Cameras (rows of the normal equations) are split into world_size contiguous
 * slices of `slice_width` rows: rank r owns [r*slice_width, min((r+1)*slice_width, n_cams)).
 * A rank passes the edges that touch at least one owned camera.  It evaluates the
 * directed (row) contributions of its owned cameras in full, so per-camera sums
 * need no reduction: slices are exchanged with an in-place all-gather; scalars
 * (cost, dot products) with a sum all-reduce.  Both callbacks receive DEVICE
 * pointers and must enqueue on `hip_stream` (or make it wait).
 * Failure handling: gsfm_rot_problem_create agrees on its status across ranks before
 * returning: every rank passes through ONE all-reduce after ALL of its rank-local work
 * (argument checks, structure build, uploads, loss set-up) -- or from its failure path --
 * and every rank fails if any did; what follows it inside create are collectives only.
 * Two rank-local failures cannot be agreed about and return at once, leaving the peers in
 * that all-reduce: a descriptor whose callbacks are NULL and a process without a HIP
 * device (the callbacks take device pointers).  Whether the GLOBAL view graph is connected
 * (it decides the PCG tolerance, see cg_relative_tolerance) is derived inside create from
 * all ranks' edges; GSFM_SHARD_DISCONNECTED is only a hint.  Inside a solve every decision (step acceptance, convergence, non-finite
 * cost, PCG termination) is taken from replicated, bitwise identical scalars, so all
 * ranks return the same status at the same point; what is NOT agreed on is a HIP
 * runtime error or a failing callback on one rank only (a lost device, a broken link):
 * that rank returns GSFM_ERR_HIP / GSFM_ERR_COMM while its peers wait in the next
 * collective until the communicator's own timeout or abort fires.                   */
#define GSFM_SHARD_CAPTURABLE 1u
#define GSFM_SHARD_DISCONNECTED 2u   /* the GLOBAL view graph has more than one connected component (see cg_relative_tolerance) */
typedef struct {
  int32_t rank;
  int32_t world_size;
  uint32_t slice_width;
  uint32_t flags;       /* GSFM_SHARD_CAPTURABLE: the callbacks only enqueue stream-ordered device work (no host synchronisation, no
                           host staging), so a chunk of PCG iterations containing them may be captured into a hipGraph (pcg_hip_graph >= 1).
                           GSFM_SHARD_DISCONNECTED: the partitioner may say that the global graph is disconnected; create works it out
                           from all ranks' local components anyway (one all-gather of a label per camera), the flag can only add to it */
  void* ctx;
  /* buf holds world_size * count doubles; rank r's input already sits at buf + r*count */
  int (*all_gather)(void* ctx, double* buf_dev, size_t count, void* hip_stream);
  int (*all_reduce_sum)(void* ctx, double* buf_dev, size_t count, void* hip_stream);
} gsfm_rot_shard;

/* ------------------------------------------------------------------------- */
/* Entry points                                                                */
/* ------------------------------------------------------------------------- */
typedef struct gsfm_rot_problem gsfm_rot_problem;

int gsfm_rot_abi_version(void);
const char* gsfm_last_error(void);
void gsfm_rot_options_default(gsfm_rot_options* opt);

/* Build the problem: replaces the edge loop that fills the ceres::Problem
 * (estimator.cpp:47-65, 110-166, 228-295).  Cameras are dense indices
 * 0..n_cams-1 (= rank of the ViewId among the views that HAVE an initial
 * orientation; edges touching other views are skipped by the caller, as the
 * reference does at :57-60).  (edge_i[e], edge_j[e]) is the ViewIdPair
 * (first, second); rel_aa[3e..] is TwoViewInfo::rotation_2.
 *   cov6          6 doubles per edge, order C00 C11 C22 C01 C02 C12 (the order of
 *                 covariance_rot.txt, src/uncertainty.cpp:185-195); required for the
 *                 *_COV* error types, ignored otherwise (may be NULL).
 *   inlier_weight one double per edge = (#common tracks)/100 (estimator.cpp:263,268);
 *                 required for the *_INLIERS types, ignored otherwise.
 *   shard         NULL for a single GPU.
 * The current HIP device of the calling thread is used.                        */
gsfm_status gsfm_rot_problem_create(uint32_t n_cams, uint64_t n_edges,
                                    const uint32_t* edge_i, const uint32_t* edge_j,
                                    const double* rel_aa, int32_t error_type,
                                    const double* cov6, const double* inlier_weight,
                                    const gsfm_rot_shard* shard,
                                    gsfm_rot_problem** out);
void gsfm_rot_problem_destroy(gsfm_rot_problem* p);

/* Run all work of this problem on an existing hipStream_t (e.g. torch's current
 * stream).  NULL = a stream owned by the problem (default). */
gsfm_status gsfm_rot_set_stream(gsfm_rot_problem* p, void* hip_stream);

/* Replaces the ceres::LossFunction* argument.  n_nodes == 0 -> NULL loss. */
gsfm_status gsfm_rot_set_loss(gsfm_rot_problem* p, const gsfm_loss_node* prog, int32_t n_nodes);
gsfm_status gsfm_rot_set_loss_callback(gsfm_rot_problem* p, gsfm_loss_callback fn, void* user);

/* Replace the per-edge scalar weights (only for the scalar-weight angle-axis
 * types; used by the sigma-consensus outer loop and exposed for callers). */
gsfm_status gsfm_rot_set_edge_weights(gsfm_rot_problem* p, const double* w /* n_edges */);

/* ceres::Solve replacement (estimator.cpp:72-78, 176-183, 299-306): LM on the
 * device; rot_aa_inout is the in/out global_orientations (3 doubles per camera). */
gsfm_status gsfm_rot_solve(gsfm_rot_problem* p, double* rot_aa_inout,
                           const gsfm_rot_options* opt, gsfm_rot_summary* summary);

/* The same solve for a caller whose rotations already live in device memory (a pipeline stage that runs on the GPU; bench.py's timed
 * region): rot_aa_dev_inout is a DEVICE pointer on the problem's device -- 3 doubles per camera, the caller's camera numbering, read at
 * entry, overwritten with the result -- and nothing crosses PCIe but the summary's scalars.  The buffer is read and written on the
 * PROBLEM's stream (gsfm_rot_set_stream): its producer must have finished, or be ordered before that stream.  Returns with the problem's stream
 * synchronised.  GSFM_ERR_INVALID_ARG for a host pointer.  (The reference's entry points take host maps, estimator.cpp:72-78: they map
 * to gsfm_rot_solve; this one has no counterpart there.)                                                                        */
gsfm_status gsfm_rot_solve_resident(gsfm_rot_problem* p, double* rot_aa_dev_inout,
                                    const gsfm_rot_options* opt, gsfm_rot_summary* summary);

/* EstimateRotationsWithSigmaConsensus (estimator.cpp:314-457): outer IRLS with
 * MAGSAC (nu = 3) weights from the residual norm, inner full LM solve.  The
 * problem must have been created with GSFM_ROT_ANGLE_AXIS.                      */
gsfm_status gsfm_rot_solve_sigma_consensus(gsfm_rot_problem* p, double* rot_aa_inout,
                                           int32_t iters_num, double sigma_max,
                                           const gsfm_rot_options* opt, gsfm_rot_summary* summary);

/* One residual + reweight sweep (kernel K1) at the given rotations; any output
 * may be NULL.  s_out[e] = ||r_e||^2, rho_out[3e..] = (rho, rho', rho''),
 * residual_out[stride*e..] = the raw residual (3, 4 or 9 values, see
 * gsfm_rot_residual_dim), *cost = sum 1/2 rho.  Outputs follow the edge order
 * given at create time (on the device the sweep stores in the problem's own edge
 * order, coalesced; the caller's order is restored at this boundary).           */
gsfm_status gsfm_rot_residuals(gsfm_rot_problem* p, const double* rot_aa,
                               double* s_out, double* rho_out, double* residual_out,
                               double* cost);
int32_t gsfm_rot_residual_dim(int32_t error_type);
/* The problem's own edge order: order_out[u] = index (in the arrays given at create time) of the edge at position u of the
 * device-side per-edge planes -- cost edges bucketed by (camera block of `first`, camera block of `second`).  Per-edge quantities
 * of the sweep K1 live on the device in THIS order; on a sharded problem only the edges this rank counts in the cost appear.
 * Copies min(cap, count) entries (order_out may be NULL) and returns the count, -1 on a NULL problem.                          */
int64_t gsfm_rot_edge_order(gsfm_rot_problem* p, uint32_t* order_out, uint64_t cap);

/* Linearise at rot_aa (kernel K2): gradient J~^T r~ (3 per camera) and the
 * diagonal 3x3 blocks of J~^T J~ (9 per camera, row-major), both with respect to
 * the reference's own parameters (additive angle-axis, or the quaternion local
 * parameterisation for the QUATERNION_ / ROTATION_MAT_ types).  Also keeps the
 * linearisation on the device for gsfm_rot_normal_matvec.                       */
gsfm_status gsfm_rot_linearize(gsfm_rot_problem* p, const double* rot_aa,
                               double* gradient, double* diag_blocks, double* cost);
/* y = (J~^T J~) v for the last linearisation (kernel K3), 3 per camera. */
gsfm_status gsfm_rot_normal_matvec(gsfm_rot_problem* p, const double* v, double* y);

/* ------------------------------------------------------------------------- */
/* The step after the solve + the edge statistic ("next" row f-3 of the scope)  */
/* ------------------------------------------------------------------------- */
/* One edge sweep with K1's device routines, no problem object needed:
 *   cov6 == NULL: s_out[e] = |log(R_ij^T R_j R_i^T)|^2, the squared loop angle that FilterViewPairsFromOrientation thresholds
 *                 (Theia filter_view_pairs_from_orientation.cc:55-122: keep iff s <= (max degrees in radians)^2);
 *   cov6 != NULL: s_out[e] = |Lt log(R_j R_i^T R_ij^T)|^2 with Lt from 1e8 * Sigma_e -- the square of what
 *                 residuals_of_relative_rot reports per edge (src/compare_reconstructions.cpp:617-647).
 * keep_out (may be NULL): 1 where s_out[e] <= max_sq_norm; *n_kept = their number.  kernel_ms (may be NULL): HIP-event time.  */
gsfm_status gsfm_rot_edge_sq_norms(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j,
                                   const double* rel_aa, const double* cov6, const double* rot_aa, double max_sq_norm,
                                   double* s_out, uint8_t* keep_out, uint64_t* n_kept, double* kernel_ms);

/* Host-only helper for partitioners: the locality relabelling gsfm_rot_problem_create would adopt for this graph on one GPU
 * (reverse Cuthill-McKee, kept only if it halves the mean index distance of the edges and brings it under 1024).
 * perm_out[c] = position of camera c in that order (the identity if nothing is to be gained).  Returns 1 if adopted, 0 if the
 * identity was returned, -1 on bad input.  Needs no device.  globalsfmpy_amd/sharding.py cuts the order into per-GPU slices. */
int32_t gsfm_rot_locality_order(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j,
                                uint32_t* perm_out);

/* Host-only: number of connected components of the view graph among the cameras that carry an edge (-1 on bad input). */
int64_t gsfm_rot_count_components(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j);

/* The problem's current native loss program evaluated ON THE DEVICE at the given squared norms, through the same device
 * routines (and kernel specialisation) the sweeps use: rho3_out[3k..] = (rho, rho', rho'')(s[k]) as K2's general path and the
 * per-edge sweep compute them, value_out[k] = rho(s[k]) as the solver's cost-only sweep computes it, rho1_fast_out[k] = rho'(s[k])
 * as K2's fast path computes it (losses with rho'' <= 0 everywhere: the single cheap leaves and the nu = 3 MAGSAC weight loss; NaN
 * for programs that have no fast path).  Any output may be NULL.
 * This is how the device is pinned directly against vectors recorded from scripts/loss_functions.py:47-458.              */
gsfm_status gsfm_rot_loss_eval(gsfm_rot_problem* p, const double* s, uint64_t n, double* rho3_out, double* value_out,
                               double* rho1_fast_out);

/* Per-iteration trace of the last solve: rows of GSFM_ROT_TRACE_COLS doubles
 * [iteration, cost, cost_change, gradient_max_norm, step_norm, relative_decrease,
 *  trust_region_radius, cg_iterations].  Returns the number of rows available.   */
#define GSFM_ROT_TRACE_COLS 8
int32_t gsfm_rot_get_trace(gsfm_rot_problem* p, double* out, int32_t cap_rows);

/* Timed loop of `reps` K1 sweeps with everything resident in HBM; returns the
 * mean kernel time (HIP events on the problem's stream).  For bench.py.         */
gsfm_status gsfm_rot_time_sweep(gsfm_rot_problem* p, const double* rot_aa, int32_t reps,
                                double* mean_kernel_ms);

/* The variants of the edge sweep K1 (and the sigma-consensus forms of K1 / K2), each as the mean of `reps` launches (HIP events on
 * the problem's stream, operands resident); per-edge outputs are stored in the problem's own edge order (gsfm_rot_edge_order):
 *   out_ms[0]  trial-cost sweep, as the solver launches it after every step: residual + rho VALUE, block-reduced, no per-edge store
 *   out_ms[1]  full sweep: residual, s and (rho, rho', rho'') stored per edge (32 B out per edge; what gsfm_rot_residuals runs)
 *   out_ms[2]  s-only sweep: pass 1 of host-callback losses (8 B out per edge)
 *   out_ms[3]  reweight sweep as SURVEY 8(d) defines it: residual, loss, rho' stored per edge (8 B out per edge)
 *   out_ms[4..7]  sigma consensus (0 unless the problem is an ANGLE_AXIS one that carries scalar weights): [4] the first cost sweep of an
 *              inner solve with the weight computation fused in (weights stored to the sweep's own plane, mean change reduced),
 *              [5] the same sweep without it, [6] the first linearisation K2 with the weights fused in, [7] K2 without
 * (In the solver the robust weights of an accepted point are evaluated inside K2, the linearisation; see time_kernels.) */
gsfm_status gsfm_rot_time_sweep_variants(gsfm_rot_problem* p, const double* rot_aa, int32_t reps, double* out_ms8);

/* Same for the hot kernels: out_ms[0] = K1 k_cost, [1] = the linearisation (K2, row-major or column-sorted form + its finish),
 * [2] = one normal-equation mat-vec (K3, likewise), [3] = reserved (0).  Mean of `reps` launches each, HIP events on the problem's
 * stream, operands resident in HBM.                                                                                       */
gsfm_status gsfm_rot_time_kernels(gsfm_rot_problem* p, const double* rot_aa, int32_t reps, double* out_ms4);

/* Bytes one mat-vec / one linearisation streams as laid out in HBM, and which form the problem uses: 0 = general 9-value blocks (76 B per
 * directed entry in the mat-vec), 1 = Laplacian form, row-major (52 B), 2 = Laplacian form, column-sorted row blocks (50 B per position with
 * the 2-byte delta-coded record -- graphs whose row blocks are dense in the cameras, e.g. the benchmark graph --, else 52, 54 from 2^19 cameras on;
 * + partial sums; large graphs without locality, csrc/colsort_kernels.hpp).  Any output may be NULL.  For roofline reporting. */
gsfm_status gsfm_rot_matvec_bytes(gsfm_rot_problem* p, double* matvec_bytes, double* linearize_bytes, int32_t* form);

/* Bytes moved per edge by one K1 sweep as laid out in HBM / as counted
 * algorithmically: for roofline reporting.                          */
gsfm_status gsfm_rot_sweep_bytes(gsfm_rot_problem* p, double* algorithmic_bytes,
                                 double* layout_bytes);

/* ------------------------------------------------------------------------- */
/* Per-edge rotation covariance ("next" row of the scope: src/uncertainty.cpp)  */
/* ------------------------------------------------------------------------- */
/* Batched get_covariance_rot (reference src/uncertainty.cpp:82-162, driver loop :164-198): for every view pair, refine
 * (rotation, translation) on the Sampson distance of its matched features -- translation on the sphere, TrivialLoss,
 * Ceres LM defaults, at most max_iterations (reference: 500) -- then the 3x3 covariance of the rotation with the
 * translation held fixed, (J_R^T J_R)^-1.  One wavefront per edge on the device.
 *   match_ptr        n_edges + 1 offsets into `matches`
 *   matches          4 doubles per match: x1 y1 x2 y2 (pixels)
 *   intrinsics       6 doubles per edge: f1 u1 v1 f2 u2 v2 (CameraIntrinsicsPrior focal length / principal point, :99-104)
 *   rot_in/trans_in  TwoViewInfo::rotation_2 / position_2 (3 doubles each)
 *   cov9_out         row-major 3x3 per edge (what ceres::Covariance::GetCovarianceBlock returns)
 *   status_out       0 ok, 1 skipped (zero translation, :123, or no matches), 2 rank-deficient information matrix (a pivot of
 *                    its pivoted Cholesky below 1e-14 of the first, ceres::Covariance's reciprocal-condition bound: identical or
 *                    collinear matches; where the reference's CHECK(covariance.Compute(...)) aborts, :157) -- no covariance
 *   iters_out        LM iterations used (may be NULL)                                                                */
gsfm_status gsfm_cov_estimate(uint64_t n_edges, const uint64_t* match_ptr, const double* matches, const double* intrinsics,
                              const double* rot_in, const double* trans_in, int32_t max_iterations, double* cov9_out,
                              double* rot_out, double* trans_out, int32_t* status_out, int32_t* iters_out,
                              double* kernel_ms /* may be NULL: HIP-event time of the kernel */);

/* MAGSAC lookup table Gamma((nu-1)/2, x/1000), x = 0..n-1 (include/gamma_values.cpp);
 * regenerated analytically.  Returns the table length; copies min(n, cap) values. */
int32_t gsfm_magsac_table(int32_t nu, double* out, int32_t cap);
/* C_nu, sigma_quantile_nu, upper_incomplete_gamma_of_k_nu (gamma_values.cpp:6-11,384-389,780-785) */
gsfm_status gsfm_magsac_constants(int32_t nu, double* C, double* sigma_quantile,
                                  double* upper_incomplete_gamma_of_k);

#ifdef __cplusplus
}
#endif
#endif /* GSFM_ROT_H_ */
