/* rdhip.h -- C ABI of librdhip: batched RobotDynamics.jl dynamics / Jacobian evaluation on MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of RobotDynamics.jl v0.4.8.  The reference has no FFI for
 * this path (it is Julia multiple dispatch, one knot point per call); each entry point below replaces a
 * per-knot Julia loop `for k in 1:N; f(model, ..., Z[k]); end` by ONE call over a packed trajectory.
 * All file:line citations are into the reference checkout (RobotDynamics.jl/).
 *
 * Conventions
 *   - plain C: pointers + sizes, `int` status return (0 = OK, negative = rdhip_status), no exceptions,
 *     no callbacks, no ownership transfer.  Pointers are not retained after the call returns.
 *   - packed trajectory ("PackedTrajectory"): Z is (n+m) x N column-major, knot point k occupies
 *     Z[k*(n+m) .. k*(n+m)+n+m), z = [x;u] exactly as KnotPoint.z (src/knotpoint.jl:146-153);
 *     t[N], dt[N] are ALWAYS fp64 (KnotPoint.t/.dt are Float64, src/knotpoint.jl:150-151).
 *   - terminal knot point: dt[k] == 0  =>  control reads as zeros (src/knotpoint.jl:67,123), and the control columns of a ForwardAD /
 *     FiniteDifference Jacobian and of every discrete rule's Jacobian are zero (they differentiate through `!is_terminal(z) * u`, or carry a factor
 *     dt).  RDHIP_USER_DEFINED with RDHIP_CONTINUOUS (or a custom model's own Jacobian body) hands back d f / d u at u = 0 as the reference does:
 *     jacobian!(fun, J, y, state(z), control(z)), src/functionbase.jl:252-257.
 *   - Jacobians: J is n x (n+m) x N, each n x (n+m) slab column-major = the storage of
 *     DynamicsJacobian (src/jacobian.jl:26-37): A = J[:, 0:n], B = J[:, n:n+m].
 *   - dtype selects the arithmetic and storage type of Z / outputs (RDHIP_F64 or RDHIP_F32).
 *   - memspace: RDHIP_DEVICE = all data pointers are device pointers and the call is asynchronous on the
 *     context's HIP stream; RDHIP_HOST = all data pointers are host pointers, the library stages them
 *     through a per-context device workspace and the call returns after the results are back on the host.
 *   - a context owns one HIP stream (or borrows the caller's) and is re-entrant per context: use one context
 *     per host thread (the reference's model objects carry mutable caches and are likewise not thread-safe,
 *     src/integration.jl:258-276).
 *   - there is NO CPU fallback: every compute entry point fails with RDHIP_ERR_NO_DEVICE without a GPU.
 */
#ifndef RDHIP_H
#define RDHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RDHIP_VERSION 10000 /* 1.0.0 */

typedef enum {
    RDHIP_OK = 0,
    RDHIP_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad enum            (Julia: ArgumentError / AssertionError) */
    RDHIP_ERR_UNSUPPORTED = -2,   /* model/rotation/rule combination not compiled in  (Julia: NotImplementedError, src/utils.jl:1-8) */
    RDHIP_ERR_NOT_IMPLEMENTED = -3, /* UserDefined Jacobian missing for this model    (src/functionbase.jl:257) */
    RDHIP_ERR_HIP = -4,           /* a HIP runtime call failed; see rdhip_last_error   */
    RDHIP_ERR_NO_DEVICE = -5,     /* no usable GPU                                     */
    RDHIP_ERR_DIM_MISMATCH = -6   /* leading dimension smaller than the state size etc. */
} rdhip_status;

/* model registry: user-defined Julia `dynamics` cannot run on the GPU, so the models of the reference's
 * examples/ and test/ fixtures are compiled in.  params layout is given per kind. */
typedef enum {
    RDHIP_MODEL_CARTPOLE = 0,          /* examples/cartpole.jl:5-38      params = {mc, mp, l, g}                              */
    RDHIP_MODEL_DOUBLE_INTEGRATOR = 1, /* test/double_integrator.jl:97   params = {D}, D in 1..3  (n = 2D, m = D)             */
    RDHIP_MODEL_LINEAR = 2,            /* xdot = A x + B u (KAT model)   params = {n, m, A[n*n] col-major, B[n*m] col-major};
                                          (n,m) in {(2,1),(4,2),(6,3)}                                                         */
    RDHIP_MODEL_SATELLITE = 3,         /* RigidBody{R}: examples/single_satellite.jl:7-27, test/rigidbody_test.jl:23-54
                                          F = q*u[1:3] (world), tau = u[4:6]   params = {mass, J[9] col-major}                */
    RDHIP_MODEL_QUADROTOR = 4,         /* RigidBody{R}: test/quadrotor.jl:21-137
                                          params = {mass, J[9], gravity[3], motor_dist, kf, km}                               */
    RDHIP_MODEL_SATELLITE_PAIR = 5,    /* LieGroupModel, LieState(R,(0,3,3)): examples/satellite_pair.jl:8-40  params = {J1[9], J2[9]} */
    RDHIP_MODEL_LIE_SATELLITE = 6,     /* LieGroupModel, LieState(R,(3,0)), x = [w; q]: test/liemodel.jl:1-22  params = {Jdiag[3]}     */
    RDHIP_MODEL_CUSTOM = 7             /* user-written dynamics, JIT-compiled with hipRTC: rdhip_model_create_custom                  */
} rdhip_model_kind;

typedef enum { RDHIP_ROT_QUAT = 0, RDHIP_ROT_MRP = 1, RDHIP_ROT_RP = 2 } rdhip_rotation;   /* src/liestate.jl:42-46 */
typedef enum { RDHIP_FRAME_WORLD = 0, RDHIP_FRAME_BODY = 1 } rdhip_velocity_frame;          /* src/rigidbody.jl:258 */
/* the `errmap` argument of state_diff (src/liestate.jl:26-32,216-217): Rotations.jl's ErrorMap types [third party, restated: SURVEY App. A] */
typedef enum { RDHIP_ERRMAP_CAYLEY = 0, RDHIP_ERRMAP_EXPONENTIAL = 1, RDHIP_ERRMAP_MRP = 2, RDHIP_ERRMAP_QUATVEC = 3, RDHIP_ERRMAP_IDENTITY = 4 } rdhip_errmap;
/* QuadratureRule (src/integration.jl:69,109,258; RK2 = v0.3 explicit midpoint); RDHIP_CONTINUOUS = no discretisation */
/* RDHIP_IMPLICIT_MIDPOINT (src/integration.jl:620-729): x1 + h f((x1+x2)/2, u1) - x2 = 0 solved by Newton's method (src/integration.jl:422-463);
 * accepted by rdhip_discrete_dynamics, rdhip_jacobian (implicit function theorem, src/integration.jl:517-538), rdhip_dynamics_error[_jacobian] */
/* RDHIP_DISCRETE: the model's OWN step -- the only rule of a user-defined DiscreteDynamics model (rdhip_model_create_custom_discrete), and of no other model */
typedef enum { RDHIP_CONTINUOUS = -1, RDHIP_EULER = 0, RDHIP_RK2 = 1, RDHIP_RK3 = 2, RDHIP_RK4 = 3, RDHIP_IMPLICIT_MIDPOINT = 4, RDHIP_DISCRETE = 5 } rdhip_rule;
typedef enum { RDHIP_FORWARD_AD = 0, RDHIP_FINITE_DIFF = 1, RDHIP_USER_DEFINED = 2 } rdhip_diffmethod; /* src/functionbase.jl:109-120 */
typedef enum { RDHIP_F64 = 0, RDHIP_F32 = 1 } rdhip_dtype;
typedef enum { RDHIP_DEVICE = 0, RDHIP_HOST = 1 } rdhip_memspace;
typedef enum { RDHIP_STREAM_OWN = 0, RDHIP_STREAM_BORROW = 1 } rdhip_stream_mode;

typedef struct rdhip_ctx rdhip_ctx;
typedef struct rdhip_model rdhip_model;

/* ---------------------------------------------------------------- library / context ------------------ */
int rdhip_version(void);
int rdhip_device_count(int* count);
/* stream_mode RDHIP_STREAM_OWN: the context creates (and owns) a non-blocking HIP stream, `stream` is ignored;
 * RDHIP_STREAM_BORROW: `stream` is a hipStream_t owned by the caller (e.g. torch.cuda.current_stream().cuda_stream;
 * NULL = the device's default stream) and all launches go to it; it must be alive whenever a call is made on the context (the library keeps no
 * stream of another context: the device-wide ordering of fused rollout -> Jacobians launches goes through an event recorded at launch time). */
int rdhip_ctx_create(int device_id, void* stream, int stream_mode, rdhip_ctx** out);
int rdhip_ctx_destroy(rdhip_ctx* ctx);
int rdhip_ctx_synchronize(rdhip_ctx* ctx);
/* Launch overlap.  Back-to-back kernels on ONE stream cannot overlap: a launch's prologue (first inputs, first columns: ~2 us of a 27 us flagship
 * launch) and its drain (~2.5 us) are exposed on every call.  With overlap on -- contexts that OWN their stream only -- consecutive rdhip_jacobian
 * calls on device arrays (compiled-in models; RDHIP_IMPLICIT_MIDPOINT when Y is given) alternate between the context's stream and one internal stream, so that a call's
 * prologue runs under its predecessor's tail (the per-knot loop `for k: jacobian!(sig, diff, model, J[k], y, Z[k])`, src/functionbase.jl:242, of two
 * batches side by side).  ORDERING RULE: results are what in-order execution gives.  A call is ordered behind every earlier call whose arrays
 * (Z, t, dt read; J, Y written) intersect its own other than read-read -- an event edge between the two streams exactly there --, every other
 * rdhip operation on the context (any other entry point, timers, graph capture, memcpy helpers, rdhip_ctx_stream, rdhip_ctx_synchronize) is ordered
 * behind everything issued before it, and rdhip_ctx_fence orders the context's stream behind everything issued so far (call it before handing
 * the stream from rdhip_ctx_stream to other code).  Off by default. */
int rdhip_ctx_set_overlap(rdhip_ctx* ctx, int on);
int rdhip_ctx_fence(rdhip_ctx* ctx);
void* rdhip_ctx_stream(rdhip_ctx* ctx);
/* Developer tuning knobs.  The dispatcher chooses kernels and launch shapes by size; every switch that overrides a choice lives in ONE list, read once
 * per process from the environment variable RDHIP_TUNE="key=value,key=value" (unknown keys are reported on stderr).  rdhip_tune() merges more pairs at
 * run time ("reset" / NULL: back to the environment's); it is meant for tests and tuning scripts, is not synchronised against concurrent launches,
 * and NO knob changes a result.  Keys (csrc/common.hpp TuneKnobs):
 *   static_grid=N            workgroups of the one-per-CU persistent launches (default: one per compute unit)
 *   dyn_grid=N               workgroups of launches with the dynamic tile schedule (default: three quarters of the compute units)
 *   dynamic_tiles=0          no launch takes the dynamic tile schedule
 *   small=tile               the persistent tile loop for every batch size (no four-workgroups-per-tile form);  small_max=N  its limit in workgroups per CU (2)
 *   rollout=plain|cascade|stream   force a rollout kernel where it applies
 *   rolljac=split            rdhip_rollout_jacobian always as two launches
 *   fused_ratio=x, fused_jac_workgroups=N, fused_tj_max_knots=N, rolljac_pipe_min_knots=N, multi_main=i     shape of the one-launch rollout -> Jacobians forms
 *   fused_two_per_cu=0       the fp32 one-launch forms keep one workgroup per compute unit (the form of rounds 4-5; default: two where the Jacobians bound the launch)
 *   mixed=0, mixed_idle=0|1  rdhip_rollout_jacobian_multi: never the one-launch form of a (satellite, cartpole) pair / where the cartpole's rollout runs in it
 *   ift=reg|3                ImplicitMidpoint Jacobians: the one-kernel register path / the three-launch path of round 2
 * Not tuning, and still their own environment variables: RDHIP_DRY_RUN (contexts without a device: queries only), RDHIP_VERBOSE (reports fall-backs on
 * stderr), RDHIP_CACHE_DIR (hipRTC code-object cache of custom models), RDHIP_RCCL_LIB / RDHIP_COMM_FORCE_RCCL (communicator). */
int rdhip_tune(const char* pairs);
/* message of the last failure on this context (ctx == NULL: last failure of a context-less call on this thread) */
const char* rdhip_last_error(rdhip_ctx* ctx);
/* Name (demangled, NUL-terminated, truncated to `capacity`) of the kernel the CALLING HOST THREAD launched last through the library: which
 * specialisation a call such as jacobian!(sig, diff, fun, J, y, z) (src/functionbase.jl:242) was dispatched to -- the reference resolves that by
 * Julia method dispatch (`@which`); here a benchmark or a test asks instead of assuming (bench.py's roofline.kernel).  Empty before the first launch.
 * A launch that was handed the counter words of the dynamic tile schedule has " [dynamic tile schedule]" appended (most kernels carry the schedule in their one
 * instantiation and take it at run time: the name alone does not say). */
int rdhip_last_kernel_name(char* buf, int capacity);
/* HIP-event stopwatch on the context's stream (for hosts without their own event API) */
int rdhip_timer_start(rdhip_ctx* ctx);
int rdhip_timer_stop(rdhip_ctx* ctx, float* elapsed_ms); /* synchronises on the stop event */
/* HIP graphs.  A batch of a few thousand knot points is launch-bound (an MI355X kernel launch costs more than streaming its output), and a
 * solver iteration issues the same few calls on the same arrays every time: record them once, replay them with one launch.
 *   rdhip_graph_begin_capture(ctx); <any RDHIP_DEVICE calls on ctx: they are recorded, not executed>; rdhip_graph_end_capture(ctx, &g);
 *   rdhip_graph_launch(ctx, g) enqueues the whole sequence on the context's stream (array CONTENTS may change between launches,
 *   addresses and sizes may not).  Run every call once before capturing (first use sets kernel attributes, JIT-compiles custom models
 *   and grows workspaces -- none of that can happen inside a capture).  RDHIP_HOST calls are refused while capturing. */
typedef struct rdhip_graph rdhip_graph;
int rdhip_graph_begin_capture(rdhip_ctx* ctx);
int rdhip_graph_end_capture(rdhip_ctx* ctx, rdhip_graph** out);
int rdhip_graph_launch(rdhip_ctx* ctx, rdhip_graph* graph);
int rdhip_graph_destroy(rdhip_graph* graph);
/* device memory helpers for hosts that do not own a device allocator (the Julia wrapper) */
int rdhip_malloc(rdhip_ctx* ctx, void** ptr, unsigned long long bytes);
int rdhip_free(rdhip_ctx* ctx, void* ptr);                                   /* ctx may be NULL (finalizers of a garbage-collected host) */
int rdhip_memcpy_h2d(rdhip_ctx* ctx, void* dst, const void* src, unsigned long long bytes); /* async on ctx stream */
int rdhip_memcpy_d2h(rdhip_ctx* ctx, void* dst, const void* src, unsigned long long bytes); /* synchronous */

/* ---------------------------------------------------------------- models and integer queries --------- */
/* host-only object (no GPU needed).  rotation: the R of RigidBody{R} / LieState{R,P} (src/rigidbody.jl:46, src/liestate.jl:75-83).  RDHIP_ROT_QUAT serves
 * every operator.  RDHIP_ROT_MRP / RDHIP_ROT_RP (n = 12 rigid bodies, src/rigidbody.jl:80-94, test/rigidbody_test.jl:79-98): dynamics, the explicit rules
 * and ImplicitMidpoint, ForwardAD / FiniteDifference Jacobians, rollout, errstate_jacobian!, state_diff for the SATELLITE and QUADROTOR kinds;
 * rdhip_error_expansion (Jacobian stack + projection with the 3 x 3 blocks, two launches); errstate_jacobian!, state_diff and nabla-errstate_jacobian!
 * for every LieState; RDHIP_USER_DEFINED -> RDHIP_ERR_NOT_IMPLEMENTED (the reference's analytic Jacobian is QuatRotation-only, src/rigidbody.jl:260). */
int rdhip_model_create(int kind, int rotation, int velocity_frame, const double* params, int nparams, rdhip_model** out);
int rdhip_model_destroy(rdhip_model* model);
/* A user-defined ContinuousDynamics model (the reference's `dynamics(model, x, u)` method, src/dynamics.jl:23-45), given as HIP source:
 * `dynamics_body` is the BODY of
 *       template <class S> __device__ void f(const S* x, const S* u, T t, S* xdot) const
 * written on a generic scalar S (the library instantiates it with T = float/double and with forward duals), with the model parameters
 * available as `p[i]` (type T), the time as `t` (type T: dynamics(model, x, u, t), src/dynamics.jl:81-83 -- the knot point's time for
 * rdhip_dynamics, the stage times t, t + h/2, t + h inside the Runge-Kutta rules, src/integration.jl:130-135,280-286; a body that never
 * mentions `t` is a time-invariant model, as all of the reference's examples are) and the dual-aware functions sin_, cos_, sqrt_, recip_, max0_ plus + - * / (scalar.hpp); `preamble` (may be NULL)
 * is pasted at namespace scope in front of the model (helper device functions).  Kernels are JIT-compiled with hipRTC for gfx950 on
 * first use per (operation, rule, dtype) and cached in the model; every rule, diff method and dtype of the compiled-in models works.
 * LieState models: pass rotation = RDHIP_ROT_QUAT and the partition P (as LieState{R,P}, src/liestate.jl:75-83), nP = 0 for Euclidean.
 * n <= 16, m <= 8, nparams <= 64.  Compilation errors come back as RDHIP_ERR_INVALID_ARG with the compiler log in rdhip_last_error(NULL)
 * (creation does not compile; the first use or rdhip_model_precompile does).  With the environment variable RDHIP_CACHE_DIR set, compiled
 * kernels are kept there as code objects (keyed by a hash of the generated source, the kernel templates and the HIP version) and later
 * processes load them instead of compiling.
 * USER BODIES ARE COMPILED WITH THE LIBRARY'S FLAGS, -ffinite-math-only -fno-signed-zeros among them (this applies to dynamics, step and Jacobian bodies
 * alike): the compiler may assume that no operand is NaN or Inf, so a guard written as `x != x`, `isnan(x)` or `isinf(x)` in a body can be folded away.
 * A body that must detect a poisoned input tests the BIT PATTERN (`__double_as_longlong(x) & 0x7ff0000000000000` == the all-ones exponent), as the
 * library does for the terminal-knot rule. */
int rdhip_model_create_custom(int n, int m, const char* dynamics_body, const char* preamble, const double* params, int nparams,
                              int rotation, const int* P, int nP, rdhip_model** out);
/* The user's OWN Jacobian of a custom continuous model -- `jacobian!(model, J, xdot, x, u, t)`, src/dynamics.jl:115-116 (test/cartpole_model.jl:57-96 is the
 * cartpole's) -- as the body of
 *       __device__ void jac(const T* x, const T* u, T t, T* J) const        J: n x (n+m), column-major (J[i + n * j]), ZERO on entry
 * in plain T arithmetic (sin, cos, sqrt ... are the HIP device functions; p[i] the parameters).  RDHIP_USER_DEFINED then evaluates THESE formulas:
 * directly for rule RDHIP_CONTINUOUS, and through the integrators' chain rule on the stage Jacobians (src/integration.jl:85-93, :149-177, :302-337:
 * the reference's literal recipe) for Euler / RK2 / RK3 / RK4 of models with n + m <= 10.  Bigger discretized models keep what every custom model
 * without a Jacobian body gets: the same chain rule taken matrix-free on forward duals of the dynamics (equal in exact arithmetic; the user's
 * formulas are then not consulted).  NULL or "" removes the body.  Compiled kernels of the model are discarded. */
int rdhip_model_set_user_jacobian(rdhip_model* model, const char* jacobian_body);
/* A user-defined DiscreteDynamics model -- `discrete_dynamics(model, x, u, t, dt)` on a DiscreteDynamics subtype, src/discrete_dynamics.jl:1-108
 * (test/discrete_dynamics_test.jl:5-63 is the reference's discrete double integrator) -- given as the body of
 *       template <class S> __device__ void step(const S* x, const S* u, T t, T h, S* xn) const
 * (S as for rdhip_model_create_custom; t and h = the knot point's time and time step, plain T) and, optionally (NULL: none), the body of its own
 *       __device__ void jac(const T* x, const T* u, T t, T h, T* J) const   -- `jacobian!(model, J, xn, x, u, t, dt)`, src/discrete_dynamics.jl:111-113
 * with J as above.  Every call on such a model takes rule = RDHIP_DISCRETE and nothing else (it has no quadrature rule and no continuous
 * dynamics: rdhip_dynamics and the implicit rule answer RDHIP_ERR_INVALID_ARG): rdhip_discrete_dynamics, rdhip_jacobian[_multi] with RDHIP_FORWARD_AD (the
 * reference's default for a DiscreteDynamics model, what `@autodiff` generates) / RDHIP_FINITE_DIFF / RDHIP_USER_DEFINED (the jac body;
 * RDHIP_ERR_NOT_IMPLEMENTED without one, "User-defined discrete dynamics Jacobian not defined."), rdhip_rollout[_layout], rdhip_rollout_jacobian[_multi],
 * rdhip_dynamics_error[_jacobian] (J2 = [-I 0], src/discrete_dynamics.jl:186-201) and, with a LieState partition, the error-state operators and
 * rdhip_error_expansion.  No cascade declaration, no column balancing (rdhip_model_set_cascade / _balance_columns: continuous models). */
int rdhip_model_create_custom_discrete(int n, int m, const char* step_body, const char* jacobian_body, const char* preamble, const double* params, int nparams,
                                       int rotation, const int* P, int nP, rdhip_model** out);
int rdhip_model_is_discrete(const rdhip_model* model, int* discrete);
/* setnewtoniters / setnewtontolerance of an ImplicitDynamicsModel (src/discretized_dynamics.jl:225-228); defaults 10 and 1e-12
 * (src/integration.jl:599-600).  The Newton loop stops per knot point when ||residual||_2 < tol.
 * PARITY TOLERANCE of the implicit rule: two correct Newton iterations that stop at ||residual|| < tol return roots that differ by up to tol times the
 * conditioning of the residual's Jacobian (the iteration has no further step that would make them agree), and the implicit-function-theorem Jacobian is
 * linearised about that root.  The parity gate for RDHIP_IMPLICIT_MIDPOINT outputs is therefore 1e-9 on x_next and 1e-8 x max(1, |J|) on the Jacobian at
 * the default tol = 1e-12 (observed: <= 3e-12 and <= 4e-11), not the 1e-10 of the explicit rules. */
int rdhip_model_set_newton(rdhip_model* model, int newton_iters, double newton_tol);
/* Custom models with n+m > 10: measure the cost of every Jacobian column (each is JIT-compiled alone, ~0.5 s per column, no GPU needed)
 * and assign the columns to the compute waves of the Jacobian kernel by cost instead of in contiguous ranges.  Optional; call it before
 * the first Jacobian (kernels compiled earlier are discarded). */
int rdhip_model_balance_columns(rdhip_model* model);
/* Declare the block structure of a custom model's dynamics -- what `Cascade<>` states for the compiled-in rigid bodies (angular velocity | attitude |
 * position, velocity: src/rigidbody.jl:213-236): state_block[i] in 0 .. nblocks-1 is the block of state component i, and the time derivative of a
 * block may read blocks 0 .. b only; control_block[j] is the first block whose derivative reads control j.  With a declaration the model gets the
 * kernels that exploit it: FiniteDifference that re-integrates only what a column can change, rollout! with one wavefront per block, Newton's method
 * block by block for ImplicitMidpoint.  The declaration is verified on first use on a GPU (the continuous ForwardAD Jacobian at 32 points must have exact
 * zeros wherever it promises them; RDHIP_ERR_INVALID_ARG with the offending entry otherwise; the check allocates and synchronises, so the first use
 * must not sit inside a graph capture -- it is refused there).  2 <= nblocks <= 6, at most 8 state components per block; nblocks <= 1 removes the
 * declaration.  The one-wavefront-per-block rollout kernels additionally need their LDS ring (stage points of all blocks but the last x 4 steps)
 * plus the I/O tiles to fit the 160 KiB of a CU and K, M < 2^26: a declaration that does not fit (e.g. 16 fp64 states with a 12-state upstream part
 * under RK4) keeps the plain one-lane-per-trajectory rollout; FiniteDifference and Newton's method use the declaration regardless.
 * Kernels compiled before the call are recompiled on next use. */
int rdhip_model_set_cascade(rdhip_model* model, int nblocks, const int* state_block, const int* control_block);
/* compile (no GPU needed) the kernel of one operation: op 0 = dynamics / discrete dynamics, 1 = rollout, 2 = ForwardAD Jacobian,
 * 3 = FiniteDifference Jacobian, 4 = UserDefined chain-rule Jacobian, 7 / 8 / 9 / 10 = ImplicitMidpoint solve / solve + Jacobian /
 * dynamics error / rollout (rule = RDHIP_IMPLICIT_MIDPOINT), 12 = fused error-state expansion (LieState models), 13 / 14 / 15 = the register-matrix
 * forms of 7 / 8 / 10 (n <= 14; run first, 7 / 8 / 10 only when a row exchange is needed), 16 = the implicit-function-theorem Jacobian as a tile
 * kernel (n + m > 10: runs between 13 and 8), 17 = the implicit rollout along a declared cascade (rdhip_model_set_cascade), 18 = the plain rollout
 * whatever is declared, 32 + op (op = 2 .. 6, 11) = the Jacobian kernels of rdhip_jacobian_multi; custom models only */
int rdhip_model_precompile(rdhip_model* model, int op, int rule, int dtype);
/* dims(model) = (n, m, p) and errstate_dim: src/functionbase.jl:130-135, src/dynamics.jl:3, src/liestate.jl:24,124,207 */
int rdhip_model_dims(const rdhip_model* model, int* n, int* m, int* p, int* e);
/* statevectortype: 0 = EuclideanState, 1 = RotationState (src/statevectortype.jl:63,73; src/liestate.jl:22) */
int rdhip_model_statevectortype(const rdhip_model* model, int* is_rotation_state);
/* LieState(model) = LieState{R,P}: rotation kind and partition P (src/liestate.jl:75-83); nP = 0 for Euclidean models */
int rdhip_model_liestate(const rdhip_model* model, int* rotation, int* nP, int* P /* capacity >= 8 */);
/* gen_inds(model::RigidBody): 1-based first index and length of r, q, v, w, u (src/rigidbody.jl:80-94) */
int rdhip_model_gen_inds(const rdhip_model* model, int first[5], int len[5]);
/* LieState index algebra for an arbitrary (R, P) (src/liestate.jl:42-46,121-131): length(s), errstate_dim(s),
 * first(rot_inds(R,P,i)) for i = 1..nP-1 and first(vec_inds(R,P,i)) for i = 1..nP, all 1-based as in Julia */
int rdhip_liestate_query(int rotation, const int* P, int nP, int* length, int* errstate_dim, int* rot_first, int* vec_first);

/* ---------------------------------------------------------------- batched hot path ------------------- */
/* NaN / Inf CONTRACT of every batched entry point.  The reference creates trajectories filled with NaN (src/trajectories.jl:83) and stores dt = Inf on
 * a last knot point that carries a control (:104-108), so both reach these calls.
 *   - A knot point (or trajectory) that holds a NaN or an Inf changes NOTHING about the others: knot points are independent, and a poisoned one's
 *     neighbours in its 64-knot tile get the bits they get without it.  One exception, to rounding (1e-13), not to the bit: ImplicitMidpoint
 *     Jacobians, where a tile with a knot point whose Newton iteration does not converge is redone by the exact LDS kernel, whose operation order
 *     differs from the register kernel's.
 *   - Every call returns: Newton stops at its iteration limit (a NaN residual never passes the convergence test), no loop waits for a value.
 *   - Function values (dynamics, discrete dynamics, x_next, errors, rollouts from that knot on) are NaN in every component whose formula reads the
 *     poisoned input; `dt === 0` is tested on the bit pattern, so dt = NaN / Inf is "not terminal".
 *   - Jacobian entries of a poisoned knot point are NaN wherever the entry's formula reads the poisoned input, and may be FINITE elsewhere: tangent
 *     seeds are compile-time constants here, so the `0 * NaN` products by which ForwardDiff spreads a NaN into every entry do not exist (a NaN force
 *     control of the satellite leaves d x_next / d u finite: the step is linear in u).  Structural zeros stay exact zeros.
 * (The library is built with -ffinite-math-only -- it is what lets the compiler fold those seeds; comparisons whose operand may be NaN are done on
 * bit patterns.  tests/test_gpu_round5.py holds the contract to the kernels.) */
/* dynamics(model, z) for every knot point: src/dynamics.jl:81-83 (RigidBody: src/rigidbody.jl:213-236).
 * Xdot: n x N. */
int rdhip_dynamics(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N,
                   const void* Z, const double* t, const double* dt, void* Xdot, int memspace);

/* discrete_dynamics(DiscretizedDynamics{rule}(model), z): src/discrete_dynamics.jl:80-81,
 * src/discretized_dynamics.jl:203-204, src/integration.jl:73-83 (Euler), :130-135 (RK3), :280-286 (RK4).
 * v0.3 spelling: discrete_dynamics(rule, model, z).  Xn: n x N. */
int rdhip_discrete_dynamics(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long N,
                            const void* Z, const double* t, const double* dt, void* Xn, int memspace);

/* jacobian!(sig, diff, fun, J, y, z) for every knot point (src/functionbase.jl:242-257), where fun is the
 * continuous model (rule = RDHIP_CONTINUOUS; generated methods src/jacobian_gen.jl:485-529, :643-685) or
 * DiscretizedDynamics{rule}(model) (src/discretized_dynamics.jl:169,208-219; chain rule src/integration.jl:85-93,
 * :149-177,:302-337).  v0.3 spellings: jacobian!(D, model, z) / discrete_jacobian!(rule, D, model, z).
 * J: n x (n+m) x N, fully overwritten.  Y: n x N or NULL; receives f(z) (what InPlace leaves in `y`).
 * rule = RDHIP_IMPLICIT_MIDPOINT: diff = RDHIP_FORWARD_AD stands for ImplicitFunctionTheorem(ForwardAD()) (the only DiffMethod the reference
 * accepts for implicit models, src/discretized_dynamics.jl:243-249): the next state is solved for, then J = -J2[:,1:n] \ J1.
 * fp64 batches of 8 or more 64-knot tiles per workgroup (131 072 knot points on a whole MI355X) are scheduled dynamically (a device-wide counter hands
 * out the tiles; same results, bit for bit); RDHIP_TUNE=dynamic_tiles=0 in the environment keeps the static order, and so does a capturing stream. */
int rdhip_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int rule, int diff, int dtype, long long N,
                   const void* Z, const double* t, const double* dt, void* J, void* Y, int memspace);

/* rdhip_jacobian over `nbatch` INDEPENDENT packed trajectories (different arrays; the per-knot loops of nbatch TrajectoryOptimization problems
 * or multiple-shooting segments: `for p in problems, k in 1:N[p]; jacobian!(sig, diff, model, J[p][k], y, Z[p][k]); end`, src/functionbase.jl:242)
 * in ONE kernel launch: N[b], Z[b], t[b], dt[b], J[b], Y[b] are the arguments rdhip_jacobian would take for batch b (Y or any Y[b] may be NULL).
 * A trajectory of a few thousand knot points is launch-bound on its own; enumerated together by the same persistent tile loop, 16 x 4 096 knot
 * points cost what one 65 536-knot batch costs, and 1 024 trajectories of 101 knot points (examples/cartpole.jl:46) are one bandwidth-regime launch.
 * The tables N / Z / t / dt / J / Y themselves are HOST arrays of nbatch entries (not retained); memspace describes the data pointers inside them.
 * Up to 32 non-empty batches travel in the kernel arguments; more are copied into a device table owned by the context (a binary search per tile,
 * none when all batches have the same length).  Fused -- ONE launch -- for RDHIP_FORWARD_AD (the reference's default diff method of a DiscretizedDynamics,
 * src/discretized_dynamics.jl:190) under every explicit rule, compiled-in models with device pointers whose J / Y start on 16-byte boundaries (custom
 * hipRTC models: every diff method, aligned or not -- their kernels are compiled on demand).  RDHIP_FINITE_DIFF, RDHIP_USER_DEFINED, unaligned outputs and
 * RDHIP_IMPLICIT_MIDPOINT have no batch-table kernel (round 6: those instantiations were half of the library's device code and nothing exercised them).
 * Such a call with device pointers and four or more runs of batches copies the batches' inputs into one stack owned by the context, runs ONE
 * rdhip_jacobian over it and copies the outputs back: three launches (five for the implicit rule) and a second copy of the outputs in device memory
 * (1 024 x 101 satellite knots under FiniteDifference: 166 us; not inside a stream capture, not in the launch-overlap mode).  Otherwise -- fewer runs,
 * host arrays, no room for the stack -- one rdhip_jacobian call per RUN of batches, where batches that follow each other in memory in every array
 * (the slices of one stack) form one run.  Same results either way.  A capturing stream with more than 32 ForwardAD batches issues groups of 32
 * (no allocation inside a capture).  rdhip_ctx_last_multi_launches tells which happened; RDHIP_VERBOSE=1 logs the fallback.
 * Batches must not overlap. */
int rdhip_jacobian_multi(rdhip_ctx* ctx, const rdhip_model* model, int rule, int diff, int dtype, int nbatch, const long long* N,
                         const void* const* Z, const double* const* t, const double* const* dt, void* const* J, void* const* Y, int memspace);
/* kernel launches the last rdhip_jacobian_multi call on this context issued: 1 = fused into one launch (the per-trajectory loops of
 * src/functionbase.jl:242 callers collapsed), 3 (5 for the implicit rule) = gathered into one stack, otherwise one (three) per run of batches */
int rdhip_ctx_last_multi_launches(rdhip_ctx* ctx, int* launches);

/* dynamics_error(model, z2, z1) = discrete_dynamics(model, z1) - state(z2) (src/discrete_dynamics.jl:134-135; in-place :146-156) for
 * the N-1 consecutive knot-point pairs (z1, z2) = (Z[k], Z[k+1]) of a packed trajectory: E[:, k] = f(z_k) - x_{k+1}, k = 0..N-2.
 * E: n x (N-1).  N < 2 is a no-op. */
int rdhip_dynamics_error(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long N,
                         const void* Z, const double* t, const double* dt, void* E, int memspace);

/* dynamics_error_jacobian!(sig, diff, model, J2, J1, y2, y1, z2, z1) (src/discrete_dynamics.jl:158-201; generated for ForwardAD /
 * FiniteDifference at src/jacobian_gen.jl:556-588,:714-745) for the same N-1 pairs and an explicit rule:
 *   J1[k] = d e_k / d z_k     = [A_k  B_k]   (the discrete dynamics Jacobian at z_k),
 *   J2[k] = d e_k / d z_{k+1} = [-I   0  ]   ("for explicit integration methods J2 should be equal to [-I(n) zeros(n,m)]", :176-177).
 * J1, J2: n x (n+m) x (N-1), either may be NULL (skipped); E: n x (N-1) or NULL receives the error itself (what the reference's
 * InPlace methods leave in y2). */
int rdhip_dynamics_error_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int rule, int diff, int dtype, long long N,
                                  const void* Z, const double* t, const double* dt, void* J2, void* J1, void* E, int memspace);

/* Error-state expansion of the discrete dynamics along a trajectory (what TrajectoryOptimization.jl / Altro.jl build from this path's
 * outputs per knot point, `error_expansion!`: D.A = G2' * A * G1, D.B = G2' * B):
 *     Ebar[k] = [ G(x_{k+1})' A_k G(x_k)  |  G(x_{k+1})' B_k ],   k = 0..N-2,
 * with [A_k B_k] = the ForwardAD Jacobian of discrete_dynamics(DiscretizedDynamics{rule}(model), z_k) (src/discretized_dynamics.jl:169)
 * and G = errstate_jacobian! (src/liestate.jl:262-298; identity for EuclideanState models, src/statevectortype.jl:149-155).
 * Fused into the Jacobian kernel for the compiled-in LieGroupModels (the n x (n+m) Jacobian never reaches memory); EuclideanState
 * models return [A_k B_k]; unaligned outputs of the other LieGroupModels and MRP / RodriguesParam states go through a workspace Jacobian stack
 * and a projection kernel (not capturable into a graph).
 * Ebar: e x (e+m) x (N-1), each slab column-major; Y: n x (N-1) or NULL receives f(z_k).  Device pointers must be 16-byte aligned. */
int rdhip_error_expansion(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long N,
                          const void* Z, const double* t, const double* dt, void* Ebar, void* Y, int memspace);

/* errstate_jacobian!(model, G, x) (v0.3: state_diff_jacobian!(G, model, z)): src/statevectortype.jl:119-122,149-155,
 * src/liestate.jl:262-298.  X: states, column k at X + k*ldx (ldx >= n; pass Z with ldx = n+m to read states out of
 * a packed trajectory).  G: n x e x N column-major.  clear != 0: G is fully overwritten (zeros elsewhere);
 * clear == 0 reproduces the reference exactly for RotationState models: only the identity diagonals and the
 * rotation blocks are written, every other entry of G is left untouched (src/liestate.jl:293-297).
 * EuclideanState models always write G = I (the reference clears in that branch).  Rotation blocks: QuatRotation 4 x 3 L(q)H (q normalised);
 * MRP 3 x 3 (1 - p.p) I + 2 (skew(p) + p p'); RodriguesParam 3 x 3 I + skew(g) + g g' (Rotations.jl grad-differential). */
int rdhip_errstate_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N,
                            const void* X, int ldx, int clear, void* G, int memspace);

/* state_diff(model, x, x0) with the default CayleyMap: src/statevectortype.jl:102-104,145-146, src/liestate.jl:210-260.
 * dX: e x N.  Served for QuatRotation, MRP and RodriguesParam LieStates (test/liestate.jl:33-68): the rotation part is the vector of
 * Rodrigues parameters of q0^-1 q whatever the parameterisation.
 * Tolerance of the parity gate: 1e-10 ABSOLUTE on rotation errors of magnitude <= 1e3 (tests/test_gpu_parity.py); the Cayley error v / w is unbounded
 * as the relative rotation approaches 180 degrees, where the gate is 1e-10 RELATIVE to |v / w|. */
int rdhip_state_diff(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N,
                     const void* X, int ldx, const void* X0, int ldx0, void* dX, int memspace);
/* state_diff(model, x, x0, errmap) / state_diff!(model, dx, x, x0, errmap): src/liestate.jl:26-32 (the RotationState methods take any Rotations.jl
 * ErrorMap), :216-217 (`Rotations.rotation_error(R(x_q...), R(x0_q...), errmap)` per rotation, = inv(errmap)(R0 \ R)), :239-250.  With
 * dq = [w; v] the quaternion of R0^-1 R (for QuatRotation states conj(q0) (x) q of the NORMALISED inputs, sign not canonicalised; MRP / RodriguesParam
 * states are converted first) the rotation part of dX is   [third party: Rotations.jl 1.x error maps, restated -- SURVEY App. A]
 *   RDHIP_ERRMAP_CAYLEY       v / w                                       (InvCayleyMap; what rdhip_state_diff computes)
 *   RDHIP_ERRMAP_EXPONENTIAL  atan(|v|, w) / |v| * v                      (InvExponentialMap = 0.5 logm(dq): half the rotation vector;
 *                             |v| <= 1e-6: (1 - |v|^2 / (3 w^2)) / w * v, Rotations.jl's series branch)
 *   RDHIP_ERRMAP_MRP          2 v / (1 + w)                               (InvMRPMap)
 *   RDHIP_ERRMAP_QUATVEC      sign(w) v                                   (InvQuatVecMap; sign(0) = 0 as in Julia)
 *   RDHIP_ERRMAP_IDENTITY     params(R0 \ R) in the state's own parameterisation: MRP states v / (1 + w), RodriguesParam states v / w.
 *                             QuatRotation states: RDHIP_ERR_UNSUPPORTED -- the reference throws there too (a RotationError holds three values,
 *                             IdentityMap of a quaternion returns four).
 * All maps agree to first order (dX ~ v, half the rotation angle).  Vector parts: x - x0. */
int rdhip_state_diff_map(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N,
                         const void* X, int ldx, const void* X0, int ldx0, int errmap, void* dX, int memspace);

/* nabla-errstate_jacobian!(model, nablaG, x, xbar) (v0.3: nabla^2 differential!): src/statevectortype.jl:137-147,
 * src/liestate.jl:300-320.  nablaG: e x e x N.  clear as for rdhip_errstate_jacobian.  Rotation blocks grad^2-differential(R(x), xbar_rot): QuatRotation
 * -(q . xbar_q) I (q normalised); MRP / RodriguesParam: the second derivative of delta -> params(p * R(delta)) at 0 contracted with xbar_rot --
 * MRP (b.p)(8 p p' - 2 (1 + p.p) I) + 2 (p c' + c p'), c = (1 - p.p) b - 2 p x b; RodriguesParam 2 (b.g) g g' + g c' + c g', c = b - g x b. */
int rdhip_nabla_errstate_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N,
                                  const void* X, int ldx, const void* Xbar, int ldxbar, int clear, void* nablaG, int memspace);

/* rollout!(sig, DiscretizedDynamics{rule}(model), Z, x0) for M independent trajectories of K knot points:
 * src/trajectories.jl:436-441, src/discrete_dynamics.jl:217-235.  Z: (n+m) x K x M (trajectory j at
 * Z + j*K*(n+m)), states of knots 2..K are overwritten in place, controls are read; t, dt: K x M.
 * X0: n x M initial states or NULL (keep state(Z[1])).  Sequential in k, parallel over trajectories.  rule may be
 * RDHIP_IMPLICIT_MIDPOINT (a Newton solve per step). */
int rdhip_rollout(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long K, long long M,
                  void* Z, const double* t, const double* dt, const void* X0, int memspace);

/* wrenches(model, x, u, t) = [forces; moments] (src/rigidbody.jl:244-248; the first thing a RigidBody's dynamics evaluate, :217) for every knot
 * point: W 6 x N; and its Jacobian with respect to z = [x; u], F 6 x (n+m) x N -- what the default wrench_jacobian!(F, model, z) of
 * src/rigidbody.jl:315-320 (ForwardDiff.jacobian of `wrenches`) is written to return; test/rigidbody_test.jl:33-46 holds the closed form for the
 * satellite wrench (dF/dq = grad-rotate(q, u[1:3]), dF/du = RotMatrix(q), dM/du = I).  The compiled-in RigidBody models (satellite, quadrotor;
 * QuatRotation, MRP and RodriguesParam attitudes); `t` may be NULL (their wrenches do not depend on time).  rdhip_model_wrench_sparsity:
 * wrench_sparsity(model) (src/rigidbody.jl:322-347), js[i + 2 j] for i = F, M and j = r, q, v, w, u (column-major 2 x 5 like the SMatrix). */
int rdhip_wrenches(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N, const void* Z, const double* t, void* W, int memspace);
int rdhip_wrench_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int dtype, long long N, const void* Z, const double* t, void* F, int memspace);
int rdhip_model_wrench_sparsity(const rdhip_model* model, int js[10]);

/* layouts of a batch of M trajectories x K knot points (Z, t, dt and the Jacobians share the knot indexing) */
#define RDHIP_TRAJECTORY_MAJOR 0   /* knot k of trajectory j at index j*K + k: M consecutive trajectories, what rdhip_rollout takes */
#define RDHIP_TIME_MAJOR 1         /* knot k of trajectory j at index k*M + j: knot k of all trajectories contiguous */

/* rollout! on SPLIT arrays -- states and controls in arrays of their own (what `states(Z)` / `controls(Z)` give, src/trajectories.jl; SURVEY 8b's
 * rdhip_rollout(x0, U, ... -> X)): X n x K x M receives every simulated state (knot 0 = X0[:, j], or what X holds there when X0 is NULL), U m x (K-1) x M is
 * read; t, dt: K x M as for rdhip_rollout_layout (dt of a trajectory's last knot is not used).  layout RDHIP_TRAJECTORY_MAJOR: state k of trajectory j at
 * X + (j K + k) n, control k at U + (j (K-1) + k) m; RDHIP_TIME_MAJOR: X + (k M + j) n, U + (k M + j) m.  Same arithmetic as rdhip_rollout.
 * Why it exists: a packed stack interleaves x and u, and an in-place rollout either writes the controls back with the states (rdhip_rollout's streaming
 * kernel: 1.93 x the algorithmic bytes of a 32 768 x 32 fp64 satellite batch) or stores partial lines; here U, t, dt are read once and X is written once
 * (168 instead of 320 bytes per satellite knot point; time-major 32 768 x 32: 42 us against 66 us for the packed stack).  Explicit rules, compiled-in
 * models; no alignment requirement; the same bits as rdhip_rollout's one-lane kernel.  src/trajectories.jl:436-441. */
int rdhip_rollout_split(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long K, long long M, int layout,
                        void* X, const void* U, const double* t, const double* dt, const void* X0, int memspace);
/* rdhip_rollout for either layout of the stack (rdhip_rollout = RDHIP_TRAJECTORY_MAJOR, the reference's "vector of trajectories").  Big batches
 * of models whose knot point is at least 128 bytes (the fp64 rigid bodies) -- from two 64-trajectory groups per compute unit on, 32 768 trajectories on an MI355X -- are streamed through LDS in chunks of knot points as
 * whole 16-byte units (csrc/rollout_stream.hpp; every run of the stack must start on a 16-byte boundary: Z 16-byte aligned and K (n+m) sizeof(T)
 * [trajectory-major] or M (n+m) sizeof(T) [time-major] a multiple of 16; other shapes take the one-lane-per-trajectory kernel).  That path
 * rewrites the CONTROLS of Z with the values it read (whole-line stores): Z must not be written concurrently by other work, which an in-place
 * rollout! forbids anyway.  Implicit rules take RDHIP_TRAJECTORY_MAJOR data only (RDHIP_ERR_UNSUPPORTED otherwise).  src/trajectories.jl:436-441. */
int rdhip_rollout_layout(rdhip_ctx* ctx, const rdhip_model* model, int rule, int dtype, long long K, long long M, int layout,
                         void* Z, const double* t, const double* dt, const void* X0, int memspace);

/* rollout!(sig, dmodel, Z, x0) followed by jacobian!(sig, diff, dmodel, J[k], y, Z[k]) at every knot point of the simulated trajectories
 * (the forward pass + linearisation of a shooting method; rollout!: src/trajectories.jl:436-441, jacobian!: src/functionbase.jl:242-257).
 * Same result as rdhip_rollout + rdhip_jacobian on the K*M knot points (the states bit for bit: rdhip_rollout runs the same kernel; the Jacobians to the
 * last bit or two: the same column code compiled into another kernel).  Z: (n+m) x (K*M), J: n x (n+m) x (K*M), t, dt: K*M, X0: n x M or NULL.
 * ONE launch in which the two overlap -- rollout workgroups (one per 64 trajectories) announce every finished chunk of 4 steps in a progress
 * word, Jacobian workgroups on the rest of the chip consume the chunks a few steps behind, so the Jacobian store stream runs under the rollout's
 * latency chain -- for the RigidBody{QuatRotation} models (satellite, quadrotor), explicit rules, RDHIP_FORWARD_AD, RDHIP_TIME_MAJOR data with
 * 16-byte aligned J and M * n * (n+m) * sizeof(T) a multiple of 16 (M even in fp64, a multiple of 4 in fp32) and at most half as many 64-trajectory
 * groups as the device has compute units; every other combination runs rollout and Jacobians as launches of their own -- back to back, or, for
 * RDHIP_TIME_MAJOR stacks of 262 144 knot points and more with at least 64 knots per trajectory and at most half as many 64-trajectory groups as the
 * device has compute units -- a rollout that is a latency chain, not one that fills the chip -- (small models, custom models, the models of a
 * mixed-model call that did not get the overlapped launch), in four pieces of consecutive knots with the Jacobians of a finished piece on an
 * internal stream beside the rollout of the next (RDHIP_TUNE=rolljac_pipe_min_knots=N in the environment moves the limit, 0 = never; results are those of
 * the two plain launches).  rdhip_ctx_last_rollout_jacobian_launches says which (1 = overlapped in one launch, 2 = back to back, more = in pieces).  RDHIP_TRAJECTORY_MAJOR stacks take the overlapped launch as well (tiles gathered
 * from 4 trajectories x 16 knots): fp32 stacks of any size (4 096 x 256 satellite knots: 231 us against 288 for the two launches; time-major: 199), fp64
 * stacks while they are latency-sized -- from 262 144 knot points on those run the two launches, which is faster there (4 096 x 101 fp64 knots: 204 vs 250 us;
 * RDHIP_TUNE=fused_tj_max_knots=N in the environment moves the limit).  An fp32 launch of either layout under RK3 / RK4 that is bound by its Jacobian
 * workgroups runs two workgroups per compute unit (RDHIP_TUNE=fused_two_per_cu=0: never).  Implicit rules take RDHIP_TRAJECTORY_MAJOR data only.
 * The Jacobian workgroups of the overlapped launch hold their compute unit while they wait for the rollout workgroups of the SAME launch; the
 * library therefore keeps at most one such launch in flight per device and process (a second one, from any context, is ordered behind the first by
 * an event edge).  Outside its reach: launches replayed from captured graphs side by side, and other processes sharing the GPU -- do not run two of
 * those concurrently on one device. */
int rdhip_rollout_jacobian(rdhip_ctx* ctx, const rdhip_model* model, int rule, int diff, int dtype, long long K, long long M, int layout,
                           void* Z, const double* t, const double* dt, const void* X0, void* J, int memspace);
int rdhip_ctx_last_rollout_jacobian_launches(rdhip_ctx* ctx, int* launches);
/* rdhip_rollout_jacobian for up to four MODELS in one call -- the mixed-model Monte-Carlo batch (even trajectories one model, odd ones another;
 * BASELINE config 5).  ONE kernel launch on the context's stream, no internal stream, for the pair that configuration names -- a RigidBody{QuatRotation}
 * satellite (world-frame velocities, diagonal inertia) and the Cartpole under the same rule (RK3 or RK4), RDHIP_FORWARD_AD, device pointers, either
 * layout (trajectory-major fp64 stacks: below 262 144 knot points), both models' stacks of the same layout: both rollouts and every Jacobian in one grid
 * (csrc/mixed.hpp; 4 096 + 4 096 trajectories x 256 knots in fp32: 229 us time-major, 280 us trajectory-major).  RDHIP_TUNE=mixed=0: never.  Every other
 * combination, as described from here on:
 * `for p in models: rollout!(sig, dmodel[p], Z[p], x0[p]); jacobian! along it`, src/trajectories.jl:436-441 + src/functionbase.jl:242-257): model p has
 * its own stack (K[p] knots x M[p] trajectories, arrays Z[p], t[p], dt[p], X0[p] or NULL, J[p]; rule[p] its quadrature rule; X0 itself may be NULL).
 * With device pointers every model runs on its own internal stream between one fork and one join on the context's stream: the rollouts' latency
 * chains (a handful of compute units each) run side by side and the Jacobian workgroups share the rest of the chip (ONE model of the call -- the one
 * with the most Jacobian bytes among those that have it -- takes the overlapped single launch of rdhip_rollout_jacobian, the others their two
 * launches: see there); the simulated states are those of the single-model calls bit for bit, the Jacobians agree with them to rounding (the one-launch and the two-launch form of a model compile the same column code separately, and which model of a call gets the one-launch form depends on the other models), and work queued on the context afterwards sees all of them.  Host arrays and capturing streams run the models one after
 * the other.  (HIP maps a process's streams onto a few hardware queues -- four by default, GPU_MAX_HW_QUEUES -- and the call uses up to two internal
 * streams per additional model: a process that HOLDS other streams, busy or not, can find two models of a call on one queue, i.e. running one after
 * the other: measured 281 instead of 160 us for the C5 share beside four busy streams, and 290 instead of 260 us for the whole of config 5 beside one
 * idle context that owned its stream and had launch overlap on.  Destroy contexts that are done, or raise GPU_MAX_HW_QUEUES.)
 * On a context that owns its stream with launch overlap on (rdhip_ctx_set_overlap) a call with TWO models runs its second model on the context's internal
 * stream and does not wait for it at its end -- the next operation of any other kind does (ORDERING RULE above; rdhip_ctx_fence before handing the stream
 * to other code) -- and a call that REPEATS the previous one argument for argument (a solver's loop) does not fork either: each stream re-issues its own
 * kernels on its own arrays, in order.  The context's stream then carries the long model's launches back to back, without the join wait, the fork record
 * and the launch behind them (three packets, ~18 us between the fused launches of consecutive calls): BASELINE config 5 on one GPU 244-252 us per call
 * instead of 257-262, its 8-way share 132-135 instead of 151. */
int rdhip_rollout_jacobian_multi(rdhip_ctx* ctx, int nmodels, const rdhip_model* const* models, const int* rule, int diff, int dtype,
                                 const long long* K, const long long* M, int layout, void* const* Z, const double* const* t, const double* const* dt,
                                 const void* const* X0, void* const* J, int memspace);

/* ---- several GPUs in ONE process (hosts without torch.distributed / RCCL, e.g. the Julia wrapper) ----
 * The multi-GPU layout of this library is one context per device and contiguous blocks of knot points per device (knot points are
 * independent: no data-path collective).  rdhip_shard_range: block i of nshards = [begin, begin + count), ceil(N / nshards) knot points
 * each, the last blocks short or empty.
 * rdhip_jacobian_sharded: rdhip_jacobian on HOST arrays of the whole batch, block i on ctxs[i], all blocks concurrently (one host
 * thread per context inside the call); the stacked result is complete in host memory when the call returns ("gather to the host").
 * Several contexts may sit on the same device.  Returns the first non-zero status; its message is in that context and in
 * rdhip_last_error(NULL).
 * rdhip_memcpy_peer: device-to-device copy between two contexts' devices over xGMI (to stack device-resident shards on one GPU):
 * ordered after the work already queued on src_ctx's stream, executed on dst_ctx's stream, asynchronous. */
int rdhip_shard_range(long long N, int nshards, int i, long long* begin, long long* count);
int rdhip_jacobian_sharded(rdhip_ctx* const* ctxs, int nctx, const rdhip_model* model, int rule, int diff, int dtype, long long N,
                           const void* Z, const double* t, const double* dt, void* J, void* Y);
int rdhip_memcpy_peer(rdhip_ctx* dst_ctx, void* dst, rdhip_ctx* src_ctx, const void* src, unsigned long long bytes);

/* ---- the one exchange: stacking device-resident result shards over RCCL / xGMI -------------------------------------------------------
 * Evaluation never communicates (knot points are independent: rank r of P owns block rdhip_shard_range(N, P, r)); a caller that needs the
 * STACKED Jacobians in one place pays this step.  A communicator has one rank per GPU and is driven either by ONE process
 * (rdhip_comm_create: ranks 0..n-1 = the given contexts, ncclCommInitAll; the layout of a Julia host) or by one process per GPU
 * (rdhip_comm_create_rank: rank 0 draws rdhip_comm_unique_id, the application hands the 128 bytes to the other processes -- MPI, a file,
 * torch.distributed's store -- and every process joins with its own context; ncclCommInitRank).  librccl is dlopen()ed on first use
 * (RDHIP_RCCL_LIB overrides the name); a 1-rank communicator never loads it.
 *
 * rdhip_gather(comm, mode, root, shard, bytes, out): `bytes[r]`, r = 0..nranks-1, is the shard size of EVERY rank (all processes pass the same
 * table); `shard[l]` / `out[l]`, l = 0..nlocal-1, are the device buffers of the ranks this process drives (rdhip_comm_rank maps l to its rank).
 * The stack is the concatenation of the shards in rank order.  RDHIP_GATHER_ROOT: out[l] of rank `root` receives it (other entries are ignored) --
 * grouped ncclSend / ncclRecv, every shard on its own xGMI link into the root (RCCL has no gather primitive; a ring would serialise on single
 * links); RDHIP_GATHER_ALL: every rank's out[l] receives it (ncclAllGather for equal shards, grouped send / recv otherwise); RDHIP_GATHER_NONE:
 * no-op (results stay sharded for a sharded consumer).  Asynchronous: ordered on each rank's context stream behind the kernels that produce
 * the shard; synchronise the receiving context(s) before reading.  A shard that already sits at its place inside out[l] is not copied.
 * rdhip_jacobian_sharded_gather / rdhip_errstate_jacobian_sharded_gather = the per-rank device-memory call on each local rank's block
 * (Z[l], t[l], dt[l], J[l], Y[l] / X[l], G[l] are that BLOCK's arrays, N is the TOTAL knot count) followed by rdhip_gather of J / G. */
#define RDHIP_COMM_ID_BYTES 128
typedef enum { RDHIP_GATHER_NONE = 0, RDHIP_GATHER_ROOT = 1, RDHIP_GATHER_ALL = 2 } rdhip_gather_mode;
typedef struct rdhip_comm rdhip_comm;
int rdhip_comm_create(rdhip_ctx* const* ctxs, int n, rdhip_comm** out);
int rdhip_comm_unique_id(char id[RDHIP_COMM_ID_BYTES]);
int rdhip_comm_create_rank(rdhip_ctx* ctx, int nranks, int rank, const char id[RDHIP_COMM_ID_BYTES], rdhip_comm** out);
int rdhip_comm_destroy(rdhip_comm* comm);
int rdhip_comm_size(const rdhip_comm* comm, int* nranks, int* nlocal);
int rdhip_comm_rank(const rdhip_comm* comm, int local, int* rank, rdhip_ctx** ctx);
int rdhip_gather(rdhip_comm* comm, int mode, int root, const void* const* shard, const unsigned long long* bytes, void* const* out);
int rdhip_jacobian_sharded_gather(rdhip_comm* comm, const rdhip_model* model, int rule, int diff, int dtype, long long N,
                                  const void* const* Z, const double* const* t, const double* const* dt, void* const* J, void* const* Y,
                                  int gather, int root, void* const* Jstack);
int rdhip_errstate_jacobian_sharded_gather(rdhip_comm* comm, const rdhip_model* model, int dtype, long long N, const void* const* X, int ldx, int clear,
                                           void* const* G, int gather, int root, void* const* Gstack);

#ifdef __cplusplus
}
#endif
#endif /* RDHIP_H */
