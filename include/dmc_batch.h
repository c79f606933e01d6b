/* dmc_batch.h -- C-ABI of the MI355X batched physics step (libdmc_hip.so).
 *
 * The reference has no plugin registry for its physics backend: the seam is the
 * set of pybind11 functions that take (MjModel, MjData) and are called from
 * dm_control/mujoco/engine.py and wrapper/core.py.  Each entry
 * point below names the reference call it replaces for a whole batch of
 * independent environments.  Plain pointers and sizes only; every function
 * returns 0 on success or a negative error code and never throws/aborts
 * (dmc_last_error() returns a thread-local message).
 *
 * Memory model: one dmc_batch owns B environments on ONE GPU.  Every mjData
 * array is stored structure-of-arrays across the batch: element k of env e of
 * field F lives at F_dev[k * B + e] (dtype = batch precision, ints int32;
 * "time" is always float64).
 */
#ifndef DMC_BATCH_H_
#define DMC_BATCH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmc_model dmc_model;
typedef struct dmc_batch dmc_batch;

const char* dmc_last_error(void);

/* Replaces mujoco.MjModel.from_xml_string as consumed by
 * wrapper.MjModel.from_xml_string (dm_control/mujoco/wrapper/core.py:180-182,289):
 * takes the compiled constant tables (include/dmc_model_layout.h) produced by
 * dm_control_amd/mjcf_compiler.py. */
int dmc_model_create(const int32_t* ints, int n_ints, const double* reals, int n_reals, dmc_model** out);
void dmc_model_destroy(dmc_model* m);

/* Replaces mujoco.MjData(model) (wrapper/core.py:475), B times.
 * precision: 32 or 64.  nconmax/njmax: per-env contact / constraint-row caps
 * (0 = automatic); exceeding them raises mjWARN_CONTACTFULL / mjWARN_CNSTRFULL
 * counters like MuJoCo's own caps.  lanes_per_env: 64, 32 or 16 (0 = automatic: 32 for nv <= 12, else 64). */
int dmc_batch_create(const dmc_model* m, int batch_size, int device_id, int precision,
                     int nconmax, int njmax, int lanes_per_env, dmc_batch** out);
/* The same with the capacity knobs as a list: caps = {nconmax, njmax, lanes_per_env, njcon, jlevel + 1} (missing or 0 =
 * automatic; jlevel: how much of the per-env scratch leaves LDS for global memory -- 0 keeps the contact rows and the kept
 * factor of M in LDS, the automatic choice for a batch of at most one environment per CU).
 * njcon: contact rows with a stored Jacobian.  The default lets every contact slot use the most rows a contact of
 * the model can have (nconmax x that); a smaller pool keeps LDS for a second / third resident environment and
 * raises DMC_WARN_CNSTRFULL (the contact is dropped, as MuJoCo does when njmax is hit) when the live contacts of one
 * environment need more. */
int dmc_batch_create_caps(const dmc_model* m, int batch_size, int device_id, int precision,
                          const int* caps, int ncaps, dmc_batch** out);
void dmc_batch_destroy(dmc_batch* b);

/* Model-specialised kernels on demand.  The library carries specialised instantiations of the step kernel for a fixed
 * list of assets (the LDS layout as a compile-time constant: offsets become immediates, the Cholesky factor lives in
 * registers); any other model -- a composer environment recompiled with another body count (composer/environment.py:
 * 377-383), a user's MJCF (engine.py:446-476 from_xml_string) -- runs the generic kernel, 2 - 2.7 x slower.
 * dm_control_amd/specialise.py compiles csrc/step_kernel_spec.hip for ONE model (hipcc, cached on disk by the hash of
 * the model blob, the caps, the precision and the sources) into a shared object; this call loads it next to the batch,
 * checks that its layout IS the batch's (byte for byte) and that it was built from the same sources, and routes the
 * batch's launches through it.  The object stays loaded for the life of the process. */
int dmc_batch_attach_specialised(dmc_batch* b, const char* so_path);

/* control.Environment.step in one launch (dm_control/rl/control.py:99-127: before_step, physics.step, reward, observation,
 * termination).  dm_control_amd/suite/fused_env.py compiles a suite task's own get_observation / get_reward /
 * termination code into a device function; a specialised kernel built with that function (`DMC_TASK_HEADER`) runs it
 * for every environment at the end of its step launch -- observation (B, nobs), reward, discount, done / first flags
 * and, where the episode ended, the start state of the next one -- so that an environment step costs no launch beside
 * the physics.  set_task_args uploads the function's argument block (device pointers of the field / output / pool
 * arrays; its size is checked against the attached kernel's); enable_task says whether the step launches issued from
 * now on run it (a host-side flag read when a launch is enqueued, so a HIP graph keeps what it was captured with). */
int dmc_batch_set_task_args(dmc_batch* b, const void* args, int nbytes);
int dmc_batch_enable_task(dmc_batch* b, int on);

/* Replaces Physics.step(nstep) = mj_step2; mj_step(nstep-1); mj_step1 when
 * legacy_step != 0 (dm_control/mujoco/engine.py:147-162) or mj_step(nstep)
 * otherwise (engine.py:176), for every env, in ONE kernel launch.
 * legacy_step == 2: the legacy step followed, in the same launch, by the rest of mj_forward at the new state (the
 * acceleration stage with its sensors, no integration) -- what a composer agent observes: the reference's observation
 * update forwards the dirty physics before the first observable is read (dm_control/mjcf/physics.py:341-342 after
 * composer/environment.py:412-465), so touch / torque / accelerometer values are those of the NEW state.
 * hip_stream: hipStream_t to launch on (NULL = default stream).  Asynchronous. */
int dmc_batch_step(dmc_batch* b, int nstep, int legacy_step, void* hip_stream);

/* Substep probe: the world position of geom `geom_id` after EVERY physics step of a legacy step launch, written to the
 * caller's device array `out_dev`, (capacity, 3, B) reals of the batch precision (slot k = after k + 1 steps; an
 * environment the launch override turns into mj_forward reports its one state in every slot).  Replaces an
 * `after_substep` hook that only reads a position -- composer/environment.py:412-465 runs the hooks between the
 * mj_step calls; entities/props/position_detector.py:200-260 is such a hook (the soccer goal / out-of-court
 * detectors with retain_substep_detections) -- so that the control step stays ONE launch.  out_dev NULL: off. */
int dmc_batch_set_step_probe(dmc_batch* b, int geom_id, void* out_dev, int capacity);

/* Random-action / open-loop rollouts without a launch per step: plays `nsteps`
 * env-steps of `n_sub_steps` physics steps each (the loop of
 * control.Environment.step, dm_control/rl/control.py:99-127, with the task's
 * before_step = set_control) in ONE launch.  ctrl_seq: device (nsteps, nu, B) or
 * NULL (keep ctrl); *_seq outputs: device (nsteps, rows, B) or NULL.  Each
 * env-step has legacy_step semantics; the closing mj_step1 of step t doubles as
 * the opening position/velocity stage of step t+1, so nothing is computed twice. */
int dmc_batch_rollout(dmc_batch* b, int nsteps, int n_sub_steps, const void* ctrl_seq, void* qpos_seq,
                      void* qvel_seq, void* sensordata_seq, void* hip_stream);

/* Replace mujoco.mj_step1 / mujoco.mj_step2 called on their own (engine.py:156-162; composer hooks that read or
 * write between the two halves of a step).  dmc_batch_step1: position / velocity stage at the current state
 * (derived arrays and position / velocity sensors written, state unchanged); dmc_batch_step2: actuation,
 * acceleration, constraint solve, acceleration-stage sensors, mj_checkAcc, Euler integration.  The stage travels
 * from one call to the other through the per-environment stash (see dmc_batch_invalidate); if the state was
 * edited in between, dmc_batch_step2 recomputes it first.  dmc_batch_step2 rejects RK4 models. */
int dmc_batch_step1(dmc_batch* b, void* hip_stream);
int dmc_batch_step2(dmc_batch* b, void* hip_stream);

/* Replaces mujoco.mj_forward (engine.py:343); disable_actuation != 0 mirrors the
 * mjDSBL_ACTUATION context used by Physics.reset/after_reset (engine.py:326-333). */
int dmc_batch_forward(dmc_batch* b, int disable_actuation, void* hip_stream);

/* Replaces mujoco.mj_resetData / mj_resetDataKeyframe (engine.py:318,323) for the
 * envs whose mask byte is non-zero (mask == NULL: all).  keyframe < 0: qpos0, zero velocities / controls /
 * activations / time, mocap poses from the model; keyframe >= 0: key_qpos, key_qvel, key_ctrl, key_time, key_act,
 * key_mpos, key_mquat of that keyframe. */
int dmc_batch_reset(dmc_batch* b, const uint8_t* env_mask, int keyframe);

/* Field access by mjData name ("qpos", "qvel", "act", "ctrl", "qacc_warmstart", "time",
 * "qfrc_applied", "xfrc_applied" (6 per body: Cartesian force, torque at the body COM; read by the kernel once it has been
 * written, bound or exposed through dmc_batch_device_ptr), "sensordata", "xpos", "xquat", "xmat", "xipos", "geom_xpos",
 * "geom_xmat", "site_xpos", "site_xmat", "subtree_com", "qacc", "actuator_force",
 * "qfrc_actuator", "qfrc_bias", "qfrc_constraint", "contact_dist", "contact_pos",
 * "contact_frame", "contact_force" (mj_contactForce per contact, valid after dmc_batch_forward;
 * wrapper/core.py:527-552), "cvel" (for mj_objectVelocity, wrapper/core.py:500-525); int32: "ncon", "nefc", "solver_iter", "warning",
 * "contact_geom1", "contact_geom2", "env_mode").  "env_mode" (one int per environment, default 0) overrides what a
 * dmc_batch_step / step1 / step2 launch does to that environment: 0 = step, 1 = mj_forward with actuation
 * disabled instead (an environment re-initialised under the reference's reset_context, rl/control.py:232-253,
 * while the rest of the batch steps), 2 = leave it untouched (also honoured by dmc_batch_forward: refresh only
 * the environments that were reset).  Replaces the numpy views MjData exposes
 * (wrapper/core.py:438-447).  Host buffers are env-major: (B, rows), float64 /
 * int32 regardless of the batch precision.  Synchronous. */
int dmc_batch_field_rows(const dmc_batch* b, const char* name, int* rows, int* is_int);
int dmc_batch_get(dmc_batch* b, const char* name, double* dst);
int dmc_batch_set(dmc_batch* b, const char* name, const double* src);
int dmc_batch_get_int(dmc_batch* b, const char* name, int32_t* dst);
int dmc_batch_set_int(dmc_batch* b, const char* name, const int32_t* src);

/* The same reads / writes of the MjData views (wrapper/core.py:438-447; engine.py:139-145 set_control writes data.ctrl),
 * asynchronous on `hip_stream` and sized for a host loop that keeps actions and observations in host memory
 * (rl/control.py:99-127 with a numpy policy): the host converts contiguously into pinned staging (host_bits = 32 on an
 * fp32 batch: no conversion at all, fp32 on the wire), copies are hipMemcpyAsync, the (B, rows) <-> (rows, B)
 * transposition is a device kernel, and a get of several fields is one device-to-host copy and one wait.
 *   dmc_batch_set_async: `src` (B, rows) of host_bits-wide floats is consumed before the call returns; the write is
 *     ordered on the stream with the launches (and bumps the stash epoch there, like dmc_batch_invalidate_async).
 *   dmc_batch_get_async: enqueues the read of `n` real fields (as of this point of the stream); one get at a time.
 *   dmc_batch_get_wait: waits for it and writes field i, (B, rows_i) of host_bits-wide floats, to dsts[i] (NULL: skip);
 *     int fields ("warning", "ncon", ...) may be part of a get and arrive as int32. */
int dmc_batch_set_async(dmc_batch* b, const char* name, const void* src, int host_bits, void* hip_stream);
int dmc_batch_get_async(dmc_batch* b, int n, const char* const* names, void* hip_stream);
int dmc_batch_get_wait(dmc_batch* b, int n, void* const* dsts, int host_bits);
/* Field i of the last completed get where it lies in the pinned staging, (B, rows_i) in the batch's own precision
 * ("time": float64) -- valid until the next dmc_batch_get_async: the zero-copy form of dmc_batch_get_wait (pass NULL
 * destinations to it, then read here). */
const void* dmc_batch_get_staged(dmc_batch* b, int i);

/* Zero-copy access: the SoA device array of a field, (rows, B) in batch
 * precision; and rebinding a field to caller-owned device memory of that shape
 * (e.g. a torch tensor) so producers/consumers on the GPU never round-trip. */
void* dmc_batch_device_ptr(dmc_batch* b, const char* name);
int dmc_batch_bind(dmc_batch* b, const char* name, void* device_ptr);

/* Which derived arrays a step/forward writes back to HBM (bit mask, see
 * dm_control_amd/csrc/step_core.h OUT_*; default: all). */
int dmc_batch_set_output_mask(dmc_batch* b, int mask);

/* Model options that tasks mutate between steps (wrapper/core.py:389-426):
 * "disableflags", "iterations", "ls_iterations" / "timestep", "tolerance",
 * "ls_tolerance", "gravity_x|y|z".
 * "islands" (int): how mj_fwdConstraint's constraint islands are honoured (`flag island`, a disable flag that defaults
 * to enable: dm_control/mjcf/schema.xml:102).  1 = one solve per island, as MuJoCo does when the flag is on, no noslip
 * pass is configured and the solver is CG / Newton; 0 = one joint solve (the same minimiser: the cross blocks are exact
 * zeros; measured 1e-14 .. 6e-9 apart with Newton, 5e-6 with CG over a few hundred steps); -1 (default) = by precision:
 * fp64 batches solve per island (they track the CPU reference) and so do CG models at any precision (round 6: CG's
 * joint answer is the one that differs measurably); fp32 Newton batches solve jointly (throughput). */
int dmc_batch_set_opt_int(dmc_batch* b, const char* name, int value);
int dmc_batch_set_opt_real(dmc_batch* b, const char* name, double value);

/* Model constants that suite tasks rewrite between episodes through
 * physics.named.model (e.g. suite/finger.py:139 dof_damping, :169-170 site_pos / site_size):
 * "dof_damping", "jnt_stiffness", "jnt_range", "jnt_margin", "qpos_spring", "site_pos",
 * "site_quat", "site_size", "actuator_ctrlrange", "actuator_forcerange", "wrap_prm"
 * (suite/point_mass.py:113-114), "body_pos", "body_quat" (suite/manipulator.py:201-208), "geom_pos",
 * "geom_quat", "geom_size" (suite/reacher.py:88-94, suite/fish.py:150-154; as with a write to mjModel, nothing
 * derived at compile time -- inertias, geom_rbound, contact-pair mixing -- follows).  `values` is the
 * whole mjModel array (count elements).  Shared by every env of the batch.  Synchronous. */
int dmc_batch_set_model_real(dmc_batch* b, const char* name, const double* values, int count);

/* Per-environment model deltas.  The reference randomises scenery per episode by editing the MJCF and recompiling
 * (composer initialize_episode_mjcf; soccer RandomizedPitch, locomotion/soccer/pitch.py:612-690).  A batch shares ONE
 * compiled model; dmc_batch_set_env_geoms declares the world-fixed geoms (children of the worldbody) whose pose and
 * size differ between environments.  It creates the field "env_geom": 16 rows per declared geom, in declaration
 * order -- pos(3), rotation matrix (9, row-major), size(3), bounding radius(1) -- initialised from the model for
 * every environment, readable / writable / bindable like any other field (dmc_batch_set "env_geom": (B, 16 n)).
 * dmc_env_geom_pack fills the 16 values of one geom from (pos, quat, size).  Contact parameters, masses and
 * everything else stay shared. */
int dmc_batch_set_env_geoms(dmc_batch* b, int n, const int* geom_ids);
int dmc_env_geom_pack(int geom_type, const double* pos, const double* quat, const double* size, double* out16);

int dmc_batch_sync(dmc_batch* b);

/* Between the mj_step1 that ends one legacy Physics.step() and the mj_step2 that begins the next
 * (dm_control/mujoco/engine.py:147-162) the reference keeps the position / velocity stage in mjData.
 * The batch keeps it in a per-environment stash in HBM (option "stash": dmc_batch_set_opt_int; off by
 * default: on MI355X recomputing the outputs of that stage is cheaper than the stash traffic), so that a legacy
 * step launch does not recompute it.  Every entry point of this
 * library that edits state, model or options invalidates the stash itself.  A caller that writes qpos / qvel /
 * act through memory it bound with dmc_batch_bind must call this afterwards (the reference's equivalent:
 * derived quantities are stale until mj_forward is run).
 * Independently of that option every batch keeps a small KINEMATIC stash (poses, COM frame, velocities -- what
 * mj_kinematics / mj_comPos / mj_comVel derive from qpos and qvel) between legacy steps; it is stored with the
 * (qpos, qvel) it was computed at and reused only when the next launch finds exactly that state, so it needs no
 * invalidation by the caller (DMC_NO_KSTASH=1 in the environment disables it). */
int dmc_batch_invalidate(dmc_batch* b);
/* The same, ordered on `hip_stream` instead of waiting for the device: the stash epoch lives in device memory and is
 * bumped by a one-thread kernel, so an invalidation issued while a HIP graph is being captured (a composer hook that
 * edits qpos through a bound tensor inside a captured control step) is part of the graph and replays with it. */
int dmc_batch_invalidate_async(dmc_batch* b, void* hip_stream);

/* Per-episode joint randomisation on the device: what suite tasks do on the host in
 * initialize_episode -- suite/utils/randomizers.py:35-88 randomize_limited_and_rotational_joints, the limited-joint
 * draw of suite/cheetah.py:66-69, the random orientation of suite/quadruped.py:243-246 -- for every environment whose
 * entry of `d_env_mask` (device, B ints; NULL: all) is non-zero, written straight into the qpos rows.  Draws come
 * from Philox4x32-10 keyed by (seed ^ env, seed >> 32) with counter (d_draw[env], joint, block, 0); d_draw (device,
 * B ints, caller-owned) is incremented for the environments that were drawn, so an (environment, draw) pair never
 * repeats and a rejection loop (suite/humanoid.py:160-165) simply calls again with the mask of those still in contact.
 * flags: DMC_RAND_LIMITED bounded hinges / sliders ~ U(range) and limited ball joints (axis ~ N(0, I) normalised,
 * angle ~ U(0, range)); DMC_RAND_UNLIMITED_HINGE ~ U(-pi, pi); DMC_RAND_QUATERNION unlimited ball joints ~ uniform on
 * the 3-sphere and free-joint quaternions ~ normalised U(0, 1)^4 (as the reference draws them), or uniform on the
 * sphere with DMC_RAND_FREE_NORMAL.  Free-joint translations are left alone.  Ordered on `hip_stream`; the stashes
 * are invalidated on the same stream. */
enum { DMC_RAND_LIMITED = 1, DMC_RAND_UNLIMITED_HINGE = 2, DMC_RAND_QUATERNION = 4, DMC_RAND_FREE_NORMAL = 8,
       DMC_RAND_ALL = 7 };
int dmc_batch_randomize_joints(dmc_batch* b, uint64_t seed, int32_t* d_draw, const int32_t* d_env_mask, int flags,
                               void* hip_stream);

/* Observation gather table: what composer's observation.Updater does per control step for MJCFFeature observables
 * (composer/observation/updater.py:285-295, composer/observation/observable/mjcf.py:43), for the whole batch in one
 * launch.  Row k of the table names element `rows[k]` of the mjData field `field_names[k]` (real-valued fields in
 * batch precision), optionally passed through a corruptor `ops[k]` with parameter `params[k]`:
 * 0 none, 1 (v > p ? 1 : 0) (walkers' touch sensors, legacy_base.py:262-265), 2 tanh(2 v / p) (torque sensors,
 * cmu_humanoid.py:462-465), 3 log1p(v), 4 asinh(v).  dmc_gather_run writes out[env * nrows + k], a (B, nrows)
 * env-major matrix in batch precision, on `hip_stream`; fields rebound with dmc_batch_bind are followed. */
typedef struct dmc_gather dmc_gather;
int dmc_gather_create(dmc_batch* b, int nrows, const char* const* field_names, const int* rows, const int* ops,
                      const double* params, dmc_gather** out);
void dmc_gather_destroy(dmc_gather* g);
int dmc_gather_run(dmc_gather* g, void* out, void* hip_stream);

/* info[0..19] = {B, precision, lanes_per_env, waves_per_block, envs_per_block,
 * lds_bytes_per_block, grid, nconmax, njmax, env_scratch_bytes, static_id,
 * jac_kmax, table_lds_bytes, envs_per_cu, njdense, njcon, stash_on, stash_bytes_per_env,
 * global_scratch_bytes_per_env, work_queue}
 * (static_id >= 0: a model-specialised kernel instantiation is in use; jac_kmax: entries per
 * compressed contact Jacobian row; envs_per_cu: environments resident on one CU under the
 * 160 KiB LDS budget; global_scratch_bytes_per_env: what a tree-sparse model (nv > 16) keeps in
 * device memory instead of LDS -- the compressed contact rows and, with noslip, the factor of M; work_queue: 1 when
 * the batch is larger than what the chip holds at once, so `grid` is only the resident workgroups and their waves
 * claim the remaining environments from a device-side queue as they finish.  Round 6: one queue per XCD, served by the
 * waves that run on it, and a Physics.step(nstep > 1) launch hands every environment out in up to 8 PIECES of physics
 * steps, round by round -- piece s of every environment before piece s + 1 of any; the state travels between pieces
 * through a per-environment hand-off record -- so that the launch ends within one physics step of the last claim
 * instead of within one env-step: trajectories bit-identical, config 4 460 k -> 577 k env-steps/s.  DMC_SLICES=1
 * restores whole items). */
int dmc_batch_info(const dmc_batch* b, int* info);

/* Profiling contract.  Replaces: Physics.enable_profiling() -> wrapper.enable_timer(True), which installs mjcb_time
 * (dm_control/mujoco/engine.py:135-137, mujoco/wrapper/core.py:77-81), and the mjData.timer[mjTIMER_STEP] record
 * suite/wrappers/mujoco_profiling.py:94-103 reads (`timer[0].duration`, `timer[0].number`).
 * With profiling enabled every launch is bracketed by hipEvents on its stream.  timer 0 = mjTIMER_STEP: seconds spent
 * in step launches and the number of physics steps (mj_step calls) they ran; timer 1 = mjTIMER_FORWARD: seconds and
 * count of mj_forward launches.  Reading a timer waits for the launches issued so far.  Launches recorded into a HIP
 * graph are not timed. */
int dmc_batch_enable_profiling(dmc_batch* b, int enabled);
int dmc_batch_get_timer(dmc_batch* b, int timer, double* duration_s, long long* number);

/* Time `reps` back-to-back step launches with hipEvents on `hip_stream`;
 * returns the average milliseconds per launch in *ms_per_launch. */
int dmc_batch_time_steps(dmc_batch* b, int nstep, int legacy_step, int reps, void* hip_stream, float* ms_per_launch);

/* Debug: dump the LDS scratch of the first `n` envs after the forward pass of
 * the last substep; read arrays back by scratch name (step_layout.h). */
int dmc_batch_debug_enable(dmc_batch* b, int n);
int dmc_batch_debug_get(dmc_batch* b, const char* scratch_name, int env, double* dst, int* count);

/* Wave trace of the last 8 launches (tuning: the tail of a launch whose environments all run at once is its slowest
 * wave; the time between the launches is what is left of a launch once its waves are accounted for).  dst == NULL:
 * enable / disable.  dst != NULL: copies the ring, (8, 8, nitems) ints -- slot (launch % 8), rows = when the item's
 * wave entered the kernel, started and finished the item on the 100 MHz constant clock (low 31 bits), its workgroup
 * index, and the clock after the opening position / velocity stage, the first acceleration stage, the first
 * integration and the trailing stage of a step launch; *nitems = ceil(B * lanes_per_env / 64). */
int dmc_batch_wave_trace(dmc_batch* b, int enable, int32_t* dst, int* nitems);

/* Per-phase shader-cycle profile of the fused kernel (libraries built with
 * -DDMC_PROFILE only; otherwise enable fails).  dst[k] = mean cycles per env of
 * phase k (order: step_core.h PROF_*), accumulated since enable. */
int dmc_batch_prof_enable(dmc_batch* b, int enable);
int dmc_batch_prof_get(dmc_batch* b, double* dst, int* n);

#ifdef __cplusplus
}
#endif
#endif  /* DMC_BATCH_H_ */
