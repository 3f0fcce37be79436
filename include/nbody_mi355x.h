/*
 * nbody_mi355x.h -- C ABI of libnbody_mi355x.so, the MI355X (gfx950) all-pairs
 * N-body hot path behind the particle_simulation.h API of CPestka/N-Body-Sim.
 *
 * Plain C: compiled by gcc/g++ on the host side (no HIP headers needed), by
 * hipcc on the implementation side.  Only pointers, sizes and PODs cross the
 * boundary; no exception does.  Every function returns NBODY_MI355X_OK (0) or
 * a negative nbody_mi355x_status; the text of the last failure on the calling
 * thread is nbody_mi355x_last_error().
 *
 * Citations "ref:" are file:line into the reference's src/ directory and name
 * the reference interface the entry point replaces.
 *
 * Two levels:
 *   (1) stateless device-pointer entry points (nbody_mi355x_step_f64 ...):
 *       the hot path itself on caller-owned device memory and a caller-owned
 *       HIP stream.  This is what a multi-process (one rank per GPU) driver
 *       uses, with its own collective between steps.
 *   (2) the handle API (nbody_mi355x_create ... _destroy): host pointers in the
 *       reference's own record layouts in, snapshots out; this is what
 *       ParticleSimulation::SimulateMI355X() binds.
 */
#ifndef NBODY_MI355X_H_
#define NBODY_MI355X_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NBODY_MI355X_OK = 0,
  NBODY_MI355X_ERR_INVALID = -1,  /* bad argument (null, size, alignment)        */
  NBODY_MI355X_ERR_ALLOC = -2,    /* ref: "Failed memory allocation on device."   */
  NBODY_MI355X_ERR_H2D = -3,      /* ref: "Failed memory copy to device."         */
  NBODY_MI355X_ERR_D2H = -4,      /* ref: "Failed to copy results to host."       */
  NBODY_MI355X_ERR_FREE = -5,     /* ref: "Failed to free device memory."         */
  NBODY_MI355X_ERR_LAUNCH = -6,   /* kernel launch / execution error (the reference
                                     never checks this, particle_simulation_GPU.cu:149-155) */
  NBODY_MI355X_ERR_NO_DEVICE = -7 /* no usable HIP device                         */
} nbody_mi355x_status;

/* The simulator's gravitational constant, ref: particle_simulation.h:713 and
 * particle_simulation_GPU.cu:75 (6.7430e-11, not 6.67430e-11). */
#define NBODY_MI355X_G_REFERENCE 6.7430e-11

/* Arithmetic of the force accumulation. */
#define NBODY_MI355X_FP64 0 /* everything in double (parity path)                */
#define NBODY_MI355X_FP32 1 /* forces MAY be evaluated in float: r^2, r^-3 and the running
                               sums; in the pair-symmetric kernels also the separations, taken
                               from coordinates RELATIVE to the row's centroid (subtracted in
                               double, then narrowed), so that a system far from the origin
                               loses nothing, and SCALED by a power of two per row, so that
                               neither r^2 nor r^-3 leaves float's range in a scene of any size
                               (masses must be float-representable: 1e-38 .. 3e38); state,
                               integration and snapshots stay double.
                               Systems of at most NBODY_MI355X_WAVE_MAX_BODIES bodies are
                               latency-bound and run the double kernels either way (faster there) */

#define NBODY_MI355X_FP64_FAST 2 /* opt-in, never the default: FP64 with the correction series of r^-3
                               cut after its linear term, y0^3 (1 + 3/2 e) instead of
                               y0^3 (1 + 3/2 e + 15/8 e^2): 17 instead of 18 FP64 operations per pair,
                               +5.7 % at N = 1M (3313 against 3129 Ginteractions/s, one lease).  The
                               truncation is <= 15/8 e^2 <= 2.7e-14 relative per pair (e <= 2^-23:
                               v_rsq_f64's seed), always towards a smaller r^-3: three to four orders
                               inside north_star's 1e-10 after 100 steps, but outside the 1e-12 /
                               1e-13 this library's FP64 path is tested to -- hence a tier of its
                               own.  It changes the pair-symmetric kernels' equal-mass form with
                               2048-body workgroup tiles (from 262 144 slots) and 512-body wave tiles
                               (from 13 312 slots); every other kernel computes plain FP64. */

/* Time integrator fused into the step's epilogue.
 * EULER: the reference's forward Euler, v' = v + dt a, r' = r + dt v (OLD v)
 *        (ref: particle_simulation.h:288-300) -- the parity path.
 * KICK_DRIFT: r' = r + dt v' (NEW v): the symplectic (semi-implicit) Euler step.  First
 *        order like EULER, but its energy error stays bounded; same kernel, same cost.
 * KDK:   kick-drift-kick leapfrog, second order, still ONE force evaluation per small step:
 *        v += dt/2 a(r);  r += dt v;  v += dt/2 a(r')   with a(r') carried over to the next
 *        step's first kick.  Neither is in the reference.
 * Level 2 (nbody_mi355x_config.integrator) takes EULER, KICK_DRIFT or KDK and sequences the
 * pieces itself.  At level 1 a KDK small step is made of
 *   nbody_mi355x_kdk_drift_f64: v += (dt/2 a) G from args->kdk_acc, pos_next = pos_cur + dt v
 *        -> exchange positions, swap buffers ->
 *   nbody_mi355x_step_f64 with integrator = KDK: forces at pos_cur, v += (dt/2 a) G, the raw
 *        sums stored to args->kdk_acc, snapshot {r, v, a G}; positions are not written;
 * or, between two small steps, of ONE call with integrator = KDK_CONTINUE, whose epilogue is
 * that second kick followed by the next step's first kick and drift (pos_next written; a
 * snapshot holds the synchronised {r, v, a G} at pos_cur).  kdk_acc of the very first step is
 * obtained by a KDK call with dt = 0. */
#define NBODY_MI355X_EULER 0
#define NBODY_MI355X_KICK_DRIFT 1
#define NBODY_MI355X_KDK 2
#define NBODY_MI355X_KDK_CONTINUE 3 /* level 1 only */

/* Tile width of the j loop; shard_slots must be a multiple of it. */
#define NBODY_MI355X_TILE 256

const char* nbody_mi355x_last_error(void);
int nbody_mi355x_device_count(int* count);

/* ------------------------------------------------------------------ level 1
 *
 * Device layout ("slab" layout).  Bodies live in num_shards slabs of
 * shard_slots slots each (slot g = shard * shard_slots + offset); slab s keeps
 * its coordinates as three contiguous runs
 *     pos[s*3*shard_slots + 0*shard_slots + offset]  x
 *     pos[s*3*shard_slots + 1*shard_slots + offset]  y
 *     pos[s*3*shard_slots + 2*shard_slots + offset]  z
 * so that one slab is one contiguous block (one in-place all-gather moves
 * every rank's updated coordinates), and mass[g] is a flat array of
 * num_shards*shard_slots doubles.  Slots past the real bodies of a slab are
 * padding: mass 0, position 0, never written.  A single-GPU run is
 * num_shards = 1.  vel, acc_in, acc_out hold the calling rank's own slab only:
 * [3][shard_slots].
 */
typedef struct {
  const double* pos_cur;  /* all slabs, read                                   */
  double* pos_next;       /* all slabs; only own slab's real slots are written */
  const double* mass;     /* all slabs                                         */
  double* vel;            /* own slab, updated in place                        */
  const double* acc_in;   /* own slab partial sums to start from, or NULL (0)  */
  double* acc_out;        /* if not NULL: store the raw sums (no G) here and
                             do NOT integrate (first half of a split step)    */
  double* snapshot;       /* if not NULL: DeviceOutputParticle<double>
                             [own_count] = {r', v', a*G} (ref:
                             particle_simulation.h:26-31), device memory      */
  int64_t shard_slots;    /* multiple of NBODY_MI355X_TILE.  The pair-symmetric
                             kernels take the largest block that divides it: round
                             to 512 slots for systems of 12 801 .. 261 632 bodies in
                             all and to 2048 beyond (the handle API does) to get
                             the blocks the measurements in DESIGN.md are for     */
  int32_t num_shards;
  int32_t own_shard;
  int64_t own_count;      /* real bodies in the own slab (<= shard_slots)      */
  /* j slots to interact with: [j0_begin, j0_end) then [j1_begin, j1_end);
   * begins must be multiples of NBODY_MI355X_TILE; an empty range has
   * begin == end.  A full step is {0, num_shards*shard_slots, 0, 0}. */
  int64_t j0_begin, j0_end, j1_begin, j1_end;
  double dt;              /* ref: stepsize_                                    */
  double G;               /* NBODY_MI355X_G_REFERENCE for parity               */
  double eps2;            /* softening^2; 0 = reference semantics: pairs with
                             r^2 == 0 contribute nothing (ref:
                             particle_simulation.h:269)                        */
  int32_t precision;      /* NBODY_MI355X_FP64 / _FP32 / _FP64_FAST             */
  int32_t variant;        /* 0 = default kernel choice; >0 selects a specific
                             kernel build (tuning / A-B benchmarking);
                             NBODY_MI355X_VARIANT_SYM* = pair-symmetric kernels */
  void* workspace;        /* device scratch for the pair-symmetric kernels, or
                             NULL (then only one-sided kernels are used)       */
  int64_t workspace_bytes;
  int32_t phase;          /* pair-symmetric kernels only: 0 = the whole force sum;
                             1 = the part that needs only the own slab's
                             positions (runs while the all-gather is in flight;
                             results stay in the workspace); 2 = the rest, then
                             output.  With num_shards > 1 the output is acc_out
                             [num_shards][3][shard_slots]: this rank's partial
                             sums for EVERY slot, to be reduce-scattered and
                             integrated by a one-sided step with acc_in and
                             empty j ranges.                                   */
  int32_t integrator;     /* NBODY_MI355X_EULER (the reference's, default),
                             _KICK_DRIFT, _KDK or _KDK_CONTINUE                */
  double* kdk_acc;        /* KDK / KDK_CONTINUE: if not NULL, [3][shard_slots] own slab,
                             receives the raw sums (no G) the kick was made with; the
                             drift (nbody_mi355x_kdk_drift_f64) reads them       */
} nbody_mi355x_step_args;

#define NBODY_MI355X_VARIANT_WAVE  7 /* one-sided, a wave per 1, 2 or 4 bodies: the default for whole
                                        FP64 steps over at most NBODY_MI355X_WAVE_MAX_BODIES slots */
#define NBODY_MI355X_WAVE_MAX_BODIES 2048
#define NBODY_MI355X_VARIANT_TINY  8 /* handle API only, at most 64 bodies: a whole big step in
                                        one launch of one wave (the default there); variants
                                        1-7 keep one launch per small step */
#define NBODY_MI355X_VARIANT_SYM  10 /* pair-symmetric, block size chosen by the library */
#define NBODY_MI355X_VARIANT_SYM1 11 /* pair-symmetric, 256-body blocks  */
#define NBODY_MI355X_VARIANT_SYM2 12 /* pair-symmetric, 512-body blocks  */
#define NBODY_MI355X_VARIANT_SYM4 14 /* pair-symmetric, 1024-body blocks */
#define NBODY_MI355X_VARIANT_SYM8 18 /* pair-symmetric, 2048-body blocks */
/* pair-symmetric with WAVE tiles (mid-size systems): a workgroup's 4 waves hold the same
 * 64, 128, 256 or 512 bodies and each takes its own groups of 64 j bodies */
#define NBODY_MI355X_VARIANT_SYMW1 21
#define NBODY_MI355X_VARIANT_SYMW2 22
#define NBODY_MI355X_VARIANT_SYMW4 24
#define NBODY_MI355X_VARIANT_SYMW8 28

/* One small step (or one half of a split one) for the own slab's bodies:
 * a_i = sum_j m_j (r_j - r_i) / |r_j - r_i|^3 over the j ranges, then
 * v' = v + (dt*a)*G, r' = r + dt*v (old v), optional snapshot.
 * Replaces ref: MakeSmallStepDevice, particle_simulation_GPU.cu:60-142 (same
 * maths as MakeSmallStep, particle_simulation.h:248-322).  Asynchronous on
 * `stream` (a hipStream_t, NULL = the default stream). */
int nbody_mi355x_step_f64(const nbody_mi355x_step_args* args, void* stream);

/* First half of a kick-drift-kick small step for the own slab's bodies (see the integrator
 * notes above): vel += (dt/2 * kdk_acc) * G, pos_next = pos_cur + dt * vel.  Fields read:
 * pos_cur, pos_next, vel, kdk_acc, shard_slots, num_shards, own_shard, own_count, dt, G.
 * O(N), memory-bound.  Sits beside the reference's Euler update, particle_simulation.h:282-300. */
int nbody_mi355x_kdk_drift_f64(const nbody_mi355x_step_args* args, void* stream);

/* Which kernel build nbody_mi355x_step_f64 runs for these args: *variant = a concrete
 * NBODY_MI355X_VARIANT_* -- SYM1 .. SYM8 / SYMW1 .. SYMW8 (pair-symmetric, block size
 * resolved), WAVE, or a one-sided build 1 .. 6.  Read-only, no launch.  With variant = 0 and
 * workspace = NULL (or too small) the answer is a one-sided build: that is how a caller sees
 * that the pair-symmetric kernels were NOT taken.  No counterpart in the reference (one
 * kernel, particle_simulation_GPU.cu:60). */
int nbody_mi355x_step_variant(const nbody_mi355x_step_args* args, int32_t* variant);

/* The launch plan of the pair-symmetric kernels for these args (a whole, unbatched step), for
 * tests and tuning tools; all 0 where nbody_mi355x_step_variant names a one-sided build.  Read-only,
 * no launch, runs without a GPU.  plan[NBODY_MI355X_PLAN_FIELDS]:
 *   [0] family: 1 workgroup tiles, 2 wave tiles     [1] slots per block     [2] block rows of the own slab
 *   [3] largest block distance D (a row's list: tiles at distances 0 .. D)
 *   [4] workgroups per block row of the main launch
 *   wave tiles only -- a row's list is [8] octets (8 rotation steps each), dealt to the row's
 *   4 * [4] waves: the first [5] waves take [6] octets each, the others [7] (even dealing: [5] = all,
 *   [7] = [6]); wave k begins at octet k [6] (k <= [5]) or [5] [6] + (k - [5]) [7]
 *   [9] R rows, [10] Cb rows of the workspace (of 3 * shard_slots doubles each)
 * No counterpart in the reference (one kernel, one launch shape, particle_simulation_GPU.cu:168-176). */
#define NBODY_MI355X_PLAN_FIELDS 12
int nbody_mi355x_step_plan(const nbody_mi355x_step_args* args, int32_t* plan);

/* How many launches of its main (force) kernel a step with these args takes: 1, or -- for the
 * pair-symmetric kernels with a workspace smaller than the unbatched size -- the number of
 * batches the block distances are walked in.  For measurement accounting (work per launch =
 * work per step / launches); read-only, no launch.  No counterpart in the reference. */
int nbody_mi355x_step_tile_launches(const nbody_mi355x_step_args* args, int32_t* launches);

/* Scratch the pair-symmetric form of the step needs for these args (fields
 * shard_slots, num_shards, variant, precision are read): *bytes = the preferred size --
 * every block distance at once, capped at 4 GiB (or four times the minimum); any size from
 * the minimum up is accepted by the step, which then walks the distances in batches --
 * or 0 if that form does not apply (slot count not a multiple of the block, fewer than two
 * blocks).  Several slabs are fine (the step then needs acc_out, see `phase`).
 * The symmetric kernels evaluate each unordered pair once (Newton's third law)
 * and reduce partial sums through this workspace in a fixed order: same maths,
 * different summation order, bit-reproducible.  No counterpart in the reference. */
int nbody_mi355x_step_workspace_bytes(const nbody_mi355x_step_args* args, int64_t* bytes);

/* The smallest workspace the same step accepts (one batch of distances at a time), for
 * callers short of device memory: anything between this and the preferred size works, more
 * is faster up to the preferred size.  *bytes = 0 where step_workspace_bytes gives 0.
 * The handles use it themselves: create / create_multi take the preferred size if it is
 * under half of the free device memory, else what is under half of it down to this
 * minimum, and only below that run the one-sided kernel.  No counterpart in the reference. */
int nbody_mi355x_step_workspace_min_bytes(const nbody_mi355x_step_args* args, int64_t* bytes);

/* The same small step for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in
 * the reference (whose GPU path allocates per run, particle_simulation_GPU.cu:13-35).  It is
 * nbody_mi355x_step_f64 restricted to the pair-symmetric kernels (variant 0 or
 * NBODY_MI355X_VARIANT_SYM*), with their partial-sum workspace in a device array that is part
 * of the library's code object -- NBODY_MI355X_STATIC_SCRATCH_BYTES on every GPU the library
 * runs on, held until the process ends.  args->workspace / workspace_bytes are ignored.
 * It calls no hipMalloc / hipFree or any other allocating routine.  Layouts whose MINIMUM
 * workspace exceeds the array (one slab: every size up to 124 928 bodies fits,
 * some layouts up to about 200 000 -- nbody_mi355x_step_workspace_min_bytes tells) are rejected with
 * NBODY_MI355X_ERR_INVALID and a text that names the size needed; so are one-sided variants
 * (they need no scratch: call nbody_mi355x_step_f64) and layouts the pair-symmetric kernels
 * do not apply to (fewer than two blocks of the variant's size; with variant 0 the smallest
 * layout is one 256-slot slab, taken as 128-slot wave-tile blocks).  Results are bit-identical to
 * nbody_mi355x_step_f64 given a workspace of NBODY_MI355X_STATIC_SCRATCH_BYTES.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two host threads) race -- serialise them.  Nothing in the array needs clearing between
 * calls.  The first call on each device resolves the array's address there
 * (hipGetSymbolAddress): make it outside of graph capture. */
#define NBODY_MI355X_STATIC_SCRATCH_BYTES 134217728 /* 128 MiB */
int nbody_mi355x_step_f64_static(const nbody_mi355x_step_args* args, void* stream);

/* Particle<double>[count] AoS {r[3], v[3]} (ref: particle_simulation.h:20-24)
 * on the device -> own slab's pos (x,y,z runs) + vel, and back.  These are the
 * layout converters either side of the step kernel. */
int nbody_mi355x_pack_f64(const double* particles_aos, int64_t count, double* pos_slab,
                          double* vel, int64_t shard_slots, void* stream);
int nbody_mi355x_unpack_f64(const double* pos_slab, const double* vel, int64_t count,
                            int64_t shard_slots, double* particles_aos, void* stream);

/* Energy diagnostic (the reference has none; SURVEY 8d correctness gate):
 * ke = sum 1/2 m v^2 over the own slab, pe = -G * sum over own i, all j > i of
 * m_i m_j / r_ij (coincident pairs skipped).  scratch: device, >= 2 doubles
 * per 256 own slots + 2; result[2] = {ke, pe}: device.  Asynchronous. */
int nbody_mi355x_energy_f64(const double* pos, const double* mass, const double* vel,
                            int64_t shard_slots, int32_t num_shards, int32_t own_shard,
                            int64_t own_count, double G, double* scratch, double* result,
                            void* stream);

/* ------------------------------------------------------------------ level 2 */
typedef struct nbody_mi355x_sim nbody_mi355x_sim; /* opaque */

typedef struct {
  int64_t num_particles; /* ref: num_particles_                                */
  double stepsize;       /* ref: stepsize_                                     */
  double G;              /* 0 -> NBODY_MI355X_G_REFERENCE                      */
  double eps2;           /* 0 = reference semantics                            */
  int32_t precision;     /* NBODY_MI355X_FP64 / _FP32 / _FP64_FAST             */
  int32_t device;        /* HIP device ordinal                                 */
  int32_t host_is_float; /* 0: host records are double (ParticleSimulation<double,W>),
                            1: float (ParticleSimulation<float,W>)             */
  int32_t variant;       /* as in nbody_mi355x_step_args                       */
  int32_t integrator;    /* NBODY_MI355X_EULER (default) / _KICK_DRIFT / _KDK     */
  int32_t reserved;
} nbody_mi355x_config;

/* ref: AllocateDeviceMemory, particle_simulation_GPU.cu:13-35 (4 cudaMalloc). */
int nbody_mi355x_create(const nbody_mi355x_config* cfg, nbody_mi355x_sim** sim);
/* The same handle spread over several GPUs driven by one host thread (SURVEY 8b
 * "device list", 8e; the reference has one GPU, particle_simulation_GPU.cu:13-35).
 * devices[k] owns the k-th contiguous i-range of bodies -- the partition of the
 * reference's CPU threads, particle_simulation.h:98-110.  Per small step the
 * shards exchange new positions (all-gather) and, on the pair-symmetric path,
 * partial sums (reduce-scatter); the host does not block inside a big step.
 * The exchanges are RCCL collectives over xGMI (one communicator over the listed
 * devices, librccl resolved at run time) whenever every shard has a GPU of its own
 * and the communicator passes a bit-exact self-check at create -- a patterned
 * buffer all-gathered and reduce-scattered both by RCCL and by peer copies;
 * otherwise, or with NBODY_MI355X_MULTI_RCCL=0 in the environment, they are peer
 * copies ordered with events (nbody_mi355x_exchange_backend says which).
 * A device ordinal may appear more than once (several shards on one GPU: how the
 * path is rehearsed on a one-GPU machine; peer copies).  cfg->device is ignored.
 * Every other handle function works on the result unchanged.  On success
 * nbody_mi355x_last_error() is empty, or holds a note: direct peer access between two of
 * the devices could not be enabled (the runtime then stages those copies through the host),
 * or RCCL could not be used and why (library missing, init failed, self-check failed). */
int nbody_mi355x_create_multi(const nbody_mi355x_config* cfg, const int32_t* devices,
                              int32_t num_devices, nbody_mi355x_sim** sim);
/* ref: FreeDeviceMemory, particle_simulation_GPU.cu:216-231. */
int nbody_mi355x_destroy(nbody_mi355x_sim* sim);
/* ref: CopyDataToDevice, particle_simulation_GPU.cu:40-54.  particles =
 * Particle<T>[N], mass = T[N], borrowed for the duration of the call. */
int nbody_mi355x_upload(nbody_mi355x_sim* sim, const void* particles, const void* mass);
/* ref: MakeBigStepDevice, particle_simulation_GPU.cu:145-178, plus that big
 * step's share of CopyResultsToHost (:198-211): runs `substeps` small steps
 * back to back on the device with no host synchronisation in between; if
 * snapshot_out is not NULL the LAST small step also writes
 * DeviceOutputParticle<T>[N] = {r, v, a*G} there (host memory).  Blocks until
 * the big step is complete (the reference's per-big-step timing needs that). */
int nbody_mi355x_run(nbody_mi355x_sim* sim, int64_t substeps, void* snapshot_out);
/* (A run that fails part-way -- ERR_LAUNCH and the like -- leaves the state undefined, on a
 * multi-device handle possibly different from shard to shard: upload or generate again.) */
/* The same in two halves, so that the host can unpack / write big step k's
 * snapshot while big step k+1 runs (the reference copies every
 * snapshot at the very end, .cu:198-211): run_async enqueues the big step and the
 * copy into snapshot_out and returns; wait blocks until both are complete.
 * snapshot_out must stay valid and untouched until wait returns; for the copy to
 * overlap with the host it should be page-locked (nbody_mi355x_host_alloc).
 * One big step may be pending per handle; upload / download / energy / destroy
 * wait for it first.  nbody_mi355x_run == run_async + wait. */
int nbody_mi355x_run_async(nbody_mi355x_sim* sim, int64_t substeps, void* snapshot_out);
int nbody_mi355x_wait(nbody_mi355x_sim* sim);
/* Page-locked host memory for such buffers (hipHostMalloc / hipHostFree): no
 * counterpart in the reference, which uses pageable new[] memory (.cu:200). */
int nbody_mi355x_host_alloc(int64_t bytes, void** buffer);
int nbody_mi355x_host_free(void* buffer);
/* Current state back as Particle<T>[N]. */
int nbody_mi355x_download(nbody_mi355x_sim* sim, void* particles);
/* Total kinetic and potential energy of the current state. */
int nbody_mi355x_energy(nbody_mi355x_sim* sim, double* ke, double* pe);
/* Which kernel a small step of this handle runs (as nbody_mi355x_step_variant; TINY for the
 * one-launch-per-big-step path) and the pair-symmetric workspace the handle holds, bytes
 * (0: none -- then a system above NBODY_MI355X_WAVE_MAX_BODIES runs a one-sided kernel, at
 * roughly half the rate: create() takes a workspace only if at least the kernels' minimum,
 * nbody_mi355x_step_workspace_min_bytes, fits in half of the free device memory).  On a multi-device handle: shard 0's.  No counterpart in the reference. */
int nbody_mi355x_kernel_in_use(nbody_mi355x_sim* sim, int32_t* variant, int64_t* workspace_bytes);
/* Device time (hipEvent) of the kernels of the last nbody_mi355x_run, seconds. */
int nbody_mi355x_last_run_seconds(nbody_mi355x_sim* sim, double* seconds);
/* Device time of the two exchanges of the LAST small step of the last completed run, seconds,
 * the longest over the shards: *positions = the all-gather of the new positions (on the comm
 * streams, under the next step's diagonal tiles), *partial_sums = the reduce-scatter of the
 * pair-symmetric partial sums (on the compute streams: NOT overlapped, this is what it costs a
 * step); 0 where the step had no such exchange and on a one-device handle.  The reference has
 * one GPU and no exchange (particle_simulation_GPU.cu:145-178); it stands beside its per-big-step
 * timer, particle_simulation.h:203-207. */
int nbody_mi355x_last_exchange_seconds(nbody_mi355x_sim* sim, double* positions, double* partial_sums);
/* How a multi-device handle exchanges: NONE (one device), PEER_COPIES or RCCL (see create_multi). */
#define NBODY_MI355X_EXCHANGE_NONE 0
#define NBODY_MI355X_EXCHANGE_PEER_COPIES 1
#define NBODY_MI355X_EXCHANGE_RCCL 2
int nbody_mi355x_exchange_backend(nbody_mi355x_sim* sim, int32_t* backend);

/* --------------------------------------------------- initial conditions (host)
 * Synthetic scenes with an in-library portable RNG (splitmix64-seeded
 * xoshiro256**), SURVEY 8d configs 3-5.  particles: Particle<double>[n],
 * mass: double[n].  Units: G*M_total = 1 with the G passed in. */
int nbody_mi355x_ic_plummer(int64_t n, uint64_t seed, double G, double* particles, double* mass);
int nbody_mi355x_ic_uniform_cube(int64_t n, uint64_t seed, double G, double* particles, double* mass);
/* A ring around a central body (body 0): the recipe of the reference's
 * saturn_ring.cc:42-63 / NonCircularOrbitMoon (particle_initializer.h:120-175)
 * for any n with the portable generator.  SI units; the
 * reference's values: central_mass 5.683e26, ring_radius 1.12e8, width_sigma
 * 3.0e7, thickness_sigma 50, v_sigma 50, mass_sigma 0.1. */
int nbody_mi355x_ic_ring(int64_t n, uint64_t seed, double central_mass, double ring_radius,
                         double width_sigma, double thickness_sigma, double v_sigma,
                         double mass_sigma, double* particles, double* mass);

/* ------------------------------- initial conditions at scale (device, and host threads)
 * The same scenes with ONE STREAM PER BODY (seeded from (seed, body index)), so that body i
 * is a pure function of (scene, i): any range of bodies can be made anywhere -- a GPU thread
 * per body, straight into the slab layout, or a host thread per range -- SURVEY 8f-3.  The
 * reference draws a whole scene from one libstdc++ stream on the host, a SetParticle() at a
 * time (saturn_ring.cc:42-63, particle_initializer.h:120-175); nbody_mi355x_ic_* above keep
 * that shape.  A scene made here is statistically, not bit for bit, the scene ic_* makes from
 * the same seed; device and host agree with each other up to the last bits of pow/log/cos. */
#define NBODY_MI355X_SCENE_PLUMMER 0      /* as ic_plummer; the centre of mass and the mean
                                             velocity are removed by nbody_mi355x_generate only */
#define NBODY_MI355X_SCENE_UNIFORM_CUBE 1 /* as ic_uniform_cube */
#define NBODY_MI355X_SCENE_RING 2         /* as ic_ring (body 0: the central mass)            */
typedef struct nbody_mi355x_scene {
  int32_t scene;         /* NBODY_MI355X_SCENE_*                                */
  int32_t reserved;      /* 0                                                   */
  uint64_t seed;
  int64_t num_particles; /* bodies of the WHOLE scene (masses are 1 / (G n))    */
  double G;              /* PLUMMER / UNIFORM_CUBE: units G * M_total = 1       */
  double central_mass, ring_radius, width_sigma, thickness_sigma, v_sigma, mass_sigma; /* RING */
} nbody_mi355x_scene;
/* Level 1: bodies [first, first + count) of the scene into one slab of the device layout
 * (pos_slab: x at +0, y at +shard_slots, z at +2*shard_slots; vel likewise, or NULL for a
 * slab whose velocities live elsewhere; mass_slab[count]).  Asynchronous on `stream`. */
int nbody_mi355x_scene_fill_f64(const nbody_mi355x_scene* scene, int64_t first, int64_t count,
                                double* pos_slab, double* vel, double* mass_slab, int64_t shard_slots,
                                void* stream);
/* sums6 (device, 6 doubles) = sums of the slab's first `count` positions and (vel != NULL)
 * velocities, added in a fixed order; and pos -= shift6[0..2], vel -= shift6[3..5] (host
 * values).  What removing a centre of mass takes. */
int nbody_mi355x_slab_sums_f64(const double* pos_slab, const double* vel, int64_t count,
                               int64_t shard_slots, double* sums6, void* stream);
int nbody_mi355x_slab_shift_f64(double* pos_slab, double* vel, int64_t count, int64_t shard_slots,
                                const double* shift6, void* stream);
/* The same bodies on the host, on up to 16 threads: Particle<double>[count], double[count]. */
int nbody_mi355x_scene_host(const nbody_mi355x_scene* scene, int64_t first, int64_t count,
                            double* particles, double* mass);
/* Level 2: the whole scene straight into the handle's device state, in place of upload()
 * (scene->num_particles must be the handle's); on a multi-device handle every device makes
 * the bodies it holds.  PLUMMER: centre of mass and mean velocity are removed.  A scene that
 * is refused (unknown, wrong size, bad parameters) leaves the state as it was; after any other
 * error the state is undefined: generate or upload again. */
int nbody_mi355x_generate(nbody_mi355x_sim* sim, const nbody_mi355x_scene* scene);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_MI355X_H_ */
