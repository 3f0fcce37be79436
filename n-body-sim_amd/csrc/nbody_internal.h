// Shared between the translation units of libnbody_mi355x.so; not installed.
#pragma once
#include <cstdarg>

#include "nbody_mi355x.h"

namespace nbody_internal {
// Text behind nbody_mi355x_last_error() (thread-local).
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// variants that name a pair-symmetric kernel build (workgroup tiles 10-18, wave tiles 21-28)
inline bool is_sym_variant(int v) {
  return v == NBODY_MI355X_VARIANT_SYM || v == NBODY_MI355X_VARIANT_SYM1 || v == NBODY_MI355X_VARIANT_SYM2 ||
         v == NBODY_MI355X_VARIANT_SYM4 || v == NBODY_MI355X_VARIANT_SYM8 || v == NBODY_MI355X_VARIANT_SYMW1 ||
         v == NBODY_MI355X_VARIANT_SYMW2 || v == NBODY_MI355X_VARIANT_SYMW4 || v == NBODY_MI355X_VARIANT_SYMW8;
}
// slots per block of a variant that pins one (0: the library chooses)
inline int64_t block_of_variant(int v) {
  switch (v) {
    case NBODY_MI355X_VARIANT_SYM1: return 256;
    case NBODY_MI355X_VARIANT_SYM2: return 512;
    case NBODY_MI355X_VARIANT_SYM4: return 1024;
    case NBODY_MI355X_VARIANT_SYM8: return 2048;
    case NBODY_MI355X_VARIANT_SYMW1: return 64;
    case NBODY_MI355X_VARIANT_SYMW2: return 128;
    case NBODY_MI355X_VARIANT_SYMW4: return 256;
    case NBODY_MI355X_VARIANT_SYMW8: return 512;
    default: return 0;
  }
}
// slots of a slab holding `bodies`: whole tiles, and whole blocks of the pair-symmetric kernels
// the library would choose -- wave tiles of up to 256 bodies below 13 312 slots, of 512 up to
// 262 143, workgroup tiles of 2048 beyond (the padding is massless bodies: <= 3 % more slots from
// 8192 bodies up; round 3: the 1024- / 2048-slot units of round 1's workgroup tiles from 8192 / 65 536
// bodies cost up to 12 % more slots -- 8448 bodies ran as 9216) -- or the variant pins
// (`shards` slabs of this size make the system: the block follows the TOTAL, so a rank's slab of a
// large system is whole 2048-slot blocks whatever the rank's own count)
inline int64_t slab_slots(int64_t bodies, int variant, int shards = 1) {
  const int64_t P = shards > 0 ? shards : 1;
  int64_t unit = bodies * P > 12800 ? 512 : NBODY_MI355X_TILE;
  if ((bodies + unit - 1) / unit * unit * P > 262143) unit = 2048;  // beyond the wave tiles (kWaveTileMaxSlots, nbody_sym.hip)
  if (block_of_variant(variant) > unit) unit = block_of_variant(variant);
  return (bodies + unit - 1) / unit * unit;
}
// nbody_kernels.hip: host-side shape checks of a step's arguments -- everything the kernels'
// indexing assumes, verified before any launch
int check_step_args(const nbody_mi355x_step_args* a);
// nbody_sym.hip: pair-symmetric step (args already validated by check_step_args)
bool sym_supported(const nbody_mi355x_step_args& a, int64_t* preferred_bytes, int64_t* least_bytes);
int64_t workspace_budget(size_t free_bytes);
int64_t workspace_to_take(const nbody_mi355x_step_args& probe, int64_t budget);
int sym_step(const nbody_mi355x_step_args& a, struct ihipStream_t* stream);
// the concrete variant (NBODY_MI355X_VARIANT_SYM1 .. SYMW8) the plan for these args resolves to, -1: none
int sym_resolve(const nbody_mi355x_step_args& a);
int sym_batches(const nbody_mi355x_step_args& a);
// the launch plan of the pair-symmetric step for these args (nbody_mi355x_step_plan); false: none applies
bool sym_plan_query(const nbody_mi355x_step_args& a, int32_t* plan);
// nbody_kernels.hip: `substeps` whole small steps of a system of at most kTinyMaxBodies bodies
// in ONE launch of one wave (in place on a.pos_cur / a.vel; a.snapshot, if set, receives the
// last small step's records).  One slab, FP64, no split ranges.
constexpr int64_t kTinyMaxBodies = 64;
// (integrator KDK: a.kdk_acc carries the sums between calls; kdk_prime: evaluate them first)
int tiny_run(const nbody_mi355x_step_args& a, int64_t substeps, bool kdk_prime, struct ihipStream_t* stream);
// nbody_multi.hip: the handle over several shards / GPUs (host records already double)
struct MultiSim;
int multi_create(const nbody_mi355x_config& cfg, const int32_t* devices, int32_t num, MultiSim** out);
int multi_destroy(MultiSim* m);
int multi_upload(MultiSim* m, const double* particles, const double* mass);
int multi_generate(MultiSim* m, const nbody_mi355x_scene& scene);
bool scene_valid(const nbody_mi355x_scene* scene, const char* who);  // nbody_scene.hip; sets the error text
int multi_enqueue(MultiSim* m, int64_t substeps, double* snapshot_out);
int multi_wait(MultiSim* m, double* seconds);
int multi_download(MultiSim* m, double* particles);
int multi_exchange_seconds(MultiSim* m, double* positions, double* sums);
bool multi_uses_rccl(const MultiSim* m);
int multi_energy(MultiSim* m, double* ke, double* pe);
int multi_kernel_in_use(MultiSim* m, int32_t* variant, int64_t* workspace_bytes);
}  // namespace nbody_internal
