// Shared between the translation units of libnbody_mi355x.so; not installed.
#pragma once
#include <cstdarg>

#include "nbody_mi355x.h"

namespace nbody_internal {
// Text behind nbody_mi355x_last_error() (thread-local).
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// nbody_kernels.hip: host-side shape checks of a step's arguments -- everything the kernels'
// indexing assumes, verified before any launch
int check_step_args(const nbody_mi355x_step_args* a);
// nbody_sym.hip: pair-symmetric step (args already validated by check_step_args)
bool sym_supported(const nbody_mi355x_step_args& a, int64_t* preferred_bytes, int64_t* least_bytes);
int sym_step(const nbody_mi355x_step_args& a, struct ihipStream_t* stream);
// nbody_kernels.hip: `substeps` whole small steps of a system of at most kTinyMaxBodies bodies
// in ONE launch of one wave (in place on a.pos_cur / a.vel; a.snapshot, if set, receives the
// last small step's records).  One slab, FP64, no split ranges.
constexpr int64_t kTinyMaxBodies = 64;
int tiny_run(const nbody_mi355x_step_args& a, int64_t substeps, struct ihipStream_t* stream);
// nbody_multi.hip: the handle over several shards / GPUs (host records already double)
struct MultiSim;
int multi_create(const nbody_mi355x_config& cfg, const int32_t* devices, int32_t num, MultiSim** out);
int multi_destroy(MultiSim* m);
int multi_upload(MultiSim* m, const double* particles, const double* mass);
int multi_enqueue(MultiSim* m, int64_t substeps, double* snapshot_out);
int multi_wait(MultiSim* m, double* seconds);
int multi_download(MultiSim* m, double* particles);
int multi_energy(MultiSim* m, double* ke, double* pe);
}  // namespace nbody_internal
