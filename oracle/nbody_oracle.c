/*
 * nbody_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference's all-pairs force + forward
 * Euler hot path.  It exists so that the HIP path can be checked against the
 * reference's arithmetic on a machine where /root/reference does not exist
 * (the GPU box).  Nothing in the product (n-body-sim_amd/, include/) may link,
 * load or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against the .npz files under tests/golden/, which were produced by the real
 * reference header compiled in the build container (oracle/ref_harness.cpp ->
 * oracle/_ref/libnbody_ref.so, generator tests/golden/make_golden.py).
 *
 * Build with the reference's own flags (README.md:38 of the reference):
 *   gcc -O3 -mavx2 -mfma -pthread -shared -fPIC     (see oracle/Makefile)
 * so that GCC's floating-point contraction decisions are the same as in the
 * reference build the golden vectors came from.
 *
 * Citations are file:line into /root/reference/src/.
 */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* particle_simulation.h:713 (and particle_simulation_GPU.cu:75): the
 * simulator's constant is 6.7430e-11, NOT 6.67430e-11. */
#define NBODY_ORACLE_G 6.7430e-11

/* ---------------------------------------------------------------- partition
 * particle_simulation.h:98-110: N/threads each, remainder to the low ids. */
static void split_range(long n, int parts, int id, long *begin, long *count) {
  long base = n / parts, rem = n % parts;
  *count = base + (id < rem ? 1 : 0);
  *begin = base * id + (id < rem ? id : rem);
}

/* ------------------------------------------------------------ scalar, f64
 * particle_simulation.h:248-322 MakeSmallStep.  cur/next are Particle<double>
 * AoS records {r[3], v[3]} (particle_simulation.h:20-24).  out9, when not
 * NULL, receives {r', v', a*G} per particle, the DeviceOutputParticle layout
 * (particle_simulation.h:26-31); m and t of OutPutParticle are O(1) host
 * bookkeeping (particle_simulation.h:317-319) and are left to the caller. */
static void small_step_f64(long n, const double *cur, double *next,
                           const double *mass, double dt, long i_begin,
                           long i_count, double *out9) {
  const double G_ = NBODY_ORACLE_G;
  for (long i = i_begin; i < i_begin + i_count; i++) {
    double a[3] = {0, 0, 0};
    for (long j = 0; j < n; j++) {                      /* :256 */
      double delta_x = cur[6 * j + 0] - cur[6 * i + 0]; /* :257-262 */
      double delta_y = cur[6 * j + 1] - cur[6 * i + 1];
      double delta_z = cur[6 * j + 2] - cur[6 * i + 2];
      double r_squared = (delta_x * delta_x) + (delta_y * delta_y) +
                         (delta_z * delta_z);           /* :263-265 */
      if (r_squared != 0) {                             /* :269 */
        double tmp = mass[j] / (r_squared * sqrt(r_squared)); /* :270 */
        a[0] += (tmp * delta_x);                        /* :271-273 */
        a[1] += (tmp * delta_y);
        a[2] += (tmp * delta_z);
      }
    }
    /* :288-293  v' = v + (dt * a * G), left to right */
    next[6 * i + 3] = cur[6 * i + 3] + (dt * a[0] * G_);
    next[6 * i + 4] = cur[6 * i + 4] + (dt * a[1] * G_);
    next[6 * i + 5] = cur[6 * i + 5] + (dt * a[2] * G_);
    /* :295-300  r' = r + dt * v_old */
    next[6 * i + 0] = cur[6 * i + 0] + (dt * cur[6 * i + 3]);
    next[6 * i + 1] = cur[6 * i + 1] + (dt * cur[6 * i + 4]);
    next[6 * i + 2] = cur[6 * i + 2] + (dt * cur[6 * i + 5]);
    if (out9) {                                         /* :304-315 */
      double *o = out9 + 9 * i;
      o[0] = next[6 * i + 0]; o[1] = next[6 * i + 1]; o[2] = next[6 * i + 2];
      o[3] = next[6 * i + 3]; o[4] = next[6 * i + 4]; o[5] = next[6 * i + 5];
      o[6] = a[0] * G_; o[7] = a[1] * G_; o[8] = a[2] * G_;
    }
  }
}

/* ------------------------------------------------------------ scalar, f32
 * The same template instantiated with float_t = float
 * (ParticleSimulation<float,8>::MakeSmallStep).  Two details of that
 * instantiation matter for the last bits:
 *  - G_ is `static constexpr float_t` (particle_simulation.h:713): a float;
 *  - `sqrt(r_squared)` at :270 is the unqualified call; which overload it
 *    binds to is decided by NBODY_ORACLE_F32_SQRT_DOUBLE below, set to what
 *    the golden vector G6 (real reference, g++ 11.4) shows. */
#ifndef NBODY_ORACLE_F32_SQRT_DOUBLE
#define NBODY_ORACLE_F32_SQRT_DOUBLE 0
#endif
static void small_step_f32(long n, const float *cur, float *next,
                           const float *mass, float dt, long i_begin,
                           long i_count, float *out9) {
  const float G_ = (float)NBODY_ORACLE_G;
  for (long i = i_begin; i < i_begin + i_count; i++) {
    float a[3] = {0, 0, 0};
    for (long j = 0; j < n; j++) {
      float delta_x = cur[6 * j + 0] - cur[6 * i + 0];
      float delta_y = cur[6 * j + 1] - cur[6 * i + 1];
      float delta_z = cur[6 * j + 2] - cur[6 * i + 2];
      float r_squared = (delta_x * delta_x) + (delta_y * delta_y) +
                        (delta_z * delta_z);
      if (r_squared != 0) {
#if NBODY_ORACLE_F32_SQRT_DOUBLE
        float tmp = mass[j] / (r_squared * sqrt((double)r_squared));
#else
        float tmp = mass[j] / (r_squared * sqrtf(r_squared));
#endif
        a[0] += (tmp * delta_x);
        a[1] += (tmp * delta_y);
        a[2] += (tmp * delta_z);
      }
    }
    next[6 * i + 3] = cur[6 * i + 3] + (dt * a[0] * G_);
    next[6 * i + 4] = cur[6 * i + 4] + (dt * a[1] * G_);
    next[6 * i + 5] = cur[6 * i + 5] + (dt * a[2] * G_);
    next[6 * i + 0] = cur[6 * i + 0] + (dt * cur[6 * i + 3]);
    next[6 * i + 1] = cur[6 * i + 1] + (dt * cur[6 * i + 4]);
    next[6 * i + 2] = cur[6 * i + 2] + (dt * cur[6 * i + 5]);
    if (out9) {
      float *o = out9 + 9 * i;
      o[0] = next[6 * i + 0]; o[1] = next[6 * i + 1]; o[2] = next[6 * i + 2];
      o[3] = next[6 * i + 3]; o[4] = next[6 * i + 4]; o[5] = next[6 * i + 5];
      o[6] = a[0] * G_; o[7] = a[1] * G_; o[8] = a[2] * G_;
    }
  }
}

/* -------------------------------------------------------------- AVX2, f64
 * particle_simulation.h:382-544 MakeSmallStepAVX2 over the AoSoA layout
 * ParticleBatch<double,4> = {r_x[4] r_y[4] r_z[4] v_x[4] v_y[4] v_z[4]}
 * (particle_simulation.h:45-53), 24 doubles per batch, 32-byte aligned, and
 * ParticleMassBatch<double,4> = {m[4]} (:55-58).
 * One deliberate difference: the reference writes out_put_data_ for the
 * padding lanes of the last batch too (:514-540), which is out of bounds when
 * N % 4 != 0; here those lanes are computed but not written to out9. */
static void small_step_avx2(long n, long num_batches, const double *cur,
                            double *next, const double *bmass, double dt,
                            long b_begin, long b_count, double *out9) {
  const double G_ = NBODY_ORACLE_G;
  for (long b = b_begin; b < b_begin + b_count; b++) {
    const double *cb = cur + 24 * b;
    double *nb = next + 24 * b;
    for (int k = 0; k < 4; k++) {                            /* :389 */
      __m256d current_x = _mm256_set1_pd(cb[0 + k]);         /* :392-397 */
      __m256d current_y = _mm256_set1_pd(cb[4 + k]);
      __m256d current_z = _mm256_set1_pd(cb[8 + k]);
      __m256d acc_x = _mm256_setzero_pd();
      __m256d acc_y = _mm256_setzero_pd();
      __m256d acc_z = _mm256_setzero_pd();
      __m256d zeros = _mm256_setzero_pd();
      for (long i = 0; i < num_batches; i++) {               /* :405 */
        const double *ob = cur + 24 * i;
        __m256d other_x = _mm256_load_pd(ob + 0);            /* :408-414 */
        __m256d other_y = _mm256_load_pd(ob + 4);
        __m256d other_z = _mm256_load_pd(ob + 8);
        __m256d other_mass = _mm256_load_pd(bmass + 4 * i);
        __m256d delta_x = _mm256_sub_pd(other_x, current_x); /* :417-419 */
        __m256d delta_y = _mm256_sub_pd(other_y, current_y);
        __m256d delta_z = _mm256_sub_pd(other_z, current_z);
        __m256d r_squared = _mm256_fmadd_pd(                 /* :423-425 */
            delta_z, delta_z,
            _mm256_fmadd_pd(delta_y, delta_y, _mm256_mul_pd(delta_x, delta_x)));
        __m256d mask = _mm256_cmp_pd(r_squared, zeros, _CMP_GT_OQ); /* :427 */
        if (_mm256_movemask_pd(mask) == 0xf) {               /* :429 */
          __m256d tmp = _mm256_div_pd(                       /* :432-433 */
              other_mass, _mm256_mul_pd(r_squared, _mm256_sqrt_pd(r_squared)));
          acc_x = _mm256_fmadd_pd(tmp, delta_x, acc_x);      /* :434-436 */
          acc_y = _mm256_fmadd_pd(tmp, delta_y, acc_y);
          acc_z = _mm256_fmadd_pd(tmp, delta_z, acc_z);
        } else {                                             /* :437-461 */
          double r2[4], om[4], dx[4], dy[4], dz[4], ax[4], ay[4], az[4];
          _mm256_storeu_pd(r2, r_squared); _mm256_storeu_pd(om, other_mass);
          _mm256_storeu_pd(dx, delta_x); _mm256_storeu_pd(dy, delta_y);
          _mm256_storeu_pd(dz, delta_z); _mm256_storeu_pd(ax, acc_x);
          _mm256_storeu_pd(ay, acc_y); _mm256_storeu_pd(az, acc_z);
          for (int j = 0; j < 4; j++) {
            if (r2[j] != 0) {
              double tmp1 = om[j] / (r2[j] * sqrt(r2[j]));
              ax[j] = tmp1 * dx[j] + ax[j];
              ay[j] = tmp1 * dy[j] + ay[j];
              az[j] = tmp1 * dz[j] + az[j];
            }
          }
          acc_x = _mm256_loadu_pd(ax); acc_y = _mm256_loadu_pd(ay);
          acc_z = _mm256_loadu_pd(az);
        }
      }
      /* :466-475  hadd, then lane0 + lane2, then * G */
      double fa[3], t[4];
      acc_x = _mm256_hadd_pd(acc_x, acc_x); _mm256_storeu_pd(t, acc_x);
      fa[0] = G_ * (t[0] + t[2]);
      acc_y = _mm256_hadd_pd(acc_y, acc_y); _mm256_storeu_pd(t, acc_y);
      fa[1] = G_ * (t[0] + t[2]);
      acc_z = _mm256_hadd_pd(acc_z, acc_z); _mm256_storeu_pd(t, acc_z);
      fa[2] = G_ * (t[0] + t[2]);
      nb[12 + k] = cb[12 + k] + (dt * fa[0]);                /* :488-496 */
      nb[16 + k] = cb[16 + k] + (dt * fa[1]);
      nb[20 + k] = cb[20 + k] + (dt * fa[2]);
      nb[0 + k] = cb[0 + k] + (dt * cb[12 + k]);             /* :498-509 */
      nb[4 + k] = cb[4 + k] + (dt * cb[16 + k]);
      nb[8 + k] = cb[8 + k] + (dt * cb[20 + k]);
      long pid = 4 * b + k;                                  /* :514 */
      if (out9 && pid < n) {
        double *o = out9 + 9 * pid;
        o[0] = nb[0 + k]; o[1] = nb[4 + k]; o[2] = nb[8 + k];
        o[3] = nb[12 + k]; o[4] = nb[16 + k]; o[5] = nb[20 + k];
        o[6] = fa[0]; o[7] = fa[1]; o[8] = fa[2];
      }
    }
  }
}

/* ------------------------------------------------------------------ driver
 * particle_simulation.h:197-240 / :329-374: big-step loop, substeps-1 silent
 * small steps + 1 snapshotting one, current/next swap after every small step.
 * The reference spawns fresh std::threads per small step (:223-232); a
 * persistent pool with a barrier gives the same values (disjoint i ranges,
 * read-only current, write-only next, :234-239), only faster. */
enum { K_F64 = 0, K_F32 = 1, K_AVX2 = 2 };

typedef struct {
  int kind, threads;
  long n, units; /* units = particles, or batches for AVX2 */
  long big_steps, substeps;
  double dt;
  void *buf[2];
  const void *mass;
  void *out9; /* [big_steps][n][9] or NULL */
  pthread_barrier_t bar;
} job_t;

typedef struct { job_t *job; int id; } worker_t;

static void *worker_main(void *arg) {
  worker_t *w = (worker_t *)arg;
  job_t *J = w->job;
  long begin, count;
  split_range(J->units, J->threads, w->id, &begin, &count);
  int cur = 0;
  for (long big = 0; big < J->big_steps; big++) {
    for (long s = 0; s < J->substeps; s++) {
      int last = (s == J->substeps - 1);
      if (J->kind == K_F64) {
        double *o = (last && J->out9) ? (double *)J->out9 + big * J->n * 9 : NULL;
        small_step_f64(J->n, (const double *)J->buf[cur], (double *)J->buf[1 - cur],
                       (const double *)J->mass, J->dt, begin, count, o);
      } else if (J->kind == K_F32) {
        float *o = (last && J->out9) ? (float *)J->out9 + big * J->n * 9 : NULL;
        small_step_f32(J->n, (const float *)J->buf[cur], (float *)J->buf[1 - cur],
                       (const float *)J->mass, (float)J->dt, begin, count, o);
      } else {
        double *o = (last && J->out9) ? (double *)J->out9 + big * J->n * 9 : NULL;
        small_step_avx2(J->n, J->units, (const double *)J->buf[cur],
                        (double *)J->buf[1 - cur], (const double *)J->mass,
                        J->dt, begin, count, o);
      }
      pthread_barrier_wait(&J->bar);
      cur = 1 - cur;
    }
  }
  return NULL;
}

static int run_job(job_t *J) {
  if (J->threads < 1) J->threads = 1;
  if (J->threads > 1024) J->threads = 1024;
  pthread_barrier_init(&J->bar, NULL, (unsigned)J->threads);
  pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)J->threads);
  worker_t *ws = (worker_t *)malloc(sizeof(worker_t) * (size_t)J->threads);
  for (int i = 0; i < J->threads; i++) { ws[i].job = J; ws[i].id = i; }
  for (int i = 1; i < J->threads; i++) pthread_create(&tid[i], NULL, worker_main, &ws[i]);
  worker_main(&ws[0]);
  for (int i = 1; i < J->threads; i++) pthread_join(tid[i], NULL);
  pthread_barrier_destroy(&J->bar);
  free(tid); free(ws);
  return 0;
}

/* SimulateCPU() restated (particle_simulation.h:197-212).  particles: in =
 * initial Particle<double>[n], out = state after big_steps*substeps small
 * steps.  out9: DeviceOutputParticle<double>[big_steps][n] or NULL. */
int nbody_oracle_simulate_f64(long n, long big_steps, long substeps, double dt,
                              int threads, double *particles,
                              const double *mass, double *out9) {
  if (n <= 0 || big_steps < 0 || substeps <= 0) return -1;
  job_t J; memset(&J, 0, sizeof J);
  J.kind = K_F64; J.threads = threads; J.n = n; J.units = n;
  J.big_steps = big_steps; J.substeps = substeps; J.dt = dt; J.mass = mass;
  J.out9 = out9;
  double *other = (double *)malloc(sizeof(double) * 6 * (size_t)n);
  if (!other) return -2;
  J.buf[0] = particles; J.buf[1] = other;
  run_job(&J);
  if ((big_steps * substeps) & 1) memcpy(particles, other, sizeof(double) * 6 * (size_t)n);
  free(other);
  return 0;
}

/* ParticleSimulation<float,W>::SimulateCPU() restated. */
int nbody_oracle_simulate_f32(long n, long big_steps, long substeps, float dt,
                              int threads, float *particles, const float *mass,
                              float *out9) {
  if (n <= 0 || big_steps < 0 || substeps <= 0) return -1;
  job_t J; memset(&J, 0, sizeof J);
  J.kind = K_F32; J.threads = threads; J.n = n; J.units = n;
  J.big_steps = big_steps; J.substeps = substeps; J.dt = dt; J.mass = mass;
  J.out9 = out9;
  float *other = (float *)malloc(sizeof(float) * 6 * (size_t)n);
  if (!other) return -2;
  J.buf[0] = particles; J.buf[1] = other;
  run_job(&J);
  if ((big_steps * substeps) & 1) memcpy(particles, other, sizeof(float) * 6 * (size_t)n);
  free(other);
  return 0;
}

/* SimulateAVX2() restated (particle_simulation.h:329-374) for <double,4>.
 * Takes and returns the flat Particle<double>[n] records; the AoSoA batches of
 * SetParticle (particle_simulation.h:158-171) are built here, padding lanes of
 * the last batch zero (make_unique value-initialises them, :129-134). */
int nbody_oracle_simulate_avx2_f64(long n, long big_steps, long substeps,
                                   double dt, int threads, double *particles,
                                   const double *mass, double *out9) {
  if (n <= 0 || big_steps < 0 || substeps <= 0) return -1;
  long nb = (n + 3) / 4; /* :93-94 */
  double *b0 = NULL, *b1 = NULL, *bm = NULL;
  if (posix_memalign((void **)&b0, 32, sizeof(double) * 24 * (size_t)nb)) return -2;
  if (posix_memalign((void **)&b1, 32, sizeof(double) * 24 * (size_t)nb)) return -2;
  if (posix_memalign((void **)&bm, 32, sizeof(double) * 4 * (size_t)nb)) return -2;
  memset(b0, 0, sizeof(double) * 24 * (size_t)nb);
  memset(b1, 0, sizeof(double) * 24 * (size_t)nb);
  memset(bm, 0, sizeof(double) * 4 * (size_t)nb);
  for (long p = 0; p < n; p++) {
    long b = p / 4; int k = (int)(p % 4);
    for (int c = 0; c < 6; c++) b0[24 * b + 4 * c + k] = particles[6 * p + c];
    bm[4 * b + k] = mass[p];
  }
  job_t J; memset(&J, 0, sizeof J);
  J.kind = K_AVX2; J.threads = threads; J.n = n; J.units = nb;
  J.big_steps = big_steps; J.substeps = substeps; J.dt = dt; J.mass = bm;
  J.out9 = out9; J.buf[0] = b0; J.buf[1] = b1;
  run_job(&J);
  const double *fin = ((big_steps * substeps) & 1) ? b1 : b0;
  for (long p = 0; p < n; p++) {
    long b = p / 4; int k = (int)(p % 4);
    for (int c = 0; c < 6; c++) particles[6 * p + c] = fin[24 * b + 4 * c + k];
  }
  free(b0); free(b1); free(bm);
  return 0;
}

/* Sampled-row force check for sizes where a full step is too slow on a CPU:
 * a*G (particle_simulation.h:256-275, :313) for the `rows` listed particles
 * against all n bodies.  acc3: [rows][3]. */
typedef struct { long n; const double *p, *m; const long *idx; long begin, count; double *acc; } rows_t;
static void *rows_main(void *arg) {
  rows_t *R = (rows_t *)arg;
  for (long q = R->begin; q < R->begin + R->count; q++) {
    long i = R->idx[q];
    double a[3] = {0, 0, 0};
    for (long j = 0; j < R->n; j++) {
      double delta_x = R->p[6 * j + 0] - R->p[6 * i + 0];
      double delta_y = R->p[6 * j + 1] - R->p[6 * i + 1];
      double delta_z = R->p[6 * j + 2] - R->p[6 * i + 2];
      double r_squared = (delta_x * delta_x) + (delta_y * delta_y) + (delta_z * delta_z);
      if (r_squared != 0) {
        double tmp = R->m[j] / (r_squared * sqrt(r_squared));
        a[0] += (tmp * delta_x); a[1] += (tmp * delta_y); a[2] += (tmp * delta_z);
      }
    }
    R->acc[3 * q + 0] = a[0] * NBODY_ORACLE_G;
    R->acc[3 * q + 1] = a[1] * NBODY_ORACLE_G;
    R->acc[3 * q + 2] = a[2] * NBODY_ORACLE_G;
  }
  return NULL;
}
int nbody_oracle_accel_rows_f64(long n, const double *particles, const double *mass,
                                const long *rows, long nrows, int threads, double *acc3) {
  if (threads < 1) threads = 1;
  if (threads > 1024) threads = 1024;
  pthread_t tid[1024]; rows_t R[1024];
  for (int t = 0; t < threads; t++) {
    R[t].n = n; R[t].p = particles; R[t].m = mass; R[t].idx = rows; R[t].acc = acc3;
    split_range(nrows, threads, t, &R[t].begin, &R[t].count);
    if (t) pthread_create(&tid[t], NULL, rows_main, &R[t]);
  }
  rows_main(&R[0]);
  for (int t = 1; t < threads; t++) pthread_join(tid[t], NULL);
  return 0;
}

/* Energy diagnostic used on BOTH the oracle's and the GPU's snapshots (the
 * reference has none): KE = sum 1/2 m v^2, PE = -G sum_{i<j} m_i m_j / r_ij,
 * Kahan-compensated, coincident pairs skipped like the force does (:269).
 * state9 stride lets it read Particle (6) or DeviceOutputParticle (9) rows. */
typedef struct { long n, stride; const double *s, *m; long begin, count; double pe; } en_t;
static void *energy_main(void *arg) {
  en_t *E = (en_t *)arg;
  double sum = 0, c = 0;
  for (long i = E->begin; i < E->begin + E->count; i++) {
    const double *pi = E->s + E->stride * i;
    double rowsum = 0, rc = 0;
    for (long j = i + 1; j < E->n; j++) {
      const double *pj = E->s + E->stride * j;
      double dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];
      double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 != 0) {
        double term = E->m[j] / sqrt(r2);
        double y = term - rc, t = rowsum + y; rc = (t - rowsum) - y; rowsum = t;
      }
    }
    double term = -NBODY_ORACLE_G * E->m[i] * rowsum;
    double y = term - c, t = sum + y; c = (t - sum) - y; sum = t;
  }
  E->pe = sum;
  return NULL;
}
int nbody_oracle_energy_f64(long n, const double *state, long stride,
                            const double *mass, int threads, double *ke, double *pe) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  double k = 0, kc = 0;
  for (long i = 0; i < n; i++) {
    const double *p = state + stride * i;
    double term = 0.5 * mass[i] * (p[3] * p[3] + p[4] * p[4] + p[5] * p[5]);
    double y = term - kc, t = k + y; kc = (t - k) - y; k = t;
  }
  *ke = k;
  pthread_t tid[256]; en_t E[256];
  /* interleave rows so the triangular work is balanced: thread t takes the
   * row blocks by equal pair count (sqrt split) */
  for (int t = 0; t < threads; t++) {
    double f0 = 1.0 - sqrt(1.0 - (double)t / threads);
    double f1 = 1.0 - sqrt(1.0 - (double)(t + 1) / threads);
    long b = (long)(f0 * (double)n), e = (t == threads - 1) ? n : (long)(f1 * (double)n);
    E[t].n = n; E[t].stride = stride; E[t].s = state; E[t].m = mass;
    E[t].begin = b; E[t].count = e - b; E[t].pe = 0;
    if (t) pthread_create(&tid[t], NULL, energy_main, &E[t]);
  }
  energy_main(&E[0]);
  double p = 0;
  for (int t = 1; t < threads; t++) pthread_join(tid[t], NULL);
  for (int t = 0; t < threads; t++) p += E[t].pe;
  *pe = p;
  return 0;
}

/* One small step for the i range [i_begin, i_begin+i_count) only: exactly what
 * one reference worker thread does (particle_simulation.h:250-253).  Used by
 * the world_size-2 gloo test of the sharded driver, where each rank plays one
 * "thread" and the position exchange replaces shared memory. */
int nbody_oracle_small_step_f64(long n, const double *cur, double *next,
                                const double *mass, double dt, long i_begin,
                                long i_count, double *out9) {
  if (n <= 0 || i_begin < 0 || i_count < 0 || i_begin + i_count > n) return -1;
  small_step_f64(n, cur, next, mass, dt, i_begin, i_count, out9);
  return 0;
}
