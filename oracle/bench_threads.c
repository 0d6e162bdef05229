/* bench_threads.c — the CPU baseline's driver (test infrastructure, like the rest of oracle/): the oracle's whole path
 * (ora_process_frame: threshold -> segment -> clusters -> quad fit -> decode -> glue -> SQPnP -> 64-byte record, the restatement of
 * crates/apriltags/src/lib.rs:293-379) over a set of frames on T POSIX threads, timed here so that no interpreter sits in the
 * measured region.  Frames are handed out through one atomic counter.
 *
 * The oracle allocates its per-frame arrays (several W x H words) with malloc/calloc.  With glibc's defaults every such block
 * is an mmap of its own that is unmapped again at free(): at 256 threads the process-wide address-space lock and the page
 * faults of freshly mapped zero pages cost more than the arithmetic (round 2's 256-thread figure was BELOW its 16-thread
 * one).  ora_bench_process therefore raises the mmap and trim thresholds first: the blocks then come from the threads' own
 * arenas, which grow once and are reused frame after frame — per-thread reusable scratch without touching the code under test. */
#include <malloc.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ck_oracle.h"

typedef struct {
    const uint8_t *frames; int w, h, stride; size_t pitch; int n_frames, n_work;
    const ck_config_t *cfg; const ck_process_params_t *pp; const double *gyro;
    atomic_int next; atomic_int valid;
} bench_job_t;

static void *bench_worker(void *arg) {
    bench_job_t *j = (bench_job_t *)arg;
    int nvalid = 0;
    for (;;) {
        const int i = atomic_fetch_add(&j->next, 1);
        if (i >= j->n_work) break;
        const int f = i % j->n_frames;
        ck_vision_measurement_t out;
        int v = 0;
        ora_process_frame(j->frames + (size_t)f * j->pitch, j->w, j->h, j->stride, j->cfg, j->pp, j->gyro[f], 1, &out, &v);
        nvalid += v ? 1 : 0;
    }
    atomic_fetch_add(&j->valid, nvalid);
    return NULL;
}

/* n_work frames (cycling through the n_frames given) on `threads` threads; returns the wall seconds of the threaded region
 * (negative on failure) and the number of frames that produced a pose. */
double ora_bench_process(const uint8_t *frames, int w, int h, int stride, size_t frame_pitch, int n_frames, int n_work,
                         const ck_config_t *cfg, const ck_process_params_t *pp, const double *gyro, int threads, int *n_valid) {
    if (!frames || !cfg || !pp || !gyro || n_frames < 1 || n_work < 1 || threads < 1) return -1.0;
    static int tuned = 0;
    if (!tuned) { /* (process-wide, once) */
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
        tuned = 1;
    }
    bench_job_t job;
    memset(&job, 0, sizeof job);
    job.frames = frames; job.w = w; job.h = h; job.stride = stride; job.pitch = frame_pitch; job.n_frames = n_frames; job.n_work = n_work;
    job.cfg = cfg; job.pp = pp; job.gyro = gyro;
    atomic_init(&job.next, 0); atomic_init(&job.valid, 0);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    if (!th) return -1.0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, bench_worker, &job) != 0) break;
    for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    if (started == 0) return -1.0;
    if (n_valid) *n_valid = atomic_load(&job.valid);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
