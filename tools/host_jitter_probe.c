// host_jitter_probe.c -- is it the HOST that makes a paced pump late?  No GPU in here.
//
//   gcc -O2 -pthread -o tools/_scratch/host_jitter_probe tools/host_jitter_probe.c && tools/_scratch/host_jitter_probe [seconds] [threads] [mode]
//
// `threads` threads each do what an idle rcf_pump thread does between group blocks: sleep <= 1 ms (clock_nanosleep), wake,
// ~20 us of work, again.  Per thread: how late the wake-ups were (overshoot p50 / p99 / max, count > 2 ms) and, from
// /proc/thread-self/schedstat, how long the thread sat RUNNABLE on a run queue without a CPU (run_delay) -- in total and
// inside the late wake-ups.  A late wake-up whose lateness is run_delay is the host scheduler (CPUs busy with someone
// else's threads: the boxes are shared, the container has a CFS quota but no CPUs of its own and no SCHED_FIFO).
// mode: 0 = threads float over the process's affinity mask; 1 = each pinned to one CPU of the mask (in order);
//       2 = each pinned to one of the IDLEST CPUs of the mask (/proc/stat sampled for 300 ms first).
// Prints one JSON object.
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static long long run_delay_ns(int fd)
{
    char b[128];
    ssize_t n = pread(fd, b, sizeof b - 1, 0);
    if (n <= 0) return -1;
    b[n] = 0;
    long long run = 0, delay = 0;
    if (sscanf(b, "%lld %lld", &run, &delay) != 2) return -1;
    return delay;
}

typedef struct {
    int cpu;
    double seconds;
    int n;
    double *over_ms;
    int late2;
    double delay_total_ms, delay_in_late_ms, late_total_ms, max_over_ms, max_delay_single_ms;
    long nivcsw;
} Arg;

static void *worker(void *a_)
{
    Arg *a = (Arg *)a_;
    if (a->cpu >= 0) {
        cpu_set_t s;
        CPU_ZERO(&s);
        CPU_SET(a->cpu, &s);
        pthread_setaffinity_np(pthread_self(), sizeof s, &s);
    }
    int fd = open("/proc/thread-self/schedstat", O_RDONLY);
    const int cap = (int)(a->seconds * 1200) + 16;
    a->over_ms = (double *)calloc((size_t)cap, sizeof(double));
    const double t_end = now_s() + a->seconds;
    const long long d_start = fd >= 0 ? run_delay_ns(fd) : -1;
    volatile double sink = 0;
    while (now_s() < t_end && a->n < cap) {
        const long long d0 = fd >= 0 ? run_delay_ns(fd) : -1;
        const double t0 = now_s();
        struct timespec ts = {0, 900000};
        clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, NULL);
        const double over = (now_s() - t0 - 0.0009) * 1e3;
        a->over_ms[a->n++] = over;
        if (over > a->max_over_ms) a->max_over_ms = over;
        if (over > 2.0) {
            ++a->late2;
            a->late_total_ms += over;
            const long long d1 = fd >= 0 ? run_delay_ns(fd) : -1;
            if (d0 >= 0 && d1 >= 0) {
                const double dm = (d1 - d0) * 1e-6;
                a->delay_in_late_ms += dm;
                if (dm > a->max_delay_single_ms) a->max_delay_single_ms = dm;
            }
        }
        const double w0 = now_s();
        while (now_s() - w0 < 20e-6) sink += 1.0;          // the planning of a group block, roughly
    }
    const long long d_end = fd >= 0 ? run_delay_ns(fd) : -1;
    a->delay_total_ms = (d_start >= 0 && d_end >= 0) ? (d_end - d_start) * 1e-6 : -1.0;
    if (fd >= 0) close(fd);
    return NULL;
}

static int cmp_d(const void *x, const void *y)
{
    const double a = *(const double *)x, b = *(const double *)y;
    return a < b ? -1 : a > b;
}

// idle jiffies per CPU from /proc/stat
static void read_idle(long long *idle, long long *total, int ncpu)
{
    FILE *f = fopen("/proc/stat", "r");
    char line[512];
    for (int i = 0; i < ncpu; ++i) idle[i] = total[i] = -1;
    while (f && fgets(line, sizeof line, f)) {
        int c;
        long long v[8] = {0};
        if (sscanf(line, "cpu%d %lld %lld %lld %lld %lld %lld %lld %lld", &c, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) >= 5 &&
            c >= 0 && c < ncpu) {
            idle[c] = v[3] + v[4];
            total[c] = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
        }
    }
    if (f) fclose(f);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int nthr = argc > 2 ? atoi(argv[2]) : 4;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    cpu_set_t mask;
    sched_getaffinity(0, sizeof mask, &mask);
    const int ncpu = (int)sysconf(_SC_NPROCESSORS_CONF);
    int allowed[4096], n_allowed = 0;
    for (int c = 0; c < ncpu && c < 4096; ++c)
        if (CPU_ISSET(c, &mask)) allowed[n_allowed++] = c;
    // how busy are the allowed CPUs (everyone's load, not only this container's)?
    long long *i0 = calloc((size_t)ncpu, sizeof *i0), *t0 = calloc((size_t)ncpu, sizeof *t0), *i1 = calloc((size_t)ncpu, sizeof *i1), *t1 = calloc((size_t)ncpu, sizeof *t1);
    read_idle(i0, t0, ncpu);
    usleep(300000);
    read_idle(i1, t1, ncpu);
    double busy_sum = 0;
    int busy_over_50 = 0;
    double *busy = calloc((size_t)ncpu, sizeof *busy);
    for (int k = 0; k < n_allowed; ++k) {
        const int c = allowed[k];
        const double tot = (double)(t1[c] - t0[c]);
        busy[c] = tot > 0 ? 1.0 - (double)(i1[c] - i0[c]) / tot : 0.0;
        busy_sum += busy[c];
        busy_over_50 += busy[c] > 0.5;
    }
    // mode 2: the idlest CPUs first
    int order[4096];
    for (int k = 0; k < n_allowed; ++k) order[k] = allowed[k];
    if (mode == 2)
        for (int a = 0; a < n_allowed; ++a)
            for (int b = a + 1; b < n_allowed; ++b)
                if (busy[order[b]] < busy[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    pthread_t th[256];
    Arg args[256];
    memset(args, 0, sizeof args);
    for (int i = 0; i < nthr && i < 256; ++i) {
        args[i].cpu = mode == 0 ? -1 : order[i % n_allowed];
        args[i].seconds = seconds;
        pthread_create(&th[i], NULL, worker, &args[i]);
    }
    for (int i = 0; i < nthr && i < 256; ++i) pthread_join(th[i], NULL);
    printf("{\"mode\": %d, \"threads\": %d, \"seconds\": %.1f, \"cpus_allowed\": %d, \"allowed_cpus_mean_busy\": %.3f, "
           "\"allowed_cpus_over_50pct_busy\": %d, \"per_thread\": [", mode, nthr, seconds, n_allowed, busy_sum / (n_allowed ? n_allowed : 1), busy_over_50);
    for (int i = 0; i < nthr && i < 256; ++i) {
        Arg *a = &args[i];
        qsort(a->over_ms, (size_t)a->n, sizeof(double), cmp_d);
        printf("%s{\"cpu\": %d, \"wakeups\": %d, \"overshoot_ms_p50\": %.3f, \"overshoot_ms_p99\": %.3f, \"overshoot_ms_max\": %.3f, "
               "\"late_over_2ms\": %d, \"late_total_ms\": %.2f, \"run_queue_delay_in_late_wakeups_ms\": %.2f, "
               "\"run_queue_delay_longest_single_ms\": %.2f, \"run_queue_delay_total_ms\": %.2f}",
               i ? ", " : "", a->cpu, a->n, a->n ? a->over_ms[a->n / 2] : 0.0, a->n ? a->over_ms[(int)(a->n * 0.99)] : 0.0, a->max_over_ms,
               a->late2, a->late_total_ms, a->delay_in_late_ms, a->max_delay_single_ms, a->delay_total_ms);
    }
    printf("]}\n");
    return 0;
}
