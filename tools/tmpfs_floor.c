// tmpfs_floor.c — what does it cost to get 11.1 GB into ONE tmpfs file on this box, by which route?  (before
// building D2H-into-a-registered-mmap, measure the floor that route would have.)  Tuning aid only.
//   gcc -O2 -pthread -o tools/tmpfs_floor tools/tmpfs_floor.c ; tools/tmpfs_floor /dev/shm/x.bin 11097281230
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static const char *path;
static uint64_t total;
static const size_t CH = 128u << 20;

struct job { uint8_t *map; uint64_t lo, hi; int mode; int fd; const uint8_t *src; };

static void *worker(void *p)
{
    struct job *j = p;
    for (uint64_t o = j->lo; o < j->hi; o += CH) {
        uint64_t n = j->hi - o < CH ? j->hi - o : CH;
        if (j->mode == 0) {  // populate
            if (madvise(j->map + o, n, MADV_POPULATE_WRITE) != 0) { for (uint64_t q = 0; q < n; q += 4096) j->map[o + q] = 1; }
        } else if (j->mode == 1) {  // memcpy into the mapping (faults + copy)
            memcpy(j->map + o, j->src, n);
        } else {  // pwrite
            uint64_t done = 0;
            while (done < n) { ssize_t w = pwrite(j->fd, j->src + done, n - done, (off_t)(o + done)); if (w <= 0) { perror("pwrite"); exit(1); } done += (uint64_t)w; }
        }
    }
    return NULL;
}

static double run(int mode, int threads, int prealloc)
{
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { perror("open"); exit(1); }
    uint8_t *src = malloc(CH);
    memset(src, 'x', CH);
    double t0 = now();
    if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); exit(1); }
    if (prealloc && fallocate(fd, 0, 0, (off_t)total) != 0) perror("fallocate");
    double t_pre = now() - t0;
    uint8_t *map = NULL;
    if (mode != 2) {
        map = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); exit(1); }
    }
    pthread_t th[64];
    struct job jobs[64];
    uint64_t per = ((total / threads) + CH - 1) / CH * CH;
    for (int t = 0; t < threads; t++) {
        uint64_t lo = per * t, hi = lo + per > total ? total : lo + per;
        if (lo > total) lo = hi = total;
        jobs[t] = (struct job){map, lo, hi, mode, fd, src};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    double dt = now() - t0;
    if (map) munmap(map, total);
    close(fd);
    unlink(path);
    free(src);
    printf("  (ftruncate%s %.3f s) ", prealloc ? "+fallocate" : "", t_pre);
    return dt;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    path = argv[1];
    total = strtoull(argv[2], 0, 10);
    const char *names[] = {"mmap + MADV_POPULATE_WRITE (pages only, what hipHostRegister would fault in)", "mmap + memcpy (fault + copy)", "pwrite (what the CLI does)"};
    for (int mode = 0; mode < 3; mode++)
        for (int threads = 1; threads <= 8; threads *= 2) {
            double dt = run(mode, threads, 0);
            printf("%-80s %d thread(s): %.3f s  %.2f GB/s\n", names[mode], threads, dt, total / dt / 1e9);
            fflush(stdout);
        }
    for (int threads = 1; threads <= 4; threads *= 4) {
        double dt = run(2, threads, 1);
        printf("%-80s %d thread(s): %.3f s  %.2f GB/s\n", "fallocate first, then pwrite", threads, dt, total / dt / 1e9);
        dt = run(1, threads, 1);
        printf("%-80s %d thread(s): %.3f s  %.2f GB/s\n", "fallocate first, then mmap + memcpy", threads, dt, total / dt / 1e9);
        fflush(stdout);
    }
    return 0;
}
