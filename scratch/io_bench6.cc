// scratch/io_bench6.cc -- fill a new /dev/shm file through per-job WINDOWS (mmap 4 MiB, MADV_POPULATE_WRITE, memcpy,
// munmap) from T threads, (a) after one fallocate of the whole file, (b) while one thread fallocates ahead in steps.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4, n = gib << 30, W = 4u << 20;
  const char* path = "/dev/shm/ghf_io_bench6.out";
  for (int mode = 0; mode < 2; ++mode)
    for (size_t step : {(size_t)64 << 20, (size_t)256 << 20})
      for (int T : {4, 8, 12}) {
        if (mode == 0 && step != ((size_t)64 << 20)) continue;
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR, 0600);
        std::atomic<size_t> ready(0), next(0);
        const double t0 = now();
        double ta = 0;
        if (mode == 0) {
          if (fallocate(fd, 0, 0, n) != 0) perror("fallocate");
          ready = n;
          ta = now() - t0;
        }
        std::thread alloc([&] {
          if (mode == 0) return;
          for (size_t o = 0; o < n; o += step) {
            if (fallocate(fd, 0, o, step) != 0) perror("fallocate");
            ready.store(o + step);
          }
          ta = now() - t0;
        });
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
          th.emplace_back([&] {
            char* b = (char*)aligned_alloc(4096, W);
            memset(b, 1, W);
            for (;;) {
              const size_t o = next.fetch_add(W);
              if (o >= n) break;
              while (ready.load() < o + W) std::this_thread::yield();
              char* w = (char*)mmap(NULL, W, PROT_READ | PROT_WRITE, MAP_SHARED, fd, o);
              if (w == MAP_FAILED) abort();
              madvise(w, W, MADV_POPULATE_WRITE);
              memcpy(w, b, W);
              munmap(w, W);
            }
            free(b);
          });
        alloc.join();
        for (auto& x : th) x.join();
        const double tw = now() - t0;
        close(fd);
        printf("%-28s step %3zu MiB T=%2d  alloc done %.3f s, all done %.3f s = %6.2f GB/s\n", mode ? "fallocate ahead, concurrent" : "fallocate all, then fill", step >> 20, T,
               ta, tw, n / tw / 1e9);
        fflush(stdout);
      }
  unlink(path);
  return 0;
}
