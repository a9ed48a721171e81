// scratch/io_bench3.cc -- the candidate for the pipeline's file sink: one thread fallocates the new /dev/shm file in
// steps while T threads copy into a MAP_SHARED mapping behind it (optionally MADV_POPULATE_WRITE per segment first).
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
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4, n = gib << 30;
  const char* path2 = "/dev/shm/ghf_io_bench3.out";
  for (size_t seg : {(size_t)4 << 20, (size_t)16 << 20})
    for (int T : {4, 8, 16})
      for (int mode = 0; mode < 3; ++mode) {  // 0 memcpy, 1 populate+memcpy, 2 pwrite (inode lock shared with fallocate)
        unlink(path2);
        int fd = open(path2, O_CREAT | O_RDWR, 0600);
        if (ftruncate(fd, n) != 0) return 1;
        char* map = (char*)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::atomic<size_t> ready(0), next(0);
        const double t0 = now();
        std::thread alloc([&] {
          const size_t step = 32u << 20;
          for (size_t o = 0; o < n; o += step) {
            if (fallocate(fd, 0, o, step) != 0) perror("fallocate");
            ready.store(o + step);
          }
        });
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
          th.emplace_back([&] {
            char* b = (char*)aligned_alloc(4096, seg);
            memset(b, 1, seg);
            for (;;) {
              const size_t o = next.fetch_add(seg);
              if (o >= n) break;
              while (ready.load() < o + seg) std::this_thread::yield();
              if (mode == 1 && madvise(map + o, seg, MADV_POPULATE_WRITE) != 0) perror("madvise");
              if (mode == 2) { if (pwrite(fd, b, seg, o) != (ssize_t)seg) abort(); }
              else memcpy(map + o, b, seg);
            }
            free(b);
          });
        alloc.join();
        const double ta = now() - t0;
        for (auto& x : th) x.join();
        const double tw = now() - t0;
        munmap(map, n);
        close(fd);
        static const char* names[] = {"memcpy", "populate+memcpy", "pwrite"};
        printf("seg %2zu MiB T=%2d %-16s alloc done %.3f s, all done %.3f s = %6.2f GB/s\n", seg >> 20, T, names[mode], ta, tw, n / tw / 1e9);
        fflush(stdout);
      }
  unlink(path2);
  return 0;
}
