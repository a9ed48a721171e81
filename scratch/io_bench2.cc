// scratch/io_bench2.cc -- ways to fill a NEW file in /dev/shm from T threads: mmap + memcpy, with and without
// MADV_POPULATE_WRITE first; fallocate alone; pwrite after fallocate.  g++ -O2 -pthread
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
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4, n = gib << 30, P = 16u << 20;
  const char* path2 = "/dev/shm/ghf_io_bench2.out";
  for (int T : {1, 2, 4, 8, 16}) {
    for (int mode = 0; mode < 4; ++mode) {
      unlink(path2);
      int fd = open(path2, O_CREAT | O_RDWR, 0600);
      double t0 = now(), tpre = 0;
      if (mode == 2 || mode == 3) {
        if (fallocate(fd, 0, 0, n) != 0) perror("fallocate");
        tpre = now() - t0;
      } else if (ftruncate(fd, n) != 0) return 1;
      char* map = (char*)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (map == MAP_FAILED) return 2;
      std::vector<std::thread> th;
      t0 = now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([=] {
          char* b = (char*)aligned_alloc(4096, P);
          memset(b, 1, P);
          for (size_t o = (size_t)t * P; o < n; o += (size_t)T * P) {
            if (mode == 1 && madvise(map + o, P, MADV_POPULATE_WRITE) != 0) perror("madvise");
            if (mode == 3) { if (pwrite(fd, b, P, o) != (ssize_t)P) abort(); }
            else memcpy(map + o, b, P);
          }
          free(b);
        });
      for (auto& x : th) x.join();
      const double tw = now() - t0;
      t0 = now();
      munmap(map, n);
      close(fd);
      const double tc = now() - t0;
      static const char* names[] = {"mmap+memcpy", "populate+memcpy", "fallocate, mmap+memcpy", "fallocate, pwrite"};
      printf("T=%2d %-24s pre %.3f s  fill %6.2f GB/s  total %6.2f GB/s (unmap %.3f s)\n", T, names[mode], tpre, n / tw / 1e9, n / (tpre + tw + tc) / 1e9, tc);
      fflush(stdout);
    }
  }
  unlink(path2);
  return 0;
}
