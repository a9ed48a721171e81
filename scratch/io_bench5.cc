// scratch/io_bench5.cc -- fallocate the whole new /dev/shm file first, THEN T threads fill it through a mapping:
// plain memcpy (minor fault per page) vs MADV_POPULATE_WRITE per segment + memcpy.
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
  const char* path2 = "/dev/shm/ghf_io_bench5.out";
  for (int T : {1, 2, 4, 8, 16})
    for (int mode = 0; mode < 2; ++mode) {
      unlink(path2);
      int fd = open(path2, O_CREAT | O_RDWR, 0600);
      double t0 = now();
      if (fallocate(fd, 0, 0, n) != 0) perror("fallocate");
      const double ta = now() - t0;
      char* map = (char*)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      std::vector<std::thread> th;
      t0 = now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([=] {
          char* b = (char*)aligned_alloc(4096, P);
          memset(b, 1, P);
          for (size_t o = (size_t)t * P; o < n; o += (size_t)T * P) {
            if (mode == 1 && madvise(map + o, P, MADV_POPULATE_WRITE) != 0) perror("madvise");
            memcpy(map + o, b, P);
          }
          free(b);
        });
      for (auto& x : th) x.join();
      const double tw = now() - t0;
      t0 = now();
      munmap(map, n);
      const double tu = now() - t0;
      close(fd);
      printf("T=%2d %-16s fallocate %.3f s  fill %6.2f GB/s  munmap %.3f s  => %6.2f GB/s\n", T, mode ? "populate+memcpy" : "memcpy", ta, n / tw / 1e9, tu,
             n / (ta + tw) / 1e9);
      fflush(stdout);
    }
  unlink(path2);
  return 0;
}
