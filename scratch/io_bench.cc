// scratch/io_bench.cc -- what /dev/shm gives T threads: pread of an existing file into private buffers, pwrite of
// a new file (fresh pages) and of the same file again (pages exist).  g++ -O2 -pthread io_bench.cc -o io_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4, n = gib << 30, P = 16u << 20;
  const char* path = "/dev/shm/ghf_io_bench.bin";
  const char* path2 = "/dev/shm/ghf_io_bench.out";
  {
    int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
    std::vector<char> b(P, 'x');
    for (size_t o = 0; o < n; o += P) if (pwrite(fd, b.data(), P, o) != (ssize_t)P) return 1;
    close(fd);
  }
  for (int T : {1, 2, 4, 8, 12, 16}) {
    double tr, tw, tw2, tt;
    auto run = [&](int fd, bool wr) {
      std::vector<std::thread> th;
      const double t0 = now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([=] {
          char* b = (char*)aligned_alloc(4096, P);
          memset(b, 1, P);
          for (size_t o = (size_t)t * P; o < n; o += (size_t)T * P)
            if ((wr ? pwrite(fd, b, P, o) : pread(fd, b, P, o)) != (ssize_t)P) abort();
          free(b);
        });
      for (auto& x : th) x.join();
      return now() - t0;
    };
    int fd = open(path, O_RDONLY);
    tr = run(fd, false);
    close(fd);
    unlink(path2);
    fd = open(path2, O_CREAT | O_WRONLY, 0600);
    tw = run(fd, true);
    tw2 = run(fd, true);
    close(fd);
    double t0 = now();
    fd = open(path2, O_TRUNC | O_WRONLY);
    tt = now() - t0;
    close(fd);
    printf("T=%2d  pread %6.2f GB/s   pwrite(new) %6.2f GB/s   pwrite(again) %6.2f GB/s   truncate %.3f s\n", T, n / tr / 1e9, n / tw / 1e9,
           n / tw2 / 1e9, tt);
    fflush(stdout);
  }
  unlink(path);
  unlink(path2);
  return 0;
}
