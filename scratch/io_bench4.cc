// scratch/io_bench4.cc -- does /dev/shm hand out huge pages?  mmap + MADV_HUGEPAGE + memcpy from T threads.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? atoi(argv[1]) : 4, n = gib << 30, P = 16u << 20;
  const char* path2 = "/dev/shm/ghf_io_bench4.out";
  for (int T : {1, 4, 8, 16}) {
    unlink(path2);
    int fd = open(path2, O_CREAT | O_RDWR, 0600);
    if (ftruncate(fd, n) != 0) return 1;
    char* map = (char*)mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (madvise(map, n, MADV_HUGEPAGE) != 0) perror("madvise(MADV_HUGEPAGE)");
    std::vector<std::thread> th;
    const double t0 = now();
    for (int t = 0; t < T; ++t)
      th.emplace_back([=] {
        char* b = (char*)aligned_alloc(4096, P);
        memset(b, 1, P);
        for (size_t o = (size_t)t * P; o < n; o += (size_t)T * P) memcpy(map + o, b, P);
        free(b);
      });
    for (auto& x : th) x.join();
    const double tw = now() - t0;
    munmap(map, n);
    close(fd);
    printf("T=%2d hugepage-advised mmap+memcpy %6.2f GB/s\n", T, n / tw / 1e9);
    fflush(stdout);
  }
  unlink(path2);
  return 0;
}
