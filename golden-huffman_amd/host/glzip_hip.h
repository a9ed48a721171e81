// golden-huffman_amd/host/glzip_hip.h -- C++ host layer: the reference's Compressor<Encoder> /
// Decompressor<Decoder> API with policy classes that run on the MI355X through the C ABI (include/ghf.h).
//
// Mirrors, member for member (same names -- including the reference's spelling caculate_frequency --, same
// argument meaning, same file-name side effects):
//   Compressor<_Encoder>      include/compressor.h:44-77   (ctor(in, out&), default ctor, set_file, clear, compress)
//   Decompressor<_Decoder>    include/compressor.h:81-95   (ctor(in, out&), decompress)
//   CanonicalHuffEncoder<>    include/canonical_huff_encoder.h:48-121, include/encoder.h:56-213
//   CanonicalHuffDecoder<> / FastCanonicalHuffDecoder<> / TableCanonicalHuffDecoder<>
//                             include/canonical_huff_encoder.h:126-209, include/encoder.h:218-241
// so that   Compressor<HipCanonicalHuffEncoder<> > c; c.set_file(in, out); c.compress();
// writes the same bytes to the same file name as the reference's Compressor<CanonicalHuffEncoder<> >.
//
// Differences, all on the side of doing MORE than the reference:
//   * failures (unopenable file, empty input, code > 32 bits, corrupt stream, HIP error) throw
//     glzip_hip::Error; the reference has no error path at all (include/encoder.h:67-70 FIXME);
//   * file I/O is a pipeline of pieces (detail::Pipe below: reader/writer threads, pinned rings, one stream per
//     direction) instead of 64 KiB stdio buffers (utils/include/buffer.h:61-317); like the reference, the encoder
//     reads its input twice when the file does not fit its residency budget, and device memory is O(piece);
//   * only the byte-keyed (unsigned char) instantiation exists; anything else is a compile error.
// No HIP header is needed to compile this file: it links against libghf.so only.
#ifndef GLZIP_HIP_H_
#define GLZIP_HIP_H_
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "ghf.h"

namespace glzip_hip {

struct Error : public std::runtime_error {
  int status;
  Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

namespace detail {

// one ghf_ctx per policy object ("one context per host thread/GPU"), created on first use so that a
// default-constructed Compressor (file-scope objects in unit_tests/test.cc:45-46) touches no GPU
class Session {
 public:
  Session() : ctx_(NULL) {}
  ~Session() {
    if (ctx_) ghf_ctx_destroy(ctx_);
  }
  ghf_ctx* ctx() const {
    if (!ctx_) {
      int rc = ghf_ctx_create(device_from_env(), &ctx_);
      if (rc) throw Error(rc, std::string("ghf_ctx_create: ") + ghf_status_string(rc) + " " + ghf_last_error(NULL));
    }
    return ctx_;
  }
  void check(int rc, const char* where) const {
    if (rc) throw Error(rc, std::string(where) + ": " + ghf_status_string(rc) + " " + ghf_last_error(ctx_));
  }
  void sync(const char* where) const {
    int rc = ghf_sync(ctx());
    if (rc) {
      ghf_clear_status(ctx_);
      throw Error(rc, std::string(where) + ": " + ghf_status_string(rc));
    }
  }
  bool live() const { return ctx_ != NULL; }
  static int device_from_env() {
    const char* e = getenv("GHF_DEVICE");
    return e ? atoi(e) : 0;
  }

 private:
  Session(const Session&);
  Session& operator=(const Session&);
  mutable ghf_ctx* ctx_;
};

struct DeviceBuf {
  const Session* s;
  void* p;
  size_t n;
  DeviceBuf() : s(NULL), p(NULL), n(0) {}
  ~DeviceBuf() { reset(); }
  void reset() {
    if (p) ghf_device_free(s->ctx(), p);
    p = NULL;
    n = 0;
  }
  void alloc(const Session& ss, size_t bytes) {
    reset();
    s = &ss;
    ss.check(ghf_device_alloc(ss.ctx(), bytes, &p), "ghf_device_alloc");
    n = bytes;
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

struct PinnedBuf {
  const Session* s;
  void* p;
  size_t n;
  PinnedBuf() : s(NULL), p(NULL), n(0) {}
  ~PinnedBuf() { reset(); }
  void reset() {
    if (p) ghf_host_free(s->ctx(), p);
    p = NULL;
    n = 0;
  }
  void alloc(const Session& ss, size_t bytes) {
    reset();
    s = &ss;
    ss.check(ghf_host_alloc(ss.ctx(), bytes, &p), "ghf_host_alloc");
    n = bytes;
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

inline size_t file_size(FILE* f) {
  fseek(f, 0, SEEK_END);
  long long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  return n < 0 ? 0 : (size_t)n;
}

// ------------------------------------------------------------------------------------------------ the file pipeline
// what replaces utils/include/buffer.h:61-317 for the canonical policies.  A file moves in pieces:
//   reader threads (pread -> pinned ring) -> copy-in stream -> kernel stream -> copy-out stream -> pinned ring ->
//   writer threads (pwrite at the piece's own file offset)
// Three contexts = three HIP streams, ordered by events (ghf_event_*), so the H2D of piece k+1, the kernels of piece k
// and the D2H of piece k-1 run at the same time and the host never waits for a stream as a whole.
inline size_t env_bytes(const char* name, size_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = NULL;
  unsigned long long v = strtoull(e, &end, 10);
  if (end && (*end == 'k' || *end == 'K')) v <<= 10;
  else if (end && (*end == 'm' || *end == 'M')) v <<= 20;
  else if (end && (*end == 'g' || *end == 'G')) v <<= 30;
  return (size_t)v;
}

// GHF_PIPE_TRACE=1: wall time of every step of the policies, on stderr
class StepTimer {
 public:
  explicit StepTimer(const char* what) : what_(what), on_(getenv("GHF_PIPE_TRACE") != NULL), t0_(now()) {}
  ~StepTimer() {
    if (on_) fprintf(stderr, "[ghf] %-20s %9.2f ms\n", what_, now() - t0_);
  }
  static double now() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

 private:
  const char* what_;
  bool on_;
  double t0_;
};

class StepSum {  // GHF_PIPE_TRACE=1: time spent inside one kind of call, summed over an operation
 public:
  explicit StepSum(const char* what) : what_(what), on_(getenv("GHF_PIPE_TRACE") != NULL), ms_(0), n_(0) {}
  ~StepSum() { report(); }
  void report() {
    if (on_ && n_) fprintf(stderr, "[ghf]   %-18s %9.2f ms in %zu calls\n", what_, ms_, n_);
    ms_ = 0, n_ = 0;
  }
  struct Scope {
    StepSum& s;
    double t0;
    explicit Scope(StepSum& ss) : s(ss), t0(ss.on_ ? StepTimer::now() : 0) {}
    ~Scope() {
      if (s.on_) s.ms_ += StepTimer::now() - t0, ++s.n_;
    }
  };

 private:
  const char* what_;
  bool on_;
  double ms_;
  size_t n_;
};

class Event {
 public:
  Event() : e_(NULL), live_(false) {}
  ~Event() {
    if (e_) ghf_event_destroy(e_);
  }
  void record(const Session& s) {
    if (!e_) s.check(ghf_event_create(s.ctx(), &e_), "ghf_event_create");
    s.check(ghf_event_record(s.ctx(), e_), "ghf_event_record");
    live_ = true;
  }
  void hold(const Session& s) const {  // s's stream waits; the host does not
    if (live_) s.check(ghf_event_wait(s.ctx(), e_), "ghf_event_wait");
  }
  void sync() const {  // the host waits
    if (live_ && ghf_event_sync(e_) != GHF_OK) throw Error(GHF_E_HIP, "ghf_event_sync");
  }

 private:
  Event(const Event&);
  Event& operator=(const Event&);
  ghf_event* e_;
  bool live_;
};

class IoPool {  // the reader/writer threads; a job's exception comes back through its future
 public:
  IoPool() : stop_(false) {}
  ~IoPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (size_t i = 0; i < th_.size(); ++i) th_[i].join();
  }
  std::future<void> submit(std::function<void()> f) {
    if (th_.empty()) {
      size_t n = env_bytes("GHF_IO_THREADS", 12);
      if (n < 1) n = 1;
      if (n > 64) n = 64;
      for (size_t i = 0; i < n; ++i) th_.push_back(std::thread(&IoPool::loop, this));
    }
    std::packaged_task<void()> t(std::move(f));
    std::future<void> fut = t.get_future();
    {
      std::lock_guard<std::mutex> g(m_);
      q_.push_back(std::move(t));
    }
    cv_.notify_one();
    return fut;
  }

 private:
  void loop() {
    for (;;) {
      std::packaged_task<void()> t;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        t = std::move(q_.front());
        q_.pop_front();
      }
      t();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::packaged_task<void()> > q_;
  std::vector<std::thread> th_;
  bool stop_;
};

inline void pread_all(int fd, uint8_t* dst, size_t len, size_t off) {
  while (len) {
    const ssize_t r = pread(fd, dst, len, (off_t)off);
    if (r <= 0) throw Error(GHF_E_INVAL, "short read");
    dst += r, off += (size_t)r, len -= (size_t)r;
  }
}
inline void pwrite_all(int fd, const uint8_t* src, size_t len, size_t off) {
  while (len) {
    const ssize_t r = pwrite(fd, src, len, (off_t)off);
    if (r <= 0) throw Error(GHF_E_INVAL, "short write");
    src += r, off += (size_t)r, len -= (size_t)r;
  }
}

struct Jobs {  // the pool jobs that still use a pinned buffer
  std::vector<std::future<void> > f;
  void add(std::future<void> x) { f.push_back(std::move(x)); }
  void wait() {
    for (size_t i = 0; i < f.size(); ++i)
      if (f[i].valid()) f[i].wait();
  }
  void drop() {
    wait();
    f.clear();
  }
  void get() {  // wait; the first error, if any, is rethrown
    std::vector<std::future<void> > t;
    t.swap(f);
    std::exception_ptr err;
    for (size_t i = 0; i < t.size(); ++i) {
      try {
        t[i].get();
      } catch (...) {
        if (!err) err = std::current_exception();
      }
    }
    if (err) std::rethrow_exception(err);
  }
};

// The output file.  Default: created / emptied ("w+b", what the reference's fopen(.., "wb") does).  GHF_SINK=reuse: an
// existing file is opened as it is -- its pages stay, and the sink copies into them instead of making new ones (a new
// file in /dev/shm is bound by the kernel's page allocation, 20 GB/s at best; see the sink notes in Pipe); the file is cut to
// its final size at the end either way.
inline bool sink_reuses() {
  const char* how = getenv("GHF_SINK");
  return how && strcmp(how, "reuse") == 0;
}
inline FILE* open_output(const std::string& name) {
  if (sink_reuses()) {
    FILE* f = fopen(name.c_str(), "r+b");
    if (f) return f;
  }
  return fopen(name.c_str(), "w+b");
}

class Pipe {
 public:
  static const int kSlots = 6;  // per ring; a ring is used first-in first-out
  static const size_t kLag = 2;
  static const size_t kFill = (size_t)4 << 20;        // bytes per copy job
  static const size_t kAllocStep = (size_t)256 << 20;  // bytes per fallocate
  static const size_t kUnmapStep = (size_t)64 << 20;   // bytes per munmap behind the copies
  Pipe() : t_read_("wait for a read"), t_slot_("wait for a slot"), t_land_("wait for a D2H"), t_alloc_("fallocate"), t_flush_("last writes"),
           piece_(0), in_next_(0), out_next_(0), sink_fd_(-1), sink_len_(0), ready_(0), map_(NULL), sized_(false), no_alloc_(false) {}
  ~Pipe() { quiesce(); }
  // One pipe (three contexts, the pinned rings, the I/O threads) per host thread, shared by the policy objects that
  // are alive on it and gone with the last of them: a Decompressor made after a Compressor (unit_tests/test.cc:136-156
  // makes a new one per call) starts warm.
  static std::shared_ptr<Pipe> shared() {
    static thread_local std::weak_ptr<Pipe> w;
    std::shared_ptr<Pipe> p = w.lock();
    if (!p) {
      p.reset(new Pipe);
      w = p;
    }
    return p;
  }

  // piece size: GHF_PIECE_BYTES (default 16 MiB; a multiple of 64 KiB so that pieces stay 16-byte aligned in the stream)
  size_t piece() {
    if (!piece_) {
      size_t p = env_bytes("GHF_PIECE_BYTES", (size_t)16 << 20);
      p = (p + 0xFFFF) & ~(size_t)0xFFFF;
      piece_ = p ? p : 0x10000;
    }
    return piece_;
  }
  const Session& in() const { return in_s_; }    // copy-in stream
  const Session& run() const { return run_s_; }  // kernel stream
  const Session& out() const { return out_s_; }  // copy-out stream
  size_t piece_bytes() const { return piece_; }

  // start of an operation: whatever an earlier one that failed half-way left behind is dropped, not written
  void begin() {
    for (int ring = 0; ring < 2; ++ring) {
      Ring& r = ring ? out_ring_ : in_ring_;
      for (size_t i = 0; i < r.size(); ++i) r[i].job.drop();
    }
    pend_.clear();
    landing_.clear();
    unmap_.drop();
    if (map_) munmap(map_ + unmapped_, sink_len_ - unmapped_);
    map_ = NULL;
    sink_fd_ = -1;
  }

  // ---- file -> device -------------------------------------------------------------------------------------------
  // The byte range [base, base + file_bytes) of fd, followed by the tail_n <= 16 bytes at `tail` (a .crs keeps the last,
  // incomplete byte of its body in front of it), in pieces of piece() bytes, each followed by `extra` look-ahead bytes
  // of the next piece, the whole zero-filled up to padded(k).  Reads run `kSlots - 2` pieces ahead on the pool.
  void open_feed(int fd, size_t base, size_t file_bytes, size_t extra, const uint8_t* tail = NULL, size_t tail_n = 0) {
    fd_ = fd, base_ = base, file_bytes_ = file_bytes, total_ = file_bytes + tail_n, extra_ = extra;
    tail_n_ = tail_n < sizeof tail_ ? tail_n : sizeof tail_;
    if (tail_n_) memcpy(tail_, tail, tail_n_);
    next_read_ = next_feed_ = 0;
    pend_.clear();
    ensure_ring(in_ring_, piece() + 64);
  }
  size_t pieces() const { return (total_ + piece_ - 1) / piece_; }
  size_t own(size_t k) const { return total_ - k * piece_ < piece_ ? total_ - k * piece_ : piece_; }
  size_t padded(size_t k) const { return extra_ ? ((own(k) + extra_ + 15) & ~(size_t)15) + 16 : own(k); }
  // the next piece (in order) goes to d_dst on the copy-in stream; `arrived` is recorded behind the copy
  void feed(uint8_t* d_dst, Event& arrived) {
    const size_t k = next_feed_++;
    prefetch(k + kSlots - 2);
    Slot& sl = in_ring_[pend_.front()];
    pend_.pop_front();
    {
      StepSum::Scope t(t_read_);
      sl.job.get();  // the read is done (or rethrows its error)
    }
    in_s_.check(ghf_copy_h2d(in_s_.ctx(), d_dst, sl.buf.p, padded(k)), "ghf_copy_h2d");
    sl.copied.record(in_s_);
    arrived.record(in_s_);
    prefetch(k + kSlots - 1);  // the slot of piece k-1 is free by now (its copy went out one feed ago)
  }

  // ---- device -> file -------------------------------------------------------------------------------------------
  // After `ready`: d_src[0, bytes) goes to fd at file_off -- D2H on the copy-out stream in segments of piece() bytes,
  // each written by the pool once it has landed.  Returns when every copy is enqueued; at most kLag + 1 landed or
  // landing segments wait for their write at any time, the oldest is handed to the pool before a new slot is taken.
  void drain(const Event& ready, const uint8_t* d_src, size_t bytes, int fd, size_t file_off) {
    ensure_ring(out_ring_, piece());
    ready.hold(out_s_);
    for (size_t off = 0; off < bytes; off += piece_) {
      while (landing_.size() > kLag) store_oldest();
      Landing l;
      l.slot = acquire(out_ring_, out_next_);
      l.len = bytes - off < piece_ ? bytes - off : piece_;
      l.fd = fd;
      l.at = file_off + off;
      Slot& sl = out_ring_[l.slot];
      out_s_.check(ghf_copy_d2h(out_s_.ctx(), sl.buf.p, d_src + off, l.len), "ghf_copy_d2h");
      sl.copied.record(out_s_);
      landing_.push_back(l);
    }
  }
  // every write that was started has reached the file (write errors surface here)
  void flush() {
    while (!landing_.empty()) store_oldest();
    for (size_t i = 0; i < out_ring_.size(); ++i) out_ring_[i].job.get();
  }

  // ---- the output file ------------------------------------------------------------------------------------------
  // A new file in /dev/shm takes 8.4 GB/s from ONE pwrite-ing thread and less from several (they queue on the inode
  // lock and on the page allocation); pages that fallocate made (20 GB/s, one call) take 54 GB/s from 8 threads that
  // MADV_POPULATE_WRITE + memcpy through a shared mapping -- as long as nobody allocates at the same time
  // (profiles/r02_io/: io_bench*.log).  So: drain() targets inside the sink go through the mapping; pages are made in
  // steps of kAllocStep by the thread that drives the pipeline, between two rounds of copies.  `bound` >= the final
  // size; close_sink(final) cuts the file to it.  GHF_SINK=pwrite keeps everything on pwrite (and so does any failure
  // of ftruncate / mmap / fallocate: a full file system is then an error code, not a SIGBUS).
  // `exact`: bound IS the final size (all of its pages are made right away, in one call).
  void open_sink(int fd, size_t bound, bool exact) {
    sink_fd_ = fd;
    sink_len_ = bound;
    ready_ = 0;
    sized_ = no_alloc_ = false;
    pre_job_.wait();
    const size_t made = pre_fd_ == fd ? pre_done_.load() : 0;
    pre_fd_ = -1;
    if (!wants_map(bound)) return;
    size_t have = 0;  // GHF_SINK=reuse: what the file already holds counts as made pages (the caller's promise: no holes)
    size_t size = 0;  // ... and its REAL size decides whether it has to grow: stores through a shared mapping behind the end of
                      // the file are unspecified, also inside its last page (tmpfs keeps them, xfs / ext4 zero that tail when the
                      // file is extended or written back) -- a new output 1..4095 bytes longer than the old one lands exactly there
    if (sink_reuses()) {
      struct stat st;
      if (fstat(fd, &st) == 0 && st.st_size > 0) {
        size = (size_t)st.st_size;
        have = (size + 4095) & ~(size_t)4095;  // (its last, partial page is a page)
      }
    }
    if (size < bound && ftruncate(fd, (off_t)bound) != 0) return;
    sized_ = true;
    void* m = mmap(NULL, bound, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return;
    map_ = static_cast<uint8_t*>(m);
    unmapped_ = 0;
    const size_t end = (bound + 4095) & ~(size_t)4095;
    ready_ = made < end ? made : end;
    if (have > ready_) ready_ = have < end ? have : end;
    if (exact) (void)reserve(bound, true);
  }
  // Pages for [0, bytes) of the file that open_sink() is given next, made by a pool thread from now on: the encoder
  // knows roughly how much it will write long before it may write (the two passes of a Huffman coder),
  // and page-making is the slowest stage of the output side.  cancel_presize() before the file is closed.
  void presize(int fd, size_t bytes) {
    cancel_presize();
    if (sink_reuses()) return;  // (the pages are there; what is missing is made by reserve())
    if (!wants_map(bytes) || env_bytes("GHF_IO_THREADS", 12) < 3) return;  // (it would sit in front of the reads in the queue)
    pre_fd_ = fd;
    pre_done_ = 0;
    pre_stop_ = false;
    std::atomic<size_t>* done = &pre_done_;
    std::atomic<bool>* stop = &pre_stop_;
    pre_job_.add(pool_.submit([fd, bytes, done, stop] {
      const size_t end = (bytes + 4095) & ~(size_t)4095;
      for (size_t o = 0; o < end && !stop->load(); o += kAllocStep) {
        const size_t len = end - o < kAllocStep ? end - o : kAllocStep;
        if (fallocate(fd, 0, (off_t)o, (off_t)len) != 0) break;
        done->store(o + len);
      }
    }));
  }
  // presize() was started after open_sink(): count what it made (call cancel_presize() first)
  void adopt_presized(int fd) {
    if (pre_fd_last_ == fd && map_ && pre_done_.load() > ready_) {
      const size_t end = (sink_len_ + 4095) & ~(size_t)4095;
      ready_ = pre_done_.load() < end ? pre_done_.load() : end;
    }
  }
  void cancel_presize() {
    pre_fd_last_ = pre_fd_;
    pre_stop_ = true;
    pre_job_.drop();
    pre_fd_ = -1;
  }
  static bool wants_map(size_t bound) {
    const char* how = getenv("GHF_SINK");
    return how ? (strcmp(how, "mmap") == 0 || strcmp(how, "reuse") == 0) : bound >= ((size_t)8 << 20);
  }
  // the driver knows the final size by now (the decoder, once everything is decoded): make the rest of the pages at once
  void reserve_all(size_t bytes) { (void)reserve(bytes, true); }
  // an operation failed half-way: nothing more is written, the mapping goes, the file is emptied
  void abandon_sink() {
    pre_stop_ = true;  // (nothing may go on making pages for a file that is about to be emptied and closed)
    pre_job_.drop();
    pre_fd_ = -1;
    for (size_t i = 0; i < out_ring_.size(); ++i) out_ring_[i].job.drop();
    landing_.clear();
    unmap_.drop();
    if (map_) munmap(map_ + unmapped_, sink_len_ - unmapped_);
    map_ = NULL;
    if (sized_ && sink_fd_ >= 0) (void)!ftruncate(sink_fd_, 0);
    sized_ = false;
    sink_fd_ = -1;
  }
  void close_sink(size_t final_size) {
    {
      StepSum::Scope t(t_flush_);
      flush();
    }
    t_read_.report(), t_slot_.report(), t_land_.report(), t_alloc_.report(), t_flush_.report();
    if (sized_ && ftruncate(sink_fd_, (off_t)final_size) != 0) throw Error(GHF_E_INVAL, "ftruncate");
    if (map_) munmap(map_ + unmapped_, sink_len_ - unmapped_);  // what the pool has not taken down yet (the last few segments)
    unmap_.drop();
    map_ = NULL;
    sink_fd_ = -1;
  }

 private:
  struct Slot {
    PinnedBuf buf;
    Event copied;           // the last hipMemcpyAsync that used buf
    Jobs job;               // the last read into / writes from buf
  };
  typedef std::vector<Slot> Ring;
  void ensure_ring(Ring& r, size_t bytes) {  // (a slot is pinned when it is first taken: the first pieces of a process's
    if (r.empty()) r = Ring(kSlots);          //  first file do not wait for 192 MiB of hipHostMalloc)
    slot_bytes_ = bytes > slot_bytes_ ? bytes : slot_bytes_;
  }
  int acquire(Ring& r, size_t& next) {  // oldest slot of the ring, once its last job and copy are over
    StepSum::Scope t(t_slot_);
    const int si = (int)(next++ % r.size());
    r[si].job.get();
    r[si].copied.sync();
    if (r[si].buf.n < slot_bytes_) r[si].buf.alloc(run_s_, slot_bytes_);
    return si;
  }
  struct Landing {  // a segment on its way from the device into the pinned ring
    int slot, fd;
    size_t len, at;
  };
  void store_oldest() {
    const Landing l = landing_.front();
    landing_.pop_front();
    Slot& sl = out_ring_[l.slot];
    {
      StepSum::Scope t(t_land_);
      sl.copied.sync();
    }
    const uint8_t* src = sl.buf.u8();
    const int fd = l.fd;
    const size_t len = l.len, at = l.at;
    if (fd == sink_fd_ && reserve(at + len)) {
      uint8_t* map = map_;
      for (size_t o = 0; o < len; o += kFill) {  // several threads per segment: the slot is free again sooner
        const size_t part = len - o < kFill ? len - o : kFill, to = at + o;
        const uint8_t* from = src + o;
        sl.job.add(pool_.submit([map, to, from, part] {
          const size_t lo = to & ~(size_t)4095, hi = (to + part + 4095) & ~(size_t)4095;
          (void)madvise(map + lo, hi - lo, 23 /* MADV_POPULATE_WRITE, Linux 5.14; without it the copy faults page by page */);
          memcpy(map + to, from, part);
        }));
      }
      // Taking a populated mapping down costs 0.16 s per 4 GiB: it is done behind the copies, by the pool, in steps.
      // Segments are stored in file order and a slot is only reused once its copies are over, so everything more
      // than 8 segments behind this one has been written.
      const size_t lag = 8 * piece_;
      if (at > lag) {
        const size_t hi = (at - lag) & ~(kUnmapStep - 1);
        if (hi > unmapped_) {
          uint8_t* lo_p = map + unmapped_;
          const size_t n = hi - unmapped_;
          unmap_.add(pool_.submit([lo_p, n] { munmap(lo_p, n); }));
          unmapped_ = hi;
        }
      }
    } else {
      sl.job.add(pool_.submit([fd, src, len, at] { pwrite_all(fd, src, len, at); }));
    }
  }
  // pages for [0, upto) of the sink exist (driver thread only; no copy into the mapping runs while pages are made)
  bool reserve(size_t upto, bool just_that = false) {
    if (!map_ || no_alloc_) return false;
    if (upto <= ready_) return true;
    StepSum::Scope t(t_alloc_);
    for (size_t i = 0; i < out_ring_.size(); ++i) out_ring_[i].job.wait();
    const size_t end = (sink_len_ + 4095) & ~(size_t)4095;
    // (a reused file is rarely outgrown by much: small steps -- one of 256 MiB behind a 4 GiB output that is 57 bytes longer
    //  than the pages counted took 13-24 ms of a 160 ms decompress)
    const size_t step = sink_reuses() ? (size_t)16 << 20 : kAllocStep;
    size_t want = (ready_ + step > upto && !just_that) ? ready_ + step : (upto + 4095) & ~(size_t)4095;
    if (want > end) want = end;
    if (fallocate(sink_fd_, 0, (off_t)ready_, (off_t)(want - ready_)) != 0) {
      no_alloc_ = true;
      return false;
    }
    ready_ = want;
    return true;
  }
  void prefetch(size_t upto) {
    const size_t np = pieces();
    for (; next_read_ < np && next_read_ < upto; ++next_read_) {
      const int si = acquire(in_ring_, in_next_);
      Slot& sl = in_ring_[si];
      uint8_t* dst = sl.buf.u8();
      const size_t k = next_read_, v0 = k * piece_, off = base_ + v0, pad = padded(k);
      const size_t avail = v0 < file_bytes_ ? file_bytes_ - v0 : 0, want = own(k) + extra_;
      const size_t len = want < avail ? want : avail;  // what comes from the file; behind it zeros and, where they fall, the tail bytes
      const int fd = fd_;
      uint8_t tb[16];
      memcpy(tb, tail_, sizeof tb);
      const size_t tail_lo = file_bytes_, tail_n = tail_n_;
      for (size_t o = 0; o < len || o == 0; o += kFill) {  // several threads per piece: one pread moves 6-9 GB/s
        const size_t part = len - o < kFill ? len - o : kFill;
        const bool last = o + part >= len;
        sl.job.add(pool_.submit([fd, dst, o, part, off, len, pad, last, v0, tail_lo, tail_n, tb] {
          if (part) pread_all(fd, dst + o, part, off + o);
          if (!last) return;
          if (pad > len) memset(dst + len, 0, pad - len);
          for (size_t i = 0; i < tail_n; ++i)
            if (tail_lo + i >= v0 && tail_lo + i - v0 < pad) dst[tail_lo + i - v0] = tb[i];
        }));
      }
      pend_.push_back(si);
    }
  }
  void quiesce() {  // nothing may still touch the rings when they are freed
    for (size_t i = 0; i < in_ring_.size(); ++i) in_ring_[i].job.wait();
    for (size_t i = 0; i < out_ring_.size(); ++i) out_ring_[i].job.wait();
    unmap_.wait();
    pre_stop_ = true;
    pre_job_.wait();
    if (map_) munmap(map_ + unmapped_, sink_len_ - unmapped_);
    if (in_s_.live()) ghf_sync(in_s_.ctx());
    if (out_s_.live()) ghf_sync(out_s_.ctx());
  }
  StepSum t_read_, t_slot_, t_land_, t_alloc_, t_flush_;
  Session in_s_, run_s_, out_s_;
  size_t piece_;
  Ring in_ring_, out_ring_;
  size_t in_next_, out_next_, slot_bytes_ = 0;
  int fd_;
  size_t base_, file_bytes_ = 0, total_, extra_, next_read_, next_feed_, tail_n_ = 0;
  uint8_t tail_[16] = {0};
  std::deque<int> pend_;  // slots of the reads that were started and not fed yet, in piece order
  std::deque<Landing> landing_;
  int sink_fd_;
  size_t sink_len_, ready_, unmapped_ = 0;
  uint8_t* map_;
  bool sized_, no_alloc_;
  Jobs unmap_, pre_job_;
  int pre_fd_ = -1, pre_fd_last_ = -1;
  std::atomic<size_t> pre_done_{0};
  std::atomic<bool> pre_stop_{false};
  IoPool pool_;           // last member: its threads stop before anything above goes away
};

}  // namespace detail

// ------------------------------------------------------------------------------------------------ Compressor
template <typename _Encoder>
class Compressor {  // include/compressor.h:44-77
 public:
  Compressor(const std::string& infile_name, std::string& outfile_name) : encoder_(infile_name, outfile_name) {}
  Compressor() {}
  void set_file(const std::string& infile_name, std::string& outfile_name) { encoder_.set_file(infile_name, outfile_name); }
  void clear() { encoder_.clear(); }
  void compress() {  // include/compressor.h:62-73: the four steps, in this order
    encoder_.caculate_frequency();
    encoder_.gen_encode();
    encoder_.write_encode_info();
    encoder_.encode_file();
  }
  _Encoder& encoder() { return encoder_; }

 private:
  _Encoder encoder_;
};

template <typename _Decoder>
class Decompressor {  // include/compressor.h:81-95
 public:
  Decompressor(const std::string& infile_name, std::string& outfile_name) : decoder_(infile_name, outfile_name) {}
  void decompress() {
    decoder_.get_encode_info();
    decoder_.decode_file();
  }
  _Decoder& decoder() { return decoder_; }

 private:
  _Decoder decoder_;
};

// ------------------------------------------------------------------------------------------------ the two passes of an encoder
namespace detail {

// What the two formats' encoders share: the files, pass 1 (K1 over the pieces) and pass 2 (K4 + K5 piece by piece).
class PieceEncoder {
 protected:
  PieceEncoder() : infile_(NULL), outfile_(NULL), n_(0) {}
  ~PieceEncoder() { close_files(); }

  // include/canonical_huff_encoder.cc:15-32, include/normal_huff_encoder.h:83-99: the output name defaults to
  // <in><ext> and is handed back
  void open_files(const std::string& infile_name, std::string& outfile_name, const char* ext) {
    close_files();
    infile_name_ = infile_name;
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ext;
    outfile_ = detail::open_output(outfile_name);  // read-write: the pipeline maps it
    if (!outfile_) throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
  }
  void close_files() {  // include/encoder.h:85-92
    if (pipe_) pipe_->cancel_presize();  // nothing may still be making pages for a file that is about to be closed
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
    infile_ = NULL;
    outfile_ = NULL;
  }
  Pipe& pipe() const {
    if (!pipe_) pipe_ = Pipe::shared();
    return *pipe_;
  }

  // include/encoder.h:99-105,123-150 -- pass 1 over the file: piece k is counted (K1, accumulating) while piece k+1
  // is on the PCIe bus and pieces k+2.. are being read.  A file of up to GHF_RESIDENT_BYTES (default 64 GiB of the
  // 288) stays in HBM for pass 2; a larger one is read a second time there, as the reference does
  // (include/canonical_huff_encoder.cc:247-248 rewinds the input), through a ring of kRing piece buffers.
  // d_hist_[0..255] <- the byte counts, [256] <- 1.
  void count_pieces() {
    pipe().begin();
    const Session& run = pipe().run();
    const size_t P = pipe().piece(), np = (n_ + P - 1) / P;
    resident_ = n_ <= env_bytes("GHF_RESIDENT_BYTES", (size_t)64 << 30);
    if (resident_ && d_in_.n < n_ + 16) {
      try {
        d_in_.alloc(run, n_ + 16);
      } catch (const Error&) {
        resident_ = false;
      }
    }
    if (!resident_) {
      d_in_.reset();
      for (int r = 0; r < kRing; ++r)
        if (d_ring_[r].n < P + 16) d_ring_[r].alloc(run, P + 16);
    }
    if (!d_hist_.p) d_hist_.alloc(run, GHF_NSYM * sizeof(uint64_t));
    uint64_t* hist = static_cast<uint64_t*>(d_hist_.p);
    run.check(ghf_memset_d(run.ctx(), hist, 0, GHF_NSYM * sizeof(uint64_t)), "ghf_memset_d");
    pipe().open_feed(fileno(infile_), 0, n_, 0);
    for (size_t k = 0; k < np; ++k) {
      const int r = (int)(k % kRing);
      uint8_t* dst = piece_at(k);
      if (!resident_) used_[r].hold(pipe().in());  // K1 of piece k - kRing has let go of this buffer
      pipe().feed(dst, arrived_[r]);
      arrived_[r].hold(run);
      run.check(ghf_histogram_add(run.ctx(), dst, pipe().own(k), hist), "ghf_histogram_add");
      used_[r].record(run);
      if (k == 0 && np > 2) {  // the first piece's counts come back early: how big will the output be, roughly?
        if (!h_first_.p) h_first_.alloc(run, GHF_NSYM * sizeof(uint64_t));
        run.check(ghf_copy_d2h(run.ctx(), h_first_.p, hist, GHF_NSYM * sizeof(uint64_t)), "ghf_copy_d2h");
        first_counted_.record(run);
      }
      if (k == 1 && np > 2) presize_from_first_piece();
    }
    run.sync("caculate_frequency");
  }

  // include/canonical_huff_encoder.cc:245-285 / include/normal_huff_encoder.h:159-186 -- pass 2, piece by piece.
  // A piece is to the stream what a shard is to the multi-GPU path (include/ghf.h, ghf_encode_sharded): K4 prices it,
  // its first bit is the running sum of the prices before it, K5 packs it at that absolute bit into a buffer of its
  // own (GHF_EMIT_REBASE), and the D2H of piece k runs while K4/K5 work on piece k+1 and the writer threads put piece
  // k-1 into the file at its own offset.  Neighbouring pieces share one 16-byte unit of the stream (a piece does not
  // end on a byte): that unit is OR-ed together here (`edge`), everything else goes from HBM to the file untouched.
  //   head / head_bytes : what the file holds in front of the body (already written); the first code starts at bit
  //                       8 * head_bytes
  //   last_flags        : GHF_EMIT_LAST for a stream that ends with the end mark (end_mark_bits long) and 1-padding
  //   out_bytes_        : the size the counts give (header + every byte that holds a bit of the body): checked
  // Returns the stream's end bit; last_byte_ <- the stream byte that holds its last bit.
  //   more_flags        : GHF_EMIT_LONG_CODES when the code has words of more than 32 bits (a .crs tree deeper than 32)
  uint64_t emit_pieces(const ghf_code* dc, const uint8_t* head, size_t head_bytes, int last_flags, uint32_t end_mark_bits,
                       int more_flags = 0) {
    pipe().begin();
    const Session &run = pipe().run(), &out = pipe().out();
    const size_t P = pipe().piece(), np = (n_ + P - 1) / P;
    const size_t cap = ghf_shard_bound(P) * ((more_flags & GHF_EMIT_LONG_CODES) ? 2 : 1);  // up to 32 (64) bits per symbol
    for (int r = 0; r < kRing; ++r)
      if (d_out_[r].n < cap) d_out_[r].alloc(run, cap);
    if (!d_scal_.p) d_scal_.alloc(run, kRing * 4 * sizeof(uint64_t));
    if (!h_scal_.p) h_scal_.alloc(run, kRing * 4 * sizeof(uint64_t) + kRing * 32 + 16);
    uint64_t* const ds = static_cast<uint64_t*>(d_scal_.p);                 // per ring slot: total bits, start bit, end[2]
    volatile uint64_t* const hs = static_cast<uint64_t*>(h_scal_.p);        // pinned mirror of [0..1]
    uint8_t* const h_edge = h_scal_.u8() + kRing * 4 * sizeof(uint64_t);    // per ring slot: first unit, last unit
    volatile uint8_t* const h_last = h_edge + kRing * 32;                   // the stream's last byte
    const int fd = fileno(outfile_);
    if (!resident_) pipe().open_feed(fileno(infile_), 0, n_, 0);
    pipe().open_sink(fd, out_bytes_, true);
    StepSum t_plan("wait for K4");
    uint64_t start = 8 * (uint64_t)head_bytes;  // absolute stream bit of the piece's first code
    uint64_t end = start;
    uint8_t edge[16];                           // the unit at stream byte (start / 128) * 16, as far as it is known
    memset(edge, 0, sizeof edge);
    memcpy(edge, head + (head_bytes & ~(size_t)15), head_bytes & 15);
    struct Span {
      bool live, last;
      int r;
      size_t F, U, E;  // stream bytes: first unit, unit shared with the next piece (E for the last piece), end
    } prev = {false, false, 0, 0, 0, 0};
    for (size_t k = 0; k < np; ++k) {
      const int r = (int)(k % kRing);
      const size_t len = n_ - k * P < P ? n_ - k * P : P;
      const bool last = k + 1 == np;
      const uint8_t* src = piece_at(k);
      if (!resident_) {
        used_[r].hold(pipe().in());
        pipe().feed(piece_at(k), arrived_[r]);
        arrived_[r].hold(run);
      }
      run.check(ghf_encode_plan(run.ctx(), src, len, dc, ds + 4 * r), "ghf_encode_plan");  // K4
      run.check(ghf_copy_d2h(run.ctx(), const_cast<uint64_t*>(hs + 4 * r), ds + 4 * r, sizeof(uint64_t)), "ghf_copy_d2h");
      planned_[r].record(run);
      if (prev.live) close_span(prev.r, prev.F, prev.U, prev.E, prev.last, edge, h_edge, fd);  // while K4 runs
      {
        StepSum::Scope t(t_plan);
        planned_[r].sync();
      }
      const uint64_t total = hs[4 * r];
      hs[4 * r + 1] = start;
      run.check(ghf_copy_h2d(run.ctx(), ds + 4 * r + 1, const_cast<uint64_t*>(hs + 4 * r + 1), sizeof(uint64_t)), "ghf_copy_h2d");
      fetched_[r].hold(run);  // the D2H of piece k - kRing has let go of d_out_[r]
      run.check(ghf_encode_emit(run.ctx(), src, len, dc, ds + 4 * r + 1, GHF_EMIT_REBASE | more_flags | (last ? last_flags : 0), d_out_[r].u8(), cap,
                                NULL, ds + 4 * r + 2),
                "ghf_encode_emit");  // K5
      used_[r].record(run);
      emitted_[r].record(run);
      end = start + total;
      if (last && (last_flags & GHF_EMIT_LAST)) end = (end + end_mark_bits + 7) & ~(uint64_t)7;  // end mark, then 1-bits up to the byte
      Span sp = {true, last, r, (size_t)(start >> 7) << 4, 0, (size_t)((end + 7) >> 3)};
      sp.U = last ? sp.E : (size_t)(end >> 7) << 4;
      // d_out_[r][0] is stream byte F.  The unit at F and the one at U are shared with the neighbours: to the host.
      emitted_[r].hold(out);
      const size_t nfirst = sp.E - sp.F < 16 ? sp.E - sp.F : 16;
      out.check(ghf_copy_d2h(out.ctx(), h_edge + 32 * r, d_out_[r].p, nfirst), "ghf_copy_d2h");
      if (!last && sp.U > sp.F && sp.E > sp.U)
        out.check(ghf_copy_d2h(out.ctx(), h_edge + 32 * r + 16, d_out_[r].u8() + (sp.U - sp.F), sp.E - sp.U), "ghf_copy_d2h");
      if (last && sp.E > sp.F)
        out.check(ghf_copy_d2h(out.ctx(), const_cast<uint8_t*>(h_last), d_out_[r].u8() + (sp.E - 1 - sp.F), 1), "ghf_copy_d2h");
      edged_[r].record(out);
      if (sp.U > sp.F + 16) pipe().drain(emitted_[r], d_out_[r].u8() + 16, sp.U - sp.F - 16, fd, sp.F + 16);
      fetched_[r].record(out);
      prev = sp;
      start += total;
    }
    close_span(prev.r, prev.F, prev.U, prev.E, prev.last, edge, h_edge, fd);
    last_byte_ = prev.E > prev.F ? h_last[0] : 0;
    pipe().close_sink(prev.E);
    run.sync("encode_file");
    if (prev.E != out_bytes_) throw Error(GHF_E_CORRUPT, "encode_file: the pieces do not add up to the size the counts give");
    fseek(outfile_, 0, SEEK_END);
    return end;
  }

  static const int kRing = 3;
  mutable std::shared_ptr<Pipe> pipe_;  // first member: the buffers below are freed through its contexts
  FILE* infile_;
  FILE* outfile_;
  std::string infile_name_;
  size_t n_, out_bytes_ = 0;
  uint8_t last_byte_ = 0;
  DeviceBuf d_hist_;

 private:
  PieceEncoder(const PieceEncoder&);
  PieceEncoder& operator=(const PieceEncoder&);
  uint8_t* piece_at(size_t k) const { return resident_ ? d_in_.u8() + k * pipe().piece_bytes() : d_ring_[k % kRing].u8(); }
  // While pass 1 still reads the file, a pool thread already makes the output file's pages (Pipe::presize): 97 % of
  // the order-0 entropy of the first piece, times the file size.  An estimate only -- pass 2 sizes the file
  // exactly (the rest of the pages, or a cut) before it writes a byte.
  void presize_from_first_piece() {
    first_counted_.sync();
    const volatile uint64_t* h = static_cast<const uint64_t*>(h_first_.p);
    double total = 0, bits = 0;
    for (int b = 0; b < 256; ++b) total += (double)h[b];
    for (int b = 0; b < 256; ++b)
      if (h[b]) bits -= (double)h[b] * log2((double)h[b] / total);
    if (total <= 0) return;
    double per_symbol = bits / total;
    if (per_symbol < 1) per_symbol = 1;  // no code is shorter than one bit
    pipe().presize(fileno(outfile_), (size_t)(0.97 * per_symbol / 8 * (double)n_));
  }
  // the piece in ring slot r has reached the host as far as the host needs it: settle the unit it shares with its
  // predecessor, write it out if the piece has moved past it, and open the next one
  void close_span(int r, size_t F, size_t U, size_t E, bool last, uint8_t* edge, const uint8_t* h_edge, int fd) {
    edged_[r].sync();
    const uint8_t* first = h_edge + 32 * r;
    const size_t nfirst = E - F < 16 ? E - F : 16;
    for (size_t i = 0; i < nfirst; ++i) edge[i] |= first[i];
    if (last) {
      pwrite_all(fd, edge, nfirst, F);
    } else if (U > F) {
      pwrite_all(fd, edge, 16, F);
      memset(edge, 0, 16);
      memcpy(edge, first + 16, E - U);
    }  // else the piece ends inside the unit it began in: the unit stays open
  }
  bool resident_ = true;
  PinnedBuf h_scal_, h_first_;
  DeviceBuf d_in_, d_ring_[kRing], d_out_[kRing], d_scal_;
  Event arrived_[kRing], used_[kRing], planned_[kRing], emitted_[kRing], fetched_[kRing], edged_[kRing], first_counted_;
};

// What the two formats' decoders share: the files and the piece loop.
class PieceDecoder {
 protected:
  // include/encoder.h:227-232: the output name defaults to <in>.de and is handed back
  PieceDecoder(const std::string& infile_name, std::string& outfile_name) : n_(0) {
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ".de";
    outfile_ = detail::open_output(outfile_name);  // read-write: the pipeline maps it
    if (!outfile_) {
      fclose(infile_);
      throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
    }
  }
  ~PieceDecoder() {
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
  }
  Pipe& pipe() const {
    if (!pipe_) pipe_ = Pipe::shared();
    return *pipe_;
  }

  // The body -- `body` bytes of the file from offset `base`, then tail_n more at `tail` -- is cut at byte positions into
  // pieces.  Neither format has sync points, but the bit at which piece k's last code ends is where piece k+1's first
  // code begins, so one K6 pass per piece (`sync`: exact first bit in, landing bit and symbol count out) both rebuilds
  // the piece's side-car and hands the cut to the next piece; K7 (`decode`) then decodes the piece block-parallel.
  // Pieces k+1.. are read and copied in while K6/K7 run on piece k, and the decoded bytes leave through the copy-out
  // stream and the writer threads at their own file offset.
  //   sync(d_piece, padded, first_bit, end_bit, &landing, &nsym, &has_end), decode(d_piece, padded, d_dst, cap)
  //   spare_bits   : bits at the end of the body's last byte that are no code bits (.crs: left_bits)
  //   needs_mark   : the stream ends at an end mark (canonical) -- else exactly at its last bit (.crs)
  template <class Sync, class Decode>
  void decode_pieces(size_t base, size_t body, const uint8_t* tail, size_t tail_n, uint32_t spare_bits, bool needs_mark, Sync sync,
                     Decode decode) {
    pipe().begin();
    const Session &run = pipe().run(), &out = pipe().out();
    const size_t P = pipe().piece();
    pipe().open_feed(fileno(infile_), base, body, 16, tail, tail_n);
    // a compressed piece decodes to at most 8 symbols per byte (no code is shorter than one bit)
    const size_t np = pipe().pieces(), in_cap = P + 64, out_cap = 8 * P + 64;
    for (int r = 0; r < kRing; ++r)
      if (d_in_[r].n < in_cap) d_in_[r].alloc(run, in_cap);
    const int fd = fileno(outfile_);
    pipe().open_sink(fd, 8 * (body + tail_n) + 16, false);  // the formats do not say how much comes out
    // Where the decoded bytes wait for the file.  Making the output file's pages is the slowest stage (20 GB/s, and no
    // copy into the file may run meanwhile), and how many are needed is only known at the end of the stream.  So: the
    // first piece tells the ratio; if the estimated output fits GHF_RESIDENT_BYTES it ALL stays in HBM (`whole_`) while
    // a pool thread makes 97 % of the estimated pages, and it leaves in one sweep at the end, when the exact size is known
    // (sweeping the pieces whose pages exist while the rest is still being made was tried: page-making and copying
    // into the same file get in each other's way, 4 GiB took 480-700 ms instead of 410-470).
    // Otherwise (or once `whole_` is full) a piece leaves as soon as it is decoded, through the kOut-deep ring -- also when
    // the output file is written over (GHF_SINK=reuse: its pages exist, so the copies out overlap the copies in; 4 GiB
    // uniform 191 instead of 236 ms, profiles/r03/pipe_trace_reuse_uniform*.log).
    struct Held {
      const uint8_t* d;
      size_t bytes, at;
    };
    std::vector<Held> held;
    size_t whole_used = 0;
    bool resident = false;
    size_t fed = 0, out_off = 0;
    uint32_t first = 0;  // the body begins on a byte
    StepSum t_sync("K6 (sync_piece)"), t_mem("hipMalloc (output)");
    bool done = false;
    uint64_t landing = 0;
    const double loop_t0 = StepTimer::now();
    for (size_t k = 0; k < np && !done; ++k) {
      for (; fed < np && fed < k + kRing - 1; ++fed) {  // pieces k+1.. arrive while this one is worked on
        const int q = (int)(fed % kRing);
        used_[q].hold(pipe().in());
        pipe().feed(d_in_[q].u8(), arrived_[q]);
      }
      const int r = (int)(k % kRing), o = (int)(k % kOut);
      arrived_[r].hold(run);
      uint64_t nsym = 0;
      int has_end = 0;
      const uint64_t end_bit = 8 * (uint64_t)pipe().own(k) - (k + 1 == np ? spare_bits : 0u);
      {
        StepSum::Scope t(t_sync);
        sync(d_in_[r].u8(), pipe().padded(k), first, end_bit, &landing, &nsym, &has_end);  // K6; the call waits for its own result
      }
      if (nsym > out_cap) throw Error(GHF_E_CORRUPT, "a piece decodes to more than 8 symbols per byte");
      if (k == 0 && np > 2 && !has_end && Pipe::wants_map(8 * body) && !detail::sink_reuses()) {
        const double est = (double)nsym / (double)pipe().own(0) * (double)(body + tail_n);
        const size_t want = (size_t)(est * 1.05) + 4 * out_cap;
        if (want <= env_bytes("GHF_RESIDENT_BYTES", (size_t)64 << 30)) {
          try {
            StepSum::Scope t(t_mem);
            if (whole_.n < want) whole_.alloc(run, want);
            resident = true;
            pipe().presize(fd, (size_t)(est * 0.97));
          } catch (const Error&) {
            resident = false;
          }
        }
      }
      if (resident && whole_used + nsym > whole_.n) {  // the estimate was too low: what is held leaves now, the rest streams
        resident = false;
        pipe().cancel_presize();
        pipe().adopt_presized(fd);
        sweep(held, fd);
      }
      uint8_t* dst = resident ? whole_.u8() + whole_used : NULL;
      if (nsym) {
        if (!resident) {
          if (d_out_[o].n < out_cap) d_out_[o].alloc(run, out_cap);
          fetched_[o].hold(run);  // the D2H of piece k - kOut has let go of d_out_[o]
          dst = d_out_[o].u8();
        }
        decode(d_in_[r].u8(), pipe().padded(k), dst, out_cap);  // K7
      }
      used_[r].record(run);
      if (nsym && resident) {
        const Held h = {dst, (size_t)nsym, out_off};
        held.push_back(h);
        whole_used += ((size_t)nsym + 255) & ~(size_t)255;
      } else if (nsym) {
        decoded_[o].record(run);
        pipe().drain(decoded_[o], dst, (size_t)nsym, fd, out_off);
        fetched_[o].record(out);
      }
      out_off += (size_t)nsym;
      first = (uint32_t)landing;
      done = has_end != 0;
    }
    if (getenv("GHF_PIPE_TRACE")) fprintf(stderr, "[ghf]   %-18s %9.2f ms\n", "piece loop", StepTimer::now() - loop_t0);
    if (resident) {
      pipe().cancel_presize();  // (it has long finished, or the estimate was too high: the exact size is known now)
      pipe().adopt_presized(fd);
      pipe().reserve_all(out_off);
      sweep(held, fd);
    }
    pipe().close_sink(out_off);
    run.sync("decode_file");
    fseek(outfile_, 0, SEEK_END);
    if (needs_mark && !done) throw Error(GHF_E_CORRUPT, "the stream ends before the end mark");
    if (!needs_mark && np && landing != 0) throw Error(GHF_E_CORRUPT, "the body does not end on a code boundary");
  }

  mutable std::shared_ptr<Pipe> pipe_;  // first member: the buffers below are freed through its contexts
  FILE* infile_;
  FILE* outfile_;
  size_t n_;

 private:
  PieceDecoder(const PieceDecoder&);
  PieceDecoder& operator=(const PieceDecoder&);
  static const int kRing = 3, kOut = 2;
  // everything that was decoded into `whole_` so far goes to the file
  template <typename V>
  void sweep(V& held, int fd) {
    swept_.record(pipe().run());
    for (size_t i = 0; i < held.size(); ++i) pipe().drain(swept_, held[i].d, held[i].bytes, fd, held[i].at);
    held.clear();
  }
  DeviceBuf d_in_[kRing], d_out_[kOut], whole_;
  Event arrived_[kRing], used_[kRing], decoded_[kOut], fetched_[kOut], swept_;
};

}  // namespace detail

// ------------------------------------------------------------------------------------------------ encoder policy
template <typename _KeyType = unsigned char>
class HipCanonicalHuffEncoder;

template <>
class HipCanonicalHuffEncoder<unsigned char> : private detail::PieceEncoder {
 public:
  HipCanonicalHuffEncoder(const std::string& infile_name, std::string& outfile_name) { set_file(infile_name, outfile_name); }
  HipCanonicalHuffEncoder() {}
  ~HipCanonicalHuffEncoder() { clear(); }

  // include/canonical_huff_encoder.cc:15-32: the output name defaults to <in>.crs2 and is handed back
  void set_file(const std::string& infile_name, std::string& outfile_name) { open_files(infile_name, outfile_name, ".crs2"); }
  void clear() { close_files(); }  // include/encoder.h:85-92

  // include/encoder.h:99-105,123-150: pass 1 of the file pipeline (detail::PieceEncoder::count_pieces)
  void caculate_frequency() {
    detail::StepTimer timer_("caculate_frequency");
    n_ = detail::file_size(infile_);
    // opt-in: set_allow_empty(true) or GHF_EMPTY_OK=1 gives the empty file a defined encoding (include/ghf.h,
    // GHF_EMPTY_OK: parity unpinned); without it the input on which the reference is undefined is refused
    const char* eenv = getenv("GHF_EMPTY_OK");
    empty_ = n_ == 0 && (allow_empty_ || (eenv && eenv[0] == '1'));
    if (empty_) return;
    if (n_ == 0) throw Error(GHF_E_EMPTY, "empty input: undefined in the reference, refused here");
    count_pieces();
  }

  // include/canonical_huff_encoder.cc:35-42
  void gen_encode() {
    detail::StepTimer timer_("gen_encode");
    if (empty_) return;  // the whole 1049-byte stream comes from ghf_compress_ex in write_encode_info
    const detail::Session& run = pipe().run();
    if (!d_code_.p) d_code_.alloc(run, sizeof(ghf_code));
    // opt-in: set_code_limit(true) or GHF_CODE_LIMIT=1 in the environment replaces the reference's
    // "undefined above 32 bits" by the optimal 32-bit-limited code; without it the behaviour is the reference's
    const char* env = getenv("GHF_CODE_LIMIT");
    const unsigned flags = (limit_ || (env && env[0] == '1')) ? GHF_CODE_LIMIT : 0u;
    run.check(ghf_build_code_ex(run.ctx(), static_cast<const uint64_t*>(d_hist_.p), static_cast<ghf_code*>(d_code_.p), flags),
              "ghf_build_code");
    run.check(ghf_copy_d2h(run.ctx(), &code_, d_code_.p, sizeof(ghf_code)), "ghf_copy_d2h");
    uint64_t hist[GHF_NSYM];
    run.check(ghf_copy_d2h(run.ctx(), hist, d_hist_.p, sizeof hist), "ghf_copy_d2h");
    run.sync("gen_encode");
    uint64_t bits = code_.length[256];  // the body's size follows from the counts: the output file is sized up front
    for (int b = 0; b < 256; ++b) bits += hist[b] * code_.length[b];
    out_bytes_ = ghf_header_bytes(code_.max_len) + (size_t)((bits + 7) >> 3);
  }

  // include/canonical_huff_encoder.cc:210-242: header at file offset 0, flushed before the body is produced
  void write_encode_info() {
    detail::StepTimer timer_("write_encode_info");
    const detail::Session& run = pipe().run();
    if (!d_hdr_.p) d_hdr_.alloc(run, 2048);  // the largest header is 1040 + 8 * 32 = 1296 bytes
    if (!h_hdr_.p) h_hdr_.alloc(run, 2048);
    size_t hdr;
    if (empty_) {
      if (!d_code_.p) d_code_.alloc(run, sizeof(ghf_code));
      run.check(ghf_compress_ex(run.ctx(), NULL, 0, d_hdr_.u8(), d_hdr_.n, NULL, static_cast<ghf_code*>(d_code_.p), NULL, GHF_EMPTY_OK),
                "ghf_compress_ex");
      hdr = out_bytes_ = ghf_header_bytes(1) + 1;  // (the one body byte included)
      run.check(ghf_copy_d2h(run.ctx(), &code_, d_code_.p, sizeof(ghf_code)), "ghf_copy_d2h");
    } else {
      hdr = ghf_header_bytes(code_.max_len);
      run.check(ghf_write_header(run.ctx(), static_cast<const ghf_code*>(d_code_.p), d_hdr_.u8(), d_hdr_.n), "ghf_write_header");
    }
    run.check(ghf_copy_d2h(run.ctx(), h_hdr_.p, d_hdr_.p, hdr), "ghf_copy_d2h");
    run.sync("write_encode_info");
    fseek(outfile_, 0, SEEK_SET);
    if (fwrite(h_hdr_.p, 1, hdr, outfile_) != hdr) throw Error(GHF_E_INVAL, "short write (header)");
    fflush(outfile_);
  }

  // include/canonical_huff_encoder.cc:245-285: pass 2 of the file pipeline (detail::PieceEncoder::emit_pieces)
  void encode_file() {
    detail::StepTimer timer_("encode_file");
    if (empty_) return;
    try {
      emit_pieces(static_cast<const ghf_code*>(d_code_.p), h_hdr_.u8(), ghf_header_bytes(code_.max_len), GHF_EMIT_LAST, code_.length[256]);
    } catch (...) {
      pipe().abandon_sink();
      throw;
    }
  }

  const ghf_code& code() const { return code_; }  // length_/codeword_/symbol_/... of the reference, for tests
  void set_code_limit(bool on) { limit_ = on; }   // not in the reference: see gen_encode()
  void set_allow_empty(bool on) { allow_empty_ = on; }  // not in the reference: see caculate_frequency()

 private:
  bool limit_ = false, allow_empty_ = false, empty_ = false;
  detail::PinnedBuf h_hdr_;
  detail::DeviceBuf d_code_, d_hdr_;
  ghf_code code_;
};

// ------------------------------------------------------------------------------------------------ decoder policy
template <typename _KeyType = unsigned char>
class HipCanonicalHuffDecoder;

template <>
class HipCanonicalHuffDecoder<unsigned char> : private detail::PieceDecoder {
 public:
  HipCanonicalHuffDecoder(const std::string& infile_name, std::string& outfile_name) : detail::PieceDecoder(infile_name, outfile_name), hdr_(0) {}

  // include/canonical_huff_encoder.cc:349-374 -- plus the validation the reference does not do
  void get_encode_info() {
    detail::StepTimer timer_("get_encode_info");
    n_ = detail::file_size(infile_);
    std::vector<uint8_t> head(n_ < 1296 ? n_ : 1296);  // the largest header: 1040 + 8 * 32
    if (fread(head.data(), 1, head.size(), infile_) != head.size()) throw Error(GHF_E_INVAL, "short read");
    pipe().run().check(ghf_parse_header(head.data(), head.size(), &code_, &hdr_), "ghf_parse_header");
    fseek(infile_, 0, SEEK_SET);
  }

  // include/canonical_huff_encoder.cc:377-419 (and :422-461, :519-568): runs until the end mark, piece by piece
  // (detail::PieceDecoder::decode_pieces with ghf_sync_piece as K6 and ghf_decode as K7)
  void decode_file() {
    detail::StepTimer timer_("decode_file");
    try {
      const detail::Session& run = pipe().run();
      if (!d_code_.p) d_code_.alloc(run, sizeof(ghf_code));
      run.check(ghf_copy_h2d(run.ctx(), d_code_.p, &code_, sizeof(ghf_code)), "ghf_copy_h2d");
      const ghf_code* dc = static_cast<const ghf_code*>(d_code_.p);
      decode_pieces(
          hdr_, n_ - hdr_, NULL, 0, 0, true,
          [&](const uint8_t* d_piece, size_t bytes, uint32_t first, uint64_t end_bit, uint64_t* landing, uint64_t* nsym, int* has_end) {
            run.check(ghf_sync_piece(run.ctx(), d_piece, bytes, first, end_bit, dc, landing, nsym, has_end), "ghf_sync_piece");
          },
          [&](const uint8_t* d_piece, size_t bytes, uint8_t* d_dst, size_t cap) {
            run.check(ghf_decode(run.ctx(), d_piece, bytes, dc, NULL, d_dst, cap, NULL), "ghf_decode");
          });
    } catch (...) {
      pipe().abandon_sink();
      throw;
    }
  }

 private:
  size_t hdr_;
  detail::DeviceBuf d_code_;
  ghf_code code_;
};

// The reference has three interchangeable decoders that all emit the same bytes (bit-serial, left-justified
// linear search, 8-bit length table).  On the GPU they are one kernel (direct table + the linear extension as
// its fallback), so the other two names are the same class.
template <typename _KeyType = unsigned char>
class HipFastCanonicalHuffDecoder : public HipCanonicalHuffDecoder<_KeyType> {
 public:
  HipFastCanonicalHuffDecoder(const std::string& in, std::string& out) : HipCanonicalHuffDecoder<_KeyType>(in, out) {}
};
template <typename _KeyType = unsigned char, int TableLength = 8>
class HipTableCanonicalHuffDecoder : public HipCanonicalHuffDecoder<_KeyType> {
 public:
  HipTableCanonicalHuffDecoder(const std::string& in, std::string& out) : HipCanonicalHuffDecoder<_KeyType>(in, out) {}
};

// ================================================================================================
// the .crs format.  Same member functions, same order (Compressor<>::compress() and
// Decompressor<>::decompress() above do not change): drop-ins for NormalHuffEncoder<> / NormalHuffDecoder<>
// (include/normal_huff_encoder.h:57-283).  Same file pipeline as the canonical pair: the two passes of
// detail::PieceEncoder with the tree's codes, detail::PieceDecoder with ghf_crs_sync_piece as its K6.
// ================================================================================================
template <typename _KeyType = unsigned char>
class HipNormalHuffEncoder;

template <>
class HipNormalHuffEncoder<unsigned char> : private detail::PieceEncoder {
 public:
  HipNormalHuffEncoder(const std::string& infile_name, std::string& outfile_name) { set_file(infile_name, outfile_name); }
  HipNormalHuffEncoder() {}
  ~HipNormalHuffEncoder() { clear(); }

  // include/normal_huff_encoder.h:83-99: the output name defaults to <in>.crs and is handed back
  void set_file(const std::string& infile_name, std::string& outfile_name) { open_files(infile_name, outfile_name, ".crs"); }
  void clear() { close_files(); }  // include/encoder.h:85-92

  // include/encoder.h:99-105,136-150 (no end-mark slot in this format: init_nhuff, normal_huff_encoder.h:189-196;
  // K1's slot [256] is simply not looked at)
  void caculate_frequency() {
    detail::StepTimer timer_("caculate_frequency");
    n_ = detail::file_size(infile_);
    if (n_ == 0) throw Error(GHF_E_EMPTY, "empty input: undefined in the reference, refused here");
    count_pieces();
  }

  // include/normal_huff_encoder.h:110-121 -> EncodeHuffTree::build_tree + gen_encode (include/huff_tree.cc:138-171)
  void gen_encode() {
    detail::StepTimer timer_("gen_encode");
    const detail::Session& run = pipe().run();
    if (!d_tree_.p) d_tree_.alloc(run, sizeof(ghf_tree));
    if (!d_code_.p) d_code_.alloc(run, sizeof(ghf_code));
    run.check(ghf_crs_build_code(run.ctx(), static_cast<const uint64_t*>(d_hist_.p), static_cast<ghf_tree*>(d_tree_.p),
                                 static_cast<ghf_code*>(d_code_.p)),
              "ghf_crs_build_code");
    run.check(ghf_copy_d2h(run.ctx(), &tree_, d_tree_.p, sizeof(ghf_tree)), "ghf_copy_d2h");
    ghf_code code;
    uint64_t hist[GHF_NSYM];
    run.check(ghf_copy_d2h(run.ctx(), &code, d_code_.p, sizeof code), "ghf_copy_d2h");
    run.check(ghf_copy_d2h(run.ctx(), hist, d_hist_.p, sizeof hist), "ghf_copy_d2h");
    run.sync("gen_encode");
    body_bits_ = 0;  // the body's size follows from the counts: the output file is sized up front
    for (int b = 0; b < 256; ++b) body_bits_ += hist[b] * code.length[b];
    long_codes_ = code.max_len > 32;
    out_bytes_ = (size_t)tree_.tree_bytes + 2 + (size_t)((body_bits_ + 7) >> 3);
  }

  // include/normal_huff_encoder.h:136-138 -> serialize_tree (include/huff_tree.cc:174-187): the tree at file offset 0
  void write_encode_info() {
    detail::StepTimer timer_("write_encode_info");
    fseek(outfile_, 0, SEEK_SET);
    if (fwrite(tree_.header, 1, tree_.tree_bytes, outfile_) != tree_.tree_bytes) throw Error(GHF_E_INVAL, "short write (tree)");
    fflush(outfile_);
  }

  // include/normal_huff_encoder.h:159-186: two placeholder bytes, the codes, then {left_bits, last byte} go back into the
  // placeholders and the body keeps its whole bytes only
  void encode_file() {
    detail::StepTimer timer_("encode_file");
    try {
      const size_t hdr = (size_t)tree_.tree_bytes + 2;
      uint8_t head[1040];
      memcpy(head, tree_.header, tree_.tree_bytes);
      head[hdr - 2] = head[hdr - 1] = 0;
      if (fwrite(head + hdr - 2, 1, 2, outfile_) != 2) throw Error(GHF_E_INVAL, "short write (prefix)");
      fflush(outfile_);
      const uint64_t end = emit_pieces(static_cast<const ghf_code*>(d_code_.p), head, hdr, 0, 0, long_codes_ ? GHF_EMIT_LONG_CODES : 0);
      if (end != 8 * (uint64_t)hdr + body_bits_) throw Error(GHF_E_CORRUPT, "encode_file: the pieces do not add up to the bits the counts give");
      const size_t whole = (size_t)(body_bits_ >> 3);
      const unsigned left = (unsigned)((8 - (body_bits_ & 7)) & 7);
      const uint8_t prefix[2] = {(uint8_t)left, left ? last_byte_ : (uint8_t)0};  // (K5 zero-fills behind its last bit)
      detail::pwrite_all(fileno(outfile_), prefix, 2, hdr - 2);
      if (ftruncate(fileno(outfile_), (off_t)(hdr + whole)) != 0) throw Error(GHF_E_INVAL, "ftruncate");
      fseek(outfile_, 0, SEEK_END);
    } catch (...) {
      pipe().abandon_sink();
      throw;
    }
  }

  const ghf_tree& tree() const { return tree_; }

 private:
  detail::DeviceBuf d_tree_, d_code_;
  uint64_t body_bits_ = 0;
  bool long_codes_ = false;  // some code has more than 32 bits (the tree is deeper than 32)
  ghf_tree tree_;
};

template <typename _KeyType = unsigned char>
class HipNormalHuffDecoder;

template <>
class HipNormalHuffDecoder<unsigned char> : private detail::PieceDecoder {
 public:
  HipNormalHuffDecoder(const std::string& infile_name, std::string& outfile_name)
      : detail::PieceDecoder(infile_name, outfile_name), tb_(0), left_(0), last_(0) {}

  // include/normal_huff_encoder.h:268-271 -> DecodeHuffTree::build_tree (include/huff_tree.cc:289-303), validated
  void get_encode_info() {
    detail::StepTimer timer_("get_encode_info");
    n_ = detail::file_size(infile_);
    std::vector<uint8_t> head(n_ < 1024 ? n_ : 1024);  // the largest tree: 2 * 511 bytes, then the two prefix bytes
    if (fread(head.data(), 1, head.size(), infile_) != head.size()) throw Error(GHF_E_INVAL, "short read");
    pipe().run().check(ghf_crs_parse_header(head.data(), head.size(), &tree_, &tb_), "ghf_crs_parse_header");
    if (tb_ + 2 > head.size()) throw Error(GHF_E_FORMAT, "the two bytes behind the tree are missing");
    left_ = head[tb_];       // include/huff_tree.cc:195-196
    last_ = head[tb_ + 1];
    if (left_ > 7) throw Error(GHF_E_FORMAT, "left_bits > 7");
    fseek(infile_, 0, SEEK_SET);
  }

  // include/normal_huff_encoder.h:272-274 -> DecodeHuffTree::decode_file (include/huff_tree.cc:191-207): the stored last
  // byte goes behind the body, where its bits belong, and the stream ends left_bits before that byte does
  void decode_file() {
    detail::StepTimer timer_("decode_file");
    try {
      const detail::Session& run = pipe().run();
      if (!d_tree_.p) d_tree_.alloc(run, sizeof(ghf_tree));
      run.check(ghf_copy_h2d(run.ctx(), d_tree_.p, &tree_, sizeof(ghf_tree)), "ghf_copy_h2d");
      const ghf_tree* dt = static_cast<const ghf_tree*>(d_tree_.p);
      decode_pieces(
          tb_ + 2, n_ - (tb_ + 2), &last_, left_ ? 1 : 0, left_, false,
          [&](const uint8_t* d_piece, size_t bytes, uint32_t first, uint64_t end_bit, uint64_t* landing, uint64_t* nsym, int* has_end) {
            *has_end = 0;  // no end mark in this format
            run.check(ghf_crs_sync_piece(run.ctx(), d_piece, bytes, first, end_bit, dt, landing, nsym), "ghf_crs_sync_piece");
          },
          [&](const uint8_t* d_piece, size_t bytes, uint8_t* d_dst, size_t cap) {
            run.check(ghf_crs_decode(run.ctx(), d_piece, bytes, 0, dt, NULL, d_dst, cap, NULL), "ghf_crs_decode");
          });
    } catch (...) {
      pipe().abandon_sink();
      throw;
    }
  }

 private:
  size_t tb_;
  uint8_t left_, last_;
  detail::DeviceBuf d_tree_;
  ghf_tree tree_;
};

}  // namespace glzip_hip
#endif  // GLZIP_HIP_H_
