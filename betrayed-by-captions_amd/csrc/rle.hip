// Host side of the inference tail ("RLE-ready bit-packing"): COCO run-length encoding of the bit-packed
// instance masks that cgg_instance_masks_picks(bitpack=1) produces, on host threads, straight from the pinned staging
// buffer of the one device->host copy per batch.
//
// The reference returns every mask as an (H, W) bool array (open_set/models/maskformer.py:205-208: one `.cpu().numpy()`
// per mask, 1 MB each at 1024^2) and the evaluation path then calls pycocotools `mask.encode` on each of them
// (open_set/datasets/coco_open.py results2json -> segm). Here the (n, H, W/8) bit planes are encoded directly:
//   * the format is the published COCO RLE (pycocotools maskApi.c, rleEncode / rleToString): runs over the mask in
//     COLUMN-major order starting with a run of zeros, counts delta-coded against counts[i-2] (i > 2) and written as
//     5-bit groups + continuation bit, offset 48 -- restated from the published algorithm, pycocotools is not a dependency;
//   * the planes are row-major with pixel x in bit (x & 7) of byte (x >> 3), so each 64 x 64 block is bit-transposed
//     (6 butterfly rounds on 64-bit words) and the runs are then peeled off the column words with count-trailing-zeros:
//     cost ~ H W / 64 word operations + the number of runs, not H W bit tests.
#include "cgg_common.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

// in-place transpose of a 64 x 64 bit matrix: a[r] bit c  <->  a[c] bit r  (LSB-first on both sides)
inline void transpose64(uint64_t a[64]) {
  uint64_t m = 0x00000000FFFFFFFFull;
  for (int j = 32; j != 0; j >>= 1, m ^= (m << j)) {
    for (int k = 0; k < 64; k = (k + j + 1) & ~j) {
      const uint64_t t = ((a[k] >> j) ^ a[k + j]) & m;
      a[k] ^= (t << j);
      a[k + j] ^= t;
    }
  }
}

struct RunWriter {
  std::string out;
  long long c2 = 0, c1 = 0;   // counts[i-2], counts[i-1]
  long long n = 0;
  void emit(long long cnt) {
    long long x = cnt;
    if (n > 2) x -= c2;
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      out.push_back(c);
    }
    c2 = c1;
    c1 = cnt;
    ++n;
  }
};

void encode_one(const uint8_t* bits, int H, int W, int row_bytes, std::string& out) {
  RunWriter rw;
  rw.out.reserve(4096);
  const int HB = (H + 63) / 64;
  std::vector<uint64_t> col((size_t)64 * HB);       // col[c * HB + by]: rows by*64 .. of column (block column c)
  int p = 0;               // value of the current run
  long long cnt = 0;       // its length so far
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int ncols = W - x0 < 64 ? W - x0 : 64;
    const int nb = (ncols + 7) / 8;                  // bytes of this block column present in a row
    for (int by = 0; by < HB; ++by) {
      uint64_t blk[64];
      uint64_t any = 0, all = ~0ull;
      const int rows = H - by * 64 < 64 ? H - by * 64 : 64;
      const uint8_t* src = bits + (size_t)(by * 64) * row_bytes + (x0 >> 3);
      if (nb == 8) {
        for (int r = 0; r < rows; ++r) {
          uint64_t w;
          std::memcpy(&w, src + (size_t)r * row_bytes, 8);       // one unaligned 8-byte load
          blk[r] = w;
          any |= w;
          all &= w;
        }
      } else {
        for (int r = 0; r < rows; ++r) {
          uint64_t w = 0;
          std::memcpy(&w, src + (size_t)r * row_bytes, (size_t)nb);
          blk[r] = w;
          any |= w;
          all &= w;
        }
      }
      for (int r = rows; r < 64; ++r) blk[r] = 0;
      // object masks are mostly uniform blocks: an all-zero / all-one block is its own transpose
      if (any == 0) {
        for (int c = 0; c < ncols; ++c) col[(size_t)c * HB + by] = 0;
      } else if (all == ~0ull && rows == 64) {
        for (int c = 0; c < ncols; ++c) col[(size_t)c * HB + by] = ~0ull;
      } else {
        transpose64(blk);
        for (int c = 0; c < ncols; ++c) col[(size_t)c * HB + by] = blk[c];
      }
    }
    for (int c = 0; c < ncols; ++c) {
      for (int by = 0; by < HB; ++by) {
        uint64_t w = col[(size_t)c * HB + by];
        int left = H - by * 64 < 64 ? H - by * 64 : 64;
        while (left > 0) {
          uint64_t diff = p ? ~w : w;                // bits that differ from the current run's value
          if (left < 64) diff &= (~0ull >> (64 - left));
          if (diff == 0) {
            cnt += left;
            break;
          }
          const int tz = __builtin_ctzll(diff);
          cnt += tz;
          rw.emit(cnt);
          cnt = 0;
          p ^= 1;
          w = tz ? (w >> tz) : w;
          left -= tz;
        }
      }
    }
  }
  rw.emit(cnt);
  out.swap(rw.out);
}

// Persistent helper pool: `run(k, f)` executes f on up to k pool threads AND on the caller, returns when all have finished.
// Several callers may be inside run() at once (the collector encodes `depth` batches concurrently): jobs queue up and idle
// workers take them; a caller never waits for a worker to become free -- it runs f itself, the helpers only add parallelism.
class RlePool {
 public:
  void run(int helpers, const std::function<void()>& f) {
    struct Job {
      const std::function<void()>* f;
      std::atomic<int> pending{0};
      std::mutex m;
      std::condition_variable cv;
    };
    auto job = std::make_shared<Job>();
    job->f = &f;
    {
      std::lock_guard<std::mutex> g(m_);
      if (owner_pid_ != getpid()) {                   // a forked child inherits `workers_` but none of the threads: start over
        workers_.clear();
        queue_.clear();
        owner_pid_ = getpid();
      }
      grow(helpers);
      const int k = helpers < (int)workers_.size() ? helpers : (int)workers_.size();
      // `pending` counts the tasks that were actually queued: a push_back that throws (bad_alloc) leaves the ones queued so far
      // valid (they hold `job` by shared_ptr and f outlives them: this function does not return before pending == 0)
      for (int i = 0; i < k; ++i) {
        try {
          job->pending.fetch_add(1);
          queue_.push_back([job]() {
            (*job->f)();
            if (job->pending.fetch_sub(1) == 1) {
              std::lock_guard<std::mutex> g2(job->m);
              job->cv.notify_all();
            }
          });
        } catch (...) {
          job->pending.fetch_sub(1);
          break;
        }
      }
    }
    cv_.notify_all();
    f();                                              // the caller works too
    std::unique_lock<std::mutex> lk(job->m);
    job->cv.wait(lk, [&] { return job->pending.load() == 0; });
  }

 private:
  void grow(int want) {                               // m_ held
    if (want > 64) want = 64;
    while ((int)workers_.size() < want) {
      try {
        workers_.emplace_back([this]() {
          for (;;) {
            std::function<void()> task;
            {
              std::unique_lock<std::mutex> lk(m_);
              cv_.wait(lk, [this] { return !queue_.empty(); });
              task = std::move(queue_.front());
              queue_.pop_front();
            }
            task();
          }
        });
        workers_.back().detach();                     // process-lifetime workers
      } catch (...) {
        break;                                        // fewer helpers: the callers still make progress
      }
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  std::vector<std::thread> workers_;
  pid_t owner_pid_ = getpid();
};

RlePool& rle_pool() {
  static RlePool* p = new RlePool();                  // never destroyed: detached workers may outlive static destruction
  return *p;
}

}  // namespace

extern "C" int64_t cgg_rle_encode_bitmasks(const uint8_t* bits, int n, int H, int W, int64_t mask_stride_bytes,
                                           int row_bytes, int threads, uint8_t* out, int64_t out_cap, int64_t* offsets) {
  if (!bits || !offsets || n < 0 || H <= 0 || W <= 0 || row_bytes * 8 < W || mask_stride_bytes < (int64_t)H * row_bytes) {
    cgg_set_error("cgg_rle_encode_bitmasks: bad arguments (n=%d H=%d W=%d row_bytes=%d)", n, H, W, row_bytes);
    return -(int64_t)CGG_EINVAL;
  }
  try {                                   // no C++ exception may cross the C ABI (std::thread / allocation failures)
  std::vector<std::string> enc((size_t)n);
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  std::atomic<int> next(0);
  std::atomic<bool> failed(false);
  auto work = [&]() {
    try {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n) break;
        encode_one(bits + (size_t)i * mask_stride_bytes, H, W, row_bytes, enc[(size_t)i]);
      }
    } catch (...) {
      failed.store(true);                 // an exception must not leave a worker thread either
    }
  };
  // round 4: the helpers come from a PERSISTENT pool (created at the first call, sized by the largest request so far, capped at
  // 64): with one std::thread per helper and call the serving loop's three concurrent encoders started ~190 threads per batch and a
  // 100-mask call took 16-21 ms instead of ~2 (measured, scratch/host_probe.py)
  if (threads == 1) {
    work();
  } else {
    rle_pool().run(threads - 1, work);
  }
  if (failed.load()) {                    // (dropped with the move to the pool in round 4: a failed mask came back as an empty RLE)
    cgg_set_error("cgg_rle_encode_bitmasks: encoding failed (allocation failure inside a worker)");
    return -(int64_t)CGG_EINVAL;
  }
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    offsets[i] = total;
    total += (int64_t)enc[(size_t)i].size();
  }
  offsets[n] = total;
  if (out != nullptr && total <= out_cap) {
    for (int i = 0; i < n; ++i) std::memcpy(out + offsets[i], enc[(size_t)i].data(), enc[(size_t)i].size());
  }
  return total;     // > out_cap: nothing was copied, call again with a larger buffer
  } catch (const std::exception& e) {
    cgg_set_error("cgg_rle_encode_bitmasks: %s", e.what());
    return -(int64_t)CGG_EINVAL;
  } catch (...) {
    cgg_set_error("cgg_rle_encode_bitmasks: unknown C++ exception");
    return -(int64_t)CGG_EINVAL;
  }
}
