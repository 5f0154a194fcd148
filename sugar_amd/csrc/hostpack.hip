// Host-side packing of a ragged batch into one staging buffer (no device code).
//
// Replaces: Utterance.collate_fn's torch.cat of the per-utterance segment stacks (sugar/database.py:144-158) for
// waveforms that live in host memory: extract_vectors (sugar_amd/vectors.py::_HostFeed) packs batch k + 1 into a pinned
// buffer while the GPU computes batch k.  From Python the copies were either serial (tensor.copy_: 10.8 GB/s) or, through
// one ctypes memmove per segment from a thread pool, parallel but with ~5 000 trips through the interpreter lock that
// the kernel-launching thread also needs.  Here the whole batch is ONE call without the lock: `n_threads` std::threads
// take contiguous byte ranges of the destination, so every thread copies the same amount whatever the segment lengths.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "common.h"

extern "C" int sugar_pack_segments(const float* const* h_src, const int64_t* h_len, int n, float* h_dst, int n_threads) {
  using namespace sugar;
  SUGAR_CHECK_ARG(h_src && h_len && h_dst && n > 0, "pack_segments: bad arguments");
  std::vector<int64_t> off((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    SUGAR_CHECK_ARG(h_src[i] && h_len[i] >= 0, "pack_segments: segment %d is null or has a negative length", i);
    off[i + 1] = off[i] + h_len[i];
  }
  const int64_t total = off[n];
  if (total == 0) return SUGAR_OK;
  int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (total < (int64_t)nt * (1 << 16)) nt = 1;                  // small batches: not worth the thread start
  auto copy_range = [&](int64_t lo, int64_t hi) {                // destination elements [lo, hi)
    int i = 0;
    while (off[i + 1] <= lo) ++i;
    for (; i < n && off[i] < hi; ++i) {
      const int64_t a = lo > off[i] ? lo : off[i], b = hi < off[i + 1] ? hi : off[i + 1];
      if (b > a) memcpy(h_dst + a, h_src[i] + (a - off[i]), (size_t)(b - a) * sizeof(float));
    }
  };
  if (nt == 1) {
    copy_range(0, total);
    return SUGAR_OK;
  }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  const int64_t per = (total + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int64_t lo = t * per, hi = lo + per < total ? lo + per : total;
    if (lo < hi) th.emplace_back(copy_range, lo, hi);
  }
  copy_range(0, per < total ? per : total);
  for (auto& t : th) t.join();
  return SUGAR_OK;
}
