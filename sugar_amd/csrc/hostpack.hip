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

// ---------------------------------------------------------------------------------------------- wav files
// Replaces, for evaluation lists of 16-bit PCM mono files (VoxCeleb): the reference's per-item Python loader
// (loadWAV, sugar/waveforms.py:47-115: scipy wavfile.read -> wrap padding -> whole utterance or linspace crops ->
// float32) run by DataLoader workers / joblib threads, followed by Utterance.collate_fn's torch.cat.  From Python
// that path is bound by the interpreter lock (4 874 files: 0.16 s for the reads alone, 0.33 s beside the launching
// thread, plus a 3.4 GB copy of the decoded samples into the staging buffer).  Here a batch is ONE call: every
// segment is read with pread straight from its file, converted int16 -> float32 (exact) and written to its place in
// the pinned staging buffer by `n_threads` threads.
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

namespace {

// RIFF/WAVE header walk: 1 = mono 16-bit PCM at `sample_rate` (n_samples / data_off set), 0 = some other format (the
// caller falls back to the Python loader), -1 = unreadable
int wav_probe_one(const char* path, int sample_rate, int64_t* n_samples, int64_t* data_off) {
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return -1;
  unsigned char h[12];
  int ok = -1;
  auto rd32 = [](const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; };
  auto rd16 = [](const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; };
  if (pread(fd, h, 12, 0) == 12 && !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4)) {
    int64_t pos = 12;
    bool fmt_ok = false, fmt_seen = false;
    unsigned char c[8];
    ok = 0;
    while (pread(fd, c, 8, pos) == 8) {
      const int64_t size = rd32(c + 4);
      if (!memcmp(c, "fmt ", 4)) {
        unsigned char f[16];
        if (size < 16 || pread(fd, f, 16, pos + 8) != 16) break;
        fmt_seen = true;
        fmt_ok = rd16(f) == 1 && rd16(f + 2) == 1 && (int)rd32(f + 4) == sample_rate && rd16(f + 14) == 16;
      } else if (!memcmp(c, "data", 4)) {
        if (fmt_seen && fmt_ok) {
          const int64_t end = lseek(fd, 0, SEEK_END);
          int64_t bytes = size;
          if (end >= 0 && pos + 8 + bytes > end) bytes = end - (pos + 8);     // truncated file: what is there (as scipy)
          *n_samples = bytes / 2;
          *data_off = pos + 8;
          ok = bytes >= 2 ? 1 : 0;
        }
        break;
      }
      pos += 8 + size + (size & 1);                    // chunks are word aligned
    }
  }
  close(fd);
  return ok;
}

}  // namespace

extern "C" int sugar_wav_probe(const char* const* h_paths, int n, int sample_rate, int64_t* h_n_samples,
                               int64_t* h_data_off, int n_threads) {
  using namespace sugar;
  SUGAR_CHECK_ARG(h_paths && h_n_samples && h_data_off && n > 0, "wav_probe: bad arguments");
  const int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  std::vector<int> bad(nt, -1);
  auto work = [&](int t) {
    for (int i = t; i < n; i += nt) {
      h_n_samples[i] = -1;
      h_data_off[i] = 0;
      const int r = h_paths[i] ? wav_probe_one(h_paths[i], sample_rate, &h_n_samples[i], &h_data_off[i]) : -1;
      if (r == 0) h_n_samples[i] = -1;
      if (r < 0 && bad[t] < 0) bad[t] = i;
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& t : th) t.join();
  for (int t = 0; t < nt; ++t)
    if (bad[t] >= 0) {
      set_error("wav_probe: cannot read %s", h_paths[bad[t]] ? h_paths[bad[t]] : "(null)");
      return SUGAR_EINVAL;
    }
  return SUGAR_OK;
}

// Segment j = h_seg_len[j] samples of file h_seg_file[j] from sample h_seg_start[j] on, continued periodically past the
// file's end (np.pad(..., mode="wrap"), sugar/waveforms.py:94-98), written to h_dst one segment after the other.
extern "C" int sugar_wav_load(const char* const* h_paths, const int64_t* h_n_samples, const int64_t* h_data_off,
                              int n_files, const int32_t* h_seg_file, const int64_t* h_seg_start, const int64_t* h_seg_len,
                              int n_segs, float* h_dst, int n_threads) {
  using namespace sugar;
  SUGAR_CHECK_ARG(h_paths && h_n_samples && h_data_off && h_seg_file && h_seg_start && h_seg_len && h_dst && n_segs > 0 &&
                      n_files > 0,
                  "wav_load: bad arguments");
  std::vector<int64_t> off((size_t)n_segs + 1, 0);
  for (int j = 0; j < n_segs; ++j) {
    SUGAR_CHECK_ARG(h_seg_file[j] >= 0 && h_seg_file[j] < n_files, "wav_load: segment %d names file %d of %d", j,
                    h_seg_file[j], n_files);
    SUGAR_CHECK_ARG(h_seg_len[j] >= 0 && h_seg_start[j] >= 0 && h_n_samples[h_seg_file[j]] > 0 && h_paths[h_seg_file[j]],
                    "wav_load: segment %d has a bad range or an empty file", j);
    off[j + 1] = off[j] + h_seg_len[j];
  }
  const int64_t total = off[n_segs];
  if (total == 0) return SUGAR_OK;
  int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (total < (int64_t)nt * (1 << 16)) nt = 1;
  std::vector<int> failed(nt, 0);
  // thread t converts destination samples [lo, hi): whole segments where it can, pieces at its borders
  auto work = [&](int t, int64_t lo, int64_t hi) {
    constexpr int64_t BUF = 1 << 16;                   // 128 KiB of int16 per read
    std::vector<int16_t> buf((size_t)BUF);
    int j = 0;
    while (off[j + 1] <= lo) ++j;
    for (; j < n_segs && off[j] < hi; ++j) {
      const int f = h_seg_file[j];
      const int64_t n = h_n_samples[f];
      int64_t a = (lo > off[j] ? lo : off[j]) - off[j], b = (hi < off[j + 1] ? hi : off[j + 1]) - off[j];
      const int fd = open(h_paths[f], O_RDONLY | O_CLOEXEC);
      if (fd < 0) { failed[t] = f + 1; return; }
      float* out = h_dst + off[j];
      while (a < b) {
        const int64_t s = (h_seg_start[j] + a) % n;    // position in the file
        int64_t cnt = b - a;
        if (cnt > n - s) cnt = n - s;                  // up to the end of the file, then wrap
        if (cnt > BUF) cnt = BUF;
        const ssize_t got = pread(fd, buf.data(), (size_t)cnt * 2, (off_t)(h_data_off[f] + s * 2));
        if (got != (ssize_t)(cnt * 2)) { close(fd); failed[t] = f + 1; return; }
        for (int64_t i = 0; i < cnt; ++i) out[a + i] = (float)buf[(size_t)i];
        a += cnt;
      }
      close(fd);
    }
  };
  std::vector<std::thread> th;
  const int64_t per = (total + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int64_t lo = t * per, hi = lo + per < total ? lo + per : total;
    if (lo < hi) th.emplace_back(work, t, lo, hi);
  }
  work(0, 0, per < total ? per : total);
  for (auto& t : th) t.join();
  for (int t = 0; t < nt; ++t)
    if (failed[t]) {
      set_error("wav_load: short read from %s", h_paths[failed[t] - 1]);
      return SUGAR_EINVAL;
    }
  return SUGAR_OK;
}
