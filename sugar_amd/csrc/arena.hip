// Device memory of a subnet's packed weights (gfx950 host code).
//
// sugar_subnet_create used to issue ~130 hipMalloc + synchronous hipMemcpy pairs per subnet (one per packed tensor) and
// a hipDeviceSynchronize, and sugar_subnet_destroy as many hipFree calls.  On this runtime every one of them is a
// driver round trip, the synchronous copies go through the legacy null stream (which torch's default stream is, so
// they queue behind whatever the caller has in flight there) and hipFree / hipDeviceSynchronize wait for the whole
// device: a NAS sweep that packs subnet i + 1 on a host thread while subnet i is being evaluated (sugar_amd/sweep.py)
// stalled on them -- 12 ms of GPU work per subnet took 22 ms.
//
// Inside an ArenaScope the packing code's dev_alloc calls are carved out of a few large chunks owned by the subnet,
// dev_upload writes into a host mirror of the chunk, and arena_flush sends each chunk with ONE asynchronous copy on a
// private non-blocking stream and waits for that stream only.  dev_free is a no-op for arena memory; arena_release
// frees the chunks.  Outside a scope (sugar_conv1d and other one-off packers) the three calls are plain
// hipMalloc / hipMemcpy / hipFree.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace sugar {

namespace {
thread_local DevArena* tl_arena = nullptr;
constexpr size_t ARENA_CHUNK = 8u << 20;     // a small sampled subnet packs into one or two, the largest supernet into six

std::mutex g_stream_mu;
hipStream_t g_upload_stream[64] = {};

int upload_stream(hipStream_t* out) {
  int dev = 0;
  SUGAR_HIP(hipGetDevice(&dev));
  SUGAR_CHECK_ARG(dev >= 0 && dev < 64, "arena: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if (!g_upload_stream[dev]) SUGAR_HIP(hipStreamCreateWithFlags(&g_upload_stream[dev], hipStreamNonBlocking));
  *out = g_upload_stream[dev];
  return SUGAR_OK;
}

DevArena::Chunk* owner(DevArena* a, const void* p) {
  if (!a) return nullptr;
  for (auto& c : a->chunks)
    if ((const char*)p >= c.dev && (const char*)p < c.dev + c.cap) return &c;
  return nullptr;
}
}  // namespace

ArenaScope::ArenaScope(DevArena* a) : prev_(tl_arena) { tl_arena = a; }
ArenaScope::~ArenaScope() { tl_arena = prev_; }

int dev_alloc(void** p, size_t bytes) {
  *p = nullptr;
  if (!tl_arena) {
    SUGAR_HIP(hipMalloc(p, bytes));
    return SUGAR_OK;
  }
  const size_t need = (size_t)round_up((int64_t)(bytes ? bytes : 1), 256);
  DevArena::Chunk* c = tl_arena->chunks.empty() ? nullptr : &tl_arena->chunks.back();
  if (!c || c->used + need > c->cap) {
    DevArena::Chunk n;
    n.cap = need > ARENA_CHUNK ? need : ARENA_CHUNK;
    n.used = 0;
    n.host = (char*)malloc(n.cap);
    if (!n.host) { set_error("arena: host alloc of %zu bytes", n.cap); return SUGAR_ENOMEM; }
    const hipError_t e = hipMalloc((void**)&n.dev, n.cap);
    if (e != hipSuccess) { free(n.host); set_error("arena: hipMalloc: %s", hipGetErrorString(e)); return SUGAR_EHIP; }
    tl_arena->chunks.push_back(n);
    c = &tl_arena->chunks.back();
  }
  *p = c->dev + c->used;
  c->used += need;
  return SUGAR_OK;
}

int dev_upload(void* d_dst, const void* h_src, size_t bytes) {
  DevArena::Chunk* c = owner(tl_arena, d_dst);
  if (c && c->host) {                                  // not flushed yet: into the mirror, sent with the chunk
    memcpy(c->host + ((char*)d_dst - c->dev), h_src, bytes);
    return SUGAR_OK;
  }
  SUGAR_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return SUGAR_OK;
}

void dev_free(void* p) {
  if (!p || owner(tl_arena, p)) return;
  (void)hipFree(p);
}

int arena_flush(DevArena* a) {
  hipStream_t s = nullptr;
  if (int rc = upload_stream(&s)) return rc;
  for (auto& c : a->chunks)
    if (c.host && c.used) SUGAR_HIP(hipMemcpyAsync(c.dev, c.host, c.used, hipMemcpyHostToDevice, s));
  SUGAR_HIP(hipStreamSynchronize(s));                  // the weights are in place; no other stream was waited for
  for (auto& c : a->chunks) {
    free(c.host);
    c.host = nullptr;
  }
  return SUGAR_OK;
}

void arena_release(DevArena* a) {
  for (auto& c : a->chunks) {
    free(c.host);
    if (c.dev) (void)hipFree(c.dev);
  }
  a->chunks.clear();
}

}  // namespace sugar
