// nh_moves.hip -- host-side generator of the stretch-move random numbers.
//
// The device-resident step loop replays a captured graph per (half-)step; the only
// per-step host work left was drawing the move's random numbers with numpy
// (permutation + 3 N uniforms + logs: 16 us per half-step at 512 walkers, 140 us at
// 4096 = 8 GPUs x 512, i.e. more than the GPU needs for the half-step).  Here a
// worker thread fills a ring of page-locked blocks ahead of the consumer, so the
// Python thread only uploads a finished block and launches graphs.
//
// Stream definition (replicated on every rank: same seed -> same moves): xoshiro256**
// seeded by splitmix64(seed); per ensemble step, in this order,
//   1. a Fisher-Yates permutation of 0..N-1 (Lemire bounded draws): halves = red / blue
//   2. for half h = 0,1 and j < N/2:  z = ((a-1) u + 1)^2 / a           (u uniform [0,1))
//   3. for half h, j:                  partner = other_half[floor(u * N/2)]
//   4. for half h, j:                  lnU = log(u)
// Block layout = what nh_move_propose/accept read: per half-step slice of 3*ns doubles
//   { z[ns] | lnU[ns] | S[ns] int32 | partner[ns] int32 }.
#include <condition_variable>
#include <mutex>
#include <thread>

#include "nh_common.h"

struct nh_moves {
  int N, ns, ksteps, depth;
  double a;
  bool pinned;
  uint64_t s[4];
  std::vector<double*> blocks;   // depth blocks of ksteps*2 slices
  std::vector<int> ready;        // 0 = free, 1 = filled, 2 = consumed but not yet handed back
  int head = 0;                  // block the consumer reads
  int offset = 0;                // steps of blocks[head] already taken
  int tail = 0;                  // block the producer fills next
  bool stop = false;
  std::mutex mu;
  std::condition_variable cv;
  std::thread worker;
  std::vector<int> perm;
};

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t xo_next(uint64_t* s) {
  const uint64_t r = rotl(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return r;
}

static inline double xo_uniform(uint64_t* s) { return (double)(xo_next(s) >> 11) * 0x1.0p-53; }

static inline uint32_t xo_bounded(uint64_t* s, uint32_t n) {  // Lemire, unbiased
  uint64_t m = (uint64_t)(uint32_t)(xo_next(s) >> 32) * n;
  uint32_t l = (uint32_t)m;
  if (l < n) {
    uint32_t t = (0u - n) % n;
    while (l < t) {
      m = (uint64_t)(uint32_t)(xo_next(s) >> 32) * n;
      l = (uint32_t)m;
    }
  }
  return (uint32_t)(m >> 32);
}

static void fill_step(nh_moves* m, double* slice0) {
  const int N = m->N, ns = m->ns;
  int* p = m->perm.data();
  for (int i = 0; i < N; ++i) p[i] = i;
  for (int i = N - 1; i > 0; --i) {
    int j = (int)xo_bounded(m->s, (uint32_t)(i + 1));
    int t = p[i]; p[i] = p[j]; p[j] = t;
  }
  double* sl[2] = {slice0, slice0 + 3 * (size_t)ns};
  for (int h = 0; h < 2; ++h) {
    int* iv = reinterpret_cast<int*>(sl[h] + 2 * (size_t)ns);
    for (int j = 0; j < ns; ++j) iv[j] = p[h * ns + j];
  }
  for (int h = 0; h < 2; ++h)
    for (int j = 0; j < ns; ++j) {
      double u = xo_uniform(m->s);
      double t = (m->a - 1.0) * u + 1.0;
      sl[h][j] = t * t / m->a;
    }
  for (int h = 0; h < 2; ++h) {
    int* iv = reinterpret_cast<int*>(sl[h] + 2 * (size_t)ns);
    const int* other = p + (1 - h) * ns;
    for (int j = 0; j < ns; ++j) iv[ns + j] = other[(int)(xo_uniform(m->s) * ns)];
  }
  for (int h = 0; h < 2; ++h)
    for (int j = 0; j < ns; ++j) sl[h][ns + j] = std::log(xo_uniform(m->s));
}

static void producer(nh_moves* m) {
  for (;;) {
    int blk;
    {
      std::unique_lock<std::mutex> lk(m->mu);
      m->cv.wait(lk, [&] { return m->stop || !m->ready[m->tail]; });
      if (m->stop) return;
      blk = m->tail;
    }
    double* base = m->blocks[blk];
    for (int k = 0; k < m->ksteps; ++k) fill_step(m, base + (size_t)k * 2 * 3 * m->ns);
    {
      std::lock_guard<std::mutex> lk(m->mu);
      m->ready[blk] = 1;
      m->tail = (blk + 1) % m->depth;
    }
    m->cv.notify_all();
  }
}

extern "C" int nh_moves_create(unsigned long long seed, int N, double a, int ksteps, int depth,
                               int pinned, nh_moves** out) {
  NH_REQUIRE(out && N >= 2 && N % 2 == 0 && a > 1.0 && ksteps >= 1 && depth >= 3, "bad argument");
  nh_moves* m = new nh_moves();
  m->N = N; m->ns = N / 2; m->ksteps = ksteps; m->depth = depth; m->a = a; m->pinned = pinned != 0;
  uint64_t z = seed;  // splitmix64 seeding
  for (int i = 0; i < 4; ++i) {
    z += 0x9e3779b97f4a7c15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    m->s[i] = x ^ (x >> 31);
  }
  m->perm.resize(N);
  const size_t bytes = (size_t)ksteps * 2 * 3 * m->ns * sizeof(double);
  for (int b = 0; b < depth; ++b) {
    void* p = nullptr;
    if (m->pinned) {
      hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
      if (e != hipSuccess) {
        delete m;
        return nh_set_error(NH_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
      }
    } else {
      p = malloc(bytes);
      if (!p) { delete m; return nh_set_error(NH_ENOMEM, "malloc(%zu) failed", bytes); }
    }
    m->blocks.push_back(static_cast<double*>(p));
    m->ready.push_back(0);
  }
  m->worker = std::thread(producer, m);
  *out = m;
  return NH_OK;
}

// up to `want` consecutive steps of the stream, contiguous in (pinned) host memory.
// Lifetime of the pointer: a used-up block is handed back to the producer ONE BLOCK LATE --
// when the consumer moves on from block j to block j+1, block j-1 is released, not block j.
// A caller that ships the steps with an asynchronous copy (the device loop: hipMemcpyAsync
// from this pinned memory) may therefore have the copy of its MOST RECENT take still queued
// when it takes again; every earlier copy must have completed (device_sampler waits on the
// marker recorded after the last-but-one upload before each take).
extern "C" int nh_moves_take(nh_moves* m, int want, const void** ptr, int* got) {
  NH_REQUIRE(m && ptr && got && want >= 1, "bad argument");
  std::unique_lock<std::mutex> lk(m->mu);
  if (m->offset >= m->ksteps) {  // current block used up: move on, hand the PREVIOUS one back
    const int prev = (m->head + m->depth - 1) % m->depth;
    if (m->ready[prev] == 2) m->ready[prev] = 0;
    m->ready[m->head] = 2;
    m->head = (m->head + 1) % m->depth;
    m->offset = 0;
    m->cv.notify_all();
  }
  m->cv.wait(lk, [&] { return m->ready[m->head] == 1; });
  int n = want < m->ksteps - m->offset ? want : m->ksteps - m->offset;
  *ptr = m->blocks[m->head] + (size_t)m->offset * 2 * 3 * m->ns;
  *got = n;
  m->offset += n;
  return NH_OK;
}

extern "C" int nh_moves_destroy(nh_moves* m) {
  if (!m) return NH_OK;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->stop = true;
  }
  m->cv.notify_all();
  if (m->worker.joinable()) m->worker.join();
  for (double* p : m->blocks) {
    if (m->pinned) (void)hipHostFree(p); else free(p);
  }
  delete m;
  return NH_OK;
}
