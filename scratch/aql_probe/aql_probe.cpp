// Standalone HSA/AQL probe: how much of the dependent-kernel bubble can be shared by
// independent kernels queued WITHOUT the barrier bit behind one barrier packet?
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <algorithm>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); \
  fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m); exit(1); } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool; static bool g_have_dev = false, g_have_karg = false;
static hsa_agent_t g_cpu; static bool g_have_cpu = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void* data) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (!alloc) return HSA_STATUS_SUCCESS;
  if (data == (void*)1) {   // gpu pools
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) { g_dev_pool = p; g_have_dev = true; }
  } else {                  // cpu pools
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_karg) { g_karg_pool = p; g_have_karg = true; }
  }
  return HSA_STATUS_SUCCESS;
}

struct Kern { uint64_t object; uint32_t karg, lds, priv; };
static Kern get_kernel(hsa_executable_t ex, const char* name) {
  hsa_executable_symbol_t sym; CK(hsa_executable_get_symbol_by_name(ex, name, &g_gpu, &sym));
  Kern k{};
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  return k;
}

struct Pkt { Kern k; uint32_t grid, block; void* karg; bool barrier; };

static hsa_queue_t* g_q; static hsa_signal_t g_done;
static hsa_queue_t* g_qs[8]; static hsa_signal_t g_dones[8];
static bool g_host_karg = false;
static int g_acq = HSA_FENCE_SCOPE_AGENT, g_rel = HSA_FENCE_SCOPE_AGENT;

static void submit(const std::vector<Pkt>& ps, hsa_queue_t* g_q, hsa_signal_t g_done) {
  const uint32_t mask = g_q->size - 1;
  for (size_t i = 0; i < ps.size(); ++i) {
    const Pkt& p = ps[i];
    uint64_t idx = hsa_queue_add_write_index_relaxed(g_q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(g_q) >= g_q->size) {}
    hsa_kernel_dispatch_packet_t* d = (hsa_kernel_dispatch_packet_t*)g_q->base_address + (idx & mask);
    d->setup = 1;   // 1 dimension
    d->workgroup_size_x = p.block; d->workgroup_size_y = 1; d->workgroup_size_z = 1;
    d->grid_size_x = p.grid; d->grid_size_y = 1; d->grid_size_z = 1;
    d->private_segment_size = p.k.priv; d->group_segment_size = p.k.lds;
    d->kernel_object = p.k.object; d->kernarg_address = p.karg; d->reserved2 = 0;
    d->completion_signal = (i + 1 == ps.size() && g_done.handle) ? g_done : hsa_signal_t{0};
    uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                      ((p.barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (g_acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                      (g_rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n((uint16_t*)d, header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(g_q->doorbell_signal, idx);
  }
}
static double run(const std::vector<Pkt>& ps, int reps = 5) {
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    hsa_signal_store_relaxed(g_done, 1);
    auto t0 = std::chrono::steady_clock::now();
    submit(ps, g_q, g_done);
    while (hsa_signal_wait_scacquire(g_done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) {
      fprintf(stderr, "timeout waiting for queue\n"); exit(2);
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us < best) best = us;
  }
  return best;
}

// the same packet list on nq queues at once; returns us until all are done
static double run_multi(const std::vector<Pkt>& ps, int nq, int reps = 5) {
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    for (int q = 0; q < nq; ++q) hsa_signal_store_relaxed(g_dones[q], 1);
    auto t0 = std::chrono::steady_clock::now();
    // interleave submission chunk-wise so all queues start together
    const size_t chunk = 16;
    for (size_t off = 0; off < ps.size(); off += chunk)
      for (int q = 0; q < nq; ++q) {
        std::vector<Pkt> part(ps.begin() + off, ps.begin() + std::min(ps.size(), off + chunk));
        const bool last = off + chunk >= ps.size();
        submit(part, g_qs[q], last ? g_dones[q] : hsa_signal_t{0});
      }
    for (int q = 0; q < nq; ++q)
      while (hsa_signal_wait_scacquire(g_dones[q], HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) { fprintf(stderr, "timeout\n"); exit(2); }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us < best) best = us;
  }
  return best;
}

// fills the code-object-v5 hidden block (block counts / group sizes) behind `explicit_bytes` of arguments
static void* make_karg(const Kern& k, const void* args, size_t explicit_bytes, uint32_t grid, uint32_t block) {
  size_t n = k.karg < 64 ? 64 : k.karg;
  std::vector<char> hostbuf(n, 0);
  char* p = hostbuf.data();
  memcpy(p, args, explicit_bytes);
  size_t h = (explicit_bytes + 7) & ~size_t(7);
  if (k.karg >= h + 24) {
    uint32_t* bc = (uint32_t*)((char*)p + h); bc[0] = grid / block; bc[1] = 1; bc[2] = 1;
    uint16_t* gs = (uint16_t*)((char*)p + h + 12); gs[0] = block; gs[1] = 1; gs[2] = 1; gs[3] = 0; gs[4] = 0; gs[5] = 0;
    if (k.karg >= h + 66) *(uint16_t*)((char*)p + h + 64) = 1;
  }
  void* d = nullptr;
  if (g_host_karg) {
    CK(hsa_amd_memory_pool_allocate(g_karg_pool, n, 0, &d));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, d));
    memcpy(d, p, n);
  } else {      // kernarg segment in device memory (what HIP does on gfx94x/gfx950)
    CK(hsa_amd_memory_pool_allocate(g_dev_pool, n, 0, &d));
    CK(hsa_memory_copy(d, p, n));
  }
  return d;
}

int main(int argc, char** argv) {
  const char* co = argc > 1 ? argv[1] : "probe_kernels.co";
  g_host_karg = argc > 2 && atoi(argv[2]) == 1;
  CK(hsa_init());
  CK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have) { fprintf(stderr, "no gpu agent\n"); return 1; }
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, pool_cb, (void*)1));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, pool_cb, (void*)0));
  if (!g_have_dev || !g_have_karg) { fprintf(stderr, "pools missing\n"); return 1; }
  std::ifstream f(co, std::ios::binary); std::string blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
  hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
  CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(ex, nullptr));
  Kern k_empty = get_kernel(ex, "empty_kernel.kd"), k_stream = get_kernel(ex, "stream_kernel.kd"), k_inc = get_kernel(ex, "inc_kernel.kd");
  printf("kernarg sizes: empty %u stream %u inc %u; lds %u %u\n", k_empty.karg, k_stream.karg, k_inc.karg, k_empty.lds, k_stream.lds);
  CK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));
  CK(hsa_signal_create(1, 0, nullptr, &g_done));

  for (int q = 0; q < 8; ++q) {
    CK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_qs[q]));
    CK(hsa_signal_create(1, 0, nullptr, &g_dones[q]));
  }
  const size_t NB = 1024, PER = 4096;   // up to 1024 blocks x 4096 float4 = 64 MiB
  float *in = nullptr, *out[8];
  CK(hsa_amd_memory_pool_allocate(g_dev_pool, NB * PER * 16, 0, (void**)&in));
  for (int i = 0; i < 8; ++i) CK(hsa_amd_memory_pool_allocate(g_dev_pool, NB * PER * 16, 0, (void**)&out[i]));

  // ---- correctness: chain of 64 inc kernels with barrier bits, then rows of 4 independent chains
  {
    const int n = 1 << 16; int* buf[5];
    for (int i = 0; i < 5; ++i) CK(hsa_amd_memory_pool_allocate(g_dev_pool, n * 4 * 2, 0, (void**)&buf[i]));
    std::vector<int> h(n, 0), r(n);
    for (int c = 0; c < 4; ++c) { for (int i = 0; i < n; ++i) h[i] = c * 1000; CK(hsa_memory_copy(buf[c], h.data(), n * 4)); }
    std::vector<Pkt> ps;
    const int rows = 64;
    for (int row = 0; row < rows; ++row)
      for (int c = 0; c < 4; ++c) {
        struct { const int* in; int* out; int n; } a{buf[c] + (row % 2) * n, buf[c] + ((row + 1) % 2) * n, n};
        ps.push_back({k_inc, (uint32_t)n, 256, make_karg(k_inc, &a, sizeof(a), n, 256), c == 0});
      }
    run(ps, 1);
    bool ok = true;
    for (int c = 0; c < 4; ++c) {
      CK(hsa_memory_copy(r.data(), buf[c] + (rows % 2) * n, n * 4));
      for (int i = 0; i < n; ++i) ok &= r[i] == c * 1000 + rows;
    }
    printf("interleaved chains correctness: %s\n", ok ? "OK" : "WRONG");
  }

  for (int scope = 0; scope < 2; ++scope) {
    g_acq = g_rel = scope == 0 ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_SYSTEM;
    printf("== fence scope %s\n", scope == 0 ? "agent" : "system");
    int dummy_arg_store[2] = {0, 0}; void* nullp = nullptr;
    void* ka_empty = make_karg(k_empty, &nullp, 8, 256, 256);
    const int N = 400;
    for (int k : {1, 2, 3, 4, 8}) {
      std::vector<Pkt> ps;
      for (int row = 0; row < N; ++row) for (int c = 0; c < k; ++c) ps.push_back({k_empty, 256 * 512, 256, ka_empty, c == 0});
      double us = run(ps);
      printf("empty 512 WGs: rows of %d (1 barrier + %d free): %.2f us/row  %.2f us/kernel\n", k, k - 1, us / N, us / N / k);
    }
    {
      std::vector<Pkt> ps;
      for (int row = 0; row < N; ++row) ps.push_back({k_empty, 256 * 512, 256, ka_empty, false});
      double us = run(ps);
      printf("empty 512 WGs: no barrier bits at all: %.2f us/kernel\n", us / N);
    }
    {
      Kern k_touch = get_kernel(ex, "touch_kernel.kd"), k_chase = get_kernel(ex, "chase_kernel.kd");
      int* idxbuf = nullptr; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 512 * 256 * 4, 0, (void**)&idxbuf));
      { std::vector<int> h(512 * 256); for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 97) % h.size()); CK(hsa_memory_copy(idxbuf, h.data(), h.size() * 4)); }
      for (int wgs : {4, 512}) for (int which = 0; which < 2; ++which) for (int k : {1, 4}) {
        std::vector<Pkt> ps; std::vector<void*> ka(k);
        for (int c = 0; c < k; ++c) {
          if (which == 0) { struct { const float* in; float* out; } a{in, out[c]}; ka[c] = make_karg(k_touch, &a, sizeof(a), 256 * wgs, 256); }
          else { struct { const int* idx; const float* in; float* out; } a{idxbuf, in, out[c]}; ka[c] = make_karg(k_chase, &a, sizeof(a), 256 * wgs, 256); }
        }
        for (int row = 0; row < N; ++row) for (int c = 0; c < k; ++c) ps.push_back({which == 0 ? k_touch : k_chase, (uint32_t)(256 * wgs), 256, ka[c], c == 0});
        double us = run(ps);
        printf("%s %3d WGs: rows of %d: %.2f us/row  %.2f us/kernel\n", which == 0 ? "touch" : "chase", wgs, k, us / N, us / N / k);
      }
    }
    {   // CP packet rate across queues: N packets per queue, all queues concurrently
      Kern k_touch = get_kernel(ex, "touch_kernel.kd");
      struct { const float* in; float* out; } a{in, out[0]};
      void* ka = make_karg(k_touch, &a, sizeof(a), 256 * 512, 256);
      void* nullp2 = nullptr; void* ka_e = make_karg(k_empty, &nullp2, 8, 256, 256);
      for (int kind = 0; kind < 3; ++kind) for (int barrier = 0; barrier < 2; ++barrier) for (int nq : {1, 2, 4, 8}) {
        std::vector<Pkt> ps;
        for (int i = 0; i < 512; ++i) {
          if (kind == 0) ps.push_back({k_empty, 256, 256, ka_e, barrier == 1});
          else ps.push_back({k_touch, (uint32_t)(256 * (kind == 1 ? 4 : 512)), 256, ka, barrier == 1});
        }
        double us = run_multi(ps, nq);
        printf("multi-queue %s barrier=%d queues=%d: %.2f us per packet per queue, %.2f us per packet overall\n",
               kind == 0 ? "empty(1 WG)" : kind == 1 ? "touch(4 WG)" : "touch(512 WG)", barrier, nq, us / 512, us / 512 / nq);
      }
    }
    for (int per : {256}) {        // bytes per launch = 512 * per * 16 * 2 (read+write)
      for (int k : {1, 2, 4}) {
        std::vector<Pkt> ps;
        std::vector<void*> ka(k);
        for (int c = 0; c < k; ++c) {
          struct { const float* in; float* out; int per; } a{in, out[c], per};
          ka[c] = make_karg(k_stream, &a, sizeof(a), 256 * 512, 256);
        }
        for (int row = 0; row < N; ++row) for (int c = 0; c < k; ++c) ps.push_back({k_stream, 256 * 512, 256, ka[c], c == 0});
        double us = run(ps);
        printf("stream %5.1f MB/launch: rows of %d: %.2f us/row  %.2f us/kernel\n", 512.0 * per * 32 / 1e6, k, us / N, us / N / k);
      }
    }
    (void)dummy_arg_store;
  }
  hsa_queue_destroy(g_q);
  hsa_shut_down();
  return 0;
}
