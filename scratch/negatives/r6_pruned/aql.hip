// Raw AQL submission of a recorded forward (gfx950, ROCr user-mode queues).
//
// Why: one forward is ~50 short, mostly dependent kernels. Through a HIP stream or a hipGraph a
// dependent kernel node costs >= 4.2 us on MI355X and only 4 hardware queues run concurrently,
// so 40 % of a forward was dispatch bubbles. Measured with scratch/aql_probe (512-workgroup empty
// kernels, own queue): 1.56 us per dependent packet with agent-scope fences, 3.1 us with
// system-scope fences, and packets WITHOUT the barrier bit start while their predecessor runs.
//
// So the engine records its launch sequence once per batch shape (vog::launch + LaunchRecorder),
// turns it into pre-built hsa_kernel_dispatch_packet_t's with kernargs resident in HBM, and
// submits by copying packets into a user-mode queue:
//   * fences are agent scope between kernels of a forward; system scope acquire on the first row
//     and system scope release on the last packet (inputs / outputs cross queues there);
//   * a program is a list of ROWS; only the first packet of a row carries the barrier bit, the rest
//     of the row (the independent language / vision chains of one forward, and the same row of
//     other forwards interleaved by aql_submit) launch behind it without waiting;
//   * completion is one hsa_signal per program on its last packet.
// Kernel objects come from the device code objects (libvog_hip.<tu>.co next to the .so, same
// sources and flags as the fat binary) loaded through the HSA loader; host stubs are mapped to
// them by their mangled name (hipKernelNameRefByPtr).
#include <cxxabi.h>
#include <dirent.h>
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <stdlib.h>
#include <chrono>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "aql.h"

namespace vog {

thread_local LaunchRecorder* g_recorder = nullptr;

#define VOG_HSA(expr)                                                                         \
  do {                                                                                        \
    hsa_status_t s_ = (expr);                                                                 \
    if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) {                            \
      const char* m_ = "";                                                                    \
      hsa_status_string(s_, &m_);                                                             \
      VOG_FAIL(-2000 - (int)(s_ & 0xfff), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, m_);  \
    }                                                                                         \
  } while (0)

struct KernelInfo {
  uint64_t object = 0;
  uint32_t karg = 0, lds = 0, priv = 0;
};

struct AqlRuntime {
  bool ready = false;
  int device = -1;
  hsa_agent_t agent{};
  std::vector<std::string> blobs;                 // code objects stay alive with the executables
  std::vector<hsa_executable_t> execs;
  std::map<std::string, KernelInfo> kernels;      // mangled name (no ".kd")
  std::vector<hsa_queue_t*> queues;
  std::vector<std::mutex*> qlocks;
  uint64_t ticks_per_us = 100;
  volatile int queue_error = 0;
};

static std::mutex g_rt_lock;
static AqlRuntime g_rt;

struct AqlProgram {
  std::vector<std::vector<hsa_kernel_dispatch_packet_t>> rows;
  void* kargs = nullptr;
  hsa_signal_t done{};
  int n_packets = 0;
  bool in_flight = false;
  bool persistent = false;     // contains a persistent BiLSTM layer (co-residency: <= 4 such programs in flight)
};

// ---- runtime bring-up ------------------------------------------------------------------------------
struct AgentPick { uint32_t want_bdf; int want_ordinal; int seen; hsa_agent_t by_bdf, by_ord; bool have_bdf, have_ord; };

static hsa_status_t pick_agent_cb(hsa_agent_t a, void* data) {
  AgentPick* p = (AgentPick*)data;
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU)
    return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0;
  if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
      (bdf & 0xffff) == (p->want_bdf & 0xffff) && !p->have_bdf) {
    p->by_bdf = a; p->have_bdf = true;
  }
  if (p->seen == p->want_ordinal) { p->by_ord = a; p->have_ord = true; }
  p->seen++;
  return HSA_STATUS_SUCCESS;
}

struct SymCtx { AqlRuntime* rt; };
static hsa_status_t symbol_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t sym, void* data) {
  AqlRuntime* rt = ((SymCtx*)data)->rt;
  hsa_symbol_kind_t kind;
  if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS ||
      kind != HSA_SYMBOL_KIND_KERNEL)
    return HSA_STATUS_SUCCESS;
  uint32_t len = 0;
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
  std::string name(len, '\0');
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]);
  while (!name.empty() && name.back() == '\0') name.pop_back();
  if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
  KernelInfo k;
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object);
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg);
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds);
  hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv);
  rt->kernels[name] = k;
  return HSA_STATUS_SUCCESS;
}

static void queue_error_cb(hsa_status_t status, hsa_queue_t*, void* data) {
  AqlRuntime* rt = (AqlRuntime*)data;
  rt->queue_error = (int)status ? (int)status : -1;
  const char* m = "";
  hsa_status_string(status, &m);
  fprintf(stderr, "libvog_hip: AQL queue error: %s\n", m);
}

static std::string lib_dir() {
  Dl_info info;
  if (dladdr((const void*)&lib_dir, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
  }
  return ".";
}

static int runtime_init(AqlRuntime& rt) {
  int dev = 0;
  VOG_HIP(hipGetDevice(&dev));
  if (rt.ready) {
    // one process drives one GPU (torchrun: one rank per device); the queues belong to that device
    if (dev != rt.device) VOG_FAIL(-2001, "AQL path: opened on HIP device %d, current device is %d", rt.device, dev);
    return 0;
  }
  char bus[64] = "";
  VOG_HIP(hipDeviceGetPCIBusId(bus, sizeof(bus), dev));
  unsigned dom = 0, b = 0, d = 0, f = 0;
  AgentPick pick{};
  pick.want_ordinal = dev;
  if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4) pick.want_bdf = (b << 8) | (d << 3) | f;
  else pick.want_bdf = 0xffffffffu;
  VOG_HSA(hsa_init());
  VOG_HSA(hsa_iterate_agents(pick_agent_cb, &pick));
  if (pick.have_bdf) rt.agent = pick.by_bdf;
  else if (pick.have_ord) rt.agent = pick.by_ord;
  else VOG_FAIL(-2001, "AQL path: no HSA GPU agent for HIP device %d (%s)", dev, bus);
  rt.device = dev;
  uint64_t freq = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) == HSA_STATUS_SUCCESS && freq >= 1000000)
    rt.ticks_per_us = freq / 1000000;

  // device code objects: libvog_hip.<tu>.co beside the shared library
  const std::string dir = lib_dir();
  std::vector<std::string> files;
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      const std::string n(e->d_name);
      if (n.rfind("libvog_hip.", 0) == 0 && n.size() > 3 && n.compare(n.size() - 3, 3, ".co") == 0)
        files.push_back(dir + "/" + n);
    }
    closedir(dp);
  }
  if (files.empty()) VOG_FAIL(-2002, "AQL path: no libvog_hip.*.co code objects in %s (run csrc/build.py)", dir.c_str());
  rt.blobs.reserve(files.size());
  for (auto& fn : files) {
    std::ifstream in(fn, std::ios::binary);
    rt.blobs.emplace_back((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::string& blob = rt.blobs.back();
    if (blob.empty()) VOG_FAIL(-2002, "AQL path: cannot read %s", fn.c_str());
    hsa_code_object_reader_t rd;
    VOG_HSA(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    hsa_executable_t ex;
    VOG_HSA(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    VOG_HSA(hsa_executable_load_agent_code_object(ex, rt.agent, rd, nullptr, nullptr));
    VOG_HSA(hsa_executable_freeze(ex, nullptr));
    rt.execs.push_back(ex);
    SymCtx sc{&rt};
    VOG_HSA(hsa_executable_iterate_agent_symbols(ex, rt.agent, symbol_cb, &sc));
  }
  if (rt.kernels.empty()) VOG_FAIL(-2002, "AQL path: code objects hold no kernels");
  rt.ready = true;
  return 0;
}

int aql_open(int n_queues) {
  std::lock_guard<std::mutex> lk(g_rt_lock);
  VOG_CHECK_ARG(n_queues >= 1 && n_queues <= 16);
  VOG_TRY(runtime_init(g_rt));
  while ((int)g_rt.queues.size() < n_queues) {
    hsa_queue_t* q = nullptr;
    VOG_HSA(hsa_queue_create(g_rt.agent, 16384, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, &g_rt, UINT32_MAX, UINT32_MAX, &q));
    g_rt.queues.push_back(q);
    g_rt.qlocks.push_back(new std::mutex());
  }
  return 0;
}

int aql_num_queues() { return (int)g_rt.queues.size(); }

// ---- program construction ------------------------------------------------------------------------------
static const KernelInfo* find_kernel(AqlRuntime& rt, const void* host_fn, std::string& name_out) {
  const char* nm = hipKernelNameRefByPtr(host_fn, nullptr);
  if (!nm) return nullptr;
  std::string name(nm);
  name_out = name;
  if (name.size() > 3 && name.compare(name.size() - 3, 3, ".kd") == 0) name.resize(name.size() - 3);
  auto it = rt.kernels.find(name);
  if (it != rt.kernels.end()) return &it->second;
  // the runtime handed back a demangled name: compare with the demangled symbols
  for (auto& kv : rt.kernels) {
    int st = 0;
    char* dm = abi::__cxa_demangle(kv.first.c_str(), nullptr, nullptr, &st);
    if (st == 0 && dm) {
      std::string d(dm);
      free(dm);
      if (d == name || d.rfind(name + "(", 0) == 0 || d.rfind("void " + name, 0) == 0) return &kv.second;
    }
  }
  return nullptr;
}

int aql_program_build(const std::vector<std::vector<LaunchRecord>>& rows, AqlProgram** out) {
  VOG_CHECK_ARG(out);
  std::lock_guard<std::mutex> lk(g_rt_lock);
  if (!g_rt.ready) VOG_FAIL(-2003, "AQL path: call vog_aql_open first");
  AqlRuntime& rt = g_rt;
  AqlProgram* p = new AqlProgram();
  std::vector<unsigned char> host;
  std::vector<size_t> offs;
  int rc = 0;
  for (auto& row : rows) {
    std::vector<hsa_kernel_dispatch_packet_t> prow;
    for (const LaunchRecord& r : row) {
      std::string nm;
      const KernelInfo* k = find_kernel(rt, r.host_fn, nm);
      if (nm.find("lstm_layer_kernel") != std::string::npos || nm.find("LstmLayerBody") != std::string::npos)
        p->persistent = true;
      if (!k) { set_error("AQL path: kernel '%s' not found in the device code objects", nm.c_str()); rc = -2004; break; }
      if (r.arg_bytes == 0xffffffffu || r.arg_bytes > k->karg) {
        set_error("AQL path: kernel '%s': %u argument bytes recorded, kernarg segment is %u", nm.c_str(), r.arg_bytes, k->karg);
        rc = -2004; break;
      }
      for (int i = 0; i < 3; ++i)
        if (r.block[i] == 0 || r.grid[i] == 0 || (uint64_t)r.grid[i] * r.block[i] > 0xffffffffull) {
          set_error("AQL path: kernel '%s': bad launch geometry", nm.c_str()); rc = -2004; break;
        }
      if (rc) break;
      const size_t seg = ((size_t)(k->karg < 64 ? 64 : k->karg) + 63) / 64 * 64;
      const size_t off = host.size();
      host.resize(off + seg, 0);
      unsigned char* ka = host.data() + off;
      memcpy(ka, r.args, r.arg_bytes);
      // code-object-v5 implicit arguments behind the explicit ones (zero elsewhere: no printf /
      // hostcall / heap / multigrid in these kernels)
      const size_t h = ((size_t)r.arg_bytes + 7) & ~(size_t)7;
      if (k->karg >= h + 24) {
        uint32_t* bc = reinterpret_cast<uint32_t*>(ka + h);
        bc[0] = r.grid[0]; bc[1] = r.grid[1]; bc[2] = r.grid[2];
        uint16_t* gs = reinterpret_cast<uint16_t*>(ka + h + 12);
        gs[0] = (uint16_t)r.block[0]; gs[1] = (uint16_t)r.block[1]; gs[2] = (uint16_t)r.block[2];
        gs[3] = gs[4] = gs[5] = 0;                                  // remainders: grids are whole workgroups
        if (k->karg >= h + 66) *reinterpret_cast<uint16_t*>(ka + h + 64) = 3;   // hidden_grid_dims
        if (k->karg >= h + 124) *reinterpret_cast<uint32_t*>(ka + h + 120) = r.dyn_lds;   // hidden_dynamic_lds_size
      }
      hsa_kernel_dispatch_packet_t d{};
      d.setup = 3;
      d.workgroup_size_x = (uint16_t)r.block[0]; d.workgroup_size_y = (uint16_t)r.block[1]; d.workgroup_size_z = (uint16_t)r.block[2];
      d.grid_size_x = r.grid[0] * r.block[0]; d.grid_size_y = r.grid[1] * r.block[1]; d.grid_size_z = r.grid[2] * r.block[2];
      d.private_segment_size = k->priv;
      d.group_segment_size = k->lds + r.dyn_lds;
      d.kernel_object = k->object;
      d.kernarg_address = nullptr;      // patched below
      offs.push_back(off);
      prow.push_back(d);
      p->n_packets++;
    }
    if (rc) break;
    if (!prow.empty()) p->rows.push_back(prow);
  }
  if (rc == 0 && p->n_packets == 0) { set_error("AQL path: empty program"); rc = -2004; }
  if (rc == 0) {
    hipError_t e = hipMalloc(&p->kargs, host.size());
    if (e == hipSuccess) e = hipMemcpy(p->kargs, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("AQL path: kernarg upload: %s", hipGetErrorString(e)); rc = -(int)e - 1000; }
  }
  if (rc == 0) {
    size_t i = 0;
    for (auto& row : p->rows)
      for (auto& d : row) d.kernarg_address = (char*)p->kargs + offs[i++];
    hsa_status_t s = hsa_signal_create(0, 0, nullptr, &p->done);
    if (s != HSA_STATUS_SUCCESS) { set_error("AQL path: hsa_signal_create failed"); rc = -2005; }
  }
  if (rc != 0) {
    if (p->kargs) (void)hipFree(p->kargs);
    delete p;
    return rc;
  }
  *out = p;
  return 0;
}

int aql_program_packets(const AqlProgram* p) { return p ? p->n_packets : 0; }
int aql_program_rows(const AqlProgram* p) { return p ? (int)p->rows.size() : 0; }

// ---- submission ----------------------------------------------------------------------------------------------
static inline uint64_t write_packet(hsa_queue_t* q, const hsa_kernel_dispatch_packet_t& src, bool barrier, int acq,
                                    int rel, hsa_signal_t completion) {
  const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
  while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}     // ring full: wait for the CP
  hsa_kernel_dispatch_packet_t* d = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
  // body first (the slot's header is INVALID), then the header with release semantics
  d->setup = src.setup;
  d->workgroup_size_x = src.workgroup_size_x; d->workgroup_size_y = src.workgroup_size_y; d->workgroup_size_z = src.workgroup_size_z;
  d->reserved0 = 0;
  d->grid_size_x = src.grid_size_x; d->grid_size_y = src.grid_size_y; d->grid_size_z = src.grid_size_z;
  d->private_segment_size = src.private_segment_size; d->group_segment_size = src.group_segment_size;
  d->kernel_object = src.kernel_object; d->kernarg_address = src.kernarg_address;
  d->reserved2 = 0;
  d->completion_signal = completion;
  const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                                     ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                     (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                     (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
  __atomic_store_n(reinterpret_cast<uint16_t*>(d), header, __ATOMIC_RELEASE);
  return idx;
}

int aql_submit(AqlProgram* const* progs, int n, int queue) {
  VOG_CHECK_ARG(progs && n >= 1 && n <= 64);
  AqlRuntime& rt = g_rt;
  if (!rt.ready || queue < 0 || queue >= (int)rt.queues.size())
    VOG_FAIL(-2003, "AQL path: queue %d not open (%d queues)", queue, (int)rt.queues.size());
  if (rt.queue_error) VOG_FAIL(-2006, "AQL path: queue error %d was reported earlier", rt.queue_error);
  size_t max_rows = 0;
  for (int i = 0; i < n; ++i) {
    VOG_CHECK_ARG(progs[i] != nullptr);
    if (progs[i]->in_flight) VOG_FAIL(-2007, "AQL path: program %d is still in flight (wait for it first)", i);
    for (int j = 0; j < i; ++j) VOG_CHECK_ARG(progs[j] != progs[i]);
    max_rows = progs[i]->rows.size() > max_rows ? progs[i]->rows.size() : max_rows;
  }
  {
    // The persistent BiLSTM layer needs its 64 workgroups (one CU each) resident together: at most 4
    // instances may EXECUTE at once on a 256-CU part. Programs of one submission are row-interleaved
    // (concurrent); submissions on one queue run one after the other; queues run concurrently. So a
    // submission may hold at most 4 / (open queues) such programs - enforced here, not left to the
    // caller (a 5th instance would spin until its timeout and poison its output with NaN).
    int add = 0;
    for (int i = 0; i < n; ++i) add += progs[i]->persistent ? 1 : 0;
    const int nq = (int)rt.queues.size();
    if (add > 0 && (nq > 4 || add > 4 / nq))
      VOG_FAIL(-2009, "AQL path: %d programs with a persistent BiLSTM in one submission with %d queues open exceed "
               "the co-residency limit of 4 concurrent instances (interleave fewer, or set lstm_persistent = 0)", add, nq);
  }
  std::lock_guard<std::mutex> lk(*rt.qlocks[queue]);
  hsa_queue_t* q = rt.queues[queue];
  for (int i = 0; i < n; ++i) { hsa_signal_store_relaxed(progs[i]->done, 1); progs[i]->in_flight = true; }
  const hsa_signal_t none{0};
  // VOG_AQL_FENCE (perf experiments only, results may be WRONG): fence scope between the kernels
  // of a forward: 0 = none, 1 = agent (default), 2 = system; +10: acquire only, +20: release only
  static const int fence_env = perf_env("VOG_AQL_FENCE") ? atoi(perf_env("VOG_AQL_FENCE")) : 1;
  const int mid_acq = (fence_env / 10 == 2) ? HSA_FENCE_SCOPE_NONE : (fence_env % 10);
  const int mid_rel = (fence_env / 10 == 1) ? HSA_FENCE_SCOPE_NONE : (fence_env % 10);
  for (size_t r = 0; r < max_rows; ++r) {
    bool first = true;
    uint64_t last_idx = 0;
    for (int i = 0; i < n; ++i) {
      AqlProgram* p = progs[i];
      if (r >= p->rows.size()) continue;
      const bool first_row = r == 0, last_row = r + 1 == p->rows.size();
      const auto& row = p->rows[r];
      for (size_t j = 0; j < row.size(); ++j) {
        const bool last_pkt = last_row && j + 1 == row.size();
        // a last row with several packets would need one signal per packet; programs end in a
        // single-kernel row by construction (checked in vog_aql_program_create)
        last_idx = write_packet(q, row[j], first, first_row ? HSA_FENCE_SCOPE_SYSTEM : mid_acq,
                                last_row ? HSA_FENCE_SCOPE_SYSTEM : mid_rel, last_pkt ? p->done : none);
        first = false;
      }
    }
    if (!first) hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)last_idx);   // one doorbell per row
  }
  return 0;
}

int aql_wait(AqlProgram* p, uint64_t timeout_us) {
  VOG_CHECK_ARG(p);
  if (!p->in_flight) return 0;
  AqlRuntime& rt = g_rt;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(p->done, HSA_SIGNAL_CONDITION_LT, 1, 200 * rt.ticks_per_us,
                                                           HSA_WAIT_STATE_ACTIVE);
    if (v < 1) break;
    if (rt.queue_error) VOG_FAIL(-2006, "AQL path: queue error %d while waiting", rt.queue_error);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us > (double)timeout_us) VOG_FAIL(-2008, "AQL path: program did not complete within %llu us", (unsigned long long)timeout_us);
  }
  p->in_flight = false;
  return 0;
}

int aql_program_destroy(AqlProgram* p) {
  if (!p) return 0;
  if (p->in_flight) (void)aql_wait(p, 2000000);
  if (p->kargs) (void)hipFree(p->kargs);
  if (p->done.handle) (void)hsa_signal_destroy(p->done);
  delete p;
  return 0;
}

}  // namespace vog
