// Microbenchmark: what does a DEPENDENT dispatch cost when the ordering is carried by the data instead of by the queue?
// A chain of L 160 x 256 x 256 layers (tools/sentinel_chain.hip's toy layer) written straight into an HSA queue as AQL packets:
//   (a) barrier bit on every packet, agent-scope acquire / release fences      = what a HIP stream / graph node is
//   (b) barrier bit, no fences (polling kernel: system-scope loads, written-through stores)
//   (c) NO barrier bit: packet i + 1 is dispatched while packet i runs; its workgroups fetch their weights and poll their A operand
//       (pre-filled with a sentinel) until the producer's words arrive -- the in-launch slot chain's hand-off
//       between SEPARATE specialised kernels, with no placement assumption (hand-offs through memory, not through one XCD's L2)
//   (d) the same with a barrier bit every B-th packet (bounds how many layers poll at once)
// HIP cannot express (c): hipExtAnyOrderLaunch is not honoured on gfx9, and independent graph nodes are spread over hardware queues
// (a second active queue costs every node ~1 us, DESIGN_history.md).  This tool prices what a native AQL replayer could buy.
//   g++ -O2 -std=c++17 -I/opt/rocm/include tools/aql_chain.cpp -L/opt/rocm/lib -lhsa-runtime64 -o tools/bin/aql_chain
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only --no-gpu-bundle-output -o tools/bin/aql_chain.hsaco tools/aql_chain_kernels.hip
//   timeout 120 tools/bin/aql_chain [layers=512] [reps=20]
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hsa_status_t s_ = (x);                                                             \
    if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) {                     \
      const char* m_ = nullptr;                                                        \
      hsa_status_string(s_, &m_);                                                      \
      printf("%s: %s\n", #x, m_ ? m_ : "?");                                           \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_amd_memory_pool_t g_coarse, g_fine, g_kernarg;
static bool g_has_coarse = false, g_has_fine = false, g_has_kernarg = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t gpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (!alloc) return HSA_STATUS_SUCCESS;
  if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_has_coarse) { g_coarse = p; g_has_coarse = true; }
  if ((fl & (HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED | HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED)) && !g_has_fine) { g_fine = p; g_has_fine = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_has_kernarg) { g_kernarg = p; g_has_kernarg = true; }
  return HSA_STATUS_SUCCESS;
}
static void* pool_alloc(hsa_amd_memory_pool_t p, size_t bytes) {
  void* ptr = nullptr;
  CK(hsa_amd_memory_pool_allocate(p, bytes, 0, &ptr));
  hsa_agent_t ags[2] = {g_gpu, g_cpu};
  hsa_amd_agents_allow_access(2, ags, nullptr, ptr);   // (may fail for device-local pools towards the CPU on small-BAR systems: ignored)
  return ptr;
}

struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };
static Kernel find_kernel(hsa_executable_t exe, const char* name) {
  hsa_executable_symbol_t sym;
  CK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
  Kernel k;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  return k;
}

static hsa_queue_t* g_q;
static hsa_signal_t g_done;
static uint16_t header(bool barrier, int acq, int rel) {
  return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                    (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}
struct Pkt { const Kernel* k; void* args; uint32_t gx, gy; bool barrier; int acq, rel; };
// writes all packets, rings the doorbell once, waits for the last one; returns microseconds from doorbell to completion
static double submit(const std::vector<Pkt>& ps, int wg = 256) {
  const uint64_t n = ps.size();
  const uint64_t base = hsa_queue_add_write_index_relaxed(g_q, n);
  hsa_kernel_dispatch_packet_t* ring = (hsa_kernel_dispatch_packet_t*)g_q->base_address;
  const uint32_t mask = g_q->size - 1;
  hsa_signal_store_relaxed(g_done, 1);
  for (uint64_t i = 0; i < n; ++i) {
    hsa_kernel_dispatch_packet_t* p = ring + ((base + i) & mask);
    const Pkt& s = ps[i];
    p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = wg; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = s.gx * wg; p->grid_size_y = s.gy; p->grid_size_z = 1;
    p->private_segment_size = s.k->priv; p->group_segment_size = s.k->group;
    p->kernel_object = s.k->object; p->kernarg_address = s.args;
    p->reserved2 = 0;
    p->completion_signal.handle = i + 1 == n ? g_done.handle : 0;
  }
  // headers last (a packet becomes valid with its header); back to front so the processor never runs into an invalid successor early
  for (uint64_t i = n; i-- > 0;) {
    hsa_kernel_dispatch_packet_t* p = ring + ((base + i) & mask);
    const Pkt& s = ps[i];
    __atomic_store_n(&p->header, header(s.barrier, s.acq, s.rel), __ATOMIC_RELEASE);
  }
  const auto t0 = std::chrono::high_resolution_clock::now();
  hsa_signal_store_screlease(g_q->doorbell_signal, (hsa_signal_value_t)(base + n - 1));
  while (hsa_signal_wait_scacquire(g_done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
}

static double submit64(const std::vector<Pkt>& ps) { return submit(ps, 64); }

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
  constexpr int M = 160, KD = 256, ND = 256;
  CK(hsa_init());
  CK(hsa_iterate_agents(agent_cb, nullptr));
  if (!g_have_gpu || !g_have_cpu) { printf("no GPU / CPU agent\n"); return 2; }
  char name[64] = {0};
  hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, gpu_pool_cb, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
  printf("agent %s: coarse pool %d, fine pool %d, kernarg pool %d\n", name, g_has_coarse, g_has_fine, g_has_kernarg);
  if (!g_has_coarse || !g_has_kernarg) return 2;
  if (L + 2 > 4000) { printf("at most 3998 layers\n"); return 2; }
  CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));
  CK(hsa_signal_create(1, 0, nullptr, &g_done));
  // code object
  std::string path = argc > 3 ? argv[3] : std::string(argv[0]) + ".hsaco";
  std::ifstream f(path, std::ios::binary);
  if (!f) { printf("cannot read %s\n", path.c_str()); return 2; }
  std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  hsa_code_object_reader_t rd;
  CK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &rd));
  hsa_executable_t exe;
  hsa_profile_t prof = HSA_PROFILE_FULL;
  hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_PROFILE, &prof);
  CK(hsa_executable_create_alt(prof, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(exe, nullptr));
  const Kernel k_layer = find_kernel(exe, "k_layer.kd"), k_poll = find_kernel(exe, "k_layer_poll.kd"), k_fill = find_kernel(exe, "k_fill.kd"),
               k_xcd = find_kernel(exe, "k_layer_poll_xcd.kd"), k_delay = find_kernel(exe, "k_delay.kd");
  printf("kernarg bytes: k_layer %u, k_layer_poll %u, k_fill %u; LDS %u\n", k_layer.kernarg, k_poll.kernarg, k_fill.kernarg, k_layer.group);

  // buffers: L + 1 activation buffers (each hand-off in its own, so that a sentinel pre-fill per pass suffices), weights, status
  const size_t act_floats = (size_t)M * KD, act_bytes = act_floats * 4;
  std::vector<float> hx(act_floats), hw((size_t)KD * ND);
  srand(1);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
  float* w = (float*)pool_alloc(g_coarse, hw.size() * 4);
  CK(hsa_memory_copy(w, hw.data(), hw.size() * 4));
  int* status = (int*)pool_alloc(g_kernarg, 64);
  int* hits_dev = (int*)pool_alloc(g_coarse, 64);   // (device memory: an atomic per workgroup into host memory costs ~1 us each)

  {   // (f) independent 10 us kernels of 16 one-wave workgroups, with and without the barrier bit
    char* kd = (char*)pool_alloc(g_coarse, 512);
    const int ticks = 1000;
    CK(hsa_memory_copy(kd, &ticks, 4));
    for (int bar = 1; bar >= 0; --bar) {
      std::vector<Pkt> ps;
      for (int l = 0; l < 256; ++l) ps.push_back(Pkt{&k_delay, kd, 16, 1, bar != 0 || l == 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE});
      for (auto& p : ps) p.gx = 16;
      double best = 1e30;
      for (int r = 0; r < 5; ++r) { const double t = submit64(ps); if (t < best) best = t; }
      printf("(f) 256 independent 10 us kernels, barrier bit %d: %.1f us -> %.2f us per kernel\n", bar, best, best / 256);
    }
  }
  for (int fine = 0; fine < (g_has_fine ? 2 : 1); ++fine) {
    float* acts = (float*)pool_alloc(fine ? g_fine : g_coarse, act_bytes * (L + 1));
    CK(hsa_memory_copy(acts, hx.data(), act_bytes));
    // kernel arguments in DEVICE memory, staged on the host (a kernel that fetches its arguments from host memory pays the
    // bus latency on every start: 10 - 20 us per dependent layer here, against 2.6 us as a HIP graph node)
    const size_t ka_bytes = (size_t)(L + 2) * 512;
    char* ka_dev = (char*)pool_alloc(g_coarse, ka_bytes);
    std::vector<char> ka_host(ka_bytes, 0);
    char* ka = ka_host.data();
    struct ArgsL { const float* x; const float* w; float* y; };
    struct ArgsP { const float* x; const float* w; float* y; int* status; int spin_limit; int sleep; };
    struct ArgsF { unsigned* p; unsigned v; long n; long stride; };
    struct ArgsX { const float* x; const float* w; float* y; int* status; int spin_limit; int fast_polls; int m_tiles; int pad; int* hits; };
    auto run = [&](const char* label, bool poll, int barrier_every, int acq, int rel, int sleep, int xcd_fast = -1) {
      std::vector<Pkt> ps;
      // sentinel pre-fill of every hand-off buffer (its own packet, fully fenced)
      ArgsF* af = (ArgsF*)(ka + (size_t)(L + 1) * 512);
      af->p = (unsigned*)(acts + act_floats); af->v = 0xFFFFFFFFu; af->n = (long)act_floats * L; af->stride = 2048L * 256;
      ps.push_back(Pkt{&k_fill, ka_dev + ((char*)af - ka), 2048, 1, true, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM});
      for (int l = 0; l < L; ++l) {
        void* a = ka + (size_t)l * 512;
        if (xcd_fast >= 0) { ArgsX* p = (ArgsX*)a; p->x = acts + act_floats * l; p->w = w; p->y = acts + act_floats * (l + 1); p->status = status; p->spin_limit = 1 << 12; p->fast_polls = xcd_fast; p->m_tiles = M / 16; p->pad = 0; p->hits = hits_dev; }
        else if (poll) { ArgsP* p = (ArgsP*)a; p->x = acts + act_floats * l; p->w = w; p->y = acts + act_floats * (l + 1); p->status = status; p->spin_limit = 1 << 12; p->sleep = sleep; }
        else { ArgsL* p = (ArgsL*)a; p->x = acts + act_floats * l; p->w = w; p->y = acts + act_floats * (l + 1); }
        const bool bar = l == 0 || barrier_every == 1 || (barrier_every > 1 && l % barrier_every == 0);
        if (xcd_fast >= 0) ps.push_back(Pkt{&k_xcd, ka_dev + ((char*)a - ka), 8u * 16u * ((M / 16 + 7) / 8), 1, bar, l == 0 ? HSA_FENCE_SCOPE_SYSTEM : acq, l + 1 == L ? HSA_FENCE_SCOPE_SYSTEM : rel});
        else ps.push_back(Pkt{poll ? &k_poll : &k_layer, ka_dev + ((char*)a - ka), ND / 16, M / 16, bar, l == 0 ? HSA_FENCE_SCOPE_SYSTEM : acq, l + 1 == L ? HSA_FENCE_SCOPE_SYSTEM : rel});
      }
      CK(hsa_memory_copy(ka_dev, ka, ka_bytes));
      *status = 0;
      { const int z[4] = {0, 0, 0, 0}; CK(hsa_memory_copy(hits_dev, z, 16)); }
      double best = 1e30, fill_us = 0.0;
      std::vector<Pkt> fill_only(ps.begin(), ps.begin() + 1);
      for (int r = 0; r < 3; ++r) fill_us = submit(fill_only);   // (the pre-fill alone, to subtract)
      double fb = 1e30;
      for (int r = 0; r < 5; ++r) { const double t = submit(fill_only); if (t < fb) fb = t; }
      fill_us = fb;
      for (int r = 0; r < reps; ++r) { const double t = submit(ps); if (t < best) best = t; if (*status != 0) break; }   // (a pass that gave up is slow: once is enough)
      std::vector<float> out(act_floats);
      CK(hsa_memory_copy(out.data(), acts + act_floats * L, act_bytes));
      double cs = 0.0; int nan = 0;
      for (float v : out) { if (std::isnan(v)) ++nan; else cs += v; }
      printf("  %-58s %8.1f us - fill %6.1f = %7.1f us -> %.3f us per layer   (status %d, checksum %.6f, NaNs %d", label, best, fill_us, best - fill_us,
             (best - fill_us) / L, *status, cs, nan);
      if (xcd_fast >= 0) { int hh[4]; CK(hsa_memory_copy(hh, hits_dev, 16)); printf("; workgroups whose operand was there at the first poll %d, later %d, by system-scope polls %d", hh[0], hh[1], hh[2]); }
      printf(")\n");
      fflush(stdout);
    };
    printf("%d layers of %d x %d x %d, hand-off buffers in %s memory:\n", L, M, KD, ND, fine ? "FINE-grained device" : "coarse-grained device");
    {   // the clocks ramp under load only (idle: a fraction of 2.4 GHz): two seconds of the plain chain before anything is timed
      std::vector<Pkt> ps;
      for (int l = 0; l < L; ++l) {
        ArgsL* p = (ArgsL*)(ka + (size_t)l * 512); p->x = acts + act_floats * l; p->w = w; p->y = acts + act_floats * (l + 1);
        ps.push_back(Pkt{&k_layer, ka_dev + ((char*)p - ka), ND / 16, M / 16, true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT});
      }
      CK(hsa_memory_copy(ka_dev, ka, ka_bytes));
      const auto t0 = std::chrono::high_resolution_clock::now();
      double last = 0.0;
      while (std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count() < 2.0) last = submit(ps);
      printf("  (warm-up: the plain chain ends at %.3f us per layer)\n", last / L);
    }
    run("(a) barrier bit, agent acquire / release, plain kernel", false, 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0);
    run("(a') barrier bit, agent fences, polling kernel", true, 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0);
    run("(b) barrier bit, NO fences, polling kernel", true, 1, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0);
    run("(c) NO barrier bit, no fences, polling kernel", true, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0);
    run("(c') the same, pollers sleep between polls", true, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 1);
    run("(e) no barrier bit, row tile <-> XCD grid, L2-level polls (64)", true, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0, 64);
    run("(e') the same grid, system-scope polls only", true, 0, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0, 0);
    run("(e'') the same grid, barrier bit + agent fences", true, 1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, 0, 64);
    run("(d) barrier bit every 2nd packet", true, 2, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0);
    run("(d) barrier bit every 4th packet", true, 4, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0);
    run("(d) barrier bit every 8th packet", true, 8, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, 0);
    hsa_amd_memory_pool_free(acts);
    hsa_amd_memory_pool_free(ka_dev);
  }
  hsa_queue_destroy(g_q);
  hsa_shut_down();
  return 0;
}
