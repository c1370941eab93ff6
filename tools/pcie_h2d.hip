// tools/pcie_h2d.hip -- the host -> device link's own ceiling: pinned buffers, hipMemcpyAsync in fixed chunks round-robin over a few streams, nothing else.
// What plink2-hip's file -> HBM leg (pread() into a pinned ring + H2D + the count pass, 36-43 GB/s on the round-5 boxes) is held against.
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_h2d.hip -o tools/_bin/pcie_h2d
//   pcie_h2d [--gib 8] [--no-bind]     one JSON line per (chunk MiB, streams) cell, then a summary line
// The buffers are allocated and first-touched by a thread bound to the device's NUMA node (sysfs numa_node of its PCI function) unless --no-bind:
// the other socket's memory costs the inter-socket fabric (30 against 39 GB/s through the loader, profiles/r05_experiments.md section 11).
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t rc_ = (x);                                                             \
    if (rc_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(rc_));                  \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static int numa_node_of_device(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus) - 1, device) != hipSuccess) {
    return -1;
  }
  std::string id(bus);
  for (char& c : id) {
    c = static_cast<char>(tolower(static_cast<unsigned char>(c)));
  }
  FILE* f = fopen(("/sys/bus/pci/devices/" + id + "/numa_node").c_str(), "r");
  int node = -1;
  if (f) {
    if (fscanf(f, "%d", &node) != 1) {
      node = -1;
    }
    fclose(f);
  }
  return node;
}

static int bind_to_node(int node) {
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) {
    return -1;
  }
  char buf[4096];
  const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[got] = 0;
  cpu_set_t want, cur;
  CPU_ZERO(&want);
  for (const char* p = buf; *p;) {
    while (*p && !isdigit(static_cast<unsigned char>(*p))) {
      ++p;
    }
    if (!*p) {
      break;
    }
    char* e = nullptr;
    long a = strtol(p, &e, 10), b = a;
    if (*e == '-') {
      b = strtol(e + 1, &e, 10);
    }
    for (long c = a; (c <= b) && (c < CPU_SETSIZE); ++c) {
      CPU_SET(static_cast<int>(c), &want);
    }
    p = e;
  }
  if (sched_getaffinity(0, sizeof(cur), &cur)) {
    return -1;
  }
  CPU_AND(&want, &want, &cur);
  if (!CPU_COUNT(&want) || sched_setaffinity(0, sizeof(want), &want)) {
    return -1;
  }
  return CPU_COUNT(&want);
}

int main(int argc, char** argv) {
  double gib = 8.0;
  bool bind = true;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--gib") && (i + 1 < argc)) {
      gib = atof(argv[++i]);
    } else if (!strcmp(argv[i], "--no-bind")) {
      bind = false;
    }
  }
  CHECK(hipSetDevice(0));
  const int node = numa_node_of_device(0);
  int cpus = -1;
  if (bind && (node >= 0)) {
    cpus = bind_to_node(node);
  }
  const size_t kMaxChunk = 64ull << 20;
  const int kMaxStreams = 4, kSlots = 8;
  uint8_t* pin[kSlots];
  uint8_t* dev[kSlots];
  for (int k = 0; k < kSlots; ++k) {
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&pin[k]), kMaxChunk, hipHostMallocDefault));
    memset(pin[k], k + 1, kMaxChunk);  // first touch on this thread's node
    CHECK(hipMalloc(reinterpret_cast<void**>(&dev[k]), kMaxChunk));
  }
  hipStream_t st[kMaxStreams];
  for (int s = 0; s < kMaxStreams; ++s) {
    CHECK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
  }
  double best = 0.0, best_d2h = 0.0;
  int best_chunk = 0, best_streams = 0;
  for (int dir = 0; dir < 2; ++dir) {
    for (size_t chunk_mib : {4, 16, 64}) {
      for (int streams : {1, 2, 4}) {
        const size_t chunk = chunk_mib << 20;
        const size_t n = static_cast<size_t>(gib * (1ull << 30) / static_cast<double>(chunk));
        for (int s = 0; s < streams; ++s) {  // warm-up
          CHECK(dir ? hipMemcpyAsync(pin[s], dev[s], chunk, hipMemcpyDeviceToHost, st[s]) : hipMemcpyAsync(dev[s], pin[s], chunk, hipMemcpyHostToDevice, st[s]));
        }
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t q = 0; q < n; ++q) {
          const int s = static_cast<int>(q % streams), k = static_cast<int>(q % kSlots);
          CHECK(dir ? hipMemcpyAsync(pin[k], dev[k], chunk, hipMemcpyDeviceToHost, st[s]) : hipMemcpyAsync(dev[k], pin[k], chunk, hipMemcpyHostToDevice, st[s]));
        }
        CHECK(hipDeviceSynchronize());
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double gbs = static_cast<double>(n) * static_cast<double>(chunk) / sec / 1e9;
        printf("{\"direction\": \"%s\", \"chunk_mib\": %zu, \"streams\": %d, \"gb\": %.2f, \"seconds\": %.4f, \"gb_per_s\": %.2f}\n", dir ? "d2h" : "h2d", chunk_mib, streams,
               static_cast<double>(n) * static_cast<double>(chunk) / 1e9, sec, gbs);
        if (!dir && (gbs > best)) {
          best = gbs;
          best_chunk = static_cast<int>(chunk_mib);
          best_streams = streams;
        }
        if (dir && (gbs > best_d2h)) {
          best_d2h = gbs;
        }
      }
    }
  }
  printf("{\"summary\": \"pinned hipMemcpyAsync, device 0\", \"numa_node\": %d, \"bound_to_node_cpus\": %d, \"h2d_best_gb_per_s\": %.2f, \"h2d_best_chunk_mib\": %d, "
         "\"h2d_best_streams\": %d, \"d2h_best_gb_per_s\": %.2f}\n",
         node, cpus, best, best_chunk, best_streams, best_d2h);
  return 0;
}
