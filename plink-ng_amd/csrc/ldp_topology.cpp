// ldp_topology.cpp -- where a device sits in the host (behind include/ldprune_hip.h; ldp_engine.cpp has the overview of the host runtime)
//
// The file -> HBM leg of a load is host threads copying rows into pinned staging memory and DMA engines reading it across PCIe.  On a
// two-socket host both go through the inter-socket fabric when the threads (and with them the first-touched pinned pages) run on the
// socket the device is NOT attached to: 30 GB/s instead of 39 on the round-5 boxes (profiles/r05_experiments.md, section 11).  The library
// does not move its caller's threads; it says where the device is, and plink2-hip binds itself there (p2h_util.cpp: bind_near_device).
#include <hip/hip_runtime.h>

#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/ldprune_hip.h"

extern "C" {

int ldp_device_numa_node(int device) {
  if ((device < 0) || (device >= ldp_device_count())) {
    return -1;
  }
  char bus[64];
  memset(bus, 0, sizeof(bus));
  if (hipDeviceGetPCIBusId(bus, static_cast<int>(sizeof(bus)) - 1, device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  std::string id(bus);
  for (char& c : id) {
    c = static_cast<char>(tolower(static_cast<unsigned char>(c)));
  }
  const std::string path = "/sys/bus/pci/devices/" + id + "/numa_node";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) {
    return -1;
  }
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) {
    node = -1;
  }
  fclose(f);
  return node;
}

}  // extern "C"
