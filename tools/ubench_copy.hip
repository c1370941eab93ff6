// ubench_copy.hip -- what a streaming read+write kernel can reach on this GPU, for the access shapes prepare_kernel
// could use.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_copy ubench_copy.hip
//   mode 0: 16-B loads, 16-B stores, fully contiguous (plain copy)
//   mode 1: 16-B loads; 8-B stores alternating between the two 64-B halves of each 128-B line (prepare's layout)
//   mode 2: read only (sum)           mode 3: write only
//   mode 4: 16-B loads, 16-B stores where each half-line (64 B) is written by a separate store instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16, uint32_t* sink) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  uint32_t acc = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
    if (MODE == 0) {
      out[i] = in[i];
    } else if (MODE == 1) {
      const uint4 v = in[i];
      // thread i owns 16 input bytes -> 8 B into half 0 and 8 B into half 1 of line (i / 8)
      uint2* o = reinterpret_cast<uint2*>(out);
      const size_t line = i >> 3, k = i & 7;
      o[line * 16 + k] = make_uint2(v.x, v.y);
      o[line * 16 + 8 + k] = make_uint2(v.z, v.w);
    } else if (MODE == 2) {
      const uint4 v = in[i];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    } else if (MODE == 3) {
      out[i] = make_uint4(static_cast<uint32_t>(i), 1, 2, 3);
    } else {
      // pairs of threads: even thread writes 16 B into half 0, odd thread into half 1 of alternating lines
      const uint4 v = in[i];
      const size_t line = i >> 3, k = i & 7;
      out[line * 8 + ((k & 3) | ((k >> 2) << 2))] = v;
    }
  }
  if (MODE == 2 && acc == 0x12345678u) {
    *sink = acc;
  }
}

// mode 5: four 16-B loads in flight per thread, then four 16-B stores;  mode 6: same with non-temporal stores;
// mode 7: non-temporal loads and stores
template <int MODE>
__global__ __launch_bounds__(256) void copy4_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16, uint32_t* sink) {
  (void)sink;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MODE == 7) {
        v[k].x = __builtin_nontemporal_load(&in[i + k * stride].x);
        v[k].y = __builtin_nontemporal_load(&in[i + k * stride].y);
        v[k].z = __builtin_nontemporal_load(&in[i + k * stride].z);
        v[k].w = __builtin_nontemporal_load(&in[i + k * stride].w);
      } else {
        v[k] = in[i + k * stride];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MODE == 5) {
        out[i + k * stride] = v[k];
      } else {
        __builtin_nontemporal_store(v[k].x, &out[i + k * stride].x);
        __builtin_nontemporal_store(v[k].y, &out[i + k * stride].y);
        __builtin_nontemporal_store(v[k].z, &out[i + k * stride].z);
        __builtin_nontemporal_store(v[k].w, &out[i + k * stride].w);
      }
    }
  }
  for (; i < n16; i += stride) {
    out[i] = in[i];
  }
}

template <int MODE>
int run4(const char* name, const uint4* in, uint4* out, size_t n16, uint32_t* sink, double bytes_moved) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int blocks_per_cu : {2, 4, 8, 16, 64}) {
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(copy4_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n16, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(copy4_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n16, sink);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %5d: %7.3f ms/launch  %6.2f TB/s\n", name, grid, ms / reps, bytes_moved / (ms / reps * 1e-3) / 1e12);
  }
  return 0;
}

template <int MODE>
int run(const char* name, const uint4* in, uint4* out, size_t n16, uint32_t* sink, double bytes_moved) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int blocks_per_cu : {4, 8, 16}) {
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n16, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, in, out, n16, sink);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %5d: %7.3f ms/launch  %6.2f TB/s\n", name, grid, ms / reps, bytes_moved / (ms / reps * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  const size_t bytes = 12ull << 30;  // 12 GiB in, 12 GiB out
  const size_t n16 = bytes / 16;
  uint4 *in, *out;
  uint32_t* sink;
  CK(hipMalloc(&in, bytes));
  CK(hipMalloc(&out, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(in, 1, bytes));
  CK(hipMemset(out, 0, bytes));
  run<0>("copy 16B/16B contiguous", in, out, n16, sink, 2.0 * bytes);
  run<1>("copy 16B loads, 8B stores to line halves", in, out, n16, sink, 2.0 * bytes);
  run<4>("copy 16B loads, 16B stores permuted in line", in, out, n16, sink, 2.0 * bytes);
  run4<5>("copy 4x16B in flight per thread", in, out, n16, sink, 2.0 * bytes);
  run4<6>("copy 4x16B in flight, nontemporal stores", in, out, n16, sink, 2.0 * bytes);
  run4<7>("copy 4x16B in flight, nontemporal ld+st", in, out, n16, sink, 2.0 * bytes);
  run<2>("read only", in, out, n16, sink, 1.0 * bytes);
  run<3>("write only", in, out, n16, sink, 1.0 * bytes);
  return 0;
}
