// mfma_probe.hip -- facts about v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands) that pair_mfma_kernel relies on,
// measured on the box instead of assumed:
//   1. C/D layout: lane l, register g  ->  (row of A, column of B)
//   2. K consistency: the same (lane half, VGPR, nibble) position of A and B meets in the contraction
//   3. E2M1 codes: 0x2 = +1, 0xA = -1, 0x0 / 0x8 = 0; E8M0 scale 0x7F = 1
//   4. integer exactness of the f32 accumulation right up to 2^24
//   5. issue rate of the instruction alone, and with the bit-plane -> FP4 unpack beside it
//   6. 2-bit genotype codes -> FP4 directly (3 VALU per 16 samples, v_bitop3_b32), magnitude 2.0 with E8M0 scale 0x7E = 1/2
//      on both operands: the six products of the pair statistics equal the CPU's integer counts
//   7. the instruction's ceiling by operand data and occupancy (zero / random operands, 2 / 4 waves per SIMD) with the
//      effective shader clock of each run: the datasheet rate needs low-toggle operands
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_bin/mfma_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e__ = (x);                                                         \
    if (e__ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ v16f mfma_fp4(const uint32_t (&a)[4], const uint32_t (&b)[4], v16f c) {
  const v8i A = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
  const v8i B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// 16 samples of 2-bit codes (00 hom-REF, 01 het, 10 hom-ALT, 11 missing; sample s at bits 2s, 2s+1) -> two dwords of E2M1
// nibbles: magnitude at nibble bit 2 (value 2.0) = !b0, sign at bit 3 = b1, i.e. x = +2 / 0 / -2 / -0.  f(X) = (X ^ 0x44444444) &
// 0xCCCCCCCC picks the odd samples of X; the even ones are the odd ones of X << 2.  One v_bitop3_b32 each.
__device__ __forceinline__ uint32_t fp4_x(uint32_t X) { return __builtin_amdgcn_bitop3_b32(X, 0x44444444u, 0xccccccccu, 0x28); }  // (a ^ b) & c
// call present (n = !(b0 & b1)) and homozygous (h = !b0 = |x|), both as 2.0 at nibble bit 2
__device__ __forceinline__ uint32_t fp4_n(uint32_t X) { return __builtin_amdgcn_bitop3_b32(X, X >> 1, 0x44444444u, 0x2a); }  // !(a & b) & c
__device__ __forceinline__ uint32_t fp4_h(uint32_t X) { return __builtin_amdgcn_bitop3_b32(X, 0x44444444u, 0x44444444u, 0x28); }  // (a ^ b) & c = !b0 at bit 2

__device__ __forceinline__ v16f mfma_fp4_half(const uint32_t (&a)[4], const uint32_t (&b)[4], v16f c, int scale) {
  const v8i A = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0};
  const v8i B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, scale, 0, scale);
}

// one wave: codes a[lane][2], b[lane][2] (32 samples each) -> the six products, out[p][lane][16]
__global__ void codes_mfma(const uint32_t* a, const uint32_t* b, float* out, int scale) {
  const int l = threadIdx.x;
  uint32_t ax[4], an[4], ah[4], bx[4], bn[4], bh[4];
  for (int q = 0; q < 2; ++q) {
    const uint32_t ca = a[l * 2 + q], cb = b[l * 2 + q];
    ax[2 * q] = fp4_x(ca);
    ax[2 * q + 1] = fp4_x(ca << 2);
    an[2 * q] = fp4_n(ca);
    an[2 * q + 1] = fp4_n(ca << 2);
    ah[2 * q] = fp4_h(ca);
    ah[2 * q + 1] = fp4_h(ca << 2);
    bx[2 * q] = fp4_x(cb);
    bx[2 * q + 1] = fp4_x(cb << 2);
    bn[2 * q] = fp4_n(cb);
    bn[2 * q + 1] = fp4_n(cb << 2);
    bh[2 * q] = fp4_h(cb);
    bh[2 * q + 1] = fp4_h(cb << 2);
  }
  v16f z;
  for (int g = 0; g < 16; ++g) {
    z[g] = 0.f;
  }
  const v16f c0 = mfma_fp4_half(ax, bx, z, scale);  // x_i . x_j
  const v16f c1 = mfma_fp4_half(an, bn, z, scale);  // n_i . n_j
  const v16f c2 = mfma_fp4_half(an, bh, z, scale);  // n_i . h_j
  const v16f c3 = mfma_fp4_half(an, bx, z, scale);  // n_i . x_j
  const v16f c4 = mfma_fp4_half(ah, bn, z, scale);  // h_i . n_j
  const v16f c5 = mfma_fp4_half(ax, bn, z, scale);  // x_i . n_j
  for (int g = 0; g < 16; ++g) {
    out[(0 * 64 + l) * 16 + g] = c0[g];
    out[(1 * 64 + l) * 16 + g] = c1[g];
    out[(2 * 64 + l) * 16 + g] = c2[g];
    out[(3 * 64 + l) * 16 + g] = c3[g];
    out[(4 * 64 + l) * 16 + g] = c4[g];
    out[(5 * 64 + l) * 16 + g] = c5[g];
  }
}

// 7: the instruction alone, NACC independent accumulators, W waves per SIMD; clk[0..1] = shader-clock / 100 MHz wall-clock ticks of block 0
template <int NACC, int W>
__global__ __launch_bounds__(256, W) void peak_kernel(const uint32_t* src, float* out, int iters, unsigned long long* clk) {
  const int l = threadIdx.x;
  v16f acc[NACC];
  for (int p = 0; p < NACC; ++p) {
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }
  const uint32_t fa[4] = {src[l], src[l + 256], src[l + 512], src[l + 768]};
  const uint32_t fb[4] = {src[l + 1024], src[l + 1280], src[l + 1536], src[l + 1792]};
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < NACC; ++p) {
      const v8i A = {(int)fa[0], (int)fa[1], (int)fa[2], (int)fa[3], 0, 0, 0, 0};
      const v8i B = {(int)fb[0], (int)fb[1], (int)fb[2], (int)fb[3], 0, 0, 0, 0};
      acc[p] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[p], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int p = 0; p < NACC; ++p) {
    for (int g = 0; g < 16; ++g) {
      sum += acc[p][g];
    }
  }
  out[blockIdx.x * 256 + l] = sum;
  if ((blockIdx.x == 0) && (l == 0)) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

// one wave: a[lane][4], b[lane][4], cin[lane][16] -> cout[lane][16]
__global__ void one_mfma(const uint32_t* a, const uint32_t* b, const float* cin, float* cout) {
  const int l = threadIdx.x;
  uint32_t A[4], B[4];
  v16f C;
  for (int q = 0; q < 4; ++q) {
    A[q] = a[l * 4 + q];
    B[q] = b[l * 4 + q];
  }
  for (int g = 0; g < 16; ++g) {
    C[g] = cin[l * 16 + g];
  }
  C = mfma_fp4(A, B, C);
  for (int g = 0; g < 16; ++g) {
    cout[l * 16 + g] = C[g];
  }
}

// bit-planes -> FP4: 32 samples (hom dword H, sign dword R) -> 4 dwords of E2M1 nibbles (bit 1 = H, bit 3 = R)
__device__ __forceinline__ void unpack32(uint32_t H, uint32_t R, uint32_t (&o)[4]) {
  const uint32_t t0 = (H & 0x33333333u) | ((R << 2) & 0xccccccccu);
  const uint32_t t1 = ((H >> 2) & 0x33333333u) | (R & 0xccccccccu);
  o[0] = (t0 << 1) & 0xaaaaaaaau;
  o[1] = t0 & 0xaaaaaaaau;
  o[2] = (t1 << 1) & 0xaaaaaaaau;
  o[3] = t1 & 0xaaaaaaaau;
}

// rate: NACC independent accumulators, `iters` rounds; mode 0 = MFMA only, 1 = + one unpack32 per MFMA (VALU beside it),
// 2 = 7 unpacks per 8 MFMAs with operands read from LDS (the shape of pair_mfma_kernel's k-step)
template <int MODE, int PAT = 0>
__global__ __launch_bounds__(256, 2) void rate_kernel(const uint32_t* src, float* out, int iters, const uint32_t* big = nullptr, size_t big_rows = 1) {
  __shared__ uint32_t lds[8192];
  const int l = threadIdx.x;
  for (int q = l; q < 8192; q += 256) {
    lds[q] = src[q];
  }
  __syncthreads();
  v16f acc[8];
  for (int p = 0; p < 8; ++p) {
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }
  uint32_t fa[4] = {src[l], src[l + 256], src[l + 512], src[l + 768]};
  uint32_t fb[4] = {src[l + 1024], src[l + 1280], src[l + 1536], src[l + 1792]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        acc[p] = mfma_fp4(fa, fb, acc[p]);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        uint32_t f[4];
        unpack32(fa[p & 3] + it, fb[p & 3] ^ it, f);
        acc[p] = mfma_fp4(f, fb, acc[p]);
      }
    } else if (MODE >= 3) {
      // the shape of pair_mfma_kernel's stage: (MODE 4, 5) this wave's share of the next stage's LDS-DMA (22 KiB per stage and
      // workgroup = 5.5 x 1 KiB per wave, from a 12.5 GB-like strided source), a counted vmcnt wait, one s_barrier, then 20
      // unpacks and 32 MFMAs from the ring.  MODE 3: barrier only.  MODE 5: the DMA without the barrier (unsafe, rate only).
      uint32_t* ring = lds;  // 2 x 16 KiB halves
      const int half = it & 1;
      if (MODE >= 4) {
        const uint8_t* g = reinterpret_cast<const uint8_t*>(big);
        // DMA instruction (t, wave) = 1 KiB of this stage for rows of 12,544 bytes, as pair_mfma_kernel issues them.
        // PAT 0: 16 rows x (32 B of the hom half-chunk + 32 B of the ref2het half-chunk, 64 B apart: the committed layout)
        // PAT 1: 16 rows x 64 contiguous bytes     PAT 2: 8 rows x 128 contiguous bytes (a full line per row)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const int lane = l & 63;
          const int per_row = (PAT == 2) ? 8 : 4;
          const size_t row = (static_cast<size_t>(blockIdx.x) * 352 + static_cast<size_t>(t * 4 + (l >> 6)) * 16 + lane / per_row) % big_rows;
          const int pc = lane % per_row;
          size_t off;
          if (PAT == 0) {
            off = static_cast<size_t>(it / 2 % 98) * 128 + (it & 1) * 32 + (pc >> 1) * 64 + (pc & 1) * 16;
          } else if (PAT == 1) {
            off = static_cast<size_t>(it % 196) * 64 + pc * 16;
          } else {
            off = static_cast<size_t>(it % 98) * 128 + pc * 16;
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + row * 12544 + off),
                                           (__attribute__((address_space(3))) void*)(ring + (half ^ 1) * 4096 + ((t * 4 + (l >> 6)) % 16) * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      if (MODE != 5) {
        __builtin_amdgcn_s_barrier();
      }
      const uint4* l4 = reinterpret_cast<const uint4*>(ring + half * 4096);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint32_t fr[5][4];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const uint4 hv = l4[((u * 4 + ks) * 64 + (l & 63)) & 1023];
          unpack32(hv.x, hv.y, fr[u]);
        }
        acc[0] = mfma_fp4(fr[2], fr[0], acc[0]);
        acc[1] = mfma_fp4(fr[3], fr[0], acc[1]);
        acc[2] = mfma_fp4(fr[4], fr[0], acc[2]);
        acc[3] = mfma_fp4(fr[0], fr[0], acc[3]);
        acc[4] = mfma_fp4(fr[3], fr[1], acc[4]);
        acc[5] = mfma_fp4(fr[4], fr[1], acc[5]);
        acc[6] = mfma_fp4(fr[0], fr[1], acc[6]);
        acc[7] = mfma_fp4(fr[1], fr[1], acc[7]);
      }
    } else {
      const uint4* l4 = reinterpret_cast<const uint4*>(lds);
      const int base = ((it & 7) * 7) * 64 + (l & 63);
      uint32_t fr[7][4];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const uint4 hv = l4[base + u * 64];
        unpack32(hv.x, hv.y, fr[u]);
      }
      acc[0] = mfma_fp4(fr[2], fr[0], acc[0]);
      acc[1] = mfma_fp4(fr[3], fr[0], acc[1]);
      acc[2] = mfma_fp4(fr[4], fr[0], acc[2]);
      acc[3] = mfma_fp4(fr[5], fr[0], acc[3]);
      acc[4] = mfma_fp4(fr[3], fr[1], acc[4]);
      acc[5] = mfma_fp4(fr[4], fr[1], acc[5]);
      acc[6] = mfma_fp4(fr[5], fr[1], acc[6]);
      acc[7] = mfma_fp4(fr[6], fr[1], acc[7]);
    }
  }
  float s = 0.f;
  for (int p = 0; p < 8; ++p) {
    for (int g = 0; g < 16; ++g) {
      s += acc[p][g];
    }
  }
  out[blockIdx.x * 256 + l] = s;
}

static void run_one(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b, const std::vector<float>& cin, std::vector<float>& cout,
                    uint32_t* da, uint32_t* db, float* dc, float* dd) {
  CHECK(hipMemcpy(da, a.data(), 256 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b.data(), 256 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, cin.data(), 1024 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  CHECK(hipDeviceSynchronize());
  cout.resize(1024);
  CHECK(hipMemcpy(cout.data(), dd, 1024 * 4, hipMemcpyDeviceToHost));
}

int main(int argc, char** argv) {
  const bool rates_only = (argc > 1) && (std::string(argv[1]) == "--rates");  // bench.py: the ceilings only
  uint32_t *da, *db;
  float *dc, *dd;
  CHECK(hipMalloc(&da, 1024));
  CHECK(hipMalloc(&db, 1024));
  CHECK(hipMalloc(&dc, 4096));
  CHECK(hipMalloc(&dd, 4096));
  std::vector<uint32_t> a(256), b(256);
  std::vector<float> cin(1024, 0.f), c;
  int bad = 0;

  if (!rates_only) {
  // 1. layout: A row r = all (+1) in every k for lanes with (l & 31) == r; B all ones.  C[l][g] = 64 iff row(l, g) == r.
  std::vector<int> row_of(1024, -1), col_of(1024, -1);
  for (int r = 0; r < 32; ++r) {
    for (int l = 0; l < 64; ++l) {
      for (int q = 0; q < 4; ++q) {
        a[l * 4 + q] = ((l & 31) == r) ? 0x22222222u : 0u;
        b[l * 4 + q] = 0x22222222u;
      }
    }
    run_one(a, b, cin, c, da, db, dc, dd);
    for (int e = 0; e < 1024; ++e) {
      if (c[e] == 64.f) {
        row_of[e] = r;
      } else if (c[e] != 0.f) {
        printf("layout(row): unexpected value %g at lane %d reg %d\n", c[e], e / 16, e % 16);
        ++bad;
      }
    }
    for (int l = 0; l < 64; ++l) {
      for (int q = 0; q < 4; ++q) {
        a[l * 4 + q] = 0x22222222u;
        b[l * 4 + q] = ((l & 31) == r) ? 0x22222222u : 0u;
      }
    }
    run_one(a, b, cin, c, da, db, dc, dd);
    for (int e = 0; e < 1024; ++e) {
      if (c[e] == 64.f) {
        col_of[e] = r;
      }
    }
  }
  int layout_ok = 1;
  for (int l = 0; l < 64; ++l) {
    for (int g = 0; g < 16; ++g) {
      const int want_row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
      const int want_col = l & 31;
      if ((row_of[l * 16 + g] != want_row) || (col_of[l * 16 + g] != want_col)) {
        layout_ok = 0;
      }
    }
  }
  printf("1. C/D layout row=(g&3)+8*(g>>2)+4*(l>>5) [A operand's lane&31], col=l&31 [B operand's lane&31]: %s\n", layout_ok ? "CONFIRMED" : "DIFFERENT");
  if (!layout_ok) {
    ++bad;
    for (int l = 0; l < 64; l += 9) {
      printf("   lane %2d:", l);
      for (int g = 0; g < 16; ++g) {
        printf(" (%d,%d)", row_of[l * 16 + g], col_of[l * 16 + g]);
      }
      printf("\n");
    }
  }

  // 2. K consistency: one-hot nibble at position pa of A row 0, pb of B col 0 (position = half*32 + vgpr*8 + nibble)
  int k_ok = 1, k_pairs = 0;
  for (int pa = 0; pa < 64; ++pa) {
    for (int pb = 0; pb < 64; ++pb) {
      if ((pa != pb) && ((pa * 7 + pb * 3) % 5)) {
        continue;  // all equal positions + a fifth of the unequal ones
      }
      std::fill(a.begin(), a.end(), 0u);
      std::fill(b.begin(), b.end(), 0u);
      a[((pa >> 5) * 32 + 0) * 4 + ((pa >> 3) & 3)] = 0x2u << (4 * (pa & 7));
      b[((pb >> 5) * 32 + 0) * 4 + ((pb >> 3) & 3)] = 0x2u << (4 * (pb & 7));
      run_one(a, b, cin, c, da, db, dc, dd);
      const float want = (pa == pb) ? 1.f : 0.f;
      if (c[0] != want) {
        if (k_ok) {
          printf("   K mismatch: A position %d x B position %d -> %g (expected %g)\n", pa, pb, c[0], want);
        }
        k_ok = 0;
      }
      ++k_pairs;
    }
  }
  printf("2. K consistency (same lane half / VGPR / nibble of A and B contract), %d position pairs: %s\n", k_pairs, k_ok ? "CONFIRMED" : "DIFFERENT");
  bad += !k_ok;

  // 3. code table
  {
    const uint32_t codes[6] = {0x0, 0x2, 0xA, 0x8, 0x1, 0x4};
    const float vals[6] = {0.f, 1.f, -1.f, 0.f, 0.5f, 2.f};
    int ok = 1;
    for (int x = 0; x < 6; ++x) {
      for (int y = 0; y < 6; ++y) {
        std::fill(a.begin(), a.end(), 0u);
        std::fill(b.begin(), b.end(), 0u);
        a[0] = codes[x];
        b[0] = codes[y];
        run_one(a, b, cin, c, da, db, dc, dd);
        if (c[0] != vals[x] * vals[y]) {
          printf("   code 0x%x x 0x%x -> %g, expected %g\n", codes[x], codes[y], c[0], vals[x] * vals[y]);
          ok = 0;
        }
      }
    }
    printf("3. E2M1 codes 0x2=+1 0xA=-1 0x0/0x8=0 (0x1=0.5 0x4=2), scale 0x7F=1: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    bad += !ok;
  }

  // 4. exactness near 2^24: C0 + 64 * (+1) and C0 - 64, mixed signs
  {
    int ok = 1;
    const float starts[4] = {16777216.f - 64.f, -(16777216.f - 64.f), 16777000.f, 8388607.f};
    for (int t = 0; t < 4; ++t) {
      for (int sign = 0; sign < 2; ++sign) {
        for (int l = 0; l < 64; ++l) {
          for (int q = 0; q < 4; ++q) {
            a[l * 4 + q] = 0x22222222u;
            b[l * 4 + q] = sign ? 0xaaaaaaaau : 0x22222222u;
          }
        }
        std::vector<float> c0(1024, starts[t]);
        run_one(a, b, c0, c, da, db, dc, dd);
        const double want = (double)starts[t] + (sign ? -64.0 : 64.0);
        if ((double)c[5] != want) {
          if (fabs(want) <= 16777216.0) {
            printf("   %.1f %c 64 -> %.1f (expected %.1f)\n", starts[t], sign ? '-' : '+', c[5], want);
            ok = 0;
          }
        }
      }
    }
    // alternating signs inside one instruction: 32 x (+1) and 32 x (-1) on top of a large C
    for (int l = 0; l < 64; ++l) {
      for (int q = 0; q < 4; ++q) {
        a[l * 4 + q] = 0x22222222u;
        b[l * 4 + q] = (l < 32) ? 0x22222222u : 0xaaaaaaaau;
      }
    }
    // (every partial sum must stay an integer below 2^24, whatever order the 64 products are added in: start 64 below)
    std::vector<float> c0(1024, 16777215.f - 64.f);
    run_one(a, b, c0, c, da, db, dc, dd);
    if (c[7] != 16777215.f - 64.f) {
      printf("   16777151 + 32 - 32 -> %.1f\n", c[7]);
      ok = 0;
    }
    c0.assign(1024, 16777215.f);
    run_one(a, b, c0, c, da, db, dc, dd);
    printf("   (for the record, beyond the guarantee the kernel needs: 16777215 + 32 - 32 -> %.1f)\n", c[7]);
    printf("4. integer-exact accumulation up to 2^24: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    bad += !ok;
  }

  // 6. 2-bit codes -> FP4 (3 VALU per 16 samples), six products against CPU counts, scale 0x7E (and 0x7F for the record)
  {
    uint32_t *dca, *dcb;
    float* dout;
    CHECK(hipMalloc(&dca, 64 * 2 * 4));
    CHECK(hipMalloc(&dcb, 64 * 2 * 4));
    CHECK(hipMalloc(&dout, 6 * 1024 * 4));
    std::vector<uint32_t> ca(128), cb(128);
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() {
      rng ^= rng << 13;
      rng ^= rng >> 7;
      rng ^= rng << 17;
      return static_cast<uint32_t>(rng >> 16);
    };
    int ok7e = 1, ok7f = 1;
    for (int trial = 0; trial < 8; ++trial) {
      for (int q = 0; q < 128; ++q) {
        ca[q] = next();
        cb[q] = next();
        if (trial == 0) {
          ca[q] = 0;           // every call hom-REF: x = +1 everywhere
          cb[q] = 0xaaaaaaaau; // every call hom-ALT: x = -1
        }
      }
      CHECK(hipMemcpy(dca, ca.data(), 512, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dcb, cb.data(), 512, hipMemcpyHostToDevice));
      for (int pass = 0; pass < 2; ++pass) {
        const int scale = pass ? 0x7f7f7f7f : 0x7e7e7e7e;
        hipLaunchKernelGGL(codes_mfma, dim3(1), dim3(64), 0, 0, dca, dcb, dout, scale);
        CHECK(hipDeviceSynchronize());
        std::vector<float> o(6 * 1024);
        CHECK(hipMemcpy(o.data(), dout, 6 * 1024 * 4, hipMemcpyDeviceToHost));
        // row r of an operand = lanes r (half 0) and 32 + r (half 1), 32 samples each
        auto code = [&](const std::vector<uint32_t>& c, int r, int s) {  // s < 64
          const int lane = (s >> 5) * 32 + r;
          const int t = s & 31;
          return (c[lane * 2 + (t >> 4)] >> (2 * (t & 15))) & 3u;
        };
        for (int l = 0; l < 64; ++l) {
          for (int g = 0; g < 16; ++g) {
            const int i = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5), j = l & 31;  // i: A row, j: B row
            int want[6] = {0, 0, 0, 0, 0, 0};
            for (int sidx = 0; sidx < 64; ++sidx) {
              const uint32_t gi = code(ca, i, sidx), gj = code(cb, j, sidx);
              const int xi = (gi == 0) ? 1 : ((gi == 2) ? -1 : 0), xj = (gj == 0) ? 1 : ((gj == 2) ? -1 : 0);
              const int ni = (gi != 3), nj = (gj != 3), hi = xi * xi, hj = xj * xj;
              want[0] += xi * xj;
              want[1] += ni * nj;
              want[2] += ni * hj;
              want[3] += ni * xj;
              want[4] += hi * nj;
              want[5] += xi * nj;
            }
            for (int p = 0; p < 6; ++p) {
              const float got = o[(p * 64 + l) * 16 + g];
              const float exp = static_cast<float>(want[p]) * (pass ? 4.f : 1.f);
              if (got != exp) {
                if ((pass ? ok7f : ok7e)) {
                  printf("   codes: product %d pair (%d,%d) scale 0x%02x -> %g, expected %g\n", p, i, j, scale & 0xff, got, exp);
                }
                (pass ? ok7f : ok7e) = 0;
              }
            }
          }
        }
      }
    }
    printf("6. 2-bit codes -> FP4 (magnitude 2.0), six products == CPU counts with E8M0 scale 0x7E (x 1/2 per operand): %s; with 0x7F they are 4x: %s\n",
           ok7e ? "CONFIRMED" : "DIFFERENT", ok7f ? "CONFIRMED" : "DIFFERENT");
    bad += !ok7e;
    CHECK(hipFree(dca));
    CHECK(hipFree(dcb));
    CHECK(hipFree(dout));
  }

  }
  // 7. the instruction's ceiling by operand data and occupancy
  {
    uint32_t* src;
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&src, 2048 * 4));
    const int blocks = 256 * 4 * 8;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int data = 0; data < 3; ++data) {
      std::vector<uint32_t> h(2048);
      for (int q = 0; q < 2048; ++q) {
        h[q] = (data == 0) ? 0u : ((data == 1) ? 0x22222222u : (0x9e3779b9u * (q + 1)));
      }
      CHECK(hipMemcpy(src, h.data(), 2048 * 4, hipMemcpyHostToDevice));
      for (int w = 0; w < 2; ++w) {
        float ms = 0.f;
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
          CHECK(hipEventRecord(e0, 0));
          if (w == 0) {
            hipLaunchKernelGGL((peak_kernel<8, 2>), dim3(blocks), dim3(256), 0, 0, src, out, iters / 2, clk);
          } else {
            hipLaunchKernelGGL((peak_kernel<4, 4>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
          }
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long hc[2];
        CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        const double mfmas = (double)blocks * 4 * iters * 4;
        printf("7. peak (%s operands, %d waves per SIMD, %d accumulators): %.3f ms = %.2f PFLOP/s; shader clock %.0f MHz (clock64 / wall_clock64 of block 0)\n",
               data == 0 ? "zero" : (data == 1 ? "all +1" : "random"), w ? 4 : 2, w ? 4 : 8, ms, 2 * mfmas * 65536.0 / (ms * 1e-3) / 1e15,
               hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0);
      }
    }
    CHECK(hipFree(src));
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
  }
  // 5. rates
  {
    uint32_t* src;
    float* out;
    CHECK(hipMalloc(&src, 8192 * 4));
    std::vector<uint32_t> h(8192);
    for (int q = 0; q < 8192; ++q) {
      h[q] = 0x9e3779b9u * (q + 1);
    }
    CHECK(hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 2 * 8;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    uint32_t* big = nullptr;
    const size_t big_rows = 400000;  // 5 GB of 12,544-byte rows: the DMA of modes 4 and 5 streams from HBM like the kernel's
    CHECK(hipMalloc(&big, big_rows * 12544));
    CHECK(hipMemset(big, 0x5a, big_rows * 12544));
    for (int mode = 3; mode < 8; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        const int stages = 196;
        CHECK(hipEventRecord(e0, 0));
        if (mode == 3) {
          hipLaunchKernelGGL((rate_kernel<3, 0>), dim3(blocks), dim3(256), 0, 0, src, out, stages, big, big_rows);
        } else if (mode == 4) {
          hipLaunchKernelGGL((rate_kernel<4, 0>), dim3(blocks), dim3(256), 0, 0, src, out, stages, big, big_rows);
        } else if (mode == 5) {
          hipLaunchKernelGGL((rate_kernel<5, 0>), dim3(blocks), dim3(256), 0, 0, src, out, stages, big, big_rows);
        } else if (mode == 6) {
          hipLaunchKernelGGL((rate_kernel<4, 1>), dim3(blocks), dim3(256), 0, 0, src, out, stages, big, big_rows);
        } else {
          hipLaunchKernelGGL((rate_kernel<4, 2>), dim3(blocks), dim3(256), 0, 0, src, out, stages, big, big_rows);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) {
          static const char* kName[5] = {"stage shape, barrier only", "stage shape, LDS-DMA ring (2 x 32 B per row, committed layout) + barrier",
                                         "the same without the barrier", "LDS-DMA ring, 64 contiguous bytes per row", "LDS-DMA ring, 128 contiguous bytes per row"};
          const double mfmas = (double)blocks * 4 * stages * 32;
          const double macs = mfmas * 65536.0;
          printf("5. mode %d (%s): %.3f ms, %.2f PFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz, %.2f us per stage and CU, DMA %.2f TB/s\n", mode, kName[mode - 3], ms,
                 2 * macs / (ms * 1e-3) / 1e15, (ms * 1e-3) * 2.4e9 / (mfmas / 1024.0), (ms * 1e-3) / (blocks / 512.0) / stages * 1e6,
                 (mode == 3) ? 0.0 : (double)blocks * stages * 24576.0 / (ms * 1e-3) / 1e12);
        }
      }
    }
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) {
          hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        } else if (mode == 1) {
          hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        } else {
          hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) {
          const double mfmas = (double)blocks * 4 * iters * 8;
          const double macs = mfmas * 65536.0;
          printf("5. mode %d (%s): %.3f ms, %.3e MAC/s = %.2f PFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", mode,
                 mode == 0 ? "MFMA only" : (mode == 1 ? "1 unpack per MFMA" : "LDS read + 7 unpacks per 8 MFMAs"), ms, macs / (ms * 1e-3),
                 2 * macs / (ms * 1e-3) / 1e15, (ms * 1e-3) * 2.4e9 / (mfmas / 1024.0));
        }
      }
    }
  }
  printf("probe: %s\n", bad ? "SOMETHING DIFFERS" : "all assumptions hold");
  return bad ? 1 : 0;
}
