// fdiv_check: is the division core WITHOUT its scaling / fix-up instructions the IEEE quotient on the operands the compare
// threshold sees?  (GPU box:  scripts/bin/fdiv_check [n_random_pairs]  -- built by __graft_entry__.build().)
//
// include/shaders/urdf_filter.frag:14-17 divides num = z_near z_far / (z_near - z_far) by d = z - z_far / (z_far - z_near) for
// every drawn pixel.  The compiler expands a correctly rounded f32 division into eleven instructions: v_div_scale_f32 twice
// (pre-scaling of operands whose exponents would make an intermediate overflow or go denormal), v_rcp_f32, a Newton step,
// the quotient with two residual corrections (the last one as v_div_fmas_f32, which undoes the scaling) and v_div_fixup_f32
// (zero, infinite, NaN and denormal operands).  With |num| and |d| within 2^+-40 no scaling happens and no special operand
// exists, so the eight instructions in between -- one v_rcp_f32 and seven 2-cycle fma / mul -- should produce the same
// bits.  This program does not argue that, it checks it: for the library's default constants and for random (num, off)
// pairs of the admitted domain, every float z in [-1, 1 + 2^-11] (2.1e9 bit patterns per pair).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

__device__ __forceinline__ float div_core(float n, float d)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __fmaf_rn(-d, r, 1.0f);
  r = __fmaf_rn(e, r, r);
  float q = __fmul_rn(n, r);
  float res = __fmaf_rn(-d, q, n);
  q = __fmaf_rn(res, r, q);
  res = __fmaf_rn(-d, q, n);
  return __fmaf_rn(res, r, q);
}

__global__ void check_kernel(float num, float off, uint32_t first_bits, uint32_t count, unsigned long long* bad, uint32_t* first_bad)
{
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const float z = __uint_as_float(first_bits + i);
    const float d = __fsub_rn(z, off);
    const float a = __fdiv_rn(num, d), b = div_core(num, d);
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) {
      if (mine == 0) atomicMin(first_bad, first_bits + i);
      mine++;
    }
  }
  if (mine) atomicAdd(bad, mine);
}

static float rnd01(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); }

int main(int argc, char** argv)
{
  const int n_random = argc > 1 ? atoi(argv[1]) : 24;
  unsigned long long* d_bad; uint32_t* d_first;
  (void)hipMalloc(&d_bad, 8); (void)hipMalloc(&d_first, 4);
  struct Pair { float num, off; const char* what; bool admitted; };
  std::vector<Pair> pairs;
  // (the host's rule, rtuf_api.cpp: |num| within 2^+-40, off in [1 + 2^-10, 2^20]; pairs outside it are checked as well, to show
  // what the rule is for -- with off closer to 1 than the z range reaches, z - off passes through zero and the fix-up matters)
  auto admitted = [](float num, float off) { return fabsf(num) >= 0x1p-40f && fabsf(num) <= 0x1p40f && off >= 1.0f + 0x1p-10f && off <= 0x1p20f; };
  auto consts = [&](float zn, float zf, const char* what) { Pair p; p.num = (zn * zf) / (zn - zf); p.off = zf / (zf - zn); p.what = what; p.admitted = admitted(p.num, p.off); return p; };
  pairs.push_back(consts(0.1f, 10.0f, "library default z_near 0.1 z_far 10"));
  pairs.push_back(consts(0.01f, 100.0f, "z_near 0.01 z_far 100"));
  pairs.push_back(consts(0.3f, 5.0f, "z_near 0.3 z_far 5"));
  pairs.push_back(consts(0.05f, 1000.0f, "z_near 0.05 z_far 1000"));
  pairs.push_back(consts(1.0f, 2.0f, "z_near 1 z_far 2"));
  uint64_t s = 0x2545F4914F6CDD1Dull;
  for (int i = 0; i < n_random; i++) {      // the admitted domain's corners and interior: |num| in [2^-40, 2^40], off in [1 + 2^-10, 2^20]
    Pair p;
    p.num = -ldexpf(1.0f + rnd01(s), (int)(rnd01(s) * 80.0f) - 40);
    p.off = i % 3 == 0 ? 1.0f + ldexpf(1.0f + rnd01(s), -10 + (int)(rnd01(s) * 9.0f)) : ldexpf(1.0f + rnd01(s), (int)(rnd01(s) * 20.0f));
    if (p.off < 1.0f + 0.0009765625f) p.off = 1.0f + 0.0009765625f;
    p.what = "random";
    p.admitted = admitted(p.num, p.off);
    pairs.push_back(p);
  }
  unsigned long long total_bad = 0, total = 0, outside_bad = 0;
  for (const Pair& p : pairs) {
    unsigned long long bad = 0; uint32_t first = 0xffffffffu;
    (void)hipMemcpy(d_bad, &bad, 8, hipMemcpyHostToDevice); (void)hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    // z in [0, 1 + 2^-11]: bit patterns 0 .. 0x3F801000, and the negatives -1 .. -0 (0x80000000 .. 0xBF800000)
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, p.num, p.off, 0u, 0x3F801001u, d_bad, d_first);
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, p.num, p.off, 0x80000000u, 0x3F800001u, d_bad, d_first);
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
    (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("num %.9g off %.9g (%s%s): %llu differing of %llu", p.num, p.off, p.what, p.admitted ? "" : "; OUTSIDE the admitted domain: the library keeps the full expansion", bad, 0x3F801001ull + 0x3F800001ull);
    if (bad) printf("  first z bits 0x%08x", first);
    printf("\n");
    if (p.admitted) { total_bad += bad; total += 0x3F801001ull + 0x3F800001ull; } else outside_bad += bad;
  }
  printf("admitted domain: %llu differing of %llu quotients; outside it: %llu differing\n", total_bad, total, outside_bad);
  return total_bad ? 1 : 0;
}
