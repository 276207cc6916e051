// valu_peak.hip -- micro-benchmark: measured wave64 VALU issue rate of gfx950 (MI355X) per instruction,
// at 8 waves/SIMD with dependence-free chains (8 independent accumulators per lane).
//
// Why: the raster kernels of this project are VALU-issue-bound, and their "fraction of peak" needs a peak.
// Round 1 asserted 4 cycles per wave64 instruction (614 G/s); the micro-architecture guide lists v_fma_f32 at
// 2 cycles (1,229 G/s).  This program MEASURES the rate of every instruction class the kernels use
// (24-bit integer multiplies, adds, min/max, shifts, compares + selects, conversions, float mul/add/fma,
// reciprocals, 64-bit integer products, f64 fma) so that profiles/valu_peak.json can replace the assertion;
// scripts/valu_mix.py combines it with each kernel's instruction histogram.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_peak.hip -o scripts/bin/valu_peak && scripts/bin/valu_peak > profiles/valu_peak_raw.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 16;     // repetitions of the 8-accumulator group per loop trip
constexpr int kAcc = 8;

// One loop trip issues kUnroll * kAcc * INSTR_PER_GROUP VALU instructions per wave plus ~3 scalar ones.
#define BODY8(ASM, ...)                                                                         \
  asm volatile(ASM : "+v"(a0) : __VA_ARGS__); asm volatile(ASM : "+v"(a1) : __VA_ARGS__);      \
  asm volatile(ASM : "+v"(a2) : __VA_ARGS__); asm volatile(ASM : "+v"(a3) : __VA_ARGS__);      \
  asm volatile(ASM : "+v"(a4) : __VA_ARGS__); asm volatile(ASM : "+v"(a5) : __VA_ARGS__);      \
  asm volatile(ASM : "+v"(a6) : __VA_ARGS__); asm volatile(ASM : "+v"(a7) : __VA_ARGS__);

#define KERNEL(NAME, T, ASM)                                                                    \
  __global__ __launch_bounds__(256) void k_##NAME(T* out, int iters, T bi, T ci)                 \
  {                                                                                             \
    T a0 = (T)(threadIdx.x + 1), a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7; \
    T b = bi, c = ci;                                                                           \
    for (int i = 0; i < iters; i++) {                                                           \
      _Pragma("unroll") for (int u = 0; u < kUnroll; u++) { BODY8(ASM, "v"(b), "v"(c)) }         \
    }                                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
  }

// dst = op(src..., dst): every instruction depends only on its own accumulator (8 chains per lane)
KERNEL(v_fma_f32, float, "v_fma_f32 %0, %1, %2, %0")
KERNEL(v_mul_f32, float, "v_mul_f32 %0, %1, %0")
KERNEL(v_add_f32, float, "v_add_f32 %0, %1, %0")
KERNEL(v_max_f32, float, "v_max_f32 %0, %1, %0")
KERNEL(v_min_f32, float, "v_min_f32 %0, %1, %0")
KERNEL(v_mul_i32_i24, int, "v_mul_i32_i24 %0, %1, %0")
KERNEL(v_mad_i32_i24, int, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL(v_mul_u32_u24, unsigned, "v_mul_u32_u24 %0, %1, %0")
KERNEL(v_add_u32, unsigned, "v_add_u32 %0, %1, %0")
KERNEL(v_sub_u32, unsigned, "v_sub_u32 %0, %1, %0")
KERNEL(v_add3_u32, unsigned, "v_add3_u32 %0, %1, %2, %0")
KERNEL(v_lshl_add_u32, unsigned, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(v_min_i32, int, "v_min_i32 %0, %1, %0")
KERNEL(v_max_i32, int, "v_max_i32 %0, %1, %0")
KERNEL(v_min3_i32, int, "v_min3_i32 %0, %1, %2, %0")
KERNEL(v_and_b32, unsigned, "v_and_b32 %0, %1, %0")
KERNEL(v_or_b32, unsigned, "v_or_b32 %0, %1, %0")
KERNEL(v_or3_b32, unsigned, "v_or3_b32 %0, %1, %2, %0")
KERNEL(v_lshlrev_b32, unsigned, "v_lshlrev_b32 %0, 1, %0")
KERNEL(v_ashrrev_i32, int, "v_ashrrev_i32 %0, 1, %0")
KERNEL(v_bfe_u32, unsigned, "v_bfe_u32 %0, %0, 3, 20")
KERNEL(v_mul_lo_u32, unsigned, "v_mul_lo_u32 %0, %1, %0")
KERNEL(v_mul_hi_u32, unsigned, "v_mul_hi_u32 %0, %1, %0")
KERNEL(v_cvt_f32_i32, float, "v_cvt_f32_i32 %0, %0")
KERNEL(v_cvt_i32_f32, float, "v_cvt_i32_f32 %0, %0")
KERNEL(v_cvt_f32_u32, float, "v_cvt_f32_u32 %0, %0")
KERNEL(v_rndne_f32, float, "v_rndne_f32 %0, %0")
KERNEL(v_rcp_f32, float, "v_rcp_f32 %0, %0")
KERNEL(v_mov_b32, unsigned, "v_mov_b32 %0, %1")
KERNEL(v_cndmask_b32, unsigned, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cmp_gt_i32_plus_cndmask, int, "v_cmp_gt_i32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc")     // 2 instructions
KERNEL(cmp_lt_f32_plus_cndmask, float, "v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc")   // 2 instructions
KERNEL(fma_f32_then_mul_i24, int, "v_fma_f32 %0, %1, %2, %0\n\tv_mul_i32_i24 %0, %1, %0")               // 2 instructions: float / integer alternating
KERNEL(add_then_min_then_mul24, int, "v_add_u32 %0, %1, %0\n\tv_min_i32 %0, %2, %0\n\tv_mul_i32_i24 %0, %1, %0")     // 3 instructions: integer mix
KERNEL(v_lshrrev_b32, unsigned, "v_lshrrev_b32 %0, 1, %0")
KERNEL(v_xor_b32, unsigned, "v_xor_b32 %0, %1, %0")
KERNEL(v_bcnt_u32_b32, unsigned, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(v_mbcnt_lo_u32_b32, unsigned, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(v_mad_u32_u24, unsigned, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(v_fmac_f32, float, "v_fmac_f32 %0, %1, %2")
KERNEL(v_mul_hi_i32_i24, int, "v_mul_hi_i32_i24 %0, %1, %0")
KERNEL(v_bfe_i32, int, "v_bfe_i32 %0, %0, 3, 20")
KERNEL(v_alignbit_b32, unsigned, "v_alignbit_b32 %0, %0, %1, 8")
KERNEL(v_add_lshl_u32, unsigned, "v_add_lshl_u32 %0, %0, %1, 1")
KERNEL(v_med3_i32, int, "v_med3_i32 %0, %1, %2, %0")
KERNEL(v_lshl_add_u64, unsigned long long, "v_lshl_add_u64 %0, %0, 1, %1")
KERNEL(v_cmp_lt_i32_to_vcc, int, "v_cmp_lt_i32 vcc, %1, %0")          // result unused: measures the compare alone
KERNEL(v_cmp_gt_f32_to_vcc, float, "v_cmp_gt_f32 vcc, %1, %0")
KERNEL(v_cmp_lt_i32_to_sgpr_e64, int, "v_cmp_lt_i32 s[20:21], %1, %0")
KERNEL(v_cndmask_b32_e64_sgpr_mask, unsigned, "v_cndmask_b32 %0, %0, %1, s[20:21]")
KERNEL(v_readlane_b32, unsigned, "v_readlane_b32 s20, %0, 5")          // VALU -> SGPR broadcast (result unused)
KERNEL(v_readfirstlane_b32, unsigned, "v_readfirstlane_b32 s20, %0")
KERNEL(min_of_three_edges_then_cmp, int, "v_min3_i32 %0, %1, %2, %0\n\tv_cmp_lt_i32 vcc, 0, %0")      // 2 instructions: the lane walk's coverage test
// 64-bit operands
KERNEL(v_fma_f64, double, "v_fma_f64 %0, %1, %2, %0")
KERNEL(v_mul_f64, double, "v_mul_f64 %0, %1, %0")
KERNEL(v_add_f64, double, "v_add_f64 %0, %1, %0")
KERNEL(v_pk_fma_f32, double, "v_pk_fma_f32 %0, %1, %2, %0")        // two f32 fmas per lane and instruction
KERNEL(v_pk_mul_f32, double, "v_pk_mul_f32 %0, %1, %0")
KERNEL(v_pk_add_f32, double, "v_pk_add_f32 %0, %1, %0")
KERNEL(v_lshlrev_b64, unsigned long long, "v_lshlrev_b64 %0, 1, %0")
// 64-bit accumulate of a 32 x 32-bit product (what a 64-bit edge function costs; the kernels avoid it with 24-bit multiplies)
__global__ __launch_bounds__(256) void k_v_mad_u64_u32(unsigned long long* out, int iters, unsigned long long bi, unsigned long long ci)
{
  unsigned long long a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned b = (unsigned)bi, c = (unsigned)ci + 5u;
  for (int i = 0; i < iters; i++) {
    _Pragma("unroll") for (int u = 0; u < kUnroll; u++) { BODY8("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(b), "v"(c) : "vcc") }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

struct Result { std::string name; int instr_per_group; double ginstr_s; double ms; };

template <typename T>
static Result run(const char* name, void (*kern)(T*, int, T, T), int instr_per_group, int waves_per_simd, int n_cu, T b, T c)
{
  const int blocks = n_cu * waves_per_simd;          // 256-thread blocks = 4 waves: one per SIMD, so n_cu * w blocks = w waves/SIMD
  T* out = nullptr;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(T)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 256;
  float ms = 0;
  for (int attempt = 0; attempt < 6; attempt++) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, b, c);      // warm-up + sizing
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, b, c);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms > 8.0f) break;
    iters *= 2;
  }
  double best = 1e30;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, b, c);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const double waves = (double)blocks * 4.0;
  const double instr = waves * (double)iters * kUnroll * kAcc * instr_per_group;
  CHECK(hipFree(out));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return Result{name, instr_per_group, instr / (best * 1e-3) / 1e9, best};
}

int main(int argc, char** argv)
{
  const int waves = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double nominal_ghz = prop.clockRate * 1e-6;
  std::vector<Result> r;
#define RUN(NAME, T, N, B, C) r.push_back(run<T>(#NAME, k_##NAME, N, waves, n_cu, (T)(B), (T)(C)))
  RUN(v_fma_f32, float, 1, 1.0000001f, 1e-9f); RUN(v_mul_f32, float, 1, 1.0000001f, 0); RUN(v_add_f32, float, 1, 1e-9f, 0);
  RUN(v_max_f32, float, 1, 0.5f, 0); RUN(v_min_f32, float, 1, 1e30f, 0);
  RUN(v_mul_i32_i24, int, 1, 3, 0); RUN(v_mad_i32_i24, int, 1, 3, 5); RUN(v_mul_u32_u24, unsigned, 1, 3, 0);
  RUN(v_add_u32, unsigned, 1, 3, 0); RUN(v_sub_u32, unsigned, 1, 3, 0); RUN(v_add3_u32, unsigned, 1, 3, 5); RUN(v_lshl_add_u32, unsigned, 1, 3, 0);
  RUN(v_min_i32, int, 1, 1 << 30, 0); RUN(v_max_i32, int, 1, 3, 0); RUN(v_min3_i32, int, 1, 1 << 30, 1 << 29);
  RUN(v_and_b32, unsigned, 1, 0xffffffffu, 0); RUN(v_or_b32, unsigned, 1, 1, 0); RUN(v_or3_b32, unsigned, 1, 1, 2);
  RUN(v_lshlrev_b32, unsigned, 1, 0, 0); RUN(v_ashrrev_i32, int, 1, 0, 0); RUN(v_bfe_u32, unsigned, 1, 0, 0);
  RUN(v_mul_lo_u32, unsigned, 1, 3, 0); RUN(v_mul_hi_u32, unsigned, 1, 3, 0);
  RUN(v_cvt_f32_i32, float, 1, 0, 0); RUN(v_cvt_i32_f32, float, 1, 0, 0); RUN(v_cvt_f32_u32, float, 1, 0, 0); RUN(v_rndne_f32, float, 1, 0, 0);
  RUN(v_rcp_f32, float, 1, 0, 0); RUN(v_mov_b32, unsigned, 1, 7, 0); RUN(v_cndmask_b32, unsigned, 1, 7, 0);
  RUN(cmp_gt_i32_plus_cndmask, int, 2, 5, 9); RUN(cmp_lt_f32_plus_cndmask, float, 2, 5.0f, 9.0f);
  RUN(fma_f32_then_mul_i24, int, 2, 3, 5); RUN(add_then_min_then_mul24, int, 3, 3, 1 << 20);
  RUN(v_lshrrev_b32, unsigned, 1, 0, 0); RUN(v_xor_b32, unsigned, 1, 5, 0); RUN(v_bcnt_u32_b32, unsigned, 1, 5, 0); RUN(v_mbcnt_lo_u32_b32, unsigned, 1, 5, 0);
  RUN(v_mad_u32_u24, unsigned, 1, 3, 5); RUN(v_fmac_f32, float, 1, 1e-9f, 1.0f); RUN(v_mul_hi_i32_i24, int, 1, 3, 0); RUN(v_bfe_i32, int, 1, 0, 0);
  RUN(v_alignbit_b32, unsigned, 1, 5, 0); RUN(v_add_lshl_u32, unsigned, 1, 5, 0); RUN(v_med3_i32, int, 1, 5, 9); RUN(v_lshl_add_u64, unsigned long long, 1, 5, 0);
  RUN(v_cmp_lt_i32_to_vcc, int, 1, 5, 0); RUN(v_cmp_gt_f32_to_vcc, float, 1, 5, 0); RUN(v_cmp_lt_i32_to_sgpr_e64, int, 1, 5, 0); RUN(v_cndmask_b32_e64_sgpr_mask, unsigned, 1, 5, 0);
  RUN(v_readlane_b32, unsigned, 1, 0, 0); RUN(v_readfirstlane_b32, unsigned, 1, 0, 0); RUN(min_of_three_edges_then_cmp, int, 2, 1 << 30, 1 << 29);
  RUN(v_fma_f64, double, 1, 1.0000001, 1e-9); RUN(v_mul_f64, double, 1, 1.0000001, 0); RUN(v_add_f64, double, 1, 1e-9, 0);
  RUN(v_pk_fma_f32, double, 1, 1.0, 1e-9); RUN(v_pk_mul_f32, double, 1, 1.0, 0); RUN(v_pk_add_f32, double, 1, 1e-9, 0);
  RUN(v_lshlrev_b64, unsigned long long, 1, 0, 0); RUN(v_mad_u64_u32, unsigned long long, 1, 3, 0);
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"nominal_clock_ghz\": %.3f, \"waves_per_simd\": %d,\n",
         prop.name, prop.gcnArchName, n_cu, n_cu * 4, nominal_ghz, waves);
  printf(" \"note\": \"wave64 VALU instructions per second over the whole GPU, dependence-free chains (8 accumulators per lane), best of 5 launches of >= 8 ms; cycles = SIMDs * nominal clock / rate\",\n");
  printf(" \"ops\": {\n");
  for (size_t i = 0; i < r.size(); i++)
    printf("  \"%s\": {\"G_wave64_instr_per_s\": %.1f, \"cycles_per_wave64_instr\": %.3f, \"instr_per_group\": %d, \"kernel_ms\": %.2f}%s\n", r[i].name.c_str(), r[i].ginstr_s,
           (double)n_cu * 4.0 * nominal_ghz / r[i].ginstr_s, r[i].instr_per_group, r[i].ms, i + 1 < r.size() ? "," : "");
  printf(" }}\n");
  return 0;
}
