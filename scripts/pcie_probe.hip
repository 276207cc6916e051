// pcie_probe.hip -- what the host link gives the host-plane path (rtuf_filter_batch_async: 315 MB of sensor planes up and
// 393 MB of results down per 256-stream batch, both directions busy at once), by HOW the bytes are moved:
//   dma    hipMemcpyAsync on a stream of its own per direction (the SDMA engines)
//   kernel a streaming copy kernel that reads / writes the pinned host buffer directly (16 bytes per lane, non-temporal)
// one direction alone and both together, in the four combinations.  Stand-alone (no library): built by __graft_entry__.build()
// into scripts/bin/pcie_probe, run on the GPU box:  scripts/bin/pcie_probe > gpurun_out/pcie_probe_hip.txt
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const u32x4 v = __builtin_nontemporal_load(src + i);
    __builtin_nontemporal_store(v, dst + i);
  }
}

struct Dir { void* host; void* dev; size_t bytes; hipStream_t st; };

static void move(const Dir& d, bool up, bool by_kernel, int grid)
{
  if (!d.bytes) return;
  if (by_kernel) {
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, d.st, (u32x4*)(up ? d.dev : d.host), (const u32x4*)(up ? d.host : d.dev), d.bytes / 16);
  } else {
    CHECK(hipMemcpyAsync(up ? d.dev : d.host, up ? d.host : d.dev, d.bytes, up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, d.st));
  }
}

int main(int argc, char** argv)
{
  const size_t up_b = 315ull * 1000 * 1000 / 16 * 16, dn_b = 393ull * 1000 * 1000 / 16 * 16;
  const int grid = argc > 1 ? std::atoi(argv[1]) : 512;
  Dir up{nullptr, nullptr, up_b, nullptr}, dn{nullptr, nullptr, dn_b, nullptr};
  CHECK(hipHostMalloc(&up.host, up_b, hipHostMallocDefault)); CHECK(hipHostMalloc(&dn.host, dn_b, hipHostMallocDefault));
  CHECK(hipMalloc(&up.dev, up_b)); CHECK(hipMalloc(&dn.dev, dn_b));
  std::memset(up.host, 1, up_b); std::memset(dn.host, 2, dn_b);
  CHECK(hipStreamCreateWithFlags(&up.st, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&dn.st, hipStreamNonBlocking));
  std::printf("# pinned host <-> HBM, %zu MB up / %zu MB down per repetition, copy kernel grid %d x 256; HSA_ENABLE_SDMA=%s\n", up_b / 1000000, dn_b / 1000000, grid,
              std::getenv("HSA_ENABLE_SDMA") ? std::getenv("HSA_ENABLE_SDMA") : "(unset)");
  std::printf("%-28s %10s %10s %9s\n", "case", "up GB/s", "down GB/s", "ms / rep");
  for (int which = 0; which < 3; which++)                 // 0: up alone, 1: down alone, 2: both
    for (int uk = 0; uk < 2; uk++)
      for (int dk = 0; dk < 2; dk++) {
        if (which == 0 && dk) continue;
        if (which == 1 && uk) continue;
        const bool do_up = which != 1, do_dn = which != 0;
        auto once = [&] { if (do_up) move(up, true, uk != 0, grid); if (do_dn) move(dn, false, dk != 0, grid); };
        once();
        CHECK(hipDeviceSynchronize());
        const int reps = 8;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) once();
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        char name[64];
        std::snprintf(name, sizeof name, "%s%s%s", do_up ? (uk ? "up:kernel " : "up:dma ") : "", do_dn ? (dk ? "down:kernel" : "down:dma") : "", "");
        std::printf("%-28s %10.1f %10.1f %9.2f\n", name, do_up ? up_b / s / 1e9 : 0.0, do_dn ? dn_b / s / 1e9 : 0.0, s * 1e3);
      }
  return 0;
}
