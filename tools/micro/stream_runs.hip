// What does the HBM deliver when a [384][N] 16-bit matrix (the reference's feature-volume layout: feature-major) is read
// tile by tile, RUN contiguous bytes of every row per tile?  The few-query similarity kernel reads 512-byte runs (256
// voxels x all 384 features per workgroup tile); this measures the access pattern alone -- one persistent 512-thread
// workgroup per CU, 64 KB of loads in flight, nothing but a checksum done with the data.  tools only.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/stream_runs tools/micro/stream_runs.hip && tools/micro/stream_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int RUN>
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ m, long long n_bytes_row, int rows, int ntiles,
                                                     unsigned* __restrict__ sink) {
  constexpr int LPR = RUN / 16;            // lanes per row
  constexpr int RPR = 512 / LPR;           // rows per round of the workgroup
  const int tid = threadIdx.x, sub = tid % LPR, rr = tid / LPR;
  unsigned acc = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const char* base = reinterpret_cast<const char*>(m) + (long long)t * RUN + sub * 16;
    for (int r0 = 0; r0 < rows; r0 += 8 * RPR) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u * RPR + rr;
        v[u] = r < rows ? *reinterpret_cast<const uint4*>(base + (long long)r * n_bytes_row) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;   // (keeps the loads alive)
}

template <int RUN>
static void run(const uint4* m, long long n_bytes_row, int rows, unsigned* sink, int cus, const char* what) {
  const int ntiles = (int)(n_bytes_row / RUN);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_kernel<RUN>, dim3(cus), dim3(512), 0, 0, m, n_bytes_row, rows, ntiles, sink);
  CHECK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_kernel<RUN>, dim3(cus), dim3(512), 0, 0, m, n_bytes_row, rows, ntiles, sink);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = (double)n_bytes_row * rows;
  printf("%-18s runs of %5d bytes: %8.3f ms for %.0f MB = %6.2f TB/s (%.3f of 8 TB/s)\n", what, RUN, ms, bytes / 1e6, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
}

int main() {
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int rows = 384;
  unsigned* sink; CHECK(hipMalloc(&sink, 4));
  for (int big = 0; big < 2; ++big) {
    const long long nvox = big ? 3LL * 262144 : 262144;     // 603 MB: nothing is served by the 256 MB Infinity Cache / 201 MB: resident
    uint4* m; CHECK(hipMalloc(&m, (size_t)nvox * 2 * rows));
    CHECK(hipMemset(m, 1, (size_t)nvox * 2 * rows));
    const char* what = big ? "603 MB (HBM)" : "201 MB (resident)";
    run<256>(m, nvox * 2, rows, sink, cus, what);
    run<512>(m, nvox * 2, rows, sink, cus, what);
    run<1024>(m, nvox * 2, rows, sink, cus, what);
    run<2048>(m, nvox * 2, rows, sink, cus, what);
    run<4096>(m, nvox * 2, rows, sink, cus, what);
    run<8192>(m, nvox * 2, rows, sink, cus, what);
    CHECK(hipFree(m));
  }
  return 0;
}
