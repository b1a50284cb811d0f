// What can ONE compute unit move to / from HBM when the others are idle?  Every persistent kernel of this repository sees
// "10-12 bytes per cycle and CU" in its memory phases (block tail row phases, GEMM epilogues, the similarity stream) -- which
// is also 1 / 256 of what the chip's HBM delivers, because all 256 workgroups are in the same phase at the same time.  This
// tool separates the two readings: G persistent workgroups (one per CU, G = 8 .. 256) stream disjoint slices of a 2 GB
// buffer -- loads only, stores only, read-modify-write -- with 64 KB in flight each.  If a CU among 8 busy ones moves several
// times what a CU among 256 moves, the figure is the chip's, and spreading the memory phases of the CUs over time would help.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/cu_bw tools/micro/cu_bw.hip && tools/micro/cu_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: loads, 1: stores, 2: read-modify-write (fp32 += 1)
template <int MODE>
__global__ __launch_bounds__(512) void bw_kernel(uint4* __restrict__ buf, long long bytes_per_wg, unsigned* __restrict__ sink) {
  const int tid = threadIdx.x;
  uint4* p = buf + (long long)blockIdx.x * (bytes_per_wg / 16) + tid;
  const long long n = bytes_per_wg / (16 * 512 * 8);          // rounds of 8 x 8 KB
  unsigned acc = 0;
  for (long long r = 0; r < n; ++r) {
    uint4 v[8];
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[u * 512];
    }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        uint4 w;
        if (MODE == 1) w = make_uint4(tid, (unsigned)r, u, 7);
        else w = make_uint4(v[u].x + 1, v[u].y + 1, v[u].z + 1, v[u].w + 1);
        p[u * 512] = w;
      }
    }
    p += 8 * 512;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(uint4* buf, long long total_bytes, int g, unsigned* sink, const char* what) {
  // every workgroup moves the same 8 MB slice count regardless of G, so that a launch lasts long enough at any G
  const long long per = total_bytes / 256;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(bw_kernel<MODE>, dim3(g), dim3(512), 0, 0, buf, per, sink);
  CHECK(hipEventRecord(a));
  const int reps = 3;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(bw_kernel<MODE>, dim3(g), dim3(512), 0, 0, buf, per, sink);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double moved = (double)per * g * (MODE == 2 ? 2 : 1);
  printf("%-18s %3d workgroups: %7.3f ms  %6.2f TB/s chip-wide  %6.1f GB/s per CU = %5.1f bytes per cycle at 2.1 GHz\n", what, g, ms,
         moved / ms / 1e9, moved / g / ms / 1e6, moved / g / ms / 1e6 / 2.1);
}

int main() {
  const long long total = 2LL << 30;
  uint4* buf; unsigned* sink;
  CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, total));
  const int gs[] = {8, 16, 32, 64, 128, 256};
  for (int g : gs) run<0>(buf, total, g, sink, "loads");
  for (int g : gs) run<1>(buf, total, g, sink, "stores");
  for (int g : gs) run<2>(buf, total, g, sink, "read-modify-write");
  return 0;
}
