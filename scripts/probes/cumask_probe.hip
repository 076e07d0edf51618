// Probe: does hipExtStreamCreateWithCUMask restrict where blocks run, and how do mask bits map to (XCC, SE, CU)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <set>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  // keep the block resident for a while so that all blocks of the grid are co-resident
  long long t0 = clock64();
  while (clock64() - t0 < (long long)spin) { __builtin_amdgcn_s_sleep(10); }
}

static int run(hipStream_t s, int nblk, unsigned* d, std::vector<unsigned>& h, const char* tag) {
  CK(hipMemsetAsync(d, 0xff, nblk * 8, s));
  hipLaunchKernelGGL(where, dim3(nblk), dim3(1024), 0, s, d, 200000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nblk; ++i) {
    unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%s: %d blocks ->", tag, nblk);
  int tot = 0;
  for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  distinct CUs %d\n", tot);
  return 0;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  unsigned* d; CK(hipMalloc(&d, 4096 * 8));
  std::vector<unsigned> h(4096 * 2);
  hipStream_t s0; CK(hipStreamCreate(&s0));
  if (run(s0, 256, d, h, "unmasked")) return 1;
  // masks: first 128 bits, last 128 bits, even bits, 8 words
  uint32_t lo[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0};
  uint32_t hi[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  uint32_t ev[8], od[8];
  for (int i = 0; i < 8; ++i) { ev[i] = 0x55555555u; od[i] = 0xaaaaaaaau; }
  struct { const char* n; uint32_t* m; } ms[2] = {{"mask lo128", lo}, {"mask hi128", hi}};
  for (auto& m : ms) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.m);
    if (e != hipSuccess) { printf("%s: create failed: %s\n", m.n, hipGetErrorString(e)); continue; }
    if (run(s, 128, d, h, m.n)) return 1;
    if (run(s, 256, d, h, m.n)) return 1;
    uint32_t got[8] = {0};
    if (hipExtStreamGetCUMask(s, 8, got) == hipSuccess) printf("   get mask: %08x %08x %08x %08x %08x %08x %08x %08x\n", got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7]);
  }
  // concurrency: two masked streams, each a 128-block grid of spinners: wall time ~ one spin if disjoint
  hipStream_t a, b;
  CK(hipExtStreamCreateWithCUMask(&a, 8, lo));
  CK(hipExtStreamCreateWithCUMask(&b, 8, hi));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned* d2; CK(hipMalloc(&d2, 4096 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, a));
    hipLaunchKernelGGL(where, dim3(128), dim3(1024), 0, a, d, 2000000);
    hipLaunchKernelGGL(where, dim3(128), dim3(1024), 0, b, d2, 2000000);
    CK(hipStreamSynchronize(b));
    CK(hipEventRecord(e1, a));
    CK(hipStreamSynchronize(a));
    float ms_; CK(hipEventElapsedTime(&ms_, e0, e1));
    printf("two masked 128-block spinners concurrently: %.3f ms (one alone ~ %.3f ms at 100 MHz clock64)\n", ms_, 2000000 / 100e6 * 1e3);
  }
  return 0;
}
