// per-segment cycle accounting of attention_bf16x6 (causal, rollout shape): hipcc -DATT_TIMING tools/microbench/att_timing.hip
#include "../ctrl-sim_amd/csrc/attention_bf16x6.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
void prof_before(int, hipStream_t) {}
void prof_after(int, double, hipStream_t) {}
int ctrlsim_option(int) { return 1; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, L = 2304, A = 24;
  size_t n = (size_t)B * L * 768;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  float *qkv, *O;
  hipMalloc(&qkv, n * 4); hipMalloc(&O, (size_t)B * L * 256 * 4);
  hipMemcpy(qkv, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_att_t), z, sizeof(z));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_attention_bf16x6(1, qkv, 768, (long)L * 768, qkv + 256, qkv + 512, 768, (long)L * 768, O, 256, (long)L * 256, nullptr,
                            nullptr, B, L, L, A, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_att_t), sizeof(z));
    if (rep) {
      const char* nm[6] = {"gload issue", "QK frags+mfma issue", "mask+softmax (incl. mfma drain)", "Psplit+PV issue", "stage split+lds write", "barrier"};
      double sub = (double)z[6], tiles = (double)z[7];
      printf("%.3f ms; subtile-waves %.0f, tile-waves %.0f\n", ms, sub, tiles);
      double tot = 0; for (int i = 0; i < 6; ++i) tot += z[i];
      for (int i = 0; i < 6; ++i) printf("  %-34s %8.1f ticks/subtile  %5.1f %%\n", nm[i], z[i] / sub, 100.0 * z[i] / tot);
      printf("  total %.1f ticks/subtile (s_memtime ticks = 100 MHz? or shader clocks; compare shares)\n", tot / sub);
    }
  }
  return 0;
}
