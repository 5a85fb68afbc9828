// Per-segment s_memtime accounting of the SHIPPED causal attention kernel (pre-split K/V images staged by LDS-DMA, visibility masks from the
// per-class table: attention_bf16x6_kernel<1, true, true>) at the rollout's context classes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DATT_TIMING -o att_timing2 tools/microbench/att_timing2.hip
#include "../../ctrl-sim_amd/csrc/attention_bf16x6.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
void prof_before(int, hipStream_t) {}
void prof_after(int, double, hipStream_t, double, int) {}
void prof_few(bool) {}
int ctrlsim_option(int) { return 1; }
int* ctrlsim_nonfinite_ptr() { return nullptr; }
unsigned long long* ctrlsim_attn_cprof_ptr() { return nullptr; }
using namespace SPLIT_NS;

static void run(int Actx, int B) {
  const int T = 32;
  const bool plain = Actx == 24;
  const int Ar = plain ? 24 : Actx - 1, Lreg = T * 3 * Ar, rep_keys = plain ? 0 : 3 * T, Lq = Lreg + rep_keys;
  const int nkt = (Lreg + 63) / 64 + (plain ? 0 : 2);
  const size_t nq = (size_t)B * Lq * 768, nimg = (size_t)B * 8 * nkt * KV_IMG;
  std::vector<float> h(nq);
  for (size_t i = 0; i < nq; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  std::vector<_Float16> hi(nimg);
  for (size_t i = 0; i < nimg; ++i) hi[i] = (_Float16)((float)((i * 40503u) % 2001) / 1000.f - 1.f);
  float *qkv, *O; void *img, *tbl;
  hipMalloc(&qkv, nq * 4); hipMalloc(&O, (size_t)B * Lq * 256 * 4); hipMalloc(&img, nimg * 2);
  hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice);
  hipMemcpy(img, hi.data(), nimg * 2, hipMemcpyHostToDevice);
  hipMalloc(&tbl, attn_mask_table_bytes(Lq, nkt));
  AttnClassHost c{B, Lq, Lreg, Ar, rep_keys, 25 - Actx, Lreg, nkt, 0, (long)Lq * 768, 0, (long)Lq * 256, 0, 0, nullptr, tbl};
  launch_attn_mask_tables(1, &c, 0);
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_att_t), z, sizeof(z));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_attention_classes(1, qkv, 768, img, O, 256, nullptr, 1, &c, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_att_t), sizeof(z));
    if (rep == 2) {
      const char* nm[6] = {"top of tile: next tile's DMA issue", "K fragments + QK mfma issue", "mask + softmax (incl. mfma drain)", "P split + V fragments + PV issue",
                           "end of tile (skipped sub-tiles, waits)", "vmcnt wait + barrier"};
      const double sub = (double)z[6], tiles = (double)z[7];
      double tot = 0; for (int i = 0; i < 6; ++i) tot += z[i];
      printf("A' = %d (L = %d, B = %d): %.3f ms; computed sub-tiles per wave-tile %.2f; cycles per computed sub-tile %.0f\n", Actx, Lq, B, ms, sub / tiles, tot / sub);
      for (int i = 0; i < 6; ++i) printf("  %-40s %8.1f cycles/sub-tile  %5.1f %%\n", nm[i], z[i] / sub, 100.0 * z[i] / tot);
    }
  }
  hipFree(qkv); hipFree(O); hipFree(img); hipFree(tbl);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  run(24, B);
  run(12, 2 * B);
  run(8, 3 * B);
  run(5, 4 * B);
  return 0;
}
