// Host-side descriptions of the classes of a multi-class launch, shared by the orchestration (forward.hip), the run-time scheme
// dispatch (dispatch.hip) and the two builds of the split-operand kernels (namespaces s1 = two fp16 planes, s0 = three bf16 planes).
#pragma once

struct AttnClassHost {
  int B, Lq, Lk, A, rep_keys, rep_mult, rep_pos0, nkt;
  long q_row0, q_bs, o_row0, o_bs, img_tile0, pad_off;     // first Q / O row of the class (rows of ldq / ldo floats), first tile
  const int* q_pos;
  const void* mask_tbl;      // causal launches over the token rows: the class's visibility-mask table (attention_bf16x6.hip), or null
};
struct KvClassHost { int B, L, Lreg, rep_k0, nkt; long tile0; };   // B contexts of L rows; class rows follow each other in A / C
struct KvTailHost { int B, key0, n, nkt; long tile0; };
struct KvRowsHost { int B, R, nkt; long row0, tile0; const int* pos; };
