/* Host-side sanitizer harness (SURVEY §5 "sanitizers"; CPU only — never run on the GPU box).
 * Built with gcc -fsanitize=address,undefined and linked against tm-gcn_amd/libtmgcn_hip.so, it calls EVERY entry
 * point of include/tmgcn.h with invalid, null and zero-size arguments.  The contract under test: argument validation
 * (TMGCN_REQUIRE) returns a negative status and sets tmgcn_last_error() BEFORE any device work, zero-size calls are
 * no-ops, and nothing is read or written through the bogus pointers — so the program must finish without a sanitizer
 * report, with or without a GPU.  `make -C tests/sanitize run` (tests/test_sanitize.py). */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "tmgcn.h"

static int failures = 0, calls = 0;

static void expect(const char* what, long long rc, int want_negative) {
  ++calls;
  const int ok = want_negative ? (rc < 0) : (rc == 0);
  if (!ok) {
    ++failures;
    printf("FAIL %-44s rc=%lld (%s)\n", what, rc, want_negative ? "expected a negative status" : "expected 0");
  } else if (want_negative) {
    const char* e = tmgcn_last_error();
    if (!e || !e[0]) {
      ++failures;
      printf("FAIL %-44s negative status without an error message\n", what);
    }
  }
}
#define BAD(call) expect(#call, (long long)(call), 1)
#define NOP(call) expect(#call, (long long)(call), 0)

int main(void) {
  /* a little real host memory for the entry points that take HOST arrays */
  void* hp[2] = {0, 0};
  const void* hg[2] = {0, 0};
  int64_t hn[2] = {4, 4};

  if (tmgcn_abi_version() != 5) { printf("FAIL abi version %d\n", tmgcn_abi_version()); return 1; }
  if (!tmgcn_last_error()) { printf("FAIL tmgcn_last_error() is NULL\n"); return 1; }

  { int64_t st[6]; BAD(tmgcn_pool_stats(0, 6)); BAD(tmgcn_pool_stats(st, 5)); }             /* null / short output */
  /* P1 */
  BAD(tmgcn_mtransform_f32(0, 4, 4, 0, 0, 0, 4, 4, 3, 0, 0, 0, 16, 0, 0, 0));            /* null M / X / Y */
  BAD(tmgcn_mtransform_f32(0, -1, 4, 0, 0, 0, 4, 4, 3, 0, 0, 0, 16, 0, 0, 0));           /* negative T */
  BAD(tmgcn_mtransform_ld_f32(0, 4, 4, 0, 0, 0, 4, 4, 3, 0, 0, 2, 0, 2, 16, 0, 0, 0));   /* ld < C, nulls */
  BAD(tmgcn_head_loss_combine_f32(0, 3, 0, 10, 6, 0));                                                     /* null pointers */
  NOP(tmgcn_head_loss_combine_f32(0, 0, 0, 10, 6, 0));                                                     /* nothing split */
  /* P2 */
  BAD(tmgcn_spmm_csr_batched_f32_plan(0, 0, 0, 0, 0, 9, 3, 128, -1.f, 0, 0, -1, 0, 0, 0, 0));              /* negative giant count */
  BAD(tmgcn_spmm_gemm_f32_plan(0, 0, 0, 0, 9, 3, 128, 0, 128, 0, 0, 0, 0, 0, 0, 0, 0, -1.f, 0, 0, 2, 1, 0, 0, 0));  /* nulls */
  if (tmgcn_spmm_giant_workspace_bytes(3, 128) != 3 * 128 * 4 || tmgcn_spmm_giant_workspace_bytes(0, 128) != 0) { ++failures; printf("FAIL giant workspace bytes\n"); }
  BAD(tmgcn_spmm_csr_batched_f32(0, 0, 0, 0, 0, 10, 3, 4, 0));                            /* rows not a multiple of N */
  BAD(tmgcn_spmm_csr_batched_f32(0, 0, 0, 0, 0, 9, 3, 4, 0));                             /* null pointers */
  BAD(tmgcn_spmm_csr_batched_f32_hint(0, 0, 0, 0, 0, 9, 3, 0, 1.0f, 0));                  /* F = 0 */
  NOP(tmgcn_spmm_gemm_supported(6, 300) != 0);
  BAD(tmgcn_spmm_gemm_f32(0, 0, 0, 0, 9, 3, 20, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0));        /* unsupported K */
  BAD(tmgcn_spmm_gemm_f32(0, 0, 0, 0, 9, 3, 128, 0, 128, 0, 0, 0, 0, 0, 0, 0, 0, 0));     /* null pointers */
  BAD(tmgcn_spmm_gemm_f32_hint(0, 0, 0, 0, 9, 3, 128, 0, 128, 0, 0, 0, 9, 0, 0, 0, 0, 3.f, 0));   /* bad activation id */
  /* P3 */
  BAD(tmgcn_gemm_f32(0, 0, 0, 0, 10, 0, 4, 0, 0, 0, 0, 0, 0));                            /* K = 0 */
  BAD(tmgcn_gemm_f32(0, 0, 0, 0, 10, 4, 4, 0, 0, 0, 0, 0, 0));                            /* null pointers */
  BAD(tmgcn_gemm_f32(0, 0, 0, 0, 10, 4, 4, 0, 0, 0, 7, 0, 0));                            /* unknown activation */
  NOP(tmgcn_gemm_f32(0, 0, 0, 0, 0, 4, 4, 0, 0, 0, 0, 0, 0));                             /* R = 0: nothing to do */
  BAD(tmgcn_gemm_bf16w_f32(0, 0, 0, 0, 10, 4, 4, 0, 0, 0, 0, 5, 0));                      /* unknown algo */
  BAD(tmgcn_gemm_dw_f32(0, 0, 0, 10, 4, 4, 0, 0, 0, 0, 0));                               /* null dW */
  BAD(tmgcn_gemm_dw_f32(0, 0, (float*)hn, 10, 0, 4, 0, 0, 0, 0, 0));                      /* K = 0 */
  NOP(tmgcn_gemm_dw_act_supported(16, 16));
  BAD(tmgcn_gemm_dw_act_f32(0, 0, 0, 3, (float*)hn, 10, 16, 16, 0, 0, 0, 0));             /* wide: not supported */
  BAD(tmgcn_gemm_dw_act_f32(0, 0, 0, 3, (float*)hn, 10, 2, 6, 0, 0, 0, 0));               /* no pre-activation */
  if (tmgcn_gemm_dw_workspace_bytes(1000, 6, 6, 0) <= 0) { ++failures; printf("FAIL gemm_dw workspace size\n"); }
  /* P5 */
  BAD(tmgcn_act_fwd_f32(0, 0, -1, 1, 0));
  BAD(tmgcn_act_fwd_f32(0, 0, 8, 9, 0));
  NOP(tmgcn_act_fwd_f32(0, 0, 0, 1, 0));
  BAD(tmgcn_act_bwd_f32(0, 0, 0, 8, 1, 0));
  /* P4 */
  NOP(tmgcn_edge_head_supported(300, 2));
  BAD(tmgcn_edge_head_fwd_f32(0, 0, 0, 0, 0, 5, 300, 2, 0));                              /* F too wide */
  BAD(tmgcn_edge_head_fwd_f32(0, 0, 0, 0, 0, 5, 6, 2, 0));                                /* null pointers */
  NOP(tmgcn_edge_head_fwd_i32_f32(0, 0, 0, 0, 0, 0, 6, 2, 0));                            /* E = 0 */
  BAD(tmgcn_edge_head_bwd_f32(0, 0, 0, 0, 0, 0, 0, (float*)hn, 0, 4, 5, 6, 2, 0, 0, 0));  /* dZ wanted, no index */
  BAD(tmgcn_edge_head_bwd_i32_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, (int64_t)1 << 32, 5, 6, 2, 0, 0, 0));   /* R does not fit 31 bits */
  if (tmgcn_edge_head_bwd_workspace_bytes(1000, 6, 2) <= 0) { ++failures; printf("FAIL edge_head workspace size\n"); }
  /* loss */
  BAD(tmgcn_wce_fwd_f32(0, 0, 0, 0, 2, -100, 0, 0, 0, 0, 0));                             /* E = 0 */
  BAD(tmgcn_wce_fwd_f32(0, 0, 0, 5, 2, 1, 0, 0, 0, 0, 0));                                /* ignore_index names a class */
  BAD(tmgcn_wce_fwd_f32(0, 0, 0, 5, 9, -100, 0, 0, 0, 0, 0));                             /* C too large */
  BAD(tmgcn_wce_bwd_f32(0, 0, 0, 0, 0, 5, 2, -100, 0, 0));
  if (tmgcn_wce_workspace_bytes(1000) <= 0) { ++failures; printf("FAIL wce workspace size\n"); }
  /* one-pass head + loss */
  NOP(tmgcn_head_loss_supported(16, 2, 0));
  BAD(tmgcn_head_loss_f32(0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 4, 4, 16, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0));   /* F unsupported */
  BAD(tmgcn_head_loss_f32(0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 4, 4, 6, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0));    /* null pointers */
  BAD(tmgcn_head_loss_f32(0, 0, 0, 0, 0, 0, 9, 0, 0, 0, 0, 0, 0, 4, 4, 6, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0));    /* n_active > R */
  if (tmgcn_head_loss_workspace_bytes(6, 2, 2) <= 0 || tmgcn_head_loss_workspace_bytes(16, 2, 0) != 0) {
    ++failures;
    printf("FAIL head_loss workspace size\n");
  }
  {
    TmgcnSgd sgd = {0, 0, .01f, .9f, 0.f, 0.f, 0, 0, 0};
    BAD(tmgcn_head_loss_sgd_f32(0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 4, 4, 6, 2, 0, 0, 0, &sgd, 0, 0, 0, 0));     /* not the folded form */
    BAD(tmgcn_head_loss_sgd_f32(0, 0, 2, 0, 0, 0, 1, 0, 0, 0, 0, 4, 4, 6, 2, 0, 0, 0, 0, 0, 0, 0, 0));        /* no optimizer settings */
  }
  /* layers 1 + 2 fused */
  NOP(tmgcn_layer12_supported(3, 6, 6));
  NOP(tmgcn_layer12_fwd_pays(0, 0, 6, 1.f));
  BAD(tmgcn_layer12_fwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 3, 6, 6, 0, 0, 0, 1.f, 0, 0, 0));                 /* widths unsupported */
  BAD(tmgcn_layer12_fwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 2, 6, 6, 0, 0, 0, 1.f, 0, 0, 0));                 /* null pointers */
  BAD(tmgcn_layer12_fwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 5, 4, 2, 6, 6, 0, 0, 0, 1.f, 0, 0, 0));                 /* rows not a multiple of N */
  NOP(tmgcn_layer12_fwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 2, 6, 6, 0, 0, 0, 1.f, 0, 0, 0));                 /* nothing to do */
  BAD(tmgcn_layer12_fwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 2, 6, 6, 0, 0, 0, 1.f, 0, 3, 0));                 /* a block count without a partition */
  BAD(tmgcn_layer12_bwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 2, 5, 6, 0, 0, 0, 1.f, 0, 0, 0, 0, 0));           /* odd width */
  BAD(tmgcn_layer12_bwd_f32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 2, 6, 6, 0, 0, 0, 1.f, 0, 0, 0, 0, 0));           /* null pointers */
  NOP(tmgcn_layer12_bwd_forms_dw2(0, 0, 6, 6, 1.f, 0));
  if (tmgcn_layer12_bwd_workspace_bytes(2, 6, 6, 1000, 0) <= 0) { ++failures; printf("FAIL layer12 workspace size\n"); }
  BAD(tmgcn_scale2_f32(0, 0, 0, 4, 0, 0, 4, 0));
  BAD(tmgcn_cast_multi(0, 0, 0, 2, 0, 0));                                                 /* null host arrays */
  BAD(tmgcn_cast_multi(hg, hp, hn, 2, 1, 0));                                              /* null device pointers */
  NOP(tmgcn_cast_multi(hg, hp, hn, 0, 1, 0));
  /* optimizer step (HOST arrays of device pointers) */
  BAD(tmgcn_sgd_step(hp, hg, hp, hn, 17, 0, .1f, .9f, 0.f, 0.f, 0, 0, 0, 0));             /* too many tensors */
  BAD(tmgcn_sgd_step(hp, hg, hp, hn, 2, 0, .1f, .9f, 0.f, 0.f, 0, 0, 0, 0));              /* null device pointers */
  BAD(tmgcn_sgd_step(0, 0, 0, 0, 2, 0, .1f, .9f, 0.f, 0.f, 0, 0, 0, 0));                  /* null host arrays */
  NOP(tmgcn_sgd_step(hp, hg, hp, hn, 0, 0, .1f, .9f, 0.f, 0.f, 0, 0, 0, 0));              /* nothing to step */
  /* adjacency pipeline */
  BAD(tmgcn_adj_make_keys(0, 0, 0, 5, 4, 0, 0));
  BAD(tmgcn_coo_sort_reduce(0, 0, 5, 0, 0, 0, 0, 0, 0, 0));
  BAD(tmgcn_adj_symmetrise(0, 0, 5, 4, 0, 0, 0));
  BAD(tmgcn_adj_edge_life(0, 0, 5, 4, 3, 0, 0, 0, 0));                                    /* window = 0 */
  BAD(tmgcn_adj_identity(8, 4, 0, 0, 0));
  BAD(tmgcn_adj_normalise(0, 0, 5, 4, 8, 0, 0, 0));
  BAD(tmgcn_adj_mproduct_expand(0, 0, 5, 4, 3, 0, 3, 1, 1, 0, 0, 0));
  BAD(tmgcn_adj_mproduct_merge_count(0, 0, 8, 4, 2, 0, 2, 70, 0, 0, 0));                  /* band wider than 64 */
  BAD(tmgcn_adj_mproduct_merge_fill(0, 0, 0, 8, 4, 2, 0, 2, 1, 0, 0, 0, 0, 0));
  BAD(tmgcn_adj_keys_to_csr(0, 5, 4, 8, 0, 0, 0));
  BAD(tmgcn_adj_transpose_keys(0, 0, 8, 4, 0, 0));
  if (tmgcn_coo_sort_reduce_workspace_bytes(1000) <= 0) { ++failures; printf("FAIL sort workspace size\n"); }

  printf("%d calls, %d failures\n", calls, failures);
  return failures ? 1 : 0;
}
