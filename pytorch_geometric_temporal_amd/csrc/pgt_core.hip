// Error reporting and build identification for libpgt_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pgt_common.h"

static thread_local char g_err[512] = "";

void pgt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pgt_abi_version(void) { return PGT_ABI_VERSION; }
extern "C" const char* pgt_last_error(void) { return g_err; }
extern "C" const char* pgt_build_target(void) { return PGT_TARGET; }

// Tuning / A-B knobs for benchmarking (defaults are the shipped configuration).  Not thread-safe; call between launches.
extern "C" int pgt_tune(const char* key, int value) {
  if (key == nullptr) {
    pgt_set_error("pgt_tune: null key");
    return PGT_ERR_INVALID;
  }
  if (strcmp(key, "gemm_small_tiles") == 0) {
    pgt_gemm_set_force_small(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_small_fill") == 0) {
    pgt_gemm_set_small_fill(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_bx") == 0) {
    pgt_gemm_bx_set(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_bx_sym") == 0) {
    pgt_gemm_bx_sym_set(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_bx_tn_pc") == 0) {
    pgt_gemm_bx_tn_pc_set(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_bx_sym_pc") == 0) {
    pgt_gemm_bx_sym_pc_set(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_tn_fullk") == 0) {
    pgt_gemm_set_tn_fullk(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_tn_pipe") == 0) {
    pgt_gemm_set_tn_pipe(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_dbp") == 0) {
    pgt_gemm_set_dbp(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_skinny") == 0) {
    pgt_gemm_set_skinny(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_pairs") == 0) {
    pgt_slab_set_pairs(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_split") == 0) {
    pgt_slab_set_split(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_threads") == 0) {
    pgt_slab_set_threads(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_wpc") == 0) {
    pgt_slab_set_wpc(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_quad") == 0) {
    pgt_slab_set_quad(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_sort") == 0) {
    pgt_slab_set_sort(value);
    return PGT_OK;
  }
  if (strcmp(key, "tgcn_probe") == 0) {
    PGT_REQUIRE(pgt_tgcn_set_probe(value), "pgt_tune: \"tgcn_probe\" selects kernels that compute wrong results on purpose; this library was built without -DPGT_LAB_PROBES");
    return PGT_OK;
  }
  if (strcmp(key, "tgcn_wgs") == 0) {
    pgt_tgcn_set_wgs(value);
    return PGT_OK;
  }
  if (strcmp(key, "tgcn_rows") == 0) {
    pgt_tgcn_set_rows(value);
    return PGT_OK;
  }
  if (strcmp(key, "slab_gu") == 0) {
    pgt_slab_set_gu(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_db64") == 0) {
    pgt_gemm_set_db64(value);
    return PGT_OK;
  }
  if (strcmp(key, "gemm_db") == 0) {
    pgt_gemm_set_db(value);
    return PGT_OK;
  }
  if (pgt_spmm_tune(key, value)) return PGT_OK;
  pgt_set_error("pgt_tune: unknown key '%s'", key);
  return PGT_ERR_INVALID;
}
