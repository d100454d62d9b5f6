// LAB HARNESS (not shipped): csrc/seq64.hip with the phase timeline of workgroup 0 recorded (wall_clock64 ticks, 100 MHz), built as
// its own small shared library (scripts/build_lab.sh -> lab/libseq64_lab.so) and driven from Python (scripts/seq64_trace.py).
// marks per (step, gate): 0 T_0 complete | 1 hop 1 gathered | 2 products T0, T1o | 3 hop 2 (o) gathered | 4 product T2o |
//                         5 T1i back in LDS | 6 hop 2 (i) gathered | 7 products T1i, T2i | 8 gate chain done
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

__device__ long long* g_sq_trace = nullptr;
constexpr int SQ_TR_STEPS = 12, SQ_TR_SLOTS = 9;
#define SQ_MARK(t, G, slot)                                                                                          \
  do {                                                                                                               \
    if (g_sq_trace != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && (t) < SQ_TR_STEPS)                            \
      g_sq_trace[((t) * 2 + (G)) * SQ_TR_SLOTS + (slot)] = (long long)wall_clock64();                                 \
  } while (0)

#ifdef SQ_SKIP
#define SQ_LAB_SKIP(bit) (((SQ_SKIP) & (bit)) != 0)
#endif

static char g_err[512];
void pgt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sq_lab_last_error() { return g_err; }

#ifdef SQ_STAGGER_RECORD         // the forward with a gather phase and the products independent of it in one run of chunk barriers, the
#include "seq64_stagger_record.hip"   // wavefront pairs of a SIMD in opposite orders (measured, withdrawn: notebook 7.10): kept for same-box A/Bs
#else
#include "../pytorch_geometric_temporal_amd/csrc/seq64.hip"
#endif

// the loader wavefront's path alone, compile-only: `hipcc -Rpass-analysis=kernel-resource-usage` must report ScratchSize 0 for these
namespace {
template <int K>
__global__ __launch_bounds__(SQ_THREADS) void sq_lab_fwd_loader_only(Seq64Args a) {
  __shared__ __attribute__((aligned(16))) char smem[SQ_LDS];
  sq_fwd_body<true, K>(a, smem);
}
template <int K>
__global__ __launch_bounds__(SQ_THREADS) void sq_lab_bwd_loader_only(Seq64BwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[SQ_LDS];
  sq_bwd_body<true, K>(a, smem);
}
template __global__ void sq_lab_fwd_loader_only<3>(Seq64Args);
template __global__ void sq_lab_fwd_loader_only<2>(Seq64Args);
template __global__ void sq_lab_bwd_loader_only<3>(Seq64BwdArgs);
template __global__ void sq_lab_bwd_loader_only<2>(Seq64BwdArgs);
}  // namespace

extern "C" int sq_lab_set_trace(long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sq_trace), &buf, sizeof(buf));
}
