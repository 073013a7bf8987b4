// bf16 operands: instantiation of the GEMM engine (gemm_impl.h)
#include "gemm_impl.h"

int launch_gemm_typed_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used) {
  return launch_typed<bf16_t>(g, g1, epi, tile, st, ksplit_used);
}
