// fp16 operands (precision=fp16; reference: torch.autocast(float16), utils.py:229-234): instantiation of the GEMM engine (gemm_impl.h)
#include "gemm_impl.h"

int launch_gemm_typed_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used) {
  return launch_typed<f16_t>(g, g1, epi, tile, st, ksplit_used);
}
