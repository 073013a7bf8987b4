// fp32 parity mode (v_mfma_f32_32x32x2_f32) and the DAC: instantiation of the GEMM engine (gemm_impl.h)
#include "gemm_impl.h"

int launch_gemm_typed_f32(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* ksplit_used) {
  return launch_typed<float>(g, g1, epi, tile, st, ksplit_used);
}
