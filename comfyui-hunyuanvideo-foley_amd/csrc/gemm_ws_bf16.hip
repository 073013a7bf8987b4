// bf16 instantiation of the wave-specialised GEMM / conv3 mainloops (gemm_ws_impl.h)
#include "gemm_ws_impl.h"

int launch_gemm_ws_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  return launch_gemm_ws_t<bf16_t>(g, g1, epi, tile, st);
}
