// bf16 instantiation of the 256x256 / BK = 32 mainloop (gemm_wide_impl.h)
#include "gemm_wide_impl.h"

int launch_gemm_wide_bf16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  return launch_gemm_wide_t<bf16_t>(g, g1, epi, tile, st);
}
