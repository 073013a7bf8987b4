// fp16 instantiation of the 256x256 / BK = 32 mainloop (gemm_wide_impl.h; precision=fp16, see gemm_ws_f16.hip)
#include "gemm_wide_impl.h"

int launch_gemm_wide_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  return launch_gemm_wide_t<f16_t>(g, g1, epi, tile, st);
}
