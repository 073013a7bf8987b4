// fp16 instantiation of the wave-specialised GEMM / conv3 mainloops (gemm_ws_impl.h): precision=fp16 /
// fp16 checkpoints, which the reference runs under torch.autocast(float16) (nodes.py:89-106, utils.py:229-234)
#include "gemm_ws_impl.h"

int launch_gemm_ws_f16(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  return launch_gemm_ws_t<f16_t>(g, g1, epi, tile, st);
}
