// The matrix-pipe tile (csrc/fused_mfma_kernel.h) for fp64 FLOAT4 eigenvectors: 16-line column tiles, 8-wave row tile.
#include "fused_mfma_kernel.h"

namespace mugiq {
int launch_mfma_tile_d4(const MTileArgs &a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream) {
  return launch_mfma_tile_t<double, 4, false>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
}
}  // namespace mugiq
