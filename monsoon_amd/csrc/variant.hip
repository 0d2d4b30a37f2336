// One instantiation of the hot kernel (kernels.h): compile with -DVAR_U=<candidate lanes per game>
// -DVAR_W=<waves per SIMD>.  Exports monsoon_variant_<U>_<W>(), the launch table monsoon_hip.hip uses.
#include "kernels.h"

using namespace msbk;

namespace {
hipError_t v_occupancy(int* blocks_per_cu, int lds_bytes) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_play<VAR_U, VAR_W>, 64, lds_bytes);
}
void v_play(int grid, int lds_bytes, hipStream_t stream, DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent,
            int parity) {
  hipLaunchKernelGGL((k_play<VAR_U, VAR_W>), dim3(grid), dim3(64), lds_bytes, stream, b, n, max_turns, rounds, write_scores, persistent, parity);
}
const VariantOps kOps = {VAR_U, VAR_W, DecideLds<VAR_U>::TOTAL, v_occupancy, v_play};
}  // namespace

#define MSB_CAT_(a, b, c) a##b##_##c
#define MSB_CAT(a, b, c) MSB_CAT_(a, b, c)
const VariantOps* MSB_CAT(monsoon_variant_, VAR_U, VAR_W)() { return &kOps; }
