// One instantiation of the hot kernel: compile with -DVAR_U=<candidate lanes per game> -DVAR_W=<waves per SIMD>
// [-DVAR_G=<games per wavefront>].  G = 1 is k_play (kernels.h), 2 / 4 k_play_multi (kernels_multi.h), 10 k_play_reg (kernels_reg.h).  Exports
// monsoon_variant_<U>_<W>_<G>(), the launch table monsoon_hip.hip uses.
#include "kernels_multi.h"
#include "kernels_reg.h"

using namespace msbk;

#ifndef VAR_G
#define VAR_G 1
#endif

namespace {
#if VAR_G == 1
#define VAR_KERNEL k_play<VAR_U, VAR_W>
constexpr int kLds = DecideLds<VAR_U>::TOTAL;
#elif VAR_G == 10   // one game per wavefront, its current record in registers (kernels_reg.h)
#define VAR_KERNEL k_play_reg<VAR_U, VAR_W>
constexpr int kLds = RegLds<VAR_U>::TOTAL;
#else
#define VAR_KERNEL k_play_multi<VAR_U, VAR_G, VAR_W>
constexpr int kLds = MultiLds<VAR_U, VAR_G>::TOTAL;
#endif
hipError_t v_occupancy(int* blocks_per_cu, int lds_bytes) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, VAR_KERNEL, 64, lds_bytes);
}
void v_play(int grid, int lds_bytes, hipStream_t stream, DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent,
            int parity) {
  hipLaunchKernelGGL((VAR_KERNEL), dim3(grid), dim3(64), lds_bytes, stream, b, n, max_turns, rounds, write_scores, persistent, parity);
}
const VariantOps kOps = {VAR_U, VAR_W, VAR_G, VAR_G >= 10 ? 1 : VAR_G, kLds, v_occupancy, v_play};
}  // namespace

#define MSB_CAT_(a, b, c, d) a##b##_##c##_##d
#define MSB_CAT(a, b, c, d) MSB_CAT_(a, b, c, d)
const VariantOps* MSB_CAT(monsoon_variant_, VAR_U, VAR_W, VAR_G)() { return &kOps; }
