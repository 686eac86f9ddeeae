// The instantiations of the neighbour-search kernels (see neighbours_inst.hpp); launched from hope_hip.hip.
#include "neighbours_inst.hpp"

namespace hope {
HOPE_NEIGHBOUR_KERNELS(HOPE_KERNEL_INSTANTIATE)
}  // namespace hope
