// Fused-MLP kernels with run-time-dispatched hidden activation (exp / sine / sigmoid / squareplus / softplus / none):
// the second instantiation of ffmlp_kernels.h, in its own translation unit so that the two build in parallel.
#include "ffmlp_kernels.h"

namespace sdn_ff {

int launch_fused_generic(int mode, uint32_t W, const FfArgs &a, hipStream_t st) {
    switch (mode) {
        case 0: return launch_fused_t<0, false>(W, a, st);
        case 1: return launch_fused_t<1, false>(W, a, st);
        default: return launch_fused_t<2, false>(W, a, st);
    }
}

}  // namespace sdn_ff
