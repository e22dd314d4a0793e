// Fast DCT axis passes (LDS-resident FFT per line tile).  Placeholder until the LDS kernel lands:
// reports "unsupported" so dct.hip uses the direct O(N^2) kernels.
#include "ops.h"

namespace bk {

bool dct_axis_fft_supported(int) { return false; }

int dct_axis_fft(bk_ctx* ctx, int, int, int, int, int, const double*, const double*, double*, const double*,
                 const double*, const double*, double, int) {
    return set_error(ctx, "dct_axis_fft: not available");
}

}  // namespace bk
