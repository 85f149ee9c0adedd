// fpx_build.hip -- GPU builder of seeded synthetic file segments (benchmarks / tests).
#include <cstring>
#include <hip/hip_runtime.h>
#include "fpx_internal.h"

namespace fpx {

int synth_segment_impl(Ctx*, uint64_t, uint32_t, uint32_t, uint32_t, int, uint32_t, uint64_t, Segment** out)
{
    *out = nullptr;
    set_error("fpx_synth_segment: not built yet");
    return FPX_E_INVAL;
}

}  // namespace fpx
