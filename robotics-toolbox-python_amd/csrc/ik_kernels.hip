// ik_kernels.hip -- placeholder until the on-device LM loop lands (next milestone).
#include "rtbhip_internal.h"
namespace rtbhip {
void ik_tune(const char *, int) {}
int launch_ik(const Chain *, const DevChain &, const double *, const double *, int64_t, const double *, const IkParams &,
              double *, int32_t *, int32_t *, int32_t *, double *, hipStream_t)
{
    set_error("ik_lm: not built yet");
    return RTBHIP_EINVAL;
}
void ik_restart_host(const Chain *, uint64_t, int64_t, int, double *) {}
}  // namespace rtbhip
