// tree_kernels.h -- what the two kernel files of the link-tree dynamics share (tree_kernels.hip: k_tree_rne; tree_dyn_kernels.hip: k_tree_dyn -- two
// translation units so that a clean build compiles them side by side: together they took 3 min 40 s, the long pole of the library).
#pragma once
#include "tree_device.h"
#include "kin_tile.h"
#ifndef __HIPCC_RTC__
#include <string>
#endif

namespace rtbhip {

typedef const __attribute__((address_space(4))) DevGroup *ConstGroups;

struct TreeParams {
    int32_t n, nslots;
    int32_t tile, pad_;           // k_tree_dyn: configurations per single-wave workgroup (64; 32 when a 64-lane tile would not fit a CU's LDS)
    int64_t N;
    double grav[3];
};

int tree_sig_enabled();      // tree_kernels.hip: rtbhip_tune("tree_sig")

}  // namespace rtbhip
