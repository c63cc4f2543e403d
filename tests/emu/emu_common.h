// tests/emu/*.cpp -- TEST INFRASTRUCTURE: executes the kernels' own per-lane phase functions
// (robotics-toolbox-python_amd/csrc/kin_tile.h, rne_device.h, ik_device.h -- all __host__ __device__)
// lane by lane on the CPU, in the same phase order and with the same LDS layout as the gfx950
// kernels.  The build container has no GPU; this lets `pytest -m "not gpu"` catch logic errors in the
// kernel bodies (indexing, staging, flush arithmetic, recursion order) before GPU minutes are spent.
// It is NOT a product path: librtbhip.so contains none of this and fails loudly without a GPU.
// emu_common.h -- what the translation units of tests/emu share (the replay is split by kernel family so that the units compile
// in parallel: one unit instantiating every kernel body for every joint count took eleven minutes).
#pragma once
#include "../../robotics-toolbox-python_amd/csrc/ik_device.h"
#include "../../robotics-toolbox-python_amd/csrc/rne_device.h"
#include "../../robotics-toolbox-python_amd/csrc/dyn_device.h"
#include "../../robotics-toolbox-python_amd/csrc/diff_device.h"
#include "../../robotics-toolbox-python_amd/csrc/tree_device.h"
#include "../../robotics-toolbox-python_amd/csrc/partial_device.h"
#include "../../robotics-toolbox-python_amd/csrc/frames_device.h"
#include "../../robotics-toolbox-python_amd/csrc/servo_device.h"
#include <vector>
#include <cstdio>
#include <cstdlib>

using namespace rtbhip;

// the chain view k_ik walks (ik_kernels.hip: ConstChainIk): segments, descriptors and the sincos constant table -- with `trig` present the
// replay takes the kernel's own template paths (table-driven sincos, the constant segments multiplied by structure class)
struct EmuChainIk {
    const DevSeg *seg;
    const int32_t *jmeta;
    const double *trig;
};
static inline EmuChainIk emu_chain_ik(const Chain *c)
{
    static const double tab[kSincosTableLen] = RTB_SINCOS_TABLE_INIT;
    const DevChain v = chain_host_view(c);
    return EmuChainIk{v.seg, v.jmeta, tab};
}

static inline Affine aff16(const double *m)
{
    Affine a;
    a.used = m != nullptr;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) a.v[4 * r + c] = m ? m[4 * r + c] : (r == c ? 1.0 : 0.0);
    return a;
}

// settings of the next emu_ik / emu_ik_wave calls (defined in emu_ik_seq.cpp)
extern double g_emu_ns[4];        // null-space terms kq, km, ps, (scalar pi) (kq <= 0: none)
extern double g_emu_pi[16];       // ... the influence distance per joint
extern double g_emu_ks;           // IK_QP (method 5): slack gain (kj is passed as lambda)
extern int64_t g_emu_target0;     // restart-generator key of row 0 (rtbhip_ik_target_base)

extern "C" int emu_kin(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16,
                       int frame, double *T, double *J, double *H, int coalesced);
